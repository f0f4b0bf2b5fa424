"""Test loops (mmedit/apis/test.py:13-126) for the hot path: one process per GPU, each runs its
shard of clips through model(test_mode=True, **data); metrics are gathered with one all-gather."""
import torch

from .datasets import collate
from .dist import gather_clip_metrics, get_dist_info, shard_indices


def _to_device(data, device):
    """Host -> device; when the clip carries raw decoder MV records (CompressedClipFolderDataset) the dense
    motion / partition maps are painted on the GPU (pnp_rasterise_side_info_f32) instead of on the host."""
    out = {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in data.items()}
    for k in ('lq', 'gt'):
        if k + '_u8' in out:       # frames uploaded as uint8 (dataset.get_uint8): RescaleToZeroOne + FramesToTensor on the device.
            # The 256 possible values come from the host's own fp32 division (a device-side `x / 255.0` multiplies by a rounded
            # reciprocal and is not bit-equal): a table lookup reproduces numpy's bits by construction
            u8 = out.pop(k + '_u8')                                    # (n, T, H, W, 3)
            lut = (torch.arange(256, dtype=torch.float32) / 255.0).to(u8.device)
            n, t, h, w, c = u8.shape
            f = torch.empty((n, t, c, h, w), dtype=torch.float32, device=u8.device)
            for i in range(n):
                for j in range(t):                                   # frame by frame: the int64 index of a 100-frame 720p clip is 2.2 GB
                    f[i, j] = lut[u8[i, j].permute(2, 0, 1).long()]
            out[k] = f
    if 'mv_records' in out:
        from .ops import rasterise_side_info
        rec, rf = out.pop('mv_records'), out.pop('rec_frame')
        t, h, w = out['lq'].shape[1], out['lq'].shape[-2], out['lq'].shape[-1]
        sl = [float(v) for v in data['slices'].reshape(-1)[:t]]
        mvs, par = rasterise_side_info(rec.float(), rf.int(), sl, h, w)
        out['mvs'], out['partitions'] = mvs.unsqueeze(0), par.unsqueeze(0)
    return out


class ClipPrefetcher:
    """Streams clips to the GPU one ahead of the compute: a worker thread reads /
    decodes clip i+1 on the host and its H2D copies run on a side HIP stream from pinned memory while clip i
    is being enhanced on the compute stream.  The reference's loop is strictly serial per rank (DataLoader ->
    scatter -> forward -> evaluate, mmedit/apis/test.py:100-119); with whole 100-frame 720p clips the input
    of one clip is 3.7 GB (lq + mvs + partitions), i.e. ~60 ms of PCIe time that this hides."""

    def __init__(self, dataset, indices, device, depth=1):
        import queue
        import threading
        self.dataset, self.indices, self.device = dataset, list(indices), torch.device(device)
        self.cuda = self.device.type == 'cuda'
        self.stream = torch.cuda.Stream(self.device) if self.cuda else None
        self.q = queue.Queue(maxsize=max(1, int(depth)))      # clips staged ahead of the consumer
        self.th = threading.Thread(target=self._work, daemon=True)
        self.th.start()

    def _work(self):
        try:
            u8 = self.cuda and hasattr(self.dataset, 'get_uint8')      # frames travel as uint8, /255 + HWC->CHW on the device
            for i in self.indices:
                data = collate([self.dataset.get_uint8(i) if u8 else self.dataset[i]])
                if self.cuda:
                    data = {k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in data.items()}
                    with torch.cuda.stream(self.stream):
                        dev = _to_device(data, self.device)
                        ev = torch.cuda.Event()
                        ev.record(self.stream)
                    self.q.put((dev, ev, data))           # keep the pinned host copies alive until consumed
                else:
                    self.q.put((_to_device(data, self.device), None, None))
            self.q.put(None)
        except BaseException as e:                        # surface loader errors in the consumer
            self.q.put(e)

    def __iter__(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            dev, ev, _host = item
            if ev is not None:
                torch.cuda.current_stream(self.device).wait_event(ev)
                for v in dev.values():
                    if torch.is_tensor(v):
                        v.record_stream(torch.cuda.current_stream(self.device))
            yield dev


def single_gpu_test(model, dataset, save_image=False, save_path=None, device='cuda'):
    model.eval()
    results = []
    for i in range(len(dataset)):
        data = _to_device(collate([dataset[i]]), device)
        with torch.no_grad():
            results.append(model(test_mode=True, save_image=save_image, save_path=save_path, **data))
    return results


GENERATOR_INPUTS = ('lq', 'QPs', 'slices', 'mvs', 'base_QPs', 'partitions')


def _pairable(a, b):
    """two clips can go through the generator as one batch: same shapes of everything it reads"""
    return all(torch.is_tensor(a.get(k)) and torch.is_tensor(b.get(k)) and a[k].shape == b[k].shape and a[k].device == b[k].device
               for k in GENERATOR_INPUTS)


def multi_gpu_test(model, dataset, save_image=False, save_path=None, device='cuda', metrics=('PSNR', 'SSIM'), clips_in_flight=1):
    """Returns, on every rank, the ordered per-clip results [{'eval_result': {...}}, ...].

    clips_in_flight = 2: two clips of equal shape are enhanced by ONE generator call (a batch of two), which the generator runs
    interleaved on two streams -- each clip's conv launches fill the partial last round of tiles and the dispatch gaps of the
    other's (+5 % at 720p, bit-identical to one clip at a time); metrics and image saving stay clip by
    clip (the reference evaluates with samples_per_gpu=1, mmedit/apis/test.py:100-119).  Costs a second workspace.
    `frames_per_s` of a PAIRED clip is the pair's throughput (frames of both clips over the pair's wall time), not the clip's own
    forward time; an unpaired leftover clip (odd count, shape change, sparse_val model) reports its own."""
    import time
    model.eval()
    rank, world = get_dist_info()
    mine = shard_indices(len(dataset), rank, world)
    local = []
    dev = torch.device(device)
    pairs = int(clips_in_flight) >= 2 and dev.type == 'cuda' and hasattr(model, 'generator') and not getattr(model, 'psnr_only', False)
    # a sparse_val generator evaluates one sample per call in eval mode (the reference indexes feature[0, ...],
    # sr_backbone_utils.py:262-275; generator.py refuses n != 1): such a model keeps the clip-by-clip loop
    pairs = pairs and not getattr(model.generator, 'sparse_val', False)

    def finish(data, out=None, fps=None):
        with torch.no_grad():
            kw = {} if out is None else {'precomputed_output': out}
            res = model(test_mode=True, save_image=save_image, save_path=save_path, **kw, **data)
        if fps is None:
            fps = data['lq'].shape[1] / model.last_forward_seconds if getattr(model, 'last_forward_seconds', None) else 0.0
        local.append([float(res['eval_result'].get(m, float('nan'))) for m in metrics] + [fps])

    held = None
    for data in ClipPrefetcher(dataset, mine, device, depth=2 if pairs else 1):
        if not pairs:
            finish(data)
            continue
        if held is None:
            held = data
            continue
        if not _pairable(held, data):
            finish(held)
            held = data
            continue
        both = [torch.cat([held[k], data[k]]) for k in GENERATOR_INPUTS]
        with torch.no_grad():
            torch.cuda.synchronize()
            t0 = time.time()
            out = model.generator(*both)
            torch.cuda.synchronize()
            dt = time.time() - t0
        model.last_forward_seconds = dt
        fps = (held['lq'].shape[1] + data['lq'].shape[1]) / dt
        finish(held, out[0:1], fps)
        finish(data, out[1:2], fps)
        held = None
    if held is not None:
        finish(held)
    table = gather_clip_metrics(local, len(dataset), device=device)
    return [dict(eval_result={m: float(table[i, j]) for j, m in enumerate(metrics)}, frames_per_s=float(table[i, -1]))
            for i in range(len(dataset))]
