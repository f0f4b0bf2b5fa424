"""Datasets for the test path.

`SyntheticCompressedClipDataset` produces clips with the tensor contract of the reference's
test pipeline (configs/HR_davis_LR_128x128.py:109-131 ->
lq, gt, mvs, slices, QPs, base_QPs, partitions + meta) from seeds -- there are no datasets in
this environment.  The reference's on-disk loader (LoadImageFromFileList_ipb) is the "next" row
SURVEY.md section 8(f)-1.
"""
from collections import defaultdict

import torch

from . import synthetic as syn
from .registry import DATASETS


class _EvalMixin:
    def evaluate(self, results, logger=None):
        """mmedit/datasets/base_sr_dataset.py:61-93: mean over clips of each metric."""
        if not isinstance(results, list):
            raise TypeError(f'results must be a list, but got {type(results)}')
        assert len(results) == len(self), (
            f'The length of results is not equal to the dataset len: {len(results)} != {len(self)}')
        results = [res['eval_result'] for res in results]
        acc = defaultdict(list)
        for res in results:
            for metric, val in res.items():
                acc[metric].append(val)
        return {metric: sum(values) / len(self) for metric, values in acc.items()}


@DATASETS.register_module()
class SyntheticCompressedClipDataset(_EvalMixin, torch.utils.data.Dataset):
    def __init__(self, num_clips=8, num_input_frames=7, height=128, width=128, slices='IBBBP', qp_mode='qp',
                 crfs=(15, 25, 35), seed=0, repeat=1, test_mode=True, pipeline=None, **unused):
        self.n, self.t, self.h, self.w = num_clips * repeat, num_input_frames, height, width
        self.slices, self.qp_mode, self.crfs, self.seed = slices, qp_mode, tuple(crfs), seed
        self.base_clips = num_clips

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        i = idx % self.base_clips
        clip = syn.make_clip(seed=self.seed * 100003 + i, n=1, t=self.t, h=self.h, w=self.w, slices=self.slices,
                             qp_mode=self.qp_mode, crf=self.crfs[i % len(self.crfs)],
                             block=8 if (self.h % 8 == 0 and self.w % 8 == 0) else 4)
        out = {k: torch.from_numpy(v[0]) for k, v in clip.items()}
        out['meta'] = dict(key=f'{i:03d}/{0:08d}', lq_path=f'synthetic/{i:03d}', gt_path=f'synthetic/{i:03d}')
        return out


@DATASETS.register_module(name='SRREDSMultipleGTCompressDataset')
@DATASETS.register_module()
class CompressedClipFolderDataset(_EvalMixin, torch.utils.data.Dataset):
    """On-disk clips in the reference's layout (mmedit/datasets/sr_reds_multiple_gt_compress_dataset.py +
    pipelines GenerateSegmentIndices_LR / LoadImageFromFileList_ipb / LoadImageFromFileList /
    RescaleToZeroOne / FramesToTensor, configs/HR_davis_LR_128x128.py:109-131):

        <lq_folder>/<clip>/<%08d>.png        e.g. dataset/REDS_test_HR/crf25/png/000/00000000.png
        <lq_folder with 'png'->'mv'>/<clip>/<%08d>.npy   decoder MV records, rows
              (direction, w, h, x_w, y_w, x, y, motion_x, motion_y, scale)
        qp_slice_file: JSON  {crfXX: {clip: {frame: {'slice': 'I|P|B', 'QP': q}}}}
        <gt_folder>/<clip>/<%08d>.png

    Frames, QP / slice side info and the raw MV records are read on the host; the dense motion and
    partition maps are NOT built here -- the records travel to the GPU and pnp_rasterise_side_info_f32
    paints them there (apis.prepare_batch).  Clips are discovered on disk (REDS4 names if present)."""

    REDS4 = ['000', '011', '015', '020']

    def __init__(self, lq_folder, gt_folder, num_input_frames=100, pipeline=None, scale=1, val_partition='REDS4',
                 repeat=1, cprs_folder=None, test_mode=True, qp_slice_file=None, replace_qp_withIPB=False, **unused):
        import json
        import os
        if not isinstance(repeat, int):
            raise TypeError(f'"repeat" must be an integer, but got {type(repeat)}.')
        self.lq_folder, self.gt_folder = str(lq_folder), str(gt_folder)
        self.num_input_frames = num_input_frames
        # the loader options live inside the pipeline list in the reference configs
        for step in (pipeline or []):
            if isinstance(step, dict) and step.get('type', '').startswith('LoadImageFromFileList_ipb'):
                qp_slice_file = step.get('qp_slice_file', qp_slice_file)
                replace_qp_withIPB = step.get('replace_qp_withIPB', replace_qp_withIPB)
        self.replace_qp = bool(replace_qp_withIPB)
        self.table = None
        if qp_slice_file is not None:
            with open(qp_slice_file) as f:
                self.table = json.load(f)
        clips = sorted(d for d in os.listdir(self.lq_folder) if os.path.isdir(os.path.join(self.lq_folder, d)))
        if val_partition == 'REDS4' and all(c in clips for c in self.REDS4):
            clips = list(self.REDS4)
        self.keys = clips * repeat

    def __len__(self):
        return len(self.keys)

    @staticmethod
    def _png(path):
        import numpy as np
        from PIL import Image
        return np.asarray(Image.open(path).convert('RGB'), dtype=np.uint8)

    #: PNG / record files of a clip are decoded on this many threads (PIL and numpy release the GIL while they inflate / read)
    decode_workers = 8

    def _pool(self):
        """the decode pool of THIS process: a forked DataLoader worker inherits the parent's executor object without its threads
        (futures submitted to it would never complete), so the pool is keyed on the pid and re-created after a fork"""
        import os
        pool = getattr(self, '_decode_pool', None)
        if pool is None or pool[0] != os.getpid():
            from concurrent.futures import ThreadPoolExecutor
            pool = self._decode_pool = (os.getpid(), ThreadPoolExecutor(max_workers=self.decode_workers,
                                                                        thread_name_prefix='pnp-decode'))
        return pool[1]

    def __getstate__(self):
        """picklable for spawn-started DataLoader workers: the executor stays behind"""
        state = dict(self.__dict__)
        state.pop('_decode_pool', None)
        return state

    def _read_clip(self, idx):
        """host side of one clip: uint8 HWC frames, side info and raw MV records (files decoded in parallel)"""
        import os
        import numpy as np
        key = self.keys[idx]
        d = os.path.join(self.lq_folder, key)
        names = sorted(f for f in os.listdir(d) if f.endswith('.png'))[:self.num_input_frames]
        crf_dir = self.lq_folder.rstrip('/').split('/')[-2] if '/' in self.lq_folder.rstrip('/') else ''
        base_qp = int(crf_dir.split('crf')[1]) if 'crf' in crf_dir else 0          # loading_ipb.py:239
        pool = self._pool()
        lq_f = [pool.submit(self._png, os.path.join(d, name)) for name in names]
        gt_f = [pool.submit(self._png, os.path.join(self.gt_folder, key, name)) for name in names]
        mv_f = [pool.submit(np.load, os.path.join(d, name).replace('.png', '.npy').replace('png', 'mv'))    # loading_ipb.py:326
                for name in names]
        slices, qps = [], []
        for name in names:
            frame = str(int(name.split('.')[0]))
            if crf_dir.startswith('crf') and self.table is not None:                # loading_ipb.py:298-312
                e = self.table[crf_dir][key][frame]
                sl = e['slice']
                qp = ord(sl) if self.replace_qp else e['QP']
            else:
                sl = 'I' if frame == '0' else 'P'
                qp = ord(sl) if self.replace_qp else 0.0
            slices.append(float(ord(sl)))
            qps.append(float(qp) / 255.0)
        recs = [f.result().astype(np.float32).reshape(-1, 10) for f in mv_f]
        rec_frame = [np.full((r.shape[0],), t, np.int32) for t, r in enumerate(recs)]
        T = len(names)
        side = dict(slices=torch.tensor(slices, dtype=torch.float32).view(T, 1, 1, 1),
                    QPs=torch.tensor(qps, dtype=torch.float32).view(T, 1, 1, 1),
                    base_QPs=torch.full((T, 1, 1, 1), base_qp / 255.0, dtype=torch.float32),
                    mv_records=torch.from_numpy(np.concatenate(recs) if recs else np.zeros((0, 10), np.float32)),
                    rec_frame=torch.from_numpy(np.concatenate(rec_frame) if rec_frame else np.zeros((0,), np.int32)),
                    meta=dict(key=f'{key}/{0:08d}', lq_path=d, gt_path=os.path.join(self.gt_folder, key)))
        return np.stack([f.result() for f in lq_f]), np.stack([f.result() for f in gt_f]), side

    def __getitem__(self, idx):
        """the reference pipeline's sample: RescaleToZeroOne + FramesToTensor give float32 (T,3,H,W) frames in [0,1]"""
        import numpy as np
        lq, gt, side = self._read_clip(idx)
        f32 = lambda a: torch.from_numpy(a.astype(np.float32) / 255.0).permute(0, 3, 1, 2).contiguous()
        return dict(lq=f32(lq), gt=f32(gt), **side)

    def get_uint8(self, idx):
        """the same sample with the frames still uint8 (T,H,W,3) under lq_u8 / gt_u8: a quarter of the host and PCIe bytes; the
        consumer (apis.ClipPrefetcher) divides by 255 and transposes on the device -- the same IEEE fp32 division, bit-identical"""
        lq, gt, side = self._read_clip(idx)
        return dict(lq_u8=torch.from_numpy(lq), gt_u8=torch.from_numpy(gt), **side)


def collate(batch):
    """samples_per_gpu=1 (forced by the reference, tools/test.py:110): add the batch dim."""
    out = {}
    for k in batch[0]:
        if k == 'meta':
            out[k] = [b[k] for b in batch]
        elif k in ('mv_records', 'rec_frame'):      # ragged per clip: samples_per_gpu is 1
            assert len(batch) == 1
            out[k] = batch[0][k]
        else:
            out[k] = torch.stack([b[k] for b in batch])
    return out


def build_dataset(cfg, default_args=None):
    from .registry import build_from_cfg
    return build_from_cfg(cfg, DATASETS, default_args)
