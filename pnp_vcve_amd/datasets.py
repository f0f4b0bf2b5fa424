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


def collate(batch):
    """samples_per_gpu=1 (forced by the reference, tools/test.py:110): add the batch dim."""
    out = {}
    for k in batch[0]:
        if k == 'meta':
            out[k] = [b[k] for b in batch]
        else:
            out[k] = torch.stack([b[k] for b in batch])
    return out


def build_dataset(cfg, default_args=None):
    from .registry import build_from_cfg
    return build_from_cfg(cfg, DATASETS, default_args)
