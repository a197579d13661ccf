"""WaveEq datasets resident in HBM, batches assembled on the device (reference: var_sep/data/wave_eq.py:29-90).

The reference keeps the normalised simulations in host memory and lets a DataLoader slice one window per item and stack them.
At this path's step rate (128 sequences x 25 frames x 16 KB per 1.8 ms = 29 GB/s) no host loader keeps up, and the whole
set is small next to 288 GB of HBM: here the simulations live on the GPU as one [n_seq, nt, H*W] fp32 tensor and a batch is
ONE gather launch (`vs_gather_windows`) driven by the sampler's item indices.  Item numbering, file selection,
normalisation and the pixel subset of `WaveEqPartial` follow the reference line by line -- including its two quirks: the
train/test split reads the first integer of the file's FULL PATH (wave_eq.py:25-26, 43-46), and `__len__` counts
windows with the frame HEIGHT where `__getitem__` uses the sequence length (wave_eq.py:62-65) -- so that a seeded sampler
visits the same windows in the same order.  `DeviceBatchLoader` replaces `DataLoader(dataset, shuffle=True)` (main.py:113):
the index stream comes from torch's own RandomSampler / BatchSampler, only the 4-byte indices cross PCIe.
"""
import os
import re

import numpy as np
import torch
from torch.utils.data import BatchSampler, RandomSampler, SequentialSampler

from .. import ops
from .._lib import VarsepHipError


def extract_id(string):
    """First run of digits of the string (wave_eq.py:25-26) -- applied to the joined path, like the reference."""
    return int(re.findall(r'\d+', string)[0])


class WaveEq:
    """`WaveEq(data_dir, nt_cond, seq_len, train, downsample)` (wave_eq.py:29-72) with the data on `device`."""

    device_resident = True

    def __init__(self, data_dir, nt_cond, seq_len, train, downsample, device=None):
        self.nt_cond, self.seq_len, self.train, self.downsample = nt_cond, seq_len, train, downsample
        device = torch.device(device if device is not None else 'cuda')
        if device.type != 'cuda':
            raise VarsepHipError('the HBM-resident WaveEq set needs an MI355X device; there is no CPU fallback '
                                 '(tests use oracle/wave_data_ref.py for CPU results)')
        base_path = os.path.join(data_dir, 'data')
        files = [os.path.join(base_path, f) for f in os.listdir(base_path)]
        max_seq = int(0.8 * len(files))
        files = [f for f in files if (extract_id(f) < max_seq) == bool(train)]
        self.size = len(files)
        sims = []
        for file in files:
            data = torch.load(file).get('simul')
            max_, min_ = data.max(), data.min()
            data = (data - min_) / (max_ - min_)                    # per-file min-max (wave_eq.py:55-57), on the host in fp32
            sims.append(data[::downsample])
        if not sims:
            raise ValueError('no simulation files selected in %s' % base_path)
        if any(s.shape != sims[0].shape for s in sims):
            raise ValueError('the simulations must share one [nt, H, W] shape to live in one HBM tensor')
        self.nt = len(sims[-1])
        self.full_seq_len = sims[-1][0].size(0)                      # wave_eq.py:62: the frame height, not nt
        self.frame_shape = tuple(sims[0].shape[1:])
        self.all_data = torch.stack(sims).reshape(self.size, self.nt, -1).to(device).contiguous()
        self.windows_per_seq = self.nt + 1 - self.seq_len            # wave_eq.py:69-70
        if self.windows_per_seq < 1:
            raise ValueError('seq_len %d exceeds the %d frames of a simulation' % (seq_len, self.nt))
        self._pixels = None

    def __len__(self):
        return self.size * (self.full_seq_len - self.seq_len + 1)

    def _check(self, idx_max):
        if idx_max >= self.size * self.windows_per_seq:
            raise IndexError('item %d is outside the %d windows of the set' % (idx_max, self.size * self.windows_per_seq))

    def batch(self, item_idx, out_dtype=torch.float32):
        """item_idx: int32 device tensor [B] (or a list of ints) -> (cond [B, nt_cond, 1, ...], target [B, seq_len-nt_cond, 1, ...])."""
        if not isinstance(item_idx, torch.Tensor):
            self._check(max(item_idx))
            item_idx = torch.tensor(list(item_idx), dtype=torch.int32).to(self.all_data.device, non_blocking=True)
        x = ops.gather_windows(self.all_data, item_idx, self.windows_per_seq, self.seq_len, self._pixels, out_dtype)
        tail = (self._pixels.numel(),) if self._pixels is not None else self.frame_shape
        x = x.view((x.shape[0], self.seq_len, 1) + tuple(tail))
        return x[:, :self.nt_cond], x[:, self.nt_cond:]

    def __getitem__(self, idx):
        cond, target = self.batch([int(idx)])
        return cond[0], target[0]


class WaveEqPartial(WaveEq):
    """`WaveEqPartial(..., n_pixels)` (wave_eq.py:75-90): every frame reduced to n_pixels fixed (row, column) positions."""

    def __init__(self, data_dir, nt_cond, seq_len, train, downsample, n_pixels, device=None):
        super().__init__(data_dir, nt_cond, seq_len, train, downsample, device=device)
        pixels = np.load(os.path.join(data_dir, 'pixels', 'pixels.npz'), allow_pickle=True)
        self.rand_w, self.rand_h = pixels['rand_w'], pixels['rand_h']
        self.n_wave_points = n_pixels
        H, W = self.frame_shape
        rows = np.asarray(self.rand_w[:n_pixels]).astype(np.int64)
        cols = np.asarray(self.rand_h[:n_pixels]).astype(np.int64)
        rows = np.where(rows < 0, rows + H, rows)                    # numpy-style negative indices, as advanced indexing allows
        cols = np.where(cols < 0, cols + W, cols)
        if rows.size and (rows.max() >= H or cols.max() >= W):
            raise IndexError('pixel table addresses positions outside the %dx%d frames' % (H, W))
        self._pixels = torch.from_numpy((rows * W + cols).astype(np.int32)).to(self.all_data.device)


class DeviceBatchLoader:
    """`DataLoader(dataset, batch_size, shuffle=True)` of main.py:113 for a device-resident set: same sampler classes (so the
    same seeded index stream and the same ragged last batch), batches gathered by one launch instead of worker processes."""

    def __init__(self, dataset, batch_size, shuffle=True, drop_last=False, sampler=None, generator=None, out_dtype=torch.float32):
        assert getattr(dataset, 'device_resident', False), 'DeviceBatchLoader serves HBM-resident datasets'
        self.dataset, self.batch_size, self.drop_last, self.out_dtype = dataset, batch_size, drop_last, out_dtype
        if sampler is None:
            sampler = RandomSampler(dataset, generator=generator) if shuffle else SequentialSampler(dataset)
        self.sampler, self.generator = sampler, generator
        self.batch_sampler = BatchSampler(sampler, batch_size, drop_last)

    def __len__(self):
        return len(self.batch_sampler)

    def __iter__(self):
        # a DataLoader iterator draws its base seed from the RNG before the sampler draws its own (torch/utils/data/dataloader.py,
        # _BaseDataLoaderIter.__init__); consume the same draw so that a seeded epoch visits the same items
        torch.empty((), dtype=torch.int64).random_(generator=self.generator)
        for items in self.batch_sampler:
            yield self.dataset.batch(items, self.out_dtype)
