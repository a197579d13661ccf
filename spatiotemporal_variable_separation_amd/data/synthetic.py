"""Seeded synthetic sequences with the shape, dtype and value range of each reference dataset.

The reference's datasets (var_sep/data/*.py) are host-side numpy/IO outside the MI355X hot path (SURVEY.md section 2,
row 10) and need files and packages that are not available offline; training, the benchmark and the tests therefore
draw batches of identical shape: U[0,1) float32 frames (Moving-MNIST /255, WaveEq and TaxiBJ min-max), or N(0,1) for
the z-scored SST data.
"""
import torch
from torch.utils.data import Dataset

SHAPES = {'mnist': [1, 64, 64], 'chairs': [3, 64, 64], 'taxibj': [2, 32, 32], 'sst': [1, 64, 64], 'wave': [1, 64, 64]}
LAST_ACTIVATION = {'mnist': 'sigmoid', 'chairs': 'sigmoid', 'taxibj': None, 'sst': None, 'wave': 'sigmoid',
                   'wave_partial': 'sigmoid'}


def data_shape(name, n_wave_points=100):
    return [1, n_wave_points] if name == 'wave_partial' else list(SHAPES[name])


class SyntheticSequences(Dataset):
    def __init__(self, name, nt_cond, nt_pred, length=2048, seed=1234, n_wave_points=100):
        self.shape = data_shape(name, n_wave_points)
        self.nt_cond, self.nt_pred, self.length, self.seed = nt_cond, nt_pred, length, seed
        self.normal = name == 'sst'

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed * 1000003 + index)
        shape = [self.nt_cond + self.nt_pred] + self.shape
        x = torch.randn(shape, generator=g) if self.normal else torch.rand(shape, generator=g)
        return x[:self.nt_cond], x[self.nt_cond:]


def synthetic_batch(name, batch, nt_cond, nt_pred, device='cpu', seed=1234, n_wave_points=100):
    """One (cond, target) batch: `torch.manual_seed(seed)`-style generator, per SURVEY.md section 8d."""
    g = torch.Generator().manual_seed(seed)
    shape = data_shape(name, n_wave_points)
    draw = torch.randn if name == 'sst' else torch.rand
    cond = draw([batch, nt_cond] + shape, generator=g)
    target = draw([batch, nt_pred] + shape, generator=g)
    return cond.to(device), target.to(device)
