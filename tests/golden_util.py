"""Helpers to replay tests/golden/*.npz (vectors produced by the reference, see oracle/make_golden.py)."""
import os

import numpy as np
import torch

from oracle.detdata import checksum

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + '.npz')) as z:
        return {k: z[k] for k in z.files}


def rel_err(a, b):
    a = torch.as_tensor(a).double().flatten()
    b = torch.as_tensor(b).double().flatten()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def check_tensor(gold, key, t, tol, what='', atol=0.0):
    """Compare tensor `t` with the stored vector (full) or its checksum; returns the worst relative error.  `atol`: elements may
    additionally differ by this much in absolute terms (post-Adam parameters: an element whose gradient is at the 1e-8 noise level
    moves by up to lr in a noise-determined direction, in the reference and here alike)."""
    t = t.detach().float().cpu()
    if key in gold:
        ref = torch.from_numpy(gold[key]).float()
        assert tuple(ref.shape) == tuple(t.shape), f'{what}{key}: shape {tuple(t.shape)} vs golden {tuple(ref.shape)}'
        e = rel_err(t, ref)
        if atol and e > tol and float((t - ref).abs().max()) <= atol:
            return tol
        assert e <= tol, f'{what}{key}: relative L2 error {e:.3e} > {tol:.1e}'
        return e
    ck = 'cs:' + key
    assert ck in gold, f'{what}{key}: not in golden file'
    ref = gold[ck]
    got = checksum(t)
    scale = max(abs(ref[1]), 1e-30)                  # L2 norm of the golden tensor
    # sum can cancel: compare it on the scale of the norm * sqrt(n); samples on the scale of the max sample
    n = t.numel()
    e_l2 = abs(got[1] - ref[1]) / scale
    e_sum = abs(got[0] - ref[0]) / (scale * np.sqrt(n))
    smax = max(np.abs(ref[2:]).max(), scale / np.sqrt(n))
    e_smp = np.abs(got[2:] - ref[2:]).max() / smax
    if atol and np.abs(got[2:] - ref[2:]).max() <= atol:
        e_smp = 0.0
    e = max(e_l2, e_sum, e_smp * 0.25)
    assert e <= tol, f'{what}{key}: checksum mismatch l2={e_l2:.2e} sum={e_sum:.2e} samples={e_smp:.2e} > {tol:.1e}'
    return e
