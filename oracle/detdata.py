"""Deterministic, RNG-free tensors for parity tests.  TEST INFRASTRUCTURE ONLY (see cpu_ref.py).

Golden fixtures must be reproducible on the GPU box, where neither `/root/reference` nor the
torch RNG stream of the build container can be relied on.  Everything here is integer hashing
done in int64 and converted once to float32, so the same call yields the same bits anywhere.
"""
import math

import numpy as np
import torch


def det_uniform(shape, salt):
    """U[0,1) float32 tensor from a multiplicative integer hash of the flat index and `salt`."""
    n = int(np.prod(shape)) if len(shape) else 1
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over='ignore'):
        h = idx + np.uint64((int(salt) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)     # wraps mod 2^64
        h ^= h >> np.uint64(33)
        h = h * np.uint64(0xFF51AFD7ED558CCD)
        h ^= h >> np.uint64(33)
        h = h * np.uint64(0xC4CEB9FE1A85EC53)
        h ^= h >> np.uint64(33)
    u = (h >> np.uint64(40)).astype(np.float64) / float(1 << 24)      # 24 random bits -> exact in fp32
    return torch.from_numpy(u.astype(np.float32)).reshape(shape)


def det_fill(net, salt=1):
    """Fill every parameter / BN buffer of `net` with hash-derived values of training-like scale.

    Linear/conv weights ~ U(-a, a) with a = 1.2 * sqrt(3 / fan_in) (fan_in counted per output element, so
    activations stay O(1) through deep stacks), biases ~ U(-0.1, 0.1) (non-zero so bias paths are
    exercised), BN weight ~ 1 +- 0.2, BN bias +-0.1, running_mean +-0.1, running_var 1 +- 0.2,
    num_batches_tracked = 0.  Modules are visited in `named_modules()` order; the salt of a tensor depends
    only on that order, so two networks with identical structure get identical values.
    """
    k = 0
    with torch.no_grad():
        for _, m in net.named_modules():
            kind = type(m).__name__
            if kind not in ('Conv2d', 'ConvTranspose2d', 'Linear', 'BatchNorm2d'):
                continue
            s = salt * 1000003 + k * 7919 + 13
            k += 1
            if kind == 'BatchNorm2d':
                m.weight.copy_(1.0 + (det_uniform(m.weight.shape, s) - 0.5) * 0.4)
                m.bias.copy_((det_uniform(m.bias.shape, s + 1) - 0.5) * 0.2)
                m.running_mean.copy_((det_uniform(m.running_mean.shape, s + 2) - 0.5) * 0.2)
                m.running_var.copy_(1.0 + (det_uniform(m.running_var.shape, s + 3) - 0.5) * 0.4)
                m.num_batches_tracked.zero_()
                continue
            w = m.weight
            if kind == 'Linear':
                fan_in = w.shape[1]
            elif kind == 'Conv2d':
                fan_in = w.shape[1] * w.shape[2] * w.shape[3]
            else:
                fan_in = w.shape[0] * w.shape[2] * w.shape[3] / float(m.stride[0] * m.stride[1])
            a = 1.2 * math.sqrt(3.0 / max(fan_in, 1.0))
            w.copy_((det_uniform(w.shape, s) - 0.5) * 2.0 * a)
            if m.bias is not None:
                m.bias.copy_((det_uniform(m.bias.shape, s + 1) - 0.5) * 0.2)
    return net


def det_normal(shape, salt, terms=8):
    """Approximately N(0, 1) float32 tensor without an RNG and without libm: the sum of `terms` hash uniforms (24 random bits each, so
    the sum is EXACT in float64 on any machine), centred and scaled to unit variance (Irwin-Hall).  Bit-reproducible anywhere."""
    acc = None
    for i in range(terms):
        u = det_uniform(shape, salt * 131 + i + 1).double()
        acc = u if acc is None else acc + u
    return ((acc - terms / 2.0) * math.sqrt(12.0 / terms)).float()


def det_init_fill(net, salt=1, gain=0.02, res_gain=1.41):
    """Weights with the STATISTICS of the reference's start of training (main.py:119-138 -> utils.py:88-109 `init_net`): encoders and decoder
    `normal` with gain 0.02 -- conv / linear weights N(0, 0.02), biases 0, BatchNorm weight N(1, 0.02), bias 0; the integrator `orthogonal`
    with gain 1.41 -- restated here as N(0, res_gain^2 / max(rows, cols)) entries (an orthogonal matrix of that gain has entries of exactly
    this variance; a QR factorisation would tie the fixture to one LAPACK build).  RNG-free (det_normal): identical bits on any machine.
    BatchNorm running statistics keep their constructor values (0 / 1), as after `init_net`."""
    k = 0
    with torch.no_grad():
        for name, m in net.named_modules():
            kind = type(m).__name__
            if kind not in ('Conv2d', 'ConvTranspose2d', 'Linear', 'BatchNorm2d'):
                continue
            s = salt * 1000003 + k * 7919 + 17
            k += 1
            in_res = name.startswith('t_resnet')
            if kind == 'BatchNorm2d':
                m.weight.copy_(1.0 + det_normal(m.weight.shape, s) * (res_gain if in_res else gain))
                m.bias.zero_()
                m.running_mean.zero_()
                m.running_var.fill_(1.0)
                m.num_batches_tracked.zero_()
                continue
            w = m.weight
            if in_res:
                rows, cols = w.shape[0], w[0].numel()
                std = res_gain / math.sqrt(max(rows, cols))
            else:
                std = gain
            w.copy_(det_normal(w.shape, s) * std)
            if m.bias is not None:
                m.bias.zero_()
    return net


def checksum(t):
    """(sum, L2, 16 strided samples) of a tensor in float64 -- pins tensors too large to commit."""
    f = t.detach().double().flatten()
    n = f.numel()
    pick = torch.linspace(0, n - 1, 16).long().clamp_(max=n - 1) if n > 16 else torch.arange(n)    # fp32 linspace can round n-1 up
    return np.concatenate([[f.sum().item(), f.norm().item()], f[pick].numpy()]).astype(np.float64)
