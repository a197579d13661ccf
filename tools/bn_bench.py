"""Micro-benchmark of the BatchNorm kernels at the layer sizes of the Moving-MNIST B=128 step (decoder: 16 calls batched as
16 groups).  Prints achieved GB/s against the algorithmic bytes of each pass.  Usage: python tools/bn_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatiotemporal_variable_separation_amd import ops  # noqa: E402

SHAPES = [(2048, 64, 32, 32, 16), (2048, 128, 16, 16, 16), (2048, 256, 8, 8, 16), (2048, 512, 4, 4, 16), (256, 64, 32, 32, 2),
          (256, 128, 16, 16, 2), (256, 512, 4, 4, 2)]


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for B, C, H, W, G in SHAPES:
    x = torch.randn(B, C, H, W, device='cuda').bfloat16()
    dy = torch.randn(B, C, H, W, device='cuda').bfloat16()
    gamma, beta = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')
    nbytes = x.numel() * 2
    mean, invstd = ops.bn_stats(x, groups=G)
    t_stats = timed(lambda: ops.bn_stats(x, groups=G))
    t_fwd = timed(lambda: ops.bn_act_fwd(x, mean, invstd, gamma, beta, 'leaky_relu', torch.bfloat16, groups=G))
    t_bwd = timed(lambda: ops.bn_act_bwd(dy, x, mean, invstd, gamma, beta, 'leaky_relu', True, torch.bfloat16, groups=G))
    print('[%4d,%3d,%2d,%2d] g=%2d  %6.1f MB | stats %6.1f us %5.0f GB/s | fwd %6.1f us %5.0f GB/s | bwd (2 kernels) %6.1f us %5.0f GB/s'
          % (B, C, H, W, G, nbytes / 1e6, t_stats, nbytes / t_stats / 1e3, t_fwd, 2 * nbytes / t_fwd / 1e3, t_bwd, 5 * nbytes / t_bwd / 1e3))
