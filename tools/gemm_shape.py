"""Time one vs_gemm shape: python tools/gemm_shape.py M N K la lb  (bf16; honours VS_GEMM_TILE / VS_GEMM_GLDS*)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatiotemporal_variable_separation_amd import ops  # noqa: E402

M, N, K, la, lb = (int(v) for v in sys.argv[1:6])
a = (torch.rand((M, K) if la == 0 else (K, M), device='cuda') - 0.5).bfloat16()
b = (torch.rand((N, K) if lb == 0 else (K, N), device='cuda') - 0.5).bfloat16()
out = torch.empty((M, N), device='cuda', dtype=torch.bfloat16)
for _ in range(3):
    ops.gemm(a, la, b, lb, M, N, K, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.gemm(a, la, b, lb, M, N, K, out=out)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print('M=%d N=%d K=%d (%d,%d) tile=%s: %.1f us  %.1f TF/s' % (M, N, K, la, lb, os.environ.get('VS_GEMM_TILE', 'plan'), us, 2.0 * M * N * K / us / 1e6))
