"""Micro-benchmark of vs_gemm on the WaveEq (config 2) shapes.  Usage: python tools/gemm_bench.py [bf16|f32] [cold]

`cold`: every launch works on its own copy of the operands and output, ~600 MB in rotation (more than the 256 MB Infinity Cache),
as inside a training step where operands were last touched a millisecond and a gigabyte of traffic ago.  Re-running one launch on
the same buffers (the default) keeps them in the Infinity Cache and flatters kernels whose loop is bound by request latency."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatiotemporal_variable_separation_amd import ops  # noqa: E402

dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == 'bf16') else torch.float32
COLD = 'cold' in sys.argv[2:]
SHAPES = [  # (name, M, N, K, la, lb) -- every GEMM of one WaveEq training step (B=128: encoder rows 256, decoder rows 3328)
    ('dec fwd 1200->4096', 3328, 4096, 1200, 0, 0),
    ('dec fwd 1200->1200 x2', 3328, 1200, 1200, 0, 0),
    ('dec fwd 32->1200', 3328, 1200, 32, 0, 0),
    ('enc fwd 20480->1200 x2', 256, 1200, 20480, 0, 0),
    ('enc fwd 1200->1200 x2', 256, 1200, 1200, 0, 0),
    ('enc fwd 1200->32 x2', 256, 32, 1200, 0, 0),
    ('dec dgrad 4096->1200', 3328, 1200, 4096, 0, 1),
    ('dec dgrad 1200->1200 x2', 3328, 1200, 1200, 0, 1),
    ('dec dgrad 1200->32', 3328, 32, 1200, 0, 1),
    ('enc dgrad 32->1200 x2', 256, 1200, 32, 0, 1),
    ('enc dgrad 1200->1200 x2', 256, 1200, 1200, 0, 1),
    ('dec wgrad 4096x1200', 4096, 1200, 3328, 1, 1),
    ('dec wgrad 1200x1200 x2', 1200, 1200, 3328, 1, 1),
    ('dec wgrad 1200x32', 1200, 32, 3328, 1, 1),
    ('enc wgrad 1200x20480 x2', 1200, 20480, 256, 1, 1),
    ('enc wgrad 1200x1200 x2', 1200, 1200, 256, 1, 1),
    ('enc wgrad 32x1200 x2', 32, 1200, 256, 1, 1),
    ('square 4096', 4096, 4096, 4096, 0, 0),
    ('square 4096 SS', 4096, 4096, 4096, 1, 1),
]
for name, M, N, K, la, lb in SHAPES:
    per_set = (M * K + N * K) * (2 if dt == torch.bfloat16 else 4) + M * N * 4
    nset = max(1, min(64, int(600e6 // per_set))) if COLD else 1
    As = [(torch.rand((M, K) if la == 0 else (K, M), device='cuda') - 0.5).to(dt) for _ in range(nset)]
    Bs = [(torch.rand((N, K) if lb == 0 else (K, N), device='cuda') - 0.5).to(dt) for _ in range(nset)]
    outs = [torch.empty((M, N), device='cuda', dtype=torch.float32) for _ in range(nset)]
    for i in range(max(3, nset)):
        ops.gemm(As[i % nset], la, Bs[i % nset], lb, M, N, K, out=outs[i % nset])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = max(20, nset)
    e0.record()
    for i in range(iters):
        ops.gemm(As[i % nset], la, Bs[i % nset], lb, M, N, K, out=outs[i % nset])
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    # torch (hipBLASLt) for orientation only
    tas = [a if la == 0 else a.t() for a in As]
    tbs = [b if lb == 0 else b.t() for b in Bs]
    for i in range(max(3, nset)):
        torch.matmul(tas[i % nset], tbs[i % nset].t())
    torch.cuda.synchronize()
    e0.record()
    for i in range(iters):
        torch.matmul(tas[i % nset], tbs[i % nset].t())
    e1.record()
    torch.cuda.synchronize()
    ms_t = e0.elapsed_time(e1) / iters
    fl = 2.0 * M * N * K
    print(f'{name:24s} {str(dt)[6:]:9s} M={M:5d} N={N:5d} K={K:5d}  vs_gemm {ms * 1e3:8.1f} us {fl / ms / 1e9:8.1f} TF/s   '
          f'[torch.matmul {ms_t * 1e3:8.1f} us {fl / ms_t / 1e9:8.1f} TF/s]' + ('  cold x%d' % nset if COLD else ''))
    del As, Bs, outs, tas, tbs
