"""Throughput of the device-side batch assembly at the WaveEq recipe's sizes (README.md:90 of the reference: 64x64 frames, 5 + 20
frames per item, batch 128): one vs_gather_windows launch per batch from a [300, 300, 4096] fp32 set resident in HBM."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatiotemporal_variable_separation_amd import ops  # noqa: E402

n_seq, nt, frame, B, seq_len = 300, 300, 4096, 128, 25
data = torch.rand(n_seq, nt, frame, device='cuda')
per = nt + 1 - seq_len
g = torch.Generator().manual_seed(0)
for dt in (torch.float32, torch.bfloat16):
    idx = [torch.randint(0, n_seq * per, (B,), generator=g, dtype=torch.int32).cuda() for _ in range(20)]
    for i in idx[:3]:
        ops.gather_windows(data, i, per, seq_len, out_dtype=dt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in idx:
        ops.gather_windows(data, i, per, seq_len, out_dtype=dt)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / len(idx) * 1e3
    nbytes = B * seq_len * frame * (4 + (4 if dt == torch.float32 else 2))
    print('%s: %.1f us per batch of %d x %d frames  (%.0f GB/s read+write, %.1f M frames/s)' % (dt, us, B, seq_len, nbytes / us / 1e3, B * seq_len / us))
