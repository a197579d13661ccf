"""Per-layer timing of the convolution kernels: the LDS-staged tap kernels against the column-matrix / implicit path.
Usage: python tools/conv_bench.py      (prints us per launch and TFLOP/s for the layers of the MNIST / TaxiBJ / SST recipes)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatiotemporal_variable_separation_amd import ops  # noqa: E402

dt = torch.bfloat16
LAYERS = [  # (name, kind, B, Cin, H, Cout)
    ('mnist dec L1 512->256 @4', 'convT4', 2048, 512, 4, 256),
    ('mnist dec L2 256->128 @8', 'convT4', 2048, 256, 8, 128),
    ('mnist dec L3 128->64 @16', 'convT4', 2048, 128, 16, 64),
    ('mnist enc dgrad 256->128 @8', 'convT4', 256, 256, 8, 128),
    ('taxibj 128->128 @16 (dec, 900)', 'conv3', 900, 128, 16, 128),
    ('taxibj 256->256 @8 (dec, 900)', 'conv3', 900, 256, 8, 256),
    ('taxibj 128->128 @16 (enc, 200)', 'conv3', 200, 128, 16, 128),
    ('sst resnet 64->512 @16', 'conv3', 8, 64, 16, 512),
    ('sst resnet 512->512 @16', 'conv3', 8, 512, 16, 512),
    ('sst resnet 512->64 @16', 'conv3', 8, 512, 16, 64),
    ('sst dec 256->256 @16 (328)', 'conv3', 328, 256, 16, 256),
]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, kind, B, Cin, H, Cout in LAYERS:
    x = (torch.rand((B, Cin, H, H), device='cuda') - 0.5).to(dt)
    if kind == 'convT4':
        w = (torch.rand((Cin, Cout, 4, 4), device='cuda') - 0.5) * 0.1
        wp = ops.conv_pack_weight(w, dt, 2, 1)
        wt = ops.convt_tap_pack_weight(w, dt)
        fl = 2.0 * B * Cin * H * H * Cout * 16
        old = timeit(lambda: ops.conv_fwd(x, w.to(dt), None, 2, 1, True, dt, w_packed=wp))
        new = timeit(lambda: ops.convt_tap_fwd(x, wt, None, Cout, groups=1, want_sums=True))
    else:
        w = (torch.rand((Cout, Cin, 3, 3), device='cuda') - 0.5) * 0.1
        wt = ops.conv_k3_tap_pack_weight(w, dt, False)
        wd = w.to(dt)
        fl = 2.0 * B * Cin * H * H * Cout * 9
        old = timeit(lambda: ops.conv_fwd(x, wd, None, 1, 1, False, dt))
        new = timeit(lambda: ops.conv_k3_tap_fwd(x, wt, None, Cout, dt, groups=1, want_sums=True))
    print(f'{name:34s} {fl / 1e9:7.1f} GF   column-matrix path {old:8.1f} us {fl / old / 1e6:7.1f} TF/s   tap kernel {new:8.1f} us {fl / new / 1e6:7.1f} TF/s')
