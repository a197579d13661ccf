"""Row-band convolution kernels alone, at the layer shapes of the BASELINE workloads (TaxiBJ VGG32 B=100, SST B=8 nt_pred 40, Moving-MNIST B=128).

    python3 tools/band_bench.py [fwd|wgrad|k4|all] [--check] [--cold]

Per layer: us per launch (median of 5 x 20 launches, interleaved over the variants named in VS_BAND_BENCH_VARIANTS) and TFLOP/s.  `--check`
compares every variant's result with an fp64 convolution of the same 16-bit operands on a reduced batch.  The variants are environment
settings the library reads per call (e.g. `VS_BAND_V2=0,1`): A/B pairs run in ONE process, interleaved (cdna_hip_programming.md rule 24).
`--cold`: every launch on its own operands (rotating through > 256 MiB), the state a layer finds inside the step."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatiotemporal_variable_separation_amd import ops  # noqa: E402

dt = torch.bfloat16
# (name, maps, Cin, H(=W), Cout)
FWD = [
    ('taxibj 64->64 @32 enc', 200, 64, 32, 64), ('taxibj 64->64 @32 dec', 900, 64, 32, 64),
    ('taxibj 64->128 @16 enc', 200, 64, 16, 128), ('taxibj 128->128 @16 enc', 200, 128, 16, 128),
    ('taxibj 128->128 @16 dec', 900, 128, 16, 128), ('taxibj 128->64 @16 dec', 900, 128, 16, 64),
    ('taxibj 128->256 @8 enc', 200, 128, 8, 256), ('taxibj 256->256 @8 enc', 200, 256, 8, 256),
    ('taxibj 256->256 @8 dec', 900, 256, 8, 256), ('taxibj 256->128 @8 dec', 900, 256, 8, 128),
    ('taxibj 256->512 @4 enc', 200, 256, 4, 512), ('taxibj 512->512 @4 enc', 200, 512, 4, 512),
    ('taxibj 512->512 @4 dec', 900, 512, 4, 512), ('taxibj 512->256 @4 dec', 900, 512, 4, 256),
    ('sst 64->64 @64 enc', 16, 64, 64, 64), ('sst 64->64 @64 dec', 328, 64, 64, 64), ('sst 128->64 @64 dec', 328, 128, 64, 64),
    ('sst 128->128 @32 enc', 16, 128, 32, 128), ('sst 192->128 @32 dec', 328, 192, 32, 128), ('sst 128->64 @32 dec', 328, 128, 32, 64),
    ('sst 256->256 @16 enc', 16, 256, 16, 256), ('sst 260->256 @16 dec', 328, 260, 16, 256), ('sst 256->256 @16 dec', 328, 256, 16, 256),
    ('sst 384->128 @16 dec', 328, 384, 16, 128),
]
# k4 s2 p1 family on parity planes: (name, maps, K (channels of the large map), h (plane height = small map height), M)
K4 = [
    ('mnist enc 64->128 planes @16', 256, 64, 16, 128), ('mnist enc 128->256 planes @8', 256, 128, 8, 256),
    ('mnist enc 256->512 planes @4', 256, 256, 4, 512),
    ('mnist dec dgrad 64<-128 planes @16', 1920, 64, 16, 128), ('mnist dec dgrad 128<-256 planes @8', 1920, 128, 8, 256),
    ('mnist dec dgrad 256<-512 planes @4', 1920, 256, 4, 512),
]


def variants():
    v = os.environ.get('VS_BAND_BENCH_VARIANTS', '')
    out = [('default', {})]
    for item in v.split(';'):
        if not item.strip():
            continue
        name, vals = item.split('=')
        for val in vals.split(','):
            out.append(('%s=%s' % (name, val), {name: val}))
    return out if len(out) == 1 else out[1:]


class Env:
    def __init__(self, kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def time_variants(fns, rounds=5, iters=20):
    """fns: {name: callable}; interleaved rounds, median us per launch."""
    res = {k: [] for k in fns}
    for k, fn in fns.items():
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, fn in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) / iters * 1e3)
    return {k: (statistics.median(v), min(v)) for k, v in res.items()}


def rel_err(a, ref):
    return float((a.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def bench_fwd(check, cold):
    print('== conv3_band (forward / input gradient): us per launch (median, min) and TFLOP/s at the median')
    for name, B, Cin, H, Cout in FWD:
        nset = max(1, min(8, int(3e8 // (B * (Cin + Cout) * H * H * 2)))) if cold else 1
        xs = [(torch.rand((B, Cin, H, H), device='cuda') - 0.5).to(dt) for _ in range(nset)]
        w = (torch.rand((Cout, Cin, 3, 3), device='cuda') - 0.5) * 0.1
        bias = torch.rand((Cout,), device='cuda') - 0.5
        if not ops.conv3_band_supported(xs[0], Cout):
            print(f'{name:38s} not served by the row-band kernel')
            continue
        fl = 2.0 * B * Cin * H * H * Cout * 9
        fns = {}
        state = {'i': 0}
        for vname, kv in variants():
            with Env(kv):
                wp = ops.conv3_img16_pack_weight(w, dt, False)

            def fn(kv=kv, wp=wp):
                with Env(kv):
                    state['i'] = (state['i'] + 1) % nset
                    return ops.conv3_band(xs[state['i']], wp, bias, Cout, dt)
            fns[vname] = fn
        if check:
            nb = min(B, 16 if H <= 8 else 3)
            xr = xs[0][:nb].contiguous()
            ref = torch.nn.functional.conv2d(xr.double(), w.to(dt).double(), bias.double(), padding=1)
            for vname, kv in variants():
                with Env(kv):
                    y = ops.conv3_band(xr, ops.conv3_img16_pack_weight(w, dt, False), bias, Cout, torch.float32)
                e = rel_err(y, ref)
                assert e < 2e-5, (name, vname, e)
        r = time_variants(fns)
        print(f'{name:38s} {fl / 1e9:7.1f} GF  ' + '  '.join(f'{k}: {m:7.1f} ({lo:7.1f}) us {fl / m / 1e6:6.0f} TF/s' for k, (m, lo) in r.items()))


def bench_wgrad(check, cold):
    print('== conv3_wgrad_band (incl. the slab finishing passes)')
    for name, B, Cin, H, Cout in FWD:
        x = (torch.rand((B, Cin, H, H), device='cuda') - 0.5).to(dt)
        dz = (torch.rand((B, Cout, H, H), device='cuda') - 0.5).to(dt)
        if not ops.conv3_wgrad_band_supported(x, Cout):
            print(f'{name:38s} not served by the row-band weight gradient')
            continue
        fl = 2.0 * B * Cin * H * H * Cout * 9
        fns = {}
        for vname, kv in variants():
            def fn(kv=kv):
                with Env(kv):
                    return ops.conv_wgrad(dz, x, (Cout, Cin, 3, 3), 1, 1, False)
            fns[vname] = fn
        if check:
            nb = min(B, 32 if H <= 8 else 4)
            nb = nb // 16 * 16 if H == 4 else (nb // 4 * 4 if H == 8 else nb)
            xr, dr = x[:nb].contiguous(), dz[:nb].contiguous()
            wz = torch.zeros((Cout, Cin, 3, 3), dtype=torch.float64, device='cuda', requires_grad=True)
            torch.nn.functional.conv2d(xr.double(), wz, None, padding=1).backward(dr.double())
            for vname, kv in variants():
                with Env(kv):
                    dw = ops.conv_wgrad(dr, xr, (Cout, Cin, 3, 3), 1, 1, False)
                e = rel_err(dw, wz.grad)
                assert e < 2e-5, (name, vname, e)
        r = time_variants(fns, iters=10)
        print(f'{name:38s} {fl / 1e9:7.1f} GF  ' + '  '.join(f'{k}: {m:7.1f} ({lo:7.1f}) us {fl / m / 1e6:6.0f} TF/s' for k, (m, lo) in r.items()))


def bench_k4(check, cold):
    print('== k4 s2 p1 on parity planes: gather (Conv2d forward / ConvTranspose2d input gradient) and weight gradient')
    for name, B, K, h, M in K4:
        planes = (torch.rand((B, 4 * K, h, h), device='cuda') - 0.5).to(dt)
        small = (torch.rand((B, M, h, h), device='cuda') - 0.5).to(dt)
        w = (torch.rand((M, K, 4, 4), device='cuda') - 0.5) * 0.1
        bias = torch.rand((M,), device='cuda') - 0.5
        fl = 2.0 * B * h * h * M * K * 16
        fns, fw = {}, {}
        for vname, kv in variants():
            with Env(kv):
                wp = ops.conv_k4s2_pack_weight(w, dt)

            def fn(kv=kv, wp=wp):
                with Env(kv):
                    return ops.conv_k4s2_gather(planes, wp, bias, M, dt)
            fns[vname] = fn
            if h >= 8:
                def fg(kv=kv):
                    with Env(kv):
                        return ops.conv_k4s2_wgrad(small, planes, (M, K, 4, 4))
                fw[vname] = fg
        if check:
            nb = 16
            pr = planes[:nb].contiguous()
            # the large map from its parity planes: big[c][2y + py][2x + px] = planes[(2 py + px) K + c][y][x]
            big = pr.view(nb, 2, 2, K, h, h).permute(0, 3, 4, 1, 5, 2).reshape(nb, K, 2 * h, 2 * h)
            ref = torch.nn.functional.conv2d(big.double(), w.to(dt).double(), bias.double(), stride=2, padding=1)
            for vname, kv in variants():
                with Env(kv):
                    y = ops.conv_k4s2_gather(pr, ops.conv_k4s2_pack_weight(w, dt), bias, M, torch.float32)
                e = rel_err(y, ref)
                assert e < 2e-5, (name, vname, e)
            if h >= 8:
                sr = small[:nb].contiguous()
                wz = torch.zeros((M, K, 4, 4), dtype=torch.float64, device='cuda', requires_grad=True)
                torch.nn.functional.conv2d(big.double(), wz, None, stride=2, padding=1).backward(sr.double())
                for vname, kv in variants():
                    with Env(kv):
                        dw = ops.conv_k4s2_wgrad(sr, pr, (M, K, 4, 4))
                    e = rel_err(dw, wz.grad)
                    assert e < 2e-5, (name, 'wgrad', vname, e)
        r = time_variants(fns)
        print(f'{name:38s} {fl / 1e9:7.1f} GF  gather ' + '  '.join(f'{k}: {m:7.1f} ({lo:7.1f}) us {fl / m / 1e6:6.0f} TF/s' for k, (m, lo) in r.items()))
        if fw:
            r = time_variants(fw, iters=10)
            print(f'{"":38s} {fl / 1e9:7.1f} GF  wgrad  ' + '  '.join(f'{k}: {m:7.1f} ({lo:7.1f}) us {fl / m / 1e6:6.0f} TF/s' for k, (m, lo) in r.items()))


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith('-') else 'all'
    check, cold = '--check' in sys.argv, '--cold' in sys.argv
    only = [a.split('=', 1)[1] for a in sys.argv if a.startswith('--only=')]
    if only:
        keys = only[0].split(',')
        FWD[:] = [r for r in FWD if any(k in r[0] for k in keys)]
        K4[:] = [r for r in K4 if any(k in r[0] for k in keys)]
    print('variants:', [v[0] for v in variants()], 'cold' if cold else 'hot')
    if what in ('fwd', 'all'):
        bench_fwd(check, cold)
    if what in ('k4', 'all'):
        bench_k4(check, cold)
    if what in ('wgrad', 'all'):
        bench_wgrad(check, cold)
