"""Which Python lines launch torch's own kernels (fills, copies, concatenations, elementwise adds) inside one eager training step of a bench
workload: torch.profiler with stacks, device time and launch count per (ATen op, innermost frame inside this package).

    python tools/aten_trace.py <workload> [precision]      ->  table on stdout"""
import os
import sys
from collections import defaultdict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'taxibj'
    precision = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
    from spatiotemporal_variable_separation_amd import functional as VF
    from spatiotemporal_variable_separation_amd.configs import BASELINE_CONFIGS
    from spatiotemporal_variable_separation_amd.data.synthetic import synthetic_batch
    from spatiotemporal_variable_separation_amd.networks.factory import build_sep_net
    from spatiotemporal_variable_separation_amd.optim import Adam
    from spatiotemporal_variable_separation_amd.train import compute_losses
    cfg = dict(BASELINE_CONFIGS[name])
    torch.manual_seed(1234)
    np.random.seed(1234)
    dev = torch.device('cuda')
    net = build_sep_net(cfg).to(dev).train()
    opt = Adam(net.parameters(), lr=4e-4, betas=(0.9, 0.99))
    cond, target = synthetic_batch(cfg['data'], cfg['batch'], cfg['nt_cond'], cfg['nt_pred'], device=dev, seed=1234)
    lam = cfg['lambdas']
    VF.set_precision(precision)
    VF.fold_repeated_gradients(True)
    t_dev = torch.full((1,), cfg['nt_cond'] + 1, dtype=torch.int32, device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        total = compute_losses(cond, target, net, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], cfg.get('skipco', False), lam['ae'], lam['s'], lam['t'],
                               lam['pred'], False, t_random=t_dev)[0]
        total.backward()
        opt.step()
        VF.flush_bn_call_counts()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    # call sites of the usual suspects (the profiler delivers no Python stacks here): wrappers that note the innermost frame in the package
    import traceback
    sites = defaultdict(int)

    def note(kind, args=()):
        for fr in reversed(traceback.extract_stack()[:-2]):
            if 'spatiotemporal_variable_separation_amd' in fr.filename:
                shp = ''
                for a in args:
                    if isinstance(a, torch.Tensor):
                        shp = str(tuple(a.shape)) + str(a.dtype).replace('torch.', ' ')
                        break
                    if isinstance(a, (list, tuple)) and a and isinstance(a[0], torch.Tensor):
                        shp = '[%d x %s]' % (len(a), tuple(a[0].shape))
                        break
                sites[(kind, '%s:%d %s' % (os.path.relpath(fr.filename, ROOT), fr.lineno, fr.name), shp)] += 1
                return
        sites[(kind, '(outside the package: autograd engine / torch)', '')] += 1
    originals = {}

    def wrap(owner, attr, kind):
        fn = getattr(owner, attr)
        originals[(owner, attr)] = fn

        def inner(*a, **k):
            note(kind, a)
            return fn(*a, **k)
        setattr(owner, attr, inner)
    for owner, attr in ((torch, 'cat'), (torch, 'zeros'), (torch, 'zeros_like'), (torch, 'stack'), (torch, 'ones_like'), (torch, 'full'),
                        (torch.Tensor, 'copy_'), (torch.Tensor, 'zero_'), (torch.Tensor, 'fill_'), (torch.Tensor, 'contiguous'), (torch.Tensor, 'float'),
                        (torch.Tensor, 'to'), (torch.Tensor, 'clone'), (torch.Tensor, 'add_'), (torch.Tensor, '__add__'), (torch.Tensor, 'sum')):
        wrap(owner, attr, '%s.%s' % (getattr(owner, '__name__', 'Tensor'), attr))
    step()
    torch.cuda.synchronize()
    for (owner, attr), fn in originals.items():
        setattr(owner, attr, fn)
    print('python-level call sites of one eager step (calls that may launch a torch kernel; no-op views included):')
    for (kind, where, shp), n in sorted(sites.items(), key=lambda kv: -kv[1])[:60]:
        print(f'{n:5d} x  {kind:22s} {where[:90]:90s} {shp}')
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    agg = defaultdict(lambda: [0, 0.0])
    pkg = 'spatiotemporal_variable_separation_amd'
    for ev in prof.key_averages(group_by_stack_n=16):
        us = getattr(ev, 'self_device_time_total', 0) or getattr(ev, 'self_cuda_time_total', 0)
        if not ev.key.startswith('aten::') or us <= 0:
            continue
        where = '?'
        for fr in ev.stack or []:
            if pkg in fr:
                where = fr.replace(ROOT + '/', '').strip()
                break
        agg[(ev.key, where, '')][0] += ev.count
        agg[(ev.key, where, '')][1] += us
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for v in agg.values())
    print(f'{name} {precision}: torch kernels of one eager step: {sum(v[0] for v in agg.values())} launches, {tot:.0f} us of device time')
    for (op, where, shp), (n, us) in rows[:40]:
        print(f'{us:8.1f} us {n:4d} x  {op:28s} {where[:110]}')


if __name__ == '__main__':
    main()
