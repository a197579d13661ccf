"""Per-call table of one eager training step: every `ops.*` wrapper call with its role, tensor shapes and HIP-event duration, summed over
identical calls.  The kernel statistics (tools/replay_stats.py) say WHICH kernels cost time; this says which LAYERS they were launched for.
usage: python3 tools/layer_trace.py <workload> [precision] [top]        (run on a GPU box; nested wrapper calls are listed indented)"""
import inspect
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def describe(v):
    if torch.is_tensor(v):
        return 'x'.join(str(d) for d in v.shape) + ('h' if v.dtype in (torch.bfloat16, torch.float16) else '')
    if isinstance(v, (list, tuple)) and v and all(torch.is_tensor(t) or isinstance(t, (list, tuple)) for t in v):
        return '[%d: %s]' % (len(v), describe(v[0]))
    if isinstance(v, (int, str, bool)) and not isinstance(v, torch.dtype):
        return str(v)
    return None


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'mnist_b128'
    precision = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    import spatiotemporal_variable_separation_amd as pkg
    pkg.configure_single_gpu_queues()
    from spatiotemporal_variable_separation_amd import functional as VF, ops
    from spatiotemporal_variable_separation_amd.configs import BASELINE_CONFIGS
    from spatiotemporal_variable_separation_amd.data.synthetic import synthetic_batch
    from spatiotemporal_variable_separation_amd.networks.factory import build_sep_net
    from spatiotemporal_variable_separation_amd.optim import Adam
    from spatiotemporal_variable_separation_amd.train import compute_losses, enable_fused_update, enable_update_in_backward
    dev = torch.device('cuda:0')
    cfg = dict(BASELINE_CONFIGS[name])
    torch.manual_seed(1234)
    np.random.seed(1234)
    net = build_sep_net(cfg).to(dev)
    net.train()
    opt = Adam(net.parameters(), lr=4e-4, betas=(0.9, 0.99))
    cond, target = synthetic_batch(cfg['data'], cfg['batch'], cfg['nt_cond'], cfg['nt_pred'], device=dev, seed=1234)
    lam = cfg['lambdas']
    VF.set_precision(precision)
    enable_update_in_backward(opt, net, None)
    if precision == 'bf16':
        enable_fused_update(opt, net, None, None)
    VF.fold_repeated_gradients(True)

    def step():
        opt.zero_grad(set_to_none=True)
        total, _, _, _ = compute_losses(cond, target, net, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], cfg.get('skipco', False),
                                        lam['ae'], lam['s'], lam['t'], lam['pred'], average_tloss=bool(cfg.get('average_tloss')))
        total.backward()
        opt.step()
        VF.flush_bn_call_counts()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    records, depth = [], [0]
    skip = {'profile_reset', 'profile_collect', 'rollout_exchange_error', 'colsum_alloc'}

    def wrap(fname, fn):
        params = list(inspect.signature(fn).parameters)

        def inner(*a, **k):
            parts = []
            for pn, v in list(zip(params, a)) + sorted(k.items()):
                d = describe(v)
                if d is not None and pn not in ('out', 'into', 'eps', 'momentum'):
                    parts.append('%s=%s' % (pn, d))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            depth[0] += 1
            try:
                return fn(*a, **k)
            finally:
                depth[0] -= 1
                e1.record()
                records.append((depth[0], fname, ' '.join(parts), e0, e1))
        return inner

    for fname, fn in list(vars(ops).items()):
        if (inspect.isfunction(fn) and not fname.startswith('_') and fname not in skip and not fname.endswith('_supported')
                and fn.__module__ == ops.__name__):
            setattr(ops, fname, wrap(fname, fn))
    n = 2
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    agg = {}
    for d, fname, desc, e0, e1 in records:
        r = agg.setdefault((d, fname, desc), [0, 0.0])
        r[0] += 1
        r[1] += e0.elapsed_time(e1) * 1e3
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for (d, _, _), v in agg.items() if d == 0) / n
    print('%s %s: %.0f us/step inside ops.* calls (event pairs around Python-issued launches: includes launch gaps)' % (name, precision, tot))
    print('%9s %7s %9s  call' % ('us/step', 'n/step', 'us avg'))
    for (d, fname, desc), (cnt, us) in rows[:top]:
        print('%9.1f %7.1f %9.1f  %s%s(%s)' % (us / n, cnt / n, us / cnt, '  ' * d, fname, desc[:230]))


if __name__ == '__main__':
    main()
