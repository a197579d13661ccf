"""Every GEMM launch of one eager training step with its shape, layouts and HIP-event time (measurement aid).

    python tools/gemm_shapes.py [workload] [precision]

The step runs three times; the table is the third.  One line per distinct (layouts, M, N, K, out dtype, accumulate) in launch
order with the number of launches, the mean time, TFLOP/s and the GB/s the algorithmic operand + result bytes amount to."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'waveeq'
    precision = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
    from spatiotemporal_variable_separation_amd import functional as VF, ops
    from spatiotemporal_variable_separation_amd.configs import BASELINE_CONFIGS
    from spatiotemporal_variable_separation_amd.data.synthetic import synthetic_batch
    from spatiotemporal_variable_separation_amd.networks.factory import build_sep_net
    from spatiotemporal_variable_separation_amd.optim import Adam
    from spatiotemporal_variable_separation_amd.train import compute_losses, enable_update_in_backward
    dev = torch.device('cuda', 0)
    cfg = dict(BASELINE_CONFIGS[name])
    torch.manual_seed(1234)
    np.random.seed(1234)
    net = build_sep_net(cfg).to(dev)
    net.train()
    opt = Adam(net.parameters(), lr=4e-4, betas=(0.9, 0.99))
    enable_update_in_backward(opt, net, None)
    cond, target = synthetic_batch(cfg['data'], cfg['batch'], cfg['nt_cond'], cfg['nt_pred'], device=dev, seed=1234)
    lam = cfg['lambdas']
    VF.set_precision(precision)
    VF.fold_repeated_gradients(True)
    log = []
    real = ops.gemm

    def logged(a, layout_a, b, layout_b, M, N, K, out=None, out_dtype=torch.float32, alpha=1.0, bias=None, act='none', mask=None,
               mask_act='none', accumulate=False, lda=None, ldb=None):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        r = real(a, layout_a, b, layout_b, M, N, K, out=out, out_dtype=out_dtype, alpha=alpha, bias=bias, act=act, mask=mask,
                 mask_act=mask_act, accumulate=accumulate, lda=lda, ldb=ldb)
        e1.record()
        key = ('RS'[layout_a] + 'RS'[layout_b], M, N, K, str(r.dtype).replace('torch.', ''), bool(accumulate), act, bias is not None,
               mask is not None, str(a.dtype).replace('torch.', ''))
        log.append((key, e0, e1, a.element_size(), r.element_size()))
        return r

    ops.gemm = logged
    for it in range(3):
        log.clear()
        opt.zero_grad(set_to_none=True)
        total = compute_losses(cond, target, net, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], cfg.get('skipco', False), lam['ae'],
                               lam['s'], lam['t'], lam['pred'], average_tloss=bool(cfg.get('average_tloss')))[0]
        total.backward()
        opt.step()
        torch.cuda.synchronize()
    rows = {}
    order = []
    for key, e0, e1, ea, eo in log:
        if key not in rows:
            rows[key] = [0, 0.0, ea, eo]
            order.append(key)
        rows[key][0] += 1
        rows[key][1] += e0.elapsed_time(e1) * 1e3
    tot = 0.0
    print('| layouts | M | N | K | in | out | acc | act | bias | mask | launches | us each | TFLOP/s | GB/s |')
    print('|---|---|---|---|---|---|---|---|---|---|---|---|---|---|')
    for key in order:
        n, us, ea, eo = rows[key]
        lay, M, N, K, odt, acc, act, bias, mask, idt = key
        each = us / n
        tot += us
        fl = 2.0 * M * N * K
        by = (M * K + N * K) * ea + M * N * eo * (2 if acc else 1)
        print(f'| {lay} | {M} | {N} | {K} | {idt} | {odt} | {int(acc)} | {act} | {int(bias)} | {int(mask)} | {n} | {each:.1f} | '
              f'{fl / each / 1e6:.0f} | {by / each / 1e3:.0f} |')
    print(f'\nGEMM launches {len(log)}, summed event time {tot:.0f} us')


if __name__ == '__main__':
    main()
