import os, sys, copy, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatiotemporal_variable_separation_amd import functional as VF
from spatiotemporal_variable_separation_amd.configs import BASELINE_CONFIGS
from spatiotemporal_variable_separation_amd.data.synthetic import synthetic_batch
from spatiotemporal_variable_separation_amd.networks.factory import build_sep_net
from spatiotemporal_variable_separation_amd.train import compute_losses
from spatiotemporal_variable_separation_amd.optim import Adam
cfg = dict(BASELINE_CONFIGS['waveeq']); cfg['batch'] = 32
VF.set_precision(sys.argv[1] if len(sys.argv) > 1 else 'bf16')
cond, target = synthetic_batch(cfg['data'], cfg['batch'], cfg['nt_cond'], cfg['nt_pred'], device='cuda', seed=1)
lam = cfg['lambdas']
def run(kind):
    torch.manual_seed(5); np.random.seed(5)
    net = build_sep_net(cfg).cuda().train()
    opt = Adam(net.parameters(), lr=4e-4, betas=(0.9, 0.99)) if kind == 'hip' else torch.optim.Adam(net.parameters(), lr=4e-4, betas=(0.9, 0.99), fused=(kind == 'fused'))
    out = []
    for i in range(12):
        opt.zero_grad(set_to_none=True)
        total = compute_losses(cond, target, net, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], False, lam['ae'], lam['s'], lam['t'], lam['pred'])[0]
        total.backward()
        opt.step()
        out.append(round(total.item(), 5))
    return out
for k in ('torch', 'fused', 'hip'):
    print(k, run(k))
