"""Loss sequence of train.GraphedStep vs the eager loop in bf16 with the HIP Adam (same NumPy stream): python tools/graph_vs_eager.py <config>"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpu_ref  # noqa: E402
from oracle.detdata import det_fill  # noqa: E402
from oracle.golden_configs import CONFIGS, make_batch  # noqa: E402
from spatiotemporal_variable_separation_amd import functional as VF  # noqa: E402
from spatiotemporal_variable_separation_amd.networks.factory import build_sep_net  # noqa: E402
from spatiotemporal_variable_separation_amd.optim import Adam  # noqa: E402
from spatiotemporal_variable_separation_amd.train import GraphedStep, compute_losses  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'dcgan_tiny'
cfg = CONFIGS[name]
lam, skipco = cfg['lambdas'], bool(cfg.get('skipco', False))
cond, target = make_batch(cfg)
cond, target = cond.cuda(), target.cuda()
o_net = det_fill(cpu_ref.build_sep_net(cfg), salt=cfg['salt'])


def fresh():
    net = build_sep_net(cfg)
    net.load_state_dict(o_net.state_dict())
    return net.cuda().train()


VF.set_precision('bf16')
VF.fold_repeated_gradients(os.environ.get('FOLD', '1') == '1')
hi = cfg['nt_cond'] + cfg['nt_pred'] + (0 if cfg['offset'] == 0 else 1)
net_e = fresh()
opt_e = Adam(net_e.parameters(), lr=1e-3)
np.random.seed(7)
le = []
for it in range(9):
    if it == 3:
        np.random.randint(cfg['nt_cond'], hi)
    opt_e.zero_grad()
    total = compute_losses(cond, target, net_e, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], skipco, lam['ae'], lam['s'], lam['t'], lam['pred'],
                           average_tloss=bool(cfg.get('average_tloss')))[0]
    total.backward()
    opt_e.step()
    VF.flush_bn_call_counts()
    le.append(round(total.item(), 5))
net_g = fresh()
opt_g = Adam(net_g.parameters(), lr=1e-3)
np.random.seed(7)
g = GraphedStep(net_g, opt_g, cond, target, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], (lam['ae'], lam['s'], lam['t'], lam['pred']),
                average_tloss=bool(cfg.get('average_tloss')), warmup=3)
lg = [round(g.step().item(), 5) for _ in range(6)]
print(name, 'eager ', le[3:])
print(name, 'graph ', lg)
