import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from oracle import cpu_ref
from oracle.detdata import det_fill
from oracle.golden_configs import CONFIGS, make_batch
from oracle.bf16_emu import emulate_product_bf16
from golden_util import rel_err
from spatiotemporal_variable_separation_amd import functional as VF
from spatiotemporal_variable_separation_amd.networks.factory import build_sep_net
from spatiotemporal_variable_separation_amd.networks.conv import run_layers
name = sys.argv[1]
cfg = CONFIGS[name]
cond, target = make_batch(cfg)
o = det_fill(cpu_ref.build_sep_net(cfg), salt=cfg['salt'])
x = cond[:, :cfg['nt_cond']].reshape(cond.shape[0], -1, *cond.shape[-2:])
netg = build_sep_net(cfg); netg.load_state_dict(o.state_dict()); netg = netg.cuda().train()
nete = build_sep_net(cfg); nete.load_state_dict(o.state_dict()); nete.train()
enc_g, enc_e = netg.Et, nete.Et
print(type(enc_g).__name__, [n for n, _ in enc_g.named_children()])
hg, he = x.cuda(), x.clone()
import torch.nn as nn
def units(m):
    kids = list(m.children())
    if isinstance(m, nn.Sequential) and kids and isinstance(kids[0], (nn.Sequential, nn.ModuleList)):
        return [u for k in kids for u in units(k)]
    return [m]
def stages(enc):
    if hasattr(enc, 'conv'):
        return list(enc.conv) + [enc.last_op]
    return [getattr(enc, n) for n in ('conv1', 'conv2', 'conv3', 'conv4')]
ug = [u for st in stages(enc_g) for u in units(st)]
ue = [u for st in stages(enc_e) for u in units(st)]
h = x.to(torch.bfloat16)
for i, (lg, le) in enumerate(zip(ug, ue)):
    last = i == len(ug) - 1
    with VF.precision('bf16'):
        yg = run_layers(lg, h.cuda(), final_fp32=last)
    with emulate_product_bf16():
        ye = run_layers(le, h.clone(), final_fp32=last)
    a_, b_ = yg.detach().cpu().float(), ye.detach().float()
    d = (a_ - b_).abs()
    print(i, [type(c).__name__ for c in lg.children()] or type(lg).__name__, tuple(yg.shape), 'differ', int((d > 0).sum()), 'of', d.numel(),
          'max abs %.3e' % d.max().item(), 'max rel-to-ulp %.2f' % (d / (b_.abs().clamp_min(1e-30) * 2.0 ** -7)).max().item())
    h = yg.detach().cpu()
