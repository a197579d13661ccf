"""How far two bf16 runs of the SAME rounding scheme drift apart through a whole conv-family step when only the fp32
accumulation order differs (implicit gather GEMM vs column-matrix GEMM): the noise floor of any bf16 step comparison."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from oracle import cpu_ref
from oracle.detdata import det_fill
from oracle.golden_configs import CONFIGS
from golden_util import load_golden, rel_err
from step_util import hip_step, emulated_product_step
for name in sys.argv[1:]:
    cfg = CONFIGS[name]
    t = int(load_golden(name)['t_random'])
    o = det_fill(cpu_ref.build_sep_net(cfg), salt=cfg['salt'])
    os.environ['VS_CONV_COLS'] = '0'
    a = hip_step(cfg, t, o, 'bf16')
    os.environ.pop('VS_CONV_COLS')
    os.environ['VS_CONV_COLS_FORCE'] = '1'
    b = hip_step(cfg, t, o, 'bf16')
    os.environ.pop('VS_CONV_COLS_FORCE')
    e = emulated_product_step(cfg, t)
    f = lambda x, y: rel_err(x[3].detach().cpu().float(), y[3].detach().cpu().float())
    c = lambda x, y: rel_err(x[4].detach().cpu().float(), y[4].detach().cpu().float())
    print(name, 'forecasts: implicit vs cols %.2e | implicit vs emu %.2e | cols vs emu %.2e ; t_codes: %.2e %.2e %.2e' % (f(a, b), f(a, e), f(b, e), c(a, b), c(a, e), c(b, e)))
