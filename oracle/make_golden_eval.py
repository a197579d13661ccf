"""Generate tests/golden/eval_<config>.npz: the REFERENCE's inference path (imported read-only from /root/reference).

TEST INFRASTRUCTURE ONLY.  Runs in the build container (the reference never travels to the GPU box):

    python -m oracle.make_golden_eval

What the reference's evaluation scripts do with a trained model (var_sep/test/wave/test.py:41-48, test/mnist/test.py:120-131,
test/mnist/test_disentanglement.py, test/utils.py:8-16): `.eval()` (BatchNorm on its running statistics), `torch.set_grad_enabled(
False)`, `sep_net.get_forecast(cond, horizon)` over a horizon longer than the training one, and the content swap
`get_forecast(cond, horizon, init_s_code=Es(other))`.  For every reduced-width config of `oracle.golden_configs.CONFIGS` the
reference networks are filled with the RNG-free weights AND BatchNorm statistics of `oracle.detdata.det_fill`, run on the seeded
batch of `make_batch`, and the forecasts / temporal codes / spatial code of both calls are stored (small tensors in full, large
ones as sum / L2 / 16-sample checksums).  The oracle (`oracle.cpu_ref`) must reproduce them before the file is written.
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import cpu_ref                                             # noqa: E402
from oracle.golden_configs import CONFIGS, fill_net, make_batch       # noqa: E402
from oracle.make_golden import _reference_modules, build_reference, pack    # noqa: E402

HORIZON = 12
SWAP_HORIZON = 5
EVAL_CONFIGS = ['mlp_mul', 'mlp_concat_partial', 'mlp_no_s', 'dcgan_tiny', 'dcgan_skip_mul', 'vgg32_tiny', 'sst_skip', 'chairs_resnet']


def _first(s):
    return s[0] if isinstance(s, (tuple, list)) else s


def run(name, cfg, mods):
    rf, rm, ru, rt = mods
    cond, _ = make_batch(cfg)
    other = cond.flip(0)
    ref = fill_net(build_reference(cfg, rf, rm, ru), cfg).eval()
    orc = fill_net(cpu_ref.build_sep_net(dict(cfg)), cfg).eval()
    with torch.no_grad():
        r_fore, r_codes, r_s, _ = ref.get_forecast(cond, HORIZON)
        skip = bool(cfg.get('skipco', False))                          # with skip connections the spatial code is (code, skips)
        r_swap = ref.get_forecast(cond, SWAP_HORIZON, init_s_code=ref.Es(other, return_skip=skip))[0]
        o_fore, o_codes, o_s, _ = orc.get_forecast(cond, HORIZON)
        o_swap = orc.get_forecast(cond, SWAP_HORIZON, init_s_code=orc.Es(other, return_skip=skip))[0]
    worst = max(((a - b).abs().max() / (b.abs().max() + 1e-30)).item()
                for a, b in ((o_fore, r_fore), (o_codes, r_codes), (_first(o_s), _first(r_s)), (o_swap, r_swap)))
    assert worst < 1e-6, (name, worst)
    # eval mode leaves the BatchNorm buffers alone
    filled = fill_net(build_reference(cfg, rf, rm, ru), cfg).state_dict()
    for k, v in ref.state_dict().items():
        assert torch.equal(v, filled[k]), k
    out = {'horizon': np.int64(HORIZON), 'swap_horizon': np.int64(SWAP_HORIZON)}
    pack(out, 'forecasts', r_fore)
    pack(out, 't_codes', r_codes)
    pack(out, 's_code', _first(r_s))
    pack(out, 'swap_forecasts', r_swap)
    path = os.path.join(ROOT, 'tests', 'golden', 'eval_' + name + '.npz')
    np.savez_compressed(path, **out)
    print(f'eval_{name:22s} forecasts {tuple(r_fore.shape)}  oracle_vs_ref relmax={worst:.1e}  -> {os.path.getsize(path) / 1024:.0f} KiB')


# long-term prediction as in the paper's Moving-MNIST evaluation (README.md:116: `--nt_pred 95`; test/mnist/test.py:101,120:
# get_forecast(x_cond, nt_cond + 95)).  The integrator's weights are scaled by 0.3 (golden_configs.fill_net, `res_scale`): with det_fill's
# gain the latent code would double at every block and overflow long before frame 95.
LONG_PRED = 95
LONG_CONFIGS = ['mlp_mul', 'dcgan_tiny', 'dcgan_skip_mul', 'vgg32_tiny', 'sst_skip']


def long_cfg(name):
    return dict(CONFIGS[name], res_scale=0.3)


def run_long(name, mods):
    rf, rm, ru, rt = mods
    cfg = long_cfg(name)
    cond, _ = make_batch(cfg)
    horizon = cfg['nt_cond'] + LONG_PRED
    ref = fill_net(build_reference(cfg, rf, rm, ru), cfg).eval()
    orc = fill_net(cpu_ref.build_sep_net(dict(cfg)), cfg).eval()
    with torch.no_grad():
        r_fore, r_codes, r_s, _ = ref.get_forecast(cond, horizon)
        o_fore, o_codes, o_s, _ = orc.get_forecast(cond, horizon)
    assert torch.isfinite(r_fore).all() and torch.isfinite(r_codes).all(), name
    worst = max(((a - b).abs().max() / (b.abs().max() + 1e-30)).item() for a, b in ((o_fore, r_fore), (o_codes, r_codes)))
    assert worst < 1e-6, (name, worst)
    out = {'horizon': np.int64(horizon), 'nt_pred': np.int64(LONG_PRED)}
    pack(out, 'forecasts', r_fore)
    pack(out, 't_codes', r_codes)
    pack(out, 'last_frame', r_fore[:, -1])                             # the frame 95 steps after the conditioning window, in full where small
    pack(out, 'last_code', r_codes[:, -1])
    path = os.path.join(ROOT, 'tests', 'golden', 'eval_long_' + name + '.npz')
    np.savez_compressed(path, **out)
    print(f'eval_long_{name:17s} forecasts {tuple(r_fore.shape)}  |code| first / last {r_codes[:, 0].norm():.3g} / {r_codes[:, -1].norm():.3g}  '
          f'oracle_vs_ref relmax={worst:.1e}  -> {os.path.getsize(path) / 1024:.0f} KiB')


def main():
    torch.set_num_threads(8)
    mods = _reference_modules()
    only = sys.argv[1:]
    for name in EVAL_CONFIGS:
        if name in CONFIGS and (not only or name in only):
            run(name, CONFIGS[name], mods)
    for name in LONG_CONFIGS:
        if not only or name in only or 'long' in only:
            run_long(name, mods)


if __name__ == '__main__':
    main()
