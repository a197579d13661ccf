"""CPU: the oracle (oracle/cpu_ref.py) reproduces the vectors the REFERENCE produced (tests/golden/*.npz)."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref
from oracle.detdata import det_fill
from oracle.golden_configs import CONFIGS, FULL_CONFIGS, fill_net, make_batch
from golden_util import load_golden, check_tensor

TOL = 1e-5          # fp32 CPU vs fp32 CPU; bit-exact in the build container, slack for other BLAS builds


# the two BASELINE configurations whose full-size oracle step takes ~1 s of CPU are replayed here too; the larger ones
# (Moving-MNIST B=128, TaxiBJ, SST with 40 predicted frames) are replayed against the HIP path in tests/test_baseline_gpu.py
FULL_ON_CPU = ['full_waveeq', 'full_mnist_b16']


@pytest.mark.parametrize('name', list(CONFIGS) + FULL_ON_CPU)
def test_oracle_matches_reference_step(name):
    cfg = CONFIGS[name] if name in CONFIGS else FULL_CONFIGS[name]
    gold = load_golden(name)
    torch.manual_seed(0)
    cond, target = make_batch(cfg)
    net = fill_net(cpu_ref.build_sep_net(cfg), cfg)
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=cfg.get('lr', 4e-4), betas=(0.9, 0.99))
    lam = cfg['lambdas']
    lamb_t = 0 if cfg.get('no_s') else lam['t']
    total, terms, forecasts, t_codes = cpu_ref.training_losses(
        cond, target, net, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], cfg.get('skipco', False), lam['ae'],
        lam['s'], lamb_t, lam['pred'], average_tloss=bool(cfg.get('average_tloss')),
        t_random=int(gold['t_random']))
    assert abs(total.item() - float(gold['total'])) <= TOL * abs(float(gold['total']))
    for k, v in terms.items():
        assert abs(v.item() - float(gold['loss:' + k])) <= TOL * max(abs(float(gold['loss:' + k])), 1e-6), k
    check_tensor(gold, 'forecasts', forecasts, TOL)
    check_tensor(gold, 't_codes', t_codes, TOL)
    opt.zero_grad()
    total.backward()
    for k, p in net.named_parameters():
        if p.grad is None:                           # never used in forward (ResNet18.bn_out): the reference has no gradient either
            assert 'grad:' + k not in gold and 'cs:grad:' + k not in gold, k
            continue
        check_tensor(gold, 'grad:' + k, p.grad, 2e-5)
    opt.step()
    for k, v in net.state_dict().items():
        check_tensor(gold, 'after:' + k, v.float(), TOL)


def test_t_random_comes_from_global_numpy_rng():
    """train.py:72-75: the window end is np.random.randint(nt_cond, T [+1 if offset != 0])."""
    cfg = CONFIGS['mlp_mul']
    gold = load_golden('mlp_mul')
    np.random.seed(cfg.get('np_seed', 1234))
    hi = cfg['nt_cond'] + cfg['nt_pred'] + (0 if cfg['offset'] == 0 else 1)
    assert int(np.random.randint(cfg['nt_cond'], hi)) == int(gold['t_random'])


EVAL_NAMES = ['mlp_mul', 'mlp_concat_partial', 'mlp_no_s', 'dcgan_tiny', 'dcgan_skip_mul', 'vgg32_tiny', 'sst_skip', 'chairs_resnet']


@pytest.mark.parametrize('name', EVAL_NAMES)
def test_oracle_eval_forecast_matches_reference_fixture(name):
    """The oracle's inference path (eval-mode BatchNorm, no_grad, horizon 12, content swap) against tests/golden/eval_<name>.npz,
    which oracle/make_golden_eval.py recorded from the reference's own `get_forecast` (test/wave/test.py:41-48 usage)."""
    import torch
    from oracle import cpu_ref
    from oracle.golden_configs import CONFIGS, fill_net, make_batch
    from golden_util import check_tensor, load_golden
    cfg = CONFIGS[name]
    gold = load_golden('eval_' + name)
    net = fill_net(cpu_ref.build_sep_net(dict(cfg)), cfg).eval()
    cond, _ = make_batch(cfg)
    skip = bool(cfg.get('skipco', False))
    with torch.no_grad():
        fore, codes, s, _ = net.get_forecast(cond, int(gold['horizon']))
        swap = net.get_forecast(cond, int(gold['swap_horizon']), init_s_code=net.Es(cond.flip(0), return_skip=skip))[0]
    check_tensor(gold, 'forecasts', fore, 1e-5)
    check_tensor(gold, 't_codes', codes, 1e-5)
    check_tensor(gold, 's_code', s[0] if isinstance(s, (tuple, list)) else s, 1e-5)
    check_tensor(gold, 'swap_forecasts', swap, 1e-5)


def test_oracle_frame_metrics_match_reference_fixture():
    """oracle/ssim_ref.py (SSIM / MSE / PSNR of the evaluation scripts) against tests/golden/frame_metrics.npz, recorded from the
    reference's `_ssim_wrapper` and the metric lines of test/mnist/test.py."""
    import torch
    from oracle import ssim_ref
    from oracle.make_golden_metrics import CASES, make_pair
    from golden_util import load_golden
    gold = load_golden('frame_metrics')
    for i, (name, shape) in enumerate(CASES.items()):
        pred, target = make_pair(shape, 100 + 10 * i)
        o = ssim_ref.frame_metrics(pred, target)
        assert torch.allclose(o['ssim_plane'], torch.from_numpy(gold[name + ':ssim']), rtol=1e-6, atol=1e-7), name
        assert torch.allclose(o['mse_plane'], torch.from_numpy(gold[name + ':mse']), rtol=1e-6, atol=1e-9), name
        assert torch.allclose(o['psnr'], torch.from_numpy(gold[name + ':psnr']), rtol=1e-6), name
