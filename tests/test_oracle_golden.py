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


@pytest.mark.parametrize('name', ['dcgan_skip_mul', 'vgg32_tiny', 'sst_skip'])
def test_independent_lowp_emulation_agrees_with_the_product_host_code_on_cpu(name):
    """oracle/bf16_emu.py holds two CPU emulations of the 16-bit modes: `emulate_bf16` interprets the ORACLE's module tree (independent of
    the product; the checker of the GPU parity tests) and `emulate_product_bf16` runs the PRODUCT's host code (networks/*.py,
    train.compute_losses) with its functional entry points replaced by the same rounding rules.  One training step through both must give
    the same forward results exactly (same CPU kernels, same rounding points, same per-call structure) and the same gradients up to the one
    stated difference: gradients of a skip tensor shared by several decoder calls are summed in fp32 and rounded once in the former, added
    one by one in 16 bits by autograd in the latter."""
    import torch
    from oracle.golden_configs import CONFIGS
    from golden_util import load_golden, rel_err
    from step_util import emulated_bf16_step, emulated_product_step, grad_err, grad_floor
    torch.set_num_threads(8)
    cfg = CONFIGS[name]
    t = int(load_golden(name)['t_random'])
    a = emulated_bf16_step(cfg, t, 'bf16')
    b = emulated_product_step(cfg, t, 'bf16')
    assert rel_err(a[3].detach().float(), b[3].detach().float()) <= 1e-6 and rel_err(a[4].detach().float(), b[4].detach().float()) <= 1e-6
    assert abs(a[1].item() - b[1].item()) <= 1e-6 * abs(b[1].item())
    bg, floor = dict(b[0].named_parameters()), grad_floor(b[0])
    for k, p in a[0].named_parameters():
        if p.grad is not None:
            assert grad_err(p.grad, bg[k].grad, floor) <= 2e-2, k
    sa, sb = a[0].state_dict(), b[0].state_dict()
    for k in sa:
        if 'running' in k:
            assert rel_err(sa[k], sb[k]) <= 1e-6, k
        if k.endswith('num_batches_tracked'):
            assert int(sa[k]) == int(sb[k]), k
