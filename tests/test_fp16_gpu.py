"""GPU: the fp16 compute mode (= the reference's --torch_amp: fp16 autocast + GradScaler, train.py:96-97, 151-155).

fp16 mode has the rounding points of the bf16 mode with IEEE-half operands (v_mfma_f32_*_f16); it is pinned the same way,
against the CPU emulation of those rounding points.  Dynamic loss scaling lives on the device (train.LossScaler): the scale is
the initial gradient of backward, vs_check_finite_multi raises found_inf, the Adam kernel unscales on the fly and skips the
whole step on overflow, vs_loss_scale_update halves / doubles the scale -- GradScaler's defaults and semantics.
"""
import numpy as np
import pytest
import torch

from oracle.detdata import det_fill
from oracle.golden_configs import CONFIGS, make_batch
from golden_util import load_golden
from step_util import compare_step_bf16, compare_step_bf16_conv

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', ['mlp_mul', 'mlp_concat_partial', 'mlp_no_s'])
def test_step_fp16_mlp_matches_rounding_point_emulation(name):
    vs_emu, vs_fp32 = compare_step_bf16(CONFIGS[name], int(load_golden(name)['t_random']), emulate=True, precision='fp16')
    print(name, 'fp16 vs emulation', {k: f'{v:.1e}' for k, v in vs_emu.items()}, 'vs fp32 oracle', {k: f'{v:.1e}' for k, v in vs_fp32.items()})


@pytest.mark.parametrize('name', ['dcgan_tiny', 'dcgan_skip_mul', 'vgg32_tiny', 'sst_skip'])
def test_step_fp16_conv_matches_rounding_point_emulation(name):
    """fp16 mode of the conv families against the independent rounding-point emulation, with a loss scale as in training (the scale keeps
    the 16-bit gradients out of the subnormal range; both sides divide it out again).  VGG / SST at batch 16 (see test_step_gpu.LOWP_BATCH)."""
    from test_step_gpu import LOWP_BATCH
    cfg = dict(CONFIGS[name], B=LOWP_BATCH.get(name, CONFIGS[name]['B']))
    compare_step_bf16_conv(name, cfg, int(load_golden(name)['t_random']), tol_out=2e-3, tol_grad=5e-2, precision='fp16', loss_scale=256.0)


def _net_and_batch(name='mlp_mul', B=8):
    from spatiotemporal_variable_separation_amd.networks.factory import build_sep_net
    cfg = dict(CONFIGS[name], B=B)
    net = det_fill(build_sep_net(cfg), salt=cfg['salt']).cuda()
    net.train()
    cond, target = make_batch(cfg)
    return cfg, net, cond.cuda(), target.cuda()


def _eager_steps(cfg, net, cond, target, precision, scaler, steps, seed=5, skip_draws=0):
    from spatiotemporal_variable_separation_amd import functional as VF
    from spatiotemporal_variable_separation_amd.optim import Adam
    from spatiotemporal_variable_separation_amd.train import compute_losses
    lam = cfg['lambdas']
    opt = Adam(net.parameters(), lr=1e-3, betas=(0.9, 0.99))
    np.random.seed(seed)
    for _ in range(skip_draws):                     # GraphedStep's capture consumes one draw of the t_random stream
        np.random.randint(cfg['nt_cond'], cond.shape[1] + target.shape[1] + (0 if cfg['offset'] == 0 else 1))
    losses = []
    with VF.precision(precision):
        for _ in range(steps):
            opt.zero_grad()
            total = compute_losses(cond, target, net, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], False, lam['ae'], lam['s'], lam['t'],
                                   lam['pred'])[0]
            if scaler is not None:
                scaler.backward(total)
                scaler.step(opt)
            else:
                total.backward()
                opt.step()
            losses.append(total.item())
    torch.cuda.synchronize()
    return opt, losses


def test_loss_scaling_is_transparent_in_fp32():
    """Scaling the loss by a power of two and unscaling inside Adam changes nothing in fp32 arithmetic (exact scaling)."""
    from spatiotemporal_variable_separation_amd.train import LossScaler
    cfg, net_a, cond, target = _net_and_batch()
    _, net_b, _, _ = _net_and_batch()
    sc = LossScaler(cond.device, init_scale=1024.0)
    _eager_steps(cfg, net_a, cond, target, 'fp32', sc, 3)
    _eager_steps(cfg, net_b, cond, target, 'fp32', None, 3)
    for (k, a), (_, b) in zip(net_a.state_dict().items(), net_b.state_dict().items()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), f'{k}: {(a - b).abs().max().item():.3e}'
    assert sc.skipped_steps() == 0 and sc.get_scale() == 1024.0


def test_overflow_skips_the_step_and_backs_the_scale_off():
    from spatiotemporal_variable_separation_amd.train import LossScaler
    cfg, net, cond, target = _net_and_batch()
    before = {k: v.clone() for k, v in net.state_dict().items()}
    # 2^100 * gradient overflows fp16 (and fp32 products): every gradient is inf / NaN -> the step must not touch anything
    sc = LossScaler(cond.device, init_scale=2.0 ** 100)
    opt, _ = _eager_steps(cfg, net, cond, target, 'fp16', sc, 1)
    for k, v in net.state_dict().items():
        assert torch.equal(v, before[k]), f'{k} changed on an overflow step'
    assert sc.skipped_steps() == 1
    assert sc.get_scale() == 2.0 ** 99
    assert float(opt.state_dict()['state'][0]['step']) == 0.0, 'the Adam step count must not advance on a skipped step'
    for st in opt.state.values():
        assert float(st['exp_avg'].abs().max()) == 0.0 and float(st['exp_avg_sq'].abs().max()) == 0.0
    # with a sane scale the next step goes through
    sc.state[0] = 1024.0
    _eager_steps(cfg, net, cond, target, 'fp16', sc, 1)
    changed = sum(int(not torch.equal(v, before[k])) for k, v in net.state_dict().items())
    assert changed > 0 and sc.skipped_steps() == 1


def test_scale_grows_after_clean_steps():
    from spatiotemporal_variable_separation_amd.train import LossScaler
    cfg, net, cond, target = _net_and_batch()
    sc = LossScaler(cond.device, init_scale=256.0, growth_interval=2)
    _eager_steps(cfg, net, cond, target, 'fp16', sc, 5)
    assert sc.get_scale() == 1024.0 and sc.skipped_steps() == 0           # doubled after steps 2 and 4


@pytest.mark.parametrize('name', ['mlp_mul', 'dcgan_tiny'])
def test_recorded_fp16_step_with_scaler_equals_eager_loop(name):
    """train.GraphedStep(scaler=...) replays what the eager fp16 loop computes (same kernels, same order), including the scale
    update; the warm-up leaves no trace (parameters, optimizer state, scale and the NumPy stream are put back)."""
    from spatiotemporal_variable_separation_amd import functional as VF
    from spatiotemporal_variable_separation_amd.optim import Adam
    from spatiotemporal_variable_separation_amd.train import GraphedStep, LossScaler
    cfg, net_a, cond, target = _net_and_batch(name, B=4)
    _, net_b, _, _ = _net_and_batch(name, B=4)
    lam = cfg['lambdas']
    sc_b = LossScaler(cond.device, init_scale=4096.0, growth_interval=2)
    _, losses_b = _eager_steps(cfg, net_b, cond, target, 'fp16', sc_b, 3, seed=9, skip_draws=1)
    with VF.precision('fp16'):
        opt = Adam(net_a.parameters(), lr=1e-3, betas=(0.9, 0.99))
        sc_a = LossScaler(cond.device, init_scale=4096.0, growth_interval=2)
        np.random.seed(9)
        gs = GraphedStep(net_a, opt, cond, target, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], (lam['ae'], lam['s'], lam['t'], lam['pred']),
                         warmup=2, scaler=sc_a)
        # (the capture itself draws one t_random -- recorded launches do not execute -- hence skip_draws=1 in the eager sequence)
        losses_a = [gs.step().item() for _ in range(3)]
    torch.cuda.synchronize()
    assert sc_a.get_scale() == sc_b.get_scale() == 8192.0
    assert np.allclose(losses_a, losses_b, rtol=2e-3), (losses_a, losses_b)
    for (k, a), (_, b) in zip(net_a.state_dict().items(), net_b.state_dict().items()):
        if a.dtype.is_floating_point:
            # same kernels in the same order; float-atomic bias sums differ in the last bit between runs and Adam (lr 1e-3)
            # amplifies that on near-zero gradients
            assert torch.allclose(a, b, rtol=2e-3, atol=2.5e-3), f'{k}: {(a - b).abs().max().item():.3e}'


def test_main_torch_amp_means_fp16_with_loss_scaling(tmp_path):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, '-m', 'spatiotemporal_variable_separation_amd.main', '--xp_dir', str(tmp_path), '--data_dir', 'synthetic',
           '--device', '0', '--epochs', '1', '--batch_size', '8', '--synthetic_len', '16', '--num_workers', '0', '--seed', '3',
           '--log_interval', '1', '--data', 'wave', '--architecture', 'mlp', '--nt_cond', '3', '--nt_pred', '4', '--offset', '3',
           '--code_size_t', '8', '--code_size_s', '8', '--mixing', 'mul', '--enc_hidden_size', '64', '--dec_hidden_size', '64',
           '--res_hidden_size', '32', '--torch_amp']
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert 'compute precision: fp16 + dynamic loss scaling' in r.stdout
