"""Shared driver for step-level parity: one training step on the HIP path vs the CPU oracle, same inputs."""
import torch

from oracle import cpu_ref
from oracle.detdata import det_fill
from oracle.golden_configs import fill_net, make_batch
from golden_util import rel_err


def grad_floor(net):
    """1e-4 x the L2 norm of the whole gradient: parameters whose true gradient is below it (a conv bias in front of a
    BatchNorm has an exactly-zero gradient that every implementation returns as summation noise) are compared in
    absolute terms on that scale instead of relative to their own (meaningless) norm."""
    tot = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in net.parameters() if p.grad is not None))
    return 1e-4 * tot.item()


def grad_err(a, b, floor):
    a, b = a.double().flatten(), b.double().flatten()
    return ((a - b).norm() / max(b.norm().item(), floor)).item()


def oracle_step(cfg, t_random, dtype=torch.float32):
    cond, target = make_batch(cfg)
    cond, target = cond.to(dtype), target.to(dtype)
    net = fill_net(cpu_ref.build_sep_net(cfg), cfg).to(dtype)
    net.train()
    lam = cfg['lambdas']
    lamb_t = 0 if cfg.get('no_s') else lam['t']
    total, terms, forecasts, t_codes = cpu_ref.training_losses(
        cond, target, net, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], cfg.get('skipco', False), lam['ae'],
        lam['s'], lamb_t, lam['pred'], average_tloss=bool(cfg.get('average_tloss')), t_random=t_random)
    total.backward()
    return net, total, terms, forecasts, t_codes


def hip_step(cfg, t_random, oracle_net, precision='fp32', fused=True):
    from spatiotemporal_variable_separation_amd import functional as VF
    from spatiotemporal_variable_separation_amd.networks.factory import build_sep_net
    from spatiotemporal_variable_separation_amd.train import compute_losses
    cond, target = make_batch(cfg)
    net = build_sep_net(cfg)
    # identical parameters AND a check that the state-dict layout matches the reference-compatible oracle
    sd = {k: v.clone() for k, v in oracle_net.state_dict().items()}
    net.load_state_dict(sd, strict=True)
    net = net.cuda()
    net.train()
    net.fused = fused
    lam = cfg['lambdas']
    lamb_t = 0 if cfg.get('no_s') else lam['t']
    with VF.precision(precision):
        total, terms, forecasts, t_codes = compute_losses(
            cond.cuda(), target.cuda(), net, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], cfg.get('skipco', False),
            lam['ae'], lam['s'], lamb_t, lam['pred'], average_tloss=bool(cfg.get('average_tloss')), t_random=t_random)
        total.backward()
    torch.cuda.synchronize()
    return net, total, terms, forecasts, t_codes


_LOWP_DTYPE = {'bf16': torch.bfloat16, 'fp16': torch.float16}


def emulated_bf16_step(cfg, t_random, precision='bf16'):
    """The fp32 oracle with the product's bf16 (fp16) rounding points patched in (oracle/bf16_emu.py)."""
    from oracle.bf16_emu import emulate_bf16
    with emulate_bf16(_LOWP_DTYPE[precision]):
        return oracle_step(cfg, t_random)


def emulated_product_step(cfg, t_random, precision='bf16'):
    """bf16 mode of the conv families: the product's module tree and host logic on the CPU with every functional entry point
    replaced by its bf16-rounding torch emulation (oracle/bf16_emu.py, second half)."""
    from oracle.bf16_emu import emulate_product_bf16
    from spatiotemporal_variable_separation_amd.networks.factory import build_sep_net
    from spatiotemporal_variable_separation_amd.train import compute_losses
    cond, target = make_batch(cfg)
    o_net = fill_net(cpu_ref.build_sep_net(cfg), cfg)
    lam = cfg['lambdas']
    lamb_t = 0 if cfg.get('no_s') else lam['t']
    with emulate_product_bf16(_LOWP_DTYPE[precision]):
        net = build_sep_net(cfg)
        net.load_state_dict({k: v.clone() for k, v in o_net.state_dict().items()}, strict=True)
        net.train()
        total, terms, forecasts, t_codes = compute_losses(
            cond, target, net, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], cfg.get('skipco', False), lam['ae'], lam['s'], lamb_t,
            lam['pred'], average_tloss=bool(cfg.get('average_tloss')), t_random=t_random)
        total.backward()
    return net, total, terms, forecasts, t_codes


def compare_step_bf16_conv(cfg, t_random, tol_out=2e-3, tol_grad=5e-2, precision='bf16'):
    """HIP bf16 step of a conv family against the emulation above: same rounding points, so outputs / losses agree to
    accumulation-order noise and one-ulp bf16 flips (tol_out); gradients go through ill-conditioned per-call BatchNorm stacks at
    batch 2-3 (see compare_step), so they get the wider tol_grad -- still 10x tighter than anything bf16 vs fp32 could give."""
    o_net0 = fill_net(cpu_ref.build_sep_net(cfg), cfg)
    h_net, h_total, h_terms, h_fore, h_tc = hip_step(cfg, t_random, o_net0, precision)
    e_net, e_total, e_terms, e_fore, e_tc = emulated_product_step(cfg, t_random, precision)
    errs = {'forecasts': rel_err(h_fore.detach().cpu().float(), e_fore.detach().float()),
            't_codes': rel_err(h_tc.detach().cpu().float(), e_tc.detach().float()),
            'total': abs(h_total.item() - e_total.item()) / abs(e_total.item())}
    for k, v in errs.items():
        assert v <= tol_out, f'{k}: HIP bf16 vs bf16-emulating product-on-CPU {v:.3e} > {tol_out:.1e}'
    eg = dict(e_net.named_parameters())
    floor = grad_floor(e_net)
    worst = 0.0
    for k, p in h_net.named_parameters():
        e = grad_err(p.grad.detach().cpu(), eg[k].grad, floor)
        worst = max(worst, e)
        assert e <= tol_grad, f'gradient {k}: HIP bf16 vs emulation {e:.3e} > {tol_grad:.1e}'
    errs['grad_worst'] = worst
    # BatchNorm running statistics after the step
    esd = e_net.state_dict()
    for k, v in h_net.state_dict().items():
        if k.endswith('running_mean') or k.endswith('running_var'):
            assert rel_err(v.detach().cpu(), esd[k]) <= tol_out, k
    return errs


def compare_step_bf16(cfg, t_random, tol=2e-3, sanity=0.6, emulate=True, precision='bf16'):
    """bf16 mode: (1) MLP family: must match the CPU emulation of its own rounding scheme to `tol` relative L2
    (accumulation-order noise only); (2) every family: outputs within 5e-2 and gradients within a loose `sanity` bound
    of the fp32 oracle (the conv kernels' bf16 arithmetic is pinned exactly, op by op, in tests/test_conv_gpu.py)."""
    o_net0 = fill_net(cpu_ref.build_sep_net(cfg), cfg)
    h_net, h_total, h_terms, h_fore, h_tc = hip_step(cfg, t_random, o_net0, precision)
    o_net, o_total, o_terms, o_fore, o_tc = oracle_step(cfg, t_random)
    if emulate:
        e_net, e_total, e_terms, e_fore, e_tc = emulated_bf16_step(cfg, t_random, precision)

    def errors(r_net, r_total, r_fore, r_tc):
        rg = dict(r_net.named_parameters())
        floor = grad_floor(r_net)
        return {'forecasts': rel_err(h_fore.detach().cpu(), r_fore.detach()),
                't_codes': rel_err(h_tc.detach().cpu(), r_tc.detach()),
                'total': abs(h_total.item() - r_total.item()) / abs(r_total.item()),
                'grad_worst': max(grad_err(p.grad.detach().cpu(), rg[k].grad, floor) for k, p in h_net.named_parameters()
                                  if rg[k].grad is not None)}
    vs_emu = errors(e_net, e_total, e_fore, e_tc) if emulate else {}
    vs_fp32 = errors(o_net, o_total, o_fore, o_tc)
    for k, v in vs_emu.items():
        assert v <= tol, f'{k}: HIP bf16 vs bf16-emulating oracle {v:.3e} > {tol:.1e}'
    for k, v in vs_fp32.items():
        if emulate:
            bound = sanity if k == 'grad_worst' else 5e-2
        else:
            # conv families: deep per-call BatchNorm stacks at batch 2-3 amplify rounding ~300x (fp32 vs fp64 oracle:
            # 2e-5 on forecasts, up to 3e-2 on gradients), so bf16 operand rounding (4e-3) moves gradients by O(1) for
            # ANY implementation; only outputs are bounded here, gradients must be finite.  The bf16 conv / BatchNorm
            # arithmetic itself is pinned exactly in tests/test_conv_gpu.py.
            bound = float('inf') if k == 'grad_worst' else 0.25
        assert v <= bound and v == v, f'{k}: HIP bf16 vs fp32 oracle {v:.3e} > bound {bound}'
    return vs_emu, vs_fp32


def compare_step(cfg, t_random, precision, tol_out, tol_grad, fused=True):
    """Returns a dict of worst relative errors; asserts against the tolerances."""
    o_net0 = fill_net(cpu_ref.build_sep_net(cfg), cfg)          # pristine weights for the HIP net
    h_net, h_total, h_terms, h_fore, h_tc = hip_step(cfg, t_random, o_net0, precision, fused=fused)
    o_net, o_total, o_terms, o_fore, o_tc = oracle_step(cfg, t_random)
    errs = {'forecasts': rel_err(h_fore.detach().cpu(), o_fore.detach()),
            't_codes': rel_err(h_tc.detach().cpu(), o_tc.detach()),
            'total': abs(h_total.item() - o_total.item()) / abs(o_total.item())}
    for k in o_terms:
        errs['loss:' + k] = abs(h_terms[k].item() - o_terms[k].item()) / max(abs(o_terms[k].item()), 1e-8)
    for k in ('forecasts', 't_codes', 'total') + tuple('loss:' + k for k in o_terms):
        assert errs[k] <= tol_out, f'{k}: {errs[k]:.3e} > {tol_out:.1e} ({precision})'
    # Gradients: deep BatchNorm stacks at batch 2-3 are ill-conditioned -- the fp32 CPU oracle itself sits 4e-3..3e-2
    # away from its own fp64 evaluation on some of these configs.  So every gradient is measured against the fp64
    # oracle and must be within max(tol_grad, 20 x the fp32 oracle's own distance to fp64) for that parameter (the
    # exact-fp32 MFMA accumulates each output in ONE k-ordered chain, K up to 4608, where oneDNN sums 16-lane partial
    # chains: a few times more rounding noise per layer, amplified alike by the ill-conditioned backward pass).
    d_net = oracle_step(cfg, t_random, dtype=torch.float64)[0]
    dg = dict(d_net.named_parameters())
    og = dict(o_net.named_parameters())
    floor = grad_floor(d_net)
    worst_g, worst_ratio = 0.0, 0.0
    for k, p in h_net.named_parameters():
        if dg[k].grad is None:                       # a parameter the forward never uses (ResNet18.bn_out, conv.py:526)
            assert p.grad is None, f'{k}: the oracle has no gradient here'
            continue
        assert p.grad is not None, f'no gradient for {k}'
        e_hip = grad_err(p.grad.detach().cpu(), dg[k].grad, floor)
        e_ref = grad_err(og[k].grad, dg[k].grad, floor)
        bound = max(tol_grad, 20.0 * e_ref)
        worst_g = max(worst_g, e_hip)
        worst_ratio = max(worst_ratio, e_hip / bound)
        assert e_hip <= bound, (f'gradient {k}: HIP vs fp64 oracle {e_hip:.3e} > bound {bound:.3e} '
                                f'(fp32 oracle vs fp64: {e_ref:.3e}) ({precision})')
    errs['grad_worst'] = worst_g
    errs['grad_worst_over_bound'] = worst_ratio
    # BN running statistics after the step (per-call updates)
    osd = o_net.state_dict()
    for k, v in h_net.state_dict().items():
        if k.endswith('running_mean') or k.endswith('running_var'):
            e = rel_err(v.detach().cpu(), osd[k])
            assert e <= tol_out, f'{k}: {e:.3e}'
        if k.endswith('num_batches_tracked'):
            assert int(v) == int(osd[k]), k
    return errs
