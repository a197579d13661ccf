"""Direct checks of the fused step kernels around the decoder: vs_mix_codes_* (decoder input of a rollout) and the
device-side target-frame selection of vs_train_losses_* (train.py:72-88, 117-149; mlp_encdec.py:43-48)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mix_ref(s, t_rand, t_codes, mixing):
    t_all = torch.cat([t_rand.unsqueeze(1), t_codes], dim=1)
    se = s.unsqueeze(1).expand(-1, t_all.shape[1], -1)
    return torch.cat([se, t_all], dim=2) if mixing == 'concat' else se * t_all


@pytest.mark.parametrize('mixing,B,n,Cs,Ct', [('concat', 128, 10, 32, 32), ('mul', 128, 10, 32, 32), ('concat', 5, 3, 7, 20),
                                               ('mul', 3, 1, 300, 300), ('concat', 2, 0, 4, 4)])
def test_mix_codes_matches_torch(mixing, B, n, Cs, Ct):
    from spatiotemporal_variable_separation_amd import functional as VF, ops
    g = torch.Generator().manual_seed(B * 100 + n)
    s, t_rand, t_codes = torch.randn(B, Cs, generator=g), torch.randn(B, Ct, generator=g), torch.randn(B, n, Ct, generator=g)
    dz = torch.randn(B, n + 1, Cs if mixing == 'mul' else Cs + Ct, generator=g)
    leaves = [x.clone().requires_grad_(True) for x in (s, t_rand, t_codes)]
    ref = _mix_ref(*leaves, mixing)
    ref.backward(dz)

    dev = [x.cuda().requires_grad_(True) for x in (s, t_rand, t_codes)]
    with VF.precision('bf16'):
        z, z_lowp = VF.MixCodes.apply(*dev, mixing)
    assert torch.equal(z.cpu(), ref.detach())                       # one product or a copy per element: exact
    assert z_lowp.dtype == torch.bfloat16 and torch.equal(z_lowp.cpu(), ref.detach().bfloat16())
    z.backward(dz.cuda())
    for got, want in zip(dev, leaves):
        torch.testing.assert_close(got.grad.cpu(), want.grad, rtol=1e-5, atol=1e-6)    # ds: sum over the frames, fp32 order
    with VF.precision('fp32'):
        z32, none = VF.MixCodes.apply(*[x.detach() for x in dev], mixing)
    assert none is None and torch.equal(z32, z)
    with pytest.raises(Exception):
        ops.mix_codes_fwd(s.cuda(), t_rand.cuda(), t_codes.cuda(), 'sum')


@pytest.mark.parametrize('offset', [0, 1])
def test_train_losses_device_window_equals_index_vector(offset):
    from spatiotemporal_variable_separation_amd import ops
    B, T, D, nt_cond, n = 6, 9, 515, 3, 5 + offset
    G = 1 + n
    g = torch.Generator().manual_seed(7 + offset)
    frames = torch.randn(B, G, D, generator=g).cuda()
    full = torch.randn(B, T + 1, D, generator=g).cuda()
    s_old, s_new, t0 = (torch.randn(B, 19, generator=g).cuda() for _ in range(3))
    fo = nt_cond if offset == 0 else 0
    lambdas = (10.0, 45.0, 0.001, 45.0)
    gt = torch.tensor(1.0).cuda()
    for t in (nt_cond, T - 1, T if offset else T - 2):
        idx = torch.tensor([t - offset] + list(range(fo, fo + n)), dtype=torch.int32).cuda()
        t_dev = torch.tensor([t], dtype=torch.int32).cuda()
        a = ops.train_losses_fwd(frames, full, idx, s_old, s_new, t0, lambdas, False)
        b = ops.train_losses_fwd(frames, full, (t_dev, offset, fo), s_old, s_new, t0, lambdas, False)
        torch.testing.assert_close(a[4:9], b[4:9], rtol=1e-5, atol=1e-6)     # partial sums meet in float atomics: order varies
        ga = ops.train_losses_bwd(frames, full, idx, s_old, s_new, t0, lambdas, False, gt)
        gb = ops.train_losses_bwd(frames, full, (t_dev, offset, fo), s_old, s_new, t0, lambdas, False, gt)
        for x, y in zip(ga, gb):
            assert torch.equal(x, y)


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
@pytest.mark.parametrize('extra_consumer', [False, True])
def test_loss_gradient_handoff_equals_two_pass_backward(precision, extra_consumer):
    """The fused loss writes the decoder chain's last pre-activation gradient itself (functional.GradHandoff); the result must be
    the one of the two-pass backward (fp32 frame gradient, then act_bwd), also when the frames feed a second consumer."""
    import torch.nn as nn
    from spatiotemporal_variable_separation_amd import functional as VF
    B, G, D, T = 4, 3, 64, 6
    torch.manual_seed(3)
    lin = [nn.Linear(8, 16).cuda(), nn.Linear(16, D).cuda()]
    x = torch.randn(B * G, 8).cuda()
    full = torch.rand(B, T, D).cuda()
    s_old, s_new, t0 = (torch.randn(B, 5).cuda() for _ in range(3))
    idx = torch.tensor([2, 3, 4], dtype=torch.int32).cuda()

    def run(handoff):
        for l in lin:
            l.zero_grad()
        with VF.precision(precision):
            y = VF.mlp_chain(x, lin, out_act='sigmoid', handoff=handoff)
            total = VF.TrainLosses.apply(y.view(B, G, D), full, idx, s_old, s_new, t0, (10.0, 1.0, 0.1, 5.0), False, handoff)[0]
            if extra_consumer:
                total = total + (y * y).sum() * 0.01
            total.backward()
        return [p.grad.clone() for l in lin for p in l.parameters()]

    want = run(None)
    h = VF.GradHandoff()
    got = run(h)
    assert h.act == 'sigmoid' and h.dz is None          # filled by the loss, consumed by the chain
    for a, b in zip(got, want):
        if extra_consumer:
            torch.testing.assert_close(a, b, rtol=2e-2 if precision == 'bf16' else 1e-5, atol=1e-4 if precision == 'bf16' else 1e-7)
        else:
            assert torch.equal(a, b)                     # same arithmetic, same rounding point


@pytest.mark.gpu
@pytest.mark.parametrize('out_dtype', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('D', [4096, 30])            # vectorised (multiples of 4) and scalar paths
def test_copy2d_pair_builds_the_two_encoder_windows_in_one_launch(out_dtype, D):
    """vs_copy2d_pair: rows [0, B) = the window that ends at the device-side t, rows [B, 2B) = the conditioning window (train.py:45-88)."""
    import torch
    from spatiotemporal_variable_separation_amd import ops
    B, T, nc = 6, 11, 3
    g = torch.Generator().manual_seed(9)
    full = torch.rand((B, T, D), generator=g).cuda()
    for t in (3, 7, 11):
        tdev = torch.tensor([t], dtype=torch.int32, device='cuda')
        out = torch.empty((2 * B, nc * D), dtype=out_dtype, device='cuda')
        ops.copy2d_pair(full.view(B, T * D), B, nc * D, T * D, out, nc * D, tdev, D, -nc * D, 0)
        want = torch.cat([full[:, t - nc:t].reshape(B, -1), full[:, :nc].reshape(B, -1)], dim=0).to(out_dtype)
        assert torch.equal(out, want), (t, D, out_dtype)


@pytest.mark.parametrize('skipco,average,spatial', [(False, False, False), (True, True, True), (True, False, False), (False, True, True)])
def test_conv_losses_fused_equal_torch_assembly(skipco, average, spatial):
    """functional.conv_losses (round 4: the conv families' four losses and their weighted sum in 4 launches forward / 3 backward, the zero-order
    loss over the code AND every skip tensor read where they lie -- fp32 code, 16-bit skips -- without concatenation) against the torch
    assembly of train.py:38-42, 85-86, 139-149: values to 1e-6, every gradient (frames, both codes, every skip in its own dtype, the initial
    temporal code) to 1e-5 / the 16-bit rounding of the skip gradients."""
    import torch.nn.functional as F
    from spatiotemporal_variable_separation_amd import functional as VF
    torch.manual_seed(3)
    B, T, n_f, C, H, W = 4, 9, 5, 2, 16, 16
    full = torch.rand(B, T, C, H, W, device='cuda')
    recon = torch.rand(B, C, H, W, device='cuda', requires_grad=True)
    fore = torch.rand(B, n_f, C, H, W, device='cuda', requires_grad=True)
    ae_frame, first = 6, 4
    ae_idx = torch.tensor([ae_frame], dtype=torch.int32, device='cuda')
    f_idx = torch.arange(first, first + n_f, dtype=torch.int32, device='cuda')
    code_shape = (B, 24, 8, 8) if spatial else (B, 40)
    t0 = torch.randn(code_shape, device='cuda', requires_grad=True)

    def make_s():
        code = torch.randn(B, 40, device='cuda', requires_grad=True)
        if not skipco:
            return code
        skips = [torch.randn(B, 16, 8, 8, device='cuda').bfloat16().requires_grad_(True), torch.randn(B, 8, 16, 16, device='cuda').bfloat16().requires_grad_(True)]
        return (code, skips)
    s_old, s_new = make_s(), make_s()
    lam = (1.7, 45.0, 0.01, 30.0)                    # (ae, s, t, pred)

    def leaves():
        out = [recon, fore, t0]
        for s in (s_old, s_new):
            out += ([s[0]] + list(s[1])) if skipco else [s]
        return out

    def torch_total():
        ae = F.mse_loss(full[:, ae_frame], recon)
        pred = F.mse_loss(fore, full[:, first:first + n_f])
        if skipco:
            a = torch.cat([s_old[0].flatten().float()] + [x.flatten().float() for x in s_old[1]])
            b = torch.cat([s_new[0].flatten().float()] + [x.flatten().float() for x in s_new[1]])
        else:
            a, b = s_old, s_new
        zero = (a - b).pow(2).mean()
        treg = 0.5 * (t0.pow(2).view(B, -1)).mean() if average else 0.5 * torch.sum(t0.pow(2), dim=1).mean()
        return lam[0] * ae + lam[1] * zero + lam[3] * pred + lam[2] * treg, (ae, zero, pred, treg)
    want, terms = torch_total()
    (want * 3.0).backward()
    ref = [x.grad.clone() for x in leaves()]
    for x in leaves():
        x.grad = None
    got = VF.conv_losses(recon, fore, full, ae_idx, f_idx, s_old, s_new, skipco, t0, lam, average)
    assert got is not None
    total, d = got
    (total * 3.0).backward()
    torch.cuda.synchronize()
    assert abs(total.item() - want.item()) <= 1e-5 * abs(want.item())
    for k, v in zip(('ae', 'zero', 'pred', 't_reg'), terms):
        assert abs(d[k].item() - v.item()) <= 1e-5 * abs(v.item()) + 1e-9, k
    for x, r in zip(leaves(), ref):
        assert x.grad is not None and x.grad.dtype == r.dtype
        tol = 1e-5 if r.dtype == torch.float32 else 8e-3
        assert ((x.grad.float() - r.float()).abs().max() <= tol * r.float().abs().max() + 1e-12), (tuple(x.shape), r.dtype)


@pytest.mark.parametrize('a_dtype,x_dtype,out_dtype', [(torch.float32, torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16, torch.bfloat16),
                                                       (torch.float16, torch.float32, torch.float16)])
def test_cat_bcast_equals_repeat_and_cat(a_dtype, x_dtype, out_dtype):
    """functional.cat_bcast (decoder inputs of a batched rollout, conv.py:228, 388-394): cat([a.repeat(n, 1, 1, 1), x], 1) in one pass, its
    gradients (d a = sum over the frames, d x = the other channels) in one pass -- bit for bit against torch in fp32, to the rounding of the
    sum in 16 bits."""
    from spatiotemporal_variable_separation_amd import functional as VF
    torch.manual_seed(1)
    B, n, Ca, Cb, H, W = 3, 5, 7, 6, 8, 16
    a = torch.randn(B, Ca, H, W, device='cuda').to(a_dtype).requires_grad_(True)
    x = torch.randn(n * B, Cb, H, W, device='cuda').to(x_dtype).requires_grad_(True)
    g = torch.randn(n * B, Ca + Cb, H, W, device='cuda').to(out_dtype)
    want = torch.cat([a.repeat(n, 1, 1, 1).to(out_dtype), x.to(out_dtype)], dim=1)
    want.backward(g)
    ga, gx = a.grad.clone(), x.grad.clone()
    a.grad = x.grad = None
    got = VF.cat_bcast(a, x, n, out_dtype)
    assert isinstance(got.grad_fn, torch.autograd.function.BackwardCFunction) and torch.equal(got, want)
    got.backward(g)
    torch.cuda.synchronize()
    assert x.grad.dtype == gx.dtype and torch.equal(x.grad, gx)
    assert a.grad.dtype == ga.dtype
    tol = 0.0 if a_dtype == torch.float32 else 8e-3
    assert (a.grad.float() - ga.float()).abs().max().item() <= tol * ga.float().abs().max().item() + (1e-6 if tol == 0.0 else 0.0)


# ---- the frame losses in the epilogue of the decoder's last GEMM (vs_gemm_frame_loss) -----------------------------------------------------
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('act', ['sigmoid', 'none'])
@pytest.mark.parametrize('B,G,N,K,offset,tile', [(16, 6, 640, 128, 3, 'big'), (7, 5, 520, 72, 0, 'big'), (128, 2, 256, 64, 2, 'big'),
                                                  (64, 8, 512, 128, 3, 'p8'), (70, 8, 776, 200, 0, 'p8'), (128, 6, 1024, 64, 2, 'p8')])
def test_gemm_frame_loss_equals_gemm_then_fused_losses(dtype, act, B, G, N, K, offset, tile, monkeypatch):
    """ops.gemm_frame_loss == ops.gemm (fp32 frames) followed by ops.train_losses_fwd_grad: the gradient of the pre-activation and the
    code gradients BIT-equal (same fp32 formula on the same values), the five scalars equal up to the order of the frame sums.  `tile`: the
    round-5 256 x 256 ring tile, or the staggered tile ('p8': tiles inside the matrix take its straight-line epilogue -- compiled for the
    activations none / sigmoid --, tiles on the ragged edge the general one; 512 x 512 and 768 x 1024 are interior only)."""
    from spatiotemporal_variable_separation_amd import ops
    from oracle.detdata import det_uniform
    monkeypatch.setenv('VS_GEMM_BIG', '2')               # the 256x256 tile kernel whatever the size (the plan takes it at WaveEq size only)
    monkeypatch.setenv('VS_GEMM_P8', '2' if tile == 'p8' else '0')
    T = G + 4
    h = ((det_uniform((B * G, K), 3) - 0.5) * 2).to(dtype).cuda()
    w = ((det_uniform((N, K), 5) - 0.5) * 0.2).to(dtype).cuda()
    bias = ((det_uniform((N,), 7) - 0.5) * 0.1).cuda()
    full = det_uniform((B, T, N), 9).cuda()
    s_old, s_new = det_uniform((B, 12), 11).cuda(), det_uniform((B, 12), 13).cuda()
    t0 = (det_uniform((B, 10), 15) - 0.5).cuda()
    t_dev = torch.tensor([G + 1], dtype=torch.int32, device='cuda')
    idx = (t_dev, offset, 2 if offset == 0 else 0)
    up = torch.full((), 0.75, dtype=torch.float32, device='cuda')
    lam = (10.0, 45.0, 0.001, 45.0)
    for avg in (False, True):
        frames = ops.gemm(h, 0, w, 0, B * G, N, K, bias=bias, act=act)
        ref = ops.train_losses_fwd_grad(frames.view(B, G, N), full, idx, s_old, s_new, t0, lam, avg, up, act if act != 'none' else 'none', dtype)
        got = ops.gemm_frame_loss(h, w, bias, act, full, idx, G, s_old, s_new, t0, lam, avg, up, dtype)
        assert got is not None
        torch.cuda.synchronize()
        assert torch.equal(got[1].view(B, G, N), ref[1]), 'dz'
        for a, b in zip(got[2:], ref[2:]):
            assert torch.equal(a, b)
        assert torch.allclose(got[0][4:9], ref[0][4:9], rtol=2e-6, atol=0), (got[0][:9], ref[0][:9])
    # a problem the 256x256 tile does not take: the caller is told to use the two launches
    monkeypatch.setenv('VS_GEMM_BIG', '1')
    monkeypatch.setenv('VS_GEMM_P8', '1')
    assert ops.gemm_frame_loss(h, w, bias, act, full, idx, G, s_old, s_new, t0, lam, False, up, dtype) is None


def test_recorded_waveeq_step_with_the_losses_in_the_gemm_epilogue(monkeypatch):
    """BASELINE configs[1] (WaveEq MLP, B = 128, bf16), recorded step: with the frame losses evaluated in the epilogue of the decoder's last
    GEMM (default) the parameters after one step equal (to the run-to-run noise of the float-atomic bias sums) those of the step that stores the frames and runs the loss pass, and the
    reported losses agree to the order of the sums."""
    import numpy as np
    from spatiotemporal_variable_separation_amd import functional as VF
    from spatiotemporal_variable_separation_amd.configs import BASELINE_CONFIGS
    from spatiotemporal_variable_separation_amd.data.synthetic import synthetic_batch
    from spatiotemporal_variable_separation_amd.networks.factory import build_sep_net
    from spatiotemporal_variable_separation_amd.optim import Adam
    from spatiotemporal_variable_separation_amd.train import GraphedStep
    cfg = dict(BASELINE_CONFIGS['waveeq'])
    results = {}
    VF.set_precision('bf16')
    try:
        for mode in ('1', '0'):
            monkeypatch.setenv('VARSEP_FUSE_FRAME_LOSS', mode)
            torch.manual_seed(1234)
            np.random.seed(1234)
            net = build_sep_net(cfg).cuda().train()
            opt = Adam(net.parameters(), lr=4e-4, betas=(0.9, 0.99))
            cond, target = synthetic_batch(cfg['data'], cfg['batch'], cfg['nt_cond'], cfg['nt_pred'], device=torch.device('cuda'), seed=1234)
            lam = cfg['lambdas']
            gs = GraphedStep(net, opt, cond, target, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], (lam['ae'], lam['s'], lam['t'], lam['pred']),
                             warmup=1)
            losses = [gs.step().item() for _ in range(1)]
            torch.cuda.synchronize()
            results[mode] = (losses, {k: v.detach().clone() for k, v in net.state_dict().items()})
            del gs, opt, net
    finally:
        VF.set_precision('fp32')
    (la, pa), (lb, pb) = results['1'], results['0']
    assert np.allclose(la, lb, rtol=1e-5), (la, lb)
    for k in pa:
        # ONE step: the loss it reports and the parameters it leaves.  The step is not bit-reproducible (float-atomic bias sums), and Adam's first
        # step is lr * sign(g): a last-bit difference of a noise-level gradient flips an element by 2 lr.  The gradient of the pre-activation itself
        # IS bit-equal (op test above); here: all but a few elements agree to 1e-6, none is further apart than 2.5 learning rates
        d = (pa[k] - pb[k]).abs()
        assert float((d > 1e-6).float().mean()) <= 1e-3 and float(d.max()) <= 2.5 * 4e-4, f'{k}: {float((d > 1e-6).float().mean()):.2e} of the elements differ, max {d.max().item():.3e}'
