"""HIP multi-tensor Adam (vs_adam_multi) against torch.optim.Adam, the optimizer the reference constructs (main.py:133)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(1200, 777), (32,), (5, 3, 4, 4), (4096 * 3 + 5,), (1,), (64, 64)]
    return [torch.nn.Parameter((torch.rand(s, generator=g) - 0.5).cuda()) for s in shapes]


@pytest.mark.parametrize('lr,betas', [(4e-4, (0.9, 0.99)), (1e-3, (0.5, 0.999))])
def test_adam_matches_torch(lr, betas):
    from spatiotemporal_variable_separation_amd.optim import Adam
    pa, pb = _params(0), _params(0)
    oa = Adam(pa, lr=lr, betas=betas)
    ob = torch.optim.Adam(pb, lr=lr, betas=betas)
    g = torch.Generator().manual_seed(1)
    for step in range(5):
        for a, b in zip(pa, pb):
            gr = (torch.rand(a.shape, generator=g) - 0.5).cuda() * (10.0 ** (step - 2))
            a.grad, b.grad = gr.clone(), gr.clone()
        if step == 3:
            pa[1].grad = None                      # a parameter without gradient is skipped, like torch does
            pb[1].grad = None
        v0 = pa[0]._version
        oa.step()
        ob.step()
        assert pa[0]._version > v0                 # operand caches keyed on the version counter see the update
        for a, b in zip(pa, pb):
            assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), (step, a.shape, (a - b).abs().max().item())
    sa, sb = oa.state_dict(), ob.state_dict()
    for k in sb['state']:
        for name in ('exp_avg', 'exp_avg_sq'):
            assert torch.allclose(sa["state"][k][name], sb["state"][k][name], rtol=1e-5, atol=1e-6)
    assert float(sa['state'][0]['step']) == 5.0


def test_adam_accepts_views_into_flat_gradient_buckets():
    """Gradients as views at odd element offsets of one flat buffer (parallel.GradAllReducer): the misaligned ones take the
    scalar path of the kernel."""
    from spatiotemporal_variable_separation_amd.optim import Adam
    pa, pb = _params(2), _params(2)
    flat = torch.zeros(sum(p.numel() for p in pa) + 3, device='cuda')
    off = 3
    for p in pa:
        p.grad = flat[off:off + p.numel()].view_as(p)
        off += p.numel()
    oa, ob = Adam(pa, lr=1e-3), torch.optim.Adam(pb, lr=1e-3)
    g = torch.Generator().manual_seed(4)
    for _ in range(3):
        for a, b in zip(pa, pb):
            gr = (torch.rand(a.shape, generator=g) - 0.5).cuda()
            a.grad.copy_(gr)
            b.grad = gr.clone()
        oa.step()
        ob.step()
    for a, b in zip(pa, pb):
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-7)


def test_adam_writes_bf16_shadow_and_is_graph_capturable():
    from spatiotemporal_variable_separation_amd import functional as VF
    from spatiotemporal_variable_separation_amd.optim import Adam
    p = _params(3)[:2]
    sh = VF.shadow(p[0], torch.bfloat16)                     # creates the cached operand copy
    opt = Adam(p, lr=1e-2, betas=(0.9, 0.99))
    for q in p:
        q.grad = torch.ones_like(q)
    opt.step()                                                # warm-up (allocates state) outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        opt.step()
    ref = [q.detach().clone() for q in p]
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert not torch.equal(ref[0], p[0])
    assert VF.shadow(p[0], torch.bfloat16).data_ptr() == sh.data_ptr()
    assert torch.equal(sh, p[0].detach().to(torch.bfloat16))  # refreshed in the same pass, no separate cast
    # 1 eager + 3 replayed steps == 4 torch steps
    q = _params(3)[:2]
    ob = torch.optim.Adam(q, lr=1e-2, betas=(0.9, 0.99))
    for _ in range(4):
        for t in q:
            t.grad = torch.ones_like(t)
        ob.step()
    for a, b in zip(p, q):
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-7)


def test_adam_rejects_cpu_parameters():
    from spatiotemporal_variable_separation_amd.optim import Adam
    from spatiotemporal_variable_separation_amd._lib import VarsepHipError
    p = [torch.nn.Parameter(torch.zeros(4))]
    p[0].grad = torch.ones(4)
    with pytest.raises(VarsepHipError):
        Adam(p).step()


def test_bf16_training_follows_the_fp32_master_weights():
    """Regression: the bf16 operand copies must track the optimizer.  HIP Adam (writes them in its own pass) and torch's
    non-fused Adam (bumps the version counters) give the same loss trajectory; torch's fused Adam is refused."""
    import numpy as np
    from oracle.detdata import det_fill
    from oracle.golden_configs import CONFIGS, make_batch
    from spatiotemporal_variable_separation_amd import functional as VF
    from spatiotemporal_variable_separation_amd.networks.factory import build_sep_net
    from spatiotemporal_variable_separation_amd.optim import Adam
    from spatiotemporal_variable_separation_amd.train import check_optimizer, compute_losses
    cfg = CONFIGS['mlp_mul']
    lam = cfg['lambdas']
    cond, target = make_batch(cfg)
    cond, target = cond.cuda(), target.cuda()

    def run(make):
        net = det_fill(build_sep_net(cfg), salt=cfg['salt']).cuda().train()
        opt = make(net.parameters())
        np.random.seed(3)
        out = []
        with VF.precision('bf16'):
            for _ in range(8):
                opt.zero_grad(set_to_none=True)
                total = compute_losses(cond, target, net, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], False, lam['ae'], lam['s'],
                                       lam['t'], lam['pred'])[0]
                total.backward()
                opt.step()
                out.append(total.item())
        return out
    hip = run(lambda ps: Adam(ps, lr=2e-3, betas=(0.9, 0.99)))
    ref = run(lambda ps: torch.optim.Adam(ps, lr=2e-3, betas=(0.9, 0.99)))
    assert hip[-1] < 0.9 * hip[0]                              # it trains
    assert np.allclose(hip, ref, rtol=2e-2), (hip, ref)
    with pytest.raises(ValueError):
        check_optimizer(torch.optim.Adam([torch.nn.Parameter(torch.zeros(4, device='cuda'))], fused=True))


@pytest.mark.parametrize('graph', [False, True])
def test_update_in_backward_equals_plain_step(graph):
    """optim.Adam.overlap_with_backward (each network updated on a side stream as soon as its gradients are final) changes
    the schedule, not the arithmetic: same parameters as the plain step() after several steps, eager and as a recorded graph."""
    import numpy as np
    from oracle.detdata import det_fill
    from oracle.golden_configs import CONFIGS, make_batch
    from spatiotemporal_variable_separation_amd.networks.factory import build_sep_net
    from spatiotemporal_variable_separation_amd.optim import Adam
    from spatiotemporal_variable_separation_amd.train import GraphedStep, compute_losses, enable_update_in_backward
    cfg = CONFIGS['mlp_mul']
    lam = cfg['lambdas']
    cond, target = make_batch(cfg)
    cond, target = cond.cuda(), target.cuda()

    def run(overlap):
        net = det_fill(build_sep_net(cfg), salt=cfg['salt']).cuda().train()
        opt = Adam(net.parameters(), lr=1e-3, betas=(0.9, 0.99))
        if overlap:
            enable_update_in_backward(opt, net, force=True)
            assert opt._buckets
        else:
            opt._buckets = [[]]                      # keeps GraphedStep from switching the overlap on
        np.random.seed(11)
        if graph:
            g = GraphedStep(net, opt, cond, target, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], (lam['ae'], lam['s'], lam['t'], lam['pred']),
                            warmup=2)
            for _ in range(4):
                g.step()
        else:
            for _ in range(5):
                opt.zero_grad(set_to_none=True)
                compute_losses(cond, target, net, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], False, lam['ae'], lam['s'], lam['t'],
                               lam['pred'])[0].backward()
                opt.step()
        torch.cuda.synchronize()
        return {k: v.detach().clone() for k, v in net.state_dict().items()}, float(opt.state_dict()['state'][0]['step'])
    a, sa = run(True)
    b, sb = run(False)
    assert sa == sb
    for k in a:
        assert torch.allclose(a[k], b[k], rtol=2e-4, atol=2e-6), k


def test_recording_with_side_streams_at_every_position_of_the_stream_pool():
    """`torch.cuda.Stream()` hands out one of 32 pooled HIP streams in turn, so the 33rd request anywhere in the process is the first stream
    again.  The streams of a recorded step (gradient lanes, the integrator's stream, the optimizer's stream) are made by
    functional.own_stream() and cannot coincide with a pooled one nor with each other: the recording is repeated with the pool advanced by one
    request each time, over more than a full turn (with pooled side streams one of these positions made the optimizer's stream wait for
    itself inside the capture and hipStreamEndCapture never returned)."""
    import numpy as np
    from oracle.detdata import det_fill
    from oracle.golden_configs import CONFIGS, make_batch
    from spatiotemporal_variable_separation_amd import functional as VF
    from spatiotemporal_variable_separation_amd.networks.factory import build_sep_net
    from spatiotemporal_variable_separation_amd.optim import Adam
    from spatiotemporal_variable_separation_amd.train import GraphedStep, enable_update_in_backward
    order = [torch.cuda.Stream().cuda_stream for _ in range(32)]
    pooled = set(order)
    assert torch.cuda.Stream().cuda_stream == order[0]                        # the pool has come round

    def advance_pool_to(k):
        nxt = (order.index(torch.cuda.Stream().cuda_stream) + 1) % 32
        for _ in range((k - nxt) % 32):
            torch.cuda.Stream()
    own = [VF.own_stream() for _ in range(3)]
    handles = [s.cuda_stream for s in own] + [s.cuda_stream for s in VF._OWN_STREAMS]
    assert len({s.cuda_stream for s in own}) == 3 and not (set(handles) & pooled)
    cfg = CONFIGS['mlp_mul']
    lam = cfg['lambdas']
    cond, target = make_batch(cfg)
    cond, target = cond.cuda(), target.cuda()
    np.random.seed(5)
    for turn in range(33):
        net = det_fill(build_sep_net(cfg), salt=cfg['salt']).cuda().train()
        opt = Adam(net.parameters(), lr=1e-3, betas=(0.9, 0.99))
        enable_update_in_backward(opt, net, force=True)
        advance_pool_to(turn % 32)
        g = GraphedStep(net, opt, cond, target, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], (lam['ae'], lam['s'], lam['t'], lam['pred']), warmup=1)
        g.step()
        used = {s.cuda_stream for s in VF._SIDE['lanes']} | ({opt._stream.cuda_stream} if opt._stream is not None else set())
        assert used and not (used & pooled), turn
    torch.cuda.synchronize()


def test_step_by_subsets_equals_one_step():
    """`step_subset()` over a partition of the parameters + `finish_step()` is `step()` (the data-parallel graph path updates
    one all-reduce bucket at a time while the next buckets are still on the wire)."""
    from spatiotemporal_variable_separation_amd.optim import Adam
    a, b = _params(7), _params(7)
    oa = Adam(a, lr=3e-4, betas=(0.9, 0.99))
    ob = Adam(b, lr=3e-4, betas=(0.9, 0.99))
    g = torch.Generator().manual_seed(3)
    for it in range(3):
        for x, y in zip(a, b):
            gr = (torch.rand(x.shape, generator=g) - 0.5).cuda()
            x.grad, y.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step_subset(b[3:])
        ob.step_subset(b[:1])
        ob.step_subset(b[1:3])
        ob.finish_step()
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert torch.equal(x, y)
        assert torch.equal(oa.state[x]['exp_avg'], ob.state[y]['exp_avg']) and torch.equal(oa.state[x]['exp_avg_sq'], ob.state[y]['exp_avg_sq'])
    b[0].grad = None
    with pytest.raises(Exception):
        ob.step_subset(b[:1])


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(1200, 2048, 256), (136, 264, 40), (512, 1000, 3328)])
def test_gemm_adam_equals_gemm_then_adam(dtype, shape):
    """vs_gemm_adam (the weight-gradient GEMM whose epilogue is the optimizer step) == the same GEMM storing its result followed by
    vs_adam_multi: bitwise the same parameters, moments and 16-bit copies over three steps, including ragged tiles."""
    import os
    from oracle.detdata import det_uniform
    from spatiotemporal_variable_separation_amd import ops, functional as VF
    from spatiotemporal_variable_separation_amd.optim import Adam
    M, N, K = shape                                 # parameter [M, N]; operands dz [K, M], h [K, N] in the S layout of a weight gradient
    p0 = (det_uniform((M, N), 3) - 0.5).cuda()
    pa, pb = torch.nn.Parameter(p0.clone()), torch.nn.Parameter(p0.clone())
    oa, ob = Adam([pa], lr=4e-4, betas=(0.9, 0.99)), Adam([pb], lr=4e-4, betas=(0.9, 0.99))
    with VF.precision('bf16' if dtype == torch.bfloat16 else 'fp16'):
        sa, sb = VF.shadow(pa, dtype), VF.shadow(pb, dtype)           # live operand copies: both paths must keep them current
        for step in range(3):
            dz = ((det_uniform((K, M), 10 + step) - 0.5) * 0.1).cuda().to(dtype)
            h = (det_uniform((K, N), 20 + step) - 0.5).cuda().to(dtype)
            # (a) fused
            oa.fuse_into_wgrad([pa])
            assert oa.can_fuse(pa, dz, h)
            oa.fused_update(pa, dz, 1, h, 1, M, N, K)
            oa.step()                                                  # nothing left to update; advances the step count
            oa.unfuse()
            # (b) the same kernel storing the gradient (128x128 ring tile, no split-K), then the optimizer's own launch
            os.environ['VS_GEMM_MID'] = '2'
            os.environ['VS_GEMM_BIG'] = '0'
            try:
                pb.grad = ops.gemm(dz, 1, h, 1, M, N, K)
            finally:
                del os.environ['VS_GEMM_MID'], os.environ['VS_GEMM_BIG']
            ob.step()
            torch.cuda.synchronize()
            assert torch.equal(pa, pb), (step, (pa - pb).abs().max().item())
            for name in ('exp_avg', 'exp_avg_sq'):
                assert torch.equal(oa.state[pa][name], ob.state[pb][name]), (step, name)
            assert torch.equal(VF.shadow(pa, dtype), VF.shadow(pb, dtype)) and torch.equal(VF.shadow(pa, dtype), pa.detach().to(dtype))
    assert float(oa.state_dict()['state'][0]['step']) == 3.0


def test_fused_update_in_the_recorded_step_equals_the_plain_step(name='mlp_mul'):
    """GraphedStep with the Linear weights' Adam steps fused into their weight-gradient GEMMs == the same recorded step with the
    optimizer's own launch for every parameter (bf16 mode; the gradients differ only in their split-K summation order)."""
    import os
    import numpy as np
    from oracle import cpu_ref
    from oracle.detdata import det_fill
    from oracle.golden_configs import CONFIGS, make_batch
    from spatiotemporal_variable_separation_amd import functional as VF
    from spatiotemporal_variable_separation_amd.networks.factory import build_sep_net
    from spatiotemporal_variable_separation_amd.optim import Adam
    from spatiotemporal_variable_separation_amd.train import GraphedStep, chain_weight_parameters
    cfg = dict(CONFIGS[name], B=8)
    lam = cfg['lambdas']
    cond, target = make_batch(cfg)
    cond, target = cond.cuda(), target.cuda()
    o_net = det_fill(cpu_ref.build_sep_net(cfg), salt=cfg['salt'])
    results = []
    for fused in (True, False):
        net = build_sep_net(cfg)
        net.load_state_dict(o_net.state_dict())
        net = net.cuda().train()
        with VF.precision('bf16'):
            opt = Adam(net.parameters(), lr=1e-3, betas=(0.9, 0.99))
            os.environ['VARSEP_FUSE_ADAM'] = '0'        # GraphedStep's own choice (weights >= 4 M elements) is off: chosen here
            try:
                if fused:
                    opt.fuse_into_wgrad(chain_weight_parameters(net))
                    assert len(opt._fused) >= 6
                np.random.seed(11)
                g = GraphedStep(net, opt, cond, target, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], (lam['ae'], lam['s'], lam['t'], lam['pred']),
                                warmup=2)
                losses = [g.step().item() for _ in range(4)]
            finally:
                del os.environ['VARSEP_FUSE_ADAM']
                opt.unfuse()
        torch.cuda.synchronize()
        results.append((losses, {k: v.clone() for k, v in net.state_dict().items()}, float(opt.state_dict()['state'][0]['step'])))
    (la, sa, ta), (lb, sb, tb) = results
    assert ta == tb == 4.0
    assert np.allclose(la, lb, rtol=2e-3), (la, lb)
    for k in sa:
        # Adam normalises: a near-zero gradient that differs in its last bits moves a weight by up to lr in either direction
        assert torch.allclose(sa[k], sb[k], rtol=2e-3, atol=2.5e-3), f'{k}: {(sa[k] - sb[k]).abs().max().item():.3e}'


def test_fused_update_leaves_the_input_gradients_of_its_layer_untouched():
    """A fused weight-gradient + Adam launch rewrites W and its 16-bit copy; the input gradient of the same layer reads that copy and must
    see W_t, not W_{t+1} (round-2 advice: with side streams off -- the eager loop, `bench.py --no_graph`, the ragged last batch of
    `train()` -- the update used to run first).  Every gradient that flows THROUGH a fused layer (biases below it, the integrator, the
    encoders) must therefore equal the unfused step's bit for bit: the same kernels compute them from the same operands."""
    from oracle import cpu_ref
    from oracle.detdata import det_fill
    from oracle.golden_configs import CONFIGS, make_batch
    from spatiotemporal_variable_separation_amd import functional as VF
    from spatiotemporal_variable_separation_amd.networks.factory import build_sep_net
    from spatiotemporal_variable_separation_amd.optim import Adam
    from spatiotemporal_variable_separation_amd.train import chain_weight_parameters, compute_losses
    cfg = dict(CONFIGS['mlp_mul'], B=8)
    lam = cfg['lambdas']
    cond, target = make_batch(cfg)
    cond, target = cond.cuda(), target.cuda()
    o_net = det_fill(cpu_ref.build_sep_net(cfg), salt=cfg['salt'])
    grads = []
    for fused in (True, False):
        net = build_sep_net(cfg)
        net.load_state_dict(o_net.state_dict())
        net = net.cuda().train()
        with VF.precision('bf16'):
            opt = Adam(net.parameters(), lr=1e-2, betas=(0.9, 0.99))      # a large step: W_{t+1} is far from W_t
            try:
                if fused:
                    opt.fuse_into_wgrad(chain_weight_parameters(net))
                    assert len(opt._fused) >= 6
                total = compute_losses(cond, target, net, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], False, lam['ae'], lam['s'], lam['t'],
                                       lam['pred'], t_random=4)[0]
                total.backward()
                torch.cuda.synchronize()
                grads.append({k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None})
                opt.step()
            finally:
                opt.unfuse()
    ga, gb = grads
    assert len(ga) < len(gb) and len(ga) > 0             # the fused weights have no stored gradient
    for k in ga:
        assert torch.equal(ga[k], gb[k]), f'{k}: differs by {(ga[k] - gb[k]).abs().max().item():.3e} (input gradient read an updated weight?)'


def test_recorded_step_with_torch_adam_reads_current_weights():
    """GraphedStep with torch.optim.Adam(capturable=True) in bf16 mode: torch's optimizer updates the fp32 masters only, so the recording
    must contain the casts to the 16-bit operand copies -- otherwise every replay trains on the weights frozen at capture time (round-2
    advice).  The loss sequence and the parameters must follow the eager loop with the same optimizer."""
    import numpy as np
    from oracle import cpu_ref
    from oracle.detdata import det_fill
    from oracle.golden_configs import CONFIGS, make_batch
    from spatiotemporal_variable_separation_amd import functional as VF
    from spatiotemporal_variable_separation_amd.networks.factory import build_sep_net
    from spatiotemporal_variable_separation_amd.train import GraphedStep, compute_losses
    cfg = dict(CONFIGS['mlp_mul'], B=8)
    lam = cfg['lambdas']
    cond, target = make_batch(cfg)
    cond, target = cond.cuda(), target.cuda()
    o_net = det_fill(cpu_ref.build_sep_net(cfg), salt=cfg['salt'])
    n_steps = 6
    out = []
    for graph in (True, False):
        net = build_sep_net(cfg)
        net.load_state_dict(o_net.state_dict())
        net = net.cuda().train()
        with VF.precision('bf16'):
            opt = torch.optim.Adam(net.parameters(), lr=5e-3, betas=(0.9, 0.99), capturable=True)
            np.random.seed(11)
            hi = cond.shape[1] + target.shape[1] + (0 if cfg['offset'] == 0 else 1)
            if graph:
                g = GraphedStep(net, opt, cond, target, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], (lam['ae'], lam['s'], lam['t'], lam['pred']),
                                warmup=2)
                losses = [g.step().item() for _ in range(n_steps)]
            else:
                np.random.randint(cfg['nt_cond'], hi)       # the recording itself consumes one draw of the t_random stream
                losses = []
                for _ in range(n_steps):
                    opt.zero_grad(set_to_none=True)
                    t_random = int(np.random.randint(cfg['nt_cond'], hi))
                    total = compute_losses(cond, target, net, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], False, lam['ae'], lam['s'], lam['t'],
                                           lam['pred'], t_random=t_random)[0]
                    total.backward()
                    opt.step()
                    losses.append(total.item())
        torch.cuda.synchronize()
        out.append((losses, {k: v.detach().clone() for k, v in net.state_dict().items()}))
    (lg, sg), (le, se) = out
    assert lg[0] != lg[-1]
    # stale weights would leave the recorded loss sequence flat / drifting: the eager loop moves the loss by >10 % over these steps
    assert abs(le[0] - le[-1]) > 0.05 * abs(le[0]), le
    assert np.allclose(lg, le, rtol=5e-3), (lg, le)
    for k in sg:
        assert torch.allclose(sg[k], se[k], rtol=5e-3, atol=3 * 5e-3), f'{k}: {(sg[k] - se[k]).abs().max().item():.3e}'


def test_loss_scaling_refuses_updates_issued_during_backward():
    """fp16 loss scaling needs every gradient before any update (finite check, 1 / scale): it refuses optimizers that update from backward."""
    from oracle.detdata import det_fill
    from oracle.golden_configs import CONFIGS
    from spatiotemporal_variable_separation_amd.networks.factory import build_sep_net
    from spatiotemporal_variable_separation_amd.optim import Adam
    from spatiotemporal_variable_separation_amd.train import LossScaler, enable_update_in_backward
    cfg = CONFIGS['mlp_mul']
    net = det_fill(build_sep_net(cfg), salt=cfg['salt']).cuda().train()
    opt = Adam(net.parameters(), lr=1e-3)
    scaler = LossScaler(torch.device('cuda'))
    enable_update_in_backward(opt, net, force=True, scaler=scaler)
    assert not opt._buckets                                 # declined
    enable_update_in_backward(opt, net, force=True)
    assert opt._buckets
    with pytest.raises(ValueError):
        scaler.step(opt)
