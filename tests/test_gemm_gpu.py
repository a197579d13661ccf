"""GPU parity of vs_gemm (through the C ABI) against an fp64 CPU contraction of the same operands.

Tolerances: compute=f32 runs the exact-fp32 MFMA (k-ordered fmaf chain) -> 2e-6 relative L2;
compute=bf16 multiplies bf16 operands exactly and accumulates in fp32 -> 2e-6 relative L2 against the fp64
product of the bf16-rounded operands (the rounding of the INPUTS is the caller's choice, not the kernel's).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(128, 128, 64), (100, 70, 50), (7, 5, 3), (33, 200, 20), (300, 1200, 1200), (256, 1200, 4104),
          (128, 96, 20480), (1000, 64, 257), (64, 4096, 1200)]


def _operand(rows, K, layout, dtype, salt):
    from oracle.detdata import det_uniform
    x = (det_uniform((rows, K), salt) - 0.5) * 2.0
    x = x.to(dtype)
    dev = x.cuda()
    return x.double(), (dev if layout == 0 else dev.t().contiguous())


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('la,lb', [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize('shape', SHAPES)
def test_gemm_layouts(dtype, la, lb, shape):
    from spatiotemporal_variable_separation_amd import ops
    M, N, K = shape
    a64, a = _operand(M, K, la, dtype, 3)
    b64, b = _operand(N, K, lb, dtype, 5)
    out = ops.gemm(a, la, b, lb, M, N, K)
    torch.cuda.synchronize()
    ref = a64 @ b64.t()
    err = ((out.cpu().double() - ref).norm() / ref.norm()).item()
    assert err < 2e-6, f'{dtype} layouts ({la},{lb}) shape {shape}: rel err {err:.3e}'


def test_gemm_integer_exact_asymmetric():
    """A = shifted identity-like, B asymmetric integers: any row/col swap or k permutation bug shows up exactly."""
    from spatiotemporal_variable_separation_amd import ops
    M, N, K = 96, 160, 72
    a = torch.zeros(M, K)
    for m in range(M):
        a[m, (m * 5 + 1) % K] = 1.0
        a[m, (m * 3) % K] += 2.0
    b = (torch.arange(N).view(N, 1) * 3 + torch.arange(K).view(1, K) * 7) % 13 - 6.0
    ref = a @ b.t()
    for dtype in (torch.float32, torch.bfloat16, torch.float16):
        for la in (0, 1):
            for lb in (0, 1):
                aa = a.to(dtype).cuda()
                bb = b.to(dtype).cuda()
                aa = aa if la == 0 else aa.t().contiguous()
                bb = bb if lb == 0 else bb.t().contiguous()
                out = ops.gemm(aa, la, bb, lb, M, N, K).cpu()
                assert torch.equal(out, ref), f'{dtype} ({la},{lb}) max diff {(out - ref).abs().max()}'


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_gemm_epilogue(dtype):
    from spatiotemporal_variable_separation_amd import ops
    from oracle.detdata import det_uniform
    M, N, K = 200, 144, 88
    a = ((det_uniform((M, K), 1) - 0.5)).to(dtype)
    b = ((det_uniform((N, K), 2) - 0.5)).to(dtype)
    bias = det_uniform((N,), 3) - 0.5
    mask = (det_uniform((M, N), 4) - 0.3).to(dtype)
    prev = det_uniform((M, N), 5)
    z = 0.5 * (a.double() @ b.double().t()) + bias.double()
    for act, fn in [('relu', torch.relu), ('leaky_relu', lambda t: torch.nn.functional.leaky_relu(t, 0.2)),
                    ('sigmoid', torch.sigmoid), ('none', lambda t: t), ('tanh', torch.tanh)]:
        out = ops.gemm(a.cuda(), 0, b.cuda(), 0, M, N, K, alpha=0.5, bias=bias.cuda(), act=act)
        ref = fn(z)
        assert ((out.cpu().double() - ref).norm() / ref.norm()).item() < 3e-6, act
    # mask (ReLU / LeakyReLU derivative from the stored output) + accumulate + bf16 output
    out = prev.clone().cuda()
    ops.gemm(a.cuda(), 0, b.cuda(), 0, M, N, K, out=out, mask=mask.cuda(), mask_act='leaky_relu', accumulate=True)
    md = mask.double()
    ref = prev.double() + (a.double() @ b.double().t()) * torch.where(md > 0, 1.0, 0.2)
    assert ((out.cpu().double() - ref).norm() / ref.norm()).item() < 3e-6
    out16 = ops.gemm(a.cuda(), 0, b.cuda(), 0, M, N, K, out_dtype=torch.bfloat16, mask=mask.cuda(), mask_act='relu')
    ref = (a.double() @ b.double().t()) * (md > 0)
    assert ((out16.cpu().double() - ref).norm() / ref.norm()).item() < 4e-3     # one bf16 rounding of the output


def test_colsum_cast_act():
    from spatiotemporal_variable_separation_amd import ops
    from oracle.detdata import det_uniform
    x = det_uniform((777, 130), 9) - 0.5
    for dt in (torch.float32, torch.bfloat16):
        xd = x.to(dt)
        s = ops.colsum(xd.cuda(), 777, 130).cpu()
        assert torch.allclose(s, xd.float().sum(0), rtol=1e-5, atol=1e-4)
    c = ops.cast(x.cuda(), torch.bfloat16).cpu()
    assert torch.equal(c, x.to(torch.bfloat16))
    y = ops.act_fwd(x.cuda(), 'sigmoid').cpu()
    assert torch.allclose(y, torch.sigmoid(x), rtol=1e-6, atol=1e-6)
    g = ops.act_bwd(x.cuda(), y.cuda(), 'sigmoid').cpu()
    assert torch.allclose(g, x * y * (1 - y), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
@pytest.mark.parametrize('dims', [(5, 8, 16, 3, 6), (128, 32, 512, 3, 25), (37, 20, 512, 1, 15), (16, 4, 8, 1, 2), (3, 5, 8, 2, 1),
                                  (20, 16, 128, 2, 4), (48, 32, 256, 4, 7), (16, 32, 512, 3, 2)])
def test_fused_rollout_equals_stepwise(precision, dims):
    """vs_mlp_rollout_{fwd,bwd} against the same recurrence run block by block through MLPChain (same kernels'
    rounding points): codes, residuals, input gradient and every weight gradient."""
    from spatiotemporal_variable_separation_amd import functional as VF
    from spatiotemporal_variable_separation_amd.networks.resnet import MLPResnet
    from oracle.detdata import det_uniform
    B, C, H, nb, n = dims
    torch.manual_seed(0)
    net = MLPResnet(C, nb, H).cuda()
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(0.5)
    x0 = (det_uniform((B, C), 7) - 0.5).cuda()
    g = (det_uniform((B, n, C), 8) - 0.5).cuda()
    with VF.precision(precision):
        xa = x0.clone().requires_grad_(True)
        codes_f, res_f = net.rollout(xa, n)
        (codes_f * g).sum().backward()
        grads_f = [None if p.grad is None else p.grad.clone() for p in net.parameters()]
        dxa = xa.grad.clone()
        net.zero_grad()
        xb = x0.clone().requires_grad_(True)
        x, codes, ress = xb, [xb], []
        for t in range(1, n):
            x, r = net(x)
            codes.append(x)
            ress.append(r)
        codes_s = torch.stack(codes, dim=1)
        (codes_s * g).sum().backward()
    from spatiotemporal_variable_separation_amd import ops
    assert ops.rollout_exchange_error(x0.device) == 0          # no bounded spin of the inter-workgroup exchange timed out
    tol = 2e-5 if precision == 'fp32' else 2e-3

    def rel(a, b):
        return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()
    assert rel(codes_f, codes_s) < tol
    for t in range(n - 1):
        for b in range(nb):
            assert rel(res_f[t][b], ress[t][b]) < tol * 5
    assert rel(dxa, xb.grad) < tol
    for gf, p in zip(grads_f, net.parameters()):
        if n == 1:
            assert gf is None and p.grad is None          # no integrator step, no weight gradient
            continue
        assert rel(gf, p.grad) < tol * 5, (gf.shape,)


@pytest.mark.parametrize('dims', [(128, 32, 512, 3, 25), (256, 32, 256, 2, 6), (128, 20, 128, 1, 9)])
def test_rollout_ring_on_one_xcd_equals_agent_scope_exchange(dims, monkeypatch):
    """The weight-stationary rollout with 8 / 16 slabs keeps every slab's ring of workgroups on one XCD and publishes its exchange granules with
    plain stores (they meet the consumers' sc1 loads in that XCD's L2; VS_ROLLOUT_XCD_LOCAL=0: agent-scope stores through the fabric).  Same
    arithmetic: codes, residuals and every gradient bit for bit, and no exchange time-out."""
    from spatiotemporal_variable_separation_amd import functional as VF, ops
    from spatiotemporal_variable_separation_amd.networks.resnet import MLPResnet
    from oracle.detdata import det_uniform
    B, C, H, nb, n = dims
    torch.manual_seed(0)
    net = MLPResnet(C, nb, H).cuda()
    x0 = (det_uniform((B, C), 7) - 0.5).cuda()
    g = (det_uniform((B, n, C), 8) - 0.5).cuda()
    out = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('VS_ROLLOUT_XCD_LOCAL', mode)
        with VF.precision('bf16'):
            net.zero_grad()
            xa = x0.clone().requires_grad_(True)
            codes, res = net.rollout(xa, n)
            (codes * g).sum().backward()
        torch.cuda.synchronize()
        assert ops.rollout_exchange_error(x0.device) == 0
        out[mode] = [codes.detach().clone(), xa.grad.clone()] + [r.detach().clone() for step in res for r in step] + [p.grad.clone() for p in net.parameters()]
    assert len(out['1']) == len(out['0'])
    for a, b in zip(out['1'], out['0']):
        if a.dim() == 1:                        # bias gradients: column sums finished with fp32 atomics (order varies from run to run)
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-4)
        else:
            assert torch.equal(a, b)


def test_rollout_launches_share_an_exchange_area_that_is_never_cleared():
    """The weight-stationary rollout never clears its exchange area: every launch tags its granules above the previous launch's (per-slab epoch
    words inside the workspace, one workspace per geometry).  Forty forward + backward launches of one geometry interleaved with launches of
    two other geometries: every repetition reproduces the first bit for bit and no wait times out."""
    from spatiotemporal_variable_separation_amd import functional as VF, ops
    from spatiotemporal_variable_separation_amd.networks.resnet import MLPResnet
    from oracle.detdata import det_uniform
    nets, data = {}, {}
    for dims in [(128, 32, 512, 3, 9), (16, 32, 512, 3, 2), (48, 32, 256, 2, 5)]:
        B, C, H, nb, n = dims
        torch.manual_seed(3)
        nets[dims] = MLPResnet(C, nb, H).cuda()
        data[dims] = ((det_uniform((B, C), 7) - 0.5).cuda(), (det_uniform((B, n, C), 8) - 0.5).cuda())

    def run(dims):
        x0, g = data[dims]
        with VF.precision('bf16'):
            xa = x0.clone().requires_grad_(True)
            codes, _ = nets[dims].rollout(xa, dims[4])
            (codes * g).sum().backward()
        nets[dims].zero_grad()
        return codes.detach().clone(), xa.grad.clone()
    first = {d: run(d) for d in nets}
    order = list(nets)
    for i in range(40):
        d = order[0] if i % 4 else order[1 + (i // 4) % 2]
        c, dx = run(d)
        assert torch.equal(c, first[d][0]) and torch.equal(dx, first[d][1]), (i, d)
    torch.cuda.synchronize()
    assert ops.rollout_exchange_error('cuda') == 0


def test_colsum_multi_shapes_and_dtypes():
    """All bias gradients of a chain in one launch: vector path (8-column units), ragged columns, odd row counts, views."""
    from spatiotemporal_variable_separation_amd import ops
    from oracle.detdata import det_uniform
    jobs, refs = [], []
    for i, (m, n, dt) in enumerate([(3328, 1200, torch.bfloat16), (257, 4096, torch.bfloat16), (300, 33, torch.float32),
                                    (1000, 520, torch.float32), (5, 7, torch.bfloat16), (777, 264, torch.bfloat16)]):
        x = (det_uniform((m, n), 20 + i) - 0.5).to(dt).cuda()
        jobs.append(x)
        refs.append(x.double().sum(0))
    wide = (det_uniform((64, 512), 31) - 0.5).to(torch.bfloat16).cuda()
    jobs.append(wide[:, 8:136])                                   # a column window of a wider matrix (ld > N)
    refs.append(wide[:, 8:136].double().sum(0))
    outs = ops.colsum_multi(jobs)
    for o, r, x in zip(outs, refs, jobs):
        assert o.shape == r.shape
        assert torch.allclose(o.double(), r, rtol=1e-5, atol=1e-3 * x.shape[0] ** 0.5 * 1e-2), (x.shape, (o.double() - r).abs().max().item())


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('dims', [(3, 512, 32, 3072, 1, 1), (3, 512, 512, 3072, 1, 1), (2, 32, 512, 96, 1, 1), (4, 100, 72, 264, 0, 0),
                                  (5, 64, 40, 128, 0, 1)])
def test_gemm_batched_equals_loop(dt, dims):
    """vs_gemm_batched == a loop of vs_gemm over the problems (same kernel, same split-K plan per problem class)."""
    from spatiotemporal_variable_separation_amd import ops
    from oracle.detdata import det_uniform
    batch, M, N, K, la, lb = dims
    a = (det_uniform((batch, M, K) if la == 0 else (batch, K, M), 41) - 0.5).to(dt).cuda()
    b = (det_uniform((batch, N, K) if lb == 0 else (batch, K, N), 42) - 0.5).to(dt).cuda()
    out = ops.gemm_batched(a, la, b, lb, M, N, K)
    for i in range(batch):
        ai = a[i].float() if la == 0 else a[i].float().t()
        bi = b[i].float() if lb == 0 else b[i].float().t()
        ref = ai.double() @ bi.double().t()
        tol = 1e-5 if dt == torch.float32 else 1e-5         # inputs are exactly representable: only accumulation order differs
        assert torch.allclose(out[i].double(), ref, rtol=1e-4, atol=tol * K ** 0.5), (i, (out[i].double() - ref).abs().max().item())


# ---- LDS-DMA staged 128x128 tile (vs_gemm_glds.h) -------------------------------------------------------------------------
# R x R bf16 problems with >= 1024 tiles of 128x128 take it by default (two LDS buffers); row tails, K tails (K % 64 != 0)
# and the bias/activation epilogue included
@pytest.mark.parametrize('shape', [(4096, 4096, 520), (4100, 4100, 72), (8192, 2048, 1208)])
def test_gemm_lds_dma_tile_default_path(shape):
    from spatiotemporal_variable_separation_amd import ops
    M, N, K = shape
    a64, a = _operand(M, K, 0, torch.bfloat16, 11)
    b64, b = _operand(N, K, 0, torch.bfloat16, 13)
    bias = torch.linspace(-1, 1, N)
    out = ops.gemm(a, 0, b, 0, M, N, K, bias=bias.cuda(), act='relu')
    ref = torch.relu(a64 @ b64.t() + bias.double())
    err = ((out.cpu().double() - ref).norm() / ref.norm()).item()
    assert err < 2e-6, f'shape {shape}: rel err {err:.3e}'


_FORCED_TILE_SCRIPT = r'''
import sys, torch
sys.path.insert(0, sys.argv[1])
from spatiotemporal_variable_separation_amd import ops
from oracle.detdata import det_uniform
worst = 0.0
for (M, N, K) in [(256, 384, 256), (300, 200, 200), (1200, 1208, 2688), (128, 136, 72), (264, 4096, 1200)]:
    for la in (0, 1):
        for lb in (0, 1):
            a = ((det_uniform((M, K), 3) - 0.5) * 2).bfloat16()
            b = ((det_uniform((N, K), 5) - 0.5) * 2).bfloat16()
            ad = a.cuda() if la == 0 else a.cuda().t().contiguous()
            bd = b.cuda() if lb == 0 else b.cuda().t().contiguous()
            out = ops.gemm(ad, la, bd, lb, M, N, K)
            ref = a.double() @ b.double().t()
            worst = max(worst, ((out.cpu().double() - ref).norm() / ref.norm()).item())
print('WORST %.3e' % worst)
'''


@pytest.mark.parametrize('stages', ['1', '2'])
def test_gemm_lds_dma_tile_all_layouts_forced(stages):
    """Every operand layout through the LDS-DMA tile (S operands use the permuted [k][rows] image and transposing reads); the
    switches are read once per process, hence the child process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VS_GEMM_GLDS='2', VS_GEMM_TILE='128x128', VS_GEMM_GLDS_STAGES=stages)
    r = subprocess.run([sys.executable, '-c', _FORCED_TILE_SCRIPT, root], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    worst = float(r.stdout.strip().split('WORST')[-1])
    assert worst < 2e-6, r.stdout


# ---- the 256x256 LDS-DMA tile (vs_gemm_big.h): taken for 16-bit problems whose 256-wide tiles (x split-K) fit one round of CUs ----
BIG_SHAPES = [(3328, 4096, 1200), (3328, 1200, 4096), (4096, 1200, 3328), (1200, 1200, 3328), (3328, 1200, 1200),
              (1000, 1016, 520), (520, 2048, 776)]


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('la,lb', [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize('shape', BIG_SHAPES)
def test_gemm_big_tile_all_layouts(dtype, la, lb, shape):
    """The decoder-sized problems of the WaveEq step (and two ragged ones): every operand layout, row / column / K tails of the
    256x256 tile with its 4-deep K ring, fp32 and 16-bit outputs, against fp64 on the same rounded operands."""
    import os
    from spatiotemporal_variable_separation_amd import ops
    M, N, K = shape
    a64, a = _operand(M, K, la, dtype, 3)
    b64, b = _operand(N, K, lb, dtype, 5)
    from oracle.detdata import det_uniform
    bias = ((det_uniform((N,), 9) - 0.5) * 0.5).cuda()
    os.environ['VS_GEMM_BIG'] = '2'                  # take the 256x256 tile whatever the plan would say (it is read per call)
    os.environ['VS_GEMM_P8'] = '0'                   # (the staggered tile has tests of its own below)
    try:
        out = ops.gemm(a, la, b, lb, M, N, K, bias=bias, act='relu')
        out16 = ops.gemm(a, la, b, lb, M, N, K, out_dtype=dtype)
    finally:
        del os.environ['VS_GEMM_BIG']
        del os.environ['VS_GEMM_P8']
    torch.cuda.synchronize()
    ref = a64 @ b64.t()
    refa = torch.relu(ref + bias.cpu().double())
    err = ((out.cpu().double() - refa).norm() / refa.norm()).item()
    assert err < 2e-6, f'{dtype} layouts ({la},{lb}) shape {shape}: rel err {err:.3e}'
    err16 = ((out16.cpu().double() - ref).norm() / ref.norm()).item()
    assert err16 < (6e-3 if dtype == torch.bfloat16 else 8e-4), f'{dtype} 16-bit output ({la},{lb}) {shape}: {err16:.3e}'


MID_SHAPES = [(3328, 1200, 1200), (256, 1200, 20480), (1200, 1200, 3328), (136, 264, 40), (520, 1000, 776), (128, 128, 32)]


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('la,lb', [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize('shape', MID_SHAPES)
@pytest.mark.parametrize('mode', ['2', '1'])
def test_gemm_mid_tile_all_layouts(dtype, la, lb, shape, mode):
    """The 128x128 ring tile (vs_gemm_mid.h): every operand layout, row / column / K tails, one K tile only, split-K as planned
    (mode 1: the plan decides and may fall through to the other kernels) and forced (mode 2), fp32 and 16-bit outputs, against
    fp64 on the same rounded operands."""
    import os
    from spatiotemporal_variable_separation_amd import ops
    M, N, K = shape
    a64, a = _operand(M, K, la, dtype, 13)
    b64, b = _operand(N, K, lb, dtype, 15)
    from oracle.detdata import det_uniform
    bias = ((det_uniform((N,), 19) - 0.5) * 0.5).cuda()
    os.environ['VS_GEMM_MID'] = mode
    os.environ['VS_GEMM_BIG'] = '0'
    os.environ['VS_GEMM_P8'] = '0'
    try:
        out = ops.gemm(a, la, b, lb, M, N, K, bias=bias, act='relu')
        out16 = ops.gemm(a, la, b, lb, M, N, K, out_dtype=dtype)
    finally:
        del os.environ['VS_GEMM_MID']
        del os.environ['VS_GEMM_BIG']
        del os.environ['VS_GEMM_P8']
    torch.cuda.synchronize()
    ref = a64 @ b64.t()
    refa = torch.relu(ref + bias.cpu().double())
    err = ((out.cpu().double() - refa).norm() / refa.norm()).item()
    assert err < 2e-6, f'{dtype} layouts ({la},{lb}) shape {shape}: rel err {err:.3e}'
    err16 = ((out16.cpu().double() - ref).norm() / ref.norm()).item()
    assert err16 < (6e-3 if dtype == torch.bfloat16 else 8e-4), f'{dtype} 16-bit output ({la},{lb}) {shape}: {err16:.3e}'


SPLITK_SHAPES = [(256, 1200, 20480), (256, 1200, 1200), (256, 32, 1200), (1200, 1200, 3328), (64, 1200, 3328), (128, 96, 20480), (3328, 32, 1200),
                 (100, 70, 4104)]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('la,lb', [(0, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize('shape', SPLITK_SHAPES)
def test_splitk_finished_in_launch_is_bit_equal(dtype, la, lb, shape):
    """Split-K finished by the last workgroup of a tile (arrival counters, slabs added in split order) == the slab-reduce launch, bit for bit,
    with the whole epilogue (bias, activation, 16-bit output), and the counters come back to zero (40 launches back to back)."""
    import os
    from spatiotemporal_variable_separation_amd import ops
    from oracle.detdata import det_uniform
    M, N, K = shape
    _, a = _operand(M, K, la, dtype, 23)
    _, b = _operand(N, K, lb, dtype, 29)
    bias = ((det_uniform((N,), 31) - 0.5) * 0.5).cuda()
    os.environ['VS_GEMM_SPLITK_FUSED'] = '0'
    try:
        ref = ops.gemm(a, la, b, lb, M, N, K, bias=bias, act='leaky_relu')
        ref16 = ops.gemm(a, la, b, lb, M, N, K, out_dtype=torch.bfloat16)
    finally:
        del os.environ['VS_GEMM_SPLITK_FUSED']
    for mode in ('1', '2'):                    # 1: the default (small tiles x few splits only), 2: every split launch
        os.environ['VS_GEMM_SPLITK_FUSED'] = mode
        try:
            for _ in range(10):
                out = ops.gemm(a, la, b, lb, M, N, K, bias=bias, act='leaky_relu')
                out16 = ops.gemm(a, la, b, lb, M, N, K, out_dtype=torch.bfloat16)
        finally:
            del os.environ['VS_GEMM_SPLITK_FUSED']
        torch.cuda.synchronize()
        assert torch.equal(out, ref) and torch.equal(out16, ref16), mode


def test_splitk_finished_in_launch_batched_and_concurrent():
    """Batched split launches (the integrator's weight gradients) and two split launches in flight on two streams at once."""
    import os
    from spatiotemporal_variable_separation_amd import ops
    nb, M, N, K = 6, 512, 64, 2304
    _, a = _operand(nb * M, K, 0, torch.bfloat16, 37)
    _, b = _operand(nb * N, K, 0, torch.bfloat16, 41)
    a3, b3 = a.view(nb, M, K), b.view(nb, N, K)
    os.environ['VS_GEMM_SPLITK_FUSED'] = '0'
    try:
        ref = ops.gemm_batched(a3, 0, b3, 0, M, N, K)
    finally:
        del os.environ['VS_GEMM_SPLITK_FUSED']
    os.environ['VS_GEMM_SPLITK_FUSED'] = '2'
    try:
        out = ops.gemm_batched(a3, 0, b3, 0, M, N, K)
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        _, x = _operand(256, 20480, 0, torch.bfloat16, 43)
        _, w = _operand(1200, 20480, 0, torch.bfloat16, 47)
        one = ops.gemm(x, 0, w, 0, 256, 1200, 20480)
        torch.cuda.synchronize()
        outs = []
        for i in range(8):
            with torch.cuda.stream(s1 if i % 2 == 0 else s2):
                outs.append(ops.gemm(x, 0, w, 0, 256, 1200, 20480))
        torch.cuda.synchronize()
        assert all(torch.equal(o, one) for o in outs)
    finally:
        del os.environ['VS_GEMM_SPLITK_FUSED']


# ---- the staggered 256 x 256 / 256 x 128 tile (vs_gemm_p8.h): two wave groups half a phase apart, two 64-deep LDS-DMA buffers, counted waits ----
P8_SHAPES = [(3328, 4096, 1200), (3328, 1200, 4096), (4096, 1200, 3328), (3328, 1200, 1200), (256, 256, 64), (264, 520, 136),
             (1000, 1016, 520), (520, 2048, 776)]


class _Env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        import os
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        import os
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('la,lb,ni', [(0, 0, '2'), (0, 1, '2'), (1, 0, '2'), (1, 1, '2'), (0, 0, '1'), (1, 0, '1'), (0, 1, '1'), (1, 1, '1'),
                                      (0, 0, '1m2'), (0, 1, '1m2'), (1, 0, '1m2'), (1, 1, '1m2')])
@pytest.mark.parametrize('shape', P8_SHAPES)
def test_gemm_p8_tile_all_layouts(dtype, la, lb, ni, shape):
    """Every operand layout and the three tile shapes of the staggered tile (256 x 256, 256 x 128, and '1m2' = 128 x 128 with two workgroups per
    CU), forced (VS_GEMM_P8=2): one K tile only, row / column / K tails (the
    partial K tile reads zeros), the straight-line epilogues (bias + relu to fp32, 16-bit output, leaky mask) and the general one (sigmoid;
    accumulate), against fp64 on the same rounded operands."""
    from spatiotemporal_variable_separation_amd import ops
    from oracle.detdata import det_uniform
    M, N, K = shape
    a64, a = _operand(M, K, la, dtype, 23)
    b64, b = _operand(N, K, lb, dtype, 25)
    bias = ((det_uniform((N,), 29) - 0.5) * 0.5).cuda()
    mask = (det_uniform((M, N), 31) - 0.3).to(dtype)
    prev = det_uniform((M, N), 33)
    acc = prev.clone().cuda()
    with _Env(VS_GEMM_P8='2', VS_GEMM_P8_NI=ni[0], VS_GEMM_P8_MI='2' if ni.endswith('m2') else '4'):
        out = ops.gemm(a, la, b, lb, M, N, K, bias=bias, act='relu')
        out16 = ops.gemm(a, la, b, lb, M, N, K, out_dtype=dtype)
        outm = ops.gemm(a, la, b, lb, M, N, K, alpha=0.5, mask=mask.cuda(), mask_act='leaky_relu')
        outs = ops.gemm(a, la, b, lb, M, N, K, alpha=0.05, bias=bias, act='sigmoid')
        ops.gemm(a, la, b, lb, M, N, K, out=acc, accumulate=True)
        # the stores of a training step (straight-line code on tiles inside the matrix): Linear forward = bias + relu to 16 bits; input gradient =
        # 16-bit result under the 16-bit relu mask of the layer below, with every kind of "not > 0" planted in the first rows
        outf = ops.gemm(a, la, b, lb, M, N, K, bias=bias, act='relu', out_dtype=dtype)
        mask16 = mask.clone()
        tiny = torch.finfo(dtype).smallest_normal / 4                      # a denormal: > 0
        special = torch.tensor([0.0, -0.0, tiny, -tiny, float('inf'), -float('inf'), float('nan'), 1.0], dtype=torch.float32).to(dtype)
        mask16[:4, :special.numel() * 8] = special.repeat(8)
        outg = ops.gemm(a, la, b, lb, M, N, K, out_dtype=dtype, mask=mask16.cuda(), mask_act='relu')
    torch.cuda.synchronize()
    ref = a64 @ b64.t()
    keep = mask16.float() > 0                                                # (NaN > 0 is False, a denormal is > 0)
    assert keep[0, :8].tolist() == [False, False, True, False, True, False, False, True]
    assert torch.equal(outg.cpu().float(), torch.where(keep, out16.cpu().float(), torch.zeros(()))), f'{dtype} ({la},{lb}) ni {ni} {shape}: 16-bit relu mask'
    assert torch.equal(outf.cpu().float(), out.cpu().to(dtype).float()), f'{dtype} ({la},{lb}) ni {ni} {shape}: bias + relu to 16 bits'

    def rel(x, r):
        return ((x.cpu().double() - r).norm() / r.norm()).item()
    assert rel(out, torch.relu(ref + bias.cpu().double())) < 2e-6, f'{dtype} ({la},{lb}) ni {ni} {shape}: bias + relu'
    assert rel(out16, ref) < (6e-3 if dtype == torch.bfloat16 else 8e-4), f'{dtype} ({la},{lb}) ni {ni} {shape}: 16-bit output'
    assert rel(outm, 0.5 * ref * torch.where(mask.double() > 0, 1.0, 0.2)) < 2e-6, f'{dtype} ({la},{lb}) ni {ni} {shape}: leaky mask'
    assert rel(outs, torch.sigmoid(0.05 * ref + bias.cpu().double())) < 2e-6, f'{dtype} ({la},{lb}) ni {ni} {shape}: sigmoid'
    assert rel(acc, prev.double() + ref) < 2e-6, f'{dtype} ({la},{lb}) ni {ni} {shape}: accumulate'


@pytest.mark.parametrize('ni', ['2', '1', '1m2'])
def test_gemm_p8_integer_exact_and_repeatable(ni):
    """Small-integer operands (every product and sum exact in fp32): the staggered tile must reproduce the fp64 contraction BIT FOR BIT, on
    every one of 40 back-to-back launches beside a bandwidth-hungry kernel on another stream (a fragment read that overtakes its LDS-DMA or
    a request that overtakes the last read of its image shows up as rare wrong tiles, not as a tolerance miss)."""
    from spatiotemporal_variable_separation_amd import ops
    g = torch.Generator().manual_seed(5)
    for (M, N, K, la, lb) in [(1000, 1016, 1224, 0, 0), (776, 520, 3336, 1, 0)] + [(1016, 776, 1224, 0, 1), (520, 1000, 2056, 1, 1)]:
        a = torch.randint(-2, 3, (M, K), generator=g).float()
        b = torch.randint(-2, 3, (N, K), generator=g).float()
        ref = a @ b.t()
        aa = a.to(torch.bfloat16).cuda()
        bb = b.to(torch.bfloat16).cuda()
        aa = aa if la == 0 else aa.t().contiguous()
        bb = bb if lb == 0 else bb.t().contiguous()
        noise = torch.empty(64 << 20, device='cuda')
        side = torch.cuda.Stream()
        with _Env(VS_GEMM_P8='2', VS_GEMM_P8_NI=ni[0], VS_GEMM_P8_MI='2' if ni.endswith('m2') else '4'):
            for it in range(40):
                if it % 3 == 0:
                    with torch.cuda.stream(side):
                        noise.add_(1.0)
                out = ops.gemm(aa, la, bb, lb, M, N, K)
                assert torch.equal(out.cpu(), ref), f'ni {ni} ({la},{lb}) {M}x{N}x{K} launch {it}: max diff {(out.cpu() - ref).abs().max()}'
        torch.cuda.synchronize()


def test_gemm_p8_default_plan_takes_the_decoder_layer():
    """Without any switch the last decoder layer of the WaveEq model (3328 x 4096 x 1200) runs on the staggered tile: same numbers as with the
    tile forced, and as the fp64 contraction."""
    from spatiotemporal_variable_separation_amd import ops
    M, N, K = 3328, 4096, 1200
    a64, a = _operand(M, K, 0, torch.bfloat16, 43)
    b64, b = _operand(N, K, 0, torch.bfloat16, 45)
    out = ops.gemm(a, 0, b, 0, M, N, K)
    with _Env(VS_GEMM_P8='2'):
        forced = ops.gemm(a, 0, b, 0, M, N, K)
    with _Env(VS_GEMM_P8='0'):
        old = ops.gemm(a, 0, b, 0, M, N, K)
    ref = a64 @ b64.t()
    assert torch.equal(out, forced)
    assert ((out.cpu().double() - ref).norm() / ref.norm()).item() < 2e-6
    assert ((old.cpu().double() - ref).norm() / ref.norm()).item() < 2e-6
