"""Data-parallel step on the HIP path.  The GPU box has one device, so (1) two ranks share cuda:0 and exchange
gradients over gloo (the reducer stages device buckets through host memory for that backend) -- this runs the real
N>1 control flow of train.GraphedStep / parallel.GradAllReducer with the product kernels -- and (2) a world-size-1
RCCL group runs the exact collective calls of the N>1 path (all-reduce AVG on the side stream between two graph replays)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle.detdata import det_fill
from oracle.golden_configs import CONFIGS, make_batch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _build(cfg, salt):
    from spatiotemporal_variable_separation_amd.networks.factory import build_sep_net
    net = det_fill(build_sep_net(cfg), salt=salt).cuda()
    net.train()
    return net


def _run(net, cond, target, cfg, sync, graph, steps):
    from spatiotemporal_variable_separation_amd.train import GraphedStep, compute_losses
    lam = cfg['lambdas']
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, capturable=graph)
    np.random.seed(7)
    losses = []
    if graph:
        gs = GraphedStep(net, opt, cond, target, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'],
                         (lam['ae'], lam['s'], lam['t'], lam['pred']), warmup=1, grad_sync=sync)
        for _ in range(steps):
            losses.append(gs.step().item())
    else:
        for _ in range(steps + 2):                  # GraphedStep takes 1 warm-up step and draws once more for the capture
            t_random = int(np.random.randint(cfg['nt_cond'], cond.shape[1] + target.shape[1] + (cfg['offset'] != 0)))
            if _ == 1:
                continue
            if sync is not None:
                sync.zero_grad()
            else:
                opt.zero_grad()
            total = compute_losses(cond, target, net, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], False, lam['ae'],
                                   lam['s'], lam['t'], lam['pred'], t_random=t_random)[0]
            total.backward()
            if sync is not None:
                sync.all_reduce()
            opt.step()
            losses.append(total.item())
    torch.cuda.synchronize()
    return losses


def _worker(rank, world, port, backend, graph, out_dir, comm='fp32', direct=False):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    from spatiotemporal_variable_separation_amd.parallel import GradAllReducer, broadcast_module_state
    cfg = dict(CONFIGS['mlp_mul'], B=8)
    cond, target = make_batch(cfg)
    per = 8 // world
    shard = slice(rank * per, rank * per + per)
    net = _build(cfg, cfg['salt'] + rank)                      # ranks start different, rank 0's state wins
    broadcast_module_state(net)
    from spatiotemporal_variable_separation_amd.train import chain_weight_parameters
    sync = GradAllReducer(net.parameters(), bucket_bytes=16 << 10, force=(world == 1),
                          comm_dtype=torch.bfloat16 if comm == 'bf16' else torch.float32,
                          lowp_direct=chain_weight_parameters(net) if direct else None)
    _run(net, cond[shard].cuda().contiguous(), target[shard].cuda().contiguous(), cfg, sync, graph, 3)
    torch.save({k: v.detach().cpu() for k, v in net.state_dict().items()}, os.path.join(out_dir, f'rank{rank}.pt'))
    dist.destroy_process_group()


def _single(graph):
    cfg = dict(CONFIGS['mlp_mul'], B=8)
    cond, target = make_batch(cfg)
    net = _build(cfg, cfg['salt'])
    _run(net, cond.cuda(), target.cuda(), cfg, None, graph, 3)
    return {k: v.detach().cpu() for k, v in net.state_dict().items()}


@pytest.mark.parametrize('graph', [False, True])
def test_two_ranks_on_one_gpu_equal_single_process(tmp_path, graph):
    mp.spawn(_worker, args=(2, _free_port(), 'gloo', graph, str(tmp_path)), nprocs=2, join=True)
    ref = _single(graph)
    r0 = torch.load(os.path.join(tmp_path, 'rank0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'rank1.pt'))
    for k, v in ref.items():
        assert torch.equal(r0[k], r1[k]), f'replicas diverged at {k}'
        assert torch.allclose(r0[k], v, rtol=2e-4, atol=2e-6), \
            f'{k}: DDP step differs from the single-process step by {(r0[k] - v).abs().max().item():.3e} (max |v| {v.abs().max().item():.3e})'


def test_rccl_world1_graphed_step_equals_plain_graph(tmp_path):
    mp.spawn(_worker, args=(1, _free_port(), 'nccl', True, str(tmp_path)), nprocs=1, join=True)
    ref = _single(True)
    r0 = torch.load(os.path.join(tmp_path, 'rank0.pt'))
    for k, v in ref.items():
        # not bit-equal: bias gradients are float-atomic column sums (last-bit run-to-run noise that Adam amplifies)
        assert torch.allclose(r0[k], v, rtol=2e-4, atol=2e-6), f'{k}: RCCL world-1 reducer changed the step'


def test_two_ranks_bf16_gradient_wire_format(tmp_path):
    """comm_dtype=bf16: the replicas stay bit-identical to each other and close to the fp32-wire step (gradients rounded to bf16
    for the exchange: 4e-3 relative, which Adam turns into parameter differences of the order of lr * 1e-2)."""
    mp.spawn(_worker, args=(2, _free_port(), 'gloo', False, str(tmp_path), 'bf16'), nprocs=2, join=True)
    ref = _single(False)
    r0 = torch.load(os.path.join(tmp_path, 'rank0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'rank1.pt'))
    for k, v in ref.items():
        assert torch.equal(r0[k], r1[k]), f'replicas diverged at {k}'
        assert torch.allclose(r0[k], v, rtol=5e-2, atol=2e-4), f'{k}: bf16-wire DDP step far from the single-process step'


@pytest.mark.parametrize('world,backend', [(1, 'nccl'), (2, 'gloo')])
def test_graphed_step_with_directly_written_wire_gradients(tmp_path, world, backend):
    """bf16 wire + lowp_direct: the chains' weight-gradient GEMMs write the bf16 wire image themselves, the all-reduce averages it
    in place and Adam reads it.  Same rounding point as casting the fp32 bucket, so the step must be the one of the plain bf16-wire
    graph path (up to the float-atomic bias sums), and replicas must stay identical."""
    a, b = tmp_path / 'direct', tmp_path / 'cast'
    a.mkdir(), b.mkdir()
    mp.spawn(_worker, args=(world, _free_port(), backend, True, str(a), 'bf16', True), nprocs=world, join=True)
    mp.spawn(_worker, args=(world, _free_port(), backend, True, str(b), 'bf16', False), nprocs=world, join=True)
    d0, c0 = torch.load(os.path.join(a, 'rank0.pt')), torch.load(os.path.join(b, 'rank0.pt'))
    for k, v in c0.items():
        # two runs whose float-atomic bias sums differ in the last bit can round a gradient to the neighbouring bf16 value (0.4 %):
        # after three Adam steps (lr 1e-3) that is a few 1e-6 on a parameter; a wrong gradient would show as ~1e-3
        assert torch.allclose(d0[k], v, rtol=1e-3, atol=2e-5), f'{k}: direct wire gradients changed the step by {(d0[k] - v).abs().max().item():.3e}'
    if world == 2:
        d1 = torch.load(os.path.join(a, 'rank1.pt'))
        for k in d0:
            assert torch.equal(d0[k], d1[k]), f'replicas diverged at {k}'



# ---- convolution families (BatchNorm per replica, the decoder's grouped calls, folded / batched weight gradients) -----------------------
def _conv_worker(rank, world, port, name, graph, precision, out_dir, segmented=False):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from spatiotemporal_variable_separation_amd import functional as VF
    from spatiotemporal_variable_separation_amd.optim import Adam
    from spatiotemporal_variable_separation_amd.parallel import GradAllReducer, broadcast_module_state
    from spatiotemporal_variable_separation_amd.train import GraphedStep, compute_losses, conv_gradient_sinks
    cfg = dict(CONFIGS[name], B=8)
    lam, skipco = cfg['lambdas'], bool(cfg.get('skipco', False))
    cond, target = make_batch(cfg)
    per = cfg['B'] // world
    cond, target = cond[rank * per:(rank + 1) * per].cuda().contiguous(), target[rank * per:(rank + 1) * per].cuda().contiguous()
    net = _build(cfg, cfg['salt'] + rank)
    broadcast_module_state(net)
    sync = GradAllReducer(net.parameters(), bucket_bytes=256 << 10, early=list(net.decoder.parameters()) if segmented else None)
    opt = Adam(net.parameters(), lr=1e-3, betas=(0.9, 0.99))
    np.random.seed(7)
    hi = cond.shape[1] + target.shape[1] + (cfg['offset'] != 0)
    losses = []
    with VF.precision(precision):
        if graph:
            gs = GraphedStep(net, opt, cond, target, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], (lam['ae'], lam['s'], lam['t'], lam['pred']),
                             bool(cfg.get('average_tloss')), warmup=1, grad_sync=sync)
            assert gs._conv_sinks, 'the convolution gradients must go straight into the reducer\'s buckets'
            assert gs.segmented == segmented and (gs.graph2 is not None) == segmented
            for _ in range(3):
                losses.append(gs.step().item())
        else:
            sinks = conv_gradient_sinks(net, sync)
            VF.fold_repeated_gradients(True)
            np.random.randint(cfg['nt_cond'], hi)            # (the recorded variant draws once for its capture)
            for _ in range(3):
                t_random = int(np.random.randint(cfg['nt_cond'], hi))
                sync.zero_grad()
                VF.set_conv_grad_outputs(sinks)
                total = compute_losses(cond, target, net, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], skipco, lam['ae'], lam['s'], lam['t'],
                                       lam['pred'], average_tloss=bool(cfg.get('average_tloss')), t_random=t_random)[0]
                total.backward()
                VF.set_conv_grad_outputs(None)
                sync.all_reduce()
                opt.step()
                VF.flush_bn_call_counts()
                losses.append(total.item())
            VF.fold_repeated_gradients(False)
    torch.cuda.synchronize()
    torch.save({'state': {k: v.detach().cpu() for k, v in net.state_dict().items()}, 'losses': losses}, os.path.join(out_dir, f'rank{rank}.pt'))
    dist.destroy_process_group()


def _two_halves_reference(name, precision, world=2):
    """What `world` replicas with per-replica BatchNorm and averaged gradients compute, in ONE process: per step, the shards are run one
    after the other through the same network (each call normalises with its own batch statistics), the gradients of the 1/world-weighted
    losses accumulate, and the BatchNorm buffers kept are those of shard 0 (rank 0's replica)."""
    from spatiotemporal_variable_separation_amd import functional as VF
    from spatiotemporal_variable_separation_amd.optim import Adam
    from spatiotemporal_variable_separation_amd.train import compute_losses
    cfg = dict(CONFIGS[name], B=8)
    lam, skipco = cfg['lambdas'], bool(cfg.get('skipco', False))
    cond, target = make_batch(cfg)
    per = cfg['B'] // world
    net = _build(cfg, cfg['salt'])
    opt = Adam(net.parameters(), lr=1e-3, betas=(0.9, 0.99))
    np.random.seed(7)
    hi = cond.shape[1] + target.shape[1] + (cfg['offset'] != 0)
    np.random.randint(cfg['nt_cond'], hi)
    losses = []
    with VF.precision(precision):
        for _ in range(3):
            t_random = int(np.random.randint(cfg['nt_cond'], hi))
            opt.zero_grad()
            kept = None
            for r in range(world):
                c, t = cond[r * per:(r + 1) * per].cuda().contiguous(), target[r * per:(r + 1) * per].cuda().contiguous()
                total = compute_losses(c, t, net, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], skipco, lam['ae'], lam['s'], lam['t'], lam['pred'],
                                       average_tloss=bool(cfg.get('average_tloss')), t_random=t_random)[0]
                (total / world).backward()
                if r == 0:
                    losses.append(total.item())
                    kept = {k: v.detach().clone() for k, v in net.named_buffers()}
            with torch.no_grad():
                for k, v in net.named_buffers():
                    v.copy_(kept[k])
            opt.step()
    torch.cuda.synchronize()
    return {k: v.detach().cpu() for k, v in net.state_dict().items()}, losses


@pytest.mark.parametrize('graph', [False, True])
@pytest.mark.parametrize('name,precision', [('vgg32_tiny', 'fp32'), ('sst_skip', 'fp32'), ('sst_skip', 'bf16'), ('dcgan_tiny', 'bf16')])
def test_conv_family_two_ranks_equal_two_independent_halves(tmp_path, name, precision, graph):
    """TaxiBJ-style (VGG) and SST-style (skip decoder, ConvResnet integrator with fused blocks and batched weight gradients in bf16) nets
    under the reducer, eager with hooks and as graph A -> all-reduce -> graph B: the convolution / BatchNorm gradients accumulate
    straight into the bucket views (gradient folding and the integrator's batched weight gradients stay ON), replicas stay bit-identical,
    and rank 0 lands on the parameters AND BatchNorm buffers of the two-halves reference."""
    mp.spawn(_conv_worker, args=(2, _free_port(), name, graph, precision, str(tmp_path)), nprocs=2, join=True)
    ref, ref_losses = _two_halves_reference(name, precision)
    r0 = torch.load(os.path.join(tmp_path, 'rank0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'rank1.pt'))
    cfg = dict(CONFIGS[name], B=8)
    init = {k: v.detach().cpu() for k, v in _build(cfg, cfg['salt']).state_dict().items()}
    assert np.allclose(r0['losses'], ref_losses, rtol=1e-4 if precision == 'fp32' else 2e-2), (r0['losses'], ref_losses)
    # Adam's update is ~lr * sign(g) in the first steps: compare the UPDATES (three steps) per tensor in relative L2.  fp32: summation
    # order only (elements whose gradient is noise-level may flip sign: a few 1e-3 of a tensor's update norm).  bf16: a last-bit
    # difference in a folded sum can round one stored value the other way (step_util.lowp_noise_floor), looser.
    tol = 3e-2 if precision == 'fp32' else 0.6
    worst = 0.0
    for k, v in ref.items():
        a = r0['state'][k]
        if k.endswith('num_batches_tracked'):
            assert int(a) == int(v) == int(r1['state'][k]), k
            continue
        if 'running' in k:                           # per-replica buffers: rank 0's are shard 0's
            # (bf16: the parameters themselves are only held to 0.6 of their update below -- a last-bit difference in a folded sum can round a stored
            # value the other way and the BatchNorm stacks amplify it; the batch means the running estimates average follow: seen 1 run in ~5 past
            # 5e-2 / 5e-3 on one buffer inside the full suite, never alone)
            assert torch.allclose(a, v, rtol=1e-3 if precision == 'fp32' else 0.2, atol=1e-5 if precision == 'fp32' else 2e-2), k
            continue
        assert torch.equal(a, r1['state'][k]), f'replicas diverged at {k}'
        du, dv = (a - init[k]).double(), (v - init[k]).double()
        if dv.norm().item() == 0.0:                  # a conv bias in front of a BatchNorm: exactly-zero gradient, never moves
            assert du.norm().item() == 0.0, k
            continue
        e = ((du - dv).norm() / dv.norm()).item()
        worst = max(worst, e)
        assert e <= tol, f'{k}: the update of the data-parallel run differs from the two-halves reference by {e:.3e} of its norm'
    print(name, precision, 'graph' if graph else 'eager', 'worst update distance %.2e' % worst, 'losses', r0['losses'])


@pytest.mark.parametrize('name,precision', [('vgg32_tiny', 'fp32'), ('sst_skip', 'fp32'), ('sst_skip', 'bf16'), ('dcgan_skip_mul', 'fp32')])
def test_recorded_step_in_two_segments_equals_one_segment(tmp_path, name, precision):
    """The recorded data-parallel step of a conv family with the decoder's gradients in leading buckets of their own: backward recorded in TWO
    segments split at the decoder's inputs, the decoder buckets' all-reduce issued between the two replays (train.GraphedStep.segmented).
    Two ranks on one GPU over gloo: same losses as the one-segment recording, replicas bit-identical, parameters within the noise of two
    runs whose float-atomic sums differ in the last bit."""
    a, b = tmp_path / 'seg', tmp_path / 'one'
    a.mkdir(), b.mkdir()
    mp.spawn(_conv_worker, args=(2, _free_port(), name, True, precision, str(a), True), nprocs=2, join=True)
    mp.spawn(_conv_worker, args=(2, _free_port(), name, True, precision, str(b), False), nprocs=2, join=True)
    s0, s1 = torch.load(os.path.join(a, 'rank0.pt')), torch.load(os.path.join(a, 'rank1.pt'))
    o0 = torch.load(os.path.join(b, 'rank0.pt'))
    assert np.allclose(s0['losses'], o0['losses'], rtol=1e-5 if precision == 'fp32' else 2e-2), (s0['losses'], o0['losses'])
    cfg = dict(CONFIGS[name], B=8)
    init = {k: v.detach().cpu() for k, v in _build(cfg, cfg['salt']).state_dict().items()}
    worst = 0.0
    for k, v in o0['state'].items():
        x = s0['state'][k]
        if k.endswith('num_batches_tracked'):
            assert int(x) == int(v), k
            continue
        if 'running' not in k:
            assert torch.equal(x, s1['state'][k]), f'replicas diverged at {k}'
        du, dv = (x - init[k]).double(), (v - init[k]).double()
        if dv.norm().item() == 0.0:
            assert du.norm().item() == 0.0, k
            continue
        e = ((du - dv).norm() / dv.norm()).item()
        worst = max(worst, e)
        assert e <= (3e-2 if precision == 'fp32' else 0.6), f'{k}: update differs by {e:.3e} of its norm between the two recordings'
    print(name, precision, 'two segments vs one: worst update distance %.2e' % worst)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['vgg32_tiny', 'dcgan_tiny'])
def test_recorded_conv_step_with_weight_gradients_on_a_gradient_stream(name, monkeypatch):
    """VARSEP_CONV_WGRAD_SIDE=1 (optional route of train.GraphedStep: convolution / BatchNorm gradients accumulate into a local flat buffer,
    weight gradients run on a gradient stream and are joined before the optimizer) computes what the plain recorded step computes."""
    cfg = dict(CONFIGS[name])
    cond, target = make_batch(cfg)
    cond, target = cond.cuda(), target.cuda()
    runs = {}
    for side in ('0', '1'):
        monkeypatch.setenv('VARSEP_CONV_WGRAD_SIDE', side)
        net = _build(cfg, salt=3)
        runs[side] = (_run(net, cond, target, cfg, None, True, 3), {k: v.detach().float().cpu() for k, v in net.state_dict().items()})
    assert np.allclose(runs['0'][0], runs['1'][0], rtol=2e-4), (runs['0'][0], runs['1'][0])
    for k, v in runs['0'][1].items():
        if v.dtype.is_floating_point and v.numel() > 1:
            d = (runs['1'][1][k] - v).norm().item() / (v.norm().item() + 1e-12)
            assert d < 2e-3, (k, d)


@pytest.mark.gpu
def test_recorded_sst_step_with_the_integrator_on_a_side_stream(monkeypatch):
    """VARSEP_ROLLOUT_SIDE=1 (optional route of train.compute_losses: the convolutional integrator between E_t and the forecast decode on a side
    stream, autograd replaying the assignment in backward) computes what the serial recorded step computes."""
    cfg = dict(CONFIGS['sst_skip'])
    cond, target = make_batch(cfg)
    cond, target = cond.cuda(), target.cuda()
    runs = {}
    for side in ('0', '1'):
        monkeypatch.setenv('VARSEP_ROLLOUT_SIDE', side)
        net = _build(cfg, salt=5)
        runs[side] = (_run(net, cond, target, cfg, None, True, 3), {k: v.detach().float().cpu() for k, v in net.state_dict().items()})
    assert np.allclose(runs['0'][0], runs['1'][0], rtol=2e-4), (runs['0'][0], runs['1'][0])
    for k, v in runs['0'][1].items():
        if v.dtype.is_floating_point and v.numel() > 1:
            d = (runs['1'][1][k] - v).norm().item() / (v.norm().item() + 1e-12)
            assert d < 2e-3, (k, d)


# ---- sharded optimizer: reduce-scatter of the bf16 wire gradients, Adam on this rank's slice, all-gather of the operand copies ----------------
def _shard_worker(rank, world, port, backend, shard, out_dir, steps=3):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    from spatiotemporal_variable_separation_amd import functional as VF
    from spatiotemporal_variable_separation_amd.optim import Adam
    from spatiotemporal_variable_separation_amd.parallel import GradAllReducer, broadcast_module_state
    from spatiotemporal_variable_separation_amd.train import GraphedStep, chain_weight_parameters, rollout_weight_stacks
    VF.set_precision('bf16')
    try:
        cfg = dict(CONFIGS['mlp_mul'], B=8)
        cond, target = make_batch(cfg)
        per = 8 // world
        part = slice(rank * per, rank * per + per)
        net = _build(cfg, cfg['salt'] + rank)
        broadcast_module_state(net)
        direct = chain_weight_parameters(net)
        sync = GradAllReducer(net.parameters(), bucket_bytes=16 << 10, force=(world == 1), comm_dtype=torch.bfloat16, lowp_direct=direct,
                              shard_direct=shard, stacked=rollout_weight_stacks(net), shard_tail_dtype=torch.bfloat16)
        lam = cfg['lambdas']
        opt = Adam(net.parameters(), lr=1e-3)
        np.random.seed(7)
        gs = GraphedStep(net, opt, cond[part].cuda().contiguous(), target[part].cuda().contiguous(), cfg['nt_cond'], cfg['nt_pred'], cfg['offset'],
                         (lam['ae'], lam['s'], lam['t'], lam['pred']), warmup=1, grad_sync=sync)
        assert gs.sharded == bool(shard)
        losses = [gs.step().item() for _ in range(steps)]
        torch.cuda.synchronize()
        copies = {}
        for name, p in net.named_parameters():
            buf = VF.shadow_buffer_for_update(p)
            if buf is not None:
                copies[name] = buf.detach().float().cpu()
        if shard and world > 1:
            # before the masters are completed, a parameter cut by the slice boundary is current in the own slice only
            assert sync.masters_dirty
        sync.sync_masters(opt)
        torch.cuda.synchronize()
        torch.save({'state': {k: v.detach().cpu() for k, v in net.state_dict().items()}, 'copies': copies, 'losses': losses,
                    'moments': {n: opt.state[p]['exp_avg'].detach().cpu() for n, p in net.named_parameters() if p in opt.state}},
                   os.path.join(out_dir, f'rank{rank}.pt'))
    finally:
        VF.set_precision('fp32')
    dist.destroy_process_group()


@pytest.mark.parametrize('world,backend', [(1, 'nccl'), (2, 'gloo')])
def test_sharded_optimizer_equals_replicated_update(tmp_path, world, backend):
    """GradAllReducer(shard_direct=True): the recorded step with reduce-scatter + Adam on the rank's slice + all-gather of the 16-bit operand
    copies takes the step of the replicated bf16-wire update (same kernel, same arithmetic per element; sums of gradients seeded with 1 / N
    instead of averages -- exact for N = 2) BIT FOR BIT, every rank ends with the same operand copies, and sync_masters() completes the fp32 masters and
    moments everywhere."""
    a, b = tmp_path / 'shard', tmp_path / 'repl'
    a.mkdir(), b.mkdir()
    mp.spawn(_shard_worker, args=(world, _free_port(), backend, True, str(a)), nprocs=world, join=True)
    mp.spawn(_shard_worker, args=(world, _free_port(), backend, False, str(b)), nprocs=world, join=True)
    s0, r0 = torch.load(os.path.join(a, 'rank0.pt')), torch.load(os.path.join(b, 'rank0.pt'))
    # shard_tail_dtype=bf16 gives the small tail the replicated path's rounding: the two steps are then the same arithmetic, element by element
    # (measured at world 1: bit-identical, and so are two replicated runs).  The bound is the sibling test's: two runs whose float-atomic bias
    # sums differ in the last bit can round a gradient to the neighbouring bf16 value, a few 1e-6 on a parameter after three steps; a wrong
    # slice, a stale operand copy or a missed 1 / N shows as ~1e-3
    exact = 0
    for k, v in r0['state'].items():
        assert torch.allclose(s0['state'][k], v, rtol=1e-3, atol=2e-5), f'{k}: sharded update differs by {(s0["state"][k] - v).abs().max().item():.3e}'
        exact += int(torch.equal(s0['state'][k], v))
    for k, v in r0['moments'].items():
        assert torch.allclose(s0['moments'][k], v, rtol=5e-3, atol=2e-4), f'{k}: exp_avg differs after sync_masters'
    assert np.allclose(s0['losses'], r0['losses'], rtol=1e-3)
    print(f'sharded vs replicated (world {world}): {exact} of {len(r0["state"])} tensors bit-identical')
    assert len(s0['copies']) >= 6
    for k, v in s0['copies'].items():
        # the operand copy is the bf16 rounding of the completed master
        assert torch.equal(v, s0['state'][k].to(torch.bfloat16).float()), f'{k}: operand copy is not the rounded master'
    if world == 2:
        s1 = torch.load(os.path.join(a, 'rank1.pt'))
        for k in s0['state']:
            assert torch.equal(s0['state'][k], s1['state'][k]), f'replicas diverged at {k}'
        for k in s0['copies']:
            assert torch.equal(s0['copies'][k], s1['copies'][k]), f'operand copies diverged at {k}'
        assert s0['losses'] != s1['losses']          # (different halves of the batch)
