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

