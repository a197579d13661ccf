"""Data-parallel path on CPU (gloo, world_size 2): flat-bucket gradient averaging reproduces the single-process step.

The product kernels need a GPU, so the replicas here run the CPU oracle network; what is under test is the
host-side data-parallel machinery of `parallel.py` (bucket layout, hook-driven launches, averaging, broadcast)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cpu_ref
from oracle.detdata import det_fill
from oracle.golden_configs import CONFIGS, make_batch


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _losses(net, cond, target, cfg, t_random):
    lam = cfg['lambdas']
    return cpu_ref.training_losses(cond, target, net, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], False, lam['ae'],
                                   lam['s'], lam['t'], lam['pred'], t_random=t_random)[0]


def _worker(rank, world, port, overlap, bucket_bytes, out_dir, early=False):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from spatiotemporal_variable_separation_amd.parallel import GradAllReducer, broadcast_module_state
    torch.set_num_threads(1)
    cfg = dict(CONFIGS['mlp_mul'], B=8)
    cond, target = make_batch(cfg)
    net = det_fill(cpu_ref.build_sep_net(cfg), salt=cfg['salt'] + rank)       # ranks start DIFFERENT ...
    broadcast_module_state(net)                                               # ... and are made equal to rank 0
    sync = GradAllReducer(net.parameters(), bucket_bytes=bucket_bytes, overlap=overlap, early=list(net.decoder.parameters()) if early else None)
    if early:
        # the decoder's gradients (complete first in backward) lead, in buckets of their own
        dec = {id(p) for p in net.decoder.parameters()}
        assert sync.early_buckets and sync.early_buckets == list(range(len(sync.early_buckets)))
        for bi, (_, plist) in enumerate(sync.buckets):
            assert all((id(p) in dec) == (bi in sync.early_buckets) for p in plist)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    shard = slice(rank * 4, rank * 4 + 4)
    for step in range(2):
        sync.zero_grad()
        _losses(net, cond[shard], target[shard], cfg, 5 + step).backward()
        sync.all_reduce()
        opt.step()
    torch.save({k: v.clone() for k, v in net.state_dict().items()}, os.path.join(out_dir, f'rank{rank}.pt'))
    torch.save([b[0].numel() for b in sync.buckets], os.path.join(out_dir, f'buckets{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.parametrize('overlap,bucket_bytes,early', [(True, 16 << 10, False), (False, 64 << 20, False), (True, 16 << 10, True)])
def test_two_replicas_equal_single_process(tmp_path, overlap, bucket_bytes, early):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, overlap, bucket_bytes, str(tmp_path), early), nprocs=2, join=True)
    # single process on the concatenated batch (mean losses => averaged shard gradients are the same gradient)
    cfg = dict(CONFIGS['mlp_mul'], B=8)
    cond, target = make_batch(cfg)
    net = det_fill(cpu_ref.build_sep_net(cfg), salt=cfg['salt'])
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    for step in range(2):
        opt.zero_grad()
        _losses(net, cond, target, cfg, 5 + step).backward()
        opt.step()
    r0 = torch.load(os.path.join(tmp_path, 'rank0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'rank1.pt'))
    for k, v in net.state_dict().items():
        assert torch.equal(r0[k], r1[k]), f'replicas diverged at {k}'
        assert torch.allclose(r0[k], v, rtol=2e-4, atol=2e-6), f'{k}: DDP step differs from the single-process step'
    nb = torch.load(os.path.join(tmp_path, 'buckets0.pt'))
    assert (len(nb) > 1) == (bucket_bytes < (1 << 20))


# ---- sharded optimizer (parallel.GradAllReducer(shard_direct=True)): the host-side machinery on CPU tensors ----------------------------------
def _adam(p, g, m, v, t, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam's arithmetic on flat views, in place."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    denom = (v.sqrt() / (1 - b2 ** t) ** 0.5).add_(eps)
    p.addcdiv_(m, denom, value=-lr / (1 - b1 ** t))


class _Opt:
    def __init__(self, state):
        self.state = state


def _shard_worker(rank, world, port, shard, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from spatiotemporal_variable_separation_amd.parallel import GradAllReducer, broadcast_module_state
    torch.set_num_threads(1)
    cfg = dict(CONFIGS['mlp_mul'], B=8)
    cond, target = make_batch(cfg)
    net = det_fill(cpu_ref.build_sep_net(cfg), salt=cfg['salt'] + rank)
    broadcast_module_state(net)
    direct = [p for mod in (net.Es, net.Et, net.decoder) for p in mod.parameters() if p.dim() == 2]
    sync = GradAllReducer(net.parameters(), bucket_bytes=16 << 10, overlap=False, comm_dtype=torch.bfloat16, lowp_direct=direct, shard_direct=shard)
    sync.direct_lowp = True
    assert sync.shard == shard and len(sync.buckets) > 1
    is_direct = {id(p) for p in direct}
    copies = {}
    if shard:
        def register(p, view):
            view.copy_(p.detach())
            copies[id(p)] = view
        sync.adopt_operand_copies(torch.bfloat16, register)
        # layout: the head of every bucket splits into `world` slices of whole 128-byte lines; the slices of all ranks tile it
        for bi, (head, heads) in sync._head.items():
            assert head % (world * 64) == 0 and sync._tail[sync.buckets[bi][0].data_ptr()] == (head if sync.tail_params(bi) else sync.buckets[bi][0].numel())
            lo, hi = sync._slice(bi)
            assert (hi - lo) * world == head
            assert sum(b - a for _, a, b in sync.shard_ranges(bi)) <= hi - lo
    else:
        for p in direct:
            copies[id(p)] = p.detach().to(torch.bfloat16)
    state = {p: {'exp_avg': torch.zeros_like(p), 'exp_avg_sq': torch.zeros_like(p)} for p in net.parameters()}
    part = slice(rank * 4, rank * 4 + 4)
    seed = torch.tensor(1.0 / world if shard else 1.0)
    for step in range(2):
        sync.zero_buffers()
        for p in direct:
            p.grad.zero_()                                   # (on the GPU the GEMM epilogue overwrites the wire image)
        _losses(net, cond[part], target[part], cfg, 5 + step).backward(seed)
        for p in direct:
            sync.lowp_views[id(p)].copy_(p.grad)             # the weight-gradient GEMM's rounding into the wire buffer
        if shard:
            sync.reduce_all()
        else:
            # the replicated reference, by hand: bf16 images of the chains' weight gradients averaged (sum of the bf16 values in fp32,
            # one rounding), everything else averaged in fp32
            for p in net.parameters():
                if id(p) in is_direct:
                    host = sync.lowp_views[id(p)].float()
                    dist.all_reduce(host, op=dist.ReduceOp.SUM)
                    sync.lowp_views[id(p)].copy_((host / world).to(torch.bfloat16))
                else:
                    dist.all_reduce(p.grad, op=dist.ReduceOp.SUM)
                    p.grad.div_(world)
        with torch.no_grad():
            for bi, (_, plist) in enumerate(sync.buckets):
                if shard:
                    for p, lo, hi in sync.shard_ranges(bi):
                        st = state[p]
                        _adam(p.data.view(-1)[lo:hi], sync.lowp_views[id(p)].view(-1)[lo:hi].float(), st['exp_avg'].view(-1)[lo:hi],
                              st['exp_avg_sq'].view(-1)[lo:hi], step + 1)
                        copies[id(p)].view(-1)[lo:hi].copy_(p.data.view(-1)[lo:hi])
                    for p in sync.tail_params(bi):
                        _adam(p.data.view(-1), p.grad.view(-1), state[p]['exp_avg'].view(-1), state[p]['exp_avg_sq'].view(-1), step + 1)
                    sync.gather_operand_copies(bi)
                else:
                    for p in plist:
                        g = sync.lowp_views[id(p)].float() if id(p) in is_direct else p.grad
                        _adam(p.data.view(-1), g.reshape(-1), state[p]['exp_avg'].view(-1), state[p]['exp_avg_sq'].view(-1), step + 1)
                        if id(p) in is_direct:
                            copies[id(p)] = p.detach().to(torch.bfloat16)
        sync.masters_dirty = shard
        before = {id(p): p.detach().clone() for p in direct}
        filled = {}
        if shard and step == 1:
            # the Ctrl-C path of train.train: no collective -- the masters outside the own slice come from the gathered copies
            assert sync.fill_masters_from_arena() and sync.masters_dirty
            filled = {id(p): p.detach().clone() for p in direct}
            for p in direct:
                p.data.copy_(before[id(p)])                  # (back to the stale state: sync_masters() below is checked on it)
        # (on the GPU the next forward pass reads the gathered 16-bit copies; the CPU oracle network reads the fp32 masters: complete them)
        sync.sync_masters(_Opt(state))
    names = {id(p): n for n, p in net.named_parameters()}
    before = {names[k]: v for k, v in before.items()}
    filled = {names[k]: v for k, v in filled.items()}
    torch.save({'state': {k: v.clone() for k, v in net.state_dict().items()}, 'before': before, 'filled': filled,
                'copies': {names[k]: v.float().clone() for k, v in copies.items()},
                'm': {names[id(p)]: st['exp_avg'].clone() for p, st in state.items()}}, os.path.join(out_dir, f'rank{rank}.pt'))
    dist.destroy_process_group()


def test_sharded_optimizer_machinery_equals_replicated_update(tmp_path):
    """World size 2 over gloo, CPU tensors: reduce-scatter of the bf16 wire gradients (sums of gradients seeded with 1 / N), Adam on each
    rank's slice, all-gather of the operand copies, sync_masters() -- bit-identical to the replicated bf16-wire update (N = 2: halving is
    exact), replicas identical, and the masters really were incomplete before sync_masters()."""
    a, b = tmp_path / 'shard', tmp_path / 'repl'
    a.mkdir(), b.mkdir()
    mp.spawn(_shard_worker, args=(2, _free_port(), True, str(a)), nprocs=2, join=True)
    mp.spawn(_shard_worker, args=(2, _free_port(), False, str(b)), nprocs=2, join=True)
    s0, s1 = torch.load(os.path.join(a, 'rank0.pt')), torch.load(os.path.join(a, 'rank1.pt'))
    r0 = torch.load(os.path.join(b, 'rank0.pt'))
    for k, v in r0['state'].items():
        assert torch.equal(s0['state'][k], v), f'{k}: sharded update != replicated update'
        assert torch.equal(s0['state'][k], s1['state'][k]), f'replicas diverged at {k}'
    for k, v in r0['m'].items():
        assert torch.equal(s0['m'][k], v) and torch.equal(s1['m'][k], v), f'{k}: exp_avg'
    for k, v in r0['copies'].items():
        assert torch.equal(s0['copies'][k], v) and torch.equal(s1['copies'][k], v), f'{k}: operand copy'
        assert torch.equal(v, r0['state'][k].to(torch.bfloat16).float())
    # some direct parameter of each rank was stale outside the rank's slice before the masters were completed
    assert any(not torch.equal(s0['before'][k], s0['state'][k]) for k in s0['before'])
    assert any(not torch.equal(s1['before'][k], s1['state'][k]) for k in s1['before'])
    # an interrupted run (no collective at the final checkpoint): fill_masters_from_arena() leaves, on every rank, each direct weight equal to
    # the true fp32 master inside the rank's slice and to its bf16 rounding outside -- never the stale value of the last sync
    for s in (s0, s1):
        assert s['filled'], 'the worker did not exercise the interrupted path'
        n_exact = n_rounded = 0
        for k, f in s['filled'].items():
            true = s['state'][k]
            exact = f == true
            rounded = f == true.to(torch.bfloat16).float()
            assert bool((exact | rounded).all()), f'{k}: filled masters are neither the fp32 master nor its bf16 rounding'
            n_exact += int((exact & ~rounded).sum())
            n_rounded += int((rounded & ~exact).sum())
        assert n_exact > 0 and n_rounded > 0, 'own slice must stay fp32, the rest must come from the 16-bit copies'
        assert any(not torch.equal(s['before'][k], s['filled'][k]) for k in s['filled'])


def _presum_worker(rank, world, port, presum, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from spatiotemporal_variable_separation_amd.parallel import GradAllReducer, broadcast_module_state
    torch.set_num_threads(1)
    cfg = dict(CONFIGS['mlp_mul'], B=8)
    cond, target = make_batch(cfg)
    net = det_fill(cpu_ref.build_sep_net(cfg), salt=cfg['salt'] + rank)
    broadcast_module_state(net)
    sync = GradAllReducer(net.parameters(), bucket_bytes=16 << 10, overlap=False)
    sync.presummed = presum
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    part = slice(rank * 4, rank * 4 + 4)
    for step in range(2):
        sync.zero_grad()
        _losses(net, cond[part], target[part], cfg, 5 + step).backward(torch.tensor(1.0 / world if presum else 1.0))
        sync.all_reduce()
        opt.step()
    torch.save({k: v.clone() for k, v in net.state_dict().items()}, os.path.join(out_dir, f'rank{rank}.pt'))
    dist.destroy_process_group()


def test_sums_of_gradients_seeded_with_one_over_n_equal_averages(tmp_path):
    """GradAllReducer.presummed (what a recorded data-parallel step uses): the loss gradient seeded with 1 / N and the buckets SUMMED is the
    averaged-gradient step bit for bit at N = 2 (halving is exact), replicas identical."""
    a, b = tmp_path / 'sum', tmp_path / 'avg'
    a.mkdir(), b.mkdir()
    mp.spawn(_presum_worker, args=(2, _free_port(), True, str(a)), nprocs=2, join=True)
    mp.spawn(_presum_worker, args=(2, _free_port(), False, str(b)), nprocs=2, join=True)
    s0, s1 = torch.load(os.path.join(a, 'rank0.pt')), torch.load(os.path.join(a, 'rank1.pt'))
    r0 = torch.load(os.path.join(b, 'rank0.pt'))
    for k, v in r0.items():
        assert torch.equal(s0[k], v), f'{k}: summed 1 / N-seeded gradients != averaged gradients'
        assert torch.equal(s0[k], s1[k]), f'replicas diverged at {k}'
