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
