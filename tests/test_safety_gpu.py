"""GPU: failure paths that must be loud -- the rollout's bounded inter-workgroup waits, the bench's multi-rank launch, the eager
fallback of a recorded data-parallel step."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle.detdata import det_fill
from oracle.golden_configs import CONFIGS, make_batch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rollout_exchange_timeout_is_sticky_and_raises():
    """A wait that gives up (forced here with a spin limit of 1) leaves a STICKY error word: the next check raises, whatever ran
    in between, and clears it."""
    from spatiotemporal_variable_separation_amd import functional as VF, ops
    from spatiotemporal_variable_separation_amd._lib import VarsepHipError
    from spatiotemporal_variable_separation_amd.networks.factory import build_sep_net
    from spatiotemporal_variable_separation_amd.train import check_rollout_exchange, compute_losses
    cfg = dict(CONFIGS['mlp_mul'], B=32, res_hidden_size=128, n_blocks=2)       # weight-stationary form: hidden 128, bf16
    net = det_fill(build_sep_net(cfg), salt=cfg['salt']).cuda()
    net.train()
    cond, target = make_batch(cfg)
    cond, target = cond.cuda(), target.cuda()
    lam = cfg['lambdas']
    dev = cond.device

    def run():
        with VF.precision('bf16'):
            total = compute_losses(cond, target, net, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], False, lam['ae'], lam['s'], lam['t'],
                                   lam['pred'], t_random=4)[0]
            total.backward()
        torch.cuda.synchronize()
        return total.item()
    ref = run()
    check_rollout_exchange(dev)                      # healthy run: nothing to report
    os.environ['VS_ROLLOUT_SPIN_LIMIT'] = '1'
    try:
        run()
    finally:
        del os.environ['VS_ROLLOUT_SPIN_LIMIT']
    again = run()                                    # a healthy launch afterwards must NOT clear the word
    assert abs(again - ref) <= 1e-5 * abs(ref)
    with pytest.raises(VarsepHipError, match='exchange timed out'):
        check_rollout_exchange(dev)
    check_rollout_exchange(dev)                      # cleared by the check that reported it
    assert ops.rollout_exchange_error(dev) == 0


def _ws_net(B=128, hidden=128, blocks=2):
    from spatiotemporal_variable_separation_amd.networks.factory import build_sep_net
    cfg = dict(CONFIGS['mlp_mul'], B=B, res_hidden_size=hidden, n_blocks=blocks)     # weight-stationary form; B = 128: 8 row slabs
    net = det_fill(build_sep_net(cfg), salt=cfg['salt']).cuda()
    net.train()
    cond, target = make_batch(cfg)
    return cfg, net, cond.cuda(), target.cuda()


def test_exchange_guard_makes_every_optimizer_launch_skip():
    """include/varsep_hip.h, vs_exchange_guard_set: while the process's guard word is non-zero, vs_adam_multi, vs_gemm_adam and the step
    counter leave parameters, moments, operand copies and the step count untouched (an update is never computed from a timed-out exchange);
    with the word cleared the same calls update."""
    from spatiotemporal_variable_separation_amd import ops
    from spatiotemporal_variable_separation_amd.optim import Adam
    dev = torch.device('cuda', torch.cuda.current_device())
    guard = ops.exchange_guard(dev)
    assert guard is not None and ops.rollout_exchange_error(dev) == 0
    torch.manual_seed(3)
    w = torch.nn.Parameter(torch.randn(256, 384, device=dev))
    b = torch.nn.Parameter(torch.randn(384, device=dev))
    opt = Adam([w, b], lr=1e-2)
    w.grad, b.grad = torch.randn_like(w), torch.randn_like(b)
    opt.step()                                                   # creates the state
    torch.cuda.synchronize()
    w0, b0 = w.detach().clone(), b.detach().clone()
    m0 = opt.state[w]['exp_avg'].clone()
    t0 = int(opt.param_groups[0]['step_dev'].item())
    assert ops.exchange_skipped_steps(dev) == 0
    guard[0] = 1
    opt.step()
    opt.step()                                                   # a second batch consumed while the (sticky) word is still up
    # the fused weight-gradient + Adam launch on the same parameter (G = A^T B as in MLPChain.backward: operands [rows, features], layout S)
    a = torch.randn(128, 256, device=dev).bfloat16()
    g = torch.randn(128, 384, device=dev).bfloat16()
    st = opt.state[w]
    shadow = torch.empty_like(w, dtype=torch.bfloat16)
    ops.gemm_adam(a, ops.LAYOUT_S, g, ops.LAYOUT_S, 256, 384, 128, w.data, st['exp_avg'], st['exp_avg_sq'], shadow, opt.param_groups[0]['step_dev'], 0, 1e-2,
                  (0.9, 0.999), 1e-8)
    torch.cuda.synchronize()
    assert torch.equal(w.detach(), w0) and torch.equal(b.detach(), b0) and torch.equal(opt.state[w]['exp_avg'], m0)
    assert int(opt.param_groups[0]['step_dev'].item()) == t0, 'the step count must not advance on a guarded step'
    # the companion word (vs_exchange_skip_counter_set) says how many steps the guard cost: two optimizer steps were issued while it was up
    assert ops.exchange_skipped_steps(dev, reset=False) == 2 and ops.exchange_skipped_steps(dev) == 2 and ops.exchange_skipped_steps(dev) == 0
    assert ops.rollout_exchange_error(dev) == 1 and ops.rollout_exchange_error(dev) == 0          # reported once, cleared by the read
    opt.step()
    torch.cuda.synchronize()
    assert not torch.equal(w.detach(), w0) and int(opt.param_groups[0]['step_dev'].item()) == t0 + 1


def test_xcd_local_probe_refuses_a_split_ring_and_results_stay_exact(monkeypatch):
    """The XCD-local exchange of the integrator is a START-UP decision (ops._probe_rollout_exchange).  VS_ROLLOUT_XCD_LOCAL=2 keeps its plain
    stores but spreads every slab's ring over the XCDs -- the placement it must never meet: the probe rollout has to come back with the error
    word raised, the process falls back to the agent-scope stores, and the real launches then give what VS_ROLLOUT_XCD_LOCAL=0 gives (the
    integrator's codes bit for bit), with no error left behind."""
    from spatiotemporal_variable_separation_amd import functional as VF, ops
    from spatiotemporal_variable_separation_amd._lib import load_library, BF16
    from spatiotemporal_variable_separation_amd.train import compute_losses
    cfg, net, cond, target = _ws_net()
    lam = cfg['lambdas']
    dev = cond.device
    lib = load_library()

    def run():
        net.zero_grad()
        with VF.precision('bf16'):
            total, _, _, t_codes = compute_losses(cond, target, net, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], False, lam['ae'], lam['s'], lam['t'],
                                                  lam['pred'], t_random=4)
            total.backward()
        torch.cuda.synchronize()
        return t_codes.detach().clone(), [p.grad.clone() for p in net.parameters() if p.grad is not None]
    try:
        monkeypatch.setenv('VS_ROLLOUT_XCD_LOCAL', '0')
        ref_codes, ref_grads = run()
        monkeypatch.setenv('VS_ROLLOUT_XCD_LOCAL', '2')
        ops._XL_PROBED.clear()
        ops.rollout_xcd_local(True)
        C, H, nb = cfg['code_size_t'], cfg['res_hidden_size'], cfg['n_blocks']
        assert lib.vs_mlp_rollout_xcd_local_get(BF16, cfg['B'], C, H, nb) == 1            # before the probe the launch WOULD take the plain stores
        codes, grads = run()
        assert lib.vs_mlp_rollout_xcd_local_get(BF16, cfg['B'], C, H, nb) == 0, 'the probe must have refused the split ring'
        assert ops.rollout_exchange_error(dev) == 0, 'the probe consumes its own verdict'
        assert torch.equal(codes, ref_codes)                       # the integrator's results: bit for bit
        for a, b in zip(grads, ref_grads):                          # (the chains' weight gradients go through split-K plans whose slab count
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-6 * float(b.abs().max()) + 1e-12)     #  follows the first call's workspace: not bitwise)
    finally:
        ops._XL_PROBED.clear()
        ops.rollout_xcd_local(True)


def test_recorded_training_survives_a_mid_run_exchange_timeout(monkeypatch, tmp_path, capfd):
    """`train()` must never apply an optimizer step computed from a timed-out exchange, and must not abort.  The probe
    is switched off and the ring split (VS_ROLLOUT_XCD_LOCAL=2), so the FIRST replay of the recorded step times out: the guard word makes the
    recorded optimizer launches skip (parameters bit-identical to the start), `train()` notices at its check after the first replay, switches the
    process to the agent-scope exchange, re-records and goes on: the parameters after the run equal those of a run that used the agent-scope
    exchange from the start on the batches the first run actually applied."""
    from spatiotemporal_variable_separation_amd import functional as VF, ops
    from spatiotemporal_variable_separation_amd.optim import Adam
    from spatiotemporal_variable_separation_amd.train import train
    dev = torch.device('cuda', torch.cuda.current_device())

    def fit(xp, loader, xl, probe):
        monkeypatch.setenv('VS_ROLLOUT_XCD_LOCAL', xl)
        monkeypatch.setenv('VARSEP_ROLLOUT_PROBE', probe)
        monkeypatch.setenv('VS_ROLLOUT_SPIN_LIMIT', str(1 << 14))
        ops._XL_PROBED.clear()
        ops.rollout_xcd_local(True)
        cfg, net, cond, target = _ws_net()
        opt = Adam(net.parameters(), lr=1e-3, betas=(0.9, 0.99))
        lam = cfg['lambdas']
        np.random.seed(5)
        VF.set_precision('bf16')
        batches = [(cond, target)] * loader
        train(str(xp), batches, dev, net, opt, None, False, False, 1, lam['ae'], lam['s'], lam['t'], lam['pred'], cfg['offset'], cfg['nt_cond'],
              cfg['nt_pred'], False, False, None, False, hip_graph=True)
        torch.cuda.synchronize()
        return {k: v.detach().clone() for k, v in net.state_dict().items()}
    try:
        start = {k: v.detach().clone() for k, v in _ws_net()[1].state_dict().items()}
        # one batch, ring split, no probe: the only step times out and must leave the parameters untouched
        one = fit(tmp_path / 'a', 1, '2', '0')
        err = capfd.readouterr().err
        assert 'exchange timed out' in err and 'agent-scope' in err
        for k in start:
            assert torch.equal(one[k], start[k]), 'a guarded step changed %s' % k
        # four batches: the first is lost to the time-out, three are applied under the re-recorded agent-scope step
        got = fit(tmp_path / 'b', 4, '2', '0')
        assert any(not torch.equal(got[k], start[k]) for k in start), 'training did not continue after the time-out'
        assert all(torch.isfinite(v).all() for v in got.values() if v.is_floating_point())
        assert ops.rollout_exchange_error(dev) == 0
    finally:
        ops._XL_PROBED.clear()
        ops.rollout_xcd_local(True)
        VF.set_precision('fp32')


def test_bench_gpus_2_launches_two_ranks():
    """`python bench.py --gpus 2` starts two rank processes itself; on the one-GPU box they share cuda:0 and average gradients
    over gloo (VARSEP_BENCH_SHARE_GPU=1).  The JSON line must say n_gpus 2 and a doubled global batch."""
    env = dict(os.environ, VARSEP_BENCH_SHARE_GPU='1')
    env.pop('RANK', None), env.pop('WORLD_SIZE', None), env.pop('LOCAL_RANK', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--repeats', '2',
                        '--batch', '32', '--no_cpu_baseline'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 64 and out['config']['parallelism'] == 'dp2'
    assert out['value'] > 0 and np.isfinite(out['config']['final_loss'])
    assert 'gloo' in out['config']['grad_allreduce']


def _ddp_worker(rank, world, port, out_dir, graph):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from spatiotemporal_variable_separation_amd import functional as VF
    from spatiotemporal_variable_separation_amd.networks.factory import build_sep_net
    from spatiotemporal_variable_separation_amd.optim import Adam
    from spatiotemporal_variable_separation_amd.parallel import GradAllReducer, broadcast_module_state
    from spatiotemporal_variable_separation_amd.train import chain_weight_parameters, train
    cfg = dict(CONFIGS['mlp_mul'], B=8)
    net = det_fill(build_sep_net(cfg), salt=cfg['salt']).cuda()
    broadcast_module_state(net)
    cond, target = make_batch(cfg)
    other_c, other_t = make_batch(dict(cfg, salt=cfg['salt'] + 100))           # different frames: different gradients
    per = 8 // world
    sh = slice(rank * per, rank * per + per)
    full = (cond[sh].cuda(), target[sh].cuda())
    ragged = (other_c[sh][:per - 1].cuda(), other_t[sh][:per - 1].cuda())       # the last batch of the epoch is one sample short
    loader = [full, full, ragged]
    sync = GradAllReducer(net.parameters(), bucket_bytes=16 << 10, comm_dtype=torch.bfloat16,
                          lowp_direct=chain_weight_parameters(net) if graph else None)
    opt = Adam(net.parameters(), lr=1e-3, betas=(0.9, 0.99))
    lam = cfg['lambdas']
    np.random.seed(3)
    if not graph:                                    # GraphedStep's capture consumes one draw of the t_random stream
        np.random.randint(cfg['nt_cond'], cond.shape[1] + target.shape[1] + 1)
    VF.set_precision('bf16')
    xp = os.path.join(out_dir, 'xp_graph' if graph else 'xp_eager')
    train(xp, loader, torch.device('cuda', 0), net, opt, None, False, False, 1, lam['ae'], lam['s'], lam['t'], lam['pred'],
          cfg['offset'], cfg['nt_cond'], cfg['nt_pred'], False, False, None, False, grad_sync=sync, hip_graph=graph)
    torch.cuda.synchronize()
    dist.barrier()
    torch.save({'state': {k: v.detach().cpu() for k, v in net.state_dict().items()}, 'saved': os.path.exists(os.path.join(xp, 'ov_Et.pt')),
                'rank_dir': os.path.exists(os.path.join(xp, 'rank1'))}, os.path.join(out_dir, f'{"g" if graph else "e"}_rank{rank}.pt'))
    dist.destroy_process_group()


def test_ddp_graph_with_ragged_last_batch_uses_this_steps_gradients(tmp_path):
    """--ddp --hip_graph --grad_comm bf16 with a ragged last batch: the eager fallback must average and apply the gradients of
    THAT batch (it used to leave the Linear chains' weights on the previous step's bf16 wire images).  Checked against the same
    three steps run entirely in the eager data-parallel loop: identical up to where the bf16 rounding of a gradient happens
    (GEMM epilogue vs cast of the fp32 bucket); a stale gradient would differ by ~lr on every chain weight.  Replicas stay
    identical and only rank 0 writes the checkpoint."""
    import socket
    import torch.multiprocessing as mp

    def port():
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        p = s.getsockname()[1]
        s.close()
        return p
    mp.spawn(_ddp_worker, args=(2, port(), str(tmp_path), True), nprocs=2, join=True)
    mp.spawn(_ddp_worker, args=(2, port(), str(tmp_path), False), nprocs=2, join=True)
    g0, g1 = torch.load(os.path.join(tmp_path, 'g_rank0.pt')), torch.load(os.path.join(tmp_path, 'g_rank1.pt'))
    e0 = torch.load(os.path.join(tmp_path, 'e_rank0.pt'))
    for k in g0['state']:
        assert torch.equal(g0['state'][k], g1['state'][k]), f'replicas diverged at {k}'
        a, b = g0['state'][k], e0['state'][k]
        assert torch.allclose(a, b, rtol=1e-3, atol=1.5e-4), f'{k}: recorded + ragged-eager differs from all-eager by {(a - b).abs().max().item():.3e}'
    assert g0['saved'] and not g0['rank_dir'], 'rank 0 writes the checkpoint, no per-rank directories'
