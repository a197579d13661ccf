"""Headline benchmark: training frames/s of the var_sep hot path on MI355X (contract in the task statement).

    python bench.py --gpus N --steps K --warmup W [--config waveeq] [--precision bf16]

One step = ae_loss + zero_order_loss + get_forecast + forecast MSE + t-regulariser, backward, gradient all-reduce
(N > 1) and the Adam update, on one seeded synthetic batch per rank that is resident in HBM before timing starts.
`value` = N * batch * nt_pred / step time (whole-job predicted frames per second).  Default workload = BASELINE.json
configs[1] (WaveEq MLP, bf16), the configuration the metric is quoted on that fits one GPU.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np          # noqa: E402
import torch                # noqa: E402


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=20)
    p.add_argument('--warmup', type=int, default=5)
    p.add_argument('--config', default='waveeq')
    p.add_argument('--precision', default='bf16', choices=['bf16', 'fp32'])
    p.add_argument('--batch', type=int, default=None, help='per-GPU batch (default: the config\'s)')
    p.add_argument('--no_cpu_baseline', action='store_true')
    p.add_argument('--no_graph', action='store_true', help='issue every kernel from Python instead of replaying a hipGraph')
    p.add_argument('--cpu_steps', type=int, default=None)
    return p.parse_args()


def dense_flops_and_bytes(net, cfg, esize):
    """Algorithmic forward FLOPs / fused bytes per step (sum over Linear/conv calls of 2*MACs and
    (|in|+|W|+|out|)*esize), measured with forward hooks on the product modules' parameter holders is not possible
    (they are never called), so dense layers are counted analytically for the MLP family."""
    B, n = cfg['batch'], cfg['nt_pred'] + cfg['offset']
    calls = {'Es': 2 * B, 'Et': 2 * B, 'decoder': (n + 1) * B, 't_resnet': (n - 1) * B}
    fl = by = 0.0
    import torch.nn as nn
    for name, rows in calls.items():
        mod = getattr(net, name)
        for m in mod.modules():
            if isinstance(m, nn.Linear):
                fl += 2.0 * rows * m.in_features * m.out_features
                by += (rows * m.in_features + m.in_features * m.out_features + rows * m.out_features) * esize
    return fl, by


def cpu_baseline(cfg, steps):
    """Time the CPU oracle (plain PyTorch fp32 restatement of the reference) on the host cores, same workload."""
    from oracle import cpu_ref
    from spatiotemporal_variable_separation_amd.data.synthetic import synthetic_batch
    torch.manual_seed(1234)
    np.random.seed(1234)
    ocfg = dict(cfg)
    net = cpu_ref.build_sep_net(ocfg)
    net.train()
    cond, target = synthetic_batch(cfg['data'], cfg['batch'], cfg['nt_cond'], cfg['nt_pred'], seed=1234)
    lam = cfg['lambdas']
    opt = torch.optim.Adam(net.parameters(), lr=4e-4, betas=(0.9, 0.99))

    def step():
        opt.zero_grad()
        total, _, _, _ = cpu_ref.training_losses(cond, target, net, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'],
                                                 cfg.get('skipco', False), lam['ae'], lam['s'], lam['t'], lam['pred'],
                                                 average_tloss=bool(cfg.get('average_tloss')))
        total.backward()
        opt.step()
    # thread counts: 8 (comparable with the survey container) and a quarter of the logical CPUs; best one is reported
    default_threads = torch.get_num_threads()
    tried = {}
    for nthr in sorted({8, max(8, min(64, (os.cpu_count() or 8) // 4))}):
        torch.set_num_threads(nthr)
        step()
        t0 = time.time()
        for _ in range(steps):
            step()
        tried[nthr] = (time.time() - t0) / steps
    torch.set_num_threads(default_threads)
    nthr, dt = min(tried.items(), key=lambda kv: kv[1])
    return {'value': cfg['batch'] * cfg['nt_pred'] / dt, 'unit': 'frames/s', 'cores': nthr,
            'kind': 'port', 'sample': f'{steps} full training steps of the same workload (batch {cfg["batch"]}, fp32, CPU '
            f'oracle = plain-PyTorch restatement of the reference) after 1 warm-up, best of thread counts '
            f'{ {k: round(v * 1e3) for k, v in tried.items()} } ms/step on {os.cpu_count()} logical CPUs',
            'ms_per_step': dt * 1e3}


def main():
    args = parse()
    # stdout carries exactly ONE line (the result JSON): libraries that print banners to the C-level stdout (RCCL prints its
    # version block there at communicator creation) are diverted to stderr for the whole run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    # VARSEP_BENCH_FORCE_DIST=1 runs the data-parallel code path (process group, flat buckets, RCCL all-reduce per step)
    # at world size 1, so the N>1 path can be exercised on a one-GPU box
    ddp = world > 1 or os.environ.get('VARSEP_BENCH_FORCE_DIST') == '1'
    if ddp:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group('nccl', device_id=dev, rank=rank, world_size=world)

    from spatiotemporal_variable_separation_amd import functional as VF, ops
    from spatiotemporal_variable_separation_amd.configs import BASELINE_CONFIGS
    from spatiotemporal_variable_separation_amd.data.synthetic import synthetic_batch
    from spatiotemporal_variable_separation_amd.networks.factory import build_sep_net
    from spatiotemporal_variable_separation_amd.parallel import GradAllReducer, broadcast_module_state
    from spatiotemporal_variable_separation_amd.train import compute_losses

    cfg = dict(BASELINE_CONFIGS[args.config])
    if args.batch:
        cfg['batch'] = args.batch
    torch.manual_seed(1234)
    np.random.seed(1234)                     # same t_random sequence on every rank
    net = build_sep_net(cfg).to(dev)
    net.train()
    if ddp:
        broadcast_module_state(net)
    # bf16 mode: gradients travel as bf16 (VARSEP_GRAD_COMM=fp32 keeps fp32 on the wire); reported in config.grad_allreduce
    comm_bf16 = args.precision == 'bf16' and os.environ.get('VARSEP_GRAD_COMM', 'bf16') == 'bf16'
    from spatiotemporal_variable_separation_amd.train import chain_weight_parameters
    direct = chain_weight_parameters(net) if (comm_bf16 and os.environ.get('VARSEP_GRAD_DIRECT_LOWP', '1') == '1') else None
    sync = GradAllReducer(net.parameters(), force=(world == 1), comm_dtype=torch.bfloat16 if comm_bf16 else torch.float32,
                          lowp_direct=direct) if ddp else None
    from spatiotemporal_variable_separation_amd.train import GraphedStep, _mlp_family
    # every family replays a recorded step (train.GraphedStep); same-box A/B: SST 72.5 -> 68.5 ms, MNIST B=16 7.5 -> 6.1 ms
    use_graph = (not args.no_graph) and os.environ.get('VARSEP_BENCH_GRAPH_ALL', '1') == '1'
    from spatiotemporal_variable_separation_amd.optim import Adam
    opt = Adam(net.parameters(), lr=4e-4, betas=(0.9, 0.99))
    from spatiotemporal_variable_separation_amd.train import enable_update_in_backward
    enable_update_in_backward(opt, net, sync)
    cond, target = synthetic_batch(cfg['data'], cfg['batch'], cfg['nt_cond'], cfg['nt_pred'], device=dev,
                                   seed=1234 + rank)
    lam = cfg['lambdas']
    VF.set_precision(args.precision)
    VF.fold_repeated_gradients(sync is None and os.environ.get('VARSEP_FOLD_GRADS') == '1')   # as train() does: off unless asked for

    def step():
        if sync is not None:
            sync.zero_grad()
        else:
            opt.zero_grad(set_to_none=True)
        total, _, _, _ = compute_losses(cond, target, net, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'],
                                        cfg.get('skipco', False), lam['ae'], lam['s'], lam['t'], lam['pred'],
                                        average_tloss=bool(cfg.get('average_tloss')))
        total.backward()
        if sync is not None:
            sync.all_reduce()
        opt.step()
        VF.flush_bn_call_counts()
        return total

    def barrier():
        if ddp:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    graphed = None
    if use_graph:
        # the timed region replays the recorded step (train.GraphedStep: the same kernels, launched by hipGraphLaunch instead
        # of ~110 Python-issued launches); with N > 1 ranks: graph(losses + backward) -> bucket all-reduces -> graph(Adam)
        graphed = GraphedStep(net, opt, cond, target, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'],
                              (lam['ae'], lam['s'], lam['t'], lam['pred']), bool(cfg.get('average_tloss')),
                              warmup=max(1, min(args.warmup, 3)), grad_sync=sync)
        timed_step = graphed.step
    else:
        timed_step = step
    for _ in range(args.warmup):
        timed_step()
    # HIP-event pairs around every vs_* launch of the timed region (events pre-created: ~2 x launches/step x steps)
    events_on = os.environ.get('VARSEP_BENCH_NO_EVENTS') is None
    n_inst = 4                               # instrumented eager steps after a graph-replay timed region
    ops.profile_reset(enable=events_on and graphed is None, pool=256 * (args.steps // 4 + 8 + n_inst))
    barrier()
    t0 = time.perf_counter()
    if graphed is not None:
        for i in range(args.steps):
            loss = graphed.step()
    else:
        for i in range(args.steps):
            ops._PROF['on'] = events_on and (i % 4 == 3 or args.steps < 8)     # sample every 4th step: keeps host overhead low
            loss = step()
    barrier()
    dt = time.perf_counter() - t0
    sampled = len([i for i in range(args.steps) if i % 4 == 3 or args.steps < 8])
    inst_ms = None
    if graphed is not None and events_on:
        # per-kernel durations cannot be taken inside a graph replay (events are not recordable there): the same step is
        # run eagerly a few times afterwards with an event pair around every vs_* launch
        torch.cuda.synchronize()
        step()
        ops.profile_reset(enable=True, pool=256 * (n_inst + 2))
        for i in range(n_inst):
            step()
        torch.cuda.synchronize()
        sampled = n_inst
    prof = ops.profile_collect()
    if ddp:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    ms = dt / args.steps * 1e3
    frames = world * cfg['batch'] * cfg['nt_pred']
    if rank != 0:
        torch.distributed.destroy_process_group()
        return

    # roofline position of every instrumented kernel family of the timed region, largest summed event time first;
    # `roofline` is the dominant one, `roofline_others` the next ones (the step is spread over several kernels)
    def roof_of(name, rec):
        base = {'kernel': name, 'launches_per_step': rec['n'] / sampled, 'avg_launch_us': round(rec['ms'] * 1e3 / rec['n'], 2),
                'share_of_step': round(rec['ms'] / (ms * sampled), 3), 'traffic': None}
        if rec['flops'] > 0:
            peak = 2500.0 if args.precision == 'bf16' else 157.3
            ach = rec['flops'] / (rec['ms'] * 1e-3) / 1e12
            base.update({'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4)})
        else:
            ach = rec['bytes'] / (rec['ms'] * 1e-3) / 1e9
            base.update({'bound': 'hbm', 'achieved': round(ach, 1), 'peak': 8000.0, 'unit': 'GB/s', 'frac': round(ach / 8000.0, 4)})
        if name in traffic:
            base['traffic'] = round(traffic[name]['bytes_per_launch'])
            base['traffic_source'] = 'profiles/r01_waveeq_bf16_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)'
        if 'rollout' in name:
            base['note'] = ('sequential recurrence: (n-1)*n_blocks*3 dependent 16-row GEMMs per slab, bound by per-CU L2 '
                            'weight streaming and barrier latency, not by MFMA rate (SURVEY.md H3)')
        return base
    # HBM-side traffic per launch from the committed rocprofv3 --pmc passes (collected separately: PMC passes cannot
    # run inside the timed region); only attached for the default workload they were measured on
    traffic = {}
    tpath = os.path.join(ROOT, 'profiles', 'r01_waveeq_bf16_traffic.json')
    if args.config == 'waveeq' and args.precision == 'bf16' and cfg['batch'] == 128 and os.path.exists(tpath):
        traffic = json.load(open(tpath))
    roof, others = None, []
    if prof:
        ranked = sorted(prof.items(), key=lambda kv: -kv[1]['ms'])
        roof = roof_of(*ranked[0])
        others = [roof_of(*kv) for kv in ranked[1:6]]
    out = {
        'metric': 'training frames/sec (seq x nt_pred)', 'value': round(frames / (ms * 1e-3), 1), 'unit': 'frames/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 4),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.precision, 'data': 'synthetic',
        'config': {'workload': f'{args.config}: {cfg["architecture"]} enc/dec, batch {cfg["batch"]}/GPU, '
                               f'nt_cond {cfg["nt_cond"]}, nt_pred {cfg["nt_pred"]}, offset {cfg["offset"]}',
                   'global_batch': world * cfg['batch'], 'parallelism': f'dp{world}',
                   'grad_allreduce': ('none (1 rank)' if not ddp else ('bf16 buckets over RCCL' if comm_bf16 else 'fp32 buckets over RCCL')), 'optimizer': 'Adam (vs_adam_multi, one HIP launch)', 'launch': ('hipGraph replay (per-kernel roofline timings from eager instrumented steps after the timed region)' if use_graph else 'eager'),
                   'final_loss': round(float(loss.item()), 5)},
        'roofline': roof, 'roofline_others': others,
    }
    if world == 1 and not args.no_cpu_baseline:
        steps = args.cpu_steps or (5 if args.config in ('waveeq', 'mnist_b16') else 2)
        out['cpu_baseline'] = cpu_baseline(cfg, steps)
        out['cpu_baseline']['value'] = round(out['cpu_baseline']['value'], 1)
        out['cpu_baseline']['ms_per_step'] = round(out['cpu_baseline']['ms_per_step'], 1)
    os.write(result_fd, (json.dumps(out) + '\n').encode())
    if ddp:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
