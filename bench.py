"""Headline benchmark: training frames/s of the var_sep hot path on MI355X (contract in the task statement).

    python bench.py --gpus N --steps K --warmup W [--config waveeq] [--precision bf16|fp16|fp32]

One step = ae_loss + zero_order_loss + get_forecast + forecast MSE + t-regulariser, backward, gradient all-reduce
(N > 1) and the Adam update, on one seeded synthetic batch per rank that is resident in HBM before timing starts.
`value` = N * batch * nt_pred / step time (whole-job predicted frames per second).  Default workload = BASELINE.json
configs[1] (WaveEq MLP, bf16), the configuration the metric is quoted on that fits one GPU.

Ranks: under `torch.distributed.run` (RANK / WORLD_SIZE in the environment) this process IS one rank.  Started plainly with
`--gpus N` (N > 1) it launches N rank processes itself -- before anything touches the GPU -- one per device over RCCL, and
rank 0 prints the JSON line.  VARSEP_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and exchanges gradients over gloo (the
one-GPU test of the N > 1 path, tests/test_bench_gpu.py).

Timing: W warm-up steps, then `--repeats` (default 5) timed regions of EXACTLY K steps each, every region bracketed by
barrier + torch.cuda.synchronize() on both sides and reduced with MAX over ranks; the reported ms_per_step is the median region
(all of them are listed in `ms_per_step_all`).

At N = 1 the line also carries `configs`: the other single-GPU BASELINE workloads (Moving-MNIST DCGAN B=128, TaxiBJ VGG B=100,
SST nt_pred 40 B=8) timed the same way with fewer steps, each with the roofline position of its dominant kernel family.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import spatiotemporal_variable_separation_amd  # noqa: E402,F401  (sets the HIP runtime's queue knobs before torch initialises HIP)

import numpy as np          # noqa: E402
import torch                # noqa: E402

EXTRA_DEFAULT = 'mnist_b128,taxibj,sst'
EXTRA_STEPS = {'mnist_b128': (10, 3), 'taxibj': (8, 3), 'sst': (3, 2), 'mnist_b16': (10, 3), 'chairs': (6, 2), 'waveeq': (20, 5)}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=20)
    p.add_argument('--warmup', type=int, default=5)
    p.add_argument('--repeats', type=int, default=5, help='timed regions of --steps steps each; the median is reported')
    p.add_argument('--config', default='waveeq')
    p.add_argument('--precision', default='bf16', choices=['bf16', 'fp16', 'fp32'])
    p.add_argument('--batch', type=int, default=None, help='per-GPU batch (default: the config\'s)')
    p.add_argument('--no_cpu_baseline', action='store_true')
    p.add_argument('--no_graph', action='store_true', help='issue every kernel from Python instead of replaying a hipGraph')
    p.add_argument('--cpu_steps', type=int, default=None)
    p.add_argument('--extra_configs', default=None,
                   help='comma list of further workloads appended under "configs" (default at N=1 with the default workload: '
                        + EXTRA_DEFAULT + '; "none" disables)')
    return p.parse_args()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args):
    """`python bench.py --gpus N` outside a launcher: start N rank processes (children of this one, which never touches the GPU:
    torch.cuda.device_count() does not initialise it) and exit with the worst return code.  Rank 0 inherits stdout."""
    n = args.gpus
    share = os.environ.get('VARSEP_BENCH_SHARE_GPU') == '1'
    have = torch.cuda.device_count()
    if have < n and not share:
        sys.stderr.write(f'bench.py: --gpus {n} but only {have} device(s) visible (VARSEP_BENCH_SHARE_GPU=1 shares cuda:0 over gloo)\n')
        sys.exit(2)
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        for k in env.pop('VARSEP_PACKAGE_SET', '').split():     # single-GPU runtime knobs the package set in THIS process
            env.pop(k, None)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rcs = [p.wait() for p in procs]
    sys.exit(max(abs(rc) for rc in rcs))


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
    return {'value': round(cfg['batch'] * cfg['nt_pred'] / dt, 1), 'unit': 'frames/s', 'cores': nthr,
            'kind': 'port', 'sample': f'{steps} full training steps of the same workload (batch {cfg["batch"]}, fp32, CPU '
            f'oracle = plain-PyTorch restatement of the reference) after 1 warm-up, best of thread counts '
            f'{ {k: round(v * 1e3) for k, v in tried.items()} } ms/step on {os.cpu_count()} logical CPUs',
            'ms_per_step': round(dt * 1e3, 1)}


class Ranks:
    """Process-group facts of this rank (one process per GPU)."""

    def __init__(self, args):
        self.rank = int(os.environ.get('RANK', 0))
        self.world = int(os.environ.get('WORLD_SIZE', 1))
        self.share = os.environ.get('VARSEP_BENCH_SHARE_GPU') == '1'
        local = 0 if self.share else int(os.environ.get('LOCAL_RANK', 0))
        torch.cuda.set_device(local)
        self.dev = torch.device('cuda', local)
        # VARSEP_BENCH_FORCE_DIST=1 runs the data-parallel code path (process group, flat buckets, RCCL all-reduce per step)
        # at world size 1, so the N>1 path can be exercised on a one-GPU box
        self.ddp = self.world > 1 or os.environ.get('VARSEP_BENCH_FORCE_DIST') == '1'
        self.backend = None
        if self.ddp:
            import torch.distributed as dist
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29511')
            self.backend = 'gloo' if self.share else 'nccl'
            if self.backend == 'nccl':
                dist.init_process_group('nccl', device_id=self.dev, rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group('gloo', rank=self.rank, world_size=self.world)
        if self.world != args.gpus and self.rank == 0:
            sys.stderr.write(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}: reporting the {self.world} rank(s) that run\n')

    def barrier(self):
        if self.ddp:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if not self.ddp:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=self.dev if self.backend == 'nccl' else 'cpu')
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return t.item()

    def close(self):
        if self.ddp:
            torch.distributed.destroy_process_group()


def run_workload(name, args, rk, steps, warmup, repeats, batch=None):
    """Build the workload `name`, warm up, time `repeats` regions of `steps` steps; returns timings + per-kernel event profile."""
    from spatiotemporal_variable_separation_amd import functional as VF, ops
    from spatiotemporal_variable_separation_amd.configs import BASELINE_CONFIGS
    from spatiotemporal_variable_separation_amd.data.synthetic import synthetic_batch
    from spatiotemporal_variable_separation_amd.networks.factory import build_sep_net
    from spatiotemporal_variable_separation_amd.optim import Adam
    from spatiotemporal_variable_separation_amd.parallel import GradAllReducer, broadcast_module_state
    from spatiotemporal_variable_separation_amd.train import (GraphedStep, chain_weight_parameters, compute_losses,
                                                               enable_fused_update, enable_update_in_backward, make_loss_scaler)
    dev = rk.dev
    cfg = dict(BASELINE_CONFIGS[name])
    if batch:
        cfg['batch'] = batch
    torch.manual_seed(1234)
    np.random.seed(1234)                     # same t_random sequence on every rank
    net = build_sep_net(cfg).to(dev)
    net.train()
    if rk.ddp:
        broadcast_module_state(net)
    # 16-bit modes: gradients travel as bf16 (VARSEP_GRAD_COMM=fp32 keeps fp32 on the wire); reported in config.grad_allreduce
    comm_bf16 = args.precision == 'bf16' and os.environ.get('VARSEP_GRAD_COMM', 'bf16') == 'bf16'
    direct = chain_weight_parameters(net) if (comm_bf16 and os.environ.get('VARSEP_GRAD_DIRECT_LOWP', '1') == '1') else None
    sync = GradAllReducer(net.parameters(), force=(rk.world == 1), comm_dtype=torch.bfloat16 if comm_bf16 else torch.float32,
                          lowp_direct=direct) if rk.ddp else None
    use_graph = (not args.no_graph) and os.environ.get('VARSEP_BENCH_GRAPH_ALL', '1') == '1'
    opt = Adam(net.parameters(), lr=4e-4, betas=(0.9, 0.99))
    enable_update_in_backward(opt, net, sync)
    cond, target = synthetic_batch(cfg['data'], cfg['batch'], cfg['nt_cond'], cfg['nt_pred'], device=dev, seed=1234 + rk.rank)
    lam = cfg['lambdas']
    VF.set_precision(args.precision)
    scaler = make_loss_scaler(dev) if args.precision == 'fp16' else None     # reference train.py:96-97: GradScaler with fp16 autocast
    if args.precision != 'fp32':
        enable_fused_update(opt, net, sync, scaler)      # (GraphedStep does the same; here for --no_graph and the instrumented eager steps)
    VF.fold_repeated_gradients(sync is None and os.environ.get('VARSEP_FOLD_GRADS', '1') == '1')

    def step():
        if sync is not None:
            sync.zero_grad()
        else:
            opt.zero_grad(set_to_none=True)
        total, _, _, _ = compute_losses(cond, target, net, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'],
                                        cfg.get('skipco', False), lam['ae'], lam['s'], lam['t'], lam['pred'],
                                        average_tloss=bool(cfg.get('average_tloss')))
        if scaler is not None:
            scaler.backward(total)
        else:
            total.backward()
        if sync is not None:
            sync.all_reduce()
        if scaler is not None:
            scaler.step(opt)
        else:
            opt.step()
        VF.flush_bn_call_counts()
        return total

    graphed = None
    if use_graph:
        # the timed region replays the recorded step (train.GraphedStep: the same kernels, launched by hipGraphLaunch instead
        # of Python-issued launches); with N > 1 ranks: graph(losses + backward) -> bucket all-reduces -> graph(Adam)
        graphed = GraphedStep(net, opt, cond, target, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'],
                              (lam['ae'], lam['s'], lam['t'], lam['pred']), bool(cfg.get('average_tloss')),
                              warmup=max(1, min(warmup, 3)), grad_sync=sync, scaler=scaler)
        timed_step = graphed.step
    else:
        timed_step = step
    for _ in range(warmup):
        timed_step()
    events_on = os.environ.get('VARSEP_BENCH_NO_EVENTS') is None
    n_inst = 3                                # instrumented eager steps after a graph-replay timed region
    ops.profile_reset(enable=False)
    regions = []
    loss = None
    for rep in range(max(1, repeats)):
        if graphed is None and events_on and rep == 0:
            # eager loop: HIP-event pairs around every vs_* launch of every 4th step of the first region (events pre-created)
            ops.profile_reset(enable=True, pool=512 * (steps // 4 + 8))
        rk.barrier()
        t0 = time.perf_counter()
        if graphed is not None:
            for i in range(steps):
                loss = graphed.step()
        else:
            for i in range(steps):
                ops._PROF['on'] = events_on and rep == 0 and (i % 4 == 3 or steps < 8)
                loss = step()
        rk.barrier()
        regions.append(rk.max_over_ranks(time.perf_counter() - t0))
        ops._PROF['on'] = False
    sampled = len([i for i in range(steps) if i % 4 == 3 or steps < 8])
    if graphed is not None and events_on:
        # per-kernel durations cannot be taken inside a graph replay (events are not recordable there): the same step is
        # run eagerly a few times afterwards with an event pair around every vs_* launch
        torch.cuda.synchronize()
        step()
        ops.profile_reset(enable=True, pool=1024 * (n_inst + 1))
        for i in range(n_inst):
            step()
        torch.cuda.synchronize()
        sampled = n_inst
    prof = ops.profile_collect() if events_on else {}
    err = ops.rollout_exchange_error(dev)
    if err:
        raise RuntimeError('the rollout kernels reported an inter-workgroup exchange time-out (code %d): results are invalid' % err)
    final_loss = float(loss.item())
    out = {'cfg': cfg, 'ms': statistics.median(regions) / steps * 1e3, 'ms_all': [round(r / steps * 1e3, 4) for r in regions],
           'prof': prof, 'sampled': sampled, 'loss': final_loss, 'comm_bf16': comm_bf16, 'use_graph': use_graph,
           'scaler': None if scaler is None else scaler.describe()}
    VF.fold_repeated_gradients(False)
    del graphed, net, opt, sync
    torch.cuda.empty_cache()
    return out


def rooflines(res, precision, traffic=None, top=6):
    """Roofline position of every instrumented kernel family, largest summed event time first."""
    prof, ms, sampled = res['prof'], res['ms'], res['sampled']
    traffic = traffic or {}

    def roof_of(name, rec):
        base = {'kernel': name, 'launches_per_step': round(rec['n'] / sampled, 2), 'avg_launch_us': round(rec['ms'] * 1e3 / rec['n'], 2),
                'share_of_step': round(rec['ms'] / (ms * sampled), 3), 'traffic': None}
        if rec['flops'] > 0 and not name.startswith('vs_gemm_adam'):     # (the fused weight-gradient + Adam launch is HBM-bound:
            peak = 157.3 if precision == 'fp32' else 2500.0              # 26 B per parameter against 2 K flop, K = 256)
            ach = rec['flops'] / (rec['ms'] * 1e-3) / 1e12
            base.update({'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4)})
        else:
            ach = rec['bytes'] / (rec['ms'] * 1e-3) / 1e9
            base.update({'bound': 'hbm', 'achieved': round(ach, 1), 'peak': 8000.0, 'unit': 'GB/s', 'frac': round(ach / 8000.0, 4)})
        if name in traffic:
            base['traffic'] = round(traffic[name]['bytes_per_launch'])
            base['traffic_source'] = traffic.get('_source', 'static')
        if 'rollout' in name:
            base['note'] = ('sequential recurrence: (n-1)*n_blocks*3 dependent 16-row GEMMs per slab, bound by per-CU L2 '
                            'weight streaming and barrier latency, not by MFMA rate (SURVEY.md H3)')
        return base
    if not prof:
        return None, []
    ranked = sorted(prof.items(), key=lambda kv: -kv[1]['ms'])
    return roof_of(*ranked[0]), [roof_of(*kv) for kv in ranked[1:top]]


def workload_text(name, cfg):
    return (f'{name}: {cfg["architecture"]} enc/dec, batch {cfg["batch"]}/GPU, nt_cond {cfg["nt_cond"]}, '
            f'nt_pred {cfg["nt_pred"]}, offset {cfg["offset"]}')


def main():
    args = parse()
    if args.gpus > 1 and 'RANK' not in os.environ:
        spawn_ranks(args)                    # does not return
    # stdout carries exactly ONE line (the result JSON): libraries that print banners to the C-level stdout (RCCL prints its
    # version block there at communicator creation) are diverted to stderr for the whole run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    rk = Ranks(args)
    res = run_workload(args.config, args, rk, args.steps, args.warmup, args.repeats, batch=args.batch)
    if rk.rank != 0:
        rk.close()
        return
    cfg, ms = res['cfg'], res['ms']
    frames = rk.world * cfg['batch'] * cfg['nt_pred']

    # HBM-side traffic per launch from the committed rocprofv3 --pmc passes (collected in their own runs: PMC passes cannot run
    # inside the timed region); static data, only attached for the workload they were measured on
    traffic = {}
    for tname in ('r02_waveeq_bf16_traffic.json', 'r01_waveeq_bf16_traffic.json'):
        tpath = os.path.join(ROOT, 'profiles', tname)
        if args.config == 'waveeq' and args.precision == 'bf16' and cfg['batch'] == 128 and os.path.exists(tpath):
            traffic = json.load(open(tpath))
            traffic['_source'] = ('static: profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per launch, collected in '
                                  'separate profiling runs of this command)' % tname)
            break
    roof, others = rooflines(res, args.precision, traffic)
    launch = ('hipGraph replay (per-kernel roofline timings: HIP events around every launch of %d EAGER steps run after the timed '
              'regions; they cannot be recorded inside a replay)' % res['sampled']) if res['use_graph'] else 'eager'
    out = {
        'metric': 'training frames/sec (seq x nt_pred)', 'value': round(frames / (ms * 1e-3), 1), 'unit': 'frames/s',
        'n_gpus': rk.world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 4),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.precision, 'data': 'synthetic',
        'repeats': len(res['ms_all']), 'ms_per_step_all': res['ms_all'],
        'config': {'workload': workload_text(args.config, cfg),
                   'global_batch': rk.world * cfg['batch'], 'parallelism': f'dp{rk.world}',
                   'grad_allreduce': ('none (1 rank)' if not rk.ddp else
                                      ('%s buckets over %s' % ('bf16' if res['comm_bf16'] else 'fp32',
                                                               'RCCL' if rk.backend == 'nccl' else 'gloo (ranks share one GPU: test mode)'))),
                   'optimizer': 'Adam (vs_adam_multi, one HIP launch)', 'launch': launch,
                   'timing': 'median of %d regions of %d steps' % (len(res['ms_all']), args.steps),
                   'final_loss': round(res['loss'], 5)},
        'roofline': roof, 'roofline_others': others,
    }
    if res['scaler']:
        out['config']['loss_scaling'] = res['scaler']
    extra = args.extra_configs
    if extra is None:
        extra = EXTRA_DEFAULT if (rk.world == 1 and args.config == 'waveeq' and not rk.ddp and args.batch is None) else 'none'
    if extra != 'none' and rk.world == 1:
        out['configs'] = {}
        for name in [e for e in extra.split(',') if e]:
            st, wu = EXTRA_STEPS.get(name, (5, 2))
            try:
                r = run_workload(name, args, rk, st, wu, 3)
            except Exception as e:          # one workload failing must not take the headline line down; it is reported
                out['configs'][name] = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
                continue
            c = r['cfg']
            rf, oth = rooflines(r, args.precision, top=4)
            out['configs'][name] = {'workload': workload_text(name, c), 'ms_per_step': round(r['ms'], 4),
                                    'value': round(c['batch'] * c['nt_pred'] / (r['ms'] * 1e-3), 1), 'unit': 'frames/s',
                                    'steps': st, 'warmup': wu, 'ms_per_step_all': r['ms_all'], 'dtype': args.precision,
                                    'final_loss': round(r['loss'], 5), 'roofline': rf, 'roofline_others': oth}
    if rk.world == 1 and not args.no_cpu_baseline:
        steps = args.cpu_steps or (5 if args.config in ('waveeq', 'mnist_b16') else 2)
        out['cpu_baseline'] = cpu_baseline(cfg, steps)
    os.write(result_fd, (json.dumps(out) + '\n').encode())
    rk.close()


if __name__ == '__main__':
    main()
