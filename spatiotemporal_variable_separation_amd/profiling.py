"""Kernel groups for roofline bookkeeping: the link between what `ops.py` accounts for per call (algorithmic FLOPs / bytes under a
family label such as `vs_conv3_band:fwd<bf16>`) and what a profiler sees (kernel symbols of libvarsep_hip.so).

One Python-level call may launch several kernels (a column-matrix convolution = gather + GEMM + split-K reduce) and one GEMM kernel
template serves several call sites, so the two views meet at the level of GROUPS: a group names the call-site families whose work it
carries and the kernel symbols that execute it.  `bench.py` takes a group's algorithmic FLOPs / bytes per step from the live accounting
of `ops.profile_collect()` and its duration either from live HIP events (eager launches) or from the `rocprofv3 --kernel-trace --stats`
table of the same command (the replayed hipGraph step: `tools/replay_stats.py` -> `profiles/<round>_<workload>_<dtype>_replay.json`);
`tools/pmc_traffic.py` uses the same table for the PMC traffic per launch.  Everything is reproducible from the committed CSV: a group's
time per step = sum of TotalDurationNs over the symbols matching its regex / steps executed (calls of `step_increment_kernel`)."""
import hashlib
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))


def source_sha():
    """sha256 (first 16 hex digits) over the kernel sources and EVERY Python file of the package: which kernels a step launches is decided
    in `ops.py` / `functional.py` but also in `train.py`, `optim.py`, `parallel.py` and `networks/*` (advisor finding, round 4: a swap of
    cat + cast for `vs_copy2d_pair` lived in `train.py` alone).  `tools/replay_stats.py` / `tools/pmc_traffic.py` / `tools/pmc_util.py` store
    it in every table they write under `profiles/`; `bench.py` uses a committed table only while it still matches and says `"stale": true`
    otherwise."""
    h = hashlib.sha256()
    files = sorted(os.path.join(_PKG, 'csrc', f) for f in os.listdir(os.path.join(_PKG, 'csrc')) if f.endswith(('.hip', '.h')))
    for root, dirs, names in sorted(os.walk(_PKG)):
        dirs[:] = sorted(d for d in dirs if d not in ('__pycache__', 'build', 'csrc'))
        files += [os.path.join(root, f) for f in sorted(names) if f.endswith('.py')]
    for f in files:
        h.update(os.path.relpath(f, _PKG).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


# (group, regex over ops.py family labels, regex over kernel symbols, bound)
GROUPS = [
    ('vs_gemm_adam', r'^vs_gemm_adam<', r'gemm_mid_kernel<\d, \d, \d, \w+, \d+, true>', 'hbm'),
    ('vs_mlp_rollout_fwd', r'^vs_mlp_rollout_fwd<', r'rollout_(ws|fwd)_kernel<\d+, true|rollout_fwd_kernel', 'mfma'),
    ('vs_mlp_rollout_bwd', r'^vs_mlp_rollout_bwd<', r'rollout_ws_kernel<\d+, false|rollout_bwd_kernel', 'mfma'),
    ('vs_conv_k4s2', r'^vs_conv_k4s2:|^vs_space_to_depth2', r'conv3_band_kernel<\d, \d+, \d+, \d, 1>|conv3_band2_kernel<\d, \d+, \d, 1, \d+>|wgrad3_band_kernel<\d, \d+, \d, 1(, \d)?>|wgrad2_band_kernel<\d, \d+, 1>|space_to_depth2_kernel|'
     r'::k4s2_\w+_kernel', 'mfma'),
    ('vs_conv3_img16_bn', r'^vs_conv3_img16_bn:', r'conv3_img16_bn_kernel|exchange_epoch_advance', 'mfma'),
    ('vs_conv3_img16', r'^vs_conv3_img16:', r'conv3_img16_kernel|(?<!grouped_)slab_sum_kernel', 'mfma'),
    ('vs_conv3_band', r'^vs_conv3_band:', r'conv3_band2?_kernel', 'mfma'),
    ('vs_conv3_wgrad_band', r'^vs_conv3_wgrad_band<', r'wgrad3_band_kernel|wgrad2_band_kernel|wgrad_slab_finish_kernel|slab_sum_grouped_kernel', 'mfma'),
    ('vs_convT_tap', r'^vs_convT_tap:', r'convt_k4s2_tap_kernel', 'mfma'),
    ('vs_conv3_tap', r'^vs_conv3_tap:', r'conv_k3s1_tap_kernel', 'mfma'),
    # dense GEMMs and the convolutions that run as (gather +) GEMM (+ split-K reduce): one pool of GEMM kernels serves them all
    ('vs_gemm+cols', r'^vs_gemm<|^vs_conv_cols:|^vs_convT_cols:', r'gemm_kernel<|gemm_glds_kernel<|gemm_big_kernel<|gemm_p8_kernel<|gemm_mid_kernel<\d, \d, \d, \w+, \d+, false>|'
     r'splitk_reduce_kernel|im2col_|convt_k4s2_small_kernel|gather_small_s1_kernel|gather_rowdot_kernel', 'mfma'),
    ('vs_bn_small', r'^vs_bn_fwd_small|^vs_bn_bwd_small', r'bn_fwd_small|bn_bwd_small', 'hbm'),
    # first / last layers (1..8 channels on the image side): one VALU pass over the many-channel map, no column matrix
    ('vs_conv_thin', r'^vs_conv_thin:', r'thin_\w+_kernel', 'hbm'),
    ('vs_bn', r'^vs_bn_stats|^vs_bn_act_fwd|^vs_bn_act_bwd|^vs_chan_sum|^vs_bn_fwd_slab',
     r'bn_stats_kernel|bn_act_fwd_kernel|bn_bwd_reduce|bn_bwd_apply|bn_running|bn_from_sums|chan_sum|group_sum2_kernel|bn_fwd_slab|bn_bwd_slab|bn_from_parts', 'hbm'),
    # resampling / joins around the convolutions (VGG pooling and nearest upsampling, the decoder's code broadcast + skip join, stand-alone activations)
    ('vs_resample+join', r'^vs_upsample|^vs_maxpool|^vs_cat_bcast|^vs_act_', r'upsample_\w+_kernel|maxpool_\w+_kernel|cat_bcast_\w+_kernel|act_fwd_kernel|act_bwd_kernel|'
     r'space_to_depth2', 'hbm'),
    # operand upkeep inside the recording: 16-bit / packed weight copies after the update, accumulator clears, dtype casts, strided copies
    ('vs_operand_upkeep', r'^vs_pack|^vs_zero|^vs_cast|^vs_copy2d', r'tap_pack_kernel|pack_weight_kernel|pack_multi_kernel|conv3_img16_pack|vs_zero_kernel|cast_kernel|'
     r'copy2d_\w*kernel|step_increment_kernel', 'hbm'),
    ('vs_adam_multi', r'^vs_adam_multi', r'adam_multi_kernel', 'hbm'),
    ('vs_train_losses', r'^vs_train_losses|^vs_frames_sse', r'train_losses_\w+_kernel|frames_sse|code_losses_\w+_kernel|frame_loss_finish_kernel', 'hbm'),
    ('vs_colsum_multi', r'^vs_colsum_multi', r'colsum_multi_kernel', 'hbm'),
    ('vs_mix_codes', r'^vs_mix_codes', r'mix_codes_\w+_kernel', 'hbm'),
    ('at::native (torch elementwise / cat / reduce)', r'^$', r'^void at::native|^at::native|__amd_rocclr', 'hbm'),
]
_COMPILED = [(g, re.compile(f), re.compile(k), b) for g, f, k, b in GROUPS]


def group_of_family(label):
    for g, f, _, _ in _COMPILED:
        if f.search(label):
            return g
    return None


def group_of_kernel(symbol):
    for g, _, k, _ in _COMPILED:
        if k.search(symbol):
            return g
    return None


def bound_of(group):
    for g, _, _, b in _COMPILED:
        if g == group:
            return b
    return 'hbm'


def replay_table(stats_rows, step_kernel='step_increment'):
    """rocprofv3 kernel-stats rows (dicts with Name / Calls / TotalDurationNs) -> (steps executed, {group: {'us_per_step',
    'launches_per_step', 'avg_launch_us'}}, unassigned us per step)."""
    def name(r):
        return r.get('Name') or r.get('KernelName')

    def calls(r):
        return int(r.get('Calls') or r.get('Count'))

    def total(r):
        return float(r.get('TotalDurationNs') or r.get('TotalDuration(ns)') or r['TotalDuration'])
    steps = max([calls(r) for r in stats_rows if step_kernel in name(r)] or [1])
    out, rest = {}, 0.0
    for r in stats_rows:
        g = group_of_kernel(name(r))
        if g is None:
            rest += total(r)
            continue
        e = out.setdefault(g, {'ns': 0.0, 'calls': 0})
        e['ns'] += total(r)
        e['calls'] += calls(r)
    table = {g: {'us_per_step': e['ns'] / steps / 1e3, 'launches_per_step': e['calls'] / steps, 'avg_launch_us': e['ns'] / e['calls'] / 1e3}
             for g, e in out.items()}
    return steps, table, rest / steps / 1e3


def hip_graph_counts(raw_graph, dot_path=None):
    """{'nodes', 'kernel_nodes', 'edges', 'roots'} of a captured hipGraph_t (an integer handle: torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph());
    `dot_path`: additionally hipGraphDebugDotPrint there.  Measurement only (train.GraphedStep.graph_stats, bench.py, tests)."""
    import ctypes
    hip = ctypes.CDLL('libamdhip64.so')
    g = ctypes.c_void_p(int(raw_graph))
    n = ctypes.c_size_t(0)
    if hip.hipGraphGetNodes(g, None, ctypes.byref(n)) != 0:
        return None
    nodes = (ctypes.c_void_p * max(n.value, 1))()
    hip.hipGraphGetNodes(g, nodes, ctypes.byref(n))
    kernels = 0
    for i in range(n.value):
        t = ctypes.c_int(-1)
        hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t))
        kernels += int(t.value == 0)                      # hipGraphNodeTypeKernel
    e = ctypes.c_size_t(0)
    hip.hipGraphGetEdges(g, None, None, ctypes.byref(e))
    r = ctypes.c_size_t(0)
    hip.hipGraphGetRootNodes(g, None, ctypes.byref(r))
    if dot_path:
        hip.hipGraphDebugDotPrint(g, dot_path.encode(), ctypes.c_uint(0))
    return {'nodes': int(n.value), 'kernel_nodes': kernels, 'edges': int(e.value), 'roots': int(r.value)}
