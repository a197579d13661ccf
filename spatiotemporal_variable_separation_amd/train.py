"""Training loop and losses of the hot path (reference: train.py:38-175); same function names and arguments."""
import os

import numpy as np
import torch
import torch.nn.functional as F

from .utils.helper import save


def zero_order_loss(s_code_old, s_code_new, skipco):
    """Mean squared difference between the spatial codes of the first and last windows (train.py:38-42)."""
    if skipco:
        s_code_old = torch.cat([s_code_old[0].flatten().float()] + [x.flatten().float() for x in s_code_old[1]])
        s_code_new = torch.cat([s_code_new[0].flatten().float()] + [x.flatten().float() for x in s_code_new[1]])
    return (s_code_old - s_code_new).pow(2).mean()


def ae_loss(cond, target, sep_net, nt_cond, offset, skipco, t_random=None):
    """Auto-encoding loss (train.py:45-88).  Returns (loss, s_code_new, s_code_old).

    `t_random` is drawn from the global NumPy RNG exactly like the reference unless given (additive argument)."""
    full_data = torch.cat([cond, target], dim=1)
    data_new = full_data[:, -nt_cond:]
    data_old = full_data[:, :nt_cond]
    s_code_old = sep_net.Es(data_old, return_skip=skipco)
    s_code_new = sep_net.Es(data_new, return_skip=skipco)
    if t_random is None:
        if offset == 0:
            t_random = np.random.randint(nt_cond, full_data.size(1))
        else:
            t_random = np.random.randint(nt_cond, full_data.size(1) + 1)
    if isinstance(t_random, torch.Tensor):
        # one int32 element ON THE DEVICE (GraphedStep): the window and the supervision frame are cut out by a kernel that reads it,
        # so the step can be recorded once and replayed while the random window moves
        window, supervision_data = _device_window(full_data, t_random, nt_cond, offset)
    else:
        window, supervision_data = full_data[:, t_random - nt_cond:t_random], full_data[:, t_random - offset]
    t_code_random = sep_net.Et(window)
    if skipco:
        reconstruction = sep_net.decoder(s_code_old[0], t_code_random, skip=s_code_old[1])
    else:
        reconstruction = sep_net.decoder(s_code_old, t_code_random)
    loss = F.mse_loss(supervision_data, reconstruction, reduction='mean')
    return loss, s_code_new, s_code_old


def _encode_pair(enc, xa, xb, return_skip):
    """enc(xa), enc(xb) as one batch of two call groups (`forward(..., groups=2)`); outputs split with unbind (one stack kernel in
    backward)."""
    B = xa.shape[0]
    out = enc(torch.cat([xa, xb], dim=0), return_skip=return_skip, groups=2)

    def halves(t):
        return t.view((2, B) + tuple(t.shape[1:])).unbind(0)
    if return_skip:
        code, skips = out
        ca, cb = halves(code)
        sk = [halves(sx) for sx in skips]
        return (ca, [p[0] for p in sk]), (cb, [p[1] for p in sk])
    return halves(out)


def _device_window(full_data, t_dev, nt_cond, offset):
    """full_data[:, t - nt_cond : t] and full_data[:, t - offset] for t = t_dev[0] read on the device (train.py:72-87)."""
    from . import ops
    B, T = full_data.shape[0], full_data.shape[1]
    flat = full_data.reshape(B, T, -1)
    if not flat.is_contiguous():
        flat = flat.contiguous()
    D = flat.shape[2]
    window = torch.empty((B, nt_cond * D), dtype=flat.dtype, device=flat.device)
    ops.copy2d(flat, B, nt_cond * D, T * D, window, nt_cond * D, col_offset_dev=t_dev, col_offset_scale=D, src_elem_offset=-nt_cond * D)
    frame = torch.empty((B, D), dtype=flat.dtype, device=flat.device)
    ops.copy2d(flat, B, D, T * D, frame, D, col_offset_dev=t_dev, col_offset_scale=D, src_elem_offset=-offset * D)
    return window.view((B, nt_cond) + tuple(full_data.shape[2:])), frame.view((B,) + tuple(full_data.shape[2:]))


_frame_index_cache = {}


def _frame_index(ae_frame, first_forecast, n, device, n_frames):
    """Device-resident [ae target frame, forecast target frames...] index vector.  One table holding the row for every
    possible auto-encoding frame is built once; picking a row is a view, so the training step never does a host-to-device
    copy (which would synchronise the stream) when the random window moves."""
    key = (first_forecast, n, n_frames, str(device))
    table = _frame_index_cache.get(key)
    if table is None:
        rows = [[a] + list(range(first_forecast, first_forecast + n)) for a in range(n_frames)]
        table = torch.tensor(rows, dtype=torch.int32, device=device)
        _frame_index_cache[key] = table
    return table if ae_frame is None else table[ae_frame]


def _mlp_family(sep_net):
    from .networks.mlp_encdec import MLPEncoder, MLPDecoder
    from .networks.resnet import MLPResnet
    from .networks.utils import ConstantS
    return (getattr(sep_net, 'fused', False) and not sep_net.skipco and isinstance(sep_net.Et, MLPEncoder)
            and isinstance(sep_net.Es, (MLPEncoder, ConstantS)) and isinstance(sep_net.decoder, MLPDecoder)
            and isinstance(sep_net.t_resnet, MLPResnet))


def _compute_losses_mlp_batched(cond, target, sep_net, nt_cond, nt_pred, offset, lamb_ae, lamb_s, lamb_t, lamb_pred,
                                average_tloss, t_random, full_data=None, frames_unused=False):
    """Same arithmetic as the generic path below, with the calls that share weights batched along the row axis:
    E_s on [first window; last window], E_t on [random window; conditioning window], D on [the auto-encoding row
    block; every rollout step].  The MLP family has no BatchNorm, so stacking rows is exact; it halves the number of
    weight-streaming GEMMs (E_s/E_t first layers are 98 MB of weights each) and leaves one gradient per parameter
    (no accumulation passes).

    `t_random` may be a python int (drawn on the host like the reference) or an int32 CUDA tensor of one element: in that
    case every use of it happens on the device (window gather with a device-side offset, target frame picked inside the loss kernels),
    so the whole step can be recorded once into a hipGraph and replayed while the random window moves."""
    from .networks.utils import ConstantS
    from . import functional as VF, ops
    if full_data is None:                            # GraphedStep keeps cond/target as views of one [B, T, ...] buffer already
        full_data = torch.cat([cond, target], dim=1)
    B, T = full_data.shape[0], full_data.shape[1]
    flat = full_data.reshape(B, T, -1)
    D = flat.shape[2]
    on_device = isinstance(t_random, torch.Tensor)
    if t_random is None:
        t_random = np.random.randint(nt_cond, T) if offset == 0 else np.random.randint(nt_cond, T + 1)

    step_start = None
    if (VF.side_streams_enabled() and flat.is_cuda and not isinstance(sep_net.Es, ConstantS) and os.environ.get('VARSEP_ES_EARLY', '0') in ('1', '2')):
        step_start = torch.cuda.Event()
        step_start.record()

    def window(end):
        return flat[:, end - nt_cond:end].reshape(B, -1)

    def spatial_codes():
        if isinstance(sep_net.Es, ConstantS):
            return sep_net.Es(full_data[:, :nt_cond]), sep_net.Es(full_data[:, -nt_cond:])
        with VF.on_chain_forward_stream():           # (the input is built where the chain's forward launches go: see `step_start` below)
            if on_device:
                # [first window; last window] in the compute type by ONE kernel (like E_t's input below) instead of a concatenation + a cast
                x_es = torch.empty((2 * B, nt_cond * D), dtype=VF.compute_dtype(), device=flat.device)
                ops.copy2d_pair(flat, B, nt_cond * D, T * D, x_es, nt_cond * D, None, 0, 0, (T - nt_cond) * D)
            else:
                x_es = torch.cat([window(nt_cond), window(T)], dim=0)
        s_both = sep_net.Es.mlp(x_es)
        return s_both.view(2, B, -1).unbind(0)       # unbind: its gradient is ONE stack kernel (two slices: fill+copy each, then add)

    if (VF.side_streams_enabled() and torch.is_grad_enabled() and VF.compute_dtype() != torch.float32
            and os.environ.get('VARSEP_PREPACK_EARLY', '0') == '1' and hasattr(sep_net.t_resnet, 'prepack')):
        # the integrator's weight packs depend on the weights only: on the integrator's stream, ahead of everything.  Measured and NOT the
        # default: 1.451-1.459 vs 1.450 ms (the 8 us launch leaves the critical path, the replayed step does not get shorter)
        main, side = torch.cuda.current_stream(), VF._side_stream('rollout')
        side.wait_stream(main)
        with torch.cuda.stream(side):
            sep_net.t_resnet.prepack()
    if on_device:
        # rows [0, B): full[:, t - nt_cond : t] cut out by a kernel that reads t on the device; rows [B, 2B): the conditioning window
        x_et = torch.empty((2 * B, nt_cond * D), dtype=VF.compute_dtype(), device=flat.device)
        ops.copy2d_pair(flat, B, nt_cond * D, T * D, x_et, nt_cond * D, t_random, D, -nt_cond * D, 0)
    else:
        x_et = torch.cat([window(t_random), window(nt_cond)], dim=0)
    t_both = sep_net.Et.mlp(x_et)
    t_rand, t0 = t_both.view(2, B, -1).unbind(0)

    n = nt_pred + offset
    if VF.side_streams_enabled():
        # the rollout (32 workgroups) runs on its own stream while E_s uses the rest of the chip; autograd replays the same
        # stream assignment in backward, where the rollout overlaps the encoder/decoder weight gradients
        main, side = torch.cuda.current_stream(), VF._side_stream('rollout')
        side.wait_stream(main)
        t0.record_stream(side)
        with torch.cuda.stream(side):
            t_codes, _ = sep_net.t_resnet.rollout(t0, n)
        if step_start is not None:
            # E_s on a stream of its own that depends on the START of the step only: in the replayed recording its first layer (a 250-workgroup
            # GEMM that streams 49 MB of weights) runs beside E_t's small layers instead of under the integrator's kernel, whose 192 resident
            # workgroups leave it 64 CUs (timeline of round 6: 84 us there against 28 us alone, and E_s's chain -- 200 us -- outlasted the
            # integrator's 148).  Only the FORWARD launches move (functional.chain_forward_stream): the chain's autograd node is created under
            # the main stream, so backward runs where it ran before -- a sixth concurrent branch in backward is the runtime's scheduling cliff
            # (2.50 ms per step, measured with the whole chain, backward included, on the new stream).
            # MEASURED AND NOT THE DEFAULT (VARSEP_ES_EARLY=1: a stream of its own, 2: the last gradient lane's stream, idle in forward), same box,
            # alternating with the default: '1' 2.64 / 2.64 / 2.67 ms -- a sixth stream in the recording is the same cliff as six hardware queues
            # (profiles/r06_queues.md) even though only three branches are ever concurrent in forward; '2' 1.178 / 1.174 / 1.175 against
            # 1.147 / 1.152 / 1.164 -- E_s's first layer then competes with E_t's chain, which the integrator (the critical path) waits for.
            es = VF._lane_stream(VF.N_LANES - 1) if os.environ.get('VARSEP_ES_EARLY', '0') == '2' else VF._side_stream('es')
            es.wait_event(step_start)
            with VF.chain_forward_stream(es):
                s_old, s_new = spatial_codes()
            main.wait_stream(es)
        else:
            s_old, s_new = spatial_codes()
        main.wait_stream(side)
        t_codes.record_stream(main)
    else:
        s_old, s_new = spatial_codes()
        t_codes, _ = sep_net.t_resnet.rollout(t0, n)
    handoff = VF.GradHandoff() if os.environ.get('VARSEP_LOSS_HANDOFF', '1') == '1' else None
    # both frame losses in one fused pass: frame 0 vs full[:, t_random - offset], frame g vs full[:, fo + g - 1]
    fo = nt_cond if offset == 0 else 0
    # t_codes[:, 0] IS t0 (the rollout copies its input there), so the regulariser reads the encoder output directly
    no_s = isinstance(sep_net.Es, ConstantS)
    s_old_f = None if no_s else s_old.reshape(B, -1).float().contiguous()
    s_new_f = None if no_s else s_new.reshape(B, -1).float().contiguous()
    t0_f = t0.reshape(B, -1).float().contiguous()
    flat_c = flat.contiguous()
    up = VF.promised_loss_gradient()
    if (frames_unused and handoff is not None and on_device and up is not None and torch.is_grad_enabled()
            and VF.compute_dtype() != torch.float32 and os.environ.get('VARSEP_FUSE_FRAME_LOSS', '1') == '1'):
        # recorded step (nobody reads the frames): the decoder's last GEMM compares them with their targets in its epilogue and
        # writes the gradient of its pre-activation; neither the fp32 frame stack nor a loss pass over it exists (functional.MLPChain)
        handoff.fuse = dict(full=flat_c, idx=(t_random, offset, fo), G=1 + n, s_old=s_old_f, s_new=s_new_f, t0=t0_f,
                            lambdas=(lamb_ae, lamb_s, lamb_t, lamb_pred), average=average_tloss, up=up)
    frames = sep_net.decoder.decode_rollout(s_old, t_rand, t_codes, handoff=handoff)                      # [B, 1+n, ...]
    forecasts = None if (handoff is not None and handoff.fused is not None) else frames[:, 1:]
    if on_device:
        idx = (t_random, offset, fo)                 # resolved inside the loss kernels: no index tensor to build per step
    else:
        idx = _frame_index(int(t_random) - offset, fo, n, frames.device, T + 1)
    total_loss, ae_loss_value, spatial_ode_loss, forecast_loss, t_reg = VF.TrainLosses.apply(
        frames.reshape(B, 1 + n, -1), flat_c, idx, s_old_f, s_new_f, t0_f, (lamb_ae, lamb_s, lamb_t, lamb_pred), average_tloss, handoff)
    terms = {'ae': ae_loss_value, 'zero': spatial_ode_loss, 'pred': forecast_loss, 't_reg': t_reg}
    if VF.side_streams_enabled() and os.environ.get('VARSEP_HOLD_WGRADS', '1') == '1':
        # backward: collect the decoder's and E_s's weight gradients and launch them under the integrator's backward kernel
        VF.hold_deferred(True)
    return total_loss, terms, forecasts, t_codes


# Stream capture in 'thread_local' error mode: with a process group alive, ProcessGroupNCCL's watchdog thread polls its work events
# (hipEventQuery) at any time; in the default 'global' mode such a call from ANOTHER thread while this thread captures is an error that
# terminates the process ("operation not permitted when stream is capturing" -- seen in 3 of 6 runs of the data-parallel GPU tests).
_CAPTURE_MODE = 'thread_local'


class GraphedStep:
    """One whole optimisation step (losses, backward, Adam) recorded into a hipGraph and replayed.  MLP family: the batched step
    with side streams; conv families: the reference's call structure (same-box eager -> replay: SST 72.5 -> 68.5 ms, MNIST B=16
    7.5 -> 6.1 ms; independent of the host's launch rate).

    The WaveEq step launches ~110 kernels of 2-800 us; issued one by one from Python the host needs ~3.9 ms per step, more than
    the GPU needs to execute them, so the eager loop is host-bound.  Stream capture (torch.cuda.CUDAGraph = hipGraph on ROCm)
    records the kernels the C-ABI library launches on the capture stream together with torch's own; the only per-step host
    inputs -- the batch and the random window end `t_random` (train.py:72-75) -- enter through static device buffers, and
    every use of `t_random` inside the step is device-side.  The optimizer is optim.Adam (HIP, always recordable) or
    torch.optim.Adam(capturable=True); a learning-rate change by a scheduler triggers a re-recording."""

    def __init__(self, sep_net, optimizer, cond, target, nt_cond, nt_pred, offset, lambdas, average_tloss=False, warmup=3,
                 side_streams=True, grad_sync=None, scaler=None, keep_warmup_updates=False):
        assert cond.is_cuda, 'GraphedStep records a hipGraph: the batch must be on the GPU'
        self.mlp = _mlp_family(sep_net)
        self.skipco = bool(getattr(sep_net, 'skipco', False))
        # deferred weight gradients require that nothing reads a gradient before join_side_streams(); with a reducer the Linear
        # chains therefore write straight into its flat buckets (VF.set_grad_outputs) instead of going through autograd's `+=`
        # side streams and direct gradient destinations need ONE gradient per parameter and step: true for the batched MLP-family
        # step only (a conv family calls E_s twice, its Linear layers get two contributions)
        self.side_streams = side_streams and self.mlp and os.environ.get('VARSEP_GRAPH_SIDE', '1') == '1'
        check_optimizer(optimizer, for_graph=True)
        self.net, self.opt, self.sync, self.scaler = sep_net, optimizer, grad_sync, scaler
        # conv families under a reducer: convolution / BatchNorm gradients accumulate straight into the reducer's bucket views
        # (functional.set_conv_grad_outputs), so gradient folding and the batched weight gradients stay on and the data-parallel
        # step runs the kernels of the single-GPU step
        self._conv_sinks = conv_gradient_sinks(sep_net, grad_sync) if (grad_sync is not None and not self.mlp) else None
        # ... and its backward pass is recorded in TWO segments split at the decoder's inputs when the reducer keeps the decoder's gradients
        # in buckets of their own (GradAllReducer(early=decoder parameters)): their all-reduce is issued between the two replays and travels
        # on the comm stream while the integrator's / encoders' backward kernels run (VARSEP_GRAPH_SEGMENTS=0: one segment, as in round 3)
        self.segmented = (grad_sync is not None and not self.mlp and bool(getattr(grad_sync, 'early_buckets', None))
                          and getattr(sep_net, 'fused', False) and getattr(sep_net.Es, 'call_groups', False) and getattr(sep_net.Et, 'call_groups', False)
                          and hasattr(sep_net.decoder, 'decode_sequence') and os.environ.get('VARSEP_GRAPH_SEGMENTS', '1') == '1'
                          and os.environ.get('VARSEP_ENCODER_PAIRS', '1') == '1')
        # single process, conv family: the same destinations WITHOUT a reducer (one flat fp32 buffer, zeroed at the start of the step) --
        # a weight gradient that accumulates into a tensor autograd never sees can run on a gradient stream beside the input-gradient /
        # BatchNorm chain it does not feed (functional._conv_weight_grad); joined before the optimizer.  VARSEP_CONV_WGRAD_SIDE=1 turns it on;
        # measured and NOT the default: with 8 hardware queues the replayed steps double (TaxiBJ 8.74 -> 17.2 ms, Moving-MNIST 7.0 -> 11.9, SST
        # 21.0 -> 43: every launch of the ~700-kernel main chain slows down while a second queue is busy), with 4 / 2 queues nothing overlaps
        # and the zero fill + accumulate cost 4 % (9.07 vs 8.74 ms).
        self._local_flat = None
        if grad_sync is None and not self.mlp and side_streams and os.environ.get('VARSEP_CONV_WGRAD_SIDE', '0') == '1':
            self._local_flat, self._conv_sinks = local_conv_gradient_sinks(sep_net)
            self.side_streams = True
        self._one = torch.ones((), dtype=torch.float32, device=cond.device)
        if grad_sync is not None and self.mlp and getattr(grad_sync, 'lowp_views', None) and hasattr(optimizer, 'step_subset'):
            # the chains' weight gradients are produced, averaged and consumed as bf16 wire images (parallel.GradAllReducer)
            from . import functional as VF
            grad_sync.direct_lowp = True
            VF.set_lowp_gradients({p: grad_sync.lowp_views[id(p)] for p in grad_sync.params if id(p) in grad_sync.lowp_views})
        # sharded optimizer (parallel.GradAllReducer(shard_direct=True)): reduce-scatter of the chains' bf16 wire gradients, Adam on this rank's
        # slice, all-gather of the 16-bit operand copies.  Needs the wire shortcut above, a 16-bit compute type (the copies ARE the weights
        # forward / backward read) and no loss scaling; sums instead of averages: the loss gradient is seeded with 1 / world_size
        from . import functional as VF
        self.sharded = bool(grad_sync is not None and getattr(grad_sync, 'shard', False) and getattr(grad_sync, 'direct_lowp', False)
                            and scaler is None and hasattr(optimizer, 'step_ranges') and VF.compute_dtype() != torch.float32)
        if grad_sync is not None and getattr(grad_sync, 'shard', False) and not self.sharded:
            grad_sync.shard = False                  # (fp32 / fp16-with-scaler / foreign optimizer: the all-reduce path)
        if self.sharded:
            grad_sync.adopt_operand_copies(VF.compute_dtype())
        # under a reducer the recorded step seeds its loss gradient with 1 / world_size and the collectives SUM (GradAllReducer.presummed);
        # with loss scaling the scale tensor is the seed, so that path keeps ReduceOp.AVG.  VARSEP_DDP_PRESUM=0: averages as before
        self.presum = bool(grad_sync is not None and hasattr(grad_sync, 'presummed')
                           and (self.sharded or (scaler is None and os.environ.get('VARSEP_DDP_PRESUM', '1') == '1')))
        if self.presum:
            self._one = torch.full((), 1.0 / grad_sync.world_size, dtype=torch.float32, device=cond.device)
        enable_fused_update(optimizer, sep_net, grad_sync, scaler)
        enable_update_in_backward(optimizer, sep_net, grad_sync, scaler=scaler)
        self.args = (nt_cond, nt_pred, offset) + tuple(lambdas) + (average_tloss,)
        # static inputs of the recording: one [B, T, ...] buffer, cond / target are views of it (no concatenation per step)
        self.full = torch.cat([cond, target], dim=1).contiguous()
        self.cond, self.target = self.full[:, :cond.shape[1]], self.full[:, cond.shape[1]:]
        self.t_dev = torch.zeros(1, dtype=torch.int32, device=cond.device)
        self.T = cond.shape[1] + target.shape[1]
        self.nt_cond, self.offset = nt_cond, offset
        from .functional import flush_bn_call_counts as VF_flush
        # the warm-up steps exist to size workspaces / build caches before the capture; they must not train: parameters, optimizer
        # state, BatchNorm buffers, the loss scale and the NumPy stream (t_random) are put back afterwards, so the first replayed
        # step is step 1 of the reference loop on batch 0 (keep_warmup_updates=True keeps them, e.g. for throughput runs)
        snap = None if keep_warmup_updates else _TrainState(sep_net, optimizer, scaler)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._draw()
                self._fwd_bwd()
                self._reduce()
                self._opt_step()
                VF_flush()
        torch.cuda.current_stream().wait_stream(side)
        if snap is not None:
            snap.restore()
        self._capture()

    def _capture(self):
        from . import functional as VF
        from .optim import Adam as HipAdam
        self.steps_replayed = 0                      # (train() checks the exchange error word after each of a recording's first replays)
        params = list(self.net.parameters())
        if getattr(self, 'sharded', False):
            self.sync.sync_masters(self.opt)         # the operand copies are re-derived from the fp32 masters below: they must be complete
        if isinstance(self.opt, HipAdam):
            # restoring the warm-up snapshot bumped every parameter's version counter (and, for a re-recording after a learning-rate
            # change, the replays since the last recording moved the replay epoch): bring the 16-bit operand copies up to date NOW, or
            # the recording would contain a cast of every weight: measured +90 us per replayed WaveEq step.  ONLY valid because the HIP
            # Adam kernel rewrites the copies in the pass that updates the masters, so they stay current from then on.  Weight
            # pre-packs are NOT refreshed here: they are stale at this point, so the recording contains them, once per step.
            VF.refresh_shadows(params)
        else:
            # any other optimizer (torch.optim.Adam(capturable=True)) updates the fp32 masters only: the recording must CONTAIN the
            # casts, or every replay would read the weights frozen at capture time.  Mark every copy stale so that its first use
            # inside the capture records the cast (replayed at the start of every step, i.e. after the previous step's update)
            VF.invalidate_shadows(params)
        grad_sync = self.sync
        self._lrs = [g['lr'] for g in self.opt.param_groups]
        self._draw()
        self.graph = torch.cuda.CUDAGraph()
        self.graph_opt = None
        # VARSEP_GRAPH_STATS=1: keep the captured hipGraph_t so that graph_stats() can count its nodes and edges (VARSEP_GRAPH_DOT=<path>: and dump it)
        self._keep = os.environ.get('VARSEP_GRAPH_STATS') == '1' or bool(os.environ.get('VARSEP_GRAPH_DOT'))
        if self._keep:
            self.graph = torch.cuda.CUDAGraph(keep_graph=True)
        if grad_sync is None:
            from . import functional as VF
            VF.bn_counts_flushed_in_capture(True)        # SST: ~200 per-call `num_batches_tracked += 1` launches become one
            try:
                with torch.cuda.graph(self.graph, capture_error_mode=_CAPTURE_MODE):
                    self.loss = self._fwd_bwd()
                    self._opt_step()
                    VF.flush_bn_call_counts()
            finally:
                VF.bn_counts_flushed_in_capture(False)
        else:
            # data parallel: losses + backward into the reducer's flat gradient buckets in one graph, the bucket
            # all-reduces issued eagerly in between (4 RCCL calls at WaveEq size; no collective inside a capture), Adam in
            # a second graph
            from . import functional as VF
            VF.bn_counts_flushed_in_capture(True)
            self.graph2 = None
            try:
                if self.segmented:
                    with torch.cuda.graph(self.graph, capture_error_mode=_CAPTURE_MODE):
                        self.loss = self._fwd_bwd(segment=1)
                    # same memory pool: the second segment reads what the first one saved for it
                    self.graph2 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self.graph2, pool=self.graph.pool(), capture_error_mode=_CAPTURE_MODE):
                        self._fwd_bwd(segment=2)
                        VF.flush_bn_call_counts()
                else:
                    with torch.cuda.graph(self.graph, capture_error_mode=_CAPTURE_MODE):
                        self.loss = self._fwd_bwd()
                        VF.flush_bn_call_counts()
            finally:
                VF.bn_counts_flushed_in_capture(False)
            self._reduce()
            if self.sharded or (self.scaler is None and hasattr(self.opt, 'step_subset') and os.environ.get('VARSEP_ADAM_PER_BUCKET', '1') == '1'):
                # one Adam recording per all-reduce bucket: the update of bucket i runs while buckets i+1.. are still on the wire
                self.graph_opt = []
                for bi, (_, plist) in enumerate(grad_sync.buckets):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode=_CAPTURE_MODE):
                        if self.sharded:
                            self.opt.step_ranges(grad_sync.shard_ranges(bi))       # this rank's slice of the chains' weights
                            tail = grad_sync.tail_params(bi)
                            if tail:
                                self.opt.step_subset(tail)                         # biases, integrator: replicated
                        else:
                            self.opt.step_subset(plist)
                    self.graph_opt.append(g)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode=_CAPTURE_MODE):
                    self.opt.finish_step()
                self.graph_opt.append(g)
            else:
                self.graph_opt = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_opt, capture_error_mode=_CAPTURE_MODE):
                    self._opt_step()

    def graph_stats(self):
        """Node and edge counts of the (first) recording -- {'nodes', 'kernel_nodes', 'edges', 'roots'} -- read from the captured hipGraph_t
        (hipGraphGetNodes / hipGraphGetEdges / hipGraphNodeGetType); needs VARSEP_GRAPH_STATS=1 at capture time.  The host pays per node and
        per dependency edge of a replay, the device per dependent kernel boundary."""
        if not getattr(self, '_keep', False):
            return None
        from .profiling import hip_graph_counts
        return hip_graph_counts(self.graph.raw_cuda_graph(), os.environ.get('VARSEP_GRAPH_DOT'))

    def _opt_step(self):
        from . import functional as VF
        pending, self._pending_join = getattr(self, '_pending_join', None), None
        if getattr(self, 'sharded', False):
            # (warm-up steps: the recorded form is one graph per bucket, see _capture)
            for bi in range(len(self.sync.buckets)):
                self.opt.step_ranges(self.sync.shard_ranges(bi))
                tail = self.sync.tail_params(bi)
                if tail:
                    self.opt.step_subset(tail)
                self.sync.gather_operand_copies(bi)
            self.opt.finish_step()
            self.sync.wait_comm()
            self.sync.masters_dirty = True
            return
        if self.scaler is not None:
            self.scaler.step(self.opt)               # finite check, unscale inside the update, skip on overflow, scale update
        elif pending:
            # the trailing fused updates (vs_gemm_adam of the encoders' first layers) are still running on their lanes: the launch for
            # the remaining parameters runs beside them (disjoint parameters), the step counter both read moves after the full join
            self.opt.step(defer_increment=True)
            VF.finish_join(pending)
            self.opt.finish_step()
        else:
            self.opt.step()

    def _draw(self):
        hi = self.T if self.offset == 0 else self.T + 1
        # a fill kernel carries the value as a launch argument: no staging buffer the host could overwrite while an earlier
        # step's copy is still queued behind a 3 ms replay
        self.t_dev.fill_(int(np.random.randint(self.nt_cond, hi)))

    def _reduce(self):
        if self.sync is not None:
            was, self.sync.presummed = self.sync.presummed, getattr(self, 'presum', False)
            try:
                self.sync.reduce_all()
            finally:
                self.sync.presummed = was

    def _fwd_bwd(self, segment=None):
        """Losses + backward.  `segment` (conv family under a reducer with early buckets): 1 = forward and the part of backward that ends at
        the decoder's inputs (the decoder's gradients are complete afterwards), 2 = the rest of backward from those tensors' gradients."""
        nt_cond, nt_pred, offset, l_ae, l_s, l_t, l_pred, avg = self.args
        from . import functional as VF
        fold_was = VF.folding_repeated_gradients()
        if segment == 2:
            (pairs, total, up), self._cuts = self._cuts, None
            VF.set_conv_grad_outputs(self._conv_sinks)
            VF.fold_repeated_gradients(True, flush=False)
            try:
                backward_rest_segment(total, up, self.net, pairs)
            finally:
                VF.set_conv_grad_outputs(None)
                VF.fold_repeated_gradients(fold_was, flush=False)
            return None
        if self.sync is not None:
            self.sync.zero_buffers()
            if self.mlp:
                lowp = self.sync.lowp_views if self.sync.direct_lowp else {}
                VF.set_grad_outputs({p: lowp.get(id(p), p.grad) for p in self.sync.params})
            elif self._conv_sinks:
                VF.set_conv_grad_outputs(self._conv_sinks)
                VF.fold_repeated_gradients(True, flush=False)
        elif self._local_flat is not None:
            self._local_flat.zero_()
            for prm, view in self._conv_sinks.items():
                prm.grad = view
            sinks = {id(p) for p in self._conv_sinks}
            for group in self.opt.param_groups:
                for prm in group['params']:
                    if id(prm) not in sinks:
                        prm.grad = None
            VF.set_conv_grad_outputs(self._conv_sinks)
            VF.fold_repeated_gradients(True, flush=False)
        else:
            self.opt.zero_grad(set_to_none=True)
        VF.enable_side_streams(self.side_streams)
        VF.collect_cuts(segment == 1)
        up = self._one if self.scaler is None else self.scaler.scale_tensor()
        VF.promise_loss_gradient(up if (self.mlp and os.environ.get('VARSEP_LOSS_ONE_PASS', '1') == '1') else None)
        try:
            if self.mlp:
                total, _, _, _ = _compute_losses_mlp_batched(self.cond, self.target, self.net, nt_cond, nt_pred, offset, l_ae, l_s, l_t,
                                                             l_pred, avg, self.t_dev, full_data=self.full, frames_unused=True)
            else:                                    # conv families: the reference's call structure with a device-side window
                total, _, _, _ = compute_losses(self.cond, self.target, self.net, nt_cond, nt_pred, offset, self.skipco, l_ae, l_s,
                                                l_t, l_pred, avg, t_random=self.t_dev)
            # a resident 1.0 (no ones_like fill per step), or the loss scale of fp16 training (train.py:152 scaler.scale(loss))
            if segment == 1:
                self._cuts = (backward_decoder_segment(total, up, self.net), total, up)
            else:
                total.backward(up)
            from .optim import Adam as HipAdam
            partial = (self.sync is None and self.scaler is None and isinstance(self.opt, HipAdam)
                       and bool(getattr(self.opt, '_fused', None)) and VF.tail_fused_updates())
            self._pending_join = VF.join_side_streams(partial=partial)
        finally:
            VF.collect_cuts(False)
            VF.promise_loss_gradient(None)
            VF.enable_side_streams(False)
            VF.set_grad_outputs(None)
            if self._conv_sinks:
                VF.set_conv_grad_outputs(None)
                VF.fold_repeated_gradients(fold_was, flush=False)
        return total.detach()

    def step(self, cond=None, target=None):
        """Replay one optimisation step; new batches are copied into the static input buffers first."""
        if cond is not None:
            self.cond.copy_(cond, non_blocking=True)
            self.target.copy_(target, non_blocking=True)
        if [g['lr'] for g in self.opt.param_groups] != self._lrs:
            self._capture()                          # a scheduler moved the learning rate: it is a launch argument of the recording
        self._draw()
        self.graph.replay()
        self.steps_replayed += 1
        if self.sync is not None:
            was_presummed, self.sync.presummed = self.sync.presummed, getattr(self, 'presum', False)
        try:
            return self._finish_step()
        finally:
            if self.sync is not None:
                self.sync.presummed = was_presummed

    def _finish_step(self):
        early = {}
        if getattr(self, 'graph2', None) is not None:
            # the decoder's buckets go on the wire now; the second segment (integrator + encoders backward) runs beside them
            early = {bi: self.sync.reduce_bucket(bi) for bi in self.sync.early_buckets}
            self.graph2.replay()
        if isinstance(self.graph_opt, list):
            events = [early[bi] if bi in early else self.sync.reduce_bucket(bi) for bi in range(len(self.sync.buckets))]
            main = torch.cuda.current_stream()
            for bi, (ev, g) in enumerate(zip(events, self.graph_opt)):
                if ev is not None:
                    main.wait_event(ev)
                g.replay()
                if self.sharded:
                    self.sync.gather_operand_copies(bi)  # comm stream, beside the next bucket's update
            self.graph_opt[-1].replay()              # step counter
            if self.sharded:
                self.sync.wait_comm()                    # the next step's forward reads the gathered copies
                self.sync.masters_dirty = True
        elif self.graph_opt is not None:
            for bi in range(len(self.sync.buckets)):
                if bi not in early:
                    self.sync.reduce_bucket(bi)
            if self.sync._comm_stream is not None:
                torch.cuda.current_stream().wait_stream(self.sync._comm_stream)
            self.graph_opt.replay()
        # the replay changed parameters and BatchNorm statistics without moving a version counter: whatever eager code derives from them
        # next (evaluation between training steps, instrumented eager steps, a re-recording) must re-derive it
        from . import functional as VF
        VF.note_replay()
        return self.loss


def backward_decoder_segment(total, up, sep_net):
    """First segment of a two-segment backward pass (functional.cut / collect_cuts were on during the forward pass): the loss heads and the
    decoder.  Afterwards the decoder's parameter gradients are complete and every leaf that stood in for a decoder input holds its gradient.
    Returns the (original, leaf) pairs for `backward_rest_segment`."""
    from . import functional as VF
    pairs = VF.cut_pairs()
    assert pairs, 'two-segment backward: no tensor was cut at the decoder\'s inputs'
    heads = [p for p in sep_net.decoder.parameters() if p.requires_grad]
    torch.autograd.backward(total, up, inputs=heads + [leaf for _, leaf in pairs], retain_graph=True)
    return pairs


def backward_rest_segment(total, up, sep_net, pairs):
    """Second segment: the leaves' gradients enter the tensors they stood in for, and `total` contributes what reaches the encoders / the
    integrator without passing the decoder (the code regularisers of train.py:120-149); accumulates into every parameter outside the decoder."""
    dec = {id(p) for p in sep_net.decoder.parameters()}
    rest = [p for p in sep_net.parameters() if p.requires_grad and id(p) not in dec]
    live = [(o, leaf) for o, leaf in pairs if leaf.grad is not None]
    torch.autograd.backward([total] + [o for o, _ in live], [up] + [leaf.grad for _, leaf in live], inputs=rest)
    for _, leaf in pairs:
        leaf.grad = None


def local_conv_gradient_sinks(sep_net):
    """(flat fp32 buffer, {convolution / BatchNorm parameter: its view}) for functional.set_conv_grad_outputs in a single-process step: the
    gradients accumulate into the views (`p.grad` of these parameters), the buffer is zeroed once per step."""
    import torch.nn as nn
    from . import functional as VF
    from .networks.conv import ConvResBlock
    for blk in sep_net.modules():
        if isinstance(blk, ConvResBlock):
            VF.mark_repeated([m.weight for m in blk.modules() if isinstance(m, nn.Conv2d)])
            blk._marked = True
    params = VF.conv_parameters(sep_net)
    offs, n = [], 0
    for prm in params:
        offs.append(n)
        n += (prm.numel() + 3) // 4 * 4                  # 16-byte aligned views
    flat = torch.zeros((n,), dtype=torch.float32, device=params[0].device)
    return flat, {prm: flat[o:o + prm.numel()].view(prm.shape) for prm, o in zip(params, offs)}


def conv_gradient_sinks(sep_net, grad_sync):
    """{convolution / BatchNorm parameter: its view in the reducer's flat gradient buckets} for functional.set_conv_grad_outputs."""
    import torch.nn as nn
    from . import functional as VF
    owned = {id(p) for p in grad_sync.params}
    # weights whose gradient may be completed by the end-of-backward flush of the batched weight gradients (functional._DEFER_W:
    # stride-1 Conv2d): their buckets are not reduced from gradient hooks
    from .networks.conv import ConvResBlock
    for blk in sep_net.modules():
        if isinstance(blk, ConvResBlock):
            VF.mark_repeated([m.weight for m in blk.modules() if isinstance(m, nn.Conv2d)])
            blk._marked = True
    late = [m.weight for m in sep_net.modules() if isinstance(m, nn.Conv2d) and tuple(m.stride) == (1, 1) and id(m.weight) in owned
            and getattr(m.weight, '_vs_repeated', False)]
    if late and hasattr(grad_sync, 'hold_params'):
        grad_sync.hold_params(late)
    return {p: p.grad for p in VF.conv_parameters(sep_net) if id(p) in owned and p.grad is not None}


def chain_weight_parameters(sep_net):
    """The 2-D weights of the encoders' and the decoder's Linear chains of an MLP-family model: the parameters whose gradients
    the recorded data-parallel step can produce directly in the reducer's wire format (GradAllReducer(lowp_direct=...))."""
    out = []
    if not _mlp_family(sep_net):
        return out
    for mod in (sep_net.Es, sep_net.Et, sep_net.decoder):
        mlp = getattr(mod, 'mlp', None)
        if mlp is not None:
            out += [lin.weight for lin in mlp.linears()]
    return out


def rollout_weight_stacks(sep_net):
    """[[W1 of every block], [W2 ...], [W3 ...]] of an MLP-family integrator: the groups GradAllReducer(stacked=...) lays back to back so
    that functional.MLPRollout.backward writes each layer's batched weight gradient straight into the bucket."""
    if not _mlp_family(sep_net):
        return []
    per_block = [[lin.weight for lin in blk.mlp.linears()] for blk in sep_net.t_resnet.blocks]
    if not per_block or any(len(ws) != 3 for ws in per_block):
        return []
    return [[ws[l] for ws in per_block] for l in range(3)]


def shard_optimizer_default():
    """VARSEP_SHARD_OPT (default 1): MLP family under a bf16-wire reducer -> reduce-scatter + sharded Adam + all-gather of the operand copies."""
    return os.environ.get('VARSEP_SHARD_OPT', '1') == '1'


def enable_update_in_backward(optimizer, sep_net, grad_sync=None, force=False, scaler=None):
    """With the HIP Adam and no gradient all-reduce: update decoder + integrator + E_s on a side stream as soon as the last
    of their gradients is final, i.e. while backward is still in E_t (optim.Adam.overlap_with_backward).  One bucket on
    purpose: the update streams HBM at >5 TB/s and, launched earlier, would run beside the integrator's backward kernel,
    whose inter-workgroup exchange is latency-bound on the same memory fabric (measured: 204 -> 275 us)."""
    # Opt-in (VARSEP_ADAM_OVERLAP=1): on the WaveEq step it is a wash (1.898 vs 1.891 ms) -- the update and E_t's backward
    # (98 MB of fp32 weight gradient per encoder) compete for the same HBM bandwidth, both just run slower side by side.
    from .optim import Adam as HipAdam
    early_only = os.environ.get('VARSEP_ADAM_EARLY_BUCKET') in ('1', '2')
    if not force and os.environ.get('VARSEP_ADAM_OVERLAP') != '1' and not early_only:
        return
    if scaler is not None:
        # fp16 loss scaling: the update needs 1 / scale and the finite check of ALL gradients, which exist only after backward
        # (LossScaler.step); an update from a backward hook would apply scaled gradients and could not be skipped on overflow
        return
    if isinstance(optimizer, HipAdam) and grad_sync is None and not optimizer._buckets and len(optimizer.param_groups) == 1:
        owned = {id(p) for p in optimizer.param_groups[0]['params']}
        fused = {id(p) for p in getattr(optimizer, '_fused', [])}        # updated inside their weight-gradient GEMMs: in no bucket
        early = [p for m in (sep_net.decoder, sep_net.Es) for p in m.parameters() if id(p) in owned and id(p) not in fused]
        if early_only and not force:
            # decoder + E_s only: their gradients are complete once the held weight gradients have run (under the integrator's
            # backward kernel), the update joins that queue; the optimizer launch at the end of the step shrinks to E_t + integrator.
            # (The integrator's own gradients are recorded at the very end of backward -- functional.run_late -- after its hooks.)
            if early:
                optimizer.overlap_with_backward([early])
            return
        buckets = [early, [p for p in sep_net.t_resnet.parameters() if id(p) in owned and id(p) not in fused],
                   [p for p in sep_net.Et.parameters() if id(p) in owned and id(p) not in fused]]
        if sum(len(b) for b in buckets) + len(fused) == len(owned):
            optimizer.overlap_with_backward(buckets)


def enable_fused_update(optimizer, sep_net, grad_sync=None, scaler=None, min_numel=1 << 22):
    """Single GPU, HIP Adam, 16-bit compute, no loss scaling: the large Linear weights of an MLP-family model (WaveEq: the encoders'
    20480 x 1200 first layers and the decoder's 1200 x 4096 last layer, 54 M of the model's 60 M parameters) take their Adam step
    in the epilogue of their weight-gradient GEMM (optim.Adam.fuse_into_wgrad): 26 B of HBM traffic per parameter instead of 34,
    and the optimizer pass at the end of the step shrinks to the remaining 6 M parameters.  VARSEP_FUSE_ADAM=0 disables."""
    from .optim import Adam as HipAdam
    if os.environ.get('VARSEP_FUSE_ADAM', '1') != '1' or not isinstance(optimizer, HipAdam):
        return False
    if grad_sync is not None or scaler is not None or not _mlp_family(sep_net) or len(optimizer.param_groups) != 1 or optimizer._buckets:
        return False
    min_numel = int(os.environ.get('VARSEP_FUSE_ADAM_MIN', min_numel))
    big = [p for p in chain_weight_parameters(sep_net) if p.numel() >= min_numel]
    if not big:
        return False
    optimizer.fuse_into_wgrad(big)
    return True


def check_optimizer(optimizer, for_graph=False):
    """torch's fused=True optimizers update parameters without bumping their version counters, which is what the bf16 operand
    copies and the pre-packed weights are keyed on: training would silently continue on stale weights.  Refuse them.  A recorded
    step (`for_graph`) additionally needs an optimizer whose step is capturable: optim.Adam (HIP) or torch's capturable=True."""
    if getattr(optimizer, 'defaults', {}).get('fused'):
        raise ValueError('torch optimizers with fused=True do not invalidate the bf16 operand copies / weight pre-packs of the HIP '
                         'path; use spatiotemporal_variable_separation_amd.optim.Adam (one HIP launch) or a non-fused optimizer')
    if for_graph:
        from .optim import Adam as HipAdam
        if not isinstance(optimizer, HipAdam) and not getattr(optimizer, 'defaults', {}).get('capturable'):
            raise ValueError('a recorded training step (hipGraph) needs spatiotemporal_variable_separation_amd.optim.Adam or a torch '
                             'optimizer constructed with capturable=True')


class _TrainState:
    """Snapshot of everything one optimisation step changes -- parameters, BatchNorm buffers, optimizer state, loss scale and the
    global NumPy stream (`t_random`, train.py:72-75) -- so that GraphedStep's warm-up steps leave no trace."""

    def __init__(self, sep_net, optimizer, scaler=None):
        self.tensors = [t for t in list(sep_net.parameters()) + list(sep_net.buffers())]
        for group in optimizer.param_groups:
            if isinstance(group.get('step_dev'), torch.Tensor):
                self.tensors.append(group['step_dev'])
        for st in optimizer.state.values():
            self.tensors += [v for v in st.values() if isinstance(v, torch.Tensor)]
        if scaler is not None:
            self.tensors.append(scaler.state)
        self.fresh_opt = optimizer if not optimizer.state else None       # state is created by the first step: drop it again
        self.values = [t.detach().clone() for t in self.tensors]
        self.np_state = np.random.get_state()
        self.skipped = {id(p): st.get('skipped') for p, st in optimizer.state.items() if isinstance(st, dict) and 'skipped' in st}
        self.opt = optimizer

    def restore(self):
        with torch.no_grad():
            for t, v in zip(self.tensors, self.values):
                t.copy_(v)                                # in-place: bumps the version counter, operand copies refresh themselves
            if self.fresh_opt is not None:
                for st in self.fresh_opt.state.values():                  # created during the warm-up: back to "never stepped"
                    for v in st.values():
                        if isinstance(v, torch.Tensor):
                            v.zero_()
                    if 'skipped' in st:
                        st['skipped'] = 0
                for group in self.fresh_opt.param_groups:
                    if isinstance(group.get('step_dev'), torch.Tensor):
                        group['step_dev'].zero_()
            else:
                for p, st in self.opt.state.items():
                    if id(p) in self.skipped:
                        st['skipped'] = self.skipped[id(p)]
        np.random.set_state(self.np_state)


class LossScaler:
    """Dynamic loss scaling of fp16 training with torch.cuda.amp.GradScaler's semantics and defaults (reference train.py:96-97,
    151-155: `scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()`), kept entirely on the device so the step
    stays recordable into a hipGraph: `state` = [scale, found_inf, growth_tracker, skipped_steps] (fp32).

      backward(loss)  -- d loss = scale (the scaled loss is never formed: the scale enters as the initial gradient);
      step(optimizer) -- vs_check_finite_multi over every gradient sets found_inf; the Adam kernel multiplies gradients by 1/scale
                         on the fly and leaves parameters, moments and the step count untouched when found_inf is set;
                         vs_loss_scale_update then backs the scale off (x0.5) or grows it (x2 after 2000 clean steps)."""

    def __init__(self, device, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.state = torch.tensor([init_scale, 0.0, 0.0, 0.0], dtype=torch.float32, device=device)
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self.init_scale = init_scale

    def scale_tensor(self):
        return self.state[0]

    def get_scale(self):
        return float(self.state[0].item())

    def skipped_steps(self):
        return int(self.state[3].item())

    def backward(self, loss):
        loss.backward(self.state[0])

    def step(self, optimizer):
        from . import functional as VF, ops
        from .optim import Adam as HipAdam
        if not isinstance(optimizer, HipAdam):
            raise ValueError('fp16 loss scaling is implemented by spatiotemporal_variable_separation_amd.optim.Adam (device-side skip)')
        if optimizer._buckets or optimizer._fused:
            raise ValueError('fp16 loss scaling cannot be combined with Adam updates issued during backward (overlap_with_backward / '
                             'fuse_into_wgrad): they would apply still-scaled gradients and could not be skipped on overflow')
        grads = []
        for group in optimizer.param_groups:
            for p in group['params']:
                g = VF.lowp_gradient(p)
                g = p.grad if g is None else g
                if g is not None:
                    grads.append(g)
        ops.check_finite_multi(grads, self.state)
        optimizer.step(grad_scale_state=self.state)
        ops.loss_scale_update(self.state, self.growth_factor, self.backoff_factor, self.growth_interval)

    def describe(self):
        return ('dynamic loss scaling (GradScaler semantics on the device: init %g, x%g after %d clean steps, x%g on overflow); '
                'scale now %g, %d step(s) skipped' % (self.init_scale, self.growth_factor, self.growth_interval, self.backoff_factor,
                                                      self.get_scale(), self.skipped_steps()))


def make_loss_scaler(device, **kw):
    return LossScaler(device, **kw)


def check_rollout_exchange(device):
    """The weight-stationary rollout kernels exchange partial sums between workgroups with bounded spins; a spin that gives up sets
    an error word and the step's numbers are garbage.  Raise instead of training on (reads device words, i.e. synchronises).  `train()`
    itself no longer raises: see `recover_exchange`."""
    from . import ops
    from ._lib import VarsepHipError
    err = ops.rollout_exchange_error(device)
    if err:
        raise VarsepHipError('rollout kernel: inter-workgroup exchange timed out (code %d) -- the integrator results of at least one '
                             'step since the last check are invalid; not continuing.  (With 8 / 16 row slabs the exchange relies on the '
                             'workgroups of a slab sharing an XCD; VS_ROLLOUT_XCD_LOCAL=0 selects the placement-independent agent-scope exchange.)' % err)


def recover_exchange(device, log=True, grad_sync=None):
    """What `train()` does about a timed-out in-launch exchange (reads device words: synchronises).  Returns the error code (0: nothing happened).

    While the process's guard word is registered (ops.exchange_guard: always, for the first device of a process) a time-out cannot reach the
    parameters: the kernels raise the guard, and every optimizer launch issued or recorded afterwards -- vs_adam_multi, vs_gemm_adam, the step
    counter -- reads it first and does nothing while it is set (include/varsep_hip.h, vs_exchange_guard_set), so the steps between the time-out
    and this call were SKIPPED, like overflow steps under fp16 loss scaling.  Here the cause is removed -- the MLP integrator switches to the
    placement-independent agent-scope exchange for the rest of the process, the one-launch ConvResBlock layers to the two-launch form -- and
    the word is cleared; the caller drops its recorded step (the exchange mode is baked into a recording) and training goes on.  Without a
    guard (a second device in one process) the old behaviour stays: raise."""
    import sys
    from . import ops
    from ._lib import VarsepHipError
    guarded = ops.exchange_guard(torch.device(device)) is not None
    skipped = ops.exchange_skipped_steps(device) if guarded else 0      # (read before the guard is cleared: counted by the step-count kernel)
    err = ops.rollout_exchange_error(device)
    if grad_sync is not None and grad_sync.world_size > 1:
        # every rank must take the same branch (a re-recording contains collectives): the codes are bit masks, every rank's bits count
        import torch.distributed as dist
        word = torch.tensor([err], dtype=torch.int32, device=device if grad_sync.backend == 'nccl' else 'cpu')
        dist.all_reduce(word, op=dist.ReduceOp.BOR, group=grad_sync.group)
        err = int(word.item())
    if not err:
        return 0
    if not guarded:
        raise VarsepHipError('rollout kernel: inter-workgroup exchange timed out (code %d) and no guard word protects the optimizer on this '
                             'device: the results of at least one step are invalid' % err)
    if err & 1:
        ops.rollout_xcd_local(False)
    if err & 2:
        os.environ['VARSEP_FUSED_RESBLOCK'] = '1'
    if log:
        # (the fused weight-gradient updates -- vs_gemm_adam -- issued in the SAME backward pass before the exchange timed out have been applied:
        # that one step is partial (decoder updated, the rest not); every later step was skipped as a whole)
        sys.stderr.write('varsep: an in-launch exchange timed out (code %d); the optimizer skipped %s (batches consumed, no update; the first of them '
                         'may have updated the weights whose Adam step runs inside their weight-gradient GEMM).  Continuing with %s\n'
                         % (err, ('%d step(s)' % skipped) if skipped else 'the affected step(s)', ' and '.join((['the agent-scope integrator exchange'] if err & 1 else []) +
                                              (['two-launch ConvResBlock layers'] if err & 2 else []))))
    return err


def compute_losses(cond, target, sep_net, nt_cond, nt_pred, offset, skipco, lamb_ae, lamb_s, lamb_t, lamb_pred,
                   average_tloss=False, t_random=None):
    """The four loss terms and their weighted sum for one batch (train.py:117-149).

    Returns (total, {'ae','zero','pred','t_reg'}, forecasts, t_codes)."""
    assert offset == nt_cond or offset == 0
    if cond.is_cuda and _mlp_family(sep_net):
        return _compute_losses_mlp_batched(cond, target, sep_net, nt_cond, nt_pred, offset, lamb_ae, lamb_s, lamb_t,
                                           lamb_pred, average_tloss, t_random)
    if cond.is_cuda and torch.is_grad_enabled():
        from . import ops
        ops.exchange_epoch_advance(cond.device)      # the fused ConvResBlock layers number their launches from 1 under a new epoch base
    if cond.is_cuda and torch.is_grad_enabled() and os.environ.get('VARSEP_PREPACK_CONV', '1') == '1':
        from . import functional as VF
        VF.prepack_conv3_weights(sep_net)            # every stale 3x3 weight pre-pack of the step in one launch
    full_data = torch.cat([cond, target], dim=1)
    pairs = (cond.is_cuda and getattr(sep_net, 'fused', False) and getattr(sep_net.Es, 'call_groups', False)
             and getattr(sep_net.Et, 'call_groups', False) and os.environ.get('VARSEP_ENCODER_PAIRS', '1') == '1')
    if pairs:
        # the reference calls each encoder twice per step (E_s on the first and the last window, E_t on the random and the
        # conditioning window): run each pair as ONE batch of two call groups -- every BatchNorm keeps per-call statistics and
        # folds its running estimates in call order, so the arithmetic is the reference's; half the launches, twice the GEMM width
        if t_random is None:
            T = full_data.size(1)
            t_random = np.random.randint(nt_cond, T) if offset == 0 else np.random.randint(nt_cond, T + 1)
        if isinstance(t_random, torch.Tensor):
            window, supervision_data = _device_window(full_data, t_random, nt_cond, offset)
        else:
            window, supervision_data = full_data[:, t_random - nt_cond:t_random], full_data[:, t_random - offset]
        # A convolutional integrator (SST's ConvResnet, resnet.py:53-88: ~470 small launches forward, as many backward, each on 8 maps) needs
        # E_t's code only; E_s, the reconstruction decode and their backward passes are independent of it.  VARSEP_ROLLOUT_SIDE=1 runs it on a
        # side stream (autograd replays the assignment in backward) between E_t and the forecast decode: same arithmetic, same per-module
        # call order of every BatchNorm.  Measured and NOT the default: the replayed SST step goes from 21.0 to 55-58 ms with 8 hardware
        # queues (every launch of two concurrently running chains of ~800 small kernels pays a cross-queue hand-over), and is unchanged
        # (21.3 ms) with 4 or 2 queues, where the runtime maps both branches onto one queue.
        rolled = None
        side_roll = (cond.is_cuda and not hasattr(sep_net.t_resnet, 'rollout') and hasattr(sep_net.decoder, 'decode_sequence')
                     and os.environ.get('VARSEP_ROLLOUT_SIDE', '0') == '1')
        if side_roll:
            from . import functional as VF
            t_rand, t_cond = _encode_pair(sep_net.Et, window, cond, False)
            main, side = torch.cuda.current_stream(), VF._side_stream('rollout')
            VF.note_main_stream(main)
            side.wait_stream(main)
            t_cond.record_stream(side)
            with torch.cuda.stream(side):
                codes, t_residuals = sep_net._roll(t_cond, nt_pred + offset)
                rolled = (torch.stack(codes, dim=1), t_residuals)
            s_old, s_recent = _encode_pair(sep_net.Es, full_data[:, :nt_cond], full_data[:, -nt_cond:], skipco)
        else:
            s_old, s_recent = _encode_pair(sep_net.Es, full_data[:, :nt_cond], full_data[:, -nt_cond:], skipco)
            t_rand, t_cond = _encode_pair(sep_net.Et, window, cond, False)
        d_s, d_t = s_old, t_rand
        if cond.is_cuda:
            from . import functional as VF
            d_s, d_t = VF.cut((s_old, t_rand))     # (two-segment backward of a recorded data-parallel step: the decoder sees detached leaves)
        if skipco:
            reconstruction = sep_net.decoder(d_s[0], d_t, skip=d_s[1])
        else:
            reconstruction = sep_net.decoder(d_s, d_t)
        fused_mse = os.environ.get('VARSEP_FUSED_FRAME_MSE', '1') == '1' and full_data.dtype == torch.float32
        ae_idx = None
        if fused_mse:
            from . import functional as VF
            if isinstance(t_random, torch.Tensor):
                ae_idx = (t_random.reshape(1) - offset).to(torch.int32)
            else:
                ae_idx = _frame_index(int(t_random) - offset, 0, 0, full_data.device, full_data.size(1) + 1)[:1]
        if rolled is not None:
            main.wait_stream(side)
            rolled[0].record_stream(main)
            forecasts, t_codes, _, _ = sep_net.get_forecast(cond, nt_pred + offset, init_t_code=t_cond, init_s_code=s_old, rolled=rolled)
        else:
            forecasts, t_codes, _, _ = sep_net.get_forecast(cond, nt_pred + offset, init_t_code=t_cond, init_s_code=s_old)
        if fused_mse and os.environ.get('VARSEP_FUSED_CONV_LOSSES', '1') == '1':
            # the four losses and their weighted sum in 4 launches (3 backward): no concatenation of the skip tensors, no scalar launches.
            # t_codes[:, 0] IS t_cond (the first code of the rollout): the regulariser reads the encoder output directly, so its gradient
            # does not travel through a zero-filled [B, n, ...] tensor
            n_f = forecasts.shape[1]
            f_idx = _frame_index(0, nt_cond if offset == 0 else 0, n_f, full_data.device, full_data.size(1) + 1)[1:]
            fused = VF.conv_losses(reconstruction, forecasts, full_data, ae_idx, f_idx, s_old, s_recent, skipco, t_cond,
                                   (lamb_ae, lamb_s, lamb_t, lamb_pred), average_tloss)
            if fused is not None:
                return fused[0], fused[1], forecasts, t_codes
        if fused_mse:
            # both frame losses through the fused kernels (no supervision-frame copy, no slices of full_data)
            ae_loss_value = VF.frames_mse(reconstruction, full_data, ae_idx)
        else:
            ae_loss_value = F.mse_loss(supervision_data, reconstruction, reduction='mean')
        spatial_ode_loss = zero_order_loss(s_old, s_recent, skipco)
    else:
        ae_loss_value, s_recent, s_old = ae_loss(cond, target, sep_net, nt_cond, offset, skipco, t_random=t_random)
        spatial_ode_loss = zero_order_loss(s_old, s_recent, skipco)
        forecasts, t_codes, _, _ = sep_net.get_forecast(cond, nt_pred + offset, init_s_code=s_old)
    forecast_offset = nt_cond if offset == 0 else 0
    if pairs and fused_mse:
        n_f = forecasts.shape[1]
        f_idx = _frame_index(0, forecast_offset, n_f, full_data.device, full_data.size(1) + 1)[1:]
        forecast_loss = VF.frames_mse(forecasts, full_data, f_idx)
    else:
        forecast_loss = F.mse_loss(forecasts, full_data[:, forecast_offset:])
    if average_tloss:
        t_reg = 0.5 * (t_codes[:, 0].pow(2).view(full_data.shape[0], -1)).mean()
    else:
        t_reg = 0.5 * torch.sum(t_codes[:, 0].pow(2), dim=1).mean()
    total_loss = lamb_ae * ae_loss_value + lamb_s * spatial_ode_loss + lamb_pred * forecast_loss + lamb_t * t_reg
    terms = {'ae': ae_loss_value, 'zero': spatial_ode_loss, 'pred': forecast_loss, 't_reg': t_reg}
    return total_loss, terms, forecasts, t_codes


def train(xp_dir, train_loader, device, sep_net, optimizer, scheduler, use_apex_amp, use_torch_amp, epochs, lamb_ae,
          lamb_s, lamb_t, lamb_pred, offset, nt_cond, nt_pred, no_s, skipco, chkpt_interval, average_tloss,
          grad_sync=None, log_interval=None, hip_graph=False, scaler=None):
    """Same 20 positional arguments as the reference's `train` (train.py:91-92).

    Additive keyword arguments: `grad_sync` (a `parallel.GradAllReducer`, data-parallel gradient averaging over
    RCCL), `log_interval` (print losses / frames-per-second every N steps), `hip_graph` (record
    the whole step once into a hipGraph -- `GraphedStep` -- and replay it; the optimizer must be optim.Adam or Adam(capturable=True))
    and `scaler` (a `LossScaler`; created here when `use_torch_amp` is set).  `use_apex_amp` is rejected (no Apex on the MI355X
    path).  `use_torch_amp` has the reference's meaning (train.py:96-97, 151-155): fp16 compute (fp16 MFMA operands and hidden
    activations, fp32 accumulation and master weights) with dynamic loss scaling; the bf16 mode is selected explicitly with
    `functional.set_precision('bf16')` / `--precision bf16` and needs no scaler.
    """
    import time
    from . import functional as VF
    check_optimizer(optimizer)
    if use_apex_amp:
        raise ImportError('Apex is not part of the MI355X-native path; use --torch_amp (fp16 MFMA + loss scaling, as torch.cuda.amp) '
                          'or --precision bf16')
    if use_torch_amp:
        VF.set_precision('fp16')
        if scaler is None:
            scaler = LossScaler(device)
    if no_s:
        lamb_t = 0
        print("No regularization on T as there is no S")
    assert offset == nt_cond or offset == 0

    step, t_last = 0, time.time()
    graphed = None
    GUARD_POLL = max(1, int(os.environ.get('VARSEP_GUARD_POLL', '50')))
    rank = grad_sync.rank if grad_sync is not None else 0
    world = grad_sync.world_size if grad_sync is not None else 1
    # folding repeated gradients (VF.fold_repeated_gradients) replaces ~1000 tiny add launches per SST step by one multi-tensor
    # add per block.  Under a reducer the convolution / BatchNorm gradients of a conv family accumulate straight into its bucket views
    # (conv_gradient_sinks): the same kernels as on one GPU; the MLP family under a reducer has one gradient per parameter anyway
    conv_sinks = conv_gradient_sinks(sep_net, grad_sync) if (grad_sync is not None and not _mlp_family(sep_net)) else None
    VF.fold_repeated_gradients(os.environ.get('VARSEP_FOLD_GRADS', '1') == '1' and (grad_sync is None or bool(conv_sinks)))

    def checkpoint(epoch_number=None, collective=True):
        # data parallel: replicas hold identical parameters; BatchNorm buffers are per replica (no SyncBN), rank 0's are the ones
        # kept and broadcast so that every replica continues from what was saved; only rank 0 writes.
        # A (sticky) exchange time-out word of the integrator is handled first (recover_exchange: the affected steps were skipped by the
        # optimizer, the exchange mode is switched, nothing raises).  `collective=False` (after Ctrl-C, which may have hit one rank only)
        # skips every collective.
        nonlocal graphed
        err = 0
        if torch.device(device).type == 'cuda':
            err = int(recover_exchange(device, grad_sync=grad_sync if collective else None))        # (a time-out cannot have reached the parameters: the guard word made the optimizer skip)
            if err:
                graphed = None                         # the exchange mode is baked into a recording
        if grad_sync is not None and collective and getattr(grad_sync, 'masters_dirty', False):
            grad_sync.sync_masters(optimizer)          # sharded optimizer: every rank's slice of the fp32 masters / moments to every rank
        elif grad_sync is not None and not collective and getattr(grad_sync, 'masters_dirty', False):
            # Ctrl-C may have reached this rank only: no collective.  Outside its own slice this rank's fp32 masters are as old as the last
            # sync_masters(); the all-gathered 16-bit operand copies are current everywhere -- save those (rounded to the compute type) rather
            # than stale weights under the normal file name
            if grad_sync.fill_masters_from_arena() and rank == 0:
                import sys
                sys.stderr.write('varsep: interrupted under the sharded optimizer: the saved weights outside rank 0\'s slice are the current '
                                 '16-bit operand copies (rounded to the compute type), not the fp32 masters held by the other ranks\n')
        if grad_sync is not None and world > 1 and collective:
            import torch.distributed as dist
            from .parallel import broadcast_buffers
            broadcast_buffers(sep_net, process_group=grad_sync.group)
        if rank == 0:
            save(xp_dir, sep_net, epoch_number=epoch_number)

    interrupted = False
    try:
        for epoch in range(epochs):
            sep_net.train()
            sampler = getattr(train_loader, 'sampler', None)
            if hasattr(sampler, 'set_epoch'):
                sampler.set_epoch(epoch)             # DistributedSampler: a new permutation per epoch, the same on every rank
            for cond, target in train_loader:
                cond, target = cond.to(device, non_blocking=True), target.to(device, non_blocking=True)
                if hip_graph:
                    if graphed is None:
                        graphed = GraphedStep(sep_net, optimizer, cond, target, nt_cond, nt_pred, offset,
                                              (lamb_ae, lamb_s, lamb_t, lamb_pred), average_tloss, grad_sync=grad_sync, scaler=scaler)
                    if cond.shape == graphed.cond.shape:
                        total_loss = graphed.step(cond, target)
                        step += 1
                        first = graphed.steps_replayed <= 3      # the first replays of a recording are checked one by one (a sync each)
                        # the guard is sticky: every step until it is read is skipped.  Read it with the log line, or every GUARD_POLL steps
                        # when nothing is logged (one synchronisation per poll)
                        if first or (log_interval and step % log_interval == 0) or (not log_interval and step % GUARD_POLL == 0):
                            torch.cuda.synchronize()
                            if recover_exchange(device, grad_sync=grad_sync):
                                graphed = None                   # re-record with the exchange mode now in force
                                continue
                        if log_interval and step % log_interval == 0:
                            dt, t_last = time.time() - t_last, time.time()
                            if rank == 0:
                                print(f'epoch {epoch} step {step}: total {total_loss.item():.5f} | '
                                      f'{world * log_interval * cond.shape[0] * nt_pred / dt:.0f} frames/s (hipGraph)')
                        continue                     # a ragged last batch falls through to the eager path below
                # eager step.  After a recorded data-parallel step with directly written bf16 wire gradients, the chains' weight
                # gradients of THIS step go through autograd into the fp32 buckets: switch the wire shortcut off for its duration,
                # or the reducer would skip them and Adam would read the previous step's bf16 images
                lowp_saved = None
                if grad_sync is not None and getattr(grad_sync, 'masters_dirty', False):
                    grad_sync.sync_masters(optimizer)    # the eager step below is the replicated update: it needs complete masters / moments
                if grad_sync is not None and getattr(grad_sync, 'direct_lowp', False):
                    lowp_saved = dict(VF._LOWP_GRAD)
                    grad_sync.direct_lowp = False
                    VF.set_lowp_gradients(None)
                try:
                    if grad_sync is not None:
                        grad_sync.zero_grad()            # gradients are views into flat all-reduce buckets
                        if conv_sinks:
                            VF.set_conv_grad_outputs(conv_sinks)
                    else:
                        optimizer.zero_grad()
                    total_loss, terms, _, _ = compute_losses(cond, target, sep_net, nt_cond, nt_pred, offset, skipco,
                                                             lamb_ae, lamb_s, lamb_t, lamb_pred, average_tloss)
                    if scaler is not None:
                        scaler.backward(total_loss)
                    else:
                        total_loss.backward()
                    if grad_sync is not None:
                        grad_sync.all_reduce()
                    if scaler is not None:
                        scaler.step(optimizer)
                    else:
                        optimizer.step()
                finally:
                    VF.set_conv_grad_outputs(None)
                    if lowp_saved is not None:
                        grad_sync.direct_lowp = True
                        VF._LOWP_GRAD.update(lowp_saved)
                VF.flush_bn_call_counts()
                step += 1
                if not log_interval and step % GUARD_POLL == 0 and torch.device(device).type == 'cuda':
                    torch.cuda.synchronize()
                    if recover_exchange(device, grad_sync=grad_sync):
                        graphed = None                   # a recording made earlier has the old exchange mode baked in
                if log_interval and step % log_interval == 0:
                    torch.cuda.synchronize()
                    if recover_exchange(device, grad_sync=grad_sync):
                        graphed = None                   # (as the graph branch and checkpoint() do)
                    dt = time.time() - t_last
                    t_last = time.time()
                    fps = log_interval * cond.shape[0] * nt_pred * world / dt
                    if rank == 0:
                        print(f'epoch {epoch} step {step}: total {total_loss.item():.5f} ' +
                              ' '.join(f'{k} {v.item():.5f}' for k, v in terms.items()) + f' | {fps:.0f} frames/s')
            if scheduler is not None:
                scheduler.step()
            if chkpt_interval is not None and (epoch + 1) % chkpt_interval == 0:
                checkpoint(epoch + 1)
    except KeyboardInterrupt:
        interrupted = True
    finally:
        VF.fold_repeated_gradients(False)
    checkpoint(collective=not interrupted)
