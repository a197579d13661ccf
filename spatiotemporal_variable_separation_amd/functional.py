"""Autograd functions of the hot path, built on the C-ABI kernels (ops.py).

Precision policy (`set_precision` / `precision(...)`):
  'fp32' -- operands, activations and accumulation in fp32 (exact-fp32 MFMA): the parity mode, compared with the
            fp32 CPU oracle at <= 1e-3 relative (measured ~1e-6).
  'bf16' -- GEMM operands and hidden activations in bf16, fp32 accumulation, fp32 master weights / gradients /
            codes / frames / losses.  bf16 shadow copies of the weights are refreshed when the parameter's
            version counter changes (i.e. once per optimizer step).
  'fp16' -- the same rounding points with IEEE half operands (v_mfma_f32_*_f16): the reference's --torch_amp mode
            (torch.cuda.amp autocast, train.py:96-97); trained with dynamic loss scaling (train.LossScaler).
"""
import contextlib
import os
import weakref

import torch

from . import ops
from ._lib import LAYOUT_R as R, LAYOUT_S as S, VarsepHipError, require_cuda

_STATE = {'precision': 'fp32'}


def set_precision(p):
    assert p in ('fp32', 'bf16', 'fp16'), p
    _STATE['precision'] = p


def get_precision():
    return _STATE['precision']


@contextlib.contextmanager
def precision(p):
    old = _STATE['precision']
    set_precision(p)
    try:
        yield
    finally:
        _STATE['precision'] = old


def folding_repeated_gradients():
    return bool(_STATE.get('fold_grads'))


def fold_repeated_gradients(flag, flush=True):
    """Modules called many times per step (the integrator's convolutions: once per rollout step) produce one weight, bias, gamma
    and beta gradient PER CALL, and autograd adds each of them to `.grad` with its own 3-4 us launch (SST, 40 predicted frames:
    ~1200 such adds per step).  With this switch on, a block whose parameters already hold a gradient adds all of its
    contributions with ONE multi-tensor launch, into the tensor autograd holds as the parameter's pending gradient, and returns
    nothing to autograd for them.  Only valid without gradient hooks
    (the bucketed all-reduce counts hook calls), so `train()` / `bench.py` turn it on for single-process runs only."""
    if not flag and flush:
        flush_bn_call_counts()
    _STATE['fold_grads'] = bool(flag)


_FOLD = {'task': None, 'acc': {}}

# ---- weight gradients of convolutions that are applied many times per step, batched over the calls -----------------------------------
# The SST integrator applies each of its 3x3 convolutions once per predicted frame (39 calls per step) on B = 8 samples of 16 x 16:
# 234 weight gradients of ~1 GFLOP, each a column-matrix gather + a split-K GEMM + a reduce (about 60 us of launch-bound work, 14 ms
# of the step).  A weight gradient is a sum over the batch axis, so the calls of one weight are ONE weight gradient over the
# concatenation of their (dz, x) pairs: in fold mode such calls only REMEMBER their pair; when the backward pass ends (an autograd
# engine callback) each weight's pairs are concatenated along the batch axis and its gradient is computed by one launch sequence
# (K = 39 x 2048 pixels: a long, efficient reduction) into the tensor autograd was handed at the weight's first call.
_DEFER_W = {'slots': {}, 'queued': False}


def mark_repeated(weights):
    """Tell the weight-gradient machinery that these convolution weights are applied MANY times per step (the integrator's blocks:
    once per predicted frame): only such weights have their (dz, x) pairs remembered and their gradient computed once per step."""
    for w in weights:
        w._vs_repeated = True


def _defer_wgrad_ok(w, dz, transposed, stride):
    # (a weight that is used once or twice per step -- encoder / decoder layers -- gains nothing from the batching and would pay a
    # zero-filled buffer and, for calls of different batch sizes, a concatenation)
    return (_STATE.get('fold_grads') and os.environ.get('VARSEP_DEFER_WGRADS', '1') == '1' and not transposed and stride == 1
            and getattr(w, '_vs_repeated', False) and dz.shape[0] * dz.shape[2] * dz.shape[3] <= 16384 and not torch.is_grad_enabled())


def flush_deferred_wgrads():
    """Compute the remembered weight gradients (see above).  Runs by itself at the end of every backward pass that deferred any."""
    slots, _DEFER_W['slots'] = _DEFER_W['slots'], {}
    _DEFER_W['queued'] = False
    for slot in slots.values():
        with torch.cuda.stream(slot['stream']):
            pairs = slot['pairs']
            if len(pairs) == 1:
                dz, xc = pairs[0]
            else:
                # 3x3 on the row-band kernel: the pairs are addressed where they lie (no concatenation: 2 x 39 copies of 2 MB in the SST step)
                if (slot['shape'][2:] == (3, 3) and slot['stride'] == 1 and slot['pad'] == 1
                        and ops.conv3_wgrad_band_pieces(pairs, slot['shape'], into=slot['g']) is not None):
                    continue
                dz = torch.cat([p[0] for p in pairs], dim=0)
                xc = torch.cat([p[1] for p in pairs], dim=0)
            ops.conv_wgrad(dz, xc, slot['shape'], slot['stride'], slot['pad'], False, into=slot['g'])
    # (the engine has already joined the streams of the pass with the caller's: work issued now on another stream is joined here)
    cur = torch.cuda.current_stream()
    for st in {slot['stream'] for slot in slots.values()}:
        if st != cur:
            cur.wait_stream(st)


# Exactly-zero gradients (a convolution bias in front of a training-mode BatchNorm: the batch mean removes it) still have to be handed to
# autograd as tensors.  One `zeros_like` each is a 4 us fill launch per convolution and step (TaxiBJ: 67 per step); here they are
# slices of ONE zero-filled pool per backward pass (sized from the previous pass), each slice a tensor of its own, so autograd keeps it
# as the parameter's gradient without cloning.  Nothing ever writes into such a gradient (later contributions of the pass are dropped
# as exact zeros too, and optimizers only read gradients).
# One pool per STREAM: a slice handed out as a parameter's gradient is added to in place by autograd when the parameter is used again in the pass
# (on the stream of ITS node); with a pool shared between streams that add could run before the pool's fill on the other stream.
_ZERO_POOLS = {}


def zero_grad_like(p):
    task = torch._C._current_graph_task_id()
    zp = _ZERO_POOLS.setdefault(torch.cuda.current_stream(p.device).cuda_stream if p.is_cuda else -1,
                                {'task': None, 'buf': None, 'off': 0, 'need': 0, 'last': 0})
    n = p.numel()
    if zp['task'] != task:
        zp['last'] = max(zp['last'], zp['need'])
        zp.update(task=task, buf=None, off=0, need=0)
    zp['need'] += n
    if p.dtype != torch.float32 or task == -1:
        return torch.zeros_like(p)
    if zp['buf'] is None or zp['buf'].device != p.device or zp['off'] + n > zp['buf'].numel():
        zp['buf'] = torch.zeros(max(zp['last'], 4 * n, 1024), dtype=torch.float32, device=p.device)
        zp['off'] = 0
        main = _SIDE.get('main')
        if main is not None and main != torch.cuda.current_stream():
            zp['buf'].record_stream(main)            # allocated inside a side-stream node, read by the optimizer on the step's stream
    out = zp['buf'][zp['off']:zp['off'] + n].view(p.shape)
    zp['off'] += n
    return out


def _fold_slots():
    """Per backward pass: parameter id -> the gradient tensor autograd already holds for it (its first contribution)."""
    task = torch._C._current_graph_task_id()
    if _FOLD['task'] != task:
        _FOLD['task'], _FOLD['acc'] = task, {}
    return _FOLD['acc']


_BN_COUNTS = {}


def count_bn_calls(bn, calls):
    """`num_batches_tracked += calls` of a training-mode BatchNorm (one per reference call).  While gradients are folded (the
    training loop owns the step and calls `flush_bn_call_counts()` at its end) the increments are collected and applied by ONE
    multi-tensor launch per step instead of one launch per BatchNorm call (SST: ~300 per step)."""
    _EPOCH['bn'] += 1                    # the call rewrites running_mean / running_var through raw pointers: eval-mode folds are outdated
    # a recording must contain the increment: either right here, or -- when the recorder has promised to call
    # flush_bn_call_counts() inside the capture (train.GraphedStep) -- in that one multi-tensor launch
    if not _STATE.get('fold_grads') or (torch.cuda.is_current_stream_capturing() and not _STATE.get('bn_counts_flushed_in_capture')):
        bn.num_batches_tracked += calls
        return
    ent = _BN_COUNTS.get(id(bn))
    if ent is None:
        _BN_COUNTS[id(bn)] = [bn.num_batches_tracked, calls]
    else:
        ent[1] += calls


def bn_counts_flushed_in_capture(flag):
    """The caller records `flush_bn_call_counts()` into its capture: per-call counter increments may be collected during it."""
    _STATE['bn_counts_flushed_in_capture'] = bool(flag)


def flush_bn_call_counts():
    if _BN_COUNTS:
        bufs = [e[0] for e in _BN_COUNTS.values()]
        incs = [e[1] for e in _BN_COUNTS.values()]
        _BN_COUNTS.clear()
        torch._foreach_add_(bufs, incs)


# ---- backward in two segments (train.GraphedStep under a reducer) --------------------------------------------------------------------
# The decoder is the LAST thing the forward pass runs and the FIRST thing backward finishes: once its parameters' gradients are complete the
# all-reduce of their bucket can travel while the integrator's and the encoders' backward passes still run.  A recorded step cannot fire
# hooks, so the recording is split there.  `cut(x)` (called on every tensor that enters the decoder: SeparableNetwork.get_forecast,
# train.compute_losses) hands the decoder a DETACHED leaf of x and remembers the pair: the decoder's part of the autograd graph then hangs off
# those leaves.  Segment 1 = forward + `backward(total, inputs=decoder parameters + leaves)` (the loss heads and the decoder);
# segment 2 = `backward([total] + originals, [d total] + [leaf.grad ...], inputs=the other parameters)`: the leaves' gradients enter the
# originals, and `total` contributes what reaches the encoders / the integrator without passing the decoder (the code regularisers).
# (Passing the decoder inputs THEMSELVES as `inputs=` of the first call is wrong whenever one of them is an ancestor of another -- the skip
# connections: h3 feeds the code -- because `inputs=` asks for total derivatives, and the second call would add the path through the
# code a second time.  A true cut needs leaves.)
_CUTS = {'on': False, 'pairs': {}}


def collect_cuts(flag):
    _CUTS['on'] = bool(flag)
    _CUTS['pairs'] = {}


def cut(x):
    """x, or -- while a two-segment backward is being recorded -- a detached leaf standing in for it (tensors, lists / tuples of tensors)."""
    if not _CUTS['on'] or x is None:
        return x
    if isinstance(x, (list, tuple)):
        return type(x)(cut(t) for t in x)
    if isinstance(x, torch.Tensor) and x.requires_grad and x.grad_fn is not None:
        ent = _CUTS['pairs'].get(id(x))
        if ent is None:
            ent = _CUTS['pairs'][id(x)] = (x, x.detach().requires_grad_(True))
        return ent[1]
    return x


def cut_pairs():
    return list(_CUTS['pairs'].values())


def promise_loss_gradient(t):
    """`t` (one fp32 element on the device, or None) is the tensor the caller WILL pass as the gradient of the total loss
    (`total.backward(t)`): TrainLosses then writes its gradients in the forward pass (one read of the frame stack per step instead of
    two).  The value is read on the device when the forward kernel runs; a backward call with any other tensor recomputes."""
    _STATE['promised_loss_grad'] = t


def promised_loss_gradient():
    return _STATE.get('promised_loss_grad')


def compute_dtype():
    return {'fp32': torch.float32, 'bf16': torch.bfloat16, 'fp16': torch.float16}[_STATE['precision']]


# ------------------------------------------------------------------------------------------------ side streams
# Two kinds of work are off the critical path of a training step and run on side HIP streams when enabled:
#   * weight/bias gradients (every wgrad GEMM, column sum): only the optimizer consumes them, so they are launched on
#     a dedicated stream behind an event and joined by `join_side_streams()` right before the optimizer step;
#   * the latent rollout (a 32-workgroup persistent kernel that leaves 7/8 of the chip idle): its forward overlaps the
#     E_s encoder, its backward overlaps the decoder/encoder weight gradients.
# Only valid when every parameter receives ONE gradient per step (the batched MLP-family step) and without hook-driven
# gradient all-reduce; `train._compute_losses_mlp_batched` / `GraphedStep` switch it on, everything else leaves it off.
_SIDE = {'on': False, 'lanes': [], 'next_lane': 0, 'rollout': None, 'es': None, 'hold': False, 'held': [], 'hold_main': None, 'late': []}
# Deferred gradient work is spread over a few streams ("lanes", one per Linear chain / integrator backward in turn): most of it
# is GEMMs of 10-50 us that fill a fraction of the chip each, and one stream would run them one after the other.
N_LANES = max(1, int(os.environ.get('VARSEP_WGRAD_LANES', '3')))


_OWN_STREAMS = []


def own_stream(device=None):
    """A HIP stream of this package's own (hipStreamCreateWithPriority, non-blocking, default priority), wrapped as torch.cuda.ExternalStream and
    never destroyed.  `torch.cuda.Stream()` hands out the next of 32 POOLED streams per device and priority: after 32 requests anywhere in
    the process two Stream objects are the same HIP stream.  Two of the streams of a step being one stream (a gradient lane and the
    optimizer's stream, say) makes `a.wait_stream(b)` a stream waiting for its own event, which inside a capture enters the stream into
    its own list of joined streams and hipStreamEndCapture recurses until the stack ends (seen as a segmentation fault late in a long test
    session, at a position that moved with the number of streams requested before).  Streams made here cannot coincide with each other
    nor with any pooled stream."""
    import ctypes
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    torch.cuda.init()
    hip = ctypes.CDLL('libamdhip64.so')
    handle = ctypes.c_void_p(0)
    with torch.cuda.device(dev):
        err = hip.hipStreamCreateWithPriority(ctypes.byref(handle), ctypes.c_uint(1), ctypes.c_int(0))        # 1 = hipStreamNonBlocking
    if err != 0 or not handle.value:
        raise RuntimeError('hipStreamCreateWithPriority failed (%d)' % err)
    s = torch.cuda.ExternalStream(handle.value, device=dev)
    _OWN_STREAMS.append(s)
    return s


def enable_side_streams(flag):
    _SIDE['on'] = bool(flag)
    _SIDE['next_lane'] = 0
    _SIDE['markers'] = {}
    if not flag:
        _SIDE['hold'], _SIDE['held'], _SIDE['late'] = False, [], []


def tail_fused_updates():
    """Default (VARSEP_ADAM_UNDER_FUSED=0 disables; WaveEq step, same box: 1.406-1.415 -> 1.378 ms): a chain's fused first-layer update (the 640 MB weight-gradient + Adam launches of the WaveEq encoders) is
    issued LAST on its lane, behind the chain's bias sums and a marker event; the optimizer launch for the remaining parameters then waits
    for the markers only and runs beside the fused updates (disjoint parameters; the step counter is advanced after the full join)."""
    return os.environ.get('VARSEP_ADAM_UNDER_FUSED', '1') == '1'


def tail_split():
    """VARSEP_TAIL_SPLIT=1 (round 5): see MLPChain.backward -- the last chain's fused first-layer update on lane 0, beside its own small gradients."""
    return os.environ.get('VARSEP_TAIL_SPLIT', '0') == '1'


def _lane_marker(lane):
    def fn():
        ev = torch.cuda.Event()
        ev.record()
        _SIDE['markers'][lane] = ev
    fn._vs_marker = True
    return fn


def _queued_on_lane(fn, lane):
    """`fn` has just been issued on gradient lane `lane`.  A marker recorded earlier on that lane stands for "everything but the trailing
    fused update of this lane is done" only while nothing else follows it: any other work queued behind it (a later chain when the lanes
    wrap around, VARSEP_WGRAD_LANES < number of chains) withdraws the marker, and join_side_streams(partial=True) then waits for the
    whole lane."""
    if not (getattr(fn, '_vs_marker', False) or getattr(fn, '_vs_tail', False)):
        (_SIDE.get('markers') or {}).pop(lane, None)


def side_streams_enabled():
    return _SIDE['on']


class chain_forward_stream:
    """While active, MLPChain.forward issues its launches on `stream` (and records what it hands to the caller and keeps for backward on the
    caller's stream); the autograd node is still created under the caller's stream, so BACKWARD runs where it would have run anyway."""

    def __init__(self, stream):
        self.stream = stream

    def __enter__(self):
        self.prev, _SIDE['chain_stream'] = _SIDE.get('chain_stream'), self.stream

    def __exit__(self, *exc):
        _SIDE['chain_stream'] = self.prev


def on_chain_forward_stream():
    """Context: the stream of the active chain_forward_stream, or nothing."""
    from contextlib import nullcontext
    s = _SIDE.get('chain_stream')
    return torch.cuda.stream(s) if s is not None else nullcontext()


def note_main_stream(stream):
    """The stream the step runs on while part of it is on a side stream: tensors of the backward pass that are allocated wherever they are first
    needed but read by the optimizer afterwards (the pooled zero gradients) are recorded on it."""
    _SIDE['main'] = stream


def _side_stream(name):
    if _SIDE[name] is None:
        _SIDE[name] = own_stream()
    return _SIDE[name]


def _lane_stream(i):
    while len(_SIDE['lanes']) <= i:
        _SIDE['lanes'].append(own_stream())
    return _SIDE['lanes'][i]


def next_lane():
    """The lane the next group of deferred launches should use (round robin)."""
    i = _SIDE['next_lane']
    _SIDE['next_lane'] = (i + 1) % N_LANES
    return i


def hold_deferred(flag=True):
    """While held, deferred gradient work is only COLLECTED; `release_deferred()` launches it.  The batched MLP-family step
    holds the decoder's and E_s's weight gradients until the integrator's backward kernel has been launched: that kernel is a
    latency-bound chain on 192 workgroups (215 us at WaveEq size with the rest of the chip idle), the weight-gradient GEMMs fill
    the chip under it, and the decoder's input-gradient chain -- the critical path up to there -- has the GPU to itself instead
    of sharing it with them (measured on the WaveEq step: the decoder's backward 432 -> 190 us)."""
    _SIDE['hold'] = bool(flag) and _SIDE['on']
    _SIDE['hold_main'] = torch.cuda.current_stream() if _SIDE['hold'] else None
    if not flag:
        _SIDE['held'] = []


def deferred_held():
    return _SIDE['on'] and _SIDE['hold']


def _record_on(ws, inputs, outs):
    for t in inputs:
        t.record_stream(ws)
    for t in (outs if isinstance(outs, (list, tuple)) else [outs]):
        if isinstance(t, torch.Tensor):
            t.record_stream(ws)


def _late_mode():
    """VARSEP_FUSED_AFTER_ROLLOUT: '2' (default) = a held fused first-layer update (E_s's 20480 x 1200 weight-gradient + Adam launch: 640 MB
    of HBM traffic) waits ON ITS OWN LANE for the integrator's backward kernel to finish and then runs beside E_t's input-gradient chain, which
    leaves HBM idle (WaveEq step, same box: 1.4505 -> 1.4045 ms); '3' = every held fused update; '1' = on a stream of its own (2.95 ms: a
    fourth gradient stream collides with the integrator's queue); '0' = released with the other held work under the integrator's kernel."""
    return os.environ.get('VARSEP_FUSED_AFTER_ROLLOUT', '2')


def _late_fused(layer):
    m = _late_mode()
    return (m in ('1', '2') and layer == 0) or m == '3'


def run_deferred(fn, *inputs, outs=None, lane=0, late=False):
    """Run `fn()` (weight-gradient launches) on a gradient stream behind everything queued so far on the current stream.
    `outs`: the tensors `fn` writes, allocated by the caller -- required for the work to be holdable (hold_deferred); they are
    returned in place of fn's result."""
    if not _SIDE['on']:
        out = fn()
        return out if outs is None else outs
    if _SIDE['hold'] and outs is not None:
        # (with the stream the inputs are produced on: a chain whose backward runs on a stream of its own -- E_s, train.py -- is not covered by
        #  the wait for the stream the hold was declared on)
        _SIDE['held'].append((fn, inputs, outs, ('late', lane) if late else lane, torch.cuda.current_stream()))
        return outs
    main, ws = torch.cuda.current_stream(), _lane_stream(lane)
    ws.wait_stream(main)
    with torch.cuda.stream(ws):
        out = fn()
    _queued_on_lane(fn, lane)
    if outs is not None:
        out = outs
    _record_on(ws, inputs, out if outs is not None else ())
    for t in (out if isinstance(out, (list, tuple)) else [out]):
        if isinstance(t, torch.Tensor):
            t.record_stream(main)
    return out


def run_late(fn, *inputs, outs, lane=0):
    """Deferred work that must not be recorded NOW: it is launched by join_side_streams(), i.e. after the rest of backward has been
    issued.  The integrator's own weight gradients use it: recorded right behind the integrator's backward kernel they end up, in
    the replayed hipGraph, on the same queue as E_t's input-gradient chain and AHEAD of it (120 us of small GEMMs in front of
    the critical path of the WaveEq step)."""
    if not _SIDE['on']:
        fn()
        return outs
    _SIDE['late'].append((fn, inputs, outs, lane, torch.cuda.current_stream()))
    return outs


def defer_call(fn, late=False):
    """Queue `fn` (launches that consume held gradients, e.g. an optimizer bucket) behind ALL the held work; False if nothing is held.
    `late`: additionally behind the event recorded after the integrator's backward kernel (release_deferred(late_after=...))."""
    if not deferred_held():
        return False
    _SIDE['held'].append((fn, (), (), 'all-late' if late else None, None))
    return True


def _behind_producer(ws, producer, main, behind):
    if producer is not None and producer != main and producer != ws and (ws, producer) not in behind:
        ws.wait_stream(producer)
        behind.add((ws, producer))


def release_deferred(after=None, late_after=None):
    """Launch everything collected since hold_deferred() on the gradient streams, in order, behind the work queued so far on the
    stream the hold was declared on (the producer of every input of the held closures) and behind the event `after`."""
    held, _SIDE['held'] = _SIDE['held'], []
    was, _SIDE['hold'] = _SIDE['hold'], False
    if not held:
        return
    main = _SIDE['hold_main'] if (was and _SIDE['hold_main'] is not None) else torch.cuda.current_stream()
    started = set()
    behind = set()                             # (gradient stream, producer stream) pairs already ordered
    for fn, inputs, outs, lane, producer in held:
        if isinstance(lane, tuple):            # ('late', lane): behind the event `late_after` (recorded after the integrator's kernel)
            own = _late_mode() == '1'          # '1': a stream of its own; '2': the closure's own lane
            key = 'late' if own else lane[1]
            ws = _lane_stream(N_LANES) if own else _lane_stream(lane[1])
            if key not in started:
                ws.wait_stream(main)
                if after is not None and not own:
                    ws.wait_event(after)
                started.add(key)
            _behind_producer(ws, producer, main, behind)
            if late_after is not None:
                ws.wait_event(late_after)
            with torch.cuda.stream(ws):
                fn()
            _queued_on_lane(fn, N_LANES if own else lane[1])
            _record_on(ws, inputs, outs)
            continue
        if lane is None or lane == 'all-late':  # consumes everything released so far: lane 0 behind the other lanes
            ws = _lane_stream(0)
            if 0 not in started:
                ws.wait_stream(main)
                started.add(0)
            for l in started:
                if l != 0 and l != 'late':
                    ws.wait_stream(_lane_stream(l))
            if lane == 'all-late' and late_after is not None:
                ws.wait_event(late_after)
        else:
            ws = _lane_stream(lane)
            if lane not in started:
                ws.wait_stream(main)
                if after is not None:
                    ws.wait_event(after)
                started.add(lane)
            _behind_producer(ws, producer, main, behind)
        with torch.cuda.stream(ws):
            fn()
        _queued_on_lane(fn, 0 if (lane is None or lane == 'all-late') else lane)
        _record_on(ws, inputs, outs)


# ---- gradient destinations ------------------------------------------------------------------------------------------------------
# Graph-replay data parallelism keeps every `param.grad` as a view into a flat all-reduce bucket.  Handing gradients to
# autograd then costs one `+=` launch per parameter (3 x 243 MB of traffic on the WaveEq model) and forbids the side streams
# (the add would read a weight gradient its stream has not finished).  With destinations registered, the Linear chains write
# their weight gradients (GEMM output) and bias gradients (column sums into the zeroed bucket) straight into those views and
# return nothing for them.
_GRAD_OUT = {}


def set_grad_outputs(mapping):
    """mapping: {parameter: fp32 tensor of its shape that must receive its gradient} or None to clear."""
    _GRAD_OUT.clear()
    if mapping:
        for prm, view in mapping.items():
            assert view.shape == prm.shape and view.dtype in (torch.float32, torch.bfloat16) and view.is_contiguous()
            _GRAD_OUT[id(prm)] = view


def grad_output(prm):
    return _GRAD_OUT.get(id(prm)) if _GRAD_OUT else None


# Convolution families under a gradient reducer.  A convolution / BatchNorm parameter receives several contributions per step (the decoder
# runs for the auto-encoding pair and for the rollout, the SST integrator's blocks once per predicted frame), so its destination is
# ACCUMULATED into: the reducer zeroes its flat buckets at the start of a step, every contribution -- the first one included -- is added
# to the parameter's bucket view (weight gradients in the epilogue of their kernel, the small vectors by one multi-tensor add per
# block, the batched weight gradients of repeatedly applied convolutions by the end-of-backward flush) and autograd is handed nothing
# for such a parameter: no AccumulateGrad launch, no per-call adds, and the step under a reducer runs the same kernels as without one.
_CONV_GRAD_OUT = {}


def set_conv_grad_outputs(mapping):
    """mapping: {conv / BatchNorm parameter: zero-initialised fp32 tensor of its shape that accumulates its gradient} or None to clear."""
    _CONV_GRAD_OUT.clear()
    if mapping:
        for prm, view in mapping.items():
            assert view.shape == prm.shape and view.dtype == torch.float32 and view.is_contiguous()
            _CONV_GRAD_OUT[id(prm)] = (prm, view)


def conv_grad_output(prm):
    if not _CONV_GRAD_OUT or prm is None:
        return None
    ent = _CONV_GRAD_OUT.get(id(prm))
    return ent[1] if ent is not None and ent[0] is prm else None


def conv_parameters(net):
    """The parameters of `net` whose gradients are produced by ConvBlock / ConvResBlockFn (Conv2d, ConvTranspose2d, BatchNorm2d)."""
    import torch.nn as nn
    out = []
    for m in net.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d, nn.BatchNorm2d)):
            out += [p for p in m.parameters(recurse=False) if p.requires_grad]
    return out


# A gradient destination may be a bf16 tensor (the reducer's wire image): the weight-gradient GEMM rounds once in its epilogue,
# the all-reduce averages that image in place and the optimizer reads it (optim.Adam consults lowp_gradient), so the fp32
# gradient of such a weight never exists and neither do the casts to and from the wire format.
_LOWP_GRAD = {}


def set_lowp_gradients(mapping):
    _LOWP_GRAD.clear()
    if mapping:
        for prm, view in mapping.items():
            assert view.shape == prm.shape and view.dtype == torch.bfloat16 and view.is_contiguous()
            _LOWP_GRAD[id(prm)] = view


def lowp_gradient(prm):
    return _LOWP_GRAD.get(id(prm)) if _LOWP_GRAD else None


# ---- optimizer fused into the weight gradient -----------------------------------------------------------------------------------
# optim.Adam.fuse_into_wgrad(params) registers 2-D Linear weights whose weight-gradient GEMM applies the Adam update in its epilogue
# (ops.gemm_adam): the gradient of such a weight is never stored and autograd gets nothing for it.  Valid where the chain's
# backward produces THE gradient of the step (one contribution, no all-reduce, no loss scaling) -- the recorded single-GPU MLP step.
_FUSED_OPT = {}


def set_fused_optimizer(mapping):
    """mapping: {parameter: optimizer with fused_update(parameter, a, la, b, lb, M, N, K)} or None to clear."""
    _FUSED_OPT.clear()
    if mapping:
        for prm, opt in mapping.items():
            _FUSED_OPT[id(prm)] = (prm, opt)


def fused_optimizer(prm):
    ent = _FUSED_OPT.get(id(prm)) if _FUSED_OPT else None
    return ent[1] if ent is not None and ent[0] is prm else None


def side_streams_in_use():
    """Streams that deferred gradient work of the current step may still be running on."""
    return (list(_SIDE['lanes']) + [_SIDE[k] for k in ('rollout', 'es') if _SIDE[k] is not None]) if _SIDE['on'] else []


def finish_join(pending):
    """Second half of join_side_streams(partial=True): wait for the lanes whose fused updates were still running."""
    cur = torch.cuda.current_stream()
    for ws in pending or ():
        cur.wait_stream(ws)


def join_side_streams(partial=False):
    """Make the current stream wait for all deferred gradient work (call before the optimizer step).  `partial`: on lanes that recorded a
    marker in front of a trailing fused update (tail_fused_updates) wait for the marker only and return those lanes: the caller runs the
    optimizer launch for the other parameters and then calls finish_join()."""
    if not _SIDE['on']:
        return []       # nothing was deferred; waiting on a stream outside the running capture would break the capture
    release_deferred()  # (a step without the integrator's backward never reached the release point)
    late, _SIDE['late'] = _SIDE['late'], []
    for fn, inputs, outs, lane, producer in late:
        if tail_split() and N_LANES > 1:
            lane = N_LANES - 1                # (lane 0 carries the last chain's 640 MB update: the late gradients would queue behind it)
        ws = _lane_stream(lane)
        ws.wait_stream(producer)
        with torch.cuda.stream(ws):
            fn()
        _queued_on_lane(fn, lane)
        _record_on(ws, inputs, outs)
    cur = torch.cuda.current_stream()
    pending = []
    markers = _SIDE.get('markers') or {}
    for i, ws in enumerate(_SIDE['lanes']):
        ev = markers.get(i) if partial else None
        if ev is not None:
            cur.wait_event(ev)
            pending.append(ws)
        else:
            cur.wait_stream(ws)
    _SIDE['markers'] = {}
    for k in ('rollout', 'es'):
        if _SIDE[k] is not None:
            cur.wait_stream(_SIDE[k])
    return pending


# ------------------------------------------------------------------------------------------------ weight shadows
# Every derived copy of a parameter (16-bit operand copy, MFMA pre-pack, eval-mode BatchNorm fold) is valid for one (version counter,
# replay epoch) pair.  The version counter moves when torch or optim.Adam updates the parameter from Python; it does NOT move when a
# recorded step is replayed (`increment_version` ran at capture time only) nor when a kernel writes BatchNorm running statistics through
# raw pointers.  `note_replay()` (train.GraphedStep.step) and `count_bn_calls` (every training-mode BatchNorm call) advance the epochs
# instead, so an eager use after replays -- evaluation between recorded training steps, a re-recording after a learning-rate change --
# re-derives what the replays have outdated (into the same storage: recorded graphs keep addressing it).
_EPOCH = {'replay': 0, 'bn': 0}
_shadow = {}


def _ver(p):
    return (p._version, _EPOCH['replay'])


def _wref(cache, key, p):
    """Weak reference to the tensor a cache entry belongs to; the entry (and the device memory of its copy) goes when the tensor does."""
    return weakref.ref(p, lambda _r: cache.pop(key, None))


def note_replay():
    """A recorded step has been replayed: parameters and BatchNorm statistics changed without moving any version counter."""
    _EPOCH['replay'] += 1
    _EPOCH['bn'] += 1


def shadow(p, dtype):
    """Parameter as a GEMM operand: itself in fp32 mode, a cached bf16 copy (refreshed on version change) otherwise."""
    if dtype == torch.float32:
        return p.detach()
    key = id(p)
    ent = _shadow.get(key)
    if ent is None or ent[0] != _ver(p) or ent[1].data_ptr() == 0 or ent[2]() is not p or ent[1].dtype != dtype:
        buf = ent[1] if ent is not None and ent[2]() is p and ent[1].shape == p.shape and ent[1].dtype == dtype else None
        buf = ops.cast(p.detach(), dtype, out=buf)
        _shadow[key] = (_ver(p), buf, _wref(_shadow, key, p))
        return buf
    return ent[1]


def adopt_shadow(p, buf):
    """Make `buf` (a 16-bit tensor of p's shape, e.g. a view into the sharded optimizer's arena: parallel.GradAllReducer) THE operand copy
    of `p`: filled from the current fp32 value now, rewritten in place by the optimizer launches and the arena's all-gather afterwards."""
    assert buf.shape == p.shape and buf.dtype in (torch.bfloat16, torch.float16) and buf.is_contiguous()
    ops.cast(p.detach(), buf.dtype, out=buf)
    _shadow[id(p)] = (_ver(p), buf, _wref(_shadow, id(p), p))


def shadow_buffer_for_update(p):
    """The live bf16 operand copy of `p`, if one exists: the optimizer kernel rewrites it in the pass that updates `p`."""
    ent = _shadow.get(id(p))
    if ent is not None and ent[2]() is p and ent[1].shape == p.shape and ent[1].dtype in (torch.bfloat16, torch.float16) and ent[1].is_contiguous():
        return ent[1]
    return None


def shadows_written(params):
    """Called after an optimizer kernel refreshed the shadows of `params` in place: mark them current."""
    for p in params:
        ent = _shadow.get(id(p))
        if ent is not None and ent[2]() is p:
            _shadow[id(p)] = (_ver(p), ent[1], ent[2])


def refresh_shadows(params, dtype=None):
    """Re-cast every shadow in place (same storage) -- the form used inside a captured hipGraph step."""
    dtype = dtype or compute_dtype()
    for p in params:
        key = id(p)
        ent = _shadow.get(key)
        if ent is not None and ent[2]() is p:
            ops.cast(p.detach(), dtype, out=ent[1])
            _shadow[key] = (_ver(p), ent[1], ent[2])


def invalidate_shadows(params):
    """Mark the operand copies (and every weight pre-pack) of `params` stale: their next use re-casts / re-packs into the same storage.  A
    recorded step whose optimizer does not maintain the copies itself calls this right before the capture, so that the recording contains
    the casts."""
    for p in params:
        ent = _shadow.get(id(p))
        if ent is not None and ent[2]() is p:
            _shadow[id(p)] = ((-1, -1), ent[1], ent[2])
        for cache in (_packed, _packed_conv, _packed_tap, _packed_k3, _packed_img, _packed_k4s2):
            for key, e in list(cache.items()):
                if e[2]() is p:
                    cache[key] = ((-1, -1), e[1], e[2])


def to_compute(x, dtype):
    x = x.detach()
    if x.dtype == dtype:
        return x if x.is_contiguous() else x.contiguous()
    return ops.cast(x, dtype)


# ------------------------------------------------------------------------------------------------ Linear chains
class GradHandoff:
    """Side channel from the Function that consumes a chain's output to the chain's backward.

    When the consumer is the fused loss (TrainLosses), d loss / d output is `k * (y - target)` and the chain's first backward
    step multiplies it by act'(y): the loss kernel can write that product directly in the compute dtype (one pass, no fp32
    gradient of the frame stack in HBM).  Autograd still wants an fp32 gradient for the output, so the consumer returns a
    zero placeholder (a 0-stride view of one resident zero, no kernel) and leaves the real tensor here; if other consumers
    add gradients of their own, the chain sees a non-placeholder `dy` and adds its activation gradient on top."""
    __slots__ = ('act', 'cdt', 'dz', 'fuse', 'fused')
    _zero = {}

    def __init__(self):
        self.act, self.cdt, self.dz = None, None, None
        # fuse: what the chain's LAST GEMM needs to evaluate the frame losses in its epilogue (train._compute_losses_mlp_batched sets it for
        # the recorded step): dict(full, idx, G, s_old, s_new, t0, lambdas, average, up); fused: the results (out, dz, ds_old, ds_new, dt0)
        self.fuse, self.fused = None, None

    @classmethod
    def placeholder(cls, like):
        return cls.zeros(like.shape, like.device)

    @classmethod
    def zeros(cls, shape, device):
        z = cls._zero.get(device)
        if z is None:
            z = cls._zero[device] = torch.zeros((), dtype=torch.float32, device=device)
        return z.expand(tuple(shape))

    @staticmethod
    def is_placeholder(t):
        return all(st == 0 for st in t.stride()) and t.numel() > 1


class MLPChain(torch.autograd.Function):
    """y = act_L(W_L ... act_1(W_1 x + b_1) ... + b_L) with every bias/activation fused into the GEMM epilogues.

    Reference: networks/mlp.py:66-75 (`MLP.forward`; ReLU is applied BEFORE each non-first Linear, which is the same
    as applying it to the output of every Linear but the last) plus the optional trailing activation of
    mlp_encdec.py:49.  Backward fuses each activation derivative into the input-gradient GEMM as an output mask.
    """

    @staticmethod
    def forward(ctx, x, x_lowp, acts, handoff, *params):
        fs = _SIDE.get('chain_stream')
        if fs is not None and fs != torch.cuda.current_stream():
            caller = torch.cuda.current_stream()
            _SIDE['chain_stream'] = None
            try:
                with torch.cuda.stream(fs):
                    out = MLPChain.forward(ctx, x, x_lowp, acts, handoff, *params)
            finally:
                _SIDE['chain_stream'] = fs
            for t in (out,) + tuple(ctx.to_save):
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(caller)
            return out
        require_cuda(x)
        cdt = compute_dtype()
        n_layers = len(params) // 2
        M = x.shape[0]
        # x_lowp: the same rows already in the compute dtype (MixCodes writes them while it builds x), saves the cast pass
        h = x_lowp if x_lowp is not None and x_lowp.dtype == cdt else to_compute(x, cdt)
        saved = [h]
        for l in range(n_layers):
            W, b = params[2 * l], params[2 * l + 1]
            N, K = W.shape
            last = l == n_layers - 1
            if last and handoff is not None and handoff.fuse is not None and cdt != torch.float32:
                # the frames are consumed by the fused losses only (recorded step): compare them with their targets in the GEMM's
                # epilogue instead of storing 4 B per element and reading them back (ops.gemm_frame_loss)
                f = handoff.fuse
                res = ops.gemm_frame_loss(h, shadow(W, cdt), b.detach() if b is not None else None, acts[l], f['full'], f['idx'], f['G'],
                                          f['s_old'], f['s_new'], f['t0'], f['lambdas'], f['average'], f['up'], cdt)
                if res is not None:
                    handoff.fused = res
                    h = GradHandoff.zeros((M, N), h.device)          # (no frames exist: a 0-stride view of one resident zero)
                    saved.append(GradHandoff.zeros((), h.device))    # (the resident zero: `new_zeros` is a fill launch per step)
                    continue
            h = ops.gemm(h, R, shadow(W, cdt), R, M, N, K, bias=b.detach() if b is not None else None, act=acts[l],
                         out_dtype=torch.float32 if last else cdt)
            saved.append(h)
        ctx.acts, ctx.cdt, ctx.n_layers = acts, cdt, n_layers
        ctx.handoff = handoff
        if handoff is not None:
            handoff.act, handoff.cdt, handoff.dz = acts[n_layers - 1], cdt, None
        ctx.x_needs_grad = x.requires_grad
        ctx.params = params
        ctx.save_for_backward(*saved)
        return h

    @staticmethod
    def backward(ctx, dy):
        saved = ctx.saved_tensors
        acts, cdt, L, params = ctx.acts, ctx.cdt, ctx.n_layers, ctx.params
        M = dy.shape[0]
        handed = None
        if ctx.handoff is not None and ctx.handoff.dz is not None:
            handed, ctx.handoff.dz = ctx.handoff.dz.view(M, -1), None
        if handed is not None and GradHandoff.is_placeholder(dy):
            dz = handed                                  # the consumer already applied act'(y) and the cast
        else:
            if saved[L].dim() == 0:
                raise VarsepHipError('the frames of this chain were never stored (losses evaluated in the GEMM epilogue): their only '
                                     'consumer may be the fused loss')
            dy = dy.contiguous()
            if acts[L - 1] not in ('none', None):
                dz = ops.act_bwd(dy, saved[L], acts[L - 1], out_dtype=cdt)
            else:
                dz = to_compute(dy, cdt)
            if handed is not None:
                dz = dz + handed
        grads = [None] * (2 * L)
        dx = None
        lane = next_lane()                   # this chain's weight / bias gradients: one gradient stream, in order
        tail_lane = lane
        if (lane > 0 and deferred_held() and _late_mode() == '2' and tail_fused_updates() and L > 1 and params[0].requires_grad
                and grad_output(params[0]) is None and fused_optimizer(params[0]) is not None and cdt != torch.float32
                and os.environ.get('VARSEP_LATE_TAIL_ALONE', '0') == '1'):
            # A chain whose fused first-layer update is HELD for the end of the integrator's backward kernel (E_s in the batched MLP step: 640 MB
            # of HBM traffic that must not run beside that latency-bound kernel) keeps its lane for that update alone; its small weight gradients
            # and bias sums ride on the previous chain's lane.  In the recording the update then depends on the chain's input gradient and on
            # the integrator's kernel only -- not on its own small gradients, which sat behind the decoder's weight gradients on a shared
            # hardware queue and started it ~180 us after the integrator's kernel had finished (timeline of round 6) -- so it runs in the
            # HBM-idle window beside E_t's input-gradient chain instead of beside E_t's own 640 MB update.  MEASURED AND NOT THE DEFAULT
            # (VARSEP_LATE_TAIL_ALONE=1 turns it on): 1.2521 / 1.2337 vs 1.2047 / 1.1930 ms -- E_t's input-gradient chain, the critical path
            # after the integrator, loses more under the update's HBM traffic than the earlier start of the tail gives back.
            lane = lane - 1
        bias_jobs = []                       # (slot, dz): all bias gradients of the chain in one launch at the end
        tail_job = None
        for l in range(L - 1, -1, -1):
            W, b = params[2 * l], params[2 * l + 1]
            N, K = W.shape
            h_in = saved[l]

            def input_gradient(dz=dz, W=W, h_in=h_in, l=l, K=K, N=N):
                masked = acts[l - 1] not in ('none', None)
                return ops.gemm(dz, R, shadow(W, cdt), S, M, K, N, out_dtype=cdt, mask=h_in if masked else None,
                                mask_act=acts[l - 1] if masked else 'none')
            dz_next = None
            if W.requires_grad:               # dW = dz^T h_in (fp32); off the critical path -> wgrad stream when enabled
                dst = grad_output(W)
                fused = fused_optimizer(W) if dst is None else None
                if fused is not None and cdt != torch.float32 and fused.can_fuse(W, dz, h_in):
                    # the fused launch REWRITES W and its 16-bit copy: the input gradient of this layer reads that copy, so it is
                    # launched first; the update (run now, on a lane behind the current stream, or held) is ordered after it
                    if l > 0:
                        dz_next = input_gradient()
                    elif ctx.x_needs_grad:
                        dx = ops.gemm(dz, R, shadow(W, cdt), S, M, K, N, out_dtype=torch.float32)
                    # the weight-gradient GEMM's epilogue IS this weight's optimizer step; nothing is stored, autograd gets nothing
                    job = (lambda dz=dz, h_in=h_in, N=N, K=K, W=W, fused=fused: fused.fused_update(W, dz, S, h_in, S, N, K, M), (dz, h_in))
                    if l == 0 and _SIDE['on'] and tail_fused_updates():
                        tail_job = job                   # issued behind the chain's bias sums and a marker (see tail_fused_updates)
                    else:
                        run_deferred(job[0], *job[1], outs=(), lane=lane, late=_late_fused(l))
                elif dst is not None:           # straight into the all-reduce bucket; autograd gets nothing for this parameter
                    # (holdable like the single-GPU path: the destination exists already)
                    run_deferred(lambda dz=dz, h_in=h_in, N=N, K=K, dst=dst: ops.gemm(dz, S, h_in, S, N, K, M, out=dst), dz, h_in,
                                 outs=dst, lane=lane)
                else:
                    # autograd keeps the tensor it is handed only if nobody else references it (it CLONES it otherwise -- here
                    # before the deferred GEMM has written it): the closure writes through a second tensor on the same storage
                    buf = torch.empty((N * K,), dtype=torch.float32, device=dz.device)
                    grads[2 * l] = buf.view(N, K)
                    run_deferred(lambda dz=dz, h_in=h_in, N=N, K=K, dw=buf.view(N, K): ops.gemm(dz, S, h_in, S, N, K, M, out=dw),
                                 dz, h_in, outs=buf, lane=lane)
            if b is not None and b.requires_grad:
                bias_jobs.append((2 * l + 1, dz, grad_output(b)))
            if l > 0:
                dz = dz_next if dz_next is not None else input_gradient()
            elif ctx.x_needs_grad and dx is None:
                dx = ops.gemm(dz, R, shadow(W, cdt), S, M, K, N, out_dtype=torch.float32)
        direct = [j for j in bias_jobs if j[2] is not None]
        bias_jobs = [j for j in bias_jobs if j[2] is None]
        if direct:                            # added to the (zeroed) bucket slices
            dzs, dsts = [j[1] for j in direct], [j[2] for j in direct]
            run_deferred(lambda: ops.colsum_multi(dzs, outs=dsts), *dzs, outs=dsts, lane=lane)
        if bias_jobs:
            dzs = [j[1] for j in bias_jobs]
            flat, views = ops.colsum_alloc(dzs)
            for (slot, _, _), db in zip(bias_jobs, views):
                grads[slot] = db
            views = ops.colsum_alloc(dzs, flat)[1]       # the closure's own views (see the weight gradients above)
            run_deferred(lambda views=views: ops.colsum_multi(dzs, outs=views, zero_flat=flat), *dzs, outs=flat, lane=lane)
        if tail_job is not None:
            tail_job[0]._vs_tail = True
            tl = tail_lane
            if tail_split() and tail_lane == N_LANES - 1 and N_LANES > 1 and not deferred_held():
                # the LAST chain of backward (E_t in the batched MLP step: lane N_LANES - 1): its small weight gradients and bias sums (~90 us of
                # 10 us launches on `lane`) run BESIDE its 640 MB first-layer update instead of in front of it -- the update goes to lane 0, whose
                # work (the decoder's weight gradients) is long done; the integrator's late weight gradients move to `lane` (join_side_streams)
                tl = 0
            run_deferred(_lane_marker(tl), outs=(), lane=tl)
            run_deferred(tail_job[0], *tail_job[1], outs=(), lane=tl, late=_late_fused(0))
        return (dx, None, None, None) + tuple(grads)


def mlp_chain(x, linears, hidden_act='relu', out_act='none', x_lowp=None, handoff=None):
    """Run a stack of nn.Linear parameter holders as one fused chain."""
    acts = [hidden_act] * (len(linears) - 1) + [out_act]
    params = []
    for lin in linears:
        params += [lin.weight, lin.bias]
    return MLPChain.apply(x, x_lowp, tuple(acts), handoff, *params)


# ------------------------------------------------------------------------------------------------ fused rollout
_packed = {}


def packed_weight(p, dtype, transpose):
    """Rollout pre-pack of a 2-D parameter (fragment order, compute dtype), cached per parameter version."""
    key = (id(p), bool(transpose), dtype)
    ent = _packed.get(key)
    if ent is None or ent[0] != _ver(p) or ent[2]() is not p:
        buf = ent[1] if ent is not None and ent[2]() is p else None
        buf = ops.pack_rollout_weight(p.detach().contiguous(), dtype, transpose, out=buf)
        _packed[key] = (_ver(p), buf, _wref(_packed, key, p))
        return buf
    return ent[1]


def prepack_weights(requests, dtype):
    """Bring several rollout pre-packs up to date with ONE launch: requests = [(parameter, transpose)]."""
    stale = []
    for p, tr in requests:
        key = (id(p), bool(tr), dtype)
        ent = _packed.get(key)
        if ent is None or ent[0] != _ver(p) or ent[2]() is not p:
            stale.append((key, p, tr, ent[1] if ent is not None and ent[2]() is p else None))
    if stale:
        bufs = ops.pack_rollout_weights([(p.detach().contiguous(), tr, buf) for _, p, tr, buf in stale], dtype)
        for (key, p, _, _), buf in zip(stale, bufs):
            _packed[key] = (_ver(p), buf, _wref(_packed, key, p))


def _stacked_grad_outputs(params, nb):
    """([nb, H, C], [nb, H, H], [nb, C, H] fp32 views over the registered gradient destinations of the integrator's weights, [bias views in
    (block, layer) order]) when every parameter has a destination and the blocks' weights of each layer lie back to back; None otherwise."""
    outs = [grad_output(p) for p in params]
    if any(o is None or o.dtype != torch.float32 for o in outs):
        return None
    stacks = []
    for l in range(3):
        ws = [outs[6 * b + 2 * l] for b in range(nb)]
        n = ws[0].numel()
        if any(w.data_ptr() != ws[0].data_ptr() + 4 * n * b or w.shape != ws[0].shape for b, w in enumerate(ws)):
            return None
        stacks.append(ws[0].as_strided((nb,) + tuple(ws[0].shape), (n,) + tuple(ws[0].stride())))
    bias = [outs[6 * b + 2 * l + 1] for b in range(nb) for l in range(3)]
    return stacks[0], stacks[1], stacks[2], bias


class MLPRollout(torch.autograd.Function):
    """All n-1 integrator steps x n_blocks residual MLP blocks in one persistent kernel per direction.

    Reference: networks/model.py:78-83 + networks/resnet.py:22-50.  Returns (t_codes [B,n,C], residuals
    [n-1,n_blocks,B,C]); the residuals are returned for API compatibility and are not differentiable on this fused
    path (no caller of the reference uses them; the per-step `MLPResnet.forward` keeps them differentiable).
    """

    @staticmethod
    def forward(ctx, x0, n_steps, *params):
        require_cuda(x0)
        cdt = compute_dtype()
        nb = len(params) // 6
        H = params[0].shape[0]
        ws, bs = [], []
        if torch.is_grad_enabled() or any(p.requires_grad for p in params):
            # forward and transposed (backward) packs of every block in one launch
            prepack_weights([(params[6 * b + 2 * l], tr) for b in range(nb) for l in range(3) for tr in (False, True)], cdt)
        for b in range(nb):
            for l in range(3):
                ws.append(packed_weight(params[6 * b + 2 * l], cdt, False))
                bs.append(params[6 * b + 2 * l + 1].detach())
        t_codes, residuals, (xin, h1, h2, m1, m2) = ops.mlp_rollout_fwd(x0.detach().float().contiguous(), ws, bs, n_steps, H)
        ctx.cdt, ctx.nb, ctx.n_steps, ctx.params = cdt, nb, n_steps, params
        ctx.save_for_backward(xin, h1, h2, m1, m2)
        ctx.mark_non_differentiable(residuals)
        # VARSEP_ROLLOUT_GRES_FILL=0: no zero tensor for the gradient of the non-differentiable residuals (autograd launches a 1.2 MB fill for it
        # on the integrator's stream, right in front of the backward kernel).  MEASURED AND NOT THE DEFAULT: without that node the replayed WaveEq
        # step is 1.282 / 1.283 ms against 1.145 / 1.151 (same box, alternating).  It is not a race between the held weight-gradient GEMMs and
        # the backward kernel for the CUs: holding the GEMMs back by one to three small launches behind `ready` leaves it at 1.29-1.30 ms.  The
        # node changes where the runtime places the branches of the recording (the same kind of cliff as a sixth stream or six hardware
        # queues, profiles/r06_queues.md), so it stays.
        if os.environ.get('VARSEP_ROLLOUT_GRES_FILL', '1') == '0':
            ctx.set_materialize_grads(False)
        ctx.codes_shape = tuple(t_codes.shape)
        return t_codes, residuals

    @staticmethod
    def backward(ctx, g_codes, _g_res):
        xin, h1, h2, m1, m2 = ctx.saved_tensors
        cdt, nb, n_steps, params = ctx.cdt, ctx.nb, ctx.n_steps, ctx.params
        if g_codes is None:
            g_codes = torch.zeros(ctx.codes_shape, dtype=torch.float32, device=xin.device)
        wts = []
        for b in range(nb):
            W1, W2, W3 = params[6 * b], params[6 * b + 2], params[6 * b + 4]
            wts += [packed_weight(W3, cdt, True), packed_weight(W2, cdt, True), packed_weight(W1, cdt, True)]
        g_codes = g_codes.contiguous().float()
        ready = None
        if deferred_held():
            # held weight gradients (decoder, E_s) fill the chip under the latency-bound kernel launched next.  They become
            # eligible together with it, not earlier: a 600-workgroup GEMM that starts first keeps the kernel's 192 workgroups
            # (all of which must be resident before the chain moves) waiting for register space for most of its run time
            ready = torch.cuda.Event()
            ready.record()
        dx0, dr, dh2, dh1 = ops.mlp_rollout_bwd(g_codes, wts, h1, h2, m1, m2, n_steps)
        done = None
        if ready is not None and (_late_mode() in ('1', '2', '3') or os.environ.get('VARSEP_ADAM_EARLY_BUCKET') == '2'):
            done = torch.cuda.Event()
            done.record()
        release_deferred(after=ready, late_after=done)
        B, C = dx0.shape
        H = h1.shape[-1]
        rows = (n_steps - 1) * B
        if rows == 0:
            return (dx0, None) + tuple(torch.zeros_like(p) for p in params)

        bias_in = []
        for b in range(nb):
            bias_in += [dh1[b].view(rows, H), dh2[b].view(rows, H), dr[b].view(rows, C)]

        def weight_grads(dW1=None, dW2=None, dW3=None, bias_out=None, bias_flat=None):
            # the three weight gradients of ALL blocks: one batched launch per layer (the saves are [block][step*B][feature])
            dW1 = ops.gemm_batched(dh1.view(nb, rows, H), S, xin.view(nb, rows, C), S, H, C, rows, out=dW1)
            dW2 = ops.gemm_batched(dh2.view(nb, rows, H), S, h1.view(nb, rows, H), S, H, H, rows, out=dW2)
            dW3 = ops.gemm_batched(dr.view(nb, rows, C), S, h2.view(nb, rows, H), S, C, H, rows, out=dW3)
            dbs = ops.colsum_multi(bias_in, outs=bias_out, zero_flat=bias_flat)
            grads = []
            for b in range(nb):
                grads += [dW1[b], dbs[3 * b], dW2[b], dbs[3 * b + 1], dW3[b], dbs[3 * b + 2]]
            return grads
        if _GRAD_OUT:
            stacked = _stacked_grad_outputs(params, nb)
            if stacked is not None:
                # the reducer laid the blocks' weights of one layer back to back (GradAllReducer(stacked=...)) and zeroes the bucket's tail at
                # the start of the step: the three batched launches write [blocks, ., .] IN the bucket, the bias sums add into their views --
                # autograd is handed nothing (no 18 `+=` launches of 4 us each at the end of backward: 86 us of the WaveEq step under a reducer)
                w1, w2, w3, bias_out = stacked
                if _SIDE['on'] and os.environ.get('VARSEP_ROLLOUT_WGRAD_LATE', '1') == '1':
                    run_late(lambda: weight_grads(w1, w2, w3, bias_out, None), dr, dh2, dh1, xin, h1, h2, outs=(w1, w2, w3), lane=next_lane())
                else:
                    weight_grads(w1, w2, w3, bias_out, None)
                return (dx0, None) + (None,) * len(params)
            # with gradient destinations registered these gradients still go through autograd's `+=` into the bucket views, which
            # runs on THIS node's stream: compute them here, not on a gradient stream
            grads = weight_grads()
        elif _SIDE['on'] and os.environ.get('VARSEP_ROLLOUT_WGRAD_LATE', '1') == '1':
            # recorded at the END of backward (run_late), into buffers allocated now; autograd is handed views of its own (a tensor
            # somebody else references would be cloned, here before it has been written)
            dev = dx0.device
            w1, w2, w3 = (torch.empty(shape, dtype=torch.float32, device=dev) for shape in ((nb, H, C), (nb, H, H), (nb, C, H)))
            flat, views = ops.colsum_alloc(bias_in)
            grads = []
            for b in range(nb):
                grads += [w1[b], views[3 * b], w2[b], views[3 * b + 1], w3[b], views[3 * b + 2]]
            mine = ops.colsum_alloc(bias_in, flat)[1]
            del views
            run_late(lambda a=w1.view(nb, H, C), b_=w2.view(nb, H, H), c=w3.view(nb, C, H): weight_grads(a, b_, c, mine, flat),
                     dr, dh2, dh1, xin, h1, h2, outs=(w1, w2, w3, flat), lane=next_lane())
            del w1, w2, w3
        else:
            grads = run_deferred(weight_grads, dr, dh2, dh1, xin, h1, h2, lane=next_lane())
        return (dx0, None) + tuple(grads)


# ------------------------------------------------------------------------------------------------ conv blocks
_packed_conv = {}


def packed_conv_weight(p, dtype, stride, pad):
    """Transposed-form pre-pack of a conv weight (ops.conv_pack_weight), cached per parameter version."""
    key = (id(p), dtype, stride, pad)
    ent = _packed_conv.get(key)
    if ent is None or ent[0] != _ver(p) or ent[2]() is not p:
        buf = ent[1] if ent is not None and ent[2]() is p else None
        buf = ops.conv_pack_weight(p.detach().contiguous(), dtype, stride, pad, out=buf)
        _packed_conv[key] = (_ver(p), buf, _wref(_packed_conv, key, p))
        return buf
    return ent[1]


_packed_tap = {}


def packed_tap_weight(p, dtype):
    """Tap-GEMM pre-pack of a ConvTranspose2d k4 s2 p1 weight (ops.convt_tap_pack_weight), cached per parameter version."""
    key = (id(p), dtype)
    ent = _packed_tap.get(key)
    if ent is None or ent[0] != _ver(p) or ent[2]() is not p:
        buf = ent[1] if ent is not None and ent[2]() is p else None
        buf = ops.convt_tap_pack_weight(p.detach().contiguous(), dtype, out=buf)
        _packed_tap[key] = (_ver(p), buf, _wref(_packed_tap, key, p))
        return buf
    return ent[1]


_packed_k3 = {}


def packed_k3_weight(p, dtype, flip):
    """Tap-GEMM pre-pack of a Conv2d k3 s1 p1 weight (forward, or flipped / transposed for the input gradient)."""
    key = (id(p), dtype, bool(flip))
    ent = _packed_k3.get(key)
    if ent is None or ent[0] != _ver(p) or ent[2]() is not p:
        buf = ent[1] if ent is not None and ent[2]() is p else None
        buf = ops.conv_k3_tap_pack_weight(p.detach().contiguous(), dtype, flip, out=buf)
        _packed_k3[key] = (_ver(p), buf, _wref(_packed_k3, key, p))
        return buf
    return ent[1]


_packed_k4s2 = {}


def packed_k4s2_weight(p, dtype):
    """Row-band pre-pack of a k4 s2 p1 weight over the parity planes of its gather operand (ops.conv_k4s2_pack_weight): a Conv2d weight
    [Cout, Cin, 4, 4] for its forward, a ConvTranspose2d weight [Cin, Cout, 4, 4] for its input gradient; cached per parameter version."""
    key = (id(p), dtype)
    ent = _packed_k4s2.get(key)
    if ent is None or ent[0] != _ver(p) or ent[2]() is not p:
        buf = ent[1] if ent is not None and ent[2]() is p else None
        buf = ops.conv_k4s2_pack_weight(p.detach().contiguous(), dtype, out=buf)
        _packed_k4s2[key] = (_ver(p), buf, _wref(_packed_k4s2, key, p))
        return buf
    return ent[1]


def _conv_lane():
    """Gradient stream of the next convolution weight gradient (VARSEP_CONV_WGRAD_LANES of them in turn, default 1)."""
    n = max(1, int(os.environ.get('VARSEP_CONV_WGRAD_LANES', '1')))
    i = _SIDE.get('conv_lane', 0)
    _SIDE['conv_lane'] = (i + 1) % n
    return i


def _conv_weight_grad(w, dz, xc, stride, pad, transposed, k4s2=None):
    """Weight gradient of one convolution call for autograd, or None when it was added to / will be batched into the tensor autograd
    already holds.  `k4s2` = (small map, parity planes of the large map): the k4 s2 p1 family on the row-band kernels (ops.conv_k4s2_wgrad)."""
    # a weight that already holds a gradient from an earlier call of this pass (the integrator's blocks: one call per predicted
    # frame) gets this call's contribution ADDED in the weight-gradient GEMM's epilogue -- no temporary, no add launch
    dw = None
    dst = conv_grad_output(w)
    first_w = dst if dst is not None else (_fold_slots().get(id(w)) if _STATE.get('fold_grads') else None)
    _STATE['wgrad_on_lane'] = False
    # a gradient that accumulates into a destination autograd never sees (dst) is nobody's input until the optimizer: with gradient streams on it
    # runs on one of them, beside the input-gradient / BatchNorm chain (train.GraphedStep joins the streams before the optimizer step)
    on_lane = (dst is not None and first_w is dst and _SIDE['on'] and not _defer_wgrad_ok(w, dz, transposed, stride)
               and dst.dtype == torch.float32 and dst.is_contiguous())
    if k4s2 is not None:
        if first_w is not None and first_w.dtype == torch.float32 and first_w.shape == w.shape and first_w.is_contiguous():
            if on_lane:
                run_deferred(lambda a=k4s2[0], b=k4s2[1]: ops.conv_k4s2_wgrad(a, b, w.shape, into=dst), k4s2[0], k4s2[1], outs=dst, lane=_conv_lane())
                _STATE['wgrad_on_lane'] = True
            else:
                ops.conv_k4s2_wgrad(k4s2[0], k4s2[1], w.shape, into=first_w)
            return None
        return ops.conv_k4s2_wgrad(k4s2[0], k4s2[1], w.shape)
    if _defer_wgrad_ok(w, dz, transposed, stride):
        slot = _DEFER_W['slots'].get(id(w))
        if slot is None:
            if first_w is not None and first_w.dtype == torch.float32 and first_w.shape == w.shape and first_w.is_contiguous():
                g = first_w                          # the weight already holds a gradient of this pass: the batch is added to it
            else:
                # zeros: the batched gradient is ADDED at the end, so other (non-deferred) calls of the same weight may add to
                # this tensor in between; autograd gets a tensor of its own on the same storage (it keeps an unshared tensor)
                buf = torch.zeros((w.numel(),), dtype=torch.float32, device=dz.device)
                g = buf.view(w.shape)
                dw = buf.view(w.shape)
            slot = _DEFER_W['slots'][id(w)] = {'g': g, 'pairs': [], 'shape': tuple(w.shape), 'stride': stride, 'pad': pad,
                                              'stream': torch.cuda.current_stream()}
            if not _DEFER_W['queued']:
                torch.autograd.Variable._execution_engine.queue_callback(flush_deferred_wgrads)
                _DEFER_W['queued'] = True
        slot['pairs'].append((dz, xc))
    elif first_w is not None and first_w.dtype == torch.float32 and first_w.shape == w.shape and first_w.is_contiguous():
        if on_lane:
            run_deferred(lambda: ops.conv_wgrad(dz, xc, w.shape, stride, pad, transposed, into=dst), dz, xc, outs=dst, lane=_conv_lane())
            _STATE['wgrad_on_lane'] = True
        else:
            ops.conv_wgrad(dz, xc, w.shape, stride, pad, transposed, into=first_w)
    else:
        dw = ops.conv_wgrad(dz, xc, w.shape, stride, pad, transposed)
    return dw


def _fold_param_grads(pairs):
    """pairs = ((parameter, gradient or None), ...) of one block -> the gradients to hand to autograd, in order.  In fold mode the FIRST
    contribution of a parameter in this backward pass goes to autograd (which keeps that very tensor as the parameter's pending
    gradient); later contributions of the pass are added INTO it with one multi-tensor launch per block and autograd gets nothing for
    them (otherwise: one add launch per parameter and contribution)."""
    out = [g for _, g in pairs]
    if not _STATE.get('fold_grads') and not _CONV_GRAD_OUT:
        return out
    acc = _fold_slots() if _STATE.get('fold_grads') else {}
    into, what = [], []
    for i, (prm, g) in enumerate(pairs):
        if g is None or prm is None:
            continue
        dst = conv_grad_output(prm)
        if dst is not None:                          # a registered destination accumulates EVERY contribution; autograd gets nothing
            into.append(dst)
            what.append(g)
            out[i] = None
            continue
        if not _STATE.get('fold_grads'):
            continue
        first = acc.get(id(prm))
        if first is None:
            # a second tensor on the same storage: autograd keeps the tensor it is handed only while nobody else references
            # it and CLONES it otherwise (one D2D copy per parameter and step: 63 x 5 us in the Moving-MNIST step)
            acc[id(prm)] = g.detach()
        elif first.shape == g.shape and first.dtype == g.dtype:
            into.append(first)
            what.append(g)
            out[i] = None
    if into:
        if _CONV_GRAD_OUT and torch.cuda.is_current_stream_capturing() and os.environ.get('VARSEP_BATCH_SMALL_ADDS', '1') == '1':
            # recorded step under a reducer: EVERY contribution (the first included) is an add into a bucket view -- 44 multi-tensor launches
            # of ~5 us per TaxiBJ step against 11 without a reducer.  Nothing reads the buckets before the backward call returns (no hooks
            # fire in a recording; train.GraphedStep reduces after the replay), so the adds of the whole pass are issued as ONE multi-tensor
            # launch by an engine callback at its end (per segment of a two-segment recording).  Eager steps keep the immediate adds: there
            # the reducer's hooks may put a bucket on the wire as soon as autograd has visited its last parameter.
            _PENDING_ADDS['into'] += into
            _PENDING_ADDS['what'] += what
            if not _PENDING_ADDS['queued']:
                torch.autograd.Variable._execution_engine.queue_callback(flush_pending_adds)
                _PENDING_ADDS['queued'] = True
        else:
            torch._foreach_add_(into, what)
    return out


_PENDING_ADDS = {'into': [], 'what': [], 'queued': False}


def flush_pending_adds():
    into, what = _PENDING_ADDS['into'], _PENDING_ADDS['what']
    _PENDING_ADDS.update(into=[], what=[], queued=False)
    # a parameter that is applied several times per step (E_s on two windows, the decoder on the auto-encoding pair and the rollout, the SST
    # integrator once per frame) appears several times: one multi-tensor launch may hold a destination only ONCE (its entries are processed
    # concurrently), so the k-th contributions of all destinations form launch k
    rounds, seen = [], {}
    for dst, g in zip(into, what):
        k = seen.get(dst.data_ptr(), 0)
        seen[dst.data_ptr()] = k + 1
        if k == len(rounds):
            rounds.append(([], []))
        rounds[k][0].append(dst)
        rounds[k][1].append(g)
    for dsts, gs in rounds:
        torch._foreach_add_(dsts, gs)


_packed_img = {}


def packed_img_weight(p, dtype, flip):
    """MFMA-fragment pre-pack of a Conv2d k3 s1 p1 weight for `ops.conv3_img16` (forward, or flipped / transposed: input gradient)."""
    key = (id(p), dtype, bool(flip))
    ent = _packed_img.get(key)
    if ent is None or ent[0] != _ver(p) or ent[2]() is not p:
        buf = ent[1] if ent is not None and ent[2]() is p else None
        buf = ops.conv3_img16_pack_weight(p.detach().contiguous(), dtype, flip, out=buf)
        _packed_img[key] = (_ver(p), buf, _wref(_packed_img, key, p))
        return buf
    return ent[1]


def prepack_conv3_weights(net, dtype=None):
    """Bring the row-band / few-maps pre-packs (forward and flipped) of every 3x3 stride-1 pad-1 convolution of `net` up to date with ONE
    launch per 96 packs (instead of one launch per pack at its first use: 36-67 launches per TaxiBJ / SST step, every step, because the
    optimizer changes every weight).  Called by the training step right before the forward pass; what is not stale is skipped."""
    import torch.nn as nn
    dtype = dtype or compute_dtype()
    if dtype == torch.float32:
        return 0
    stale = []
    for m in net.modules():
        if not (isinstance(m, nn.Conv2d) and tuple(m.kernel_size) == (3, 3) and tuple(m.stride) == (1, 1) and tuple(m.padding) == (1, 1) and m.groups == 1):
            continue
        p = m.weight
        if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
            continue
        for flip in (False, True):
            K = p.shape[0] if flip else p.shape[1]          # (any K: the pack pads the contraction to whole 64-channel phases)
            key = (id(p), dtype, flip)
            ent = _packed_img.get(key)
            if ent is None or ent[0] != _ver(p) or ent[2]() is not p:
                stale.append((key, p, flip, ent[1] if ent is not None and ent[2]() is p else None))
    if stale:
        bufs = ops.conv3_img16_pack_weights([(p.detach(), flip, buf) for _, p, flip, buf in stale], dtype)
        for (key, p, _, _), buf in zip(stale, bufs):
            _packed_img[key] = (_ver(p), buf, _wref(_packed_img, key, p))
    return len(stale)


def conv_res_block_fusable(x, convs, bns, cdt=None):
    """Whether `ConvResBlockFn` serves a ConvResBlock (resnet.py:53-70): three Conv2d k3 s1 p1 + training-mode BatchNorm on a few 16x16
    maps in a 16-bit compute type, identity skip."""
    cdt = cdt or compute_dtype()
    if os.environ.get('VARSEP_FUSED_RESBLOCK', '1') not in ('1', '2') or cdt == torch.float32 or not x.is_cuda or x.dim() != 4:
        return False
    if x.dtype != torch.float32 or not x.is_contiguous() or len(convs) != 3 or convs[2].out_channels != x.shape[1]:
        return False
    probe = torch.empty((0,), dtype=cdt, device=x.device)
    cin = x.shape[1]
    for conv, bn in zip(convs, bns):
        if (bn is None or not bn.training or not bn.track_running_stats or tuple(conv.kernel_size) != (3, 3) or tuple(conv.stride) != (1, 1)
                or tuple(conv.padding) != (1, 1) or conv.in_channels != cin or conv.groups != 1):
            return False
        shape_in = (x.shape[0], cin, x.shape[2], x.shape[3])
        if not ops.conv3_img16_supported(probe.new_empty(shape_in), conv.out_channels):
            return False
        if not ops.conv3_img16_supported(probe.new_empty((x.shape[0], conv.out_channels, x.shape[2], x.shape[3])), cin):
            return False
        if not ops.bn_small_supported_shape(cdt, x.shape[0], conv.out_channels, x.shape[2] * x.shape[3]):
            return False
        cin = conv.out_channels
    return True


class ConvResBlockFn(torch.autograd.Function):
    """One ConvResBlock (resnet.py:53-70) of the ConvResnet integrator on a few 16x16 maps: x -> (x + r, r), r = BN(conv(act(BN(conv(act(BN(conv
    x))))))).  VARSEP_FUSED_RESBLOCK=2 (round 4, opt-in): every layer ONE launch each way (`ops.conv3_img16_bn_fwd` / `_bwd`: convolution + BatchNorm,
    the splits' partial sums and the maps' statistics exchanged inside the launch) -- forward 3 launches, backward 5 (the top BatchNorm backward, two
    fused layers, the block input's gradient convolution + its slab sum).  Measured and NOT the default: an exchange across the chip inside a
    launch costs what the kernel boundary it replaces costs: SST 19.8 ms (two launches per layer) vs 21.1 (one
    launch, unsplit layers only) / 21.9 (every layer).  VARSEP_FUSED_RESBLOCK=1 (default), the round-3 form: forward = 6 launches (per layer: `ops.conv3_img16` into split slabs, then slab sum + bias + BatchNorm + activation in one;
    the last one also adds the skip and writes the 16-bit copy the next block's convolution reads), backward = 7 (per layer: BatchNorm
    backward that takes its upstream gradient straight from the slabs of the following input-gradient launch and adds d gamma / d beta to
    the pending gradients, then the input-gradient launch; the skip gradient joins in the last slab sum).  Weight gradients are batched
    over the calls of the step as in `ConvBlock`.

    apply(x, x16 or None, w1, b1, g1, be1, w2, ..., be3, cfg) with cfg = ((rmean, rvar, momentum, eps, act) x 3); returns
    (x + r fp32, its 16-bit copy (not differentiable), r fp32, x + r once more: the same storage as a second autograd output)."""

    @staticmethod
    def forward(ctx, x, x16, *rest):
        prm, cfg = rest[:12], rest[12]
        cdt = compute_dtype()
        h = x16 if (x16 is not None and x16.dtype == cdt and x16.shape == x.shape) else to_compute(x, cdt)
        saved = []
        one_launch = os.environ.get('VARSEP_FUSED_RESBLOCK', '1') == '2'       # '1' (default): convolution and BatchNorm as two launches per layer
        for li in range(3):
            w, b, gm, bt = prm[4 * li:4 * li + 4]
            rmean, rvar, momentum, eps, act = cfg[li]
            bias = b.detach() if b is not None else None
            if one_launch and ops.conv3_img16_bn_supported(h.shape[0], h.shape[1], w.shape[0], cdt, act, cdt if li < 2 else torch.float32):
                # convolution + BatchNorm of the layer in ONE launch: the splits' partial sums and the maps' statistics meet inside it
                if li < 2:
                    y, z, mean, invstd = ops.conv3_img16_bn_fwd(h, packed_img_weight(w, cdt, False), bias, gm.detach(), bt.detach(), act, cdt, w.shape[0],
                                                                rmean, rvar, momentum, eps)
                else:
                    y, z, mean, invstd, xnew, xnew16 = ops.conv3_img16_bn_fwd(h, packed_img_weight(w, cdt, False), bias, gm.detach(), bt.detach(), act,
                                                                              torch.float32, w.shape[0], rmean, rvar, momentum, eps, skip=x.detach(),
                                                                              want16=True)
                saved += [h, z, mean, invstd]
                h = y
                continue
            slabs = ops.conv3_img16(h, packed_img_weight(w, cdt, False), w.shape[0])
            if li < 2:
                y, z, mean, invstd = ops.bn_train_fwd_small_slabs(slabs, bias, cdt, gm.detach(), bt.detach(), act, cdt, rmean, rvar, momentum, eps)
            else:
                y, z, mean, invstd, xnew, xnew16 = ops.bn_train_fwd_small_slabs(slabs, bias, cdt, gm.detach(), bt.detach(), act, torch.float32, rmean,
                                                                                rvar, momentum, eps, skip=x.detach(), want16=True)
            saved += [h, z, mean, invstd]
            h = y
        ctx.save_for_backward(*saved)
        ctx.prm, ctx.acts, ctx.cdt = prm, tuple(c[4] for c in cfg), cdt
        ctx.x_needs_grad = x.requires_grad
        ctx.mark_non_differentiable(xnew16)
        ctx.set_materialize_grads(False)          # an unused output arrives as None, not as a tensor of zeros
        # The block output a SECOND time (same storage): a code of the rollout is read by the next block-step AND by the stack of codes the
        # decoder gets (model.py:76-86).  Handed out as two outputs, their gradients arrive separately and join inside the launches below
        # (two upstream operands of the BatchNorm backward, two addends of the skip gradient) -- as ONE output autograd adds them first, one
        # launch per predicted frame (39 of the 41 ATen adds of an SST step).
        return xnew, xnew16, y, xnew.view_as(xnew)

    @staticmethod
    def backward(ctx, g_new, _g16, g_res, g_alias=None):
        prm, cdt, saved = ctx.prm, ctx.cdt, ctx.saved_tensors
        if g_new is None and g_alias is not None:
            g_new, g_alias = g_alias, None
        if g_new is None and g_res is None:
            return (None,) * 15
        if g_alias is not None and g_res is not None:
            g_new, g_alias = g_new + g_alias, None      # (three upstream gradients: the residuals are read too -- not a training path)
        dy_a = g_res if g_res is not None else g_new
        dy_b = g_new if g_res is not None else g_alias
        dy_a = dy_a.contiguous()
        if dy_b is not None:
            dy_b = dy_b.contiguous().float()
        grads = [None] * 12
        dz_up = None                                 # dz of the layer above (li + 1)
        fold = _STATE.get('fold_grads')
        one_launch = os.environ.get('VARSEP_FUSED_RESBLOCK', '1') == '2'
        for li in (2, 1, 0):
            w, b, gm, bt = prm[4 * li:4 * li + 4]
            h, z, mean, invstd = saved[4 * li:4 * li + 4]
            acc = None
            dg_dst, db_dst = conv_grad_output(gm), conv_grad_output(bt)
            if dg_dst is not None and db_dst is not None:
                acc = (dg_dst, db_dst)
            elif fold:
                fg, fb = _fold_slots().get(id(gm)), _fold_slots().get(id(bt))
                if fg is not None and fb is not None and fg.dtype == torch.float32 and fb.dtype == torch.float32 and fg.is_contiguous() and fb.is_contiguous():
                    acc = (fg, fb)
            if dz_up is None:
                dz, dgamma, dbeta = ops.bn_act_bwd_small_ex(z, mean, invstd, gm.detach(), bt.detach(), ctx.acts[li], cdt, dy_a=dy_a, dy_b=dy_b, acc=acc)
            else:
                w_up = prm[4 * (li + 1)]              # the upstream gradient is the input gradient of the layer above
                if one_launch and ops.conv3_img16_bn_supported(dz_up.shape[0], w_up.shape[0], w_up.shape[1], cdt, ctx.acts[li], backward=True):
                    dz, dgamma, dbeta = ops.conv3_img16_bn_bwd(dz_up, packed_img_weight(w_up, cdt, True), w_up.shape[1], z, mean, invstd, gm.detach(),
                                                               bt.detach(), ctx.acts[li], acc=acc)
                else:
                    slabs = ops.conv3_img16(dz_up, packed_img_weight(w_up, cdt, True), w_up.shape[1], role='dgrad')
                    dz, dgamma, dbeta = ops.bn_act_bwd_small_ex(z, mean, invstd, gm.detach(), bt.detach(), ctx.acts[li], cdt, slabs=slabs, acc=acc)
            db = None
            if b is not None and b.requires_grad:
                # exactly zero in front of a training-mode BatchNorm (see ConvBlock.backward)
                db = None if ((fold and id(b) in _fold_slots()) or conv_grad_output(b) is not None) else zero_grad_like(b)
            dw = _conv_weight_grad(w, dz, h, 1, 1, False) if w.requires_grad else None
            grads[4 * li:4 * li + 4] = _fold_param_grads(((w, dw), (b, db), (gm, dgamma), (bt, dbeta)))
            dz_up = dz
        slabs = ops.conv3_img16(dz_up, packed_img_weight(prm[0], cdt, True), prm[0].shape[1], role='dgrad') if ctx.x_needs_grad else None
        dx = None
        if ctx.x_needs_grad:
            dx = ops.slab_sum(slabs, None, torch.float32, addend=g_new.contiguous().float() if g_new is not None else None,
                              addend2=g_alias.contiguous().float() if g_alias is not None else None)
        return (dx, None) + tuple(grads) + (None,)


# ---- inference: BatchNorm folded into the convolution ---------------------------------------------------------------------------------
# In `.eval()` a BatchNorm2d normalises with its RUNNING statistics, i.e. it is a fixed per-channel affine map of the convolution's
# output: act(gamma (conv(x) + b - mean) / sqrt(var + eps) + beta) = act(conv'(x) + b') with W' = W s, b' = (b - mean) s + beta,
# s = gamma / sqrt(var + eps).  The folded block is ONE kernel (the activation sits in the convolution's epilogue or in the pass that
# follows it) instead of convolution + BatchNorm pass; SURVEY section 8f rank 1.  Only without autograd (`torch.no_grad()`: the way the
# reference's evaluation scripts run, test/mnist/test.py:99) -- with gradients enabled the unfolded block is kept so that d gamma / d beta
# exist.  In the 16-bit modes the rounding points move with it: the folded weight is rounded once (instead of the weight), the convolution
# output is not stored before the affine map.
_folded = {}


def folded_conv_bn(conv, bn):
    """(W', b') fp32 tensors of the eval-mode conv -> BatchNorm pair, cached until any of the six tensors involved changes.  The tensors
    keep their identity across refreshes (in-place update), so the operand copies / weight pre-packs keyed on them refresh themselves."""
    srcs = (conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var)
    key = id(conv.weight)
    vers = (tuple(-1 if t is None else t._version for t in srcs) + tuple(0 if t is None else t.data_ptr() for t in srcs)
            + (_EPOCH['replay'], _EPOCH['bn']))
    ent = _folded.get(key)
    if ent is not None and ent[0] == vers and ent[3]() is conv.weight:
        return ent[1], ent[2]
    with torch.no_grad():
        s = bn.weight.detach().float() * torch.rsqrt(bn.running_var.detach().float() + bn.eps)
        shape = (1, -1, 1, 1) if isinstance(conv, torch.nn.ConvTranspose2d) else (-1, 1, 1, 1)
        wf = conv.weight.detach().float() * s.view(shape)
        b0 = conv.bias.detach().float() if conv.bias is not None else torch.zeros_like(s)
        bf = (b0 - bn.running_mean.detach().float()) * s + bn.bias.detach().float()
        if ent is not None and ent[3]() is conv.weight and ent[1].shape == wf.shape:
            ent[1].copy_(wf)
            ent[2].copy_(bf)
            wf, bf = ent[1], ent[2]
        else:
            wf, bf = wf.contiguous(), bf.contiguous()
    if ent is None or ent[3]() is not conv.weight:
        # the entry dies with the weight it belongs to (models built and evaluated repeatedly in one process: tests, sweeps)
        _folded[key] = (vers, wf, bf, weakref.ref(conv.weight, lambda _r, k=key: _drop_folded(k)))
    else:
        _folded[key] = (vers, wf, bf, ent[3])
    return wf, bf


def _drop_folded(key):
    ent = _folded.pop(key, None)
    if ent is not None:
        for cache in (_shadow, _packed, _packed_conv, _packed_tap, _packed_k3, _packed_img, _packed_k4s2):
            for k in [k for k, e in cache.items() if e[2]() is ent[1] or e[2]() is ent[2]]:
                cache.pop(k, None)


def _bn_apply(z, training, rmean, rvar, momentum, eps, groups, gamma, beta, act, out_dt):
    """BatchNorm (+ activation) of a convolution output z on the two-launch path: (y, mean, invstd).  Training with tracked running estimates:
    the fold of the running estimates rides in the apply launch (VS_BN_RUNNING_FUSED=0: the separate bn_running launch)."""
    if training:
        if ops.bn_slab_supported(z, groups):
            # (a call's channel slab fits one workgroup's registers: statistics and apply from ONE read of z)
            return ops.bn_train_fwd_slab(z, gamma.detach(), beta.detach(), act, out_dt, rmean, rvar, momentum, eps, groups=groups)
        if rmean is not None and os.environ.get('VS_BN_RUNNING_FUSED', '1') == '1':
            mean, invstd, ub = ops.bn_stats_ub(z, eps, groups=groups)
            y = ops.bn_act_fwd(z, mean, invstd, gamma.detach(), beta.detach(), act, out_dt, groups=groups, running=(ub, rmean, rvar, momentum))
            return y, mean, invstd
        mean, invstd = ops.bn_stats(z, rmean, rvar, momentum, eps, groups=groups)
    else:
        mean = rmean.detach().unsqueeze(0).expand(groups, -1).contiguous()
        invstd = torch.rsqrt(rvar.detach() + eps).unsqueeze(0).expand(groups, -1).contiguous()
    return ops.bn_act_fwd(z, mean, invstd, gamma.detach(), beta.detach(), act, out_dt, groups=groups), mean, invstd


def band_ok(x, w, transposed, stride, pad, dgrad=False):
    """Conv2d k3 s1 p1 of `x` with weight `w` ([Cout, Cin, 3, 3]) -- or, dgrad=True, its input gradient from x = dz -- on the row-band kernel."""
    return (not transposed and stride == 1 and pad == 1 and w.shape[2] == 3 and w.shape[3] == 3
            and ops.conv3_band_supported(x, w.shape[1] if dgrad else w.shape[0]))


class ConvBlock(torch.autograd.Function):
    """conv / transposed conv -> [BatchNorm2d (per-call batch statistics)] -> [activation]  (conv.py:41-60).

    cfg = (transposed, stride, pad, has_bn, act, training, momentum, eps, out_fp32, groups); `groups` > 1 means the
    batch is that many reference calls stacked along dim 0, each normalised with its own statistics.  Forward keeps the pre-BN conv
    output z (compute dtype) and the batch statistics; backward recomputes the normalised value from z, so no
    post-activation tensor has to be kept for BN blocks (save what OUR backward needs)."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, rmean, rvar, cfg):
        require_cuda(x)
        transposed, stride, pad, has_bn, act, training, momentum, eps, out_fp32, groups = cfg
        cdt = compute_dtype()
        out_dt = torch.float32 if out_fp32 else cdt
        xc = to_compute(x, cdt)
        bias = b.detach() if b is not None else None
        # ConvTranspose2d k4 s2 p1 on 4x4 / 8x8 / 16x16 maps (the DCGAN decoder's middle layers): LDS-staged tap GEMM with the
        # col2im and the BatchNorm sums in its epilogue -- no column matrix, no separate statistics pass
        tap_geom = (transposed and stride == 2 and pad == 1 and w.shape[2] == 4 and w.shape[3] == 4
                    and ops.convt_tap_supported(xc, w.shape[1], groups))
        tap = tap_geom and has_bn
        k3 = (not transposed and stride == 1 and pad == 1 and w.shape[2] == 3 and w.shape[3] == 3
              and ops.conv_k3_tap_supported(xc, w.shape[0], groups))
        # Conv2d k3 s1 p1 on a few 16x16 maps (the SST ConvResnet integrator, resnet.py:53-88): one chip-filling launch that leaves
        # split partial sums; the slab sum, the bias and the whole BatchNorm forward are the next (single) launch
        img = (not transposed and stride == 1 and pad == 1 and w.shape[2] == 3 and w.shape[3] == 3 and groups == 1
               and ops.conv3_img16_supported(xc, w.shape[0]))
        # Conv2d k4 s2 p1 (the DCGAN encoder's stride-2 layers): on the four parity planes of the input the 4x4 stride-2 window is a 3x3
        # stride-1 window, so the row-band kernel carries it -- no column matrix; the planes are what the weight gradient needs, too
        k4 = (not transposed and stride == 2 and pad == 1 and w.shape[2] == 4 and w.shape[3] == 4 and ops.conv_k4s2_supported(xc, w.shape[0]))
        ctx.k4_planes = False
        if k4:
            planes = ops.space_to_depth2(xc)
            wpk = packed_k4s2_weight(w, cdt)
            ctx.k4_planes = True
            if has_bn:
                Bp, C4, Hp, Wp = planes.shape
                if training and ops.conv_band_bn_supported(Bp, C4, Hp, Wp, w.shape[0], groups, cdt):
                    # the BatchNorm sums are taken in the convolution's epilogue: no statistics pass over z
                    if ops.band_bn_mode() == '1':
                        sums = ops.bn_sums_buffer(gamma.data_ptr(), groups, w.shape[0], planes.device)
                        z = ops.conv_k4s2_gather(planes, wpk, bias, w.shape[0], cdt, bn_sums=sums, groups=groups)
                        mean, invstd = ops.bn_stats_from_sums_fold(sums, (Bp // groups) * Hp * Wp, rmean, rvar, momentum, eps)
                    else:                            # per-workgroup partial sums + one fold launch (no atomics)
                        z, parts = ops.conv3_band_parts(planes, wpk, bias, w.shape[0], cdt, k4=True)
                        mean, invstd = ops.bn_stats_from_parts_fold(parts, groups, (Bp // groups) * Hp * Wp, rmean, rvar, momentum, eps)
                    y = ops.bn_act_fwd(z, mean, invstd, gamma.detach(), beta.detach(), act, out_dt, groups=groups)
                else:
                    z = ops.conv_k4s2_gather(planes, wpk, bias, w.shape[0], cdt)
                    y, mean, invstd = _bn_apply(z, training, rmean, rvar, momentum, eps, groups, gamma, beta, act, out_dt)
                ctx.save_for_backward(planes, z, mean, invstd)
            else:
                y = ops.conv_k4s2_gather(planes, wpk, bias, w.shape[0], out_dt)
                if act not in ('none', None):
                    ops.act_fwd(y, act, out=y)
                ctx.save_for_backward(planes, y)
            ctx.x_shape = tuple(xc.shape)
        elif (has_bn and not transposed and stride == 2 and pad == 1 and w.shape[2] == 4 and w.shape[3] == 4
              and ops.conv_k4s2_gather_supported(xc, w.shape[0])):
            # 8 x 8 -> 4 x 4 (the DCGAN encoder's c4, conv.py:122): the forward on the parity planes; the weight gradient keeps the column matrix
            # (the row-band weight gradient does not serve 4 x 4 planes), so backward gets x itself
            planes = ops.space_to_depth2(xc)
            z = ops.conv_k4s2_gather(planes, packed_k4s2_weight(w, cdt), bias, w.shape[0], cdt)
            y, mean, invstd = _bn_apply(z, training, rmean, rvar, momentum, eps, groups, gamma, beta, act, out_dt)
            ctx.save_for_backward(xc, z, mean, invstd)
        elif img:
            slabs = ops.conv3_img16(xc, packed_img_weight(w, cdt, False), w.shape[0])
            if has_bn:
                if training and ops.bn_small_supported_shape(cdt, xc.shape[0], w.shape[0], 256):
                    y, z, mean, invstd = ops.bn_train_fwd_small_slabs(slabs, bias, cdt, gamma.detach(), beta.detach(), act, out_dt, rmean, rvar,
                                                                      momentum, eps)
                else:
                    z = ops.slab_sum(slabs, bias, cdt)
                    y, mean, invstd = _bn_apply(z, training, rmean, rvar, momentum, eps, 1, gamma, beta, act, out_dt)
                ctx.save_for_backward(xc, z, mean, invstd)
            else:
                y = ops.slab_sum(slabs, bias, out_dt)
                if act not in ('none', None):
                    ops.act_fwd(y, act, out=y)
                ctx.save_for_backward(xc, y)
        elif band_ok(xc, w, transposed, stride, pad) and not tap:
            # 3x3 on many maps of width 16 / 32 / 64: row bands through LDS, no column matrix; BatchNorm as for the column-matrix path
            wpk = packed_img_weight(w, cdt, False)
            if has_bn:
                Bx, Cx, Hx, Wx = xc.shape
                small = training and ops.bn_small_supported_shape(cdt, Bx, w.shape[0], Hx * Wx) and (groups == 1 or ops.bn_small_groups_enabled())
                if training and not small and ops.conv_band_bn_supported(Bx, Cx, Hx, Wx, w.shape[0], groups, cdt):
                    # the BatchNorm sums are taken in the convolution's epilogue: no statistics pass over z
                    if ops.band_bn_mode() == '1':
                        sums = ops.bn_sums_buffer(gamma.data_ptr(), groups, w.shape[0], xc.device)
                        z = ops.conv3_band(xc, wpk, bias, w.shape[0], cdt, bn_sums=sums, groups=groups)
                        mean, invstd = ops.bn_stats_from_sums_fold(sums, (Bx // groups) * Hx * Wx, rmean, rvar, momentum, eps)
                    else:                            # per-workgroup partial sums + one fold launch (no atomics)
                        z, parts = ops.conv3_band_parts(xc, wpk, bias, w.shape[0], cdt)
                        mean, invstd = ops.bn_stats_from_parts_fold(parts, groups, (Bx // groups) * Hx * Wx, rmean, rvar, momentum, eps)
                    y = ops.bn_act_fwd(z, mean, invstd, gamma.detach(), beta.detach(), act, out_dt, groups=groups)
                    ctx.save_for_backward(xc, z, mean, invstd)
                    ctx.cfg, ctx.cdt = cfg, cdt
                    ctx.w, ctx.b, ctx.gamma, ctx.beta = w, b, gamma, beta
                    ctx.x_dtype, ctx.x_needs_grad = x.dtype, x.requires_grad
                    ctx.x_shape = tuple(xc.shape)
                    return y
                z = ops.conv3_band(xc, wpk, bias, w.shape[0], cdt)
                if training and ops.bn_small_supported(z, groups):
                    y, mean, invstd = ops.bn_train_fwd_small(z, gamma.detach(), beta.detach(), act, out_dt, rmean, rvar, momentum, eps, groups=groups)
                else:
                    y, mean, invstd = _bn_apply(z, training, rmean, rvar, momentum, eps, groups, gamma, beta, act, out_dt)
                ctx.save_for_backward(xc, z, mean, invstd)
            else:
                y = ops.conv3_band(xc, wpk, bias, w.shape[0], out_dt)
                if act not in ('none', None):
                    ops.act_fwd(y, act, out=y)
                ctx.save_for_backward(xc, y)
        elif tap or (k3 and has_bn):
            if tap:
                z, sums = ops.convt_tap_fwd(xc, packed_tap_weight(w, cdt), bias, w.shape[1], groups=groups, want_sums=training)
            else:
                z, sums = ops.conv_k3_tap_fwd(xc, packed_k3_weight(w, cdt, False), bias, w.shape[0], cdt, groups=groups, want_sums=training)
            if training:
                n_per = (z.shape[0] // groups) * z.shape[2] * z.shape[3]
                mean, invstd = ops.bn_stats_from_sums_fold(sums, n_per, rmean, rvar, momentum, eps, reset=False)     # (one launch: statistics + running fold)
            else:
                mean = rmean.detach().unsqueeze(0).expand(groups, -1).contiguous()
                invstd = torch.rsqrt(rvar.detach() + eps).unsqueeze(0).expand(groups, -1).contiguous()
            y = ops.bn_act_fwd(z, mean, invstd, gamma.detach(), beta.detach(), act, out_dt, groups=groups)
            ctx.save_for_backward(xc, z, mean, invstd)
        elif has_bn:
            wc = shadow(w, cdt)
            wp = packed_conv_weight(w, cdt, stride, pad) if transposed else None
            z = ops.conv_fwd(xc, wc, bias, stride, pad, transposed, cdt, w_packed=wp)
            if training and ops.bn_small_supported(z, groups):
                # small maps (the SST integrator: 8 x 16 x 16 per channel): statistics, running update, affine + activation in one launch
                y, mean, invstd = ops.bn_train_fwd_small(z, gamma.detach(), beta.detach(), act, out_dt, rmean, rvar, momentum, eps, groups=groups)
            else:
                y, mean, invstd = _bn_apply(z, training, rmean, rvar, momentum, eps, groups, gamma, beta, act, out_dt)
            ctx.save_for_backward(xc, z, mean, invstd)
        else:
            if k3:
                y, _ = ops.conv_k3_tap_fwd(xc, packed_k3_weight(w, cdt, False), bias, w.shape[0], out_dt, groups=1)
            elif tap_geom and out_dt == cdt:
                # (inference with the BatchNorm folded into the weight: the tap kernel without its statistics epilogue)
                y, _ = ops.convt_tap_fwd(xc, packed_tap_weight(w, cdt), bias, w.shape[1], groups=groups, want_sums=False)
            else:
                wc = shadow(w, cdt)
                wp = packed_conv_weight(w, cdt, stride, pad) if transposed else None
                y = ops.conv_fwd(xc, wc, bias, stride, pad, transposed, out_dt, w_packed=wp)
            if act not in ('none', None):
                ops.act_fwd(y, act, out=y)
            ctx.save_for_backward(xc, y)
        ctx.cfg, ctx.cdt = cfg, cdt
        ctx.w, ctx.b, ctx.gamma, ctx.beta = w, b, gamma, beta
        ctx.x_dtype, ctx.x_needs_grad = x.dtype, x.requires_grad
        if not ctx.k4_planes:
            ctx.x_shape = tuple(xc.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        transposed, stride, pad, has_bn, act, training, momentum, eps, out_fp32, groups = ctx.cfg
        cdt, w, b = ctx.cdt, ctx.w, ctx.b
        dgamma = dbeta = None
        if has_bn:
            xc, z, mean, invstd = ctx.saved_tensors
            dz, dgamma, dbeta = ops.bn_act_bwd(dy, z, mean, invstd, ctx.gamma.detach(), ctx.beta.detach(), act, training, cdt,
                                               groups=groups)
        else:
            xc, y = ctx.saved_tensors
            dz = ops.act_bwd(dy, y, act, out_dtype=cdt) if act not in ('none', None) else to_compute(dy, cdt)
        db = None
        if b is not None and b.requires_grad:
            # a conv bias in front of a training-mode BatchNorm has an exactly-zero gradient (the batch mean removes it);
            # the reference returns fp32 summation noise there, we return the exact value without a reduction pass
            if has_bn and training:
                # exactly zero: nothing to add once the parameter has a pending gradient in this pass
                db = None if ((_STATE.get('fold_grads') and id(b) in _fold_slots()) or conv_grad_output(b) is not None) else zero_grad_like(b)
            else:
                db = ops.chan_sum(dz)
        x_shape = ctx.x_shape
        dz_planes = None
        if ctx.k4_planes:
            # forward kept the parity planes of the input: they are the large operand of the weight gradient
            dw = _conv_weight_grad(w, dz, None, stride, pad, transposed, k4s2=(dz, xc)) if w.requires_grad else None
        elif (transposed and stride == 2 and pad == 1 and w.shape[2] == 4 and w.shape[3] == 4 and dz.dtype == xc.dtype
              and ops.conv_k4s2_supported(dz, w.shape[0])):
            # ConvTranspose2d k4 s2 p1 (the DCGAN decoder): the parity planes of the output gradient serve the weight gradient (small map
            # = the layer input) AND the input gradient (a k4 s2 p1 gather of dz with the weight read as [out = Cin][in = Cout])
            dz_planes = ops.space_to_depth2(dz)
            dw = _conv_weight_grad(w, dz, xc, stride, pad, transposed, k4s2=(xc, dz_planes)) if w.requires_grad else None
        else:
            dw = _conv_weight_grad(w, dz, xc, stride, pad, transposed) if w.requires_grad else None
        dx = None
        if ctx.x_needs_grad and dz_planes is not None:
            dx = ops.conv_k4s2_gather(dz_planes, packed_k4s2_weight(w, cdt), None, w.shape[0], ctx.x_dtype, role='dgrad')
        elif ctx.x_needs_grad:
            # the input gradient of Conv2d k4 s2 p1 IS a ConvTranspose2d k4 s2 p1 of dz with the same weight tensor ([Cout, Cin, 4, 4]
            # read as [in, out, 4, 4]): on 4x4 / 8x8 / 16x16 gradient maps it takes the LDS-staged tap kernel (no column matrix)
            if (not transposed and stride == 2 and pad == 1 and w.shape[2] == 4 and w.shape[3] == 4 and ctx.x_dtype == dz.dtype
                    and x_shape[2] == 2 * dz.shape[2] and x_shape[3] == 2 * dz.shape[3] and ops.convt_tap_supported(dz, w.shape[1], 1)):
                dx, _ = ops.convt_tap_fwd(dz, packed_tap_weight(w, cdt), None, w.shape[1], groups=1, want_sums=False, role='dgrad')
            elif (not transposed and stride == 1 and pad == 1 and w.shape[2] == 3 and w.shape[3] == 3 and groups == 1
                  and ops.conv3_img16_supported(dz, w.shape[1])):
                # few 16x16 maps: the same one-launch kernel on dz with the weight packed transposed and flipped
                dx = ops.slab_sum(ops.conv3_img16(dz, packed_img_weight(w, cdt, True), w.shape[1], role='dgrad'), None, ctx.x_dtype)
            elif band_ok(dz, w, transposed, stride, pad, dgrad=True):
                dx = ops.conv3_band(dz, packed_img_weight(w, cdt, True), None, w.shape[1], ctx.x_dtype, role='dgrad')
            elif (not transposed and stride == 1 and pad == 1 and w.shape[2] == 3 and w.shape[3] == 3
                  and ops.conv_k3_tap_supported(dz, w.shape[1], 1)):
                # Conv2d k3 s1 p1: the input gradient is the same convolution of dz with the weight transposed and flipped
                dx, _ = ops.conv_k3_tap_fwd(dz, packed_k3_weight(w, cdt, True), None, w.shape[1], ctx.x_dtype, groups=1, role='dgrad')
            else:
                wp = None if transposed else packed_conv_weight(w, cdt, stride, pad)
                dx = ops.conv_dgrad(dz, shadow(w, cdt), x_shape, stride, pad, transposed, ctx.x_dtype, w_packed=wp,
                                    cols_from_wgrad=transposed and bool(w.requires_grad) and not _STATE.get('wgrad_on_lane'))
        dw, db, dgamma, dbeta = _fold_param_grads(((w, dw), (b, db), (ctx.gamma, dgamma), (ctx.beta, dbeta)))
        return dx, dw, db, dgamma, dbeta, None, None, None


class MaxPool2(torch.autograd.Function):
    """nn.MaxPool2d(2, 2) (conv.py:151-169, 330-335)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return ops.maxpool2_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.maxpool2_bwd(x, dy.to(x.dtype) if dy.dtype != x.dtype else dy)


class MaxPool3s2(torch.autograd.Function):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (conv.py:517, ResNet18 stem)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return ops.maxpool3s2_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.maxpool3s2_bwd(x, dy.to(x.dtype) if dy.dtype != x.dtype else dy)


class Upsample2(torch.autograd.Function):
    """nn.Upsample(scale_factor=2, mode='nearest') (conv.py:296-314, 371-377, 406-413)."""

    @staticmethod
    def forward(ctx, x):
        ctx.dt = x.dtype
        return ops.upsample2_fwd(x.contiguous())

    @staticmethod
    def backward(ctx, dy):
        return ops.upsample2_bwd(dy, ctx.dt)


class Activation(torch.autograd.Function):
    """Stand-alone activation (only where the reference applies one outside a conv block, e.g. conv.py:396 out_f)."""

    @staticmethod
    def forward(ctx, x, act):
        y = ops.act_fwd(x, act)
        ctx.save_for_backward(y)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.act_bwd(dy, y, ctx.act, out_dtype=y.dtype), None


class FramesMSE(torch.autograd.Function):
    """F.mse_loss(frames, full[:, idx]) of a [B, G, D] stack of decoded frames against observed frames picked by a device-side
    index vector: one pass forward, one backward (the same two kernels as FrameLosses), instead of slice / sub / pow / mean and
    their five backward launches.  The conv families' two frame losses (train.py:85-86, 139)."""

    @staticmethod
    def forward(ctx, frames, full, idx):
        B, G, D = frames.shape
        sums = ops.frames_sse_fwd(frames, full, idx)
        ctx.save_for_backward(frames, full, idx)
        ctx.scale = 1.0 / (B * G * D)
        return (sums[0] + sums[1]) * ctx.scale

    @staticmethod
    def backward(ctx, g):
        frames, full, idx = ctx.saved_tensors
        coef = (g.float() * (2.0 * ctx.scale)).reshape(1).expand(2).contiguous()
        return ops.frames_sse_bwd(frames, full, idx, coef), None, None


def frames_mse(frames, full, idx):
    """mean((frames - full[:, idx])^2); frames [B, G, ...] or [B, ...] (G = 1), full [B, T, ...], idx int32 [G] on the device."""
    B, T = full.shape[0], full.shape[1]
    flat = full.reshape(B, T, -1)
    if not flat.is_contiguous():
        flat = flat.contiguous()
    G = idx.numel()
    return FramesMSE.apply(frames.reshape(B, G, -1).contiguous().float(), flat.float(), idx)


class CatBcast(torch.autograd.Function):
    """cat([a.repeat(n, 1, 1, 1), x], dim=1) for the decoder inputs of a batched rollout (conv.py:228, 388-394: the skip tensors / the spatial
    code of the B sequences are shared by the n frame calls): one pass forward (no repeated copy of `a`), one backward (d a = sum over the
    frames, d x = the other channels)."""

    @staticmethod
    def forward(ctx, a, x, n, out_dtype):
        ctx.meta = (a.shape[0], int(n), a.shape[1], a.dtype, x.dtype)
        return ops.cat_bcast_fwd(a.detach().contiguous(), x.detach().contiguous(), int(n), out_dtype)

    @staticmethod
    def backward(ctx, dout):
        B, n, Ca, a_dtype, x_dtype = ctx.meta
        da, dx = ops.cat_bcast_bwd(dout, B, n, Ca, a_dtype, x_dtype, need_a=ctx.needs_input_grad[0], need_x=ctx.needs_input_grad[1])
        return da, dx, None, None


def cat_bcast(a, x, n, out_dtype=None):
    """cat([a repeated n times along the batch axis, x], dim=1) in `out_dtype` (default: x's); falls back to torch ops where the kernel does not
    take the tensors (CPU, odd plane sizes) or VARSEP_CAT_BCAST=0."""
    out_dtype = out_dtype or x.dtype
    if (n > 1 and os.environ.get('VARSEP_CAT_BCAST', '1') == '1' and a.is_cuda and a.dim() == 4 and x.dim() == 4
            and ops.cat_bcast_supported(a.detach().contiguous(), x.detach().contiguous(), n)):
        return CatBcast.apply(a, x, n, out_dtype)
    return torch.cat([a.repeat(n, 1, 1, 1).to(out_dtype), x.to(out_dtype)], dim=1)


class ConvLosses(torch.autograd.Function):
    """The four losses of a conv-family step and their weighted sum (train.py:85-86, 38-42, 139-149) in 4 launches forward (two frame-sum
    kernels, the code-loss partials, a one-block finish) and 3 backward (one launch for every code gradient and the frame coefficients, the two
    frame kernels) -- instead of frames_mse x 2 + zero_order_loss (two concatenations of every skip tensor with skip connections) + ~30
    scalar / reduction launches each way.  apply(recon [B, 1, D], fore [B, G, D], full [B, T, D], ae_idx, f_idx, t0, meta, a_1, b_1, ..., a_k,
    b_k) -> (total, ae, zero, pred, t_reg); meta = (lambdas (ae, s, t, pred), inv_t).  Only `total` is differentiable."""

    @staticmethod
    def forward(ctx, recon, fore, full, ae_idx, f_idx, t0, meta, *flat):
        lambdas, inv_t = meta
        pairs = [(flat[2 * j].detach(), flat[2 * j + 1].detach()) for j in range(len(flat) // 2)]
        sse_ae = ops.frames_sse_fwd(recon, full, ae_idx)
        sse_pred = ops.frames_sse_fwd(fore, full, f_idx)
        scale_ae, scale_pred = 1.0 / recon.numel(), 1.0 / fore.numel()
        out = ops.code_losses_fwd(pairs, t0.detach(), sse_ae, sse_pred, scale_ae, scale_pred, lambdas, inv_t)
        ctx.save_for_backward(recon, fore, full, ae_idx, f_idx, t0, *flat)
        ctx.meta = (tuple(float(v) for v in lambdas), float(inv_t), scale_ae, scale_pred)
        ctx.set_materialize_grads(False)
        total, ae, zero, pred, treg = out[0], out[1], out[2], out[3], out[4]
        ctx.mark_non_differentiable(ae, zero, pred, treg)
        return total, ae, zero, pred, treg

    @staticmethod
    def backward(ctx, g, *_unused):
        recon, fore, full, ae_idx, f_idx, t0 = ctx.saved_tensors[:6]
        flat = ctx.saved_tensors[6:]
        lambdas, inv_t, scale_ae, scale_pred = ctx.meta
        if g is None:
            return (None,) * (7 + len(flat))
        pairs = [(flat[2 * j].detach(), flat[2 * j + 1].detach()) for j in range(len(flat) // 2)]
        need = [(ctx.needs_input_grad[7 + 2 * j], ctx.needs_input_grad[8 + 2 * j]) for j in range(len(pairs))]
        da, db, dt0, coefs = ops.code_losses_bwd(pairs, need, t0.detach(), g.detach().float().reshape(1), scale_ae, scale_pred, lambdas, inv_t)
        d_recon = ops.frames_sse_bwd(recon, full, ae_idx, coefs[0:2]) if ctx.needs_input_grad[0] else None
        d_fore = ops.frames_sse_bwd(fore, full, f_idx, coefs[2:4]) if ctx.needs_input_grad[1] else None
        grads = []
        for j in range(len(pairs)):
            grads += [da[j], db[j]]
        return (d_recon, d_fore, None, None, None, dt0 if ctx.needs_input_grad[5] else None, None) + tuple(grads)


def conv_losses(reconstruction, forecasts, full_data, ae_idx, f_idx, s_old, s_new, skipco, t0, lambdas, average_tloss):
    """(total, {'ae', 'zero', 'pred', 't_reg'}) through ConvLosses, or None when the fused kernels do not take these tensors (the caller then
    assembles the losses from torch ops as before).  lambdas = (ae, s, t, pred)."""
    B, T = full_data.shape[0], full_data.shape[1]
    flat = full_data.reshape(B, T, -1)
    if not flat.is_contiguous():
        flat = flat.contiguous()
    if flat.dtype != torch.float32 or t0.dtype != torch.float32:
        return None
    if skipco:
        olds, news = [s_old[0]] + list(s_old[1]), [s_new[0]] + list(s_new[1])
    else:
        olds, news = [s_old], [s_new]
    pairs = [(a.contiguous(), b.contiguous()) for a, b in zip(olds, news)]
    t0c = t0.contiguous()
    if not ops.code_losses_supported([(a.detach(), b.detach()) for a, b in pairs], t0c.detach()):
        return None
    G = f_idx.numel()
    recon = reconstruction.reshape(B, 1, -1).contiguous().float()
    fore = forecasts.reshape(B, G, -1).contiguous().float()
    # train.py:141-146: 0.5 * mean over all elements, or 0.5 * sum over dim 1, mean over the rest
    inv_t = 1.0 / t0c.numel() if average_tloss else float(t0c.shape[1]) / t0c.numel()
    flat_pairs = []
    for a, b in pairs:
        flat_pairs += [a, b]
    total, ae, zero, pred, treg = ConvLosses.apply(recon, fore, flat, ae_idx, f_idx, t0c, (tuple(lambdas), inv_t), *flat_pairs)
    return total, {'ae': ae, 'zero': zero, 'pred': pred, 't_reg': treg}


# ------------------------------------------------------------------------------------------------ all training losses
class TrainLosses(torch.autograd.Function):
    """total = l_ae*ae + l_s*zero + l_pred*pred + l_t*t_reg and the four terms (train.py:117-149) from the decoded frame stack
    [B, 1+n, D], the spatial codes of the first/last window and the initial temporal code: one kernel forward, one backward
    (ops.train_losses_*), instead of ~30 elementwise / reduction launches each way.  Returns (total, ae, zero, pred, t_reg); only
    `total` is differentiable, the terms are for logging."""

    @staticmethod
    def forward(ctx, frames, full, idx, s_old, s_new, t0, lambdas, average_tloss, handoff=None):
        ctx.handoff = handoff
        # idx: int32 [1+n] target frames on the device, or (t_random int32 [1] on the device, ae_shift, first_forecast): frame 0
        # <-> full[:, t_random - ae_shift], frame g <-> full[:, first_forecast + g - 1], resolved inside the kernels
        ctx.window = None
        if isinstance(idx, tuple):
            ctx.window, idx = (int(idx[1]), int(idx[2])), idx[0]
        idx_arg = idx if ctx.window is None else (idx,) + ctx.window
        ctx.early = None
        up = _STATE.get('promised_loss_grad')
        if handoff is not None and handoff.fused is not None:
            # the chain's last GEMM evaluated the frame losses in its epilogue (no frames exist): take its results
            (out, dz, ds_old, ds_new, dt0), handoff.fused = handoff.fused, None
            up = handoff.fuse['up']
            ctx.early = (up.data_ptr(), up._version, dz.view(frames.shape), ds_old, ds_new, dt0)
        elif (up is not None and handoff is not None and handoff.act not in ('none', None) and frames.shape[-1] % 4 == 0 and ctx.needs_input_grad[0]
                and up.device == frames.device):
            # the caller has promised the tensor it will pass to backward (a recorded step: its resident 1.0 / loss scale): the
            # gradients are written by the same pass that sums the losses; backward hands them out if the promise was kept
            out, dz, ds_old, ds_new, dt0 = ops.train_losses_fwd_grad(frames, full, idx_arg, s_old, s_new, t0, lambdas, average_tloss, up,
                                                                       handoff.act, handoff.cdt)
            ctx.early = (up.data_ptr(), up._version, dz, ds_old, ds_new, dt0)
        else:
            out = ops.train_losses_fwd(frames, full, idx_arg, s_old, s_new, t0, lambdas, average_tloss)
        ctx.set_materialize_grads(False)             # no zero tensors for the (unused) gradients of the four logging terms
        ctx.save_for_backward(frames, full, idx, s_old, s_new, t0)
        ctx.lambdas, ctx.average = tuple(float(v) for v in lambdas), bool(average_tloss)
        total, ae, zero, pred, treg = out[4], out[5], out[6], out[7], out[8]
        ctx.mark_non_differentiable(ae, zero, pred, treg)
        return total, ae, zero, pred, treg

    @staticmethod
    def backward(ctx, g_total, *_unused):
        frames, full, idx, s_old, s_new, t0 = ctx.saved_tensors
        if g_total is None:
            return (None,) * 9
        idx_arg = idx if ctx.window is None else (idx,) + ctx.window
        h = ctx.handoff
        if ctx.early is not None and g_total.data_ptr() == ctx.early[0] and g_total.numel() == 1:
            _, _, h.dz, ds_old, ds_new, dt0 = ctx.early           # computed by the forward pass for exactly this upstream gradient
            ctx.early = None
            return GradHandoff.placeholder(frames), None, None, ds_old, ds_new, dt0, None, None, None
        if h is not None and h.act not in ('none', None) and frames.shape[-1] % 4 == 0:
            # the frames are the outputs of the producing chain's last activation: write d/d(pre-activation) for it directly
            h.dz, ds_old, ds_new, dt0 = ops.train_losses_bwd(frames, full, idx_arg, s_old, s_new, t0, ctx.lambdas, ctx.average,
                                                             g_total.float().contiguous(), frames_act=h.act, dz_dtype=h.cdt)
            dframes = GradHandoff.placeholder(frames)
        else:
            dframes, ds_old, ds_new, dt0 = ops.train_losses_bwd(frames, full, idx_arg, s_old, s_new, t0, ctx.lambdas, ctx.average,
                                                                g_total.float().contiguous())
        return dframes, None, None, ds_old, ds_new, dt0, None, None, None


# ------------------------------------------------------------------------------------------------ decoder input of a rollout
class MixCodes(torch.autograd.Function):
    """z[b, g] = mix(s[b], [t_rand ; t_codes][b, g]) for the auto-encoding pair (g = 0) and every rollout step: the decoder
    input of model.py:74-83 with the mixing of mlp_encdec.py:43-48, one launch forward (instead of expand, cat, mul, cast)
    and one backward (instead of the mul / sum-over-frames / slice / cat gradients).  Returns (z fp32, z in the compute
    dtype or None); the second output is not differentiable and feeds mlp_chain(x_lowp=...)."""

    @staticmethod
    def forward(ctx, s, t_rand, t_codes, mixing):
        cdt = compute_dtype()
        z, z_lowp = ops.mix_codes_fwd(s, t_rand, t_codes, mixing, lowp=None if cdt == torch.float32 else cdt)
        ctx.save_for_backward(s, t_rand, t_codes)
        ctx.mixing = mixing
        # (no zero tensor for the gradient of the non-differentiable second output: autograd would launch a fill per step for it)
        ctx.set_materialize_grads(False)
        if z_lowp is None:
            return z, None
        ctx.mark_non_differentiable(z_lowp)
        return z, z_lowp

    @staticmethod
    def backward(ctx, dz, _unused=None):
        s, t_rand, t_codes = ctx.saved_tensors
        if dz is None:
            return None, None, None, None
        ds, dt_rand, dt_codes = ops.mix_codes_bwd(dz.float().contiguous(), s, t_rand, t_codes, ctx.mixing)
        return ds, dt_rand, dt_codes, None


# ------------------------------------------------------------------------------------------------ fused frame losses
class FrameLosses(torch.autograd.Function):
    """(ae_mse, forecast_mse) of a [B, 1+n, D] stack of decoded frames against the frames `full[:, idx[g]]` in one pass.

    Reference: train.py:85-86 (F.mse_loss of the reconstruction) and train.py:139 (F.mse_loss of the forecasts); both are
    plain means of squared errors, so they are two partial sums of the same kernel.  Backward writes
    d frames = 2/N_k * upstream_k * (frames - target) directly (no slicing / concatenation gradients to materialise)."""

    @staticmethod
    def forward(ctx, frames, full, idx):
        B, G, D = frames.shape
        sums = ops.frames_sse_fwd(frames, full, idx)
        ctx.save_for_backward(frames, full, idx)
        ctx.scale = (1.0 / (B * D), 1.0 / (B * max(G - 1, 1) * D))      # python floats: nothing is copied from the host
        return sums[0] * ctx.scale[0], sums[1] * ctx.scale[1]

    @staticmethod
    def backward(ctx, g_ae, g_pred):
        frames, full, idx = ctx.saved_tensors
        coef = torch.stack([g_ae.float() * (2.0 * ctx.scale[0]), g_pred.float() * (2.0 * ctx.scale[1])])
        return ops.frames_sse_bwd(frames, full, idx, coef), None, None
