"""Thin tensor-level wrappers over the C ABI (include/varsep_hip.h).  No autograd here; see functional.py."""
import torch

from . import _lib
from ._lib import F32, BF16, F16, ACT, LAYOUT_R, LAYOUT_S, check, code_of, dtype_code, stream_ptr, require_cuda

_DT = {F32: 'f32', BF16: 'bf16', F16: 'fp16'}

_ws_cache = {}
_retired = []

# ---- optional per-launch timing with HIP events on the launch stream (bench.py: roofline of the dominant kernel) ----
_PROF = {'on': False, 'events': []}
_LNAME = {0: 'R', 1: 'S'}


def profile_reset(enable=True, pool=0):
    """Start (or stop) per-launch event timing.  `pool` events are created and recorded once up front so that the timed
    region only re-records existing HIP events (creating thousands of events inside the region slows the host)."""
    _PROF['on'] = enable
    _PROF['events'] = []
    _PROF['pool'] = []
    if enable and pool:
        for _ in range(pool):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            _PROF['pool'].append(e)
        torch.cuda.synchronize()


def _new_event():
    pool = _PROF.get('pool')
    e = pool.pop() if pool else torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _pb():
    if not _PROF['on']:
        return None
    return _new_event()


def _pe(e0, name, flops=0.0, nbytes=0.0):
    if e0 is None:
        return
    _PROF['events'].append((name, e0, _new_event(), flops, nbytes))


def profile_collect():
    """{kernel family: {'ms': summed event time, 'n': launches, 'flops': algorithmic FLOPs, 'bytes': algorithmic bytes}}"""
    torch.cuda.synchronize()
    out = {}
    for name, e0, e1, fl, by in _PROF['events']:
        r = out.setdefault(name, {'ms': 0.0, 'n': 0, 'flops': 0.0, 'bytes': 0.0})
        r['ms'] += e0.elapsed_time(e1)
        r['n'] += 1
        r['flops'] += fl
        r['bytes'] += by
    _PROF['on'] = False
    _PROF['events'] = []
    return out


def _workspace(nbytes, device):
    """One grow-only split-K workspace per (device, stream): launches on different streams may overlap, and a workspace is only
    ordered by the stream it is used on (allocated by torch, so legal inside graph capture after warm-up)."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _retired.append(buf)      # a recorded hipGraph may hold its address: outgrown workspaces are never freed
        buf = torch.empty(max(nbytes, 2 * buf.numel() if buf is not None else 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def gemm(a, layout_a, b, layout_b, M, N, K, out=None, out_dtype=torch.float32, alpha=1.0, bias=None, act='none',
         mask=None, mask_act='none', accumulate=False, lda=None, ldb=None):
    """out[M,N] = epi(sum_k A(m,k) B(n,k)); A/B are 2-D (possibly row-strided) tensors of identical dtype."""
    require_cuda(a, b, out, bias, mask)
    lib = _lib.load_library()
    if a.dtype != b.dtype:
        raise _lib.VarsepHipError('gemm operands must share a dtype (%s vs %s)' % (a.dtype, b.dtype))
    assert a.stride(-1) == 1 and b.stride(-1) == 1
    compute = dtype_code(a)
    lda = a.stride(0) if lda is None else lda
    ldb = b.stride(0) if ldb is None else ldb
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    assert out.stride(-1) == 1
    ws_bytes = lib.vs_gemm_workspace_bytes(M, N, K)
    ws = _workspace(ws_bytes, a.device) if ws_bytes else None
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.is_contiguous()
    e0 = _pb()
    check(lib.vs_gemm(compute, M, N, K, a.data_ptr(), lda, layout_a, b.data_ptr(), ldb, layout_b, out.data_ptr(),
                      out.stride(0), dtype_code(out), float(alpha), _ptr(bias), ACT[act], _ptr(mask),
                      mask.stride(0) if mask is not None else 0, dtype_code(mask) if mask is not None else 0,
                      ACT[mask_act], int(bool(accumulate)), _ptr(ws), ws.numel() if ws is not None else 0, stream_ptr()),
          'vs_gemm')
    _pe(e0, 'vs_gemm<%s,%s%s>' % (_DT[compute], _LNAME[layout_a], _LNAME[layout_b]),
        flops=2.0 * M * N * K, nbytes=float((M * K + N * K) * a.element_size() + M * N * out.element_size()))
    return out


def gemm_adam(a, layout_a, b, layout_b, M, N, K, param, exp_avg, exp_avg_sq, shadow, step_dev, skipped, lr, betas, eps, alpha=1.0):
    """param [M, N] (fp32, contiguous) takes one Adam step with G = alpha * A B^T as its gradient; G is never stored
    (vs_gemm_adam).  Raises VarsepHipError(code VS_ERR_UNSUPPORTED) when the operands do not fit the LDS-DMA loader."""
    require_cuda(a, b, param, exp_avg, exp_avg_sq, step_dev)
    lib = _lib.load_library()
    assert a.dtype == b.dtype and a.dtype in (torch.bfloat16, torch.float16) and a.stride(-1) == 1 and b.stride(-1) == 1
    assert param.dtype == torch.float32 and param.is_contiguous() and tuple(param.shape) == (M, N)
    assert exp_avg.is_contiguous() and exp_avg_sq.is_contiguous() and exp_avg.shape == param.shape == exp_avg_sq.shape
    assert shadow is None or (shadow.is_contiguous() and shadow.shape == param.shape and shadow.dtype in (torch.bfloat16, torch.float16))
    e0 = _pb()
    check(lib.vs_gemm_adam(dtype_code(a), M, N, K, a.data_ptr(), a.stride(0), layout_a, b.data_ptr(), b.stride(0), layout_b, float(alpha),
                           param.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), _ptr(shadow),
                           code_of(shadow.dtype) if shadow is not None else BF16, step_dev.data_ptr(), int(skipped), float(lr),
                           float(betas[0]), float(betas[1]), float(eps), stream_ptr()), 'vs_gemm_adam')
    # algorithmic bytes: the operands, p / m / v read and written, the 16-bit copy
    _pe(e0, 'vs_gemm_adam<%s,%s%s>' % (_DT[dtype_code(a)], _LNAME[layout_a], _LNAME[layout_b]), flops=2.0 * M * N * K,
        nbytes=float((M * K + N * K) * a.element_size() + M * N * (24 + (2 if shadow is not None else 0))))


def gemm_batched(a, layout_a, b, layout_b, M, N, K, out_dtype=torch.float32, out=None):
    """a [batch, ., .], b [batch, ., .] contiguous 3-D tensors of one dtype: out[i] = A_i B_i^T for every i in ONE launch."""
    require_cuda(a, b)
    lib = _lib.load_library()
    if a.dtype != b.dtype:
        raise _lib.VarsepHipError('gemm operands must share a dtype (%s vs %s)' % (a.dtype, b.dtype))
    assert a.dim() == 3 and b.dim() == 3 and a.is_contiguous() and b.is_contiguous() and a.shape[0] == b.shape[0]
    batch = a.shape[0]
    compute = dtype_code(a)
    if out is None:
        out = torch.empty((batch, M, N), dtype=out_dtype, device=a.device)
    assert out.is_contiguous() and tuple(out.shape) == (batch, M, N)
    ws_bytes = lib.vs_gemm_batched_workspace_bytes(batch, M, N, K)
    ws = _workspace(ws_bytes, a.device) if ws_bytes else None
    e0 = _pb()
    check(lib.vs_gemm_batched(compute, batch, M, N, K, a.data_ptr(), a.stride(1), a.stride(0), layout_a, b.data_ptr(), b.stride(1),
                              b.stride(0), layout_b, out.data_ptr(), N, M * N, dtype_code(out), 1.0, 0, _ptr(ws),
                              ws.numel() if ws is not None else 0, stream_ptr()), 'vs_gemm_batched')
    _pe(e0, 'vs_gemm<%s,%s%s>' % (_DT[compute], _LNAME[layout_a], _LNAME[layout_b]),
        flops=2.0 * batch * M * N * K, nbytes=float(batch * ((M * K + N * K) * a.element_size() + M * N * out.element_size())))
    return out


def cast(src, dtype, out=None):
    require_cuda(src)
    src = src.contiguous()
    if out is None:
        out = torch.empty(src.shape, dtype=dtype, device=src.device)
    check(_lib.load_library().vs_cast(src.data_ptr(), dtype_code(src), out.data_ptr(), dtype_code(out), src.numel(),
                                      stream_ptr()), 'vs_cast')
    return out


def copy2d(src, rows, cols, lds, out, ldd, col_offset_dev=None, col_offset_scale=0, src_elem_offset=0):
    require_cuda(src, out)
    esz = src.element_size()
    check(_lib.load_library().vs_copy2d(src.data_ptr() + src_elem_offset * esz, dtype_code(src), lds, out.data_ptr(),
                                        dtype_code(out), ldd, rows, cols, _ptr(col_offset_dev), col_offset_scale,
                                        stream_ptr()), 'vs_copy2d')
    return out


def copy2d_pair(src, rows, cols, lds, out, ldd, col_offset_dev, col_offset_scale, elem_offset_a, elem_offset_b):
    """out[0:rows] = window of `src` rows at elem_offset_a + *col_offset_dev * col_offset_scale, out[rows:2 rows] = the window at
    elem_offset_b, one launch (with the element-type conversion)."""
    require_cuda(src, out)
    check(_lib.load_library().vs_copy2d_pair(src.data_ptr(), dtype_code(src), lds, out.data_ptr(), dtype_code(out), ldd, rows, cols,
                                             _ptr(col_offset_dev), col_offset_scale, elem_offset_a, elem_offset_b, stream_ptr()), 'vs_copy2d_pair')
    return out


def colsum(x, M, N, out=None, accumulate=False):
    require_cuda(x)
    if out is None:
        out = torch.empty((N,), dtype=torch.float32, device=x.device)
    check(_lib.load_library().vs_colsum(x.data_ptr(), dtype_code(x), x.stride(0), M, N, out.data_ptr(),
                                        int(bool(accumulate)), stream_ptr()), 'vs_colsum')
    return out


def colsum_alloc(jobs, flat=None):
    """(flat, views): the result vectors colsum_multi(jobs) would allocate, for callers that launch it later (zero_flat=flat)."""
    if flat is None:
        flat = torch.empty((sum(x.shape[1] for x in jobs),), dtype=torch.float32, device=jobs[0].device)
    views, off = [], 0
    for x in jobs:
        views.append(flat[off:off + x.shape[1]])
        off += x.shape[1]
    return flat, views


def colsum_multi(jobs, outs=None, zero_flat=None):
    """jobs: list of (x [M, N] 2-D tensor).  Returns one fp32 vector of column sums per job (views of one flat buffer),
    computed by a single launch (+ one memset).  `outs`: caller-provided fp32 vectors that are ALREADY ZERO (the partial sums
    are added to them, e.g. slices of a zeroed all-reduce bucket) -- then nothing is allocated or filled here; with `zero_flat`
    (colsum_alloc) they are consecutive views of that buffer and are zeroed by the launch."""
    import ctypes
    results = []
    for i in range(0, len(jobs), 12):
        chunk = jobs[i:i + 12]
        n = len(chunk)
        require_cuda(*chunk)
        total = sum(x.shape[1] for x in chunk)
        if outs is None:
            flat = torch.empty((total,), dtype=torch.float32, device=chunk[0].device)
            views, off = [], 0
            for x in chunk:
                views.append(flat[off:off + x.shape[1]])
                off += x.shape[1]
            zero_base, zero_count = flat.data_ptr(), total
        else:
            views = outs[i:i + 12]
            for v, x in zip(views, chunk):
                assert v.dtype == torch.float32 and v.is_contiguous() and v.numel() == x.shape[1]
            require_cuda(*views)
            zero_base, zero_count = None, 0
            if zero_flat is not None:
                zero_base, zero_count = views[0].data_ptr(), sum(v.numel() for v in views)
        VP, I32, I64 = ctypes.c_void_p * n, ctypes.c_int * n, ctypes.c_int64 * n
        e0 = _pb()
        check(_lib.load_library().vs_colsum_multi(
            n, VP(*[x.data_ptr() for x in chunk]), I32(*[dtype_code(x) for x in chunk]), I64(*[x.stride(0) for x in chunk]),
            I64(*[x.shape[0] for x in chunk]), I64(*[x.shape[1] for x in chunk]), VP(*[v.data_ptr() for v in views]),
            zero_base, zero_count, stream_ptr()), 'vs_colsum_multi')
        _pe(e0, 'vs_colsum_multi', nbytes=float(sum(x.numel() * x.element_size() for x in chunk)))
        results += list(views)
    return results


def act_fwd(x, act, out=None, out_dtype=None):
    require_cuda(x)
    x = x.contiguous()
    if out is None:
        out = torch.empty(x.shape, dtype=out_dtype or x.dtype, device=x.device)
    check(_lib.load_library().vs_act_fwd(x.data_ptr(), dtype_code(x), out.data_ptr(), dtype_code(out), ACT[act],
                                         x.numel(), stream_ptr()), 'vs_act_fwd')
    return out


def act_bwd(dy, y, act, out_dtype=None):
    require_cuda(dy, y)
    dy, y = dy.contiguous(), y.contiguous()
    out = torch.empty(dy.shape, dtype=out_dtype or dy.dtype, device=dy.device)
    check(_lib.load_library().vs_act_bwd(dy.data_ptr(), dtype_code(dy), y.data_ptr(), dtype_code(y), out.data_ptr(),
                                         dtype_code(out), ACT[act], dy.numel(), stream_ptr()), 'vs_act_bwd')
    return out


def transpose_cast(src, dtype, out=None):
    """out[c, r] = src[r, c] converted to `dtype` (2-D, contiguous)."""
    require_cuda(src)
    assert src.dim() == 2 and src.is_contiguous()
    rows, cols = src.shape
    if out is None:
        out = torch.empty((cols, rows), dtype=dtype, device=src.device)
    check(_lib.load_library().vs_transpose_cast(src.data_ptr(), dtype_code(src), out.data_ptr(), dtype_code(out), rows,
                                                cols, stream_ptr()), 'vs_transpose_cast')
    return out


_roll_ws = {}


def _rollout_workspace(nbytes, device, geometry):
    """Exchange area of the multi-workgroup rollout: one per device AND geometry (compute type, B, C, H), zero-filled here once and then left to
    the library -- the weight-stationary form keeps per-slab epoch words in it and never clears it (include/varsep_hip.h), so an area is
    never shared between layouts."""
    if not nbytes:
        return None
    key = (device.index,) + tuple(geometry)
    buf = _roll_ws.get(key)
    if buf is None or buf.numel() != nbytes:
        buf = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        _roll_ws[key] = buf
    return buf


_GUARD = {}


def exchange_guard(device):
    """The process's exchange guard word (include/varsep_hip.h: vs_exchange_guard_set): one int32 device word that the kernels with a bounded
    in-launch exchange raise on a time-out and that every optimizer launch reads first (non-zero: no update).  Created and registered at the
    first use of such a kernel -- forward, i.e. before the step's first optimizer launch is issued or recorded.  One GPU per process (8e):
    a second device in the same process keeps the per-workspace words."""
    index = device.index if device.index is not None else torch.cuda.current_device()
    if index in _GUARD:
        return _GUARD[index]
    lib = _lib.load_library()
    if _GUARD or lib.vs_exchange_guard_get():
        _GUARD[index] = None
        return None
    word = torch.zeros((4,), dtype=torch.int32, device=device)              # (16 bytes: a line of its own would be better still; it is read-mostly)
    check(lib.vs_exchange_guard_set(word.data_ptr()), 'vs_exchange_guard_set')
    check(lib.vs_exchange_skip_counter_set(word.data_ptr() + 4), 'vs_exchange_skip_counter_set')      # word[1]: steps the guard made the optimizer skip
    _GUARD[index] = word
    return word


def exchange_skipped_steps(device, reset=True):
    """How many optimisation steps the raised guard word has made the step-count kernel skip since the last call (0 without a guard).
    Synchronises the device (reads a word)."""
    device = torch.device(device) if not isinstance(device, torch.device) else device
    index = device.index if device.index is not None else torch.cuda.current_device()
    guard = _GUARD.get(index)
    if guard is None:
        return 0
    n = int(guard[1].item())
    if reset and n:
        guard[1].zero_()
    return n


def rollout_exchange_error(device, reset=True):
    """Non-zero iff a bounded spin of an in-launch exchange timed out in any launch since the last call (the words are sticky: the
    library never clears them): bit 1 = the MLP integrator, bit 2 = the one-launch ConvResBlock layers.  Synchronises the device (reads words)."""
    device = torch.device(device) if not isinstance(device, torch.device) else device
    index = device.index if device.index is not None else torch.cuda.current_device()
    err = 0
    guard = _GUARD.get(index)
    if guard is not None:
        err |= int(guard[0].item())
        if reset and err:
            guard[0].zero_()
    for key, buf in list(_roll_ws.items()):
        if key[0] != index:
            continue
        word = buf[buf.numel() // 16 * 16 - 16:buf.numel() // 16 * 16 - 12].view(torch.int32)
        e1 = int(word.item())
        err |= e1
        if reset and e1:
            word.zero_()
    st = _IMGBN.get(index)                              # the fused ConvResBlock layers' exchange (conv3_img16_bn_*): word 1 of their workspace
    if st is not None:
        e2 = int(st['ws'][1].item())
        if e2:
            err |= 2
            if reset:
                st['ws'][1].zero_()
    return err


_XL_PROBED = set()


def rollout_xcd_local(allowed=None):
    """Process-wide switch of the integrator's XCD-local exchange (None: leave).  False selects the placement-independent agent-scope stores."""
    if allowed is not None:
        check(_lib.load_library().vs_mlp_rollout_xcd_local_set(1 if allowed else 0), 'vs_mlp_rollout_xcd_local_set')


def _probe_rollout_exchange(code, B, C, H, nb, device):
    """START-UP decision of the exchange mode (once per device and geometry, outside stream captures).  The XCD-local form of the
    weight-stationary integrator (csrc/vs_rollout.hip, wsr::gstore<true>) leans on workgroups 8 apart sharing an XCD, which HIP does not
    promise.  One probe rollout -- three steps on zero weights, a fresh zero-filled exchange area, a short spin limit -- either completes (the
    property holds on this device / runtime: keep the 210 ns hops) or leaves the error word raised: then every launch of this process uses
    the agent-scope stores (385 ns hops, any placement).  A mid-run change of the dispatch order is still caught by the epoch tags -> the
    guard word -> skipped optimizer steps -> train.recover_exchange."""
    import os
    import sys
    key = (device.index, code, B, C, H, nb)
    if key in _XL_PROBED or torch.cuda.is_current_stream_capturing():
        return
    _XL_PROBED.add(key)
    lib = _lib.load_library()
    if not lib.vs_mlp_rollout_xcd_local_get(code, B, C, H, nb) or os.environ.get('VARSEP_ROLLOUT_PROBE', '1') == '0':
        return
    import ctypes
    dt = {BF16: torch.bfloat16, F16: torch.float16}[code]
    nbytes = lib.vs_mlp_rollout_workspace_bytes(code, B, C, H)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=device)
    n = 3
    weights = []
    for _ in range(nb):
        for (N, K) in ((H, C), (H, H), (C, H)):
            weights.append(torch.zeros((lib.vs_rollout_packed_elems(code, N, K),), dtype=dt, device=device))
    biases = []
    for _ in range(nb):
        biases += [torch.zeros((H,), device=device), torch.zeros((H,), device=device), torch.zeros((C,), device=device)]
    x0 = torch.zeros((B, C), device=device)
    t_codes = torch.empty((B, n, C), device=device)
    xin = torch.empty((nb, n - 1, B, C), dtype=dt, device=device)
    h1 = torch.empty((nb, n - 1, B, H), dtype=dt, device=device)
    h2 = torch.empty((nb, n - 1, B, H), dtype=dt, device=device)
    parts = lib.vs_mlp_rollout_parts(code, B, C, H)
    Bp = (B + 15) // 16 * 16
    m1 = torch.empty((nb, n - 1, Bp, parts, 32), dtype=torch.int32, device=device)
    m2 = torch.empty_like(m1)
    guard = _GUARD.get(device.index if device.index is not None else torch.cuda.current_device())
    before = int(guard[0].item()) if guard is not None else 0
    saved = os.environ.get('VS_ROLLOUT_SPIN_LIMIT')
    os.environ['VS_ROLLOUT_SPIN_LIMIT'] = os.environ.get('VARSEP_ROLLOUT_PROBE_SPINS', str(1 << 16))
    try:
        wa, ba = _ptr_array(weights), _ptr_array(biases)
        check(lib.vs_mlp_rollout_fwd(code, B, C, H, nb, n, x0.data_ptr(), ctypes.cast(wa, ctypes.c_void_p), ctypes.cast(ba, ctypes.c_void_p),
                                     t_codes.data_ptr(), None, xin.data_ptr(), h1.data_ptr(), h2.data_ptr(), m1.data_ptr(), m2.data_ptr(),
                                     ws.data_ptr(), ws.numel(), stream_ptr()), 'vs_mlp_rollout_fwd (placement probe)')
        torch.cuda.current_stream().synchronize()
    finally:
        if saved is None:
            os.environ.pop('VS_ROLLOUT_SPIN_LIMIT', None)
        else:
            os.environ['VS_ROLLOUT_SPIN_LIMIT'] = saved
    if guard is not None:
        failed = int(guard[0].item()) != 0
        guard[0].fill_(before)                               # the probe's verdict is consumed here; an earlier error stays
    else:
        failed = int(ws[ws.numel() // 16 * 16 - 16:ws.numel() // 16 * 16 - 12].view(torch.int32).item()) != 0
    if failed:
        rollout_xcd_local(False)
        sys.stderr.write('varsep: the integrator\'s XCD-local exchange did not complete its start-up probe on this device (workgroups 8 apart do not '
                         'share an XCD here): using the agent-scope exchange for this process\n')


def _ptr_array(tensors):
    import ctypes
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


def pack_rollout_weight(w, dtype, transpose, out=None):
    """fp32 master weight [rows, cols] -> packed MFMA-fragment order in `dtype` (logical L = w or w^T)."""
    require_cuda(w)
    assert w.dim() == 2 and w.is_contiguous() and w.dtype == torch.float32
    N, K = (w.shape[1], w.shape[0]) if transpose else (w.shape[0], w.shape[1])
    lib = _lib.load_library()
    code = code_of(dtype)
    n = lib.vs_rollout_packed_elems(code, N, K)
    if out is None:
        out = torch.empty((n,), dtype=dtype, device=w.device)
    check(lib.vs_pack_rollout_weight(code, w.data_ptr(), int(bool(transpose)), N, K, out.data_ptr(), stream_ptr()),
          'vs_pack_rollout_weight')
    return out


def pack_rollout_weights(jobs, dtype):
    """jobs: list of (fp32 weight [rows, cols], transpose, out buffer or None).  One launch; returns the packed buffers."""
    import ctypes
    lib = _lib.load_library()
    code = code_of(dtype)
    outs = []
    for i in range(0, len(jobs), 48):
        chunk = jobs[i:i + 48]
        n = len(chunk)
        Ns, Ks, bufs = [], [], []
        for w, tr, out in chunk:
            require_cuda(w)
            assert w.dim() == 2 and w.is_contiguous() and w.dtype == torch.float32
            N, K = (w.shape[1], w.shape[0]) if tr else (w.shape[0], w.shape[1])
            if out is None:
                out = torch.empty((lib.vs_rollout_packed_elems(code, N, K),), dtype=dtype, device=w.device)
            Ns.append(N); Ks.append(K); bufs.append(out)
        VP, I32 = ctypes.c_void_p * n, ctypes.c_int * n
        check(lib.vs_pack_rollout_weights(code, n, VP(*[w.data_ptr() for w, _, _ in chunk]), I32(*[int(bool(tr)) for _, tr, _ in chunk]),
                                          I32(*Ns), I32(*Ks), VP(*[b.data_ptr() for b in bufs]), stream_ptr()), 'vs_pack_rollout_weights')
        outs += bufs
    return outs


def mlp_rollout_fwd(x0, weights, biases, n_steps, H, want_residuals=True):
    """weights: packed [W1,W2,W3]*n_blocks in the compute dtype; biases fp32.  Returns (t_codes, residuals, saves)."""
    import ctypes
    require_cuda(x0)
    B, C = x0.shape
    nb = len(weights) // 3
    cdt, dev = weights[0].dtype, x0.device
    t_codes = torch.empty((B, n_steps, C), dtype=torch.float32, device=dev)
    steps = max(n_steps - 1, 0)
    residuals = torch.empty((steps, nb, B, C), dtype=torch.float32, device=dev) if want_residuals else None
    xin = torch.empty((nb, steps, B, C), dtype=cdt, device=dev)
    h1 = torch.empty((nb, steps, B, H), dtype=cdt, device=dev)
    h2 = torch.empty((nb, steps, B, H), dtype=cdt, device=dev)
    lib = _lib.load_library()
    code = dtype_code(weights[0])
    parts = lib.vs_mlp_rollout_parts(code, B, C, H)
    Bp = (B + 15) // 16 * 16                 # sign-bit arrays: rows padded to whole 16-row slabs
    m1 = torch.empty((nb, steps, Bp, parts, 32), dtype=torch.int32, device=dev)
    m2 = torch.empty((nb, steps, Bp, parts, 32), dtype=torch.int32, device=dev)
    exchange_guard(dev)
    _probe_rollout_exchange(code, B, C, H, nb, dev)
    xws = _rollout_workspace(lib.vs_mlp_rollout_workspace_bytes(code, B, C, H), dev, (code, B, C, H))
    wa, ba = _ptr_array(weights), _ptr_array(biases)
    e0 = _pb()
    check(_lib.load_library().vs_mlp_rollout_fwd(dtype_code(weights[0]), B, C, H, nb, n_steps, x0.data_ptr(),
                                                 ctypes.cast(wa, ctypes.c_void_p), ctypes.cast(ba, ctypes.c_void_p),
                                                 t_codes.data_ptr(), _ptr(residuals), xin.data_ptr(), h1.data_ptr(),
                                                 h2.data_ptr(), m1.data_ptr(), m2.data_ptr(), _ptr(xws),
                                                 xws.numel() if xws is not None else 0, stream_ptr()),
          'vs_mlp_rollout_fwd')
    fl = 2.0 * B * steps * nb * (2 * C * H + H * H)
    _pe(e0, 'vs_mlp_rollout_fwd<%s>' % _DT[code_of(cdt)], flops=fl)
    return t_codes, residuals, (xin, h1, h2, m1, m2)


def mlp_rollout_bwd(grad_t_codes, weights_t, h1, h2, m1, m2, n_steps):
    """weights_t: packed [W3^T, W2^T, W1^T]*n_blocks (compute dtype).  Returns (dx0, dr, dh2, dh1)."""
    import ctypes
    require_cuda(grad_t_codes)
    B, n, C = grad_t_codes.shape
    nb = len(weights_t) // 3
    H = h1.shape[-1]
    cdt, dev = weights_t[0].dtype, grad_t_codes.device
    steps = max(n_steps - 1, 0)
    dx0 = torch.empty((B, C), dtype=torch.float32, device=dev)
    dr = torch.empty((nb, steps, B, C), dtype=cdt, device=dev)
    dh2 = torch.empty((nb, steps, B, H), dtype=cdt, device=dev)
    dh1 = torch.empty((nb, steps, B, H), dtype=cdt, device=dev)
    wa = _ptr_array(weights_t)
    lib = _lib.load_library()
    xws = _rollout_workspace(lib.vs_mlp_rollout_workspace_bytes(dtype_code(weights_t[0]), B, C, H), dev, (dtype_code(weights_t[0]), B, C, H))
    e0 = _pb()
    check(_lib.load_library().vs_mlp_rollout_bwd(dtype_code(weights_t[0]), B, C, H, nb, n_steps, grad_t_codes.data_ptr(),
                                                 ctypes.cast(wa, ctypes.c_void_p), h1.data_ptr(), h2.data_ptr(),
                                                 m1.data_ptr(), m2.data_ptr(), dx0.data_ptr(), dr.data_ptr(), dh2.data_ptr(), dh1.data_ptr(),
                                                 _ptr(xws), xws.numel() if xws is not None else 0, stream_ptr()),
          'vs_mlp_rollout_bwd')
    fl = 2.0 * B * steps * nb * (2 * C * H + H * H)
    _pe(e0, 'vs_mlp_rollout_bwd<%s>' % _DT[code_of(cdt)], flops=fl)
    return dx0, dr, dh2, dh1


# ------------------------------------------------------------------------------------------------ convolutions
def _conv_out_hw(H, W, k, stride, pad, transposed):
    if transposed:
        return (H - 1) * stride - 2 * pad + k, (W - 1) * stride - 2 * pad + k
    return (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1


def conv_pack_weight(w_master, dtype, stride, pad, out=None):
    """fp32 master weight [D0, D1, k, k] -> packed operand of the transposed-form contractions (compute dtype)."""
    require_cuda(w_master)
    assert w_master.dtype == torch.float32 and w_master.is_contiguous()
    D0, D1, kh, kw = w_master.shape
    lib = _lib.load_library()
    if out is None:
        out = torch.empty((lib.vs_conv_packed_elems(D0, D1, kh, kw, stride, pad),), dtype=dtype, device=w_master.device)
    check(lib.vs_conv_pack_weight(code_of(dtype), w_master.data_ptr(), D0, D1, kh, kw, stride, pad,
                                  out.data_ptr(), stream_ptr()), 'vs_conv_pack_weight')
    return out


def _conv_workspace(x_dtype_code, B, Cin, H, W, Cout, k, stride, pad, device):
    """Transient column-matrix + split-K workspace of a convolution (VS_CONV_COLS=0: none -> gather inside the MFMA loop)."""
    import os
    if os.environ.get('VS_CONV_COLS') == '0':
        return None
    nbytes = _lib.load_library().vs_conv_workspace_bytes(x_dtype_code, B, Cin, H, W, Cout, k, k, stride, pad)
    return _workspace(nbytes, device) if nbytes else None


# ---- convolutions with 1..8 channels on the image side (csrc/vs_conv_thin.hip): dispatched from conv_fwd / conv_dgrad / conv_wgrad ----------
def _thin_plan(op, code, B, Cin, H, W, Cout, OH, OW, k, stride, pad, transposed, wgrad_max_m=None):
    """Whether this convolution call is one of the thin forms; returns (kind, C, Hb, Wb, M, sc, sm, flip) -- the many-channel map's channels and
    size, the thin channel count and how element (c, m, t) of the kernel's weight view is found in the weight tensor -- or None.
    VS_CONV_THIN=0: never.  The weight gradient takes its thin kernel up to `wgrad_max_m` thin channels (VS_CONV_THIN_WGRAD_MAX_M, default 8 since
    round 4: no convolution of the BASELINE steps builds a column matrix any more.  With 4 / 5 / 8 thin channels -- the first encoder layers --
    the column-matrix GEMM was the faster form, 57-72 us against 78-129 us: the VALU form re-reads the map per pair of thin channels; 2
    restores that choice)."""
    import os
    if wgrad_max_m is None:
        wgrad_max_m = int(os.environ.get('VS_CONV_THIN_WGRAD_MAX_M', '8'))
    if os.environ.get('VS_CONV_THIN') == '0' or code == F32 or pad != 1 or (k, stride) not in ((3, 1), (4, 2)):
        return None
    k2 = k * k
    plan = None
    if op == 'fwd':
        if not transposed and Cin <= 8 and (H, W) == (stride * OH, stride * OW):
            plan = ('expand', Cout, OH, OW, Cin, Cin * k2, k2, 0)
        elif not transposed and stride == 1 and Cout <= 4:
            plan = ('reduce', Cin, H, W, Cout, k2, Cin * k2, 1)
        elif transposed and Cout <= (4 if k == 3 else 2) and (OH, OW) == (stride * H, stride * W):
            plan = ('reduce', Cin, H, W, Cout, Cout * k2, k2, 0)
    elif op == 'dgrad':
        if transposed and Cout <= 8 and (OH, OW) == (stride * H, stride * W):
            plan = ('expand', Cin, H, W, Cout, Cout * k2, k2, 0)
        elif not transposed and stride == 1 and Cout <= 8:
            plan = ('expand', Cin, H, W, Cout, k2, Cin * k2, 1)
    else:
        if not transposed and Cin <= wgrad_max_m and (H, W) == (stride * OH, stride * OW):
            plan = ('wgrad_big_dy', Cout, OH, OW, Cin, Cin * k2, k2, 0)
        elif not transposed and stride == 1 and Cout <= wgrad_max_m:
            plan = ('wgrad_big_x', Cin, H, W, Cout, k2, Cin * k2, 1)
        elif transposed and Cout <= wgrad_max_m and (OH, OW) == (stride * H, stride * W):
            plan = ('wgrad_big_x', Cin, H, W, Cout, Cout * k2, k2, 0)
    if plan is None or not _lib.load_library().vs_conv_thin_supported(code, B, plan[1], plan[2], plan[3], plan[4], k, stride, pad):
        return None
    return plan


def conv_thin_expand(thin, w, bias, out_shape, out_dtype, k, stride, sc, sm, flip, role='fwd'):
    """out [B, C, H, W] = bias[c] + sum_{m, t} thin[B, M, S H, S W][m][S p + t - 1] * w(c, m, t)  (vs_conv_thin_expand)."""
    require_cuda(thin, w, bias)
    B, C, H, W = out_shape
    out = torch.empty(tuple(out_shape), dtype=out_dtype, device=thin.device)
    e0 = _pb()
    check(_lib.load_library().vs_conv_thin_expand(dtype_code(thin), thin.data_ptr(), w.data_ptr(), sc, sm, flip, _ptr(bias), out.data_ptr(), dtype_code(out),
                                                  B, C, H, W, thin.shape[1], k, stride, stream_ptr()), 'vs_conv_thin_expand')
    _pe(e0, 'vs_conv_thin:%s<%s>' % (role, _DT[dtype_code(thin)]), flops=2.0 * B * C * H * W * thin.shape[1] * k * k,
        nbytes=float(thin.numel() * thin.element_size() + out.numel() * out.element_size()))
    return out


def conv_thin_reduce(big, w, bias, M, out_dtype, k, stride, sc, sm, flip, role='fwd'):
    """out [B, M, S H, S W] = bias[m] + sum over the pixels / taps of big [B, C, H, W] that meet each output pixel  (vs_conv_thin_reduce)."""
    require_cuda(big, w, bias)
    B, C, H, W = big.shape
    out = torch.empty((B, M, stride * H, stride * W), dtype=out_dtype, device=big.device)
    e0 = _pb()
    check(_lib.load_library().vs_conv_thin_reduce(dtype_code(big), big.data_ptr(), w.data_ptr(), sc, sm, flip, _ptr(bias), out.data_ptr(), dtype_code(out),
                                                  B, C, H, W, M, k, stride, stream_ptr()), 'vs_conv_thin_reduce')
    _pe(e0, 'vs_conv_thin:%s<%s>' % (role, _DT[dtype_code(big)]), flops=2.0 * B * C * H * W * M * k * k,
        nbytes=float(big.numel() * big.element_size() + out.numel() * out.element_size()))
    return out


def conv_thin_wgrad(big, thin, w_shape, k, stride, sc, sm, flip, into=None, out=None):
    """dW (the weight's own layout, fp32) (+= into) = sum_{maps, p} big[c][p] * thin[m][S p + t - 1]  (vs_conv_thin_wgrad)."""
    require_cuda(big, thin)
    B, C, H, W = big.shape
    M = thin.shape[1]
    lib = _lib.load_library()
    dw = into if into is not None else (out if out is not None else torch.empty(tuple(w_shape), dtype=torch.float32, device=big.device))
    ws = _workspace(lib.vs_conv_thin_wgrad_workspace_bytes(B, C, H, W, M, k), big.device)
    e0 = _pb()
    check(lib.vs_conv_thin_wgrad(dtype_code(big), big.data_ptr(), thin.data_ptr(), ws.data_ptr(), ws.numel(), _ptr(into), dw.data_ptr(), sc, sm, flip,
                                 B, C, H, W, M, k, stride, stream_ptr()), 'vs_conv_thin_wgrad')
    _pe(e0, 'vs_conv_thin:wgrad<%s>' % _DT[dtype_code(big)], flops=2.0 * B * C * H * W * M * k * k,
        nbytes=float(big.numel() * big.element_size() + thin.numel() * thin.element_size() + dw.numel() * 4))
    return dw


# ---- ConvTranspose2d on a 1x1 map (the decoders' first_upconv, reference conv.py:258, 295: [B, C, 1, 1] -> [B, Cout, k, k]) IS a dense GEMM with
# the weight [Cin][Cout k k] as it lies in memory: no column matrix, no scatter ------------------------------------------------------------------
def _convt_1x1(x_shape, k, stride, pad, transposed):
    import os
    return (transposed and x_shape[2] == 1 and x_shape[3] == 1 and pad == 0 and stride == 1 and os.environ.get('VS_CONVT_1X1_GEMM', '1') == '1')


def _conv_full(x_shape, k, stride, pad, transposed):
    """Conv2d whose window is the whole (unpadded) map -- the encoders' `last_op` (reference conv.py:123, 169: Conv2d(8 nf, nh, 4, 1, 0) on a
    4 x 4 map -> 1 x 1): a Linear layer on the map as it lies in memory, y[b][co] = sum_j x[b][j] w[co][j], j = (ci, ky, kx).  No gather."""
    import os
    return (not transposed and pad == 0 and x_shape[2] == k and x_shape[3] == k and os.environ.get('VS_CONV_FULL_GEMM', '1') == '1')


# ---- fp32 results from the 16-bit row-band kernels (VARSEP_FP32_SPLIT=1) ---------------------------------------------------------------------
# The row-band kernels take 16-bit operands, so the fp32 parity tests (1e-3 against the reference's fixture) ran the column-matrix route and
# never touched the kernels bench.py times.  A convolution is bilinear: with x = x0 + x1 + x2 and w = w0 + w1 + w2 (three bf16 pieces each: the
# bf16 head, the bf16 rounding of the remainder, and again -- 24 significant bits, i.e. the fp32 value exactly),
#   conv(x, w) = sum over i + j <= 2 of conv(x_i, w_j)  +  O(2^-24 |x| |w|):
# six launches of the SAME kernel (bf16 products are exact in the fp32 accumulator, fp32 output) reproduce the fp32 convolution to fp32
# rounding.  (Two pieces / three products leave 2^-16, which the VGG stack's per-call BatchNorms amplify to 1.3e-2 on the first BatchNorm
# weight's gradient: measured, above the 1e-2 bar.)  A test mode -- 6x the launches --, not a training mode.
_SPLIT_TERMS = ((0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1))


def fp32_split_enabled():
    import os
    return os.environ.get('VARSEP_FP32_SPLIT', '0') == '1'


def _split16(t):
    parts, rest = [], t
    for _ in range(3):
        p = rest.to(torch.bfloat16)
        parts.append(p)
        rest = rest - p.float()
    return parts


def _conv3_split(x, w, bias, flip, role):
    """fp32 Conv2d k3 s1 p1 (flip: its input gradient, w [Cout, Cin, 3, 3] applied transposed) through six bf16 row-band launches."""
    Cout = w.shape[1] if flip else w.shape[0]
    xs = _split16(x)
    few = conv3_img16_supported(xs[0], Cout)         # the SST integrator's few 16 x 16 maps: the few-maps kernel (same weight pre-pack)
    if not few and not conv3_band_supported(xs[0], Cout):
        return None
    ws = [conv3_img16_pack_weight(p.float().contiguous(), torch.bfloat16, flip) for p in _split16(w)]
    y = None
    for i, j in _SPLIT_TERMS:
        if few:
            t = slab_sum(conv3_img16(xs[i], ws[j], Cout, role=role), bias if y is None else None, torch.float32)
        else:
            t = conv3_band(xs[i], ws[j], bias if y is None else None, Cout, torch.float32, role=role)
        y = t if y is None else y.add_(t)
    return y


def _k4s2_split_gather(big, w, bias, role):
    """fp32 k4 s2 p1 gather (Conv2d forward on x / ConvTranspose2d input gradient on dy; w [M, K, 4, 4]) through six launches of the 16-bit
    parity-plane row-band kernel."""
    M = w.shape[0]
    bs = _split16(big)
    if not conv_k4s2_gather_supported(bs[0], M):
        return None
    planes = [space_to_depth2(b) for b in bs]
    ws = [conv_k4s2_pack_weight(p.float().contiguous(), torch.bfloat16) for p in _split16(w)]
    y = None
    for i, j in _SPLIT_TERMS:
        t = conv_k4s2_gather(planes[i], ws[j], bias if y is None else None, M, torch.float32, role=role)
        y = t if y is None else y.add_(t)
    return y


def _tap_split(x, w, bias, role):
    """fp32 k4 s2 p1 SCATTER (ConvTranspose2d forward of x with w [Cin, Cout, 4, 4]; the input gradient of a Conv2d k4 s2 p1 is the same operation
    on dy with its weight read as [in = Cout, out = Cin]) through six launches of the 16-bit tap kernel with fp32 output."""
    Cout = w.shape[1]
    xs = _split16(x)
    if not convt_tap_supported(xs[0], Cout, 1):
        return None
    ws = [convt_tap_pack_weight(p.float().contiguous(), torch.bfloat16) for p in _split16(w)]
    B, Cin, H, W = x.shape
    lib = _lib.load_library()
    y = None
    for i, j in _SPLIT_TERMS:
        t = torch.empty((B, Cout, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
        e0 = _pb()
        check(lib.vs_convt_k4s2_tap_fwd_f32(_lib.BF16, xs[i].data_ptr(), ws[j].data_ptr(), _ptr(bias) if y is None else None, t.data_ptr(), B, Cin, H, W, Cout,
                                            stream_ptr()), 'vs_convt_k4s2_tap_fwd_f32')
        _pe(e0, 'vs_convT_tap:%s<bf16>' % role, flops=2.0 * B * H * W * Cin * Cout * 16, nbytes=float(xs[i].numel() * 2 + t.numel() * 4))
        y = t if y is None else y.add_(t)
    return y


def _k4s2_split_wgrad(small, big, w_shape, into, out):
    M = w_shape[0]
    bs = _split16(big)
    if not conv_k4s2_supported(bs[0], M):
        return None
    planes = [space_to_depth2(b) for b in bs]
    ss = _split16(small)
    dw = None
    for i, j in _SPLIT_TERMS:
        dw = conv_k4s2_wgrad(ss[j], planes[i], w_shape, into=into if dw is None else dw, out=out if dw is None else None)
    return dw


def conv_fwd(x, w, bias, stride, pad, transposed, out_dtype, w_packed=None):
    """x [B,Cin,H,W], w Conv2d [Cout,Cin,k,k] / ConvTranspose2d [Cin,Cout,k,k] in the same (compute) dtype.
    The transposed form consumes `w_packed` (conv_pack_weight); it is built on the fly from `w` when not given."""
    require_cuda(x, w, bias)
    assert x.is_contiguous() and w.is_contiguous() and x.dtype == w.dtype
    if (x.dtype == torch.float32 and out_dtype == torch.float32 and not transposed and tuple(w.shape[2:]) == (3, 3) and stride == 1 and pad == 1
            and fp32_split_enabled()):
        y = _conv3_split(x, w, bias, False, 'fwd')
        if y is not None:
            return y
    if (x.dtype == torch.float32 and out_dtype == torch.float32 and not transposed and tuple(w.shape[2:]) == (4, 4) and stride == 2 and pad == 1
            and fp32_split_enabled()):
        y = _k4s2_split_gather(x, w, bias, 'fwd')
        if y is not None:
            return y
    if (x.dtype == torch.float32 and out_dtype == torch.float32 and transposed and tuple(w.shape[2:]) == (4, 4) and stride == 2 and pad == 1
            and fp32_split_enabled()):
        y = _tap_split(x, w, bias, 'fwd')
        if y is not None:
            return y
    if x.dtype == torch.float32 and out_dtype == torch.float32 and w.shape[2] == w.shape[3] and fp32_split_enabled():
        # the thin-channel edge layers (first encoder / last decoder convolution): the same six products on the VALU kernels
        B_, Cin_, H_, W_ = x.shape
        k_ = w.shape[2]
        Cout_ = w.shape[1] if transposed else w.shape[0]
        OH_, OW_ = _conv_out_hw(H_, W_, k_, stride, pad, transposed)
        thin = _thin_plan('fwd', BF16, B_, Cin_, H_, W_, Cout_, OH_, OW_, k_, stride, pad, transposed)
        if thin is not None:
            xs, ws, y = _split16(x), _split16(w), None
            for i, j in _SPLIT_TERMS:
                b_ = bias if y is None else None
                t = (conv_thin_expand(xs[i], ws[j], b_, (B_, Cout_, OH_, OW_), torch.float32, k_, stride, *thin[5:]) if thin[0] == 'expand'
                     else conv_thin_reduce(xs[i], ws[j], b_, Cout_, torch.float32, k_, stride, *thin[5:]))
                y = t if y is None else y.add_(t)
            return y
    if transposed:
        if w_packed is None:
            w_packed = conv_pack_weight(w.float().contiguous(), x.dtype, stride, pad)
        w_arg = w_packed
    else:
        w_arg = w
    B, Cin, H, W = x.shape
    k = w.shape[2]
    Cout = w.shape[1] if transposed else w.shape[0]
    OH, OW = _conv_out_hw(H, W, k, stride, pad, transposed)
    if _convt_1x1(x.shape, k, stride, pad, transposed):
        # y[b][(co, ky, kx)] = sum_ci x[b][ci] w[ci][(co, ky, kx)] + bias[co]
        n = Cout * k * w.shape[3]
        brep = bias.repeat_interleave(k * w.shape[3]) if bias is not None else None
        y = gemm(x.view(B, Cin), LAYOUT_R, w.view(Cin, n), LAYOUT_S, B, n, Cin, out_dtype=out_dtype, bias=brep)
        return y.view(B, Cout, k, w.shape[3])
    if _conv_full(x.shape, k, stride, pad, transposed) and w.shape[3] == k:
        n = Cin * k * k
        return gemm(x.view(B, n), LAYOUT_R, w.view(Cout, n), LAYOUT_R, B, Cout, n, out_dtype=out_dtype, bias=bias).view(B, Cout, 1, 1)
    thin = _thin_plan('fwd', dtype_code(x), B, Cin, H, W, Cout, OH, OW, k, stride, pad, transposed) if w.shape[2] == w.shape[3] else None
    if thin is not None:
        if thin[0] == 'expand':
            return conv_thin_expand(x, w, bias, (B, Cout, OH, OW), out_dtype, k, stride, *thin[5:])
        return conv_thin_reduce(x, w, bias, Cout, out_dtype, k, stride, *thin[5:])
    y = torch.empty((B, Cout, OH, OW), dtype=out_dtype, device=x.device)
    lib = _lib.load_library()
    fn = lib.vs_conv_transpose2d_fwd if transposed else lib.vs_conv2d_fwd
    ws = _conv_workspace(dtype_code(x), B, Cin, H, W, Cout, k, stride, pad, x.device)
    e0 = _pb()
    check(fn(dtype_code(x), x.data_ptr(), w_arg.data_ptr(), _ptr(bias), y.data_ptr(), dtype_code(y), B, Cin, H, W, Cout, k, k,
             stride, pad, _ptr(ws), ws.numel() if ws is not None else 0, stream_ptr()), 'vs_conv_fwd')
    _pe(e0, 'vs_conv%s_cols:fwd<%s>' % ('T' if transposed else '', _DT[dtype_code(x)]),
        flops=2.0 * B * Cout * OH * OW * Cin * k * k if not transposed else 2.0 * B * Cin * H * W * Cout * k * k,
        nbytes=float(x.numel() * x.element_size() + w.numel() * w.element_size() + y.numel() * y.element_size()))
    return y


def conv_dgrad(dy, w, x_shape, stride, pad, transposed, out_dtype, w_packed=None, cols_from_wgrad=False):
    """cols_from_wgrad (ConvTranspose2d only): conv_wgrad of the same dy / geometry ran just before on this stream, reuse its column
    matrix instead of gathering dy a second time."""
    require_cuda(dy, w)
    assert dy.is_contiguous() and w.is_contiguous() and dy.dtype == w.dtype
    if (dy.dtype == torch.float32 and out_dtype == torch.float32 and not transposed and tuple(w.shape[2:]) == (3, 3) and stride == 1 and pad == 1
            and fp32_split_enabled()):
        dx = _conv3_split(dy, w, None, True, 'dgrad')
        if dx is not None:
            return dx
    if (dy.dtype == torch.float32 and out_dtype == torch.float32 and transposed and tuple(w.shape[2:]) == (4, 4) and stride == 2 and pad == 1
            and fp32_split_enabled()):
        dx = _k4s2_split_gather(dy, w, None, 'dgrad')
        if dx is not None:
            return dx
    if (dy.dtype == torch.float32 and out_dtype == torch.float32 and not transposed and tuple(w.shape[2:]) == (4, 4) and stride == 2 and pad == 1
            and fp32_split_enabled()):
        dx = _tap_split(dy, w, None, 'dgrad')        # w [Cout, Cin, 4, 4] read as a ConvTranspose2d weight [in = Cout, out = Cin]
        if dx is not None:
            return dx
    if dy.dtype == torch.float32 and out_dtype == torch.float32 and w.shape[2] == w.shape[3] and fp32_split_enabled():
        B_, Cin_, H_, W_ = x_shape
        k_ = w.shape[2]
        Cout_ = w.shape[1] if transposed else w.shape[0]
        thin = _thin_plan('dgrad', BF16, B_, Cin_, H_, W_, Cout_, dy.shape[2], dy.shape[3], k_, stride, pad, transposed)
        if thin is not None:
            ds, ws, dx = _split16(dy), _split16(w), None
            for i, j in _SPLIT_TERMS:
                t = conv_thin_expand(ds[i], ws[j], None, (B_, Cin_, H_, W_), torch.float32, k_, stride, *thin[5:], role='dgrad')
                dx = t if dx is None else dx.add_(t)
            return dx
    if not transposed:
        if w_packed is None:
            w_packed = conv_pack_weight(w.float().contiguous(), dy.dtype, stride, pad)
        w_arg = w_packed
    else:
        w_arg = w
    B, Cin, H, W = x_shape
    k = w.shape[2]
    Cout = w.shape[1] if transposed else w.shape[0]
    if _convt_1x1(x_shape, k, stride, pad, transposed):
        # dx[b][ci] = sum_j dy[b][j] w[ci][j], j = (co, ky, kx)
        n = Cout * k * w.shape[3]
        return gemm(dy.view(B, n), LAYOUT_R, w.view(Cin, n), LAYOUT_R, B, Cin, n, out_dtype=out_dtype).view(B, Cin, 1, 1)
    if _conv_full(x_shape, k, stride, pad, transposed) and w.shape[3] == k:
        # dx[b][j] = sum_co dy[b][co] w[co][j]
        n = Cin * k * k
        return gemm(dy.view(B, Cout), LAYOUT_R, w.view(Cout, n), LAYOUT_S, B, n, Cout, out_dtype=out_dtype).view(B, Cin, k, k)
    thin = (_thin_plan('dgrad', dtype_code(dy), B, Cin, H, W, Cout, dy.shape[2], dy.shape[3], k, stride, pad, transposed)
            if w.shape[2] == w.shape[3] else None)
    if thin is not None:
        return conv_thin_expand(dy, w, None, (B, Cin, H, W), out_dtype, k, stride, *thin[5:], role='dgrad')
    dx = torch.empty((B, Cin, H, W), dtype=out_dtype, device=dy.device)
    lib = _lib.load_library()
    fn = lib.vs_conv_transpose2d_dgrad if transposed else lib.vs_conv2d_dgrad
    ws = _conv_workspace(dtype_code(dy), B, Cin, H, W, Cout, k, stride, pad, dy.device)
    e0 = _pb()
    extra = (int(bool(cols_from_wgrad)),) if transposed else ()
    check(fn(dtype_code(dy), dy.data_ptr(), w_arg.data_ptr(), dx.data_ptr(), dtype_code(dx), B, Cin, H, W, Cout, k, k, stride, pad,
             _ptr(ws), ws.numel() if ws is not None else 0, *extra, stream_ptr()), 'vs_conv_dgrad')
    OH, OW = dy.shape[2], dy.shape[3]
    _pe(e0, 'vs_conv%s_cols:dgrad<%s>' % ('T' if transposed else '', _DT[dtype_code(dy)]),
        flops=2.0 * B * Cout * OH * OW * Cin * k * k if not transposed else 2.0 * B * Cin * H * W * Cout * k * k,
        nbytes=float(dy.numel() * dy.element_size() + w.numel() * w.element_size() + dx.numel() * dx.element_size()))
    return dx


def conv3_wgrad_band_supported(x, Cout, dtype_code_of=None):
    """Whether the weight gradient of Conv2d k3 s1 p1 with input `x` takes the row-band kernel (no column matrix).  VS_CONV_WGRAD_BAND=0: never.
    (`x` may be a meta tensor that only carries the shape; then `dtype_code_of` is the operand type code.)"""
    import os
    if os.environ.get('VS_CONV_WGRAD_BAND') == '0' or x.dtype == torch.float32 or x.dim() != 4:
        return False
    B, Cin, H, W = x.shape
    if (Cin * Cout * 9) % 4 != 0:
        return False
    return bool(_lib.load_library().vs_conv3_wgrad_band_supported(dtype_code_of if dtype_code_of is not None else dtype_code(x), B, Cin, H, W, Cout))


def _slab_reduce(slabs, nslabs, w_shape, into, dw):
    """dw = sum of `nslabs` fp32 partial gradients (+ the pending gradient `into`); two passes when a small weight has many slabs."""
    lib = _lib.load_library()
    src, n = slabs, nslabs
    if nslabs > 24:
        # many slabs of a small weight: a first pass in 16 groups (every workgroup adds <= nslabs / 16 coalesced slabs), then the 16 partials
        src = torch.empty((16,) + tuple(w_shape), dtype=torch.float32, device=slabs.device)
        check(lib.vs_slab_sum_grouped(slabs.data_ptr(), nslabs, 16, src.data_ptr(), dw.numel(), stream_ptr()), 'vs_slab_sum_grouped')
        n = -(-nslabs // (-(-nslabs // 16)))          # groups that received slabs: ceil(nslabs / per)
    # the slabs are laid out [tap][Cout][Cin] (coalesced stores in the kernel): the last pass transposes into the weight's [Cout][Cin][3][3]
    check(lib.vs_conv3_wgrad_band_finish(src.data_ptr(), n, _ptr(into), dw.data_ptr(), w_shape[0], w_shape[1], stream_ptr()), 'vs_conv3_wgrad_band_finish')


def conv3_wgrad_band_pieces(pairs, w_shape, into=None):
    """Weight gradient of Conv2d k3 s1 p1 over the batch formed by the (dz, x) pairs -- equal shapes, each contiguous -- WITHOUT concatenating
    them (`vs_conv3_wgrad_band_pieces`: the piece pointers travel as kernel arguments).  Returns None when the row-band kernel does not
    serve the concatenated shape or there are more than 64 pieces (the caller concatenates)."""
    import ctypes
    dz0, x0 = pairs[0]
    n = len(pairs)
    mp, Cin, H, W = x0.shape
    Cout = w_shape[0]
    if (W == 8 and mp % 4 != 0) or (W == 4 and mp % 16 != 0) or n > 64 or any(p[0].shape != dz0.shape or p[1].shape != x0.shape or not p[0].is_contiguous() or not p[1].is_contiguous()
                     or p[0].dtype != x0.dtype or p[1].dtype != x0.dtype for p in pairs):
        return None
    if not conv3_wgrad_band_supported(torch.empty((n * mp, Cin, H, W), dtype=x0.dtype, device='meta'), Cout, dtype_code_of=dtype_code(x0)):
        return None
    lib = _lib.load_library()
    B = n * mp
    nslabs = lib.vs_conv3_wgrad_band_slabs(B, Cin, H, W, Cout)
    slabs = torch.empty((nslabs,) + tuple(w_shape), dtype=torch.float32, device=x0.device)
    dw = into if into is not None else torch.empty(tuple(w_shape), dtype=torch.float32, device=x0.device)
    xs = (ctypes.c_void_p * n)(*[p[1].data_ptr() for p in pairs])
    dzs = (ctypes.c_void_p * n)(*[p[0].data_ptr() for p in pairs])
    e0 = _pb()
    check(lib.vs_conv3_wgrad_band_pieces(dtype_code(x0), n, xs, dzs, mp, slabs.data_ptr(), Cin, H, W, Cout, stream_ptr()), 'vs_conv3_wgrad_band_pieces')
    _slab_reduce(slabs, nslabs, w_shape, into, dw)
    _pe(e0, 'vs_conv3_wgrad_band<%s>' % _DT[dtype_code(x0)], flops=2.0 * B * Cout * H * W * Cin * 9,
        nbytes=float(n * (dz0.numel() + x0.numel()) * x0.element_size() + slabs.numel() * 8))
    return dw


def conv_wgrad(dy, x, w_shape, stride, pad, transposed, into=None, out=None):
    """Weight gradient of a (transposed) convolution; `into`: an fp32 tensor of the weight's shape that the result is ADDED to
    (in the GEMM / split-K epilogue) instead of a fresh tensor; `out`: an fp32 tensor that receives it."""
    require_cuda(dy, x)
    assert dy.is_contiguous() and x.is_contiguous() and dy.dtype == x.dtype
    B, Cin, H, W = x.shape
    k = w_shape[2]
    Cout = w_shape[1] if transposed else w_shape[0]
    OH, OW = dy.shape[2], dy.shape[3]
    if into is not None:
        assert into.dtype == torch.float32 and into.is_contiguous() and tuple(into.shape) == tuple(w_shape)
    if out is not None:
        assert into is None and out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == tuple(w_shape)
    if (x.dtype == torch.float32 and not transposed and tuple(w_shape[2:]) == (3, 3) and stride == 1 and pad == 1 and fp32_split_enabled()):
        xs = _split16(x)
        if conv3_wgrad_band_supported(xs[0], Cout):
            # dW is bilinear in (dy, x): the same six products, accumulated by the band kernel's own finishing pass (`into`)
            ds = _split16(dy)
            dw = None
            for i, j in _SPLIT_TERMS:
                dw = conv_wgrad(ds[j], xs[i], w_shape, stride, pad, transposed, into=into if dw is None else dw, out=out if dw is None else None)
            return dw
    if x.dtype == torch.float32 and tuple(w_shape[2:]) == (4, 4) and stride == 2 and pad == 1 and fp32_split_enabled():
        # k4 s2 p1: the LARGE map (x of a Conv2d, dy of a ConvTranspose2d) on its parity planes, the small one as it is
        dw = _k4s2_split_wgrad(x, dy, w_shape, into, out) if transposed else _k4s2_split_wgrad(dy, x, w_shape, into, out)
        if dw is not None:
            return dw
    if x.dtype == torch.float32 and w_shape[2] == w_shape[3] and fp32_split_enabled():
        thin = _thin_plan('wgrad', BF16, B, Cin, H, W, Cout, OH, OW, k, stride, pad, transposed)
        if thin is not None:
            big, small = (dy, x) if thin[0] == 'wgrad_big_dy' else (x, dy)
            bs, ss, dw = _split16(big), _split16(small), None
            for i, j in _SPLIT_TERMS:
                dw = conv_thin_wgrad(bs[i], ss[j], w_shape, k, stride, *thin[5:], into=into if dw is None else dw, out=out if dw is None else None)
            return dw
    if _convt_1x1(x.shape, k, stride, pad, transposed):
        # dW[ci][j] (+)= sum_b x[b][ci] dy[b][j]
        n = Cout * k * w_shape[3]
        dw = into if into is not None else (out if out is not None else torch.empty(tuple(w_shape), dtype=torch.float32, device=x.device))
        gemm(x.view(B, Cin), LAYOUT_S, dy.view(B, n), LAYOUT_S, Cin, n, B, out=dw.view(Cin, n), accumulate=into is not None)
        return dw
    if _conv_full(x.shape, k, stride, pad, transposed) and w_shape[3] == k:
        # dW[co][j] (+)= sum_b dy[b][co] x[b][j]
        n = Cin * k * k
        dw = into if into is not None else (out if out is not None else torch.empty(tuple(w_shape), dtype=torch.float32, device=x.device))
        gemm(dy.view(B, Cout), LAYOUT_S, x.view(B, n), LAYOUT_S, Cout, n, B, out=dw.view(Cout, n), accumulate=into is not None)
        return dw
    thin = _thin_plan('wgrad', dtype_code(x), B, Cin, H, W, Cout, OH, OW, k, stride, pad, transposed) if w_shape[2] == w_shape[3] else None
    if thin is not None:
        big, small = (dy, x) if thin[0] == 'wgrad_big_dy' else (x, dy)
        return conv_thin_wgrad(big, small, w_shape, k, stride, *thin[5:], into=into, out=out)
    dw = into if into is not None else (out if out is not None else torch.empty(tuple(w_shape), dtype=torch.float32, device=x.device))
    lib = _lib.load_library()
    if not transposed and k == 3 and stride == 1 and pad == 1 and conv3_wgrad_band_supported(x, Cout):
        # 3x3 on maps of width 16 / 32 / 64: row bands in LDS, no column matrix; partial gradients in slabs, one launch adds them
        nslabs = lib.vs_conv3_wgrad_band_slabs(B, Cin, H, W, Cout)
        slabs = torch.empty((nslabs,) + tuple(w_shape), dtype=torch.float32, device=x.device)
        e0 = _pb()
        check(lib.vs_conv3_wgrad_band(dtype_code(x), x.data_ptr(), dy.data_ptr(), slabs.data_ptr(), B, Cin, H, W, Cout, stream_ptr()), 'vs_conv3_wgrad_band')
        _slab_reduce(slabs, nslabs, w_shape, into, dw)
        _pe(e0, 'vs_conv3_wgrad_band<%s>' % _DT[dtype_code(x)], flops=2.0 * B * Cout * OH * OW * Cin * 9,
            nbytes=float(dy.numel() * dy.element_size() + x.numel() * x.element_size() + slabs.numel() * 8))
        return dw
    pix_h, pix_w = (H, W) if transposed else (OH, OW)
    ws = _conv_workspace(dtype_code(x), B, Cin, H, W, Cout, k, stride, pad, x.device)
    if ws is None:
        ws_bytes = lib.vs_conv_wgrad_workspace_bytes(B, Cin, pix_h, pix_w, Cout, k, k)
        ws = _workspace(ws_bytes, x.device) if ws_bytes else None
    fn = lib.vs_conv_transpose2d_wgrad_acc if transposed else lib.vs_conv2d_wgrad_acc
    e0 = _pb()
    check(fn(dtype_code(x), dy.data_ptr(), x.data_ptr(), dw.data_ptr(), B, Cin, H, W, Cout, k, k, stride, pad, _ptr(ws),
             ws.numel() if ws is not None else 0, 1 if into is not None else 0, stream_ptr()), 'vs_conv_wgrad')
    _pe(e0, 'vs_conv%s_cols:wgrad<%s>' % ('T' if transposed else '', _DT[dtype_code(x)]),
        flops=2.0 * B * Cout * OH * OW * Cin * k * k if not transposed else 2.0 * B * Cin * H * W * Cout * k * k,
        nbytes=float(dy.numel() * dy.element_size() + x.numel() * x.element_size() + dw.numel() * 4))
    return dw


# ---- 4x4 stride-2 pad-1 convolutions on the parity planes of their input (csrc/vs_conv_k4s2.hip) --------------------------------
def conv_k4s2_supported(big, M):
    """Whether the gather-type operations of a k4 s2 p1 (transposed) convolution on the LARGE map `big` [B, C, H, W] -- Conv2d forward /
    weight gradient (big = the input, M = Cout), ConvTranspose2d input / weight gradient (big = the output gradient, M = Cin) -- run on
    the parity planes of `big` with the row-band kernels (no column matrix).  VS_CONV_K4S2=0: never."""
    import os
    if os.environ.get('VS_CONV_K4S2') == '0' or big.dtype == torch.float32 or big.dim() != 4:
        return False
    B, C, H, W = big.shape
    lib = _lib.load_library()
    code = dtype_code(big)
    if C % 16 != 0 or not lib.vs_space_to_depth2_supported(code, B, C, H, W):       # (the plane pre-packs hold whole 64-channel phases)
        return False
    return bool(lib.vs_conv3_band_supported(code, B, 4 * C, H // 2, W // 2, M)) and bool(lib.vs_conv3_wgrad_band_supported(code, B, 4 * C, H // 2, W // 2, M))


def conv_k4s2_gather_supported(big, M):
    """`conv_k4s2_supported` for the GATHER alone (Conv2d forward / ConvTranspose2d input gradient on the parity planes): 8 x 8 maps have 4 x 4
    planes, which the row-band forward kernel serves (sixteen maps per workgroup) but the row-band weight gradient does not."""
    import os
    if os.environ.get('VS_CONV_K4S2') == '0' or big.dtype == torch.float32 or big.dim() != 4:
        return False
    B, C, H, W = big.shape
    lib = _lib.load_library()
    code = dtype_code(big)
    return (C % 16 == 0 and bool(lib.vs_space_to_depth2_supported(code, B, C, H, W))
            and bool(lib.vs_conv3_band_supported(code, B, 4 * C, H // 2, W // 2, M)))


def space_to_depth2(x):
    """x [B, C, H, W] (16-bit) -> its four parity planes [B, 4 C, H/2, W/2], channel = (row parity * 2 + column parity) * C + c."""
    require_cuda(x)
    assert x.is_contiguous()
    B, C, H, W = x.shape
    y = torch.empty((B, 4 * C, H // 2, W // 2), dtype=x.dtype, device=x.device)
    e0 = _pb()
    check(_lib.load_library().vs_space_to_depth2(dtype_code(x), x.data_ptr(), y.data_ptr(), B, C, H, W, stream_ptr()), 'vs_space_to_depth2')
    _pe(e0, 'vs_space_to_depth2', nbytes=float(2 * x.numel() * x.element_size()))
    return y


def conv_k4s2_pack_weight(w_master, dtype, out=None):
    """fp32 [M, K, 4, 4] -> the row-band kernels' pre-pack over the 4 K plane channels (vs_conv_k4s2_pack_weight)."""
    require_cuda(w_master)
    assert w_master.dtype == torch.float32 and w_master.is_contiguous() and tuple(w_master.shape[2:]) == (4, 4)
    M, K = w_master.shape[0], w_master.shape[1]
    lib = _lib.load_library()
    if out is None:
        out = torch.empty((lib.vs_conv_k4s2_packed_elems(K, M),), dtype=dtype, device=w_master.device)
    check(lib.vs_conv_k4s2_pack_weight(code_of(dtype), w_master.data_ptr(), K, M, out.data_ptr(), stream_ptr()), 'vs_conv_k4s2_pack_weight')
    return out


def conv_k4s2_gather(planes, w_packed, bias, M, out_dtype, role='fwd', bn_sums=None, groups=1):
    """out [B, M, h, w] = k4 s2 p1 gather of the large map whose parity planes are `planes` [B, 4 K, h, w] with the packed weight
    (Conv2d forward: role 'fwd'; ConvTranspose2d input gradient: role 'dgrad')."""
    require_cuda(planes, w_packed, bias)
    assert planes.is_contiguous() and planes.dtype == w_packed.dtype
    B, C4, H, W = planes.shape
    y = torch.empty((B, M, H, W), dtype=out_dtype, device=planes.device)
    e0 = _pb()
    check(_lib.load_library().vs_conv_k4s2_band_bn(dtype_code(planes), planes.data_ptr(), w_packed.data_ptr(), _ptr(bias), y.data_ptr(), dtype_code(y), B,
                                                   C4 // 4, H, W, M, _ptr(bn_sums), groups, stream_ptr()), 'vs_conv_k4s2_band')
    # algorithmic work of the 4x4 window: 16 taps x K channels (the zero taps of the 3x3 form are not counted)
    _pe(e0, 'vs_conv_k4s2:%s<%s>' % (role, _DT[dtype_code(planes)]), flops=2.0 * B * H * W * M * (C4 // 4) * 16,
        nbytes=float(planes.numel() * planes.element_size() + M * (C4 // 4) * 16 * 2 + y.numel() * y.element_size()))
    return y


def conv_k4s2_wgrad(small, planes, w_shape, into=None, out=None):
    """dW [M, K, 4, 4] (fp32) of a k4 s2 p1 (transposed) convolution from the SMALL map `small` [B, M, h, w] and the parity planes of the
    large one [B, 4 K, h, w]; `into`: ADDED to this pending gradient, `out`: written there."""
    require_cuda(small, planes)
    assert small.is_contiguous() and planes.is_contiguous() and small.dtype == planes.dtype
    B, M, H, W = small.shape
    K = planes.shape[1] // 4
    assert tuple(w_shape) == (M, K, 4, 4), (tuple(w_shape), M, K)
    lib = _lib.load_library()
    dw = into if into is not None else (out if out is not None else torch.empty(tuple(w_shape), dtype=torch.float32, device=small.device))
    assert dw.dtype == torch.float32 and dw.is_contiguous() and tuple(dw.shape) == tuple(w_shape)
    nslabs = lib.vs_conv_k4s2_wgrad_band_slabs(B, K, H, W, M)
    n3 = M * 4 * K * 9
    slabs = torch.empty((nslabs, n3), dtype=torch.float32, device=small.device)
    e0 = _pb()
    check(lib.vs_conv_k4s2_wgrad_band(dtype_code(planes), planes.data_ptr(), small.data_ptr(), slabs.data_ptr(), B, K, H, W, M, stream_ptr()),
          'vs_conv_k4s2_wgrad_band')
    src, n = slabs, nslabs
    if nslabs > 24:
        src = torch.empty((16, n3), dtype=torch.float32, device=small.device)
        check(lib.vs_slab_sum_grouped(slabs.data_ptr(), nslabs, 16, src.data_ptr(), n3, stream_ptr()), 'vs_slab_sum_grouped')
        n = -(-nslabs // (-(-nslabs // 16)))
    check(lib.vs_conv_k4s2_wgrad_finish(src.data_ptr(), n, _ptr(into), dw.data_ptr(), M, K, stream_ptr()), 'vs_conv_k4s2_wgrad_finish')
    _pe(e0, 'vs_conv_k4s2:wgrad<%s>' % _DT[dtype_code(planes)], flops=2.0 * B * H * W * M * K * 16,
        nbytes=float(small.numel() * small.element_size() + planes.numel() * planes.element_size() + slabs.numel() * 8))
    return dw


# ---- ConvTranspose2d k4 s2 p1 forward as tap GEMM + col2im epilogue (csrc/vs_conv_tap.hip) -------------------------------------
def convt_tap_supported(x, Cout, groups):
    import os
    if os.environ.get('VS_CONV_TAP') == '0' or x.dtype == torch.float32:
        return False
    B, Cin, H, W = x.shape
    return bool(_lib.load_library().vs_convt_tap_supported(dtype_code(x), B, Cin, H, W, Cout, groups))


def convt_tap_pack_weight(w_master, dtype, out=None):
    """fp32 ConvTranspose2d weight [Cin, Cout, 4, 4] -> [ceil(Cout/16), 16 taps x 16 channels, Cin] in `dtype`."""
    require_cuda(w_master)
    assert w_master.dtype == torch.float32 and w_master.is_contiguous() and tuple(w_master.shape[2:]) == (4, 4)
    Cin, Cout = w_master.shape[0], w_master.shape[1]
    lib = _lib.load_library()
    if out is None:
        out = torch.empty((lib.vs_convt_tap_packed_elems(Cin, Cout),), dtype=dtype, device=w_master.device)
    check(lib.vs_convt_tap_pack_weight(code_of(dtype), w_master.data_ptr(), Cin, Cout, out.data_ptr(), stream_ptr()), 'vs_convt_tap_pack_weight')
    return out


def convt_tap_fwd(x, w_tap, bias, Cout, groups=1, want_sums=True, role='fwd'):
    """-> (y [B, Cout, 2H, 2W] in x's dtype, fp64 sums [groups, Cout, 2] of the stored outputs or None)."""
    require_cuda(x, w_tap, bias)
    assert x.is_contiguous() and x.dtype == w_tap.dtype
    B, Cin, H, W = x.shape
    y = torch.empty((B, Cout, 2 * H, 2 * W), dtype=x.dtype, device=x.device)
    sums = torch.empty((groups, Cout, 2), dtype=torch.float64, device=x.device) if want_sums else None
    e0 = _pb()
    check(_lib.load_library().vs_convt_k4s2_tap_fwd(dtype_code(x), x.data_ptr(), w_tap.data_ptr(), _ptr(bias), y.data_ptr(), _ptr(sums), B, Cin, H, W,
                                                    Cout, groups, stream_ptr()), 'vs_convt_k4s2_tap_fwd')
    _pe(e0, 'vs_convT_tap:%s<%s>' % (role, _DT[dtype_code(x)]), flops=2.0 * B * Cin * H * W * Cout * 16,
        nbytes=float(x.numel() * x.element_size() + w_tap.numel() * 2 + y.numel() * y.element_size()))
    return y, sums


def conv_k3_tap_supported(x, Cout, groups, force=False):
    """Whether Conv2d k3 s1 p1 on `x` takes the tap kernel.  Measured (tools/conv_bench.py): with 9 taps x 28 channels per tile the
    epilogue (shift-sum of nine maps) outweighs the saved column matrix unless the contraction is long -- 512 -> 512 at 16x16
    (the SST integrator): 45 vs 65 us; 128 -> 128 (K = 4 tiles): 489 vs 371 us -- so it is taken for Cin >= 384, Cout >= 256
    (VS_CONV_TAP=2 or force=True: whenever the geometry is supported; VS_CONV_TAP=0: never)."""
    import os
    mode = os.environ.get('VS_CONV_TAP')
    if mode == '0' or x.dtype == torch.float32:
        return False
    B, Cin, H, W = x.shape
    if not (force or mode == '2') and (Cin < 384 or Cout < 256):
        return False
    return bool(_lib.load_library().vs_conv_k3_tap_supported(dtype_code(x), B, Cin, H, W, Cout, groups))


def conv3_img16_supported(x, Cout):
    """Whether Conv2d k3 s1 p1 on `x` takes the few-images kernel (`conv3_img16`): 16x16 maps, 16-bit, Cin a multiple of 64 and a batch
    small enough that the per-image tiling pays (the SST integrator: 8 maps).  VS_CONV_IMG=0: never."""
    import os
    if os.environ.get('VS_CONV_IMG') == '0' or x.dtype == torch.float32 or x.dim() != 4:
        return False
    B, Cin, H, W = x.shape
    if B > 32:
        return False
    return bool(_lib.load_library().vs_conv3_img16_supported(dtype_code(x), B, Cin, H, W, Cout))


def conv3_img16_pack_weight(w_master, dtype, flip, out=None):
    """fp32 Conv2d weight [Cout, Cin, 3, 3] -> MFMA-fragment order for `conv3_img16` in `dtype`.  flip=False: forward (rows = Cout,
    contraction = Cin); flip=True: input gradient (rows = Cin, contraction = Cout, taps flipped)."""
    require_cuda(w_master)
    assert w_master.dtype == torch.float32 and w_master.is_contiguous() and tuple(w_master.shape[2:]) == (3, 3)
    Cout, Cin = w_master.shape[0], w_master.shape[1]
    M, K = (Cin, Cout) if flip else (Cout, Cin)
    lib = _lib.load_library()
    if out is None:
        out = torch.empty((lib.vs_conv3_img16_packed_elems(K, M),), dtype=dtype, device=w_master.device)
    check(lib.vs_conv3_img16_pack_weight(code_of(dtype), w_master.data_ptr(), K, M, int(bool(flip)), out.data_ptr(), stream_ptr()),
          'vs_conv3_img16_pack_weight')
    return out


def conv3_img16_pack_weights(jobs, dtype):
    """jobs = [(w_master fp32 [Cout, Cin, 3, 3] contiguous, flip, out buffer or None)] -> list of packed buffers, ONE launch per 96 jobs."""
    import ctypes
    lib = _lib.load_library()
    outs = []
    for w, flip, buf in jobs:
        Cout, Cin = w.shape[0], w.shape[1]
        M, K = (Cin, Cout) if flip else (Cout, Cin)
        if buf is None:
            buf = torch.empty((lib.vs_conv3_img16_packed_elems(K, M),), dtype=dtype, device=w.device)
        outs.append((w, int(bool(flip)), buf, M, K))
    for i in range(0, len(outs), 96):
        chunk = outs[i:i + 96]
        n = len(chunk)
        VP, I32 = ctypes.c_void_p * n, ctypes.c_int32 * n
        check(lib.vs_conv3_img16_pack_weights(code_of(dtype), n, VP(*[c[0].data_ptr() for c in chunk]), I32(*[c[4] for c in chunk]),
                                              I32(*[c[3] for c in chunk]), I32(*[c[1] for c in chunk]), VP(*[c[2].data_ptr() for c in chunk]),
                                              stream_ptr()), 'vs_conv3_img16_pack_weights')
    return [c[2] for c in outs]


def conv3_img16(x, w_packed, Cout, role='fwd'):
    """Conv2d k3 s1 p1 of x [B, Cin, 16, 16] -> fp32 split slabs [S, B, Cout, 16, 16] WITHOUT bias (S = vs_conv3_img16_splits); consumers:
    `bn_train_fwd_small_slabs`, `slab_sum`."""
    require_cuda(x, w_packed)
    assert x.is_contiguous() and x.dtype == w_packed.dtype
    B, Cin, H, W = x.shape
    lib = _lib.load_library()
    S = lib.vs_conv3_img16_splits(B, Cin, Cout)
    slabs = torch.empty((S, B, Cout, H, W), dtype=torch.float32, device=x.device)
    e0 = _pb()
    check(lib.vs_conv3_img16(dtype_code(x), x.data_ptr(), w_packed.data_ptr(), slabs.data_ptr(), B, Cin, Cout, stream_ptr()), 'vs_conv3_img16')
    _pe(e0, 'vs_conv3_img16:%s<%s>' % (role, _DT[dtype_code(x)]), flops=2.0 * B * Cin * H * W * Cout * 9,
        nbytes=float(x.numel() * x.element_size() + w_packed.numel() * 2 + slabs.numel() * 4))
    return slabs


def conv3_band_supported(x, Cout):
    """Whether Conv2d k3 s1 p1 on `x` takes the row-band kernel (`conv3_band`: many maps of width 4 / 8 / 16 / 32 / 64, any channel count --
    a last 64-channel phase is filled with zeros --, no column matrix).  VS_CONV_BAND=0: never.  Below 16 channels on either side the
    thin-channel kernels (one VALU pass, no 32 x 64-wide MFMA tiles of padding) keep the layer: VS_CONV_BAND_MIN_C."""
    import os
    if os.environ.get('VS_CONV_BAND') == '0' or x.dtype == torch.float32 or x.dim() != 4:
        return False
    B, Cin, H, W = x.shape
    min_c = int(os.environ.get('VS_CONV_BAND_MIN_C', '16'))
    if Cin < min_c or Cout < min_c:
        return False
    return bool(_lib.load_library().vs_conv3_band_supported(dtype_code(x), B, Cin, H, W, Cout))


# ---- BatchNorm sums taken in a convolution's epilogue: persistent fp64 buffers [groups, C, 2], zero between steps (vs_bn_stats_from_sums_fold
# resets what it has read, so no fill launch is needed per convolution) ----------------------------------------------------------------------
_BN_SUMS = {}


def bn_sums_buffer(key, groups, C, device):
    """The persistent sums buffer of one BatchNorm call site (`key`: e.g. the data pointer of its running mean), zero-filled when created."""
    k = (device.index, key, groups, C)
    buf = _BN_SUMS.get(k)
    if buf is None:
        buf = _BN_SUMS[k] = torch.zeros((groups, C, 2), dtype=torch.float64, device=device)
    return buf


def band_bn_mode():
    """How the row-band kernels leave the BatchNorm statistics of their output (VS_BAND_BN_SUMS): '0' (default: a statistics pass over the
    stored output), 'parts' (round 4: per-workgroup partial sums in a table + a fold launch, no atomics, reproducible), '1' (round 3: fp64
    atomics).  Both epilogue forms are MEASURED SLOWER than the pass they replace and stay opt-in: the pass reads the 16-bit tensor once at
    HBM rate (11-80 us per layer), while the epilogue reduction (32 lanes x 16 channels x 2 sums per wave through cross-lane moves, behind the
    x shift of the result) lengthens every convolution launch -- same-box A/B of the replayed steps: TaxiBJ 8.13 (pass) / 8.33 (table) /
    10.7 ms (atomics), SST 20.22 / 20.67 / 22.9, Moving-MNIST 7.07 / 7.09."""
    import os
    return os.environ.get('VS_BAND_BN_SUMS', '0')


def conv_band_bn_supported(B, Cin, H, W, Cout, groups, dtype):
    """Whether vs_conv3_band (Cin = 4 K plane channels for the k4 s2 family) leaves the BatchNorm sums of its output (band_bn_mode)."""
    if band_bn_mode() == '0' or dtype == torch.float32:
        return False
    return bool(_lib.load_library().vs_conv3_band_bn_supported(code_of(dtype), B, Cin, H, W, Cout, groups))


def conv3_band_parts(x, w_packed, bias, Cout, out_dtype, role='fwd', k4=False):
    """vs_conv3_band / the k4 s2 gather on planes with the per-workgroup (sum, sum of squares) of the stored output written to a table:
    (y, parts [rows, Cout, 2] fp32); bn_stats_from_parts_fold finishes.  No atomics, no zero fill, reproducible."""
    require_cuda(x, w_packed, bias)
    assert x.is_contiguous() and x.dtype == w_packed.dtype
    B, Cin, H, W = x.shape
    lib = _lib.load_library()
    rows = lib.vs_conv3_band_bn_parts_rows(B, H, W)
    assert rows > 0
    y = torch.empty((B, Cout, H, W), dtype=out_dtype, device=x.device)
    parts = torch.empty((rows, Cout, 2), dtype=torch.float32, device=x.device)
    e0 = _pb()
    check(lib.vs_conv3_band_bn_parts(dtype_code(x), x.data_ptr(), w_packed.data_ptr(), _ptr(bias), y.data_ptr(), dtype_code(y), B, Cin, H, W, Cout,
                                     parts.data_ptr(), 1 if k4 else 0, stream_ptr()), 'vs_conv3_band_bn_parts')
    if k4:
        _pe(e0, 'vs_conv_k4s2:%s<%s>' % (role, _DT[dtype_code(x)]), flops=2.0 * B * H * W * Cout * (Cin // 4) * 16,
            nbytes=float(x.numel() * x.element_size() + Cout * (Cin // 4) * 16 * 2 + y.numel() * y.element_size()))
    else:
        _pe(e0, 'vs_conv3_band:%s<%s>' % (role, _DT[dtype_code(x)]), flops=2.0 * B * Cin * H * W * Cout * 9,
            nbytes=float(x.numel() * x.element_size() + w_packed.numel() * 2 + y.numel() * y.element_size()))
    return y, parts


def bn_stats_from_parts_fold(parts, groups, n_per_group, running_mean=None, running_var=None, momentum=0.1, eps=1e-5):
    """(mean, invstd) [groups, C] from the partial-sum table of conv3_band_parts (rows of a call group consecutive) in ONE launch; the running
    estimates are folded in call order."""
    require_cuda(parts)
    rows, C = parts.shape[0], parts.shape[1]
    assert rows % groups == 0
    stats = torch.empty((3, groups, C), dtype=torch.float32, device=parts.device)
    e0 = _pb()
    check(_lib.load_library().vs_bn_stats_from_parts_fold(parts.data_ptr(), rows // groups, groups, C, int(n_per_group), stats[0].data_ptr(),
                                                          stats[1].data_ptr(), stats[2].data_ptr(), _ptr(running_mean), _ptr(running_var),
                                                          float(momentum), float(eps), stream_ptr()), 'vs_bn_stats_from_parts_fold')
    _pe(e0, 'vs_bn_stats_from_sums', nbytes=float(parts.numel() * 4))
    return stats[0], stats[1]


def bn_stats_from_sums_fold(sums, n_per_group, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, reset=True):
    """(mean, invstd) [groups, C] from epilogue sums in ONE launch (running estimates folded in call order; the sums are zeroed for the next step)."""
    require_cuda(sums)
    groups, C = sums.shape[0], sums.shape[1]
    stats = torch.empty((2, groups, C), dtype=torch.float32, device=sums.device)
    e0 = _pb()
    check(_lib.load_library().vs_bn_stats_from_sums_fold(sums.data_ptr(), groups, C, int(n_per_group), stats[0].data_ptr(), stats[1].data_ptr(),
                                                         _ptr(running_mean), _ptr(running_var), float(momentum), float(eps), int(bool(reset)),
                                                         stream_ptr()), 'vs_bn_stats_from_sums_fold')
    _pe(e0, 'vs_bn_stats_from_sums', nbytes=float(sums.numel() * 8))
    return stats[0], stats[1]


def conv3_band(x, w_packed, bias, Cout, out_dtype, role='fwd', bn_sums=None, groups=1):
    """Conv2d k3 s1 p1 of x [B, Cin, H, W] (16-bit) with the `conv3_img16_pack_weight` pre-pack -> y [B, Cout, H, W] in out_dtype (+ bias).
    bn_sums [groups, Cout, 2] (fp64, see bn_sums_buffer): the sums of the stored outputs are added to it in the epilogue."""
    require_cuda(x, w_packed, bias)
    assert x.is_contiguous() and x.dtype == w_packed.dtype
    B, Cin, H, W = x.shape
    y = torch.empty((B, Cout, H, W), dtype=out_dtype, device=x.device)
    e0 = _pb()
    check(_lib.load_library().vs_conv3_band_bn(dtype_code(x), x.data_ptr(), w_packed.data_ptr(), _ptr(bias), y.data_ptr(), dtype_code(y), B, Cin, H, W,
                                               Cout, _ptr(bn_sums), groups, stream_ptr()), 'vs_conv3_band')
    _pe(e0, 'vs_conv3_band:%s<%s>' % (role, _DT[dtype_code(x)]), flops=2.0 * B * Cin * H * W * Cout * 9,
        nbytes=float(x.numel() * x.element_size() + w_packed.numel() * 2 + y.numel() * y.element_size()))
    return y


def slab_sum(slabs, bias, out_dtype, addend=None, addend2=None):
    """sum over the leading (split) axis of fp32 slabs [S, B, C, H, W] (+ bias[c]) (+ addend (+ addend2), fp32 [B, C, H, W]) -> [B, C, H, W] in
    out_dtype, one launch."""
    require_cuda(slabs)
    S, B, C = slabs.shape[0], slabs.shape[1], slabs.shape[2]
    HW = slabs.numel() // (S * B * C)
    if addend is None and addend2 is not None:
        addend, addend2 = addend2, None
    for a in (addend, addend2):
        assert a is None or (a.dtype == torch.float32 and a.is_contiguous() and a.numel() == B * C * HW)
    out = torch.empty(slabs.shape[1:], dtype=out_dtype, device=slabs.device)
    check(_lib.load_library().vs_slab_sum2(slabs.data_ptr(), S, _ptr(bias), _ptr(addend), _ptr(addend2), out.data_ptr(), dtype_code(out), B, C, HW,
                                           stream_ptr()), 'vs_slab_sum')
    return out


def conv_k3_tap_pack_weight(w_master, dtype, flip, out=None):
    """fp32 Conv2d weight [Cout, Cin, 3, 3] -> tap-GEMM rows [ceil(M/28), 9 taps x 28 channels (256 rows), K] in `dtype`.
    flip=False: forward (M = Cout, K = Cin).  flip=True: input gradient (M = Cin, K = Cout, taps flipped)."""
    require_cuda(w_master)
    assert w_master.dtype == torch.float32 and w_master.is_contiguous() and tuple(w_master.shape[2:]) == (3, 3)
    Cout, Cin = w_master.shape[0], w_master.shape[1]
    M, K = (Cin, Cout) if flip else (Cout, Cin)
    lib = _lib.load_library()
    if out is None:
        out = torch.empty((lib.vs_conv_k3_tap_packed_elems(K, M),), dtype=dtype, device=w_master.device)
    check(lib.vs_conv_k3_tap_pack_weight(code_of(dtype), w_master.data_ptr(), K, M, int(bool(flip)), out.data_ptr(), stream_ptr()),
          'vs_conv_k3_tap_pack_weight')
    return out


def conv_k3_tap_fwd(x, w_tap, bias, Cout, out_dtype, groups=1, want_sums=False, role='fwd'):
    """-> (y [B, Cout, H, W] in out_dtype, fp64 sums [groups, Cout, 2] of the stored outputs or None)."""
    require_cuda(x, w_tap, bias)
    assert x.is_contiguous() and x.dtype == w_tap.dtype
    B, Cin, H, W = x.shape
    y = torch.empty((B, Cout, H, W), dtype=out_dtype, device=x.device)
    sums = torch.empty((groups, Cout, 2), dtype=torch.float64, device=x.device) if want_sums else None
    e0 = _pb()
    check(_lib.load_library().vs_conv_k3s1_tap_fwd(dtype_code(x), x.data_ptr(), w_tap.data_ptr(), _ptr(bias), y.data_ptr(), dtype_code(y), _ptr(sums),
                                                   B, Cin, H, W, Cout, groups, stream_ptr()), 'vs_conv_k3s1_tap_fwd')
    _pe(e0, 'vs_conv3_tap:%s<%s>' % (role, _DT[dtype_code(x)]), flops=2.0 * B * Cin * H * W * Cout * 9,
        nbytes=float(x.numel() * x.element_size() + w_tap.numel() * 2 + y.numel() * y.element_size()))
    return y, sums


def bn_stats_from_sums(sums, n_per_group, running_mean=None, running_var=None, momentum=0.1, eps=1e-5):
    """(mean, invstd) [groups, C] from the fp64 (sum, sum of squares) a fused conv epilogue produced; folds the running statistics."""
    require_cuda(sums)
    groups, C = sums.shape[0], sums.shape[1]
    mean = torch.empty((groups, C), dtype=torch.float32, device=sums.device)
    invstd = torch.empty((groups, C), dtype=torch.float32, device=sums.device)
    scratch = torch.empty((groups, C), dtype=torch.float32, device=sums.device) if running_mean is not None else None
    check(_lib.load_library().vs_bn_stats_from_sums(sums.data_ptr(), groups, C, int(n_per_group), mean.data_ptr(), invstd.data_ptr(), _ptr(scratch),
                                                    _ptr(running_mean), _ptr(running_var), float(momentum), float(eps), stream_ptr()),
          'vs_bn_stats_from_sums')
    return mean, invstd


# ------------------------------------------------------------------------------------------------ norm / pool / upsample
def bn_stats(x, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, groups=1):
    """Per-(group, channel) batch statistics; returns (mean [groups, C], invstd [groups, C])."""
    require_cuda(x)
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    mean = torch.empty((groups, C), dtype=torch.float32, device=x.device)
    invstd = torch.empty((groups, C), dtype=torch.float32, device=x.device)
    scratch = torch.empty((groups, C), dtype=torch.float32, device=x.device) if running_mean is not None else None
    e0 = _pb()
    check(_lib.load_library().vs_bn_stats(x.data_ptr(), dtype_code(x), B, C, HW, groups, mean.data_ptr(), invstd.data_ptr(),
                                          _ptr(scratch), _ptr(running_mean), _ptr(running_var), float(momentum), float(eps),
                                          stream_ptr()), 'vs_bn_stats')
    _pe(e0, 'vs_bn_stats', nbytes=float(x.numel() * x.element_size()))
    return mean, invstd


def bn_stats_ub(x, eps=1e-5, groups=1):
    """bn_stats without the running update: (mean, invstd, unbiased variance) [groups, C]; `bn_act_fwd(..., running=...)` folds the last one."""
    require_cuda(x)
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    stats = torch.empty((3, groups, C), dtype=torch.float32, device=x.device)
    e0 = _pb()
    check(_lib.load_library().vs_bn_stats_ub(x.data_ptr(), dtype_code(x), B, C, HW, groups, stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(),
                                             float(eps), stream_ptr()), 'vs_bn_stats_ub')
    _pe(e0, 'vs_bn_stats', nbytes=float(x.numel() * x.element_size()))
    return stats[0], stats[1], stats[2]


def bn_small_groups_enabled():
    import os
    return os.environ.get('VS_BN_SMALL_GROUPS', '0') == '1'


def bn_small_supported(x, groups=1):
    B, C = x.shape[0], x.shape[1]
    if groups > 1:
        # several calls stacked along the batch axis in one launch (vs_bn_train_fwd_small_groups): measured on the TaxiBJ step and NOT the
        # default -- 8.95 ms with it, 8.93 without (the two-launch path's kernels overlap with their neighbours just as well)
        import os
        if os.environ.get('VS_BN_SMALL_GROUPS', '0') != '1' or B % groups != 0:
            return False
    return x.is_contiguous() and bool(_lib.load_library().vs_bn_train_fwd_small_supported(dtype_code(x), B // groups, C, x.numel() // (B * C)))


def bn_small_supported_shape(dtype, B, C, HW):
    """`bn_small_supported` for a tensor that does not exist yet ([B, C, ...] of HW elements per plane in `dtype`)."""
    return bool(_lib.load_library().vs_bn_train_fwd_small_supported(code_of(dtype), B, C, HW))


def bn_train_fwd_small(x, gamma, beta, act, out_dtype, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, groups=1):
    """Training-mode BatchNorm + activation of one call in ONE launch (small tensors).  Returns (y, mean [1, C], invstd [1, C]).
    groups > 1: that many calls stacked along the batch axis, statistics per call ([groups, C]), running estimates folded in call order."""
    require_cuda(x)
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    if groups > 1:
        stats = torch.empty((3, groups, C), dtype=torch.float32, device=x.device)
        e0 = _pb()
        check(_lib.load_library().vs_bn_train_fwd_small_groups(x.data_ptr(), dtype_code(x), y.data_ptr(), dtype_code(y), gamma.data_ptr(), beta.data_ptr(),
                                                               ACT[act], stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(),
                                                               _ptr(running_mean), _ptr(running_var), float(momentum), float(eps), B, C, HW, groups,
                                                               stream_ptr()), 'vs_bn_train_fwd_small_groups')
        _pe(e0, 'vs_bn_fwd_small', nbytes=float(x.numel() * (x.element_size() + y.element_size())))
        return y, stats[0], stats[1]
    mean = torch.empty((1, C), dtype=torch.float32, device=x.device)
    invstd = torch.empty((1, C), dtype=torch.float32, device=x.device)
    e0 = _pb()
    check(_lib.load_library().vs_bn_train_fwd_small(x.data_ptr(), dtype_code(x), y.data_ptr(), dtype_code(y), gamma.data_ptr(), beta.data_ptr(),
                                                    ACT[act], mean.data_ptr(), invstd.data_ptr(), _ptr(running_mean), _ptr(running_var),
                                                    float(momentum), float(eps), B, C, HW, stream_ptr()), 'vs_bn_train_fwd_small')
    _pe(e0, 'vs_bn_fwd_small', nbytes=float(x.numel() * (x.element_size() + y.element_size())))
    return y, mean, invstd


def bn_slab_supported(x, groups=1):
    """A (call group, channel) slab of `x` fits the registers of one workgroup (`bn_train_fwd_slab`: the tensor is read once)."""
    B, C = x.shape[0], x.shape[1]
    return (B % groups == 0 and x.is_contiguous() and x.data_ptr() % 16 == 0
            and bool(_lib.load_library().vs_bn_train_fwd_slab_supported(dtype_code(x), B // groups, C, x.numel() // (B * C))))


def bn_train_fwd_slab(x, gamma, beta, act, out_dtype, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, groups=1):
    """Training-mode BatchNorm + activation of `groups` calls stacked along the batch axis, every (call, channel) slab read ONCE (held in
    registers between the statistics and the apply phase).  Returns (y, mean [groups, C], invstd [groups, C])."""
    require_cuda(x)
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    stats = torch.empty((3, groups, C), dtype=torch.float32, device=x.device)
    e0 = _pb()
    check(_lib.load_library().vs_bn_train_fwd_slab(x.data_ptr(), dtype_code(x), y.data_ptr(), dtype_code(y), gamma.data_ptr(), beta.data_ptr(), ACT[act],
                                                   stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(), _ptr(running_mean), _ptr(running_var),
                                                   float(momentum), float(eps), B, C, HW, groups, stream_ptr()), 'vs_bn_train_fwd_slab')
    _pe(e0, 'vs_bn_fwd_slab', nbytes=float(x.numel() * (x.element_size() + y.element_size())))
    return y, stats[0], stats[1]


def bn_train_fwd_small_slabs(slabs, bias, z_dtype, gamma, beta, act, out_dtype, running_mean=None, running_var=None, momentum=0.1, eps=1e-5,
                             skip=None, want16=False):
    """`bn_train_fwd_small` on the split slabs [S, B, C, H, W] (fp32) of `conv3_img16`: the slab sum, the conv bias, the 16-bit rounding of
    the conv output z and the BatchNorm forward in one launch.  Returns (y, z, mean [1, C], invstd [1, C]); with `skip` (fp32, the block
    input of a residual block) additionally (skip + y in fp32, its `z_dtype` copy or None)."""
    require_cuda(slabs)
    S, B, C = slabs.shape[0], slabs.shape[1], slabs.shape[2]
    HW = slabs.numel() // (S * B * C)
    z = torch.empty(slabs.shape[1:], dtype=z_dtype, device=slabs.device)
    y = torch.empty(slabs.shape[1:], dtype=out_dtype, device=slabs.device)
    mean = torch.empty((1, C), dtype=torch.float32, device=slabs.device)
    invstd = torch.empty((1, C), dtype=torch.float32, device=slabs.device)
    xnew = xnew16 = None
    if skip is not None:
        assert skip.dtype == torch.float32 and skip.is_contiguous() and skip.numel() == y.numel()
        xnew = torch.empty(slabs.shape[1:], dtype=torch.float32, device=slabs.device)
        xnew16 = torch.empty(slabs.shape[1:], dtype=z_dtype, device=slabs.device) if want16 else None
    e0 = _pb()
    check(_lib.load_library().vs_bn_train_fwd_small_slabs(slabs.data_ptr(), S, _ptr(bias), z.data_ptr(), dtype_code(z), y.data_ptr(), dtype_code(y),
                                                          gamma.data_ptr(), beta.data_ptr(), ACT[act], mean.data_ptr(), invstd.data_ptr(),
                                                          _ptr(running_mean), _ptr(running_var), float(momentum), float(eps), _ptr(skip),
                                                          _ptr(xnew), _ptr(xnew16), B, C, HW, stream_ptr()), 'vs_bn_train_fwd_small_slabs')
    _pe(e0, 'vs_bn_fwd_small_slabs', nbytes=float(slabs.numel() * 4 + z.numel() * (z.element_size() + y.element_size())))
    if skip is not None:
        return y, z, mean, invstd, xnew, xnew16
    return y, z, mean, invstd


def bn_act_bwd_small_ex(z, mean, invstd, gamma, beta, act, dx_dtype, dy_a=None, dy_b=None, slabs=None, acc=None):
    """One-launch training-mode BatchNorm + activation backward on a small 16-bit z with the upstream gradient from split slabs
    [S, B, C, H, W] or from dy_a (+ dy_b, fp32).  `acc` = (dgamma, dbeta) vectors the parameter gradients are ADDED to (returns
    (dx, None, None)); otherwise fresh (dx, dgamma, dbeta)."""
    require_cuda(z)
    B, C = z.shape[0], z.shape[1]
    HW = z.numel() // (B * C)
    dx = torch.empty(z.shape, dtype=dx_dtype, device=z.device)
    if acc is None:
        dgamma = torch.empty((C,), dtype=torch.float32, device=z.device)
        dbeta = torch.empty((C,), dtype=torch.float32, device=z.device)
    else:
        dgamma, dbeta = acc
        assert dgamma.dtype == torch.float32 and dbeta.dtype == torch.float32 and dgamma.numel() == C and dbeta.numel() == C
    if slabs is not None:
        assert slabs.dtype == torch.float32 and slabs.is_contiguous() and slabs.numel() == slabs.shape[0] * z.numel()
    else:
        assert dy_a is not None and dy_a.is_contiguous() and dy_a.numel() == z.numel()
        assert dy_b is None or (dy_b.dtype == torch.float32 and dy_b.is_contiguous() and dy_b.numel() == z.numel())
    e0 = _pb()
    check(_lib.load_library().vs_bn_act_bwd_small_ex(_ptr(dy_a), dtype_code(dy_a) if dy_a is not None else F32, _ptr(dy_b), _ptr(slabs),
                                                     slabs.shape[0] if slabs is not None else 0, z.data_ptr(), dtype_code(z), mean.data_ptr(),
                                                     invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ACT[act], dgamma.data_ptr(),
                                                     dbeta.data_ptr(), int(acc is not None), dx.data_ptr(), dtype_code(dx), B, C, HW, stream_ptr()),
          'vs_bn_act_bwd_small_ex')
    _pe(e0, 'vs_bn_bwd_small_ex', nbytes=float(z.numel() * (z.element_size() + dx.element_size() + 4)))
    if acc is not None:
        return dx, None, None
    return dx, dgamma, dbeta


# ---- a ConvResBlock layer in one launch (csrc/vs_conv_img.hip: conv3_img16_bn_kernel) -------------------------------------------------
_IMGBN = {}          # device index -> {'ws': int32 workspace (word 0: epoch base, word 1: sticky error flag), 'idx': launches since the last advance}


def _imgbn_state(device):
    """The exchange workspace of the fused layers on `device` (one per device: the integrator's launches follow each other on one stream).
    Created OUTSIDE a stream capture (GraphedStep's warm-up steps reach it first): a workspace created while capturing would be zero-filled
    and re-initialised by every replay."""
    index = device.index if device.index is not None else torch.cuda.current_device()
    st = _IMGBN.get(index)
    if st is None:
        exchange_guard(device)
        nbytes = _lib.load_library().vs_conv3_img16_bn_workspace_bytes()
        ws = torch.zeros((nbytes // 4,), dtype=torch.int32, device=device)
        ws[0] = 65536                                    # epochs start above zero: a zero-filled granule never looks current
        st = _IMGBN[index] = {'ws': ws, 'idx': 0}
    return st


def _imgbn_next_call(st):
    st['idx'] += 1
    if st['idx'] >= 65535:                               # (a step has ~1000 such launches; callers outside a training step never advance)
        check(_lib.load_library().vs_exchange_epoch_advance(st['ws'].data_ptr(), stream_ptr()), 'vs_exchange_epoch_advance')
        st['idx'] = 1
    return st['idx']


def exchange_epoch_advance(device=None):
    """Start of a training step: the launch numbers of the fused layers restart at 1 under a new epoch base (one 1-thread launch, recordable:
    a replayed step then tags its exchanges with epochs no earlier replay used).  No-op while no fused layer has run on the device."""
    index = torch.cuda.current_device() if device is None or device.index is None else device.index
    st = _IMGBN.get(index)
    if st is None:
        return
    check(_lib.load_library().vs_exchange_epoch_advance(st['ws'].data_ptr(), stream_ptr()), 'vs_exchange_epoch_advance')
    st['idx'] = 0


def conv3_img16_bn_supported(B, Cin, Cout, dtype, act='leaky_relu', out_dtype=None, backward=False):
    """Whether conv3_img16_bn_fwd / _bwd serve a Conv2d k3 s1 p1 (Cin -> Cout) -> BatchNorm -> `act` layer on B maps of 16 x 16 with output type
    `out_dtype` (forward).  VS_IMG_BN_FUSED=0: never; VS_IMG_BN_SPLITS (default '1'): the input-channel split counts served -- with several
    splits the partial sums cross the chip inside the launch, which costs what the kernel boundary it replaces costs (measured)."""
    import os
    if os.environ.get('VS_IMG_BN_FUSED', '1') == '0' or dtype == torch.float32:
        return False
    lib = _lib.load_library()
    if not lib.vs_conv3_img16_bn_supported(code_of(dtype), B, Cin, Cout):
        return False
    if not lib.vs_conv3_img16_bn_form_supported(int(backward), ACT[act], code_of(out_dtype if out_dtype is not None else dtype), code_of(dtype)):
        return False
    allowed = os.environ.get('VS_IMG_BN_SPLITS', '1').split(',')
    return str(lib.vs_conv3_img16_splits(B, Cin, Cout)) in allowed


def conv3_img16_bn_fwd(x, w_packed, bias, gamma, beta, act, out_dtype, Cout, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, skip=None,
                       want16=False):
    """act(BatchNorm_train(conv3x3(x) + bias)) of ONE call in ONE launch: (y, z, mean [1, C], invstd [1, C]) like conv3_img16 +
    bn_train_fwd_small_slabs; with `skip` additionally (skip + y in fp32, its 16-bit copy or None)."""
    require_cuda(x, w_packed)
    assert x.is_contiguous() and x.dtype == w_packed.dtype and tuple(x.shape[2:]) == (16, 16)
    B, Cin = x.shape[0], x.shape[1]
    st = _imgbn_state(x.device)
    z = torch.empty((B, Cout, 16, 16), dtype=x.dtype, device=x.device)
    y = torch.empty((B, Cout, 16, 16), dtype=out_dtype, device=x.device)
    mean = torch.empty((1, Cout), dtype=torch.float32, device=x.device)
    invstd = torch.empty((1, Cout), dtype=torch.float32, device=x.device)
    xnew = xnew16 = None
    if skip is not None:
        assert skip.dtype == torch.float32 and skip.is_contiguous() and skip.numel() == y.numel()
        xnew = torch.empty((B, Cout, 16, 16), dtype=torch.float32, device=x.device)
        xnew16 = torch.empty((B, Cout, 16, 16), dtype=x.dtype, device=x.device) if want16 else None
    e0 = _pb()
    check(_lib.load_library().vs_conv3_img16_bn_fwd(dtype_code(x), x.data_ptr(), w_packed.data_ptr(), st['ws'].data_ptr(), _imgbn_next_call(st), _ptr(bias),
                                                    gamma.data_ptr(), beta.data_ptr(), ACT[act], _ptr(running_mean), _ptr(running_var), float(momentum),
                                                    float(eps), z.data_ptr(), y.data_ptr(), dtype_code(y), mean.data_ptr(), invstd.data_ptr(), _ptr(skip),
                                                    _ptr(xnew), _ptr(xnew16), B, Cin, Cout, stream_ptr()), 'vs_conv3_img16_bn_fwd')
    _pe(e0, 'vs_conv3_img16_bn:fwd<%s>' % _DT[dtype_code(x)], flops=2.0 * B * 256 * Cout * Cin * 9,
        nbytes=float(x.numel() * x.element_size() + w_packed.numel() * 2 + z.numel() * (z.element_size() + y.element_size())))
    if skip is not None:
        return y, z, mean, invstd, xnew, xnew16
    return y, z, mean, invstd


def conv3_img16_bn_bwd(dz_next, w_packed_flipped, Cout, z, mean, invstd, gamma, beta, act, acc=None):
    """The backward layer in ONE launch: dz = BatchNorm + activation backward (from the stored z, mean, invstd) of dy = the input gradient of the
    FOLLOWING layer's convolution of dz_next (`w_packed_flipped`: that layer's weight, flip = 1).  `acc` = (dgamma, dbeta) the parameter
    gradients are ADDED to (returns (dz, None, None)); otherwise fresh (dz, dgamma, dbeta)."""
    require_cuda(dz_next, w_packed_flipped, z)
    assert dz_next.is_contiguous() and z.is_contiguous() and dz_next.dtype == z.dtype == w_packed_flipped.dtype
    B, Cin = dz_next.shape[0], dz_next.shape[1]
    assert z.shape[0] == B and z.shape[1] == Cout
    st = _imgbn_state(z.device)
    dz = torch.empty(z.shape, dtype=z.dtype, device=z.device)
    if acc is None:
        dgamma = torch.empty((Cout,), dtype=torch.float32, device=z.device)
        dbeta = torch.empty((Cout,), dtype=torch.float32, device=z.device)
    else:
        dgamma, dbeta = acc
        assert dgamma.dtype == torch.float32 and dbeta.dtype == torch.float32 and dgamma.numel() == Cout and dbeta.numel() == Cout
    e0 = _pb()
    check(_lib.load_library().vs_conv3_img16_bn_bwd(dtype_code(z), dz_next.data_ptr(), w_packed_flipped.data_ptr(), st['ws'].data_ptr(), _imgbn_next_call(st),
                                                    z.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ACT[act],
                                                    dz.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), int(acc is not None), B, Cin, Cout, stream_ptr()),
          'vs_conv3_img16_bn_bwd')
    _pe(e0, 'vs_conv3_img16_bn:bwd<%s>' % _DT[dtype_code(z)], flops=2.0 * B * 256 * Cout * Cin * 9,
        nbytes=float(dz_next.numel() * 2 + w_packed_flipped.numel() * 2 + 2 * z.numel() * z.element_size()))
    if acc is not None:
        return dz, None, None
    return dz, dgamma, dbeta


def bn_act_fwd(x, mean, invstd, gamma, beta, act, out_dtype, groups=1, running=None):
    """y = act(gamma * (x - mean) * invstd + beta) per (call group, channel).  running = (ubvar [groups, C], running_mean, running_var, momentum):
    the running estimates are folded in call order by the same launch (bn_stats_ub produced ubvar)."""
    require_cuda(x)
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    e0 = _pb()
    if running is not None:
        ub, rm, rv, momentum = running
        check(_lib.load_library().vs_bn_act_fwd_running(x.data_ptr(), dtype_code(x), y.data_ptr(), dtype_code(y), mean.data_ptr(), invstd.data_ptr(),
                                                        gamma.data_ptr(), beta.data_ptr(), ACT[act], B, C, HW, groups, ub.data_ptr(), rm.data_ptr(),
                                                        rv.data_ptr(), float(momentum), stream_ptr()), 'vs_bn_act_fwd_running')
    else:
        check(_lib.load_library().vs_bn_act_fwd(x.data_ptr(), dtype_code(x), y.data_ptr(), dtype_code(y), mean.data_ptr(),
                                                invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ACT[act], B, C, HW, groups,
                                                stream_ptr()), 'vs_bn_act_fwd')
    _pe(e0, 'vs_bn_act_fwd', nbytes=float(x.numel() * (x.element_size() + y.element_size())))
    return y


def bn_act_bwd(dy, x, mean, invstd, gamma, beta, act, training, out_dtype, groups=1):
    """Returns (dx, dgamma [C], dbeta [C]) -- the per-call-group sums are added up by the kernel itself (vs_bn_act_bwd_gsum)."""
    require_cuda(dy, x)
    dy = dy.contiguous()
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    per_group = torch.empty((2, groups, C), dtype=torch.float32, device=x.device)
    dgamma, dbeta = per_group[0], per_group[1]
    sums = torch.empty((2, C), dtype=torch.float32, device=x.device) if groups > 1 else None
    dx = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    e0 = _pb()
    check(_lib.load_library().vs_bn_act_bwd_gsum(dy.data_ptr(), dtype_code(dy), x.data_ptr(), dtype_code(x), mean.data_ptr(),
                                                 invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ACT[act], int(bool(training)),
                                                 groups, dgamma.data_ptr(), dbeta.data_ptr(), dx.data_ptr(), dtype_code(dx), B, C, HW,
                                                 _ptr(sums[0]) if sums is not None else 0, _ptr(sums[1]) if sums is not None else 0,
                                                 stream_ptr()), 'vs_bn_act_bwd')
    _pe(e0, 'vs_bn_act_bwd', nbytes=float(x.numel() * (2 * x.element_size() + 2 * dy.element_size() + dx.element_size())))
    if groups == 1:
        return dx, dgamma[0], dbeta[0]
    return dx, sums[0], sums[1]


def chan_sum(x):
    require_cuda(x)
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    out = torch.empty((C,), dtype=torch.float32, device=x.device)
    lib = _lib.load_library()
    ws = _workspace(lib.vs_chan_sum_workspace_bytes(B, C, HW), x.device)
    e0 = _pb()
    check(lib.vs_chan_sum_ws(x.data_ptr(), dtype_code(x), B, C, HW, ws.data_ptr(), ws.numel(), out.data_ptr(), stream_ptr()), 'vs_chan_sum_ws')
    _pe(e0, 'vs_chan_sum<%s>' % _DT[dtype_code(x)], nbytes=float(x.numel() * x.element_size()))
    return out


def maxpool2_fwd(x):
    require_cuda(x)
    B, C, H, W = x.shape
    y = torch.empty((B, C, H // 2, W // 2), dtype=x.dtype, device=x.device)
    check(_lib.load_library().vs_maxpool2_fwd(x.data_ptr(), dtype_code(x), y.data_ptr(), dtype_code(y), B * C, H, W, stream_ptr()),
          'vs_maxpool2_fwd')
    return y


def maxpool2_bwd(x, dy):
    require_cuda(x, dy)
    B, C, H, W = x.shape
    dy = dy.contiguous()
    dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    check(_lib.load_library().vs_maxpool2_bwd(x.data_ptr(), dtype_code(x), dy.data_ptr(), dtype_code(dy), dx.data_ptr(),
                                              dtype_code(dx), B * C, H, W, stream_ptr()), 'vs_maxpool2_bwd')
    return dx


def maxpool3s2_fwd(x):
    """nn.MaxPool2d(3, 2, 1) on [B, C, H, W]."""
    require_cuda(x)
    B, C, H, W = x.shape
    y = torch.empty((B, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=x.dtype, device=x.device)
    check(_lib.load_library().vs_maxpool3s2_fwd(x.data_ptr(), dtype_code(x), y.data_ptr(), dtype_code(y), B * C, H, W, stream_ptr()),
          'vs_maxpool3s2_fwd')
    return y


def maxpool3s2_bwd(x, dy):
    require_cuda(x, dy)
    B, C, H, W = x.shape
    dy = dy.contiguous()
    dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    check(_lib.load_library().vs_maxpool3s2_bwd(x.data_ptr(), dtype_code(x), dy.data_ptr(), dtype_code(dy), dx.data_ptr(),
                                                dtype_code(dx), B * C, H, W, stream_ptr()), 'vs_maxpool3s2_bwd')
    return dx


def upsample2_fwd(x):
    require_cuda(x)
    B, C, H, W = x.shape
    y = torch.empty((B, C, 2 * H, 2 * W), dtype=x.dtype, device=x.device)
    check(_lib.load_library().vs_upsample2_fwd(x.data_ptr(), dtype_code(x), y.data_ptr(), dtype_code(y), B * C, H, W, stream_ptr()),
          'vs_upsample2_fwd')
    return y


def upsample2_bwd(dy, out_dtype):
    require_cuda(dy)
    dy = dy.contiguous()
    B, C, H2, W2 = dy.shape
    dx = torch.empty((B, C, H2 // 2, W2 // 2), dtype=out_dtype, device=dy.device)
    check(_lib.load_library().vs_upsample2_bwd(dy.data_ptr(), dtype_code(dy), dx.data_ptr(), dtype_code(dx), B * C, H2 // 2, W2 // 2,
                                               stream_ptr()), 'vs_upsample2_bwd')
    return dx


# ------------------------------------------------------------------------------------------------ fused frame losses
def frames_sse_fwd(frames, full, idx):
    """frames [B, G, D] fp32, full [B, T, D] fp32, idx [G] int32 (device) -> sums [2] (frame 0; frames 1..)."""
    require_cuda(frames, full, idx)
    assert frames.is_contiguous() and full.is_contiguous() and frames.dtype == torch.float32 and full.dtype == torch.float32
    B, G, D = frames.shape
    sums = torch.empty((2,), dtype=torch.float32, device=frames.device)
    e0 = _pb()
    check(_lib.load_library().vs_frames_sse_fwd(frames.data_ptr(), full.data_ptr(), idx.data_ptr(), B, G, full.shape[1], D,
                                                sums.data_ptr(), stream_ptr()), 'vs_frames_sse_fwd')
    _pe(e0, 'vs_frames_sse_fwd', nbytes=float(2 * frames.numel() * 4))
    return sums


def _loss_args(frames, full, idx, s_old, s_new, t0, lambdas, average_tloss):
    import ctypes
    if isinstance(idx, tuple):                       # (t_random int32 [1] on the device, ae_shift, first_forecast)
        t_dev, ae_shift, first_forecast = idx
        assert t_dev.dtype == torch.int32 and t_dev.numel() == 1
        require_cuda(t_dev)
        idx_args = (None, t_dev.data_ptr(), int(ae_shift), int(first_forecast))
    else:
        require_cuda(idx)
        assert idx.dtype == torch.int32 and idx.is_contiguous()
        idx_args = (idx.data_ptr(), None, 0, 0)
    require_cuda(frames, full, t0)
    assert frames.is_contiguous() and full.is_contiguous() and t0.is_contiguous()
    assert frames.dtype == torch.float32 and full.dtype == torch.float32 and t0.dtype == torch.float32
    B, G, D = frames.shape
    n_s = 0 if s_old is None else s_old.numel()
    if n_s:
        require_cuda(s_old, s_new)
        assert s_old.is_contiguous() and s_new.is_contiguous() and s_old.dtype == torch.float32 and s_new.dtype == torch.float32
    lam = (ctypes.c_float * 4)(*[float(v) for v in lambdas])
    return (frames.data_ptr(), full.data_ptr()) + idx_args + (B, G, full.shape[1], D, _ptr(s_old) if n_s else None,
            _ptr(s_new) if n_s else None, n_s, t0.data_ptr(), t0.shape[0], t0.numel() // t0.shape[0], int(bool(average_tloss)),
            ctypes.cast(lam, ctypes.c_void_p)), lam


def train_losses_fwd(frames, full, idx, s_old, s_new, t0, lambdas, average_tloss):
    """-> out [10] device floats: [4] total, [5] ae, [6] zero-order, [7] pred, [8] t_reg.  lambdas = (ae, s, t, pred)."""
    args, keep = _loss_args(frames, full, idx, s_old, s_new, t0, lambdas, average_tloss)
    out = torch.empty((10,), dtype=torch.float32, device=frames.device)
    e0 = _pb()
    check(_lib.load_library().vs_train_losses_fwd(*args, out.data_ptr(), stream_ptr()), 'vs_train_losses_fwd')
    _pe(e0, 'vs_train_losses_fwd', nbytes=float(2 * frames.numel() * 4))
    return out


def train_losses_fwd_grad(frames, full, idx, s_old, s_new, t0, lambdas, average_tloss, grad_total, frames_act, dz_dtype):
    """train_losses_fwd and train_losses_bwd (dz form) in one pass, for the upstream gradient `grad_total` (one device float) known now.
    -> (out [10], dz, ds_old, ds_new, dt0)."""
    args, keep = _loss_args(frames, full, idx, s_old, s_new, t0, lambdas, average_tloss)
    require_cuda(grad_total)
    assert grad_total.dtype == torch.float32 and grad_total.numel() == 1
    out = torch.empty((16 + 2 * 4096,), dtype=torch.float32, device=frames.device)      # [0..9] results, [16..] per-workgroup partial sums
    dz = torch.empty(frames.shape, dtype=dz_dtype, device=frames.device)
    ds_old = torch.empty_like(s_old) if s_old is not None else None
    ds_new = torch.empty_like(s_new) if s_new is not None else None
    dt0 = torch.empty_like(t0)
    e0 = _pb()
    check(_lib.load_library().vs_train_losses_fwd_grad(*args, out.data_ptr(), grad_total.data_ptr(), _ptr(ds_old), _ptr(ds_new), dt0.data_ptr(),
                                                       ACT[frames_act], dz.data_ptr(), dtype_code(dz), stream_ptr()), 'vs_train_losses_fwd_grad')
    _pe(e0, 'vs_train_losses_fwd', nbytes=float(2 * frames.numel() * 4 + dz.numel() * dz.element_size()))
    return out, dz, ds_old, ds_new, dt0


def gemm_frame_loss(h, w, bias, act, full, idx, G, s_old, s_new, t0, lambdas, average_tloss, grad_total, dz_dtype):
    """The decoder's last layer with the frame losses in its epilogue (vs_gemm_frame_loss): h [B * G, K] and w [N, K] 16-bit, the frames
    act(h w^T + bias) are compared with full [B, T, N] in registers and never stored.  idx = (t_random int32 [1] on the device, ae_shift,
    first_forecast).  -> (out, dz [B * G, N], ds_old, ds_new, dt0) as train_losses_fwd_grad, or None when the problem does not run on the
    256 x 256 tile kernel (the caller then stores the frames and uses train_losses_fwd_grad)."""
    import ctypes
    require_cuda(h, w, bias, full, s_old, s_new, t0, grad_total)
    t_dev, ae_shift, first_forecast = idx
    require_cuda(t_dev)
    assert h.dtype == w.dtype and h.dtype in (torch.bfloat16, torch.float16) and h.stride(-1) == 1 and w.stride(-1) == 1
    assert full.dtype == torch.float32 and full.is_contiguous() and full.dim() == 3 and t0.is_contiguous() and t0.dtype == torch.float32
    assert t_dev.dtype == torch.int32 and t_dev.numel() == 1 and grad_total.dtype == torch.float32 and grad_total.numel() == 1
    M, K = h.shape
    N = w.shape[0]
    assert M % G == 0 and full.shape[0] == M // G and full.shape[2] == N
    n_s = 0 if s_old is None else s_old.numel()
    lam = (ctypes.c_float * 4)(*[float(v) for v in lambdas])
    out = torch.empty((16 + 2 * 4096,), dtype=torch.float32, device=h.device)
    dz = torch.empty((M, N), dtype=dz_dtype, device=h.device)
    ds_old = torch.empty_like(s_old) if n_s else None
    ds_new = torch.empty_like(s_new) if n_s else None
    dt0 = torch.empty_like(t0)
    e0 = _pb()
    rc = _lib.load_library().vs_gemm_frame_loss(
        dtype_code(h), M, N, K, h.data_ptr(), h.stride(0), w.data_ptr(), w.stride(0), _ptr(bias), ACT[act], full.data_ptr(), t_dev.data_ptr(),
        int(ae_shift), int(first_forecast), int(G), full.shape[1], _ptr(s_old) if n_s else None, _ptr(s_new) if n_s else None, n_s, t0.data_ptr(),
        t0.shape[0], t0.numel() // t0.shape[0], int(bool(average_tloss)), ctypes.cast(lam, ctypes.c_void_p), grad_total.data_ptr(), dz.data_ptr(),
        dtype_code(dz), _ptr(ds_old), _ptr(ds_new), dt0.data_ptr(), out.data_ptr(), stream_ptr())
    if rc == -4:                                     # VS_ERR_UNSUPPORTED
        return None
    check(rc, 'vs_gemm_frame_loss')
    _pe(e0, 'vs_gemm<%s,RR>' % _DT[dtype_code(h)], flops=2.0 * M * N * K,
        nbytes=float((M * K + N * K) * h.element_size() + M * N * (4 + dz.element_size())))
    return out, dz, ds_old, ds_new, dt0


def train_losses_bwd(frames, full, idx, s_old, s_new, t0, lambdas, average_tloss, grad_total, frames_act=None, dz_dtype=None):
    """-> (dframes, ds_old, ds_new, dt0).  With `frames_act` (the activation whose outputs `frames` are) the first result is instead
    the gradient of that activation's input, dframes * act'(frames), in `dz_dtype`."""
    args, keep = _loss_args(frames, full, idx, s_old, s_new, t0, lambdas, average_tloss)
    require_cuda(grad_total)
    fused = frames_act is not None
    dframes = torch.empty(frames.shape, dtype=dz_dtype if fused else torch.float32, device=frames.device)
    ds_old = torch.empty_like(s_old) if s_old is not None else None
    ds_new = torch.empty_like(s_new) if s_new is not None else None
    dt0 = torch.empty_like(t0)
    e0 = _pb()
    check(_lib.load_library().vs_train_losses_bwd(*args, grad_total.data_ptr(), None if fused else dframes.data_ptr(), _ptr(ds_old),
                                                  _ptr(ds_new), dt0.data_ptr(), ACT[frames_act] if fused else 0,
                                                  dframes.data_ptr() if fused else None, dtype_code(dframes) if fused else 0, stream_ptr()),
          'vs_train_losses_bwd')
    _pe(e0, 'vs_train_losses_bwd', nbytes=float(2 * frames.numel() * 4 + dframes.numel() * dframes.element_size()))
    return dframes, ds_old, ds_new, dt0


def gather_windows(data, item_idx, windows_per_seq, seq_len, pixel_idx=None, out_dtype=torch.float32):
    """data [n_seq, nt, frame] fp32 on the device, item_idx int32 [B] -> [B, seq_len, frame (or len(pixel_idx))]."""
    require_cuda(data, item_idx, pixel_idx)
    assert data.dtype == torch.float32 and data.is_contiguous() and data.dim() == 3
    assert item_idx.dtype == torch.int32 and item_idx.is_contiguous()
    n_seq, nt, frame = data.shape
    B = item_idx.numel()
    n_pix = 0
    if pixel_idx is not None:
        assert pixel_idx.dtype == torch.int32 and pixel_idx.is_contiguous()
        n_pix = pixel_idx.numel()
    out = torch.empty((B, seq_len, n_pix if pixel_idx is not None else frame), dtype=out_dtype, device=data.device)
    e0 = _pb()
    check(_lib.load_library().vs_gather_windows(data.data_ptr(), n_seq, nt, frame, item_idx.data_ptr(), B, int(windows_per_seq), int(seq_len),
                                                _ptr(pixel_idx), n_pix, out.data_ptr(), dtype_code(out), stream_ptr()), 'vs_gather_windows')
    _pe(e0, 'vs_gather_windows', nbytes=float(out.numel() * (4 + out.element_size())))
    return out


_MIXING = {'concat': 0, 'mul': 1}


def mix_codes_fwd(s, t_rand, t_codes, mixing, lowp=None):
    """-> (z [B, 1+n, Cz] fp32, the same in the 16-bit dtype `lowp` or None); mixing 'concat' | 'mul' (mlp_encdec.py:43-48)."""
    require_cuda(s, t_rand, t_codes)
    assert s.dtype == t_rand.dtype == t_codes.dtype == torch.float32
    assert s.is_contiguous() and t_rand.is_contiguous() and t_codes.is_contiguous()
    B, Cs = s.shape
    n, Ct = t_codes.shape[1], t_codes.shape[2]
    assert t_rand.shape == (B, Ct) and t_codes.shape[0] == B
    Cz = Cs if mixing == 'mul' else Cs + Ct
    z = torch.empty((B, n + 1, Cz), dtype=torch.float32, device=s.device)
    z_lowp = torch.empty((B, n + 1, Cz), dtype=lowp, device=s.device) if lowp is not None else None
    e0 = _pb()
    check(_lib.load_library().vs_mix_codes_fwd(s.data_ptr(), t_rand.data_ptr(), t_codes.data_ptr(), B, n, Cs, Ct, _MIXING[mixing],
                                               z.data_ptr(), _ptr(z_lowp), code_of(lowp) if lowp is not None else BF16, stream_ptr()), 'vs_mix_codes_fwd')
    _pe(e0, 'vs_mix_codes_fwd', nbytes=float(z.numel() * 8))
    return z, z_lowp


def mix_codes_bwd(dz, s, t_rand, t_codes, mixing):
    require_cuda(dz, s, t_rand, t_codes)
    assert dz.dtype == torch.float32 and dz.is_contiguous()
    B, Cs = s.shape
    n, Ct = t_codes.shape[1], t_codes.shape[2]
    ds, dt_rand, dt_codes = torch.empty_like(s), torch.empty_like(t_rand), torch.empty_like(t_codes)
    e0 = _pb()
    check(_lib.load_library().vs_mix_codes_bwd(dz.data_ptr(), s.data_ptr(), t_rand.data_ptr(), t_codes.data_ptr(), B, n, Cs, Ct,
                                               _MIXING[mixing], ds.data_ptr(), dt_rand.data_ptr(), dt_codes.data_ptr(), stream_ptr()),
          'vs_mix_codes_bwd')
    _pe(e0, 'vs_mix_codes_bwd', nbytes=float(dz.numel() * 8))
    return ds, dt_rand, dt_codes


def frames_sse_bwd(frames, full, idx, coef):
    require_cuda(frames, full, idx, coef)
    B, G, D = frames.shape
    out = torch.empty_like(frames)
    e0 = _pb()
    check(_lib.load_library().vs_frames_sse_bwd(frames.data_ptr(), full.data_ptr(), idx.data_ptr(), B, G, full.shape[1], D,
                                                coef.data_ptr(), out.data_ptr(), stream_ptr()), 'vs_frames_sse_bwd')
    _pe(e0, 'vs_frames_sse_bwd', nbytes=float(3 * frames.numel() * 4))
    return out


def cat_bcast_supported(a, x, n):
    return (a.is_cuda and x.is_cuda and a.dim() == 4 and x.dim() == 4 and a.is_contiguous() and x.is_contiguous() and x.shape[0] == n * a.shape[0]
            and a.shape[2:] == x.shape[2:] and (a.shape[2] * a.shape[3]) % 8 == 0 and a.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0
            and a.dtype in (torch.float32, torch.bfloat16, torch.float16) and x.dtype in (torch.float32, torch.bfloat16, torch.float16))


def cat_bcast_fwd(a, x, n, out_dtype):
    """cat([a.repeat(n, 1, 1, 1), x], dim=1) in out_dtype: a [B, Ca, H, W], x [n B, Cb, H, W] -> [n B, Ca + Cb, H, W], one pass."""
    require_cuda(a, x)
    B, Ca = a.shape[0], a.shape[1]
    Cb, HW = x.shape[1], x.shape[2] * x.shape[3]
    out = torch.empty((n * B, Ca + Cb) + tuple(x.shape[2:]), dtype=out_dtype, device=x.device)
    e0 = _pb()
    check(_lib.load_library().vs_cat_bcast_fwd(a.data_ptr(), dtype_code(a), x.data_ptr(), dtype_code(x), out.data_ptr(), dtype_code(out), B, n, Ca, Cb, HW,
                                               stream_ptr()), 'vs_cat_bcast_fwd')
    _pe(e0, 'vs_cat_bcast', nbytes=float(x.numel() * x.element_size() + out.numel() * out.element_size()))
    return out


def cat_bcast_bwd(dout, B, n, Ca, a_dtype, x_dtype, need_a=True, need_x=True):
    """(da [B, Ca, H, W] = sum over the n frames of dout[:, :Ca], dx [n B, Cb, H, W] = dout[:, Ca:]) in one pass over dout."""
    require_cuda(dout)
    dout = dout.contiguous()
    C, H, W = dout.shape[1], dout.shape[2], dout.shape[3]
    Cb = C - Ca
    da = torch.empty((B, Ca, H, W), dtype=a_dtype, device=dout.device) if need_a else None
    dx = torch.empty((n * B, Cb, H, W), dtype=x_dtype, device=dout.device) if need_x else None
    e0 = _pb()
    check(_lib.load_library().vs_cat_bcast_bwd(dout.data_ptr(), dtype_code(dout), _ptr(da), code_of(a_dtype), _ptr(dx), code_of(x_dtype), B, n, Ca, Cb, H * W,
                                               stream_ptr()), 'vs_cat_bcast_bwd')
    _pe(e0, 'vs_cat_bcast', nbytes=float(dout.numel() * dout.element_size() * 2))
    return da, dx


def _code_loss_tables(pairs):
    import ctypes
    n = len(pairs)
    VP, I32, I64 = ctypes.c_void_p * max(n, 1), ctypes.c_int32 * max(n, 1), ctypes.c_int64 * max(n, 1)
    return n, VP, I32(*([dtype_code(a) for a, _ in pairs] or [0])), I64(*([a.numel() for a, _ in pairs] or [0]))


def code_losses_supported(pairs, t0):
    """Whether the fused code-loss kernels take these (a, b) pairs: <= 10 of them, equal shapes / types, contiguous, element counts multiples of 8."""
    if len(pairs) > 10 or t0 is None or t0.dtype != torch.float32 or not t0.is_contiguous() or not t0.is_cuda:
        return False
    for a, b in pairs:
        if (a.shape != b.shape or a.dtype != b.dtype or not a.is_cuda or not a.is_contiguous() or not b.is_contiguous() or a.numel() % 8 != 0
                or a.numel() == 0 or a.dtype not in (torch.float32, torch.bfloat16, torch.float16) or a.data_ptr() % 16 or b.data_ptr() % 16):
            return False
    return True


def code_losses_fwd(pairs, t0, sse_ae, sse_pred, scale_ae, scale_pred, lambdas, inv_t):
    """out [5] = (total, ae, zero, pred, t_reg) from the pairs of the zero-order loss, the initial temporal code and the raw frame sums
    (vs_code_losses_fwd: a partial-sum launch over 4096-element chunks + a one-block finish)."""
    require_cuda(t0, sse_ae, sse_pred)
    lib = _lib.load_library()
    n, VP, dts, cnts = _code_loss_tables(pairs)
    total = sum(a.numel() for a, _ in pairs)
    chunks = lib.vs_code_losses_chunks(n, cnts, t0.numel())
    partial = torch.empty((max(int(chunks), 1),), dtype=torch.float32, device=t0.device)
    out = torch.empty((5,), dtype=torch.float32, device=t0.device)
    l_ae, l_s, l_t, l_pred = (float(v) for v in lambdas)
    e0 = _pb()
    check(lib.vs_code_losses_fwd(n, VP(*([a.data_ptr() for a, _ in pairs] or [0])), VP(*([b.data_ptr() for _, b in pairs] or [0])), dts, cnts, t0.data_ptr(),
                                 t0.numel(), sse_ae.data_ptr(), sse_pred.data_ptr(), float(scale_ae), float(scale_pred), l_ae, l_s, l_pred, l_t,
                                 1.0 / max(total, 1), float(inv_t), partial.data_ptr(), out.data_ptr(), stream_ptr()), 'vs_code_losses_fwd')
    _pe(e0, 'vs_code_losses_fwd', nbytes=float(sum(2 * a.numel() * a.element_size() for a, _ in pairs) + t0.numel() * 4))
    return out


def code_losses_bwd(pairs, need, t0, g, scale_ae, scale_pred, lambdas, inv_t):
    """(d a_j or None, d b_j or None per pair, d t0, coefs [4]) for the upstream gradient g (one fp32 element on the device); need[j] = (bool, bool)."""
    require_cuda(t0, g)
    lib = _lib.load_library()
    n, VP, dts, cnts = _code_loss_tables(pairs)
    total = sum(a.numel() for a, _ in pairs)
    da = [torch.empty_like(a) if need[j][0] else None for j, (a, _) in enumerate(pairs)]
    db = [torch.empty_like(b) if need[j][1] else None for j, (_, b) in enumerate(pairs)]
    dt0 = torch.empty_like(t0)
    coefs = torch.empty((4,), dtype=torch.float32, device=t0.device)
    l_ae, l_s, l_t, l_pred = (float(v) for v in lambdas)
    e0 = _pb()
    check(lib.vs_code_losses_bwd(n, VP(*([a.data_ptr() for a, _ in pairs] or [0])), VP(*([b.data_ptr() for _, b in pairs] or [0])),
                                 VP(*([_ptr(t) for t in da] or [0])), VP(*([_ptr(t) for t in db] or [0])), dts, cnts, t0.data_ptr(), dt0.data_ptr(), t0.numel(),
                                 g.data_ptr(), float(scale_ae), float(scale_pred), l_ae, l_s, l_pred, l_t, 1.0 / max(total, 1), float(inv_t), coefs.data_ptr(),
                                 stream_ptr()), 'vs_code_losses_bwd')
    _pe(e0, 'vs_code_losses_bwd', nbytes=float(sum(4 * a.numel() * a.element_size() for a, _ in pairs) + t0.numel() * 8))
    return da, db, dt0, coefs


# ------------------------------------------------------------------------------------------------ fp16 loss scaling
def check_finite_multi(grads, scale_state):
    """scale_state[1] (found_inf) = 1 when any gradient tensor holds an inf / NaN; one launch per 64 tensors."""
    import ctypes
    require_cuda(scale_state, *grads)
    assert scale_state.dtype == torch.float32 and scale_state.numel() >= 4
    lib = _lib.load_library()
    found = scale_state.data_ptr() + 4
    for i in range(0, len(grads), 64):
        chunk = grads[i:i + 64]
        n = len(chunk)
        for g in chunk:
            assert g.is_contiguous()
        VP, I32, I64 = ctypes.c_void_p * n, ctypes.c_int32 * n, ctypes.c_int64 * n
        e0 = _pb()
        check(lib.vs_check_finite_multi(n, VP(*[g.data_ptr() for g in chunk]), I32(*[dtype_code(g) for g in chunk]),
                                        I64(*[g.numel() for g in chunk]), found, stream_ptr()), 'vs_check_finite_multi')
        _pe(e0, 'vs_check_finite_multi', nbytes=float(sum(g.numel() * g.element_size() for g in chunk)))


def loss_scale_update(scale_state, growth_factor, backoff_factor, growth_interval):
    require_cuda(scale_state)
    check(_lib.load_library().vs_loss_scale_update(scale_state.data_ptr(), float(growth_factor), float(backoff_factor), int(growth_interval),
                                                   stream_ptr()), 'vs_loss_scale_update')


# ------------------------------------------------------------------------------------------------ evaluation metrics
def frame_metrics(pred, target, max_val=1.0, k1=0.01, k2=0.03, sigma=1.5, want_ssim=True):
    """pred, target [..., H, W] fp32 (any leading dims): per-plane (mse, mean SSIM) with the leading shape (vs_frame_metrics)."""
    require_cuda(pred, target)
    assert pred.shape == target.shape and pred.dim() >= 2
    H, W = pred.shape[-2], pred.shape[-1]
    lead = tuple(pred.shape[:-2])
    p = pred.reshape(-1, H, W).float().contiguous()
    t = target.reshape(-1, H, W).float().contiguous()
    mse = torch.empty((p.shape[0],), dtype=torch.float32, device=p.device)
    ssim = torch.empty((p.shape[0],), dtype=torch.float32, device=p.device) if want_ssim else None
    check(_lib.load_library().vs_frame_metrics(p.data_ptr(), t.data_ptr(), p.shape[0], H, W, float(max_val), float(k1), float(k2), float(sigma),
                                               mse.data_ptr(), _ptr(ssim), stream_ptr()), 'vs_frame_metrics')
    return mse.view(lead), (ssim.view(lead) if want_ssim else None)
