"""Thin tensor-level wrappers over the C ABI (include/varsep_hip.h).  No autograd here; see functional.py."""
import torch

from . import _lib
from ._lib import F32, BF16, ACT, LAYOUT_R, LAYOUT_S, check, dtype_code, stream_ptr, require_cuda

_ws_cache = {}


def _workspace(nbytes, device):
    """One grow-only split-K workspace per device (allocated by torch, so legal inside graph capture after warm-up)."""
    key = device.index
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
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
    check(lib.vs_gemm(compute, M, N, K, a.data_ptr(), lda, layout_a, b.data_ptr(), ldb, layout_b, out.data_ptr(),
                      out.stride(0), dtype_code(out), float(alpha), _ptr(bias), ACT[act], _ptr(mask),
                      mask.stride(0) if mask is not None else 0, dtype_code(mask) if mask is not None else 0,
                      ACT[mask_act], int(bool(accumulate)), _ptr(ws), ws.numel() if ws is not None else 0, stream_ptr()),
          'vs_gemm')
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


def colsum(x, M, N, out=None, accumulate=False):
    require_cuda(x)
    if out is None:
        out = torch.empty((N,), dtype=torch.float32, device=x.device)
    check(_lib.load_library().vs_colsum(x.data_ptr(), dtype_code(x), x.stride(0), M, N, out.data_ptr(),
                                        int(bool(accumulate)), stream_ptr()), 'vs_colsum')
    return out


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
