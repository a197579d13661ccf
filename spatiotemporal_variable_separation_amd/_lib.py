"""ctypes binding of libvarsep_hip.so (C ABI declared in include/varsep_hip.h).

The library is built in-tree by `build_library()` (hipcc, --offload-arch=gfx950) and loaded lazily.  There is
no CPU fallback: if the shared object is missing or a call fails, the product path raises.
"""
import ctypes
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, 'csrc')
LIB_PATH = os.path.join(_HERE, 'libvarsep_hip.so')
SOURCES = ['vs_gemm.hip', 'vs_eltwise.hip', 'vs_conv.hip', 'vs_rollout.hip', 'vs_norm.hip', 'vs_optim.hip', 'vs_data.hip', 'vs_conv_tap.hip', 'vs_metrics.hip', 'vs_conv_img.hip', 'vs_conv_k4s2.hip',
           'vs_conv_thin.hip', 'vs_conv_band2.hip', 'vs_conv_wgrad2.hip']

F32, BF16, F16 = 0, 1, 2
TORCH_DTYPE = {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}
ACT = {'none': 0, None: 0, 'identity': 0, 'relu': 1, 'leaky_relu': 2, 'sigmoid': 3, 'tanh': 4, 'elu': 5}
LAYOUT_R, LAYOUT_S = 0, 1

_lib = None


class VarsepHipError(RuntimeError):
    pass


def _sources():
    return [os.path.join(_CSRC, s) for s in SOURCES if os.path.exists(os.path.join(_CSRC, s))]


def _headers():
    """Every header a source may include: all of csrc/*.h plus the public C-ABI header."""
    import glob
    return sorted(glob.glob(os.path.join(_CSRC, '*.h'))) + [os.path.join(_HERE, '..', 'include', 'varsep_hip.h')]


def _deps(src, _seen=None):
    """`src` and the project headers it includes, transitively (quoted includes, resolved in csrc/ and include/): a header edit recompiles
    only the sources that see it."""
    import re
    seen = _seen if _seen is not None else set()
    if src in seen or not os.path.exists(src):
        return seen
    seen.add(src)
    with open(src, errors='replace') as f:
        for name in re.findall(r'^\s*#\s*include\s+"([^"]+)"', f.read(), flags=re.M):
            for d in (os.path.dirname(src), _CSRC, os.path.join(_HERE, '..', 'include')):
                cand = os.path.normpath(os.path.join(d, name))
                if os.path.exists(cand):
                    _deps(cand, seen)
                    break
    return seen


def library_is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in _sources() + _headers())


def build_library(force=False, verbose=False):
    """Compile csrc/*.hip for gfx950 into libvarsep_hip.so (cross-compiles without a GPU)."""
    if not force and not library_is_stale():
        return LIB_PATH
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objdir = os.path.join(_HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC'] + os.environ.get('VARSEP_HIPCC_FLAGS', '').split()

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + '.o')
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(d) for d in _deps(src)):
            return obj
        cmd = [hipcc] + flags + ['-c', src, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise VarsepHipError('hipcc failed for %s:\n%s' % (src, r.stderr[-4000:]))
        if verbose:
            print('compiled', os.path.basename(src), file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, _sources()))
    tmp = LIB_PATH + '.tmp'
    r = subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', tmp] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise VarsepHipError('link failed:\n' + r.stderr[-4000:])
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


_i64, _i32, _f32, _vp, _sz = ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol declared in include/varsep_hip.h
SIGNATURES = {
    'vs_version': (ctypes.c_char_p, []),
    'vs_last_error': (ctypes.c_char_p, []),
    'vs_gemm_workspace_bytes': (_sz, [_i64, _i64, _i64]),
    'vs_gemm': (_i32, [_i32, _i64, _i64, _i64, _vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _i32, _f32, _vp, _i32,
                       _vp, _i64, _i32, _i32, _i32, _vp, _sz, _vp]),
    'vs_cast': (_i32, [_vp, _i32, _vp, _i32, _i64, _vp]),
    'vs_copy2d': (_i32, [_vp, _i32, _i64, _vp, _i32, _i64, _i64, _i64, _vp, _i64, _vp]),
    'vs_copy2d_pair': (_i32, [_vp, _i32, _i64, _vp, _i32, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp]),
    'vs_colsum': (_i32, [_vp, _i32, _i64, _i64, _i64, _vp, _i32, _vp]),
    'vs_colsum_multi': (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    'vs_adam_multi': (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                             ctypes.c_double, _vp]),
    'vs_adam_step_increment': (_i32, [_vp, _vp]),
    'vs_check_finite_multi': (_i32, [_i32, _vp, _vp, _vp, _vp, _vp]),
    'vs_adam_set_max_blocks': (_i32, [_i32]),
    'vs_gemm_adam': (_i32, [_i32, _i64, _i64, _i64, _vp, _i64, _i32, _vp, _i64, _i32, ctypes.c_float, _vp, _vp, _vp, _vp, _i32, _vp, _i32,
                     ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _vp]),
    'vs_adam_multi_scaled': (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                    ctypes.c_double, _vp, _vp]),
    'vs_adam_step_increment_scaled': (_i32, [_vp, _vp, _vp]),
    'vs_loss_scale_update': (_i32, [_vp, _f32, _f32, _i32, _vp]),
    'vs_gemm_batched_workspace_bytes': (_sz, [_i32, _i64, _i64, _i64]),
    'vs_gemm_batched': (_i32, [_i32, _i32, _i64, _i64, _i64, _vp, _i64, _i64, _i32, _vp, _i64, _i64, _i32, _vp, _i64, _i64, _i32, _f32, _i32,
                               _vp, _sz, _vp]),
    'vs_train_losses_fwd': (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i64, _i32, _i32, _i64, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _vp]),
    'vs_train_losses_fwd_grad': (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i64, _i32, _i32, _i64, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _vp,
                                        _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp]),
    'vs_gemm_frame_loss': (_i32, [_i32, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _i64, _vp, _i64,
                                  _i64, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    'vs_train_losses_bwd': (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i64, _i32, _i32, _i64, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp,
                                   _vp, _vp, _i32, _vp, _i32, _vp]),
    'vs_convt_tap_supported': (_i32, [_i32] * 7),
    'vs_convt_tap_packed_elems': (_sz, [_i32, _i32]),
    'vs_convt_tap_pack_weight': (_i32, [_i32, _vp, _i32, _i32, _vp, _vp]),
    'vs_convt_k4s2_tap_fwd': (_i32, [_i32, _vp, _vp, _vp, _vp, _vp] + [_i32] * 6 + [_vp]),
    'vs_convt_k4s2_tap_fwd_f32': (_i32, [_i32, _vp, _vp, _vp, _vp] + [_i32] * 5 + [_vp]),
    'vs_conv_k3_tap_supported': (_i32, [_i32] * 7),
    'vs_conv_k3_tap_packed_elems': (_sz, [_i32, _i32]),
    'vs_conv_k3_tap_pack_weight': (_i32, [_i32, _vp, _i32, _i32, _i32, _vp, _vp]),
    'vs_conv_k3s1_tap_fwd': (_i32, [_i32, _vp, _vp, _vp, _vp, _i32, _vp] + [_i32] * 6 + [_vp]),
    'vs_bn_stats_from_sums': (_i32, [_vp, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _vp]),
    'vs_moving_mnist_batch': (_i32, [_vp, _i64, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    'vs_gather_windows': (_i32, [_vp, _i64, _i64, _i64, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp]),
    'vs_mix_codes_fwd': (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    'vs_mix_codes_bwd': (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    'vs_pack_rollout_weights': (_i32, [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    'vs_frames_sse_fwd': (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i64, _vp, _vp]),
    'vs_frames_sse_bwd': (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i64, _vp, _vp, _vp]),
    'vs_cat_bcast_fwd': (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _vp]),
    'vs_cat_bcast_bwd': (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _vp]),
    'vs_code_losses_chunks': (_i64, [_i32, _vp, _i64]),
    'vs_code_losses_fwd': (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp]),
    'vs_code_losses_bwd': (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _vp, _vp]),
    'vs_act_bwd': (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i64, _vp]),
    'vs_act_fwd': (_i32, [_vp, _i32, _vp, _i32, _i32, _i64, _vp]),
    'vs_conv_wgrad_workspace_bytes': (_sz, [_i32] * 7),
    'vs_conv_packed_elems': (_sz, [_i32] * 6),
    'vs_conv_pack_weight': (_i32, [_i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    'vs_conv_workspace_bytes': (_sz, [_i32] * 10),
    'vs_conv2d_fwd': (_i32, [_i32, _vp, _vp, _vp, _vp, _i32] + [_i32] * 9 + [_vp, _sz, _vp]),
    'vs_conv2d_dgrad': (_i32, [_i32, _vp, _vp, _vp, _i32] + [_i32] * 9 + [_vp, _sz, _vp]),
    'vs_conv2d_wgrad': (_i32, [_i32, _vp, _vp, _vp] + [_i32] * 9 + [_vp, _sz, _vp]),
    'vs_frame_metrics': (_i32, [_vp, _vp, _i64, _i32, _i32, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp]),
    'vs_conv3_img16_supported': (_i32, [_i32] * 6),
    'vs_conv3_img16_splits': (_i32, [_i32] * 3),
    'vs_conv3_img16_packed_elems': (_sz, [_i32, _i32]),
    'vs_conv3_img16_pack_weight': (_i32, [_i32, _vp, _i32, _i32, _i32, _vp, _vp]),
    'vs_conv3_img16_pack_weights': (_i32, [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    'vs_conv3_img16': (_i32, [_i32, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    'vs_space_to_depth2_supported': (_i32, [_i32] * 5),
    'vs_space_to_depth2': (_i32, [_i32, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    'vs_conv_k4s2_packed_elems': (_sz, [_i32, _i32]),
    'vs_conv_k4s2_pack_weight': (_i32, [_i32, _vp, _i32, _i32, _vp, _vp]),
    'vs_conv_k4s2_wgrad_finish': (_i32, [_vp, _i32, _vp, _vp, _i32, _i32, _vp]),
    'vs_conv_k4s2_skip_form': (_i32, [_i32]),
    'vs_conv_k4s2_band': (_i32, [_i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    'vs_conv_k4s2_wgrad_band': (_i32, [_i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    'vs_conv_k4s2_wgrad_band_slabs': (_i32, [_i32, _i32, _i32, _i32, _i32]),
    'vs_conv_thin_supported': (_i32, [_i32] * 9),
    'vs_conv_thin_expand': (_i32, [_i32, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _i32] + [_i32] * 7 + [_vp]),
    'vs_conv_thin_reduce': (_i32, [_i32, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _i32] + [_i32] * 7 + [_vp]),
    'vs_conv_thin_wgrad_workspace_bytes': (_sz, [_i32] * 6),
    'vs_conv_thin_wgrad': (_i32, [_i32, _vp, _vp, _vp, _sz, _vp, _vp, _i64, _i64, _i32] + [_i32] * 7 + [_vp]),
    'vs_conv3_band_supported': (_i32, [_i32] * 6),
    'vs_conv3_band_bn_supported': (_i32, [_i32] * 7),
    'vs_conv3_band_bn': (_i32, [_i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    'vs_conv_k4s2_band_bn': (_i32, [_i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    'vs_bn_stats_from_sums_fold': (_i32, [_vp, _i32, _i32, _i64, _vp, _vp, _vp, _vp, ctypes.c_float, ctypes.c_float, _i32, _vp]),
    'vs_conv3_band_bn_parts_rows': (_i32, [_i32, _i32, _i32]),
    'vs_conv3_band_bn_parts': (_i32, [_i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    'vs_bn_stats_from_parts_fold': (_i32, [_vp, _i32, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, ctypes.c_float, _vp]),
    'vs_conv3_band': (_i32, [_i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    'vs_conv3_wgrad_band_supported': (_i32, [_i32] * 6),
    'vs_conv3_wgrad_band_slabs': (_i32, [_i32] * 5),
    'vs_conv3_wgrad_band': (_i32, [_i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    'vs_conv3_wgrad_band_pieces': (_i32, [_i32, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp]),
    'vs_conv3_wgrad_band_finish': (_i32, [_vp, _i32, _vp, _vp, _i32, _i32, _vp]),
    'vs_slab_sum_grouped': (_i32, [_vp, _i32, _i32, _vp, _i64, _vp]),
    'vs_slab_sum': (_i32, [_vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _vp]),
    'vs_slab_sum2': (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _vp]),
    'vs_conv3_img16_bn_workspace_bytes': (ctypes.c_size_t, []),
    'vs_conv3_img16_bn_supported': (_i32, [_i32, _i32, _i32, _i32]),
    'vs_conv3_img16_bn_form_supported': (_i32, [_i32, _i32, _i32, _i32]),
    'vs_exchange_epoch_advance': (_i32, [_vp, _vp]),
    'vs_conv3_img16_bn_fwd': (_i32, [_i32, _vp, _vp, _vp, ctypes.c_uint, _vp, _vp, _vp, _i32, _vp, _vp, ctypes.c_float, ctypes.c_float, _vp, _vp, _i32,
                                     _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    'vs_conv3_img16_bn_bwd': (_i32, [_i32, _vp, _vp, _vp, ctypes.c_uint, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    'vs_bn_train_fwd_small_slabs': (_i32, [_vp, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, ctypes.c_float, ctypes.c_float,
                                           _vp, _vp, _vp, _i32, _i32, _i64, _vp]),
    'vs_bn_act_bwd_small_ex': (_i32, [_vp, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i64, _vp]),
    'vs_bn_train_fwd_small_supported': (_i32, [_i32, _i32, _i32, _i64]),
    'vs_bn_train_fwd_small': (_i32, [_vp, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, ctypes.c_float, ctypes.c_float, _i32, _i32, _i64, _vp]),
    'vs_bn_train_fwd_small_groups': (_i32, [_vp, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, ctypes.c_float, _i32, _i32, _i64, _i32,
                                            _vp]),
    'vs_conv2d_wgrad_acc': (_i32, [_i32, _vp, _vp, _vp] + [_i32] * 9 + [_vp, _sz, _i32, _vp]),
    'vs_conv_transpose2d_wgrad_acc': (_i32, [_i32, _vp, _vp, _vp] + [_i32] * 9 + [_vp, _sz, _i32, _vp]),
    'vs_conv_transpose2d_fwd': (_i32, [_i32, _vp, _vp, _vp, _vp, _i32] + [_i32] * 9 + [_vp, _sz, _vp]),
    'vs_conv_transpose2d_dgrad': (_i32, [_i32, _vp, _vp, _vp, _i32] + [_i32] * 9 + [_vp, _sz, _i32, _vp]),
    'vs_conv_transpose2d_wgrad': (_i32, [_i32, _vp, _vp, _vp] + [_i32] * 9 + [_vp, _sz, _vp]),
    'vs_bn_stats': (_i32, [_vp, _i32, _i32, _i32, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _vp]),
    'vs_bn_train_fwd_slab_supported': (_i32, [_i32, _i32, _i32, _i64]),
    'vs_bn_train_fwd_slab': (_i32, [_vp, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, ctypes.c_float, _i32, _i32, _i64, _i32, _vp]),
    'vs_bn_act_fwd': (_i32, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _i32, _vp]),
    'vs_bn_act_bwd': (_i32, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _vp]),
    'vs_bn_act_bwd_gsum': (_i32, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _vp, _vp, _vp]),
    'vs_bn_stats_ub': (_i32, [_vp, _i32, _i32, _i32, _i64, _i32, _vp, _vp, _vp, ctypes.c_float, _vp]),
    'vs_bn_act_fwd_running': (_i32, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i64, _i32, _vp, _vp, _vp, ctypes.c_float, _vp]),
    'vs_chan_sum': (_i32, [_vp, _i32, _i32, _i32, _i64, _vp, _vp]),
    'vs_chan_sum_workspace_bytes': (_sz, [_i32, _i32, _i64]),
    'vs_chan_sum_ws': (_i32, [_vp, _i32, _i32, _i32, _i64, _vp, _sz, _vp, _vp]),
    'vs_maxpool2_fwd': (_i32, [_vp, _i32, _vp, _i32, _i64, _i32, _i32, _vp]),
    'vs_maxpool2_bwd': (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _i64, _i32, _i32, _vp]),
    'vs_maxpool3s2_fwd': (_i32, [_vp, _i32, _vp, _i32, _i64, _i32, _i32, _vp]),
    'vs_maxpool3s2_bwd': (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _i64, _i32, _i32, _vp]),
    'vs_upsample2_fwd': (_i32, [_vp, _i32, _vp, _i32, _i64, _i32, _i32, _vp]),
    'vs_upsample2_bwd': (_i32, [_vp, _i32, _vp, _i32, _i64, _i32, _i32, _vp]),
    'vs_transpose_cast': (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _vp]),
    'vs_rollout_packed_elems': (_sz, [_i32, _i32, _i32]),
    'vs_pack_rollout_weight': (_i32, [_i32, _vp, _i32, _i32, _i32, _vp, _vp]),
    'vs_mlp_rollout_parts': (_i32, [_i32] * 4),
    'vs_mlp_rollout_workspace_bytes': (_sz, [_i32] * 4),
    'vs_mlp_rollout_xcd_local_get': (_i32, [_i32] * 5),
    'vs_mlp_rollout_xcd_local_set': (_i32, [_i32]),
    'vs_exchange_guard_set': (_i32, [_vp]),
    'vs_exchange_guard_get': (_vp, []),
    'vs_exchange_skip_counter_set': (_i32, [_vp]),
    'vs_mlp_rollout_fwd': (_i32, [_i32] * 6 + [_vp] * 11 + [_sz, _vp]),
    'vs_mlp_rollout_bwd': (_i32, [_i32] * 6 + [_vp] * 11 + [_sz, _vp]),
}


def load_library():
    """dlopen the in-tree library and attach argtypes; raises VarsepHipError when it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VarsepHipError(
            'libvarsep_hip.so is missing (%s). Build it with `python -c "import __graft_entry__ as g; g.build()"`; '
            'there is no CPU fallback for the HIP path.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch, fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise VarsepHipError('%s failed (%d): %s' % (what, rc, load_library().vs_last_error().decode()))


def dtype_code(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float16:
        return F16
    raise VarsepHipError('unsupported tensor dtype %s' % t.dtype)


def code_of(dtype):
    """dtype code of a torch dtype (F32 | BF16 | F16)."""
    for code, dt in TORCH_DTYPE.items():
        if dt == dtype:
            return code
    raise VarsepHipError('unsupported dtype %s' % dtype)


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise VarsepHipError('the HIP path needs tensors on an MI355X device (got a %s tensor); there is no CPU '
                                 'fallback -- use oracle/cpu_ref.py in tests for CPU results' % t.device)
