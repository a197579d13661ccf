"""MI355X-native (gfx950) implementation of the var_sep training hot path.

Package layout: `csrc/` holds the HIP kernels and the C ABI (`include/varsep_hip.h`), `_lib.py` binds it with
ctypes, `ops.py`/`functional.py` wrap it for torch tensors and autograd, and `networks/`, `train.py`, `options.py`,
`main.py` mirror the reference's Python surface (`var_sep.*`) for this path.
"""
import os as _os

# HIP runtime knobs, read by the runtime when it initialises (before the first HIP call of the process).  SINGLE-GPU runs only: a
# recorded training step forks into the integrator's stream and up to three gradient streams beside the main chain; with the defaults
# (4 hardware queues, 4 graph streams) independent branches of a replayed hipGraph are mapped onto the same queue and run one after
# the other (WaveEq step, same box: 1.68 -> 1.58 ms with 8 / 8).  Data-parallel ranks keep the defaults: with 8 hardware queues the
# graph A -> RCCL all-reduce -> graph B step ran at 4.1 ms instead of 2.2 ms.  Explicit settings in the environment win.
if int(_os.environ.get('WORLD_SIZE', '1') or 1) <= 1 and _os.environ.get('VARSEP_BENCH_FORCE_DIST') != '1':
    for _k in ('GPU_MAX_HW_QUEUES', 'DEBUG_HIP_FORCE_GRAPH_QUEUES'):
        if _k not in _os.environ:
            _os.environ[_k] = '8'
            # (a launcher that starts data-parallel ranks from this process must not hand them these: bench.spawn_ranks drops
            # what is listed here)
            _os.environ['VARSEP_PACKAGE_SET'] = (_os.environ.get('VARSEP_PACKAGE_SET', '') + ' ' + _k).strip()

__version__ = '0.2.0'
