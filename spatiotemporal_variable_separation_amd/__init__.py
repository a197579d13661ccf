"""MI355X-native (gfx950) implementation of the var_sep training hot path.

Package layout: `csrc/` holds the HIP kernels and the C ABI (`include/varsep_hip.h`), `_lib.py` binds it with
ctypes, `ops.py`/`functional.py` wrap it for torch tensors and autograd, and `networks/`, `train.py`, `options.py`,
`main.py` mirror the reference's Python surface (`var_sep.*`) for this path.
"""
import os as _os

__version__ = '0.3.0'

_QUEUE_KNOBS = ('GPU_MAX_HW_QUEUES', 'DEBUG_HIP_FORCE_GRAPH_QUEUES')


def configure_single_gpu_queues(force=False):
    """HIP runtime knobs for SINGLE-GPU training processes; call before the first HIP call of the process (main.py and bench.py do, right
    at start-up -- importing the package does NOT touch the environment).  A recorded training step forks into the integrator's stream and
    up to three gradient streams beside the main chain; with the runtime's defaults (4 hardware queues, 4 graph streams) independent
    branches of a replayed hipGraph are mapped onto the same queue and run one after the other (WaveEq step, same box: 1.68 -> 1.58 ms
    with 8 / 8).  Data-parallel ranks keep the defaults: with 8 hardware queues the graph A -> RCCL all-reduce -> graph B step ran at
    4.1 ms instead of 2.2 ms -- so nothing is set when WORLD_SIZE > 1 or VARSEP_BENCH_FORCE_DIST=1, and a launcher that starts
    data-parallel ranks from a process that called this must drop the variables listed in VARSEP_PACKAGE_SET from the children's
    environment (bench.spawn_ranks does).  Explicit settings in the environment win.  Returns the names it set."""
    import sys
    import warnings
    if not force and (int(_os.environ.get('WORLD_SIZE', '1') or 1) > 1 or _os.environ.get('VARSEP_BENCH_FORCE_DIST') == '1'):
        return []
    torch = sys.modules.get('torch')
    if torch is not None and torch.cuda.is_initialized():
        missing = [k for k in _QUEUE_KNOBS if k not in _os.environ]
        if missing:
            warnings.warn('configure_single_gpu_queues(): the HIP runtime is already initialised; %s would have no effect in this '
                          'process and are NOT set (call it before the first GPU call)' % ', '.join(missing))
        return []
    done = []
    for k in _QUEUE_KNOBS:
        if k not in _os.environ:
            _os.environ[k] = '8'
            done.append(k)
    if done:
        _os.environ['VARSEP_PACKAGE_SET'] = (_os.environ.get('VARSEP_PACKAGE_SET', '') + ' ' + ' '.join(done)).strip()
    return done
