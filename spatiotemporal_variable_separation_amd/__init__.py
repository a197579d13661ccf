"""MI355X-native (gfx950) implementation of the var_sep training hot path.

Package layout: `csrc/` holds the HIP kernels and the C ABI (`include/varsep_hip.h`), `_lib.py` binds it with
ctypes, `ops.py`/`functional.py` wrap it for torch tensors and autograd, and `networks/`, `train.py`, `options.py`,
`main.py` mirror the reference's Python surface (`var_sep.*`) for this path.
"""
import os as _os

# HIP runtime knobs, read by the runtime when it initialises (before the first HIP call of the process): a recorded training step
# forks into the integrator's stream and up to three gradient streams beside the main chain; with the defaults (4 hardware
# queues, 4 graph streams) independent branches of a replayed hipGraph are mapped onto the same queue and run one after the other
# (WaveEq step, same box: 1.79 -> 1.68 ms with 8 / 8).  Explicit settings in the environment win.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
_os.environ.setdefault('DEBUG_HIP_FORCE_GRAPH_QUEUES', '8')

__version__ = '0.2.0'
