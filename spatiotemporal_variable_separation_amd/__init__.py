"""MI355X-native (gfx950) implementation of the var_sep training hot path.

Package layout: `csrc/` holds the HIP kernels and the C ABI (`include/varsep_hip.h`), `_lib.py` binds it with
ctypes, `ops.py`/`functional.py` wrap it for torch tensors and autograd, and `networks/`, `train.py`, `options.py`,
`main.py` mirror the reference's Python surface (`var_sep.*`) for this path.
"""
__version__ = '0.1.0'
