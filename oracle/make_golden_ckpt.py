"""Generate tests/golden/ckpt_<config>/{ov_Et,ov_Es,decoder,t_resnet}.pt: checkpoints written by the REFERENCE's own `save`.

TEST INFRASTRUCTURE ONLY.  Runs in the build container (the reference never travels to the GPU box):

    python -m oracle.make_golden_ckpt

The reference saves whole modules (`var_sep/utils/helper.py:22-33`: `torch.save(sep_net.Et, ...)`), i.e. pickles that name classes of
the `var_sep` package, and its evaluation scripts load them back with `torch.load` (`test/utils.py:8-16`).  For the reduced-width
configs below the reference networks are filled with the RNG-free weights and BatchNorm statistics of `oracle.detdata.det_fill` --
the state the eval fixtures `tests/golden/eval_<config>.npz` were recorded with (`oracle/make_golden_eval.py`) -- and saved by the
reference's `save`.  The files are DATA (tensors + class paths); `tests/test_eval_gpu.py` loads them on the GPU box, where `var_sep`
does not exist, and must reproduce the eval fixtures.
"""
import os
import sys

import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle.golden_configs import CONFIGS, fill_net                   # noqa: E402
from oracle.make_golden import REF, _reference_modules, build_reference    # noqa: E402

CKPT_CONFIGS = ['mlp_mul', 'dcgan_tiny', 'dcgan_skip_mul', 'vgg32_tiny']      # small enough to commit (<= 1 MB per config)


def main():
    mods = _reference_modules()
    rf, rm, ru, rt = mods
    sys.path.insert(0, REF)
    from var_sep.utils.helper import save as reference_save
    for name in CKPT_CONFIGS:
        cfg = CONFIGS[name]
        net = fill_net(build_reference(cfg, rf, rm, ru), cfg).eval()
        out = os.path.join(ROOT, 'tests', 'golden', 'ckpt_' + name)
        os.makedirs(out, exist_ok=True)
        reference_save(out, net)
        size = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out))
        # the files really name the reference's classes
        blob = open(os.path.join(out, 'ov_Et.pt'), 'rb').read()
        assert b'var_sep.networks' in blob, name
        print(f'ckpt_{name:16s} {size / 1024:.0f} KiB  ({", ".join(sorted(os.listdir(out)))})')


if __name__ == '__main__':
    main()
