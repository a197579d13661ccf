"""Generate tests/golden/frame_metrics.npz from the REFERENCE's evaluation metrics (imported read-only from /root/reference).

TEST INFRASTRUCTURE ONLY; runs in the build container:    python -m oracle.make_golden_metrics

Inputs are RNG-free (oracle.detdata.det_uniform): "predictions" = smooth blobs + noise in [0, 1], "targets" = the same frames blended with other noise, shapes
[B, nt, C, H, W] for the Moving-MNIST (1 x 64 x 64), TaxiBJ (2 x 32 x 32) and chairs (3 x 64 x 64) frame sizes.  Outputs: the reference's
`_ssim_wrapper` (var_sep/test/utils.py:19-24 -> utils/ssim.py) per (sample, frame, channel) and the mse / psnr / ssim per sample of
var_sep/test/mnist/test.py:136-142.  The oracle (oracle/ssim_ref.py) must reproduce them before the file is written."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
REF = os.environ.get('VARSEP_REFERENCE', '/root/reference')

from oracle import ssim_ref                      # noqa: E402
from oracle.detdata import det_uniform           # noqa: E402

CASES = {'mnist': (3, 4, 1, 64, 64), 'taxibj': (2, 3, 2, 32, 32), 'chairs': (2, 2, 3, 64, 64), 'small': (2, 2, 1, 12, 17)}


def make_pair(shape, salt):
    pred = det_uniform(shape, salt)
    # smooth it a little so that the SSIM is not dominated by noise, keep [0, 1]
    B, T, C, H, W = shape
    p = F.avg_pool2d(pred.view(-1, 1, H, W), 3, 1, 1).view(shape)
    pred = (0.5 * pred + 0.5 * p).clamp(0, 1)
    target = (0.85 * pred + 0.15 * det_uniform(shape, salt + 1)).clamp(0, 1)           # correlated: SSIM well inside (0, 1)
    return pred.contiguous(), target.contiguous()


def main():
    sys.path.insert(0, REF)
    from var_sep.test.utils import _ssim_wrapper
    out = {}
    for i, (name, shape) in enumerate(CASES.items()):
        pred, target = make_pair(shape, 100 + 10 * i)
        ssim = _ssim_wrapper(pred, target)                                              # reference
        mse = torch.mean(F.mse_loss(pred, target, reduction='none'), dim=[3, 4])        # test/mnist/test.py:138
        psnr = (10 * torch.log10(1 / mse)).mean(2).mean(1)
        o = ssim_ref.frame_metrics(pred, target)
        assert torch.allclose(o['ssim_plane'], ssim, rtol=1e-6, atol=1e-7) and torch.allclose(o['mse_plane'], mse, rtol=1e-6, atol=1e-9), name
        assert torch.allclose(o['psnr'], psnr, rtol=1e-6)
        out[name + ':ssim'] = ssim.numpy()
        out[name + ':mse'] = mse.numpy()
        out[name + ':psnr'] = psnr.numpy()
        out[name + ':ssim_sample'] = ssim.mean(2).mean(1).numpy()
        print(f'{name:8s} {shape}  ssim {ssim.mean().item():.5f}  mse {mse.mean().item():.5f}  oracle == reference')
    path = os.path.join(ROOT, 'tests', 'golden', 'frame_metrics.npz')
    np.savez_compressed(path, **out)
    print('->', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
