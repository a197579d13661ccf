"""Evaluation metrics of the reference's test scripts on the device (reference: var_sep/test/utils.py:19-24 `_ssim_wrapper`,
var_sep/utils/ssim.py:81-149 `ssim_loss`, var_sep/test/mnist/test.py:136-142 mse / psnr / ssim per sample)."""
import torch

from .. import ops


def ssim_loss(input, target, max_val, filter_size=11, k1=0.01, k2=0.03, sigma=1.5, kernel=None, size_average=None, reduce=None,
              reduction='mean'):
    """Same arguments as the reference's `ssim_loss`.  reduction='none' returns the PER-PLANE mean of the SSIM map, shape
    [N, C, 1, 1] (every caller of the reference averages the map over the window positions at once, test/utils.py:24); the map
    itself is never materialised on this path.  A custom `kernel` or `filter_size != 11` is not supported."""
    if input.size() != target.size():
        raise ValueError('Expected input size ({}) to match target size ({}).'.format(input.size(0), target.size(0)))
    if kernel is not None or filter_size != 11:
        raise NotImplementedError('the fused metric kernel implements the 11 x 11 Gaussian window of the evaluation scripts')
    if size_average is not None or reduce is not None:
        reduction = 'mean' if (size_average is None or size_average) and (reduce is None or reduce) else ('sum' if reduce is None or reduce else 'none')
    while input.dim() < 4:
        input, target = input.unsqueeze(0), target.unsqueeze(0)
    if input.dim() != 4:
        raise ValueError('Expected 2, 3, or 4 dimensions (got {})'.format(input.dim()))
    _, ssim = ops.frame_metrics(input, target, max_val=max_val, k1=k1, k2=k2, sigma=sigma)
    if reduction == 'none':
        return ssim[:, :, None, None]
    # mean / sum over the whole map == mean over planes of the per-plane means (all planes have the same number of windows)
    n_win = (input.shape[-2] - 10) * (input.shape[-1] - 10)
    return ssim.mean() if reduction == 'mean' else ssim.sum() * n_win


def _ssim_wrapper(pred, gt):
    """[B, nt, C] mean SSIM per frame and channel (test/utils.py:19-24)."""
    _, ssim = ops.frame_metrics(pred, gt, max_val=1.0)
    return ssim


def frame_metrics(pred, target):
    """{'mse', 'psnr', 'ssim'} per sample as test/mnist/test.py:136-142 computes them from [B, nt, C, H, W] predictions."""
    mse, ssim = ops.frame_metrics(pred, target, max_val=1.0)
    return {'mse': mse.mean(2).mean(1), 'psnr': (10 * torch.log10(1 / mse)).mean(2).mean(1), 'ssim': ssim.mean(2).mean(1)}
