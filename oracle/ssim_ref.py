"""CPU restatement of the reference's evaluation metrics.  TEST INFRASTRUCTURE ONLY (imported by tests/ only).

Follows var_sep/utils/ssim.py:81-111 (`_fspecial_gaussian`, `_ssim`: five depthwise "valid" conv2d with an 11 x 11 Gaussian window,
sigma 1.5, k1 = 0.01, k2 = 0.03), var_sep/test/utils.py:19-24 (`_ssim_wrapper`: SSIM map averaged per frame and channel) and
var_sep/test/mnist/test.py:136-142 (mse / psnr / ssim per sample).  Pinned against the reference's own functions by
tests/golden/frame_metrics.npz (oracle/make_golden_metrics.py)."""
import torch
import torch.nn.functional as F


def gaussian_window(size, channel, sigma):
    # ssim.py:81-89: softmax over the 2-D grid of -(x^2 + y^2) / (2 sigma^2)
    coords = torch.tensor([(x - (size - 1.) / 2.) for x in range(size)])
    coords = -coords ** 2 / (2. * sigma ** 2)
    grid = (coords.view(1, -1) + coords.view(-1, 1)).view(1, -1).softmax(-1)
    return grid.view(1, 1, size, size).expand(channel, 1, size, size).contiguous()


def ssim_map(x, y, max_val=1.0, k1=0.01, k2=0.03, size=11, sigma=1.5):
    # ssim.py:92-111
    c = x.shape[1]
    w = gaussian_window(size, c, sigma).to(x)
    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    mu1, mu2 = F.conv2d(x, w, groups=c), F.conv2d(y, w, groups=c)
    mu1_sq, mu2_sq, mu12 = mu1 ** 2, mu2 ** 2, mu1 * mu2
    s1 = F.conv2d(x * x, w, groups=c) - mu1_sq
    s2 = F.conv2d(y * y, w, groups=c) - mu2_sq
    s12 = F.conv2d(x * y, w, groups=c) - mu12
    v1, v2 = 2 * s12 + c2, s1 + s2 + c2
    return ((2 * mu12 + c1) * v1) / ((mu1_sq + mu2_sq + c1) * v2)


def ssim_wrapper(pred, gt):
    # test/utils.py:19-24
    b, nt = pred.shape[0], pred.shape[1]
    img = pred.shape[2:]
    m = ssim_map(pred.reshape(b * nt, *img), gt.reshape(b * nt, *img), max_val=1.)
    return m.mean(dim=[2, 3]).view(b, nt, img[0])


def frame_metrics(pred, target):
    # test/mnist/test.py:136-142
    mse = torch.mean(F.mse_loss(pred, target, reduction='none'), dim=[3, 4])
    return {'mse': mse.mean(2).mean(1), 'psnr': (10 * torch.log10(1 / mse)).mean(2).mean(1), 'ssim': ssim_wrapper(pred, target).mean(2).mean(1),
            'mse_plane': mse, 'ssim_plane': ssim_wrapper(pred, target)}
