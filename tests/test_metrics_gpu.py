"""GPU: the evaluation metrics (vs_frame_metrics: per-plane MSE and mean SSIM in one launch) against vectors recorded from the
reference's `_ssim_wrapper` (var_sep/test/utils.py:19-24 -> utils/ssim.py:81-111) and metric lines (test/mnist/test.py:136-142)."""
import pytest
import torch

from golden_util import load_golden
from oracle.make_golden_metrics import CASES, make_pair

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', list(CASES))
def test_frame_metrics_match_reference_fixture(name):
    from spatiotemporal_variable_separation_amd.utils import metrics
    gold = load_golden('frame_metrics')
    i = list(CASES).index(name)
    pred, target = make_pair(CASES[name], 100 + 10 * i)
    pred, target = pred.cuda(), target.cuda()
    ssim = metrics._ssim_wrapper(pred, target)
    m = metrics.frame_metrics(pred, target)
    torch.cuda.synchronize()
    ref = torch.from_numpy(gold[name + ':ssim'])
    assert tuple(ssim.shape) == tuple(ref.shape)
    assert torch.allclose(ssim.cpu(), ref, rtol=2e-5, atol=2e-6), (ssim.cpu() - ref).abs().max().item()
    assert torch.allclose(m['psnr'].cpu(), torch.from_numpy(gold[name + ':psnr']), rtol=1e-5)
    assert torch.allclose(m['ssim'].cpu(), torch.from_numpy(gold[name + ':ssim_sample']), rtol=2e-5, atol=2e-6)
    assert torch.allclose(m['mse'].cpu(), torch.from_numpy(gold[name + ':mse']).mean(2).mean(1), rtol=1e-5)


def test_ssim_loss_surface_and_limits():
    from oracle import ssim_ref
    from spatiotemporal_variable_separation_amd.utils import metrics
    from spatiotemporal_variable_separation_amd._lib import VarsepHipError
    pred, target = make_pair((2, 1, 3, 40, 56), 7)
    x, y = pred[:, 0].cuda(), target[:, 0].cuda()
    ref = ssim_ref.ssim_map(pred[:, 0], target[:, 0])
    assert abs(metrics.ssim_loss(x, y, max_val=1.).item() - ref.mean().item()) < 2e-6
    per_plane = metrics.ssim_loss(x, y, max_val=1., reduction='none')
    assert torch.allclose(per_plane[:, :, 0, 0].cpu(), ref.mean(dim=[2, 3]), rtol=2e-5, atol=2e-6)
    assert abs(metrics.ssim_loss(x, x, max_val=1.).item() - 1.0) < 1e-6                  # identical images
    with pytest.raises(VarsepHipError):
        metrics.ssim_loss(torch.rand(1, 1, 256, 256).cuda(), torch.rand(1, 1, 256, 256).cuda(), max_val=1.)   # plane does not fit the LDS
    with pytest.raises(VarsepHipError):
        metrics.ssim_loss(torch.rand(1, 1, 64, 64), torch.rand(1, 1, 64, 64), max_val=1.)                      # CPU tensors: no fallback
