import sys, torch
sys.path.insert(0, '.')
import os
os.environ['VS_IMG_BN_SPLITS'] = '1,2,8'
from spatiotemporal_variable_separation_amd import ops
torch.manual_seed(0)
for dtype in (torch.bfloat16, torch.float16):
    for (B, Cin, Cout) in [(4, 64, 128), (4, 128, 128), (4, 128, 64), (8, 512, 512)]:
        x = (torch.randn(B, Cin, 16, 16) * 0.5).to(dtype).cuda()
        w = (torch.randn(Cout, Cin, 3, 3) * 0.05).cuda()
        bias = torch.randn(Cout).cuda()
        gamma, beta = (1 + 0.3 * torch.randn(Cout)).cuda(), (0.2 * torch.randn(Cout)).cuda()
        wp = ops.conv3_img16_pack_weight(w, dtype, False)
        slabs = ops.conv3_img16(x, wp, Cout)
        y0, z0, m0, i0 = ops.bn_train_fwd_small_slabs(slabs, bias, dtype, gamma, beta, 'leaky_relu', dtype, None, None, 0.1, 1e-5)
        y1, z1, m1, i1 = ops.conv3_img16_bn_fwd(x, wp, bias, gamma, beta, 'leaky_relu', dtype, Cout)
        torch.cuda.synchronize()
        print(dtype, (B, Cin, Cout), 'splits', ops._lib.load_library().vs_conv3_img16_splits(B, Cin, Cout), 'z equal', torch.equal(z0, z1),
              'mean rel', ((m1 - m0).abs().max() / m0.abs().max()).item(), 'invstd rel', ((i1 - i0).abs() / i0).max().item(),
              'y flips', (y1 != y0).float().mean().item(), 'y max diff', (y1.float() - y0.float()).abs().max().item())
