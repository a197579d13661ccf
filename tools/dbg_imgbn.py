import sys, torch, os
sys.path.insert(0, '.')
os.environ['VS_IMG_BN_SPLITS'] = '1,2,8'
from spatiotemporal_variable_separation_amd import ops
torch.manual_seed(0)
for dtype in (torch.bfloat16, torch.float16):
    B = 4
    chans = [(64, 128, 'leaky_relu'), (128, 128, 'leaky_relu'), (128, 64, 'none')]
    x = torch.randn(B, 64, 16, 16).cuda()
    hA = hB = x.to(dtype)
    for li, (ci, co, act) in enumerate(chans):
        w = (torch.randn(co, ci, 3, 3) * (1.0 / (3 * ci ** 0.5))).cuda()
        bias = (0.1 * torch.randn(co)).cuda()
        gamma, beta = torch.ones(co).cuda(), torch.zeros(co).cuda()
        wp = ops.conv3_img16_pack_weight(w, dtype, False)
        od = dtype if li < 2 else torch.float32
        kw = dict(skip=x, want16=True) if li == 2 else {}
        rA = ops.bn_train_fwd_small_slabs(ops.conv3_img16(hA, wp, co), bias, dtype, gamma, beta, act, od, None, None, 0.1, 1e-5, **kw)
        rB = ops.conv3_img16_bn_fwd(hB, wp, bias, gamma, beta, act, od, co, **kw)
        rC = ops.conv3_img16_bn_fwd(hA, wp, bias, gamma, beta, act, od, co, **kw)       # same input as the two-launch path
        torch.cuda.synchronize()
        print(dtype, li, 'same-input: z equal', torch.equal(rA[1], rC[1]), 'y rel', ((rA[0].float() - rC[0].float()).norm() / rA[0].float().norm()).item(),
              '| chained: z flips', (rA[1] != rB[1]).float().mean().item(), 'y rel', ((rA[0].float() - rB[0].float()).norm() / rA[0].float().norm()).item(),
              'y moved', ((rA[0].float() - rB[0].float()).abs() > 1e-5 * rA[0].float().abs().max()).float().mean().item())
        hA, hB = rA[0], rB[0]
