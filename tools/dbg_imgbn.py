import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from spatiotemporal_variable_separation_amd import ops
torch.manual_seed(0)
for (B, Cin, Cout) in [(8, 64, 512), (8, 512, 512), (8, 512, 64)]:
    dtype = torch.bfloat16
    x = (torch.randn(B, Cin, 16, 16) * 0.5).to(dtype).cuda()
    w = (torch.randn(Cout, Cin, 3, 3) * 0.05).cuda()
    bias = torch.randn(Cout).cuda()
    gamma, beta = torch.ones(Cout).cuda(), torch.zeros(Cout).cuda()
    wp = ops.conv3_img16_pack_weight(w, dtype, False)
    ref = F.conv2d(x.float(), w.to(dtype).float(), bias, padding=1)
    slabs = ops.conv3_img16(x, wp, Cout)
    z0 = ops.slab_sum(slabs, bias, dtype)
    y1, z1, m1, i1 = ops.conv3_img16_bn_fwd(x, wp, bias, gamma, beta, 'none', torch.float32, Cout)
    torch.cuda.synchronize()
    print((B, Cin, Cout), 'two-launch vs torch', (z0.float() - ref).abs().max().item(), 'one-launch vs torch', (z1.float() - ref).abs().max().item(),
          'err flag', ops.rollout_exchange_error(x.device))
    d = (z1.float() - ref).abs()
    bad = (d > 0.05).nonzero()
    print(' bad count', bad.shape[0], 'of', d.numel(), 'first', bad[:6].tolist())
    # which (b, c, px) pattern is wrong
    if bad.shape[0]:
        print(' bad channels', sorted(set(bad[:, 1].tolist()))[:40], 'bad maps', sorted(set(bad[:, 0].tolist())))
    mu = ref.to(dtype).float().mean(dim=(0, 2, 3))
    print(' mean err', (m1.flatten() - mu).abs().max().item())
