"""Phase stamps of the one-launch ConvResBlock layer (conv3_img16_bn_kernel built with -DVS_IMGBN_STAMP: VARSEP_HIPCC_FLAGS=-DVS_IMGBN_STAMP
python -c "from spatiotemporal_variable_separation_amd import _lib; _lib.build_library(force=True)"): workgroup 0's wall clock (100 MHz) at
0 start, 1 tile done, 2 reduce-scatter done, 3 local sums done, 4 all-gather done, 5 statistics combined, 6 end."""
import os
import sys
import torch
sys.path.insert(0, '.')
os.environ['VS_IMG_BN_SPLITS'] = '1,2,8'
from spatiotemporal_variable_separation_amd import ops


def stamps():
    torch.cuda.synchronize()
    ws = ops._IMGBN[torch.cuda.current_device()]['ws']
    off = (256 + (1 << 20)) // 4 - 32
    st = ws[off:off + 32].view(torch.int64).tolist()
    return [round((st[k + 1] - st[k]) / 100.0, 2) for k in range(6)], round((st[6] - st[0]) / 100.0, 2)


for (B, Cin, Cout) in [(8, 64, 512), (8, 512, 512), (8, 512, 64)]:
    dtype = torch.bfloat16
    x = (torch.randn(B, Cin, 16, 16) * 0.5).to(dtype).cuda()
    w = (torch.randn(Cout, Cin, 3, 3) * 0.05).cuda()
    bias = torch.randn(Cout).cuda()
    gamma, beta = torch.ones(Cout).cuda(), torch.zeros(Cout).cuda()
    wp = ops.conv3_img16_pack_weight(w, dtype, False)
    for _ in range(3):
        y, z, mean, invstd = ops.conv3_img16_bn_fwd(x, wp, bias, gamma, beta, 'leaky_relu', dtype, Cout)
    print((B, Cin, Cout), 'fwd phases (us):', *stamps())
    # backward layer with the same tile geometry: dz_next has Cin channels, this layer Cout
    w_up = (torch.randn(Cin, Cout, 3, 3) * 0.05).cuda()
    wf = ops.conv3_img16_pack_weight(w_up, dtype, True)
    dzn = torch.randn(B, Cin, 16, 16).to(dtype).cuda()
    for _ in range(3):
        ops.conv3_img16_bn_bwd(dzn, wf, Cout, z, mean, invstd, gamma, beta, 'leaky_relu')
    print((B, Cin, Cout), 'bwd phases (us):', *stamps())
    flush = torch.zeros(256 << 20, device='cuda')
    for _ in range(2):
        flush.add_(1.0)                          # 1 GiB through the caches: the operands below come from HBM
        ops.conv3_img16_bn_bwd(dzn, wf, Cout, z, mean, invstd, gamma, beta, 'leaky_relu')
    print((B, Cin, Cout), 'bwd phases, cold operands (us):', *stamps())
    for _ in range(2):
        flush.add_(1.0)
        ops.conv3_img16_bn_fwd(x, wp, bias, gamma, beta, 'leaky_relu', dtype, Cout)
    print((B, Cin, Cout), 'fwd phases, cold operands (us):', *stamps())
    del flush
