"""Phase stamps of the one-launch ConvResBlock layer (conv3_img16_bn_kernel built with -DVS_IMGBN_STAMP: VARSEP_HIPCC_FLAGS=-DVS_IMGBN_STAMP
python -c "from spatiotemporal_variable_separation_amd import _lib; _lib.build_library(force=True)"): workgroup 0's wall clock (100 MHz) at
0 start, 1 tile done, 2 reduce-scatter done, 3 local sums done, 4 all-gather done, 5 statistics combined, 6 end."""
import sys
import torch
sys.path.insert(0, '.')
from spatiotemporal_variable_separation_amd import ops

for (B, Cin, Cout) in [(8, 64, 512), (8, 512, 512), (8, 512, 64)]:
    dtype = torch.bfloat16
    x = (torch.randn(B, Cin, 16, 16) * 0.5).to(dtype).cuda()
    w = (torch.randn(Cout, Cin, 3, 3) * 0.05).cuda()
    bias = torch.randn(Cout).cuda()
    gamma, beta = torch.ones(Cout).cuda(), torch.zeros(Cout).cuda()
    skip = torch.randn(B, Cout, 16, 16).cuda()
    wp = ops.conv3_img16_pack_weight(w, dtype, False)
    for _ in range(3):
        ops.conv3_img16_bn_fwd(x, wp, bias, gamma, beta, 'leaky_relu', dtype, Cout)
    torch.cuda.synchronize()
    ws = ops._IMGBN[torch.cuda.current_device()]['ws']
    off = (256 + (1 << 20)) // 4 - 32
    st = ws[off:off + 32].view(torch.int64).tolist()
    print((B, Cin, Cout), 'fwd phases (us):', [round((st[k + 1] - st[k]) / 100.0, 2) for k in range(6)], 'total', round((st[6] - st[0]) / 100.0, 2))
