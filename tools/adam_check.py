import os, sys, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatiotemporal_variable_separation_amd.optim import Adam
torch.manual_seed(0)
shapes = [(1200, 20480), (1200,), (1200, 20480), (4096, 1200), (512, 512), (32,)]
pa = [torch.nn.Parameter(torch.randn(s, device='cuda') * 0.1) for s in shapes]
pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
oa = Adam(pa, lr=4e-4, betas=(0.9, 0.99)); ob = torch.optim.Adam(pb, lr=4e-4, betas=(0.9, 0.99), fused=True)
for step in range(3):
    for a, b in zip(pa, pb):
        g = torch.randn_like(a) * 0.01
        a.grad = g.clone(); b.grad = g.clone()
    oa.step(); ob.step()
    print(step, [f'{(a - b).abs().max().item():.2e}' for a, b in zip(pa, pb)])
for name, o in (('hip', oa), ('torch fused', ob)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): o.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    n = sum(p.numel() for p in pa)
    print(name, f'{dt * 1e6:.1f} us/step, {n * 28 / dt / 1e12:.2f} TB/s')
