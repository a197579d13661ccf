"""Host time to ENQUEUE one replayed step against the GPU time to EXECUTE it (WaveEq, BASELINE configs[1], bf16): is the replay loop host-bound?

    python tools/host_vs_gpu.py [workload] [steps]

Prints the host time per `GraphedStep.step()` call with the device never waited for (the queue runs ahead) and the wall time per step of the
same loop including the final synchronisation.  Host time well below the step time = the GPU never waits for the host."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spatiotemporal_variable_separation_amd  # noqa: E402
spatiotemporal_variable_separation_amd.configure_single_gpu_queues()
import numpy as np          # noqa: E402
import torch                # noqa: E402


def main():
    from spatiotemporal_variable_separation_amd import functional as VF
    from spatiotemporal_variable_separation_amd.configs import BASELINE_CONFIGS
    from spatiotemporal_variable_separation_amd.data.synthetic import synthetic_batch
    from spatiotemporal_variable_separation_amd.networks.factory import build_sep_net
    from spatiotemporal_variable_separation_amd.optim import Adam
    from spatiotemporal_variable_separation_amd.train import GraphedStep, enable_fused_update, enable_update_in_backward
    name = sys.argv[1] if len(sys.argv) > 1 else 'waveeq'
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    cfg = dict(BASELINE_CONFIGS[name])
    dev = torch.device('cuda', 0)
    torch.manual_seed(1234)
    np.random.seed(1234)
    net = build_sep_net(cfg).to(dev)
    net.train()
    opt = Adam(net.parameters(), lr=4e-4, betas=(0.9, 0.99))
    cond, target = synthetic_batch(cfg['data'], cfg['batch'], cfg['nt_cond'], cfg['nt_pred'], device=dev, seed=1234)
    lam = cfg['lambdas']
    VF.set_precision('bf16')
    enable_update_in_backward(opt, net, None, scaler=None)
    enable_fused_update(opt, net, None, None)
    VF.fold_repeated_gradients(True)
    g = GraphedStep(net, opt, cond, target, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], (lam['ae'], lam['s'], lam['t'], lam['pred']),
                    bool(cfg.get('average_tloss')), warmup=3)
    for _ in range(10):
        g.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('%s: host %.1f us per step() call (enqueue only), %.1f us per step with the final synchronisation; GPU idle at the end of the loop: %.1f ms'
          % (name, (t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6, (t2 - t1) * 1e3))


if __name__ == '__main__':
    main()
