"""Writes tests/golden/lowp_bounds.json: per 16-bit step test case, the rounding-point emulation's distance to a second, equally valid
evaluation of itself (step_util.lowp_noise_floor: every parameter moved by 2e-7 relative, the size of fp32 summation-order noise).

    python tests/make_lowp_bounds.py [reduced|full|init|all]        (CPU only; `full` takes ~15 min on 8 cores)

The GPU tests bound HIP-vs-emulation distances by max(stated floor, 3 x these COMMITTED constants), never above 0.5 (step_util.noise_bound):
round 3 evaluated the self-distance live and relaxed to 1.0 / 1.5 where it saturated, so a regression that doubled a gradient error on the
deep BatchNorm stacks still passed.  Nothing of the product runs here: the emulation interprets the oracle's module tree
(oracle/bf16_emu.emulate_bf16)."""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from oracle.golden_configs import CONFIGS, FULL_CONFIGS  # noqa: E402
from golden_util import load_golden  # noqa: E402
from step_util import lowp_case_key, lowp_noise_floor  # noqa: E402

CONV_CONFIGS = ['dcgan_tiny', 'dcgan_skip_mul', 'vgg32_tiny', 'vgg64_skip', 'sst_skip', 'sst_noskip']
LOWP_BATCH = {'vgg32_tiny': 16, 'vgg64_skip': 16, 'sst_skip': 16, 'sst_noskip': 16}
REDUCED = [(n, 'bf16', None) for n in CONV_CONFIGS] + [(n, 'fp16', 256.0) for n in ('dcgan_tiny', 'dcgan_skip_mul', 'vgg32_tiny', 'sst_skip')]
FULL = [('full_mnist_b128', 'bf16', None), ('full_taxibj', 'bf16', None), ('full_sst', 'fp16', 1024.0), ('full_sst', 'bf16', None)]
# round 5: the same workloads on weights with the statistics of the reference's init_net (oracle.detdata.det_init_fill)
FULL_INIT = [('full_mnist_b128_init', 'bf16', None), ('full_taxibj_init', 'bf16', None), ('full_sst_init', 'fp16', 1024.0), ('full_sst_init', 'bf16', None)]


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    path = os.path.join(HERE, 'golden', 'lowp_bounds.json')
    out = json.load(open(path)) if os.path.exists(path) else {}
    out['_note'] = ('emulation self-distances (relative L2; gradients: worst tensor per sub-network on the scale max(tensor norm, 1e-3 of the '
                    'whole gradient)) written by tests/make_lowp_bounds.py; tests use max(floor, 3 x value) capped at 0.5')
    cases = []
    if what in ('reduced', 'all'):
        cases += [(n, dict(CONFIGS[n], B=LOWP_BATCH.get(n, CONFIGS[n]['B'])), p, ls) for n, p, ls in REDUCED]
    if what in ('full', 'all'):
        cases += [(n, FULL_CONFIGS[n], p, ls) for n, p, ls in FULL]
    if what in ('init', 'all'):
        cases += [(n, FULL_CONFIGS[n], p, ls) for n, p, ls in FULL_INIT]
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    for name, cfg, prec, ls in cases:
        t_random = int(load_golden(name)['t_random'])
        noise = lowp_noise_floor(cfg, t_random, prec, loss_scale=ls)
        key = lowp_case_key(name, cfg, prec, ls)
        out[key] = {k: float('%.4g' % v) for k, v in noise.items()}
        print(key, {k: '%.2e' % v for k, v in noise.items() if not k.startswith('loss')}, flush=True)
        json.dump(out, open(path, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
