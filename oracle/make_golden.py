"""Generate tests/golden/*.npz by running the REFERENCE itself (imported read-only from /root/reference).

TEST INFRASTRUCTURE ONLY.  Runs in the build container (the reference never travels to the GPU box):

    python -m oracle.make_golden            # writes tests/golden/<config>.npz, prints a summary
    python -m oracle.make_golden full       # additionally the five BASELINE configurations at full size (checksum fixtures)

For every config in `oracle.golden_configs.CONFIGS` the script
  1. builds the four networks with the reference's own factories and fills them with the RNG-free
     weights of `oracle.detdata.det_fill` (so no state dict needs to be committed),
  2. runs ONE real optimisation step through the reference's `var_sep.train.train()` (one epoch over a
     one-batch loader) and records parameters / BN buffers after the Adam step,
  3. replays the same step through the reference's `ae_loss` / `zero_order_loss` / `get_forecast` to record
     the four loss terms, the forecasts, codes and every parameter gradient, and checks that this replay
     lands on exactly the parameters of (2),
  4. runs the oracle (`oracle.cpu_ref`) on the same inputs and requires it to agree with the reference
     (bitwise on this container; the tolerance used by the committed test is 1e-5 relative),
  5. writes the vectors: small tensors in full, large ones as (sum, L2, 16 samples) checksums.
"""
import os
import sys
import tempfile

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = os.environ.get('VARSEP_REFERENCE', '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import cpu_ref                                    # noqa: E402
from oracle.detdata import det_fill, det_uniform, checksum    # noqa: E402
from oracle.golden_configs import CONFIGS, FULL_CONFIGS, fill_net, make_batch, FULL_LIMIT   # noqa: E402


def _reference_modules():
    sys.path.insert(0, REF)
    import var_sep.networks.factory as rf
    import var_sep.networks.model as rm
    import var_sep.networks.utils as ru
    import var_sep.train as rt
    return rf, rm, ru, rt


def build_reference(cfg, rf, rm, ru):
    shape = list(cfg['shape'])
    arch = cfg['architecture']
    dec_arch = cfg.get('decoder_architecture') or arch
    if cfg.get('no_s'):
        Es = ru.ConstantS(return_value=1, code_size=cfg['code_size_t'])
    else:
        Es = rf.get_encoder(arch, shape, cfg['code_size_s'], cfg['enc_hidden_size'], cfg.get('enc_n_layers', 3),
                            cfg['nt_cond'], 'normal', 0.02)
    Et = rf.get_encoder(arch, shape, cfg['code_size_t'], cfg['enc_hidden_size'], cfg.get('enc_n_layers', 3),
                        cfg['nt_cond'], 'normal', 0.02)
    dec = rf.get_decoder(dec_arch, shape, cfg['code_size_t'], cfg['code_size_s'], cfg.get('last_activation'),
                         cfg['dec_hidden_size'], cfg.get('dec_n_layers', 3), cfg.get('mixing', 'concat'),
                         cfg.get('skipco', False), 'normal', 0.02)
    res = rf.get_resnet(cfg['code_size_t'], cfg.get('n_blocks', 1), cfg.get('res_hidden_size', 512), 'orthogonal',
                        1.41, arch == 'encoderSST')
    net = rm.SeparableNetwork(Es, Et, res, dec, cfg['nt_cond'], cfg.get('skipco', False))
    return net


def reference_losses(cfg, net, cond, target, rt):
    """train.py:120-149 driven through the reference's own functions."""
    import torch.nn.functional as F
    nt_cond, nt_pred, offset, skipco = cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], cfg.get('skipco', False)
    lam = cfg['lambdas']
    ae, s_new, s_old = rt.ae_loss(cond, target, net, nt_cond, offset, skipco)
    zero = rt.zero_order_loss(s_old, s_new, skipco)
    full = torch.cat([cond, target], dim=1)
    forecasts, t_codes, s_code, _ = net.get_forecast(cond, nt_pred + offset, init_s_code=s_old)
    pred = F.mse_loss(forecasts, full[:, (nt_cond if offset == 0 else 0):])
    if cfg.get('average_tloss'):
        t_reg = 0.5 * (t_codes[:, 0].pow(2).view(full.shape[0], -1)).mean()
    else:
        t_reg = 0.5 * torch.sum(t_codes[:, 0].pow(2), dim=1).mean()
    lamb_t = 0 if cfg.get('no_s') else lam['t']
    total = 0
    total += lam['ae'] * ae
    total += lam['s'] * zero
    total += lam['pred'] * pred
    total += lamb_t * t_reg
    return total, dict(ae=ae, zero=zero, pred=pred, t_reg=t_reg), forecasts, t_codes, s_code


def pack(out, key, t):
    t = t.detach()
    if t.numel() <= FULL_LIMIT:
        out[key] = t.numpy().copy()
    else:
        out['cs:' + key] = checksum(t)


def run_config(name, cfg, mods):
    rf, rm, ru, rt = mods
    torch.manual_seed(0)
    cond, target = make_batch(cfg)
    seed = cfg.get('np_seed', 1234)
    adam = dict(lr=cfg.get('lr', 4e-4), betas=(0.9, 0.99))
    lam = cfg['lambdas']

    # (2) the real training loop, one step
    net_a = fill_net(build_reference(cfg, rf, rm, ru), cfg)
    opt = torch.optim.Adam(net_a.parameters(), **adam)
    np.random.seed(seed)
    with tempfile.TemporaryDirectory() as tmp:
        rt.train(tmp, [(cond, target)], torch.device('cpu'), net_a, opt, None, False, False, 1, lam['ae'], lam['s'],
                 lam['t'], lam['pred'], cfg['offset'], cfg['nt_cond'], cfg['nt_pred'], bool(cfg.get('no_s')),
                 cfg.get('skipco', False), None, bool(cfg.get('average_tloss')))

    # (3) replay through the reference's functions
    net_b = fill_net(build_reference(cfg, rf, rm, ru), cfg)
    net_b.train()
    opt_b = torch.optim.Adam(net_b.parameters(), **adam)
    np.random.seed(seed)
    total, terms, forecasts, t_codes, s_code = reference_losses(cfg, net_b, cond, target, rt)
    np.random.seed(seed)
    hi = cond.size(1) + target.size(1) + (0 if cfg['offset'] == 0 else 1)
    t_random = int(np.random.randint(cfg['nt_cond'], hi))
    opt_b.zero_grad()
    total.backward()
    grads = {k: p.grad.clone() for k, p in net_b.named_parameters() if p.grad is not None}    # ResNet18.bn_out is never used
    opt_b.step()
    sa, sb = net_a.state_dict(), net_b.state_dict()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), f'{name}: replay of train.py diverges from train() at {k}'

    # (4) the oracle on the same inputs
    ocfg = dict(cfg)
    net_o = cpu_ref.build_sep_net(ocfg)
    fill_net(net_o, cfg)
    missing = set(net_o.state_dict()) ^ set(build_reference(cfg, rf, rm, ru).state_dict())
    assert not missing, f'{name}: state_dict keys differ: {sorted(missing)[:5]}'
    net_o.train()
    opt_o = torch.optim.Adam(net_o.parameters(), **adam)
    lamb_t = 0 if cfg.get('no_s') else lam['t']
    o_total, o_terms, o_fore, o_tc = cpu_ref.training_losses(
        cond, target, net_o, cfg['nt_cond'], cfg['nt_pred'], cfg['offset'], cfg.get('skipco', False), lam['ae'],
        lam['s'], lamb_t, lam['pred'], average_tloss=bool(cfg.get('average_tloss')), t_random=t_random)
    opt_o.zero_grad()
    o_total.backward()
    worst = 0.0
    for k, p in net_o.named_parameters():
        if k not in grads:
            assert p.grad is None, k
            continue
        g = grads[k]
        worst = max(worst, ((p.grad - g).abs().max() / (g.abs().max() + 1e-30)).item())
    opt_o.step()
    so = net_o.state_dict()
    exact = all(torch.equal(so[k], sa[k]) for k in sa) and torch.equal(o_fore, forecasts)
    assert worst < 1e-6 and (o_total - total).abs().item() <= 1e-6 * abs(total.item()), (name, worst)

    # (5) vectors
    out = {'t_random': np.int64(t_random), 'total': np.float64(total.item())}
    for k, v in terms.items():
        out['loss:' + k] = np.float64(v.item())
    pack(out, 'forecasts', forecasts)
    pack(out, 't_codes', t_codes)
    pack(out, 's_code', s_code[0] if isinstance(s_code, tuple) else s_code)
    for k, g in grads.items():
        pack(out, 'grad:' + k, g)
    for k, v in sa.items():
        pack(out, 'after:' + k, v.float() if v.dtype != torch.float32 else v)
    path = os.path.join(ROOT, 'tests', 'golden', name + '.npz')
    np.savez_compressed(path, **out)
    nparam = sum(p.numel() for p in net_a.parameters())
    print(f'{name:22s} params={nparam:9d} t_random={t_random:2d} total={total.item():.6f} '
          f'oracle_vs_ref grad_relmax={worst:.1e} bit_exact={exact}  -> {os.path.getsize(path) / 1024:.0f} KiB')


def main():
    torch.set_num_threads(8)
    mods = _reference_modules()
    only = sys.argv[1:]
    for name, cfg in CONFIGS.items():
        if only and name not in only:
            continue
        run_config(name, cfg, mods)
    # the BASELINE configurations at full size (minutes of CPU time): only when asked for by name, or with `full`
    for name, cfg in FULL_CONFIGS.items():
        if name in only or 'full' in only:
            run_config(name, cfg, mods)


if __name__ == '__main__':
    main()
