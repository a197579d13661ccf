"""Parity-test configurations shared by oracle/make_golden.py and tests/.  TEST INFRASTRUCTURE ONLY.

Each entry is a reduced-width instance of one architecture family of SURVEY.md section 8 (same layer
types, strides, BN placement, skip wiring, mixing and loss options as the README recipes; smaller channel
counts where the reference lets them be chosen).  The SST encoder/decoder have hard-coded widths
(conv.py:323-426), so those two configs are full width and their large tensors are pinned by checksum.
"""
from oracle.detdata import det_uniform

FULL_LIMIT = 4096      # tensors up to this many elements are stored in full, larger ones as checksums

_L = dict(ae=10.0, s=45.0, t=0.001, pred=45.0)          # options.py:95-102 defaults

CONFIGS = {
    # Moving-MNIST recipe shape (options.py defaults), reduced widths
    'dcgan_tiny': dict(architecture='dcgan', shape=[1, 64, 64], nt_cond=3, nt_pred=4, offset=3, B=3,
                       code_size_s=6, code_size_t=5, enc_hidden_size=4, dec_hidden_size=4, res_hidden_size=8,
                       n_blocks=1, mixing='concat', last_activation='sigmoid', skipco=False, lambdas=_L, salt=11),
    # skip connections + multiplicative mixing + offset 0 (forecast-only supervision)
    'dcgan_skip_mul': dict(architecture='dcgan', shape=[1, 64, 64], nt_cond=2, nt_pred=3, offset=0, B=2,
                           code_size_s=6, code_size_t=6, enc_hidden_size=4, dec_hidden_size=4, res_hidden_size=8,
                           n_blocks=2, mixing='mul', last_activation='sigmoid', skipco=True, lambdas=_L, salt=12),
    # TaxiBJ recipe (README.md:82): vgg32, 2 channels, no last activation
    'vgg32_tiny': dict(architecture='vgg', shape=[2, 32, 32], nt_cond=2, nt_pred=2, offset=2, B=3,
                       code_size_s=6, code_size_t=5, enc_hidden_size=4, dec_hidden_size=4, res_hidden_size=8,
                       n_blocks=1, mixing='concat', last_activation=None, skipco=False,
                       lambdas=dict(ae=45.0, s=0.0001, t=0.001, pred=45.0), salt=13),
    'vgg64_skip': dict(architecture='vgg', shape=[1, 64, 64], nt_cond=2, nt_pred=2, offset=2, B=2,
                       code_size_s=5, code_size_t=4, enc_hidden_size=4, dec_hidden_size=4, res_hidden_size=8,
                       n_blocks=1, mixing='concat', last_activation='sigmoid', skipco=True, lambdas=_L, salt=14),
    # WaveEq recipe (README.md:90): mlp, mul mixing, 3 residual blocks, 4-layer decoder
    'mlp_mul': dict(architecture='mlp', shape=[1, 8, 8], nt_cond=3, nt_pred=5, offset=3, B=5,
                    code_size_s=8, code_size_t=8, enc_hidden_size=24, dec_hidden_size=24, dec_n_layers=4,
                    res_hidden_size=16, n_blocks=3, mixing='mul', last_activation='sigmoid', skipco=False,
                    lambdas=dict(ae=1.0, s=45.0, t=0.001, pred=45.0), salt=15),
    # WaveEq-100 recipe (README.md:94): 2-D data shape [1, n_points], concat mixing
    'mlp_concat_partial': dict(architecture='mlp', shape=[1, 10], nt_cond=2, nt_pred=3, offset=2, B=4,
                               code_size_s=6, code_size_t=4, enc_hidden_size=20, dec_hidden_size=12,
                               res_hidden_size=16, n_blocks=1, mixing='concat', last_activation='sigmoid',
                               skipco=False, lambdas=_L, salt=16),
    # --no_s (main.py:124-129): constant S, lamb_t forced to 0 (train.py:99-101)
    'mlp_no_s': dict(architecture='mlp', shape=[1, 6, 6], nt_cond=2, nt_pred=3, offset=2, B=3, no_s=True,
                     code_size_s=5, code_size_t=5, enc_hidden_size=16, dec_hidden_size=16, res_hidden_size=8,
                     n_blocks=1, mixing='mul', last_activation='sigmoid', skipco=False, lambdas=_L, salt=17),
    # SST recipe (README.md:86): conv integrator, skip decoder, offset 0, averaged t-loss
    'sst_skip': dict(architecture='encoderSST', decoder_architecture='decoderSST', shape=[1, 64, 64], nt_cond=2,
                     nt_pred=2, offset=0, B=2, code_size_s=12, code_size_t=8, enc_hidden_size=64,
                     dec_hidden_size=64, res_hidden_size=16, n_blocks=2, mixing='concat', last_activation=None,
                     skipco=True, average_tloss=True, data_range='normal',
                     lambdas=dict(ae=1.0, s=100.0, t=5e-6, pred=45.0), salt=18),
    'sst_noskip': dict(architecture='encoderSST', decoder_architecture='decoderSST', shape=[1, 64, 64], nt_cond=2,
                       nt_pred=2, offset=0, B=2, code_size_s=10, code_size_t=6, enc_hidden_size=64,
                       dec_hidden_size=64, res_hidden_size=8, n_blocks=1, mixing='concat', last_activation=None,
                       skipco=False, average_tloss=True, data_range='normal',
                       lambdas=dict(ae=1.0, s=100.0, t=5e-6, pred=45.0), salt=19),
    # chairs recipe (README.md:78): ResNet18 encoders (fixed widths 64..512, conv.py:509-564) + DCGAN decoder, 3-channel frames;
    # ~22 M encoder parameters, pinned by checksum
    'chairs_resnet': dict(architecture='resnet', decoder_architecture='dcgan', shape=[3, 64, 64], nt_cond=2, nt_pred=2, offset=2, B=2,
                          code_size_s=6, code_size_t=5, enc_hidden_size=4, dec_hidden_size=4, res_hidden_size=8, n_blocks=1,
                          mixing='concat', last_activation='sigmoid', skipco=False, lambdas=dict(ae=1.0, s=1.0, t=0.001, pred=45.0),
                          salt=19),
}


# The five BASELINE.json configurations at FULL width and batch (README recipes; same dictionaries as the product's
# configs.BASELINE_CONFIGS, restated here because test infrastructure does not import the product).  Their tensors are far above
# FULL_LIMIT, so the fixtures `tests/golden/full_<name>.npz` hold checksums (sum, L2, 16 samples) of forecasts, codes, every
# gradient and every post-Adam parameter of ONE reference training step.
# `res_scale`: det_fill's weights have gain ~1.2 per layer, which makes the residual integrator double the latent code at every
# block; over the 14-40 steps x 1-3 blocks of the full recipes the code grows to 1e12 and the step becomes numerically meaningless
# (the fp32 oracle itself then sits 1e-3 / O(1) away from its own fp64 evaluation on forecasts / gradients).  The integrator's
# parameters are therefore scaled by `res_scale` after det_fill (`fill_net` below; the reference's own init uses gain 0.71-1.41 on
# orthogonal matrices, i.e. a contraction of similar size).
FULL_CONFIGS = {
    'full_mnist_b16': dict(res_scale=0.3, architecture='dcgan', shape=[1, 64, 64], nt_cond=5, nt_pred=10, offset=5, B=16, code_size_s=128,
                           code_size_t=20, enc_hidden_size=64, dec_hidden_size=64, res_hidden_size=512, n_blocks=1,
                           mixing='concat', last_activation='sigmoid', skipco=False, lambdas=_L, salt=31),
    'full_waveeq': dict(res_scale=0.3, architecture='mlp', shape=[1, 64, 64], nt_cond=5, nt_pred=20, offset=5, B=128, code_size_s=32,
                        code_size_t=32, enc_hidden_size=1200, dec_hidden_size=1200, enc_n_layers=3, dec_n_layers=4,
                        res_hidden_size=512, n_blocks=3, mixing='mul', last_activation='sigmoid', skipco=False,
                        lambdas=dict(ae=1.0, s=45.0, t=0.001, pred=45.0), salt=32),
    'full_mnist_b128': dict(res_scale=0.3, architecture='dcgan', shape=[1, 64, 64], nt_cond=5, nt_pred=10, offset=5, B=128, code_size_s=128,
                            code_size_t=20, enc_hidden_size=64, dec_hidden_size=64, res_hidden_size=512, n_blocks=1,
                            mixing='concat', last_activation='sigmoid', skipco=False, lambdas=_L, salt=33),
    'full_taxibj': dict(res_scale=0.3, architecture='vgg', shape=[2, 32, 32], nt_cond=4, nt_pred=4, offset=4, B=100, code_size_s=128,
                        code_size_t=20, enc_hidden_size=64, dec_hidden_size=64, res_hidden_size=512, n_blocks=1,
                        mixing='concat', last_activation=None, skipco=False,
                        lambdas=dict(ae=45.0, s=0.0001, t=0.001, pred=45.0), salt=34),
    'full_sst': dict(res_scale=0.3, architecture='encoderSST', decoder_architecture='decoderSST', shape=[1, 64, 64], nt_cond=4, nt_pred=40,
                     offset=0, B=8, code_size_s=196, code_size_t=64, enc_hidden_size=64, dec_hidden_size=64,
                     res_hidden_size=512, n_blocks=2, mixing='concat', last_activation=None, skipco=True, average_tloss=True,
                     data_range='normal', lambdas=dict(ae=1.0, s=100.0, t=5e-6, pred=45.0), salt=35),
}


# Round 5: the three convolution workloads again with weights that have the statistics of the reference's OWN start of
# training (`init_net`, main.py:119-138: normal 0.02 for the encoders / decoder, orthogonal gain 1.41 for the integrator) instead of the
# hash fill above -- to tell whether the chaotic 16-bit gradients of the deep BatchNorm stacks are a property of `det_fill` (gain 1.2 per
# layer, BatchNorm weights 1 +- 0.2) or of the networks.  `fill='init'` -> oracle.detdata.det_init_fill (RNG-free).
FULL_CONFIGS['full_mnist_b128_init'] = dict(FULL_CONFIGS['full_mnist_b128'], fill='init', res_scale=None, salt=43)
FULL_CONFIGS['full_taxibj_init'] = dict(FULL_CONFIGS['full_taxibj'], fill='init', res_scale=None, salt=44)
FULL_CONFIGS['full_sst_init'] = dict(FULL_CONFIGS['full_sst'], fill='init', res_scale=None, salt=45)


def fill_net(net, cfg):
    """RNG-free weights of a parity configuration: det_fill, then the integrator scaled by cfg['res_scale'] when given (or, with
    cfg['fill'] == 'init', weights with the statistics of the reference's `init_net`).  Works on the oracle's, the reference's and the
    product's SeparableNetwork alike (all expose `.t_resnet`)."""
    import torch
    from oracle.detdata import det_fill, det_init_fill
    if cfg.get('fill') == 'init':
        return det_init_fill(net, salt=cfg['salt'])
    det_fill(net, salt=cfg['salt'])
    scale = cfg.get('res_scale')
    if scale:
        with torch.no_grad():
            for name, p in net.t_resnet.named_parameters():
                if p.dim() > 1:                          # weights only; biases / BatchNorm affine parameters keep their scale
                    p.mul_(scale)
    return net


def make_batch(cfg):
    """(cond, target) of the dataset's shape: U[0,1) frames, or roughly z-scored (U-0.5)*3 for SST."""
    B, shape = cfg['B'], list(cfg['shape'])
    cond = det_uniform([B, cfg['nt_cond']] + shape, cfg['salt'] * 31 + 1)
    target = det_uniform([B, cfg['nt_pred']] + shape, cfg['salt'] * 31 + 2)
    if cfg.get('data_range') == 'normal':
        cond, target = (cond - 0.5) * 3.0, (target - 0.5) * 3.0
    return cond, target
