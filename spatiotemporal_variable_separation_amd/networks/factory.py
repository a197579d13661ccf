"""String -> module factories (reference: networks/factory.py:25-87); same names, arguments and asserts."""
import numpy as np

from .mlp_encdec import MLPEncoder, MLPDecoder
from .resnet import MLPResnet
from .utils import init_net


def _conv():
    from . import conv          # imported lazily: the conv family pulls in the conv kernels
    return conv


# One builder per architecture name; each checks the geometry the reference's classes assume and returns the un-initialised module.
# (conv families are imported lazily: they pull in the conv kernels.)
def _enc_dcgan(shape, code, hidden, n_layers, nt_cond):
    assert shape[-1] == 64
    return _conv().DCGAN64Encoder(shape[0] * nt_cond, code, hidden)


def _enc_vgg(shape, code, hidden, n_layers, nt_cond):
    assert shape[-1] in [32, 64]
    return _conv().VGG64Encoder(shape[0] * nt_cond, code, hidden, vgg32=shape[-1] == 32)


def _enc_resnet(shape, code, hidden, n_layers, nt_cond):
    return _conv().ResNet18(code, shape[0] * nt_cond)


def _enc_sst(shape, code, hidden, n_layers, nt_cond):
    return _conv().EncoderSST(shape[0] * nt_cond, code)


def _enc_mlp(shape, code, hidden, n_layers, nt_cond):
    return MLPEncoder(int(nt_cond * np.prod(np.array(shape))), hidden, code, n_layers)


_ENCODERS = {'dcgan': _enc_dcgan, 'vgg': _enc_vgg, 'resnet': _enc_resnet, 'encoderSST': _enc_sst, 'mlp': _enc_mlp}


def get_encoder(nn_type, shape, output_size, hidden_size, n_layers, nt_cond, init_type, init_gain):
    """factory.py:25-44 of the reference: same name, arguments and geometry asserts."""
    if nn_type not in _ENCODERS:
        raise ValueError(f'unknown encoder architecture `{nn_type}`')
    encoder = _ENCODERS[nn_type](shape, output_size, hidden_size, n_layers, nt_cond)
    init_net(encoder, init_type=init_type, init_gain=init_gain)
    return encoder


def _dec_dcgan(shape, zdim, act, hidden, n_layers, mixing, skipco):
    assert shape[-1] == 64
    return _conv().DCGAN64Decoder(shape[0], zdim, hidden, skipco, act, mixing)


def _dec_vgg(shape, zdim, act, hidden, n_layers, mixing, skipco):
    assert shape[-1] in [32, 64]
    return _conv().VGG64Decoder(shape[0], zdim, hidden, skipco, act, mixing, vgg32=shape[-1] == 32)


def _dec_mlp(shape, zdim, act, hidden, n_layers, mixing, skipco):
    return MLPDecoder(zdim, hidden, shape, n_layers, act, mixing)


def _dec_sst(shape, zdim, act, hidden, n_layers, mixing, skipco):
    assert mixing == 'concat'
    cls = _conv().DecoderSST_Skip if skipco else _conv().DecoderSST
    return cls(zdim, shape[0], act)


_DECODERS = {'dcgan': _dec_dcgan, 'vgg': _dec_vgg, 'mlp': _dec_mlp, 'decoderSST': _dec_sst}


def get_decoder(nn_type, shape, code_size_t, code_size_s, last_activation, hidden_size, n_layers, mixing, skipco,
                init_type, init_gain):
    """factory.py:47-76 of the reference: skip connections only for the conv decoders, `mul` mixing needs equal code sizes."""
    assert not skipco or nn_type in ['dcgan', 'vgg', 'decoderSST']
    if mixing == 'mul':
        assert code_size_t == code_size_s
    zdim = code_size_t if mixing == 'mul' else code_size_t + code_size_s
    if nn_type not in _DECODERS:
        raise ValueError(f'unknown decoder architecture `{nn_type}`')
    decoder = _DECODERS[nn_type](shape, zdim, last_activation, hidden_size, n_layers, mixing, skipco)
    init_net(decoder, init_type=init_type, init_gain=init_gain)
    return decoder


def get_resnet(latent_size, n_blocks, hidden_size, init_type, gain_res, fully_conv=False):
    """factory.py:79-87: the latent integrator, convolutional for the SST architecture."""
    resnet = _conv().ConvResnet(latent_size, n_blocks=n_blocks, nf=hidden_size) if fully_conv else MLPResnet(latent_size, n_blocks, hidden_size)
    init_net(resnet, init_type=init_type, init_gain=gain_res)
    return resnet


def build_sep_net(cfg):
    """Assemble the four networks from a dict of CLI-style options the way main.py:119-140 does."""
    from .model import SeparableNetwork
    from .utils import ConstantS
    shape = list(cfg['shape'])
    arch = cfg['architecture']
    dec_arch = cfg.get('decoder_architecture') or arch
    ie, ge = cfg.get('init_encoder', 'normal'), cfg.get('gain_encoder', 0.02)
    if cfg.get('no_s'):
        assert not cfg.get('skipco', False)
        cfg = dict(cfg, code_size_s=cfg['code_size_t'], mixing='mul')
        Es = ConstantS(return_value=1, code_size=cfg['code_size_s'])
    else:
        Es = get_encoder(arch, shape, cfg['code_size_s'], cfg['enc_hidden_size'], cfg.get('enc_n_layers', 3),
                         cfg['nt_cond'], ie, ge)
    Et = get_encoder(arch, shape, cfg['code_size_t'], cfg['enc_hidden_size'], cfg.get('enc_n_layers', 3),
                     cfg['nt_cond'], ie, ge)
    dec = get_decoder(dec_arch, shape, cfg['code_size_t'], cfg['code_size_s'], cfg.get('last_activation'),
                      cfg['dec_hidden_size'], cfg.get('dec_n_layers', 3), cfg.get('mixing', 'concat'),
                      cfg.get('skipco', False), ie, ge)
    res = get_resnet(cfg['code_size_t'], cfg.get('n_blocks', 1), cfg.get('res_hidden_size', 512),
                     cfg.get('init_resnet', 'orthogonal'), cfg.get('gain_resnet', 1.41), arch == 'encoderSST')
    return SeparableNetwork(Es, Et, res, dec, cfg['nt_cond'], cfg.get('skipco', False))
