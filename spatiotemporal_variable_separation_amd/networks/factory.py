"""String -> module factories (reference: networks/factory.py:25-87); same names, arguments and asserts."""
import numpy as np

from .mlp_encdec import MLPEncoder, MLPDecoder
from .resnet import MLPResnet
from .utils import init_net


def _conv():
    from . import conv          # imported lazily: the conv family pulls in the conv kernels
    return conv


def get_encoder(nn_type, shape, output_size, hidden_size, n_layers, nt_cond, init_type, init_gain):
    nc = shape[0]
    dim = shape[-1]
    if nn_type == 'dcgan':
        assert dim == 64
        encoder = _conv().DCGAN64Encoder(nc * nt_cond, output_size, hidden_size)
    elif nn_type == 'vgg':
        assert dim in [32, 64]
        encoder = _conv().VGG64Encoder(nc * nt_cond, output_size, hidden_size, vgg32=dim == 32)
    elif nn_type == 'encoderSST':
        encoder = _conv().EncoderSST(nc * nt_cond, output_size)
    elif nn_type == 'mlp':
        input_size = int(nt_cond * np.prod(np.array(shape)))
        encoder = MLPEncoder(input_size, hidden_size, output_size, n_layers)
    elif nn_type == 'resnet':
        encoder = _conv().ResNet18(output_size, nc * nt_cond)
    else:
        raise ValueError(f'unknown encoder architecture `{nn_type}`')
    init_net(encoder, init_type=init_type, init_gain=init_gain)
    return encoder


def get_decoder(nn_type, shape, code_size_t, code_size_s, last_activation, hidden_size, n_layers, mixing, skipco,
                init_type, init_gain):
    assert not skipco or nn_type in ['dcgan', 'vgg', 'decoderSST']
    if mixing == 'mul':
        assert code_size_t == code_size_s
        input_size = code_size_t
    else:
        input_size = code_size_t + code_size_s
    nc = shape[0]
    dim = shape[-1]
    if nn_type == 'dcgan':
        assert dim == 64
        decoder = _conv().DCGAN64Decoder(nc, input_size, hidden_size, skipco, last_activation, mixing)
    elif nn_type == 'vgg':
        assert dim in [32, 64]
        decoder = _conv().VGG64Decoder(nc, input_size, hidden_size, skipco, last_activation, mixing, vgg32=dim == 32)
    elif nn_type == 'mlp':
        decoder = MLPDecoder(input_size, hidden_size, shape, n_layers, last_activation, mixing)
    elif nn_type == 'decoderSST':
        assert mixing == 'concat'
        if skipco:
            decoder = _conv().DecoderSST_Skip(input_size, nc, last_activation)
        else:
            decoder = _conv().DecoderSST(input_size, nc, last_activation)
    else:
        raise ValueError(f'unknown decoder architecture `{nn_type}`')
    init_net(decoder, init_type=init_type, init_gain=init_gain)
    return decoder


def get_resnet(latent_size, n_blocks, hidden_size, init_type, gain_res, fully_conv=False):
    if fully_conv:
        resnet = _conv().ConvResnet(latent_size, n_blocks=n_blocks, nf=hidden_size)
    else:
        resnet = MLPResnet(latent_size, n_blocks, hidden_size)
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
