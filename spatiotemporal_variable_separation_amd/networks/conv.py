"""Convolutional encoders / decoders on the HIP conv blocks (reference: networks/conv.py:41-426).

The module trees are the reference's (so `state_dict` keys match and `init_net`'s class-name dispatch works), but
the containers are only parameter holders: `run_layers` walks them and issues one fused HIP block per
conv -> [BatchNorm2d] -> [activation] group, plus pool / upsample / flatten+linear ops.
"""
import os

import torch
import torch.nn as nn

from .. import functional as VF
from .utils import activation_factory, activation_name

_ACT_TYPES = (nn.ReLU, nn.LeakyReLU, nn.ELU, nn.Sigmoid, nn.Tanh)


def make_conv_block(conv, activation, bn=True):
    """conv -> [BatchNorm2d] -> [activation] as an nn.Sequential (conv.py:41-60)."""
    modules = [conv]
    if bn:
        modules.append(nn.BatchNorm2d(conv.out_channels))
    if activation != 'none':
        modules.append(activation_factory(activation))
    return nn.Sequential(*modules)


def _flatten_modules(module, out):
    if isinstance(module, (nn.Sequential, nn.ModuleList)):
        for m in module:
            _flatten_modules(m, out)
    else:
        out.append(module)
    return out


def run_layers(module, h, final_act='none', final_fp32=False, groups=1):
    """Execute a (nested) Sequential of reference layer objects on the HIP path.

    `final_act` is an activation the caller applies right after the last layer (decoder `last_activation`); it is fused
    into the last conv block when that block has no activation of its own.  `final_fp32` makes the last block write
    fp32 (module outputs are fp32 in every precision mode).  `groups` > 1: dim 0 of `h` holds that many reference CALLS
    stacked (e.g. the decoder calls of all rollout steps); every BatchNorm then keeps per-call statistics and updates
    its running estimates call by call, so the batched execution is the reference's arithmetic."""
    layers = [m for m in _flatten_modules(module, []) if not isinstance(m, nn.Identity)]
    # group: conv [bn] [act]
    blocks, i = [], 0
    while i < len(layers):
        m = layers[i]
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            bn = act = None
            j = i + 1
            if j < len(layers) and isinstance(layers[j], nn.BatchNorm2d):
                bn = layers[j]
                j += 1
            if j < len(layers) and isinstance(layers[j], _ACT_TYPES):
                act = layers[j]
                j += 1
            blocks.append(('conv', m, bn, act))
            i = j
        else:
            blocks.append(('other', m, None, None))
            i += 1
    for gi, (kind, m, bn, act) in enumerate(blocks):
        last = gi == len(blocks) - 1
        if kind == 'conv':
            act_name = activation_name(act) if act is not None else 'none'
            extra = None
            if last and final_act not in ('none', None):
                if act_name == 'none':
                    act_name = final_act
                else:
                    extra = final_act
            assert m.kernel_size[0] == m.kernel_size[1] and m.stride[0] == m.stride[1] and m.padding[0] == m.padding[1]
            training = bn.training if bn is not None else False
            if bn is not None and training:
                VF.count_bn_calls(bn, groups)          # += groups: one per call, like nn.BatchNorm2d
            if (bn is not None and not training and not torch.is_grad_enabled() and bn.track_running_stats and bn.affine
                    and os.environ.get('VARSEP_FOLD_BN_EVAL', '1') == '1'):
                # inference: the BatchNorm is a fixed affine map of the convolution's output -> folded into weight and bias, one kernel
                wf, bf = VF.folded_conv_bn(m, bn)
                cfg = (isinstance(m, nn.ConvTranspose2d), m.stride[0], m.padding[0], False, act_name, False, 0.1, 1e-5,
                       bool(final_fp32 and last and extra is None), groups)
                h = VF.ConvBlock.apply(h, wf, bf, None, None, None, None, cfg)
            else:
                cfg = (isinstance(m, nn.ConvTranspose2d), m.stride[0], m.padding[0], bn is not None, act_name, training,
                       bn.momentum if bn is not None else 0.1, bn.eps if bn is not None else 1e-5,
                       bool(final_fp32 and last and extra is None), groups)
                h = VF.ConvBlock.apply(h, m.weight, m.bias, bn.weight if bn is not None else None,
                                       bn.bias if bn is not None else None, bn.running_mean if bn is not None else None,
                                       bn.running_var if bn is not None else None, cfg)
            if extra is not None:
                h = VF.Activation.apply(h.float() if final_fp32 else h, extra)
        elif isinstance(m, nn.MaxPool2d):
            h = VF.MaxPool2.apply(h)
        elif isinstance(m, nn.Upsample):
            h = VF.Upsample2.apply(h)
        elif isinstance(m, nn.Flatten):
            h = h.reshape(h.shape[0], -1)
        elif isinstance(m, nn.Linear):
            h = VF.mlp_chain(h, [m], out_act=final_act if last else 'none')
        else:
            raise NotImplementedError(f'layer {type(m).__name__} has no HIP implementation')
    if final_fp32 and h.dtype != torch.float32:
        h = h.float()
    return h


def _fold_time(x):
    """[B, nt, C, H, W] -> [B, nt*C, H, W] (conv.py:90): the temporal window becomes channels."""
    return x.reshape(x.size(0), -1, x.size(3), x.size(4))


class BaseEncoder(nn.Module):
    """Encoder forward (conv.py:81-99): stages collect skips, `last_op` produces the flat code."""

    def __init__(self, nh):
        super().__init__()
        self.nh = nh

    call_groups = True          # forward(..., groups=g): dim 0 holds g reference calls, every BatchNorm keeps per-call statistics

    def forward(self, x, return_skip=False, groups=1):
        h = _fold_time(x)
        skips = []
        for layer in self.conv:
            h = run_layers(layer, h, groups=groups)
            skips.append(h)
        h = run_layers(self.last_op, h, final_fp32=True, groups=groups).view(-1, self.nh)
        if return_skip:
            return h, skips[::-1]
        return h


class DCGAN64Encoder(BaseEncoder):
    """conv.py:102-124: four k4 s2 p1 convolutions (no BN on the first), then Flatten + Linear."""

    def __init__(self, nc, nh, nf):
        super().__init__(nh)
        chans = [nc, nf, nf * 2, nf * 4, nf * 8]
        self.conv = nn.ModuleList([
            make_conv_block(nn.Conv2d(chans[i], chans[i + 1], 4, 2, 1), activation='leaky_relu', bn=(i > 0))
            for i in range(4)])
        self.last_op = nn.Sequential(nn.Flatten(), nn.Linear(nf * 8 * 4 * 4, nh))


def _c3(cin, cout, activation='leaky_relu', bn=True):
    return make_conv_block(nn.Conv2d(cin, cout, 3, 1, 1), activation=activation, bn=bn)


def _pool():
    return nn.MaxPool2d(kernel_size=2, stride=2, padding=0)


def _up():
    return nn.Upsample(scale_factor=2, mode='nearest')


class VGG64Encoder(BaseEncoder):
    """conv.py:127-171: VGG stages (2,2,3,3 convolutions) with 2x2 max-pooling, k4 valid conv + BN as last op."""

    def __init__(self, nc, nh, nf, vgg32=False):
        super().__init__(nh)
        widths = [(nc, [nf, nf]), (nf, [nf * 2] * 2), (nf * 2, [nf * 4] * 3), (nf * 4, [nf * 8] * 3)]
        stages = []
        for si, (cin, couts) in enumerate(widths):
            mods = [] if si == 0 else [_pool()]
            for cout in couts:
                mods.append(_c3(cin, cout))
                cin = cout
            stages.append(nn.Sequential(*mods))
        self.conv = nn.ModuleList(stages)
        self.last_op = nn.Sequential(_pool() if not vgg32 else nn.Identity(),
                                     make_conv_block(nn.Conv2d(nf * 8, nh, 4, 1, 0), activation='none'))


class BaseDecoder(nn.Module):
    """Decoder forward (conv.py:207-230): mix codes, 1x1 -> 4x4 up-convolution, stages with optional skip concat."""

    def __init__(self, ny, skip, last_activation, mixing):
        super().__init__()
        self.ny = ny
        self.skip = skip
        self.mixing = mixing
        self.last_activation = activation_factory(last_activation)

    def forward(self, z1, z2, skip=None, groups=1):
        assert skip is None and not self.skip or self.skip and skip is not None
        z = torch.cat([z1, z2], dim=1) if self.mixing == 'concat' else z1 * z2
        h = run_layers(self.first_upconv, z.view(*z.shape, 1, 1), groups=groups)
        n_stage = len(self.conv)
        for i, layer in enumerate(self.conv):
            if skip is not None:
                h = torch.cat([h, skip[i].to(h.dtype)], 1)
            if i == n_stage - 1:
                h = run_layers(layer, h, final_act=activation_name(self.last_activation), final_fp32=True, groups=groups)
            else:
                h = run_layers(layer, h, groups=groups)
        return h

    def decode_sequence(self, z1, t_codes, skip=None):
        """All n decoder calls of a rollout (model.py:74-83) as ONE batch of n*B samples ordered [step][sample]; every
        BatchNorm keeps per-step statistics (`groups = n`), so this is the reference's arithmetic with n x larger GEMMs.

        z1 [B, Cs], t_codes [B, n, Ct] -> frames [B, n, C, H, W] (a transposed view of the [n, B, ...] result)."""
        B, n = t_codes.shape[0], t_codes.shape[1]
        z2 = t_codes.transpose(0, 1).reshape(n * B, -1)
        z1e = z1.repeat(n, 1)
        skips = None if skip is None else [s.repeat(n, 1, 1, 1) for s in skip]
        out = self.forward(z1e, z2, skip=skips, groups=n)
        return out.view(n, B, *out.shape[1:]).transpose(0, 1)


class DCGAN64Decoder(BaseDecoder):
    """conv.py:233-264."""

    def __init__(self, nc, ny, nf, skip, last_activation, mixing):
        super().__init__(ny, skip, last_activation, mixing)
        coef = 2 if skip else 1
        self.first_upconv = make_conv_block(nn.ConvTranspose2d(ny, nf * 8, 4, 1, 0), activation='leaky_relu')
        self.conv = nn.ModuleList([
            make_conv_block(nn.ConvTranspose2d(nf * 8 * coef, nf * 4, 4, 2, 1), activation='leaky_relu'),
            make_conv_block(nn.ConvTranspose2d(nf * 4 * coef, nf * 2, 4, 2, 1), activation='leaky_relu'),
            make_conv_block(nn.ConvTranspose2d(nf * 2 * coef, nf, 4, 2, 1), activation='leaky_relu'),
            nn.ConvTranspose2d(nf * coef, nc, 4, 2, 1),
        ])


class VGG64Decoder(BaseDecoder):
    """conv.py:267-320."""

    def __init__(self, nc, ny, nf, skip, last_activation, mixing, vgg32=False):
        super().__init__(ny, skip, last_activation, mixing)
        coef = 2 if skip else 1
        self.first_upconv = nn.Sequential(
            make_conv_block(nn.ConvTranspose2d(ny, nf * 8, 4, 1, 0), activation='leaky_relu'),
            _up() if not vgg32 else nn.Identity())
        self.conv = nn.ModuleList([
            nn.Sequential(_c3(nf * 8 * coef, nf * 8), _c3(nf * 8, nf * 8), _c3(nf * 8, nf * 4), _up()),
            nn.Sequential(_c3(nf * 4 * coef, nf * 4), _c3(nf * 4, nf * 4), _c3(nf * 4, nf * 2), _up()),
            nn.Sequential(_c3(nf * 2 * coef, nf * 2), _c3(nf * 2, nf), _up()),
            nn.Sequential(_c3(nf * coef, nf), nn.ConvTranspose2d(nf, nc, 3, 1, 1)),
        ])


class EncoderSST(nn.Module):
    """conv.py:323-356: spatial (16x16) code and three skips."""

    def __init__(self, in_c, out_c):
        super().__init__()
        self.conv1 = nn.Sequential(_c3(in_c, 64), _c3(64, 64))
        self.conv2 = nn.Sequential(_pool(), _c3(64, 128), _c3(128, 128))
        self.conv3 = nn.Sequential(_pool(), _c3(128, 256), _c3(256, 256), _c3(256, 256))
        self.conv4 = nn.Sequential(_c3(256, 512), _c3(512, out_c), _c3(out_c, out_c, activation='none', bn=False))

    call_groups = True

    def forward(self, x, return_skip=False, groups=1):
        h1 = run_layers(self.conv1, _fold_time(x), groups=groups)
        h2 = run_layers(self.conv2, h1, groups=groups)
        h3 = run_layers(self.conv3, h2, groups=groups)
        h4 = run_layers(self.conv4, h3, final_fp32=True, groups=groups)
        if return_skip:
            return h4, [h3, h2, h1]
        return h4


def _decode_sequence_spatial(dec, s_code, t_codes, skip):
    """decode_sequence for the SST decoders: s_code [B, Cs, h, w], t_codes [B, n, Ct, h, w] -> [B, n, C, H, W]."""
    B, n = t_codes.shape[0], t_codes.shape[1]
    z2 = t_codes.transpose(0, 1).reshape(n * B, *t_codes.shape[2:])
    # the spatial code and the skips of the B sequences are shared by the n frame calls: they are NOT repeated here -- the concatenations
    # inside forward read them by sequence index (functional.cat_bcast)
    out = dec.forward(s_code, z2, skip, groups=n, rep=n)
    return out.view(n, B, *out.shape[1:]).transpose(0, 1)


def _cat_shared(a, x, rep, dtype=None):
    """torch.cat([a, x], 1) where `a` belongs to B sequences and `x` to rep x B frame calls stacked frame-major (rep = 1: a plain cat)."""
    if rep == 1:
        return torch.cat([a if dtype is None else a.to(dtype), x], dim=1)
    return VF.cat_bcast(a, x, rep, dtype or x.dtype)


class DecoderSST_Skip(nn.Module):
    """conv.py:359-396."""

    def __init__(self, in_c, out_c, out_f):
        super().__init__()
        self.conv1 = nn.Sequential(_c3(in_c, 256), _c3(256, 256), _c3(256, 128))
        self.conv2 = nn.Sequential(_c3(256 + 128, 128), _c3(128, 64), _c3(64, 64), _up())
        self.conv3 = nn.Sequential(_c3(128 + 64, 128), _c3(128, 64), _c3(64, 64), _up())
        self.conv4 = nn.Sequential(_c3(64 * 2, 64), _c3(64, 64), _c3(64, out_c))
        self.out_f = activation_factory(out_f)

    def forward(self, s_code, t_code, skip, groups=1, rep=1):
        h3, h2, h1 = skip
        out = run_layers(self.conv1, _cat_shared(s_code, t_code, rep), groups=groups)
        out = run_layers(self.conv2, _cat_shared(h3, out, rep, out.dtype), groups=groups)
        out = run_layers(self.conv3, _cat_shared(h2, out, rep, out.dtype), groups=groups)
        return run_layers(self.conv4, _cat_shared(h1, out, rep, out.dtype), final_act=activation_name(self.out_f),
                          final_fp32=True, groups=groups)

    def decode_sequence(self, s_code, t_codes, skip):
        return _decode_sequence_spatial(self, s_code, t_codes, skip)


class DecoderSST(nn.Module):
    """conv.py:399-426."""

    def __init__(self, in_c, out_c, out_f):
        super().__init__()
        self.conv1 = nn.Sequential(_c3(in_c, 256), _c3(256, 256), _c3(256, 128), _up())
        self.conv2 = nn.Sequential(_c3(128, 128), _c3(128, 128), _c3(128, 64), _up())
        self.conv3 = nn.Sequential(_c3(64, 64), _c3(64, out_c))
        self.out_f = activation_factory(out_f)

    def forward(self, s_code, t_code, skip=None, groups=1, rep=1):
        x = run_layers(self.conv1, _cat_shared(s_code, t_code, rep), groups=groups)
        x = run_layers(self.conv2, x, groups=groups)
        return run_layers(self.conv3, x, final_act=activation_name(self.out_f), final_fp32=True, groups=groups)

    def decode_sequence(self, s_code, t_codes, skip=None):
        return _decode_sequence_spatial(self, s_code, t_codes, skip)


class ConvResBlock(nn.Module):
    """resnet.py:53-70."""

    def __init__(self, in_c, out_c, nf=64):
        super().__init__()
        self.conv = nn.Sequential(_c3(in_c, nf), _c3(nf, nf), _c3(nf, out_c, activation='none'))
        self.up = _c3(in_c, out_c, activation='none') if in_c != out_c else nn.Identity()

    def forward(self, x, return_alias=False):
        """`return_alias=True` (not a reference argument; `ConvResnet` asks for it on behalf of `model._roll`): a third result, the block's
        output a SECOND time as its own autograd output (None when the block does not run as one fused node) -- see below."""
        if not getattr(self, '_marked', False):
            # the integrator applies this block once per predicted frame: its weight gradients are batched over the step's calls
            VF.mark_repeated([m.weight for m in self.modules() if isinstance(m, nn.Conv2d)])
            self._marked = True
        if isinstance(self.up, nn.Identity) and torch.is_grad_enabled():
            # a few 16x16 maps in a 16-bit compute type (the SST integrator): the whole block as 6 launches forward / 7 backward
            layers = [m for m in _flatten_modules(self.conv, []) if not isinstance(m, nn.Identity)]
            convs = [m for m in layers if isinstance(m, nn.Conv2d)]
            if len(convs) == 3 and len(layers) <= 8:
                bns, acts = [], []
                for c in convs:
                    i = layers.index(c)
                    bn = layers[i + 1] if i + 1 < len(layers) and isinstance(layers[i + 1], nn.BatchNorm2d) else None
                    j = i + (2 if bn is not None else 1)
                    act = layers[j] if j < len(layers) and isinstance(layers[j], _ACT_TYPES) else None
                    bns.append(bn)
                    acts.append(activation_name(act) if act is not None else 'none')
                if acts[2] == 'none' and VF.conv_res_block_fusable(x, convs, bns):
                    args, cfg = [], []
                    for c, bn, act in zip(convs, bns, acts):
                        VF.count_bn_calls(bn, 1)
                        args += [c.weight, c.bias, bn.weight, bn.bias]
                        cfg.append((bn.running_mean, bn.running_var, bn.momentum, bn.eps, act))
                    prev = getattr(x, '_vs16', None)
                    xnew, x16, residual, alias = VF.ConvResBlockFn.apply(x, prev[0] if prev is not None and prev[1] == x._version else None, *args,
                                                                         tuple(cfg))
                    # the next block's convolution operand (same values, already in the compute type), valid while xnew is not written to
                    xnew._vs16 = (x16, xnew._version)
                    # `alias` = the same output as a second autograd output: what a caller that ALSO keeps the code (model._roll stacks it for the
                    # decoder) should keep, so that the two gradients join inside the block's backward launches instead of in an add launch.
                    # It travels through the RETURN VALUE only: it is a view of xnew, so an attribute of xnew that holds it would be a
                    # reference cycle through the C++ base pointer that the garbage collector cannot see (one leaked map per block call).
                    return (xnew, residual, alias) if return_alias else (xnew, residual)
        residual = run_layers(self.conv, x, final_fp32=True)
        skip = x if isinstance(self.up, nn.Identity) else run_layers(self.up, x, final_fp32=True)
        return (skip + residual, residual, None) if return_alias else (skip + residual, residual)


class ConvResnet(nn.Module):
    """resnet.py:73-88: the conv integrator used with `--architecture encoderSST`."""

    def __init__(self, in_c, n_blocks=1, nf=64):
        super().__init__()
        self.n_blocks = n_blocks
        self.resblock_modules = nn.ModuleList([ConvResBlock(in_c, in_c, nf=nf) for _ in range(n_blocks)])

    supports_alias = True        # forward(..., return_alias=True): see ConvResBlock.forward

    def forward(self, x, return_res=True, return_alias=False):
        residuals, alias = [], None
        last = len(self.resblock_modules) - 1
        for i, blk in enumerate(self.resblock_modules):
            if return_alias and i == last:
                x, residual, alias = blk(x, return_alias=True)
            else:
                x, residual = blk(x)
            residuals.append(residual)
        if return_alias:
            return x, residuals, alias
        return (x, residuals) if return_res else x


# ------------------------------------------------------------------------------------------------ chairs: ResNet18 encoder
class BasicBlock(nn.Module):
    """conv.py:440-468: conv3x3(stride) + BN + ReLU, conv3x3 + BN, [1x1 stride-s conv + BN on the shortcut], add, ReLU.
    The attribute names are the reference's (`conv1 bn1 conv2 bn2 downsample`) so that state dicts interchange."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x, groups=1):
        out = run_layers(nn.Sequential(self.conv1, self.bn1, self.relu), x, groups=groups)
        out = run_layers(nn.Sequential(self.conv2, self.bn2), out, groups=groups)
        residual = run_layers(self.downsample, x, groups=groups) if self.downsample is not None else x
        if residual.dtype != out.dtype:
            residual = residual.to(out.dtype)
        return VF.Activation.apply(out + residual, 'relu')


class ResNet18(nn.Module):
    """conv.py:509-564 (after DrNet): k5 s2 p3 stem + BN + ReLU, 3x3 s2 max-pool, four stages of two BasicBlocks (64, 128, 256,
    512 planes; stride 2 from the second stage on), 3x3 valid `conv_out` to the code, optional output activation.  `bn_out` is
    constructed -- and therefore initialised, saved and loaded -- but never applied, exactly like the reference."""

    def __init__(self, pose_dim, nc=3, out_f=None):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(nc, 64, kernel_size=5, stride=2, padding=3)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, 2)
        self.layer2 = self._make_layer(128, 2, stride=2)
        self.layer3 = self._make_layer(256, 2, stride=2)
        self.layer4 = self._make_layer(512, 2, stride=2)
        self.conv_out = nn.Conv2d(512, pose_dim, kernel_size=3)
        self.bn_out = nn.BatchNorm2d(pose_dim)
        self.out_function = activation_factory(out_f)

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes, kernel_size=1, stride=stride), nn.BatchNorm2d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        layers += [BasicBlock(planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    call_groups = True

    def forward(self, x, return_skip=False, groups=1):
        h = x.reshape(x.size(0), -1, x.size(3), x.size(4))
        h = run_layers(nn.Sequential(self.conv1, self.bn1, self.relu), h, groups=groups)
        h = VF.MaxPool3s2.apply(h)
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
            for block in stage:
                h = block(h, groups=groups)
        h = run_layers(nn.Sequential(self.conv_out), h, final_act=activation_name(self.out_function), final_fp32=True)
        return h.reshape(len(h), -1)

