"""CPU oracle for the var_sep training hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (fp32, CPU) restatement of the reference's algorithm for the
path named in BASELINE.json / SURVEY.md section 8: encoders E_s/E_t, the residual latent
time-stepper, the decoder D, `get_forecast`, and the four training losses.  It exists so that
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg have something to
check the HIP path against on machines where `/root/reference` is absent.  Nothing under
`spatiotemporal_variable_separation_amd/` imports it; the product path never routes here.

Pinning: the reference holds no tests, golden vectors or fixtures for this path (SURVEY.md
section 4), so the oracle is pinned against outputs of the reference itself, imported in the
build container by `oracle/make_golden.py`; the resulting vectors live in `tests/golden/`
and `tests/test_oracle_golden.py` replays them against this file.

Every network is assembled from the small layer table below so that `state_dict()` keys and
tensor shapes are identical to the reference's modules (a reference-trained state dict loads
here and vice versa).  Reference locations (relative to /root/reference/var_sep):
  networks/conv.py:41-60     conv block = conv -> [BatchNorm2d] -> [activation]
  networks/conv.py:81-99     encoder forward (fold time into channels, collect skips)
  networks/conv.py:102-124   DCGAN encoder         networks/conv.py:127-171  VGG encoder
  networks/conv.py:207-230   decoder forward       networks/conv.py:233-264  DCGAN decoder
  networks/conv.py:267-320   VGG decoder           networks/conv.py:323-356  SST encoder
  networks/conv.py:359-396   SST skip decoder      networks/conv.py:399-426  SST decoder
  networks/mlp.py:24-75      MLP (activation BEFORE every non-first Linear)
  networks/mlp_encdec.py     MLP encoder / decoder
  networks/resnet.py:22-88   MLP / conv residual integrators
  networks/utils.py:21-109   ConstantS, activations, init_net
  networks/factory.py:25-87  string -> module factories
  networks/model.py:52-89    SeparableNetwork.get_forecast
  train.py:38-88,111-149     zero_order_loss, ae_loss, loss assembly
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- activations
def act(name):
    """networks/utils.py:50-72 -- in-place ReLU / LeakyReLU(0.2) / ELU, Sigmoid, Tanh, Identity."""
    table = {
        'relu': lambda: nn.ReLU(inplace=True),
        'leaky_relu': lambda: nn.LeakyReLU(0.2, inplace=True),
        'elu': lambda: nn.ELU(inplace=True),
        'sigmoid': nn.Sigmoid,
        'tanh': nn.Tanh,
        'identity': nn.Identity,
        None: nn.Identity,
    }
    if name not in table:
        raise ValueError(f'Activation function `{name}` not yet implemented')
    return table[name]()


def cba(conv, activation='leaky_relu', bn=True):
    """conv.py:41-60: Sequential(conv, [BatchNorm2d(out)], [activation])."""
    layers = [conv]
    if bn:
        layers.append(nn.BatchNorm2d(conv.out_channels))
    if activation != 'none':
        layers.append(act(activation))
    return nn.Sequential(*layers)


def c3(i, o, **kw):
    return cba(nn.Conv2d(i, o, 3, 1, 1), **kw)


def pool():
    return nn.MaxPool2d(kernel_size=2, stride=2, padding=0)


def up():
    return nn.Upsample(scale_factor=2, mode='nearest')


# --------------------------------------------------------------------------- conv encoders
class _FlatEncoder(nn.Module):
    """conv.py:81-99."""

    def __init__(self, nh):
        super().__init__()
        self.nh = nh

    def forward(self, x, return_skip=False):
        h = x.view(x.size(0), -1, x.size(3), x.size(4))
        skips = []
        for stage in self.conv:
            h = stage(h)
            skips.append(h)
        code = self.last_op(h).view(-1, self.nh)
        return (code, skips[::-1]) if return_skip else code


class DCGAN64Encoder(_FlatEncoder):
    """conv.py:102-124."""

    def __init__(self, nc, nh, nf):
        super().__init__(nh)
        widths = [nc, nf, 2 * nf, 4 * nf, 8 * nf]
        self.conv = nn.ModuleList(
            [cba(nn.Conv2d(widths[i], widths[i + 1], 4, 2, 1), bn=i > 0) for i in range(4)])
        self.last_op = nn.Sequential(nn.Flatten(), nn.Linear(8 * nf * 16, nh))


class VGG64Encoder(_FlatEncoder):
    """conv.py:127-171."""

    def __init__(self, nc, nh, nf, vgg32=False):
        super().__init__(nh)
        plan = [(nc, [nf, nf]), (nf, [2 * nf, 2 * nf]), (2 * nf, [4 * nf] * 3), (4 * nf, [8 * nf] * 3)]
        stages = []
        for si, (cin, outs) in enumerate(plan):
            mods = [pool()] if si > 0 else []
            for o in outs:
                mods.append(c3(cin, o))
                cin = o
            stages.append(nn.Sequential(*mods))
        self.conv = nn.ModuleList(stages)
        self.last_op = nn.Sequential(nn.Identity() if vgg32 else pool(),
                                     cba(nn.Conv2d(8 * nf, nh, 4, 1, 0), activation='none'))


class EncoderSST(nn.Module):
    """conv.py:323-356 (spatial 16x16 code, three skips)."""

    def __init__(self, in_c, out_c):
        super().__init__()
        self.conv1 = nn.Sequential(c3(in_c, 64), c3(64, 64))
        self.conv2 = nn.Sequential(pool(), c3(64, 128), c3(128, 128))
        self.conv3 = nn.Sequential(pool(), c3(128, 256), c3(256, 256), c3(256, 256))
        self.conv4 = nn.Sequential(c3(256, 512), c3(512, out_c), c3(out_c, out_c, activation='none', bn=False))

    def forward(self, x, return_skip=False):
        h1 = self.conv1(x.view(x.size(0), -1, x.size(3), x.size(4)))
        h2 = self.conv2(h1)
        h3 = self.conv3(h2)
        h4 = self.conv4(h3)
        return (h4, [h3, h2, h1]) if return_skip else h4


# --------------------------------------------------------------------------- conv decoders
class _FlatDecoder(nn.Module):
    """conv.py:207-230."""

    def __init__(self, ny, skip, last_activation, mixing):
        super().__init__()
        self.ny, self.skip, self.mixing = ny, skip, mixing
        self.last_activation = act(last_activation)

    def forward(self, z1, z2, skip=None):
        assert skip is None and not self.skip or self.skip and skip is not None
        z = torch.cat([z1, z2], dim=1) if self.mixing == 'concat' else z1 * z2
        h = self.first_upconv(z.view(*z.shape, 1, 1))
        for i, stage in enumerate(self.conv):
            if skip is not None:
                h = torch.cat([h, skip[i]], 1)
            h = stage(h)
        return self.last_activation(h)


class DCGAN64Decoder(_FlatDecoder):
    """conv.py:233-264."""

    def __init__(self, nc, ny, nf, skip, last_activation, mixing):
        super().__init__(ny, skip, last_activation, mixing)
        m = 2 if skip else 1
        self.first_upconv = cba(nn.ConvTranspose2d(ny, 8 * nf, 4, 1, 0))
        self.conv = nn.ModuleList([
            cba(nn.ConvTranspose2d(8 * nf * m, 4 * nf, 4, 2, 1)),
            cba(nn.ConvTranspose2d(4 * nf * m, 2 * nf, 4, 2, 1)),
            cba(nn.ConvTranspose2d(2 * nf * m, nf, 4, 2, 1)),
            nn.ConvTranspose2d(nf * m, nc, 4, 2, 1),
        ])


class VGG64Decoder(_FlatDecoder):
    """conv.py:267-320."""

    def __init__(self, nc, ny, nf, skip, last_activation, mixing, vgg32=False):
        super().__init__(ny, skip, last_activation, mixing)
        m = 2 if skip else 1
        self.first_upconv = nn.Sequential(cba(nn.ConvTranspose2d(ny, 8 * nf, 4, 1, 0)),
                                          nn.Identity() if vgg32 else up())
        self.conv = nn.ModuleList([
            nn.Sequential(c3(8 * nf * m, 8 * nf), c3(8 * nf, 8 * nf), c3(8 * nf, 4 * nf), up()),
            nn.Sequential(c3(4 * nf * m, 4 * nf), c3(4 * nf, 4 * nf), c3(4 * nf, 2 * nf), up()),
            nn.Sequential(c3(2 * nf * m, 2 * nf), c3(2 * nf, nf), up()),
            nn.Sequential(c3(nf * m, nf), nn.ConvTranspose2d(nf, nc, 3, 1, 1)),
        ])


class DecoderSST_Skip(nn.Module):
    """conv.py:359-396."""

    def __init__(self, in_c, out_c, out_f):
        super().__init__()
        self.conv1 = nn.Sequential(c3(in_c, 256), c3(256, 256), c3(256, 128))
        self.conv2 = nn.Sequential(c3(256 + 128, 128), c3(128, 64), c3(64, 64), up())
        self.conv3 = nn.Sequential(c3(128 + 64, 128), c3(128, 64), c3(64, 64), up())
        self.conv4 = nn.Sequential(c3(128, 64), c3(64, 64), c3(64, out_c))
        self.out_f = act(out_f)

    def forward(self, s_code, t_code, skip):
        h3, h2, h1 = skip
        out = self.conv1(torch.cat([s_code, t_code], dim=1))
        out = self.conv2(torch.cat([h3, out], dim=1))
        out = self.conv3(torch.cat([h2, out], dim=1))
        out = self.conv4(torch.cat([h1, out], dim=1))
        return self.out_f(out)


class DecoderSST(nn.Module):
    """conv.py:399-426."""

    def __init__(self, in_c, out_c, out_f):
        super().__init__()
        self.conv1 = nn.Sequential(c3(in_c, 256), c3(256, 256), c3(256, 128), up())
        self.conv2 = nn.Sequential(c3(128, 128), c3(128, 128), c3(128, 64), up())
        self.conv3 = nn.Sequential(c3(64, 64), c3(64, out_c))
        self.out_f = act(out_f)

    def forward(self, s_code, t_code, skip=None):
        return self.out_f(self.conv3(self.conv2(self.conv1(torch.cat([s_code, t_code], dim=1)))))


# --------------------------------------------------------------------------- MLP family
class MLP(nn.Module):
    """mlp.py:44-75: layer 0 is Sequential(Linear); layer i>0 is Sequential(act, Linear)."""

    def __init__(self, ninp, nhid, nout, nlayers, activation='relu'):
        super().__init__()
        assert nhid == 0 or nlayers > 1
        blocks = []
        for il in range(nlayers):
            lin = nn.Linear(ninp if il == 0 else nhid, nout if il == nlayers - 1 else nhid)
            blocks.append(nn.Sequential(lin) if il == 0 else nn.Sequential(act(activation), lin))
        self.module = nn.Sequential(*blocks)

    def forward(self, x):
        return self.module(x)


class MLPEncoder(nn.Module):
    """mlp_encdec.py:25-32."""

    def __init__(self, input_size, hidden_size, output_size, nlayers):
        super().__init__()
        self.mlp = MLP(input_size, hidden_size, output_size, nlayers)

    def forward(self, x, return_skip=False):
        return self.mlp(x.view(len(x), -1))


class MLPDecoder(nn.Module):
    """mlp_encdec.py:35-50."""

    def __init__(self, latent_size, hidden_size, output_shape, nlayers, last_activation, mixing):
        super().__init__()
        self.output_shape = list(output_shape)
        self.mixing = mixing
        self.mlp = MLP(latent_size, hidden_size, int(np.prod(np.array(output_shape))), nlayers)
        self.last_activation = act(last_activation)

    def forward(self, z1, z2, skip=None):
        z = torch.cat([z1, z2], dim=1) if self.mixing == 'concat' else z1 * z2
        return self.last_activation(self.mlp(z)).view([-1] + self.output_shape)


class MLPResBlock(nn.Module):
    """resnet.py:22-29."""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.mlp = MLP(input_size, hidden_size, input_size, 3)

    def forward(self, x):
        r = self.mlp(x)
        return x + r, r


class MLPResnet(nn.Module):
    """resnet.py:32-50."""

    def __init__(self, input_size, n_blocks, hidden_size):
        super().__init__()
        self.in_size, self.n_blocks = input_size, n_blocks
        self.blocks = nn.ModuleList([MLPResBlock(input_size, hidden_size) for _ in range(n_blocks)])

    def forward(self, x, return_res=True):
        res = []
        for blk in self.blocks:
            x, r = blk(x)
            res.append(r)
        return (x, res) if return_res else x


class ConvResBlock(nn.Module):
    """resnet.py:53-70."""

    def __init__(self, in_c, out_c, nf=64):
        super().__init__()
        self.conv = nn.Sequential(c3(in_c, nf), c3(nf, nf), c3(nf, out_c, activation='none'))
        self.up = nn.Identity() if in_c == out_c else c3(in_c, out_c, activation='none')

    def forward(self, x):
        r = self.conv(x)
        return self.up(x) + r, r


class ConvResnet(nn.Module):
    """resnet.py:73-88."""

    def __init__(self, in_c, n_blocks=1, nf=64):
        super().__init__()
        self.n_blocks = n_blocks
        self.resblock_modules = nn.ModuleList([ConvResBlock(in_c, in_c, nf=nf) for _ in range(n_blocks)])

    def forward(self, x, return_res=True):
        res = []
        for blk in self.resblock_modules:
            x, r = blk(x)
            res.append(r)
        return (x, res) if return_res else x


class ConstantS(nn.Module):
    """utils.py:21-29 (--no_s)."""

    def __init__(self, return_value=1, code_size=1):
        super().__init__()
        self.code_size, self.return_value = code_size, return_value

    def forward(self, x, return_skip=False):
        return torch.ones(len(x), self.code_size).to(x) * self.return_value


# --------------------------------------------------------------------------- init + factories
def init_net(net, init_type='normal', init_gain=0.02):
    """utils.py:75-109: class-name dispatch over Conv2d/ConvTranspose2d/Linear and BatchNorm2d."""
    def visit(m):
        kind = type(m).__name__
        if kind in ('Conv2d', 'ConvTranspose2d', 'Linear'):
            if init_type == 'normal':
                nn.init.normal_(m.weight.data, 0.0, init_gain)
            elif init_type == 'xavier':
                nn.init.xavier_normal_(m.weight.data, gain=init_gain)
            elif init_type == 'kaiming':
                nn.init.kaiming_normal_(m.weight.data, a=0, mode='fan_in')
            elif init_type == 'orthogonal':
                nn.init.orthogonal_(m.weight.data, gain=init_gain)
            else:
                raise NotImplementedError('initialization method [%s] is not implemented' % init_type)
            if getattr(m, 'bias', None) is not None:
                nn.init.constant_(m.bias.data, 0.0)
        elif kind == 'BatchNorm2d':
            if m.weight is not None:
                nn.init.normal_(m.weight.data, 1.0, init_gain)
            if m.bias is not None:
                nn.init.constant_(m.bias.data, 0.0)
    net.apply(visit)


class BasicBlock(nn.Module):
    """conv.py:440-468."""

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=3, stride=stride, padding=1)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        residual = x if self.downsample is None else self.downsample(x)
        out = out + residual
        return self.relu(out)


class ResNet18(nn.Module):
    """conv.py:509-564: the chairs encoder.  `bn_out` exists but is never applied (conv.py:526, 559)."""

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
        self.out_function = act(out_f)

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes, kernel_size=1, stride=stride), nn.BatchNorm2d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        layers += [BasicBlock(planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x, return_skip=False):
        x = x.view(x.size(0), -1, x.size(3), x.size(4))
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = self.out_function(self.conv_out(x))
        return x.view(len(x), -1)


def get_encoder(nn_type, shape, output_size, hidden_size, n_layers, nt_cond, init_type, init_gain):
    """factory.py:25-44."""
    nc, dim = shape[0], shape[-1]
    if nn_type == 'dcgan':
        assert dim == 64
        enc = DCGAN64Encoder(nc * nt_cond, output_size, hidden_size)
    elif nn_type == 'vgg':
        assert dim in [32, 64]
        enc = VGG64Encoder(nc * nt_cond, output_size, hidden_size, vgg32=dim == 32)
    elif nn_type == 'encoderSST':
        enc = EncoderSST(nc * nt_cond, output_size)
    elif nn_type == 'resnet':
        enc = ResNet18(output_size, nc * nt_cond)
    elif nn_type == 'mlp':
        enc = MLPEncoder(int(nt_cond * np.prod(np.array(shape))), hidden_size, output_size, n_layers)
    else:
        raise NotImplementedError(nn_type)
    init_net(enc, init_type=init_type, init_gain=init_gain)
    return enc


def get_decoder(nn_type, shape, code_size_t, code_size_s, last_activation, hidden_size, n_layers, mixing, skipco,
                init_type, init_gain):
    """factory.py:47-76."""
    assert not skipco or nn_type in ['dcgan', 'vgg', 'decoderSST']
    if mixing == 'mul':
        assert code_size_t == code_size_s
        input_size = code_size_t
    else:
        input_size = code_size_t + code_size_s
    nc, dim = shape[0], shape[-1]
    if nn_type == 'dcgan':
        assert dim == 64
        dec = DCGAN64Decoder(nc, input_size, hidden_size, skipco, last_activation, mixing)
    elif nn_type == 'vgg':
        assert dim in [32, 64]
        dec = VGG64Decoder(nc, input_size, hidden_size, skipco, last_activation, mixing, vgg32=dim == 32)
    elif nn_type == 'mlp':
        dec = MLPDecoder(input_size, hidden_size, shape, n_layers, last_activation, mixing)
    elif nn_type == 'decoderSST':
        assert mixing == 'concat'
        dec = (DecoderSST_Skip if skipco else DecoderSST)(input_size, nc, last_activation)
    else:
        raise NotImplementedError(nn_type)
    init_net(dec, init_type=init_type, init_gain=init_gain)
    return dec


def get_resnet(latent_size, n_blocks, hidden_size, init_type, gain_res, fully_conv=False):
    """factory.py:79-87."""
    net = ConvResnet(latent_size, n_blocks=n_blocks, nf=hidden_size) if fully_conv \
        else MLPResnet(latent_size, n_blocks, hidden_size)
    init_net(net, init_type=init_type, init_gain=gain_res)
    return net


# --------------------------------------------------------------------------- orchestrator
class SeparableNetwork(nn.Module):
    """model.py:20-89."""

    def __init__(self, Es, Et, t_resnet, decoder, nt_cond, skipco):
        super().__init__()
        for m in (Es, Et, t_resnet, decoder):
            assert isinstance(m, nn.Module)
        self.Es, self.Et, self.decoder, self.t_resnet = Es, Et, decoder, t_resnet
        self.nt_cond, self.skipco = nt_cond, skipco

    def get_forecast(self, cond, n_forecast, init_t_code=None, init_s_code=None):
        s_code = self.Es(cond, return_skip=self.skipco) if init_s_code is None else init_s_code
        s_skip = None
        if self.skipco:
            s_code, s_skip = s_code
        t_code = self.Et(cond) if init_t_code is None else init_t_code
        t_codes, frames, residuals = [t_code], [self.decoder(s_code, t_code, skip=s_skip)], []
        for _ in range(1, n_forecast):
            t_code, res = self.t_resnet(t_code)
            t_codes.append(t_code)
            residuals.append(res)
            frames.append(self.decoder(s_code, t_code, skip=s_skip))
        return torch.stack(frames, dim=1), torch.stack(t_codes, dim=1), s_code, residuals


# --------------------------------------------------------------------------- losses
def zero_order_loss(s_old, s_new, skipco):
    """train.py:38-42."""
    if skipco:
        s_old = torch.cat([s_old[0].flatten()] + [x.flatten() for x in s_old[1]])
        s_new = torch.cat([s_new[0].flatten()] + [x.flatten() for x in s_new[1]])
    return (s_old - s_new).pow(2).mean()


def ae_loss(cond, target, sep_net, nt_cond, offset, skipco, t_random=None):
    """train.py:45-88.  `t_random=None` draws from the global NumPy RNG exactly as the reference does."""
    full = torch.cat([cond, target], dim=1)
    s_old = sep_net.Es(full[:, :nt_cond], return_skip=skipco)
    s_new = sep_net.Es(full[:, -nt_cond:], return_skip=skipco)
    if t_random is None:
        t_random = np.random.randint(nt_cond, full.size(1) + (0 if offset == 0 else 1))
    t_code = sep_net.Et(full[:, t_random - nt_cond:t_random])
    if skipco:
        rec = sep_net.decoder(s_old[0], t_code, skip=s_old[1])
    else:
        rec = sep_net.decoder(s_old, t_code)
    return F.mse_loss(full[:, t_random - offset], rec, reduction='mean'), s_new, s_old


def training_losses(cond, target, sep_net, nt_cond, nt_pred, offset, skipco, lamb_ae, lamb_s, lamb_t, lamb_pred,
                    average_tloss=False, t_random=None):
    """train.py:111-149: returns (total, dict of the four un-weighted terms, forecasts, t_codes)."""
    assert offset == nt_cond or offset == 0
    ae, s_new, s_old = ae_loss(cond, target, sep_net, nt_cond, offset, skipco, t_random=t_random)
    zero = zero_order_loss(s_old, s_new, skipco)
    full = torch.cat([cond, target], dim=1)
    forecasts, t_codes, _, _ = sep_net.get_forecast(cond, nt_pred + offset, init_s_code=s_old)
    pred = F.mse_loss(forecasts, full[:, (nt_cond if offset == 0 else 0):])
    if average_tloss:
        t_reg = 0.5 * t_codes[:, 0].pow(2).view(full.shape[0], -1).mean()
    else:
        t_reg = 0.5 * torch.sum(t_codes[:, 0].pow(2), dim=1).mean()
    total = lamb_ae * ae + lamb_s * zero + lamb_pred * pred + lamb_t * t_reg
    return total, {'ae': ae, 'zero': zero, 'pred': pred, 't_reg': t_reg}, forecasts, t_codes


# --------------------------------------------------------------------------- configs
def build_sep_net(cfg, seed=None):
    """Assemble Es/Et/decoder/t_resnet the way main.py:119-140 does, from a dict of CLI-style options."""
    if seed is not None:
        torch.manual_seed(seed)
    shape = list(cfg['shape'])
    arch = cfg['architecture']
    dec_arch = cfg.get('decoder_architecture') or arch
    if cfg.get('no_s'):                      # main.py:124-129: constant S, forces mul mixing with code_s = code_t
        assert not cfg.get('skipco', False)
        cfg = dict(cfg, code_size_s=cfg['code_size_t'], mixing='mul')
        Es = ConstantS(return_value=1, code_size=cfg['code_size_s'])
    else:
        Es = get_encoder(arch, shape, cfg['code_size_s'], cfg['enc_hidden_size'], cfg.get('enc_n_layers', 3),
                         cfg['nt_cond'], cfg.get('init_encoder', 'normal'), cfg.get('gain_encoder', 0.02))
    Et = get_encoder(arch, shape, cfg['code_size_t'], cfg['enc_hidden_size'], cfg.get('enc_n_layers', 3),
                     cfg['nt_cond'], cfg.get('init_encoder', 'normal'), cfg.get('gain_encoder', 0.02))
    dec = get_decoder(dec_arch, shape, cfg['code_size_t'], cfg['code_size_s'], cfg.get('last_activation'),
                      cfg['dec_hidden_size'], cfg.get('dec_n_layers', 3), cfg.get('mixing', 'concat'),
                      cfg.get('skipco', False), cfg.get('init_encoder', 'normal'), cfg.get('gain_encoder', 0.02))
    res = get_resnet(cfg['code_size_t'], cfg.get('n_blocks', 1), cfg.get('res_hidden_size', 512),
                     cfg.get('init_resnet', 'orthogonal'), cfg.get('gain_resnet', 1.41), arch == 'encoderSST')
    return SeparableNetwork(Es, Et, res, dec, cfg['nt_cond'], cfg.get('skipco', False))
