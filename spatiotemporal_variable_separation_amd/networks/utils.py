"""Activation table, weight initialisation and the constant-S encoder (reference: networks/utils.py:21-109)."""
import torch
import torch.nn as nn

_ACTS = {
    'relu': lambda: nn.ReLU(inplace=True),
    'leaky_relu': lambda: nn.LeakyReLU(0.2, inplace=True),
    'elu': lambda: nn.ELU(inplace=True),
    'sigmoid': nn.Sigmoid,
    'tanh': nn.Tanh,
    'identity': nn.Identity,
    None: nn.Identity,
}


def activation_factory(name):
    """Placeholder module for an activation (utils.py:50-72).  The HIP path fuses the arithmetic into the producing
    kernel; the module only keeps the module tree (and hence `state_dict` indices) identical to the reference."""
    if name not in _ACTS:
        raise ValueError(f'Activation function `{name}` not yet implemented')
    return _ACTS[name]()


def activation_name(module):
    """Inverse of `activation_factory` (used when a plan is derived from a module tree)."""
    kind = type(module).__name__
    return {'ReLU': 'relu', 'LeakyReLU': 'leaky_relu', 'ELU': 'elu', 'Sigmoid': 'sigmoid', 'Tanh': 'tanh',
            'Identity': 'none'}[kind]


class ConstantS(nn.Module):
    """`--no_s`: a spatial code of ones (utils.py:21-29)."""

    def __init__(self, return_value=1, code_size=1):
        super().__init__()
        self.code_size = code_size
        self.return_value = return_value

    def forward(self, x, return_skip=False):
        return torch.ones(len(x), self.code_size).to(x) * self.return_value


def init_net(net, init_type='normal', init_gain=0.02):
    """Same initial distribution as utils.py:75-109: dispatch on the class NAME of each sub-module."""
    fillers = {
        'normal': lambda w: nn.init.normal_(w, 0.0, init_gain),
        'xavier': lambda w: nn.init.xavier_normal_(w, gain=init_gain),
        'kaiming': lambda w: nn.init.kaiming_normal_(w, a=0, mode='fan_in'),
        'orthogonal': lambda w: nn.init.orthogonal_(w, gain=init_gain),
    }

    def visit(m):
        kind = type(m).__name__
        if kind in ('Conv2d', 'ConvTranspose2d', 'Linear'):
            if init_type not in fillers:
                raise NotImplementedError('initialization method [%s] is not implemented' % init_type)
            fillers[init_type](m.weight.data)
            if getattr(m, 'bias', None) is not None:
                nn.init.constant_(m.bias.data, 0.0)
        elif kind == 'BatchNorm2d':
            if m.weight is not None:
                nn.init.normal_(m.weight.data, 1.0, init_gain)
            if m.bias is not None:
                nn.init.constant_(m.bias.data, 0.0)

    net.apply(visit)
