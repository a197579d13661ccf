"""MLP on the fused HIP Linear chain (reference: networks/mlp.py:24-75)."""
import torch.nn as nn

from .. import functional as VF
from .utils import activation_factory


class MLP(nn.Module):
    """`nlayers` Linear layers with `activation` between them.

    The module tree mirrors the reference (`module.0 = Sequential(Linear)`, `module.i = Sequential(act, Linear)`) so
    state dicts are interchangeable; the forward pass does not run those containers but hands the Linear parameters
    to one fused chain (`functional.MLPChain`): bias + activation live in the GEMM epilogues.
    """

    def __init__(self, ninp, nhid, nout, nlayers, activation='relu'):
        super().__init__()
        assert nhid == 0 or nlayers > 1
        self.activation = activation
        blocks = []
        for il in range(nlayers):
            lin = nn.Linear(ninp if il == 0 else nhid, nout if il == nlayers - 1 else nhid)
            blocks.append(nn.Sequential(lin) if il == 0 else nn.Sequential(activation_factory(activation), lin))
        self.module = nn.Sequential(*blocks)

    def linears(self):
        return [blk[-1] for blk in self.module]

    def hidden_activation(self):
        """Name of the activation between the layers; read off the module tree when the object was restored from a reference
        checkpoint (whole-module pickle of `var_sep.networks.mlp.MLP`, which keeps no such attribute)."""
        act = self.__dict__.get('activation')
        if act is None:
            from .utils import activation_name
            act = activation_name(self.module[1][0]) if len(self.module) > 1 else 'relu'
            self.activation = act
        return act

    def forward(self, x, out_act='none'):
        return VF.mlp_chain(x, self.linears(), hidden_act=self.hidden_activation(), out_act=out_act)
