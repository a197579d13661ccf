"""MLP encoder / decoder (reference: networks/mlp_encdec.py:25-50)."""
import numpy as np
import torch
import torch.nn as nn

from .mlp import MLP
from .utils import activation_factory, activation_name


class MLPEncoder(nn.Module):
    def __init__(self, input_size, hidden_size, output_size, nlayers):
        super().__init__()
        self.mlp = MLP(input_size, hidden_size, output_size, nlayers)

    def forward(self, x, return_skip=False):
        return self.mlp(x.reshape(len(x), -1))


class MLPDecoder(nn.Module):
    def __init__(self, latent_size, hidden_size, output_shape, nlayers, last_activation, mixing):
        super().__init__()
        self.output_shape = list(output_shape)
        self.mixing = mixing
        self.mlp = MLP(latent_size, hidden_size, int(np.prod(np.array(output_shape))), nlayers)
        self.last_activation = activation_factory(last_activation)

    def forward(self, z1, z2, skip=None):
        z = torch.cat([z1, z2], dim=1) if self.mixing == 'concat' else z1 * z2
        x = self.mlp(z, out_act=activation_name(self.last_activation))     # trailing activation fused in the last GEMM
        return x.view([-1] + self.output_shape)

    def decode_sequence(self, z1, t_codes, skip=None):
        """All frames of a rollout in one pass: rows (b, t) -> one Linear chain over B*n rows (there is no BatchNorm in
        the MLP family, so batching the n decoder calls of model.py:74-83 over time is exact).

        z1 [B, Cs], t_codes [B, n, Ct] -> frames [B, n, *output_shape]."""
        B, n = t_codes.shape[0], t_codes.shape[1]
        z1e = z1.unsqueeze(1).expand(B, n, z1.shape[1])
        z = torch.cat([z1e, t_codes], dim=2) if self.mixing == 'concat' else z1e * t_codes
        x = self.mlp(z.reshape(B * n, -1), out_act=activation_name(self.last_activation))
        return x.view([B, n] + self.output_shape)

    def decode_rollout(self, z1, t_first, t_codes, handoff=None):
        """decode_sequence(z1, cat([t_first[:, None], t_codes], 1)) with the decoder input built by one kernel
        (VF.MixCodes): the auto-encoding pair (train.py:85) and the forecasts of every rollout step (model.py:74-83).

        z1 [B, Cs], t_first [B, Ct], t_codes [B, n, Ct] -> frames [B, 1+n, *output_shape].  `handoff`: a VF.GradHandoff the
        consumer of the frames (the fused loss) may fill with the gradient of the last pre-activation."""
        from .. import functional as VF
        B, n = t_codes.shape[0], t_codes.shape[1]
        z, z_lowp = VF.MixCodes.apply(z1.float().contiguous(), t_first.float().contiguous(), t_codes.float().contiguous(), self.mixing)
        x = VF.mlp_chain(z.reshape(B * (n + 1), -1), self.mlp.linears(), hidden_act=self.mlp.hidden_activation(),
                         out_act=activation_name(self.last_activation),
                         x_lowp=None if z_lowp is None else z_lowp.reshape(B * (n + 1), -1), handoff=handoff)
        return x.view([B, n + 1] + self.output_shape)
