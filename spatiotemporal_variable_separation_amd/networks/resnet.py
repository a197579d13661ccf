"""Residual latent integrators (reference: networks/resnet.py:22-88)."""
import torch.nn as nn

from .. import functional as VF
from .mlp import MLP


class MLPResBlock(nn.Module):
    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.mlp = MLP(input_size, hidden_size, input_size, 3)

    def forward(self, x):
        residual = self.mlp(x)
        return x + residual, residual


class MLPResnet(nn.Module):
    def __init__(self, input_size, n_blocks, hidden_size):
        super().__init__()
        self.in_size = input_size
        self.n_blocks = n_blocks
        self.blocks = nn.ModuleList([MLPResBlock(input_size, hidden_size) for _ in range(n_blocks)])

    def forward(self, x, return_res=True):
        residuals = []
        for blk in self.blocks:
            x, res = blk(x)
            residuals.append(res)
        return (x, residuals) if return_res else x

    def prepack(self):
        """Bring the integrator kernels' weight packs (forward and transposed) up to date now -- one launch on the current stream -- so
        that `rollout` finds them ready: the training step calls this on the integrator's stream BEFORE the encoders, where it runs
        beside E_t's first layers instead of between E_t and the recurrence."""
        ws = [lin.weight for blk in self.blocks for lin in blk.mlp.linears()]
        VF.prepack_weights([(w, tr) for w in ws for tr in (False, True)], VF.compute_dtype())

    def rollout(self, x0, n_steps):
        """Fused form of `for t in 1..n_steps-1: x, res = self(x)` (model.py:78-83): one persistent kernel.

        Returns (t_codes [B, n_steps, C] with t_codes[:, 0] = x0, t_residuals as the reference's list of lists)."""
        if n_steps <= 1:
            return x0.unsqueeze(1), []
        params = []
        for blk in self.blocks:
            for lin in blk.mlp.linears():
                params += [lin.weight, lin.bias]
        t_codes, res = VF.MLPRollout.apply(x0, n_steps, *params)
        t_residuals = [[res[t, b] for b in range(self.n_blocks)] for t in range(n_steps - 1)]
        return t_codes, t_residuals
