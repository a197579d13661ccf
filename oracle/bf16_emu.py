"""CPU emulation of the product's bf16 mixed-precision scheme.  TEST INFRASTRUCTURE ONLY (see cpu_ref.py).

The bf16 mode of the HIP path is DEFINED by where values are rounded to bfloat16 (round-to-nearest-even); between
those points all arithmetic is fp32 (MFMA accumulates bf16 x bf16 products, which are exact in fp32, into fp32).
This module patches the fp32 oracle so that it rounds at exactly those points, giving bf16 mode an oracle of its
own that the kernels must match to accumulation-order noise (~1e-5), instead of a loose "close to fp32" bound --
on tiny test networks a single ReLU unit flipped by an input rounding moves gradients by several percent, for the
product and for torch's own autocast alike.

Rounding points of a Linear chain (functional.MLPChain):
  forward : chain input -> bf16;  weights -> bf16 (shadow copies);  every hidden activation (after bias + ReLU) ->
            bf16;  the chain output (after the optional trailing activation) stays fp32;  biases stay fp32.
  backward: the gradient entering the chain, times the trailing activation's derivative, -> bf16;  every hidden
            gradient (after the ReLU mask) -> bf16;  weight gradients, bias gradients (sums of the bf16-rounded
            gradients) and the gradient leaving the chain are fp32.
"""
import torch

from . import cpu_ref


def bf(t):
    return t.to(torch.bfloat16).float()


def _act_grad_from_out(y, name):
    if name == 'relu':
        return (y > 0).float()
    if name == 'leaky_relu':
        return torch.where(y > 0, torch.ones_like(y), torch.full_like(y, 0.2))
    if name == 'sigmoid':
        return y * (1 - y)
    if name == 'tanh':
        return 1 - y * y
    return torch.ones_like(y)


_ACT = {'relu': torch.relu, 'leaky_relu': lambda z: torch.nn.functional.leaky_relu(z, 0.2), 'sigmoid': torch.sigmoid,
        'tanh': torch.tanh, 'none': lambda z: z, None: lambda z: z}


class EmuChain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, acts, *params):
        L = len(params) // 2
        h = bf(x)
        saved = [h]
        for l in range(L):
            W, b = params[2 * l], params[2 * l + 1]
            z = h @ bf(W).t() + b
            h = _ACT[acts[l]](z)
            if l < L - 1:
                h = bf(h)
            saved.append(h)
        ctx.save_for_backward(*saved)
        ctx.params, ctx.acts = params, acts
        return saved[-1]

    @staticmethod
    def backward(ctx, dy):
        saved, params, acts = ctx.saved_tensors, ctx.params, ctx.acts
        L = len(params) // 2
        dz = bf(dy * _act_grad_from_out(saved[L], acts[L - 1]))
        grads = [None] * (2 * L)
        dx = None
        for l in range(L - 1, -1, -1):
            W = params[2 * l]
            grads[2 * l] = dz.t() @ saved[l]
            grads[2 * l + 1] = dz.sum(0)
            if l > 0:
                dz = bf((dz @ bf(W)) * _act_grad_from_out(saved[l], acts[l - 1]))
            else:
                dx = dz @ bf(W)
        return (dx, None) + tuple(grads)


def _mlp_forward(self, x, out_act='none'):
    lins = [blk[-1] for blk in self.module]
    params = []
    for lin in lins:
        params += [lin.weight, lin.bias]
    acts = tuple(['relu'] * (len(lins) - 1) + [out_act])
    return EmuChain.apply(x, acts, *params)


def _mlp_decoder_forward(self, z1, z2, skip=None):
    z = torch.cat([z1, z2], dim=1) if self.mixing == 'concat' else z1 * z2
    name = {'Sigmoid': 'sigmoid', 'Tanh': 'tanh', 'ReLU': 'relu', 'Identity': 'none'}[type(self.last_activation).__name__]
    return _mlp_forward(self.mlp, z, out_act=name).view([-1] + self.output_shape)


class emulate_bf16:
    """Context manager: inside it, the oracle's MLP-family modules round like the product's bf16 mode."""

    def __enter__(self):
        self._saved = (cpu_ref.MLP.forward, cpu_ref.MLPDecoder.forward)
        cpu_ref.MLP.forward = _mlp_forward
        cpu_ref.MLPDecoder.forward = _mlp_decoder_forward
        return self

    def __exit__(self, *exc):
        cpu_ref.MLP.forward, cpu_ref.MLPDecoder.forward = self._saved
        return False
