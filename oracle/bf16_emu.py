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


# the 16-bit type whose rounding is emulated: bfloat16 (the 'bf16' mode) or float16 (the 'fp16' mode = the reference's
# --torch_amp operand type); the rounding POINTS are the same in both modes
_LOWP = {'dtype': torch.bfloat16, 'precision': 'bf16'}


def _lowp():
    return _LOWP['dtype']


def bf(t):
    return t.to(_LOWP['dtype']).float()


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
    """Context manager: inside it, the oracle's MLP-family modules round like the product's bf16 mode (or, with
    dtype=torch.float16, like its fp16 mode)."""

    def __init__(self, dtype=torch.bfloat16):
        self.dtype = dtype

    def __enter__(self):
        self._lowp_saved = dict(_LOWP)
        _LOWP['dtype'], _LOWP['precision'] = self.dtype, 'fp16' if self.dtype == torch.float16 else 'bf16'
        self._saved = (cpu_ref.MLP.forward, cpu_ref.MLPDecoder.forward)
        cpu_ref.MLP.forward = _mlp_forward
        cpu_ref.MLPDecoder.forward = _mlp_decoder_forward
        return self

    def __exit__(self, *exc):
        cpu_ref.MLP.forward, cpu_ref.MLPDecoder.forward = self._saved
        _LOWP.update(self._lowp_saved)
        return False


# ======================================================================================================================
# Convolution families: the product's OWN module tree and host logic (networks/*.py, train.compute_losses) run on the CPU
# with every functional entry point they use replaced by a plain-torch emulation that rounds to bfloat16 exactly where the
# HIP kernels do.  What this checks is the bf16 ARITHMETIC of the kernels (conv / BatchNorm / pool / upsample / Linear
# chains / integrator), layer by layer through a whole training step; the structure of the networks is pinned separately,
# in fp32, against the independent oracle above.  Rounding points of a conv block (functional.ConvBlock):
#   forward : block input -> bf16; weights -> bf16; conv accumulates in fp32; with BatchNorm the conv output z is STORED in
#             bf16, statistics are taken from that stored z (fp64 sums), y = act(gamma * xhat + beta) is stored in bf16
#             (fp32 for a module's final block); without BatchNorm the conv output is stored, then the activation is applied
#             to the stored value and stored again.
#   backward: the incoming gradient has the dtype of the block output; dz (BatchNorm / activation backward, fp32 math on the
#             stored z) -> bf16; weight gradient fp32 from bf16 dz and bf16 input; input gradient in the input's dtype.
import torch.nn.functional as F  # noqa: E402



def _conv(x32, w32, bias, stride, pad, transposed):
    if transposed:
        return F.conv_transpose2d(x32, w32, bias, stride=stride, padding=pad)
    return F.conv2d(x32, w32, bias, stride=stride, padding=pad)


def _act_grad(pre_or_out, name):
    return _act_grad_from_out(pre_or_out, name)


class EmuConvBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, rmean, rvar, cfg):
        transposed, stride, pad, has_bn, act, training, momentum, eps, out_fp32, groups = cfg
        out_dt = torch.float32 if out_fp32 else _lowp()
        xc, wc = x.detach().to(_lowp()), w.detach().to(_lowp())
        bias = b.detach().float() if b is not None else None
        z32 = _conv(xc.float(), wc.float(), bias, stride, pad, transposed)
        if has_bn:
            z = z32.to(_lowp())
            zf = z.double()
            Bt, C = z.shape[0], z.shape[1]
            per = Bt // groups
            mean = torch.empty((groups, C), dtype=torch.float32)
            invstd = torch.empty((groups, C), dtype=torch.float32)
            for gi in range(groups):
                chunk = zf[gi * per:(gi + 1) * per]
                if training:
                    m = chunk.mean(dim=(0, 2, 3))
                    v = chunk.var(dim=(0, 2, 3), unbiased=False)
                    n = chunk.numel() // C
                    with torch.no_grad():
                        rmean.mul_(1 - momentum).add_(momentum * m.float())
                        rvar.mul_(1 - momentum).add_(momentum * (v * n / max(n - 1, 1)).float())
                    mean[gi], invstd[gi] = m.float(), (1.0 / torch.sqrt(v + eps)).float()
                else:
                    mean[gi], invstd[gi] = rmean.detach(), torch.rsqrt(rvar.detach() + eps)
            mrow = mean.repeat_interleave(per, dim=0)[:, :, None, None]
            irow = invstd.repeat_interleave(per, dim=0)[:, :, None, None]
            pre = gamma.detach().float()[None, :, None, None] * ((z.float() - mrow) * irow) + beta.detach().float()[None, :, None, None]
            y = _ACT[act](pre).to(out_dt)
            ctx.save_for_backward(xc, z, mean, invstd)
        else:
            y = z32.to(out_dt)
            if act not in ('none', None):
                y = _ACT[act](y.float()).to(out_dt)
            ctx.save_for_backward(xc, y)
        ctx.cfg, ctx.w, ctx.b, ctx.gamma, ctx.beta = cfg, w, b, gamma, beta
        ctx.x_dtype, ctx.x_needs_grad = x.dtype, x.requires_grad
        return y

    @staticmethod
    def backward(ctx, dy):
        transposed, stride, pad, has_bn, act, training, momentum, eps, out_fp32, groups = ctx.cfg
        w, b = ctx.w, ctx.b
        dgamma = dbeta = None
        dyf = dy.float()
        if has_bn:
            xc, z, mean, invstd = ctx.saved_tensors
            Bt, C = z.shape[0], z.shape[1]
            per = Bt // groups
            mrow = mean.repeat_interleave(per, dim=0)[:, :, None, None]
            irow = invstd.repeat_interleave(per, dim=0)[:, :, None, None]
            g32 = ctx.gamma.detach().float()[None, :, None, None]
            xhat = (z.float() - mrow) * irow
            pre = g32 * xhat + ctx.beta.detach().float()[None, :, None, None]
            dpre = dyf * _act_grad(_ACT[act](pre), act) if act not in ('none', None) else dyf
            dz = torch.empty_like(dpre)
            dgamma = torch.zeros(C)
            dbeta = torch.zeros(C)
            for gi in range(groups):
                sl = slice(gi * per, (gi + 1) * per)
                dp, xh = dpre[sl].double(), xhat[sl].double()
                sb, sg = dp.sum(dim=(0, 2, 3)), (dp * xh).sum(dim=(0, 2, 3))
                n = dp.numel() // C
                if training:
                    d = g32.double() * irow[sl].double() * (dp - sb[None, :, None, None] / n - xh * sg[None, :, None, None] / n)
                else:
                    d = g32.double() * irow[sl].double() * dp
                dz[sl] = d.float()
                dgamma += sg.float()
                dbeta += sb.float()
            dz = dz.to(_lowp())
        else:
            xc, y = ctx.saved_tensors
            dz = (dyf * _act_grad(y.float(), act)).to(_lowp()) if act not in ('none', None) else dy.to(_lowp())
        db = None
        if b is not None and b.requires_grad:
            db = torch.zeros_like(b) if (has_bn and training) else dz.float().sum(dim=(0, 2, 3))
        with torch.enable_grad():
            xr = xc.float().requires_grad_(True)
            wr = w.detach().to(_lowp()).float().requires_grad_(True)
            zz = _conv(xr, wr, None, stride, pad, transposed)
            gx, gw = torch.autograd.grad(zz, (xr, wr), dz.float())
        dx = gx.to(ctx.x_dtype) if ctx.x_needs_grad else None
        return dx, gw, db, dgamma, dbeta, None, None, None


class _EmuPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y, idx = F.max_pool2d(x.float(), 2, 2, return_indices=True)
        ctx.save_for_backward(idx)
        ctx.shape, ctx.dt = x.shape, x.dtype
        return y.to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        return F.max_unpool2d(dy.float(), idx, 2, 2, output_size=ctx.shape[-2:]).to(ctx.dt)


class _EmuUpsample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.dt = x.dtype
        return F.interpolate(x.float(), scale_factor=2, mode='nearest').to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        d = dy.float()
        return (d[:, :, 0::2, 0::2] + d[:, :, 0::2, 1::2] + d[:, :, 1::2, 0::2] + d[:, :, 1::2, 1::2]).to(ctx.dt)


class _EmuActivation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        y = _ACT[act](x.float()).to(x.dtype)
        ctx.save_for_backward(y)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return (dy.float() * _act_grad(y.float(), ctx.act)).to(y.dtype), None


def _emu_mlp_chain(x, linears, hidden_act='relu', out_act='none'):
    params = []
    for lin in linears:
        params += [lin.weight, lin.bias]
    acts = tuple([hidden_act] * (len(linears) - 1) + [out_act])
    return EmuChain.apply(x.float(), acts, *params)


class _EmuRollout:
    """functional.MLPRollout.apply(x0, n_steps, *params): the residual blocks step by step through EmuChain (same rounding
    points as the fused kernels: block input and hidden activations bf16, fp32 accumulation and residual sum)."""

    @staticmethod
    def apply(x0, n_steps, *params):
        nb = len(params) // 6
        x, codes, ress = x0.float(), [x0.float()], []
        for _ in range(1, n_steps):
            row = []
            for bi in range(nb):
                r = EmuChain.apply(x, ('relu', 'relu', 'none'), *params[6 * bi:6 * bi + 6])
                x = x + r
                row.append(r)
            codes.append(x)
            ress.append(torch.stack(row, 0))
        res = torch.stack(ress, 0).detach() if ress else torch.zeros((0, nb) + tuple(x0.shape))
        return torch.stack(codes, dim=1), res


class emulate_product_bf16:
    """Context manager: the product's functional entry points used by its network classes become CPU emulations of the bf16
    kernels (module docstring above).  Inside it, build the product network on the CPU and run train.compute_losses."""

    def __init__(self, dtype=torch.bfloat16):
        self.dtype = dtype

    def __enter__(self):
        from spatiotemporal_variable_separation_amd import functional as VF
        self.VF = VF
        self._lowp_saved = dict(_LOWP)
        _LOWP['dtype'], _LOWP['precision'] = self.dtype, 'fp16' if self.dtype == torch.float16 else 'bf16'
        self._saved = (VF.ConvBlock, VF.MaxPool2, VF.Upsample2, VF.Activation, VF.mlp_chain, VF.MLPRollout, VF._STATE['precision'])
        VF.ConvBlock, VF.MaxPool2, VF.Upsample2, VF.Activation = EmuConvBlock, _EmuPool, _EmuUpsample, _EmuActivation
        VF.mlp_chain, VF.MLPRollout = _emu_mlp_chain, _EmuRollout
        VF._STATE['precision'] = _LOWP['precision']
        return self

    def __exit__(self, *exc):
        VF = self.VF
        VF.ConvBlock, VF.MaxPool2, VF.Upsample2, VF.Activation, VF.mlp_chain, VF.MLPRollout, VF._STATE['precision'] = self._saved
        _LOWP.update(self._lowp_saved)
        return False
