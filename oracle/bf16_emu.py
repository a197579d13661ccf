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
        ctx.lp = _lowp()
        return saved[-1]

    @staticmethod
    def backward(ctx, dy):
        saved, params, acts = ctx.saved_tensors, ctx.params, ctx.acts
        L = len(params) // 2
        lp = ctx.lp

        def bf(t):                                    # the 16-bit type of THIS chain's forward pass
            return t.to(lp).float()
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


# ======================================================================================================================
# Convolution families on the ORACLE's module tree (independent of the product's host code).
#
# The classes of oracle/cpu_ref.py are plain nn.Sequential stacks of Conv2d / ConvTranspose2d / BatchNorm2d / activation / pool /
# upsample / Linear modules in the reference's call structure (one decoder call per frame, one encoder call per window, one
# ConvResBlock call per step and block).  `emulate_bf16` swaps their `forward` methods for interpreters of the SAME module lists that
# round to the 16-bit type at the mode's rounding points and otherwise compute in fp32 with torch's CPU kernels.  Nothing of the
# product (networks/*.py, functional.py, train.py) is imported or executed: layer grouping, skip wiring, per-call BatchNorm statistics
# and the losses are the oracle's.
#
# Rounding points of the 16-bit modes, conv families (`lp` = the 16-bit type):
#   conv block  = conv -> [BatchNorm2d] -> [activation]                                         (reference conv.py:41-60)
#     forward : block input -> lp; weight -> lp; products exact, accumulation + bias in fp32;
#               with BatchNorm: conv output z STORED in lp, batch statistics (fp64 sums) of the stored z, running statistics from
#               them, y = act(gamma * xhat + beta) stored in lp -- in fp32 when the block is the last one of an encoder / decoder /
#               residual branch (module outputs are fp32);
#               without BatchNorm: conv output stored, activation applied to the stored value and stored again.
#     backward: incoming gradient in the dtype of the block output; dz (BatchNorm / activation backward in fp32 on the stored z) -> lp;
#               weight gradient fp32 from lp dz and lp input; bias gradient fp32 sum of the lp dz -- EXACTLY ZERO in front of a
#               training-mode BatchNorm; d gamma / d beta fp32 (fp64 sums); input gradient in the dtype of the block input.
#   max-pool / nearest up-sampling: on lp values (pooling is exact; the up-sampling gradient sums its four taps in fp32, then lp).
#   skip tensors (encoder stage outputs) are lp; the codes leaving an encoder and the frames leaving a decoder are fp32.
#   Linear layers: the chain rules of the first half (EmuChain).
#   A tensor consumed by several calls (the spatial code and the skips of E_s across the n + 1 decoder calls): gradient contributions are
#   summed in fp32 and rounded to the tensor's dtype once.
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402

_ACT_MODULES = {'ReLU': 'relu', 'LeakyReLU': 'leaky_relu', 'ELU': 'elu', 'Sigmoid': 'sigmoid', 'Tanh': 'tanh'}


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
        ctx.lp = _lowp()                              # backward rounds to the type of ITS forward pass, whatever context it runs in
        return y

    @staticmethod
    def backward(ctx, dy):
        transposed, stride, pad, has_bn, act, training, momentum, eps, out_fp32, groups = ctx.cfg
        w, b = ctx.w, ctx.b
        lp = ctx.lp
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
            dz = dz.to(lp)
        else:
            xc, y = ctx.saved_tensors
            dz = (dyf * _act_grad(y.float(), act)).to(lp) if act not in ('none', None) else dy.to(lp)
        db = None
        if b is not None and b.requires_grad:
            db = torch.zeros_like(b) if (has_bn and training) else dz.float().sum(dim=(0, 2, 3))
        with torch.enable_grad():
            xr = xc.float().requires_grad_(True)
            wr = w.detach().to(lp).float().requires_grad_(True)
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


class _EmuPool3s2(torch.autograd.Function):
    """nn.MaxPool2d(3, 2, 1) on 16-bit values (exact); overlapping windows: the gradient of an input sums its windows' (fp32, then 16 bits)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return F.max_pool2d(x.float(), 3, 2, 1).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        with torch.enable_grad():
            xr = x.float().requires_grad_(True)
            (g,) = torch.autograd.grad(F.max_pool2d(xr, 3, 2, 1), xr, dy.float())
        return g.to(x.dtype)


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



class _SharedGrad(torch.autograd.Function):
    """y = x.float() for a 16-bit tensor that several calls consume: the calls cast y back to 16 bits (an exact round trip), their
    gradients meet on y in fp32 and are rounded to x's dtype ONCE here."""

    @staticmethod
    def forward(ctx, x):
        ctx.dt = x.dtype
        return x.float()

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dt)


def _flat(module, out):
    if isinstance(module, (nn.Sequential, nn.ModuleList)):
        for m in module:
            _flat(m, out)
    elif not isinstance(module, nn.Identity):
        out.append(module)
    return out


def _emu_layers(module, h, final_act='none', final_fp32=False):
    """Interpret a (nested) Sequential of the oracle's layer objects with the mode's rounding points.  `final_act`: activation the
    caller applies to the result (a decoder's `last_activation`), taken into the last block when that block has none of its own;
    `final_fp32`: the last block writes fp32 (module outputs)."""
    layers = _flat(module, [])
    i, n = 0, len(layers)
    while i < n:
        m = layers[i]
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            bn = act = None
            j = i + 1
            if j < n and isinstance(layers[j], nn.BatchNorm2d):
                bn, j = layers[j], j + 1
            if j < n and type(layers[j]).__name__ in _ACT_MODULES:
                act, j = _ACT_MODULES[type(layers[j]).__name__], j + 1
            last = j >= n
            act = act or 'none'
            extra = None
            if last and final_act not in ('none', None):
                if act == 'none':
                    act = final_act
                else:
                    extra = final_act
            training = bn.training if bn is not None else False
            if bn is not None and training:
                with torch.no_grad():
                    bn.num_batches_tracked += 1
            cfg = (isinstance(m, nn.ConvTranspose2d), m.stride[0], m.padding[0], bn is not None, act, training,
                   bn.momentum if bn is not None else 0.1, bn.eps if bn is not None else 1e-5, bool(final_fp32 and last and extra is None), 1)
            h = EmuConvBlock.apply(h, m.weight, m.bias, bn.weight if bn is not None else None, bn.bias if bn is not None else None,
                                   bn.running_mean if bn is not None else None, bn.running_var if bn is not None else None, cfg)
            if extra is not None:
                h = _EmuActivation.apply(h.float() if final_fp32 else h, extra)
            i = j
            continue
        if isinstance(m, nn.MaxPool2d):
            assert m.kernel_size == 2 and m.stride == 2
            h = _EmuPool.apply(h)
        elif isinstance(m, nn.Upsample):
            h = _EmuUpsample.apply(h)
        elif isinstance(m, nn.Flatten):
            h = h.reshape(h.shape[0], -1)
        elif isinstance(m, nn.Linear):
            last = i == n - 1
            h = EmuChain.apply(h.float(), (final_act if (last and final_act) else 'none',), m.weight, m.bias)
        else:
            raise NotImplementedError(type(m).__name__)
        i += 1
    if final_fp32 and h.dtype != torch.float32:
        h = h.float()
    return h


def _last_act_name(module):
    return {'Sigmoid': 'sigmoid', 'Tanh': 'tanh', 'ReLU': 'relu', 'LeakyReLU': 'leaky_relu', 'ELU': 'elu', 'Identity': 'none'}[type(module).__name__]


def _share(t):
    """Mark a (possibly 16-bit) tensor that several calls will consume (see _SharedGrad); idempotent per tensor object."""
    if t.dtype == torch.float32:
        return t
    return _SharedGrad.apply(t)


def _flat_encoder_forward(self, x, return_skip=False):                         # cpu_ref._FlatEncoder (conv.py:81-99)
    h = x.reshape(x.size(0), -1, x.size(3), x.size(4))
    skips = []
    for stage in self.conv:
        h = _emu_layers(stage, h)
        skips.append(h)
    code = _emu_layers(self.last_op, h, final_fp32=True).view(-1, self.nh)
    if return_skip:
        return code, [_share(s) for s in skips[::-1]]
    return code


def _sst_encoder_forward(self, x, return_skip=False):                          # cpu_ref.EncoderSST (conv.py:346-356)
    h1 = _emu_layers(self.conv1, x.reshape(x.size(0), -1, x.size(3), x.size(4)))
    h2 = _emu_layers(self.conv2, h1)
    h3 = _emu_layers(self.conv3, h2)
    h4 = _emu_layers(self.conv4, h3, final_fp32=True)
    if return_skip:
        return h4, [_share(h3), _share(h2), _share(h1)]
    return h4


def _flat_decoder_forward(self, z1, z2, skip=None):                            # cpu_ref._FlatDecoder (conv.py:207-230)
    assert skip is None and not self.skip or self.skip and skip is not None
    z = torch.cat([z1, z2], dim=1) if self.mixing == 'concat' else z1 * z2
    h = _emu_layers(self.first_upconv, z.view(*z.shape, 1, 1))
    n_stage = len(self.conv)
    for i, stage in enumerate(self.conv):
        if skip is not None:
            h = torch.cat([h, skip[i].to(h.dtype)], 1)
        if i == n_stage - 1:
            h = _emu_layers(stage, h, final_act=_last_act_name(self.last_activation), final_fp32=True)
        else:
            h = _emu_layers(stage, h)
    return h


def _sst_skip_decoder_forward(self, s_code, t_code, skip):                     # cpu_ref.DecoderSST_Skip (conv.py:385-396)
    h3, h2, h1 = skip
    out = _emu_layers(self.conv1, torch.cat([s_code, t_code], dim=1))
    out = _emu_layers(self.conv2, torch.cat([h3.to(out.dtype), out], dim=1))
    out = _emu_layers(self.conv3, torch.cat([h2.to(out.dtype), out], dim=1))
    return _emu_layers(self.conv4, torch.cat([h1.to(out.dtype), out], dim=1), final_act=_last_act_name(self.out_f), final_fp32=True)


def _sst_decoder_forward(self, s_code, t_code, skip=None):                     # cpu_ref.DecoderSST (conv.py:419-426)
    x = _emu_layers(self.conv1, torch.cat([s_code, t_code], dim=1))
    x = _emu_layers(self.conv2, x)
    return _emu_layers(self.conv3, x, final_act=_last_act_name(self.out_f), final_fp32=True)


def _conv_res_block_forward(self, x):                                          # cpu_ref.ConvResBlock (resnet.py:53-70)
    r = _emu_layers(self.conv, x, final_fp32=True)
    skip = x if isinstance(self.up, nn.Identity) else _emu_layers(self.up, x, final_fp32=True)
    return skip + r, r


def _basic_block_forward(self, x):                                             # cpu_ref.BasicBlock (conv.py:440-468)
    out = _emu_layers(nn.Sequential(self.conv1, self.bn1, self.relu), x)
    out = _emu_layers(nn.Sequential(self.conv2, self.bn2), out)
    residual = x if self.downsample is None else _emu_layers(self.downsample, x)
    if residual.dtype != out.dtype:
        residual = residual.to(out.dtype)
    return _EmuActivation.apply(out + residual, 'relu')


def _resnet18_forward(self, x, return_skip=False):                             # cpu_ref.ResNet18 (conv.py:546-564)
    h = x.reshape(x.size(0), -1, x.size(3), x.size(4))
    h = _emu_layers(nn.Sequential(self.conv1, self.bn1, self.relu), h)
    h = _EmuPool3s2.apply(h)
    for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
        for block in stage:
            h = _basic_block_forward(block, h)
    h = _emu_layers(nn.Sequential(self.conv_out), h, final_act=_last_act_name(self.out_function), final_fp32=True)
    return h.reshape(len(h), -1)


class emulate_bf16:
    """Context manager: inside it, the oracle's modules round like the product's bf16 mode (or, with dtype=torch.float16, like its fp16
    mode): Linear chains through EmuChain, convolution blocks / pooling / up-sampling through the interpreter above."""

    _PATCHES = (('MLP', _mlp_forward), ('MLPDecoder', _mlp_decoder_forward), ('_FlatEncoder', _flat_encoder_forward),
                ('EncoderSST', _sst_encoder_forward), ('_FlatDecoder', _flat_decoder_forward), ('DecoderSST_Skip', _sst_skip_decoder_forward),
                ('DecoderSST', _sst_decoder_forward), ('ConvResBlock', _conv_res_block_forward), ('ResNet18', _resnet18_forward))

    def __init__(self, dtype=torch.bfloat16):
        self.dtype = dtype

    def __enter__(self):
        self._lowp_saved = dict(_LOWP)
        _LOWP['dtype'], _LOWP['precision'] = self.dtype, 'fp16' if self.dtype == torch.float16 else 'bf16'
        self._saved = [(name, getattr(cpu_ref, name).forward) for name, _ in self._PATCHES]
        for name, fn in self._PATCHES:
            getattr(cpu_ref, name).forward = fn
        return self

    def __exit__(self, *exc):
        for name, fn in self._saved:
            getattr(cpu_ref, name).forward = fn
        _LOWP.update(self._lowp_saved)
        return False


# ======================================================================================================================
# The product's OWN module tree and host logic (networks/*.py, train.compute_losses) on the CPU with every functional entry point
# replaced by the emulation classes above (`emulate_product_bf16`).  This checks the product's HOST code (launch structure: grouped
# BatchNorm over stacked calls, batched decoding, fused blocks) against the same rounding rules; it is NOT independent of the product
# and is used only (a) block by block in tests/test_conv_gpu.py and (b) here on the CPU, where tests/test_oracle_golden.py checks that it
# agrees with the independent emulation above.
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
