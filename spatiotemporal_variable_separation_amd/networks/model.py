"""Model orchestrator (reference: networks/model.py:20-89)."""
import os

import torch
import torch.nn as nn


class SeparableNetwork(nn.Module):
    """Holds E_s, E_t, the latent integrator and the decoder; `get_forecast` is the only compute method."""

    def __init__(self, Es, Et, t_resnet, decoder, nt_cond, skipco):
        super().__init__()
        # registration order fixes parameters() / state_dict() order: Es, Et, decoder, t_resnet as in the reference (model.py:31-38)
        for name, module in (('Es', Es), ('Et', Et), ('decoder', decoder), ('t_resnet', t_resnet)):
            assert isinstance(module, nn.Module), f'{name} must be an nn.Module'
            setattr(self, name, module)
        self.nt_cond, self.skipco = nt_cond, skipco
        self._grad = True
        self.fused = True          # additive switch: False forces the reference's per-step launch structure

    @property
    def grad(self):
        return self._grad

    @grad.setter
    def grad(self, grad):
        assert isinstance(grad, bool)
        self._grad = grad

    def get_forecast(self, cond, n_forecast, init_t_code=None, init_s_code=None, rolled=None):
        """Encode once, decode frame 0, then roll the temporal code forward n_forecast-1 times (model.py:52-89).

        Returns (forecasts [B,n,C,H,W], t_codes [B,n,...], s_code, t_residuals) like the reference.
        """
        s_code = self.Es(cond, return_skip=self.skipco) if init_s_code is None else init_s_code
        s_skipco = None
        if self.skipco:
            s_code, s_skipco = s_code
        t_code = self.Et(cond) if init_t_code is None else init_t_code

        # MI355X fast path (identical arithmetic, different launch structure): the MLP integrator runs the whole
        # recurrence in one persistent kernel, and the n decoder calls run as one batch over time -- for the conv
        # decoders with per-call BatchNorm statistics kept per step (grouped BatchNorm), so nothing changes numerically.
        if self.fused and hasattr(self.decoder, 'decode_sequence') and t_code.is_cuda:
            if rolled is not None:
                t_codes, t_residuals = rolled           # (the caller rolled the code forward already: train.compute_losses, on a side stream)
            elif hasattr(self.t_resnet, 'rollout'):
                t_codes, t_residuals = self.t_resnet.rollout(t_code, n_forecast)
            else:
                codes, t_residuals = self._roll(t_code, n_forecast)
                t_codes = torch.stack(codes, dim=1)
            from .. import functional as VF
            # (a recorded data-parallel step splits its backward pass at the decoder's inputs: the decoder then sees detached leaves)
            d_s, d_t, d_skip = VF.cut((s_code, t_codes, s_skipco))
            forecasts = self.decoder.decode_sequence(d_s, d_t, skip=d_skip)
            return forecasts, t_codes, s_code, t_residuals

        # the reference's launch structure: one decoder call per code
        codes, t_residuals = self._roll(t_code, n_forecast)
        frames = [self.decoder(s_code, code, skip=s_skipco) for code in codes]
        return torch.stack(frames, dim=1), torch.stack(codes, dim=1), s_code, t_residuals

    def _roll(self, t_code, n_forecast):
        """[t_0, ..., t_{n-1}] with t_{k+1} = t_resnet(t_k), and the per-step residual lists."""
        codes, residuals = [t_code], []
        want_alias = getattr(self.t_resnet, 'supports_alias', False) and os.environ.get('VARSEP_RESBLOCK_ALIAS', '1') == '1'
        while len(codes) < n_forecast:
            if want_alias:
                # a fused ConvResBlock hands its output out twice: the recurrence goes on with one, the list keeps the other, and the two
                # gradients join inside the block's backward launches (functional.ConvResBlockFn) instead of in an add launch per frame
                t_code, res, alias = self.t_resnet(t_code, return_alias=True)
                codes.append(alias if alias is not None else t_code)
            else:
                t_code, res = self.t_resnet(t_code)
                codes.append(t_code)
            residuals.append(res)
        return codes, residuals
