"""Adam on the HIP path (reference: main.py:133 `optim.Adam(sep_net.parameters(), lr, betas)` + train.py:156-158).

`Adam` is a `torch.optim.Optimizer` (same constructor arguments, `state_dict()` layout: per-parameter `step`, `exp_avg`,
`exp_avg_sq`), whose `step()` updates every parameter of a group with ONE `vs_adam_multi` launch per 64 tensors instead
of torch's per-dtype/per-device multi-tensor kernels: the update is HBM-bound (28 bytes per parameter) and the launch
walks the tensors in 16 KiB chunks so that the two 24.6 M-element encoder matrices and the 32-element biases of the
WaveEq model keep every CU streaming.  The step count lives on the device, so `step()` can be recorded into a hipGraph
(train.GraphedStep) without `capturable=True` plumbing.  Only what the reference uses is supported: no weight decay, no
amsgrad, no maximize; fp32 CUDA parameters (anything else raises: there is no CPU fallback on the product path)."""
import ctypes
import os

import torch

from . import _lib
from ._lib import VarsepHipError


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, capturable=True, fused=None):
        if weight_decay != 0 or amsgrad:
            raise ValueError('the MI355X Adam implements what the reference uses: weight_decay=0, amsgrad=False')
        if not 0.0 < lr or not 0.0 < eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError('invalid Adam hyper-parameters')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._tables = {}
        self._buckets, self._bucket_of, self._left, self._launched, self._stream = [], {}, [], set(), None
        self._fused, self._fused_done = [], set()

    # ---- update-in-backward ------------------------------------------------------------------------------------------
    def overlap_with_backward(self, buckets):
        """The update is HBM-bound and the tail of backward (the encoders' small GEMMs) is not: let each bucket of parameters
        (e.g. one per network: decoder, integrator, E_s, E_t) be updated on a side stream as soon as ITS gradients are final,
        while backward continues.  `step()` then only handles what is left and joins.  Contract: exactly one backward pass per
        `step()` (as in the reference loop), every parameter in one param group, no gradient hooks that modify gradients after
        accumulation (so: not together with the data-parallel reducer)."""
        assert len(self.param_groups) == 1, 'overlap_with_backward: one param group'
        self._buckets = [[p for p in b if p.requires_grad] for b in buckets]
        self._buckets = [b for b in self._buckets if b]
        self._bucket_of = {id(p): i for i, b in enumerate(self._buckets) for p in b}
        self._left = [len(b) for b in self._buckets]
        self._launched = set()
        for b in self._buckets:
            for p in b:
                p.register_post_accumulate_grad_hook(self._on_grad)

    # ---- update fused into the weight-gradient GEMM ---------------------------------------------------------------
    def fuse_into_wgrad(self, params):
        """The listed 2-D weights take their Adam step inside the epilogue of their weight-gradient GEMM (functional.MLPChain ->
        ops.gemm_adam): per parameter 26 B of HBM traffic instead of 4 (gradient store) + 30 (this optimizer's pass), and the
        gradient is never materialised (`p.grad` stays None).  Contract as overlap_with_backward: one backward pass per step(),
        one gradient contribution per listed weight, no gradient all-reduce, no loss scaling; the learning rate is read when the
        GEMM is launched (a recorded step is re-recorded when it changes).  `step()` skips what was updated this way."""
        from . import functional as VF
        assert len(self.param_groups) == 1, 'fuse_into_wgrad: one param group'
        owned = {id(p) for p in self.param_groups[0]['params']}
        self._fused = [p for p in params if id(p) in owned and p.dim() == 2 and p.requires_grad]
        self._fused_done = set()
        VF.set_fused_optimizer({p: self for p in self._fused})

    def unfuse(self):
        from . import functional as VF
        self._fused, self._fused_done = [], set()
        VF.set_fused_optimizer(None)

    def can_fuse(self, p, a, b):
        return (getattr(self, '_scale_state', None) is None and p.is_contiguous() and p.dtype == torch.float32 and a.dtype == b.dtype
                and a.dtype in (torch.bfloat16, torch.float16) and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0
                and a.stride(0) % 8 == 0 and b.stride(0) % 8 == 0 and p.shape[0] % 8 == 0 and p.shape[1] % 8 == 0)

    @torch.no_grad()
    def fused_update(self, p, a, layout_a, b, layout_b, M, N, K):
        """One Adam step of `p` [M, N] with a(M, K) . b(N, K)^T as its gradient (operand conventions of ops.gemm)."""
        from . import functional as VF, ops
        group = self.param_groups[0]
        self._init_group(0, group)
        st = self.state[p]
        shadow = VF.shadow_buffer_for_update(p)
        ops.gemm_adam(a, layout_a, b, layout_b, M, N, K, p, st['exp_avg'], st['exp_avg_sq'], shadow, group['step_dev'], st.get('skipped', 0),
                      group['lr'], group['betas'], group['eps'])
        torch.autograd.graph.increment_version(p)
        if shadow is not None:
            VF.shadows_written([p])
        self._fused_done.add(id(p))

    def _on_grad(self, p):
        bi = self._bucket_of.get(id(p))
        if bi is None or bi in self._launched:
            return
        self._left[bi] -= 1
        if self._left[bi] > 0:
            return
        group = self.param_groups[0]
        self._init_group(0, group)
        from . import functional as VF
        bg = int(os.environ.get('VARSEP_ADAM_BG_BLOCKS', '512'))
        if VF.defer_call(lambda: self._update(0, group, self._buckets[bi], max_blocks=bg), late=os.environ.get('VARSEP_ADAM_EARLY_BUCKET') == '2'):
            # the bucket's weight gradients are being held back (functional.hold_deferred): the update joins that queue and runs
            # on the gradient stream right behind them; step() joins that stream like any deferred gradient work
            self._launched.add(bi)
            return
        main = torch.cuda.current_stream(p.device)
        if self._stream is None:
            self._stream = VF.own_stream(p.device)
        self._stream.wait_stream(main)
        for s in VF.side_streams_in_use():             # deferred weight gradients are produced on their own stream
            self._stream.wait_stream(s)
        with torch.cuda.stream(self._stream):
            self._update(0, group, self._buckets[bi])
        self._launched.add(bi)

    # ---- state -------------------------------------------------------------------------------------------------
    def _init_group(self, gi, group):
        """Device step counter per group + (exp_avg, exp_avg_sq) per parameter, created on first use."""
        for p in group['params']:
            st = self.state[p]
            if len(st) == 0:
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise VarsepHipError('HIP Adam needs contiguous fp32 CUDA parameters (no CPU fallback)')
                st['step'] = torch.zeros((), dtype=torch.float32, device=p.device)        # torch's capturable layout
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        if 'step_dev' not in group:
            ps = group['params']
            steps = [int(self.state[p]['step'].item()) for p in ps]
            start = max(steps) if steps else 0
            for p, t in zip(ps, steps):
                self.state[p]['skipped'] = start - t
            group['step_dev'] = torch.full((1,), start, dtype=torch.int32, device=ps[0].device)

    def _table(self, gi, plist, shadows):
        """ctypes pointer tables of one chunk of <= 64 tensors; rebuilt only when a pointer changes (a new .grad tensor after
        zero_grad(set_to_none=True) in the eager loop; stable inside a recorded graph and with flat gradient buckets)."""
        from . import functional as VF
        key = (gi, id(plist[0]), len(plist))
        gsrc = [VF.lowp_gradient(p) for p in plist]                  # bf16 wire image of the gradient, where one is registered
        gsrc = [p.grad if g is None else g for p, g in zip(plist, gsrc)]
        ptrs = tuple((p.data_ptr(), g.data_ptr(), 0 if s is None else s.data_ptr(), self.state[p].get('skipped', 0))
                     for p, g, s in zip(plist, gsrc, shadows))
        ent = self._tables.get(key)
        if ent is not None and ent[0] == ptrs:
            return ent[1]
        n = len(plist)
        VP, I64, I32 = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int32 * n
        tab = (VP(*[p.data_ptr() for p in plist]), VP(*[g.data_ptr() for g in gsrc]),
               VP(*[self.state[p]['exp_avg'].data_ptr() for p in plist]), VP(*[self.state[p]['exp_avg_sq'].data_ptr() for p in plist]),
               VP(*[None if s is None else s.data_ptr() for s in shadows]), I64(*[p.numel() for p in plist]),
               I32(*[self.state[p].get('skipped', 0) for p in plist]),
               I32(*[_lib.BF16 if g.dtype == torch.bfloat16 else _lib.F32 for g in gsrc]))
        self._tables[key] = (ptrs, tab)
        return tab

    @torch.no_grad()
    def step(self, closure=None, grad_scale_state=None, defer_increment=False):
        """`grad_scale_state` (train.LossScaler.state, fp32 [scale, found_inf, ...] on the device): gradients are used as g / scale and
        the whole step -- parameters, moments, step count -- is skipped when found_inf is set (GradScaler.step semantics)."""
        self._scale_state = grad_scale_state
        # defer_increment: the caller advances the step counter itself with finish_step() -- updates fused into weight-gradient GEMMs that are
        # still running read the counter, so it may only move once they are done
        self._defer_increment = bool(defer_increment) and grad_scale_state is None
        try:
            return self._step(closure)
        finally:
            self._scale_state = None
            self._defer_increment = False

    def _step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load_library()
        for gi, group in enumerate(self.param_groups):
            self._init_group(gi, group)
            done = set()
            if gi == 0 and self._launched:          # buckets already updated from the backward hooks
                for bi in self._launched:
                    done.update(id(p) for p in self._buckets[bi])
            fused_done = getattr(self, '_fused_done', set())
            done |= fused_done                       # updated by their weight-gradient GEMMs during backward
            live = [p for p in group['params'] if p.grad is not None and id(p) not in done]
            for p in group['params']:
                if p.grad is None and id(p) not in fused_done:   # torch counts steps per parameter: this one falls one behind the group
                    self.state[p]['skipped'] = self.state[p].get('skipped', 0) + 1
            if live:
                self._update(gi, group, live)
            main = torch.cuda.current_stream(group['params'][0].device)
            if gi == 0 and self._launched and self._stream is not None:
                main.wait_stream(self._stream)
            if getattr(self, '_scale_state', None) is not None:
                _lib.check(lib.vs_adam_step_increment_scaled(group['step_dev'].data_ptr(), self._scale_state.data_ptr(), main.cuda_stream),
                           'vs_adam_step_increment_scaled')
            elif not getattr(self, '_defer_increment', False):
                _lib.check(lib.vs_adam_step_increment(group['step_dev'].data_ptr(), main.cuda_stream), 'vs_adam_step_increment')
        if self._buckets:
            self._left = [len(b) for b in self._buckets]
            self._launched = set()
        if getattr(self, '_fused_done', None):
            self._fused_done = set()
        return loss

    @torch.no_grad()
    def step_subset(self, params):
        """The Adam update of `params` (all of one param group, all with gradients) WITHOUT advancing the step counter: the caller
        updates disjoint subsets one after the other -- e.g. one per all-reduce bucket, each as soon as its bucket has arrived --
        and finishes with `finish_step()`.  Together they are exactly `step()`."""
        for gi, group in enumerate(self.param_groups):
            self._init_group(gi, group)
            ids = {id(p) for p in group['params']}
            live = [p for p in params if id(p) in ids]
            if any(p.grad is None for p in live):
                raise VarsepHipError('step_subset needs a gradient for every listed parameter')
            if live:
                self._update(gi, group, live)

    @torch.no_grad()
    def step_ranges(self, ranges):
        """The Adam update of ELEMENT RANGES [(parameter, lo, hi)] -- this rank's slice of the parameters under a sharded optimizer
        (parallel.GradAllReducer(shard_direct=True)): fp32 master, both moments and the 16-bit operand copy of elements lo..hi-1, gradient
        from the bf16 wire image (functional.lowp_gradient).  Same kernel and arithmetic as step_subset; the step counter is not advanced
        (finish_step).  lo / hi are multiples of 8 (16-byte vector accesses on every operand)."""
        from . import functional as VF
        lib = _lib.load_library()
        if not ranges:
            return
        group = self.param_groups[0]
        self._init_group(0, group)
        stream = torch.cuda.current_stream(ranges[0][0].device).cuda_stream
        lr, (b1, b2), eps = group['lr'], group['betas'], group['eps']
        written = []
        for i in range(0, len(ranges), 64):
            chunk = ranges[i:i + 64]
            n = len(chunk)
            P, G, M_, V, S, NUM, SK, GD = [], [], [], [], [], [], [], []
            sdt = _lib.BF16
            for p, lo, hi in chunk:
                g = VF.lowp_gradient(p)
                if g is None:
                    g = p.grad
                if g is None or not (0 <= lo < hi <= p.numel()):
                    raise VarsepHipError('step_ranges: a range needs a gradient and 0 <= lo < hi <= numel')
                st = self.state[p]
                sh = VF.shadow_buffer_for_update(p)
                P.append(p.data_ptr() + 4 * lo); M_.append(st['exp_avg'].data_ptr() + 4 * lo); V.append(st['exp_avg_sq'].data_ptr() + 4 * lo)
                G.append(g.data_ptr() + g.element_size() * lo)
                GD.append(_lib.BF16 if g.dtype == torch.bfloat16 else _lib.F32)
                S.append(None if sh is None else sh.data_ptr() + 2 * lo)
                if sh is not None:
                    sdt = _lib.code_of(sh.dtype)
                    written.append(p)
                NUM.append(hi - lo); SK.append(st.get('skipped', 0))
            VP, I64, I32 = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int32 * n
            tab = (VP(*P), VP(*G), VP(*M_), VP(*V), VP(*S), I64(*NUM), I32(*SK), I32(*GD))
            self._tables[('ranges', id(chunk[0][0]), chunk[0][1], n)] = (None, tab)          # keep the tables alive (recorded launches copy them at launch)
            rc = lib.vs_adam_multi_scaled(n, ctypes.cast(tab[0], ctypes.c_void_p), ctypes.cast(tab[1], ctypes.c_void_p),
                                          ctypes.cast(tab[7], ctypes.c_void_p), ctypes.cast(tab[2], ctypes.c_void_p),
                                          ctypes.cast(tab[3], ctypes.c_void_p), ctypes.cast(tab[4], ctypes.c_void_p), sdt,
                                          ctypes.cast(tab[5], ctypes.c_void_p), ctypes.cast(tab[6], ctypes.c_void_p),
                                          group['step_dev'].data_ptr(), lr, b1, b2, eps, None, stream)
            _lib.check(rc, 'vs_adam_multi (ranges)')
        for p in {id(p): p for p, _, _ in ranges}.values():
            torch.autograd.graph.increment_version(p)
        VF.shadows_written(written)

    @torch.no_grad()
    def finish_step(self):
        lib = _lib.load_library()
        for gi, group in enumerate(self.param_groups):
            self._init_group(gi, group)
            main = torch.cuda.current_stream(group['params'][0].device)
            _lib.check(lib.vs_adam_step_increment(group['step_dev'].data_ptr(), main.cuda_stream), 'vs_adam_step_increment')

    def _update(self, gi, group, live, max_blocks=0):
        """One vs_adam_multi launch per 64 tensors of `live` on the current stream (the step counter is not touched).
        `max_blocks` > 0: a background update with that many workgroups (vs_adam_set_max_blocks)."""
        from . import functional as VF, ops
        lib = _lib.load_library()
        if max_blocks:
            prev = lib.vs_adam_set_max_blocks(int(max_blocks))
            try:
                return self._update(gi, group, live)
            finally:
                lib.vs_adam_set_max_blocks(prev)
        for p in live:
            if p.grad.dtype != torch.float32 or not p.grad.is_contiguous() or not p.grad.is_cuda:
                raise VarsepHipError('HIP Adam needs contiguous fp32 CUDA gradients')
        stream = torch.cuda.current_stream(live[0].device).cuda_stream
        lr, (b1, b2), eps = group['lr'], group['betas'], group['eps']
        scale_state = getattr(self, '_scale_state', None)
        written = []
        for i in range(0, len(live), 64):
            chunk = live[i:i + 64]
            shadows = [VF.shadow_buffer_for_update(p) for p in chunk]
            # one 16-bit type per launch: the live operand copies are all of the current compute dtype; a stray copy of the other
            # type (left from an earlier mode) is not rewritten here and refreshes itself on next use (version counter)
            kinds = [s.dtype for s in shadows if s is not None]
            sdt = _lib.code_of(kinds[0]) if kinds else _lib.BF16
            shadows = [s if (s is not None and s.dtype == kinds[0]) else None for s in shadows]
            written += [p for p, s in zip(chunk, shadows) if s is not None]
            tab = self._table(gi, chunk, shadows)
            e0 = ops._pb()
            rc = lib.vs_adam_multi_scaled(len(chunk), ctypes.cast(tab[0], ctypes.c_void_p), ctypes.cast(tab[1], ctypes.c_void_p),
                                          ctypes.cast(tab[7], ctypes.c_void_p), ctypes.cast(tab[2], ctypes.c_void_p),
                                          ctypes.cast(tab[3], ctypes.c_void_p), ctypes.cast(tab[4], ctypes.c_void_p), sdt,
                                          ctypes.cast(tab[5], ctypes.c_void_p), ctypes.cast(tab[6], ctypes.c_void_p),
                                          group['step_dev'].data_ptr(), lr, b1, b2, eps,
                                          None if scale_state is None else scale_state.data_ptr(), stream)
            _lib.check(rc, 'vs_adam_multi')
            if e0 is not None:
                # algorithmic bytes: p, m, v read + written (24 B), the gradient read (4 or 2 B), the 16-bit operand copy (2 B)
                nb = sum(p.numel() * (24 + (2 if VF.lowp_gradient(p) is not None else 4) + (2 if s is not None else 0))
                         for p, s in zip(chunk, shadows))
                ops._pe(e0, 'vs_adam_multi', nbytes=float(nb))
        for p in live:
            # the kernel wrote through raw pointers: tell autograd / the operand caches that the parameter changed
            torch.autograd.graph.increment_version(p)
        VF.shadows_written(written)

    def state_dict(self):
        # keep the per-parameter `step` entries (torch layout) in sync with the device counter before serialising
        for group in self.param_groups:
            if 'step_dev' in group:
                t = float(group['step_dev'].item())
                for p in group['params']:
                    if p in self.state and 'step' in self.state[p]:
                        self.state[p]['step'].fill_(t - self.state[p].get('skipped', 0))
        sd = super().state_dict()
        for g in sd['param_groups']:
            g.pop('step_dev', None)
        # super().state_dict() shares the live per-parameter dicts: strip the private entry from COPIES
        sd['state'] = {k: {kk: vv for kk, vv in st.items() if kk != 'skipped'} for k, st in sd['state'].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for group in self.param_groups:
            group.pop('step_dev', None)              # re-created from the loaded per-parameter step on the next step()
        self._tables = {}
