"""Batch-sharded data parallelism: one process per GPU, gradient averaging over RCCL (xGMI) -- new functionality with
no reference counterpart.  Correctness contract: N replicas on equal-sized shards with
averaged gradients take exactly the step a single process would take on the concatenated batch's mean losses
(BatchNorm stays per replica, as the reference has no SyncBN).

Gradients live in a few large flat fp32 buckets (`param.grad` are views into them), so an all-reduce moves one
contiguous buffer per bucket instead of one small tensor per parameter: MI355X's xGMI links are point-to-point
(7 x ~153 GB/s per GPU) and RCCL's ring/direct algorithms only reach link rate on multi-megabyte messages.  Buckets
are filled in reverse registration order (t_resnet, decoder, Et, Es), the order in which backward finishes them; each
bucket's all-reduce is issued on a side HIP stream as soon as the last gradient of the bucket has been accumulated,
overlapping the remaining backward kernels, and `all_reduce()` only waits for completion.
"""
import torch
import torch.distributed as dist


class GradAllReducer:
    def __init__(self, params, bucket_bytes=64 << 20, process_group=None, overlap=True, force=False, comm_dtype=torch.float32,
                 lowp_direct=None, early=None, shard_direct=False, stacked=None, shard_tail_dtype=torch.float32):
        self.params = [p for p in params if p.requires_grad]
        # `early`: parameters whose gradients are complete long before the end of backward (the decoder's: it is the first thing backward
        # finishes).  They get leading buckets of their own (`early_buckets`), so that their all-reduce can start -- from the hook of the last
        # of them in the eager loop, between the two segments of a recorded step (train.GraphedStep) -- while the rest of backward runs.
        self._early_ids = {id(p) for p in (early or [])}
        self.early_buckets = []
        self.group = process_group
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.backend = dist.get_backend(process_group) if dist.is_initialized() else None
        self.buckets = []            # (flat, [params])
        self._pending = {}
        self._handles = []
        self._force = force             # run the collectives even at world size 1 (exercises the RCCL path on one GPU)
        # comm_dtype=torch.bfloat16: the bucket is rounded to bf16 for the wire (half the xGMI bytes: the all-reduce of the WaveEq
        # model's 243 MB of fp32 gradients is as long as the whole compute step) and averaged in bf16 -- the usual gradient
        # compression of bf16 training; fp32 (default) keeps the N-replica step equal to the single-process step.
        assert comm_dtype in (torch.float32, torch.bfloat16)
        self.comm_dtype = comm_dtype
        self._wire = {}
        # lowp_direct (bf16 wire only): parameters whose producer can write the bf16 wire image of their gradient itself (the
        # weight-gradient GEMMs of the Linear chains in the recorded step).  They are laid out FIRST in their bucket; while
        # `direct_lowp` is switched on (train.GraphedStep), only the small tail of the bucket is cast to / from the wire format
        # and the optimizer reads the averaged bf16 image of the others (`lowp_views`).
        self._direct_ids = {id(p) for p in (lowp_direct or [])} if comm_dtype != torch.float32 else set()
        self.lowp_views = {}          # id(param) -> bf16 view of the wire buffer
        self._tail = {}               # flat.data_ptr() -> first element of the not-direct region
        self.direct_lowp = False
        # shard_direct (bf16 wire + lowp_direct, recorded step only -- train.GraphedStep): the direct parameters' gradients are
        # REDUCE-SCATTERED instead of all-reduced, every rank runs Adam on ITS 1/N slice of their fp32 masters / moments and rewrites its slice
        # of their 16-bit operand copies, which live in one flat arena per bucket and are ALL-GATHERED afterwards (see "Sharded optimizer"
        # below).  Wire bytes as the all-reduce (RS + AG of 2 B per parameter), optimizer traffic and time 1/N of the replicated update.
        self.shard = bool(shard_direct) and bool(self._direct_ids)
        # the small tail (biases, integrator) of a sharded bucket is summed in fp32 by default; torch.bfloat16 rounds it before and after
        # the sum like the replicated bf16-wire path does -- then the sharded step is BIT-identical to the replicated one (the GPU tests use it)
        self.shard_tail_dtype = shard_tail_dtype
        # presummed: the caller seeds its loss gradient with 1 / world_size (train.GraphedStep), so gradients only have to be SUMMED: same
        # arithmetic as ReduceOp.AVG (RCCL pre-multiplies by 1 / N and sums; exact for N a power of two), but no pre-multiply kernel -- and at
        # world size 1 (the forced N > 1 path) an in-place sum is no launch at all, where AVG runs a 32-workgroup copy at 0.25 TB/s
        self.presummed = False
        self.masters_dirty = False       # fp32 masters / moments of the direct parameters are current in the own slice only
        self._head = {}                  # bucket index -> (padded head length, [(param, offset)])
        self._arena = {}                 # bucket index -> flat 16-bit arena of the direct parameters' operand copies
        self._rs_tmp, self._ag_tmp = {}, {}   # N > 1: receive / send buffers of one slice per bucket
        # stacked: lists of same-shaped parameters whose gradients must lie back to back in a bucket, in the given order, so that ONE batched
        # launch can write all of them (the integrator's per-layer weights: functional.MLPRollout.backward writes [blocks, H, C] in place)
        self._stacked = [list(g) for g in (stacked or [])]
        self._build(bucket_bytes)
        self._comm_stream = None
        self._held = set()               # buckets that hooks must not launch (hold_params)
        self._overlap = overlap and (self.world_size > 1 or force)
        if self._overlap:
            self._install_hooks()

    def _build(self, bucket_bytes):
        order = list(reversed(self.params))              # reverse order = order gradients become final
        groups = [[p for p in order if id(p) in self._early_ids], [p for p in order if id(p) not in self._early_ids]]
        for gi, group in enumerate(groups):
            cur, cur_bytes = [], 0
            for p in group:
                cur.append(p)
                cur_bytes += p.numel() * 4
                if cur_bytes >= bucket_bytes:
                    self._finish_bucket(cur, early=gi == 0)
                    cur, cur_bytes = [], 0
            if cur:
                self._finish_bucket(cur, early=gi == 0)

    def _finish_bucket(self, plist, early=False):
        if early:
            self.early_buckets.append(len(self.buckets))
        direct = [p for p in plist if id(p) in self._direct_ids]
        rest = [p for p in plist if id(p) not in self._direct_ids]
        # stacked groups whose members are all in this bucket: moved to the front of the not-direct region, members adjacent and in order
        in_rest = {id(p) for p in rest}
        front = []
        for g in self._stacked:
            if g and all(id(p) in in_rest for p in g):
                front += g
        taken = {id(p) for p in front}
        rest = front + [p for p in rest if id(p) not in taken]
        plist = direct + rest
        # the direct (head) region is padded so that it splits into world_size slices of whole 128-byte lines (reduce-scatter / all-gather)
        head = sum(p.numel() for p in direct)
        if self.shard and head:
            q = self.world_size * 64
            head = (head + q - 1) // q * q
        total = head + sum(p.numel() for p in rest) if (self.shard and direct) else sum(p.numel() for p in plist)
        flat = torch.zeros(total, dtype=torch.float32, device=plist[0].device)
        wire = torch.zeros(total, dtype=self.comm_dtype, device=plist[0].device) if self._direct_ids else None
        off = 0
        self._tail[flat.data_ptr()] = total
        heads = []
        for p in plist:
            if id(p) not in self._direct_ids and self._tail[flat.data_ptr()] == total:
                if self.shard and direct:
                    off = head
                self._tail[flat.data_ptr()] = off
            p.grad = flat[off:off + p.numel()].view_as(p)
            if id(p) in self._direct_ids:
                self.lowp_views[id(p)] = wire[off:off + p.numel()].view_as(p)
                heads.append((p, off))
            off += p.numel()
        if self.shard and direct:
            if self._tail[flat.data_ptr()] == total:
                self._tail[flat.data_ptr()] = head if rest else total
            self._head[len(self.buckets)] = (head, heads)
        if wire is not None:
            self._wire[flat.data_ptr()] = wire
        self.buckets.append((flat, list(plist)))

    def zero_grad(self):
        """Replaces optimizer.zero_grad(): keeps the views alive and zeroes one flat buffer per bucket."""
        for flat, plist in self.buckets:
            flat.zero_()
        for bi, (_, plist) in enumerate(self.buckets):
            self._pending[bi] = len(plist)
        self._handles = []

    # ---- overlap: fire a bucket's all-reduce from the hook of its last-arriving gradient ----------------------
    def _install_hooks(self):
        self._bucket_of = {}
        for bi, (_, plist) in enumerate(self.buckets):
            for p in plist:
                self._bucket_of[id(p)] = bi
                p.register_post_accumulate_grad_hook(self._hook)

    def _hook(self, p):
        bi = self._bucket_of[id(p)]
        left = self._pending.get(bi)
        if left is None:
            return
        left -= 1
        self._pending[bi] = left
        if left == 0 and bi not in self._held:
            self._launch(bi)

    def hold_params(self, params):
        """The buckets of `params` are reduced by `all_reduce()` only, never from a gradient hook: for parameters whose gradient is
        completed AFTER autograd has visited them -- the batched weight gradients of repeatedly applied stride-1 convolutions are
        computed by an end-of-backward callback and added straight into the bucket (functional.set_conv_grad_outputs), while the
        post-accumulate hooks of such parameters fire earlier (autograd calls them even when it was handed no gradient)."""
        ids = {id(p) for p in params}
        for bi, (_, plist) in enumerate(self.buckets):
            if any(id(p) in ids for p in plist):
                self._held.add(bi)

    def _launch(self, bi):
        flat = self.buckets[bi][0]
        if flat.is_cuda:
            if self._comm_stream is None:
                from . import functional as VF
                self._comm_stream = VF.own_stream(flat.device)
            self._comm_stream.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(self._comm_stream):
                self._reduce(flat)
        else:
            self._reduce(flat)
        self._pending[bi] = None

    def _reduce_guard(self, device):
        """The exchange guard word (ops.exchange_guard: raised by a timed-out in-launch exchange, read by every optimizer launch) is
        OR-reduced with every bucket: a replica whose integrator timed out has put garbage on the wire, so ALL replicas must skip the
        update, or they would drift apart (one tiny collective per bucket on the comm stream, N > 1 only).  The word is a bit mask (bit 0:
        MLP integrator, bit 1: one-launch ConvResBlock layer): OR keeps rank A's bit beside rank B's, MAX would drop the smaller one."""
        if self.world_size == 1 or device.type != 'cuda':
            return
        from . import ops
        word = ops.exchange_guard(device)
        if word is None:
            return
        if self.backend == 'nccl':
            dist.all_reduce(word, op=dist.ReduceOp.BOR, group=self.group)
        else:
            host = word.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.BOR, group=self.group)
            word.copy_(host)

    def _reduce(self, flat):
        self._reduce_guard(flat.device)
        if self.shard and self.direct_lowp:
            return self._reduce_sharded(flat)
        if self.comm_dtype != torch.float32:
            wire = self._wire.get(flat.data_ptr())
            if wire is None:
                wire = self._wire[flat.data_ptr()] = torch.empty_like(flat, dtype=self.comm_dtype)
            # with direct_lowp the head of the bucket already IS in wire format (written by the gradient GEMMs) and is consumed
            # in wire format: only the tail (biases, integrator parameters) is converted
            t0 = self._tail.get(flat.data_ptr(), 0) if self.direct_lowp else 0
            src, dst = (flat[t0:], wire[t0:]) if t0 else (flat, wire)
            if src.numel():
                dst.copy_(src)                                 # fp32 -> bf16 (round to nearest even)
            div = 1 if self.presummed else self.world_size
            if self.backend == 'nccl':
                dist.all_reduce(wire, op=dist.ReduceOp.SUM if self.presummed else dist.ReduceOp.AVG, group=self.group)
            else:                                              # gloo (tests): no bf16 reduction there -- sum the bf16 values in fp32
                host = wire.float().cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                wire.copy_((host / div).to(self.comm_dtype))
            if src.numel():
                src.copy_(dst)
            return
        div = 1 if self.presummed else self.world_size
        if self.backend == 'nccl':
            dist.all_reduce(flat, op=dist.ReduceOp.SUM if self.presummed else dist.ReduceOp.AVG, group=self.group)
        elif flat.is_cuda:
            # gloo with device buckets (tests: several ranks sharing one GPU): staged through host memory
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            flat.copy_(host.div_(div))
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            flat.div_(div)

    def all_reduce(self):
        """Average gradients over ranks; returns when the current stream may consume them."""
        if self.world_size == 1 and not self._force:
            return
        for bi in range(len(self.buckets)):
            if self._pending.get(bi, len(self.buckets[bi][1])) is not None:      # not launched by a hook
                self._launch(bi)
        if self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)

    # ---- graph-replay mode: the backward pass is a recorded hipGraph, so no hooks fire -----------------------------
    def zero_buffers(self):
        """Zero the flat buckets without arming the hooks (capturable: one memset per bucket)."""
        for flat, _ in self.buckets:
            # with direct_lowp the head of a bucket is never read in fp32 (its gradients are written whole, in wire format, by the
            # producing GEMMs every step): only the tail accumulates and needs zeros
            t0 = self._tail.get(flat.data_ptr(), 0) if self.direct_lowp else 0
            if t0 < flat.numel():
                (flat[t0:] if t0 else flat).zero_()
        self._pending = {}

    def reduce_all(self):
        """All-reduce every bucket now (comm stream ordered after the current stream) and make the current stream wait."""
        if self.world_size == 1 and not self._force:
            return
        for bi in range(len(self.buckets)):
            self._launch(bi)
        if self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)

    def reduce_bucket(self, bi):
        """Start the all-reduce of bucket `bi` on the comm stream (ordered after the current stream) and return an event that is
        recorded when its averaged gradients are in place; None when nothing runs (single rank).  Buckets issued in order are
        reduced in order, so a consumer that waits on event i can start while buckets i+1.. are still on the wire."""
        if self.world_size == 1 and not self._force:
            return None
        self._launch(bi)
        if self._comm_stream is None:
            return None
        ev = torch.cuda.Event()
        ev.record(self._comm_stream)
        return ev

    # ---- Sharded optimizer (MLP family, bf16 wire, recorded step) -----------------------------------------------------------------
    # No reference counterpart (the reference is single-GPU).  Per bucket and step:
    #   backward     : the chains' weight-gradient GEMMs round their result once into the bf16 wire buffer (head region, as before); the loss
    #                  gradient is seeded with 1 / world_size (train.GraphedStep), so SUMS over ranks are averages -- no AVG pre-multiply pass
    #   reduce-scatter (SUM, in place): rank r receives the summed slice r of the head; the small tail (biases, integrator) is all-reduced
    #   Adam         : optim.Adam.step_ranges on the element ranges of the direct parameters inside slice r (fp32 master, exp_avg, exp_avg_sq:
    #                  1/N of the replicated update's 26 B per parameter) + the replicated update of the tail
    #   all-gather   : the 16-bit operand copies (what forward / backward read) live in one arena per bucket; every rank contributes slice r
    # fp32 masters and moments of the direct parameters are current in the OWN slice only (`masters_dirty`): `sync_masters()` before anything
    # reads them (checkpoint, an eager step, evaluation in fp32).
    def _slice(self, bi):
        head, _ = self._head[bi]
        n = head // self.world_size
        return self.rank * n, (self.rank + 1) * n

    def _reduce_sharded(self, flat):
        bi = next(i for i, (f, _) in enumerate(self.buckets) if f.data_ptr() == flat.data_ptr())
        wire = self._wire[flat.data_ptr()]
        t0 = self._tail.get(flat.data_ptr(), flat.numel())
        head = self._head.get(bi, (0, []))[0]
        if head:
            lo, hi = self._slice(bi)
            if self.world_size == 1 and self.backend == 'nccl':
                dist.reduce_scatter_tensor(wire[lo:hi], wire[:head], op=dist.ReduceOp.SUM, group=self.group)      # in place: no launch at all
            elif self.world_size == 1:
                pass                                       # gloo, one rank: the slice is the head and already holds the sums
            else:
                # N > 1, either backend: the slice is received in a buffer of its own (1 / N of the head) and copied into place -- no aliasing of
                # a collective's input and output.  The buffer handling is shared, only the collective differs, so the two-rank gloo tests on
                # device tensors (tests/test_ddp_gpu.py) run these lines although RCCL itself has never seen more than one rank here.
                tmp = self._rs_tmp.get(bi)
                if tmp is None:
                    tmp = self._rs_tmp[bi] = torch.empty(hi - lo, dtype=wire.dtype, device=wire.device)
                if self.backend == 'nccl':
                    dist.reduce_scatter_tensor(tmp, wire[:head], op=dist.ReduceOp.SUM, group=self.group)
                else:                                      # gloo (tests): no reduce-scatter / 16-bit arithmetic there -- fp32 sums on the host
                    host = wire[:head].float().cpu()
                    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                    tmp.copy_(host[lo:hi].to(self.comm_dtype))
                    wire[:head].fill_(float('nan'))        # what a reduce-scatter leaves outside the slice is undefined: nobody may read it
                wire[lo:hi].copy_(tmp)
        if t0 < flat.numel():                              # tail: small fp32 gradients, summed in fp32 (seeded with 1 / world_size)
            tail = flat[t0:]
            lowp = self.shard_tail_dtype != torch.float32
            if lowp:
                tail.copy_(tail.to(self.shard_tail_dtype).float())
            if self.backend == 'nccl':
                dist.all_reduce(tail, op=dist.ReduceOp.SUM, group=self.group)
            elif self.world_size > 1:
                host = tail.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                tail.copy_(host)
            if lowp:
                tail.copy_(tail.to(self.shard_tail_dtype).float())

    def shard_ranges(self, bi):
        """[(parameter, lo, hi)]: the element ranges of bucket `bi`'s direct parameters that lie in this rank's slice."""
        if bi not in self._head:
            return []
        lo, hi = self._slice(bi)
        out = []
        for p, off in self._head[bi][1]:
            a, b = max(lo, off), min(hi, off + p.numel())
            if a < b:
                out.append((p, a - off, b - off))
        return out

    def tail_params(self, bi):
        return [p for p in self.buckets[bi][1] if id(p) not in self._direct_ids]

    def adopt_operand_copies(self, dtype, register=None):
        """Move the direct parameters' 16-bit operand copies into one flat arena per bucket (same layout as the head of the wire buffer).
        `register(parameter, view)`: what makes the view THE operand copy (default functional.adopt_shadow; the CPU tests pass their own)."""
        if register is None:
            from . import functional as VF
            register = VF.adopt_shadow
        for bi, (head, heads) in self._head.items():
            arena = self._arena.get(bi)
            if arena is None or arena.dtype != dtype:
                arena = self._arena[bi] = torch.zeros(head, dtype=dtype, device=heads[0][0].device)
            for p, off in heads:
                register(p, arena[off:off + p.numel()].view_as(p))

    def gather_operand_copies(self, bi):
        """All-gather the arena of bucket `bi` on the comm stream, behind the current stream (the Adam launch that wrote this rank's slice)."""
        arena = self._arena.get(bi)
        if arena is None:
            return
        lo, hi = self._slice(bi)
        if arena.is_cuda:
            if self._comm_stream is None:
                from . import functional as VF
                self._comm_stream = VF.own_stream(arena.device)
            self._comm_stream.wait_stream(torch.cuda.current_stream(arena.device))
            ctx = torch.cuda.stream(self._comm_stream)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        with ctx:
            if self.world_size == 1:
                if self.backend == 'nccl':
                    dist.all_gather_into_tensor(arena, arena[lo:hi], group=self.group)              # in place
            else:
                tmp = self._ag_tmp.get(bi)                                                          # (as above: the contribution from a copy of the slice)
                if tmp is None:
                    tmp = self._ag_tmp[bi] = torch.empty(hi - lo, dtype=arena.dtype, device=arena.device)
                tmp.copy_(arena[lo:hi])
                if self.backend == 'nccl':
                    dist.all_gather_into_tensor(arena, tmp, group=self.group)
                else:
                    arena.fill_(float('nan'))                        # (an all-gather defines every element: whatever survives this is a bug)
                    mine = tmp.view(torch.int32).cpu()               # (gloo moves no 16-bit types; a slice is a whole number of 128-byte lines)
                    parts = [torch.empty_like(mine) for _ in range(self.world_size)]
                    dist.all_gather(parts, mine, group=self.group)
                    arena.view(torch.int32).copy_(torch.cat(parts))

    def wait_comm(self):
        if self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)

    def sync_masters(self, optimizer):
        """Every rank receives the other ranks' slices of the direct parameters' fp32 masters and Adam moments (checkpoint, eager step)."""
        if not self.shard or not self.masters_dirty:
            return
        self.wait_comm()
        for bi, (head, heads) in self._head.items():
            lo, hi = self._slice(bi)
            tensors = [[p.data for p, _ in heads]]
            for key in ('exp_avg', 'exp_avg_sq'):
                if all(key in optimizer.state.get(p, {}) for p, _ in heads):
                    tensors.append([optimizer.state[p][key] for p, _ in heads])
            for group in tensors:
                full = torch.zeros(head, dtype=torch.float32, device=heads[0][0].device)
                for (p, off), t in zip(heads, group):
                    a, b = max(lo, off), min(hi, off + p.numel())
                    if a < b:
                        full[a:b] = t.reshape(-1)[a - off:b - off]
                if self.world_size > 1:
                    if self.backend == 'nccl':
                        dist.all_reduce(full, op=dist.ReduceOp.SUM, group=self.group)
                    else:
                        host = full.cpu()
                        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                        full.copy_(host)
                    for (p, off), t in zip(heads, group):
                        t.reshape(-1).copy_(full[off:off + p.numel()])
        self.masters_dirty = False

    def fill_masters_from_arena(self):
        """No collective possible (Ctrl-C reached one rank only: train.train's final checkpoint): complete the direct parameters' fp32 masters
        OUTSIDE this rank's slice from the all-gathered 16-bit operand copies -- the current weights of every slice, rounded to the compute
        type -- instead of leaving what the last sync_masters() put there (stale by every step since).  The own slice keeps its fp32 values;
        the Adam moments outside the slice stay incomplete (`masters_dirty` stays set: a later sync_masters() completes both exactly).
        Returns True when something was filled in."""
        if not self.shard or not self.masters_dirty:
            return False
        self.wait_comm()
        done = False
        with torch.no_grad():
            for bi, (head, heads) in self._head.items():
                arena = self._arena.get(bi)
                if arena is None:
                    continue
                lo, hi = self._slice(bi)
                for p, off in heads:
                    n = p.numel()
                    flat = p.data.reshape(-1)
                    a, b = min(max(lo, off), off + n), max(min(hi, off + n), off)       # own range [a, b) in bucket coordinates (empty: a >= b)
                    if a >= b:
                        flat.copy_(arena[off:off + n].float())
                    else:
                        if a > off:
                            flat[:a - off].copy_(arena[off:a].float())
                        if b < off + n:
                            flat[b - off:].copy_(arena[b:off + n].float())
                    done = True
        return done

    def payload_bytes(self):
        return sum(flat.numel() * 4 for flat, _ in self.buckets)


def broadcast_module_state(module, src=0, process_group=None):
    """Make every replica start from rank `src`'s parameters and buffers (BN running statistics included)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    staged = dist.get_backend(process_group) != 'nccl'
    for t in list(module.parameters()) + list(module.buffers()):
        if staged and t.is_cuda:
            host = t.data.cpu()
            dist.broadcast(host, src=src, group=process_group)
            t.data.copy_(host)
        else:
            dist.broadcast(t.data, src=src, group=process_group)


def broadcast_buffers(module, src=0, process_group=None):
    """Rank `src`'s buffers (BatchNorm running statistics, num_batches_tracked) to every replica: called at checkpoint time, the
    replicas' statistics are per-replica in between (no SyncBN in the reference)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    staged = dist.get_backend(process_group) != 'nccl'
    for t in module.buffers():
        if staged and t.is_cuda:
            host = t.data.cpu()
            dist.broadcast(host, src=src, group=process_group)
            t.data.copy_(host)
        else:
            dist.broadcast(t.data, src=src, group=process_group)


def broadcast_seed(seed, src=0, process_group=None):
    """One seed for every rank (rank `src`'s): the shuffle permutation of the DistributedSampler and the global NumPy stream that
    draws `t_random` (train.py:72-75) must agree across replicas."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return int(seed)
    box = [int(seed)]
    dist.broadcast_object_list(box, src=src, group=process_group)
    return int(box[0])
