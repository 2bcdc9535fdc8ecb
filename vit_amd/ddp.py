"""Data-parallel gradient exchange: the replacement for the reference's
``torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank])`` (train.py:635,
utils/ops.py:124-133) on one 8 x MI355X node: one process per GPU, RCCL over xGMI
(torch.distributed backend "nccl" on ROCm; "gloo" for CPU rehearsals).

Design (MI355X first, not a DDP translation):
  * gradients LIVE in a few large flat fp32 slabs: every used parameter's ``.grad`` is a view into
    its bucket.  The fused blocks' weight-gradient GEMMs write straight into those views
    (functional.grad_slot: a zeroed view, handed out once per step); any other gradient is
    copied into its view by the post-accumulate hook -- one copy, where accumulating into
    pre-zeroed views cost a zero-fill plus a read-modify-write of the whole gradient;
  * buckets follow REVERSE registration order (the order backward produces gradients) and are
    sized for xGMI: 7 point-to-point links x ~153 GB/s per GPU make a ring all-reduce per-link
    bound, so few, large (default 128 MiB) collectives amortise latency while still leaving
    ~10 buckets per step to overlap with the remaining backward;
  * a bucket's all-reduce is issued from the post-accumulate hook of its LAST parameter, on
    RCCL's own stream (async_op), overlapping the rest of backward; ``finish()`` makes the
    compute stream wait for all of them and applies 1/world;
  * parameters that never receive a gradient (23 in the disentangled models: pos_embed, out.*,
    pos_key/query_proj, the layer-level q/k/v_proj -- SURVEY.md section 7) are found on the first
    step and left with ``grad = None`` so AdamW skips them exactly as in the reference; stock DDP
    would raise on them.
"""
import os

import torch
import torch.distributed as dist


def init_process_group(backend='nccl', device=None, **kw):
    """torch.distributed.init_process_group (env:// rendezvous, as utils/ops.py:124-133 of the reference) with RCCL's stream
    taken from the HIGH-PRIORITY class.  HIP runs its streams on a handful of hardware queues and streams that share one
    run in turns: with a normal-priority RCCL stream the collectives landed on the main stream's queue and every bucket's
    reduction ran IN LINE with backward instead of beside it (rocprofv3 trace, profiles/r03_reducer_*).  Priority classes
    have queues of their own, and functional's side stream stays in the normal class.

    Only for a process that has not run the fused blocks yet (the reference's order too: init_distributed_mode, then the
    model, train.py:631-635).  A high-priority RCCL stream made AFTER the process had trained on its main and side streams
    left the next step's kernels pending for good (torch.cuda.synchronize() never returned; tools/diag_reducer_hang.py:
    main stream, side stream and every bucket pending -- with a normal-priority RCCL stream the same sequence runs).  The
    cause inside the runtime is not known; such a process gets the normal class, and the side stream is chosen again so
    that the overlap probe sees the queues as they are now.  VITK_RCCL_HIGH_PRIORITY=0 / =1 forces either."""
    if backend == 'nccl':
        if device is not None:
            kw.setdefault('device_id', device)
        from . import functional as VF
        want = os.environ.get('VITK_RCCL_HIGH_PRIORITY')
        high = (want != '0') if want in ('0', '1') else not VF._side_streams
        if not high and VF._side_streams:
            VF.forget_side_streams()
        if high:
            try:
                opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
                return dist.init_process_group(backend, pg_options=opts, **kw)
            except (AttributeError, TypeError):
                pass
    return dist.init_process_group(backend, **kw)


class GradReducer:
    def __init__(self, module, bucket_mb=128, process_group=None, broadcast=True, comm_dtype=torch.float32,
                 always_reduce=False, reserve_cus=None):
        if not dist.is_initialized():
            raise RuntimeError('GradReducer needs torch.distributed to be initialised')
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.bucket_bytes = int(bucket_mb * 1024 * 1024)
        self.built = False
        self.buckets = []        # flat fp32 tensors
        self._pending = []       # per bucket: parameters still missing this step
        self._bucket_of = {}
        self._view_of = {}
        self._handles = []
        self._hooks = []
        self._copies = []        # per bucket: (view, gradient) pairs waiting for the bucket's one multi-tensor copy
        self._avg = dist.ReduceOp.AVG if dist.get_backend(process_group) == 'nccl' else dist.ReduceOp.SUM
        # comm_dtype: what travels over xGMI.  fp32 (default) reduces the slabs in place, bit-compatible with the
        # reference's DDP; bf16 halves the bytes per step (SURVEY 8(e) budgets 612 MB for the B preset instead of
        # 1.22 GB) at the price of one rounding of each rank's gradient to 8 significant bits before the sum
        # (two cast passes over the bucket at HBM speed).
        if comm_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError('GradReducer: comm_dtype must be torch.float32 or torch.bfloat16')
        self.comm_dtype = comm_dtype
        # always_reduce: issue the collectives at world size 1 too (they are copies there): lets one GPU rehearse the whole
        # exchange -- RCCL's stream, the async handles, their capture into a HIP graph -- as a run on N GPUs would do it
        self.always_reduce = bool(always_reduce)
        self._stage = {}
        # reserve_cus: CUs the persistent GEMM launches leave free while collectives are in flight (vitk_set_reserved_cus).
        # A gemm256 workgroup owns its CU's whole LDS until it has walked all its tiles, so on a fully taken chip RCCL's
        # reduction kernels would wait for a GEMM to end, bucket after bucket.  None: leave the process-wide setting.
        if reserve_cus is not None and (self.world > 1 or self.always_reduce):
            from . import ops
            ops.set_reserved_cus(reserve_cus)
        if broadcast and self.world > 1:
            self.broadcast_parameters(module)

    def broadcast_parameters(self, module):
        """Rank 0's weights and buffers to every rank (what DDP's constructor does), one flat
        broadcast per dtype instead of one per tensor."""
        with torch.no_grad():
            # the tensors themselves, not .data: copy_() then bumps their version counters, which is what invalidates
            # the cached compute-dtype copies (functional.shadow) of a model that has already run a forward
            tensors = list(module.parameters()) + list(module.buffers())
            by_type = {}
            for t in tensors:
                by_type.setdefault((t.dtype, t.device), []).append(t)
            for ts in by_type.values():
                flat = torch.cat([t.reshape(-1) for t in ts])
                dist.broadcast(flat, 0, group=self.group)
                off = 0
                for t in ts:
                    t.copy_(flat[off:off + t.numel()].view_as(t))
                    off += t.numel()

    # ---------------------------------------------------------------- bucket construction
    def _build(self):
        used = [p for p in self.params if p.grad is not None]
        order = list(reversed(used))
        groups, cur, cur_bytes = [], [], 0
        for p in order:
            nbytes = p.numel() * 4
            if cur and cur_bytes + nbytes > self.bucket_bytes:
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            groups.append(cur)
        for bi, ps in enumerate(groups):
            n = sum((p.numel() + 63) // 64 * 64 for p in ps)    # 256-byte aligned views
            flat = torch.zeros(n, device=ps[0].device, dtype=torch.float32)
            off = 0
            for p in ps:
                view = flat[off:off + p.numel()].view_as(p)
                view.copy_(p.grad)
                p.grad = view
                self._view_of[p] = view
                self._bucket_of[p] = bi
                off += (p.numel() + 63) // 64 * 64
            self.buckets.append(flat)
        self._groups = groups
        for p in used:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.built = True

    def _on_grad(self, p):
        view = self._view_of[p]
        g = p.grad
        bi = self._bucket_of[p]
        if g is not view and g.data_ptr() != view.data_ptr():     # produced elsewhere: into the slab, once
            # not one copy kernel per parameter (a B-preset step made 349 of them, 5 us each, in line on the main
            # stream): the pairs of a bucket go out as ONE multi-tensor copy right before its all-reduce
            self._copies[bi].append((view, g))
            p.grad = view
        p._vitk_grad_slot = None
        pend = self._pending[bi]
        pend.discard(p)
        if not pend and self._launched is not None and not self._launched[bi]:
            self._launch(bi)

    def _flush_copies(self, bi):
        pairs = self._copies[bi]
        if pairs:
            same = [(v, g) for v, g in pairs if g.dtype == v.dtype and g.shape == v.shape]
            if same:
                torch._foreach_copy_([v for v, _ in same], [g for _, g in same])
            for v, g in pairs:
                if not (g.dtype == v.dtype and g.shape == v.shape):
                    v.copy_(g)
            pairs.clear()

    def _launch(self, bi):
        self._launched[bi] = True
        self._flush_copies(bi)
        if self.world > 1 or self.always_reduce:
            flat = self.buckets[bi]
            if self.comm_dtype != torch.float32:
                stage = self._stage.get(bi)
                if stage is None:
                    stage = self._stage[bi] = torch.empty_like(flat, dtype=self.comm_dtype)
                stage.copy_(flat)
                flat = stage
            self._handles.append((bi, dist.all_reduce(flat, op=self._avg, group=self.group, async_op=True)))

    # ---------------------------------------------------------------- per-step protocol
    def prepare(self):
        """Call instead of optimizer.zero_grad(), before backward."""
        self._handles = []
        if not self.built:
            for p in self.params:
                p.grad = None
            self._launched = None
            return
        # No slab-wide zero-fill: every view is either written whole by the hook's copy or zeroed here because
        # a GEMM will accumulate into it (the slot protocol of functional.grad_slot); the pads were zeroed at build.
        direct = []
        for ps in self._groups:
            for p in ps:
                p.grad = None
                if getattr(p, '_vitk_direct', False):       # set by the fused blocks on the weights they can write in place
                    view = self._view_of[p]
                    direct.append(view)
                    p._vitk_grad_slot = view
        if direct:
            torch._foreach_zero_(direct)                    # a few multi-tensor fills instead of one kernel per weight
        self._pending = [set(ps) for ps in self._groups]
        self._copies = [[] for _ in self._groups]
        self._launched = [False] * len(self.buckets)

    def finish(self):
        """Call after backward, before optimizer.step()."""
        if not self.built:
            self._build()
            self._pending = [set() for _ in self.buckets]
            self._launched = [False] * len(self.buckets)
            self._copies = [[] for _ in self.buckets]
        for bi in range(len(self.buckets)):
            if not self._launched[bi]:       # a parameter got no gradient this step: reduce zeros for it
                for p in self._pending[bi]:
                    self._view_of[p].zero_()
                    p.grad = self._view_of[p]
                    p._vitk_grad_slot = None
                self._launch(bi)
        for bi, h in self._handles:
            h.wait()
            if self.comm_dtype != torch.float32:
                self.buckets[bi].copy_(self._stage[bi])
        self._handles = []
        if self.world > 1 and self._avg != dist.ReduceOp.AVG:
            for flat in self.buckets:
                flat.mul_(1.0 / self.world)

    def bytes_per_step(self):
        """Bytes each rank hands to the collectives per step."""
        esz = 4 if self.comm_dtype == torch.float32 else 2
        return sum(b.numel() * esz for b in self.buckets)
