"""AdamW over the HIP library: the optimizer either side of the hot path.

Reference: ``optimizer = optim.AdamW(model.parameters(), lr=args.lr)`` (train.py:275), stepped once per
batch (train.py:549) -- torch.optim.AdamW with torch's defaults (betas 0.9 / 0.999, eps 1e-8,
weight_decay 1e-2, no amsgrad).  Same constructor surface, same ``param_groups`` (so
``StepLR`` of train.py:276 drives ``lr``), same update rule, same state names (``exp_avg``,
``exp_avg_sq``, ``step``).

One launch per parameter group and step (``vitk_adamw_multi``): every fp32 GPU parameter of the group, 1-D ones
included, is one record of a descriptor table in device memory, and the pass that updates it also rewrites the
bf16 copies of the new value the GEMMs read -- the [out, in] copy (forward, dW) and the transposed copy (dX) --
so no cast / transpose kernel runs in steady state (functional.shadow).  The table is rebuilt only when a
pointer in it changes (gradients that live in the data-parallel reducer's slabs, or that the caching allocator
hands back at the same address every step, keep it valid).  The step-dependent scalars travel to the device as
KERNEL ARGUMENTS of a one-wave kernel (``vitk_set_f32``): no host buffer has to outlive the call, so the host
may run any number of steps / graph replays ahead of the GPU.
Anything else (non-fp32, CPU, misaligned) takes torch's fused multi-tensor kernel; there is no CPU path.
"""
import importlib

import numpy as np
import torch

from . import functional as VF
from ._lib import lib, check
from .ops import _stream


_torch_adamw = importlib.import_module('torch.optim.adamw').adamw     # functional multi-tensor AdamW
_TILE = 64


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, *,
                 maximize=False, capturable=False):
        if amsgrad or maximize:
            raise ValueError('vit_amd.optim.AdamW: amsgrad / maximize are not implemented (the reference uses neither)')
        if lr < 0 or eps < 0 or weight_decay < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError('vit_amd.optim.AdamW: invalid hyper-parameter')
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        # capturable: step() launches nothing that depends on the step count (the scalars are written by update_dyn(),
        # which must run before each step / graph replay), so that it can be captured in a HIP graph (vit_amd/graph.py)
        self.capturable = bool(capturable)
        self._t = 0
        self._count = {}          # id(param) -> steps taken (the authoritative counter; state['step'] is materialised from it)
        self._plans = {}

    # ------------------------------------------------------------------ step-dependent scalars
    @staticmethod
    def _scalars(group, t):
        b1, b2 = group['betas']
        vals = np.array([1.0 - group['lr'] * group['weight_decay'], group['lr'] / (1.0 - b1 ** t),
                         1.0 / (1.0 - b2 ** t) ** 0.5], dtype=np.float64).astype(np.float32)    # rounded once, as torch does
        return float(vals[0]), float(vals[1]), float(vals[2])

    def _dyn(self, group, slot):
        bufs = group.setdefault('dyn', {})
        buf = bufs.get(slot)
        if buf is None:
            buf = bufs[slot] = torch.zeros(4, device=group['params'][0].device, dtype=torch.float32)
        return buf

    def update_dyn(self):
        """capturable mode: advance the step count and send each group's {1 - lr wd, lr / (1 - b1^t), 1 / sqrt(1 - b2^t)}
        to its device scalars -- stream-ordered, by value (legal right before a graph replay, not inside a capture)."""
        self._t += 1
        L = lib()
        stream = _stream()
        for group in self.param_groups:
            counts = {self._count.get(id(p), 0) for p in group['params'] if self.state.get(p)}
            if len(counts) > 1:
                raise RuntimeError('vit_amd.optim.AdamW(capturable=True): the parameters of a group must share a step count')
            for p in group['params']:
                self._count[id(p)] = self._t
                st = self.state.get(p)
                if st and st.get('native'):
                    st['step'] += 1               # a CPU scalar: state[p]['step'] stays readable (checkpoints, tests)
            if not group['params'] or not group['params'][0].is_cuda:
                continue
            v = self._scalars(group, self._t)
            check(L.vitk_set_f32(self._dyn(group, 0).data_ptr(), 3, v[0], v[1], v[2], 0.0, stream), 'vitk_set_f32')

    # ------------------------------------------------------------------ state
    @staticmethod
    def _native(p):
        """Parameters the HIP kernel updates: fp32, on the GPU, contiguous (a 1-D parameter is one row)."""
        return p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.numel() > 0 and p.data_ptr() % 16 == 0

    def _sync_step_tensors(self):
        for group in self.param_groups:
            for p in group['params']:
                st = self.state.get(p)
                if st and st.get('native'):
                    st['step'] = torch.tensor(float(self._count.get(id(p), 0)), dtype=torch.float32)

    def state_dict(self):
        """torch.optim.AdamW's layout (state: step / exp_avg / exp_avg_sq; groups: lr, betas, eps, weight_decay, ...):
        a checkpoint written here loads into torch.optim.AdamW and the other way round (train.py:589)."""
        self._sync_step_tensors()
        sd = super().state_dict()
        sd['state'] = {k: {n: v for n, v in st.items() if n != 'native'} for k, st in sd['state'].items()}
        sd['param_groups'] = [{n: v for n, v in g.items() if n != 'dyn'} for g in sd['param_groups']]
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._plans.clear()                          # the moment buffers were replaced
        self._count.clear()
        t = 0
        for group in self.param_groups:
            group.pop('dyn', None)                   # device scalars are rebuilt on demand
            group.pop('dyn_host', None)
            for p in group['params']:
                st = self.state.get(p)
                if not st:
                    continue
                st['native'] = self._native(p)
                for n in ('exp_avg', 'exp_avg_sq'):
                    st[n] = st[n].to(device=p.device, dtype=torch.float32).contiguous()
                step = torch.as_tensor(st['step'], dtype=torch.float32)
                st['step'] = step.detach().cpu().reshape(()) if st['native'] else step.to(p.device).reshape(())
                self._count[id(p)] = int(st['step'].item())
                t = max(t, self._count[id(p)])
        self._t = t

    # ------------------------------------------------------------------ descriptor tables
    @staticmethod
    def _copies(p):
        """(shadow ptr, ld, transposed shadow ptr, ld) of the live bf16 copies of p that the update pass rewrites."""
        rows = p.shape[0] if p.dim() >= 2 else 1
        cols = p.numel() // rows
        ps = lds = pt = ldt = 0
        for key, buf, off, transposed in VF.shadow_targets(p):
            ok = buf.dtype == torch.bfloat16 and buf.is_contiguous() and buf.numel() > 0 and p.dim() >= 2 and cols % 4 == 0
            if ok and not transposed and not ps and buf.numel() // buf.shape[0] == cols and off + rows <= buf.shape[0] \
                    and buf.data_ptr() % 8 == 0:
                lds = cols
                ps = buf.data_ptr() + off * lds * 2
            elif ok and transposed and not pt and buf.dim() == 2 and buf.shape[0] == cols and off % 8 == 0 \
                    and buf.data_ptr() % 16 == 0 and off + rows <= buf.shape[1]:
                ldt = buf.stride(0)
                pt = buf.data_ptr() + off * 2
            else:
                VF.shadow_drop(key)      # cannot be rewritten here: rebuilt from p at its next use
        return ps, lds, pt, ldt

    def _table(self, slot, items):
        """Device descriptor table of ``items`` = [(p, g, state)], cached while no pointer in it changes."""
        gen = VF.shadow_generation()
        key = (gen,) + tuple((id(p), p.data_ptr(), g.data_ptr()) for p, g, _ in items)
        plan = self._plans.get(slot)
        if plan is not None and plan[0] == key:
            return plan[1:]
        rec = np.zeros((len(items) + 1, 9), dtype=np.int64)
        tile0 = 0
        for i, (p, g, st) in enumerate(items):
            rows = p.shape[0] if p.dim() >= 2 else 1
            cols = p.numel() // rows
            ps, lds, pt, ldt = self._copies(p)
            tx, ty = -(-cols // _TILE), -(-rows // _TILE)
            rec[i] = (p.data_ptr(), g.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), ps, pt,
                      lds | (ldt << 32), rows | (cols << 32), tile0 | (tx << 32))
            tile0 += tx * ty
        rec[len(items), 8] = tile0
        if VF.shadow_generation() != gen:            # _copies dropped a copy: the key must carry the new generation
            key = (VF.shadow_generation(),) + key[1:]
        # through kernel arguments (vitk_upload_args): stream-ordered, legal inside a stream capture -- where the
        # gradients of the captured backward get their addresses -- and `rec` is free again when the call returns
        table = torch.empty(rec.shape, device=items[0][0].device, dtype=torch.int64)
        check(lib().vitk_upload_args(table.data_ptr(), rec.ctypes.data, rec.nbytes, _stream()), 'vitk_upload_args')
        self._plans[slot] = (key, table, len(items), tile0)
        return table, len(items), tile0

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = lib()
        if L.vitk_adamw_desc_bytes() != 72:
            raise RuntimeError('vit_amd.optim.AdamW: descriptor layout differs from libvitk.so')
        stream = None
        for gi, group in enumerate(self.param_groups):
            lr, (b1, b2), eps, wd = group['lr'], group['betas'], group['eps'], group['weight_decay']
            lr, b1, b2, eps, wd = float(lr), float(b1), float(b2), float(eps), float(wd)
            small, small_g, small_m, small_v, small_t = [], [], [], [], []
            by_t = {}
            for p in group['params']:
                g = p.grad
                if g is None:
                    continue
                st = self.state[p]
                if not st:
                    if g.is_sparse:
                        raise RuntimeError('vit_amd.optim.AdamW does not support sparse gradients')
                    st['native'] = self._native(p)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['step'] = torch.zeros((), dtype=torch.float32) if st['native'] else \
                        torch.zeros((), dtype=torch.float32, device=p.device)
                if not st['native']:
                    small.append(p)
                    small_g.append(g)
                    small_m.append(st['exp_avg'])
                    small_v.append(st['exp_avg_sq'])
                    small_t.append(st['step'])
                    continue
                if g.dtype != torch.float32 or not g.is_contiguous() or g.data_ptr() % 16 or g.shape != p.shape:
                    g = g.float().contiguous().clone().view_as(p)
                if self.capturable:
                    t = self._count.get(id(p), 0)                # advanced by update_dyn()
                else:
                    t = self._count[id(p)] = self._count.get(id(p), 0) + 1
                    self._t = max(self._t, t)
                    st['step'] += 1
                by_t.setdefault(t, []).append((p, g, st))
            if small and self.capturable:
                raise RuntimeError('vit_amd.optim.AdamW(capturable=True): every parameter must be contiguous fp32 on the '
                                   'GPU; got %s' % [tuple(q.shape) for q in small])
            if self.capturable and len(by_t) > 1:
                raise RuntimeError('vit_amd.optim.AdamW(capturable=True): the parameters of a group must share a step count')
            for slot, (t, items) in enumerate(sorted(by_t.items())):
                if stream is None:
                    stream = _stream()
                if self.capturable:
                    if 'dyn' not in group or t < 1:
                        raise RuntimeError('vit_amd.optim.AdamW(capturable=True): call update_dyn() before step()')
                    dyn = self._dyn(group, 0)
                else:
                    dyn = self._dyn(group, slot)
                    v = self._scalars(group, t)
                    check(L.vitk_set_f32(dyn.data_ptr(), 3, v[0], v[1], v[2], 0.0, stream), 'vitk_set_f32')
                table, n, tiles = self._table((gi, slot), items)
                check(L.vitk_adamw_multi(n, table.data_ptr(), tiles, b1, b2, eps, dyn.data_ptr(), stream), 'vitk_adamw_multi')
            if small:
                if small[0].is_cuda:
                    _torch_adamw(small, small_g, small_m, small_v, [], small_t, amsgrad=False, beta1=b1,
                                 beta2=b2, lr=lr, weight_decay=wd, eps=eps, maximize=False, fused=True,
                                 grad_scale=None, found_inf=None)
                else:
                    raise RuntimeError('vit_amd.optim.AdamW: parameters must live on the GPU (no CPU path)')
        return loss
