"""AdamW over the HIP library: the optimizer either side of the hot path.

Reference: ``optimizer = optim.AdamW(model.parameters(), lr=args.lr)`` (train.py:275), stepped once per
batch (train.py:549) -- torch.optim.AdamW with torch's defaults (betas 0.9 / 0.999, eps 1e-8,
weight_decay 1e-2, no amsgrad).  Same constructor surface, same ``param_groups`` (so
``StepLR`` of train.py:276 drives ``lr``), same update rule, same state names (``exp_avg``,
``exp_avg_sq``, ``step``).

Every parameter with >= 2 dims (99.9 % of the bytes) goes through ``vitk_adamw``: one pass that also
rewrites the bf16 copies of the new value the GEMMs read -- the [out, in] copy (forward, dW) and the
transposed copy (dX) -- so no cast / transpose kernel runs in steady state (functional.shadow).
The 1-D rest (biases, LayerNorm weights) takes torch's fused multi-tensor kernel (plumbing).
There is no CPU path: the HIP library must be present.
"""
import importlib

import torch

from . import functional as VF
from ._lib import lib, check
from .ops import _stream


_torch_adamw = importlib.import_module('torch.optim.adamw').adamw     # functional multi-tensor AdamW


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, *,
                 maximize=False, capturable=False):
        if amsgrad or maximize:
            raise ValueError('vit_amd.optim.AdamW: amsgrad / maximize are not implemented (the reference uses neither)')
        if lr < 0 or eps < 0 or weight_decay < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError('vit_amd.optim.AdamW: invalid hyper-parameter')
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        # capturable: every step-dependent scalar lives in device memory (vitk_adamw_dyn) and EVERY fp32 parameter,
        # 1-D ones included, goes through the HIP kernel, so that step() can be captured in a HIP graph
        # (vit_amd/graph.py); update_dyn() must then run before each step / graph replay.
        self.capturable = bool(capturable)
        self._t = 0

    def update_dyn(self):
        """Advance the step count and write each group's {1 - lr wd, lr / (1 - b1^t), 1 / sqrt(1 - b2^t)} to its device
        scalars (a stream-ordered host-to-device copy: legal right before a graph replay, not inside a capture)."""
        self._t += 1
        for group in self.param_groups:
            b1, b2 = group['betas']
            vals = [1.0 - group['lr'] * group['weight_decay'], group['lr'] / (1.0 - b1 ** self._t),
                    1.0 / (1.0 - b2 ** self._t) ** 0.5]
            if 'dyn' not in group:
                dev = group['params'][0].device
                group['dyn'] = torch.zeros(3, device=dev, dtype=torch.float32)
                group['dyn_host'] = torch.zeros(3, dtype=torch.float32).pin_memory()
            group['dyn_host'].copy_(torch.tensor(vals, dtype=torch.float64).to(torch.float32))
            group['dyn'].copy_(group['dyn_host'], non_blocking=True)

    def _native(self, p):
        """Parameters the HIP kernel updates: fp32, on the GPU, >= 2-D (capturable: any rank, a 1-D parameter
        is one row), rows of whole 16-byte vectors."""
        if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.numel() > 0 and p.data_ptr() % 16 == 0):
            return False
        if p.dim() < 2:
            return self.capturable          # one row; a ragged length takes the kernel's element-wise form
        return (p.numel() // p.shape[0]) % 4 == 0

    def state_dict(self):
        """torch.optim.AdamW's layout (state: step / exp_avg / exp_avg_sq; groups: lr, betas, eps, weight_decay, ...):
        a checkpoint written here loads into torch.optim.AdamW and the other way round (train.py:589)."""
        sd = super().state_dict()
        sd['state'] = {k: {n: v for n, v in st.items() if n != 'native'} for k, st in sd['state'].items()}
        sd['param_groups'] = [{n: v for n, v in g.items() if n not in ('dyn', 'dyn_host')} for g in sd['param_groups']]
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self.__dict__.get('_plans', {}).clear()      # the moment buffers were replaced
        t = 0
        for group in self.param_groups:
            group.pop('dyn', None)                   # device scalars are rebuilt by update_dyn()
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
                t = max(t, int(st['step'].item()))
        self._t = t

    def _plan(self, p, st):
        """Launch arguments of one native parameter that do not change from step to step (everything
        but the gradient pointer): valid while functional's set of cached copies is unchanged."""
        rows = p.shape[0] if p.dim() >= 2 else 1
        cols = p.numel() // rows
        ps = lds = pt = ldt = 0
        for key, buf, off, transposed in VF.shadow_targets(p):
            ok = buf.dtype == torch.bfloat16 and buf.is_contiguous() and buf.numel() > 0 and p.dim() >= 2
            if ok and not transposed and not ps and buf.numel() // buf.shape[0] == cols and cols % 4 == 0 \
                    and off + rows <= buf.shape[0]:
                lds = cols
                ps = buf.data_ptr() + off * lds * 2
            elif ok and transposed and not pt and buf.dim() == 2 and buf.shape[0] == cols and off % 8 == 0 \
                    and buf.data_ptr() % 16 == 0 and off + rows <= buf.shape[1]:
                ldt = buf.stride(0)
                pt = buf.data_ptr() + off * 2
            else:
                VF.shadow_drop(key)      # cannot be rewritten here: rebuilt from p at its next use
        return (rows, cols, p.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), ps, lds, pt, ldt)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = lib()
        stream = None
        plans = self.__dict__.setdefault('_plans', {})
        if self.__dict__.get('_plan_gen') != VF.shadow_generation():
            plans.clear()
        for group in self.param_groups:
            lr, (b1, b2), eps, wd = group['lr'], group['betas'], group['eps'], group['weight_decay']
            lr, b1, b2, eps, wd = float(lr), float(b1), float(b2), float(eps), float(wd)
            small, small_g, small_m, small_v, small_t = [], [], [], [], []
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
                if not self.capturable:
                    st['step'] += 1
                if g.dtype != torch.float32 or not g.is_contiguous() or g.data_ptr() % 16 or g.shape != p.shape:
                    g = g.float().contiguous().clone().view_as(p)
                plan = plans.get(id(p))
                if plan is None or plan[2] != p.data_ptr():
                    plan = plans[id(p)] = self._plan(p, st)
                if stream is None:
                    stream = _stream()
                rows, cols, pp, pm, pv, ps, lds, pt, ldt = plan
                if self.capturable:
                    if 'dyn' not in group:
                        raise RuntimeError('vit_amd.optim.AdamW(capturable=True): call update_dyn() before step()')
                    check(L.vitk_adamw_dyn(rows, cols, pp, g.data_ptr(), pm, pv, b1, b2, eps, group['dyn'].data_ptr(),
                                           ps, lds, pt, ldt, stream), 'vitk_adamw_dyn')
                else:
                    check(L.vitk_adamw(rows, cols, pp, g.data_ptr(), pm, pv, lr, b1, b2, eps, wd, int(st['step']),
                                       ps, lds, pt, ldt, stream), 'vitk_adamw')
            if small and self.capturable:
                raise RuntimeError('vit_amd.optim.AdamW(capturable=True): every parameter must be fp32 on the GPU with a '
                                   'multiple of 4 elements per row (>= 2-D); got %s' % [tuple(q.shape) for q in small])
            if small:
                if small[0].is_cuda:
                    _torch_adamw(small, small_g, small_m, small_v, [], small_t, amsgrad=False, beta1=b1,
                                 beta2=b2, lr=lr, weight_decay=wd, eps=eps, maximize=False, fused=True,
                                 grad_scale=None, found_inf=None)
                else:
                    raise RuntimeError('vit_amd.optim.AdamW: parameters must live on the GPU (no CPU path)')
        self._plan_gen = VF.shadow_generation()
        return loss
