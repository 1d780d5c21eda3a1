"""Flat-buffer parameter storage + fused Adam / grad-clip on the HIP kernels
(reference: torch.optim.Adam + clip_grad_norm_, new_autoencoder.py:549-556, 648-655)."""
from __future__ import annotations

import torch

from . import ops

ALIGN = 64  # elements (256 B): every parameter view starts on a cache line


class FlatParams:
    """Re-homes parameters (and their .grad) as views of two contiguous fp32 buffers so that the
    optimiser and the gradient all-reduce each see ONE tensor.  Pure torch: works on CPU too
    (used by the gloo tests of the data-parallel engine)."""

    def __init__(self, params):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError('FlatParams: empty parameter list')
        dev, dt = self.params[0].device, self.params[0].dtype
        self.offsets, total = [], 0
        for p in self.params:
            if p.device != dev or p.dtype != dt:
                raise ValueError('FlatParams: all parameters must share device and dtype')
            self.offsets.append(total)
            total += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total = total
        self.data = torch.zeros(total, dtype=dt, device=dev)
        self.grad = torch.zeros(total, dtype=dt, device=dev)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                n = p.numel()
                self.data[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.data[off:off + n].view(p.shape)
                p.grad = self.grad[off:off + n].view(p.shape)

    def zero_grad(self):
        self.grad.zero_()
        for p, off in zip(self.params, self.offsets):  # re-attach views dropped by set_to_none
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + off * self.grad.element_size():
                p.grad = self.grad[off:off + p.numel()].view(p.shape)

    def span(self, i):
        return self.offsets[i], self.offsets[i] + self.params[i].numel()


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam / AdamW semantics (no amsgrad) in one HIP launch over the flat
    buffer; clip_grad_norm() folds the clip coefficient into that launch (device side, no host
    synchronisation)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        """weight_decay > 0 gives torch.optim.AdamW's DECOUPLED decay (the distillation step's optimiser,
        weight_distill_train.py:317); torch.optim.Adam's L2-in-gradient decay is not offered."""
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise ValueError('FusedAdam supports a single parameter group')
        self.flat = FlatParams(self.param_groups[0]['params'])
        if not self.flat.data.is_cuda:
            from .lib import EovaeError
            raise EovaeError('FusedAdam: parameters must live on an AMD GPU (no CPU fallback)')
        self.exp_avg = torch.zeros_like(self.flat.data)
        self.exp_avg_sq = torch.zeros_like(self.flat.data)
        self.step_count = 0
        self._norm = torch.zeros(2, dtype=torch.float32, device=self.flat.data.device)
        self._clip_pending = False
        self.grad_scale = 1.0
        for i, p in enumerate(self.flat.params):  # torch.optim.Adam-shaped state for checkpoints
            a, b = self.flat.span(i)
            self.state[p] = {'step': torch.tensor(0.0), 'exp_avg': self.exp_avg[a:b].view(p.shape),
                             'exp_avg_sq': self.exp_avg_sq[a:b].view(p.shape)}

    def zero_grad(self, set_to_none: bool = True):
        self.flat.zero_grad()  # views are kept alive: autograd accumulates in place

    def clip_grad_norm(self, max_norm: float) -> torch.Tensor:
        """Global L2 norm of the (already averaged) gradient; returns a device scalar."""
        ops.grad_norm(self.flat.grad, max_norm, self._norm)
        self._clip_pending = bool(max_norm)
        return self._norm[0]

    @torch.no_grad()
    def step(self, closure=None):
        group = self.param_groups[0]
        self.step_count += 1
        b1, b2 = group['betas']
        ops.adam_step(self.flat.data, self.flat.grad, self.exp_avg, self.exp_avg_sq, float(group['lr']),
                      b1, b2, group['eps'], self.step_count,
                      clip=self._norm if self._clip_pending else None, grad_scale=self.grad_scale,
                      weight_decay=float(group.get('weight_decay', 0.0)))
        self._clip_pending = False
        for st in self.state.values():
            st['step'] += 1
        return None
