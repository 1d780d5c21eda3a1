"""Flat-buffer parameter storage + fused Adam / grad-clip on the HIP kernels
(reference: torch.optim.Adam + clip_grad_norm_, new_autoencoder.py:549-556, 648-655)."""
from __future__ import annotations

import torch

from . import ops

ALIGN = 64  # elements (256 B): every parameter view starts on a cache line


class FlatParams:
    """Re-homes parameters as views of ONE contiguous fp32 buffer and keeps ONE flat gradient buffer, so
    that the optimiser and the gradient all-reduce each see a single tensor.

    Two gradient modes:
      * views (default; pure torch, also on CPU -- the gloo tests of the data-parallel engine): every
        p.grad is a view of the flat gradient and autograd accumulates into it in place (one add launch
        per parameter and step, plus the zero fill);
      * detached_grads=True (what FusedAdam uses on the GPU): p.grad is left to autograd (None after
        zero_grad, so the engine simply keeps the tensor a backward op returned) and gather() copies
        all of them -- or one bucket's worth -- into the flat buffer with ONE HIP launch."""

    CHUNK = 16384  # elements per gather-table row
    NBUF = 4       # pointer-table slots (>= buckets in flight + 1)

    def __init__(self, params, detached_grads: bool = False):
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
        self.detached = bool(detached_grads)
        if self.detached and not (dev.type == 'cuda' and dt == torch.float32):
            raise ValueError('FlatParams: detached_grads needs fp32 parameters on an AMD GPU')
        self.data = torch.zeros(total, dtype=dt, device=dev)
        self.grad = torch.zeros(total, dtype=dt, device=dev)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                n = p.numel()
                self.data[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.data[off:off + n].view(p.shape)
                p.grad = None if self.detached else self.grad[off:off + n].view(p.shape)
        if self.detached:
            self._build_tables(dev)

    # ---- detached mode: gather tables -------------------------------------------------------
    def _build_tables(self, dev):
        import numpy as np
        rows_param, rows_src, rows_dst, rows_cnt, self.row_span = [], [], [], [], []
        for i, (p, off) in enumerate(zip(self.params, self.offsets)):
            n, first = p.numel(), len(rows_param)
            for c0 in range(0, n, self.CHUNK):
                rows_param.append(i)
                rows_src.append(c0 * 4)                       # byte offset inside the gradient tensor
                rows_dst.append(off + c0)
                rows_cnt.append(min(self.CHUNK, n - c0))
            self.row_span.append((first, len(rows_param)))
        self._row_param = np.asarray(rows_param, dtype=np.int64)
        self._row_src = np.asarray(rows_src, dtype=np.int64)
        nrows = len(rows_param)
        # a small ring of pinned staging tables: a gather may be built while earlier uploads are still in
        # flight, and re-using a slot only waits for the gather issued NBUF calls ago
        self._host = [torch.empty((nrows, 3), dtype=torch.int64).pin_memory() for _ in range(self.NBUF)]
        for h in self._host:
            h[:, 1] = torch.from_numpy(np.asarray(rows_dst, dtype=np.int64))
            h[:, 2] = torch.from_numpy(np.asarray(rows_cnt, dtype=np.int64))
        self._dev = [torch.empty((nrows, 3), dtype=torch.int64, device=dev) for _ in range(self.NBUF)]
        self._events = [None] * self.NBUF
        self._flip = 0
        self._keep = []          # contiguous copies that must outlive the gather launch
        self.gathered = False

    def gather(self, first: int = 0, last: int | None = None) -> None:
        """Copy the gradients of parameters [first, last) into the flat buffer (zeros where a parameter
        received none).  One launch; the pointer table goes up by an async pinned copy."""
        if not self.detached:
            return
        import numpy as np
        last = len(self.params) if last is None else last
        r0, r1 = self.row_span[first][0], self.row_span[last - 1][1]
        k = self._flip
        self._flip = (k + 1) % self.NBUF
        if self._events[k] is not None:
            self._events[k].synchronize()
        base = np.zeros(len(self.params), dtype=np.int64)
        for i in range(first, last):
            g = self.params[i].grad
            if g is None:
                continue
            if g.dtype != torch.float32 or not g.is_contiguous():
                g = g.float().contiguous()
                self._keep.append(g)
            base[i] = g.data_ptr()
        rp = self._row_param[r0:r1]
        src = base[rp]
        src = np.where(src != 0, src + self._row_src[r0:r1], 0)
        host = self._host[k]
        host[r0:r1, 0] = torch.from_numpy(src)
        dev = self._dev[k]
        dev[r0:r1].copy_(host[r0:r1], non_blocking=True)
        from . import lib as L
        L.check(L.load().eovae_multi_gather(L.ptr(dev[r0:r1]), r1 - r0, L.ptr(self.grad), L.stream()),
                'eovae_multi_gather')
        self._events[k] = torch.cuda.Event()
        self._events[k].record()

    def zero_grad(self):
        if self.detached:
            for p in self.params:
                p.grad = None
            self._keep = []
            self.gathered = False
            return
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
        if not self.param_groups[0]['params'][0].is_cuda:
            from .lib import EovaeError
            raise EovaeError('FusedAdam: parameters must live on an AMD GPU (no CPU fallback)')
        import os
        self.flat = FlatParams(self.param_groups[0]['params'],
                               detached_grads=os.environ.get('EOVAE_DETACHED_GRADS', '1') != '0')
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
        self._ensure_gathered()
        ops.grad_norm(self.flat.grad, max_norm, self._norm)
        self._clip_pending = bool(max_norm)
        return self._norm[0]

    def _ensure_gathered(self):
        """Single-process path: the gradients autograd left on the parameters go into the flat buffer
        here; under data parallelism the reducer has already gathered them bucket by bucket."""
        if self.flat.detached and not self.flat.gathered:
            self.flat.gather()
            self.flat.gathered = True

    @torch.no_grad()
    def step(self, closure=None):
        group = self.param_groups[0]
        self._ensure_gathered()
        self.step_count += 1
        b1, b2 = group['betas']
        ops.adam_step(self.flat.data, self.flat.grad, self.exp_avg, self.exp_avg_sq, float(group['lr']),
                      b1, b2, group['eps'], self.step_count,
                      clip=self._norm if self._clip_pending else None, grad_scale=self.grad_scale,
                      weight_decay=float(group.get('weight_decay', 0.0)))
        self._clip_pending = False
        self.flat.gathered = False      # the next step gathers whatever gradients the parameters hold then
        for st in self.state.values():
            st['step'] += 1
        return None
