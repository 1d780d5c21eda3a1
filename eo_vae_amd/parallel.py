"""Data parallelism: one process per GPU, gradient all-reduce over RCCL/xGMI.

The reference only ever reaches this through Lightning's `strategy: ddp` (configs/eo-vae.yaml:80,
SURVEY.md 2b C1-C6).  Here the engine is explicit and small:
  * parameters and gradients live in ONE flat buffer each (optim.FlatParams);
  * the gradient buffer is cut into buckets from the END (decoder gradients are produced
    first); a post-accumulate hook counts a bucket's parameters down and launches its
    all-reduce (async, RCCL stream) the moment the last one is ready, so communication
    overlaps the rest of the backward pass;
  * averaging uses ReduceOp.AVG on RCCL (SUM + scale on gloo for the CPU tests);
  * gradient clipping runs after finish() on the averaged gradient, so every rank computes
    the same norm with no further collective (same order as manual_backward -> clip_grad_norm_,
    new_autoencoder.py:648-652)."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_distributed(backend: str | None = None, force: bool = False):
    """Initialise torch.distributed from the torchrun environment; returns (rank, world, local).
    `force` also initialises a world of 1 (the single-GPU RCCL rehearsal in tests/test_gpu_dp.py)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:  # EOVAE_DIST_BACKEND=gloo: single-GPU rehearsal of the N>1 path
            backend = os.environ.get('EOVAE_DIST_BACKEND') or \
                ('nccl' if torch.cuda.is_available() else 'gloo')  # 'nccl' IS RCCL on ROCm
        if os.environ.get('EOVAE_ALL_RANKS_ONE_GPU') == '1':
            local = 0
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


# HIP streams other than the main one on which parts of the backward pass run (e.g. the decoder's
# kernel generator, dynamic_conv.DynamicConv_decoder.prefetch).  A bucket may hold gradients produced on
# several of them, so its all-reduce is ordered after all of them, not only after the stream whose hook
# happened to complete the bucket.
SIDE_STREAMS: list = []


class BucketedGradReducer:
    def __init__(self, flat, group=None, bucket_mb: float = 48.0, force: bool = False):
        """`force` keeps the hooks and collectives even in a world of 1 (RCCL rehearsal on one GPU)."""
        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.use_avg = dist.is_initialized() and dist.get_backend(group) == 'nccl'
        cap = int(bucket_mb * (1 << 20) / flat.grad.element_size())
        # buckets = contiguous [start, end) ranges of the flat gradient, built back to front
        self.buckets, self.bucket_of = [], {}
        end, members = flat.total, []
        for i in reversed(range(len(flat.params))):
            members.append(i)
            start = flat.offsets[i]
            if end - start >= cap or i == 0:
                self.buckets.append((start, end, list(members)))
                end, members = start, []
        for b, (_, _, idxs) in enumerate(self.buckets):
            for i in idxs:
                self.bucket_of[id(flat.params[i])] = b
        self.pending = [len(m) for _, _, m in self.buckets]
        self.launched = [False] * len(self.buckets)
        self.works = []
        self.handles = []
        self.main_stream = torch.cuda.current_stream() if flat.grad.is_cuda else None
        # Fault agreement: a rank-local fault (a GroupNorm-backward time-out: its gradients are NaN) must stop
        # EVERY rank in the same step -- a rank that raises alone leaves the others in the next all-reduce.  One MAX
        # all-reduce of a flag rides with every step's buckets; its result is copied to pinned host memory without
        # blocking and read `HEALTH_LAG` steps later behind that step's event, so every rank examines the same reduced
        # value of the same step (no host synchronisation with the running step).
        self.local_fault = self._gn_timeout if flat.grad.is_cuda else (lambda: False)
        self.health = torch.zeros(1, dtype=torch.float32, device=flat.grad.device)
        self.health_slots, self.step_no = [], 0
        if self.active:
            for _ in range(self.HEALTH_LAG + 1):
                host = torch.zeros(1, dtype=torch.float32)
                self.health_slots.append([host.pin_memory() if flat.grad.is_cuda else host, None, -1])
            import weakref
            from . import ops
            ops.GN_TIMEOUT_AGREED = self.fault_agreed
            self._hook = weakref.WeakMethod(self.fault_agreed)
            ops.GN_TIMEOUT_HOOKS[:] = [r for r in ops.GN_TIMEOUT_HOOKS if r() is not None] + [self._hook]
            for p in flat.params:
                if p.requires_grad:
                    self.handles.append(p.register_post_accumulate_grad_hook(self._on_grad))

    HEALTH_LAG = 2

    @staticmethod
    def _gn_timeout():
        from . import lib as L
        return L.load().eovae_gn_bwd_timeouts(0) != 0

    def _reduce_health(self):
        self.health.fill_(1.0 if self.local_fault() else 0.0)
        dist.all_reduce(self.health, op=dist.ReduceOp.MAX, group=self.group)
        slot = self.health_slots[self.step_no % len(self.health_slots)]
        slot[0].copy_(self.health, non_blocking=True)
        slot[1] = torch.cuda.current_stream().record_event() if self.health.is_cuda else None
        slot[2] = self.step_no
        self.step_no += 1

    def fault_agreed(self) -> bool:
        """True on every rank in the same step: some rank reported a fault HEALTH_LAG steps ago."""
        want = self.step_no - self.HEALTH_LAG
        if not self.active or want < 0:
            return False
        slot = self.health_slots[want % len(self.health_slots)]
        if slot[2] != want:
            return False
        if slot[1] is not None:
            slot[1].synchronize()
        return bool(slot[0].item() > 0)

    # ---- start-of-training synchronisation (DDP constructor broadcast, C1) ----
    def broadcast_parameters(self, buffers=()):
        if not self.active:
            return
        dist.broadcast(self.flat.data, src=0, group=self.group)
        for b in buffers:
            dist.broadcast(b, src=0, group=self.group)

    def _on_grad(self, p):
        b = self.bucket_of[id(p)]
        self.pending[b] -= 1
        if self.pending[b] == 0:
            self._launch(b)

    def _launch(self, b):
        start, end, _ = self.buckets[b]
        if self.flat.grad.is_cuda:
            cur = torch.cuda.current_stream()
            for st in [self.main_stream] + SIDE_STREAMS:
                if st is not None and st != cur:
                    cur.wait_stream(st)
        if getattr(self.flat, 'detached', False):   # this bucket's gradients -> flat buffer, one launch
            idxs = self.buckets[b][2]
            self.flat.gather(min(idxs), max(idxs) + 1)
        op = dist.ReduceOp.AVG if self.use_avg else dist.ReduceOp.SUM
        self.works.append(dist.all_reduce(self.flat.grad[start:end], op=op, group=self.group,
                                          async_op=True))
        self.launched[b] = True

    def finish(self):
        """Call after backward: launches buckets whose parameters received no gradient, waits
        for every collective, leaves the AVERAGED gradient in flat.grad."""
        if not self.active:
            return
        for b in range(len(self.buckets)):
            if not self.launched[b]:
                self._launch(b)
        for w in self.works:
            w.wait()
        self._reduce_health()
        if not self.use_avg:
            self.flat.grad.mul_(1.0 / self.world)
        if getattr(self.flat, 'detached', False):
            self.flat.gathered = True       # every bucket was gathered on its way to the all-reduce
        self.works = []
        self.pending = [len(m) for _, _, m in self.buckets]
        self.launched = [False] * len(self.buckets)

    def remove(self):
        for h in self.handles:
            h.remove()
        self.handles = []
        from . import ops
        if ops.GN_TIMEOUT_AGREED == self.fault_agreed:
            ops.GN_TIMEOUT_AGREED = None
        hook = getattr(self, '_hook', None)
        ops.GN_TIMEOUT_HOOKS[:] = [r for r in ops.GN_TIMEOUT_HOOKS if r is not hook and r() is not None]
