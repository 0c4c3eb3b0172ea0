"""Data-parallel gradient exchange (new capability; the reference is single-process).

Semantics = PyTorch DDP: one process per GPU, full replica each, local BatchNorm
statistics and local BCE+Dice loss on the rank's own images, gradients AVERAGED across ranks before
``optimizer.step()``, parameters and BN buffers broadcast from rank 0 at start.

Mechanics: the engine's backward is cut into 10 stages; stage i completes gradient bucket i, a
contiguous slice of the flat fp32 gradient buffer (head+dec2..4 | dec1 | dec0 | layer4.2 | layer4.1 |
layer4.0 | layer3.3-5 | layer3.0-2 | layer2 | layer1+stem).  After each stage the bucket is handed to
``torch.distributed.all_reduce(async_op=True)`` — backend "nccl" is RCCL over xGMI — which runs on
RCCL's own stream behind an event, overlapping the remaining backward kernels.  ``finish()`` waits for
all handles before the optimizer runs.  The 1/world factor is folded into the AdamW kernel
(``FusedAdamW.grad_inv_scale``) or applied by ``scale_grads=True``.

The class only needs a flat gradient tensor and bucket ranges, so the same code is exercised on CPU
with the gloo backend (tests/test_parallel_cpu.py)."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


class GradientReducer:
    def __init__(self, flat_grads_getter, world_size: Optional[int] = None, process_group=None,
                 scale_grads: bool = False, force: bool = False):
        self._get = flat_grads_getter
        self.pg = process_group
        self.world = world_size if world_size is not None else (dist.get_world_size(process_group) if dist.is_initialized() else 1)
        self.scale_grads = scale_grads
        self._handles: List = []
        self._ranges: List[Tuple[int, int]] = []
        self.enabled = self.world > 1 or force        # force: run the collectives even with one rank (testing)
        self.timing = False                           # bench.py: bracket the wait for the collectives with two events
        self._tail_events: List = []

    @property
    def inv_world(self) -> float:
        return 1.0 / self.world

    def bucket_ready(self, index: int, rng: Tuple[int, int]):
        if not self.enabled:
            return
        b0, b1 = rng
        if b1 <= b0:
            return
        view = self._get()[b0:b1]
        self._handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        self._ranges.append(rng)

    def finish(self):
        timed = self.timing and self.enabled and self._handles and torch.cuda.is_available()
        if timed:      # exposed tail = what the compute stream still has to wait for after its last backward kernel
            e1, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e1.record()
        for h in self._handles:
            h.wait()
        if timed:
            e2.record()
            self._tail_events.append((e1, e2))
        if self.enabled and self.scale_grads:
            g = self._get()
            for b0, b1 in self._ranges:
                g[b0:b1].mul_(self.inv_world)
        self._handles.clear()
        self._ranges.clear()


    def exposed_tail_ms(self) -> List[float]:
        """Per step: time the compute stream waited for outstanding all-reduces after the last backward kernel (needs
        ``timing = True``; synchronises)."""
        out = []
        for e1, e2 in self._tail_events:
            e2.synchronize()
            out.append(e1.elapsed_time(e2))
        self._tail_events.clear()
        return out


def all_reduce_scalars(values: torch.Tensor, process_group=None, average: bool = True) -> torch.Tensor:
    """loss / metric scalars for logging under data parallelism — the reference logs the sample-weighted mean loss
    of its one process (train.py:452-459); with one process per GPU every rank holds its shard's value.  One small all-reduce
    of a device tensor (no host sync here); returns the mean (or the sum) over ranks.  No-op without a process group."""
    if not dist.is_initialized():
        return values
    out = values.detach().clone()
    dist.all_reduce(out, op=dist.ReduceOp.SUM, group=process_group)
    if average:
        out /= dist.get_world_size(process_group)
    return out


def broadcast_model(model, src: int = 0, process_group=None):
    """Rank-0 parameters and BatchNorm buffers to every rank (three flat tensors, three collectives)."""
    if not dist.is_initialized():
        return
    for k in ("params", "bufs", "nbt"):
        dist.broadcast(model._flat[k], src=src, group=process_group)
    model.mark_weights_dirty()


def make_data_parallel(model, optimizer=None, process_group=None, force: bool = False) -> GradientReducer:
    """Attach a GradientReducer to ``model``; with a FusedAdamW the averaging is folded into its kernel."""
    fold = optimizer is not None and hasattr(optimizer, "grad_inv_scale")
    red = GradientReducer(lambda: model.flat_grads, process_group=process_group, scale_grads=not fold, force=force)
    if fold:
        optimizer.grad_inv_scale = red.inv_world
    model._reducer = red
    broadcast_model(model, 0, process_group)
    return red
