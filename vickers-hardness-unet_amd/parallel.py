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

WHEN the collectives are issued is a policy (``GradientReducer.policy``), because on this chip it decides what they cost: the
K >= 128 tile kernels and the weight-gradient kernels launch exactly one workgroup per compute unit (<= 256 workgroups, 111-160 KiB
of LDS each), and a communication kernel that holds even 8 CUs while they run pushes those launches into a second round
(tests/diag/cu_hold.py, profiles/r03/cu_hold*.log: +35 % step time with 8-32 CUs held for the whole step, +13-15 % with co-resident
streaming workgroups).
  * ``"eager"``    — every bucket is reduced as soon as its stage has finished (maximal overlap, r01/r02 behaviour);
  * ``"deferred"`` — (default) buckets 0 .. defer_until-1 are contiguous in the flat buffer (the bucket table runs from its tail to its
                     head) and are reduced by ONE collective issued when stage defer_until-1 has finished, i.e. 99 % of the bytes
                     travel under the last stage (layer 1 + stem: HBM-bound kernels with thousands of workgroups, which lose a few
                     per cent to held CUs, not a whole round); the last, tiny bucket follows at the end;
  * ``"tail"``     — one collective over the whole buffer after the last stage (no overlap: the robust lower bound).
``reserved_cus`` > 0 additionally caps the grids of the one-workgroup-per-CU weight-gradient kernels at (CUs - reserved_cus) while
collectives are in flight (vk_set_reserved_cus).

The class only needs a flat gradient tensor and bucket ranges, so the same code is exercised on CPU
with the gloo backend (tests/test_parallel_cpu.py)."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


class GradientReducer:
    POLICIES = ("eager", "deferred", "tail")

    def __init__(self, flat_grads_getter, world_size: Optional[int] = None, process_group=None,
                 scale_grads: bool = False, force: bool = False, policy: str = "deferred", defer_until: int = 9,
                 reserved_cus: int = 0):
        self._get = flat_grads_getter
        self.pg = process_group
        self.world = world_size if world_size is not None else (dist.get_world_size(process_group) if dist.is_initialized() else 1)
        self.scale_grads = scale_grads
        self._handles: List = []
        self._ranges: List[Tuple[int, int]] = []
        self.enabled = self.world > 1 or force        # force: run the collectives even with one rank (testing)
        self.timing = False                           # bench.py: bracket the wait for the collectives with two events
        self._tail_events: List = []
        if policy not in self.POLICIES:
            raise ValueError("policy must be one of %s" % (self.POLICIES,))
        self.policy = policy
        self.defer_until = int(defer_until)           # "deferred": stages [0, defer_until) travel as one collective
        self.reserved_cus = int(reserved_cus)
        self._held: List[Tuple[int, int]] = []        # finished, not yet issued buckets
        self.in_flight = False                        # a collective has been issued and not yet waited for

    @property
    def inv_world(self) -> float:
        return 1.0 / self.world

    def _issue(self, b0: int, b1: int):
        view = self._get()[b0:b1]
        self._handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        self._ranges.append((b0, b1))
        self.in_flight = True

    def _flush_held(self):
        """One collective per maximal contiguous run of the held buckets (the engine's table is one run)."""
        runs: List[List[int]] = []
        for b0, b1 in sorted(self._held):
            if runs and runs[-1][1] == b0:
                runs[-1][1] = b1
            else:
                runs.append([b0, b1])
        for b0, b1 in runs:
            self._issue(b0, b1)
        self._held.clear()

    def stage_groups(self, nbuckets: int) -> List[Tuple[int, int]]:
        """Backward stages whose gradients are first needed TOGETHER, as [begin, end) ranges: the model runs each range as one
        ``vk_unet_backward`` call, i.e. one weight-gradient batch (include/vk_unet.h, vk_conv_wgrad_batch).  "eager" needs every
        stage at once, "deferred" nothing before stage ``defer_until - 1`` is done, "tail" nothing before the end."""
        if not self.enabled or self.policy == "tail":
            return [(0, nbuckets)]
        if self.policy == "deferred":
            first = max(1, min(self.defer_until, nbuckets))
            return [(0, first)] + [(s, s + 1) for s in range(first, nbuckets)]
        return [(s, s + 1) for s in range(nbuckets)]

    def bucket_ready(self, index: int, rng: Tuple[int, int]):
        if not self.enabled:
            return
        b0, b1 = rng
        if b1 <= b0:
            return
        if self.policy == "eager":
            self._issue(b0, b1)
            return
        self._held.append((b0, b1))
        if self.policy == "deferred" and index >= self.defer_until - 1:
            self._flush_held()                        # stage defer_until-1 done: everything so far goes out; later buckets one by one

    def finish(self):
        if self.enabled and self._held:
            self._flush_held()
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
        self.in_flight = False

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


def make_data_parallel(model, optimizer=None, process_group=None, force: bool = False, policy: Optional[str] = None,
                       reserved_cus: Optional[int] = None) -> GradientReducer:
    """Attach a GradientReducer to ``model``; with a FusedAdamW the averaging is folded into its kernel.  ``policy`` / ``reserved_cus``
    default to the environment (VK_DP_POLICY = eager | deferred | tail, VK_DP_RESERVED_CUS = n) and then to "deferred" / 0."""
    import os
    fold = optimizer is not None and hasattr(optimizer, "grad_inv_scale")
    policy = policy or os.environ.get("VK_DP_POLICY", "deferred")
    reserved_cus = int(os.environ.get("VK_DP_RESERVED_CUS", "0")) if reserved_cus is None else reserved_cus
    red = GradientReducer(lambda: model.flat_grads, process_group=process_group, scale_grads=not fold, force=force, policy=policy,
                          reserved_cus=reserved_cus)
    if fold:
        optimizer.grad_inv_scale = red.inv_world
    model._reducer = red
    broadcast_model(model, 0, process_group)
    return red
