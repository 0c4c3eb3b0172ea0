"""``FusedAdamW`` — ``torch.optim.AdamW(model.parameters(), lr, weight_decay=1e-4)`` (reference
train.py:606) as ONE HIP launch over the model's flat fp32 parameter / gradient / moment buffers.

It is a real ``torch.optim.Optimizer`` (single param group), so ``CosineAnnealingLR`` (train.py:607),
``optimizer.param_groups[0]["lr"]`` (train.py:656), ``zero_grad(set_to_none=True)`` (train.py:428) and
``GradScaler.step(optimizer)`` (train.py:444) work unchanged.  Arithmetic follows
torch/optim/adam.py's single-tensor path (decoupled decay, bias-corrected).

GradScaler (train.py:441-445, 610-611): the optimizer declares ``_step_supports_amp_scaling``, so
``scaler.step(optimizer)`` hands it the scale and the overflow flag as DEVICE tensors
(``optimizer.grad_scale`` / ``optimizer.found_inf``); ``vk_adamw_step_amp`` reads both on the device,
folds the unscale into the update, skips the whole step on an overflow and keeps its step counter on the
device — no ``.item()``, no host round trip.  ``vk.GradScaler`` (below) additionally replaces torch's
foreach inf check over the 140 gradient views by one pass over the flat buffer."""
from __future__ import annotations

import torch

from . import _lib
from ._lib import VkError, check, lib


class FusedAdamW(torch.optim.Optimizer):
    # torch.amp.GradScaler: pass grad_scale / found_inf as attributes instead of unscaling + syncing on the host
    _step_supports_amp_scaling = True

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 amsgrad: bool = False):
        if amsgrad:
            raise NotImplementedError("amsgrad is not used by the reference")
        params = list(params)
        if not params or isinstance(params[0], dict):
            raise VkError("FusedAdamW takes model.parameters() of one vickers-hardness-unet_amd.Unet (single param group)")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._model = None
        self._m = self._v = None
        self._step_dev = None           # int32[1] on the device: optimizer steps actually taken (skipped AMP steps do not count)
        self._scratch = None            # float32[4] device scratch of vk_adamw_step_amp
        self._pending_step = 0          # step count loaded from a state dict before the device buffers exist
        self.grad_inv_scale = 1.0       # extra factor applied to gradients inside the kernel (DP averaging)

    def attach(self, model) -> "FusedAdamW":
        """Bind to the Unet whose parameters were passed (needed to reach its flat buffers)."""
        mine = {id(p) for p in self.param_groups[0]["params"]}
        theirs = {id(p) for p in model.parameters()}
        if mine != theirs:
            raise VkError("FusedAdamW must own exactly the parameters of the attached model")
        self._model = model
        return self

    def _find_model(self):
        if self._model is None:
            raise VkError("call FusedAdamW.attach(model) (or use vk.adamw_for(model, ...)) before step()")
        return self._model

    @property
    def step_count(self) -> int:
        """Optimizer steps taken so far (reads the device counter: a host sync; for logging / checkpoints only)."""
        return int(self._step_dev.item()) if self._step_dev is not None else int(self._pending_step)

    def _ensure_state(self, p: torch.Tensor):
        dev = p.device
        if self._m is None:
            self._m = torch.zeros_like(p)
            self._v = torch.zeros_like(p)
        elif self._m.device != dev:
            # moments loaded from a checkpoint mapped elsewhere, or the model moved: move them, never re-zero them silently
            self._m = self._m.to(dev)
            self._v = self._v.to(dev)
        if self._m.shape != p.shape:
            raise VkError("optimizer state does not match the model's flat parameter buffer")
        if self._step_dev is None or self._step_dev.device != dev:
            start = self.step_count
            self._step_dev = torch.full((1,), start, dtype=torch.int32, device=dev)
            self._scratch = torch.zeros(4, dtype=torch.float32, device=dev)

    @torch.no_grad()
    def zero_grad(self, set_to_none: bool = True):
        # p.grad = None for API fidelity; the flat buffer itself is zeroed lazily by the next backward
        super().zero_grad(set_to_none=True)

    @torch.no_grad()
    def step(self, closure=None, found_inf: torch.Tensor | None = None, grad_scale: torch.Tensor | None = None):
        if closure is not None:
            raise NotImplementedError("closure is not used by the reference")
        m = self._find_model()
        p, g = m.flat_params, m.flat_grads
        if not p.is_cuda:
            raise VkError("parameters are on %s: no CPU fallback" % p.device)
        for q in self.param_groups[0]["params"]:
            if not q.requires_grad:
                raise VkError("FusedAdamW updates the whole flat buffer: frozen parameters (requires_grad=False) are not supported")
        first = self.param_groups[0]["params"][0]
        if first.grad is None:
            raise VkError("optimizer.step() before backward(): gradients are None (zero_grad(set_to_none=True) was the last call)")
        self._ensure_state(p)
        grp = self.param_groups[0]
        # GradScaler support: torch sets these attributes around step() (device tensors; never read on the host here)
        fi = found_inf if found_inf is not None else getattr(self, "found_inf", None)
        gs = grad_scale if grad_scale is not None else getattr(self, "grad_scale", None)
        if gs is not None and not isinstance(gs, torch.Tensor):
            raise VkError("grad_scale must be a device tensor (a float here would have to come from a host sync)")
        if fi is not None:
            fi = fi.reshape(-1)[:1].to(device=p.device, dtype=torch.float32)
        if gs is not None:
            gs = gs.reshape(-1)[:1].to(device=p.device, dtype=torch.float32)
        check(lib().vk_adamw_step_amp(p.numel(), p.data_ptr(), g.data_ptr(), self._m.data_ptr(), self._v.data_ptr(),
                                      float(grp["lr"]), float(grp["betas"][0]), float(grp["betas"][1]), float(grp["eps"]),
                                      float(grp["weight_decay"]), self._step_dev.data_ptr(), float(self.grad_inv_scale),
                                      _lib.ptr(gs), _lib.ptr(fi), self._scratch.data_ptr(), 0, 0, _lib.current_stream()),
              "vk_adamw_step_amp")
        m.mark_weights_dirty()
        return None

    def state_dict(self):
        sd = super().state_dict()
        sd["fused"] = {"step": self.step_count, "exp_avg": self._m, "exp_avg_sq": self._v}
        return sd

    def load_state_dict(self, sd):
        sd = dict(sd)                       # never mutate the caller's dict
        fused = sd.pop("fused", None)
        super().load_state_dict(sd)
        if fused is not None:
            self._pending_step = int(fused["step"])
            self._step_dev = None           # re-created on the parameters' device from _pending_step at the next step()
            self._m, self._v = fused["exp_avg"], fused["exp_avg_sq"]


class GradScaler(torch.amp.GradScaler):
    """``torch.amp.GradScaler('cuda')`` (reference train.py:610-611) whose inf check / unscale of a ``FusedAdamW``'s gradients
    is ONE launch over the model's flat gradient buffer (``vk_amp_unscale_check``) instead of torch's foreach
    kernels over the 140 strided views.  Everything else — scale growth/backoff, ``scale()``, ``update()``, state dict — is
    torch's own code; the stock ``torch.amp.GradScaler`` also works with ``FusedAdamW`` (sync-free as well)."""

    def _unscale_grads_(self, optimizer, inv_scale, found_inf, allow_fp16):
        model = getattr(optimizer, "_model", None)
        if not isinstance(optimizer, FusedAdamW) or model is None or not model.flat_params.is_cuda:
            return super()._unscale_grads_(optimizer, inv_scale, found_inf, allow_fp16)
        g = model.flat_grads
        dev = g.device
        fi = found_inf.to(dev, non_blocking=True) if found_inf.device != dev else found_inf
        inv = inv_scale.to(dev, non_blocking=True) if inv_scale.device != dev else inv_scale
        check(lib().vk_amp_unscale_check(g.numel(), g.data_ptr(), inv.data_ptr(), fi.data_ptr(), _lib.current_stream()),
              "vk_amp_unscale_check")
        return {dev: fi}


def adamw_for(model, lr: float, weight_decay: float = 1e-4, **kw) -> FusedAdamW:
    """``torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=1e-4)`` of train.py:606, fused."""
    return FusedAdamW(model.parameters(), lr=lr, weight_decay=weight_decay, **kw).attach(model)
