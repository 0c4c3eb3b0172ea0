"""``FusedAdamW`` — ``torch.optim.AdamW(model.parameters(), lr, weight_decay=1e-4)`` (reference
train.py:606) as ONE HIP launch over the model's flat fp32 parameter / gradient / moment buffers.

It is a real ``torch.optim.Optimizer`` (single param group), so ``CosineAnnealingLR`` (train.py:607),
``optimizer.param_groups[0]["lr"]`` (train.py:656), ``zero_grad(set_to_none=True)`` (train.py:428) and
``GradScaler.step(optimizer)`` (train.py:444) work unchanged.  Arithmetic follows
torch/optim/adam.py's single-tensor path (decoupled decay, bias-corrected)."""
from __future__ import annotations

import torch

from . import _lib
from ._lib import VkError, check, lib


class FusedAdamW(torch.optim.Optimizer):
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
        self._step = 0
        self.grad_inv_scale = 1.0     # extra factor applied to gradients inside the kernel (DP averaging)

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

    @torch.no_grad()
    def zero_grad(self, set_to_none: bool = True):
        # p.grad = None for API fidelity; the flat buffer itself is zeroed lazily by the next backward
        super().zero_grad(set_to_none=True)

    @torch.no_grad()
    def step(self, closure=None, found_inf: torch.Tensor | None = None):
        if closure is not None:
            raise NotImplementedError("closure is not used by the reference")
        m = self._find_model()
        p, g = m.flat_params, m.flat_grads
        if not p.is_cuda:
            raise VkError("parameters are on %s: no CPU fallback" % p.device)
        if self._m is None or self._m.device != p.device:
            self._m = torch.zeros_like(p)
            self._v = torch.zeros_like(p)
        grp = self.param_groups[0]
        self._step += 1
        # GradScaler support: torch passes found_inf / grad_scale through these attributes
        fi = found_inf if found_inf is not None else getattr(self, "found_inf", None)
        inv = float(self.grad_inv_scale)
        gs = getattr(self, "grad_scale", None)
        if gs is not None:
            inv = inv / float(gs)
        fi_i32 = None
        if fi is not None:
            fi_i32 = (fi.reshape(-1)[:1] != 0).to(torch.int32)
        check(lib().vk_adamw_step(p.numel(), p.data_ptr(), g.data_ptr(), self._m.data_ptr(), self._v.data_ptr(),
                                  float(grp["lr"]), float(grp["betas"][0]), float(grp["betas"][1]), float(grp["eps"]),
                                  float(grp["weight_decay"]), self._step, inv, _lib.ptr(fi_i32), 0, 0,
                                  _lib.current_stream()), "vk_adamw_step")
        m.mark_weights_dirty()
        return None

    def state_dict(self):
        sd = super().state_dict()
        sd["fused"] = {"step": self._step, "exp_avg": self._m, "exp_avg_sq": self._v}
        return sd

    def load_state_dict(self, sd):
        fused = sd.pop("fused", None)
        super().load_state_dict(sd)
        if fused is not None:
            self._step = fused["step"]
            self._m, self._v = fused["exp_avg"], fused["exp_avg_sq"]


def adamw_for(model, lr: float, weight_decay: float = 1e-4, **kw) -> FusedAdamW:
    """``torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=1e-4)`` of train.py:606, fused."""
    return FusedAdamW(model.parameters(), lr=lr, weight_decay=weight_decay, **kw).attach(model)
