"""``Unet`` — drop-in for ``segmentation_models_pytorch.Unet("resnet34", in_channels=3, classes=1)``
as the reference builds it (train.py:357-379 ``build_model``; infer_pth_gui.py:31-33; ui_infer_*.py
``Segmenter``), executed by the HIP engine in libvkunet.so.

API surface kept: constructor arguments, ``forward(x[N,3,S,S] fp32) -> logits
[N,1,S,S] fp32``, ``train()/eval()``, ``.to(device)``, ``parameters()``, ``state_dict()`` /
``load_state_dict(strict=True)`` with smp's 278 keys (conv weights are logical OIHW tensors whose memory
is KRSC = torch channels_last, all living in one flat fp32 buffer), autograd participation
(``loss.backward()`` fills ``p.grad``), ``torch.autocast`` selects the 16-bit plan exactly where the
reference autocasts (train.py:431-435) while un-autocast calls (validate, train.py:510) run fp32.

There is no CPU execution path: a CPU tensor or a missing libvkunet.so raises.
"""
from __future__ import annotations

import os
import ctypes as C
import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import VkError, check, lib


class _Holder(nn.Module):
    """Attribute container so that parameter names reproduce smp's module tree."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("structural container only; call the Unet")


class _Plan:
    """One engine handle + workspace for a fixed (N, size, dtype, training) — static shapes, static
    addresses (hipGraph-friendly)."""

    def __init__(self, model: "Unet", N: int, S, dtype: torch.dtype, training: bool):
        L = lib()
        H, W = (S, S) if isinstance(S, int) else S          # S: the side of a square input or (height, width)
        self.key = (N, S, dtype, training)
        self.N, self.S, self.H, self.W, self.dtype, self.training = N, S, H, W, dtype, training
        cfg = _lib.vk_unet_config(N, H, _lib.dtype_code(dtype), 1 if training else 0, W)
        h = C.c_void_p()
        check(L.vk_unet_create(C.byref(cfg), C.byref(h)), "vk_unet_create")
        self.h = h
        self.ws_bytes = L.vk_unet_workspace_bytes(h)
        dev = model._flat["params"].device
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
        self.loss_out = torch.zeros(4, dtype=torch.float32, device=dev)
        self.weights_version = None
        self.nbuckets = L.vk_unet_num_buckets(h)
        self.buckets: List[Tuple[int, int]] = []
        for b in range(self.nbuckets):
            b0, b1 = C.c_int64(), C.c_int64()
            check(L.vk_unet_bucket_range(h, b, C.byref(b0), C.byref(b1)))
            self.buckets.append((b0.value, b1.value))
        self.bind(model)

    def bind(self, model: "Unet"):
        f = model._flat
        grads = model._ensure_grads() if self.training else None
        check(lib().vk_unet_bind(self.h, f["params"].data_ptr(), _lib.ptr(grads), f["bufs"].data_ptr(),
                                 f["nbt"].data_ptr(), self.ws.data_ptr(), self.ws_bytes), "vk_unet_bind")
        self.weights_version = None

    def debug_tensor(self, name: str) -> torch.Tensor:
        """Copy of a named intermediate (NHWC) — parity/debug only."""
        p = C.c_void_p()
        dims = (C.c_int * 4)()
        check(lib().vk_unet_debug_tensor(self.h, name.encode(), C.byref(p), C.byref(dims)), "vk_unet_debug_tensor")
        shape = [d for d in dims]
        is_f32 = name.startswith(("scale:", "shift:", "dlogits"))
        dt = torch.float32 if is_f32 else self.dtype
        off = p.value - self.ws.data_ptr()
        n = 1
        for d in shape:
            n *= d
        nbytes = n * torch.empty((), dtype=dt).element_size()
        return self.ws[off:off + nbytes].view(dt).view(shape).clone()

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().vk_unet_destroy(self.h)
                self.h = None
        except Exception:
            pass


class _UnetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, model, plan):
        ctx.model, ctx.plan = model, plan
        return model._run_forward(plan, x, True)

    @staticmethod
    def backward(ctx, g):
        ctx.model._run_backward(ctx.plan, g.contiguous().float())
        return None, None, None, None


class Unet(nn.Module):
    def __init__(self, encoder_name: str = "resnet34", encoder_depth: int = 5, encoder_weights: Optional[str] = "imagenet",
                 decoder_use_batchnorm: bool = True, decoder_channels=(256, 128, 64, 32, 16),
                 decoder_attention_type: Optional[str] = None, in_channels: int = 3, classes: int = 1,
                 activation: Optional[str] = None, aux_params: Optional[dict] = None, *,
                 compute_dtype: torch.dtype = torch.float32):
        super().__init__()
        if encoder_name != "resnet34":
            raise NotImplementedError("only the reference's configuration encoder_name='resnet34' is implemented")
        if encoder_weights is not None:
            raise VkError("encoder_weights=%r needs a checkpoint download; no network here — pass None and "
                          "load_state_dict() a checkpoint instead" % (encoder_weights,))
        if (encoder_depth != 5 or tuple(decoder_channels) != (256, 128, 64, 32, 16) or not decoder_use_batchnorm
                or decoder_attention_type is not None or in_channels != 3 or classes != 1 or activation is not None
                or aux_params is not None):
            raise NotImplementedError("only in_channels=3, classes=1, activation=None, default decoder are implemented "
                                      "(reference train.py:372-378)")
        self.compute_dtype = compute_dtype
        L = lib()
        cfg = _lib.vk_unet_config(1, 32, _lib.VK_F32, 0)
        h = C.c_void_p()
        check(L.vk_unet_create(C.byref(cfg), C.byref(h)), "vk_unet_create")
        try:
            self._table = []
            for i in range(L.vk_unet_num_tensors(h)):
                ti = _lib.vk_tensor_info()
                check(L.vk_unet_tensor_info(h, i, C.byref(ti)))
                self._table.append((ti.name.decode(), ti.kind, [ti.dims[j] for j in range(ti.ndim)], ti.offset, ti.numel))
            n_params = L.vk_unet_param_numel(h)
            n_bufs = L.vk_unet_buffer_numel(h)
        finally:
            L.vk_unet_destroy(h)
        n_bn = sum(1 for t in self._table if t[1] == 3)
        self._flat: Dict[str, Optional[torch.Tensor]] = {
            "params": torch.zeros(n_params, dtype=torch.float32),
            "grads": None,
            "bufs": torch.zeros(n_bufs, dtype=torch.float32),
            "nbt": torch.zeros(n_bn, dtype=torch.int64),
        }
        self._leaves: Dict[str, Tuple[nn.Module, str]] = {}
        self._plans: Dict[tuple, _Plan] = {}
        self._dirty = 0
        self._reducer = None
        self._time_groups = None
        self._group_events = []
        self._anchor = torch.zeros((), requires_grad=True)
        self._build_tree()
        self._rebuild_views()
        self._default_init()

    # ------------------------------------------------------------------ structure
    def _view(self, kind: int, dims: List[int], off: int, numel: int, which: str = "params") -> torch.Tensor:
        if kind == 0:
            K, Cc, R, S = dims
            return self._flat[which][off:off + numel].view(K, R, S, Cc).permute(0, 3, 1, 2)
        if kind == 1:
            return self._flat[which][off:off + numel]
        if kind == 2:
            return self._flat["bufs"][off:off + numel]
        return self._flat["nbt"][off]

    def _build_tree(self):
        for name, kind, dims, off, numel in self._table:
            parts = name.split(".")
            mod: nn.Module = self
            for p in parts[:-1]:
                if p not in mod._modules:
                    mod.add_module(p, _Holder())
                mod = mod._modules[p]
            leaf = parts[-1]
            if kind in (0, 1):
                mod.register_parameter(leaf, nn.Parameter(torch.empty(0), requires_grad=True))
            else:
                mod.register_buffer(leaf, torch.empty(0))
            self._leaves[name] = (mod, leaf)

    def _rebuild_views(self):
        for name, kind, dims, off, numel in self._table:
            mod, leaf = self._leaves[name]
            v = self._view(kind, dims, off, numel)
            if kind in (0, 1):
                p = mod._parameters[leaf]
                p.data = v
                p.grad = None
            else:
                mod._buffers[leaf] = v
        self._anchor = torch.zeros((), requires_grad=True, device=self._flat["params"].device)
        self._param_list = [self._leaves[t[0]][0]._parameters[self._leaves[t[0]][1]] for t in self._table if t[1] in (0, 1)]

    def _apply(self, fn, recurse=True):
        for k in ("params", "grads", "bufs", "nbt"):
            t = self._flat[k]
            if t is not None:
                nt = fn(t)
                want = torch.int64 if k == "nbt" else torch.float32
                if nt.dtype != want:
                    raise VkError("the fp32 master copy cannot be converted to %s; choose the compute type with "
                                  "torch.autocast or compute_dtype=" % nt.dtype)
                self._flat[k] = nt
        self._rebuild_views()
        self._plans.clear()
        self._dirty += 1
        return self

    def _default_init(self):
        """smp / torchvision default initialisation, drawing from torch's global RNG in the same order
        and with the same shapes as the constructors would ([upstream]), so that
        ``set_seed(42); Unet(...)`` reproduces the oracle's weights."""
        sd = {name: self._view(kind, dims, off, numel) for name, kind, dims, off, numel in self._table}

        def ctor_draw(shape, bias=False):   # nn.Conv2d / nn.Linear reset_parameters()
            w = torch.empty(shape)
            nn.init.kaiming_uniform_(w, a=math.sqrt(5))
            if bias:
                fan_in = w[0].numel()
                bound = 1 / math.sqrt(fan_in)
                nn.init.uniform_(torch.empty(shape[0]), -bound, bound)

        with torch.no_grad():
            for name, kind, dims, off, numel in self._table:
                if kind == 1 and name.endswith(".weight") and len(dims) == 1:
                    sd[name].fill_(1.0)                       # BN gamma
                elif kind == 2 and name.endswith("running_var"):
                    sd[name].fill_(1.0)
            enc = [t for t in self._table if t[1] == 0 and t[0].startswith("encoder.")]
            dec = [t for t in self._table if t[1] == 0 and t[0].startswith("decoder.")]
            head = [t for t in self._table if t[1] == 0 and t[0].startswith("segmentation_head.")][0]
            # torchvision constructs conv1, then per block conv1, conv2, (downsample) — but the
            # downsample conv is built BEFORE the block in _make_layer
            def ctor_order(ts):
                out, i = [], 0
                while i < len(ts):
                    if i + 2 < len(ts) and ".downsample.0" in ts[i + 2][0]:
                        out += [ts[i + 2], ts[i], ts[i + 1]]
                        i += 3
                    else:
                        out.append(ts[i])
                        i += 1
                return out
            for t in ctor_order(enc):
                ctor_draw(t[2])
            ctor_draw([1000, 512], bias=True)                  # the fc layer torchvision builds and smp deletes
            for t in enc:                                      # ResNet.__init__ init loop (module order)
                w = torch.empty(t[2])
                nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")
                sd[t[0]].copy_(w)
            for t in dec:
                ctor_draw(t[2])
            ctor_draw(head[2], bias=True)
            for t in dec:                                      # smp initialize_decoder
                w = torch.empty(t[2])
                nn.init.kaiming_uniform_(w, mode="fan_in", nonlinearity="relu")
                sd[t[0]].copy_(w)
            w = torch.empty(head[2])                           # smp initialize_head
            nn.init.xavier_uniform_(w)
            sd[head[0]].copy_(w)
            sd["segmentation_head.0.bias"].zero_()
        self._dirty += 1

    # ------------------------------------------------------------------ flat-buffer access (optimizer / DP)
    @property
    def flat_params(self) -> torch.Tensor:
        return self._flat["params"]

    @property
    def flat_grads(self) -> torch.Tensor:
        return self._ensure_grads()

    def _ensure_grads(self) -> torch.Tensor:
        if self._flat["grads"] is None:
            self._flat["grads"] = torch.zeros_like(self._flat["params"])
        return self._flat["grads"]

    def _attach_grads(self):
        g = self._ensure_grads()
        for name, kind, dims, off, numel in self._table:
            if kind in (0, 1):
                mod, leaf = self._leaves[name]
                mod._parameters[leaf].grad = self._view(kind, dims, off, numel, "grads")

    def mark_weights_dirty(self):
        """Call after writing the flat parameter buffer through a raw pointer (FusedAdamW does)."""
        self._dirty += 1

    def _weights_version(self):
        # Parameter.data views carry their own version counters, so sum them (in-place edits by a stock
        # torch optimizer or load_state_dict bump them); raw-pointer writers call mark_weights_dirty().
        return (self._dirty, sum(p._version for p in self._param_list))

    # ------------------------------------------------------------------ plans
    def plan_for(self, N: int, S, dtype: torch.dtype, training: bool) -> _Plan:
        key = (N, S, dtype, training)
        p = self._plans.get(key)
        if p is None and not training:
            p = self._plans.get((N, S, dtype, True))    # a training plan can also run eval forwards
        if p is None:
            p = _Plan(self, N, S, dtype, training)
            self._plans[key] = p
        return p

    def _check_input(self, x: torch.Tensor):
        if not isinstance(x, torch.Tensor) or x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("expected input [N,3,H,W], got %s" % (tuple(x.shape),))
        h, w = x.shape[-2:]
        if h % 32 or w % 32:
            raise RuntimeError(f"Wrong input shape height={h}, width={w}. Expected image height and width divisible by 32.")
        if not x.is_cuda:
            raise VkError("input is on %s: this package runs on an MI355X only and has no CPU fallback" % x.device)
        if self._flat["params"].device != x.device:
            raise VkError("model is on %s but the input is on %s" % (self._flat["params"].device, x.device))

    def _run_forward(self, plan: _Plan, x: torch.Tensor, training: bool) -> torch.Tensor:
        L = lib()
        st = _lib.current_stream()
        ver = self._weights_version()
        if plan.weights_version != ver:
            check(L.vk_unet_refresh_weights(plan.h, st), "vk_unet_refresh_weights")
            plan.weights_version = ver
        x = x.detach().contiguous().float()
        logits = torch.empty(plan.N, 1, plan.H, plan.W, dtype=torch.float32, device=x.device)
        check(L.vk_unet_forward(plan.h, x.data_ptr(), logits.data_ptr(), 1 if training else 0, st), "vk_unet_forward")
        plan._last_x = x      # keep the input alive until backward has consumed the plan's x4 copy
        return logits

    def _run_backward(self, plan: _Plan, dlogits: Optional[torch.Tensor]):
        L = lib()
        st = _lib.current_stream()
        first = next(iter(self.parameters()))
        if first.grad is None:
            check(L.vk_unet_zero_grad(plan.h, st), "vk_unet_zero_grad")
            self._attach_grads()
        red = self._reducer
        if red is not None and getattr(red, "enabled", True) and not getattr(plan, "_side_off", False):
            check(L.vk_unet_set_side_stream(plan.h, 0), "vk_unet_set_side_stream")    # see include/vk_unet.h
            plan._side_off = True
        capped = False
        # bench.py's overlap budget: `_time_groups` = [begin, end) stage ranges to run as separate calls with a HIP event pair around
        # each (the grouping of the "deferred" policy is [(0, 9), (9, 10)]); durations are read with group_times_ms()
        tg = getattr(self, "_time_groups", None)
        if red is None and tg is None and not os.environ.get("VK_BACKWARD_PER_STAGE"):
            # nobody needs a stage's gradients before the end: one call, so that the weight gradients of ALL stages run as one batch
            check(L.vk_unet_backward(plan.h, _lib.ptr(dlogits), 0, plan.nbuckets, st), "vk_unet_backward")
            return
        if os.environ.get("VK_BACKWARD_PER_STAGE"):
            groups = [(s, s + 1) for s in range(plan.nbuckets)]
        elif red is not None and hasattr(red, "stage_groups"):
            groups = red.stage_groups(plan.nbuckets)
        elif tg is not None:
            groups = [tuple(g) for g in tg]
        else:
            groups = [(s, s + 1) for s in range(plan.nbuckets)]
        timed = tg is not None or (red is not None and getattr(red, "timing", False))
        try:
            for s0, s1 in groups:
                if red is not None and getattr(red, "reserved_cus", 0) > 0 and getattr(red, "in_flight", False) and not capped:
                    L.vk_set_reserved_cus(red.reserved_cus)      # collectives share the chip from here on: see parallel.py
                    capped = True
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                check(L.vk_unet_backward(plan.h, _lib.ptr(dlogits), s0, s1, st), "vk_unet_backward")
                if timed:
                    e1.record()
                    self._group_events.append((s0, s1, e0, e1))
                if red is not None:
                    for s in range(s0, s1):
                        red.bucket_ready(s, plan.buckets[s])
        finally:
            if capped:
                L.vk_set_reserved_cus(0)
        if red is not None:
            red.finish()

    def group_times_ms(self):
        """Durations of the timed backward groups since the last call: {(begin, end): [ms per step, ...]} (synchronises)."""
        out = {}
        for s0, s1, e0, e1 in self._group_events:
            e1.synchronize()
            out.setdefault((s0, s1), []).append(e0.elapsed_time(e1))
        self._group_events.clear()
        return out

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._check_input(x)
        N, _, H, W = x.shape
        S = H if H == W else (H, W)           # plans are keyed by the side of a square input or by (height, width)
        if torch.is_autocast_enabled():
            dtype = torch.get_autocast_dtype('cuda')
        else:
            dtype = self.compute_dtype
        need_grad = self.training and torch.is_grad_enabled()
        plan = self.plan_for(N, S, dtype, need_grad or self.training)
        if need_grad:
            return _UnetFn.apply(x, self._anchor, self, plan)
        return self._run_forward(plan, x, self.training)

    # ------------------------------------------------------------------ fused step (no autograd graph)
    def loss_and_backward(self, x: torch.Tensor, y: torch.Tensor, grad_scale: float = 1.0,
                          dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        """forward (batch-stat BN) + BCE+Dice + backward in one call; the engine's loss kernel feeds the
        head gradient directly.  Returns a device tensor [total, bce, dice] (no host sync).
        Equivalent to train.py:436-448 ``logits = model(x); loss = bce + dice; loss.backward()``."""
        self._check_input(x)
        N, _, H, W = x.shape
        plan = self.plan_for(N, H if H == W else (H, W), dtype or self.compute_dtype, True)
        logits = self._run_forward(plan, x, True)
        y = y.detach().contiguous().float()
        check(lib().vk_unet_loss(plan.h, logits.data_ptr(), y.data_ptr(), plan.loss_out.data_ptr(), float(grad_scale),
                                 1.0, 1.0, _lib.current_stream()), "vk_unet_loss")
        self._run_backward(plan, None)
        self.last_logits = logits
        return plan.loss_out[:3]


def build_model(encoder: str = "resnet34", weights: Optional[str] = "imagenet") -> Unet:
    """Mirror of reference train.py:357-379."""
    return Unet(encoder_name=encoder, encoder_weights=weights, in_channels=3, classes=1, activation=None)
