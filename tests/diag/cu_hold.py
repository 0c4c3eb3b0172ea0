"""What losing compute units costs the training step, and what the reducer policies of parallel.py make of it.
RCCL's channel kernels hold CUs while gradient buckets are reduced; the K >= 128 tile kernels and the weight-gradient kernels launch
exactly one workgroup per CU.  `vk_debug_hold_cus` parks W workgroups on a second stream:
   exclusive  : 163,840 B of LDS each -> a held CU takes no other workgroup (worst case: the CU is gone)
   co-resident: 256 threads, no LDS, streaming 16-byte loads (a channel kernel that shares its CU)
Scenarios (bf16, bs 32, 512x512, fused step; ms/step over 8 steps):
   whole step   : holders resident for the whole step            = policy "eager" (collectives from stage 0 on)
   last stage   : holders started when stage 8 has finished, 1 ms = policy "deferred" (one collective under layer 1 + stem)
                  with and without the weight-gradient grid cap (reserved_cus)
   "tail"       : no overlap: step + the collective's own time (not simulated: add ~1 ms)
   python tests/diag/cu_hold.py > profiles/r03/cu_hold.log"""
import importlib
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
vk = importlib.import_module("vickers-hardness-unet_amd")
L = vk.lib()
dev = torch.device("cuda:0")
vk.seed_everything(42)
model = vk.Unet(encoder_weights=None).to(dev)
opt = vk.adamw_for(model, lr=5e-5, weight_decay=1e-4)
x, y = vk.synthetic_batch(32, 512, seed=1234)
x, y = x.to(dev), y.to(dev)
model.train()
side = torch.cuda.Stream()
sink = torch.zeros(4, device=dev)
traffic = torch.zeros(256 << 20, dtype=torch.uint8, device=dev)


def hold(W, mode, us):
    ev = torch.cuda.Event()
    ev.record()                                  # the holders start behind what the compute stream has enqueued so far
    with torch.cuda.stream(side):
        side.wait_event(ev)
        if mode == "exclusive":
            vk._lib.check(L.vk_debug_hold_cus(W, 64, 163840, us, None, 0, sink.data_ptr(), side.cuda_stream))
        else:
            vk._lib.check(L.vk_debug_hold_cus(W, 256, 0, us, traffic.data_ptr(), traffic.numel(), sink.data_ptr(), side.cuda_stream))


class FakeReducer:
    """Stands where a GradientReducer stands (model._reducer): starts the holders where the policy would start its collective."""
    enabled = True
    timing = False

    def __init__(self, W, mode, at_stage, us, reserved):
        self.W, self.mode, self.at, self.us, self.reserved_cus, self.in_flight = W, mode, at_stage, us, reserved, False

    def bucket_ready(self, s, rng):
        if s == self.at:
            hold(self.W, self.mode, self.us)
            self.in_flight = True

    def finish(self):
        self.in_flight = False


def step():
    opt.zero_grad(set_to_none=True)
    model.loss_and_backward(x, y, dtype=torch.bfloat16)
    opt.step()


def run(red=None, whole=None, steps=8):
    model._reducer = None
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    model._reducer = red
    if whole:
        hold(whole[0], whole[1], 400000)
        time.sleep(0.02)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / steps
    torch.cuda.synchronize()
    model._reducer = None
    return ms


print(f"device: {torch.cuda.get_device_name(dev)}")
base = run()
print(f"no holders                                              : {base:7.3f} ms/step")
for mode in ("exclusive", "co-resident"):
    for W in (8, 16, 32):
        ms = run(whole=(W, mode))
        print(f"whole step  {W:3d} {mode:11s} holders                 : {ms:7.3f} ms/step  ({100 * (ms / base - 1):+5.1f} %)")
for mode in ("exclusive", "co-resident"):
    for W in (16, 32):
        for reserved in (0, W):
            ms = run(red=FakeReducer(W, mode, 8, 1000, reserved))
            print(f"last stage  {W:3d} {mode:11s} holders 1 ms, reserved {reserved:2d} : {ms:7.3f} ms/step  ({100 * (ms / base - 1):+5.1f} %)")
for W in (16,):
    for reserved in (0, W):
        ms = run(red=FakeReducer(W, "exclusive", 0, 10000, reserved))
        print(f"from stage 0 {W:2d} exclusive   holders 10 ms, reserved {reserved:2d}: {ms:7.3f} ms/step  ({100 * (ms / base - 1):+5.1f} %)")
print(f"no holders (again)                                      : {run():7.3f} ms/step")
print("policy \"tail\": no overlap = the no-holder step + the collective's own time (97.7 MB fp32 over 8 ranks: ~1 ms, not simulated here)")
