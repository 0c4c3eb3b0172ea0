"""What losing compute units costs the training step: RCCL's channel kernels hold CUs while gradient buckets are
reduced, and the K >= 128 tile kernels / weight-gradient kernels launch exactly one workgroup per CU (<= 256 workgroups).
`vk_debug_hold_cus` parks W workgroups on a second stream for the duration of a few steps:
   exclusive  : 163,840 B of LDS each -> a held CU takes no other workgroup (worst case: the CU is gone)
   co-resident: 256 threads, no LDS, streaming 16-byte loads (what a channel kernel looks like: it shares the CU)
Prints ms/step (bf16, bs 32, 512x512, fused step) for W = 0 / 8 / 16 / 32, and with VK_RESERVED_CUS set to W (grid cap of the
persistent <= 256-workgroup kernels).   python tests/diag/cu_hold.py > profiles/r03/cu_hold.log"""
import importlib
import os
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


def step():
    opt.zero_grad(set_to_none=True)
    model.loss_and_backward(x, y, dtype=torch.bfloat16)
    opt.step()


def run(W, mode, steps=8):
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    if W:
        with torch.cuda.stream(side):
            if mode == "exclusive":
                vk._lib.check(L.vk_debug_hold_cus(W, 64, 163840, 400000, None, 0, sink.data_ptr(), side.cuda_stream))
            else:
                vk._lib.check(L.vk_debug_hold_cus(W, 256, 0, 400000, traffic.data_ptr(), traffic.numel(), sink.data_ptr(), side.cuda_stream))
        time.sleep(0.02)            # the holders are resident before the first timed launch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / steps
    torch.cuda.synchronize()        # the holders time out by themselves (0.4 s)
    return ms


print(f"device: {torch.cuda.get_device_name(dev)}; VK_RESERVED_CUS={os.environ.get('VK_RESERVED_CUS', '')}")
base = run(0, "none")
print(f"no holders                         : {base:7.3f} ms/step")
for mode in ("exclusive", "co-resident"):
    for W in (8, 16, 32):
        ms = run(W, mode)
        print(f"{W:3d} {mode:11s} holder workgroups : {ms:7.3f} ms/step  ({100 * (ms / base - 1):+5.1f} %)")
print(f"no holders (again)                 : {run(0, 'none'):7.3f} ms/step")
