"""Diagnostic: run backward stage 0 (head + decoder blocks 4..2) with the column-staged and the row-staged tile kernels
in one process and report where the intermediate gradient buffers differ."""
import importlib, os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
vk = importlib.import_module("vickers-hardness-unet_amd")
from oracle import unet_oracle as O
L = vk.lib(); L_ = vk._lib
dev = torch.device("cuda:0")
O.set_seed(42); model = vk.Unet(encoder_weights=None).to(dev)
x, y = O.synthetic_batch(2, 64, seed=1234)
x, y = x.to(dev), y.to(dev)
model.train()
plan = model.plan_for(2, 64, torch.float32, True)
names = ["g:decoder.blocks.2.conv1.0", "g:decoder.blocks.1.conv2.0", "gout:encoder.layer1.2", "g:decoder.blocks.2.conv2.0", "g:encoder.conv1"]

def run(nstages):
    st = L_.current_stream()
    logits = model._run_forward(plan, x, True)
    L_.check(L.vk_unet_loss(plan.h, logits.data_ptr(), y.data_ptr(), plan.loss_out.data_ptr(), 1.0, 1.0, 1.0, st))
    L_.check(L.vk_unet_zero_grad(plan.h, st))
    L_.check(L.vk_unet_backward(plan.h, None, 0, nstages, st))
    torch.cuda.synchronize()
    return {n: plan.debug_tensor(n).clone() for n in names}

model._ensure_grads() if hasattr(model, "_ensure_grads") else None
for nst in (1, 2):
    os.environ.pop("VK_HALO_ROWSTAGED", None)
    a = run(nst)
    os.environ["VK_HALO_ROWSTAGED"] = "1"
    b = run(nst)
    print(f"--- after {nst} backward stage(s)")
    for n in names:
        d = (a[n].float() - b[n].float()).abs()
        mx = b[n].float().abs().max().item()
        bad = (d > 1e-5 * mx).nonzero()
        print(f"{n:34s} shape {tuple(a[n].shape)} max|d| {d.max().item():.3e} max|ref| {mx:.3e}  #bad {len(bad)}")
        if len(bad):
            print("    first bad idx (n,y,x,c):", bad[:6].tolist(), " ch set:", sorted(set(bad[:, 3].tolist()))[:20], " y set:", sorted(set(bad[:, 1].tolist())), " x set:", sorted(set(bad[:, 2].tolist())))
