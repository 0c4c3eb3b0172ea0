"""Diagnostic: fused eval decoder tail vs separate launches, several runs each; where do they differ?"""
import importlib, os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
vk = importlib.import_module("vickers-hardness-unet_amd")
from oracle import unet_oracle as O
dev = torch.device("cuda:0")
O.set_seed(5)
model = vk.Unet(encoder_name="resnet34", encoder_weights=None, in_channels=3, classes=1, activation=None).to(dev).eval()
g = torch.Generator().manual_seed(9)
with torch.no_grad():
    for _, bm in model.named_buffers():
        if bm.dtype == torch.float32:
            v = torch.rand(bm.shape, generator=g) * 0.5 + (0.75 if bm.min() >= 1.0 else -0.25)
            bm.copy_(v.to(bm.device))
model.mark_weights_dirty()
for (n, s) in [(2, 64), (1, 160)]:
    x, _ = O.synthetic_batch(n, s, seed=77)
    xd = x.to(dev)
    def run(fuse):
        if fuse: os.environ.pop("VK_NO_TAIL_FUSION", None)
        else: os.environ["VK_NO_TAIL_FUSION"] = "1"
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return model(xd).float().clone()
    outs = [("sep", run(False)), ("fus", run(True)), ("sep", run(False)), ("fus", run(True)), ("fus", run(True)), ("sep", run(False))]
    torch.cuda.synchronize()
    base = outs[0][1]
    for i, (k, o) in enumerate(outs):
        d = (o - base).abs()
        nz = (d > 0).nonzero()
        print(n, s, i, k, "max diff vs run0", d.max().item(), "count", nz.shape[0], "first", nz[:3].tolist() if nz.shape[0] else None)
