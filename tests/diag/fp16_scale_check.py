"""Diagnostic: fp16 gradients at 1024x1024 with and without the reference's loss scale (GradScaler default 2^16) against the fp32 plan."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
vk = importlib.import_module("vickers-hardness-unet_amd")
from oracle import unet_oracle as O
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
O.set_seed(42); model = vk.Unet(encoder_weights=None).to(dev)
x, y = O.synthetic_batch(N, S, seed=1234)
x, y = x.to(dev), y.to(dev)
model.train()


def grads(dtype, scale):
    for p in model.parameters():
        p.grad = None
    model.loss_and_backward(x, y, grad_scale=scale, dtype=dtype)
    torch.cuda.synchronize()
    return (model.flat_grads / scale).double().clone()


def cos(a, b):
    return (a @ b / (a.norm() * b.norm() + 1e-300)).item()


g32 = grads(torch.float32, 1.0)
for name, dt, sc in (("fp16 unscaled", torch.float16, 1.0), ("fp16 x 2^16", torch.float16, 65536.0), ("fp16 x 2^12", torch.float16, 4096.0),
                     ("bf16", torch.bfloat16, 1.0)):
    g = grads(dt, sc)
    print(f"{name:14s} finite {bool(torch.isfinite(g).all())}  cos vs fp32 {cos(g, g32):.5f}  |g|/|g32| {(g.norm() / g32.norm()).item():.4f}")
    # per-layer-group cosine
    tab = {t[0]: (t[3], t[4]) for t in model._table if t[1] == 0}
    for k in ("encoder.conv1.weight", "encoder.layer1.0.conv1.weight", "encoder.layer3.0.conv1.weight", "decoder.blocks.4.conv2.0.weight"):
        o, n = tab[k]
        print(f"      {k:36s} cos {cos(g[o:o + n], g32[o:o + n]):.5f}")
