"""Diagnostic: per-parameter gradient error of the fp32 plan against the CPU oracle, with the oracle ALSO run in float64 as the
arbiter (which of the two fp32 implementations is closer to exact arithmetic?).

    python tests/diag/grad_err.py [N] [S] [seed]
"""
import copy
import importlib
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
vk = importlib.import_module("vickers-hardness-unet_amd")
from oracle import unet_oracle as O  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
SEED = int(sys.argv[3]) if len(sys.argv) > 3 else 1234
O.set_seed(42); ref = O.build_model()
O.set_seed(42); model = vk.Unet(encoder_weights=None).to(dev)
ref64 = copy.deepcopy(ref).double()
x, y = O.synthetic_batch(N, S, seed=SEED)
ref.train(); model.train(); ref64.train()


def relu_masks(m, xin):
    masks, hooks = [], []
    for _, mod in m.named_modules():
        if isinstance(mod, torch.nn.ReLU):
            hooks.append(mod.register_forward_hook(lambda mm, i, o: masks.append((o > 0).clone())))
    out = m(xin)
    for h in hooks:
        h.remove()
    return out, masks


lo, m32 = relu_masks(ref, x)
O.total_loss(lo, y).backward()
lo64, m64 = relu_masks(ref64, x.double())
O.total_loss(lo64, y.double()).backward()
lg = model(x.to(dev))
(torch.nn.BCEWithLogitsLoss()(lg, y.to(dev)) + vk.DiceLoss(mode="binary")(lg, y.to(dev))).backward()
torch.cuda.synchronize()
flips32 = sum(int((a != b).sum()) for a, b in zip(m32, m64))
total = sum(a.numel() for a in m32)
print(f"N={N} S={S} seed={SEED}: logit err engine-vs-oracle32 {(lg.detach().cpu() - lo.detach()).abs().max().item():.3e}, "
      f"oracle32-vs-oracle64 {(lo.detach().double() - lo64.detach()).abs().max().item():.3e}; "
      f"ReLU decisions oracle32 vs oracle64: {flips32} of {total} differ")
n32, n64 = dict(ref.named_parameters()), dict(ref64.named_parameters())
rows = []
for k, p in model.named_parameters():
    g64 = n64[k].grad
    gg, go = p.grad.cpu().double(), n32[k].grad.double()
    den = g64.norm() + 1e-30
    rows.append((((gg - go).norm() / (go.norm() + 1e-30)).item(), ((gg - g64).norm() / den).item(), ((go - g64).norm() / den).item(), k))
rows.sort(reverse=True)
print("  L2-rel: engine-vs-oracle32 | engine-vs-oracle64 | oracle32-vs-oracle64")
for r in rows[:8]:
    print(f"  {r[0]:.5f} | {r[1]:.5f} | {r[2]:.5f}   {r[3]}")
print(f"  worst engine-vs-32 {rows[0][0]:.5f}; worst engine-vs-64 {max(r[1] for r in rows):.5f}; worst oracle32-vs-64 {max(r[2] for r in rows):.5f}; "
      f"#params engine-vs-32 > 1e-2: {sum(r[0] > 1e-2 for r in rows)}; median engine-vs-32 {sorted(r[0] for r in rows)[len(rows) // 2]:.2e}")
