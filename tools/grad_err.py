"""Diagnostic: per-parameter gradient error of the fp32 plan against the CPU oracle (N=2, S=64), max-abs and L2 relative."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
vk = importlib.import_module("vickers-hardness-unet_amd")
from oracle import unet_oracle as O
dev = torch.device("cuda:0")
O.set_seed(42); ref = O.build_model()
O.set_seed(42); model = vk.Unet(encoder_weights=None).to(dev)
SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 1234
x, y = O.synthetic_batch(2, 64, seed=SEED)
ref.train(); model.train()
lo = ref(x); O.total_loss(lo, y).backward()
lg = model(x.to(dev))
(torch.nn.BCEWithLogitsLoss()(lg, y.to(dev)) + vk.DiceLoss(mode="binary")(lg, y.to(dev))).backward()
torch.cuda.synchronize()
print("logit err", (lg.detach().cpu() - lo.detach()).abs().max().item())
named_o = dict(ref.named_parameters())
rows = []
for k, p in model.named_parameters():
    go, gg = named_o[k].grad, p.grad.cpu()
    rows.append((((gg - go).abs().max() / (go.abs().max() + 1e-12)).item(), ((gg - go).norm() / (go.norm() + 1e-12)).item(), k))
rows.sort(reverse=True)
if len(sys.argv) <= 2:
    for r in rows[:8]:
        print(f"max-rel {r[0]:.4f}  l2-rel {r[1]:.5f}  {r[2]}")
print("seed", SEED, "worst max-rel %.4f (%s)" % (rows[0][0], rows[0][2]), " #params>2e-2:", sum(r[0] > 2e-2 for r in rows))
print("median max-rel", sorted(r[0] for r in rows)[len(rows) // 2], " worst l2-rel", max(r[1] for r in rows))
