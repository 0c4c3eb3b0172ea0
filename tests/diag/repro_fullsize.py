"""Diagnostic: two replicas, same seed, the BASELINE workload (bf16, bs 32, 512x512), three steps each: same bits?"""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
vk = importlib.import_module("vickers-hardness-unet_amd")
dev = torch.device("cuda:0")
finals = []
for rep in range(2):
    vk.seed_everything(21)
    m = vk.Unet(encoder_name="resnet34", encoder_weights=None, in_channels=3, classes=1, activation=None).to(dev).train()
    opt = vk.adamw_for(m, lr=1e-3, weight_decay=1e-4)
    for step in range(3):
        x, y = vk.synthetic_batch(32, 512, seed=300 + step, device=dev) if "device" in vk.synthetic_batch.__code__.co_varnames else vk.synthetic_batch(32, 512, seed=300 + step)
        opt.zero_grad(set_to_none=True)
        out = m.loss_and_backward(x.to(dev), y.to(dev), dtype=torch.bfloat16)
        opt.step()
    torch.cuda.synchronize()
    finals.append(({k: v.detach().clone() for k, v in m.state_dict().items()}, out.clone()))
    del m, opt
(a, la), (b, lb) = finals
bad = [k for k in a if not torch.equal(a[k], b[k])]
print("loss", la.tolist(), lb.tolist(), "equal", torch.equal(la, lb))
print("tensors differing:", len(bad), bad[:5])
