"""Diagnostic: is the 16-bit eval forward of small plans run-to-run deterministic?  Toggles isolate the kernel class."""
import importlib, os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
vk = importlib.import_module("vickers-hardness-unet_amd")
from oracle import unet_oracle as O
dev = torch.device("cuda:0")
rand_stats = os.environ.get("DIAG_RANDOM_STATS", "1") == "1"
def build():
    O.set_seed(5)
    m = vk.Unet(encoder_name="resnet34", encoder_weights=None, in_channels=3, classes=1, activation=None).to(dev).eval()
    if rand_stats:
        g = torch.Generator().manual_seed(9)
        with torch.no_grad():
            for _, bm in m.named_buffers():
                if bm.dtype == torch.float32:
                    v = torch.rand(bm.shape, generator=g) * 0.5 + (0.75 if bm.min() >= 1.0 else -0.25)
                    bm.copy_(v.to(bm.device))
        m.mark_weights_dirty()
    return m
for toggles in [{}, {"VK_NO_TAIL_FUSION": "1"}, {"VK_NO_TAIL_FUSION": "1", "VK_NO_STREAM": "1"}, {"VK_NO_TAIL_FUSION": "1", "VK_NO_SPLITK": "1"},
                {"VK_NO_TAIL_FUSION": "1", "VK_NO_SPLITK": "1", "VK_NO_STREAM": "1"}, {"VK_NO_TAIL_FUSION": "1", "VK_NO_HALO": "1"}]:
    for k in ("VK_NO_TAIL_FUSION", "VK_NO_STREAM", "VK_NO_SPLITK", "VK_NO_HALO"):
        os.environ.pop(k, None)
    os.environ.update(toggles)
    model = build()
    for (n, s, dt) in [(2, 64, torch.bfloat16), (1, 160, torch.bfloat16), (1, 160, torch.float32)]:
        x, _ = O.synthetic_batch(n, s, seed=77)
        xd = x.to(dev)
        outs = []
        for i in range(8):
            with torch.no_grad():
                if dt == torch.float32:
                    outs.append(model(xd).float().clone())
                else:
                    with torch.autocast("cuda", dtype=dt):
                        outs.append(model(xd).float().clone())
        torch.cuda.synchronize()
        bad = [i for i in range(1, 8) if not torch.equal(outs[i], outs[0])]
        print(toggles, n, s, str(dt).split(".")[-1], "runs differing from run 0:", bad, flush=True)
