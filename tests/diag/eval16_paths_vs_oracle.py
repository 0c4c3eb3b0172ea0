"""Diagnostic (r04): 16-bit inference through the three kernel families (default / tile kernels only / tap-by-tap) against the fp32
oracle, with the oracle under CPU autocast as the yardstick — the numbers behind the bars of
tests/test_model_gpu.py::test_eval_16bit_kernel_paths_agree_and_repeat."""
import importlib, os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
vk = importlib.import_module("vickers-hardness-unet_amd")
from oracle import unet_oracle as O
dev = torch.device("cuda:0")


def build():
    O.set_seed(11)
    return vk.Unet(encoder_name="resnet34", encoder_weights=None, in_channels=3, classes=1, activation=None).to(dev).eval()


for (n, s) in [(2, 64), (1, 160), (3, 96), (4, 256)]:
    x, _ = O.synthetic_batch(n, s, seed=91)
    O.set_seed(11)
    ref = O.build_model().eval()
    with torch.no_grad():
        lo = ref(x)
    for dtype in (torch.bfloat16, torch.float16):
        with torch.no_grad(), torch.autocast("cpu", dtype=dtype):
            yard = ref(x).float()
        outs = {}
        for name, env in (("default", {}), ("tile", {"VK_NO_STREAM": "1", "VK_NO_TAIL_FUSION": "1"}), ("tap", {"VK_NO_HALO": "1", "VK_NO_TAIL_FUSION": "1"})):
            for k in ("VK_NO_STREAM", "VK_NO_TAIL_FUSION", "VK_NO_HALO"):
                os.environ.pop(k, None)
            os.environ.update(env)
            m = build()
            with torch.no_grad(), torch.autocast("cuda", dtype=dtype):
                outs[name] = m(x.to(dev)).float().cpu()
        for k in ("VK_NO_STREAM", "VK_NO_TAIL_FUSION", "VK_NO_HALO"):
            os.environ.pop(k, None)
        ey = (yard - lo).abs()
        line = f"n={n} s={s} {str(dtype)[6:]:9s} |logit|max {lo.abs().max():6.2f}  yardstick max {ey.max():.4f} mean {ey.mean():.5f} |"
        for name, o in outs.items():
            e = (o - lo).abs()
            line += f" {name}: max {e.max():.4f} mean {e.mean():.5f} |"
        line += f" default-tile {(outs['default'] - outs['tile']).abs().max():.4f} default-tap {(outs['default'] - outs['tap']).abs().max():.4f}"
        print(line, flush=True)
