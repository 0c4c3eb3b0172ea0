"""Layer-by-layer comparison of the HIP engine against the oracle (GPU box only; debugging aid).

    python tests/diag/debug_layers.py [--n 2] [--size 64] [--dtype f32|bf16] [--train]
prints max|err| / max|ref| for every raw conv output z, every block output and (with --train) every
gradient buffer, so the first diverging kernel is obvious."""
import argparse
import importlib
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
vk = importlib.import_module("vickers-hardness-unet_amd")
from oracle import unet_oracle as O  # noqa: E402


def nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--train", action="store_true")
    a = ap.parse_args()
    dt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[a.dtype]
    dev = torch.device("cuda:0")
    O.set_seed(42)
    ref = O.build_model()
    O.set_seed(42)
    model = vk.Unet(encoder_weights=None, compute_dtype=dt).to(dev)
    x, y = O.synthetic_batch(a.n, a.size, seed=1234)
    ref.train(a.train)
    model.train(a.train)
    acts, gacts = {}, {}

    def hook(name):
        def f(mod, inp, out):
            acts[name] = out.detach()
            if a.train and out.requires_grad:
                out.register_hook(lambda g, name=name: gacts.__setitem__(name, g.detach()))
        return f

    for name, mod in ref.named_modules():
        if isinstance(mod, torch.nn.Conv2d) or isinstance(mod, O.BasicBlock):
            mod.register_forward_hook(hook(name))
    lo = ref(x)
    if a.train:
        loss_o = O.total_loss(lo, y)
        loss_o.backward()
    xd, yd = x.to(dev), y.to(dev)
    if a.train:
        out3 = model.loss_and_backward(xd, yd)
        lg = model.last_logits
        print("loss gpu/oracle", out3.tolist(), loss_o.item())
    else:
        with torch.no_grad():
            lg = model(xd)
    torch.cuda.synchronize()
    plan = next(iter(model._plans.values()))

    def report(tag, got, want):
        err = (got - want).abs().max().item()
        mx = want.abs().max().item()
        flag = "" if err <= (1e-3 if dt == torch.float32 else 5e-2) * (mx + 1e-6) else "   <<<<<"
        print(f"{tag:58s} err {err:10.3e}  ref {mx:10.3e}{flag}")

    for name, t in acts.items():
        if name == "segmentation_head.0":
            continue
        key = ("out:" if isinstance(dict(ref.named_modules())[name], O.BasicBlock) else "z:") + name
        try:
            got = nchw(plan.debug_tensor(key))
        except Exception as e:
            print("skip", key, e)
            continue
        report(key, got, t)
    report("logits", lg.cpu(), lo.detach())
    if a.train:
        for name, g in gacts.items():
            if name == "segmentation_head.0":
                continue
            isblk = isinstance(dict(ref.named_modules())[name], O.BasicBlock)
            key = ("gout:" if isblk else "g:") + name
            try:
                got = nchw(plan.debug_tensor(key))
            except Exception as e:
                print("skip", key, e)
                continue
            report(key + (" (dz)" if not isblk else ""), got, g)
        named = dict(ref.named_parameters())
        sd = {k: v for k, v in model.named_parameters()}
        for k, p in named.items():
            report("grad:" + k, sd[k].grad.cpu(), p.grad)


if __name__ == "__main__":
    main()
