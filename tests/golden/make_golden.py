"""Generates the committed golden fixtures.  Run ONLY in the build container (needs /root/reference):

    python tests/golden/make_golden.py

Fixtures written next to this file:

* ``lr_history.json``    — the ``lr`` columns of the reference's own run logs
                           (runs/unet_r34_512/history.json, history_0.json; data, written by
                           reference train.py:656) + lr0 / T_max.  Pins the cosine schedule.
* ``metrics_ref.json``   — outputs of the REFERENCE's own ``dice_coef`` / ``iou_coef``
                           (train.py:230-281) on seeded probability/target tensors.  The reference
                           module is imported with empty stand-ins for the absent third-party
                           modules (cv2, albumentations, segmentation_models_pytorch) purely so that
                           ``import train`` succeeds; only its pure-torch metric functions are called.
* ``manifest.json``      — state-dict key -> shape list + parameter count (G1).
* ``oracle_small.npz``   — oracle logits (train- and eval-mode), loss, four named gradients and a
                           3-step AdamW loss trajectory at N=2, S=64 under seed 42 / data seed 1234
                           (G2-G4).  Detects drift of the oracle itself.

* ``loops_ref.json``     — returns of the REFERENCE's own ``train_one_epoch`` (train.py:381-459) and ``validate``
                           (train.py:461-529) run here on CPU over three seeded (2, 2, 1)-image batches at 64 x 64 with the oracle
                           network, torch AdamW / BCE and the oracle's DiceLoss; per-step loss components, parameter checksums
                           and encoder.bn1 running statistics (G4/G5).  Pins the oracle's loop restatements.

* ``real_masks.npz``     — eight of the reference's own hand-labelled masks (data/masks/*.png: data files, mode L),
                           binarised `> 0` as train.py:163-170 reads them and bit-packed: real indentation shapes
                           (1280x1024 and 3072x2048 micrographs, 0.1 % to 27 % foreground) as inputs of both geometry
                           post-processing paths.  Inputs only: the reference holds no detection list to compare with.

The reference's source never travels: only these data files do.
"""
import json
import sys
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
REF = Path("/root/reference")
sys.path.insert(0, str(ROOT))

from oracle import unet_oracle as O  # noqa: E402


def lr_history():
    out = {}
    for name, lr0 in (("history.json", 5e-5), ("history_0.json", 1e-3)):
        h = json.load(open(REF / "runs" / "unet_r34_512" / name))
        out[name] = {"lr0": lr0, "t_max": len(h), "lr": [r["lr"] for r in h]}
    json.dump(out, open(HERE / "lr_history.json", "w"))


def metrics_ref():
    nthreads = torch.get_num_threads()
    ref_train = _import_reference_train()
    torch.set_num_threads(nthreads)
    cases = []
    for seed, n, s, thr in ((0, 2, 32, 0.5), (1, 3, 64, 0.3), (2, 1, 16, 0.9), (3, 4, 32, 0.0)):
        g = torch.Generator().manual_seed(seed)
        prob = torch.rand(n, 1, s, s, generator=g)
        tgt = (torch.rand(n, 1, s, s, generator=g) > thr).float()
        if seed == 3:
            tgt.zero_()          # empty-target edge case: eps keeps it finite
            prob.mul_(0.4)       # and empty prediction -> Dice = IoU = 1
        cases.append({"seed": seed, "n": n, "s": s, "thr": thr,
                      "dice": ref_train.dice_coef(prob, tgt), "iou": ref_train.iou_coef(prob, tgt)})
    json.dump(cases, open(HERE / "metrics_ref.json", "w"), indent=1)


def _import_reference_train():
    """`import train` of the reference with empty stand-ins for the absent third-party modules (an ordinary
    ModuleNotFoundError otherwise, not a permission denial); only its pure-torch functions are called."""
    for mod in ("cv2", "albumentations", "albumentations.pytorch", "segmentation_models_pytorch"):
        sys.modules.setdefault(mod, types.ModuleType(mod))
    sys.modules["albumentations.pytorch"].ToTensorV2 = object
    if str(REF) not in sys.path:
        sys.path.insert(0, str(REF))
    import train as ref_train  # reference train.py (sets torch.set_num_threads(4) as a side effect)
    return ref_train


class _Recorder(torch.nn.Module):
    """Wraps a loss module and keeps every value it returned (the reference's loops only return epoch means)."""

    def __init__(self, fn):
        super().__init__()
        self.fn, self.values = fn, []

    def forward(self, a, b):
        v = self.fn(a, b)
        self.values.append(float(v.detach()))
        return v


LOOPS_THREADS = 4      # the reference's own setting (train.py:19); the CPU test pins the same count


def loops_batches():
    """The loaders of the loop fixture: what a DataLoader over the reference's dataset yields, (x, y, names) — three training
    batches of 2, 2 and 1 images (a ragged last batch exercises the sample-weighted epoch mean, train.py:452-459) and two
    validation batches — at 64 x 64, seeded."""
    x, y = O.synthetic_batch(5, 64, seed=1234)
    train = [(x[0:2], y[0:2], ["a", "b"]), (x[2:4], y[2:4], ["c", "d"]), (x[4:5], y[4:5], ["e"])]
    xv, yv = O.synthetic_batch(3, 64, seed=4321)
    val = [(xv[0:2], yv[0:2], ["v0", "v1"]), (xv[2:3], yv[2:3], ["v2"])]
    return train, val


def loops_ref():
    """G4/G5: the REFERENCE's own `train_one_epoch` (train.py:381-459) and `validate` (train.py:461-529), executed here on CPU,
    driving the oracle network (the reference's network class lives in the absent smp package) with torch's AdamW / BCE and the
    oracle's DiceLoss: returns, per-step loss components, parameter checksums and BatchNorm running statistics."""
    ref_train = _import_reference_train()
    torch.set_num_threads(LOOPS_THREADS)
    O.set_seed(42)
    model = O.build_model()
    opt = torch.optim.AdamW(model.parameters(), lr=5e-5, weight_decay=1e-4)          # train.py:606
    bce, dice = _Recorder(torch.nn.BCEWithLogitsLoss()), _Recorder(O.DiceLoss(mode="binary"))   # train.py:600-601
    train, val = loops_batches()
    epoch_loss = ref_train.train_one_epoch(model, train, opt, bce, dice, "cpu", scaler=None)
    steps = [{"bce": b, "dice": d} for b, d in zip(bce.values, dice.values)]
    bce.values, dice.values = [], []
    v_loss, v_dice, v_iou = ref_train.validate(model, val, bce, dice, "cpu", out_vis_dir=None)
    named = dict(model.named_parameters())
    out = {
        "threads": LOOPS_THREADS, "torch": torch.__version__,
        "train_one_epoch": epoch_loss, "train_steps": steps,
        "validate": {"loss": v_loss, "dice": v_dice, "iou": v_iou,
                     "steps": [{"bce": b, "dice": d} for b, d in zip(bce.values, dice.values)]},
        "checksums": {k: {"sum": float(named[k].detach().double().sum()), "abs_sum": float(named[k].detach().double().abs().sum())}
                      for k in ("encoder.conv1.weight", "decoder.blocks.4.conv2.0.weight", "segmentation_head.0.bias",
                                "encoder.layer4.2.bn2.weight")},
        "bn1_running_mean": model.encoder.bn1.running_mean.double().tolist(),
        "bn1_running_var": model.encoder.bn1.running_var.double().tolist(),
        "bn1_num_batches_tracked": int(model.encoder.bn1.num_batches_tracked),
    }
    json.dump(out, open(HERE / "loops_ref.json", "w"), indent=1)
    print("loops_ref:", epoch_loss, (v_loss, v_dice, v_iou))


def manifest_and_small():
    O.set_seed(42)
    model = O.build_model()
    man = O.state_dict_manifest(model)
    json.dump({"param_count": sum(p.numel() for p in model.parameters()), "entries": man},
              open(HERE / "manifest.json", "w"))

    x, y = O.synthetic_batch(2, 64, seed=1234)
    model.eval()
    with torch.no_grad():
        logits_eval = model(x).numpy()
    model.train()
    logits = model(x)
    loss = O.total_loss(logits, y)
    loss.backward()
    named = dict(model.named_parameters())
    grads = {k: named[k].grad.numpy().copy() for k in (
        "encoder.conv1.weight", "encoder.layer3.0.downsample.0.weight",
        "decoder.blocks.3.conv1.0.weight", "segmentation_head.0.bias")}
    # fresh model for the trajectory so BN running stats start from init
    O.set_seed(42)
    model2 = O.build_model()
    opt = torch.optim.AdamW(model2.parameters(), lr=5e-5, weight_decay=1e-4)
    traj = O.train_steps(model2, opt, [(x, y)] * 3)
    np.savez_compressed(
        HERE / "oracle_small.npz",
        logits_eval=logits_eval, logits_train=logits.detach().numpy(), loss=np.float64(loss.item()),
        traj=np.array(traj, dtype=np.float64),
        bn1_running_mean=model2.encoder.bn1.running_mean.numpy(),
        **{"grad::" + k: v for k, v in grads.items()})


REAL_MASKS = ("1.png", "4 (2).png", "image001 (4).png", "image003 (15).png", "image005 (3).png", "image011_(7)_dual.png")


def real_masks():
    """Bit-packed copies of a few label masks of the reference's dataset (largest / smallest foreground, both micrograph sizes)."""
    from PIL import Image

    files = sorted((REF / "data" / "masks").glob("*.png"))
    by_name = {f.name: f for f in files}
    pick = [by_name[n] for n in REAL_MASKS if n in by_name]
    big = [f for f in files if Image.open(f).size == (3072, 2048)]
    stats = sorted(((float((np.array(Image.open(f)) > 0).mean()), f) for f in big), key=lambda t: t[0])
    pick += [stats[0][1], stats[len(stats) // 2][1], stats[-1][1]]          # 3072x2048: least / median / most foreground
    out = {}
    names = []
    for i, f in enumerate(dict.fromkeys(pick)):
        m = np.array(Image.open(f).convert("L")) > 0
        out[f"bits_{i}"] = np.packbits(m, axis=1)
        out[f"shape_{i}"] = np.array(m.shape, dtype=np.int32)
        names.append(f.name)
    out["names"] = np.array(names)
    np.savez_compressed(HERE / "real_masks.npz", **out)
    print("real masks:", names, {k: v.shape for k, v in out.items() if k.startswith("shape")})


if __name__ == "__main__":
    real_masks()
    lr_history()
    metrics_ref()
    manifest_and_small()
    loops_ref()
    print("golden fixtures written to", HERE)
