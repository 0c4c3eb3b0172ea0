"""Diagnostic: the three-step fp16 + GradScaler trajectory of tests/test_model_gpu.py::test_amp_fp16_gradscaler_trajectory_vs_oracle
under different kernel selections (environment switches of DESIGN.md section 14), to tell a kernel defect from the run-to-run
sensitivity of three Adam steps (updates of lr * sign(g) for the many near-zero gradients).

    python tests/diag/amp_traj.py            # prints the engine's losses next to the fp32 and the fp16-autocast oracle
"""
import importlib
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

CHILD = r'''
import importlib, sys, torch
sys.path.insert(0, %r)
vk = importlib.import_module("vickers-hardness-unet_amd")
from oracle import unet_oracle as O
x, y = O.synthetic_batch(6, 64, seed=21)
batches = [(x[i:i + 2], y[i:i + 2]) for i in (0, 2, 4)]
O.set_seed(42); model = vk.Unet(encoder_weights=None).to("cuda:0")
opt = vk.adamw_for(model, lr=5e-5, weight_decay=1e-4)
scaler = vk.GradScaler("cuda", enabled=True)
lg = [O.train_one_epoch(model, [(xb, yb, ["a", "b"])], opt, torch.nn.BCEWithLogitsLoss(), vk.DiceLoss(mode="binary"), "cuda", scaler) for xb, yb in batches]
print("ENGINE", " ".join(f"{v:.6f}" for v in lg))
'''


def main():
    import torch
    sys.path.insert(0, str(ROOT / "tests"))
    from oracle import unet_oracle as O
    tm = importlib.import_module("test_model_gpu")
    x, y = O.synthetic_batch(6, 64, seed=21)
    batches = [(x[i:i + 2], y[i:i + 2]) for i in (0, 2, 4)]
    _, l32, _ = tm._oracle_amp_steps(O, batches, autocast=False)
    _, l16, _ = tm._oracle_amp_steps(O, batches, autocast=True)
    print("oracle fp32          ", " ".join(f"{v:.6f}" for v in l32))
    print("oracle fp16 autocast ", " ".join(f"{v:.6f}" for v in l16))
    for name, env in [("default", {}), ("VK_NO_STEM_TILE", {"VK_NO_STEM_TILE": "1"}), ("VK_NO_S2_TILE", {"VK_NO_S2_TILE": "1"}),
                      ("both", {"VK_NO_STEM_TILE": "1", "VK_NO_S2_TILE": "1"}), ("VK_COL_PIPE=0", {"VK_COL_PIPE": "0"}),
                      ("VK_NO_WGRAD_HALO", {"VK_NO_WGRAD_HALO": "1"}), ("VK_NO_HALO", {"VK_NO_HALO": "1"})]:
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, "-c", CHILD % str(ROOT)], env=e, capture_output=True, text=True).stdout
        line = [l for l in out.splitlines() if l.startswith("ENGINE")]
        print(f"{name:21s}", line[0][7:] if line else "FAILED")


if __name__ == "__main__":
    main()
