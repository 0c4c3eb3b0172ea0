"""Two (or more) ranks on ONE GPU through gloo: the data-parallel invariants of vk.make_data_parallel on the real engine.

    torchrun --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29566 tests/diag/dp_rehearsal.py

Checks on different per-rank batches: (1) after the first backward the reduced gradient buffer of every rank equals — to fp32
round-off — the sum of the per-rank gradients a single process computes by itself (no reducer); (2) after three optimizer
steps every rank holds bit-identical parameters.  (Parameters are not compared with the single process: Adam's first updates
are lr * sign(g), so gradients at round-off level legitimately move a parameter by a whole lr.)
A rehearsal of the control flow and arithmetic, not a measurement (RCCL refuses two ranks on one device, hence gloo)."""
import importlib
import os
import sys
from pathlib import Path

import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
vk = importlib.import_module("vickers-hardness-unet_amd")
from oracle import unet_oracle as O      # synthetic data + seeding only


def build(dev):
    O.set_seed(42)
    m = vk.Unet(encoder_name="resnet34", encoder_weights=None, in_channels=3, classes=1, activation=None).to(dev)
    return m, vk.adamw_for(m, lr=1e-3, weight_decay=1e-4)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    N, S, steps = 2, 64, 3
    model, opt = build(dev)
    vk.make_data_parallel(model, opt)
    model.train()
    g_first = None
    for k in range(steps):
        x, y = O.synthetic_batch(N, S, seed=100 * k + rank)
        opt.zero_grad(set_to_none=True)
        model.loss_and_backward(x.to(dev), y.to(dev), dtype=torch.float32)
        if k == 0:
            model._reducer.finish()          # what opt.step() does first: wait for the bucket all-reduces
            torch.cuda.synchronize()
            g_first = model.flat_grads.detach().float().cpu().clone()      # after the all-reduce: the SUM over ranks (1/world is folded into AdamW)
        opt.step()
    torch.cuda.synchronize()
    flat = model.flat_params.detach().float().cpu()
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    ok = same
    if rank == 0:
        # single-process restatement of the first step's gradients: the per-rank batches one after the other, summed by hand
        ref, _ = build(dev)
        ref.train()
        acc = None
        for r in range(world):
            x, y = O.synthetic_batch(N, S, seed=r)
            ref.flat_grads.zero_()
            ref.loss_and_backward(x.to(dev), y.to(dev), dtype=torch.float32)
            torch.cuda.synchronize()
            g = ref.flat_grads.detach().float().cpu().clone()
            acc = g if acc is None else acc + g
        d = (acc - g_first).abs().max().item()
        scale = acc.abs().max().item()
        print(f"ranks identical after {steps} steps: {same};  first-step gradients: max |reduced - hand-summed| = {d:.3e} "
              f"(max |g| {scale:.3e})", flush=True)
        ok = ok and d <= 1e-4 * scale
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
