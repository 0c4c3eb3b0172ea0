"""Stand-alone timing of the convolution kernels at the U-Net's layer shapes (N=32, S=512 geometry).

    python tools/microbench.py [--only L3] [--ops fwd,dgrad,wgrad] [--reps 20] [--dtype bf16]
Prints per (layer, op): microseconds per launch and algorithmic TFLOP/s (fraction of 2.5 PF)."""
import argparse
import ctypes as C
import importlib
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
vk = importlib.import_module("vickers-hardness-unet_amd")
L_ = vk._lib

# name: (H, [(C, up)], K)
LAYERS = {
    "L1": (128, [(64, 0)], 64), "L2": (64, [(128, 0)], 128), "L3": (32, [(256, 0)], 256), "L4": (16, [(512, 0)], 512),
    "D0c1": (32, [(512, 1), (256, 0)], 256), "D1c1": (64, [(256, 1), (128, 0)], 128), "D2c1": (128, [(128, 1), (64, 0)], 64),
    "D3c1": (256, [(64, 1), (64, 0)], 32), "D3c2": (256, [(32, 0)], 32), "D4c1": (512, [(32, 1)], 16), "D4c2": (512, [(16, 0)], 16),
    # the shapes of the two concat data gradients with a one- / two-chunk reduction (forward op of the same kernel class)
    "D3c1g": (256, [(32, 0)], 128), "D2c1g": (128, [(64, 0)], 192),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--ops", default="fwd,dgrad,wgrad")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--prof", action="store_true")
    ap.add_argument("--no-affine", action="store_true", help="sources without the BN+ReLU prologue (materialised activations)")
    ap.add_argument("--no-stats", action="store_true", help="forward without the BatchNorm partial sums")
    ap.add_argument("--ab", default="", help="A/B in ONE process, interleaved rounds: NAME=v1,v2[,v3] sets the environment variable NAME "
                                             "before each round of launches (the library reads its switches per launch)")
    ap.add_argument("--rounds", type=int, default=5)
    a = ap.parse_args()
    dt = {"bf16": torch.bfloat16, "f32": torch.float32, "f16": torch.float16}[a.dtype]
    dev = torch.device("cuda:0")
    lib = vk.lib()
    st = torch.cuda.current_stream().cuda_stream
    N = a.n
    for name, (H, srcs, K) in LAYERS.items():
        if a.only and name not in a.only.split(","):
            continue
        Ctot = sum(c for c, _ in srcs)
        ts, ss = [], []
        for c, up in srcs:
            t = torch.randn(N, H >> up, H >> up, c, device=dev).to(dt)
            sc = torch.rand(c, device=dev) + 0.5
            sh = torch.randn(c, device=dev) * 0.1
            ts.append((t, sc, sh))
            if a.no_affine:
                ss.append(L_.vk_src(t.data_ptr(), c, up, None, None, 0))
            else:
                ss.append(L_.vk_src(t.data_ptr(), c, up, sc.data_ptr(), sh.data_ptr(), 1))
        s1 = ss[1] if len(ss) > 1 else L_.vk_src(None, 0, 0, None, None, 0)
        w = (torch.randn(K, 3, 3, Ctot, device=dev) * 0.05).to(dt)
        wt = (torch.randn(Ctot, 3, 3, K, device=dev) * 0.05).to(dt)
        y = torch.empty(N, H, H, K, device=dev, dtype=dt)
        dz = torch.randn(N, H, H, K, device=dev).to(dt)
        dx = torch.empty(N, H, H, Ctot, device=dev, dtype=dt)
        dw = torch.zeros(K, 3, 3, Ctot, device=dev)
        stats = torch.zeros(32 * 2 * K, dtype=torch.float64, device=dev)
        wsl = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
        d_f = L_.vk_conv_desc(L_.dtype_code(dt), N, H, H, H, H, K, 3, 3, 1, 1, 0, ss[0], s1)
        d_d = L_.vk_conv_desc(L_.dtype_code(dt), N, H, H, H, H, Ctot, 3, 3, 1, 1, 1,
                              L_.vk_src(dz.data_ptr(), K, 0, None, None, 0), L_.vk_src(None, 0, 0, None, None, 0))
        flops = 2.0 * N * H * H * K * 9 * Ctot
        eb = 4 if dt == torch.float32 else 2
        act_bytes = (sum(t.numel() for t, _, _ in ts) + N * H * H * K) * eb          # operands read / written once

        def weights(d, plain, rows, red):       # halo pack when the descriptor runs on the 3x3 tile kernels
            if not lib.vk_conv_uses_halo_pack(C.byref(d)):
                return plain, lib.vk_conv_fwd
            pk = torch.empty_like(plain)
            L_.check(lib.vk_halo_pack(L_.dtype_code(dt), rows, red, plain.data_ptr(), pk.data_ptr(), st))
            return pk, lib.vk_conv_fwd_packed
        wf, fn_f = weights(d_f, w, K, Ctot)
        wd_, fn_d = weights(d_d, wt, Ctot, K)
        zprev = torch.randn(N, H, H, Ctot, device=dev).to(dt)          # raw conv output the data gradient is masked by
        bsc, bsh = torch.rand(Ctot, device=dev) + 0.5, torch.randn(Ctot, device=dev) * 0.1
        bsums = torch.zeros(32 * 2 * Ctot, dtype=torch.float64, device=dev)
        bnr = L_.vk_bnr(zprev.data_ptr(), bsc.data_ptr(), bsh.data_ptr(), bsums.data_ptr())
        ops = {
            "fwd": lambda: fn_f(C.byref(d_f), wf.data_ptr(), y.data_ptr(), None, 0, 0, None if a.no_stats else stats.data_ptr(), st),
            "dgrad": lambda: fn_d(C.byref(d_d), wd_.data_ptr(), dx.data_ptr(), None, 0, 0, None, st),
            "dgrad_acc": lambda: fn_d(C.byref(d_d), wd_.data_ptr(), dx.data_ptr(), None, 0, 1, None, st),
            "dgrad_bnr": lambda: lib.vk_conv_dgrad_fused(C.byref(d_d), wd_.data_ptr(), dx.data_ptr(), None, 0, 0, C.byref(bnr), st),
            "wgrad": lambda: lib.vk_conv_wgrad(C.byref(d_f), dz.data_ptr(), dw.data_ptr(), wsl.data_ptr(), wsl.numel(), st),
        }
        for op in a.ops.split(","):
            f = ops[op]
            if a.ab:
                import os
                import statistics
                var, vals = a.ab.split("=")
                vals = vals.split(",")
                res = {x: [] for x in vals}
                for rnd_ in range(a.rounds + 1):
                    for x in vals:
                        if x == "":
                            os.environ.pop(var, None)
                        else:
                            os.environ[var] = x
                        for _ in range(2):
                            L_.check(f())
                        torch.cuda.synchronize()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(a.reps):
                            f()
                        e1.record()
                        torch.cuda.synchronize()
                        if rnd_:
                            res[x].append(e0.elapsed_time(e1) * 1e3 / a.reps)
                txt = "  ".join(f"{var}={x}: median {statistics.median(r):8.1f} min {min(r):8.1f} us ({flops / statistics.median(r) / 1e6 / 2500 * 100:5.1f} %)" for x, r in res.items())
                print(f"{name:5s} {op:9s} H{H:<4d} C{Ctot:<4d} K{K:<4d} {txt}", flush=True)
                os.environ.pop(var, None)
                continue
            for _ in range(3):
                L_.check(f())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                f()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.reps
            if a.prof:
                lib.vk_prof_enable(1)
                for _ in range(5):
                    f()
                torch.cuda.synchronize()
                lib.vk_prof_enable(0)
                for k, v in L_.prof_collect().items():
                    print(f"        {k:40s} {v['ms'] / v['n'] * 1e3:9.1f} us/launch")
            print(f"{name:5s} {op:6s} H{H:<4d} C{Ctot:<4d} K{K:<4d} {us:9.1f} us  {flops / us / 1e6:8.1f} TF  ({flops / us / 1e6 / 2500 * 100:5.1f} % of MFMA peak)  {act_bytes / us / 1e3:7.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
