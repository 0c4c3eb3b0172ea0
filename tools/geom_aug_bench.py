"""Timing of the two SURVEY 8(f) rank 3/4 device components with inputs resident in HBM (HIP events on the launch stream):

    geometry   vk_geom_minarearect (threshold -> open/close -> 8-connected components -> hull -> min-area rectangle -> diagonals)
    augment    vk_augment_batch (flip / rot90 / rotate / brightness-contrast / CLAHE / blur / noise / normalise; one fused pass + a tile-histogram pass for CLAHE samples)

    python tools/geom_aug_bench.py            (run through rocprofv3 --kernel-trace --stats for per-kernel numbers)"""
import importlib
import math
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
vk = importlib.import_module("vickers-hardness-unet_amd")
dev = torch.device("cuda:0")


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def prob_maps(B, h, w, seed=0, n=3):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    out = np.empty((B, h, w), dtype=np.float32)
    for b in range(B):
        p = np.full((h, w), 0.05, dtype=np.float32)
        for _ in range(n):
            cx, cy, half, th = rng.uniform(0.2, 0.8) * w, rng.uniform(0.2, 0.8) * h, rng.uniform(0.05, 0.15) * min(h, w), rng.uniform(0, math.pi / 2)
            u = (xx - cx) * math.cos(th) + (yy - cy) * math.sin(th)
            v = -(xx - cx) * math.sin(th) + (yy - cy) * math.cos(th)
            p = np.maximum(p, 1.0 / (1.0 + np.exp((np.maximum(np.abs(u), np.abs(v)) - half) / 1.5)))
        out[b] = np.clip(p + rng.normal(scale=0.05, size=p.shape), 0, 1)
    return out


def main():
    import ctypes as C
    L = vk._lib
    lib = vk.lib()
    print("== geometry post-processing (ui_infer_rectangle.py:291-381), device time per call incl. all 14 launches, results left on the device")
    for B, h, w in ((1, 512, 512), (32, 512, 512), (1, 2048, 3072), (8, 2048, 3072)):
        prob = torch.from_numpy(prob_maps(min(B, 4), h, w)).to(dev)
        prob = prob.repeat((B + prob.shape[0] - 1) // prob.shape[0], 1, 1)[:B].contiguous()
        desc = L.vk_geom_desc(h, w, 0.5, 3, 1, 1, max(200, int(0.0008 * h * w)), 64)
        nbytes = lib.vk_geom_workspace_bytes(C.byref(desc), B)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        clean = torch.empty(B, h, w, dtype=torch.uint8, device=dev)
        dets = torch.zeros(B * 64 * C.sizeof(L.vk_geom_det), dtype=torch.uint8, device=dev)
        counts = torch.zeros(B, dtype=torch.int32, device=dev)
        st = torch.cuda.current_stream().cuda_stream

        def run():
            L.check(lib.vk_geom_minarearect(C.byref(desc), B, prob.data_ptr(), clean.data_ptr(), dets.data_ptr(), counts.data_ptr(), ws.data_ptr(), nbytes, st))
        us = timed(run)
        t0 = time.perf_counter()
        _, det = vk.postprocess_minarearect_batch(prob)
        torch.cuda.synchronize()
        host_ms = (time.perf_counter() - t0) * 1e3
        mb = B * h * w * 5 / 1e6
        print(f"  B={B:3d} {h}x{w}: {us:9.1f} us = {us / B:8.1f} us/map ({mb / us * 1e3:7.1f} GB/s of 5 algorithmic B/px); "
              f"host mirror incl. allocation + read-back {host_ms:7.2f} ms; components kept: {[len(d) for d in det][:4]}")
    print("== 4-vertex fit (ui_infer_quadrilateral.py:262-530): steps 1-3 + one wave per component (dilation, border following, hull, "
          "approxPolyDP bisection, ranking)")
    for B, h, w in ((1, 512, 512), (32, 512, 512), (1, 2048, 3072), (8, 2048, 3072)):
        prob = torch.from_numpy(prob_maps(min(B, 4), h, w)).to(dev)
        prob = prob.repeat((B + prob.shape[0] - 1) // prob.shape[0], 1, 1)[:B].contiguous()
        desc = L.vk_geom_desc(h, w, 0.45, 3, 1, 1, max(200, int(0.0008 * h * w)), 64)
        nbytes = lib.vk_geom_workspace_bytes(C.byref(desc), B)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        clean = torch.empty(B, h, w, dtype=torch.uint8, device=dev)
        dets = torch.zeros(B * 64 * C.sizeof(L.vk_geom_quad), dtype=torch.uint8, device=dev)
        counts = torch.zeros(B, dtype=torch.int32, device=dev)
        st = torch.cuda.current_stream().cuda_stream

        def runq():
            L.check(lib.vk_geom_quadrilateral(C.byref(desc), 2, B, prob.data_ptr(), clean.data_ptr(), dets.data_ptr(), counts.data_ptr(), ws.data_ptr(), nbytes, st))
        us = timed(runq)
        _, det = vk.postprocess_quadrilateral_batch(prob)
        pts = [d["contour_points"] for ds_ in det for d in ds_]
        print(f"  B={B:3d} {h}x{w}: {us:9.1f} us = {us / B:8.1f} us/map; components {sum(len(d) for d in det)}, border points per component "
              f"{min(pts) if pts else 0}-{max(pts) if pts else 0}, branches {sorted({d['branch'] for ds_ in det for d in ds_})}")
    print("== augmentation (train.py:67-113), one fused launch per batch")
    rng = np.random.default_rng(1)
    imgs = [rng.integers(0, 256, (1024, 1280, 3), dtype=np.uint8) for _ in range(4)]
    masks = [(rng.random((1024, 1280)) > 0.9).astype(np.uint8) * 255 for _ in range(4)]
    t0 = time.perf_counter()
    ds = vk.DeviceDataset(imgs, masks, img_size=512, device=dev)
    torch.cuda.synchronize()
    print(f"  dataset upload + letterbox of 4 images 1280x1024: {(time.perf_counter() - t0) * 1e3:.1f} ms (one-off)")
    sm = vk.AugmentSampler(seed=3)
    for n in (8, 32, 128):
        idx = [i % 4 for i in range(n)]
        draws = [sm.sample() for _ in range(n)]
        arr = vk.augment._params_array(draws, 512)
        ws = torch.empty(int(lib.vk_augment_workspace_bytes(n, 512)), dtype=torch.uint8, device=dev)
        n_clahe = sum(1 for d in draws if d["photo"] == 2)
        index = torch.tensor(idx, dtype=torch.int32, device=dev)
        x = torch.empty(n, 3, 512, 512, device=dev)
        y = torch.empty(n, 1, 512, 512, device=dev)
        pdev = torch.empty(n * C.sizeof(L.vk_aug_params), dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream().cuda_stream

        def run():
            L.check(lib.vk_augment_batch(n, 512, 4, ds.images.data_ptr(), ds.masks.data_ptr(), index.data_ptr(), arr, pdev.data_ptr(), ds._tables.data_ptr(), ws.data_ptr(), ws.numel(), x.data_ptr(), y.data_ptr(), st))
        us = timed(run)
        t0 = time.perf_counter()
        for _ in range(10):
            ds.batch(idx, sm)
        torch.cuda.synchronize()
        host_us = (time.perf_counter() - t0) / 10 * 1e6
        by = n * 512 * 512 * 20.0
        print(f"  n={n:3d} ({n_clahe} CLAHE): {us:8.1f} us/batch = {n / us * 1e6:10.0f} img/s ({by / us * 1e-3:6.0f} GB/s of 20 algorithmic B/px: 4 read + 16 written); "
              f"through DeviceDataset.batch incl. host sampling {host_us:8.1f} us/batch")


if __name__ == "__main__":
    main()
