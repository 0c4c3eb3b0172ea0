"""Timing of the pre/post-processing kernels on one MI355X, inputs resident in HBM:

    python tests/diag/prepost_bench.py [--reps 200] [--no-cpu]

Per case: microseconds per launch (HIP events on the launch stream), algorithmic bytes (source bytes the gather can touch +
destination bytes) / time against the 8 TB/s HBM roofline, the PCIe-inclusive time of the pre-processing call (pageable
uint8 upload + kernel) and the CPU restatement (oracle/prepost_oracle.py, numpy, one thread of work) beside it.
Ends with the batched wrapper: 16 images -> Segmenter.infer_batch (pre + fp32 forward + 16 post)."""
import argparse
import importlib
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
vk = importlib.import_module("vickers-hardness-unet_amd")
from oracle import prepost_oracle as P      # CPU leg only

CASES = [(1200, 1600, "centered"), (1200, 1600, "pad_br"), (2048, 2048, "centered"), (512, 512, "centered"), (300, 400, "pad_br")]


def timed(fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    S = 512
    rows = []
    for h, w, conv in CASES:
        rng = np.random.default_rng(h + w)
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        src = torch.from_numpy(img).to(dev)
        geo = vk.prepost.letterbox_geometry(h, w, S, conv)
        _, nh, nw, top, left = geo
        meta = (geo[0], geo, (h, w))
        x = torch.empty(3, S, S, device=dev)
        lg = torch.randn(S, S, device=dev) * 3
        same = (nh, nw) == (h, w)
        pre_b = (3 * h * w if same else min(3 * h * w, 12 * nh * nw)) + 12 * S * S
        mask_b = min(4 * nh * nw, 4 * h * w) + h * w
        prob_b = 4 * nh * nw + 4 * h * w
        t_pre = timed(lambda: vk.prepost.preprocess(src, S, conv, dev, out=x), a.reps)
        t_mask = timed(lambda: vk.prepost.postprocess_mask(lg, meta), a.reps)
        t_prob = timed(lambda: vk.prepost.postprocess_prob(lg, meta), a.reps)
        # PCIe-inclusive: pageable host image -> device -> kernel, synchronised per image as a GUI would
        t0 = time.perf_counter()
        for _ in range(20):
            vk.prepost.preprocess(img, S, conv, dev, out=x)
            torch.cuda.synchronize()
        t_pcie = (time.perf_counter() - t0) / 20 * 1e6
        row = dict(h=h, w=w, convention=conv, resized=[nh, nw],
                   pre_us=round(t_pre, 2), pre_GBps=round(pre_b / t_pre / 1e3, 1), pre_pcie_inclusive_us=round(t_pcie, 1),
                   mask_us=round(t_mask, 2), mask_GBps=round(mask_b / t_mask / 1e3, 1),
                   prob_us=round(t_prob, 2), prob_GBps=round(prob_b / t_prob / 1e3, 1))
        if not a.no_cpu:
            lgc = lg.cpu().numpy()
            t0 = time.perf_counter(); P.preprocess(img, S, conv); row["cpu_pre_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
            t0 = time.perf_counter(); P.postprocess_mask(lgc, nh, nw, top, left, (h, w)); row["cpu_mask_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
            t0 = time.perf_counter(); P.postprocess_prob(lgc, nh, nw, top, left, (h, w)); row["cpu_prob_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
        rows.append(row)
        print(json.dumps(row), flush=True)
    # the batched wrapper around the fp32 eval forward (BASELINE configs[1] geometry: 16 images of 512x512 network input)
    model = vk.Unet(encoder_name="resnet34", encoder_weights=None, in_channels=3, classes=1, activation=None).to(dev).eval()
    seg = vk.prepost.Segmenter(model, 512, dev)
    rng = np.random.default_rng(0)
    imgs = [torch.from_numpy(rng.integers(0, 256, (1200, 1600, 3), dtype=np.uint8)).to(dev) for _ in range(16)]

    def batch():
        xb, metas = vk.prepost.preprocess_batch(imgs, 512, "centered", dev)
        with torch.no_grad():
            lo = model(xb)
        return [vk.prepost.postprocess_prob(lo[i, 0], m) for i, m in enumerate(metas)]

    def fwd_only():
        with torch.no_grad():
            return model(xb0)

    xb0, _ = vk.prepost.preprocess_batch(imgs, 512, "centered", dev)
    t_all = timed(batch, 20)
    t_fwd = timed(fwd_only, 20)
    print(json.dumps(dict(case="16 x 1200x1600 BGR (resident) -> probability maps, fp32 forward", total_ms=round(t_all / 1e3, 3),
                          forward_only_ms=round(t_fwd / 1e3, 3), pre_post_share=round(1 - t_fwd / t_all, 4),
                          images_per_s=round(16 / (t_all / 1e6), 1))), flush=True)
    # the GUI's own unit of work: ONE pageable host image -> probability map on the host (upload, pre, forward bs 1, post, download)
    host_img = rng.integers(0, 256, (1200, 1600, 3), dtype=np.uint8)
    for dt_name, dt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        model.compute_dtype = dt
        for _ in range(3):
            seg.infer(host_img)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            seg.infer(host_img)
        t1 = (time.perf_counter() - t0) / 20 * 1e3
        print(json.dumps(dict(case=f"Segmenter.infer, one 1200x1600 host image, {dt_name} forward, host to host", ms=round(t1, 3))), flush=True)


if __name__ == "__main__":
    main()
