#!/usr/bin/env python3
"""bench.py — images/sec of one full training step (forward + BCE/Dice + backward + AdamW
[+ gradient all-reduce]) of the ResNet-34 U-Net on synthetic 512x512 batches, bs=32 per GPU, bf16
(BASELINE.json configs[2]; configs[3] when launched under torchrun with N ranks).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 32] [--size 512] [--dtype bf16]
                    [--mode train|infer] [--no-cpu-baseline]

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     : the dominant kernel family (most GPU time in the profiled steps), its algorithmic
                 FLOP/s (or B/s) from live HIP-event timing on the launch stream vs the gfx950 peak
  cpu_baseline : the oracle (torch CPU restatement of the reference path) timed on this host's cores
                 on a bounded sample (rank 0, N=1 only) — a reported baseline, never the target.
and, beside `ms_per_step` (the fused step model.loss_and_backward + optimizer.step):
  api_path_ms_per_step   : the drop-in path INTEGRATION.md describes — model(x) under torch.autocast, torch's BCEWithLogitsLoss +
                           vk.DiceLoss, loss.backward() through autograd, optimizer.step() (GradScaler in fp16 mode)
  sync_each_step_ms      : the same with `loss.item()` after every step, as the reference loop does (train.py:452)
"""
import argparse
import importlib
import json
import os
import sys
import time
from pathlib import Path

# multi-process GPU work on this pool needs dmabuf IPC (RCCL fails with `hipIpcGetMemHandle: invalid argument` otherwise);
# the launcher's environment normally carries it already — must be set before HIP initialises
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

_T0 = time.perf_counter()
ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_MFMA_16B = 2.5e15     # dense bf16/fp16 MFMA, MI355X_MICROARCH.md
PEAK_MFMA_F32 = 157.3e12
PEAK_HBM = 8.0e12

FWD_GFLOP_PER_IMG_512 = 62.512
TRAIN_GFLOP_PER_IMG_512 = 186.30


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--mode", default="train", choices=["train", "infer"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-variants", action="store_true", help="skip the 4-thread / bs 8 CPU baseline variant")
    ap.add_argument("--prof-steps", type=int, default=3)
    ap.add_argument("--api-steps", type=int, default=10, help="timed steps of the drop-in API path (0 = skip)")
    args = ap.parse_args()

    # --gpus N without a launcher: start the N ranks ourselves (a child process, never an exec: nothing has touched the GPU yet, and
    # the child's ranks initialise it themselves), relay their output and exit with the launcher's code
    action, info = launch_plan(args.gpus, os.environ)
    if action == "spawn":
        raise SystemExit(self_launch(info, sys.argv[1:]))
    if action == "error":
        raise SystemExit(info)
    rank = int(os.environ.get("RANK", "0"))
    world = info
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    # rehearsal of the multi-rank control flow on a ONE-GPU box: VK_BENCH_REHEARSAL=1 puts every rank on cuda:0 and uses gloo
    # (RCCL refuses two ranks on one device).  Never set for a measurement.
    rehearsal = bool(os.environ.get("VK_BENCH_REHEARSAL"))
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    import torch.distributed as dist
    force_dist = bool(os.environ.get("VK_BENCH_FORCE_DIST"))      # exercise the RCCL path with a single rank (testing)
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    vk = importlib.import_module("vickers-hardness-unet_amd")

    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    N, S = args.batch, args.size

    vk.seed_everything(42)
    model = vk.Unet(encoder_name="resnet34", encoder_weights=None, in_channels=3, classes=1, activation=None).to(dev)
    opt = vk.adamw_for(model, lr=5e-5, weight_decay=1e-4)
    if world > 1 or force_dist:
        vk.make_data_parallel(model, opt, force=force_dist)
    x, y = vk.synthetic_batch(N, S, seed=1234 + rank)      # rank r owns images [r*N, (r+1)*N)
    x, y = x.to(dev), y.to(dev)
    _log(f"rank {rank}/{world}: model + data resident on {torch.cuda.get_device_name(dev)}")

    if args.mode == "train":
        model.train()

        # fp16 (configs[4]): the reference's loss scale (GradScaler default 2^16, train.py:610-611) — without it the 1024x1024
        # gradients underflow in fp16; folded into the loss-gradient kernel and, as a device scalar, into the AdamW kernel
        loss_scale = 65536.0 if dtype == torch.float16 else 1.0
        scale_t = torch.full((1,), loss_scale, device=dev) if dtype == torch.float16 else None

        def step():
            opt.zero_grad(set_to_none=True)
            out = model.loss_and_backward(x, y, grad_scale=loss_scale, dtype=dtype)
            opt.step(grad_scale=scale_t)
            return out
    else:
        model.eval()
        model.compute_dtype = dtype

        def step():
            with torch.no_grad():
                return model(x)

    def barrier():
        if world > 1 or force_dist:
            dist.barrier() if rehearsal else dist.barrier(device_ids=[local_rank])

    for i in range(args.warmup):
        out = step()
        if i == 0:
            torch.cuda.synchronize()
            _log("first step done")
    torch.cuda.synchronize()
    _log("warm-up done")
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _log(f"timed region: {dt / args.steps * 1e3:.2f} ms/step")
    if world > 1 or force_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    last = [float(v) for v in out.flatten()[:3].tolist()] if args.mode == "train" else None

    # ---- the same step through the drop-in API, and with the reference's per-step host sync
    api_ms = sync_ms = None
    if args.mode == "train" and args.api_steps > 0:
        bce, dice = torch.nn.BCEWithLogitsLoss(), vk.DiceLoss(mode="binary")
        scaler = vk.GradScaler("cuda", enabled=(dtype == torch.float16))

        def step_api(sync):
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=dtype, enabled=(dtype != torch.float32)):
                logits = model(x)
                loss = bce(logits, y) + dice(logits, y)
            if dtype == torch.float16:                      # train.py:441-445
                scaler.scale(loss).backward()
                scaler.step(opt)
                scaler.update()
            else:                                           # train.py:448-449
                loss.backward()
                opt.step()
            return loss.item() if sync else loss            # train.py:452

        def timed(sync):
            for _ in range(2):
                step_api(sync)
            torch.cuda.synchronize()
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.api_steps):
                step_api(sync)
            torch.cuda.synchronize()
            barrier()
            d = time.perf_counter() - t1
            if world > 1 or force_dist:
                tt = torch.tensor([d], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                d = tt.item()
            return d / args.api_steps * 1e3

        api_ms = timed(False)
        sync_ms = timed(True)
        _log(f"drop-in API path: {api_ms:.2f} ms/step; with loss.item() every step: {sync_ms:.2f} ms/step")

    # ---- per-kernel-family timing with HIP events on the launch stream (a few extra, untimed-for-value steps)
    roof = None
    table = {}
    # EVERY rank runs these steps (they contain the gradient all-reduces: rank 0 alone would wait for its peers for ever);
    # only rank 0 brackets its launches with events and reports
    L = vk.lib()
    dp_info = None
    red = getattr(model, "_reducer", None)
    if red is not None and args.mode == "train":
        red.timing = True
    if rank == 0:
        L.vk_prof_enable(1)
    for _ in range(args.prof_steps):
        step()
    torch.cuda.synchronize()
    if red is not None and args.mode == "train":
        red.timing = False
        tails = red.exposed_tail_ms()
        plan = next(iter(model._plans.values()))
        dp_info = {"policy": red.policy, "reserved_cus": red.reserved_cus,
                   "buckets_mb_fp32": [round((b1 - b0) * 4 / 1e6, 2) for b0, b1 in plan.buckets],
                   "exposed_allreduce_tail_ms": [round(t, 3) for t in tails],
                   "note": "tail = time the compute stream waits for outstanding bucket all-reduces after its last backward "
                           "kernel and before AdamW (HIP events on the compute stream, rank 0, profile steps)"}
    if rank == 0:
        L.vk_prof_enable(0)
        table = vk._lib.prof_collect()
    # ---- overlap budget of the data-parallel policies: the backward issued in the "deferred" grouping — stages 0-8 as one call (one
    # weight-gradient batch), stage 9 (layer 1 + stem) as the second — with a HIP event pair around each call.  The collective that
    # carries 99 % of the gradient bytes is issued between the two, so the second duration is the window it can hide under.
    if args.mode == "train":
        if red is not None:
            groups_ms = model.group_times_ms()         # the profile steps above ran with red.timing: the policy's own grouping
        else:
            model._time_groups = [(0, 9), (9, 10)]
            for _ in range(max(1, args.prof_steps)):
                step()
            torch.cuda.synchronize()
            groups_ms = model.group_times_ms()
            model._time_groups = None
        budget = {f"stages_{a}_{b - 1}_ms" if b - a > 1 else f"stage_{a}_ms": round(sum(v) / len(v), 3) for (a, b), v in sorted(groups_ms.items())}
        if dp_info is None:
            dp_info = {"policy": "deferred (grouping only: single rank, no collective)"}
        dp_info["backward_group_ms"] = budget
        dp_info["overlap_window_note"] = ("HIP events around each vk_unet_backward call of the policy's grouping; under 'deferred' the one large "
                                          "all-reduce (buckets 0-8, ~97 MB fp32) is issued after stages 0-8 and can only hide under stage 9")
    if rank == 0:
        _log("event profile collected")
        if table:
            # group like rocprofv3 does (by kernel symbol): forward and data-gradient launches of one kernel share a family
            fam = {}
            for k, v in table.items():
                b = k[:-6] if k.endswith("_dgrad") else k
                f = fam.setdefault(b, dict(n=0, ms=0.0, flops=0.0, bytes=0.0))
                for q in ("n", "ms", "flops", "bytes"):
                    f[q] += v[q]
            dom = max(fam.items(), key=lambda kv: kv[1]["ms"])
            tag, r = dom
            per_launch_ms = r["ms"] / r["n"]
            if r["flops"] > 0 and any(k in tag for k in ("igemm", "wgrad", "stem", "halo", "col_", "colq_", "s2dg_")):
                peak = PEAK_MFMA_F32 if "f32" in tag else PEAK_MFMA_16B
                ach = r["flops"] / (r["ms"] * 1e-3)
                roof = {"kernel": tag, "bound": "mfma", "achieved": round(ach / 1e12, 2), "peak": peak / 1e12, "unit": "TFLOP/s",
                        "frac": round(ach / peak, 4), "traffic": None, "launches_per_step": r["n"] / args.prof_steps,
                        "avg_launch_ms": round(per_launch_ms, 4)}
            else:
                ach = r["bytes"] / (r["ms"] * 1e-3)
                roof = {"kernel": tag, "bound": "hbm", "achieved": round(ach / 1e9, 1), "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                        "frac": round(ach / PEAK_HBM, 4), "traffic": None, "launches_per_step": r["n"] / args.prof_steps,
                        "avg_launch_ms": round(per_launch_ms, 4)}
            # HBM traffic of the dominant kernel: offline rocprofv3 PMC measurement committed under profiles/ (bench.py itself
            # cannot run the profiler); null when that kernel family was not measured
            roof["algorithmic_bytes_per_launch"] = round(r["bytes"] / r["n"])
            import re as _re
            m_ = _re.match(r"(colq?)_(16b|f32)_t(\d+)_bn(\d+)_w(\d+)", tag)
            if m_:      # the symbol rocprofv3 --stats lists this family under (profiles/r02/*kernel_stats.csv)
                roof["kernel_symbol"] = (f"vk::conv3x3_{m_.group(1)}_kernel<{'float' if m_.group(2) == 'f32' else ('vk::bf16_t' if args.dtype == 'bf16' else 'vk::f16_t')}, {m_.group(3)}, {m_.group(4)}, ...> "
                                         f"({m_.group(5)} waves; forward and data-gradient launches)")
            try:
                here = kernel_source_hash()
                tpath, tj = latest_traffic(here)
                if tj.get("_src_sha256") != here:
                    roof["traffic_source"] = (f"{tpath.relative_to(ROOT)} was measured on kernel sources {str(tj.get('_src_sha256'))[:12]}, "
                                              f"this tree is {here[:12]}: stale, not reported")
                elif tag in tj:
                    roof["traffic"] = round(tj[tag]["fetch_bytes"] + tj[tag]["write_bytes"])
                    roof["traffic_source"] = (f"{tpath.relative_to(ROOT)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py, mean per "
                                              f"launch of the kernel symbol, FETCH doubled for gfx950), measured on kernel sources {here[:12]} = this tree")
            except Exception:
                pass
            if roof["bound"] == "mfma" and "f32" not in tag:
                # what the matrix pipes deliver on THIS box at the clock it sustains under load: a bare loop of independent 16x16x32
                # bf16 MFMAs on register operands, two waves on every SIMD, ~2 ms (vk_probe_mfma_rate) — context for `frac`, which
                # stays priced against the guide's 2.5 PFLOP/s
                try:
                    import ctypes as _C
                    sink = torch.zeros(4, device=dev)
                    fl = _C.c_double(0.0)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    st_ = torch.cuda.current_stream().cuda_stream
                    vk._lib.check(L.vk_probe_mfma_rate(2000, 2, sink.data_ptr(), _C.byref(fl), st_), "vk_probe_mfma_rate")   # warm-up / clock ramp
                    e0.record()
                    vk._lib.check(L.vk_probe_mfma_rate(20000, 2, sink.data_ptr(), _C.byref(fl), st_), "vk_probe_mfma_rate")
                    e1.record()
                    torch.cuda.synchronize()
                    sustained = fl.value / (e0.elapsed_time(e1) * 1e-3)
                    roof["mfma_sustained_on_this_box"] = {"tflops": round(sustained / 1e12, 1), "frac_of_peak": round(sustained / peak, 4),
                                                          "kernel_frac_of_sustained": round(ach / sustained, 4),
                                                          "how": "bare v_mfma_f32_16x16x32_bf16 loop on pseudo-random register operands, 2 waves per SIMD on every CU, ~2 ms"}
                except Exception as ex_:      # a measurement aid: never fails the bench line
                    roof["mfma_sustained_on_this_box"] = {"error": str(ex_)[:200]}
            tot = sum(v["ms"] for v in table.values())
            roof["share_of_gpu_time"] = round(r["ms"] / tot, 3)
            roof["all_kernels_ms_per_step"] = {k: round(v["ms"] / args.prof_steps, 3) for k, v in
                                               sorted(table.items(), key=lambda kv: -kv[1]["ms"])}

    # ---- CPU baseline: the oracle on this host's cores, bounded sample (rank 0, single-GPU runs only)
    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import unet_oracle as O      # the checker: CPU baseline + parity leg only (never the thing measured)
        cores = _usable_cores()
        torch.set_num_threads(cores)
        _log(f"cpu baseline on {cores} threads ...")
        O.set_seed(42)
        ref = O.build_model()
        bs = 4
        xc, yc = O.synthetic_batch(bs, S, seed=1234)
        if args.mode == "train":
            ref.train()
            ropt = torch.optim.AdamW(ref.parameters(), lr=5e-5, weight_decay=1e-4)
            O.train_steps(ref, ropt, [(xc, yc)])                  # warm-up
            t1 = time.perf_counter()
            nst = 8                                                # ~10-20 s of CPU work on a 16-core share
            O.train_steps(ref, ropt, [(xc, yc)] * nst)
            cdt = time.perf_counter() - t1
            sample = f"{nst} fp32 train steps (fwd+loss+bwd+AdamW, per-step loss.item()) at bs={bs}, {S}x{S}, after 1 warm-up"
        else:
            ref.eval()
            with torch.no_grad():
                ref(xc)
                t1 = time.perf_counter()
                nst = 12
                for _ in range(nst):
                    ref(xc)
                cdt = time.perf_counter() - t1
            sample = f"{nst} fp32 eval forwards at bs={bs}, {S}x{S}, after 1 warm-up"
        _log(f"cpu baseline done: {cdt / nst:.2f} s/step")
        cpu = {"value": round(bs * nst / cdt, 3), "unit": "images/s", "cores": cores, "kind": "port", "sample": sample,
               "cpu_model": _cpu_model()}
        # the reference's own settings beside it: torch.set_num_threads(4) (train.py:19) and batch 8 (train.py:741)
        if args.mode == "train" and not args.no_cpu_variants:
            try:
                torch.set_num_threads(min(4, cores))
                x8, y8 = O.synthetic_batch(8, S, seed=1234)
                O.set_seed(42)
                ref4 = O.build_model(); ref4.train()
                opt4 = torch.optim.AdamW(ref4.parameters(), lr=5e-5, weight_decay=1e-4)
                O.train_steps(ref4, opt4, [(x8[:2], y8[:2])])        # warm-up (allocator, oneDNN primitives)
                t1 = time.perf_counter()
                O.train_steps(ref4, opt4, [(x8, y8)])
                c4 = time.perf_counter() - t1
                cpu["variants"] = [{"value": round(8 / c4, 3), "unit": "images/s", "cores": min(4, cores), "batch": 8,
                                    "sample": f"1 fp32 train step at bs=8, {S}x{S}, torch.set_num_threads(4) as train.py:19, batch as "
                                              f"train.py:741, after a bs=2 warm-up"}]
                _log(f"cpu baseline (4 threads, bs 8): {8 / c4:.2f} img/s")
                del ref4, opt4
            finally:
                torch.set_num_threads(cores)
        # ---- "mask IoU vs ref" (BASELINE.json metric, second half): the weights the timed steps left behind, loaded into the
        # oracle (same 278 state-dict keys), fp32 eval forward of both on the same images (reference validate(), train.py:495-529)
        try:
            ref.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()}, strict=True)
            ref.eval(); model.eval()
            saved = model.compute_dtype
            model.compute_dtype = torch.float32
            with torch.no_grad():
                lo = ref(xc)
                lg = model(xc.to(dev)).float().cpu()
            model.compute_dtype = saved
            po, pg = torch.sigmoid(lo), torch.sigmoid(lg)
            mo, mg = po > 0.5, pg > 0.5
            inter, union = (mo & mg).sum().item(), (mo | mg).sum().item()
            parity = {"mask_iou_vs_target_engine": round(float(O.iou_coef(pg, yc)), 6), "mask_iou_vs_target_oracle": round(float(O.iou_coef(po, yc)), 6),
                      "mask_agreement_iou": round(inter / union, 6) if union else 1.0,
                      "max_abs_logit_err": round((lg - lo).abs().max().item(), 6),
                      "sample": f"fp32 eval forward, bs={bs}, {S}x{S}, weights after the timed steps"}
            parity["abs_iou_diff"] = round(abs(parity["mask_iou_vs_target_engine"] - parity["mask_iou_vs_target_oracle"]), 6)
            _log(f"parity leg: {parity}")
        except Exception as e:      # the baseline leg must never take the bench line down
            parity = {"error": repr(e)}

    if rank == 0:
        ips = world * N * args.steps / dt
        ci = baseline_config_index(args.mode, S, args.dtype, N, world)
        cfg_label = f"BASELINE.json configs[{ci}]" if ci is not None else "not a BASELINE.json config"
        rec = {
            "metric": ("512x512 images/sec (train fwd+bwd)" if args.mode == "train" else "512x512 images/sec (eval fwd)")
            if S == 512 else f"{S}x{S} images/sec ({args.mode})",
            "value": round(ips, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"unet_r34_{S} {args.dtype} {'train fwd+loss+bwd+AdamW, BCE+Dice' if args.mode == 'train' else 'eval forward'}, "
                                   f"bs={N}/GPU, synthetic {S}x{S} ({cfg_label})",
                       "global_batch": world * N, "image_size": S,
                       "parallelism": (f"dp{world} (RCCL all-reduce of fp32 gradients over 10 backward-ordered buckets, issue policy "
                                       f"{getattr(getattr(model, '_reducer', None), 'policy', '?')}: see parallel.py)") if world > 1 else "single GPU"},
            "conv_tflops": round((TRAIN_GFLOP_PER_IMG_512 if args.mode == "train" else FWD_GFLOP_PER_IMG_512) * (S / 512) ** 2 * ips / 1e3, 2),
            "last_loss": last,
            "api_path_ms_per_step": None if api_ms is None else round(api_ms, 3),
            "sync_each_step_ms": None if sync_ms is None else round(sync_ms, 3),
            "roofline": roof, "cpu_baseline": cpu, "parity": parity, "dp": dp_info,
        }
        print(json.dumps(rec), flush=True)
    if world > 1 or force_dist:
        dist.destroy_process_group()


def launch_plan(gpus: int, env) -> tuple:
    """What `bench.py --gpus N` has to do in this environment: ("run", world) — this process is a rank (or the single process) and the
    world size agrees with --gpus; ("spawn", N) — no launcher around us and N > 1: start N ranks; ("error", message) — the launcher's
    WORLD_SIZE contradicts --gpus (a line with the wrong n_gpus must never be printed)."""
    if gpus < 1:
        return "error", f"--gpus {gpus}: need at least one GPU"
    ws = env.get("WORLD_SIZE")
    if ws is None:
        return ("run", 1) if gpus == 1 else ("spawn", gpus)
    world = int(ws)
    if world != gpus:
        return "error", f"--gpus {gpus} but WORLD_SIZE={world}: refusing to print a line with the wrong n_gpus"
    return "run", world


def self_launch(n: int, argv) -> int:
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1 --master-port P bench.py <argv>` as a
    child process (the form the driver itself uses for N > 1); its stdout / stderr are inherited, its exit code is returned."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + list(argv)
    _log(f"--gpus {n} without a launcher: starting {' '.join(cmd[1:8])} ...")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def baseline_config_index(mode: str, size: int, dtype: str, batch: int, world: int):
    """Which BASELINE.json configs[] entry a run is (None: not one of them)."""
    if mode == "infer" and size == 512 and dtype == "fp32" and batch == 16 and world == 1:
        return 1
    if mode == "train" and size == 512 and dtype == "bf16" and batch == 32:
        return 2 if world == 1 else 3
    if mode == "train" and size == 1024 and dtype == "fp16" and batch == 8:
        return 4
    return None


def kernel_source_hash() -> str:
    """sha256 over the kernel sources the library is built from (stable across rebuilds, unlike the .so): ties a committed
    PMC traffic measurement to the code being benched."""
    import hashlib
    h = hashlib.sha256()
    files = sorted((ROOT / "vickers-hardness-unet_amd" / "csrc").glob("*.hip")) + sorted((ROOT / "vickers-hardness-unet_amd" / "csrc").glob("*.h")) \
        + [ROOT / "include" / "vk_unet.h"]
    for f in files:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()


def latest_traffic(src_hash):
    """The committed PMC traffic measurement to quote: the newest profiles/rNN/traffic.json whose source stamp matches this tree,
    else the newest one at all (reported as stale by the caller)."""
    cands = sorted((ROOT / "profiles").glob("r[0-9][0-9]/traffic.json"), reverse=True)
    loaded = [(p, json.load(open(p))) for p in cands]
    for p, tj in loaded:
        if tj.get("_src_sha256") == src_hash:
            return p, tj
    if not loaded:
        raise FileNotFoundError("no profiles/rNN/traffic.json")
    return loaded[0]


def _log(msg):
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def _usable_cores():
    """Threads we may actually run: affinity, capped by the cgroup CPU quota and by the 16-core share of a
    one-GPU box (oversubscribing OpenMP threads makes the CPU leg crawl)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 16))


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
