"""CPU tests of the host side: the C-ABI library loads without a GPU and exports every symbol that
include/vk_unet.h declares; the parameter table equals smp's state-dict (via the oracle manifest);
default initialisation reproduces the oracle bit-for-bit; argument errors are reported, not thrown;
compute entry points refuse CPU tensors (no fallback)."""
import ctypes as C
import json
import re
from pathlib import Path

import pytest
import torch

from conftest import GOLDEN, ROOT


def test_header_symbols_all_exported(vk):
    hdr = (ROOT / "include" / "vk_unet.h").read_text()
    declared = set(re.findall(r"\b(vk_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"vk_unet_param_info"}      # names used in prose only
    L = vk.lib()
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, missing
    assert declared == set(vk._lib.SIGNATURES), declared ^ set(vk._lib.SIGNATURES)
    assert L.vk_version() == 1


def test_param_table_matches_smp_manifest(vk):
    man = json.load(open(GOLDEN / "manifest.json"))
    m = vk.Unet(encoder_name="resnet34", encoder_weights=None, in_channels=3, classes=1, activation=None)
    sd = m.state_dict()
    assert list(sd.keys()) == list(man["entries"].keys())
    for k, shp in man["entries"].items():
        assert list(sd[k].shape) == shp, k
    assert sum(p.numel() for p in m.parameters()) == 24_436_369 == man["param_count"]
    assert len(list(m.parameters())) == 140 and len(sd) == 278
    w = sd["encoder.layer1.0.conv1.weight"]
    assert w.permute(0, 2, 3, 1).is_contiguous()        # KRSC in memory == channels_last


def test_default_init_reproduces_oracle(vk, oracle):
    oracle.set_seed(42)
    ref = oracle.build_model()
    oracle.set_seed(42)
    m = vk.build_model("resnet34", None)
    so, sg = ref.state_dict(), m.state_dict()
    for k in so:
        assert torch.equal(so[k], sg[k]), k


def test_load_state_dict_strict_and_dirty_tracking(vk, oracle):
    oracle.set_seed(3)
    ref = oracle.build_model()
    m = vk.Unet(encoder_weights=None)
    v0 = m._weights_version()
    m.load_state_dict(ref.state_dict(), strict=True)
    assert m._weights_version() != v0
    assert torch.equal(m.state_dict()["decoder.blocks.0.conv1.0.weight"], ref.state_dict()["decoder.blocks.0.conv1.0.weight"])
    bad = dict(ref.state_dict())
    bad.pop("segmentation_head.0.bias")
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad, strict=True)


def test_checkpoint_file_round_trip(vk, oracle, tmp_path):
    """train.py:668-678 writes `torch.save(model.state_dict(), last.pth)`; infer_pth_gui.py:35-43 reads it back with
    `torch.load(..., weights_only=True)` + `load_state_dict` (strict).  Both directions against the smp-keyed oracle
    model, all 278 tensors bit-identical (the part that needs no real .pth)."""
    oracle.set_seed(11)
    ref = oracle.build_model()
    m = vk.Unet(encoder_weights=None)
    # reference-written checkpoint -> this package
    torch.save(ref.state_dict(), tmp_path / "last.pth")
    sd = torch.load(tmp_path / "last.pth", map_location="cpu", weights_only=True)
    m.load_state_dict(sd)
    for k, v in ref.state_dict().items():
        assert torch.equal(m.state_dict()[k], v), k
    # checkpoint written by this package -> reference model
    torch.save(m.state_dict(), tmp_path / "best.pth")
    sd2 = torch.load(tmp_path / "best.pth", map_location="cpu", weights_only=True)
    assert list(sd2.keys()) == list(ref.state_dict().keys())
    assert all(sd2[k].dtype == v.dtype and sd2[k].shape == v.shape for k, v in ref.state_dict().items())
    oracle.set_seed(12)
    other = oracle.build_model()
    other.load_state_dict(sd2, strict=True)
    for k, v in ref.state_dict().items():
        assert torch.equal(other.state_dict()[k], v), k


def test_constructor_rejects_what_the_reference_does_not_use(vk):
    with pytest.raises(vk.VkError):
        vk.Unet(encoder_weights="imagenet")          # needs a download
    with pytest.raises(NotImplementedError):
        vk.Unet(encoder_name="resnet50", encoder_weights=None)
    with pytest.raises(NotImplementedError):
        vk.Unet(encoder_weights=None, classes=2)
    with pytest.raises(NotImplementedError):
        vk.DiceLoss(mode="multiclass")


def test_no_cpu_fallback(vk):
    m = vk.Unet(encoder_weights=None)
    with pytest.raises(vk.VkError):
        m(torch.zeros(1, 3, 64, 64))
    with pytest.raises(RuntimeError, match="divisible by 32"):
        m(torch.zeros(1, 3, 70, 70))
    with pytest.raises(vk.VkError):
        vk.DiceLoss()(torch.zeros(1, 1, 8, 8, requires_grad=True), torch.zeros(1, 1, 8, 8))
    opt = vk.adamw_for(m, lr=5e-5)
    with pytest.raises(vk.VkError):
        opt.step()


def test_c_abi_argument_errors(vk):
    L = vk.lib()
    h = C.c_void_p()
    cfg = vk._lib.vk_unet_config(1, 100, 0, 0)         # 100 % 32 != 0
    assert L.vk_unet_create(C.byref(cfg), C.byref(h)) < 0
    assert b"divisible by 32" in L.vk_last_error_string()
    cfg = vk._lib.vk_unet_config(1, 64, 0, 1)
    assert L.vk_unet_create(C.byref(cfg), C.byref(h)) == 0
    assert L.vk_unet_forward(h, None, None, 0, None) < 0          # not bound
    assert L.vk_unet_workspace_bytes(h) > 0
    ti = vk._lib.vk_tensor_info()
    assert L.vk_unet_tensor_info(h, 10_000, C.byref(ti)) < 0
    L.vk_unet_destroy(h)
    assert L.vk_adamw_step(0, None, None, None, None, 0.0, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, None, None, 0, None) < 0
    assert L.vk_adamw_step_amp(0, None, None, None, None, 0.0, 0.9, 0.999, 1e-8, 0.0, None, 1.0, None, None, None, None, 0, None) < 0
    assert L.vk_amp_unscale_check(0, None, None, None, None) < 0
    # pre/post-processing descriptors are checked on the host before anything is launched
    lb = vk._lib.vk_letterbox_desc
    assert L.vk_letterbox_preprocess(None, None, None, None) < 0
    assert L.vk_letterbox_preprocess(C.byref(lb(10, 10, 30, 64, 70, 10, 0, 0, 0)), None, None, None) < 0      # 70 rows in a 64 square
    assert b"does not fit" in L.vk_last_error_string()
    assert L.vk_letterbox_preprocess(C.byref(lb(10, 10, 30, 64, 10, 10, 0, 0, 300)), None, None, None) < 0     # border value > 255
    assert L.vk_letterbox_postprocess_mask(C.byref(lb(10, 10, 0, 64, 10, 10, 60, 0, 0)), None, 0.5, None, None) < 0   # top + nh > S
    assert L.vk_letterbox_postprocess_prob(C.byref(lb(10, 10, 0, 64, 10, 10, 0, 0, 0)), None, None, None) < 0  # null buffers
    assert L.vk_conv_fwd_splitk(None, None, None, None, 0, None) < 0


def test_optimizer_is_a_torch_optimizer(vk):
    m = vk.Unet(encoder_weights=None)
    opt = vk.adamw_for(m, lr=5e-5, weight_decay=1e-4)
    assert isinstance(opt, torch.optim.Optimizer)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=500)
    assert opt.param_groups[0]["lr"] == 5e-5 and opt.param_groups[0]["weight_decay"] == 1e-4
    lr_hist = json.load(open(GOLDEN / "lr_history.json"))["history.json"]["lr"]
    for e in range(3):
        sch.step()
        assert opt.param_groups[0]["lr"] == pytest.approx(lr_hist[e], rel=1e-7)


def test_optimizer_state_dict_is_not_mutated_and_amp_protocol_declared(vk):
    """load_state_dict must not pop keys out of the caller's dict, loaded moments are kept (moved, never re-zeroed),
    and the optimizer declares the GradScaler fast path (train.py:441-445)."""
    m = vk.Unet(encoder_weights=None)
    opt = vk.adamw_for(m, lr=5e-5, weight_decay=1e-4)
    assert getattr(opt, "_step_supports_amp_scaling", False) is True
    assert issubclass(vk.GradScaler, torch.amp.GradScaler)
    n = m.flat_params.numel()
    sd = opt.state_dict()
    sd["fused"] = {"step": 7, "exp_avg": torch.full((n,), 0.25), "exp_avg_sq": torch.full((n,), 0.5)}
    keys = set(sd)
    opt2 = vk.adamw_for(m, lr=5e-5, weight_decay=1e-4)
    opt2.load_state_dict(sd)
    assert set(sd) == keys and "fused" in sd
    assert opt2.step_count == 7 and float(opt2._m[0]) == 0.25 and float(opt2._v[-1]) == 0.5
    assert opt2.state_dict()["fused"]["step"] == 7


def test_synthetic_matches_oracle(vk, oracle):
    """bench.py's GPU leg takes its inputs from the package (vk.synthetic_batch / vk.seed_everything), the CPU-baseline and every
    parity test from the oracle: both generators must draw the same numbers."""
    for n, s, seed in ((2, 64, 1234), (3, 96, 1235), (1, 32, 7)):
        xa, ya = vk.synthetic_batch(n, s, seed=seed)
        xb, yb = oracle.synthetic_batch(n, s, seed=seed)
        assert torch.equal(xa, xb) and torch.equal(ya, yb)
        fg = ya.mean(dim=(1, 2, 3))
        assert (fg > 0.005).all() and (fg < 0.13).all()          # 1-12 % foreground, never empty
    vk.seed_everything(42); a = torch.rand(3)
    oracle.set_seed(42); b = torch.rand(3)
    assert torch.equal(a, b)
    vk.seed_everything(42); m1 = vk.Unet(encoder_weights=None)
    oracle.set_seed(42); m2 = vk.Unet(encoder_weights=None)
    assert torch.equal(m1.flat_params, m2.flat_params)


def test_package_holds_no_reference_loop(vk):
    """The epoch loop and the metric definitions are the reference's own code and stay on the user's side:
    the package exports device kernels, not a restated train.py."""
    assert not hasattr(vk, "train_one_epoch") and not hasattr(vk, "validate")
    assert not (ROOT / "vickers-hardness-unet_amd" / "train.py").exists()
    assert not (ROOT / "vickers-hardness-unet_amd" / "metrics.py").exists()


def test_bench_launch_plan_and_config_labels():
    """bench.py --gpus N: a plain invocation must start N ranks itself, a launcher's WORLD_SIZE that contradicts --gpus must never
    produce a line, and the config label follows mode / size / dtype / batch / world (BASELINE.json configs)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("vk_bench", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.launch_plan(1, {}) == ("run", 1)
    assert bench.launch_plan(8, {}) == ("spawn", 8)
    assert bench.launch_plan(2, {"WORLD_SIZE": "2", "RANK": "1"}) == ("run", 2)
    for gpus, ws in ((8, "1"), (1, "2"), (4, "8")):
        action, msg = bench.launch_plan(gpus, {"WORLD_SIZE": ws})
        assert action == "error" and f"WORLD_SIZE={ws}" in msg
    assert bench.launch_plan(0, {})[0] == "error"
    assert bench.baseline_config_index("infer", 512, "fp32", 16, 1) == 1
    assert bench.baseline_config_index("train", 512, "bf16", 32, 1) == 2
    assert bench.baseline_config_index("train", 512, "bf16", 32, 8) == 3
    assert bench.baseline_config_index("train", 1024, "fp16", 8, 8) == 4
    assert bench.baseline_config_index("train", 1024, "fp16", 8, 1) == 4
    assert bench.baseline_config_index("infer", 512, "bf16", 16, 1) is None
    assert bench.baseline_config_index("train", 512, "fp32", 32, 1) is None


def test_bench_self_launch_spawns_ranks_and_relays_exit_code(tmp_path, monkeypatch):
    """The spawn path end to end without a GPU: `bench.py --gpus 2` (no WORLD_SIZE) starts torch.distributed.run with two ranks;
    each rank gets as far as the "needs an MI355X" refusal here, and the parent exits with the launcher's non-zero code."""
    import subprocess, sys, os
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "starting -m torch.distributed.run --nnodes=1 --nproc-per-node=2" in r.stderr
    assert r.stderr.count("bench.py needs an MI355X") >= 2, r.stderr[-2000:]


def test_shipped_binary_has_no_uncovered_mfma_hazard(vk):
    """tools/mfma_hazard_audit.py over the gfx950 code objects inside the shipped libvkunet.so: along every control-flow path, no
    non-MFMA instruction touches an MFMA's destination earlier than the wait states this toolchain pads on straight-line code.
    (r03's wrong 16-bit inference was exactly such a site — `v_accvgpr_read` 2 states behind its MFMA across a taken branch —
    produced by hipcc, not by the source: the check runs on the binary that ships.)"""
    import subprocess, sys
    so = Path(vk._lib.LIB_PATH)
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "mfma_hazard_audit.py"), str(so)], capture_output=True, text=True, timeout=600)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:]
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    m = re.search(r"(\d+) kernels with MFMAs: 0 violation", tail)
    assert m and int(m.group(1)) >= 100, tail
    # no kernel of the compute path spills or uses scratch (the geometry post-processing kernels keep small per-thread arrays there by design)
    scratch = [l for l in r.stdout.splitlines() if l.startswith("scratch: ")]
    assert all(l.split()[1].startswith("_ZN2vk11k_geom_") or l.split()[1].startswith("_ZN2vk12k_geom_") for l in scratch), scratch
