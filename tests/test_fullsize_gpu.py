"""The BASELINE workloads THEMSELVES under the oracle (-m gpu): configs[2] (bf16, bs 32, 512x512, three AdamW steps) and the
configs[4] geometry (fp16 + loss scale 2^16, bs 8, 1024x1024, one step) run on the engine and on the fp32 CPU oracle
(train.py:423-452 step order; `validate` train.py:495-529 for the metric leg) from identical seeds (42 weights / 1234 data).

Compared, with the tolerance in each assert:
  * step-1 loss and both of its components (BCE, Dice)           rel <= 1e-2
  * the three-step loss trajectory                               rel <= 2e-2 per step, and it decreases like the oracle's
  * BatchNorm running statistics of the first and the last BN    fp16: <= 2e-3 of the statistic's scale; bf16: <= 1e-2 / 3e-2
    (rounding a filter to 8 mantissa bits moves a channel mean of N(0,1) inputs by ~2^-9 of the filter norm: the oracle under CPU
    bf16 autocast, printed beside it, is off by the same amount)
  * gradient direction of named parameters against the fp32 oracle, with the oracle under CPU bf16 autocast — the reference's
    own mixed-precision arithmetic on another backend — as the yardstick (cosine no worse than the yardstick's - 0.02)
The oracle steps cost ~10 s each on the 16 host threads of a GPU box; nothing here reads /root/reference."""
import importlib
import time

import pytest
import torch

pytestmark = pytest.mark.gpu
vk = importlib.import_module("vickers-hardness-unet_amd")

NAMED = ["encoder.conv1.weight", "encoder.layer1.0.conv1.weight", "encoder.layer2.0.downsample.0.weight", "encoder.layer3.2.conv2.weight",
         "encoder.layer4.2.conv2.weight", "decoder.blocks.0.conv1.0.weight", "decoder.blocks.3.conv1.0.weight",
         "decoder.blocks.4.conv2.0.weight", "decoder.blocks.4.conv2.1.weight", "segmentation_head.0.weight", "segmentation_head.0.bias"]


def dev():
    return torch.device("cuda:0")


def _cos(a, b):
    a, b = a.flatten().double(), b.flatten().double()
    return (a @ b / (a.norm() * b.norm() + 1e-300)).item()


def _oracle_steps(O, x, y, steps, autocast_dtype=None):
    """`steps` AdamW steps of the oracle on (x, y); returns per-step (total, bce, dice), the step-1 gradients of NAMED and the model."""
    O.set_seed(42)
    ref = O.build_model()
    ref.train()
    opt = torch.optim.AdamW(ref.parameters(), lr=5e-5, weight_decay=1e-4)
    bce, dice = torch.nn.BCEWithLogitsLoss(), O.DiceLoss()
    rec, grads = [], None
    ref.stats_after_step1 = None
    for s in range(steps):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cpu", dtype=autocast_dtype or torch.bfloat16, enabled=autocast_dtype is not None):
            logits = ref(x)
            lb, ld = bce(logits.float(), y), dice(logits.float(), y)
            loss = lb + ld
        loss.backward()
        if s == 0:
            named = dict(ref.named_parameters())
            grads = {k: named[k].grad.detach().clone() for k in NAMED}
            ref.stats_after_step1 = {k: v.clone() for k, v in ref.state_dict().items() if "running_" in k}
        opt.step()
        rec.append((loss.item(), lb.item(), ld.item()))
    return rec, grads, ref


def _engine_steps(O, x, y, steps, dtype, loss_scale=1.0):
    O.set_seed(42)
    model = vk.Unet(encoder_weights=None).to(dev())
    model.train()
    opt = vk.adamw_for(model, lr=5e-5, weight_decay=1e-4)
    xd, yd = x.to(dev()), y.to(dev())
    scale_t = torch.full((1,), loss_scale, device=dev()) if loss_scale != 1.0 else None
    rec, grads = [], None
    for s in range(steps):
        opt.zero_grad(set_to_none=True)
        out = model.loss_and_backward(xd, yd, grad_scale=loss_scale, dtype=dtype)
        if s == 0:
            named = dict(model.named_parameters())
            grads = {k: (named[k].grad.detach() / loss_scale).cpu().clone() for k in NAMED}
        opt.step(grad_scale=scale_t)
        rec.append(tuple(out.tolist()))
    torch.cuda.synchronize()
    return rec, grads, model


BN_KEYS = ("encoder.bn1.running_mean", "encoder.bn1.running_var", "decoder.blocks.4.conv2.1.running_mean", "decoder.blocks.4.conv2.1.running_var")


def _compare(tag, eng, ora, yard, g_e, g_o, g_y, model, ref, steps, bn_bars=(2e-3, 2e-3, 2e-3, 2e-3), yard_stats=None, cos_floor=0.90):
    print(f"[{tag}] losses (total, bce, dice) engine {eng} | fp32 oracle {ora}" + (f" | autocast oracle {yard}" if yard else ""))
    # step 1: total and both components
    for j, nm in enumerate(("total", "bce", "dice")):
        assert eng[0][j] == pytest.approx(ora[0][j], rel=1e-2), (nm, eng[0], ora[0])
    # trajectory
    for s in range(steps):
        assert eng[s][0] == pytest.approx(ora[s][0], rel=2e-2), (s, eng, ora)
    if steps > 1:
        assert eng[-1][0] < eng[0][0] and ora[-1][0] < ora[0][0]
        assert (eng[0][0] - eng[-1][0]) == pytest.approx(ora[0][0] - ora[-1][0], rel=0.25), (eng, ora)
    # BatchNorm running statistics after `steps` momentum-0.1 updates
    sd_e, sd_o = model.state_dict(), ref.state_dict()
    for k, bar in zip(BN_KEYS, bn_bars):
        a, b = sd_e[k].cpu(), sd_o[k]
        err = (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)
        note = ""
        if yard_stats is not None:      # the mixed-precision oracle after ITS one step against the fp32 oracle after one step
            b1 = ref.stats_after_step1[k]
            note = f" (autocast oracle after one step: {(yard_stats[k] - b1).abs().max().item() / (b1.abs().max().item() + 1e-12):.2e})"
        print(f"[{tag}] {k}: max|diff| / max|oracle| = {err:.2e}{note}")
        assert err <= bar, (k, err)
    assert int(sd_e["encoder.bn1.num_batches_tracked"]) == steps == int(sd_o["encoder.bn1.num_batches_tracked"])
    # gradient direction, step 1 (every parameter is printed before anything is asserted)
    worst, bad = 1.0, []
    for k in NAMED:
        c_e = _cos(g_e[k], g_o[k])
        c_y = _cos(g_y[k], g_o[k]) if g_y is not None else None
        n_e = (g_e[k].double().norm() / (g_o[k].double().norm() + 1e-300)).item()
        print(f"[{tag}] grad {k}: cosine vs fp32 oracle engine {c_e:.4f}" + (f", autocast oracle {c_y:.4f}" if c_y is not None else "")
              + f"; norm ratio {n_e:.3f}")
        fails = []
        if c_y is not None and c_e < c_y - 0.02:
            fails.append(("below the autocast yardstick", k, c_e, c_y))
        if c_e < cos_floor:
            fails.append(("below the floor", k, c_e, cos_floor))
        if not 0.8 <= n_e <= 1.25:
            fails.append(("norm", k, n_e))
        worst = min(worst, c_e)
        bad += fails
    assert not bad, bad
    return worst


def test_config3_bf16_bs32_512_three_steps_vs_oracle():
    """BASELINE.json configs[2] — the workload the headline images/s is quoted on — against the oracle (BASELINE.md section 2 row 3:
    loss-trajectory parity)."""
    from oracle import unet_oracle as O
    torch.set_num_threads(min(16, torch.get_num_threads()))
    x, y = O.synthetic_batch(32, 512, seed=1234)
    t0 = time.perf_counter()
    ora, g_o, ref = _oracle_steps(O, x, y, 3)
    t1 = time.perf_counter()
    yard, g_y, ref_y = _oracle_steps(O, x, y, 1, autocast_dtype=torch.bfloat16)
    yard_stats = ref_y.stats_after_step1
    del ref_y
    t2 = time.perf_counter()
    eng, g_e, model = _engine_steps(O, x, y, 3, torch.bfloat16)
    print(f"[configs[2]] oracle fp32 3 steps {t1 - t0:.1f} s, autocast oracle 1 step {t2 - t1:.1f} s")
    _compare("configs[2] bf16 bs32 512", eng, ora, yard, g_e, g_o, g_y, model, ref, 3, bn_bars=(1e-2, 1e-2, 3e-2, 3e-2), yard_stats=yard_stats,
             cos_floor=0.5)      # bf16 at random init: the reference's own mixed-precision arithmetic (CPU autocast) reaches 0.61 on the stem filter


def test_config5_fp16_bs8_1024_one_step_vs_oracle():
    """The configs[4] workload (fp16, bs 8, 1024x1024, the reference's loss scale 2^16 folded into the loss-gradient and AdamW kernels)
    for one step against the fp32 oracle; yardstick: the oracle under CPU bf16 autocast (CPU fp16 autocast convolutions are not
    dependable across oneDNN builds; bf16 has the coarser mantissa, so it is the more lenient yardstick only for rounding — the
    absolute 0.90 / norm bars hold regardless)."""
    from oracle import unet_oracle as O
    torch.set_num_threads(min(16, torch.get_num_threads()))
    x, y = O.synthetic_batch(8, 1024, seed=1234)
    ora, g_o, ref = _oracle_steps(O, x, y, 1)
    eng, g_e, model = _engine_steps(O, x, y, 1, torch.float16, loss_scale=65536.0)
    _compare("configs[4] fp16 bs8 1024", eng, ora, None, g_e, g_o, None, model, ref, 1)


def test_validate_and_metrics_vs_oracle():
    """`validate` (train.py:495-529) on the engine against the oracle on the same loader and weights: loss, Dice, IoU; and the
    device metric kernel (vk.dice_coef / vk.iou_coef / vk.seg_metrics) against the oracle's restatement on the same probabilities."""
    from oracle import unet_oracle as O
    O.set_seed(42); ref = O.build_model()
    model = vk.Unet(encoder_weights=None).to(dev())
    # calibrated running statistics (one momentum-1 train pass of the oracle), as an inference checkpoint has
    x, y = O.synthetic_batch(6, 128, seed=77)
    for m in ref.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = 1.0
    ref.train()
    with torch.no_grad():
        ref(x)
    model.load_state_dict(ref.state_dict(), strict=True)
    batches = [(x[:4], y[:4]), (x[4:], y[4:])]
    vl_o, vd_o, vi_o = O.validate(ref, batches)
    bce, dice = torch.nn.BCEWithLogitsLoss(), vk.DiceLoss(mode="binary")
    vl_e, vd_e, vi_e = O.validate_epoch(model, [(a, b, None) for a, b in batches], bce, dice, "cuda")
    print(f"validate: engine ({vl_e:.6f}, {vd_e:.6f}, {vi_e:.6f}) oracle ({vl_o:.6f}, {vd_o:.6f}, {vi_o:.6f})")
    assert vl_e == pytest.approx(vl_o, rel=1e-4)
    assert abs(vd_e - vd_o) <= 1e-4 and abs(vi_e - vi_o) <= 1e-4            # north_star: IoU within 1e-4
    # the same validate with the package's device metrics in place of the five torch reductions
    model.eval()
    dd, uu = [], []
    with torch.no_grad():
        for a, b in batches:
            lg = model(a.to(dev()))
            prob = torch.sigmoid(lg)
            d1, u1 = vk.dice_coef(prob, b.to(dev())), vk.iou_coef(prob, b.to(dev()))
            assert d1 == pytest.approx(O.dice_coef(prob.cpu(), b), abs=2e-7) and u1 == pytest.approx(O.iou_coef(prob.cpu(), b), abs=2e-7)
            d2, u2 = vk.seg_metrics(lg, b.to(dev()), from_logits=True)
            assert abs(d2 - d1) <= 1e-6 and abs(u2 - u1) <= 1e-6
            dd.append(d1); uu.append(u1)
    assert abs(sum(dd) / 2 - vd_o) <= 1e-4 and abs(sum(uu) / 2 - vi_o) <= 1e-4
