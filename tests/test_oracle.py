"""CPU tests that pin the oracle: known-answer parameter count, state-dict key
set, the reference's own lr log columns, the reference's own dice_coef/iou_coef outputs, and
drift detection against the committed small golden tensors."""
import json
import math

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_loops_batches


def test_param_count_and_keys(oracle):
    oracle.set_seed(42)
    m = oracle.build_model()
    assert sum(p.numel() for p in m.parameters()) == 24_436_369
    sd = m.state_dict()
    assert len(sd) == 278
    assert len(list(m.parameters())) == 140
    man = json.load(open(GOLDEN / "manifest.json"))
    assert man["param_count"] == 24_436_369
    assert list(man["entries"].keys()) == list(sd.keys())
    for k, shp in man["entries"].items():
        assert list(sd[k].shape) == shp, k
    # spot checks from SURVEY.md §8(b)
    assert sd["encoder.conv1.weight"].shape == (64, 3, 7, 7)
    assert sd["encoder.layer2.0.downsample.0.weight"].shape == (128, 64, 1, 1)
    assert sd["decoder.blocks.0.conv1.0.weight"].shape == (256, 768, 3, 3)
    assert sd["decoder.blocks.4.conv1.0.weight"].shape == (16, 32, 3, 3)
    assert sd["segmentation_head.0.weight"].shape == (1, 16, 3, 3)
    assert sd["segmentation_head.0.bias"].shape == (1,)
    assert not any("ConvTranspose" in type(x).__name__ for x in m.modules())


def test_group_param_counts(oracle):
    """Per-group counts listed in SURVEY.md §2.2 (gradient bucketing table)."""
    m = oracle.build_model()
    conv = lambda prefix: sum(v.numel() for k, v in m.named_parameters()
                              if k.startswith(prefix) and v.dim() == 4)
    assert conv("encoder.conv1") == 9_408
    assert conv("encoder.layer1") == 221_184
    assert conv("encoder.layer2") == 1_114_112
    assert conv("encoder.layer3") == 6_815_744
    assert conv("encoder.layer4") == 13_107_200
    assert conv("decoder.blocks.0") == 2_359_296
    assert conv("decoder.blocks.1") == 589_824
    assert conv("decoder.blocks.2") == 147_456
    assert conv("decoder.blocks.3") == 46_080
    assert conv("decoder.blocks.4") == 6_912
    bn_affine = sum(v.numel() for k, v in m.named_parameters() if v.dim() == 1 and "segmentation_head" not in k)
    assert bn_affine == 19_008


@pytest.mark.parametrize("name", ["history.json", "history_0.json"])
def test_lr_schedule_against_reference_logs(oracle, name):
    rec = json.load(open(GOLDEN / "lr_history.json"))[name]
    lr0, t_max = rec["lr0"], rec["t_max"]
    # closed form
    for e, lr in enumerate(rec["lr"], start=1):
        assert math.isclose(oracle.cosine_lr(lr0, e, t_max), lr, rel_tol=1e-9, abs_tol=1e-18)
    # and the torch scheduler the reference uses (train.py:606-607, 647, 656)
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=lr0, weight_decay=1e-4)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=t_max)
    for lr in rec["lr"][:50]:
        opt.step()
        sch.step()
        assert math.isclose(opt.param_groups[0]["lr"], lr, rel_tol=1e-7, abs_tol=1e-15)


def test_metrics_against_reference_functions(oracle):
    cases = json.load(open(GOLDEN / "metrics_ref.json"))
    for c in cases:
        g = torch.Generator().manual_seed(c["seed"])
        prob = torch.rand(c["n"], 1, c["s"], c["s"], generator=g)
        tgt = (torch.rand(c["n"], 1, c["s"], c["s"], generator=g) > c["thr"]).float()
        if c["seed"] == 3:
            tgt.zero_()
            prob.mul_(0.4)
        assert oracle.dice_coef(prob, tgt) == pytest.approx(c["dice"], abs=1e-7)
        assert oracle.iou_coef(prob, tgt) == pytest.approx(c["iou"], abs=1e-7)


def test_bce_formula(oracle):
    """mean(max(x,0) − x·y + log1p(exp(−|x|)))."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 1, 16, 16, generator=g, dtype=torch.float64) * 4
    y = (torch.rand(2, 1, 16, 16, generator=g) > 0.5).double()
    ref = torch.nn.BCEWithLogitsLoss()(x, y)
    mine = (x.clamp_min(0) - x * y + torch.log1p(torch.exp(-x.abs()))).mean()
    assert torch.allclose(ref, mine, atol=1e-14)


def test_dice_loss_edge_cases(oracle):
    d = oracle.DiceLoss()
    x = torch.randn(2, 1, 8, 8)
    assert d(x, torch.zeros(2, 1, 8, 8)).item() == 0.0          # empty target => masked to 0
    y = torch.ones(2, 1, 8, 8)
    big = torch.full((2, 1, 8, 8), 30.0)
    assert d(big, y).item() == pytest.approx(0.0, abs=1e-6)      # perfect prediction
    assert d(-big, y).item() == pytest.approx(1.0, abs=1e-6)


def test_oracle_matches_committed_golden(oracle):
    z = np.load(GOLDEN / "oracle_small.npz")
    oracle.set_seed(42)
    m = oracle.build_model()
    x, y = oracle.synthetic_batch(2, 64, seed=1234)
    m.eval()
    with torch.no_grad():
        le = m(x).numpy()
    assert np.allclose(le, z["logits_eval"], atol=1e-5)
    m.train()
    lt = m(x)
    assert np.allclose(lt.detach().numpy(), z["logits_train"], atol=1e-4)
    loss = oracle.total_loss(lt, y)
    assert loss.item() == pytest.approx(float(z["loss"]), rel=1e-5)
    loss.backward()
    named = dict(m.named_parameters())
    for k in z.files:
        if k.startswith("grad::"):
            gref = z[k]
            g = named[k[6:]].grad.numpy()
            assert np.allclose(g, gref, atol=1e-5 + 1e-3 * np.abs(gref).max()), k


def test_upsample_is_floor_div(oracle):
    """SURVEY.md §8(a) row 4 [probe]: nearest ×2 => out[y,x] = in[y//2,x//2]."""
    a = torch.arange(12.0).view(1, 1, 3, 4)
    u = torch.nn.functional.interpolate(a, scale_factor=2, mode="nearest")
    for yy in range(6):
        for xx in range(8):
            assert u[0, 0, yy, xx] == a[0, 0, yy // 2, xx // 2]


def test_synthetic_masks_nonempty(oracle):
    _, y = oracle.synthetic_batch(8, 128)
    frac = y.mean(dim=(1, 2, 3))
    assert (frac > 0.005).all() and (frac < 0.2).all()


def _loops_fixture_model(oracle):
    oracle.set_seed(42)
    model = oracle.build_model()
    opt = torch.optim.AdamW(model.parameters(), lr=5e-5, weight_decay=1e-4)
    return model, opt


def test_loop_restatements_against_reference_loops(oracle):
    """8c, loop half: tests/golden/loops_ref.json holds what the REFERENCE's own train_one_epoch (train.py:381-459) and validate
    (train.py:461-529) returned in the build container over three seeded batches (2, 2, 1 images) and two validation batches.
    The oracle's restatements — train_one_epoch / validate_epoch (the reference's signatures) and train_steps / validate (the
    list-of-pairs forms every GPU parity test uses) — must reproduce them: same torch CPU ops, same order, same thread count,
    so the bar is float equality up to 1e-6 relative (bit-equal in the container the fixture was made in)."""
    ref = json.load(open(GOLDEN / "loops_ref.json"))
    train, val = golden_loops_batches()
    nthreads = torch.get_num_threads()
    torch.set_num_threads(ref["threads"])
    try:
        bce, dice = torch.nn.BCEWithLogitsLoss(), oracle.DiceLoss(mode="binary")
        model, opt = _loops_fixture_model(oracle)
        epoch = oracle.train_one_epoch(model, train, opt, bce, dice, "cpu", scaler=None)
        assert epoch == pytest.approx(ref["train_one_epoch"], rel=1e-6)
        vl, vd, vi = oracle.validate_epoch(model, val, bce, dice, "cpu")
        assert vl == pytest.approx(ref["validate"]["loss"], rel=1e-6)
        assert vd == pytest.approx(ref["validate"]["dice"], rel=1e-6)
        assert vi == pytest.approx(ref["validate"]["iou"], rel=1e-6)
        named = dict(model.named_parameters())
        for k, c in ref["checksums"].items():
            assert float(named[k].detach().double().sum()) == pytest.approx(c["sum"], rel=1e-6, abs=1e-9), k
            assert float(named[k].detach().double().abs().sum()) == pytest.approx(c["abs_sum"], rel=1e-7), k
        assert np.allclose(model.encoder.bn1.running_mean.double().numpy(), ref["bn1_running_mean"], rtol=1e-6, atol=1e-9)
        assert np.allclose(model.encoder.bn1.running_var.double().numpy(), ref["bn1_running_var"], rtol=1e-6)
        assert int(model.encoder.bn1.num_batches_tracked) == ref["bn1_num_batches_tracked"] == 3

        # the list-of-pairs forms: per-step losses = bce + dice of the reference's steps; the epoch mean is sample-weighted
        model2, opt2 = _loops_fixture_model(oracle)
        steps = oracle.train_steps(model2, opt2, [(x, y) for x, y, _ in train])
        want = [s_["bce"] + s_["dice"] for s_ in ref["train_steps"]]
        assert steps == pytest.approx(want, rel=1e-6)
        sizes = [x.size(0) for x, _, _ in train]
        assert sum(l * n for l, n in zip(steps, sizes)) / sum(sizes) == pytest.approx(ref["train_one_epoch"], rel=1e-6)
        v2 = oracle.validate(model2, [(x, y) for x, y, _ in val])
        assert v2 == pytest.approx((ref["validate"]["loss"], ref["validate"]["dice"], ref["validate"]["iou"]), rel=1e-6)
    finally:
        torch.set_num_threads(nthreads)
