"""Data-parallel plumbing on CPU with the gloo backend, world_size 2.

The GradientReducer only needs a flat gradient tensor and bucket ranges, so it is exercised here with
the ORACLE producing the per-rank gradients (tests may use the oracle; the product path never does):
  * bucketed async all-reduce + 1/world scaling == mean of the per-shard gradients ("K shards through
    identical weights on CPU, mean of shard gradients" — the DP oracle of SURVEY.md §8(e)),
  * rank-0 broadcast of parameters / BN buffers,
  * the engine's bucket table (host-only C call) is contiguous, ordered by backward completion and
    covers the whole flat buffer."""
import ctypes as C
import importlib
import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    vk = importlib.import_module("vickers-hardness-unet_amd")
    from oracle import unet_oracle as O

    # every rank builds the SAME replica (seed 42), rank r owns the images of seed 1234 + r
    O.set_seed(42)
    model = O.build_model()
    model.train()
    x, y = O.synthetic_batch(1, 64, seed=1234 + rank)
    loss = O.total_loss(model(x), y)
    loss.backward()
    params = list(model.parameters())
    flat = torch.cat([p.grad.flatten() for p in params])
    local = flat.clone()
    # split the flat buffer into 10 uneven buckets, handed over tail-first like the engine's stages
    n = flat.numel()
    cuts = sorted({0, n} | {int(n * f) for f in (0.02, 0.05, 0.3, 0.5, 0.55, 0.8, 0.9, 0.97, 0.99)})
    ranges = list(zip(cuts[:-1], cuts[1:]))[::-1]
    red = vk.GradientReducer(lambda: flat, world_size=world, scale_grads=True)
    for i, rg in enumerate(ranges):
        red.bucket_ready(i, rg)
    red.finish()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    want = torch.stack(gathered).mean(0)
    ok_avg = torch.allclose(flat, want, rtol=1e-6, atol=1e-9)
    # the three issue policies reduce the same buffer to the same bits; "deferred" sends buckets 0..8 as ONE collective once stage 8 is
    # done and bucket 9 afterwards, "tail" one collective in finish(), "eager" ten
    issued = {}
    for pol in ("eager", "deferred", "tail"):
        fp = local.clone()
        rp = vk.GradientReducer(lambda fp=fp: fp, world_size=world, scale_grads=True, policy=pol, defer_until=9)
        counts = []
        for i, rg in enumerate(ranges):
            rp.bucket_ready(i, rg)
            counts.append(len(rp._handles))
        rp.finish()
        issued[pol] = counts
        ok_avg = ok_avg and torch.equal(fp, flat) and not rp.in_flight
    ok_avg = ok_avg and issued["eager"] == list(range(1, 11)) and issued["deferred"] == [0] * 8 + [1, 2] and issued["tail"] == [0] * 10
    # folded scaling variant: SUM only, optimizer applies 1/world
    flat2 = local.clone()
    red2 = vk.GradientReducer(lambda: flat2, world_size=world, scale_grads=False)
    red2.bucket_ready(0, (0, n))
    red2.finish()
    ok_sum = torch.allclose(flat2 * red2.inv_world, want, rtol=1e-6, atol=1e-9)
    # broadcast of a product model's flat buffers (CPU tensors are fine for the collective itself)
    torch.manual_seed(100 + rank)
    m2 = vk.Unet(encoder_weights=None)
    before = m2.flat_params.clone()
    vk.broadcast_model(m2, 0)
    g2 = [torch.zeros_like(before) for _ in range(world)]
    dist.all_gather(g2, m2.flat_params)
    ok_bc = all(torch.equal(g2[0], t) for t in g2) and (rank == 0 or not torch.equal(before, m2.flat_params))
    # C3: loss / metric scalars for logging (train.py:452-459 logs the mean loss): mean over ranks of [loss, dice, iou]
    sc = vk.all_reduce_scalars(torch.tensor([float(loss), 0.25 * (rank + 1), 1.0]))
    l_all = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(l_all, torch.tensor([float(loss)]))
    ok_bc = ok_bc and abs(sc[0].item() - torch.stack(l_all).mean().item()) < 1e-6 and abs(sc[1].item() - 0.25 * (world + 1) / 2) < 1e-6 \
        and sc[2].item() == 1.0
    q.put((rank, bool(ok_avg), bool(ok_sum), bool(ok_bc), float(loss)))
    dist.destroy_process_group()


def test_gradient_reducer_gloo_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert [r[0] for r in res] == [0, 1]
    assert all(r[1] and r[2] and r[3] for r in res), res
    assert res[0][4] != res[1][4]          # different shards -> different local losses


def test_engine_bucket_table(vk):
    L = vk.lib()
    cfg = vk._lib.vk_unet_config(2, 64, vk._lib.VK_BF16, 1)
    h = C.c_void_p()
    vk._lib.check(L.vk_unet_create(C.byref(cfg), C.byref(h)))
    try:
        nb = L.vk_unet_num_buckets(h)
        assert nb == 10
        P = L.vk_unet_param_numel(h)
        ranges = []
        for b in range(nb):
            b0, b1 = C.c_int64(), C.c_int64()
            vk._lib.check(L.vk_unet_bucket_range(h, b, C.byref(b0), C.byref(b1)))
            ranges.append((b0.value, b1.value))
        # backward completion order == walking the flat buffer from its tail to its head
        assert ranges[0][1] == P and ranges[-1][0] == 0
        for (a0, a1), (b0, b1) in zip(ranges[:-1], ranges[1:]):
            assert b1 == a0 and a0 < a1
        sizes = [b - a for a, b in ranges]
        assert sum(sizes) == P
        # the last-finishing bucket is tiny, layer4 is split in three
        assert sizes[-1] < 300_000 and all(3_500_000 < s < 5_000_000 for s in sizes[3:6])
    finally:
        L.vk_unet_destroy(h)


def test_reducer_stage_groups(vk):
    """Which backward stages may run as ONE vk_unet_backward call (= one weight-gradient batch): exactly those whose gradients the
    policy does not look at before the last of them has finished."""
    flat = torch.zeros(16)

    def groups(**kw):
        r = vk.GradientReducer(lambda: flat, world_size=2, **kw)
        r.enabled = True           # no process group in this test: only the grouping logic is asked
        return r.stage_groups(10)

    assert groups(policy="eager") == [(s, s + 1) for s in range(10)]
    assert groups(policy="deferred", defer_until=9) == [(0, 9), (9, 10)]
    assert groups(policy="deferred", defer_until=4) == [(0, 4)] + [(s, s + 1) for s in range(4, 10)]
    assert groups(policy="deferred", defer_until=0) == [(s, s + 1) for s in range(10)]
    assert groups(policy="deferred", defer_until=99) == [(0, 10)]
    assert groups(policy="tail") == [(0, 10)]
    r = vk.GradientReducer(lambda: flat, world_size=1)
    assert r.stage_groups(10) == [(0, 10)]          # a disabled reducer (one rank) needs nothing early
    # the groups cover every stage once, in order
    for g in (groups(policy="eager"), groups(policy="deferred", defer_until=7), groups(policy="tail")):
        assert [s for a, b in g for s in range(a, b)] == list(range(10))
