"""Diagnostic (r04): WHERE the reconstructed r03 streaming-kernel bug puts its wrong values.

Run with VK_LIB pointing at a diagnostic build (tests/diag/build_stream_variants.sh).  For the 32 -> 32 forward without a
statistics pointer (the failing launch class) and its siblings: six runs each, compared with the tile kernel's result of the same
library (VK_NO_STREAM=1); prints how many elements differ per run, which output rows / strips / columns they sit in, and whether
the wrong value equals the value of another row (a stale ring slot) or nothing recognisable."""
import ctypes as C, importlib, os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
vk = importlib.import_module("vickers-hardness-unet_amd")
L_ = vk._lib
lib = vk.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
dt = torch.bfloat16
print("library:", L_.LIB_PATH)


def gen(*shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


total_bad = 0
for (name, Cin, up, K) in [("dec3_conv2", 32, 0, 32), ("dec4_conv1", 32, 1, 16), ("dec4_conv2", 16, 0, 16)]:
    for (N, H, W) in [(1, 80, 80), (1, 160, 160), (2, 64, 64), (2, 72, 40), (4, 256, 256)]:
        Hs, Wsrc = (H // 2, W // 2) if up else (H, W)
        x = gen(N, Hs, Wsrc, Cin, seed=1).to(dt).to(dev)
        sc = (torch.rand(Cin, generator=torch.Generator().manual_seed(2)) + 0.5).to(dev)
        sh = (gen(Cin, seed=3) * 0.1).to(dev)
        w = (gen(K, 3, 3, Cin, seed=4) * 0.05).to(dt).to(dev)
        none = L_.vk_src(None, 0, 0, None, None, 0)
        d = L_.vk_conv_desc(L_.dtype_code(dt), N, H, W, H, W, K, 3, 3, 1, 1, 0, L_.vk_src(x.data_ptr(), Cin, up, sc.data_ptr(), sh.data_ptr(), 1), none)
        packed = lib.vk_conv_uses_halo_pack(C.byref(d)) != 0
        wp = torch.empty_like(w)
        if packed:
            L_.check(lib.vk_halo_pack(L_.dtype_code(dt), K, Cin, w.data_ptr(), wp.data_ptr(), st))
        else:
            wp.copy_(w)
        fn = lib.vk_conv_fwd_packed if packed else lib.vk_conv_fwd
        os.environ["VK_NO_STREAM"] = "1"
        yt = torch.full((N, H, W, K), float("nan"), device=dev, dtype=dt)
        L_.check(fn(C.byref(d), wp.data_ptr(), yt.data_ptr(), None, 0, 0, None, st))
        torch.cuda.synchronize()
        os.environ.pop("VK_NO_STREAM")
        for with_stats in (0, 1):
            for i in range(6):
                y = torch.full((N, H, W, K), float("nan"), device=dev, dtype=dt)
                stats = torch.zeros(32 * 2 * K, dtype=torch.float64, device=dev)
                L_.check(fn(C.byref(d), wp.data_ptr(), y.data_ptr(), None, 0, 0, stats.data_ptr() if with_stats else None, st))
                torch.cuda.synchronize()
                # the stream kernel differs from the tile kernel by fp32 summation order at C = 32 (a last-bit flip of the stored value):
                # "wrong" = off by more than 4 bf16 ulps of the larger magnitude, or NaN
                a, b = y.float(), yt.float()
                tol = 4 * 2.0 ** -8 * torch.maximum(a.abs(), b.abs()) + 1e-6
                bad = ~((a - b).abs() <= tol)
                nb = int(bad.sum().item())
                if nb == 0:
                    continue
                total_bad += nb
                idx = bad.nonzero()
                rows = sorted(set(idx[:, 1].tolist()))
                cols = sorted(set(idx[:, 2].tolist()))
                imgs = sorted(set(idx[:, 0].tolist()))
                chans = sorted(set(idx[:, 3].tolist()))
                print(f"{name} N{N} {H}x{W} stats={with_stats} run {i}: {nb} wrong elements; images {imgs}; rows {rows[:12]}{'...' if len(rows) > 12 else ''} (all = 3 mod 4: {all(r % 4 == 3 for r in rows)}); "
                      f"cols {cols[:8]}..{cols[-1]} ({len(cols)} distinct); channels {chans}; NaN {int(torch.isnan(a).sum())}", flush=True)
                # is a wrong row the correct result of a NEIGHBOURING row (stale ring slot)?  compare the first wrong pixel's channel vector with rows y-4..y+4
                n0, y0, x0, _ = idx[0].tolist()
                v = a[n0, y0, x0]
                match = [dy for dy in range(-4, 5) if 0 <= y0 + dy < H and torch.equal(v, b[n0, y0 + dy, x0])]
                print(f"    first wrong pixel (n={n0}, y={y0}, x={x0}): equals the tile result of rows y+{match} (empty = no neighbouring row)", flush=True)
print("TOTAL wrong elements over all cases:", total_bad)
