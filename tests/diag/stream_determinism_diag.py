"""Diagnostic: run-to-run determinism of the streaming convolution kernels (forward, with and without statistics)."""
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
def gen(*shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))
for (name, Cin, up, K) in [("dec4_conv1", 32, 1, 16), ("dec4_conv2", 16, 0, 16), ("dec3_conv2", 32, 0, 32)]:
    for (N, H, W) in [(1, 80, 80), (1, 160, 160), (2, 64, 64), (2, 72, 40)]:
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
        for with_stats in (0, 1):
            outs = []
            for i in range(6):
                y = torch.full((N, H, W, K), float("nan"), device=dev, dtype=dt)
                stats = torch.zeros(32 * 2 * K, dtype=torch.float64, device=dev)
                L_.check(fn(C.byref(d), wp.data_ptr(), y.data_ptr(), None, 0, 0, stats.data_ptr() if with_stats else None, st))
                torch.cuda.synchronize()
                outs.append(y)
            bad = [i for i in range(1, 6) if not torch.equal(outs[i], outs[0])]
            nan = int(torch.isnan(outs[0].float()).sum().item())
            os.environ["VK_NO_STREAM"] = "1"
            yt = torch.full((N, H, W, K), float("nan"), device=dev, dtype=dt)
            L_.check(fn(C.byref(d), wp.data_ptr(), yt.data_ptr(), None, 0, 0, None, st))
            torch.cuda.synchronize()
            os.environ.pop("VK_NO_STREAM")
            dmax = (outs[0].float() - yt.float()).abs().max().item()
            print(f"{name} N{N} {H}x{W} stats={with_stats}: runs differing from run 0: {bad}; NaNs {nan}; max |stream - tile| {dmax:.4g}", flush=True)
