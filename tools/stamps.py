"""Diagnostic: per-segment cycle shares of the halo conv kernel (needs `make -C .../csrc stamp`).
   VK_LIB=vickers-hardness-unet_amd/libvkunet_stamp.so python tools/stamps.py [L3]"""
import ctypes as C, importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
vk = importlib.import_module("vickers-hardness-unet_amd")
L_ = vk._lib
import tools.microbench as mb
name = sys.argv[1] if len(sys.argv) > 1 else "L3"
H, srcs, K = mb.LAYERS[name]
dev = torch.device("cuda:0"); dt = torch.bfloat16; N = 32
lib = vk.lib(); st = torch.cuda.current_stream().cuda_stream
Ctot = sum(c for c, _ in srcs)
ts, ss = [], []
for c, up in srcs:
    t = torch.randn(N, H >> up, H >> up, c, device=dev).to(dt); sc = torch.rand(c, device=dev) + 0.5; sh = torch.randn(c, device=dev) * 0.1
    ts.append((t, sc, sh)); ss.append(L_.vk_src(t.data_ptr(), c, up, sc.data_ptr(), sh.data_ptr(), 1))
s1 = ss[1] if len(ss) > 1 else L_.vk_src(None, 0, 0, None, None, 0)
w = (torch.randn(K, 3, 3, Ctot, device=dev) * 0.05).to(dt)
y = torch.empty(N, H, H, K, device=dev, dtype=dt)
stats = torch.zeros(32 * 2 * K, dtype=torch.float64, device=dev)
d = L_.vk_conv_desc(L_.dtype_code(dt), N, H, H, H, H, K, 3, 3, 1, 1, 0, ss[0], s1)
buf = torch.zeros(8 * 4 * 200000, dtype=torch.int64, device=dev)
lib.vk_debug_set_stamp_buffer(buf.data_ptr())
pk = torch.empty_like(w)
L_.check(lib.vk_halo_pack(L_.dtype_code(dt), K, Ctot, w.data_ptr(), pk.data_ptr(), st))
for _ in range(3):
    L_.check(lib.vk_conv_fwd_packed(C.byref(d), pk.data_ptr(), y.data_ptr(), None, 0, 0, stats.data_ptr(), st))
torch.cuda.synchronize()
b = buf.view(-1, 8).cpu().double()
b = b[b[:, 7] > 0]
m = b.mean(0)
nst = int(m[7])
mf = 48 * 16          # MFMA cycles per wave per stage at TP = TC = 4
print(f"{name}: waves={len(b)} stages={nst}  per-wave cycles: prologue {m[5]:.0f}  [dma+halo-issue {m[0]:.0f}  compute {m[1]:.0f}  halo-store {m[2]:.0f}  barrier {m[3]:.0f}]  epilogue {m[4]:.0f}  kernel-body {m[6]:.0f}")
tot = m[0] + m[1] + m[2] + m[3]
print(f"   stage loop {tot:.0f} cycles = {tot / nst:.0f} per stage: issue {m[0]/tot:.1%} compute {m[1]/tot:.1%} store {m[2]/tot:.1%} barrier {m[3]/tot:.1%};  own MFMA cycles per stage = {mf}")
print(f"   kernel body: prologue {m[5]/m[6]:.1%}  loop {tot/m[6]:.1%}  epilogue {m[4]/m[6]:.1%}")
