"""Diagnostic: where the staggered K >= 128 tile kernel (conv3x3_cols_kernel) spends its slots, early waves (0-3) and late waves (4-7)
separately (needs `make -C vickers-hardness-unet_amd/csrc stamp`).
   VK_COL_PIPE=2 VK_LIB=vickers-hardness-unet_amd/libvkunet_stamp.so python tools/stamps_cols.py [L3]"""
import ctypes as C, importlib, os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
os.environ["VK_COL_PIPE"] = "2"
vk = importlib.import_module("vickers-hardness-unet_amd")
L_ = vk._lib
import tools.microbench as mb
dev = torch.device("cuda:0"); dt = torch.bfloat16; N = 32
lib = vk.lib(); st = torch.cuda.current_stream().cuda_stream
for name in (sys.argv[1:] or ["L2", "L3", "L4", "D0c1"]):
    H, srcs, K = mb.LAYERS[name]
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
    buf = torch.zeros(8 * 8 * 100000, dtype=torch.int64, device=dev)
    lib.vk_debug_set_stamp_buffer(buf.data_ptr())
    pk = torch.empty_like(w)
    L_.check(lib.vk_halo_pack(L_.dtype_code(dt), K, Ctot, w.data_ptr(), pk.data_ptr(), st))
    for _ in range(3):
        L_.check(lib.vk_conv_fwd_packed(C.byref(d), pk.data_ptr(), y.data_ptr(), None, 0, 0, stats.data_ptr(), st))
    torch.cuda.synchronize()
    b = buf.view(-1, 8, 8).cpu().double()          # [workgroup][wave][field]
    b = b[b[:, 0, 7] > 0]
    nst = int(b[0, 0, 7])
    for grp, sl in (("early waves 0-3", slice(0, 4)), ("late  waves 4-7", slice(4, 8))):
        m = b[:, sl, :].reshape(-1, 8).mean(0)
        loop = m[0] + m[1] + m[2] + m[3] + m[4]
        print(f"{name} {grp}: stages={nst} per stage: X(MFMA) {m[0] / nst:6.0f}  wait-vm {m[1] / nst:5.0f}  barrier1 {m[2] / nst:5.0f}  Y(work) {m[3] / nst:6.0f}  barrier2 {m[4] / nst:5.0f}"
              f"  = {loop / nst:6.0f} cycles (own MFMA 768);  prologue {m[5]:.0f}  body {m[6]:.0f}")
