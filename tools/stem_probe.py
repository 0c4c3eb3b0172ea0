import ctypes as C, importlib, sys, torch
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
vk = importlib.import_module("vickers-hardness-unet_amd")
L_ = vk._lib
lib = vk.lib()
dev = torch.device("cuda:0"); dt = torch.bfloat16
N, H = 32, 512
x4 = torch.randn(N, H, H, 4, device=dev).to(dt)
wp = (torch.randn(64 * 7 * 32, device=dev) * 0.05).to(dt)
y = torch.empty(N, H // 2, H // 2, 64, device=dev, dtype=dt)
stats = torch.zeros(32 * 2 * 64, dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run(s):
    L_.check(lib.vk_stem_fwd(L_.dtype_code(dt), N, H, H, x4.data_ptr(), wp.data_ptr(), y.data_ptr(), s, st))
for name, s in (("with stats", stats.data_ptr()), ("no stats", None)):
    for _ in range(3): run(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run(s)
    e1.record(); torch.cuda.synchronize()
    print(name, e0.elapsed_time(e1) / 20 * 1e3, "us")
