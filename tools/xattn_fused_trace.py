"""Dev tool: phase timeline of the fused cross-attention launch (csrc/xattn_fused.hip built with -DTRIS_XF_TRACE into a side
library; run on the GPU box):  python tools/xattn_fused_trace.py"""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = "/tmp/libxf_trace.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DTRIS_XF_TRACE",
                       f"-I{ROOT}/include", f"-I{ROOT}/tris_amd/csrc", f"{ROOT}/tris_amd/csrc/xattn_fused.hip", "-o", so,
                       f"-L{ROOT}/tris_amd", "-l:libtris_hip.so", f"-Wl,-rpath,{ROOT}/tris_amd"])
from tris_amd import _lib
_lib.load()
lib = ctypes.CDLL(so)
B, P, N, C = 48, 100, 48, 1024
g = torch.Generator().manual_seed(0)
Qv, Kv, Vv = (torch.randn(B, P, C, generator=g).cuda() for _ in range(3))
Qt, Kt, Vt = (torch.randn(N, C, generator=g).cuda() for _ in range(3))
nv = torch.empty(B, P, C, device="cuda"); nl = torch.empty(B, N, C, device="cuda"); probs = torch.empty(B, 4, P, N, device="cuda")
lib.tris_xattn_fused_ws_bytes.restype = ctypes.c_long
wsb = lib.tris_xattn_fused_ws_bytes(B, N, C)
ws = torch.zeros(wsb // 4 + 4, device="cuda"); sync = torch.zeros(16 + 16 * B, dtype=torch.int32, device="cuda")
V = ctypes.c_void_p
def run():
    rc = lib.tris_xattn_fused_fwd_f32(V(Qv.data_ptr()), V(Kv.data_ptr()), V(Vv.data_ptr()), V(Qt.data_ptr()), V(Kt.data_ptr()),
                                      V(Vt.data_ptr()), V(nv.data_ptr()), V(nl.data_ptr()), V(probs.data_ptr()), B, P, N, C,
                                      V(ws.data_ptr()), ctypes.c_long(ws.numel() * 4), V(sync.data_ptr()),
                                      V(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
for _ in range(5):
    run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    run()
b.record(); torch.cuda.synchronize()
print(f"prep + fused: {a.elapsed_time(b) / 20 * 1e3:.1f} us per call")
off = (wsb - B * 8 * 16 * 8) // 8
raw = ws[:wsb // 4].view(torch.int64)[off:off + B * 8 * 16].view(B * 8, 16).cpu().double()
order = [0, 1, 2, 3, 10, 11, 12, 13, 4, 5, 6]     # stamp ids in program order
tr = raw[:, order]
names = ["start", "logits done", "published", "stage-1 flags", "reduced", "soft-max done", "published 2", "stage-2 flags", "gathered",
         "new_vis done", "end"]
clk = 2.0e9   # s_memtime ticks at the shader clock (~2 GHz under load: 77 us per call = 1.55e5 ticks); per-XCD counters are not aligned -> per-workgroup deltas
d = (tr[:, 1:] - tr[:, :-1]) / clk * 1e6
for i in range(len(names) - 1):
    col = d[:, i]
    print(f"{names[i]:13s} -> {names[i + 1]:13s}  min {col.min():7.2f}  median {col.median():7.2f}  max {col.max():7.2f} us")
tot = (tr[:, -1] - tr[:, 0]) / clk * 1e6
print(f"workgroup lifetime  min {tot.min():.2f}  median {tot.median():.2f}  max {tot.max():.2f} us")
