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
# ---- where the time between the median workgroup and the whole launch goes -------------------------------
# (s_memtime counters are not aligned across the chip -- neither per XCD nor in a way a start stamp reveals --, so only
# per-workgroup DELTAS are meaningful: the launch ramp and the drain are the difference between the longest workgroup lifetime
# and the event-timed duration of the call printed at the top)
# the slowest workgroup's own timeline, and the late starters (second workgroup of a doubly loaded CU starts when? -- with 384
# workgroups on 256 CUs all are co-resident, so a late start is dispatch latency, not queueing)
w = int(tot.argmax())
print(f"slowest workgroup {w} (image {w // 8}, slice {w % 8}, XCD {w % 8}): " + " | ".join(f"{names[i + 1]} {float(d[w, i]):.1f}" for i in range(len(names) - 1)))
# waits: time spent polling flags = the skew between the eight workgroups of an image
print(f"wait phases (stage-1 flags + stage-2 flags): median {float((d[:, 2] + d[:, 6]).median()):.2f}  max {float((d[:, 2] + d[:, 6]).max()):.2f} us")
# bytes a workgroup moves per phase (P = 100, N = 48, C = 1024, 8 slices) and the per-CU rate they imply
KB = 1024.0
phases = [("logits (Qv, Kv slices: 2 x P x C/8 x 4 B)", 2 * P * C / 8 * 4 / KB, 0), ("publish partial logits (write-through)", 43.0, 1),
          ("reduce 1/8 of the rows (sc1 loads)", 46.0, 3), ("publish probabilities", 5.0, 5), ("gather At, Av", 50.0, 7),
          ("new_vis (Vt planes from L2, store P x C/8)", P * C / 8 * 4 / KB, 8), ("new_lan (Vv slice read + N x C/8 store)", (P + N) * C / 8 * 4 / KB, 9)]
for name, kb, i in phases:
    us = float(d[:, i].median())
    print(f"  {name:52s} {kb:6.1f} KB  median {us:5.2f} us  -> {kb * KB / (us * 1e-6) / 1e9:6.1f} GB/s per workgroup")
