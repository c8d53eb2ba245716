"""Dev tool: phase timeline of the pixel-row cross-attention launch (csrc/xattn_px.hip built with -DTRIS_XP_TRACE into a side
library; run on the GPU box):  python tools/xattn_px_trace.py [B P N C]"""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = "/tmp/libxp_trace.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DTRIS_XP_TRACE", *[f"-D{x}" for x in os.environ.get("XP_DEFS", "").split()], *os.environ.get("XP_FLAGS", "").split(),
                       f"-I{ROOT}/include", f"-I{ROOT}/tris_amd/csrc", f"{ROOT}/tris_amd/csrc/xattn_px.hip", "-o", so,
                       f"-L{ROOT}/tris_amd", "-l:libtris_hip.so", f"-Wl,-rpath,{ROOT}/tris_amd"])
from tris_amd import _lib
_lib.load()
lib = ctypes.CDLL(so)
B, P, N, C = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (48, 100, 48, 1024)
g = torch.Generator().manual_seed(0)
Qv, Kv, Vv = (torch.randn(B, P, C, generator=g).cuda() for _ in range(3))
Qt, Kt, Vt = (torch.randn(N, C, generator=g).cuda() for _ in range(3))
nv = torch.empty(B, P, C, device="cuda"); nl = torch.empty(B, N, C, device="cuda"); probs = torch.empty(B, 4, P, N, device="cuda")
lib.tris_xattn_px_ws_bytes.restype = ctypes.c_long
wsb = lib.tris_xattn_px_ws_bytes(B, N, C)
ws = torch.zeros(wsb // 4 + 4, device="cuda"); sync = torch.zeros(16 + 16 * B, dtype=torch.int32, device="cuda")
V = ctypes.c_void_p
H2 = os.environ.get("XP_H2", "1") != "0"     # the h2 form (two fp16 pieces) unless XP_H2=0
words = []
if H2:
    main = _lib.load()
    for t in (Qv, Kv, Vv, Qt, Kt, Vt):
        w = torch.zeros(2048, dtype=torch.int32, device="cuda")
        main.tris_amax_bits_f32(V(t.data_ptr()), ctypes.c_long(t.numel()), V(w.data_ptr()), V(torch.cuda.current_stream().cuda_stream))
        words.append(w)
print("arithmetic:", "h2 (two fp16 pieces, three MFMAs per product)" if H2 else "x3 (three bf16 pieces, six MFMAs per product)")
def run():
    if H2:
        assert lib.tris_xattn_amax_next(*[V(w.data_ptr()) for w in words]) == 0
    rc = lib.tris_xattn_px_fwd_f32(V(Qv.data_ptr()), V(Kv.data_ptr()), V(Vv.data_ptr()), V(Qt.data_ptr()), V(Kt.data_ptr()),
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
print(f"prep + px launch (trace build): {a.elapsed_time(b) / 20 * 1e3:.1f} us per call; time-out word {int(sync[2])}")
S = min(8, 256 // B, P)
NT = (N + 15) // 16
off = (wsb - 64 - B * 8 * 32 * 8) // 8   # (the six h2 scales sit behind the trace region)
raw = ws[:wsb // 4].view(torch.int64)[off:off + B * S * 32].view(B * S, 32).cpu().double()
# stamp ids (thread 0 = first wave of the sentence->pixel half T, thread 256 = first wave of the pixel->sentence half V):
# 0 start | 1 logits done (T) | 2 flag raised | 5 flags seen | 6 gathered | 7 At soft-max done | 10 At planes written |
# V: 3 row soft-max done, 9 new_vis done | 4 both halves joined | 8 end
clk = 2.0e9
us = lambda a, b: (raw[:, b] - raw[:, a]) / clk * 1e6
rows = [("start -> logits done", 0, 1), ("T: logits done -> flag raised (reduce, publish, drain)", 1, 2), ("T: flag raised -> flags seen", 2, 5),
        ("T: gather", 5, 6), ("T: At soft-max", 6, 7), ("T: At planes, probs", 7, 10), ("T: logits done -> At planes written", 1, 10),
        ("V: logits done (T) -> row soft-max done", 1, 3), ("V: new_vis", 3, 9), ("V: logits done (T) -> new_vis done", 1, 9),
        ("T: At planes written -> joined", 10, 4), ("new_lan", 4, 8), ("workgroup lifetime", 0, 8)]
for name, i, j in rows:
    col = us(i, j)
    print(f"{name:56s} min {col.min():7.2f}  median {col.median():7.2f}  max {col.max():7.2f} us")
print(f"({B * S} workgroups, S = {S})")
ld = (raw[:, 16:24] - raw[:, 0:1]) / clk * 1e6      # per-wave end of the logits loop, from the workgroup's start stamp
print("per-wave end of the logits loop (us from start), median over workgroups: " + " ".join(f"{float(ld[:, w].median()):.2f}" for w in range(8)))
jn = (raw[:, 24:32] - raw[:, 0:1]) / clk * 1e6
print("per-wave arrival at the join (us from start), median:                    " + " ".join(f"{float(jn[:, w].median()):.2f}" for w in range(8)))
print(f"V: reduce done (after its first group sync) from T's logits-done stamp: median {float(us(1, 12).median()):.2f} us")
