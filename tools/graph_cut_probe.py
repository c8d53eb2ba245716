"""What a CUT in a chain of hipGraphs costs on the device: the same 240 small kernels replayed as 1 graph, as 24 graphs of 10 and
as 240 eager launches on one stream; and the same with an EXTERNAL event record node (torch.cuda.Event(external=True)) every 10
kernels that a second stream waits on before launching its own graph (the shape of the weight-gradient release).
    python tools/graph_cut_probe.py            (GPU)"""
import time
import torch


def main():
    dev = torch.device("cuda:0")
    x = torch.zeros(1 << 20, device=dev)          # 4 MB: ~5 us per pass
    big = torch.zeros(16 << 20, device=dev)       # 64 MB: ~40 us per pass
    s = torch.cuda.Stream()
    s2 = torch.cuda.Stream()

    def body(n, t):
        for _ in range(n):
            t.add_(1.0)

    def capture(n, t, stream):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(stream):
            with torch.cuda.graph(g, stream=stream, capture_error_mode="relaxed"):
                body(n, t)
        return g

    def timed(fn, reps=20):
        with torch.cuda.stream(s):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3

    for name, t in (("4MB", x), ("64MB", big)):
        with torch.cuda.stream(s):
            body(3, t)
        torch.cuda.synchronize()
        g1 = capture(240, t, s)
        g24 = [capture(10, t, s) for _ in range(24)]
        g240 = [capture(1, t, s) for _ in range(240)]
        print(f"{name}: one graph of 240            {timed(lambda: g1.replay()):8.3f} ms")
        print(f"{name}: 24 graphs of 10             {timed(lambda: [g.replay() for g in g24]):8.3f} ms")
        print(f"{name}: 240 graphs of 1             {timed(lambda: [g.replay() for g in g240]):8.3f} ms")
        print(f"{name}: 240 eager launches          {timed(lambda: body(240, t)):8.3f} ms")
    # external event record nodes inside ONE graph, a second stream released by each (torch refuses Event(external=True) on ROCm;
    # the runtime call itself is there: hipEventRecordWithFlags(ev, stream, hipEventRecordExternal))
    import ctypes
    path = next((l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l), "libamdhip64.so")   # the runtime torch loaded
    print("HIP runtime:", path)
    hip = ctypes.CDLL(path)
    hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
    hip.hipEventRecordWithFlags.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
    hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]

    def new_event():
        h = ctypes.c_void_p()
        assert hip.hipEventCreateWithFlags(ctypes.byref(h), 0x2) == 0
        return h

    def rec_ext(h, stream):
        e = hip.hipEventRecordWithFlags(h, ctypes.c_void_p(stream.cuda_stream), 1)
        assert e == 0, f"hipEventRecordWithFlags -> {e}"

    def wait(stream, h):
        e = hip.hipStreamWaitEvent(ctypes.c_void_p(stream.cuda_stream), h, 0)
        assert e == 0, f"hipStreamWaitEvent -> {e}"

    evs = [new_event() for _ in range(24)]
    y = torch.zeros(1 << 20, device=dev)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s, capture_error_mode="relaxed"):
                for i in range(24):
                    body(10, x)
                    rec_ext(evs[i], s)
    except AssertionError as e:
        print("external event record inside a capture is refused by this runtime:", e)
        return
    side = [capture(4, y, s2) for _ in range(24)]

    def run_ext():
        g.replay()
        for i in range(24):
            wait(s2, evs[i])
            with torch.cuda.stream(s2):
                side[i].replay()
        s.wait_stream(s2)

    def run_cut():
        for i in range(24):
            g24x[i].replay()
            e = torch.cuda.Event()
            e.record(s)
            s2.wait_event(e)
            with torch.cuda.stream(s2):
                side[i].replay()
        s.wait_stream(s2)
    g24x = [capture(10, x, s) for _ in range(24)]
    print(f"main 1 graph + 24 external records, side 24 graphs of 4   {timed(run_ext):8.3f} ms")
    print(f"main 24 graphs + 24 eager records,  side 24 graphs of 4   {timed(run_cut):8.3f} ms")
    # does the side stream really wait?  main: long kernels; side graph reads what main wrote
    a = torch.zeros(64 << 20, device=dev)
    out = torch.zeros(1, device=dev)
    ev = new_event()
    gm = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(gm, stream=s, capture_error_mode="relaxed"):
            for _ in range(20):
                a.add_(1.0)
            rec_ext(ev, s)
            for _ in range(20):
                a.add_(1.0)
    gs = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s2):
        with torch.cuda.graph(gs, stream=s2, capture_error_mode="relaxed"):
            out.copy_(a[:1])
    bad = 0
    for rep in range(10):
        with torch.cuda.stream(s):
            a.zero_()
            s2.wait_stream(s)
            gm.replay()
        wait(s2, ev)
        with torch.cuda.stream(s2):
            gs.replay()
        torch.cuda.synchronize()
        v = float(out.item())
        if not (20.0 <= v <= 40.0):
            bad += 1
        if rep < 3:
            print(f"side stream saw a[0] = {v} (20 = released exactly at the record node, < 20 = did not wait)")
    print("ordering violations:", bad)


if __name__ == "__main__":
    main()
