"""Chamfer nearest-neighbour rate, N = M = 16384: (a) host clock over 10 back-to-back calls (includes the Python side of a call:
at B = 1 the host is the limit), (b) the same 20 calls captured in a hipGraph and replayed: device time incl. the gaps between the
launches, (c) HIP events around each entry point (kernel time only)."""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("2dimageto3dmodel_amd")
ops = importlib.import_module("2dimageto3dmodel_amd.ops")
N = 16384
for B in (1, 2, 4, 8, 16):
    a = torch.rand(B, N, 3, device="cuda") - 0.5; b = torch.rand(B, N, 3, device="cuda") - 0.5
    for _ in range(3): ops.chamfer_nn(a, b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ops.chamfer_nn(a, b)
    torch.cuda.synchronize(); host = (time.perf_counter() - t0) / 10
    # graph replay
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): ops.chamfer_nn(a, b)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): out = ops.chamfer_nn(a, b)
    g.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); dev = (time.perf_counter() - t0) / 100
    pkg._lib.enable_kernel_timers(True)
    for _ in range(10): ops.chamfer_nn(a, b)
    torch.cuda.synchronize()
    kt = pkg._lib.collect_kernel_timers(); pkg._lib.enable_kernel_timers(False)
    ker = sum(v[1] for v in kt.values()) / 10 * 1e-3
    fl = 8.0 * B * N * N
    print(f"B {B:2d}: host clock {host*1e6:7.1f} us {fl/host/1e12:5.1f} TF | graph replay {dev*1e6:7.1f} us {fl/dev/1e12:5.1f} TF | HIP events {ker*1e6:7.1f} us {fl/ker/1e12:5.1f} TF  "
          + " ".join(f"{k}:{v[1]/v[0]*1e3:.1f}us" for k, v in kt.items()))
