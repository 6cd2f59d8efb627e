"""Ablation timing of the render kernels: builds variants with -DM355_ABL=n on the GPU box and times them."""
import ctypes, os, subprocess, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(ROOT, "2dimageto3dmodel_amd", "csrc")
def build(abl):
    out = f"/tmp/libabl{abl}.so"
    srcs = [os.path.join(C, f) for f in ("error.cpp", "proj_transform.hip", "proj_render.hip", "proj_render21.hip", "sil_loss.hip")]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           f"-DM355_ABL={abl}", "-o", out] + srcs
    subprocess.check_call(cmd)
    return ctypes.CDLL(out)
B, N, S = 64, 2048, 128
g = torch.Generator().manual_seed(0)
pc = ((torch.rand(B, N, 3, generator=g) - 0.5) * 0.7).cuda()
q = torch.randn(B, 4, generator=g).cuda()
sc = torch.sigmoid(torch.randn(B, generator=g)).cuda()
cam = torch.empty_like(pc); key = torch.empty(B, N, dtype=torch.int32, device="cuda")
tstart = torch.empty(B, 257, dtype=torch.int32, device="cuda"); tpts = torch.empty(B, 4 * N, 4, device="cuda")
proj = torch.empty(B, S, S, device="cuda"); dproj = torch.rand(B, S, S, device="cuda")
slots = torch.empty(B, N, 4, 3, device="cuda"); dsp = torch.empty(B, 256, device="cuda")
taps = torch.full((21,), 1 / 21.0, device="cuda")
P = ctypes.c_void_p
for abl in [int(x) for x in sys.argv[1:]] or [0, 1, 2]:
    L = build(abl)
    st = P(torch.cuda.current_stream().cuda_stream)
    def binf():
        L.m355_proj_bin_fwd(P(pc.data_ptr()), P(q.data_ptr()), P(cam.data_ptr()), None, P(tstart.data_ptr()), P(tpts.data_ptr()), B, N, S, ctypes.c_float(1.875), ctypes.c_float(2.0), st)
    binf()
    def fwd():
        L.m355_proj_render_fwd(P(tstart.data_ptr()), P(tpts.data_ptr()), P(sc.data_ptr()), P(taps.data_ptr()), 21, P(proj.data_ptr()), B, N, S, 0, st)
    def bwd():
        L.m355_proj_render_bwd(P(tstart.data_ptr()), P(tpts.data_ptr()), P(sc.data_ptr()), P(taps.data_ptr()), 21, P(dproj.data_ptr()), ctypes.c_float(1.0), P(slots.data_ptr()), P(dsp.data_ptr()), B, N, S, 0, st)
    for name, f in (("bin", binf), ("fwd", fwd), ("bwd", bwd)):
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        print(f"ABL={abl} {name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
