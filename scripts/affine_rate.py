"""affine_act_fwd (the fused CBN apply) plain and with the residual read through the nearest x2 upsample: TB/s of the bytes moved."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
L = importlib.import_module("2dimageto3dmodel_amd._lib")
def rate(name, f, nbytes, n=200):
    for _ in range(10): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); us = e0.elapsed_time(e1) / n * 1e3
    print(f"{name:64s} {us:7.1f} us  {nbytes / us / 1e6:5.2f} TB/s", flush=True)
for (N, H, W, C) in ((64, 256, 128, 64), (64, 128, 64, 128), (64, 64, 32, 128), (64, 32, 16, 256)):
    x = torch.randn(N, H, W, C, device="cuda").bfloat16(); y = torch.empty_like(x)
    a = torch.rand(N, C, device="cuda"); b = torch.rand(N, C, device="cuda")
    res_full = torch.randn(N, H, W, C, device="cuda").bfloat16(); res_half = torch.randn(N, H // 2, W // 2, C, device="cuda").bfloat16()
    nb = x.numel() * 2
    rate(f"[{N},{H},{W},{C}] plain", lambda: L.launch("affine_act_fwd", L.ptr(x), L.ptr(a), L.ptr(b), None, 0, L.ptr(y), N, H * W, C, 0.2, 1.0, L.stream()), 2 * nb)
    rate(f"[{N},{H},{W},{C}] + residual (same resolution)", lambda: L.launch("affine_act_fwd", L.ptr(x), L.ptr(a), L.ptr(b), L.ptr(res_full), 0, L.ptr(y), N, H * W, C, 0.2, 1.0, L.stream()), 3 * nb)
    rate(f"[{N},{H},{W},{C}] + residual through the x2 upsample, out_slope", lambda: L.launch("affine_act_fwd", L.ptr(x), L.ptr(a), L.ptr(b), L.ptr(res_half), W, L.ptr(y), N, H * W, C, 0.2, 0.2, L.stream()), 2 * nb + nb // 4)
