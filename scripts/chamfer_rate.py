import importlib, sys, time, torch
sys.path.insert(0, "/root/repo")
ops = importlib.import_module("2dimageto3dmodel_amd.ops")
for B in (1, 2, 4, 8, 16):
    N = 16384
    a = torch.rand(B, N, 3, device="cuda") - 0.5; b = torch.rand(B, N, 3, device="cuda") - 0.5
    for _ in range(3): ops.chamfer_nn(a, b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ops.chamfer_nn(a, b)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(B, f"{dt*1e3:.3f} ms  {8.0*B*N*N/dt/1e12:.1f} TF")
