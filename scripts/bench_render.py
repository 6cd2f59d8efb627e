"""Times Renderer.forward / backward (csrc/dibr_raster.hip) on a deformed UV-sphere template: B meshes, H x W pixels."""
import importlib
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
R = importlib.import_module("2dimageto3dmodel_amd.render")
M = importlib.import_module("2dimageto3dmodel_amd.mesh")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 256
with tempfile.TemporaryDirectory() as tmp:
    for seg, rings in ((32, 16), (64, 31)):
        tpl = M.MeshTemplate(M.write_uv_sphere_obj(os.path.join(tmp, f"s{rings}.obj"), segments=seg, rings=rings), is_symmetric=True,
                             device="cuda")
        dmap = (0.02 * torch.randn(B, 3, 32, 32, device="cuda")).requires_grad_()
        tex = torch.rand(B, 3, 256, 128, device="cuda", requires_grad=True)
        ren = R.Renderer(H, W)

        def step(bwd):
            vtx = tpl.get_vertex_positions(dmap) * 0.8 - torch.tensor([0.0, 0.0, 2.0], device="cuda")
            img, alpha = tpl.forward_renderer(ren, vtx, tex)
            if bwd:
                dmap.grad = tex.grad = None
                (img.sum() + alpha.sum()).backward()
            return alpha

        for bwd in (False, True):
            for _ in range(3):
                a = step(bwd)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 10
            for _ in range(n):
                a = step(bwd)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            print(f"faces {tpl.mesh.faces.shape[0]:5d}  B {B}  {H}x{W}  {'fwd+bwd' if bwd else 'fwd    '} {dt*1e3:7.3f} ms  "
                  f"{B*H*W/dt/1e9:6.2f} Gpix/s  coverage {float((a == 1).float().mean()):.2f}", flush=True)
lib = importlib.import_module("2dimageto3dmodel_amd._lib")
lib.enable_kernel_timers(True)
step(True)
torch.cuda.synchronize()
for k, v in sorted(lib.collect_kernel_timers().items(), key=lambda kv: -kv[1][1])[:8]:
    print(f"   {k:24s} {v[1]*1e3:8.1f} us")
