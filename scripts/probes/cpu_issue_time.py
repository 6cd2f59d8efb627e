"""Round 6 probe: is the eager GAN cycle CPU-bound?  Time for the Python loop to ISSUE K cycles (it returns when the last launch is queued)
against the time until the GPU has finished them.  python scripts/probes/cpu_issue_time.py [batch] [cycles]"""
import argparse, importlib, os, sys, tempfile, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import bench  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
train = importlib.import_module("2dimageto3dmodel_amd.train"); mesh_mod = importlib.import_module("2dimageto3dmodel_amd.mesh")
gargs = argparse.Namespace(norm_g="syncbatch", norm_d="none", conditional_class=True, conditional_color=False, conditional_text=False,
                           n_classes=[200], texture_resolution=256, mask_output=True, num_discriminators=2, texture_only=False, text_embedding_dim=256)
torch.manual_seed(1237)
with tempfile.TemporaryDirectory() as tmp:
    template = mesh_mod.MeshTemplate(mesh_mod.write_uv_sphere_obj(os.path.join(tmp, "uv.obj")), is_symmetric=True, device="cuda")
tr = train.GanTrainer(gargs, device="cuda", mesh_template=template); tr.train(); tr.epoch = 0
batches = [bench.make_textures(B, 256, 1237 + i, "cuda") for i in range(3)]
def cycles(k):
    for _ in range(k):
        for b in batches:
            tr.iteration(*b)
cycles(3); tr.finish_pending(); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); cycles(K); t1 = time.perf_counter(); tr.finish_pending(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"batch {B}: {K} cycles issued in {(t1 - t0) / K * 1e3:.2f} ms per cycle (CPU), finished in {(t2 - t0) / K * 1e3:.2f} ms per cycle (GPU)")
