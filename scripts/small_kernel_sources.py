"""Which Python lines launch the small ATen kernels (fills, copies, adds) of one GAN cycle?  torch.profiler with stacks over one
eager cycle of bench.py's trainer; prints kernel time and launches per (kernel, first frame inside the package)."""
import argparse, collections, importlib, os, sys, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("2dimageto3dmodel_amd"); train = importlib.import_module("2dimageto3dmodel_amd.train")
mesh_mod = importlib.import_module("2dimageto3dmodel_amd.mesh")
dev = torch.device("cuda", 0); B, R = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 256
gargs = argparse.Namespace(norm_g="syncbatch", norm_d="none", conditional_class=True, conditional_color=False, conditional_text=False,
                           n_classes=[200], texture_resolution=R, mask_output=True, num_discriminators=2, texture_only=False, text_embedding_dim=256)
torch.manual_seed(1237)
with tempfile.TemporaryDirectory() as tmp:
    template = mesh_mod.MeshTemplate(mesh_mod.write_uv_sphere_obj(os.path.join(tmp, "uv.obj")), is_symmetric=True, device=dev)
trainer = train.GanTrainer(gargs, device=dev, mesh_template=template); trainer.train(); trainer.epoch = 0
batches = [bench.make_textures(B, R, 1237 + i, dev) for i in range(3)]
def cycle():
    for b in batches: trainer.iteration(*b)
for _ in range(3): cycle()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    cycle(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    ks = getattr(ev, "kernels", None)
    if not ks: continue
    chain, e = [], ev
    while e is not None and len(chain) < 6:
        chain.append(e.name.replace("autograd::engine::evaluate_function: ", "eval:")[:40]); e = e.cpu_parent
    frame = " < ".join(chain)
    if ev.name in ("aten::add_", "aten::add", "aten::copy_", "aten::cat", "aten::fill_") and ev.input_shapes: frame += "  " + str(ev.input_shapes[:2])
    for k in ks:
        name = k.name
        if "m355" in name or "k_sn_fin" in name: continue
        short = name.split("<")[0].replace("void at::native::", "").replace("(anonymous namespace)::", "")[:48]
        if "elementwise" in short and "<" in name: short += " " + name.split("at::native::")[2].split("<")[0][:28] if name.count("at::native::") >= 2 else ""
        a = agg[(short, frame[:200])]; a[0] += 1; a[1] += k.duration
tot = sum(v[1] for v in agg.values())
print(f"# ATen / runtime kernels of one GAN cycle: {sum(v[0] for v in agg.values())} launches, {tot / 1e3:.3f} ms")
for (short, frame), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{us / 1e3:7.3f} ms x{n:3d}  {short:60s} {frame}")
