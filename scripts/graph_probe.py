"""Can one GAN cycle (1 G step + 2 D steps, Adam, EMA) be captured in a hipGraph and replayed?  Timing A/B."""
import argparse, importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
train = importlib.import_module("2dimageto3dmodel_amd.train")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
gargs = argparse.Namespace(norm_g="syncbatch", norm_d="none", conditional_class=True, conditional_color=False, conditional_text=False,
                           n_classes=[200], texture_resolution=256, mask_output=True, num_discriminators=2, texture_only=False, text_embedding_dim=256)
torch.manual_seed(3)
tr = train.GanTrainer(gargs, device="cuda", capturable=True); tr.train()
batches = [bench.make_textures(B, 256, 5 + i, "cuda") for i in range(3)]
def cyc():
    out = {}
    for b in batches: out.update(tr.iteration(*b))
    return out
def timeit(f, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(3): cyc()
print("eager ms/cycle", timeit(cyc))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): cyc()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = cyc()
print("captured; losses", {k: float(v) for k, v in out.items()})
print("graph ms/cycle", timeit(g.replay))
print("losses after replays", {k: float(v) for k, v in out.items()})
