"""deterministic mode, ONE fresh process: a hash of every weight / buffer / Adam moment / gradient after each iteration"""
import hashlib, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("2dimageto3dmodel_amd")
train = importlib.import_module("2dimageto3dmodel_amd.train")
gops = importlib.import_module("2dimageto3dmodel_amd.gan_ops")
import test_gan_modules as T

B = int(sys.argv[1]); R = int(sys.argv[2]); NIT = int(sys.argv[3]); REPEAT = int(sys.argv[4]) if len(sys.argv) > 4 else 1
batches = T._cycle_batches(B, R, seed0=7300)
pkg.set_deterministic(True)
gops.STREAMS_ON = False
for rep in range(REPEAT):
    torch.manual_seed(733)
    tr = train.GanTrainer(T._trainer_args(texture_resolution=R), device="cuda:0", mesh_template=None)
    tr.train()
    hs = []
    for i in range(NIT):
        b, z = batches[i % 3]
        out = tr.iteration(*b, noise=z, epoch=0)
        tr.finish_pending()
        h = hashlib.sha256()
        for k, v in sorted(T._state_bits(tr).items()):
            h.update(v.cpu().contiguous().reshape(-1).view(torch.uint8).numpy().tobytes())
        hs.append(h.hexdigest()[:8] + "/" + "/".join(f"{float(v):.9g}" for v in out.values()))
    print(f"B{B} R{R} rep{rep}:", " ".join(hs))
