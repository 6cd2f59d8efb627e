"""ONE fresh process, deterministic mode: hash of every sub-module output (forward hooks, hashed on the spot) + every gradient"""
import hashlib, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("2dimageto3dmodel_amd")
train = importlib.import_module("2dimageto3dmodel_amd.train")
gops = importlib.import_module("2dimageto3dmodel_amd.gan_ops")
import test_gan_modules as T

B = int(sys.argv[1]); R = int(sys.argv[2]); NIT = int(sys.argv[3]); OUT = sys.argv[4]
SYNC = os.environ.get("TRACE_SYNC", "1") == "1"
batches = T._cycle_batches(B, R, seed0=7300)
pkg.set_deterministic(True)
gops.STREAMS_ON = False
torch.manual_seed(733)
tr = train.GanTrainer(T._trainer_args(texture_resolution=R), device="cuda:0", mesh_template=None)
tr.train()
log, keep, it = [], [], [0]


def hh(t):
    return hashlib.sha256(t.detach().cpu().contiguous().reshape(-1).view(torch.uint8).numpy().tobytes()).hexdigest()[:10]


def flat(o):
    if torch.is_tensor(o):
        return [o]
    if isinstance(o, (list, tuple)):
        return [t for x in o for t in flat(x)]
    return []


def hook(name):
    def h(mod, inp, out):
        for j, t in enumerate(flat(out)):
            if t.is_cuda:
                if SYNC:
                    log.append(f"it{it[0]} fwd {name} out{j} {tuple(t.shape)} {hh(t)}")
                else:
                    keep.append((f"it{it[0]} fwd {name} out{j} {tuple(t.shape)}", t.detach().clone()))
    return h


for pre, net in (("G", tr.generator), ("D", tr.discriminator)):
    for n, m in net.named_modules():
        if n:
            m.register_forward_hook(hook(f"{pre}.{n}"))
for i in range(NIT):
    it[0] = i
    b, z = batches[i % 3]
    out = tr.iteration(*b, noise=z, epoch=0)
    tr.finish_pending()
    for name, t in keep:
        log.append(f"{name} {hh(t)}")
    keep.clear()
    log.append(f"it{i} losses " + "/".join(f"{float(v):.9g}" for v in out.values()))
    for k, p in list(tr.generator.named_parameters()) + [("D." + k, p) for k, p in tr.discriminator.named_parameters()]:
        if p.grad is not None:
            log.append(f"it{i} grad {k} {hh(p.grad)}")
open(OUT, "w").write("\n".join(log) + "\n")
print("trace", OUT, len(log), hashlib.sha256("\n".join(log).encode()).hexdigest()[:10])
