"""diagnostic: deterministic mode, second stream on/off -- which losses / tensors differ, and is each mode self-consistent"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
pkg = importlib.import_module("2dimageto3dmodel_amd")
train = importlib.import_module("2dimageto3dmodel_amd.train")
gops = importlib.import_module("2dimageto3dmodel_amd.gan_ops")
import test_gan_modules as T

batches = T._cycle_batches(4, 128, seed0=6100)
pkg.set_deterministic(True)


def run(streams, iters=2):
    gops.STREAMS_ON = streams
    torch.manual_seed(616)
    tr = train.GanTrainer(T._trainer_args(), device="cuda:0", mesh_template=None)
    tr.train()
    losses = []
    for _ in range(iters):
        for b, z in batches:
            losses += [float(v) for v in tr.iteration(*b, noise=z, epoch=0).values()]
    torch.cuda.synchronize()
    return T._state_bits(tr), losses


def diff(a, b, name):
    (sa, la), (sb, lb) = a, b
    print(name, "losses equal:", la == lb)
    if la != lb:
        for i, (x, y) in enumerate(zip(la, lb)):
            if x != y:
                print("   first differing loss", i, x, y)
                break
    bad = [k for k in sa if not torch.equal(sa[k], sb[k])]
    print("   differing tensors:", len(bad), "of", len(sa), bad[:12])


for iters in (1, 2):
    print("== iters", iters)
    on1, on2, off1, off2 = run(True, iters), run(True, iters), run(False, iters), run(False, iters)
    diff(on1, on2, "on vs on ")
    diff(off1, off2, "off vs off")
    diff(on1, off1, "on vs off")
