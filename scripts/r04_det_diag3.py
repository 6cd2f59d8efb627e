"""diagnostic: deterministic mode, two fresh trainers, forward hooks on every sub-module: the FIRST module output that differs"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("2dimageto3dmodel_amd")
train = importlib.import_module("2dimageto3dmodel_amd.train")
gops = importlib.import_module("2dimageto3dmodel_amd.gan_ops")
import test_gan_modules as T

B = int(sys.argv[1]); R = int(sys.argv[2]); NIT = int(sys.argv[3]) if len(sys.argv) > 3 else 3
batches = T._cycle_batches(B, R, seed0=7300)
pkg.set_deterministic(True)
gops.STREAMS_ON = False


def flat(o):
    if torch.is_tensor(o):
        return [o]
    if isinstance(o, (list, tuple)):
        return [t for x in o for t in flat(x)]
    return []


def run():
    torch.manual_seed(733)
    tr = train.GanTrainer(T._trainer_args(texture_resolution=R), device="cuda:0", mesh_template=None)
    tr.train()
    log = []
    it = [0]

    def hook(name):
        def h(mod, inp, out):
            for j, t in enumerate(flat(out)):
                if t.is_cuda:
                    log.append((f"it{it[0]} {name} out{j} {tuple(t.shape)} {str(t.dtype)[6:]}", t.detach().clone()))
        return h

    for pre, net in (("G", tr.generator), ("D", tr.discriminator)):
        for n, m in net.named_modules():
            m.register_forward_hook(hook(f"{pre}.{n}"))
    for i in range(NIT):
        it[0] = i
        b, z = batches[i % 3]
        out = tr.iteration(*b, noise=z, epoch=0)
        log.append((f"it{i} losses", torch.stack([v.detach().float().reshape(()) for v in out.values()])))
        for k, p in list(tr.generator.named_parameters()) + [("D." + k, p) for k, p in tr.discriminator.named_parameters()]:
            if p.grad is not None:
                log.append((f"it{i} grad {k}", p.grad.detach().clone()))
    tr.finish_pending()
    torch.cuda.synchronize()
    return log


a, b = run(), run()
assert len(a) == len(b)
n = 0
for (ka, ta), (kb, tb) in zip(a, b):
    assert ka == kb
    if not torch.equal(ta, tb):
        d = (ta.float() - tb.float()).abs()
        print(f"DIFF {ka}: max {d.max().item():.3e}, {int((d > 0).sum())} of {d.numel()} elements; first index {int(torch.nonzero(d.reshape(-1) > 0)[0])}")
        n += 1
        if n >= 25:
            break
print(f"B {B} R {R}: {len(a)} captures, {n} differ" + (" (first 25 shown)" if n >= 25 else ""))
