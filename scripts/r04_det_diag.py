"""diagnostic: deterministic mode at a given batch / resolution -- same mode twice, iteration by iteration: which tensors differ first"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("2dimageto3dmodel_amd")
train = importlib.import_module("2dimageto3dmodel_amd.train")
gops = importlib.import_module("2dimageto3dmodel_amd.gan_ops")
import test_gan_modules as T

B = int(sys.argv[1]); R = int(sys.argv[2])
batches = T._cycle_batches(B, R, seed0=7300)
pkg.set_deterministic(True)
gops.STREAMS_ON = bool(int(os.environ.get("DIAG_STREAMS", "0")))


def run(nit):
    torch.manual_seed(733)
    tr = train.GanTrainer(T._trainer_args(texture_resolution=R), device="cuda:0", mesh_template=None)
    tr.train()
    losses = []
    for i in range(nit):
        b, z = batches[i % 3]
        losses += [float(v) for v in tr.iteration(*b, noise=z, epoch=0).values()]
    tr.finish_pending()
    torch.cuda.synchronize()
    st = T._state_bits(tr)
    for k, p in list(tr.generator.named_parameters()) + [("D." + k, p) for k, p in tr.discriminator.named_parameters()]:
        if p.grad is not None:
            st["grad." + k] = p.grad.detach().clone()
    return st, losses


for nit in (1, 2, 3, 4, 6):
    (sa, la), (sb, lb) = run(nit), run(nit)
    bad = [k for k in sa if not torch.equal(sa[k], sb[k])]
    print(f"== B {B} R {R} streams {gops.STREAMS_ON}: {nit} iteration(s): losses equal {la == lb}; differing tensors {len(bad)} of {len(sa)}")
    for k in bad[:50]:
        print("     ", k, tuple(sa[k].shape), f"max diff {(sa[k].float() - sb[k].float()).abs().max().item():.3e}")
    if bad:
        break
