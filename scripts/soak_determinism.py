"""soak: N training cycles, deterministic mode -- eager twice (second stream on / off) and hipGraph replay, all from one seed on the
same loader batches: every weight, buffer and Adam moment bit-identical after N cycles, no NaN on the way.  A rare race (a missing
stream wait, a workspace shared by two launches in flight, a buffer handed back too early) has N x ~750 launches to show up."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("2dimageto3dmodel_amd")
train = importlib.import_module("2dimageto3dmodel_amd.train")
gops = importlib.import_module("2dimageto3dmodel_amd.gan_ops")
import test_gan_modules as T

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
R = int(sys.argv[3]) if len(sys.argv) > 3 else 128
ND = int(sys.argv[4]) if len(sys.argv) > 4 else 2
batches = T._cycle_batches(B, R, seed0=7300)
pkg.set_deterministic(True)


def fresh():
    torch.manual_seed(733)
    tr = train.GanTrainer(T._trainer_args(texture_resolution=R, num_discriminators=ND), device="cuda:0", mesh_template=None, capturable=True)
    tr.train()
    return tr


def eager(streams):
    gops.STREAMS_ON = streams
    tr = fresh()
    t0 = time.perf_counter()
    last = {}
    for _ in range(N):
        for b, z in batches:
            last.update(tr.iteration(*b, noise=z, epoch=0))
    torch.cuda.synchronize()
    return T._state_bits(tr), {k: float(v) for k, v in last.items()}, time.perf_counter() - t0


def graph(streams):
    gops.STREAMS_ON = streams
    tr = fresh()
    cyc = tr.capture_cycle([b for b, _ in batches], epoch=0, warmup=2, noises=[z for _, z in batches])
    t0 = time.perf_counter()
    for _ in range(N):
        out = cyc.replay()
    torch.cuda.synchronize()
    return T._state_bits(tr), {k: float(v) for k, v in out.items()}, time.perf_counter() - t0


runs = {"eager, one stream": eager(False), "eager, two streams": eager(True), "graph, one stream": graph(False), "graph, two streams": graph(True)}
ref_name = "eager, one stream"
ref = runs[ref_name]
ok = True
for name, (st, losses, dt) in runs.items():
    bad = [k for k in st if not torch.equal(st[k], ref[0][k])]
    nan = [k for k in st if torch.is_floating_point(st[k]) and not bool(torch.isfinite(st[k]).all())]
    ok &= not bad and not nan and losses == ref[1]
    print(f"{name:20s} {N} cycles (batch {B}, {R}^2, nd {ND}) in {dt:6.2f} s  losses {losses}  differing tensors vs '{ref_name}': {len(bad)} of {len(st)}  non-finite: {len(nan)}")
print("SOAK", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
