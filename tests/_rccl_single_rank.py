"""Helper of tests/test_distributed_gpu.py::test_rccl_single_rank (own process: it creates a process group).

backend="nccl" (= RCCL) with ONE rank on cuda:0 and M355_FORCE_COLLECTIVES=1: every collective of the data-parallel GAN
path -- parameter broadcast, the fused SyncBN [sum|sumsq|count] / moment all-reduces, the flat gradient all-reduce --
goes through RCCL exactly as it does with N ranks (same tensors, same streams), with a trivially known result."""
import argparse
import importlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", M355_FORCE_COLLECTIVES="1")
os.environ.setdefault("MASTER_PORT", "29611")
par = importlib.import_module("2dimageto3dmodel_amd.parallel")
train = importlib.import_module("2dimageto3dmodel_amd.train")
rank, local_rank, world = par.init_from_env("cuda", force=True)
import torch.distributed as dist

assert dist.get_backend() == "nccl" and par.collectives_on()
dev = "cuda:0"
gargs = argparse.Namespace(norm_g="syncbatch", norm_d="none", conditional_class=True, conditional_color=False,
                           conditional_text=False, n_classes=[200], texture_resolution=128, mask_output=True,
                           num_discriminators=2, texture_only=False, text_embedding_dim=256)


def run(force):
    if force:
        os.environ["M355_FORCE_COLLECTIVES"] = "1"
    else:
        os.environ.pop("M355_FORCE_COLLECTIVES", None)
    torch.manual_seed(11)
    tr = train.GanTrainer(gargs, device=dev)
    tr.train()
    g = torch.Generator().manual_seed(50)
    par.reset_stats(time_allreduce=True)
    losses = []
    for it in range(3):
        B, R = 2, 128
        x_tex = (torch.rand(B, 3, R, R, generator=g) * 2 - 1).to(dev)
        x_alpha = (torch.rand(B, 1, R, R, generator=g) > 0.4).float().to(dev)
        x_mesh = (0.05 * torch.randn(B, 3, 32, 32, generator=g)).to(dev)
        c = torch.randint(0, 200, (B, 1), generator=g).to(dev)
        z = torch.randn(B, 64, generator=g).to(dev)
        out = tr.iteration(x_tex, x_alpha, x_mesh, c, noise=z)
        losses += [float(v) for v in out.values()]
    tr.finish_pending()   # (the last D step's asynchronous all-reduce + optimiser step)
    torch.cuda.synchronize()
    w = tr.generator.blk6.conv2.weight_orig.detach().clone()
    st = dict(par.stats)
    st["n_cbn"] = sum(1 for m in tr.generator.modules() if m.__class__.__name__ == "ConditionalBatchNorm2d")
    return losses, w, st, par.allreduce_ms()


def overlap_probe():
    """the asynchronous discriminator all-reduce (GanTrainer.overlap_comm: issued after D's backward on the gradient communicator,
    awaited -- with optimizer_d.step() -- after the NEXT generator forward) against the synchronous order, deterministic mode, one
    RCCL rank with the collectives forced on: every weight and Adam moment must be bit-identical after two cycles"""
    pkg = importlib.import_module("2dimageto3dmodel_amd")
    os.environ["M355_FORCE_COLLECTIVES"] = "1"
    prev = pkg.set_deterministic(True)
    try:
        g = torch.Generator().manual_seed(52)
        B, R = 2, 128
        batches, noises = [], []
        for _ in range(3):
            batches.append([(torch.rand(B, 3, R, R, generator=g) * 2 - 1).to(dev), (torch.rand(B, 1, R, R, generator=g) > 0.4).float().to(dev),
                            (0.05 * torch.randn(B, 3, 32, 32, generator=g)).to(dev), torch.randint(0, 200, (B, 1), generator=g).to(dev)])
            noises.append(torch.randn(B, 64, generator=g).to(dev))
        states, pend = [], []
        for overlap in (True, False):
            torch.manual_seed(13)
            t = train.GanTrainer(gargs, device=dev)
            t.overlap_comm = overlap
            t.train()
            for _ in range(2):
                for b, z in zip(batches, noises):
                    t.iteration(*b, noise=z, epoch=0)
            pend.append(bool(t._pending_d))
            t.finish_pending()
            torch.cuda.synchronize()
            states.append([v.detach().clone() for v in t.state_dict().values()] +
                          [sv.detach().clone() for p_ in t.discriminator.parameters() for sv in t.optimizer_d.state[p_].values() if torch.is_tensor(sv)])
        # the generator's gradients as two messages (the first issued from inside the backward, parallel.BucketedGradReducer) against
        # one flat message at the end
        torch.manual_seed(13)
        t = train.GanTrainer(gargs, device=dev)
        t.overlap_comm, t.overlap_g = False, False
        t.train()
        for _ in range(2):
            for b, z in zip(batches, noises):
                t.iteration(*b, noise=z, epoch=0)
        torch.cuda.synchronize()
        flat = [v.detach().clone() for v in t.state_dict().values()] + \
               [sv.detach().clone() for p_ in t.discriminator.parameters() for sv in t.optimizer_d.state[p_].values() if torch.is_tensor(sv)]
        return {"pending_after_last_iteration": pend, "bit_identical": all(torch.equal(a, b) for a, b in zip(*states)),
                "buckets_bit_identical": all(torch.equal(a, b) for a, b in zip(states[1], flat))}
    finally:
        pkg.set_deterministic(prev)


def graph_probe():
    """RCCL collectives INSIDE a captured training cycle: capture one cycle (G, D, D) with the collectives forced
    on -- SyncBN all-reduces on the default communicator, the flat gradient all-reduces (one of them asynchronous) on the gradient
    communicator -- replay it twice and compare with two eager cycles in deterministic mode.  Reports what happened instead of
    asserting: whether RCCL can be captured on this stack is the question."""
    pkg = importlib.import_module("2dimageto3dmodel_amd")
    os.environ["M355_FORCE_COLLECTIVES"] = "1"
    res = {"captured": False, "replayed": False, "bit_identical_to_eager": None, "error": None}
    prev = pkg.set_deterministic(True)
    try:
        g = torch.Generator().manual_seed(51)
        B, R = 2, 128
        batches, noises = [], []
        for _ in range(3):
            batches.append([(torch.rand(B, 3, R, R, generator=g) * 2 - 1).to(dev), (torch.rand(B, 1, R, R, generator=g) > 0.4).float().to(dev),
                            (0.05 * torch.randn(B, 3, 32, 32, generator=g)).to(dev), torch.randint(0, 200, (B, 1), generator=g).to(dev)])
            noises.append(torch.randn(B, 64, generator=g).to(dev))

        def fresh():
            torch.manual_seed(12)
            t = train.GanTrainer(gargs, device=dev, capturable=True)
            t.train()
            return t

        A = fresh()
        for _ in range(2):
            for b, z in zip(batches, noises):
                A.iteration(*b, noise=z, epoch=0)
        A.finish_pending()
        Bt = fresh()
        cyc = Bt.capture_cycle(batches, epoch=0, warmup=2, noises=noises)
        res["captured"] = True
        for _ in range(2):
            cyc.replay()
        torch.cuda.synchronize()
        res["replayed"] = True
        same = all(torch.equal(a, b) for a, b in zip(A.state_dict().values(), Bt.state_dict().values()))
        res["bit_identical_to_eager"] = bool(same)
    except Exception as e:  # noqa: BLE001
        res["error"] = f"{type(e).__name__}: {str(e)[:400]}"
    finally:
        pkg.set_deterministic(prev)
    return res


l1, w1, st, ms = run(True)
l0, w0, st0, _ = run(False)
overlap = overlap_probe() if "--overlap" in sys.argv else None
graph = graph_probe() if "--graph" in sys.argv else None
try:
    dist.destroy_process_group()
except Exception:  # noqa: BLE001  (a failed capture can leave the communicator unusable: the result line matters)
    pass
print(json.dumps({"graph": graph, "overlap": overlap,"losses_rccl": l1, "losses_plain": l0, "max_w_diff": float((w1 - w0).abs().max()),
                  "frac_w_diff": float(((w1 - w0).abs() > 1e-6).float().mean()),
                  "grad_allreduces": st["grad_allreduces"], "n_cbn": st["n_cbn"], "syncbn_collectives": st["syncbn_collectives"],
                  "allreduce_ms": ms, "plain_collectives": st0["grad_allreduces"] + st0["syncbn_collectives"]}))
