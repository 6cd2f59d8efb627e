"""CPU, world_size 2 over gloo: the N > 1 path of the GAN half -- flat gradient all-reduce and SyncBN statistics
(torch path of gan_ops.BatchNorm2d: same collectives as the HIP path, which needs a GPU)."""
import importlib
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    par = importlib.import_module("2dimageto3dmodel_amd.parallel")
    G = importlib.import_module("2dimageto3dmodel_amd.gan_ops")
    r, _, w = par.init_from_env("cpu")
    assert (r, w) == (rank, world) and par.world_size() == world
    try:
        # ---- broadcast_parameters + FlatGradReducer: gradients become the mean over ranks
        torch.manual_seed(100 + rank)
        lin = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
        par.broadcast_parameters(lin, src=0)
        ref = [p.detach().clone() for p in lin.parameters()]
        gathered = [torch.zeros_like(ref[0]) for _ in range(world)]
        dist.all_gather(gathered, ref[0])
        assert all(torch.equal(g, gathered[0]) for g in gathered)
        x = torch.full((4, 5), float(rank + 1))
        lin(x).sum().backward()
        local = [p.grad.clone() for p in lin.parameters()]
        par.FlatGradReducer(lin.parameters())()
        for p, g in zip(lin.parameters(), local):
            both = [torch.zeros_like(g) for _ in range(world)]
            dist.all_gather(both, g)
            assert torch.allclose(p.grad, sum(both) / world, atol=1e-6)
        # ---- SynchronizedBatchNorm2d: statistics and gradients equal single-process BN over the global batch
        torch.manual_seed(7)
        full = torch.randn(2 * world, 6, 5, 16)                      # NHWC, global batch
        scale = 1 + 0.1 * torch.randn(2 * world, 16)
        shift = 0.1 * torch.randn(2 * world, 16)
        sl = slice(2 * rank, 2 * rank + 2)
        xs = full[sl].clone().requires_grad_()
        sbn = G.SynchronizedBatchNorm2d(16)
        y = sbn(xs, scale[sl], shift[sl], 0.2).float()
        w_out = torch.linspace(0.5, 1.5, y.numel()).view_as(y)
        (y * w_out).sum().backward()
        xf = full.clone().requires_grad_()
        bn = G.BatchNorm2d(16)
        yf = bn(xf, scale, shift, 0.2).float()
        wf = torch.zeros_like(yf)
        wf[sl] = w_out
        # every rank back-propagates its own shard's loss; the SyncBN backward sums the statistic gradients over ranks
        allw = [torch.zeros_like(w_out) for _ in range(world)]
        dist.all_gather(allw, w_out)
        (yf * torch.cat(allw, 0)).sum().backward()
        assert torch.allclose(y, yf[sl].detach(), atol=2e-2)          # bf16 outputs
        assert torch.allclose(sbn.running_mean, bn.running_mean, atol=1e-6)
        assert torch.allclose(sbn.running_var, bn.running_var, atol=1e-6)
        assert torch.allclose(xs.grad, xf.grad[sl], atol=2e-2, rtol=2e-2)
        # ---- projection path: shard by SOURCE IMAGE.  Per-cloud silhouette losses stand in for the HIP kernel's
        # output (the kernel needs a GPU; tests/test_proj_gpu.py runs the real loss on shards): the per-image argmin over the K
        # candidates of the ranks' blocks is the global one, and the weighted local means reduce to the global mean
        n_img, K = 7, 4
        torch.manual_seed(99)
        sse = torch.rand(n_img * K)                                   # identical on every rank: the "global batch"
        pose_w = torch.nn.Linear(3, 1)
        par.broadcast_parameters(pose_w, src=0)
        feats = torch.randn(n_img, 3)
        want_idx = sse.view(-1, K).argmin(-1)
        sh = par.shard_by_image(n_img, K)
        assert (sh.img_hi - sh.img_lo) == (4 if rank == 0 else 3) and sh.cloud_lo == sh.img_lo * K
        loc = sh.clouds(sse).view(-1, K)
        li = loc.argmin(-1)
        both = [torch.zeros(4, dtype=torch.long) for _ in range(world)]
        pad = torch.full((4,), -1, dtype=torch.long)
        pad[:li.numel()] = li
        dist.all_gather(both, pad)
        got = torch.cat([b[b >= 0] for b in both])
        assert torch.equal(got, want_idx)
        # loss = mean over the rank's images (as unsup:120,134), scaled by the shard weight, gradients averaged over ranks
        best = loc[torch.arange(li.numel()), li]
        (sh.weight * (best * pose_w(sh.images(feats)).squeeze(-1)).mean()).backward()
        par.FlatGradReducer(pose_w.parameters())()
        ref_lin = torch.nn.Linear(3, 1)
        ref_lin.load_state_dict(pose_w.state_dict())
        gbest = sse.view(-1, K)[torch.arange(n_img), want_idx]
        (gbest * ref_lin(feats).squeeze(-1)).mean().backward()
        assert torch.allclose(pose_w.weight.grad, ref_lin.weight.grad, atol=1e-6)
        with pytest.raises(ValueError):
            par.check_image_groups(K * 3 + 1, 3, K)
        # ---- BucketedGradReducer + GradBarrier (round 6): the "early" layers' all-reduce is issued from INSIDE the backward pass, when
        # the gradient reaches the trunk; the averages are the flat reducer's bit for bit, and the barrier fired exactly once with the
        # early gradients complete and the late ones not yet there
        torch.manual_seed(300 + rank)
        trunk, head_a, head_b = torch.nn.Linear(6, 8), torch.nn.Linear(8, 5), torch.nn.Linear(8, 3)
        for m in (trunk, head_a, head_b):
            par.broadcast_parameters(m, src=0)
        early, late = list(head_a.parameters()) + list(head_b.parameters()), list(trunk.parameters())
        red = par.BucketedGradReducer([early, late])
        seen = []

        def ready():
            seen.append((all(p.grad is not None for p in early), any(p.grad is not None for p in late)))
            assert red.start_bucket(0)

        xin = torch.randn(4, 6) * (rank + 1)
        t = G.GradBarrier.apply(torch.tanh(trunk(xin)), ready)
        (head_a(t).pow(2).sum() + head_b(t).sum()).backward()
        assert seen == [(True, False)], seen
        local = [p.grad.clone() for p in early + late]
        red()
        assert red.started == [False, False]
        flat_params = [torch.nn.Parameter(p.detach().clone()) for p in early + late]
        for fp, g in zip(flat_params, local):
            fp.grad = g.clone()
        par.FlatGradReducer(flat_params)()
        for p, fp in zip(early + late, flat_params):
            assert torch.equal(p.grad, fp.grad)                       # bucketed == flat, bit for bit
        assert par.stats["grad_allreduces"] >= 3
        # exposed / overlapped accounting: the waits are a part of the issue-to-completion spans
        par.reset_stats(time_allreduce=True)
        for fp, g in zip(flat_params, local):
            fp.grad = g.clone()
        par.FlatGradReducer(flat_params)()
        assert 0.0 <= par.allreduce_exposed_ms() <= par.allreduce_ms() + 1e-6
        par.reset_stats()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback

        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(30)
    assert sorted(r for r, _ in res) == [0, 1]
    for r, msg in res:
        assert msg == "ok", f"rank {r}:\n{msg}"


# ---- bench.py's own launch path: `python bench.py --gpus N` started as ONE process becomes N ranks
def _run_bench(*argv, timeout=300):
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=timeout,
                          env=env, cwd=ROOT)


@pytest.mark.timeout(400)
def test_bench_gpus2_spawns_two_ranks_gloo():
    """the same spawn path the driver's `python bench.py --gpus N` takes (re-exec under torch.distributed.run, rank/device
    binding, barriers, MAX-over-ranks timing, one JSON line from rank 0), on gloo with the collectives-only workload"""
    import json
    r = _run_bench("--gpus", "2", "--backend", "gloo", "--workload", "collectives", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["averaged_correctly"] is True
    assert out["grad_allreduces_per_step"] == 2 and out["allreduce_ms_per_step"] > 0
    assert abs(out["grad_allreduce_mb_per_step"] - 61.0) < 1e-6


@pytest.mark.timeout(400)
def test_bench_strong_scaling_two_ranks_gloo():
    """`--scaling strong`: --batch is the GLOBAL batch, split over the ranks (SURVEY 8d cfg 4: global 64 split N ways); the line
    says so, and a batch the ranks cannot share is refused"""
    import json
    r = _run_bench("--gpus", "2", "--backend", "gloo", "--workload", "collectives", "--steps", "1", "--warmup", "0", "--scaling",
                   "strong", "--batch", "64")
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["scaling"] == "strong" and out["n_gpus"] == 2 and out["config"]["per_gpu_batch"] == 32
    r = _run_bench("--gpus", "2", "--backend", "gloo", "--workload", "collectives", "--steps", "1", "--warmup", "0", "--scaling",
                   "strong", "--batch", "63")
    assert r.returncode != 0 and "not divisible" in r.stderr


def test_bench_refuses_more_gpus_than_visible():
    """`bench.py --gpus 9` on a box with fewer GPUs must fail loudly, never fall back to a 1-GPU measurement"""
    r = _run_bench("--gpus", "9", "--steps", "1", "--warmup", "0", timeout=200)
    assert r.returncode != 0
    assert "only" in r.stderr and "GPU(s) are visible" in r.stderr and not any(ln.startswith("{") for ln in r.stdout.splitlines())


def test_bench_rejects_world_size_mismatch():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True, timeout=200,
                       env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_wgrad_arena_bookkeeping_cpu(pkg):
    """conv.WgradArena (host logic, no kernel): outside a backward pass there is no arena; inside, the first pass only learns
    its size, later passes hand out disjoint, zeroed, 64-element-aligned slices of one buffer that is zero-filled once per pass"""
    import importlib
    import torch
    conv = importlib.import_module("2dimageto3dmodel_amd.conv")
    dev = torch.device("cpu")
    conv.WgradArena._state.pop(dev, None)
    assert conv.WgradArena.take(100, dev) is None   # not inside autograd's backward
    seen = []

    class Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, g):
            a, b = conv.WgradArena.take(100, dev), conv.WgradArena.take(1000, dev)
            if a is not None:
                assert a.numel() == 128 and b.numel() == 1024 and float(a.abs().sum()) == 0 and float(b.abs().sum()) == 0
                assert a.data_ptr() + 4 * 128 <= b.data_ptr() or b.data_ptr() + 4 * 1024 <= a.data_ptr()
                a.fill_(1.0); b.fill_(2.0)   # the next pass must find zeros again
            seen.append(None if a is None else (a.data_ptr(), b.data_ptr()))
            return g

    for _ in range(3):
        Probe.apply(torch.zeros(1, requires_grad=True)).sum().backward()
    assert seen[0] is None and seen[1] is not None and seen[1] == seen[2], seen
