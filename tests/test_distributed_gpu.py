"""GPU, world_size 2 (two processes sharing cuda:0, gloo transport -- RCCL refuses two ranks on one device): the
FUSED SynchronizedBatchNorm2d path (csrc/gan_glue.hip + one all-reduce each way) equals single-process BatchNorm2d
over the global batch, and one sharded GanTrainer cycle runs and keeps the ranks' weights identical."""
import argparse
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
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
                      LOCAL_RANK="0")
    try:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        G = importlib.import_module("2dimageto3dmodel_amd.gan_ops")
        train = importlib.import_module("2dimageto3dmodel_amd.train")
        dev = "cuda:0"
        torch.manual_seed(7)
        full = (torch.randn(2 * world, 12, 10, 64) * 1.3 + 0.2).bfloat16()     # NHWC, global batch
        gamma = 0.1 * torch.randn(2 * world, 64)
        beta = 0.1 * torch.randn(2 * world, 64)
        res = torch.randn(2 * world, 12, 10, 64).bfloat16()
        w_out = torch.randn(2 * world, 12, 10, 64).bfloat16()
        sl = slice(2 * rank, 2 * rank + 2)
        xs = full[sl].to(dev).requires_grad_()
        gs, bs = gamma[sl].to(dev).requires_grad_(), beta[sl].to(dev).requires_grad_()
        sbn = G.SynchronizedBatchNorm2d(64).to(dev)
        y = sbn(xs, gs, bs, 0.2, res[sl].to(dev))
        assert y.grad_fn.__class__.__name__ == "CbnActFnBackward"             # the fused HIP path ran
        y.backward(w_out[sl].to(dev))
        # single-process reference over the global batch (same kernels, no collective)
        xf = full.to(dev).requires_grad_()
        gf, bf = gamma.to(dev).requires_grad_(), beta.to(dev).requires_grad_()
        bn = G.BatchNorm2d(64).to(dev)
        yf = bn(xf, gf, bf, 0.2, res.to(dev))
        yf.backward(w_out.to(dev))
        assert torch.equal(y, yf[sl])
        assert torch.allclose(sbn.running_mean, bn.running_mean, atol=1e-6)
        assert torch.allclose(sbn.running_var, bn.running_var, atol=1e-6)
        assert (xs.grad.float() - xf.grad[sl].float()).abs().max().item() < 2e-2 * xf.grad.float().abs().max().item()
        assert torch.allclose(gs.grad, gf.grad[sl], rtol=1e-4, atol=1e-4)
        assert torch.allclose(bs.grad, bf.grad[sl], rtol=1e-4, atol=1e-4)
        # ---- ragged shards (1 + 3 samples): the fused path all-reduces the real pixel count with the sums
        rsl = slice(0, 1) if rank == 0 else slice(1, 4)
        xr = full[rsl].to(dev).requires_grad_()
        sbn2 = G.SynchronizedBatchNorm2d(64).to(dev)
        yr = sbn2(xr, gamma[rsl].to(dev), beta[rsl].to(dev), 0.2, res[rsl].to(dev))
        yr.backward(w_out[rsl].to(dev))
        assert torch.equal(yr, yf[rsl])
        assert torch.allclose(sbn2.running_var, bn.running_var, atol=1e-6)
        assert (xr.grad.float() - xf.grad[rsl].float()).abs().max().item() < 2e-2 * xf.grad.float().abs().max().item()
        # ---- one GAN cycle, batch sharded over the ranks
        gargs = argparse.Namespace(norm_g="syncbatch", norm_d="none", conditional_class=True, conditional_color=False,
                                   conditional_text=False, n_classes=[200], texture_resolution=128, mask_output=True,
                                   num_discriminators=2, texture_only=False, text_embedding_dim=256)
        torch.manual_seed(11)
        tr = train.GanTrainer(gargs, device=dev)
        tr.train()
        g = torch.Generator().manual_seed(50 + rank)
        B, R = 2, 128
        for _ in range(3):
            x_tex = (torch.rand(B, 3, R, R, generator=g) * 2 - 1).to(dev)
            x_alpha = (torch.rand(B, 1, R, R, generator=g) > 0.4).float().to(dev)
            x_mesh = (0.05 * torch.randn(B, 3, 32, 32, generator=g)).to(dev)
            c = torch.randint(0, 200, (B, 1), generator=g).to(dev)
            out = tr.iteration(x_tex, x_alpha, x_mesh, c)
            assert all(torch.isfinite(v).all() for v in out.values())
        # the last D step's gradient all-reduce is still in flight (issued asynchronously, awaited after the NEXT generator
        # forward -- train.GanTrainer.overlap_comm): its optimiser step has not been applied yet
        assert tr._pending_d
        tr.finish_pending()
        assert not tr._pending_d
        assert tr.reduce_g.started == [False, False] and tr.generator.grad_barrier is None
        for p in list(tr.generator.parameters())[:6] + list(tr.discriminator.parameters())[:6]:
            both = [torch.zeros_like(p) for _ in range(world)]
            dist.all_gather(both, p.detach())
            assert torch.equal(both[0], both[1]), "ranks diverged"
        # ---- the mesh-estimation step (recon_train.ReconTrainer), batch sharded over the ranks: global-batch batch-norm
        # statistics + one flat all-reduce of the generator's and DatasetParams' gradients = the single-process step on the
        # concatenated batch (the reference script, run_reconstruction.py:409-445, is single-GPU)
        import tempfile
        rt = importlib.import_module("2dimageto3dmodel_amd.recon_train")
        mesh_mod = importlib.import_module("2dimageto3dmodel_amd.mesh")
        conv = importlib.import_module("2dimageto3dmodel_amd.conv")
        prev_det = conv.set_deterministic(True)
        with tempfile.TemporaryDirectory() as tmp:
            tpl = mesh_mod.MeshTemplate(mesh_mod.write_uv_sphere_obj(os.path.join(tmp, "uvsphere_16rings.obj")), is_symmetric=True, device=dev)
        Bg, n_data, res = 4, 16, 256      # (the encoder is built for 256 x 256 inputs)
        gi = torch.Generator().manual_seed(321)
        yy, xx = torch.meshgrid(torch.linspace(-1, 1, res), torch.linspace(-1, 1, res), indexing="ij")
        alpha = ((xx ** 2 + yy ** 2) < 0.3).float().expand(Bg, 1, -1, -1)
        X = torch.cat((torch.tanh(torch.nn.functional.interpolate(torch.randn(Bg, 3, 8, 8, generator=gi), size=(res, res), mode="bilinear")) * alpha,
                       alpha), dim=1).to(dev)
        gt_scale = (0.5 + 0.15 * torch.rand(Bg, 1, generator=gi)).to(dev)
        gt_tr = torch.cat((0.2 * (torch.rand(Bg, 2, generator=gi) - 0.5), torch.zeros(Bg, 1)), dim=1).to(dev)
        gt_rot = torch.nn.functional.normalize(torch.randn(Bg, 4, generator=gi) * torch.tensor([0.3, 1.0, 0.3, 0.3]) +
                                               torch.tensor([1.0, 0.0, 0.0, 0.0]), dim=-1).to(dev)
        gt_idx = torch.tensor([3, 7, 7, 12]).to(dev)          # (a repeated index: its rows meet on different ranks)

        def make(dp):
            torch.manual_seed(99)
            t = rt.ReconTrainer(tpl, dataset_size=n_data, texture_resolution=64, image_resolution=res, device=dev, data_parallel=dp)
            with torch.no_grad():
                t.generator.conv_mesh.weight.normal_(0, 0.005)
            return t.train()

        tr_dp, tr_one = make(True), make(False)
        sh = slice(2 * rank, 2 * rank + 2)
        tot, _, _, _, _ = tr_dp.losses(X[sh], gt_scale[sh], gt_tr[sh], gt_rot[sh], gt_idx[sh])
        tot.backward()
        tr_dp.reduce()
        tot1, _, _, _, _ = tr_one.losses(X, gt_scale, gt_tr, gt_rot, gt_idx)      # the whole batch in one process, local statistics
        tot1.backward()
        both = [torch.zeros_like(tot.detach()) for _ in range(world)]
        dist.all_gather(both, tot.detach())
        # (the silhouette term lives on edge pixels that move with the bf16 network's sub-pixel vertex noise: the two runs tile the
        # batch differently; measured 6e-3)
        assert abs(float(sum(both) / world) / float(tot1.detach()) - 1) < 2e-2, (both, tot1)
        # Elementwise on the well-conditioned gradients (texture decoder, per-image pose offsets), by norm on all of them: the
        # mesh branch / encoder gradients of this step are dominated by the flat-loss term, which is chaotic in the displacement
        # map on a random-weight mesh (tests/test_recon_step.py (c): the reference's own fp32 code gives cosine 0.53 under a 1 %
        # perturbation), and the two runs differ by bf16 rounding (different batch tilings, split-K plans)
        named_dp = dict(list(tr_dp.generator.named_parameters()) + list(tr_dp.dataset_params.named_parameters()))
        named_1 = dict(list(tr_one.generator.named_parameters()) + list(tr_one.dataset_params.named_parameters()))
        cosv = lambda a, b: float(torch.dot(a.flatten().double(), b.flatten().double()) / (a.norm().double() * b.norm().double() + 1e-300))
        for k in ("blk5_tex.conv2.weight", "conv_tex.weight", "blk4_tex.conv1.weight", "ds_translation", "ds_scale"):
            assert cosv(named_dp[k].grad, named_1[k].grad) > 0.98, (k, cosv(named_dp[k].grad, named_1[k].grad))
        ratios = [float(named_dp[k].grad.norm() / named_1[k].grad.norm()) for k in named_1 if float(named_1[k].grad.norm()) > 1e-10]
        assert 0.8 < float(np.median(ratios)) < 1.25, float(np.median(ratios))
        bn = [m for m in tr_dp.generator.modules() if isinstance(m, rt.BatchNormAct2d)][0]
        bn1 = [m for m in tr_one.generator.modules() if isinstance(m, rt.BatchNormAct2d)][0]
        assert torch.allclose(bn.running_mean, bn1.running_mean, atol=2e-3) and int(bn.num_batches_tracked) == 1
        # a full iteration keeps the ranks in lock step
        tr_dp.iteration(X[sh], gt_scale[sh], gt_tr[sh], gt_rot[sh], gt_idx[sh])
        for p in list(tr_dp.generator.parameters())[:8] + list(tr_dp.dataset_params.parameters()):
            both = [torch.zeros_like(p) for _ in range(world)]
            dist.all_gather(both, p.detach())
            assert torch.equal(both[0], both[1]), "mesh-estimation step: ranks diverged"
        conv.set_deterministic(prev_det)
        q.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        import traceback

        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_ranks_one_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=840) for _ in procs]   # (a cold `import torch` in the children can take minutes)
    for p in procs:
        p.join(30)
    for r, msg in res:
        assert msg == "ok", f"rank {r}:\n{msg}"


@pytest.mark.timeout(900)
def test_rccl_single_rank():
    """backend "nccl" (RCCL) really executes: one rank on cuda:0 with the collectives forced on runs three GanTrainer
    iterations through RCCL broadcasts / SyncBN all-reduces / the flat gradient all-reduce and lands on the same weights
    and losses as the collective-free run (a 1-rank sum is the identity)."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    env["MASTER_PORT"] = str(_free_port())
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_single_rank.py"), "--overlap"], capture_output=True,
                       text=True, timeout=850, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    # the asynchronous discriminator all-reduce (left in flight across the iteration boundary) against the synchronous order
    assert out["overlap"]["pending_after_last_iteration"] == [True, False] and out["overlap"]["bit_identical"] is True, out["overlap"]
    # 1 G step (every CBN layer forward + backward) + 2 D steps (forward under no_grad): 4 SyncBN collectives per layer;
    # 4 gradient all-reduces (the generator's travel as two messages, the first issued inside its backward: round 6)
    assert out["n_cbn"] == 12
    assert out["grad_allreduces"] == 4 and out["syncbn_collectives"] == 4 * out["n_cbn"] and out["plain_collectives"] == 0
    # ... and the two-message form leaves every weight and Adam moment where the one-message form does (deterministic mode)
    assert out["overlap"]["buckets_bit_identical"] is True, out["overlap"]
    assert out["allreduce_ms"] > 0
    # two runs of the same cycle differ in the summation order of the split-K weight-gradient atomics, and Adam's first
    # step is lr * sign(g): a near-zero gradient may flip (a 2 * lr = 2e-4 difference on that weight).  The collectives
    # themselves are exact for one rank: losses agree to rounding and at most a few weights flip.
    assert out["max_w_diff"] <= 2.5e-4 and out["frac_w_diff"] < 0.02, out
    assert abs(out["losses_rccl"][0] - out["losses_plain"][0]) < 2e-3          # first iteration: no update in between
    assert max(abs(a - b) for a, b in zip(out["losses_rccl"], out["losses_plain"])) < 5e-2


@pytest.mark.timeout(900)
def test_rccl_collectives_inside_a_captured_cycle():
    """one RCCL rank, collectives forced on, a whole training cycle captured into a hipGraph (SyncBN all-reduces,
    the flat gradient all-reduces incl. the asynchronous one) and replayed.  If the stack can capture RCCL the replays must equal
    the eager cycles bit for bit (deterministic mode); if it cannot, the failure mode is recorded in the assertion message and
    `bench.py --graph` stays a single-GPU option."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    env["MASTER_PORT"] = str(_free_port())
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_single_rank.py"), "--graph"], capture_output=True,
                           text=True, timeout=420, env=env, cwd=ROOT)
    except subprocess.TimeoutExpired:
        pytest.xfail("RCCL collectives inside a hipGraph capture: the probe did not finish in 420 s (hang)")
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if not lines:
        # measured on ROCm 7.2 / torch 2.10 (round 4): with a GLOBAL-mode capture about one probe in fifteen died -- ProcessGroupNCCL's
        # watchdog thread queried an event while the capture was open and turned the error into terminate().  CycleGraph now
        # captures thread-locally after a pause (16 of 16 since); should the process still die, the death is recorded (stderr
        # kept under gpurun_out/ for the next reader) instead of failing the suite
        what = [ln for ln in r.stderr.splitlines() if "what()" in ln or "Error" in ln or "error" in ln][:3]
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "rccl_graph_probe_death.txt"), "w") as f:
                f.write(f"rc {r.returncode}\n--- stdout\n{r.stdout[-4000:]}\n--- stderr\n{r.stderr[-12000:]}\n")
        except OSError:
            pass
        pytest.xfail(f"RCCL collectives inside a hipGraph capture: the probe process died (rc {r.returncode}): {what}")
    g = json.loads(lines[-1])["graph"]
    print("RCCL-in-graph probe:", g)
    if g["error"] is not None:
        pytest.xfail("RCCL collectives could not be captured / replayed on this stack: " + g["error"])
    assert g["captured"] and g["replayed"] and g["bit_identical_to_eager"], g


@pytest.mark.timeout(600)
def test_bench_gpus2_fails_loudly_on_one_gpu():
    """`python bench.py --gpus 2` on a 1-GPU box must not print an n_gpus=1 line"""
    import subprocess
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=550, env=env, cwd=ROOT)
    assert r.returncode != 0 and "GPU(s) are visible" in r.stderr
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())


@pytest.mark.timeout(900)
def test_bench_two_ranks_sharing_one_gpu():
    """bench.py's whole multi-rank flow (self-spawn under torch.distributed.run, per-rank seeds, parameter broadcast, SyncBN and
    gradient collectives, barrier + MAX-over-ranks timing, one JSON line from rank 0) with two ranks sharing cuda:0 over gloo
    (M355_SHARE_GPU=1; RCCL refuses two ranks on one device) at a small batch"""
    import json
    import subprocess
    env = dict(os.environ, M355_SHARE_GPU="1", OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
                        "--batch", "8", "--no-cpu-baseline"], capture_output=True, text=True, timeout=850, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 16 and out["config"]["parallelism"] == "dp2"
    assert out["grad_allreduces_per_step"] == 4 and out["syncbn_collectives_per_step"] == 56
    assert abs(out["grad_allreduce_mb_per_step"] - 75.0) < 3.0 and out["allreduce_ms_per_step"] > 0
    assert 0.0 <= out["allreduce_exposed_ms_per_step"] <= out["allreduce_ms_per_step"] + 1e-3   # (waits are a part of the spans)
    assert out["value"] > 0 and "cpu_baseline" not in out
    assert all(np.isfinite(v) for v in out["config"]["losses"].values())


# ---- round 6: the SyncBN messages as ONE kernel launch over peer-mapped device memory (csrc/ipc_exchange.hip, parallel.IpcAllReduce)
def _ipc_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        par = importlib.import_module("2dimageto3dmodel_amd.parallel")
        G = importlib.import_module("2dimageto3dmodel_amd.gan_ops")
        conv = importlib.import_module("2dimageto3dmodel_amd.conv")
        train = importlib.import_module("2dimageto3dmodel_amd.train")
        dev = "cuda:0"
        torch.cuda.set_device(0)
        report = {}
        # (a) the exchange itself: sums equal the rank-ordered sum of the ranks' vectors, bit for bit, on both ranks; more messages
        # than slots; sizes 1 .. the largest SyncBN message
        ex = par.IpcAllReduce(dev, timeout_ms=3000)
        gcpu = torch.Generator().manual_seed(900)
        for n in (1, 129, 1025, 4097):
            data = [torch.randn(world, n, generator=gcpu) * 10 ** float(torch.randint(-3, 4, (1,), generator=gcpu)) for _ in range(11)]
            outs = [ex(d[rank].to(dev).clone()) for d in data]
            torch.cuda.synchronize()
            ex.check()
            for d, o in zip(data, outs):
                want = d[0].clone()
                for r in range(1, world):
                    want = want + d[r]
                assert torch.equal(o.cpu(), want), (n, (o.cpu() - want).abs().max())
        # (a2) channels are independent: the ranks issue two call sites in OPPOSITE orders (rank 0 on two streams, as a forked branch
        # does; rank 1 serially, the other way round) -- a global message order would pair the wrong vectors or wait forever
        va, vb = torch.full((33,), float(rank + 1), device=dev), torch.full((77,), 10.0 * (rank + 1), device=dev)
        if rank == 0:
            s2 = torch.cuda.Stream()
            s2.wait_stream(torch.cuda.current_stream())
            ex(va, channel=1)
            with torch.cuda.stream(s2):
                ex(vb, channel=2)
            torch.cuda.current_stream().wait_stream(s2)
        else:
            ex(vb, channel=2)
            ex(va, channel=1)
        torch.cuda.synchronize()
        ex.check()
        assert torch.equal(va.cpu(), torch.full((33,), 3.0)) and torch.equal(vb.cpu(), torch.full((77,), 30.0))
        # (a3) inside a hipGraph: the per-channel sequence numbers live in device memory, so a captured launch replays (the kernel's
        # arguments are the same every time; what pairs the ranks' messages is the counter in the region)
        src, vg = torch.full((65,), float(rank + 1), device=dev), torch.zeros(65, device=dev)
        torch.cuda.synchronize()
        dist.barrier()
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph):
            vg.copy_(src)
            ex(vg, channel=3)
            ex(vg, channel=4)
        for _ in range(5):
            vg.zero_()
            gph.replay()
            torch.cuda.synchronize()
            assert torch.equal(vg.cpu(), torch.full((65,), 2.0 * sum(range(1, world + 1)))), vg[:4]
        ex.check()
        # (b) latency of a message as the stream sees it (two processes time-share ONE GPU here: an upper bound for xGMI peers)
        v = torch.ones(513, device=dev)
        for _ in range(20):
            ex(v)
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            ex(v)
        e1.record()
        torch.cuda.synchronize()
        ex.check()
        report["us_per_message"] = e0.elapsed_time(e1) * 1e3 / 200
        # (c) a peer that does not show up: the wait is BOUNDED and reported (rank 1 skips one message, then catches up)
        ex.timeout_ms = 300
        if rank == 0:
            ex(torch.ones(8, device=dev))
            torch.cuda.synchronize()
            with pytest.raises(RuntimeError, match="did not arrive"):
                ex.check()
        dist.barrier()
        if rank == 1:
            ex(torch.ones(8, device=dev))      # (finds rank 0's message of that sequence number in place: no wait)
            torch.cuda.synchronize()
            ex.check()
        dist.barrier()
        ex.timeout_ms = 3000
        out = ex(torch.full((5,), float(rank + 1), device=dev))   # back in step
        torch.cuda.synchronize()
        ex.check()
        assert torch.equal(out.cpu(), torch.full((5,), float(sum(range(1, world + 1)))))
        ex.close()
        # (d) SyncBN through the exchange == SyncBN through torch.distributed, bit for bit; then a training cycle
        torch.manual_seed(7)
        full = (torch.randn(2 * world, 12, 10, 64) * 1.3 + 0.2).bfloat16()
        gamma, beta = 0.1 * torch.randn(2 * world, 64), 0.1 * torch.randn(2 * world, 64)
        w_out = torch.randn(2 * world, 12, 10, 64).bfloat16()
        sl = slice(2 * rank, 2 * rank + 2)

        def sbn_pass():
            xs = full[sl].to(dev).requires_grad_()
            gs, bs = gamma[sl].to(dev).requires_grad_(), beta[sl].to(dev).requires_grad_()
            sbn = G.SynchronizedBatchNorm2d(64).to(dev)
            y = sbn(xs, gs, bs, 0.2)
            y.backward(w_out[sl].to(dev))
            torch.cuda.synchronize()
            return [t.detach().clone() for t in (y, xs.grad, gs.grad, bs.grad, sbn.running_mean, sbn.running_var)]

        os.environ["M355_SYNCBN_IPC"] = "0"
        ref = sbn_pass()
        os.environ["M355_SYNCBN_IPC"] = "1"
        got = sbn_pass()
        assert par.syncbn_ipc(torch.device(dev)) is not None and par.syncbn_ipc(torch.device(dev)).messages == 2
        for a, b in zip(got, ref):
            assert torch.equal(a, b)
        gargs = argparse.Namespace(norm_g="syncbatch", norm_d="none", conditional_class=True, conditional_color=False,
                                   conditional_text=False, n_classes=[200], texture_resolution=128, mask_output=True,
                                   num_discriminators=2, texture_only=False, text_embedding_dim=256)
        prev = conv.set_deterministic(True)

        def cycle(ipc):
            os.environ["M355_SYNCBN_IPC"] = "1" if ipc else "0"
            torch.manual_seed(11)
            tr = train.GanTrainer(gargs, device=dev)
            tr.train()
            g = torch.Generator().manual_seed(50 + rank)
            B, R = 2, 128
            for _ in range(3):
                b = ((torch.rand(B, 3, R, R, generator=g) * 2 - 1).to(dev), (torch.rand(B, 1, R, R, generator=g) > 0.4).float().to(dev),
                     (0.05 * torch.randn(B, 3, 32, 32, generator=g)).to(dev), torch.randint(0, 200, (B, 1), generator=g).to(dev))
                z = torch.randn(B, 64, generator=g).to(dev)
                tr.iteration(*b, noise=z, epoch=0)
            tr.finish_pending()
            torch.cuda.synchronize()
            return [v.detach().clone() for v in tr.state_dict().values()]

        s_rccl = cycle(False)
        s_ipc = cycle(True)
        exch = par.syncbn_ipc(torch.device(dev))
        exch.check()
        report["messages_per_cycle"] = exch.messages - 2
        conv.set_deterministic(prev)
        assert all(torch.equal(a, b) for a, b in zip(s_ipc, s_rccl)), "SyncBN over the exchange changed the training cycle's bits"
        for t in s_ipc[:12]:
            both = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(both, t)
            assert torch.equal(both[0], both[1]), "ranks diverged"
        os.environ["M355_SYNCBN_IPC"] = "0"
        q.put((rank, "ok", report))
    except Exception:  # noqa: BLE001
        import traceback

        q.put((rank, traceback.format_exc(), {}))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_syncbn_messages_over_peer_mapped_memory():
    """parallel.IpcAllReduce (csrc/ipc_exchange.hip) with two processes sharing cuda:0 -- rank-ordered sums bit
    for bit, slot reuse, a bounded and reported wait when a peer is missing, SyncBN and a whole training cycle bit-identical to the
    torch.distributed path, ranks in lock step.  Prints the per-message latency on this (time-shared) device."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ipc_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=840) for _ in procs]
    for p in procs:
        p.join(30)
    for r, msg, rep in res:
        assert msg == "ok", f"rank {r}:\n{msg}"
    print("IpcAllReduce:", [rep for _, _, rep in res])
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "ipc_exchange_report.txt"), "w") as f:
            f.write(repr([rep for _, _, rep in res]) + "\n")
    except OSError:
        pass
    assert all(rep["messages_per_cycle"] == 4 * 12 for _, _, rep in res), res     # 12 CBN layers at 128^2: fwd x 3 forwards + bwd x 1
