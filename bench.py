"""Headline benchmark: train-step samples/sec (proj + loss + GAN fwd/bwd), batch 64 per GPU, N MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

One "step" = one pass of BOTH hot paths over one batch of 64 synthetic samples resident in HBM:
  (P) EffectiveLossFunction.forward -> SupervisedLoss.forward -> backward to (point_cloud, rotation, scale) on
      64 clouds of 2048 points into a 128^3 grid (BASELINE configs[1] shape at the metric's batch), called
      through the drop-in nn.Module API as Learner.one_batch does (code/training_test_shape_net.py:69-100);
  (G) one GAN cycle of code/main.py:691-723 at batch 64, CUB-shaped 256x256 (BASELINE configs[2] shape at the
      metric's batch): 1 generator step + 2 discriminator steps, each with its Adam step (+ EMA generator),
      bf16 MFMA convolutions, fp32 master weights.
`value` counts 64 samples per step (conservative: the GAN cycle alone consumes 3 loader batches = 192 textures);
`proj_samples_per_s` and `gan_samples_per_s` give the two halves separately.
Weak scaling: every rank runs the same per-GPU batch; (P) has no collective, (G) all-reduces gradients (one flat
RCCL all-reduce per optimiser step) and SyncBN statistics.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed over the same K steps) and
`cpu_baseline` (the CPU oracle of the projection path, a scalar C port, on a bounded sample).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy ceiling)
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak (2:1-sparsity marketing figure excluded)


def make_clouds(B, N, S, seed, device):
    g = torch.Generator().manual_seed(seed)
    pc = (torch.rand(B, N, 3, generator=g) - 0.5) * 0.7
    q = torch.randn(B, 4, generator=g)
    sc = torch.sigmoid(torch.randn(B, 1, generator=g))
    mask = (torch.rand(B, 2 * S, 2 * S, generator=g) > 0.5).float()
    return [x.to(device) for x in (pc, q, sc, mask)]


def make_textures(B, R, seed, device):
    g = torch.Generator().manual_seed(seed)
    x_tex = torch.rand(B, 3, R, R, generator=g) * 2 - 1
    x_alpha = (torch.rand(B, 1, R, R, generator=g) > 0.4).float()
    x_mesh = 0.05 * torch.randn(B, 3, 32, 32, generator=g)
    c = torch.randint(0, 200, (B, 1), generator=g)
    return [x.to(device) for x in (x_tex, x_alpha, x_mesh, c)]


def cpu_baseline(N, S, seconds_budget=15.0):
    """oracle/p_oracle.c (scalar C port of the reference's literal projection arithmetic) on 1 host core, fwd+bwd,
    on a bounded sample of the same workload (clouds of N points into an S^3 grid).  The GAN half has no CPU leg
    here: the reference's torch-CPU cycle at B=16 takes 12.75 s on 8 cores = 3.8 samples/s."""
    from oracle import p_oracle as po

    rs = np.random.RandomState(0)
    taps = po.taps(3.0, 21, True)
    done, t0 = 0, time.perf_counter()
    while True:
        pc = ((rs.rand(1, N, 3) - 0.5) * 0.7).astype(np.float32)
        q = rs.randn(1, 4).astype(np.float32)
        sc = (1 / (1 + np.exp(-rs.randn(1, 1)))).astype(np.float32)
        mask = (rs.rand(1, 2 * S, 2 * S) > 0.5).astype(np.float32)
        proj = po.forward(pc, q, sc, S, taps)
        po.sup_loss(proj, mask)
        po.backward(pc, q, sc, po.sup_loss_bwd(proj, mask), S, taps)
        done += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or done >= 64:
            break
    return {"value": done / el, "unit": "samples/s", "cores": 1, "kind": "port",
            "sample": f"projection half only: {done} clouds of {N} points -> {S}^3 grid, fwd+bwd, oracle/p_oracle.c, "
                      f"{el:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="samples per GPU")
    ap.add_argument("--points", type=int, default=2048)
    ap.add_argument("--grid", type=int, default=128)
    ap.add_argument("--res", type=int, default=256, help="texture resolution of the GAN half")
    ap.add_argument("--workload", choices=["both", "proj", "gan"], default="both")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    pkg = importlib.import_module("2dimageto3dmodel_amd")
    par = importlib.import_module("2dimageto3dmodel_amd.parallel")
    train = importlib.import_module("2dimageto3dmodel_amd.train")
    rank, local_rank, world = par.init_from_env("cuda")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist

    B, N, S, R = args.batch, args.points, args.grid, args.res
    do_p, do_g = args.workload in ("both", "proj"), args.workload in ("both", "gan")

    if do_p:
        pc, q, sc, mask = make_clouds(B, N, S, 1234 + 2 + 17 * rank, dev)
        for t in (pc, q, sc):
            t.requires_grad_()
        elf = pkg.EffectiveLossFunction(voxel_size=S).to(dev)
        crit = pkg.SupervisedLoss()
    if do_g:
        gargs = argparse.Namespace(norm_g="syncbatch", norm_d="none", conditional_class=True, conditional_color=False,
                                   conditional_text=False, n_classes=[200], texture_resolution=R, mask_output=True,
                                   num_discriminators=2, texture_only=False, text_embedding_dim=256)
        torch.manual_seed(1234 + 3)
        # the G step's mesh smoothness regulariser (code/main.py:697-705) on a procedural 32 x 16 UV sphere (same topology
        # as the reference's template asset, which is not redistributed)
        import tempfile
        mesh_mod = importlib.import_module("2dimageto3dmodel_amd.mesh")
        with tempfile.TemporaryDirectory() as tmp:
            template = mesh_mod.MeshTemplate(mesh_mod.write_uv_sphere_obj(os.path.join(tmp, "uvsphere_16rings.obj")),
                                             is_symmetric=True, device=dev)
        trainer = train.GanTrainer(gargs, device=dev, mesh_template=template)
        trainer.train()
        batches = [make_textures(B, R, 1234 + 3 + 17 * rank + i, dev) for i in range(3)]

    last = {}

    def step_p():
        pc.grad = q.grad = sc.grad = None
        loss = crit(elf(pc, q, sc), mask)["full_loss"]
        loss.backward()
        last["proj_loss"] = loss

    def step_g():
        for x_tex, x_alpha, x_mesh, c in batches:  # 1 G step + 2 D steps, one loader batch each
            last.update(trainer.iteration(x_tex, x_alpha, x_mesh, c))

    def step():
        if do_p:
            step_p()
        if do_g:
            step_g()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, k):
        barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        barrier()
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item()

    for _ in range(args.warmup):
        step()
    dt = timed(step, args.steps)

    # ---- the two halves separately, and per-kernel HIP-event timing of the same K steps (events on the launch
    #      stream = torch's current stream); separate passes so that no marker sits inside the timed region above
    dt_p = timed(step_p, args.steps) if do_p else None
    dt_g = timed(step_g, args.steps) if do_g else None
    pkg._lib.enable_kernel_timers(True)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    kt = pkg._lib.collect_kernel_timers()  # name -> (launches, total_ms, total algorithmic work)
    pkg._lib.enable_kernel_timers(False)

    if rank == 0:
        # the conv entry points are timed per KERNEL FAMILY they dispatched to (m355_last_kernel: k_conv_glds, k_conv_halo,
        # k_wgrad_dma, ...), i.e. under the names rocprofv3 lists
        fam = kt
        dom = max(fam, key=lambda k: fam[k][1])
        cnt, tot_ms, work = fam[dom]
        is_conv = dom.startswith("k_conv") or dom.startswith("k_wgrad")
        traffic = rocprof_us = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):  # measured by separate rocprofv3 passes (scripts/make_profile.sh)
            e = json.load(open(tpath)).get(dom, {})
            traffic, rocprof_us = e.get("hbm_bytes_per_launch"), e.get("rocprof_avg_us")
        rate = work / (tot_ms * 1e-3) / (1e12 if is_conv else 1e9)
        peak = MFMA_BF16_PEAK_TF if is_conv else HBM_PEAK_GBS
        conv_ms = sum(v[1] for k, v in kt.items() if k.startswith("k_conv") or k.startswith("k_wgrad"))
        conv_fl = sum(v[2] for k, v in kt.items() if k.startswith("k_conv") or k.startswith("k_wgrad"))
        workload = []
        if do_p:
            workload.append(f"projection+silhouette-loss fwd/bwd on {B} clouds/GPU of {N} pts -> {S}x{S}")
        if do_g:
            workload.append(f"GAN cycle (1 G + 2 D steps, Adam) at batch {B}/GPU, {R}x{R}, nd=2, class-conditional, "
                            f"syncbatch, mesh flat-loss regulariser in the G step")
        out = {
            "metric": "train-step samples/sec (proj+loss+GAN fwd/bwd), batch 64", "value": world * B * args.steps / dt,
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if do_g else "f32", "data": "synthetic",
            "config": {"workload": " + ".join(workload), "global_batch": world * B, "points": N, "grid": S,
                       "texture_resolution": R, "parallelism": f"dp{world}",
                       "losses": {k: float(v) for k, v in last.items()}},
            "proj_samples_per_s": (world * B * args.steps / dt_p) if do_p else None,
            "gan_samples_per_s": (world * 3 * B * args.steps / dt_g) if do_g else None,
            "roofline": {"bound": "mfma" if is_conv else "hbm", "kernel": dom, "achieved": rate, "peak": peak,
                         "unit": "TFLOP/s" if is_conv else "GB/s", "frac": rate / peak, "traffic": traffic,
                         "avg_kernel_us": tot_ms / cnt * 1e3, "rocprof_avg_kernel_us": rocprof_us,
                         "launches_per_step": cnt / args.steps,
                         "share_of_kernel_time": tot_ms / sum(v[1] for v in kt.values()),
                         "all_conv_tflops": (conv_fl / (conv_ms * 1e-3) / 1e12) if conv_ms else None,
                         "work_per_launch": work / cnt,
                         "note": "kernel = the family of template instantiations behind the named entry points; "
                                 "achieved = algorithmic work (2*M*N*K per conv pass with the real channel counts; "
                                 "SURVEY 8d volume-based bytes for the projection kernels, which keep the volume in "
                                 "LDS) / HIP-event time on the launch stream; traffic = HBM bytes per launch from "
                                 "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (profiles/pmc_traffic.json, "
                                 "(2*FETCH+WRITE)*1024, gfx950 fetch correction), null if not collected"},
            "kernels_ms_per_step": {k: v[1] / args.steps for k, v in sorted(kt.items(), key=lambda kv: -kv[1][1])},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, S)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
