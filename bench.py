"""Headline benchmark: projection + silhouette-loss train step (fwd + bwd) on N MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

A "step" is one pass of the hot path over one batch of synthetic clouds resident in HBM:
  EffectiveLossFunction.forward -> SupervisedLoss.forward -> backward to (point_cloud, rotation, scale),
called through the drop-in nn.Module API exactly as the reference's Learner.one_batch would
(code/training_test_shape_net.py:69-100), minus the encoder/decoder (out of scope).
Workload = BASELINE.json metric's configuration: batch 64 per GPU, 2048-point clouds, 128x128 silhouette
(configs[1] shape at the metric's batch).  The path shards by cloud with no data-path collective
(the projection has no parameters), so N > 1 is weak scaling.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed) and `cpu_baseline`
(the CPU oracle = scalar C port of the reference path, timed on a bounded sample of the same workload).
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

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy ceiling)


def make_batch(B, N, S, seed, device):
    g = torch.Generator().manual_seed(seed)
    pc = (torch.rand(B, N, 3, generator=g) - 0.5) * 0.7
    q = torch.randn(B, 4, generator=g)
    sc = torch.sigmoid(torch.randn(B, 1, generator=g))
    mask = (torch.rand(B, 2 * S, 2 * S, generator=g) > 0.5).float()
    return [x.to(device) for x in (pc, q, sc, mask)]


def cpu_baseline(N, S, seconds_budget=20.0):
    """oracle/p_oracle.c (scalar C port of the reference's literal arithmetic) on 1 host core, fwd+bwd,
    on a bounded sample of the same workload (clouds of N points into an S^3 grid)."""
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
            "sample": f"{done} clouds of {N} points -> {S}^3 grid, fwd+bwd, oracle/p_oracle.c, {el:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="clouds per GPU")
    ap.add_argument("--points", type=int, default=2048)
    ap.add_argument("--grid", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    pkg = importlib.import_module("2dimageto3dmodel_amd")
    ops = pkg.ops
    B, N, S = args.batch, args.points, args.grid
    pc, q, sc, mask = make_batch(B, N, S, 1234 + 2 + rank, dev)
    pc.requires_grad_()
    q.requires_grad_()
    sc.requires_grad_()
    elf = pkg.EffectiveLossFunction(voxel_size=S).to(dev)
    crit = pkg.SupervisedLoss()

    def step():
        pc.grad = q.grad = sc.grad = None
        proj = elf(pc, q, sc)
        loss = crit(proj, mask)["full_loss"]
        loss.backward()
        return loss

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()

    # ---- per-kernel HIP-event timing of the same K steps (events on the launch stream = torch's current
    #      stream); a separate pass so that the event markers do not sit inside the timed region above
    ops.enable_kernel_timers(True)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    ktimes = ops.collect_kernel_timers()  # name -> (count, total_ms)
    ops.enable_kernel_timers(False)

    if rank == 0:
        value = world * B * args.steps / dt
        # dominant kernel + algorithmic bytes per launch (operator-granular definition)
        dom = max(ktimes, key=lambda k: ktimes[k][1])
        alg = {
            "proj_render_fwd": B * (12 * N + 20 + 8 * S ** 3 + 4 * S ** 2),
            "proj_render_bwd": B * (4 * S ** 2 + 12 * S ** 3 + 24 * N + 20),
        }
        compulsory = {
            "proj_render_fwd": B * (16 * N + 4 * S ** 2 + 4),
            "proj_render_bwd": B * (16 * N + 4 * S ** 2 + 48 * N + 4),
        }
        cnt, tot = ktimes[dom]
        avg_s = tot / cnt / 1e3
        ach = alg.get(dom, 0) / avg_s / 1e9
        out = {
            "metric": "train-step samples/sec (proj+loss fwd/bwd)", "value": value, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"projection+silhouette-loss fwd/bwd, batch {B}/GPU, {N}-pt clouds -> {S}x{S} "
                                   f"(BASELINE configs[1] shape at the metric's batch)",
                       "global_batch": world * B, "points": N, "grid": S, "parallelism": f"dp{world}",
                       "loss": float(loss.item())},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": None,
                         "avg_kernel_us": avg_s * 1e6,
                         "algorithmic_bytes": alg.get(dom, 0),
                         "compulsory_io_bytes": compulsory.get(dom, 0),
                         "note": "fused kernel keeps the S^3 volume in LDS: algorithmic (volume-based) bytes "
                                 "exceed what actually moves, so frac may exceed 1 (SURVEY.md 8d)"},
            "kernels_us": {k: v[1] / v[0] * 1e3 for k, v in ktimes.items()},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, S)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
