"""Data parallelism for the GAN half: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Replaces the reference's single-process nn.DataParallel + DataParallelWithCallback (code/main.py:530-548,
code/sync_batchnorm/replicate.py:50-67): parameters stay resident on every rank (no per-iteration Broadcast),
inputs are sharded by the loader (no scatter/gather), gradients are averaged by ONE flat all-reduce per optimiser
step (G 47 MB / D 14 MB fp32 at 256^2: a single large message uses all 7 xGMI links, whereas many small buckets
would be latency bound), and SyncBN statistics are one fused [sum|sumsq|count] all-reduce per layer
(gan_ops._SyncMoments).  The projection path has no parameters and needs no collective at all.
"""
import os

import torch
import torch.distributed as dist


DEFAULT_MASTER_PORT = 29533   # rendezvous port of hand-launched ranks when MASTER_PORT is unset (launchers always set it)

# collective bookkeeping (bench.py reports it): how many collectives the GAN path issued and, when `time_allreduce` is
# on, device/host time of the gradient all-reduces
stats = {"grad_allreduces": 0, "grad_allreduce_bytes": 0, "syncbn_collectives": 0, "time_allreduce": False, "events": []}


def reset_stats(time_allreduce=False):
    stats.update(grad_allreduces=0, grad_allreduce_bytes=0, syncbn_collectives=0, time_allreduce=bool(time_allreduce))
    stats["events"] = []


def allreduce_ms():
    """total time of the gradient all-reduces recorded since reset_stats(time_allreduce=True); sync the device first"""
    return sum(e if isinstance(e, float) else e[0].elapsed_time(e[1]) for e in stats["events"])


def init_from_env(device_type="cuda", force=False):
    """torch.distributed.run contract: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.
    device_type "cuda" -> backend "nccl" (RCCL over xGMI on ROCm), bound to cuda:LOCAL_RANK; "cpu" -> gloo (tests).
    force: initialise the process group even for a single rank (exercises the RCCL code path on a 1-GPU box)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            # the launcher (torch.distributed.run, bench.py's respawn) owns the rendezvous port; the launcher-less single-rank
            # path (force: the 1-GPU RCCL smoke test) picks a free one itself -- two jobs on a host must not collide
            if world > 1:
                # hand-launched ranks (RANK / WORLD_SIZE exported by a SLURM-style wrapper): every rank must agree on the port
                # without talking to each other, so it is a documented constant; two such jobs on one host
                # need MASTER_PORT set explicitly
                os.environ["MASTER_PORT"] = str(DEFAULT_MASTER_PORT)
            else:
                import socket
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if device_type == "cuda":
            if local_rank >= torch.cuda.device_count():
                raise RuntimeError(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible "
                                   "(one process per GPU)")
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    return rank, local_rank, world


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def collectives_on():
    """True when the data-parallel collectives must run: more than one rank, or M355_FORCE_COLLECTIVES=1 with an
    initialised group (single-rank RCCL smoke test: same calls, trivially correct result)"""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("M355_FORCE_COLLECTIVES", "") == "1"   # ("0" / "false" do not force)


def broadcast_parameters(module, src=0):
    """make every rank start from rank `src`'s parameters and buffers (the reference's replicate() does this
    on every forward; here once)"""
    if not collectives_on():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)


_GRAD_GROUP = [None]


def grad_group():
    """A process group (= an RCCL communicator with its own stream) for the gradient all-reduces alone: a collective queued on
    the default group's stream runs behind everything queued there before it, so a flat all-reduce left in flight while the next
    generator forward starts would hold up that forward's first SyncBN all-reduce.  Created lazily BY EVERY RANK at the same
    point (the first reducer call); single-rank forced mode uses it too."""
    if _GRAD_GROUP[0] is None and dist.is_available() and dist.is_initialized():
        _GRAD_GROUP[0] = dist.new_group(ranks=list(range(dist.get_world_size())))
    return _GRAD_GROUP[0]


class FlatGradReducer:
    """Averages the gradients of `params` over the ranks with one all-reduce of a persistent flat buffer.
    reducer() = start() + finish(); start() issues the collective asynchronously on the gradient group's communicator (the
    compute stream is not blocked), finish() makes the compute stream wait for it and writes the averages back -- the trainer
    puts the next generator forward between the two (train.GanTrainer: overlap of the discriminator's all-reduce)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.flat = None
        self._pending = None

    def start(self):
        if not collectives_on():
            return False
        ps = [p for p in self.params if p.grad is not None]
        n = sum(p.numel() for p in ps)
        if self.flat is None or self.flat.numel() != n or self.flat.device != ps[0].device:
            self.flat = torch.empty(n, dtype=torch.float32, device=ps[0].device)
        views, off = [], 0
        for p in ps:
            views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        torch._foreach_copy_(views, [p.grad for p in ps])
        timed = stats["time_allreduce"]
        ev = t0 = None
        if timed and self.flat.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
        elif timed:
            import time
            t0 = time.perf_counter()
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=grad_group(), async_op=True)
        self._pending = (work, ps, views, ev, t0, n)
        return True

    def finish(self):
        if self._pending is None:
            return
        work, ps, views, ev, t0, n = self._pending
        self._pending = None
        work.wait()   # (RCCL: the CURRENT STREAM waits for the collective; gloo: the host does)
        timed = stats["time_allreduce"]
        if timed and len(stats["events"]) >= 4096:   # a measurement window, not a log: never grows without bound
            del stats["events"][:2048]
        if ev is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()   # behind the wait: start-to-completion as the compute stream sees it (includes what overlapped)
            stats["events"].append((ev, e1))
        elif t0 is not None:
            import time
            stats["events"].append((time.perf_counter() - t0) * 1e3)
        stats["grad_allreduces"] += 1
        stats["grad_allreduce_bytes"] += 4 * n
        self.flat.mul_(1.0 / world_size())
        torch._foreach_copy_([p.grad for p in ps], views)

    def __call__(self):
        if self.start():
            self.finish()


# ------------------------------------------------------------------------------------------------ projection path
class ImageShard:
    """what rank `rank` of `world` owns of a batch of `n_images` source images with `group` candidate clouds each"""
    __slots__ = ("img_lo", "img_hi", "cloud_lo", "cloud_hi", "group", "weight")

    def __init__(self, img_lo, img_hi, group, weight):
        self.img_lo, self.img_hi, self.group, self.weight = img_lo, img_hi, group, weight
        self.cloud_lo, self.cloud_hi = img_lo * group, img_hi * group

    @property
    def n_images(self):
        return self.img_hi - self.img_lo

    def images(self, t):
        """the rank's rows of a per-image tensor ([n_images, ...]: masks, student poses)"""
        return t[self.img_lo:self.img_hi]

    def clouds(self, t):
        """the rank's rows of a per-cloud tensor ([n_images * group, ...], candidate index fastest: point clouds, projections,
        ensemble poses -- the layout `projection_loss.view(-1, K)` of unsupervised_part.py:117 assumes)"""
        if t.shape[0] % self.group:
            raise ValueError(f"per-cloud tensor has {t.shape[0]} rows, not a multiple of the {self.group} candidates per image")
        return t[self.cloud_lo:self.cloud_hi]


def shard_by_image(n_images, group=1, rank=None, world=None):
    """the projection path shards by SOURCE IMAGE, never by cloud -- the `group` = K (x V views) candidate clouds of one
    image must meet on one rank, because UnsupervisedLoss takes the argmin of their silhouette losses per image
    (/root/reference/code/models/unsupervised_part.py:117-126: `.view(-1, K)`, `argmin(dim=-1)`); a cloud-granular split would
    put candidates of one image on two ranks and silently turn the argmin over K into two argmins over fewer.  Contiguous image
    blocks, the first n_images % world ranks one image larger.  The path has no parameters and no collective of its own: the
    encoder / pose-decoder gradients that flow out of it are averaged by the caller's FlatGradReducer.  `weight` scales a loss
    that is a MEAN over the rank's images (every loss of the path is, unsup:120,134 / sup:72) so that the all-reduce MEAN of the
    ranks' gradients is the gradient of the global-batch mean even when the blocks are uneven: weight = n_local * world / n."""
    if rank is None:
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    if world is None:
        world = world_size()
    if n_images < 0 or group < 1 or not (0 <= rank < world):
        raise ValueError(f"shard_by_image: n_images={n_images}, group={group}, rank={rank}, world={world}")
    q, r = divmod(n_images, world)
    lo = rank * q + min(rank, r)
    hi = lo + q + (1 if rank < r else 0)
    return ImageShard(lo, hi, group, (hi - lo) * world / n_images if n_images else 0.0)


def check_image_groups(n_clouds, n_images, group):
    """raise unless a rank's local batch holds WHOLE candidate groups (what a loader that shards by cloud would violate)"""
    if n_clouds != n_images * group:
        raise ValueError(f"projection shard holds {n_clouds} clouds for {n_images} images x {group} candidates: the candidates of "
                         "an image must stay on one rank (parallel.shard_by_image)")
