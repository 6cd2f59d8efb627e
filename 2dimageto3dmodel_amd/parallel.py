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
stats = {"grad_allreduces": 0, "grad_allreduce_bytes": 0, "syncbn_collectives": 0, "time_allreduce": False, "events": [],
         "wait_events": []}


def reset_stats(time_allreduce=False):
    stats.update(grad_allreduces=0, grad_allreduce_bytes=0, syncbn_collectives=0, time_allreduce=bool(time_allreduce))
    stats["events"] = []
    stats["wait_events"] = []


def _ms(events):
    return sum(e if isinstance(e, float) else e[0].elapsed_time(e[1]) for e in events)


def allreduce_ms():
    """total time of the gradient all-reduces recorded since reset_stats(time_allreduce=True), issue to completion as the compute
    stream sees it (INCLUDES whatever compute overlapped them); sync the device first"""
    return _ms(stats["events"])


def allreduce_exposed_ms():
    """... and the part of it the compute stream actually spent WAITING (the span of finish()'s wait on that stream): what the
    collectives cost the step; allreduce_ms() - allreduce_exposed_ms() ran under compute"""
    return _ms(stats["wait_events"])


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
        timed = stats["time_allreduce"]
        w0 = tw = None
        if timed and ev is not None:
            w0 = torch.cuda.Event(enable_timing=True)
            w0.record()   # (in front of the wait: [w0, e1] is what the compute stream spends blocked on the collective)
        elif timed:
            import time
            tw = time.perf_counter()
        work.wait()   # (RCCL: the CURRENT STREAM waits for the collective; gloo: the host does)
        for key in ("events", "wait_events"):
            if timed and len(stats[key]) >= 4096:   # a measurement window, not a log: never grows without bound
                del stats[key][:2048]
        if ev is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()   # behind the wait: start-to-completion as the compute stream sees it (includes what overlapped)
            stats["events"].append((ev, e1))
            stats["wait_events"].append((w0, e1))
        elif t0 is not None:
            import time
            now = time.perf_counter()
            stats["events"].append((now - t0) * 1e3)
            stats["wait_events"].append((now - tw) * 1e3)
        stats["grad_allreduces"] += 1
        stats["grad_allreduce_bytes"] += 4 * n
        self.flat.mul_(1.0 / world_size())
        torch._foreach_copy_([p.grad for p in ps], views)

    def __call__(self):
        if self.start():
            self.finish()


# ------------------------------------------------------------------------------------------------ SyncBN messages without RCCL
class IpcAllReduce:
    """The small-message all-reduce of csrc/ipc_exchange.hip (include/m355.h m355_ipc_*): every rank's fine-grained device region is
    mapped into every peer (hipIpc handles exchanged ONCE through torch.distributed, whatever its backend), after which a message is
    one kernel launch on the compute stream -- publish with a release-stored sequence flag, acquire-spin on the peers' flags, add in
    rank order (bit-equal on all ranks).  Opt-in (M355_SYNCBN_IPC=1): the default SyncBN message is an RCCL all-reduce, which is what an
    N > 1 box has to confirm first; this path has been exercised with two processes sharing ONE GPU (tests/test_distributed_gpu.py),
    where the peers' "xGMI" reads are local.  Every in-kernel wait is bounded (`timeout_ms`): a rank that does not show up raises a bit
    in the status word, which check() turns into an exception -- never a hung GPU."""

    def __init__(self, device, timeout_ms=2000):
        import ctypes
        from ._lib import check, lib
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("IpcAllReduce needs an initialised process group (the handles travel through it once)")
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        L = lib()
        if self.world > 16:
            raise RuntimeError("IpcAllReduce: at most M355_IPC_MAX_RANKS = 16 ranks (one node)")
        self.device = torch.device(device)
        self.timeout_ms = int(timeout_ms)
        self.max_floats = int(L.m355_ipc_max_floats())
        self.n_channels = int(L.m355_ipc_channels())
        self._channel_of = {}    # call-site key -> channel, in order of first use (= program order, the same on every rank)
        with torch.cuda.device(self.device):
            mine, handle = ctypes.c_void_p(), (ctypes.c_char * 64)()
            check(L.m355_ipc_alloc(ctypes.byref(mine), handle), "ipc_alloc")
            handles = [None] * self.world
            dist.all_gather_object(handles, bytes(handle.raw))
            self._own = mine.value
            regions = []
            for r, h in enumerate(handles):
                if r == self.rank:
                    regions.append(self._own)
                    continue
                p = ctypes.c_void_p()
                check(L.m355_ipc_open(ctypes.create_string_buffer(h, 64), ctypes.byref(p)), f"ipc_open (rank {r})")
                regions.append(p.value)
            self._regions = (ctypes.c_void_p * self.world)(*regions)
            self.status = torch.zeros(1, dtype=torch.int32, device=self.device)
        dist.barrier()   # (every region is mapped everywhere before the first message)
        self.messages = 0

    def channel(self, key):
        """the channel of a call site (None: all M355_IPC_CHANNELS are taken -- the caller falls back to torch.distributed).  Channels are
        handed out in order of first use: the host program order, identical on every rank."""
        ch = self._channel_of.get(key)
        if ch is None and len(self._channel_of) < self.n_channels:
            ch = self._channel_of[key] = len(self._channel_of)
        return ch

    def __call__(self, vec, channel=0):
        """vec: contiguous fp32 device vector, summed over the ranks IN PLACE (asynchronous on the current stream).  Messages of one
        channel pair up in order across the ranks; different channels are independent (they may run on different streams)."""
        from ._lib import launch, ptr, stream
        assert vec.is_cuda and vec.dtype == torch.float32 and vec.is_contiguous() and vec.numel() <= self.max_floats, \
            (vec.dtype, vec.numel(), self.max_floats)
        launch("ipc_allreduce", ptr(vec), vec.numel(), self._regions, self.rank, self.world, int(channel), ptr(self.status),
               self.timeout_ms, stream())
        self.messages += 1
        return vec

    def check(self):
        """raise if any message since the last check timed out waiting for a peer (synchronises the device)"""
        st = int(self.status.item())
        if st:
            self.status.zero_()
            raise RuntimeError(f"IpcAllReduce: a peer did not arrive within {self.timeout_ms} ms (status bits {st:#x}: 1 = slot not "
                               "released, 2 = message missing) -- the ranks issued different message sequences, or one of them died")

    def close(self):
        from ._lib import lib
        L = lib()
        for r in range(self.world):
            if r != self.rank and self._regions[r]:
                L.m355_ipc_close(self._regions[r])
        if self._own:
            L.m355_ipc_free(self._own)
        self._own = None


_IPC = {}


def syncbn_ipc(device):
    """the IpcAllReduce of `device` when the SyncBN messages are to travel through it (M355_SYNCBN_IPC=1 and > 1 rank), else None ->
    torch.distributed (RCCL).  Created lazily BY EVERY RANK at the same point: the first SyncBN forward."""
    if os.environ.get("M355_SYNCBN_IPC", "") != "1" or not collectives_on() or world_size() < 2:
        return None
    key = torch.device(device).index
    ex = _IPC.get(key)
    if ex is None:
        ex = _IPC[key] = IpcAllReduce(device, int(os.environ.get("M355_IPC_TIMEOUT_MS", "2000")))
    return ex


def syncbn_all_reduce(vec, site=None):
    """SUM over the ranks of a SyncBN statistics vector, in place: the one-launch exchange when switched on (`site`: a hashable key of
    the call site -- layer and direction -- that picks the exchange's channel), else an RCCL all-reduce"""
    ex = syncbn_ipc(vec.device) if (vec.is_cuda and site is not None) else None
    ch = ex.channel(site) if ex is not None else None
    if ch is not None and vec.numel() <= ex.max_floats and vec.is_contiguous():
        ex(vec, ch)
    else:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)


class BucketedGradReducer:
    """FlatGradReducer in `buckets` that become ready at different points of ONE backward pass: bucket k's all-reduce is issued (on
    the gradient communicator, asynchronously) as soon as the caller says its gradients exist -- from inside the backward, through a
    gan_ops.GradBarrier -- and runs under the rest of the pass; reducer() at the end starts what has not been started and finishes
    all.  The averages are the flat reducer's bit for bit at two ranks (an elementwise sum over ranks does not depend on how the
    elements are grouped into messages; tests/test_distributed_cpu.py) -- for more ranks RCCL's reduction order per element may
    differ between message sizes by the rounding of fp32 addition.
    Why only two buckets for the generator (train.GanTrainer): 9.1 M of its 11.75 M parameters (fc, blk1, blk2 and every conditioning
    Linear, whose batched GEMM's backward runs last) finish at the very END of the backward; what is ready early -- the convs of blk3a ..
    conv_final and the mesh head, 2.6 M parameters = 10.5 MB of the 47 MB -- is ready when the trunk's gradient arrives, about 1 ms
    before the end at batch 64: that message leaves the critical path, the 36.5 MB one cannot."""

    def __init__(self, buckets):
        self.reducers = [FlatGradReducer(b) for b in buckets]
        self.started = [False] * len(self.reducers)

    @property
    def params(self):
        return [p for r in self.reducers for p in r.params]

    def start_bucket(self, k):
        if not self.started[k]:
            self.started[k] = bool(self.reducers[k].start())
        return self.started[k]

    def start(self):
        any_ = False
        for k in range(len(self.reducers)):
            any_ = self.start_bucket(k) or any_
        return any_

    def finish(self):
        for k, r in enumerate(self.reducers):
            r.finish()
            self.started[k] = False

    def __call__(self):
        if self.start():
            self.finish()
        else:
            self.started = [False] * len(self.reducers)


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
