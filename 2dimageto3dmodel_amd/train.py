"""The GAN training iteration of code/main.py (ModelWrapper.forward :476-526 and the loop :691-723) on the drop-in
modules: schedule 1 generator step : d_steps_per_g discriminator steps, Adam(betas=(0,0.9)), EMA generator.

The mesh smoothness regulariser of the G step (main.py:697-705) is applied when the trainer is given a
`mesh_template` (2dimageto3dmodel_amd.mesh.MeshTemplate).  Not reproduced: the text encoder."""
import math
import os
import warnings

import torch

from . import conv as CV
from . import gan as G
from . import gan_ops as O
from . import parallel as P


def divide_pred(pred):
    """code/main.py:414-422: split the [fake; real] batch of every discriminator output; a missing list (None) and
    missing entries (the masks when args.mask_output is off) pass through as None"""
    if pred is None:
        return None, None
    if isinstance(pred, list):
        fake = [t[:t.shape[0] // 2] if t is not None else None for t in pred]
        real = [t[t.shape[0] // 2:] if t is not None else None for t in pred]
        return fake, real
    return pred[:pred.shape[0] // 2], pred[pred.shape[0] // 2:]


def ema_alpha(alpha, epoch):
    """code/main.py:433-438: the running-average generator follows faster during the first epochs"""
    if epoch < 10:
        return math.pow(alpha, 100)
    if epoch < 100:
        return math.pow(alpha, 10)
    return alpha


# which of the three spectral-norm prefetch sites are taken (bit 0: the discriminator's chain under the G step's backward, bit 1: the
# generator's under the D step, bit 2: the discriminator's after its optimiser step): A/B switch, all on by default
_SN_SITES = int(os.environ.get("M355_SN_PREFETCH_SITES", "7"))

class GanTrainer(torch.nn.Module):
    def __init__(self, args, latent_dim=64, lr_g=1e-4, lr_d=4e-4, d_steps_per_g=2, loss="hinge", device="cuda",
                 symmetric_g=True, use_mesh=True, ema_alpha=0.999, capturable=False, mesh_template=None,
                 mesh_regularization=0.0001):
        super().__init__()
        # main.py:152,697-705: flat_loss = loss_flat(mesh, compute_normals(get_vertex_positions(pred_mesh))) joins the G loss
        self.mesh_template, self.mesh_regularization = (mesh_template if use_mesh else None), mesh_regularization
        self.args, self.latent_dim, self.d_steps_per_g, self.ema_alpha = args, latent_dim, d_steps_per_g, ema_alpha
        # main.py:541-545: the discriminator is constructed first (it is an argument of ModelWrapper(...)), then the two
        # generators -- same order here, so a run under the same torch.manual_seed starts from the reference's weights
        discriminator = G.MultiScaleDiscriminator(args, 4)
        self.generator = G.Generator(args, latent_dim, symmetric=symmetric_g, mesh_head=use_mesh)
        # main.py:453-457: a SECOND Generator(...) is instantiated (drawing its own initial weights from the global RNG, as the
        # reference does -- so the RNG stream after construction is the reference's) and then overwritten with the first one's
        self.generator_running_avg = G.Generator(args, latent_dim, symmetric=symmetric_g, mesh_head=use_mesh)
        self.generator_running_avg.load_state_dict(self.generator.state_dict())
        for p in self.generator_running_avg.parameters():
            p.requires_grad = False
        self.discriminator = discriminator
        self.criterion_gan = G.GANLoss(loss)
        self.to(device)
        for m in (self.generator, self.generator_running_avg, self.discriminator):
            P.broadcast_parameters(m)
        # betas=(0, 0.9) as main.py:588-589 (floats: torch >= 2.10 rejects the int/float mix)
        # capturable: the step counters live on the device, so a whole cycle can be recorded into a hipGraph
        # fused: one multi-tensor kernel per step instead of ~10 foreach kernels (same update rule, main.py:588-589)
        fused = torch.device(device).type == "cuda" and not os.environ.get("M355_NO_FUSED_ADAM")
        self.optimizer_g = torch.optim.Adam(self.generator.parameters(), lr=lr_g, betas=(0.0, 0.9), capturable=capturable,
                                            fused=fused)
        self.optimizer_d = torch.optim.Adam(self.discriminator.parameters(), lr=lr_d, betas=(0.0, 0.9), capturable=capturable,
                                            fused=fused)
        # data parallel: the generator's gradients travel as TWO messages.  The convs of blk3a .. conv_final and of the mesh head are
        # complete when the backward pass reaches the trunk (gan_ops.GradBarrier in Generator.forward): their all-reduce is issued
        # there and runs under the backward of blk2 / blk1 / fc; everything else (fc, blk1, blk2, the embeddings and every conditioning
        # Linear -- one batched GEMM whose backward runs last) is complete only at the end.  Same averages as one flat message.
        early = []
        for name, m in self.generator.named_children():
            if name in ("blk3a", "blk3b", "blk3c", "blk4", "blk5", "blk6", "blk3_mesh"):
                early += [p for c in (m.conv1, m.conv2, m.shortcut) for p in c.parameters()]
            elif name in ("conv_final", "conv_mesh"):
                early += list(m.parameters())
        early_ids = {id(p) for p in early}
        late = [p for p in self.generator.parameters() if id(p) not in early_ids]
        self.reduce_g = P.BucketedGradReducer([early, late])
        self.overlap_g = not os.environ.get("M355_NO_G_BUCKETS")
        self.reduce_d = P.FlatGradReducer(self.discriminator.parameters())
        self.total_it = 0
        # data parallel: the discriminator's gradient all-reduce (14 MB) is issued asynchronously after its backward and awaited
        # -- followed by optimizer_d.step() -- only where the NEXT iteration needs the discriminator: after that iteration's
        # generator forward, which depends on no discriminator weight and overlaps the collective.  Identical results (the step is
        # applied before anything reads the weights); finish_pending() completes it on demand (checkpoints, end of a run).
        self.overlap_comm = True
        self._pending_d = False
        self._epoch, self._epoch_set = 0, False   # the caller's epoch counter (main.py:668); only the running-average ramp reads it
        self.capturable = capturable

    def _d_weight(self):
        a = self.args
        return [2, 1] if a.num_discriminators == 2 and a.texture_resolution >= 512 else None   # main.py:486-489

    def forward(self, mode, X_tex, X_alpha, X_mesh=None, C=None, caption=None, noise=None):
        """ModelWrapper.forward (main.py:476-526)"""
        assert mode in ['g', 'd', 'inference']
        if noise is None:
            noise = torch.randn((X_alpha.shape[0], self.latent_dim), device=X_alpha.device)
        w = self._d_weight()
        if mode == 'g':
            pred_tex, pred_mesh = self.generator(noise, C, caption)
            self.finish_pending()                                       # (the previous D step's all-reduce + optimiser step)
            X_fake = O.MaskedInput(pred_tex, X_alpha)                  # cat((pred_tex * X_alpha, X_alpha), dim=1), built in D's loaders
            disc, mask = self.discriminator(X_fake, pred_mesh, C, caption)
            if not self._pending_d:
                self._prefetch_sn(self.discriminator, site=0)          # (for the next D step, under this step's backward)
            loss = self.criterion_gan(disc, True, for_discriminator=False, mask=mask, weight=w)
            return loss, pred_tex, pred_mesh
        if mode == 'd':
            with torch.no_grad():
                pred_tex, pred_mesh = self.generator(noise, C, caption)
                # (the generator is not updated before its next forward -- the next D step's, or the next cycle's G step's)
                self._prefetch_sn(self.generator, cross_cycle=(self.total_it + 1) % (1 + self.d_steps_per_g) == 0, site=1)
                assert (X_mesh is None) == (pred_mesh is None)
                # cat((cat((pred_tex * X_alpha, X_alpha), 1), cat((X_tex, X_alpha), 1)), 0) in one pass
                X_comb = O.MaskedInput(pred_tex, X_alpha, X_tex)
                C_comb = torch.cat((C, C), dim=0) if C is not None else None
                M_comb = torch.cat((pred_mesh, X_mesh), dim=0) if pred_mesh is not None else None
            self.finish_pending()                                       # (the previous D step's all-reduce + optimiser step)
            cap_comb = [torch.cat((t, t), dim=0) for t in caption] if caption is not None else None
            disc, mask = self.discriminator(X_comb, M_comb, C_comb, cap_comb)
            # divide_pred + the two criterion calls of main.py:516-519
            loss_fake, loss_real = self.criterion_gan.d_losses(disc, mask, w)
            return loss_fake, loss_real, pred_tex, pred_mesh
        with torch.no_grad():
            return self.generator_running_avg(noise, C, caption, return_attention=True)

    def _early_grads_ready(self):
        """(from the generator's backward, gan_ops.GradBarrier) the trunk's gradient has arrived: run the queued weight-gradient
        epilogues -- they write the .grad of blk3a .. conv_final and the mesh head -- and send those gradients on their way"""
        CV.flush_wgrad_finish()
        self.reduce_g.start_bucket(0)

    def finish_pending(self):
        """complete a discriminator step whose gradient all-reduce is still in flight (see overlap_comm)"""
        if self._pending_d:
            self._pending_d = False
            self.reduce_d.finish()
            self.optimizer_d.step()
            # (total_it already counts the pending step: a multiple of the cycle length = that step closed a cycle, and the forward this
            # prefetch serves belongs to the next one -- not issued inside a hipGraph capture, which must end with every stream joined)
            self._prefetch_sn(self.discriminator, cross_cycle=self.total_it % (1 + self.d_steps_per_g) == 0, site=2)

    # ---- spectral norm ahead of its forward (gan_ops.SpectralNormGroup.prefetch).  A network's power iteration + bf16 weight views
    # depend on its weights and u / v only, so they are issued on a second stream as soon as those are final for the next forward:
    #   discriminators: after optimizer_d.step() (for the next D forward) and, in the G step, right after the D forward (the G step
    #                   does not update D: the step for D1 runs under the G step's backward);
    #   generator:      after the no-grad forward of a D step (no generator update until after the next G forward).  NOT after
    #                   optimizer_g.step(): the running-average update that follows it reads u / v (they are buffers of the
    #                   generator's state_dict, main.py:431-447) and the next forward is the very next thing on the stream.
    # Inside a hipGraph capture the prefetch that would cross into the NEXT cycle is skipped (the graph must end with every stream
    # joined, and the next replay recomputes from the live u / v).
    def _prefetch_sn(self, net, cross_cycle=False, site=0):
        if not (_SN_SITES >> site) & 1:
            return
        if cross_cycle and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            return
        net._sn_group().prefetch(net.training)

    def cancel_sn_prefetch(self):
        """put u / v back where the last FORWARD left them (checkpoints, snapshots, hipGraph capture)"""
        for net in (self.generator, self.discriminator):
            net._sn_group().cancel_prefetch()

    def state_dict(self, *args, **kwargs):
        self.finish_pending()
        self.cancel_sn_prefetch()   # (u / v of a checkpoint are those of the last forward, as the reference's are)
        return super().state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop("_ema_lists", None)   # .to() / .cuda() replace the buffer tensors
        return super()._apply(fn, *args, **kwargs)

    @torch.no_grad()
    def update_generator_running_avg(self, epoch=None):
        """main.py:431-447, including the alpha ramp over the first 100 epochs.  The (dst, src) tensor lists are
        collected once: state_dict() on two 300-entry modules per step is pure host overhead."""
        if epoch is None and not self._epoch_set and not self.__dict__.get("_warned_epoch"):
            self.__dict__["_warned_epoch"] = True
            warnings.warn("GanTrainer: the running-average ramp (main.py:433-438) reads the epoch, which was never given "
                          "(iteration(..., epoch=e) or trainer.epoch = e): staying on the epoch-0 alpha")
        alpha = ema_alpha(self.ema_alpha, self.epoch if epoch is None else epoch)
        ema = self.__dict__.get("_ema_lists")
        if ema is None:
            src = self.generator.state_dict(keep_vars=True)
            fl_dst, fl_src, other = [], [], []
            for k, v in self.generator_running_avg.state_dict(keep_vars=True).items():
                if torch.is_floating_point(v):
                    fl_dst.append(v)
                    fl_src.append(src[k])
                else:
                    other.append((v, src[k]))
            ema = self.__dict__["_ema_lists"] = (fl_dst, fl_src, other)
        fl_dst, fl_src, other = ema
        if other:   # (the integer buffers -- num_batches_tracked of every batch norm: one launch, not one device copy per layer)
            torch._foreach_copy_([v for v, _ in other], [sv for _, sv in other])
        torch._foreach_mul_(fl_dst, alpha)
        torch._foreach_add_(fl_dst, fl_src, alpha=1 - alpha)

    @property
    def epoch(self):
        return self._epoch

    @epoch.setter
    def epoch(self, e):
        self._epoch, self._epoch_set = int(e), True

    def capture_cycle(self, batches, epoch=None, warmup=2, noises=None, restore_after_warmup=True):
        """One training cycle (1 G step + d_steps_per_g D steps, optimiser steps and the running-average update included) as ONE
        hipGraph: returns a CycleGraph whose replay() costs one graph launch instead of ~750 kernel launches issued from Python
        (13 ms of host time per cycle regardless of the batch: the limit below batch ~24 per GPU).  Needs
        GanTrainer(capturable=True) (Adam's step counters on the device).  `batches`: 1 + d_steps_per_g loader batches
        (X_tex, X_alpha, X_mesh, C); their storage is copied into static buffers that CycleGraph.load() refills.  noises: one
        fixed latent batch per iteration (tests); default: fresh torch.randn noise on every replay.
        warmup: cycles run before the capture so that every allocation, lazily built table and optimiser state exists; they are
        real training cycles on `batches`, and with restore_after_warmup (default) everything they changed is put back -- the
        trainer is at the same step, with the same weights and the same RNG position, after capture_cycle() as before it."""
        return CycleGraph(self, batches, epoch, warmup, noises, restore_after_warmup)

    def iteration(self, X_tex, X_alpha, X_mesh, C, caption=None, noise=None, epoch=None):
        """one pass of the loop body main.py:691-723 on one loader batch; returns the scalar losses.  `noise` fixes the
        latent batch (ModelWrapper.forward's own argument); `epoch` overrides self.epoch for the running-average ramp."""
        if self.total_it % (1 + self.d_steps_per_g) == 0:
            self.optimizer_g.zero_grad(set_to_none=True)
            # The G step only needs dL/d(input) from the discriminator: the reference lets autograd also compute D's
            # weight gradients and then throws them away (optimizer_d.zero_grad(), code/main.py:716; SURVEY 8a G-bwd).
            # Switching requires_grad off for the duration skips those wgrad kernels; the generator's gradients are
            # bit-identical.
            d_params = [p for p in self.discriminator.parameters() if p.requires_grad]
            for p in d_params:
                p.requires_grad_(False)
            self.generator.grad_barrier = self._early_grads_ready if (self.overlap_g and P.collectives_on()) else None
            try:
                loss, _, pred_mesh = self('g', None, X_alpha, None, C, caption, noise)
                loss = loss.mean()
                flat = None
                if self.mesh_template is not None and pred_mesh is not None:
                    from . import mesh as M
                    vtx = self.mesh_template.get_vertex_positions(pred_mesh)
                    flat = M.loss_flat(self.mesh_template.mesh, self.mesh_template.compute_normals(vtx))
                    with CV.deferred_wgrad_finish():   # (gradients are first read by the reducer / optimiser below)
                        (loss + self.mesh_regularization * flat).backward()
                else:
                    with CV.deferred_wgrad_finish():
                        loss.backward()
            finally:
                self.generator.grad_barrier = None
                for p in d_params:
                    p.requires_grad_(True)
            self.reduce_g()
            self.optimizer_g.step()
            self.update_generator_running_avg(epoch)
            out = {"g": loss.detach()}
            if flat is not None:
                out["flat"] = flat.detach()
        else:
            loss_fake, loss_real, _, _ = self('d', X_tex, X_alpha, X_mesh, C, caption, noise)
            loss_fake, loss_real = loss_fake.mean(), loss_real.mean()
            self.optimizer_d.zero_grad(set_to_none=True)   # (behind the forward: a pending step of the previous iteration is done)
            with CV.deferred_wgrad_finish():
                (loss_fake + loss_real).backward()
            if self.overlap_comm and self.reduce_d.start():
                self._pending_d = True                     # finished after the next iteration's generator forward
            else:
                self.reduce_d()
                self.optimizer_d.step()
                self._prefetch_sn(self.discriminator, cross_cycle=(self.total_it + 1) % (1 + self.d_steps_per_g) == 0, site=2)
            out = {"d_fake": loss_fake.detach(), "d_real": loss_real.detach()}
        self.total_it += 1
        return out


class CycleGraph:
    """A captured training cycle of a GanTrainer (GanTrainer.capture_cycle).  What is baked into the graph: the tensor
    addresses of the static input buffers, the spectral-norm slots the cycle's six network forwards rotate through, the
    running-average alpha of the epoch it was captured at (replay(epoch=...) re-captures when the ramp of main.py:433-438
    moves to another value), the batch shapes.  What stays live across replays: parameters, optimiser state, spectral-norm
    vectors, batch-norm running statistics, and the latent noise -- torch.randn inside a capture draws from the graph-safe
    Philox state, so every replay sees fresh noise.  The returned losses are tensors of the graph's memory pool, overwritten
    by every replay.
    Capturing has NO training side effect: the `warmup` cycles torch's capture protocol needs (they are real optimisation steps
    on `batches`) are undone -- weights, optimiser state, running statistics, spectral-norm vectors, the running-average
    generator, total_it and the RNG are restored in place before the capture (restore_after_warmup=False keeps them, i.e. the
    trainer then starts `warmup` cycles ahead).  replay(epoch=e) re-captures when the running-average ramp moves to another
    alpha: a capture pass executes nothing."""

    def __init__(self, trainer, batches, epoch=None, warmup=2, noises=None, restore_after_warmup=True):
        self.restore_after_warmup = restore_after_warmup
        if not trainer.capturable:
            raise RuntimeError("capture_cycle needs GanTrainer(capturable=True): Adam's step counters must live on the device")
        n = 1 + trainer.d_steps_per_g
        if len(batches) != n:
            raise ValueError(f"capture_cycle: {n} loader batches per cycle (1 G step + {trainer.d_steps_per_g} D steps), got {len(batches)}")
        if trainer.total_it % n:
            raise RuntimeError("capture_cycle: the trainer is in the middle of a cycle")
        self.trainer, self.n = trainer, n
        self.static = [[None if t is None else t.clone() for t in b] for b in batches]
        self.noises = [None] * n if noises is None else [z.clone() for z in noises]
        self.epoch = trainer.epoch if epoch is None else epoch
        self.alpha = ema_alpha(trainer.ema_alpha, self.epoch)
        self.out = None
        self._capture(warmup)

    def _run(self):
        out = {}
        for b, z in zip(self.static, self.noises):
            out.update(self.trainer.iteration(*b, noise=z, epoch=self.epoch))
        self.trainer.finish_pending()   # a captured cycle is self-contained: the last D step's all-reduce + optimiser step inside
        return out

    def _snapshot(self):
        """everything a training cycle changes on the device, cloned: parameters and buffers of the three networks (weights, batch-norm
        running statistics, spectral-norm vectors, the running-average generator), the state of both optimisers, the RNG"""
        tr = self.trainer
        tensors = [t for m in (tr.generator, tr.generator_running_avg, tr.discriminator)
                   for t in list(m.parameters()) + list(m.buffers())]
        opt = []
        for o in (tr.optimizer_g, tr.optimizer_d):
            for group in o.param_groups:
                for p in group["params"]:
                    st = o.state.get(p)
                    opt.append((o, p, None if not st else {k: v.clone() for k, v in st.items() if torch.is_tensor(v)}))
        return tensors, [t.detach().clone() for t in tensors], opt, tr.total_it, torch.cuda.get_rng_state(tensors[0].device)

    def _restore(self, snap):
        """put the snapshot back IN PLACE (the addresses are what the capture records): optimiser state that did not exist before
        the warm-up is zeroed -- Adam's initial state -- rather than deleted, so that the capture pass does not allocate it"""
        tensors, values, opt, total_it, rng = snap
        with torch.no_grad():
            for t, v in zip(tensors, values):
                t.copy_(v)
            for o, p, saved in opt:
                st = o.state.get(p)
                if not st:
                    continue
                for k, v in st.items():
                    if torch.is_tensor(v):
                        v.copy_(saved[k]) if saved is not None else v.zero_()
        self.trainer.total_it = total_it
        torch.cuda.set_rng_state(rng, tensors[0].device)

    def _capture(self, warmup):
        # warm-up on a side stream (torch's capture protocol): allocator pools, the weight-gradient arena, the lazily uploaded
        # positional planes and the optimiser state all exist before the capture starts.  The warm-up cycles are REAL training
        # cycles on the supplied batches; what they did to the weights, the optimiser state, the running statistics, total_it and
        # the RNG is undone afterwards (restore_after_warmup), so capturing at step k leaves the trainer at step k.
        # a discriminator step left pending by eager iterations before this capture (overlap_comm) is completed NOW, once and for
        # real: inside the warm-up it would be rolled back with the snapshot (one update silently lost), and on the re-capture
        # path (warmup = 0) its finish() / step() would be recorded into the graph instead of executed
        self.trainer.finish_pending()
        self.trainer.cancel_sn_prefetch()   # (a spectral-norm step computed ahead by eager iterations: the snapshot is of the last FORWARD's u / v)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            snap = self._snapshot() if (warmup and self.restore_after_warmup) else None
            for _ in range(warmup):
                self._run()
            self.trainer.cancel_sn_prefetch()   # (the warm-up's last D step prefetched for a G step the capture must compute itself)
            if snap is not None:
                self._restore(snap)
        torch.cuda.current_stream().wait_stream(side)
        import os
        import time
        # (... and a pause first: synchronize, then let the watchdog -- it wakes every 100 ms -- retire the warm-up's work objects, so
        # that its list is empty while the capture is open; a death with "event last recorded in a capturing stream" was seen once
        # in 16 thread-local probes without it)
        settle = float(os.environ.get("M355_CAPTURE_SETTLE_MS", "400")) * 1e-3
        if settle > 0 and P.collectives_on():
            torch.cuda.synchronize()
            time.sleep(settle)
        # With collectives in the cycle the capture is THREAD-LOCAL: ProcessGroupNCCL's watchdog thread polls the events of the
        # warm-up's collectives (hipEventQuery) whenever it wakes up, and in the default global mode such a call from ANY thread
        # while a capture is open is an error -- the watchdog then throws and the process aborts (seen in 3 of ~40 probe runs:
        # "operation not permitted when stream is capturing", gpurun_out/rccl_graph_probe_death.txt).  Thread-local mode only
        # polices the capturing thread.
        mode = os.environ.get("M355_CAPTURE_MODE") or ("thread_local" if P.collectives_on() else "global")
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode=mode):
            self.out = self._run()
        self.trainer.total_it -= self.n   # (the capture pass executed nothing; it only rotated the spectral-norm slots on the host)

    def load(self, batches):
        """copy the next cycle's loader batches into the static buffers (device-to-device, on the current stream)"""
        for dst, src in zip(self.static, batches):
            for d, t in zip(dst, src):
                if d is not None:
                    d.copy_(t, non_blocking=True)

    def replay(self, batches=None, epoch=None):
        if epoch is not None and ema_alpha(self.trainer.ema_alpha, epoch) != self.alpha:
            self.epoch, self.alpha = epoch, ema_alpha(self.trainer.ema_alpha, epoch)
            self._capture(0)          # the ramp moved: the new alpha has to be baked in
        if batches is not None:
            self.load(batches)
        self.graph.replay()
        self.trainer.total_it += self.n
        return self.out
