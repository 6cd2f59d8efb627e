"""Drop-in GAN modules (reference: code/models/gan.py, code/utils/losses.py, code/rendering/utils.py).

Same class names, constructor arguments, forward signatures, parameter / buffer names and creation order as the
reference, so `state_dict`s are interchangeable key for key and a module built under the same `torch.manual_seed`
starts from the same weights.  What differs is the execution: activations live in NHWC bf16, every convolution
runs as a bf16 MFMA implicit GEMM (csrc/conv_mfma.hip) with the reference's W pads (replicate / circular) and
nearest x2 upsampling folded into its loader; inputs/outputs at the module boundary stay NCHW fp32.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import conv as C
from . import gan_ops as G

LRELU = 0.2  # nn.LeakyReLU(0.2) everywhere in models/gan.py


# ------------------------------------------------------------------------------------------------ helpers
def symmetrize_texture(x):
    """rendering/utils.py:15-18: even symmetry along W (length N -> 2N) of an NCHW tensor"""
    xf = torch.flip(x, (x.dim() - 1,))
    h = xf.shape[3] // 2
    return torch.cat((xf[:, :, :, h:], x, xf[:, :, :, :h]), dim=-1)


def adjust_poles(tex):
    """rendering/utils.py:21-26: replace the first / last row by their mean (mesh displacement map only)"""
    top = tex[:, :, :1].mean(dim=3, keepdim=True).expand(-1, -1, -1, tex.shape[3])
    bottom = tex[:, :, -1:].mean(dim=3, keepdim=True).expand(-1, -1, -1, tex.shape[3])
    return torch.cat((top, tex[:, :, 1:-1], bottom), dim=2)


def circpad(x, amount=1):
    """rendering/utils.py:29-33 (kept for callers; the convolutions here wrap indices instead)"""
    return torch.cat((x[:, :, :, -amount:], x, x[:, :, :, :amount]), dim=3)


def positional_encoding(Ny, Nx):
    """models/gan.py:9-20: cos/sin of the two sphere angles, x wrapping smoothly; half width when symmetric"""
    symmetric = (Nx == Ny // 2)
    n = Ny
    ty = np.linspace(0, np.pi, n, endpoint=False)
    tx = np.linspace(-np.pi, np.pi, n, endpoint=False)
    Y, X = np.meshgrid(tx, ty)
    enc = np.stack((np.cos(X), np.sin(X), np.cos(Y), np.sin(Y)))
    return enc[:, :, n // 4:-(n // 4)] if symmetric else enc


class Conv2d(nn.Conv2d):
    """nn.Conv2d parameters (so spectral_norm, init and state_dict keys are the reference's), MFMA execution.
    Takes / returns NHWC bf16; `pad_w` columns are produced by index arithmetic in the kernel.  When wrapped by
    `spectral_norm` below, the weight is weight_orig / sigma with sigma from the owning network's
    SpectralNormGroup (one batched power iteration per forward) -- or from a private one-layer group when the
    conv is called on its own."""

    def __init__(self, cin, cout, k, stride=1, pad_h=0, pad_w=0, pad_w_mode=C.PAD_ZERO, bias=True):
        super().__init__(cin, cout, k, stride=stride, padding=(pad_h, 0), bias=bias)
        self.m355 = (stride, pad_h, pad_w, pad_w_mode)
        # is this conv called with the nearest x2 upsample folded in?  (its bf16 weight buffers then also carry the sub-pixel views,
        # csrc/conv_mfma.hip `up3`; the owning SpectralNormGroup prepares them with the whole network's.)  Learnt from the calls:
        # a group whose views were built for the other setting serves one forward through the per-call weight_prep and rebuilds.
        self.m355_ups = 0
        # > 0: the input channels beyond the first m355_dx_lead are constants of the model (positional encodings): the backward
        # produces the gradient of the leading channels only (csrc: m355_conv2d_dgrad_lead)
        self.m355_dx_lead = 0
        self._sn_state = None
        self._sn_own = None

    def forward(self, x, upsample=0, slope=1.0, out_f32_nchw=False, in_slope=1.0, premasked=False, want_stats=False, dx_pair=None):
        stride, pad_h, pad_w, mode = self.m355
        self.m355_ups = int(upsample)
        sn, weight = None, None
        if "weight_orig" in self._parameters:
            sn, self._sn_state = self._sn_state, None
            if sn is None:  # not driven by a network-level group
                if self._sn_own is None:
                    self._sn_own = G.SpectralNormGroup([self])
                self._sn_own.step(self.training)
                sn, self._sn_state = self._sn_state, None
            weight = self.weight_orig
        else:
            weight = self.weight
        return G.conv2d(x, weight, self.bias, stride, pad_h, pad_w, mode, upsample, slope, out_f32_nchw, sn, in_slope,
                        premasked, want_stats, self.m355_dx_lead, dx_pair)


def spectral_norm(conv):
    """torch.nn.utils.spectral_norm (same parameter / buffer names, initialisation and state_dict hooks) minus its
    per-module forward pre-hook: sigma comes from the batched kernel (gan_ops.SpectralNormGroup)."""
    return G.strip_sn_hook(nn.utils.spectral_norm(conv))


def sn_convs(module):
    return [m for m in module.modules() if isinstance(m, Conv2d) and "weight_orig" in m._parameters]


class _Identity(nn.Module):
    def forward(self, x):
        return x


# ------------------------------------------------------------------------------------------------ generator
class ConditionalBatchNorm2d(nn.Module):
    """models/gan.py:264-286: BN(affine=False) then x*(1+fc_gamma(z)) + fc_beta(z); here fused with the
    LeakyReLU that always follows it in ResBlockUp."""

    def __init__(self, args, ch, emb_dim):
        super().__init__()
        if args.norm_g == 'syncbatch':
            self.norm = G.SynchronizedBatchNorm2d(ch, affine=False)
        elif args.norm_g == 'batch':
            self.norm = G.BatchNorm2d(ch, affine=False)
        elif args.norm_g == 'instance':
            self.norm = G.InstanceNorm2d(ch)
        elif args.norm_g == 'none':
            self.norm = G.NoNorm()
        else:
            raise ValueError(f"norm_g={args.norm_g!r}")
        self.fc_gamma = nn.Linear(emb_dim, ch)
        self.fc_beta = nn.Linear(emb_dim, ch)

    def forward(self, x, z, slope=1.0, gb=None, res=None, out_slope=1.0, part=None):
        """gb: (gamma, beta) of this layer when the owner evaluated all fc_gamma / fc_beta in one GEMM;
        res: tensor added after the activation (the block's shortcut branch), fused into the same pass;
        out_slope: a LeakyReLU on top of everything whose backward the (single) consumer applies (gan_ops.head_conv);
        part: batch-statistics partial sums of x from the conv launch that produced it (gan_ops.conv2d(want_stats=True))"""
        gamma, beta = gb if gb is not None else (self.fc_gamma(z), self.fc_beta(z))
        return self.norm(x, gamma, beta, slope, res, out_slope, part)


class ResBlockUp(nn.Module):
    """models/gan.py:288-312"""

    def __init__(self, args, ch_in, ch_out, emb_dim, pad_fn):
        super().__init__()
        ch_middle = min(ch_in, ch_out)
        self.ch_out = ch_out
        mode = pad_fn  # PAD_REPLICATE (symmetric generator, gan.py:329) or PAD_CIRCULAR (gan.py:331)
        self.conv1 = spectral_norm(Conv2d(ch_in, ch_middle, 3, pad_h=1, pad_w=1, pad_w_mode=mode, bias=False))
        self.conv2 = spectral_norm(Conv2d(ch_middle, ch_out, 3, pad_h=1, pad_w=1, pad_w_mode=mode, bias=False))
        self.norm1 = ConditionalBatchNorm2d(args, ch_middle, emb_dim)
        self.norm2 = ConditionalBatchNorm2d(args, ch_out, emb_dim)
        self.relu = nn.LeakyReLU(LRELU, inplace=True)
        self.pad = pad_fn
        if ch_in != ch_out:
            self.shortcut = spectral_norm(Conv2d(ch_in, ch_out, 1, bias=False))
        else:
            self.shortcut = _Identity()

    def forward(self, x, z, upsample=0, gb=None, out_slope=1.0):
        """x is the block input BEFORE the nearest x2 upsample that precedes the block in Generator.forward
        (gan.py:386-404) when upsample=1; the upsample is folded into conv1 and the shortcut.
        gb: {norm module: (gamma, beta)} from the generator's batched conditioning GEMM."""
        # the shortcut branch stays at the block's INPUT resolution: a 1x1 conv (or the identity) commutes with the
        # nearest x2 upsample, and the fused norm2 kernel reads the residual through that upsample
        sc = x if isinstance(self.shortcut, _Identity) else self.shortcut(x)
        g1 = gb.get(self.norm1) if gb is not None else None
        g2 = gb.get(self.norm2) if gb is not None else None
        # batch statistics of the two conv outputs come out of the conv launches where the kernel can (conv_fwd_stats)
        st1 = self.training and isinstance(self.norm1.norm, G.BatchNorm2d) and g1 is not None and g1[0].dtype == torch.float32
        st2 = self.training and isinstance(self.norm2.norm, G.BatchNorm2d) and g2 is not None and g2[0].dtype == torch.float32
        y1, p1 = self.conv1(x, upsample=upsample, want_stats=True) if st1 else (self.conv1(x, upsample=upsample), None)
        h = self.norm1(y1, z, LRELU, g1, part=p1)
        y2, p2 = self.conv2(h, want_stats=True) if st2 else (self.conv2(h), None)
        return self.norm2(y2, z, LRELU, g2, sc, out_slope, part=p2)


class Generator(nn.Module):
    """models/gan.py:314-426"""
    grad_barrier = None   # callable run in the backward pass when the trunk's gradient arrives (gan_ops.GradBarrier)

    def __init__(self, args, emb_dim, symmetric=True, mesh_head=True):
        super().__init__()
        self.relu = nn.LeakyReLU(LRELU, inplace=True)
        self.width = 8
        self.height = 8
        self.args = args
        self.symmetric = symmetric
        if symmetric:
            self.width //= 2
            self.pad = C.PAD_REPLICATE  # even-mirror padding emulated with replication (gan.py:328-329)
        else:
            self.pad = C.PAD_CIRCULAR
        if args.conditional_class and args.conditional_color:
            self.emb_class = nn.Embedding(args.n_classes[0], emb_dim // 2)
            self.emb_color = nn.Embedding(args.n_classes[1], emb_dim // 2)
            emb_dim += emb_dim
        elif args.conditional_class:
            self.emb_class = nn.Embedding(args.n_classes[0], emb_dim)
            emb_dim += emb_dim
        self.fc = nn.Linear(emb_dim, self.height * self.width * 512)
        self.blk1 = ResBlockUp(args, 512, 512, emb_dim, self.pad)
        self.blk2 = ResBlockUp(args, 512, 256, emb_dim, self.pad)
        for name, res in (("blk3a", 256), ("blk3b", 512), ("blk3c", 1024)):
            if args.texture_resolution >= res:
                setattr(self, name, ResBlockUp(args, 256, 256, emb_dim, self.pad))
        if args.conditional_text:
            self.att = SpatialAttention(256, args.text_embedding_dim)
        self.blk4 = ResBlockUp(args, 256, 128, emb_dim, self.pad)
        self.blk5 = ResBlockUp(args, 128, 128, emb_dim, self.pad)
        self.blk6 = ResBlockUp(args, 128, 64, emb_dim, self.pad)
        self.conv_final = Conv2d(64, 3, 5, pad_h=2, pad_w=2, pad_w_mode=self.pad)
        self.mesh_head = mesh_head
        if mesh_head:
            self.blk3_mesh = ResBlockUp(args, 256, 64, emb_dim, self.pad)
            self.conv_mesh = Conv2d(64, 3, 5, pad_h=2, pad_w=2, pad_w_mode=self.pad)
            self.conv_mesh.weight.data[:] = 0  # zero-initialised for smoothness (gan.py:366-368)
            self.conv_mesh.bias.data[:] = 0
        # every block but blk1 sits behind the nearest x2 upsample of gan.py:386-404, folded into its conv1 (forward below)
        for name, blk in self.named_children():
            if isinstance(blk, ResBlockUp) and name != "blk1":
                blk.conv1.m355_ups = 1

    def forward(self, z, c=None, caption=None, return_attention=False):
        a = self.args
        if a.conditional_class:
            assert c is not None
            parts = [z, self.emb_class(c[:, 0])]
            if a.conditional_color:
                parts.append(self.emb_color(c[:, 1]))
            z = torch.cat(parts, dim=1)
        self._sn_group().step(self.training)   # one batched power iteration for all 18 spectral-normed convs
        gb = self._conditioning(z)             # all fc_gamma / fc_beta in one GEMM
        x = self.fc(z).view(z.shape[0], -1, self.height, self.width)  # NCHW fp32 [B,512,8,4]
        x = G.to_nhwc_bf16(x)
        x = self.blk1(x, z, gb=gb)
        x = self.blk2(x, z, upsample=1, gb=gb)
        attention_map = None
        if a.conditional_text:
            att_out, attention_map = self.att(G.to_nchw_f32(x), *caption)
            x = x + G.to_nhwc_bf16(att_out)
        # data parallel: every gradient of blk3a .. conv_final and of the mesh head flows back through x -- when x's gradient arrives,
        # those layers' weight gradients are complete and their all-reduce can start under the backward of blk2 / blk1 / fc
        # (train.GanTrainer sets grad_barrier; None on a single GPU)
        if self.grad_barrier is not None and torch.is_grad_enabled() and x.requires_grad:
            x = G.GradBarrier.apply(x, self.grad_barrier)
        # the mesh head (blk3_mesh + conv_mesh on 32 x 16 maps: small, latency-bound layers) does not depend on the texture branch
        # below it: it runs on the second stream and fills the gaps of blk3a .. conv_final (gan_ops.Fork)
        sym = G.HT_SYMM if self.symmetric else 0
        x_mesh, fork = None, None

        def mesh_branch():
            m = self.blk3_mesh(x, z, upsample=1, gb=gb, out_slope=LRELU)
            return G.head_conv(m, self.conv_mesh, G.HT_POLES | sym, in_slope=LRELU)

        # (issued BEFORE the texture branch with or without the second stream: autograd sums the trunk's three incoming gradients
        # in reverse issue order, and that order must not depend on the stream setting -- same bits either way)
        if self.mesh_head and G.fork_ok(x, z):
            shared = [x, z] + ([t for pair in gb.values() for t in pair] if gb else [])
            with G.Fork(shared) as fork:
                x_mesh = mesh_branch()
        elif self.mesh_head:
            x_mesh = mesh_branch()
        t = x  # every later stage starts with the x2 upsample of gan.py:391 / :395-404
        for name in ("blk3a", "blk3b", "blk3c"):
            if hasattr(self, name):
                t = getattr(self, name)(t, z, upsample=1, gb=gb)
        t = self.blk4(t, z, upsample=1, gb=gb)
        t = self.blk5(t, z, upsample=1, gb=gb)
        # heads (gan.py:406-419): relu -> conv -> tanh_ | adjust_poles -> symmetrize.  The LeakyReLU is applied by the
        # block's last fused pass (out_slope) and differentiated by the head; the tail is one elementwise kernel.
        t = self.blk6(t, z, upsample=1, gb=gb, out_slope=LRELU)
        x_tex = G.head_conv(t, self.conv_final, G.HT_TANH | sym, in_slope=LRELU)
        if fork is not None:
            fork.join([x_mesh])
        if self.symmetric and attention_map is not None:
            attention_map = symmetrize_texture(attention_map)
        if self.training and self._nbt:
            torch._foreach_add_(self._nbt, 1)  # num_batches_tracked of all the batch norms in one launch
        return (x_tex, x_mesh, attention_map) if return_attention else (x_tex, x_mesh)

    # ---- network-level batching of the per-layer glue (csrc/gan_glue.hip); plain attributes, not sub-modules
    def _sn_group(self):
        g = self.__dict__.get("_sn")
        if g is None:
            g = self.__dict__["_sn"] = G.SpectralNormGroup(sn_convs(self))
        return g

    def _conditioning(self, z):
        """{ConditionalBatchNorm2d: (gamma [B,C], beta [B,C])}: the 2 x 14 Linear(emb, C) of gan.py:279-280 as ONE
        GEMM over the concatenated weights (views of its output; split's backward is a single cat)"""
        layers = self.__dict__.get("_cbn")
        if layers is None:
            layers = self.__dict__["_cbn"] = [m for m in self.modules() if isinstance(m, ConditionalBatchNorm2d)]
            self.__dict__["_nbt"] = []
            for m in layers:
                if isinstance(m.norm, G.BatchNorm2d):
                    m.norm._defer_count = True
                    if not m.norm.sync:   # SynchronizedBatchNorm2d never counts its batches (batchnorm.py:66-98)
                        self.__dict__["_nbt"].append(m.norm.num_batches_tracked)
        if not layers:
            return None
        ws = [w for m in layers for w in (m.fc_gamma.weight, m.fc_beta.weight)]
        bs = [b for m in layers for b in (m.fc_gamma.bias, m.fc_beta.bias)]
        out = F.linear(z, torch.cat(ws), torch.cat(bs))
        parts = out.split([w.shape[0] for w in ws], dim=1)
        return {m: (parts[2 * i], parts[2 * i + 1]) for i, m in enumerate(layers)}

    def state_dict(self, *args, **kwargs):
        g = self.__dict__.get("_sn")
        if g is not None:
            g.cancel_prefetch()   # (a spectral-norm step computed ahead of its forward: a checkpoint holds the last FORWARD's u / v)
        return super().state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .half() replace parameter and buffer tensors: the batching caches (device tables of the
        # spectral-norm group, the layer list and the num_batches_tracked references) must be rebuilt from the new ones
        for k in ("_sn", "_cbn", "_nbt"):
            self.__dict__.pop(k, None)
        return super()._apply(fn, *args, **kwargs)

    def __deepcopy__(self, memo):
        # the batching caches hold device pointers / module references of THIS instance: rebuild them in the copy
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k in ("_sn", "_cbn", "_nbt"):
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        return new


# ------------------------------------------------------------------------------------------------ discriminators
def _d_norm(args, ch):
    if args.norm_d == 'instance':
        return nn.InstanceNorm2d(ch, affine=True), False
    if args.norm_d == 'none':
        return None, True
    raise ValueError(f"norm_d={args.norm_d!r}")


class _DiscBase(nn.Module):
    _sn_external = False  # True when a MultiScaleDiscriminator steps the spectral norm of all its members at once

    def _sn_step(self):
        if self._sn_external:
            return
        g = self.__dict__.get("_sn")
        if g is None:
            g = self.__dict__["_sn"] = G.SpectralNormGroup(sn_convs(self))
        g.step(self.training)

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k != "_sn":
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def _pos(self, x):
        """cached positional embedding [1,4,H,W] for an NCHW tensor (gan.py:87-90, 204-207)"""
        if self.pos_emb is None:
            self.pos_emb = torch.FloatTensor(positional_encoding(x.shape[2], x.shape[3])).unsqueeze(0)
        # (the reference re-uploads the CPU tensor on every forward; the device copy is kept here: one H2D copy per
        # device instead of one blocking copy per call, and the forward becomes hipGraph-capturable)
        dev = self.__dict__.get("_pos_dev")
        if dev is None or dev.device != x.device:
            dev = self.__dict__["_pos_dev"] = self.pos_emb.to(x.device)
        return dev.expand(x.shape[0], -1, -1, -1)

    def _pos_hw(self, h, w, device):
        """[4,h,w] positional encoding on `device` (same lazy cache as _pos: fixed by the first call's shape)"""
        if self.pos_emb is None:
            self.pos_emb = torch.FloatTensor(positional_encoding(h, w)).unsqueeze(0)
        dev = self.__dict__.get("_pos_dev")
        if dev is None or dev.device != device:
            dev = self.__dict__["_pos_dev"] = self.pos_emb.to(device)
        return dev[0]

    @staticmethod
    def _out_shape(conv, s):
        """NHWC shape of conv's output for an input of NHWC shape s"""
        st, ph, pw, _ = conv.m355
        kh, kw = conv.kernel_size
        return (s[0], (s[1] + 2 * ph - kh) // st + 1, (s[2] + 2 * pw - kw) // st + 1, conv.out_channels)

    @staticmethod
    def _masks_its_input(conv, s):
        """can conv's dgrad, at input shape s, apply the LeakyReLU backward of the layer below in its epilogue?  (Not for every
        size: the direct-form dgrad it rides on ends at 2 GiB tensors -- a batch of 256 at 256^2 is past that, and the pair then
        runs unfused: the producer keeps its activation and differentiates it itself.)"""
        st, ph, pw, mode = conv.m355
        if mode == C.PAD_REPLICATE:
            return False
        kh, kw = conv.kernel_size
        return C.dgrad_mask_ok(C.make_desc(s[0], s[1], s[2], s[3], conv.out_channels, kh, kw, st, ph, pw, mode, 0))

    def _act(self, conv, norm, x, in_act=False, sole_consumer_masks=False):
        """conv -> [InstanceNorm] -> LeakyReLU on NHWC bf16.  With the activation in the conv epilogue (no norm):
        in_act = x is the previous layer's fused conv+LeakyReLU output and this conv is its only consumer -> this
        conv's dgrad applies that activation's backward; sole_consumer_masks = the same arrangement one layer up,
        i.e. this layer's incoming gradient is already masked."""
        if norm is None:
            fuse = x.is_cuda and conv.m355[3] != C.PAD_REPLICATE
            return conv(x, slope=LRELU, in_slope=LRELU if (in_act and fuse) else 1.0,
                        premasked=sole_consumer_masks and fuse)
        y = G.to_nchw_f32(conv(x))
        return G.to_nhwc_bf16(F.leaky_relu(norm(y), LRELU))

    def _tail(self, conv_feat, norm_feat, conv_out, h, in_act, c, caption):
        """last feature conv -> LeakyReLU -> (logit conv, projection term).  The activation has two consumers, so its
        backward cannot ride on ONE consumer's dgrad -- but masking is linear: when BOTH consumers mask their own branch of the
        gradient (the logit conv's dgrad via mask_x, the projection kernel via mask_slope), the feature conv is `premasked`
        and the separate activation-backward pass over the summed gradient (134 MB at batch 128) disappears."""
        a = self.args
        pm = False
        if (norm_feat is None and h.is_cuda and conv_feat.m355[3] != C.PAD_REPLICATE and not a.conditional_text
                and torch.is_grad_enabled()):
            cf = conv_feat.out_channels
            st, ph, pw, _ = conv_feat.m355
            kh, kw = conv_feat.kernel_size
            hf, wf = (h.shape[1] + 2 * ph - kh) // st + 1, (h.shape[2] + 2 * pw - kw) // st + 1
            so, pho, pwo, mo = conv_out.m355
            d = C.make_desc(h.shape[0], hf, wf, cf, conv_out.out_channels, conv_out.kernel_size[0], conv_out.kernel_size[1], so,
                            pho, pwo, mo, 0)
            proj_ok = (not a.conditional_class) or (self.projector.weight.dtype == torch.float32 and cf % 8 == 0 and cf <= 2048
                                                     and 256 % (cf // 8) == 0)
            pm = bool(proj_ok and C.dgrad_mask_ok(d))
        h = self._act(conv_feat, norm_feat, h, in_act, pm)
        # premasked + class projection: ONE kernel produces h's gradient from both consumers (gan_ops.TailPair).  The projection term
        # is evaluated BEFORE the logit conv -- autograd runs the younger node's backward first, and the pair needs the conv's first
        pair = None
        if pm and a.conditional_class:
            so, pho, pwo, mo = conv_out.m355
            if G.tail_pair_ok(h, (conv_out.kernel_size[0], conv_out.kernel_size[1], so, pho, pwo), mo, conv_out.out_channels):
                pair = G.TailPair()
        if pair is not None:
            p = self._projection_term(h, c, LRELU, pair)
            y = conv_out(h, out_f32_nchw=True, in_slope=LRELU, dx_pair=pair)
            return y + p.unsqueeze(1)
        y = conv_out(h, out_f32_nchw=True, in_slope=LRELU if pm else 1.0)
        return self._project(y, h, c, caption, LRELU if pm else 1.0)

    def _projection_term(self, feat, c, in_slope=1.0, pair=None):
        a = self.args
        c_emb = self.projector(c[:, 0])
        if a.conditional_color:
            c_emb = c_emb + self.projector_col1(c[:, 1])
        return G.class_projection(feat, c_emb, in_slope, pair)

    def _project(self, y, feat, c, caption, in_slope=1.0):
        """projection discriminator (gan.py:104-116, 216-228): y += sum_c feat * emb"""
        a = self.args
        if a.conditional_class:
            y = y + self._projection_term(feat, c, in_slope).unsqueeze(1)
        elif a.conditional_text:
            att_out, _ = self.att(G.to_nchw_f32(feat), *caption)
            y = y + torch.sum(G.to_nchw_f32(feat) * att_out, dim=1, keepdim=True)
        return y


class MeshDiscriminator(_DiscBase):
    """models/gan.py:23-121"""

    def __init__(self, args, nc, circular=True, positional_embeddings=True):
        super().__init__()
        n2, bias = _d_norm(args, 128)
        n3, _ = _d_norm(args, 256)
        self.args = args
        if args.conditional_text:
            self.att = SpatialAttention(256, args.text_embedding_dim)
        self.circular = circular
        self.positional_embeddings = positional_embeddings
        mode = C.PAD_CIRCULAR if circular else C.PAD_ZERO
        if positional_embeddings:
            self.pos_emb = None
            nc += 4
        self.conv1 = spectral_norm(Conv2d(nc, 64, 5, pad_h=2, pad_w=2, pad_w_mode=mode))
        self.conv2 = spectral_norm(Conv2d(64, 128, 4, stride=2, pad_h=1, pad_w=1, pad_w_mode=mode, bias=bias))
        if n2 is not None:
            self.bn2 = n2
        self.conv3 = spectral_norm(Conv2d(128, 256, 4, stride=2, pad_h=1, pad_w=1, pad_w_mode=mode, bias=bias))
        if n3 is not None:
            self.bn3 = n3
        self.conv4 = spectral_norm(Conv2d(256, 1, 5, pad_h=2, pad_w=2, pad_w_mode=mode))
        self.relu = nn.LeakyReLU(LRELU, inplace=True)
        if args.conditional_class:
            self.projector = nn.Embedding(args.n_classes[0], 256)
            if args.conditional_color:
                self.projector_col1 = nn.Embedding(args.n_classes[1], 256)

    def input_spec(self, texture, mesh_map):
        """(pool factor, takes the mesh map, positional planes, packed channels, mask pool) for gan_ops.disc_inputs"""
        f = texture.shape[2] // mesh_map.shape[2]
        pos = self._pos_hw(texture.shape[2] // f, texture.shape[3] // f, texture.device) if self.positional_embeddings else None
        nch = texture.shape[1] + mesh_map.shape[1] + (4 if self.positional_embeddings else 0)
        return (f, True, pos, (nch + 7) // 8 * 8, 4 if self.args.mask_output else 0)

    def forward(self, texture, mesh_map, c=None, caption=None):
        self._sn_step()
        spec = self.input_spec(texture, mesh_map)
        if G.disc_inputs_ok(texture, mesh_map, [spec]):
            (h,), (mask,) = G.disc_inputs(texture, mesh_map, [spec])
            return self.trunk(h, mask, c, caption)
        x = F.avg_pool2d(texture, texture.shape[2] // mesh_map.shape[2])
        parts = [x, mesh_map]
        if self.positional_embeddings:
            parts.append(self._pos(x))
        x = torch.cat(parts, dim=1)
        mask = None
        if self.args.mask_output:
            with torch.no_grad():
                mask = F.avg_pool2d(x[:, 3:4], 4)
        return self.trunk(G.to_nhwc_bf16(x, pad_to=8), mask, c, caption)

    def trunk(self, h, mask, c=None, caption=None):
        """conv1 .. conv4 + projection on the packed NHWC bf16 input (gan.py:100-121)"""
        # conv1 -> conv2 -> conv3 are single-consumer chains when norm_d == 'none': each dgrad carries the LeakyReLU
        # backward of the layer below (conv3's output also feeds the projection term, so it keeps its own)
        n2, n3 = getattr(self, "bn2", None), getattr(self, "bn3", None)
        s2 = self._out_shape(self.conv1, h.shape)
        p12 = n2 is None and h.is_cuda and self._masks_its_input(self.conv2, s2)
        p23 = n2 is None and n3 is None and h.is_cuda and self._masks_its_input(self.conv3, self._out_shape(self.conv2, s2))
        h = self._act(self.conv1, None, h, False, p12)
        h = self._act(self.conv2, n2, h, p12, p23)
        return self._tail(self.conv3, n3, self.conv4, h, p23, c, caption), mask


class TextureDiscriminator(_DiscBase):
    """models/gan.py:123-233"""

    def __init__(self, args, nc, downsample=1, circular=True, positional_embeddings=True):
        super().__init__()
        n2, bias = _d_norm(args, 128)
        n3, _ = _d_norm(args, 256)
        n4, _ = _d_norm(args, 512)
        self.args = args
        if args.conditional_text:
            self.att = SpatialAttention(512, args.text_embedding_dim)
        self.circular = circular
        self.positional_embeddings = positional_embeddings
        mode = C.PAD_CIRCULAR if circular else C.PAD_ZERO
        nc_image = nc
        if positional_embeddings:
            self.pos_emb = None
            nc += 4
        self.stride_first = (downsample == 1 and args.texture_resolution >= 512) or args.texture_resolution >= 1024 \
            or args.conditional_text
        if self.stride_first:
            self.conv1 = spectral_norm(Conv2d(nc, 64, 4, stride=2, pad_h=1, pad_w=1, pad_w_mode=mode))
        else:
            self.conv1 = spectral_norm(Conv2d(nc, 64, 5, pad_h=2, pad_w=2, pad_w_mode=mode))
        if positional_embeddings and nc_image <= 4:
            # conv1's input = the image channels + four batch-constant positional planes (gan.py:204-207): nothing reads the planes'
            # gradient, the backward only produces the image channels' (the G step's dL/d texture)
            self.conv1.m355_dx_lead = nc_image
        self.conv2 = spectral_norm(Conv2d(64, 128, 4, stride=2, pad_h=1, pad_w=1, pad_w_mode=mode, bias=bias))
        if n2 is not None:
            self.bn2 = n2
        self.conv3 = spectral_norm(Conv2d(128, 256, 4, stride=2, pad_h=1, pad_w=1, pad_w_mode=mode, bias=bias))
        if n3 is not None:
            self.bn3 = n3
        self.conv4 = spectral_norm(Conv2d(256, 512, 4, stride=2, pad_h=1, pad_w=1, pad_w_mode=mode, bias=bias))
        if n4 is not None:
            self.bn4 = n4
        self.conv5 = spectral_norm(Conv2d(512, 1, 5, pad_h=2, pad_w=2, pad_w_mode=mode))
        self.relu = nn.LeakyReLU(LRELU, inplace=True)
        self.downsample = downsample
        if args.conditional_class:
            self.projector = nn.Embedding(args.n_classes[0], 512)
            if args.conditional_color:
                self.projector_col1 = nn.Embedding(args.n_classes[1], 512)

    def input_spec(self, x):
        """(pool factor, takes the mesh map, positional planes, packed channels, mask pool) for gan_ops.disc_inputs"""
        f = self.downsample
        pos = self._pos_hw(x.shape[2] // f, x.shape[3] // f, x.device) if self.positional_embeddings else None
        return (f, False, pos, 8, (16 if self.stride_first else 8) if self.args.mask_output else 0)

    def forward(self, x, c=None, caption=None):
        self._sn_step()
        spec = self.input_spec(x)
        if G.disc_inputs_ok(x, None, [spec]):
            (h,), (mask,) = G.disc_inputs(x, None, [spec])
            return self.trunk(h, mask, c, caption)
        if self.downsample > 1:
            x = F.avg_pool2d(x, self.downsample)
        mask = None
        if self.args.mask_output:
            with torch.no_grad():
                mask = F.avg_pool2d(x[:, 3:4], 16 if self.stride_first else 8)
        # cat((x, positional encoding)) -> NHWC bf16 (8 channels) in one pass
        h = G.pack_nhwc8(x, self._pos(x)[0] if self.positional_embeddings else None)
        return self.trunk(h, mask, c, caption)

    def trunk(self, h, mask, c=None, caption=None):
        """conv1 .. conv5 + projection on the packed NHWC bf16 input (gan.py:212-233)"""
        n2, n3, n4 = getattr(self, "bn2", None), getattr(self, "bn3", None), getattr(self, "bn4", None)
        s2 = self._out_shape(self.conv1, h.shape)
        s3 = self._out_shape(self.conv2, s2)
        p12 = n2 is None and h.is_cuda and self._masks_its_input(self.conv2, s2)
        p23 = n2 is None and n3 is None and h.is_cuda and self._masks_its_input(self.conv3, s3)
        p34 = n3 is None and n4 is None and h.is_cuda and self._masks_its_input(self.conv4, self._out_shape(self.conv3, s3))
        h = self._act(self.conv1, None, h, False, p12)
        h = self._act(self.conv2, n2, h, p12, p23)
        h = self._act(self.conv3, n3, h, p23, p34)
        return self._tail(self.conv4, n4, self.conv5, h, p34, c, caption), mask


class MultiScaleDiscriminator(nn.Module):
    """models/gan.py:235-260"""

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k != "_sn":
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def __init__(self, args, nc):
        super().__init__()
        self.args = args
        self.d1 = TextureDiscriminator(args, nc, 1)
        if not args.texture_only:
            self.d2 = MeshDiscriminator(args, nc + 3)
        else:
            self.d2 = TextureDiscriminator(args, nc, 2)
        if args.num_discriminators == 3:
            self.d3 = TextureDiscriminator(args, nc, 4)
        elif args.num_discriminators != 2:
            raise ValueError(f"num_discriminators={args.num_discriminators}")

    def _sn_group(self):
        g = self.__dict__.get("_sn")
        if g is None:
            g = self.__dict__["_sn"] = G.SpectralNormGroup(sn_convs(self))
            for m in self.children():
                if isinstance(m, _DiscBase):
                    m.__dict__["_sn_external"] = True
        return g

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop("_sn", None)   # (.to() / .cuda() replace the parameter and buffer tensors the group's tables point at)
        return super()._apply(fn, *args, **kwargs)

    def state_dict(self, *args, **kwargs):
        g = self.__dict__.get("_sn")
        if g is not None:
            g.cancel_prefetch()   # (see Generator.state_dict)
        return super().state_dict(*args, **kwargs)

    def forward(self, x, mesh_map=None, c=None, caption=None):
        self._sn_group().step(self.training)
        members = [self.d1, self.d2] + ([self.d3] if self.args.num_discriminators == 3 else [])
        # every member's input assembly (pooling, mesh / positional planes, masks, NHWC bf16 packing) from ONE read
        # interface: one launch per member, one backward launch for all of them (gan_ops.DiscInputsFn)
        lazy = isinstance(x, G.MaskedInput)   # (the trainer's not-yet-concatenated input: assembled by the loaders below)
        if (lazy or (torch.is_tensor(x) and x.is_cuda)) and (self.args.texture_only or mesh_map is not None):
            specs = [m.input_spec(x, mesh_map) if isinstance(m, MeshDiscriminator) else m.input_spec(x) for m in members]
            extra = None if self.args.texture_only else mesh_map
            if G.disc_inputs_ok(x, extra, specs):
                hs, masks = G.disc_inputs(x, extra, specs)
                # the mesh discriminator (32 x 32 .. 8 x 8 maps: latency-bound layers that leave most of the chip idle) runs on
                # the second stream, under the texture discriminator's big layers (gan_ops.Fork)
                side = [k for k, m in enumerate(members) if isinstance(m, MeshDiscriminator)] if len(members) > 1 else []
                outs = [None] * len(members)
                fork = None
                if side and caption is None and G.fork_ok(hs[side[0]], c):
                    k = side[0]
                    with G.Fork([hs[k], masks[k], c]) as fork:
                        outs[k] = members[k].trunk(hs[k], masks[k], c, caption)
                elif side:   # (same issue order without the second stream)
                    outs[side[0]] = members[side[0]].trunk(hs[side[0]], masks[side[0]], c, caption)
                for k, (m, h, mk) in enumerate(zip(members, hs, masks)):
                    if outs[k] is None:
                        outs[k] = m.trunk(h, mk, c, caption)
                if fork is not None:
                    fork.join([outs[side[0]][0], outs[side[0]][1]])
                return [o[0] for o in outs], [o[1] for o in outs]
        if lazy:
            x = x.materialize()
        d1, m1 = self.d1(x, c, caption)
        if self.args.texture_only:
            d2, m2 = self.d2(x, c, caption)
        else:
            d2, m2 = self.d2(x, mesh_map, c, caption)
        if self.args.num_discriminators == 3:
            d3, m3 = self.d3(x, c, caption)
            return [d1, d2, d3], [m1, m2, m3]
        return [d1, d2], [m1, m2]


class SpatialAttention(nn.Module):
    """models/gan.py:433-481 (text conditioning; tiny bmm's on [B, <=512, <=18] -- plain torch, NCHW fp32)"""

    def __init__(self, input_dim, context_dim):
        super().__init__()
        self.conv_context = nn.Conv2d(context_dim, input_dim, 1, stride=1, padding=0, bias=False)
        self.sm = nn.Softmax(dim=1)

    def forward(self, input, context, mask):
        ih, iw = input.size(2), input.size(3)
        B, L = context.size(0), context.size(2)
        q = input.view(B, -1, ih * iw).transpose(1, 2)                     # B x queryL x idf
        src = self.conv_context(context.unsqueeze(3)).squeeze(3)          # B x idf x sourceL
        attn = torch.bmm(q, src).view(B * ih * iw, L)
        if mask is not None:
            attn = attn + mask.unsqueeze(1).expand(-1, ih * iw, -1).reshape(B * ih * iw, -1).float() * -10000
        attn = self.sm(attn).view(B, ih * iw, L).transpose(1, 2)          # B x sourceL x queryL
        out = torch.bmm(src, attn).view(B, -1, ih, iw)
        return out, attn.reshape(B, -1, ih, iw)


# ------------------------------------------------------------------------------------------------ loss
class GANLoss(nn.Module):
    """utils/losses.py:21-120 (hinge / ls / original / w, masked per-sample mean, per-discriminator weights).
    The hinge loss over a list of CUDA logits (the training configuration) is one fused kernel each way
    (csrc/gan_io.hip k_hinge_*); the other modes -- a few KB of logits per call -- use torch ops."""

    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0, tensor=torch.FloatTensor, opt=None):
        super().__init__()
        if gan_mode not in ('ls', 'original', 'w', 'hinge'):
            raise ValueError('Unexpected gan_mode {}'.format(gan_mode))
        self.real_label, self.fake_label = target_real_label, target_fake_label
        self.Tensor, self.gan_mode, self.opt = tensor, gan_mode, opt

    @staticmethod
    def mean(x, mask=None, weight=None):
        weight = 1 if weight is None else weight
        if mask is None:
            return torch.mean(x) * weight
        assert x.shape == mask.shape, (x.shape, mask.shape)
        per = torch.sum(x * mask, dim=[1, 2, 3]) / torch.sum(mask, dim=[1, 2, 3])  # NaN on an empty mask (D14)
        return torch.mean(per) * weight

    def loss(self, input, target_is_real, for_discriminator=True, mask=None, weight=None):
        if self.gan_mode == 'original':
            t = torch.full_like(input, self.real_label if target_is_real else self.fake_label)
            return F.binary_cross_entropy_with_logits(input, t)
        if self.gan_mode == 'ls':
            t = torch.full_like(input, self.real_label if target_is_real else self.fake_label)
            return F.mse_loss(input, t)
        if self.gan_mode == 'hinge':
            if for_discriminator:
                v = torch.clamp_max((input if target_is_real else -input) - 1, 0)
                return -self.mean(v, mask, weight)
            assert target_is_real, "The generator's hinge loss must be aiming for real"
            return -self.mean(input, mask, weight)
        return -input.mean() if target_is_real else input.mean()

    def d_losses(self, input, mask=None, weight=None):
        """(loss vs target False on the first half of the batch, loss vs target True on the second half) of discriminator
        outputs computed on a [fake; real] batch: divide_pred (main.py:414-422) + the two criterion calls of
        main.py:518-519 in one pass"""
        if self.gan_mode == 'hinge' and G.hinge_ok(input, mask) and input[0].shape[0] % 2 == 0:
            return G.hinge_losses(input, mask, weight, 1, input[0].shape[0] // 2)
        half = lambda ts: (None, None) if ts is None else ([None if t is None else t[:t.shape[0] // 2] for t in ts],
                                                           [None if t is None else t[t.shape[0] // 2:] for t in ts])
        (f, r), (mf, mr) = half(input), half(mask)
        return self(f, False, True, mf, weight), self(r, True, True, mr, weight)

    def __call__(self, input, target_is_real, for_discriminator=True, mask=None, weight=None):
        if not isinstance(input, list):
            return self.loss(input, target_is_real, for_discriminator, mask)
        if mask is not None:
            assert isinstance(mask, list) and len(input) == len(mask)
        if self.gan_mode == 'hinge' and G.hinge_ok(input, mask):
            B = input[0].shape[0]
            if for_discriminator:
                lf, lr = G.hinge_losses(input, mask, weight, 1, 0 if target_is_real else B)
                return lr if target_is_real else lf
            assert target_is_real, "The generator's hinge loss must be aiming for real"
            return G.hinge_losses(input, mask, weight, 0, B)[0]
        total = 0
        for i, pred in enumerate(input):
            if isinstance(pred, list):
                pred = pred[-1]
            t = self.loss(pred, target_is_real, for_discriminator, None if mask is None else mask[i],
                          None if weight is None else weight[i])
            bs = 1 if t.dim() == 0 else t.size(0)
            total = total + torch.mean(t.view(bs, -1), dim=1)
        return total / (len(input) if weight is None else sum(weight))
