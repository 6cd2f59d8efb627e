"""TEST INFRASTRUCTURE ONLY -- fp32 torch-CPU restatement of the reference's GAN cycle as pure functions of a state_dict.

Only tests/ and bench.py's `cpu_baseline_gan` leg may import this module; the product path
(2dimageto3dmodel_amd/) never does.  It exists because /root/reference does not travel to the GPU box: the
committed goldens pin it (tests/test_oracle_golden.py::test_gan_cpu_*: same seeds -> same weights -> the
reference's outputs, losses and gradient tensors to ~1e-5), and then it serves as the checker at shapes the goldens
do not hold and as the CPU leg timed beside the HIP path.

Each function cites the reference lines (under /root/reference/code) it restates.  The arithmetic is the reference's
(F.conv2d / F.batch_norm / F.instance_norm / torch.nn.utils.spectral_norm's power iteration); only the packaging
differs: no nn.Modules, weights are looked up by their state_dict key.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

SLOPE = 0.2  # nn.LeakyReLU(0.2), models/gan.py:67,179,303,318


class Weights:
    """A state_dict split into differentiable leaves (parameters) and buffers; `sub(prefix)` scopes the keys."""

    def __init__(self, state_dict, prefix="", store=None, grad=True):
        if store is None:
            store = {}
            for k, v in state_dict.items():
                t = v.detach().to("cpu", copy=True)
                if grad and torch.is_floating_point(t) and not _is_buffer(k):
                    t.requires_grad_(True)
                store[k] = t
        self.store, self.prefix = store, prefix

    def sub(self, name):
        return Weights(None, self.prefix + name + ".", self.store)

    def __getitem__(self, k):
        return self.store[self.prefix + k]

    def has(self, k):
        return (self.prefix + k) in self.store

    def grads(self):
        return {k: v.grad for k, v in self.store.items() if v.requires_grad and v.grad is not None}

    def zero_grad(self):
        for v in self.store.values():
            v.grad = None


def _is_buffer(k):
    leaf = k.rsplit(".", 1)[-1]
    return leaf in ("weight_u", "weight_v", "running_mean", "running_var", "num_batches_tracked")


# ------------------------------------------------------------------------------------------------ layer pieces
def sn_weight(w, name, training):
    """torch.nn.utils.spectral_norm's forward pre-hook (one power iteration in training mode, u / v updated in place,
    sigma = u . (W v) differentiable through W) -- what models/gan.py:57-65,163-177,294-302 wrap every conv in"""
    if not w.has(name + "_orig"):
        return w[name]
    W, u, v = w[name + "_orig"], w[name + "_u"], w[name + "_v"]
    Wm = W.reshape(W.shape[0], -1)
    if training:
        with torch.no_grad():
            v.copy_(F.normalize(torch.mv(Wm.t(), u), dim=0, eps=1e-12))
            u.copy_(F.normalize(torch.mv(Wm, v), dim=0, eps=1e-12))
    sigma = torch.dot(u.clone(), torch.mv(Wm, v.clone()))
    return W / sigma


def pad_w(x, amount, mode):
    """W padding in front of a convolution: 'replicate' (gan.py:328-329), 'circular' (rendering/utils.py:29-33), or none"""
    if amount == 0 or mode == "zero":
        return x
    if mode == "replicate":
        return F.pad(x, (amount, amount, 0, 0), mode="replicate")
    return torch.cat((x[:, :, :, -amount:], x, x[:, :, :, :amount]), dim=3)


def conv(w, name, x, k, stride, mode, training):
    """nn.Conv2d(..., k, padding=(k//2 or 1, 0), stride) behind the W pad of its caller"""
    ph = 1 if k == 4 else k // 2
    c = w.sub(name)
    bias = c["bias"] if c.has("bias") else None
    return F.conv2d(pad_w(x, ph, mode), sn_weight(c, "weight", training), bias, stride=stride, padding=(ph, 0))


def cbn(w, name, x, z, norm, training):
    """ConditionalBatchNorm2d.forward, gan.py:282-286"""
    c = w.sub(name)
    if norm in ("batch", "syncbatch"):   # sync_batchnorm/batchnorm.py:70-73: F.batch_norm on one device
        n = c.sub("norm")
        x = F.batch_norm(x, n["running_mean"], n["running_var"], None, None, training, 0.1, 1e-5)
        if training and norm == "batch":   # nn.BatchNorm2d counts; SynchronizedBatchNorm2d.forward does not (batchnorm.py:66-98)
            n["num_batches_tracked"].add_(1)
    elif norm == "instance":
        x = F.instance_norm(x)
    gamma = F.linear(z, c["fc_gamma.weight"], c["fc_gamma.bias"])[:, :, None, None]
    beta = F.linear(z, c["fc_beta.weight"], c["fc_beta.bias"])[:, :, None, None]
    return x * (1 + gamma) + beta


def res_block_up(w, name, x, z, norm, mode, training):
    """ResBlockUp.forward, gan.py:306-312"""
    b = w.sub(name)
    sc = conv(b, "shortcut", x, 1, 1, "zero", training) if b.has("shortcut.weight_orig") else x
    h = F.leaky_relu(cbn(b, "norm1", conv(b, "conv1", x, 3, 1, mode, training), z, norm, training), SLOPE)
    h = F.leaky_relu(cbn(b, "norm2", conv(b, "conv2", h, 3, 1, mode, training), z, norm, training), SLOPE)
    return h + sc


def spatial_attention(w, name, x, context, mask):
    """SpatialAttention.forward, gan.py:446-481"""
    ih, iw = x.shape[2:]
    B, L = context.shape[0], context.shape[2]
    src = F.conv2d(context.unsqueeze(3), w[name + ".conv_context.weight"]).squeeze(3)
    attn = torch.bmm(x.reshape(B, -1, ih * iw).transpose(1, 2), src).view(B * ih * iw, L)
    if mask is not None:
        attn = attn + mask.unsqueeze(1).expand(-1, ih * iw, -1).reshape(B * ih * iw, -1).float() * -10000
    attn = torch.softmax(attn, dim=1).view(B, ih * iw, L).transpose(1, 2)
    return torch.bmm(src, attn).view(B, -1, ih, iw), attn.reshape(B, -1, ih, iw)


def symmetrize(x):
    """rendering/utils.py:15-18"""
    xf = torch.flip(x, (3,))
    h = xf.shape[3] // 2
    return torch.cat((xf[..., h:], x, xf[..., :h]), dim=-1)


def poles(t):
    """rendering/utils.py:21-26"""
    top = t[:, :, :1].mean(3, keepdim=True).expand(-1, -1, -1, t.shape[3])
    bot = t[:, :, -1:].mean(3, keepdim=True).expand(-1, -1, -1, t.shape[3])
    return torch.cat((top, t[:, :, 1:-1], bot), dim=2)


def positional(Ny, Nx):
    """gan.py:9-20"""
    n = Ny
    ty, tx = np.linspace(0, np.pi, n, endpoint=False), np.linspace(-np.pi, np.pi, n, endpoint=False)
    Y, X = np.meshgrid(tx, ty)
    enc = np.stack((np.cos(X), np.sin(X), np.cos(Y), np.sin(Y)))
    if Nx == Ny // 2:
        enc = enc[:, :, n // 4:-(n // 4)]
    return torch.FloatTensor(enc).unsqueeze(0)


# ------------------------------------------------------------------------------------------------ networks
def generator(w, args, z, c=None, caption=None, symmetric=True, training=True):
    """Generator.forward, gan.py:372-426 -> (x_tex, x_mesh or None)"""
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
    mode = "replicate" if symmetric else "circular"
    norm = args.norm_g
    if args.conditional_class:
        parts = [z, F.embedding(c[:, 0], w["emb_class.weight"])]
        if args.conditional_color:
            parts.append(F.embedding(c[:, 1], w["emb_color.weight"]))
        z = torch.cat(parts, dim=1)
    x = F.linear(z, w["fc.weight"], w["fc.bias"]).view(z.shape[0], -1, 8, 4 if symmetric else 8)
    x = up(res_block_up(w, "blk1", x, z, norm, mode, training))
    x = res_block_up(w, "blk2", x, z, norm, mode, training)
    if args.conditional_text:
        att, _ = spatial_attention(w, "att", x, *caption)
        x = x + att
    x = up(x)
    t = x
    for name, res in (("blk3a", 256), ("blk3b", 512), ("blk3c", 1024)):
        if args.texture_resolution >= res:
            t = up(res_block_up(w, name, t, z, norm, mode, training))
    t = up(res_block_up(w, "blk4", t, z, norm, mode, training))
    t = up(res_block_up(w, "blk5", t, z, norm, mode, training))
    t = F.leaky_relu(res_block_up(w, "blk6", t, z, norm, mode, training), SLOPE)
    x_tex = torch.tanh(F.conv2d(pad_w(t, 2, mode), w["conv_final.weight"], w["conv_final.bias"], padding=(2, 0)))
    x_mesh = None
    if w.has("conv_mesh.weight"):
        m = F.leaky_relu(res_block_up(w, "blk3_mesh", x, z, norm, mode, training), SLOPE)
        x_mesh = poles(F.conv2d(pad_w(m, 2, mode), w["conv_mesh.weight"], w["conv_mesh.bias"], padding=(2, 0)))
    if symmetric:
        x_tex = symmetrize(x_tex)
        if x_mesh is not None:
            x_mesh = symmetrize(x_mesh)
    return x_tex, x_mesh


def _d_norm(w, name, x, args):
    if args.norm_d == "instance":   # nn.InstanceNorm2d(nc, affine=True), gan.py:29-31,129-131
        return F.instance_norm(x, weight=w[name + ".weight"], bias=w[name + ".bias"])
    return x


def _project(w, args, y, feat, c, caption):
    """projection discriminator, gan.py:104-116,216-228"""
    if args.conditional_class:
        e = F.embedding(c[:, 0], w["projector.weight"])
        if args.conditional_color:
            e = e + F.embedding(c[:, 1], w["projector_col1.weight"])
        y = y + torch.sum(feat * e[:, :, None, None], dim=1, keepdim=True)
    elif args.conditional_text:
        att, _ = spatial_attention(w, "att", feat, *caption)
        y = y + torch.sum(feat * att, dim=1, keepdim=True)
    return y


def texture_discriminator(w, args, x, c=None, caption=None, downsample=1, training=True):
    """TextureDiscriminator.forward, gan.py:192-233"""
    if downsample > 1:
        x = F.avg_pool2d(x, downsample)
    stride_first = (downsample == 1 and args.texture_resolution >= 512) or args.texture_resolution >= 1024 \
        or args.conditional_text                                                  # gan.py:158-160
    mask = None
    if args.mask_output:
        with torch.no_grad():
            mask = F.avg_pool2d(x[:, 3:4], 16 if stride_first else 8)
    x = torch.cat((x, positional(x.shape[2], x.shape[3]).expand(x.shape[0], -1, -1, -1)), dim=1)
    lr = lambda t: F.leaky_relu(t, SLOPE)
    x = lr(conv(w, "conv1", x, 4, 2, "circular", training) if stride_first else conv(w, "conv1", x, 5, 1, "circular", training))
    x = lr(_d_norm(w, "bn2", conv(w, "conv2", x, 4, 2, "circular", training), args))
    x = lr(_d_norm(w, "bn3", conv(w, "conv3", x, 4, 2, "circular", training), args))
    x = lr(_d_norm(w, "bn4", conv(w, "conv4", x, 4, 2, "circular", training), args))
    y = conv(w, "conv5", x, 5, 1, "circular", training)
    return _project(w, args, y, x, c, caption), mask


def mesh_discriminator(w, args, texture, mesh_map, c=None, caption=None, training=True):
    """MeshDiscriminator.forward, gan.py:79-121"""
    x = F.avg_pool2d(texture, texture.shape[2] // mesh_map.shape[2])
    x = torch.cat((x, mesh_map, positional(x.shape[2], x.shape[3]).expand(x.shape[0], -1, -1, -1)), dim=1)
    mask = None
    if args.mask_output:
        with torch.no_grad():
            mask = F.avg_pool2d(x[:, 3:4], 4)
    lr = lambda t: F.leaky_relu(t, SLOPE)
    x = lr(conv(w, "conv1", x, 5, 1, "circular", training))
    x = lr(_d_norm(w, "bn2", conv(w, "conv2", x, 4, 2, "circular", training), args))
    x = lr(_d_norm(w, "bn3", conv(w, "conv3", x, 4, 2, "circular", training), args))
    y = conv(w, "conv4", x, 5, 1, "circular", training)
    return _project(w, args, y, x, c, caption), mask


def discriminator(w, args, x, mesh_map, c=None, caption=None, training=True):
    """MultiScaleDiscriminator.forward, gan.py:250-260 (d2 = MeshDiscriminator; texture_only cannot run in the reference)"""
    outs = [texture_discriminator(w.sub("d1"), args, x, c, caption, 1, training),
            mesh_discriminator(w.sub("d2"), args, x, mesh_map, c, caption, training)]
    if args.num_discriminators == 3:
        outs.append(texture_discriminator(w.sub("d3"), args, x, c, caption, 4, training))
    return [o[0] for o in outs], [o[1] for o in outs]


# ------------------------------------------------------------------------------------------------ hinge loss, steps
def _masked_mean(x, mask, weight):
    """GANLoss.mean, utils/losses.py:49-58"""
    weight = 1 if weight is None else weight
    if mask is None:
        return torch.mean(x) * weight
    return torch.mean(torch.sum(x * mask, dim=[1, 2, 3]) / torch.sum(mask, dim=[1, 2, 3])) * weight


def hinge(preds, target_is_real, for_discriminator, masks=None, weights=None):
    """GANLoss.__call__ + .loss for gan_mode='hinge', utils/losses.py:73-120"""
    total = 0
    for i, p in enumerate(preds):
        m = None if masks is None else masks[i]
        wt = None if weights is None else weights[i]
        if for_discriminator:
            v = torch.clamp_max((p if target_is_real else -p) - 1, 0)
            total = total - _masked_mean(v, m, wt)
        else:
            total = total - _masked_mean(p, m, wt)
    return total / (len(preds) if weights is None else sum(weights))


def d_weight(args):
    """main.py:486-489"""
    return [2, 1] if args.num_discriminators == 2 and args.texture_resolution >= 512 else None


def g_step(wg, wd, args, z, c, x_alpha, caption=None, symmetric=True):
    """ModelWrapper.forward('g'), main.py:491-498 -> (loss, pred_tex, pred_mesh, logits, masks); call loss.backward()"""
    pred_tex, pred_mesh = generator(wg, args, z, c, caption, symmetric, True)
    disc, mask = discriminator(wd, args, torch.cat((pred_tex * x_alpha, x_alpha), dim=1), pred_mesh, c, caption, True)
    loss = hinge(disc, True, False, mask if args.mask_output else None, d_weight(args))
    return loss, pred_tex, pred_mesh, disc, mask


def d_step(wg, wd, args, z, c, x_tex, x_alpha, x_mesh, caption=None, symmetric=True):
    """ModelWrapper.forward('d'), main.py:499-520 -> (loss_fake, loss_real, logits)"""
    B = z.shape[0]
    with torch.no_grad():
        ft, fm = generator(wg, args, z, c, caption, symmetric, True)
        xc = torch.cat((torch.cat((ft * x_alpha, x_alpha), 1), torch.cat((x_tex, x_alpha), 1)), 0)
        cc = torch.cat((c, c), 0) if c is not None else None
        capc = [torch.cat((t, t), 0) for t in caption] if caption is not None else None
        mc = torch.cat((fm, x_mesh), 0)
    disc, mask = discriminator(wd, args, xc, mc, cc, capc, True)
    mf = [t[:B] for t in mask] if args.mask_output else None
    mr = [t[B:] for t in mask] if args.mask_output else None
    w = d_weight(args)
    return hinge([t[:B] for t in disc], False, True, mf, w), hinge([t[B:] for t in disc], True, True, mr, w), disc


def adam_step(params, grads, state, lr, step, b1=0.0, b2=0.9, eps=1e-8):
    """torch.optim.Adam(betas=(0, 0.9)) as main.py:588-589 builds it (no weight decay, no amsgrad)"""
    with torch.no_grad():
        for k, p in params.items():
            g = grads.get(k)
            if g is None:
                continue
            m, v = state.setdefault(k, (torch.zeros_like(p), torch.zeros_like(p)))
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v.sqrt() / math.sqrt(1 - b2 ** step)).add_(eps)
            p.addcdiv_(m, denom, value=-lr / (1 - b1 ** step))


# ------------------------------------------------------------------------------------------------ bf16-faithful generator forward
# TEST INFRASTRUCTURE like the rest of this file.  The HIP path keeps activations and GEMM weight views in bf16 (fp32 accumulation,
# fp32 statistics, fp32 master weights): against the fp32 restatement above its outputs agree to a few 1e-3 -- bf16 noise through
# ~30 layers, which hides any logic error smaller than that.  This variant rounds WHERE THE HIP PATH ROUNDS (2dimageto3dmodel_amd/
# gan.py ResBlockUp.forward, csrc/gan_elem.hip k_affine_act, csrc/gan_glue.hip k_bn_finalize, conv.weight_prep), so that only the
# summation order inside the convolutions and reductions is left between the two: module-level agreement an order of magnitude
# tighter (tests/test_gan_modules.py::test_generator_forward_vs_bf16_faithful_oracle).
def _r(t):
    return t.bfloat16().float()


def _fma(x, a, b):
    """x * a + b rounded once (the kernels' fused multiply-add), via fp64"""
    return (x.double() * a.double() + b.double()).float()


def _conv_bf16(w, name, x, k, mode, training):
    """bf16 operands, fp32 accumulation: -> the fp32 results (the caller rounds them where the kernel's epilogue does)"""
    c = w.sub(name)
    ph = k // 2
    return F.conv2d(pad_w(x, ph, mode), _r(sn_weight(c, "weight", training)), c["bias"] if c.has("bias") else None, padding=(ph, 0))


def _cbn_act_bf16(w, name, y32, z, res=None, out_slope=1.0):
    """k_bn_finalize + k_affine_act on a conv's fp32 results y32: batch statistics of the fp32 values (the conv epilogue's partial
    sums), a = rstd (1 + gamma), b = beta - mean a; then on the bf16-rounded tensor: lrelu(fma(x, a, b)) -> bf16 [+ res -> bf16]
    [-> lrelu(out_slope) -> bf16]"""
    c = w.sub(name)
    mean = y32.mean(dim=(0, 2, 3))
    var = ((y32 * y32).mean(dim=(0, 2, 3)) - mean * mean).clamp_min(0.0)
    rstd = torch.rsqrt(var + 1e-5)
    gamma = F.linear(z, c["fc_gamma.weight"], c["fc_gamma.bias"])
    beta = F.linear(z, c["fc_beta.weight"], c["fc_beta.bias"])
    a = rstd[None, :] * (1.0 + gamma)
    b = beta - mean[None, :] * a
    t = _r(F.leaky_relu(_fma(_r(y32), a[:, :, None, None], b[:, :, None, None]), SLOPE))
    if res is not None:
        t = t + res
        if out_slope != 1.0:
            t = F.leaky_relu(_r(t), out_slope)
        t = _r(t)
    elif out_slope != 1.0:
        t = _r(F.leaky_relu(t, out_slope))
    return t


def _res_block_up_bf16(w, name, x, sc_in, z, mode, training, out_slope=1.0):
    """x: the block input at the block's resolution (already upsampled), sc_in: the same tensor BEFORE the upsample (the shortcut
    conv runs there and is read through the upsample: identical values, one rounding)"""
    b = w.sub(name)
    up = (lambda t: F.interpolate(t, scale_factor=2, mode="nearest")) if sc_in.shape[2] != x.shape[2] else (lambda t: t)
    sc = up(_r(_conv_bf16(b, "shortcut", sc_in, 1, "zero", training))) if b.has("shortcut.weight_orig") else up(sc_in)
    h = _cbn_act_bf16(b, "norm1", _conv_bf16(b, "conv1", x, 3, mode, training), z)
    return _cbn_act_bf16(b, "norm2", _conv_bf16(b, "conv2", h, 3, mode, training), z, sc, out_slope)


@torch.no_grad()
def generator_bf16(w, args, z, c=None, symmetric=True, training=True):
    """Generator.forward with the HIP path's storage precision (class-conditional or unconditional, batch / syncbatch norm, no
    text conditioning) -> (x_tex, x_mesh or None)"""
    assert args.norm_g in ("batch", "syncbatch") and not args.conditional_text and training
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
    mode = "replicate" if symmetric else "circular"
    if args.conditional_class:
        parts = [z, F.embedding(c[:, 0], w["emb_class.weight"])]
        if args.conditional_color:
            parts.append(F.embedding(c[:, 1], w["emb_color.weight"]))
        z = torch.cat(parts, dim=1)
    x = _r(F.linear(z, w["fc.weight"], w["fc.bias"]).view(z.shape[0], -1, 8, 4 if symmetric else 8))
    x = _res_block_up_bf16(w, "blk1", x, x, z, mode, training)
    x = _res_block_up_bf16(w, "blk2", up(x), x, z, mode, training)
    t = x
    for name, res in (("blk3a", 256), ("blk3b", 512), ("blk3c", 1024)):
        if args.texture_resolution >= res:
            t = _res_block_up_bf16(w, name, up(t), t, z, mode, training)
    t = _res_block_up_bf16(w, "blk4", up(t), t, z, mode, training)
    t = _res_block_up_bf16(w, "blk5", up(t), t, z, mode, training)
    t = _res_block_up_bf16(w, "blk6", up(t), t, z, mode, training, out_slope=SLOPE)
    x_tex = torch.tanh(F.conv2d(pad_w(t, 2, mode), _r(w["conv_final.weight"]), w["conv_final.bias"], padding=(2, 0)))
    x_mesh = None
    if w.has("conv_mesh.weight"):
        m = _res_block_up_bf16(w, "blk3_mesh", up(x), x, z, mode, training, out_slope=SLOPE)
        x_mesh = poles(F.conv2d(pad_w(m, 2, mode), _r(w["conv_mesh.weight"]), w["conv_mesh.bias"], padding=(2, 0)))
    if symmetric:
        x_tex = symmetrize(x_tex)
        if x_mesh is not None:
            x_mesh = symmetrize(x_mesh)
    return x_tex, x_mesh
