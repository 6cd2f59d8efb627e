"""ReconstructionNetwork (code/models/reconstruction.py:7-137) -- SURVEY.md 8f row 4: the image -> (texture, mesh
displacement map) network of run_reconstruction.py, on the same NHWC bf16 MFMA conv kernels and fused
BatchNorm + activation kernels as the GAN path (csrc/conv_*.hip, gan_elem.hip, gan_glue.hip).

Same constructor arguments, sub-module names, parameter / buffer names and initialisers as the reference, so
`state_dict`s are interchangeable and `torch.manual_seed` reproduces the reference's weights.  Differences of
execution, not of meaning:
  * the nearest x2 upsample that FOLLOWS a block (`self.up(self.blkN(bb))`, :112-127) is folded into the NEXT block's
    first conv and shortcut (index arithmetic in the conv loader), the replicate / circular W pads likewise;
  * BatchNorm2d + ReLU (+ the residual add of ResBlock.forward :24-25) is one fused pass;
  * stride-2 5x5 / 3x3 encoder convs: forward on the generic implicit-GEMM kernel, dgrad through the padded even kernel.
interpolation_mode='bilinear' (:43-44, non-default): the x2 upsample cannot be folded into a conv loader (it mixes four
pixels); it runs between the blocks as F.interpolate on the NHWC bf16 activation, everything else is unchanged.
`DatasetParams` (:140-180): the per-image translation / scale / z0 offsets run_reconstruction.py optimises next to the network.
"""
import torch
import torch.nn as nn

from . import conv as C
from . import gan_ops as G
from .gan import Conv2d, _Identity, adjust_poles, symmetrize_texture


class BatchNormAct2d(G.BatchNorm2d):
    """nn.BatchNorm2d(ch) (affine) followed by ReLU / LeakyReLU [+ residual], fused; parameter and buffer names are
    nn.BatchNorm2d's (weight, bias, running_mean, running_var, num_batches_tracked)"""

    def __init__(self, ch, eps=1e-5, momentum=0.1):
        super().__init__(ch, affine=False, eps=eps, momentum=momentum)
        self.weight = nn.Parameter(torch.ones(ch))
        self.bias = nn.Parameter(torch.zeros(ch))

    def forward(self, x, slope=0.0, res=None, out_slope=1.0):
        n = x.shape[0]
        # the fused kernels take per-sample (gamma, beta) with y = x_hat * (1 + gamma) + beta
        gamma = (self.weight - 1.0).unsqueeze(0).expand(n, -1)
        beta = self.bias.unsqueeze(0).expand(n, -1)
        return super().forward(x, gamma, beta, slope, res, out_slope)


class BatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d of the encoder's two fully connected layers (code/models/reconstruction.py:64-68), same parameters / buffers /
    state_dict.  `sync` (set by recon_train.ReconTrainer for data-parallel runs): training-mode statistics over the GLOBAL batch --
    one all-reduce of [sum | sum of squares | count] per layer, its adjoint in the backward (gan_ops._SyncMoments) -- so that N
    ranks with B / N samples each normalise exactly as one process with B samples does (a batch norm over a rank's 2-sample shard is a
    different function).  Without sync, or outside training, this is nn.BatchNorm1d itself."""
    sync = False

    def forward(self, x):
        from . import parallel
        if not (self.sync and self.training and parallel.collectives_on()):
            return super().forward(x)
        cnt = x.new_tensor([float(x.shape[0])])
        v = G._SyncMoments.apply(torch.cat((x.sum(0), (x * x).sum(0), cnt)))
        c = x.shape[1]
        n = v[2 * c]
        mean = v[:c] / n
        var = (v[c:2 * c] / n - mean * mean).clamp_min(0)
        with torch.no_grad():
            m = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked + 1)
            self.running_mean.mul_(1 - m).add_(mean.detach(), alpha=m)
            self.running_var.mul_(1 - m).add_((var * (n / (n - 1))).detach() if float(n) > 1 else var.detach(), alpha=m)
            self.num_batches_tracked += 1
        return (x - mean) * torch.rsqrt(var + self.eps) * self.weight + self.bias


class ResBlock(nn.Module):
    """models/reconstruction.py:7-26"""

    def __init__(self, ch_in, ch_out, pad_mode):
        super().__init__()
        self.conv1 = Conv2d(ch_in, ch_in, 3, pad_h=1, pad_w=1, pad_w_mode=pad_mode, bias=False)
        self.conv2 = Conv2d(ch_in, ch_out, 3, pad_h=1, pad_w=1, pad_w_mode=pad_mode, bias=False)
        self.bn1 = BatchNormAct2d(ch_in)
        self.bn2 = BatchNormAct2d(ch_out)
        if ch_in != ch_out:
            self.shortcut = Conv2d(ch_in, ch_out, 1, bias=False)
        else:
            self.shortcut = _Identity()

    def forward(self, x, upsample=0, out_slope=1.0):
        """x: the block input BEFORE the nearest x2 upsample that the caller applied to the previous block's output
        (upsample=1); the shortcut stays at the input resolution and is read through the upsample by the fused add.
        out_slope = 0: the ReLU the network applies to this block's output in front of a head, fused into the last pass
        (its backward is applied by the head, gan_ops.head_conv(in_slope=0))"""
        sc = x if isinstance(self.shortcut, _Identity) else self.shortcut(x)
        h = self.bn1(self.conv1(x, upsample=upsample))
        return self.bn2(self.conv2(h), res=sc, out_slope=out_slope)


def _up_bilinear(x_nhwc):
    """F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False) (models/reconstruction.py:44) of an NHWC bf16
    activation: the interpolation itself in fp32 (as the reference's fp32 activations), stored back as bf16"""
    y = torch.nn.functional.interpolate(x_nhwc.permute(0, 3, 1, 2).float(), scale_factor=2, mode='bilinear', align_corners=False)
    return y.permute(0, 2, 3, 1).contiguous().to(G._act())   # (bf16, or fp32 under the EXACT build: _lib.act_dtype)


class DatasetParams(nn.Module):
    """models/reconstruction.py:140-180: learnable per-image corrections of the (noisy) ground-truth poses --
    ds_translation [N,2], ds_scale [N,1] (args.optimize_deltas) and ds_z0 [N,1] (args.optimize_z0, perspective strength).
    forward(indices, 'deltas') -> (translation_delta [B,3] with z = 0, scale_delta [B,1]); forward(indices, 'z0') ->
    1 + exp(z0) [B,1].  indices in [N, 2N) address the mirrored copy of image index - N (data augmentation): the x translation
    changes sign.  indices None: the dataset mean (evaluation of unseen images).  Same parameter names and shapes as the
    reference, so its checkpoints load."""

    def __init__(self, args, dataset_size):
        super().__init__()
        self.dataset_size = dataset_size
        if args.optimize_deltas:
            self.ds_translation = nn.Parameter(torch.zeros(dataset_size, 2))
            self.ds_scale = nn.Parameter(torch.zeros(dataset_size, 1))
        if args.optimize_z0:
            self.ds_z0 = nn.Parameter(torch.ones(dataset_size, 1))

    def forward(self, indices, mode):
        assert mode in ['deltas', 'z0']
        mirrored_sign = 1
        if indices is not None:
            mirrored_sign = (1 - 2 * torch.div(indices, self.dataset_size, rounding_mode='floor').float()).unsqueeze(-1)
            indices = indices % self.dataset_size
        pick = (lambda table: table[indices]) if indices is not None else (lambda table: table.mean(dim=0, keepdim=True))
        if mode == 'z0':
            return 1 + torch.exp(pick(self.ds_z0))
        t = pick(self.ds_translation)
        translation_delta = torch.cat((t[:, :1] * mirrored_sign, t[:, 1:2], torch.zeros_like(t[:, :1])), dim=1)
        return translation_delta, pick(self.ds_scale)


class ReconstructionNetwork(nn.Module):
    """models/reconstruction.py:28-137"""

    def __init__(self, symmetric=True, texture_res=64, mesh_res=32, interpolation_mode='nearest'):
        super().__init__()
        self.symmetric = symmetric
        self.pad = C.PAD_REPLICATE if symmetric else C.PAD_CIRCULAR   # (:34-37)
        if interpolation_mode not in ('nearest', 'bilinear'):
            raise ValueError(f"interpolation_mode={interpolation_mode!r}")   # (the reference: a bare `raise`, :45-46)
        self.interpolation_mode = interpolation_mode
        assert mesh_res >= 32
        assert texture_res >= 64

        # encoder: five stride-2 convolutions halve 256 x 256 down to 8 x 8 (:53-63)
        self.conv1e = Conv2d(4, 64, 5, stride=2, pad_h=2, pad_w=2, bias=False)
        self.bn1e = BatchNormAct2d(64)
        self.conv2e = Conv2d(64, 128, 3, stride=2, pad_h=1, pad_w=1, bias=False)
        self.bn2e = BatchNormAct2d(128)
        self.conv3e = Conv2d(128, 256, 3, stride=2, pad_h=1, pad_w=1, bias=False)
        self.bn3e = BatchNormAct2d(256)
        self.conv4e = Conv2d(256, 512, 3, stride=2, pad_h=1, pad_w=1, bias=False)
        self.bn4e = BatchNormAct2d(512)

        bottleneck_dim = 256
        self.conv5e = Conv2d(512, 64, 3, stride=2, pad_h=1, pad_w=1, bias=False)
        self.bn5e = BatchNormAct2d(64)
        self.fc1e = nn.Linear(64 * 8 * 8, bottleneck_dim, bias=False)
        self.bnfc1e = BatchNorm1d(bottleneck_dim)
        self.fc3e = nn.Linear(bottleneck_dim, 1024, bias=False)
        self.bnfc3e = BatchNorm1d(1024)

        # texture decoder: 4 x (2|4) seed, one x2 upsample after every block but the last (:72-90); texture_res 128 / 256
        # insert one / two more 256-channel blocks
        self.base_res_h = 4
        self.base_res_w = 2 if symmetric else 4
        self.fc1_tex = nn.Linear(1024, self.base_res_h * self.base_res_w * 256)
        self.blk1 = ResBlock(256, 512, self.pad)
        self.blk2 = ResBlock(512, 256, self.pad)
        self.blk3 = ResBlock(256, 256, self.pad)
        assert texture_res in [64, 128, 256]
        self.texture_res = texture_res
        if texture_res >= 128:
            self.blk3b_tex = ResBlock(256, 256, self.pad)
        if texture_res >= 256:
            self.blk3c_tex = ResBlock(256, 256, self.pad)
        self.blk4_tex = ResBlock(256, 128, self.pad)
        self.blk5_tex = ResBlock(128, 64, self.pad)
        self.conv_tex = Conv2d(64, 3, 5, pad_h=2, pad_w=2, pad_w_mode=self.pad)

        # displacement-map decoder branches off at 32 x 32 (:92-99)
        self.blk4_mesh = ResBlock(256, 64, self.pad)
        self.conv_mesh = Conv2d(64, 3, 5, pad_h=2, pad_w=2, pad_w_mode=self.pad)
        # zero-initialised mesh output layer for stability (avoids self-intersections, :97-99)
        self.conv_mesh.bias.data[:] = 0
        self.conv_mesh.weight.data[:] = 0

    def encode_convs(self, x):
        """conv1e .. bn5e (:108-112): x [B,4,256,256] NCHW fp32 -> [B,64,8,8] NCHW fp32"""
        h = G.to_nhwc_bf16(x, pad_to=8)
        for conv, bn in ((self.conv1e, self.bn1e), (self.conv2e, self.bn2e), (self.conv3e, self.bn3e),
                         (self.conv4e, self.bn4e), (self.conv5e, self.bn5e)):
            h = bn(conv(h))
        return G.to_nchw_f32(h)

    def encode(self, x):
        """image -> bottleneck code z [B,1024] (:108-116)"""
        z = self.encode_convs(x).reshape(x.shape[0], -1)        # flatten in the reference's (C,H,W) order
        z = torch.relu(self.bnfc1e(self.fc1e(z)))
        return torch.relu(self.bnfc3e(self.fc3e(z)))

    def decode(self, z):
        """bottleneck code -> (tex [B,3,R,R], mesh_map [B,3,32,32]) NCHW fp32 (:118-137).  The heads are the generator's:
        ReLU fused into the last block's pass, conv, then tanh_ / adjust_poles / symmetrize in one elementwise kernel."""
        bb = G.to_nhwc_bf16(self.fc1_tex(z).view(z.shape[0], -1, self.base_res_h, self.base_res_w))
        # nearest: every later block starts with the x2 upsample of the line before it, folded into its first conv (u = 1);
        # bilinear: the upsample is a pass of its own between the blocks (up), the blocks run at u = 0
        nearest = self.interpolation_mode == 'nearest'
        u = 1 if nearest else 0
        up = (lambda t: t) if nearest else _up_bilinear
        bb = up(self.blk1(bb))
        bb = up(self.blk2(bb, upsample=u))
        bb = up(self.blk3(bb, upsample=u))
        bb_mesh = bb
        if self.texture_res >= 128:
            bb = up(self.blk3b_tex(bb, upsample=u))
        if self.texture_res >= 256:
            bb = up(self.blk3c_tex(bb, upsample=u))
        sym = G.HT_SYMM if self.symmetric else 0
        # (:126, :133: blk4_mesh and blk5_tex are NOT followed by an upsample; blk4_tex is)
        mesh_map = G.head_conv(self.blk4_mesh(bb_mesh, upsample=u, out_slope=0.0), self.conv_mesh, G.HT_POLES | sym, in_slope=0.0)
        tex = up(self.blk4_tex(bb, upsample=u))
        tex = G.head_conv(self.blk5_tex(tex, upsample=u, out_slope=0.0), self.conv_tex, G.HT_TANH | sym, in_slope=0.0)
        return tex, mesh_map

    def forward(self, x):
        """x [B,4,256,256] (RGB + mask, NCHW fp32) -> (tex [B,3,R,R], mesh_map [B,3,32,32]) NCHW fp32 (:106-137)"""
        return self.decode(self.encode(x))
