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
Only interpolation_mode='nearest' (the default) is supported.
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

    def forward(self, x, slope=0.0, res=None):
        n = x.shape[0]
        # the fused kernels take per-sample (gamma, beta) with y = x_hat * (1 + gamma) + beta
        gamma = (self.weight - 1.0).unsqueeze(0).expand(n, -1)
        beta = self.bias.unsqueeze(0).expand(n, -1)
        return super().forward(x, gamma, beta, slope, res)


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

    def forward(self, x, upsample=0):
        """x: the block input BEFORE the nearest x2 upsample that the caller applied to the previous block's output
        (upsample=1); the shortcut stays at the input resolution and is read through the upsample by the fused add"""
        sc = x if isinstance(self.shortcut, _Identity) else self.shortcut(x)
        h = self.bn1(self.conv1(x, upsample=upsample))
        return self.bn2(self.conv2(h), res=sc)


class ReconstructionNetwork(nn.Module):
    """models/reconstruction.py:28-137"""

    def __init__(self, symmetric=True, texture_res=64, mesh_res=32, interpolation_mode='nearest'):
        super().__init__()
        self.symmetric = symmetric
        self.pad = C.PAD_REPLICATE if symmetric else C.PAD_CIRCULAR   # (:34-37)
        if interpolation_mode != 'nearest':
            raise NotImplementedError("ReconstructionNetwork: only interpolation_mode='nearest' runs on the fused kernels")
        assert mesh_res >= 32
        assert texture_res >= 64

        self.conv1e = Conv2d(4, 64, 5, stride=2, pad_h=2, pad_w=2, bias=False)       # 256 -> 128
        self.bn1e = BatchNormAct2d(64)
        self.conv2e = Conv2d(64, 128, 3, stride=2, pad_h=1, pad_w=1, bias=False)     # -> 64
        self.bn2e = BatchNormAct2d(128)
        self.conv3e = Conv2d(128, 256, 3, stride=2, pad_h=1, pad_w=1, bias=False)    # -> 32
        self.bn3e = BatchNormAct2d(256)
        self.conv4e = Conv2d(256, 512, 3, stride=2, pad_h=1, pad_w=1, bias=False)    # -> 16
        self.bn4e = BatchNormAct2d(512)

        bottleneck_dim = 256
        self.conv5e = Conv2d(512, 64, 3, stride=2, pad_h=1, pad_w=1, bias=False)     # -> 8
        self.bn5e = BatchNormAct2d(64)
        self.fc1e = nn.Linear(64 * 8 * 8, bottleneck_dim, bias=False)
        self.bnfc1e = nn.BatchNorm1d(bottleneck_dim)
        self.fc3e = nn.Linear(bottleneck_dim, 1024, bias=False)
        self.bnfc3e = nn.BatchNorm1d(1024)

        # texture generation
        self.base_res_h = 4
        self.base_res_w = 2 if symmetric else 4
        self.fc1_tex = nn.Linear(1024, self.base_res_h * self.base_res_w * 256)
        self.blk1 = ResBlock(256, 512, self.pad)    # 4 -> 8
        self.blk2 = ResBlock(512, 256, self.pad)    # 8 -> 16
        self.blk3 = ResBlock(256, 256, self.pad)    # 16 -> 32 (k=1)
        assert texture_res in [64, 128, 256]
        self.texture_res = texture_res
        if texture_res >= 128:
            self.blk3b_tex = ResBlock(256, 256, self.pad)   # k = 2
        if texture_res >= 256:
            self.blk3c_tex = ResBlock(256, 256, self.pad)   # k = 4
        self.blk4_tex = ResBlock(256, 128, self.pad)        # k*32 -> k*64
        self.blk5_tex = ResBlock(128, 64, self.pad)         # k*64 -> k*64 (no upsampling)
        self.conv_tex = Conv2d(64, 3, 5, pad_h=2, pad_w=2, pad_w_mode=self.pad)

        # mesh generation
        self.blk4_mesh = ResBlock(256, 64, self.pad)        # 32 -> 32 (no upsampling)
        self.conv_mesh = Conv2d(64, 3, 5, pad_h=2, pad_w=2, pad_w_mode=self.pad)
        # zero-initialised mesh output layer for stability (avoids self-intersections, :97-99)
        self.conv_mesh.bias.data[:] = 0
        self.conv_mesh.weight.data[:] = 0

    def forward(self, x):
        """x [B,4,256,256] (RGB + mask, NCHW fp32) -> (tex [B,3,R,R], mesh_map [B,3,32,32]) NCHW fp32 (:106-137)"""
        h = G.to_nhwc_bf16(x, pad_to=8)
        for conv, bn in ((self.conv1e, self.bn1e), (self.conv2e, self.bn2e), (self.conv3e, self.bn3e),
                         (self.conv4e, self.bn4e), (self.conv5e, self.bn5e)):
            h = bn(conv(h))
        z = G.to_nchw_f32(h).reshape(h.shape[0], -1)            # flatten in the reference's (C,H,W) order
        z = torch.relu(self.bnfc1e(self.fc1e(z)))
        z = torch.relu(self.bnfc3e(self.fc3e(z)))

        bb = G.to_nhwc_bf16(self.fc1_tex(z).view(z.shape[0], -1, self.base_res_h, self.base_res_w))
        bb = self.blk1(bb)
        bb = self.blk2(bb, upsample=1)      # every later block starts with the x2 upsample of the line before it
        bb = self.blk3(bb, upsample=1)
        bb_mesh = bb
        if self.texture_res >= 128:
            bb = self.blk3b_tex(bb, upsample=1)
        if self.texture_res >= 256:
            bb = self.blk3c_tex(bb, upsample=1)

        mesh_map = self.blk4_mesh(bb_mesh, upsample=1)
        mesh_map = adjust_poles(self.conv_mesh(torch.relu(mesh_map), out_f32_nchw=True))

        tex = self.blk4_tex(bb, upsample=1)
        tex = self.blk5_tex(tex, upsample=1)
        tex = torch.tanh(self.conv_tex(torch.relu(tex), out_f32_nchw=True))

        if self.symmetric:
            tex = symmetrize_texture(tex)
            mesh_map = symmetrize_texture(mesh_map)
        return tex, mesh_map
