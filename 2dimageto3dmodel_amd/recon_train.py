"""The mesh-estimation training step of code/run_reconstruction.py (the loop body :409-445 with `transform_vertices`
:237-252 and `mean_iou` :225-231) composed from the drop-in pieces -- SURVEY.md 8f rows 1, 2 and 4 in one step:

    image [B,4,256,256] --ReconstructionNetwork--> (texture, displacement map)            reconstruction.py  (MFMA convs)
    displacement map --MeshTemplate.get_vertex_positions--> object-space vertices           mesh.py            (mesh_deform.hip)
    vertices --transform_vertices (scale, qrot, translate, flip y/z [, z0 perspective])--> camera-space vertices
    (vertices, texture) --MeshTemplate.forward_renderer(Renderer)--> (image, alpha)          render.py          (dibr_raster.hip)
    loss = criterion(cat(image, alpha), input) + mesh_regularization * warmup * loss_flat(normals of the raw vertices)

The rasteriser behind `Renderer` is the from-scratch DIB-R of csrc/dibr_raster.hip: PARITY UNPINNED for that stage (Kaolin is
neither in this image nor under /root/reference, see render.py); every other stage is pinned by reference-executed goldens.
Not reproduced: datasets / loaders, checkpoint files, tensorboard, pseudo-ground-truth export (host code outside the hot path).
"""
import torch
import torch.nn as nn

from . import mesh as M
from . import parallel as P
from .reconstruction import BatchNorm1d, BatchNormAct2d, DatasetParams, ReconstructionNetwork
from .render import Renderer


def mean_iou(alpha_pred, alpha_real):
    """run_reconstruction.py:225-231: intersection over union of the thresholded silhouettes [B,H,W], batch mean"""
    a, b = alpha_pred > 0.5, alpha_real > 0.5
    inter = (a & b).float().sum(dim=[1, 2])
    union = (a | b).float().sum(dim=[1, 2])
    return torch.mean(inter / union)


def transform_vertices(vtx, gt_scale, gt_translation, gt_rot, gt_idx=None, dataset_params=None, optimize_deltas=True,
                       optimize_z0=False):
    """run_reconstruction.py:237-252: object space -> the renderer's camera space.
    vtx [B,V,3]; gt_scale [B,1]; gt_translation [B,3]; gt_rot [B,4] quaternion; gt_idx [B] dataset indices (or None: the
    dataset-mean offsets) when `dataset_params` holds learnable per-image deltas (optimize_deltas) / perspective (optimize_z0)."""
    scale_delta, translation_delta = 0, 0
    if optimize_deltas and dataset_params is not None:
        translation_delta, scale_delta = dataset_params(gt_idx, 'deltas')
    vtx = M.qrot(gt_rot, (gt_scale + scale_delta).unsqueeze(-1) * vtx) + (gt_translation + translation_delta).unsqueeze(1)
    vtx = vtx * vtx.new_tensor([1.0, -1.0, -1.0])
    if optimize_z0:
        z0 = dataset_params(gt_idx, 'z0').unsqueeze(-1)
        z = vtx[:, :, 2:]
        factor = (z0 + z / 2) / (z0 - z / 2)
        vtx = torch.cat((vtx[:, :, :2] * factor, z), dim=2)
    elif dataset_params is not None:
        assert 'ds_z0' not in dataset_params._parameters, 'Model was trained with --optimize_z0'
    return vtx


class ReconTrainer(nn.Module):
    """generator / mesh template / renderer / optimisers of run_reconstruction.py:79-89,332-355 and one call per loop iteration.
    Defaults are the script's argparse defaults (:37-64): texture 128, mesh map 32, image 256, MSE, lr 1e-4 for both
    optimisers, mesh_regularization 5e-5 with the 10 -> 1 warm-up of :355,439-440, optimize_deltas on, optimize_z0 off.

    Data parallel (one process per GPU, torch.distributed initialised -- parallel.init_from_env): the reference script is single-GPU,
    so the N-rank step is defined as THE SAME STEP ON THE GLOBAL BATCH: every rank holds the whole model and the whole
    DatasetParams table, takes its shard of the loader batch, and
      * the network's batch norms take their statistics over the global batch (sync_bn: the fused [sum | sum of squares | count]
        all-reduce of gan_ops, one message per layer and direction) -- without it a rank would normalise with its shard's
        statistics, which is a different model from the single-GPU one;
      * both losses are means over the rank's samples, equal shards: ONE flat all-reduce (MEAN) of the generator's and the
        DatasetParams' gradients per iteration (parallel.FlatGradReducer) gives the global-batch gradient -- the per-image rows of
        the DatasetParams table a rank did not touch contribute zeros, as they do in the single-GPU step;
      * parameters are broadcast from rank 0 once; the optimisers then stay in lock step (identical gradients, identical state).
    tests/test_distributed_gpu.py runs two ranks against the single-process step on the concatenated batch."""

    def __init__(self, mesh_template, dataset_size=None, symmetric=True, texture_resolution=128, mesh_resolution=32,
                 image_resolution=256, loss='mse', lr=1e-4, lr_dataset=1e-4, mesh_regularization=0.00005, optimize_deltas=True,
                 optimize_z0=False, interpolation_mode='nearest', device='cuda', data_parallel=True, sync_bn=True):
        super().__init__()
        import argparse
        self.mesh_template = mesh_template
        self.generator = ReconstructionNetwork(symmetric=symmetric, texture_res=texture_resolution, mesh_res=mesh_resolution,
                                               interpolation_mode=interpolation_mode)
        self.renderer = Renderer(image_resolution, image_resolution)   # (:86-89: the network's input resolution)
        self.optimize_deltas, self.optimize_z0 = bool(optimize_deltas), bool(optimize_z0)
        self.dataset_params = None
        if (optimize_deltas or optimize_z0) and dataset_size is not None:
            self.dataset_params = DatasetParams(argparse.Namespace(optimize_deltas=optimize_deltas, optimize_z0=optimize_z0),
                                                dataset_size)
        self.criterion = {'mse': nn.MSELoss(), 'l1': nn.L1Loss()}[loss]
        self.mesh_regularization, self.flat_warmup = mesh_regularization, 10.0
        self.to(device)
        self.data_parallel = bool(data_parallel)
        self.reduce = None
        if self.data_parallel:
            for m in (self.generator, self.dataset_params):
                if m is not None:
                    P.broadcast_parameters(m)
            if sync_bn:
                for m in self.generator.modules():
                    if isinstance(m, BatchNormAct2d):
                        m.sync, m._count_batches = True, True   # global-batch statistics; still counts like nn.BatchNorm2d
                    elif isinstance(m, BatchNorm1d):
                        m.sync = True                           # (the two fully connected layers of the encoder)
            self.reduce = P.FlatGradReducer(list(self.generator.parameters()) +
                                            ([] if self.dataset_params is None else list(self.dataset_params.parameters())))
        self.optimizer = torch.optim.Adam(self.generator.parameters(), lr=lr)
        self.optimizer_dataset = None if self.dataset_params is None else torch.optim.Adam(self.dataset_params.parameters(),
                                                                                           lr=lr_dataset)
        self.total_it = 0

    def render(self, X_real, gt_scale, gt_translation, gt_rot, gt_idx=None):
        """the forward half of an iteration (:421-430) -> (X_fake [B,4,H,W], raw_vtx [B,V,3], pred_tex, mesh_map)"""
        pred_tex, mesh_map = self.generator(X_real)
        raw_vtx = self.mesh_template.get_vertex_positions(mesh_map)
        vtx = transform_vertices(raw_vtx, gt_scale, gt_translation, gt_rot, gt_idx, self.dataset_params, self.optimize_deltas,
                                 self.optimize_z0)
        image_pred, alpha_pred = self.mesh_template.forward_renderer(self.renderer, vtx, pred_tex)
        X_fake = torch.cat((image_pred, alpha_pred), dim=3).permute(0, 3, 1, 2)
        return X_fake, raw_vtx, pred_tex, mesh_map

    def losses(self, X_real, gt_scale, gt_translation, gt_rot, gt_idx=None):
        """-> (total, recon_loss, flat_loss, miou, X_fake) with the CURRENT warm-up coefficient (does not advance it)"""
        X_fake, raw_vtx, _, _ = self.render(X_real, gt_scale, gt_translation, gt_rot, gt_idx)
        recon_loss = self.criterion(X_fake, X_real)
        flat_loss = M.loss_flat(self.mesh_template.mesh, self.mesh_template.compute_normals(raw_vtx))
        with torch.no_grad():
            miou = mean_iou(X_fake[:, 3], X_real[:, 3])   # on the alpha channel
        total = recon_loss + (self.mesh_regularization * self.flat_warmup) * flat_loss
        return total, recon_loss, flat_loss, miou, X_fake

    def iteration(self, X_real, gt_scale, gt_translation, gt_rot, gt_idx=None):
        """one pass of the loop body :409-445; returns the scalar losses (tensors, no host sync)"""
        if self.dataset_params is None:
            gt_idx = None      # (:416-419)
        self.optimizer.zero_grad(set_to_none=True)
        if self.optimizer_dataset is not None:
            self.optimizer_dataset.zero_grad(set_to_none=True)
        total, recon_loss, flat_loss, miou, _ = self.losses(X_real, gt_scale, gt_translation, gt_rot, gt_idx)
        self.flat_warmup = max(self.flat_warmup - 0.1, 1)
        total.backward()
        if self.reduce is not None:
            self.reduce()          # (no-op unless more than one rank: the gradients become the global-batch means)
        self.optimizer.step()
        if self.optimizer_dataset is not None:
            self.optimizer_dataset.step()
        self.total_it += 1
        return {"loss": total.detach(), "recon_loss": recon_loss.detach(), "flat_loss": flat_loss.detach(), "miou": miou}

    def decay_lr(self, factor=0.5):
        """the halving every `lr_decay_every` epochs (:468-470) -- the generator's optimiser only, as the script"""
        for group in self.optimizer.param_groups:
            group['lr'] *= factor
