"""On-disk formats of the reference's two training scripts -- SURVEY.md 8f row 3 -- so that caches and checkpoints are
interchangeable with the reference in both directions:

  * pseudo-ground-truth cache  cache/<dataset>/pseudogt_<R>x<R>/<idx>.npz   (run_reconstruction.py:601-611 writes,
    data/abstract_dataset.py:68-81 reads): np.savez_compressed(data=<dict>), i.e. ONE pickled object array holding
    {'mesh' [3,32,32] f32, 'texture' [3,R,R] f16, 'texture_alpha' [1,R,R] f16, 'image' [C,299,299] f16} torch tensors;
  * poses_metadata.npz (run_reconstruction.py:613-622): {'scale', 'translation', 'rotation' tensors, 'path' list};
  * GAN checkpoints checkpoint_<it>.pth (main.py:749-770): torch.save of a dict with the keys of `CHECKPOINT_KEYS`;
  * OBJ / MTL / PNG export: MeshTemplate.export_obj (mesh.py).
Host-side I/O only; nothing here touches the GPU.
"""
import os

import numpy as np
import torch

CHECKPOINT_KEYS = ("optimizer_g", "optimizer_d", "generator", "generator_running_avg", "discriminator", "epoch", "iteration",
                   "g_curve", "d_fake_curve", "d_real_curve", "flat_curve", "args")


def pseudogt_dir(cache_dir, texture_resolution):
    return os.path.join(cache_dir, f"pseudogt_{texture_resolution}x{texture_resolution}")


def save_pseudo_ground_truth(cache_dir, texture_resolution, idx, mesh, texture, texture_alpha, image):
    """one sample of the reconstruction stage's output (run_reconstruction.py:590-611): half precision for everything
    but the mesh displacement map, CPU tensors, one pickled dict per file"""
    d = pseudogt_dir(cache_dir, texture_resolution)
    os.makedirs(d, exist_ok=True)
    data = {
        "mesh": mesh.detach().float().cpu().clone(),
        "texture": texture.detach().half().cpu().clone(),
        "texture_alpha": texture_alpha.detach().half().cpu().clone(),
        "image": image.detach().half().cpu().clone(),
    }
    np.savez_compressed(os.path.join(d, f"{int(idx)}"), data=data)


def load_pseudo_ground_truth(cache_dir, texture_resolution, idx):
    """data/abstract_dataset.py:68-81: -> {'image' [3,299,299] in [0,1], 'texture', 'texture_alpha' (fp32), 'mesh'}"""
    z = np.load(os.path.join(pseudogt_dir(cache_dir, texture_resolution), f"{int(idx)}.npz"), allow_pickle=True)
    data = z["data"].item()
    return {
        "image": data["image"][:3].float() / 2 + 0.5,
        "texture": data["texture"].float(),
        "texture_alpha": data["texture_alpha"].float(),
        "mesh": data["mesh"],
    }


def save_poses_metadata(cache_dir, scale, translation, rotation, paths):
    """run_reconstruction.py:613-622"""
    os.makedirs(cache_dir, exist_ok=True)
    data = {"scale": scale.detach().cpu().clone(), "translation": translation.detach().cpu().clone(),
            "rotation": rotation.detach().cpu().clone(), "path": list(paths)}
    if not (len(data["path"]) == data["scale"].shape[0] == data["translation"].shape[0] == data["rotation"].shape[0]):
        raise ValueError("poses_metadata: scale / translation / rotation / path must have one entry per image")
    np.savez_compressed(os.path.join(cache_dir, "poses_metadata"), data=data)


def load_poses_metadata(cache_dir):
    return np.load(os.path.join(cache_dir, "poses_metadata.npz"), allow_pickle=True)["data"].item()


def save_checkpoint(path, trainer, epoch, g_curve=(), d_fake_curve=(), d_real_curve=(), flat_curve=(), args=None):
    """main.py:749-770 from a train.GanTrainer; `args` defaults to the trainer's namespace"""
    a = args if args is not None else trainer.args
    # data parallel with overlap_comm: the last discriminator step's all-reduce + optimizer_d.step() may still be pending (it is
    # normally completed inside the next forward, train.GanTrainer); the optimiser / discriminator state read below must be the
    # state AFTER step total_it, as the reference's checkpoint is (main.py:749-770)
    if hasattr(trainer, "finish_pending"):
        trainer.finish_pending()
    out = {
        "optimizer_g": trainer.optimizer_g.state_dict(),
        "optimizer_d": trainer.optimizer_d.state_dict(),
        "generator": trainer.generator.state_dict(),
        "generator_running_avg": trainer.generator_running_avg.state_dict(),
        "discriminator": trainer.discriminator.state_dict(),
        "epoch": int(epoch),
        "iteration": int(trainer.total_it),
        "g_curve": list(g_curve),
        "d_fake_curve": list(d_fake_curve),
        "d_real_curve": list(d_real_curve),
        "flat_curve": list(flat_curve),
        "args": dict(vars(a)) if not isinstance(a, dict) else dict(a),
    }
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(out, path)
    return out


def load_checkpoint(path, trainer, map_location="cpu", strict=True):
    """restore a checkpoint written by this package OR by the reference's main.py into a GanTrainer"""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    missing = [k for k in ("generator", "generator_running_avg", "discriminator") if k not in ck]
    if missing:
        raise KeyError(f"{path}: not a GAN checkpoint (missing {missing})")
    # a discriminator step still pending here would be applied -- with the OLD run's averaged gradients -- to the loaded weights by
    # the next forward: complete it first (its result is overwritten below)
    if hasattr(trainer, "finish_pending"):
        trainer.finish_pending()
    trainer.generator.load_state_dict(ck["generator"], strict=strict)
    trainer.generator_running_avg.load_state_dict(ck["generator_running_avg"], strict=strict)
    trainer.discriminator.load_state_dict(ck["discriminator"], strict=strict)
    if "optimizer_g" in ck:
        trainer.optimizer_g.load_state_dict(ck["optimizer_g"])
    if "optimizer_d" in ck:
        trainer.optimizer_d.load_state_dict(ck["optimizer_d"])
    trainer.total_it = int(ck.get("iteration", 0))
    return ck
