"""MI355X-native hot path of NikolaZubic/2dimageto3dmodel (point-cloud projection + silhouette loss, GAN convs).

The directory name starts with a digit, so import it with
    importlib.import_module("2dimageto3dmodel_amd")
or put `2dimageto3dmodel_amd/dropin` on sys.path and use the reference's own module paths.

Modules: projection / ops (point-cloud projection + silhouette loss), gan / gan_ops / conv / train (GAN stacks and
their training iteration), parallel (RCCL reducers), and the SURVEY 8f widenings mesh (template deformation + flat
loss), reconstruction (ReconstructionNetwork), formats (on-disk caches and checkpoints).
"""
from . import _lib, ops  # noqa: F401
from .conv import is_deterministic, set_deterministic  # noqa: F401  (M355_DETERMINISTIC=1: bit-reproducible training cycles)
from .projection import (CameraUtilities, EffectiveLossFunction, SupervisedLoss,  # noqa: F401
                         TrilinearInterpolation, UnsupervisedLoss, VoxelsSmooth)
