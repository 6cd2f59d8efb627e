"""MI355X-native hot path of NikolaZubic/2dimageto3dmodel (point-cloud projection + silhouette loss, GAN convs).

The directory name starts with a digit, so import it with
    importlib.import_module("2dimageto3dmodel_amd")
or put `2dimageto3dmodel_amd/dropin` on sys.path and use the reference's own module paths.
"""
from . import _lib, ops  # noqa: F401
from .projection import (CameraUtilities, EffectiveLossFunction, SupervisedLoss,  # noqa: F401
                         TrilinearInterpolation, UnsupervisedLoss, VoxelsSmooth)
