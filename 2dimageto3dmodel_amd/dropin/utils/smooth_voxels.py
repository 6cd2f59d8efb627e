"""Drop-in for code/utils/smooth_voxels.py."""
from _m355 import projection as _p

VoxelsSmooth = _p.VoxelsSmooth
