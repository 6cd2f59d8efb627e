"""Drop-in for code/rendering/utils.py: the tensor helpers of :6-64 used by the GAN path, the mesh template and the
training scripts (grid_sample_bilinear, symmetrize_texture, adjust_poles, circpad, qrot, qmul)."""
import importlib

from _m355 import pkg as _pkg  # noqa: F401

_g = importlib.import_module("2dimageto3dmodel_amd.gan")
_m = importlib.import_module("2dimageto3dmodel_amd.mesh")
symmetrize_texture = _g.symmetrize_texture
adjust_poles = _g.adjust_poles
circpad = _g.circpad
grid_sample_bilinear = _m.grid_sample_bilinear
qrot = _m.qrot
qmul = _m.qmul
