"""Drop-in for the tensor helpers of code/rendering/utils.py:15-33 used by the GAN path."""
import importlib

from _m355 import pkg as _pkg  # noqa: F401

_g = importlib.import_module("2dimageto3dmodel_amd.gan")
symmetrize_texture = _g.symmetrize_texture
adjust_poles = _g.adjust_poles
circpad = _g.circpad
