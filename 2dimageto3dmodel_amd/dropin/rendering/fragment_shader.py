"""Drop-in for code/rendering/fragment_shader.py."""
import importlib

from _m355 import pkg as _pkg  # noqa: F401

_r = importlib.import_module("2dimageto3dmodel_amd.render")
fragmentshader = _r.fragmentshader
texinterpolation = _r.texinterpolation
