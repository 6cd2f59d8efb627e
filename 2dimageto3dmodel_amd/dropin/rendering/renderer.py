"""Drop-in for code/rendering/renderer.py (the Kaolin DIB-R rasteriser it imports is replaced by libm355)."""
import importlib

from _m355 import pkg as _pkg  # noqa: F401

_r = importlib.import_module("2dimageto3dmodel_amd.render")
Renderer = _r.Renderer
ortho_projection = _r.ortho_projection
linear_rasterizer = _r.linear_rasterizer
datanormalize = _r.datanormalize
