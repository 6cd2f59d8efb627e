"""Drop-in for code/rendering/mesh_template.py (no Kaolin): MeshTemplate with the per-step methods on libm355
(2dimageto3dmodel_amd/mesh.py; `forward_renderer` drives the DIB-R rasteriser of 2dimageto3dmodel_amd/render.py)."""
import importlib

from _m355 import pkg as _pkg  # noqa: F401

MeshTemplate = importlib.import_module("2dimageto3dmodel_amd.mesh").MeshTemplate
