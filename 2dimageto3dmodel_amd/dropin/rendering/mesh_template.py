"""Drop-in for code/rendering/mesh_template.py (no Kaolin): MeshTemplate with the per-step methods on libm355
(2dimageto3dmodel_amd/mesh.py; `forward_renderer` needs the DIB-R rasteriser, SURVEY 8f row 2, and is not provided)."""
import importlib

from _m355 import pkg as _pkg  # noqa: F401

MeshTemplate = importlib.import_module("2dimageto3dmodel_amd.mesh").MeshTemplate
