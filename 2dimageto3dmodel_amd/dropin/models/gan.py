"""Drop-in for code/models/gan.py: same names, constructor arguments, forward signatures and state_dict keys; the
convolutions / norms run on the MI355X kernels (2dimageto3dmodel_amd/gan.py)."""
import importlib

from _m355 import pkg as _pkg  # noqa: F401  (puts the package on sys.path)

_g = importlib.import_module("2dimageto3dmodel_amd.gan")
Generator = _g.Generator
MultiScaleDiscriminator = _g.MultiScaleDiscriminator
TextureDiscriminator = _g.TextureDiscriminator
MeshDiscriminator = _g.MeshDiscriminator
ResBlockUp = _g.ResBlockUp
ConditionalBatchNorm2d = _g.ConditionalBatchNorm2d
SpatialAttention = _g.SpatialAttention
positional_encoding = _g.positional_encoding
