"""Drop-in for code/models/reconstruction.py: ResBlock / ReconstructionNetwork / DatasetParams on libm355."""
import importlib

from _m355 import pkg as _pkg  # noqa: F401

_r = importlib.import_module("2dimageto3dmodel_amd.reconstruction")
ResBlock = _r.ResBlock
ReconstructionNetwork = _r.ReconstructionNetwork
DatasetParams = _r.DatasetParams
