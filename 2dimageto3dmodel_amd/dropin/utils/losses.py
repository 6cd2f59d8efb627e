"""Drop-in for code/utils/losses.py: GANLoss (:21-120) and the mesh smoothness regulariser loss_flat (:5-17)."""
import importlib

from _m355 import pkg as _pkg  # noqa: F401

GANLoss = importlib.import_module("2dimageto3dmodel_amd.gan").GANLoss
loss_flat = importlib.import_module("2dimageto3dmodel_amd.mesh").loss_flat
