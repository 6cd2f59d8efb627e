"""Drop-in for the GAN loss of code/utils/losses.py:21-120 (loss_flat needs the mesh template: SURVEY 8f row 1)."""
import importlib

from _m355 import pkg as _pkg  # noqa: F401

GANLoss = importlib.import_module("2dimageto3dmodel_amd.gan").GANLoss
