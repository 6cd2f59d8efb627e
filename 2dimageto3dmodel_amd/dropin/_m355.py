"""Locates the 2dimageto3dmodel_amd package (its directory name starts with a digit) for the drop-in shims."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.append(_ROOT)
pkg = importlib.import_module("2dimageto3dmodel_amd")
projection = importlib.import_module("2dimageto3dmodel_amd.projection")
