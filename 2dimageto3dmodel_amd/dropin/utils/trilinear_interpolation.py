"""Drop-in for code/utils/trilinear_interpolation.py."""
from _m355 import projection as _p

TrilinearInterpolation = _p.TrilinearInterpolation
