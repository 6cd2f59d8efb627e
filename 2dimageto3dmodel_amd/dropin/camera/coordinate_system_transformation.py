"""Drop-in for code/camera/coordinate_system_transformation.py."""
from _m355 import projection as _p

CameraUtilities = _p.CameraUtilities
