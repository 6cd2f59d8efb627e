"""Drop-in for code/quaternions/points_quaternions.py."""
from _m355 import projection as _p

PointsQuaternionsConverter = _p.PointsQuaternionsConverter
PointsQuaternionsRotator = _p.PointsQuaternionsRotator
