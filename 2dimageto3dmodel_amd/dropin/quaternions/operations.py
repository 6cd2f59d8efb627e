"""Drop-in for code/quaternions/operations.py (loss-side helpers on tiny [.,4] tensors)."""
from _m355 import projection as _p

QuaternionOperations = _p.QuaternionOperations
