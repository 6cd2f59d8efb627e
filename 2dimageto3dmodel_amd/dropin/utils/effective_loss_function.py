"""Drop-in for code/utils/effective_loss_function.py (same import path with dropin/ on sys.path)."""
from _m355 import projection as _p

EffectiveLossFunction = _p.EffectiveLossFunction
