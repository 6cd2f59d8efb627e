"""Drop-in for the loss of code/models/unsupervised_part.py."""
from _m355 import projection as _p

UnsupervisedLoss = _p.UnsupervisedLoss
