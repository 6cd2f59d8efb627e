"""Drop-in for code/quaternions/operations.py (loss-side helpers on tiny [.,4] tensors)."""
from _m355 import projection as _p


class QuaternionOperations(object):
    def quaternion_multiplication(self, q1, q2):
        return _p.quaternion_multiplication(q1, q2)

    def quaternion_conjugate(self, q):
        return _p.quaternion_conjugate(q)
