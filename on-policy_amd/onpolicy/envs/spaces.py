"""Minimal gym-style space types.  The buffer / networks dispatch on the class NAME ('Box',
'Discrete', ...; reference onpolicy/utils/util.py:31-52), so these are interchangeable with
gym / gymnasium spaces, which are not installed in the build image."""
import numpy as np


class Box(object):
    def __init__(self, low=-np.inf, high=np.inf, shape=None, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    def __repr__(self):
        return "Box%s" % (self.shape,)


class Discrete(object):
    def __init__(self, n):
        self.n = int(n)

    def __repr__(self):
        return "Discrete(%d)" % self.n
