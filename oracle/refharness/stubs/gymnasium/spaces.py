"""Minimal `gymnasium.spaces` stand-in (oracle harness only): records constructor arguments."""
import numpy as np


class Space:
    pass


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        if shape is None:
            shape = np.shape(low)
        self.shape = tuple(shape)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()


class Discrete(Space):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)


class Dict(Space):
    def __init__(self, spaces):
        self.spaces = dict(sorted(spaces.items()))   # Gymnasium sorts Dict keys

    def keys(self):
        return self.spaces.keys()

    def __getitem__(self, k):
        return self.spaces[k]
