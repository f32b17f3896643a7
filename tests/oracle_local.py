"""Oracle-backed stand-in for the per-GPU projector, used only by the CPU (gloo) tests of
the sharded frame sequence.  Same duck type as rtr_amd.sharded.HipLocal."""
import numpy as np
import torch


class OracleLocal:
    def __init__(self, orc, xyzw, rgba, W, H):
        self.orc, self.xyzw, self.rgba, self.W, self.H = orc, xyzw, rgba, W, H
        self.depth = np.empty(W * H, np.uint32)
        self.acc = np.empty(W * H * 4, np.uint32)
        self.img = None
        self.filtered = None

    def depth_tensor(self):
        return torch.from_numpy(self.depth.view(np.int32))

    def accum_tensor(self):
        return torch.from_numpy(self.acc.view(np.int32))

    def clear(self):
        self.depth[:] = self.orc.EMPTY_DEPTH
        self.acc[:] = 0

    def min_depth_pass(self, P):
        self.orc.min_depth_pass(self.xyzw, P, self.W, self.H, self.depth)

    def accumulate_pass(self, P):
        self.orc.accumulate_pass(self.xyzw, self.rgba, P, self.W, self.H, self.depth, self.acc)

    def resolve(self):
        self.img = self.orc.resolve(self.acc, self.W, self.H)

    def filter(self):
        self.filtered = self.orc.filter(self.depth.reshape(self.H, self.W), self.img)
