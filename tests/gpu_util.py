"""Helpers for the -m gpu parity tests: torch is only device memory here."""
import numpy as np
import torch


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def ptr(t):
    return t.data_ptr()


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
