"""mi3dgs: MI355X-native 3D Gaussian Splatting hot path (drop-in for the reference's
`Train-Stage1` subprocess, source/container/src/main.py:1270-1347).

Host side of the C-ABI in include/mi3dgs.h.  Importing this package does not load the HIP
library; the first operator call does, and raises if it is missing (no CPU fallback).
"""
from . import _lib, ops, scenes  # noqa: F401
from .ops import rasterization  # noqa: F401

__all__ = ["rasterization", "ops", "scenes", "_lib"]
