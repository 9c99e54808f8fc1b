"""Type and constant definitions shared by the host-side mirror of the reference interface.

The reference takes its dtypes from GPflow's config, which user.contexts.Environment forces to float64 / int32
(base/definitions.py:39-46; user/contexts.py:67). This backend computes in fp64 only, so they are constants here.
"""
from __future__ import annotations

from pathlib import Path          # noqa: F401  (re-exported, as the reference's definitions module does)
from typing import *              # noqa: F401,F403

import numpy as np
import pandas as pd               # noqa: F401

EFFECTIVELY_ZERO = 1.0E-64        #: Tolerance when testing floats for equality (base/definitions.py:36).


def INT() -> type:
    """Integer dtype of slices and indices (gf.config.default_int in the reference)."""
    return np.int32


def FLOAT() -> type:
    """Floating dtype of every tensor on the path."""
    return np.float64


class NP:
    """NumPy type aliases used in signatures (documentation only)."""
    Array = Tensor = Vector = Covector = Matrix = Tensor3 = Tensor4 = np.ndarray
    VectorLike = MatrixLike = ArrayLike = TensorLike = Any      # noqa: F405
    Slice = np.ndarray            #: a pair of ints [m0, m1) selecting input dimensions
