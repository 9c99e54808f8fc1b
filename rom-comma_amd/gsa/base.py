"""GSA calibrator interface (reference gsa/base.py:44-49). The reference's ``Gaussian`` helper class (gsa/base.py:52-126)
has no counterpart here: its arithmetic is fused into csrc/sobol.hip ("Sobol algebra")."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Dict

import numpy as np


class Calibrator(ABC):
    """Anything whose ``marginalize(m)`` returns a dict of results for the slice ``[m[0], m[1])`` of the input dimensions."""

    @abstractmethod
    def marginalize(self, m) -> Dict[str, np.ndarray]:
        raise NotImplementedError('This is an abstract class.')
