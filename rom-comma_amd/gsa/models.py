"""GSA driver (reference gsa/models.py:35-214): builds the dimension slices of each kind, asks the calibrator for each,
post-processes (total = full - closed-of-complement), writes S.csv / V.csv with the reference's index and column labels."""
from __future__ import annotations

from abc import abstractmethod
from enum import IntEnum, auto
from typing import Any, Dict, List, NamedTuple

import numpy as np
import pandas as pd

from romcomma_amd.base.classes import Data, Frame, Model
from romcomma_amd.gpr.models import GPR
from romcomma_amd.gsa.base import Calibrator
from romcomma_amd.gsa.calibrators import ClosedSobol, ClosedSobolWithError


class GSA(Model):
    """A generic sensitivity calculation on a fitted GP."""

    class Kind(IntEnum):
        FIRST_ORDER = auto()
        CLOSED = auto()
        TOTAL = auto()

    @classmethod
    @property
    def ALL_KINDS(cls) -> List['GSA.Kind']:
        return [kind for kind in cls.Kind]

    def __init__(self, gp: GPR, kind: 'GSA.Kind', m: int = -1, is_error_calculated: bool = False, **kwargs: Any):
        """Results go to ``gp.folder / 'gsa' / <kind>[.m]``; ``m`` outside [0, M) means every m (gsa/models.py:139-160)."""
        self.gp = gp
        self.is_error_calculated = is_error_calculated
        self.kind = kind
        m = m if 0 <= m < gp.M else -1
        name = kind.name.lower() if m == -1 else f'{kind.name.lower()}.{m}'
        folder = gp.folder / 'gsa' / name
        super().__init__(folder, read_data=False)
        self.meta = {'folder': str(folder), 'm': m, 'M': gp.M} | self.META | kwargs
        self.write_meta(self.meta)

    @staticmethod
    def _columns(M: int, m_cols: int, m_list: List[int]) -> pd.Index:
        """Column labels m = 0..M-1 plus M for the appended full-model column (gsa/models.py:49-63)."""
        if m_cols > len(m_list):
            m_list = m_list + [M]
        if m_cols > len(m_list):
            m_list = [-1] + m_list
        return pd.Index(m_list, name='m')

    @staticmethod
    def _index(shape: List[int]) -> pd.MultiIndex:
        """Row MultiIndex (l.0, l.1) over the leading (L, L) axes (gsa/models.py:65-75)."""
        ranges = [list(range(n)) for n in shape[:-1]]
        return pd.MultiIndex.from_product(ranges, names=[f'l.{i}' for i in range(len(ranges))])

    @property
    def _m_dataset(self) -> List[np.ndarray]:
        """One int32 pair per marginalisation: FIRST_ORDER [m, m+1), CLOSED [0, m+1), TOTAL [m+1, M) (gsa/models.py:77-90)."""
        m, M = self.meta['m'], self.meta['M']
        ms = range(M) if m < 0 else [m]
        if self.kind == GSA.Kind.FIRST_ORDER:
            pairs = [(i, i + 1) for i in ms]
        elif self.kind == GSA.Kind.CLOSED:
            pairs = [(0, i + 1) for i in ms]
        elif self.kind == GSA.Kind.TOTAL:
            pairs = [(i + 1, M) for i in ms]
        else:
            pairs = []
        return [np.array(p, dtype=np.int32) for p in pairs]

    @property
    @abstractmethod
    def calibrator(self) -> Calibrator:
        raise NotImplementedError

    @abstractmethod
    def _post_calibrate(self, calibrator: Calibrator, results: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
        raise NotImplementedError

    def _compose_and_save(self, results: Dict[str, np.ndarray]):
        """Each result (L, L, columns) flattened to rows (l.0, l.1), written with float_format '%.6f' (gsa/models.py:102-115)."""
        m, M = self.meta['m'], self.meta['M']
        m_list = list(range(M)) if m < 0 else [m]
        for key, frame in self.data.asdict().items():
            result = results.get(key, None)
            if result is not None:
                shape = list(result.shape)
                table = pd.DataFrame(np.reshape(result, (-1, shape[-1])), columns=GSA._columns(M, shape[-1], m_list), index=GSA._index(shape))
                Frame(frame.csv, table, float_format='%.6f')

    def calibrate(self, method: str = None, **kwargs) -> Dict[str, Any]:
        """Marginalise every slice, stack on a new last axis, post-process, save (gsa/models.py:117-137)."""
        calibrator = self.calibrator
        results: Dict[str, np.ndarray] = {}
        for m in self._m_dataset:
            for key, value in calibrator.marginalize(m).items():
                value = np.asarray(value)[..., None]
                results[key] = value if key not in results else np.concatenate([results[key], value], axis=-1)
        results = self._post_calibrate(calibrator, results)
        self.results = results          # in-memory values (the csv files are rounded to 6 decimals)
        self._compose_and_save(results)
        return self.meta


class Sobol(GSA):
    """Sobol indices S (and conditional variances V) of one kind."""

    class Data(Data):
        class NamedTuple(NamedTuple):
            S: Any = np.atleast_2d(None)       # the Sobol index
            T: Any = np.atleast_2d(None)       # its standard deviation (with errors only)
            V: Any = np.atleast_2d(None)       # the conditional variances behind S
            W: Any = np.atleast_2d(None)       # the covariances behind T (with errors only)

    @classmethod
    @property
    def META(cls) -> Dict[str, Any]:
        return ClosedSobolWithError.META

    @property
    def calibrator(self) -> ClosedSobol:
        meta = {k: v for k, v in self.meta.items()}
        if self.is_error_calculated:
            return ClosedSobolWithError(self.gp, **meta)
        # The three kinds of one gp share every conditional variance: reuse the calibrator while the hyper-parameters stand.
        if hasattr(self.gp, 'hyper_signature'):            # an OutputShard (outputs on different ranks)
            signature = self.gp.hyper_signature()
        else:
            signature = tuple(np.concatenate([np.ravel(self.gp.kernel.data.frames.lengthscales.np), np.ravel(self.gp.kernel.data.frames.variance.np),
                                              np.ravel(self.gp.likelihood.data.frames.variance.np)]))
        cached = getattr(self.gp, '_closed_sobol', None)
        if cached is None or cached[0] != signature:
            cached = (signature, ClosedSobol(self.gp, **meta))
            self.gp._closed_sobol = cached
        return cached[1]

    def _post_calibrate(self, calibrator: ClosedSobol, results: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
        """Append the full-model column; TOTAL index = S_full - S_closed(complement) (gsa/models.py:207-214)."""
        results['V'] = np.concatenate([results['V'], calibrator.V[0][..., None]], axis=-1)
        if self.kind == GSA.Kind.TOTAL:
            results['S'] = calibrator.S[..., None] - results['S']
        results['S'] = np.concatenate([results['S'], calibrator.S[..., None]], axis=-1)
        if 'T' in results and not self.meta['is_T_partial']:
            if self.kind == GSA.Kind.TOTAL:
                results['T'] = calibrator.T[..., None] + results['T']
            results['T'] = np.concatenate([results['T'], calibrator.T[..., None]], axis=-1)
        return results
