"""Dataset store: Repository (data.csv + meta.json), Fold (adds test.csv, normalization.csv, X_rotation.csv) and
Normalization -- the "Fold" half of the plugin API that ``GPR.__init__`` consumes (reference data/storage.py:39-558).

File formats are the reference's: ``data.csv`` / ``test.csv`` carry a two-row header (level 0 = the X and Y group names,
level 1 = column names) and an index column; ``meta.json`` uses indent 8; ``normalization.csv`` has rows
mean, std, rng, min, max. One deliberate addition: ``into_K_folds(..., seed=...)`` makes the fold assignment
reproducible (the reference shuffles with the unseeded global ``random``: data/storage.py:184,195).
"""
from __future__ import annotations

import itertools
import json
import random
import shutil
from copy import deepcopy
from enum import IntEnum, auto
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import pandas as pd
import scipy.stats


class Frame:
    """A DataFrame with the two-row column header, backed by a csv file given WITH its suffix (data/storage.py:39-88)."""

    CSV_OPTIONS: Dict[str, Any] = {'sep': ',', 'header': [0, 1], 'index_col': 0}

    def __init__(self, csv: Path | str = Path(), df: pd.DataFrame | None = None, **kwargs: Any):
        """An empty ``df`` means read ``csv``; otherwise ``df`` is stored and written. ``kwargs`` update the read options."""
        self._csv = Path(csv)
        df = pd.DataFrame() if df is None else df
        if self.is_empty:
            assert df.empty, 'csv is an empty path, but df is not an empty pd.DataFrame.'
            self.df = df
        elif df.empty:
            self.df = pd.read_csv(self._csv, **{**Frame.CSV_OPTIONS, **kwargs})
        else:
            self.df = df
            self.write()

    @property
    def csv(self) -> Path:
        return self._csv

    @property
    def is_empty(self) -> bool:
        return len(self._csv.parts) == 0

    def write(self):
        assert not self.is_empty, 'Cannot write when frame.is_empty.'
        self.df.to_csv(path_or_buf=self._csv, sep=Frame.CSV_OPTIONS['sep'], index=True)

    def __repr__(self) -> str:
        return str(self._csv)

    def __str__(self) -> str:
        return self._csv.name


class Repository:
    """A folder holding ``data.csv`` and ``meta.json``: the global dataset, to be split into Folds (data/storage.py:91-343)."""

    class _InitMode(IntEnum):
        READ_META_ONLY = auto()
        READ = auto()
        CREATE = auto()

    META: Dict[str, Any] = {'csv_kwargs': Frame.CSV_OPTIONS, 'data': {}, 'K': 0, 'shuffle before folding': False}
    CSV_OPTIONS: Dict[str, Any] = {'skiprows': None, 'index_col': 0}

    def __init__(self, folder: Path | str, **kwargs: Any):
        self._folder = Path(folder)
        self._meta_json = self._folder / 'meta.json'
        self._csv = self._folder / 'data.csv'
        self._data = None
        mode = Repository._InitMode(kwargs.get('init_mode', Repository._InitMode.READ))
        if mode is Repository._InitMode.CREATE:              # a fresh, empty folder; the caller supplies meta and data next
            if self._folder.exists():
                shutil.rmtree(self._folder)
            self._folder.mkdir(mode=0o777, parents=True)
            return
        self._meta = self.read_meta()                        # READ and READ_META_ONLY both need meta.json ...
        if mode is Repository._InitMode.READ:
            self._data = Frame(self._csv)                    # ... only READ loads data.csv

    # ---- plain accessors
    @property
    def folder(self) -> Path:
        return self._folder

    @property
    def data(self) -> Frame:
        return self._data

    @property
    def meta(self) -> Dict[str, Any]:
        return self._meta

    @property
    def X(self) -> pd.DataFrame:
        """Inputs, an (N, M) frame selected by the level-0 heading (data/storage.py:105-108)."""
        return self._data.df[self._meta['data']['X_heading']]

    @property
    def Y(self) -> pd.DataFrame:
        return self._data.df[self._meta['data']['Y_heading']]

    @property
    def N(self) -> int:
        return self._meta['data']['N']

    @property
    def M(self) -> int:
        return self._meta['data']['M']

    @property
    def L(self) -> int:
        return self._meta['data']['L']

    @property
    def K(self) -> int:
        return self._meta['K']

    @property
    def folds(self) -> range:
        """Fold indices; a positive K given to ``into_K_folds`` also created the improper fold K (data/storage.py:154-160)."""
        if isinstance(self, Fold) or self.K < 1:
            return range(0, 0)
        return range(self.K + (1 if self.meta['has_improper_fold'] else 0))

    def read_meta(self) -> Dict[str, Any]:
        with open(self._meta_json, mode='r') as file:
            return json.load(file)

    def write_meta(self):
        with open(self._meta_json, mode='w') as file:
            json.dump(self._meta, file, indent=8)

    def _update_meta(self):
        columns = self._data.df.columns.values
        self._meta.update({'data': {'X_heading': columns[0][0], 'Y_heading': columns[-1][0]}})
        self._meta['data'].update({'N': self.data.df.shape[0], 'M': self.X.shape[1], 'L': self.Y.shape[1]})
        self.write_meta()

    def fold_folder(self, k: int) -> Path:
        return self._folder / f'fold.{k:d}'

    # ---- folding
    def into_K_folds(self, K: int, shuffle_before_folding: bool = False, normalization: Optional[Path | str] = None,
                     is_normalization_applicable: bool = True, seed: Optional[int] = None) -> 'Repository':
        """Split into |K| folds indexed by range(|K|); K > 0 adds the improper fold K trained and tested on everything
        (data/storage.py:162-204). Assignment: rows are dealt in blocks of |K|, each block a random permutation of the fold
        indices, so fold sizes differ by at most one. ``seed`` (an addition) makes that permutation reproducible."""
        rng = random.Random(seed) if seed is not None else random
        data = self.data.df
        N = data.shape[0]
        if not (1 <= abs(K) <= N):
            raise IndexError(f'K={K:d} does not lie between 1 and N={N:d} inclusive.')
        for k in range(max(abs(K), self.K) + 1):
            shutil.rmtree(self.fold_folder(k), ignore_errors=True)
        index = list(range(N))
        if shuffle_before_folding:
            rng.shuffle(index)
        self._meta.update({'K': abs(K), 'shuffle before folding': shuffle_before_folding, 'has_improper_fold': K > 0})
        self.write_meta()
        normalization = Normalization(self, self._data.df).csv if normalization is None else normalization
        if K > 0:
            Fold.from_dfs(parent=self, k=K, data=data.iloc[index], test_data=data.iloc[index], normalization=normalization,
                          is_normalization_applicable=is_normalization_applicable)
        K = abs(K)
        blocks = [list(range(K)) for _ in range(N // K)] + [list(range(N % K))]
        for block in blocks:
            rng.shuffle(block)
        indicator = list(itertools.chain(*blocks))
        for k in range(K):
            train = [i for i, which in zip(index, indicator) if which != k]
            test = [i for i, which in zip(index, indicator) if which == k]
            train = test if not train else train
            Fold.from_dfs(parent=self, k=k, data=data.iloc[train], test_data=data.iloc[test], normalization=normalization,
                          is_normalization_applicable=is_normalization_applicable)
        return self

    def rotate_folds(self, rotation: np.ndarray | None) -> 'Repository':
        """Apply one (M, M) rotation to the inputs of every fold; None = identity; a malformed matrix is replaced by a
        random rotation (data/storage.py:206-221)."""
        M = self.M
        if rotation is None:
            rotation = np.eye(M)
        elif rotation.shape != (M, M) or not np.allclose(np.dot(rotation, rotation.T), np.eye(M)):
            rotation = scipy.stats.special_ortho_group.rvs(M)
        for k in self.folds:
            Fold(self, k).X_rotation = rotation
        return self

    def Y_split(self):
        """One single-output Repository ``Y.l`` per output column (data/storage.py:226-243)."""
        if isinstance(self, Fold):
            raise TypeError('Cannot Y_split a Fold, only a Repository.')
        M, frame = self.M, self.data.df
        for l in range(self.L):
            one_output = pd.concat([frame.iloc[:, :M], frame.iloc[:, [M + l]]], axis=1)       # all inputs + output column l
            split_meta = deepcopy(self._meta)
            split_meta['data'] = {**split_meta['data'], 'L': 1}
            Repository.from_df(self.folder / f'Y.{l:d}', one_output, split_meta)

    @property
    def Y_splits(self) -> List[Tuple[int, Path]]:
        return [(int(path.suffix[1:]), path) for path in self.folder.glob('Y.[0-9]*')]

    # ---- construction
    @classmethod
    def from_df(cls, folder: Path | str, df: pd.DataFrame, meta: Dict | None = None) -> 'Repository':
        repo = Repository(folder, init_mode=Repository._InitMode.CREATE)
        repo._meta = deepcopy(cls.META) | ({} if meta is None else meta)
        repo._data = Frame(repo._csv, df)
        repo._update_meta()
        return repo

    @classmethod
    def from_csv(cls, folder: Path | str, csv: Path | str, meta: Dict | None = None, **kwargs: Any) -> 'Repository':
        """Create from a csv with the two-row header (data/storage.py:302-320). The reference's PCA side-path is not on the
        hot path and is not provided."""
        csv = Path(csv)
        origin_csv_kwargs = cls.CSV_OPTIONS | kwargs
        data = Frame(csv, **origin_csv_kwargs)
        meta = deepcopy(cls.META) | ({} if meta is None else meta)
        meta['origin'] = {'csv': str(csv.absolute()), 'origin_csv_kwargs': origin_csv_kwargs}
        return cls.from_df(folder, data.df, meta)

    def __repr__(self) -> str:
        return str(self._folder)

    def __str__(self) -> str:
        return self._folder.name


class Fold(Repository):
    """A Repository with held-out ``test.csv`` and its Normalization (data/storage.py:346-437)."""

    def __init__(self, parent: Repository, k: int, **kwargs: Any):
        init_mode = kwargs.get('init_mode', Repository._InitMode.READ)
        super().__init__(parent.fold_folder(k), init_mode=init_mode)
        self._X_rotation = self.folder / 'X_rotation.csv'
        self._test_csv = self.folder / 'test.csv'
        if init_mode == Repository._InitMode.READ:
            self._test_data = Frame(self._test_csv)
            self._normalization = Normalization(self)

    @property
    def normalization(self) -> 'Normalization':
        return self._normalization

    @property
    def test_csv(self) -> Path:
        return self._test_csv

    @property
    def test_data(self) -> Frame:
        return self._test_data

    @property
    def test_x(self) -> pd.DataFrame:
        return self._test_data.df[self._meta['data']['X_heading']]

    @property
    def test_y(self) -> pd.DataFrame:
        return self._test_data.df[self._meta['data']['Y_heading']]

    def _X_rotate(self, frame: Frame, rotation: np.ndarray):
        frame.df.iloc[:, :self.M] = np.einsum('Nm,Mm->NM', frame.df.iloc[:, :self.M], rotation)
        frame.write()

    @property
    def X_rotation(self) -> np.ndarray:
        """Cumulative rotation applied to the inputs, stored in X_rotation.csv (data/storage.py:385-396)."""
        return Frame(self._X_rotation, header=[0]).df.values if self._X_rotation.exists() else np.eye(self.M)

    @X_rotation.setter
    def X_rotation(self, value: np.ndarray):
        self._X_rotate(self._data, value)
        self._X_rotate(self._test_data, value)
        Frame(self._X_rotation, pd.DataFrame(np.matmul(self.X_rotation, value)))

    @classmethod
    def from_dfs(cls, parent: Repository, k: int, data: pd.DataFrame, test_data: pd.DataFrame, normalization: Optional[Path | str] = None,
                 is_normalization_applicable: bool = True) -> 'Fold':
        fold = cls(parent, k, init_mode=Repository._InitMode.CREATE)
        fold._meta = deepcopy(cls.META) | deepcopy(parent.meta) | {'k': k}
        fold._normalization = Normalization(fold, data, is_normalization_applicable)
        if normalization is not None:
            shutil.copy(Path(normalization), fold._normalization.csv)
            fold._normalization._frame = None
        fold._data = Frame(fold._csv, fold.normalization.apply_to(data))
        fold._test_data = Frame(fold._test_csv, fold.normalization.apply_to(test_data))
        fold._update_meta()
        return fold


class Normalization:
    """X assumed uniform: mapped to U[0,1] with min = mean - sqrt(3) std, rng = 2 sqrt(3) std, clipped to [1e-12, 1-1e-12], then
    probit-transformed to N(0,1). Y z-scored. pandas ``std`` (ddof = 1) throughout (data/storage.py:440-558)."""

    UNIFORM_MARGIN: float = 1.0E-12

    def __init__(self, fold: Repository, data: Optional[pd.DataFrame] = None, is_applicable: bool = True):
        self._fold = fold
        self._is_applicable = is_applicable
        if self.csv.exists():
            self._frame = Frame(self.csv)
        elif data is None:
            self._frame = None
        else:
            mean, std = data.mean(), data.std()
            semi_range = std * np.sqrt(3)
            stats = pd.concat((mean.rename('mean'), std.rename('std'), (2 * semi_range).rename('rng'), (mean - semi_range).rename('min'),
                               (mean + semi_range).rename('max')), axis=1)
            self._frame = Frame(self.csv, stats.T)

    @property
    def csv(self) -> Path:
        return self._fold.folder / 'normalization.csv'

    @property
    def frame(self) -> Frame:
        if self._frame is None:
            self._frame = Frame(self.csv)
        return self._frame

    @property
    def is_applicable(self) -> bool:
        return self._is_applicable

    @property
    def _relevant_stats(self) -> Tuple[pd.Series, pd.Series, pd.Series, pd.Series]:
        df, M = self.frame.df, self._fold.M
        return df.loc['min'].iloc[:M], df.loc['rng'].iloc[:M], df.loc['mean'].iloc[M:], df.loc['std'].iloc[M:]

    def apply_to(self, df: pd.DataFrame) -> pd.DataFrame:
        if not self._is_applicable:
            return df
        X_min, X_rng, Y_mean, Y_std = self._relevant_stats
        M = self._fold.M
        X = df.iloc[:, :M].copy(deep=True)
        Y = df.iloc[:, M:].copy(deep=True)
        X = X.sub(X_min, axis=1).div(X_rng, axis=1).clip(lower=self.UNIFORM_MARGIN, upper=1 - self.UNIFORM_MARGIN)
        X.iloc[:, :] = scipy.stats.norm.ppf(X, loc=0, scale=1)
        Y = Y.sub(Y_mean, axis=1).div(Y_std, axis=1)
        return pd.concat((X, Y), axis=1)

    def undo_from(self, df: pd.DataFrame) -> pd.DataFrame:
        if not self._is_applicable:
            return df
        X_min, X_rng, Y_mean, Y_std = self._relevant_stats
        M = self._fold.M
        X = df.iloc[:, :M].copy(deep=True)
        Y = df.iloc[:, M:].copy(deep=True)
        X.iloc[:, :] = scipy.stats.norm.cdf(X, loc=0, scale=1)
        X = X.mul(X_rng, axis=1).add(X_min, axis=1)
        Y = Y.mul(Y_std, axis=1).add(Y_mean, axis=1)
        return pd.concat((X, Y), axis=1)

    def unscale_Y(self, dfY: pd.DataFrame) -> pd.DataFrame:
        """Undo the Y scaling without adding the mean back (for standard deviations) (data/storage.py:505-513)."""
        if not self._is_applicable:
            return dfY
        Y_std = self._relevant_stats[3]
        return dfY.copy(deep=True).mul(Y_std, axis=1)

    def X_gradient(self, X: np.ndarray, m: int | List[int]):
        """d(unnormalised X[m]) / d(normalised Z[m]) (data/storage.py:515-524)."""
        X_rng = self._relevant_stats[1].values[m]
        return X_rng * scipy.stats.norm.pdf(X[..., m], loc=0, scale=1) if self._is_applicable else np.ones_like(X[..., m])

    def __repr__(self) -> str:
        return str(self.csv)

    def __str__(self) -> str:
        return self.csv.name
