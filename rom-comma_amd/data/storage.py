"""On-disk dataset store behind ``GPR.__init__``: ``Repository`` (the whole dataset), ``Fold`` (one train / test split of it, already
normalised) and ``Normalization`` (the per-fold statistics and the two maps built from them).

Written against the store CONTRACT of the reference (interface: data/storage.py:39-558; layout: SURVEY.md Appendix D), not its text:

    <repo>/data.csv            two header rows (group 'X' | 'Y', then the column name), row index in column 0
    <repo>/meta.json           {csv_kwargs, data: {X_heading, Y_heading, N, M, L}, K, 'shuffle before folding', has_improper_fold[, origin]}, indent 8
    <repo>/fold.<k>/data.csv   normalised training rows      <repo>/fold.<k>/test.csv   normalised held-out rows
    <repo>/fold.<k>/meta.json  the parent's meta + {'k': k}, data.N = training rows of this fold
    <repo>/fold.<k>/normalization.csv   rows mean, std, rng, min, max over every X and Y column (pandas std, ddof = 1)
    <repo>/fold.<k>/X_rotation.csv      cumulative (M, M) input rotation, single header row
    <repo>/Y.<l>/              single-output copies made by Y_split

Own organisation: the fold assignment is one vectorised NumPy expression (``_deal``), the statistics live in a small frozen dataclass
(``_Moments``), json files are written through a temporary file and renamed (a reader on another rank never sees half a file), and the
split is reproducible when ``seed`` is given -- the reference shuffles with the unseeded global ``random`` (data/storage.py:184,195).
"""

import copy
import json
import os
import shutil
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Optional

import numpy as np
import pandas as pd
from scipy.special import ndtr, ndtri
from scipy.stats import special_ortho_group

_TWO_ROW_HEADER: dict[str, Any] = {'sep': ',', 'header': [0, 1], 'index_col': 0}
_STAT_ROWS = ('mean', 'std', 'rng', 'min', 'max')


def _load_json(path: Path) -> dict[str, Any]:
    return json.loads(path.read_text())


def _dump_json(path: Path, content: dict[str, Any]):
    """meta.json with the reference's indent, replaced in one step: ranks of a sharded run read each other's folders."""
    scratch = path.with_name(f'.{path.name}.{os.getpid()}.tmp')
    scratch.write_text(json.dumps(content, indent=8))
    os.replace(scratch, path)


class Frame:
    """A ``pd.DataFrame`` tied to the csv file it is stored in (path given WITH its suffix; data/storage.py:39-88).
    ``Frame(csv)`` reads, ``Frame(csv, df)`` stores ``df`` and writes it, ``Frame()`` is the empty placeholder."""

    CSV_OPTIONS: dict[str, Any] = dict(_TWO_ROW_HEADER)

    def __init__(self, csv: Path | str = Path(), df: Optional[pd.DataFrame] = None, **kwargs: Any):
        self._csv = Path(csv)
        has_content = df is not None and not df.empty
        if self.is_empty:
            if has_content:
                raise AssertionError('a Frame without a csv path cannot hold a DataFrame')
            self.df = pd.DataFrame()
        elif has_content:
            self.df = df
            self.write()
        else:
            self.df = pd.read_csv(self._csv, **{**self.CSV_OPTIONS, **kwargs})

    @property
    def csv(self) -> Path:
        return self._csv

    @property
    def is_empty(self) -> bool:
        return self._csv == Path()

    def write(self):
        if self.is_empty:
            raise AssertionError('a Frame without a csv path cannot be written')
        self.df.to_csv(self._csv, sep=self.CSV_OPTIONS['sep'], index=True)

    def __repr__(self) -> str:
        return str(self._csv)

    def __str__(self) -> str:
        return self._csv.name


def _deal(N: int, K: int, rng: np.random.Generator) -> np.ndarray:
    """Fold label of every position 0..N-1: positions are dealt K at a time, each hand an independent random permutation of the K
    labels, the short last hand a permutation of the first N mod K labels -- so fold sizes differ by at most one and fold k < N mod K
    is the larger kind (the dealing rule of data/storage.py:191-198)."""
    hands, left_over = divmod(N, K)
    full = rng.permuted(np.tile(np.arange(K), (hands, 1)), axis=1).ravel()
    return np.concatenate([full, rng.permutation(left_over)]).astype(int)


class Repository:
    """A folder with ``data.csv`` + ``meta.json``: the global dataset from which Folds are cut (data/storage.py:91-343)."""

    META: dict[str, Any] = {'csv_kwargs': dict(_TWO_ROW_HEADER), 'data': {}, 'K': 0, 'shuffle before folding': False}
    CSV_OPTIONS: dict[str, Any] = {'skiprows': None, 'index_col': 0}

    def __init__(self, folder: Path | str, **kwargs: Any):
        """Open an existing repository. ``meta_only=True`` skips ``data.csv``."""
        self._root = Path(folder)
        self._meta: dict[str, Any] = _load_json(self._meta_path)
        self._table: Optional[Frame] = None if kwargs.get('meta_only', False) else Frame(self._table_path)

    @classmethod
    def _blank(cls, folder: Path, *args: Any) -> 'Repository':
        """An instance on a freshly emptied folder, nothing read: the from_* constructors fill it."""
        self = object.__new__(cls)
        self._root = Path(folder)
        if self._root.exists():
            shutil.rmtree(self._root)
        self._root.mkdir(parents=True)
        self._meta, self._table = {}, None
        return self

    # ---- where things are
    @property
    def folder(self) -> Path:
        return self._root

    @property
    def _meta_path(self) -> Path:
        return self._root / 'meta.json'

    @property
    def _table_path(self) -> Path:
        return self._root / 'data.csv'

    def fold_folder(self, k: int) -> Path:
        return Path(self._root, f'fold.{int(k)}')

    # ---- what is in it
    @property
    def data(self) -> Frame:
        return self._table

    @property
    def meta(self) -> dict[str, Any]:
        return self._meta

    def _group(self, frame: Frame, which: str) -> pd.DataFrame:
        return frame.df[self._meta['data'][which]]

    @property
    def X(self) -> pd.DataFrame:
        """The (N, M) inputs: every column under the first header group."""
        return self._group(self._table, 'X_heading')

    @property
    def Y(self) -> pd.DataFrame:
        """The (N, L) outputs: every column under the last header group."""
        return self._group(self._table, 'Y_heading')

    N = property(lambda self: self._meta['data']['N'])
    M = property(lambda self: self._meta['data']['M'])
    L = property(lambda self: self._meta['data']['L'])
    K = property(lambda self: self._meta['K'])

    @property
    def folds(self) -> range:
        """Indices of the folds on disk: 0..K-1, and K itself when ``into_K_folds`` was given a positive K (the improper fold, trained
        and tested on every row). A Fold has none."""
        if self.K < 1 or isinstance(self, Fold):
            return range(0)
        return range(self.K + int(bool(self._meta.get('has_improper_fold', False))))

    def read_meta(self) -> dict[str, Any]:
        return _load_json(self._meta_path)

    def write_meta(self):
        _dump_json(self._meta_path, self._meta)

    def _describe(self):
        """Record headings and shape of the table in meta['data'] and save."""
        groups = self._table.df.columns.get_level_values(0)
        self._meta['data'] = {'X_heading': groups[0], 'Y_heading': groups[-1]}
        self._meta['data'].update(N=len(self._table.df), M=self.X.shape[1], L=self.Y.shape[1])
        self.write_meta()

    # ---- folding
    def into_K_folds(self, K: int, shuffle_before_folding: bool = False, normalization: Optional[Path | str] = None,
                     is_normalization_applicable: bool = True, seed: Optional[int] = None) -> 'Repository':
        """Cut |K| folds ``fold.0 .. fold.|K|-1``, each holding out the rows dealt to it (``_deal``) and training on the rest; a positive
        K also writes the improper ``fold.K`` that trains and tests on everything (data/storage.py:162-204). Every fold is normalised
        with the statistics of the WHOLE repository unless a ``normalization`` csv is supplied. ``seed`` (not in the reference) fixes
        the deal and the optional pre-shuffle."""
        table, n_folds = self._table.df, abs(K)
        N = len(table)
        if not 1 <= n_folds <= N:
            raise IndexError(f'cannot cut {n_folds:d} folds from N={N:d} rows (need 1 <= |K| <= N, got K={K:d})')
        for stale in range(max(n_folds, self.K) + 1):
            shutil.rmtree(self.fold_folder(stale), ignore_errors=True)
        rng = np.random.default_rng(seed)
        order = rng.permutation(N) if shuffle_before_folding else np.arange(N)
        self._meta.update({'K': n_folds, 'shuffle before folding': shuffle_before_folding, 'has_improper_fold': K > 0})
        self.write_meta()
        stats_csv = Normalization(self, table).csv if normalization is None else Path(normalization)
        cut = dict(parent=self, normalization=stats_csv, is_normalization_applicable=is_normalization_applicable)
        if K > 0:
            Fold.from_dfs(k=K, data=table.iloc[order], test_data=table.iloc[order], **cut)
        label = _deal(N, n_folds, rng)
        for k in range(n_folds):
            held_out = order[label == k]
            kept = order[label != k] if n_folds > 1 else held_out          # a single fold trains on what it tests on
            Fold.from_dfs(k=k, data=table.iloc[kept], test_data=table.iloc[held_out], **cut)
        return self

    def rotate_folds(self, rotation: Optional[np.ndarray]) -> 'Repository':
        """Rotate the inputs of every fold by one (M, M) matrix: None = identity, anything that is not an (M, M) orthogonal matrix =
        a random rotation drawn here (data/storage.py:206-221)."""
        M = self.M
        if rotation is None:
            rotation = np.eye(M)
        elif np.shape(rotation) != (M, M) or not np.allclose(rotation @ rotation.T, np.eye(M)):
            rotation = special_ortho_group.rvs(M)
        for k in self.folds:
            setattr(Fold(self, k), 'X_rotation', rotation)
        return self

    def Y_split(self):
        """Write one single-output repository ``Y.<l>`` per output column: all inputs + column l (data/storage.py:226-243)."""
        if type(self) is not Repository:
            raise TypeError('Y_split works on a Repository; this is a Fold')
        table, M = self._table.df, self.M
        for l in range(self.L):
            columns = list(range(M)) + [M + l]
            meta = copy.deepcopy(self._meta)
            meta['data']['L'] = 1
            Repository.from_df(self._root / f'Y.{l:d}', table.iloc[:, columns], meta)

    @property
    def Y_splits(self) -> list[tuple[int, Path]]:
        """(l, folder) of every ``Y.<l>`` present."""
        return [(int(path.name.split('.')[1]), path) for path in self._root.glob('Y.[0-9]*')]

    # ---- construction
    @classmethod
    def from_df(cls, folder: Path | str, df: pd.DataFrame, meta: Optional[dict[str, Any]] = None) -> 'Repository':
        """A new repository (the folder is emptied first) holding ``df``, whose columns carry the two-level header."""
        repo = Repository._blank(Path(folder))
        repo._meta = {**copy.deepcopy(cls.META), **(meta or {})}
        repo._table = Frame(repo._table_path, df)
        repo._describe()
        return repo

    @classmethod
    def from_csv(cls, folder: Path | str, csv: Path | str, meta: Optional[dict[str, Any]] = None, **kwargs: Any) -> 'Repository':
        """A new repository from a csv that already has the two header rows; where it came from is kept under meta['origin']
        (data/storage.py:302-320). The reference's PCA side-path is outside the hot path and not provided."""
        source = Path(csv)
        how = {**cls.CSV_OPTIONS, **kwargs}
        origin = {'origin': {'csv': str(source.absolute()), 'origin_csv_kwargs': how}}
        return cls.from_df(folder, Frame(source, **how).df, {**(meta or {}), **origin})

    def __repr__(self) -> str:
        return str(self._root)

    def __str__(self) -> str:
        return self._root.name


class Fold(Repository):
    """``fold.<k>`` of a parent repository: normalised training rows (``data``), normalised held-out rows (``test_data``) and the
    ``Normalization`` both went through (data/storage.py:346-437)."""

    def __init__(self, parent: Repository, k: int, **kwargs: Any):
        super().__init__(parent.fold_folder(k), **kwargs)
        self._held_out = Frame(self.test_csv)
        self._norm = Normalization(self)

    @property
    def normalization(self) -> 'Normalization':
        return self._norm

    @property
    def test_csv(self) -> Path:
        return self._root / 'test.csv'

    @property
    def _rotation_csv(self) -> Path:
        return self._root / 'X_rotation.csv'

    @property
    def test_data(self) -> Frame:
        return self._held_out

    @property
    def test_x(self) -> pd.DataFrame:
        return self._group(self._held_out, 'X_heading')

    @property
    def test_y(self) -> pd.DataFrame:
        return self._group(self._held_out, 'Y_heading')

    @property
    def X_rotation(self) -> np.ndarray:
        """Product of every rotation applied to this fold's inputs so far (identity if none)."""
        if not self._rotation_csv.exists():
            return np.eye(self.M)
        return Frame(self._rotation_csv, header=[0]).df.to_numpy()

    @X_rotation.setter
    def X_rotation(self, value: np.ndarray):
        """Rotate training and test inputs in place (row x -> value @ x) and accumulate the product in X_rotation.csv."""
        so_far, M = self.X_rotation, self.M
        for frame in (self._table, self._held_out):
            frame.df.iloc[:, :M] = frame.df.iloc[:, :M].to_numpy() @ np.asarray(value).T
            frame.write()
        Frame(self._rotation_csv, pd.DataFrame(so_far @ value))

    @classmethod
    def from_dfs(cls, parent: Repository, k: int, data: pd.DataFrame, test_data: pd.DataFrame, normalization: Optional[Path | str] = None,
                 is_normalization_applicable: bool = True) -> 'Fold':
        """Write ``fold.<k>`` from raw (un-normalised) training and test rows. The statistics come from ``normalization`` (a csv, copied
        into the fold) when given, from ``data`` otherwise."""
        fold = cls._blank(parent.fold_folder(k))
        fold._meta = {**copy.deepcopy(cls.META), **copy.deepcopy(parent.meta), 'k': k}
        if normalization:
            shutil.copy(Path(normalization), fold._root / 'normalization.csv')
        fold._norm = Normalization(fold, data, is_normalization_applicable)
        fold._table = Frame(fold._table_path, fold._norm.apply_to(data))
        fold._held_out = Frame(fold.test_csv, fold._norm.apply_to(test_data))
        fold._describe()
        return fold


@dataclass(frozen=True)
class _Moments:
    """What the two maps need from normalization.csv: lower end and width of the assumed-uniform inputs, mean and spread of the outputs."""
    x_low: pd.Series
    x_width: pd.Series
    y_mean: pd.Series
    y_spread: pd.Series

    @staticmethod
    def table(raw: pd.DataFrame) -> pd.DataFrame:
        """The five statistics rows of normalization.csv. An input assumed uniform on [min, max] has std = (max - min) / sqrt(12), hence
        min / max = mean -/+ sqrt(3) std and rng = 2 sqrt(3) std (data/storage.py:547-558)."""
        mean, std = raw.mean(), raw.std()                       # pandas: ddof = 1
        half = np.sqrt(3.0) * std
        return pd.DataFrame([mean, std, 2.0 * half, mean - half, mean + half], index=list(_STAT_ROWS))

    @classmethod
    def read(cls, stats: pd.DataFrame, M: int) -> '_Moments':
        return cls(stats.loc['min'].iloc[:M], stats.loc['rng'].iloc[:M], stats.loc['mean'].iloc[M:], stats.loc['std'].iloc[M:])


class Normalization:
    """Inputs: affine map of [min, max] onto [0, 1], clipped to [1e-12, 1 - 1e-12], then the standard-normal quantile function, so a
    uniform input becomes N(0, 1). Outputs: z-scores (data/storage.py:440-558)."""

    UNIFORM_MARGIN: float = 1.0E-12

    def __init__(self, fold: Repository, data: Optional[pd.DataFrame] = None, is_applicable: bool = True):
        """Statistics are read from ``<fold>/normalization.csv`` if that exists, else computed from ``data`` and written there; with
        neither they are looked for again on first use."""
        self._fold = fold
        self._active = is_applicable
        self._stats: Optional[Frame] = None
        if not self.csv.exists() and data is not None:
            self._stats = Frame(self.csv, _Moments.table(data))

    @property
    def csv(self) -> Path:
        return Path(self._fold.folder, 'normalization.csv')

    @property
    def frame(self) -> Frame:
        if self._stats is None:
            self._stats = Frame(self.csv)
        return self._stats

    @property
    def is_applicable(self) -> bool:
        return self._active

    @property
    def _moments(self) -> _Moments:
        return _Moments.read(self.frame.df, self._fold.M)

    def _halves(self, df: pd.DataFrame) -> tuple[pd.DataFrame, pd.DataFrame]:
        M = self._fold.M
        return df.iloc[:, :M], df.iloc[:, M:]

    def apply_to(self, df: pd.DataFrame) -> pd.DataFrame:
        """Raw (N, M+L) rows -> normalised rows."""
        if not self._active:
            return df
        mo, (X, Y) = self._moments, self._halves(df)
        unit = ((X - mo.x_low) / mo.x_width).clip(self.UNIFORM_MARGIN, 1.0 - self.UNIFORM_MARGIN)
        gaussian = pd.DataFrame(ndtri(unit.to_numpy()), index=X.index, columns=X.columns)
        return pd.concat([gaussian, (Y - mo.y_mean) / mo.y_spread], axis=1)

    def undo_from(self, df: pd.DataFrame) -> pd.DataFrame:
        """Normalised rows -> raw rows (exact inverse wherever the clip did not bite)."""
        if not self._active:
            return df
        mo, (X, Y) = self._moments, self._halves(df)
        unit = pd.DataFrame(ndtr(X.to_numpy()), index=X.index, columns=X.columns)
        return pd.concat([unit * mo.x_width + mo.x_low, Y * mo.y_spread + mo.y_mean], axis=1)

    def unscale_Y(self, dfY: pd.DataFrame) -> pd.DataFrame:
        """Output standard deviations back on the raw scale: multiplied by the spread, no mean added."""
        return dfY * self._moments.y_spread if self._active else dfY

    def X_gradient(self, X: np.ndarray, m: int | list[int]):
        """d raw input / d normalised input at normalised ``X[..., m]``: width x standard-normal density (1 when not applicable)."""
        z = X[..., m]
        if not self._active:
            return np.ones_like(z)
        return self._moments.x_width.to_numpy()[m] * np.exp(-0.5 * z * z) / np.sqrt(2.0 * np.pi)

    def __repr__(self) -> str:
        return str(self.csv)

    def __str__(self) -> str:
        return self.csv.name
