"""The checkpoint / resume mechanism behind every model of the plugin API (contract: reference base/classes.py:34-321, SURVEY.md
Appendix D): a model is a folder; each of its parameters is a small table mirrored to ``<folder>/<field>.csv`` the moment it is
assigned; ``meta.json`` (indent 8) holds the options. Folders written by the reference can be read here and vice versa:
parameter tables are ``DataFrame.to_csv`` with the default integer header and index column, read back with ``index_col=0``.

Three layers:
    ``Frame``  one table <-> one csv file (path given WITHOUT the suffix);
    ``Data``   the named tables of one folder; subclasses declare them through an inner ``NamedTuple`` whose defaults are the defaults
               of the parameters;
    ``Model``  folder + ``meta.json`` + a ``Data``.
"""
from __future__ import annotations

import json
import shutil
from abc import ABC, abstractmethod
from pathlib import Path
from typing import Any, Dict, Iterable, NamedTuple, Tuple

import numpy as np
import pandas as pd

_CSV_SUFFIX = '.csv'
_META_INDENT = 8                       # base/classes.py:292-295


def _fresh_folder(folder: Path | str) -> Path:
    """``folder`` emptied (created if need be)."""
    folder = Path(folder)
    if folder.exists():
        shutil.rmtree(folder, ignore_errors=True)
    folder.mkdir(mode=0o777, parents=True, exist_ok=False)
    return folder


class Frame:
    """One parameter table and its csv file."""

    def __init__(self, csv: Path | str, data=None, index=None, columns=None, dtype=None, copy=None, **kwargs: Any):
        """Without ``data`` the file is read (``kwargs`` -> ``pd.read_csv``, ``index_col=0`` unless overridden); with it the table is
        built like ``pd.DataFrame(data, index, columns, dtype, copy)`` and written at once (``kwargs`` -> ``to_csv``, remembered)."""
        self.csv = Path(csv)
        self._to_csv_options: Dict[str, Any] = {}
        if data is None:
            options = {'index_col': 0}
            options.update(kwargs)
            self._table = pd.read_csv(self.path, **options)
            return
        self._table = pd.DataFrame(data, index, columns, dtype, copy)
        self.write(**kwargs)

    @property
    def path(self) -> Path:
        """The file itself: ``csv`` with '.csv' appended to whatever suffix the name already carries ('a.b' -> 'a.b.csv')."""
        return self.csv.with_suffix(self.csv.suffix + _CSV_SUFFIX)

    # -- views of the table
    @property
    def df(self) -> pd.DataFrame:
        return self._table

    def _get_values(self) -> np.ndarray:
        return self._table.values

    def _set_values(self, value) -> None:
        self._table.iloc[:, :] = value
        self.write()

    np = property(_get_values, _set_values, doc='The values; assignment writes through to the file.')
    tf = property(_get_values, lambda self, value: self._set_values(np.asarray(value)),
                  doc='The reference returns a tf.Tensor here (base/classes.py:52-54); this backend has no TensorFlow: the same NumPy view.')

    def __call__(self, *args, **kwargs) -> np.ndarray:
        return self._get_values()

    # -- persistence
    def write(self, **kwargs: Any) -> 'Frame':
        """To csv; options given once stay in force for later writes (the GSA tables keep their ``float_format`` that way)."""
        self._to_csv_options.update(kwargs)
        self._table.to_csv(self.path, **self._to_csv_options)
        return self

    def broadcast_value(self, target_shape: Tuple[int, int], is_diagonal: bool = True) -> 'Frame':
        """Grow the table to ``target_shape`` by NumPy broadcasting and write it. A target with more than one row keeps only its
        diagonal when ``is_diagonal`` (independent parameters promoted to a covariance-shaped table). What cannot be broadcast --
        shrinking included -- raises IndexError."""
        try:
            grown = np.broadcast_to(self._get_values(), target_shape)
        except ValueError as error:
            raise IndexError(f'{self!r} has shape {self._table.shape} which cannot be broadcast to {target_shape}.') from error
        grown = np.diag(np.diagonal(grown)) if (is_diagonal and target_shape[0] > 1) else np.array(grown)
        self._table = pd.DataFrame(grown)
        return self.write()

    def __repr__(self) -> str:
        return str(self.csv)

    def __str__(self) -> str:
        return self.csv.name


class Data(ABC):
    """The tables of one folder, addressed by field name through ``frames`` (an instance of the subclass's ``NamedTuple``)."""

    class NamedTuple(NamedTuple):
        NotImplemented: Any = np.atleast_2d('NotImplemented')

    # -- what a subclass declares
    @classmethod
    @property
    def fields(cls) -> Tuple[str, ...]:
        return cls.NamedTuple._fields

    @classmethod
    @property
    def field_defaults(cls) -> Dict[str, Any]:
        return cls.NamedTuple._field_defaults

    @classmethod
    def make(cls, iterable: Iterable):
        return cls.NamedTuple._make(iterable)

    # -- construction
    def __init__(self, folder: Path | str, **kwargs: Any):
        """Tables from ``kwargs`` (defaults for the rest), written under ``folder``; a folder that does not exist yet is created."""
        folder = Path(folder)
        self._folder = folder if folder.exists() else _fresh_folder(folder)
        self._frames = None
        self.replace(**self.NamedTuple(**kwargs)._asdict())

    @classmethod
    def read(cls, folder: Path | str, **kwargs: Any) -> 'Data':
        """Every field read from ``folder``, except those given in ``kwargs``, which are written instead."""
        folder = Path(folder)
        tables = {name: Frame(folder / name, kwargs.get(name)) for name in cls.fields}
        return cls(folder, **tables)

    def replace(self, **kwargs: Any) -> 'Data':
        """Assign fields. A value that is not already a ``Frame`` becomes one -- i.e. is written to ``<folder>/<field>.csv`` now."""
        as_frames = {}
        for name, value in kwargs.items():
            as_frames[name] = value if isinstance(value, Frame) else Frame(self._folder / name, np.atleast_2d(np.asarray(value)))
        self._frames = self.NamedTuple(**as_frames) if self._frames is None else self._frames._replace(**as_frames)
        return self

    # -- access
    @property
    def folder(self) -> Path:
        return self._folder

    @property
    def frames(self):
        return self._frames

    def asdict(self) -> Dict[str, Any]:
        return self._frames._asdict()

    def __call__(self, *args, **kwargs):
        return self._frames

    def __repr__(self) -> str:
        return str(self._folder)

    def __str__(self) -> str:
        return self._folder.name

    # -- folders
    def move(self, dst_folder: Path | str) -> 'Data':
        """Re-home the tables in a fresh ``dst_folder``."""
        target = _fresh_folder(dst_folder)
        for name, frame in self.asdict().items():
            Frame(target / name, frame.df)
        self._folder = target
        return self

    @staticmethod
    def delete(folder: Path | str) -> Path:
        shutil.rmtree(Path(folder), ignore_errors=True)
        return Path(folder)

    @staticmethod
    def empty(folder: Path | str) -> Path:
        return _fresh_folder(folder)

    @staticmethod
    def copy(src_folder: Path | str, dst_folder: Path | str) -> Path:
        """Destructive copy of a whole model folder (the isotropic -> anisotropic warm start of ``run.gpr``)."""
        target = Data.delete(dst_folder)
        shutil.copytree(src=src_folder, dst=target)
        return target


class Model(ABC):
    """A folder with options (``meta.json``) and parameters (``Data``); ``calibrate`` is what a concrete model must supply."""

    class Data(Data):
        class NamedTuple(NamedTuple):
            NotImplemented: Any = np.atleast_2d('NotImplemented')

    @classmethod
    @property
    def META(cls) -> Dict[str, Any]:
        """Default options."""
        return {}

    @abstractmethod
    def __init__(self, folder: Path | str, read_data: bool = False, **kwargs: Any):
        """``read_data``: parameters come from the folder's csv files and ``kwargs`` override them; otherwise the folder is made
        if need be and the parameters are ``kwargs`` over the defaults."""
        self._folder = Path(folder)
        self._meta_json = self._folder / 'meta.json'
        self._implementation = None
        if read_data:
            self._data = self.Data.read(self._folder).replace(**kwargs)
            return
        self._folder.mkdir(mode=0o777, parents=True, exist_ok=True)
        self._data = self.Data(self._folder, **kwargs)

    @abstractmethod
    def calibrate(self, method: str, **kwargs) -> Dict[str, Any]:
        raise NotImplementedError('base.calibrate() must never be called.')

    @property
    def folder(self) -> Path:
        return self._folder

    @property
    def data(self) -> Data:
        return self._data

    @data.setter
    def data(self, value: Data):
        self._data = value

    def read_meta(self) -> Dict[str, Any]:
        return json.loads(self._meta_json.read_text())

    def write_meta(self, meta: Dict[str, Any]):
        self._meta_json.write_text(json.dumps(meta, indent=_META_INDENT))

    def __repr__(self) -> str:
        return str(self._folder)

    def __str__(self) -> str:
        return self._folder.name
