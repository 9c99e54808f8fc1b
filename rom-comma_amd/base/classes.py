"""Frame / Data / Model: every model parameter is a small DataFrame mirrored to ``<folder>/<field>.csv`` the moment it
changes, and every model owns a ``meta.json``. This is the reference's checkpoint/resume mechanism (base/classes.py:34-321);
the on-disk layout and file formats are kept so folders written by either implementation can be read by the other.
"""
from __future__ import annotations

import json
import shutil
from abc import ABC, abstractmethod
from pathlib import Path
from typing import Any, Dict, Iterable, NamedTuple, Tuple, Type

import numpy as np
import pandas as pd


class Frame:
    """A pandas DataFrame backed by ``<csv>.csv``. ``csv`` is given WITHOUT the suffix (base/classes.py:37,69)."""

    def __init__(self, csv: Path | str, data=None, index=None, columns=None, dtype=None, copy=None, **kwargs: Any):
        """``data is None`` reads the file (index_col=0 unless overridden); anything else is stored and written at once.
        ``kwargs`` go to ``pd.read_csv`` or ``DataFrame.to_csv`` respectively (base/classes.py:102-123)."""
        self.csv = Path(csv)
        self._write_options: Dict[str, Any] = {}
        if data is None:
            self._df = pd.read_csv(self._file, **({'index_col': 0} | kwargs))
        else:
            self._df = pd.DataFrame(data, index, columns, dtype, copy)
            self.write(**kwargs)

    @property
    def _file(self) -> Path:
        return self.csv.with_suffix(f'{self.csv.suffix}.csv')

    @property
    def df(self) -> pd.DataFrame:
        return self._df

    @property
    def np(self) -> np.ndarray:
        return self._df.values

    @np.setter
    def np(self, value):
        self._df.iloc[:, :] = value
        self.write()

    @property
    def tf(self) -> np.ndarray:
        """The reference returns a tf.Tensor here (base/classes.py:52-54); this backend has no TensorFlow: a NumPy view."""
        return self.np

    @tf.setter
    def tf(self, value):
        self.np = np.asarray(value)

    def write(self, **kwargs: Any) -> 'Frame':
        """Write to csv; the options are remembered for later writes (base/classes.py:61-70)."""
        self._write_options |= kwargs
        self._df.to_csv(self._file, **self._write_options)
        return self

    def broadcast_value(self, target_shape: Tuple[int, int], is_diagonal: bool = True) -> 'Frame':
        """Broadcast to ``target_shape``; a square target keeps only the diagonal when ``is_diagonal``. Raises IndexError when
        the value cannot be broadcast (base/classes.py:72-89)."""
        try:
            values = np.array(np.broadcast_to(self.np, target_shape))
        except ValueError:
            raise IndexError(f'{self!r} has shape {self.df.shape} which cannot be broadcast to {target_shape}.')
        if is_diagonal and target_shape[0] > 1:
            values = np.diag(np.diagonal(values))
        self._df = pd.DataFrame(values)
        return self.write()

    def __call__(self, *args, **kwargs) -> np.ndarray:
        return self.np

    def __repr__(self) -> str:
        return str(self.csv)

    def __str__(self) -> str:
        return self.csv.name


class Data(ABC):
    """A NamedTuple of Frames living in one folder (base/classes.py:127-236). Subclasses override ``NamedTuple``."""

    class NamedTuple(NamedTuple):
        NotImplemented: Any = np.atleast_2d('NotImplemented')

    def __init__(self, folder: Path | str, **kwargs: Any):
        folder = Path(folder)
        self._folder = folder if folder.exists() else self.empty(folder)
        self._frames = None
        self.replace(**self.NamedTuple(**kwargs)._asdict())

    @classmethod
    def make(cls, iterable: Iterable):
        return cls.NamedTuple._make(iterable)

    @classmethod
    @property
    def fields(cls) -> Tuple[str, ...]:
        return cls.NamedTuple._fields

    @classmethod
    @property
    def field_defaults(cls) -> Dict[str, Any]:
        return cls.NamedTuple._field_defaults

    def asdict(self) -> Dict[str, Any]:
        return self._frames._asdict()

    def replace(self, **kwargs: Any) -> 'Data':
        """Assign fields; each assignment is written to ``<folder>/<field>.csv`` immediately (base/classes.py:155-160)."""
        frames = {key: (value if isinstance(value, Frame) else Frame(self._folder / key, np.atleast_2d(np.asarray(value))))
                  for key, value in kwargs.items()}
        self._frames = self.NamedTuple(**frames) if self._frames is None else self._frames._replace(**frames)
        return self

    @property
    def folder(self) -> Path:
        return self._folder

    @property
    def frames(self):
        return self._frames

    def __call__(self, *args, **kwargs):
        return self._frames

    def __repr__(self) -> str:
        return str(self._folder)

    def __str__(self) -> str:
        return self._folder.name

    @classmethod
    def read(cls, folder: Path | str, **kwargs: Any) -> 'Data':
        """Read every field from ``folder``; ``kwargs`` override fields after reading (base/classes.py:203-215)."""
        folder = Path(folder)
        return cls(folder, **{field: Frame(folder / field, kwargs.get(field, None)) for field in cls.fields})

    def move(self, dst_folder: Path | str) -> 'Data':
        dst = self.empty(dst_folder)
        for key, frame in self.asdict().items():
            Frame(dst / key, frame.df)
        self._folder = dst
        return self

    @staticmethod
    def delete(folder: Path | str) -> Path:
        folder = Path(folder)
        shutil.rmtree(folder, ignore_errors=True)
        return folder

    @staticmethod
    def empty(folder: Path | str) -> Path:
        folder = Data.delete(folder)
        folder.mkdir(mode=0o777, parents=True, exist_ok=False)
        return folder

    @staticmethod
    def copy(src_folder: Path | str, dst_folder: Path | str) -> Path:
        dst_folder = Data.delete(dst_folder)
        shutil.copytree(src=src_folder, dst=dst_folder)
        return dst_folder


class Model(ABC):
    """Folder + ``meta.json`` + ``Data`` (base/classes.py:239-321)."""

    class Data(Data):
        class NamedTuple(NamedTuple):
            NotImplemented: Any = np.atleast_2d('NotImplemented')

    @classmethod
    @property
    def META(cls) -> Dict[str, Any]:
        return {}

    @abstractmethod
    def __init__(self, folder: Path | str, read_data: bool = False, **kwargs: Any):
        self._folder = Path(folder)
        self._meta_json = self._folder / 'meta.json'
        if read_data:
            self._data = self.Data.read(self._folder).replace(**kwargs)
        else:
            self._folder.mkdir(mode=0o777, parents=True, exist_ok=True)
            self._data = self.Data(self._folder, **kwargs)
        self._implementation = None

    @property
    def folder(self) -> Path:
        return self._folder

    @property
    def data(self) -> Data:
        return self._data

    @data.setter
    def data(self, value: Data):
        self._data = value

    @abstractmethod
    def calibrate(self, method: str, **kwargs) -> Dict[str, Any]:
        raise NotImplementedError('base.calibrate() must never be called.')

    def read_meta(self) -> Dict[str, Any]:
        with open(self._meta_json, mode='r') as file:
            return json.load(file)

    def write_meta(self, meta: Dict[str, Any]):
        with open(self._meta_json, mode='w') as file:
            json.dump(meta, file, indent=8)

    def __repr__(self) -> str:
        return str(self._folder)

    def __str__(self) -> str:
        return self._folder.name
