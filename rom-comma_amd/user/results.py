"""Collect: concatenate same-named csv files across folders or across the folds of a Repository, tagging each block with
extra leading columns (reference user/results.py:45-128)."""
from __future__ import annotations

from pathlib import Path
from shutil import rmtree
from typing import Any, Dict, Union

import numpy as np
import pandas as pd

from romcomma_amd.base.classes import Data
from romcomma_amd.data.storage import Fold, Repository


def copy(src: Path | str, dst: Path | str) -> Path:
    """Destructive folder copy."""
    Data.copy(src, dst)
    return Path(dst)


class Collect:
    """``csvs``: {csv name without extension: pd.read_csv options}. ``folders``: {folder: {column name: value}} -- the columns
    are inserted at the left, last key leftmost. Output is written with index=False, float_format='%.6f'."""

    def __init__(self, csvs: Dict[str, Dict[str, Any]] | None = None, folders: Dict[str, Dict[str, Any]] | None = None,
                 ignore_missing: bool = False, **kwargs: Any):
        self.csvs = {} if csvs is None else csvs
        self.folders = {} if folders is None else folders
        self.ignore_missing = ignore_missing
        self.write_options = {'index': False, 'float_format': '%.6f'} | kwargs

    def __call__(self, dst: Union[Repository, Path, str], is_existing_deleted: bool = False, **kwargs: Any) -> 'Collect':
        if isinstance(dst, Repository):
            return self.from_folds(dst, is_existing_deleted, **kwargs)
        return self.from_folders(dst, is_existing_deleted, **kwargs)

    def from_folders(self, dst: Union[Path, str], is_existing_deleted: bool = False, **kwargs: Any) -> 'Collect':
        dst = Path(dst)
        if is_existing_deleted:
            rmtree(dst, ignore_errors=True)
        dst.mkdir(mode=0o777, parents=True, exist_ok=True)
        for csv, read_options in self.csvs.items():
            blocks = []
            for folder, columns in self.folders.items():
                file = Path(folder) / f'{csv}.csv'
                if not file.exists() and self.ignore_missing:
                    continue
                block = pd.read_csv(file, **read_options)
                for key, value in columns.items():
                    block.insert(0, key, np.full(block.shape[0], value), True)
                blocks.append(block)
            if blocks:
                pd.concat(blocks, axis=0, ignore_index=True).to_csv(dst / f'{csv}.csv', **(self.write_options | kwargs))
            elif not self.ignore_missing:
                raise FileNotFoundError(f'no {csv}.csv found in {list(self.folders)}')
        return self

    def from_folds(self, dst: Repository, is_existing_deleted: bool = False, **kwargs: Any) -> 'Collect':
        """For every sub-folder in ``self.folders`` gather it from each fold of ``dst`` into ``dst.folder / sub_folder`` with extra
        'fold' and 'N' columns (user/results.py:98-114)."""
        if isinstance(dst, Fold):
            raise NotADirectoryError('dst is a Fold, which cannot contain other Folds, so cannot be Collected from.')
        folds = tuple(Fold(dst, k) for k in dst.folds)
        for sub_folder, extra_columns in self.folders.items():
            folders = {fold.folder / sub_folder: {'fold': fold.meta['k'], 'N': fold.N} | extra_columns for fold in folds}
            Collect(self.csvs, folders, self.ignore_missing, **self.write_options).from_folders(dst.folder / sub_folder, is_existing_deleted, **kwargs)
        return self
