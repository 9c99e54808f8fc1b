"""``Collect``: stack same-named csv files found in several folders -- or in the same sub-folder of every fold of a Repository -- into one
csv per name, each source block tagged by extra leading columns (interface: reference user/results.py:45-128).

    Collect({'S': {}, 'V': {}}, {'gpr.v.a/gsa/closed': {}}).from_folds(repo)     ->  <repo>/gpr.v.a/gsa/closed/{S,V}.csv with 'fold', 'N' columns

Written with ``index=False, float_format='%.6f'`` unless overridden.
"""
import shutil
from pathlib import Path
from typing import Any, Optional

import pandas as pd

from romcomma_amd.base.classes import Data
from romcomma_amd.data.storage import Fold, Repository


def copy(src: Path | str, dst: Path | str) -> Path:
    """Replace folder ``dst`` by a copy of folder ``src``."""
    Data.copy(src, dst)
    return Path(dst)


def _tagged(block: pd.DataFrame, tags: dict[str, Any]) -> pd.DataFrame:
    """``block`` with one constant column per tag in front; the LAST tag given ends up leftmost."""
    for column, constant in tags.items():
        block.insert(0, column, constant, allow_duplicates=True)
    return block


class Collect:
    """``csvs``: {file stem: ``pd.read_csv`` options}. ``folders``: {source folder: {tag column: value}}."""

    def __init__(self, csvs: Optional[dict[str, dict[str, Any]]] = None, folders: Optional[dict[Any, dict[str, Any]]] = None,
                 ignore_missing: bool = False, **kwargs: Any):
        self.csvs = dict(csvs or {})
        self.folders = dict(folders or {})
        self.ignore_missing = ignore_missing
        self.write_options = {'index': False, 'float_format': '%.6f', **kwargs}

    def __call__(self, dst: Repository | Path | str, is_existing_deleted: bool = False, **kwargs: Any) -> 'Collect':
        gather = self.from_folds if isinstance(dst, Repository) else self.from_folders
        return gather(dst, is_existing_deleted, **kwargs)

    def from_folders(self, dst: Path | str, is_existing_deleted: bool = False, **kwargs: Any) -> 'Collect':
        """One ``<dst>/<stem>.csv`` per stem: the blocks read from every source folder, in folder order. A stem found nowhere raises
        FileNotFoundError (as does one missing from a single folder) unless ``ignore_missing``."""
        target = Path(dst)
        if is_existing_deleted and target.exists():
            shutil.rmtree(target, ignore_errors=True)
        target.mkdir(parents=True, exist_ok=True)
        how_to_write = {**self.write_options, **kwargs}
        for stem, how_to_read in self.csvs.items():
            sources = [(Path(source) / f'{stem}.csv', tags) for source, tags in self.folders.items()]
            present = [(file, tags) for file, tags in sources if file.exists() or not self.ignore_missing]
            blocks = [_tagged(pd.read_csv(file, **how_to_read), tags) for file, tags in present]
            if blocks:
                pd.concat(blocks, ignore_index=True).to_csv(target / f'{stem}.csv', **how_to_write)
            elif not self.ignore_missing:
                raise FileNotFoundError(f'no {stem}.csv found in {list(self.folders)}')
        return self

    def from_folds(self, dst: Repository, is_existing_deleted: bool = False, **kwargs: Any) -> 'Collect':
        """``self.folders`` names sub-folders RELATIVE to a fold: each is gathered over all folds of ``dst`` into the same sub-folder
        of ``dst`` itself, blocks tagged 'fold' (k) and 'N' (training rows of that fold) ahead of the sub-folder's own tags."""
        if isinstance(dst, Fold):
            raise NotADirectoryError(f'{dst!r} is a Fold: it has no folds of its own to collect from')
        per_fold = [(fold.folder, {'fold': fold.meta['k'], 'N': fold.N}) for fold in (Fold(dst, k) for k in dst.folds)]
        for sub_folder, tags in self.folders.items():
            sources = {where / sub_folder: {**fold_tags, **tags} for where, fold_tags in per_fold}
            Collect(self.csvs, sources, self.ignore_missing, **self.write_options).from_folders(dst.folder / sub_folder, is_existing_deleted, **kwargs)
        return self
