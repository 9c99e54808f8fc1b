"""Data storage: Repository, Fold, Normalization, Frame."""
from romcomma_amd.data.storage import Frame, Repository, Fold, Normalization   # noqa: F401
