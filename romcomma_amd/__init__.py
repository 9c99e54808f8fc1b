"""Importable alias for the package directory ``rom-comma_amd/`` (a hyphen is not a legal module name).

``import romcomma_amd`` resolves every submodule (``romcomma_amd.gpr``, ``romcomma_amd._lib`` ...) inside ``rom-comma_amd/``.
"""
from pathlib import Path as _Path

__path__ = [str(_Path(__file__).resolve().parent.parent / 'rom-comma_amd')]
exec(compile((_Path(__path__[0]) / '__init__.py').read_text(), str(_Path(__path__[0]) / '__init__.py'), 'exec'))
