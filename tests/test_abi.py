"""CPU tests of the drop-in boundary: librcgp.so builds, loads, exports every symbol include/rcgp.h declares, and refuses to
compute without a GPU (no CPU fallback, no route through oracle/)."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / 'include' / 'rcgp.h').read_text()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(rcgp_[a-z_0-9]+)\s*\(', text)))


def test_header_symbols_all_exported():
    from romcomma_amd import _lib
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/rcgp.h but not exported by librcgp.so'
    assert sorted(_lib.SIGNATURES) == declared, 'ctypes SIGNATURES and include/rcgp.h disagree'
    assert lib.rcgp_version() == 100


def test_only_abi_symbols_are_exported():
    """-fvisibility=hidden: the dynamic symbol table holds the C ABI and nothing with C++ linkage from our sources."""
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', str(ROOT / 'rom-comma_amd' / 'librcgp.so')], capture_output=True, text=True).stdout
    names = [line.split()[-1] for line in out.splitlines() if ' T ' in line]
    ours = [n for n in names if n.startswith('rcgp_')]
    assert sorted(ours) == _declared_symbols()
    assert not [n for n in names if n.startswith('_Z') and 'rc_' in n]


def test_no_silent_cpu_fallback():
    from romcomma_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip('a GPU is visible: the no-device error path cannot be exercised here')
    with pytest.raises(_lib.RcgpError, match='no HIP device'):
        _lib.RcGP(np.zeros((8, 2)), np.zeros(8))


def test_bad_arguments_rejected_before_touching_a_device():
    from romcomma_amd import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    x = np.zeros((4, 2))
    y = np.zeros(4)
    dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    assert lib.rcgp_create(ctypes.byref(h), 0, 0, 2, dp(x), dp(y)) == -2          # N < 1
    assert lib.rcgp_create(ctypes.byref(h), 0, 4, 257, dp(x), dp(y)) == -2        # M > 256
    assert lib.rcgp_create_mo(ctypes.byref(h), 0, 4, 257, 2, dp(x), dp(x)) == -2  # M > 256, covariant
    assert lib.rcgp_create_mo(ctypes.byref(h), 0, 4, 2, 65, dp(x), dp(x)) == -2   # L > 64
    assert lib.rcgp_create(ctypes.byref(h), 0, 4, 0, dp(x), dp(y)) == -2          # M < 1
    assert b'bad argument' in lib.rcgp_last_error(None)
    assert lib.rcgp_destroy(None) == -1
    assert lib.rcgp_lml(None, None) == -1


def test_product_does_not_import_oracle():
    """The product package must never import, call or link anything under oracle/."""
    import ast
    for p in (ROOT / 'rom-comma_amd').rglob('*'):
        if not p.is_file():
            continue
        if p.suffix == '.py':
            for node in ast.walk(ast.parse(p.read_text())):
                names = []
                if isinstance(node, ast.Import):
                    names = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    names = [node.module or '']
                assert not [n for n in names if n == 'oracle' or n.startswith('oracle.')], f'{p} imports the oracle'
            assert 'import_module(' not in p.read_text() and '__import__(' not in p.read_text(), f'{p} imports dynamically'
        elif p.suffix in ('.hip', '.h', '.cpp'):
            includes = [line for line in p.read_text().splitlines() if line.lstrip().startswith('#include')]
            assert not [line for line in includes if 'oracle' in line], f'{p} includes oracle code'
    assert 'oracle' not in (ROOT / 'rom-comma_amd' / 'csrc' / 'Makefile').read_text()
