"""Timing and environment context managers (reference user/contexts.py:32-82)."""
from __future__ import annotations

import time
from contextlib import contextmanager
from datetime import timedelta


@contextmanager
def Timer(name: str = '', is_inline: bool = True):
    """Prints 'Running <name> took H:MM:SS' around the block, truncated to whole seconds like the reference
    (user/contexts.py:50); an empty name is silent. The elapsed seconds are also yielded in a one-element list."""
    start = time.time()
    elapsed = [0.0]
    if name:
        print(f'Running {name}', end='' if is_inline else '...\n', flush=True)
    try:
        yield elapsed
    finally:
        elapsed[0] = time.time() - start
        if name:
            print(f'{" " if is_inline else "..."}took {timedelta(seconds=int(elapsed[0]))}.', flush=True)


@contextmanager
def Environment(name: str = '', device: str = '', **kwargs):
    """The reference forces float64 and selects a TensorFlow device here (user/contexts.py:55-82). This backend is fp64-only
    and GPU-only; ``device`` may end in 'GPU:<i>' to pin the process to HIP device i (otherwise LOCAL_RANK decides)."""
    import os
    with Timer(name):
        if name:
            print(' using librcgp(float=\'float64\')', end='')
        position = device.rfind('GPU:')
        if position >= 0 and device[position + 4:].isdigit():
            os.environ['LOCAL_RANK'] = device[position + 4:]
            print(f' on /GPU:{device[position + 4:]}', end='')
        elif device.rfind('CPU') >= 0:
            raise RuntimeError('this backend has no CPU path: request a GPU device')
        if name:
            print('...')
        yield
        if name:
            print('...Running ' + name, end='')
