"""MI355X-native GP-regression + closed-form Sobol backend behind romcomma's gpr/gsa plugin API.

Layout: ``csrc/`` hand-written gfx950 HIP kernels + the C ABI (``include/rcgp.h``); ``_lib`` the ctypes binding;
``base``/``data``/``gpr``/``gsa`` the host-side mirror of the reference's Model/Fold/GPR/GSA interface.
"""
__version__ = '0.1.0'

import os as _os

# dmabuf IPC for multi-rank RCCL on one node; must precede the process's first GPU call (see dist.prepare_environment)
_os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
