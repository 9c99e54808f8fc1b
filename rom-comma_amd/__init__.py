"""MI355X-native GP-regression + closed-form Sobol backend behind romcomma's gpr/gsa plugin API.

Layout: ``csrc/`` hand-written gfx950 HIP kernels + the C ABI (``include/rcgp.h``); ``_lib`` the ctypes binding;
``base``/``data``/``gpr``/``gsa`` the host-side mirror of the reference's Model/Fold/GPR/GSA interface.
"""
__version__ = '0.1.0'
