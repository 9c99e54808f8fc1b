"""GP regression: the romcomma.gpr plugin surface (GPR ABC, Kernel, Likelihood) with a HIP-backed implementation."""
