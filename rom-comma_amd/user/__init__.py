"""User-level helpers: synthetic data, timers, the run.gpr / run.gsa orchestration."""
