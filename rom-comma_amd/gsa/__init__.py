"""Global sensitivity analysis: closed-form Sobol indices of a fitted GP (the romcomma.gsa plugin surface)."""
