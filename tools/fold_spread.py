import sys; sys.path.insert(0,'.')
import numpy as np, time
from romcomma_amd import _lib
from romcomma_amd.gpr.optimize import fit_lbfgsb
from romcomma_amd.user.sample import synthetic_cv_fold
N,M=int(sys.argv[1]),int(sys.argv[2])
for k in range(8):
    X,y=synthetic_cv_fold(N,M,k)
    gp=_lib.RcGP(X,y); t=time.time(); f=fit_lbfgsb(gp,5.0*np.ones(M),2.0,0.02); print(k, X.shape, f['nfev'], round(f['log_marginal'],3), round(time.time()-t,2),'s', flush=True); gp.close()
