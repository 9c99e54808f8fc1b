"""GP regression models: the ``GPR`` plugin contract and ``HipGP``, its MI355X implementation.

``GPR`` restates the abstract interface of reference gpr/models.py:88-320 (same constructor signature, same abstract members,
same inherited helpers and on-disk side effects). ``HipGP`` stands where the reference's GPflow-backed ``MOGP`` stands
(gpr/models.py:324-463): every numeric step -- Gram matrix, Cholesky, solves, log marginal likelihood and its gradient,
prediction -- runs in librcgp.so on the GPU; this file only moves parameters between the CSV store and the C ABI.
``MOGP`` is exported as an alias so existing ``run`` scripts resolve the same name.
"""
from __future__ import annotations

from abc import abstractmethod
from pathlib import Path
from typing import Any, Dict, NamedTuple, Tuple

import numpy as np
import pandas as pd

from romcomma_amd import _lib
from romcomma_amd.base.classes import Data, Frame, Model
from romcomma_amd.data.storage import Fold
from romcomma_amd.data.storage import Frame as DataFrameCSV
from romcomma_amd.gpr.kernels import Kernel
from romcomma_amd.gpr.optimize import fit_lbfgsb, fit_lbfgsb_batch, fit_lbfgsb_mo


class Likelihood(Model):
    """Gaussian likelihood (noise) variance store (gpr/models.py:35-84)."""

    class Data(Data):
        class NamedTuple(NamedTuple):
            variance: Any = np.atleast_2d(0.02)          # (1,L) independent noise variances; reference default 0.02
            log_marginal: Any = np.atleast_2d(1.0)       # output: the log marginal likelihood after calibration

    @classmethod
    @property
    def META(cls) -> Dict[str, Any]:
        return {'variance': True, 'covariance': True}

    VARIANCE_FLOOR: float = 1.0001E-6                    # gpr/models.py:62-65

    def __init__(self, parent: 'GPR', read_data: bool = False, **kwargs: Any):
        super().__init__(parent.folder / 'likelihood', read_data, **kwargs)
        self._parent = parent
        self._trainable = self.META

    @property
    def is_covariant(self) -> bool:
        return self._data.frames.variance.df.shape[0] > 1

    def calibrate(self, **kwargs: Any) -> Dict[str, Any]:
        """Records whether the noise variance is trainable (gpr/models.py:71-80)."""
        self._trainable = self.META | kwargs
        return self._trainable

    @property
    def trainable(self) -> Dict[str, Any]:
        return self._trainable


class GPR(Model):
    """Interface to a Gaussian Process (gpr/models.py:88-320)."""

    class Data(Data):
        class NamedTuple(NamedTuple):
            kernel: Any = np.atleast_2d(None)            # [[type identifier]], e.g. 'kernels.RBF'; never set externally

    KERNEL_FOLDER_NAME: str = 'kernel'

    def __init__(self, name: str, fold: Fold, is_read: bool | None, is_covariant: bool, is_isotropic: bool,
                 kernel_parameters: Kernel.Data | None = None, likelihood_variance: np.ndarray | None = None):
        """A GP called ``name`` inside ``fold``. Contract (reference gpr/models.py:290-320):

        * the training arrays are float64 COPIES of the fold's X (N,M) and Y (N,L) -- later edits of the fold do not reach the GP;
        * the model store is ``fold.folder / name``; ``is_read`` decides between reading it and starting from defaults;
        * an explicit ``likelihood_variance`` / ``kernel_parameters`` wins over what is on file; a kernel that is read back is
          re-created from the type identifier in ``kernel.csv``, a new one records its identifier there;
        * finally everything is broadcast to L outputs (and M or 1 lengthscales), which also builds the implementation.
        """
        self._fold = fold
        self._N, self._M, self._L = fold.N, fold.M, fold.L
        self._X, self._Y = (np.array(frame.to_numpy(), dtype=np.float64, order='C', copy=True) for frame in (fold.X, fold.Y))
        super().__init__(fold.folder / name, is_read)
        noise_override = {} if likelihood_variance is None else {'variance': likelihood_variance}
        self._likelihood = Likelihood(self, is_read, **noise_override)
        self._kernel = self._restored_kernel() if (is_read and kernel_parameters is None) else self._fresh_kernel(is_read, kernel_parameters)
        self.broadcast_parameters(is_covariant, is_isotropic)

    def _restored_kernel(self) -> Kernel:
        """The kernel whose type ``kernel.csv`` names, read from the kernel folder."""
        identifier = self.data.frames.kernel.np[0, 0]
        return Kernel.TypeFromIdentifier(identifier)(self._folder / self.KERNEL_FOLDER_NAME, True)

    def _fresh_kernel(self, is_read: bool | None, parameters: Kernel.Data | None) -> Kernel:
        """A kernel of the type ``parameters`` belong to (default parameters when none are given); its identifier goes to ``kernel.csv``."""
        kernel_folder = self._folder / self.KERNEL_FOLDER_NAME
        if parameters is None:
            parameters = Kernel.Data(kernel_folder)
        kernel_type = Kernel.TypeFromParameters(parameters)
        kernel = kernel_type(kernel_folder, is_read, **parameters.asdict())
        self._data.replace(kernel=np.atleast_2d(kernel_type.TYPE_IDENTIFIER))
        return kernel

    # ---- plain accessors
    @classmethod
    @property
    @abstractmethod
    def META(cls) -> Dict[str, Any]:
        """Hyper-parameter optimiser options."""

    @property
    def fold(self) -> Fold:
        return self._fold

    @property
    def test_csv(self) -> Path:
        return self._folder / 'test.csv'

    @property
    def test_summary_csv(self) -> Path:
        return self._folder / 'test_summary.csv'

    @property
    def kernel(self) -> Kernel:
        return self._kernel

    @property
    def likelihood(self) -> Likelihood:
        return self._likelihood

    @property
    def L(self) -> int:
        return self._L

    @property
    def M(self) -> int:
        return self._M

    @property
    def N(self) -> int:
        return self._N

    # ---- the contract an implementation must fulfil
    @property
    @abstractmethod
    def implementation(self) -> Tuple[Any, ...]:
        """The backend objects behind this GP."""

    @property
    @abstractmethod
    def X(self) -> Any:
        """Training inputs (N,M)."""

    @property
    @abstractmethod
    def Y(self) -> Any:
        """Training outputs (N,L)."""

    @property
    @abstractmethod
    def K_cho(self) -> np.ndarray:
        """Cholesky factor of kernel(X,X) + noise: (L,N,N) for independent outputs."""

    @property
    @abstractmethod
    def K_inv_Y(self) -> np.ndarray:
        """(L,1,N): ChoSolve(K_cho, Y)."""

    @abstractmethod
    def calibrate(self, **kwargs) -> Dict[str, Any]:
        raise NotImplementedError

    @abstractmethod
    def predict(self, x: np.ndarray, y_instead_of_f: bool = True) -> Tuple[np.ndarray, np.ndarray]:
        """(mean (o,L), standard deviation (o,L)) at the (o,M) inputs ``x``."""

    @abstractmethod
    def predict_gradient(self, x: np.ndarray, y_instead_of_f: bool = True) -> Tuple[np.ndarray, np.ndarray]:
        """Gradient GP dy/dx."""

    # ---- inherited helpers (pure host code)
    def predict_df(self, x: np.ndarray, y_instead_of_f: bool = True, is_normalized: bool = True) -> pd.DataFrame:
        """The prediction at ``x`` (o,M) as one frame: the M input columns under the fold's X heading, then L columns 'Mean' and L
        columns 'SD', second-level labels as in the fold's test data. ``is_normalized=False`` maps the inputs and the mean back to
        the original units and rescales (not shifts) the SD (contract: SURVEY.md section 8f rank 3; reference gpr/models.py:202-222)."""
        labels = self._fold.test_data.df.columns                      # (X heading, X.m) ... (Y heading, Y.l)
        y_heading = self._fold.meta['data']['Y_heading']
        y_labels = labels[self._M:]
        x = np.asarray(x, dtype=np.float64)
        mean, sd = self.predict(x, y_instead_of_f)
        located = pd.DataFrame(np.hstack([x, mean]), columns=labels)
        spread = pd.DataFrame(np.asarray(sd, dtype=np.float64), columns=y_labels)
        if not is_normalized:
            normalization = self._fold.normalization
            located, spread = normalization.undo_from(located), normalization.unscale_Y(spread)
        relabel = lambda frame, new: frame.rename(columns={y_heading: new}, level=0)      # noqa: E731
        return pd.concat([relabel(located, 'Mean'), relabel(spread, 'SD')], axis=1)

    def test(self) -> DataFrameCSV:
        """Score the GP on the fold's held-out rows: per row Mean, SD, Abs Error, Z Score, Outlier (Z^2 > 4, plus Any/All
        outputs) -> ``test.csv``; per output RMSE, mean SD, outlier fraction -> ``test_summary.csv`` (gpr/models.py:235-272)."""
        Y_heading = self._fold.meta['data']['Y_heading']
        frame = DataFrameCSV(self.test_csv, self._fold.test_data.df.copy())
        truth = frame.df.loc[:, [Y_heading]]
        mean, sd = self.predict(self._fold.test_x.values)

        def like_truth(values, label: str) -> pd.DataFrame:
            columns = truth.rename(columns={Y_heading: label}, level=0).columns
            return pd.DataFrame(np.asarray(values), index=truth.index, columns=columns)
        error = truth.to_numpy(dtype=float) - mean
        z = error / sd
        outlier = z ** 2 > 4.0
        outliers = like_truth(outlier, 'Outlier')
        outliers[('Outlier', 'Any Output')] = np.logical_or.reduce(outlier, axis=1)
        outliers[('Outlier', 'All Outputs')] = np.logical_and.reduce(outlier, axis=1)
        frame.df = frame.df.join([like_truth(mean, 'Mean'), like_truth(sd, 'SD'), like_truth(np.abs(error), 'Abs Error'),
                                  like_truth(z, 'Z Score'), outliers])
        frame.write()
        rmse = pd.DataFrame(like_truth(error ** 2, 'RMSE').mean(axis=0) ** 0.5).transpose()
        mean_sd = pd.DataFrame(like_truth(sd, 'SD').mean(axis=0)).transpose()
        fraction = pd.DataFrame(outliers.mean(axis=0)).transpose()
        DataFrameCSV(self.test_summary_csv, rmse.join([mean_sd, fraction]))
        return frame

    def broadcast_parameters(self, is_covariant: bool, is_isotropic: bool) -> 'GPR':
        """Grow the stored parameters to this GP's shape and rebuild the implementation; returns ``self``. Variances (noise and
        kernel) become (1,L), or (L,L) with only the diagonal kept for a covariant GP; lengthscales (L,M), or (L,1) when isotropic.
        Growing is silent, an impossible broadcast raises IndexError in the frame (contract: reference gpr/models.py:274-288)."""
        L, n_lengthscales = self._L, (1 if is_isotropic else self._M)
        variance_shape = (L, L) if is_covariant else (1, L)
        self._likelihood.data.frames.variance.broadcast_value(target_shape=variance_shape, is_diagonal=True)
        self._kernel.broadcast_parameters(variance_shape=variance_shape, M=n_lengthscales)
        self._reset_implementation()
        return self

    def _reset_implementation(self) -> None:
        self._implementation = None            # the property rebuilds on None
        self._implementation = self.implementation


def default_units_per_gpu(N: int) -> int:
    """How many equal-sized units one GPU is given at once when nobody says (argument ``units_per_gpu``, environment RCGP_UNITS): by
    what a batched evaluation was measured to gain on an MI355X (a batch of 16 at N = 4096 runs 2.6x as many
    evaluations per second as one unit at a time (6x at N = 2048, 11x at N = 1024: there an evaluation is a string of launch latencies and
    a batch costs little more than one unit), a batch of 4 at N = 8192 1.25x, two units at N = 16384 1.05x)."""
    import os
    wanted = int(os.environ.get('RCGP_UNITS', 0))
    if wanted > 0:
        return min(wanted, _lib.MAX_BATCH)
    padded = -(-int(N) // 128) * 128
    return 16 if padded <= 4096 else (4 if padded <= 12288 else 2)


class _PoolSlots:
    """The handles of a ``HipGP`` pool dealt to more outputs than there are handles: output ``u`` gets a free slot when its fit starts
    (``bind``: the slot's handle with the output's targets loaded) and gives it back when the fit has ended (``release``)."""

    def __init__(self, gp: 'HipGP'):
        self._gp, self._free, self._slot = gp, list(range(gp.pool_size)), {}

    def bind(self, u: int):
        slot = self._free.pop(0)
        self._slot[u] = slot
        self._gp._unit_signature.pop(slot, None)
        return self._gp._load_output(slot, u)

    def release(self, u: int):
        slot = self._slot.pop(u)
        self._gp._unit_signature.pop(slot, None)           # (the handle holds the optimum of ``u``; _select compares and re-sends as needed)
        self._free.append(slot)


class HipGP(GPR):
    """ARD-RBF GPs on one MI355X through librcgp.so.

    Independent outputs: a POOL of device handles (units), each with its own copy of X and its own N x N work matrices. Output ``l``
    lives on unit ``l mod P``; with P = L every output keeps its factor, L^-1 and alpha on the device between calls. Where the
    reference loops ``for gp in self._implementation`` one output after the other (gpr/models.py:360-361), ``calibrate`` fits the
    outputs of a pool-sized group AT ONCE: their L-BFGS-B runs advance in lockstep and every round of evaluations is one batched
    schedule on the GPU (``rcgp_lml_grad_batch``) -- a single factorisation of the sizes the reference is run at cannot fill this chip.
    Each output's fit is, to the last bit, the fit it would have had alone. ``units_per_gpu`` (argument, else the environment
    variable RCGP_UNITS, else chosen from N) bounds P; P = 1 is the reference's one-after-the-other loop on one handle.
    To spread outputs or folds over GPUs, run one process per GPU and give each its share (``romcomma_amd.user.run``).

    Covariant outputs (``is_covariant``): one (L N) x (L N) system, the reference's ``romcomma.gpf.models.MOGPR`` behind the
    covariant branches of gpr/models.py:335-338, 363-367, 377-379, 429-431 -- an ``rcgp_create_mo`` handle.
    """

    @classmethod
    @property
    def META(cls) -> Dict[str, Any]:
        return {'maxiter': 5000, 'gtol': 1E-16}          # gpr/models.py:327-330

    def __init__(self, name: str, fold: Fold, is_read: bool | None, is_covariant: bool, is_isotropic: bool,
                 kernel_parameters: Kernel.Data | None = None, likelihood_variance: np.ndarray | None = None, device: int | None = None,
                 units_per_gpu: int | None = None):
        self._is_covariant = bool(is_covariant)
        self._device = device
        self._units_per_gpu = units_per_gpu
        self._units: Dict[int, _lib.RcGP] = {}            # pool slot -> device handle
        self._unit_output: Dict[int, int] = {}             # pool slot -> the output whose y it holds
        self._unit_signature: Dict[int, Any] = {}          # pool slot -> (output, hyper-parameters) current on the device
        self._cache: Dict[int, Dict[str, np.ndarray]] = {}
        self._is_isotropic = bool(is_isotropic)
        self._live_noise = None            # covariant GP: the full fitted likelihood covariance, from calibrate() until the next rebuild
        super().__init__(name, fold, is_read, is_covariant, is_isotropic, kernel_parameters, likelihood_variance)

    def broadcast_parameters(self, is_covariant: bool, is_isotropic: bool) -> 'GPR':
        self._live_noise = None            # a rebuilt model starts from the stored, diagonalised likelihood variance
        return super().broadcast_parameters(is_covariant, is_isotropic)

    # ---- device plumbing
    @property
    def device(self) -> int:
        if self._device is None:
            import os
            self._device = int(os.environ.get('LOCAL_RANK', 0)) % max(_lib.device_count(), 1)
        return self._device

    UNIT_MEMORY_BUDGET: float = 96.0e9     #: bytes of HBM the pool's N x N work matrices (three per unit) may take together

    @property
    def units_at_once(self) -> int:
        """How many units of this GP's size the GPU is given at once: ``units_per_gpu`` (argument, RCGP_UNITS or by N) within what one
        batched call takes and what fits in memory."""
        if self._is_covariant:
            return 1
        wanted = int(self._units_per_gpu) if self._units_per_gpu is not None else default_units_per_gpu(self._N)
        padded = -(-self._N // 128) * 128
        fits = int(self.UNIT_MEMORY_BUDGET // (3 * 8 * padded * padded))
        return max(1, min(wanted, _lib.MAX_BATCH, fits))

    @property
    def pool_size(self) -> int:
        """P, the number of device handles the independent outputs are dealt to (1 for a covariant GP: one joint system)."""
        return max(1, min(self.units_at_once, self._L))

    def _unit(self, slot: int) -> _lib.RcGP:
        """The device handle of pool slot ``slot`` (created on first use: X uploaded once per unit)."""
        if slot not in self._units:
            if self._is_covariant:
                self._units[slot] = _lib.RcMOGP(self._X, self._Y, device=self.device)
                self._unit_output[slot] = 0
            else:
                first = slot                               # the first output the slot serves
                self._units[slot] = _lib.RcGP(self._X, self._Y[:, first], device=self.device)
                self._unit_output[slot] = first
        return self._units[slot]

    @property
    def handle(self) -> _lib.RcGP:
        """The device handle of pool slot 0 (the only one of a covariant GP, or with ``units_per_gpu = 1``)."""
        return self._unit(0)

    def close(self):
        for unit in self._units.values():
            unit.close()
        self._units.clear()
        self._unit_output.clear()
        self._unit_signature.clear()

    def _hyper(self, l: int) -> Tuple[np.ndarray, float, float]:
        record = self._kernel.implementation[l]
        noise = max(float(self._likelihood.data.frames.variance.np[0, l]), Likelihood.VARIANCE_FLOOR)   # gpr/models.py:341
        return np.broadcast_to(record['lengthscales'], (self._M,)).copy(), record['variance'], noise

    def _hyper_mo(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(lengthscales (L,M), kernel variance (L,L), likelihood variance (L,L)) of the covariant GP as MOGPR.__init__ receives
        them (gpr/models.py:335-338). On CONSTRUCTION the reference reduces any likelihood variance to its diagonal: the shape test
        at gpf/models.py:121 compares a bound method with a tuple, is always true, and band_part(., 0, 0) follows (:122-123). But
        the object it goes on using after ``calibrate`` is the trained MOGPR itself (``self._implementation[0]``, :359-367), whose
        likelihood covariance was trained in full (``'covariance': True``): until the model is rebuilt (broadcast_parameters, i.e.
        construction or a re-read), test / predict / K_inv_Y / Sobol see the full fitted Sigma -- kept here in ``_live_noise``."""
        record = self._kernel.implementation[0]
        lengthscales = np.broadcast_to(np.asarray(record['lengthscales'], dtype=np.float64), (self._L, self._M)).copy()
        if self._live_noise is not None:
            return lengthscales, np.asarray(record['variance'], dtype=np.float64), self._live_noise
        noise = np.broadcast_to(np.asarray(self._likelihood.data.frames.variance.np, dtype=np.float64), (self._L, self._L))
        return lengthscales, np.asarray(record['variance'], dtype=np.float64), np.diag(np.diag(noise))

    def _select_mo(self) -> _lib.RcMOGP:
        """The covariant GP with its stored hyper-parameters current on the device."""
        gp = self.handle
        lengthscales, variance, noise = self._hyper_mo()
        signature = ('mo', lengthscales.tobytes(), variance.tobytes(), noise.tobytes())
        if self._unit_signature.get(0) != signature:
            gp.set_hyper(lengthscales, (variance + variance.T) / 2, noise)
            self._unit_signature[0] = signature
        return gp

    def _load_output(self, slot: int, l: int) -> _lib.RcGP:
        """Pool slot ``slot`` holding the targets of output ``l``."""
        gp = self._unit(slot)
        if self._unit_output[slot] != l:
            gp.set_y(self._Y[:, l])
            self._unit_output[slot] = l
            self._unit_signature.pop(slot, None)
        return gp

    def _select(self, l: int) -> _lib.RcGP:
        """Make output ``l`` with its stored hyper-parameters current on its unit of the pool. Re-sending identical values is skipped
        so the cached Cholesky factor / L^-1 / alpha on the device survive between K_inv_Y, predict and Sobol calls (with one unit
        per output they survive for good)."""
        slot = l % self.pool_size
        gp = self._load_output(slot, l)
        lengthscales, variance, noise = self._hyper(l)
        signature = (l, tuple(lengthscales), variance, noise)
        if self._unit_signature.get(slot) != signature:
            gp.set_hyper(lengthscales, variance, noise)
            self._unit_signature[slot] = signature
        return gp

    def _factor_outputs(self) -> None:
        """With one unit per output: factor (and invert) every output that needs it in ONE batched schedule (``rcgp_factor_batch``) --
        a model read back from disk (``run.gsa``) otherwise pays L factorisations one after the other when its K_inv_Y is first asked
        for (the reference: two Cholesky factorisations per output and GSA kind, gsa/calibrators.py:126-127)."""
        if self._is_covariant or self._L < 2 or self.pool_size < self._L:
            return
        units = [self._select(l) for l in range(self._L)]
        status = _lib.factor_batch(units)
        for l, k in enumerate(status):
            if k > 0:
                raise _lib.NotPositiveDefiniteError(int(k), f'output {l}: matrix is not positive definite: leading minor {int(k)}')

    @property
    def implementation(self) -> Tuple[Any, ...]:
        """One (output index, hyper-parameter record) pair per independent output; the device handles are a pool (``pool_size``)."""
        if self._implementation is None:
            self._cache = {}
            self._implementation = tuple((l, record) for l, record in enumerate(self._kernel.implementation))   # one record if covariant
        return self._implementation

    # ---- the GPR contract
    @property
    def X(self) -> np.ndarray:
        return self._X

    @property
    def Y(self) -> np.ndarray:
        return self._Y

    def calibrate(self, method: str = 'L-BFGS-B', **kwargs) -> Dict[str, Any]:
        """Fit the hyper-parameters of every output by L-BFGS-B on -LML and persist them: likelihood/variance.csv,
        likelihood/log_marginal.csv, kernel/variance.csv, kernel/lengthscales.csv and meta.json with the optimiser result
        (gpr/models.py:345-373)."""
        meta, kernel_options, likelihood_options = self._calibration_options(kwargs)
        gp = self.handle
        if self._is_covariant:
            # one optimisation over the Cholesky-parametrised (L,L) variances (gpr/models.py:359-367, gpf/base.py:32-96)
            lengthscales, variance, noise = self._hyper_mo()
            self._unit_signature.clear()
            fit = fit_lbfgsb_mo(gp, lengthscales, variance, noise, is_isotropic=self._is_isotropic,
                                train_kernel_variance=bool(kernel_options['variance']),
                                train_kernel_covariance=bool(kernel_options['covariance']),
                                train_lengthscales=bool(kernel_options['lengthscales']['covariant']),
                                train_likelihood_variance=bool(likelihood_options['variance']),
                                train_likelihood_covariance=bool(likelihood_options['covariance']), method=method, **meta)
            meta.update({'result': str((fit['result'],)), 'kernel': kernel_options, 'likelihood': likelihood_options})
            self.write_meta(meta)
            self._likelihood.data.replace(variance=fit['noise'], log_marginal=np.atleast_2d(fit['log_marginal']))
            n_ell = 1 if self._is_isotropic else self._M
            self._kernel.data.replace(variance=fit['variance'], lengthscales=fit['lengthscales'][:, :n_ell])
            self._kernel._implementation = None
            self._reset_implementation()
            self._live_noise = np.array(fit['noise'], dtype=np.float64)        # the live model keeps its full Sigma (see _hyper_mo)
            return meta
        # The reference fits output after output (gpr/models.py:360-361). Here as many outputs as the pool has handles are fitted together:
        # their L-BFGS-B runs in lockstep, each round of evaluations one batched schedule on the GPU (gpr/optimize.py::fit_lbfgsb_batch);
        # with more outputs than handles the next output starts on the handle of one whose fit has ended.
        common = self._fit_options(method, meta, kernel_options, likelihood_options)
        if self.pool_size >= self._L:
            units, starts = self._fit_units(range(self._L))
            fits = fit_lbfgsb_batch(units, starts, max_units=self.units_at_once, **common)
        else:
            slots = _PoolSlots(self)
            fits = fit_lbfgsb_batch(None, self._fit_starts(range(self._L)), max_units=self.pool_size, bind=slots.bind, release=slots.release,
                                    M=self._M, **common)
        for fit in fits:
            if isinstance(fit, Exception):                 # as in the reference, a failing output fails the calibration
                raise fit
        return self._store_fits(fits, meta, kernel_options, likelihood_options)

    def _calibration_options(self, kwargs: Dict[str, Any]) -> Tuple[Dict[str, Any], Dict[str, Any], Dict[str, Any]]:
        """(optimiser options, what the kernel trains, what the likelihood trains) for a ``calibrate(**kwargs)`` (gpr/models.py:345-358)."""
        meta = self.read_meta() if self._meta_json.exists() else self.META
        kernel_options = self._kernel.calibrate(**(meta.pop('kernel', {}) | kwargs.pop('kernel', {})))
        likelihood_options = self._likelihood.calibrate(**(meta.pop('likelihood', {}) | kwargs.pop('likelihood', {})))
        meta.update(kwargs)
        meta.pop('result', None)
        return meta, kernel_options, likelihood_options

    def _fit_options(self, method: str, meta, kernel_options, likelihood_options) -> Dict[str, Any]:
        return dict(is_isotropic=self._is_isotropic, train_lengthscales=bool(kernel_options['lengthscales']['variant']),
                    train_variance=bool(kernel_options['variance']), train_noise=bool(likelihood_options['variance']), method=method, **meta)

    def _fit_starts(self, outputs) -> list:
        """The start point of every output in ``outputs``: its stored hyper-parameters."""
        starts = []
        for l in outputs:
            lengthscales, variance, noise = self._hyper(l)
            starts.append({'lengthscales': lengthscales[0] if self._is_isotropic else lengthscales, 'variance': variance, 'noise': noise})
        return starts

    def _fit_units(self, outputs) -> Tuple[list, list]:
        """The pool's device handles loaded with ``outputs`` (at most ``pool_size`` of them, consecutive) and their start points."""
        pool, units = self.pool_size, []
        for l in outputs:
            units.append(self._load_output(l % pool, l))
            self._unit_signature.pop(l % pool, None)
        return units, self._fit_starts(outputs)

    def _store_fits(self, fits, meta, kernel_options, likelihood_options) -> Dict[str, Any]:
        """The side effects of MOGP.calibrate (gpr/models.py:362-372): meta.json with the optimiser results, the four parameter csv files."""
        meta.update({'result': str(tuple(fit['result'] for fit in fits)), 'kernel': kernel_options, 'likelihood': likelihood_options})
        self.write_meta(meta)
        self._likelihood.data.replace(variance=np.array([[fit['noise'] for fit in fits]]),
                                      log_marginal=np.array([[fit['log_marginal'] for fit in fits]]))
        n_ell = 1 if self._is_isotropic else self._M
        self._kernel.data.replace(variance=np.array([[fit['variance'] for fit in fits]]),
                                  lengthscales=np.stack([fit['lengthscales'][:n_ell] for fit in fits]))
        self._kernel._implementation = None
        self._reset_implementation()
        return meta

    @classmethod
    def calibrate_group(cls, gps: 'list[HipGP]', method: str = 'L-BFGS-B', **kwargs) -> list:
        """``calibrate`` for several GPs AT ONCE on one GPU -- the folds of a cross-validation, which the reference fits one after the
        other (user/run.py:60-61): every output of every GP is one unit, the units' L-BFGS-B runs advance in lockstep and each round of
        evaluations is one batched schedule (``rcgp_lml_grad_batch``). Each GP ends with exactly the files ``calibrate`` alone gives it.
        Returns per GP its ``meta`` dict or the exception that stopped it (the other GPs are not affected). GPs that cannot share a
        batch (covariant; more outputs than pool slots; another padded size, M or device) are calibrated on their own, in turn."""
        outcomes: list = [None] * len(gps)
        groups: Dict[Any, list] = {}
        for i, gp in enumerate(gps):
            if gp._is_covariant or gp.pool_size < gp.L:
                try:
                    outcomes[i] = gp.calibrate(method=method, **dict(kwargs))
                except Exception as failure:
                    outcomes[i] = failure
            else:
                groups.setdefault((gp.device, -(-gp.N // 128), gp.M), []).append(i)
        for members in groups.values():
            # every output of every member is a unit with a handle of its own; at most ``units_at_once`` of them are in flight, the
            # next one starting when one has ended (``fit_lbfgsb_batch``)
            at_once = min(gps[i].units_at_once for i in members)
            contexts, units, starts, owner = {}, [], [], []
            for i in members:
                gp = gps[i]
                try:
                    options = gp._calibration_options(dict(kwargs))
                    u, s = gp._fit_units(range(gp.L))
                    common = gp._fit_options(method, *options)
                except Exception as failure:
                    outcomes[i] = failure
                    continue
                contexts[i] = options
                units += u
                starts += [common | start for start in s]     # (per GP: the stored meta.json may differ between folds)
                owner += [i] * len(u)
            fits = fit_lbfgsb_batch(units, starts, max_units=at_once) if units else []
            for i in contexts:
                mine = [fit for fit, o in zip(fits, owner) if o == i]
                failed = [fit for fit in mine if isinstance(fit, Exception)]
                try:
                    if failed:
                        raise failed[0]
                    outcomes[i] = gps[i]._store_fits(mine, *contexts[i])
                except Exception as failure:
                    outcomes[i] = failure
        return outcomes

    def log_marginal_likelihood(self) -> np.ndarray:
        """(L,) log marginal likelihood at the stored hyper-parameters ((1,) for a covariant GP: one joint likelihood)."""
        if self._is_covariant:
            return np.array([self._select_mo().lml()])
        return np.array([self._select(l).lml() for l, _ in self.implementation])

    def predict(self, X: np.ndarray, y_instead_of_f: bool = True) -> Tuple[np.ndarray, np.ndarray]:
        """(mean (o,L), SD (o,L)): predict_y when ``y_instead_of_f`` else predict_f (gpr/models.py:375-384)."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        if self._is_covariant:                                                 # gpr/models.py:377-379, gpf/models.py:84-113
            mean, sd = self._select_mo().predict(X, y_instead_of_f)
            return np.atleast_2d(mean), np.atleast_2d(sd)
        results = [self._select(l).predict(X, y_instead_of_f) for l, _ in self.implementation]
        mean = np.stack([r[0] for r in results], axis=1)
        sd = np.stack([r[1] for r in results], axis=1)
        return np.atleast_2d(mean), np.atleast_2d(sd)

    def predict_gradient(self, x: np.ndarray, y_instead_of_f: bool = True) -> Tuple[np.ndarray, np.ndarray]:
        """The gradient GP dy/dx at the (o, M) inputs ``x`` (gpr/models.py:386-415, independent branch): mean (o, L, M) and
        covariance (o, o, L, M, M). As in the reference: mean = dK^T alpha; cov = -(L^-1 dK)^T (L^-1 dK) with
        k(x_O, x_o) / ell_M^2 added where the two gradient components coincide (M == m); ``y_instead_of_f`` is accepted and,
        exactly as in the reference, has no effect. dK = d k(X, x)/dx is analytic here (tape.jacobian in the reference)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        if self._is_covariant:
            # covariant branch (gpr/models.py:392-405): mean (o, L, M); var (o, L, o, L, M, M), where the first L is the TRAINING
            # output block the Cholesky-solved rows belong to -- the reference's einsum 'LNlOM, LNlom -> OLolMm' sums over N only
            o = x.shape[0]
            lengthscales, variance, _ = self._hyper_mo()
            m_lom, cov = self._select_mo().predict_gradient(x)                  # (l, o, M), (Lb, l, O, M, l', o, m)
            mean = np.transpose(m_lom, (1, 0, 2))
            same_l = np.einsum('BlOMlom->OBolMm', cov)                          # l' = l
            var = -same_l
            u = x[None, :, :] / lengthscales[:, None, :]                        # (L, o, M)
            d = u[:, :, None, None, :] - u[None, None, :, :, :]
            kxx = variance[:, None, :, None] * np.exp(-0.5 * np.einsum('...M,...M->...', d, d))      # kernel(x): (L, O, l, o)
            lam = 1.0 / lengthscales                                            # Lambda (broadcast over the points, :388)
            ddxxkxx = np.einsum('LM,lM,LOlo->OLolM', lam, lam, kxx)             # :399-400
            idx = np.arange(self._M)
            var[..., idx, idx] += ddxxkxx                                       # set_diag(var, diag_part(var) + ddxxkxx) (:406)
            return mean, var
        o = x.shape[0]
        mean = np.empty((o, self._L, self._M))
        var = np.empty((o, o, self._L, self._M, self._M))
        for l, record in self.implementation:
            m_l, cov = self._select(l).predict_gradient(x)                     # (o, M), (o, M, o, M)
            mean[:, l, :] = m_l
            var[:, :, l, :, :] = -np.transpose(cov, (0, 2, 1, 3))              # 'LNOM, LNom -> OoLMm'
            z = x / record['lengthscales']
            r2 = np.sum(z * z, axis=1)[:, None] + np.sum(z * z, axis=1)[None, :] - 2.0 * z @ z.T
            kxx = record['variance'] * np.exp(-0.5 * r2)                        # kernel(x): (o, o)
            lam2 = np.broadcast_to(1.0 / record['lengthscales'] ** 2, (self._M,))
            idx = np.arange(self._M)
            var[:, :, l, idx, idx] += kxx[:, :, None] * lam2[None, None, :]    # set_diag(var, diag_part(var) + ddxxkxx)
        return mean, var

    @property
    def K_cho(self) -> np.ndarray:
        """(L,N,N) lower Cholesky factors of K_l + noise_l I (gpr/models.py:427-439). Copies N^2 doubles per output to the
        host: meant for inspection, the accelerated consumers use the device-resident factor. Covariant: (LN, LN) (:429-431)."""
        if self._is_covariant:
            return self._select_mo().k_cho()
        return np.stack([self._select(l).k_cho() for l, _ in self.implementation])

    @property
    def K_inv_Y(self) -> np.ndarray:
        """(L,1,N): alpha_l = (K_l + noise_l I)^-1 y_l (gpr/models.py:441-444); covariant: the (LN) solve reshaped."""
        if self._is_covariant:
            return self._select_mo().k_inv_y()
        self._factor_outputs()
        return np.stack([self._select(l).k_inv_y() for l, _ in self.implementation])[:, None, :]

    def check_K_inv_Y(self, x: np.ndarray) -> np.ndarray:
        """FOR TESTING: RMS over the o rows of k(x,X) . K_inv_Y - predict(x); ~0 (gpr/models.py:446-463)."""
        predicted = self.predict(x)[0]
        alpha = self.K_inv_Y[:, 0, :]
        if self._is_covariant:                                                 # 'loLN, LiN -> ol' (gpr/models.py:456-457)
            lengthscales, variance, _ = self._hyper_mo()
            u = (np.asarray(x)[None, :, :] / lengthscales[:, None, :])                        # (L, o, M)
            U = (self._X[None, :, :] / lengthscales[:, None, :])                              # (L, N, M)
            d = u[:, :, None, None, :] - U[None, None, :, :, :]
            kernel = variance[:, None, :, None] * np.exp(-0.5 * np.einsum('...M,...M->...', d, d))
            result = np.einsum('loLN,LN->ol', kernel, alpha) - predicted
            return np.sqrt(np.sum(result * result, axis=0) / predicted.shape[0])
        result = np.empty_like(predicted)
        for l, record in self.implementation:
            z = x / record['lengthscales']
            Z = self._X / record['lengthscales']
            r2 = np.sum(z * z, axis=1)[:, None] + np.sum(Z * Z, axis=1)[None, :] - 2.0 * z @ Z.T
            result[:, l] = record['variance'] * np.exp(-0.5 * r2) @ alpha[l]
        result -= predicted
        return np.sqrt(np.sum(result * result, axis=0) / predicted.shape[0])


MOGP = HipGP     #: the name the reference's run scripts import (gpr/models.py:324)
