"""Synthetic transcriptomes + capture sampling on the device: the counterpart of the reference's ``memento/simulate.py``
(SURVEY.md section 8f rank 4), same function names and argument meaning:

  extract_parameters(data, q, min_mean)                       reference simulate.py:13-33
  simulate_transcriptomes(n_cells, means, variances, Nc, norm_cov='indep')      :52-68  (independent negative binomials)
  capture_sampling(transcriptomes, q, q_sq=None, process='hyper')               :91-115 (hypergeometric or Poisson capture)

Differences, all forced by scale: the reference returns dense cells x genes numpy arrays (80 GB at 1M x 20k); here a
``Transcriptomes`` object only carries the negative-binomial parameters and a seed -- every count is a pure function of
(seed, cell, gene) and is regenerated inside the HIP kernel (csrc/simulate.hip) -- and ``capture_sampling`` returns the captured
counts as an ``engine.DeviceCSR`` resident in HBM (``.to_scipy()`` for small cases), ready for ``setup_memento(device_csr=...)``.
The Gaussian-copula branch of simulate_transcriptomes (``norm_cov`` None or a matrix, simulate.py:70-89: a dense
genes x genes covariance and scipy's nbinom.ppf) is not built.  Random draws are not numpy's: parity is statistical
(tests/test_gpu_simulate.py recovers the simulated moments through the HIP estimators, as the reference's
analysis/simulation/estimator_validation.ipynb does).  ``sequencing_sampling`` is dead code in the reference (:118-128).
"""


import numpy as np
import scipy.stats as stats

from .. import _lib, engine


class Transcriptomes:
    """Lazy cells x genes matrix of NB(mean_g, theta_g) molecule counts (reference: the array simulate_transcriptomes returns)."""

    def __init__(self, n_cells, means, thetas, seed):
        self.shape = (int(n_cells), int(len(means)))
        self.means = np.asarray(means, dtype=np.float64)
        self.thetas = np.asarray(thetas, dtype=np.float64)
        self.seed = int(seed) & ((1 << 64) - 1)
        self._d_mu, self._d_theta = engine.dev(self.means), engine.dev(self.thetas)
        self._totals = None

    def _launch(self, d_qs, seed_c, process, mode, totals=None, row_nnz=None, row_ptr=None, idx=None, val=None):
        P = engine.P
        _lib.call("mm_simulate", P(self._d_mu), P(self._d_theta), self.shape[1], self.shape[0], P(d_qs), self.seed,
                  int(seed_c) & ((1 << 64) - 1), int(process), int(mode), P(totals), P(row_nnz), P(row_ptr), P(idx), P(val), engine._stream())

    def totals(self):
        """Molecules per cell (device int64 tensor), computed once."""
        if self._totals is None:
            torch = engine._torch()
            self._totals = engine.empty((max(1, self.shape[0]),), torch.int64)
            self._launch(None, 0, 2, 0, totals=self._totals)
        return self._totals

    def _csr(self, d_qs, seed_c, process):
        torch = engine._torch()
        n = self.shape[0]
        totals = self.totals() if process == 0 else None
        row_nnz = engine.empty((max(1, n),), torch.int64)
        self._launch(d_qs, seed_c, process, 1, totals=totals, row_nnz=row_nnz)
        indptr = engine.zeros((n + 1,), torch.int64)
        if n:
            indptr[1:] = torch.cumsum(row_nnz[:n], 0)
        nnz = int(indptr[-1].item())
        idx = engine.empty((max(1, nnz),), torch.int32)
        val = engine.empty((max(1, nnz),), torch.float32)
        self._launch(d_qs, seed_c, process, 2, totals=totals, row_ptr=indptr, idx=idx, val=val)
        return engine.DeviceCSR.from_device(indptr, idx[:nnz], val[:nnz], self.shape)

    def to_device_csr(self):
        """The transcriptome counts themselves (no capture) as a device CSR."""
        return self._csr(None, 0, 2)


def device_csr_to_scipy(csr):
    import scipy.sparse as sp

    return sp.csr_matrix((engine.host(csr.data), engine.host(csr.indices), engine.host(csr.indptr)), shape=csr.shape)


def gamma_params_from_moments(m, v):
    """reference simulate.py:36-38."""
    return m ** 2 / v, v / m


def convert_params_nb(mu, theta):
    """Mean / dispersion -> scipy's (n, p) negative-binomial parameters (reference simulate.py:41-50)."""
    r = theta
    var = mu + 1 / r * mu ** 2
    p = (var - mu) / var
    return r, 1 - p


def extract_parameters(data, q=0.1, min_mean=0.001):
    """Parameters of a real dataset (reference simulate.py:13-33; ``data``: scipy CSR or engine.DeviceCSR): relative moments
    (x_mean, x_var) with the raw row sums as size factors (_estimate_size_factor(total=True), estimator.py:64-69), absolute
    moments (z_mean, z_var), per-cell molecule counts Nc = row sum / q, and the indices of genes with plain mean > min_mean.
    Row sums and the moment sums are kernel launches (mm_csr_rowsum, K0/K1)."""
    csr = data if isinstance(data, engine.DeviceCSR) else engine.DeviceCSR(data)
    n = csr.shape[0]
    rowsum = csr.rowsum()
    blocks = engine.CountBlocks(csr, np.zeros(n, dtype=np.int32), 1)
    with np.errstate(divide="ignore"):
        S, sumx, _ = blocks.moments(1.0 / rowsum)
    x_mean = S[0, 0] / n
    x_var = S[1, 0] / n - (1 - q) * S[2, 0] / n - x_mean ** 2                           # estimator.py:179-183
    good_idx = np.where(sumx[0].astype(np.float64) / n > min_mean)[0]
    Nc = rowsum / q
    z_mean = x_mean * Nc.mean()
    z_var = (x_var + x_mean ** 2) * (Nc ** 2).mean() - x_mean ** 2 * Nc.mean() ** 2
    return (x_mean[good_idx], x_var[good_idx]), (z_mean[good_idx], z_var[good_idx]), Nc, good_idx


def simulate_transcriptomes(n_cells, means, variances, Nc=None, norm_cov='indep', seed=0):
    """Independent negative-binomial transcriptomes (reference simulate.py:52-68): dispersion = (var - mean) / mean^2, floored at
    1e-5, theta = 1 / dispersion.  ``Nc`` is unused by this branch in the reference as well.  Returns a lazy ``Transcriptomes``."""
    if not isinstance(norm_cov, str):
        raise NotImplementedError("HIP path: only the independent-gene branch (norm_cov='indep', simulate.py:66-68); the "
                                  "Gaussian-copula branch needs a dense genes x genes covariance")
    means = np.asarray(means, dtype=np.float64)
    variances = np.asarray(variances, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        dispersions = (variances - means) / means ** 2
    dispersions[~(dispersions > 0)] = 1e-5                                              # simulate.py:63 (< 0; also 0 and 0/0)
    return Transcriptomes(n_cells, means, 1.0 / dispersions, seed)


def capture_sampling(transcriptomes, q, q_sq=None, process='hyper', seed=42343):
    """Capture a fraction of every cell's molecules (reference simulate.py:91-115).  ``q_sq`` None: every cell has capture
    rate ``q``; otherwise per-cell rates ~ Beta with mean q and second moment q_sq (drawn on the host with scipy, N values).
    process='hyper': multivariate hypergeometric draw of round(q_c * total_c) molecules; anything else: Poisson(q_c * z).
    Returns (qs host array, captured counts as engine.DeviceCSR)."""
    n = transcriptomes.shape[0]
    if q_sq is None:
        qs = np.ones(n) * q
    else:
        m = q
        v = q_sq - q ** 2
        alpha = m * (m * (1 - m) / v - 1)
        beta = (1 - m) * (m * (1 - m) / v - 1)
        qs = stats.beta.rvs(alpha, beta, size=n)
    csr = transcriptomes._csr(engine.dev(qs), seed, 0 if process == 'hyper' else 1)
    return qs, csr
