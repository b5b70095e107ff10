"""Synthetic transcriptomes + capture sampling on the device: the counterpart of the reference's ``memento/simulate.py``
(SURVEY.md section 8f rank 4), same function names and argument meaning:

  extract_parameters(data, q, min_mean)                       reference simulate.py:13-33
  simulate_transcriptomes(n_cells, means, variances, Nc, norm_cov)              :52-89  norm_cov='indep': independent negative
                                                                                        binomials; None / matrix: Gaussian copula
  capture_sampling(transcriptomes, q, q_sq=None, process='hyper')               :91-115 (hypergeometric or Poisson capture)

Differences, all forced by scale: the reference returns dense cells x genes numpy arrays (80 GB at 1M x 20k); here a
``Transcriptomes`` object only carries the negative-binomial parameters and a seed -- every count is a pure function of
(seed, cell, gene) and is regenerated inside the HIP kernel (csrc/simulate.hip) -- and ``capture_sampling`` returns the captured
counts as an ``engine.DeviceCSR`` resident in HBM (``.to_scipy()`` for small cases), ready for ``setup_memento(device_csr=...)``.
The Gaussian-copula branch (``norm_cov`` None or a genes x genes matrix, simulate.py:70-89) keeps ONE dense array, the correlated
standard-normal scores [gene][cell] in fp32 (Cholesky factor of the correlation matrix, computed on the host, times white noise
from mm_std_normal -- a library GEMM); the kernel turns a score into nbinom.ppf(Phi(score)) and rescales the cell to its size on
the fly.  Random draws are not numpy's: parity is statistical
(tests/test_gpu_simulate.py recovers the simulated moments through the HIP estimators, as the reference's
analysis/simulation/estimator_validation.ipynb does).  ``sequencing_sampling`` is dead code in the reference (:118-128).
"""


import numpy as np
import scipy.stats as stats

from .. import _lib, engine


class Transcriptomes:
    """Lazy cells x genes matrix of NB(mean_g, theta_g) molecule counts (reference: the array simulate_transcriptomes returns)."""

    def __init__(self, n_cells, means, thetas, seed, scores=None, cell_sizes=None):
        """``scores``: device float32 [genes][cells] of correlated standard-normal scores (copula branch; None = independent
        genes); ``cell_sizes``: per-cell molecule totals the copula branch rescales to (None = the raw quantiles)."""
        self.shape = (int(n_cells), int(len(means)))
        self.means = np.asarray(means, dtype=np.float64)
        self.thetas = np.asarray(thetas, dtype=np.float64)
        self.seed = int(seed) & ((1 << 64) - 1)
        self._d_mu, self._d_theta = engine.dev(self.means), engine.dev(self.thetas)
        self._totals = None
        self._scores = scores
        self._d_cell_size = None if cell_sizes is None else engine.dev(np.asarray(cell_sizes, dtype=np.float64))
        self._raw_totals = None
        if scores is not None:
            torch = engine._torch()
            assert tuple(scores.shape) == (self.shape[1], self.shape[0]) and scores.dtype == torch.float32 and scores.is_contiguous()
            self._raw_totals = engine.empty((max(1, self.shape[0]),), torch.int64)
            self._launch(None, 0, 2, 3)

    def _launch(self, d_qs, seed_c, process, mode, totals=None, row_nnz=None, row_ptr=None, idx=None, val=None):
        P = engine.P
        _lib.call("mm_simulate", P(self._d_mu), P(self._d_theta), self.shape[1], self.shape[0], P(d_qs), self.seed,
                  int(seed_c) & ((1 << 64) - 1), int(process), int(mode), P(totals), P(row_nnz), P(row_ptr), P(idx), P(val),
                  P(self._scores), P(self._d_cell_size), P(self._raw_totals), engine._stream())

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


def make_spd_matrix(n_dim, rng=None):
    """A random symmetric positive-definite matrix (what sklearn.datasets.make_spd_matrix builds: U (1 + diag(rand)) V^T from the
    SVD of A^T A, A uniform): the reference's default covariance of the copula (simulate.py:72)."""
    rng = np.random if rng is None else rng
    A = rng.random((n_dim, n_dim))
    U, _, Vt = np.linalg.svd(A.T @ A, full_matrices=False)
    return U @ (1.0 + np.diag(rng.random(n_dim))) @ Vt


def correlated_scores(n_cells, corr, seed):
    """Device float32 [genes][cells]: standard-normal scores with correlation matrix ``corr`` = chol(corr) x white noise."""
    torch = engine._torch()
    G = corr.shape[0]
    L = np.linalg.cholesky(np.asarray(corr, dtype=np.float64))
    noise = engine.empty((G, max(1, n_cells)), torch.float32)
    _lib.call("mm_std_normal", int(seed) & ((1 << 64) - 1), G * max(1, n_cells), engine.P(noise), engine._stream())
    return torch.matmul(engine.dev(L.astype(np.float32)), noise)[:, :n_cells].contiguous()      # plain library GEMM


def simulate_transcriptomes(n_cells, means, variances, Nc=None, norm_cov=None, seed=0):
    """reference simulate.py:52-89.  dispersion = (var - mean) / mean^2, floored at 1e-5, theta = 1 / dispersion.
    ``norm_cov`` a string (the reference passes 'indep'): independent negative binomials, ``Nc`` unused (as in the reference).
    ``norm_cov`` None or a genes x genes covariance: Gaussian copula -- scores ~ N(0, corr(norm_cov)), count = nbinom.ppf(Phi(score)),
    every cell rescaled to a size drawn from ``Nc`` with np.random.choice (the global numpy stream, as the reference does) and
    rounded.  Returns a lazy ``Transcriptomes``."""
    means = np.asarray(means, dtype=np.float64)
    variances = np.asarray(variances, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        dispersions = (variances - means) / means ** 2
    dispersions[~(dispersions > 0)] = 1e-5                                              # simulate.py:63 (< 0; also 0 and 0/0)
    thetas = 1.0 / dispersions
    if isinstance(norm_cov, str):
        return Transcriptomes(n_cells, means, thetas, seed)
    n_genes = len(means)
    cov = make_spd_matrix(n_genes) if norm_cov is None else np.asarray(norm_cov, dtype=np.float64)
    sd = np.sqrt(np.diag(cov))
    scores = correlated_scores(n_cells, cov / np.outer(sd, sd), seed + 0x5C0FE)         # (the copula's mean vector cancels: :80)
    cell_sizes = np.random.choice(np.asarray(Nc), size=n_cells)                         # simulate.py:83
    return Transcriptomes(n_cells, means, thetas, seed, scores=scores, cell_sizes=cell_sizes)


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
