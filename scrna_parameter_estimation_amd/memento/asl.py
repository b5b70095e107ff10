"""Host side of K10/K12: achieved significance level from the device's null statistics.

The HIP kernel mm_contract_stats returns, per test, the observed coefficient, the number of valid null
replicates, the two-sided extreme count and the null mean/std.  What is left for the host
(/root/reference/memento/hypothesis_test.py:57-141):

* ``approx=True``  -> normal tail areas from (mean, std) of the null (:77-83);
* extreme count > 10 -> (c + 1) / (n + 1)  (:90-92);
* otherwise the generalised-extreme-value tail fit (scipy genextreme MLE + KS gate, :94-141) on the
  sorted null of only those tests -- the reference's own scipy calls, run in a process pool.
"""

import warnings

import numpy as np
import scipy.stats as stats

TAIL_SIZES = tuple(range(300, 50, -30))  # N_exec = 300, 270, ..., 60  (hypothesis_test.py:102-116)


# ---- genextreme.fit, the same numbers for a fifth of the time -----------------------------------------------------------------
# scipy's fit is Nelder-Mead (optimize.fmin) on rv_continuous._penalized_nnlf, ~300 evaluations of ~120 us each, most of it spent
# in generic wrappers (_lazywhere, argsreduce, support masks built through np.where).  _fast_nnlf below is the SAME arithmetic in the
# same order with the same primitives (scipy.special.log1p, np.exp, np.sum, np.log) for the one distribution needed here, so every
# evaluation returns the same double, the simplex takes the same path and the fitted parameters are bit-identical -- checked against
# scipy's own fit on the first tail every process sees (a mismatch, e.g. another scipy version, switches the fast path off) and in
# tests/test_cpu_host.py on a few hundred tails.
try:
    import scipy.special as _sc
    from scipy import optimize as _optimize
    from scipy.stats import _continuous_distns as _cd
    from scipy.stats import _distn_infrastructure as _di

    _LOGXMAX, _XMIN = _di._LOGXMAX, _cd._XMIN
    _FAST_FIT = {"state": "unchecked"}          # unchecked -> on | off
except Exception:                               # private names moved: scipy's fit only
    _FAST_FIT = {"state": "off"}


def _fast_nnlf(theta, x):
    """genextreme._penalized_nnlf(theta, x) (scipy/stats/_distn_infrastructure.py, _continuous_distns.py), bit for bit."""
    c, loc, scale = theta[0], theta[1], theta[2]
    if not np.isfinite(c) or scale <= 0:
        return np.inf
    z = (x - loc) / scale
    n_log_scale = len(z) * np.log(scale)
    if c > 0:                                   # support (_get_support): z <= 1 / c
        keep = z <= 1.0 / np.maximum(c, _XMIN)
    elif c < 0:                                 #                         z >= 1 / c
        keep = 1.0 / np.minimum(c, -_XMIN) <= z
    else:
        keep = None
    n_bad = 0
    if keep is not None:
        n_bad = z.size - np.count_nonzero(keep)
        if n_bad > 0:
            z = z[keep]
    if c != 0:                                  # _logpdf / _loglogcdf
        cx = c * z
        logex2 = _sc.log1p(-cx)
        logpex2 = logex2 / c
    else:
        cx = np.zeros_like(z)
        logex2 = _sc.log1p(-cx)
        logpex2 = -z
    logpdf = -np.exp(logpex2) + logpex2 - logex2
    edge = (cx == 1) | (cx == -np.inf)
    if edge.any():
        logpdf = np.where(edge, -np.inf, logpdf)
    if c == 1:
        logpdf = np.where(z == 1, 0.0, logpdf)
    fin = np.isfinite(logpdf)                   # _sum_finite
    nf = np.count_nonzero(fin)
    if nf != logpdf.size:
        n_bad += logpdf.size - nf
        logpdf = logpdf[fin]
    return -np.sum(logpdf) + n_bad * _LOGXMAX * 100 + n_log_scale


def _fast_gev_fit(data):
    """rv_continuous.fit for genextreme with the objective above (same start, same optimizer, same acceptance checks)."""
    data = np.asarray(data)
    if not np.isfinite(data).all():
        raise ValueError("The data contains non-finite values.")
    x0 = stats.genextreme._fitstart(data)
    vals = _optimize.fmin(_fast_nnlf, x0, args=(np.ravel(data),), disp=0)
    obj = _fast_nnlf(vals, data)
    vals = tuple(vals)
    if not (np.isfinite(vals[0]) and vals[2] > 0) or not np.isfinite(obj):
        raise RuntimeError("genextreme fit did not converge to admissible parameters")
    return vals


def gev_fit(tail):
    """stats.genextreme.fit(tail) -- through the lean objective once it has reproduced scipy's answer in this process."""
    if _FAST_FIT["state"] == "on":
        return _fast_gev_fit(tail)
    ref = stats.genextreme.fit(tail)            # (raises what scipy raises)
    if _FAST_FIT["state"] == "unchecked":
        try:
            same = tuple(_fast_gev_fit(tail)) == tuple(ref)
        except Exception:
            same = False
        _FAST_FIT["state"] = "on" if same else "off"
    return ref


def _gev_tail(tail, at, upper):
    """Fit genextreme to ``tail``; return the tail area at ``at`` if the KS gate passes, else None."""
    params = gev_fit(tail)
    if stats.kstest(tail, "genextreme", args=params)[1] > 0.05:
        return stats.genextreme.sf(at, *params) if upper else stats.genextreme.cdf(at, *params)
    return None


def tail_fit_asl(coef_row, extreme_count, centred=True):
    """ASL of one test whose two-sided extreme count is <= 10.  ``coef_row``: coefficient per replicate
    column with column 0 = observed value; NaN marks dropped replicate columns.  ``centred``: the null is
    coef[1:] - coef[0] (resampling == 'bootstrap'), else coef[1:] as it is (hypothesis_test.py:66-70)."""
    row = np.asarray(coef_row, dtype=np.float64)
    row = row[~np.isnan(row)]
    stat = row[0]
    null = row[1:] - stat if centred else row[1:]
    null = null[np.isfinite(null)]
    fallback = (extreme_count + 1) / (null.shape[0] + 1)
    a = abs(stat)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            srt = np.sort(null)
            n = srt.shape[0]
            left = None
            for k in TAIL_SIZES:
                area = _gev_tail(srt[:k], -a, upper=False)
                if area is not None:
                    left = (k / n) * area
                    break
            if left is None:
                return fallback
            for k in TAIL_SIZES:
                area = _gev_tail(srt[-k:], a, upper=True)
                if area is not None:
                    return (k / n) * area + left
            return fallback
        except Exception:
            return fallback


def _tail_job(args):
    return tail_fit_asl(*args)


_POOL = {}


def get_pool(num_cpus):
    """Persistent spawn-context process pool for the tail fits (created once, reused across calls; workers
    import only numpy/scipy -- never the GPU runtime)."""
    pool = _POOL.get(num_cpus)
    if pool is None:
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor

        if not _POOL:
            import atexit

            atexit.register(shutdown_pools)
        pool = _POOL[num_cpus] = ProcessPoolExecutor(max_workers=num_cpus, mp_context=mp.get_context("spawn"))
    return pool


def shutdown_pools():
    """Stop the tail-fit workers (also registered with atexit, so interpreter shutdown is clean)."""
    for k in list(_POOL):
        _POOL.pop(k).shutdown(wait=True, cancel_futures=True)


class _main_hidden:
    """While workers are being started, hide ``__main__``'s file/spec from multiprocessing's spawn preparation, so a user
    script without an ``if __name__ == '__main__'`` guard is NOT re-executed in every worker (the workers only need this
    module, which they import by name; re-running a script that opens the GPU would also multiply GPU processes)."""

    def __enter__(self):
        import sys

        self.main = sys.modules.get("__main__")
        self.saved = {}
        for a in ("__file__", "__spec__"):
            if self.main is not None and hasattr(self.main, a):
                self.saved[a] = getattr(self.main, a)
        if "__spec__" in self.saved:
            self.main.__spec__ = None
        if "__file__" in self.saved:
            del self.main.__file__

    def __exit__(self, *exc):
        for a, v in self.saved.items():
            setattr(self.main, a, v)
        return False


def asl_from_stats(stats_arr, approx, fetch_rows, num_cpus=1, resampling='bootstrap', defer=False):
    """Vector of ASLs for all tests.

    ``stats_arr`` [n_tests][8] = {coef0, se, n_valid, extreme_count, null_mean, all_equal, extreme_count_raw, range}
    (include/memento_hip.h); ``fetch_rows(idx)`` returns the coefficient rows (host, [len(idx)][B+1]) of the tests that
    need a tail fit.  ``resampling``: 'bootstrap' centres the null on the observed value, anything else does not
    (hypothesis_test.py:66-70) -- the replicates themselves are the same bootstrap either way.
    ``defer``: return a function that yields the vector instead -- the tail fits are already running in the worker pool,
    so the caller can enqueue more device work (and more fits) before collecting them.
    """
    st = np.asarray(stats_arr)
    coef0, se, n, c, nmean, alleq = st[:, 0], st[:, 1], st[:, 2], st[:, 3], st[:, 4], st[:, 5]
    centred = resampling == 'bootstrap'
    if not centred:
        c, nmean = st[:, 6], nmean + coef0
    asl = np.full(st.shape[0], np.nan)
    live = np.isfinite(coef0) & (alleq == 0) & (n > 0)
    if approx:
        a = np.abs(coef0[live])
        with np.errstate(invalid="ignore", divide="ignore"):
            asl[live] = stats.norm.sf(a, nmean[live], se[live]) + stats.norm.cdf(-a, nmean[live], se[live])
        return (lambda: asl) if defer else asl
    big = live & (c > 10)
    asl[big] = (c[big] + 1) / (n[big] + 1)
    need = np.flatnonzero(live & (c <= 10))
    it = None
    if len(need):
        rows = fetch_rows(need)
        jobs = [(rows[i], float(c[t]), centred) for i, t in enumerate(need)]
        if num_cpus and num_cpus > 1 and (len(jobs) > 1 or defer):       # (a lone deferred fit still overlaps the caller's device work)
            with _main_hidden():        # workers are spawned on submit
                it = get_pool(num_cpus).map(_tail_job, jobs, chunksize=max(1, len(jobs) // (4 * num_cpus)))
        else:
            it = (tail_fit_asl(*j) for j in jobs)

    def finish():
        if it is not None:
            asl[need] = list(it)
        return asl

    return finish if defer else finish()
