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


def _gev_tail(tail, at, upper):
    """Fit genextreme to ``tail``; return the tail area at ``at`` if the KS gate passes, else None."""
    params = stats.genextreme.fit(tail)
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
