"""csrc/npy_rng.h (the product's PCG64 + binomial + multinomial restatement) compiled for the HOST and
compared draw-for-draw with numpy's Generator(PCG64) -- the third-party implementation the reference
calls (bootstrap.py:102-103).  CPU only."""

import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    out = tmp_path_factory.mktemp("shim") / "libnpyrng_host.so"
    src = os.path.join(ROOT, "tests", "host_shim", "npy_rng_host.cpp")
    inc = os.path.join(ROOT, "scrna_parameter_estimation_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-I", inc, src, "-o", str(out)])
    return ctypes.CDLL(str(out))


@pytest.fixture(scope="module")
def shim_perturbed(tmp_path_factory):
    """The same header with every cheap primitive of the guarded fast paths perturbed by 2-4x the hardware instruction's error."""
    out = tmp_path_factory.mktemp("shim_p") / "libnpyrng_host_p.so"
    src = os.path.join(ROOT, "tests", "host_shim", "npy_rng_host.cpp")
    inc = os.path.join(ROOT, "scrna_parameter_estimation_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-DNPY_HOST_PERTURB", "-fPIC", "-shared", "-I", inc, src, "-o", str(out)])
    return ctypes.CDLL(str(out))


def pcg_state(seed):
    s = np.random.PCG64(seed).state["state"]
    m = (1 << 64) - 1
    return (ctypes.c_uint64 * 4)(s["state"] >> 64, s["state"] & m, s["inc"] >> 64, s["inc"] & m)


def test_pcg64_pins(shim):
    # pins recorded in SURVEY.md section 8c for PCG64(5)
    out = (ctypes.c_uint64 * 4)()
    shim.host_pcg64_raw(pcg_state(5), 4, out)
    assert list(out) == [14849682912918955432, 14903876974979881461, 9506078739185184192, 5272104914398938230]
    ref = np.random.PCG64(5).random_raw(4)
    assert list(out) == list(ref)


@pytest.mark.parametrize("n,p", [(20, 0.1), (50000, 0.9), (7, 0.5), (100000, 0.0002), (100000, 0.03), (300, 0.4),
                                 (12345, 0.5), (1 << 40, 1e-9), (1 << 33, 0.25), (3, 0.99)])
def test_binomial_matches_numpy(shim, n, p):
    cnt = 4000
    out = np.zeros(cnt, dtype=np.int64)
    shim.host_binomial(pcg_state(5), ctypes.c_double(p), ctypes.c_int64(n), cnt, out.ctypes.data_as(ctypes.c_void_p))
    ref = np.random.Generator(np.random.PCG64(5)).binomial(n, p, cnt)
    np.testing.assert_array_equal(out, ref)


def _multi(shim, seed, n, pv, B, fn="host_multinomial"):
    d = len(pv)
    out = np.zeros((B, d), dtype=np.int64)
    pv = np.ascontiguousarray(pv, dtype=np.float64)
    getattr(shim, fn)(pcg_state(seed), ctypes.c_int64(n), pv.ctypes.data_as(ctypes.c_void_p), d, B,
                          out.ctypes.data_as(ctypes.c_void_p))
    return out


def test_multinomial_pins(shim):
    np.testing.assert_array_equal(_multi(shim, 5, 10, [.2, .3, .5], 4), [[3, 4, 3], [2, 2, 6], [0, 3, 7], [2, 1, 7]])
    np.testing.assert_array_equal(_multi(shim, 5, 50000, [.9, .05, .03, .02], 3),
                                  [[44942, 2552, 1477, 1029], [45134, 2385, 1508, 973], [44995, 2439, 1515, 1051]])


def test_multinomial_random_shapes(shim):
    rng = np.random.default_rng(0)
    for trial in range(60):
        d = int(rng.integers(2, 200))
        # scRNA-like multiplicities: a few huge zero-count bins, many small ones
        mult = np.concatenate([rng.integers(1, 5000, size=max(1, d // 4)), rng.integers(1, 40, size=d - max(1, d // 4))])
        rng.shuffle(mult)
        n = int(mult.sum()) if trial % 3 else int(rng.integers(1, 10 ** 6))
        if trial % 7 == 0:
            n = int(rng.integers(10 ** 8, 2 ** 30))   # large N_g: still inside the kernel's int32 range
        pv = mult / mult.sum()
        B = 50
        ref = np.random.Generator(np.random.PCG64(5)).multinomial(n, pv, size=B)
        for fn in ("host_multinomial", "host_multinomial_pre", "host_multinomial_fast", "host_multinomial_capped", "host_multinomial_async", "host_multinomial_async_bf"):
            got = _multi(shim, 5, n, pv, B, fn)
            np.testing.assert_array_equal(got, ref, err_msg=f"{fn} trial {trial} d={d} n={n}")


def test_multinomial_golden_weights(shim, internals_small):
    it = internals_small
    for k in range(int(it["n_picks"])):
        mult = it[f"p{k}_counts"]
        for fn in ("host_multinomial", "host_multinomial_pre", "host_multinomial_fast", "host_multinomial_capped", "host_multinomial_async", "host_multinomial_async_bf"):
            got = _multi(shim, 5, int(it[f"p{k}_n_obs"]), mult / mult.sum(), int(it["num_boot"]), fn)
            np.testing.assert_array_equal(got.T, it[f"p{k}_weights"])


def _fallbacks(shim, reset=True):
    out = (ctypes.c_long * 2)()
    shim.host_fallback_counts(out, int(reset))
    return out[0], out[1]


@pytest.mark.parametrize("n,p", [(20, 0.1), (50000, 0.9), (7, 0.5), (100000, 0.0002), (100000, 0.03), (300, 0.4), (12345, 0.5),
                                 (1 << 30, 1e-9), (1 << 30, 2.5e-8), (3, 0.99), (50000, 1e-12), (48000, 6.2e-4), (60, 0.5), (61, 0.49),
                                 (2000, 0.0149), (25, 0.3), (9, 0.05), (1000000, 2.9e-5)])
def test_guarded_fast_binomial_matches_numpy(shim, n, p):
    """binomial_pre<., FAST=true>: fp32 search loops behind a guard, exact fp64 fallback -- draws identical to numpy's."""
    cnt = 20000
    out = np.zeros(cnt, dtype=np.int64)
    _fallbacks(shim)
    shim.host_binomial_fast(pcg_state(5), ctypes.c_double(p), ctypes.c_int64(n), cnt, out.ctypes.data_as(ctypes.c_void_p))
    ref = np.random.Generator(np.random.PCG64(5)).binomial(n, p, cnt)
    np.testing.assert_array_equal(out, ref)
    inv_fb, f_fb = _fallbacks(shim)
    assert inv_fb + f_fb < 0.02 * cnt + 5, (inv_fb, f_fb)          # the fast path really is the common path


@pytest.mark.parametrize("which", ["plain", "perturbed"])
def test_guarded_fast_inversion_small_n_stress(shim, shim_perturbed, which):
    """The fp32 inversion search on small n, where its recurrence factor (n + 1) s / x - s (one fused multiply-add) cancels towards the end
    of the support: 4,000 random (n <= 150, n p <= 30) x 300 draws, every draw numpy's; the guard (1.5e-4 on either side of every step of the CDF) sends ~1 % to the exact search."""
    lib = shim if which == "plain" else shim_perturbed
    rng = np.random.default_rng(33)
    cnt = 300
    out = np.zeros(cnt, dtype=np.int64)
    _fallbacks(lib)
    for trial in range(4000):
        n = int(rng.integers(1, 151))
        p = float(min(0.5, rng.uniform(0.0, 30.0) / n)) if trial % 4 else float(rng.uniform(0.3, 0.5))
        if p * n > 30.0:
            p = 30.0 / n
        lib.host_binomial_fast(pcg_state(5), ctypes.c_double(p), ctypes.c_int64(n), cnt, out.ctypes.data_as(ctypes.c_void_p))
        ref = np.random.Generator(np.random.PCG64(5)).binomial(n, p, cnt)
        np.testing.assert_array_equal(out, ref, err_msg=f"n={n} p={p!r}")
    inv_fb, f_fb = _fallbacks(lib)
    assert inv_fb < 0.02 * 4000 * cnt and f_fb == 0, (inv_fb, f_fb)


@pytest.mark.parametrize("which", ["plain", "perturbed"])
@pytest.mark.parametrize("fn", ["host_multinomial_async", "host_multinomial_async_bf"])
def test_resumable_samplers_multinomial_stress(shim, shim_perturbed, which, fn):
    """The phase-wise (resumable) form of the samplers that the lane-asynchronous tile kernel runs: C3-like chains and chains
    with long inversion searches (n p up to 30: the search continues over several passes), > 2.5e6 draws, every weight numpy's;
    also with the cheap primitives perturbed; ``host_multinomial_async_bf``: the branch-free forms of the common phases, called on
    every pass whatever the lane's state, as the kernel calls them."""
    lib = shim if which == "plain" else shim_perturbed
    rng = np.random.default_rng(21)
    draws = 0
    _fallbacks(lib)
    for trial in range(40):
        d = int(rng.integers(40, 350))
        heavy = max(3, d // 3)
        mult = np.concatenate([rng.integers(200, 9000, size=heavy // 3 + 1), rng.integers(5, 60, size=heavy),
                               rng.integers(1, 20, size=d - heavy - heavy // 3 - 1)])
        rng.shuffle(mult)
        n = int(mult.sum())
        pv = mult / mult.sum()
        B = 400
        ref = np.random.Generator(np.random.PCG64(5)).multinomial(n, pv, size=B)
        got = _multi(lib, 5, n, pv, B, fn)
        np.testing.assert_array_equal(got, ref, err_msg=f"trial {trial} d={d} n={n}")
        draws += (d - 1) * B
    inv_fb, f_fb = _fallbacks(lib)
    assert draws > 2_500_000 and inv_fb < 1e-2 * draws and f_fb < 5e-3 * draws, (draws, inv_fb, f_fb)


@pytest.mark.parametrize("fn", ["host_multinomial_fast", "host_multinomial_capped"])
def test_guarded_fast_multinomial_stress(shim, fn):
    """C3-like chains (48k-cell groups, 60-350 bins, scRNA-like multiplicities) through the guarded fast paths: > 4e6 binomial
    draws, every weight equal to numpy's; the guards send about 1 draw in 1,000 to the exact arithmetic.  ``host_multinomial_capped``:
    one or two BTPE attempts per call, the call repeated until the draw is complete (the tile kernel's form)."""
    rng = np.random.default_rng(11)
    draws = 0
    _fallbacks(shim)
    for trial in range(50):
        d = int(rng.integers(60, 350))
        heavy = max(3, d // 3)
        mult = np.concatenate([rng.integers(200, 9000, size=heavy // 3 + 1), rng.integers(20, 200, size=heavy),
                               rng.integers(1, 20, size=d - heavy - heavy // 3 - 1)])
        rng.shuffle(mult)
        n = int(mult.sum())
        pv = mult / mult.sum()
        B = 500
        ref = np.random.Generator(np.random.PCG64(5)).multinomial(n, pv, size=B)
        got = _multi(shim, 5, n, pv, B, fn)
        np.testing.assert_array_equal(got, ref, err_msg=f"trial {trial} d={d} n={n}")
        draws += (d - 1) * B
    inv_fb, f_fb = _fallbacks(shim)
    assert draws > 4_000_000
    print(f"\n{draws} draws: inversion fallbacks {inv_fb} ({inv_fb / draws:.2e}), explicit-product fallbacks {f_fb} ({f_fb / draws:.2e})")
    assert inv_fb < 5e-3 * draws and f_fb < 5e-3 * draws


@pytest.mark.parametrize("perturb,lazy", [(False, False), (True, False), (True, True)])
def test_guarded_fast_btpe_never_disagrees_with_exact_btpe(tmp_path, perturb, lazy):
    """binomial_btpe_fast (fp64 set-up through fast reciprocals, fp32 logarithms / explicit product / Stirling bound, every decision
    guarded) vs the exact BTPE on ~6e6 random (n, p, state) over n in [60, 2^31), p in (0, 0.5]: whenever the fast path returns a
    draw, the draw and the generator state after it are identical; it decides > 99 % of the draws.  With ``perturb`` every cheap
    primitive carries 2-4x the error of the hardware instruction it maps to on the GPU; ``lazy`` = the instantiation of the
    one-chain-per-wave kernel (set-up of the rarer branches computed inside them)."""
    exe = tmp_path / "btpe_stress"
    src = os.path.join(ROOT, "tests", "host_shim", "btpe_stress.cpp")
    inc = os.path.join(ROOT, "scrna_parameter_estimation_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-I", inc] + (["-DNPY_HOST_PERTURB"] if perturb else []) +
                          (["-DSTRESS_LAZY"] if lazy else []) + [src, "-o", str(exe)])
    r = subprocess.run([str(exe), "6000000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    words = r.stdout.split()
    assert "mismatches 0" in r.stdout and float(words[words.index("fast-decided") + 2].strip("()")) > 0.99, r.stdout


def test_guarded_fast_multinomial_with_perturbed_primitives(shim_perturbed):
    """The guards, not the host's correctly rounded arithmetic, carry the exactness: with every fp32 / fast-reciprocal primitive
    perturbed the multinomial weights are still numpy's."""
    rng = np.random.default_rng(12)
    for trial in range(25):
        d = int(rng.integers(40, 300))
        mult = np.concatenate([rng.integers(100, 9000, size=d // 3), rng.integers(1, 100, size=d - d // 3)])
        rng.shuffle(mult)
        n = int(mult.sum())
        pv = mult / mult.sum()
        ref = np.random.Generator(np.random.PCG64(5)).multinomial(n, pv, size=300)
        for fn in ("host_multinomial_fast", "host_multinomial_capped"):
            got = _multi(shim_perturbed, 5, n, pv, 300, fn)
            np.testing.assert_array_equal(got, ref, err_msg=f"{fn} trial {trial}")


def test_btpe_squeeze_bounds_hold_for_every_k_below_half_nrq():
    """binomial_btpe_fast settles numpy's EXPLICIT candidates (k <= 20) through the squeeze bounds too, which numpy itself consults only
    for k > 20: t - rho <= log(f(y)/f(m)) <= t + rho must therefore hold for every 1 <= k < nrq/2 - 1 (Kachitvichyanukul & Schmeiser 1988,
    step 5.2; the reference reaches numpy's copy of it through Generator.multinomial, memento/bootstrap.py:103).  Checked in fp64
    against the explicit product on ~4e6 (n, p, k): n p > 30, p in [1e-7, 0.5], n up to 2^31, both sides of the mode."""
    rng = np.random.default_rng(0)
    checked = 0
    worst = np.inf
    for trial in range(40000):
        p = float(np.exp(rng.uniform(np.log(1e-7), np.log(0.5))))
        nmin = int(30 / p) + 2
        n = int(min(2 ** 31 - 2, nmin * np.exp(rng.uniform(0, np.log(2000))))) if trial % 3 else nmin + int(rng.integers(0, 50))
        if n * p <= 30:
            continue
        q = 1 - p
        nrq = n * p * q
        m = int(np.floor(n * p + p))
        kmax = int(min(80, np.ceil(nrq / 2 - 1) - 1))
        if kmax < 1:
            continue
        s_, aa = p / q, (p / q) * (n + 1)
        for sign in (1, -1):
            i = np.arange(m + 1, m + kmax + 1) if sign > 0 else np.arange(m, max(0, m - kmax), -1)
            if sign > 0:
                i = i[i <= n]
            if len(i) == 0:
                continue
            log_f = np.cumsum(np.log(aa / i - s_)) * sign            # log(f(m + sign k) / f(m)), k = 1 .. len(i)
            kf = np.arange(1, len(i) + 1, dtype=np.float64)
            rho = (kf / nrq) * ((kf * (kf / 3.0 + 0.625) + 0.1666666666666) / nrq + 0.5)
            t = -kf * kf / (2 * nrq)
            slack = np.minimum(log_f - (t - rho), (t + rho) - log_f)
            assert (slack >= -1e-13 * (1 + np.abs(log_f))).all(), (n, p, sign, int(np.argmin(slack)) + 1)
            worst = min(worst, float((slack / rho).min()))
            checked += len(i)
    assert checked > 3_000_000
    print(f"\n{checked} (n, p, k): smallest slack {worst:.2e} of rho")
