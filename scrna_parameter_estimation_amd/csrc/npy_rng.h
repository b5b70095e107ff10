// npy_rng.h -- from-scratch restatement of the numpy random path memento's bootstrap calls:
//   np.random.Generator(np.random.PCG64(5)).multinomial(N, pvals, size=B)
// (call sites: /root/reference/memento/bootstrap.py:102-103 and :135-137).
//
// numpy (pinned here: 2.2.6; C sources are NOT in the wheel) implements this as
//   PCG64 = PCG XSL-RR 128/64 (128-bit LCG, multiplier 0x2360ED051FC65DA44385DF649FCCF645);
//   next_double = (next64 >> 11) * 2^-53;
//   multinomial = chain of conditional binomials over the bins, early exit when nothing is left;
//   binomial(n, p): p > 0.5 -> n - binomial(n, 1-p);  n*p <= 30 -> sequential inversion;
//                   otherwise BTPE (Kachitvichyanukul & Schmeiser 1988) as published.
// This file restates those published algorithms so that the integer draws are identical to numpy's,
// which is what makes bit-level replay of the reference bootstrap possible on the GPU.  One lane owns
// one generator stream; everything is fp64 with contraction OFF (numpy's x86-64 build has no FMA).
//
// Usable from host C++ (tests compile it with g++ and compare against numpy) and from HIP device code.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define NPY_HD __host__ __device__ __forceinline__
#else
#define NPY_HD static inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace npyrng {

struct Pcg64 {
  uint64_t s_hi, s_lo;  // 128-bit LCG state
  uint64_t i_hi, i_lo;  // 128-bit increment (odd)
};

NPY_HD uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (uint64_t)(((unsigned __int128)a * (unsigned __int128)b) >> 64);
#endif
}

// state = state * MULT + inc  (mod 2^128); output = rotr64(hi ^ lo, hi >> 58) of the NEW state.
NPY_HD uint64_t pcg64_next64(Pcg64 &g) {
  const uint64_t M_HI = 2549297995355413924ULL, M_LO = 4865540595714422341ULL;
  uint64_t lo = g.s_lo * M_LO;
  uint64_t hi = mulhi64(g.s_lo, M_LO) + g.s_hi * M_LO + g.s_lo * M_HI;
  uint64_t nlo = lo + g.i_lo;
  uint64_t nhi = hi + g.i_hi + (nlo < lo ? 1ULL : 0ULL);
  g.s_lo = nlo;
  g.s_hi = nhi;
  uint64_t x = nhi ^ nlo;
  unsigned rot = (unsigned)(nhi >> 58);
  return (x >> rot) | (x << ((64u - rot) & 63u));
}

NPY_HD double pcg64_next_double(Pcg64 &g) {
  return (double)(pcg64_next64(g) >> 11) * (1.0 / 9007199254740992.0);
}

// ---- binomial: inversion for n*p <= 30 -------------------------------------------------------
template <typename Int>
NPY_HD Int binomial_inversion(Pcg64 &g, Int n, double p) {
  double q = 1.0 - p;
  double qn = exp((double)n * log(q));
  double np_ = (double)n * p;
  double bd = np_ + 10.0 * sqrt(np_ * q + 1);
  Int bound = (Int)((double)n < bd ? (double)n : bd);
  Int X = 0;
  double px = qn;
  double U = pcg64_next_double(g);
  while (U > px) {
    X++;
    if (X > bound) {
      X = 0;
      px = qn;
      U = pcg64_next_double(g);
    } else {
      U -= px;
      px = (((double)(n - X) + 1.0) * p * px) / ((double)X * q);
    }
  }
  return X;
}

NPY_HD double btpe_stirling(double x, double x2) {
  return (13680. - (462. - (132. - (99. - 140. / x2) / x2) / x2) / x2) / x / 166320.;
}

// ---- binomial: BTPE for n*p > 30, p <= 0.5 ----------------------------------------------------
template <typename Int>
NPY_HD Int binomial_btpe(Pcg64 &g, Int n, double p) {
  double r = p < 1.0 - p ? p : 1.0 - p;
  double q = 1.0 - r;
  double fm = (double)n * r + r;
  Int m = (Int)floor(fm);
  double p1 = floor(2.195 * sqrt((double)n * r * q) - 4.6 * q) + 0.5;
  double xm = (double)m + 0.5;
  double xl = xm - p1;
  double xr = xm + p1;
  double c = 0.134 + 20.5 / (15.3 + (double)m);
  double a = (fm - xl) / (fm - xl * r);
  double laml = a * (1.0 + a / 2.0);
  a = (xr - fm) / (xr * q);
  double lamr = a * (1.0 + a / 2.0);
  double p2 = p1 * (1.0 + 2.0 * c);
  double p3 = p2 + c / laml;
  double p4 = p3 + c / lamr;
  double nrq = (double)n * r * q;
  Int y;
  for (;;) {
    double u = pcg64_next_double(g) * p4;
    double v = pcg64_next_double(g);
    if (u <= p1) {
      y = (Int)floor(xm - p1 * v + u);
      break;  // accept (triangular region)
    }
    if (u <= p2) {  // parallelogram
      double x = xl + (u - p1) / c;
      v = v * c + 1.0 - fabs((double)m - x + 0.5) / p1;
      if (v > 1.0) continue;
      y = (Int)floor(x);
    } else if (u <= p3) {  // left exponential tail
      y = (Int)floor(xl + log(v) / laml);
      if (y < 0 || v == 0.0) continue;
      v = v * (u - p2) * laml;
    } else {  // right exponential tail
      y = (Int)floor(xr - log(v) / lamr);
      if (y > n || v == 0.0) continue;
      v = v * (u - p3) * lamr;
    }
    Int k = y > m ? y - m : m - y;
    if (!((k > 20) && ((double)k < nrq / 2.0 - 1))) {
      // explicit evaluation of f(y)/f(m)
      double s = r / q;
      double aa = s * ((double)n + 1.0);
      double F = 1.0;
      if (m < y) {
        for (Int i = m + 1; i <= y; i++) F *= (aa / (double)i - s);
      } else if (m > y) {
        for (Int i = y + 1; i <= m; i++) F /= (aa / (double)i - s);
      }
      if (v > F) continue;
      break;
    }
    // squeeze, then Stirling-corrected comparison
    double kd = (double)k;
    double rho = (kd / nrq) * ((kd * (kd / 3.0 + 0.625) + 0.16666666666666666) / nrq + 0.5);
    double t = -(kd * kd) / (2 * nrq);  // == (double)(-k*k): the exact integer k^2 rounds the same way
    double A = log(v);
    if (A < (t - rho)) break;
    if (A > (t + rho)) continue;
    double x1 = (double)y + 1.0;
    double f1 = (double)m + 1.0;
    double z = (double)(n - m) + 1.0;  // exact: |n - m| < 2^53
    double w = (double)(n - y) + 1.0;
    double x2 = x1 * x1, f2 = f1 * f1, z2 = z * z, w2 = w * w;
    double bound = xm * log(f1 / x1) + ((double)(n - m) + 0.5) * log(z / w) +
                   (double)(y - m) * log(w * r / (x1 * q)) + btpe_stirling(f1, f2) +
                   btpe_stirling(z, z2) + btpe_stirling(x1, x2) + btpe_stirling(w, w2);
    if (A > bound) continue;
    break;
  }
  return y;
}

// ---- hoisted form used by the bootstrap kernel -------------------------------------------------
// In numpy's multinomial chain the success probability of bin k, pix[k]/remaining_p, does not depend on
// the draws (remaining_p only shrinks by the pix of earlier bins until the chain stops), so everything
// that depends on p alone -- the p > 0.5 flip, q = 1 - p and log(q) -- is computed ONCE per bin
// instead of once per replicate.  The arithmetic and its rounding are unchanged.
// log(1 - p) for the p that random_binomial hands to the inversion sampler (after the p > 0.5 flip)
NPY_HD double binomial_lq(double pk) {
  double p = pk <= 0.5 ? pk : 1.0 - pk;
  return log(1.0 - p);
}

template <typename Int>
NPY_HD Int binomial_inversion_pre(Pcg64 &g, Int n, double p, double lq) {
  double q = 1.0 - p;
  double qn = exp((double)n * lq);
  Int bound = -1;  // computed lazily: np + 10*sqrt(np*q+1) >= 10, so X <= min(n, 9) can never exceed it
  Int X = 0;
  double px = qn;
  double U = pcg64_next_double(g);
#ifdef NPY_ABLATE_INV_LOOP  // timing experiments only: wrong results
  return (Int)(U > px);
#endif
  while (U > px) {
    X++;
    bool over = false;
    if (X > 9 || X > n) {
      if (bound < 0) {
        double np_ = (double)n * p;
        double bd = np_ + 10.0 * sqrt(np_ * q + 1);
        bound = (Int)((double)n < bd ? (double)n : bd);
      }
      over = X > bound;
    }
    if (over) {
      X = 0;
      px = qn;
      U = pcg64_next_double(g);
    } else {
      U -= px;
      px = (((double)(n - X) + 1.0) * p * px) / ((double)X * q);
    }
  }
  return X;
}

// binomial(pk, n) with lq = binomial_lq(pk) precomputed; identical draws to binomial(g, pk, n).
template <typename Int>
NPY_HD Int binomial_pre(Pcg64 &g, double pk, double lq, Int n) {
  if (n == 0 || pk == 0.0) return 0;
  bool flip = !(pk <= 0.5);
  double p = flip ? 1.0 - pk : pk;
#ifdef NPY_ABLATE_BTPE  // timing experiments only: wrong results
  Int X = (p * (double)n <= 30.0) ? binomial_inversion_pre<Int>(g, n, p, lq) : (Int)((double)n * p);
#else
  Int X = (p * (double)n <= 30.0) ? binomial_inversion_pre<Int>(g, n, p, lq) : binomial_btpe<Int>(g, n, p);
#endif
  return flip ? n - X : X;
}

template <typename Int>
NPY_HD Int binomial(Pcg64 &g, double p, Int n) {
  if (n == 0 || p == 0.0) return 0;
  if (p <= 0.5) {
    if (p * (double)n <= 30.0) return binomial_inversion<Int>(g, n, p);
    return binomial_btpe<Int>(g, n, p);
  }
  double q = 1.0 - p;
  if (q * (double)n <= 30.0) return n - binomial_inversion<Int>(g, n, q);
  return n - binomial_btpe<Int>(g, n, q);
}

}  // namespace npyrng
