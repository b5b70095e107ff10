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

#ifndef NPY_NOTE_FALLBACK
#define NPY_NOTE_FALLBACK(which)   // host tests count how often the guarded fast paths defer to the exact arithmetic
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
NPY_HD int btpe_explicit_fast(double v, Int n, Int m, Int y, double r, double q);

template <typename Int, bool FAST = false>
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
      if (FAST) {
        int dec = btpe_explicit_fast<Int>(v, n, m, y, r, q);
        if (dec == 0) continue;
        if (dec == 1) break;
        NPY_NOTE_FALLBACK(1);
      }
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

// ``U`` = the first uniform of this draw, already taken from the stream by the caller (the guarded fast path below looks at it
// first); further uniforms are drawn only by numpy's restart rule.
template <typename Int>
NPY_HD Int binomial_inversion_pre(Pcg64 &g, Int n, double p, double lq, double U) {
  double q = 1.0 - p;
  double qn = exp((double)n * lq);
  Int bound = -1;  // computed lazily: np + 10*sqrt(np*q+1) >= 10, so X <= min(n, 9) can never exceed it
  Int X = 0;
  double px = qn;
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

// ---- guarded single-precision fast paths -------------------------------------------------------------------------------
// The two data-dependent loops of the samplers -- the inversion search (one fp64 division per step) and BTPE's explicit
// f(y)/f(m) product (one fp64 division per factor) -- only feed COMPARISONS (U > px, v > F).  They are evaluated here in
// fp32 (2-cycle VALU ops, native v_rcp_f32 / v_exp_f32 instead of ~14-instruction fp64 division and ~70-instruction exp
// sequences) and the result is used only when every comparison it decided is farther from its threshold than a guard that is
// >= 6x the worst-case fp32 error; otherwise the caller runs numpy's exact fp64 arithmetic on the same uniforms.  So the integer
// draw -- and with it the number of uniforms consumed -- is numpy's in every case; the guard only decides which arithmetic
// computed it (about 1 draw in 1,000 falls back).  Host tests compare millions of draws with numpy (tests/test_npy_rng_host.py).
#if defined(__HIP_DEVICE_COMPILE__)
NPY_HD float f_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
NPY_HD float f_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
NPY_HD float f_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
#else
NPY_HD float f_rcp(float x) { return 1.0f / x; }
NPY_HD float f_exp(float x) { return exp2f(x * 1.44269504088896341f); }
NPY_HD float f_sqrt(float x) { return sqrtf(x); }
#endif

#ifndef NPY_INV_GUARD
#define NPY_INV_GUARD 1.5e-4f // absolute, on U - CDF.  fp32 error of exp + recurrence + running subtraction: worst-case bound 4e-5 for X <= 60, largest seen in 2e7 random draws 8e-6
#endif
#ifndef NPY_F_GUARD
#define NPY_F_GUARD 2e-4f     // relative, on v vs f(y)/f(m): fp32 error of a product of <= 64 factors < 3e-5
#endif

// Inversion search in fp32.  Returns X >= 0 when every decision of the search is outside the guard, -1 otherwise.
// Decisions of numpy's loop: U_x > px_x for x < X and U_X <= px_X, with U_{x+1} = U_x - px_x; the margins of the x < X
// decisions are U_{x+1} >= U_X, so two checks at the end cover them all: U_X > G (for X > 0) and px_X - U_X > G.
template <typename Int>
NPY_HD int32_t binomial_inversion_fast(double U, Int n, double p, double lq) {
  float nf = (float)n, pf = (float)p;
  float qf = 1.0f - pf;                       // p <= 0.5
  float s = pf * f_rcp(qf);
  float qn = f_exp((float)((double)n * lq));  // n*lq >= -1.39 n p >= about -42 for n*p <= 30 (p <= 0.5): inside the fp32 range
  float npf = nf * pf;
  // numpy restarts when X exceeds bound = min(n, np + 10 sqrt(npq + 1)); stay strictly below it (and below 60: longer searches
  // are ~6 sigma events for n*p <= 30 and go to the exact path)
  float capf = npf + 10.0f * f_sqrt(npf * qf + 1.0f) - 1.5f;
  capf = capf < nf ? capf : nf;
  capf = capf < 60.0f ? capf : 60.0f;
  int32_t cap = (int32_t)capf;
  float Uf = (float)U, px = qn;
  int32_t X = 0;
  while (Uf > px) {
    X++;
    if (X > cap) return -1;
    Uf -= px;
    px = px * ((nf - (float)X + 1.0f) * s) * f_rcp((float)X);
  }
  bool ok = (px - Uf > NPY_INV_GUARD) && (X == 0 || Uf > NPY_INV_GUARD);
  return ok ? X : -1;
}

// BTPE's explicit evaluation of f(y)/f(m) against v, in fp32.  +1: v <= F (numpy breaks: accept), 0: v > F (numpy continues:
// reject), -1: inside the guard or too many factors -> the caller evaluates numpy's fp64 product.
template <typename Int>
NPY_HD int btpe_explicit_fast(double v, Int n, Int m, Int y, double r, double q) {
  Int k = y > m ? y - m : m - y;
  if (k > 64) return -1;
  float s = (float)r * f_rcp((float)q);
  float aa = s * ((float)n + 1.0f);
  Int lo = m < y ? m : y;
  float P = 1.0f;
  for (Int i = lo + 1; i <= lo + k; i++) P *= (aa * f_rcp((float)i) - s);   // every factor is ~ (1 - r)/q +- small: no cancellation
  float vf = (float)v;
  // m < y: F = P;  m > y: F = 1/P (P > 0), compare v*P with 1
  float a_ = m <= y ? vf : vf * P;
  float b_ = m <= y ? P : 1.0f;
  float d = a_ - b_;
  float mag = fabsf(a_) > fabsf(b_) ? fabsf(a_) : fabsf(b_);
  if (!(fabsf(d) > NPY_F_GUARD * mag)) return -1;
  return d > 0.0f ? 0 : 1;
}

// binomial(pk, n) with lq = binomial_lq(pk) precomputed; identical draws to binomial(g, pk, n).  FAST selects the guarded
// fp32 evaluation of the two search loops (same draws, fewer instructions); FAST = false is numpy's arithmetic throughout.
template <typename Int, bool FAST = false>
NPY_HD Int binomial_pre(Pcg64 &g, double pk, double lq, Int n) {
  if (n == 0 || pk == 0.0) return 0;
  bool flip = !(pk <= 0.5);
  double p = flip ? 1.0 - pk : pk;
  Int X;
  if (p * (double)n <= 30.0) {
    double U = pcg64_next_double(g);
    int32_t xf = FAST ? binomial_inversion_fast<Int>(U, n, p, lq) : -1;
    if (FAST && xf < 0) NPY_NOTE_FALLBACK(0);
    X = xf >= 0 ? (Int)xf : binomial_inversion_pre<Int>(g, n, p, lq, U);
  } else {
#ifdef NPY_ABLATE_BTPE  // timing experiments only: wrong results
    X = (Int)((double)n * p);
#else
    X = binomial_btpe<Int, FAST>(g, n, p);
#endif
  }
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
