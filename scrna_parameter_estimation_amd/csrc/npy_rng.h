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
#define NPY_HDM __host__ __device__ __forceinline__
#else
#define NPY_HD static inline
#define NPY_HDM inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#if defined(BOOT_STAMPS) && defined(__HIPCC__)   // diagnostic build: cycle accumulators inside the fast BTPE
#define NPY_ST_PARAM , uint64_t *npy_st
#if defined(__HIP_DEVICE_COMPILE__)
// one asm statement (s_memtime returns out of order with LDS / scalar loads) fenced against instruction scheduling on both sides
#define NPY_CLOCK(t_) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                           __builtin_amdgcn_sched_barrier(0); } while (0)
#define NPY_ST(i) do { uint64_t t_; NPY_CLOCK(t_); npy_st[i] += t_ - npy_st[7]; npy_st[7] = t_; } while (0)
#else
#define NPY_CLOCK(t_) ((t_) = 0)
#define NPY_ST(i) (void)npy_st
#endif
#else
#define NPY_ST_PARAM
#define NPY_ST(i)
#endif
// NPY_KEEP(tk, x, alt): x, written so that the compiler cannot move what is computed from it out of the attempt loop.  ``tk`` is a
// condition that is true on every pass but that only the run time knows (attempt <= n; BTPE runs for n > 60 and gives up after 16
// attempts), ``alt`` any other run-time value.  The guarded fast paths keep their rarely taken branches cheap for the common path
// with it: the reciprocals, Stirling terms etc. that only such a branch needs would otherwise be hoisted in front of the loop and
// computed ahead of every draw, although the loop usually runs once and leaves through the triangular region.  (An empty asm
// statement would do the same but makes the value lane-dependent in the compiler's eyes, which the wave-uniform chain kernel of
// csrc/boot.hip must avoid.)
// Only the one-chain-per-wave kernel asks for it (template parameter LAZY): in a 64-wide tile some lane takes every branch on nearly
// every step, and there hoisting the loop-invariant set-up out of the attempt loop is exactly right.
#define NPY_KEEP(tk, x, alt) ((!LAZY || (tk)) ? (x) : (alt))
#ifndef NPY_NOTE_FALLBACK
#define NPY_NOTE_FALLBACK(which)   // host tests count how often the guarded fast paths defer to the exact arithmetic
#endif

namespace npyrng {

struct Pcg64 {
  uint64_t s_hi, s_lo;  // 128-bit LCG state
  uint64_t i_hi, i_lo;  // 128-bit increment (odd)
  // generator interface the samplers are written against (the wave-cooperative generator of csrc/boot.hip implements it too):
  // mark() / rewind() bracket a guarded fast path that may have to be redone on the same uniforms; reserve(n) promises that
  // n uniforms can be taken after a mark() without invalidating it
  struct Mark { uint64_t hi, lo; };
  NPY_HDM Mark mark() const { return Mark{s_hi, s_lo}; }
  NPY_HDM void rewind(const Mark &m) { s_hi = m.hi; s_lo = m.lo; }
  NPY_HDM void reserve(int) {}
  NPY_HDM int max_attempts() const { return 16; }   // attempts the guarded fast BTPE may make before it must be able to rewind
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
#ifdef NPY_ABLATE_PCG  // timing experiments only (wrong draws): what the 128-bit multiply of the generator costs
  {
    uint64_t x = g.s_lo;
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    g.s_lo = x;
    return x + g.s_hi;
  }
#endif
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
template <typename Int, typename Gen>
NPY_HD Int binomial_inversion(Gen &g, Int n, double p) {
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

template <typename Int, typename Gen>
NPY_HD Int binomial_btpe(Gen &g, Int n, double p) {
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

// ``U`` = the first uniform of this draw, already taken from the stream by the caller (the guarded fast path below looks at it
// first); further uniforms are drawn only by numpy's restart rule.
template <typename Int, typename Gen>
NPY_HD Int binomial_inversion_pre(Gen &g, Int n, double p, double lq, double U) {
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
#elif defined(NPY_HOST_PERTURB)
// host stress tests only: every cheap primitive is given a relative error of the size the hardware instruction may have
// (alternating sign), so that the guards are exercised against worse arithmetic than the host's correctly rounded one
static int npy_pert_flip = 0;
NPY_HD float npy_pert(float x, float rel) { npy_pert_flip ^= 1; return x * (1.0f + (npy_pert_flip ? rel : -rel)); }
NPY_HD float f_rcp(float x) { return npy_pert(1.0f / x, 2.4e-7f); }
NPY_HD float f_exp(float x) { return npy_pert(exp2f(x * 1.44269504088896341f), 4e-7f); }
NPY_HD float f_sqrt(float x) { return npy_pert(sqrtf(x), 2.4e-7f); }
#else
NPY_HD float f_rcp(float x) { return 1.0f / x; }
NPY_HD float f_exp(float x) { return exp2f(x * 1.44269504088896341f); }
NPY_HD float f_sqrt(float x) { return sqrtf(x); }
#endif

#if defined(__HIP_DEVICE_COMPILE__)
NPY_HD double d_rcp(double x) {               // v_rcp_f64 + one Newton step: a few ulp (the IEEE division is ~14 instructions)
  double r = __builtin_amdgcn_rcp(x);
  return r + r * (1.0 - x * r);
}
NPY_HD float f_log(float x) { return __builtin_amdgcn_logf(x) * 0.693147180559945309f; }   // v_log_f32 (log2)
#elif defined(NPY_HOST_PERTURB)
NPY_HD double d_rcp(double x) { npy_pert_flip ^= 1; return (1.0 / x) * (1.0 + (npy_pert_flip ? 1e-15 : -1e-15)); }
NPY_HD float f_log(float x) { npy_pert_flip ^= 1; return logf(x) * (1.0f + (npy_pert_flip ? 4e-7f : -4e-7f)) + (npy_pert_flip ? 1.5e-7f : -1.5e-7f); }
#else
NPY_HD double d_rcp(double x) { return 1.0 / x; }
NPY_HD float f_log(float x) { return logf(x); }
#endif
// log(1 + d) for |d| <= 0.35 with ~2e-7 RELATIVE accuracy (also for tiny d, where log(1 + d) in fp32 would lose everything):
// 2 atanh(d / (2 + d))
NPY_HD float f_log1p_small(float d) {
  float t = d * f_rcp(2.0f + d), t2 = t * t;
  return 2.0f * t * (1.0f + t2 * (0.333333333f + t2 * (0.2f + t2 * (0.142857143f + t2 * 0.111111111f))));
}
// log(1 + d) for any d > -1: the series above for |d| <= 0.35, the fp32 logarithm of 1 + d beyond (absolute error <= 5e-7 |log| + 2e-7:
// callers that multiply it by c add ~1e-6 c to their guard, see binomial_btpe_fast)
#ifndef NPY_BTPE_ANYD
#define NPY_BTPE_ANYD 1       // 0: log1p arguments beyond 0.35 send the draw to the exact redo (round 2)
#endif
NPY_HD float f_log1p_any(float d) {
  if (!NPY_BTPE_ANYD) return f_log1p_small(d);
  return fabsf(d) <= 0.35f ? f_log1p_small(d) : f_log(1.0f + d);
}
NPY_HD float f_stirling_true(float x) {       // Stirling's series itself, 1/(12x) - 1/(360x^3) + 1/(1260x^5) - 1/(1680x^7) + 1/(1188x^9): numpy's
  float ix = f_rcp(x), ix2 = ix * ix;          // btpe_stirling carries 13680 where the series has 13860 = 166320/12 (reproduced there, not here)
  return (13860.0f - (462.0f - (132.0f - (99.0f - 140.0f * ix2) * ix2) * ix2) * ix2) * ix * (1.0f / 166320.0f);
}
NPY_HD float f_stirling(float x) {            // btpe_stirling in fp32: ~1/(12 x), a small correction term
  float ix = f_rcp(x), ix2 = ix * ix;
  return (13680.0f - (462.0f - (132.0f - (99.0f - 140.0f * ix2) * ix2) * ix2) * ix2) * ix * (1.0f / 166320.0f);
}

// Guard of the fp32 inversion search, absolute, on U - CDF(X): G = G0 + GA |n log q| + GX X (+ GS for n < 128), four times this
// bound of the fp32 evaluation's error after X steps:
//   q^n = exp(n log q): the argument (<= 42 in magnitude) is rounded to fp32 and multiplied by log2(e) -> relative 1.5e-7 |arg| + 3e-7;
//   every step multiplies px by a factor of relative error <= 4e-7 (s = p/q through v_rcp_f32, the fused multiply-add; for n < 128 the
//   cancellation in (n + 1) s / x - s adds at most 1e-5 in all, GS) and subtracts it from U with a rounding of <= 6e-8:
//   |error of U_X, of px_X| <= 1e-7 (X + 1) + (1.5e-7 |arg| + 3e-7 + 4e-7 X) * CDF.
// (Round 2 used the constant 1.5e-4 = the bound at X = 60: a search that stops at X <= 3 -- most of them -- was sent to the exact
// fp64 search 30 times more often than its own error warrants, and in a 64-wide tile every such draw costs the whole wave a
// sequential fp64 search with a division per step.)
#ifndef NPY_INV_GUARD
#define NPY_INV_GUARD 1.5e-4f // the constant guard (the bound at X = 60): still used by the resumable forms below (lane_inv_finish, lane_inv_att_bf)
#endif
#ifndef NPY_INV_G0
#define NPY_INV_G0 2.4e-6f
#define NPY_INV_GA 8e-7f
#define NPY_INV_GX 2.4e-6f
#define NPY_INV_GS 1e-5f
#endif
#ifndef NPY_BTPE_LOGF
#define NPY_BTPE_LOGF 1       // binomial_btpe_fast: the explicit-product acceptance test through the closed log form where that is accurate
#endif
#ifndef NPY_F_GUARD
#define NPY_F_GUARD 2e-4f     // relative, on v vs f(y)/f(m): fp32 error of a product of <= 64 factors < 3e-5
#endif

// Inversion search in fp32.  Returns X >= 0 when every decision of the search is outside the guard, -1 otherwise.
// Decisions of numpy's loop: U_x > px_x for x < X and U_X <= px_X, with U_{x+1} = U_x - px_x; the margins of the x < X
// decisions are U_{x+1} >= U_X, so two checks at the end cover them all: U_X > G (for X > 0) and px_X - U_X > G.
#ifndef NPY_INV_NOCAP
#define NPY_INV_NOCAP 1      // 1: the search's bound is checked once, after the search; 0: in every step (round 2)
#endif
// Steps IT .. LAST of the fp32 search as explicitly nested ifs; returns the number of steps taken (= X).  The step number, 1/x and
// (float)x are literals (a loop the compiler may choose not to unroll costs a conversion and a v_rcp_f32 per step: measured 3.4-3.7 s
// against 3.0 s for the C3 launch); X as the ladder's return value rather than an assignment in every step measured 1 % faster.
template <int IT, int LAST>
NPY_HD int32_t inversion_steps(float &Uf, float &px, const float a_s, const float s) {
  if constexpr (IT <= LAST) {
    if (Uf > px) {
      Uf -= px;
      px = px * __builtin_fmaf(a_s, 1.0f / (float)IT, -s);
      return inversion_steps<IT + 1, LAST>(Uf, px, a_s, s);
    }
  }
  return IT - 1;
}

template <typename Int>
NPY_HD int32_t binomial_inversion_fast(double U, Int n, double p, double lq) {
  float nf = (float)n, pf = (float)p;
  float qf = 1.0f - pf;                       // p <= 0.5
  float s = pf * f_rcp(qf);
  const float argf = (float)((double)n * lq); // n*lq >= -1.39 n p >= about -42 for n*p <= 30 (p <= 0.5): inside the fp32 range
  float qn = f_exp(argf);
  float Uf = (float)U, px = qn;
  int32_t X = 0;
#ifdef NPY_ABLATE_INV_LOOP  // timing experiments only: wrong results
  return (int32_t)(Uf > px);
#endif
  // fully unrolled: the step number is a compile-time constant, so 1/X and (float)X are literals and an iteration is five
  // full-rate fp32 instructions (no conversion, no v_rcp_f32); lock-stepped lanes share the trip count anyway.
  // numpy restarts when X exceeds bound = min(n, np + 10 sqrt(npq + 1)) >= min(n, 10): the first nine steps can never reach it, so
  // the bound (a square root) is only worked out by searches that get that far.
  // the factor (n + 1 - x) s / x of the recurrence as one fused multiply-add, (n + 1) s * (1/x) - s (1/x a literal): its
  // cancellation costs at most (n + 1)/(n + 1 - x) ulps of the factor, i.e. something only for n < ~120 near the end of the support,
  // where it adds < 1e-5 to the absolute error of the running sum (NPY_INV_GS in the guard)
  const float a_s = (nf + 1.0f) * s;
#if NPY_INV_NOCAP
  // The search may not pass numpy's bound (nor n): checked ONCE, after the search, not in every step.  Past x = n the factor is zero,
  // then negative, px stays (-)0 and the search runs on to its last step, where X > cap sends the draw to the exact path; it takes a U
  // above the whole fp32 CDF to get there (~1e-6 of the draws with n < 60, none otherwise).
  int32_t cap_end = n < (Int)9 ? (int32_t)n : 9;
  X = inversion_steps<1, 9>(Uf, px, a_s, s);
  if (X == 9 && Uf > px) {
    float npf = nf * pf;
    float capf = npf + 10.0f * f_sqrt(npf * qf + 1.0f) - 1.5f;
    capf = capf < nf ? capf : nf;
    capf = capf < 60.0f ? capf : 60.0f;
    cap_end = (int32_t)capf;
    X = inversion_steps<10, 60>(Uf, px, a_s, s);
  }
  const float G = (NPY_INV_G0 + (n < (Int)128 ? NPY_INV_GS : 0.0f)) + NPY_INV_GA * fabsf(argf) + NPY_INV_GX * (float)X;
  bool ok = (px - Uf > G) && (X == 0 || Uf > G) && X <= cap_end;
  return ok ? X : -1;
#else    // round 2's form: the bound checked in every step (kept for A/B runs)
  const int32_t cap9 = n < (Int)9 ? (int32_t)n : 9;
#pragma unroll
  for (int it = 1; it <= 9; it++) {
    if (!(Uf > px) || it > cap9) break;
    X = it;
    Uf -= px;
    px = px * __builtin_fmaf(a_s, 1.0f / (float)it, -s);
  }
  if (X == 9 && Uf > px) {
    // stay strictly below numpy's bound (and below 60: longer searches are ~6 sigma events for n*p <= 30 and go to the exact path)
    float npf = nf * pf;
    float capf = npf + 10.0f * f_sqrt(npf * qf + 1.0f) - 1.5f;
    capf = capf < nf ? capf : nf;
    capf = capf < 60.0f ? capf : 60.0f;
    // (steps 10 .. 60 stay unrolled with a per-lane ``break`` although every step is then one more level of nested control flow and
    // the compiler keeps the deeper levels' saved exec masks in VGPR lanes -- two v_writelane per step going in, two v_readlane + s_or
    // per level coming out: measured at C3, a rolled loop with one exec mask and v_rcp_f32 for 1/x takes 3.37 s against 3.00 s, an
    // unrolled search without nesting -- stopped lanes keep X, U, px through selects, a wave-uniform branch leaves it -- 3.16 s)
    int32_t cap = (int32_t)capf;
#pragma unroll
    for (int it = 10; it <= 60; it++) {
      if (!(Uf > px) || it > cap) break;
      X = it;
      Uf -= px;
      px = px * __builtin_fmaf(a_s, 1.0f / (float)it, -s);
    }
  }
  const float G = (NPY_INV_G0 + (n < (Int)128 ? NPY_INV_GS : 0.0f)) + NPY_INV_GA * fabsf(argf) + NPY_INV_GX * (float)X;
  bool ok = (px - Uf > G) && (X == 0 || Uf > G);   // also false when the search stopped at a cap
  return ok ? X : -1;
#endif
}

// BTPE's explicit evaluation of f(y)/f(m) against v, in fp32.  +1: v <= F (numpy breaks: accept), 0: v > F (numpy continues:
// reject), -1: inside the guard or too many factors -> the caller evaluates numpy's fp64 product.
template <typename Int>
NPY_HD int btpe_explicit_fast(double v, Int n, Int m, Int y, double r, double q) {
  Int k = y > m ? y - m : m - y;
  if (k > 64 || m > (Int)8000000) return -1;      // (consecutive integers must be exact in fp32 for the running index below)
  float s = (float)r * f_rcp((float)q);           // (the caller passes q through NPY_KEEP: this set-up stays inside the branch)
  float aa = s * ((float)n + 1.0f);
  Int lo = m < y ? m : y;
  // P = prod (aa/i - s) = prod (aa - s i)/i as a quotient of two running products, both scaled by 1/m so that every factor is ~1:
  // four full-rate instructions per factor, no reciprocal inside the loop
  float ic = f_rcp((float)m + 0.5f);
  float aa_c = aa * ic, s_c = s * ic;
  float i_f = (float)lo;
  float Pn = 1.0f, Pd = 1.0f;
  for (Int i = 0; i < k; i++) {
    i_f += 1.0f;
    Pn *= aa_c - s_c * i_f;
    Pd *= i_f * ic;
  }
  float P = Pn * f_rcp(Pd);
  float vf = (float)v;
  // m < y: F = P;  m > y: F = 1/P (P > 0), compare v*P with 1
  float a_ = m <= y ? vf : vf * P;
  float b_ = m <= y ? P : 1.0f;
  float d = a_ - b_;
  float mag = fabsf(a_) > fabsf(b_) ? fabsf(a_) : fabsf(b_);
  if (!(fabsf(d) > NPY_F_GUARD * mag)) return -1;
  return d > 0.0f ? 0 : 1;
}

// BTPE with numpy's decisions but cheaper arithmetic: the set-up quotients through d_rcp (fp64, a few ulp), the square root of the
// set-up in fp32 when the floor() it feeds is not close to a step, the logarithms, the explicit product and the Stirling bound in
// fp32; every comparison and every floor() guarded by a margin several times the worst-case error of the cheaper arithmetic.
// Returns the draw y >= 0, having consumed exactly the uniforms numpy consumes, or -1 when some decision fell inside its guard:
// the caller then rewinds the generator and runs binomial_btpe (numpy's arithmetic).
// What only the rarer branches need (1/c, 1/p1, 1/nrq, the Stirling terms) is computed inside them, behind NPY_KEEP: three draws
// in four are accepted in the triangular region on the first attempt and pay for the set-up of p1 .. p4 only.
// ``cap`` > 0 bounds the attempts of THIS call: -2 is returned when they were all rejected (decided, uniforms consumed) -- the
// caller calls again for the same (n, r) and the draw goes on where it stopped, since the set-up depends on (n, r) only and the
// attempts are independent (the lock-step tile kernel makes one attempt per bin step and lets the lane retry in the next).
template <typename Int, bool LAZY = false, typename Gen>
NPY_HD Int binomial_btpe_fast(Gen &g, Int n, double r, int cap NPY_ST_PARAM) {   // r = p <= 0.5
  const double q = 1.0 - r;
  const double fm = (double)n * r + r;
  const double md = floor(fm);
  const Int m = (Int)md;
  const double nrq = (double)n * r * q;
  // p1 = floor(2.195 sqrt(nrq) - 4.6 q) + 0.5: an integer, decided in numpy's own arithmetic unless the fp32 evaluation of the
  // argument is clear of every integer (its error: ~3e-7 relative from the square root and the products)
  double p1;
  {
    float t = 2.195f * f_sqrt((float)nrq) - 4.6f * (float)q;
    float ft = floorf(t);
    float gt = 2e-6f * t + 1e-4f;
    if (t - ft > gt && ft + 1.0f - t > gt) {
      p1 = (double)ft + 0.5;
    } else {
      p1 = floor(2.195 * sqrt(nrq) - 4.6 * q) + 0.5;
    }
  }
  const double xm = md + 0.5, xl = xm - p1, xr = xm + p1;
  const double c = 0.134 + 20.5 * d_rcp(15.3 + md);
  double a = (fm - xl) * d_rcp(fm - xl * r);
  const double laml = a * (1.0 + a * 0.5);
  a = (xr - fm) * d_rcp(xr * q);
  const double lamr = a * (1.0 + a * 0.5);
  const double rlaml = d_rcp(laml), rlamr = d_rcp(lamr);
  const double p2 = p1 * (1.0 + 2.0 * c);
  const double p3 = p2 + c * rlaml;
  const double p4 = p3 + c * rlamr;
  const double gu = 1e-11 * p4;                                       // set-up values are within ~1e-15 (relative) of numpy's
  NPY_ST(0);
  const int max_att = cap > 0 ? cap : g.max_attempts();
  for (int attempt = 0; attempt < max_att; attempt++) {
    const bool tk = (Int)attempt <= n;   // always true (see NPY_KEEP)
    double u = pcg64_next_double(g) * p4;
    double v = pcg64_next_double(g);
    NPY_ST(1);
    if (fabs(u - p1) < gu || fabs(u - p2) < gu || fabs(u - p3) < gu) return -1;
    if (u <= p1) {                       // triangular region: accepted at once
      double x = xm - p1 * v + u;
      double fx = floor(x);
      double gx = 1e-10 * (fabs(x) + 1.0);
      if (x - fx < gx || fx + 1.0 - x < gx) return -1;
      if (fx < 0.0 || fx > (double)n) return -1;    // cannot happen inside the central regions; be safe
      return (Int)fx;
    }
    double x, gx;
    if (u <= p2) {                       // parallelogram
      const double c_ = NPY_KEEP(tk, c, r), p1_ = NPY_KEEP(tk, p1, r);
      x = xl + (u - p1) * d_rcp(c_);
      v = v * c + 1.0 - fabs(md - x + 0.5) * d_rcp(p1_);
      if (fabs(v - 1.0) < 1e-10) return -1;
      if (v > 1.0) continue;
      gx = 1e-10 * (fabs(x) + 1.0);
    } else {                             // exponential tails: fp32 logarithm
      if (v == 0.0) continue;
      float lv = f_log((float)v);
      bool left = u <= p3;
      double rl = left ? rlaml : rlamr;
      x = left ? xl + (double)lv * rl : xr - (double)lv * rl;
      gx = (2e-6 * fabs((double)lv) + 4e-7) * rl + 1e-10 * (fabs(x) + 1.0);
      v = left ? v * (u - p2) * laml : v * (u - p3) * lamr;
    }
    NPY_ST(2);
    double fx = floor(x);
    if (x - fx < gx || fx + 1.0 - x < gx) return -1;
    if (fx < 0.0 || fx > (double)n) {
      if (u <= p2) return -1;            // cannot happen inside the two central regions; be safe
      continue;                          // numpy: y < 0 (left tail) / y > n (right tail)
    }
    Int y = (Int)fx;
    Int k = y > m ? y - m : m - y;
    NPY_ST(3);
    // numpy accepts through the explicit product F = f(y)/f(m) (k factors) when k <= 20 or k >= nrq/2 - 1, and through the squeeze +
    // Stirling-corrected log form otherwise.  The log form IS log F (up to the truncation of the Stirling series, < 1e-9 for arguments
    // >= 8), so the explicit case can take it too -- no loop of up to 64 factors that every lane of a wave waits for: v <= F iff
    // log v <= bound.  The product is kept for small arguments and large log1p arguments.
    // Before any of that, the squeeze: t - rho <= log(f(y)/f(m)) <= t + rho holds for EVERY k < nrq/2 - 1 (Kachitvichyanukul & Schmeiser's
    // bounds; numpy consults them for k > 20 only because the product is cheap below -- checked on 2e7 (n, r, k), tests/test_npy_rng_host.py).
    // For numpy's squeeze candidates it IS numpy's decision; for its explicit candidates it is a sound shortcut to the same decision
    // (v <= F): nine candidates in ten are settled here, and the logarithms, reciprocals and Stirling terms below are worked out in
    // the bin steps where some lane of the wave is left undecided, not in all of them.
    const bool sq_ok = (double)k < nrq / 2.0 - 1;
    const bool expl = !((k > 20) && sq_ok);
    if (v < 1e-11) {
      if (v > -1e-11) return -1;
      return y;       // numpy: v <= 0 < F (explicit case); log of a negative number is NaN and every comparison fails (squeeze case): accepted
    }
    float A = f_log((float)v);
    if (sq_ok) {
      const float rnrq = f_rcp((float)NPY_KEEP(tk, nrq, r));
      float kf = (float)k;
      float rho = (kf * rnrq) * ((kf * (kf * 0.333333333f + 0.625f) + 0.16666666666666666f) * rnrq + 0.5f);
      float t = -(kf * kf) * 0.5f * rnrq;
      float gs = 1e-5f * (1.0f + fabsf(A)) + 6e-6f * (fabsf(t) + rho);
      float lo_ = t - rho, hi_ = t + rho;
      NPY_ST(5);
      if (A < lo_ - gs) return y;
      if (A > hi_ + gs) continue;
      if (!expl && (A < lo_ + gs || A > hi_ - gs)) return -1;      // numpy's own decision changes at lo / hi: too close to tell
    }
    float yf1 = (float)y + 1.0f;                                        // x1
    float wf = (float)(n - y) + 1.0f;                                   // w
    float d1 = (float)(m - y) * f_rcp(yf1);                             // f1/x1 - 1 = (m - y)/(y + 1)
    float d2 = (float)(y - m) * f_rcp(wf);                              // z/w - 1 = (y - m)/(n - y + 1)
    double num3 = ((double)n + 2.0) * r - ((double)y + 1.0);            // w r - x1 q, without the cancellation
    float d3 = (float)num3 * f_rcp(yf1 * (float)q);                     // w r/(x1 q) - 1
    const bool small_d = !(fabsf(d1) > 0.35f || fabsf(d2) > 0.35f || fabsf(d3) > 0.35f);
    const bool sane_d = d1 > -0.9f && d2 > -0.9f && d3 > -0.9f && d1 < 8.0f && d2 < 8.0f && d3 < 8.0f;   // the fp32 logarithm's range of use
    const Int amin = (y < m ? y : m) < (n - (y > m ? y : m)) ? (y < m ? y : m) : (n - (y > m ? y : m));
    if (expl && !(NPY_BTPE_LOGF && small_d && amin >= (Int)7)) {
      int dec = btpe_explicit_fast<Int>(v, n, m, y, r, NPY_KEEP(tk, q, r));
      NPY_ST(4);
      if (dec < 0) return -1;
      if (dec == 0) continue;
      return y;
    }
    if (!expl && !(NPY_BTPE_ANYD ? sane_d : small_d)) return -1;
    // (a bin with n r of 30-100 and a candidate 21+ away from the mode has |d| > 0.35: the fp32 logarithm then, with its absolute
    // error in the guard -- these were 80 % of the draws that went to the exact redo)
    const float c1 = (float)xm, c2 = (float)(n - m) + 0.5f, c3 = (float)(y - m);
    float T1 = c1 * f_log1p_any(d1);
    float T2 = c2 * f_log1p_any(d2);
    float T3 = c3 * f_log1p_any(d3);
    const float mf1 = (float)NPY_KEEP(tk, m, y) + 1.0f, zf = (float)(n - NPY_KEEP(tk, m, y)) + 1.0f;
    // log(f(y)/f(m)) = T1 + T2 + T3 + st(f1) + st(z) - st(x1) - st(w) (Stirling's series for the four factorials).  numpy's test for
    // k > 20 adds all four correction terms, with 13680 for the series' 13860 (as numpy's source has it): reproduced as it is there; the
    // explicit case compares with the true F, so it gets the true signs and the true series.
    float bound = T1 + T2 + T3 + (expl ? (f_stirling_true(mf1) + f_stirling_true(zf)) - (f_stirling_true(yf1) + f_stirling_true(wf))
                                       : f_stirling(mf1) + f_stirling(zf) + f_stirling(yf1) + f_stirling(wf));
    float gb = 4e-6f * (fabsf(T1) + fabsf(T2) + fabsf(T3)) + 1e-5f * (1.0f + fabsf(A));
    if (!small_d) gb += 1e-6f * ((fabsf(d1) > 0.35f ? c1 : 0.0f) + (fabsf(d2) > 0.35f ? c2 : 0.0f) + (fabsf(d3) > 0.35f ? fabsf(c3) : 0.0f));
    NPY_ST(6);
    if (A > bound + gb) continue;
    if (A < bound - gb) return y;
    return -1;
  }
  return cap > 0 ? (Int)-2 : (Int)-1;
}

// ---- resumable form: one binomial draw as a small state machine ---------------------------------------------------------
// The lane-per-chain tile kernel used to walk the 64 chains of a wave in lock step: every bin step ran the inversion sampler and
// BTPE one after the other, each to completion, for whichever lanes needed it -- so every lane waited for the longest search,
// the largest number of BTPE attempts and every rarely taken branch of its 63 neighbours (in-kernel stamps: a lane spends ~1.5k
// of the ~8k cycles a 64-wide BTPE call takes in code of its own).  Chains are independent, though: nothing but the instruction
// stream ties the lanes of a wave together.  The functions below cut a draw into PHASES -- start (classify, set up), <= 9 steps
// of the inversion search, one BTPE attempt, the explicit product, the squeeze / Stirling test, the exact redo -- so that a kernel
// can run a fixed sequence of phases per pass, each for the lanes that are in it, and let every lane move on to its next bin as
// soon as ITS draw is done (csrc/boot.hip: k_boot1d_async).  The arithmetic, the guards and the fallback rule are those of
// binomial_inversion_fast / binomial_btpe_fast above: the draw and the uniforms it consumes are numpy's in every case.
enum LaneState : int32_t { LS_START = 0, LS_INV, LS_ATT, LS_ATT2, LS_EXPL, LS_SQZ, LS_XINV, LS_XBT, LS_DONE, LS_FINISH, LS_RESTART, LS_IDLE };

struct LaneDraw {
  int32_t n;                 // cells left when the draw starts
  int32_t flip;              // pk > 0.5: the draw is n - binomial(n, 1 - pk)
  int32_t w;                 // the finished draw (after the flip)
  double p;                  // min(pk, 1 - pk)
  // inversion search
  double lq, U;              // log(1 - p); the draw's first uniform (the exact redo starts from it)
  float Uf, px, s, nf1;
  int32_t X, it, cap;
  // BTPE
  double p1, p2, p3, p4, xm, c, laml, lamr, nrq;
  int32_t m, attempts;
  uint64_t mk_hi, mk_lo;     // generator state at the start of the draw (the exact redo rewinds to it)
  double u, v;               // the attempt in flight: its two uniforms (u scaled by p4), then v and y of the pending acceptance test
  int32_t y;
};

// Phase START: classify the draw binomial(n, pk) and set its sampler up.  n > 0.
template <typename Gen>
NPY_HD int32_t lane_begin(LaneDraw &D, Gen &g, double pk, double lq, int32_t n) {
  D.n = n;
  if (pk == 0.0) {
    D.w = 0;
    return LS_DONE;
  }
  D.flip = !(pk <= 0.5);
  const double p = D.flip ? 1.0 - pk : pk;
  D.p = p;
  if (p * (double)n <= 30.0) {
    D.lq = lq;
    D.U = pcg64_next_double(g);
    float nf = (float)n, pf = (float)p;
    float qf = 1.0f - pf;
    D.s = pf * f_rcp(qf);
    D.px = f_exp((float)((double)n * lq));
    D.Uf = (float)D.U;
    D.X = 0;
    D.it = 1;
    D.nf1 = nf + 1.0f;
    D.cap = n < 9 ? n : 9;
    return LS_INV;
  }
  // BTPE set-up, as in binomial_btpe_fast
  const double r = p, q = 1.0 - r;
  const double fm = (double)n * r + r;
  const double md = floor(fm);
  D.m = (int32_t)md;
  const double nrq = (double)n * r * q;
  D.nrq = nrq;
  double p1;
  {
    float t = 2.195f * f_sqrt((float)nrq) - 4.6f * (float)q;
    float ft = floorf(t);
    float gt = 2e-6f * t + 1e-4f;
    if (t - ft > gt && ft + 1.0f - t > gt) {
      p1 = (double)ft + 0.5;
    } else {
      p1 = floor(2.195 * sqrt(nrq) - 4.6 * q) + 0.5;
    }
  }
  const double xm = md + 0.5, xl = xm - p1, xr = xm + p1;
  const double c = 0.134 + 20.5 * d_rcp(15.3 + md);
  double a = (fm - xl) * d_rcp(fm - xl * r);
  const double laml = a * (1.0 + a * 0.5);
  a = (xr - fm) * d_rcp(xr * q);
  const double lamr = a * (1.0 + a * 0.5);
  const double p2 = p1 * (1.0 + 2.0 * c);
  const double p3 = p2 + c * d_rcp(laml);
  D.p1 = p1;
  D.p2 = p2;
  D.p3 = p3;
  D.p4 = p3 + c * d_rcp(lamr);
  D.xm = xm;
  D.c = c;
  D.laml = laml;
  D.lamr = lamr;
  D.attempts = 0;
  typename Gen::Mark mk = g.mark();
  D.mk_hi = mk.hi;
  D.mk_lo = mk.lo;
  return LS_ATT;
}

NPY_HD int32_t lane_inv_finish(LaneDraw &D) {
  bool ok = (D.px - D.Uf > NPY_INV_GUARD) && (D.X == 0 || D.Uf > NPY_INV_GUARD);   // also false when the search stopped at a cap
  if (!ok) return LS_XINV;
  D.w = D.flip ? D.n - D.X : D.X;
  return LS_DONE;
}

// Phase INV: the next segment of the search -- steps 1..9 with literal reciprocals, then eight steps at a time.  Written without
// branches: in a wave some lane nearly always walks the whole segment, so every lane computes all of it (the step factors ahead
// of the running product, which is then one multiply per step) and keeps the values at which ITS search stopped.
NPY_HD int32_t lane_inv(LaneDraw &D) {
  float Uf = D.Uf, px = D.px;
  const float s = D.s, nf1 = D.nf1;
  int32_t X = D.X;
  bool stopped = false;
  if (D.it == 1) {
    const int32_t cap9 = D.cap;
    float f[9];
#pragma unroll
    for (int it = 1; it <= 9; it++) f[it - 1] = ((nf1 - (float)it) * s) * (1.0f / (float)it);
#pragma unroll
    for (int it = 1; it <= 9; it++) {
      stopped = stopped || !(Uf > px) || it > cap9;
      float Un = Uf - px, pn = px * f[it - 1];
      X = stopped ? X : it;
      Uf = stopped ? Uf : Un;
      px = stopped ? px : pn;
    }
    D.Uf = Uf;
    D.px = px;
    D.X = X;
    if (!(X == 9 && Uf > px)) return lane_inv_finish(D);
    // the search goes on: now numpy's bound matters (see binomial_inversion_fast)
    float nf = nf1 - 1.0f, pf = (float)D.p, qf = 1.0f - pf;
    float npf = nf * pf;
    float capf = npf + 10.0f * f_sqrt(npf * qf + 1.0f) - 1.5f;
    capf = capf < nf ? capf : nf;
    capf = capf < 60.0f ? capf : 60.0f;
    D.cap = (int32_t)capf;
    D.it = 10;
    return LS_INV;
  }
  // later segments: eight steps per pass, the step number a run-time value; 1/it through the reciprocal instruction (<= 1 ulp
  // against the correctly rounded literal of the first segment: one more rounding per step, well inside NPY_INV_GUARD)
  const int32_t it0 = D.it;
  const int32_t cap = D.cap;
  float f[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    float itf = (float)(it0 + j);
    f[j] = ((nf1 - itf) * s) * f_rcp(itf);
  }
  int32_t it = it0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    stopped = stopped || !(Uf > px) || it > cap || it > 60;
    float Un = Uf - px, pn = px * f[j];
    X = stopped ? X : it;
    Uf = stopped ? Uf : Un;
    px = stopped ? px : pn;
    it = stopped ? it : it + 1;
  }
  D.Uf = Uf;
  D.px = px;
  D.X = X;
  D.it = it;
  // not stopped after eight steps: go on in the next pass (a ninth test would read step it0 + 8's state, which is what D holds)
  return stopped ? lane_inv_finish(D) : (int32_t)LS_INV;
}

// Phase XINV: the search in numpy's arithmetic, from the draw's first uniform.
template <typename Gen>
NPY_HD int32_t lane_xinv(LaneDraw &D, Gen &g) {
  NPY_NOTE_FALLBACK(0);
  int32_t X = binomial_inversion_pre<int32_t>(g, D.n, D.p, D.lq, D.U);
  D.w = D.flip ? D.n - X : X;
  return LS_DONE;
}

NPY_HD int32_t lane_bt_accept(LaneDraw &D, int32_t y) {
  D.w = D.flip ? D.n - y : y;
  return LS_DONE;
}

// Phase ATT: one BTPE attempt (two uniforms).  Accepted in the triangle -> LS_DONE; rejected -> LS_ATT (the next pass draws
// again); otherwise the acceptance test that the candidate needs: LS_EXPL or LS_SQZ; a decision inside a guard -> LS_XBT.
template <typename Gen>
NPY_HD int32_t lane_att(LaneDraw &D, Gen &g) {
  if (D.attempts >= 16) return LS_XBT;
  D.attempts++;
  const double p1 = D.p1, p2 = D.p2, p3 = D.p3, p4 = D.p4, xm = D.xm;
  const int32_t n = D.n, m = D.m;
  const double gu = 1e-11 * p4;
  double u = pcg64_next_double(g) * p4;
  double v = pcg64_next_double(g);
  if (fabs(u - p1) < gu || fabs(u - p2) < gu || fabs(u - p3) < gu) return LS_XBT;
  if (u <= p1) {                       // triangular region: accepted at once
    double x = xm - p1 * v + u;
    double fx = floor(x);
    double gx = 1e-10 * (fabs(x) + 1.0);
    if (x - fx < gx || fx + 1.0 - x < gx) return LS_XBT;
    if (fx < 0.0 || fx > (double)n) return LS_XBT;
    return lane_bt_accept(D, (int32_t)fx);
  }
  const double xl = xm - p1, xr = xm + p1;
  double x, gx;
  if (u <= p2) {                       // parallelogram
    const double c = D.c;
    x = xl + (u - p1) * d_rcp(c);
    v = v * c + 1.0 - fabs((double)m - x + 0.5) * d_rcp(p1);
    if (fabs(v - 1.0) < 1e-10) return LS_XBT;
    if (v > 1.0) return LS_ATT;
    gx = 1e-10 * (fabs(x) + 1.0);
  } else {                             // exponential tails: fp32 logarithm
    if (v == 0.0) return LS_ATT;
    float lv = f_log((float)v);
    bool left = u <= p3;
    double lam = left ? D.laml : D.lamr;
    double rl = d_rcp(lam);
    x = left ? xl + (double)lv * rl : xr - (double)lv * rl;
    gx = (2e-6 * fabs((double)lv) + 4e-7) * rl + 1e-10 * (fabs(x) + 1.0);
    v = left ? v * (u - p2) * lam : v * (u - p3) * lam;
  }
  double fx = floor(x);
  if (x - fx < gx || fx + 1.0 - x < gx) return LS_XBT;
  if (fx < 0.0 || fx > (double)n) {
    if (u <= p2) return LS_XBT;        // cannot happen inside the two central regions; be safe
    return LS_ATT;                     // numpy: y < 0 (left tail) / y > n (right tail)
  }
  int32_t y = (int32_t)fx;
  int32_t k = y > m ? y - m : m - y;
  D.v = v;
  D.y = y;
  return (!((k > 20) && ((double)k < D.nrq / 2.0 - 1))) ? LS_EXPL : LS_SQZ;
}

// btpe_explicit_fast with the running products split over four independent accumulator pairs: a lone wave issues a dependent
// instruction only every ~9 cycles, so four short chains finish in a third of the time of one long one.  Other rounding order,
// same guard (the error of a product of <= 64 fp32 factors is far below NPY_F_GUARD in any order).
NPY_HD int btpe_explicit_fast4(double v, int32_t n, int32_t m, int32_t y, double r, double q) {
  int32_t k = y > m ? y - m : m - y;
  if (k > 64 || m > 8000000) return -1;
  float s = (float)r * f_rcp((float)q);
  float aa = s * ((float)n + 1.0f);
  int32_t lo = m < y ? m : y;
  float ic = f_rcp((float)m + 0.5f);
  float aa_c = aa * ic, s_c = s * ic;
  float base = (float)lo;
  float Pn0 = 1.0f, Pn1 = 1.0f, Pn2 = 1.0f, Pn3 = 1.0f, Pd0 = 1.0f, Pd1 = 1.0f, Pd2 = 1.0f, Pd3 = 1.0f;
  for (int32_t i = 0; i < k; i += 4) {
    float i0 = base + (float)(i + 1), i1 = i0 + 1.0f, i2 = i0 + 2.0f, i3 = i0 + 3.0f;
    bool h1 = i + 1 < k, h2 = i + 2 < k, h3 = i + 3 < k;
    Pn0 *= aa_c - s_c * i0;
    Pd0 *= i0 * ic;
    Pn1 *= h1 ? aa_c - s_c * i1 : 1.0f;
    Pd1 *= h1 ? i1 * ic : 1.0f;
    Pn2 *= h2 ? aa_c - s_c * i2 : 1.0f;
    Pd2 *= h2 ? i2 * ic : 1.0f;
    Pn3 *= h3 ? aa_c - s_c * i3 : 1.0f;
    Pd3 *= h3 ? i3 * ic : 1.0f;
  }
  float P = ((Pn0 * Pn1) * (Pn2 * Pn3)) * f_rcp((Pd0 * Pd1) * (Pd2 * Pd3));
  float vf = (float)v;
  float a_ = m <= y ? vf : vf * P;
  float b_ = m <= y ? P : 1.0f;
  float d = a_ - b_;
  float mag = fabsf(a_) > fabsf(b_) ? fabsf(a_) : fabsf(b_);
  if (!(fabsf(d) > NPY_F_GUARD * mag)) return -1;
  return d > 0.0f ? 0 : 1;
}

// Phase EXPL: explicit f(y)/f(m) against v.
NPY_HD int32_t lane_expl(LaneDraw &D) {
  int dec = btpe_explicit_fast4(D.v, D.n, D.m, D.y, D.p, 1.0 - D.p);
  if (dec < 0) return LS_XBT;
  if (dec == 0) return LS_ATT;
  return lane_bt_accept(D, D.y);
}

// Phase SQZ: squeeze, then the Stirling-corrected bound.
NPY_HD int32_t lane_sqz(LaneDraw &D) {
  const double v = D.v, r = D.p, q = 1.0 - D.p, nrq = D.nrq, xm = D.xm;
  const int32_t n = D.n, m = D.m, y = D.y;
  const int32_t k = y > m ? y - m : m - y;
  if (v < 1e-11) {
    if (v > -1e-11) return LS_XBT;
    return lane_bt_accept(D, y);       // numpy: log of a negative number is NaN, every comparison fails, the draw is accepted
  }
  const float rnrq = f_rcp((float)nrq);
  float kf = (float)k;
  float rho = (kf * rnrq) * ((kf * (kf * 0.333333333f + 0.625f) + 0.16666666666666666f) * rnrq + 0.5f);
  float t = -(kf * kf) * 0.5f * rnrq;
  float A = f_log((float)v);
  float gs = 1e-5f * (1.0f + fabsf(A)) + 6e-6f * (fabsf(t) + rho);
  float lo_ = t - rho, hi_ = t + rho;
  if (A < lo_ - gs) return lane_bt_accept(D, y);
  if (A > hi_ + gs) return LS_ATT;
  if (A < lo_ + gs || A > hi_ - gs) return LS_XBT;
  float yf1 = (float)y + 1.0f;                                        // x1
  float d1 = (float)(m - y) * f_rcp(yf1);                             // f1/x1 - 1 = (m - y)/(y + 1)
  float wf = (float)(n - y) + 1.0f;                                   // w
  float d2 = (float)(y - m) * f_rcp(wf);                              // z/w - 1 = (y - m)/(n - y + 1)
  double num3 = ((double)n + 2.0) * r - ((double)y + 1.0);            // w r - x1 q, without the cancellation
  float d3 = (float)num3 * f_rcp(yf1 * (float)q);                     // w r/(x1 q) - 1
  if (fabsf(d1) > 0.35f || fabsf(d2) > 0.35f || fabsf(d3) > 0.35f) return LS_XBT;
  float T1 = (float)xm * f_log1p_small(d1);
  float T2 = ((float)(n - m) + 0.5f) * f_log1p_small(d2);
  float T3 = (float)(y - m) * f_log1p_small(d3);
  float bound = T1 + T2 + T3 + f_stirling((float)m + 1.0f) + f_stirling((float)(n - m) + 1.0f) + f_stirling(yf1) + f_stirling(wf);
  float gb = 4e-6f * (fabsf(T1) + fabsf(T2) + fabsf(T3)) + 1e-5f * (1.0f + fabsf(A));
  if (A > bound + gb) return LS_ATT;
  if (A < bound - gb) return lane_bt_accept(D, y);
  return LS_XBT;
}

// ---- the two common phases WITHOUT branches ------------------------------------------------------------------------------
// A wave that runs alone on its SIMD issues a DEPENDENT instruction only every ~9 cycles but independent ones every 4: a pass
// that runs "start", "inversion segment" and "BTPE attempt" one after the other, each behind its own branch, is latency-bound.
// The forms below compute both samplers' next piece for EVERY lane in one straight line (values of lanes that are in another
// state are computed on whatever their fields hold and dropped by selects), so that the compiler can interleave the two
// dependency chains.  Same arithmetic as lane_begin / lane_inv / lane_att up to the order of two multiplications in the
// inversion search (all segments now take 1/it from the reciprocal instruction); the guards and the exact redo are unchanged.
NPY_HD float lane_inv_cap(int32_t n, float pf) {
  // numpy restarts the search when X exceeds bound = min(n, np + 10 sqrt(npq + 1)) >= min(n, 10): stay strictly below it, below 60
  // (longer searches go to the exact path), but never below min(n, 9), which cannot reach the bound
  float nf = (float)n, qf = 1.0f - pf, npf = nf * pf;
  float capf = npf + 10.0f * f_sqrt(npf * qf + 1.0f) - 1.5f;
  float lo = nf < 9.0f ? nf : 9.0f;
  capf = capf > lo ? capf : lo;
  capf = capf < nf ? capf : nf;
  return capf < 60.0f ? capf : 60.0f;
}

template <typename Gen>
NPY_HD int32_t lane_begin_bf(LaneDraw &D, Gen &g, double pk, double lq, int32_t n, bool starting) {
  const bool zero = pk == 0.0;
  const bool flip = !(pk <= 0.5);
  const double p = flip ? 1.0 - pk : pk;
  const bool inv = p * (double)n <= 30.0;
  // inversion set-up (the generator moves on only for lanes that start an inversion draw)
  Gen g2 = g;
  const double U = pcg64_next_double(g2);
  const float nf = (float)n, pf = (float)p;
  const float s_ = pf * f_rcp(1.0f - pf);
  const float px0 = f_exp((float)((double)n * lq));
  const int32_t cap = (int32_t)lane_inv_cap(n, pf);
  // BTPE set-up, as in binomial_btpe_fast (for every lane; only the fallback of the fp32 square root is a branch)
  const double r = p, q = 1.0 - r;
  const double fm = (double)n * r + r;
  const double md = floor(fm);
  const double nrq = (double)n * r * q;
  const float t = 2.195f * f_sqrt((float)nrq) - 4.6f * (float)q;
  const float ft = floorf(t);
  const float gt = 2e-6f * t + 1e-4f;
  double p1 = (double)ft + 0.5;
  if (starting && !zero && !inv && !(t - ft > gt && ft + 1.0f - t > gt)) p1 = floor(2.195 * sqrt(nrq) - 4.6 * q) + 0.5;
  const double xm = md + 0.5, xl = xm - p1, xr = xm + p1;
  const double c = 0.134 + 20.5 * d_rcp(15.3 + md);
  double a = (fm - xl) * d_rcp(fm - xl * r);
  const double laml = a * (1.0 + a * 0.5);
  a = (xr - fm) * d_rcp(xr * q);
  const double lamr = a * (1.0 + a * 0.5);
  const double p2 = p1 * (1.0 + 2.0 * c);
  const double p3 = p2 + c * d_rcp(laml);
  const double p4 = p3 + c * d_rcp(lamr);
  if (!starting) return -1;
  D.n = n;
  D.flip = flip;
  D.p = p;
  if (zero) {
    D.w = 0;
    return LS_DONE;
  }
  if (inv) {
    g = g2;
    D.lq = lq;
    D.U = U;
    D.s = s_;
    D.px = px0;
    D.Uf = (float)U;
    D.X = 0;
    D.it = 1;
    D.nf1 = nf + 1.0f;
    D.cap = cap;
    return LS_INV;
  }
  D.m = (int32_t)md;
  D.nrq = nrq;
  D.p1 = p1;
  D.p2 = p2;
  D.p3 = p3;
  D.p4 = p4;
  D.xm = xm;
  D.c = c;
  D.laml = laml;
  D.lamr = lamr;
  D.attempts = 0;
  typename Gen::Mark mk = g.mark();
  D.mk_hi = mk.hi;
  D.mk_lo = mk.lo;
  return LS_ATT;
}

// nine steps of the search for lanes in LS_INV, one attempt up to the triangular region for lanes in LS_ATT
template <typename Gen>
NPY_HD int32_t lane_inv_att_bf(LaneDraw &D, Gen &g, int32_t state) {
  const bool inI = state == LS_INV, inA = state == LS_ATT;
  // ---- inversion segment
  float Uf = D.Uf, px = D.px;
  const float s = D.s, nf1 = D.nf1;
  int32_t X = D.X, it = D.it;
  const int32_t cap = D.cap;
  float f[9];
#pragma unroll
  for (int j = 0; j < 9; j++) {
    float itf = (float)(D.it + j);
    f[j] = ((nf1 - itf) * s) * f_rcp(itf);
  }
  bool stopped = false;
#pragma unroll
  for (int j = 0; j < 9; j++) {
    stopped = stopped || !(Uf > px) || it > cap || it > 60;
    float Un = Uf - px, pn = px * f[j];
    X = stopped ? X : it;
    Uf = stopped ? Uf : Un;
    px = stopped ? px : pn;
    it = stopped ? it : it + 1;
  }
  const bool okI = (px - Uf > NPY_INV_GUARD) && (X == 0 || Uf > NPY_INV_GUARD);      // false when the search stopped at a cap
  const int32_t sI = !stopped ? (int32_t)LS_INV : (okI ? (int32_t)LS_DONE : (int32_t)LS_XINV);
  // ---- BTPE attempt: two uniforms, region, the triangular region's candidate
  Gen g2 = g;
  const double p1 = D.p1, p2 = D.p2, p3 = D.p3, p4 = D.p4, xm = D.xm;
  const double gu = 1e-11 * p4;
  const double u = pcg64_next_double(g2) * p4;
  const double v = pcg64_next_double(g2);
  const bool near = fabs(u - p1) < gu || fabs(u - p2) < gu || fabs(u - p3) < gu;
  const bool tri = u <= p1;
  const double x = xm - p1 * v + u;
  const double fx = floor(x);
  const double gx = 1e-10 * (fabs(x) + 1.0);
  const bool bad = (x - fx < gx || fx + 1.0 - x < gx) || fx < 0.0 || fx > (double)D.n;
  const bool tired = D.attempts >= 16;
  const int32_t sA = (tired || near) ? (int32_t)LS_XBT : (tri ? (bad ? (int32_t)LS_XBT : (int32_t)LS_DONE) : (int32_t)LS_ATT2);
  const int32_t yA = (tri && !bad) ? (int32_t)fx : 0;
  // ---- commit
  if (inI) {
    D.Uf = Uf;
    D.px = px;
    D.X = X;
    D.it = it;
    if (sI == LS_DONE) D.w = D.flip ? D.n - X : X;
    return sI;
  }
  if (inA) {
    if (!tired) {
      g = g2;
      D.attempts++;
      D.u = u;
      D.v = v;
    }
    if (sA == LS_DONE) D.w = D.flip ? D.n - yA : yA;
    return sA;
  }
  return state;
}

// Phase ATT2: the attempt's candidate outside the triangular region (parallelogram / exponential tails) -> the acceptance test it
// needs (LS_EXPL / LS_SQZ), a rejection (LS_ATT), or LS_XBT.
NPY_HD int32_t lane_att_rest(LaneDraw &D) {
  const double p1 = D.p1, p2 = D.p2, p3 = D.p3, xm = D.xm, u = D.u;
  double v = D.v;
  const int32_t n = D.n, m = D.m;
  const double xl = xm - p1, xr = xm + p1;
  double x, gx;
  if (u <= p2) {                       // parallelogram
    const double c = D.c;
    x = xl + (u - p1) * d_rcp(c);
    v = v * c + 1.0 - fabs((double)m - x + 0.5) * d_rcp(p1);
    if (fabs(v - 1.0) < 1e-10) return LS_XBT;
    if (v > 1.0) return LS_ATT;
    gx = 1e-10 * (fabs(x) + 1.0);
  } else {                             // exponential tails: fp32 logarithm
    if (v == 0.0) return LS_ATT;
    float lv = f_log((float)v);
    bool left = u <= p3;
    double lam = left ? D.laml : D.lamr;
    double rl = d_rcp(lam);
    x = left ? xl + (double)lv * rl : xr - (double)lv * rl;
    gx = (2e-6 * fabs((double)lv) + 4e-7) * rl + 1e-10 * (fabs(x) + 1.0);
    v = left ? v * (u - p2) * lam : v * (u - p3) * lam;
  }
  double fx = floor(x);
  if (x - fx < gx || fx + 1.0 - x < gx) return LS_XBT;
  if (fx < 0.0 || fx > (double)n) {
    if (u <= p2) return LS_XBT;        // cannot happen inside the two central regions; be safe
    return LS_ATT;                     // numpy: y < 0 (left tail) / y > n (right tail)
  }
  int32_t y = (int32_t)fx;
  int32_t k = y > m ? y - m : m - y;
  D.v = v;
  D.y = y;
  return (!((k > 20) && ((double)k < D.nrq / 2.0 - 1))) ? LS_EXPL : LS_SQZ;
}

// Phase XBT: rewind the generator to the start of the draw and run numpy's BTPE.
template <typename Gen>
NPY_HD int32_t lane_xbt(LaneDraw &D, Gen &g) {
  NPY_NOTE_FALLBACK(1);
  typename Gen::Mark mk{D.mk_hi, D.mk_lo};
  g.rewind(mk);
  int32_t y = binomial_btpe<int32_t>(g, D.n, D.p);
  return lane_bt_accept(D, y);
}

// binomial(pk, n) with lq = binomial_lq(pk) precomputed; identical draws to binomial(g, pk, n).  FAST selects the guarded
// fp32 evaluation of the two search loops (same draws, fewer instructions); FAST = false is numpy's arithmetic throughout.
template <typename Int, bool FAST = false, bool LAZY = false, typename Gen>
NPY_HD Int binomial_pre(Gen &g, double pk, double lq, Int n) {
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
    if (FAST) {
      g.reserve(34);                   // the fast path gives up after 16 attempts (2 uniforms each)
      typename Gen::Mark saved = g.mark();
#if defined(BOOT_STAMPS) && defined(__HIPCC__)
      uint64_t npy_dummy[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      X = binomial_btpe_fast<Int, LAZY>(g, n, p, 0, npy_dummy);
#else
      X = binomial_btpe_fast<Int, LAZY>(g, n, p, 0);
#endif
      if (X < 0) {
        NPY_NOTE_FALLBACK(1);
        g.rewind(saved);
        X = binomial_btpe<Int>(g, n, p);
      }
    } else {
      X = binomial_btpe<Int>(g, n, p);
    }
#endif
  }
  return flip ? n - X : X;
}

// binomial_pre<Int, true> with at most ``cap`` BTPE attempts in this call: ``pending`` is set (and the return value is
// meaningless) when they were all rejected; calling again with the same arguments continues the draw.  A draw that falls inside a
// guard in a later call is redone in numpy's arithmetic from the generator's position at THAT call: the rejected attempts before
// it were decided outside every guard, so numpy's loop rejects them too and arrives at the same position with the same set-up.
template <typename Int, typename Gen>
NPY_HD Int binomial_pre_capped(Gen &g, double pk, double lq, Int n, int cap, bool &pending) {
  pending = false;
  if (n == 0 || pk == 0.0) return 0;
  bool flip = !(pk <= 0.5);
  double p = flip ? 1.0 - pk : pk;
  Int X;
  if (p * (double)n <= 30.0) {
    double U = pcg64_next_double(g);
    int32_t xf = binomial_inversion_fast<Int>(U, n, p, lq);
    if (xf < 0) NPY_NOTE_FALLBACK(0);
    X = xf >= 0 ? (Int)xf : binomial_inversion_pre<Int>(g, n, p, lq, U);
  } else {
    g.reserve(34);
    typename Gen::Mark saved = g.mark();
#if defined(BOOT_STAMPS) && defined(__HIPCC__)
    uint64_t npy_dummy[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    X = binomial_btpe_fast<Int, false>(g, n, p, cap, npy_dummy);
#else
    X = binomial_btpe_fast<Int, false>(g, n, p, cap);
#endif
    if (X == (Int)-2) {
      pending = true;
      return 0;
    }
    if (X < 0) {
      NPY_NOTE_FALLBACK(1);
      g.rewind(saved);
      X = binomial_btpe<Int>(g, n, p);
    }
  }
  return flip ? n - X : X;
}

template <typename Int, typename Gen>
NPY_HD Int binomial(Gen &g, double p, Int n) {
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
