// boot.hip -- K6+K7+K8: the unique-value bootstrap, replayed draw-for-draw against numpy.
//
// Reference behaviour replaced:
//   bootstrap._bootstrap_1d            memento/bootstrap.py:97-110
//     gen = Generator(PCG64(5)); w = gen.multinomial(N_g, counts/counts.sum(), size=B).T
//   estimator._hyper_1d_relative tuple branch   memento/estimator.py:171-174, :182-183
//   estimator._residual_variance + hypothesis_test._fill + np.log
//                                      memento/estimator.py:103-111, hypothesis_test.py:23-33, :186-197
//
// One lane = one (gene, group) pair = one sequential PCG64 stream (the reference re-seeds PCG64(5) per
// pair).  The 64 pairs of a tile walk their bins in lock step so every operand load is a coalesced
// 512-B row; the multinomial weights never leave registers (the K x B matrix is never materialised).
// fp64, contraction OFF (-ffp-contract=off): replicate means/variances are bit-identical to numpy's.
#include "mm_common.h"
#include "npy_rng.h"

#ifndef BOOT_MIN_WAVES
#define BOOT_MIN_WAVES 2
#endif
#ifndef BOOT_SETPRIO
#define BOOT_SETPRIO 2
#endif
// MINW = waves per SIMD the register budget is set for: 2 when every tile is resident (<= 2048 tiles, the pairing order below
// assumes two per SIMD), 3 in the many-tile regime where a third resident wave adds a little issue throughput.
template <int MINW, bool FAST>
__global__ __launch_bounds__(256, MINW) void k_boot1d_replay(const double *__restrict__ pk_, const double *__restrict__ lq_,
                                                       const double *__restrict__ v, const double *__restrict__ a,
                                                       const double *__restrict__ b,
                                                       const int64_t *__restrict__ tile_ptr, int64_t n_tiles,
                                                       const int32_t *__restrict__ slot_K, const double *__restrict__ slot_nobs,
                                                       const double *__restrict__ slot_omq, const int64_t *__restrict__ slot_row, uint64_t st0, uint64_t st1,
                                                       uint64_t st2, uint64_t st3, int32_t num_boot, int32_t mean_only,
                                                       int64_t ld, double *__restrict__ out_mean, double *__restrict__ out_var,
                                                       int32_t *__restrict__ w_dump, int32_t kmax_dump,
                                                       int64_t *__restrict__ wave_clock) {
  int lane = mm_lane();
  int64_t tile = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (tile >= n_tiles) return;
  int64_t t_start = wave_clock ? (int64_t)wall_clock64() : 0;
  // Two waves share a SIMD.  The host puts the long tiles in the first half of the grid (engine.pair_tiles) and pairs
  // each with a short one from the second half: the long tile is the critical path, so it is served first.
  if (tile * 2 < n_tiles) __builtin_amdgcn_s_setprio(BOOT_SETPRIO);
  int64_t slot = tile * 64 + lane;
  int K = slot_K[slot];
  int64_t row = slot_row[slot];
  if (K <= 0 || row < 0) K = 0;  // unused lane
  int64_t row0 = tile_ptr[tile];
  int kmax = (int)(tile_ptr[tile + 1] - row0);
  double nobs = slot_nobs[slot];
  double omq = slot_omq[slot];  // 1 - q of the pair's group
  int32_t n = (int32_t)nobs;  // N_g < 2^31 (checked by the host)
  double *om = out_mean + row * ld + 1;
  double *ov = out_var + row * ld + 1;
  if (K == 1) {  // bootstrap.py:97-98: a single bin -> all-NaN replicates
    for (int r = 0; r < num_boot; r++) {
      om[r] = NAN;
      ov[r] = NAN;
    }
  }
  npyrng::Pcg64 g{st0, st1, st2, st3};
  const bool run = K >= 2;
#ifdef BOOT_STAMPS
  uint64_t stamp_inv = 0, stamp_btpe = 0, stamp_t0 = __builtin_amdgcn_s_memtime();
  uint64_t stamp_fastcall = 0;                       // wave time inside the fast BTPE call (the rest of stamp_btpe is the exact redo)
  uint64_t cnt_bt = 0, cnt_fb = 0;                   // BTPE draws of this lane / of them redone in the exact arithmetic
  uint64_t stamp_bt[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // inside the fast BTPE: set-up | uniforms | regions | floor + k | explicit product | squeeze | Stirling
#endif
  // operands of the NEXT bin step are loaded while the current one computes (one lane = one latency-bound
  // sequential chain, so an exposed L2/HBM round trip per step would be a large part of the step)
  const int64_t obase = row0 * 64 + lane;
  double c_pk = pk_[obase], c_lq = lq_[obase], c_v = v[obase], c_a = a[obase], c_b = b[obase];
  for (int r = 0; r < num_boot; r++) {
    double M1 = 0.0, M2 = 0.0;
    int32_t dn = n;
    bool live = true;
    for (int k = 0; k < kmax; k++) {
      int kn = k + 1 < kmax ? k + 1 : 0;
      int64_t on = obase + (int64_t)kn * 64;
      double n_pk = pk_[on], n_lq = lq_[on], n_v = v[on], n_a = a[on], n_b = b[on];
      if (run && k < K) {
        int32_t w;
        if (k < K - 1) {
          w = 0;
          if (live) {
#ifdef BOOT_STAMPS  // diagnostic build only (tools/replay_stamps.sh): where a wave-step spends its cycles.  Same draws.
            {
              uint64_t s0, s1, s2;
              NPY_CLOCK(s0);
              bool flip = !(c_pk <= 0.5);
              double p = flip ? 1.0 - c_pk : c_pk;
              int32_t X = 0;
              bool zero = (dn == 0 || c_pk == 0.0), inv = !zero && (p * (double)dn <= 30.0);
              if (inv) {
                double U = npyrng::pcg64_next_double(g);
                int32_t xf = FAST ? npyrng::binomial_inversion_fast<int32_t>(U, dn, p, c_lq) : -1;
                X = xf >= 0 ? xf : npyrng::binomial_inversion_pre<int32_t>(g, dn, p, c_lq, U);
              }
              NPY_CLOCK(s1);
              npyrng::Pcg64 saved = g;
              bool bt = !zero && !inv;
              if (bt) {
                NPY_CLOCK(stamp_bt[7]);
                X = FAST ? npyrng::binomial_btpe_fast<int32_t>(g, dn, p, stamp_bt) : -1;
                cnt_bt++;
              }
              uint64_t s15;
              NPY_CLOCK(s15);
              stamp_fastcall += s15 - s1;
              if (bt && X < 0) {
                cnt_fb++;
                g = saved;
                X = npyrng::binomial_btpe<int32_t>(g, dn, p);
              }
              NPY_CLOCK(s2);
              stamp_inv += s1 - s0;
              stamp_btpe += s2 - s1;
              w = zero ? 0 : (flip ? dn - X : X);
            }
#else
            w = npyrng::binomial_pre<int32_t, FAST>(g, c_pk, c_lq, dn);
#endif
            dn -= w;
            if (dn <= 0) live = false;
          }
        } else {
          w = dn > 0 ? dn : 0;
        }
        if (w_dump) w_dump[((int64_t)slot * kmax_dump + k) * num_boot + r] = (int32_t)w;
        if (w != 0) {
          double wd = (double)w;
          M1 += (c_v * wd) * c_a;
          M2 += ((c_v * c_v) * wd) * c_b - ((omq * c_v) * wd) * c_b;
        }
      }
      c_pk = n_pk; c_lq = n_lq; c_v = n_v; c_a = n_a; c_b = n_b;
    }
    if (run) {
      double mean = M1 / nobs;
      double var = M2 / nobs - mean * mean;
      if (mean_only) {  // estimator._mean_only_1p (estimator.py:188-204): [mean + 1, 10]
        mean = mean + 1;
        var = 10.0;
      }
      om[r] = mean;
      ov[r] = var;
    }
  }
#ifdef BOOT_STAMPS
  // the inner stamps are accumulated by every lane while it is active in that code; the busiest lane's sum is the (lower bound of
  // the) wave's time there
  for (int i = 0; i < 7; i++)
    for (int off = 32; off > 0; off >>= 1) {
      uint64_t o = (uint64_t)__shfl_xor((long long)stamp_bt[i], off, 64);
      stamp_bt[i] = o > stamp_bt[i] ? o : stamp_bt[i];
    }
  for (int off = 32; off > 0; off >>= 1) {
    cnt_bt += (uint64_t)__shfl_xor((long long)cnt_bt, off, 64);
    cnt_fb += (uint64_t)__shfl_xor((long long)cnt_fb, off, 64);
  }
  if (wave_clock && lane == 0) wave_clock[(n_tiles + tile) * 8 + 7] = (int64_t)((cnt_fb << 40) | cnt_bt);
  if (wave_clock && lane == 0) {  // shader-clock cycles: total, inside the inversion sampler, inside BTPE (wave_clock slots 2, 3 reused)
    wave_clock[tile * 4 + 0] = t_start;
    wave_clock[tile * 4 + 1] = (int64_t)(__builtin_amdgcn_s_memtime() - stamp_t0);
    wave_clock[tile * 4 + 2] = (int64_t)stamp_inv;
    wave_clock[tile * 4 + 3] = (int64_t)stamp_btpe;
    for (int i = 0; i < 6; i++) wave_clock[(n_tiles + tile) * 8 + i] = (int64_t)stamp_bt[i];   // second half of the debug buffer
    wave_clock[(n_tiles + tile) * 8 + 6] = (int64_t)stamp_fastcall;
    return;
  }
#endif
  if (wave_clock && lane == 0) {  // profiling hook (mm_debug_wave_clock): when and where this wave ran
    wave_clock[tile * 4 + 0] = t_start;
    wave_clock[tile * 4 + 1] = (int64_t)wall_clock64();
    wave_clock[tile * 4 + 2] = (int64_t)__builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID: se / cu / simd / wave slot
    wave_clock[tile * 4 + 3] = (int64_t)__builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
  }
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t x) {  // splitmix64 finaliser (counter-based fill RNG)
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// One wave per row.  Pass 1: res_var, validity, counts.  Pass 2 (fill_mode 0): each invalid entry takes
// a uniformly chosen VALID replicate (stored negated so later readers still see it as "not original").
// Pass 3: log.
__global__ __launch_bounds__(256) void k_boot_fill_log(double *__restrict__ mean, double *__restrict__ var, int64_t n_rows,
                                                       int64_t ld, int32_t num_boot, double f0, double f1, double f2,
                                                       int32_t fill_mode, uint64_t seed, int32_t *__restrict__ n_invalid) {
  int lane = mm_lane();
  int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= n_rows) return;
  double *m = mean + row * ld + 1;
  double *s = var + row * ld + 1;
  int bad_m = 0, bad_v = 0;
  for (int r = lane; r < num_boot; r += 64) {
    double mm = m[r], vv = s[r];
    double rv = NAN;
    if (mm > 0.0 && vv > 0.0) {
      double lm = log(mm);
      double pred = ((0.0 * lm + f0) * lm + f1) * lm + f2;  // np.poly1d Horner order
      rv = exp(log(vv) - pred);
    }
    if (!(mm > 0.0)) {
      m[r] = NAN;
      bad_m++;
    }
    if (!(rv > 0.0)) {
      rv = NAN;
      bad_v++;
    }
    s[r] = rv;
  }
  for (int off = 32; off > 0; off >>= 1) {
    bad_m += __shfl_xor(bad_m, off, 64);
    bad_v += __shfl_xor(bad_v, off, 64);
  }
  bool none_m = bad_m == num_boot, none_v = bad_v == num_boot;
  if (lane == 0) {
    n_invalid[row * 2] = none_m ? -1 : bad_m;
    n_invalid[row * 2 + 1] = none_v ? -1 : bad_v;
  }
  __threadfence_block();
  if (fill_mode == 0) {
    for (int which = 0; which < 2; which++) {
      double *x = which ? s : m;
      int nb = which ? bad_v : bad_m;
      if (nb == 0 || nb == num_boot) continue;
      for (int r = lane; r < num_boot; r += 64) {
        double cur = x[r];
        if (!(cur > 0.0) && !(cur < 0.0)) {  // NaN => invalid and not yet filled
          uint64_t ctr = mix64(seed ^ mix64((uint64_t)row * 2 + which) ^ ((uint64_t)r << 20));
          double pick = NAN;
          for (int attempt = 0; attempt < 4096; attempt++) {
            ctr = mix64(ctr + attempt);
            int idx = (int)(ctr % (uint64_t)num_boot);
            double c = x[idx];
            if (c > 0.0) {
              pick = c;
              break;
            }
          }
          x[r] = -pick;
        }
      }
      __threadfence_block();
    }
  }
  for (int r = lane; r < num_boot; r += 64) {
    double mm = m[r], vv = s[r];
    m[r] = log(fabs(mm));  // NaN stays NaN; filled entries were stored negated
    s[r] = log(fabs(vv));
  }
}

// ------------------------------------------------------------------------------------------------
// FAST mode (rng='fast'): the same multinomial chain and replicate moments, but one lane = one REPLICATE and
// one wave = 64 replicates of ONE (gene, group) pair.  Every lane of a wave walks the same bins, so the
// operands are wave-uniform, lanes take the same sampler (inversion or BTPE) at each step, and pairs x
// replicates give the chip millions of independent chains.  Each (pair, replicate) owns a PCG64 stream
// derived from (seed, pair, replicate) -- NOT numpy's single stream: results are statistically equivalent
// to the reference (same algorithm, different random numbers), not draw-for-draw identical.
__device__ __forceinline__ uint64_t mix64b(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void k_boot1d_fast(const double *__restrict__ pk_, const double *__restrict__ lq_,
                                                     const double *__restrict__ v, const double *__restrict__ a,
                                                     const double *__restrict__ b, const int64_t *__restrict__ tile_ptr,
                                                     int64_t n_slots, const int32_t *__restrict__ slot_K,
                                                     const double *__restrict__ slot_nobs, const double *__restrict__ slot_omq,
                                                     const int64_t *__restrict__ slot_row, uint64_t seed, int32_t num_boot,
                                                     int32_t mean_only, int32_t chunks, int64_t ld,
                                                     double *__restrict__ out_mean, double *__restrict__ out_var) {
  int lane = mm_lane();
  int64_t wid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int64_t slot = wid / chunks;
  int chunk = (int)(wid % chunks);
  if (slot >= n_slots) return;
  int K = slot_K[slot];
  int64_t row = slot_row[slot];
  if (K < 2 || row < 0) return;  // K == 1 rows stay NaN (bootstrap.py:97-98)
  int r = chunk * 64 + lane;
  bool mine = r < num_boot;
  int64_t obase = tile_ptr[slot >> 6] * 64 + (slot & 63);
  double nobs = slot_nobs[slot], omq = slot_omq[slot];
  int32_t n = (int32_t)nobs;
  uint64_t h = mix64b(seed ^ mix64b((uint64_t)row * 0x100000001B3ull + (uint64_t)r));
  npyrng::Pcg64 g{mix64b(h), mix64b(h + 1), mix64b(h + 2), mix64b(h + 3) | 1ull};
  double M1 = 0.0, M2 = 0.0;
  int32_t dn = n;
  for (int k = 0; k < K; k++) {
    int64_t o = obase + (int64_t)k * 64;   // wave-uniform address: one broadcast load
    int32_t w;
    if (k < K - 1) {
      w = dn > 0 ? npyrng::binomial_pre<int32_t, true>(g, pk_[o], lq_[o], dn) : 0;
      dn -= w;
    } else {
      w = dn > 0 ? dn : 0;
    }
    double wd = (double)w, vv = v[o], bb = b[o];
    M1 += (vv * wd) * a[o];
    M2 += ((vv * vv) * wd) * bb - ((omq * vv) * wd) * bb;
  }
  if (mine) {
    double mean = M1 / nobs;
    double var = M2 / nobs - mean * mean;
    if (mean_only) {
      mean = mean + 1;
      var = 10.0;
    }
    out_mean[row * ld + 1 + r] = mean;
    out_var[row * ld + 1 + r] = var;
  }
}

// ------------------------------------------------------------------------------------------------
// 2D replay: same multinomial chain over the (x_i, x_j, sf_bin) bins of a gene pair; per replicate the
// covariance and the two variances (bootstrap.py:141-155, estimator.py:214-218, :171-174) are folded
// into the correlation exactly as estimator._corr_from_cov does (:281-292: 5.0 sentinel where a variance
// is <= 0, then clip to [-1, 1]).  Writes corr_b to out[row*ld + 1 + b].
template <int MINW, bool FAST>
__global__ __launch_bounds__(256, MINW) void k_boot2d_replay(const double *__restrict__ pk_, const double *__restrict__ lq_,
                                                       const double *__restrict__ v1_, const double *__restrict__ v2_,
                                                       const double *__restrict__ a, const double *__restrict__ b,
                                                       const int64_t *__restrict__ tile_ptr, int64_t n_tiles,
                                                       const int32_t *__restrict__ slot_K, const double *__restrict__ slot_nobs,
                                                       const double *__restrict__ slot_omq, const int64_t *__restrict__ slot_row,
                                                       uint64_t st0, uint64_t st1, uint64_t st2, uint64_t st3, int32_t num_boot,
                                                       int64_t ld, double *__restrict__ out_corr) {
  int lane = mm_lane();
  int64_t tile = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (tile >= n_tiles) return;
  if (tile * 2 < n_tiles) __builtin_amdgcn_s_setprio(BOOT_SETPRIO);  // long tiles first, see k_boot1d_replay
  int64_t slot = tile * 64 + lane;
  int K = slot_K[slot];
  int64_t row = slot_row[slot];
  if (K <= 0 || row < 0) K = 0;
  int64_t row0 = tile_ptr[tile];
  int kmax = (int)(tile_ptr[tile + 1] - row0);
  double nobs = slot_nobs[slot];
  double omq = slot_omq[slot];
  int32_t n = (int32_t)nobs;
  double *oc = out_corr + row * ld + 1;
  npyrng::Pcg64 g{st0, st1, st2, st3};
  const bool run = K >= 1;
  // operands of the NEXT bin step are loaded while the current one computes (as in k_boot1d_replay): the longest pair runs
  // alone in its wave and is the critical path of the launch, so an exposed L2 round trip per step is paid in full there
  const int64_t obase = row0 * 64 + lane;
  double c_pk = pk_[obase], c_lq = lq_[obase], c_x1 = v1_[obase], c_x2 = v2_[obase], c_a = a[obase], c_b = b[obase];
  for (int r = 0; r < num_boot; r++) {
    double A1 = 0.0, A2 = 0.0, MX = 0.0, Q1 = 0.0, Q2 = 0.0;
    int32_t dn = n;
    bool live = true;
    for (int k = 0; k < kmax; k++) {
      int kn = k + 1 < kmax ? k + 1 : 0;
      int64_t on = obase + (int64_t)kn * 64;
      double n_pk = pk_[on], n_lq = lq_[on], n_x1 = v1_[on], n_x2 = v2_[on], n_a = a[on], n_b = b[on];
      if (run && k < K) {
        int32_t w;
        if (k < K - 1) {
          w = 0;
          if (live) {
            w = npyrng::binomial_pre<int32_t, FAST>(g, c_pk, c_lq, dn);
            dn -= w;
            if (dn <= 0) live = false;
          }
        } else {
          w = dn > 0 ? dn : 0;
        }
        if (w != 0) {
          double wd = (double)w, x1 = c_x1, x2 = c_x2, aa = c_a, bb = c_b;
          A1 += (x1 * wd) * aa;
          A2 += (x2 * wd) * aa;
          MX += ((x1 * x2) * wd) * bb;
          Q1 += ((x1 * x1) * wd) * bb - ((omq * x1) * wd) * bb;
          Q2 += ((x2 * x2) * wd) * bb - ((omq * x2) * wd) * bb;
        }
      }
      c_pk = n_pk; c_lq = n_lq; c_x1 = n_x1; c_x2 = n_x2; c_a = n_a; c_b = n_b;
    }
    if (run) {
      double m1 = A1 / nobs, m2 = A2 / nobs;
      double cov = MX / nobs - m1 * m2;
      double var1 = Q1 / nobs - m1 * m1;
      double var2 = Q2 / nobs - m2 * m2;
      double corr = 5.0;
      if (var1 > 0.0 && var2 > 0.0) {
        double vp = sqrt(var1 * var2);
        if (isfinite(vp)) corr = cov / vp;
      }
      if (corr > 1.0) corr = 1.0;
      if (corr < -1.0) corr = -1.0;
      oc[r] = corr;
    }
  }
}

static int64_t *g_wave_clock = nullptr;  // set by mm_debug_wave_clock; nullptr = no profiling writes
static int g_exact_arith = 0;            // set by mm_debug_replay_arith: 1 = numpy's fp64 arithmetic in every search loop (A/B timing, tests)

extern "C" {

int mm_debug_wave_clock(int64_t *d_buf) {
  g_wave_clock = d_buf;
  return MM_OK;
}

int mm_debug_replay_arith(int32_t exact) {
  g_exact_arith = exact ? 1 : 0;
  return MM_OK;
}

int mm_boot1d_replay(const double *d_pk, const double *d_lq, const double *d_v, const double *d_a, const double *d_b,
                     const int64_t *d_tile_ptr, int64_t n_tiles, const int32_t *d_slot_K, const double *d_slot_nobs,
                     const double *d_slot_omq, const int64_t *d_slot_row, const uint64_t pcg_state[4], int32_t num_boot,
                     int32_t mean_only, int64_t ld, double *d_out_mean, double *d_out_var, int32_t *d_w_dump, int32_t kmax_dump,
                     void *stream) {
  MM_ARG(d_pk && d_lq && d_v && d_a && d_b && d_tile_ptr && d_slot_K && d_slot_nobs && d_slot_omq && d_slot_row && pcg_state);
  MM_ARG(d_out_mean && d_out_var && n_tiles >= 0 && num_boot > 0 && ld >= (int64_t)num_boot + 1);
  if (n_tiles == 0) return MM_OK;
  MM_ARG(n_tiles < 2147483647LL);
  // 4 tiles per 256-thread workgroup: tiles t and t + 1024 (+-3) then meet on one SIMD, which engine.pair_tiles relies on
  // (one tile per workgroup was measured too: worse when everything is resident, a wash in the many-tile regime)
  auto kern = n_tiles > 2048 ? (g_exact_arith ? k_boot1d_replay<3, false> : k_boot1d_replay<3, true>)
                             : (g_exact_arith ? k_boot1d_replay<BOOT_MIN_WAVES, false> : k_boot1d_replay<BOOT_MIN_WAVES, true>);
  hipLaunchKernelGGL(kern, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_pk, d_lq, d_v, d_a, d_b,
                     d_tile_ptr, n_tiles, d_slot_K, d_slot_nobs, d_slot_omq, d_slot_row, pcg_state[0], pcg_state[1], pcg_state[2],
                     pcg_state[3], num_boot, mean_only, ld, d_out_mean, d_out_var, d_w_dump, kmax_dump, g_wave_clock);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_boot_fill_log(double *d_mean, double *d_var, int64_t n_rows, int64_t ld, int32_t num_boot, const double mv_fit[3],
                     int32_t fill_mode, uint64_t fill_seed, int32_t *d_n_invalid, void *stream) {
  MM_ARG(d_mean && d_var && mv_fit && d_n_invalid && n_rows >= 0 && num_boot > 0 && ld >= (int64_t)num_boot + 1);
  MM_ARG(fill_mode == 0 || fill_mode == 1);
  if (n_rows == 0) return MM_OK;
  int64_t blocks = (n_rows + 3) / 4;
  hipLaunchKernelGGL(k_boot_fill_log, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_mean, d_var, n_rows, ld, num_boot,
                     mv_fit[0], mv_fit[1], mv_fit[2], fill_mode, fill_seed, d_n_invalid);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_boot2d_replay(const double *d_pk, const double *d_lq, const double *d_v1, const double *d_v2, const double *d_a,
                     const double *d_b, const int64_t *d_tile_ptr, int64_t n_tiles, const int32_t *d_slot_K,
                     const double *d_slot_nobs, const double *d_slot_omq, const int64_t *d_slot_row, const uint64_t pcg_state[4],
                     int32_t num_boot, int64_t ld, double *d_out_corr, void *stream) {
  MM_ARG(d_pk && d_lq && d_v1 && d_v2 && d_a && d_b && d_tile_ptr && d_slot_K && d_slot_nobs && d_slot_omq && d_slot_row && pcg_state);
  MM_ARG(d_out_corr && n_tiles >= 0 && num_boot > 0 && ld >= (int64_t)num_boot + 1);
  if (n_tiles == 0) return MM_OK;
  auto kern = n_tiles > 2048 ? (g_exact_arith ? k_boot2d_replay<3, false> : k_boot2d_replay<3, true>)
                             : (g_exact_arith ? k_boot2d_replay<BOOT_MIN_WAVES, false> : k_boot2d_replay<BOOT_MIN_WAVES, true>);
  hipLaunchKernelGGL(kern, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_pk, d_lq, d_v1, d_v2, d_a, d_b,
                     d_tile_ptr, n_tiles, d_slot_K, d_slot_nobs, d_slot_omq, d_slot_row, pcg_state[0], pcg_state[1], pcg_state[2],
                     pcg_state[3], num_boot, ld, d_out_corr);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_boot1d_fast(const double *d_pk, const double *d_lq, const double *d_v, const double *d_a, const double *d_b,
                   const int64_t *d_tile_ptr, int64_t n_slots, const int32_t *d_slot_K, const double *d_slot_nobs,
                   const double *d_slot_omq, const int64_t *d_slot_row, uint64_t seed, int32_t num_boot, int32_t mean_only,
                   int64_t ld, double *d_out_mean, double *d_out_var, void *stream) {
  MM_ARG(d_pk && d_lq && d_v && d_a && d_b && d_tile_ptr && d_slot_K && d_slot_nobs && d_slot_omq && d_slot_row);
  MM_ARG(d_out_mean && d_out_var && n_slots >= 0 && num_boot > 0 && ld >= (int64_t)num_boot + 1);
  if (n_slots == 0) return MM_OK;
  int32_t chunks = (num_boot + 63) / 64;
  int64_t waves = n_slots * chunks;
  int64_t blocks = (waves + 3) / 4;
  MM_ARG(blocks < 2147483647LL);
  hipLaunchKernelGGL(k_boot1d_fast, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_pk, d_lq, d_v, d_a, d_b, d_tile_ptr,
                     n_slots, d_slot_K, d_slot_nobs, d_slot_omq, d_slot_row, seed, num_boot, mean_only, chunks, ld, d_out_mean, d_out_var);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

}  // extern "C"
