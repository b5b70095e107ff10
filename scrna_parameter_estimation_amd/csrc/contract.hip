// contract.hip -- K9+K10: per-test linear contraction over replicate groups and null statistics.
//
// Reference behaviour replaced (memento/hypothesis_test.py):
//   :249-251  valid_boostrap_iters -- drop replicate columns with any non-finite entry (mean OR var rows)
//   :262-271, :290-291  weighted mean / (residualise on covariates + _cross_coef): LINEAR in the response,
//             so the host folds it into one weight row W[t][:] per test (see memento/design.py)
//   :297-298  nanstd of the replicate coefficients -> standard error
//   :62-92    _compute_asl: all-equal check, null = coef[1:] - coef[0], two-sided extreme count
// One 256-thread workgroup per test; replicates across lanes (coalesced along b).
#include "mm_common.h"
#include <math.h>

#define K9_THREADS 256

__device__ __forceinline__ double wg_sum(double x, double *red) {
  for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < K9_THREADS / 64; i++) t += red[i];
  return t;
}
__device__ __forceinline__ double wg_min(double x, double *red) {
  for (int off = 32; off > 0; off >>= 1) x = fmin(x, __shfl_xor(x, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  double t = red[0];
  for (int i = 1; i < K9_THREADS / 64; i++) t = fmin(t, red[i]);
  return t;
}
__device__ __forceinline__ double wg_max(double x, double *red) {
  for (int off = 32; off > 0; off >>= 1) x = fmax(x, __shfl_xor(x, off, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  double t = red[0];
  for (int i = 1; i < K9_THREADS / 64; i++) t = fmax(t, red[i]);
  return t;
}

__global__ __launch_bounds__(K9_THREADS) void k_contract_stats(const double *__restrict__ ym, const double *__restrict__ yv, int64_t ld,
                                                               int32_t num_boot, int32_t n_groups, const int32_t *__restrict__ test_gene,
                                                               const double *__restrict__ W, const uint8_t *__restrict__ good,
                                                               int32_t which, double *__restrict__ coef, double *__restrict__ stats) {
  extern __shared__ double sm[];
  double *wrow = sm;                              // [n_groups]
  int32_t *glist = (int32_t *)(sm + n_groups);    // [n_groups] indices of good groups
  __shared__ double red[K9_THREADS / 64];
  __shared__ int n_good_s;
  int64_t t = blockIdx.x;
  int gene = test_gene[t];
  const uint8_t *gd = good + (int64_t)gene * n_groups;
  if (threadIdx.x == 0) {
    int ng = 0;
    for (int j = 0; j < n_groups; j++)
      if (gd[j]) glist[ng++] = j;
    n_good_s = ng;
  }
  for (int j = threadIdx.x; j < n_groups; j += K9_THREADS) wrow[j] = W[t * n_groups + j];
  __syncthreads();
  int n_good = n_good_s;
  double *crow = coef + t * ld;
  double *st = stats + t * 8;
  int64_t row_base = (int64_t)gene * n_groups;
  int n_cols = num_boot + 1;
  if (n_good == 0) {
    for (int c = threadIdx.x; c < n_cols; c += K9_THREADS) crow[c] = NAN;
    if (threadIdx.x == 0) {
      st[0] = NAN; st[1] = NAN; st[2] = 0; st[3] = 0; st[4] = NAN; st[5] = 0; st[6] = NAN; st[7] = NAN;
    }
    return;
  }
  // pass A: coefficients
  double s_sum = 0.0, s_cnt = 0.0, s_min = INFINITY, s_max = -INFINITY;
  for (int c = threadIdx.x; c < n_cols; c += K9_THREADS) {
    double acc = 0.0;
    bool ok = true;
    for (int q = 0; q < n_good; q++) {
      int j = glist[q];
      int64_t o = (row_base + j) * ld + c;
      double a = ym[o], b = yv[o];
      ok = ok && isfinite(a) && isfinite(b);
      acc += wrow[j] * (which ? b : a);
    }
    double val = ok ? acc : NAN;
    crow[c] = val;
    if (ok) {
      s_min = fmin(s_min, val);
      s_max = fmax(s_max, val);
      if (c > 0) {
        s_sum += val;
        s_cnt += 1.0;
      }
    }
  }
  double tot = wg_sum(s_sum, red);
  double cnt = wg_sum(s_cnt, red);
  double mn = wg_min(s_min, red);
  double mx = wg_max(s_max, red);
  __threadfence_block();
  __syncthreads();
  double c0 = crow[0];
  double mean1 = cnt > 0 ? tot / cnt : NAN;
  double a0 = fabs(c0);
  // pass B: variance about the mean of coef[1:], extreme count of null = coef[1:] - coef[0]
  double s_sq = 0.0, s_ext = 0.0, s_raw = 0.0;
  for (int c = 1 + threadIdx.x; c < n_cols; c += K9_THREADS) {
    double val = crow[c];
    if (val == val) {  // dropped replicates are stored as NaN
      double d = val - mean1;
      s_sq += d * d;
      double nul = val - c0;
      if (nul > a0 || nul < -a0) s_ext += 1.0;
      if (val > a0 || val < -a0) s_raw += 1.0;   // null NOT centred on the observed value (resampling != 'bootstrap')
    }
  }
  double sq = wg_sum(s_sq, red);
  double ext = wg_sum(s_ext, red);
  double raw = wg_sum(s_raw, red);
  if (threadIdx.x == 0) {
    st[0] = c0;
    st[1] = cnt > 0 ? sqrt(sq / cnt) : NAN;
    st[2] = cnt;
    st[3] = ext;
    st[4] = mean1 - c0;
    st[5] = (mn == mx) ? 1.0 : 0.0;
    st[6] = raw;
    st[7] = mx - mn;
  }
}

// ------------------------------------------------------------------------------------------------
// resample_rep=True (hypothesis_test.py:273-286, :231-239): hierarchical resampling of the replicate groups.
// Step 0: which replicate columns survive hypothesis_test.py:249-251 (a column is dropped when ANY good group has a
// non-finite mean OR variance entry).  One workgroup per gene: col_map[gene][k] = k-th surviving column (ascending),
// n_valid[gene] = how many.  With nothing dropped col_map is the identity and n_valid = num_boot + 1.
__global__ __launch_bounds__(256) void k_valid_cols(const double *__restrict__ ym, const double *__restrict__ yv, int64_t ld,
                                                    int32_t num_boot, int32_t n_groups, const uint8_t *__restrict__ good,
                                                    int32_t *__restrict__ col_map, int32_t *__restrict__ n_valid) {
  extern __shared__ int32_t glist_v[];             // [n_groups] indices of good groups
  __shared__ int n_good_s, base_s;
  __shared__ int wave_cnt[4];
  int64_t gene = blockIdx.x;
  const uint8_t *gd = good + gene * n_groups;
  if (threadIdx.x == 0) {
    int ng = 0;
    for (int j = 0; j < n_groups; j++)
      if (gd[j]) glist_v[ng++] = j;
    n_good_s = ng;
    base_s = 0;
  }
  __syncthreads();
  int n_good = n_good_s;
  int n_cols = num_boot + 1;
  int32_t *cm = col_map + gene * (int64_t)n_cols;
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int c0 = 0; c0 < n_cols; c0 += 256) {
    int c = c0 + threadIdx.x;
    bool ok = c < n_cols;
    if (ok) {
      for (int q = 0; q < n_good; q++) {
        int64_t o = (gene * n_groups + glist_v[q]) * ld + c;
        ok = ok && isfinite(ym[o]) && isfinite(yv[o]);
      }
    }
    uint64_t bal = __ballot(ok);
    int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wv] = __popcll(bal);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wv; w++) off += wave_cnt[w];
    if (ok) cm[off + before] = c;
    __syncthreads();
    if (threadIdx.x == 0) base_s += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) n_valid[gene] = base_s;
}

// Step 1: residualise the rows of a gene on the covariates, dst = M src  (M = I - H of the weighted fit, zero on bad
// groups), column by column.  Any number of groups (one group per donor is the reference's real use of resample_rep,
// analysis/lupus/run_memento.py:31-52): row i of M is staged in LDS, the column entries come back from L1/L2.
// The sum runs over j in ascending order and skips exact zeros of M (rows that are all zero, i.e. bad groups, give NaN).
__global__ __launch_bounds__(256) void k_residualize(const double *__restrict__ src, double *__restrict__ dst, int64_t ld,
                                                     int32_t n_cols, int32_t n_groups, int32_t col_tiles,
                                                     const int32_t *__restrict__ gene_mask /* [n_genes] index into M */,
                                                     const double *__restrict__ M /* [n_masks][ng][ng] */) {
  extern __shared__ double mrow[];                 // [n_groups]
  int64_t gene = blockIdx.x / col_tiles;
  int tile = (int)(blockIdx.x % col_tiles);
  const double *Mg = M + (int64_t)gene_mask[gene] * n_groups * n_groups;
  int c = tile * 256 + threadIdx.x;
  bool mine = c < n_cols;
  const double *sb = src + gene * n_groups * ld + c;
  double *db = dst + gene * n_groups * ld + c;
  for (int i = 0; i < n_groups; i++) {
    __syncthreads();
    for (int j = threadIdx.x; j < n_groups; j += 256) mrow[j] = Mg[(int64_t)i * n_groups + j];
    __syncthreads();
    if (!mine) continue;
    double acc = 0.0;
    bool any = false;
    for (int j = 0; j < n_groups; j++) {
      double m = mrow[j];
      if (m != 0.0) {
        acc += m * sb[(int64_t)j * ld];
        any = true;
      }
    }
    db[(int64_t)i * ld] = any ? acc : NAN;
  }
}

__device__ __forceinline__ uint64_t rr_mix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// Step 2: per test and resampled column c < nb (nb = surviving columns - 1, hypothesis_test.py:249-254): rows i = 0..n-1 take
// group rep[i][c] and the bcol[i][c]-th SURVIVING replicate column of the residualised response; coefficient = weighted slope
// on the residualised treatment of the drawn groups (_cross_coef_resampled).  rep/bcol NULL -> drawn on the fly from a
// counter-based RNG (column 0 is always the identity / observed column).  rep/bcol rows have stride num_boot.  Then the same
// null statistics as k_contract_stats.
__global__ __launch_bounds__(K9_THREADS) void k_cross_resampled(const double *__restrict__ yt, int64_t ld, int32_t num_boot,
                                                                int32_t n_groups, const int32_t *__restrict__ test_gene,
                                                                const double *__restrict__ tt /* [n_tests][ng] residualised treatment */,
                                                                const uint8_t *__restrict__ good, const double *__restrict__ Nc,
                                                                const int16_t *__restrict__ rep, const int32_t *__restrict__ bcol,
                                                                const int32_t *__restrict__ col_map, const int32_t *__restrict__ n_valid,
                                                                uint64_t seed, double *__restrict__ coef, double *__restrict__ stats) {
  extern __shared__ double sm[];
  double *tts = sm;                               // [ng]
  double *ncs = sm + n_groups;                    // [ng]
  int32_t *glist = (int32_t *)(sm + 2 * n_groups);
  __shared__ double red[K9_THREADS / 64];
  __shared__ int n_good_s;
  int64_t t = blockIdx.x;
  int gene = test_gene[t];
  const uint8_t *gd = good + (int64_t)gene * n_groups;
  if (threadIdx.x == 0) {
    int ng = 0;
    for (int j = 0; j < n_groups; j++)
      if (gd[j]) glist[ng++] = j;
    n_good_s = ng;
  }
  for (int j = threadIdx.x; j < n_groups; j += K9_THREADS) {
    tts[j] = tt[t * n_groups + j];
    ncs[j] = Nc[j];
  }
  __syncthreads();
  int n = n_good_s;
  double *crow = coef + t * ld;
  double *st = stats + t * 8;
  int64_t row_base = (int64_t)gene * n_groups;
  if (n == 0) {
    for (int c = threadIdx.x; c <= num_boot; c += K9_THREADS) crow[c] = NAN;
    if (threadIdx.x == 0) {
      st[0] = NAN; st[1] = NAN; st[2] = 0; st[3] = 0; st[4] = NAN; st[5] = 0; st[6] = NAN; st[7] = NAN;
    }
    return;
  }
  const int16_t *rg = rep ? rep + (int64_t)gene * n_groups * num_boot : nullptr;
  const int32_t *bg = bcol ? bcol + (int64_t)gene * n_groups * num_boot : nullptr;
  // surviving replicate columns (hypothesis_test.py:249-254): nb resampled columns, indices through col_map
  const int32_t *cm = col_map ? col_map + (int64_t)gene * (num_boot + 1) : nullptr;
  const int nb = n_valid ? n_valid[gene] - 1 : num_boot;
  if (nb < 1) {    // nothing (or only one column) survives: the reference returns NaNs ("skipped") or has no null at all
    for (int c = threadIdx.x; c <= num_boot; c += K9_THREADS) crow[c] = NAN;
    if (threadIdx.x == 0) {
      st[0] = NAN; st[1] = NAN; st[2] = 0; st[3] = 0; st[4] = NAN; st[5] = 0; st[6] = NAN; st[7] = NAN;
    }
    return;
  }
  double s_sum = 0.0, s_cnt = 0.0, s_min = INFINITY, s_max = -INFINITY;
  for (int c = threadIdx.x; c < nb; c += K9_THREADS) {
    double sw = 0.0, swy = 0.0, swa = 0.0, amax = 0.0;
    // first pass: weighted means
    for (int i = 0; i < n; i++) {
      int r, bb;
      if (c == 0) { r = i; bb = 0; }
      else if (rg) { r = rg[(int64_t)i * num_boot + c]; bb = bg[(int64_t)i * num_boot + c]; }
      else {
        uint64_t h = rr_mix(seed ^ rr_mix(((uint64_t)gene << 32) ^ ((uint64_t)i << 24) ^ (uint64_t)c));
        r = (int)(h % (uint64_t)n);
        bb = (int)(rr_mix(h) % (uint64_t)nb) + 1;
      }
      if (cm) bb = cm[bb];
      int j = glist[r];
      double y = yt[(row_base + j) * ld + bb];
      double w = ncs[j];
      sw += w; swy += w * y; swa += w * tts[j];
      amax = fmax(amax, fabs(tts[j]));
    }
    double mB = swy / sw, mA = swa / sw;
    double ss = 0.0, num = 0.0;
    for (int i = 0; i < n; i++) {
      int r, bb;
      if (c == 0) { r = i; bb = 0; }
      else if (rg) { r = rg[(int64_t)i * num_boot + c]; bb = bg[(int64_t)i * num_boot + c]; }
      else {
        uint64_t h = rr_mix(seed ^ rr_mix(((uint64_t)gene << 32) ^ ((uint64_t)i << 24) ^ (uint64_t)c));
        r = (int)(h % (uint64_t)n);
        bb = (int)(rr_mix(h) % (uint64_t)nb) + 1;
      }
      if (cm) bb = cm[bb];
      int j = glist[r];
      double y = yt[(row_base + j) * ld + bb];
      double w = ncs[j], da = tts[j] - mA;
      ss += da * da * w;
      num += (da * w) * (y - mB);
    }
    // DELIBERATE DEVIATION: a degenerate column -- every drawn group has the same (residualised)
    // treatment, so the slope is 0/0.  The reference gets NaN there when its weighted mean happens to round to exactly that
    // value and O(1) noise otherwise (hypothesis_test.py:234-239: a ratio of two round-off residues, one of which comes out of
    // LAPACK's least-squares residuals and is not reproducible bit for bit); here such a column is always NaN, which np.nanstd
    // and the isfinite filter of _compute_asl ignore.  With >= 12 groups such columns do not occur (P < 1e-3 per column).
    double val = num / sw / (ss / sw);
    if (ss / sw <= 1e-24 * amax * amax) val = NAN;
    crow[c] = val;
    if (val == val) {
      s_min = fmin(s_min, val);
      s_max = fmax(s_max, val);
      if (c > 0) {
        s_sum += val;
        s_cnt += 1.0;
      }
    }
  }
  for (int c = nb + threadIdx.x; c <= num_boot; c += K9_THREADS) crow[c] = NAN;  // the resampled row uses nb <= num_boot slots
  double tot = wg_sum(s_sum, red);
  double cnt = wg_sum(s_cnt, red);
  double mn = wg_min(s_min, red);
  double mx = wg_max(s_max, red);
  __threadfence_block();
  __syncthreads();
  double c0 = crow[0];
  double mean1 = cnt > 0 ? tot / cnt : NAN;
  double a0 = fabs(c0);
  double s_sq = 0.0, s_ext = 0.0, s_raw = 0.0;
  for (int c = 1 + threadIdx.x; c < nb; c += K9_THREADS) {
    double val = crow[c];
    if (val == val) {
      double d = val - mean1;
      s_sq += d * d;
      double nul = val - c0;
      if (nul > a0 || nul < -a0) s_ext += 1.0;
      if (val > a0 || val < -a0) s_raw += 1.0;   // null NOT centred on the observed value (resampling != 'bootstrap')
    }
  }
  double sq = wg_sum(s_sq, red);
  double ext = wg_sum(s_ext, red);
  double raw = wg_sum(s_raw, red);
  if (threadIdx.x == 0) {
    st[0] = c0;
    st[1] = cnt > 0 ? sqrt(sq / cnt) : NAN;
    st[2] = cnt;
    st[3] = ext;
    st[4] = mean1 - c0;
    st[5] = (mn == mx) ? 1.0 : 0.0;
    st[6] = raw;
    st[7] = mx - mn;
  }
}

// ------------------------------------------------------------------------------------------------
// Two-group contrasts against a shared control (Perturb-seq style, BASELINE config 5): test t compares group
// test_grp[t] with the control group of gene test_gene[t]:  coef_b = y[gene, grp][b] - y[gene, ctrl][b]
// (the weighted slope of _cross_coef for two groups and a binary treatment, hypothesis_test.py:218-228, reduces to this
// difference).  The control's bootstrap rows are computed once and shared by every guide.  Nothing per-replicate is
// stored: both passes recompute the difference from the resident replicate rows.  One launch does mean and variance.
__global__ __launch_bounds__(K9_THREADS) void k_contrast_stats(const double *__restrict__ ym, const double *__restrict__ yv,
                                                               int64_t ld, int32_t num_boot, int32_t n_groups, int32_t ctrl,
                                                               const int32_t *__restrict__ test_gene, const int32_t *__restrict__ test_grp,
                                                               const uint8_t *__restrict__ good, double *__restrict__ stats_m,
                                                               double *__restrict__ stats_v) {
  __shared__ double red[K9_THREADS / 64];
  int64_t t = blockIdx.x;
  int gene = test_gene[t], grp = test_grp[t];
  double *sm_ = stats_m + t * 8, *sv_ = stats_v + t * 8;
  const uint8_t *gd = good + (int64_t)gene * n_groups;
  if (!gd[grp] || !gd[ctrl]) {
    if (threadIdx.x == 0) {
      for (int i = 0; i < 8; i++) {
        sm_[i] = (i == 2 || i == 3 || i == 5) ? 0.0 : NAN;
        sv_[i] = (i == 2 || i == 3 || i == 5) ? 0.0 : NAN;
      }
    }
    return;
  }
  const double *ma = ym + ((int64_t)gene * n_groups + grp) * ld, *mc = ym + ((int64_t)gene * n_groups + ctrl) * ld;
  const double *va = yv + ((int64_t)gene * n_groups + grp) * ld, *vc = yv + ((int64_t)gene * n_groups + ctrl) * ld;
  int n_cols = num_boot + 1;
  double sum_m = 0, sum_v = 0, cnt = 0, mn_m = INFINITY, mx_m = -INFINITY, mn_v = INFINITY, mx_v = -INFINITY;
  for (int c = threadIdx.x; c < n_cols; c += K9_THREADS) {
    double a = ma[c], b = mc[c], p = va[c], q = vc[c];
    if (isfinite(a) && isfinite(b) && isfinite(p) && isfinite(q)) {
      double dm = a - b, dv = p - q;
      mn_m = fmin(mn_m, dm); mx_m = fmax(mx_m, dm);
      mn_v = fmin(mn_v, dv); mx_v = fmax(mx_v, dv);
      if (c > 0) { sum_m += dm; sum_v += dv; cnt += 1.0; }
    }
  }
  double n = wg_sum(cnt, red);
  double tm = wg_sum(sum_m, red), tv = wg_sum(sum_v, red);
  double lo_m = wg_min(mn_m, red), hi_m = wg_max(mx_m, red), lo_v = wg_min(mn_v, red), hi_v = wg_max(mx_v, red);
  double c0m = ma[0] - mc[0], c0v = va[0] - vc[0];
  double mean_m = n > 0 ? tm / n : NAN, mean_v = n > 0 ? tv / n : NAN;
  double am = fabs(c0m), av = fabs(c0v);
  double sq_m = 0, sq_v = 0, ex_m = 0, ex_v = 0, rw_m = 0, rw_v = 0;
  for (int c = 1 + threadIdx.x; c < n_cols; c += K9_THREADS) {
    double a = ma[c], b = mc[c], p = va[c], q = vc[c];
    if (isfinite(a) && isfinite(b) && isfinite(p) && isfinite(q)) {
      double dm = a - b, dv = p - q;
      sq_m += (dm - mean_m) * (dm - mean_m);
      sq_v += (dv - mean_v) * (dv - mean_v);
      double nm = dm - c0m, nv = dv - c0v;
      if (nm > am || nm < -am) ex_m += 1.0;
      if (nv > av || nv < -av) ex_v += 1.0;
      if (dm > am || dm < -am) rw_m += 1.0;
      if (dv > av || dv < -av) rw_v += 1.0;
    }
  }
  double qm = wg_sum(sq_m, red), qv = wg_sum(sq_v, red), em = wg_sum(ex_m, red), ev = wg_sum(ex_v, red);
  double rm = wg_sum(rw_m, red), rv = wg_sum(rw_v, red);
  if (threadIdx.x == 0) {
    sm_[0] = c0m; sm_[1] = n > 0 ? sqrt(qm / n) : NAN; sm_[2] = n; sm_[3] = em; sm_[4] = mean_m - c0m;
    sm_[5] = (lo_m == hi_m) ? 1.0 : 0.0; sm_[6] = rm; sm_[7] = hi_m - lo_m;
    sv_[0] = c0v; sv_[1] = n > 0 ? sqrt(qv / n) : NAN; sv_[2] = n; sv_[3] = ev; sv_[4] = mean_v - c0v;
    sv_[5] = (lo_v == hi_v) ? 1.0 : 0.0; sv_[6] = rv; sv_[7] = hi_v - lo_v;
  }
}

// coefficient rows of selected contrasts (for the host-side tail fits of the few tests that need them)
__global__ __launch_bounds__(256) void k_contrast_rows(const double *__restrict__ ym, const double *__restrict__ yv, int64_t ld,
                                                       int32_t num_boot, int32_t n_groups, int32_t ctrl,
                                                       const int32_t *__restrict__ test_gene, const int32_t *__restrict__ test_grp,
                                                       int32_t which, double *__restrict__ out) {
  int64_t t = blockIdx.x;
  int gene = test_gene[t], grp = test_grp[t];
  const double *ma = ym + ((int64_t)gene * n_groups + grp) * ld, *mc = ym + ((int64_t)gene * n_groups + ctrl) * ld;
  const double *va = yv + ((int64_t)gene * n_groups + grp) * ld, *vc = yv + ((int64_t)gene * n_groups + ctrl) * ld;
  for (int c = threadIdx.x; c <= num_boot; c += 256) {
    double a = ma[c], b = mc[c], p = va[c], q = vc[c];
    bool ok = isfinite(a) && isfinite(b) && isfinite(p) && isfinite(q);
    out[t * ld + c] = ok ? (which ? p - q : a - b) : NAN;
  }
}

extern "C" {

int mm_contract_stats(const double *d_ym, const double *d_yv, int64_t ld, int32_t num_boot, int32_t n_groups,
                      const int32_t *d_test_gene, const double *d_W, const uint8_t *d_good, int64_t n_tests, int32_t which,
                      double *d_coef, double *d_stats, void *stream) {
  MM_ARG(d_ym && d_yv && d_test_gene && d_W && d_good && d_coef && d_stats);
  MM_ARG(n_tests >= 0 && n_groups > 0 && num_boot > 0 && ld >= (int64_t)num_boot + 1 && (which == 0 || which == 1));
  if (n_tests == 0) return MM_OK;
  size_t shm = (size_t)n_groups * 12 + 8;
  hipLaunchKernelGGL(k_contract_stats, dim3((unsigned)n_tests), dim3(K9_THREADS), shm, (hipStream_t)stream, d_ym, d_yv, ld, num_boot,
                     n_groups, d_test_gene, d_W, d_good, which, d_coef, d_stats);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_valid_cols(const double *d_ym, const double *d_yv, int64_t ld, int32_t num_boot, int32_t n_groups, const uint8_t *d_good,
                  int64_t n_genes, int32_t *d_col_map, int32_t *d_n_valid, void *stream) {
  MM_ARG(d_ym && d_yv && d_good && d_col_map && d_n_valid && n_groups > 0 && num_boot > 0 && ld >= (int64_t)num_boot + 1);
  MM_ARG(n_genes >= 0 && n_genes < 2147483647LL);
  if (n_genes == 0) return MM_OK;
  size_t shm = (size_t)n_groups * 4;
  hipLaunchKernelGGL(k_valid_cols, dim3((unsigned)n_genes), dim3(256), shm, (hipStream_t)stream, d_ym, d_yv, ld, num_boot, n_groups,
                     d_good, d_col_map, d_n_valid);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_residualize(const double *d_src, double *d_dst, int64_t ld, int32_t n_cols, int32_t n_groups, int64_t n_genes,
                   const int32_t *d_gene_mask, const double *d_M, void *stream) {
  MM_ARG(d_src && d_dst && d_src != d_dst && d_gene_mask && d_M && n_cols > 0 && n_groups > 0 && n_genes >= 0);
  MM_ARG((size_t)n_groups * 8 <= 64 * 1024);
  if (n_genes == 0) return MM_OK;
  int32_t col_tiles = (n_cols + 255) / 256;
  MM_ARG(n_genes * col_tiles < 2147483647LL);
  size_t shm = (size_t)n_groups * 8;
  hipLaunchKernelGGL(k_residualize, dim3((unsigned)(n_genes * col_tiles)), dim3(256), shm, (hipStream_t)stream, d_src, d_dst, ld, n_cols,
                     n_groups, col_tiles, d_gene_mask, d_M);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_cross_resampled(const double *d_yt, int64_t ld, int32_t num_boot, int32_t n_groups, const int32_t *d_test_gene,
                       const double *d_tt, const uint8_t *d_good, const double *d_Nc, const int16_t *d_rep, const int32_t *d_bcol,
                       const int32_t *d_col_map, const int32_t *d_n_valid, uint64_t seed, int64_t n_tests, double *d_coef,
                       double *d_stats, void *stream) {
  MM_ARG(d_yt && d_test_gene && d_tt && d_good && d_Nc && d_coef && d_stats);
  MM_ARG(n_tests >= 0 && n_tests < 2147483647LL && n_groups > 0 && n_groups <= 32767 && num_boot > 1 && ld >= (int64_t)num_boot + 1);
  MM_ARG(((d_rep == nullptr) == (d_bcol == nullptr)) && ((d_col_map == nullptr) == (d_n_valid == nullptr)));
  if (n_tests == 0) return MM_OK;
  size_t shm = (size_t)n_groups * 20 + 8;
  hipLaunchKernelGGL(k_cross_resampled, dim3((unsigned)n_tests), dim3(K9_THREADS), shm, (hipStream_t)stream, d_yt, ld, num_boot, n_groups,
                     d_test_gene, d_tt, d_good, d_Nc, d_rep, d_bcol, d_col_map, d_n_valid, seed, d_coef, d_stats);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_contrast_stats(const double *d_ym, const double *d_yv, int64_t ld, int32_t num_boot, int32_t n_groups, int32_t ctrl,
                      const int32_t *d_test_gene, const int32_t *d_test_grp, const uint8_t *d_good, int64_t n_tests,
                      double *d_stats_mean, double *d_stats_var, void *stream) {
  MM_ARG(d_ym && d_yv && d_test_gene && d_test_grp && d_good && d_stats_mean && d_stats_var);
  MM_ARG(n_tests >= 0 && n_tests < 2147483647LL && n_groups > 0 && ctrl >= 0 && ctrl < n_groups && num_boot > 0 && ld >= (int64_t)num_boot + 1);
  if (n_tests == 0) return MM_OK;
  hipLaunchKernelGGL(k_contrast_stats, dim3((unsigned)n_tests), dim3(K9_THREADS), 0, (hipStream_t)stream, d_ym, d_yv, ld, num_boot, n_groups,
                     ctrl, d_test_gene, d_test_grp, d_good, d_stats_mean, d_stats_var);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_contrast_rows(const double *d_ym, const double *d_yv, int64_t ld, int32_t num_boot, int32_t n_groups, int32_t ctrl,
                     const int32_t *d_test_gene, const int32_t *d_test_grp, int64_t n_tests, int32_t which, double *d_out,
                     void *stream) {
  MM_ARG(d_ym && d_yv && d_test_gene && d_test_grp && d_out && n_tests >= 0 && n_tests < 2147483647LL && (which == 0 || which == 1));
  if (n_tests == 0) return MM_OK;
  hipLaunchKernelGGL(k_contrast_rows, dim3((unsigned)n_tests), dim3(256), 0, (hipStream_t)stream, d_ym, d_yv, ld, num_boot, n_groups, ctrl,
                     d_test_gene, d_test_grp, which, d_out);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

}  // extern "C"
