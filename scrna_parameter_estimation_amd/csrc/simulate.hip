// simulate.hip -- synthetic transcriptomes and capture sampling on the device.
//
// Reference behaviour replaced (memento/simulate.py):
//   :52-68   simulate_transcriptomes, independent-gene branch: z_cg ~ NB(mean_g, theta_g), theta = 1 / dispersion
//   :91-115  capture_sampling: per cell, a multivariate hypergeometric draw of round(q_c * sum_g z_cg) molecules
//            (process='hyper'), or x_cg ~ Poisson(q_c * z_cg) (process='poisson')
// The reference materialises dense cells x genes arrays with scipy / numpy generators; at 1M x 20k that is 80 GB per array.
// Here nothing dense exists: every z_cg is a pure function of (seed, cell, gene) (counter-based streams), so the three passes
// -- per-cell totals, per-cell non-zero counts, CSR write -- regenerate it, and the captured counts go straight into a CSR in
// HBM.  One lane = one cell walking its genes in order (the hypergeometric draw is sequential in the remaining molecules);
// all lanes of a wave are at the same gene, so the NB parameters are wave-uniform loads.
// Gaussian-copula branch (simulate.py:70-89): the correlated standard-normal scores y_cg arrive as a dense [gene][cell] fp32
// matrix (Cholesky factor x white noise, a plain library GEMM on the host side of the C-ABI); here
// nb_cg = nbinom.ppf(Phi(y_cg)) and z_cg = round(nb_cg / sum_g nb_cg * cell_size_c), again regenerated in every pass.
// Draw-level parity with numpy/scipy is NOT a goal (different generators): "parity unpinned", validated statistically
// (moment recovery through the estimators, tests/test_gpu_simulate.py) -- the reference's own acceptance style
// (analysis/simulation/estimator_validation.ipynb).
#include "mm_common.h"
#include <math.h>

namespace sim {

struct Rng {
  uint64_t s;
};

__device__ __forceinline__ uint64_t mix(uint64_t x) {  // splitmix64 finaliser
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ Rng make(uint64_t seed, uint64_t a, uint64_t b) {
  return Rng{mix(seed ^ mix(a * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull) ^ mix(b + 0xD1B54A32D192ED03ull))};
}
__device__ __forceinline__ double uniform(Rng &g) {  // (0, 1)
  g.s += 0x9E3779B97F4A7C15ull;
  return ((double)(mix(g.s) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ double normal(Rng &g) {
  double u1 = uniform(g), u2 = uniform(g);
  return sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
}
// Marsaglia & Tsang (2000), shape a > 0, scale 1
__device__ double gamma(Rng &g, double a) {
  double boost = 1.0;
  if (a < 1.0) {
    boost = pow(uniform(g), 1.0 / a);
    a += 1.0;
  }
  double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
  for (int it = 0; it < 64; it++) {
    double x = normal(g), v = 1.0 + c * x;
    if (v <= 0.0) continue;
    v = v * v * v;
    double u = uniform(g), x2 = x * x;
    if (u < 1.0 - 0.0331 * x2 * x2 || log(u) < 0.5 * x2 + d * (1.0 - v + log(v))) return boost * d * v;
  }
  return boost * d;  // (never reached in practice: acceptance > 95 %)
}
// Poisson: multiplication method below 10, Hoermann's PTRS (1993) above
__device__ int64_t poisson(Rng &g, double lam) {
  if (!(lam > 0.0)) return 0;
  if (lam < 10.0) {
    double L = exp(-lam), p = 1.0;
    int64_t k = 0;
    do {
      k++;
      p *= uniform(g);
    } while (p > L && k < 1000);
    return k - 1;
  }
  double slam = sqrt(lam), loglam = log(lam);
  double b = 0.931 + 2.53 * slam, a = -0.059 + 0.02483 * b;
  double invalpha = 1.1239 + 1.1328 / (b - 3.4), vr = 0.9277 - 3.6224 / (b - 2.0);
  for (int it = 0; it < 256; it++) {
    double U = uniform(g) - 0.5, V = uniform(g);
    double us = 0.5 - fabs(U);
    double kf = floor((2.0 * a / us + b) * U + lam + 0.43);
    if (us >= 0.07 && V <= vr) return (int64_t)kf;
    if (kf < 0.0 || (us < 0.013 && V > us)) continue;
    if (log(V) + log(invalpha) - log(a / (us * us) + b) <= -lam + kf * loglam - lgamma(kf + 1.0)) return (int64_t)kf;
  }
  return (int64_t)lam;
}
// z ~ NB(mean mu, size theta) as a gamma-Poisson mixture (what scipy.stats.nbinom.rvs(theta, theta / (theta + mu)) samples)
__device__ __forceinline__ int64_t neg_binomial(uint64_t seed, int64_t cell, int32_t gene, double mu, double theta) {
  if (!(mu > 0.0)) return 0;
  Rng g = make(seed, (uint64_t)cell, (uint64_t)gene);
  double lam = gamma(g, theta) * (mu / theta);
  return poisson(g, lam);
}

// Quantile of NB(mean mu, size theta) at Phi(y): the smallest k with CDF(k) >= u (scipy.stats.nbinom.ppf).  The pmf is summed
// upwards from k0 = max(0, mean - 9 sd) (the mass below k0 is < 1e-17) with the recurrence pmf(k+1) = pmf(k) (k + theta) / (k + 1) q.
__device__ int64_t nb_quantile(double y, double mu, double theta) {
  if (!(mu > 0.0)) return 0;
  double u = 0.5 * erfc(-y * 0.70710678118654752440);
  double p = theta / (theta + mu), q = mu / (theta + mu);
  double sd = sqrt(mu + mu * mu / theta);
  double k0 = floor(fmax(0.0, mu - 9.0 * sd));
  double lp = lgamma(k0 + theta) - lgamma(theta) - lgamma(k0 + 1.0) + theta * log(p) + k0 * log(q);
  double pmf = exp(lp), cum = pmf, k = k0;
  double kmax = mu + 40.0 * sd + 60.0;
  while (cum < u && k < kmax) {
    pmf *= (k + theta) / (k + 1.0) * q;
    k += 1.0;
    cum += pmf;
  }
  return (int64_t)k;
}

// molecules of (cell, gene): independent branch = own NB draw; copula branch = quantile of the correlated score, rescaled to the
// cell's size when raw totals are given
__device__ __forceinline__ int64_t molecules(uint64_t seed, int64_t cell, int32_t gene, int64_t n_cells, double mu, double theta,
                                             const float *__restrict__ gauss, double scale) {
  if (gauss == nullptr) return neg_binomial(seed, cell, gene, mu, theta);
  int64_t nb = nb_quantile((double)gauss[(int64_t)gene * n_cells + cell], mu, theta);
  return scale < 0.0 ? nb : (int64_t)rint((double)nb * scale);      // np.round: half to even (simulate.py:89)
}

__device__ __forceinline__ double normal_at(uint64_t seed, int64_t i) {
  Rng g = make(seed, (uint64_t)i, 0x6A55ull);
  return normal(g);
}

}  // namespace sim

// mode 0: totals[cell] = sum_g z_cg.   mode 1: row_nnz[cell] = captured non-zeros.   mode 2: write the row at row_ptr[cell].
// mode 3 (copula branch only): raw_totals[cell] = sum_g nb_cg before the rescaling to the cell's size.
// process 0: hypergeometric capture of rint(q_c * total) molecules (needs totals), 1: Poisson capture, 2: none (x = z).
__global__ __launch_bounds__(256) void k_simulate(const double *__restrict__ mu, const double *__restrict__ theta, int32_t n_genes,
                                                  int64_t n_cells, const double *__restrict__ qs, uint64_t seed_z, uint64_t seed_c,
                                                  int32_t process, int32_t mode, int64_t *__restrict__ totals,
                                                  int64_t *__restrict__ row_nnz, const int64_t *__restrict__ row_ptr,
                                                  int32_t *__restrict__ out_idx, float *__restrict__ out_val,
                                                  const float *__restrict__ gauss, const double *__restrict__ cell_size,
                                                  int64_t *__restrict__ raw_totals) {
  int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= n_cells) return;
  if (mode == 3) {
    int64_t T = 0;
    for (int32_t g = 0; g < n_genes; g++) T += sim::molecules(seed_z, cell, g, n_cells, mu[g], theta[g], gauss, -1.0);
    raw_totals[cell] = T;
    return;
  }
  double scale = -1.0;                           // copula branch: z = round(nb / raw total * cell size)
  if (gauss != nullptr && cell_size != nullptr) scale = raw_totals[cell] > 0 ? cell_size[cell] / (double)raw_totals[cell] : 0.0;
  if (mode == 0) {
    int64_t T = 0;
    for (int32_t g = 0; g < n_genes; g++) T += sim::molecules(seed_z, cell, g, n_cells, mu[g], theta[g], gauss, scale);
    totals[cell] = T;
    return;
  }
  double q = process == 2 ? 1.0 : qs[cell];
  int64_t t_rem = 0, s_rem = 0;
  if (process == 0) {
    t_rem = totals[cell];
    s_rem = (int64_t)rint(q * (double)t_rem);  // np.round: half to even (simulate.py:107)
    if (s_rem > t_rem) s_rem = t_rem;
  }
  sim::Rng cap = sim::make(seed_c, (uint64_t)cell, 0x5EEDull);  // one capture stream per cell (the urn draw is sequential)
  int64_t k = 0, base = mode == 2 ? row_ptr[cell] : 0;
  for (int32_t g = 0; g < n_genes; g++) {
    int64_t z = sim::molecules(seed_z, cell, g, n_cells, mu[g], theta[g], gauss, scale);
    int64_t x;
    if (process == 2) {
      x = z;
    } else if (process == 1) {
      x = sim::poisson(cap, q * (double)z);
    } else {
      // selection sampling (Knuth, Algorithm S) over this gene's z molecules: each is taken with probability
      // (molecules still to take) / (molecules still in the urn) -- exactly the multivariate hypergeometric law
      x = 0;
      for (int64_t i = 0; i < z; i++) {
        if (s_rem > 0 && sim::uniform(cap) * (double)t_rem < (double)s_rem) {
          x++;
          s_rem--;
        }
        t_rem--;
      }
    }
    if (x > 0) {
      if (mode == 2) {
        out_idx[base + k] = g;
        out_val[base + k] = (float)x;
      }
      k++;
    }
  }
  if (mode == 1) row_nnz[cell] = k;
}

__global__ __launch_bounds__(256) void k_std_normal(uint64_t seed, int64_t n, float *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)sim::normal_at(seed, i);
}

extern "C" {

int mm_std_normal(uint64_t seed, int64_t n, float *d_out, void *stream) {
  MM_ARG(d_out && n >= 0);
  if (n == 0) return MM_OK;
  int64_t blocks = (n + 255) / 256;
  MM_ARG(blocks < 2147483647LL);
  hipLaunchKernelGGL(k_std_normal, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, seed, n, d_out);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_simulate(const double *d_mean, const double *d_theta, int32_t n_genes, int64_t n_cells, const double *d_qs, uint64_t seed_z,
                uint64_t seed_capture, int32_t process, int32_t mode, int64_t *d_totals, int64_t *d_row_nnz, const int64_t *d_row_ptr,
                int32_t *d_out_indices, float *d_out_data, const float *d_gauss, const double *d_cell_size, int64_t *d_raw_totals,
                void *stream) {
  MM_ARG(d_mean && d_theta && n_genes > 0 && n_cells >= 0 && process >= 0 && process <= 2 && mode >= 0 && mode <= 3);
  MM_ARG(mode != 3 || (d_gauss && d_raw_totals));
  MM_ARG(!(d_gauss && d_cell_size) || d_raw_totals);
  MM_ARG(mode != 0 || d_totals);
  MM_ARG(mode != 1 || d_row_nnz);
  MM_ARG(mode != 2 || (d_row_ptr && d_out_indices && d_out_data));
  MM_ARG(mode == 0 || process == 2 || d_qs);
  MM_ARG(mode == 0 || process != 0 || d_totals);
  if (n_cells == 0) return MM_OK;
  int64_t blocks = (n_cells + 255) / 256;
  MM_ARG(blocks < 2147483647LL);
  hipLaunchKernelGGL(k_simulate, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_mean, d_theta, n_genes, n_cells, d_qs,
                     seed_z, seed_capture, process, mode, d_totals, d_row_nnz, d_row_ptr, d_out_indices, d_out_data, d_gauss, d_cell_size,
                     d_raw_totals);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

}  // extern "C"
