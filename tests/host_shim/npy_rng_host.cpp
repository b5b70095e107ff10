// Host build of the product header csrc/npy_rng.h so the numpy-replay algorithm can be checked
// against numpy itself on a machine without a GPU (tests/test_npy_rng_host.py).
#include "npy_rng.h"

extern "C" {

void host_pcg64_raw(const uint64_t st[4], int n, uint64_t *out) {
  npyrng::Pcg64 g{st[0], st[1], st[2], st[3]};
  for (int i = 0; i < n; i++) out[i] = npyrng::pcg64_next64(g);
}

void host_binomial(const uint64_t st[4], double p, int64_t n, int count, int64_t *out) {
  npyrng::Pcg64 g{st[0], st[1], st[2], st[3]};
  for (int i = 0; i < count; i++) out[i] = npyrng::binomial<int64_t>(g, p, n);
}

// out is B x d row-major, zero-initialised by the caller (numpy layout before the .T)
void host_multinomial(const uint64_t st[4], int64_t n, const double *pix, int d, int B, int64_t *out) {
  npyrng::Pcg64 g{st[0], st[1], st[2], st[3]};
  for (int b = 0; b < B; b++) {
    int64_t *mn = out + (int64_t)b * d;
    double remaining_p = 1.0;
    int64_t dn = n;
    for (int j = 0; j < d - 1; j++) {
      mn[j] = npyrng::binomial<int64_t>(g, pix[j] / remaining_p, dn);
      dn -= mn[j];
      if (dn <= 0) break;
      remaining_p -= pix[j];
    }
    if (dn > 0) mn[d - 1] = dn;
  }
}

// same chain through the hoisted per-bin preparation (what the HIP kernel runs)
void host_multinomial_pre(const uint64_t st[4], int64_t n, const double *pix, int d, int B, int64_t *out) {
  npyrng::Pcg64 g{st[0], st[1], st[2], st[3]};
  double *pk = new double[d], *lq = new double[d];
  double rem = 1.0;
  for (int j = 0; j < d - 1; j++) {
    pk[j] = pix[j] / rem;
    lq[j] = npyrng::binomial_lq(pk[j]);
    rem -= pix[j];
  }
  for (int b = 0; b < B; b++) {
    int64_t *mn = out + (int64_t)b * d;
    int64_t dn = n;
    int32_t dn32 = (int32_t)n;  // the device kernel runs the chain in int32 (N_g < 2^31)
    for (int j = 0; j < d - 1; j++) {
      mn[j] = npyrng::binomial_pre<int32_t>(g, pk[j], lq[j], dn32);
      dn32 -= (int32_t)mn[j];
      if (dn32 <= 0) break;
    }
    dn = dn32;
    if (dn > 0) mn[d - 1] = dn;
  }
  delete[] pk;
  delete[] lq;
}
}
