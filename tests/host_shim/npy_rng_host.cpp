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
  for (int i = 0; i < count; i++) out[i] = npyrng::binomial(g, p, n);
}

// out is B x d row-major, zero-initialised by the caller (numpy layout before the .T)
void host_multinomial(const uint64_t st[4], int64_t n, const double *pix, int d, int B, int64_t *out) {
  npyrng::Pcg64 g{st[0], st[1], st[2], st[3]};
  for (int b = 0; b < B; b++) {
    int64_t *mn = out + (int64_t)b * d;
    double remaining_p = 1.0;
    int64_t dn = n;
    for (int j = 0; j < d - 1; j++) {
      mn[j] = npyrng::binomial(g, pix[j] / remaining_p, dn);
      dn -= mn[j];
      if (dn <= 0) break;
      remaining_p -= pix[j];
    }
    if (dn > 0) mn[d - 1] = dn;
  }
}
}
