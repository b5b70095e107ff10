// Host build of the product header csrc/npy_rng.h so the numpy-replay algorithm can be checked
// against numpy itself on a machine without a GPU (tests/test_npy_rng_host.py).
static long g_fallbacks[2] = {0, 0};
#define NPY_NOTE_FALLBACK(which) (g_fallbacks[which]++)
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

// the guarded fp32 fast paths of the search loops (what the HIP replay kernels run by default): draws must still be numpy's
void host_multinomial_fast(const uint64_t st[4], int64_t n, const double *pix, int d, int B, int64_t *out) {
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
    int32_t dn32 = (int32_t)n;
    for (int j = 0; j < d - 1; j++) {
      mn[j] = npyrng::binomial_pre<int32_t, true>(g, pk[j], lq[j], dn32);
      dn32 -= (int32_t)mn[j];
      if (dn32 <= 0) break;
    }
    if (dn32 > 0) mn[d - 1] = dn32;
  }
  delete[] pk;
  delete[] lq;
}

// the guarded fast paths with ONE BTPE attempt per call (binomial_pre_capped), the call repeated until the draw is complete: what a
// lane of the lock-step tile kernel does over consecutive bin steps (csrc/boot.hip: BOOT_BTPE_CAP)
void host_multinomial_capped(const uint64_t st[4], int64_t n, const double *pix, int d, int B, int64_t *out) {
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
    int32_t dn32 = (int32_t)n;
    for (int j = 0; j < d - 1; j++) {
      bool pending = true;
      int32_t w = 0;
      int cap = 1 + (b + j) % 2;                    // one or two attempts per call
      while (pending) w = npyrng::binomial_pre_capped<int32_t>(g, pk[j], lq[j], dn32, cap, pending);
      mn[j] = w;
      dn32 -= w;
      if (dn32 <= 0) break;
    }
    if (dn32 > 0) mn[d - 1] = dn32;
  }
  delete[] pk;
  delete[] lq;
}

// the same chain through the RESUMABLE form of the samplers (csrc/npy_rng.h: lane_begin / lane_inv / lane_att / ...), driven the
// way one lane of the lane-asynchronous tile kernel (csrc/boot.hip: k_boot1d_async) drives it: one phase per pass
void host_multinomial_async(const uint64_t st[4], int64_t n, const double *pix, int d, int B, int64_t *out) {
  using namespace npyrng;
  Pcg64 g{st[0], st[1], st[2], st[3]};
  double *pk = new double[d], *lq = new double[d];
  double rem = 1.0;
  for (int j = 0; j < d - 1; j++) {
    pk[j] = pix[j] / rem;
    lq[j] = binomial_lq(pk[j]);
    rem -= pix[j];
  }
  LaneDraw D;
  for (int b = 0; b < B; b++) {
    int64_t *mn = out + (int64_t)b * d;
    int32_t dn = (int32_t)n;
    int k = 0;
    int32_t state = LS_START;
    while (k < d - 1 && dn > 0) {
      if (state == LS_START) state = lane_begin(D, g, pk[k], lq[k], dn);
      if (state == LS_INV) state = lane_inv(D);
      if (state == LS_ATT) state = lane_att(D, g);
      if (state == LS_EXPL) state = lane_expl(D);
      if (state == LS_SQZ) state = lane_sqz(D);
      if (state == LS_XINV) state = lane_xinv(D, g);
      if (state == LS_XBT) state = lane_xbt(D, g);
      if (state == LS_DONE) {
        mn[k] = D.w;
        dn -= D.w;
        k++;
        state = LS_START;
      }
    }
    if (dn > 0) mn[d - 1] = dn;
  }
  delete[] pk;
  delete[] lq;
}

// ... and through the branch-free forms of the two common phases (lane_begin_bf / lane_inv_att_bf / lane_att_rest), called the way
// k_boot1d_async calls them: for every pass, whatever the lane's state
void host_multinomial_async_bf(const uint64_t st[4], int64_t n, const double *pix, int d, int B, int64_t *out) {
  using namespace npyrng;
  Pcg64 g{st[0], st[1], st[2], st[3]};
  double *pk = new double[d], *lq = new double[d];
  double rem = 1.0;
  for (int j = 0; j < d - 1; j++) {
    pk[j] = pix[j] / rem;
    lq[j] = binomial_lq(pk[j]);
    rem -= pix[j];
  }
  LaneDraw D = LaneDraw();
  for (int b = 0; b < B; b++) {
    int64_t *mn = out + (int64_t)b * d;
    int32_t dn = (int32_t)n;
    int k = 0;
    int32_t state = LS_START;
    long pass = 0;
    while (k < d - 1 && dn > 0) {
      pass++;
      int32_t s0 = lane_begin_bf(D, g, pk[k], lq[k], dn > 0 ? dn : 1, state == LS_START);
      if (state == LS_START) state = s0;
      state = lane_inv_att_bf(D, g, state);
      if (state == LS_ATT2) state = lane_att_rest(D);
      if (state == LS_EXPL) state = lane_expl(D);
      if ((pass & 1) == 0 && state == LS_SQZ) state = lane_sqz(D);
      if ((pass & 7) == 0 && state == LS_XINV) state = lane_xinv(D, g);
      if ((pass & 7) == 0 && state == LS_XBT) state = lane_xbt(D, g);
      if (state == LS_DONE) {
        mn[k] = D.w;
        dn -= D.w;
        k++;
        state = LS_START;
      }
    }
    if (dn > 0) mn[d - 1] = dn;
  }
  delete[] pk;
  delete[] lq;
}

// count binomial draws through binomial_pre<.., true> with explicit (n, p) and return the two fallback counters
void host_binomial_fast(const uint64_t st[4], double p, int64_t n, int count, int64_t *out) {
  npyrng::Pcg64 g{st[0], st[1], st[2], st[3]};
  double lq = npyrng::binomial_lq(p);
  for (int i = 0; i < count; i++) out[i] = npyrng::binomial_pre<int32_t, true>(g, p, lq, (int32_t)n);
}

void host_fallback_counts(long out[2], int reset) {
  out[0] = g_fallbacks[0];
  out[1] = g_fallbacks[1];
  if (reset) g_fallbacks[0] = g_fallbacks[1] = 0;
}
}
