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

// Chains that run one per WAVE (below: chain_body, k_boot1d_chain).  The tile kernel takes them as well: a tile flagged in
// ``tile_chain`` is such a chain, so that the host can put chain waves at chosen places of ONE launch's dispatch order.
#define MM_CHAIN_CLOCK_OFF (1 << 18)   // engine.CHAIN_CLOCK_OFF: the chain waves' records in the mm_debug_wave_clock buffer
struct ChainArgs {
  const double *ops;           // 8-double operand records (mm_bins_order, MM_CHAIN_SLOT pairs)
  const int64_t *base;         // [chain] first record
  const int32_t *K;            // [chain] bins
  const double *nobs, *omq;    // [chain] N_g, 1 - q_g
  const int64_t *row;          // [chain] output row
  const uint64_t *jump;        // [64][4] PCG64 jump constants
  const int32_t *tile_chain;   // [tile] chain index or -1 (tile kernel only; may be NULL)
  int32_t *w_dump;             // optional weights [chain][k][b]
  int32_t kmax_dump;
  int64_t debug_rows_mod;      // timing experiments only (mm_debug_replay_rows_mod): > 0 = tiles read operand rows modulo this (WRONG results)
};
template <bool FAST>
__device__ __forceinline__ void chain_body(const ChainArgs &ca, int64_t ch, int lane, uint64_t st0, uint64_t st1, int32_t num_boot,
                                           int32_t mean_only, int64_t ld, double *__restrict__ out_mean, double *__restrict__ out_var,
                                           int64_t *__restrict__ wave_clock);

// Every chain of a launch replays the SAME stream: the reference seeds PCG64(5) anew for every (gene, group) pair
// (memento/bootstrap.py:102).  So the stream's uniforms can be produced ONCE (k_pcg64_stream: lane-parallel jump-ahead, a few
// milliseconds for millions of outputs) and a chain's generator shrinks to its position in that table: one gather load per
// uniform instead of a 128-bit multiply-add, an xor-shift-rotate and an integer -> double conversion per lane (~45 VALU
// instructions, ten of them quarter-rate multiplies: about a quarter of the tile kernel's VALU time).  Same uniforms, same draws.
namespace npyrng {
struct TableRng {
  const double *__restrict__ tab;
  int64_t len, pos;
  int32_t *overflow;                 // set when a chain runs past the table (the host then redoes the launch with the arithmetic generator)
  typedef int64_t Mark;
  __device__ __forceinline__ Mark mark() const { return pos; }
  __device__ __forceinline__ void rewind(Mark m) { pos = m; }
  __device__ __forceinline__ void reserve(int) {}
  __device__ __forceinline__ int max_attempts() const { return 16; }
};
__device__ __forceinline__ double pcg64_next_double(TableRng &g) {
  int64_t p = g.pos++;
  if (p >= g.len) {
    *g.overflow = 1;
    p = g.len - 1;
  }
  return g.tab[p];
}
// RING generator: the lane's own PCG64 stream, produced AHEAD of its use into a 16-slot ring in LDS.  In the lock-step tile
// kernel every sampler call site costs the whole wave a PCG64 step (~45 VALU instructions, a third of them quarter-rate
// multiplies) however few lanes draw there: seven sites per bin step (one for the inversion sampler, two per BTPE attempt of the
// unluckiest lane) for ~1.5 uniforms a lane actually uses.  With the ring ALL lanes step their generators together, a fixed
// number of times per bin step (top_up), and a draw just reads its uniforms back (ds_read); a lane that runs dry mid-draw steps
// its generator on the spot.  Same stream, same uniforms in the same order.  Rewinding (the exact redo of a guarded draw) moves
// the read position back: the ring keeps the last 16 uniforms, so the fast BTPE may use 14 (7 attempts) before it must hand over.
struct RingRng {
  uint64_t s_hi, s_lo, i_hi, i_lo;   // generator state at stream position ``tail``
  int32_t head, tail;                // uniforms consumed / produced so far (slot = position & 15)
  double *ring;                      // this lane's column of the [16][256] LDS ring (stride 256 doubles)
  typedef int32_t Mark;
  __device__ __forceinline__ Mark mark() const { return head; }
  __device__ __forceinline__ void rewind(Mark m) { head = m; }
  __device__ __forceinline__ void reserve(int) {}
  __device__ __forceinline__ int max_attempts() const { return 7; }
  __device__ __forceinline__ double step() {
    Pcg64 g{s_hi, s_lo, i_hi, i_lo};
    double u = pcg64_next_double(g);
    s_hi = g.s_hi;
    s_lo = g.s_lo;
    return u;
  }
  __device__ __forceinline__ void top_up(int rounds) {      // every lane with room produces ``rounds`` more uniforms (wave-uniform trip count)
    for (int j = 0; j < rounds; j++) {
      if (tail - head < 16) {
        ring[(tail & 15) * 256] = step();
        tail++;
      }
    }
  }
};
__device__ __forceinline__ double pcg64_next_double(RingRng &g) {
  double u;
  if (g.head == g.tail) {            // ring empty: produce on the spot (and keep it, a rewind may come back to it)
    u = g.step();
    g.ring[(g.tail & 15) * 256] = u;
    g.tail++;
  } else {
    u = g.ring[(g.head & 15) * 256];
  }
  g.head++;
  return u;
}
template <int MODE> struct GenOf { typedef Pcg64 type; };     // MODE 0: arithmetic, 1: stream table, 2: ring
template <> struct GenOf<1> { typedef TableRng type; };
template <> struct GenOf<2> { typedef RingRng type; };
}  // namespace npyrng

// out[i] = the (i + 1)-th uniform of the stream that starts at ``state`` (numpy: Generator(PCG64).random()): every thread jumps
// to the start of its run of 64 outputs (PCG's O(log n) advance) and steps through it.
__global__ __launch_bounds__(256) void k_pcg64_stream(double *__restrict__ out, int64_t n, uint64_t st0, uint64_t st1, uint64_t st2,
                                                      uint64_t st3) {
  typedef unsigned __int128 u128;
  const u128 MULT = ((u128)2549297995355413924ULL << 64) | 4865540595714422341ULL;
  int64_t first = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 64;
  if (first >= n) return;
  u128 state = ((u128)st0 << 64) | st1, inc = ((u128)st2 << 64) | st3;
  u128 acc_mult = 1, acc_plus = 0, cur_mult = MULT, cur_plus = inc;
  for (uint64_t delta = (uint64_t)first; delta > 0; delta >>= 1) {
    if (delta & 1) {
      acc_mult *= cur_mult;
      acc_plus = acc_plus * cur_mult + cur_plus;
    }
    cur_plus = (cur_mult + 1) * cur_plus;
    cur_mult *= cur_mult;
  }
  state = acc_mult * state + acc_plus;
  npyrng::Pcg64 g{(uint64_t)(state >> 64), (uint64_t)state, st2, st3};
  int64_t last = first + 64 < n ? first + 64 : n;
  for (int64_t i = first; i < last; i++) out[i] = npyrng::pcg64_next_double(g);
}

#ifndef BOOT_MIN_WAVES
#define BOOT_MIN_WAVES 2
#endif
#ifndef BOOT_SETPRIO
#define BOOT_SETPRIO 2
#endif
// MINW = waves per SIMD the register budget is set for: 2 when every tile is resident (<= 2048 tiles, the pairing order below
// assumes two per SIMD), 3 in the many-tile regime where a third resident wave adds a little issue throughput.
#ifndef BOOT_RING_ROUNDS
#define BOOT_RING_ROUNDS 2       // uniforms every lane produces ahead per bin step in ring mode (a lane uses ~1.5 on average)
#endif
#ifndef BOOT_BTPE_CAP
#define BOOT_BTPE_CAP 1          // BTPE attempts a lane makes per bin step of its tile (0: as many as the draw takes, every lane waiting)
#endif
#ifndef BOOT_TAIL_LANES
#define BOOT_TAIL_LANES 4        // with at most this many lanes still inside the replicate, draws run to completion
#endif
template <int MINW, bool FAST, int TAB>
__global__ __launch_bounds__(256, MINW) void k_boot1d_replay(const double *__restrict__ pk_, const double *__restrict__ lq_,
                                                       const double *__restrict__ v, const double *__restrict__ a,
                                                       const double *__restrict__ b,
                                                       const int64_t *__restrict__ tile_ptr, int64_t n_tiles,
                                                       const int32_t *__restrict__ slot_K, const double *__restrict__ slot_nobs,
                                                       const double *__restrict__ slot_omq, const int64_t *__restrict__ slot_row, uint64_t st0, uint64_t st1,
                                                       uint64_t st2, uint64_t st3, int32_t num_boot, int32_t mean_only,
                                                       int64_t ld, double *__restrict__ out_mean, double *__restrict__ out_var,
                                                       int32_t *__restrict__ w_dump, int32_t kmax_dump,
                                                       int64_t *__restrict__ wave_clock, ChainArgs ca,
                                                       const double *__restrict__ stream_tab, int64_t stream_len,
                                                       int32_t *__restrict__ stream_overflow) {
  int lane = mm_lane();
  int64_t tile = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (tile >= n_tiles) return;
  if (ca.tile_chain) {                                  // this wave's "tile" may be a chain of the one-wave-per-chain form
    int cidx = __builtin_amdgcn_readfirstlane(ca.tile_chain[tile]);
    if (cidx >= 0) {
      if (tile * 2 < n_tiles) __builtin_amdgcn_s_setprio(BOOT_SETPRIO);
      chain_body<FAST>(ca, (int64_t)cidx, lane, st0, st1, num_boot, mean_only, ld, out_mean, out_var,
                       wave_clock ? wave_clock + MM_CHAIN_CLOCK_OFF : nullptr);
      return;
    }
  }
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
  if (ca.debug_rows_mod > 0) row0 %= ca.debug_rows_mod;     // (timing experiment: operands out of a cache-resident region)
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
  typename npyrng::GenOf<TAB>::type g;
  if constexpr (TAB == 2) {
    __shared__ double ring_lds[16 * 256];
    g.s_hi = st0;
    g.s_lo = st1;
    g.i_hi = st2;
    g.i_lo = st3;
    g.head = 0;
    g.tail = 0;
    g.ring = ring_lds + threadIdx.x;
    g.top_up(10);
  } else if constexpr (TAB == 1) {
    g.tab = stream_tab;
    g.len = stream_len;
    g.pos = 0;
    g.overflow = stream_overflow;
  } else {
    g.s_hi = st0;
    g.s_lo = st1;
    g.i_hi = st2;
    g.i_lo = st3;
  }
  const bool run = K >= 2;
#ifdef BOOT_STAMPS
  uint64_t stamp_inv = 0, stamp_btpe = 0, stamp_t0 = __builtin_amdgcn_s_memtime();
  uint64_t stamp_fastcall = 0;                       // wave time inside the fast BTPE call (the rest of stamp_btpe is the exact redo)
  uint64_t stamp_iters = 0, stamp_tail_iters = 0;    // bin steps the wave really made (retries included) / of them with only stragglers left
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
    // The lanes share the instruction stream, not the bin index: a BTPE draw makes BOOT_BTPE_CAP attempt(s) per bin step, and a lane
    // whose attempts were all rejected stays on its bin and tries again in the next step, beside its neighbours' next draws
    // (binomial_pre_capped) -- so a step costs the wave one attempt, not as many as its unluckiest lane needs.  The lanes meet again
    // at the end of the replicate; while only stragglers (<= BOOT_TAIL_LANES lanes) are left, draws run to completion.
    int kl = 0;
    for (;;) {
      const bool act = run && kl < K;
      const uint64_t act_mask = __ballot(act);
      if (act_mask == 0) break;
#ifdef BOOT_STAMPS
      stamp_iters++;
      if (__popcll(act_mask) <= BOOT_TAIL_LANES) stamp_tail_iters++;
#endif
      const int cap = (BOOT_BTPE_CAP > 0 && __popcll(act_mask) > BOOT_TAIL_LANES) ? BOOT_BTPE_CAP : 0;
      int kn = kl + 1 < K ? kl + 1 : 0;
      int64_t on = obase + (int64_t)kn * 64;
      double n_pk = pk_[on], n_lq = lq_[on], n_v = v[on], n_a = a[on], n_b = b[on];
      if constexpr (TAB == 2) g.top_up(BOOT_RING_ROUNDS);
      bool adv = false;
      if (act) {
        int32_t w;
        bool pending = false;
        if (kl < K - 1) {
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
              auto saved = g.mark();
              bool bt = !zero && !inv;
              if (bt) {
                NPY_CLOCK(stamp_bt[7]);
                X = FAST ? npyrng::binomial_btpe_fast<int32_t>(g, dn, p, cap, stamp_bt) : -1;
                cnt_bt++;
              }
              uint64_t s15;
              NPY_CLOCK(s15);
              stamp_fastcall += s15 - s1;
              if (bt && X == -2) {
                pending = true;
                cnt_bt--;
#ifdef STAMP_RETRY       // (the second counter then counts retried attempts instead of exact redos)
                cnt_fb++;
#endif
              } else if (bt && X < 0) {
#ifndef STAMP_RETRY
                cnt_fb++;
#endif
                g.rewind(saved);
                X = npyrng::binomial_btpe<int32_t>(g, dn, p);
              }
              NPY_CLOCK(s2);
              stamp_inv += s1 - s0;
              stamp_btpe += s2 - s1;
              w = zero ? 0 : (flip ? dn - X : X);
            }
#else
            if constexpr (FAST) w = npyrng::binomial_pre_capped<int32_t>(g, c_pk, c_lq, dn, cap, pending);
            else w = npyrng::binomial_pre<int32_t, false>(g, c_pk, c_lq, dn);
#endif
            if (!pending) {
              dn -= w;
              if (dn <= 0) live = false;
            }
          }
        } else {
          w = dn > 0 ? dn : 0;
        }
        if (!pending) {
          if (w_dump) w_dump[((int64_t)slot * kmax_dump + kl) * num_boot + r] = (int32_t)w;
          if (w != 0) {
            double wd = (double)w;
            M1 += (c_v * wd) * c_a;
            M2 += ((c_v * c_v) * wd) * c_b - ((omq * c_v) * wd) * c_b;
          }
          adv = true;
        }
      }
      if (adv) {
        kl++;
        c_pk = n_pk; c_lq = n_lq; c_v = n_v; c_a = n_a; c_b = n_b;
      }
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
    wave_clock[tile * 4 + 0] = (int64_t)((stamp_tail_iters << 40) | stamp_iters);
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
// CHAIN kernel: one WAVE per (gene, group) chain.  A chain is one sequential PCG64 stream, so a chain that sits alone in a
// 64-wide tile of k_boot1d_replay leaves 63 lanes idle and still pays the per-lane (exec-masked) control flow of that kernel.
// Here every value of the chain is wave-uniform -- operands come in through scalar loads, the samplers' branches are scalar
// branches -- and the 64 lanes do the one thing that parallelises: the generator.  PCG64 is an LCG, so the state j steps on
// is A^j * s + C_j (mod 2^128): lane j holds (A^(j+1), C_(j+1)) and one 128-bit multiply-add per lane produces the next 64
// outputs of the stream at once; the samplers take them one by one with v_readlane.  Same draws, same replicate moments (bit
// for bit) as k_boot1d_replay; the host sends the long chains here and packs the rest into tiles (engine.Bootstrap1D.run).
namespace npyrng {
__device__ __forceinline__ uint64_t wave_read64(uint64_t x, int lane) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, lane);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), lane);
  return ((uint64_t)hi << 32) | lo;
}

struct WaveRng {
  uint64_t s_hi, s_lo;              // per lane: the state (lane + 1) steps after the batch's base state
  double u;                         // per lane: the uniform that state yields
  uint64_t a_hi, a_lo, c_hi, c_lo;  // per lane: A^(lane+1), C_(lane+1)
  int pos;                          // wave-uniform: next lane to hand out
  typedef int Mark;
  __device__ __forceinline__ Mark mark() const { return pos; }
  __device__ __forceinline__ void rewind(Mark m) { pos = m; }
  __device__ __forceinline__ void fill(uint64_t b_hi, uint64_t b_lo) {   // b = wave-uniform base state
    uint64_t lo = a_lo * b_lo;
    uint64_t hi = __umul64hi(a_lo, b_lo) + a_hi * b_lo + a_lo * b_hi;
    uint64_t nlo = lo + c_lo;
    uint64_t nhi = hi + c_hi + (nlo < lo ? 1ULL : 0ULL);
    s_lo = nlo;
    s_hi = nhi;
    uint64_t x = nhi ^ nlo;
    unsigned rot = (unsigned)(nhi >> 58);
    uint64_t out = (x >> rot) | (x << ((64u - rot) & 63u));
    u = (double)(out >> 11) * (1.0 / 9007199254740992.0);
    pos = 0;
  }
  __device__ __forceinline__ void refill() {   // continue behind the last uniform handed out
    if (pos == 0) return;
    uint64_t b_hi = wave_read64(s_hi, pos - 1), b_lo = wave_read64(s_lo, pos - 1);
    fill(b_hi, b_lo);
  }
  __device__ __forceinline__ void reserve(int n) {
    if (pos > 64 - n) refill();
  }
  __device__ __forceinline__ int max_attempts() const { return 16; }
};

__device__ __forceinline__ double pcg64_next_double(WaveRng &g) {
  if (g.pos == 64) g.refill();
  uint64_t bits = wave_read64((uint64_t)__double_as_longlong(g.u), g.pos);
  g.pos++;
  return __longlong_as_double((long long)bits);
}
}  // namespace npyrng

#ifndef CHAIN_MIN_WAVES
#define CHAIN_MIN_WAVES 3
#endif
#ifndef CHAIN_PRIO
#define CHAIN_PRIO 3      // issue priority of a chain wave against the tile kernel's waves (first half of its grid: BOOT_SETPRIO = 2)
#endif
// Four chains (waves) per 256-thread workgroup and the register budget of the tile kernel's three-waves-per-SIMD build: a
// workgroup of either kernel then takes the same resources, so the chain waves and the tile waves of one bootstrap are all
// resident together however the dispatcher interleaves the two launches (64-thread workgroups with their own register / LDS
// footprint fragmented the CUs: tile workgroups waited for seconds, measured in profiles/README.md).
template <bool FAST>
__device__ __forceinline__ void chain_body(const ChainArgs &ca, int64_t ch, int lane, uint64_t st0, uint64_t st1, int32_t num_boot,
                                           int32_t mean_only, int64_t ld, double *__restrict__ out_mean, double *__restrict__ out_var,
                                           int64_t *__restrict__ wave_clock) {
  const double *__restrict__ ops = ca.ops;
  const int64_t *__restrict__ ch_base = ca.base;
  const uint64_t *__restrict__ jump = ca.jump;
  int32_t *__restrict__ w_dump = ca.w_dump;
  const int32_t kmax_dump = ca.kmax_dump;
  int64_t t_start = wave_clock ? (int64_t)wall_clock64() : 0;
  const int K = ca.K[ch];
  const int64_t row = ca.row[ch];
  const double nobs = ca.nobs[ch], omq = ca.omq[ch];
  const int32_t n = (int32_t)nobs;
  // wave-uniform addresses: the compiler reads the operand records with scalar loads, one bin ahead (staging them in LDS
  // instead measured the same step time: the scalar cache / L2 round trip is already hidden behind the step)
  const double *__restrict__ op = ops + ch_base[ch] * 8;      // [K][8]: pk, lq, v, a, b, (3 spare)
  double *om = out_mean + row * ld + 1;
  double *ov = out_var + row * ld + 1;
  npyrng::WaveRng g;
  g.a_hi = jump[lane * 4 + 0];
  g.a_lo = jump[lane * 4 + 1];
  g.c_hi = jump[lane * 4 + 2];
  g.c_lo = jump[lane * 4 + 3];
  g.fill(st0, st1);
  double keep_m = 0.0, keep_v = 0.0;     // lane (r & 63) keeps replicate r until 64 of them go out as one coalesced store
#ifdef BOOT_STAMPS
  uint64_t st_n_inv = 0, st_c_inv = 0, st_n_bt = 0, st_c_bt = 0, st_t0 = __builtin_amdgcn_s_memtime();
#endif
  for (int r = 0; r < num_boot; r++) {
    double M1 = 0.0, M2 = 0.0;
    int32_t dn = n;
    double c_pk = op[0], c_lq = op[1], c_v = op[2], c_a = op[3], c_b = op[4];
    for (int k = 0; k < K - 1; k++) {
      const double *nx = op + (int64_t)(k + 1) * 8;          // k + 1 <= K - 1: the last bin's v, a, b are read here
      double n_pk = nx[0], n_lq = nx[1], n_v = nx[2], n_a = nx[3], n_b = nx[4];
#ifdef BOOT_STAMPS   // diagnostic build (tools/chain_micro.py): shader cycles inside the sampler call, by sampler
      uint64_t s0_, s1_;
      const double pe_ = c_pk <= 0.5 ? c_pk : 1.0 - c_pk;
      const bool inv_ = pe_ * (double)dn <= 30.0;
      NPY_CLOCK(s0_);
#endif
      int32_t w = npyrng::binomial_pre<int32_t, FAST, true>(g, c_pk, c_lq, dn);
#ifdef BOOT_STAMPS
      NPY_CLOCK(s1_);
      if (c_pk != 0.0) {
        if (inv_) { st_n_inv++; st_c_inv += s1_ - s0_; } else { st_n_bt++; st_c_bt += s1_ - s0_; }
      }
#endif
      dn -= w;
      if (w_dump && lane == 0) w_dump[((int64_t)ch * kmax_dump + k) * num_boot + r] = w;
      if (w != 0) {
        double wd = (double)w;
        M1 += (c_v * wd) * c_a;
        M2 += ((c_v * c_v) * wd) * c_b - ((omq * c_v) * wd) * c_b;
      }
      c_pk = n_pk; c_lq = n_lq; c_v = n_v; c_a = n_a; c_b = n_b;
      if (dn <= 0) break;                                      // numpy stops the chain here; nothing is left for later bins
    }
    if (dn > 0) {                                              // the loop ran to its end: c_* hold the last bin, which takes the rest
      if (w_dump && lane == 0) w_dump[((int64_t)ch * kmax_dump + (K - 1)) * num_boot + r] = dn;
      double wd = (double)dn;
      M1 += (c_v * wd) * c_a;
      M2 += ((c_v * c_v) * wd) * c_b - ((omq * c_v) * wd) * c_b;
    }
    double mean = M1 / nobs;
    double var = M2 / nobs - mean * mean;
    if (mean_only) {
      mean = mean + 1;
      var = 10.0;
    }
    if (lane == (r & 63)) {
      keep_m = mean;
      keep_v = var;
    }
    if ((r & 63) == 63 || r == num_boot - 1) {
      int r0 = r & ~63;
      if (r0 + lane <= r) {
        om[r0 + lane] = keep_m;
        ov[r0 + lane] = keep_v;
      }
    }
  }
  if (wave_clock && lane == 0) {   // 8 int64 per chain: start, end (100 MHz); the stamps build adds sampler calls / shader cycles
    wave_clock[ch * 8 + 0] = t_start;
    wave_clock[ch * 8 + 1] = (int64_t)wall_clock64();
#ifdef BOOT_STAMPS
    wave_clock[ch * 8 + 2] = (int64_t)st_n_inv;
    wave_clock[ch * 8 + 3] = (int64_t)st_c_inv;
    wave_clock[ch * 8 + 4] = (int64_t)st_n_bt;
    wave_clock[ch * 8 + 5] = (int64_t)st_c_bt;
    wave_clock[ch * 8 + 6] = (int64_t)(__builtin_amdgcn_s_memtime() - st_t0);
#endif
  }
}

template <bool FAST>
__global__ __launch_bounds__(256, CHAIN_MIN_WAVES) void k_boot1d_chain(ChainArgs ca, int64_t n_chains, uint64_t st0, uint64_t st1,
                                                       int32_t num_boot, int32_t mean_only, int64_t ld, double *__restrict__ out_mean,
                                                       double *__restrict__ out_var, int64_t *__restrict__ wave_clock) {
  const int lane = threadIdx.x & 63;
  const int64_t ch = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform
  if (ch >= n_chains) return;
  __builtin_amdgcn_s_setprio(CHAIN_PRIO);
  chain_body<FAST>(ca, ch, lane, st0, st1, num_boot, mean_only, ld, out_mean, out_var, wave_clock);
}

// ------------------------------------------------------------------------------------------------
// ASYNC tile kernel: one lane = one chain, like k_boot1d_replay, but the lanes of a wave are NOT in lock step.  Every lane
// carries the state of its own draw (npyrng::LaneDraw) and its own position (replicate r, bin k); one pass of the wave runs a
// fixed sequence of phases -- retire / start, <= 9 steps of the inversion search, one BTPE attempt, explicit product, squeeze,
// exact redo -- each for the lanes that are in it, and a lane whose draw is done goes on to ITS next bin in the next pass whatever
// its neighbours are doing.  No lane waits for the longest search, the unluckiest BTPE draw or the longest chain of its wave; a
// lane's operands are the 8-double records of its chain (the mm_boot1d_chain layout), read one bin ahead.  Draws and replicate
// moments are those of k_boot1d_replay bit for bit (same samplers, same arithmetic per draw, same accumulation order).
#ifndef ASYNC_MIN_WAVES
#define ASYNC_MIN_WAVES 2     // 64 chains per wave: the chains of a launch rarely fill two waves per SIMD; no register spills
#endif
struct BinOps {
  double pk, lq, v, a, b;
};
__device__ __forceinline__ BinOps load_bin(const double *__restrict__ rec) {
  BinOps o;
  o.pk = rec[0];
  o.lq = rec[1];
  o.v = rec[2];
  o.a = rec[3];
  o.b = rec[4];
  return o;
}

template <bool FAST>
__global__ __launch_bounds__(256, ASYNC_MIN_WAVES) void k_boot1d_async(const double *__restrict__ ops, const int64_t *__restrict__ ch_base,
                                                        const int32_t *__restrict__ ch_K, const double *__restrict__ ch_nobs,
                                                        const double *__restrict__ ch_omq, const int64_t *__restrict__ ch_row,
                                                        int64_t n_slots, uint64_t st0, uint64_t st1, uint64_t st2, uint64_t st3,
                                                        int32_t num_boot, int32_t mean_only, int64_t ld, double *__restrict__ out_mean,
                                                        double *__restrict__ out_var, int32_t *__restrict__ w_dump, int32_t kmax_dump,
                                                        int64_t *__restrict__ wave_clock) {
  using namespace npyrng;
  const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t t_start = wave_clock ? (int64_t)wall_clock64() : 0;
  int K = slot < n_slots ? ch_K[slot] : 0;
  if (K < 2) K = 0;                                             // unused lane (K == 1 rows are NaN rows, written by the host)
  const int64_t sl = K ? slot : 0;
  const double nobs = ch_nobs[sl], omq = ch_omq[sl];
  const int32_t n = (int32_t)nobs;
  const double *__restrict__ rec = ops + ch_base[sl] * 8;      // [K][8]: pk, lq, v, a, b, (3 spare)
  const int64_t row = ch_row[sl];
  double *om = out_mean + row * ld + 1;
  double *ov = out_var + row * ld + 1;
  Pcg64 g{st0, st1, st2, st3};
  LaneDraw D = LaneDraw();
  int32_t state = K ? LS_RESTART : LS_IDLE;
  int32_t r = 0, k = 0, dn = n;
  double M1 = 0.0, M2 = 0.0;
  BinOps cur = load_bin(rec), nxt = load_bin(rec + (K ? 8 : 0));
  int64_t passes = 0;
#ifdef BOOT_STAMPS   // diagnostic build (tools/chain_sweep.py): shader cycles of a wave per phase of the pass
  uint64_t ph_c[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_t;
#define ASYNC_STAMP(i) do { uint64_t t_; NPY_CLOCK(t_); ph_c[i] += t_ - ph_t; ph_t = t_; } while (0)
#else
#define ASYNC_STAMP(i)
#endif
  while (__ballot(state != LS_IDLE)) {
    passes++;
#ifdef BOOT_STAMPS
    NPY_CLOCK(ph_t);
#endif
    // ---- retire the finished draw; finish the replicate or move on to the next bin ----------------------------------------
    if (state == LS_DONE) {
      const int32_t w = D.w;
      if (w_dump) w_dump[((int64_t)slot * kmax_dump + k) * num_boot + r] = w;
      if (w != 0) {
        double wd = (double)w;
        M1 += (cur.v * wd) * cur.a;
        M2 += ((cur.v * cur.v) * wd) * cur.b - ((omq * cur.v) * wd) * cur.b;
      }
      dn -= w;
      k++;
      cur = nxt;                                                 // record k
      if (dn <= 0 || k == K - 1) {
        state = LS_FINISH;                                       // the replicate is complete: closed in one of the passes below
      } else {
        nxt = load_bin(rec + (int64_t)(k + 1) * 8);              // k + 1 <= K - 1
        state = LS_START;
      }
    } else if (state == LS_RESTART) {
      M1 = 0.0;
      M2 = 0.0;
      dn = n;
      k = 0;
      state = LS_START;
    }
    // closing a replicate (the last bin's remainder, two fp64 divisions, the stores, the next replicate's first operands) is ~100
    // dependent instructions that some lane of a 64-wide wave needs on nearly every pass: done every fourth pass, for all lanes
    // that wait for it (a lane loses 1.5 passes per replicate on average, every other lane saves the time on three passes in four)
    if ((passes & 3) == 0 && state == LS_FINISH) {
      if (dn > 0) {                                              // every bin but the last has drawn: the last one takes the rest
        if (w_dump) w_dump[((int64_t)slot * kmax_dump + k) * num_boot + r] = dn;
        double wd = (double)dn;
        M1 += (cur.v * wd) * cur.a;
        M2 += ((cur.v * cur.v) * wd) * cur.b - ((omq * cur.v) * wd) * cur.b;
      }
      double mean = M1 / nobs;
      double var = M2 / nobs - mean * mean;
      if (mean_only) {
        mean = mean + 1;
        var = 10.0;
      }
      om[r] = mean;
      ov[r] = var;
      r++;
      if (r < num_boot) {
        cur = load_bin(rec);                                     // in flight until the lane starts its next replicate (next pass)
        nxt = load_bin(rec + 8);
        state = LS_RESTART;
      } else {
        state = LS_IDLE;
      }
    }
    ASYNC_STAMP(0);
    // ---- the draw of bin k, phase by phase (csrc/npy_rng.h) ------------------------------------------------------------------
    if (FAST) {
      // start + inversion segment + BTPE attempt in one straight line for every lane (csrc/npy_rng.h: the branch-free forms):
      // the independent dependency chains interleave instead of queueing behind three branches
      int32_t s0 = lane_begin_bf(D, g, cur.pk, cur.lq, dn > 0 ? dn : 1, state == LS_START);
      state = state == LS_START ? s0 : state;
      ASYNC_STAMP(1);
      state = lane_inv_att_bf(D, g, state);
      ASYNC_STAMP(2);
      if (state == LS_ATT2) state = lane_att_rest(D);
    } else if (state == LS_START) {                              // mm_debug_replay_arith(1): numpy's arithmetic, draw by draw
      D.w = binomial_pre<int32_t, false>(g, cur.pk, cur.lq, dn);
      state = LS_DONE;
    }
    ASYNC_STAMP(3);
    if (state == LS_EXPL) state = lane_expl(D);
    ASYNC_STAMP(4);
    // the rarer phases do not run on every pass: a lane in one of them waits a pass or a few, every other lane saves the time
    // (squeeze / Stirling: ~3 % of the lanes, every second pass; the exact redo: ~0.3 % of the draws, every 32nd pass)
    if ((passes & 1) == 0) {
      if (state == LS_SQZ) state = lane_sqz(D);
    }
    ASYNC_STAMP(5);
    if ((passes & 31) == 0) {
      if (state == LS_XINV) state = lane_xinv(D, g);
      if (state == LS_XBT) state = lane_xbt(D, g);
    }
    ASYNC_STAMP(6);
  }
  if (wave_clock && mm_lane() == 0) {
    int64_t wave = slot >> 6;
    wave_clock[wave * 4 + 0] = t_start;
    wave_clock[wave * 4 + 1] = (int64_t)wall_clock64();
    wave_clock[wave * 4 + 2] = passes;
    wave_clock[wave * 4 + 3] = 0;
#ifdef BOOT_STAMPS
    for (int i = 0; i < 7; i++) wave_clock[(1 << 19) + wave * 8 + i] = (int64_t)ph_c[i];
#endif
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
                                                       int32_t fill_mode, uint64_t seed, int32_t *__restrict__ n_invalid,
                                                       const int64_t *__restrict__ row_key) {
  int lane = mm_lane();
  int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= n_rows) return;
  // the refill draws of a row are a function of (seed, key of the row, replicate): with the caller's keys (gene position in the
  // unsharded gene order x groups + group) they do not depend on how the genes were chunked or sharded over GPUs
  const uint64_t key = row_key ? (uint64_t)row_key[row] : (uint64_t)row;
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
          uint64_t ctr = mix64(seed ^ mix64(key * 2 + which) ^ ((uint64_t)r << 20));
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
// FREE-RUNNING tile kernel: one lane = one chain, the lanes of a wave share the instruction stream and nothing else.  As in
// k_boot1d_replay a BTPE draw makes BOOT_BTPE_CAP attempt(s) per bin step and a rejected lane retries in the next step -- but here a
// lane that finishes a replicate starts its next one at once instead of waiting for the slowest lane of its wave (in-kernel stamps of
// k_boot1d_replay: a 64-wide tile makes 1.23 bin steps per nominal step, 1.095 is the lanes' average), so every lane carries its own
// replicate index as well.  The lanes drift apart without bound, and operand ROWS shared by the wave (one coalesced 512-B row per plane
// and step) would turn into 64 separate lines per plane: the operands are therefore the chain's own 8-double RECORDS (mm_bins_order,
// MM_CHAIN_SLOT: pk, lq, count, 1/sf, 1/sf^2, one 64-B half line per bin, the form chain_body reads), fetched one bin ahead.
// Same samplers, same arithmetic per draw, same accumulation order per replicate: the replicate moments are those of
// k_boot1d_replay bit for bit.  A tile flagged in ca.tile_chain is a chain of the one-wave-per-chain form, as there.
template <int MINW, bool FAST>
__global__ __launch_bounds__(256, MINW) void k_boot1d_free(const double *__restrict__ recs, const int64_t *__restrict__ slot_rec,
                                                           int64_t n_tiles, const int32_t *__restrict__ slot_K,
                                                           const double *__restrict__ slot_nobs, const double *__restrict__ slot_omq,
                                                           const int64_t *__restrict__ slot_row, uint64_t st0, uint64_t st1, uint64_t st2,
                                                           uint64_t st3, int32_t num_boot, int32_t mean_only, int64_t ld,
                                                           double *__restrict__ out_mean, double *__restrict__ out_var,
                                                           int32_t *__restrict__ w_dump, int32_t kmax_dump,
                                                           int64_t *__restrict__ wave_clock, ChainArgs ca) {
  int lane = mm_lane();
  int64_t tile = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (tile >= n_tiles) return;
  if (ca.tile_chain) {
    int cidx = __builtin_amdgcn_readfirstlane(ca.tile_chain[tile]);
    if (cidx >= 0) {
      if (tile * 2 < n_tiles) __builtin_amdgcn_s_setprio(BOOT_SETPRIO);
      chain_body<FAST>(ca, (int64_t)cidx, lane, st0, st1, num_boot, mean_only, ld, out_mean, out_var,
                       wave_clock ? wave_clock + MM_CHAIN_CLOCK_OFF : nullptr);
      return;
    }
  }
  int64_t t_start = wave_clock ? (int64_t)wall_clock64() : 0;
  if (tile * 2 < n_tiles) __builtin_amdgcn_s_setprio(BOOT_SETPRIO);
  int64_t slot = tile * 64 + lane;
  int K = slot_K[slot];
  int64_t row = slot_row[slot];
  int64_t rec0 = slot_rec[slot];
  if (K <= 0 || row < 0 || rec0 < 0) K = 0;  // unused lane
  double nobs = slot_nobs[slot];
  double omq = slot_omq[slot];
  int32_t n = (int32_t)nobs;
  double *om = out_mean + row * ld + 1;
  double *ov = out_var + row * ld + 1;
  if (K == 1) {  // bootstrap.py:97-98: a single bin -> all-NaN replicates
    for (int r = 0; r < num_boot; r++) {
      om[r] = NAN;
      ov[r] = NAN;
    }
  }
  npyrng::Pcg64 g{st0, st1, st2, st3};
  const double *__restrict__ rec = recs + (K >= 2 ? rec0 : 0) * 8;
  bool more = K >= 2;                       // this lane still has replicates to draw
  int rl = 0, kl = 0;                       // its replicate and its bin
  double M1 = 0.0, M2 = 0.0;
  int32_t dn = n;
  bool live = true;
  double2 c01 = *(const double2 *)(rec), c23 = *(const double2 *)(rec + 2);
  double c_b = rec[4];
  int64_t steps = 0;
  for (;;) {
#ifdef BOOT_FREE_SYNC    // experiment: the lanes meet at the end of every replicate, as in k_boot1d_replay (isolates the cost of the record layout)
    const bool act = more && kl < K;
    const uint64_t act_mask = __ballot(act);
    if (act_mask == 0) {
      if (more) {
        double mean = M1 / nobs;
        double var = M2 / nobs - mean * mean;
        if (mean_only) {
          mean = mean + 1;
          var = 10.0;
        }
        om[rl] = mean;
        ov[rl] = var;
        rl++;
        kl = 0;
        M1 = 0.0;
        M2 = 0.0;
        dn = n;
        live = true;
        more = rl < num_boot;
      }
      if (__ballot(more) == 0) break;
      continue;
    }
#else
    const bool act = more;
    const uint64_t act_mask = __ballot(more);
    if (act_mask == 0) break;
#endif
    steps++;
    const int cap = (BOOT_BTPE_CAP > 0 && __popcll(act_mask) > BOOT_TAIL_LANES) ? BOOT_BTPE_CAP : 0;
    const int kn = kl + 1 < K ? kl + 1 : 0;
    const double *nx = rec + (int64_t)kn * 8;
    double2 n01 = *(const double2 *)(nx), n23 = *(const double2 *)(nx + 2);
    double n_b = nx[4];
    if (act) {
      const double c_pk = c01.x, c_lq = c01.y, c_v = c23.x, c_a = c23.y;
      int32_t w;
      bool pending = false;
      if (kl < K - 1) {
        w = 0;
        if (live) {
          if constexpr (FAST) w = npyrng::binomial_pre_capped<int32_t>(g, c_pk, c_lq, dn, cap, pending);
          else w = npyrng::binomial_pre<int32_t, false>(g, c_pk, c_lq, dn);
          if (!pending) {
            dn -= w;
            if (dn <= 0) live = false;
          }
        }
      } else {
        w = dn > 0 ? dn : 0;
      }
      if (!pending) {
        if (w_dump) w_dump[((int64_t)slot * kmax_dump + kl) * num_boot + rl] = (int32_t)w;
        if (w != 0) {
          double wd = (double)w;
          M1 += (c_v * wd) * c_a;
          M2 += ((c_v * c_v) * wd) * c_b - ((omq * c_v) * wd) * c_b;
        }
        kl++;
        c01 = n01; c23 = n23; c_b = n_b;
#ifndef BOOT_FREE_SYNC
        if (kl == K) {                       // the replicate is complete (the operands just taken over are bin 0's again)
          double mean = M1 / nobs;
          double var = M2 / nobs - mean * mean;
          if (mean_only) {  // estimator._mean_only_1p (estimator.py:188-204): [mean + 1, 10]
            mean = mean + 1;
            var = 10.0;
          }
#ifdef BOOT_FREE_ABLATE_STORES   // timing experiment (WRONG results): one replicate in 16 is stored
          if ((rl & 15) == 15) {
            om[rl] = mean;
            ov[rl] = var;
          }
#elif defined(BOOT_FREE_ABLATE_DIV)   // timing experiment (WRONG results): no division in the replicate's epilogue
          om[rl] = M1;
          ov[rl] = M2;
#else
          om[rl] = mean;
          ov[rl] = var;
#endif
          rl++;
          kl = 0;
          M1 = 0.0;
          M2 = 0.0;
          dn = n;
          live = true;
          more = rl < num_boot;
        }
#endif
      }
    }
  }
  if (wave_clock && lane == 0) {  // profiling hook (mm_debug_wave_clock): when this wave ran, how many bin steps it made
    wave_clock[tile * 4 + 0] = t_start;
    wave_clock[tile * 4 + 1] = (int64_t)wall_clock64();
    wave_clock[tile * 4 + 2] = steps;
    wave_clock[tile * 4 + 3] = (int64_t)__builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
  }
}

// ------------------------------------------------------------------------------------------------
// 2D replay: same multinomial chain over the (x_i, x_j, sf_bin) bins of a gene pair; per replicate the
// covariance and the two variances (bootstrap.py:141-155, estimator.py:214-218, :171-174) are folded
// into the correlation exactly as estimator._corr_from_cov does (:281-292: 5.0 sentinel where a variance
// is <= 0, then clip to [-1, 1]).  Writes corr_b to out[row*ld + 1 + b].
#ifndef BOOT2D_BTPE_CAP
#define BOOT2D_BTPE_CAP 0
#endif
#ifndef BOOT2D_REC_BTPE_CAP
#define BOOT2D_REC_BTPE_CAP 1    // attempts per bin step when the chains read their own operand records (REC): no shared rows to scatter
#endif
template <int MINW, bool FAST, bool REC>
__global__ __launch_bounds__(256, MINW) void k_boot2d_replay(const double *__restrict__ pk_, const double *__restrict__ lq_,
                                                       const double *__restrict__ v1_, const double *__restrict__ v2_,
                                                       const double *__restrict__ a, const double *__restrict__ b,
                                                       const int64_t *__restrict__ tile_ptr, int64_t n_tiles,
                                                       const int32_t *__restrict__ slot_K, const double *__restrict__ slot_nobs,
                                                       const double *__restrict__ slot_omq, const int64_t *__restrict__ slot_row,
                                                       uint64_t st0, uint64_t st1, uint64_t st2, uint64_t st3, int32_t num_boot,
                                                       int64_t ld, double *__restrict__ out_corr, const int64_t *__restrict__ slot_rec) {
  int lane = mm_lane();
  int64_t tile = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (tile >= n_tiles) return;
  if (tile * 2 < n_tiles) __builtin_amdgcn_s_setprio(BOOT_SETPRIO);  // long tiles first, see k_boot1d_replay
  int64_t slot = tile * 64 + lane;
  int K = slot_K[slot];
  int64_t row = slot_row[slot];
  if (K <= 0 || row < 0) K = 0;
  // REC: pk_ is the record buffer (8 doubles per bin: pk, lq, x_i, x_j, 1/sf, 1/sf^2, mm_bins_order2d with MM_CHAIN_SLOT pairs) and the
  // lane's chain starts at record slot_rec[slot]; otherwise six [row][64] planes, the tile's rows from tile_ptr[tile]
  int64_t row0 = REC ? 0 : tile_ptr[tile];
  double nobs = slot_nobs[slot];
  double omq = slot_omq[slot];
  int32_t n = (int32_t)nobs;
  double *oc = out_corr + row * ld + 1;
  npyrng::Pcg64 g{st0, st1, st2, st3};
  const bool run = K >= 1;
  // operands of the NEXT bin step are loaded while the current one computes (as in k_boot1d_replay): the longest pair runs
  // alone in its wave and is the critical path of the launch, so an exposed L2 round trip per step is paid in full there
  const int64_t obase = row0 * 64 + lane;
  const double *__restrict__ rec = pk_;
  if constexpr (REC) {
    int64_t r0 = slot_rec[slot];
    if (r0 < 0) K = 0;
    rec = pk_ + (K >= 1 ? r0 : 0) * 8;
  }
  double c_pk, c_lq, c_x1, c_x2, c_a, c_b;
  if constexpr (REC) {
    double2 t0 = *(const double2 *)(rec), t1 = *(const double2 *)(rec + 2), t2 = *(const double2 *)(rec + 4);
    c_pk = t0.x; c_lq = t0.y; c_x1 = t1.x; c_x2 = t1.y; c_a = t2.x; c_b = t2.y;
  } else {
    c_pk = pk_[obase]; c_lq = lq_[obase]; c_x1 = v1_[obase]; c_x2 = v2_[obase]; c_a = a[obase]; c_b = b[obase];
  }
  for (int r = 0; r < num_boot; r++) {
    double A1 = 0.0, A2 = 0.0, MX = 0.0, Q1 = 0.0, Q2 = 0.0;
    int32_t dn = n;
    bool live = true;
    // this lane's bin, as in k_boot1d_replay -- but BOOT2D_BTPE_CAP is 0: measured on configs[3]'s share (250 x 2000 pairs, 335 bins
    // per chain on average, 64-wide tiles in several rounds) one attempt per step takes 29.2 s against 16.8 s: the steps of this
    // kernel are short (small inversion searches), and lanes that fall behind by different amounts turn the six coalesced 512-B
    // operand rows of a step into up to 64 separate lines each.
    int kl = 0;
    for (;;) {
      const bool act = run && kl < K;
      const uint64_t act_mask = __ballot(act);
      if (act_mask == 0) break;
      constexpr int CAP2D = REC ? BOOT2D_REC_BTPE_CAP : BOOT2D_BTPE_CAP;
      const int cap = (CAP2D > 0 && __popcll(act_mask) > BOOT_TAIL_LANES) ? CAP2D : 0;
      int kn = kl + 1 < K ? kl + 1 : 0;
      double n_pk, n_lq, n_x1, n_x2, n_a, n_b;
      if constexpr (REC) {
        const double *nx = rec + (int64_t)kn * 8;
        double2 t0 = *(const double2 *)(nx), t1 = *(const double2 *)(nx + 2), t2 = *(const double2 *)(nx + 4);
        n_pk = t0.x; n_lq = t0.y; n_x1 = t1.x; n_x2 = t1.y; n_a = t2.x; n_b = t2.y;
      } else {
        int64_t on = obase + (int64_t)kn * 64;
        n_pk = pk_[on]; n_lq = lq_[on]; n_x1 = v1_[on]; n_x2 = v2_[on]; n_a = a[on]; n_b = b[on];
      }
      bool adv = false;
      if (act) {
        int32_t w;
        bool pending = false;
        if (kl < K - 1) {
          w = 0;
          if (live) {
            if constexpr (FAST) w = npyrng::binomial_pre_capped<int32_t>(g, c_pk, c_lq, dn, cap, pending);
            else w = npyrng::binomial_pre<int32_t, false>(g, c_pk, c_lq, dn);
            if (!pending) {
              dn -= w;
              if (dn <= 0) live = false;
            }
          }
        } else {
          w = dn > 0 ? dn : 0;
        }
        if (!pending) {
          if (w != 0) {
            double wd = (double)w, x1 = c_x1, x2 = c_x2, aa = c_a, bb = c_b;
            A1 += (x1 * wd) * aa;
            A2 += (x2 * wd) * aa;
            MX += ((x1 * x2) * wd) * bb;
            Q1 += ((x1 * x1) * wd) * bb - ((omq * x1) * wd) * bb;
            Q2 += ((x2 * x2) * wd) * bb - ((omq * x2) * wd) * bb;
          }
          adv = true;
        }
      }
      if (adv) {
        kl++;
        c_pk = n_pk; c_lq = n_lq; c_x1 = n_x1; c_x2 = n_x2; c_a = n_a; c_b = n_b;
      }
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
static int g_ring_rng = 0;               // set by mm_debug_replay_ring: 1 = the tile kernel's lanes produce their uniforms ahead into an LDS ring
static int64_t g_debug_rows_mod = 0;    // MM_DEBUG_ROWS_MOD in the environment of the process (read once): timing experiments only
static int g_exact_arith = 0;            // set by mm_debug_replay_arith: 1 = numpy's fp64 arithmetic in every search loop (A/B timing, tests)

extern "C" {

int mm_debug_wave_clock(int64_t *d_buf) {
  g_wave_clock = d_buf;
  return MM_OK;
}

int mm_debug_replay_rows_mod(int64_t rows) {
  g_debug_rows_mod = rows > 0 ? rows : 0;
  return MM_OK;
}

int mm_debug_replay_ring(int32_t on) {
  g_ring_rng = on ? 1 : 0;
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
                     int64_t co_resident_waves, const mm_chain_tiles *chains, const double *d_stream, int64_t stream_len,
                     int32_t *d_stream_overflow, void *stream) {
  MM_ARG(d_pk && d_lq && d_v && d_a && d_b && d_tile_ptr && d_slot_K && d_slot_nobs && d_slot_omq && d_slot_row && pcg_state);
  ChainArgs ca{};
  ca.debug_rows_mod = g_debug_rows_mod;
  if (chains) {
    MM_ARG(chains->d_tile_chain && chains->d_ops && chains->d_ch_base && chains->d_ch_K && chains->d_ch_nobs && chains->d_ch_omq &&
           chains->d_ch_row && chains->d_jump);
    ca = ChainArgs{chains->d_ops, chains->d_ch_base, chains->d_ch_K, chains->d_ch_nobs, chains->d_ch_omq, chains->d_ch_row, chains->d_jump,
                   chains->d_tile_chain, chains->d_w_dump, chains->kmax_dump, g_debug_rows_mod};
  }
  MM_ARG(d_out_mean && d_out_var && n_tiles >= 0 && num_boot > 0 && ld >= (int64_t)num_boot + 1);
  if (n_tiles == 0) return MM_OK;
  MM_ARG(n_tiles < 2147483647LL);
  // 4 tiles per 256-thread workgroup: tiles t and t + 1024 (+-3) then meet on one SIMD, which engine.pair_tiles relies on
  // (one tile per workgroup was measured too: worse when everything is resident, a wash in the many-tile regime)
  // the 168-VGPR build (three waves per SIMD) when this launch and the chain-kernel waves running beside it need more than two
  // wave slots per SIMD; the two-wave build otherwise
  MM_ARG(!d_stream || (stream_len > 0 && d_stream_overflow));
  const bool three = n_tiles + (co_resident_waves > 0 ? co_resident_waves : 0) > 2048;
  const int mode = d_stream ? 1 : (g_ring_rng ? 2 : 0);
  auto kern = mode == 1 ? (three ? (g_exact_arith ? k_boot1d_replay<3, false, 1> : k_boot1d_replay<3, true, 1>)
                                 : (g_exact_arith ? k_boot1d_replay<BOOT_MIN_WAVES, false, 1> : k_boot1d_replay<BOOT_MIN_WAVES, true, 1>))
            : mode == 2 ? (three ? (g_exact_arith ? k_boot1d_replay<3, false, 2> : k_boot1d_replay<3, true, 2>)
                                 : (g_exact_arith ? k_boot1d_replay<BOOT_MIN_WAVES, false, 2> : k_boot1d_replay<BOOT_MIN_WAVES, true, 2>))
                        : (three ? (g_exact_arith ? k_boot1d_replay<3, false, 0> : k_boot1d_replay<3, true, 0>)
                                 : (g_exact_arith ? k_boot1d_replay<BOOT_MIN_WAVES, false, 0> : k_boot1d_replay<BOOT_MIN_WAVES, true, 0>));
  hipLaunchKernelGGL(kern, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_pk, d_lq, d_v, d_a, d_b,
                     d_tile_ptr, n_tiles, d_slot_K, d_slot_nobs, d_slot_omq, d_slot_row, pcg_state[0], pcg_state[1], pcg_state[2],
                     pcg_state[3], num_boot, mean_only, ld, d_out_mean, d_out_var, d_w_dump, kmax_dump, g_wave_clock, ca, d_stream,
                     stream_len, d_stream_overflow);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_boot1d_free(const double *d_recs, const int64_t *d_slot_rec, int64_t n_tiles, const int32_t *d_slot_K, const double *d_slot_nobs,
                   const double *d_slot_omq, const int64_t *d_slot_row, const uint64_t pcg_state[4], int32_t num_boot, int32_t mean_only,
                   int64_t ld, double *d_out_mean, double *d_out_var, int32_t *d_w_dump, int32_t kmax_dump, const mm_chain_tiles *chains,
                   void *stream) {
  MM_ARG(d_recs && d_slot_rec && d_slot_K && d_slot_nobs && d_slot_omq && d_slot_row && pcg_state);
  ChainArgs ca{};
  if (chains) {
    MM_ARG(chains->d_tile_chain && chains->d_ops && chains->d_ch_base && chains->d_ch_K && chains->d_ch_nobs && chains->d_ch_omq &&
           chains->d_ch_row && chains->d_jump);
    ca = ChainArgs{chains->d_ops, chains->d_ch_base, chains->d_ch_K, chains->d_ch_nobs, chains->d_ch_omq, chains->d_ch_row, chains->d_jump,
                   chains->d_tile_chain, chains->d_w_dump, chains->kmax_dump, 0};
  }
  MM_ARG(d_out_mean && d_out_var && n_tiles >= 0 && num_boot > 0 && ld >= (int64_t)num_boot + 1);
  if (n_tiles == 0) return MM_OK;
  MM_ARG(n_tiles < 2147483647LL);
  const bool three = n_tiles > 2048;       // the register budgets of mm_boot1d_replay: three waves per SIMD beyond 2,048 tiles
  auto kern = three ? (g_exact_arith ? k_boot1d_free<3, false> : k_boot1d_free<3, true>)
                    : (g_exact_arith ? k_boot1d_free<BOOT_MIN_WAVES, false> : k_boot1d_free<BOOT_MIN_WAVES, true>);
  hipLaunchKernelGGL(kern, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_recs, d_slot_rec, n_tiles, d_slot_K,
                     d_slot_nobs, d_slot_omq, d_slot_row, pcg_state[0], pcg_state[1], pcg_state[2], pcg_state[3], num_boot, mean_only, ld,
                     d_out_mean, d_out_var, d_w_dump, kmax_dump, g_wave_clock, ca);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_pcg64_stream(const uint64_t pcg_state[4], int64_t n, double *d_out, void *stream) {
  MM_ARG(pcg_state && d_out && n >= 0);
  if (n == 0) return MM_OK;
  int64_t threads = (n + 63) / 64;
  hipLaunchKernelGGL(k_pcg64_stream, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_out, n, pcg_state[0],
                     pcg_state[1], pcg_state[2], pcg_state[3]);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_boot1d_chain(const double *d_ops, const int64_t *d_ch_base, const int32_t *d_ch_K, const double *d_ch_nobs,
                    const double *d_ch_omq, const int64_t *d_ch_row, int64_t n_chains, const uint64_t *d_jump,
                    const uint64_t pcg_state[4], int32_t num_boot, int32_t mean_only, int64_t ld, double *d_out_mean,
                    double *d_out_var, int32_t *d_w_dump, int32_t kmax_dump, void *stream) {
  MM_ARG(d_ops && d_ch_base && d_ch_K && d_ch_nobs && d_ch_omq && d_ch_row && d_jump && pcg_state && d_out_mean && d_out_var);
  MM_ARG(n_chains >= 0 && n_chains < 2147483647LL && num_boot > 0 && ld >= (int64_t)num_boot + 1);
  if (n_chains == 0) return MM_OK;
  auto kern = g_exact_arith ? k_boot1d_chain<false> : k_boot1d_chain<true>;
  ChainArgs ca{d_ops, d_ch_base, d_ch_K, d_ch_nobs, d_ch_omq, d_ch_row, d_jump, nullptr, d_w_dump, kmax_dump, 0};
  hipLaunchKernelGGL(kern, dim3((unsigned)((n_chains + 3) / 4)), dim3(256), 0, (hipStream_t)stream, ca, n_chains, pcg_state[0], pcg_state[1],
                     num_boot, mean_only, ld, d_out_mean, d_out_var, g_wave_clock ? g_wave_clock + MM_CHAIN_CLOCK_OFF : nullptr);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_boot1d_async(const double *d_ops, const int64_t *d_ch_base, const int32_t *d_ch_K, const double *d_ch_nobs,
                    const double *d_ch_omq, const int64_t *d_ch_row, int64_t n_slots, const uint64_t pcg_state[4], int32_t num_boot,
                    int32_t mean_only, int64_t ld, double *d_out_mean, double *d_out_var, int32_t *d_w_dump, int32_t kmax_dump,
                    void *stream) {
  MM_ARG(d_ops && d_ch_base && d_ch_K && d_ch_nobs && d_ch_omq && d_ch_row && pcg_state && d_out_mean && d_out_var);
  MM_ARG(n_slots >= 0 && n_slots < (1LL << 38) && num_boot > 0 && ld >= (int64_t)num_boot + 1);
  if (n_slots == 0) return MM_OK;
  auto kern = g_exact_arith ? k_boot1d_async<false> : k_boot1d_async<true>;
  hipLaunchKernelGGL(kern, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_ops, d_ch_base, d_ch_K, d_ch_nobs,
                     d_ch_omq, d_ch_row, n_slots, pcg_state[0], pcg_state[1], pcg_state[2], pcg_state[3], num_boot, mean_only, ld,
                     d_out_mean, d_out_var, d_w_dump, kmax_dump, g_wave_clock);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_boot_fill_log(double *d_mean, double *d_var, int64_t n_rows, int64_t ld, int32_t num_boot, const double mv_fit[3],
                     int32_t fill_mode, uint64_t fill_seed, int32_t *d_n_invalid, const int64_t *d_row_key, void *stream) {
  MM_ARG(d_mean && d_var && mv_fit && d_n_invalid && n_rows >= 0 && num_boot > 0 && ld >= (int64_t)num_boot + 1);
  MM_ARG(fill_mode == 0 || fill_mode == 1);
  if (n_rows == 0) return MM_OK;
  int64_t blocks = (n_rows + 3) / 4;
  hipLaunchKernelGGL(k_boot_fill_log, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_mean, d_var, n_rows, ld, num_boot,
                     mv_fit[0], mv_fit[1], mv_fit[2], fill_mode, fill_seed, d_n_invalid, d_row_key);
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
  auto kern = n_tiles > 2048 ? (g_exact_arith ? k_boot2d_replay<3, false, false> : k_boot2d_replay<3, true, false>)
                             : (g_exact_arith ? k_boot2d_replay<BOOT_MIN_WAVES, false, false> : k_boot2d_replay<BOOT_MIN_WAVES, true, false>);
  hipLaunchKernelGGL(kern, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_pk, d_lq, d_v1, d_v2, d_a, d_b,
                     d_tile_ptr, n_tiles, d_slot_K, d_slot_nobs, d_slot_omq, d_slot_row, pcg_state[0], pcg_state[1], pcg_state[2],
                     pcg_state[3], num_boot, ld, d_out_corr, (const int64_t *)nullptr);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_boot2d_replay_rec(const double *d_recs, const int64_t *d_slot_rec, int64_t n_tiles, const int32_t *d_slot_K,
                         const double *d_slot_nobs, const double *d_slot_omq, const int64_t *d_slot_row, const uint64_t pcg_state[4],
                         int32_t num_boot, int64_t ld, double *d_out_corr, void *stream) {
  MM_ARG(d_recs && d_slot_rec && d_slot_K && d_slot_nobs && d_slot_omq && d_slot_row && pcg_state);
  MM_ARG(d_out_corr && n_tiles >= 0 && num_boot > 0 && ld >= (int64_t)num_boot + 1);
  if (n_tiles == 0) return MM_OK;
  auto kern = n_tiles > 2048 ? (g_exact_arith ? k_boot2d_replay<3, false, true> : k_boot2d_replay<3, true, true>)
                             : (g_exact_arith ? k_boot2d_replay<BOOT_MIN_WAVES, false, true> : k_boot2d_replay<BOOT_MIN_WAVES, true, true>);
  const double *nul = nullptr;
  hipLaunchKernelGGL(kern, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_recs, nul, nul, nul, nul, nul,
                     (const int64_t *)nullptr, n_tiles, d_slot_K, d_slot_nobs, d_slot_omq, d_slot_row, pcg_state[0], pcg_state[1],
                     pcg_state[2], pcg_state[3], num_boot, ld, d_out_corr, d_slot_rec);
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
