// hist.hip -- K5: integer (group, gene, sf_bin, count) histograms from the SELL count blocks, their
// compaction into bins, and the np.unique replay ORDER of the bins.
//
// Reference behaviour replaced: bootstrap._unique_expr  (memento/bootstrap.py:62-71):
//     code = expr.dot(np.random.random(1)) + np.random.random()*approx_sf ; np.unique(code, ...)
// The bins as a SET are deterministic integers (bit-exact); their ORDER is ascending `code`, computed
// here in IEEE fp64 with contraction off so it equals numpy's sort order.  Compile with -ffp-contract=off.
#include "mm_common.h"
#include "npy_rng.h"
#include <math.h>

// ------------------------------------------------------------------------------------------------
// Lane-per-gene histogram: scattered uint32 atomics into the pair's dense [sf_bin][count] table.
#define K5_THREADS 256
__global__ __launch_bounds__(K5_THREADS) void k_hist1d_sell(const u32x4 *__restrict__ ent, const int64_t *__restrict__ blk_base,
                                                            const int32_t *__restrict__ slice_w, const int32_t *__restrict__ slice_ptr,
                                                            const int32_t *__restrict__ item_ptr, const int32_t *__restrict__ perm,
                                                            const int32_t *__restrict__ blk_cell0, const int32_t *__restrict__ blk_group,
                                                            const uint8_t *__restrict__ sf_bin, int32_t n_slices, int32_t split,
                                                            const int32_t *__restrict__ gene_pairbase, const int64_t *__restrict__ tab_ptr,
                                                            const int32_t *__restrict__ xcap, uint32_t *__restrict__ tab) {
  __shared__ uint8_t bin_lds[MM_BLOCK_CELLS];
  __shared__ int32_t ip[1025];
  int b = blockIdx.x / split, part = blockIdx.x % split;
  int c0 = blk_cell0[b], nc = blk_cell0[b + 1] - c0;
  for (int i = threadIdx.x; i < MM_BLOCK_CELLS; i += K5_THREADS) bin_lds[i] = i < nc ? sf_bin[c0 + i] : 0;
  for (int i = threadIdx.x; i <= n_slices; i += K5_THREADS) ip[i] = item_ptr[(int64_t)b * (n_slices + 1) + i];
  __syncthreads();
  int lane = mm_lane();
  int wave = part * (K5_THREADS / 64) + (threadIdx.x >> 6);
  int nwaves = split * (K5_THREADS / 64);
  const int32_t *sw = slice_w + (int64_t)b * n_slices;
  const int32_t *sp = slice_ptr + (int64_t)b * (n_slices + 1);
  const int32_t *pm = perm + (int64_t)b * n_slices * 64;
  int n_items = ip[n_slices];
  int64_t base = blk_base[b];
  int grp = blk_group[b];
  for (int item = wave; item < n_items; item += nwaves) {
    int lo = 0, hi = n_slices;
    while (hi - lo > 1) {
      int mid = (lo + hi) >> 1;
      if (ip[mid] <= item) lo = mid; else hi = mid;
    }
    int t = lo;
    int gene = pm[t * 64 + lane];
    int pb = gene >= 0 ? gene_pairbase[gene] : -1;
    if (__ballot(pb >= 0) == 0ull) continue;  // nobody in this slice is tested
    int64_t tp = 0;
    uint32_t xc = 0;
    if (pb >= 0) {
      tp = tab_ptr[pb + grp];
      xc = (uint32_t)xcap[pb + grp];
    }
    int k = item - ip[t];
    int r0 = k * MM_ITEM_ROWS;
    int r1 = min(sw[t], r0 + MM_ITEM_ROWS);
    const u32x4 *p = ent + (base + sp[t] + r0) * 64 + lane;
    for (int r = 0; r < r1 - r0; r++) {
      u32x4 e4 = __builtin_nontemporal_load(p + (int64_t)r * 64);
      uint32_t ee[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
      for (int u = 0; u < 4; u++) {
        uint32_t x = ee[u] >> MM_CELL_BITS;
        if (x != 0 && x < xc) {
          uint32_t bin = bin_lds[ee[u] & (MM_BLOCK_CELLS - 1)];
          atomicAdd(&tab[tp + (int64_t)bin * xc + x], 1u);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Wave per pair: fill column 0 with the zero-count cells of each sf bin and count non-empty bins.
__global__ __launch_bounds__(256) void k_bins_count(uint32_t *__restrict__ tab, const int64_t *__restrict__ tab_ptr,
                                                    const int32_t *__restrict__ xcap, int64_t n_pairs, int32_t n_groups,
                                                    int32_t n_sf_bins, const uint32_t *__restrict__ grp_bin_cells,
                                                    int32_t *__restrict__ K) {
  int lane = mm_lane();
  int64_t p = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (p >= n_pairs) return;
  int grp = (int)(p % n_groups);
  int64_t tp = tab_ptr[p];
  int xc = xcap[p];
  int k = 0;
  for (int bin = 0; bin < n_sf_bins; bin++) {
    uint32_t *row = tab + tp + (int64_t)bin * xc;
    uint32_t s = 0;
    int nz = 0;
    for (int x = 1 + lane; x < xc; x += 64) {
      uint32_t c = row[x];
      s += c;
      nz += c != 0;
    }
    for (int off = 32; off > 0; off >>= 1) {
      s += __shfl_xor(s, off, 64);
      nz += __shfl_xor(nz, off, 64);
    }
    uint32_t zero = grp_bin_cells[grp * n_sf_bins + bin] - s;
    if (lane == 0) row[0] = zero;
    k += nz + (zero != 0);
  }
  if (lane == 0) K[p] = k;
}

// ------------------------------------------------------------------------------------------------
// One workgroup per pair: compact the table, compute the replay hash code, bitonic-sort ascending,
// write the bootstrap operand rows into the pair's lane of its 64-wide tile.
template <int CAP, int NT>
__global__ __launch_bounds__(NT) void k_bins_order(const uint32_t *__restrict__ tab, const int64_t *__restrict__ tab_ptr,
                                                   const int32_t *__restrict__ xcap, const int32_t *__restrict__ Karr,
                                                   const int64_t *__restrict__ pair_list, int64_t n_list, int32_t n_groups,
                                                   int32_t n_sf_bins, const double *__restrict__ sf_table,
                                                   const double *__restrict__ r1a, const double *__restrict__ r0a,
                                                   const int64_t *__restrict__ pair_slot, const int64_t *__restrict__ tile_ptr,
                                                   const double *__restrict__ grp_ncells,
                                                   double *__restrict__ o_pk, double *__restrict__ o_lq, double *__restrict__ o_v,
                                                   double *__restrict__ o_a, double *__restrict__ o_b,
                                                   int32_t *__restrict__ status) {
  extern __shared__ double smem_d[];
  double *code = smem_d;                       // [CAP]
  uint32_t *pay = (uint32_t *)(code + CAP);    // [CAP]  sf_bin << 19 | count
  uint32_t *mult = pay + CAP;                  // [CAP]
  __shared__ int n_found;
  if (blockIdx.x >= n_list) return;
  int64_t p = pair_list[blockIdx.x];
  int64_t slot = pair_slot[p];
  if (slot < 0) return;
  int K = Karr[p];
  if (K > CAP) {
    if (threadIdx.x == 0) atomicOr(status, 2);
    return;
  }
  int grp = (int)(p % n_groups);
  int64_t tp = tab_ptr[p];
  int xc = xcap[p];
  double r1 = r1a[p], r0 = r0a[p];
  int tid = threadIdx.x;
  for (int i = tid; i < CAP; i += NT) code[i] = INFINITY;
  if (tid == 0) n_found = 0;
  __syncthreads();
  // compaction by wave 0 in canonical (sf_bin major, count minor) order
  if (tid < 64) {
    int total = n_sf_bins * xc;
    int pos = 0;
    for (int i0 = 0; i0 < total; i0 += 64) {
      int i = i0 + tid;
      uint32_t c = i < total ? tab[tp + i] : 0u;
      unsigned long long m = __ballot(c != 0);
      if (c != 0) {
        int at = pos + __popcll(m & ((1ull << tid) - 1ull));
        if (at < CAP) {
          uint32_t bin = (uint32_t)(i / xc), x = (uint32_t)(i % xc);
          double cx = (double)x * r1;
          double cs = r0 * sf_table[bin];
          code[at] = cx + cs;
          pay[at] = (bin << 19) | x;
          mult[at] = c;
        }
      }
      pos += __popcll(m);
    }
    if (tid == 0) n_found = pos;
  }
  __syncthreads();
  if (n_found != K) {
    if (tid == 0) atomicOr(status, 4);
    return;
  }
  // bitonic sort (ascending) over the smallest power of two >= K
  int n = 1;
  while (n < K) n <<= 1;
  for (int k2 = 2; k2 <= n; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < n; i += NT) {
        int ixj = i ^ j;
        if (ixj > i) {
          bool up = (i & k2) == 0;
          double ci = code[i], cj = code[ixj];
          if ((ci > cj) == up && ci != cj) {
            code[i] = cj;
            code[ixj] = ci;
            uint32_t tpay = pay[i];
            pay[i] = pay[ixj];
            pay[ixj] = tpay;
            uint32_t tm = mult[i];
            mult[i] = mult[ixj];
            mult[ixj] = tm;
          }
        }
      }
      __syncthreads();
    }
  }
  // slot = tile*64 + lane of the 64-wide tile layout, or MM_CHAIN_SLOT | row for a chain of the one-wave-per-chain kernel
  // (mm_boot1d_chain): its operands go, bin by bin, into 8-double records at o_pk + 8*(row + k)
  const bool chain = (slot & MM_CHAIN_SLOT) != 0;
  int64_t tile = slot >> 6, ln = slot & 63;
  int64_t row0 = chain ? (slot & (MM_CHAIN_SLOT - 1)) : tile_ptr[tile];
  double N = grp_ncells[grp];
  bool tie = false;
  for (int k = tid; k + 1 < K; k += NT)
    if (code[k] == code[k + 1]) tie = true;
  __syncthreads();
  // numpy: remaining_p starts at 1.0 and loses pix[j] after every drawn bin (sequential rounding), so the
  // success probability of bin k, pix[k]/remaining_p, is replicate-independent: compute it once here.
  if (tid == 0) {
    double rem = 1.0;
    for (int k = 0; k < K; k++) {
      code[k] = rem;  // the sort keys are no longer needed
      rem -= (double)mult[k] / N;
    }
  }
  __syncthreads();
  for (int k = tid; k < K; k += NT) {
    uint32_t bin = pay[k] >> 19, x = pay[k] & ((1u << 19) - 1u);
    double sf = sf_table[bin];
    double pk = ((double)mult[k] / N) / code[k];
    if (chain) {
      double *rec = o_pk + (row0 + k) * 8;
      rec[0] = pk;
      rec[1] = npyrng::binomial_lq(pk);
      rec[2] = (double)x;
      rec[3] = 1.0 / sf;
      rec[4] = 1.0 / (sf * sf);
      continue;
    }
    int64_t o = (row0 + k) * 64 + ln;
    o_pk[o] = pk;
    o_lq[o] = npyrng::binomial_lq(pk);
    o_v[o] = (double)x;
    o_a[o] = 1.0 / sf;
    o_b[o] = 1.0 / (sf * sf);
  }
  if (tie) atomicOr(status, 8);  // np.unique would merge these two bins; caller must handle (never seen in practice)
}

// ------------------------------------------------------------------------------------------------
// 2D variant: bins over (x_i, x_j, sf_bin) of a gene pair; code = x_i*r[0] + x_j*r[1] + r0*approx_sf
// (bootstrap.py:62-65 with a two-column expr).  Table layout [sf_bin][x_i][x_j].
template <int CAP, int NT>
__global__ __launch_bounds__(NT) void k_bins_order2d(const uint32_t *__restrict__ tab, const int64_t *__restrict__ tab_ptr,
                                                     const int32_t *__restrict__ xcap_i, const int32_t *__restrict__ xcap_j,
                                                     const int32_t *__restrict__ Karr, const int64_t *__restrict__ pair_list,
                                                     int64_t n_list, int32_t n_groups, int32_t n_sf_bins,
                                                     const double *__restrict__ sf_table, const double *__restrict__ r1a,
                                                     const double *__restrict__ r1b, const double *__restrict__ r0a,
                                                     const int64_t *__restrict__ pair_slot, const int64_t *__restrict__ tile_ptr,
                                                     const double *__restrict__ grp_ncells, double *__restrict__ o_pk,
                                                     double *__restrict__ o_lq, double *__restrict__ o_v1, double *__restrict__ o_v2,
                                                     double *__restrict__ o_a, double *__restrict__ o_b, int32_t *__restrict__ status) {
  extern __shared__ double smem_d[];
  double *code = smem_d;                              // [CAP]
  uint64_t *pay = (uint64_t *)(code + CAP);           // [CAP]  sf_bin << 40 | x_i << 20 | x_j
  uint32_t *mult = (uint32_t *)(pay + CAP);           // [CAP]
  __shared__ int n_found;
  if (blockIdx.x >= n_list) return;
  int64_t p = pair_list[blockIdx.x];
  int64_t slot = pair_slot[p];
  if (slot < 0) return;
  int K = Karr[p];
  if (K > CAP) {
    if (threadIdx.x == 0) atomicOr(status, 2);
    return;
  }
  int grp = (int)(p % n_groups);
  int64_t tp = tab_ptr[p];
  int ci = xcap_i[p], cj = xcap_j[p];
  double ra = r1a[p], rb = r1b[p], r0 = r0a[p];
  int tid = threadIdx.x;
  for (int i = tid; i < CAP; i += NT) code[i] = INFINITY;
  if (tid == 0) n_found = 0;
  __syncthreads();
  if (tid < 64) {
    int total = n_sf_bins * ci * cj;
    int pos = 0;
    for (int i0 = 0; i0 < total; i0 += 64) {
      int i = i0 + tid;
      uint32_t c = i < total ? tab[tp + i] : 0u;
      unsigned long long m = __ballot(c != 0);
      if (c != 0) {
        int at = pos + __popcll(m & ((1ull << tid) - 1ull));
        if (at < CAP) {
          uint32_t xj = (uint32_t)(i % cj), xi = (uint32_t)((i / cj) % ci), bin = (uint32_t)(i / (cj * ci));
          double c1 = (double)xi * ra;
          double c2 = (double)xj * rb;
          double cs = r0 * sf_table[bin];
          code[at] = (c1 + c2) + cs;
          pay[at] = ((uint64_t)bin << 40) | ((uint64_t)xi << 20) | (uint64_t)xj;
          mult[at] = c;
        }
      }
      pos += __popcll(m);
    }
    if (tid == 0) n_found = pos;
  }
  __syncthreads();
  if (n_found != K) {
    if (tid == 0) atomicOr(status, 4);
    return;
  }
  int n = 1;
  while (n < K) n <<= 1;
  for (int k2 = 2; k2 <= n; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < n; i += NT) {
        int ixj = i ^ j;
        if (ixj > i) {
          bool up = (i & k2) == 0;
          double a_ = code[i], b_ = code[ixj];
          if ((a_ > b_) == up && a_ != b_) {
            code[i] = b_;
            code[ixj] = a_;
            uint64_t tpay = pay[i];
            pay[i] = pay[ixj];
            pay[ixj] = tpay;
            uint32_t tm = mult[i];
            mult[i] = mult[ixj];
            mult[ixj] = tm;
          }
        }
      }
      __syncthreads();
    }
  }
  // slot = tile*64 + lane of the 64-wide tile layout, or MM_CHAIN_SLOT | first record: the chain's operands then go, bin by bin, into
  // 8-double records at o_pk + 8*(record + k): pk, lq, x_i, x_j, 1/sf, 1/sf^2 (mm_boot2d_replay_rec)
  const bool recs = (slot & MM_CHAIN_SLOT) != 0;
  int64_t tile = slot >> 6, ln = slot & 63;
  int64_t row0 = recs ? (slot & (MM_CHAIN_SLOT - 1)) : tile_ptr[tile];
  double N = grp_ncells[grp];
  bool tie = false;
  for (int k = tid; k + 1 < K; k += NT)
    if (code[k] == code[k + 1]) tie = true;
  __syncthreads();
  if (tid == 0) {
    double rem = 1.0;
    for (int k = 0; k < K; k++) {
      code[k] = rem;
      rem -= (double)mult[k] / N;
    }
  }
  __syncthreads();
  for (int k = tid; k < K; k += NT) {
    uint32_t bin = (uint32_t)(pay[k] >> 40), xi = (uint32_t)((pay[k] >> 20) & 0xFFFFFu), xj = (uint32_t)(pay[k] & 0xFFFFFu);
    double sf = sf_table[bin];
    double pk = ((double)mult[k] / N) / code[k];
    if (recs) {
      double *rec = o_pk + (row0 + k) * 8;
      rec[0] = pk;
      rec[1] = npyrng::binomial_lq(pk);
      rec[2] = (double)xi;
      rec[3] = (double)xj;
      rec[4] = 1.0 / sf;
      rec[5] = 1.0 / (sf * sf);
      continue;
    }
    int64_t o = (row0 + k) * 64 + ln;
    o_pk[o] = pk;
    o_lq[o] = npyrng::binomial_lq(pk);
    o_v1[o] = (double)xi;
    o_v2[o] = (double)xj;
    o_a[o] = 1.0 / sf;
    o_b[o] = 1.0 / (sf * sf);
  }
  if (tie) atomicOr(status, 8);
}

extern "C" {

int mm_hist1d_sell(const uint32_t *d_ent, const int64_t *d_blk_base, const int32_t *d_slice_w, const int32_t *d_slice_ptr,
                   const int32_t *d_item_ptr, const int32_t *d_perm, const int32_t *d_blk_cell0, const int32_t *d_blk_group,
                   const uint8_t *d_sf_bin, int32_t n_blocks, int32_t n_genes, const int32_t *d_gene_pairbase,
                   const int64_t *d_tab_ptr, const int32_t *d_xcap, uint32_t *d_tab, void *stream) {
  MM_ARG(d_ent && d_blk_base && d_slice_w && d_slice_ptr && d_item_ptr && d_perm && d_blk_cell0 && d_blk_group && d_sf_bin);
  MM_ARG(d_gene_pairbase && d_tab_ptr && d_xcap && d_tab && n_blocks >= 0 && n_genes > 0);
  if (n_blocks == 0) return MM_OK;
  int32_t n_slices = (n_genes + 63) / 64;
  int split = (4096 + n_blocks - 1) / n_blocks;
  if (split < 1) split = 1;
  if (split > 128) split = 128;
  hipLaunchKernelGGL(k_hist1d_sell, dim3((unsigned)(n_blocks * split)), dim3(K5_THREADS), 0, (hipStream_t)stream,
                     (const u32x4 *)d_ent, d_blk_base, d_slice_w, d_slice_ptr, d_item_ptr, d_perm, d_blk_cell0, d_blk_group, d_sf_bin,
                     n_slices, split, d_gene_pairbase, d_tab_ptr, d_xcap, d_tab);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_bins_count(uint32_t *d_tab, const int64_t *d_tab_ptr, const int32_t *d_xcap, int64_t n_pairs, int32_t n_groups,
                  int32_t n_sf_bins, const uint32_t *d_grp_bin_cells, int32_t *d_K, void *stream) {
  MM_ARG(d_tab && d_tab_ptr && d_xcap && d_grp_bin_cells && d_K && n_pairs >= 0 && n_groups > 0 && n_sf_bins > 0 && n_sf_bins <= 256);
  if (n_pairs == 0) return MM_OK;
  int64_t blocks = (n_pairs + 3) / 4;
  hipLaunchKernelGGL(k_bins_count, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_tab, d_tab_ptr, d_xcap, n_pairs,
                     n_groups, n_sf_bins, d_grp_bin_cells, d_K);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_bins_order(const uint32_t *d_tab, const int64_t *d_tab_ptr, const int32_t *d_xcap, const int32_t *d_K,
                  const int64_t *d_pair_list, int64_t n_list, int32_t big, int32_t n_groups, int32_t n_sf_bins,
                  const double *d_sf_table, const double *d_r1, const double *d_r0, const int64_t *d_pair_slot,
                  const int64_t *d_tile_ptr, const double *d_grp_ncells, double *d_pk, double *d_lq, double *d_v,
                  double *d_a, double *d_b, int32_t *d_status, void *stream) {
  MM_ARG(d_tab && d_tab_ptr && d_xcap && d_K && d_pair_list && d_sf_table && d_r1 && d_r0 && d_pair_slot && d_tile_ptr);
  MM_ARG(d_grp_ncells && d_pk && d_lq && d_v && d_a && d_b && d_status && n_list >= 0 && n_sf_bins <= 256);
  if (n_list == 0) return MM_OK;
  if (!big) {
    constexpr int CAP = 1024, NT = 64;
    size_t shm = (size_t)CAP * 16;
    hipLaunchKernelGGL((k_bins_order<CAP, NT>), dim3((unsigned)n_list), dim3(NT), shm, (hipStream_t)stream, d_tab, d_tab_ptr, d_xcap,
                       d_K, d_pair_list, n_list, n_groups, n_sf_bins, d_sf_table, d_r1, d_r0, d_pair_slot, d_tile_ptr, d_grp_ncells,
                       d_pk, d_lq, d_v, d_a, d_b, d_status);
  } else {
    constexpr int CAP = 8192, NT = 512;
    size_t shm = (size_t)CAP * 16;
    MM_HIP(hipFuncSetAttribute((const void *)k_bins_order<CAP, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL((k_bins_order<CAP, NT>), dim3((unsigned)n_list), dim3(NT), shm, (hipStream_t)stream, d_tab, d_tab_ptr, d_xcap,
                       d_K, d_pair_list, n_list, n_groups, n_sf_bins, d_sf_table, d_r1, d_r0, d_pair_slot, d_tile_ptr, d_grp_ncells,
                       d_pk, d_lq, d_v, d_a, d_b, d_status);
  }
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_bins_order2d(const uint32_t *d_tab, const int64_t *d_tab_ptr, const int32_t *d_xcap_i, const int32_t *d_xcap_j,
                    const int32_t *d_K, const int64_t *d_pair_list, int64_t n_list, int32_t big, int32_t n_groups, int32_t n_sf_bins,
                    const double *d_sf_table, const double *d_r1a, const double *d_r1b, const double *d_r0,
                    const int64_t *d_pair_slot, const int64_t *d_tile_ptr, const double *d_grp_ncells, double *d_pk, double *d_lq,
                    double *d_v1, double *d_v2, double *d_a, double *d_b, int32_t *d_status, void *stream) {
  MM_ARG(d_tab && d_tab_ptr && d_xcap_i && d_xcap_j && d_K && d_pair_list && d_sf_table && d_r1a && d_r1b && d_r0 && d_pair_slot);
  MM_ARG(d_tile_ptr && d_grp_ncells && d_pk && d_lq && d_v1 && d_v2 && d_a && d_b && d_status && n_list >= 0 && n_sf_bins <= 256);
  if (n_list == 0) return MM_OK;
  if (!big) {
    constexpr int CAP = 1024, NT = 64;
    size_t shm = (size_t)CAP * 20;
    hipLaunchKernelGGL((k_bins_order2d<CAP, NT>), dim3((unsigned)n_list), dim3(NT), shm, (hipStream_t)stream, d_tab, d_tab_ptr, d_xcap_i,
                       d_xcap_j, d_K, d_pair_list, n_list, n_groups, n_sf_bins, d_sf_table, d_r1a, d_r1b, d_r0, d_pair_slot, d_tile_ptr,
                       d_grp_ncells, d_pk, d_lq, d_v1, d_v2, d_a, d_b, d_status);
  } else {
    constexpr int CAP = 4096, NT = 512;  // 20 B per bin: 80 KiB of LDS
    size_t shm = (size_t)CAP * 20;
    MM_HIP(hipFuncSetAttribute((const void *)k_bins_order2d<CAP, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL((k_bins_order2d<CAP, NT>), dim3((unsigned)n_list), dim3(NT), shm, (hipStream_t)stream, d_tab, d_tab_ptr, d_xcap_i,
                       d_xcap_j, d_K, d_pair_list, n_list, n_groups, n_sf_bins, d_sf_table, d_r1a, d_r1b, d_r0, d_pair_slot, d_tile_ptr,
                       d_grp_ncells, d_pk, d_lq, d_v1, d_v2, d_a, d_b, d_status);
  }
  MM_LAUNCH_CHECK();
  return MM_OK;
}

}  // extern "C"
