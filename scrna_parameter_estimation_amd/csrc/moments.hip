// moments.hip -- K1+K2: per-gene 1D moment sums from the SELL count blocks.  THE HBM-roofline kernel.
//
// Reference behaviour replaced (memento/estimator.py:177-180, sparse branch of _hyper_1d_relative):
//     S1 = sum_c x/sf_c,  S2 = sum_c x^2/sf_c^2,  S3 = sum_c x/sf_c^2    (three CSC.vector products)
// and memento/main.py:201, :206 (plain per-group mean and max of the counts, for the gene filters).
//
// Mapping: one lane per gene of a 64-gene slice, so the 5 accumulators live in registers and there is NO
// cross-lane reduction; a wave streams its work item (<= 64 rows of 1 KiB, one dwordx4 per lane per row,
// perfectly coalesced); the block's per-cell 1/sf (fp64) is staged once in LDS and gathered per entry.
// Algorithmic bytes per entry: 4 (the packed entry); fp64 accumulation.
#include "mm_common.h"

#ifndef K1_THREADS
#define K1_THREADS 1024
#endif
#ifndef K1_UNROLL
#define K1_UNROLL 8
#endif
#define K1_MAX_SLICES 1024  // G <= 65536
#ifndef K1_PIPE
#define K1_PIPE 0
#endif
#ifndef K1_WGS
#define K1_WGS 2048  // target workgroups per launch
#endif

__device__ __forceinline__ void k1_acc(uint32_t e, const double *__restrict__ w_lds, double &a1, double &a2, double &a3,
                                       uint32_t &sx, uint32_t &mx) {
  uint32_t x = e >> MM_CELL_BITS;
  double w = w_lds[e & (MM_BLOCK_CELLS - 1)];
  double xd = (double)x;
  double xw = xd * w;
  double xw2 = xw * w;
  a1 += xw;
  a3 += xw2;
  a2 += xd * xw2;
  sx += x;
  mx = max(mx, x);
}

#ifndef K1_MIN_WAVES
#define K1_MIN_WAVES 1
#endif
__global__ __launch_bounds__(K1_THREADS, K1_MIN_WAVES) void k_moments1d_sell(const u32x4 *__restrict__ ent, const int64_t *__restrict__ blk_base,
                                                               const int32_t *__restrict__ slice_w, const int32_t *__restrict__ slice_ptr,
                                                               const int32_t *__restrict__ item_ptr, const int64_t *__restrict__ blk_item_base,
                                                               const int32_t *__restrict__ blk_cell0, const double *__restrict__ inv_sf,
                                                               int32_t n_slices, int32_t split, int32_t n_wg, u32x4 *__restrict__ slab) {
  __shared__ double w_lds[MM_BLOCK_CELLS];
  __shared__ int32_t ip[K1_MAX_SLICES + 1], sp[K1_MAX_SLICES + 1], sw[K1_MAX_SLICES];
#ifndef K1_XCD_REMAP
#define K1_XCD_REMAP 1
#endif
  // Workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), each with its own L2.  Renumber them so that the `split`
  // workgroups of one count block run on ONE XCD: its 1/size-factor vector and slice tables are then fetched into one L2
  // instead of eight.  The grid is padded to a multiple of 8; ids past the real count exit.
  int wg = blockIdx.x;
  if (K1_XCD_REMAP) {
    int per = gridDim.x >> 3;
    wg = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  }
  if (wg >= n_wg) return;
  int b = wg / split, part = wg % split;
  int c0 = blk_cell0[b], nc = blk_cell0[b + 1] - c0;
  for (int i = threadIdx.x; i < MM_BLOCK_CELLS; i += K1_THREADS) w_lds[i] = i < nc ? inv_sf[c0 + i] : 0.0;
  // the block's slice tables go to LDS too: the item -> slice lookup must not be a chain of global loads
  for (int i = threadIdx.x; i <= n_slices; i += K1_THREADS) {
    ip[i] = item_ptr[(int64_t)b * (n_slices + 1) + i];
    sp[i] = slice_ptr[(int64_t)b * (n_slices + 1) + i];
    if (i < n_slices) sw[i] = slice_w[(int64_t)b * n_slices + i];
  }
  __syncthreads();
  int lane = mm_lane();
  int wave = part * (K1_THREADS / 64) + (threadIdx.x >> 6);
  int nwaves = split * (K1_THREADS / 64);
  int n_items = ip[n_slices];
  int64_t base = blk_base[b];
  int64_t ibase = blk_item_base[b];
  // items are numbered slice-major; walk slices, taking this wave's share of the item ids
  for (int item = wave; item < n_items; item += nwaves) {
    int lo = 0, hi = n_slices;  // last slice t with ip[t] <= item (wave-uniform binary search in LDS)
    while (hi - lo > 1) {
      int mid = (lo + hi) >> 1;
      if (ip[mid] <= item) lo = mid; else hi = mid;
    }
    int t = lo;
    int k = item - ip[t];
    int r0 = k * MM_ITEM_ROWS;
    int r1 = min(sw[t], r0 + MM_ITEM_ROWS);
    const u32x4 *p = ent + (base + sp[t] + r0) * 64 + lane;
    double a1 = 0.0, a2 = 0.0, a3 = 0.0;
    uint32_t sx = 0, mx = 0;
    int nr = r1 - r0, r = 0;
#if K1_PIPE
    // two-stage software pipeline: the loads of batch i+1 are in flight while batch i is consumed
    u32x4 ea[K1_UNROLL], eb[K1_UNROLL];
    int nb = nr / K1_UNROLL;
    if (nb > 0) {
#pragma unroll
      for (int u = 0; u < K1_UNROLL; u++) ea[u] = __builtin_nontemporal_load(p + (int64_t)u * 64);
    }
    for (int bi = 0; bi < nb; bi += 2) {
      if (bi + 1 < nb) {
#pragma unroll
        for (int u = 0; u < K1_UNROLL; u++) eb[u] = __builtin_nontemporal_load(p + (int64_t)((bi + 1) * K1_UNROLL + u) * 64);
      }
#pragma unroll
      for (int u = 0; u < K1_UNROLL; u++) {
        k1_acc(ea[u].x, w_lds, a1, a2, a3, sx, mx);
        k1_acc(ea[u].y, w_lds, a1, a2, a3, sx, mx);
        k1_acc(ea[u].z, w_lds, a1, a2, a3, sx, mx);
        k1_acc(ea[u].w, w_lds, a1, a2, a3, sx, mx);
      }
      if (bi + 1 < nb) {
        if (bi + 2 < nb) {
#pragma unroll
          for (int u = 0; u < K1_UNROLL; u++) ea[u] = __builtin_nontemporal_load(p + (int64_t)((bi + 2) * K1_UNROLL + u) * 64);
        }
#pragma unroll
        for (int u = 0; u < K1_UNROLL; u++) {
          k1_acc(eb[u].x, w_lds, a1, a2, a3, sx, mx);
          k1_acc(eb[u].y, w_lds, a1, a2, a3, sx, mx);
          k1_acc(eb[u].z, w_lds, a1, a2, a3, sx, mx);
          k1_acc(eb[u].w, w_lds, a1, a2, a3, sx, mx);
        }
      }
    }
    r = nb * K1_UNROLL;
#else
    for (; r + K1_UNROLL <= nr; r += K1_UNROLL) {
      u32x4 e[K1_UNROLL];
#pragma unroll
      for (int u = 0; u < K1_UNROLL; u++) e[u] = __builtin_nontemporal_load(p + (int64_t)(r + u) * 64);
#pragma unroll
      for (int u = 0; u < K1_UNROLL; u++) {
        k1_acc(e[u].x, w_lds, a1, a2, a3, sx, mx);
        k1_acc(e[u].y, w_lds, a1, a2, a3, sx, mx);
        k1_acc(e[u].z, w_lds, a1, a2, a3, sx, mx);
        k1_acc(e[u].w, w_lds, a1, a2, a3, sx, mx);
      }
    }
#endif
    if (r < nr) {
      // the last (nr mod UNROLL) rows in ONE batch as well: the row count is wave-uniform, so the guards are scalar branches and
      // all loads are in flight together (one row at a time would pay the full memory latency up to UNROLL-1 times per item);
      // a zero word is padding and adds nothing
      u32x4 e[K1_UNROLL];
#pragma unroll
      for (int u = 0; u < K1_UNROLL; u++) {
        e[u] = u32x4{0u, 0u, 0u, 0u};
        if (r + u < nr) e[u] = __builtin_nontemporal_load(p + (int64_t)(r + u) * 64);
      }
#pragma unroll
      for (int u = 0; u < K1_UNROLL; u++) {
        if (r + u < nr) {
          k1_acc(e[u].x, w_lds, a1, a2, a3, sx, mx);
          k1_acc(e[u].y, w_lds, a1, a2, a3, sx, mx);
          k1_acc(e[u].z, w_lds, a1, a2, a3, sx, mx);
          k1_acc(e[u].w, w_lds, a1, a2, a3, sx, mx);
        }
      }
    }
    // one 32-byte record per (item, lane): {S1, S2 | S3, sum x, max x}; a wave writes ONE contiguous 2 KiB run per work item
    // (two dwordx4 stores per lane).  Exchanging words between lanes so that each store instruction covers a fully contiguous
    // 1 KiB was measured too: 1.4 % slower (profiles/README.md).
    u32x4 *rec = slab + ((ibase + item) * 64 + lane) * 2;
    u32x4 r0v, r1v;
    r0v.x = (uint32_t)__double2loint(a1); r0v.y = (uint32_t)__double2hiint(a1);
    r0v.z = (uint32_t)__double2loint(a2); r0v.w = (uint32_t)__double2hiint(a2);
    r1v.x = (uint32_t)__double2loint(a3); r1v.y = (uint32_t)__double2hiint(a3);
    r1v.z = sx; r1v.w = mx;
    rec[0] = r0v;
    rec[1] = r1v;
  }
}

// Deterministic reduction over the items of a gene's slice and over the blocks of a group.
__global__ __launch_bounds__(256) void k_moments1d_reduce(const u32x4 *__restrict__ slab, const int32_t *__restrict__ rank,
                                                          const int32_t *__restrict__ item_ptr, const int64_t *__restrict__ blk_item_base,
                                                          const int32_t *__restrict__ grp_blk0, int32_t n_groups, int32_t n_genes,
                                                          int32_t n_slices, double *__restrict__ out_S, uint64_t *__restrict__ out_sumx,
                                                          uint32_t *__restrict__ out_maxx) {
  int g = blockIdx.x * blockDim.x + threadIdx.x;
  int grp = blockIdx.y;
  if (g >= n_genes) return;
  double a1 = 0.0, a2 = 0.0, a3 = 0.0;
  uint64_t sx = 0;
  uint32_t mx = 0;
  for (int b = grp_blk0[grp]; b < grp_blk0[grp + 1]; b++) {
    int s = rank[(int64_t)b * n_genes + g];
    int t = s >> 6, ln = s & 63;
    const int32_t *ip = item_ptr + (int64_t)b * (n_slices + 1);
    int64_t ib = blk_item_base[b];
    for (int it = ip[t]; it < ip[t + 1]; it++) {
      const u32x4 *rec = slab + ((ib + it) * 64 + ln) * 2;
      u32x4 lo = rec[0], hi = rec[1];
      a1 += __hiloint2double((int)lo.y, (int)lo.x);
      a2 += __hiloint2double((int)lo.w, (int)lo.z);
      a3 += __hiloint2double((int)hi.y, (int)hi.x);
      sx += hi.z;
      mx = max(mx, hi.w);
    }
  }
  int64_t o = (int64_t)grp * n_genes + g;
  int64_t plane = (int64_t)n_groups * n_genes;
  out_S[o] = a1;
  out_S[plane + o] = a2;
  out_S[2 * plane + o] = a3;
  out_sumx[o] = sx;
  out_maxx[o] = mx;
}

extern "C" {

int mm_moments1d_sell(const uint32_t *d_ent, const int64_t *d_blk_base, const int32_t *d_slice_w, const int32_t *d_slice_ptr,
                      const int32_t *d_item_ptr, const int64_t *d_blk_item_base, const int32_t *d_blk_cell0,
                      const double *d_inv_sf, int32_t n_blocks, int32_t n_genes, void *d_slab, void *stream) {
  MM_ARG(d_ent && d_blk_base && d_slice_w && d_slice_ptr && d_item_ptr && d_blk_item_base && d_blk_cell0 && d_inv_sf);
  MM_ARG(d_slab && n_blocks >= 0 && n_genes > 0 && n_genes <= 64 * K1_MAX_SLICES);
  if (n_blocks == 0) return MM_OK;
  int32_t n_slices = (n_genes + 63) / 64;
  // enough workgroups to fill 256 CUs x 2 resident (64 KiB LDS each)
  int split = (K1_WGS + n_blocks - 1) / n_blocks;
  if (split < 1) split = 1;
  if (split > 64) split = 64;
  int n_wg = n_blocks * split;
  hipLaunchKernelGGL(k_moments1d_sell, dim3((unsigned)((n_wg + 7) / 8 * 8)), dim3(K1_THREADS), 0, (hipStream_t)stream,
                     (const u32x4 *)d_ent, d_blk_base, d_slice_w, d_slice_ptr, d_item_ptr, d_blk_item_base, d_blk_cell0, d_inv_sf,
                     n_slices, split, n_wg, (u32x4 *)d_slab);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_moments1d_reduce(const void *d_slab, const int32_t *d_rank, const int32_t *d_item_ptr, const int64_t *d_blk_item_base,
                        const int32_t *d_grp_blk0, int32_t n_groups, int32_t n_genes, double *d_out_S, uint64_t *d_out_sumx,
                        uint32_t *d_out_maxx, void *stream) {
  MM_ARG(d_slab && d_rank && d_item_ptr && d_blk_item_base && d_grp_blk0);
  MM_ARG(d_out_S && d_out_sumx && d_out_maxx && n_groups > 0 && n_genes > 0);
  int32_t n_slices = (n_genes + 63) / 64;
  hipLaunchKernelGGL(k_moments1d_reduce, dim3((n_genes + 255) / 256, n_groups), dim3(256), 0, (hipStream_t)stream, (const u32x4 *)d_slab,
                     d_rank, d_item_ptr, d_blk_item_base, d_grp_blk0, n_groups, n_genes, n_slices, d_out_S,
                     d_out_sumx, d_out_maxx);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

}  // extern "C"
