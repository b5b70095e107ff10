// pairs.hip -- K11: gene-pair (2D) sums and histograms from the SELL count blocks.
//
// Reference behaviour replaced:
//   estimator._hyper_cov_relative sparse branch   memento/estimator.py:220-231
//       prod_p = sum_c x_ci x_cj / sf_c^2   (the only new O(nnz) quantity; the means and the i==j
//       correction come from the 1D sums S1, S3 of K1)
//   estimator._hyper_corr_symmetric (all-by-all)   memento/estimator.py:253-257  -- same kernel, all pairs
//   bootstrap._unique_expr on two columns          memento/bootstrap.py:62-71 (the (x_i, x_j, sf_bin) bins)
//
// Two steps.  (1) k_extract_cols copies the columns of the genes that occur in the pair list out of the
// SELL blocks into a gene-contiguous store (one pass, lane-per-gene, coalesced reads).  (2) one
// workgroup per (block, left gene): the left gene's column is scattered into a dense per-cell LDS vector
// (the join needs no sorted columns), then one wave per partner streams the partner's contiguous column,
// gathers from LDS and reduces with __shfl_xor.  No MFMA: the operands are ~3 % dense.
#include "mm_common.h"

// ------------------------------------------------------------------------------------------------
// (1) column extraction.  col_id[gene] = column slot m or -1; col_ptr[b][m] = first entry of (block, m).
#define KX_THREADS 256
__global__ __launch_bounds__(KX_THREADS) void k_extract_cols(const u32x4 *__restrict__ ent, const int64_t *__restrict__ blk_base,
                                                             const int32_t *__restrict__ slice_w, const int32_t *__restrict__ slice_ptr,
                                                             const int32_t *__restrict__ item_ptr, const int32_t *__restrict__ perm,
                                                             int32_t n_slices, int32_t split, const int32_t *__restrict__ col_id,
                                                             int32_t n_cols, const int64_t *__restrict__ col_ptr,
                                                             uint32_t *__restrict__ out) {
  __shared__ int32_t ip[1025];
  int b = blockIdx.x / split, part = blockIdx.x % split;
  for (int i = threadIdx.x; i <= n_slices; i += KX_THREADS) ip[i] = item_ptr[(int64_t)b * (n_slices + 1) + i];
  __syncthreads();
  int lane = mm_lane();
  int wave = part * (KX_THREADS / 64) + (threadIdx.x >> 6);
  int nwaves = split * (KX_THREADS / 64);
  const int32_t *sw = slice_w + (int64_t)b * n_slices;
  const int32_t *sp = slice_ptr + (int64_t)b * (n_slices + 1);
  const int32_t *pm = perm + (int64_t)b * n_slices * 64;
  int n_items = ip[n_slices];
  int64_t base = blk_base[b];
  for (int item = wave; item < n_items; item += nwaves) {
    int lo = 0, hi = n_slices;
    while (hi - lo > 1) {
      int mid = (lo + hi) >> 1;
      if (ip[mid] <= item) lo = mid; else hi = mid;
    }
    int t = lo;
    int gene = pm[t * 64 + lane];
    int m = gene >= 0 ? col_id[gene] : -1;
    if (__ballot(m >= 0) == 0ull) continue;
    int k = item - ip[t];
    int r0 = k * MM_ITEM_ROWS;
    int r1 = min(sw[t], r0 + MM_ITEM_ROWS);
    if (m < 0) continue;
    uint32_t *dst = out + col_ptr[(int64_t)b * n_cols + m];
    const u32x4 *p = ent + (base + sp[t] + r0) * 64 + lane;
    for (int r = r0; r < r1; r++) {
      u32x4 e = p[(int64_t)(r - r0) * 64];
      // entries of a gene are packed front-to-back, so the stored position equals the running index
      if (e.x) dst[4 * r + 0] = e.x;
      if (e.y) dst[4 * r + 1] = e.y;
      if (e.z) dst[4 * r + 2] = e.z;
      if (e.w) dst[4 * r + 3] = e.w;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// (2) per (block, left gene): cross sums  prod[b][pair] = sum_c x_i x_j / sf^2  and/or 2D histograms.
// left_ptr[l] .. left_ptr[l+1] = the pairs whose left column is left_col[l]; right_col[pair] = partner.
#define KP_THREADS 512
__global__ __launch_bounds__(KP_THREADS) void k_pair_cross(const uint32_t *__restrict__ cols, const int64_t *__restrict__ col_ptr,
                                                           int32_t n_cols, const int32_t *__restrict__ blk_cell0,
                                                           const double *__restrict__ inv_sf, const int32_t *__restrict__ left_col,
                                                           const int64_t *__restrict__ left_ptr, const int32_t *__restrict__ right_col,
                                                           int64_t n_pairs, double *__restrict__ prod /* [nb][n_pairs] */) {
  __shared__ double xw2[MM_BLOCK_CELLS];  // x_i / sf^2 per cell of the block (0 where gene i is not expressed)
  int b = blockIdx.y;
  int l = blockIdx.x;
  int c0 = blk_cell0[b];
  for (int i = threadIdx.x; i < MM_BLOCK_CELLS; i += KP_THREADS) xw2[i] = 0.0;
  __syncthreads();
  int ci = left_col[l];
  const int64_t *cp = col_ptr + (int64_t)b * (n_cols + 1);
  for (int64_t e = cp[ci] + threadIdx.x; e < cp[ci + 1]; e += KP_THREADS) {
    uint32_t v = cols[e];
    uint32_t cell = v & (MM_BLOCK_CELLS - 1);
    double w = inv_sf[c0 + cell];
    xw2[cell] = ((double)(v >> MM_CELL_BITS) * w) * w;
  }
  __syncthreads();
  int lane = mm_lane(), wave = threadIdx.x >> 6;
  for (int64_t p = left_ptr[l] + wave; p < left_ptr[l + 1]; p += KP_THREADS / 64) {
    int cj = right_col[p];
    double acc = 0.0;
    for (int64_t e = cp[cj] + lane; e < cp[cj + 1]; e += 64) {
      uint32_t v = cols[e];
      acc += (double)(v >> MM_CELL_BITS) * xw2[v & (MM_BLOCK_CELLS - 1)];
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) prod[(int64_t)b * n_pairs + p] = acc;
  }
}

// deterministic sum of the per-block partials over the blocks of each group -> [n_groups][n_pairs]
__global__ __launch_bounds__(256) void k_pair_reduce(const double *__restrict__ prod, const int32_t *__restrict__ grp_blk0,
                                                     int64_t n_pairs, double *__restrict__ out) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int grp = blockIdx.y;
  if (p >= n_pairs) return;
  double acc = 0.0;
  for (int b = grp_blk0[grp]; b < grp_blk0[grp + 1]; b++) acc += prod[(int64_t)b * n_pairs + p];
  out[(int64_t)grp * n_pairs + p] = acc;
}

// 2D histogram: table of (pair, group) q = pair*n_groups + group is [n_sf_bins][xcap_i][xcap_j] at tab_ptr[q];
// this kernel counts the cells with x_j > 0 (x_i may be 0); the x_j == 0 column is derived afterwards.
__global__ __launch_bounds__(KP_THREADS) void k_pair_hist(const uint32_t *__restrict__ cols, const int64_t *__restrict__ col_ptr,
                                                          int32_t n_cols, const int32_t *__restrict__ blk_cell0,
                                                          const int32_t *__restrict__ blk_group, const uint8_t *__restrict__ sf_bin,
                                                          const int32_t *__restrict__ left_col, const int64_t *__restrict__ left_ptr,
                                                          const int32_t *__restrict__ right_col, int32_t n_groups,
                                                          const int64_t *__restrict__ tab_ptr, const int32_t *__restrict__ xcap_i,
                                                          const int32_t *__restrict__ xcap_j, uint32_t *__restrict__ tab) {
  __shared__ uint32_t xi[MM_BLOCK_CELLS];
  __shared__ uint8_t bins[MM_BLOCK_CELLS];
  int b = blockIdx.y, l = blockIdx.x;
  int c0 = blk_cell0[b], nc = blk_cell0[b + 1] - c0;
  int grp = blk_group[b];
  for (int i = threadIdx.x; i < MM_BLOCK_CELLS; i += KP_THREADS) {
    xi[i] = 0;
    bins[i] = i < nc ? sf_bin[c0 + i] : 0;
  }
  __syncthreads();
  int ci = left_col[l];
  const int64_t *cp = col_ptr + (int64_t)b * (n_cols + 1);
  for (int64_t e = cp[ci] + threadIdx.x; e < cp[ci + 1]; e += KP_THREADS) {
    uint32_t v = cols[e];
    xi[v & (MM_BLOCK_CELLS - 1)] = v >> MM_CELL_BITS;
  }
  __syncthreads();
  int lane = mm_lane(), wave = threadIdx.x >> 6;
  for (int64_t p = left_ptr[l] + wave; p < left_ptr[l + 1]; p += KP_THREADS / 64) {
    int cj = right_col[p];
    int64_t q = p * n_groups + grp;
    int64_t tp = tab_ptr[q];
    uint32_t ci_cap = (uint32_t)xcap_i[q], cj_cap = (uint32_t)xcap_j[q];
    for (int64_t e = cp[cj] + lane; e < cp[cj + 1]; e += 64) {
      uint32_t v = cols[e];
      uint32_t cell = v & (MM_BLOCK_CELLS - 1), xj = v >> MM_CELL_BITS, xa = xi[cell];
      if (xa < ci_cap && xj < cj_cap) atomicAdd(&tab[tp + ((int64_t)bins[cell] * ci_cap + xa) * cj_cap + xj], 1u);
    }
  }
}

// Wave per (pair, group): derive the x_j == 0 column from the left gene's 1D histogram
// (hist_i[q] -> [n_sf_bins][xcap_i], with column 0 already holding the zero-count cells) and count bins.
__global__ __launch_bounds__(256) void k_pair_bins_count(uint32_t *__restrict__ tab, const int64_t *__restrict__ tab_ptr,
                                                         const int32_t *__restrict__ xcap_i, const int32_t *__restrict__ xcap_j,
                                                         const uint32_t *__restrict__ hist_i, const int64_t *__restrict__ hist_ptr,
                                                         int64_t n_q, int32_t n_sf_bins, int32_t *__restrict__ K) {
  int lane = mm_lane();
  int64_t q = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (q >= n_q) return;
  int ci = xcap_i[q], cj = xcap_j[q];
  uint32_t *t = tab + tab_ptr[q];
  const uint32_t *h = hist_i + hist_ptr[q];
  int k = 0;
  for (int bin = 0; bin < n_sf_bins; bin++) {
    for (int xa = 0; xa < ci; xa++) {
      uint32_t *row = t + ((int64_t)bin * ci + xa) * cj;
      uint32_t s = 0;
      int nz = 0;
      for (int x = 1 + lane; x < cj; x += 64) {
        uint32_t c = row[x];
        s += c;
        nz += c != 0;
      }
      for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_xor(s, off, 64);
        nz += __shfl_xor(nz, off, 64);
      }
      uint32_t zero = h[bin * ci + xa] - s;  // cells with x_i == xa in this sf bin whose x_j is 0
      if (lane == 0) row[0] = zero;
      k += nz + (zero != 0);
    }
  }
  if (lane == 0) K[q] = k;
}

extern "C" {

int mm_extract_cols(const uint32_t *d_ent, const int64_t *d_blk_base, const int32_t *d_slice_w, const int32_t *d_slice_ptr,
                    const int32_t *d_item_ptr, const int32_t *d_perm, int32_t n_blocks, int32_t n_genes, const int32_t *d_col_id,
                    int32_t n_cols, const int64_t *d_col_ptr, uint32_t *d_out, void *stream) {
  MM_ARG(d_ent && d_blk_base && d_slice_w && d_slice_ptr && d_item_ptr && d_perm && d_col_id && d_col_ptr && d_out);
  MM_ARG(n_blocks >= 0 && n_genes > 0 && n_genes <= 65536 && n_cols > 0);
  if (n_blocks == 0) return MM_OK;
  int32_t n_slices = (n_genes + 63) / 64;
  int split = (4096 + n_blocks - 1) / n_blocks;
  if (split < 1) split = 1;
  if (split > 128) split = 128;
  // col_ptr rows have n_cols+1 entries; the kernel indexes [b*n_cols + m] into a view without the end markers
  hipLaunchKernelGGL(k_extract_cols, dim3((unsigned)(n_blocks * split)), dim3(KX_THREADS), 0, (hipStream_t)stream, (const u32x4 *)d_ent,
                     d_blk_base, d_slice_w, d_slice_ptr, d_item_ptr, d_perm, n_slices, split, d_col_id, n_cols + 1, d_col_ptr, d_out);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_pair_cross(const uint32_t *d_cols, const int64_t *d_col_ptr, int32_t n_cols, const int32_t *d_blk_cell0,
                  const int32_t *d_grp_blk0, int32_t n_blocks, int32_t n_groups, const double *d_inv_sf, const int32_t *d_left_col,
                  const int64_t *d_left_ptr, int32_t n_left, const int32_t *d_right_col, int64_t n_pairs, double *d_scratch,
                  double *d_out, void *stream) {
  MM_ARG(d_cols && d_col_ptr && d_blk_cell0 && d_grp_blk0 && d_inv_sf && d_left_col && d_left_ptr && d_right_col && d_scratch && d_out);
  MM_ARG(n_blocks > 0 && n_groups > 0 && n_left >= 0 && n_pairs >= 0 && n_blocks <= 65535);
  if (n_left == 0 || n_pairs == 0) return MM_OK;
  hipLaunchKernelGGL(k_pair_cross, dim3((unsigned)n_left, (unsigned)n_blocks), dim3(KP_THREADS), 0, (hipStream_t)stream, d_cols, d_col_ptr,
                     n_cols, d_blk_cell0, d_inv_sf, d_left_col, d_left_ptr, d_right_col, n_pairs, d_scratch);
  MM_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_pair_reduce, dim3((unsigned)((n_pairs + 255) / 256), (unsigned)n_groups), dim3(256), 0, (hipStream_t)stream,
                     d_scratch, d_grp_blk0, n_pairs, d_out);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_pair_hist(const uint32_t *d_cols, const int64_t *d_col_ptr, int32_t n_cols, const int32_t *d_blk_cell0,
                 const int32_t *d_blk_group, int32_t n_blocks, const uint8_t *d_sf_bin, const int32_t *d_left_col,
                 const int64_t *d_left_ptr, int32_t n_left, const int32_t *d_right_col, int32_t n_groups, const int64_t *d_tab_ptr,
                 const int32_t *d_xcap_i, const int32_t *d_xcap_j, uint32_t *d_tab, void *stream) {
  MM_ARG(d_cols && d_col_ptr && d_blk_cell0 && d_blk_group && d_sf_bin && d_left_col && d_left_ptr && d_right_col && d_tab_ptr);
  MM_ARG(d_xcap_i && d_xcap_j && d_tab && n_blocks > 0 && n_blocks <= 65535 && n_groups > 0 && n_left >= 0);
  if (n_left == 0) return MM_OK;
  hipLaunchKernelGGL(k_pair_hist, dim3((unsigned)n_left, (unsigned)n_blocks), dim3(KP_THREADS), 0, (hipStream_t)stream, d_cols, d_col_ptr,
                     n_cols, d_blk_cell0, d_blk_group, d_sf_bin, d_left_col, d_left_ptr, d_right_col, n_groups, d_tab_ptr, d_xcap_i,
                     d_xcap_j, d_tab);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_pair_bins_count(uint32_t *d_tab, const int64_t *d_tab_ptr, const int32_t *d_xcap_i, const int32_t *d_xcap_j,
                       const uint32_t *d_hist_i, const int64_t *d_hist_ptr, int64_t n_q, int32_t n_sf_bins, int32_t *d_K,
                       void *stream) {
  MM_ARG(d_tab && d_tab_ptr && d_xcap_i && d_xcap_j && d_hist_i && d_hist_ptr && d_K && n_q >= 0 && n_sf_bins > 0 && n_sf_bins <= 256);
  if (n_q == 0) return MM_OK;
  hipLaunchKernelGGL(k_pair_bins_count, dim3((unsigned)((n_q + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_tab, d_tab_ptr, d_xcap_i,
                     d_xcap_j, d_hist_i, d_hist_ptr, n_q, n_sf_bins, d_K);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

}  // extern "C"
