// ingest.hip -- K0: CSR -> group-ordered SELL-64x4 "count blocks", and K3 row sums.
//
// Reference behaviour replaced: util._select_cells -> adata.X[mask].tocsc() per group
// (memento/util.py:8-13, main.py:128) and X.sum(axis=1) / X.multiply(mask).sum(axis=1)
// (memento/estimator.py:65, :73).  Everything here is integer/index work: bit-exact by construction.
#include "mm_common.h"

// ------------------------------------------------------------------------------------------------
// K3: one wave per CSR row, coalesced index/data reads, __shfl_xor reduction (HBM-bound).
__global__ __launch_bounds__(256) void k_csr_rowsum(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                    const float *__restrict__ data, int64_t n_rows,
                                                    const uint8_t *__restrict__ mask, double *__restrict__ out) {
  int lane = mm_lane();
  int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < n_rows; r += nwaves) {
    int64_t s = indptr[r], e = indptr[r + 1];
    double acc = 0.0;
    for (int64_t i = s + lane; i < e; i += 64) {
      float x = data[i];
      if (mask == nullptr || mask[indices[i]]) acc += (double)x;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) out[r] = acc;
  }
}

// ------------------------------------------------------------------------------------------------
// Gene sharding on the device (multi-GPU, one process per GPU: every rank keeps all cells x ITS genes): the column range
// [lo, hi) of the resident CSR as a new CSR with renumbered columns.  Replaces the host-side X[:, lo:hi] (scipy, O(nnz))
// the sharded drivers had to do before.  Pass 1 counts per row, the caller turns the counts into row pointers (exclusive
// scan), pass 2 writes the surviving entries in their original order (ballot compaction, one wave per row).
__global__ __launch_bounds__(256) void k_csr_colcount(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                      int64_t n_rows, int32_t lo, int32_t hi, int64_t *__restrict__ row_nnz) {
  int lane = mm_lane();
  int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < n_rows; r += nwaves) {
    int64_t s = indptr[r], e = indptr[r + 1];
    int cnt = 0;
    for (int64_t i = s + lane; i < e; i += 64) {
      int g = indices[i];
      cnt += (g >= lo && g < hi) ? 1 : 0;
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (lane == 0) row_nnz[r] = cnt;
  }
}

__global__ __launch_bounds__(256) void k_csr_colsplit(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                      const float *__restrict__ data, int64_t n_rows, int32_t lo, int32_t hi,
                                                      const int64_t *__restrict__ out_indptr, int32_t *__restrict__ out_indices,
                                                      float *__restrict__ out_data) {
  int lane = mm_lane();
  int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < n_rows; r += nwaves) {
    int64_t s = indptr[r], e = indptr[r + 1];
    int64_t o = out_indptr[r];
    for (int64_t i0 = s; i0 < e; i0 += 64) {   // wave-uniform trip count: every lane reaches the ballot
      int64_t i = i0 + lane;
      int g = i < e ? indices[i] : -1;
      bool keep = g >= lo && g < hi;
      uint64_t bal = __ballot(keep);
      if (keep) {
        int64_t pos = o + __popcll(bal & ((1ull << lane) - 1ull));
        out_indices[pos] = g - lo;
        out_data[pos] = data[i];
      }
      o += __popcll(bal);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K0 step 1: nnz per (block, gene) with LDS counters; validates the counts.
#define CNT_TILE 32768
__global__ __launch_bounds__(1024) void k_sell_count(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                     const float *__restrict__ data, const int32_t *__restrict__ cell_order,
                                                     const int32_t *__restrict__ blk_cell0, int32_t n_genes,
                                                     uint16_t *__restrict__ blk_cnt, int32_t *__restrict__ status) {
  __shared__ uint32_t cnt[CNT_TILE];
  int b = blockIdx.x;
  int c0 = blk_cell0[b], c1 = blk_cell0[b + 1];
  int lane = mm_lane(), wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  int bad = 0;
  for (int g0 = 0; g0 < n_genes; g0 += CNT_TILE) {
    int gt = min(CNT_TILE, n_genes - g0);
    for (int i = threadIdx.x; i < gt; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    for (int r = c0 + wave; r < c1; r += nw) {
      int cell = cell_order[r];
      int64_t s = indptr[cell], e = indptr[cell + 1];
      for (int64_t i = s + lane; i < e; i += 64) {
        int g = indices[i];
        float x = data[i];
        if (g0 == 0) {
          // counts must be positive integers that fit the 19-bit field
          if (!(x >= 1.0f && x <= (float)MM_MAX_COUNT && x == floorf(x)) || g < 0 || g >= n_genes) bad = 1;
        }
        unsigned gl = (unsigned)(g - g0);
        if (gl < (unsigned)gt) atomicAdd(&cnt[gl], 1u);
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < gt; i += blockDim.x) blk_cnt[(int64_t)b * n_genes + g0 + i] = (uint16_t)cnt[i];
    __syncthreads();
  }
  if (bad) atomicOr(status, 1);
}

// ------------------------------------------------------------------------------------------------
// K0 step 2: per block, rank genes by descending nnz (counting sort on the length), derive the slice
// widths (in dwordx4 rows), slice pointers (rows) and work-item pointers.
#define LAYOUT_MAXLEN (MM_BLOCK_CELLS + 1)
__global__ __launch_bounds__(1024) void k_sell_layout(const uint16_t *__restrict__ blk_cnt, int32_t n_genes, int32_t n_slices,
                                                      int32_t *__restrict__ rank, int32_t *__restrict__ perm,
                                                      int32_t *__restrict__ slice_w, int32_t *__restrict__ slice_ptr,
                                                      int32_t *__restrict__ item_ptr, int64_t *__restrict__ blk_rows,
                                                      int32_t *__restrict__ blk_items) {
  extern __shared__ uint32_t smem[];
  uint32_t *start = smem;                                  // [LAYOUT_MAXLEN] (+ pad)
  uint32_t *scan_tmp = smem + LAYOUT_MAXLEN + 3;           // [1024]
  uint16_t *len_by_rank = (uint16_t *)(scan_tmp + 1024);   // [n_slices*64]
  int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const uint16_t *cnt = blk_cnt + (int64_t)b * n_genes;
  for (int i = tid; i < LAYOUT_MAXLEN; i += nt) start[i] = 0;
  for (int i = tid; i < n_slices * 64; i += nt) len_by_rank[i] = 0;
  __syncthreads();
  for (int g = tid; g < n_genes; g += nt) atomicAdd(&start[cnt[g]], 1u);
  __syncthreads();
  // start[L] <- number of genes with length > L  (descending exclusive scan over L)
  {
    const int per = (LAYOUT_MAXLEN + 1023) / 1024;  // 9
    int hi = LAYOUT_MAXLEN - 1 - tid * per;         // this thread owns L = hi, hi-1, ..., hi-per+1
    uint32_t local = 0;
    for (int k = 0; k < per; k++) {
      int L = hi - k;
      if (L >= 0) local += start[L];
    }
    scan_tmp[tid] = local;
    __syncthreads();
    // inclusive scan of scan_tmp (Hillis-Steele, 1024 threads)
    for (int off = 1; off < 1024; off <<= 1) {
      uint32_t v = tid >= off ? scan_tmp[tid - off] : 0;
      __syncthreads();
      scan_tmp[tid] += v;
      __syncthreads();
    }
    uint32_t run = scan_tmp[tid] - local;  // exclusive prefix = genes with L greater than this thread's range
    for (int k = 0; k < per; k++) {
      int L = hi - k;
      if (L >= 0) {
        uint32_t h = start[L];
        start[L] = run;
        run += h;
      }
    }
  }
  __syncthreads();
  for (int g = tid; g < n_genes; g += nt) {
    uint32_t L = cnt[g];
    uint32_t s = atomicAdd(&start[L], 1u);  // ties broken by arrival; slot order is irrelevant downstream
    rank[(int64_t)b * n_genes + g] = (int32_t)s;
    perm[(int64_t)b * n_slices * 64 + s] = g;
    len_by_rank[s] = (uint16_t)L;
  }
  for (int s = n_genes + tid; s < n_slices * 64; s += nt) perm[(int64_t)b * n_slices * 64 + s] = -1;
  __syncthreads();
  // slice widths in dwordx4 rows, pointers by exclusive scan (n_slices <= 1024 * k handled by a serial carry)
  uint32_t carry_rows = 0, carry_items = 0;
  for (int t0 = 0; t0 < n_slices; t0 += 1024) {
    int t = t0 + tid;
    uint32_t w4 = 0, items = 0;
    if (t < n_slices) {
      w4 = ((uint32_t)len_by_rank[t * 64] + MM_JVEC - 1) / MM_JVEC;
      items = (w4 + MM_ITEM_ROWS - 1) / MM_ITEM_ROWS;
      slice_w[(int64_t)b * n_slices + t] = (int32_t)w4;
    }
    // two scans sharing scan_tmp: rows then items
    for (int pass = 0; pass < 2; pass++) {
      uint32_t mine = pass == 0 ? w4 : items;
      __syncthreads();
      scan_tmp[tid] = mine;
      __syncthreads();
      for (int off = 1; off < 1024; off <<= 1) {
        uint32_t v = tid >= off ? scan_tmp[tid - off] : 0;
        __syncthreads();
        scan_tmp[tid] += v;
        __syncthreads();
      }
      uint32_t excl = scan_tmp[tid] - mine + (pass == 0 ? carry_rows : carry_items);
      uint32_t total = scan_tmp[1023];
      if (t < n_slices) {
        if (pass == 0) slice_ptr[(int64_t)b * (n_slices + 1) + t] = (int32_t)excl;
        else item_ptr[(int64_t)b * (n_slices + 1) + t] = (int32_t)excl;
      }
      if (pass == 0) carry_rows += total; else carry_items += total;
    }
  }
  if (tid == 0) {
    slice_ptr[(int64_t)b * (n_slices + 1) + n_slices] = (int32_t)carry_rows;
    item_ptr[(int64_t)b * (n_slices + 1) + n_slices] = (int32_t)carry_items;
    blk_rows[b] = carry_rows;
    blk_items[b] = (int32_t)carry_items;
  }
}

// ------------------------------------------------------------------------------------------------
// K0 step 3: scatter.  One wave per cell row (coalesced CSR reads); per-gene cursors live in LDS as
// packed 16-bit halves; each entry lands at  (blk_base + slice_ptr[t] + j/4)*256 + lane*4 + j%4.
__global__ __launch_bounds__(1024) void k_sell_scatter(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                       const float *__restrict__ data, const int32_t *__restrict__ cell_order,
                                                       const int32_t *__restrict__ blk_cell0, int32_t n_genes, int32_t n_slices,
                                                       const int32_t *__restrict__ rank, const int32_t *__restrict__ slice_ptr,
                                                       const int64_t *__restrict__ blk_base, uint32_t *__restrict__ ent) {
  extern __shared__ uint32_t smem[];
  uint32_t *cur = smem;                         // [(n_genes+1)/2] packed u16 cursors
  int32_t *sptr = (int32_t *)(smem + (n_genes + 1) / 2);  // [n_slices]
  int b = blockIdx.x;
  int c0 = blk_cell0[b], c1 = blk_cell0[b + 1];
  int lane = mm_lane(), wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int i = threadIdx.x; i < (n_genes + 1) / 2; i += blockDim.x) cur[i] = 0;
  for (int i = threadIdx.x; i < n_slices; i += blockDim.x) sptr[i] = slice_ptr[(int64_t)b * (n_slices + 1) + i];
  __syncthreads();
  const int32_t *rk = rank + (int64_t)b * n_genes;
  int64_t base = blk_base[b];
  for (int r = c0 + wave; r < c1; r += nw) {
    int cell = cell_order[r];
    uint32_t cell_local = (uint32_t)(r - c0);
    int64_t s = indptr[cell], e = indptr[cell + 1];
    for (int64_t i = s + lane; i < e; i += 64) {
      int g = indices[i];
      uint32_t x = (uint32_t)data[i];
      int sl = rk[g];
      unsigned sh = (g & 1) * 16;
      uint32_t old = atomicAdd(&cur[g >> 1], 1u << sh);
      uint32_t j = (old >> sh) & 0xFFFFu;
      int64_t row = base + sptr[sl >> 6] + (j >> 2);
      ent[row * 256 + (sl & 63) * 4 + (j & 3)] = cell_local | (x << MM_CELL_BITS);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K0, range-partitioned form (rows with ascending column indices -- the canonical CSR every scipy / device producer here
// emits).  The scattered 4-byte stores of k_sell_scatter above re-open every 64-byte line of the block's 20 MB entry region up
// to 16 times, and with ~17 blocks in flight per XCD those lines do not survive in the 4 MB L2: measured at BASELINE configs[2]
// (profiles/r02_k1_traffic_C3.json) 18.7 GB written + 17 GB fetched for 2.4 GB of entries.  Here a workgroup owns (block, a
// contiguous range of gene ids): in a sorted row its entries are ONE contiguous segment, and its per-gene state is a few KB of LDS
// -- small enough to park every gene's open 16-byte group there (k_sell_scatter_quads) instead of storing entry by entry.
//
// Step 0: per row, where each gene range starts (R + 1 absolute positions in indices / data), and the structural checks
// (column indices inside [0, G), strictly ascending).  One wave per row, coalesced index reads, R - 1 ballots per 64 entries.
__global__ __launch_bounds__(256) void k_sell_split(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                    const int32_t *__restrict__ cell_order, int64_t n_sel, int32_t n_genes,
                                                    int32_t n_ranges, int64_t *__restrict__ rowsplit, int32_t *__restrict__ status) {
  __shared__ int32_t bnd[33];                       // bnd[k] = first gene id of range k
  if (threadIdx.x <= n_ranges) bnd[threadIdx.x] = (int32_t)(((int64_t)threadIdx.x * n_genes) / n_ranges);
  __syncthreads();
  int lane = mm_lane();
  int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int bad = 0;
  for (int64_t r = wave; r < n_sel; r += nwaves) {
    int cell = cell_order[r];
    int64_t s = indptr[cell], e = indptr[cell + 1];
    // lane k (1 <= k < R) accumulates the number of entries with column < bnd[k]
    uint32_t mine = 0;
    int prev_last = -1;
    for (int64_t i0 = s; i0 < e; i0 += 64) {
      int64_t i = i0 + lane;
      int g = i < e ? indices[i] : 0x7fffffff;
      int gp = __shfl_up(g, 1, 64);
      if (lane == 0) gp = prev_last;
      if (i < e && (g < 0 || g >= n_genes || g <= gp)) bad = 1;
      prev_last = __shfl(g, 63, 64);
      for (int k = 1; k < n_ranges; k++) {
        uint64_t bal = __ballot(g < bnd[k]);        // wave-uniform bound (LDS broadcast), one s_bcnt1 per range
        if (lane == k) mine += (uint32_t)__popcll(bal);
      }
    }
    if (lane == 0) mine = 0;
    if (lane == n_ranges) mine = (uint32_t)(e - s);
    if (lane <= n_ranges) rowsplit[r * (n_ranges + 1) + lane] = s + (int64_t)mine;   // ABSOLUTE position in indices / data
  }
  if (bad) atomicOr(status, 2);
}

// Step 1: nnz per (block, gene).  One workgroup per (block, gene range); LDS counters of the range's genes only.
__global__ __launch_bounds__(1024) void k_sell_count_ranges(const int32_t *__restrict__ indices, const int32_t *__restrict__ blk_cell0,
                                                            int32_t n_blocks, int32_t n_genes, int32_t n_ranges,
                                                            const int64_t *__restrict__ rowsplit, uint16_t *__restrict__ blk_cnt) {
  extern __shared__ uint32_t smem[];
  // workgroup -> (block, range): the R workgroups of one block get ids that are equal mod 8, i.e. land on one XCD (speed only)
  int x = blockIdx.x & 7, t = blockIdx.x >> 3;
  int b = (t / n_ranges) * 8 + x, rg = t % n_ranges;
  if (b >= n_blocks) return;
  int g0 = (int)(((int64_t)rg * n_genes) / n_ranges), g1 = (int)(((int64_t)(rg + 1) * n_genes) / n_ranges);
  int ngr = g1 - g0;
  uint32_t *cur = smem;                                  // [ngr] counts of this range's genes
  for (int i = threadIdx.x; i < ngr; i += blockDim.x) cur[i] = 0;
  __syncthreads();
  int c0 = blk_cell0[b], c1 = blk_cell0[b + 1];
  int lane = mm_lane(), wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int64_t ld = n_ranges + 1;
  // one row segment per wave iteration; the NEXT row's segment bounds are fetched while the current one is processed
  int r = c0 + wave;
  int64_t a = 0, z = 0;
  if (r < c1) {
    a = rowsplit[r * ld + rg];
    z = rowsplit[r * ld + rg + 1];
  }
  for (; r < c1; r += nw) {
    int rn = r + nw;
    int64_t an = 0, zn = 0;
    if (rn < c1) {
      an = rowsplit[rn * ld + rg];
      zn = rowsplit[rn * ld + rg + 1];
    }
    for (int64_t i = a + lane; i < z; i += 64) atomicAdd(&cur[indices[i] - g0], 1u);
    a = an;
    z = zn;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ngr; i += blockDim.x) blk_cnt[(int64_t)b * n_genes + g0 + i] = (uint16_t)cur[i];
}

// Step 3, deterministic and write-combined.  Measured on this chip (profiles/README.md, round 2): a 4-byte store to a line that
// is not being written by the same wave instruction costs ~26 B of HBM write traffic -- stores are written through, L2 does not
// merge them over time -- so the per-entry scatter above writes 15.9 GB for 2.4 GB of entries even with XCD-local workgroups.
// Here an entry is first parked in LDS and only complete 16-byte groups (the 4 consecutive entries of one gene = one lane's
// dwordx4 of a slice row) go to HBM.  To make that race-free WITHOUT atomics a gene belongs to exactly one WAVE: the workgroup owns
// (block, gene range) as before and wave w of it owns the w-th quarter of that range; every wave walks ALL rows of the block in
// order, loads the row's range segment (one coalesced load, shared through L1 by the 4 waves) and keeps the lanes whose gene is
// its own.  Entries of one row have distinct genes, so the lanes of an iteration never collide, and a gene's entries arrive in
// cell order: the entry order inside a gene is now DETERMINISTIC (ascending cell), as is every fp64 sum K1 forms from it.
#define RS_WAVES 4
__global__ __launch_bounds__(64 * RS_WAVES) void k_sell_scatter_quads(const int32_t *__restrict__ indices, const float *__restrict__ data,
                                                                       const int32_t *__restrict__ blk_cell0, int32_t n_blocks,
                                                                       int32_t n_genes, int32_t n_slices, int32_t n_ranges,
                                                                       const int64_t *__restrict__ rowsplit, const int32_t *__restrict__ rank,
                                                                       const int32_t *__restrict__ slice_ptr, const int64_t *__restrict__ blk_base,
                                                                       uint32_t *__restrict__ ent, int32_t *__restrict__ status) {
  extern __shared__ u32x4 smem_q[];                      // 16-byte aligned dynamic LDS (the group buffers are read as dwordx4)
  uint32_t *smem = (uint32_t *)smem_q;
  int x = blockIdx.x & 7, t = blockIdx.x >> 3;
  int b = (t / n_ranges) * 8 + x, rg = t % n_ranges;
  if (b >= n_blocks) return;
  int g0 = (int)(((int64_t)rg * n_genes) / n_ranges), g1 = (int)(((int64_t)(rg + 1) * n_genes) / n_ranges);
  int ngr = g1 - g0;
  uint32_t *stage = smem;                                // [ngr][4] the open 16-byte group of every gene
  uint32_t *cur = smem + (size_t)ngr * 4;                // [ngr] entries seen so far
  int64_t *dst = (int64_t *)(cur + ngr + (ngr & 1));     // [ngr] ent index of the gene's entry 0: (base + sptr[slice])*256 + lane*4
  for (int i = threadIdx.x; i < ngr; i += blockDim.x) {
    cur[i] = 0;
    int sl = rank[(int64_t)b * n_genes + g0 + i];
    dst[i] = (blk_base[b] + slice_ptr[(int64_t)b * (n_slices + 1) + (sl >> 6)]) * 256 + (sl & 63) * 4;
  }
  __syncthreads();
  int c0 = blk_cell0[b], c1 = blk_cell0[b + 1];
  int lane = mm_lane(), wave = threadIdx.x >> 6;
  int lo = (int)(((int64_t)wave * ngr) / RS_WAVES), hi = (int)(((int64_t)(wave + 1) * ngr) / RS_WAVES);   // this wave's genes
  const int64_t ld = n_ranges + 1;
  int bad = 0;
  // A wave walks the rows strictly in order, so its latency per row is what bounds the kernel: the segment bounds and the first
  // 64 column indices of the next RS_AHEAD rows are kept in flight (software pipeline in registers).
#define RS_AHEAD 4
  int64_t ca[RS_AHEAD], cz[RS_AHEAD], na[RS_AHEAD], nz[RS_AHEAD];
  int cg[RS_AHEAD];
  float cd[RS_AHEAD];
  auto load_bounds = [&](int r0, int64_t *pa, int64_t *pz) {
#pragma unroll
    for (int u = 0; u < RS_AHEAD; u++) {
      int r = r0 + u;
      pa[u] = pz[u] = 0;
      if (r < c1) {
        pa[u] = rowsplit[(int64_t)r * ld + rg];
        pz[u] = rowsplit[(int64_t)r * ld + rg + 1];
      }
    }
  };
  load_bounds(c0, ca, cz);
#pragma unroll
  for (int u = 0; u < RS_AHEAD; u++) {
    bool in = ca[u] + lane < cz[u];
    cg[u] = in ? indices[ca[u] + lane] : 0;
    cd[u] = in ? data[ca[u] + lane] : 1.0f;
  }
  load_bounds(c0 + RS_AHEAD, na, nz);
  for (int r0 = c0; r0 < c1; r0 += RS_AHEAD) {
    // issued first, consumed in the NEXT pass: the next group's column indices and the bounds of the group after it
    int ng[RS_AHEAD];
    float nd[RS_AHEAD];
    int64_t fa[RS_AHEAD], fz[RS_AHEAD];
#pragma unroll
    for (int u = 0; u < RS_AHEAD; u++) {
      bool in = na[u] + lane < nz[u];
      ng[u] = in ? indices[na[u] + lane] : 0;
      nd[u] = in ? data[na[u] + lane] : 1.0f;      // all 64 lanes (the 4 waves share the lines through L1): no dependent load later
    }
    load_bounds(r0 + 2 * RS_AHEAD, fa, fz);
#pragma unroll
    for (int u = 0; u < RS_AHEAD; u++) {
      int r = r0 + u;
      if (r < c1) {
        uint32_t cell_local = (uint32_t)(r - c0);
        int64_t a = ca[u], z = cz[u];
        for (int64_t i = a + lane; i < z; i += 64) {
          bool first = i < a + 64;
          int gl = (first ? cg[u] : indices[i]) - g0;
          if (gl >= lo && gl < hi) {
            float xf = first ? cd[u] : data[i];
            if (!(xf >= 1.0f && xf <= (float)MM_MAX_COUNT && xf == floorf(xf))) {  // counts: positive integers inside the 19-bit field
              bad = 1;
              xf = 1.0f;
            }
            uint32_t j = cur[gl];
            cur[gl] = j + 1;
            stage[gl * 4 + (j & 3)] = cell_local | ((uint32_t)xf << MM_CELL_BITS);
            if ((j & 3) == 3) {                            // the group is complete: one 16-byte store
              u32x4 q = *(const u32x4 *)(stage + gl * 4);
              *(u32x4 *)(ent + dst[gl] + (int64_t)(j >> 2) * 256) = q;
            }
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < RS_AHEAD; u++) {
      ca[u] = na[u]; cz[u] = nz[u]; cg[u] = ng[u]; cd[u] = nd[u];
      na[u] = fa[u]; nz[u] = fz[u];
    }
  }
  // leftovers: the last, incomplete group of every gene of this wave (<= 3 entries; the rest of the group stays zero = padding)
  for (int gl = lo + lane; gl < hi; gl += 64) {
    uint32_t n = cur[gl];
    for (uint32_t k = n & ~3u; k < n; k++) ent[dst[gl] + (int64_t)(k >> 2) * 256 + (k & 3)] = stage[gl * 4 + (k & 3)];
  }
  if (bad) atomicOr(status, 1);
}

// ------------------------------------------------------------------------------------------------
extern "C" {

int mm_csr_rowsum(const int64_t *d_indptr, const int32_t *d_indices, const float *d_data, int64_t n_rows,
                  const uint8_t *d_gene_mask, double *d_out, void *stream) {
  MM_ARG(d_indptr && d_indices && d_data && d_out && n_rows >= 0);
  if (n_rows == 0) return MM_OK;
  int64_t blocks = (n_rows + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_csr_rowsum, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_indptr, d_indices, d_data, n_rows,
                     d_gene_mask, d_out);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_csr_colcount(const int64_t *d_indptr, const int32_t *d_indices, int64_t n_rows, int32_t col_lo, int32_t col_hi,
                    int64_t *d_row_nnz, void *stream) {
  MM_ARG(d_indptr && d_indices && d_row_nnz && n_rows >= 0 && col_lo >= 0 && col_hi >= col_lo);
  if (n_rows == 0) return MM_OK;
  int64_t blocks = (n_rows + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_csr_colcount, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_indptr, d_indices, n_rows, col_lo,
                     col_hi, d_row_nnz);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_csr_colsplit(const int64_t *d_indptr, const int32_t *d_indices, const float *d_data, int64_t n_rows, int32_t col_lo,
                    int32_t col_hi, const int64_t *d_out_indptr, int32_t *d_out_indices, float *d_out_data, void *stream) {
  MM_ARG(d_indptr && d_indices && d_data && d_out_indptr && d_out_indices && d_out_data && n_rows >= 0 && col_lo >= 0 && col_hi >= col_lo);
  if (n_rows == 0) return MM_OK;
  int64_t blocks = (n_rows + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_csr_colsplit, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_indptr, d_indices, d_data, n_rows,
                     col_lo, col_hi, d_out_indptr, d_out_indices, d_out_data);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_sell_count(const int64_t *d_indptr, const int32_t *d_indices, const float *d_data, const int32_t *d_cell_order,
                  const int32_t *d_blk_cell0, int32_t n_blocks, int32_t n_genes, uint16_t *d_blk_cnt, int32_t *d_status,
                  void *stream) {
  MM_ARG(d_indptr && d_indices && d_data && d_cell_order && d_blk_cell0 && d_blk_cnt && d_status);
  MM_ARG(n_blocks >= 0 && n_genes > 0);
  if (n_blocks == 0) return MM_OK;
  hipLaunchKernelGGL(k_sell_count, dim3(n_blocks), dim3(1024), 0, (hipStream_t)stream, d_indptr, d_indices, d_data, d_cell_order,
                     d_blk_cell0, n_genes, d_blk_cnt, d_status);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_sell_layout(const uint16_t *d_blk_cnt, int32_t n_blocks, int32_t n_genes, int32_t *d_rank, int32_t *d_perm,
                   int32_t *d_slice_w, int32_t *d_slice_ptr, int32_t *d_item_ptr, int64_t *d_blk_rows, int32_t *d_blk_items,
                   void *stream) {
  MM_ARG(d_blk_cnt && d_rank && d_perm && d_slice_w && d_slice_ptr && d_item_ptr && d_blk_rows && d_blk_items);
  MM_ARG(n_blocks >= 0 && n_genes > 0 && n_genes <= 60000);
  if (n_blocks == 0) return MM_OK;
  int32_t n_slices = (n_genes + 63) / 64;
  size_t shm = (size_t)(LAYOUT_MAXLEN + 3 + 1024) * 4 + (size_t)n_slices * 64 * 2;
  MM_HIP(hipFuncSetAttribute((const void *)k_sell_layout, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  hipLaunchKernelGGL(k_sell_layout, dim3(n_blocks), dim3(1024), shm, (hipStream_t)stream, d_blk_cnt, n_genes, n_slices, d_rank,
                     d_perm, d_slice_w, d_slice_ptr, d_item_ptr, d_blk_rows, d_blk_items);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_sell_scatter(const int64_t *d_indptr, const int32_t *d_indices, const float *d_data, const int32_t *d_cell_order,
                    const int32_t *d_blk_cell0, int32_t n_blocks, int32_t n_genes, const int32_t *d_rank,
                    const int32_t *d_slice_ptr, const int64_t *d_blk_base, uint32_t *d_ent, void *stream) {
  MM_ARG(d_indptr && d_indices && d_data && d_cell_order && d_blk_cell0 && d_rank && d_slice_ptr && d_blk_base && d_ent);
  MM_ARG(n_blocks >= 0 && n_genes > 0 && n_genes <= 60000);
  if (n_blocks == 0) return MM_OK;
  int32_t n_slices = (n_genes + 63) / 64;
  size_t shm = (size_t)((n_genes + 1) / 2 + n_slices) * 4;
  MM_HIP(hipFuncSetAttribute((const void *)k_sell_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  hipLaunchKernelGGL(k_sell_scatter, dim3(n_blocks), dim3(1024), shm, (hipStream_t)stream, d_indptr, d_indices, d_data, d_cell_order,
                     d_blk_cell0, n_genes, n_slices, d_rank, d_slice_ptr, d_blk_base, d_ent);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_sell_split(const int64_t *d_indptr, const int32_t *d_indices, const int32_t *d_cell_order, int64_t n_sel, int32_t n_genes,
                  int32_t n_ranges, int64_t *d_rowsplit, int32_t *d_status, void *stream) {
  MM_ARG(d_indptr && d_indices && d_cell_order && d_rowsplit && d_status && n_sel >= 0 && n_genes > 0);
  MM_ARG(n_ranges >= 1 && n_ranges <= 32 && n_ranges <= n_genes);
  if (n_sel == 0) return MM_OK;
  int64_t blocks = (n_sel + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(k_sell_split, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_indptr, d_indices, d_cell_order, n_sel,
                     n_genes, n_ranges, d_rowsplit, d_status);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_sell_count_ranges(const int64_t *d_indptr, const int32_t *d_indices, const int32_t *d_cell_order, const int32_t *d_blk_cell0,
                         int32_t n_blocks, int32_t n_genes, int32_t n_ranges, const int64_t *d_rowsplit, uint16_t *d_blk_cnt,
                         void *stream) {
  (void)d_indptr;
  (void)d_cell_order;   // the row positions come from d_rowsplit; kept in the signature next to mm_sell_count's
  MM_ARG(d_indices && d_blk_cell0 && d_rowsplit && d_blk_cnt);
  MM_ARG(n_blocks >= 0 && n_genes > 0 && n_genes <= 65536 && n_ranges >= 1 && n_ranges <= 32 && n_ranges <= n_genes);
  if (n_blocks == 0) return MM_OK;
  size_t shm = (size_t)(n_genes / n_ranges + 1) * 4;
  MM_ARG(shm <= 150 * 1024);
  int64_t grid = (int64_t)((n_blocks + 7) / 8) * n_ranges * 8;
  MM_ARG(grid < 2147483647LL);
  MM_HIP(hipFuncSetAttribute((const void *)k_sell_count_ranges, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  hipLaunchKernelGGL(k_sell_count_ranges, dim3((unsigned)grid), dim3(1024), shm, (hipStream_t)stream, d_indices, d_blk_cell0, n_blocks,
                     n_genes, n_ranges, d_rowsplit, d_blk_cnt);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_sell_scatter_ranges(const int64_t *d_indptr, const int32_t *d_indices, const float *d_data, const int32_t *d_cell_order,
                           const int32_t *d_blk_cell0, int32_t n_blocks, int32_t n_genes, int32_t n_ranges,
                           const int64_t *d_rowsplit, const int32_t *d_rank, const int32_t *d_slice_ptr, const int64_t *d_blk_base,
                           uint32_t *d_ent, int32_t *d_status, void *stream) {
  (void)d_indptr;
  (void)d_cell_order;
  MM_ARG(d_indices && d_data && d_blk_cell0 && d_rowsplit && d_rank && d_slice_ptr && d_blk_base && d_ent);
  MM_ARG(d_status && n_blocks >= 0 && n_genes > 0 && n_genes <= 65536 && n_ranges >= 1 && n_ranges <= 32 && n_ranges <= n_genes);
  if (n_blocks == 0) return MM_OK;
  int32_t n_slices = (n_genes + 63) / 64;
  int32_t ngr_max = n_genes / n_ranges + 1;
  size_t shm = (size_t)ngr_max * (16 + 4 + 8) + 8;       // group buffer + cursor + destination per gene of the range
  MM_ARG(shm <= 150 * 1024);
  int64_t grid = (int64_t)((n_blocks + 7) / 8) * n_ranges * 8;
  MM_ARG(grid < 2147483647LL);
  MM_HIP(hipFuncSetAttribute((const void *)k_sell_scatter_quads, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  hipLaunchKernelGGL(k_sell_scatter_quads, dim3((unsigned)grid), dim3(64 * RS_WAVES), shm, (hipStream_t)stream, d_indices, d_data,
                     d_blk_cell0, n_blocks, n_genes, n_slices, n_ranges, d_rowsplit, d_rank, d_slice_ptr, d_blk_base, d_ent, d_status);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

}  // extern "C"
