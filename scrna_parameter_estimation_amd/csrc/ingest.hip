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

// Cost-balanced gene shards are NOT contiguous ranges: the same two passes with a column map (new id of a kept column, -1 =
// dropped; ascending in the old id, so rows stay sorted), and the per-gene totals the balancing is computed from.
__global__ __launch_bounds__(256) void k_csr_mapcount(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                      int64_t n_rows, const int32_t *__restrict__ col_map, int64_t *__restrict__ row_nnz) {
  int lane = mm_lane();
  int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < n_rows; r += nwaves) {
    int64_t s = indptr[r], e = indptr[r + 1];
    int cnt = 0;
    for (int64_t i = s + lane; i < e; i += 64) cnt += col_map[indices[i]] >= 0 ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (lane == 0) row_nnz[r] = cnt;
  }
}

__global__ __launch_bounds__(256) void k_csr_mapsplit(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                      const float *__restrict__ data, int64_t n_rows, const int32_t *__restrict__ col_map,
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
      int g = i < e ? col_map[indices[i]] : -1;
      bool keep = g >= 0;
      uint64_t bal = __ballot(keep);
      if (keep) {
        int64_t pos = o + __popcll(bal & ((1ull << lane) - 1ull));
        out_indices[pos] = g;
        out_data[pos] = data[i];
      }
      o += __popcll(bal);
    }
  }
}

// per-gene totals of the CSR (fp64 atomics: counts are integers < 2^53, so the sums are exact whatever the order)
__global__ __launch_bounds__(256) void k_csr_colsum(const int32_t *__restrict__ indices, const float *__restrict__ data, int64_t nnz,
                                                    double *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nnz; i += stride) atomicAdd(out + indices[i], (double)data[i]);
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
  // Placement in GENE ORDER, so that genes of equal length take their slots by ascending gene id: the whole layout (hence every
  // count block, bit for bit) is a pure function of the counts.  Step 1, all waves: for every gene the number of genes with the
  // same length among the LOWER lanes of its 64-gene chunk (64 readlanes; parked in the rank array).  Step 2, one wave walks the
  // chunks in order: slot = running counter of the length + that number; the counters move on only after every lane of the
  // chunk has read them (LDS operations of one wave complete in order).  Four chunks' operands are loaded ahead.
  {
    const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    for (int g0 = wave * 64; g0 < n_genes; g0 += nw * 64) {
      int g = g0 + lane;
      uint32_t L = g < n_genes ? (uint32_t)cnt[g] : 0xFFFFFFFFu, r = 0;
      for (int i = 0; i < 64; i++) {
        uint32_t Li = (uint32_t)__builtin_amdgcn_readlane((int)L, i);
        r += (Li == L && i < lane) ? 1u : 0u;
      }
      if (g < n_genes) rank[(int64_t)b * n_genes + g] = (int32_t)r;
    }
    __syncthreads();
    if (wave == 0) {
      for (int g0 = 0; g0 < n_genes; g0 += 256) {
        uint32_t Lq[4], rq[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          int g = g0 + 64 * q + lane;
          Lq[q] = g < n_genes ? (uint32_t)cnt[g] : 0u;
          rq[q] = g < n_genes ? (uint32_t)rank[(int64_t)b * n_genes + g] : 0u;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          int g = g0 + 64 * q + lane;
          uint32_t L = Lq[q];
          if (g < n_genes) {
            uint32_t sl = ((volatile uint32_t *)start)[L] + rq[q];
            rank[(int64_t)b * n_genes + g] = (int32_t)sl;
            perm[(int64_t)b * n_slices * 64 + sl] = g;
            len_by_rank[sl] = (uint16_t)L;
          }
          __builtin_amdgcn_wave_barrier();
          if (g < n_genes) atomicAdd(&start[L], 1u);
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
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
// range of MM_RANGE_GENES consecutive gene ids): in a sorted row its entries are ONE contiguous segment, and its per-gene state
// is 48 B x 1024 of LDS -- small enough to assemble every gene's 16-byte groups there instead of storing entry by entry.
//
// Steps 0 + 1 in one pass over the column indices: per row, where each gene range starts (R + 1 absolute positions in indices /
// data) with the structural checks (column indices inside [0, G), strictly ascending), and nnz per (block, gene).
// A workgroup takes SC_ROWS consecutive rows of ONE block, 64 at a time (4 per wave): a lane whose range id differs from its
// predecessor's owns the boundaries in between (no ballots, no search); the 64 rows' table leaves through LDS transposed --
// rowsplit[range][row], so that the scatter's (block, range) workgroup reads its bounds as contiguous runs.  The counts are LDS
// atomics on 16-bit halves (a block has at most 8192 cells: no carry) merged into blk_cnt by global atomics at the end (integer
// sums: the result does not depend on the order).
#ifndef SC_ROWS
#define SC_ROWS 1024
#endif
#define SC_THREADS 1024
#ifndef SC_AHEAD
#define SC_AHEAD 4
#endif
__global__ __launch_bounds__(SC_THREADS) void k_sell_split_count(const int64_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                                                 const int32_t *__restrict__ cell_order, const int32_t *__restrict__ blk_cell0,
                                                                 int32_t n_blocks, int64_t n_sel, int32_t n_genes, int32_t n_ranges,
                                                                 int64_t *__restrict__ rowsplit, uint32_t *__restrict__ blk_cnt_words,
                                                                 int32_t *__restrict__ status) {
  extern __shared__ uint32_t cnt[];                          // [(n_genes + 1) / 2] two 16-bit counters per word
  __shared__ volatile int32_t pos[MM_MAX_RANGES + 1][64];    // offsets inside the row
  __shared__ int64_t rstart[64];
  constexpr int CH = MM_BLOCK_CELLS / SC_ROWS, NW = SC_THREADS / 64;
  int b = blockIdx.x / CH, ch = blockIdx.x % CH;
  int c0 = blk_cell0[b], c1 = blk_cell0[b + 1];
  int ra = c0 + ch * SC_ROWS, rz = min(c1, ra + SC_ROWS);
  if (ra >= rz) return;
  int lane = mm_lane(), wv = threadIdx.x >> 6;
  const int nw = (n_genes + 1) / 2;
  for (int i = threadIdx.x; i < nw; i += SC_THREADS) cnt[i] = 0;
  __syncthreads();
  int bad = 0;
  for (int rb = ra; rb < rz; rb += 64) {
    for (int u = 0; u < 64 / NW; u++) {
      int rl = u * NW + wv;
      int r = rb + rl;
      if (r >= rz) break;
      int cell = cell_order[r];
      int64_t s = indptr[cell], e = indptr[cell + 1];
      if (lane == 0) rstart[rl] = s;
      int prev_g = -1, prev_key = -1;
      for (int64_t i0 = s; i0 <= e; i0 += 64 * SC_AHEAD) {       // position e takes part as the sentinel (range id R)
        int gq[SC_AHEAD];
#pragma unroll
        for (int q = 0; q < SC_AHEAD; q++) {                       // SC_AHEAD chunks of 64 column indices in flight
          int64_t i = i0 + 64 * q + lane;
          gq[q] = i < e ? indices[i] : 0x7fffffff;
        }
#pragma unroll
        for (int q = 0; q < SC_AHEAD; q++) {
          int64_t i = i0 + 64 * q + lane;
          if (i0 + 64 * q <= e) {                                  // wave-uniform
            int g = gq[q];
            int key = n_ranges;
            if (i < e) {
              if (g < 0 || g >= n_genes) {
                bad = 1;
                g = g < 0 ? 0 : n_genes - 1;
              }
              key = g >> MM_RANGE_SHIFT;
              atomicAdd(&cnt[g >> 1], 1u << (16 * (g & 1)));
            }
            // the predecessor's column and range id: wave_shr:1 in the DPP network, lane 0 takes the previous chunk's last
            int gp = __builtin_amdgcn_update_dpp(prev_g, g, 0x138, 0xf, 0xf, false);
            int kp = __builtin_amdgcn_update_dpp(prev_key, key, 0x138, 0xf, 0xf, false);
            if (i < e && g <= gp) bad = 1;          // not strictly ascending
            if (i <= e)
              for (int k = kp + 1; k <= key; k++) pos[k][rl] = (int32_t)(i - s);   // (an unsorted row may leave holes: flagged, not used)
            prev_g = __builtin_amdgcn_readlane(g, 63);
            prev_key = __builtin_amdgcn_readlane(key, 63);
          }
        }
      }
    }
    __syncthreads();
    int r = rb + lane;
    if (r < rz)
      for (int k = wv; k <= n_ranges; k += NW) rowsplit[(int64_t)k * n_sel + r] = rstart[lane] + pos[k][lane];
    __syncthreads();
  }
  // merge: element (b, g) of the uint16 table [n_blocks][n_genes] is half (e & 1) of word e >> 1, e = b * G + g
  int64_t e0 = (int64_t)b * n_genes;
  for (int i = threadIdx.x; i < nw; i += SC_THREADS) {
    uint32_t w = cnt[i];
    uint32_t lo = w & 0xFFFFu, hi = w >> 16;
    int64_t ea = e0 + 2 * i;
    if (lo) atomicAdd(&blk_cnt_words[ea >> 1], lo << (16 * (ea & 1)));
    if (hi) atomicAdd(&blk_cnt_words[(ea + 1) >> 1], hi << (16 * ((ea + 1) & 1)));
  }
  if (bad) atomicOr(status, 2);
}

// Step 3, deterministic and write-combined.  Measured on this chip (profiles/README.md, round 2): a 4-byte store to a line that
// is not being written by the same wave instruction costs ~26 B of HBM write traffic -- stores are written through, L2 does not
// merge them over time -- so the per-entry scatter above writes 15.9 GB for 2.4 GB of entries even with XCD-local workgroups.
// Here the workgroup of (block, gene range) takes the block's rows in TILES of up to 128 rows, and only complete 16-byte groups
// (4 consecutive entries of one gene = one lane's dwordx4 of a slice row) go to HBM:
//   P1  every entry of the tile (each handled by exactly one lane) sets bit `row` in its gene's 128-bit row mask (LDS atomicOr:
//       the result does not depend on the order of arrival);
//   P2  per gene: popcounts of the four mask words -> entries n of this tile, and an exclusive scan of n over the range's genes
//       gives the gene's run in the tile buffer;
//   P3  every entry again: its rank inside the gene = bits below `row` in the mask, so the entry's position in the gene is
//       cur[gene] + rank -- a pure function of the data, ascending in the cell index -- and it goes to the tile buffer; the entry
//       that lands on slot 3 of a group files the group for storing;
//   P4  filed groups are read back (entries of earlier tiles from the gene's 4-slot carry) and stored, one dwordx4 each;
//   P5  per gene: the open group moves to the carry, cur += n, mask cleared.
// No entry is ever placed by arrival order: two ingests of one CSR give every gene the same entries in the same order, hence
// bit-identical fp64 sums; k_sell_layout breaks length ties by gene id, so the count blocks as a whole are bit-identical too.
#ifdef INGEST_STAMPS
__device__ unsigned long long g_ing[16];
#define ING_ST(k)                                          \
  do {                                                     \
    unsigned long long n_ = __builtin_readcyclecounter();  \
    st_acc[k] += n_ - st_last;                             \
    st_last = n_;                                          \
  } while (0)
#else
#define ING_ST(k)
#endif
#define MR_T 128          // rows per tile (= bits of the row mask)
#define MR_EMAX 4096      // entries per tile (tile buffer)
#ifndef MR_THREADS
#define MR_THREADS 512
#endif
#define MR_GPT (MM_RANGE_GENES / MR_THREADS)   // genes per thread in the per-gene phases
#define MR_TPR (MR_THREADS / MR_T)            // threads per row when the row of every entry is written out
// groups a tile can complete: per gene ceil(n / 4) <= n / 4 + 3 / 4, summed over the range's genes
#define MR_FILED (MR_EMAX / 4 + MM_RANGE_GENES)
#define MR_WAVES (MR_THREADS / 64)
#define MR_WROWS (MR_T / MR_WAVES)   // 16 rows of a tile per wave

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains the wave's outstanding GLOBAL loads (vmcnt(0)),
// which would end the software prefetch of the next tile at the first barrier.
__device__ __forceinline__ void mm_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Inclusive scan over the 64 lanes in the VALU's DPP network (no LDS crossbar): row_shr 1, 2, 4, 8 inside every row of 16,
// then row_bcast:15 / row_bcast:31 carry the row totals over.  All 64 lanes must be active.
__device__ __forceinline__ int mm_wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // lane 15 of rows 0 / 2 -> rows 1 / 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // lane 31 -> rows 2 and 3
  return v;
}

__global__ __launch_bounds__(MR_THREADS, MR_THREADS / 128) void k_sell_scatter_tiles(const int32_t *__restrict__ indices, const float *__restrict__ data,
                                                                       const int32_t *__restrict__ blk_cell0, int32_t n_blocks,
                                                                       int32_t n_genes, int32_t n_slices, int32_t n_ranges, int64_t n_sel,
                                                                       const int64_t *__restrict__ rowsplit, const int32_t *__restrict__ rank,
                                                                       const int32_t *__restrict__ slice_ptr, const int64_t *__restrict__ blk_base,
                                                                       uint32_t *__restrict__ ent, int32_t *__restrict__ status) {
  extern __shared__ u32x4 smem_q[];
  u32x4 *mask = smem_q;                                        // [1024] row mask of the tile per gene
  u32x4 *stage = mask + MM_RANGE_GENES;                        // [1024] carry: the gene's open group (slot = position & 3)
  uint2 *pc = (uint2 *)(stage + MM_RANGE_GENES);               // [1024] .x popcount prefixes of the mask words | n << 24
                                                               //        .y cur (entries placed before this tile) | run start << 16
  uint32_t *dst = (uint32_t *)(pc + MM_RANGE_GENES);           // [1024] (slice row of entry 0, relative to the block) << 6 | lane slot
  int64_t *rowA = (int64_t *)(dst + MM_RANGE_GENES);           // [128] position of the row's segment in indices / data
  int32_t *rowS = (int32_t *)(rowA + MR_T);                    // [128] entries of the tile before the row
  uint32_t *tilebuf = (uint32_t *)(rowS + MR_T);               // [MR_EMAX]
  uint32_t *filed = tilebuf + MR_EMAX;                         // [MR_FILED] gene | position << 10 of every completed group
  uint8_t *rowOf = (uint8_t *)filed;                           //   (before P3: [MR_EMAX] row of every entry of the tile)
  uint32_t *wtot = filed + MR_FILED;                        // [MR_WAVES] + [1] number of filed groups
  uint32_t *nfiled = wtot + MR_WAVES;
  uint32_t *maskw = (uint32_t *)mask;
  uint32_t *stagew = (uint32_t *)stage;

  int x = blockIdx.x & 7, t = blockIdx.x >> 3;
  int b = (t / n_ranges) * 8 + x, rg = t % n_ranges;
  if (b >= n_blocks) return;
  const int tid = threadIdx.x, lane = mm_lane(), wave = tid >> 6;
  const int g0 = rg << MM_RANGE_SHIFT, ngr = min(MM_RANGE_GENES, n_genes - g0);
  for (int i = tid; i < MM_RANGE_GENES; i += MR_THREADS) {
    mask[i] = u32x4{0, 0, 0, 0};
    stage[i] = u32x4{0, 0, 0, 0};
    pc[i] = uint2{0, 0};
    uint32_t d = 0;
    if (i < ngr) {
      int sl = rank[(int64_t)b * n_genes + g0 + i];
      d = ((uint32_t)slice_ptr[(int64_t)b * (n_slices + 1) + (sl >> 6)] << 6) | (uint32_t)(sl & 63);
    }
    dst[i] = d;
  }
  if (tid == 0) *nfiled = 0;
  __syncthreads();
  const int c0 = blk_cell0[b], c1 = blk_cell0[b + 1];
  const int64_t base_row = blk_base[b];
  const int64_t *rsA = rowsplit + (int64_t)rg * n_sel, *rsZ = rsA + n_sel;    // [range][row]: contiguous per workgroup
  constexpr int KMAX = MR_EMAX / MR_THREADS;
  int bad = 0;
#ifdef INGEST_STAMPS
  unsigned long long st_acc[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_readcyclecounter(), st_tiles = 0;
#endif

  // Software pipeline over the tiles: while tile t is ranked and stored, the entries of tile t + 1 are already on their way
  // (registers gr2 / en2) and the segment bounds of tile t + 2 as well.
  // Segment bounds of a tile's rows: lane L holds rows r + L and r + 64 + L.
  int64_t a0 = 0, a1 = 0;
  uint32_t z0 = 0, z1 = 0;                        // low halves of the segment ends (only end - start is used)
  // (rows past the block's end read its last row -- valid memory -- and count as empty in plan_tile: a conditional load would
  // put a use of the loaded value, hence a wait, right behind the load)
  auto load_bounds = [&](int r) {
    int ra = min(r + lane, c1 - 1), rb = min(r + 64 + lane, c1 - 1);
    a0 = rsA[ra];
    z0 = ((const uint32_t *)rsZ)[2 * ra];        // (a full 8-byte load would leave a dead high register that the allocator
    a1 = rsA[rb];                                //  reuses at once -- a write-after-write wait on the load in flight)
    z1 = ((const uint32_t *)rsZ)[2 * rb];
  };
  // from the bounds in (a0 .. z1) of the tile starting at row r: extent T and entries E (every wave derives the same values, no
  // exchange), the row tables and the row of every entry in LDS (4 threads per row)
  auto plan_tile = [&](int r, int &T, int &E) {
    int l0 = r + lane < c1 ? (int)(z0 - (uint32_t)a0) : 0, l1 = r + 64 + lane < c1 ? (int)(z1 - (uint32_t)a1) : 0;
    int inc0 = mm_wave_incl_scan(l0);
    int tot0 = __builtin_amdgcn_readlane(inc0, 63);
    int inc1 = tot0 + mm_wave_incl_scan(l1);
    T = __popcll(__ballot(inc0 <= MR_EMAX)) + __popcll(__ballot(inc1 <= MR_EMAX));   // cumulative counts are monotone
    T = min(T, c1 - r);                          // (>= 1 while r < c1: one segment holds at most 1024 entries)
    E = 0;
    if (T > 0) E = T <= 64 ? __builtin_amdgcn_readlane(inc0, (T - 1) & 63) : __builtin_amdgcn_readlane(inc1, (T - 1) & 63);
    if (wave == 0 && lane < T) {
      rowA[lane] = a0;
      rowS[lane] = inc0 - l0;
    }
    if (wave == 1 && 64 + lane < T) {
      rowA[64 + lane] = a1;
      rowS[64 + lane] = inc1 - l1;
    }
    int row = tid / MR_TPR;                      // the rows of one wave lie all in the first or all in the second half
    int st = __shfl(wave < MR_WAVES / 2 ? inc0 - l0 : inc1 - l1, row & 63, 64);
    int ln = __shfl(wave < MR_WAVES / 2 ? l0 : l1, row & 63, 64);
    if (row >= T) ln = 0;
    for (int k = tid % MR_TPR; k < ln; k += MR_TPR) rowOf[st + k] = (uint8_t)row;
  };
  // thread -> entries tid, tid + 512, ... of the tile (coalesced along the rows).  In flight: raw column index, raw value and the
  // row (a byte each, four per register); nothing is computed from a loaded value before the next pass, so the loads stay in
  // flight behind the ranking and storing of the current tile.  After P1: gr = gene (10 bits) | row << 10, en = packed entry.
  uint32_t gr[KMAX], en[KMAX], gi2[KMAX], en2[KMAX], rw[KMAX / 4], rw2[KMAX / 4];
  auto issue_loads = [&](int E, uint32_t *pg, uint32_t *pe, uint32_t *pr) {
#pragma unroll
    for (int k = 0; k < KMAX / 4; k++) pr[k] = 0;
    const int elast = max(E - 1, 0);
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
      int e = tid + k * MR_THREADS;
      int row = rowOf[min(e, elast)] & (MR_T - 1);          // branch-free: LDS reads stay inside their arrays
      int64_t pos = rowA[row] + (e - rowS[row]);
      if (e >= E) {                              // entries past the tile's end read position 0: valid memory, never used
        pos = 0;
        row = 0;
      }
      pr[k >> 2] |= (uint32_t)row << (8 * (k & 3));
      pg[k] = (uint32_t)indices[pos];
      pe[k] = __float_as_uint(data[pos]);
    }
  };
  int r0 = c0, T = 0, E = 0;
  load_bounds(r0);
  plan_tile(r0, T, E);
  __syncthreads();
  issue_loads(E, gr, en, rw);
  load_bounds(r0 + T);                           // bounds of tile 1
  __syncthreads();                               // (the row tables are rewritten for tile 1 right away)
  while (r0 < c1) {
    ING_ST(9);
    const uint32_t cell_base = (uint32_t)(r0 - c0);
    // tile t + 1: extent and row tables (its bounds arrived during tile t - 1)
    int rn = r0 + T, Tn = 0, En = 0;
    plan_tile(rn, Tn, En);
    ING_ST(0);
    // ---- P1: row masks of tile t
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
      uint32_t raw = gr[k];
      gr[k] = ~0u;
      if (tid + k * MR_THREADS < E) {
        float v = __uint_as_float(en[k]);
        if (!(v >= 1.0f && v <= (float)MM_MAX_COUNT && v == floorf(v))) {   // counts: positive integers inside the 19-bit field
          bad = 1;
          v = 1.0f;
        }
        uint32_t gl = raw - (uint32_t)g0, row = (rw[k >> 2] >> (8 * (k & 3))) & 0xFFu;
        gr[k] = gl | (row << MM_RANGE_SHIFT);
        en[k] = (cell_base + row) | ((uint32_t)v << MM_CELL_BITS);
        atomicOr(&maskw[gl * 4 + (row >> 5)], 1u << (row & 31));
      }
    }
    ING_ST(1);
    mm_lds_barrier();
    ING_ST(2);
    issue_loads(En, gi2, en2, rw2);              // tile t + 1: in flight until the next pass,
    load_bounds(rn + Tn);                        // and the bounds of tile t + 2 behind them (the counter of outstanding loads
                                                 // retires in order: the youngest loads must be the ones needed last)
    ING_ST(10);

    // ---- P2: per gene: prefixes of the mask popcounts, run starts by a workgroup-wide exclusive scan
    const int ga = tid * MR_GPT;
    uint32_t pp[MR_GPT], nn[MR_GPT], nsum = 0;
#pragma unroll
    for (int u = 0; u < MR_GPT; u++) {
      u32x4 m = mask[ga + u];
      uint32_t pa = __popc(m.x), pb = pa + __popc(m.y), pcn = pb + __popc(m.z);
      nn[u] = pcn + __popc(m.w);
      pp[u] = pa | (pb << 8) | (pcn << 16) | (nn[u] << 24);
      nsum += nn[u];
    }
    int incl = mm_wave_incl_scan((int)nsum);
    if (lane == 63) wtot[wave] = (uint32_t)incl;
    ING_ST(11);
    mm_lds_barrier();
    ING_ST(12);
    uint32_t off = 0;
#pragma unroll
    for (int w = 0; w < MR_WAVES; w++) off += w < wave ? wtot[w] : 0;
    off += (uint32_t)incl - nsum;
#pragma unroll
    for (int u = 0; u < MR_GPT; u++) {
      pc[ga + u] = uint2{pp[u], (pc[ga + u].y & 0xFFFFu) | (off << 16)};
      off += nn[u];
    }
    mm_lds_barrier();
    ING_ST(3);

    // ---- P3: place every entry of the tile in its gene's run; slot-3 entries file their group (one LDS atomic per wave and tile)
    uint32_t fl[KMAX];
    uint32_t nmine = 0;
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
      fl[k] = ~0u;
      if (gr[k] != ~0u) {
        uint32_t gl = gr[k] & (MM_RANGE_GENES - 1), row = gr[k] >> MM_RANGE_SHIFT, w = row >> 5;
        uint32_t m = maskw[gl * 4 + w];
        uint2 q = pc[gl];
        uint32_t below = w == 0 ? 0u : (q.x >> (8 * (w - 1))) & 0xFFu;
        uint32_t rk = below + __popc(m & ((1u << (row & 31)) - 1u));
        tilebuf[(q.y >> 16) + rk] = en[k];
        uint32_t j = (q.y & 0xFFFFu) + rk;
        if ((j & 3u) == 3u) {
          fl[k] = gl | (j << MM_RANGE_SHIFT);
          nmine++;
        }
      }
    }
    {
      int incl_f = mm_wave_incl_scan((int)nmine);
      int tot_f = __builtin_amdgcn_readlane(incl_f, 63);
      uint32_t base = 0;
      if (tot_f) {
        if (lane == 0) base = atomicAdd(nfiled, (uint32_t)tot_f);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base) + (uint32_t)incl_f - nmine;
#pragma unroll
        for (int k = 0; k < KMAX; k++)
          if (fl[k] != ~0u) filed[base++] = fl[k];
      }
    }
    ING_ST(4);
    mm_lds_barrier();
    ING_ST(5);

    // ---- P4: store the completed groups
    uint32_t nf = *nfiled;
    for (uint32_t k = tid; k < nf; k += MR_THREADS) {
      uint32_t f = filed[k];
      uint32_t gl = f & (MM_RANGE_GENES - 1), j = f >> MM_RANGE_SHIFT;          // j = position of the group's last entry
      uint32_t c_o = pc[gl].y, c = c_o & 0xFFFFu, off = c_o >> 16;
      uint32_t e0 = j - 3u;
      u32x4 q;
      q.x = e0 >= c ? tilebuf[off + e0 - c] : stagew[gl * 4 + 0];
      q.y = e0 + 1 >= c ? tilebuf[off + e0 + 1 - c] : stagew[gl * 4 + 1];
      q.z = e0 + 2 >= c ? tilebuf[off + e0 + 2 - c] : stagew[gl * 4 + 2];
      q.w = tilebuf[off + j - c];
      uint32_t d = dst[gl];
      *(u32x4 *)(ent + (base_row + (d >> 6) + (j >> 2)) * 256 + (d & 63u) * 4) = q;
    }
    ING_ST(6);
    mm_lds_barrier();
    ING_ST(7);

    // ---- P5: per gene: open group -> carry, cur += n, mask cleared
#pragma unroll
    for (int u = 0; u < MR_GPT; u++) {
      int gl = ga + u;
      uint2 q = pc[gl];
      uint32_t n = q.x >> 24;
      if (n) {
        uint32_t c = q.y & 0xFFFFu, off = q.y >> 16, nc = c + n;
        uint32_t first = max(c, nc & ~3u);
        for (uint32_t e = first; e < nc; e++) stagew[gl * 4 + (e & 3u)] = tilebuf[off + e - c];
        pc[gl].y = nc;
        mask[gl] = u32x4{0, 0, 0, 0};
      }
    }
    if (tid == 0) *nfiled = 0;
    mm_lds_barrier();
    ING_ST(8);
    r0 = rn;
    T = Tn;
    E = En;
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
      gr[k] = gi2[k];
      en[k] = en2[k];
    }
#pragma unroll
    for (int k = 0; k < KMAX / 4; k++) rw[k] = rw2[k];
#ifdef INGEST_STAMPS
    st_tiles++;
#endif
  }
#ifdef INGEST_STAMPS
  if (lane == 0) {
    for (int k = 0; k < 13; k++) atomicAdd(&g_ing[k < 10 ? k : k + 1], st_acc[k]);
    atomicAdd(&g_ing[10], st_tiles);
  }
#endif
  // leftovers: the last, incomplete group of every gene (<= 3 entries; the rest of the group is zero = padding)
  for (int gl = tid; gl < ngr; gl += MR_THREADS) {
    uint32_t n = pc[gl].y & 0xFFFFu, k = n & 3u;
    if (k) {
      u32x4 q = stage[gl];
      if (k < 2) q.y = 0;
      if (k < 3) q.z = 0;
      q.w = 0;
      uint32_t d = dst[gl];
      *(u32x4 *)(ent + (base_row + (d >> 6) + (n >> 2)) * 256 + (d & 63u) * 4) = q;
    }
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

int mm_csr_mapcount(const int64_t *d_indptr, const int32_t *d_indices, int64_t n_rows, const int32_t *d_col_map, int64_t *d_row_nnz,
                    void *stream) {
  MM_ARG(d_indptr && d_indices && d_col_map && d_row_nnz && n_rows >= 0);
  if (n_rows == 0) return MM_OK;
  int64_t blocks = (n_rows + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_csr_mapcount, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_indptr, d_indices, n_rows, d_col_map,
                     d_row_nnz);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_csr_mapsplit(const int64_t *d_indptr, const int32_t *d_indices, const float *d_data, int64_t n_rows, const int32_t *d_col_map,
                    const int64_t *d_out_indptr, int32_t *d_out_indices, float *d_out_data, void *stream) {
  MM_ARG(d_indptr && d_indices && d_data && d_col_map && d_out_indptr && d_out_indices && d_out_data && n_rows >= 0);
  if (n_rows == 0) return MM_OK;
  int64_t blocks = (n_rows + 3) / 4;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(k_csr_mapsplit, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_indptr, d_indices, d_data, n_rows,
                     d_col_map, d_out_indptr, d_out_indices, d_out_data);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_csr_colsum(const int32_t *d_indices, const float *d_data, int64_t nnz, double *d_out /* [n_genes], zeroed by the caller */,
                  void *stream) {
  MM_ARG(d_indices && d_data && d_out && nnz >= 0);
  if (nnz == 0) return MM_OK;
  int64_t blocks = (nnz + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(k_csr_colsum, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_indices, d_data, nnz, d_out);
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

int mm_sell_split_count(const int64_t *d_indptr, const int32_t *d_indices, const int32_t *d_cell_order, const int32_t *d_blk_cell0,
                        int32_t n_blocks, int64_t n_sel, int32_t n_genes, int32_t n_ranges, int64_t *d_rowsplit, uint16_t *d_blk_cnt,
                        int32_t *d_status, void *stream) {
  MM_ARG(d_indptr && d_indices && d_cell_order && d_blk_cell0 && d_rowsplit && d_blk_cnt && d_status);
  MM_ARG(n_blocks >= 0 && n_sel >= 0 && n_genes > 0 && ((uintptr_t)d_blk_cnt & 3) == 0);
  MM_ARG(n_ranges == (n_genes + MM_RANGE_GENES - 1) / MM_RANGE_GENES && n_ranges <= MM_MAX_RANGES);
  if (n_blocks == 0 || n_sel == 0) return MM_OK;
  int64_t grid = (int64_t)n_blocks * (MM_BLOCK_CELLS / SC_ROWS);
  MM_ARG(grid < 2147483647LL);
  size_t shm = (size_t)((n_genes + 1) / 2) * 4;
  MM_HIP(hipFuncSetAttribute((const void *)k_sell_split_count, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  hipLaunchKernelGGL(k_sell_split_count, dim3((unsigned)grid), dim3(SC_THREADS), shm, (hipStream_t)stream, d_indptr, d_indices,
                     d_cell_order, d_blk_cell0, n_blocks, n_sel, n_genes, n_ranges, d_rowsplit, (uint32_t *)d_blk_cnt, d_status);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

int mm_sell_scatter_ranges(const int64_t *d_indptr, const int32_t *d_indices, const float *d_data, const int32_t *d_cell_order,
                           const int32_t *d_blk_cell0, int32_t n_blocks, int32_t n_genes, int32_t n_ranges, int64_t n_sel,
                           const int64_t *d_rowsplit, const int32_t *d_rank, const int32_t *d_slice_ptr, const int64_t *d_blk_base,
                           uint32_t *d_ent, int32_t *d_status, void *stream) {
  (void)d_indptr;
  (void)d_cell_order;
  MM_ARG(d_indices && d_data && d_blk_cell0 && d_rowsplit && d_rank && d_slice_ptr && d_blk_base && d_ent);
  MM_ARG(d_status && n_blocks >= 0 && n_genes > 0 && n_sel >= 0);
  MM_ARG(n_ranges == (n_genes + MM_RANGE_GENES - 1) / MM_RANGE_GENES && n_ranges <= MM_MAX_RANGES);
  if (n_blocks == 0) return MM_OK;
  int32_t n_slices = (n_genes + 63) / 64;
  size_t shm = (size_t)MM_RANGE_GENES * (16 + 16 + 8 + 4) + (size_t)MR_T * 12 + (size_t)MR_EMAX * 4 + (size_t)MR_FILED * 4 + (MR_WAVES + 1) * 4;
  int64_t grid = (int64_t)((n_blocks + 7) / 8) * n_ranges * 8;
  MM_ARG(grid < 2147483647LL);
  MM_HIP(hipFuncSetAttribute((const void *)k_sell_scatter_tiles, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  hipLaunchKernelGGL(k_sell_scatter_tiles, dim3((unsigned)grid), dim3(MR_THREADS), shm, (hipStream_t)stream, d_indices, d_data,
                     d_blk_cell0, n_blocks, n_genes, n_slices, n_ranges, n_sel, d_rowsplit, d_rank, d_slice_ptr, d_blk_base, d_ent,
                     d_status);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

#ifdef INGEST_STAMPS
int mm_debug_ingest_stamps(unsigned long long *host_out16, int reset) {
  MM_HIP(hipDeviceSynchronize());
  if (host_out16) MM_HIP(hipMemcpyFromSymbol(host_out16, HIP_SYMBOL(g_ing), 16 * sizeof(unsigned long long)));
  if (reset) {
    unsigned long long z[16] = {0};
    MM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_ing), z, sizeof(z)));
  }
  return MM_OK;
}
#endif

}  // extern "C"
