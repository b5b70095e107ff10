/* memento_hip.h -- C-ABI of libmemento_hip.so (hand-written HIP for gfx950 / MI355X).
 *
 * The reference (atarashansky/scrna-parameter-estimation, package `memento`) is pure Python; it has no
 * FFI.  These entry points are what a ctypes binding inside memento/main.py would call in place of the
 * numpy/scipy expressions cited on each function (paths relative to /root/reference/).  See
 * INTEGRATION.md for the reference-side stub.
 *
 * Conventions
 *  - every pointer named d_* is a DEVICE pointer (hipMalloc / torch .data_ptr()); h_* is host memory;
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous on it
 *    unless stated otherwise; the library never allocates in a launch function;
 *  - return value: 0 = ok, negative = error (mm_last_error() gives the text, thread-local);
 *  - no torch types anywhere; plain pointers and sizes only.
 *
 * Device data layout ("count blocks"): cells are ordered by group and cut into blocks of <= 8192
 * cells of ONE group.  Each block is stored gene-major as SELL-64 (sliced ELLPACK, 64 genes per slice,
 * genes sorted by their nnz inside the block so a slice has no padding to speak of):
 *      entry (uint32) = cell_local (13 bits) | count << 13 (19 bits, 0 = padding)
 *      ent[blk_base[b] + slice_ptr[b][t] + j*64 + lane]   j < slice_w[b][t],  gene = perm[b][t*64+lane]
 * so that lane-per-gene kernels read perfectly coalesced 256-B rows and need no cross-lane reduction.
 */
#ifndef MEMENTO_HIP_H
#define MEMENTO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MM_CELL_BITS 13
#define MM_BLOCK_CELLS (1 << MM_CELL_BITS) /* max cells per count block */
#define MM_MAX_COUNT ((1u << (32 - MM_CELL_BITS)) - 1u)

const char *mm_last_error(void);
int mm_version(void);
int mm_device_count(void);
int mm_set_device(int dev);

/* raw device-memory helpers so a ctypes-only caller needs nothing but this library */
int mm_malloc(void **d_ptr, size_t bytes);
int mm_free(void *d_ptr);
int mm_memset(void *d_ptr, int value, size_t bytes, void *stream);
int mm_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream);
int mm_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream);
int mm_sync(void *stream);
/* HIP-event timing of whatever is enqueued between begin and end on `stream` (bench.py roofline) */
int mm_timer_create(void **timer);
int mm_timer_begin(void *timer, void *stream);
int mm_timer_end(void *timer, void *stream);
int mm_timer_elapsed_ms(void *timer, float *ms); /* synchronises on the end event */
int mm_timer_destroy(void *timer);

/* Measurement aid (no reference counterpart): streaming read of n_bytes (multiple of 64 KiB) with the access pattern of
 * mm_moments1d_sell (64 KiB work items, dwordx4 per lane, 8 rows in flight).  mode 0: loads only -- the read bandwidth this
 * device reaches for that pattern; 1: + the per-entry 8-byte LDS gather; 2: + the fp64 arithmetic; 3: + five result stores per
 * (chunk, lane) (d_sink must then hold n_bytes / 65536 * 64 * 32 bytes).  tools/hbm_read_peak.py */
int mm_debug_read_probe(const void *d_src, int64_t n_bytes, int32_t n_workgroups, int32_t mode, uint32_t *d_sink, void *stream);

/* ---- K3: row sums of the CSR, optionally restricted to a gene mask ---------------------------
 * replaces X.sum(axis=1) / X.multiply(mask).sum(axis=1)   memento/estimator.py:65, :73 */
int mm_csr_rowsum(const int64_t *d_indptr, const int32_t *d_indices, const float *d_data, int64_t n_rows,
                  const uint8_t *d_gene_mask /* NULL = all genes */, double *d_out, void *stream);

/* ---- gene sharding on the device (multi-GPU: every rank keeps all cells x its own gene range) --------------------
 * replaces the host-side X[:, lo:hi] (scipy fancy indexing, O(nnz)) in front of the reference's per-gene fan-out
 * (memento/main.py:379-397).  mm_csr_colcount: d_row_nnz[r] = entries of row r with col_lo <= column < col_hi.
 * The caller builds d_out_indptr as the exclusive scan of d_row_nnz (n_rows + 1 values); mm_csr_colsplit then writes those
 * entries, in their original order and with column - col_lo, to d_out_indices / d_out_data. */
int mm_csr_colcount(const int64_t *d_indptr, const int32_t *d_indices, int64_t n_rows, int32_t col_lo, int32_t col_hi,
                    int64_t *d_row_nnz, void *stream);
int mm_csr_colsplit(const int64_t *d_indptr, const int32_t *d_indices, const float *d_data, int64_t n_rows, int32_t col_lo,
                    int32_t col_hi, const int64_t *d_out_indptr, int32_t *d_out_indices, float *d_out_data, void *stream);

/* The same for an arbitrary gene SET (cost-balanced shards are not contiguous): d_col_map[g] = new id of a kept gene, -1 = dropped;
 * the new ids must ascend with g so that rows stay sorted.  mm_csr_colsum: per-gene totals (the cost the balancing uses). */
int mm_csr_mapcount(const int64_t *d_indptr, const int32_t *d_indices, int64_t n_rows, const int32_t *d_col_map, int64_t *d_row_nnz,
                    void *stream);
int mm_csr_mapsplit(const int64_t *d_indptr, const int32_t *d_indices, const float *d_data, int64_t n_rows, const int32_t *d_col_map,
                    const int64_t *d_out_indptr, int32_t *d_out_indices, float *d_out_data, void *stream);
int mm_csr_colsum(const int32_t *d_indices, const float *d_data, int64_t nnz, double *d_out /* [n_genes], zeroed by the caller */,
                  void *stream);

/* ---- K0: ingest = CSR -> group-ordered SELL count blocks --------------------------------------
 * replaces util._select_cells(adata, group) = adata.X[mask].tocsc() per group
 *   memento/util.py:8-13, memento/main.py:128
 * d_cell_order[n_sel]: original cell index of every selected cell, sorted by group;
 * d_blk_cell0[n_blocks+1]: block b covers cell_order[blk_cell0[b] .. blk_cell0[b+1]) (one group each).
 * Step 1 counts nnz per (block, gene) and validates the data (integer valued, 0 < x <= MM_MAX_COUNT;
 * d_status[0] != 0 afterwards means invalid data).  */
int mm_sell_count(const int64_t *d_indptr, const int32_t *d_indices, const float *d_data, const int32_t *d_cell_order,
                  const int32_t *d_blk_cell0, int32_t n_blocks, int32_t n_genes, uint16_t *d_blk_cnt /* [nb][G] */,
                  int32_t *d_status, void *stream);
/* Step 2: per block sort genes by descending count -> rank/perm, slice widths/pointers and work items.
 * n_slices = ceil(G/64).  A slice is stored as slice_w[t] rows of 64 lanes x 4 entries (one dwordx4 per
 * lane per row; lane = gene slot, 4 consecutive entries of that gene).  slice_ptr is in rows, relative to
 * the block; d_blk_rows[b] = rows of block b (host scans it into blk_base, in rows).  A work item is
 * <= 64 rows of one slice; item_ptr[b][t] numbers them slice-major, d_blk_items[b] = items of block b. */
int mm_sell_layout(const uint16_t *d_blk_cnt, int32_t n_blocks, int32_t n_genes, int32_t *d_rank /* [nb][G] */,
                   int32_t *d_perm /* [nb][n_slices*64], -1 = no gene */, int32_t *d_slice_w /* [nb][n_slices] */,
                   int32_t *d_slice_ptr /* [nb][n_slices+1] */, int32_t *d_item_ptr /* [nb][n_slices+1] */,
                   int64_t *d_blk_rows /* [nb] */, int32_t *d_blk_items /* [nb] */, void *stream);
/* Step 3: scatter entries (d_ent must be zero-filled, 256 uint32 per row, sum(blk_rows) rows). */
int mm_sell_scatter(const int64_t *d_indptr, const int32_t *d_indices, const float *d_data, const int32_t *d_cell_order,
                    const int32_t *d_blk_cell0, int32_t n_blocks, int32_t n_genes, const int32_t *d_rank,
                    const int32_t *d_slice_ptr, const int64_t *d_blk_base, uint32_t *d_ent, void *stream);
/* Range-partitioned form of steps 1 and 3 for rows with strictly ascending column indices (canonical CSR): the path the
 * Python driver takes; mm_sell_count / mm_sell_scatter above remain for unsorted rows.  The gene ids are cut into
 * n_ranges = ceil(G / MM_RANGE_GENES) ranges of MM_RANGE_GENES = 1024 consecutive ids (G <= 65536).
 * mm_sell_split_count, one pass over the column indices: d_rowsplit[k][r], k = 0..R = absolute position in d_indices / d_data at
 * which range k begins in selected row r (block order, n_sel rows; range-major so that a (block, range) workgroup reads contiguous
 * runs); d_status |= 2 if some row is not strictly ascending or has a column outside [0, G); and d_blk_cnt as mm_sell_count fills
 * it, without the value check (d_blk_cnt must be ZERO-FILLED, 4-byte aligned and hold an even number of uint16: partial counts
 * are merged by 32-bit atomics).
 * mm_sell_scatter_ranges places the entries in the layout of mm_sell_scatter and validates the values (d_status |= 1: not a
 * positive integer count <= MM_MAX_COUNT); d_indices / d_data need no alignment beyond their element size.
 * One workgroup per (block, range).  Stores are written through on this chip -- a lone 4-byte store costs ~26 B of HBM write
 * traffic (measured: the unpartitioned scatter writes 7.8x its algorithmic bytes, profiles/r02_k1_traffic_C3.json) -- so the
 * scatter assembles entries in LDS, tile of <= 128 rows by tile, and stores only complete 16-byte groups (4 consecutive entries
 * of one gene).  An entry's position inside its gene is computed, not handed out: entries before the tile + bits below its row
 * in the gene's 128-bit row mask of the tile.  So a gene's entries are in ascending cell order and the ingest is DETERMINISTIC. */
#define MM_RANGE_SHIFT 10
#define MM_RANGE_GENES (1 << MM_RANGE_SHIFT)
#define MM_MAX_RANGES 64
int mm_sell_split_count(const int64_t *d_indptr, const int32_t *d_indices, const int32_t *d_cell_order, const int32_t *d_blk_cell0,
                        int32_t n_blocks, int64_t n_sel, int32_t n_genes, int32_t n_ranges,
                        int64_t *d_rowsplit /* [n_ranges+1][n_sel] */, uint16_t *d_blk_cnt /* [nb][G], zero-filled */,
                        int32_t *d_status, void *stream);
int mm_sell_scatter_ranges(const int64_t *d_indptr, const int32_t *d_indices, const float *d_data, const int32_t *d_cell_order,
                           const int32_t *d_blk_cell0, int32_t n_blocks, int32_t n_genes, int32_t n_ranges, int64_t n_sel,
                           const int64_t *d_rowsplit, const int32_t *d_rank, const int32_t *d_slice_ptr, const int64_t *d_blk_base,
                           uint32_t *d_ent, int32_t *d_status, void *stream);

/* ---- K1+K2: per-(item, gene slot) moment sums from the count blocks  (the HBM-roofline kernel) ---
 * replaces estimator._hyper_1d_relative sparse branch  memento/estimator.py:177-180
 *          and group_cells.mean(axis=0) / .max(axis=0)   memento/main.py:201, :206
 * d_inv_sf[n_sel]: 1/size_factor per selected cell in block order.
 * d_slab [sum(blk_items)][64] records of 32 bytes: {S1 = sum x/sf, S2 = sum x^2/sf^2, S3 = sum x/sf^2 (fp64),
 * SX = sum x (uint32, exact), MX = max x (uint32)} -- one contiguous 2 KiB run per work item.
 * d_blk_item_base = exclusive scan of blk_items. */
int mm_moments1d_sell(const uint32_t *d_ent, const int64_t *d_blk_base, const int32_t *d_slice_w, const int32_t *d_slice_ptr,
                      const int32_t *d_item_ptr, const int64_t *d_blk_item_base, const int32_t *d_blk_cell0,
                      const double *d_inv_sf, int32_t n_blocks, int32_t n_genes, void *d_slab, void *stream);
/* deterministic reduction of the slab over items and over the blocks of each group -> [n_groups][G] */
int mm_moments1d_reduce(const void *d_slab, const int32_t *d_rank, const int32_t *d_item_ptr, const int64_t *d_blk_item_base,
                        const int32_t *d_grp_blk0 /* [n_groups+1] first block of each group */, int32_t n_groups,
                        int32_t n_genes, double *d_out_S /* [3][n_groups][G] */, uint64_t *d_out_sumx /* [n_groups][G] */,
                        uint32_t *d_out_maxx /* [n_groups][G] */, void *stream);

/* ---- K5: integer histograms keyed (group, gene, sf_bin, count) --------------------------------
 * replaces bootstrap._unique_expr's np.unique over cells  memento/bootstrap.py:62-71 (the bins as a SET;
 * their replay ORDER is applied by mm_bins_order).  Pair p = gene_slot*n_groups + group for the tested
 * genes; d_gene_pairbase[G] = gene_slot*n_groups or -1 if the gene is not tested.  Table of pair p starts
 * at d_tab_ptr[p], is [n_sf_bins][xcap[p]] uint32 with xcap[p] = max count of the pair + 1
 * (column 0 is filled with the zero-count cells by mm_bins_count). Tables must be zeroed by the caller. */
int mm_hist1d_sell(const uint32_t *d_ent, const int64_t *d_blk_base, const int32_t *d_slice_w, const int32_t *d_slice_ptr,
                   const int32_t *d_item_ptr, const int32_t *d_perm, const int32_t *d_blk_cell0, const int32_t *d_blk_group,
                   const uint8_t *d_sf_bin /* [n_sel] block order */, int32_t n_blocks, int32_t n_genes,
                   const int32_t *d_gene_pairbase, const int64_t *d_tab_ptr, const int32_t *d_xcap, uint32_t *d_tab, void *stream);
/* zero-count column + number of non-empty bins per pair.  d_grp_bin_cells[n_groups][n_sf_bins] = cells per (group, sf bin) */
int mm_bins_count(uint32_t *d_tab, const int64_t *d_tab_ptr, const int32_t *d_xcap, int64_t n_pairs, int32_t n_groups,
                  int32_t n_sf_bins, const uint32_t *d_grp_bin_cells, int32_t *d_K /* [n_pairs] */, void *stream);

/* ---- K5b+K6 prep: order the bins like np.unique(code) and lay them out for the bootstrap ---------
 * code = count*r1 + r0*approx_sf[sf_bin] in IEEE fp64 (memento/bootstrap.py:62-67); ascending.
 * Pairs are assigned to lanes of 64-wide tiles (d_pair_slot[p] = tile*64+lane or -1 to skip; a tile may
 * leave lanes empty); tile t owns bin rows [tile_ptr[t], tile_ptr[t+1]) of 64 lanes each.  Per bin k the
 * kernel writes, at [(tile_ptr[t]+k)*64 + lane]:
 *   pk = pix[k]/remaining_p  with pix = mult/N_g and remaining_p as numpy's random_multinomial updates it
 *        (replicate-independent), lq = log(1-p) as the inversion sampler needs it,
 *   v = count, a = 1/sf, b = 1/sf^2                                              (fp64 each).
 * A pair may instead be given to the one-wave-per-chain kernel (mm_boot1d_chain): d_pair_slot[p] = MM_CHAIN_SLOT | row, and
 * the kernel writes the same five values of bin k as one 8-double record {pk, lq, v, a, b, -, -, -} at d_pk + 8*(row + k)
 * (the caller reserves that region behind the tile rows of the d_pk allocation).
 * d_status[0] |= 8 if two bins of a pair collide in code (np.unique would merge them). */
#define MM_CHAIN_SLOT (1LL << 62)
int mm_bins_order(const uint32_t *d_tab, const int64_t *d_tab_ptr, const int32_t *d_xcap, const int32_t *d_K,
                  const int64_t *d_pair_list, int64_t n_list /* pairs handled by this launch */,
                  int32_t big /* 0: K <= 1024 (one wave per pair); 1: K <= 8192 (512 threads per pair) */, int32_t n_groups,
                  int32_t n_sf_bins, const double *d_sf_table /* [n_sf_bins] approx size factor of each bin */,
                  const double *d_r1, const double *d_r0 /* per pair */, const int64_t *d_pair_slot, const int64_t *d_tile_ptr,
                  const double *d_grp_ncells /* [n_groups] */, double *d_pk, double *d_lq, double *d_v, double *d_a, double *d_b,
                  int32_t *d_status, void *stream);

/* Profiling hook (no reference counterpart): when d_buf != NULL, every later mm_boot1d_replay launch writes, per tile
 * (wave), {start, end (100 MHz wall clock), HW_ID, XCC_ID} into d_buf[tile*4 .. tile*4+3]; d_buf must hold 4*n_tiles
 * int64.  Pass NULL to switch it off.  Process-global; used by tools/replay_balance.py only. */
int mm_debug_wave_clock(int64_t *d_buf);
/* Measurement / test aid: exact != 0 makes the replay kernels (1D and 2D) evaluate every sampler search loop in numpy's fp64
 * arithmetic; 0 (default) uses the guarded fp32 evaluation of those loops (csrc/npy_rng.h: same integer draws, the guard sends
 * the ~0.1-0.5 % of draws that land near a decision threshold to the fp64 arithmetic).  Process-wide switch. */
int mm_debug_replay_arith(int32_t exact);
/* Measurement / test aid: on != 0 (default) lets every lane of mm_boot1d_replay produce its PCG64 uniforms AHEAD of their use, all
 * lanes together a fixed number of times per bin step, into a 16-slot ring in LDS (the draws read them back); 0 = every sampler
 * call site steps the generator for the lanes that draw there (rounds 1-2).  Same stream, same draws.  Process-wide switch. */
int mm_debug_replay_ring(int32_t on);
/* Timing experiments only -- WRONG results: rows > 0 makes every tile of mm_boot1d_replay read its operand rows modulo ``rows``
 * (a cache-resident region), which separates the kernel's arithmetic from its operand traffic.  0 (default) = off. */
int mm_debug_replay_rows_mod(int64_t rows);

/* ---- K6+K7: replay bootstrap -- numpy Generator(PCG64).multinomial draw-for-draw + replicate moments
 * replaces bootstrap._bootstrap_1d  memento/bootstrap.py:97-110 and the tuple branch of
 * estimator._hyper_1d_relative  memento/estimator.py:171-174, :182-183.
 * One lane per (gene, group) pair, one sequential PCG64 stream per lane (the reference re-seeds PCG64(5)
 * for every pair, bootstrap.py:102).  pcg_state = {state_hi, state_lo, inc_hi, inc_lo}.
 * d_slot_nobs = N_g, d_slot_omq = 1 - q_g of the slot's group.
 * Writes mean_b / var_b to d_out_mean[row*ld + 1 + b], row = d_slot_row[slot]; K == 1 pairs get NaN rows.
 * d_w_dump (optional, NULL in production) receives the int32 weights [slot][k][b] with stride kmax_dump. */
/* Optional argument of mm_boot1d_replay: tiles that are really chains of the one-wave-per-chain form (see mm_boot1d_chain, whose
 * operand records, per-chain arrays and jump table these are).  d_tile_chain[tile] = chain index or -1; a flagged tile owns no
 * bin rows (tile_ptr[t + 1] == tile_ptr[t]) and its wave runs the chain instead -- at the tile's place in the launch's dispatch
 * order, which is what the host's packing arranges. */
typedef struct mm_chain_tiles {
  const int32_t *d_tile_chain;
  const double *d_ops;
  const int64_t *d_ch_base;
  const int32_t *d_ch_K;
  const double *d_ch_nobs, *d_ch_omq;
  const int64_t *d_ch_row;
  const uint64_t *d_jump;
  int32_t *d_w_dump;
  int32_t kmax_dump;
} mm_chain_tiles;
int mm_boot1d_replay(const double *d_pk, const double *d_lq, const double *d_v, const double *d_a, const double *d_b,
                     const int64_t *d_tile_ptr, int64_t n_tiles, const int32_t *d_slot_K, const double *d_slot_nobs,
                     const double *d_slot_omq, const int64_t *d_slot_row, const uint64_t pcg_state[4], int32_t num_boot,
                     int32_t mean_only /* 1: estimator._mean_only_1p, replicates are [mean+1, 10] (estimator.py:188-204) */,
                     int64_t ld, double *d_out_mean, double *d_out_var, int32_t *d_w_dump, int32_t kmax_dump,
                     int64_t co_resident_waves /* waves of mm_boot1d_chain launched beside this call (0 = none): with n_tiles they
                                                  decide between the two- and the three-waves-per-SIMD build of the kernel */,
                     const mm_chain_tiles *chains /* NULL = every tile is a tile */,
                     const double *d_stream /* optional: the stream's uniforms from mm_pcg64_stream; every lane then reads its
                                               uniforms from this table at its own position instead of stepping PCG64 */,
                     int64_t stream_len, int32_t *d_stream_overflow /* set to 1 if a chain ran past the table: redo without it */,
                     void *stream);
/* K6+K7, FREE-RUNNING tiles (memento/bootstrap.py:97-110, estimator.py:171-174; the launch the timed path uses): slot s = 64 * tile + lane
 * runs the chain whose 8-double operand records (mm_bins_order, MM_CHAIN_SLOT pairs) are [d_slot_rec[s], d_slot_rec[s] + d_slot_K[s]) of
 * d_recs; d_slot_rec[s] < 0 or d_slot_K[s] <= 0 = unused lane.  The lanes of a tile share the instruction stream only: a BTPE draw makes
 * one attempt per bin step and a rejected lane retries in the next, a lane that finishes a replicate starts its next one at once.  Same
 * draws, replicate means and variances as mm_boot1d_replay, bit for bit.  ``chains`` as there (tiles that are one-wave-per-chain chains).
 * d_w_dump (optional): int32 weights [slot][k][b], stride kmax_dump. */
int mm_boot1d_free(const double *d_recs, const int64_t *d_slot_rec, int64_t n_tiles, const int32_t *d_slot_K, const double *d_slot_nobs,
                   const double *d_slot_omq, const int64_t *d_slot_row, const uint64_t pcg_state[4], int32_t num_boot, int32_t mean_only,
                   int64_t ld, double *d_out_mean, double *d_out_var, int32_t *d_w_dump, int32_t kmax_dump, const mm_chain_tiles *chains,
                   void *stream);
/* d_out[i] = the (i + 1)-th uniform of Generator(PCG64(state)).random(): the ONE stream every chain of a launch replays
 * (memento/bootstrap.py:102 seeds PCG64(5) per pair), produced once by lane-parallel jump-ahead. */
int mm_pcg64_stream(const uint64_t pcg_state[4], int64_t n, double *d_out, void *stream);

/* K6+K7 for LONG chains: one WAVE per (gene, group) chain instead of one lane.  A chain is one sequential PCG64 stream
 * (memento/bootstrap.py:102 re-seeds PCG64(5) per pair), so nothing but the generator itself parallelises: the 64 lanes
 * produce the next 64 outputs of the stream in one step (PCG64 is an LCG: state j steps on = A^j s + C_j; d_jump[lane] =
 * {A^(lane+1) hi, lo, C_(lane+1) hi, lo} for the stream's increment) and the wave-uniform samplers consume them in order.
 * Draws, replicate means and variances are bit-identical to mm_boot1d_replay's.  Operands: 8-double records written by
 * mm_bins_order for MM_CHAIN_SLOT pairs; chain c reads records [d_ch_base[c], d_ch_base[c] + d_ch_K[c]) of d_ops, needs
 * d_ch_K[c] >= 2, and writes row d_ch_row[c].  d_w_dump (optional) receives int32 weights [chain][k][b], stride kmax_dump. */
int mm_boot1d_chain(const double *d_ops, const int64_t *d_ch_base, const int32_t *d_ch_K, const double *d_ch_nobs,
                    const double *d_ch_omq, const int64_t *d_ch_row, int64_t n_chains, const uint64_t *d_jump /* [64][4] */,
                    const uint64_t pcg_state[4], int32_t num_boot, int32_t mean_only, int64_t ld, double *d_out_mean,
                    double *d_out_var, int32_t *d_w_dump, int32_t kmax_dump, void *stream);

/* K6+K7, lane-ASYNCHRONOUS tiles: one lane per chain like mm_boot1d_replay, but every lane walks its own chain at its own pace
 * (the draw is a small state machine, csrc/npy_rng.h: lane_begin / lane_inv / lane_att / ...; one pass of a wave runs each phase
 * for the lanes that are in it), so no lane waits for the longest search, the unluckiest BTPE draw or the longest chain of its
 * wave.  Slot s = 64 * wave + lane runs chain s: records [d_ch_base[s], d_ch_base[s] + d_ch_K[s]) of d_ops (the 8-double records
 * mm_bins_order writes for MM_CHAIN_SLOT pairs), row d_ch_row[s]; d_ch_K[s] < 2 = unused lane.  Same draws and replicate moments
 * as mm_boot1d_replay, bit for bit.  d_w_dump (optional): int32 weights [slot][k][b], stride kmax_dump. */
int mm_boot1d_async(const double *d_ops, const int64_t *d_ch_base, const int32_t *d_ch_K, const double *d_ch_nobs,
                    const double *d_ch_omq, const int64_t *d_ch_row, int64_t n_slots, const uint64_t pcg_state[4], int32_t num_boot,
                    int32_t mean_only, int64_t ld, double *d_out_mean, double *d_out_var, int32_t *d_w_dump, int32_t kmax_dump,
                    void *stream);

/* FAST mode of K6+K7: one lane = one replicate, one wave = 64 replicates of one pair; every (pair, replicate)
 * has its own PCG64 stream derived from (seed, row, replicate).  Same sampler code and moment arithmetic as the
 * replay kernel, different random numbers: statistically equivalent to the reference, not draw-for-draw
 * identical.  n_slots = number of slots in the tile layout written by mm_bins_order (64 * n_tiles). */
int mm_boot1d_fast(const double *d_pk, const double *d_lq, const double *d_v, const double *d_a, const double *d_b,
                   const int64_t *d_tile_ptr, int64_t n_slots, const int32_t *d_slot_K, const double *d_slot_nobs,
                   const double *d_slot_omq, const int64_t *d_slot_row, uint64_t seed, int32_t num_boot, int32_t mean_only,
                   int64_t ld, double *d_out_mean, double *d_out_var, void *stream);

/* ---- K8: residual variance, invalid-replicate fill, log  ---------------------------------------
 * replaces estimator._residual_variance + hypothesis_test._fill + np.log
 *   memento/estimator.py:103-111, memento/hypothesis_test.py:186-197.
 * In place on rows [n_rows][ld]; column 0 (true values, already logged by the host) is left alone.
 * fill_mode 0: replace invalid entries by a uniformly chosen valid replicate using a counter-based
 *   RNG (own stream: statistically equivalent to np.random.choice, not the same draws);
 * fill_mode 1: leave invalid entries as NaN (the strict host driver patches them with np.random).
 * d_n_invalid[row][2] = number of invalid (mean, res_var) replicates; a row with no valid entry is
 * reported as -1. */
int mm_boot_fill_log(double *d_mean, double *d_var, int64_t n_rows, int64_t ld, int32_t num_boot, const double mv_fit[3],
                     int32_t fill_mode, uint64_t fill_seed, int32_t *d_n_invalid,
                     const int64_t *d_row_key /* optional [n_rows]: the refill stream of a row is keyed by this instead of the row
                                                 number, so that it is the same however the rows were chunked / sharded */,
                     void *stream);

/* ---- K9+K10: per-test linear contraction over groups and null statistics ------------------------
 * replaces hypothesis_test._regress_1d (linear part) and the counting part of _compute_asl
 *   memento/hypothesis_test.py:249-251, :262-271, :290-298, :62-92.
 * test t: rows of gene test_gene[t] are [gene*n_groups + j]; coef_b = sum_j W[t][j] * y[row_j][b] over
 * good groups (d_good[gene][j] != 0); replicate b is dropped if any good row is non-finite in either
 * d_ym or d_yv (valid_boostrap_iters).  which = 0 uses d_ym, 1 uses d_yv as the response.
 * d_coef[t][ld] receives the coefficient row (NaN where dropped); d_stats[t][8] =
 *   {coef0, se (nanstd, ddof 0), n_valid_null, extreme_count, null_mean, all_equal, extreme_count_raw, range}:
 *   extreme_count / null_mean are those of null = coef[1:] - coef[0] (resampling == 'bootstrap', :66-68),
 *   extreme_count_raw counts |coef[b]| > |coef0| on the un-centred replicates (any other resampling value, :69-70; their
 *   mean is null_mean + coef0); range = max - min over all columns. */
int mm_contract_stats(const double *d_ym, const double *d_yv, int64_t ld, int32_t num_boot, int32_t n_groups,
                      const int32_t *d_test_gene, const double *d_W /* [n_tests][n_groups] */, const uint8_t *d_good /* [n_genes][n_groups] */,
                      int64_t n_tests, int32_t which, double *d_coef, double *d_stats, void *stream);

/* ---- resample_rep=True: hierarchical resampling of replicate groups ------------------------------------
 * replaces hypothesis_test._regress_1d :249-254, :273-286, _regress_2d :372-377, :393-404 and _cross_coef_resampled :231-239.
 * mm_valid_cols: the replicate columns that survive valid_boostrap_iters (:249-251): d_col_map[gene][k] = k-th column in
 * which every good group is finite in BOTH d_ym and d_yv (pass the same pointer twice for the 2D correlation rows),
 * d_n_valid[gene] = their number.  d_col_map is [n_genes][num_boot + 1].
 * mm_residualize: d_dst = M d_src on the [n_genes*n_groups][ld] rows (columns 0..n_cols-1; d_dst must not alias d_src);
 * gene g uses the n_groups x n_groups matrix d_M[d_gene_mask[g]] (zero rows/columns on invalid groups -> NaN rows).
 * Any number of groups (one group per donor: analysis/lupus/run_memento.py:31-52). */
int mm_valid_cols(const double *d_ym, const double *d_yv, int64_t ld, int32_t num_boot, int32_t n_groups, const uint8_t *d_good,
                  int64_t n_genes, int32_t *d_col_map, int32_t *d_n_valid, void *stream);
int mm_residualize(const double *d_src, double *d_dst, int64_t ld, int32_t n_cols, int32_t n_groups, int64_t n_genes,
                   const int32_t *d_gene_mask, const double *d_M, void *stream);
/* mm_cross_resampled: coefficient of resampled column c < nb (column 0 = observed), nb = d_n_valid[gene] - 1 (num_boot when
 * d_col_map / d_n_valid are NULL).  d_tt[test][group] = residualised treatment; d_rep/d_bcol [gene][n_groups][num_boot] =
 * group index (into the gene's valid groups) and index (1..nb, into the surviving columns) of the replicate column drawn for
 * row i of column c (np.random.choice replay), or both NULL to draw them on the device.  Columns whose drawn groups all share
 * one treatment value (0/0 in the reference) are reported as NaN. */
int mm_cross_resampled(const double *d_yt, int64_t ld, int32_t num_boot, int32_t n_groups, const int32_t *d_test_gene,
                       const double *d_tt, const uint8_t *d_good, const double *d_Nc, const int16_t *d_rep, const int32_t *d_bcol,
                       const int32_t *d_col_map, const int32_t *d_n_valid, uint64_t seed, int64_t n_tests, double *d_coef,
                       double *d_stats, void *stream);

/* ---- two-group contrasts against a shared control (Perturb-seq batching) ---------------------
 * test t: coef_b = y[test_gene[t], test_grp[t]][b] - y[test_gene[t], ctrl][b] -- what _regress_1d computes for the two
 * groups {control, guide} with a binary treatment (hypothesis_test.py:269-291); the control's bootstrap is shared by all
 * guides instead of being recomputed per guide as in the reference's per-guide loop.  stats layout as mm_contract_stats,
 * for the mean and the variability response in one launch; nothing per-replicate is stored. */
int mm_contrast_stats(const double *d_ym, const double *d_yv, int64_t ld, int32_t num_boot, int32_t n_groups, int32_t ctrl,
                      const int32_t *d_test_gene, const int32_t *d_test_grp, const uint8_t *d_good, int64_t n_tests,
                      double *d_stats_mean, double *d_stats_var, void *stream);
/* coefficient rows [n_tests][ld] of selected contrasts (NaN where a replicate column is dropped) for the tail fits */
int mm_contrast_rows(const double *d_ym, const double *d_yv, int64_t ld, int32_t num_boot, int32_t n_groups, int32_t ctrl,
                     const int32_t *d_test_gene, const int32_t *d_test_grp, int64_t n_tests, int32_t which, double *d_out,
                     void *stream);

/* ==== 2D (gene pairs) ===========================================================================
 * K11 step 1: copy the columns of the n_cols genes with d_col_id[gene] = m >= 0 out of the SELL blocks into a
 * gene-contiguous store: entries of (block b, column m) at d_out[col_ptr[b*(n_cols+1)+m] .. col_ptr[b*(n_cols+1)+m+1])
 * (same packed uint32 entries; order inside a column is the storage order of the block). */
int mm_extract_cols(const uint32_t *d_ent, const int64_t *d_blk_base, const int32_t *d_slice_w, const int32_t *d_slice_ptr,
                    const int32_t *d_item_ptr, const int32_t *d_perm, int32_t n_blocks, int32_t n_genes, const int32_t *d_col_id,
                    int32_t n_cols, const int64_t *d_col_ptr /* [nb][n_cols+1] */, uint32_t *d_out, void *stream);
/* K11 step 2: prod[group][pair] = sum_c x_ci x_cj / sf_c^2
 * replaces estimator._hyper_cov_relative sparse branch (X.multiply(Y).sum)  memento/estimator.py:225-228
 * and the Gram product of _hyper_corr_symmetric  memento/estimator.py:253-255.
 * Pairs are grouped by their left column: left l owns pairs [left_ptr[l], left_ptr[l+1]), right_col[pair] = partner
 * column.  d_scratch: [n_blocks][n_pairs] fp64; d_out: [n_groups][n_pairs]. */
int mm_pair_cross(const uint32_t *d_cols, const int64_t *d_col_ptr, int32_t n_cols, const int32_t *d_blk_cell0,
                  const int32_t *d_grp_blk0, int32_t n_blocks, int32_t n_groups, const double *d_inv_sf, const int32_t *d_left_col,
                  const int64_t *d_left_ptr, int32_t n_left, const int32_t *d_right_col, int64_t n_pairs, double *d_scratch,
                  double *d_out, void *stream);
/* 2D histograms: table of q = pair*n_groups + group is [n_sf_bins][xcap_i[q]][xcap_j[q]] uint32 at tab_ptr[q] (zeroed
 * by the caller); counts the cells with x_j > 0.  replaces np.unique over two columns, memento/bootstrap.py:62-71 */
int mm_pair_hist(const uint32_t *d_cols, const int64_t *d_col_ptr, int32_t n_cols, const int32_t *d_blk_cell0,
                 const int32_t *d_blk_group, int32_t n_blocks, const uint8_t *d_sf_bin, const int32_t *d_left_col,
                 const int64_t *d_left_ptr, int32_t n_left, const int32_t *d_right_col, int32_t n_groups, const int64_t *d_tab_ptr,
                 const int32_t *d_xcap_i, const int32_t *d_xcap_j, uint32_t *d_tab, void *stream);
/* x_j == 0 column from the left gene's 1D table (d_hist_i + hist_ptr[q], [n_sf_bins][xcap_i[q]], as completed by
 * mm_bins_count) and the number of non-empty bins K[q]. */
int mm_pair_bins_count(uint32_t *d_tab, const int64_t *d_tab_ptr, const int32_t *d_xcap_i, const int32_t *d_xcap_j,
                       const uint32_t *d_hist_i, const int64_t *d_hist_ptr, int64_t n_q, int32_t n_sf_bins, int32_t *d_K,
                       void *stream);
/* replay order of the 2D bins: code = x_i*r1a + x_j*r1b + r0*approx_sf (bootstrap.py:62-65, two-column expr);
 * big = 0: K <= 1024, big = 1: K <= 4096 */
int mm_bins_order2d(const uint32_t *d_tab, const int64_t *d_tab_ptr, const int32_t *d_xcap_i, const int32_t *d_xcap_j,
                    const int32_t *d_K, const int64_t *d_pair_list, int64_t n_list, int32_t big, int32_t n_groups, int32_t n_sf_bins,
                    const double *d_sf_table, const double *d_r1a, const double *d_r1b, const double *d_r0,
                    const int64_t *d_pair_slot, const int64_t *d_tile_ptr, const double *d_grp_ncells, double *d_pk, double *d_lq,
                    double *d_v1, double *d_v2, double *d_a, double *d_b, int32_t *d_status, void *stream);
/* 2D replay bootstrap: replicate covariance + the two variances folded into the correlation
 * replaces bootstrap._bootstrap_2d + estimator._corr_from_cov   memento/bootstrap.py:119-157, estimator.py:273-292 */
int mm_boot2d_replay(const double *d_pk, const double *d_lq, const double *d_v1, const double *d_v2, const double *d_a,
                     const double *d_b, const int64_t *d_tile_ptr, int64_t n_tiles, const int32_t *d_slot_K,
                     const double *d_slot_nobs, const double *d_slot_omq, const int64_t *d_slot_row, const uint64_t pcg_state[4],
                     int32_t num_boot, int64_t ld, double *d_out_corr, void *stream);

/* The same replay with per-chain operand RECORDS instead of shared rows: slot s = 64 * tile + lane runs the chain whose 8-double records
 * (mm_bins_order2d for pairs given as d_pair_slot[q] = MM_CHAIN_SLOT | first record: pk, lq, x_i, x_j, 1/sf, 1/sf^2, two spare) are
 * [d_slot_rec[s], d_slot_rec[s] + d_slot_K[s]) of d_recs; d_slot_rec[s] < 0 = unused lane.  A lane then reads memory of its own whatever
 * bin it is on, so a BTPE draw can make one attempt per bin step and retry in the next (as mm_boot1d_replay does).  Same draws and
 * replicate correlations as mm_boot2d_replay, bit for bit.   memento/bootstrap.py:119-157, estimator.py:273-292 */
int mm_boot2d_replay_rec(const double *d_recs, const int64_t *d_slot_rec, int64_t n_tiles, const int32_t *d_slot_K,
                         const double *d_slot_nobs, const double *d_slot_omq, const int64_t *d_slot_row, const uint64_t pcg_state[4],
                         int32_t num_boot, int64_t ld, double *d_out_corr, void *stream);

/* ---- synthetic data generator (replaces memento/simulate.py:52-89 and :91-115) ----
 * Counter-based: transcriptome count z[cell][gene] ~ NB(mean[gene], size theta[gene]) is a pure function of (seed_z, cell, gene),
 * so nothing dense is ever stored.  One launch per pass, selected by `mode`:
 *   0: d_totals[cell]  = sum_g z                         (needed by the hypergeometric capture)
 *   1: d_row_nnz[cell] = number of genes with a captured count > 0
 *   2: write row `cell` of the captured CSR at d_row_ptr[cell] (d_out_indices int32 ascending, d_out_data float32 counts)
 *   3: d_raw_totals[cell] = sum_g nb (Gaussian-copula branch only, before the rescaling below)
 * process 0: multivariate hypergeometric capture of rint(q[cell] * total) molecules (simulate.py:104-109; needs d_totals);
 *         1: Poisson capture x ~ Poisson(q[cell] * z) (simulate.py:110-112);  2: no capture (x = z, d_qs may be NULL).
 * Gaussian-copula branch (simulate.py:70-89): d_gauss != NULL holds the correlated standard-normal scores, [gene][cell] float32
 * (Cholesky factor of the correlation matrix x white noise from mm_std_normal: a library GEMM on the caller's side); then
 * nb = nbinom.ppf(Phi(score)) replaces the own NB draw, and with d_cell_size != NULL (needs d_raw_totals from a mode-3 pass)
 * z = rint(nb / raw_total[cell] * cell_size[cell]) (simulate.py:86-89).  d_gauss == NULL: independent branch (:66-68).
 * Draws are NOT numpy's (different generators): statistical equivalence only. */
int mm_simulate(const double *d_mean, const double *d_theta, int32_t n_genes, int64_t n_cells, const double *d_qs, uint64_t seed_z,
                uint64_t seed_capture, int32_t process, int32_t mode, int64_t *d_totals, int64_t *d_row_nnz, const int64_t *d_row_ptr,
                int32_t *d_out_indices, float *d_out_data, const float *d_gauss, const double *d_cell_size, int64_t *d_raw_totals,
                void *stream);
/* n independent standard normals (float32), element i a pure function of (seed, i). */
int mm_std_normal(uint64_t seed, int64_t n, float *d_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MEMENTO_HIP_H */
