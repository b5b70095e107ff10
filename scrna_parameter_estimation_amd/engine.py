"""Host-side driver of the HIP kernels: device CSR, SELL count blocks, moments, histograms, bootstrap.

torch is used only as plumbing here (device allocations, H2D/D2H copies, the current HIP stream); all
arithmetic on the hot path happens in libmemento_hip.so (csrc/*.hip) through the C-ABI in
include/memento_hip.h.  Nothing in this module has a CPU fallback.
"""

import ctypes
import os

from ctypes import c_void_p

import numpy as np

from . import _lib

BLOCK_CELLS = 8192
MAX_COUNT = (1 << 19) - 1
RANGE_GENES, MAX_RANGES = 1024, 64   # include/memento_hip.h: MM_RANGE_GENES, MM_MAX_RANGES (range-partitioned ingest)
ORDER_SMALL_CAP = 1024
ORDER_BIG_CAP = 8192
ORDER_BIG_CAP_2D = 4096
# Lane packing of the replay kernel.  (a) tile widths: every tile gets the same budget K_max * (C0 + C1 * lanes), the budget
# is set so that about PACK_WAVES tiles come out (2 per SIMD, never more than 2048: a third wave on a SIMD is a second round);
# C0/C1/PACK_WAVES were tuned on the hardware together with the pairing order below (profiles/README.md).
PACK_C0 = 220.0
PACK_C1 = 3.0
PACK_WAVES = 2000
# tile target and constants when a third wave per SIMD pays (pack_lanes): 3 x 1024 slots.  Round 3: re-swept for the 1D tile kernel
# that makes one BTPE attempt per bin step -- its wide tiles step 17 % faster, so the balance moves to fewer, wider tiles and more
# lone chains (tools/pack_sweep.py, C3: 3.79 s with round 2's 220 / 3 / 2750, 3.24-3.26 s with these; C2, two waves per SIMD, keeps
# 220 / 3: 267-270 ms against 283).  The 2D kernel keeps the constants it was tuned with.
PACK3_C0 = 260.0
PACK3_C1 = 2.2
PACK_WAVES3 = 4000
PACK2D = (220.0, 3.0, 220.0, 3.0, 2750)
PACK_OVERSUB = 1024      # tiles the three-waves-per-SIMD packing may have beyond the 3 x 1024 resident slots: the shortest ones, dispatched last, start as
                         # the first workgroups retire (C3, tools/pack_sweep.py: 3.18 s with 3,071 tiles, 3.07 / 3.02 / 3.00 / 3.07 / 3.57 s with 3,300 / 3,600 / 4,000 / 4,500 / 5,000)
# (b) measured time of one bin step of a wave of L chains running as the OLDER wave of its SIMD, us (tools/replay_balance.py,
# C3): used to rank tiles by length for the dispatch order (pair_tiles) and to predict a packing's longest tile.  PACK_COST: the 1D
# kernel of round 3 (one BTPE attempt per bin step; a lone chain is a chain wave: 0.70-0.73 us for the longest ones, which are served
# first, 0.9-1.2 for the others); PACK_COST_2D: the 2D kernel (round-2 table).
_PACK_L = (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 18, 20, 24, 28, 32, 36, 40, 44, 48, 56, 64)
_PACK_US = (0.72, 1.42, 1.66, 1.88, 2.00, 2.08, 2.18, 2.26, 2.41, 2.49, 2.61, 2.68, 2.77, 2.86, 3.02, 3.13, 3.16, 3.17, 3.18, 3.26, 3.32, 3.60, 3.87)
_PACK_US_2D = (0.96, 1.30, 1.52, 1.72, 1.87, 1.97, 2.09, 2.20, 2.39, 2.56, 2.74, 2.83, 2.99, 3.13, 3.32, 3.55, 3.67, 3.79, 3.90, 4.05, 4.17, 4.50, 4.81)
PACK_COST = np.interp(np.arange(1, 65), _PACK_L, _PACK_US)
PACK_COST_2D = np.interp(np.arange(1, 65), _PACK_L, _PACK_US_2D if os.environ.get('MM_PACK2D_COST', '2d') == '2d' else _PACK_US)   # (env: tools only)
# (c) many-chain regime (more tiles than resident wave slots): 2 x 1024 slots; a SIMD retires ~1.46 lone-wave-seconds of tile
# time per second (older wave 1.0 + younger ~0.46); a tile may take at most PACK_TAIL of the work-bound run time
PACK_MAX_RESIDENT = 2048
PACK_RATE = 1024 * 1.46
PACK_TAIL = float(os.environ.get('MM_PACK_TAIL', '0.5'))   # (env: tools only)
PACK_FORCE = ""          # tools only (tools/pack_sweep.py): "res" / "cap" forces one of the two packings
PACK_LAST = {}          # diagnostics of the last pack_lanes decision (tools/)


def _torch():
    import torch

    return torch


def _stream():
    return c_void_p(_torch().cuda.current_stream().cuda_stream)


def dev(a, dtype=None):
    """numpy -> device tensor (contiguous)."""
    torch = _torch()
    a = np.ascontiguousarray(a, dtype=dtype)
    if a.dtype == np.uint64:  # torch has limited uint64 support: ship the bits as int64
        return torch.from_numpy(a.view(np.int64)).cuda()
    if a.dtype == np.uint32:
        return torch.from_numpy(a.view(np.int32)).cuda()
    if a.dtype == np.uint16:
        return torch.from_numpy(a.view(np.int16)).cuda()
    return torch.from_numpy(a).cuda()


def empty(shape, dtype):
    torch = _torch()
    return torch.empty(shape, dtype=dtype, device="cuda")


def zeros(shape, dtype):
    torch = _torch()
    return torch.zeros(shape, dtype=dtype, device="cuda")


def P(t):
    """device pointer of a tensor (or NULL)."""
    return c_void_p(0) if t is None else c_void_p(t.data_ptr())


def host(t, dtype=None):
    a = t.cpu().numpy()
    return a.view(dtype) if dtype is not None else a


class DeviceCSR:
    """The user's CSR on the device: indptr int64, indices int32, data float32 (integer-valued counts)."""

    @classmethod
    def from_device(cls, indptr, indices, data, shape):
        """Wrap CSR arrays that already live in HBM (torch cuda tensors: int64 / int32 / float32).  Precondition (checked by
        the ingest kernel, which raises): every stored value is a positive integer count -- no explicitly stored zeros."""
        torch = _torch()
        assert indptr.dtype == torch.int64 and indices.dtype == torch.int32 and data.dtype == torch.float32
        self = cls.__new__(cls)
        self.shape = (int(shape[0]), int(shape[1]))
        self.nnz = int(indices.numel())
        self.indptr, self.indices, self.data = indptr.contiguous(), indices.contiguous(), data.contiguous()
        return self

    def __init__(self, X):
        import scipy.sparse as sp

        if not sp.isspmatrix_csr(X):
            raise TypeError("X must be a scipy.sparse.csr_matrix")
        if not X.has_canonical_format:
            X = X.copy()
            X.sum_duplicates()
        if X.nnz and (X.data == 0).any():
            # explicitly stored zeros (common after subsetting or arithmetic on adata.X; the reference accepts any scipy CSR,
            # main.py:44): they carry no count, drop them from the device copy
            X = X.copy()
            X.eliminate_zeros()
        self.shape = X.shape
        self.nnz = int(X.nnz)
        data = X.data
        if data.dtype != np.float32:
            d32 = data.astype(np.float32)
            if not np.array_equal(d32.astype(data.dtype), data):
                raise ValueError("count matrix values are not exactly representable in float32")
            data = d32
        self.indptr = dev(X.indptr, np.int64)
        self.indices = dev(X.indices, np.int32)
        self.data = dev(data, np.float32)

    @property
    def nbytes(self):
        return self.indptr.numel() * 8 + self.indices.numel() * 4 + self.data.numel() * 4

    def colsplit(self, lo, hi):
        """The gene (column) range [lo, hi) as a new device CSR with columns renumbered from 0 -- the device-side replacement
        of the host ``X[:, lo:hi]`` in front of gene-sharded multi-GPU runs (mm_csr_colcount + mm_csr_colsplit)."""
        torch = _torch()
        lo, hi = int(lo), int(hi)
        if not (0 <= lo <= hi <= self.shape[1]):
            raise ValueError("column range out of bounds")
        n = self.shape[0]
        row_nnz = empty((max(1, n),), torch.int64)
        _lib.call("mm_csr_colcount", P(self.indptr), P(self.indices), n, lo, hi, P(row_nnz), _stream())
        indptr = zeros((n + 1,), torch.int64)
        if n:
            indptr[1:] = torch.cumsum(row_nnz[:n], 0)
        nnz = int(indptr[-1].item())
        indices = empty((max(1, nnz),), torch.int32)
        data = empty((max(1, nnz),), torch.float32)
        _lib.call("mm_csr_colsplit", P(self.indptr), P(self.indices), P(self.data), n, lo, hi, P(indptr), P(indices), P(data), _stream())
        return DeviceCSR.from_device(indptr, indices[:nnz], data[:nnz], (n, hi - lo))

    def colselect(self, genes):
        """The columns ``genes`` (ascending indices) as a new device CSR with columns renumbered 0..len(genes)-1: the gene SET
        of a cost-balanced shard (mm_csr_mapcount + mm_csr_mapsplit)."""
        torch = _torch()
        genes = np.asarray(genes, dtype=np.int64)
        if len(genes) and (np.diff(genes) <= 0).any():
            raise ValueError("gene indices must ascend")
        if len(genes) and not (0 <= genes[0] and genes[-1] < self.shape[1]):
            raise ValueError("gene index out of bounds")
        n = self.shape[0]
        col_map = np.full(self.shape[1], -1, dtype=np.int32)
        col_map[genes] = np.arange(len(genes), dtype=np.int32)
        d_map = dev(col_map)
        row_nnz = empty((max(1, n),), torch.int64)
        _lib.call("mm_csr_mapcount", P(self.indptr), P(self.indices), n, P(d_map), P(row_nnz), _stream())
        indptr = zeros((n + 1,), torch.int64)
        if n:
            indptr[1:] = torch.cumsum(row_nnz[:n], 0)
        nnz = int(indptr[-1].item())
        indices = empty((max(1, nnz),), torch.int32)
        data = empty((max(1, nnz),), torch.float32)
        _lib.call("mm_csr_mapsplit", P(self.indptr), P(self.indices), P(self.data), n, P(d_map), P(indptr), P(indices), P(data), _stream())
        return DeviceCSR.from_device(indptr, indices[:nnz], data[:nnz], (n, len(genes)))

    def colsum(self):
        """Per-gene totals (host array): what the cost-balanced gene sharding is computed from."""
        torch = _torch()
        out = zeros((max(1, self.shape[1]),), torch.float64)
        _lib.call("mm_csr_colsum", P(self.indices), P(self.data), self.nnz, P(out), _stream())
        return host(out)[: self.shape[1]]

    def rowsum(self, gene_mask=None):
        """K3: per-cell sums, optionally over a gene mask (estimator.py:65, :73)."""
        torch = _torch()
        out = empty((self.shape[0],), torch.float64)
        m = dev(np.asarray(gene_mask, dtype=np.uint8)) if gene_mask is not None else None
        _lib.call("mm_csr_rowsum", P(self.indptr), P(self.indices), P(self.data), self.shape[0], P(m), P(out), _stream())
        return host(out)


def auto_max_rows(ld, arrays=2, fraction=0.4):
    """Rows of replicate buffers ([rows][ld] fp64, ``arrays`` of them plus one coefficient buffer) that fit in
    ``fraction`` of the currently free HBM."""
    torch = _torch()
    free, _ = torch.cuda.mem_get_info()
    free += max(0, torch.cuda.memory_reserved() - torch.cuda.memory_allocated())    # blocks the caching allocator holds but nobody uses
    return max(1024, int(free * fraction) // (int(ld) * 8 * (arrays + 1)))


def plan_blocks(group_id, n_groups, block_cells=BLOCK_CELLS):
    """Order cells by group (stable) and cut every group into near-equal blocks of <= block_cells.

    Returns cell_order, blk_cell0 (nb+1), blk_group (nb), grp_blk0 (n_groups+1), grp_ncells."""
    group_id = np.asarray(group_id)
    sel = np.flatnonzero(group_id >= 0)
    order = sel[np.argsort(group_id[sel], kind="stable")].astype(np.int32)
    counts = np.bincount(group_id[sel], minlength=n_groups).astype(np.int64)
    blk_cell0, blk_group, grp_blk0 = [0], [], [0]
    pos = 0
    for g in range(n_groups):
        n = int(counts[g])
        nb = max(1, -(-n // block_cells)) if n > 0 else 0
        for k in range(nb):
            pos_end = pos + (n * (k + 1)) // nb - (n * k) // nb
            blk_group.append(g)
            blk_cell0.append(pos_end)
            pos = pos_end
        grp_blk0.append(len(blk_group))
    return (order, np.asarray(blk_cell0, dtype=np.int32), np.asarray(blk_group, dtype=np.int32),
            np.asarray(grp_blk0, dtype=np.int32), counts)


class CountBlocks:
    """K0 result: the group-ordered SELL-64x4 count blocks living in HBM (see include/memento_hip.h)."""

    def __init__(self, csr, group_id, n_groups, timing=None):
        """``timing``: an optional dict that receives the HIP-event time (ms) of each of the three K0 launches
        (bench.py's CSR -> moments roofline; the launches are otherwise untimed)."""
        torch = _torch()
        call = _lib.call if timing is None else (lambda name, *a: _timed_call(timing, name, *a))
        self.G = int(csr.shape[1])
        self.n_groups = int(n_groups)
        (self.cell_order, self.blk_cell0, self.blk_group, self.grp_blk0, self.grp_ncells) = plan_blocks(group_id, n_groups)
        nb = self.n_blocks = len(self.blk_group)
        G = self.G
        self.n_slices = (G + 63) // 64
        ns = self.n_slices
        s = _stream()
        self.d_cell_order = dev(self.cell_order)
        self.d_blk_cell0 = dev(self.blk_cell0)
        self.d_blk_group = dev(self.blk_group)
        self.d_grp_blk0 = dev(self.grp_blk0)
        blk_cnt_buf = zeros((nb * G + 2,), torch.int16)     # (the fused split + count merges 16-bit halves by 32-bit atomics)
        blk_cnt = blk_cnt_buf[:nb * G].view(nb, G)
        status = zeros((1,), torch.int32)
        bad_data = ValueError(f"count matrix must hold positive integer counts <= {MAX_COUNT} with valid column indices")
        # Range-partitioned ingest (rows with ascending column indices: every canonical CSR): a workgroup owns (block, range of
        # 1024 consecutive gene ids), so a row contributes one contiguous segment to it and the per-gene state of the range lives
        # in LDS (csrc/ingest.hip).
        n_sel = len(self.cell_order)
        R = -(-G // RANGE_GENES)
        if R > MAX_RANGES:
            raise ValueError(f"at most {MAX_RANGES * RANGE_GENES} genes per ingest")
        rowsplit = empty((R + 1, max(1, n_sel)), torch.int64)     # [range][row]
        call("mm_sell_split_count", P(csr.indptr), P(csr.indices), P(self.d_cell_order), P(self.d_blk_cell0), nb, n_sel, G, R,
             P(rowsplit), P(blk_cnt_buf), P(status), s)
        self.ranged = (int(status.item()) & 2) == 0
        if not self.ranged:       # unsorted rows (or column indices out of range: reported by the count kernel): the unpartitioned kernels
            status.zero_()
            call("mm_sell_count", P(csr.indptr), P(csr.indices), P(csr.data), P(self.d_cell_order), P(self.d_blk_cell0), nb, G,
                 P(blk_cnt), P(status), s)
            if int(status.item()) != 0:
                raise bad_data
        self.rank = empty((nb, G), torch.int32)
        self.perm = empty((nb, ns * 64), torch.int32)
        self.slice_w = empty((nb, ns), torch.int32)
        self.slice_ptr = empty((nb, ns + 1), torch.int32)
        self.item_ptr = empty((nb, ns + 1), torch.int32)
        blk_rows = empty((nb,), torch.int64)
        blk_items = empty((nb,), torch.int32)
        call("mm_sell_layout", P(blk_cnt), nb, G, P(self.rank), P(self.perm), P(self.slice_w), P(self.slice_ptr),
                  P(self.item_ptr), P(blk_rows), P(blk_items), s)
        rows = host(blk_rows)
        items = host(blk_items).astype(np.int64)
        base = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
        ibase = np.concatenate([[0], np.cumsum(items)]).astype(np.int64)
        self.total_rows = int(base[-1])
        self.total_items = int(ibase[-1])
        self.blk_base = dev(base[:-1])
        self.blk_item_base = dev(ibase[:-1])
        self.ent = zeros((max(1, self.total_rows) * 256,), torch.int32)
        if self.ranged:
            call("mm_sell_scatter_ranges", P(csr.indptr), P(csr.indices), P(csr.data), P(self.d_cell_order), P(self.d_blk_cell0), nb, G, R,
                 n_sel, P(rowsplit), P(self.rank), P(self.slice_ptr), P(self.blk_base), P(self.ent), P(status), s)
            if int(status.item()) & 1:
                raise bad_data
        else:
            call("mm_sell_scatter", P(csr.indptr), P(csr.indices), P(csr.data), P(self.d_cell_order), P(self.d_blk_cell0), nb, G,
                 P(self.rank), P(self.slice_ptr), P(self.blk_base), P(self.ent), s)
        del rowsplit
        self.blk_cnt = host(blk_cnt, np.uint16)            # nnz per (block, gene), host copy [nb][G]
        self.nnz_sel = int(self.blk_cnt.astype(np.int64).sum())
        del blk_cnt
        # slab for K1 partial sums (reused across calls)
        n = max(1, self.total_items) * 64
        self._slab = empty((n * 8,), torch.int32)          # 32-byte record per (item, lane): S1, S2, S3 (fp64), sum x, max x

    @property
    def ent_bytes(self):
        return self.total_rows * 1024

    def moments_bytes(self):
        """Algorithmic HBM bytes of one mm_moments1d_sell launch on this layout: the packed entries (4 B each
        incl. the slice padding actually stored), per-item pointers, the staged 1/sf and the slab written."""
        return (self.total_rows * 1024 + self.n_blocks * (self.n_slices * 12 + 8) + len(self.cell_order) * 8
                + self.total_items * 64 * 32)

    def launch_moments(self, d_inv_sf):
        """Enqueue only the K1 kernel (used by bench.py to time the roofline kernel)."""
        _lib.call("mm_moments1d_sell", P(self.ent), P(self.blk_base), P(self.slice_w), P(self.slice_ptr), P(self.item_ptr),
                  P(self.blk_item_base), P(self.d_blk_cell0), P(d_inv_sf), self.n_blocks, self.G, P(self._slab), _stream())

    def moments(self, inv_sf_cells):
        """K1+K2.  ``inv_sf_cells``: 1/size_factor per ORIGINAL cell index.  Returns host arrays
        S (3, n_groups, G) float64, sumx (n_groups, G) uint64, maxx (n_groups, G) uint32."""
        torch = _torch()
        d_inv = dev(np.asarray(inv_sf_cells, dtype=np.float64)[self.cell_order])
        self.launch_moments(d_inv)
        out_S = empty((3, self.n_groups, self.G), torch.float64)
        out_sx = empty((self.n_groups, self.G), torch.int64)
        out_mx = empty((self.n_groups, self.G), torch.int32)
        _lib.call("mm_moments1d_reduce", P(self._slab), P(self.rank),
                  P(self.item_ptr), P(self.blk_item_base), P(self.d_grp_blk0), self.n_groups, self.G, P(out_S), P(out_sx),
                  P(out_mx), _stream())
        return host(out_S), host(out_sx, np.uint64), host(out_mx, np.uint32)


def _timed_call(timing, name, *args):
    """_lib.call bracketed by HIP events on the launch stream (last argument); elapsed ms -> timing[name]."""
    t = c_void_p()
    ms = ctypes.c_float()
    _lib.call("mm_timer_create", ctypes.byref(t))
    _lib.call("mm_timer_begin", t, args[-1])
    _lib.call(name, *args)
    _lib.call("mm_timer_end", t, args[-1])
    _lib.call("mm_timer_elapsed_ms", t, ctypes.byref(ms))
    _lib.call("mm_timer_destroy", t)
    timing[name] = timing.get(name, 0.0) + float(ms.value)


def _tiles_from_lanes(lanes, n_act):
    slot_of = np.zeros(n_act, dtype=np.int64)
    n_tiles, pos = 0, 0
    while pos < n_act:
        L = int(lanes[pos])
        run_end = int(np.searchsorted(lanes, L, side="right"))
        cnt = run_end - pos
        idx = np.arange(cnt)
        slot_of[pos:run_end] = (n_tiles + idx // L) * 64 + idx % L
        n_tiles += -(-cnt // L)
        pos = run_end
    return slot_of, n_tiles


def pack_lanes(K_sorted_desc, target_waves, dense=False, consts=None, cost=None):
    """Lane packing of sequential chains into 64-wide tiles (one tile = one wave).

    ``K_sorted_desc``: steps of each chain, descending.  A chain is one sequential stream, so a wave costs about
    K_max(lanes) x c(L), c(L) = C0 + C1*L per bin step (lanes diverge between the inversion and BTPE samplers and every
    data-dependent loop runs for the slowest lane).  Wide waves are the most instruction-efficient, but the heaviest chain
    bounds the makespan; so every wave gets the same cost budget: L(K) = largest lane count with K*c(L) <= budget, and the
    budget is chosen (bisection) to yield about ``target_waves`` waves, all resident at once ("resident" packing).  With very
    many chains a second, work-bound packing (see below) is built as well and the one with the shorter predicted run time
    is returned.  ``dense``: plain 64-wide tiles (fast mode: one wave per chain, lanes = replicates).
    Returns (slot of every chain = tile*64 + lane, number of tiles)."""
    Ks = np.maximum(np.asarray(K_sorted_desc, dtype=np.float64), 1.0)
    n_act = len(Ks)
    if n_act == 0:
        return np.zeros(0, dtype=np.int64), 0
    if dense:
        return _tiles_from_lanes(np.full(n_act, 64, dtype=np.int64), n_act)
    C0, C1, C0_3, C1_3, WAVES3 = consts if consts is not None else (PACK_C0, PACK_C1, PACK3_C0, PACK3_C1, PACK_WAVES3)

    def resident(n_waves, C0, C1):
        def lanes_for(budget):
            return np.clip(np.floor((budget / Ks - C0) / C1), 1, 64)

        lo_b, hi_b = C0 + C1, float(Ks.max()) * (C0 + 64 * C1) * 4.0
        for _ in range(50):
            mid = 0.5 * (lo_b + hi_b)
            if (1.0 / lanes_for(mid)).sum() > n_waves:
                lo_b = mid
            else:
                hi_b = mid
        return np.maximum.accumulate(lanes_for(hi_b).astype(np.int64))

    lanes = resident(target_waves, C0, C1)
    c = PACK_COST if cost is None else cost
    # MANY chains (2D pair lists, hundreds of groups): 64-wide tiles no longer fit the resident wave slots, or they fit only
    # because even the longest chains were made 64 wide (a 64-wide step costs 6x a lone chain's, and that tile then runs
    # long after everything else has finished).  There the schedule should be work-bound instead: tiles run in several
    # rounds, the dispatcher refills slots as waves retire, every tile is capped at T = PACK_TAIL x (total work / machine
    # rate) with the measured step costs -- T is a fixed point because narrower tiles for the long chains add work.

    def width(T):
        return np.clip(np.searchsorted(c, T / Ks, side="right"), 1, 64)

    def work(L):
        return float((Ks * c[L - 1] / L).sum()) / PACK_RATE

    def gap(T):
        return T - PACK_TAIL * work(width(T))

    lo_t, hi_t = float(Ks[0]) * c[0], float(Ks[0]) * c[63]
    if gap(lo_t) >= 0:
        T = lo_t                      # the longest chain, alone in its wave, already is the tail
    elif gap(hi_t) <= 0:
        T = hi_t                      # so much work that even the longest chains can ride 64-wide
    else:
        for _ in range(40):
            mid = 0.5 * (lo_t + hi_t)
            if gap(mid) < 0:
                lo_t = mid
            else:
                hi_t = mid
        T = hi_t
    lanes_w = np.maximum.accumulate(width(T).astype(np.int64))
    # choose by predicted run time (in units of lone-wave step time): resident tiles finish when their longest tile does
    # (measured / predicted: C3 6.21 / 6.25 s; 2D kernel 1.33-1.38x on three sizes); the work-bound schedule takes about 1.35x
    # the larger of its ideal work time and its tile cap (C3 1.33x; 2D 1.4x on four sizes, after the same kernel factor)
    longest_resident = float((Ks * c[lanes - 1]).max())
    n_res = float((1.0 / lanes).sum())
    use_work_bound = n_res > PACK_MAX_RESIDENT or 1.35 * max(work(lanes_w), T) < longest_resident
    if PACK_FORCE:
        use_work_bound = PACK_FORCE == "cap" or n_res > PACK_MAX_RESIDENT
    PACK_LAST.update(chains=n_act, tiles_resident=n_res, longest_resident=longest_resident, work_resident=work(lanes),
                     T_work_bound=T, work_work_bound=work(lanes_w), tiles_work_bound=float((1.0 / lanes_w).sum()),
                     chosen="work-bound" if use_work_bound else "resident")
    if use_work_bound:
        lanes = lanes_w
    elif target_waves == PACK_WAVES and WAVES3 > PACK_WAVES:
        # A third wave per SIMD (the kernels' 3-wave build, taken above 2048 tiles): narrower tiles for the same chains.  It
        # pays when the longest tile gets shorter -- the waves of this kernel leave ~40 % of the issue slots empty, a third wave
        # fills some -- and costs when the longest chain is already alone in its tile (then only the work grows).  Measured
        # (tools/pack_sweep.py, 2000 vs 2750 tiles): C3 shapes with 250k / 500k / 1M cells 2889 -> 2738, 3162 -> 2993,
        # 3815 -> 3591 ms (model: longest tile -8 to -10 %); C2 368 -> 391 ms (model: longest tile unchanged).  Round-3 kernel: C3
        # 3.32-3.38 s with two waves per SIMD (2,018 tiles), 3.24-3.28 s with three (3,071; model -5 %); C2 (bounded by its longest
        # chain alone in a wave, 0.25 of 0.26 s) stays at two.
        lanes3, w3 = resident(WAVES3, C0_3, C1_3), WAVES3
        while _tiles_from_lanes(lanes3, n_act)[1] > 3 * PAIR_SLOTS + PACK_OVERSUB and w3 > PACK_WAVES:   # (whole tiles: every lane count rounds up)
            w3 -= 16
            lanes3 = resident(w3, C0_3, C1_3)
        if float((1.0 / lanes3).sum()) <= 3 * PAIR_SLOTS + PACK_OVERSUB and float((Ks * c[lanes3 - 1]).max()) <= 0.97 * longest_resident:
            lanes = lanes3
            PACK_LAST.update(chosen="resident, 3 waves / SIMD", tiles_resident=float((1.0 / lanes3).sum()))
    return _tiles_from_lanes(lanes, n_act)


# Which chains run one per WAVE (mm_boot1d_chain: wave-uniform samplers, the 64 lanes produce the PCG64 stream in batches)
# instead of one per lane of a tile.  CHAIN_LONE: every chain the packer leaves ALONE in its tile -- the long ones; a lone chain
# steps in 0.75 us there against 0.96-1.2 us as the only lane of a tile wave.  CHAIN_MIN_K > 0: in addition every chain with at
# least that many bins (tests / tools; 2 = all of them).  Both kernels replay the same draws, bit for bit.
CHAIN_SLOT = 1 << 62      # include/memento_hip.h: MM_CHAIN_SLOT
CHAIN_LONE = os.environ.get('MM_CHAIN_LONE', '1') != '0'        # (the MM_* environment overrides below exist for the measurement tools)
CHAIN_MIN_K = int(os.environ.get('MM_CHAIN_MIN_K', '0'))
# FEW chains (a gene shard of a multi-GPU run): with at most this many chains in a launch every chain gets a wave of its own --
# three rounds of the chip's wave slots at most; lane-sharing tiles only pay when there are far more chains than wave slots.
# Measured on the 8 / 4 / 2 cost-balanced gene shards of C3 (5.4k / 10.8k / 21.6k chains; bench.py --predict-shards): all chains one
# per wave 1.71-2.49 s / 3.13-3.19 s / 6.0 s per shard against 2.53-2.87 / 2.98-3.09 / 3.2-3.3 s with tiles + lone chains.
CHAIN_ALL_MAX = int(os.environ.get('MM_CHAIN_ALL_MAX', '8192'))
# The other chains: TILE_MODE "async" = lane-asynchronous tile kernel (mm_boot1d_async: every lane walks its own chain at its
# own pace, 64 chains of similar length per wave; chains with >= ASYNC_CHAIN_MIN_K bins go one per wave to mm_boot1d_chain);
# "lockstep" = round 1-2's tile kernel (mm_boot1d_replay with the cost-model packing below; kept for A/B runs and the 2D path).
TILE_MODE = os.environ.get('MM_TILE_MODE', 'lockstep')
# Tiles whose lanes also run free ACROSS replicates (mm_boot1d_free: per-chain operand records instead of shared rows).  OFF, measured at
# C3: 3.57 s against 2.99 s.  A lane no longer waits for the slowest lane of its replicate (1.13 bin steps per nominal step instead of
# 1.23) -- but those waits were cheap: the last steps of a replicate run with few lanes left and cost accordingly (3.21 us per step made on
# average against 4.12 us for a step of 64 busy lanes), so the launch pays for lane-steps either way; the records alone cost 3 %
# (3.09 s with the lanes meeting at the end of each replicate), the per-lane replicate epilogue 4 %.  Bit-identical, kept as a tested option.
TILE_FREE = os.environ.get('MM_TILE_FREE', '0') != '0'
# 2D replay on per-chain operand records with one BTPE attempt per bin step (mm_boot2d_replay_rec) instead of shared rows, every attempt in its step
BOOT2D_RECORDS = os.environ.get('MM_BOOT2D_RECORDS', '1') != '0'
ASYNC_CHAIN_MIN_K = int(os.environ.get('MM_ASYNC_CHAIN_MIN_K', '160'))
ASYNC_LANES = int(os.environ.get('MM_ASYNC_LANES', '64'))      # chains per wave of the async kernel (the other lanes idle)
CHAIN_CLOCK_OFF = 1 << 18   # int64 offset of the chain kernel's records in the mm_debug_wave_clock buffer (tools/)

PAIR_SLOTS = 1024      # SIMDs: tiles t and t + PAIR_SLOTS share one
PAIR_TILES = os.environ.get('MM_PAIR_TILES', '1') != '0'      # tools only: False = plain longest-first dispatch order


def pair_tiles(slot_of, n_tiles, K_of, cost=None):
    """Dispatch order of the replay tiles.  Two waves share a SIMD and the OLDER one is served first (measured: the
    younger wave runs at ~0.46 of its lone speed until the older retires, tools/replay_balance.py).  Workgroups are handed
    out round-robin, so tile t and tile t + 1024 meet on one SIMD: put the 1024 longest tiles first, longest first, and
    behind them the next 1024 in ASCENDING length -- the longest tile then shares its SIMD with the shortest partner
    instead of one of its own size; anything beyond 2048 follows longest-first and refills slots as they free up."""
    if not PAIR_TILES or n_tiles <= 2:
        return slot_of
    tile = slot_of // 64
    lanes = np.bincount(tile, minlength=n_tiles)
    kmax = np.zeros(n_tiles)
    np.maximum.at(kmax, tile, np.asarray(K_of, dtype=np.float64))
    est = kmax * (PACK_COST if cost is None else cost)[np.clip(lanes, 1, 64) - 1]
    by_len = np.argsort(-est, kind="stable")
    h = min(PAIR_SLOTS, n_tiles // 2 + n_tiles % 2)
    first, rest = by_len[:h], by_len[h:]
    second, tail = rest[:h][::-1], rest[h:]
    new_order = np.concatenate([first, second, tail])
    new_id = np.empty(n_tiles, dtype=np.int64)
    new_id[new_order] = np.arange(n_tiles)
    return new_id[tile] * 64 + slot_of % 64


_PCG_MULT = 0x2360ED051FC65DA44385DF649FCCF645
_JUMP_CACHE = {}
_SIDE_STREAM = []


def pcg64_jump_table(seed=5):
    """Device table [64][4] uint64 for mm_boot1d_chain: row j = (A^(j+1) hi, lo, C_(j+1) hi, lo) with
    state_(i+j+1) = A^(j+1) * state_i + C_(j+1) (mod 2^128) for np.random.PCG64(seed)'s increment."""
    if seed not in _JUMP_CACHE:
        inc = np.random.PCG64(seed).state["state"]["inc"]
        m64, M = (1 << 64) - 1, 1 << 128
        a, c, rows = 1, 0, []
        for _ in range(64):
            a = (a * _PCG_MULT) % M
            c = (c * _PCG_MULT + inc) % M
            rows.append([a >> 64, a & m64, c >> 64, c & m64])
        _JUMP_CACHE[seed] = dev(np.array(rows, dtype=np.uint64))
    return _JUMP_CACHE[seed]


REPLAY_RING = os.environ.get('MM_REPLAY_RING', '0') != '0'       # tile kernel: uniforms produced ahead into an LDS ring (mm_debug_replay_ring);
                            # OFF: measured slower at every top-up rate (C3: 3.61 / 3.71 / 3.80 s with 2 / 3 / 4 per step against 3.53 s)
_STREAM_CACHE = {}
STREAM_TABLE = os.environ.get('MM_STREAM_TABLE', '0') != '0'     # tile kernel: uniforms from the precomputed stream table;
                            # OFF: measured slower (C3: 4.12-4.16 s against 3.53 s -- a wave's lanes sit at 64 different places of the
                            # table, and the gather's latency is exposed in every attempt; profiles/README.md)
STREAM_PER_STEP = 3.0       # table length = this x (longest tile chain's steps) x replicates: a draw takes 1 (inversion) or 2 per
                            # BTPE attempt (~2.4 on average) uniforms; a chain past the table flags the launch, which is redone


def pcg64_stream_table(seed, n):
    """Device table of the first ``n`` uniforms of Generator(PCG64(seed)) (mm_pcg64_stream); cached and grown on demand -- its
    content depends on nothing but the seed."""
    torch = _torch()
    t = _STREAM_CACHE.get(seed)
    if t is None or t.numel() < n:
        n_alloc = int(n * 1.25) + 4096
        t = empty((n_alloc,), torch.float64)
        _lib.call("mm_pcg64_stream", pcg64_state(seed), n_alloc, P(t), _stream())
        _STREAM_CACHE[seed] = t
    return t


def side_stream():
    """A second HIP stream (torch plumbing) on which the chain kernel runs beside the tile kernel."""
    if not _SIDE_STREAM:
        _SIDE_STREAM.append(_torch().cuda.Stream())
    return _SIDE_STREAM[0]


def pcg64_state(seed=5):
    """(state_hi, state_lo, inc_hi, inc_lo) of np.random.PCG64(seed) -- what bootstrap.py:102 constructs."""
    st = np.random.PCG64(seed).state["state"]
    m = (1 << 64) - 1
    return (ctypes.c_uint64 * 4)(st["state"] >> 64, st["state"] & m, st["inc"] >> 64, st["inc"] & m)


class Bootstrap1D:
    """K5 -> K8 for a set of tested genes: histograms, bins, replay order, replay bootstrap, fill+log.

    After ``run`` the device holds ym/yv [n_pairs][B+1] (pair = gene_slot*n_groups + group; column 0 =
    log true mean / log true residual variance) ready for the contraction kernel.
    """

    def __init__(self, blocks, gene_idx, maxx, sf_bin_cells, sf_table, grp_q, num_boot):
        torch = _torch()
        self.blocks = blocks
        self.ng = blocks.n_groups
        self.gene_idx = np.asarray(gene_idx, dtype=np.int64)
        self.n_tested = len(self.gene_idx)
        self.n_pairs = self.n_tested * self.ng
        self.B = int(num_boot)
        self.ld = self.B + 1
        self.n_bins = int(len(sf_table))
        if self.n_bins > 256:
            raise ValueError("at most 256 size-factor bins are supported")
        self.sf_table = np.asarray(sf_table, dtype=np.float64)
        self.grp_q = np.asarray(grp_q, dtype=np.float64)
        s = _stream()
        b = blocks
        # per (group, sf_bin) cell counts (the zero-count bins come from these)
        bins_sorted = np.asarray(sf_bin_cells, dtype=np.uint8)[b.cell_order]
        grp_of_sorted = np.repeat(np.arange(self.ng), b.grp_ncells)
        self.grp_bin_cells = np.bincount(grp_of_sorted.astype(np.int64) * self.n_bins + bins_sorted, minlength=self.ng * self.n_bins) \
            .reshape(self.ng, self.n_bins).astype(np.uint32)
        d_bins = dev(bins_sorted)
        pairbase = np.full(b.G, -1, dtype=np.int32)
        pairbase[self.gene_idx] = (np.arange(self.n_tested) * self.ng).astype(np.int32)
        xcap = (np.asarray(maxx, dtype=np.int64)[:, self.gene_idx].T.reshape(-1) + 1).astype(np.int32)  # [pair]
        tab_ptr = np.concatenate([[0], np.cumsum(xcap.astype(np.int64) * self.n_bins)]).astype(np.int64)
        self.d_xcap, self.d_tab_ptr = dev(xcap), dev(tab_ptr[:-1])
        self.tab = zeros((max(1, int(tab_ptr[-1])),), torch.int32)
        d_pairbase = dev(pairbase)  # NB: every device operand must stay referenced until after the call
        _lib.call("mm_hist1d_sell", P(b.ent), P(b.blk_base), P(b.slice_w), P(b.slice_ptr), P(b.item_ptr), P(b.perm),
                  P(b.d_blk_cell0), P(b.d_blk_group), P(d_bins), b.n_blocks, b.G, P(d_pairbase), P(self.d_tab_ptr),
                  P(self.d_xcap), P(self.tab), s)
        self.d_K = empty((self.n_pairs,), torch.int32)
        d_gbc = dev(self.grp_bin_cells)
        _lib.call("mm_bins_count", P(self.tab), P(self.d_tab_ptr), P(self.d_xcap), self.n_pairs, self.ng, self.n_bins,
                  P(d_gbc), P(self.d_K), s)
        self.K = host(self.d_K)
        self.xcap, self.tab_ptr = xcap, tab_ptr

    def bins_of_pair(self, p):
        """Debug/test helper: the canonical (sf_bin, count, multiplicity) bins of pair p (host arrays)."""
        t0, xc = int(self.tab_ptr[p]), int(self.xcap[p])
        tab = host(self.tab[t0:t0 + self.n_bins * xc], np.uint32).reshape(self.n_bins, xc)
        bi, xi = np.nonzero(tab)
        return bi, xi, tab[bi, xi]

    def _order_on_host(self, p, r1, r0, slot, tile_ptr, ops):
        """Replay order + bootstrap operands of ONE pair computed on the host (same arithmetic as k_bins_order:
        code = count*r1 + r0*approx_sf ascending; pk = pix/remaining_p; log(1-p)) and written into its tile lane."""
        torch = _torch()
        bi, xi, mu = self.bins_of_pair(p)
        code = xi.astype(np.float64) * r1 + r0 * self.sf_table[bi]
        o = np.argsort(code, kind="stable")
        if len(o) > 1 and (np.diff(code[o]) == 0).any():
            raise NotImplementedError("two bins of one pair collided in the replay hash (np.unique would merge them)")
        bi, xi, mu = bi[o], xi[o].astype(np.float64), mu[o].astype(np.float64)
        pix = mu / float(self.blocks.grp_ncells[p % self.ng])
        rem = np.empty_like(pix)
        acc = 1.0
        for k in range(len(pix)):      # sequential rounding, exactly like numpy's remaining_p
            rem[k] = acc
            acc -= pix[k]
        pk = pix / rem
        peff = np.where(pk <= 0.5, pk, 1.0 - pk)
        with np.errstate(divide="ignore", invalid="ignore"):
            lq = np.log(1.0 - peff)
        sf = self.sf_table[bi]
        if slot & CHAIN_SLOT:      # 8-double records of the chain kernel, addressed from the start of ops[0]'s allocation
            rec = np.zeros((len(pk), 8))
            rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3], rec[:, 4] = pk, lq, xi, 1.0 / sf, 1.0 / (sf * sf)
            r0_ = (slot & (CHAIN_SLOT - 1)) * 8
            self._opsbuf[r0_:r0_ + rec.size] = dev(rec.reshape(-1))
            return
        idx = dev((int(tile_ptr[slot >> 6]) + np.arange(len(pk), dtype=np.int64)) * 64 + (slot & 63))
        for arr, vals in zip(ops, (pk, lq, xi, 1.0 / sf, 1.0 / (sf * sf))):
            arr[idx] = dev(vals)

    def weights_of(self, p):
        """Test helper (after run(dump_weights=True)): the int32 multinomial weights [K][B] of pair p, whichever kernel drew them."""
        K = int(self.K[p])
        if self.async_index[p] >= 0:
            return host(self.w_dump_async[int(self.async_slot[self.async_index[p]]), :K, :])
        if self.chain_index[p] >= 0:
            return host(self.w_dump_chain[int(self.chain_index[p]), :K, :])
        return host(self.w_dump[int(self.tile_slot[p]), :K, :])

    def alloc_outputs(self, true_mean_log, true_rv_log):
        """ym/yv [n_pairs][B+1] = NaN, column 0 = log true mean / log true residual variance (hypothesis_test.py:174)."""
        torch = _torch()
        self.ym = torch.full((self.n_pairs, self.ld), float("nan"), dtype=torch.float64, device="cuda")
        self.yv = torch.full((self.n_pairs, self.ld), float("nan"), dtype=torch.float64, device="cuda")
        self.ym[:, 0] = dev(np.asarray(true_mean_log, dtype=np.float64))
        self.yv[:, 0] = dev(np.asarray(true_rv_log, dtype=np.float64))

    def run(self, skip, r1, r0, mv_fit, fill_mode=0, fill_seed=0, dump_weights=False, pcg_seed=5, first_pair=0, target_waves=None,
            fast=False, mean_only=False, fill_keys=None):
        """Order bins, replay the bootstrap and fill/log for every pair >= ``first_pair`` that is not skipped.

        ``skip``[pair] bool; ``r1``/``r0``[pair] the two uniforms of bootstrap.py:62,65.  Rows below
        ``first_pair`` are left untouched (used by the strict replay driver).  ``fill_keys`` [pair] int64: keys of the device
        refill streams (fill_mode 0; default: the row number).  Returns n_invalid
        [n_pairs - first_pair][2] (host): invalid (mean, res_var) replicates per row, -1 = no valid one."""
        torch = _torch()
        s = _stream()
        ng, B, ld = self.ng, self.B, self.ld
        active = (~np.asarray(skip, dtype=bool)) & (self.K >= 2)
        active[:first_pair] = False
        act = np.flatnonzero(active)
        order_all = act[np.argsort(-self.K[act], kind="stable")]
        # the longest chains run one per wave (mm_boot1d_chain), the others one per lane of a tile (mm_boot1d_replay)
        n_chain = 0 if (fast or not CHAIN_MIN_K) else int(np.searchsorted(-self.K[order_all], -CHAIN_MIN_K, side="right"))
        if not fast and target_waves is None and 0 < len(order_all) <= CHAIN_ALL_MAX:
            n_chain = len(order_all)
        use_async = TILE_MODE == "async" and not fast and target_waves is None
        if use_async and ASYNC_CHAIN_MIN_K:
            n_chain = max(n_chain, int(np.searchsorted(-self.K[order_all], -ASYNC_CHAIN_MIN_K, side="right")))
        chain_pairs, order = order_all[:n_chain], order_all[n_chain:]
        async_pairs = order if use_async else order[:0]           # K descending: the 64 chains of a wave have similar lengths
        if use_async:
            order = order[:0]
        n_async = len(async_pairs)
        # replay: cost-model lane packing (one lane = one sequential chain); fast: dense 64-wide tiles (one WAVE per pair)
        slot_of, n_tiles = pack_lanes(self.K[order], PACK_WAVES if target_waves is None else target_waves, dense=fast)
        if not fast:
            slot_of = pair_tiles(slot_of, n_tiles, self.K[order])
        n_launch = n_chain                    # chains of the separate mm_boot1d_chain launch (CHAIN_MIN_K / ASYNC_CHAIN_MIN_K rules)
        tile_chain = None
        if CHAIN_LONE and not fast and n_tiles:
            # A chain the packer left alone in its tile takes a whole wave either way: that wave runs it in the wave-uniform
            # form (chain_body inside the tile kernel), at the tile's place in the dispatch order -- the packing and pairing
            # above stay what they are, the lone waves just step faster.
            tile = slot_of // 64
            lone = np.bincount(tile, minlength=n_tiles)[tile] == 1
            if lone.any():
                tile_chain = np.full(n_tiles, -1, dtype=np.int32)
                tile_chain[tile[lone]] = n_chain + np.arange(int(lone.sum()))
                chain_pairs = np.concatenate([chain_pairs, order[lone]])
                n_chain = len(chain_pairs)
                order, slot_of = order[~lone], slot_of[~lone]
        n_act = len(order)
        self.n_tiles, self.n_chain, self.n_async, self.n_chain_launch = n_tiles, n_chain, n_async, n_launch
        pair_slot = np.full(self.n_pairs, -1, dtype=np.int64)
        pair_slot[order] = slot_of
        slot_pair = np.full(n_tiles * 64, -1, dtype=np.int64)
        slot_pair[slot_of] = order
        slot_K = np.zeros(n_tiles * 64, dtype=np.int32)
        slot_K[slot_of] = self.K[order]
        tile_k = slot_K.reshape(n_tiles, 64).max(axis=1) if n_tiles else np.zeros(0, dtype=np.int32)
        tile_ptr = np.concatenate([[0], np.cumsum(tile_k.astype(np.int64))]).astype(np.int64)
        rows = int(tile_ptr[-1])
        self.draws_per_replicate = int(np.maximum(self.K[order_all] - 1, 0).sum())
        # one allocation: five [rows][64] operand planes of the tiles, then the chains' 8-double records
        plane = max(1, rows) * 64
        use_free = TILE_FREE and not fast and not use_async and not REPLAY_RING and not STREAM_TABLE and n_tiles > 0 and len(order) > 0
        if use_free:
            plane = 64                                              # no operand rows: every chain of the launch reads records
        rec_pairs = np.concatenate([chain_pairs, async_pairs, order if use_free else order[:0]])      # chains whose operands are 8-double records
        rec_K = self.K[rec_pairs].astype(np.int64)
        rec_base = np.concatenate([[0], np.cumsum(rec_K)]).astype(np.int64)
        ch_K, ch_base = rec_K[:n_chain], rec_base[:n_chain + 1]
        self._opsbuf = empty((5 * plane + 8 * max(1, int(rec_base[-1])),), torch.float64)
        ops = [self._opsbuf[i * plane:(i + 1) * plane] for i in range(5)]
        pair_slot[rec_pairs] = CHAIN_SLOT | (5 * plane // 8 + rec_base[:-1])
        self.chain_index = np.full(self.n_pairs, -1, dtype=np.int64)
        self.chain_index[chain_pairs] = np.arange(n_chain)
        self.async_index = np.full(self.n_pairs, -1, dtype=np.int64)
        self.async_index[async_pairs] = np.arange(n_async)
        self.chain_pairs, self.async_pairs = chain_pairs, async_pairs
        self.tile_slot = np.full(self.n_pairs, -1, dtype=np.int64)          # (tile, lane) slot of the chains that run as lanes of a tile
        self.tile_slot[order] = slot_of
        slot_rec = None
        if use_free:
            slot_rec = np.full(n_tiles * 64, -1, dtype=np.int64)
            slot_rec[slot_of] = rec_base[n_chain + n_async:-1]
        d_pair_slot, d_tile_ptr = dev(pair_slot), dev(tile_ptr)
        status = zeros((1,), torch.int32)
        d_r1, d_r0 = dev(np.asarray(r1, dtype=np.float64)), dev(np.asarray(r0, dtype=np.float64))
        d_sf, d_nc = dev(self.sf_table), dev(self.blocks.grp_ncells.astype(np.float64))
        small = order_all[self.K[order_all] <= ORDER_SMALL_CAP]
        big = order_all[(self.K[order_all] > ORDER_SMALL_CAP) & (self.K[order_all] <= ORDER_BIG_CAP)]
        huge = order_all[self.K[order_all] > ORDER_BIG_CAP]
        for lst, is_big in ((small, 0), (big, 1)):
            if len(lst):
                d_lst = dev(lst)
                _lib.call("mm_bins_order", P(self.tab), P(self.d_tab_ptr), P(self.d_xcap), P(self.d_K), P(d_lst), len(lst),
                          is_big, ng, self.n_bins, P(d_sf), P(d_r1), P(d_r0), P(d_pair_slot), P(d_tile_ptr), P(d_nc),
                          *[P(o) for o in ops], P(status), s)
        for p in huge:   # more bins than the in-LDS sort holds (very highly expressed genes): order them on the host
            self._order_on_host(int(p), float(r1[p]), float(r0[p]), int(pair_slot[p]), tile_ptr, ops)
        nobs = np.zeros(n_tiles * 64, dtype=np.float64)
        nobs[slot_of] = self.blocks.grp_ncells[order % ng]
        omq = np.zeros(n_tiles * 64, dtype=np.float64)
        omq[slot_of] = 1.0 - self.grp_q[order % ng]
        kmax_dump = int(tile_k.max()) if (dump_weights and n_tiles) else 0
        self.w_dump = zeros((n_tiles * 64, kmax_dump, B), torch.int32) if dump_weights else None
        self.kmax_dump = kmax_dump
        self.slot_pair, self.slot_K, self.pair_slot, self.tile_ptr = slot_pair, slot_K, pair_slot, tile_ptr
        self._ops, self._nobs = ops, nobs          # kept for diagnostics (tools/replay_balance.py)
        d_slot_K, d_nobs, d_omq, d_slot_pair = dev(slot_K), dev(nobs), dev(omq), dev(slot_pair)
        self.w_dump_chain = None
        chain_tiles = None
        if n_chain:
            kd = int(ch_K.max()) if dump_weights else 0
            self.w_dump_chain = zeros((n_chain, kd, B), torch.int32) if dump_weights else None
            d_chb, d_chK = dev(ch_base[:-1]), dev(ch_K.astype(np.int32))
            d_chn, d_cho = dev(self.blocks.grp_ncells[chain_pairs % ng].astype(np.float64)), dev(1.0 - self.grp_q[chain_pairs % ng])
            d_chr = dev(chain_pairs.astype(np.int64))
            d_recs = c_void_p(self._opsbuf.data_ptr() + 5 * plane * 8)
            if tile_chain is not None:          # chains that run as waves of the tile kernel's own launch
                d_tc = dev(tile_chain)
                chain_tiles = _lib.ChainTiles(P(d_tc), d_recs, P(d_chb), P(d_chK), P(d_chn), P(d_cho), P(d_chr), P(pcg64_jump_table(pcg_seed)),
                                              P(self.w_dump_chain), kd)
        if n_launch:
            # the chain kernel goes out first, on its own stream, and runs beside the tile kernel; the launch stream waits for
            # it before the fill / log pass
            side, main = side_stream(), torch.cuda.current_stream()
            side.wait_stream(main)
            _lib.call("mm_boot1d_chain", d_recs, P(d_chb), P(d_chK), P(d_chn), P(d_cho), P(d_chr),
                      n_launch, P(pcg64_jump_table(pcg_seed)), pcg64_state(pcg_seed), B, int(mean_only), ld, P(self.ym), P(self.yv),
                      P(self.w_dump_chain), kd, c_void_p(side.cuda_stream))
        self.w_dump_async = None
        if n_async:
            ka = int(rec_K[n_chain:].max()) if dump_weights else 0
            # slot = 64 * wave + lane; a wave takes ASYNC_LANES consecutive chains of the K-descending list, its other lanes idle
            L = max(1, min(64, int(ASYNC_LANES)))
            idx = np.arange(n_async)
            a_slot = (idx // L) * 64 + idx % L
            n_slots = int(a_slot[-1]) + 1
            self.async_slot = a_slot

            def spread(vals, dtype):
                out = np.zeros(n_slots, dtype=dtype)
                out[a_slot] = vals
                return dev(out)

            self.w_dump_async = zeros((n_slots, ka, B), torch.int32) if dump_weights else None
            d_ab, d_aK = spread(rec_base[n_chain:-1], np.int64), spread(rec_K[n_chain:], np.int32)
            d_an, d_ao = spread(self.blocks.grp_ncells[async_pairs % ng], np.float64), spread(1.0 - self.grp_q[async_pairs % ng], np.float64)
            d_ar = spread(async_pairs, np.int64)
            _lib.call("mm_boot1d_async", c_void_p(self._opsbuf.data_ptr() + 5 * plane * 8), P(d_ab), P(d_aK), P(d_an), P(d_ao), P(d_ar), n_slots,
                      pcg64_state(pcg_seed), B, int(mean_only), ld, P(self.ym), P(self.yv), P(self.w_dump_async), ka, s)
        if n_tiles and fast:
            _lib.call("mm_boot1d_fast", *[P(o) for o in ops], P(d_tile_ptr), n_tiles * 64, P(d_slot_K), P(d_nobs), P(d_omq), P(d_slot_pair),
                      int(fill_seed) & ((1 << 64) - 1), B, int(mean_only), ld, P(self.ym), P(self.yv), s)
        elif n_tiles and use_free:
            tab_over = None
            d_slot_rec = dev(slot_rec)
            _lib.call("mm_boot1d_free", c_void_p(self._opsbuf.data_ptr() + 5 * plane * 8), P(d_slot_rec), n_tiles, P(d_slot_K), P(d_nobs), P(d_omq),
                      P(d_slot_pair), pcg64_state(pcg_seed), B, int(mean_only), ld, P(self.ym), P(self.yv), P(self.w_dump), kmax_dump,
                      ctypes.byref(chain_tiles) if chain_tiles is not None else None, s)
        elif n_tiles:
            _lib.call("mm_debug_replay_ring", 1 if REPLAY_RING else 0)
            _lib.call("mm_debug_replay_rows_mod", int(os.environ.get('MM_DEBUG_ROWS_MOD', '0')))      # timing experiments only
            d_tab, tab_len, tab_over = None, 0, None
            if STREAM_TABLE and int(tile_k.max()) > 1:
                tab_len = int(STREAM_PER_STEP * (int(tile_k.max()) - 1) * B) + 4096
                d_tab = pcg64_stream_table(pcg_seed, tab_len)
                tab_over = zeros((1,), torch.int32)
            _lib.call("mm_boot1d_replay", *[P(o) for o in ops], P(d_tile_ptr), n_tiles, P(d_slot_K), P(d_nobs), P(d_omq), P(d_slot_pair),
                      pcg64_state(pcg_seed), B, int(mean_only), ld, P(self.ym), P(self.yv), P(self.w_dump), kmax_dump, n_launch,
                      ctypes.byref(chain_tiles) if chain_tiles is not None else None, P(d_tab), tab_len, P(tab_over), s)
        if n_launch:
            torch.cuda.current_stream().wait_stream(side)
        if n_tiles and not fast and tab_over is not None and int(tab_over.item()):
            # a chain consumed more uniforms than the table holds (never seen): the tile launch again with the arithmetic generator
            _lib.call("mm_boot1d_replay", *[P(o) for o in ops], P(d_tile_ptr), n_tiles, P(d_slot_K), P(d_nobs), P(d_omq), P(d_slot_pair),
                      pcg64_state(pcg_seed), B, int(mean_only), ld, P(self.ym), P(self.yv), P(self.w_dump), kmax_dump, n_launch,
                      ctypes.byref(chain_tiles) if chain_tiles is not None else None, None, 0, None, s)
        st = int(status.item())
        if st & 2 or st & 4:
            raise RuntimeError(f"mm_bins_order inconsistency (status {st})")
        if st & 8:
            raise NotImplementedError("two bins of one pair collided in the replay hash (np.unique would merge them)")
        self.raw_mean = self.ym.clone() if dump_weights else None
        self.raw_var = self.yv.clone() if dump_weights else None
        n_rows = self.n_pairs - first_pair
        n_inv = empty((max(1, n_rows), 2), torch.int32)
        fit = (ctypes.c_double * 3)(*[float(x) for x in mv_fit])
        if n_rows > 0:
            d_keys = dev(np.asarray(fill_keys, dtype=np.int64)[first_pair:]) if fill_keys is not None else None
            _lib.call("mm_boot_fill_log", c_void_p(self.ym.data_ptr() + first_pair * ld * 8), c_void_p(self.yv.data_ptr() + first_pair * ld * 8),
                      n_rows, ld, B, fit, int(fill_mode), int(fill_seed) & ((1 << 64) - 1), P(n_inv), P(d_keys), s)
        self.active = active
        return host(n_inv)[:n_rows]

    def contract(self, test_gene, W, good, which):
        """K9+K10 for tests (gene slot, weight row).  Returns (coef device tensor [n_tests][ld], stats host [n_tests][8])."""
        torch = _torch()
        n_tests = len(test_gene)
        coef = empty((max(1, n_tests), self.ld), torch.float64)
        stats = empty((max(1, n_tests), 8), torch.float64)
        d_tg, d_W, d_good = dev(np.asarray(test_gene, dtype=np.int32)), dev(np.asarray(W, dtype=np.float64)), dev(np.asarray(good, dtype=np.uint8))
        _lib.call("mm_contract_stats", P(self.ym), P(self.yv), self.ld, self.B, self.ng, P(d_tg), P(d_W), P(d_good), n_tests,
                  int(which), P(coef), P(stats), _stream())
        return coef, host(stats)[:n_tests]


    def contrast(self, test_gene, test_grp, ctrl, good):
        """Two-group contrasts against the control group ``ctrl`` (mm_contrast_stats): returns (stats_mean, stats_var)
        host arrays [n_tests][8]; ``rows(which, idx)`` gives the coefficient rows of selected tests on demand."""
        torch = _torch()
        n_tests = len(test_gene)
        st_m = empty((max(1, n_tests), 8), torch.float64)
        st_v = empty((max(1, n_tests), 8), torch.float64)
        d_tg, d_gr = dev(np.asarray(test_gene, dtype=np.int32)), dev(np.asarray(test_grp, dtype=np.int32))
        d_good = dev(np.asarray(good, dtype=np.uint8))
        _lib.call("mm_contrast_stats", P(self.ym), P(self.yv), self.ld, self.B, self.ng, int(ctrl), P(d_tg), P(d_gr), P(d_good), n_tests,
                  P(st_m), P(st_v), _stream())

        def rows(which, idx):
            idx = np.asarray(idx, dtype=np.int64)
            out = empty((max(1, len(idx)), self.ld), torch.float64)
            a, b = dev(np.asarray(test_gene, dtype=np.int32)[idx]), dev(np.asarray(test_grp, dtype=np.int32)[idx])
            _lib.call("mm_contrast_rows", P(self.ym), P(self.yv), self.ld, self.B, self.ng, int(ctrl), P(a), P(b), len(idx), int(which),
                      P(out), _stream())
            return host(out)[: len(idx)]

        return host(st_m)[:n_tests], host(st_v)[:n_tests], rows

    def valid_cols(self, good):
        """hypothesis_test.py:249-251 on the device: (col_map device tensor [n_tested][B+1], n_valid host [n_tested]) --
        the replicate columns in which every good group is finite in both the mean and the variability rows."""
        return _valid_cols(self.ym, self.yv, self.ld, self.B, self.ng, good, self.n_tested)

    def contract_resampled(self, test_gene, tt, good, which, gene_mask, M, Nc, rep=None, bcol=None, seed=0, col_map=None, n_valid=None):
        """resample_rep=True: residualise the response rows (into a scratch copy), then the resampled weighted slopes + null
        statistics.  ``rep``/``bcol`` [n_genes][n_groups][B] (int16/int32) replay np.random.choice; None -> drawn on the
        device.  ``col_map``/``n_valid`` (from ``valid_cols``): surviving replicate columns, None = all of them."""
        return _contract_resampled(self.yv if which else self.ym, self.ld, self.B, self.ng, self.n_tested, test_gene, tt, good, gene_mask,
                                   M, Nc, rep, bcol, seed, col_map, n_valid)


def _valid_cols(ym, yv, ld, B, ng, good, n_genes):
    torch = _torch()
    col_map = empty((max(1, n_genes), B + 1), torch.int32)
    n_valid = empty((max(1, n_genes),), torch.int32)
    d_good = dev(np.asarray(good, dtype=np.uint8))
    _lib.call("mm_valid_cols", P(ym), P(yv), ld, B, ng, P(d_good), n_genes, P(col_map), P(n_valid), _stream())
    return col_map, host(n_valid)[:n_genes]


def _contract_resampled(src, ld, B, ng, n_genes, test_gene, tt, good, gene_mask, M, Nc, rep, bcol, seed, col_map, n_valid):
    torch = _torch()
    s = _stream()
    n_tests = len(test_gene)
    if ng > 32767:
        raise NotImplementedError("resample_rep=True with more than 32767 groups")
    yt = torch.empty_like(src)
    d_gm, d_M = dev(np.asarray(gene_mask, dtype=np.int32)), dev(np.asarray(M, dtype=np.float64))
    _lib.call("mm_residualize", P(src), P(yt), ld, ld, ng, n_genes, P(d_gm), P(d_M), s)
    coef = empty((max(1, n_tests), ld), torch.float64)
    stats = empty((max(1, n_tests), 8), torch.float64)
    d_tg, d_tt, d_good = dev(np.asarray(test_gene, dtype=np.int32)), dev(np.asarray(tt, dtype=np.float64)), dev(np.asarray(good, dtype=np.uint8))
    d_nc = dev(np.asarray(Nc, dtype=np.float64))
    d_rep = dev(np.asarray(rep, dtype=np.int16)) if rep is not None else None
    d_bcol = dev(np.asarray(bcol, dtype=np.int32)) if bcol is not None else None
    d_nv = dev(np.asarray(n_valid, dtype=np.int32)) if n_valid is not None else None
    _lib.call("mm_cross_resampled", P(yt), ld, B, ng, P(d_tg), P(d_tt), P(d_good), P(d_nc), P(d_rep), P(d_bcol),
              P(col_map if n_valid is not None else None), P(d_nv), int(seed) & ((1 << 64) - 1), n_tests, P(coef), P(stats), s)
    return coef, host(stats)[:n_tests]


# =================================================================================================
# 2D: gene pairs
# =================================================================================================


class GeneColumns:
    """Gene-contiguous copy of the columns of ``genes`` (indices into the blocks' gene space), per block
    (mm_extract_cols).  The pair kernels join two columns through a dense per-cell LDS vector."""

    def __init__(self, blocks, genes):
        torch = _torch()
        self.blocks = blocks
        self.genes = np.asarray(genes, dtype=np.int64)
        M = self.n_cols = len(self.genes)
        nb = blocks.n_blocks
        lens = blocks.blk_cnt[:, self.genes].astype(np.int64)               # [nb][M]
        ptr = np.zeros((nb, M + 1), dtype=np.int64)
        flat = np.cumsum(lens.reshape(-1))
        total = int(flat[-1]) if flat.size else 0
        starts = np.concatenate([[0], flat[:-1]]).reshape(nb, M) if flat.size else np.zeros((nb, M), dtype=np.int64)
        ptr[:, :M] = starts
        ptr[:, M] = starts[:, -1] + lens[:, -1] if M else 0
        self.col_ptr = dev(ptr)
        col_id = np.full(blocks.G, -1, dtype=np.int32)
        col_id[self.genes] = np.arange(M, dtype=np.int32)
        d_col_id = dev(col_id)
        self.cols = zeros((max(1, total),), torch.int32)
        _lib.call("mm_extract_cols", P(blocks.ent), P(blocks.blk_base), P(blocks.slice_w), P(blocks.slice_ptr), P(blocks.item_ptr),
                  P(blocks.perm), nb, blocks.G, P(d_col_id), M, P(self.col_ptr), P(self.cols), _stream())
        self.slot_of_gene = {int(g): i for i, g in enumerate(self.genes)}


def _group_pairs(col1, col2):
    """Sort pairs by left column -> (order, left_col, left_ptr, right_col_sorted)."""
    col1 = np.asarray(col1, dtype=np.int64)
    col2 = np.asarray(col2, dtype=np.int64)
    order = np.argsort(col1, kind="stable")
    c1s = col1[order]
    left_col, start = np.unique(c1s, return_index=True)
    left_ptr = np.concatenate([start, [len(c1s)]]).astype(np.int64)
    return order, left_col.astype(np.int32), left_ptr, col2[order].astype(np.int32)


def pair_cross(cols, col1, col2, inv_sf_cells):
    """prod[group][pair] = sum_c x_ci x_cj / sf_c^2 for pairs of column slots (K11).  Host array."""
    torch = _torch()
    b = cols.blocks
    order, left_col, left_ptr, right = _group_pairs(col1, col2)
    n_pairs = len(order)
    d_inv = dev(np.asarray(inv_sf_cells, dtype=np.float64)[b.cell_order])
    d_left, d_lptr, d_right = dev(left_col), dev(left_ptr), dev(right)
    scratch = empty((b.n_blocks, max(1, n_pairs)), torch.float64)
    out = empty((b.n_groups, max(1, n_pairs)), torch.float64)
    _lib.call("mm_pair_cross", P(cols.cols), P(cols.col_ptr), cols.n_cols, P(b.d_blk_cell0), P(b.d_grp_blk0), b.n_blocks, b.n_groups,
              P(d_inv), P(d_left), P(d_lptr), len(left_col), P(d_right), n_pairs, P(scratch), P(out), _stream())
    res = np.empty((b.n_groups, n_pairs))
    res[:, order] = host(out)[:, :n_pairs]
    return res


def pair_table_bytes(maxx, genes, col1, col2, n_groups, n_bins):
    """Bytes of the dense (x_i, x_j, sf_bin) histogram tables Bootstrap2D allocates, per pair (all groups): used to size pair
    chunks against the free HBM (a pair of highly expressed genes costs tens of MB per group)."""
    cap = np.asarray(maxx, dtype=np.int64)[:, np.asarray(genes, dtype=np.int64)] + 1            # [group][column slot]
    return (cap[:, np.asarray(col1, dtype=np.int64)] * cap[:, np.asarray(col2, dtype=np.int64)]).sum(axis=0) * int(n_bins) * 4


class Bootstrap2D:
    """2D analogue of Bootstrap1D for a list of gene pairs (column slots of a GeneColumns store):
    (x_i, x_j, sf_bin) histograms -> bins -> replay order -> replay bootstrap of the correlation."""

    def __init__(self, cols, col1, col2, maxx, sf_bin_cells, sf_table, grp_q, num_boot):
        torch = _torch()
        self.cols = cols
        b = self.blocks = cols.blocks
        ng = self.ng = b.n_groups
        self.B, self.ld = int(num_boot), int(num_boot) + 1
        self.n_bins = len(sf_table)
        self.sf_table = np.asarray(sf_table, dtype=np.float64)
        self.grp_q = np.asarray(grp_q, dtype=np.float64)
        self.order, left_col, left_ptr, right = _group_pairs(col1, col2)      # kernels see pairs in this order
        self.n_pairs = len(self.order)
        self.n_q = self.n_pairs * ng
        s = _stream()
        # 1D tables of every column (left genes need them for the x_j == 0 column)
        self.h1 = Bootstrap1D(b, cols.genes, maxx, sf_bin_cells, sf_table, grp_q, num_boot)
        c1s = np.asarray(col1, dtype=np.int64)[self.order]
        c2s = np.asarray(col2, dtype=np.int64)[self.order]
        q_left = (c1s[:, None] * ng + np.arange(ng)[None, :]).reshape(-1)
        q_right = (c2s[:, None] * ng + np.arange(ng)[None, :]).reshape(-1)
        xcap_i = self.h1.xcap[q_left].astype(np.int32)
        xcap_j = self.h1.xcap[q_right].astype(np.int32)
        sizes = self.n_bins * xcap_i.astype(np.int64) * xcap_j.astype(np.int64)
        tab_ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        self.d_xi, self.d_xj, self.d_tab_ptr = dev(xcap_i), dev(xcap_j), dev(tab_ptr[:-1])
        self.tab = zeros((max(1, int(tab_ptr[-1])),), torch.int32)
        bins_sorted = np.asarray(sf_bin_cells, dtype=np.uint8)[b.cell_order]
        d_bins = dev(bins_sorted)
        d_left, d_lptr, d_right = dev(left_col), dev(left_ptr), dev(right)
        _lib.call("mm_pair_hist", P(cols.cols), P(cols.col_ptr), cols.n_cols, P(b.d_blk_cell0), P(b.d_blk_group), b.n_blocks, P(d_bins),
                  P(d_left), P(d_lptr), len(left_col), P(d_right), ng, P(self.d_tab_ptr), P(self.d_xi), P(self.d_xj), P(self.tab), s)
        d_hptr = dev(self.h1.tab_ptr[q_left])
        self.d_K = empty((max(1, self.n_q),), torch.int32)
        _lib.call("mm_pair_bins_count", P(self.tab), P(self.d_tab_ptr), P(self.d_xi), P(self.d_xj), P(self.h1.tab), P(d_hptr), self.n_q,
                  self.n_bins, P(self.d_K), s)
        self.K = host(self.d_K)[:self.n_q]
        self.xcap_i, self.xcap_j, self.tab_ptr = xcap_i, xcap_j, tab_ptr

    def bins_of(self, q):
        t0 = int(self.tab_ptr[q])
        ci, cj = int(self.xcap_i[q]), int(self.xcap_j[q])
        tab = host(self.tab[t0:t0 + self.n_bins * ci * cj], np.uint32).reshape(self.n_bins, ci, cj)
        bi, xi, xj = np.nonzero(tab)
        return bi, xi, xj, tab[bi, xi, xj]

    def _order_on_host(self, q, ra, rb, r0, slot, tile_ptr, ops):
        """Replay order + bootstrap operands of ONE (pair, group) with more bins than the in-LDS sort holds, computed on the host
        with the same arithmetic as k_bins_order2d: code = (x_i*r[0] + x_j*r[1]) + r0*approx_sf ascending (bootstrap.py:62-67)."""
        bi, xi, xj, mu = self.bins_of(q)
        code = (xi.astype(np.float64) * ra + xj.astype(np.float64) * rb) + r0 * self.sf_table[bi]
        o = np.argsort(code, kind="stable")
        if len(o) > 1 and (np.diff(code[o]) == 0).any():
            raise NotImplementedError("two bins of one pair collided in the replay hash (np.unique would merge them)")
        bi, xi, xj, mu = bi[o], xi[o].astype(np.float64), xj[o].astype(np.float64), mu[o].astype(np.float64)
        pix = mu / float(self.blocks.grp_ncells[q % self.ng])
        rem = np.empty_like(pix)
        acc = 1.0
        for k in range(len(pix)):      # sequential rounding, exactly like numpy's remaining_p
            rem[k] = acc
            acc -= pix[k]
        pk = pix / rem
        peff = np.where(pk <= 0.5, pk, 1.0 - pk)
        with np.errstate(divide="ignore", invalid="ignore"):
            lq = np.log(1.0 - peff)
        sf = self.sf_table[bi]
        if slot & CHAIN_SLOT:      # 8-double records (mm_boot2d_replay_rec), addressed from the start of ops[0]
            rec = np.zeros((len(pk), 8))
            rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3], rec[:, 4], rec[:, 5] = pk, lq, xi, xj, 1.0 / sf, 1.0 / (sf * sf)
            r0_ = (slot & (CHAIN_SLOT - 1)) * 8
            ops[0][r0_:r0_ + rec.size] = dev(rec.reshape(-1))
            return
        idx = dev((int(tile_ptr[slot >> 6]) + np.arange(len(pk), dtype=np.int64)) * 64 + (slot & 63))
        for arr, vals in zip(ops, (pk, lq, xi, xj, 1.0 / sf, 1.0 / (sf * sf))):
            arr[idx] = dev(vals)

    def run(self, skip, r1a, r1b, r0, true_corr, pcg_seed=5, target_waves=None):
        """All arrays are indexed by q = sorted_pair*n_groups + group (see ``self.order``).  Leaves the
        replicate correlations in self.yc [n_q][B+1] (column 0 = true correlation)."""
        torch = _torch()
        s = _stream()
        ng, B, ld = self.ng, self.B, self.ld
        active = (~np.asarray(skip, dtype=bool)) & (self.K >= 1)
        act = np.flatnonzero(active)
        order = act[np.argsort(-self.K[act], kind="stable")]
        slot_of, n_tiles = pack_lanes(self.K[order], PACK_WAVES if target_waves is None else target_waves, consts=PACK2D, cost=PACK_COST_2D)
        slot_of = pair_tiles(slot_of, n_tiles, self.K[order], cost=PACK_COST_2D)
        self.n_tiles = n_tiles
        pair_slot = np.full(self.n_q, -1, dtype=np.int64)
        pair_slot[order] = slot_of
        slot_pair = np.full(n_tiles * 64, -1, dtype=np.int64)
        slot_pair[slot_of] = order
        slot_K = np.zeros(n_tiles * 64, dtype=np.int32)
        slot_K[slot_of] = self.K[order]
        tile_k = slot_K.reshape(n_tiles, 64).max(axis=1) if n_tiles else np.zeros(0, dtype=np.int32)
        tile_ptr = np.concatenate([[0], np.cumsum(tile_k.astype(np.int64))]).astype(np.int64)
        rows = int(tile_ptr[-1])
        self.draws_per_replicate = int(np.maximum(self.K[order] - 1, 0).sum())
        self.wave_steps_per_replicate = rows
        use_rec = BOOT2D_RECORDS and n_tiles > 0
        slot_rec = None
        if use_rec:
            # per-chain operand records (8 doubles per bin) instead of [row][64] planes: a lane reads memory of its own wherever it is in
            # its chain, so the kernel can let a rejected BTPE attempt retry in the next bin step (mm_boot2d_replay_rec)
            rec_K = self.K[order].astype(np.int64)
            rec_base = np.concatenate([[0], np.cumsum(rec_K)]).astype(np.int64)
            ops = [empty((8 * max(1, int(rec_base[-1])),), torch.float64)] + [empty((8,), torch.float64) for _ in range(5)]
            self.tile_slot = pair_slot.copy()
            pair_slot[order] = CHAIN_SLOT | rec_base[:-1]
            slot_rec = np.full(n_tiles * 64, -1, dtype=np.int64)
            slot_rec[slot_of] = rec_base[:-1]
        else:
            ops = [empty((max(1, rows) * 64,), torch.float64) for _ in range(6)]
        d_pair_slot, d_tile_ptr = dev(pair_slot), dev(tile_ptr)
        status = zeros((1,), torch.int32)
        d_ra, d_rb, d_r0 = dev(np.asarray(r1a, np.float64)), dev(np.asarray(r1b, np.float64)), dev(np.asarray(r0, np.float64))
        d_sf, d_nc = dev(self.sf_table), dev(self.blocks.grp_ncells.astype(np.float64))
        small = order[self.K[order] <= ORDER_SMALL_CAP]
        big = order[(self.K[order] > ORDER_SMALL_CAP) & (self.K[order] <= ORDER_BIG_CAP_2D)]
        huge = order[self.K[order] > ORDER_BIG_CAP_2D]
        for q in huge:   # more bins than the in-LDS sort holds (two highly expressed genes): ordered on the host, like the 1D path
            self._order_on_host(int(q), float(r1a[q]), float(r1b[q]), float(r0[q]), int(pair_slot[q]), tile_ptr, ops)
        for lst, is_big in ((small, 0), (big, 1)):
            if len(lst):
                d_lst = dev(lst)
                _lib.call("mm_bins_order2d", P(self.tab), P(self.d_tab_ptr), P(self.d_xi), P(self.d_xj), P(self.d_K), P(d_lst), len(lst),
                          is_big, ng, self.n_bins, P(d_sf), P(d_ra), P(d_rb), P(d_r0), P(d_pair_slot), P(d_tile_ptr), P(d_nc),
                          *[P(o) for o in ops], P(status), s)
        nobs = np.zeros(n_tiles * 64)
        nobs[slot_of] = self.blocks.grp_ncells[order % ng]
        omq = np.zeros(n_tiles * 64)
        omq[slot_of] = 1.0 - self.grp_q[order % ng]
        self.yc = torch.full((max(1, self.n_q), ld), float("nan"), dtype=torch.float64, device="cuda")
        self.yc[: self.n_q, 0] = dev(np.asarray(true_corr, dtype=np.float64))
        d_slot_K, d_nobs, d_omq, d_slot_pair = dev(slot_K), dev(nobs), dev(omq), dev(slot_pair)
        if n_tiles and use_rec:
            d_slot_rec = dev(slot_rec)
            _lib.call("mm_boot2d_replay_rec", P(ops[0]), P(d_slot_rec), n_tiles, P(d_slot_K), P(d_nobs), P(d_omq), P(d_slot_pair),
                      pcg64_state(pcg_seed), B, ld, P(self.yc), s)
        elif n_tiles:
            _lib.call("mm_boot2d_replay", *[P(o) for o in ops], P(d_tile_ptr), n_tiles, P(d_slot_K), P(d_nobs), P(d_omq), P(d_slot_pair),
                      pcg64_state(pcg_seed), B, ld, P(self.yc), s)
        st = int(status.item())
        if st & 6:
            raise RuntimeError(f"mm_bins_order2d inconsistency (status {st})")
        if st & 8:
            raise NotImplementedError("two bins of one pair collided in the replay hash (np.unique would merge them)")
        self.active = active
        self.pair_slot = pair_slot

    def valid_cols(self, good):
        """hypothesis_test.py:372-373 on the device (see Bootstrap1D.valid_cols)."""
        return _valid_cols(self.yc, self.yc, self.ld, self.B, self.ng, good, self.n_q // self.ng)

    def contract_resampled(self, test_pair, tt, good, pair_mask, M, Nc, rep=None, bcol=None, seed=0, col_map=None, n_valid=None):
        """resample_rep=True on the correlation rows (hypothesis_test.py:393-404).  Same kernels and conventions as
        Bootstrap1D.contract_resampled."""
        return _contract_resampled(self.yc, self.ld, self.B, self.ng, self.n_q // self.ng, test_pair, tt, good, pair_mask, M, Nc, rep, bcol,
                                   seed, col_map, n_valid)

    def contract(self, test_pair, W, good):
        """K9/K10 on the correlation rows: tests (sorted pair index, weight row)."""
        torch = _torch()
        n_tests = len(test_pair)
        coef = empty((max(1, n_tests), self.ld), torch.float64)
        stats = empty((max(1, n_tests), 8), torch.float64)
        d_tp, d_W, d_good = dev(np.asarray(test_pair, dtype=np.int32)), dev(np.asarray(W, dtype=np.float64)), dev(np.asarray(good, dtype=np.uint8))
        _lib.call("mm_contract_stats", P(self.yc), P(self.yc), self.ld, self.B, self.ng, P(d_tp), P(d_W), P(d_good), n_tests, 0, P(coef),
                  P(stats), _stream())
        return coef, host(stats)[:n_tests]
