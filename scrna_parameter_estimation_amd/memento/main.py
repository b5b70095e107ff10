"""Public API: the same names, arguments and ``adata.uns['memento']`` schema as the reference's
``memento/main.py`` -- with the hot path running in hand-written HIP kernels (no CPU fallback).

Reference signatures mirrored (paths relative to /root/reference/):
  setup_memento        memento/main.py:26-34      create_groups      :94-99
  compute_1d_moments   :171-176                   ht_1d_moments      :341-350
  compute_2d_moments   :293                       ht_2d_moments      :418-427      get_corr_matrix :277
  get_groups           :156-168                   get_1d_moments / get_1d_ht_result  :523, :635
Extra keyword-only knobs (do not disturb the reference's positional order):
  ht_1d_moments(..., rng='replay', strict=False, fill_seed=0)
  compute_1d_moments(..., subset_var=True)

Host code here is argument handling, the uns schema, O(G) post-processing (polyfit, filters, design
matrix, p-values); every O(nnz) or O(G x groups x B) step is a kernel launch via ``engine``.
"""

import itertools

import numpy as np
import pandas as pd
import scipy.stats as stats
from scipy.sparse import csr_matrix

from .. import engine
from . import asl as _asl
from . import design as _design


# ----------------------------------------------------------------------------------------------
# state kept next to the AnnData (device handles are not serialisable: prepare_to_save drops them)
# ----------------------------------------------------------------------------------------------


class _HipState:
    """Device-resident data of one AnnData: CSR, count blocks, per-(group, gene) integer sums."""

    def __init__(self):
        self.csr = None
        self.blocks = None
        self.sumx = None
        self.maxx = None
        self.gene_idx = None  # original column index of every currently kept gene

    def __deepcopy__(self, memo):  # shared, read-only after construction
        return self


class _GroupCellsView:
    """Stands in for the reference's per-group CSC copy (main.py:128): only ``shape`` is ever read
    by the API (main.py:364, :534); the counts themselves live in the device count blocks."""

    def __init__(self, n_cells, n_genes):
        self.shape = (int(n_cells), int(n_genes))

    def __repr__(self):
        return f"<device count-block view {self.shape[0]} cells x {self.shape[1]} genes>"


def _mv_fit(mean, var):
    """np.polyfit(log m, log v, 2) over m>0 & v>0  (estimator.py:84-93)."""
    ok = (mean > 0) & (var > 0)
    return np.polyfit(np.log(mean[ok]), np.log(var[ok]), 2)


def _res_var(mean, var, fit):
    """estimator.py:103-111."""
    ok = (mean > 0) & (var > 0)
    out = np.full(mean.shape, np.nan)
    with np.errstate(invalid="ignore"):
        out[ok] = np.exp(np.log(var[ok]) - np.poly1d(fit)(np.log(mean[ok])))
    return out


def _moments_from_sums(S, n_obs, q):
    """mean = S1/n ; var = S2/n - (1-q) S3/n - mean^2   (estimator.py:179-183)."""
    mean = S[0] / n_obs
    var = S[1] / n_obs - (1 - q) * S[2] / n_obs - mean ** 2
    return mean, var


def _plain_means(blocks, sumx, n_cells, thresh, kind):
    """Plain per-group mean count of every gene, as the reference's scipy calls round it, for the `mean < thresh` gene filters
    (main.py:67 on the CSR of all cells, :201-203 on a group's CSC copy).

    The device returns the exact integer sum of the counts, so mean = sumx / n is exact to one rounding -- but scipy computes
    sum_c fl(x_c * (1/n)) (sequentially in cell order for a CSR, first + pairwise(rest) for a CSC), which lands a few ulp away.
    That only matters when sumx / n IS the threshold (e.g. 189 counts in 2,700 cells = 0.07): for such genes -- and only for
    them -- the gene's counts are pulled from the count blocks in cell order and summed the way scipy does, so the filter
    decides exactly like the reference.  Returns float64 [n_groups][G]."""
    n_cells = np.asarray(n_cells, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        mean = sumx.astype(np.float64) / n_cells[:, None]
    near = np.abs(mean - thresh) <= 1e-9 * abs(thresh)
    if not near.any():
        return mean
    genes = np.flatnonzero(near.any(axis=0))
    cols = engine.GeneColumns(blocks, genes)
    data = engine.host(cols.cols, np.uint32)
    ptr = engine.host(cols.col_ptr)
    for gi, m_ in zip(*np.nonzero(near[:, genes])):
        parts = []
        for b in range(int(blocks.grp_blk0[gi]), int(blocks.grp_blk0[gi + 1])):
            e = data[int(ptr[b, m_]):int(ptr[b, m_ + 1])]
            e = e[np.argsort(e & (engine.BLOCK_CELLS - 1), kind="stable")]        # cell order inside the block
            parts.append((e >> 13).astype(np.float64))
        v = np.concatenate(parts) if parts else np.zeros(0)
        v = np.ascontiguousarray(v * (1.0 / n_cells[gi]))
        if len(v) == 0:
            val = 0.0
        elif kind == 'csr':
            val = float(np.cumsum(v)[-1])                    # csc_matvec of the transpose: sequential in cell order
        else:
            val = float(np.add.reduceat(v, [0])[0])          # _minor_reduce: first + pairwise(rest)
        mean[gi, genes[m_]] = val
    return mean


# ----------------------------------------------------------------------------------------------
# setup_memento / create_groups
# ----------------------------------------------------------------------------------------------


def setup_memento(adata, q_column, inplace=True, filter_mean_thresh=0.07, trim_percent=0.1, shrinkage=0.5, num_bins=30,
                  estimator_type='hyper_relative', *, device_csr=None, comm=None, shard=False, size_factor=None):
    """Compute size factors and the all-cell moments (reference: memento/main.py:26-91).

    Extensions: ``device_csr`` -- an ``engine.DeviceCSR`` already resident in HBM (``adata.X`` is then only
    used for its shape); ``comm`` -- a ``dist.Comm`` when the GENES are sharded over ranks (every rank holds
    all cells x its gene shard): per-cell totals are all-reduced so every rank gets the global size factors.
    ``shard=True`` (with ``comm``): ``adata`` holds ALL genes on every rank; this rank's genes are cut out of the resident CSR
    on the device -- no host-side ``X[:, genes]`` -- and the later calls work on that shard (``adata`` itself is left whole;
    ``uns['memento']['gene_list']`` etc. name the shard's genes).  ``shard=True`` / ``'balanced'``: a cost-balanced gene SET
    (``dist.shard_genes_balanced`` over the predicted chain cost of every gene, from the per-gene totals of the resident matrix:
    each rank gets the same mix of long and short bootstrap chains); ``'contiguous'``: the range ``dist.shard_genes`` gives.
    ``shard=<array>``: ``adata`` already holds only this rank's genes, at these positions of the unsharded gene order.
    ``size_factor`` -- precomputed size factors (N values, e.g. those of the whole matrix when ``adata`` is a gene subset):
    the estimation of estimator.py:49-81 is skipped and they are used as they are."""
    if not inplace:
        adata = adata.copy()
    assert adata.obs[q_column].max() < 1
    assert type(adata.X) == csr_matrix, 'please make sure that adata.X is a scipy CSR matrix'
    if estimator_type not in ('hyper_relative', 'mean_only'):
        raise NotImplementedError("the HIP path implements estimator_type 'hyper_relative' and 'mean_only'")
    m = adata.uns['memento'] = {}
    m['q_column'] = q_column
    m['all_q'] = adata.obs[q_column].values.mean()
    m['estimator_type'] = estimator_type
    m['filter_mean_thresh'] = filter_mean_thresh
    m['num_bins'] = num_bins

    st = m['_hip'] = _HipState()
    if device_csr is None:
        device_csr = getattr(adata, 'device_csr', None)       # h5ad.read_h5ad(to_device=True) attaches the resident matrix
    st.csr = device_csr if device_csr is not None else engine.DeviceCSR(adata.X)
    st.comm = comm
    N, G = adata.shape
    assert tuple(st.csr.shape) == (N, G)
    names0 = np.asarray(adata.var.index)
    st.shard = None
    if isinstance(shard, (np.ndarray, list, tuple)):
        # the caller pre-sliced X: these are the positions of adata's genes in the unsharded gene order (needed to put the
        # gathered results, and the global np.random stream, back in that order)
        st.shard = np.asarray(shard, dtype=np.int64)
        assert st.shard.shape == (G,)
    elif shard:
        if comm is None:
            raise ValueError("shard=True needs comm")
        from ..dist import gene_cost, shard_genes, shard_genes_balanced
        if shard == 'contiguous':
            lo, hi = shard_genes(G, comm.rank, comm.world)
            mine = np.arange(lo, hi)
            st.csr = st.csr.colsplit(lo, hi)         # device-side column split; the full CSR is released
        else:
            mine = shard_genes_balanced(gene_cost(st.csr.colsum() / N, filter_mean_thresh), comm.rank, comm.world)
            st.csr = st.csr.colselect(mine)
        st.shard = mine                              # this rank's genes: positions in the unsharded gene order (ascending)
        st.var_names = names0 = names0[mine]
        G = len(mine)
    st.gene_idx = np.arange(G)
    if size_factor is not None:
        size_factor = np.asarray(size_factor, dtype=np.float64)
        assert size_factor.shape == (N,)
        adata.obs['memento_size_factor'] = size_factor
        m['least_variable_genes'] = []
        blocks_all = engine.CountBlocks(st.csr, np.zeros(N, dtype=np.int32), 1)
        S, _, _ = blocks_all.moments(1.0 / size_factor)
        m['all_1d_moments'] = list(_moments_from_sums(S[:, 0], N, m['all_q']))
        if estimator_type == 'mean_only':
            m['all_1d_moments'] = [m['all_1d_moments'][0] + 1, np.ones(G) * 10]
        return adata if not inplace else None
    naive = st.csr.rowsum()                                                   # estimator.py:64-69
    if comm is not None:
        naive = comm.allreduce_sum(naive)
    blocks_all = engine.CountBlocks(st.csr, np.zeros(N, dtype=np.int32), 1)
    with np.errstate(divide="ignore"):
        S, sumx, _ = blocks_all.moments(1.0 / naive)
    all_m, all_v = _moments_from_sums(S[:, 0], N, m['all_q'])                 # main.py:62-66
    all_m = all_m.copy()
    all_m[_plain_means(blocks_all, sumx, [N], filter_mean_thresh, 'csr')[0] < filter_mean_thresh] = 0   # main.py:67
    if comm is None:
        all_rv = _res_var(all_m, all_v, _mv_fit(all_m, all_v))                # main.py:68
        rv_ulim = np.quantile(all_rv[np.isfinite(all_rv)], trim_percent)      # main.py:71
    else:                                   # the fit and the quantile are over ALL genes: gather the shards
        gm, gv = comm.allgather_concat(all_m), comm.allgather_concat(all_v)
        fit0 = _mv_fit(gm, gv)
        grv = _res_var(gm, gv, fit0)
        rv_ulim = np.quantile(grv[np.isfinite(grv)], trim_percent)
        all_rv = _res_var(all_m, all_v, fit0)
    all_rv[~np.isfinite(all_rv)] = np.inf
    mask = all_rv < rv_ulim                                                   # main.py:73
    m['least_variable_genes'] = names0[mask].tolist()
    nrc = st.csr.rowsum(mask)                                                 # estimator.py:73
    if comm is not None:
        nrc = comm.allreduce_sum(nrc)
    nrc = nrc + np.quantile(nrc, shrinkage)                                   # estimator.py:74
    size_factor = nrc / nrc.mean()                                            # estimator.py:75-76
    adata.obs['memento_size_factor'] = size_factor
    S, _, _ = blocks_all.moments(1.0 / size_factor)                           # main.py:86-90
    m['all_1d_moments'] = list(_moments_from_sums(S[:, 0], N, m['all_q']))
    if estimator_type == 'mean_only':                                         # estimator.py:188-204
        m['all_1d_moments'] = [m['all_1d_moments'][0] + 1, np.ones(G) * 10]
    if not inplace:
        return adata


def create_groups(adata, label_columns, label_delimiter='^', inplace=True):
    """Discrete groups from obs columns; builds the group-ordered device count blocks
    (reference: memento/main.py:94-135, util.py:8-13)."""
    if not inplace:
        adata = adata.copy()
    m = adata.uns['memento']
    lab = 'sg' + label_delimiter
    for idx, col_name in enumerate(label_columns):
        lab = lab + adata.obs[col_name].astype(str)
        if idx != len(label_columns) - 1:
            lab = lab + label_delimiter
    adata.obs['memento_group'] = lab
    m['label_columns'] = label_columns
    m['label_delimiter'] = label_delimiter
    codes, uniques = pd.factorize(adata.obs['memento_group'].values)          # first-appearance order == drop_duplicates
    m['groups'] = list(uniques)
    m['q'] = adata.obs[m['q_column']].values
    st = m['_hip']
    st.group_id = codes.astype(np.int32)
    st.blocks = engine.CountBlocks(st.csr, st.group_id, len(uniques))
    G = adata.shape[1]
    m['group_cells'] = {g: _GroupCellsView(st.blocks.grp_ncells[i], G) for i, g in enumerate(m['groups'])}
    qsum = np.bincount(codes, weights=m['q'], minlength=len(uniques))
    m['group_q'] = {g: qsum[i] / st.blocks.grp_ncells[i] for i, g in enumerate(m['groups'])}
    for k in ('size_factor', 'approx_size_factor', 'all_approx_size_factor'):
        m.pop(k, None)
    if not inplace:
        return adata


def _bin_size_factor(adata):
    """30 equal-width bins over all cells' size factors; cell -> bin mean; the max keeps its own value
    (reference: memento/main.py:138-153).  Also records the integer bin id per cell for the device."""
    m = adata.uns['memento']
    size_factor = adata.obs['memento_size_factor'].values
    means, _, idx = stats.binned_statistic(size_factor, size_factor, bins=m['num_bins'], statistic='mean')
    idx = np.clip(idx, a_min=1, a_max=means.shape[0]) - 1
    max_sf = size_factor.max()
    is_max = size_factor == max_sf
    table = np.concatenate([means, [max_sf]])            # one extra "bin" for the max cell(s)
    bin_id = idx.astype(np.int64)
    bin_id[is_max] = means.shape[0]
    approx_sf = table[bin_id]
    st = m['_hip']
    st.sf_bin = bin_id.astype(np.uint8) if table.shape[0] <= 256 else None
    st.sf_table = np.nan_to_num(table, nan=1.0)          # empty bins are never referenced
    m['all_approx_size_factor'] = approx_sf
    # per-group views in the cells' original order: the count blocks' cell order is the stable sort by group (engine.plan_blocks)
    order, ends = st.blocks.cell_order, np.cumsum(st.blocks.grp_ncells)
    a_sorted, s_sorted = approx_sf[order], size_factor[order]
    m['approx_size_factor'] = {g: a_sorted[ends[i] - st.blocks.grp_ncells[i]:ends[i]] for i, g in enumerate(m['groups'])}
    m['size_factor'] = {g: s_sorted[ends[i] - st.blocks.grp_ncells[i]:ends[i]] for i, g in enumerate(m['groups'])}


def get_groups(adata):
    """DataFrame of group label components, rows in uns order (reference: memento/main.py:156-168)."""
    m = adata.uns['memento']
    rows = [g.split(m['label_delimiter'])[1:] for g in m['groups']]
    df = pd.DataFrame(rows, index=m['groups'], columns=m['label_columns'])
    for col in df.columns:
        try:
            df[col] = pd.to_numeric(df[col])
        except (ValueError, TypeError):
            pass
    return df


# ----------------------------------------------------------------------------------------------
# compute_1d_moments
# ----------------------------------------------------------------------------------------------


def compute_1d_moments(adata, inplace=True, min_perc_group=0.7, filter_genes=True, gene_list=None, subset_var=True):
    """Mean, variance and residual variance per group (reference: memento/main.py:171-274).

    ``subset_var=False`` skips the host-side column subset of ``adata`` itself (an O(nnz) scipy copy the
    reference performs at main.py:229); the device blocks and uns['memento']['gene_list'] are what the
    later calls use."""
    assert 'memento' in adata.uns
    if not inplace:
        adata = adata.copy()
    m = adata.uns['memento']
    st = m['_hip']
    if 'size_factor' not in m.keys():
        _bin_size_factor(adata)
    groups = m['groups']
    ng = len(groups)
    Nc = st.blocks.grp_ncells.astype(np.float64)
    gq = np.array([m['group_q'][g] for g in groups])
    S, sumx, maxx = st.blocks.moments(1.0 / adata.obs['memento_size_factor'].values)          # K1+K2
    cur = st.gene_idx                                    # columns of the device blocks that adata currently holds
    names_cur = _var_names(adata)
    if getattr(st, 'shard', None) is not None and adata.shape[1] != st.csr.shape[1]:
        subset_var = False                               # adata holds all genes of all ranks; only the shard's names move
    mean = S[0][:, cur] / Nc[:, None]
    var = S[1][:, cur] / Nc[:, None] - (1 - gq)[:, None] * S[2][:, cur] / Nc[:, None] - mean ** 2
    if m['estimator_type'] == 'mean_only':                                                     # estimator.py:188-204
        mean, var = mean + 1, np.ones(mean.shape) * 10
    st.sumx, st.maxx, st.S = sumx, maxx, S
    obs_mean = _plain_means(st.blocks, sumx, Nc, m['filter_mean_thresh'], 'csc')[:, cur]      # main.py:201
    gene_filter = (obs_mean > m['filter_mean_thresh']) & (var > 0)                             # main.py:202-203
    gene_rv_filter = maxx[:, cur] >= 2                                                         # main.py:206-207
    m['gene_filter'] = {g: gene_filter[i] for i, g in enumerate(groups)}
    overall = gene_filter.mean(axis=0) > min_perc_group                                        # main.py:210-212
    m['overall_gene_filter'] = overall
    m['gene_list'] = names_cur[overall].tolist()
    if filter_genes:                                                                           # main.py:219-229
        mean, var, gene_rv_filter = mean[:, overall], var[:, overall], gene_rv_filter[:, overall]
        st.gene_idx = cur[overall]
        for g in groups:
            m['group_cells'][g] = _GroupCellsView(m['group_cells'][g].shape[0], int(overall.sum()))
        if subset_var:
            adata._inplace_subset_var(overall)
        else:
            st.var_names = names_cur[overall]
    m['gene_rv_filter'] = {g: gene_rv_filter[i] for i, g in enumerate(groups)}
    fm = np.concatenate([mean[i][gene_rv_filter[i]] for i in range(ng)])
    fv = np.concatenate([var[i][gene_rv_filter[i]] for i in range(ng)])
    comm = getattr(st, 'comm', None)
    if comm is not None:                    # gene-sharded: the pooled fit sees every rank's genes
        fm, fv = comm.allgather_concat(fm), comm.allgather_concat(fv)
    fit = _mv_fit(fm, fv)                                                                      # main.py:232-245
    m['mv_regressor'] = {'all': fit}
    for g in groups:
        m['mv_regressor'][g] = fit
    m['1d_moments'] = {g: [mean[i], var[i], _res_var(mean[i], var[i], fit)] for i, g in enumerate(groups)}  # main.py:248-255
    if gene_list is not None:                                                                  # main.py:258-271
        assert type(gene_list) == list
        names = _var_names(adata)
        given = np.isin(names, gene_list)
        for g in groups:
            m['1d_moments'][g] = [a[given] for a in m['1d_moments'][g]]
            m['group_cells'][g] = _GroupCellsView(m['group_cells'][g].shape[0], int(given.sum()))
        st.gene_idx = st.gene_idx[given]
        if subset_var:
            adata._inplace_subset_var(given)
        else:
            st.var_names = names[given]
    if not inplace:
        return adata


def _var_names(adata):
    st = adata.uns['memento']['_hip']
    names = getattr(st, 'var_names', None)
    if names is not None and len(names) == len(st.gene_idx):
        return np.asarray(names)
    return np.asarray(adata.var.index)


# ----------------------------------------------------------------------------------------------
# ht_1d_moments
# ----------------------------------------------------------------------------------------------


def _pair_skip(true_mean, true_rv):
    """hypothesis_test.py:167-171, vectorised over [n_groups][G] -> [pair] (gene-major)."""
    with np.errstate(invalid="ignore"):
        skip = np.isnan(true_mean) | np.isnan(true_rv) | (true_mean == 0) | (true_rv < 0)
    return skip.T.reshape(-1)


def _host_fill(row):
    """hypothesis_test._fill on an already-logged row: NaN = invalid; draws from the global np.random
    stream exactly like np.random.choice(val[~cond], num_invalid) (hypothesis_test.py:23-33)."""
    bad = np.isnan(row)
    nbad = int(bad.sum())
    if nbad == row.shape[0]:
        return None
    row = row.copy()
    row[bad] = np.random.choice(row[~bad], nbad)
    return row


def ht_1d_moments(adata, covariate, treatment, treatment_for_gene=None, inplace=True, num_boot=10000, verbose=1, num_cpus=1,
                  rng='replay', strict=False, fill_seed=0, max_rows=None, **kwargs):
    """Bootstrap hypothesis test of mean / residual-variance differences (reference: memento/main.py:341-415).

    ``rng='replay'``: the multinomial resampling replays numpy's ``Generator(PCG64(5))`` stream draw for
    draw (bootstrap.py:102-103) and the bin order uses the two uniforms the reference takes from the global
    ``np.random`` state per (gene, group) (bootstrap.py:62, :65).
    ``rng='fast'``: same samplers and arithmetic, but every (pair, replicate) has its own counter-derived PCG64
    stream and replicates run in parallel lanes -- statistically equivalent, much faster, not draw-identical.
    ``strict=True`` additionally replays the reference's ``_fill`` draws from the global stream in gene order
    (exactly reproducible against the reference at ``num_cpus=1``; sequential, meant for validation);
    with ``strict=False`` invalid replicates are re-filled on the device with a counter-based RNG.
    kwargs: ``resampling`` (required, as in the reference), ``approx``, ``resample_rep``.
    """
    if 'resampling' not in kwargs:
        raise TypeError("_compute_asl() missing 1 required positional argument: 'resampling'")
    resampling = kwargs['resampling']       # 'bootstrap' centres the null on the observed value, anything else does not (:66-70)
    resample_rep = bool(kwargs.get('resample_rep', False))
    if rng not in ('replay', 'fast'):
        raise ValueError("rng must be 'replay' or 'fast'")
    if strict and rng != 'replay':
        raise ValueError("strict=True needs rng='replay'")
    approx = bool(kwargs.get('approx', False))
    if not inplace:
        adata = adata.copy()
    m = adata.uns['memento']
    st = m['_hip']
    mean_only = m['estimator_type'] == 'mean_only'
    groups = m['groups']
    ng = len(groups)
    names = _var_names(adata)
    G = len(st.gene_idx)
    Nc_list = np.array([m['group_cells'][g].shape[0] for g in groups], dtype=np.float64)
    cov = np.asarray(covariate.values, dtype=np.float64)
    trt_all = np.asarray(treatment.values, dtype=np.float64)
    trt_cols = list(treatment.columns)
    gq = np.array([m['group_q'][g] for g in groups])
    true_mean = np.stack([m['1d_moments'][g][0] for g in groups])
    true_rv = np.stack([m['1d_moments'][g][2] for g in groups])
    fit = m['mv_regressor'][groups[0]]
    if st.sf_bin is None:
        raise NotImplementedError("more than 255 size-factor bins")

    def run_range(g0, g1):
        """One chunk of genes [g0, g1): K5 histograms -> bootstrap -> contraction -> p-values."""
        G = g1 - g0
        bs = engine.Bootstrap1D(st.blocks, st.gene_idx[g0:g1], st.maxx, st.sf_bin, st.sf_table, gq, num_boot)   # K5
        skip = _pair_skip(true_mean[:, g0:g1], true_rv[:, g0:g1])
        with np.errstate(invalid="ignore", divide="ignore"):
            tm_log = np.where(skip, np.nan, np.log(true_mean[:, g0:g1].T.reshape(-1)))
            tv_log = np.where(skip, np.nan, np.log(true_rv[:, g0:g1].T.reshape(-1)))
        bs.alloc_outputs(tm_log, tv_log)
        n_pairs = bs.n_pairs
        r1, r0 = np.zeros(n_pairs), np.zeros(n_pairs)
        live = np.flatnonzero(~skip)

        pair_gene = np.arange(n_pairs) // ng
        known_bad = np.zeros(n_pairs, dtype=bool)       # pairs whose fill found no valid replicate (strict mode bookkeeping)
        rep_assign = bcol_assign = None
        if resample_rep and strict:
            rep_assign = np.zeros((G, ng, num_boot), dtype=np.int16)
            bcol_assign = np.zeros((G, ng, num_boot), dtype=np.int32)

        def gene_uses_resampling(gi, n_good):
            if not resample_rep or n_good == 0:
                return False
            cols = None if treatment_for_gene is None else [trt_cols.index(c) for c in treatment_for_gene[names[g0 + gi]]]
            t = trt_all if cols is None else trt_all[:, cols]
            gmask = ((~skip) & (bs.K >= 2) & ~known_bad)[gi * ng:(gi + 1) * ng]
            return not (t[gmask] == 1).mean() == 1                                                   # hypothesis_test.py:262

        nb_eff = {}      # genes whose replicate columns are not all finite: columns left after hypothesis_test.py:249-251, minus one

        def draw_assignments(gi):
            n = int(((~skip) & (bs.K >= 2) & ~known_bad)[gi * ng:(gi + 1) * ng].sum())
            if gene_uses_resampling(gi, n):
                nb = nb_eff.get(gi, num_boot)                                  # hypothesis_test.py:253
                if nb < 1:
                    return
                ra = np.random.choice(n, size=(n, nb))                         # hypothesis_test.py:275-278
                ra[:, 0] = np.arange(n)
                ba = np.random.choice(nb, (n, nb)) + 1
                ba[:, 0] = 0
                rep_assign[gi, :n, :nb], bcol_assign[gi, :n, :nb] = ra, ba

        def draw_stream(first, stop_pair=None, pending=None):
            """Consume the global np.random stream exactly as the reference does from pair ``first`` on: per gene the two
            hash uniforms of every live group (bootstrap.py:62,65) and -- with resample_rep -- the two np.random.choice
            draws of _regress_1d (hypothesis_test.py:275-278) after the gene's last group.  ``pending``: a gene whose
            groups are all done but whose choice draws are still due; ``stop_pair``: stop right after that pair's hash."""
            if not (resample_rep and strict):
                idx = live[live >= first] if stop_pair is None else live[(live >= first) & (live <= stop_pair)]
                u = np.random.random(2 * len(idx))      # same stream positions as random(1) then random() per pair
                r1[idx], r0[idx] = u[0::2], u[1::2]
                return
            if pending is not None:
                draw_assignments(pending)
            for gi in range(int(first // ng), G):
                lo_p = max(first, gi * ng)
                hi_p = (gi + 1) * ng - 1 if stop_pair is None else min((gi + 1) * ng - 1, stop_pair)
                idx = live[(live >= lo_p) & (live <= hi_p)]
                u = np.random.random(2 * len(idx))
                r1[idx], r0[idx] = u[0::2], u[1::2]
                if stop_pair is not None and stop_pair < (gi + 1) * ng:
                    return
                draw_assignments(gi)

        if not strict:
            if shard_stream is not None:      # gene-sharded: this rank's slice of the ONE global stream (see below)
                r1[:], r0[:] = shard_stream[0][g0 * ng:g1 * ng], shard_stream[1][g0 * ng:g1 * ng]
            else:
                draw_stream(0)
            keys = (fill_pos[g0:g1, None] * ng + np.arange(ng)[None, :]).reshape(-1)      # refill streams keyed by (gene, group)
            n_inv = bs.run(skip, r1, r0, fit, fill_mode=0, fill_seed=fill_seed, fast=(rng == 'fast'), mean_only=mean_only,
                           fill_keys=keys)                                                # K6-K8
            bad_fill = (n_inv < 0).any(axis=1)
            # how much of the result depends on the device refill (strict=True replays the reference's own _fill draws instead):
            # chains / genes with at least one refilled replicate -- all others are bit-identical to the strict path
            refilled = (n_inv > 0).any(axis=1) & ~skip
            rs = st.refill_stats
            rs['chains'] += int(((~skip) & (bs.K >= 2)).sum())
            rs['chains_refilled'] += int(refilled.sum())
            rs['genes'] += G
            rs['genes_refilled'] += int(refilled.reshape(G, ng).any(axis=1).sum())
            rs['gene_refilled'][g0:g1] = refilled.reshape(G, ng).any(axis=1)
        else:
            def strict_pass():
                """One sequential replay of the reference's global-stream consumption over all genes (speculate, then roll back to
                the first pair whose _fill draws -- or, under resample_rep, shrinking num_rep -- shift the stream)."""
                n_inv_all = np.zeros((n_pairs, 2), dtype=np.int32)
                first, pending = 0, None
                while first < n_pairs:
                    saved = np.random.get_state()
                    draw_stream(first, pending=pending)
                    after = np.random.get_state()
                    n_inv = bs.run(skip, r1, r0, fit, fill_mode=1, first_pair=first, mean_only=mean_only)
                    n_inv_all[first:] = n_inv
                    event = (n_inv > 0).any(axis=1)
                    if resample_rep:
                        event |= (n_inv < 0).any(axis=1) & ~known_bad[first:]      # a group without valid replicates shrinks num_rep
                    needs = np.flatnonzero((~skip[first:]) & event) + first
                    if len(needs) == 0:
                        np.random.set_state(after)
                        pending = None
                        break
                    p = int(needs[0])
                    np.random.set_state(saved)
                    draw_stream(first, stop_pair=p, pending=pending)                # everything the reference drew up to pair p's hash
                    for t, col in ((bs.ym, 0), (bs.yv, 1)):
                        if n_inv_all[p, col] > 0:
                            row = engine.host(t[p, 1:])
                            t[p, 1:] = engine.dev(_host_fill(row))
                            n_inv_all[p, col] = 0
                    if (n_inv_all[p] < 0).any():
                        known_bad[p] = True
                    pending = p // ng if (resample_rep and p % ng == ng - 1) else None
                    first = p + 1
                if pending is not None:
                    draw_assignments(pending)
                return (n_inv_all < 0).any(axis=1)

            stream0 = np.random.get_state()
            for _attempt in range(3):
                bad_fill = strict_pass()
                if not resample_rep:
                    break
                # The reference draws a gene's assignments for the replicate columns that SURVIVE hypothesis_test.py:249-251,
                # which is known only after its bootstrap: when a resampled gene lost columns, replay once more with that count.
                good_now = ((~skip) & (bs.K >= 2) & ~bad_fill).reshape(G, ng)
                _, nv = bs.valid_cols(good_now)
                redo = False
                for gi in np.flatnonzero(nv != num_boot + 1):
                    gi = int(gi)
                    if gene_uses_resampling(gi, int(good_now[gi].sum())) and nb_eff.get(gi, num_boot) != int(nv[gi]) - 1:
                        nb_eff[gi] = int(nv[gi]) - 1
                        redo = True
                if not redo:
                    break
                np.random.set_state(stream0)
                known_bad[:] = False

        active_all = (~skip) & (bs.K >= 2)                       # bootstrap.py:97-98: a single bin gives NaN replicates
        good = (active_all & ~bad_fill).reshape(G, ng)           # hypothesis_test.py:193-200

        # tests: gene-major x treatment column (main.py:399-404)
        test_gene, test_rows = [], []
        cache = {}
        for gi in range(G):
            if treatment_for_gene is None:
                cols = None
                nt = trt_all.shape[1]
            else:
                cols = [trt_cols.index(c) for c in treatment_for_gene[names[g0 + gi]]]
                nt = len(cols)
            key = (good[gi].tobytes(), None if cols is None else tuple(cols))
            W = cache.get(key)
            if W is None:
                t = trt_all if cols is None else trt_all[:, cols]
                W = _design.weight_rows(cov, t, Nc_list, good[gi])
                if W.shape[0] != nt:
                    W = np.repeat(W[:1], nt, axis=0)
                cache[key] = W
            test_gene.extend([gi] * nt)
            test_rows.append(W)
        n_tests = len(test_gene)
        Wmat = np.concatenate(test_rows, axis=0) if test_rows else np.zeros((0, ng))
        out = {}
        use_rr = resample_rep and n_tests > 0
        if use_rr:
            # residual maker / residualised treatment per valid-group mask; tests whose treatment is all ones keep the
            # weighted-average branch (hypothesis_test.py:262-265) and are not resampled
            masks, gene_mask, tt_rows, rr_test = {}, np.zeros(G, dtype=np.int32), [], np.zeros(n_tests, dtype=bool)
            Ms = []
            ti = 0
            for gi in range(G):
                cols = None if treatment_for_gene is None else [trt_cols.index(c) for c in treatment_for_gene[names[g0 + gi]]]
                t = trt_all if cols is None else trt_all[:, cols]
                key = (good[gi].tobytes(), None if cols is None else tuple(cols))
                if key not in masks:
                    Mg, ttg = _design.residual_parts(cov, t, Nc_list, good[gi])
                    masks[key] = (len(Ms), ttg)
                    Ms.append(Mg)
                gene_mask[gi], ttg = masks[key]
                nt = t.shape[1]
                allones = good[gi].any() and (t[good[gi]] == 1).mean() == 1
                tt_rows.append(ttg)
                rr_test[ti:ti + nt] = not allones
                ti += nt
            tt_mat = np.concatenate(tt_rows, axis=0)
            Mstack = np.stack(Ms)
        col_map = n_valid = None
        if use_rr and rr_test.any():
            col_map, n_valid = bs.valid_cols(good)                  # hypothesis_test.py:249-251
            if (n_valid[good.any(axis=1)] == num_boot + 1).all():
                col_map = n_valid = None                            # nothing dropped (the usual case): identity map
        for which, tag in ((0, 'mean'), (1, 'var')):
            coef, stt = bs.contract(test_gene, Wmat, good, which)                                         # K9+K10
            if use_rr and rr_test.any():
                coef_r, stt_r = bs.contract_resampled(test_gene, tt_mat, good, which, gene_mask, Mstack, Nc_list,
                                                      rep_assign, bcol_assign, seed=fill_seed + 17, col_map=col_map, n_valid=n_valid)
                stt = np.where(rr_test[:, None], stt_r, stt)
                rr_idx = engine.dev(np.flatnonzero(rr_test))
                coef[rr_idx] = coef_r[rr_idx]
            no_group = ~good[np.asarray(test_gene, dtype=np.int64)].any(axis=1) if n_tests else np.zeros(0, bool)
            c0, se = stt[:, 0].copy(), stt[:, 1].copy()
            # the tail fits of the mean tests run in the worker pool while the variance contraction is enqueued
            fin = _asl.asl_from_stats(stt, approx, lambda idx, coef=coef: engine.host(coef[engine.dev(np.asarray(idx, dtype=np.int64))]),
                                      num_cpus, resampling, defer=True)
            c0[no_group], se[no_group] = np.nan, np.nan                                                   # hypothesis_test.py:203-204
            out[tag + '_coef'], out[tag + '_se'], out[tag + '_asl'] = c0, se, (fin, no_group)
        for tag in ('mean', 'var'):
            fin, no_group = out[tag + '_asl']
            p = fin()
            p[no_group] = np.nan
            out[tag + '_asl'] = p
        st.last_bootstrap = bs
        st.last_assignments = (rep_assign, bcol_assign)
        return out

    G_all = len(st.gene_idx)
    comm = getattr(st, 'comm', None)
    sharded = comm is not None and comm.world > 1
    shard_stream = gene_pos = None
    if sharded:
        # The reference scatters every gene's result back from ONE sequential np.random stream (main.py:399-404, bootstrap.py:62-65):
        # two hash uniforms per live (gene, group) chain, gene-major.  A rank must therefore use the uniforms at the positions its
        # genes have in the UNSHARDED order: gather (position, live chains) of every kept gene, draw the whole stream (the same
        # on every rank: the caller seeds all ranks alike) and keep this rank's entries.  N-rank results then equal 1-rank results.
        shard = getattr(st, 'shard', None)
        if shard is not None:
            gene_pos = np.asarray(shard, dtype=np.int64)[st.gene_idx]
        else:                               # the caller pre-sliced X: contiguous blocks in rank order
            sizes = comm.allgather_objects(int(st.csr.shape[1]))
            gene_pos = int(sum(sizes[:comm.rank])) + np.asarray(st.gene_idx, dtype=np.int64)
        if strict:
            # strict=True also replays the reference's _fill draws, which shift the global stream gene after gene: sequential over
            # ALL genes.  With contiguous shards the ranks take turns in rank order and hand the stream state on (the validation
            # mode: exact, no speed-up); interleaved (cost-balanced) shards cannot be replayed that way.
            spans = comm.allgather_objects((int(gene_pos.min()), int(gene_pos.max())) if len(gene_pos) else None)
            spans = [sp_ for sp_ in spans if sp_ is not None]
            if any(a[1] >= b[0] for a, b in zip(spans[:-1], spans[1:])):
                raise NotImplementedError("strict=True over several ranks needs contiguous gene shards in rank order "
                                          "(setup_memento(shard='contiguous') or a pre-sliced range); cost-balanced shards interleave")
        else:
            from ..dist import shard_stream_uniforms
            shard_stream = shard_stream_uniforms(comm, gene_pos, (~_pair_skip(true_mean, true_rv)).reshape(G_all, ng))
    # position of every kept gene in the unsharded, unfiltered gene order: keys the device refill streams, so that the timed mode's
    # results do not depend on gene chunking or on the sharding over GPUs
    fill_pos = (np.asarray(st.shard, dtype=np.int64)[st.gene_idx] if getattr(st, 'shard', None) is not None
                else (gene_pos if gene_pos is not None else np.asarray(st.gene_idx, dtype=np.int64)))
    st.refill_stats = dict(chains=0, chains_refilled=0, genes=0, genes_refilled=0, gene_refilled=np.zeros(G_all, dtype=bool))
    st.last_bootstrap = None                   # the previous call's replicate rows (GBs) go back to the caching allocator BEFORE this call allocates its own
    if max_rows is None:                       # replicate buffers sized to the free HBM (288 GB on MI355X)
        max_rows = engine.auto_max_rows(num_boot + 1, arrays=2)
    chunk = G_all if strict else max(1, int(max_rows) // max(1, ng))   # strict replay is sequential over all genes
    if sharded and strict:
        parts = []
        for turn in range(comm.world):                # rank after rank, the np.random state handed on
            if turn == comm.rank:
                parts = [run_range(g0, min(G_all, g0 + chunk)) for g0 in range(0, G_all, chunk)] if G_all else []
            states = comm.allgather_objects(np.random.get_state() if turn == comm.rank else None)
            np.random.set_state(states[turn])
    else:
        parts = [run_range(g0, min(G_all, g0 + chunk)) for g0 in range(0, G_all, chunk)] if G_all else []
    keys = ('mean_coef', 'mean_se', 'mean_asl', 'var_coef', 'var_se', 'var_asl')
    out = {k: (np.concatenate([p_[k] for p_ in parts]) if parts else np.zeros(0)) for k in keys}
    m['1d_ht'] = {}
    if sharded:
        # gene-sharded run: scatter-back of the reference (main.py:399-412) across ranks -- every rank ends with the flat result
        # vectors of ALL genes in the unsharded run's order, plus the gene names they belong to
        from ..dist import gather_1d_ht
        nt_gene = (np.full(G_all, trt_all.shape[1], dtype=np.int64) if treatment_for_gene is None
                   else np.array([len(treatment_for_gene[n_]) for n_ in names[:G_all]], dtype=np.int64))
        st.local_ht = {k: np.asarray(v).copy() for k, v in out.items()}          # this rank's own tests (bench: per-rank accounting)
        m['1d_ht']['gene_names'], out = gather_1d_ht(comm, names, out, gene_pos=gene_pos, n_tests=nt_gene)
    if treatment_for_gene is not None:
        m['1d_ht']['treatment_for_gene'] = treatment_for_gene
    m['1d_ht']['treatment'] = treatment
    m['1d_ht']['covariate'] = covariate
    for k in ('mean_coef', 'mean_se', 'mean_asl', 'var_coef', 'var_se', 'var_asl'):
        m['1d_ht'][k] = out[k]
    if not inplace:
        return adata


def ht_1d_vs_control(adata, control, num_boot=10000, num_cpus=1, rng='replay', fill_seed=0, max_rows=None, approx=False,
                     resampling='bootstrap'):
    """Perturb-seq style batch test: every group against one shared ``control`` group (a label of
    ``adata.uns['memento']['groups']`` or its index), for all kept genes, in one call.

    The reference handles this design with a Python loop over guides (subset to control + guide, create_groups,
    compute_1d_moments, ht_1d_moments; e.g. analysis/sciplex/sciplex_dv.py:18-59), re-bootstrapping the control group
    for every guide.  Here each (gene, group) is bootstrapped once and the two-group test statistic -- for the groups
    {control, guide} with a binary treatment and an intercept, _regress_1d (hypothesis_test.py:269-291) reduces to
    log-moment(guide) - log-moment(control) -- is taken for every guide from the shared control rows.  The moments,
    gene filters and the mean-variance fit are those of ``compute_1d_moments`` over ALL groups (the per-guide loop refits
    them on each two-group subset), so numbers differ from that loop by the fit, not by the test.

    Returns a DataFrame (gene, group, de_coef, de_se, de_pval, dv_coef, dv_se, dv_pval) and stores the arrays in
    ``uns['memento']['1d_ht_vs_control']``."""
    m = adata.uns['memento']
    st = m['_hip']
    groups = m['groups']
    ng = len(groups)
    ctrl = groups.index(control) if not isinstance(control, (int, np.integer)) else int(control)
    others = [j for j in range(ng) if j != ctrl]
    mean_only = m['estimator_type'] == 'mean_only'
    names = _var_names(adata)
    gq = np.array([m['group_q'][g] for g in groups])
    true_mean = np.stack([m['1d_moments'][g][0] for g in groups])
    true_rv = np.stack([m['1d_moments'][g][2] for g in groups])
    fit = m['mv_regressor'][groups[0]]
    G_all = len(st.gene_idx)
    if max_rows is None:                       # (40 % of the free HBM: BASELINE configs[4] then runs in two chunks, the second reusing the
        max_rows = engine.auto_max_rows(num_boot + 1, arrays=2)   # first one's buffers; ONE 108 GB chunk is 1 s faster in a warm process but 1.6 s
                                                                 # slower in a fresh one -- mapping fresh device memory costs ~30 ms per GB)
    chunk = max(1, int(max_rows) // max(1, ng))
    cols = {k: [] for k in ('mean_coef', 'mean_se', 'mean_asl', 'var_coef', 'var_se', 'var_asl')}
    bs = rows = None
    st.last_bootstrap = None                   # (see ht_1d_moments: free the previous call's replicate rows first)
    for g0 in range(0, G_all, chunk):
        g1 = min(G_all, g0 + chunk)
        G = g1 - g0
        del bs, rows            # release the previous chunk's replicate rows first: the caching allocator hands them back
        bs = engine.Bootstrap1D(st.blocks, st.gene_idx[g0:g1], st.maxx, st.sf_bin, st.sf_table, gq, num_boot)
        skip = _pair_skip(true_mean[:, g0:g1], true_rv[:, g0:g1])
        with np.errstate(invalid="ignore", divide="ignore"):
            tm_log = np.where(skip, np.nan, np.log(true_mean[:, g0:g1].T.reshape(-1)))
            tv_log = np.where(skip, np.nan, np.log(true_rv[:, g0:g1].T.reshape(-1)))
        bs.alloc_outputs(tm_log, tv_log)
        r1, r0 = np.zeros(bs.n_pairs), np.zeros(bs.n_pairs)
        live = np.flatnonzero(~skip)
        u = np.random.random(2 * len(live))
        r1[live], r0[live] = u[0::2], u[1::2]
        n_inv = bs.run(skip, r1, r0, fit, fill_mode=0, fill_seed=fill_seed, fast=(rng == 'fast'), mean_only=mean_only)
        good = ((~skip) & (bs.K >= 2) & ~(n_inv < 0).any(axis=1)).reshape(G, ng)
        test_gene = np.repeat(np.arange(G), len(others))
        test_grp = np.tile(np.asarray(others), G)
        st_m, st_v, rows = bs.contrast(test_gene, test_grp, ctrl, good)
        for tag, stt, which in (('mean', st_m, 0), ('var', st_v, 1)):
            p = _asl.asl_from_stats(stt, approx, lambda idx, w=which: rows(w, idx), num_cpus, resampling)
            cols[tag + '_coef'].append(stt[:, 0])
            cols[tag + '_se'].append(stt[:, 1])
            cols[tag + '_asl'].append(p)
    out = {k: (np.concatenate(v) if v else np.zeros(0)) for k, v in cols.items()}
    st.last_bootstrap, st.last_chunk = bs, ((g0, g1) if G_all else (0, 0))       # diagnostics / tests: the last gene chunk's replicate rows
    m['1d_ht_vs_control'] = dict(out, control=groups[ctrl], groups=[groups[j] for j in others])
    df = pd.DataFrame({'gene': np.repeat(names, len(others)), 'group': np.tile([groups[j] for j in others], G_all)})
    df['de_coef'], df['de_se'], df['de_pval'] = out['mean_coef'], out['mean_se'], out['mean_asl']
    df['dv_coef'], df['dv_se'], df['dv_pval'] = out['var_coef'], out['var_se'], out['var_asl']
    return df


# ----------------------------------------------------------------------------------------------
# 2D: compute_2d_moments / ht_2d_moments / get_corr_matrix
# ----------------------------------------------------------------------------------------------


def _corr_from_cov(cov, var_1, var_2):
    """estimator._corr_from_cov (estimator.py:273-292): variances <= 0 become NaN (in place, as the
    reference does), entries without a finite sqrt(v1 v2) keep the 5.0 sentinel and are then clipped to 1."""
    corr = np.full(cov.shape, 5.0)
    var_1[var_1 <= 0] = np.nan
    var_2[var_2 <= 0] = np.nan
    vp = np.sqrt(var_1 * var_2)
    ok = np.isfinite(vp)
    corr[ok] = cov[ok] / vp[ok]
    corr[corr > 1] = 1
    corr[corr < -1] = -1
    return corr


def compute_2d_moments(adata, gene_pairs, inplace=True):
    """Covariance and correlation of the given gene pairs per group (reference: memento/main.py:293-338,
    estimator.py:207-233).  The only O(nnz) quantity, sum_c x_i x_j / sf^2, comes from mm_pair_cross; the
    means and the i == j correction reuse the 1D sums of compute_1d_moments."""
    if not inplace:
        adata = adata.copy()
    m = adata.uns['memento']
    st = m['_hip']
    if 'size_factor' not in m.keys():
        _bin_size_factor(adata)
    groups = m['groups']
    names = _var_names(adata)
    mapping = dict(zip(names, np.arange(len(names))))
    n_pairs = len(gene_pairs)
    idx1 = np.array([mapping[a] for a, _ in gene_pairs], dtype=int)
    idx2 = np.array([mapping[b] for _, b in gene_pairs], dtype=int)
    m['2d_moments'] = {'gene_pairs': gene_pairs, 'gene_idx_1': idx1, 'gene_idx_2': idx2}
    used = np.unique(np.concatenate([idx1, idx2])) if n_pairs else np.zeros(0, dtype=int)
    st.cols = engine.GeneColumns(st.blocks, st.gene_idx[used])          # columns of the genes in the pair list
    st.cols_local = used
    slot = {int(g): i for i, g in enumerate(used)}
    c1 = np.array([slot[int(i)] for i in idx1], dtype=np.int64)
    c2 = np.array([slot[int(j)] for j in idx2], dtype=np.int64)
    prod = engine.pair_cross(st.cols, c1, c2, 1.0 / adata.obs['memento_size_factor'].values)   # [n_groups][n_pairs]
    Nc = st.blocks.grp_ncells.astype(np.float64)
    same = idx1 == idx2
    g1, g2 = st.gene_idx[idx1], st.gene_idx[idx2]
    for gi, group in enumerate(groups):
        q = m['group_q'][group]
        p = prod[gi] / Nc[gi]
        if same.any():
            p[same] = p[same] - (1 - q) * st.S[2][gi, g1[same]] / Nc[gi]                       # estimator.py:229-230
        cov = p - (st.S[0][gi, g1] / Nc[gi]) * (st.S[0][gi, g2] / Nc[gi])                      # estimator.py:231
        var_1 = m['1d_moments'][group][1][idx1]
        var_2 = m['1d_moments'][group][1][idx2]
        corr = _corr_from_cov(cov, var_1, var_2)
        m['2d_moments'][group] = {'cov': cov, 'corr': corr, 'var_1': var_1, 'var_2': var_2}
    if not inplace:
        return adata


def get_corr_matrix(adata, group):
    """All-by-all correlation matrix of one group (reference: memento/main.py:277-291,
    estimator._hyper_corr_symmetric estimator.py:236-270)."""
    m = adata.uns['memento']
    st = m['_hip']
    gi = m['groups'].index(group)
    G = len(st.gene_idx)
    cols = engine.GeneColumns(st.blocks, st.gene_idx)
    iu, ju = np.triu_indices(G)
    prod = engine.pair_cross(cols, iu, ju, 1.0 / adata.obs['memento_size_factor'].values)[gi]
    n = float(st.blocks.grp_ncells[gi])
    q = m['group_q'][group]
    P = np.zeros((G, G))
    P[iu, ju] = prod / n
    P[ju, iu] = prod / n
    d = np.arange(G)
    P[d, d] -= (1 - q) * st.S[2][gi, st.gene_idx] / n                                          # estimator.py:256
    mu = st.S[0][gi, st.gene_idx] / n
    cov = P - np.outer(mu, mu)
    var = m['1d_moments'][group][1]
    # estimator.py:259-263: the reference NaNs only fancy-index copies; var_prod comes from the untouched variances
    # (negative x negative -> finite product, negative x positive -> NaN, zero -> division by zero -> NaN below)
    with np.errstate(invalid="ignore", divide="ignore"):
        vp = np.sqrt(np.outer(var, var))
        corr = np.full(cov.shape, 5.0)
        ok = np.isfinite(vp)
        corr[ok] = cov[ok] / vp[ok]
    inside = (corr < 1.05) & (corr > -1.05)
    corr[inside] = np.clip(corr[inside], a_min=-1, a_max=1)
    corr[(corr > 1) | (corr < -1)] = np.nan
    return corr


def ht_2d_moments(adata, covariate, treatment, treatment_for_gene=None, inplace=True, num_boot=10000, verbose=3, num_cpus=1,
                  max_rows=None, fill_seed=0, strict=False, **kwargs):
    """Bootstrap hypothesis test of correlation differences (reference: memento/main.py:418-520,
    hypothesis_test._ht_2d :303-364).  Same replay semantics as ht_1d_moments.  ``resample_rep=True``
    (hypothesis_test.py:393-404): with ``strict=True`` the reference's two np.random.choice draws per pair are replayed from
    the global stream in pair order (exact agreement with the reference at num_cpus=1); otherwise the group /
    replicate-column assignments are drawn on the device (seeded by ``fill_seed``): statistically equivalent, not
    draw-identical.  Without resample_rep the 2D path consumes the global stream exactly like the reference either way.
    ``treatment_for_gene`` as the reference BEHAVES (main.py:492): a pair's treatment columns are looked up under
    ``frozenset({name of the pair's first gene})`` -- the key is built from idx_1 twice -- and one number per pair is stored,
    so the list must hold exactly one column (the reference raises ValueError on more); pinned by fixture ``api_tfg2d``."""
    if 'resampling' not in kwargs:
        raise TypeError("_compute_asl() missing 1 required positional argument: 'resampling'")
    resampling = kwargs['resampling']
    resample_rep = bool(kwargs.get('resample_rep', False))
    approx = bool(kwargs.get('approx', False))
    if not inplace:
        adata = adata.copy()
    m = adata.uns['memento']
    st = m['_hip']
    st.last_bootstrap2d = None                 # (free the previous call's replicate rows before this call allocates its own)
    groups = m['groups']
    ng = len(groups)
    Nc_list = np.array([m['group_cells'][g].shape[0] for g in groups], dtype=np.float64)
    cov = np.asarray(covariate.values, dtype=np.float64)
    trt = np.asarray(treatment.values, dtype=np.float64)
    gq = np.array([m['group_q'][g] for g in groups])
    idx1, idx2 = m['2d_moments']['gene_idx_1'], m['2d_moments']['gene_idx_2']
    n_conv = idx1.shape[0]
    # unordered pairs, first appearance wins; self pairs skipped (main.py:467-482)
    first, members = [], {}
    for c in range(n_conv):
        a, b = int(idx1[c]), int(idx2[c])
        if a == b:
            continue
        key = frozenset((a, b))
        if key in members:
            members[key].append(c)
            continue
        members[key] = [c]
        first.append(c)
    first = np.asarray(first, dtype=np.int64)
    P_ = len(first)
    tcol = np.zeros(P_, dtype=np.int64)                   # treatment column of every tested pair (column 0 without treatment_for_gene)
    if treatment_for_gene is not None:
        names2, trt_cols = np.asarray(adata.var.index), list(treatment.columns)
        for k, c in enumerate(first):
            cols = treatment_for_gene[frozenset({names2[int(idx1[c])]})]
            if len(cols) != 1:
                raise ValueError("setting an array element with a sequence.")          # what main.py:507 does with more columns
            tcol[k] = trt_cols.index(cols[0])
    slot = {int(g): i for i, g in enumerate(st.cols_local)}
    c1 = np.array([slot[int(idx1[c])] for c in first], dtype=np.int64)
    c2 = np.array([slot[int(idx2[c])] for c in first], dtype=np.int64)
    true_corr = np.stack([m['2d_moments'][g]['corr'][first] for g in groups], axis=1) if P_ else np.zeros((0, ng))   # [pair][group]
    with np.errstate(invalid="ignore"):
        skip = np.isnan(true_corr) | (np.abs(true_corr) == 1)                                  # hypothesis_test.py:325
    live = ~skip.reshape(-1)
    r1a, r1b, r0 = (np.zeros(P_ * ng) for _ in range(3))
    replay_rr = resample_rep and strict                   # the choice draws interleave with the hash uniforms, pair by pair
    if not replay_rr:
        u = np.random.random(3 * int(live.sum()))        # r = random(2) then r0 = random() per live (pair, group), in order
        r1a[live], r1b[live], r0[live] = u[0::3], u[1::3], u[2::3]
    corr_coef, corr_se, corr_asl = (np.full(n_conv, np.nan) for _ in range(3))
    bs = None
    # pairs are independent: process them in chunks so the replicate rows ([pair x group][B+1] fp64) stay bounded
    if max_rows is None:
        max_rows = min(1 << 19, engine.auto_max_rows(num_boot + 1, arrays=1))   # also bounds the per-pair 2D tables
    chunk = max(1, int(max_rows) // max(1, ng))
    # chunk boundaries: at most ``chunk`` pairs (replicate rows) AND at most a third of the free HBM in histogram tables
    tab_bytes = engine.pair_table_bytes(st.maxx, st.cols.genes, c1, c2, ng, len(st.sf_table)) if P_ else np.zeros(0, dtype=np.int64)
    budget = max(1 << 28, engine._torch().cuda.mem_get_info()[0] // 3)
    bounds, acc = [0], 0
    for k in range(P_):
        if k - bounds[-1] >= chunk or (acc + int(tab_bytes[k]) > budget and k > bounds[-1]):
            bounds.append(k)
            acc = 0
        acc += int(tab_bytes[k])
    bounds.append(P_)
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        if hi <= lo:
            continue
        n_ch = hi - lo
        bs = engine.Bootstrap2D(st.cols, c1[lo:hi], c2[lo:hi], st.maxx, st.sf_bin, st.sf_table, gq, num_boot)
        so = bs.order                                     # device pair order (sorted by left column)
        sl = slice(lo * ng, hi * ng)
        rep_assign = bcol_assign = None
        if replay_rr:
            # global-stream order of the reference for one pair (_ht_2d :319-343, _regress_2d :395-398): three uniforms per
            # live group, then -- unless the treatment of the good groups is all ones -- the two np.random.choice draws
            inv = np.empty(n_ch, dtype=np.int64)
            inv[so] = np.arange(n_ch)
            K_orig = bs.K.reshape(n_ch, ng)[inv]                                  # bins per (pair, group), original pair order
            rep_assign = np.zeros((n_ch, ng, num_boot), dtype=np.int16)          # device pair order
            bcol_assign = np.zeros((n_ch, ng, num_boot), dtype=np.int32)
            for pi in range(n_ch):
                lv = ~skip[lo + pi]
                idx = (lo + pi) * ng + np.flatnonzero(lv)
                uu = np.random.random(3 * len(idx))
                r1a[idx], r1b[idx], r0[idx] = uu[0::3], uu[1::3], uu[2::3]
                gd = lv & (K_orig[pi] >= 1)
                n = int(gd.sum())
                t_pi = trt if treatment_for_gene is None else trt[:, [tcol[lo + pi]]]
                if n and not (t_pi[gd] == 1).mean() == 1:
                    ra = np.random.choice(n, size=(n, num_boot))
                    ra[:, 0] = np.arange(n)
                    ba = np.random.choice(num_boot, (n, num_boot)) + 1
                    ba[:, 0] = 0
                    rep_assign[inv[pi], :n], bcol_assign[inv[pi], :n] = ra, ba

        def to_dev_order(a):
            return a[sl].reshape(n_ch, ng)[so].reshape(-1)

        bs.run(to_dev_order(skip.reshape(-1)), to_dev_order(r1a), to_dev_order(r1b), to_dev_order(r0),
               to_dev_order(np.where(skip, np.nan, true_corr).reshape(-1)))
        good = bs.active.reshape(n_ch, ng)                # device order
        cache, rows = {}, []
        tc = tcol[lo:hi][so]                               # device order
        for k in range(n_ch):
            key = (good[k].tobytes(), int(tc[k]))
            W = cache.get(key)
            if W is None:
                W = cache[key] = _design.weight_rows(cov, trt[:, [tc[k]]], Nc_list, good[k])[:1]
            rows.append(W)
        Wmat = np.concatenate(rows, axis=0) if rows else np.zeros((0, ng))
        coef, stt = bs.contract(np.arange(n_ch), Wmat, good)
        if resample_rep and n_ch:
            # residual maker / residualised treatment per valid-group mask; an all-ones treatment keeps the weighted-average
            # branch (hypothesis_test.py:384-386) and is not resampled
            masks, pair_mask, tt_rows, rr_test, Ms = {}, np.zeros(n_ch, dtype=np.int32), [], np.zeros(n_ch, dtype=bool), []
            for k in range(n_ch):
                key = (good[k].tobytes(), int(tc[k]))
                if key not in masks:
                    t_k = trt if treatment_for_gene is None else trt[:, [tc[k]]]
                    Mg, ttg = _design.residual_parts(cov, t_k, Nc_list, good[k])
                    masks[key] = (len(Ms), ttg[:1], bool(good[k].any() and (t_k[good[k]] == 1).mean() == 1))
                    Ms.append(Mg)
                pair_mask[k], ttg, allones = masks[key]
                tt_rows.append(ttg)
                rr_test[k] = good[k].any() and not allones
            if rr_test.any():
                col_map, n_valid = bs.valid_cols(good)                            # hypothesis_test.py:372-373
                if (n_valid[good.any(axis=1)] == num_boot + 1).all():
                    col_map = n_valid = None
                elif replay_rr:
                    raise NotImplementedError("strict replay of resample_rep with non-finite correlation replicates")
                coef_r, stt_r = bs.contract_resampled(np.arange(n_ch), np.concatenate(tt_rows, axis=0), good, pair_mask, np.stack(Ms),
                                                      Nc_list, rep=rep_assign, bcol=bcol_assign, seed=fill_seed + 17 + lo,
                                                      col_map=col_map, n_valid=n_valid)
                stt = np.where(rr_test[:, None], stt_r, stt)
                rr_idx = engine.dev(np.flatnonzero(rr_test))
                coef[rr_idx] = coef_r[rr_idx]
        pvals = _asl.asl_from_stats(stt, approx, lambda idx: engine.host(coef[engine.dev(np.asarray(idx, dtype=np.int64))]), num_cpus, resampling)
        for k in range(n_ch):
            c = int(first[lo + so[k]])
            if not good[k].any():
                continue
            for cc in members[frozenset((int(idx1[c]), int(idx2[c])))]:
                corr_coef[cc], corr_se[cc], corr_asl[cc] = stt[k, 0], stt[k, 1], pvals[k]
    m['2d_ht'] = {'treatment': treatment, 'covariate': covariate, 'corr_coef': corr_coef, 'corr_se': corr_se, 'corr_asl': corr_asl}
    if treatment_for_gene is not None:
        m['2d_ht']['treatment_for_gene'] = treatment_for_gene                      # main.py:513-514
    st.last_bootstrap2d = bs
    st.last_chunk2d = (bounds[-2], bounds[-1]) if P_ else (0, 0)                   # diagnostics / tests: pair range of the last chunk
    if not inplace:
        return adata


# ----------------------------------------------------------------------------------------------
# getters
# ----------------------------------------------------------------------------------------------


def get_1d_moments(adata, groupby=None):
    """log-mean / log-residual-variance tables per group (reference: memento/main.py:523-582, groupby=None form)."""
    m = adata.uns['memento']
    names = _var_names(adata).tolist()
    mean_df = pd.DataFrame({'gene': names})
    var_df = pd.DataFrame({'gene': names})
    cell_counts = {k: v.shape[0] for k, v in m['group_cells'].items()}
    with np.errstate(invalid="ignore", divide="ignore"):
        for group, val in m['1d_moments'].items():
            mean_df[group] = np.log(val[0])
            var_df[group] = np.log(val[2])
    if groupby is None:
        return mean_df, var_df, cell_counts
    # cell-count weighted averages of the log moments over the groups whose label contains the key
    # (reference: memento/main.py:544-582)
    keys = adata.obs[groupby].astype(str).drop_duplicates().values if groupby != 'ALL' else ['sg']
    gm, gv = pd.DataFrame({'gene': names}), pd.DataFrame({'gene': names})
    for key in keys:
        sm = sv = cm = cv = 0
        for group, val in m['1d_moments'].items():
            if group == 'all' or key not in group:
                continue
            with np.errstate(invalid="ignore", divide="ignore"):
                lm, lv = np.log(val[0]), np.log(val[2])
            lm[np.isnan(lm)] = 0
            lv[np.isnan(lv)] = 0
            sm = sm + lm * cell_counts[group]
            cm = cm + (val[0] > 0) * cell_counts[group]
            sv = sv + lv * cell_counts[group]
            cv = cv + (val[2] > 0) * cell_counts[group]
        with np.errstate(invalid="ignore", divide="ignore"):
            gm[groupby + '_' + key] = sm / cm
            gv[groupby + '_' + key] = sv / cv
    return gm.copy(), gv.copy()


def get_1d_ht_result(adata):
    """DataFrame of DE / DV coefficients, standard errors and p-values (reference: memento/main.py:635-655)."""
    ht = adata.uns['memento']['1d_ht']
    names = ht['gene_names'] if 'gene_names' in ht else _var_names(adata)       # 'gene_names': gathered multi-GPU result
    if 'treatment_for_gene' in ht:
        pairs = [(g, t) for g in names for t in ht['treatment_for_gene'][g]]
    else:
        pairs = list(itertools.product(names, ht['treatment'].columns))
    df = pd.DataFrame(pairs, columns=['gene', 'tx'])
    df['de_coef'], df['de_se'], df['de_pval'] = ht['mean_coef'], ht['mean_se'], ht['mean_asl']
    df['dv_coef'], df['dv_se'], df['dv_pval'] = ht['var_coef'], ht['var_se'], ht['var_asl']
    return df


def prepare_to_save(adata, keep=False):
    """Drop objects that cannot be written to disk (reference: memento/main.py:673-683) -- here also the
    device handles."""
    m = adata.uns['memento']
    m.pop('_hip', None)
    for group in m['groups'] + ['all']:
        m['mv_regressor'].pop(group, None)
    m['group_cells'] = {g: v.shape for g, v in m['group_cells'].items()}


def get_2d_moments(adata, groupby=None):
    """Correlation table per group (reference: memento/main.py:585-632, groupby=None form)."""
    m = adata.uns['memento']
    df = pd.DataFrame(m['2d_moments']['gene_pairs'], columns=['gene_1', 'gene_2'])
    cell_counts = {k: v.shape[0] for k, v in m['group_cells'].items()}
    for group, val in m['2d_moments'].items():
        if 'sg^' not in group:
            continue
        df[group] = val['corr']
    if groupby is None:
        return df, cell_counts
    keys = adata.obs[groupby].astype(str).drop_duplicates().values if groupby != 'ALL' else ['sg']
    out = pd.DataFrame({'gene_1': df['gene_1'], 'gene_2': df['gene_2']})       # reference: memento/main.py:604-632
    for key in keys:
        sc = cc = 0
        for group, val in m['2d_moments'].items():
            if 'sg^' not in group or key not in group:
                continue
            c = val['corr'].copy()
            valid = ~np.isnan(c)
            c[~valid] = 0
            sc = sc + c * cell_counts[group]
            cc = cc + valid * cell_counts[group]
        with np.errstate(invalid="ignore", divide="ignore"):
            out[groupby + '_' + key] = sc / cc
    return out.copy()


def get_2d_ht_result(adata):
    """DataFrame of correlation coefficients, standard errors and p-values (reference: memento/main.py:658-670)."""
    m = adata.uns['memento']
    df = pd.DataFrame(m['2d_moments']['gene_pairs'], columns=['gene_1', 'gene_2'])
    df['corr_coef'], df['corr_se'], df['corr_pval'] = m['2d_ht']['corr_coef'], m['2d_ht']['corr_se'], m['2d_ht']['corr_asl']
    return df
