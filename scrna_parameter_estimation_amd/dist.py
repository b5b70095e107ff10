"""Multi-GPU: one process per GPU, GENES sharded over ranks (every rank holds all cells x its genes).

Genes are independent in every phase of the hot path after the size factors (SURVEY.md section 8e;
/root/reference/memento/main.py:379-397), so the data path needs NO collective.  Two small exchanges
remain, both outside the O(nnz) kernels:
  * per-cell totals for the size factors  -> all-reduce(sum) of an N-vector  (estimator.py:65, :73)
  * the pooled mean-variance fit          -> all-gather of per-gene moments   (main.py:232-245, :68-71)
They run over torch.distributed: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.

One-call use (every rank runs the same script under torchrun, the reference's fan-out + scatter-back of
memento/main.py:379-412 spread over GPUs):
    comm = Comm()
    memento.setup_memento(adata, 'q', comm=comm, shard=True)   # full X on every rank -> device-side column split
    memento.create_groups(adata, [...]); memento.compute_1d_moments(adata)
    memento.ht_1d_moments(adata, covariate=..., treatment=..., ...)   # ends with the gather below
    memento.get_1d_ht_result(adata)                            # the FULL table, gene order of the unsharded run, on every rank
"""

import numpy as np


class Comm:
    """Thin numpy-in / numpy-out wrapper around torch.distributed for the two exchanges above."""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist

        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")

    def allreduce_sum(self, a):
        t = self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def allgather_concat(self, a):
        """Concatenate 1-D float arrays of different lengths from all ranks, in rank order."""
        torch, dist = self.torch, self.dist
        a = np.ascontiguousarray(a, dtype=np.float64).ravel()
        n = torch.tensor([a.shape[0]], dtype=torch.int64, device=self.device)
        sizes = [torch.zeros_like(n) for _ in range(self.world)]
        dist.all_gather(sizes, n)
        sizes = [int(s.item()) for s in sizes]
        mx = max(sizes + [1])
        buf = torch.zeros(mx, dtype=torch.float64, device=self.device)
        buf[: a.shape[0]] = torch.from_numpy(a).to(self.device)
        out = [torch.zeros_like(buf) for _ in range(self.world)]
        dist.all_gather(out, buf)
        return np.concatenate([o[:s].cpu().numpy() for o, s in zip(out, sizes)])


    def allgather_objects(self, obj):
        """Small python objects (dicts of numpy arrays, name lists) from every rank, in rank order."""
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out


def gather_1d_ht(comm, names, out, gene_pos=None, n_tests=None):
    """Result gather of the gene-sharded 1D test: every rank contributes its genes' flat result vectors (gene-major x treatment,
    reference order memento/main.py:399-404), the names of its kept genes, their positions ``gene_pos`` in the UNSHARDED gene
    order and the number of tests of each (``n_tests``; one per treatment column unless treatment_for_gene varies it).  The
    genes of all ranks are put back in the unsharded order -- shards need not be contiguous ranges (cost-balanced shards are
    not).  ``gene_pos`` None: contiguous shards in rank order (the concatenation in rank order is the unsharded order).
    Returns (all names, dict of full vectors)."""
    names = list(names)
    k0 = next(iter(out))
    if n_tests is None:
        per = len(out[k0]) // max(1, len(names))
        n_tests = np.full(len(names), per, dtype=np.int64)
    # the numeric vectors travel as tensors through the backend's all-gather (RCCL on GPUs); only the gene names go as objects
    cat = {k: comm.allgather_concat(np.asarray(v, dtype=np.float64)) for k, v in out.items()}
    nt = comm.allgather_concat(np.asarray(n_tests, dtype=np.float64)).astype(np.int64)
    name_parts = comm.allgather_objects(names)
    cat_names = np.array([n for part in name_parts for n in part], dtype=object)
    have_pos = comm.allgather_concat(np.array([0.0 if gene_pos is None else 1.0]))
    if not have_pos.all():
        return cat_names.tolist(), cat
    pos = comm.allgather_concat(np.asarray(gene_pos, dtype=np.float64)).astype(np.int64)
    order = np.argsort(pos, kind="stable")
    first = np.concatenate([[0], np.cumsum(nt)])[:-1]                    # first test of every gene in the rank-order concatenation
    idx = np.concatenate([np.arange(first[g], first[g] + nt[g]) for g in order]) if len(order) else np.zeros(0, dtype=np.int64)
    return cat_names[order].tolist(), {k: v[idx] for k, v in cat.items()}


def shard_stream_uniforms(comm, gene_pos, live):
    """The hash uniforms of this rank's chains out of the ONE global np.random stream of the unsharded run.

    The reference takes two uniforms per live (gene, group) chain from the global stream, gene after gene in the unsharded order
    (memento/bootstrap.py:62, :65; fan-out and scatter-back memento/main.py:379-404).  ``gene_pos`` [genes of this rank]: their
    positions in that order; ``live`` [genes][groups] bool: chains that draw.  Every rank gathers (position, live chains) of all
    genes, draws the whole stream -- identical on every rank, the caller seeds all ranks alike -- and keeps the entries at its
    own chains' positions.  Returns (r1, r0), each [genes * groups] (0 where the chain is not live).  All ranks leave with the
    global stream in the same state, as after the unsharded call."""
    live = np.asarray(live, dtype=bool)
    G, ng = live.shape
    parts = comm.allgather_objects({"pos": np.asarray(gene_pos, dtype=np.int64), "live": live.sum(axis=1).astype(np.int64)})
    pos = np.concatenate([p_["pos"] for p_ in parts])
    nlive = np.concatenate([p_["live"] for p_ in parts])
    order = np.argsort(pos, kind="stable")
    start = np.empty(len(pos), dtype=np.int64)
    start[order] = 2 * (np.cumsum(nlive[order]) - nlive[order])
    first = sum(len(p_["pos"]) for p_ in parts[:comm.rank])
    mine = start[first:first + G]
    u = np.random.random(int(2 * nlive.sum()))
    within = 2 * (np.cumsum(live, axis=1) - live)                 # [gene][group]: offset of a live chain inside its gene
    at = (mine[:, None] + within).reshape(-1)
    livef = live.reshape(-1)
    r1, r0 = np.zeros(G * ng), np.zeros(G * ng)
    r1[livef], r0[livef] = u[at[livef]], u[at[livef] + 1]
    return r1, r0


def gather_pair_results(comm, positions, values, n_total):
    """2D counterpart: ``values`` (dict of arrays aligned with this rank's pair block) and the block's ``positions`` in the
    caller's pair list (from ``shard_pairs``) -> dict of full-length arrays in the caller's order, on every rank."""
    parts = comm.allgather_objects({"pos": np.asarray(positions, dtype=np.int64), "val": {k: np.asarray(v) for k, v in values.items()}})
    full = {k: np.full(n_total, np.nan) for k in values}
    for p_ in parts:
        for k in values:
            full[k][p_["pos"]] = p_["val"][k]
    return full


def shard_genes(n_genes, rank, world):
    """Contiguous gene shard [lo, hi) of rank ``rank``."""
    lo = (n_genes * rank) // world
    hi = (n_genes * (rank + 1)) // world
    return lo, hi


def gene_cost(mean, filter_mean_thresh=0.07):
    """Predicted bootstrap cost of a gene from its mean count per cell (all a rank knows before the groups exist): genes below
    the expression filter are never tested (memento/main.py:202-215) and cost next to nothing; a tested gene's chains have about
    n_sf_bins x (distinct counts) bins, which grows like the spread of its counts -- ~sqrt(mean) on top of a floor."""
    mean = np.asarray(mean, dtype=np.float64)
    return np.where(mean >= 0.8 * filter_mean_thresh, 1.0 + 1.5 * np.sqrt(mean), 0.002)


def shard_genes_balanced(cost, rank, world):
    """Cost-balanced gene shard of ``rank`` (ascending gene indices): longest-processing-time-first over the predicted costs --
    genes in descending cost, each to the rank with the least load so far (ties: the lowest rank) -- so that every rank gets the
    same mix of long and short bootstrap chains.  A contiguous split (shard_genes) gives each rank whatever its range holds.
    Deterministic: every rank computes the same assignment from the same costs."""
    import heapq

    cost = np.asarray(cost, dtype=np.float64)
    order = np.argsort(-cost, kind="stable")
    heap = [(0.0, r) for r in range(world)]
    owner = np.empty(len(cost), dtype=np.int32)
    for g in order:
        load, r = heapq.heappop(heap)
        owner[g] = r
        heapq.heappush(heap, (load + cost[g], r))
    return np.flatnonzero(owner == rank)


def shard_pairs(gene_pairs, rank, world):
    """Block of the gene-pair list owned by ``rank`` (2D path: compute_2d_moments / ht_2d_moments).

    Pairs are independent (reference: memento/main.py:485-501), so every rank runs the unchanged 2D API on its own
    block and there is no collective; each rank needs the count columns of the genes in its block only
    (SURVEY.md section 8e).  Pairs are ordered by (first gene, second gene) before cutting, so a rank's block
    touches few distinct first genes; the inverse permutation puts gathered results back in the caller's order.
    Returns (pairs of this rank, their positions in ``gene_pairs``)."""
    order = sorted(range(len(gene_pairs)), key=lambda i: (str(gene_pairs[i][0]), str(gene_pairs[i][1])))
    lo = (len(order) * rank) // world
    hi = (len(order) * (rank + 1)) // world
    mine = order[lo:hi]
    return [gene_pairs[i] for i in mine], np.asarray(mine, dtype=np.int64)
