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


def gather_1d_ht(comm, names, out):
    """Result gather of the gene-sharded 1D test: every rank contributes its genes' flat result vectors (gene-major x treatment,
    reference order memento/main.py:399-404) and the names of its kept genes; since the shards are contiguous gene ranges in
    rank order, the concatenation in rank order IS the unsharded run's ordering.  Returns (all names, dict of full vectors)."""
    parts = comm.allgather_objects({"names": list(names), "out": {k: np.asarray(v) for k, v in out.items()}})
    all_names = [n for p_ in parts for n in p_["names"]]
    full = {k: np.concatenate([p_["out"][k] for p_ in parts]) for k in out}
    return all_names, full


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
