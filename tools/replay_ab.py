"""A/B of the replay bootstrap kernel's arithmetic in ONE process (interleaved rounds): numpy's fp64 arithmetic in every sampler
search loop vs the guarded fp32 fast paths (csrc/npy_rng.h).  Checks that both give bit-identical replicate moments.
usage: python tools/replay_ab.py [config=C3] [rounds=3]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, torch, scipy.sparse as sp
import bench
from scrna_parameter_estimation_amd import AnnDataLite, engine, memento, _lib

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = bench.CONFIGS[name]
N, G, B = cfg["cells"], cfg["genes"], cfg["num_boot"]
ng = cfg["n_cond"] * cfg["n_rep"]
csr = bench.synth_device_csr(cfg, 20250117, torch)
rng = np.random.default_rng(20250117)
grp = rng.integers(0, ng, size=N)
obs = pd.DataFrame({"cond": grp // cfg["n_rep"], "rep": grp % cfg["n_rep"], "q": np.full(N, 0.07)})
adata = AnnDataLite(sp.csr_matrix((N, G), dtype=np.float32), obs, pd.DataFrame(index=[f"g{i}" for i in range(G)]))
memento.setup_memento(adata, q_column="q", device_csr=csr)
memento.create_groups(adata, label_columns=["cond", "rep"])
memento.compute_1d_moments(adata, min_perc_group=0.7, subset_var=False)
m = adata.uns["memento"]; st = m["_hip"]
gq = np.array([m["group_q"][g] for g in m["groups"]])
bs = engine.Bootstrap1D(st.blocks, st.gene_idx, st.maxx, st.sf_bin, st.sf_table, gq, B)
skip = ~(bs.K >= 2)
r = np.random.default_rng(0).random((2, bs.n_pairs))
zeros = np.zeros(bs.n_pairs)
bs.alloc_outputs(zeros, zeros)
fit = m["mv_regressor"]["all"]
timer = ctypes.c_void_p(); _lib.call("mm_timer_create", ctypes.byref(timer)); s = engine._stream(); ms = ctypes.c_float()
res = {0: [], 1: []}
keep = {}
for rd in range(rounds):
    for exact in (1, 0):
        _lib.call("mm_debug_replay_arith", exact)
        _lib.call("mm_timer_begin", timer, s)
        bs.run(skip, r[0], r[1], fit, fill_mode=1)
        _lib.call("mm_timer_end", timer, s)
        _lib.call("mm_timer_elapsed_ms", timer, ctypes.byref(ms))
        res[exact].append(ms.value)
        if rd == 0:
            keep[exact] = (bs.ym.clone(), bs.yv.clone())
_lib.call("mm_debug_replay_arith", 0)
same = bool(torch.equal(keep[0][0].view(torch.int64), keep[1][0].view(torch.int64)) and torch.equal(keep[0][1].view(torch.int64), keep[1][1].view(torch.int64)))
print(f"{name}: chains {int((~skip).sum())} tiles {bs.n_tiles} draws/replicate {bs.draws_per_replicate}  exact fp64 ms {np.round(res[1], 1).tolist()}  "
      f"guarded fp32 ms {np.round(res[0], 1).tolist()}  speed-up {np.median(res[1]) / np.median(res[0]):.3f}  bit-identical replicates: {same}", flush=True)
assert same
