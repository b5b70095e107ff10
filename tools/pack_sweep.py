"""Sweep of the replay kernel's lane-packing constants (engine.PACK_C0 / PACK_C1 / PACK_WAVES) on one shape, in ONE process.
These are module attributes the tools set directly; the product path reads no environment variable.
usage: python tools/pack_sweep.py [config=C3 | C3@cells[@num_boot]] "c0,c1,waves[,max_resident[,pair_slots[,waves3[,oversubscription]]]]" ..."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, torch, scipy.sparse as sp
import bench
from scrna_parameter_estimation_amd import AnnDataLite, engine, memento, _lib

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
combos = [tuple(float(x) for x in a.split(",")) for a in sys.argv[2:]] or [(220, 3, 2000)]
cfg = dict(bench.CONFIGS[name.split("@")[0]])
if "@" in name:                       # e.g. C3@500000 or C3@500000@1000: the same shape with other cells / bootstraps
    cfg["cells"] = int(name.split("@")[1])
    if len(name.split("@")) > 2:
        cfg["num_boot"] = int(name.split("@")[2])
N, G, B = cfg["cells"], cfg["genes"], cfg["num_boot"]
ng = cfg["n_cond"] * cfg["n_rep"]
csr = bench.synth_device_csr(cfg, 20250117, torch)
grp = np.random.default_rng(20250117).integers(0, ng, size=N)
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
bs.alloc_outputs(np.zeros(bs.n_pairs), np.zeros(bs.n_pairs))
timer = ctypes.c_void_p(); _lib.call("mm_timer_create", ctypes.byref(timer)); s = engine._stream(); ms = ctypes.c_float()
for combo in combos:
    c0, c1, waves = combo[:3]
    engine.PACK_C0, engine.PACK_C1, engine.PACK_WAVES = float(c0), float(c1), int(waves)
    engine.PACK3_C0, engine.PACK3_C1 = float(c0), float(c1)
    engine.PACK_MAX_RESIDENT = int(combo[3]) if len(combo) > 3 else 2048
    engine.PAIR_SLOTS = int(combo[4]) if len(combo) > 4 else 1024
    engine.PACK_WAVES3 = int(combo[5]) if len(combo) > 5 else 2750
    engine.PACK_OVERSUB = int(combo[6]) if len(combo) > 6 else 0
    _lib.call("mm_timer_begin", timer, s)
    bs.run(skip, r[0], r[1], m["mv_regressor"]["all"], fill_mode=1)
    _lib.call("mm_timer_end", timer, s)
    _lib.call("mm_timer_elapsed_ms", timer, ctypes.byref(ms))
    print(f"{name} C0={c0:g} C1={c1:g} waves={int(waves)} waves3={engine.PACK_WAVES3} -> tiles {bs.n_tiles} of them chain waves {bs.n_chain} ({engine.PACK_LAST.get('chosen')})  bootstrap {ms.value:.0f} ms; model: longest tile {engine.PACK_LAST.get('longest_resident', 0) * B * 1e-6:.2f} s, work / rate {engine.PACK_LAST.get('work_resident', 0) * B * 1e-6:.2f} s, chains {engine.PACK_LAST.get('chains')}, K max {int(bs.K.max())} mean {bs.K[bs.K >= 2].mean():.0f}", flush=True)
