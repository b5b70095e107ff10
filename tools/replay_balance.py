"""Where does the replay kernel's time go across waves?  Runs one C3-shaped (or --config) bootstrap with the
mm_debug_wave_clock hook and reports, per wave: duration, lanes, steps, share of BTPE steps; per SIMD: busy time.
usage: python tools/replay_balance.py [config] [out.npz]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, torch, scipy.sparse as sp
import bench
from scrna_parameter_estimation_amd import AnnDataLite, memento, engine, _lib


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C3"
    out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/replay_balance.npz"
    cfg = bench.CONFIGS[name]
    N, G, B = cfg["cells"], cfg["genes"], cfg["num_boot"]
    csr = bench.synth_device_csr(cfg, 20250117, torch)
    rng = np.random.default_rng(20250117)
    grp = rng.integers(0, cfg["n_cond"] * cfg["n_rep"], size=N)
    obs = pd.DataFrame({"cond": grp // cfg["n_rep"], "rep": grp % cfg["n_rep"], "q": np.full(N, 0.07)})
    adata = AnnDataLite(sp.csr_matrix((N, G), dtype=np.float32), obs, pd.DataFrame(index=[f"g{i}" for i in range(G)]))
    memento.setup_memento(adata, q_column="q", device_csr=csr)
    memento.create_groups(adata, label_columns=["cond", "rep"])
    memento.compute_1d_moments(adata, min_perc_group=0.7, subset_var=False)
    m = adata.uns["memento"]
    st = m["_hip"]
    groups = m["groups"]
    gq = np.array([m["group_q"][g] for g in groups])
    bs = engine.Bootstrap1D(st.blocks, st.gene_idx, st.maxx, st.sf_bin, st.sf_table, gq, B)
    tm = np.stack([m["1d_moments"][g][0] for g in groups]).T.reshape(-1)
    tv = np.stack([m["1d_moments"][g][2] for g in groups]).T.reshape(-1)
    with np.errstate(all="ignore"):
        skip = ~(np.isfinite(np.log(tm)) & np.isfinite(np.log(tv)))
        bs.alloc_outputs(np.log(tm), np.log(tv))
    r = np.random.default_rng(0).random((2, bs.n_pairs))
    buf = torch.zeros((1 << 20,), dtype=torch.int64, device="cuda")   # tiles at 0, the chain kernel at engine.CHAIN_CLOCK_OFF
    _lib.call("mm_debug_wave_clock", engine.P(buf))
    torch.cuda.synchronize(); t0 = time.time()
    bs.run(skip, r[0], r[1], m["mv_regressor"]["all"])
    torch.cuda.synchronize(); dt = time.time() - t0
    _lib.call("mm_debug_wave_clock", None)
    nt = bs.n_tiles
    wc = buf.cpu().numpy().reshape(-1, 4)[:nt]
    start, end, hw, xcc = wc[:, 0], wc[:, 1], wc[:, 2], wc[:, 3] & 0xF
    dur = (end - start) / 1e8                       # 100 MHz
    t_first = start.min()
    lanes = (bs.slot_K.reshape(nt, 64) > 0).sum(axis=1)
    steps = np.diff(bs.tile_ptr)
    # BTPE share per tile-step: n*pk_eff > 30 in expectation (n ~ nobs * remaining_p == cells left)
    pk = engine.host(bs._ops[0]).reshape(-1, 64)
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    uid = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    print(f"{name}: run {dt:.2f}s, tiles {nt}, kernel span {(end.max()-t_first)/1e8:.2f}s; packing {engine.PACK_LAST.get('chosen')}, "
          f"K mean {bs.K[bs.K >= 2].mean():.0f} max {bs.K.max()}")
    print("wave duration s: min %.2f p10 %.2f median %.2f p90 %.2f max %.2f ; sum %.1f" % (dur.min(), *np.quantile(dur, [.1, .5, .9]), dur.max(), dur.sum()))
    print("start offsets s: max %.3f ; waves starting later than 0.1 s: %d" % ((start.max() - t_first) / 1e8, ((start - t_first) / 1e8 > 0.1).sum()))
    u, inv = np.unique(uid, return_inverse=True)
    busy = np.bincount(inv, weights=dur); cnt = np.bincount(inv)
    print(f"distinct SIMDs used {len(u)}; waves per SIMD: " + ", ".join(f"{k}:{(cnt==k).sum()}" for k in np.unique(cnt)))
    print("per-SIMD summed wave time s: min %.2f median %.2f max %.2f" % (busy.min(), np.median(busy), busy.max()))
    print("per-wave: corr(dur, steps) %.3f corr(dur, lanes) %.3f" % (np.corrcoef(dur, steps)[0, 1], np.corrcoef(dur, lanes)[0, 1]))
    per_step = dur / (steps * B) * 1e6
    for lo, hi in ((1, 8), (8, 16), (16, 24), (24, 32), (32, 48), (48, 65)):
        sel = (lanes >= lo) & (lanes < hi)
        if sel.any():
            print(f"  lanes [{lo},{hi}): {sel.sum():5d} waves, us per wave-step median {np.median(per_step[sel]):.2f} (p10 {np.quantile(per_step[sel], .1):.2f}, p90 {np.quantile(per_step[sel], .9):.2f}); steps median {np.median(steps[sel]):.0f}")
    # step time of the OLDER wave of each SIMD (first half of the grid) by exact lane count: the table engine._PACK_US holds
    older = np.arange(nt) < min(1024, nt // 2 + nt % 2)
    tab = []
    for L in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 18, 20, 24, 28, 32, 36, 40, 44, 48, 56, 64):
        sel = older & (lanes == L)
        if sel.sum() >= 3:
            tab.append((L, round(float(np.median(per_step[sel])), 2), int(sel.sum())))
    print("older-wave us per step by lanes (L, us, n):", tab)
    np.savez_compressed(out, wc=wc, lanes=lanes, steps=steps, uid=uid, B=B)


if __name__ == "__main__":
    main()
