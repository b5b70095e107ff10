"""The one-wave-per-chain kernel (mm_boot1d_chain) by itself: step time of the N longest chains of a config at several
occupancies, and -- when the library was built with -DBOOT_STAMPS -- shader cycles per sampler call by sampler.
usage: python tools/chain_micro.py [config=C3 | C3@cells[@num_boot]] n_chains [n_chains ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd, torch, scipy.sparse as sp
import bench
from scrna_parameter_estimation_amd import AnnDataLite, engine, memento, _lib


def main():
    spec = sys.argv[1] if len(sys.argv) > 1 else "C3"
    name, *rest = spec.split("@")
    cfg = dict(bench.CONFIGS[name])
    if rest:
        cfg["cells"] = int(rest[0])
    if len(rest) > 1:
        cfg["num_boot"] = int(rest[1])
    counts = [int(x) for x in sys.argv[2:]] or [1024]
    N, G, B = cfg["cells"], cfg["genes"], cfg["num_boot"]
    ng = cfg["n_cond"] * cfg["n_rep"]
    csr = bench.synth_device_csr(cfg, 20250117, torch)
    grp = np.random.default_rng(20250117).integers(0, ng, size=N)
    obs = pd.DataFrame({"cond": grp // cfg["n_rep"], "rep": grp % cfg["n_rep"], "q": np.full(N, 0.07)})
    adata = AnnDataLite(sp.csr_matrix((N, G), dtype=np.float32), obs, pd.DataFrame(index=[f"g{i}" for i in range(G)]))
    memento.setup_memento(adata, q_column="q", device_csr=csr)
    memento.create_groups(adata, label_columns=["cond", "rep"])
    memento.compute_1d_moments(adata, min_perc_group=0.7, subset_var=False)
    m = adata.uns["memento"]
    st = m["_hip"]
    gq = np.array([m["group_q"][g] for g in m["groups"]])
    bs = engine.Bootstrap1D(st.blocks, st.gene_idx, st.maxx, st.sf_bin, st.sf_table, gq, B)
    by_len = np.argsort(-bs.K, kind="stable")
    r = np.random.default_rng(0).random((2, bs.n_pairs))
    buf = torch.zeros((1 << 20,), dtype=torch.int64, device="cuda")
    for mode in ("chain", "tile"):
        for n in counts:
            skip = np.ones(bs.n_pairs, dtype=bool)
            skip[by_len[:n]] = False
            engine.CHAIN_MIN_K = 2 if mode == "chain" else 0
            engine.CHAIN_LONE = False
            engine.PACK_MAX_RESIDENT = 10 ** 9
            bs.alloc_outputs(np.zeros(bs.n_pairs), np.zeros(bs.n_pairs))
            buf.zero_()
            _lib.call("mm_debug_wave_clock", engine.P(buf))
            bs.run(skip, r[0], r[1], m["mv_regressor"]["all"], fill_mode=1, target_waves=(10 ** 7 if mode == "tile" else None))
            torch.cuda.synchronize()
            _lib.call("mm_debug_wave_clock", None)
            raw = buf.cpu().numpy()
            if mode == "chain":
                wc = raw[engine.CHAIN_CLOCK_OFF: engine.CHAIN_CLOCK_OFF + 8 * bs.n_chain].reshape(-1, 8)
                Kc = bs.K[bs.chain_pairs]
                dur = (wc[:, 1] - wc[:, 0]) / 1e8
                us = dur / ((Kc - 1) * B) * 1e6
                line = (f"chain kernel alone, {bs.n_chain} chains (K {Kc.max()}..{Kc.min()}): span {(wc[:, 1].max() - wc[:, 0].min()) / 1e8:.3f} s; "
                        f"us/step first 64: {np.median(us[:64]):.3f}, all: median {np.median(us):.3f} p90 {np.quantile(us, .9):.3f}")
                if wc[:, 6].any():
                    s = wc[:64].astype(np.float64)
                    steps = ((Kc[:64] - 1) * B).astype(np.float64)
                    line += (f"\n     stamps (first 64 chains): inversion calls {s[:, 2].sum() / steps.sum():.2f}/step at {s[:, 3].sum() / max(1, s[:, 2].sum()):.0f} cyc, "
                             f"BTPE calls {s[:, 4].sum() / steps.sum():.2f}/step at {s[:, 5].sum() / max(1, s[:, 4].sum()):.0f} cyc, whole step {s[:, 6].sum() / steps.sum():.0f} cyc, "
                             f"shader clock {s[:, 6].sum() / dur[:64].sum() / 1e9:.2f} GHz")
                print(line, flush=True)
            else:
                nt = bs.n_tiles
                wt = raw[: nt * 4].reshape(-1, 4)
                lanes = (bs.slot_K.reshape(nt, 64) > 0).sum(axis=1)
                steps = np.diff(bs.tile_ptr) * B
                dur = (wt[:, 1] - wt[:, 0]) / 1e8
                us = dur / steps * 1e6
                first = np.argsort(-steps)[:64]
                print(f"tile kernel alone, {n} chains in {nt} tiles (lanes max {lanes.max()}): span {(wt[:, 1].max() - wt[:, 0].min()) / 1e8:.3f} s; "
                      f"us/step of the 64 longest: {np.median(us[first]):.3f}, all: median {np.median(us):.3f}", flush=True)


if __name__ == "__main__":
    main()
