"""Split of the replay bootstrap between the one-wave-per-chain kernel (mm_boot1d_chain) and the tile kernel: for every
setting given, time one launch pair at the named config, check that the replicate rows are bit-identical to the first
setting's, and report the chain kernel's step time by chain length from its wave clocks.
usage: python tools/chain_sweep.py [config=C3 | C3@cells[@num_boot]] setting [setting ...]
setting: "tiles" (lock-step tile kernel only) | "lone[:waves]" (lock-step tiles; chains alone in their tile -> chain kernel; packer
target waves) | "async:K[:L]" (lane-asynchronous tiles of L chains per wave; chains with >= K bins -> chain kernel, 0 = none) | min_k (int: lock-step tiles,
chains with at least that many bins -> chain kernel, no lone rule)"""
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
    mins = sys.argv[2:] or ["tiles", "lone"]
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
    skip = ~(bs.K >= 2)
    K = bs.K[~skip]
    print(f"{spec}: chains {len(K)}, K mean {K.mean():.1f} max {K.max()}; quantiles 10/25/50/75/90/99 % "
          f"{np.quantile(K, [.1, .25, .5, .75, .9, .99]).astype(int).tolist()}; sum(K-1) {int((K - 1).sum())}", flush=True)
    edges = [2, 25, 50, 75, 100, 125, 150, 200, 250, 300, 400, 100000]
    hist = np.histogram(K, bins=edges)[0]
    print("K histogram:", ", ".join(f"[{a},{b}) {c}" for a, b, c in zip(edges[:-1], edges[1:], hist)), flush=True)
    r = np.random.default_rng(0).random((2, bs.n_pairs))
    if os.environ.get("MM_ROWS_MOD"):       # timing experiment (WRONG results): tiles read their operand rows out of a cache-resident region
        _lib.call("mm_debug_replay_rows_mod", int(os.environ["MM_ROWS_MOD"]))
    timer, ms = ctypes.c_void_p(), ctypes.c_float()
    _lib.call("mm_timer_create", ctypes.byref(timer))
    buf = torch.zeros((1 << 20,), dtype=torch.int64, device="cuda")
    ref = None
    for mk in mins:
        target = None
        engine.TILE_MODE = "lockstep"
        if mk.startswith("async"):
            engine.TILE_MODE = "async"
            engine.CHAIN_MIN_K, engine.CHAIN_LONE = 0, False
            parts = mk.split(":")
            engine.ASYNC_CHAIN_MIN_K = int(parts[1]) if len(parts) > 1 else 160
            engine.ASYNC_LANES = int(parts[2]) if len(parts) > 2 else 64
        elif mk == "tiles":
            engine.CHAIN_MIN_K, engine.CHAIN_LONE = 0, False
        elif mk.startswith("lone"):
            engine.CHAIN_MIN_K, engine.CHAIN_LONE = 0, True
            if ":" in mk:
                target = int(mk.split(":")[1])
                engine.PACK_WAVES3 = target
        else:
            engine.CHAIN_MIN_K, engine.CHAIN_LONE = int(mk), False
        bs.alloc_outputs(np.zeros(bs.n_pairs), np.zeros(bs.n_pairs))
        times = []
        for rep in range(2):
            buf.zero_()
            _lib.call("mm_debug_wave_clock", engine.P(buf))
            stream = engine._stream()
            _lib.call("mm_timer_begin", timer, stream)
            bs.run(skip, r[0], r[1], m["mv_regressor"]["all"], fill_mode=1)
            _lib.call("mm_timer_end", timer, stream)
            _lib.call("mm_timer_elapsed_ms", timer, ctypes.byref(ms))
            _lib.call("mm_debug_wave_clock", None)
            times.append(round(ms.value, 1))
        same = ""
        if ref is None:
            ref = (bs.ym.clone(), bs.yv.clone())
        else:
            eq = lambda a, b: bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all().item())
            same = f"  rows bit-identical to the first setting: {eq(ref[0], bs.ym) and eq(ref[1], bs.yv)}"
        print(f"setting {mk}: chains one-per-wave {bs.n_chain}, async lanes {bs.n_async}, lock-step tiles {bs.n_tiles}; ms {times}{same}", flush=True)
        raw = buf.cpu().numpy()
        if bs.n_chain:
            wc = raw[engine.CHAIN_CLOCK_OFF: engine.CHAIN_CLOCK_OFF + 8 * bs.n_chain].reshape(-1, 8)
            dur = (wc[:, 1] - wc[:, 0]) / 1e8
            Kc = bs.K[bs.chain_pairs]
            us = dur / ((Kc - 1) * B) * 1e6
            t0 = wc[:, 0].min()
            print(f"   chain kernel: span {(wc[:, 1].max() - t0) / 1e8:.3f} s, last start {(wc[:, 0].max() - t0) / 1e8:.3f} s, "
                  f"sum of wave time {dur.sum():.1f} s; us per step: all {np.median(us):.3f}", flush=True)
            for lo, hi in ((0, 64), (64, 1024), (1024, 4096), (4096, 10 ** 9)):
                sel = (np.arange(bs.n_chain) >= lo) & (np.arange(bs.n_chain) < hi)
                if sel.any():
                    print(f"     chains [{lo},{min(hi, bs.n_chain)}) (K {Kc[sel].max()}..{Kc[sel].min()}): us/step median {np.median(us[sel]):.3f} "
                          f"p10 {np.quantile(us[sel], .1):.3f} p90 {np.quantile(us[sel], .9):.3f}; wave s max {dur[sel].max():.3f}", flush=True)
        if bs.n_async:
            L = max(1, min(64, engine.ASYNC_LANES))
            nw = -(-bs.n_async // L)
            wt = raw[: nw * 4].reshape(-1, 4)
            dur = (wt[:, 1] - wt[:, 0]) / 1e8
            Ka = bs.K[bs.async_pairs]
            kmax = np.array([Ka[i * L:(i + 1) * L].max() for i in range(nw)])
            ksum = np.array([(Ka[i * L:(i + 1) * L] - 1).sum() for i in range(nw)])
            print(f"   async kernel: {nw} waves, span {(wt[:, 1].max() - wt[:, 0].min()) / 1e8:.3f} s, longest wave {dur.max():.3f} s, sum of wave time {dur.sum():.1f} s; "
                  f"passes per (longest chain's) draw: median {np.median(wt[:, 2] / (kmax * B)):.2f}; us per pass: median {np.median(dur / wt[:, 2]) * 1e6:.2f} "
                  f"(first wave {dur[0] / wt[0, 2] * 1e6:.2f}, last {dur[-1] / wt[-1, 2] * 1e6:.2f}); us per draw of a lane: {np.median(dur * L / (ksum * B)) * 1e6:.3f}", flush=True)
            ph = raw[(1 << 19): (1 << 19) + nw * 8].reshape(-1, 8)[:, :7].astype(np.float64)
            if ph.any():
                per = ph.sum(axis=0) / wt[:, 2].sum()
                print("   async stamps, shader cycles per pass: " + ", ".join(f"{n} {v:.0f}" for n, v in zip(
                    ["retire/restart", "start (both set-ups)", "inversion segment + attempt", "attempt outside the triangle", "explicit", "squeeze", "exact redo"], per)) + f"; sum {per.sum():.0f}", flush=True)
        if bs.n_tiles:
            wt = raw[: bs.n_tiles * 4].reshape(-1, 4)
            lanes = (bs.slot_K.reshape(bs.n_tiles, 64) > 0).sum(axis=1)
            real = lanes > 0                      # (tiles whose only chain runs as a chain wave leave no record here)
            dur = ((wt[:, 1] - wt[:, 0]) / 1e8)[real]
            kmax = np.diff(bs.tile_ptr)[real]
            ln = lanes[real]
            print(f"   tile kernel: {real.sum()} lock-step tiles, span {(wt[real, 1].max() - wt[real, 0].min()) / 1e8:.3f} s, longest wave {dur.max():.3f} s, "
                  f"sum of wave time {dur.sum():.1f} s", flush=True)
            for lo, hi in ((2, 8), (8, 24), (24, 48), (48, 65)):
                sel = (ln >= lo) & (ln < hi)
                if sel.any():
                    print(f"     tiles of [{lo},{hi}) lanes: {sel.sum()} (bins {kmax[sel].min()}..{kmax[sel].max()}): us per (nominal) step median "
                          f"{np.median(dur[sel] / (kmax[sel] * B)) * 1e6:.2f}; wave s median {np.median(dur[sel]):.3f} max {dur[sel].max():.3f}", flush=True)
            if engine.TILE_FREE:
                made = wt[real, 2].astype(np.float64)
                print(f"     free-running tiles: bin steps made / nominal, median {np.median(made / (kmax * B)):.3f} (wide tiles {np.median((made / (kmax * B))[ln >= 48]):.3f}); "
                      f"us per step MADE, wide tiles {np.median((dur / made)[ln >= 48]) * 1e6:.2f}", flush=True)
            top = np.argsort(-dur)[:6]
            print("     longest tiles (s, lanes, bins): " + ", ".join(f"({dur[i]:.3f}, {ln[i]}, {kmax[i]})" for i in top), flush=True)


if __name__ == "__main__":
    main()
