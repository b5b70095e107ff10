"""HBM traffic of the K1 moments kernel (and the K0 ingest kernels) from rocprofv3 PMC counters, per round.

GPU box, two separate counter passes (FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md 'rocprofv3 PMC slots'):
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/k1_traffic.py run
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 tools/k1_traffic.py run
  python3 tools/k1_traffic.py parse gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/r02_k1_traffic_C3.json
gfx950 correction (same guide, section HBM): FETCH_SIZE reports exactly 1/2 of a 16-B/lane coalesced streaming read -> x2 for K1's
dwordx4 stream; WRITE_SIZE is exact for 16-B-per-lane streaming stores.  The ingest kernels' reads are 4-B-per-lane: their
FETCH_SIZE is reported raw AND doubled (uncalibrated width, the guide says so)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPE = (1_000_000, 20_000, 0.03, 20)     # BASELINE.json configs[2] on one GPU


def run():
    import numpy as np
    import torch

    import bench
    from scrna_parameter_estimation_amd import engine

    cells, genes, dens, groups = SHAPE
    csr = bench.synth_device_csr(dict(cells=cells, genes=genes, density=dens), 20250117, torch)
    gid = np.random.default_rng(20250117).integers(0, groups, size=cells).astype(np.int32)
    blocks = engine.CountBlocks(csr, gid, groups)
    d_inv = engine.dev(np.random.default_rng(1).lognormal(0, .3, size=cells)[blocks.cell_order])
    for _ in range(24):
        blocks.launch_moments(d_inv)
    torch.cuda.synchronize()
    print(json.dumps({"nnz": int(csr.nnz), "n_blocks": int(blocks.n_blocks), "algorithmic_bytes_per_launch": int(blocks.moments_bytes()),
                      "ent_bytes": int(blocks.ent_bytes), "csr_bytes": int(csr.nbytes)}), flush=True)


def _collect(d, counter):
    per = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"].split("(")[0]
            per.setdefault(name, []).append(float(row["Counter_Value"]))
    return per


def parse(d_fetch, d_write, out):
    fetch, write = _collect(d_fetch, "FETCH_SIZE"), _collect(d_write, "WRITE_SIZE")
    res = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 tools/k1_traffic.py run (two separate passes)",
           "workload": "1,000,000 cells x 20,000 genes, 3% nnz, 20 groups (BASELINE configs[2] on one GPU)",
           "correction": "gfx950: FETCH_SIZE reports 1/2 of a 16-B/lane coalesced streaming read (MI355X_MICROARCH.md, HBM) -> x2 for K1; WRITE_SIZE exact",
           "kernels": {}}
    for name in sorted(set(fetch) | set(write)):
        f, w = fetch.get(name, []), write.get(name, [])
        if not any(k in name for k in ("k_moments1d_sell", "k_sell_", "k_csr_")):
            continue
        fk = sum(f) / len(f) if f else None
        wk = sum(w) / len(w) if w else None
        res["kernels"][name] = {"dispatches": [len(f), len(w)], "FETCH_SIZE_KB_raw": fk, "WRITE_SIZE_KB_raw": wk}
    k1 = next((v for k, v in res["kernels"].items() if "k_moments1d_sell" in k), None)
    if k1 and k1["FETCH_SIZE_KB_raw"] is not None and k1["WRITE_SIZE_KB_raw"] is not None:
        rd, wr = int(k1["FETCH_SIZE_KB_raw"] * 1024 * 2), int(k1["WRITE_SIZE_KB_raw"] * 1024)
        res.update(kernel="k_moments1d_sell", hbm_read_bytes=rd, hbm_write_bytes=wr, hbm_bytes_per_launch=rd + wr)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if sys.argv[1:2] == ["run"]:
        run()
    elif sys.argv[1:2] == ["parse"]:
        parse(*sys.argv[2:5])
    else:
        print(__doc__)
