"""The C-ABI stands on its own: a process that imports neither torch nor this package drives
mm_csr_rowsum -> mm_sell_split_count / _layout / _scatter_ranges (the product's ingest) -> mm_moments1d_sell -> mm_moments1d_reduce with
nothing but ctypes, numpy and
the library's own mm_malloc / mm_memcpy_* helpers (include/memento_hip.h:41-47) -- exactly what INTEGRATION.md's reference-side
stub does -- and gets the oracle's moment sums."""

import json
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes, json, sys
import numpy as np
assert "torch" not in sys.modules
lib = ctypes.CDLL(sys.argv[1])
lib.mm_last_error.restype = ctypes.c_char_p
V = ctypes.c_void_p
def ck(rc):
    if rc != 0:
        raise RuntimeError(lib.mm_last_error().decode())
def dmalloc(nbytes):
    p = V(); ck(lib.mm_malloc(ctypes.byref(p), ctypes.c_size_t(max(nbytes, 16)))); return p
def to_dev(a):
    a = np.ascontiguousarray(a); p = dmalloc(a.nbytes)
    if a.nbytes:
        ck(lib.mm_memcpy_h2d(p, a.ctypes.data_as(V), ctypes.c_size_t(a.nbytes), None))
    return p
def zeros_dev(nbytes):
    p = dmalloc(nbytes); ck(lib.mm_memset(p, 0, ctypes.c_size_t(max(nbytes, 16)), None)); return p
def to_host(p, shape, dtype):
    out = np.empty(shape, dtype=dtype)
    ck(lib.mm_sync(None))
    if out.nbytes:
        ck(lib.mm_memcpy_d2h(out.ctypes.data_as(V), p, ctypes.c_size_t(out.nbytes), None))
    return out
i32, i64 = ctypes.c_int32, ctypes.c_int64
d = np.load(sys.argv[2])
indptr, indices, data, gid, inv_sf = d["indptr"].astype(np.int64), d["indices"].astype(np.int32), d["data"].astype(np.float32), d["gid"], d["inv_sf"]
N, G, ng = len(indptr) - 1, int(d["G"]), int(d["ng"])
assert lib.mm_device_count() >= 1
d_ip, d_ix, d_dt = to_dev(indptr), to_dev(indices), to_dev(data)
# K3: row sums
d_rs = dmalloc(N * 8)
ck(lib.mm_csr_rowsum(d_ip, d_ix, d_dt, i64(N), None, d_rs, None))
rowsum = to_host(d_rs, (N,), np.float64)
# host plan: cells ordered by group, one block per group (every group here has <= 8192 cells)
order = np.argsort(gid, kind="stable").astype(np.int32)
counts = np.bincount(gid, minlength=ng)
blk_cell0 = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
nb, ns = ng, (G + 63) // 64
d_order, d_bc0 = to_dev(order), to_dev(blk_cell0)
n_sel, R = len(order), (G + 1023) // 1024          # ranges of MM_RANGE_GENES = 1024 gene ids
d_cnt, d_status = zeros_dev(nb * G * 2 + 4), zeros_dev(4)
d_split = dmalloc((R + 1) * n_sel * 8)
ck(lib.mm_sell_split_count(d_ip, d_ix, d_order, d_bc0, i32(nb), i64(n_sel), i32(G), i32(R), d_split, d_cnt, d_status, None))
assert to_host(d_status, (1,), np.int32)[0] == 0
d_rank, d_perm, d_sw = dmalloc(nb * G * 4), dmalloc(nb * ns * 64 * 4), dmalloc(nb * ns * 4)
d_sp, d_itp, d_rows, d_items = dmalloc(nb * (ns + 1) * 4), dmalloc(nb * (ns + 1) * 4), dmalloc(nb * 8), dmalloc(nb * 4)
ck(lib.mm_sell_layout(d_cnt, i32(nb), i32(G), d_rank, d_perm, d_sw, d_sp, d_itp, d_rows, d_items, None))
rows, items = to_host(d_rows, (nb,), np.int64), to_host(d_items, (nb,), np.int32).astype(np.int64)
base = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
ibase = np.concatenate([[0], np.cumsum(items)]).astype(np.int64)
d_base, d_ibase = to_dev(base[:-1]), to_dev(ibase[:-1])
d_ent = zeros_dev(int(base[-1]) * 1024)
ck(lib.mm_sell_scatter_ranges(d_ip, d_ix, d_dt, d_order, d_bc0, i32(nb), i32(G), i32(R), i64(n_sel), d_split, d_rank, d_sp, d_base, d_ent,
                              d_status, None))
assert to_host(d_status, (1,), np.int32)[0] == 0
d_inv = to_dev(inv_sf[order].astype(np.float64))
d_slab = dmalloc(int(ibase[-1]) * 64 * 32)
ck(lib.mm_moments1d_sell(d_ent, d_base, d_sw, d_sp, d_itp, d_ibase, d_bc0, d_inv, i32(nb), i32(G), d_slab, None))
d_gb0 = to_dev(np.arange(ng + 1, dtype=np.int32))
d_S, d_sx, d_mx = dmalloc(3 * ng * G * 8), dmalloc(ng * G * 8), dmalloc(ng * G * 4)
ck(lib.mm_moments1d_reduce(d_slab, d_rank, d_itp, d_ibase, d_gb0, i32(ng), i32(G), d_S, d_sx, d_mx, None))
S, sx, mx = to_host(d_S, (3, ng, G), np.float64), to_host(d_sx, (ng, G), np.uint64), to_host(d_mx, (ng, G), np.uint32)
np.savez(sys.argv[3], rowsum=rowsum, S=S, sx=sx, mx=mx)
for p in (d_ip, d_ix, d_dt, d_rs, d_order, d_bc0, d_cnt, d_status, d_split, d_rank, d_perm, d_sw, d_sp, d_itp, d_rows, d_items, d_base, d_ibase, d_ent, d_inv,
          d_slab, d_gb0, d_S, d_sx, d_mx):
    ck(lib.mm_free(p))
print(json.dumps({"torch_loaded": "torch" in sys.modules, "version": lib.mm_version()}))
'''


def test_cabi_without_torch(tmp_path):
    from oracle import memento_oracle as orc

    rng = np.random.default_rng(12)
    N, G, ng = 3000, 150, 3
    X = sp.csr_matrix(rng.poisson(0.5, size=(N, G)).astype(np.float32))
    gid = rng.integers(0, ng, size=N).astype(np.int64)
    sf = rng.lognormal(0, 0.3, size=N)
    np.savez(tmp_path / "in.npz", indptr=X.indptr, indices=X.indices, data=X.data, gid=gid, inv_sf=1.0 / sf, G=G, ng=ng)
    child = tmp_path / "child.py"
    child.write_text(CHILD)
    lib = os.path.join(ROOT, "scrna_parameter_estimation_amd", "csrc", "libmemento_hip.so")
    r = subprocess.run([sys.executable, str(child), lib, str(tmp_path / "in.npz"), str(tmp_path / "out.npz")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["torch_loaded"] is False
    out = np.load(tmp_path / "out.npz")
    np.testing.assert_array_equal(out["rowsum"], np.asarray(X.sum(axis=1)).ravel())
    X64 = X.astype(np.float64).tocsc()
    for k in range(ng):
        sel = np.flatnonzero(gid == k)
        w = 1.0 / sf[sel]
        np.testing.assert_allclose(out["S"][0, k], X64[sel].T.dot(w), rtol=1e-12)
        np.testing.assert_allclose(out["S"][1, k], X64[sel].power(2).T.dot(w ** 2), rtol=1e-12)
        np.testing.assert_allclose(out["S"][2, k], X64[sel].T.dot(w ** 2), rtol=1e-12)
        np.testing.assert_array_equal(out["sx"][k], np.asarray(X64[sel].sum(axis=0)).ravel().astype(np.uint64))
        np.testing.assert_array_equal(out["mx"][k], np.asarray(X64[sel].max(axis=0).todense()).ravel().astype(np.uint32))
        m_ref, v_ref = orc.moments_1d_sparse(X64[sel], sf[sel], 0.1)
        n = len(sel)
        mean = out["S"][0, k] / n
        var = out["S"][1, k] / n - 0.9 * out["S"][2, k] / n - mean ** 2
        np.testing.assert_allclose(mean, m_ref, rtol=1e-12)
        np.testing.assert_allclose(var, v_ref, rtol=1e-9, atol=1e-13)
