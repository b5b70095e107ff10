"""Writes tests/golden/h5ad_*.h5ad + *_truth.npz: small AnnData files in the on-disk layout anndata uses, written with the REAL
HDF5 library through h5py -- which this image only has under /opt/conda (python3.9), not in the interpreter the package runs on:
    /opt/conda/bin/python3.9 tests/golden/make_h5ad_fixture.py
(anndata itself is not installed anywhere here; the layout follows its format specification, encoding versions 0.2.0 and 0.1.0.)
The files are DATA for tests/test_h5ad.py, which reads them with the package's own ctypes binding of libhdf5."""
import os
import h5py
import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(20250117)
vstr = h5py.string_dtype("utf-8")


def counts(n, g, dens):
    X = sp.random(n, g, density=dens, random_state=np.random.RandomState(3), format="csr", dtype=np.float64)
    X.data = np.ceil(X.data * 6)
    X.sort_indices()
    return X


def write_v2(path, X, obs, var_names, layout="csr", dtype=np.float32, idx_dtype=np.int32, compress=True):
    """encoding 0.2.0: categorical columns are groups {codes, categories}; strings variable-length UTF-8; booleans enums."""
    kw = dict(compression="gzip", chunks=True) if compress else {}
    with h5py.File(path, "w") as f:
        f.attrs["encoding-type"], f.attrs["encoding-version"] = "anndata", "0.1.0"
        M = X.tocsr() if layout == "csr" else X.tocsc()
        g = f.create_group("X")
        g.attrs["encoding-type"], g.attrs["encoding-version"] = layout + "_matrix", "0.1.0"
        g.attrs["shape"] = np.array(X.shape, dtype=np.int64)
        g.create_dataset("data", data=M.data.astype(dtype), **kw)
        g.create_dataset("indices", data=M.indices.astype(idx_dtype), **kw)
        g.create_dataset("indptr", data=M.indptr.astype(idx_dtype), **kw)
        for name, index, cols in (("obs", [f"cell{i}" for i in range(X.shape[0])], obs), ("var", var_names, {})):
            d = f.create_group(name)
            d.attrs["encoding-type"], d.attrs["encoding-version"] = "dataframe", "0.2.0"
            d.attrs["_index"] = "_index"
            d.attrs.create("column-order", np.array(list(cols), dtype=object), dtype=vstr) if cols else d.attrs.create(
                "column-order", np.array([], dtype=object), dtype=vstr)
            d.create_dataset("_index", data=np.array(index, dtype=object), dtype=vstr)
            for c, v in cols.items():
                if isinstance(v, tuple):                      # (codes, categories)
                    cg = d.create_group(c)
                    cg.attrs["encoding-type"], cg.attrs["encoding-version"], cg.attrs["ordered"] = "categorical", "0.2.0", False
                    cg.create_dataset("codes", data=v[0].astype(np.int8))
                    cg.create_dataset("categories", data=np.array(v[1], dtype=object), dtype=vstr)
                elif v.dtype == object:
                    d.create_dataset(c, data=v, dtype=vstr)
                else:
                    d.create_dataset(c, data=v)               # numpy bool -> HDF5 enum, as anndata stores it


def write_v1(path, X, obs, var_names):
    """encoding 0.1.0: categorical = codes dataset + __categories/<col>; fixed-length strings; dense X; no compression."""
    with h5py.File(path, "w") as f:
        f.create_dataset("X", data=X.toarray().astype(np.float32))
        for name, index, cols in (("obs", [f"c{i}" for i in range(X.shape[0])], obs), ("var", var_names, {})):
            d = f.create_group(name)
            d.attrs["_index"] = "names"
            d.attrs["column-order"] = np.array(list(cols), dtype="S16")
            d.create_dataset("names", data=np.array(index, dtype="S24"))
            for c, v in cols.items():
                if isinstance(v, tuple):
                    d.create_dataset(c, data=v[0].astype(np.int8))
                    d.require_group("__categories").create_dataset(c, data=np.array(v[1], dtype="S12"))
                elif v.dtype == object:
                    d.create_dataset(c, data=v.astype("S12"))
                else:
                    d.create_dataset(c, data=v)


def truth(path, X, obs, var_names):
    out = dict(indptr=X.indptr.astype(np.int64), indices=X.indices.astype(np.int32), data=X.data.astype(np.float32), shape=np.array(X.shape),
               var_names=np.array(var_names))
    for c, v in obs.items():
        out["obs_" + c] = np.array(v[1])[v[0]] if isinstance(v, tuple) else (v.astype("U") if v.dtype == object else v)
    np.savez_compressed(path, **out)


n, g = 1500, 90
X = counts(n, g, 0.12)
obs = {"q": np.full(n, 0.07), "stim": (rng.integers(0, 2, n), ["ctrl", "stim"]), "ind": (rng.integers(0, 4, n), ["d1", "d2", "d3", "d4"]),
       "n_counts": np.asarray(X.sum(axis=1)).ravel().astype(np.int64), "keep": rng.random(n) < 0.5,
       "label": np.array([f"b{i % 7}" for i in range(n)], dtype=object)}
var_names = [f"GENE{i}" for i in range(g)]
write_v2(os.path.join(HERE, "h5ad_csr_v2.h5ad"), X, obs, var_names)
truth(os.path.join(HERE, "h5ad_csr_v2_truth.npz"), X, obs, var_names)
write_v2(os.path.join(HERE, "h5ad_csc_f64.h5ad"), X, obs, var_names, layout="csc", dtype=np.float64, idx_dtype=np.int64, compress=False)
n2, g2 = 300, 40
X2 = counts(n2, g2, 0.2)
obs2 = {"q": np.full(n2, 0.1), "grp": (rng.integers(0, 3, n2), ["a", "b", "c"]), "tag": np.array([f"t{i % 3}" for i in range(n2)], dtype=object)}
vn2 = [f"g{i}" for i in range(g2)]
write_v1(os.path.join(HERE, "h5ad_dense_v1.h5ad"), X2, obs2, vn2)
truth(os.path.join(HERE, "h5ad_dense_v1_truth.npz"), X2, obs2, vn2)
print("written")
