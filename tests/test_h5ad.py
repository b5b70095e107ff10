"""h5ad ingestion (scrna_parameter_estimation_amd/h5ad.py).  The fixtures under tests/golden/h5ad_* were written
with the real HDF5 library through h5py (tests/golden/make_h5ad_fixture.py, build container's conda python) in anndata's on-disk
layout; here they are read back with the package's ctypes binding of libhdf5 and compared with the arrays that were written."""
import os

import numpy as np
import pandas as pd
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _h5ad():
    from scrna_parameter_estimation_amd import h5ad

    try:
        h5ad._load()
    except h5ad.Hdf5LibraryMissing as e:           # the image keeps libhdf5 under /opt/conda/lib; a box without it cannot read HDF5
        pytest.skip(str(e))
    return h5ad


def _check_obs(adata, truth, cats):
    for k in truth.files:
        if not k.startswith("obs_"):
            continue
        col = adata.obs[k[4:]]
        want = truth[k]
        if k[4:] in cats:
            assert isinstance(col.dtype, pd.CategoricalDtype) and list(col.cat.categories) == cats[k[4:]]
            np.testing.assert_array_equal(np.asarray(col.astype(str)), want.astype(str))
        elif want.dtype.kind in "US":
            np.testing.assert_array_equal(np.asarray(col).astype(str), want.astype(str))
        else:
            np.testing.assert_array_equal(np.asarray(col), want)
            assert np.asarray(col).dtype.kind == want.dtype.kind


@pytest.mark.parametrize("name,cats", [("h5ad_csr_v2", {"stim": ["ctrl", "stim"], "ind": ["d1", "d2", "d3", "d4"]}),
                                       ("h5ad_dense_v1", {"grp": ["a", "b", "c"]})])
def test_read_h5ad_on_the_host(name, cats):
    h5ad = _h5ad()
    truth = np.load(os.path.join(GOLD, name + "_truth.npz"))
    adata = h5ad.read_h5ad(os.path.join(GOLD, name + ".h5ad"), to_device=False)
    X = adata.X
    assert X.shape == tuple(truth["shape"]) and X.has_sorted_indices
    np.testing.assert_array_equal(X.indptr, truth["indptr"])
    np.testing.assert_array_equal(X.indices, truth["indices"])
    np.testing.assert_array_equal(X.data, truth["data"])
    assert list(adata.var.index) == list(truth["var_names"]) and len(adata.obs) == X.shape[0]
    assert list(adata.obs.index[:2]) == (["cell0", "cell1"] if name == "h5ad_csr_v2" else ["c0", "c1"])
    _check_obs(adata, truth, cats)


def test_read_h5ad_csc_float64_int64_gives_the_same_csr():
    h5ad = _h5ad()
    truth = np.load(os.path.join(GOLD, "h5ad_csr_v2_truth.npz"))
    X = h5ad.read_h5ad(os.path.join(GOLD, "h5ad_csc_f64.h5ad"), to_device=False).X
    np.testing.assert_array_equal(X.indptr, truth["indptr"])
    np.testing.assert_array_equal(X.indices, truth["indices"])
    np.testing.assert_array_equal(X.data, truth["data"].astype(np.float64))


def test_read_h5ad_errors():
    h5ad = _h5ad()
    with pytest.raises(IOError):
        h5ad.read_h5ad(os.path.join(GOLD, "h5ad_csr_v2_truth.npz"), to_device=False)      # not an HDF5 file
    with pytest.raises(IOError):
        h5ad.read_h5ad(os.path.join(GOLD, "no_such_file.h5ad"), to_device=False)


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", [256 << 20, 1 << 12])
def test_read_h5ad_straight_to_the_device(chunk):
    """The counts go file -> pinned chunk -> HBM (several chunks with the tiny chunk size), the API runs on the resident matrix and
    gives what the host-loaded matrix gives."""
    from scrna_parameter_estimation_amd import engine, memento

    h5ad = _h5ad()
    path = os.path.join(GOLD, "h5ad_csr_v2.h5ad")
    truth = np.load(os.path.join(GOLD, "h5ad_csr_v2_truth.npz"))
    ad = h5ad.read_h5ad(path, to_device=True, chunk_bytes=chunk)
    d = ad.device_csr
    assert ad.X.nnz == 0 and ad.X.shape == tuple(truth["shape"]) and d.shape == tuple(truth["shape"])
    np.testing.assert_array_equal(engine.host(d.indptr), truth["indptr"])
    np.testing.assert_array_equal(engine.host(d.indices), truth["indices"])
    np.testing.assert_array_equal(engine.host(d.data), truth["data"])
    # the CSC / float64 / int64 file takes the host detour and ends on the device as the same matrix
    d2 = h5ad.read_h5ad(os.path.join(GOLD, "h5ad_csc_f64.h5ad"), to_device=True).device_csr
    np.testing.assert_array_equal(engine.host(d2.indices), truth["indices"])
    np.testing.assert_array_equal(engine.host(d2.data), truth["data"])
    host = h5ad.read_h5ad(path, to_device=False)
    out = []
    for a in (ad, host):
        memento.setup_memento(a, q_column="q", filter_mean_thresh=0.01)
        memento.create_groups(a, label_columns=["stim", "ind"])
        memento.compute_1d_moments(a, min_perc_group=0.5, filter_genes=False)
        m = a.uns["memento"]
        out.append((a.obs["memento_size_factor"].values.copy(), [m["1d_moments"][g][0].copy() for g in m["groups"]], list(m["groups"])))
    np.testing.assert_array_equal(out[0][0], out[1][0])
    assert out[0][2] == out[1][2] and len(out[0][2]) == 8
    for x, y in zip(out[0][1], out[1][1]):
        np.testing.assert_array_equal(x, y)
