import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def api_small():
    return load_golden("api_small")


@pytest.fixture(scope="session")
def api_approx():
    return load_golden("api_approx")


@pytest.fixture(scope="session")
def api_meanonly():
    return load_golden("api_meanonly")


@pytest.fixture(scope="session")
def api_perm():
    return load_golden("api_perm")


@pytest.fixture(scope="session")
def api_tfg2d():
    return load_golden("api_tfg2d")


@pytest.fixture(scope="session")
def api_opts():
    return load_golden("api_opts")


@pytest.fixture(scope="session")
def internals_small():
    return load_golden("internals_small")


@pytest.fixture(scope="session")
def regress2d_rr():
    return load_golden("regress2d_rr")


@pytest.fixture(scope="session")
def regress_asl():
    return load_golden("regress_asl")


@pytest.fixture(scope="session")
def corrmat_negvar():
    return load_golden("corrmat_negvar")


@pytest.fixture(scope="session")
def api_rr16():
    return load_golden("api_rr16")


@pytest.fixture(scope="session")
def api_c1():
    return load_golden("api_c1")


@pytest.fixture(scope="session")
def guide_loop():
    return load_golden("guide_loop")


@pytest.fixture(scope="session")
def guide_loop_small():
    return load_golden("guide_loop_small")


def golden_inputs(g):
    """Rebuild (X csr float64, group_id, n_groups, q) from a golden api_* fixture."""
    import scipy.sparse as sp

    X = sp.csr_matrix((g["in_data"], g["in_indices"], g["in_indptr"]), shape=tuple(g["in_shape"]))
    labels = np.array([f"sg^{c}^{r}" for c, r in zip(g["in_cond"], g["in_rep"])])
    groups = list(g["groups"])
    gid = np.array([groups.index(l) for l in labels], dtype=np.int32)
    return X, gid, len(groups), g["in_q"]


def c1_inputs(g):
    """(X csr float64, group_id, n_groups, q) of the api_c1 fixture (BASELINE configs[0] shape; compact integer storage)."""
    import scipy.sparse as sp

    X = sp.csr_matrix((g["in_data"].astype(np.float64), g["in_indices"].astype(np.int32), g["in_indptr"].astype(np.int32)),
                      shape=tuple(g["in_shape"]))
    labels = np.array([f"sg^{c}" for c in g["in_cond"]])
    groups = list(g["groups"])
    gid = np.array([groups.index(l) for l in labels], dtype=np.int32)
    return X, gid, len(groups), float(g["in_q"])
