import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, pandas as pd
from conftest import c1_inputs, load_golden
from oracle import memento_oracle as orc
from scrna_parameter_estimation_amd import engine
from scrna_parameter_estimation_amd.memento import main as M
g = load_golden("api_c1")
X, gid, ng, q = c1_inputs(g)
N, G = X.shape
csr = engine.DeviceCSR(X.astype(np.float32))
naive = csr.rowsum()
naive_ref = np.asarray(X.sum(axis=1)).ravel()
print("naive rowsum max abs diff", np.abs(naive - naive_ref).max())
blocks = engine.CountBlocks(csr, np.zeros(N, dtype=np.int32), 1)
S, sumx, maxx = blocks.moments(1.0 / naive)
m_ref, v_ref = orc.moments_1d_sparse(X, naive_ref, q)
am, av = M._moments_from_sums(S[:, 0], N, q)
print("mean rel diff", np.nanmax(np.abs(am - m_ref) / np.maximum(np.abs(m_ref), 1e-300)), "var abs diff", np.nanmax(np.abs(av - v_ref)))
print("sumx diff", np.abs(sumx[0].astype(float) - np.asarray(X.sum(axis=0)).ravel()).max())
am = am.copy(); am[(sumx[0].astype(np.float64) / N) < 0.07] = 0
mr = m_ref.copy(); mr[np.asarray(X.mean(axis=0)).ravel() < 0.07] = 0
print("zeroed same:", np.array_equal(am == 0, mr == 0))
rv = M._res_var(am, av, M._mv_fit(am, av)); rvr = orc.residual_variance(mr, v_ref, orc.poly_mv_fit(mr, v_ref))
print("fit", M._mv_fit(am, av), orc.poly_mv_fit(mr, v_ref))
print("rv max rel diff", np.nanmax(np.abs(rv - rvr) / np.abs(rvr)))
ul = np.quantile(rv[np.isfinite(rv)], 0.1); ulr = np.quantile(rvr[np.isfinite(rvr)], 0.1)
rv[~np.isfinite(rv)] = np.inf; rvr[~np.isfinite(rvr)] = np.inf
mask, maskr = rv < ul, rvr < ulr
print("mask sizes", mask.sum(), maskr.sum(), "equal", np.array_equal(mask, maskr))
nrc = csr.rowsum(mask); nrcr = np.asarray(X.multiply(maskr).sum(axis=1)).ravel()
print("masked rowsum diff", np.abs(nrc - nrcr).max(), "with same mask:", np.abs(csr.rowsum(maskr) - nrcr).max())
