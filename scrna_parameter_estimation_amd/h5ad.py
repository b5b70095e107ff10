"""``.h5ad`` ingestion straight to the device: the counterpart of ``scanpy.read_h5ad`` for what
memento touches -- ``X`` (CSR, CSC or dense), ``obs`` and ``var`` -- every caller of the reference starts with it
(e.g. /root/reference/analysis/lupus/run_memento.py:23, the tutorials' ``sc.read``).

There is no h5py / anndata in the image, but the HDF5 C library is (``libhdf5.so``): this module binds the two dozen C calls it needs
with ctypes.  The count matrix never exists as a scipy object on the way: ``X/data``, ``X/indices``, ``X/indptr`` are read in
hyperslab chunks into one pinned staging buffer and copied to HBM chunk by chunk (converted to float32 / int32 / int64 on the
device), giving an ``engine.DeviceCSR`` that ``setup_memento`` takes as it is.

On-disk layout followed (anndata's format specification, encoding versions 0.1.0 / 0.2.0):
  X            group, attrs ``encoding-type`` in {csr_matrix, csc_matrix}, ``shape``; datasets data / indices / indptr -- or a dense dataset
  obs, var     group, attrs ``_index`` (name of the index dataset), ``column-order``; one member per column:
               a dataset (numbers, booleans as HDF5 enums, strings), a categorical group {codes, categories} (0.2.0), a dataset
               with companion ``__categories/<column>`` (0.1.0), or a nullable group {values, mask}
"""

import ctypes
import ctypes.util
import glob
import os

import numpy as np
import pandas as pd
import scipy.sparse as sp

hid_t = ctypes.c_int64
hsize_t = ctypes.c_uint64
_H5T_INTEGER, _H5T_FLOAT, _H5T_STRING, _H5T_ENUM = 0, 1, 3, 8
_H5S_SELECT_SET = 0
_H5T_VARIABLE = ctypes.c_size_t(-1).value
_LIB = None


class Hdf5LibraryMissing(ImportError):
    pass


def _load():
    """libhdf5 through ctypes: $MM_HDF5_LIB, the loader's search path, then the places this image keeps it."""
    global _LIB
    if _LIB is not None:
        return _LIB
    cands = [os.environ.get("MM_HDF5_LIB"), ctypes.util.find_library("hdf5")]
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5*.so*", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*",
                "/usr/local/lib/libhdf5.so*"):
        cands += sorted(glob.glob(pat), key=len)
    err = None
    for c in cands:
        if not c:
            continue
        try:
            lib = ctypes.CDLL(c)
            lib.H5open()
            break
        except (OSError, AttributeError) as e:       # not loadable / not an HDF5 library
            err = e
    else:
        raise Hdf5LibraryMissing(f"no usable libhdf5 found (set MM_HDF5_LIB); last error: {err}")
    sig = {
        "H5Fopen": (hid_t, [ctypes.c_char_p, ctypes.c_uint, hid_t]), "H5Fclose": (ctypes.c_int, [hid_t]),
        "H5Gopen2": (hid_t, [hid_t, ctypes.c_char_p, hid_t]), "H5Gclose": (ctypes.c_int, [hid_t]),
        "H5Dopen2": (hid_t, [hid_t, ctypes.c_char_p, hid_t]), "H5Dclose": (ctypes.c_int, [hid_t]),
        "H5Oopen": (hid_t, [hid_t, ctypes.c_char_p, hid_t]), "H5Oclose": (ctypes.c_int, [hid_t]),
        "H5Iget_type": (ctypes.c_int, [hid_t]),
        "H5Lexists": (ctypes.c_int, [hid_t, ctypes.c_char_p, hid_t]),
        "H5Aexists": (ctypes.c_int, [hid_t, ctypes.c_char_p]), "H5Aopen": (hid_t, [hid_t, ctypes.c_char_p, hid_t]),
        "H5Aclose": (ctypes.c_int, [hid_t]), "H5Aget_type": (hid_t, [hid_t]), "H5Aget_space": (hid_t, [hid_t]),
        "H5Aread": (ctypes.c_int, [hid_t, hid_t, ctypes.c_void_p]),
        "H5Dget_type": (hid_t, [hid_t]), "H5Dget_space": (hid_t, [hid_t]),
        "H5Dread": (ctypes.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, ctypes.c_void_p]),
        "H5Dvlen_reclaim": (ctypes.c_int, [hid_t, hid_t, hid_t, ctypes.c_void_p]),
        "H5Sget_simple_extent_ndims": (ctypes.c_int, [hid_t]),
        "H5Sget_simple_extent_dims": (ctypes.c_int, [hid_t, ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t)]),
        "H5Sget_simple_extent_npoints": (ctypes.c_int64, [hid_t]),
        "H5Screate_simple": (hid_t, [ctypes.c_int, ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t)]),
        "H5Sselect_hyperslab": (ctypes.c_int, [hid_t, ctypes.c_int, ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t),
                                                ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t)]),
        "H5Sclose": (ctypes.c_int, [hid_t]),
        "H5Tget_class": (ctypes.c_int, [hid_t]), "H5Tget_size": (ctypes.c_size_t, [hid_t]), "H5Tget_sign": (ctypes.c_int, [hid_t]),
        "H5Tis_variable_str": (ctypes.c_int, [hid_t]), "H5Tget_super": (hid_t, [hid_t]), "H5Tcopy": (hid_t, [hid_t]),
        "H5Tset_size": (ctypes.c_int, [hid_t, ctypes.c_size_t]), "H5Tset_cset": (ctypes.c_int, [hid_t, ctypes.c_int]),
        "H5Tclose": (ctypes.c_int, [hid_t]),
        "H5Eset_auto2": (ctypes.c_int, [hid_t, ctypes.c_void_p, ctypes.c_void_p]),
    }
    for name, (res, args) in sig.items():
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    lib.H5Eset_auto2(0, None, None)          # errors come back as negative return values, not as text on stderr
    _LIB = lib
    return lib


def _native(kind, size, signed=True):
    """hid_t of the native memory type (the H5T_NATIVE_* macros are globals filled by H5open) and the numpy dtype."""
    lib = _load()
    if kind == "f":
        name, dt = ("H5T_NATIVE_FLOAT_g", np.float32) if size == 4 else ("H5T_NATIVE_DOUBLE_g", np.float64)
    else:
        name = f"H5T_NATIVE_{'' if signed else 'U'}INT{size * 8}_g"
        dt = np.dtype(f"{'i' if signed else 'u'}{size}")
    return hid_t.in_dll(lib, name).value, np.dtype(dt)


class _Obj:
    """An open group or dataset (closed on exit)."""

    def __init__(self, loc, name):
        self.lib = _load()
        self.id = self.lib.H5Oopen(loc, name.encode(), 0)
        if self.id < 0:
            raise KeyError(f"no object '{name}' in the HDF5 file")
        self.is_dataset = self.lib.H5Iget_type(self.id) == 5          # H5I_DATASET

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.lib.H5Oclose(self.id)

    def has(self, name):
        return self.lib.H5Lexists(self.id, name.encode(), 0) > 0

    def attr(self, name, default=None):
        lib = self.lib
        if lib.H5Aexists(self.id, name.encode()) <= 0:
            return default
        a = lib.H5Aopen(self.id, name.encode(), 0)
        try:
            tid, sid = lib.H5Aget_type(a), lib.H5Aget_space(a)
            out = _read_any(lambda mem, buf: lib.H5Aread(a, mem, buf), tid, sid)
            lib.H5Tclose(tid)
            lib.H5Sclose(sid)
            return out
        finally:
            lib.H5Aclose(a)

    def shape(self):
        lib = self.lib
        sid = lib.H5Dget_space(self.id)
        nd = lib.H5Sget_simple_extent_ndims(sid)
        dims = (hsize_t * max(nd, 1))()
        lib.H5Sget_simple_extent_dims(sid, dims, None)
        lib.H5Sclose(sid)
        return tuple(int(d) for d in dims[:nd])

    def read(self):
        lib = self.lib
        tid, sid = lib.H5Dget_type(self.id), lib.H5Dget_space(self.id)
        out = _read_any(lambda mem, buf: lib.H5Dread(self.id, mem, 0, 0, 0, buf), tid, sid)
        lib.H5Tclose(tid)
        lib.H5Sclose(sid)
        return out

    def numeric_type(self):
        """(memory hid_t, numpy dtype) of a numeric dataset."""
        lib = self.lib
        tid = lib.H5Dget_type(self.id)
        cls, size, sign = lib.H5Tget_class(tid), lib.H5Tget_size(tid), lib.H5Tget_sign(tid)
        lib.H5Tclose(tid)
        if cls not in (_H5T_INTEGER, _H5T_FLOAT):
            raise TypeError("not a numeric dataset")
        return _native("f" if cls == _H5T_FLOAT else "i", size, sign != 0)

    def read_slice_into(self, start, count, buf_ptr, mem_type):
        """Elements [start, start + count) of a 1-D dataset into ``buf_ptr`` (hyperslab selection)."""
        lib = self.lib
        fs = lib.H5Dget_space(self.id)
        st, ct = (hsize_t * 1)(start), (hsize_t * 1)(count)
        lib.H5Sselect_hyperslab(fs, _H5S_SELECT_SET, st, None, ct, None)
        ms = lib.H5Screate_simple(1, ct, None)
        rc = lib.H5Dread(self.id, mem_type, ms, fs, 0, buf_ptr)
        lib.H5Sclose(ms)
        lib.H5Sclose(fs)
        if rc < 0:
            raise IOError("H5Dread failed")


def _read_any(reader, tid, sid):
    """Read a whole attribute / dataset of type ``tid`` and space ``sid``: numbers, enums (booleans), strings."""
    lib = _load()
    nd = lib.H5Sget_simple_extent_ndims(sid)
    dims = (hsize_t * max(nd, 1))()
    if nd > 0:
        lib.H5Sget_simple_extent_dims(sid, dims, None)
    shape = tuple(int(d) for d in dims[:nd])
    n = int(lib.H5Sget_simple_extent_npoints(sid))
    cls, size = lib.H5Tget_class(tid), lib.H5Tget_size(tid)
    if cls in (_H5T_INTEGER, _H5T_FLOAT, _H5T_ENUM):
        base = tid
        if cls == _H5T_ENUM:                                        # anndata stores booleans as an int8 enum {FALSE, TRUE}
            base = lib.H5Tget_super(tid)
        mem, dt = _native("f" if lib.H5Tget_class(base) == _H5T_FLOAT else "i", lib.H5Tget_size(base), lib.H5Tget_sign(base) != 0)
        if cls == _H5T_ENUM:
            lib.H5Tclose(base)
        out = np.empty(max(n, 1), dtype=dt)
        if n and reader(mem, out.ctypes.data_as(ctypes.c_void_p)) < 0:
            raise IOError("HDF5 read failed")
        out = out[:n].reshape(shape)
        if cls == _H5T_ENUM:
            out = out.astype(bool)
        return out[()] if nd == 0 else out
    if cls == _H5T_STRING:
        mem = lib.H5Tcopy(hid_t.in_dll(lib, "H5T_C_S1_g").value)
        if lib.H5Tis_variable_str(tid) > 0:
            lib.H5Tset_size(mem, _H5T_VARIABLE)
            lib.H5Tset_cset(mem, 1)                                 # UTF-8
            ptrs = (ctypes.c_char_p * max(n, 1))()
            if n and reader(mem, ctypes.cast(ptrs, ctypes.c_void_p)) < 0:
                raise IOError("HDF5 read failed")
            vals = [(p.decode("utf-8", "replace") if p is not None else "") for p in ptrs[:n]]
            if n:
                lib.H5Dvlen_reclaim(mem, sid, 0, ctypes.cast(ptrs, ctypes.c_void_p))
        else:
            lib.H5Tset_size(mem, size)
            raw = ctypes.create_string_buffer(max(n, 1) * size)
            if n and reader(mem, ctypes.cast(raw, ctypes.c_void_p)) < 0:
                raise IOError("HDF5 read failed")
            vals = [raw.raw[i * size:(i + 1) * size].split(b"\0", 1)[0].decode("utf-8", "replace") for i in range(n)]
        lib.H5Tclose(mem)
        if nd == 0:
            return vals[0] if vals else ""
        return np.asarray(vals, dtype=object).reshape(shape)
    raise TypeError(f"unsupported HDF5 type class {cls}")


def _as_str(x, default=None):
    if x is None:
        return default
    if isinstance(x, np.ndarray):
        x = x.reshape(-1)[0] if x.size else default
    return x.decode() if isinstance(x, bytes) else x


def _read_column(grp, name):
    """One obs / var column -> 1-D numpy array or pandas Categorical."""
    with _Obj(grp.id, name) as o:
        if o.is_dataset:
            vals = o.read()
            if grp.has("__categories/" + name):                     # encoding 0.1.0: codes + companion categories dataset
                with _Obj(grp.id, "__categories/" + name) as c:
                    cats = c.read()
                return pd.Categorical.from_codes(np.asarray(vals, dtype=np.int64), categories=list(cats))
            return vals
        enc = _as_str(o.attr("encoding-type"), "")
        if enc == "categorical" or (o.has("codes") and o.has("categories")):
            with _Obj(o.id, "codes") as c, _Obj(o.id, "categories") as k:
                codes, cats = c.read(), k.read()
            return pd.Categorical.from_codes(np.asarray(codes, dtype=np.int64), categories=list(cats), ordered=bool(o.attr("ordered", False)))
        if o.has("values") and o.has("mask"):                        # nullable integer / boolean
            with _Obj(o.id, "values") as v, _Obj(o.id, "mask") as k:
                vals, mask = v.read().astype(np.float64), k.read().astype(bool)
            vals[mask] = np.nan
            return vals
        raise TypeError(f"column '{name}': unsupported encoding '{enc}'")


def _read_frame(file_id, name):
    """obs / var group -> DataFrame (index from the dataset the ``_index`` attribute names; columns in ``column-order``)."""
    with _Obj(file_id, name) as g:
        idx_name = _as_str(g.attr("_index"), "_index")
        order = g.attr("column-order")
        cols = [] if order is None else [c.decode() if isinstance(c, bytes) else str(c) for c in np.asarray(order).reshape(-1)]
        with _Obj(g.id, idx_name) as i:
            index = pd.Index([str(v) for v in i.read()], name=None if idx_name == "_index" else idx_name)
        data = {c: _read_column(g, c) for c in cols}
    return pd.DataFrame(data, index=index, columns=cols)


def _csr_to_device(xgrp, shape, chunk_bytes):
    """X/data, X/indices, X/indptr -> engine.DeviceCSR, chunk by chunk through one pinned staging buffer."""
    from . import engine

    torch = engine._torch()
    targets = {"indptr": torch.int64, "indices": torch.int32, "data": torch.float32}
    np_of = {torch.int64: np.int64, torch.int32: np.int32, torch.float32: np.float32}
    stage = torch.empty(max(1 << 20, int(chunk_bytes)), dtype=torch.uint8).pin_memory()
    out = {}
    for name, tdt in targets.items():
        with _Obj(xgrp.id, name) as d:
            (n,) = d.shape()
            mem, ndt = d.numeric_type()
            dev = torch.empty(max(1, n), dtype=tdt, device="cuda")
            per = max(1, stage.numel() // ndt.itemsize)
            view = stage.numpy()[:per * ndt.itemsize].view(ndt)
            for s in range(0, n, per):
                c = min(per, n - s)
                d.read_slice_into(s, c, ctypes.c_void_p(view.ctypes.data), mem)
                src = torch.from_numpy(view[:c])
                dev[s:s + c].copy_(src if ndt == np_of[tdt] else src.to(device="cuda").to(tdt))   # dtype conversion on the device
                torch.cuda.current_stream().synchronize()                                          # the staging buffer is reused
            out[name] = dev[:n]
    if int(out["indptr"].numel()) != shape[0] + 1:
        raise ValueError("X/indptr does not match X's shape")
    return engine.DeviceCSR.from_device(out["indptr"], out["indices"], out["data"], shape)


def read_h5ad(path, to_device=True, chunk_bytes=256 << 20):
    """Read ``X``, ``obs``, ``var`` of an .h5ad file.  Returns an ``AnnDataLite``.

    ``to_device=True`` (needs a GPU): a CSR-encoded ``X`` goes from the file to HBM in chunks of ``chunk_bytes`` and is attached as
    ``adata.device_csr`` (``setup_memento`` picks it up); ``adata.X`` is then an EMPTY scipy CSR of the right shape, as in
    ``setup_memento(..., device_csr=...)``.  Stored zeros are not expected in a count matrix on disk; if the file has them, or X is
    CSC / dense, the matrix is assembled with scipy on the host first and uploaded whole.
    ``to_device=False``: everything on the host (``adata.X`` a scipy CSR) -- no GPU needed."""
    from .anndata_lite import AnnDataLite

    lib = _load()
    fid = lib.H5Fopen(os.fsencode(path), 0, 0)
    if fid < 0:
        raise IOError(f"cannot open '{path}' as an HDF5 file")
    try:
        obs, var = _read_frame(fid, "obs"), _read_frame(fid, "var")
        device_csr = X = None
        with _Obj(fid, "X") as xg:
            if xg.is_dataset:                                        # dense X
                shape = xg.shape()
                X = sp.csr_matrix(np.asarray(xg.read()).reshape(shape))
            else:
                enc = _as_str(xg.attr("encoding-type"), _as_str(xg.attr("h5sparse_format"), "csr_matrix"))
                shp = xg.attr("shape", xg.attr("h5sparse_shape"))
                shape = (int(shp[0]), int(shp[1]))
                if to_device and enc.startswith("csr"):
                    device_csr = _csr_to_device(xg, shape, chunk_bytes)
                    if device_csr.nnz and bool((device_csr.data == 0).any().item()):
                        device_csr = None                            # stored zeros: take the host path below (scipy drops them)
                if device_csr is None:
                    parts = {}
                    for k in ("data", "indices", "indptr"):
                        with _Obj(xg.id, k) as d:
                            parts[k] = d.read()
                    M = (sp.csc_matrix if enc.startswith("csc") else sp.csr_matrix)((parts["data"], parts["indices"], parts["indptr"]), shape=shape)
                    X = sp.csr_matrix(M)
    finally:
        lib.H5Fclose(fid)
    if len(obs) != shape[0] or len(var) != shape[1]:
        raise ValueError("obs / var do not match X's shape")
    if device_csr is None and to_device:
        from . import engine

        device_csr = engine.DeviceCSR(X.astype(np.float32))
    if device_csr is not None:
        adata = AnnDataLite(sp.csr_matrix(shape, dtype=np.float32), obs, var)
        adata.device_csr = device_csr
    else:
        adata = AnnDataLite(X, obs, var)
    return adata
