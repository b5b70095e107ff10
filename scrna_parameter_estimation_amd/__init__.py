"""MI355X-native hot path for memento (method-of-moments scRNA-seq estimation + bootstrap tests).

``from scrna_parameter_estimation_amd import memento`` mirrors the reference's ``memento.*`` API
(/root/reference/memento/__init__.py:1); the compute runs in hand-written HIP kernels for gfx950
behind the C-ABI declared in ``include/memento_hip.h``.
"""

from .anndata_lite import AnnDataLite  # noqa: F401


def read_h5ad(path, to_device=True, **kw):
    """``scanpy.read_h5ad`` for what memento touches, the counts going straight to HBM (see ``h5ad.read_h5ad``)."""
    from .h5ad import read_h5ad as _read

    return _read(path, to_device=to_device, **kw)


__version__ = "0.1.0"
