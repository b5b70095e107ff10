"""Drop-in for the reference's ``memento`` package API (/root/reference/memento/__init__.py:1), HIP-backed.

    from scrna_parameter_estimation_amd import memento
    memento.setup_memento(adata, q_column='q'); memento.create_groups(adata, ['stim', 'ind'])
    memento.compute_1d_moments(adata); memento.ht_1d_moments(adata, covariate=..., treatment=..., resampling='bootstrap')
"""

from .main import (setup_memento, create_groups, compute_1d_moments, compute_2d_moments, ht_1d_moments,  # noqa: F401
                   ht_2d_moments, get_1d_moments, get_2d_moments, get_1d_ht_result, get_2d_ht_result, prepare_to_save,
                   get_corr_matrix, get_groups, ht_1d_vs_control)
from . import simulate  # noqa: F401,E402  (reference: memento/main.py:23 imports memento.simulate as well)
