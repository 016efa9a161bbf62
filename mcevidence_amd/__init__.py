"""mcevidence_amd -- MI355X-native kNN Bayesian-evidence estimator.

Drop-in for yabebalFantaye/MCEvidence's ``MCEvidence`` class: same constructor and
``.evidence()`` API; the scikit-learn neighbour search and the volume/weight
reduction inside ``evidence()`` run as hand-written gfx950 HIP kernels behind a C ABI
(``include/mcevidence_hip.h``).  See DESIGN.md.
"""
from .chains import MCSamples
from .evidence import HipBackend, MCEvidence, evidence_many
from .prior import cosmo_params_list, get_prior_volume, iscosmo_param, params_info

__all__ = ["MCEvidence", "evidence_many", "MCSamples", "HipBackend", "params_info", "get_prior_volume", "iscosmo_param",
           "cosmo_params_list"]
__version__ = "0.1.0"
