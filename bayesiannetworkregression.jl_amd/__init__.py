"""MI355X-native Gibbs hot path of BayesianNetworkRegression.jl behind the reference's Fit!/generate_samples! API.

Compute lives in libbnr_hip.so (hand-written gfx950 HIP kernels, C ABI in include/bnr_hip.h); this package is the
host-side mirror of the reference interface."""
from ._build import build, LIB                                    # noqa: F401
from ._capi import BnrError, Chain, Comm, Group, XInput, device_count, device_synchronize, runtime_version, ess_from_stats, new_table, rhat_from_stats, lib, EXPORTS   # noqa: F401
from .api import (BNRSummary, ChainSet, Fit, Results, Summary, create_lower_tri, device_summary, generate_samples,   # noqa: F401
                  generate_samples_dbl, initialize_and_run, lower_triangle, return_psrf_VOI, run, setup_X,
                  allgather_stats, local_chain_ids, make_comm, shared_seed)
from .synthetic import make_synthetic                             # noqa: F401
