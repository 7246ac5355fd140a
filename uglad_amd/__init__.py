"""uglad_amd: the unrolled-GLAD hot path of Harshs27/uGLAD as hand-written gfx950 (MI355X) HIP kernels behind uGLAD's own
Python surface.  Layout mirrors the reference package: `uglad_amd.main` (uGLAD_GL, uGLAD_multitask, forward_uGLAD,
loss_uGLAD, run_uGLAD_*), `uglad_amd.glad.glad` (glad), `uglad_amd.glad.glad_params` (GladParams),
`uglad_amd.utils.prepare_data` / `.metrics`.  Importing is free of GPU work; the first kernel call loads
csrc/libuglad_hip.so and raises if it (or a GPU) is missing -- there is no CPU fallback.
"""
from .glad.glad_params import GladParams  # noqa: F401
from .glad.glad import glad, get_optimizers, batch_symeig, regime_monitor, UgladRegimeWarning  # noqa: F401
from .main import (  # noqa: F401
    uGLAD_GL,
    uGLAD_multitask,
    forward_uGLAD,
    loss_uGLAD,
    init_uGLAD,
    run_uGLAD_direct,
    run_uGLAD_CV,
    run_uGLAD_missing,
    run_uGLAD_multitask,
    get_final_precision_from_batch,
    mean_imputation,
    get_partial_correlations,
    device_report_metrics,
    conditional_gaussian_batch,
    conditional_gaussian_with_probabilities,
    compute_map_estimate,
    save_uGLAD_model,
    load_uGLAD_model,
)

__version__ = "0.1.0"
