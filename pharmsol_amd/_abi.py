"""ctypes mirror of ``include/pmx.h`` (struct layouts and enum values).

Shared by the product binding (``_ffi.py`` -> libpmx_hip.so) and by the test
oracle's binding (``oracle/__init__.py`` -> libpmx_oracle.so); it holds no
compute.
"""
from __future__ import annotations

import ctypes as C

PMX_ABI_VERSION = 3
PMX_CENSOR_NONE, PMX_CENSOR_BLOQ, PMX_CENSOR_ALOQ = 0, 1, -1

PMX_MAX_STATES = 8
PMX_MAX_INPUTS = 8
PMX_MAX_OUT = 4
PMX_MAX_KPARAMS = 8
PMX_MAX_DERIVED = 4
PMX_MAX_USER_DERIVED = 16
PMX_MAX_FACTORS = 2
PMX_MAX_PARAMS = 16
PMX_MAX_COVARIATES = 8

# pmx_status
PMX_OK = 0
PMX_ERR_INVALID_ARGUMENT = 1
PMX_ERR_INPUT_OUT_OF_RANGE = 2
PMX_ERR_OUTEQ_OUT_OF_RANGE = 3
PMX_ERR_UNSUPPORTED = 4
PMX_ERR_NO_DEVICE = 5
PMX_ERR_HIP = 6
PMX_ERR_OUT_OF_MEMORY = 7
PMX_ERR_PAIR_FAILED = 8
PMX_ERR_ERROR_MODEL = 9

PMX_PAIR_OK = 0
PMX_PAIR_COMPLEX_ROOTS = 1
PMX_PAIR_NONFINITE = 2
PMX_PAIR_BAD_LAG = 3

PMX_EV_OBSERVATION = 0
PMX_EV_BOLUS = 1
PMX_EV_INFUSION = 2

PMX_EQ_ODE = 0
PMX_EQ_ANALYTICAL = 1

# AnalyticalKernel (pharmsol-dsl/src/analysis.rs:187-200), name -> id
ANALYTICAL_KERNELS = {
    "one_compartment": 0,
    "one_compartment_cl": 1,
    "one_compartment_cl_with_absorption": 2,
    "one_compartment_with_absorption": 3,
    "two_compartments": 4,
    "two_compartments_cl": 5,
    "two_compartments_cl_with_absorption": 6,
    "two_compartments_with_absorption": 7,
    "three_compartments": 8,
    "three_compartments_cl": 9,
    "three_compartments_cl_with_absorption": 10,
    "three_compartments_with_absorption": 11,
}

# AnalyticalKernel::required_parameter_names (analysis.rs:240-255)
KERNEL_PARAMETER_NAMES = {
    "one_compartment": ["ke"],
    "one_compartment_cl": ["cl", "v"],
    "one_compartment_cl_with_absorption": ["ka", "cl", "v"],
    "one_compartment_with_absorption": ["ka", "ke"],
    "two_compartments": ["ke", "kcp", "kpc"],
    "two_compartments_cl": ["cl", "q", "vc", "vp"],
    "two_compartments_cl_with_absorption": ["ka", "cl", "q", "vc", "vp"],
    "two_compartments_with_absorption": ["ke", "ka", "kcp", "kpc"],
    "three_compartments": ["k10", "k12", "k13", "k21", "k31"],
    "three_compartments_cl": ["cl", "q2", "q3", "vc", "v2", "v3"],
    "three_compartments_cl_with_absorption": ["ka", "cl", "q2", "q3", "vc", "v2", "v3"],
    "three_compartments_with_absorption": ["ka", "k10", "k12", "k13", "k21", "k31"],
}

# AnalyticalKernel::state_count (analysis.rs:259-270)
KERNEL_STATE_COUNT = {
    "one_compartment": 1,
    "one_compartment_cl": 1,
    "one_compartment_cl_with_absorption": 2,
    "one_compartment_with_absorption": 2,
    "two_compartments": 2,
    "two_compartments_cl": 2,
    "two_compartments_cl_with_absorption": 3,
    "two_compartments_with_absorption": 3,
    "three_compartments": 3,
    "three_compartments_cl": 3,
    "three_compartments_cl_with_absorption": 4,
    "three_compartments_with_absorption": 4,
}

ODE_MODELS = {
    "one_cmt_iv": 0,
    "one_cmt_oral": 1,
    "two_cmt_iv": 2,
    "two_cmt_oral": 3,
    "three_cmt_iv": 4,
    "three_cmt_oral": 5,
    "one_cmt_mm": 6,
}
PMX_ODE_CUSTOM = 100
PMX_SOLVER_RK4, PMX_SOLVER_DOPRI5, PMX_SOLVER_ROS2 = 0, 1, 2
PMX_PAIR_SOLVER_FAIL = 4
ODE_STATE_COUNT = {"one_cmt_iv": 1, "one_cmt_oral": 2, "two_cmt_iv": 2, "two_cmt_oral": 3, "three_cmt_iv": 3,
                   "three_cmt_oral": 4, "one_cmt_mm": 1}
ODE_PARAM_COUNT = {"one_cmt_iv": 1, "one_cmt_oral": 2, "two_cmt_iv": 3, "two_cmt_oral": 4, "three_cmt_iv": 5,
                   "three_cmt_oral": 6, "one_cmt_mm": 3}

PMX_SRC_NONE = 0
PMX_SRC_PRIMARY = 1
PMX_SRC_DERIVED = 2

PMX_F_NONE = 0
PMX_F_POW = 1
PMX_F_LIN = 2

PMX_COV_TIME_SEGMENT_DT = 0
PMX_COV_TIME_SEGMENT_END_ABS = 1

STATUS_NAMES = {
    PMX_OK: "OK",
    PMX_ERR_INVALID_ARGUMENT: "InvalidArgument",
    PMX_ERR_INPUT_OUT_OF_RANGE: "InputOutOfRange",
    PMX_ERR_OUTEQ_OUT_OF_RANGE: "OuteqOutOfRange",
    PMX_ERR_UNSUPPORTED: "Unsupported",
    PMX_ERR_NO_DEVICE: "NoDevice",
    PMX_ERR_HIP: "HipError",
    PMX_ERR_OUT_OF_MEMORY: "OutOfMemory",
    PMX_ERR_PAIR_FAILED: "PairFailed",
    PMX_ERR_ERROR_MODEL: "ErrorModelError",
}


class pmx_population_desc(C.Structure):
    _fields_ = [
        ("n_subjects", C.c_int64),
        ("n_occasions", C.c_int64),
        ("n_events", C.c_int64),
        ("subj_occ_off", C.POINTER(C.c_int64)),
        ("occ_ev_off", C.POINTER(C.c_int64)),
        ("occ_index", C.POINTER(C.c_int32)),
        ("ev_time", C.POINTER(C.c_double)),
        ("ev_value", C.POINTER(C.c_double)),
        ("ev_duration", C.POINTER(C.c_double)),
        ("ev_kind", C.POINTER(C.c_uint8)),
        ("ev_io", C.POINTER(C.c_uint16)),
        ("n_covariates", C.c_int32),
        ("presorted", C.c_int32),
        ("cov_knot_off", C.POINTER(C.c_int64)),
        ("cov_knot_time", C.POINTER(C.c_double)),
        ("cov_knot_value", C.POINTER(C.c_double)),
        ("cov_fixed", C.POINTER(C.c_uint8)),
        ("ev_errorpoly", C.POINTER(C.c_double)),
        ("ev_censor", C.POINTER(C.c_int8)),
    ]


class pmx_factor(C.Structure):
    _fields_ = [("op", C.c_int32), ("cov", C.c_int32), ("ref", C.c_double), ("coef", C.c_double)]


class pmx_derived(C.Structure):
    _fields_ = [("src_param", C.c_int32), ("n_factors", C.c_int32), ("f", pmx_factor * PMX_MAX_FACTORS)]


class pmx_bind(C.Structure):
    _fields_ = [("src", C.c_int32), ("index", C.c_int32)]


class pmx_out(C.Structure):
    _fields_ = [("state", C.c_int32), ("vol_src", C.c_int32), ("vol_index", C.c_int32), ("reserved", C.c_int32)]


class pmx_model_desc(C.Structure):
    _fields_ = [
        ("eq_kind", C.c_int32),
        ("kernel", C.c_int32),
        ("nstates", C.c_int32),
        ("ndrugs", C.c_int32),
        ("nout", C.c_int32),
        ("nparams", C.c_int32),
        ("n_covariates", C.c_int32),
        ("n_derived", C.c_int32),
        ("derived", pmx_derived * PMX_MAX_DERIVED),
        ("n_bind", C.c_int32),
        ("bind", pmx_bind * PMX_MAX_KPARAMS),
        ("out", pmx_out * PMX_MAX_OUT),
        ("cov_time_mode", C.c_int32),
        ("pmetrics_indexing", C.c_int32),
        ("init_param", C.c_int32 * PMX_MAX_STATES),
        ("lag_param", C.c_int32 * PMX_MAX_INPUTS),
        ("fa_param", C.c_int32 * PMX_MAX_INPUTS),
        ("bolus_dest", C.c_int32 * PMX_MAX_INPUTS),
        ("infusion_dest", C.c_int32 * PMX_MAX_INPUTS),
        ("rk4_h_max", C.c_double),
        ("ode_solver", C.c_int32),
        ("reserved_", C.c_int32),
        ("ode_rtol", C.c_double),
        ("ode_atol", C.c_double),
    ]


PMX_EM_NONE, PMX_EM_ADDITIVE, PMX_EM_PROPORTIONAL = 0, 1, 2
PMX_EM_RES_CONSTANT, PMX_EM_RES_PROPORTIONAL, PMX_EM_RES_COMBINED, PMX_EM_RES_EXPONENTIAL = 3, 4, 5, 6


class pmx_error_model(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("c", C.c_double * 4), ("scalar", C.c_double)]


class pmx_op_stream_view(C.Structure):
    _fields_ = [
        ("n_subjects", C.c_int64), ("n_ops", C.c_int64),
        ("n_cov", C.c_int32), ("n_rate", C.c_int32),
        ("max_input_used", C.c_int32), ("max_outeq", C.c_int32),
        ("subj_op_off", C.POINTER(C.c_int64)),
        ("op_meta", C.POINTER(C.c_uint32)),
        ("op_a", C.POINTER(C.c_double)),
        ("op_b", C.POINTER(C.c_double)),
        ("op_n", C.POINTER(C.c_int32)),
        ("op_rate", C.POINTER(C.c_double)),
        ("op_cov", C.POINTER(C.c_double)),
        ("subj_order", C.POINTER(C.c_int32)),
        ("owner", C.c_void_p),
    ]


PMX_OP_RESET, PMX_OP_BOLUS, PMX_OP_OBS, PMX_OP_PROP = 0, 1, 2, 3


class PmxError(RuntimeError):
    """A failed C-ABI call (the Python face of ``PharmsolError``, src/error/mod.rs:13-49)."""

    def __init__(self, status: int, message: str):
        self.status = status
        self.status_name = STATUS_NAMES.get(status, str(status))
        super().__init__(f"{self.status_name}: {message}")


# ---- user closures (pmx_model_create_user) ----------------------------------------------------------------------
PMX_K_CUSTOM = 100
PMX_FN_DYNAMICS, PMX_FN_OUTPUTS, PMX_FN_INIT, PMX_FN_DERIVE = 1, 2, 4, 8
PMX_FN_ROUTE_LAG, PMX_FN_ROUTE_BIOAVAILABILITY, PMX_FN_SEQ_EQ, PMX_FN_EQ = 16, 32, 64, 128
PMX_FN_DYNAMICS_BOLUS = 256
USER_FUNCTION_BITS = {"pmx_dynamics": PMX_FN_DYNAMICS, "pmx_dynamics_bolus": PMX_FN_DYNAMICS_BOLUS, "pmx_outputs": PMX_FN_OUTPUTS, "pmx_init": PMX_FN_INIT,
                      "pmx_derive": PMX_FN_DERIVE, "pmx_route_lag": PMX_FN_ROUTE_LAG,
                      "pmx_route_bioavailability": PMX_FN_ROUTE_BIOAVAILABILITY, "pmx_seq_eq": PMX_FN_SEQ_EQ,
                      "pmx_eq": PMX_FN_EQ}


def user_functions_of(source: str) -> int:
    """PMX_FN_* mask of the ``PMX_DEVICE void pmx_<role>(`` definitions a source text holds."""
    import re

    mask = 0
    for name, bit in USER_FUNCTION_BITS.items():
        if re.search(r"PMX_DEVICE\s+void\s+" + name + r"\s*\(", source):
            mask |= bit
    return mask

# parameter names of the built-in diffeq bodies, in the order the bodies read them (include/pmx.h PMX_ODE_*)
ODE_PARAMETER_NAMES = {"one_cmt_iv": ["ke"], "one_cmt_oral": ["ka", "ke"], "two_cmt_iv": ["ke", "kcp", "kpc"],
                       "two_cmt_oral": ["ke", "ka", "kcp", "kpc"], "three_cmt_iv": ["k10", "k12", "k13", "k21", "k31"],
                       "three_cmt_oral": ["ka", "k10", "k12", "k13", "k21", "k31"], "one_cmt_mm": ["vmax", "km", "v"]}
