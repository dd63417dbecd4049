"""ctypes binding of the C-ABI in include/cugp.h (cugp_amd/lib/libcugp.so).

This is plumbing for the Python tests, bench.py and the multi-GPU driver; the product is the
shared library.  There is NO CPU fallback: if the library is missing or no GPU is visible the
calls raise.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libcugp.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
OBJECTIVE = C.CFUNCTYPE(None, C.c_void_p, _dp, _dp, _dp)
VALUE_FN = C.CFUNCTYPE(None, C.c_void_p, _dp, _dp)
GRADIENT_FN = C.CFUNCTYPE(None, C.c_void_p, _dp, _dp)

CUGP_OK = 0
CUGP_ERR_INVALID, CUGP_ERR_NOMEM, CUGP_ERR_DEVICE, CUGP_ERR_NODEVICE = -1, -2, -3, -4
ERR_NAMES = {-1: "CUGP_ERR_INVALID", -2: "CUGP_ERR_NOMEM", -3: "CUGP_ERR_DEVICE", -4: "CUGP_ERR_NODEVICE",
             -5: "CUGP_ERR_BUSY"}


class CugpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERR_NAMES.get(code, "CUGP_ERR"), code, msg))
        self.code = code


# name -> (restype, argtypes); every symbol include/cugp.h declares
SIGNATURES = {
    "cugp_version": (C.c_int, []),
    "cugp_build_id": (C.c_char_p, []),
    "cugp_last_error": (C.c_char_p, []),
    "cugp_device_count": (C.c_int, [_ip]),
    "cugp_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "cugp_create_padded": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "cugp_destroy": (C.c_int, [C.c_void_p]),
    "cugp_dims": (C.c_int, [C.c_void_p, _ip, _ip, _ip]),
    "cugp_set_overlap": (C.c_int, [C.c_void_p, C.c_int]),
    "cugp_set_data": (C.c_int, [C.c_void_p, _dp, _dp]),
    "cugp_set_data_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cugp_set_loghyper": (C.c_int, [C.c_void_p, _dp]),
    "cugp_get_loghyper": (C.c_int, [C.c_void_p, _dp]),
    "cugp_loglik": (C.c_int, [C.c_void_p, _dp]),
    "cugp_loglik_grad": (C.c_int, [C.c_void_p, _dp, _dp]),
    "cugp_grad": (C.c_int, [C.c_void_p, _dp]),
    "cugp_loglik_grad_enqueue": (C.c_int, [C.c_void_p, C.c_int]),
    "cugp_loglik_grad_fetch": (C.c_int, [C.c_void_p, _dp, _dp]),
    "cugp_last_quad_logdet": (C.c_int, [C.c_void_p, _dp, _dp]),
    "cugp_predict": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp, _dp]),
    "cugp_nlpp": (C.c_int, [_dp, _dp, _dp, C.c_int, _dp]),
    "cugp_compute_K_train": (C.c_int, [C.c_void_p, _dp]),
    "cugp_compute_squared_dist": (C.c_int, [C.c_void_p, C.c_double, _dp]),
    "cugp_compute_k_test": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp]),
    "cugp_get_cholesky": (C.c_int, [C.c_void_p, _dp]),
    "cugp_get_K_inverse": (C.c_int, [C.c_void_p, _dp]),
    "cugp_get_alpha": (C.c_int, [C.c_void_p, _dp]),
    "cugp_potrf": (C.c_int, [C.c_int, _dp, _dp, C.c_int]),
    "cugp_potri": (C.c_int, [C.c_int, _dp, _dp, C.c_int]),
    "cugp_chol_and_det": (C.c_int, [C.c_int, _dp, _dp, _dp, _dp, C.c_int]),
    "cugp_potrs_vec": (C.c_int, [C.c_int, _dp, _dp, _dp, C.c_int]),
    "cugp_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "cugp_get_phase_ms": (C.c_int, [C.c_void_p, _dp]),
    "cugp_get_kernel_stats": (C.c_int, [C.c_void_p, _dp, C.POINTER(C.c_longlong), _dp, C.c_int]),
    "cugp_get_kernel_stats_kind": (C.c_int, [C.c_void_p, C.c_int, _dp, C.POINTER(C.c_longlong), _dp, C.c_int]),
    "cugp_get_kernel_stats_dispatch_ms": (C.c_int, [C.c_void_p, C.c_int, _dp]),
    "cugp_get_stream": (C.c_void_p, [C.c_void_p]),
    "cugp_cg_minimize": (C.c_int, [OBJECTIVE, C.c_void_p, _dp, C.c_int, _dp, C.c_int, _ip]),
    "cugp_rprop_minimize": (C.c_int, [OBJECTIVE, C.c_void_p, _dp, C.c_int, _dp, C.c_int, _ip]),
    "cugp_cg_solve": (C.c_int, [C.c_void_p, C.c_int, _dp, C.c_int, _ip]),
    "cugp_cg_minimize_sparing": (C.c_int, [VALUE_FN, GRADIENT_FN, C.c_void_p, _dp, C.c_int, _dp, C.c_int, _ip, _ip]),
    "cugp_cg_solve_sparing": (C.c_int, [C.c_void_p, C.c_int, _dp, C.c_int, _ip, _ip]),
    "cugp_rprop_solve": (C.c_int, [C.c_void_p, C.c_int, _dp, C.c_int, _ip]),
    "cugp_bcm_create": (C.c_int, [C.c_int, _ip, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "cugp_bcm_create_split": (C.c_int, [_dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "cugp_bcm_create_multi": (C.c_int, [C.c_int, _ip, C.c_int, _ip, C.c_int, C.POINTER(C.c_void_p)]),
    "cugp_bcm_create_split_multi": (C.c_int, [_dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int, _ip, C.POINTER(C.c_void_p)]),
    "cugp_bcm_destroy": (C.c_int, [C.c_void_p]),
    "cugp_bcm_num_experts": (C.c_int, [C.c_void_p, _ip]),
    "cugp_bcm_expert": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "cugp_bcm_set_expert_data": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp]),
    "cugp_bcm_set_loghyper": (C.c_int, [C.c_void_p, _dp]),
    "cugp_bcm_get_loghyper": (C.c_int, [C.c_void_p, _dp]),
    "cugp_bcm_loglik_grad": (C.c_int, [C.c_void_p, _dp, _dp, _dp]),
    "cugp_bcm_loglik_grad_rows": (C.c_int, [C.c_void_p, _dp]),
    "cugp_bcm_loglik_grad_rows_device": (C.c_int, [C.c_void_p, C.c_void_p, _ip]),
    "cugp_comm_unique_id": (C.c_int, [C.c_void_p, C.c_int]),
    "cugp_comm_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "cugp_comm_destroy": (C.c_int, [C.c_void_p]),
    "cugp_bcm_loglik_grad_allgather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, _dp]),
    "cugp_bcm_predict_partial": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp, _dp]),
    "cugp_poe_finish": (C.c_int, [_dp, _dp, C.c_int, _dp, _dp]),
    "cugp_bcm_predict": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp, _dp]),
    "cugp_bcm_cg_solve": (C.c_int, [C.c_void_p, C.c_int, _dp, C.c_int, _ip]),
    "cugp_test_gemm_nt": (C.c_int, [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, C.c_int]),
    "cugp_mfma_peak_tflops": (C.c_int, [C.c_int, _dp]),
    "cugp_set_tuning": (C.c_int, [C.c_int, C.c_int]),
    "cugp_set_handle_tuning": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "cugp_get_handle_tuning": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "cugp_bench_la": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _dp]),
    "cugp_bench_la_check": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp]),
    "cugp_potrf_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _ip]),
    "cugp_potrf_plan_sub": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _ip]),
}

_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  The PyTorch-ROCm wheel carries its own libamdhip64 / libhsa-runtime64 (same
    SONAMEs as /opt/rocm's, an older build): whichever copy a process loads first serves every later user.  With torch
    imported first libcugp.so simply runs on torch's copy; the other way round torch finds /opt/rocm's runtime under
    its own libraries' names and fails at the first CUDA call ("No HIP GPUs are available").  So when a torch wheel is
    installed and not yet imported, its runtime is loaded (not torch itself) before libcugp.so.
    CUGP_OWN_HIP_RUNTIME=1 keeps the library on the runtime it was linked against."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("CUGP_OWN_HIP_RUNTIME", "0") not in ("", "0"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    rt = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(rt):
        try:
            C.CDLL(rt, mode=C.RTLD_GLOBAL)
        except OSError:
            pass                                    # not loadable here: libcugp.so keeps its own


def lib():
    """Load libcugp.so (built in tree by cugp_amd.build); raises if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                "%s not built -- run `python -m cugp_amd.build` (there is no CPU fallback)" % LIB_PATH)
        _share_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != CUGP_OK:
        msg = lib().cugp_last_error()
        raise CugpError(rc, msg.decode() if msg else "")
    return rc


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def ptr(a):
    return a.ctypes.data_as(_dp)
