"""ctypes binding of the C ABI in include/multical_hip.h.

The shared library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There is no
CPU fallback: if the library is missing or no HIP device is present the backend raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MULTICAL_HIP_LIB selects another build of the same library (e.g. the phase-timing diagnostic build)
LIB_PATH = os.environ.get("MULTICAL_HIP_LIB") or os.path.join(_HERE, "libmulticalhip.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_ctx = C.c_void_p

# name -> (restype, argtypes); kept in one table so the tests can check it against the header
SIGNATURES = {
    "mc_create": (_ctx, [C.c_int]),
    "mc_destroy": (None, [_ctx]),
    "mc_last_error": (C.c_char_p, [_ctx]),
    "mc_add_design_variable": (C.c_int, [_ctx, C.c_int, _dp, C.c_int, C.c_int, C.c_double]),
    "mc_add_spline": (C.c_int, [_ctx, C.c_int, C.c_int, _dp, C.c_int, _dp, C.c_int, C.c_int]),
    "mc_spline_first_dv": (C.c_int, [_ctx, C.c_int]),
    "mc_add_camera": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mc_add_reprojection_batch": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _ip, _ip, _ip,
                                            _dp, _dp, _dp, C.c_double, C.c_int, C.c_double]),
    "mc_add_static_reprojection_batch": (C.c_int, [_ctx, C.c_int, C.c_int, _ip, _ip, C.c_int, _ip, _ip, _ip, _dp, _dp, _dp,
                                                   C.c_int, C.c_double]),
    "mc_add_accelerometer_batch": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp,
                                             _dp, C.c_int, C.c_double]),
    "mc_add_gyroscope_batch": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, C.c_int,
                                         C.c_double]),
    "mc_add_gyroscope_prior_batch": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, C.c_double]),
    "mc_add_transformation_batch": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_double, C.c_double,
                                              C.c_int, C.c_double]),
    "mc_add_matrix_design_variable": (C.c_int, [_ctx, _dp, _ip, C.c_int, C.c_double]),
    "mc_add_accelerometer_batch_sm": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _ip, C.c_int,
                                                _dp, _dp, _dp, C.c_int, C.c_double]),
    "mc_add_gyroscope_batch_sm": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_int, _dp, _dp, _dp, C.c_int, C.c_double]),
    "mc_add_lidar_batch": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _ip, _ip,
                                     C.c_double, C.c_int, C.c_double]),
    "mc_add_motion_term": (C.c_int, [_ctx, C.c_int, _dp, C.c_int]),
    "mc_set_last_batch_landmarks": (C.c_int, [_ctx, C.c_int, _ip]),
    "mc_set_last_batch_sqrt_inv_r": (C.c_int, [_ctx, C.c_int, C.c_int, _dp]),
    "mc_compute_matrix_sqrt": (C.c_int, [C.c_int, _dp, _dp]),
    "mc_set_shard": (C.c_int, [_ctx, C.c_int, C.c_int]),
    "mc_shard_info": (C.c_int, [_ctx, _ip, _ip]),
    "mc_set_clear_mode": (C.c_int, [_ctx, C.c_int]),
    "mc_finalize": (C.c_int, [_ctx]),
    "mc_dims": (C.c_int, [_ctx, _ip, _ip, _ip, _ip, _ip]),
    "mc_dv_block": (C.c_int, [_ctx, C.c_int, _ip, _ip]),
    "mc_get_pattern": (C.c_int, [_ctx, _ip, _ip]),
    "mc_evaluate_error": (C.c_int, [_ctx, C.c_int, _dp, _dp]),
    "mc_build_system": (C.c_int, [_ctx, C.c_int, _dp, _dp]),
    "mc_evaluate_error_device": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mc_build_system_device": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mc_build_system_device_noalloc": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mc_evaluate_error_and_build_system_device": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mc_get_term_chi2": (C.c_int, [_ctx, _dp]),
    "mc_apply_update": (C.c_int, [_ctx, _dp]),
    "mc_revert_update": (C.c_int, [_ctx]),
    "mc_get_dv_params": (C.c_int, [_ctx, C.c_int, _dp]),
    "mc_device_status": (C.c_int, [_ctx]),
    "mc_lidar_find_points_on_targets": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _ip, _ip, _dp, _dp, C.c_double,
                                                  C.c_int, _dp, C.POINTER(C.c_uint), _dp]),
    "mc_jt_nnz": (C.c_int, [_ctx, C.POINTER(C.c_longlong)]),
    "mc_get_jt_pattern": (C.c_int, [_ctx, _ip, _ip]),
    "mc_build_jacobian_transpose": (C.c_int, [_ctx, C.c_int, _dp]),
    "mc_build_jacobian_transpose_device": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p]),
    "mc_shared_count": (C.c_int, [_ctx]),
    "mc_get_shared_indices": (C.c_int, [_ctx, _ip]),
    "mc_pack_shared": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mc_unpack_shared": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mc_shared_rhs_count": (C.c_int, [_ctx]),
    "mc_get_shared_rhs_indices": (C.c_int, [_ctx, _ip]),
    "mc_pack_shared_rhs": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mc_unpack_shared_rhs": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mc_num_launches": (C.c_int, [_ctx]),
    "mc_launch_info": (C.c_int, [_ctx, C.c_int, _ip, _ip, _ip]),
    "mc_launch_counts": (C.c_int, [_ctx, C.c_int, _dp, _dp, _dp]),
    "mc_phase_cycles": (C.c_int, [_ctx, C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]),
    "mc_build_system_timed": (C.c_int, [_ctx, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "mc_device_info": (C.c_int, [_ctx, C.c_char_p, C.c_int, _ip, _dp]),
}



class OptimizerOptions(C.Structure):
    """mc_optimizer_options (include/multical_hip.h)"""
    _fields_ = [("convergenceJDescentRatioThreshold", C.c_double), ("convergenceDeltaX", C.c_double),
                ("maxIterations", C.c_int), ("verbose", C.c_int), ("nThreads", C.c_int), ("trustRegionPolicy", C.c_int),
                ("levenbergMarquardtLambdaInit", C.c_double), ("linearSolver", C.c_int)]


class SolutionReturnValue(C.Structure):
    """mc_solution_return_value (include/multical_hip.h)"""
    _fields_ = [("JStart", C.c_double), ("JFinal", C.c_double), ("iterations", C.c_int), ("failedIterations", C.c_int),
                ("dXFinal", C.c_double), ("dJFinal", C.c_double), ("linearSolverFailure", C.c_int),
                ("seconds_initialize", C.c_double), ("seconds_evaluate_error", C.c_double),
                ("seconds_build_system", C.c_double), ("seconds_solve_system", C.c_double),
                ("seconds_state_update", C.c_double), ("factor_nnz", C.c_double), ("factor_flops", C.c_double)]


SIGNATURES.update({
    "mc_get_block_bases": (C.c_int, [_ctx, _ip]),
    "mc_get_block_times": (C.c_int, [_ctx, _dp]),
    "mc_get_block_splines": (C.c_int, [_ctx, _ip]),
    "mc_schur_create": (C.c_void_p, [_ctx, C.c_void_p, C.c_int]),
    "mc_schur_destroy": (None, [C.c_void_p]),
    "mc_schur_last_error": (C.c_char_p, [C.c_void_p]),
    "mc_schur_planned": (C.c_int, [C.c_void_p]),
    "mc_schur_solve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, _dp, C.c_void_p]),
    "mc_schur_info": (C.c_int, [C.c_void_p, _ip, _dp]),
    "mc_cholesky_create": (C.c_void_p, []),
    "mc_cholesky_destroy": (None, [C.c_void_p]),
    "mc_cholesky_last_error": (C.c_char_p, [C.c_void_p]),
    "mc_cholesky_analyze": (C.c_int, [C.c_void_p, C.c_int, _ip, _ip, C.c_int, _ip]),
    "mc_cholesky_set_block_coordinates": (C.c_int, [C.c_void_p, _dp, C.c_int]),
    "mc_cholesky_factorize": (C.c_int, [C.c_void_p, _dp]),
    "mc_cholesky_solve": (C.c_int, [C.c_void_p, _dp, _dp]),
    "mc_cholesky_covariance_blocks": (C.c_int, [C.c_void_p, C.c_int, _ip, _ip, C.POINTER(C.c_longlong), _dp]),
    "mc_cholesky_stats": (C.c_int, [C.c_void_p, _ip, C.POINTER(C.c_longlong), _dp]),
    "mc_cholesky_block_permutation": (C.c_int, [C.c_void_p, _ip]),
    "mc_optimizer_options_default": (None, [C.POINTER(OptimizerOptions)]),
    "mc_optimize": (C.c_int, [_ctx, C.POINTER(OptimizerOptions), C.POINTER(SolutionReturnValue)]),
    "mc_host_last_error": (C.c_char_p, []),
})

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_longlong)
BROADCAST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int)
SIGNATURES["mc_optimize_sharded"] = (C.c_int, [_ctx, C.POINTER(OptimizerOptions), C.POINTER(SolutionReturnValue), ALLREDUCE_FN,
                                               BROADCAST_FN, C.c_void_p])

_lib = None


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so.7; if this library pulled in
    the system copy as well, whichever runtime initialises second finds no device ("No HIP GPUs are available").  When
    torch is installed, its copy is loaded first with RTLD_GLOBAL so that libmulticalhip.so binds to the same one and
    device pointers / streams can be shared with torch (bench.py, RCCL all-reduce)."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load libmulticalhip.so (once) and declare every entry point.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "multical_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback." % LIB_PATH)
    _preload_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
