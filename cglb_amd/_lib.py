"""ctypes binding of libcglb_hip.so (the C ABI declared in include/cglb_hip.h).

The product path has no CPU fallback: if the shared library is missing or a call fails, this
module raises.  torch is imported first so that the ROCm runtime already mapped by torch
(libamdhip64.so.7, librocblas.so.5, librocsolver.so.0) is the one the library binds to —
one HIP runtime per process.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int64, c_void_p

import torch  # noqa: F401  (must precede the dlopen below, see module docstring)

OK, ERR_BAD_ARG, ERR_NOT_PD, ERR_HIP, ERR_BLAS, ERR_STATE, ERR_COMM = range(7)
COMM_ID_BYTES = 128
#: collective callbacks of cglb_comm_init_callbacks: (user, buf, count, dtype, stream) -> 0 on success
ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p)
ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p)
RBF, MATERN32 = 0, 1
F64, F32 = 0, 1

# CGLB_HIP_LIB lets a developer point at an experimental build of the same ABI (tools/ only; there is still no CPU fallback)
_LIB_PATH = os.environ.get("CGLB_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libcglb_hip.so")

#: every symbol include/cglb_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "cglb_version": (c_int, []),
    "cglb_ctx_create": (c_int, [POINTER(c_void_p), c_int64, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "cglb_ctx_destroy": (c_int, [c_void_p]),
    "cglb_last_error": (c_char_p, [c_void_p]),
    "cglb_set_data": (c_int, [c_void_p, c_void_p, c_void_p]),
    "cglb_set_hypers": (c_int, [c_void_p, POINTER(c_double), c_double, c_double, c_double, c_void_p, c_double]),
    "cglb_setup": (c_int, [c_void_p]),
    "cglb_shard_setup_local": (c_int, [c_void_p]),
    "cglb_aat_buffer": (c_void_p, [c_void_p]),
    "cglb_shard_setup_finish": (c_int, [c_void_p]),
    "cglb_logdet": (c_int, [c_void_p, POINTER(c_double)]),
    "cglb_matvec": (c_int, [c_void_p, c_void_p, c_void_p]),
    "cglb_matvec_dot": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "cglb_shard_rhs": (c_int, [c_void_p, c_void_p]),
    "cglb_cross_matvec": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "cglb_precond_apply": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(c_double)]),
    "cglb_shard_precond_u": (c_int, [c_void_p, c_void_p, c_void_p]),
    "cglb_shard_precond_z": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "cglb_shard_precond_z_seg": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64]),
    "cglb_shard_dot": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "cglb_shard_update_v_r": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "cglb_shard_residual": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "cglb_shard_update_p": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "cglb_pcg_solve": (c_int, [c_void_p, c_void_p, c_void_p, c_double, c_int, c_int, POINTER(c_int), POINTER(c_double)]),
    "cglb_objective_and_grad": (c_int, [c_void_p, c_void_p, c_int, c_double, c_int, c_int, POINTER(c_double), POINTER(c_double),
                                        POINTER(c_int), POINTER(c_double)]),
    "cglb_objective_grad_v": (c_int, [c_void_p, c_void_p]),
    "cglb_shard_obj_phase1": (c_int, [c_void_p, c_void_p, c_void_p]),
    "cglb_shard_obj_phase2": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "cglb_shard_obj_phase3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "cglb_shard_obj_finish": (c_int, [c_void_p, c_void_p, POINTER(c_double)]),
    "cglb_set_parallel": (c_int, [c_void_p, c_int, c_int]),
    "cglb_matvec_cyclic": (c_int, [c_void_p, c_void_p, c_void_p]),
    "cglb_rhs_full": (c_int, [c_void_p, c_void_p]),
    "cglb_vec_dot": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "cglb_vec_update_v_r": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "cglb_vec_residual": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "cglb_vec_update_p": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "cglb_vec_update_p_seg": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "cglb_vec_axpy": (c_int, [c_void_p, c_int64, c_double, c_void_p, c_void_p]),
    "cglb_shard_obj_phase1_kv": (c_int, [c_void_p, c_void_p, c_void_p]),
    "cglb_shard_obj_w": (c_int, [c_void_p, c_void_p]),
    "cglb_shard_obj_phase3_cyclic": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "cglb_predict": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "cglb_shard_predict_u": (c_int, [c_void_p, c_void_p, c_void_p]),
    "cglb_shard_predict_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "cglb_comm_get_unique_id": (c_int, [c_void_p]),
    "cglb_comm_init_rccl": (c_int, [c_void_p, c_void_p, c_int, c_int]),
    "cglb_comm_init_callbacks": (c_int, [c_void_p, c_int, c_int, ALLREDUCE_FN, ALLGATHER_FN, c_void_p]),
    "cglb_comm_destroy": (c_int, [c_void_p]),
    "cglb_dist_setup": (c_int, [c_void_p]),
    "cglb_dist_matvec": (c_int, [c_void_p, c_void_p, c_void_p]),
    "cglb_dist_precond_apply": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(c_double)]),
    "cglb_dist_pcg_solve": (c_int, [c_void_p, c_void_p, c_void_p, c_double, c_int, c_int, POINTER(c_int), POINTER(c_double)]),
    "cglb_dist_objective_and_grad": (c_int, [c_void_p, c_void_p, c_int, c_double, c_int, c_int, POINTER(c_double), POINTER(c_double),
                                             POINTER(c_int), POINTER(c_double)]),
    "cglb_dist_predict": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "cglb_get_stat": (c_int, [c_void_p, c_char_p, POINTER(c_double)]),
    "cglb_select_inducing": (c_int, [c_void_p, POINTER(c_double), c_double, c_double, POINTER(c_int64), c_void_p, POINTER(c_double)]),
    "cglb_get_matrix": (c_int, [c_void_p, c_int, c_void_p]),
    "cglb_time_kernel": (c_int, [c_void_p, c_int, c_int, POINTER(c_double)]),
    "cglb_set_option": (c_int, [c_void_p, c_char_p, c_int64]),
}

_lib = None


class CGLBLibraryError(RuntimeError):
    pass


def lib_path() -> str:
    return _LIB_PATH


def load() -> ctypes.CDLL:
    """dlopen the HIP library and bind every declared symbol.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise CGLBLibraryError(
            f"{_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C cglb_amd/csrc`). There is no CPU fallback.")
    lib = ctypes.CDLL(_LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, ctx=None) -> None:
    """Map C error codes to the exceptions the reference raises at the same places."""
    if rc == OK:
        return
    msg = load().cglb_last_error(ctx).decode("utf-8", "replace") if _lib is not None else ""
    if rc == ERR_BAD_ARG:
        raise ValueError(f"cglb_hip: bad argument: {msg}")
    if rc == ERR_NOT_PD:
        # the reference lets torch.cholesky raise a RuntimeError (models.py:202, :210)
        raise RuntimeError(f"cglb_hip: {msg}")
    if rc == ERR_STATE:
        raise RuntimeError(f"cglb_hip: call order: {msg}")
    if rc == ERR_COMM:
        raise RuntimeError(f"cglb_hip: collective: {msg}")
    raise RuntimeError(f"cglb_hip: error {rc}: {msg}")
