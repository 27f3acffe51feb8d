"""ctypes binding of libkrylov_hip.so (C ABI declared in include/krylov_hip.h).

This is the exact call surface the Julia shim (`julia/KrylovKitHIP.jl`) uses
through `ccall`; the Python host mirror issues the same calls.  There is no CPU fallback: if
the shared library is missing or no gfx950 device is visible the import / context creation
raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("KRYLOV_HIP_LIB", _HERE.parent / "lib" / "libkrylov_hip.so"))

KK_OK = 0
KK_ERR_INVALID = -1
KK_ERR_DIM = -2
KK_ERR_HIP = -3
KK_ERR_NOMEM = -4
KK_ERR_ZERO_NORM = -5
KK_ERR_UNSUPPORTED = -6
KK_ERR_NO_DEVICE = -7
KK_OP_SYMMETRIC = 1

ORTH_CODES = {"cgs": 0, "mgs": 1, "cgs2": 2, "mgs2": 3, "cgsir": 4, "mgsir": 5}


class KrylovHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libkrylov_hip error {code}: {msg}")
        self.code = code


class DimensionMismatch(KrylovHipError, ValueError):
    """KK_ERR_DIM -- mirrors Julia's DimensionMismatch (src/orthonormal.jl:93,140,296)."""


class NoDeviceError(KrylovHipError):
    """KK_ERR_NO_DEVICE -- the product path never falls back to the CPU."""


c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
c_i64p = C.POINTER(C.c_int64)
c_i32p = C.POINTER(C.c_int32)
c_vp = C.c_void_p
c_vpp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); every symbol include/krylov_hip.h declares
SIGNATURES = {
    "kk_version": (C.c_int, []),
    "kk_last_error": (C.c_char_p, []),
    "kk_device_count": (C.c_int, [c_ip]),
    "kk_ctx_create": (C.c_int, [C.c_int, c_vpp]),
    "kk_ctx_destroy": (C.c_int, [c_vp]),
    "kk_ctx_set_stream": (C.c_int, [c_vp, c_vp]),
    "kk_ctx_get_stream": (C.c_int, [c_vp, c_vpp]),
    "kk_ctx_sync": (C.c_int, [c_vp]),
    "kk_ctx_set_option": (C.c_int, [c_vp, C.c_char_p, C.c_double]),
    "kk_ctx_get_option": (C.c_int, [c_vp, C.c_char_p, c_dp]),
    "kk_ctx_timer_start": (C.c_int, [c_vp]),
    "kk_ctx_timer_stop": (C.c_int, [c_vp, c_dp]),
    "kk_ctx_prof_enable": (C.c_int, [c_vp, C.c_int]),
    "kk_ctx_prof_reset": (C.c_int, [c_vp]),
    "kk_ctx_prof_get": (C.c_int, [c_vp, C.c_char_p, c_dp, c_i64p]),
    "kk_ctx_set_allreduce": (C.c_int, [c_vp, c_vp, c_vp]),
    "kk_ctx_workspace_size": (C.c_int, [c_vp, c_i64p, c_i64p]),
    "kk_ctx_set_workspace": (C.c_int, [c_vp, c_vp, c_vp]),
    "kk_op_set_halo_hook": (C.c_int, [c_vp, c_vp, c_vp]),
    "kk_gather_ptr": (C.c_int, [c_vp, c_vp, c_vp, C.c_int64, c_vp]),
    "kk_basis_create": (C.c_int, [c_vp, C.c_int64, C.c_int, c_vpp]),
    "kk_basis_free": (C.c_int, [c_vp]),
    "kk_basis_info": (C.c_int, [c_vp, c_i64p, c_i64p, c_ip, c_vpp]),
    "kk_basis_upload": (C.c_int, [c_vp, C.c_int, c_dp]),
    "kk_basis_download": (C.c_int, [c_vp, C.c_int, c_dp]),
    "kk_basis_upload_device": (C.c_int, [c_vp, C.c_int, c_vp]),
    "kk_basis_download_device": (C.c_int, [c_vp, C.c_int, c_vp]),
    "kk_basis_invalidate_gram": (C.c_int, [c_vp]),
    "kk_vec_dot": (C.c_int, [c_vp, C.c_int, c_vp, C.c_int, c_dp]),
    "kk_vec_nrm2": (C.c_int, [c_vp, C.c_int, c_dp]),
    "kk_vec_axpby": (C.c_int, [c_vp, C.c_int, c_vp, C.c_int, C.c_double, C.c_double]),
    "kk_vec_scal": (C.c_int, [c_vp, C.c_int, C.c_double]),
    "kk_vec_copy_scal": (C.c_int, [c_vp, C.c_int, c_vp, C.c_int, C.c_double]),
    "kk_vec_zero": (C.c_int, [c_vp, C.c_int]),
    "kk_vec_fill_random": (C.c_int, [c_vp, C.c_int, C.c_uint64]),
    "kk_csr_create": (C.c_int, [c_vp, C.c_int64, C.c_int64, C.c_int64, c_i64p, c_i32p, c_dp, C.c_int, C.c_int, c_vpp]),
    "kk_csc_create": (C.c_int, [c_vp, C.c_int64, C.c_int64, C.c_int64, c_i64p, c_i64p, c_dp, C.c_int, C.c_int, c_vpp]),
    "kk_op_free": (C.c_int, [c_vp]),
    "kk_op_info": (C.c_int, [c_vp, c_i64p, c_i64p, c_i64p, c_ip, c_i64p]),
    "kk_op_set_ghost": (C.c_int, [c_vp, C.c_int64, C.c_int64, c_vp]),
    "kk_spmv": (C.c_int, [c_vp, C.c_int, c_vp, C.c_int, c_vp, C.c_int]),
    "kk_spmv_affine": (C.c_int, [c_vp, c_vp, C.c_int, c_vp, C.c_int, C.c_double, C.c_double]),
    "kk_spmv_affine_dot": (C.c_int, [c_vp, c_vp, C.c_int, c_vp, C.c_int, C.c_double, C.c_double, c_dp]),
    "kk_cg_update": (C.c_int, [c_vp, C.c_int, c_vp, C.c_int, c_vp, C.c_int, c_vp, C.c_int, C.c_double, c_dp]),
    "kk_cg_iterate": (C.c_int, [c_vp, c_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int,
                                C.c_double, c_dp, c_dp]),
    "kk_bicgstab_half": (C.c_int, [c_vp, c_vp, C.POINTER(C.c_int), C.c_double, C.c_double, C.c_int, C.c_double, c_dp, c_dp]),
    "kk_bicgstab_full": (C.c_int, [c_vp, c_vp, C.POINTER(C.c_int), C.c_double, C.c_double, C.c_int, C.POINTER(C.c_int),
                                   c_dp, c_dp, c_dp]),
    "kk_lsmr_step_u": (C.c_int, [c_vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, c_dp]),
    "kk_lsmr_update": (C.c_int, [c_vp, C.c_int, C.c_int, C.c_int, c_vp, C.c_int, C.c_double, C.c_double, C.c_double]),
    "kk_gather": (C.c_int, [c_vp, C.c_int, c_vp, C.c_int64, c_vp]),
    "kk_lanczos_coef_dev": (C.c_int, [c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_int, c_vp, c_vp]),
    "kk_norm_scalars_dev": (C.c_int, [c_vp, c_vp, c_vp, c_vp]),
    "kk_project": (C.c_int, [c_vp, C.c_int, C.c_int, c_vp, C.c_int, C.c_double, C.c_double, c_dp]),
    "kk_unproject": (C.c_int, [c_vp, C.c_int, c_vp, C.c_int, C.c_int, c_dp, C.c_double, C.c_double]),
    "kk_rank1update": (C.c_int, [c_vp, C.c_int, C.c_int, c_vp, C.c_int, c_dp, C.c_double, C.c_double]),
    "kk_basistransform": (C.c_int, [c_vp, C.c_int, C.c_int, C.c_int, c_dp, C.c_int]),
    "kk_givens_rmul": (C.c_int, [c_vp, C.c_int, C.c_int, C.c_double, C.c_double]),
    "kk_householder_rmul": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_double]),
    "kk_orthogonalize": (C.c_int, [c_vp, C.c_int, C.c_int, c_vp, C.c_int, C.c_int, C.c_double, c_dp, c_dp, c_ip]),
    "kk_orthonormalize": (C.c_int, [c_vp, C.c_int, C.c_int, c_vp, C.c_int, C.c_int, C.c_double, c_dp, c_dp, c_ip]),
    "kk_orthogonalize_vec": (C.c_int, [c_vp, C.c_int, c_vp, C.c_int, C.c_int, C.c_double, c_dp, c_dp]),
    "kk_lanczos_expand": (C.c_int, [c_vp, c_vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, c_dp, c_dp, c_ip]),
    "kk_lanczos_initialize": (C.c_int, [c_vp, c_vp, C.c_int, C.c_int, C.c_double, c_dp, c_dp]),
    "kk_arnoldi_expand": (C.c_int, [c_vp, c_vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, c_dp, c_dp, c_ip]),
    "kk_arnoldi_initialize": (C.c_int, [c_vp, c_vp, C.c_int, C.c_int, C.c_double, c_dp, c_dp]),
    "kk_gkl_expand": (C.c_int, [c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_double, C.c_double, c_dp, c_dp, c_ip, c_ip]),
    "kk_gkl_initialize": (C.c_int, [c_vp, c_vp, c_vp, c_dp, c_dp]),
    "kk_block_inner": (C.c_int, [c_vp, C.c_int, C.c_int, c_vp, C.c_int, C.c_int, c_dp, C.c_int]),
    "kk_block_apply": (C.c_int, [c_vp, c_vp, C.c_int, c_vp, C.c_int, C.c_int]),
    "kk_block_update": (C.c_int, [c_vp, C.c_int, C.c_int, c_vp, C.c_int, C.c_int, c_dp, C.c_int, C.c_double, C.c_double, c_dp]),
    "kk_block_qr": (C.c_int, [c_vp, C.c_int, C.c_int, C.c_int, C.c_double, c_dp, C.c_int, c_ip, c_ip, c_ip]),
    "kk_block_reorthogonalize": (C.c_int, [c_vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "kk_blocklanczos_initialize": (C.c_int, [c_vp, c_vp, C.c_int, C.c_int, C.c_int, C.c_double, c_ip, c_dp, C.c_int, c_dp]),
    "kk_blocklanczos_expand": (C.c_int, [c_vp, c_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, c_ip, c_dp, C.c_int, c_dp, C.c_int, c_dp, c_ip]),
    "kk_apply_fused_dev": (C.c_int, [c_vp, c_vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, c_vp]),
    "kk_apply_fused_dev2": (C.c_int, [c_vp, c_vp, C.c_int, C.c_int, C.c_int, c_vp, c_vp, C.c_double, C.c_int, c_vp]),
    "kk_unproject_devcoef": (C.c_int, [c_vp, C.c_int, c_vp, C.c_int, C.c_int, c_vp, C.c_double, C.c_double, c_vp]),
    "kk_project_dev": (C.c_int, [c_vp, C.c_int, C.c_int, c_vp, C.c_int, C.c_int, c_vp]),
    "kk_unproject_dev": (C.c_int, [c_vp, C.c_int, c_vp, C.c_int, C.c_int, c_dp, C.c_double, C.c_double, c_vp]),
    "kk_dot_dev": (C.c_int, [c_vp, C.c_int, c_vp, C.c_int, c_vp]),
    "kk_nrm2_dev": (C.c_int, [c_vp, C.c_int, c_vp]),
    "kk_axpy_dev": (C.c_int, [c_vp, C.c_int, c_vp, C.c_int, c_vp, C.c_double]),
    "kk_scal_rsqrt_dev": (C.c_int, [c_vp, C.c_int, c_vp]),
    # multi-GPU: RCCL inside the library
    "kk_comm_get_unique_id": (C.c_int, [c_vp]),
    "kk_comm_init": (C.c_int, [c_vp, c_vp, C.c_int, C.c_int, C.c_int]),
    "kk_comm_destroy": (C.c_int, [c_vp]),
    "kk_comm_info": (C.c_int, [c_vp, c_ip, c_ip, c_ip]),
    "kk_comm_stats": (C.c_int, [c_vp, c_i64p, c_i64p, c_i64p]),
    "kk_comm_allreduce": (C.c_int, [c_vp, c_vp, C.c_int64, C.c_int]),
    "kk_comm_barrier": (C.c_int, [c_vp]),
    "kk_csr_create_sharded": (C.c_int, [c_vp, C.c_int64, c_i64p, C.c_int64, c_i64p, c_i64p, c_dp, C.c_int, C.c_int, c_vpp]),
    "kk_csr_create_sharded_rect": (C.c_int, [c_vp, C.c_int64, C.c_int64, C.c_int64, c_i64p, c_i64p, c_dp, C.c_int, c_vpp, c_i64p]),
}
KK_MAX_M = 256   # basis vectors per project / unproject / fused expand call (csrc/kk_internal.h)
KK_COMM_ID_BYTES = 128
KK_COMM_FORCE_COLLECTIVES = 1

_lib = None


def load():
    """dlopen libkrylov_hip.so and attach prototypes.  Raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C krylovkit.jl_amd` (or __graft_entry__.build()); "
            "krylovkit_hip has no CPU fallback")
    lib = C.CDLL(str(LIB_PATH), mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int):
    if status == KK_OK:
        return
    msg = load().kk_last_error().decode("utf-8", "replace")
    if status == KK_ERR_DIM:
        raise DimensionMismatch(status, msg)
    if status == KK_ERR_NO_DEVICE:
        raise NoDeviceError(status, msg)
    raise KrylovHipError(status, msg)
