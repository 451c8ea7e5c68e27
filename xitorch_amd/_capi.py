"""ctypes binding of libxitorch_amd.so — the C ABI declared in include/xitorch_amd.h.

The product path has NO fallback: if the HIP library is missing or a call
returns non-zero this raises.  PyTorch is only used for device memory and the
current HIP stream (`torch.cuda.current_stream().cuda_stream`).
"""
import ctypes
import os
import re
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (XITORCH_AMD_LIB: a measurement build of the same ABI — trial kernels are compared against the shipped library this way)
LIB_PATH = os.environ.get("XITORCH_AMD_LIB") or os.path.join(_HERE, "csrc", "libxitorch_amd.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "xitorch_amd.h")

_lib = None
ABI_VERSION = 1          # xk_abi_version() of the library this package was written against

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_long = ctypes.c_long
c_double = ctypes.c_double


class NativeLibraryError(RuntimeError):
    pass


def lib():
    """Load (once) and return the native library; raise loudly if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                "xitorch_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU/eager fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        if os.environ.get("XITORCH_AMD_LIB"):
            # a measurement build stands in for the shipped library: say so, and refuse one of another ABI
            # (version entry point + every function the header declares)
            import warnings
            if not hasattr(L, "xk_abi_version") or int(L.xk_abi_version()) != ABI_VERSION:
                raise NativeLibraryError("xitorch_amd: XITORCH_AMD_LIB=%s does not report ABI version %d"
                                         % (LIB_PATH, ABI_VERSION))
            missing = [n for n in header_symbols() if not hasattr(L, n)]
            if missing:
                raise NativeLibraryError("xitorch_amd: XITORCH_AMD_LIB=%s lacks %d of the declared entry points (%s ...)"
                                         % (LIB_PATH, len(missing), ", ".join(missing[:3])))
            warnings.warn("xitorch_amd: native library overridden by XITORCH_AMD_LIB=%s (measurement build)" % LIB_PATH)
        _lib = L
        _declare(_lib)
    return _lib


def header_symbols():
    """All function names declared in include/xitorch_amd.h (used by the CPU tests)."""
    txt = open(HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(xk_[a-z0-9_]+)\s*\(", txt)))


def _declare(L):
    P, I, Lg, D = c_void_p, c_int, c_long, c_double
    sigs = {
        "xk_abi_version": (I, []),
        "xk_stream_create_cu_masked": (I, [I, I, P]),
        "xk_stream_create_cu_masked_pattern": (I, [I, I, I, P]),
        "xk_probe_xcc": (I, [P, P, I, I, P]),
        "xk_stream_destroy": (I, [P]),
        "xk_stream_read": (I, [P, Lg, Lg, P, P]),
        "xk_dense_mm_workspace_elems": (Lg, [I, I, I, I, I]),
        "xk_dense_mm_f64": (I, [P, P, P, P, Lg, I, I, I, I, Lg, Lg, Lg, Lg, Lg, Lg, I, I, I, P]),
        "xk_dense_mm_f32": (I, [P, P, P, P, Lg, I, I, I, I, Lg, Lg, Lg, Lg, Lg, Lg, I, I, I, P]),
    }
    for sfx in ("f64", "f32"):
        sigs["xk_lincomb_" + sfx] = (I, [P, P, P, I, I, I, I, Lg, Lg, Lg, Lg, Lg, Lg, Lg, D, D, P])
        sigs["xk_ritz_residual_" + sfx] = (I, [P, P, P, P, P, P, P, I, I, I, I] + [Lg] * 12 + [P])
        sigs["xk_panel_chol_" + sfx] = (I, [P, P, P, I, I, Lg, Lg, P])
        sigs["xk_panel_transform_" + sfx] = (I, [P, P, I, I, I, Lg, Lg, P])
        sigs["xk_diag_precond_" + sfx] = (I, [P, P, P, P, I, I, I, Lg, Lg, Lg, Lg, Lg, D, P])
        sigs["xk_small_eigh_" + sfx] = (I, [P, P, P, P, Lg, P, I, I, I, I, I, Lg, Lg, P])
    for sfx in ("f64", "f32"):
        sigs["xk_davidson_ritz_" + sfx] = (I, [P] * 12 + [I, I, I, I] + [Lg] * 12 + [P, Lg, P, Lg, P])
        sigs["xk_ritz_guard_" + sfx] = (I, [P, P, P, I, I, I, Lg, Lg, Lg, Lg, P, Lg, P, Lg, P])
        sigs["xk_davidson_orth_" + sfx] = (I, [P, I, I, I, I, Lg, Lg, P, P, P, P, P, Lg, I, P])
        sigs["xk_davidson_extend_t_" + sfx] = (I, [P, P, P, P, I, I, I, I, Lg, Lg, Lg, Lg, Lg, Lg, P, Lg, P])
    sigs["xk_small_eigh_workspace_elems"] = (Lg, [I, I, I])
    sigs["xk_small_eigh_tri_lds_bytes"] = (Lg, [I, I, I])
    for sfx in ("f64", "f32"):
        sigs["xk_small_eigh_tri_" + sfx] = (I, [P, P, P, P, I, I, I, I, Lg, Lg, I, P, P])
    sigs["xk_small_eigh_big_batch"] = (I, [I, I, I])
    sigs["xk_small_eigh_big_workspace_elems"] = (Lg, [I, I, I])
    for sfx in ("f64", "f32"):
        sigs["xk_small_eigh_big_" + sfx] = (I, [P, P, P, P, Lg, P, I, I, I, I, Lg, Lg, I, I, I, P])
    sigs["xk_kry_max_partials"] = (I, [])
    sigs["xk_dense_symm_workspace_elems"] = (Lg, [I, I, I, I])
    sigs["xk_dense_symm_wide_workspace_elems"] = (Lg, [I, I])
    sigs["xk_dense_symm_wide_f32"] = (I, [P, P, P, P, Lg, I, I, I, Lg, Lg, Lg, Lg, Lg, Lg, I, P])
    sigs["xk_dense_symm_wide_tiles_f32"] = (I, [P, P, P, Lg, I, I, I, Lg, Lg, Lg, Lg, I, P])
    sigs["xk_dense_symm_wide_fold_f32"] = (I, [P, P, Lg, I, I, I, Lg, Lg, I, P])
    sigs["xk_dense_wide_workspace_elems"] = (Lg, [I, I, I, I, I])
    sigs["xk_dense_wide_padded_width"] = (I, [I, I])
    sigs["xk_dense_rows_wide_workspace_elems"] = (Lg, [I, I, I, I, I])
    for sfx in ("f64", "f32"):
        sigs["xk_dense_rows_wide_" + sfx] = (I, [P, P, P, P, Lg, I, I, I, I, Lg, Lg, Lg, Lg, Lg, Lg, P])
    for sfx in ("f64", "f32"):
        sigs["xk_dense_wide_" + sfx] = (I, [P, P, P, P, Lg, I, I, I, I, Lg, Lg, Lg, Lg, Lg, Lg, P])
    for sfx in ("f64", "f32"):
        sigs["xk_group_status_" + sfx] = (I, [P, P, P, P, P, I, P])
        sigs["xk_dense_symm_" + sfx] = (I, [P, P, P, P, Lg, I, I, I, Lg, Lg, Lg, Lg, Lg, Lg, I, P])
        sigs["xk_dense_symm_tiles_" + sfx] = (I, [P, P, P, Lg, I, I, I, Lg, Lg, Lg, Lg, I, P])
        sigs["xk_dense_symm_fold_" + sfx] = (I, [P, P, Lg, I, I, I, Lg, Lg, I, P])
    for sfx in ("f64", "f32"):
        sigs["xk_banded_mm_" + sfx] = (I, [P, P, P, I, I, I, I, Lg, Lg, Lg, Lg, Lg, I, P])
        sigs["xk_kry_dots_" + sfx] = (I, [P] * 8 + [I, I, Lg, I, P])
        sigs["xk_bicg_p_" + sfx] = (I, [P] * 8 + [I, I, Lg, I, D, I, P])
        sigs["xk_bicg_s_" + sfx] = (I, [P] * 6 + [I, I, Lg, I, D, P])
        sigs["xk_bicg_final_" + sfx] = (I, [P] * 14 + [I, I, Lg, I, D, I, P])
        sigs["xk_kry_resid_" + sfx] = (I, [P] * 6 + [I, I, Lg, I, P])
        sigs["xk_cg_update_" + sfx] = (I, [P] * 8 + [I, I, Lg, I, D, I, P])
        sigs["xk_cg_p_" + sfx] = (I, [P] * 4 + [I, I, Lg, I, D, P])
        sigs["xk_kry_status_" + sfx] = (I, [P] * 4 + [I, I, P])
        sigs["xk_banded_grad_" + sfx] = (I, [P, P, P, I, I, I, I, Lg, Lg, Lg, Lg, Lg, I, P])
        sigs["xk_dense_outer_" + sfx] = (I, [P, P, P, I, I, I, I, Lg, Lg, Lg, Lg, Lg, Lg, I, P])
    for sfx in ("f64", "f32"):
        sigs["xk_gmres_step_" + sfx] = (I, [P, Lg, P, Lg, I, I, P, P, P, P, P, P, I, P])
        sigs["xk_gmres_finish_" + sfx] = (I, [P, P, Lg, P, I, I, I, Lg, Lg, P])
        sigs["xk_gmres_solve_" + sfx] = (I, [P, P, P, Lg, I, I, I, P])
    for sfx in ("c64", "c128"):
        sigs["xk_kry_dots_" + sfx] = (I, [P] * 8 + [I, I, Lg, I, I, P])
        sigs["xk_bicg_p_" + sfx] = (I, [P] * 8 + [I, I, Lg, I, D, I, P])
        sigs["xk_bicg_s_" + sfx] = (I, [P] * 6 + [I, I, Lg, I, D, P])
        sigs["xk_bicg_final_" + sfx] = (I, [P] * 14 + [I, I, Lg, I, D, I, P])
        sigs["xk_kry_resid_" + sfx] = (I, [P] * 6 + [I, I, Lg, I, P])
        sigs["xk_cg_update_" + sfx] = (I, [P] * 8 + [I, I, Lg, I, D, I, P])
        sigs["xk_cg_p_" + sfx] = (I, [P] * 4 + [I, I, Lg, I, D, P])
    sigs["xk_comm_available"] = (I, [])
    sigs["xk_comm_unique_id"] = (I, [P])
    sigs["xk_comm_init_rank"] = (I, [P, I, I, I, P])
    sigs["xk_comm_init_all"] = (I, [I, P, P])
    sigs["xk_comm_size"] = (I, [P, P, P])
    sigs["xk_comm_destroy"] = (I, [P])
    sigs["xk_allreduce_f64"] = (I, [P, P, Lg, I, P])
    sigs["xk_allreduce_f32"] = (I, [P, P, Lg, I, P])
    sigs["xk_vec_dots_workspace_elems"] = (Lg, [])
    for sfx in ("f64", "f32"):
        sigs["xk_vec_dots_" + sfx] = (I, [P] * 8 + [I, Lg, P, Lg, P, P])
        sigs["xk_broyden_axpy_" + sfx] = (I, [P, P, D, P, D, P, Lg, P, P, I, D, Lg, P])
    for name, (res, args) in sigs.items():
        if not hasattr(L, name):
            continue  # reported by the symbol test, and by check() at call time
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args


def check(rc, what):
    if rc != 0:
        raise NativeLibraryError("xitorch_amd native call %s failed with code %d" % (what, rc))


def stream_ptr():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return c_void_p(0)
    return c_void_p(t.data_ptr())


def suffix(dtype):
    if dtype == torch.float64:
        return "f64"
    if dtype == torch.float32:
        return "f32"
    if dtype == torch.complex128:
        return "c128"
    if dtype == torch.complex64:
        return "c64"
    raise NativeLibraryError("xitorch_amd native kernels support float64/float32, got %s" % dtype)


def require_device(t, what="tensor"):
    if not t.is_cuda:
        raise NativeLibraryError(
            "xitorch_amd: %s must live on a HIP device (got %s); the native path has no CPU fallback"
            % (what, t.device))


def fn(name):
    L = lib()
    if not hasattr(L, name):
        raise NativeLibraryError("xitorch_amd: symbol %s missing from %s" % (name, LIB_PATH))
    return getattr(L, name)
