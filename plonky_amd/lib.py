"""ctypes loader for libplonky_hip.so -- the C ABI of include/plonky_hip.h.

There is deliberately NO fallback: if the shared library is missing or a call fails the
caller gets an exception.  Nothing here imports the oracle.
"""
import ctypes
import os
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# PLK_HIP_LIB points the loader at another build of the same library (tools/ntt_experiments.sh)
SO_PATH = os.environ.get("PLK_HIP_LIB") or os.path.join(_CSRC, "libplonky_hip.so")

PLK_OK = 0
PLK_ERR_INVALID_ARG = -1
PLK_ERR_SIZE_MISMATCH = -2
PLK_ERR_NOT_POW2 = -3
PLK_ERR_TWO_ADICITY = -4
PLK_ERR_HIP = -5
PLK_ERR_NO_DEVICE = -6
PLK_ERR_OOM = -7

# every symbol declared in include/plonky_hip.h: (name, restype, argtypes)
_vp, _i, _u, _sz, _u64, _cp = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_char_p
SYMBOLS = [
    ("plk_init", _i, [_i]),
    ("plk_init_devices", _i, [_i]),
    ("plk_device_count", _i, []),
    ("plk_set_thread_device", _i, [_i]),
    ("plk_group_copy_stats", _i, [_vp, _vp]),
    ("plk_thread_hip_device", _i, [_i]),
    ("plk_multi_plan", _i, [_u, _u, _sz, _u, _vp, _vp, _vp, _vp]),
    ("plk_shutdown", None, []),
    ("plk_last_error", _cp, []),
    ("plk_min_gpu_log_n", _u, []),
    ("plk_field_limbs", _i, [_i]),
    ("plk_curve_limbs", _i, [_i]),
    ("plk_curve_scalar_field", _i, [_i]),
    ("plk_ntt_precompute", _i, [_i, _u]),
    ("plk_ntt_clear_cache", _i, []),
    ("plk_ntt_precompute_table", _i, [_i, _u, _vp]),
    ("plk_ntt_precompute_table_dev", _i, [_i, _u, _vp, _vp]),
    ("plk_ntt", _i, [_i, _u, _i, _vp, _vp]),
    ("plk_ntt_batch", _i, [_i, _u, _i, _u, _vp, _vp]),
    ("plk_ntt_dev", _i, [_i, _u, _i, _u, _vp, _vp, _vp]),
    ("plk_ntt_padded", _i, [_i, _u, _vp, _sz, _vp]),
    ("plk_ntt_padded_batch", _i, [_i, _u, _u, _vp, _vp, _vp]),
    ("plk_ntt_padded_dev", _i, [_i, _u, _u, _vp, _sz, _sz, _vp, _vp]),
    ("plk_poly_divide_by_z_h", _i, [_i, _vp, _sz, _sz, _vp, _sz, _vp]),
    ("plk_poly_divide_by_z_h_dev", _i, [_i, _vp, _sz, _sz, _vp, _sz, _vp, _vp]),
    ("plk_poly_mul", _i, [_i, _vp, _sz, _vp, _sz, _vp, _sz, _vp]),
    ("plk_poly_mul_dev", _i, [_i, _vp, _sz, _vp, _sz, _vp, _sz, _vp, _vp]),
    ("plk_plonk_vanishing_points_dev", _i, [_i, _u, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("plk_plonk_vanishing_points", _i, [_i, _u, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("plk_plonk_evaluate_all_constraints", _i, [_i, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("plk_msm_precompute", _i, [_i, _sz, _vp, _vp, _u, _vp]),
    ("plk_msm_precompute_dev", _i, [_i, _sz, _vp, _vp, _u, _vp, _vp]),
    ("plk_msm_precompute_ex", _i, [_i, _sz, _vp, _vp, _u, _u, _vp]),
    ("plk_msm_precompute_dev_ex", _i, [_i, _sz, _vp, _vp, _u, _u, _vp, _vp]),
    ("plk_msm_free", _i, [_vp]),
    ("plk_msm_ctx_len", _sz, [_vp]),
    ("plk_msm_ctx_window", _u, [_vp]),
    ("plk_msm_execute", _i, [_vp, _vp, _sz, _vp, _vp]),
    ("plk_msm_execute_batch", _i, [_vp, _u, _vp, _sz, _vp, _vp]),
    ("plk_msm_execute_dev", _i, [_vp, _u, _vp, _sz, _vp, _vp, _vp]),
    ("plk_msm_execute_parts_dev", _i, [_vp, _u, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("plk_msm_execute_parts_buckets_dev", _i, [_vp, _u, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("plk_msm_execute_projective", _i, [_vp, _vp, _sz, _vp, _vp]),
    ("plk_msm_execute_projective_dev", _i, [_vp, _u, _vp, _sz, _vp, _vp, _vp]),
    ("plk_msm", _i, [_i, _sz, _vp, _vp, _vp, _vp, _vp]),
    ("plk_curve_sum_affine", _i, [_i, _sz, _vp, _vp, _vp, _vp]),
    ("plk_msm_partials_bytes", _sz, [_i, _u]),
    ("plk_msm_combine_partials_dev", _i, [_i, _u, _u, _u, _vp, _vp, _vp, _vp]),
    ("plk_msm_table_digits", _i, [_i, _u]),
    ("plk_msm_precompute_table", _i, [_i, _sz, _vp, _vp, _u, _vp, _vp]),
    ("plk_msm_precompute_table_dev", _i, [_i, _sz, _vp, _vp, _u, _vp, _vp, _vp]),
    ("plk_curve_fold_pairs", _i, [_i, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("plk_curve_fold_pairs_dev", _i, [_i, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("plk_field_batch_inverse", _i, [_i, _vp, _vp, _sz]),
    ("plk_field_batch_inverse_opt", _i, [_i, _vp, _vp, _vp, _sz]),
    ("plk_field_batch_inverse_dev", _i, [_i, _vp, _vp, _vp, _sz, _vp]),
    ("plk_curve_batch_to_affine", _i, [_i, _sz, _vp, _vp, _vp, _vp]),
    ("plk_curve_batch_to_affine_dev", _i, [_i, _sz, _vp, _vp, _vp, _vp, _vp]),
    ("plk_field_to_bytes", _i, [_i, _vp, _sz, _vp]),
    ("plk_field_from_bytes", _i, [_i, _vp, _sz, _vp]),
    ("plk_curve_point_to_bytes", _i, [_i, _vp, _vp, _sz, _vp]),
    ("plk_curve_point_from_bytes", _i, [_i, _vp, _sz, _vp, _vp, _vp]),
    ("plk_field_inner_product_dev", _i, [_i, _vp, _vp, _sz, _vp, _vp]),
    ("plk_field_fold_slices_dev", _i, [_i, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    ("plk_halo_begin_dev", _i, [_i, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _u, _vp, _vp]),
    ("plk_halo_begin_tabled_dev", _i, [_i, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _sz, _vp, _u, _u, _vp, _vp]),
    ("plk_curve_fold_multi_dev", _i, [_i, _sz, _u, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("plk_halo_begin", _i, [_i, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _u, _vp]),
    ("plk_halo_round_lr", _i, [_vp, _vp, _vp, _vp, _vp]),
    ("plk_halo_round_fold", _i, [_vp, _vp, _vp]),
    ("plk_halo_len", _sz, [_vp]),
    ("plk_halo_frozen", _i, [_vp]),
    ("plk_halo_read", _i, [_vp, _vp, _vp, _vp, _vp]),
    ("plk_halo_free", _i, [_vp]),
    ("plk_selftest_quad", _i, [_i, _vp, _sz, _u, _vp]),
    ("plk_checked_build", _i, []),
    ("plk_checked_failures", _i, [_vp]),
    ("plk_ntt_set_profiling", _i, [_i]),
    ("plk_ntt_get_timings", _i, [_vp, _vp]),
    ("plk_msm_set_profiling", _i, [_vp, _i]),
    ("plk_msm_get_timings", _i, [_vp, _vp, _vp]),
    ("plk_bench_ceilings", _i, [_vp, _u]),
    ("plk_msm_debug_digits", _i, [_i, _u, _sz, _vp, _vp, _vp]),
    ("plk_field_op", _i, [_i, _i, _vp, _vp, _vp, _sz]),
    ("plk_curve_gen_bases_dev", _i, [_i, _sz, _u64, _vp, _vp, _vp, _vp]),
]


class PlonkyHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libplonky_hip error %d: %s" % (code, msg))
        self.code = code


def _source_hash():
    """sha256 over everything the two libraries are built from (sources, headers, Makefile)."""
    import hashlib
    h = hashlib.sha256()
    names = sorted(f for f in os.listdir(_CSRC) if f.endswith((".hip", ".cuh", ".h", ".cpp")) or f == "Makefile")
    for f in names + [os.path.join("..", "..", "include", "plonky_hip.h")]:
        h.update(f.encode())
        with open(os.path.join(_CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


_STAMP = os.path.join(_CSRC, ".build_stamp")


def build(force=False):
    """Compile libplonky_hip.so and its checked twin (-DPLK_CHECKED) for gfx950 in-tree (hipcc cross-compiles without a GPU).
    A stamp with the hash of the sources sits next to the libraries: a tree whose libraries were built from exactly these
    sources (the snapshot on a GPU lease, which carries the .so files but not the objects) is left alone."""
    want = _source_hash()
    have = open(_STAMP).read().strip() if os.path.exists(_STAMP) else ""
    libs = [os.path.join(_CSRC, "libplonky_hip.so"), os.path.join(_CSRC, "libplonky_hip_checked.so")]
    if force or have != want or not all(os.path.exists(p) for p in libs):
        args = ["make", "-C", _CSRC, "-j8", "all", "checked"]
        if force:
            args.append("-B")
        subprocess.check_call(args, stdout=subprocess.DEVNULL)
        with open(_STAMP, "w") as fh:
            fh.write(want + "\n")
    build_host_harness()
    return SO_PATH


def build_host_harness():
    """tests/capi_host: a C++ host program above the C ABI only (what the Rust shim of INTEGRATION.md does), run by the GPU suite."""
    root = os.path.dirname(os.path.dirname(_CSRC))
    src, out = os.path.join(root, "tests", "capi_host.cpp"), os.path.join(root, "tests", "capi_host")
    if not os.path.exists(src):
        return None
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(src), os.path.getmtime(SO_PATH)):
        return out
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", src, "-o", out, "-L" + _CSRC, "-lplonky_hip",
                           "-Wl,-rpath,$ORIGIN/../plonky_amd/csrc", "-Wl,-rpath,/opt/rocm/lib"])
    return out


_lib = None


def load():
    """Load the library; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(the HIP path has no CPU fallback)" % SO_PATH)
        # One HIP runtime per process: torch bundles its own libamdhip64.so.7 and hands us its
        # streams and device pointers, so it must be loaded first; libplonky_hip.so then binds to
        # the already-loaded runtime (same SONAME) instead of pulling in a second copy.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(SO_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != PLK_OK:
        raise PlonkyHipError(rc, load().plk_last_error().decode("utf-8", "replace"))
    return rc
