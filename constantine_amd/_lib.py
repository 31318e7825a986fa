"""ctypes loader for libctt_msm_hip.so. The product path has no fallback: a missing library is an error."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CTT_MSM_HIP_LIB") or os.path.join(HERE, "libctt_msm_hip.so")

_lib = None
ABI_VERSION = 9  # ctt_hip_msm_abi_version() of the library this package was written against


class HipLibraryMissing(RuntimeError):
    pass


def exported_symbols():
    """Every symbol include/ctt_msm_hip.h declares."""
    syms = []
    for stem, par in (("bls12_381_g1", True), ("bls12_381_g2", False), ("bn254_snarks_g1", True),
                      ("bn254_snarks_g2", False), ("pallas_ec", True), ("vesta_ec", True)):
        for coord in ("jac", "prj"):
            for coef in ("big", "fr"):
                syms.append(f"ctt_{stem}_{coord}_multi_scalar_mul_{coef}_coefs_vartime")
                if par:
                    syms.append(f"ctt_{stem}_{coord}_multi_scalar_mul_{coef}_coefs_vartime_parallel")
            syms.append(f"ctt_{stem}_{coord}_batch_affine")
            for coef in ("big", "fr"):
                syms.append(f"ctt_hip_msm_{stem}_{coord}_{coef}")   # neutral spellings (header part 1c)
    syms += ["ctt_hip_sum_reduce", "ctt_hip_batch_affine", "ctt_hip_msm_abi_version", "ctt_hip_msm_ctx_create", "ctt_hip_msm_ctx_destroy", "ctt_hip_msm_set_option",
             "ctt_hip_msm_device", "ctt_hip_msm_device_submit", "ctt_hip_msm_device_finish", "ctt_hip_msm_sync", "ctt_hip_msm_bases_create", "ctt_hip_msm_bases_destroy",
             "ctt_hip_msm_with_bases", "ctt_hip_msm_with_bases_submit", "ctt_hip_msm_bases_create_table", "ctt_hip_msm_bases_window_bits", "ctt_hip_msm_last_timings", "ctt_hip_msm_last_plan", "ctt_hip_gen_points",
             "ctt_hip_field_op", "ctt_hip_ec_sum_affine", "ctt_hip_msm_stream", "ctt_hip_msm_wait_stream",
             "ctt_hip_msm_set_devices", "ctt_hip_msm_set_shard_min", "ctt_hip_subgroup_check", "ctt_hip_fr_quotient",
             "ctt_hip_msm_host", "ctt_hip_msm_available", "ctt_hip_last_error", "ctt_hip_last_error_message", "ctt_hip_clear_last_error",
             # part 3: the MSM's callers under the reference's names + their host-only pieces
             "ctt_eth_kzg_context_new", "ctt_eth_kzg_context_new_with_precompute", "ctt_eth_kzg_context_delete", "ctt_eth_kzg_blob_to_kzg_commitment", "ctt_eth_kzg_compute_kzg_proof",
             "ctt_eth_kzg_compute_blob_kzg_proof", "ctt_eth_kzg_blob_to_kzg_commitment_parallel", "ctt_eth_kzg_compute_kzg_proof_parallel",
             "ctt_eth_kzg_compute_blob_kzg_proof_parallel", "ctt_eth_evm_bls12381_g1msm", "ctt_eth_evm_bls12381_g2msm",
             "ctt_hip_eth_kzg_context_from_srs", "ctt_hip_sha256", "ctt_hip_bls12_381_g1_decompress", "ctt_hip_bls12_381_g1_compress",
             "ctt_hip_eth_kzg_blob_to_scalars", "ctt_hip_eth_kzg_challenge", "ctt_hip_eth_kzg_quotient_host"]
    return syms


GPU_UNAVAILABLE = 0xF0   # CTT_HIP_STATUS_GPU_UNAVAILABLE: what a protocol symbol returns when the GPU cannot serve the call


class GpuUnavailable(RuntimeError):
    """A call the GPU could not serve (no device, out of device memory, a failed HIP call, both in-flight slots taken): the
    library reports it through the call's return value; `code` / the message are ctt_hip_last_error() / _message() of this thread.
    There is no CPU path inside the library -- what to do instead is the caller's decision."""

    def __init__(self, what=""):
        L = lib()
        self.code = int(L.ctt_hip_last_error())
        msg = L.ctt_hip_last_error_message()
        super().__init__(f"{what}: GPU unavailable ({self.code}): {msg.decode(errors='replace') if msg else ''}")


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64 and asks for it by the unversioned name,
    so if this library pulled in /opt/rocm's copy first, torch would load a second runtime that finds no GPU.  When
    torch is installed, load ITS runtime first (same SONAME libamdhip64.so.7 -> our NEEDED entry binds to it)."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C constantine_amd/csrc). There is no CPU fallback.")
    _share_hip_runtime_with_torch()
    L = ctypes.CDLL(LIB_PATH)
    vp, sz, i32, u32, u64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64
    # an OLDER build named explicitly for a same-box comparison (tools/ab_prev.sh) may predate a symbol: only with the explicit
    # opt-in CTT_MSM_HIP_ALLOW_OLD_ABI=1 is a missing symbol skipped; any other library must export what the header declares
    allow_old = bool(os.environ.get("CTT_MSM_HIP_LIB")) and os.environ.get("CTT_MSM_HIP_ALLOW_OLD_ABI") == "1"
    missing = []
    for name in exported_symbols():
        if not hasattr(L, name):
            if allow_old:
                missing.append(name)
                continue
            raise AttributeError(f"{LIB_PATH} does not export {name} (include/ctt_msm_hip.h declares it); an older build "
                                 "for a comparison needs CTT_MSM_HIP_ALLOW_OLD_ABI=1")
        fn = getattr(L, name)
        if name.startswith("ctt_hip_msm_") and name.rsplit("_", 2)[-2] in ("jac", "prj") and name.rsplit("_", 1)[-1] in ("big", "fr"):
            fn.argtypes = [vp, vp, vp, sz]
            fn.restype = i32
        elif name.endswith("_vartime"):
            fn.argtypes = [vp, vp, vp, sz]
            fn.restype = None
        elif name.endswith("_vartime_parallel"):
            fn.argtypes = [vp, vp, vp, vp, sz]
            fn.restype = None
        elif name.endswith("_batch_affine") and not name.startswith("ctt_hip"):
            fn.argtypes = [vp, vp, sz]
            fn.restype = None
    if missing:
        for name in missing:   # calls of a symbol the old build lacks fail loudly instead of running with default prototypes
            def _absent(*a, _n=name, **k):
                raise AttributeError(f"{LIB_PATH} (old ABI) has no {_n}")
            setattr(L, name, _absent)
    if "ctt_hip_msm_host" not in missing:
        L.ctt_hip_msm_host.argtypes = [i32, i32, i32, vp, vp, vp, sz]
        L.ctt_hip_msm_host.restype = i32
        L.ctt_hip_msm_available.argtypes = []
        L.ctt_hip_msm_available.restype = i32
    if "ctt_hip_last_error" not in missing:
        L.ctt_hip_last_error.argtypes = []
        L.ctt_hip_last_error.restype = i32
        L.ctt_hip_last_error_message.argtypes = []
        L.ctt_hip_last_error_message.restype = ctypes.c_char_p
        L.ctt_hip_clear_last_error.argtypes = []
        L.ctt_hip_clear_last_error.restype = None
    if "ctt_eth_kzg_context_new" not in missing:
        u8 = ctypes.c_uint8   # the reference's status enums are __attribute__((__packed__)): one byte
        L.ctt_eth_kzg_context_new.argtypes = [ctypes.POINTER(vp), ctypes.c_char_p, u8]
        L.ctt_eth_kzg_context_new.restype = u8
        L.ctt_eth_kzg_context_new_with_precompute.argtypes = [ctypes.POINTER(vp), ctypes.c_char_p, u8, i32, i32]
        L.ctt_eth_kzg_context_new_with_precompute.restype = u8
        L.ctt_eth_kzg_context_delete.argtypes = [vp]
        L.ctt_eth_kzg_context_delete.restype = None
        L.ctt_eth_kzg_blob_to_kzg_commitment.argtypes = [vp, vp, vp]
        L.ctt_eth_kzg_blob_to_kzg_commitment.restype = u8
        L.ctt_eth_kzg_compute_kzg_proof.argtypes = [vp, vp, vp, vp, vp]
        L.ctt_eth_kzg_compute_kzg_proof.restype = u8
        L.ctt_eth_kzg_compute_blob_kzg_proof.argtypes = [vp, vp, vp, vp]
        L.ctt_eth_kzg_compute_blob_kzg_proof.restype = u8
        for nm in ("ctt_eth_kzg_blob_to_kzg_commitment", "ctt_eth_kzg_compute_kzg_proof", "ctt_eth_kzg_compute_blob_kzg_proof"):
            par = getattr(L, nm + "_parallel")       # (tp, ...) -- ethereum_eip4844_kzg_parallel.h
            par.argtypes = [vp] + list(getattr(L, nm).argtypes)
            par.restype = u8
        for nm in ("ctt_eth_evm_bls12381_g1msm", "ctt_eth_evm_bls12381_g2msm"):
            getattr(L, nm).argtypes = [vp, sz, vp, sz]
            getattr(L, nm).restype = u8
        L.ctt_hip_eth_kzg_context_from_srs.argtypes = [ctypes.POINTER(vp), vp, sz, i32, i32]
        L.ctt_hip_eth_kzg_context_from_srs.restype = i32
        L.ctt_hip_sha256.argtypes = [vp, vp, sz]
        L.ctt_hip_sha256.restype = None
        L.ctt_hip_bls12_381_g1_decompress.argtypes = [vp, vp]
        L.ctt_hip_bls12_381_g1_decompress.restype = i32
        L.ctt_hip_bls12_381_g1_compress.argtypes = [vp, vp]
        L.ctt_hip_bls12_381_g1_compress.restype = None
        L.ctt_hip_eth_kzg_blob_to_scalars.argtypes = [vp, vp]
        L.ctt_hip_eth_kzg_blob_to_scalars.restype = i32
        L.ctt_hip_eth_kzg_challenge.argtypes = [vp, vp, vp]
        L.ctt_hip_eth_kzg_challenge.restype = None
        L.ctt_hip_eth_kzg_quotient_host.argtypes = [vp, vp, vp, vp]
        L.ctt_hip_eth_kzg_quotient_host.restype = None
    L.ctt_hip_sum_reduce.argtypes = [vp, i32, i32, vp, vp, sz, i32]
    L.ctt_hip_sum_reduce.restype = i32
    L.ctt_hip_batch_affine.argtypes = [vp, i32, i32, vp, vp, sz, i32]
    L.ctt_hip_batch_affine.restype = i32
    L.ctt_hip_msm_abi_version.restype = i32
    L.ctt_hip_msm_ctx_create.argtypes = [i32]
    L.ctt_hip_msm_ctx_create.restype = vp
    L.ctt_hip_msm_ctx_destroy.argtypes = [vp]
    L.ctt_hip_msm_ctx_destroy.restype = None
    L.ctt_hip_msm_set_option.argtypes = [vp, ctypes.c_char_p, i32]
    L.ctt_hip_msm_device.argtypes = [vp, i32, i32, i32, vp, vp, vp, sz]
    L.ctt_hip_msm_device_submit.argtypes = [vp, i32, i32, vp, vp, sz]
    L.ctt_hip_msm_device_finish.argtypes = [vp, i32, i32, vp]
    L.ctt_hip_msm_bases_create.argtypes = [vp, i32, vp, sz, i32]
    L.ctt_hip_msm_bases_create.restype = vp
    L.ctt_hip_msm_bases_create_table.argtypes = [vp, i32, vp, sz, i32, i32]
    L.ctt_hip_msm_bases_create_table.restype = vp
    L.ctt_hip_msm_bases_window_bits.argtypes = [vp]
    L.ctt_hip_msm_bases_window_bits.restype = i32
    L.ctt_hip_msm_with_bases_submit.argtypes = [vp, vp, i32, vp, sz]
    L.ctt_hip_msm_with_bases_submit.restype = i32
    L.ctt_hip_msm_bases_destroy.argtypes = [vp, vp]
    L.ctt_hip_msm_bases_destroy.restype = None
    L.ctt_hip_msm_with_bases.argtypes = [vp, vp, i32, i32, vp, vp, sz, i32]
    L.ctt_hip_msm_sync.argtypes = [vp]
    L.ctt_hip_msm_sync.restype = None
    L.ctt_hip_msm_last_timings.argtypes = [vp, vp, i32]
    L.ctt_hip_msm_last_plan.argtypes = [vp, vp, i32]
    L.ctt_hip_gen_points.argtypes = [vp, i32, u64, u64, u32, vp]
    L.ctt_hip_field_op.argtypes = [vp, i32, i32, vp, vp, vp, u32]
    L.ctt_hip_ec_sum_affine.argtypes = [i32, i32, vp, vp, sz]
    L.ctt_hip_msm_stream.argtypes = [vp]
    L.ctt_hip_msm_stream.restype = vp
    L.ctt_hip_msm_wait_stream.argtypes = [vp, vp]
    L.ctt_hip_msm_set_devices.argtypes = [ctypes.POINTER(i32), i32]
    L.ctt_hip_msm_set_shard_min.argtypes = [sz]
    L.ctt_hip_msm_set_shard_min.restype = None
    L.ctt_hip_subgroup_check.argtypes = [vp, i32, vp, vp, sz, i32]
    if "ctt_hip_fr_quotient" not in missing:
        L.ctt_hip_fr_quotient.argtypes = [vp, i32, vp, vp, vp, vp, vp, u32]
        L.ctt_hip_fr_quotient.restype = i32
    _lib = L
    return L
