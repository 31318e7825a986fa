"""ctypes binding of tests/emu/libmsm_emu.so (CPU emulation of the HIP pipeline). Test-only."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CURVE_ID = {"bls12_381_g1": 0, "bls12_381_g2": 1, "bn254_snarks_g1": 2, "bn254_snarks_g2": 3, "pallas": 4, "vesta": 5}
AFF_BYTES = {"bls12_381_g1": 96, "bls12_381_g2": 192, "bn254_snarks_g1": 64, "bn254_snarks_g2": 128, "pallas": 64, "vesta": 64}
_lib = None


def build():
    if os.environ.get("EMU_LIB"):   # an emulator built elsewhere with other options (experiments)
        return os.environ["EMU_LIB"]
    jobs = str(max(1, min(8, os.cpu_count() or 1)))
    # one builder at a time: the two ranks of tests/test_dist_gloo.py import this module at the same moment, and two `make`s writing
    # the same objects handed one of them a half-written library once (the test then waited out its 300 s queue timeout)
    import fcntl
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        subprocess.check_call(["make", "-s", "-j", jobs, "-C", HERE])
    return os.path.join(HERE, "libmsm_emu.so")


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        vp, sz, i32, u64, u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint32
        L.emu_msm.argtypes = [i32, i32, i32, vp, vp, vp, sz, i32, i32, i32, vp]
        L.emu_gen_points.argtypes = [i32, u64, u64, u32, vp]
        L.emu_field_op.argtypes = [i32, i32, vp, vp, vp]
        L.emu_field_op_dev.argtypes = [i32, i32, vp, vp, vp]
        L.emu_dev_field_info.argtypes = [i32, vp, vp]
        L.emu_msm_host.argtypes = [i32, i32, i32, vp, vp, vp, sz, i32, i32]
        L.emu_msm_table.argtypes = [i32, i32, i32, vp, vp, vp, sz, sz, i32, i32, i32]
        L.emu_sum_reduce.argtypes = [i32, i32, vp, vp, sz, i32]
        L.emu_batch_affine.argtypes = [i32, i32, vp, vp, sz, i32]
        L.emu_msm_slots.argtypes = [i32, vp, vp, vp, sz]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def msm(curve, coefs, points, coef_is_fr=False, out_kind=0, c=0, K=0, S=0):
    coefs = np.ascontiguousarray(coefs, dtype=np.uint8)
    points = np.ascontiguousarray(points, dtype=np.uint8)
    n = coefs.shape[0]
    nco = 2 if out_kind == 0 else 3
    out = np.zeros(AFF_BYTES[curve] // 2 * nco, dtype=np.uint8)
    plan = np.zeros(8, dtype=np.int32)
    rc = lib().emu_msm(CURVE_ID[curve], int(coef_is_fr), out_kind, _p(out), _p(coefs), _p(points), n, c, K, S, _p(plan))
    assert rc == 0
    return out, plan


def msm_slots(curve, coefs, points):
    """Three MSMs over the same input with tickets finished out of order (submit A, complete B, complete C, finish A).
    Returns (three affine results, refused submits)."""
    coefs = np.ascontiguousarray(coefs, dtype=np.uint8)
    points = np.ascontiguousarray(points, dtype=np.uint8)
    out = np.zeros((3, AFF_BYTES[curve]), dtype=np.uint8)
    refused = lib().emu_msm_slots(CURVE_ID[curve], _p(out), _p(coefs), _p(points), coefs.shape[0])
    return out, refused


def fr_quotient(curve, poly, dom_mont, z_mont, scale_mont, K=8):
    """The per-lane bodies of ctt_hip_fr_quotient on the CPU: (q canonical (n, 32), y canonical (32,))."""
    poly = np.ascontiguousarray(poly, dtype=np.uint8)
    dom_mont = np.ascontiguousarray(dom_mont, dtype=np.uint8)
    n = poly.shape[0]
    q = np.zeros((n, 32), dtype=np.uint8)
    y = np.zeros(32, dtype=np.uint8)
    L = lib()
    L.emu_fr_quotient.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    zm = np.ascontiguousarray(z_mont, dtype=np.uint8)
    sm = np.ascontiguousarray(scale_mont, dtype=np.uint8)
    assert L.emu_fr_quotient(CURVE_ID[curve], _p(poly), _p(dom_mont), _p(zm), _p(sm), n, K, _p(q), _p(y)) == 0
    return q, y


def msm_host(curve, coefs, points, coef_is_fr=False, out_kind=0, c=0, chunks=0):
    """The host-pointer form of the engine (MsmEngine::submit_host): inputs uploaded in slices, one bucket set per slice.
    Returns (result, slices used)."""
    coefs = np.ascontiguousarray(coefs, dtype=np.uint8)
    points = np.ascontiguousarray(points, dtype=np.uint8)
    n = coefs.shape[0]
    nco = 2 if out_kind == 0 else 3
    out = np.zeros(AFF_BYTES[curve] // 2 * nco, dtype=np.uint8)
    used = lib().emu_msm_host(CURVE_ID[curve], int(coef_is_fr), out_kind, _p(out), _p(coefs), _p(points), n, c, chunks)
    assert used >= 1
    return out, used


def msm_table(curve, coefs, points, coef_is_fr=False, out_kind=0, c=0, K=0, chunks=0):
    """Cached bases with a window table over all of `points` (MsmEngine::prepare_table); the MSM uses the first len(coefs).
    chunks > 0: host-resident coefficients uploaded in that many slices (MsmEngine::submit_host with cached bases).
    Returns (result, window bits of the table)."""
    coefs = np.ascontiguousarray(coefs, dtype=np.uint8)
    points = np.ascontiguousarray(points, dtype=np.uint8)
    nco = 2 if out_kind == 0 else 3
    out = np.zeros(AFF_BYTES[curve] // 2 * nco, dtype=np.uint8)
    cu = lib().emu_msm_table(CURVE_ID[curve], int(coef_is_fr), out_kind, _p(out), _p(coefs), _p(points), points.shape[0],
                             coefs.shape[0], c, K, chunks)
    assert cu >= 0
    return out, cu


def plan(n, bits, lanes, table_c=0, ntab=0):
    """The engine's plan for n pairs (make_plan / make_table_plan) as a dict."""
    out = np.zeros(16, dtype=np.uint32)
    L = lib()
    L.emu_plan.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_uint32, ctypes.c_void_p]
    L.emu_plan(n, bits, lanes, table_c, ntab, _p(out))
    return dict(zip(("c", "W", "Wd", "B", "K", "G", "S", "slice", "NG", "gshift", "nent", "cb", "r", "merge_steps"), (int(x) for x in out)))


def table_window_bits(ntab, bits):
    return lib().emu_table_window_bits(ctypes.c_uint32(ntab), bits)


def gen_points(curve, seed, n, first=0):
    out = np.zeros((n, AFF_BYTES[curve]), dtype=np.uint8)
    lib().emu_gen_points(CURVE_ID[curve], seed, first, n, _p(out))
    return out


def field_op(curve, op, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = a if b is None else np.ascontiguousarray(b, dtype=np.uint8)
    out = np.zeros_like(a)
    lib().emu_field_op(CURVE_ID[curve], op, _p(a), _p(b), _p(out))
    return out


def dev_field_info(curve):
    """(limb_bits, limb_count) of the curve's device field when it is the carry-free one, else None."""
    lb, nl = ctypes.c_int(0), ctypes.c_int(0)
    ok = lib().emu_dev_field_info(CURVE_ID[curve], ctypes.byref(lb), ctypes.byref(nl))
    return (lb.value, nl.value) if ok else None


def field_op_dev(curve, op, a, b=None):
    """Device-field probe: a, b in the reference representation; returns raw device limbs (uint32[NL])."""
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = a if b is None else np.ascontiguousarray(b, dtype=np.uint8)
    out = np.zeros(64, dtype=np.uint32)
    nl = lib().emu_field_op_dev(CURVE_ID[curve], op, _p(a), _p(b), _p(out))
    return out[:nl]


def sum_reduce(curve, points, out_kind=0, K=0):
    points = np.ascontiguousarray(points, dtype=np.uint8).reshape(-1, AFF_BYTES[curve])
    nco = 2 if out_kind == 0 else 3
    out = np.zeros(AFF_BYTES[curve] // 2 * nco, dtype=np.uint8)
    k = lib().emu_sum_reduce(CURVE_ID[curve], out_kind, _p(out), _p(points), points.shape[0], K)
    assert k >= 0
    return out


def batch_affine(curve, src, src_kind=1, K=8):
    """src: [n][3 coordinates] Jacobian (src_kind 1) or projective (2) -> [n] affine."""
    src = np.ascontiguousarray(src, dtype=np.uint8).reshape(-1, AFF_BYTES[curve] // 2 * 3)
    out = np.zeros((src.shape[0], AFF_BYTES[curve]), dtype=np.uint8)
    assert lib().emu_batch_affine(CURVE_ID[curve], src_kind, _p(out), _p(src), src.shape[0], K) == 0
    return out
