"""ctypes binding of oracle/libmsm_ref.so (CPU restatement in C++). TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CURVE_ID = {"bls12_381_g1": 0, "bls12_381_g2": 1, "bn254_snarks_g1": 2, "bn254_snarks_g2": 3, "pallas": 4, "vesta": 5}
AFF_BYTES = {"bls12_381_g1": 96, "bls12_381_g2": 192, "bn254_snarks_g1": 64, "bn254_snarks_g2": 128, "pallas": 64, "vesta": 64}

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "libmsm_ref.so"])


def build_native():
    """Rebuild the same source for the CPU this process runs on (g++ -march=native) and switch to that build: the shipped
    libmsm_ref.so targets x86-64-v3 so that it runs on any box.  Returns the flags of the build now in use."""
    global _lib
    path = os.path.join(HERE, "libmsm_ref_native.so")
    try:
        subprocess.check_call(["make", "-s", "-B", "-C", HERE, "libmsm_ref_native.so"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _lib = None
        lib(path)
        return "g++ -O3 -march=native (built on this host)"
    except Exception:
        _lib = None
        lib()
        return "g++ -O3 -march=x86-64-v3 (shipped build; native rebuild failed)"


def lib(path=None):
    global _lib
    if _lib is None:
        path = path or os.path.join(HERE, "libmsm_ref.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        vp, sz, i32, u64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint64
        L.oracle_msm.argtypes = [i32, vp, vp, sz, i32, i32, vp]
        L.oracle_msm.restype = i32
        L.oracle_fr_from_mont.argtypes = [i32, vp, vp, sz]
        L.oracle_fr_to_mont.argtypes = [i32, vp, vp, sz]
        L.oracle_gen_points.argtypes = [i32, u64, sz, sz, vp, i32]
        L.oracle_scalar_mul.argtypes = [i32, vp, vp, vp]
        L.oracle_best_bucket_bit_size.argtypes = [sz, i32]
        L.oracle_gen_points_unknown_log.argtypes = [i32, u64, sz, sz, vp, i32]
        L.oracle_psi_g2.argtypes = [i32, vp, vp]
        L.oracle_decompose_g2.argtypes = [i32, vp, vp, vp]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def msm(curve: str, scalars: np.ndarray, points: np.ndarray, nthreads: int = 1, c: int = 0):
    """scalars: (n,32) u8 canonical LE; points: (n,AFF) u8 Montgomery affine. -> (affine bytes, c used)"""
    scalars = np.ascontiguousarray(scalars, dtype=np.uint8)
    points = np.ascontiguousarray(points, dtype=np.uint8)
    n = scalars.shape[0]
    assert points.shape[0] == n
    out = np.zeros(AFF_BYTES[curve], dtype=np.uint8)
    used = lib().oracle_msm(CURVE_ID[curve], _ptr(scalars), _ptr(points), n, nthreads, c, _ptr(out))
    assert used >= 0
    return out, used


def gen_points(curve: str, seed: int, n: int, first: int = 0, nthreads: int = 0) -> np.ndarray:
    out = np.zeros((n, AFF_BYTES[curve]), dtype=np.uint8)
    nt = nthreads or (os.cpu_count() or 1)
    lib().oracle_gen_points(CURVE_ID[curve], seed & (2**64 - 1), first, n, _ptr(out), nt)
    return out


def fr_to_mont(curve: str, a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint8)
    out = np.zeros_like(a)
    lib().oracle_fr_to_mont(CURVE_ID[curve], _ptr(a), _ptr(out), a.shape[0])
    return out


def fr_from_mont(curve: str, a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint8)
    out = np.zeros_like(a)
    lib().oracle_fr_from_mont(CURVE_ID[curve], _ptr(a), _ptr(out), a.shape[0])
    return out


def scalar_mul(curve: str, k: np.ndarray, p: np.ndarray) -> np.ndarray:
    k = np.ascontiguousarray(k, dtype=np.uint8)
    p = np.ascontiguousarray(p, dtype=np.uint8)
    out = np.zeros(AFF_BYTES[curve], dtype=np.uint8)
    lib().oracle_scalar_mul(CURVE_ID[curve], _ptr(k), _ptr(p), _ptr(out))
    return out


def gen_points_unknown_log(curve: str, seed: int, n: int, first: int = 0, nthreads: int = 0) -> np.ndarray:
    """Random points of the prime-order subgroup with UNKNOWN discrete logarithms (random x, square root, cofactor clearing: the
    reference's bench inputs, msm_ref.cpp gen_points_unknown_log); all six curves of the path."""
    out = np.zeros((n, AFF_BYTES[curve]), dtype=np.uint8)
    nt = nthreads or (os.cpu_count() or 1)
    assert lib().oracle_gen_points_unknown_log(CURVE_ID[curve], seed & (2**64 - 1), first, n, _ptr(out), nt) == 0
    return out


def psi_g2(curve: str, p: np.ndarray) -> np.ndarray:
    """psi(P) of the M = 4 endomorphism pre-split on G2 (msm_ref.cpp psi)"""
    p = np.ascontiguousarray(p, dtype=np.uint8)
    out = np.zeros(AFF_BYTES[curve], dtype=np.uint8)
    assert lib().oracle_psi_g2(CURVE_ID[curve], _ptr(p), _ptr(out)) == 0
    return out


def decompose_g2(curve: str, k: int):
    """-> ([m0, m1, m2, m3], [neg0..neg3]): the port's M = 4 decomposition of the scalar k (decomposeEndo, split_scalars.nim:37-123)"""
    kb = np.frombuffer(int(k).to_bytes(32, "little"), dtype=np.uint8).copy()
    mini = np.zeros(4 * 32, dtype=np.uint8)
    neg = (ctypes.c_int * 4)()
    assert lib().oracle_decompose_g2(CURVE_ID[curve], _ptr(kb), _ptr(mini), neg) == 0
    return [int.from_bytes(bytes(mini[32 * j:32 * j + 32]), "little") for j in range(4)], [bool(neg[j]) for j in range(4)]


def synth_scalars(seed: int, n: int, bits: int, first: int = 0) -> np.ndarray:
    """Vectorised pyoracle.synth_scalar: (n,32) u8, uniform in [0,2^bits), not reduced."""
    idx = (np.arange(first, first + n, dtype=np.uint64)[:, None] * np.uint64(4) + np.arange(4, dtype=np.uint64)[None, :])
    with np.errstate(over="ignore"):
        x = idx + np.uint64(seed & (2**64 - 1))
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    top = bits - 192
    z[:, 3] &= np.uint64((1 << top) - 1)
    return z.view(np.uint8).reshape(n, 32).copy()

def gen_point_scalars(seed: int, n: int, first: int = 0) -> np.ndarray:
    """The discrete logs of the synthetic points: gen_points(name, seed, n)[i] = [s_i]G with s_i = pyoracle.synth_scalar(seed ^ 0xA5A5..,
    i, 128) | 1 (pyoracle.synth_point, msm_ref.cpp gen_points, msm_bodies.h gen_point_body).  (n, 16) u8, little-endian."""
    idx = (np.arange(first, first + n, dtype=np.uint64)[:, None] * np.uint64(4) + np.arange(2, dtype=np.uint64)[None, :])
    with np.errstate(over="ignore"):
        x = idx + np.uint64((seed ^ 0xA5A5A5A5A5A5A5A5) & (2**64 - 1))
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    z[:, 0] |= np.uint64(1)
    return z.view(np.uint8).reshape(n, 16).copy()


def msm_by_discrete_logs(name: str, point_seed: int, scalars: np.ndarray, first: int = 0):
    """MSM of the synthetic points gen_points(name, point_seed, n) without any elliptic-curve arithmetic but ONE scalar multiplication of
    the big-integer oracle (pinned by the reference's scalar-mul vectors): sum a_i P_i = [sum a_i s_i mod r] G.  An answer for 2^20 .. 2^24
    pairs that owes nothing to the C++ port (msm_ref.cpp) the GPU is otherwise compared with.  -> affine point of pyoracle (or None).
    The dot product runs on 16-bit limbs as a float64 matrix product over blocks of 2^20 rows (a product < 2^32, any partial sum < 2^52:
    exact whatever the order of summation)."""
    from . import pyoracle as po
    curve = po.CURVES[name]
    sc = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
    n = sc.shape[0]
    logs = gen_point_scalars(point_seed, n, first)
    t = 0
    for lo in range(0, n, 1 << 20):
        a = sc[lo:lo + (1 << 20)].view("<u2").astype(np.float64)         # (m, 16)
        b = logs[lo:lo + (1 << 20)].view("<u2").astype(np.float64)       # (m, 8)
        m = a.T @ b                                                      # (16, 8): every partial sum < 2^52, exact in binary64
        for j in range(16):
            for k in range(8):
                t += int(m[j, k]) << (16 * (j + k))
    return curve.scalar_mul(t % curve.order, curve.gen)
