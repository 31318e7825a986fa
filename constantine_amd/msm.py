"""
Host-side mirror of the reference's MSM entry points, over the C ABI of libctt_msm_hip.so.

Names and argument meaning follow the reference:
  multiScalarMul_vartime(r, coefs, points)                 constantine/math/elliptic/ec_multi_scalar_mul.nim:525-568
  multiScalarMul_vartime_parallel(tp, r, coefs, points)    constantine/math/elliptic/ec_multi_scalar_mul_parallel.nim:588-639
  CttEngine.msm(coeffs, bases)                             constantine-rust/constantine-halo2-zal/src/lib.rs:22-58
    (Halo2-ZAL MsmAccel: BN254 G1, Fr coefficients, projective result; descriptors at lib.rs:60-95)

Arrays are numpy uint8 buffers in the C-API memory layout (include/ctt_msm_hip.h): coefs (n,32) --
BigInt canonical little-endian, or Fr Montgomery when fr_coefs=True -- and points (n, 2*coord) affine
Montgomery.  Like the reference, nothing is validated (no on-curve / subgroup checks) and the functions are
variable-time: public inputs only.  coefs and points must have the same length (the reference asserts it in
debug builds, ec_multi_scalar_mul.nim:536).
"""
import ctypes

import numpy as np

from . import _lib
from .curves import COEF_BIG, COEF_FR, CURVES, OUT_AFF, OUT_JAC, OUT_PRJ

_COORD = {"jac": OUT_JAC, "prj": OUT_PRJ, "aff": OUT_AFF}


def _check(curve, coefs, points):
    info = CURVES[curve]
    coefs = np.ascontiguousarray(coefs, dtype=np.uint8)
    points = np.ascontiguousarray(points, dtype=np.uint8)
    if coefs.ndim != 2 or coefs.shape[1] != 32:
        raise ValueError("coefs must have shape (n, 32)")
    if points.ndim != 2 or points.shape[1] != info.aff_bytes:
        raise ValueError(f"points must have shape (n, {info.aff_bytes})")
    if coefs.shape[0] != points.shape[0]:
        raise ValueError("coefs and points must have the same length")
    return info, coefs, points


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def multiScalarMul_vartime(curve, coefs, points, coord="jac", fr_coefs=False):
    """r <- sum coefs[i]*points[i], via ctt_<curve>_<coord>_multi_scalar_mul_{big,fr}_coefs_vartime.
    Returns r as a uint8 array of 3 coordinates (X,Y,Z) in `coord` ("jac" or "prj")."""
    info, coefs, points = _check(curve, coefs, points)
    L = _lib.lib()
    fn = getattr(L, f"ctt_{info.sym}_{coord}_multi_scalar_mul_{'fr' if fr_coefs else 'big'}_coefs_vartime")
    r = np.zeros(info.jac_bytes, dtype=np.uint8)
    fn(_ptr(r), _ptr(coefs), _ptr(points), coefs.shape[0])
    return r


def multiScalarMul_vartime_parallel(tp, curve, coefs, points, coord="jac", fr_coefs=False):
    """Same through the *_vartime_parallel symbol; `tp` (a ctt_threadpool* or None) is accepted and ignored,
    the GPU replaces the pool.  Groups without an upstream parallel symbol (G2) raise AttributeError."""
    info, coefs, points = _check(curve, coefs, points)
    if not info.has_parallel:
        raise AttributeError(f"the reference exports no *_vartime_parallel MSM for {curve}")
    L = _lib.lib()
    fn = getattr(L, f"ctt_{info.sym}_{coord}_multi_scalar_mul_{'fr' if fr_coefs else 'big'}_coefs_vartime_parallel")
    r = np.zeros(info.jac_bytes, dtype=np.uint8)
    fn(ctypes.c_void_p(tp or 0), _ptr(r), _ptr(coefs), _ptr(points), coefs.shape[0])
    return r


class MsmRefused(RuntimeError):
    """A neutral host-pointer symbol returned an error code: -1 refused (bad id / length / all slots busy), -2 out of device
    memory.  The caller that has a CPU implementation next to it (the Nim binding of INTEGRATION.md part B) falls back to it."""

    def __init__(self, code):
        super().__init__(f"ctt_hip_msm_host refused the call: {code}")
        self.code = code


def msm_available():
    """ctt_hip_msm_available(): 1 when a HIP device is present, never aborts."""
    return bool(_lib.lib().ctt_hip_msm_available())


def msm_host(curve, coefs, points, coord="jac", fr_coefs=False, typed=True):
    """The neutral host-pointer symbols (include/ctt_msm_hip.h part 1c): ctt_hip_msm_<curve>_<coord>_<big|fr> (typed=True,
    "jac" / "prj") or the generic ctt_hip_msm_host (also "aff").  Same arguments and result as multiScalarMul_vartime; raises
    MsmRefused with the symbol's status instead of aborting."""
    info, coefs, points = _check(curve, coefs, points)
    L = _lib.lib()
    nco = 2 if coord == "aff" else 3
    r = np.zeros(nco * info.coord_bytes, dtype=np.uint8)
    if typed and coord != "aff":
        rc = getattr(L, f"ctt_hip_msm_{info.sym}_{coord}_{'fr' if fr_coefs else 'big'}")(_ptr(r), _ptr(coefs), _ptr(points), coefs.shape[0])
    else:
        rc = L.ctt_hip_msm_host(info.cid, COEF_FR if fr_coefs else COEF_BIG, _COORD[coord], _ptr(r), _ptr(coefs), _ptr(points),
                                coefs.shape[0])
    if rc != 0:
        raise MsmRefused(rc)
    return r


class DeviceMsm:
    """Device-resident MSM: inputs already in HBM (torch CUDA tensors or raw device pointers)."""

    def __init__(self, device=0):
        self.L = _lib.lib()
        self.device = device
        self.ctx = self.L.ctt_hip_msm_ctx_create(device)
        if not self.ctx:
            raise RuntimeError("ctt_hip_msm_ctx_create failed")

    def close(self):
        if self.ctx:
            self.L.ctt_hip_msm_ctx_destroy(self.ctx)
            self.ctx = None

    def set_option(self, key, value):
        """ctt_hip_msm_set_option: "c", "K", "S", "chunks", "horner_bits", "host_window_sums", "sort_staged", "sort_xcd", "timings", "timings_every"
        (include/ctt_msm_hip.h); 0 = automatic / off.  KeyError for an unknown key."""
        if self.L.ctt_hip_msm_set_option(self.ctx, key.encode(), int(value)) != 0:
            raise KeyError(key)

    @staticmethod
    def _dptr(t):
        return ctypes.c_void_p(t.data_ptr() if hasattr(t, "data_ptr") else int(t))

    def _order(self, *tensors):
        """Order the engine's stream after torch's current stream when an argument is a torch CUDA tensor: the engine
        runs on its own non-blocking streams, so without this a kernel that is still producing the tensor could be
        overtaken (ctt_hip_msm_wait_stream; INTEGRATION.md "Stream ordering").  Raw integer addresses carry no stream:
        their owner orders them (wait_stream / a synchronize)."""
        for t in tensors:
            if hasattr(t, "data_ptr") and getattr(t, "is_cuda", False):
                import torch
                s = torch.cuda.current_stream(t.device)
                # Nothing in flight on the producer stream: whatever wrote the tensors is done, and there is nothing to order against.
                # (Round 6: the event record + two stream waits per submit are a barrier packet on the producer's queue and ~10 us of host
                # time -- a pipelined caller of small MSMs whose producer is torch's legacy null stream measured 0.67 -> 0.80 ms per MSM at
                # 2^17 pairs with three MSMs in flight because of them; the query is one hipStreamQuery.)
                if s.query():
                    return
                self.wait_stream(s.cuda_stream)
                return

    def wait_stream(self, stream):
        """Everything the engine enqueues from now on waits for the work `stream` (a hipStream_t address) holds now."""
        if self.L.ctt_hip_msm_wait_stream(self.ctx, ctypes.c_void_p(int(stream))) != 0:
            raise RuntimeError("ctt_hip_msm_wait_stream failed")

    def msm(self, curve, d_coefs, d_points, n, coord="aff", fr_coefs=False):
        info = CURVES[curve]
        nco = 2 if coord == "aff" else 3
        r = np.zeros(nco * info.coord_bytes, dtype=np.uint8)
        self._order(d_coefs, d_points)
        rc = self.L.ctt_hip_msm_device(self.ctx, info.cid, COEF_FR if fr_coefs else COEF_BIG, _COORD[coord], _ptr(r),
                                       self._dptr(d_coefs), self._dptr(d_points), n)
        if rc != 0:
            raise RuntimeError("ctt_hip_msm_device failed")
        return r

    def submit(self, curve, d_coefs, d_points, n, fr_coefs=False):
        """Enqueue one MSM and return a ticket at once (at most three outstanding per curve: a fourth raises)."""
        info = CURVES[curve]
        self._order(d_coefs, d_points)
        t = self.L.ctt_hip_msm_device_submit(self.ctx, info.cid, COEF_FR if fr_coefs else COEF_BIG,
                                             self._dptr(d_coefs), self._dptr(d_points), n)
        if t < 0:
            raise RuntimeError("ctt_hip_msm_device_submit failed (bad arguments, or three tickets already outstanding)")
        return (curve, t)

    def finish(self, ticket, coord="aff"):
        """Wait for a submitted MSM, run the host tail, return the result."""
        curve, t = ticket
        info = CURVES[curve]
        nco = 2 if coord == "aff" else 3
        r = np.zeros(nco * info.coord_bytes, dtype=np.uint8)
        if self.L.ctt_hip_msm_device_finish(self.ctx, t, _COORD[coord], _ptr(r)) != 0:
            raise RuntimeError("ctt_hip_msm_device_finish failed (ticket not outstanding)")
        return r

    def sync(self):
        """Wait until everything enqueued on the engine's stream(s) has completed."""
        self.L.ctt_hip_msm_sync(self.ctx)

    def gen_points(self, curve, seed, n, d_out, first=0):
        self._order(d_out)
        rc = self.L.ctt_hip_gen_points(self.ctx, CURVES[curve].cid, seed & (2**64 - 1), first, n, self._dptr(d_out))
        if rc != 0:
            raise RuntimeError("ctt_hip_gen_points failed")

    def field_op(self, curve, op, d_a, d_b, d_r, n):
        self._order(d_a, d_b, d_r)
        rc = self.L.ctt_hip_field_op(self.ctx, CURVES[curve].cid, op, self._dptr(d_a), self._dptr(d_b), self._dptr(d_r), n)
        if rc != 0:
            raise RuntimeError("ctt_hip_field_op failed")

    def sum_reduce(self, curve, d_points, n, coord="aff"):
        """r = sum of n affine points resident in HBM (sum_reduce_vartime)."""
        info = CURVES[curve]
        nco = 2 if coord == "aff" else 3
        r = np.zeros(nco * info.coord_bytes, dtype=np.uint8)
        self._order(d_points)
        if self.L.ctt_hip_sum_reduce(self.ctx, info.cid, _COORD[coord], _ptr(r), self._dptr(d_points), n, 1) != 0:
            raise RuntimeError("ctt_hip_sum_reduce failed")
        return r

    def batch_affine(self, curve, d_dst, d_src, n, src_coord="jac"):
        """d_dst[i] = affine(d_src[i]); both in HBM (batchAffine_vartime).  Returns when d_dst is complete."""
        self._order(d_src, d_dst)
        if self.L.ctt_hip_batch_affine(self.ctx, CURVES[curve].cid, _COORD[src_coord], self._dptr(d_dst),
                                       self._dptr(d_src), n, 1) != 0:
            raise RuntimeError("ctt_hip_batch_affine failed")

    def enable_timings(self, on=True):
        """Record the per-stage HIP events that last_timings() reads (off by default: they are host time per MSM and barrier
        packets in the queue).  on = 2: the accumulate stage and the total only (what a roofline needs)."""
        self.set_option("timings", int(on))

    def last_timings(self):
        ms = np.zeros(6, dtype=np.float32)
        self.L.ctt_hip_msm_last_timings(self.ctx, _ptr(ms), 6)
        return dict(zip(("digits", "sort", "accumulate", "merge", "reduce", "total"), (float(x) for x in ms)))

    def last_plan(self):
        p = np.zeros(6, dtype=np.int32)
        self.L.ctt_hip_msm_last_plan(self.ctx, _ptr(p), 6)
        return dict(zip(("c", "W", "K", "G", "S", "lanes"), (int(x) for x in p)))


def set_devices(devices):
    """GPUs the host-pointer entry points (multiScalarMul_vartime[_parallel], CttEngine.msm) shard a call over, by points
    (ctt_hip_msm_set_devices; the reference's msm-level split, ec_multi_scalar_mul_parallel.nim:386-431).  A device id may
    repeat (one context each); fewer than two ids turns sharding off."""
    arr = (ctypes.c_int * max(1, len(devices)))(*devices)
    if _lib.lib().ctt_hip_msm_set_devices(arr, len(devices)) != 0:
        raise ValueError(f"device id out of range in {list(devices)}")


def set_shard_min(pairs_per_device):
    _lib.lib().ctt_hip_msm_set_shard_min(int(pairs_per_device))


def subgroup_check(curve, points, ctx=None):
    """ok[i] = ([r]points[i] is the neutral element), r = the curve order, for all points in one launch
    (ctt_hip_subgroup_check; the reference checks deserialised points one by one: isInSubgroup)."""
    info = CURVES[curve]
    pts = np.ascontiguousarray(points, dtype=np.uint8).reshape(-1, info.aff_bytes)
    ok = np.zeros(pts.shape[0], dtype=np.uint8)
    if _lib.lib().ctt_hip_subgroup_check(ctx, info.cid, _ptr(ok), _ptr(pts), pts.shape[0], 0) != 0:
        raise RuntimeError("ctt_hip_subgroup_check failed")
    return ok.astype(bool)


def ec_sum_affine(curve, pts_aff, coord="aff"):
    """Host-only sum of affine points (combining per-GPU partial MSMs)."""
    info = CURVES[curve]
    pts = np.ascontiguousarray(pts_aff, dtype=np.uint8).reshape(-1, info.aff_bytes)
    nco = 2 if coord == "aff" else 3
    r = np.zeros(nco * info.coord_bytes, dtype=np.uint8)
    rc = _lib.lib().ctt_hip_ec_sum_affine(info.cid, _COORD[coord], _ptr(r), _ptr(pts), pts.shape[0])
    if rc != 0:
        raise RuntimeError("ctt_hip_ec_sum_affine failed")
    return r


def sum_reduce_vartime(curve, points, coord="jac"):
    """Mirror of sum_reduce_vartime(r, points) (ec_shortweierstrass_batch_ops.nim:649-663), host arrays in, GPU sum."""
    info = CURVES[curve]
    pts = np.ascontiguousarray(points, dtype=np.uint8).reshape(-1, info.aff_bytes)
    nco = 2 if coord == "aff" else 3
    r = np.zeros(nco * info.coord_bytes, dtype=np.uint8)
    if _lib.lib().ctt_hip_sum_reduce(None, info.cid, _COORD[coord], _ptr(r), _ptr(pts), pts.shape[0], 0) != 0:
        raise RuntimeError("ctt_hip_sum_reduce failed")
    return r


def batchAffine_vartime(curve, points, coord="jac"):
    """Mirror of batchAffine_vartime(affs, projs) through the Constantine C symbol `ctt_<curve>_<coord>_batch_affine`
    (bindings/c_curve_decls.nim:395-396): [n][3 coordinates] Montgomery bytes -> [n] affine points."""
    info = CURVES[curve]
    src = np.ascontiguousarray(points, dtype=np.uint8).reshape(-1, 3 * info.coord_bytes)
    dst = np.zeros((src.shape[0], info.aff_bytes), dtype=np.uint8)
    fn = getattr(_lib.lib(), f"ctt_{info.sym}_{coord}_batch_affine")
    fn(_ptr(dst), _ptr(src), src.shape[0])
    return dst


class CachedBases:
    """Base points resident in HBM in the device representation (ctt_hip_msm_bases_*).

    table=True also keeps their multiples 2^(c*w) * P for every digit window (ctt_hip_msm_bases_create_table): (bits/c + 1)
    times the memory, and every later MSM over these bases needs bits/c + 1 accumulations per pair into one bucket set
    with a larger c than a table-less MSM (window_bits = 0: chosen from the number of bases), and no window combine."""

    def __init__(self, curve, points, ctx=None, on_device=False, table=False, window_bits=0):
        self.L = _lib.lib()
        self.curve = curve
        self.info = CURVES[curve]
        self.ctx = ctx
        if on_device:
            import torch
            self.n = int(points.shape[0])
            torch.cuda.current_stream(points.device).synchronize()  # the records are made from the tensor as it is now
            ptr = ctypes.c_void_p(points.data_ptr())
        else:
            points = np.ascontiguousarray(points, dtype=np.uint8)
            if points.ndim != 2 or points.shape[1] != self.info.aff_bytes:
                raise ValueError(f"points must have shape (n, {self.info.aff_bytes})")
            self.n = points.shape[0]
            ptr = _ptr(points)
        if table:
            self.handle = self.L.ctt_hip_msm_bases_create_table(ctx, self.info.cid, ptr, self.n, 1 if on_device else 0,
                                                                int(window_bits))
        else:
            self.handle = self.L.ctt_hip_msm_bases_create(ctx, self.info.cid, ptr, self.n, 1 if on_device else 0)
        if not self.handle:
            raise RuntimeError("ctt_hip_msm_bases_create failed")
        self.window_bits = self.L.ctt_hip_msm_bases_window_bits(self.handle)  # 0 = plain records

    def submit(self, d_coefs, n, fr_coefs=False):
        """Coefficients resident on the device (torch CUDA tensor or raw address): enqueue and return a ticket for
        DeviceMsm.finish / CachedBases.finish (at most three outstanding per curve)."""
        if n > self.n:
            raise AssertionError("more coefficients than cached bases")
        if hasattr(d_coefs, "data_ptr") and getattr(d_coefs, "is_cuda", False):
            import torch
            if self.L.ctt_hip_msm_wait_stream(self.ctx, ctypes.c_void_p(torch.cuda.current_stream(d_coefs.device).cuda_stream)) != 0:
                raise RuntimeError("ctt_hip_msm_wait_stream failed")
        t = self.L.ctt_hip_msm_with_bases_submit(self.ctx, self.handle, COEF_FR if fr_coefs else COEF_BIG,
                                                 DeviceMsm._dptr(d_coefs), n)
        if t < 0:
            raise RuntimeError("ctt_hip_msm_with_bases_submit failed (wrong context, or three tickets already outstanding)")
        return (self.curve, t)

    def finish(self, ticket, coord="prj"):
        curve, t = ticket
        nco = 2 if coord == "aff" else 3
        r = np.zeros(nco * self.info.coord_bytes, dtype=np.uint8)
        if self.L.ctt_hip_msm_device_finish(self.ctx, t, _COORD[coord], _ptr(r)) != 0:
            raise RuntimeError("ctt_hip_msm_device_finish failed (ticket not outstanding)")
        return r

    def msm(self, coefs, coord="prj", fr_coefs=False):
        if hasattr(coefs, "data_ptr") and getattr(coefs, "is_cuda", False):
            return self.finish(self.submit(coefs, int(coefs.shape[0]), fr_coefs=fr_coefs), coord=coord)
        coefs = np.ascontiguousarray(coefs, dtype=np.uint8)
        if coefs.ndim != 2 or coefs.shape[1] != 32:
            raise ValueError("coefs must have shape (n, 32)")
        if coefs.shape[0] > self.n:
            raise AssertionError("more coefficients than cached bases")
        nco = 2 if coord == "aff" else 3
        r = np.zeros(nco * self.info.coord_bytes, dtype=np.uint8)
        rc = self.L.ctt_hip_msm_with_bases(self.ctx, self.handle, COEF_FR if fr_coefs else COEF_BIG, _COORD[coord], _ptr(r),
                                           _ptr(coefs), coefs.shape[0], 0)
        if rc != 0:
            raise RuntimeError("ctt_hip_msm_with_bases failed")
        return r

    def close(self):
        if self.handle:
            self.L.ctt_hip_msm_bases_destroy(self.ctx, self.handle)
            self.handle = None


class CttEngine:
    """Mirror of the Halo2-ZAL engine (constantine-halo2-zal/src/lib.rs:22-96): BN254-Snarks G1,
    coefficients are Fr elements in Montgomery form, result is a projective point."""

    CURVE = "bn254_snarks_g1"

    def __init__(self, num_threads=0):
        self.num_threads = num_threads  # kept for signature parity; the GPU path has no thread pool

    def msm(self, coeffs, bases):
        # lib.rs:42-58 -> ctt_bn254_snarks_g1_prj_multi_scalar_mul_fr_coefs_vartime_parallel
        if len(coeffs) != len(bases):
            raise AssertionError("coeffs and bases must have the same length")  # assert_eq! at lib.rs:43
        return multiScalarMul_vartime_parallel(None, self.CURVE, coeffs, bases, coord="prj", fr_coefs=True)

    # descriptor API (lib.rs:60-95). Upstream these are pass-throughs with a "put device specific preprocessing
    # here" note; the base descriptor here uploads + converts the points once and keeps them in HBM.
    def get_coeffs_descriptor(self, coeffs):
        return np.ascontiguousarray(coeffs, dtype=np.uint8)

    TABLE_BYTES_AUTO = 2 << 30

    def get_base_descriptor(self, bases, table=None):
        # table=True: the window table (CachedBases) -- bits/c + 1 multiples of every base resident in HBM (13-17x the plain
        # records) and c*(bits/c) doublings per base to build, one bucket set per MSM afterwards.  Automatic (None): only while
        # the table stays below TABLE_BYTES_AUTO; the library itself falls back to plain records when a table does not fit.
        if table is None:
            # rows of the table the library would build: one per digit window, bits/c + 1 with c ~ log2(n) (choose_table_window_bits,
            # msm_pipeline.h: 13 rows of 128-byte records at 2^20 bases = 1.74 GB; round 3 estimated 17 rows and dropped the table
            # for exactly that size)
            n = max(1, len(bases))
            c_est = min(22, max(8, int(np.ceil(np.log2(n)))))
            rows = -(-(CURVES[self.CURVE].scalar_bits + 1) // c_est)
            table = n * rows * 128 <= self.TABLE_BYTES_AUTO
        return CachedBases(self.CURVE, bases, table=table)

    def msm_with_cached_scalars(self, coeffs_desc, bases):
        return self.msm(coeffs_desc, bases)

    def msm_with_cached_base(self, coeffs, base_desc):
        if len(coeffs) != base_desc.n:
            raise AssertionError("coeffs and bases must have the same length")
        return base_desc.msm(coeffs, coord="prj", fr_coefs=True)

    def msm_with_cached_inputs(self, coeffs_desc, base_desc):
        return self.msm_with_cached_base(coeffs_desc, base_desc)
