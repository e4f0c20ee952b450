"""ctypes binding of include/bppp.h.  Data crosses as numpy uint64 arrays (4 limbs per scalar,
8 per affine point, little-endian) or as raw device pointers (ints, e.g. torch.Tensor.data_ptr())."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
NUM_STAGES = 6
STAGE_NAMES = ["digits", "sort", "acc_points", "acc_records", "reduce", "finish"]

# every symbol include/bppp.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "bppp_ctx_create", "bppp_ctx_destroy", "bppp_ctx_set_stream", "bppp_last_error", "bppp_version",
    "bppp_msm", "bppp_msm_device", "bppp_msm_batch_device", "bppp_sum_points", "bppp_rational_reduce",
    "bppp_fold_points", "bppp_fold_points_device", "bppp_rational_reduce_eis", "bppp_fold_points_eis_device",
    "bppp_norm_round_sums_device", "bppp_lin_round_sums_device",
    "bppp_norm_round_openings_device", "bppp_lin_round_openings_device",
    "bppp_fold_scalars_device", "bppp_tensor_device", "bppp_batch_inverse_device",
    "bppp_nl_create", "bppp_nl_destroy", "bppp_nl_lengths", "bppp_nl_round_commit", "bppp_nl_round_collapse",
    "bppp_nl_get_witness", "bppp_nl_download", "bppp_nl_verify", "bppp_nl_verify_batch_device", "bppp_nl_prove", "bppp_nl_verify_challenges",
    "bppp_nlb_create", "bppp_nlb_destroy", "bppp_nlb_lengths", "bppp_nlb_round_commit", "bppp_nlb_round_collapse", "bppp_nlb_get_witness",
    "bppp_ip_create", "bppp_ip_destroy", "bppp_ip_lengths", "bppp_ip_round_commit", "bppp_ip_round_collapse", "bppp_ip_get_witness", "bppp_ip_verify", "bppp_ip_verify_batch_device",
    "bppp_lift_x_device", "bppp_device_alloc", "bppp_device_free", "bppp_upload", "bppp_download", "bppp_host_alloc", "bppp_host_free",
    "bppp_profile_enable", "bppp_profile_read",
    "bppp_trrp_create", "bppp_trrp_destroy", "bppp_trrp_public_device",
    "bppp_glv_decompose_device", "bppp_msm_glv_device",
    "bppp_basis_create", "bppp_basis_create_device", "bppp_basis_destroy", "bppp_basis_info", "bppp_msm_basis", "bppp_basis_enable_comb",
    "bppp_rp_create", "bppp_rp_create_binary", "bppp_rp_destroy", "bppp_rp_info", "bppp_rp_set_option", "bppp_rp_shape_of", "bppp_rp_digits", "bppp_hash_to_scalar", "bppp_rp_verify_batch", "bppp_rp_verify_batch_device", "bppp_rp_verify_shard_device", "bppp_rp_prove_batch",
]


class BpppError(RuntimeError):
    pass


def lib_path() -> str:
    return os.path.join(_HERE, "lib", "libbppp_hip.so")


def load_library() -> C.CDLL:
    p = lib_path()
    if not os.path.exists(p):
        raise BpppError(f"{p} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(p)
    u64p, vp, sz, i = C.POINTER(C.c_uint64), C.c_void_p, C.c_size_t, C.c_int
    lib.bppp_ctx_create.argtypes = [i, C.POINTER(vp)]
    lib.bppp_ctx_destroy.argtypes = [vp]
    lib.bppp_ctx_destroy.restype = None
    lib.bppp_ctx_set_stream.argtypes = [vp, vp]
    lib.bppp_last_error.argtypes = [vp]
    lib.bppp_last_error.restype = C.c_char_p
    lib.bppp_version.restype = C.c_char_p
    lib.bppp_msm.argtypes = [vp, vp, vp, sz, vp]
    lib.bppp_msm_device.argtypes = [vp, vp, vp, sz, i, vp]
    lib.bppp_msm_batch_device.argtypes = [vp, vp, vp, sz, sz, i, i, vp]
    lib.bppp_sum_points.argtypes = [vp, vp, sz, vp]
    lib.bppp_glv_decompose_device.argtypes = [vp, vp, sz, vp, vp, vp]
    lib.bppp_msm_glv_device.argtypes = [vp, vp, vp, sz, vp]
    lib.bppp_rational_reduce.argtypes = [vp, vp, C.POINTER(i), vp, C.POINTER(i)]
    lib.bppp_rational_reduce_eis.argtypes = [vp, vp, vp, vp, vp]
    lib.bppp_fold_points_eis_device.argtypes = [vp, vp, vp, vp, vp, vp, sz, vp]
    lib.bppp_fold_points.argtypes = [vp, vp, i, vp, i, vp, sz, vp]
    lib.bppp_fold_points_device.argtypes = [vp, vp, i, vp, i, vp, sz, vp]
    lib.bppp_norm_round_sums_device.argtypes = [vp, vp, sz, vp, vp, vp]
    lib.bppp_lin_round_sums_device.argtypes = [vp, vp, vp, sz, vp, vp]
    lib.bppp_norm_round_openings_device.argtypes = [vp, vp, sz, vp, vp, vp, vp]
    lib.bppp_lin_round_openings_device.argtypes = [vp, vp, sz, vp, vp]
    lib.bppp_fold_scalars_device.argtypes = [vp, vp, vp, vp, sz, vp]
    lib.bppp_batch_inverse_device.argtypes = [vp, vp, sz, i, vp]
    lib.bppp_tensor_device.argtypes = [vp, vp, sz, vp, vp, sz, vp]
    lib.bppp_lift_x_device.argtypes = [vp, vp, sz, vp]
    lib.bppp_trrp_create.argtypes = [vp, i, i, sz, sz, sz, vp, vp, vp, vp, vp, vp, vp, sz, vp, vp, vp, sz, vp, vp, vp, C.POINTER(vp)]
    lib.bppp_trrp_destroy.argtypes = [vp]
    lib.bppp_trrp_destroy.restype = None
    lib.bppp_trrp_public_device.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp]
    lib.bppp_nl_create.argtypes = [vp, vp, vp, vp, vp, vp, sz, vp, vp, vp, sz, C.POINTER(vp)]
    lib.bppp_nl_destroy.argtypes = [vp]
    lib.bppp_nl_destroy.restype = None
    lib.bppp_nl_lengths.argtypes = [vp, C.POINTER(sz), C.POINTER(sz)]
    lib.bppp_nl_round_commit.argtypes = [vp, vp, vp, vp, vp]
    lib.bppp_nl_round_collapse.argtypes = [vp, vp]
    lib.bppp_nl_get_witness.argtypes = [vp, vp, vp]
    lib.bppp_nl_download.argtypes = [vp] + [vp] * 9
    lib.bppp_nl_prove.argtypes = [vp, sz, vp, vp, vp, C.POINTER(sz), sz, vp, vp]
    lib.bppp_nl_verify_challenges.argtypes = [vp, vp, vp, sz, vp, C.POINTER(sz), sz, vp]
    lib.bppp_nlb_create.argtypes = [vp, sz, vp, vp, vp, vp, vp, sz, vp, vp, vp, sz, C.POINTER(vp)]
    lib.bppp_nlb_destroy.argtypes = [vp]
    lib.bppp_nlb_destroy.restype = None
    lib.bppp_nlb_lengths.argtypes = [vp, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]
    lib.bppp_nlb_round_commit.argtypes = [vp, vp, vp, vp, vp]
    lib.bppp_nlb_round_collapse.argtypes = [vp, vp]
    lib.bppp_nlb_get_witness.argtypes = [vp, vp, vp, vp]
    lib.bppp_ip_create.argtypes = [vp, vp, vp, vp, vp, vp, sz, vp, vp, vp, sz, C.POINTER(vp)]
    lib.bppp_ip_destroy.argtypes = [vp]
    lib.bppp_ip_destroy.restype = None
    lib.bppp_ip_lengths.argtypes = [vp, C.POINTER(sz), C.POINTER(sz)]
    lib.bppp_ip_round_commit.argtypes = [vp, vp, vp, vp, vp]
    lib.bppp_ip_round_collapse.argtypes = [vp, vp]
    lib.bppp_ip_get_witness.argtypes = [vp, vp, vp, vp]
    lib.bppp_ip_verify.argtypes = [vp, vp, vp, vp, vp, vp, sz, vp, vp, vp, sz, vp, sz, vp, sz, vp, sz, vp, vp, sz, vp, vp]
    lib.bppp_nl_verify_batch_device.argtypes = [vp] + [sz] * 7 + [vp] * 16
    lib.bppp_ip_verify_batch_device.argtypes = [vp] + [sz] * 7 + [vp] * 16
    lib.bppp_nl_verify.argtypes = [vp, vp, vp, vp, vp, vp, sz, vp, vp, vp, sz, vp, sz, vp, sz, vp, sz, vp, vp, sz, vp, vp]
    lib.bppp_device_alloc.argtypes = [vp, sz, C.POINTER(vp)]
    lib.bppp_device_free.argtypes = [vp, vp]
    lib.bppp_host_alloc.argtypes = [vp, sz, C.POINTER(vp)]
    lib.bppp_host_free.argtypes = [vp, vp]
    lib.bppp_upload.argtypes = [vp, vp, vp, sz]
    lib.bppp_download.argtypes = [vp, vp, vp, sz]
    lib.bppp_basis_create.argtypes = [vp, vp, sz, i, sz, C.POINTER(vp)]
    lib.bppp_basis_create_device.argtypes = [vp, vp, sz, i, sz, C.POINTER(vp)]
    lib.bppp_basis_destroy.argtypes = [vp]
    lib.bppp_basis_destroy.restype = None
    lib.bppp_basis_info.argtypes = [vp, C.POINTER(sz), C.POINTER(i), C.POINTER(sz)]
    lib.bppp_msm_basis.argtypes = [vp, vp, sz, sz, vp]
    lib.bppp_basis_enable_comb.argtypes = [vp, C.c_int, sz, vp, vp]
    lib.bppp_rp_create.argtypes = [vp, i, i, vp, sz, vp, sz, vp, sz, C.c_char_p, C.POINTER(vp)]
    lib.bppp_rp_create_binary.argtypes = [vp, i, i, vp, sz, vp, vp, sz, C.c_char_p, C.POINTER(vp)]
    lib.bppp_rp_destroy.argtypes = [vp]
    lib.bppp_rp_destroy.restype = None
    lib.bppp_rp_info.argtypes = [vp, vp]
    lib.bppp_rp_set_option.argtypes = [vp, i, C.c_uint64]
    lib.bppp_rp_shape_of.argtypes = [i, i, vp, sz, vp]
    lib.bppp_rp_digits.argtypes = [vp, vp, vp, sz, C.POINTER(sz), C.POINTER(i)]
    lib.bppp_hash_to_scalar.argtypes = [vp, sz, vp]
    lib.bppp_rp_verify_batch.argtypes = [vp, sz, vp, vp, vp, C.POINTER(i), vp, vp, vp]
    lib.bppp_rp_verify_batch_device.argtypes = [vp, sz, vp, vp, vp, C.POINTER(i), vp, vp, vp]
    lib.bppp_rp_verify_shard_device.argtypes = [vp, sz, C.c_uint64, vp, vp, vp, C.POINTER(i), vp, vp, vp]
    lib.bppp_rp_prove_batch.argtypes = [vp, sz, vp, vp, vp, vp, sz, vp, vp]
    lib.bppp_profile_enable.argtypes = [vp, i]
    lib.bppp_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), i]
    return lib


def load_test_library() -> C.CDLL:
    """libbppp_hip_test.so: the bppp_test_* hooks of include/bppp_test.h (parity tests, bench.py's multiply-rate probe).
    They take the product library's context handle; nothing in the product depends on them."""
    p = os.path.join(_HERE, "lib", "libbppp_hip_test.so")
    if not os.path.exists(p):
        raise BpppError(f"{p} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(p)
    vp, sz, i = C.c_void_p, C.c_size_t, C.c_int
    lib.bppp_test_fe_op.argtypes = [vp, i, i, vp, vp, sz, vp]
    lib.bppp_test_point_op.argtypes = [vp, i, vp, vp, sz, vp]
    lib.bppp_test_mulmod_rate.argtypes = [vp, i, C.POINTER(C.c_double)]
    return lib


class RpRange(C.Structure):
    """bppp_rp_range (include/bppp.h)"""
    _fields_ = [("base", C.c_uint32), ("flags", C.c_uint32), ("min", C.c_uint64 * 4), ("max", C.c_uint64 * 4)]


class RpPublic(C.Structure):
    """bppp_rp_public"""
    _fields_ = [("is_output", C.c_uint32), ("reserved", C.c_uint32), ("type", C.c_uint64 * 4), ("amount", C.c_uint64 * 4)]


class RpShape(C.Structure):
    """bppp_rp_shape"""
    _fields_ = [(n, C.c_size_t) for n in ("nranges", "norm_len", "lin_len", "rounds", "final_norm", "final_lin", "coms_bytes", "proof_bytes",
                                          "challenges_per_proof")]


RP_OPTIONS = {"comb_min": 1, "comb_budget": 2, "comb_bits": 3, "split_min": 4, "host_oracle_max": 5, "fold_points": 6, "host_algebra": 7, "timing": 8}
RP_SHARED, RP_OUTPUT, RP_ASSUMED = 1, 2, 4
RP_VALID, RP_INVALID, RP_MALFORMED = 0, 1, 2


# ---- integer <-> limb helpers (host-side glue for tests / bench)
def int_to_limbs(x: int, n: int = 4) -> np.ndarray:
    return np.array([(x >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(n)], dtype=np.uint64)


def limbs_to_int(a) -> int:
    return sum(int(v) << (64 * k) for k, v in enumerate(np.asarray(a).reshape(-1)))


def scalars_to_array(xs: Sequence[int]) -> np.ndarray:
    out = np.zeros((len(xs), 4), dtype=np.uint64)
    for r, x in enumerate(xs):
        out[r] = int_to_limbs(x)
    return out


def points_to_array(ps) -> np.ndarray:
    out = np.zeros((len(ps), 8), dtype=np.uint64)
    for r, p in enumerate(ps):
        if p is not None:
            out[r, :4] = int_to_limbs(p[0])
            out[r, 4:] = int_to_limbs(p[1])
    return out


def array_to_point(a):
    a = np.asarray(a).reshape(8)
    x, y = limbs_to_int(a[:4]), limbs_to_int(a[4:])
    return None if x == 0 and y == 0 else (x, y)


def array_to_scalars(a):
    a = np.asarray(a).reshape(-1, 4)
    return [limbs_to_int(r) for r in a]


class _Ptr(C.c_void_p):
    """c_void_p that keeps the numpy array it points into alive until the foreign call returns."""
    _keep = None


def _ptr(a) -> C.c_void_p:
    if a is None:
        return C.c_void_p(0)
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"] and a.dtype in (np.uint64, np.uint32)
        p = _Ptr(a.ctypes.data)
        p._keep = a
        return p
    return C.c_void_p(int(a))  # raw device pointer


class Bppp:
    """One context = one GPU + one HIP stream (include/bppp.h)."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.bppp_ctx_create(device, C.byref(h))
        if rc != 0:
            raise BpppError(f"bppp_ctx_create(device={device}) failed with {rc} "
                            "(-3 = no GPU visible; this library has no CPU path)")
        self.h = h
        import weakref
        self._children = weakref.WeakSet()      # child handles (NormLinearBP, ...): closed before the context
        if stream is not None:
            self.set_stream(stream)

    def _adopt(self, child):
        self._children.add(child)

    def close(self):
        """Closes the child handles first, then the context.  (The C ABI is safe in either order — a child keeps its context
        alive until it is destroyed itself — this just releases the device memory promptly.)"""
        if getattr(self, "h", None):
            for ch in list(getattr(self, "_children", ())):
                try:
                    ch.close()
                except Exception:
                    pass
            self.lib.bppp_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise BpppError(f"{what} failed ({rc}): {self.lib.bppp_last_error(self.h).decode()}")

    def set_stream(self, stream: int):
        self._check(self.lib.bppp_ctx_set_stream(self.h, C.c_void_p(stream)), "bppp_ctx_set_stream")

    # ---- MSM
    def msm(self, scalars: np.ndarray, points: np.ndarray):
        n = scalars.shape[0] if scalars.size else 0
        out = np.zeros(8, dtype=np.uint64)
        self._check(self.lib.bppp_msm(self.h, _ptr(np.ascontiguousarray(scalars)), _ptr(np.ascontiguousarray(points)), n, _ptr(out)), "bppp_msm")
        return array_to_point(out)

    def msm_device(self, d_scalars: int, d_points: int, n: int, window_bits: int = 0):
        out = np.zeros(8, dtype=np.uint64)
        self._check(self.lib.bppp_msm_device(self.h, _ptr(d_scalars), _ptr(d_points), n, window_bits, _ptr(out)), "bppp_msm_device")
        return array_to_point(out)

    def msm_batch_device(self, d_scalars: int, d_points: int, n: int, batch: int, shared_points: bool, window_bits: int = 0):
        out = np.zeros((batch, 8), dtype=np.uint64)
        self._check(self.lib.bppp_msm_batch_device(self.h, _ptr(d_scalars), _ptr(d_points), n, batch, int(shared_points), window_bits, _ptr(out)),
                    "bppp_msm_batch_device")
        return [array_to_point(out[b]) for b in range(batch)]

    def basis(self, points, window_bits: int = 0, batch_hint: int = 1, device: bool = False, n: int = 0) -> "Basis":
        """register a basis: `points` is an (n, 8) uint64 array, or a device pointer with device=True and n given"""
        return Basis(self, points, window_bits, batch_hint, device, n)

    def msm_glv_device(self, d_scalars: int, d_points: int, n: int):
        """the MSM through the reference's endomorphism decomposition (same group element as msm_device)"""
        out = np.zeros(8, dtype=np.uint64)
        self._check(self.lib.bppp_msm_glv_device(self.h, _ptr(d_scalars), _ptr(d_points), n, _ptr(out)), "bppp_msm_glv_device")
        return array_to_point(out)

    def glv_decompose(self, scalars):
        """decomposeFastPrimeEis for a list of scalars: [(a, b)] as signed Python integers"""
        n = len(scalars)
        d_s = self.to_device(scalars_to_array([s for s in scalars]))
        d_a, d_b, d_g = self.alloc(max(n, 1) * 32), self.alloc(max(n, 1) * 32), self.alloc(max(n, 1) * 4 + 16)
        try:
            self._check(self.lib.bppp_glv_decompose_device(self.h, _ptr(d_s), n, _ptr(d_a), _ptr(d_b), _ptr(d_g)), "bppp_glv_decompose_device")
            a, b = array_to_scalars(self.download(d_a, (n, 4))), array_to_scalars(self.download(d_b, (n, 4)))
            g = self.download(d_g, (n,), dtype=np.uint32)
        finally:
            for p_ in (d_s, d_a, d_b, d_g):
                self.free(p_)
        assert not any(int(x) & 4 for x in g)
        return [(-x if int(f) & 1 else x, -y if int(f) & 2 else y) for x, y, f in zip(a, b, g)]

    def sum_points(self, points: np.ndarray):
        """sum of a few affine points ((n, 8) uint64): the combine step after the all-gather of a sharded MSM"""
        pts = np.ascontiguousarray(points)
        out = np.zeros(8, dtype=np.uint64)
        self._check(self.lib.bppp_sum_points(self.h, _ptr(pts), pts.shape[0] if pts.size else 0, _ptr(out)), "bppp_sum_points")
        return array_to_point(out)

    # ---- reduced scalars / folds
    def rational_reduce(self, x: int) -> Tuple[int, int]:
        am, bm = np.zeros(3, dtype=np.uint64), np.zeros(3, dtype=np.uint64)
        an, bn = C.c_int(0), C.c_int(0)
        rc = self.lib.bppp_rational_reduce(_ptr(int_to_limbs(x)), _ptr(am), C.byref(an), _ptr(bm), C.byref(bn))
        if rc != 0:
            raise BpppError(f"bppp_rational_reduce failed ({rc})")
        a, b = limbs_to_int(am), limbs_to_int(bm)
        return (-a if an.value else a, -b if bn.value else b)

    def rational_reduce_eis(self, x: int):
        """((a0, a1), (b0, b1)): the Eisenstein rational reduction x = (a0 + a1 w) / (b0 + b1 w) of the FastPrime configuration"""
        am, bm = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
        an, bn = np.zeros(2, dtype=np.int32), np.zeros(2, dtype=np.int32)
        rc = self.lib.bppp_rational_reduce_eis(_ptr(int_to_limbs(x)), _ptr(am), C.c_void_p(an.ctypes.data), _ptr(bm), C.c_void_p(bn.ctypes.data))
        if rc != 0:
            raise BpppError(f"bppp_rational_reduce_eis failed ({rc})")
        comp = lambda m, s, k: (-1 if s[k] else 1) * (int(m[2 * k]) | (int(m[2 * k + 1]) << 64))
        return (comp(am, an, 0), comp(am, an, 1)), (comp(bm, bn, 0), comp(bm, bn, 1))

    def fold_points_eis(self, b, a, points: np.ndarray) -> np.ndarray:
        """out[j] = b * pts[2j] + a * pts[2j+1] for Eisenstein reduced scalars b = (b0, b1), a = (a0, a1)"""
        n = points.shape[0]
        mags = lambda e: np.array([abs(e[0]) & (2**64 - 1), abs(e[0]) >> 64, abs(e[1]) & (2**64 - 1), abs(e[1]) >> 64], dtype=np.uint64)
        negs = lambda e: np.array([int(e[0] < 0), int(e[1] < 0)], dtype=np.int32)
        d_p = self.to_device(np.ascontiguousarray(points))
        d_o = self.alloc(((n + 1) // 2) * 64)
        bn, an = negs(b), negs(a)
        try:
            self._check(self.lib.bppp_fold_points_eis_device(self.h, _ptr(mags(b)), C.c_void_p(bn.ctypes.data), _ptr(mags(a)), C.c_void_p(an.ctypes.data), _ptr(d_p), n,
                                                             _ptr(d_o)), "bppp_fold_points_eis_device")
            return self.download(d_o, ((n + 1) // 2, 8))
        finally:
            self.free(d_p); self.free(d_o)

    def fold_points(self, b: int, a: int, points: np.ndarray) -> np.ndarray:
        n = points.shape[0]
        out = np.zeros(((n + 1) // 2, 8), dtype=np.uint64)
        self._check(self.lib.bppp_fold_points(self.h, _ptr(int_to_limbs(abs(b), 3)), int(b < 0), _ptr(int_to_limbs(abs(a), 3)), int(a < 0),
                                              _ptr(np.ascontiguousarray(points)), n, _ptr(out)), "bppp_fold_points")
        return out

    def fold_points_device(self, b: int, a: int, d_points: int, n: int, d_out: int):
        self._check(self.lib.bppp_fold_points_device(self.h, _ptr(int_to_limbs(abs(b), 3)), int(b < 0), _ptr(int_to_limbs(abs(a), 3)), int(a < 0),
                                                     _ptr(d_points), n, _ptr(d_out)), "bppp_fold_points_device")

    # ---- round scalar kernels
    def norm_round_sums(self, d_x: int, n: int, q4: int) -> Tuple[int, int]:
        sx, sr = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
        self._check(self.lib.bppp_norm_round_sums_device(self.h, _ptr(d_x), n, _ptr(int_to_limbs(q4)), _ptr(sx), _ptr(sr)), "bppp_norm_round_sums_device")
        return limbs_to_int(sx), limbs_to_int(sr)

    def lin_round_sums(self, d_c: int, d_x: int, n: int) -> Tuple[int, int]:
        sx, sr = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
        self._check(self.lib.bppp_lin_round_sums_device(self.h, _ptr(d_c), _ptr(d_x), n, _ptr(sx), _ptr(sr)), "bppp_lin_round_sums_device")
        return limbs_to_int(sx), limbs_to_int(sr)

    def norm_round_openings(self, d_x: int, n: int, q: int, qinv: int, d_xw: int, d_rw: int):
        self._check(self.lib.bppp_norm_round_openings_device(self.h, _ptr(d_x), n, _ptr(int_to_limbs(q)), _ptr(int_to_limbs(qinv)), _ptr(d_xw), _ptr(d_rw)),
                    "bppp_norm_round_openings_device")

    def lin_round_openings(self, d_x: int, n: int, d_xw: int, d_rw: int):
        self._check(self.lib.bppp_lin_round_openings_device(self.h, _ptr(d_x), n, _ptr(d_xw), _ptr(d_rw)), "bppp_lin_round_openings_device")

    def fold_scalars(self, u: int, v: int, d_x: int, n: int, d_out: int):
        self._check(self.lib.bppp_fold_scalars_device(self.h, _ptr(int_to_limbs(u)), _ptr(int_to_limbs(v)), _ptr(d_x), n, _ptr(d_out)), "bppp_fold_scalars_device")

    def tensor(self, bs: Sequence[int], es: Sequence[int], qs: Sequence[int], d_out: int):
        k = len(es)
        assert len(qs) == k
        self._check(self.lib.bppp_tensor_device(self.h, _ptr(scalars_to_array(bs)), len(bs), _ptr(scalars_to_array(es)) if k else None,
                                                _ptr(scalars_to_array(qs)) if k else None, k, _ptr(d_out)), "bppp_tensor_device")

    def batch_inverse(self, d_x: int, n: int, modulus: int, d_out: int):
        self._check(self.lib.bppp_batch_inverse_device(self.h, _ptr(d_x), n, modulus, _ptr(d_out)), "bppp_batch_inverse_device")

    def lift_x(self, d_x: int, n: int, d_points: int):
        self._check(self.lib.bppp_lift_x_device(self.h, _ptr(d_x), n, _ptr(d_points)), "bppp_lift_x_device")

    # ---- device memory
    def alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._check(self.lib.bppp_device_alloc(self.h, nbytes, C.byref(p)), "bppp_device_alloc")
        return int(p.value)

    def free(self, d_ptr: int):
        self._check(self.lib.bppp_device_free(self.h, _ptr(d_ptr)), "bppp_device_free")

    def host_alloc(self, nbytes: int) -> np.ndarray:
        """page-locked host memory as a uint8 array (bppp_host_alloc): copies from it are DMA transfers; release with host_free"""
        p = C.c_void_p()
        self._check(self.lib.bppp_host_alloc(self.h, nbytes, C.byref(p)), "bppp_host_alloc")
        buf = (C.c_uint8 * max(nbytes, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=np.uint8, count=nbytes)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def host_free(self, arr: np.ndarray):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p is None:
            raise ValueError("not an array from host_alloc")
        self._check(self.lib.bppp_host_free(self.h, _ptr(p)), "bppp_host_free")

    def upload(self, d_dst: int, src: np.ndarray):
        src = np.ascontiguousarray(src)
        self._check(self.lib.bppp_upload(self.h, _ptr(d_dst), C.c_void_p(src.ctypes.data), src.nbytes), "bppp_upload")

    def download(self, d_src: int, shape, dtype=np.uint64) -> np.ndarray:
        out = np.zeros(shape, dtype=dtype)
        self._check(self.lib.bppp_download(self.h, C.c_void_p(out.ctypes.data), _ptr(d_src), out.nbytes), "bppp_download")
        return out

    def to_device(self, arr: np.ndarray) -> int:
        arr = np.ascontiguousarray(arr)
        p = self.alloc(max(arr.nbytes, 16))
        if arr.nbytes:
            self.upload(p, arr)
        return p

    # ---- profiling
    def profile_enable(self, on: bool = True):
        self._check(self.lib.bppp_profile_enable(self.h, int(on)), "bppp_profile_enable")

    def profile_read(self, reset: bool = True):
        ms = (C.c_double * NUM_STAGES)()
        calls = C.c_uint64(0)
        self._check(self.lib.bppp_profile_read(self.h, ms, C.byref(calls), int(reset)), "bppp_profile_read")
        return {STAGE_NAMES[k]: ms[k] for k in range(NUM_STAGES)}, int(calls.value)


class Basis:
    """bppp_basis: points registered once with their fixed-base table; msm() runs over the first n_terms of them."""

    def __init__(self, gpu: Bppp, points, window_bits: int = 0, batch_hint: int = 1, device: bool = False, n: int = 0):
        self.gpu, self.h = gpu, None
        h = C.c_void_p()
        if device:
            rc = gpu.lib.bppp_basis_create_device(gpu.h, _ptr(points), n, window_bits, batch_hint, C.byref(h))
        else:
            pts = np.ascontiguousarray(points)
            rc = gpu.lib.bppp_basis_create(gpu.h, _ptr(pts), pts.shape[0], window_bits, batch_hint, C.byref(h))
        gpu._check(rc, "bppp_basis_create")
        self.h = h
        gpu._adopt(self)
        n_, c_, tb = C.c_size_t(0), C.c_int(0), C.c_size_t(0)
        gpu._check(gpu.lib.bppp_basis_info(h, C.byref(n_), C.byref(c_), C.byref(tb)), "bppp_basis_info")
        self.n, self.window_bits, self.table_bytes = int(n_.value), int(c_.value), int(tb.value)

    def close(self):
        if self.h:
            self.gpu.lib.bppp_basis_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def enable_comb(self, window_bits: int = 0, budget_bytes: int = 32 << 30):
        """bppp_basis_enable_comb: keep every multiple of every window; returns (window_bits, table_bytes)"""
        c_, tb = C.c_int(0), C.c_size_t(0)
        self.gpu._check(self.gpu.lib.bppp_basis_enable_comb(self.h, window_bits, budget_bytes, C.byref(c_), C.byref(tb)), "bppp_basis_enable_comb")
        return int(c_.value), int(tb.value)

    def msm(self, d_scalars: int, n_terms: int, batch: int = 1):
        out = np.zeros((batch, 8), dtype=np.uint64)
        self.gpu._check(self.gpu.lib.bppp_msm_basis(self.h, _ptr(d_scalars), n_terms, batch, _ptr(out)), "bppp_msm_basis")
        return [array_to_point(out[b]) for b in range(batch)]
