"""ctypes loader for libspartan_hip.so (include/spartan_hip.h). Harness glue only: tests and bench.py use it
to call the C ABI exactly as a Rust `extern "C"` block would. There is no fallback: if the
library is missing or no gfx950 device is present, calls raise.
"""
import ctypes
import os
import re
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_PKG)
LIB_DIR = os.path.join(_PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libspartan_hip.so")
HEADER = os.path.join(ROOT, "include", "spartan_hip.h")

c_u64p = ctypes.POINTER(ctypes.c_uint64)
c_u8p = ctypes.POINTER(ctypes.c_uint8)
SIZE_MAX = (1 << 64) - 1


def build(jobs: int = 8):
    env = dict(os.environ)
    subprocess.check_call(["make", "-s", "-j", str(jobs), "-C", os.path.join(_PKG, "csrc")], env=env)


def declared_symbols():
    """Every function name declared in include/spartan_hip.h."""
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sp_[a-z0-9_]+)\s*\(", txt)))


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback exists)")
        # One HIP runtime per process. torch ships its own libamdhip64 / libhsa-runtime64 (same SONAMEs as /opt/rocm's); whichever copy is mapped first
        # serves both. Mapped in the other order - this library (-> /opt/rocm) first, torch second - the process ends up with two ROCr instances and the
        # one initialised second sees no device ("no ROCm-capable device is detected"), so torch goes first wherever it is installed.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        L.sp_last_error.restype = ctypes.c_char_p
        _lib = L
    return _lib


class SpartanHipError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise SpartanHipError(f"rc={rc}: {lib().sp_last_error().decode()}")


def p64(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(c_u64p)


def p8(a):
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_u8p)


def _bytes(b: bytes):
    arr = np.frombuffer(b, dtype=np.uint8).copy() if len(b) else np.zeros(1, dtype=np.uint8)
    return arr, ctypes.c_size_t(len(b))


class Context:
    def __init__(self, device: int = 0):
        self.h = ctypes.c_void_p()
        check(lib().sp_ctx_create(int(device), ctypes.byref(self.h)))

    def close(self):
        if self.h:
            lib().sp_ctx_destroy(self.h)
            self.h = ctypes.c_void_p()

    def synchronize(self):
        check(lib().sp_ctx_synchronize(self.h))

    def reset_stats(self, enable=True):
        check(lib().sp_ctx_reset_stats(self.h, int(enable)))

    def stats_filter(self, only: str = ""):
        check(lib().sp_ctx_stats_filter(self.h, only.encode()))

    def mail_stats(self):
        """(answers taken from the host-memory mirror, watchdog trips, wanted seq, device-line seq at the last mirror answer, ring in device memory)"""
        out = (ctypes.c_uint64 * 5)()
        check(lib().sp_ctx_mail_stats(self.h, out))
        return tuple(int(v) for v in out)

    def kernel_stats(self, what: str):
        ms = ctypes.c_double()
        n = ctypes.c_uint64()
        b = ctypes.c_uint64()
        check(lib().sp_ctx_kernel_stats(self.h, what.encode(), ctypes.byref(ms), ctypes.byref(n), ctypes.byref(b)))
        return ms.value, n.value, b.value

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Table:
    """MultilinearPolynomial resident in HBM (src/polys/multilinear.rs:34-44)."""

    def __init__(self, ctx: Context, handle):
        self.ctx = ctx
        self.h = handle

    @classmethod
    def from_host(cls, ctx, z: np.ndarray, lo_eff=SIZE_MAX, hi_eff=SIZE_MAX):
        z = np.ascontiguousarray(z, dtype=np.uint64).reshape(-1, 4)
        h = ctypes.c_void_p()
        check(lib().sp_table_from_host(ctx.h, p64(z), ctypes.c_size_t(z.shape[0]), ctypes.c_size_t(lo_eff), ctypes.c_size_t(hi_eff), ctypes.byref(h)))
        return cls(ctx, h)

    @classmethod
    def zeros(cls, ctx, n, lo_eff=SIZE_MAX, hi_eff=SIZE_MAX):
        h = ctypes.c_void_p()
        check(lib().sp_table_zeros(ctx.h, ctypes.c_size_t(n), ctypes.c_size_t(lo_eff), ctypes.c_size_t(hi_eff), ctypes.byref(h)))
        return cls(ctx, h)

    @classmethod
    def eq(cls, ctx, r: np.ndarray):
        r = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)
        h = ctypes.c_void_p()
        check(lib().sp_eq_table(ctx.h, p64(r) if r.shape[0] else None, ctypes.c_size_t(r.shape[0]), ctypes.byref(h)))
        return cls(ctx, h)

    @staticmethod
    def eq_begin(ctx, r_known: np.ndarray, ell: int):
        """sp_eq_table_begin: the first ell - 2 coordinates of a point still being drawn."""
        r_known = np.ascontiguousarray(r_known, dtype=np.uint64).reshape(-1, 4)
        check(lib().sp_eq_table_begin(ctx.h, p64(r_known), ctypes.c_size_t(r_known.shape[0]), ctypes.c_size_t(ell)))

    def eq_finish(self, r: np.ndarray):
        """sp_eq_table_finish into this table (capacity >= 2^ell)."""
        r = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)
        check(lib().sp_eq_table_finish(self.ctx.h, p64(r), ctypes.c_size_t(r.shape[0]), self.h))

    def info(self):
        n, lo, hi = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
        check(lib().sp_table_info(self.h, ctypes.byref(n), ctypes.byref(lo), ctypes.byref(hi)))
        return n.value, lo.value, hi.value

    def __len__(self):
        return self.info()[0]

    def read(self, off=0, cnt=None):
        if cnt is None:
            cnt = len(self) - off
        out = np.zeros((cnt, 4), dtype=np.uint64)
        check(lib().sp_table_read(self.ctx.h, self.h, ctypes.c_size_t(off), ctypes.c_size_t(cnt), p64(out)))
        return out

    def write(self, off, z):
        z = np.ascontiguousarray(z, dtype=np.uint64).reshape(-1, 4)
        check(lib().sp_table_write(self.ctx.h, self.h, ctypes.c_size_t(off), p64(z), ctypes.c_size_t(z.shape[0])))

    def write_u64(self, off, vals):
        """sp_table_write_u64: machine words in, Montgomery-form elements formed on the device (the is_small witness path)"""
        vals = np.ascontiguousarray(vals, dtype=np.uint64).reshape(-1)
        check(lib().sp_table_write_u64(self.ctx.h, self.h, ctypes.c_size_t(off), p64(vals) if len(vals) else None, ctypes.c_size_t(len(vals))))

    def write_bits(self, off, bits, cnt):
        """sp_table_write_bits: cnt 0/1 values packed 8 a byte (value i = bit i & 7 of byte i >> 3)"""
        bits = np.ascontiguousarray(bits, dtype=np.uint8).reshape(-1)
        assert len(bits) * 8 >= cnt
        check(lib().sp_table_write_bits(self.ctx.h, self.h, ctypes.c_size_t(off), bits.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)) if len(bits) else None, ctypes.c_size_t(cnt)))

    def set_len(self, n, lo_eff=SIZE_MAX, hi_eff=SIZE_MAX):
        """logical length + MultilinearPolynomial::new_with_halves bounds (src/polys/multilinear.rs:62-76)"""
        check(lib().sp_table_set_len(self.h, ctypes.c_size_t(n), ctypes.c_size_t(lo_eff), ctypes.c_size_t(hi_eff)))

    def copy_from(self, dst_off, src: "Table", src_off, cnt):
        check(lib().sp_table_copy(self.ctx.h, self.h, ctypes.c_size_t(dst_off), src.h, ctypes.c_size_t(src_off), ctypes.c_size_t(cnt)))

    def bind_top(self, r):
        r = np.ascontiguousarray(r, dtype=np.uint64).reshape(4)
        check(lib().sp_table_bind_top(self.ctx.h, self.h, p64(r)))

    def free(self):
        if self.h:
            lib().sp_table_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Transcript:
    """Keccak256Transcript (src/provider/keccak.rs:26-105)."""

    def __init__(self, ctx, label: bytes):
        self.h = ctypes.c_void_p()
        a, n = _bytes(label)
        check(lib().sp_transcript_new(ctx.h if ctx else None, p8(a), n, ctypes.byref(self.h)))

    def absorb(self, label: bytes, data: bytes):
        la, ln = _bytes(label)
        da, dn = _bytes(data)
        check(lib().sp_transcript_absorb(self.h, p8(la), ln, p8(da), dn))

    def dom_sep(self, data: bytes):
        da, dn = _bytes(data)
        check(lib().sp_transcript_dom_sep(self.h, p8(da), dn))

    def absorb_prepared(self, label: bytes, data: bytes):
        """absorb() through sp_transcript_preabsorb + sp_transcript_absorb_prepared (valid right after new / squeeze only)."""
        la, ln = _bytes(label)
        da, dn = _bytes(data)
        st = ctypes.c_void_p()
        check(lib().sp_transcript_preabsorb(p8(la), ln, p8(da), dn, ctypes.byref(st)))
        try:
            check(lib().sp_transcript_absorb_prepared(self.h, st))
        finally:
            lib().sp_absorb_state_free(st)

    def squeeze(self, label: bytes):
        la, ln = _bytes(label)
        out = np.zeros(4, dtype=np.uint64)
        check(lib().sp_transcript_squeeze(self.h, p8(la), ln, p64(out)))
        return out

    def set_async(self, on=True):
        check(lib().sp_transcript_set_async(self.h, int(on)))

    def clone(self):
        t = Transcript.__new__(Transcript)
        t.h = ctypes.c_void_p()
        check(lib().sp_transcript_clone(self.h, ctypes.byref(t.h)))
        return t

    def __del__(self):
        try:
            if self.h:
                lib().sp_transcript_free(self.h)
        except Exception:
            pass


def sumcheck_cubic3(ctx, claim, taus, A: Table, B: Table, C: Table, tr: Transcript):
    """SumcheckProof::prove_cubic_with_three_inputs (src/sumcheck.rs:502-571)."""
    taus = np.ascontiguousarray(taus, dtype=np.uint64).reshape(-1, 4)
    ell = taus.shape[0]
    claim = np.ascontiguousarray(claim, dtype=np.uint64).reshape(4)
    polys = np.zeros((ell, 3, 4), dtype=np.uint64)
    r = np.zeros((ell, 4), dtype=np.uint64)
    fin = np.zeros((3, 4), dtype=np.uint64)
    check(lib().sp_sumcheck_cubic3(ctx.h, p64(claim), p64(taus), ctypes.c_size_t(ell), A.h, B.h, C.h, tr.h, p64(polys), p64(r), p64(fin)))
    return polys, r, fin


def sumcheck_cubic3_round0(ctx, claim, taus, A: Table, B: Table, C: Table, p0: Table, p1: Table, tr: Transcript):
    """sp_sumcheck_cubic3_round0: the cubic prover with round 1's per-pair products supplied (Shape.multiply_vec_incremental_round0)."""
    taus = np.ascontiguousarray(taus, dtype=np.uint64).reshape(-1, 4)
    ell = taus.shape[0]
    polys = np.zeros((ell, 3, 4), dtype=np.uint64)
    r = np.zeros((ell, 4), dtype=np.uint64)
    fin = np.zeros((3, 4), dtype=np.uint64)
    check(lib().sp_sumcheck_cubic3_round0(ctx.h, p64(np.ascontiguousarray(claim, dtype=np.uint64).reshape(4)), p64(taus), ctypes.c_size_t(ell), A.h, B.h, C.h, p0.h, p1.h,
                                          tr.h, p64(polys), p64(r), p64(fin)))
    return polys, r, fin


def sumcheck_quad(ctx, claim, rounds, A: Table, B: Table, tr: Transcript):
    """SumcheckProof::prove_quad (src/sumcheck.rs:190-247)."""
    claim = np.ascontiguousarray(claim, dtype=np.uint64).reshape(4)
    polys = np.zeros((rounds, 2, 4), dtype=np.uint64)
    r = np.zeros((rounds, 4), dtype=np.uint64)
    fin = np.zeros((2, 4), dtype=np.uint64)
    check(lib().sp_sumcheck_quad(ctx.h, p64(claim), ctypes.c_size_t(rounds), A.h, B.h, tr.h, p64(polys), p64(r), p64(fin)))
    return polys, r, fin


def table_dot(ctx, a: Table, b: Table, n: int):
    out = np.zeros(4, dtype=np.uint64)
    check(lib().sp_table_dot(ctx.h, a.h, b.h, ctypes.c_size_t(n), p64(out)))
    return out


# ---- group / MSM / Hyrax -----------------------------------------------------------------------------------------
def msm(ctx, scalars, bases):
    """DlogGroupExt::vartime_multiscalar_mul (src/provider/msm.rs:187-222); returns the affine result (8 limbs)."""
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, 8)
    assert scalars.shape[0] == bases.shape[0]
    out = np.zeros(8, dtype=np.uint64)
    n = scalars.shape[0]
    check(lib().sp_msm(ctx.h, p64(scalars) if n else None, p64(bases) if n else None, ctypes.c_size_t(n), p64(out)))
    return out


class Points:
    """affine points resident in HBM (sp_points_upload)"""

    def __init__(self, ctx, points):
        points = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 8)
        self.ctx, self.n = ctx, points.shape[0]
        self.h = ctypes.c_void_p()
        lib().sp_points_free.argtypes = [ctypes.c_void_p]
        check(lib().sp_points_upload(ctx.h, p64(points), ctypes.c_size_t(self.n), ctypes.byref(self.h)))

    def free(self):
        if self.h:
            lib().sp_points_free(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def msm_points(ctx, scalars: "Table", off, n, points: Points, first=0, window=0):
    """sp_msm_points: sum scalars[off + i] * points[first + i] on device-resident operands; window = 0 (library's choice) or a forced Pippenger width."""
    out = np.zeros(8, dtype=np.uint64)
    check(lib().sp_msm_points(ctx.h, scalars.h, ctypes.c_size_t(off), ctypes.c_size_t(n), points.h, ctypes.c_size_t(first), int(window), p64(out)))
    return out


def msm_eq(ctx, points, r):
    """sum_i eq(r, i) * points[i] through sp_points_upload + sp_msm_eq_begin + sp_msm_job_finish (the homomorphic form of comm_LZ)."""
    points = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 8)
    r = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)
    pts = ctypes.c_void_p()
    job = ctypes.c_void_p()
    out = np.zeros(8, dtype=np.uint64)
    L = lib()
    L.sp_points_free.argtypes = [ctypes.c_void_p]
    check(L.sp_points_upload(ctx.h, p64(points), ctypes.c_size_t(points.shape[0]), ctypes.byref(pts)))
    try:
        check(L.sp_msm_eq_begin(ctx.h, pts, p64(r) if r.shape[0] else None, ctypes.c_size_t(r.shape[0]), ctypes.byref(job)))
        check(L.sp_msm_job_finish(ctx.h, job, p64(out)))
    finally:
        L.sp_points_free(pts)
    return out


def msm_small(ctx, scalars_u64, bases):
    scalars_u64 = np.ascontiguousarray(scalars_u64, dtype=np.uint64).reshape(-1)
    bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, 8)
    out = np.zeros(8, dtype=np.uint64)
    check(lib().sp_msm_small_u64(ctx.h, p64(scalars_u64), p64(bases), ctypes.c_size_t(len(scalars_u64)), p64(out)))
    return out


def vartime_scalar_mul(ctx, points, scalar):
    """vartime_scalar_mul (src/provider/msm.rs:779-867) of every point by one scalar."""
    points = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 8)
    out = np.zeros_like(points)
    check(lib().sp_vartime_scalar_mul(ctx.h, p64(points), ctypes.c_size_t(points.shape[0]), p64(np.ascontiguousarray(scalar, dtype=np.uint64).reshape(4)), p64(out)))
    return out


def fold_commitments2(ctx, p_rows, q_rows, w):
    """FoldingEngineTrait::fold_commitments for weights (1, w) (hyrax_pc.rs:757-776): p[i] + w * q[i]."""
    p_rows = np.ascontiguousarray(p_rows, dtype=np.uint64).reshape(-1, 8)
    q_rows = np.ascontiguousarray(q_rows, dtype=np.uint64).reshape(-1, 8)
    out = np.zeros_like(p_rows)
    check(lib().sp_fold_commitments2(ctx.h, p64(p_rows), p64(q_rows), ctypes.c_size_t(p_rows.shape[0]), p64(np.ascontiguousarray(w, dtype=np.uint64).reshape(4)), p64(out)))
    return out


def fold_commitments2_split(ctx, p_rows, q_rows, w, drop=False):
    """sp_fold_commitments2_begin (q's doubling ladders on the polling host threads) + _finish: p[i] + w * q[i] for a weight that arrives late."""
    p_rows = np.ascontiguousarray(p_rows, dtype=np.uint64).reshape(-1, 8)
    q_rows = np.ascontiguousarray(q_rows, dtype=np.uint64).reshape(-1, 8)
    job = ctypes.c_void_p()
    check(lib().sp_fold_commitments2_begin(ctx.h, p64(q_rows), ctypes.c_size_t(q_rows.shape[0]), ctypes.byref(job)))
    if drop:
        lib().sp_fold_commitments2_drop(job)
        return None
    out = np.zeros_like(p_rows)
    check(lib().sp_fold_commitments2_finish(ctx.h, job, p64(p_rows), p64(np.ascontiguousarray(w, dtype=np.uint64).reshape(4)), p64(out)))
    return out


def eval_cubic_zero_check_round0(ctx, taus, A, B):
    """EqSumCheckInstance::evaluation_points_zero_check_round0 (src/sumcheck.rs:1163-1271) -> (eval_0, eval_2, eval_3)."""
    taus = np.ascontiguousarray(taus, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros((3, 4), dtype=np.uint64)
    check(lib().sp_eval_cubic_zero_check_round0(ctx.h, p64(taus), ctypes.c_size_t(taus.shape[0]), A.h, B.h, p64(out)))
    return out


def point_sum(points):
    """Sum of affine points (combine step of a point-range-sharded MSM)."""
    points = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 8)
    out = np.zeros(8, dtype=np.uint64)
    check(lib().sp_point_sum(p64(points) if points.shape[0] else None, ctypes.c_size_t(points.shape[0]), p64(out)))
    return out


class CommitmentKey:
    """HyraxCommitmentKey (src/provider/pcs/hyrax_pc.rs:56-72) resident on the device."""

    def __init__(self, ctx, ck_aff, h_aff):
        ck_aff = np.ascontiguousarray(ck_aff, dtype=np.uint64).reshape(-1, 8)
        h_aff = np.ascontiguousarray(h_aff, dtype=np.uint64).reshape(8)
        self.ctx = ctx
        self.num_cols = ck_aff.shape[0]
        self.h = ctypes.c_void_p()
        check(lib().sp_ck_create(ctx.h, p64(ck_aff), ctypes.c_size_t(self.num_cols), p64(h_aff), ctypes.byref(self.h)))

    def commit(self, table: Table, off, n, blinds, is_small=True):
        rows = (n + self.num_cols - 1) // self.num_cols
        blinds = np.ascontiguousarray(blinds, dtype=np.uint64).reshape(rows, 4)
        out = np.zeros((rows, 8), dtype=np.uint64)
        check(lib().sp_hyrax_commit(self.ctx.h, self.h, table.h, ctypes.c_size_t(off), ctypes.c_size_t(n), p64(blinds), int(is_small), p64(out)))
        return out

    def commit_rows_host(self, scalars, blinds):
        """sp_hyrax_commit_rows_host: PCS::commit of a host vector on a narrow key, the latency form (one launch through mapped memory, rows added on the host)."""
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        n = scalars.shape[0]
        rows = (n + self.num_cols - 1) // self.num_cols
        blinds = np.ascontiguousarray(blinds, dtype=np.uint64).reshape(rows, 4)
        out = np.zeros((rows, 8), dtype=np.uint64)
        check(lib().sp_hyrax_commit_rows_host(self.ctx.h, self.h, p64(scalars), ctypes.c_size_t(n), p64(blinds), p64(out)))
        return out

    def commit_without_blind(self, table: Table, off, n, is_small=True):
        """PCS::commit_without_blind (hyrax_pc.rs:533-568): the raw per-row MSMs, (0,0) for an all-zero row"""
        rows = (n + self.num_cols - 1) // self.num_cols
        out = np.zeros((rows, 8), dtype=np.uint64)
        check(lib().sp_hyrax_commit_without_blind(self.ctx.h, self.h, table.h, ctypes.c_size_t(off), ctypes.c_size_t(n), int(is_small), p64(out)))
        return out

    def commit_incremental(self, raw_rows, delta: Table, off, n, blinds):
        """PCS::commit_incremental (hyrax_pc.rs:570-607): raw + MSM(delta) + h * blind per row"""
        rows = (n + self.num_cols - 1) // self.num_cols
        raw = np.ascontiguousarray(raw_rows, dtype=np.uint64).reshape(-1, 8)
        blinds = np.ascontiguousarray(blinds, dtype=np.uint64).reshape(rows, 4)
        out = np.zeros((rows, 8), dtype=np.uint64)
        check(lib().sp_hyrax_commit_incremental(self.ctx.h, self.h, p64(raw) if raw.size else None, ctypes.c_size_t(raw.shape[0]), delta.h, ctypes.c_size_t(off),
                                                ctypes.c_size_t(n), p64(blinds), p64(out)))
        return out

    def prove(self, key_eval, tr, comm_rows, poly, n, blinds, point, comm_eval, blind_eval, rng):
        """HyraxPCS::prove (hyrax_pc.rs:387-478) as one ABI call (sp_hyrax_prove): rng = (>= cols + 2, 64) uniform bytes (d_vec, r_delta, r_beta in draw
        order); returns delta (8) | beta (8) | z_vec | z_delta | z_beta words."""
        comm_rows = np.ascontiguousarray(comm_rows, dtype=np.uint64).reshape(-1, 8)
        point = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 4)
        rows = comm_rows.shape[0]
        cols = n // rows
        rng = np.ascontiguousarray(rng, dtype=np.uint8).reshape(-1, 64)
        out = np.zeros(16 + 4 * cols + 8, dtype=np.uint64)
        c = lambda a, shape: np.ascontiguousarray(a, dtype=np.uint64).reshape(shape)
        check(lib().sp_hyrax_prove(self.ctx.h, self.h, key_eval.h, tr.h, p64(comm_rows), ctypes.c_size_t(rows), poly.h, ctypes.c_size_t(n), p64(c(blinds, (rows, 4))),
                                   p64(point), ctypes.c_size_t(point.shape[0]), p64(c(comm_eval, (8,))), p64(c(blind_eval, (4,))), p8(rng), ctypes.c_size_t(rng.shape[0]),
                                   p64(out)))
        return out

    def prove_announce(self, comm_rows, poly, n, blinds, rng):
        """sp_hyrax_prove_announce: the opening that the next prove() on this context will be asked for (its delta / hashing / mask vector start now)."""
        comm_rows = np.ascontiguousarray(comm_rows, dtype=np.uint64).reshape(-1, 8)
        rows = comm_rows.shape[0]
        rng = np.ascontiguousarray(rng, dtype=np.uint8).reshape(-1, 64)
        check(lib().sp_hyrax_prove_announce(self.ctx.h, self.h, p64(comm_rows), ctypes.c_size_t(rows), poly.h, ctypes.c_size_t(n),
                                            p64(np.ascontiguousarray(blinds, dtype=np.uint64).reshape(rows, 4)), p8(rng), ctypes.c_size_t(rng.shape[0])))

    def prove_announce_tables(self, comm_rows, poly, n, blinds, rng, row_tables, nfixed):
        """sp_hyrax_prove_announce_tables: the same with FixedBaseMul tables (FbTables) of the first nfixed rows and of h; rows from nfixed on are blind * h."""
        comm_rows = np.ascontiguousarray(comm_rows, dtype=np.uint64).reshape(-1, 8)
        rows = comm_rows.shape[0]
        rng = np.ascontiguousarray(rng, dtype=np.uint8).reshape(-1, 64)
        check(lib().sp_hyrax_prove_announce_tables(self.ctx.h, self.h, p64(comm_rows), ctypes.c_size_t(rows), poly.h, ctypes.c_size_t(n),
                                                   p64(np.ascontiguousarray(blinds, dtype=np.uint64).reshape(rows, 4)), p8(rng), ctypes.c_size_t(rng.shape[0]),
                                                   row_tables.h, ctypes.c_size_t(nfixed)))

    def prove_retract(self):
        check(lib().sp_hyrax_prove_retract(self.ctx.h))

    def rerandomize(self, comm_rows, r_old, r_new):
        """PCS::rerandomize_commitment (hyrax_pc.rs:321-344)."""
        comm_rows = np.ascontiguousarray(comm_rows, dtype=np.uint64).reshape(-1, 8)
        rows = comm_rows.shape[0]
        out = np.zeros_like(comm_rows)
        check(lib().sp_hyrax_rerandomize(self.ctx.h, self.h, p64(comm_rows), ctypes.c_size_t(rows), p64(np.ascontiguousarray(r_old, dtype=np.uint64).reshape(rows, 4)),
                                         p64(np.ascontiguousarray(r_new, dtype=np.uint64).reshape(rows, 4)), p64(out)))
        return out

    def fixed_base_mul_h(self, scalars):
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros((scalars.shape[0], 8), dtype=np.uint64)
        check(lib().sp_fixed_base_mul_h(self.ctx.h, self.h, p64(scalars), ctypes.c_size_t(scalars.shape[0]), p64(out)))
        return out

    def fixed_base_mul_h_begin(self, scalars):
        """asynchronous form (auxiliary stream); returns the job handle for fixed_base_mul_h_finish. One job per context at a time."""
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        job = ctypes.c_void_p()
        check(lib().sp_fixed_base_mul_h_begin(self.ctx.h, self.h, p64(scalars), ctypes.c_size_t(scalars.shape[0]), ctypes.byref(job)))
        return job, scalars.shape[0]

    def fixed_base_mul_h_finish(self, job_n):
        job, n = job_n
        out = np.zeros((n, 8), dtype=np.uint64)
        check(lib().sp_fixed_base_mul_h_finish(self.ctx.h, job, p64(out)))
        return out

    def msm(self, scalars, blind=None):
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros(8, dtype=np.uint64)
        b = None if blind is None else p64(np.ascontiguousarray(blind, dtype=np.uint64).reshape(4))
        check(lib().sp_msm_ck(self.ctx.h, self.h, p64(scalars), ctypes.c_size_t(scalars.shape[0]), b, p64(out)))
        return out

    def msm_range(self, scalars, first: int, blind=None):
        """<scalars, ck[first : first + n]> through the asynchronous pair sp_msm_ck_range_begin / sp_msm_ck_finish (a rank's point range of a
        sharded MSM)."""
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros(8, dtype=np.uint64)
        job = ctypes.c_void_p()
        check(lib().sp_msm_ck_range_begin(self.ctx.h, self.h, p64(scalars), ctypes.c_size_t(first), ctypes.c_size_t(scalars.shape[0]), ctypes.byref(job)))
        b = None if blind is None else p64(np.ascontiguousarray(blind, dtype=np.uint64).reshape(4))
        check(lib().sp_msm_ck_finish(self.ctx.h, self.h, job, b, p64(out)))
        return out

    def commit_small(self, scalars, blind):
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        blind = np.ascontiguousarray(blind, dtype=np.uint64).reshape(4)
        out = np.zeros(8, dtype=np.uint64)
        check(lib().sp_hyrax_commit_small(self.ctx.h, self.h, p64(scalars), ctypes.c_size_t(scalars.shape[0]), p64(blind), p64(out)))
        return out

    def commit_split_available(self, cols_used=16):
        return bool(lib().sp_hyrax_commit_split_available(self.h, ctypes.c_size_t(cols_used)))

    def commit_split(self, early_cols, early_scalars, blind, late_cols, late_scalars, drop=False):
        """sp_hyrax_commit_split_begin (terms known early, the blind or None) + _finish (the rest): the commitment of the assembled row."""
        ec = np.ascontiguousarray(early_cols, dtype=np.uint32).reshape(-1)
        es = np.ascontiguousarray(early_scalars, dtype=np.uint64).reshape(-1, 4)
        lc = np.ascontiguousarray(late_cols, dtype=np.uint32).reshape(-1)
        ls = np.ascontiguousarray(late_scalars, dtype=np.uint64).reshape(-1, 4)
        assert ec.shape[0] == es.shape[0] and lc.shape[0] == ls.shape[0]
        bp = None if blind is None else p64(np.ascontiguousarray(blind, dtype=np.uint64).reshape(4))
        job = ctypes.c_void_p()
        pu32 = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
        check(lib().sp_hyrax_commit_split_begin(self.ctx.h, self.h, pu32(ec), p64(es), ctypes.c_size_t(ec.shape[0]), bp, ctypes.byref(job)))
        if drop:
            lib().sp_hyrax_commit_split_drop(job)
            return None
        out = np.zeros(8, dtype=np.uint64)
        check(lib().sp_hyrax_commit_split_finish(self.ctx.h, job, pu32(lc), p64(ls), ctypes.c_size_t(lc.shape[0]), p64(out)))
        return out

    def commit_small_with_term(self, scalars, blind_term_aff):
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        term = np.ascontiguousarray(blind_term_aff, dtype=np.uint64).reshape(8)
        out = np.zeros(8, dtype=np.uint64)
        check(lib().sp_hyrax_commit_small_with_term(self.ctx.h, self.h, p64(scalars), ctypes.c_size_t(scalars.shape[0]), p64(term), p64(out)))
        return out

    def __del__(self):
        try:
            if self.h:
                lib().sp_ck_free(self.h)
        except Exception:
            pass


class FixedBaseTables:
    """FixedBaseMul::precompute over n <= 512 points + multi_mul in one launch (src/provider/msm.rs:637-773): sp_fbtables_*."""

    def __init__(self, ctx: Context, points, queued=False):
        """queued: sp_fbtables_create_async - returns once the build is queued on the context's table stream (ready() tells / waits)"""
        points = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 8)
        self.ctx, self.n = ctx, points.shape[0]
        self.h = ctypes.c_void_p()
        fn = lib().sp_fbtables_create_async if queued else lib().sp_fbtables_create
        check(fn(ctx.h, p64(points), ctypes.c_size_t(self.n), ctypes.byref(self.h)))

    def ready(self, wait=False):
        r = lib().sp_fbtables_ready(self.h, int(wait))
        if r < 0:
            check(r)
        return r == 1

    def read(self, first, count):
        """count table entries (affine points) from entry `first` of the n x 32 x 255 array: entry (i, j, d - 1) = d 2^(8j) point_i"""
        out = np.zeros((count, 8), dtype=np.uint64)
        check(lib().sp_fbtables_read(self.h, ctypes.c_size_t(first), ctypes.c_size_t(count), p64(out)))
        return out

    def multi_mul(self, scalars):
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros(8, dtype=np.uint64)
        check(lib().sp_fbtables_multi_mul(self.ctx.h, self.h, p64(scalars), ctypes.c_size_t(scalars.shape[0]), p64(out)))
        return out

    def multi_mul_eq(self, P, nfixed, s01, r_last):
        """sp_fbtables_multi_mul_begin_eq + _finish: the scalars are eq(r_1..r_k, .) handed over one level short (P = eq(r_1..r_(k-1), .)), the last table's
        scalar is S0 + r_k (S1 - S0)."""
        P = np.ascontiguousarray(P, dtype=np.uint64).reshape(-1, 4)
        s01 = np.ascontiguousarray(s01, dtype=np.uint64).reshape(2, 4)
        r_last = np.ascontiguousarray(r_last, dtype=np.uint64).reshape(4)
        out = np.zeros(8, dtype=np.uint64)
        check(lib().sp_fbtables_multi_mul_begin_eq(self.ctx.h, self.h, p64(P), ctypes.c_size_t(nfixed), p64(s01), p64(r_last)))
        check(lib().sp_fbtables_multi_mul_finish(self.ctx.h, p64(out)))
        return out

    def close(self):
        if self.h:
            lib().sp_fbtables_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rowmat_vec(ctx, poly: Table, rows, cols, L):
    """bind_with_delayed (src/provider/pcs/hyrax_pc.rs:38-54)."""
    L = np.ascontiguousarray(L, dtype=np.uint64).reshape(rows, 4)
    out = np.zeros((cols, 4), dtype=np.uint64)
    check(lib().sp_rowmat_vec(ctx.h, poly.h, ctypes.c_size_t(rows), ctypes.c_size_t(cols), p64(L), p64(out)))
    return out


def rowmat_vec_eq(ctx, poly: Table, r, cols, addend=None, scale=None):
    """bind_with_delayed with L = eq(r, .) as an asynchronous job (sp_rowmat_vec_eq_begin[_with] / _finish[_scaled]): LZ, or scale * LZ + addend
    (z_vec of InnerProductArgumentLinear::prove, ipa.rs:160-163) when both are given."""
    r = np.ascontiguousarray(r, dtype=np.uint64).reshape(-1, 4)
    job = ctypes.c_void_p()
    out = np.zeros((cols, 4), dtype=np.uint64)
    if addend is None:
        check(lib().sp_rowmat_vec_eq_begin(ctx.h, poly.h, p64(r), ctypes.c_size_t(len(r)), ctypes.c_size_t(cols), ctypes.byref(job)))
    else:
        addend = np.ascontiguousarray(addend, dtype=np.uint64).reshape(cols, 4)
        check(lib().sp_rowmat_vec_eq_begin_with(ctx.h, poly.h, p64(r), ctypes.c_size_t(len(r)), ctypes.c_size_t(cols), p64(addend), ctypes.byref(job)))
    if scale is None:
        check(lib().sp_rowmat_vec_eq_finish(ctx.h, job, p64(out)))
    else:
        scale = np.ascontiguousarray(scale, dtype=np.uint64).reshape(4)
        check(lib().sp_rowmat_vec_eq_finish_scaled(ctx.h, job, p64(scale), p64(out)))
    return out


# ---- R1CS ----------------------------------------------------------------------------------------------------------
class _Csr(ctypes.Structure):
    _fields_ = [("data", c_u64p), ("indices", ctypes.POINTER(ctypes.c_uint32)), ("indptr", c_u64p)]


class _Dims(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("num_cons", "num_cons_unpadded", "num_shared", "num_precommitted", "num_rest", "num_shared_unpadded",
                                               "num_precommitted_unpadded", "num_rest_unpadded", "num_public", "num_challenges")]


class Shape:
    """SplitR1CSShape on the device. mats: 3 x (data F (nnz,4) uint64, indices uint32, indptr uint64) in the padded layout."""

    def __init__(self, ctx, mats, dims: dict):
        self.ctx = ctx
        self._keep = []
        cs = []
        for d, i, p_ in mats:
            d = np.ascontiguousarray(d, dtype=np.uint64).reshape(-1, 4)
            i = np.ascontiguousarray(i, dtype=np.uint32)
            p_ = np.ascontiguousarray(p_, dtype=np.uint64)
            self._keep += [d, i, p_]
            cs.append(_Csr(p64(d) if d.shape[0] else None, i.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), p64(p_)))
        dd = _Dims(**{k: int(v) for k, v in dims.items()})
        self.dims = dims
        self.h = ctypes.c_void_p()
        check(lib().sp_shape_from_csr(ctx.h, ctypes.byref(cs[0]), ctypes.byref(cs[1]), ctypes.byref(cs[2]), ctypes.byref(dd), ctypes.byref(self.h)))

    def multiply_vec(self, z: Table, az: Table, bz: Table, cz: Table):
        check(lib().sp_multiply_vec(self.ctx.h, self.h, z.h, az.h, bz.h, cz.h))

    def multiply_vec_batched(self, zs, azs, bzs, czs):
        """SplitR1CSShape::multiply_vec_batched (src/r1cs/mod.rs:1130-1166): the three products for every z in zs."""
        n = len(zs)
        arr = lambda ts: (ctypes.c_void_p * n)(*[t.h for t in ts])
        check(lib().sp_multiply_vec_batched(self.ctx.h, self.h, arr(zs), ctypes.c_size_t(n), arr(azs), arr(bzs), arr(czs)))

    def multiply_vec_incremental_round0(self, z, caz, cbz, ccz, az, bz, cz, p0, p1):
        check(lib().sp_multiply_vec_incremental_round0(self.ctx.h, self.h, z.h, caz.h, cbz.h, ccz.h, az.h, bz.h, cz.h, p0.h, p1.h))

    def multiply_vec_incremental(self, z, caz, cbz, ccz, az, bz, cz):
        check(lib().sp_multiply_vec_incremental(self.ctx.h, self.h, z.h, caz.h, cbz.h, ccz.h, az.h, bz.h, cz.h))

    def poly_abc(self, rx: Table, r, out_len, out: Table):
        r = np.ascontiguousarray(r, dtype=np.uint64).reshape(4)
        check(lib().sp_poly_abc(self.ctx.h, self.h, rx.h, p64(r), ctypes.c_size_t(out_len), out.h))

    def __del__(self):
        try:
            if self.h:
                lib().sp_shape_free(self.h)
        except Exception:
            pass


# ---- NeutronNova kernel-level rows ------------------------------------------------------------------------------------------
def weights_from_r(r_bs, n: int):
    """weights_from_r (src/r1cs/mod.rs:153-166)."""
    r_bs = np.ascontiguousarray(r_bs, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros((n, 4), dtype=np.uint64)
    check(lib().sp_weights_from_r(p64(r_bs) if r_bs.shape[0] else None, ctypes.c_size_t(r_bs.shape[0]), ctypes.c_size_t(n), p64(out)))
    return out


def fold_tables(ctx, tables, weights, length: int, out: Table):
    """R1CSWitness::fold_multiple (src/r1cs/mod.rs:570-660): out[j] = sum_i weights[i] * tables[i][j]."""
    weights = np.ascontiguousarray(weights, dtype=np.uint64).reshape(len(tables), 4)
    arr = (ctypes.c_void_p * len(tables))(*[t.h for t in tables])
    check(lib().sp_fold_tables(ctx.h, arr, ctypes.c_size_t(len(tables)), p64(weights), ctypes.c_size_t(length), out.h))


def msm_shared_weights(ctx, weights, bases_rows):
    """vartime_multiscalar_mul_shared_weights (src/provider/msm.rs:228-356); bases_rows: (rows, n, 8)."""
    bases_rows = np.ascontiguousarray(bases_rows, dtype=np.uint64)
    rows, n = bases_rows.shape[0], bases_rows.shape[1]
    weights = np.ascontiguousarray(weights, dtype=np.uint64).reshape(n, 4)
    out = np.zeros((rows, 8), dtype=np.uint64)
    check(lib().sp_msm_shared_weights(ctx.h, p64(weights), ctypes.c_size_t(n), p64(bases_rows.reshape(-1)), ctypes.c_size_t(rows), p64(out)))
    return out


def eval_cubic_outer_pow(ctx, pow_left: Table, pow_right, A: Table, B: Table, C: Table):
    """compute_eval_points_cubic_with_additive_term_with_outer_pow (src/sumcheck.rs:366-498) -> (eval0, eval2, eval3)."""
    out = np.zeros((3, 4), dtype=np.uint64)
    check(lib().sp_eval_cubic_outer_pow(ctx.h, pow_left.h, pow_right.h if pow_right is not None else None, A.h, B.h, C.h, p64(out)))
    return out


# ---- NeutronNova NIFS rounds (src/neutronnova_zk.rs:511-1273) ----------------------------------------------------------------
def pow_split_evals(tau, ell: int, left: int, right: int):
    """PowPolynomial::split_evals (src/polys/power.rs:64-87)."""
    out = np.zeros((left + right, 4), dtype=np.uint64)
    check(lib().sp_pow_split_evals(p64(np.ascontiguousarray(tau, dtype=np.uint64).reshape(4)), ctypes.c_size_t(ell), ctypes.c_size_t(left), ctypes.c_size_t(right),
                                   p64(out)))
    return out


def to_small_vec_or_zero(ctx, table: Table, cnt: int):
    """to_small_vec_or_zero (src/big_num/small_value.rs:41-86) -> (i64 values, indices of the large positions)."""
    out = np.zeros(cnt, dtype=np.int64)
    large = np.zeros(cnt, dtype=np.uint8)
    check(lib().sp_to_small_vec_or_zero(ctx.h, table.h, ctypes.c_size_t(cnt), out.ctypes.data_as(ctypes.c_void_p), p8(large) if cnt else None))
    return out, np.nonzero(large)[0]


class Nifs:
    """Device state of NeutronNovaNIFS::prove: instance layers Az/Bz/Cz, the rounds, the final folded layers."""

    def __init__(self, ctx: Context, n_padded: int, left: int, right: int):
        self.ctx, self.n_padded, self.left, self.right = ctx, n_padded, left, right
        self.h = ctypes.c_void_p()
        check(lib().sp_nifs_create(ctx.h, ctypes.c_size_t(n_padded), ctypes.c_size_t(left), ctypes.c_size_t(right), ctypes.byref(self.h)))

    def layer(self, which: int, idx: int) -> Table:
        h = ctypes.c_void_p()
        check(lib().sp_nifs_layer(self.h, int(which), ctypes.c_size_t(idx), ctypes.byref(h)))
        return Table(self.ctx, h)

    def begin(self, E_eq, rhos, small_values=False):
        E_eq = np.ascontiguousarray(E_eq, dtype=np.uint64).reshape(self.left + self.right, 4)
        rhos = np.ascontiguousarray(rhos, dtype=np.uint64).reshape(-1, 4)
        check(lib().sp_nifs_begin(self.h, p64(E_eq), p64(rhos), ctypes.c_size_t(rhos.shape[0]), 1 if small_values else 0))

    def round(self, t: int):
        out = np.zeros((4, 4), dtype=np.uint64)
        check(lib().sp_nifs_round(self.h, ctypes.c_size_t(t), p64(out)))
        return out

    def challenge(self, r_b):
        check(lib().sp_nifs_challenge(self.h, p64(np.ascontiguousarray(r_b, dtype=np.uint64).reshape(4))))

    def finish(self, A: Table, B: Table, C: Table = None):
        """C = None skips the C fold (sharded batches fold C per shard)."""
        T, eq = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
        check(lib().sp_nifs_finish(self.h, A.h, B.h, C.h if C is not None else None, p64(T), p64(eq)))
        return T, eq

    # ---- sharded batches ----
    def begin_shard(self, E_eq, rhos, first_instance: int, small_values=False):
        E_eq = np.ascontiguousarray(E_eq, dtype=np.uint64).reshape(self.left + self.right, 4)
        rhos = np.ascontiguousarray(rhos, dtype=np.uint64).reshape(-1, 4)
        check(lib().sp_nifs_begin_shard(self.h, p64(E_eq), p64(rhos), ctypes.c_size_t(rhos.shape[0]), ctypes.c_size_t(first_instance), 1 if small_values else 0))

    def cvals(self):
        out = np.zeros((self.n_padded, 4), dtype=np.uint64)
        check(lib().sp_nifs_cvals(self.h, p64(out)))
        return out

    def set_cvals(self, allv):
        allv = np.ascontiguousarray(allv, dtype=np.uint64).reshape(-1, 4)
        check(lib().sp_nifs_set_cvals(self.h, p64(allv), ctypes.c_size_t(allv.shape[0])))

    def round_sums(self, t: int):
        out = np.zeros((2, 4), dtype=np.uint64)
        check(lib().sp_nifs_round_sums(self.h, ctypes.c_size_t(t), p64(out)))
        return out

    def round_finish(self, t: int, sums):
        out = np.zeros((4, 4), dtype=np.uint64)
        check(lib().sp_nifs_round_finish(self.h, ctypes.c_size_t(t), p64(np.ascontiguousarray(sums, dtype=np.uint64).reshape(2, 4)), p64(out)))
        return out

    def fold_pending(self):
        check(lib().sp_nifs_fold_pending(self.h))

    def current_layer(self, which: int, idx: int) -> Table:
        h = ctypes.c_void_p()
        check(lib().sp_nifs_current_layer(self.h, int(which), ctypes.c_size_t(idx), ctypes.byref(h)))
        return Table(self.ctx, h)

    def state(self):
        T, eq = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
        check(lib().sp_nifs_state(self.h, p64(T), p64(eq)))
        return T, eq

    def resume(self, E_eq, rhos, t_start: int, r_bs, T_cur, acc_eq, c_vals_all):
        E_eq = np.ascontiguousarray(E_eq, dtype=np.uint64).reshape(self.left + self.right, 4)
        rhos = np.ascontiguousarray(rhos, dtype=np.uint64).reshape(-1, 4)
        r_bs = np.ascontiguousarray(r_bs, dtype=np.uint64).reshape(t_start, 4)
        cv = np.ascontiguousarray(c_vals_all, dtype=np.uint64).reshape(-1, 4)
        check(lib().sp_nifs_resume(self.h, p64(E_eq), p64(rhos), ctypes.c_size_t(rhos.shape[0]), ctypes.c_size_t(t_start), p64(r_bs),
                                   p64(np.ascontiguousarray(T_cur, dtype=np.uint64).reshape(4)), p64(np.ascontiguousarray(acc_eq, dtype=np.uint64).reshape(4)), p64(cv)))

    def free(self):
        if self.h:
            lib().sp_nifs_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ---- NeutronNova batched ZK sum-checks (src/sumcheck.rs:702-917) ---------------------------------------------------------------
ROUND_HOOK = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, c_u64p, c_u64p, ctypes.c_size_t, c_u64p)


def _round_hook(py_hook):
    def raw(_user, rnd, cs, cc, ncoeffs, out_ptr):
        try:
            s = np.ctypeslib.as_array(cs, shape=(4 * ncoeffs,)).reshape(ncoeffs, 4).copy()
            k = np.ctypeslib.as_array(cc, shape=(4 * ncoeffs,)).reshape(ncoeffs, 4).copy()
            r = np.ascontiguousarray(py_hook(int(rnd), s, k), dtype=np.uint64).reshape(4)
            for i in range(4):
                out_ptr[i] = int(r[i])
            return 0
        except Exception:  # surfaces as SP_ERR_INTERNAL from the sum-check
            return -5

    return ROUND_HOOK(raw)


def sumcheck_quad_batched(ctx, claims, num_rounds, A0: Table, A1: Table, B0: Table, B1: Table, start_round, py_hook):
    """prove_quad_batched_zk (src/sumcheck.rs:702-782) -> (r_y, [A0[0], A1[0], B0[0], B1[0]])."""
    claims = np.ascontiguousarray(claims, dtype=np.uint64).reshape(2, 4)
    out_r = np.zeros((num_rounds, 4), dtype=np.uint64)
    fin = np.zeros((4, 4), dtype=np.uint64)
    cb = _round_hook(py_hook)
    check(lib().sp_sumcheck_quad_batched(ctx.h, p64(claims), ctypes.c_size_t(num_rounds), A0.h, A1.h, B0.h, B1.h, ctypes.c_size_t(start_round), cb, None, p64(out_r),
                                         p64(fin)))
    return out_r, fin


def sumcheck_cubic_outer_pow_batched(ctx, num_rounds, pow_left: Table, pow_right: Table, step, core, t_out_step, start_round, py_hook):
    """prove_cubic_with_additive_term_batched_zk (src/sumcheck.rs:786-917); step / core = (A, B, C) tables -> r_x."""
    out_r = np.zeros((num_rounds, 4), dtype=np.uint64)
    cb = _round_hook(py_hook)
    t = np.ascontiguousarray(t_out_step, dtype=np.uint64).reshape(4)
    check(lib().sp_sumcheck_cubic_outer_pow_batched(ctx.h, ctypes.c_size_t(num_rounds), pow_left.h, pow_right.h, step[0].h, step[1].h, step[2].h, core[0].h, core[1].h,
                                                    core[2].h, p64(t), ctypes.c_size_t(start_round), cb, None, p64(out_r)))
    return out_r


# ---- sum-checks on a table slice -----------------------------------------------------------------------------------
REDUCE_HOOK = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, c_u64p, ctypes.c_size_t)


def _reduce_hook(py_reduce):
    def raw(_user, sums_ptr, count):
        try:
            arr = np.ctypeslib.as_array(sums_ptr, shape=(4 * count,)).reshape(count, 4)
            out = np.ascontiguousarray(py_reduce(arr.copy()), dtype=np.uint64).reshape(count, 4)
            arr[:] = out
            return 0
        except Exception:
            return -5

    return REDUCE_HOOK(raw)


def sumcheck_cubic3_sharded(ctx, claim, p, taus, A: Table, B: Table, C: Table, tr: Transcript, scale=None, py_reduce=None):
    """sp_sumcheck_cubic3_sharded -> (polys (ell,3,4), r (ell,4), final (3,4), claim_out, p_out)."""
    taus = np.ascontiguousarray(taus, dtype=np.uint64).reshape(-1, 4)
    ell = taus.shape[0]
    claim_io = np.ascontiguousarray(claim, dtype=np.uint64).reshape(4).copy()
    p_io = np.ascontiguousarray(p, dtype=np.uint64).reshape(4).copy()
    polys = np.zeros((ell, 3, 4), dtype=np.uint64)
    r = np.zeros((ell, 4), dtype=np.uint64)
    fin = np.zeros((3, 4), dtype=np.uint64)
    cb = _reduce_hook(py_reduce) if py_reduce is not None else None
    sc = np.ascontiguousarray(scale, dtype=np.uint64).reshape(4) if scale is not None else None
    check(lib().sp_sumcheck_cubic3_sharded(ctx.h, p64(claim_io), p64(p_io), p64(taus), ctypes.c_size_t(ell), A.h, B.h, C.h, tr.h, p64(sc) if sc is not None else None,
                                           cb if cb is not None else ctypes.cast(None, REDUCE_HOOK), None, p64(polys), p64(r), p64(fin)))
    return polys, r, fin, claim_io, p_io


def sumcheck_quad_sharded(ctx, claim, rounds, A: Table, B: Table, tr: Transcript, py_reduce=None):
    """sp_sumcheck_quad_sharded -> (polys (rounds,2,4), r, final (2,4), claim_out)."""
    claim_io = np.ascontiguousarray(claim, dtype=np.uint64).reshape(4).copy()
    polys = np.zeros((rounds, 2, 4), dtype=np.uint64)
    r = np.zeros((rounds, 4), dtype=np.uint64)
    fin = np.zeros((2, 4), dtype=np.uint64)
    cb = _reduce_hook(py_reduce) if py_reduce is not None else None
    check(lib().sp_sumcheck_quad_sharded(ctx.h, p64(claim_io), ctypes.c_size_t(rounds), A.h, B.h, tr.h, cb if cb is not None else ctypes.cast(None, REDUCE_HOOK), None,
                                         p64(polys), p64(r), p64(fin)))
    return polys, r, fin, claim_io


# ---- wire formats and key digests (include/spartan_hip.h "wire formats"; host code of the library, no device needed) ------------------------------
class SpartanLayout(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in ("rows_shared", "rows_precommitted", "rows_rest", "num_public", "num_challenges", "rounds_x", "rounds_y", "z_len")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class CsrView(ctypes.Structure):  # sp_csr
    _fields_ = [("data", c_u64p), ("indices", ctypes.POINTER(ctypes.c_uint32)), ("indptr", c_u64p)]


def sha256(data: bytes) -> bytes:
    arr, n = _bytes(data)
    out = np.zeros(32, dtype=np.uint8)
    check(lib().sp_sha256(p8(arr), n, p8(out)))
    return out.tobytes()


def proof_serialize(layout: dict, words) -> bytes:
    """flat proof words -> bincode bytes of SpartanSNARK (sp_proof_serialize)"""
    L = SpartanLayout(**layout)
    words = np.ascontiguousarray(words, dtype=np.uint64)
    n = ctypes.c_size_t(0)
    check(lib().sp_proof_serialize(ctypes.byref(L), p64(words), ctypes.c_size_t(len(words)), None, ctypes.c_size_t(0), ctypes.byref(n)))
    out = np.zeros(max(n.value, 1), dtype=np.uint8)
    check(lib().sp_proof_serialize(ctypes.byref(L), p64(words), ctypes.c_size_t(len(words)), p8(out), ctypes.c_size_t(n.value), ctypes.byref(n)))
    return out[:n.value].tobytes()


def proof_deserialize(data: bytes):
    """bincode bytes -> (layout dict, flat proof words); raises SpartanHipError when the bytes do not decode"""
    arr, n = _bytes(data)
    L = SpartanLayout()
    nw = ctypes.c_size_t(0)
    check(lib().sp_proof_deserialize(p8(arr), n, ctypes.byref(L), None, ctypes.c_size_t(0), ctypes.byref(nw)))
    words = np.zeros(max(nw.value, 1), dtype=np.uint64)
    check(lib().sp_proof_deserialize(p8(arr), n, ctypes.byref(L), p64(words), ctypes.c_size_t(nw.value), ctypes.byref(nw)))
    return L.as_dict(), words[:nw.value]


def vk_digest(dims, csr_field, ck, h, ck_s, h_s) -> bytes:
    """sp_vk_digest. dims: the ten numbers in sp_dims order; csr_field: three (data (nnz, 4) u64, indices u32, indptr u64) of the PADDED shape."""
    d = (ctypes.c_uint64 * 10)(*[int(x) for x in dims])
    keep, views = [], []
    for data, idx, ptr in csr_field:
        data = np.ascontiguousarray(data, dtype=np.uint64)
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        ptr = np.ascontiguousarray(ptr, dtype=np.uint64)
        keep += [data, idx, ptr]
        views.append(CsrView(p64(data) if data.size else None, idx.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)) if idx.size else None, p64(ptr)))
    ck = np.ascontiguousarray(ck, dtype=np.uint64).reshape(-1, 8)
    ck_s = np.ascontiguousarray(ck_s, dtype=np.uint64).reshape(-1, 8)
    h = np.ascontiguousarray(h, dtype=np.uint64)
    h_s = np.ascontiguousarray(h_s, dtype=np.uint64)
    out = np.zeros(32, dtype=np.uint8)
    check(lib().sp_vk_digest(d, ctypes.byref(views[0]), ctypes.byref(views[1]), ctypes.byref(views[2]), p64(ck), ctypes.c_size_t(ck.shape[0]), p64(h), p64(ck_s),
                             ctypes.c_size_t(ck_s.shape[0]), p64(h_s), p8(out)))
    return out.tobytes()
