"""A Python-integer SpartanSNARK::verify, written from the reference alone and independent of both C++ verifiers (oracle/spartan.hpp verify and
spartan2_amd/host/spartan_snark.cpp verify):

  SpartanSNARK::verify                      src/spartan.rs:469-578
  Keccak256Transcript                       src/provider/keccak.rs:18-104 (Keccak-f[1600] below is FIPS 202's, padding 0x01: pre-standard Keccak)
  transcript encodings                      src/provider/traits.rs:282-305 (big-endian coordinates), src/polys/univariate.rs:182-190 (to_repr),
                                            src/provider/pcs/hyrax_pc.rs:718-728, src/provider/pcs/ipa.rs:62-69, src/traits/transcript.rs:35-42
  SplitR1CSInstance::validate / to_regular  src/r1cs/mod.rs:1490-1550
  SumcheckProof::verify, decompress         src/sumcheck.rs:67-114, src/polys/univariate.rs:136-179
  SparsePolynomial::evaluate                src/polys/multilinear.rs:190-207
  evaluate_with_tables_fast                 src/r1cs/mod.rs:1216-1226 (= sum of val * T_x[row] * T_y[col] over the padded matrices)
  HyraxPCS::verify                          src/provider/pcs/hyrax_pc.rs:480-531
  InnerProductArgumentLinear::verify        src/provider/pcs/ipa.rs:173-221

Inputs: the circuit (integer CSR triples), the keys as affine limb arrays, the proof as the flat word layout of DESIGN.md section 4. Everything is
converted to Python integers first; the group law is textbook Jacobian arithmetic on y^2 = x^3 - 3x + b over the reference's base modulus
(pt256.rs:56; a and b are halo2curves::t256's, third-party: every point that enters is checked against the equation, so a wrong b cannot pass).
Raises VerifyError naming the failed check; returns the public values."""
import numpy as np

import pywire
from pywire import P_BASE as P
from pywire import P_SCALAR as Q

CURVE_B = 0xB441071B12F4A0366FB552F8E21ED4AC36B06ACEEB354224863E60F20219FC56


class VerifyError(Exception):
    pass


# ---- Keccak-256 (the sha3 crate's Keccak256: rate 136, padding 0x01 .. 0x80) ------------------------------------------------------------------------
_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001, 0x8000000080008081,
       0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A, 0x000000008000808B, 0x800000000000008B,
       0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A, 0x8000000080008081,
       0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]  # [x][y]
_M64 = (1 << 64) - 1


def _keccak_f(a):  # a[x][y]
    rol = lambda v, n: ((v << n) | (v >> (64 - n))) & _M64 if n else v
    for rc in _RC:
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = rol(a[x][y], _ROT[x][y])
        a = [[b[x][y] ^ ((~b[(x + 1) % 5][y]) & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        a[0][0] ^= rc
    return a


def keccak256(data: bytes) -> bytes:
    rate = 136
    msg = bytearray(data)
    msg.append(0x01)
    msg.extend(b"\x00" * (-len(msg) % rate))
    msg[-1] |= 0x80
    a = [[0] * 5 for _ in range(5)]
    for off in range(0, len(msg), rate):
        for i in range(rate // 8):
            a[i % 5][i // 5] ^= int.from_bytes(msg[off + 8 * i:off + 8 * i + 8], "little")
        a = _keccak_f(a)
    return b"".join(a[i % 5][i // 5].to_bytes(8, "little") for i in range(4))


class Transcript:
    """Keccak256Transcript (keccak.rs:24-104)"""

    def __init__(self, label: bytes):
        self.round = 0
        self.pending = b""
        self.state = self._updated(b"", b"NoTR" + label)

    @staticmethod
    def _updated(acc, inp):  # compute_updated_state (:33-54)
        data = acc + inp
        return keccak256(data + b"\x00") + keccak256(data + b"\x01")

    def squeeze(self, label: bytes) -> int:
        out = self._updated(self.pending, b"NoDS" + self.round.to_bytes(2, "little") + self.state + label)
        self.round += 1
        self.state = out
        self.pending = b""
        return int.from_bytes(out, "little") % Q  # from_uniform (traits.rs:275-280): the 512-bit little-endian integer reduced

    def absorb(self, label: bytes, data: bytes):
        self.pending += label + data

    def dom_sep(self, data: bytes):
        self.pending += b"NoDS" + data


# ---- transcript encodings ------------------------------------------------------------------------------------------------------------------------
def scalar_bytes(v):  # to_bytes().rev(): big-endian
    return v.to_bytes(32, "big")


def point_bytes(pt):  # affine x | y, big-endian each; the reference unwraps the coordinates (identity never reaches the transcript)
    if pt is None:
        raise VerifyError("identity point in the transcript")
    return pt[0].to_bytes(32, "big") + pt[1].to_bytes(32, "big")


def commitment_bytes(rows):
    return b"poly_commitment_begin" + b"".join(point_bytes(r) for r in rows) + b"poly_commitment_end"


# ---- T256 in Jacobian coordinates (None = identity for affine points, Z == 0 for Jacobian) ----------------------------------------------------------
def on_curve(pt):
    return pt is None or (pt[1] * pt[1] - (pt[0] * pt[0] * pt[0] - 3 * pt[0] + CURVE_B)) % P == 0


def jdbl(p1):
    X, Y, Z = p1
    if Z == 0 or Y == 0:
        return (1, 1, 0)
    zz = Z * Z % P
    m = 3 * (X - zz) * (X + zz) % P  # a = -3
    yy = Y * Y % P
    s = 4 * X * yy % P
    x3 = (m * m - 2 * s) % P
    return (x3, (m * (s - x3) - 8 * yy * yy) % P, 2 * Y * Z % P)


def jadd(p1, p2):
    if p1[2] == 0:
        return p2
    if p2[2] == 0:
        return p1
    z1z1, z2z2 = p1[2] * p1[2] % P, p2[2] * p2[2] % P
    u1, u2 = p1[0] * z2z2 % P, p2[0] * z1z1 % P
    s1, s2 = p1[1] * p2[2] * z2z2 % P, p2[1] * p1[2] * z1z1 % P
    if u1 == u2:
        return jdbl(p1) if s1 == s2 else (1, 1, 0)
    h, r = (u2 - u1) % P, (s2 - s1) % P
    hh = h * h % P
    hhh = h * hh % P
    v = u1 * hh % P
    x3 = (r * r - hhh - 2 * v) % P
    return (x3, (r * (v - x3) - s1 * hhh) % P, p1[2] * p2[2] * h % P)


def to_jac(pt):
    return (1, 1, 0) if pt is None else (pt[0], pt[1], 1)


def to_aff(p1):
    if p1[2] == 0:
        return None
    zi = pow(p1[2], -1, P)
    return (p1[0] * zi * zi % P, p1[1] * zi * zi * zi % P)


def smul(pt, k):
    acc, base = (1, 1, 0), to_jac(pt)
    k %= Q
    while k:
        if k & 1:
            acc = jadd(acc, base)
        base = jdbl(base)
        k >>= 1
    return acc


def msm(scalars, points, c=8):
    """sum_i scalars[i] * points[i] by buckets of c-bit windows (the result is what matters, not the schedule)"""
    total = (1, 1, 0)
    jp = [to_jac(p_) for p_ in points]
    for w in reversed(range((256 + c - 1) // c)):
        for _ in range(c):
            total = jdbl(total)
        buckets = [None] * (1 << c)
        for s, p_ in zip(scalars, jp):
            d = (s >> (w * c)) & ((1 << c) - 1)
            if d:
                buckets[d] = p_ if buckets[d] is None else jadd(buckets[d], p_)
        run, acc = (1, 1, 0), (1, 1, 0)
        for d in range((1 << c) - 1, 0, -1):
            if buckets[d] is not None:
                run = jadd(run, buckets[d])
            acc = jadd(acc, run)
        total = jadd(total, acc)
    return total


def jeq(p1, p2):
    return to_aff(p1) == to_aff(p2)


# ---- polynomials --------------------------------------------------------------------------------------------------------------------------------
def eq_evals(r):  # EqPolynomial::evals_from_points: r[0] is the most significant variable
    out = [1]
    for ri in r:
        out = [v for e in out for v in (e * (1 - ri) % Q, e * ri % Q)]
    return out


def eq_evaluate(r, x):  # EqPolynomial::evaluate
    acc = 1
    for a, b in zip(r, x):
        acc = acc * (a * b + (1 - a) * (1 - b)) % Q
    return acc


def sumcheck_verify(tr, claim, num_rounds, degree, polys):
    """SumcheckProof::verify: polys[i] = the compressed coefficients (all but the linear term); -> (final claim, challenges)"""
    if len(polys) != num_rounds:
        raise VerifyError("sum-check: wrong number of rounds")
    e, rs = claim, []
    for cp in polys:
        if len(cp) != degree:  # degree + 1 coefficients, one omitted
            raise VerifyError("sum-check: degree bound")
        lin = (e - 2 * cp[0] - sum(cp[1:])) % Q  # decompress (univariate.rs:166-179)
        coeffs = [cp[0], lin] + list(cp[1:])
        tr.absorb(b"p", b"".join(c.to_bytes(32, "little") for c in cp))  # UniPoly::to_transcript_bytes: to_repr of the compressed coefficients
        r_i = tr.squeeze(b"c")
        rs.append(r_i)
        e, power = coeffs[0], r_i
        for c in coeffs[1:]:
            e = (e + power * c) % Q
            power = power * r_i % Q
    return e, rs


def sparse_poly_evaluate(num_vars, Z, r):  # SparsePolynomial::evaluate (multilinear.rs:190-207)
    assert len(r) == num_vars
    nvz = max(0, (len(Z) - 1).bit_length())  # Z.len().next_power_of_two().log_2()
    chis = eq_evals(r[num_vars - 1 - nvz:])
    part = sum(z * c for z, c in zip(Z, chis)) % Q
    common = 1
    for i in range(num_vars - 1 - nvz):
        common = common * (1 - r[i]) % Q
    return common * part % Q


def matrix_evals(mats, num_cons, T_x, T_y):
    """sum over the entries of A[i][j] * T_x[i] * T_y[j], per matrix (data, column indices, row pointers)"""
    evals = []
    for data, cols, ptr in mats:
        acc = 0
        for row in range(num_cons):
            lo, hi = ptr[row], ptr[row + 1]
            if hi > lo:
                acc += T_x[row] * (sum(int(data[k]) * T_y[int(cols[k])] for k in range(lo, hi)) % Q)
        evals.append(acc % Q)
    return evals


def hyrax_verify(tr, ck_pts, h_pt, cks_pt, hs_pt, comm, point, comm_eval, ipa):
    """PCS::verify (hyrax_pc.rs:480-531) + InnerProductArgumentLinear::verify (ipa.rs:173-221). comm: the row commitments (affine); comm_eval: the
    commitment to the evaluation (Jacobian); ipa: dict(delta, beta, z_vec, z_delta, z_beta); cks_pt / hs_pt: the evaluation key's first generator and h."""
    tr.absorb(b"poly_com", commitment_bytes(comm))
    n, num_cols = 1 << len(point), len(ck_pts)
    num_rows = -(-n // num_cols)
    nvr = num_rows.bit_length() - 1
    if nvr == 0:
        R, comm_LZ = eq_evals(point), to_jac(comm[0])
    else:
        L, R = eq_evals(point[:nvr]), eq_evals(point[nvr:])
        if len(comm) < len(L):
            raise VerifyError("commitment: fewer rows than the point addresses")
        comm_LZ = msm(L, comm[:len(L)])
    tr.dom_sep(b"inner product argument (linear)")
    tr.absorb(b"U", point_bytes(to_aff(comm_LZ)) + point_bytes(to_aff(comm_eval)))
    tr.absorb(b"delta", point_bytes(ipa["delta"]))
    tr.absorb(b"beta", point_bytes(ipa["beta"]))
    rr = tr.squeeze(b"r")
    z = ipa["z_vec"]
    if len(z) != len(R) or num_cols < len(z):
        raise VerifyError("inner product argument: length of z_vec")
    lhs = jadd(smul(to_aff(comm_LZ), rr), to_jac(ipa["delta"]))
    if not jeq(lhs, jadd(msm(z, ck_pts[:len(z)]), smul(h_pt, ipa["z_delta"]))):
        raise VerifyError("inner product argument: first equation")
    ip = sum(a * b for a, b in zip(z, R)) % Q
    if not jeq(jadd(smul(to_aff(comm_eval), rr), to_jac(ipa["beta"])), jadd(smul(cks_pt, ip), smul(hs_pt, ipa["z_beta"]))):
        raise VerifyError("inner product argument: second equation")


# ---- the proof ------------------------------------------------------------------------------------------------------------------------------------
def _pt(words):
    x, y = pywire._canon(words[:4], P), pywire._canon(words[4:8], P)
    return None if x == 0 and y == 0 else (x, y)


def _sc(words):
    return pywire._canon(words, Q)


def parse_proof(words, rows_shared, rows_pre, rows_rest, num_public, num_challenges, lx, ly, nz):
    c = pywire._Cursor(words)
    pr = {}
    pr["comm_shared"] = [_pt(w) for w in c.take(rows_shared, 8)]
    pr["comm_pre"] = [_pt(w) for w in c.take(rows_pre, 8)]
    pr["comm_rest"] = [_pt(w) for w in c.take(rows_rest, 8)]
    pr["public"] = [_sc(w) for w in c.take(num_public, 4)]
    pr["challenges"] = [_sc(w) for w in c.take(num_challenges, 4)]
    pr["outer"] = [[_sc(w) for w in rnd] for rnd in c.take(lx, 12).reshape(lx, 3, 4)]
    pr["claims_outer"] = [_sc(w) for w in c.take(3, 4)]
    pr["inner"] = [[_sc(w) for w in rnd] for rnd in c.take(ly, 8).reshape(ly, 2, 4)]
    pr["eval_W"] = _sc(c.take(1, 4)[0])
    pr["blind_eval_W"] = _sc(c.take(1, 4)[0])
    pr["delta"], pr["beta"] = _pt(c.take(1, 8)[0]), _pt(c.take(1, 8)[0])
    pr["z_vec"] = [_sc(w) for w in c.take(nz, 4)]
    pr["z_delta"], pr["z_beta"] = _sc(c.take(1, 4)[0]), _sc(c.take(1, 4)[0])
    c.done()
    return pr


class _ByteReader:
    """bincode DefaultOptions + little-endian + fixint (src/digest.rs:33-41), reading side: u64 lengths, 32-byte little-endian field elements (canonical
    or rejected), projective points as x | y | z (any representative; z = 0 the identity)"""

    def __init__(self, data: bytes):
        self.d, self.o = memoryview(data), 0

    def take(self, n):
        if self.o + n > len(self.d):
            raise VerifyError("proof bytes: truncated")
        out = bytes(self.d[self.o:self.o + n])
        self.o += n
        return out

    def u8(self):
        return self.take(1)[0]

    def u64(self):
        return int.from_bytes(self.take(8), "little")

    def fe(self, modulus):
        v = int.from_bytes(self.take(32), "little")
        if v >= modulus:
            raise VerifyError("proof bytes: non-canonical field element")
        return v

    def scalar(self):
        return self.fe(Q)

    def scalars(self):
        n = self.u64()
        if n > (len(self.d) - self.o) // 32:
            raise VerifyError("proof bytes: vector length exceeds the input")
        return [self.scalar() for _ in range(n)]

    def point(self):
        x, y, z = self.fe(P), self.fe(P), self.fe(P)
        if z == 0:
            return None
        zi = pow(z, -1, P)
        return (x * zi * zi % P, y * zi * zi * zi % P)

    def commitment(self):
        n = self.u64()
        if n > (len(self.d) - self.o) // 96:
            raise VerifyError("proof bytes: vector length exceeds the input")
        return [self.point() for _ in range(n)]

    def option_commitment(self):
        tag = self.u8()
        if tag > 1:
            raise VerifyError("proof bytes: Option tag")
        return self.commitment() if tag else []

    def sumcheck(self):
        n = self.u64()
        if n > (len(self.d) - self.o) // 8:
            raise VerifyError("proof bytes: vector length exceeds the input")
        return [self.scalars() for _ in range(n)]


def parse_proof_bytes(data: bytes):
    """SpartanSNARK { U: SplitR1CSInstance { comm_W_shared, comm_W_precommitted, comm_W_rest, public_values, challenges }, sc_proof_outer, claims_outer,
    sc_proof_inner, eval_W, blind_eval_W, eval_arg: HyraxEvaluationArgument { ipa { delta, beta, z_vec, z_delta, z_beta } } } (src/spartan.rs:125-137,
    src/r1cs/mod.rs:1423-1437) from its bincode bytes"""
    rd = _ByteReader(data)
    pr = {"comm_shared": rd.option_commitment(), "comm_pre": rd.option_commitment(), "comm_rest": rd.commitment(), "public": rd.scalars(), "challenges": rd.scalars(),
          "outer": rd.sumcheck()}
    pr["claims_outer"] = [rd.scalar() for _ in range(3)]
    pr["inner"] = rd.sumcheck()
    pr["eval_W"] = rd.scalar()
    blind = rd.scalars()  # HyraxBlind { blind: Vec<Scalar> }, one row
    if len(blind) != 1:
        raise VerifyError("proof bytes: blind_eval_W must have one entry")
    pr["blind_eval_W"] = blind[0]
    pr["delta"], pr["beta"] = rd.point(), rd.point()
    pr["z_vec"] = rd.scalars()
    pr["z_delta"], pr["z_beta"] = rd.scalar(), rd.scalar()
    if rd.o != len(data):
        raise VerifyError("proof bytes: trailing bytes")
    return pr


def verify_bytes(inst, ck, h, ck_s, h_s, data: bytes, vk_digest=None):
    """SpartanSNARK::verify of a proof given as the reference's bincode bytes"""
    return verify(inst, ck, h, ck_s, h_s, None, None, vk_digest=vk_digest, parsed=parse_proof_bytes(data))


def verify(inst, ck, h, ck_s, h_s, words, layout, vk_digest=None, parsed=None):
    """SpartanSNARK::verify. `vk_digest`: the 32 digest bytes, or None to recompute them (tests/pywire.py spartan_vk_digest)."""
    dims, mats, _ = pywire.pad_shape(inst)
    ck_pts = [_pt(w) for w in np.asarray(ck, dtype=np.uint64).reshape(-1, 8)]
    h_pt, cks_pt, hs_pt = _pt(np.asarray(h, dtype=np.uint64)), _pt(np.asarray(ck_s, dtype=np.uint64).reshape(-1)[:8]), _pt(np.asarray(h_s, dtype=np.uint64))
    pr = parsed if parsed is not None else parse_proof(words, **layout)
    for name in ("comm_shared", "comm_pre", "comm_rest"):
        if not all(on_curve(p_) for p_ in pr[name]):
            raise VerifyError(f"{name}: not on the curve")
    if not (on_curve(pr["delta"]) and on_curve(pr["beta"]) and all(on_curve(p_) for p_ in ck_pts[:4]) and on_curve(h_pt)):
        raise VerifyError("point not on the curve")
    tr = Transcript(b"SpartanSNARK")
    tr.absorb(b"vk", bytes(vk_digest) if vk_digest is not None else pywire.spartan_vk_digest(inst, ck, h, ck_s, h_s))
    tr.absorb(b"public_values", b"".join(scalar_bytes(v) for v in pr["public"]))
    # SplitR1CSInstance::validate (r1cs/mod.rs:1490-1533)
    rows = lambda n: -(-n // pywire.WIDTH)
    if dims["num_shared"] > 0:
        if len(pr["comm_shared"]) != rows(dims["num_shared"]):
            raise VerifyError("comm_W_shared: wrong number of rows")
        tr.absorb(b"comm_W_shared", commitment_bytes(pr["comm_shared"]))
    if dims["num_precommitted"] > 0:
        if len(pr["comm_pre"]) != rows(dims["num_precommitted"]):
            raise VerifyError("comm_W_precommitted: wrong number of rows")
        tr.absorb(b"comm_W_precommitted", commitment_bytes(pr["comm_pre"]))
    if [tr.squeeze(b"challenge") for _ in range(dims["num_challenges"])] != pr["challenges"]:
        raise VerifyError("Challenges do not match")
    if len(pr["comm_rest"]) != rows(dims["num_rest"]):
        raise VerifyError("comm_W_rest: wrong number of rows")
    tr.absorb(b"comm_W_rest", commitment_bytes(pr["comm_rest"]))
    comm_W = pr["comm_shared"] + pr["comm_pre"] + pr["comm_rest"]  # to_regular_instance (:1535-1550)
    X = pr["public"] + pr["challenges"]
    num_vars = dims["num_shared"] + dims["num_precommitted"] + dims["num_rest"]
    lx, ly = dims["num_cons"].bit_length() - 1, num_vars.bit_length()
    # outer sum-check (spartan.rs:495-523)
    tau = [tr.squeeze(b"t") for _ in range(lx)]
    claim_outer_final, r_x = sumcheck_verify(tr, 0, lx, 3, pr["outer"])
    cA, cB, cC = pr["claims_outer"]
    if claim_outer_final != eq_evaluate(tau, r_x) * (cA * cB - cC) % Q:
        raise VerifyError("outer sum-check: final claim")
    tr.absorb(b"claims_outer", b"".join(scalar_bytes(v) for v in (cA, cB, cC)))
    # inner sum-check (:525-562)
    r = tr.squeeze(b"r")
    claim_inner_final, r_y = sumcheck_verify(tr, (cA + r * cB + r * r * cC) % Q, ly, 2, pr["inner"])
    eval_X = sparse_poly_evaluate(ly - 1, [1] + X, r_y[1:])
    eval_Z = ((1 - r_y[0]) * pr["eval_W"] + r_y[0] * eval_X) % Q
    evals = matrix_evals(mats, dims["num_cons"], eq_evals(r_x), eq_evals(r_y))
    if claim_inner_final != (evals[0] + r * evals[1] + r * r * evals[2]) * eval_Z % Q:
        raise VerifyError("inner sum-check: final claim")
    # PCS::verify (hyrax_pc.rs:480-531) of comm_W at r_y[1..] against commit(ck_s, [eval_W], blind_eval_W)
    comm_eval = jadd(smul(cks_pt, pr["eval_W"]), smul(hs_pt, pr["blind_eval_W"]))
    hyrax_verify(tr, ck_pts, h_pt, cks_pt, hs_pt, comm_W, r_y[1:], comm_eval, pr)
    return pr["public"]
