"""A third, Python-integer restatement of the wire formats, written from the reference's struct definitions alone
and independent of both C++ implementations (oracle/wire*.hpp and spartan2_amd/csrc/wire.hpp): bincode `DefaultOptions` + little-endian +
fixint (src/digest.rs:33-41), SHA-256 from hashlib.

Field element = 32 bytes little-endian of the canonical value (`to_repr`), affine point = x | y, projective point = x | y | z written
normalised (z = 1; identity = three zeros) — the one third-party assumption, stated in oracle/wire.hpp.
Inputs are the flat word layouts of DESIGN.md section 4 (Montgomery limbs), converted with Python integers."""
import hashlib
import struct

import numpy as np

P_SCALAR = 0xFFFFFFFF00000001000000000000000000000000FFFFFFFFFFFFFFFFFFFFFFFF  # src/provider/pt256.rs:55
P_BASE = 0xFFFFFFFF0000000100000000000000017E72B42B30E7317793135661B1C4B117  # src/provider/pt256.rs:56
_RINV = {p: pow(1 << 256, -1, p) for p in (P_SCALAR, P_BASE)}
WIDTH = 2048  # DEFAULT_COMMITMENT_WIDTH


def _canon(limbs, p):
    v = 0
    for i in range(4):
        v |= int(limbs[i]) << (64 * i)
    return v * _RINV[p] % p


class Writer:
    def __init__(self):
        self.parts = []

    def u8(self, v):
        self.parts.append(bytes([v]))

    def u64(self, v):
        self.parts.append(struct.pack("<Q", int(v)))

    def scalar(self, limbs):  # E::Scalar
        self.parts.append(_canon(limbs, P_SCALAR).to_bytes(32, "little"))

    def coord(self, limbs):
        self.parts.append(_canon(limbs, P_BASE).to_bytes(32, "little"))

    def scalars(self, arr):  # Vec<E::Scalar>
        arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
        self.u64(arr.shape[0])
        for row in arr:
            self.scalar(row)

    def usizes(self, arr):
        self.u64(len(arr))
        for v in arr:
            self.u64(v)

    def affine(self, pt):  # AffineGroupElement: {x, y}
        self.coord(pt[:4])
        self.coord(pt[4:8])

    def point(self, pt):  # E::GE from the flat layout's affine (x, y); (0, 0) = identity
        if not np.asarray(pt).any():
            self.parts.append(bytes(96))
        else:
            self.coord(pt[:4])
            self.coord(pt[4:8])
            self.parts.append((1).to_bytes(32, "little"))

    def commitment(self, rows):  # HyraxCommitment { comm: Vec<E::GE> }
        rows = np.asarray(rows, dtype=np.uint64).reshape(-1, 8)
        self.u64(rows.shape[0])
        for r in rows:
            self.point(r)

    def option_commitment(self, rows):
        rows = np.asarray(rows, dtype=np.uint64).reshape(-1, 8)
        self.u8(1 if rows.shape[0] else 0)
        if rows.shape[0]:
            self.commitment(rows)

    def hyrax_key(self, ck, h):  # HyraxCommitmentKey / HyraxVerifierKey { num_cols, ck: Vec<Affine>, h: GE }
        ck = np.asarray(ck, dtype=np.uint64).reshape(-1, 8)
        self.u64(ck.shape[0])
        self.u64(ck.shape[0])
        for b in ck:
            self.affine(b)
        self.point(np.asarray(h, dtype=np.uint64).reshape(8))

    def sumcheck(self, polys):  # SumcheckProof { compressed_polys: Vec<CompressedUniPoly { coeffs_except_linear_term: Vec<F> }> }
        polys = np.asarray(polys, dtype=np.uint64)
        self.u64(polys.shape[0])
        for pl in polys:
            self.scalars(pl)

    def bytes(self):
        return b"".join(self.parts)


class _Cursor:
    def __init__(self, words):
        self.w = np.ascontiguousarray(words, dtype=np.uint64)
        self.o = 0

    def take(self, n, width):
        out = self.w[self.o:self.o + n * width].reshape(n, width)
        assert out.shape == (n, width), "flat proof too short"
        self.o += n * width
        return out

    def done(self):
        assert self.o == len(self.w), "flat proof has trailing words"


def spartan_proof_bytes(words, rows_shared, rows_pre, rows_rest, num_public, num_challenges, lx, ly, nz):
    """SpartanSNARK { U, sc_proof_outer, claims_outer, sc_proof_inner, eval_W, blind_eval_W, eval_arg } (src/spartan.rs:125-137)"""
    c = _Cursor(words)
    w = Writer()
    w.option_commitment(c.take(rows_shared, 8))  # U.comm_W_shared
    w.option_commitment(c.take(rows_pre, 8))  # U.comm_W_precommitted
    w.commitment(c.take(rows_rest, 8))  # U.comm_W_rest
    w.scalars(c.take(num_public, 4))
    w.scalars(c.take(num_challenges, 4))
    w.sumcheck(c.take(lx, 12).reshape(lx, 3, 4))
    for f in c.take(3, 4):  # (Scalar, Scalar, Scalar)
        w.scalar(f)
    w.sumcheck(c.take(ly, 8).reshape(ly, 2, 4))
    w.scalar(c.take(1, 4)[0])  # eval_W
    w.scalars(c.take(1, 4))  # blind_eval_W: HyraxBlind { blind: Vec } with one row
    w.point(c.take(1, 8)[0])  # ipa.delta
    w.point(c.take(1, 8)[0])  # ipa.beta
    w.scalars(c.take(nz, 4))  # ipa.z_vec
    w.scalar(c.take(1, 4)[0])
    w.scalar(c.take(1, 4)[0])
    c.done()
    return w.bytes()


def pad_shape(inst):
    """SplitR1CSShape::new (src/r1cs/mod.rs:810-911) on integer CSR triples: -> (dims dict, [(data ints, col indices, indptr)] * 3, num_cols)"""
    pad = lambda n: -(-n // WIDTH) * WIDTH
    np2 = lambda n: 1 << max(0, (n - 1).bit_length())
    sp, pp, rp = pad(inst.num_shared), pad(inst.num_precommitted), pad(inst.num_rest)
    nvp = sp + pp + rp
    if nvp < inst.num_public + inst.num_challenges + 1:
        rp = max(inst.num_public + inst.num_challenges + 1, nvp) - (sp + pp)
    nvp = sp + pp + rp
    if np2(nvp) != nvp:
        rp = np2(nvp) - (sp + pp)
    nvp = sp + pp + rp
    num_vars = inst.num_shared + inst.num_precommitted + inst.num_rest
    ncp = np2(inst.num_cons)
    dims = dict(num_cons=ncp, num_cons_unpadded=inst.num_cons, num_shared_unpadded=inst.num_shared, num_precommitted_unpadded=inst.num_precommitted,
                num_rest_unpadded=inst.num_rest, num_shared=sp, num_precommitted=pp, num_rest=rp, num_public=inst.num_public, num_challenges=inst.num_challenges)
    mats = []
    for d, i, p_ in inst.csr:
        cols = np.asarray(i, dtype=np.int64).copy()
        c0 = cols.copy()
        in_pre = (c0 >= inst.num_shared) & (c0 < inst.num_shared + inst.num_precommitted)
        in_rest = (c0 >= inst.num_shared + inst.num_precommitted) & (c0 < num_vars)
        in_io = c0 >= num_vars
        cols[in_pre] += sp - inst.num_shared
        cols[in_rest] += sp + pp - inst.num_shared - inst.num_precommitted
        cols[in_io] += nvp - num_vars
        ptr = list(np.asarray(p_, dtype=np.int64)[:inst.num_cons + 1])
        ptr += [ptr[-1]] * (ncp + 1 - len(ptr))
        mats.append((np.asarray(d, dtype=np.int64), cols, ptr))
    return dims, mats, nvp + 1 + inst.num_public + inst.num_challenges


def equalize(shape_a, shape_b):
    """SplitR1CSShape::equalize (src/r1cs/mod.rs:913-971) on two results of pad_shape: -> the two (dims, mats, num_cols) after it"""
    nv = lambda d: d["num_shared"] + d["num_precommitted"] + d["num_rest"]
    cons, nvars = max(shape_a[0]["num_cons"], shape_b[0]["num_cons"]), max(nv(shape_a[0]), nv(shape_b[0]))
    out = []
    for dims, mats, num_cols in (shape_a, shape_b):
        dims, old_vars, old_cons = dict(dims), nv(dims), dims["num_cons"]
        dims["num_cons"] = cons
        if old_vars != nvars:
            dims["num_rest"] = nvars - (dims["num_shared"] + dims["num_precommitted"])
        grown = []
        for data, cols, ptr in mats:
            cols = np.asarray(cols, dtype=np.int64).copy()
            cols[cols >= old_vars] += nvars - old_vars  # 1 | public | challenges
            ptr = list(ptr) + [ptr[-1] if len(ptr) else 0] * (cons - old_cons)
            grown.append((data, cols, ptr))
        out.append((dims, grown, num_cols + nvars - old_vars))
    return out


_DIM_ORDER = ("num_cons", "num_cons_unpadded", "num_shared_unpadded", "num_precommitted_unpadded", "num_rest_unpadded", "num_shared", "num_precommitted",
              "num_rest", "num_public", "num_challenges")  # field order of SplitR1CSShape (src/r1cs/mod.rs:743-755) = write order of write_bytes (:777-786)


def _coeff_bytes(data):
    """to_repr of small integer coefficients, vectorised: 32 bytes LE of (v mod p)"""
    out = np.zeros((len(data), 32), dtype=np.uint8)
    lut = {}
    for k, v in enumerate(data):
        v = int(v)
        b = lut.get(v)
        if b is None:
            b = lut[v] = np.frombuffer((v % P_SCALAR).to_bytes(32, "little"), dtype=np.uint8)
        out[k] = b
    return out.tobytes()


def shape_write_bytes(h, dims, mats, num_cols):
    """SplitR1CSShape::write_bytes (src/r1cs/mod.rs:775-794) + SparseMatrix::write_digest_bytes (src/r1cs/sparse.rs:398-417) into hasher h"""
    h.update(b"".join(struct.pack("<Q", dims[k]) for k in _DIM_ORDER))
    for data, cols, ptr in mats:
        h.update(struct.pack("<QQQQ", len(data), len(cols), len(ptr), num_cols))
        h.update(_coeff_bytes(data))
        h.update(np.asarray(cols, dtype="<u8").tobytes())
        h.update(np.asarray(ptr, dtype="<u8").tobytes())


def spartan_vk_digest(inst, ck, h, ck_s, h_s):
    """SHA-256 over SpartanVerifierKey::write_bytes (src/spartan.rs:73-90): bincode(vk_ee) || bincode(ck_s) || S.write_bytes()"""
    hs = hashlib.sha256()
    w = Writer()
    w.hyrax_key(ck, h)
    w.hyrax_key(ck_s, h_s)
    hs.update(w.bytes())
    shape_write_bytes(hs, *pad_shape(inst))
    return hs.digest()


def nn_proof_bytes(words, rows_sh, rows_pre, rows_rest, n_steps, npub_step, npub_core, nz, vc_rows_per_round, vc_public, vc_chals_per_round, vc_cons_rows, vc_io,
                   lx, ly, width, rows_pre_core=None, rows_rest_core=None):
    """NeutronNovaZkSNARK (src/neutronnova_zk.rs:1373-1385) from the flat layout of oracle NNProof::serialize. rows_*_core: the core circuit's own split of
    its rows into precommitted | rest (default: the step's)"""
    c = _Cursor(words)
    w = Writer()
    w.option_commitment(c.take(rows_sh, 8))

    def inst(npub, r_pre, r_rest):  # SplitR1CSInstance, comm_W_shared = None (:2069-2078)
        w.u8(0)
        w.option_commitment(c.take(r_pre, 8))
        w.commitment(c.take(r_rest, 8))
        w.scalars(c.take(npub, 4))
        w.scalars(np.zeros((0, 4), dtype=np.uint64))

    w.u64(n_steps)
    for _ in range(n_steps):
        inst(npub_step, rows_pre, rows_rest)
    inst(npub_core, rows_pre if rows_pre_core is None else rows_pre_core, rows_rest if rows_rest_core is None else rows_rest_core)
    w.point(c.take(1, 8)[0])
    w.point(c.take(1, 8)[0])
    w.scalars(c.take(nz, 4))
    w.scalar(c.take(1, 4)[0])
    w.scalar(c.take(1, 4)[0])
    w.u64(len(vc_rows_per_round))  # U_verifier.comm_w_per_round
    for r in vc_rows_per_round:
        w.commitment(c.take(r, 8))
    w.scalars(c.take(vc_public, 4))
    w.u64(len(vc_chals_per_round))
    for k in vc_chals_per_round:
        w.scalars(c.take(k, 4))
    w.commitment(c.take(vc_cons_rows, 8))  # nifs.comm_T
    w.commitment(c.take(sum(vc_rows_per_round), 8))  # random_U.comm_W
    w.commitment(c.take(vc_cons_rows, 8))  # random_U.comm_E
    u = c.take(1, 4)[0]
    w.scalars(c.take(vc_io, 4))  # X
    w.scalar(u)  # u (declared after X, src/r1cs/mod.rs:213-218)
    w.sumcheck(c.take(lx, 12).reshape(lx, 3, 4))
    for f in c.take(3, 4):
        w.scalar(f)
    w.sumcheck(c.take(ly, 8).reshape(ly, 2, 4))
    w.scalars(c.take(width, 4))
    w.scalar(c.take(1, 4)[0])
    w.scalars(c.take(width, 4))
    w.scalar(c.take(1, 4)[0])
    c.done()
    return w.bytes()
