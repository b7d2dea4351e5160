"""ctypes view of libspartan_host.so — the C++ host side above the C ABI (spartan2_amd/host/spartan_snark.cpp), which mirrors
SpartanSNARK::{setup, prep_prove, prove} (src/spartan.rs:146-466) and calls only include/spartan_hip.h."""
import ctypes
import os

import numpy as np

from . import hip

LIB_PATH = os.path.join(hip.LIB_DIR, "libspartan_host.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        hip.lib()  # libspartan_hip.so first (rpath $ORIGIN resolves it too)
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
        L = ctypes.CDLL(LIB_PATH)
        L.ss_last_error.restype = ctypes.c_char_p
        L.ss_proof_words.restype = ctypes.c_size_t
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise hip.SpartanHipError(f"rc={rc}: {lib().ss_last_error().decode()}")


def _inst_args(inst):
    args = [ctypes.c_size_t(inst.num_cons), ctypes.c_size_t(inst.num_shared), ctypes.c_size_t(inst.num_precommitted), ctypes.c_size_t(inst.num_rest),
            ctypes.c_size_t(inst.num_public), ctypes.c_size_t(inst.num_challenges)]
    keep = []
    for d, i, p_ in inst.csr:
        d = np.ascontiguousarray(d, dtype=np.int64)
        i = np.ascontiguousarray(i, dtype=np.uint32)
        p_ = np.ascontiguousarray(p_, dtype=np.uint64)
        keep += [d, i, p_]
        args += [d.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), i.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), hip.p64(p_)]
    return args, keep


DIM_NAMES = ("num_cons", "num_cons_unpadded", "num_shared", "num_precommitted", "num_rest", "num_shared_unpadded", "num_precommitted_unpadded",
             "num_rest_unpadded", "num_public", "num_challenges")


def _padded_handle(inst):
    args, keep = _inst_args(inst)
    h = ctypes.c_void_p()
    _check(lib().ss_pad_shape(*args, ctypes.byref(h)))
    return h


def pad_shape(inst):
    """SplitR1CSShape::new (src/r1cs/mod.rs:810-911): padded CSR matrices with field coefficients + dims dict."""
    return _padded_export(_padded_handle(inst))


def pad_shapes_equalized(inst_a, inst_b):
    """SplitR1CSShape::new on both, then SplitR1CSShape::equalize (src/r1cs/mod.rs:913-971): -> ((mats, dims), (mats, dims))"""
    ha, hb = _padded_handle(inst_a), _padded_handle(inst_b)
    lib().ss_padded_equalize(ha, hb)
    return _padded_export(ha), _padded_export(hb)


def _padded_export(h):
    d = (ctypes.c_uint64 * 10)()
    lib().ss_padded_dims(h, d)
    dims = {k: int(v) for k, v in zip(DIM_NAMES, d)}
    mats = []
    for which in range(3):
        data = hip.c_u64p()
        idx = ctypes.POINTER(ctypes.c_uint32)()
        ptr = hip.c_u64p()
        nnz = ctypes.c_uint64()
        lib().ss_padded_csr(h, which, ctypes.byref(data), ctypes.byref(idx), ctypes.byref(ptr), ctypes.byref(nnz))
        n = int(nnz.value)
        mats.append((np.ctypeslib.as_array(data, (max(n, 1) * 4,))[: n * 4].reshape(n, 4).copy(), np.ctypeslib.as_array(idx, (max(n, 1),))[:n].copy(),
                     np.ctypeslib.as_array(ptr, (dims["num_cons"] + 1,)).copy()))
    lib().ss_padded_free(h)
    return mats, dims


def from_label(label: bytes, n: int):
    out = np.zeros((n, 8), dtype=np.uint64)
    _check(lib().ss_from_label(label, ctypes.c_size_t(n), hip.p64(out)))
    return out


REST_HOOK = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, hip.c_u64p, ctypes.c_size_t, hip.c_u64p)
PHASES = ("witness_commit", "matrix_vector_multiply", "outer_sumcheck", "prepare_poly_ABC", "inner_sumcheck", "pcs_prove", "total")


class SpartanSNARK:
    """setup -> prep_prove -> prove, as benches/sha256_spartan.rs:171-243 drives them."""

    def __init__(self, ctx: hip.Context, inst):
        self.ctx = ctx
        self.inst = inst
        args, keep = _inst_args(inst)
        self.pk = ctypes.c_void_p()
        _check(lib().ss_setup(ctx.h, *args, ctypes.byref(self.pk)))
        d = (ctypes.c_uint64 * 10)()
        dig = np.zeros(32, dtype=np.uint8)
        lib().ss_pk_info(self.pk, d, hip.p8(dig))
        self.dims = {k: int(v) for k, v in zip(DIM_NAMES, d)}
        self.vk_digest = dig
        self.ps = None
        si = (ctypes.c_uint64 * 8)()
        _check(lib().ss_pk_shape_info(self.pk, si))
        self.shape_info = {"nnz": [int(si[i]) for i in range(3)], "nnz_filtered": [int(si[3 + i]) for i in range(3)], "long_columns": int(si[6]),
                           "short_columns": int(si[7])}

    def prep_prove(self, tape: np.ndarray, is_small=True):
        used = ctypes.c_size_t(0)
        w = np.ascontiguousarray(self.inst.witness, dtype=np.uint64)
        ps = ctypes.c_void_p()
        _check(lib().ss_prep_prove(self.pk, hip.p64(w), ctypes.c_size_t(len(w)), int(is_small), hip.p8(tape), ctypes.c_size_t(tape.shape[0]), ctypes.byref(used),
                                   ctypes.byref(ps)))
        if self.ps:
            lib().ss_prep_free(self.ps)
        self.ps = ps
        return used.value

    def set_flags(self, prefix_cache=None, lz_direct=None, reference_order=None):
        """Driver options of the current prep state (spartan_snark.cpp FLAG_*): prefix_cache = keep the transcript prefix's sponge state across
        proves instead of re-hashing it in every prove (the reference re-hashes); lz_direct = the opening in the reference's own order."""
        lib().ss_prep_get_flags.restype = ctypes.c_uint
        f = lib().ss_prep_get_flags(self.ps)
        if prefix_cache is not None:
            f = (f | 1) if prefix_cache else (f & ~1)
        if lz_direct is not None:
            f = (f | 2) if lz_direct else (f & ~2)
        if reference_order is not None:  # one thread, the reference's statement order, only ABI calls (spartan_snark.cpp prove_reference_order)
            f = (f | 4) if reference_order else (f & ~4)
        lib().ss_prep_set_flags(self.ps, ctypes.c_uint(f))

    def prep_export(self):
        d = self.dims
        rows = ((d["num_shared"] + 2047) // 2048 if d["num_shared_unpadded"] else 0) + ((d["num_precommitted"] + 2047) // 2048 if d["num_precommitted_unpadded"] else 0)
        N = self.dims["num_cons"]
        comm = np.zeros((rows, 8), dtype=np.uint64)
        caz = np.zeros((N, 4), dtype=np.uint64)
        cbz = np.zeros_like(caz)
        ccz = np.zeros_like(caz)
        _check(lib().ss_prep_export(self.pk, self.ps, hip.p64(comm) if rows else None, hip.p64(caz), hip.p64(cbz), hip.p64(ccz)))
        return comm, caz, cbz, ccz

    def prove(self, tape: np.ndarray, synthesize=None):
        """Returns (proof words in the canonical layout, tape blocks used, {phase: ms}). Circuits with verifier challenges pass
        synthesize(challenges (k, 4)) -> rest witness (num_rest_unpadded, 4) Montgomery limbs (circuit.synthesize, bellpepper/r1cs.rs:443-461)."""
        n = lib().ss_proof_words(self.pk)
        words = np.zeros(n, dtype=np.uint64)
        used = ctypes.c_size_t(0)
        ms = (ctypes.c_double * 7)()
        pub = np.ascontiguousarray(self.inst.publics, dtype=np.uint64)
        cb = None
        if synthesize is not None:
            nrest = self.dims["num_rest_unpadded"]

            def raw(_user, ch_ptr, nch, out_ptr):
                try:
                    ch = np.ctypeslib.as_array(ch_ptr, shape=(4 * nch,)).reshape(nch, 4).copy()
                    rest = np.ascontiguousarray(synthesize(ch), dtype=np.uint64).reshape(nrest, 4)
                    np.ctypeslib.as_array(out_ptr, shape=(4 * max(nrest, 1),))[: 4 * nrest] = rest.reshape(-1)
                    return 0
                except Exception:  # noqa: BLE001
                    import traceback

                    traceback.print_exc()
                    return 1

            cb = REST_HOOK(raw)
        _check(lib().ss_prove_hook(self.pk, self.ps, hip.p64(pub) if len(pub) else None, ctypes.c_size_t(len(pub)), hip.p8(tape), ctypes.c_size_t(tape.shape[0]),
                                   ctypes.byref(used), hip.p64(words), ctypes.c_size_t(n), ms, cb, None))
        return words, used.value, dict(zip(PHASES, list(ms)))

    def verify(self, words: np.ndarray) -> int:
        """SpartanSNARK::verify (src/spartan.rs:469-578) with the matrix evaluations and MSMs on the device: 0 = accept, 1..6 = failed check."""
        words = np.ascontiguousarray(words, dtype=np.uint64)
        pub = np.zeros((max(self.dims["num_public"], 1), 4), dtype=np.uint64)
        rc = lib().ss_verify(self.pk, hip.p64(words), ctypes.c_size_t(words.shape[0]), hip.p64(pub))
        if rc < 0:
            _check(rc)
        self.verified_publics = pub[: self.dims["num_public"]] if rc == 0 else None  # what verify() returns in the reference (src/spartan.rs:577)
        return rc

    # ---- wire formats (bincode framing of the reference's serde types; include/spartan_hip.h "wire formats") ----
    def proof_layout(self) -> dict:
        L = hip.SpartanLayout()
        lib().ss_proof_layout(self.pk, ctypes.byref(L))
        return L.as_dict()

    def proof_to_bytes(self, words: np.ndarray) -> bytes:
        words = np.ascontiguousarray(words, dtype=np.uint64)
        n = ctypes.c_size_t(0)
        _check(lib().ss_proof_to_bytes(self.pk, hip.p64(words), ctypes.c_size_t(len(words)), None, ctypes.c_size_t(0), ctypes.byref(n)))
        out = np.zeros(n.value, dtype=np.uint8)
        _check(lib().ss_proof_to_bytes(self.pk, hip.p64(words), ctypes.c_size_t(len(words)), hip.p8(out), ctypes.c_size_t(n.value), ctypes.byref(n)))
        return out.tobytes()

    def verify_bytes(self, data: bytes) -> int:
        """verify() on a serialised proof: 0 = accept, 1..6 = failed check (1 also for bytes that do not decode to a proof of this key's shape)."""
        arr = np.frombuffer(bytes(data), dtype=np.uint8).copy() if len(data) else np.zeros(1, dtype=np.uint8)
        pub = np.zeros((max(self.dims["num_public"], 1), 4), dtype=np.uint64)
        rc = lib().ss_verify_bytes(self.pk, hip.p8(arr), ctypes.c_size_t(len(data)), hip.p64(pub))
        if rc < 0:
            _check(rc)
        return rc

    def close(self):
        if self.ps:
            lib().ss_prep_free(self.ps)
            self.ps = None
        if self.pk:
            lib().ss_pk_free(self.pk)
            self.pk = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- NeutronNovaNIFS::prove (spartan2_amd/host/neutronnova_nifs.cpp) ----------------------------------------------------------
NIFS_HOOK = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_size_t, hip.c_u64p, hip.c_u64p)


def _c_hook(py_hook):
    def raw(_user, t, coeffs_ptr, out_ptr):
        coeffs = np.ctypeslib.as_array(coeffs_ptr, shape=(16,)).reshape(4, 4).copy()
        r = py_hook(int(t), coeffs)
        if r is not None:
            r = np.ascontiguousarray(r, dtype=np.uint64).reshape(4)
            for i in range(4):
                out_ptr[i] = int(r[i])

    return NIFS_HOOK(raw)


def tensor_decomp(n: int):
    """compute_tensor_decomp (src/neutronnova_zk.rs:56-67)."""
    ell = max(n - 1, 0).bit_length()
    return ell, 1 << ((ell + 1) // 2), 1 << (ell // 2)


_P256 = 0xFFFFFFFF00000001000000000000000000000000FFFFFFFFFFFFFFFFFFFFFFFF  # scalar field of the bench engine (src/provider/pt256.rs:55)


def mont_limbs_from_u64(vals):
    """(n,) uint64 integers -> (n, 4) Montgomery limbs of the engine's scalar field (harness helper: witness vectors of the integer R1CS generators)."""
    vals = np.ascontiguousarray(vals, dtype=np.uint64)
    u, inv = np.unique(vals, return_inverse=True)
    tab = np.array([[((int(v) << 256) % _P256 >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in u], dtype=np.uint64).reshape(-1, 4)
    return tab[inv]


def padded_witness_limbs(dims: dict, witness):
    """[shared | precommitted | rest] with every segment zero-padded to its padded length (SplitR1CSShape layout, src/r1cs/mod.rs:810-911)."""
    w = mont_limbs_from_u64(witness)
    s, p, r = dims["num_shared_unpadded"], dims["num_precommitted_unpadded"], dims["num_rest_unpadded"]
    out = np.zeros((dims["num_shared"] + dims["num_precommitted"] + dims["num_rest"], 4), dtype=np.uint64)
    out[:s] = w[:s]
    out[dims["num_shared"] : dims["num_shared"] + p] = w[s : s + p]
    out[dims["num_shared"] + dims["num_precommitted"] : dims["num_shared"] + dims["num_precommitted"] + r] = w[s + p : s + p + r]
    return out


def nifs_prepare(ctx: hip.Context, shape: hip.Shape, dims: dict, X, W_tables, small_values: bool):
    """The prep_prove part of the NIFS (cached_step_matvec / cached_step_i64, src/neutronnova_zk.rs:1520-1600): layers (+ i64 mirrors) of the
    instances. Returns an opaque handle for nifs_prove(prepared=...); free with nifs_free."""
    n = len(W_tables)
    d = dims["num_public"]
    X = np.ascontiguousarray(X, dtype=np.uint64).reshape(n, d, 4)
    d10 = (ctypes.c_uint64 * 10)(*[dims[k] for k in DIM_NAMES])
    warr = (ctypes.c_void_p * n)(*[t.h for t in W_tables])
    h = ctypes.c_void_p()
    _check(lib().nn_nifs_prepare(ctx.h, shape.h, d10, ctypes.c_size_t(n), hip.p64(X.reshape(-1)) if d else None, warr, 1 if small_values else 0, ctypes.byref(h)))
    return h


def nifs_free(handle):
    hip.lib().sp_nifs_free(handle)


def nifs_prove(ctx: hip.Context, shape: hip.Shape, dims: dict, ck: hip.CommitmentKey, comms, X, W_tables, r_W, small_values: bool, tr: hip.Transcript, py_hook,
               prepared=None):
    """NeutronNovaNIFS::prove (src/neutronnova_zk.rs:511-1273). comms (n, rows, 8), X (n, d, 4), W_tables: n resident witness tables,
    r_W (n, rows, 4); py_hook(t, coeffs (4,4)) -> r_b is the caller's `process_round`. Returns a dict of host arrays and device tables."""
    comms = np.ascontiguousarray(comms, dtype=np.uint64)
    n, rows = comms.shape[0], comms.shape[1]
    d = dims["num_public"]
    X = np.ascontiguousarray(X, dtype=np.uint64).reshape(n, d, 4)
    r_W = np.ascontiguousarray(r_W, dtype=np.uint64).reshape(n, rows, 4)
    n_padded = max(2, 1 << (n - 1).bit_length())
    ell_b = n_padded.bit_length() - 1
    _, left, right = tensor_decomp(dims["num_cons"])
    N, nv = dims["num_cons"], dims["num_shared"] + dims["num_precommitted"] + dims["num_rest"]
    out = dict(polys=np.zeros((ell_b, 4, 4), dtype=np.uint64), r_bs=np.zeros((ell_b, 4), dtype=np.uint64), E_eq=np.zeros((left + right, 4), dtype=np.uint64),
               tail=np.zeros((2, 4), dtype=np.uint64), folded_rW=np.zeros((rows, 4), dtype=np.uint64), folded_X=np.zeros((max(d, 1), 4), dtype=np.uint64),
               folded_comm=np.zeros((rows, 8), dtype=np.uint64))
    tabs = dict(A=hip.Table.zeros(ctx, N), B=hip.Table.zeros(ctx, N), C=hip.Table.zeros(ctx, N), folded_W=hip.Table.zeros(ctx, nv))
    d10 = (ctypes.c_uint64 * 10)(*[dims[k] for k in DIM_NAMES])
    warr = (ctypes.c_void_p * n)(*[t.h for t in W_tables])
    cb = _c_hook(py_hook)
    _check(lib().nn_nifs_prove(ctx.h, shape.h, d10, ck.h, ctypes.c_size_t(n), ctypes.c_size_t(rows), hip.p64(comms.reshape(-1)), hip.p64(X.reshape(-1)) if d else None,
                               warr, hip.p64(r_W.reshape(-1)), 1 if small_values else 0, prepared, tr.h, cb, None, hip.p64(out["polys"]), hip.p64(out["r_bs"]),
                               hip.p64(out["E_eq"]), hip.p64(out["tail"]), hip.p64(out["folded_rW"]), hip.p64(out["folded_X"]), hip.p64(out["folded_comm"]),
                               tabs["A"].h, tabs["B"].h, tabs["C"].h, tabs["folded_W"].h))
    out["folded_X"] = out["folded_X"][:d]
    out.update(tabs)
    return out


def nifs_prove_sharded(ctx: hip.Context, comm, shape: hip.Shape, dims: dict, ck: hip.CommitmentKey, comms_local, X_local, W_tables_local, r_W_local, small_values: bool,
                       tr: hip.Transcript, py_hook, prepared=None, out_tabs=None):
    """NeutronNovaNIFS::prove with the instances sharded over the ranks of `comm` (spartan2_amd/host/neutronnova_nifs.cpp nifs_prove_sharded,
    SURVEY.md 8(e)): this rank passes its n_local consecutive instances; every rank returns the same outputs as `nifs_prove` on the whole batch."""
    comms_local = np.ascontiguousarray(comms_local, dtype=np.uint64)
    n_local, rows = comms_local.shape[0], comms_local.shape[1]
    d = dims["num_public"]
    X_local = np.ascontiguousarray(X_local, dtype=np.uint64).reshape(n_local, d, 4)
    r_W_local = np.ascontiguousarray(r_W_local, dtype=np.uint64).reshape(n_local, rows, 4)
    n = n_local * comm.world
    ell_b = n.bit_length() - 1
    _, left, right = tensor_decomp(dims["num_cons"])
    N, nv = dims["num_cons"], dims["num_shared"] + dims["num_precommitted"] + dims["num_rest"]
    out = dict(polys=np.zeros((ell_b, 4, 4), dtype=np.uint64), r_bs=np.zeros((ell_b, 4), dtype=np.uint64), E_eq=np.zeros((left + right, 4), dtype=np.uint64),
               tail=np.zeros((2, 4), dtype=np.uint64), folded_rW=np.zeros((rows, 4), dtype=np.uint64), folded_X=np.zeros((max(d, 1), 4), dtype=np.uint64),
               folded_comm=np.zeros((rows, 8), dtype=np.uint64))
    tabs = out_tabs if out_tabs is not None else dict(A=hip.Table.zeros(ctx, N), B=hip.Table.zeros(ctx, N), C=hip.Table.zeros(ctx, N), folded_W=hip.Table.zeros(ctx, nv))
    d10 = (ctypes.c_uint64 * 10)(*[dims[k] for k in DIM_NAMES])
    warr = (ctypes.c_void_p * n_local)(*[t.h for t in W_tables_local])
    cb = _c_hook(py_hook)
    _check(lib().nn_nifs_prove_sharded(ctx.h, comm.h, shape.h, d10, ck.h, ctypes.c_size_t(n_local), ctypes.c_size_t(rows), hip.p64(comms_local.reshape(-1)),
                                       hip.p64(X_local.reshape(-1)) if d else None, warr, hip.p64(r_W_local.reshape(-1)), 1 if small_values else 0, prepared, tr.h, cb,
                                       None, hip.p64(out["polys"]), hip.p64(out["r_bs"]), hip.p64(out["E_eq"]), hip.p64(out["tail"]), hip.p64(out["folded_rW"]),
                                       hip.p64(out["folded_X"]), hip.p64(out["folded_comm"]), tabs["A"].h, tabs["B"].h, tabs["C"].h, tabs["folded_W"].h))
    out["folded_X"] = out["folded_X"][:d]
    out.update(tabs)
    return out


# ---- multi-GPU: the exchange layer and the sharded prover (spartan2_amd/host/{comm.hpp, sharded_snark.cpp}) -----------------------------
_ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)


class Comm:
    """One all-gather primitive over the ranks of the job. backend "rccl": ncclAllGather from C++ on a communicator of its own
    (the unique id travels through torch.distributed once, at creation); backend "torch": a callback into torch.distributed.all_gather (gloo in
    the tests, where the ranks share one GPU and RCCL cannot be used); world 1 without torch.distributed: "rccl" still creates a one-rank
    communicator so that the production path is the one exercised."""

    def __init__(self, rank: int, world: int, backend: str, device: int = 0):
        import torch

        self.rank, self.world, self.backend = rank, world, backend
        self.h = ctypes.c_void_p()
        self._cb = None
        if backend == "rccl":
            uid = np.zeros(128, dtype=np.uint8)
            if rank == 0:
                _check(lib().ssc_rccl_unique_id(hip.p8(uid)))
            if world > 1:
                import torch.distributed as dist

                t = torch.from_numpy(uid)
                if dist.get_backend() == "nccl":  # (PyTorch's name for its RCCL backend on ROCm)
                    t = t.to("cuda")  # (PyTorch's device-type name for the HIP device)
                dist.broadcast(t, src=0)
                uid = t.cpu().numpy().copy()
            _check(lib().ssc_comm_rccl(int(device), int(rank), int(world), hip.p8(uid), ctypes.byref(self.h)))
        elif backend == "torch":
            import torch.distributed as dist

            def raw(_user, send, nbytes, recv):
                try:
                    src = np.ctypeslib.as_array(ctypes.cast(send, ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes,))
                    mine = torch.from_numpy(src.copy())
                    outs = [torch.empty_like(mine) for _ in range(world)]
                    dist.all_gather(outs, mine)
                    dst = np.ctypeslib.as_array(ctypes.cast(recv, ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes * world,))
                    for r, o in enumerate(outs):
                        dst[r * nbytes : (r + 1) * nbytes] = o.numpy()
                    return 0
                except Exception:  # noqa: BLE001 — an exception must not unwind through the C frames
                    import traceback

                    traceback.print_exc()
                    return 1

            self._cb = _ALLGATHER_FN(raw)
            _check(lib().ssc_comm_callback(int(rank), int(world), self._cb, None, ctypes.byref(self.h)))
        else:
            raise ValueError(backend)

    def allgather(self, arr: np.ndarray) -> np.ndarray:
        arr = np.ascontiguousarray(arr)
        out = np.zeros((self.world,) + arr.shape, dtype=arr.dtype)
        _check(lib().ssc_comm_allgather(self.h, arr.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(arr.nbytes), out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def stats(self):
        s = (ctypes.c_uint64 * 2)()
        lib().ssc_comm_stats(self.h, s)
        lib().ssc_comm_small_calls.restype = ctypes.c_uint64
        return {"exchanges": int(s[0]), "bytes_gathered": int(s[1]), "copy_free_exchanges": int(lib().ssc_comm_small_calls(self.h))}

    def close(self):
        if self.h:
            lib().ssc_comm_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


SHARDED_PHASES = PHASES + ("exchanges",)


class ShardedSpartanSNARK:
    """ONE proof over all ranks of `comm` (row-sharded commitment and Az/Bz/Cz, slice-sharded sum-checks, column-sharded poly_ABC, point-range
    MSMs); every rank calls every method with the same arguments and receives the same proof words."""

    def __init__(self, ctx: hip.Context, comm: Comm, inst):
        self.ctx, self.comm, self.inst = ctx, comm, inst
        args, keep = _inst_args(inst)
        self.pk = ctypes.c_void_p()
        _check(lib().ssd_setup(ctx.h, comm.h, *args, ctypes.byref(self.pk)))
        d = (ctypes.c_uint64 * 10)()
        lib().ssd_pk_info(self.pk, d)
        self.dims = {k: int(v) for k, v in zip(DIM_NAMES, d)}
        self.ps = None

    def prep_prove(self, tape: np.ndarray, is_small=True):
        used = ctypes.c_size_t(0)
        w = np.ascontiguousarray(self.inst.witness, dtype=np.uint64)
        ps = ctypes.c_void_p()
        _check(lib().ssd_prep_prove(self.pk, hip.p64(w), ctypes.c_size_t(len(w)), int(is_small), hip.p8(tape), ctypes.c_size_t(tape.shape[0]), ctypes.byref(used),
                                    ctypes.byref(ps)))
        if self.ps:
            lib().ssd_prep_free(self.ps)
        self.ps = ps
        return used.value

    def proof_words(self):
        d = self.dims
        M = d["num_shared"] + d["num_precommitted"] + d["num_rest"]
        lx, ly = d["num_cons"].bit_length() - 1, M.bit_length()
        return 8 * (M // 2048) + 4 * d["num_public"] + 12 * lx + 12 + 8 * ly + 8 + 16 + 4 * 2048 + 8

    def prove(self, tape: np.ndarray):
        n = self.proof_words()
        words = np.zeros(n, dtype=np.uint64)
        used = ctypes.c_size_t(0)
        ms = (ctypes.c_double * 8)()
        pub = np.ascontiguousarray(self.inst.publics, dtype=np.uint64)
        _check(lib().ssd_prove(self.pk, self.ps, hip.p64(pub) if len(pub) else None, ctypes.c_size_t(len(pub)), hip.p8(tape), ctypes.c_size_t(tape.shape[0]),
                               ctypes.byref(used), hip.p64(words), ctypes.c_size_t(n), ms))
        return words, used.value, dict(zip(SHARDED_PHASES, list(ms)))

    def close(self):
        if self.ps:
            lib().ssd_prep_free(self.ps)
            self.ps = None
        if self.pk:
            lib().ssd_pk_free(self.pk)
            self.pk = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sharded_commit(ctx: hip.Context, comm: Comm, key: hip.CommitmentKey, table: hip.Table, n_local: int, blinds_local, is_small=False):
    """PCS::commit with the rows sharded by row over the ranks (BASELINE config 4's MSM leg): this rank's rows are resident in `table`;
    returns all rows (world * rows_local, 8), rank-major."""
    rows_local = (n_local + 2047) // 2048
    blinds_local = np.ascontiguousarray(blinds_local, dtype=np.uint64).reshape(rows_local, 4)
    out = np.zeros((comm.world * rows_local, 8), dtype=np.uint64)
    _check(lib().ssd_commit(ctx.h, comm.h, key.h, table.h, ctypes.c_size_t(n_local), hip.p64(blinds_local), int(is_small), hip.p64(out)))
    return out


# ---- NeutronNovaZkSNARK (spartan2_amd/host/neutronnova_zk.cpp) ---------------------------------------------------------------------------------
NN_PHASES = ("instances", "nifs", "outer_sumcheck", "inner_sumcheck", "verifier_circuit_instance", "pcs_prove", "total", "of_which_vc_round_commits")


class NeutronNovaZkSNARK:
    """setup -> prep_prove -> prove -> verify (src/neutronnova_zk.rs:1394-2343) for step / core circuits without verifier challenges: shared + precommitted
    variables (the bench circuits, benches/sha256_neutronnova.rs) or rest variables only (the reference's test circuit, src/neutronnova_zk.rs:2357-2418; beside
    shared / precommitted ones the reference's fold drops them, refused). step_insts / core_inst: frontend.R1CSInstanceInt."""

    def __init__(self, ctx: hip.Context, step_insts, core_inst):
        self.ctx, self.steps, self.core = ctx, step_insts, core_inst
        a1, k1 = _inst_args(step_insts[0])
        a2, k2 = _inst_args(core_inst)
        self.pk = ctypes.c_void_p()
        _check(lib().nnz_setup(ctx.h, ctypes.c_size_t(len(step_insts)), *a1, *a2, ctypes.byref(self.pk)))
        info = (ctypes.c_uint64 * 8)()
        dig = np.zeros(32, dtype=np.uint8)
        lib().nnz_pk_info(self.pk, info, hip.p8(dig))
        self.info = dict(zip(("nb", "nx", "ny", "vc_rounds", "vc_vars", "vc_cons", "vc_cons_unpadded", "vc_public"), [int(x) for x in info]))
        self.vk_digest = dig
        self.ps = None
        lib().nnz_proof_words.restype = ctypes.c_size_t

    def prep_prove(self, tape: np.ndarray, is_small=True):
        sw = np.ascontiguousarray(np.stack([np.asarray(i.witness, dtype=np.uint64) for i in self.steps]))
        sp = np.ascontiguousarray(np.stack([np.asarray(i.publics, dtype=np.uint64) for i in self.steps]))
        cw = np.ascontiguousarray(self.core.witness, dtype=np.uint64)
        cp = np.ascontiguousarray(self.core.publics, dtype=np.uint64)
        used = ctypes.c_size_t(0)
        ps = ctypes.c_void_p()
        _check(lib().nnz_prep_prove(self.pk, ctypes.c_size_t(len(self.steps)), hip.p64(sw), ctypes.c_size_t(sw.shape[1]), hip.p64(sp), ctypes.c_size_t(sp.shape[1]), hip.p64(cw),
                                    hip.p64(cp), int(is_small), hip.p8(tape), ctypes.c_size_t(tape.shape[0]), ctypes.byref(used), ctypes.byref(ps)))
        if self.ps:
            lib().nnz_prep_free(self.ps)
        self.ps = ps
        return used.value

    def prove(self, tape: np.ndarray, reference_order=False):
        """reference_order: the one-thread driver in the statement order of src/neutronnova_zk.rs:1609-2093 (no side jobs, the opening as one sp_hyrax_prove)"""
        n = lib().nnz_proof_words(self.pk)
        words = np.zeros(n, dtype=np.uint64)
        used = ctypes.c_size_t(0)
        ms = (ctypes.c_double * 8)()
        fn = lib().nnz_prove_reference_order if reference_order else lib().nnz_prove
        _check(fn(self.pk, self.ps, hip.p8(tape), ctypes.c_size_t(tape.shape[0]), ctypes.byref(used), hip.p64(words), ctypes.c_size_t(n), ms))
        return words, used.value, dict(zip(NN_PHASES, list(ms)))

    def verify(self, words: np.ndarray) -> int:
        """NeutronNovaZkSNARK::verify (src/neutronnova_zk.rs:2096-2343) with the commitment fold, the six matrix evaluations and the opening's MSMs on the
        device: 0 = accept, else the failed check (1 shape / encoding, 2 verifier-circuit instance, 4 relaxed Spartan proof, 5 public values, 6 opening)."""
        words = np.ascontiguousarray(words, dtype=np.uint64)
        rc = lib().nnz_verify(self.pk, hip.p64(words), ctypes.c_size_t(words.shape[0]))
        if rc < 0:
            _check(rc)
        return rc

    def proof_to_bytes(self, words: np.ndarray) -> bytes:
        """NeutronNovaZkSNARK as bincode bytes (src/neutronnova_zk.rs:1373-1385)"""
        words = np.ascontiguousarray(words, dtype=np.uint64)
        n = ctypes.c_size_t(0)
        _check(lib().nnz_proof_to_bytes(self.pk, hip.p64(words), ctypes.c_size_t(len(words)), None, ctypes.c_size_t(0), ctypes.byref(n)))
        out = np.zeros(n.value, dtype=np.uint8)
        _check(lib().nnz_proof_to_bytes(self.pk, hip.p64(words), ctypes.c_size_t(len(words)), hip.p8(out), ctypes.c_size_t(n.value), ctypes.byref(n)))
        return out.tobytes()

    def proof_from_bytes(self, data: bytes) -> np.ndarray:
        arr = np.frombuffer(bytes(data), dtype=np.uint8).copy() if len(data) else np.zeros(1, dtype=np.uint8)
        lib().nnz_proof_words.restype = ctypes.c_size_t
        words = np.zeros(lib().nnz_proof_words(self.pk), dtype=np.uint64)
        _check(lib().nnz_proof_from_bytes(self.pk, hip.p8(arr), ctypes.c_size_t(len(data)), hip.p64(words), ctypes.c_size_t(len(words))))
        return words

    def verify_bytes(self, data: bytes) -> int:
        arr = np.frombuffer(bytes(data), dtype=np.uint8).copy() if len(data) else np.zeros(1, dtype=np.uint8)
        rc = lib().nnz_verify_bytes(self.pk, hip.p8(arr), ctypes.c_size_t(len(data)))
        if rc < 0:
            _check(rc)
        return rc

    def close(self):
        if self.ps:
            lib().nnz_prep_free(self.ps)
            self.ps = None
        if self.pk:
            lib().nnz_pk_free(self.pk)
            self.pk = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
