/* spartan_hip.h — C ABI of libspartan_hip.so, the MI355X (gfx950) implementation of the Spartan2 prover
 * hot path. Plain pointers and sizes only; no C++/torch types. Each entry point cites the reference
 * interface (path:line in microsoft/Spartan2) it replaces; INTEGRATION.md shows the Rust `extern "C"`
 * binding a maintainer would add on the reference side.
 *
 * Conventions:
 *   F    = uint64_t[4], little-endian limbs, Montgomery form (x * 2^256 mod p), canonical — the in-memory
 *          form of halo2curves field elements the reference reaches through `.0` (src/big_num/macros.rs:59-72).
 *          Scalars are in the engine's scalar field (T256: the P-256 base prime, src/provider/pt256.rs:55).
 *   Aff  = uint64_t[8] = x | y in the engine's base field (pt256.rs:56); (0,0) encodes the identity.
 *   Jacobian results are always returned normalised to Aff (canonical), never as raw (X,Y,Z).
 *   Return value: 0 = ok; negative = -(error class) mirroring `enum SpartanError` (src/errors.rs:13-110);
 *   sp_last_error() gives the message for the calling thread.
 *   Host buffers are caller-owned and only read/written during the call. sp_* handles own device memory;
 *   a handle may be used from one thread at a time, distinct handles concurrently.
 *   There is NO CPU fallback: every entry point fails with SP_ERR_NO_DEVICE if no gfx950 device is usable.
 *
 * Interoperability: this ABI takes the commitment key (generators) and every transcript byte from its caller, so a Rust caller that derives the
 * key with the reference's `from_label` and absorbs the reference's vk digest gets reference-compatible group elements and challenges out of
 * these entry points. The C++ drivers of this repo (spartan2_amd/host/) compute the vk digest and the proof bytes in the reference's own framing
 * (bincode + SHA-256, "wire formats" below); what stays a documented substitute is the generator derivation (third-party hash_to_curve,
 * DESIGN.md section 6), so THEIR keys are not the reference binary's keys and no byte equality with a reference-produced proof is claimed.
 */
#ifndef SPARTAN_HIP_H
#define SPARTAN_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SP_OK 0
#define SP_ERR_INVALID_INPUT_LENGTH (-1) /* SpartanError::InvalidInputLength   (errors.rs) */
#define SP_ERR_INVALID_WITNESS_LENGTH (-2) /* SpartanError::InvalidWitnessLength */
#define SP_ERR_DIVISION_BY_ZERO (-3)     /* SpartanError::DivisionByZero       */
#define SP_ERR_INTERNAL_TRANSCRIPT (-4)  /* SpartanError::InternalTranscriptError */
#define SP_ERR_INTERNAL (-5)             /* SpartanError::InternalError        */
#define SP_ERR_NO_DEVICE (-100)          /* no usable gfx950 device / HIP failure */

typedef struct sp_ctx sp_ctx;               /* one device + stream + scratch */
typedef struct sp_table sp_table;           /* MultilinearPolynomial<Scalar> resident in HBM */
typedef struct sp_transcript sp_transcript; /* Keccak256Transcript */
typedef struct sp_shape sp_shape;           /* SplitR1CSShape with PrecomputedSparseMatrix / FilteredSpmv */
typedef struct sp_ck sp_ck;                 /* HyraxCommitmentKey: bases + h + FixedBaseMul table of h */

const char* sp_last_error(void);
/* Cooperative waiting. Every host-side wait of the library for the device — a round's result slot, a mailbox answer, a helper's flag — spins on a
 * `pause`; a thread that has installed a hook gets the hook called instead, once per poll. A driver that keeps several proofs in flight on ONE thread
 * (each on a stack of its own) switches to another proof there: what the reference gets from rayon, whose workers take other work while a task
 * waits. Per calling thread; NULL restores the spin. sp_relax() = one such wait step, for waits a driver above the ABI implements itself. */
typedef void (*sp_wait_hook)(void* user);
int sp_set_wait_hook(sp_wait_hook hook, void* user);
void sp_relax(void);

/* one context per GPU; device = HIP ordinal (LOCAL_RANK under torch.distributed.run). */
int sp_ctx_create(int device, sp_ctx** out);
/* Makes the context's device current for the CALLING thread (HIP keeps a current device per thread; a new thread starts on device 0). The thread that
 * created the context needs no call; a helper thread (see sp_points_upload) calls this once before its first call on the context. */
int sp_ctx_bind_thread(sp_ctx* ctx);
/* the HIP ordinal the context was created on (a host driver that wants a second context on the same GPU for work it runs beside its main
 * stream of calls — e.g. transcript-independent commitments under a sum-check — creates it with this) */
int sp_ctx_device(const sp_ctx* ctx);
/* A promise about the round hooks handed to the batched sum-checks (sp_sumcheck_cubic_outer_pow_batched, sp_sumcheck_quad_batched) on this context: they
 * do NOT queue work on the context's main stream or wait for it (the reference's hook is `process_round`, src/sumcheck.rs:747-755 - synthesis, a narrow
 * commitment, transcript: host work when the commitment goes through sp_hyrax_commit_split_*). With the promise the library queues the next round's fused
 * bind + evaluate launch BEFORE it calls the hook; the kernel waits at the challenge mailbox, so its launch and dispatch run under the hook's host time.
 * A hook that breaks the promise would queue behind that waiting kernel: it is released by the mailbox watchdog after 8 s and the call fails. Default off. */
int sp_ctx_round_hooks_host_only(sp_ctx* ctx, int on);
void sp_ctx_destroy(sp_ctx* ctx);
int sp_ctx_synchronize(sp_ctx* ctx);
/* time of the most recent instrumented kernel class, measured with hipEvents on the context's stream
 * (what = "bind", "eval_cubic", "eval_quad", ...): accumulated milliseconds, launches and algorithmic bytes. */
int sp_ctx_kernel_stats(sp_ctx* ctx, const char* what, double* ms, uint64_t* launches, uint64_t* alg_bytes);
int sp_ctx_reset_stats(sp_ctx* ctx, int enable_timing);
/* restrict the instrumentation to one kernel class (NULL or "" = all) */
int sp_ctx_stats_filter(sp_ctx* ctx, const char* only);
/* diagnostics of the challenge mailbox (no reference counterpart: the reference's rounds are function calls). out[0] = challenges a waiting kernel took
 * from the host-memory mirror because its device-memory line had not answered, out[1] = waits ended by the watchdog, out[2] / out[3] = sequence number
 * wanted / sequence number the device line showed at the most recent mirror answer, out[4] = 1 when the ring lives in device memory. Reads and clears. */
int sp_ctx_mail_stats(sp_ctx* ctx, uint64_t out[5]);

/* ---- MultilinearPolynomial (src/polys/multilinear.rs:34-164) ------------------------------------- */
/* MultilinearPolynomial::new / new_with_halves (:62-75). lo_eff/hi_eff = SIZE_MAX for "unknown". */
int sp_table_from_host(sp_ctx* ctx, const uint64_t* z, size_t len, size_t lo_eff, size_t hi_eff, sp_table** out);
/* Witness upload in the form the reference's frontend holds it for is_small circuits (machine words, src/bellpepper/r1cs.rs:303-409; msm_small takes them as
 * such, src/provider/pcs/hyrax_pc.rs:266-292): t[off + i] = vals[i] as a field element for i < cnt - 8 bytes a value cross the bus instead of 32 and the
 * Montgomery form is produced on the device (one product by R^2 per value that is neither 0 nor 1). Ordered on the context's stream; `vals` is free on return. */
int sp_table_write_u64(sp_ctx* ctx, sp_table* t, size_t off, const uint64_t* vals, size_t cnt);
/* the same for a 0/1 witness packed 8 values a byte, value i = bit (i & 7) of bits[i >> 3] (the booleanised form of a SHA-256 / bit-decomposition witness:
 * 128 KiB for 2^20 values) */
int sp_table_write_bits(sp_ctx* ctx, sp_table* t, size_t off, const uint8_t* bits, size_t cnt);
/* zero-filled table of `len` elements with the given zero-structure hints */
int sp_table_zeros(sp_ctx* ctx, size_t len, size_t lo_eff, size_t hi_eff, sp_table** out);
/* host -> device write of cnt elements at element offset off */
int sp_table_write(sp_ctx* ctx, sp_table* t, size_t off, const uint64_t* z, size_t cnt);
/* the same without waiting for the copy: the elements (at most 2048) are staged in pinned memory owned by the context, so the caller's buffer is free
 * on return and the write is ordered on the context's stream like every other table operation */
int sp_table_write_async(sp_ctx* ctx, sp_table* t, size_t off, const uint64_t* z, size_t cnt);
/* zero cnt elements starting at element offset off (device memset) */
int sp_table_zero(sp_ctx* ctx, sp_table* t, size_t off, size_t cnt);
/* device -> device copy */
int sp_table_copy(sp_ctx* ctx, sp_table* dst, size_t dst_off, const sp_table* src, size_t src_off, size_t cnt);
/* dst[dst_off, dst_off + cnt) = src[src_off, ..) and dst[zero_off, zero_off + zero_cnt) = 0 (device -> device, one launch) on a stream BESIDE the context's
 * main one: for table contents no call reads for a while (the bulk of z = [W | 1 | public | 0...], src/spartan.rs:246-253, whose next reader after the
 * matrix-vector product is the inner sum-check). behind_queued != 0: it runs behind whatever the context had queued when it was called; 0: the caller
 * vouches that no queued work reads or writes the two ranges of dst (or writes the range of src). Calls that follow are NOT ordered behind it until
 * sp_ctx_aside_join, which makes everything queued after the join wait for all earlier `_aside` work. */
int sp_table_assemble_aside(sp_ctx* ctx, sp_table* dst, size_t dst_off, const sp_table* src, size_t src_off, size_t cnt, size_t zero_off, size_t zero_cnt,
                            int behind_queued);
int sp_ctx_aside_join(sp_ctx* ctx);
/* dst[dst_off + j] = src[src_off + j * stride], j < cnt (device -> device). The slice of a table sharded on its LAST k variables (rank g of 2^k
 * holds Z[(j << k) | g]: src_off = g, stride = 2^k) — SURVEY.md 8(e), "sum-check by evaluation-table slice". */
int sp_table_gather_strided(sp_ctx* ctx, sp_table* dst, size_t dst_off, const sp_table* src, size_t src_off, size_t stride, size_t cnt);
/* the inverse: dst[dst_off + j * stride] = src[src_off + j], j < cnt - rank g's gathered slice back into the interleaved table (dst_off = g, stride = 2^k) */
int sp_table_scatter_strided(sp_ctx* ctx, sp_table* dst, size_t dst_off, size_t stride, const sp_table* src, size_t src_off, size_t cnt);
/* Index / into_vec (:166-173, :87-89) */
int sp_table_read(sp_ctx* ctx, const sp_table* t, size_t off, size_t cnt, uint64_t* out);
int sp_table_info(const sp_table* t, size_t* len, size_t* lo_eff, size_t* hi_eff);
int sp_table_set_len(sp_table* t, size_t len, size_t lo_eff, size_t hi_eff);
/* Non-owning window [off, off + len) onto t's storage (valid while t lives; free with sp_table_free, which leaves the storage alone): the slices
 * of an all-gathered buffer handed to sp_fold_tables, a row block of a witness. */
int sp_table_view(const sp_table* t, size_t off, size_t len, sp_table** out);
/* The device address of the table (and the bytes allocated behind it) for a collective the caller issues itself (ncclAllGather on layers:
 * SURVEY.md 8(e)); work queued on the context must be complete (sp_ctx_synchronize) before another stream touches it. */
int sp_table_device_ptr(const sp_table* t, void** out, size_t* cap_bytes);
void sp_table_free(sp_table* t);
/* MultilinearPolynomial::bind_poly_var_top(&r) (:95-164): in place, len halves, all three zero-structure branches */
int sp_table_bind_top(sp_ctx* ctx, sp_table* t, const uint64_t r[4]);
/* EqPolynomial::evals_from_points[_into] (src/polys/eq.rs:59-117): table of 2^ell evaluations, r[0] on the index MSB */
int sp_eq_table(sp_ctx* ctx, const uint64_t* r, size_t ell, sp_table** out);
/* the `_into` form (src/polys/eq.rs:96-117) reusing an existing allocation of at least 2^ell elements */
int sp_eq_table_into(sp_ctx* ctx, const uint64_t* r, size_t ell, sp_table* out);
/* The same table for a point a sum-check is still drawing (evals_rx of src/spartan.rs:316 needs r_x, whose last coordinate is the outer sum-check's last
 * challenge): `_begin`, given all but the last two, three or four coordinates (ell - 4 <= n_known <= ell - 2, 12 <= ell <= 20), builds the half tables
 * on a stream of its own under the remaining rounds; `_finish`, given all of them, is one launch. Without a matching `_begin` (or with a different
 * prefix) `_finish` is sp_eq_table_into. */
int sp_eq_table_begin(sp_ctx* ctx, const uint64_t* r_known, size_t n_known, size_t ell);
int sp_eq_table_finish(sp_ctx* ctx, const uint64_t* r, size_t ell, sp_table* out);

/* ---- Keccak256Transcript (src/provider/keccak.rs:18-105, trait src/traits/transcript.rs:21-33) ----- */
int sp_transcript_new(sp_ctx* ctx, const uint8_t* label, size_t n, sp_transcript** out);
int sp_transcript_absorb(sp_transcript* t, const uint8_t* label, size_t ln, const uint8_t* bytes, size_t n);
int sp_transcript_dom_sep(sp_transcript* t, const uint8_t* bytes, size_t n);
/* on != 0: absorbs of >= 4 KiB (a commitment's 64 bytes per row: hundreds of Keccak blocks) are hashed on the library's hashing thread while the caller
 * goes on to its next call; every later use of the transcript joins first, so the sponge sees the reference's byte sequence. For single-threaded
 * callers that follow src/spartan.rs statement by statement (the commitment of the rest segment and the build of z run beside the hashing); a driver
 * that already hashes on a thread of its own leaves it off (default). */
int sp_transcript_set_async(sp_transcript* t, int on);
int sp_transcript_squeeze(sp_transcript* t, const uint8_t* label, size_t ln, uint64_t out[4]);
/* absorb(label, bytes) (keccak.rs:96-99) split in two for long inputs that are known early (comm_W is 64 KiB = 480 Keccak blocks, 0.2 ms of a
 * 1.9 ms prove): sp_transcript_preabsorb hashes label || bytes into a sponge state on any thread, without a transcript;
 * sp_transcript_absorb_prepared installs it. The running hasher restarts at every squeeze (keccak.rs:92), so the result equals the plain absorb
 * exactly when nothing has been absorbed since the last squeeze (or since new) — otherwise the call fails with SP_ERR_INTERNAL_TRANSCRIPT and
 * changes nothing. */
typedef struct sp_absorb_state sp_absorb_state;
int sp_transcript_preabsorb(const uint8_t* label, size_t ln, const uint8_t* bytes, size_t n, sp_absorb_state** out);
int sp_transcript_absorb_prepared(sp_transcript* t, const sp_absorb_state* s);
void sp_absorb_state_free(sp_absorb_state* s);
/* Keccak256Transcript derives Clone (keccak.rs:25); lets a caller keep the state after a prefix that repeats across proves */
int sp_transcript_clone(const sp_transcript* t, sp_transcript** out);
void sp_transcript_free(sp_transcript* t);

/* ---- sum-check (src/sumcheck.rs) -------------------------------------------------------------------- */
/* SumcheckProof::prove_cubic_with_three_inputs (:502-571) incl. eq_sumcheck::EqSumCheckInstance (:920-1429).
 * Tables are bound in place down to length 1. out_cpolys: ell x 3 F (compressed polys: c0, c2, c3). */
int sp_sumcheck_cubic3(sp_ctx* ctx, const uint64_t claim[4], const uint64_t* taus, size_t ell, sp_table* A, sp_table* B, sp_table* C,
                       sp_transcript* tr, uint64_t* out_cpolys, uint64_t* out_r, uint64_t out_final[12]);
/* The same prover with the per-pair products of round 1 supplied by sp_multiply_vec_incremental_round0 (p0, p1: N/2 elements each): round 1's
 * sums are then t0 = sum E(x) p0[x], t_inf = sum E(x) p1[x]. Bit-identical output. */
int sp_sumcheck_cubic3_round0(sp_ctx* ctx, const uint64_t claim[4], const uint64_t* taus, size_t ell, sp_table* A, sp_table* B, sp_table* C, const sp_table* p0,
                              const sp_table* p1, sp_transcript* tr, uint64_t* out_cpolys, uint64_t* out_r, uint64_t out_final[12]);
/* SumcheckProof::prove_quad (:190-247) with compute_eval_points_quad's eff_pairs bound (:128-174).
 * out_cpolys: rounds x 2 F (c0, c2). */
/* sp_sumcheck_quad_observed additionally calls observe(user, round, r) right after the challenge of each round (0-based) has been drawn, so the
 * caller can start work that needs only a prefix of the challenges (the driver starts comm_LZ's MSM once the row variables are bound). */
typedef void (*sp_challenge_hook)(void* user, size_t round, const uint64_t r[4]);
int sp_sumcheck_quad_observed(sp_ctx* ctx, const uint64_t claim[4], size_t rounds, sp_table* A, sp_table* B, sp_transcript* tr, sp_challenge_hook observe,
                              void* user, uint64_t* out_cpolys, uint64_t* out_r, uint64_t out_final[8]);
int sp_sumcheck_quad(sp_ctx* ctx, const uint64_t claim[4], size_t rounds, sp_table* A, sp_table* B, sp_transcript* tr, uint64_t* out_cpolys,
                     uint64_t* out_r, uint64_t out_final[8]);
/* The same two provers on HOST tables of 2^ell / 2^rounds elements (bound in place), every round on the calling thread and the library's polling host
 * threads (round 6): for the small sum-checks whose tables the caller holds on the host anyway - RelaxedR1CSSpartanProof::prove over the ZK verifier
 * circuit's instance (src/spartan_relaxed.rs:98-213: 2^9 and 2^12 elements), where a round is ~10 n products against ~10 us a round of launch and bus
 * latency on the device. Same polynomials, transcript and final claims as the device forms. `ctx` is not used (may be NULL). */
int sp_sumcheck_cubic3_host(sp_ctx* ctx, const uint64_t claim[4], const uint64_t* taus, size_t ell, uint64_t* A, uint64_t* B, uint64_t* C, sp_transcript* tr,
                            uint64_t* out_cpolys, uint64_t* out_r, uint64_t out_final[12]);
int sp_sumcheck_quad_host(sp_ctx* ctx, const uint64_t claim[4], size_t rounds, uint64_t* A, uint64_t* B, sp_transcript* tr, uint64_t* out_cpolys, uint64_t* out_r,
                          uint64_t out_final[8]);
/* EqSumCheckInstance::evaluation_points_zero_check_round0 (src/sumcheck.rs:1163-1271; the round-0 shortcut of the *_zk cubic provers, :595): on a
 * zero-check (claim 0, A o B = C on the hypercube) t(0) vanishes, so only t_inf = sum E(x) (A1 - A0)(B1 - B0) is computed (C is not read) and the
 * evaluations (s(0), s(2), s(3)) of the round polynomial are derived from it (derive_from_claim :1276-1324, or the tau = 0 fallback :1244-1268).
 * out = {eval_0, eval_2, eval_3}; the tables are not modified. */
int sp_eval_cubic_zero_check_round0(sp_ctx* ctx, const uint64_t* taus, size_t ell, const sp_table* A, const sp_table* B, uint64_t out[12]);
/* DelayedReduction dot product reduce(sum a_i * b_i) over the first n elements (src/big_num/delayed_reduction.rs:41-84;
 * call sites src/spartan.rs:330-341) */
int sp_table_dot(sp_ctx* ctx, const sp_table* a, const sp_table* b, size_t n, uint64_t out[4]);

/* compute_eval_points_cubic_with_additive_term_with_outer_pow (src/sumcheck.rs:366-498; NeutronNova batched outer rounds):
 * out = evaluations at 0, 2, 3 (3 F). pow_left/pow_right are the two halves of the power-of-tau table (PowPolynomial::split_evals,
 * src/polys/power.rs:65-86); when len(A)/2 < len(pow_left) the reference's 4-table fallback (:262-342) is taken. */
int sp_eval_cubic_outer_pow(sp_ctx* ctx, const sp_table* pow_left, const sp_table* pow_right, const sp_table* A, const sp_table* B, const sp_table* C,
                            uint64_t out[12]);
/* R1CSWitness::fold_multiple (src/r1cs/mod.rs:570-660): out[j] = sum_i weights[i] * Ws[i][j] for j < len */
int sp_fold_tables(sp_ctx* ctx, const sp_table* const* Ws, size_t n, const uint64_t* weights, size_t len, sp_table* out);
/* weights_from_r (src/r1cs/mod.rs:153-166): n weights from ell challenges (host-side, O(n ell)) */
int sp_weights_from_r(const uint64_t* r_bs, size_t ell, size_t n, uint64_t* out);

/* ---- R1CS (src/r1cs/sparse.rs, src/r1cs/mod.rs) ------------------------------------------------------- */
typedef struct sp_csr {
  const uint64_t* data;    /* nnz F */
  const uint32_t* indices; /* nnz column ids (already padded layout, SplitR1CSShape::new :855-881) */
  const uint64_t* indptr;  /* rows+1 */
} sp_csr;
typedef struct sp_dims {
  uint64_t num_cons, num_cons_unpadded;
  uint64_t num_shared, num_precommitted, num_rest; /* padded */
  uint64_t num_shared_unpadded, num_precommitted_unpadded, num_rest_unpadded;
  uint64_t num_public, num_challenges;
} sp_dims;
/* SplitR1CSShape::precompute (src/r1cs/mod.rs:1059-1073): classify entries (+1 / -1 / small / general,
 * src/r1cs/sparse.rs:49-134), build the filtered COO (:305-358) and the column-major copy used by poly_ABC. */
int sp_shape_from_csr(sp_ctx* ctx, const sp_csr* A, const sp_csr* B, const sp_csr* C, const sp_dims* dims, sp_shape** out);
void sp_shape_free(sp_shape* s);
/* entry counts of the precomputed structures: out = {nnz A, B, C (multiply_vec / poly_ABC), filtered nnz A, B, C (multiply_vec_incremental_into),
 * long columns, short columns} — the algorithmic-byte accounting of SURVEY.md 8(d) needs them */
int sp_shape_info(const sp_shape* s, uint64_t out[8]);
/* SplitR1CSShape::multiply_vec (:1075-1107). z has num_vars + 1 + num_public + num_challenges elements. */
int sp_multiply_vec(sp_ctx* ctx, const sp_shape* s, const sp_table* z, sp_table* az, sp_table* bz, sp_table* cz);
/* SplitR1CSShape::multiply_vec_batched (:1130-1166 -> PrecomputedSparseMatrix::multiply_vec_batched, sparse.rs:237-302): the three products for
 * `count` vectors z_k; az / bz / cz are arrays of `count` output tables */
int sp_multiply_vec_batched(sp_ctx* ctx, const sp_shape* s, const sp_table* const* zs, size_t count, sp_table* const* az, sp_table* const* bz, sp_table* const* cz);
/* SplitR1CSShape::multiply_vec_incremental_into (:1170-1211) */
int sp_multiply_vec_incremental(sp_ctx* ctx, const sp_shape* s, const sp_table* z, const sp_table* caz, const sp_table* cbz, const sp_table* ccz,
                                sp_table* az, sp_table* bz, sp_table* cz);
/* multiply_vec_incremental_into (:1170-1211) that also emits the tau-independent halves of the outer sum-check's first evaluation
 * (evaluation_points_cubic_with_three_inputs, src/sumcheck.rs:1041-1105): p0[i] = Az[i] Bz[i] - Cz[i] and p1[i] = (Az[i + N/2] - Az[i])(Bz[i + N/2] - Bz[i])
 * for i < N/2, in a second streaming pass queued right behind the product. sp_sumcheck_cubic3_round0 consumes them; the values of every proof element are
 * unchanged (the weighting by the eq tables, which needs tau, stays in the sum-check). */
int sp_multiply_vec_incremental_round0(sp_ctx* ctx, const sp_shape* s, const sp_table* z, const sp_table* caz, const sp_table* cbz, const sp_table* ccz,
                                       sp_table* az, sp_table* bz, sp_table* cz, sp_table* p0, sp_table* p1);
/* SplitR1CSShape::bind_and_prepare_poly_ABC[_full] (:1235-1321): out[col] = sum_row rx[row] (A + r B + r^2 C)[row,col],
 * written into the first out_len elements of `out` */
int sp_poly_abc(sp_ctx* ctx, const sp_shape* s, const sp_table* rx, const uint64_t r[4], size_t out_len, sp_table* out);

/* ---- group / MSM (src/provider/traits.rs:118-162 DlogGroupExt, src/provider/msm.rs) --------------------- */
/* DlogGroupExt::vartime_multiscalar_mul (msm.rs:187-222): sum s_i * g_i. scalars / bases on the host. Below 4096 points the one-block-per-window
 * latency form of the Hyrax row MSMs, from there on the multi-block Pippenger (window width by n; multi-block counting sort; see sp_msm_points). */
int sp_msm(sp_ctx* ctx, const uint64_t* scalars, const uint64_t* bases, size_t n, uint64_t out_aff[8]);
/* DlogGroupExt::vartime_multiscalar_mul_small (msm.rs:367-409) */
int sp_msm_small_u64(sp_ctx* ctx, const uint64_t* scalars, const uint64_t* bases, size_t n, uint64_t out_aff[8]);

/* DlogGroupExt::vartime_multiscalar_mul_shared_weights (src/provider/traits.rs:158-161 -> msm.rs:228-356): one weight vector,
 * `rows` base rows of n affine points each (row-major); one affine result per row. FoldingEngineTrait::fold_commitments
 * (hyrax_pc.rs:737-793) is this call on the instances' commitment rows. */
int sp_msm_shared_weights(sp_ctx* ctx, const uint64_t* weights, size_t n, const uint64_t* bases_rows_aff, size_t rows, uint64_t* out_rows_aff);
/* the same on the context's auxiliary stream and its workspaces: callable from a helper thread beside the owner's calls on the main stream (fold_commitments,
 * src/neutronnova_zk.rs:1204-1211, beside fold_witnesses / the layer folds: the group work needs the weights alone) */
int sp_msm_shared_weights_aux(sp_ctx* ctx, const uint64_t* weights, size_t n, const uint64_t* bases_rows_aff, size_t rows, uint64_t* out_rows_aff);
/* vartime_scalar_mul (src/provider/msm.rs:779-867, width-5 wNAF) of n points by ONE scalar: out[i] = scalar * points[i]. The call site is the
 * two-term fold with a unit weight (hyrax_pc.rs:757-776): see sp_fold_commitments2. Few points run on the host side of the library (a dependent
 * chain of ~300 group operations: one CPU core finishes it 40x sooner than one GPU lane), many on the device, one lane per point. */
int sp_vartime_scalar_mul(sp_ctx* ctx, const uint64_t* points_aff, size_t n, const uint64_t scalar[4], uint64_t* out_aff);
/* FixedBaseMul over arbitrary points (src/provider/msm.rs:637-773): `precompute` builds 32 x 255 affine window multiples per point (1 <= n <= 2048;
 * 510 KiB per point), `multi_mul` = sum_i scalars[i] * point_i as table lookups added in ONE launch on the context's auxiliary stream (callable from a
 * helper thread beside the owner's calls on the main stream), scalars and result through mapped memory: no copy, no stream synchronise, no host tail.
 * Call site: comm_LZ of HyraxPCS::prove (hyrax_pc.rs:387-478) as sum_i L_i * comm_W[i] over the row commitments a prepared witness already holds. */
typedef struct sp_fbtables sp_fbtables;
int sp_fbtables_create(sp_ctx* ctx, const uint64_t* points_aff, size_t n, sp_fbtables** out);
/* the same, returning once the build is QUEUED on a lowest-priority stream of the context's own: what a shim's prep_prove (src/spartan.rs:176-216) calls
 * for the rows it has just committed - the tables are first read at the end of the first prove on the state. sp_fbtables_ready: 1 = built, 0 = still
 * building (wait != 0: blocks until built), < 0 = error. sp_fbtables_multi_mul* wait by themselves; sp_hyrax_prove_announce_tables does NOT wait - an
 * opening announced before the tables have landed walks the key's tables instead (same proof). */
int sp_fbtables_create_async(sp_ctx* ctx, const uint64_t* points_aff, size_t n, sp_fbtables** out);
int sp_fbtables_ready(const sp_fbtables* t, int wait);
/* test / diagnostic access: `count` entries (affine points, 8 words each) from entry `first` of the n x 32 x 255 array; entry (i, j, d - 1) = d 2^(8j) point_i */
int sp_fbtables_read(const sp_fbtables* t, size_t first, size_t count, uint64_t* out);
void sp_fbtables_free(sp_fbtables* t);
int sp_fbtables_multi_mul(sp_ctx* ctx, const sp_fbtables* t, const uint64_t* scalars, size_t n, uint64_t out_aff[8]);
/* the same in two halves: _begin launches, _finish waits for the result (one multiplication in flight per context) */
int sp_fbtables_multi_mul_begin(sp_ctx* ctx, const sp_fbtables* t, const uint64_t* scalars, size_t n);
/* the same for scalars eq(r_1 .. r_k, .) over the first nfixed tables and one more scalar for the last table (comm_LZ = <L, comm_W> with the zero rows folded
 * into h, hyrax_pc.rs:446-455), handed over ONE LEVEL SHORT: P = eq(r_1 .. r_(k-1), .) (ceil(nfixed / 2) elements, EqPolynomial::evals_from_points order,
 * src/polys/eq.rs:66-76), S01 = S0 | S1 (the last scalar is S0 + r_k (S1 - S0)) and r_k; the kernel forms the last level (one product per scalar instead of
 * 2^(k-1) host products between the challenge and the launch). t->n must be nfixed + 1 <= 1023. Collected by sp_fbtables_multi_mul_finish. */
int sp_fbtables_multi_mul_begin_eq(sp_ctx* ctx, const sp_fbtables* t, const uint64_t* P, size_t nfixed, const uint64_t S01[8], const uint64_t r_last[4]);
int sp_fbtables_multi_mul_finish(sp_ctx* ctx, uint64_t out_aff[8]);
/* FoldingEngineTrait::fold_commitments for two commitments with weights (1, w) (hyrax_pc.rs:757-776): out[i] = p[i] + w * q[i] per row */
int sp_fold_commitments2(sp_ctx* ctx, const uint64_t* p_rows_aff, const uint64_t* q_rows_aff, size_t rows, const uint64_t w[4], uint64_t* out_rows_aff);
/* The same in two calls for a weight that is drawn late (round 6). comm = p + w * q of the folded opening (src/neutronnova_zk.rs:2019-2051 -> fold_commitments,
 * hyrax_pc.rs:757-776) has q - the core instance's commitment rows, the commitment of one evaluation - long before c_eval: _begin starts the doubling ladders
 * 2^j q_i (j = 0 .. 256, normalised) of every row on the library's polling host threads and returns at once; _finish adds the ladder points of w's
 * non-adjacent form (~85 mixed additions a row instead of 256 doublings + 51 additions behind the challenge) and p. Same points as sp_fold_commitments2;
 * without walkers _finish is that call. One _finish or _drop per _begin; q_rows_aff is copied. */
typedef struct sp_fold2_job sp_fold2_job;
int sp_fold_commitments2_begin(sp_ctx* ctx, const uint64_t* q_rows_aff, size_t rows, sp_fold2_job** job);
int sp_fold_commitments2_finish(sp_ctx* ctx, sp_fold2_job* job, const uint64_t* p_rows_aff, const uint64_t w[4], uint64_t* out_rows_aff);
void sp_fold_commitments2_drop(sp_fold2_job* job);
/* sum of n affine points (host side of the library; the combine step of a point-range-sharded MSM: RCCL has no EC-add reduction,
 * so ranks all-gather their partial points and add them locally — SURVEY.md 8(e)) */
int sp_point_sum(const uint64_t* points_aff, size_t n, uint64_t out_aff[8]);

/* ---- Hyrax PCS (src/provider/pcs/hyrax_pc.rs, src/provider/pcs/ipa.rs) ------------------------------------ */
/* HyraxCommitmentKey from explicit generators (PCS::setup derives them with a third-party hash-to-curve,
 * hyrax_pc.rs:152-177 — the caller supplies them) + precompute_ck (:179-190): uploads the bases and builds the
 * 8-bit FixedBaseMul table of h on the device (msm.rs:653-689). */
int sp_ck_create(sp_ctx* ctx, const uint64_t* ck_aff, size_t num_cols, const uint64_t h_aff[8], sp_ck** out);
void sp_ck_free(sp_ck* ck);
/* PCS::commit (:207-303) on n elements of a device table starting at `off`; one Aff per row of num_cols */
int sp_hyrax_commit(sp_ctx* ctx, const sp_ck* ck, const sp_table* v, size_t off, size_t n, const uint64_t* blinds, int is_small, uint64_t* out_rows_aff);
/* PCS::commit_without_blind (:533-568): the per-row MSMs alone ((0,0) for an all-zero row) — what SpartanZkSNARK caches across proves
 * (cached_rest_msm, src/spartan_zk.rs:335-366) */
int sp_hyrax_commit_without_blind(sp_ctx* ctx, const sp_ck* ck, const sp_table* v, size_t off, size_t n, int is_small, uint64_t* out_rows_aff);
/* PCS::commit_incremental (:570-607): out[i] = raw[i] + MSM(row i of delta) + h * blind[i]; raw rows beyond nraw count as the identity; delta = the
 * change of the committed vector since `raw` was computed (n elements of a device table at `off`), all-zero rows cost only the blind term */
int sp_hyrax_commit_incremental(sp_ctx* ctx, const sp_ck* ck, const uint64_t* raw_rows_aff, size_t nraw, const sp_table* delta, size_t off, size_t n,
                                const uint64_t* blinds, uint64_t* out_rows_aff);
/* PCS::commit_zeros (:305-319) and the per-row h * blind of rerandomize (:321-344): FixedBaseMul::mul (msm.rs:691-725) */
int sp_fixed_base_mul_h(sp_ctx* ctx, const sp_ck* ck, const uint64_t* scalars, size_t n, uint64_t* out_aff);
/* PCS::rerandomize_commitment (hyrax_pc.rs:321-344): out[i] = comm[i] + h * (r_new[i] - r_old[i]) (FixedBaseMul::mul per row) */
int sp_hyrax_rerandomize(sp_ctx* ctx, const sp_ck* ck, const uint64_t* comm_rows_aff, size_t rows, const uint64_t* r_old, const uint64_t* r_new, uint64_t* out_rows_aff);
/* HyraxPCS::prove (src/provider/pcs/hyrax_pc.rs:387-478) with InnerProductArgumentLinear::prove (src/provider/pcs/ipa.rs:125-170) inside, the trait
 * method SpartanSNARK::prove calls at src/spartan.rs:425-435: comm = `rows` affine row commitments of poly (a device-resident table of n = 2^npt
 * elements, rows x cols), blinds = the rows' blinds, point = the evaluation point (row variables first), comm_eval / blind_eval = the commitment to the
 * claimed evaluation under ck_eval (narrow key) and its blind. The IPA's randomness is an input (injected randomness) in the form the
 * reference consumes it: `rng` = a stream of 64-byte uniform blocks, one per E::Scalar::random call in draw order — the mask vector d (cols blocks,
 * ipa.rs:139-145), then the blinds of delta and beta (:146-149) — each reduced as from_uniform (src/provider/traits.rs:275-280); cols + 2 blocks are
 * consumed. Absorbs / squeezes on `tr` exactly as the reference does. out = delta (8 words, affine) | beta (8) | z_vec (4 * cols) | z_delta (4) |
 * z_beta (4): the InnerProductArgumentLinear fields. */
int sp_hyrax_prove(sp_ctx* ctx, const sp_ck* ck, const sp_ck* ck_eval, sp_transcript* tr, const uint64_t* comm_rows_aff, size_t rows, const sp_table* poly, size_t n,
                   const uint64_t* blinds, const uint64_t* point, size_t npt, const uint64_t comm_eval_aff[8], const uint64_t blind_eval[4], const uint8_t* rng,
                   size_t rng_blocks, uint64_t* out);
/* PCS::prove announced ahead of its call. src/spartan.rs calls PCS::prove last (:425-435), but the commitment and its blinds exist when
 * r1cs_instance_and_witness returns (:238-245), the IPA's randomness is independent of everything (the reference draws it inside
 * InnerProductArgumentLinear::prove, ipa.rs:139-149) and the ROW half of the evaluation point exists once the inner sum-check has drawn it. A caller that
 * announces the opening here (the shim's r1cs_instance_and_witness wrapper, with the randomness blocks it will later hand to sp_hyrax_prove) lets the
 * library start under the two sum-checks what sp_hyrax_prove would start behind them: the commitment's transcript encoding + Keccak blocks and the mask
 * vector's reductions (helper thread), delta's table walk (auxiliary stream), and — when sp_sumcheck_quad on this context reports the row challenges —
 * L^T W and comm_LZ's walk. sp_hyrax_prove compares what it is given with what was announced (key, table, commitment, blinds, randomness, row
 * challenges) and then only collects; on any difference the announcement is dropped and everything is computed as without it. Optional, one per context,
 * consumed by the next sp_hyrax_prove; no proof value depends on it. A key without window tables (narrow keys, SPARTAN_KEY_TABLES=0) ignores it. */
int sp_hyrax_prove_announce(sp_ctx* ctx, const sp_ck* ck, const uint64_t* comm_rows_aff, size_t rows, const sp_table* poly, size_t n, const uint64_t* blinds,
                            const uint8_t* rng, size_t rng_blocks);
/* The same when the caller holds FixedBaseMul tables (sp_fbtables_create, msm.rs:653-689) of the first `nfixed` commitment rows followed by one of the key's h
 * - a shim builds them in prep_prove, where the precommitted rows are committed (src/spartan.rs:173-217), and keeps them in its PrepSNARK - and every row
 * from `nfixed` on is blind_i * h (commit_zeros, hyrax_pc.rs:230-300: no rest variables): comm_LZ = sum_{i < nfixed} L_i comm[i] + (sum_{i >= nfixed}
 * L_i blind_i) h is then one walk over those tables that needs eq(r_rows, .) alone, so it starts right behind the last row challenge instead of behind
 * L^T W. row_tables->n must be nfixed + 1; the tables must stay alive like the key. The result is the same group element (the rows ARE the
 * commitments of W's rows); sp_hyrax_prove's checks are those of sp_hyrax_prove_announce. */
int sp_hyrax_prove_announce_tables(sp_ctx* ctx, const sp_ck* ck, const uint64_t* comm_rows_aff, size_t rows, const sp_table* poly, size_t n, const uint64_t* blinds,
                                   const uint8_t* rng, size_t rng_blocks, const sp_fbtables* row_tables, size_t nfixed);
/* withdraws an announcement that will not be followed by its sp_hyrax_prove (an error exit of the caller's prove): waits for what it started, frees it.
 * The announced table and key must stay alive until the announcement is consumed, replaced or retracted. */
int sp_hyrax_prove_retract(sp_ctx* ctx);
/* asynchronous form: begin() enqueues upload + kernel + download and returns, finish() waits and normalises. One job per context at a time: the jobs
 * share the context's landing area, so begin() fails with SP_ERR_INVALID_INPUT_LENGTH while an earlier job (n above the host threshold) has not been
 * finished; finish() consumes the job whatever it returns. */
typedef struct sp_fb_job sp_fb_job;
int sp_fixed_base_mul_h_begin(sp_ctx* ctx, const sp_ck* ck, const uint64_t* scalars, size_t n, sp_fb_job** job);
int sp_fixed_base_mul_h_finish(sp_ctx* ctx, sp_fb_job* job, uint64_t* out_aff);
/* bind_with_delayed (:38-54): out[i] = sum_j L[j] * poly[j*cols + i]; out has `cols` F on the host */
int sp_rowmat_vec(sp_ctx* ctx, const sp_table* poly, size_t rows, size_t cols, const uint64_t* L, uint64_t* out);
/* vartime_multiscalar_mul(scalars, ck[..n]) + h * blind against a device-resident key (hyrax_pc.rs:454-455, ipa.rs:147);
 * blind may be NULL for the bare MSM */
int sp_msm_ck(sp_ctx* ctx, const sp_ck* ck, const uint64_t* scalars, size_t n, const uint64_t* blind, uint64_t out_aff[8]);
/* The same MSM split in two so it can overlap other work (the reference gets this overlap from rayon): begin() enqueues the
 * device part on the context's auxiliary stream and returns; finish() waits, adds h * blind (may be NULL) and frees the job. */
typedef struct sp_msm_job sp_msm_job;
/* comm_LZ of HyraxPCS::prove (hyrax_pc.rs:430-455) is commit(L . W; <L, r_W>) with L = eq(point[..nvr]); by the homomorphism of the Pedersen
 * commitment it equals sum_i L[i] * comm_W[i] — the same group element, so the same affine bytes — an MSM over the ROW COMMITMENTS that needs
 * only the challenges of the row variables, which the inner sum-check has produced half-way through its rounds. sp_points_upload puts a
 * point vector (x || y Montgomery limbs, (0, 0) = identity) on the device, reusing *io when it is large enough; sp_msm_eq_begin enqueues
 * sum_i eq(r, i) * pts[i] (2^ell points, r[0] on the index MSB as EqPolynomial::evals_from_points, src/polys/eq.rs:59-93) on the auxiliary
 * stream; sp_msm_job_finish = sp_msm_ck_finish without a blind. These three touch only the auxiliary stream's state, so ONE other thread may
 * call them while the context's owner runs a sum-check on the main stream, provided no other *_begin / *_finish call overlaps them. */
typedef struct sp_points sp_points;
int sp_points_upload(sp_ctx* ctx, const uint64_t* aff, size_t n, sp_points** io);
void sp_points_free(sp_points* p);
/* DlogGroupExt::vartime_multiscalar_mul (msm.rs:187-222) on operands resident in HBM: sum of scalars[off + i] * points[first + i], i < n — what a caller
 * that keeps its bases on the device between calls uses, and what the point-range sharding of SURVEY 8(e) runs per rank on its range of the points
 * (the ranks' affine partial sums are then gathered and added). From 4096 points up the multi-block Pippenger (kernels_pippenger.hpp) with the
 * library's window width; window = 8 / 10 / 12 / 13 / 14 forces that path and width (tests, measurements), 0 = the library's choice. */
int sp_msm_points(sp_ctx* ctx, const sp_table* scalars, size_t off, size_t n, const sp_points* bases, size_t first, int window, uint64_t out_aff[8]);
/* the window width the library picks for an n-point MSM (the reference's rule is c = ceil(ln n), msm.rs:201-205; same trade, this kernel's constants) */
int sp_msm_pippenger_window(size_t n);
int sp_msm_eq_begin(sp_ctx* ctx, const sp_points* pts, const uint64_t* r, size_t ell, sp_msm_job** job);
int sp_msm_job_finish(sp_ctx* ctx, sp_msm_job* job, uint64_t out_aff[8]);
/* bind_with_delayed (hyrax_pc.rs:38-54) with L = eq(r, .) (ell <= 20 row variables; up to 10 formed on the device, more uploaded), on a stream of its own, so that LZ —
 * needed only for the IPA's z vector — is computed beside comm_LZ's MSM rather than before it. One job at a time; same threading rule as above. */
typedef struct sp_vec_job sp_vec_job;
int sp_rowmat_vec_eq_begin(sp_ctx* ctx, const sp_table* poly, const uint64_t* r, size_t ell, size_t cols, sp_vec_job** job);
/* The same with the IPA's mask along: `addend` (cols elements, or NULL) is uploaded behind the product, and _finish_scaled returns scale * LZ + addend
 * - z_vec = r * LZ + d of InnerProductArgumentLinear::prove (ipa.rs:160-163), the last step of a prove - formed on the device and delivered through
 * mapped memory, instead of LZ for the caller to scale (2048 products on the host). A job takes ONE of the two finishes. */
int sp_rowmat_vec_eq_begin_with(sp_ctx* ctx, const sp_table* poly, const uint64_t* r, size_t ell, size_t cols, const uint64_t* addend, sp_vec_job** job);
int sp_rowmat_vec_eq_finish_scaled(sp_ctx* ctx, sp_vec_job* job, const uint64_t scale[4], uint64_t* out);
int sp_rowmat_vec_eq_finish(sp_ctx* ctx, sp_vec_job* job, uint64_t* out);
int sp_msm_ck_begin(sp_ctx* ctx, const sp_ck* ck, const uint64_t* scalars, size_t n, sp_msm_job** job);
int sp_msm_ck_finish(sp_ctx* ctx, const sp_ck* ck, sp_msm_job* job, const uint64_t* blind, uint64_t out_aff[8]);
/* the same over the bases [first, first + n) of the key: a rank's point range of a sharded MSM; finish with sp_msm_ck_finish */
int sp_msm_ck_range_begin(sp_ctx* ctx, const sp_ck* ck, const uint64_t* scalars, size_t first, size_t n, sp_msm_job** job);
/* PCS::commit for keys of width <= 64, where the reference uses per-base FixedBaseMul tables (hyrax_pc.rs:221-260,
 * msm.rs:727-773 multi_mul): sum_i scalars[i] * ck[i] + h * blind, host scalars, n <= num_cols <= 64 */
int sp_hyrax_commit_small(sp_ctx* ctx, const sp_ck* ck, const uint64_t* scalars, size_t n, const uint64_t blind[4], uint64_t out_aff[8]);
/* PCS::commit of a HOST vector of n scalars on a narrow key (<= 64 columns; rows x (num_cols + 1) <= 640), one blind a row, the latency form: one launch of
 * the cooperative table walk through mapped memory, the rows' points added by the polling host threads (hyrax_pc.rs:221-260; the cross term of NovaNIFS,
 * src/nifs.rs:34-61, is 512 scalars = 16 rows of the width-32 key and sits in the transcript chain). Same rows as sp_hyrax_commit on a staged table. */
int sp_hyrax_commit_rows_host(sp_ctx* ctx, const sp_ck* ck, const uint64_t* scalars, size_t n, const uint64_t* blinds, uint64_t* out_rows_aff);
/* the same with the blind's term h * blind handed in as an affine point the caller computed beforehand (sp_fixed_base_mul_h: blinds come from the
 * randomness stream and are known long before the scalars); n <= 6; identity = all-zero coordinates */
int sp_hyrax_commit_small_with_term(sp_ctx* ctx, const sp_ck* ck, const uint64_t* scalars, size_t n, const uint64_t blind_term_aff[8], uint64_t out_aff[8]);
/* PCS::commit on a narrow key in TWO calls (round 6): the commitment of a round of the ZK verifier circuit (src/bellpepper/r1cs.rs:735-816
 * process_round -> PCS::commit, hyrax_pc.rs:221-260 -> FixedBaseMul::multi_mul, msm.rs:727-773) is sum_i row[i] * ck[i] + blind * h, linear in the
 * row, and most of a round's row does not depend on the round's own prover message: the Horner steps of the PREVIOUS polynomial at its challenge and
 * the blind are known one device round earlier. _begin posts those terms (columns `cols[i]` with `scalars[i]`, and `blind` unless NULL) to the
 * process's table walkers - polling host threads that add 16-bit-window table entries (SPARTAN_WALKERS, default 8; spartan2_amd/csrc/walk_pool.hpp
 * says why this one commitment is host work) - and returns at once; _finish adds the remaining terms, walked by the caller and the walkers together,
 * and returns the affine commitment. The result equals sp_hyrax_commit_small on the assembled row for any split. Columns must lie below the number
 * sp_hyrax_commit_split_available was asked about (the key keeps host tables of its first 16 columns and of h). One _finish or _drop per _begin. */
typedef struct sp_split_commit sp_split_commit;
int sp_walkers(void);                              /* polling walker threads of this process (0: the split form is not offered) */
int sp_walkers_keep_hot(uint64_t microseconds);    /* a prove starts: wake the walkers and keep them polling for this long */
/* fn(arg, part, nparts) for part = 0 .. nparts - 1 on the walkers and the calling thread; returns when every part has run (parts nobody claims are the
 * caller's). What `par_iter` is to the reference's host loops over a few hundred to a few thousand field elements (the verifier-circuit instance of
 * NeutronNovaZkSNARK::prove, src/neutronnova_zk.rs:1948-2017: multiply_vec, the NovaNIFS cross term, the folds). nparts <= 32. */
typedef void (*sp_part_fn)(void* arg, unsigned part, unsigned nparts);
int sp_host_parallel_for(unsigned nparts, sp_part_fn fn, void* arg);
int sp_hyrax_commit_split_available(const sp_ck* ck, size_t cols_used);
int sp_hyrax_commit_split_begin(sp_ctx* ctx, const sp_ck* ck, const uint32_t* cols, const uint64_t* scalars, size_t n, const uint64_t* blind, sp_split_commit** job);
int sp_hyrax_commit_split_finish(sp_ctx* ctx, sp_split_commit* job, const uint32_t* cols, const uint64_t* scalars, size_t n, uint64_t out_aff[8]);
void sp_hyrax_commit_split_drop(sp_split_commit* job);

/* ---- sum-checks on a table slice ("sum-check by evaluation-table slice, one reduce per round") -----------------------------
 * Tables sharded on their LAST k variables: rank g holds Z_g[j] = Z[(j << k) | g], so the pairs (i, i + n/2) of the first ell - k rounds are
 * rank-local. Each rank calls the _sharded form on its slice with taus[0 .. ell-k): per round the slice's sums are multiplied by
 * scale = eq(taus[ell-k .. ell), bits of g) (cubic only) and handed to `reduce` (user, sums, count) which must return the field sum over all
 * ranks in place (RCCL has no modular-add reduction: all-gather + local adds; ~100 B per round). Every rank then derives the same round
 * polynomial and runs the transcript redundantly. claim_io / p_io carry the running claim and the eq(tau, r) product across calls: after the
 * local rounds the ranks gather their final (A, B, C) values into 2^k-element tables and EVERY rank finishes the last k rounds with a second,
 * unsharded call (scale = NULL, reduce = NULL) on taus[ell-k .. ell) continuing from claim_io / p_io. */
typedef int (*sp_reduce_hook)(void* user, uint64_t* sums, size_t count);
int sp_sumcheck_cubic3_sharded(sp_ctx* ctx, uint64_t claim_io[4], uint64_t p_io[4], const uint64_t* taus, size_t ell, sp_table* A, sp_table* B, sp_table* C,
                               sp_transcript* tr, const uint64_t* scale, sp_reduce_hook reduce, void* reduce_user, uint64_t* out_cpolys, uint64_t* out_r,
                               uint64_t out_final[12]);
int sp_sumcheck_quad_sharded(sp_ctx* ctx, uint64_t claim_io[4], size_t rounds, sp_table* A, sp_table* B, sp_transcript* tr, sp_reduce_hook reduce,
                             void* reduce_user, uint64_t* out_cpolys, uint64_t* out_r, uint64_t out_final[8]);
/* The first run_rounds rounds only (0 < run_rounds < ell / rounds): the tables are left bound to 2^(ell - run_rounds) elements and the running claim
 * (and eq product) go out through claim_io / p_io. A sharded prover exchanges sums only while its slice is large - the rounds where sharding pays -
 * then gathers the slices into tables of 2^(ell - run_rounds + k) elements (sp_table_scatter_strided) and finishes with an ordinary call on every
 * rank, instead of one exchange in every round. */
/* the slice form with the round-0 products of the slice's own pairs (sp_multiply_vec_incremental_round0 on the row-slice shape; src/sumcheck.rs:1041-1105 first
 * evaluation); run_rounds = 0 or ell: every round, else the first run_rounds only */
int sp_sumcheck_cubic3_sharded_round0(sp_ctx* ctx, uint64_t claim_io[4], uint64_t p_io[4], const uint64_t* taus, size_t ell, size_t run_rounds, sp_table* A, sp_table* B,
                                      sp_table* C, const sp_table* p0, const sp_table* p1, sp_transcript* tr, const uint64_t* scale, sp_reduce_hook reduce,
                                      void* reduce_user, uint64_t* out_cpolys, uint64_t* out_r, uint64_t out_final[12]);
int sp_sumcheck_cubic3_sharded_partial(sp_ctx* ctx, uint64_t claim_io[4], uint64_t p_io[4], const uint64_t* taus, size_t ell, size_t run_rounds, sp_table* A, sp_table* B,
                                       sp_table* C, sp_transcript* tr, const uint64_t* scale, sp_reduce_hook reduce, void* reduce_user, uint64_t* out_cpolys,
                                       uint64_t* out_r);
int sp_sumcheck_quad_sharded_partial(sp_ctx* ctx, uint64_t claim_io[4], size_t rounds, size_t run_rounds, sp_table* A, sp_table* B, sp_transcript* tr,
                                     sp_reduce_hook reduce, void* reduce_user, sp_challenge_hook observe, void* observe_user, uint64_t* out_cpolys, uint64_t* out_r);
/* the same with sp_sumcheck_quad_observed's hook: a sharded HyraxPCS::prove starts its rank's part of comm_LZ as soon as the row challenges exist */
int sp_sumcheck_quad_sharded_observed(sp_ctx* ctx, uint64_t claim_io[4], size_t rounds, sp_table* A, sp_table* B, sp_transcript* tr, sp_reduce_hook reduce,
                                      void* reduce_user, sp_challenge_hook observe, void* observe_user, uint64_t* out_cpolys, uint64_t* out_r, uint64_t out_final[8]);

/* sp_sumcheck_cubic3 / sp_sumcheck_cubic3_round0 with an observer (see sp_sumcheck_quad_observed): observe(user, round, r) is called after the challenge of
 * `round` (0-based) has been handed to the device. p0 / p1 = the round-0 product tables or both NULL. */
int sp_sumcheck_cubic3_observed(sp_ctx* ctx, const uint64_t claim[4], const uint64_t* taus, size_t ell, sp_table* A, sp_table* B, sp_table* C, const sp_table* p0,
                                const sp_table* p1, sp_transcript* tr, sp_challenge_hook observe, void* user, uint64_t* out_cpolys, uint64_t* out_r,
                                uint64_t out_final[12]);

/* ---- NeutronNova batched ZK sum-checks (src/sumcheck.rs:702-917) --------------------------------------------------------------------
 * The reference obtains each round's challenge from the ZK verifier circuit (`SatisfyingAssignment::process_round`, :747-755, :864-872) —
 * a commit + transcript step that stays with the caller. It enters as a callback: hook(user, round, coeffs_step, coeffs_core, ncoeffs,
 * r_out) receives the UniPoly coefficients (degree order, ncoeffs = 3 or 4) of both branches and writes the challenge; a non-zero
 * return aborts the sum-check with that code. */
typedef int (*sp_round_hook)(void* user, size_t round, const uint64_t* coeffs_step, const uint64_t* coeffs_core, size_t ncoeffs, uint64_t r_out[4]);
/* prove_quad_batched_zk (:702-782). claims = {step, core}; tables are bound in place; out_final = {A0[0], A1[0], B0[0], B1[0]} */
int sp_sumcheck_quad_batched(sp_ctx* ctx, const uint64_t claims[8], size_t num_rounds, sp_table* A0, sp_table* A1, sp_table* B0, sp_table* B1, size_t start_round,
                             sp_round_hook hook, void* user, uint64_t* out_r, uint64_t out_final[16]);
/* prove_cubic_with_additive_term_batched_zk (:786-917). pow_left / pow_right = PowPolynomial::split_evals halves as tables; the step and
 * core tables are bound in place (their element 0 holds the final claims); element 0 of pow_left receives base_tau (:913). */
int sp_sumcheck_cubic_outer_pow_batched(sp_ctx* ctx, size_t num_rounds, sp_table* pow_left, const sp_table* pow_right, sp_table* A_step, sp_table* B_step,
                                        sp_table* C_step, sp_table* A_core, sp_table* B_core, sp_table* C_core, const uint64_t t_out_step[4], size_t start_round,
                                        sp_round_hook hook, void* user, uint64_t* out_r);

/* ---- NeutronNova NIFS rounds (src/neutronnova_zk.rs:511-1273, NeutronNovaNIFS::prove) ---------------------
 * The layers Az_b, Bz_b, Cz_b of the n_padded (power of two) instances live in three contiguous device arrays [layer][left*right];
 * sp_nifs_layer hands out non-owning sp_table views so sp_multiply_vec can write a layer in place (:576-596). Per round the reference
 * computes (e0, quad_coeff) over all instance pairs (:779-1097), builds the cubic in `finish_round!` (:703-735), obtains r_b from the
 * verifier-circuit `process_round` (which stays on the caller's side of the ABI: it is a transcript/commit step, not data-parallel
 * work), and folds the layers with r_b merged into the next round. The calls map one to one:
 *   sp_nifs_begin      E_eq = PowPolynomial::split_evals(tau) (left | right entries, :563-566), rhos (:568-571); resets the state and
 *                      computes c_vals[b] = sum_k E[k] Cz_b[k] (:652-703). small_values != 0 additionally builds the i64 mirrors
 *                      (to_small_vec_or_zero, src/big_num/small_value.rs:41-86, global large positions as :1548-1586) and takes
 *                      round 0 / c_vals from them (prove_helper_small :255-320) — same values, a quarter of the bytes.
 *   sp_nifs_round      (t) -> the four coefficients [d, c, b, a] of poly_t (:719-721)
 *   sp_nifs_challenge  (r_b): acc_eq *= eq(r_b, rho_t), T_cur = poly_t(r_b) (:729-731)
 *   sp_nifs_finish     final fold of A, B (:1122-1165), Cz = sum_b w_b Cz_b with w = weights_from_r(r_bs) (:1168-1203),
 *                      T_out = T_cur / acc_eq (:1205-1206, DivisionByZero when acc_eq = 0), eq_rho_at_rb = acc_eq.
 * Witness / instance folding after the rounds is sp_fold_tables, sp_msm_shared_weights and sp_fixed_base_mul_h. */
typedef struct sp_nifs sp_nifs;
int sp_nifs_create(sp_ctx* ctx, size_t n_padded, size_t left, size_t right, sp_nifs** out);
void sp_nifs_free(sp_nifs* n);
/* which: 0 = Az, 1 = Bz, 2 = Cz. The view is valid until sp_nifs_free; free it with sp_table_free (does not release the storage). */
int sp_nifs_layer(sp_nifs* n, int which, size_t idx, sp_table** view);
/* small_values: 0 = field layers only; 1 = build the i64 mirrors now; 2 = use the mirrors sp_nifs_prepare_small built on these layers (the
 * reference's split: cached_step_i64 is made in prep_prove, src/neutronnova_zk.rs:1548-1586, and only consumed by prove) */
int sp_nifs_begin(sp_nifs* n, const uint64_t* E_eq, const uint64_t* rhos, size_t ell_b, int small_values);
/* The rounds only READ the instance layers and their mirrors (folds go to storage of their own), so an object whose layers were written once -
 * prep_prove's cached_step_matvec / cached_step_i64 (:1520-1590) - serves any number of proves: each starts with sp_nifs_begin. */
int sp_nifs_prepare_small(sp_nifs* n);
int sp_nifs_round(sp_nifs* n, size_t t, uint64_t out_coeffs[16]);
int sp_nifs_challenge(sp_nifs* n, const uint64_t r_b[4]);
int sp_nifs_finish(sp_nifs* n, sp_table* A_out, sp_table* B_out, sp_table* C_out, uint64_t out_T_out[4], uint64_t out_eq_rho_at_rb[4]);
/* Sharded batches (instances / N per GPU, one process per GPU). Each rank holds an aligned block of the 2^ell_b instances:
 *   sp_nifs_begin_shard     like sp_nifs_begin for instances [first_instance, first_instance + n_padded) of the batch
 *   sp_nifs_cvals / _set_cvals   the shard's c_vals out / the all-gathered batch-wide vector in
 *   sp_nifs_round_sums      this shard's part of (e0, quad_coeff) of round t (pair weights use the batch-wide pair index); the caller adds
 *                           the parts of all ranks (field additions, any order) and hands the totals to
 *   sp_nifs_round_finish    the `finish_round!` algebra -> the four coefficients (sp_nifs_round = both calls in one)
 *   after log2(n_padded) rounds a shard has no local pair left:
 *   sp_nifs_fold_pending    applies the last challenge's fold; sp_nifs_current_layer then exposes the single remaining A / B layer, which the
 *                           ranks gather on one of them (the only bulk exchange of the path: 2 layers per rank, once);
 *   sp_nifs_resume          on that rank: a fresh sp_nifs with one layer per rank continues at round t_start with the state of sp_nifs_state.
 * sp_nifs_finish(C_out = NULL) skips the C fold; each rank folds its C layers with its slice of weights_from_r (sp_fold_tables). */
int sp_nifs_begin_shard(sp_nifs* n, const uint64_t* E_eq, const uint64_t* rhos, size_t ell_b, size_t first_instance, int small_values);
int sp_nifs_cvals(const sp_nifs* n, uint64_t* out_local);
int sp_nifs_set_cvals(sp_nifs* n, const uint64_t* all, size_t count);
int sp_nifs_round_sums(sp_nifs* n, size_t t, uint64_t out_sums[8]);
int sp_nifs_round_finish(sp_nifs* n, size_t t, const uint64_t sums[8], uint64_t out_coeffs[16]);
int sp_nifs_fold_pending(sp_nifs* n);
int sp_nifs_current_layer(sp_nifs* n, int which, size_t idx, sp_table** view);
int sp_nifs_state(const sp_nifs* n, uint64_t out_T_cur[4], uint64_t out_acc_eq[4]);
int sp_nifs_resume(sp_nifs* n, const uint64_t* E_eq, const uint64_t* rhos, size_t ell_b, size_t t_start, const uint64_t* r_bs, const uint64_t T_cur[4],
                   const uint64_t acc_eq[4], const uint64_t* c_vals_all);
/* to_small_vec_or_zero (src/big_num/small_value.rs:41-86) of a resident table: out_i64[cnt] and out_large[cnt] (0/1) on the host */
int sp_to_small_vec_or_zero(sp_ctx* ctx, const sp_table* t, size_t cnt, int64_t* out_i64, uint8_t* out_large);
/* PowPolynomial::split_evals (src/polys/power.rs:64-87), host side: left | right entries */
int sp_pow_split_evals(const uint64_t tau[4], size_t ell, size_t left, size_t right, uint64_t* out);

/* ---- wire formats and key digests (src/digest.rs:22-77) ---------------------------------- */
/* The reference serialises keys and proofs with serde + bincode `DefaultOptions::new().with_little_endian().with_fixint_encoding()` (src/digest.rs:33-41)
 * and digests keys with SHA-256 over `write_bytes` (DigestComputer, :49-77). Framing: usize = 8 bytes LE; Vec<T> = u64 length + elements; Option<T> = one
 * tag byte + T; structs / tuples = their fields in order. Third-party element layouts (halo2curves `derive_serde`, Cargo.toml:41-46 — the one documented
 * assumption): F = its 32 `to_repr()` bytes (canonical, little-endian; >= p is rejected on read); affine point = {x, y}; projective point (`E::GE`) =
 * {x, y, z}, written normalised as (x, y, 1) / (0, 0, 0) for the identity, read in any Jacobian representative. Host code only. */
int sp_sha256(const uint8_t* data, size_t n, uint8_t out[32]);
int sp_sha256_accelerated(void); /* 1 when the x86 SHA extensions are in use */
/* byte sink in the role of the `io::Write` of Digestible::write_bytes (:24-27): hashing = 0 collects the bytes, 1 streams them into SHA-256 */
typedef struct sp_wire sp_wire;
int sp_wire_new(int hashing, sp_wire** out);
void sp_wire_free(sp_wire* w);
int sp_wire_raw(sp_wire* w, const uint8_t* bytes, size_t n);
int sp_wire_u8(sp_wire* w, uint8_t v);                                                /* bool / Option tag */
int sp_wire_u64s(sp_wire* w, const uint64_t* v, size_t n, int with_len);              /* usize values; with_len: as Vec<usize> */
int sp_wire_u32s_as_u64(sp_wire* w, const uint32_t* v, size_t n, int with_len);       /* column indices held as u32, usize on the wire */
int sp_wire_scalars(sp_wire* w, const uint64_t* f, size_t n, int with_len);           /* E::Scalar values / Vec<E::Scalar> */
int sp_wire_affines(sp_wire* w, const uint64_t* aff, size_t n, int with_len);         /* AffineGroupElement {x, y} */
int sp_wire_points(sp_wire* w, const uint64_t* aff, size_t n, int with_len);          /* E::GE from its affine form; with_len: HyraxCommitment { comm } */
/* HyraxCommitmentKey / HyraxVerifierKey { num_cols, ck, h } (src/provider/pcs/hyrax_pc.rs:56-108; the tables are #[serde(skip)]) */
int sp_wire_hyrax_key(sp_wire* w, const uint64_t* ck, size_t num_cols, const uint64_t* h);
/* SparseMatrix: digest_form != 0 = write_digest_bytes (src/r1cs/sparse.rs:398-417), 0 = the derived Serialize (:383-394) */
int sp_wire_matrix(sp_wire* w, const sp_csr* M, size_t rows, size_t cols, int digest_form);
/* SplitR1CSShape: digest_form != 0 = write_bytes (src/r1cs/mod.rs:775-794), 0 = the derived Serialize (:742-773) */
int sp_wire_shape(sp_wire* w, const sp_dims* dims, const sp_csr* A, const sp_csr* B, const sp_csr* C, int digest_form);
size_t sp_wire_len(const sp_wire* w); /* bytes written so far */
int sp_wire_bytes(sp_wire* w, uint8_t* out, size_t cap);
int sp_wire_digest(sp_wire* w, uint8_t out[32]);
/* byte source (borrows `bytes`): what bincode's deserializer does for the same types; every call fails on short input */
typedef struct sp_unwire sp_unwire;
int sp_unwire_new(const uint8_t* bytes, size_t n, sp_unwire** out);
void sp_unwire_free(sp_unwire* r);
size_t sp_unwire_left(const sp_unwire* r);
int sp_unwire_u8(sp_unwire* r, uint8_t* out);
int sp_unwire_u64s(sp_unwire* r, size_t n, uint64_t* out);
int sp_unwire_len(sp_unwire* r, size_t elem_bytes, size_t* out); /* Vec length, refused when the input cannot hold that many elements */
int sp_unwire_scalars(sp_unwire* r, size_t n, uint64_t* out);
int sp_unwire_affines(sp_unwire* r, size_t n, uint64_t* out);    /* on-curve check */
int sp_unwire_points(sp_unwire* r, size_t n, uint64_t* out_aff); /* any representative -> Aff; on-curve check */
int sp_unwire_done(const sp_unwire* r);                          /* fails on trailing bytes, as bincode's DefaultOptions do */
/* DigestHelperTrait::digest of SpartanVerifierKey (src/spartan.rs:73-104): SHA-256(bincode(vk_ee) || bincode(ck_s) || S.write_bytes()).
 * (ck, num_cols, h) = the witness key (vk_ee holds the same three fields), (ck_s, num_cols_s, h_s) = the width-1 key of src/spartan.rs:152. */
int sp_vk_digest(const sp_dims* dims, const sp_csr* A, const sp_csr* B, const sp_csr* C, const uint64_t* ck, size_t num_cols, const uint64_t* h,
                 const uint64_t* ck_s, size_t num_cols_s, const uint64_t* h_s, uint8_t out[32]);
/* SpartanSNARK (src/spartan.rs:125-137) between bincode bytes and this build's flat word layout:
 *   comm_W rows (Aff: shared, precommitted, rest) | public_values | challenges | outer polys (rounds_x x 3 F) | claims_outer (3 F) |
 *   inner polys (rounds_y x 2 F) | eval_W | blind_eval_W | delta | beta (Aff) | z_vec (z_len F) | z_delta | z_beta.
 * A segment's Option<Commitment> is Some exactly when it has rows. `_serialize`: out may be NULL to learn *len. `_deserialize` fills the layout from
 * the bytes' own length prefixes (words may be NULL to learn *nwords); the caller compares it with the key's shape before verifying. */
typedef struct sp_spartan_layout {
  uint64_t rows_shared, rows_precommitted, rows_rest, num_public, num_challenges, rounds_x, rounds_y, z_len;
} sp_spartan_layout;
size_t sp_proof_words(const sp_spartan_layout* layout);
int sp_proof_serialize(const sp_spartan_layout* layout, const uint64_t* words, size_t nwords, uint8_t* out, size_t cap, size_t* len);
int sp_proof_deserialize(const uint8_t* bytes, size_t n, sp_spartan_layout* layout, uint64_t* words, size_t cap_words, size_t* nwords);

#ifdef __cplusplus
}
#endif
#endif
