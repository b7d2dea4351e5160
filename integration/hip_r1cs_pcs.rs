// src/provider/hip_r1cs_pcs.rs — the second half of the reference-side binding (see hip_provider.rs for the first: DlogGroupExt, the HBM-resident
// MultilinearPolynomial, the transcript, the two Spartan sum-checks, HyraxPCS::prove). This file covers what src/neutronnova_zk.rs needs on top:
//
//   * `HipShape`         SplitR1CSShape::{precompute, multiply_vec, multiply_vec_precommitted, multiply_vec_incremental_into,
//                        bind_and_prepare_poly_ABC[_full]}                                      src/r1cs/mod.rs:1059-1270
//   * `HipCommitmentKey` PCSEngineTrait::{commit, commit_zeros, rerandomize_commitment, commit_without_blind, commit_incremental} and
//                        FoldingEngineTrait::{fold_commitments, fold_blinds, fold_commitments_partial}
//                                                                                                src/traits/pcs.rs:32-232, src/provider/pcs/hyrax_pc.rs:207-344, :533-607, :737-874
//   * `HipNifs`          NeutronNovaNIFS::prove's rounds (prove_helper*, the merged fold + prove, c_vals, finish_round!)   src/neutronnova_zk.rs:98-432, :511-1273
//   * `prove_quad_batched_zk` / `prove_cubic_with_additive_term_batched_zk` bodies                src/sumcheck.rs:702-917
//   * `R1CSWitness::fold_multiple`, `weights_from_r`                                             src/r1cs/mod.rs:153-166, :570-660
//   * `digest` / `to_bytes` helpers over the wire sink (bincode framing + SHA-256)               src/digest.rs:22-77
//
// NOT compiled in this repository's image (no rustc). The compiled stand-ins for the call sequences are `prove_reference_order`
// (spartan2_amd/host/spartan_snark.cpp) and `nn_prove(.., reference_order = true)` (spartan2_amd/host/neutronnova_zk.cpp): one thread, these
// same ABI calls in the statement order of src/spartan.rs:226-466 / src/neutronnova_zk.rs:1609-2093, bit-exact against the oracle.
#![allow(non_snake_case)]
use crate::{
  errors::SpartanError,
  provider::{hip_ffi::*, hip_provider::{check, ctx, HipTable, HipTranscript}},
  traits::Engine,
};
use core::ffi::{c_int, c_void};
use std::{marker::PhantomData, ptr};

#[inline]
fn limbs<F>(v: &[F]) -> *const u64 {
  v.as_ptr() as *const u64
}
#[inline]
fn limbs_mut<F>(v: &mut [F]) -> *mut u64 {
  v.as_mut_ptr() as *mut u64
}

// ---- SplitR1CSShape on the device (src/r1cs/mod.rs:743-1270) -------------------------------------------------------------------------------
/// The padded shape's three CSR matrices (`SparseMatrix { data, indices, indptr }`, src/r1cs/sparse.rs:385-394) classified and uploaded once:
/// `SplitR1CSShape::precompute` (:1059-1073). `indices` are narrowed to u32 on the way (columns < 2^32).
pub struct HipShape<E: Engine> {
  pub(crate) s: *mut sp_shape,
  pub num_cons: usize,
  pub num_vars: usize,
  pub num_extra: usize, // 1 + num_public + num_challenges
  _p: PhantomData<E>,
}
unsafe impl<E: Engine> Send for HipShape<E> {}
unsafe impl<E: Engine> Sync for HipShape<E> {}

pub struct CsrRef<'a, F> {
  pub data: &'a [F],
  pub indices: &'a [usize],
  pub indptr: &'a [usize],
}

impl<E: Engine> HipShape<E> {
  pub fn precompute(dims: &sp_dims, a: CsrRef<E::Scalar>, b: CsrRef<E::Scalar>, c: CsrRef<E::Scalar>) -> Result<Self, SpartanError> {
    let narrow = |m: &CsrRef<E::Scalar>| -> (Vec<u32>, Vec<u64>) {
      (m.indices.iter().map(|&i| i as u32).collect(), m.indptr.iter().map(|&p| p as u64).collect())
    };
    let (ai, ap) = narrow(&a);
    let (bi, bp) = narrow(&b);
    let (ci, cp) = narrow(&c);
    let ca = sp_csr { data: limbs(a.data), indices: ai.as_ptr(), indptr: ap.as_ptr() };
    let cb = sp_csr { data: limbs(b.data), indices: bi.as_ptr(), indptr: bp.as_ptr() };
    let cc = sp_csr { data: limbs(c.data), indices: ci.as_ptr(), indptr: cp.as_ptr() };
    let mut s = ptr::null_mut();
    check(unsafe { sp_shape_from_csr(ctx(), &ca, &cb, &cc, dims, &mut s) })?;
    let num_vars = (dims.num_shared + dims.num_precommitted + dims.num_rest) as usize;
    Ok(Self { s, num_cons: dims.num_cons as usize, num_vars, num_extra: (1 + dims.num_public + dims.num_challenges) as usize, _p: PhantomData })
  }
  fn outputs(&self) -> Result<[HipTable<E::Scalar>; 3], SpartanError>
  where
    E::Scalar: Copy + Default,
  {
    Ok([HipTable::zeros(self.num_cons)?, HipTable::zeros(self.num_cons)?, HipTable::zeros(self.num_cons)?])
  }
  /// multiply_vec (:1075-1107): z = [W | 1 | X | challenges] resident in HBM -> (Az, Bz, Cz)
  pub fn multiply_vec(&self, z: &HipTable<E::Scalar>) -> Result<[HipTable<E::Scalar>; 3], SpartanError>
  where
    E::Scalar: Copy + Default,
  {
    let o = self.outputs()?;
    check(unsafe { sp_multiply_vec(ctx(), self.s, z.t, o[0].t, o[1].t, o[2].t) })?;
    Ok(o)
  }
  /// multiply_vec_incremental_into (:1170-1211): cached products of the shared + precommitted columns + the filtered entries of the rest
  pub fn multiply_vec_incremental_into(
    &self,
    z: &HipTable<E::Scalar>,
    cached: &[HipTable<E::Scalar>; 3],
    out: &mut [HipTable<E::Scalar>; 3],
  ) -> Result<(), SpartanError> {
    check(unsafe { sp_multiply_vec_incremental(ctx(), self.s, z.t, cached[0].t, cached[1].t, cached[2].t, out[0].t, out[1].t, out[2].t) })
  }
  /// bind_and_prepare_poly_ABC (:1235-1244; `full` = the 2 * num_vars form of :1250-1270 that NeutronNova hands to the batched sum-check):
  /// rx = EqPolynomial::evals_from_points(r_x) as a table (HipTable::eq)
  pub fn bind_and_prepare_poly_ABC(&self, rx: &HipTable<E::Scalar>, r: &E::Scalar, full: bool) -> Result<HipTable<E::Scalar>, SpartanError>
  where
    E::Scalar: Copy + Default,
  {
    let out_len = if full { 2 * self.num_vars } else { self.num_vars + self.num_extra };
    let out = HipTable::zeros(2 * self.num_vars)?;
    check(unsafe { sp_poly_abc(ctx(), self.s, rx.t, r as *const _ as *const u64, out_len, out.t) })?;
    Ok(out)
  }
}
impl<E: Engine> Drop for HipShape<E> {
  fn drop(&mut self) {
    unsafe { sp_shape_free(self.s) }
  }
}

impl<F: Copy + Default> HipTable<F> {
  /// Round 6 - the witness as the frontend holds it for `is_small` circuits (machine words, src/bellpepper/r1cs.rs:303-409): written at its padded
  /// offset, the Montgomery form is produced on the device (8 bytes a value over PCIe instead of 32, no host loop over 2^20 values)
  pub fn write_u64(&mut self, off: usize, vals: &[u64]) -> Result<(), SpartanError> {
    check(unsafe { sp_table_write_u64(ctx(), self.t, off, vals.as_ptr(), vals.len()) })
  }
  /// ... and a 0/1 witness packed 8 values a byte (value i = bit i & 7 of byte i >> 3)
  pub fn write_bits(&mut self, off: usize, bits: &[u8], count: usize) -> Result<(), SpartanError> {
    check(unsafe { sp_table_write_bits(ctx(), self.t, off, bits.as_ptr(), count) })
  }
  pub fn zeros(len: usize) -> Result<Self, SpartanError> {
    let mut t = ptr::null_mut();
    check(unsafe { sp_table_zeros(ctx(), len, usize::MAX, usize::MAX, &mut t) })?;
    Ok(Self::from_raw(t))
  }
  /// EqPolynomial::evals_from_points (src/polys/eq.rs:59-92) as a resident table
  pub fn eq(r: &[F]) -> Result<Self, SpartanError> {
    let mut t = ptr::null_mut();
    check(unsafe { sp_eq_table(ctx(), limbs(r), r.len(), &mut t) })?;
    Ok(Self::from_raw(t))
  }
}

// ---- HyraxPCS: commitments and folds (src/provider/pcs/hyrax_pc.rs) ---------------------------------------------------------------------------
/// Points cross as affine x | y limbs (8 words, (0, 0) = identity); `pack` / `unpack` of hip_provider::t256_msm convert to and from `E::GE`.
pub struct HipCommitmentKey {
  pub(crate) k: *mut sp_ck,
  pub num_cols: usize,
}
unsafe impl Send for HipCommitmentKey {}
unsafe impl Sync for HipCommitmentKey {}
impl HipCommitmentKey {
  /// from the generators PCS::setup derived (:152-177) + precompute_ck (:179-190)
  pub fn new(ck_aff: &[u64], h_aff: &[u64; 8]) -> Result<Self, SpartanError> {
    let mut k = ptr::null_mut();
    check(unsafe { sp_ck_create(ctx(), ck_aff.as_ptr(), ck_aff.len() / 8, h_aff.as_ptr(), &mut k) })?;
    Ok(Self { k, num_cols: ck_aff.len() / 8 })
  }
  fn rows(&self, n: usize) -> usize {
    n.div_ceil(self.num_cols)
  }
  /// PCS::commit (:207-303) of v[off .. off + n): one affine point per row
  pub fn commit<F>(&self, v: &HipTable<F>, off: usize, n: usize, blinds: &[F], is_small: bool) -> Result<Vec<u64>, SpartanError> {
    let mut out = vec![0u64; 8 * self.rows(n)];
    check(unsafe { sp_hyrax_commit(ctx(), self.k, v.t, off, n, limbs(blinds), is_small as c_int, out.as_mut_ptr()) })?;
    Ok(out)
  }
  /// PCS::commit_zeros (:305-319): h * blind per row
  pub fn commit_zeros<F>(&self, blinds: &[F]) -> Result<Vec<u64>, SpartanError> {
    let mut out = vec![0u64; 8 * blinds.len()];
    check(unsafe { sp_fixed_base_mul_h(ctx(), self.k, limbs(blinds), blinds.len(), out.as_mut_ptr()) })?;
    Ok(out)
  }
  /// PCS::rerandomize_commitment (:321-344)
  pub fn rerandomize<F>(&self, comm_rows: &[u64], r_old: &[F], r_new: &[F]) -> Result<Vec<u64>, SpartanError> {
    let mut out = vec![0u64; comm_rows.len()];
    check(unsafe { sp_hyrax_rerandomize(ctx(), self.k, comm_rows.as_ptr(), comm_rows.len() / 8, limbs(r_old), limbs(r_new), out.as_mut_ptr()) })?;
    Ok(out)
  }
  /// PCS::commit_without_blind (:533-568) and commit_incremental (:570-607): the cache of src/spartan_zk.rs:335-366
  pub fn commit_without_blind<F>(&self, v: &HipTable<F>, off: usize, n: usize, is_small: bool) -> Result<Vec<u64>, SpartanError> {
    let mut out = vec![0u64; 8 * self.rows(n)];
    check(unsafe { sp_hyrax_commit_without_blind(ctx(), self.k, v.t, off, n, is_small as c_int, out.as_mut_ptr()) })?;
    Ok(out)
  }
  pub fn commit_incremental<F>(&self, raw_rows: &[u64], delta: &HipTable<F>, off: usize, n: usize, blinds: &[F]) -> Result<Vec<u64>, SpartanError> {
    let mut out = vec![0u64; 8 * self.rows(n)];
    check(unsafe { sp_hyrax_commit_incremental(ctx(), self.k, raw_rows.as_ptr(), raw_rows.len() / 8, delta.t, off, n, limbs(blinds), out.as_mut_ptr()) })?;
    Ok(out)
  }
  /// Announces the opening that PCS::prove will be asked for at the end of SpartanSNARK::prove (src/spartan.rs:425-435). Called by the shim's
  /// `r1cs_instance_and_witness` wrapper once comm_W and its (combined) blinds exist (:238-245), with the 64-byte randomness blocks that the shim's
  /// OsRng draw will later hand to `hyrax_prove` (cols + 2 of them, ipa.rs:139-149, drawn here instead of there: the draws are independent of
  /// everything else). The library then works on the opening under the two sum-checks; `hyrax_prove` finds it by comparing its arguments.
  /// An `Err` exit of prove between the two calls goes through `retract_opening` (a drop guard in the wrapper).
  pub fn announce_opening<F>(&self, comm_rows: &[u64], poly: &HipTable<F>, n: usize, blinds: &[F], rng: &[u8]) -> Result<(), SpartanError> {
    check(unsafe { sp_hyrax_prove_announce(ctx(), self.k, comm_rows.as_ptr(), comm_rows.len() / 8, poly.t, n, limbs(blinds), rng.as_ptr(), rng.len() / 64) })
  }
  /// The same when prep_prove built FixedBaseMul tables of the rows it committed (`RowTables::new` over the shared + precommitted rows of comm_W, then
  /// the key's h; kept in the shim's PrepSNARK) and the remaining rows are commit_zeros rows (no rest variables: each is blind * h,
  /// hyrax_pc.rs:230-300): comm_LZ = <L, comm_W> is then one walk over those tables behind the last row challenge, independent of L^T W.
  pub fn announce_opening_with_tables<F>(&self, comm_rows: &[u64], poly: &HipTable<F>, n: usize, blinds: &[F], rng: &[u8], tables: &RowTables, nfixed: usize) -> Result<(), SpartanError> {
    check(unsafe {
      sp_hyrax_prove_announce_tables(ctx(), self.k, comm_rows.as_ptr(), comm_rows.len() / 8, poly.t, n, limbs(blinds), rng.as_ptr(), rng.len() / 64, tables.t, nfixed)
    })
  }
  pub fn retract_opening() -> Result<(), SpartanError> {
    check(unsafe { sp_hyrax_prove_retract(ctx()) })
  }
}
/// FixedBaseMul::precompute over a list of points (msm.rs:653-689): the committed rows of comm_W followed by h, built once in prep_prove.
/// Round 6 - a narrow commitment in two calls (src/bellpepper/r1cs.rs:735-816 process_round -> PCS::commit on the width-32 key, hyrax_pc.rs:221-260): the terms
/// that do not depend on the round's own prover message are posted when the previous challenge is drawn and walked by the library's polling host threads
/// under the device's round; `finish` adds the rest. `process_round` keeps the job in its MultiRoundState.
pub struct SplitCommit {
  job: *mut sp_split_commit,
}
impl HipCommitmentKey {
  pub fn split_available(&self, cols_used: usize) -> bool {
    unsafe { sp_hyrax_commit_split_available(self.k, cols_used) != 0 }
  }
  pub fn commit_split_begin<F>(&self, cols: &[u32], scalars: &[F], blind: Option<&F>) -> Result<SplitCommit, SpartanError> {
    let mut job = std::ptr::null_mut();
    let b = blind.map_or(std::ptr::null(), |x| x as *const F as *const u64);
    check(unsafe { sp_hyrax_commit_split_begin(ctx(), self.k, cols.as_ptr(), scalars.as_ptr() as *const u64, cols.len(), b, &mut job) })?;
    Ok(SplitCommit { job })
  }
}
impl SplitCommit {
  pub fn finish<F>(mut self, cols: &[u32], scalars: &[F]) -> Result<[u64; 8], SpartanError> {
    let mut out = [0u64; 8];
    let job = std::mem::replace(&mut self.job, std::ptr::null_mut());
    check(unsafe { sp_hyrax_commit_split_finish(ctx(), job, cols.as_ptr(), scalars.as_ptr() as *const u64, cols.len(), out.as_mut_ptr()) })?;
    Ok(out)
  }
}
impl Drop for SplitCommit {
  fn drop(&mut self) {
    unsafe { sp_hyrax_commit_split_drop(self.job) }  // (null after finish: a no-op)
  }
}
/// Round 6 - fold_commitments with weights (1, w) where w is drawn late (src/neutronnova_zk.rs:2019-2051): the doubling ladders of q's rows are built ahead
pub struct Fold2 {
  job: *mut sp_fold2_job,
}
impl Fold2 {
  pub fn begin(q_rows_aff: &[u64]) -> Result<Self, SpartanError> {
    let mut job = std::ptr::null_mut();
    check(unsafe { sp_fold_commitments2_begin(ctx(), q_rows_aff.as_ptr(), q_rows_aff.len() / 8, &mut job) })?;
    Ok(Fold2 { job })
  }
  pub fn finish<F>(mut self, p_rows_aff: &[u64], w: &F) -> Result<Vec<u64>, SpartanError> {
    let mut out = vec![0u64; p_rows_aff.len()];
    let job = std::mem::replace(&mut self.job, std::ptr::null_mut());
    check(unsafe { sp_fold_commitments2_finish(ctx(), job, p_rows_aff.as_ptr(), w as *const F as *const u64, out.as_mut_ptr()) })?;
    Ok(out)
  }
}
impl Drop for Fold2 {
  fn drop(&mut self) {
    unsafe { sp_fold_commitments2_drop(self.job) }
  }
}
impl HipCommitmentKey {
  /// PCS::commit of a host vector on a narrow key, the latency form (src/nifs.rs:34-61 comm_T)
  pub fn commit_rows_host<F>(&self, v: &[F], blinds: &[F]) -> Result<Vec<u64>, SpartanError> {
    let mut out = vec![0u64; 8 * blinds.len()];
    check(unsafe { sp_hyrax_commit_rows_host(ctx(), self.k, v.as_ptr() as *const u64, v.len(), blinds.as_ptr() as *const u64, out.as_mut_ptr()) })?;
    Ok(out)
  }
}

/// for the duration of a prove whose round hooks commit through `SplitCommit`: the batched sum-checks may queue a round's launch ahead of the hook
pub struct HostOnlyHooks;
impl HostOnlyHooks {
  pub fn promise() -> Result<Self, SpartanError> {
    check(unsafe { sp_walkers_keep_hot(20_000) })?;
    check(unsafe { sp_ctx_round_hooks_host_only(ctx(), 1) })?;
    Ok(HostOnlyHooks)
  }
}
impl Drop for HostOnlyHooks {
  fn drop(&mut self) {
    unsafe { sp_ctx_round_hooks_host_only(ctx(), 0) };
  }
}

pub struct RowTables {
  pub(crate) t: *mut sp_fbtables,
}
impl RowTables {
  pub fn new(points_aff: &[u64]) -> Result<Self, SpartanError> {
    let mut t = std::ptr::null_mut();
    check(unsafe { sp_fbtables_create(ctx(), points_aff.as_ptr(), points_aff.len() / 8, &mut t) })?;
    Ok(Self { t })
  }
  /// Round 6: the build is only QUEUED (a lowest-priority stream of the context's own; ~3 ms of device time for the 513 rows of a 2^20-variable
  /// witness) - what prep_prove calls, so that it returns without waiting for tables that are first read at the end of the first prove.
  /// `announce_opening_with_tables` takes them when they have landed and the key's own tables otherwise (same proof).
  pub fn new_queued(points_aff: &[u64]) -> Result<Self, SpartanError> {
    let mut t = std::ptr::null_mut();
    check(unsafe { sp_fbtables_create_async(ctx(), points_aff.as_ptr(), points_aff.len() / 8, &mut t) })?;
    Ok(Self { t })
  }
  pub fn ready(&self, wait: bool) -> Result<bool, SpartanError> {
    let r = unsafe { sp_fbtables_ready(self.t, wait as c_int) };
    if r < 0 { check(r)?; }
    Ok(r == 1)
  }
}
impl Drop for RowTables {
  fn drop(&mut self) {
    unsafe { sp_fbtables_free(self.t) }
  }
}
impl Drop for HipCommitmentKey {
  fn drop(&mut self) {
    unsafe { sp_ck_free(self.k) }
  }
}

/// FoldingEngineTrait::fold_commitments (:737-793). `comms[i]` = the rows of commitment i (8 words per row). Two commitments with a unit first weight
/// take the wNAF path of :757-776 (sp_fold_commitments2); otherwise one shared-weights MSM per row (:778-792).
pub fn fold_commitments<F: PartialEq + Copy>(comms: &[&[u64]], weights: &[F], one: &F) -> Result<Vec<u64>, SpartanError> {
  if comms.len() != weights.len() || comms.is_empty() {
    return Err(SpartanError::InvalidInputLength { reason: "fold_commitments: one weight per commitment".into() });
  }
  let rows = comms[0].len() / 8;
  let mut out = vec![0u64; 8 * rows];
  if comms.len() == 2 && weights[0] == *one {
    check(unsafe { sp_fold_commitments2(ctx(), comms[0].as_ptr(), comms[1].as_ptr(), rows, &weights[1] as *const F as *const u64, out.as_mut_ptr()) })?;
    return Ok(out);
  }
  // row-major [row][instance] bases
  let n = comms.len();
  let mut bases = vec![0u64; 8 * rows * n];
  for (i, c) in comms.iter().enumerate() {
    if c.len() != 8 * rows {
      return Err(SpartanError::InvalidInputLength { reason: "fold_commitments: commitments of different lengths".into() });
    }
    for r in 0..rows {
      bases[8 * (r * n + i)..8 * (r * n + i) + 8].copy_from_slice(&c[8 * r..8 * r + 8]);
    }
  }
  check(unsafe { sp_msm_shared_weights(ctx(), limbs(weights), n, bases.as_ptr(), rows, out.as_mut_ptr()) })?;
  Ok(out)
}
/// fold_commitments_partial (:795-874): data rows as above, the rows beyond `num_data_rows` as h * folded_blind[row]
pub fn fold_commitments_partial<F: PartialEq + Copy>(
  comms: &[&[u64]],
  weights: &[F],
  one: &F,
  num_data_rows: usize,
  folded_blind: &[F],
  ck: &HipCommitmentKey,
) -> Result<Vec<u64>, SpartanError> {
  let data: Vec<&[u64]> = comms.iter().map(|c| &c[..8 * num_data_rows]).collect();
  let mut out = fold_commitments(&data, weights, one)?;
  out.extend(ck.commit_zeros(&folded_blind[num_data_rows..])?);
  Ok(out)
}
/// R1CSWitness::fold_multiple (src/r1cs/mod.rs:570-660) on resident witnesses; weights_from_r (:153-166)
pub fn fold_multiple<F: Copy + Default>(ws: &[&HipTable<F>], weights: &[F], len: usize) -> Result<HipTable<F>, SpartanError> {
  let ptrs: Vec<*const sp_table> = ws.iter().map(|w| w.t as *const sp_table).collect();
  let out = HipTable::zeros(len)?;
  check(unsafe { sp_fold_tables(ctx(), ptrs.as_ptr(), ptrs.len(), limbs(weights), len, out.t) })?;
  Ok(out)
}
pub fn weights_from_r<F: Copy + Default>(r_bs: &[F], n: usize) -> Vec<F> {
  let mut w = vec![F::default(); n];
  unsafe { sp_weights_from_r(limbs(r_bs), r_bs.len(), n, limbs_mut(&mut w)) };
  w
}

// ---- NeutronNovaNIFS::prove rounds (src/neutronnova_zk.rs:511-1273) -----------------------------------------------------------------------------
/// The instance layers (Az, Bz, Cz of every step instance, written by multiply_vec into `layer(which, i)`) stay resident across proves
/// (`cached_step_matvec`, :1520-1590); a prove is `begin` + ell_b x (`round` -> the caller's process_round -> `challenge`) + `finish`.
pub struct HipNifs<F> {
  n: *mut sp_nifs,
  _p: PhantomData<F>,
}
unsafe impl<F> Send for HipNifs<F> {}
impl<F: Copy + Default> HipNifs<F> {
  pub fn new(n_padded: usize, left: usize, right: usize) -> Result<Self, SpartanError> {
    let mut n = ptr::null_mut();
    check(unsafe { sp_nifs_create(ctx(), n_padded, left, right, &mut n) })?;
    Ok(Self { n, _p: PhantomData })
  }
  /// window onto layer `idx` of matrix `which` (0 = A, 1 = B, 2 = C): the output table of that instance's multiply_vec
  pub fn layer(&mut self, which: usize, idx: usize) -> Result<HipTable<F>, SpartanError> {
    let mut t = ptr::null_mut();
    check(unsafe { sp_nifs_layer(self.n, which as c_int, idx, &mut t) })?;
    Ok(HipTable::from_raw(t))
  }
  /// `cached_step_i64` (:1548-1586): to_small_vec_or_zero mirrors of the layers, built once in prep_prove
  pub fn prepare_small(&mut self) -> Result<(), SpartanError> {
    check(unsafe { sp_nifs_prepare_small(self.n) })
  }
  /// E_eq = PowPolynomial::split_evals(tau) (left | right entries), rhos = the ell_b squeezed scalars; small = use the i64 mirrors for rounds 0-1
  pub fn begin(&mut self, e_eq: &[F], rhos: &[F], small: bool) -> Result<(), SpartanError> {
    check(unsafe { sp_nifs_begin(self.n, limbs(e_eq), limbs(rhos), rhos.len(), if small { 2 } else { 0 }) })
  }
  /// the cubic of round t as four coefficients (finish_round!, :703-735)
  pub fn round(&mut self, t: usize) -> Result<[F; 4], SpartanError> {
    let mut c = [F::default(); 4];
    check(unsafe { sp_nifs_round(self.n, t, limbs_mut(&mut c)) })?;
    Ok(c)
  }
  pub fn challenge(&mut self, r_b: &F) -> Result<(), SpartanError> {
    check(unsafe { sp_nifs_challenge(self.n, r_b as *const F as *const u64) })
  }
  /// the folded layers + (T_out, eq(rho, r_b))
  pub fn finish(&mut self, a: &mut HipTable<F>, b: &mut HipTable<F>, c: &mut HipTable<F>) -> Result<(F, F), SpartanError> {
    let (mut t_out, mut eq) = (F::default(), F::default());
    check(unsafe { sp_nifs_finish(self.n, a.t, b.t, c.t, &mut t_out as *mut F as *mut u64, &mut eq as *mut F as *mut u64) })?;
    Ok((t_out, eq))
  }
}
impl<F> Drop for HipNifs<F> {
  fn drop(&mut self) {
    unsafe { sp_nifs_free(self.n) }
  }
}

// ---- batched ZK sum-checks (src/sumcheck.rs:702-917) ---------------------------------------------------------------------------------------------
// Both take the verifier circuit's `process_round` as a closure: (round, coefficients of the step instance's polynomial, of the core's) -> the
// challenge r_i it squeezed after committing the round's witness (src/neutronnova_zk.rs:1771-1796, :1900-1925).
unsafe extern "C" fn round_trampoline<F, H>(user: *mut c_void, round: usize, cs: *const u64, cc: *const u64, ncoeffs: usize, r_out: *mut u64) -> c_int
where
  F: Copy,
  H: FnMut(usize, &[F], &[F]) -> Result<F, SpartanError>,
{
  let hook = &mut *(user as *mut H);
  let cs = std::slice::from_raw_parts(cs as *const F, ncoeffs);
  let cc = std::slice::from_raw_parts(cc as *const F, ncoeffs);
  match hook(round, cs, cc) {
    Ok(r) => {
      *(r_out as *mut F) = r;
      0
    }
    Err(_) => SP_ERR_INTERNAL,
  }
}
/// prove_quad_batched_zk (:702-782): tables [step A, core A, step B, core B]; returns (r, the four final claims)
pub fn prove_quad_batched_zk<F, H>(claims: &[F; 2], num_rounds: usize, t: &mut [HipTable<F>; 4], start_round: usize, mut hook: H) -> Result<(Vec<F>, [F; 4]), SpartanError>
where
  F: Copy + Default,
  H: FnMut(usize, &[F], &[F]) -> Result<F, SpartanError>,
{
  let mut r = vec![F::default(); num_rounds];
  let mut fin = [F::default(); 4];
  check(unsafe {
    sp_sumcheck_quad_batched(ctx(), limbs(claims), num_rounds, t[0].t, t[1].t, t[2].t, t[3].t, start_round, Some(round_trampoline::<F, H>),
                             &mut hook as *mut H as *mut c_void, limbs_mut(&mut r), limbs_mut(&mut fin))
  })?;
  Ok((r, fin))
}
/// prove_cubic_with_additive_term_batched_zk (:786-917) with the outer power table split as PowPolynomial::split_evals (left x right)
pub fn prove_cubic_with_additive_term_batched_zk<F, H>(
  num_rounds: usize,
  pow_left: &mut HipTable<F>,
  pow_right: &HipTable<F>,
  step: &mut [HipTable<F>; 3],
  core: &mut [HipTable<F>; 3],
  t_out_step: &F,
  start_round: usize,
  mut hook: H,
) -> Result<Vec<F>, SpartanError>
where
  F: Copy + Default,
  H: FnMut(usize, &[F], &[F]) -> Result<F, SpartanError>,
{
  let mut r = vec![F::default(); num_rounds];
  check(unsafe {
    sp_sumcheck_cubic_outer_pow_batched(ctx(), num_rounds, pow_left.t, pow_right.t, step[0].t, step[1].t, step[2].t, core[0].t, core[1].t, core[2].t,
                                        t_out_step as *const F as *const u64, start_round, Some(round_trampoline::<F, H>), &mut hook as *mut H as *mut c_void,
                                        limbs_mut(&mut r))
  })?;
  Ok(r)
}

// ---- wire formats (src/digest.rs:22-77) -----------------------------------------------------------------------------------------------------------
/// DigestHelperTrait::digest of SpartanVerifierKey (src/spartan.rs:73-104) without a serde pass over the matrices: the same byte stream
/// (bincode(vk_ee) || bincode(ck_s) || S.write_bytes()) fed to SHA-256 by the library. Points as affine limbs.
pub fn spartan_vk_digest(dims: &sp_dims, a: &sp_csr, b: &sp_csr, c: &sp_csr, ck: &[u64], h: &[u64; 8], ck_s: &[u64], h_s: &[u64; 8]) -> Result<[u8; 32], SpartanError> {
  let mut out = [0u8; 32];
  check(unsafe { sp_vk_digest(dims, a, b, c, ck.as_ptr(), ck.len() / 8, h.as_ptr(), ck_s.as_ptr(), ck_s.len() / 8, h_s.as_ptr(), out.as_mut_ptr()) })?;
  Ok(out)
}
/// SpartanSNARK <-> bincode bytes from / to the flat limb layout the prover functions above produce (include/spartan_hip.h sp_proof_serialize)
pub fn spartan_proof_to_bytes(layout: &sp_spartan_layout, words: &[u64]) -> Result<Vec<u8>, SpartanError> {
  let mut len = 0usize;
  check(unsafe { sp_proof_serialize(layout, words.as_ptr(), words.len(), ptr::null_mut(), 0, &mut len) })?;
  let mut out = vec![0u8; len];
  check(unsafe { sp_proof_serialize(layout, words.as_ptr(), words.len(), out.as_mut_ptr(), len, &mut len) })?;
  Ok(out)
}
pub fn spartan_proof_from_bytes(bytes: &[u8]) -> Result<(sp_spartan_layout, Vec<u64>), SpartanError> {
  let mut layout: sp_spartan_layout = unsafe { std::mem::zeroed() };
  let mut n = 0usize;
  check(unsafe { sp_proof_deserialize(bytes.as_ptr(), bytes.len(), &mut layout, ptr::null_mut(), 0, &mut n) })?;
  let mut words = vec![0u64; n];
  check(unsafe { sp_proof_deserialize(bytes.as_ptr(), bytes.len(), &mut layout, words.as_mut_ptr(), n, &mut n) })?;
  Ok((layout, words))
}
