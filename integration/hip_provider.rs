// src/provider/hip_provider.rs — the reference-side half of the binding (microsoft/Spartan2, Rust). It sits on top of hip_ffi.rs (generated from
// include/spartan_hip.h) and gives the reference's own types the trait impls / wrappers through which src/spartan.rs and src/neutronnova_zk.rs keep
// calling the same methods while the hot path runs on an MI355X:
//
//   * `impl DlogGroupExt for t256::Point`          src/provider/traits.rs:311-352 (the MSMs behind Hyrax commits)
//   * `HipTable`  = MultilinearPolynomial in HBM    src/polys/multilinear.rs:34-164
//   * `HipTranscript<E>: TranscriptEngineTrait<E>`  src/traits/transcript.rs:21-33 over src/provider/keccak.rs:26-105
//   * `prove_cubic_with_three_inputs` / `prove_quad` bodies   src/sumcheck.rs:190-247, 502-571
//   * `HyraxPCS::prove` body                        src/provider/pcs/hyrax_pc.rs:387-478 (one call: sp_hyrax_prove)
//
// The second half — SplitR1CSShape, the PCS commit / fold functions, NeutronNovaNIFS, the batched ZK sum-checks, the wire formats — is
// hip_r1cs_pcs.rs.
//
// This image has no Rust toolchain, so the file has NOT been compiled here; it is written against the reference's trait definitions as they
// stand in /root/reference at the surveyed commit and against the generated FFI module, and is the source a maintainer starts from instead of
// retyping INTEGRATION.md. The same call sequence, compiled and tested, is spartan2_amd/host/spartan_snark.cpp `prove_reference_order`.
#![allow(non_snake_case)]
use crate::{
  errors::SpartanError,
  provider::hip_ffi::*,
  traits::{transcript::{TranscriptEngineTrait, TranscriptReprTrait}, Engine},
};
use core::ffi::c_int;
use std::{ffi::CStr, marker::PhantomData, ptr, sync::OnceLock};

// ---- context: one per process and GPU (LOCAL_RANK picks the device under a one-process-per-GPU launcher) ---------------------------------
pub struct HipCtx(pub *mut sp_ctx);
unsafe impl Send for HipCtx {}
unsafe impl Sync for HipCtx {}
static CTX: OnceLock<HipCtx> = OnceLock::new();

pub fn ctx() -> *mut sp_ctx {
  CTX
    .get_or_init(|| {
      let dev: c_int = std::env::var("LOCAL_RANK").ok().and_then(|v| v.parse().ok()).unwrap_or(0);
      let mut c = ptr::null_mut();
      let rc = unsafe { sp_ctx_create(dev, &mut c) };
      assert!(rc == SP_OK, "libspartan_hip has no CPU fallback: {}", last_error());
      HipCtx(c)
    })
    .0
}

fn last_error() -> String {
  unsafe { CStr::from_ptr(sp_last_error()) }.to_string_lossy().into_owned()
}

/// 0 / -(error class) -> Result (src/errors.rs:13-110)
pub fn check(rc: c_int) -> Result<(), SpartanError> {
  match rc {
    SP_OK => Ok(()),
    SP_ERR_INVALID_INPUT_LENGTH => Err(SpartanError::InvalidInputLength { reason: last_error() }),
    SP_ERR_INVALID_WITNESS_LENGTH => Err(SpartanError::InvalidWitnessLength),
    SP_ERR_DIVISION_BY_ZERO => Err(SpartanError::DivisionByZero),
    SP_ERR_INTERNAL_TRANSCRIPT => Err(SpartanError::InternalTranscriptError),
    _ => Err(SpartanError::InternalError { reason: last_error() }), // incl. SP_ERR_NO_DEVICE
  }
}

// Field elements are #[repr(transparent)] over [u64; 4] Montgomery limbs (src/big_num/macros.rs:59-72): a slice of scalars IS the limb array.
#[inline]
fn limbs<F>(v: &[F]) -> *const u64 {
  v.as_ptr() as *const u64
}
#[inline]
fn limbs_mut<F>(v: &mut [F]) -> *mut u64 {
  v.as_mut_ptr() as *mut u64
}

// ---- DlogGroupExt (src/provider/traits.rs:311-352) ---------------------------------------------------------------------------------------
// Affine points cross as x | y limbs with (0, 0) for the identity; results come back canonical affine.
pub mod t256_msm {
  use super::*;
  use crate::provider::{pt256::t256, traits::{DlogGroup, DlogGroupExt}};
  use halo2curves::CurveAffine;
  use num_integer::Integer;
  use num_traits::ToPrimitive;

  fn pack(bases: &[t256::Affine]) -> Vec<u64> {
    let mut out = vec![0u64; 8 * bases.len()];
    for (i, b) in bases.iter().enumerate() {
      let c = b.coordinates();
      if bool::from(c.is_some()) {
        let c = c.unwrap();
        out[8 * i..8 * i + 4].copy_from_slice(&c.x().0);
        out[8 * i + 4..8 * i + 8].copy_from_slice(&c.y().0);
      }
    }
    out
  }
  fn unpack(a: &[u64]) -> t256::Point {
    if a.iter().all(|w| *w == 0) {
      return <t256::Point as DlogGroup>::zero();
    }
    let x = t256::Base(a[0..4].try_into().unwrap());
    let y = t256::Base(a[4..8].try_into().unwrap());
    t256::Point::from(t256::Affine::from_xy(x, y).unwrap())
  }

  impl DlogGroupExt for t256::Point {
    fn vartime_multiscalar_mul(scalars: &[Self::Scalar], bases: &[Self::AffineGroupElement], _par: bool) -> Result<Self, SpartanError> {
      if scalars.len() != bases.len() {
        return Err(SpartanError::InvalidInputLength { reason: "MSM: Coefficients and bases must have the same length".into() }); // msm.rs:194-198
      }
      let b = pack(bases);
      let mut out = [0u64; 8];
      check(unsafe { sp_msm(ctx(), limbs(scalars), b.as_ptr(), scalars.len(), out.as_mut_ptr()) })?;
      Ok(unpack(&out))
    }
    fn vartime_multiscalar_mul_small<T: Integer + Into<u64> + Copy + Sync + ToPrimitive>(
      scalars: &[T],
      bases: &[Self::AffineGroupElement],
      _par: bool,
    ) -> Result<Self, SpartanError> {
      let s: Vec<u64> = scalars.iter().map(|v| (*v).into()).collect();
      let b = pack(&bases[..s.len()]);
      let mut out = [0u64; 8];
      check(unsafe { sp_msm_small_u64(ctx(), s.as_ptr(), b.as_ptr(), s.len(), out.as_mut_ptr()) })?;
      Ok(unpack(&out))
    }
    fn vartime_multiscalar_mul_shared_weights(scalars: &[Self::Scalar], bases_rows: &[&[Self::AffineGroupElement]]) -> Result<Vec<Self>, SpartanError> {
      let n = scalars.len();
      let mut b = Vec::with_capacity(8 * n * bases_rows.len());
      for row in bases_rows {
        b.extend_from_slice(&pack(&row[..n]));
      }
      let mut out = vec![0u64; 8 * bases_rows.len()];
      check(unsafe { sp_msm_shared_weights(ctx(), limbs(scalars), n, b.as_ptr(), bases_rows.len(), out.as_mut_ptr()) })?;
      Ok(out.chunks(8).map(unpack).collect())
    }
  }
}

// ---- MultilinearPolynomial resident in HBM (src/polys/multilinear.rs:34-164) -------------------------------------------------------------
pub struct HipTable<F> {
  pub(crate) t: *mut sp_table,
  _p: PhantomData<F>,
}
unsafe impl<F> Send for HipTable<F> {}
impl<F> HipTable<F> {
  /// adopt a table handle the library returned (sp_table_zeros, sp_eq_table, sp_nifs_layer ...)
  pub(crate) fn from_raw(t: *mut sp_table) -> Self {
    Self { t, _p: PhantomData }
  }
}
impl<F: Copy + Default> HipTable<F> {
  /// MultilinearPolynomial::new (:62-66) / new_with_halves (:68-75): usize::MAX = "unknown" zero structure
  pub fn new(z: &[F], lo_eff: usize, hi_eff: usize) -> Result<Self, SpartanError> {
    let mut t = ptr::null_mut();
    check(unsafe { sp_table_from_host(ctx(), limbs(z), z.len(), lo_eff, hi_eff, &mut t) })?;
    Ok(Self { t, _p: PhantomData })
  }
  /// bind_poly_var_top (:95-164)
  pub fn bind_poly_var_top(&mut self, r: &F) -> Result<(), SpartanError> {
    check(unsafe { sp_table_bind_top(ctx(), self.t, r as *const F as *const u64) })
  }
  /// Index / into_vec
  pub fn read(&self, off: usize, cnt: usize) -> Result<Vec<F>, SpartanError> {
    let mut v = vec![F::default(); cnt];
    check(unsafe { sp_table_read(ctx(), self.t, off, cnt, limbs_mut(&mut v)) })?;
    Ok(v)
  }
}
impl<F> Drop for HipTable<F> {
  fn drop(&mut self) {
    unsafe { sp_table_free(self.t) }
  }
}

// ---- transcript (src/provider/keccak.rs:26-105 behind src/traits/transcript.rs:21-33) ----------------------------------------------------
pub struct HipTranscript<E: Engine> {
  pub(crate) t: *mut sp_transcript,
  _p: PhantomData<E>,
}
unsafe impl<E: Engine> Send for HipTranscript<E> {}
unsafe impl<E: Engine> Sync for HipTranscript<E> {}
impl<E: Engine> TranscriptEngineTrait<E> for HipTranscript<E> {
  fn new(label: &'static [u8]) -> Self {
    let mut t = ptr::null_mut();
    let rc = unsafe { sp_transcript_new(ctx(), label.as_ptr(), label.len(), &mut t) };
    assert!(rc == SP_OK);
    unsafe { sp_transcript_set_async(t, 1) }; // spartan.rs is single-threaded around its transcript: long absorbs hash beside its next calls
    Self { t, _p: PhantomData }
  }
  fn squeeze(&mut self, label: &'static [u8]) -> Result<E::Scalar, SpartanError> {
    let mut out = [0u64; 4];
    check(unsafe { sp_transcript_squeeze(self.t, label.as_ptr(), label.len(), out.as_mut_ptr()) })?;
    Ok(<E::Scalar as crate::big_num::montgomery::MontgomeryLimbs>::from_limbs(out))
  }
  fn absorb<T: TranscriptReprTrait<E::GE>>(&mut self, label: &'static [u8], o: &T) {
    // long inputs (a commitment's 64 bytes per row) are hashed on the library's thread while the caller goes on (sp_transcript_absorb)
    let b = o.to_transcript_bytes();
    unsafe { sp_transcript_absorb(self.t, label.as_ptr(), label.len(), b.as_ptr(), b.len()) };
  }
  fn dom_sep(&mut self, bytes: &'static [u8]) {
    unsafe { sp_transcript_dom_sep(self.t, bytes.as_ptr(), bytes.len()) };
  }
}
impl<E: Engine> Drop for HipTranscript<E> {
  fn drop(&mut self) {
    unsafe { sp_transcript_free(self.t) }
  }
}

// ---- sum-checks (src/sumcheck.rs) --------------------------------------------------------------------------------------------------------
/// SumcheckProof::prove_cubic_with_three_inputs (:502-571): returns (compressed polys: 3 coefficients per round, r, [A(r), B(r), C(r)])
pub fn prove_cubic_with_three_inputs<E: Engine>(
  claim: &E::Scalar,
  taus: &[E::Scalar],
  a: &mut HipTable<E::Scalar>,
  b: &mut HipTable<E::Scalar>,
  c: &mut HipTable<E::Scalar>,
  tr: &mut HipTranscript<E>,
) -> Result<(Vec<E::Scalar>, Vec<E::Scalar>, [E::Scalar; 3]), SpartanError>
where
  E::Scalar: Default + Copy,
{
  let ell = taus.len();
  let mut polys = vec![E::Scalar::default(); 3 * ell];
  let mut r = vec![E::Scalar::default(); ell];
  let mut fin = [E::Scalar::default(); 3];
  check(unsafe {
    sp_sumcheck_cubic3(ctx(), claim as *const _ as *const u64, limbs(taus), ell, a.t, b.t, c.t, tr.t, limbs_mut(&mut polys), limbs_mut(&mut r), limbs_mut(&mut fin))
  })?;
  Ok((polys, r, fin))
}
/// SumcheckProof::prove_quad (:190-247); src/spartan.rs:323-394 passes the 2M-long tables with (lo_eff, hi_eff) = (M, num_extra) instead of the manual round 0
pub fn prove_quad<E: Engine>(
  claim: &E::Scalar,
  rounds: usize,
  a: &mut HipTable<E::Scalar>,
  b: &mut HipTable<E::Scalar>,
  tr: &mut HipTranscript<E>,
) -> Result<(Vec<E::Scalar>, Vec<E::Scalar>, [E::Scalar; 2]), SpartanError>
where
  E::Scalar: Default + Copy,
{
  let mut polys = vec![E::Scalar::default(); 2 * rounds];
  let mut r = vec![E::Scalar::default(); rounds];
  let mut fin = [E::Scalar::default(); 2];
  check(unsafe { sp_sumcheck_quad(ctx(), claim as *const _ as *const u64, rounds, a.t, b.t, tr.t, limbs_mut(&mut polys), limbs_mut(&mut r), limbs_mut(&mut fin)) })?;
  Ok((polys, r, fin))
}

// ---- HyraxPCS::prove (src/provider/pcs/hyrax_pc.rs:387-478) as one call -------------------------------------------------------------------
/// `comm_rows` / `comm_eval`: affine x | y limbs (8 words per point); `rng`: (cols + 2) * 64 uniform bytes in the draw order of ipa.rs:139-149
/// (the reference draws them with E::Scalar::random(&mut OsRng) inside InnerProductArgumentLinear::prove; the shim fills the buffer from OsRng).
/// Returns (delta, beta, z_vec, z_delta, z_beta) as limbs: 8 | 8 | 4 * cols | 4 | 4 words.
pub fn hyrax_prove<E: Engine>(
  ck: *const sp_ck,
  ck_eval: *const sp_ck,
  tr: &mut HipTranscript<E>,
  comm_rows: &[u64],
  poly: &HipTable<E::Scalar>,
  n: usize,
  blinds: &[E::Scalar],
  point: &[E::Scalar],
  comm_eval: &[u64; 8],
  blind_eval: &E::Scalar,
  rng: &[u8],
) -> Result<Vec<u64>, SpartanError> {
  let rows = comm_rows.len() / 8;
  let cols = n / rows;
  let mut out = vec![0u64; 16 + 4 * cols + 8];
  check(unsafe {
    sp_hyrax_prove(ctx(), ck, ck_eval, tr.t, comm_rows.as_ptr(), rows, poly.t, n, limbs(blinds), limbs(point), point.len(), comm_eval.as_ptr(),
                   blind_eval as *const _ as *const u64, rng.as_ptr(), rng.len() / 64, out.as_mut_ptr())
  })?;
  Ok(out)
}
