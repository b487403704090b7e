//! Drop-in replacement for `mopro_msm::msm::metal_msm::metal_variable_base_msm`
//! (reference: mopro-msm/src/msm/metal_msm/metal_msm.rs:642-695), backed by the MI355X HIP engine
//! through the C ABI declared in `include/msm_hip.h`.
//!
//! Semantics kept from the reference:
//!   * empty `bases` or `scalars`        -> `Err("Empty input")`            (metal_msm.rs:647-649)
//!   * unequal lengths                   -> truncated to the shorter slice  (metal_msm.rs:652-656)
//!   * result equals `G1Projective::msm(bases, scalars)` as a group element (T/cuzk/e2e.rs:58-61)
//! and, unlike the reference, also for points at infinity, duplicated bases and any `n >= 1`.
//!
//! This file cannot be compiled in the development image (no Rust toolchain); it is the binding a
//! maintainer adds.  Layout note: arkworks' `Fq` is `Fp<MontBackend<FqConfig,4>,4>(BigInt<4>([u64;4]))`
//! in Montgomery form with R = 2^256, which is exactly the engine's `MSM_FORM_MONT` word format
//! (little-endian u64 limbs == pairs of little-endian u32 words on x86-64 / aarch64).
use ark_bn254::{Fq, Fr, G1Affine, G1Projective};
use ark_ff::{BigInt, PrimeField};
use once_cell::sync::Lazy;
use std::error::Error;
use std::os::raw::c_char;
use std::sync::Mutex;

#[repr(C)]
struct MsmConfig {
    device: i32,
    window_bits: u32,
    flags: u32,
    stream_chunk_log2: u32,
    max_points: u64,
}
#[repr(C)]
struct MsmCtx {
    _private: [u8; 0],
}
const MSM_FORM_MONT: u32 = 1;

extern "C" {
    fn msm_ctx_create(cfg: *const MsmConfig, out: *mut *mut MsmCtx) -> i32;
    fn msm_last_error(ctx: *const MsmCtx) -> *const c_char;
    fn msm_bn254_g1(
        ctx: *mut MsmCtx, bases_xy: *const u32, base_form: u32, inf_mask: *const u8, scalars: *const u32, n: usize,
        out_jacobian_mont: *mut u32, out_affine_std: *mut u32, out_is_inf: *mut u8,
    ) -> i32;
    fn msm_bn254_g1_arkworks(
        ctx: *mut MsmCtx, bases: *const core::ffi::c_void, stride: usize, x_off: usize, y_off: usize, inf_off: usize,
        scalars_mont: *const u32, n: usize, out_jacobian_mont: *mut u32, out_affine_std: *mut u32, out_is_inf: *mut u8,
    ) -> i32;
    fn msm_bn254_g1_upload_compressed(ctx: *mut MsmCtx, compressed: *const u8, n: usize, first_invalid: *mut i64) -> i32;
    fn msm_bn254_g1_resident(
        ctx: *mut MsmCtx, scalars: *const u32, n: usize, out_jacobian_mont: *mut u32, out_affine_std: *mut u32, out_is_inf: *mut u8,
    ) -> i32;
}

struct Ctx(*mut MsmCtx);
unsafe impl Send for Ctx {}

/// Process-global context: the reference rebuilds its whole Metal pipeline on every call
/// (metal_msm.rs:693); here device, stream and HBM workspace persist.
static CTX: Lazy<Mutex<Result<Ctx, String>>> = Lazy::new(|| {
    let cfg = MsmConfig { device: -1, window_bits: 0, flags: 0, stream_chunk_log2: 0, max_points: 0 };
    let mut p: *mut MsmCtx = std::ptr::null_mut();
    let rc = unsafe { msm_ctx_create(&cfg, &mut p) };
    Mutex::new(if rc == 0 { Ok(Ctx(p)) } else { Err(last_error(std::ptr::null())) })
});

fn last_error(ctx: *const MsmCtx) -> String {
    unsafe {
        let s = msm_last_error(ctx);
        if s.is_null() { "msm_hip error".into() } else { std::ffi::CStr::from_ptr(s).to_string_lossy().into_owned() }
    }
}

/// Same name and signature as the reference entry point.
pub fn metal_variable_base_msm(mut bases: &[G1Affine], mut scalars: &[Fr]) -> Result<G1Projective, Box<dyn Error>> {
    if bases.is_empty() || scalars.is_empty() {
        return Err("Empty input".into());
    }
    let n = bases.len().min(scalars.len());
    bases = &bases[..n];
    scalars = &scalars[..n];

    // Pack: x.0.0 / y.0.0 are already Montgomery (R = 2^256) limbs -> plain copy, no field reduction
    // (the reference does 3 Montgomery reductions + 3 heap allocations per point here,
    //  utils/limbs_conversion.rs:311-378).  G1Affine is not repr(C): read fields, never offsets.
    let mut xy = vec![0u64; n * 8];
    let mut inf = vec![0u8; n];
    let mut sc = vec![0u64; n * 4];
    for i in 0..n {
        let b = &bases[i];
        if b.infinity {
            inf[i] = 1;
        } else {
            xy[i * 8..i * 8 + 4].copy_from_slice(&b.x.0 .0);
            xy[i * 8 + 4..i * 8 + 8].copy_from_slice(&b.y.0 .0);
        }
        sc[i * 4..i * 4 + 4].copy_from_slice(&scalars[i].into_bigint().0); // standard form, < r
    }

    let guard = CTX.lock().unwrap();
    let ctx = guard.as_ref().map_err(|e| e.clone())?;
    let mut jac = [0u64; 12];
    let mut is_inf = 0u8;
    let rc = unsafe {
        msm_bn254_g1(
            ctx.0, xy.as_ptr() as *const u32, MSM_FORM_MONT, inf.as_ptr(), sc.as_ptr() as *const u32, n,
            jac.as_mut_ptr() as *mut u32, std::ptr::null_mut(), &mut is_inf,
        )
    };
    if rc != 0 {
        return Err(last_error(ctx.0).into());
    }
    // Jacobian Montgomery limbs -> G1Projective without any conversion (reference: metal_msm.rs:228-241)
    let f = |w: &[u64]| Fq::new_unchecked(BigInt::<4>([w[0], w[1], w[2], w[3]]));
    Ok(G1Projective::new_unchecked(f(&jac[0..4]), f(&jac[4..8]), f(&jac[8..12])))
}

/// Zero-copy variant (include/msm_hip.h `msm_bn254_g1_arkworks`): the `[G1Affine]` and `[Fr]` slices go to the GPU as
/// they are.  The struct layout is MEASURED here (G1Affine is not repr(C)); Fr is `Fp<MontBackend<_,4>,4>` = [u64;4].
pub fn hip_variable_base_msm_zero_copy(bases: &[G1Affine], scalars: &[Fr]) -> Result<G1Projective, Box<dyn Error>> {
    if bases.is_empty() || scalars.is_empty() {
        return Err("Empty input".into());
    }
    let n = bases.len().min(scalars.len());
    let probe = &bases[0];
    let base_addr = probe as *const G1Affine as usize;
    let x_off = core::ptr::addr_of!(probe.x) as usize - base_addr;
    let y_off = core::ptr::addr_of!(probe.y) as usize - base_addr;
    let inf_off = core::ptr::addr_of!(probe.infinity) as usize - base_addr;
    assert_eq!(core::mem::size_of::<Fr>(), 32);
    let guard = CTX.lock().unwrap();
    let ctx = guard.as_ref().map_err(|e| e.clone())?;
    let mut jac = [0u64; 12];
    let mut is_inf = 0u8;
    let rc = unsafe {
        msm_bn254_g1_arkworks(
            ctx.0, bases.as_ptr() as *const core::ffi::c_void, core::mem::size_of::<G1Affine>(), x_off, y_off, inf_off,
            scalars.as_ptr() as *const u32, n, jac.as_mut_ptr() as *mut u32, std::ptr::null_mut(), &mut is_inf,
        )
    };
    if rc != 0 {
        return Err(last_error(ctx.0).into());
    }
    let f = |w: &[u64]| Fq::new_unchecked(BigInt::<4>([w[0], w[1], w[2], w[3]]));
    Ok(G1Projective::new_unchecked(f(&jac[0..4]), f(&jac[4..8]), f(&jac[8..12])))
}

/// The benchmark harness's read path (utils/preprocess.rs:101-131 + arkworks_pippenger.rs:7-43) without the CPU square
/// roots: `points_file_instance` is ONE instance of the `points` file exactly as `Vec<G1Affine>::serialize_compressed`
/// wrote it (8-byte little-endian length, then 32 bytes per point); the images are decoded on the GPU
/// (`msm_bn254_g1_upload_compressed`) and stay resident for the MSM.  `scalars` as in the harness: `BigInt<4>` standard form.
pub fn hip_msm_from_compressed_instance(
    points_file_instance: &[u8], scalars: &[BigInt<4>],
) -> Result<G1Projective, Box<dyn Error>> {
    if points_file_instance.len() < 8 {
        return Err("failed to read at least one instance from file".into());
    }
    let n_pts = u64::from_le_bytes(points_file_instance[0..8].try_into().unwrap()) as usize;
    let images = &points_file_instance[8..];
    if n_pts == 0 || scalars.is_empty() {
        return Err("Empty input".into());
    }
    if images.len() < 32 * n_pts {
        return Err("could not serialize".into());
    }
    let guard = CTX.lock().unwrap();
    let ctx = guard.as_ref().map_err(|e| e.clone())?;
    let mut bad: i64 = -1;
    let rc = unsafe { msm_bn254_g1_upload_compressed(ctx.0, images.as_ptr(), n_pts, &mut bad) };
    if rc != 0 {
        return Err(last_error(ctx.0).into()); // SerializationError::InvalidData: `bad` = first image that does not decode
    }
    let n = n_pts.min(scalars.len());
    let mut jac = [0u64; 12];
    let mut is_inf = 0u8;
    let rc = unsafe {
        msm_bn254_g1_resident(ctx.0, scalars.as_ptr() as *const u32, n, jac.as_mut_ptr() as *mut u32, std::ptr::null_mut(), &mut is_inf)
    };
    if rc != 0 {
        return Err(last_error(ctx.0).into());
    }
    let f = |w: &[u64]| Fq::new_unchecked(BigInt::<4>([w[0], w[1], w[2], w[3]]));
    Ok(G1Projective::new_unchecked(f(&jac[0..4]), f(&jac[4..8]), f(&jac[8..12])))
}

/// Alias under the engine's own name.
pub fn hip_variable_base_msm(bases: &[G1Affine], scalars: &[Fr]) -> Result<G1Projective, Box<dyn Error>> {
    metal_variable_base_msm(bases, scalars)
}

#[cfg(test)]
mod tests {
    use super::*;
    use ark_ec::{CurveGroup, VariableBaseMSM};
    use ark_std::{test_rng, UniformRand};

    // mirror of the reference's e2e test (metal_msm.rs:739-760, T/cuzk/e2e.rs:14-63)
    #[test]
    fn hip_msm_matches_arkworks() {
        let mut rng = test_rng();
        for log_n in [0usize, 1, 5, 10, 16] {
            let n = 1 << log_n;
            let bases: Vec<G1Affine> = (0..n).map(|_| G1Projective::rand(&mut rng).into_affine()).collect();
            let scalars: Vec<Fr> = (0..n).map(|_| Fr::rand(&mut rng)).collect();
            assert_eq!(metal_variable_base_msm(&bases, &scalars).unwrap(), G1Projective::msm(&bases, &scalars).unwrap());
        }
        assert!(metal_variable_base_msm(&[], &[]).is_err());
    }

    // the compressed image format is restated from ark-serialize 0.4 on the C side: this is the test that pins it
    #[test]
    fn compressed_instance_matches_arkworks() {
        use ark_ff::PrimeField;
        use ark_serialize::CanonicalSerialize;
        let mut rng = test_rng();
        let n = 1 << 10;
        let mut bases: Vec<G1Affine> = (0..n).map(|_| G1Projective::rand(&mut rng).into_affine()).collect();
        bases[7] = G1Affine::identity();
        let scalars: Vec<Fr> = (0..n).map(|_| Fr::rand(&mut rng)).collect();
        let mut file = Vec::new();
        bases.serialize_compressed(&mut file).unwrap();
        let bigints: Vec<BigInt<4>> = scalars.iter().map(|s| s.into_bigint()).collect();
        assert_eq!(hip_msm_from_compressed_instance(&file, &bigints).unwrap(), G1Projective::msm(&bases, &scalars).unwrap());
    }
}
