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
    batch_layout: u32,
    host_threads: u32,
}
#[repr(C)]
struct MsmCtx {
    _private: [u8; 0],
}
#[repr(C)]
struct MsmMulti {
    _private: [u8; 0],
}
const MSM_FORM_MONT: u32 = 1;

extern "C" {
    fn msm_ctx_create(cfg: *const MsmConfig, out: *mut *mut MsmCtx) -> i32;
    fn msm_ctx_destroy(ctx: *mut MsmCtx);
    fn msm_last_error(ctx: *const MsmCtx) -> *const c_char;
    fn msm_bn254_g1(
        ctx: *mut MsmCtx, bases_xy: *const u32, base_form: u32, inf_mask: *const u8, scalars: *const u32, n: usize,
        out_jacobian_mont: *mut u32, out_affine_std: *mut u32, out_is_inf: *mut u8,
    ) -> i32;
    fn msm_bn254_g1_arkworks(
        ctx: *mut MsmCtx, bases: *const core::ffi::c_void, stride: usize, x_off: usize, y_off: usize, inf_off: usize,
        scalars_mont: *const u32, n: usize, out_jacobian_mont: *mut u32, out_affine_std: *mut u32, out_is_inf: *mut u8,
    ) -> i32;
    fn msm_multi_create(devices: *const i32, ndev: i32, cfg: *const MsmConfig, exchange: u32, out: *mut *mut MsmMulti) -> i32;
    fn msm_multi_num_devices(m: *const MsmMulti) -> i32;
    fn msm_multi_destroy(m: *mut MsmMulti);
    fn msm_multi_last_error(m: *const MsmMulti) -> *const c_char;
    fn msm_bn254_g1_multi_arkworks(
        m: *mut MsmMulti, bases: *const core::ffi::c_void, stride: usize, x_off: usize, y_off: usize, inf_off: usize,
        scalars_mont: *const u32, n: usize, out_jacobian_mont: *mut u32, out_affine_std: *mut u32, out_is_inf: *mut u8,
    ) -> i32;
    fn msm_bn254_g1_upload_compressed(ctx: *mut MsmCtx, compressed: *const u8, n: usize, first_invalid: *mut i64) -> i32;
    fn msm_bn254_g1_resident(
        ctx: *mut MsmCtx, scalars: *const u32, n: usize, out_jacobian_mont: *mut u32, out_affine_std: *mut u32, out_is_inf: *mut u8,
    ) -> i32;
    fn msm_bn254_g1_upload_bases(ctx: *mut MsmCtx, bases_xy: *const u32, base_form: u32, inf_mask: *const u8, n: usize) -> i32;
    fn msm_bn254_g1_resident_batch(
        ctx: *mut MsmCtx, scalars: *const *const u32, n: usize, count: usize, out_jacobian_mont: *mut u32, out_affine_std: *mut u32,
        out_is_inf: *mut u8,
    ) -> i32;
    fn msm_bn254_g1_resident_device(
        ctx: *mut MsmCtx, d_scalars: *const core::ffi::c_void, n: usize, hip_stream: *mut core::ffi::c_void, out_jacobian_mont: *mut u32,
        out_affine_std: *mut u32, out_is_inf: *mut u8,
    ) -> i32;
    fn msm_abi_version() -> u32;
    fn msm_bn254_g1_combine_flags(
        partials_jacobian_mont: *const u32, k: usize, flags: u32, out_jacobian_mont: *mut u32, out_affine_std: *mut u32, out_is_inf: *mut u8,
    ) -> i32;
}

struct Ctx(*mut MsmCtx);
unsafe impl Send for Ctx {}

/// Process-global context: the reference rebuilds its whole Metal pipeline on every call
/// (metal_msm.rs:693); here device, stream and HBM workspace persist.
/// The header this file's `#[repr(C)]` structs and externs were written against (include/msm_hip.h MSM_HIP_ABI_VERSION): a library of
/// another ABI would read `MsmConfig` past its end, so every handle creation checks it first.
const MSM_HIP_ABI_VERSION: u32 = 7;
fn abi_check() -> Result<(), String> {
    let have = unsafe { msm_abi_version() };
    if have == MSM_HIP_ABI_VERSION { Ok(()) } else { Err(format!("libmsm_hip.so has ABI {have}, mopro-msm-hip is built against ABI {MSM_HIP_ABI_VERSION}")) }
}
/// `msm_config_t.flags` of the process-global handles: `MSM_HIP_DETERMINISTIC=1` in the environment asks for the canonical Z = 1
/// Jacobian limbs (MSM_FLAG_DETERMINISTIC) from `metal_variable_base_msm` -- the opt-in for callers that hash or `memcmp` a
/// `G1Projective` instead of comparing it with `==` (include/msm_hip.h, "Determinism").
fn global_flags() -> u32 {
    match std::env::var("MSM_HIP_DETERMINISTIC") {
        Ok(v) if !v.is_empty() && v != "0" => MSM_FLAG_DETERMINISTIC,
        _ => 0,
    }
}
static CTX: Lazy<Mutex<Result<Ctx, String>>> = Lazy::new(|| {
    if let Err(e) = abi_check() {
        return Mutex::new(Err(e));
    }
    let cfg = MsmConfig { device: -1, window_bits: 0, flags: global_flags(), stream_chunk_log2: 0, max_points: 0, batch_layout: 0, host_threads: 0 };
    let mut p: *mut MsmCtx = std::ptr::null_mut();
    let rc = unsafe { msm_ctx_create(&cfg, &mut p) };
    Mutex::new(if rc == 0 { Ok(Ctx(p)) } else { Err(last_error(std::ptr::null())) })
});

fn multi_error(m: *const MsmMulti) -> String {
    unsafe {
        let s = msm_multi_last_error(m);
        if s.is_null() { "msm_hip multi-GPU error".into() } else { std::ffi::CStr::from_ptr(s).to_string_lossy().into_owned() }
    }
}

fn last_error(ctx: *const MsmCtx) -> String {
    unsafe {
        let s = msm_last_error(ctx);
        if s.is_null() { "msm_hip error".into() } else { std::ffi::CStr::from_ptr(s).to_string_lossy().into_owned() }
    }
}

/// How `[G1Affine]` sits in memory.  The struct is not `repr(C)`: the offsets are MEASURED once (addr_of!) and then CHECKED against
/// the field accessors on a probe value before the raw slice is ever handed to the GPU.
#[derive(Clone, Copy)]
struct Layout {
    stride: usize,
    x_off: usize,
    y_off: usize,
    inf_off: usize,
}
static LAYOUT: Lazy<Option<Layout>> = Lazy::new(|| {
    use ark_ec::AffineRepr;
    let probe = [G1Affine::generator(), G1Affine::identity()];
    let base = &probe[0] as *const G1Affine as usize;
    let l = Layout {
        stride: core::mem::size_of::<G1Affine>(),
        x_off: core::ptr::addr_of!(probe[0].x) as usize - base,
        y_off: core::ptr::addr_of!(probe[0].y) as usize - base,
        inf_off: core::ptr::addr_of!(probe[0].infinity) as usize - base,
    };
    // what the C side requires (include/msm_hip.h msm_bn254_g1_arkworks) and what the raw bytes must say about the two probes
    let ok = core::mem::size_of::<Fq>() == 32
        && core::mem::size_of::<Fr>() == 32
        && l.stride >= 64 && l.stride % 4 == 0 && l.x_off % 4 == 0 && l.y_off % 4 == 0
        && l.x_off + 32 <= l.stride && l.y_off + 32 <= l.stride && l.inf_off < l.stride
        && (l.x_off + 32 <= l.y_off || l.y_off + 32 <= l.x_off)
        && unsafe {
            let raw = core::slice::from_raw_parts(base as *const u8, 2 * l.stride);
            let word = |off: usize| u64::from_le_bytes(raw[off..off + 8].try_into().unwrap());
            (0..4).all(|k| word(l.x_off + 8 * k) == probe[0].x.0 .0[k] && word(l.y_off + 8 * k) == probe[0].y.0 .0[k])
                && raw[l.inf_off] == 0 && raw[l.stride + l.inf_off] == 1
        };
    if ok { Some(l) } else { None }
});

/// Multi-GPU handle (include/msm_hip.h "multi-GPU"): one context + host thread per device, point-range shards, partials
/// exchanged with RCCL.  OPT-IN: only when `MSM_HIP_DEVICES=0,1,...` names the devices -- a one-process-per-GPU deployment
/// (every rank calling this function on its own device) must not open contexts and a communicator on all GPUs of the node.
/// A handle that ends up with a single device is destroyed again (nothing is kept, nothing leaks).
struct Multi(*mut MsmMulti);
unsafe impl Send for Multi {}
static MULTI: Lazy<Mutex<Option<Multi>>> = Lazy::new(|| {
    if std::env::var_os("MSM_HIP_DEVICES").is_none() || abi_check().is_err() {
        return Mutex::new(None);
    }
    let cfg = MsmConfig { device: -1, window_bits: 0, flags: global_flags(), stream_chunk_log2: 0, max_points: 0, batch_layout: 0, host_threads: 0 };
    let mut p: *mut MsmMulti = std::ptr::null_mut();
    let rc = unsafe { msm_multi_create(std::ptr::null(), 0, &cfg, 0 /* MSM_MULTI_EXCHANGE_AUTO */, &mut p) };
    if rc == 0 && unsafe { msm_multi_num_devices(p) } > 1 {
        return Mutex::new(Some(Multi(p)));
    }
    if !p.is_null() {
        unsafe { msm_multi_destroy(p) };
    }
    Mutex::new(None)
});
/// below this many points one GPU is faster than several (per-GPU fixed costs ~0.35 ms)
const MULTI_MIN_POINTS: usize = 1 << 19;

fn to_projective(jac: &[u64; 12]) -> G1Projective {
    // Jacobian Montgomery limbs -> G1Projective without any conversion (reference: metal_msm.rs:228-241)
    let f = |w: &[u64]| Fq::new_unchecked(BigInt::<4>([w[0], w[1], w[2], w[3]]));
    G1Projective::new_unchecked(f(&jac[0..4]), f(&jac[4..8]), f(&jac[8..12]))
}

/// Same name and signature as the reference entry point (metal_msm.rs:642-645).
///
/// The two slices go to the GPU(s) AS THEY ARE: `Fq`/`Fr` are arkworks' Montgomery words (R = 2^256), the engine reads the struct
/// array through the measured layout and reduces the scalars on the device -- the reference's whole `pack_affine_and_scalars`
/// stage (utils/limbs_conversion.rs:311-378: 3 CPU Montgomery reductions + 3 heap allocations per point) is gone.  With
/// `MSM_HIP_DEVICES` set, calls of 2^19 points and more shard the point range over those devices behind this unchanged signature.  Only if the layout probe fails
/// (a future arkworks changing `G1Affine`) are the points repacked -- in parallel, still without any field reduction.
pub fn metal_variable_base_msm(bases: &[G1Affine], scalars: &[Fr]) -> Result<G1Projective, Box<dyn Error>> {
    if bases.is_empty() || scalars.is_empty() {
        return Err("Empty input".into()); // metal_msm.rs:647-649
    }
    let n = bases.len().min(scalars.len()); // metal_msm.rs:652-656
    let repacked: Vec<[u64; 9]>;
    let (ptr, l) = match *LAYOUT {
        Some(l) => (bases.as_ptr() as *const core::ffi::c_void, l),
        None => {
            use ark_std::{cfg_iter, vec::Vec};
            #[cfg(feature = "parallel")]
            use rayon::prelude::*;
            repacked = cfg_iter!(bases[..n])
                .map(|b| {
                    let mut r = [0u64; 9];
                    r[0..4].copy_from_slice(&b.x.0 .0);
                    r[4..8].copy_from_slice(&b.y.0 .0);
                    r[8] = b.infinity as u64;
                    r
                })
                .collect::<Vec<_>>();
            (repacked.as_ptr() as *const core::ffi::c_void, Layout { stride: 72, x_off: 0, y_off: 32, inf_off: 64 })
        }
    };
    let mut jac = [0u64; 12];
    let mut is_inf = 0u8;
    if n >= MULTI_MIN_POINTS {
        if let Some(m) = MULTI.lock().unwrap().as_ref() {
            let rc = unsafe {
                msm_bn254_g1_multi_arkworks(
                    m.0, ptr, l.stride, l.x_off, l.y_off, l.inf_off, scalars.as_ptr() as *const u32, n,
                    jac.as_mut_ptr() as *mut u32, std::ptr::null_mut(), &mut is_inf,
                )
            };
            return if rc == 0 { Ok(to_projective(&jac)) } else { Err(multi_error(m.0).into()) };
        }
    }
    let guard = CTX.lock().unwrap();
    let ctx = guard.as_ref().map_err(|e| e.clone())?;
    let rc = unsafe {
        msm_bn254_g1_arkworks(
            ctx.0, ptr, l.stride, l.x_off, l.y_off, l.inf_off, scalars.as_ptr() as *const u32, n,
            jac.as_mut_ptr() as *mut u32, std::ptr::null_mut(), &mut is_inf,
        )
    };
    if rc != 0 {
        return Err(last_error(ctx.0).into());
    }
    Ok(to_projective(&jac))
}

/// The packed-word call (`msm_bn254_g1`): coordinates copied out of the structs (Montgomery words, no reduction), scalars through
/// `into_bigint()` on the CPU.  Kept for callers that already hold packed words; `metal_variable_base_msm` does not use it.
pub fn hip_variable_base_msm_packed(bases: &[G1Affine], scalars: &[Fr]) -> Result<G1Projective, Box<dyn Error>> {
    if bases.is_empty() || scalars.is_empty() {
        return Err("Empty input".into());
    }
    let n = bases.len().min(scalars.len());
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
    Ok(to_projective(&jac))
}

/// Earlier name of the zero-copy call; `metal_variable_base_msm` is that call now.
pub fn hip_variable_base_msm_zero_copy(bases: &[G1Affine], scalars: &[Fr]) -> Result<G1Projective, Box<dyn Error>> {
    metal_variable_base_msm(bases, scalars)
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
    Ok(to_projective(&jac))
}

/// Several MSMs against the SAME bases -- what a prover does per proof (one call of `metal_variable_base_msm` per scalar vector in the
/// reference).  The bases are uploaded once and stay in HBM; the scalar vectors (`BigInt<4>`, standard form, as the harness holds them)
/// run through `msm_bn254_g1_resident_batch`, two MSMs in flight: per MSM 1.63 ms instead of 2.28 at 2^20, 0.40 instead of 0.56 at 2^17.
pub fn hip_variable_base_msm_batch(bases: &[G1Affine], scalar_sets: &[&[BigInt<4>]]) -> Result<Vec<G1Projective>, Box<dyn Error>> {
    if bases.is_empty() || scalar_sets.is_empty() || scalar_sets.iter().any(|s| s.is_empty()) {
        return Err("Empty input".into());
    }
    let len0 = scalar_sets[0].len();
    if scalar_sets.iter().any(|s| s.len() != len0) {
        return Err("hip_variable_base_msm_batch: the scalar vectors of one batch must have the same length".into());
    }
    let n = len0.min(bases.len());
    let mut xy = vec![0u64; n * 8];
    let mut inf = vec![0u8; n];
    for i in 0..n {
        let b = &bases[i];
        if b.infinity {
            inf[i] = 1;
        } else {
            xy[i * 8..i * 8 + 4].copy_from_slice(&b.x.0 .0);
            xy[i * 8 + 4..i * 8 + 8].copy_from_slice(&b.y.0 .0);
        }
    }
    let guard = CTX.lock().unwrap();
    let ctx = guard.as_ref().map_err(|e| e.clone())?;
    if unsafe { msm_bn254_g1_upload_bases(ctx.0, xy.as_ptr() as *const u32, MSM_FORM_MONT, inf.as_ptr(), n) } != 0 {
        return Err(last_error(ctx.0).into());
    }
    let ptrs: Vec<*const u32> = scalar_sets.iter().map(|s| s.as_ptr() as *const u32).collect();
    let mut jac = vec![[0u64; 12]; ptrs.len()];
    let rc = unsafe {
        msm_bn254_g1_resident_batch(
            ctx.0, ptrs.as_ptr(), n, ptrs.len(), jac.as_mut_ptr() as *mut u32, std::ptr::null_mut(), std::ptr::null_mut(),
        )
    };
    if rc != 0 {
        return Err(last_error(ctx.0).into());
    }
    Ok(jac.iter().map(to_projective).collect())
}

/// Fold the partial results of several GPUs / processes in the order given (host arithmetic inside the library; the "all-reduce" of a
/// one-process-per-GPU job after its all-gather).  `deterministic`: hand out the canonical Z = 1 limbs -- the same limbs a single GPU with
/// MSM_FLAG_DETERMINISTIC returns for this group element (msm_bn254_g1_combine_flags, ABI 7).
pub fn hip_combine_partials(partials: &[G1Projective], deterministic: bool) -> Result<G1Projective, Box<dyn Error>> {
    if partials.is_empty() {
        return Err("Empty input".into());
    }
    abi_check()?;
    let mut words = vec![0u64; partials.len() * 12];
    for (i, p) in partials.iter().enumerate() {
        words[i * 12..i * 12 + 4].copy_from_slice(&p.x.0 .0);
        words[i * 12 + 4..i * 12 + 8].copy_from_slice(&p.y.0 .0);
        words[i * 12 + 8..i * 12 + 12].copy_from_slice(&p.z.0 .0);
    }
    let mut jac = [0u64; 12];
    let rc = unsafe {
        msm_bn254_g1_combine_flags(
            words.as_ptr() as *const u32, partials.len(), if deterministic { MSM_FLAG_DETERMINISTIC } else { 0 }, jac.as_mut_ptr() as *mut u32,
            std::ptr::null_mut(), std::ptr::null_mut(),
        )
    };
    if rc == 0 { Ok(to_projective(&jac)) } else { Err(format!("msm_bn254_g1_combine_flags failed ({rc})").into()) }
}

/// `msm_config_t.flags` bit: resident sets carry their window table (include/msm_hip.h MSM_FLAG_WINDOW_TABLE).
pub const MSM_FLAG_WINDOW_TABLE: u32 = 4;
/// `msm_config_t.flags` bit (ABI 6): the Jacobian words handed back are the canonical Z = 1 representative -- the same limbs for the same group
/// element on every call.  Without it `G1Projective` values returned by two identical calls are EQUAL (`==` compares projectively, as the
/// reference's own `assert_eq!` relies on, T/cuzk/e2e.rs:58-61) but their limbs may differ: the GPU sort places the entries of a bucket with
/// atomics, and X : Y : Z depends on the order of a bucket's additions (include/msm_hip.h, "Determinism").
pub const MSM_FLAG_DETERMINISTIC: u32 = 8;

/// A base set that stays in HBM for the lifetime of the value -- a prover's proving key.  The reference has no counterpart (it re-packs
/// and re-uploads the bases on every `metal_variable_base_msm` call, metal_msm.rs:94, 274-344); `hip_variable_base_msm_batch` above uploads
/// per call.  With `window_table = true` the upload also builds T_j[i] = 2^(c j) P_i for every window (39 ms and 872 MB at 2^20 points) and
/// every MSM on the whole set adds all windows into ONE bucket array: -7..-11 % per MSM at 2^20, -10..-20 % from 2^14 to 2^18 points.  The
/// build pays for itself after a few hundred MSMs at 2^20 (tens at 2^16): use it for keys that live as long as the process.
/// Owns a context of its own, so the process-global one behind `metal_variable_base_msm` is not disturbed.
pub struct HipResidentBases {
    ctx: *mut MsmCtx,
    n: usize,
}
unsafe impl Send for HipResidentBases {}

impl HipResidentBases {
    pub fn new(bases: &[G1Affine], window_table: bool) -> Result<Self, Box<dyn Error>> {
        Self::with_flags(bases, if window_table { MSM_FLAG_WINDOW_TABLE } else { 0 })
    }
    /// `flags`: any of MSM_FLAG_WINDOW_TABLE | MSM_FLAG_DETERMINISTIC (canonical Z = 1 limbs from every MSM on this set)
    pub fn with_flags(bases: &[G1Affine], flags: u32) -> Result<Self, Box<dyn Error>> {
        if bases.is_empty() {
            return Err("Empty input".into());
        }
        abi_check()?;
        let cfg = MsmConfig { device: -1, window_bits: 0, flags, stream_chunk_log2: 0, max_points: 0, batch_layout: 0, host_threads: 0 };
        let mut p: *mut MsmCtx = std::ptr::null_mut();
        if unsafe { msm_ctx_create(&cfg, &mut p) } != 0 {
            return Err(last_error(std::ptr::null()).into());
        }
        let n = bases.len();
        let mut xy = vec![0u64; n * 8];
        let mut inf = vec![0u8; n];
        for (i, b) in bases.iter().enumerate() {
            if b.infinity {
                inf[i] = 1;
            } else {
                xy[i * 8..i * 8 + 4].copy_from_slice(&b.x.0 .0);
                xy[i * 8 + 4..i * 8 + 8].copy_from_slice(&b.y.0 .0);
            }
        }
        let me = HipResidentBases { ctx: p, n };
        if unsafe { msm_bn254_g1_upload_bases(p, xy.as_ptr() as *const u32, MSM_FORM_MONT, inf.as_ptr(), n) } != 0 {
            return Err(last_error(p).into()); // `me` is dropped: the context is destroyed
        }
        Ok(me)
    }

    /// one MSM; fewer scalars than bases truncate to the shorter (metal_msm.rs:652-656)
    pub fn msm(&self, scalars: &[BigInt<4>]) -> Result<G1Projective, Box<dyn Error>> {
        if scalars.is_empty() {
            return Err("Empty input".into());
        }
        let mut jac = [0u64; 12];
        let mut is_inf = 0u8;
        let rc = unsafe {
            msm_bn254_g1_resident(
                self.ctx, scalars.as_ptr() as *const u32, scalars.len().min(self.n), jac.as_mut_ptr() as *mut u32, std::ptr::null_mut(), &mut is_inf,
            )
        };
        if rc != 0 {
            return Err(last_error(self.ctx).into());
        }
        Ok(to_projective(&jac))
    }

    /// scalars that already live in HBM (`n` x `BigInt<4>` at the device address `d_scalars`, produced on `hip_stream` or NULL):
    /// a prover whose witness is computed on the GPU.  Nothing but the result crosses PCIe (`msm_bn254_g1_resident_device`).
    ///
    /// # Safety
    /// `d_scalars` must be a device pointer to at least `n` scalars on this handle's device, valid until the call returns.
    pub unsafe fn msm_device_scalars(&self, d_scalars: *const core::ffi::c_void, n: usize, hip_stream: *mut core::ffi::c_void) -> Result<G1Projective, Box<dyn Error>> {
        if n == 0 || d_scalars.is_null() {
            return Err("Empty input".into());
        }
        let mut jac = [0u64; 12];
        let mut is_inf = 0u8;
        let rc = msm_bn254_g1_resident_device(
            self.ctx, d_scalars, n.min(self.n), hip_stream, jac.as_mut_ptr() as *mut u32, std::ptr::null_mut(), &mut is_inf,
        );
        if rc != 0 {
            return Err(last_error(self.ctx).into());
        }
        Ok(to_projective(&jac))
    }

    /// several scalar vectors, two MSMs in flight (`msm_bn254_g1_resident_batch`).  Every vector must have the SAME length: the C call
    /// takes one `n` for the whole batch, and truncating the longer vectors to the shortest would silently return results that differ
    /// from per-call `msm()` (which truncates each call to `min(len, bases)` on its own, metal_msm.rs:652-656) -- so unequal lengths are
    /// an error here.  The window table (and the GLV records) serve calls on the WHOLE resident set only: a batch of shorter vectors runs
    /// the plain pipeline on the first `n` bases.
    pub fn msm_batch(&self, scalar_sets: &[&[BigInt<4>]]) -> Result<Vec<G1Projective>, Box<dyn Error>> {
        if scalar_sets.is_empty() || scalar_sets.iter().any(|s| s.is_empty()) {
            return Err("Empty input".into());
        }
        let len0 = scalar_sets[0].len();
        if scalar_sets.iter().any(|s| s.len() != len0) {
            return Err("msm_batch: the scalar vectors of one batch must have the same length (call msm() per vector, or batch by length)".into());
        }
        let n = len0.min(self.n);
        let ptrs: Vec<*const u32> = scalar_sets.iter().map(|s| s.as_ptr() as *const u32).collect();
        let mut jac = vec![[0u64; 12]; ptrs.len()];
        let rc = unsafe {
            msm_bn254_g1_resident_batch(
                self.ctx, ptrs.as_ptr(), n, ptrs.len(), jac.as_mut_ptr() as *mut u32, std::ptr::null_mut(), std::ptr::null_mut(),
            )
        };
        if rc != 0 {
            return Err(last_error(self.ctx).into());
        }
        Ok(jac.iter().map(to_projective).collect())
    }
}

impl Drop for HipResidentBases {
    fn drop(&mut self) {
        unsafe { msm_ctx_destroy(self.ctx) };
    }
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
        // a batch against the same bases equals the single calls
        let n = 1 << 12;
        let bases: Vec<G1Affine> = (0..n).map(|_| G1Projective::rand(&mut rng).into_affine()).collect();
        let sets: Vec<Vec<Fr>> = (0..5).map(|_| (0..n).map(|_| Fr::rand(&mut rng)).collect()).collect();
        let big: Vec<Vec<BigInt<4>>> = sets.iter().map(|s| s.iter().map(|x| x.into_bigint()).collect()).collect();
        let refs: Vec<&[BigInt<4>]> = big.iter().map(|v| v.as_slice()).collect();
        let got = hip_variable_base_msm_batch(&bases, &refs).unwrap();
        for (s, g) in sets.iter().zip(got.iter()) {
            assert_eq!(*g, G1Projective::msm(&bases, s).unwrap());
        }
        // points at infinity inside the slice, truncation to the shorter slice, and the packed-word variant
        let n = 1 << 10;
        let mut bases: Vec<G1Affine> = (0..n).map(|_| G1Projective::rand(&mut rng).into_affine()).collect();
        bases[3] = G1Affine::identity();
        let scalars: Vec<Fr> = (0..n + 5).map(|_| Fr::rand(&mut rng)).collect();
        let want = G1Projective::msm(&bases, &scalars[..n]).unwrap();
        assert_eq!(metal_variable_base_msm(&bases, &scalars).unwrap(), want);
        assert_eq!(hip_variable_base_msm_packed(&bases, &scalars).unwrap(), want);
        assert!(LAYOUT.is_some(), "G1Affine layout probe failed: the shim repacks (correct, slower)");
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
