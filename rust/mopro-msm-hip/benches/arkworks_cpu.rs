//! arkworks CPU baseline beside the HIP engine.  Run on a host that has cargo:
//!   RAYON_NUM_THREADS=<cores> cargo bench --bench arkworks_cpu
use ark_bn254::{Fr, G1Affine, G1Projective};
use ark_ec::{CurveGroup, VariableBaseMSM};
use ark_std::{test_rng, UniformRand};
use criterion::{criterion_group, criterion_main, Criterion};

fn bench(c: &mut Criterion) {
    let mut rng = test_rng();
    for log_n in [16usize, 20] {
        let n = 1 << log_n;
        let bases: Vec<G1Affine> = (0..n).map(|_| G1Projective::rand(&mut rng).into_affine()).collect();
        let scalars: Vec<Fr> = (0..n).map(|_| Fr::rand(&mut rng)).collect();
        c.bench_function(&format!("arkworks_msm_2^{log_n}"), |b| b.iter(|| G1Projective::msm(&bases, &scalars).unwrap()));
        c.bench_function(&format!("hip_msm_2^{log_n}"), |b| {
            b.iter(|| mopro_msm_hip::metal_variable_base_msm(&bases, &scalars).unwrap())
        });
    }
}
criterion_group!(benches, bench);
criterion_main!(benches);
