// msm_hip.hip -- host runtime + C ABI (include/msm_hip.h) of the MI355X-native BN254 G1 MSM.
//
// Replaces, for the one path metal_variable_base_msm (metal_msm.rs:642-695):
//   MetalMSMPipeline::{new,execute_pipeline,final_reduction}      metal_msm.rs:48-261
//   ShaderManager / MetalHelper / gpu::{create_buffer,read_buffer} host/shader_manager.rs:98-167,
//                                                                 host/metal_wrapper.rs:55-217, host/gpu.rs:3-31
// Design differences (MI355X-first):
//   * a persistent context owns the device, one HIP stream, the HBM workspace and the hipEvents; the
//     reference re-opens the device, reloads the metallib twice and builds six pipeline states on
//     EVERY call (metal_msm.rs:693 -> 64 -> 48, window_size_optimizer.rs:79-92);
//   * all intermediates stay in HBM -- the reference round-trips every stage through a host Vec<u32>
//     (metal_msm.rs:331-339, 403-407, 505-507, 630-632);
//   * launches are queued back to back on one stream; the only host synchronisation is the final
//     copy of W window sums (W*96 bytes).  The reference blocks after each of its 9 submits.
// There is NO CPU fallback: without a HIP device every compute entry point returns MSM_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/msm_hip.h"
#include "host_g1.hpp"
#include "msm_kernels.hpp"

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

enum { EV_START, EV_H2D, EV_CONVERT, EV_DECOMP, EV_SORT, EV_ACC0, EV_ACC1, EV_REDUCE, EV_COUNT };

}  // namespace

// Small persistent host thread pool for the CPU finish (per-window Horner chains are independent).  The
// reference runs its CPU finish under rayon (metal_msm.rs:214-247); std::thread + a condition variable here.
// run() returns when every JOB is done, not when every worker has checked in: a worker the OS wakes late (seen as
// 3-10 ms outliers of the finish stage) simply finds nothing left, because the caller and the punctual workers pull jobs
// from one ticket counter.  The ticket carries the generation, so a late worker can never take a job of a later run().
class HostPool {
public:
    explicit HostPool(int nthreads) {
        for (int i = 0; i < nthreads; i++) th_.emplace_back([this] { worker(); });
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
            gen_++;
            ticket_.store(gen_ << 32, std::memory_order_release);  // releases workers spinning in the armed state
        }
        cv_work_.notify_all();
        for (auto& t : th_) t.join();
    }
    int size() const { return (int)th_.size(); }
    // Wake the workers NOW and let them spin until the next run() publishes its jobs (or ~20 ms pass): called when a
    // pipeline is enqueued, so that the condition-variable wake-up (the source of the remaining 2-5 ms outliers: ~1 % of
    // the calls on a busy host) happens during the GPU's milliseconds instead of on the critical path of the finish.
    void arm() {
        {
            std::lock_guard<std::mutex> lk(m_);
            njobs_ = 0;  // "armed": nothing to pull yet
            const uint64_t gen = ++gen_;
            ticket_.store(gen << 32, std::memory_order_release);
        }
        cv_work_.notify_all();
    }
    // run fn(0..njobs-1) on the workers and the calling thread; returns when all jobs are done
    void run(int njobs, const std::function<void(int)>& fn) {
        uint64_t gen;
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &fn;
            njobs_ = njobs;
            gen = ++gen_;
            done_.store(0, std::memory_order_relaxed);
            ticket_.store(gen << 32, std::memory_order_release);
        }
        cv_work_.notify_all();
        pull(gen, njobs, fn);
        for (int spins = 0; done_.load(std::memory_order_acquire) < njobs; spins++)
            if (spins > 2000) std::this_thread::yield();  // the stragglers are <= one window chain (~40 us) long
    }

private:
    void pull(uint64_t gen, int njobs, const std::function<void(int)>& fn) {
        for (;;) {
            uint64_t v = ticket_.load(std::memory_order_acquire);
            if ((v >> 32) != gen || (int)(uint32_t)v >= njobs) return;
            if (!ticket_.compare_exchange_weak(v, v + 1, std::memory_order_acq_rel)) continue;
            fn((int)(uint32_t)v);  // fn outlives this call: run(gen) cannot return before done_ counts it
            done_.fetch_add(1, std::memory_order_release);
        }
    }
    void worker() {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)>* job;
            int njobs;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_work_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                job = job_;
                njobs = njobs_;
            }
            if (njobs == 0) {  // armed: spin (bounded) until run() moves the ticket to the next generation
                const auto t0 = std::chrono::steady_clock::now();
                for (uint32_t spins = 1; (ticket_.load(std::memory_order_acquire) >> 32) == seen; spins++) {
                    if ((spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
                    __builtin_ia32_pause();
                }
                continue;  // re-read generation and job under the lock (cv wait returns at once if run() has published)
            }
            pull(seen, njobs, *job);
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_work_;
    const std::function<void(int)>* job_ = nullptr;
    std::atomic<uint64_t> ticket_{0};  // generation << 32 | next job index
    std::atomic<int> done_{0};
    int njobs_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

struct msm_ctx {
    std::mutex mu;
    HostPool* pool = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;     // host->HBM chunk uploads of the streamed path
    hipEvent_t ev_copied[2]{}, ev_free[2]{};
    hipEvent_t ev_fork = nullptr, ev_bases = nullptr;  // base conversion runs on copy_stream beside the sort kernels
    DevBuf sbases[2], sscalars[2], sinf[2];  // double-buffered raw inputs of the streamed path
    uint32_t* h_sq = nullptr;              // pinned: per-chunk bit sums + flags of the streamed path
    size_t h_sq_cap = 0;
    msm_config_t cfg{};
    std::string err;
    hipEvent_t ev[EV_COUNT]{};
    // HBM workspace
    DevBuf bases, ibases, inf, scalars, digits, ranks, sorted, hist, offsets, blocksums, buckets, heads, tails, chunkmap, rc, flags,
        pow2, tilecounts, longlist, longdone, midlist, ccounts, cregion, bigslot, big;
    bool pow2_ready = false;
    uint32_t* h_qsums = nullptr;  // pinned: W x (kb+1) Jacobian bit sums
    uint32_t* h_flags = nullptr;    // pinned
    // resident bases
    size_t resident_n = 0;
    size_t wide_max = 40960;  // pairwise levels up to this many additions use 8 lanes per addition (MSM_HIP_WIDE_MAX; 0 = never)
    bool resident_has_inf = false;
    bool resident_glv = false;  // the resident set holds the phi records too (index resident_n + i)
    msm_timings_t tm{};
    bool stage_timing = false;  // record the per-stage hipEvents (each costs ~6 us of stream time); k_accumulate's pair is always on
    double acc_ms_sum = 0;
    uint64_t acc_launches = 0;
};

namespace {

int32_t fail(msm_ctx* c, int32_t code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_create_error = buf;
    return code;
}

#define HIPCHK(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(ctx, e_ == hipErrorOutOfMemory ? MSM_ERR_OOM : MSM_ERR_HIP, "%s failed: %s (%s:%d)", #call, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                    \
    } while (0)

int32_t ensure(msm_ctx* c, DevBuf& b, size_t bytes) {
    if (b.cap >= bytes) return MSM_OK;
    if (b.p) HIPCHK(c, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 8;  // a little slack so slowly growing n does not realloc each call
    HIPCHK(c, hipMalloc(&b.p, want));
    b.cap = want;
    return MSM_OK;
}
void release(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

// ---- planner: replaces the N -> window_size / scale_factor tables (metal_msm.rs:661-691).  The cuZK cost model
// (utils/window_size_optimizer.rs:38-51: per window N mixed adds plus ~2 full adds per bucket) gives the shape, but two
// measured effects decide the table below (tools/sweep_c.py, profiles/NOTES_r1.md "window sweep"):
//  * r < 2^254, so the top window only holds 254 mod c bits.  For c = 7, 9, 11, 12, 14 that is 1-2 bits: every point
//    lands in one of <= 3 buckets of that window, which serialises the LDS sort cursors and makes those buckets
//    thousands of chunks long (c = 12 at N = 2^19: 6.7 ms against 1.2 ms).  Only c in {8, 10, 13, 15, 16} (6, 4, 7, 14, 14
//    top bits) are used.
//  * below ~2^17 points the per-window fixed costs (dependent reduction levels, launches) outweigh the bucket count:
//    fewer, wider windows win earlier than the arithmetic model says.
// c is capped where one window's histogram still fits the LDS sort path (nb <= 32768: 16 signed, 15 unsigned).
uint32_t plan_window_bits(size_t n, bool is_signed) {
    // re-measured after the reduction-tree and host-latency work (tools/sweep_c.py): 2^13: c = 8 0.335 ms (13: 0.455);
    // 2^14: c = 10 0.397 (13: 0.431); 2^15: 10 0.412 (13: 0.440); 2^16: 13 0.489; 2^17: 15 0.587; 2^18: 15 0.769 (16: 0.790);
    // 2^19: 16 1.125 (15: 1.197).  c = 10 leaves the top window 4 bits (9 buckets of n/16 points): fine for the long-bucket path.
    uint32_t c = n <= ((size_t)1 << 13) ? 8u : n <= ((size_t)1 << 15) ? 10u : n <= ((size_t)1 << 16) ? 13u : n <= ((size_t)1 << 18) ? 15u : 16u;
    if (!is_signed && c > 15u) c = 15u;
    return c;
}
// GLV (glv_bn254.hpp): 2n virtual points with 127-bit scalars -- the same additions in half the windows: half the buckets to
// reduce, half the host's Horner chain.  Interleaved A/B against the unsplit pipeline (tools/ab_glv.py): 2^10 -12.8 %, 2^14 -11.9 %,
// 2^16 -12.3 %, 2^17 -12.2 %, 2^18 -9.6 %, 2^19 +1.1 %, 2^20 -0.4 %, 2^21 +9.2 %, 2^22 +6.8 % (twice the base records to gather
// from, k_accumulate unchanged, and the fixed costs it halves no longer matter).  With the chunk length following the bucket
// occupancy: 2^18 -10.4 %, 2^19 -3.4 %, 2^20 -0.3 %, 2^21 +3.6 %, 2^22 +9.9 %  => on by default up to 2^19 points.
constexpr size_t GLV_MAX_POINTS = (size_t)1 << 19;
uint32_t plan_window_bits_glv(size_t n, bool is_signed) {
    // measured (tools/sweep_c.py, split on): 2^10 c = 9/10 0.247/0.249 ms; 2^12 10/11 0.286/0.281; 2^13 10 0.308 (16: 0.364);
    // 2^14 10 0.330 (16: 0.440); 2^15 12 0.365 (16: 0.435); 2^16 16 0.420 (13: 0.439); 2^17 16 0.504 (13: 0.541); 2^18 16 0.671
    // (15: 1.06); 2^20 16 1.681.  127 = 7*16 + 15: eight windows, the top one 15 bits wide -- no degenerate window.
    uint32_t c = n <= ((size_t)1 << 14) ? 10u : n <= ((size_t)1 << 15) ? 12u : 16u;
    if (!is_signed && c > 15u) c = 15u;
    return c;
}
int32_t make_plan(size_t n, uint32_t window_bits, uint32_t flags, msm_plan_t* out) {
    if (flags & ~(MSM_FLAG_UNSIGNED_DIGITS | MSM_FLAG_NO_GLV)) return MSM_ERR_BAD_ARG;
    bool is_signed = !(flags & MSM_FLAG_UNSIGNED_DIGITS);
    size_t glv_max = GLV_MAX_POINTS;
    if (const char* e = std::getenv("MSM_HIP_GLV_MAX_LOG2")) glv_max = (size_t)1 << std::min(23, std::max(0, std::atoi(e)));  // A/B knob
    bool use_glv = !(flags & MSM_FLAG_NO_GLV) && n <= glv_max;
    uint32_t c = window_bits ? window_bits : (use_glv ? plan_window_bits_glv(n, is_signed) : plan_window_bits(n, is_signed));
    if (c < 2 || c > 20) return MSM_ERR_BAD_ARG;
    if ((is_signed ? c - 1 : c) > 17) use_glv = false;  // windows wider than the LDS sort covers (forced c >= 19) run unsplit
    const uint32_t bits = use_glv ? (uint32_t)glv::SPLIT_BITS : 254u;
    out->window_bits = c;
    out->signed_digits = is_signed;
    out->glv = use_glv ? 1u : 0u;
    out->scalar_bits = bits;
    out->virtual_points = use_glv ? 2 * (uint64_t)n : (uint64_t)n;
    // signed: one spare window position so the top digit never overflows (r < 2^254, |k_j| < 2^127): W = floor(bits/c) + 1
    out->num_windows = is_signed ? (bits / c + 1) : ((bits + c - 1) / c);
    out->num_buckets = is_signed ? (1u << (c - 1)) : (1u << c);
    size_t nv = (size_t)out->virtual_points;
    size_t pairs = (size_t)out->num_windows * nv;
    size_t tb = (size_t)out->num_windows * out->num_buckets;
    out->workspace_bytes = nv * 64 + n * (32 + 1) + pairs * 12 + tb * (8 + 144) + tb * 144 * 3 / 2;
    return MSM_OK;
}
// does a call on n points (context configuration + per-call extra flags) use the GLV split, i.e. 2n base records?
inline bool plan_glv(const msm_ctx* c, size_t n, uint32_t extra_flags = 0) {
    msm_plan_t pl;
    return make_plan(n, c->cfg.window_bits, c->cfg.flags | extra_flags, &pl) == MSM_OK && pl.glv != 0;
}
constexpr size_t XB = msmk::XW * 4;               // bytes per XYZZ record (4 coordinates x 9 x 29-bit limbs)
constexpr size_t LDS_HIST_BYTES = 128 * 1024;   // one window's bucket histogram must fit here for the LDS sort path
constexpr size_t MAX_QSUM_POINTS = 128 * 21;  // W <= 128 windows (c >= 2), kb + 1 <= 21 bit sums each

uint32_t ilog2(uint32_t v) {
    uint32_t l = 0;
    while ((1u << (l + 1)) <= v) l++;
    return l;
}

// elapsed ms between two stage events, 0 when stage timing is off
float stage_ms(const msm_ctx* c, int a, int b) {
    float ms = 0;
    if (c->stage_timing) (void)hipEventElapsedTime(&ms, c->ev[a], c->ev[b]);
    return ms;
}

inline dim3 grid1(size_t n, unsigned block) { return dim3((unsigned)((n + block - 1) / block)); }

int32_t ensure_pow2_table(msm_ctx* c) {
    if (c->pow2_ready) return MSM_OK;
    using namespace hostg1;
    std::vector<uint32_t> tab((size_t)msmk::SCALAR_BITS * 16);
    Jac g{ONE, dbl(ONE), ONE};  // (1, 2) -- SH/constants.metal:121-174
    for (int j = 0; j < msmk::SCALAR_BITS; j++) {
        Fq zi = inv(g.z), zi2 = sqr(zi);
        Fq x = mul(g.x, zi2), y = mul(g.y, mul(zi2, zi));  // Montgomery affine
        store_words(&tab[(size_t)j * 16], x);
        store_words(&tab[(size_t)j * 16 + 8], y);
        g = jdbl(g);
    }
    int32_t rc = ensure(c, c->pow2, tab.size() * 4);
    if (rc) return rc;
    void* raw = nullptr;
    HIPCHK(c, hipMalloc(&raw, tab.size() * 4));
    HIPCHK(c, hipMemcpyAsync(raw, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, c->stream));
    msmk::k_convert_bases<<<grid1(2 * (size_t)msmk::SCALAR_BITS, 64), 64, 0, c->stream>>>((const uint32_t*)raw, (uint32_t*)c->pow2.p,
                                                                                     (uint32_t)msmk::SCALAR_BITS, 1u, 0u);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(raw));
    c->pow2_ready = true;
    return MSM_OK;
}

void finish_outputs(const hostg1::Jac& r, uint32_t* out_jac, uint32_t* out_aff, uint8_t* out_inf) {
    if (out_jac) hostg1::store_jac(out_jac, r);
    if (out_inf) *out_inf = hostg1::is_identity(r) ? 1 : 0;
    if (out_aff) {  // the only inversion of the whole call (~10 us): callers that want the reference's result type
                    // (Jacobian, metal_msm.rs:228-241) pass NULL and skip it
        hostg1::Fq x, y;
        (void)hostg1::to_affine_std(r, x, y);
        hostg1::store_words(out_aff, x);
        hostg1::store_words(out_aff + 8, y);
    }
}

struct PipeGeom {
    uint32_t W, nb, cbits, kb;
};

// Queue the whole device pipeline for one (chunk of an) MSM on stream st; the W*(kb+1) bit sums and the flag words
// are written by the last kernel straight into h_qsums_dst / h_flags_dst (pinned host memory).  No host synchronisation here.
// d_bases: INTERNAL-domain records; with the GLV split (make_plan) 2*n_real of them, phi(P_i) at index n_real + i.
int32_t enqueue_pipeline(msm_ctx* c, const uint32_t* d_bases, const uint8_t* d_inf, const uint32_t* d_scalars, size_t n_real,
                         hipStream_t st, uint32_t* h_qsums_dst, uint32_t* h_flags_dst, PipeGeom* geom, uint32_t scalars_mont = 0,
                         hipEvent_t bases_ready = nullptr, uint32_t extra_flags = 0) {
    if (n_real > 0x3FFFFFFFull) return fail(c, MSM_ERR_BAD_ARG, "n = %zu exceeds 2^30-1 points per context call", n_real);
    msm_plan_t pl;
    int32_t rc = make_plan(n_real, c->cfg.window_bits, c->cfg.flags | extra_flags, &pl);
    if (rc) return fail(c, rc, "bad window_bits/flags (%u, 0x%x)", c->cfg.window_bits, c->cfg.flags);
    const size_t n = (size_t)pl.virtual_points;  // what the sort, the accumulation and the reduction see
    const uint32_t W = pl.num_windows, nb = pl.num_buckets, cbits = pl.window_bits;
    const size_t pairs = (size_t)W * n, tb = (size_t)W * nb;
    if (pairs > 0xFFFFFFFFull) return fail(c, MSM_ERR_BAD_ARG, "n*W = %zu does not fit 32-bit offsets", pairs);
    const uint32_t kb = ilog2(nb), kb_lo = kb / 2, kb_hi = kb - kb_lo;  // bucket index = hi * n_lo + lo
    const uint32_t n_lo = 1u << kb_lo, n_hi = 1u << kb_hi;
    const uint32_t ntiles = (uint32_t)((tb + msmk::SCAN_TILE - 1) / msmk::SCAN_TILE);
    // sorted entries folded by one k_accumulate thread: ~2^19 chunks per call (2.7 rounds of the 196608 threads that
    // 3 wavefronts/SIMD hold) keep the tail short, and the chunk grows with N so that buckets (mean n / nb entries)
    // are cut into few pieces for k_combine.  Measured sweep at N = 2^20: L = 32 (profiles/NOTES_r1.md).
    // 16 from 2^13 points up (8 loses there: more buckets are cut 3+ times than the finer granularity wins back); tiny instances
    // (<= 2^17 sorted entries: a quarter of the SIMDs would hold a wavefront at 16) take 8: 0.262 vs 0.296 ms at 2^10
    uint32_t chunk_len = pairs <= ((size_t)1 << 17) ? 8 : 16;
    // ... and from there the chunk follows the mean bucket occupancy n/nb (32 at 2^20 unsplit, 64 with the GLV split: measured
    // 1.728 ms at L = 32 against 1.698 at 64), as long as ~2^17 chunks remain to fill the chip
    while (chunk_len < 1024 && chunk_len < n / nb && pairs / (chunk_len * 2) >= 131072) chunk_len *= 2;
    if (const char* e = std::getenv("MSM_HIP_CHUNK_LEN")) {  // tuning knob (any value >= 1 is correct)
        int v = std::atoi(e);
        if (v >= 1 && v <= 4096) chunk_len = (uint32_t)v;
    }
    const size_t nchunks_max = (pairs + chunk_len - 1) / chunk_len;
    if ((rc = ensure(c, c->digits, pairs * 4))) return rc;
    if ((rc = ensure(c, c->sorted, pairs * 4))) return rc;
    if ((rc = ensure(c, c->hist, tb * 4))) return rc;
    if ((rc = ensure(c, c->offsets, (tb + 1) * 4))) return rc;
    if ((rc = ensure(c, c->blocksums, ((size_t)ntiles + 1) * 4))) return rc;
    if ((rc = ensure(c, c->buckets, tb * XB))) return rc;
    if ((rc = ensure(c, c->heads, nchunks_max * XB > pairs * 4 ? nchunks_max * XB : pairs * 4))) return rc;  // also stages the 2-level sort
    if ((rc = ensure(c, c->tails, nchunks_max * XB))) return rc;
    if ((rc = ensure(c, c->chunkmap, nchunks_max * 4))) return rc;
    {   // a long bucket owns >= LONG_SPAN chunks and gets one (bucket, segment) entry per LONG_SEG pieces
        const size_t entries = nchunks_max / msmk::LONG_SPAN + nchunks_max / msmk::LONG_SEG + 32;
        if ((rc = ensure(c, c->longlist, entries * 8))) return rc;
        const size_t had = c->longdone.cap;
        if ((rc = ensure(c, c->longdone, entries * 4))) return rc;
        if (c->longdone.cap != had) HIPCHK(c, hipMemsetAsync(c->longdone.p, 0, c->longdone.cap, st));  // self-cleaning afterwards
    }
    if ((rc = ensure(c, c->midlist, (nchunks_max / 2 + 16) * 4))) return rc;                  // a listed bucket owns >= 2 chunk borders
    if ((rc = ensure(c, c->rc, (tb + tb / 2 + 4) * XB))) return rc;  // two families x (1/2 + 1/4) ping-pong levels
    if ((rc = ensure(c, c->flags, 64))) return rc;

    uint32_t* hist = (uint32_t*)c->hist.p;
    uint32_t* offsets = (uint32_t*)c->offsets.p;
    uint32_t* flags = (uint32_t*)c->flags.p;
    // Counting-sort plan: when one window's histogram fits LDS (nb <= 32768) the bucket counts and arrival
    // ranks come from per-tile LDS histograms, otherwise from device-scope atomics in k_decompose.
    const bool tiled = (size_t)nb * 4 <= LDS_HIST_BYTES;
    // two-level LDS sort: coarse = top bits of the bucket index, fine = the rest (<= 7 bits)
    // (8 coarse bits up to n = 2^21, then 9 and 10, so that a (window, coarse bin) region stays ~8192 elements and
    // fits the fine sort's LDS staging)
    uint32_t coarse_bits = 8;
    while (coarse_bits < 10 && (n >> coarse_bits) > 8192) coarse_bits++;
    if (kb > coarse_bits + 7) coarse_bits = std::min(10u, kb - 7);  // wide windows (up to 2^17 buckets): the fine part stays 7 bits
    if (coarse_bits > kb) coarse_bits = kb;
    const uint32_t fine_bits = kb - coarse_bits, idx_bits = 31 - fine_bits;
    const uint32_t ncoarse = 1u << coarse_bits;
    const bool two_level = fine_bits <= 7 && n <= ((size_t)1 << idx_bits) && !std::getenv("MSM_HIP_DIRECT_SCATTER");
    const bool lds_counts = two_level || tiled;  // no device-scope histogram / rank atomics in k_decompose
    const uint32_t NS = (uint32_t)((n + msmk::SUBTILE - 1) / msmk::SUBTILE);
    uint32_t T = 1, tile_len = (uint32_t)n;
    if (!lds_counts && (rc = ensure(c, c->ranks, pairs * 4))) return rc;
    if (two_level) {
        if ((rc = ensure(c, c->bigslot, (size_t)W * ncoarse * 4))) return rc;
        if ((rc = ensure(c, c->big, msmk::BIG_WORDS * 4))) return rc;
        if ((rc = ensure(c, c->ccounts, (size_t)W * ncoarse * NS * 4))) return rc;
        if ((rc = ensure(c, c->cregion, ((size_t)W * ncoarse * 2 + 2) * 4))) return rc;
    } else if (tiled) {
        T = (uint32_t)((n + 65535) / 65536);
        if (T > 64) T = 64;
        tile_len = (uint32_t)((n + T - 1) / T);
        if ((rc = ensure(c, c->tilecounts, (size_t)W * T * nb * 4))) return rc;
    } else {
        HIPCHK(c, hipMemsetAsync(hist, 0, tb * 4, st));
    }
    HIPCHK(c, hipMemsetAsync(flags, 0, 64, st));
    if (c->stage_timing) HIPCHK(c, hipEventRecord(c->ev[EV_CONVERT], st));
    // K1b: digits + signed recode (with the GLV split: two 127-bit halves per scalar, 2*n_real digit columns)
    {
        uint32_t *dg = (uint32_t*)c->digits.p, *rk = (uint32_t*)c->ranks.p;
        dim3 g = grid1(n_real, 256);
        const uint32_t nr = (uint32_t)n_real;
        if (pl.glv) {
            if (!lds_counts) return fail(c, MSM_ERR_BAD_ARG, "window_bits %u needs the non-GLV path (MSM_FLAG_NO_GLV)", cbits);
            if (pl.signed_digits) msmk::k_decompose_glv<true><<<g, 256, 0, st>>>(d_scalars, d_inf, nr, cbits, W, dg, flags, scalars_mont);
            else msmk::k_decompose_glv<false><<<g, 256, 0, st>>>(d_scalars, d_inf, nr, cbits, W, dg, flags, scalars_mont);
        } else if (pl.signed_digits && lds_counts) msmk::k_decompose<true, false><<<g, 256, 0, st>>>(d_scalars, d_inf, nr, cbits, W, nb, hist, dg, rk, flags, scalars_mont);
        else if (pl.signed_digits) msmk::k_decompose<true, true><<<g, 256, 0, st>>>(d_scalars, d_inf, nr, cbits, W, nb, hist, dg, rk, flags, scalars_mont);
        else if (lds_counts) msmk::k_decompose<false, false><<<g, 256, 0, st>>>(d_scalars, d_inf, nr, cbits, W, nb, hist, dg, rk, flags, scalars_mont);
        else msmk::k_decompose<false, true><<<g, 256, 0, st>>>(d_scalars, d_inf, nr, cbits, W, nb, hist, dg, rk, flags, scalars_mont);
    }
    if (c->stage_timing) HIPCHK(c, hipEventRecord(c->ev[EV_DECOMP], st));
    if (two_level) {
        const uint32_t nregions = W * ncoarse;
        uint32_t* counts = (uint32_t*)c->ccounts.p;
        uint32_t* rtotal = (uint32_t*)c->cregion.p;
        uint32_t* rstart = rtotal + nregions;
        uint32_t* tmp = (uint32_t*)c->heads.p;  // staging copy; k_accumulate only writes heads later
        msmk::k_coarse_hist<<<dim3(NS, W), msmk::TILE_BLOCK, 0, st>>>((uint32_t*)c->digits.p, counts, (uint32_t)n, fine_bits, ncoarse, NS);
        msmk::k_coarse_prefix<<<grid1(nregions, 256), 256, 0, st>>>(counts, rtotal, NS, ncoarse, nregions);
        // regions too large for one workgroup's staging area are cut into batches that worker blocks share (skewed scalars)
        const uint32_t fine_block = (n >> coarse_bits) <= 1024 ? 256u : (n >> coarse_bits) <= 2048 ? 512u : 1024u;
        const uint32_t fine_cap = fine_block * 16u;
        uint32_t* bigslot = (uint32_t*)c->bigslot.p;
        uint32_t* big = (uint32_t*)c->big.p;
        msmk::k_coarse_starts<<<1, msmk::SCAN_BLOCK, 0, st>>>(rtotal, rstart, nregions, flags + 4, offsets + tb, bigslot, big, fine_cap);
        msmk::k_coarse_scatter<<<dim3(NS, W), msmk::TILE_BLOCK, 0, st>>>((uint32_t*)c->digits.p, counts, rstart, tmp, (uint32_t)n, fine_bits,
                                                                       idx_bits, ncoarse, NS);
        // workgroup size by mean region size (a workgroup stages up to 16 elements per thread); grid.x = the window's regions +
        // BIG_WORKERS_X worker blocks for the batches of oversized regions, which k_big_place then places
        const dim3 gf(ncoarse + msmk::BIG_WORKERS_X, W), gp(msmk::BIG_WORKERS_X, W);
        uint32_t* srt = (uint32_t*)c->sorted.p;
        if (fine_block == 256) {
            msmk::k_fine_sort<256><<<gf, 256, 0, st>>>(tmp, rstart, offsets, srt, nb, fine_bits, idx_bits, ncoarse, bigslot, big);
            msmk::k_big_place<256><<<gp, 256, 0, st>>>(tmp, rstart, offsets, srt, nb, fine_bits, idx_bits, ncoarse, bigslot, big);
        } else if (fine_block == 512) {  // (up to a mean of 2048: at 4096 the 512-thread variant is 1.5 us faster on uniform scalars only)
            msmk::k_fine_sort<512><<<gf, 512, 0, st>>>(tmp, rstart, offsets, srt, nb, fine_bits, idx_bits, ncoarse, bigslot, big);
            msmk::k_big_place<512><<<gp, 512, 0, st>>>(tmp, rstart, offsets, srt, nb, fine_bits, idx_bits, ncoarse, bigslot, big);
        } else {
            msmk::k_fine_sort<1024><<<gf, 1024, 0, st>>>(tmp, rstart, offsets, srt, nb, fine_bits, idx_bits, ncoarse, bigslot, big);
            msmk::k_big_place<1024><<<gp, 1024, 0, st>>>(tmp, rstart, offsets, srt, nb, fine_bits, idx_bits, ncoarse, bigslot, big);
        }
    } else {
        // K2/1: per-tile LDS histograms, then per-bucket prefix over tiles
        if (tiled) {
            msmk::k_tile_hist<<<dim3(T, W), msmk::TILE_BLOCK, (size_t)nb * 4, st>>>((uint32_t*)c->digits.p, (uint32_t*)c->tilecounts.p,
                                                                                  (uint32_t)n, nb, tile_len, T);
            msmk::k_tile_prefix<<<grid1(tb, 256), 256, 0, st>>>((uint32_t*)c->tilecounts.p, hist, nb, T, (uint32_t)tb);
        }
        // K2/2: bucket offsets
        msmk::k_scan_tiles<<<ntiles, msmk::SCAN_BLOCK, 0, st>>>(hist, offsets, (uint32_t*)c->blocksums.p, (uint32_t)tb);
        msmk::k_scan_block_sums<<<1, msmk::SCAN_BLOCK, 0, st>>>((uint32_t*)c->blocksums.p, ntiles, flags + 4);
        msmk::k_scan_add<<<grid1(tb, 256), 256, 0, st>>>(offsets, (uint32_t*)c->blocksums.p, (uint32_t)tb, flags + 4);
        // K2/3: scatter
        if (tiled) {
            msmk::k_tile_scatter<<<dim3(T, W), msmk::TILE_BLOCK, (size_t)nb * 4, st>>>((uint32_t*)c->digits.p, offsets, (uint32_t*)c->tilecounts.p,
                                                                                     (uint32_t*)c->sorted.p, (uint32_t)n, nb, tile_len, T);
        } else {
            dim3 g((unsigned)((n + 255) / 256), W);
            msmk::k_scatter<<<g, 256, 0, st>>>((uint32_t*)c->digits.p, (uint32_t*)c->ranks.p, offsets, (uint32_t*)c->sorted.p, (uint32_t)n, nb);
        }
    }
    if (c->stage_timing) HIPCHK(c, hipEventRecord(c->ev[EV_SORT], st));
    // K3: bucket accumulation (the graded kernel) -- bracketed by its own events on its own stream
    msmk::k_chunk_map<<<grid1(tb, 1024), 1024, 0, st>>>(offsets, (uint32_t*)c->chunkmap.p, (uint32_t)tb, chunk_len, flags + 8,
                                                      (uint32_t*)c->longlist.p, flags + 9, (uint32_t*)c->midlist.p);
    if (bases_ready) HIPCHK(c, hipStreamWaitEvent(st, bases_ready, 0));  // d_bases is being converted on another stream
    HIPCHK(c, hipEventRecord(c->ev[EV_ACC0], st));
    msmk::k_accumulate<<<grid1(nchunks_max, 256), 256, 0, st>>>(d_bases, (uint32_t*)c->sorted.p, offsets, (uint32_t*)c->chunkmap.p,
                                                              (uint32_t*)c->buckets.p, (uint32_t*)c->heads.p, (uint32_t*)c->tails.p,
                                                              flags + 4, chunk_len, (uint32_t)tb);
    HIPCHK(c, hipEventRecord(c->ev[EV_ACC1], st));
    msmk::k_combine<<<dim3(msmk::MID_BLOCKS + (unsigned)((tb + 255) / 256)), 256, 0, st>>>(offsets, (uint32_t*)c->heads.p, (uint32_t*)c->tails.p,
                                                                                         (uint32_t*)c->buckets.p, (uint32_t)tb, chunk_len, flags + 9,
                                                                                         (uint32_t*)c->midlist.p);
    msmk::k_combine_long<<<1024, 512, 0, st>>>(offsets, (uint32_t*)c->heads.p, (uint32_t*)c->tails.p, (uint32_t*)c->buckets.p, flags + 8,
                                               (uint32_t*)c->longlist.p, (uint32_t*)c->longdone.p, chunk_len);
    // K4/K5: bucket reduction -- plain row/column sums by dense pairwise levels, then per-bit sums; the weights
    // are applied on the host
    {
        const uint32_t* bk = (const uint32_t*)c->buckets.p;
        // ping-pong buffers per family: [0, tb/2) and [tb/2, tb/2 + tb/4) elements
        uint32_t* rbuf[2] = {(uint32_t*)c->rc.p, (uint32_t*)c->rc.p + (tb / 2) * msmk::XW};
        uint32_t* cbuf[2] = {(uint32_t*)c->rc.p + (tb / 2 + tb / 4 + 1) * msmk::XW, (uint32_t*)c->rc.p + (tb + tb / 4 + 1) * msmk::XW};
        const uint32_t *rin = bk, *cin = bk;
        size_t rn = tb, cn = tb;  // current element counts
        uint32_t levels = kb_hi > kb_lo ? kb_hi : kb_lo;
        for (uint32_t l = 0; l < levels; l++) {
            msmk::pair_job ja{nullptr, nullptr, 0, 1}, jb{nullptr, nullptr, 0, 1};
            if (l < kb_lo) {
                rn /= 2;
                ja = msmk::pair_job{rin, rbuf[l & 1], (uint32_t)rn, 1};
                rin = rbuf[l & 1];
            }
            if (l < kb_hi) {
                cn /= 2;
                jb = msmk::pair_job{cin, cbuf[l & 1], (uint32_t)cn, n_lo};
                cin = cbuf[l & 1];
            }
            // levels with fewer additions than an eighth of the lanes the chip keeps resident: eight lanes per addition
            const size_t nadds = (size_t)ja.n_out + jb.n_out;
            if (nadds <= c->wide_max) msmk::k_pair_level_wide<<<grid1(nadds * msmk::WIDE_LANES, 256), 256, 0, st>>>(ja, jb);
            else msmk::k_pair_level<<<grid1(nadds, 256), 256, 0, st>>>(ja, jb);
        }
        // the bit sums (and the flag words) are written by the kernel straight into the caller's PINNED host buffers:
        // a D2H copy engine transfer started ~11 us after the kernel and took two launches (24 KB + 32 B)
        uint32_t *q_dev = nullptr, *f_dev = nullptr;
        HIPCHK(c, hipHostGetDevicePointer((void**)&q_dev, h_qsums_dst, 0));
        HIPCHK(c, hipHostGetDevicePointer((void**)&f_dev, h_flags_dst, 0));
        if (c->wide_max && n_hi / 2 <= msmk::WIDE_TREE_MAX && n_lo <= msmk::WIDE_TREE_MAX)
            msmk::k_reduce_bits_wide<<<W * (kb + 1), 512, 0, st>>>(rin, cin, q_dev, n_hi, n_lo, kb_lo, kb, flags, f_dev);
        else
            msmk::k_reduce_bits<<<W * (kb + 1), 64, 0, st>>>(rin, cin, q_dev, n_hi, n_lo, kb_lo, kb, flags, f_dev);
    }
    if (c->stage_timing) HIPCHK(c, hipEventRecord(c->ev[EV_REDUCE], st));
    *geom = PipeGeom{W, nb, cbits, kb};
    return MSM_OK;
}

// final_reduction (metal_msm.rs:204-261) on the CPU.  With S_w = Q_all,w + sum_u 2^u Q_w,u the result is
//     sum_w 2^(c*w) S_w = sum over bit positions p = c*w + u of 2^p * (Q_w,u  [+ Q_all,w when u == 0]),
// ONE Horner chain over the ~254 positions (one doubling and about one addition per position) instead of the reference's
// chain per window plus c doublings between windows (metal_msm.rs:249-258).  The chain is cut into a few segments of
// geometrically shrinking length (a segment starting at position lo pays lo extra doublings to shift its sum), one per
// host thread: 2 threads reach ~60 % of the serial time, 4 threads ~45 %, more add nothing because the shift of the top
// segment is serial.  TWO threads by default: every further worker lowers the median by a few microseconds and raises the
// MEAN through 2-8 ms outliers in ~1.3 % of the calls (busy hosts; a pool of 15: 2.5 %) -- tools/step_jitter.py.
hostg1::Jac host_finish(msm_ctx* c, const uint32_t* h_qsums, const PipeGeom& g) {
    const uint32_t W = g.W, kb = g.kb, cbits = g.cbits;
    const uint32_t npos = cbits * (W - 1) + (kb > 0 ? kb : 1);  // positions 0 .. npos-1 carry terms
    auto segment = [&](uint32_t lo, uint32_t hi) {             // sum over p in [lo, hi) of 2^p * term(p)
        hostg1::Jac acc = hostg1::identity();
        for (uint32_t p = hi; p-- > lo;) {
            acc = hostg1::jdbl(acc);
            const uint32_t w = p / cbits, u = p % cbits;
            const uint32_t* qw = h_qsums + (size_t)w * (kb + 1) * 24;
            if (u < kb) acc = hostg1::jadd(acc, hostg1::load_jac(qw + (size_t)u * 24));
            if (u == 0) acc = hostg1::jadd(acc, hostg1::load_jac(qw + (size_t)kb * 24));
        }
        for (uint32_t k = 0; k < lo; k++) acc = hostg1::jdbl(acc);
        return acc;
    };
    const int nseg = c->pool ? std::min<int>(c->pool->size() + 1, 8) : 1;
    if (nseg == 1 || npos < 32) return segment(0, npos);
    // segment k has length proportional to 0.7^k (a doubling costs ~0.3 of a position's doubling + addition)
    uint32_t bound[9];
    double tot = 0, wgt = 1;
    for (int k = 0; k < nseg; k++, wgt *= 0.7) tot += wgt;
    double run = 0;
    wgt = 1;
    bound[0] = 0;
    for (int k = 0; k < nseg; k++, wgt *= 0.7) {
        run += wgt;
        bound[k + 1] = k + 1 == nseg ? npos : (uint32_t)(npos * (run / tot) + 0.5);
    }
    std::vector<hostg1::Jac> part((size_t)nseg);
    c->pool->run(nseg, [&](int k) { part[(size_t)k] = segment(bound[k], bound[k + 1]); });  // job 0 (the longest) is taken first
    hostg1::Jac total = part[0];
    for (int k = 1; k < nseg; k++) total = hostg1::jadd(total, part[(size_t)k]);
    return total;
}

int32_t check_flags(msm_ctx* c, const uint32_t* h_flags) {
    if (h_flags[0] & 1u) return fail(c, MSM_ERR_BAD_ARG, "a scalar is >= 2^254 (not a canonical Fr element)");
    if (h_flags[0] & 2u) return fail(c, MSM_ERR_HIP, "internal: signed-digit carry out of the top window");
    if (h_flags[0] & 4u) return fail(c, MSM_ERR_HIP, "internal: a GLV half exceeds 127 bits");
    return MSM_OK;
}

// The pipeline proper: everything in HBM, one stream.  d_bases: INTERNAL-domain packed coordinates.
int32_t run_pipeline(msm_ctx* c, const uint32_t* d_bases, const uint8_t* d_inf, const uint32_t* d_scalars, size_t n,
                     hipStream_t st, uint32_t* out_jac, uint32_t* out_aff, uint8_t* out_inf, uint32_t scalars_mont = 0,
                     hipEvent_t bases_ready = nullptr, uint32_t extra_flags = 0) {
    PipeGeom g;
    int32_t rc = enqueue_pipeline(c, d_bases, d_inf, d_scalars, n, st, c->h_qsums, c->h_flags, &g, scalars_mont, bases_ready, extra_flags);
    if (rc) return rc;
    if (c->pool && n >= 256) c->pool->arm();  // workers wake up while the GPU works
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    auto t_fin0 = std::chrono::steady_clock::now();
    if ((rc = check_flags(c, c->h_flags))) return rc;
    hostg1::Jac total = host_finish(c, c->h_qsums, g);
    finish_outputs(total, out_jac, out_aff, out_inf);
    auto t_fin1 = std::chrono::steady_clock::now();
    // timings
    float ms = 0;
    msm_timings_t& tm = c->tm;
    tm.decompose_ms = stage_ms(c, EV_CONVERT, EV_DECOMP);
    tm.sort_ms = stage_ms(c, EV_DECOMP, EV_SORT);
    (void)hipEventElapsedTime(&ms, c->ev[EV_ACC0], c->ev[EV_ACC1]);
    tm.accumulate_ms = ms;
    c->acc_ms_sum += ms;
    c->acc_launches += 1;
    tm.reduce_ms = stage_ms(c, EV_ACC1, EV_REDUCE);
    tm.finish_ms = std::chrono::duration<float, std::milli>(t_fin1 - t_fin0).count();
    tm.num_points = n;
    tm.num_adds = c->h_flags[4];
    return MSM_OK;
}

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int32_t check_common(msm_ctx* c, const void* a, const void* b, size_t n) {
    if (!c) return MSM_ERR_BAD_ARG;
    if (n == 0) return fail(c, MSM_ERR_EMPTY, "Empty input");  // metal_msm.rs:647-649
    if (!a || !b) return fail(c, MSM_ERR_BAD_ARG, "NULL bases/scalars pointer");
    return MSM_OK;
}

// raw caller coordinates (host) -> c->bases (staging) -> c->ibases (internal domain)
int32_t upload_bases_locked(msm_ctx* c, const uint32_t* bases_xy, uint32_t form, const uint8_t* inf_mask, size_t n) {
    if (form != MSM_FORM_STD && form != MSM_FORM_MONT) return fail(c, MSM_ERR_BAD_ARG, "unknown base_form %u", form);
    int32_t rc;
    const bool glv = plan_glv(c, n);
    if ((rc = ensure(c, c->bases, n * 64))) return rc;
    if ((rc = ensure(c, c->ibases, (glv ? 2 : 1) * n * 64))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->bases.p, bases_xy, n * 64, hipMemcpyHostToDevice, c->stream));
    if (inf_mask) {
        if ((rc = ensure(c, c->inf, n))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->inf.p, inf_mask, n, hipMemcpyHostToDevice, c->stream));
    }
    if (c->stage_timing) HIPCHK(c, hipEventRecord(c->ev[EV_H2D], c->stream));
    msmk::k_convert_bases<<<grid1(2 * n, 256), 256, 0, c->stream>>>((const uint32_t*)c->bases.p, (uint32_t*)c->ibases.p, (uint32_t)n,
                                                                  form == MSM_FORM_MONT ? 1u : 0u, glv ? 1u : 0u);
    return MSM_OK;
}

// BASELINE config 5: the instance does not have to be resident.  The point range is cut into chunks of 2^k points;
// chunk j+1 travels host->HBM on the copy stream while the full pipeline of chunk j runs on the compute stream
// (MSM is linear, so every chunk is an independent MSM and the partial results are added on the host).  Inputs are
// double-buffered; nothing but W*(kb+1) bit sums per chunk comes back.
int32_t run_streamed(msm_ctx* c, const uint32_t* bases_xy, uint32_t form, const uint8_t* inf_mask, const uint32_t* scalars,
                     size_t n, size_t chunk, uint32_t* out_jac, uint32_t* out_aff, uint8_t* out_inf) {
    if (form != MSM_FORM_STD && form != MSM_FORM_MONT) return fail(c, MSM_ERR_BAD_ARG, "unknown base_form %u", form);
    const size_t nchunks = (n + chunk - 1) / chunk;
    const size_t slot_words = MAX_QSUM_POINTS * 24 + 8;
    int32_t rc;
    if (c->h_sq_cap < nchunks * slot_words) {
        if (c->h_sq) HIPCHK(c, hipHostFree(c->h_sq));
        c->h_sq = nullptr;
        c->h_sq_cap = 0;
        HIPCHK(c, hipHostMalloc((void**)&c->h_sq, nchunks * slot_words * 4, hipHostMallocDefault));
        c->h_sq_cap = nchunks * slot_words;
    }
    for (int s = 0; s < 2; s++) {
        if ((rc = ensure(c, c->sbases[s], chunk * 64))) return rc;
        if ((rc = ensure(c, c->sscalars[s], chunk * 32))) return rc;
        if (inf_mask && (rc = ensure(c, c->sinf[s], chunk))) return rc;
    }
    if ((rc = ensure(c, c->ibases, 2 * chunk * 64))) return rc;  // room for the phi records of a GLV chunk
    std::vector<PipeGeom> geom(nchunks);
    hipStream_t st = c->stream, cs = c->copy_stream;
    for (size_t j = 0; j < nchunks; j++) {
        const int s = (int)(j & 1);
        const size_t lo = j * chunk, cnt = (lo + chunk <= n) ? chunk : n - lo;
        if (j >= 2) HIPCHK(c, hipStreamWaitEvent(cs, c->ev_free[s], 0));  // the pipeline that read this slot is done
        HIPCHK(c, hipMemcpyAsync(c->sbases[s].p, bases_xy + lo * 16, cnt * 64, hipMemcpyHostToDevice, cs));
        HIPCHK(c, hipMemcpyAsync(c->sscalars[s].p, scalars + lo * 8, cnt * 32, hipMemcpyHostToDevice, cs));
        if (inf_mask) HIPCHK(c, hipMemcpyAsync(c->sinf[s].p, inf_mask + lo, cnt, hipMemcpyHostToDevice, cs));
        HIPCHK(c, hipEventRecord(c->ev_copied[s], cs));
        HIPCHK(c, hipStreamWaitEvent(st, c->ev_copied[s], 0));
        msmk::k_convert_bases<<<grid1(2 * cnt, 256), 256, 0, st>>>((const uint32_t*)c->sbases[s].p, (uint32_t*)c->ibases.p, (uint32_t)cnt,
                                                                 form == MSM_FORM_MONT ? 1u : 0u, plan_glv(c, cnt) ? 1u : 0u);
        uint32_t* slot = c->h_sq + j * slot_words;
        rc = enqueue_pipeline(c, (const uint32_t*)c->ibases.p, inf_mask ? (const uint8_t*)c->sinf[s].p : nullptr,
                              (const uint32_t*)c->sscalars[s].p, cnt, st, slot + 8, slot, &geom[j]);
        if (rc) return rc;
        HIPCHK(c, hipEventRecord(c->ev_free[s], st));
    }
    HIPCHK(c, hipStreamSynchronize(st));
    HIPCHK(c, hipGetLastError());
    hostg1::Jac total = hostg1::identity();
    uint64_t adds = 0;
    for (size_t j = 0; j < nchunks; j++) {
        const uint32_t* slot = c->h_sq + j * slot_words;
        if ((rc = check_flags(c, slot))) return rc;
        adds += slot[4];
        total = hostg1::jadd(total, host_finish(c, slot + 8, geom[j]));
    }
    finish_outputs(total, out_jac, out_aff, out_inf);
    c->tm = msm_timings_t{};
    c->tm.num_points = n;
    c->tm.num_adds = adds;
    return MSM_OK;
}

}  // namespace

extern "C" {

uint32_t msm_abi_version(void) { return MSM_HIP_ABI_VERSION; }

const char* msm_last_error(const msm_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int32_t msm_ctx_create(const msm_config_t* cfg, msm_ctx** out) {
    if (!out) return fail(nullptr, MSM_ERR_BAD_ARG, "out == NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, MSM_ERR_NO_DEVICE, "no HIP device visible: this engine has no CPU fallback");
    msm_config_t c0{};
    c0.device = -1;
    if (cfg) c0 = *cfg;
    msm_plan_t probe;
    if (make_plan(1, c0.window_bits, c0.flags, &probe) != MSM_OK)
        return fail(nullptr, MSM_ERR_BAD_ARG, "bad window_bits/flags (%u, 0x%x)", c0.window_bits, c0.flags);
    if (c0.stream_chunk_log2 && (c0.stream_chunk_log2 < 8 || c0.stream_chunk_log2 > 28))
        return fail(nullptr, MSM_ERR_BAD_ARG, "stream_chunk_log2 = %u out of range [8, 28]", c0.stream_chunk_log2);
    int dev = c0.device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return fail(nullptr, MSM_ERR_NO_DEVICE, "hipGetDevice failed");
    if (dev >= ndev) return fail(nullptr, MSM_ERR_NO_DEVICE, "device %d out of range (%d visible)", dev, ndev);
    msm_ctx* c = new (std::nothrow) msm_ctx();
    if (!c) return fail(nullptr, MSM_ERR_OOM, "host allocation failed");
    c->device = dev;
    c->cfg = c0;
    DeviceGuard g(dev);
    hipError_t e = g.ok ? hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) : hipErrorInvalidDevice;
    for (int i = 0; i < EV_COUNT && e == hipSuccess; i++) e = hipEventCreate(&c->ev[i]);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_bases, hipEventDisableTiming);
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
        e = hipEventCreateWithFlags(&c->ev_copied[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_free[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_qsums, MAX_QSUM_POINTS * 96, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_flags, 64, hipHostMallocDefault);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)msmk::k_tile_hist, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_HIST_BYTES);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)msmk::k_tile_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_HIST_BYTES);

    if (e != hipSuccess) {
        fail(nullptr, MSM_ERR_HIP, "context setup failed: %s", hipGetErrorString(e));
        msm_ctx_destroy(c);
        return MSM_ERR_HIP;
    }
    if (c0.max_points) {
        msm_plan_t pl;
        make_plan(c0.max_points, c0.window_bits, c0.flags, &pl);
        size_t pairs = (size_t)pl.num_windows * (size_t)pl.virtual_points, tb = (size_t)pl.num_windows * pl.num_buckets;
        int32_t rc = MSM_OK;
        if (!rc) rc = ensure(c, c->bases, c0.max_points * 64);
        if (!rc) rc = ensure(c, c->scalars, c0.max_points * 32);
        if (!rc) rc = ensure(c, c->digits, pairs * 4);
        if (!rc) rc = ensure(c, c->sorted, pairs * 4);
        if (!rc) rc = ensure(c, c->ibases, (pl.glv ? 2 : 1) * c0.max_points * 64);
        if (!rc) rc = ensure(c, c->buckets, tb * XB);
        if (rc) {
            g_create_error = c->err;
            msm_ctx_destroy(c);
            return rc;
        }
    }
    {
        // host finish threads: MSM_HIP_HOST_THREADS=0 forces the serial path
        int want = (int)std::thread::hardware_concurrency() - 1;
        if (want > 1) want = 1;  // the calling thread + ONE worker: measured over 1500 calls at 2^17 (tools/step_jitter.py),
                                 // mean latency 0.608 / 0.575 / 0.588 / 0.590 ms with 1 / 2 / 3 / 4 threads -- the median keeps
                                 // falling (0.606 / 0.570 / 0.556 / 0.555) but 3+ threads bring 2-8 ms outliers in ~1.3 % of the calls
        if (const char* e = std::getenv("MSM_HIP_HOST_THREADS")) want = std::atoi(e) - 1;
        if (const char* e = std::getenv("MSM_HIP_WIDE_MAX")) c->wide_max = (size_t)std::max(0, std::atoi(e));
        if (want >= 1) c->pool = new (std::nothrow) HostPool(want);
    }
    *out = c;
    return MSM_OK;
}

void msm_ctx_destroy(msm_ctx* c) {
    if (!c) return;
    delete c->pool;
    c->pool = nullptr;
    {
        DeviceGuard g(c->device);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        DevBuf* bufs[] = {&c->bases,   &c->inf,       &c->scalars, &c->digits,  &c->ranks,  &c->sorted, &c->hist,
                          &c->offsets, &c->blocksums, &c->buckets, &c->rc,      &c->flags,  &c->pow2,
                          &c->heads,   &c->tails,     &c->chunkmap, &c->tilecounts, &c->ibases, &c->longlist, &c->longdone, &c->midlist, &c->ccounts, &c->cregion, &c->bigslot, &c->big};
        for (DevBuf* b : bufs) release(*b);
        if (c->h_qsums) (void)hipHostFree(c->h_qsums);
        if (c->h_flags) (void)hipHostFree(c->h_flags);
        if (c->h_sq) (void)hipHostFree(c->h_sq);
        for (int i = 0; i < 2; i++) {
            release(c->sbases[i]);
            release(c->sscalars[i]);
            release(c->sinf[i]);
            if (c->ev_copied[i]) (void)hipEventDestroy(c->ev_copied[i]);
            if (c->ev_free[i]) (void)hipEventDestroy(c->ev_free[i]);
        }
        if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
        if (c->ev_bases) (void)hipEventDestroy(c->ev_bases);
        if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
        for (int i = 0; i < EV_COUNT; i++)
            if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
        if (c->stream) (void)hipStreamDestroy(c->stream);
    }
    delete c;
}

int32_t msm_bn254_g1(msm_ctx* c, const uint32_t* bases_xy, uint32_t base_form, const uint8_t* inf_mask,
                     const uint32_t* scalars, size_t n, uint32_t out_jac[24], uint32_t out_aff[16], uint8_t* out_inf) {
    int32_t rc = check_common(c, bases_xy, scalars, n);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    auto t0 = std::chrono::steady_clock::now();
    c->resident_n = 0;  // the scratch copies below are not a resident set
    {
        // Streaming needs copies that really run beside the kernels.  Measured (tools/pin_test.py, N = 2^22): from
        // PINNED caller memory 10.5 ms streamed vs 15.0 ms single-shot; from pageable memory the runtime's staged
        // copy does not overlap and chunking only adds its fixed costs (16.0 vs 15.1 ms); hipHostRegister costs as
        // much as the copy itself (10.9 ms per 256 MB).  So: explicit stream_chunk_log2 => always stream;
        // default => stream only when both caller buffers are pinned.
        uint32_t lg = c->cfg.stream_chunk_log2 ? c->cfg.stream_chunk_log2 : 21u;
        size_t chunk = (size_t)1 << lg;
        bool want = n >= 2 * chunk;
        if (want && !c->cfg.stream_chunk_log2) {
            hipPointerAttribute_t a0{}, a1{};
            want = hipPointerGetAttributes(&a0, bases_xy) == hipSuccess && a0.type == hipMemoryTypeHost &&
                   hipPointerGetAttributes(&a1, scalars) == hipSuccess && a1.type == hipMemoryTypeHost;
            (void)hipGetLastError();  // an unregistered pointer is not an error here
        }
        if (want) {
            rc = run_streamed(c, bases_xy, base_form, inf_mask, scalars, n, chunk, out_jac, out_aff, out_inf);
            if (rc) return rc;
            c->tm.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
            return MSM_OK;
        }
    }
    if (c->stage_timing) HIPCHK(c, hipEventRecord(c->ev[EV_START], c->stream));
    if ((rc = ensure(c, c->scalars, n * 32))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->scalars.p, scalars, n * 32, hipMemcpyHostToDevice, c->stream));
    if ((rc = upload_bases_locked(c, bases_xy, base_form, inf_mask, n))) return rc;
    rc = run_pipeline(c, (const uint32_t*)c->ibases.p, inf_mask ? (const uint8_t*)c->inf.p : nullptr,
                      (const uint32_t*)c->scalars.p, n, c->stream, out_jac, out_aff, out_inf);
    if (rc) return rc;
    c->tm.h2d_ms = stage_ms(c, EV_START, EV_H2D);
    c->tm.convert_ms = stage_ms(c, EV_H2D, EV_CONVERT);
    c->tm.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return MSM_OK;
}

int32_t msm_bn254_g1_arkworks(msm_ctx* c, const void* bases, size_t stride, size_t x_off, size_t y_off, size_t inf_off,
                              const uint32_t* scalars_mont, size_t n, uint32_t out_jac[24], uint32_t out_aff[16], uint8_t* out_inf) {
    int32_t rc = check_common(c, bases, scalars_mont, n);
    if (rc) return rc;
    const bool has_inf = inf_off != (size_t)-1;
    if (stride < 64 || (stride & 3) || (x_off & 3) || (y_off & 3) || x_off + 32 > stride || y_off + 32 > stride ||
        (has_inf && inf_off >= stride) || ((uintptr_t)bases & 3))
        return fail(c, MSM_ERR_BAD_ARG, "bad G1Affine layout: stride %zu x %zu y %zu inf %zu", stride, x_off, y_off, inf_off);
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    auto t0 = std::chrono::steady_clock::now();
    c->resident_n = 0;
    if (c->stage_timing) HIPCHK(c, hipEventRecord(c->ev[EV_START], c->stream));
    if ((rc = ensure(c, c->scalars, n * 32))) return rc;
    const bool glv = plan_glv(c, n);
    if ((rc = ensure(c, c->bases, n * stride))) return rc;
    if ((rc = ensure(c, c->ibases, (glv ? 2 : 1) * n * 64))) return rc;
    if ((rc = ensure(c, c->inf, n))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->scalars.p, scalars_mont, n * 32, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->bases.p, bases, n * stride, hipMemcpyHostToDevice, c->stream));
    if (c->stage_timing) HIPCHK(c, hipEventRecord(c->ev[EV_H2D], c->stream));
    msmk::k_import_ark<<<grid1(2 * n, 256), 256, 0, c->stream>>>((const uint8_t*)c->bases.p, (uint64_t)stride, (uint32_t)x_off, (uint32_t)y_off,
                                                               has_inf ? (uint32_t)inf_off : 0u, has_inf ? 1u : 0u, (uint32_t)n,
                                                               (uint32_t*)c->ibases.p, (uint8_t*)c->inf.p, glv ? 1u : 0u);
    rc = run_pipeline(c, (const uint32_t*)c->ibases.p, (const uint8_t*)c->inf.p, (const uint32_t*)c->scalars.p, n, c->stream, out_jac,
                      out_aff, out_inf, 1u);
    if (rc) return rc;
    c->tm.h2d_ms = stage_ms(c, EV_START, EV_H2D);
    c->tm.convert_ms = stage_ms(c, EV_H2D, EV_CONVERT);
    c->tm.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return MSM_OK;
}

int32_t msm_bn254_g1_upload_bases(msm_ctx* c, const uint32_t* bases_xy, uint32_t base_form, const uint8_t* inf_mask,
                                  size_t n) {
    int32_t rc = check_common(c, bases_xy, bases_xy, n);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    c->resident_n = 0;
    if ((rc = upload_bases_locked(c, bases_xy, base_form, inf_mask, n))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    c->resident_n = n;
    c->resident_glv = plan_glv(c, n);
    c->resident_has_inf = inf_mask != nullptr;
    return MSM_OK;
}

// row f3: n x 32-byte arkworks compressed images (host) -> c->ibases (+ c->inf); out_ark picks the word domain written
static int32_t decompress_locked(msm_ctx* c, const uint8_t* compressed, size_t n, uint32_t out_ark, int64_t* first_invalid) {
    int32_t rc;
    if (first_invalid) *first_invalid = -1;
    if (n > 0xFFFFFFF0ull) return fail(c, MSM_ERR_BAD_ARG, "too many points: %zu", n);
    const bool glv = !out_ark && plan_glv(c, n);
    if ((rc = ensure(c, c->bases, n * 32 + 16))) return rc;
    if ((rc = ensure(c, c->ibases, (glv ? 2 : 1) * n * 64))) return rc;
    if ((rc = ensure(c, c->inf, n))) return rc;
    uint32_t* d_bad = (uint32_t*)((uint8_t*)c->bases.p + n * 32);  // lowest failing index, kept behind the images
    const uint32_t none = 0xFFFFFFFFu;
    HIPCHK(c, hipMemcpyAsync(c->bases.p, compressed, n * 32, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_bad, &none, 4, hipMemcpyHostToDevice, c->stream));
    msmk::k_decompress<<<grid1(n, 256), 256, 0, c->stream>>>((const uint32_t*)c->bases.p, (uint32_t)n, (uint32_t*)c->ibases.p,
                                                          (uint8_t*)c->inf.p, d_bad, out_ark, glv ? 1u : 0u);
    uint32_t bad = none;
    HIPCHK(c, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    if (bad != none) {
        if (first_invalid) *first_invalid = (int64_t)bad;
        return fail(c, MSM_ERR_INVALID_DATA, "compressed point %u does not decode (flags, x >= p, or x^3+3 not a square)", bad);
    }
    return MSM_OK;
}

int32_t msm_bn254_g1_decompress(msm_ctx* c, const uint8_t* compressed, size_t n, uint32_t* out_xy_mont, uint8_t* out_inf,
                                int64_t* first_invalid) {
    int32_t rc = check_common(c, compressed, out_xy_mont, n);
    if (rc) return rc;
    if (!out_inf) return fail(c, MSM_ERR_BAD_ARG, "NULL out_inf");
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    c->resident_n = 0;  // ibases is used as the output staging area
    if ((rc = decompress_locked(c, compressed, n, 1u, first_invalid))) return rc;
    HIPCHK(c, hipMemcpyAsync(out_xy_mont, c->ibases.p, n * 64, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(out_inf, c->inf.p, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MSM_OK;
}

int32_t msm_bn254_g1_upload_compressed(msm_ctx* c, const uint8_t* compressed, size_t n, int64_t* first_invalid) {
    int32_t rc = check_common(c, compressed, compressed, n);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    c->resident_n = 0;
    if ((rc = decompress_locked(c, compressed, n, 0u, first_invalid))) return rc;
    c->resident_n = n;
    c->resident_glv = plan_glv(c, n);
    c->resident_has_inf = true;
    return MSM_OK;
}

// host-side inverse (ark-ec 0.4 serialize_compressed): x standard form LE | bit 255 = y > p - y | bit 254 = infinity
int32_t msm_bn254_g1_compress(const uint32_t* bases_xy, uint32_t base_form, const uint8_t* inf_mask, size_t n, uint8_t* out) {
    if (n == 0) return MSM_ERR_EMPTY;
    if (!bases_xy || !out || (base_form != MSM_FORM_STD && base_form != MSM_FORM_MONT)) return MSM_ERR_BAD_ARG;
    static constexpr uint64_t HALF[4] = {0x9e10460b6c3e7ea3ULL, 0xcbc0b548b438e546ULL, 0xdc2822db40c0ac2eULL, 0x183227397098d014ULL};  // (p-1)/2
    auto work = [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (inf_mask && inf_mask[i]) {
                w[7] = 1u << 30;
            } else {
                hostg1::Fq x = hostg1::load_words(bases_xy + i * 16), y = hostg1::load_words(bases_xy + i * 16 + 8);
                if (base_form == MSM_FORM_MONT) x = hostg1::from_mont(x), y = hostg1::from_mont(y);
                hostg1::store_words(w, x);
                bool larger = false;
                for (int k = 3; k >= 0; k--)
                    if (y.l[k] != HALF[k]) {
                        larger = y.l[k] > HALF[k];
                        break;
                    }
                if (larger) w[7] |= 1u << 31;
            }
            std::memcpy(out + i * 32, w, 32);
        }
    };
    const size_t nt = n < 8192 ? 1 : std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency()));
    if (nt == 1) {
        work(0, n);
    } else {
        std::vector<std::thread> th;
        for (size_t t = 0; t < nt; t++) th.emplace_back(work, n * t / nt, n * (t + 1) / nt);
        for (auto& t : th) t.join();
    }
    return MSM_OK;
}

int32_t msm_bn254_g1_resident(msm_ctx* c, const uint32_t* scalars, size_t n, uint32_t out_jac[24], uint32_t out_aff[16],
                              uint8_t* out_inf) {
    int32_t rc = check_common(c, scalars, scalars, n);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->resident_n == 0) return fail(c, MSM_ERR_STATE, "no resident bases: call msm_bn254_g1_upload_bases first");
    if (n > c->resident_n) n = c->resident_n;  // unequal lengths truncate to the shorter (metal_msm.rs:652-656)
    DeviceGuard g(c->device);
    auto t0 = std::chrono::steady_clock::now();
    if (c->stage_timing) HIPCHK(c, hipEventRecord(c->ev[EV_START], c->stream));
    if ((rc = ensure(c, c->scalars, n * 32))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->scalars.p, scalars, n * 32, hipMemcpyHostToDevice, c->stream));
    if (c->stage_timing) HIPCHK(c, hipEventRecord(c->ev[EV_H2D], c->stream));
    // the phi records sit at index resident_n + i: a call on fewer scalars (truncation) or a set uploaded without them runs unsplit
    const uint32_t extra = (c->resident_glv && n == c->resident_n) ? 0u : MSM_FLAG_NO_GLV;
    rc = run_pipeline(c, (const uint32_t*)c->ibases.p, c->resident_has_inf ? (const uint8_t*)c->inf.p : nullptr,
                      (const uint32_t*)c->scalars.p, n, c->stream, out_jac, out_aff, out_inf, 0, nullptr, extra);
    if (rc) return rc;
    c->tm.h2d_ms = stage_ms(c, EV_START, EV_H2D);
    c->tm.convert_ms = 0;
    c->tm.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return MSM_OK;
}

int32_t msm_bn254_g1_device(msm_ctx* c, const void* d_bases_mont, const void* d_inf_mask, const void* d_scalars, size_t n,
                            void* hip_stream, uint32_t out_jac[24], uint32_t out_aff[16], uint8_t* out_inf) {
    int32_t rc = check_common(c, d_bases_mont, d_scalars, n);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    auto t0 = std::chrono::steady_clock::now();
    const uint32_t glv = plan_glv(c, n) ? 1u : 0u;
    if ((rc = ensure(c, c->ibases, (glv ? 2 : 1) * n * 64))) return rc;
    // Each event record / cross-stream wait costs ~6 us of stream time (measured gaps in the kernel trace), so the
    // conversion only moves to the second stream when it is longer than that (n > 2^18: 32 us at 2^20, 5 us at 2^16).
    if (c->stage_timing || n <= ((size_t)1 << 18)) {  // serialised (also: so that convert_ms means something)
        if (c->stage_timing) HIPCHK(c, hipEventRecord(c->ev[EV_H2D], st));
        msmk::k_convert_bases<<<grid1(2 * n, 256), 256, 0, st>>>((const uint32_t*)d_bases_mont, (uint32_t*)c->ibases.p, (uint32_t)n, 1u, glv);
        rc = run_pipeline(c, (const uint32_t*)c->ibases.p, (const uint8_t*)d_inf_mask, (const uint32_t*)d_scalars, n, st, out_jac,
                          out_aff, out_inf);
    } else {  // the bases are not needed before k_accumulate: convert them on the second stream beside the sort
        if (hip_stream) {  // the caller's stream may still be producing the inputs; the context's own stream is idle between calls
            HIPCHK(c, hipEventRecord(c->ev_fork, st));
            HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev_fork, 0));
        }
        msmk::k_convert_bases<<<grid1(2 * n, 256), 256, 0, c->copy_stream>>>((const uint32_t*)d_bases_mont, (uint32_t*)c->ibases.p,
                                                                           (uint32_t)n, 1u, glv);
        HIPCHK(c, hipEventRecord(c->ev_bases, c->copy_stream));
        rc = run_pipeline(c, (const uint32_t*)c->ibases.p, (const uint8_t*)d_inf_mask, (const uint32_t*)d_scalars, n, st, out_jac,
                          out_aff, out_inf, 0, c->ev_bases);
    }
    if (rc) return rc;
    c->tm.h2d_ms = 0;
    c->tm.convert_ms = stage_ms(c, EV_H2D, EV_CONVERT);
    c->tm.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return MSM_OK;
}

int32_t msm_bn254_g1_combine(const uint32_t* partials, size_t k, uint32_t out_jac[24], uint32_t out_aff[16],
                             uint8_t* out_inf) {
    if (!partials) return MSM_ERR_BAD_ARG;
    if (k == 0) return MSM_ERR_EMPTY;
    hostg1::Jac total = hostg1::identity();
    for (size_t i = 0; i < k; i++) total = hostg1::jadd(total, hostg1::load_jac(partials + i * 24));  // fixed rank order
    finish_outputs(total, out_jac, out_aff, out_inf);
    return MSM_OK;
}

int32_t msm_plan(size_t n, uint32_t window_bits, uint32_t flags, msm_plan_t* out) {
    if (!out) return MSM_ERR_BAD_ARG;
    if (n == 0) return MSM_ERR_EMPTY;
    return make_plan(n, window_bits, flags, out);
}

int32_t msm_get_timings(const msm_ctx* c, msm_timings_t* out) {
    if (!c || !out) return MSM_ERR_BAD_ARG;
    *out = c->tm;
    return MSM_OK;
}
int32_t msm_get_accumulate_kernel_stats(const msm_ctx* c, double* avg_ms, uint64_t* launches) {
    if (!c) return MSM_ERR_BAD_ARG;
    if (avg_ms) *avg_ms = c->acc_launches ? c->acc_ms_sum / (double)c->acc_launches : 0.0;
    if (launches) *launches = c->acc_launches;
    return MSM_OK;
}
int32_t msm_set_stage_timing(msm_ctx* c, int32_t enabled) {
    if (!c) return MSM_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    c->stage_timing = enabled != 0;
    return MSM_OK;
}
void msm_reset_kernel_stats(msm_ctx* c) {
    if (!c) return;
    c->acc_ms_sum = 0;
    c->acc_launches = 0;
}

int32_t msm_bn254_g1_generate_device(msm_ctx* c, uint64_t base_seed, uint64_t scalar_seed, size_t n, void* d_bases_out,
                                     void* d_scalars_out) {
    if (!c) return MSM_ERR_BAD_ARG;
    if (n == 0) return fail(c, MSM_ERR_EMPTY, "Empty input");
    if (n > 0x7FFFFFFFull) return fail(c, MSM_ERR_BAD_ARG, "n too large");
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    int32_t rc;
    if (d_bases_out) {
        if ((rc = ensure_pow2_table(c))) return rc;
        msmk::k_gen_bases<<<grid1(n, 128), 128, 0, c->stream>>>(base_seed, (uint32_t)n, (const uint32_t*)c->pow2.p,
                                                             (uint32_t*)d_bases_out);
    }
    if (d_scalars_out) msmk::k_gen_scalars<<<grid1(n, 256), 256, 0, c->stream>>>(scalar_seed, (uint32_t)n, (uint32_t*)d_scalars_out);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    return MSM_OK;
}

int32_t msm_bn254_generate_scalars_host(uint64_t seed, size_t n, int nonzero, uint32_t* out) {
    if (!out) return MSM_ERR_BAD_ARG;
    for (size_t i = 0; i < n; i++) msmk::gen_scalar(seed, i, nonzero != 0, out + i * 8);
    return MSM_OK;
}

// ---- device-math unit-test hooks --------------------------------------------------------------------
static int32_t run_test_kernel(msm_ctx* c, bool g1, uint32_t op, const uint32_t* a, size_t a_words, const uint32_t* b,
                               size_t b_words, uint32_t* out, size_t out_words, size_t n) {
    if (!c || !a || !out) return MSM_ERR_BAD_ARG;
    if (n == 0) return MSM_ERR_EMPTY;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    void *da = nullptr, *db = nullptr, *dout = nullptr;
    HIPCHK(c, hipMalloc(&da, n * a_words * 4));
    HIPCHK(c, hipMalloc(&dout, n * out_words * 4));
    HIPCHK(c, hipMemcpy(da, a, n * a_words * 4, hipMemcpyHostToDevice));
    if (b) {
        HIPCHK(c, hipMalloc(&db, n * b_words * 4));
        HIPCHK(c, hipMemcpy(db, b, n * b_words * 4, hipMemcpyHostToDevice));
    }
    if (g1 && op == MSM_OP_G1_ADD_WIDE) msmk::k_test_g1_wide<<<grid1(n, 8), 64, 0, c->stream>>>((uint32_t*)da, (uint32_t*)db, (uint32_t*)dout, (uint32_t)n);
    else if (g1) msmk::k_test_g1<<<grid1(n, 64), 64, 0, c->stream>>>(op, (uint32_t*)da, (uint32_t*)db, (uint32_t*)dout, (uint32_t)n);
    else msmk::k_test_fp<<<grid1(n, 64), 64, 0, c->stream>>>(op, (uint32_t*)da, (uint32_t*)db, (uint32_t*)dout, (uint32_t)n);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpy(out, dout, n * out_words * 4, hipMemcpyDeviceToHost));
    (void)hipFree(da);
    (void)hipFree(dout);
    if (db) (void)hipFree(db);
    return MSM_OK;
}
int32_t msm_test_fp_op(msm_ctx* c, uint32_t op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n) {
    if (op > MSM_OP_FP_INV) return MSM_ERR_BAD_ARG;
    if (op <= MSM_OP_FP_MONT_MUL && !b) return MSM_ERR_BAD_ARG;
    return run_test_kernel(c, false, op, a, 8, op <= MSM_OP_FP_MONT_MUL ? b : nullptr, 8, out, 8, n);
}
int32_t msm_test_g1_op(msm_ctx* c, uint32_t op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n) {
    if (op > MSM_OP_G1_ADD_WIDE) return MSM_ERR_BAD_ARG;
    if (op != MSM_OP_G1_DBL && !b) return MSM_ERR_BAD_ARG;
    return run_test_kernel(c, true, op, a, 24, op == MSM_OP_G1_DBL ? nullptr : b, op == MSM_OP_G1_MADD ? 16 : 24, out, 24, n);
}
int32_t msm_calibrate(msm_ctx* c, double* mad_per_s, double* fp_mul_per_s) {
    if (!c) return MSM_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    int32_t rc;
    if ((rc = ensure(c, c->flags, 64))) return rc;
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, c->device));
    const unsigned blocks = (unsigned)prop.multiProcessorCount * 4u * 4u;  // 256-thread blocks: 4 wavefronts on each of a CU's 4 SIMDs
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0));
    HIPCHK(c, hipEventCreate(&e1));
    double out[2] = {0, 0};
    for (uint32_t what = 0; what < 2; what++) {
        const uint32_t iters = what == 0 ? 500u : 100u;  // ~1 ms each
        const double ops_per_thread = what == 0 ? 64.0 * iters : 4.0 * iters;
        msmk::k_calibrate<<<blocks, 256, 0, c->stream>>>(what, iters / 10, (uint32_t*)c->flags.p + 15);  // warm-up
        HIPCHK(c, hipEventRecord(e0, c->stream));
        msmk::k_calibrate<<<blocks, 256, 0, c->stream>>>(what, iters, (uint32_t*)c->flags.p + 15);
        HIPCHK(c, hipEventRecord(e1, c->stream));
        HIPCHK(c, hipEventSynchronize(e1));
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
        out[what] = ops_per_thread * (double)blocks * 256.0 / ((double)ms * 1e-3);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    HIPCHK(c, hipGetLastError());
    if (mad_per_s) *mad_per_s = out[0];
    if (fp_mul_per_s) *fp_mul_per_s = out[1];
    return MSM_OK;
}

int32_t msm_test_decompose(msm_ctx* c, const uint32_t* scalars, size_t n, uint32_t window_bits, int32_t* digits) {
    if (!c || !scalars || !digits) return MSM_ERR_BAD_ARG;
    if (n == 0) return MSM_ERR_EMPTY;
    msm_plan_t pl;
    if (make_plan(n, window_bits ? window_bits : c->cfg.window_bits, c->cfg.flags | MSM_FLAG_NO_GLV, &pl)) return MSM_ERR_BAD_ARG;  // plain 254-bit digits
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    void *ds = nullptr, *dd = nullptr;
    size_t out_bytes = (size_t)pl.num_windows * n * 4;
    HIPCHK(c, hipMalloc(&ds, n * 32));
    HIPCHK(c, hipMalloc(&dd, out_bytes));
    HIPCHK(c, hipMemcpy(ds, scalars, n * 32, hipMemcpyHostToDevice));
    if (pl.signed_digits)
        msmk::k_decompose_plain<true><<<grid1(n, 256), 256, 0, c->stream>>>((uint32_t*)ds, (uint32_t)n, pl.window_bits,
                                                                         pl.num_windows, (int32_t*)dd);
    else
        msmk::k_decompose_plain<false><<<grid1(n, 256), 256, 0, c->stream>>>((uint32_t*)ds, (uint32_t)n, pl.window_bits,
                                                                          pl.num_windows, (int32_t*)dd);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpy(digits, dd, out_bytes, hipMemcpyDeviceToHost));
    (void)hipFree(ds);
    (void)hipFree(dd);
    return MSM_OK;
}

}  // extern "C"
