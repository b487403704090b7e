// msm_hip.hip -- host runtime + C ABI (include/msm_hip.h) of the MI355X-native BN254 G1 MSM.
//
// Replaces, for the one path metal_variable_base_msm (metal_msm.rs:642-695):
//   MetalMSMPipeline::{new,execute_pipeline,final_reduction}      metal_msm.rs:48-261
//   ShaderManager / MetalHelper / gpu::{create_buffer,read_buffer} host/shader_manager.rs:98-167,
//                                                                 host/metal_wrapper.rs:55-217, host/gpu.rs:3-31
// Design differences (MI355X-first):
//   * a persistent context owns the device, two HIP streams, the HBM workspace and the hipEvents; the
//     reference re-opens the device, reloads the metallib twice and builds six pipeline states on
//     EVERY call (metal_msm.rs:693 -> 64 -> 48, window_size_optimizer.rs:79-92);
//   * all intermediates stay in HBM -- the reference round-trips every stage through a host Vec<u32>
//     (metal_msm.rs:331-339, 403-407, 505-507, 630-632);
//   * launches are queued back to back; the only host synchronisation is the final wait for W*(kb+1) bit sums that
//     the last kernel writes into pinned host memory.  The reference blocks after each of its 9 submits;
//   * host inputs are pipelined: the bases travel on the copy stream while the scalars are already being sorted, and
//     from 2^19 points on the point range is cut into chunks that travel while the previous chunk is accumulated INTO the
//     shared bucket array.
// There is NO CPU fallback: without a HIP device every compute entry point returns MSM_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/msm_hip.h"
#include "host_g1.hpp"
#include "msm_host_pool.hpp"
#include "msm_planner.hpp"
#include "msm_kernels.hpp"

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

enum { EV_START, EV_H2D, EV_CONVERT, EV_DECOMP, EV_SORT, EV_PLAN, EV_ACC0, EV_ACC1, EV_COMBINE, EV_REDUCE, EV_COUNT };

// ---- observability (reference counterpart: LOG_DEBUG in build.rs:134-138 and the Metal capture scopes of
// host/gpu.rs:34-114).  MSM_HIP_ROCTX=1: every stage is bracketed by a roctx range (librocprofiler-sdk-roctx / libroctx64 is
// dlopen'ed, nothing is linked), so `rocprofv3 --marker-trace` shows the pipeline structure.  MSM_HIP_TRACE=1: one line per
// call on stderr (plan, path taken, per-stage device times -- turns the per-stage events on).
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char* e = std::getenv("MSM_HIP_ROCTX");
        if (!e || !*e || *e == '0') return;
        for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
                pop = (int (*)())dlsym(h, "roctxRangePop");
                if (push && pop) return;
                push = nullptr, pop = nullptr;
            }
        }
    }
};
const Roctx& roctx() {
    static Roctx r;
    return r;
}
struct Range {  // RAII roctx range; free when MSM_HIP_ROCTX is unset
    bool on;
    explicit Range(const char* name) : on(roctx().push != nullptr) {
        if (on) roctx().push(name);
    }
    ~Range() {
        if (on) roctx().pop();
    }
};
bool trace_enabled() {
    static const bool on = [] {
        const char* e = std::getenv("MSM_HIP_TRACE");
        return e && *e && *e != '0';
    }();
    return on;
}

}  // namespace

// Test / A-B knobs of the HOOKS build (libmsm_hip_hooks.so, -DMSM_HIP_TEST_HOOKS), read ONCE when a context is created: a call never looks
// at the environment, so an upload and the resident calls after it, or the two pipelines of a batch, cannot see different plans.
// The PRODUCT library reads none of them (round 5; the reference's whole configuration surface is one struct, metal_msm.rs:16-28): what a
// caller may choose is in msm_config_t (window width, flags, stream chunk, workspace pre-sizing, batch layout, host threads); the product's
// only environment variables are MSM_HIP_TRACE / MSM_HIP_ROCTX (observability, per process) and MSM_HIP_DEVICES / MSM_HIP_MULTI_TIMEOUT_S
// (msm_multi_create: deployment).
struct Knobs {
    size_t glv_max = msmplan::GLV_MAX_POINTS;  // MSM_HIP_GLV_MAX_LOG2
    uint32_t piece_len = 0;                    // MSM_HIP_PIECE_LEN: longest whole bucket = split length of k_accumulate_pieces' work items; 0 = by size (tests force 1, 7, 26, 35)
    bool direct_scatter = false;               // MSM_HIP_DIRECT_SCATTER: skip the two-level LDS sort
    bool ark_slow = false;                     // MSM_HIP_ARK_SLOW: struct arrays always through k_import_ark (A/B and tests of the path a set infinity flag falls back to)
    uint32_t mid_lane_min = msmk::MID_LANE_MIN;  // MSM_HIP_MID_LANE_MIN: k_combine_pieces folds mid lists longer than this one lane per bucket
    size_t split_target = msmplan::SPLIT_PIECES_TARGET;  // MSM_HIP_SPLIT_TARGET: pieces the runs of very long buckets are sized for (make_piece_plan)
    uint32_t split_shift = msmplan::SPLIT_ENTRIES_SHIFT;  // MSM_HIP_SPLIT_SHIFT: the kernels shorten the runs of very long buckets to entries >> this (msmk::effective_psplit); 0 = never
    uint32_t pair8_max_mb = 200;               // MSM_HIP_PAIR8_MAX_MB: bucket arrays up to this size take k_pair_level8 (three levels in one launch)
    bool no_poll = false;                      // MSM_HIP_NO_POLL: wait for the stream instead of polling the last kernel's result pairs (A/B: tools/ab_env.py)
    uint32_t device_chunk_log2 = 22;           // MSM_HIP_DEVICE_CHUNK_LOG2: point ranges of device-resident instances; 0 = never cut
    uint32_t stream_min_log2 = 19;             // MSM_HIP_STREAM_MIN_LOG2: host calls are streamed from this size on (tools/host_path_sweep.py)
    uint32_t stream_chunk_log2 = 0;            // MSM_HIP_STREAM_CHUNK_LOG2: 0 = by size (product: msm_config_t.stream_chunk_log2)
    std::vector<uint32_t> stream_schedule;     // MSM_HIP_STREAM_SCHEDULE="17,18,18,18,17": the chunks of a streamed host call as log2 sizes, used when they sum to n (tools/host_schedule_sweep.py)
    int host_threads = -1;                     // MSM_HIP_HOST_THREADS: CPU finish threads incl. the caller; -1 = msm_config_t.host_threads
    msmplan::table_knobs table;                // MSM_HIP_TABLE_C / MSM_HIP_TABLE_F / MSM_HIP_TABLE_MAX_GB / MSM_HIP_TABLE_GLV_MAX_LOG2 (window table of a resident set)
    static Knobs from_env() {
        Knobs k;
#ifdef MSM_HIP_TEST_HOOKS
        auto num = [](const char* name, long lo, long hi, long dflt) {
            const char* e = std::getenv(name);
            if (!e || !*e) return dflt;
            return std::max(lo, std::min(hi, std::atol(e)));
        };
        k.glv_max = msmplan::glv_max_from_env();
        k.piece_len = (uint32_t)num("MSM_HIP_PIECE_LEN", 0, msmk::PIECE_BINS, 0);
        k.direct_scatter = std::getenv("MSM_HIP_DIRECT_SCATTER") != nullptr;  // (tests: the one-level LDS / global-atomic sort fallbacks)
        k.ark_slow = std::getenv("MSM_HIP_ARK_SLOW") != nullptr;
        k.no_poll = std::getenv("MSM_HIP_NO_POLL") != nullptr;
        k.pair8_max_mb = (uint32_t)num("MSM_HIP_PAIR8_MAX_MB", 0, 4096, 200);
        k.split_target = (size_t)num("MSM_HIP_SPLIT_TARGET", 1024, 1 << 30, (long)msmplan::SPLIT_PIECES_TARGET);
        k.split_shift = (uint32_t)num("MSM_HIP_SPLIT_SHIFT", 0, 24, msmplan::SPLIT_ENTRIES_SHIFT);
        k.mid_lane_min = (uint32_t)num("MSM_HIP_MID_LANE_MIN", 0, 0x7FFFFFFF, msmk::MID_LANE_MIN);
        k.device_chunk_log2 = (uint32_t)num("MSM_HIP_DEVICE_CHUNK_LOG2", 0, 30, 22);
        k.stream_min_log2 = (uint32_t)num("MSM_HIP_STREAM_MIN_LOG2", 9, 31, 19);
        k.stream_chunk_log2 = std::getenv("MSM_HIP_STREAM_CHUNK_LOG2") ? (uint32_t)num("MSM_HIP_STREAM_CHUNK_LOG2", 8, 28, 0) : 0u;
        if (const char* e = std::getenv("MSM_HIP_STREAM_SCHEDULE")) {
            for (const char* p = e; *p;) {
                char* end = nullptr;
                const long v = std::strtol(p, &end, 10);
                if (end == p) break;
                if (v >= 8 && v <= 28) k.stream_schedule.push_back((uint32_t)v);
                p = *end ? end + 1 : end;
            }
        }
        if (std::getenv("MSM_HIP_HOST_THREADS")) k.host_threads = (int)num("MSM_HIP_HOST_THREADS", 0, 64, 2);
        k.table.c = (uint32_t)num("MSM_HIP_TABLE_C", 0, 20, 0);
        k.table.f = (uint32_t)num("MSM_HIP_TABLE_F", 0, 128, 0);
        k.table.max_bytes = (size_t)num("MSM_HIP_TABLE_MAX_GB", 0, 1024, 64) << 30;
        if (std::getenv("MSM_HIP_TABLE_GLV_MAX_LOG2")) k.table.glv_max = (size_t)1 << num("MSM_HIP_TABLE_GLV_MAX_LOG2", 0, 23, 18);  // (tools/table_sweep.py)
#endif
        return k;
    }
};

// Chunks of a streamed host call in flight: the transfer of chunk j + STREAM_SLOTS reuses the raw buffers of chunk j and waits for its
// accumulation.  With two slots the link stalls when a chunk's transfer is shorter than the accumulation of the chunk two before it (the two
// half-size chunks a ragged instance ends on); three cost one more chunk of HBM.
constexpr int STREAM_SLOTS = 3;
struct msm_ctx {
    std::mutex mu;
    Knobs knobs;
    HostPool* pool = nullptr;        // CPU finish: the caller + one worker
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;     // host->HBM uploads; the coordinate pass of a host call's bases beside the sort (standard form: k_convert_bases; arkworks words: k_phi_records of a split plan)
    uint32_t num_cus = 0;                  // compute units of the device (rounds of k_accumulate workgroups)
    hipEvent_t ev_body = nullptr;          // batch: behind k_combine on the shared stream; the bucket reduction waits for it on another
    uint32_t tuned_layout[2] = {0, 0};     // msm_tune_batch: the measured MSM_BATCH_LAYOUT_* per size class (below / from 2^19 points); 0 = not tuned
    uint32_t last_batch_layout = 0;        // what the last batch call ran under (msm_timings_t.batch_layout)
    bool red_active = false;               // the reduce stream is in use by the batch call that is running
    hipEvent_t ev_copied[STREAM_SLOTS]{}, ev_free[STREAM_SLOTS]{}, ev_scal[STREAM_SLOTS]{};  // streamed host call: slot's bases+scalars there / slot free again / its scalars there
    hipEvent_t ev_fork = nullptr, ev_bases = nullptr;  // the bases' copy / coordinate pass runs on copy_stream beside the sort kernels
    DevBuf sbases[STREAM_SLOTS], sscalars[STREAM_SLOTS], sinf[STREAM_SLOTS];  // raw inputs of the streamed path, one set per chunk in flight
    DevBuf sibases;                           // the chunk being accumulated: its converted bases (standard form, structs) or the phi records of its arkworks words (they share the compute stream with the accumulation)
    msm_config_t cfg{};
    std::string err;
    hipEvent_t ev[EV_COUNT]{};
    // HBM workspace
    DevBuf bases, ibases, inf, scalars, digits, ranks, sorted, hist, offsets, blocksums, buckets, sorttmp, rc, flags,
        pow2, tilecounts, longlist, longdone, midlist, ccounts, cregion, bigslot, big,
        phist, pcursor, pbase, plist, partials;  // the accumulation's work items (k_piece_count / _scatter, k_accumulate_pieces, k_combine_pieces)
    std::mutex batch_mu;           // batch on one compute stream: a whole MSM is enqueued at a time
    bool batch_shared_stream = false;
    std::mutex copy_mu;            // batch: the two pipelines' scalar uploads take turns (see resident_on_lane)
    hipEvent_t last_copy = nullptr;  // ... the other pipeline's latest upload, recorded under copy_mu
    msm_ctx* lane1 = nullptr;      // second pipeline of msm_bn254_g1_resident_batch: a context of its own (streams, workspace, pinned results)
    HostPool* batch_pool = nullptr;  // ... and the host thread that drives it
    DevBuf clk;           // 4 x u64: shader cycles, constant-rate ticks, samples, additions of k_accumulate's first workgroup (clock probe)
    uint32_t wall_clock_khz = 0;  // rate of the constant counter (hipDeviceAttributeWallClockRate)
    DevBuf rbases, rinf;  // the RESIDENT base set (msm_bn254_g1_upload_bases / _upload_compressed): never used as scratch
    bool pow2_ready = false;
    uint32_t* h_qsums = nullptr;  // pinned: W x (kb+1) Jacobian bit sums as the last kernel writes them: 24 (word, sequence number) PAIRS each (msmk::store_words8_tagged)
    uint32_t done_seq = 0;        // sequence number of the latest bucket reduction queued on this context (the tag of its pairs)
    uint32_t* h_flags = nullptr;  // pinned: the 8 flag words, as pairs as well
    std::vector<uint32_t> qsums;  // the bit sums of the call being finished, 24 words each, taken out of the pairs once every tag is the call's (gather_results)
    uint32_t flagw[8] = {};       // ... and its flag words
    // resident bases
    size_t resident_n = 0;
    bool resident_has_inf = false;
    bool resident_glv = false;  // the resident set holds the phi records too (index resident_n + i)
    // window table of the resident set (MSM_FLAG_WINDOW_TABLE): rbases holds table_f levels of nv records, level j = 2^(table_c * j) * level 0
    uint32_t table_c = 0, table_f = 1;
    msm_timings_t tm{};
    bool stage_timing = false;  // record the per-stage hipEvents (each costs ~6 us of stream time); k_accumulate's pair is always on
    double acc_ms_sum = 0;
    uint64_t acc_launches = 0;
    uint32_t ktime_every = 0, ktime_count = 0;  // msm_set_kernel_timing: every n-th launch of k_accumulate_pieces carries its pair of events (0 = none)
    bool acc_last_timed = false;                // ... and whether the latest launch did
    std::chrono::steady_clock::time_point t_prepare{};  // trace: when the call started preparing / enqueueing
    float enqueue_ms = 0;         // trace: host time from there until everything was queued (what a hipGraph could shorten)
    bool flags_clean = false;     // the device flag words are known to be zero (set when an MSM completes)
    uint32_t last_sort_path = 0;  // 2 = two-level LDS sort, 1 = tiled LDS histogram, 0 = global atomics (stage tests, trace)
    bool no_host_pin = false;     // a context of an msm_multi handle: the handle pins the caller's arrays once for all its ranks (HostPin)
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
// scratch device allocation that is freed on every exit path
struct DevTmp {
    void* p = nullptr;
    ~DevTmp() {
        if (p) (void)hipFree(p);
    }
};

static_assert(msmplan::GLV_SPLIT_BITS == (uint32_t)glv::SPLIT_BITS, "planner and GLV split disagree");
static_assert(msmplan::PIECE_BINS_MAX == msmk::PIECE_BINS, "planner and piece sort disagree");
using msmplan::make_plan;
using msmplan::table_top_shift;
// the plan of a call on n points under this context's configuration and knobs (+ per-call extra flags)
inline int32_t ctx_plan(const msm_ctx* c, size_t n, uint32_t extra_flags, msm_plan_t* pl) {
    return make_plan(n, c->cfg.window_bits, c->cfg.flags | extra_flags, pl, c->knobs.glv_max);
}
// does such a call use the GLV split, i.e. 2n base records?
inline bool plan_glv(const msm_ctx* c, size_t n, uint32_t extra_flags = 0) {
    msm_plan_t pl;
    return ctx_plan(c, n, extra_flags, &pl) == MSM_OK && pl.glv != 0;
}
constexpr size_t XB = msmk::XW * 4;               // bytes per XYZZ record (4 coordinates x 9 x 29-bit limbs)
constexpr size_t LDS_HIST_BYTES = 128 * 1024;   // one window's bucket histogram must fit here for the LDS sort path
constexpr size_t WIDE_MAX_ADDS = 40960;     // pairwise levels of at most this many additions use eight lanes per addition (k_pair_level_wide)
constexpr size_t MAX_QSUM_POINTS = 4096;  // (pseudo-)windows x (rkb + 1) bit sums: c = 2: 128 x 2; c = 20 unsigned: 13 x 16 slices x 17 = 3536

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

// canonical = MSM_FLAG_DETERMINISTIC: the Jacobian result is handed out as its Z = 1 representative (x*R, y*R, R), the identity as (R, R, 0)
// -- the same 24 words for the same group element, whatever order the buckets were filled in (the sort places entries inside a bucket with
// LDS atomics: the XYZZ sums, hence X : Y : Z, differ between identical calls; only the group element does not).  Costs the inversion the
// affine output pays anyway (~10 us of host time), shared when both are asked for.
void finish_outputs(const hostg1::Jac& r, uint32_t* out_jac, uint32_t* out_aff, uint8_t* out_inf, bool canonical = false) {
    if (out_inf) *out_inf = hostg1::is_identity(r) ? 1 : 0;
    if (canonical && out_jac) {
        if (hostg1::is_identity(r)) {
            hostg1::store_jac(out_jac, hostg1::identity());
            if (out_aff) std::memset(out_aff, 0, 64);
            return;
        }
        const hostg1::Fq zi = hostg1::inv(r.z), zi2 = hostg1::sqr(zi);
        const hostg1::Fq xm = hostg1::mul(r.x, zi2), ym = hostg1::mul(r.y, hostg1::mul(zi2, zi));
        hostg1::store_jac(out_jac, hostg1::Jac{xm, ym, hostg1::ONE});
        if (out_aff) {
            hostg1::store_words(out_aff, hostg1::from_mont(xm));
            hostg1::store_words(out_aff + 8, hostg1::from_mont(ym));
        }
        return;
    }
    if (out_jac) hostg1::store_jac(out_jac, r);
    if (out_aff) {  // the only inversion of the whole call (~10 us): callers that want the reference's result type
                    // (Jacobian, metal_msm.rs:228-241) pass NULL and skip it
        hostg1::Fq x, y;
        (void)hostg1::to_affine_std(r, x, y);
        hostg1::store_words(out_aff, x);
        hostg1::store_words(out_aff + 8, y);
    }
}
// msm_bn254_g1_combine with the representative chosen by the caller's flags (the multi-GPU fold of a MSM_FLAG_DETERMINISTIC handle)
int32_t combine_partials(const uint32_t* partials, size_t k, uint32_t* out_jac, uint32_t* out_aff, uint8_t* out_inf, bool canonical) {
    if (!partials) return MSM_ERR_BAD_ARG;
    if (k == 0) return MSM_ERR_EMPTY;
    hostg1::Jac total = hostg1::identity();
    for (size_t i = 0; i < k; i++) total = hostg1::jadd(total, hostg1::load_jac(partials + i * 24));  // fixed rank order
    finish_outputs(total, out_jac, out_aff, out_inf, canonical);
    return MSM_OK;
}

// Everything the three enqueue steps of one (chunk of an) MSM share.  The PLAN (window width, digit form, GLV split: what the
// bucket array looks like) is made for plan_n points -- the whole instance, also when only a chunk of it is sorted and
// accumulated -- the sort geometry follows the points at hand.
struct PipeState {
    msm_plan_t pl{};
    size_t n_real = 0, n = 0;  // real / virtual (x2 with the GLV split) points of this chunk: what k_decompose sees
    uint32_t W = 0, nb = 0, cbits = 0, kb = 0;  // decomposition: W windows of cbits bits, nb = 2^kb buckets per bucket array
    // SORT / ACCUMULATE view.  Without a window table every window has its own bucket array: sW = W sort windows of sn = n entries.
    // With the window table of a resident base set (factor tf: T_j = 2^(c*j) P, j < tf; k_table_next) the tf windows of a group share
    // one array: sW = W / tf sort windows of sn = tf * n entries each, and a sorted entry is the TABLE index j * n + i.
    uint32_t tf = 1, sW = 0;
    size_t sn = 0;
    uint32_t top_shift = 0;  // full table (tf == W): the short top window's digits enter as d * 2^top_shift (table_top_shift)
    uint32_t top_bits = 0;   // == pl.top_digit_bits: index bits of the top window that carry the digit; the kb - top_bits above it carry point-index
                             // bits (split plans: msmplan::glv_top_digit_bits), which host_finish leaves out
    // REDUCTION view: arrays of more than 2^17 buckets are reduced as 2^pw_bits PSEUDO-windows of 2^rkb buckets each (bucket index
    // b = q * 2^rkb + b'); the host adds q * 2^rkb * (plain sum of pseudo-window q) back in (host_finish).  rW = sW << pw_bits.
    uint32_t rW = 0, rkb = 0, pw_bits = 0, kb_lo = 0, kb_hi = 0, n_lo = 0, n_hi = 0;
    size_t pairs = 0, tb = 0;  // sorted entries at most (= W * n = sW * sn), buckets in all (= sW * nb)
    // k_accumulate_pieces' work items: a whole bucket of at most pmax entries, or a run of psplit entries of a longer one
    uint32_t pmax = 0, psplit = 0;
    size_t maxpieces = 0, maxpartials = 0;
};


// plan + workspace.  May reallocate buffers (hipFree synchronises the device), so with chunks in flight it must not grow
// anything: the first chunk of a streamed MSM is the largest.
// first != nullptr: a later chunk / point range of an instance whose FIRST (largest) one was prepared as *first -- the piece lengths are
// fixed once per instance, so that a smaller chunk can never need more pieces or partial-sum slots than the workspace sized for the first
// one holds (lengths re-derived per chunk could outgrow the 12.5 % slack of ensure() and make it reallocate -- a device
// synchronisation -- with chunks in flight).
int32_t pipe_prepare(msm_ctx* c, size_t n_real, size_t plan_n, uint32_t extra_flags, hipStream_t st, PipeState* ps, uint32_t table_c = 0,
                     uint32_t table_f = 1, const PipeState* first = nullptr) {
    if (trace_enabled() && c->flags_clean) c->t_prepare = std::chrono::steady_clock::now();  // (flags_clean: first prepare of a call)
    if (n_real > 0x3FFFFFFFull) return fail(c, MSM_ERR_BAD_ARG, "n = %zu exceeds 2^30-1 points per context call", n_real);
    // (a table call re-derives the plan its upload made: same width, and the split wherever make_table_plan allowed it -- the caller
    // passes MSM_FLAG_NO_GLV when the uploaded set has no phi records)
    const size_t table_glv = std::max(c->knobs.glv_max, c->knobs.table.glv_max ? c->knobs.table.glv_max : msmplan::TABLE_GLV_MAX_POINTS);
    int32_t rc = table_c ? make_plan(plan_n ? plan_n : n_real, table_c, c->cfg.flags | extra_flags, &ps->pl, table_glv)
                         : ctx_plan(c, plan_n ? plan_n : n_real, extra_flags, &ps->pl);
    if (rc) return fail(c, rc, "bad window_bits/flags (%u, 0x%x)", table_c ? table_c : c->cfg.window_bits, c->cfg.flags);
    const msm_plan_t& pl = ps->pl;
    ps->n_real = n_real;
    const size_t n = ps->n = pl.glv ? 2 * n_real : n_real;  // what the sort, the accumulation and the reduction see
    const uint32_t W = ps->W = pl.num_windows, nb = ps->nb = pl.num_buckets;
    ps->cbits = pl.window_bits;
    ps->tf = table_f ? table_f : 1u;
    if (W % ps->tf) return fail(c, MSM_ERR_BAD_ARG, "internal: table factor %u does not divide %u windows", ps->tf, W);
    ps->sW = W / ps->tf;
    ps->sn = (size_t)ps->tf * n;
    ps->top_shift = table_top_shift(pl, ps->tf);
    if (ps->tf > 1) ps->pl.top_digit_bits = ilog2(nb);  // (make_plan does not know of the table: its top window is spread by top_shift)
    ps->top_bits = pl.top_digit_bits;
    const size_t pairs = ps->pairs = (size_t)W * n, tb = ps->tb = (size_t)ps->sW * nb;
    if (pairs > 0xFFFFFFFFull) return fail(c, MSM_ERR_BAD_ARG, "n*W = %zu does not fit 32-bit offsets", pairs);
    ps->kb = ilog2(nb);
    ps->pw_bits = ps->kb > 17 ? ps->kb - 16 : 0;  // up to 2^17 buckets the reduction kernels take an array whole
    ps->rkb = ps->kb - ps->pw_bits;
    ps->rW = ps->sW << ps->pw_bits;
    if ((size_t)ps->rW * (ps->rkb + 1) > MAX_QSUM_POINTS) return fail(c, MSM_ERR_BAD_ARG, "window_bits %u: too many bit sums for the result buffer", ps->cbits);
    ps->kb_lo = ps->rkb / 2, ps->kb_hi = ps->rkb - ps->kb_lo;  // bucket index inside a (pseudo-)window = hi * n_lo + lo
    ps->n_lo = 1u << ps->kb_lo, ps->n_hi = 1u << ps->kb_hi;
    const uint32_t ntiles = (uint32_t)((tb + msmk::SCAN_TILE - 1) / msmk::SCAN_TILE);
    // k_accumulate_pieces' work items (msmplan::make_piece_plan): whole buckets up to pmax entries, runs of psplit entries of longer ones
    {
        msmplan::piece_plan fp;
        if (first) fp.pmax = first->pmax, fp.psplit = first->psplit;
        const msmplan::piece_plan pp = msmplan::make_piece_plan(pairs, ps->sn / nb, tb, c->knobs.piece_len, first ? &fp : nullptr, c->knobs.split_target, c->knobs.split_shift);
        ps->pmax = pp.pmax, ps->psplit = pp.psplit, ps->maxpieces = pp.max_pieces, ps->maxpartials = pp.max_partials;
    }
    const size_t maxpartials = ps->maxpartials;
    if ((rc = ensure(c, c->digits, pairs * 4))) return rc;
    if ((rc = ensure(c, c->sorted, pairs * 4))) return rc;
    if ((rc = ensure(c, c->hist, tb * 4))) return rc;
    if ((rc = ensure(c, c->offsets, (tb + 1) * 4))) return rc;
    if ((rc = ensure(c, c->blocksums, ((size_t)ntiles + 1) * 4))) return rc;
    if ((rc = ensure(c, c->buckets, tb * XB))) return rc;
    if ((rc = ensure(c, c->sorttmp, pairs * 4))) return rc;  // stages the two-level sort
    if ((rc = ensure(c, c->partials, maxpartials * XB))) return rc;
    if ((rc = ensure(c, c->plist, ps->maxpieces * 16))) return rc;
    if ((rc = ensure(c, c->pbase, tb * 4))) return rc;
    // (no memset: every sort chain zeroes the bins itself, at its head -- msmk::clear_piece_bins in k_decompose* / k_coarse_hist)
    if ((rc = ensure(c, c->phist, (msmk::PIECE_BINS + 1) * 4))) return rc;
    if ((rc = ensure(c, c->pcursor, (msmk::PIECE_BINS + 1) * 4))) return rc;
    {   // a long bucket owns >= LONG_SPAN partial sums and gets one (bucket, segment) entry per LONG_SEG of them
        const size_t entries = maxpartials / msmk::LONG_SPAN + maxpartials / msmk::LONG_SEG + 32;
        if ((rc = ensure(c, c->longlist, entries * 8))) return rc;
        const size_t had = c->longdone.cap;
        if ((rc = ensure(c, c->longdone, entries * 4))) return rc;
        if (c->longdone.cap != had) HIPCHK(c, hipMemsetAsync(c->longdone.p, 0, c->longdone.cap, st));  // self-cleaning afterwards
    }
    // two lists: the two-piece buckets from entry 0 (at most tb of them), those of 3 .. LONG_SPAN-1 pieces from entry tb (piece_tally_publish)
    if ((rc = ensure(c, c->midlist, (tb + std::min(tb, maxpartials / 3) + 16) * 4))) return rc;
    if ((rc = ensure(c, c->rc, (tb + tb / 2 + 4) * XB))) return rc;  // two families x (1/2 + 1/4) ping-pong levels
    if ((rc = ensure(c, c->flags, 64))) return rc;
    return MSM_OK;
}

// the run length of very long buckets as the kernels take it (msmk::effective_psplit): the plan's psplit, the shift that shortens it for instances with few
// entries, and "fixed" when a test forces the lengths
inline uint32_t split_arg(const msm_ctx* c, const PipeState& ps) { return msmk::psplit_arg(ps.psplit, c->knobs.split_shift, c->knobs.piece_len != 0); }

// K1b: digits + signed recode of one (chunk of an) MSM on stream st (with the GLV split: two halves below 7 * 2^123 per scalar, 2*n_real digit
// columns).  Needs the scalars (and the infinity mask), NOT the bases.  first = false: a later chunk of a streamed MSM (error bits and
// the running count of additions survive).
struct SortGeom {  // counting-sort plan of (sW sort windows, sn entries each, nb buckets per window)
    bool tiled = false, two_level = false, lds_counts = false;
    bool d16 = false;   // the digits travel as 16-bit codes (msm_kernels.hpp DIGIT16_*): two-level sort, <= 2^15 buckets per window, no window table
    uint32_t drow = 0;  // entries per row of the digit array (sn, padded to an even number for 16-bit codes)
    uint32_t coarse_bits = 0, fine_bits = 0, idx_bits = 0, ncoarse = 0, NS = 0, nsuper = 1;
};
SortGeom sort_geometry(const msm_ctx* c, const PipeState& ps) {
    SortGeom g;
    const size_t sn = ps.sn;
    const uint32_t kb = ps.kb;
    // Counting-sort plan: when one window's histogram fits LDS (nb <= 32768) the bucket counts and arrival
    // ranks come from per-tile LDS histograms, otherwise from device-scope atomics in k_decompose.
    g.tiled = (size_t)ps.nb * 4 <= LDS_HIST_BYTES;
    // two-level LDS sort: coarse = top bits of the bucket index, fine = the rest (<= 7 bits; up to 9 for windows of 2^18 and 2^19 buckets)
    // (8 coarse bits up to n = 2^21, then 9 and 10, so that a (window, coarse bin) region stays ~8192 elements and
    // fits the fine sort's LDS staging)
    uint32_t coarse_bits = 8;
    while (coarse_bits < 10 && (sn >> coarse_bits) > 8192) coarse_bits++;
    if (kb > coarse_bits + 7) coarse_bits = std::min(10u, kb - 7);  // wide windows: the fine part stays 7 bits up to 2^17 buckets
    if (coarse_bits > kb) coarse_bits = kb;
    g.coarse_bits = coarse_bits;
    g.fine_bits = kb - coarse_bits;
    g.idx_bits = 31 - g.fine_bits;
    g.ncoarse = 1u << coarse_bits;
    // a sort window longer than 2^idx_bits entries is cut into super-tiles whose number is recovered from an element's position (sort_hi)
    g.nsuper = (uint32_t)((sn + ((size_t)1 << g.idx_bits) - 1) >> g.idx_bits);
    if (g.nsuper == 0) g.nsuper = 1;
    g.two_level = g.fine_bits <= 9 && g.nsuper <= msmk::SUPER_MAX && !c->knobs.direct_scatter;
    g.lds_counts = g.two_level || g.tiled;  // no device-scope histogram / rank atomics in k_decompose
    g.NS = (uint32_t)((sn + msmk::SUBTILE - 1) / msmk::SUBTILE);
    g.d16 = g.two_level && ps.tf == 1 && kb <= 15;
    g.drow = (uint32_t)(g.d16 ? (sn + 1) & ~(size_t)1 : sn);  // (d16: tf == 1, a sort window is a decomposition window)
    return g;
}

// phi_src / phi_dst (split plans, both or neither): the decomposition also writes the phi records of arkworks-form bases (k_decompose_glv<.., PHI>)
int32_t enqueue_decompose(msm_ctx* c, const PipeState& ps, const uint8_t* d_inf, const uint32_t* d_scalars, uint32_t scalars_mont,
                          hipStream_t st, bool first, const uint32_t* phi_src = nullptr, uint32_t* phi_dst = nullptr) {
    Range r_("msm:decompose");
    int32_t rc;
    const msm_plan_t& pl = ps.pl;
    const size_t n_real = ps.n_real, pairs = ps.pairs, tb = ps.tb;
    const uint32_t W = ps.W, nb = ps.nb, cbits = ps.cbits;
    uint32_t* hist = (uint32_t*)c->hist.p;
    uint32_t* flags = (uint32_t*)c->flags.p;
    const SortGeom sg = sort_geometry(c, ps);
    if (!sg.lds_counts && ps.tf > 1) return fail(c, MSM_ERR_BAD_ARG, "window_bits %u: the window table needs the LDS sort paths", cbits);
    if (!sg.lds_counts && (rc = ensure(c, c->ranks, pairs * 4))) return rc;
    if (!sg.lds_counts) HIPCHK(c, hipMemsetAsync(hist, 0, tb * 4, st));
    // The flag words clean themselves (a hipMemsetAsync is its own dispatch: ~5 us + a ~12 us bubble in front of it, per call and per
    // streamed chunk): k_decompose zeroes the list counters of the chunk, the kernel that ends an MSM zeroes the error and count
    // words after copying them out.  Only a context whose last call did not complete (error paths) is cleaned from the host.
    if (first && !c->flags_clean) {  // (error recovery only; waited for, so that no kernel of this call on the OTHER stream -- k_ark_repack raises an error bit -- can be overtaken by it)
        HIPCHK(c, hipMemsetAsync(flags, 0, 64, st));
        HIPCHK(c, hipStreamSynchronize(st));
    }
    if (first) c->flags_clean = false;
    if (c->stage_timing) HIPCHK(c, hipEventRecord(c->ev[EV_CONVERT], st));
    void* dg = c->digits.p;
    uint32_t *rk = (uint32_t*)c->ranks.p, *ph = (uint32_t*)c->phist.p, *pcu = (uint32_t*)c->pcursor.p;
    dim3 g = grid1(n_real, 256);
    const uint32_t nr = (uint32_t)n_real, drow = sg.d16 ? sg.drow : (uint32_t)ps.n;  // row of ONE window (a sort window is tf of them)
    const uint32_t spread_mask = ps.top_bits < ps.kb ? (1u << (ps.kb - ps.top_bits)) - 1u : 0u;  // (plans without a table)
#define MSM_DECOMP_ARGS d_scalars, d_inf, nr, cbits, W, nb, hist, dg, drow, rk, flags, scalars_mont, ps.top_shift, ps.top_bits, spread_mask, ph, pcu
#define MSM_DECOMP_GLV_ARGS d_scalars, d_inf, nr, cbits, W, dg, drow, flags, scalars_mont, ps.top_shift, ps.top_bits, spread_mask, ph, pcu, phi_src, phi_dst
    if (pl.glv) {
        if (!sg.lds_counts) return fail(c, MSM_ERR_BAD_ARG, "window_bits %u needs the non-GLV path (MSM_FLAG_NO_GLV)", cbits);
        const bool phi = phi_src != nullptr && phi_dst != nullptr;
#define MSM_DECOMP_GLV(S, D) \
    do { \
        if (phi) msmk::k_decompose_glv<S, D, true><<<g, 256, 0, st>>>(MSM_DECOMP_GLV_ARGS); \
        else msmk::k_decompose_glv<S, D, false><<<g, 256, 0, st>>>(MSM_DECOMP_GLV_ARGS); \
    } while (0)
        if (pl.signed_digits && sg.d16) MSM_DECOMP_GLV(true, true);
        else if (pl.signed_digits) MSM_DECOMP_GLV(true, false);
        else if (sg.d16) MSM_DECOMP_GLV(false, true);
        else MSM_DECOMP_GLV(false, false);
#undef MSM_DECOMP_GLV
    } else if (pl.signed_digits && sg.d16) msmk::k_decompose<true, false, true><<<g, 256, 0, st>>>(MSM_DECOMP_ARGS);
    else if (pl.signed_digits && sg.lds_counts) msmk::k_decompose<true, false, false><<<g, 256, 0, st>>>(MSM_DECOMP_ARGS);
    else if (pl.signed_digits) msmk::k_decompose<true, true, false><<<g, 256, 0, st>>>(MSM_DECOMP_ARGS);
    else if (sg.d16) msmk::k_decompose<false, false, true><<<g, 256, 0, st>>>(MSM_DECOMP_ARGS);
    else if (sg.lds_counts) msmk::k_decompose<false, false, false><<<g, 256, 0, st>>>(MSM_DECOMP_ARGS);
    else msmk::k_decompose<false, true, false><<<g, 256, 0, st>>>(MSM_DECOMP_ARGS);
#undef MSM_DECOMP_ARGS
#undef MSM_DECOMP_GLV_ARGS
    if (c->stage_timing) HIPCHK(c, hipEventRecord(c->ev[EV_DECOMP], st));
    return MSM_OK;
}

// K2 + chunk map on stream st: the digits of ps.sW sort windows x ps.sn entries -> offsets / sorted /
// chunk owners.  into = true: the bucket array keeps what earlier chunks / window ranges of the same MSM left in it.
int32_t enqueue_sort(msm_ctx* c, const PipeState& ps, hipStream_t st, bool into) {
    Range r_("msm:sort");
    int32_t rc;
    const size_t sn = ps.sn, tb = ps.tb;
    const uint32_t sW = ps.sW, nb = ps.nb;
    const uint32_t ntiles = (uint32_t)((tb + msmk::SCAN_TILE - 1) / msmk::SCAN_TILE);
    uint32_t* hist = (uint32_t*)c->hist.p;
    uint32_t* offsets = (uint32_t*)c->offsets.p;
    uint32_t* flags = (uint32_t*)c->flags.p;
    const uint32_t* digits = (const uint32_t*)c->digits.p;  // (32-bit codes on every path but the two-level sort with sg.d16)
    const SortGeom sg = sort_geometry(c, ps);
    const uint32_t coarse_bits = sg.coarse_bits, fine_bits = sg.fine_bits, idx_bits = sg.idx_bits, ncoarse = sg.ncoarse, NS = sg.NS;
    uint32_t T = 1, tile_len = (uint32_t)sn;
    c->last_sort_path = sg.two_level ? 2u : sg.tiled ? 1u : 0u;
    if (sg.two_level) {
        if ((rc = ensure(c, c->bigslot, (size_t)sW * ncoarse * 4))) return rc;
        if ((rc = ensure(c, c->big, msmk::BIG_WORDS * 4))) return rc;
        if ((rc = ensure(c, c->ccounts, (size_t)sW * ncoarse * NS * 4))) return rc;
        if ((rc = ensure(c, c->cregion, ((size_t)sW * ncoarse * 3 + 4) * 4))) return rc;  // region totals, region starts (+ 1), non-empty buckets per region (k_fine_sort -> k_place_count)
        const uint32_t nregions = sW * ncoarse;
        uint32_t* counts = (uint32_t*)c->ccounts.p;
        uint32_t* rtotal = (uint32_t*)c->cregion.p;
        uint32_t* rstart = rtotal + nregions;
        uint32_t* tmp = (uint32_t*)c->sorttmp.p;  // staging copy of the two-level sort
        uint32_t *ph = (uint32_t*)c->phist.p, *pcu = (uint32_t*)c->pcursor.p;
        if (sg.d16) msmk::k_coarse_hist<true><<<dim3(NS, sW), msmk::TILE_BLOCK, 0, st>>>(c->digits.p, sg.drow, counts, (uint32_t)sn, fine_bits, ncoarse, NS, flags, ph, pcu);
        else msmk::k_coarse_hist<false><<<dim3(NS, sW), msmk::TILE_BLOCK, 0, st>>>(c->digits.p, sg.drow, counts, (uint32_t)sn, fine_bits, ncoarse, NS, flags, ph, pcu);
        msmk::k_coarse_prefix<<<grid1(nregions, msmk::PREFIX_REGIONS), 1024, 0, st>>>(counts, rtotal, NS, ncoarse, nregions);
        // regions too large for one workgroup's staging area are cut into batches that worker blocks share (skewed scalars)
        uint32_t fine_block = (sn >> coarse_bits) <= 1024 ? 256u : (sn >> coarse_bits) <= 2048 ? 512u : 1024u;
        if (fine_block < (1u << fine_bits)) fine_block = 512u;  // one thread per fine bin in the scans (8 and 9 fine bits)
        const uint32_t fine_cap = fine_block * 16u;
        uint32_t* bigslot = (uint32_t*)c->bigslot.p;
        uint32_t* big = (uint32_t*)c->big.p;
        // (a region up to four staging areas long is still sorted by its owner, batch after batch, placing directly: the worker-block
        // path pays a global cursor add per bucket and batch and is meant for the few huge regions of skewed scalars -- with every
        // region ~3 x oversized (2^22 points x 13 windows in one array) it took 5 ms where the owners take 0.6)
        msmk::k_coarse_starts<<<1, msmk::SCAN_BLOCK, 0, st>>>(rtotal, rstart, nregions, flags + msmk::FLAG_PAIRS, offsets + tb, bigslot, big, 4 * fine_cap, fine_cap);
        if (sg.d16) msmk::k_coarse_scatter<true><<<dim3(NS, sW), msmk::TILE_BLOCK, 0, st>>>(c->digits.p, sg.drow, counts, rstart, tmp, (uint32_t)sn, fine_bits, idx_bits, ncoarse, NS);
        else msmk::k_coarse_scatter<false><<<dim3(NS, sW), msmk::TILE_BLOCK, 0, st>>>(c->digits.p, sg.drow, counts, rstart, tmp, (uint32_t)sn, fine_bits, idx_bits, ncoarse, NS);
        // workgroup size by mean region size (a workgroup stages up to 16 elements per thread); grid.x = the window's regions +
        // BIG_WORKERS_X worker blocks for the batches of oversized regions, which k_big_place then places
        const uint32_t wx = std::max(msmk::BIG_WORKERS_X, (msmk::BIG_WORKERS_MIN + sW - 1) / sW);  // worker blocks per sort window
        const dim3 gf(ncoarse + wx, sW);
        uint32_t* srt = (uint32_t*)c->sorted.p;
        const msmk::sort_hi hi{counts, NS, sg.nsuper, (uint32_t)(((size_t)1 << idx_bits) / msmk::SUBTILE)};
        // ... in the SAME launch that tallies the accumulation's pieces (k_place_count, round 6: k_big_place was an empty launch on uniform scalars,
        // 4.5 us of the dependent chain at every size).  The sort stage ends with the fine sort (EV_SORT); the merged launch belongs to plan_ms.
        const uint32_t count_blocks = (uint32_t)((tb + msmk::PLACE_COUNT_SPAN - 1) / msmk::PLACE_COUNT_SPAN);
#define MSM_FINE_AND_PLACE(FB) \
    do { \
        msmk::k_fine_sort<FB><<<gf, FB, 0, st>>>(tmp, rstart, offsets, srt, nb, fine_bits, idx_bits, ncoarse, bigslot, big, hi, rstart + nregions + 2); \
        if (c->stage_timing) HIPCHK(c, hipEventRecord(c->ev[EV_SORT], st)); \
        msmk::k_place_count<FB><<<count_blocks + wx * sW, FB, 0, st>>>(tmp, rstart, offsets, srt, nb, fine_bits, idx_bits, ncoarse, bigslot, big, hi, count_blocks, wx, sW, \
                                                                  (uint32_t)tb, ps.pmax, split_arg(c, ps), (uint32_t*)c->phist.p, flags, (uint32_t*)c->longlist.p, \
                                                                  (uint32_t*)c->midlist.p, (uint32_t*)c->pbase.p, (uint32_t*)c->buckets.p, into ? 1u : 0u); \
    } while (0)
        if (fine_block == 256) MSM_FINE_AND_PLACE(256);
        else if (fine_block == 512) MSM_FINE_AND_PLACE(512);  // (up to a mean of 2048: at 4096 the 512-thread variant is 1.5 us faster on uniform scalars only)
        else MSM_FINE_AND_PLACE(1024);
#undef MSM_FINE_AND_PLACE
    } else {
        // K2/1: per-tile LDS histograms, then per-bucket prefix over tiles
        if (sg.tiled) {
            T = (uint32_t)((sn + 65535) / 65536);
            if (T > 64) T = 64;
            tile_len = (uint32_t)((sn + T - 1) / T);
            if ((rc = ensure(c, c->tilecounts, (size_t)sW * T * nb * 4))) return rc;
            msmk::k_tile_hist<<<dim3(T, sW), msmk::TILE_BLOCK, (size_t)nb * 4, st>>>(digits, (uint32_t*)c->tilecounts.p, (uint32_t)sn, nb, tile_len, T);
            msmk::k_tile_prefix<<<grid1(tb, 256), 256, 0, st>>>((uint32_t*)c->tilecounts.p, hist, nb, T, (uint32_t)tb);
        }
        // K2/2: bucket offsets
        msmk::k_scan_tiles<<<ntiles, msmk::SCAN_BLOCK, 0, st>>>(hist, offsets, (uint32_t*)c->blocksums.p, (uint32_t)tb);
        msmk::k_scan_block_sums<<<1, msmk::SCAN_BLOCK, 0, st>>>((uint32_t*)c->blocksums.p, ntiles, flags + msmk::FLAG_PAIRS);
        msmk::k_scan_add<<<grid1(tb, 256), 256, 0, st>>>(offsets, (uint32_t*)c->blocksums.p, (uint32_t)tb, flags + msmk::FLAG_PAIRS);
        // K2/3: scatter
        if (sg.tiled) {
            msmk::k_tile_scatter<<<dim3(T, sW), msmk::TILE_BLOCK, (size_t)nb * 4, st>>>(digits, offsets, (uint32_t*)c->tilecounts.p,
                                                                                      (uint32_t*)c->sorted.p, (uint32_t)sn, nb, tile_len, T);
        } else {
            dim3 g((unsigned)((sn + 255) / 256), sW);
            msmk::k_scatter<<<g, 256, 0, st>>>(digits, (uint32_t*)c->ranks.p, offsets, (uint32_t*)c->sorted.p, (uint32_t)sn, nb);
        }
    }
    // the piece list, longest first (its histogram and bin cursors were zeroed at the head of this chain: msmk::clear_piece_bins)
    if (!sg.two_level) {  // (the two-level sort tallied the pieces in its last launch, k_place_count)
        if (c->stage_timing) HIPCHK(c, hipEventRecord(c->ev[EV_SORT], st));
        msmk::k_piece_count<<<grid1(tb, 1024), 1024, 0, st>>>(offsets, (uint32_t)tb, ps.pmax, split_arg(c, ps), (uint32_t*)c->phist.p, flags, (uint32_t*)c->longlist.p,
                                                            (uint32_t*)c->midlist.p, (uint32_t*)c->pbase.p, (uint32_t*)c->buckets.p, into ? 1u : 0u);
    }
    msmk::k_piece_scatter<<<grid1(tb, 1024), 1024, 0, st>>>(offsets, (uint32_t)tb, ps.pmax, split_arg(c, ps), (const uint32_t*)c->phist.p, (uint32_t*)c->pcursor.p,
                                                          (const uint32_t*)c->pbase.p, (uint4*)c->plist.p, flags);
    if (c->stage_timing) HIPCHK(c, hipEventRecord(c->ev[EV_PLAN], st));  // msm_timings_t.plan_ms
    return MSM_OK;
}

// K1b + K2 of one (chunk of an) MSM: what every path without a window table queues before its accumulation
int32_t enqueue_digits_sort(msm_ctx* c, const PipeState& ps, const uint8_t* d_inf, const uint32_t* d_scalars, uint32_t scalars_mont,
                            hipStream_t st, bool first) {
    int32_t rc = enqueue_decompose(c, ps, d_inf, d_scalars, scalars_mont, st, first);
    if (rc) return rc;
    return enqueue_sort(c, ps, st, !first);
}

// Where k_accumulate_pieces gathers its base records from.  m256 = false: `rec` holds INTERNAL-domain packed records (with the GLV split
// 2*n_real of them, phi(P_i) at index n_real + i) -- resident sets, window tables, standard-form and struct inputs (k_convert_bases /
// k_import_ark / k_decompress wrote them).  m256 = true (round 5): `rec` are the CALLER's arkworks words (R = 2^256 Montgomery) as they
// arrived, entries from nsplit on are phi records in the same form (k_phi_records; split plans only).
struct BaseSrc {
    const uint32_t* rec = nullptr;
    const uint32_t* phi = nullptr;
    uint32_t nsplit = 0xFFFFFFFFu;
    bool m256 = false;
    uint32_t* phi_fill = nullptr;  // m256, split plan: == phi, and nothing has been launched for it yet -- the DECOMPOSITION fills it (k_decompose_glv<.., PHI>)
    BaseSrc() = default;
    BaseSrc(const uint32_t* internal_records) : rec(internal_records) {}  // NOLINT: the internal-domain form converts implicitly
    BaseSrc shifted(size_t lo) const {  // the point range starting at lo (unsplit plans only)
        BaseSrc s = *this;
        s.rec = rec + lo * 16;
        return s;
    }
};

// K3: bucket accumulation (the graded kernel) -- bracketed by its own events on its own stream -- and the buckets cut by chunk
// borders.  d_bases: INTERNAL-domain records; with the GLV split 2*n_real of them, phi(P_i) at index n_real + i.
// into = true: the buckets keep what earlier chunks of the same MSM left in them (k_accumulate<true, true>); chunked: a chunk of a
// streamed host call (own kernel symbol).
int32_t enqueue_accumulate(msm_ctx* c, const PipeState& ps, const BaseSrc& src, hipStream_t st, hipEvent_t bases_ready, bool into,
                           bool chunked = false) {
    Range r_("msm:accumulate");
    uint32_t* flags = (uint32_t*)c->flags.p;
    uint32_t* offsets = (uint32_t*)c->offsets.p;
    if (bases_ready) HIPCHK(c, hipStreamWaitEvent(st, bases_ready, 0));  // d_bases is being converted on another stream
    // The graded kernel's own pair of events rides on its DISPATCH (hipExtLaunchKernelGGL: start / stop are the kernel's begin and end
    // timestamps, no barrier packets): two hipEventRecord calls around it cost ~6 us of stream time each (kernel traces of rounds 1-3).
    const dim3 gp = grid1(ps.maxpieces, 256);  // (threads beyond the number of pieces, known on the device only, leave at once)
    const uint32_t *srt = (const uint32_t*)c->sorted.p, *np = flags + msmk::FLAG_PIECES;
    const uint4* pl = (const uint4*)c->plist.p;
    uint32_t *bk = (uint32_t*)c->buckets.p, *pt = (uint32_t*)c->partials.p;
    unsigned long long* clk = (unsigned long long*)c->clk.p;  // clock probe of the launch's first workgroup (msm_get_clock_stats)
    const hipEvent_t e0 = c->ev[EV_ACC0], e1 = c->ev[EV_ACC1];
    // every ktime_every-th launch carries the pair of events (msm_set_kernel_timing): they ride on the dispatch, but a dispatch that is timed does not
    // overlap its neighbours' launch latency -- ~6.6 us in front of the kernel and ~4.6 behind it (profiles/r5_final_call_timeline_2p20_2p17.txt)
    const bool timed = c->stage_timing || (c->ktime_every && (c->ktime_count++ % c->ktime_every) == 0);
    c->acc_last_timed = timed;
#define MSM_ACC_LAUNCH(INTO, CHUNK, M256) \
    do { \
        if (timed) hipExtLaunchKernelGGL((msmk::k_accumulate_pieces<INTO, CHUNK, M256>), gp, dim3(256), 0, st, e0, e1, 0, src.rec, phi_biased, src.nsplit, srt, pl, np, bk, pt, clk); \
        else hipLaunchKernelGGL((msmk::k_accumulate_pieces<INTO, CHUNK, M256>), gp, dim3(256), 0, st, src.rec, phi_biased, src.nsplit, srt, pl, np, bk, pt, clk); \
    } while (0)
    // the kernel indexes the phi records by the ENTRY (nsplit + i): hand it the array biased by -nsplit records (never dereferenced below nsplit)
    const uint32_t* phi_biased = src.phi ? (const uint32_t*)((uintptr_t)src.phi - (uintptr_t)src.nsplit * 64u) : src.rec;
    if (src.m256) {
        if (into) MSM_ACC_LAUNCH(true, true, true);
        else if (chunked) MSM_ACC_LAUNCH(false, true, true);
        else MSM_ACC_LAUNCH(false, false, true);
    } else {
        if (into) MSM_ACC_LAUNCH(true, true, false);
        else if (chunked) MSM_ACC_LAUNCH(false, true, false);
        else MSM_ACC_LAUNCH(false, false, false);
    }
#undef MSM_ACC_LAUNCH
    msmk::k_combine_pieces<<<dim3(msmk::LONG_BLOCKS + msmk::MID_BLOCKS + msmk::MID2_BLOCKS), msmk::COMBINE_BLOCK, 0, st>>>(
        offsets, pt, bk, ps.pmax, split_arg(c, ps), (const uint32_t*)c->pbase.p, flags + msmk::FLAG_MID, flags + msmk::FLAG_MID2, (const uint32_t*)c->midlist.p, (uint32_t)ps.tb,
        flags + msmk::FLAG_LONG,
        (const uint32_t*)c->longlist.p, (uint32_t*)c->longdone.p, c->knobs.mid_lane_min, flags + msmk::FLAG_PAIRS);
    if (c->stage_timing) HIPCHK(c, hipEventRecord(c->ev[EV_COMBINE], st));  // msm_timings_t.combine_ms
    return MSM_OK;
}

// K4/K5: bucket reduction -- plain row/column sums by dense pairwise levels, then per-bit sums; the weights are applied on
// the host.  The W*(kb+1) bit sums and the flag words are written by the last kernel straight into h_qsums_dst / h_flags_dst
// (pinned host memory).  No host synchronisation here.
int32_t enqueue_reduce(msm_ctx* c, const PipeState& ps, hipStream_t st, uint32_t* h_qsums_dst, uint32_t* h_flags_dst) {
    Range r_("msm:reduce");
    const size_t tb = ps.tb;
    // (pseudo-)windows of 2^kb buckets: the whole bucket array of a window up to 2^17 buckets, 2^16-bucket slices of it above
    const uint32_t W = ps.rW, kb = ps.rkb, kb_lo = ps.kb_lo, kb_hi = ps.kb_hi, n_lo = ps.n_lo, n_hi = ps.n_hi;
    uint32_t* flags = (uint32_t*)c->flags.p;
    const uint32_t* bk = (const uint32_t*)c->buckets.p;
    // ping-pong buffers per family: [0, tb/2) and [tb/2, tb/2 + tb/4) elements
    uint32_t* rbuf[2] = {(uint32_t*)c->rc.p, (uint32_t*)c->rc.p + (tb / 2) * msmk::XW};
    uint32_t* cbuf[2] = {(uint32_t*)c->rc.p + (tb / 2 + tb / 4 + 1) * msmk::XW, (uint32_t*)c->rc.p + (tb + tb / 4 + 1) * msmk::XW};
    const uint32_t *rin = bk, *cin = bk;
    size_t rn = tb, cn = tb;  // current element counts
    const uint32_t levels = kb_hi > kb_lo ? kb_hi : kb_lo;
    // Round 2: the first three levels (ALU-bound, most of the adds) run as ONE launch that reads every bucket once per family and never
    // writes the two intermediate levels (-14..-16 us of kernel time at every size up to 2^20).  Round 2 kept it below ~100 MB of buckets (c = 17,
    // 15 x 65536 buckets = 141 MB: its strided record reads lost 20-35 us to the pairwise levels there); with the next record in flight during an addition
    // (round 6) it wins there too: reduce 0.308 -> 0.298 ms at 2^21, 0.324 -> 0.307 at 2^22 (MSM_HIP_PAIR8_MAX_MB of the hooks build) => up to 200 MB.  A matching single launch for the
    // LAST levels (one LDS tree of eight-lane additions per output) was slower than the launch-bound k_pair_level_wide levels it
    // replaced (2^20: 52 vs 37 us: a tree's upper levels leave most lanes of its wavefront idle): profiles/NOTES_r2.md.
    uint32_t l = 0;
    if (levels >= 3 && kb_lo >= 3 && tb * XB <= ((size_t)c->knobs.pair8_max_mb << 20)) {
        rn = tb / 8, cn = tb / 8;
        // lanes per output: one lane while every SIMD has a wavefront of outputs (the kernel is multiplier bound then: 8 x 32768 buckets,
        // 2^14 .. 2^20 points: 1 / 2 / 4 lanes 0.471 / 0.470 / 0.487 ms at 2^17); below that the seven dependent additions of an output are
        // what the kernel waits for and 2 or 4 lanes cut the chain to 4 or 3 (2^12: 0.307 / 0.283 / 0.275 ms; one shared array of 2^15 buckets
        // of a window table: profiles/r3_pair8_lanes_ab.txt)
        const size_t outs = rn + cn;
        const uint32_t lanes = outs >= 65536 ? 1u : outs * 2 >= 65536 ? 2u : 4u;
        if (lanes == 1) msmk::k_pair_level8<1><<<grid1(outs, 256), 256, 0, st>>>(bk, rbuf[0], cbuf[0], (uint32_t)rn, n_lo);
        else if (lanes == 2) msmk::k_pair_level8<2><<<grid1(outs * 2, 256), 256, 0, st>>>(bk, rbuf[0], cbuf[0], (uint32_t)rn, n_lo);
        else msmk::k_pair_level8<4><<<grid1(outs * 4, 256), 256, 0, st>>>(bk, rbuf[0], cbuf[0], (uint32_t)rn, n_lo);
        rin = rbuf[0], cin = cbuf[0];  // where level 2 would have left them
        l = 3;
    }
    for (; l < levels; l++) {
        // Round 6: once a level is small enough for eight lanes per addition, ALL the levels that are left run in ONE launch, inside the workgroups'
        // LDS (k_pair_tail: 2.7 us per level instead of 5.2 per dependent launch) -- when at least two are left and the partial sums per output fit a tree
        {
            const size_t nadds_now = (l < kb_lo ? rn / 2 : 0) + (l < kb_hi ? cn / 2 : 0);
            const uint32_t np_r = l < kb_lo ? 1u << (kb_lo - l) : 1u, np_c = l < kb_hi ? 1u << (kb_hi - l) : 1u;
            if (nadds_now <= WIDE_MAX_ADDS && levels - l >= 2 && np_r <= msmk::WIDE_TREE_MAX && np_c <= msmk::WIDE_TREE_MAX) {
                const uint32_t n_rout = (uint32_t)(rn / np_r), n_cout = (uint32_t)(cn / np_c);
                const uint32_t rblocks = np_r > 1 ? (n_rout + msmk::WIDE_TREE_MAX / np_r - 1) / (msmk::WIDE_TREE_MAX / np_r) : 0u;
                const uint32_t cblocks = np_c > 1 ? (n_cout + msmk::WIDE_TREE_MAX / np_c - 1) / (msmk::WIDE_TREE_MAX / np_c) : 0u;
                // (outputs go to the ping-pong halves this level would have written -- they hold nothing that is still read and are large enough:
                // a level writes half its input, this launch at most that)
                msmk::k_pair_tail<<<rblocks + cblocks, 512, 0, st>>>(rin, rbuf[l & 1], n_rout, np_r > 1 ? np_r : 2u, cin, cbuf[l & 1], n_cout, np_c > 1 ? np_c : 2u, n_lo, rblocks);
                if (np_r > 1) rin = rbuf[l & 1], rn = n_rout;
                if (np_c > 1) cin = cbuf[l & 1], cn = n_cout;
                break;
            }
        }
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
        if (nadds <= WIDE_MAX_ADDS) msmk::k_pair_level_wide<<<grid1(nadds * msmk::WIDE_LANES, 256), 256, 0, st>>>(ja, jb);
        else msmk::k_pair_level<<<grid1(nadds, 256), 256, 0, st>>>(ja, jb);
    }
    // the bit sums (and the flag words) are written by the kernel straight into the caller's PINNED host buffers:
    // a D2H copy engine transfer started ~11 us after the kernel and took two launches (24 KB + 32 B)
    uint32_t *q_dev = nullptr, *f_dev = nullptr;
    HIPCHK(c, hipHostGetDevicePointer((void**)&q_dev, h_qsums_dst, 0));
    HIPCHK(c, hipHostGetDevicePointer((void**)&f_dev, h_flags_dst, 0));
    if (++c->done_seq == 0) c->done_seq = 1;  // (0 is what fresh pairs hold)
    if (n_hi / 2 <= msmk::WIDE_TREE_MAX && n_lo <= msmk::WIDE_TREE_MAX)
        msmk::k_reduce_bits_wide<7><<<W * (kb + 1), 512, 0, st>>>(rin, cin, q_dev, n_hi, n_lo, kb_lo, kb, flags, f_dev, c->done_seq);
    else
        msmk::k_reduce_bits<<<W * (kb + 1), 64, 0, st>>>(rin, cin, q_dev, n_hi, n_lo, kb_lo, kb, flags, f_dev, c->done_seq);
    if (c->stage_timing) HIPCHK(c, hipEventRecord(c->ev[EV_REDUCE], st));
    return MSM_OK;
}

// final_reduction (metal_msm.rs:204-261) on the CPU.  The device returns, for every (pseudo-)window q of every bucket array v, the bit
// sums Q_u (u < rkb: buckets whose index has bit u set) and the plain sum A.  With S_v = sum_b (b + 1) * B[v][b] and b = q * 2^rkb + b'
//     S_v = sum_q [ A_q + sum_u 2^u Q_q,u ]  +  2^rkb * sum_q q * A_q          (second term: arrays cut into pseudo-windows only)
// and the result is sum_v 2^(cbits * tf * v) S_v (tf windows share an array with a window table, tf = 1 without): ONE Horner chain
// over the bit positions p = cbits*tf*v + u with the terms
//     u < rkb:  sum_q Q_q,u   (+ sum_q A_q at u == 0);      rkb <= u < kb:  sum over {q : bit u-rkb of q set} of A_q
// (one doubling and about one addition per position) instead of the reference's chain per window plus c doublings between windows
// (metal_msm.rs:249-258).  The chain is cut into a few segments of geometrically shrinking length (a segment starting at position
// lo pays lo extra doublings to shift its sum), one per host thread: 2 threads reach ~60 % of the serial time, 4 threads ~45 %, more
// add nothing because the shift of the top segment is serial.  TWO threads by default: every further worker lowers the median by a
// few microseconds and raises the MEAN through 2-8 ms outliers in ~1.3 % of the calls (busy hosts; a pool of 15: 2.5 %) --
// tools/step_jitter.py.  With one shared bucket array (full window table) the chain is cbits - 1 positions long instead of 254.
hostg1::Jac host_finish(msm_ctx* c, const uint32_t* h_qsums, const PipeState& g) {
    Range r_("msm:host_finish");
    const uint32_t V = g.sW, kb = g.kb, rkb = g.rkb, PW = 1u << g.pw_bits, spacing = g.cbits * g.tf;
    const uint32_t npos = spacing * (V - 1) + (kb > 0 ? kb : 1);  // positions 0 .. npos-1 carry terms
    auto qsum = [&](uint32_t v, uint32_t q, uint32_t u) { return hostg1::load_jac(h_qsums + ((size_t)(v * PW + q) * (rkb + 1) + u) * 24); };
    auto segment = [&](uint32_t lo, uint32_t hi) {             // sum over p in [lo, hi) of 2^p * term(p)
        hostg1::Jac acc = hostg1::identity();
        for (uint32_t p = hi; p-- > lo;) {
            acc = hostg1::jdbl(acc);
            const uint32_t v = p / spacing, u = p % spacing;
            if (v == V - 1 && u >= g.top_bits) {
                // index bits of a spread top window that hold point-index bits, not the digit (msmplan::glv_top_digit_bits): no weight
            } else if (u < rkb)
                for (uint32_t q = 0; q < PW; q++) acc = hostg1::jadd(acc, qsum(v, q, u));
            else if (u < kb)
                for (uint32_t q = 0; q < PW; q++)
                    if ((q >> (u - rkb)) & 1u) acc = hostg1::jadd(acc, qsum(v, q, rkb));
            if (u == 0)
                for (uint32_t q = 0; q < PW; q++) acc = hostg1::jadd(acc, qsum(v, q, rkb));
        }
        for (uint32_t k = 0; k < lo; k++) acc = hostg1::jdbl(acc);
        return acc;
    };
    const int nseg = c->pool ? std::min<int>(c->pool->size() + 1, 8) : 1;
    if (nseg == 1 || npos < 16) return segment(0, npos);
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

constexpr int32_t ARK_RETRY_SLOW = 1;  // (internal, never handed out: see KIND_ARKFAST)
int32_t check_flags(msm_ctx* c, const uint32_t* h_flags) {
    if (h_flags[0] & 1u) return fail(c, MSM_ERR_BAD_ARG, "a scalar is >= 2^254 (not a canonical Fr element)");
    if (h_flags[0] & 2u) return fail(c, MSM_ERR_HIP, "internal: signed-digit carry out of the top window");
    if (h_flags[0] & 4u) return fail(c, MSM_ERR_HIP, "internal: a GLV half exceeds its bound (7 * 2^123)");
    if (h_flags[0] & 8u) return fail(c, MSM_ERR_HIP, "internal: a digit collides with the 16-bit skip code");
    if (h_flags[0] & 16u) return ARK_RETRY_SLOW;  // a struct of the optimistic struct-array path had its infinity flag set (run_host_input repeats the call)
    return MSM_OK;
}

void trace_line(const msm_ctx* c, const char* entry, const PipeState& ps) {
    if (!trace_enabled()) return;
    const msm_timings_t& t = c->tm;
    std::fprintf(stderr,
                 "[msm_hip] %s dev %d n %zu c %u W %u nb %u glv %u sort_path %u pieces <= %u / split %u stream_chunks %u | h2d %.3f convert %.3f "
                 "decompose %.3f sort %.3f accumulate %.3f reduce %.3f finish %.3f total %.3f ms (host enqueue %.3f ms), %llu adds\n",
                 entry, c->device, (size_t)t.num_points, ps.cbits, ps.W, ps.nb, ps.pl.glv, c->last_sort_path, ps.pmax, ps.psplit, t.stream_chunks,
                 t.h2d_ms, t.convert_ms, t.decompose_ms, t.sort_ms, t.accumulate_ms, t.reduce_ms, t.finish_ms, t.total_ms,
                 c->enqueue_ms, (unsigned long long)t.num_adds);
}

// The last kernel's results out of pinned memory: every word arrives as an aligned 8-byte (word, sequence number) pair written by ONE device store and read
// here by ONE load, so a word is taken only with the tag of THIS call (msm_kernels.hpp, store_words8_tagged: why a sequence word behind a fence was not enough).
// false: some pair is not there yet (nothing is half-taken: the caller polls on, or reports an error when the kernel has retired).
bool gather_results(msm_ctx* c, uint32_t nblk, uint32_t seq) {
    const volatile uint64_t* q64 = reinterpret_cast<const volatile uint64_t*>(c->h_qsums);
    const volatile uint64_t* f64 = reinterpret_cast<const volatile uint64_t*>(c->h_flags);
    uint32_t* out = c->qsums.data();
    const size_t npairs = (size_t)nblk * 24;
    for (size_t k = 0; k < npairs; k++) {
        const uint64_t v = q64[k];
        if ((uint32_t)(v >> 32) != seq) return false;
        out[k] = (uint32_t)v;
    }
    for (int k = 0; k < 8; k++) {
        const uint64_t v = f64[k];
        if ((uint32_t)(v >> 32) != seq) return false;
        c->flagw[k] = (uint32_t)v;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return true;
}

// wait for the queued pipeline, finish on the CPU, fill outputs and timings
int32_t finish_sync(msm_ctx* c, const PipeState& ps, size_t n_total, hipStream_t st, uint32_t* out_jac, uint32_t* out_aff, uint8_t* out_inf,
                    hipEvent_t done = nullptr /* recorded after the last kernel when `st` carries other work too */) {
    if (trace_enabled()) c->enqueue_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - c->t_prepare).count();
    if (c->pool && n_total >= 256) c->pool->arm();  // workers wake up while the GPU works
    // Round 6: the host polls the results themselves -- (word, call number) pairs the last kernel's workgroups write into pinned memory (gather_results) -- instead of
    // waiting for the stream: they are there a few microseconds before the kernel has retired and the runtime has woken the waiting thread.
    // Every ~1000 polls the stream is queried, so that a failed launch or a lost device ends the wait with an error instead of a hang.
    // (Stage timing reads events recorded behind the kernel: it waits for the stream as before.)
    const uint32_t nblk = ps.rW * (ps.rkb + 1), seq = c->done_seq;
    if (c->stage_timing || c->knobs.no_poll) {
        if (done) HIPCHK(c, hipEventSynchronize(done));
        else HIPCHK(c, hipStreamSynchronize(st));
        if (!gather_results(c, nblk, seq)) return fail(c, MSM_ERR_HIP, "internal: the bucket reduction finished without publishing its %u bit sums", nblk);
    } else {
        // cheap test first: the LAST pair of every bit sum; then every pair (a pair that is not there yet: keep polling)
        const volatile uint64_t* q64 = reinterpret_cast<const volatile uint64_t*>(c->h_qsums);
        for (uint32_t spins = 1;; spins++) {
            uint32_t i = 0;
            while (i < nblk && (uint32_t)(q64[(size_t)i * 24 + 23] >> 32) == seq) i++;
            if (i == nblk && gather_results(c, nblk, seq)) break;
            if ((spins & 0x3FFu) == 0) {
                const hipError_t q = done ? hipEventQuery(done) : hipStreamQuery(st);
                if (q == hipSuccess) {  // retired: the pairs must be there now
                    if (gather_results(c, nblk, seq)) break;
                    return fail(c, MSM_ERR_HIP, "internal: the bucket reduction finished without publishing its %u bit sums", nblk);
                }
                if (q != hipErrorNotReady) HIPCHK(c, q);
            }
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
        }
#ifdef MSM_HIP_TEST_HOOKS
        if (std::getenv("MSM_HIP_POLL_VERIFY")) {  // diagnosis (tools/race_hunt.py): is what the host accepted what the RETIRED kernel left?
            const std::vector<uint32_t> snap(c->qsums.begin(), c->qsums.begin() + (size_t)nblk * 24);
            uint32_t fl[8];
            std::memcpy(fl, c->flagw, sizeof fl);
            if (done) HIPCHK(c, hipEventSynchronize(done));
            else HIPCHK(c, hipStreamSynchronize(st));
            if (!gather_results(c, nblk, seq)) return fail(c, MSM_ERR_HIP, "internal: POLL_VERIFY: pairs missing after the kernel retired");
            for (uint32_t k = 0; k < nblk * 24; k++)
                if (snap[k] != c->qsums[k]) {
                    std::fprintf(stderr, "[msm_hip] POLL_VERIFY: bit sum %u word %u accepted as %08x, is %08x (seq %u, %u sums)\n", k / 24, k % 24, snap[k], c->qsums[k], seq, nblk);
                    break;
                }
            if (std::memcmp(fl, c->flagw, sizeof fl)) std::fprintf(stderr, "[msm_hip] POLL_VERIFY: flag words accepted early differ\n");
        }
#endif
    }
    HIPCHK(c, hipGetLastError());
    c->flags_clean = true;  // the last kernel zeroed the flag words after copying them out
    auto t_fin0 = std::chrono::steady_clock::now();
    int32_t rc;
    if ((rc = check_flags(c, c->flagw))) return rc;
    hostg1::Jac total = host_finish(c, c->qsums.data(), ps);
    finish_outputs(total, out_jac, out_aff, out_inf, (c->cfg.flags & MSM_FLAG_DETERMINISTIC) != 0);
    auto t_fin1 = std::chrono::steady_clock::now();
    float ms = 0;
    msm_timings_t& tm = c->tm;
    tm = msm_timings_t{};
    tm.decompose_ms = stage_ms(c, EV_CONVERT, EV_DECOMP);
    tm.sort_ms = stage_ms(c, EV_DECOMP, EV_SORT);
    if (c->acc_last_timed) {  // (msm_set_kernel_timing: every n-th launch carries events; 0 otherwise)
        (void)hipEventElapsedTime(&ms, c->ev[EV_ACC0], c->ev[EV_ACC1]);  // of the last chunk, when the MSM was streamed
        c->acc_ms_sum += ms;
        c->acc_launches += 1;
    }
    tm.accumulate_ms = ms;
    tm.plan_ms = stage_ms(c, EV_SORT, EV_PLAN);
    tm.combine_ms = stage_ms(c, EV_ACC1, EV_COMBINE);
    tm.reduce_ms = stage_ms(c, EV_COMBINE, EV_REDUCE);
    tm.batch_layout = c->last_batch_layout;  // (of the last BATCH call: a single call in between does not reset it -- ADVICE r4)
    tm.finish_ms = std::chrono::duration<float, std::milli>(t_fin1 - t_fin0).count();
    tm.num_points = n_total;
    tm.num_adds = (uint64_t)c->flagw[msmk::FLAG_ADDS64] | ((uint64_t)c->flagw[msmk::FLAG_ADDS64 + 1] << 32);
    return MSM_OK;
}

// K1b + K2 + K3 of a whole MSM whose inputs are in HBM, on stream st.  d_bases: INTERNAL-domain records (with a window table: the
// table, record j * n + i = 2^(c*j) P_i; the planner only makes tables whose shared array is sorted in ONE piece --
// msmplan::TABLE_MAX_ENTRIES: cutting the windows into ranges that accumulate INTO the array was built, found bit-exact and 17 % slower
// than no table at 2^22 points, profiles/r3_f4_shared_buckets.txt).
int32_t enqueue_body(msm_ctx* c, const PipeState& ps, const BaseSrc& d_bases, const uint8_t* d_inf, const uint32_t* d_scalars,
                     uint32_t scalars_mont, hipStream_t st, hipEvent_t bases_ready) {
    int32_t rc;
    // (d_bases.phi_fill: the phi records of arkworks-form bases have not been made yet -- the decomposition writes them)
    const bool fuse = d_bases.m256 && d_bases.phi_fill && ps.pl.glv;
    if ((rc = enqueue_decompose(c, ps, d_inf, d_scalars, scalars_mont, st, true, fuse ? d_bases.rec : nullptr, fuse ? d_bases.phi_fill : nullptr))) return rc;
    if ((rc = enqueue_sort(c, ps, st, false))) return rc;
    return enqueue_accumulate(c, ps, d_bases, st, bases_ready, false);
}

// The pipeline proper: everything in HBM, one stream.  d_bases: INTERNAL-domain packed coordinates (or the window table of the
// resident set, table_f > 1: planned with table_c bits per window).
int32_t run_pipeline(msm_ctx* c, const BaseSrc& d_bases, const uint8_t* d_inf, const uint32_t* d_scalars, size_t n,
                     hipStream_t st, uint32_t* out_jac, uint32_t* out_aff, uint8_t* out_inf, uint32_t scalars_mont = 0,
                     hipEvent_t bases_ready = nullptr, uint32_t extra_flags = 0, PipeState* ps_out = nullptr, uint32_t table_c = 0,
                     uint32_t table_f = 1) {
    PipeState ps;
    int32_t rc;
    // From 2^23 points a (window, coarse bin) region of the sort outgrows the fine sort's LDS staging (10 coarse bits at most) and
    // the regions take the oversized-region path: sort 2.5 ms at 2^24 where four sorts of 2^22 take 1.5.  Such an instance is
    // cut into point ranges of 2^22 that accumulate INTO the shared bucket array, like the chunks of a streamed host call:
    // 2^23 11.42 -> 11.0-11.1 ms, 2^24 22.76 -> 21.90 (tools/device_chunk_ab.py; every further cut costs ~0.1 ms: 2^22 in ranges of
    // 2^20 loses 0.5 ms, so smaller instances stay whole).
    const uint32_t dev_chunk_log2 = c->knobs.device_chunk_log2;  // (MSM_HIP_DEVICE_CHUNK_LOG2 at context creation; 0 = never cut)
    const size_t dchunk = dev_chunk_log2 ? (size_t)1 << dev_chunk_log2 : 0;
    if (table_f <= 1 && dchunk && n >= 2 * dchunk && !plan_glv(c, n, extra_flags)) {
        if ((rc = pipe_prepare(c, dchunk, n, extra_flags, st, &ps))) return rc;
        const PipeState ps_first = ps;  // piece lengths: fixed by the first, largest range
        uint32_t nch = 0;
        for (size_t lo = 0; lo < n; lo += dchunk, nch++) {
            const size_t cnt = std::min(dchunk, n - lo);
            if ((rc = pipe_prepare(c, cnt, n, extra_flags, st, &ps, 0, 1, &ps_first))) return rc;
            if ((rc = enqueue_digits_sort(c, ps, d_inf ? d_inf + lo : nullptr, d_scalars + lo * 8, scalars_mont, st, lo == 0))) return rc;
            if ((rc = enqueue_accumulate(c, ps, d_bases.shifted(lo), st, lo == 0 ? bases_ready : nullptr, lo > 0, true))) return rc;
        }
        if ((rc = enqueue_reduce(c, ps, st, c->h_qsums, c->h_flags))) return rc;
        if (ps_out) *ps_out = ps;
        if ((rc = finish_sync(c, ps, n, st, out_jac, out_aff, out_inf))) return rc;
        c->tm.stream_chunks = nch;
        return MSM_OK;
    }
    if ((rc = pipe_prepare(c, n, 0, extra_flags, st, &ps, table_c, table_f))) return rc;
    if ((rc = enqueue_body(c, ps, d_bases, d_inf, d_scalars, scalars_mont, st, bases_ready))) return rc;
    if ((rc = enqueue_reduce(c, ps, st, c->h_qsums, c->h_flags))) return rc;
    if (ps_out) *ps_out = ps;
    return finish_sync(c, ps, n, st, out_jac, out_aff, out_inf);
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

// ---- host inputs ------------------------------------------------------------------------------------------------------------
// What the two host-pointer entries hand over: packed x||y words (standard or Montgomery form) with an optional byte mask of
// points at infinity, or an array of arkworks G1Affine structs read through (stride, offsets); scalars in standard form or as
// arkworks Fr Montgomery words.
// KIND_ARKFAST: a struct array taken OPTIMISTICALLY as free of points at infinity (k_ark_repack); a flag that is set makes the call return ARK_RETRY_SLOW
// internally and run_host_input repeats it as KIND_ARK
enum : uint32_t { KIND_STD = 0, KIND_MONT = 1, KIND_ARK = 2, KIND_ARKFAST = 3 };
struct HostInput {
    const uint8_t* bases = nullptr;
    size_t stride = 64;  // bytes per base record
    uint32_t kind = KIND_STD;
    uint32_t x_off = 0, y_off = 0, inf_off = 0;
    bool has_inf_field = false;         // KIND_ARK: the struct has an `infinity` byte
    const uint8_t* inf_mask = nullptr;  // packed kinds: nullable
    const uint32_t* scalars = nullptr;
    uint32_t scalars_mont = 0;
    bool carries_inf() const { return kind == KIND_ARK || inf_mask != nullptr; }
};

// host -> device copy on stream cs (hipMemcpyAsync: from pageable memory the runtime stages the data itself -- ~40 GB/s -- and the
// call returns when the source has been consumed; from pinned memory it is fully asynchronous at the link rate, 56 GB/s).
// Tried and dropped in round 2 (profiles/NOTES_r2.md): a pinned staging ring filled by 8 host threads (slower than the runtime's
// own staging) and kernels reading pinned caller memory in place (their wavefronts wait on PCIe while holding the slots
// k_accumulate needs: 4.2 ms against 2.7 ms at 2^20).
int32_t h2d(msm_ctx* c, void* dst, const void* src, size_t bytes, hipStream_t cs) {
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, cs));
    return MSM_OK;
}
// K1's coordinate half for cnt raw base records that sit in HBM as the caller sent them, on stream st; returns what the accumulation gathers
// from.  arkworks-form words (R = 2^256 Montgomery): the records stay where they are -- nothing to do for an unsplit plan, the phi records of a
// split one go to d_out (k_phi_records, round 5).  Standard-form words and struct arrays: internal-domain records in d_out (+ infinity bytes
// for the struct form).
BaseSrc launch_convert(const HostInput& in, const void* d_raw, size_t cnt, uint32_t* d_out, uint8_t* d_inf, bool glv, hipStream_t st,
                       uint32_t* err_flags = nullptr) {
    if (in.kind == KIND_ARKFAST) {  // repacked to x || y words in d_out (+ the phi records behind them): gathered like the packed-words call's
        msmk::k_ark_repack<<<grid1(cnt, 256), 256, 0, st>>>((const uint8_t*)d_raw, (uint64_t)in.stride, in.x_off, in.y_off,
                                                          in.has_inf_field ? in.inf_off : 0u, in.has_inf_field ? 1u : 0u, (uint32_t)cnt, d_out, err_flags,
                                                          glv ? 1u : 0u);
        BaseSrc src;
        src.m256 = true;
        src.rec = d_out;
        if (glv) src.phi = d_out + cnt * 16, src.nsplit = (uint32_t)cnt;
        return src;
    }
    if (in.kind == KIND_ARK) {
        msmk::k_import_ark<<<grid1(2 * cnt, 256), 256, 0, st>>>((const uint8_t*)d_raw, (uint64_t)in.stride, in.x_off, in.y_off,
                                                             in.has_inf_field ? in.inf_off : 0u, in.has_inf_field ? 1u : 0u, (uint32_t)cnt,
                                                             d_out, d_inf, glv ? 1u : 0u);
        return BaseSrc(d_out);
    }
    if (in.kind == KIND_MONT) {
        BaseSrc src;
        src.m256 = true;
        src.rec = (const uint32_t*)d_raw;
        if (glv) {
            msmk::k_phi_records<<<grid1(cnt, 256), 256, 0, st>>>((const uint32_t*)d_raw, d_out, (uint32_t)cnt);
            src.phi = d_out;
            src.nsplit = (uint32_t)cnt;
        }
        return src;
    }
    msmk::k_convert_bases<<<grid1(2 * cnt, 256), 256, 0, st>>>((const uint32_t*)d_raw, d_out, (uint32_t)cnt, 0u, glv ? 1u : 0u);  // standard form
    return BaseSrc(d_out);
}

// scalars (+ the infinity mask of the packed forms) of points [lo, lo+cnt) -> d_scalars / d_inf on stream s
int32_t feed_scalars(msm_ctx* c, const HostInput& in, size_t lo, size_t cnt, void* d_scalars, void* d_inf, hipStream_t s) {
    int32_t rc;
    if ((rc = h2d(c, d_scalars, in.scalars + lo * 8, cnt * 32, s))) return rc;
    if (in.inf_mask && (rc = h2d(c, d_inf, in.inf_mask + lo, cnt, s))) return rc;
    return MSM_OK;
}
// base records of points [lo, lo+cnt) -> raw staging d_raw -> internal-domain records d_ibases (+ infinity bytes d_inf for the
// struct form), all on stream s
int32_t feed_bases(msm_ctx* c, const HostInput& in, size_t lo, size_t cnt, void* d_raw, uint32_t* d_ibases, uint8_t* d_inf, bool glv,
                   hipStream_t s, BaseSrc* src) {
    int32_t rc = h2d(c, d_raw, in.bases + lo * in.stride, cnt * in.stride, s);
    if (rc) return rc;
    *src = launch_convert(in, d_raw, cnt, d_ibases, d_inf, glv, s, (uint32_t*)c->flags.p);
    return MSM_OK;
}

// Single shot: scalars (and the infinity mask) go first on the compute stream and are sorted while the bases -- two thirds of
// the bytes, only needed by k_accumulate -- still travel and are converted on the copy stream.  (The struct form carries the
// infinity flags inside the base records, which k_decompose needs: there the two transfers merely share the link.)
int32_t run_single(msm_ctx* c, const HostInput& in, size_t n, uint32_t* out_jac, uint32_t* out_aff, uint8_t* out_inf) {
    int32_t rc;
    hipStream_t st = c->stream, cs = c->copy_stream;
    const bool glv = plan_glv(c, n);
    PipeState ps;
    if ((rc = ensure(c, c->scalars, n * 32))) return rc;
    if ((rc = ensure(c, c->bases, n * in.stride))) return rc;
    if ((rc = ensure(c, c->ibases, (glv ? 2 : 1) * n * 64))) return rc;
    if (in.carries_inf() && (rc = ensure(c, c->inf, n))) return rc;
    if ((rc = pipe_prepare(c, n, 0, 0, st, &ps))) return rc;
    const bool overlap = n >= 4096 && !c->stage_timing;  // below that two cross-stream waits cost more than the copy
    hipStream_t bs = overlap ? cs : st;                  // stream the bases travel and are converted on
    if (c->stage_timing) HIPCHK(c, hipEventRecord(c->ev[EV_START], st));
    if (overlap && in.kind == KIND_ARKFAST) {
        // k_ark_repack raises its error bit in the flag words from the COPY stream.  The previous call returned when its results were in host memory, possibly
        // before its last kernel -- which zeroes those words -- had retired and written its lines back: the copy stream waits for everything queued on the
        // compute stream so far (nothing of this call yet), so that the bit cannot be lost under that write-back.
        HIPCHK(c, hipEventRecord(c->ev_free[0], st));  // (an event of the streamed path: unused by a single-shot call)
        HIPCHK(c, hipStreamWaitEvent(cs, c->ev_free[0], 0));
    }
    const uint8_t* d_inf = in.carries_inf() ? (const uint8_t*)c->inf.p : nullptr;
    uint32_t* ib = (uint32_t*)c->ibases.p;
    BaseSrc src;
    {
        Range r_("msm:h2d");
        if (in.kind == KIND_ARK) {
            if ((rc = feed_bases(c, in, 0, n, c->bases.p, ib, (uint8_t*)c->inf.p, glv, bs, &src))) return rc;
            if (overlap) HIPCHK(c, hipEventRecord(c->ev_bases, bs));
            if ((rc = feed_scalars(c, in, 0, n, c->scalars.p, nullptr, st))) return rc;
            if (c->stage_timing) HIPCHK(c, hipEventRecord(c->ev[EV_H2D], st));
            if (overlap) HIPCHK(c, hipStreamWaitEvent(st, c->ev_bases, 0));  // k_decompose reads the infinity bytes
            if ((rc = enqueue_digits_sort(c, ps, d_inf, (const uint32_t*)c->scalars.p, in.scalars_mont, st, true))) return rc;
            if ((rc = enqueue_accumulate(c, ps, src, st, nullptr, false))) return rc;
        } else {
            if ((rc = feed_scalars(c, in, 0, n, c->scalars.p, c->inf.p, st))) return rc;
            if (overlap) {
                // queue the sort BEFORE the bases are touched: a copy from pageable memory blocks the host, the GPU sorts meanwhile
                if ((rc = enqueue_digits_sort(c, ps, d_inf, (const uint32_t*)c->scalars.p, in.scalars_mont, st, true))) return rc;
                if ((rc = feed_bases(c, in, 0, n, c->bases.p, ib, nullptr, glv, bs, &src))) return rc;
                HIPCHK(c, hipEventRecord(c->ev_bases, bs));
                if ((rc = enqueue_accumulate(c, ps, src, st, c->ev_bases, false))) return rc;
            } else {
                if ((rc = feed_bases(c, in, 0, n, c->bases.p, ib, nullptr, glv, st, &src))) return rc;
                if (c->stage_timing) HIPCHK(c, hipEventRecord(c->ev[EV_H2D], st));
                if ((rc = enqueue_digits_sort(c, ps, d_inf, (const uint32_t*)c->scalars.p, in.scalars_mont, st, true))) return rc;
                if ((rc = enqueue_accumulate(c, ps, src, st, nullptr, false))) return rc;
            }
        }
    }
    if ((rc = enqueue_reduce(c, ps, st, c->h_qsums, c->h_flags))) return rc;
    if ((rc = finish_sync(c, ps, n, st, out_jac, out_aff, out_inf))) return rc;
    c->tm.h2d_ms = stage_ms(c, EV_START, EV_H2D);
    c->tm.convert_ms = stage_ms(c, EV_H2D, EV_CONVERT);
    trace_line(c, "host single-shot", ps);
    return MSM_OK;
}

// BASELINE config 5, and every host-pointer call from 2^19 points on: the instance does not have to be resident.  The point
// range is cut into chunks; chunk j+1 travels host->HBM and is converted on the copy stream while chunk j is sorted and
// accumulated on the compute stream.  MSM is linear in the points, so every chunk adds into the SAME bucket array
// (k_accumulate<INTO>, one plan for the whole instance): ONE bucket reduction, ONE host finish and W*(kb+1) bit sums back,
// however many chunks.  Raw inputs live in STREAM_SLOTS buffers; the coordinate pass (round 5: none for arkworks words but a split plan's phi records -- the
// accumulation gathers from the slot itself) runs on the compute stream right before the chunk's
// accumulation (the copy stream carries copies only).  Transfer (96 B per point at 52-56 GB/s from pinned, ~40 GB/s from pageable memory:
// 0.48 ms per 2^18 points) and per-chunk work (sort 0.09 + conversion 0.02 + accumulation INTO the buckets 0.29-0.31 + fold 0.02 = 0.43 ms)
// are about level: the call costs the first transfer, then the slower of the two per chunk, then the last chunk's work, the reduction
// and the host finish (2^20: 2.43-2.53 ms pinned against 1.45 resident; timeline in profiles/r4_host_call_timeline.txt).
int32_t run_streamed(msm_ctx* c, const HostInput& in, size_t n, const std::vector<size_t>& sizes, uint32_t* out_jac,
                     uint32_t* out_aff, uint8_t* out_inf) {
    int32_t rc;
    const bool glv = plan_glv(c, n);  // of the WHOLE instance: all chunks share one bucket array
    const size_t chunk = *std::max_element(sizes.begin(), sizes.end());
    for (int s = 0; s < STREAM_SLOTS; s++) {
        if ((rc = ensure(c, c->sbases[s], chunk * in.stride))) return rc;
        if ((rc = ensure(c, c->sscalars[s], chunk * 32))) return rc;
        if (in.carries_inf() && (rc = ensure(c, c->sinf[s], chunk))) return rc;
    }
    if ((rc = ensure(c, c->sibases, (glv ? 2 : 1) * chunk * 64))) return rc;
    // ONE copy stream for scalars and bases: on two streams (two SDMA queues) the transfers of a chunk share the link at a LOWER combined
    // rate (24 MB in 0.61 ms instead of 0.47) -- the 16-20 us of command turnaround between two transfers of one queue cost less
    // (round 4, profiles/r4_host_call_timeline.txt)
    hipStream_t st = c->stream, cs = c->copy_stream;
    PipeState ps;
    if ((rc = pipe_prepare(c, chunk, n, 0, st, &ps))) return rc;  // workspace for the largest chunk before anything is in flight
    const PipeState ps_first = ps;                                 // piece lengths: fixed by the largest chunk
    size_t lo = 0;
    for (size_t j = 0; j < sizes.size(); j++) {
        const int s = (int)(j % STREAM_SLOTS);
        const size_t cnt = sizes[j];
        uint8_t* d_inf = in.carries_inf() ? (uint8_t*)c->sinf[s].p : nullptr;
        {
            Range r_("msm:h2d chunk");
            if (j >= (size_t)STREAM_SLOTS) HIPCHK(c, hipStreamWaitEvent(cs, c->ev_free[s], 0));  // the pipeline that read this slot is done
            if ((rc = feed_scalars(c, in, lo, cnt, c->sscalars[s].p, d_inf, cs))) return rc;
            HIPCHK(c, hipEventRecord(c->ev_scal[s], cs));
            // the copy stream carries COPIES only: with the conversion kernel between two chunks' transfers the link idled ~40 us per
            // chunk -- and the link is what a host-pointer call waits for
            if ((rc = h2d(c, c->sbases[s].p, in.bases + lo * in.stride, cnt * in.stride, cs))) return rc;
            HIPCHK(c, hipEventRecord(c->ev_copied[s], cs));
        }
        // the sort only needs the scalars (a third of the chunk's bytes): it starts while the bases still travel.  (The struct form
        // carries the infinity flags inside the base records, which k_decompose reads: there the sort waits for the whole chunk.)
        const bool early_sort = in.kind != KIND_ARK;
        if ((rc = pipe_prepare(c, cnt, n, 0, st, &ps, 0, 1, &ps_first))) return rc;
        HIPCHK(c, hipStreamWaitEvent(st, c->ev_scal[s], 0));
        if (early_sort) {
            if ((rc = enqueue_digits_sort(c, ps, d_inf, (const uint32_t*)c->sscalars[s].p, in.scalars_mont, st, j == 0))) return rc;
        }
        HIPCHK(c, hipStreamWaitEvent(st, c->ev_copied[s], 0));
        const BaseSrc src = launch_convert(in, c->sbases[s].p, cnt, (uint32_t*)c->sibases.p, d_inf, glv, st, (uint32_t*)c->flags.p);  // (arkworks words: gathered from the slot itself)
        if (!early_sort && (rc = enqueue_digits_sort(c, ps, d_inf, (const uint32_t*)c->sscalars[s].p, in.scalars_mont, st, j == 0))) return rc;
        if ((rc = enqueue_accumulate(c, ps, src, st, nullptr, j > 0, true))) return rc;
        HIPCHK(c, hipEventRecord(c->ev_free[s], st));
        lo += cnt;
    }
    if ((rc = enqueue_reduce(c, ps, st, c->h_qsums, c->h_flags))) return rc;
    if ((rc = finish_sync(c, ps, n, st, out_jac, out_aff, out_inf))) return rc;
    c->tm.stream_chunks = (uint32_t)sizes.size();
    trace_line(c, "host streamed", ps);
    return MSM_OK;
}

// chunk schedule of the streamed path for n points; empty = single shot
std::vector<size_t> stream_schedule(const msm_ctx* c, size_t n) {
    std::vector<size_t> sizes;
    if (c->cfg.stream_chunk_log2) {  // explicit: uniform chunks of 2^k points, ragged last one
        const size_t chunk = (size_t)1 << c->cfg.stream_chunk_log2;
        if (n < 2 * chunk) return sizes;
        for (size_t lo = 0; lo < n; lo += chunk) sizes.push_back(std::min(chunk, n - lo));
        return sizes;
    }
    // automatic.  A chunk costs its transfer (96 B per point: 0.43 ms per 2^18 points at the link's 56 GB/s, ~0.6 ms from pageable
    // memory, which the runtime stages at ~40 GB/s) or its sort + accumulation + combine (~0.47 ms per 2^18 points, of which
    // ~0.2 ms do not shrink with the chunk: sort launches, k_combine over all buckets, bucket read-modify-write), whichever is
    // longer.  Uniform chunks of 2^18 points (2^19 / 2^20 for large instances): shorter chunks at the end, meant to leave less work
    // after the last byte, cost more in fixed per-chunk work than they hide (measured: 2^20 in 4 chunks 2.89 ms, 3 x 2^18 +
    // 2^17 + 2 x 2^16: 3.48 ms -- profiles/NOTES_r2.md; round 3, with the sort already overlapped: the last chunk halved once / twice /
    // three times costs +0.12 / +0.24 / +0.41 ms at 2^20, profiles/r3_stream_tail_split.txt; round 4, copy stream without the conversion
    // kernels and three slots: halved once / twice / three times +0.08 / +0.2 / +0.3 ms -- a 2^17-point chunk still costs 0.30 ms of
    // sort + accumulation INTO 2^18 buckets, two of them 0.17 ms more than the chunk they replace, and the compute stream has no slack
    // left to hide it).  A remainder below half a chunk joins the last chunk.
    if (!c->knobs.stream_schedule.empty()) {  // (hooks build: an explicit schedule, taken when it covers exactly n points)
        size_t tot = 0;
        for (uint32_t lg2 : c->knobs.stream_schedule) tot += (size_t)1 << lg2;
        if (tot == n) {
            for (uint32_t lg2 : c->knobs.stream_schedule) sizes.push_back((size_t)1 << lg2);
            return sizes;
        }
    }
    const uint32_t min_log2 = c->knobs.stream_min_log2;
    uint32_t lg = n < ((size_t)1 << 21) ? 18u : n < ((size_t)1 << 23) ? 19u : 20u;
    if (c->knobs.stream_chunk_log2) lg = c->knobs.stream_chunk_log2;
    const size_t chunk = (size_t)1 << lg;
    if (n < ((size_t)1 << min_log2) || n < 2 * chunk) return sizes;
    size_t left = n;
    while (left >= chunk + chunk / 2) {
        sizes.push_back(chunk);
        left -= chunk;
    }
    if (left > chunk) {  // (chunk, 1.5 chunk): two halves
        sizes.push_back((left / 2 + 63) & ~(size_t)63);
        left -= sizes.back();
    }
    if (left) sizes.push_back(left);
    return sizes;
}

// Pageable caller memory, pinned IN PLACE for the duration of the call (round 6).  hipMemcpyAsync from pageable memory is staged by the runtime at ~40 GB/s and blocks
// the calling thread; from registered memory it is a DMA at the link's 53-55 GB/s -- and on this platform hipHostRegister / hipHostUnregister of 96 MB take ~3 us
// each (tools/host_register_probe.py): 2^20 points from plain heap memory 2.55 -> 2.42 ms, what torch-pinned memory gives.  Memory the runtime already knows
// (pinned by the caller, or registered by an msm_multi handle for all its ranks) is left alone; a range that cannot be registered (read-only mappings) simply
// travels as before.  Unregistered when the call has consumed it (every copy is ordered in front of the kernel whose results the call waited for).
// Registrations are shared process-wide: two contexts handed the SAME array at the same time (two threads of a prover, the ranks of an msm_multi handle) must not
// unregister it under each other's copies -- a range this library registered is reference-counted, and a pointer inside such a range takes a reference on it.
struct HostPinRegistry {
    struct Entry {
        uintptr_t lo, hi;
        int refs;
    };
    std::mutex mu;
    std::vector<Entry> live;
    static HostPinRegistry& get() {
        static HostPinRegistry r;
        return r;
    }
};
struct HostPin {
    uintptr_t key = 0;  // start of the registered range this pin holds a reference on (0: none)
    void pin(const void* ptr, size_t bytes, bool all_devices = false) {
        if (!ptr || bytes < ((size_t)1 << 20) || key) return;
        HostPinRegistry& R = HostPinRegistry::get();
        std::lock_guard<std::mutex> lk(R.mu);
        const uintptr_t a = (uintptr_t)ptr;
        for (auto& e : R.live)
            if (a >= e.lo && a < e.hi) {  // ours already (possibly a larger range: the whole arrays of an msm_multi call)
                e.refs++;
                key = e.lo;
                return;
            }
        hipPointerAttribute_t at{};
        const hipError_t q = hipPointerGetAttributes(&at, ptr);
        (void)hipGetLastError();
        if (q == hipSuccess && at.type != hipMemoryTypeUnregistered) return;  // pinned / registered by the caller: left alone
        if (hipHostRegister(const_cast<void*>(ptr), bytes, all_devices ? hipHostRegisterPortable : hipHostRegisterDefault) == hipSuccess) {
            R.live.push_back({a, a + bytes, 1});
            key = a;
        } else {
            (void)hipGetLastError();  // (read-only mappings, exotic memory: the data travels as pageable memory does)
        }
    }
    ~HostPin() {
        if (!key) return;
        HostPinRegistry& R = HostPinRegistry::get();
        std::lock_guard<std::mutex> lk(R.mu);
        for (size_t i = 0; i < R.live.size(); i++)
            if (R.live[i].lo == key) {
                if (--R.live[i].refs == 0) {
                    (void)hipHostUnregister((void*)key);
                    (void)hipGetLastError();
                    R.live.erase(R.live.begin() + (long)i);
                }
                break;
            }
    }
    HostPin() = default;
    HostPin(const HostPin&) = delete;
    HostPin& operator=(const HostPin&) = delete;
};

int32_t run_host_input(msm_ctx* c, const HostInput& in0, size_t n, uint32_t* out_jac, uint32_t* out_aff, uint8_t* out_inf) {
    auto t0 = std::chrono::steady_clock::now();
    HostPin pin_b, pin_s, pin_i;
    if (!c->no_host_pin) {
        pin_b.pin(in0.bases, n * in0.stride);
        pin_s.pin(in0.scalars, n * 32);
        pin_i.pin(in0.inf_mask, n);
    }
    const std::vector<size_t> sizes = stream_schedule(c, n);
    // Struct arrays (round 6): first as if no point were at infinity -- the words are repacked and the call runs like the packed-words call, its sort
    // overlapped with the transfer of the bases; a set flag (error bit 16, seen when the call has finished) repeats the call with the flags as a mask.
    HostInput in = in0;
    if (in.kind == KIND_ARK && !c->knobs.ark_slow) in.kind = KIND_ARKFAST;
    int32_t rc = sizes.empty() ? run_single(c, in, n, out_jac, out_aff, out_inf) : run_streamed(c, in, n, sizes, out_jac, out_aff, out_inf);
    if (rc == ARK_RETRY_SLOW) {
        in.kind = KIND_ARK;
        rc = sizes.empty() ? run_single(c, in, n, out_jac, out_aff, out_inf) : run_streamed(c, in, n, sizes, out_jac, out_aff, out_inf);
    }
    if (rc) return rc;
    c->tm.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return MSM_OK;
}

// The plan of resident calls on a set of n bases under this context: GLV split or not, and the window table (table_factor > 1 with
// MSM_FLAG_WINDOW_TABLE).  Decided ONCE per upload; the calls that follow use what the upload stored.
int32_t resident_plan(const msm_ctx* c, size_t n, msm_plan_t* pl) {
    return msmplan::make_table_plan(n, c->cfg.window_bits, c->cfg.flags, pl, c->knobs.glv_max, c->knobs.table);
}
// levels 1 .. f-1 of the window table behind level 0 (the converted bases, nv records) in c->rbases, on the context's stream
int32_t build_table_locked(msm_ctx* c, const msm_plan_t& pl) {
    c->table_c = 0, c->table_f = 1;
    if (pl.table_factor <= 1) return MSM_OK;
    Range r_("msm:window_table");
    const size_t nv = (size_t)pl.virtual_points;
    uint32_t* t = (uint32_t*)c->rbases.p;
    const uint32_t top_shift = table_top_shift(pl, pl.table_factor);
    for (uint32_t j = 1; j < pl.table_factor; j++)  // (the top level of a full table: top_shift doublings fewer, see table_top_shift)
        msmk::k_table_next<<<grid1((nv + 1) / 2, 256), 256, 0, c->stream>>>(t + (size_t)(j - 1) * nv * 16, t + (size_t)j * nv * 16, (uint32_t)nv,
                                                                 pl.window_bits - (j + 1 == pl.table_factor ? top_shift : 0u));
    c->table_c = pl.window_bits, c->table_f = pl.table_factor;
    return MSM_OK;
}
// raw caller coordinates (host) -> c->bases (staging) -> the RESIDENT set c->rbases (internal domain; + the window table)
int32_t upload_resident_locked(msm_ctx* c, const uint32_t* bases_xy, uint32_t form, const uint8_t* inf_mask, size_t n) {
    if (form != MSM_FORM_STD && form != MSM_FORM_MONT) return fail(c, MSM_ERR_BAD_ARG, "unknown base_form %u", form);
    int32_t rc;
    msm_plan_t pl;
    if ((rc = resident_plan(c, n, &pl))) return fail(c, rc, "bad window_bits/flags (%u, 0x%x)", c->cfg.window_bits, c->cfg.flags);
    const bool glv = pl.glv != 0;
    if ((rc = ensure(c, c->bases, n * 64))) return rc;
    if ((rc = ensure(c, c->rbases, (size_t)pl.table_factor * (glv ? 2 : 1) * n * 64))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->bases.p, bases_xy, n * 64, hipMemcpyHostToDevice, c->stream));
    if (inf_mask) {
        if ((rc = ensure(c, c->rinf, n))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->rinf.p, inf_mask, n, hipMemcpyHostToDevice, c->stream));
    }
    msmk::k_convert_bases<<<grid1(2 * n, 256), 256, 0, c->stream>>>((const uint32_t*)c->bases.p, (uint32_t*)c->rbases.p, (uint32_t)n,
                                                                  form == MSM_FORM_MONT ? 1u : 0u, glv ? 1u : 0u);
    if ((rc = build_table_locked(c, pl))) return rc;
    c->resident_glv = glv;
    return MSM_OK;
}

}  // namespace

extern "C" {

uint32_t msm_abi_version(void) { return MSM_HIP_ABI_VERSION; }

const char* msm_last_error(const msm_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

// main_priority 1: the second pipeline of msm_bn254_g1_resident_batch.  Its main stream comes from the LOW-priority pool, so it
// never shares a hardware queue with the first pipeline's main stream (equal-priority streams are dealt round-robin onto a few
// HSA queues; two on one queue serialise and the batch gains nothing -- measured, profiles/NOTES_r2.md) and its kernels fill the
// gaps of the first pipeline instead of competing with it.
static int32_t ctx_create_impl(const msm_config_t* cfg, msm_ctx** out, int main_priority /* 0 default pool, 1 low, 2 high */,
                               const Knobs* inherit = nullptr /* second pipeline of a batch: the owner's knobs */) {
    if (!out) return fail(nullptr, MSM_ERR_BAD_ARG, "out == NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, MSM_ERR_NO_DEVICE, "no HIP device visible: this engine has no CPU fallback");
    msm_config_t c0{};
    c0.device = -1;
    if (cfg) c0 = *cfg;
    const Knobs knobs = inherit ? *inherit : Knobs::from_env();
    msm_plan_t probe;
    if (make_plan(1, c0.window_bits, c0.flags, &probe, knobs.glv_max) != MSM_OK)
        return fail(nullptr, MSM_ERR_BAD_ARG, "bad window_bits/flags (%u, 0x%x)", c0.window_bits, c0.flags);
    if (c0.stream_chunk_log2 && (c0.stream_chunk_log2 < 8 || c0.stream_chunk_log2 > 28))
        return fail(nullptr, MSM_ERR_BAD_ARG, "stream_chunk_log2 = %u out of range [8, 28]", c0.stream_chunk_log2);
    if (c0.batch_layout > MSM_BATCH_LAYOUT_TWO_STREAMS) return fail(nullptr, MSM_ERR_BAD_ARG, "unknown batch_layout %u", c0.batch_layout);
    if (c0.host_threads > 64) return fail(nullptr, MSM_ERR_BAD_ARG, "host_threads = %u out of range [0, 64]", c0.host_threads);
    int dev = c0.device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return fail(nullptr, MSM_ERR_NO_DEVICE, "hipGetDevice failed");
    if (dev >= ndev) return fail(nullptr, MSM_ERR_NO_DEVICE, "device %d out of range (%d visible)", dev, ndev);
    msm_ctx* c = new (std::nothrow) msm_ctx();
    if (!c) return fail(nullptr, MSM_ERR_OOM, "host allocation failed");
    c->device = dev;
    c->cfg = c0;
    c->knobs = knobs;
    c->stage_timing = trace_enabled();
    DeviceGuard g(dev);
    int least = 0, greatest = 0, cus = 0;
    if (g.ok && hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
    if (g.ok && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) c->num_cus = (uint32_t)cus;
    hipError_t e = !g.ok ? hipErrorInvalidDevice
                   : (main_priority && least != greatest) ? hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, main_priority == 1 ? least : greatest)
                                                              : hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    for (int i = 0; i < EV_COUNT && e == hipSuccess; i++) e = hipEventCreate(&c->ev[i]);
    if (e == hipSuccess) {
        // The copy stream must own a HARDWARE queue of its own.  Streams of equal priority are spread over a small pool of HSA
        // queues shared with every other stream of the process (PyTorch's included); when both of this context's streams landed on
        // one queue, the barrier packets of their cross-stream event waits executed in queue order and every chunk upload waited
        // for the PREVIOUS chunk's kernels (rocprofv3 kernel trace: one Queue_Id, profiles/NOTES_r2.md).  A stream of another
        // priority comes from another pool.
        e = least != greatest ? hipStreamCreateWithPriority(&c->copy_stream, hipStreamNonBlocking, greatest)
                              : hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_bases, hipEventDisableTiming);
    for (int i = 0; i < STREAM_SLOTS && e == hipSuccess; i++) {
        e = hipEventCreateWithFlags(&c->ev_copied[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_free[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_scal[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_qsums, MAX_QSUM_POINTS * 192, hipHostMallocDefault);  // 24 (word, seq) pairs per bit sum
    if (e == hipSuccess) std::memset(c->h_qsums, 0, MAX_QSUM_POINTS * 192);
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_flags, 64, hipHostMallocDefault);  // 8 pairs
    if (e == hipSuccess) std::memset(c->h_flags, 0, 64);
    if (e == hipSuccess) {
        try {
            c->qsums.assign(MAX_QSUM_POINTS * 24, 0u);
        } catch (const std::bad_alloc&) {  // (no exception crosses the C ABI)
            e = hipErrorOutOfMemory;
        }
    }
    if (e == hipSuccess) e = hipMalloc(&c->clk.p, 32);
    if (e == hipSuccess) {
        c->clk.cap = 32;
        e = hipMemset(c->clk.p, 0, 32);
    }
    if (e == hipSuccess) {
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0) c->wall_clock_khz = (uint32_t)khz;
    }
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
        make_plan(c0.max_points, c0.window_bits, c0.flags, &pl, knobs.glv_max);
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
        if (c0.host_threads) want = (int)c0.host_threads - 1;        // msm_config_t.host_threads: the caller included
        if (knobs.host_threads >= 0) want = knobs.host_threads - 1;  // MSM_HIP_HOST_THREADS (hooks build)
        if (want >= 1) c->pool = new (std::nothrow) HostPool(want);
    }
    *out = c;
    return MSM_OK;
}

int32_t msm_ctx_create(const msm_config_t* cfg, msm_ctx** out) { return ctx_create_impl(cfg, out, 0); }

void msm_ctx_destroy(msm_ctx* c) {
    if (!c) return;
    delete c->pool;
    c->pool = nullptr;
    delete c->batch_pool;
    c->batch_pool = nullptr;
    if (c->lane1) msm_ctx_destroy(c->lane1);
    c->lane1 = nullptr;
    {
        DeviceGuard g(c->device);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
        DevBuf* bufs[] = {&c->bases,   &c->inf,       &c->scalars, &c->digits,  &c->ranks,  &c->sorted, &c->hist,
                          &c->offsets, &c->blocksums, &c->buckets, &c->rc,      &c->flags,  &c->pow2,
                          &c->sorttmp, &c->tilecounts, &c->ibases, &c->longlist, &c->longdone, &c->midlist, &c->ccounts, &c->cregion, &c->bigslot, &c->big,
                          &c->rbases,  &c->rinf,      &c->clk,     &c->phist,   &c->pcursor, &c->pbase,  &c->plist,  &c->partials};
        for (DevBuf* b : bufs) release(*b);
        if (c->h_qsums) (void)hipHostFree(c->h_qsums);
        if (c->h_flags) (void)hipHostFree(c->h_flags);
        release(c->sibases);
        for (int i = 0; i < STREAM_SLOTS; i++) {
            release(c->sbases[i]);
            release(c->sscalars[i]);
            release(c->sinf[i]);
            if (c->ev_copied[i]) (void)hipEventDestroy(c->ev_copied[i]);
            if (c->ev_free[i]) (void)hipEventDestroy(c->ev_free[i]);
            if (c->ev_scal[i]) (void)hipEventDestroy(c->ev_scal[i]);
        }
        if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
        if (c->ev_bases) (void)hipEventDestroy(c->ev_bases);
        if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
        if (c->ev_body) (void)hipEventDestroy(c->ev_body);
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
    if (base_form != MSM_FORM_STD && base_form != MSM_FORM_MONT) return fail(c, MSM_ERR_BAD_ARG, "unknown base_form %u", base_form);
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    Range r_("msm_bn254_g1");
    HostInput in;
    in.bases = (const uint8_t*)bases_xy;
    in.stride = 64;
    in.kind = base_form == MSM_FORM_MONT ? KIND_MONT : KIND_STD;
    in.inf_mask = inf_mask;
    in.scalars = scalars;
    return run_host_input(c, in, n, out_jac, out_aff, out_inf);
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
    Range r_("msm_bn254_g1_arkworks");
    HostInput in;
    in.bases = (const uint8_t*)bases;
    in.stride = stride;
    in.kind = KIND_ARK;
    in.x_off = (uint32_t)x_off, in.y_off = (uint32_t)y_off, in.inf_off = has_inf ? (uint32_t)inf_off : 0u;
    in.has_inf_field = has_inf;
    in.scalars = scalars_mont;
    in.scalars_mont = 1u;
    return run_host_input(c, in, n, out_jac, out_aff, out_inf);
}

int32_t msm_bn254_g1_upload_bases(msm_ctx* c, const uint32_t* bases_xy, uint32_t base_form, const uint8_t* inf_mask,
                                  size_t n) {
    int32_t rc = check_common(c, bases_xy, bases_xy, n);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    c->resident_n = 0;
    c->tuned_layout[0] = c->tuned_layout[1] = 0;  // a measured batch layout belongs to the base set it was measured on
    HostPin pin_b;
    pin_b.pin(bases_xy, n * 64);
    if ((rc = upload_resident_locked(c, bases_xy, base_form, inf_mask, n))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    c->resident_n = n;
    c->resident_has_inf = inf_mask != nullptr;
    return MSM_OK;
}

// row f3: n x 32-byte arkworks compressed images (host) -> d_out (+ d_inf); out_ark picks the word domain written
static int32_t decompress_locked(msm_ctx* c, const uint8_t* compressed, size_t n, uint32_t out_ark, bool glv, uint32_t* d_out,
                                 uint8_t* d_inf, int64_t* first_invalid) {
    int32_t rc;
    if (first_invalid) *first_invalid = -1;
    if ((rc = ensure(c, c->bases, n * 32 + 16))) return rc;
    uint32_t* d_bad = (uint32_t*)((uint8_t*)c->bases.p + n * 32);  // lowest failing index, kept behind the images
    const uint32_t none = 0xFFFFFFFFu;
    HIPCHK(c, hipMemcpyAsync(c->bases.p, compressed, n * 32, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_bad, &none, 4, hipMemcpyHostToDevice, c->stream));
    msmk::k_decompress<<<grid1(n, 256), 256, 0, c->stream>>>((const uint32_t*)c->bases.p, (uint32_t)n, d_out, d_inf, d_bad, out_ark,
                                                          glv ? 1u : 0u);
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
    if (n > 0xFFFFFFF0ull) return fail(c, MSM_ERR_BAD_ARG, "too many points: %zu", n);
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    if ((rc = ensure(c, c->ibases, n * 64))) return rc;  // scratch: the resident set lives in its own buffers
    if ((rc = ensure(c, c->inf, n))) return rc;
    if ((rc = decompress_locked(c, compressed, n, 1u, false, (uint32_t*)c->ibases.p, (uint8_t*)c->inf.p, first_invalid))) return rc;
    HIPCHK(c, hipMemcpyAsync(out_xy_mont, c->ibases.p, n * 64, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(out_inf, c->inf.p, n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MSM_OK;
}

int32_t msm_bn254_g1_upload_compressed(msm_ctx* c, const uint8_t* compressed, size_t n, int64_t* first_invalid) {
    int32_t rc = check_common(c, compressed, compressed, n);
    if (rc) return rc;
    if (n > 0xFFFFFFF0ull) return fail(c, MSM_ERR_BAD_ARG, "too many points: %zu", n);
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    c->resident_n = 0;
    c->tuned_layout[0] = c->tuned_layout[1] = 0;
    msm_plan_t pl;
    if ((rc = resident_plan(c, n, &pl))) return fail(c, rc, "bad window_bits/flags (%u, 0x%x)", c->cfg.window_bits, c->cfg.flags);
    const bool glv = pl.glv != 0;
    if ((rc = ensure(c, c->rbases, (size_t)pl.table_factor * (glv ? 2 : 1) * n * 64))) return rc;
    if ((rc = ensure(c, c->rinf, n))) return rc;
    if ((rc = decompress_locked(c, compressed, n, 0u, glv, (uint32_t*)c->rbases.p, (uint8_t*)c->rinf.p, first_invalid))) return rc;
    if ((rc = build_table_locked(c, pl))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    c->resident_n = n;
    c->resident_glv = glv;
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

// one MSM of `scalars` (host) against the resident set of `owner`, on the pipeline (streams, workspace, pinned results) of `w`
// (w == owner, or owner's second lane)
static int32_t resident_on_lane(msm_ctx* w, const msm_ctx* owner, const uint32_t* scalars, size_t n, uint32_t* out_jac, uint32_t* out_aff,
                                uint8_t* out_inf, bool batch = false, const uint32_t* d_scalars = nullptr /* already in HBM: no upload */,
                                hipStream_t caller_stream = nullptr) {
    int32_t rc;
    if (n > owner->resident_n) n = owner->resident_n;  // unequal lengths truncate to the shorter (metal_msm.rs:652-656)
    Range r_("msm_bn254_g1_resident");
    auto t0 = std::chrono::steady_clock::now();
    const hipStream_t st0 = caller_stream ? caller_stream : w->stream;
    if (w->stage_timing) HIPCHK(w, hipEventRecord(w->ev[EV_START], st0));
    if (!d_scalars && (rc = ensure(w, w->scalars, n * 32))) return rc;
    // the phi records sit at index resident_n + i: a call on fewer scalars (truncation) or a set uploaded without them runs unsplit
    const uint32_t extra = (owner->resident_glv && n == owner->resident_n) ? 0u : MSM_FLAG_NO_GLV;
    // the window table (MSM_FLAG_WINDOW_TABLE) serves calls on the WHOLE resident set; a call on fewer scalars runs the plain pipeline
    // on the first n records of T_0, which are the bases themselves
    const bool use_table = owner->table_f > 1 && n == owner->resident_n;
    const uint32_t tab_c = use_table ? owner->table_c : 0u, tab_f = use_table ? owner->table_f : 1u;
    const uint32_t* rb = (const uint32_t*)owner->rbases.p;
    const uint8_t* ri = owner->resident_has_inf ? (const uint8_t*)owner->rinf.p : nullptr;
    PipeState ps;
    if (!batch) {
        if (!d_scalars) HIPCHK(w, hipMemcpyAsync(w->scalars.p, scalars, n * 32, hipMemcpyHostToDevice, st0));
        if (w->stage_timing) HIPCHK(w, hipEventRecord(w->ev[EV_H2D], st0));
        rc = run_pipeline(w, rb, ri, d_scalars ? d_scalars : (const uint32_t*)w->scalars.p, n, st0, out_jac, out_aff, out_inf, 0, nullptr, extra, &ps,
                          tab_c, tab_f);
        if (rc) return rc;
    } else {
        // Two MSMs in flight.  The pipelines must run in ANTI-phase: one uploads its scalars (DMA, 0.6 ms at 2^20) and finishes on
        // the CPU while the other computes.  Left alone, two host threads start together and stay in lock-step -- two uploads
        // sharing the link, then two pipelines sharing the CUs -- and kernels of two queues do not share a busy GPU fairly (a 5 us
        // k_combine_long waited 880 us for the other pipeline's k_accumulate; rocprofv3 timelines in profiles/NOTES_r2.md).  So:
        //   - uploads go to each pipeline's copy stream and take turns (each waits for the other pipeline's latest one);
        //   - with owner->batch_shared_stream the kernels of BOTH pipelines go to the owner's stream, one whole MSM at a time
        //     (enqueue under batch_mu), completion by event: the GPU runs MSM after MSM without a gap while the uploads and the
        //     host finishes happen beside it;  without it (small sizes, where no kernel fills the GPU) each pipeline keeps its
        //     own stream and the kernels overlap.
        msm_ctx* o = const_cast<msm_ctx*>(owner);
        const bool shared = o->batch_shared_stream;
        // (shared: BOTH pipelines upload through the owner's copy stream.  Uploads queued on the second pipeline's own copy stream
        // slowed the kernels running beside them 1.4-4.6x -- k_coarse_scatter 33 -> 47 us, k_fine_sort 42 -> 85, k_chunk_map 18 -> 84 --
        // while the owner's did not: 1.61-1.62 -> 1.51-1.58 ms per MSM at 2^20, NOTES_r2.md section 7)
        hipStream_t cs = shared ? o->copy_stream : w->stream;
        hipStream_t st = shared ? o->stream : w->stream;
        {
            std::lock_guard<std::mutex> cp(o->copy_mu);
            if (o->last_copy && o->last_copy != w->ev_fork) HIPCHK(w, hipStreamWaitEvent(cs, o->last_copy, 0));
            HIPCHK(w, hipMemcpyAsync(w->scalars.p, scalars, n * 32, hipMemcpyHostToDevice, cs));
            HIPCHK(w, hipEventRecord(w->ev_fork, cs));
            o->last_copy = w->ev_fork;
        }
        {
            std::unique_lock<std::mutex> q;
            if (shared) q = std::unique_lock<std::mutex>(o->batch_mu);
            if (w->stage_timing) HIPCHK(w, hipEventRecord(w->ev[EV_H2D], st));
            if (shared) HIPCHK(w, hipStreamWaitEvent(st, w->ev_fork, 0));
            if ((rc = pipe_prepare(w, n, 0, extra, st, &ps, tab_c, tab_f))) return rc;
            if ((rc = enqueue_body(w, ps, rb, ri, (const uint32_t*)w->scalars.p, 0, st, nullptr))) return rc;
            // The bucket reduction -- launch-bound pairwise levels that leave most of the GPU idle -- leaves the shared stream: it runs on
            // the second pipeline's copy stream (idle in this mode, HIGH-priority pool) beside the decomposition and sort of the next MSM:
            // per MSM 2^19 0.929 -> 0.875-0.895 ms, 2^20 1.55-1.56 -> 1.49-1.51, with the window table 1.46-1.475 -> 1.40-1.425, 2^21
            // 2.85 -> 2.75 (profiles/r3_batch_reduce_stream.txt).  The priority is what makes it work: on a plain stream -- also one with
            // a hardware queue of its own (full CU mask) -- the levels queue behind the next accumulation's workgroups and the batch
            // LOSES 10-18 %; leaving 8-32 CUs out of the shared stream's CU mask for them loses 5-10 %; k_combine moved along: no gain.
            // (o->red_active: set by the batch call for MSM_BATCH_LAYOUT_ONE_STREAM_REDUCE -- msm_config_t.batch_layout or msm_tune_batch.)
            hipStream_t rs = st;
            if (shared && o->red_active && o->lane1 && w->ev_body) {
                rs = o->lane1->copy_stream;
                HIPCHK(w, hipEventRecord(w->ev_body, st));
                HIPCHK(w, hipStreamWaitEvent(rs, w->ev_body, 0));
            }
            if ((rc = enqueue_reduce(w, ps, rs, w->h_qsums, w->h_flags))) return rc;
            if (shared) HIPCHK(w, hipEventRecord(w->ev_bases, rs));
        }
        if ((rc = finish_sync(w, ps, n, st, out_jac, out_aff, out_inf, shared ? w->ev_bases : nullptr))) return rc;
    }
    w->tm.h2d_ms = stage_ms(w, EV_START, EV_H2D);
    w->tm.convert_ms = 0;
    w->tm.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    trace_line(w, d_scalars ? "resident device" : "resident", ps);
    return MSM_OK;
}

int32_t msm_bn254_g1_resident(msm_ctx* c, const uint32_t* scalars, size_t n, uint32_t out_jac[24], uint32_t out_aff[16],
                              uint8_t* out_inf) {
    int32_t rc = check_common(c, scalars, scalars, n);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->resident_n == 0) return fail(c, MSM_ERR_STATE, "no resident bases: call msm_bn254_g1_upload_bases first");
    DeviceGuard g(c->device);
    HostPin pin_s;  // (pageable scalars travel at the pinned rate: 32 MB at 2^20)
    pin_s.pin(scalars, std::min(n, c->resident_n) * 32);
    return resident_on_lane(c, c, scalars, n, out_jac, out_aff, out_inf);
}

// scalars already in HBM (a prover whose witness lives on the GPU) against the resident set -- and its window table, when it has one
int32_t msm_bn254_g1_resident_device(msm_ctx* c, const void* d_scalars, size_t n, void* hip_stream, uint32_t out_jac[24],
                                     uint32_t out_aff[16], uint8_t* out_inf) {
    int32_t rc = check_common(c, d_scalars, d_scalars, n);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->resident_n == 0) return fail(c, MSM_ERR_STATE, "no resident bases: call msm_bn254_g1_upload_bases first");
    DeviceGuard g(c->device);
    return resident_on_lane(c, c, nullptr, n, out_jac, out_aff, out_inf, false, (const uint32_t*)d_scalars, (hipStream_t)hip_stream);
}

// `count` MSMs against the resident bases with TWO of them in flight: a second pipeline inside the context (a context of its
// own: streams, workspace, pinned result buffers; created on first use) is driven by a second host thread.  How provers call MSM
// (several scalar vectors per proof against fixed bases; SURVEY.md section 8 row f2 "multiple MSMs in flight").  Measured per MSM,
// single calls -> batch (tools/two_ctx_throughput.py, pinned host scalars): 2^14 0.40 -> 0.24 ms, 2^16 0.45 -> 0.30, 2^17 0.56 -> 0.40,
// 2^18 0.81 -> 0.63, 2^20 2.28 -> 1.63, 2^22 8.54 -> 5.80 (= the kernel time of one MSM: upload and host finish fully hidden).
// the layout of a batch call on n (already clamped) points: configured, else tuned (msm_tune_batch), else by size
static uint32_t batch_layout_for(const msm_ctx* c, size_t n) {
    if (c->cfg.batch_layout) return c->cfg.batch_layout;
    const int cls = n >= ((size_t)1 << 19) ? 1 : 0;
    if (c->tuned_layout[cls]) return c->tuned_layout[cls];
    return cls ? MSM_BATCH_LAYOUT_ONE_STREAM : MSM_BATCH_LAYOUT_TWO_STREAMS;
}

// msm_bn254_g1_resident_batch under the context's mutex; forced_layout != 0: msm_tune_batch measuring that layout
static int32_t resident_batch_locked(msm_ctx* c, const uint32_t* const* scalars, size_t n, size_t count, uint32_t* out_jac, uint32_t* out_aff,
                                     uint8_t* out_inf, uint32_t forced_layout) {
    if (count > 1 && !c->lane1) {
        msm_config_t cfg = c->cfg;
        cfg.device = c->device;
        cfg.max_points = 0;
        int32_t rc = ctx_create_impl(&cfg, &c->lane1, 1 /* a stream of the low-priority pool */, &c->knobs);
        if (rc) return fail(c, rc, "second pipeline: %s", msm_last_error(nullptr));
        c->batch_pool = new (std::nothrow) HostPool(1);
        if (!c->batch_pool) return fail(c, MSM_ERR_OOM, "host allocation failed");
    }
    std::atomic<size_t> next{0};
    const size_t upto = count;
    std::atomic<int32_t> rcs[2] = {{MSM_OK}, {MSM_OK}};
    auto lane = [&](int k) {
        msm_ctx* w = k == 0 ? c : c->lane1;
        DeviceGuard gl(c->device);  // per host thread
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= upto || rcs[0] != MSM_OK || rcs[1] != MSM_OK) break;
            const int32_t rc = resident_on_lane(w, c, scalars[i], n, out_jac + i * 24, out_aff ? out_aff + i * 16 : nullptr,
                                                out_inf ? out_inf + i : nullptr, count > 1);
            if (rc) rcs[k] = rc;
        }
    };
    c->last_copy = nullptr;
    // How the two pipelines share the GPU: msm_config_t.batch_layout, else what msm_tune_batch measured for this size class, else by size
    // (include/msm_hip.h MSM_BATCH_LAYOUT_*).  A pure function of the configuration, the tuned choice and the CLAMPED n: round 3 timed the
    // first four batch calls of a context to choose, and one noisy call decided for the context's life.
    const uint32_t layout = forced_layout ? forced_layout : batch_layout_for(c, std::min(n, c->resident_n));
    const bool red_ok = count > 1;  // the reduce stream is the second pipeline's copy stream
    c->batch_shared_stream = layout != MSM_BATCH_LAYOUT_TWO_STREAMS;
    c->red_active = layout == MSM_BATCH_LAYOUT_ONE_STREAM_REDUCE && red_ok;
    c->last_batch_layout = layout;
    if (count > 1)
        for (msm_ctx* w : {c, c->lane1})
            if (!w->ev_body) HIPCHK(c, hipEventCreateWithFlags(&w->ev_body, hipEventDisableTiming));
    if (count > 1) c->batch_pool->run(2, lane);
    else lane(0);
    c->red_active = false;
    c->tm.batch_layout = c->last_batch_layout;
    if (rcs[0] == MSM_OK && rcs[1] != MSM_OK) c->err = c->lane1->err;
    return rcs[0] != MSM_OK ? rcs[0] : rcs[1];
}

int32_t msm_bn254_g1_resident_batch(msm_ctx* c, const uint32_t* const* scalars, size_t n, size_t count, uint32_t* out_jac,
                                    uint32_t* out_aff, uint8_t* out_inf) {
    if (!c) return MSM_ERR_BAD_ARG;
    if (n == 0 || count == 0) return fail(c, MSM_ERR_EMPTY, "Empty input");
    if (!scalars || !out_jac) return fail(c, MSM_ERR_BAD_ARG, "NULL scalars / result pointer");
    for (size_t i = 0; i < count; i++)
        if (!scalars[i]) return fail(c, MSM_ERR_BAD_ARG, "NULL scalar vector %zu", i);
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->resident_n == 0) return fail(c, MSM_ERR_STATE, "no resident bases: call msm_bn254_g1_upload_bases first");
    DeviceGuard g(c->device);
    std::unique_ptr<HostPin[]> pins(new (std::nothrow) HostPin[count]);  // (a vector listed twice is registered once: the second look finds it known)
    if (pins)
        for (size_t i = 0; i < count; i++) pins[i].pin(scalars[i], std::min(n, c->resident_n) * 32);
    return resident_batch_locked(c, scalars, n, count, out_jac, out_aff, out_inf, 0);
}

// The explicit measurement of the batch layout (include/msm_hip.h).  Every layout: one untimed batch (hardware queues, the second
// pipeline's workspace), then `reps` timed ones, the minimum counts.
int32_t msm_tune_batch(msm_ctx* c, const uint32_t* const* scalars, size_t n, size_t count, uint32_t reps, uint32_t* chosen, double* ms_per_msm) {
    if (!c) return MSM_ERR_BAD_ARG;
    if (n == 0 || count == 0) return fail(c, MSM_ERR_EMPTY, "Empty input");
    if (!scalars) return fail(c, MSM_ERR_BAD_ARG, "NULL scalars pointer");
    if (count < 2) return fail(c, MSM_ERR_BAD_ARG, "msm_tune_batch needs at least two scalar vectors (two MSMs in flight)");
    for (size_t i = 0; i < count; i++)
        if (!scalars[i]) return fail(c, MSM_ERR_BAD_ARG, "NULL scalar vector %zu", i);
    if (reps == 0) reps = 3;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->resident_n == 0) return fail(c, MSM_ERR_STATE, "no resident bases: call msm_bn254_g1_upload_bases first");
    DeviceGuard g(c->device);
    std::vector<uint32_t> jac(count * 24);
    const uint32_t layouts[3] = {MSM_BATCH_LAYOUT_ONE_STREAM, MSM_BATCH_LAYOUT_ONE_STREAM_REDUCE, MSM_BATCH_LAYOUT_TWO_STREAMS};
    double best[3] = {0, 0, 0};
    for (int pass = 0; pass < 2; pass++)  // pass 0 warms every layout up, pass 1 times them (interleaved, so that a drifting clock hits all three)
        for (uint32_t r = 0; r < (pass ? reps : 1u); r++)
            for (int k = 0; k < 3; k++) {
                const auto t0 = std::chrono::steady_clock::now();
                const int32_t rc = resident_batch_locked(c, scalars, n, count, jac.data(), nullptr, nullptr, layouts[k]);
                if (rc) return rc;
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (double)count;
                if (pass && (best[k] == 0 || ms < best[k])) best[k] = ms;
            }
    int win = 0;
    for (int k = 1; k < 3; k++)
        if (best[k] < best[win]) win = k;
    const size_t nc = std::min(n, c->resident_n);
    c->tuned_layout[nc >= ((size_t)1 << 19) ? 1 : 0] = layouts[win];
    if (chosen) *chosen = c->cfg.batch_layout ? c->cfg.batch_layout : layouts[win];
    if (ms_per_msm)
        for (int k = 0; k < 3; k++) ms_per_msm[k] = best[k];
    if (trace_enabled())
        std::fprintf(stderr, "[msm_hip] msm_tune_batch n %zu count %zu: one stream %.4f, one stream + reduce stream %.4f, two streams %.4f ms per MSM -> layout %u\n",
                     nc, count, best[0], best[1], best[2], layouts[win]);
    return MSM_OK;
}

int32_t msm_bn254_g1_device(msm_ctx* c, const void* d_bases_mont, const void* d_inf_mask, const void* d_scalars, size_t n,
                            void* hip_stream, uint32_t out_jac[24], uint32_t out_aff[16], uint8_t* out_inf) {
    int32_t rc = check_common(c, d_bases_mont, d_scalars, n);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    Range r_("msm_bn254_g1_device");
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->stream;
    auto t0 = std::chrono::steady_clock::now();
    const bool glv = plan_glv(c, n);
    if ((rc = ensure(c, c->ibases, glv ? n * 64 : 0))) return rc;  // the phi records of a split plan; an unsplit plan needs no copy of the bases at all
    PipeState ps;
    uint32_t* ib = (uint32_t*)c->ibases.p;
    if (c->stage_timing) HIPCHK(c, hipEventRecord(c->ev[EV_H2D], st));
    // Round 5: nothing is launched for the coordinate half of K1.  The accumulation gathers the caller's arkworks words as they are
    // (k_accumulate_pieces<.., M256>); the phi records of a split plan are written by the decomposition itself (k_decompose_glv<.., PHI>) --
    // one stream, no cross-stream event, at every size.
    BaseSrc src;
    src.m256 = true, src.rec = (const uint32_t*)d_bases_mont;
    if (glv) src.phi = ib, src.phi_fill = ib, src.nsplit = (uint32_t)n;
    rc = run_pipeline(c, src, (const uint8_t*)d_inf_mask, (const uint32_t*)d_scalars, n, st, out_jac, out_aff, out_inf, 0, nullptr, 0, &ps);
    if (rc) return rc;
    c->tm.h2d_ms = 0;
    c->tm.convert_ms = stage_ms(c, EV_H2D, EV_CONVERT);
    c->tm.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    trace_line(c, "device", ps);
    return MSM_OK;
}

int32_t msm_bn254_g1_combine(const uint32_t* partials, size_t k, uint32_t out_jac[24], uint32_t out_aff[16],
                             uint8_t* out_inf) {
    return combine_partials(partials, k, out_jac, out_aff, out_inf, false);
}

int32_t msm_bn254_g1_combine_flags(const uint32_t* partials, size_t k, uint32_t flags, uint32_t out_jac[24], uint32_t out_aff[16],
                                   uint8_t* out_inf) {
    if (flags & ~(uint32_t)MSM_FLAG_DETERMINISTIC) return MSM_ERR_BAD_ARG;
    return combine_partials(partials, k, out_jac, out_aff, out_inf, (flags & MSM_FLAG_DETERMINISTIC) != 0);
}

int32_t msm_plan(size_t n, uint32_t window_bits, uint32_t flags, msm_plan_t* out) {
    if (!out) return MSM_ERR_BAD_ARG;
    if (n == 0) return MSM_ERR_EMPTY;
    // what a context created NOW would plan (no context here)
    if (flags & MSM_FLAG_WINDOW_TABLE) return msmplan::make_table_plan(n, window_bits, flags, out, msmplan::glv_max_from_env(), Knobs::from_env().table);
    return make_plan(n, window_bits, flags, out, msmplan::glv_max_from_env());
}

int32_t msm_get_timings(const msm_ctx* c, msm_timings_t* out) {
    if (!c || !out) return MSM_ERR_BAD_ARG;
    *out = c->tm;
    return MSM_OK;
}
int32_t msm_get_timings_sized(const msm_ctx* c, void* out, size_t out_size) {
    if (!c || !out || out_size == 0) return MSM_ERR_BAD_ARG;
    std::memcpy(out, &c->tm, std::min(out_size, sizeof(msm_timings_t)));
    return MSM_OK;
}
int32_t msm_set_kernel_timing(msm_ctx* c, uint32_t every_n) {
    if (!c) return MSM_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    c->ktime_every = every_n;
    c->ktime_count = 0;
    if (c->lane1) msm_set_kernel_timing(c->lane1, every_n);
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
    c->stage_timing = enabled != 0 || trace_enabled();
    return MSM_OK;
}
void msm_reset_kernel_stats(msm_ctx* c) {
    if (!c) return;
    std::lock_guard<std::mutex> lk(c->mu);
    c->acc_ms_sum = 0;
    c->acc_launches = 0;
    if (c->clk.p) {
        DeviceGuard g(c->device);
        (void)hipMemsetAsync(c->clk.p, 0, 32, c->stream);
        (void)hipStreamSynchronize(c->stream);
    }
}
// Clock probe of k_accumulate since the last reset: the first workgroup of every launch reads the shader-cycle counter and the
// constant-rate counter around its chunk.  sclk_ghz = cycles / ticks x the constant counter's rate: the shader clock the kernel really
// sustained; cycles_per_addition = shader cycles one wavefront needed per mixed addition (with its two neighbours on the SIMD).
int32_t msm_get_clock_stats(msm_ctx* c, double* sclk_ghz, double* cycles_per_addition, uint64_t* samples) {
    if (!c) return MSM_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(c->mu);
    DeviceGuard g(c->device);
    unsigned long long v[4] = {0, 0, 0, 0};
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(v, c->clk.p, 32, hipMemcpyDeviceToHost));
    if (sclk_ghz) *sclk_ghz = v[1] ? (double)v[0] / (double)v[1] * (double)c->wall_clock_khz * 1e-6 : 0.0;
    if (cycles_per_addition) *cycles_per_addition = v[3] ? (double)v[0] / (double)v[3] : 0.0;
    if (samples) *samples = v[2];
    return MSM_OK;
}

}  // extern "C"

#include "msm_multi.inc"

#ifdef MSM_HIP_TEST_HOOKS
#include "msm_testhooks.inc"
#endif
