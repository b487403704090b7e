// tools/host_asan_check.cpp -- HOST-only AddressSanitizer + UBSan build of the parts of the host runtime that need no GPU
// (SURVEY.md section 5 "sanitizer host build"; GPU ASan is not available on this pool): the planner (msm_planner.hpp), the thread pool
// (msm_host_pool.hpp), the CPU finish arithmetic (host_g1.hpp), the host side of the GLV split (glv_bn254.hpp) and the shard /
// chunk-schedule arithmetic.  Built by `make -C gpu-acceleration_amd/csrc asan`, run by tests/test_host_asan.py.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#define FP_HD inline
#include "../gpu-acceleration_amd/csrc/glv_bn254.hpp"
#include "../gpu-acceleration_amd/csrc/host_g1.hpp"
#include "../gpu-acceleration_amd/csrc/msm_host_pool.hpp"
#include "../gpu-acceleration_amd/csrc/msm_planner.hpp"

#define REQUIRE(c)                                                              \
    do {                                                                        \
        if (!(c)) {                                                             \
            std::fprintf(stderr, "FAILED: %s (%s:%d)\n", #c, __FILE__, __LINE__); \
            return 1;                                                           \
        }                                                                       \
    } while (0)

// row f4: the window-table planner (make_table_plan) and the top-window shift
static int check_table_planner() {
    const msmplan::table_knobs tk{};
    for (uint32_t base : {0u, (uint32_t)MSM_FLAG_NO_GLV})
        for (uint32_t c = 0; c <= 20; c++) {
            if (c == 1) continue;
            for (size_t n : {(size_t)1, (size_t)255, (size_t)1 << 10, (size_t)1 << 16, (size_t)1 << 18, ((size_t)1 << 18) + 1, (size_t)1 << 20, (size_t)1 << 21,
                             ((size_t)1 << 21) + 1, (size_t)1 << 24}) {
                msm_plan_t p, q;
                REQUIRE(msmplan::make_table_plan(n, c, base | MSM_FLAG_WINDOW_TABLE, &p, msmplan::GLV_MAX_POINTS, tk) == MSM_OK);
                REQUIRE(msmplan::make_plan(n, c, base, &q) == MSM_OK);
                if (p.table_factor > 1) {
                    REQUIRE(p.table_factor == p.num_windows && p.bucket_arrays == 1);  // the planner only makes FULL tables
                    REQUIRE(p.table_bytes == (uint64_t)p.table_factor * p.virtual_points * 64);
                    REQUIRE((uint64_t)p.table_factor * p.virtual_points <= msmplan::TABLE_MAX_ENTRIES);
                    REQUIRE((uint64_t)p.num_windows * p.window_bits >= p.scalar_bits);
                    const uint32_t s = msmplan::table_top_shift(p, p.table_factor);
                    const uint32_t top_bits = p.scalar_bits - p.window_bits * (p.num_windows - 1);
                    REQUIRE(((uint64_t)1 << (top_bits + s)) <= ((uint64_t)1 << (p.window_bits - 1)));  // shifted top digit <= H
                    REQUIRE(s == 0 || top_bits + s == p.window_bits - 1);
                    if (c) REQUIRE(p.window_bits == c);
                    if (n > msmplan::TABLE_GLV_MAX_POINTS) REQUIRE(!p.glv);
                } else {  // no table: exactly the ordinary plan
                    REQUIRE(p.window_bits == q.window_bits && p.num_windows == q.num_windows && p.glv == q.glv && p.table_bytes == 0);
                }
            }
        }
    msm_plan_t p;
    REQUIRE(msmplan::make_table_plan((size_t)1 << 20, 0, MSM_FLAG_WINDOW_TABLE, &p, msmplan::GLV_MAX_POINTS, tk) == MSM_OK);
    REQUIRE(p.window_bits == 20 && p.num_windows == 13 && p.table_factor == 13 && !p.glv && msmplan::table_top_shift(p, 13) == 5);
    REQUIRE(msmplan::make_table_plan((size_t)1 << 17, 0, MSM_FLAG_WINDOW_TABLE, &p, msmplan::GLV_MAX_POINTS, tk) == MSM_OK);
    REQUIRE(p.window_bits == 16 && p.num_windows == 8 && p.table_factor == 8 && p.glv && msmplan::table_top_shift(p, 8) == 0);
    REQUIRE(msmplan::make_table_plan((size_t)1 << 22, 0, MSM_FLAG_WINDOW_TABLE, &p, msmplan::GLV_MAX_POINTS, tk) == MSM_OK && p.table_factor == 1);
    REQUIRE(msmplan::make_table_plan((size_t)1 << 16, 0, MSM_FLAG_WINDOW_TABLE | MSM_FLAG_UNSIGNED_DIGITS, &p, msmplan::GLV_MAX_POINTS, tk) == MSM_OK && p.table_factor == 1);
    msmplan::table_knobs small = tk;
    small.max_bytes = (size_t)1 << 20;  // 1 MiB cap: a 2^16 table (67 MB) is not made
    REQUIRE(msmplan::make_table_plan((size_t)1 << 16, 0, MSM_FLAG_WINDOW_TABLE, &p, msmplan::GLV_MAX_POINTS, small) == MSM_OK && p.table_factor == 1);
    msmplan::table_knobs part = tk;
    part.c = 16, part.f = 4;  // a forced partial table: 2 arrays of 4 windows, no top shift
    REQUIRE(msmplan::make_table_plan((size_t)1 << 16, 0, MSM_FLAG_WINDOW_TABLE, &p, msmplan::GLV_MAX_POINTS, part) == MSM_OK);
    REQUIRE(p.table_factor == 4 && p.bucket_arrays == 2 && msmplan::table_top_shift(p, 4) == 0);
    part.f = 3;  // does not divide 8 windows: the ordinary plan
    REQUIRE(msmplan::make_table_plan((size_t)1 << 16, 0, MSM_FLAG_WINDOW_TABLE, &p, msmplan::GLV_MAX_POINTS, part) == MSM_OK && p.table_factor == 1);
    return 0;
}

static int check_planner() {
    for (uint32_t flags = 0; flags < 4; flags++)
        for (uint32_t c = 0; c <= 20; c++) {
            if (c == 1) continue;
            for (size_t n : {(size_t)1, (size_t)2, (size_t)255, (size_t)1 << 10, (size_t)1 << 16, ((size_t)1 << 19) + 1, (size_t)1 << 24, (size_t)1 << 30}) {
                msm_plan_t p;
                int32_t rc = msmplan::make_plan(n, c, flags, &p);
                REQUIRE(rc == MSM_OK);
                REQUIRE(p.num_windows >= 1 && p.num_windows <= 128);
                REQUIRE((uint64_t)p.num_windows * p.window_bits >= p.scalar_bits);  // the windows cover the scalar
                REQUIRE(p.virtual_points == (p.glv ? 2 * n : n));
                REQUIRE(p.num_buckets == (p.signed_digits ? 1u << (p.window_bits - 1) : 1u << p.window_bits));
                if (c) REQUIRE(p.window_bits == c);
            }
        }
    msm_plan_t p;
    REQUIRE(msmplan::make_plan(16, 1, 0, &p) == MSM_ERR_BAD_ARG);
    REQUIRE(msmplan::make_plan(16, 21, 0, &p) == MSM_ERR_BAD_ARG);
    REQUIRE(msmplan::make_plan(16, 0, 16, &p) == MSM_ERR_BAD_ARG);
    return 0;
}

static int check_pool() {
    for (int workers : {1, 3, 7}) {
        HostPool pool(workers);
        for (int round = 0; round < 200; round++) {
            const int njobs = 1 + round % 13;
            std::vector<int> hits((size_t)njobs, 0);
            if (round % 3 == 0) pool.arm();
            pool.run(njobs, [&](int k) { hits[(size_t)k]++; });
            for (int k = 0; k < njobs; k++) REQUIRE(hits[(size_t)k] == 1);
        }
    }  // destructor joins armed and idle workers
    return 0;
}

static int check_g1() {
    using namespace hostg1;
    const Jac g{ONE, dbl(ONE), ONE};  // (1, 2)
    Fq x, y;
    // known answers: 2G, 3G
    const uint64_t x2[4] = {0xd3c208c16d87cfd3ULL, 0xd97816a916871ca8ULL, 0x9b85045b68181585ULL, 0x030644e72e131a02ULL};
    const uint64_t x3[4] = {0xf2d355961915abf0ULL, 0x9315d84715b8e679ULL, 0xf40232bcb1b6bd15ULL, 0x0769bf9ac56bea3fULL};
    REQUIRE(!to_affine_std(jdbl(g), x, y) && std::memcmp(x.l, x2, 32) == 0);
    REQUIRE(!to_affine_std(jadd(jdbl(g), g), x, y) && std::memcmp(x.l, x3, 32) == 0);
    // group identities through the complete addition: P + P = 2P, P + (-P) = 0, 0 + P = P, (a+b)G by two routes
    Jac acc = identity(), p7 = identity();
    for (int i = 0; i < 7; i++) p7 = jadd(p7, g);
    acc = jadd(jdbl(jdbl(g)), jadd(jdbl(g), g));  // 4G + 3G
    Fq xa, ya, xb, yb;
    REQUIRE(!to_affine_std(acc, xa, ya) && !to_affine_std(p7, xb, yb) && std::memcmp(xa.l, xb.l, 32) == 0 && std::memcmp(ya.l, yb.l, 32) == 0);
    Jac neg = p7;
    neg.y = sub(Fq{{0, 0, 0, 0}}, p7.y);
    REQUIRE(is_identity(jadd(p7, neg)));
    REQUIRE(!to_affine_std(jadd(p7, p7), xa, ya) && !to_affine_std(jdbl(p7), xb, yb) && std::memcmp(xa.l, xb.l, 32) == 0);
    REQUIRE(!to_affine_std(jadd(identity(), p7), xa, ya) && !to_affine_std(p7, xb, yb) && std::memcmp(ya.l, yb.l, 32) == 0);
    // word round trip
    uint32_t w[24];
    store_jac(w, p7);
    Jac back = load_jac(w);
    REQUIRE(std::memcmp(back.x.l, p7.x.l, 32) == 0 && std::memcmp(back.z.l, p7.z.l, 32) == 0);
    // field: a * a^-1 = 1 for a few elements
    std::mt19937_64 rng(5);
    for (int i = 0; i < 20; i++) {
        Fq a{{rng(), rng(), rng(), rng() >> 3}};
        if (geq_mod(a)) sub_mod_inplace(a);
        if (is_zero(a)) continue;
        Fq one = mul(a, inv(a));
        REQUIRE(std::memcmp(one.l, ONE.l, 32) == 0);
    }
    return 0;
}

static int check_glv() {
    std::mt19937_64 rng(11);
    for (int it = 0; it < 20000; it++) {
        uint32_t k[8];
        for (auto& v : k) v = (uint32_t)rng();
        k[7] &= 0x3FFFFFFFu;
        if (it == 0) std::memset(k, 0, sizeof k);
        if (it == 1) std::memset(k, 0xFF, sizeof k), k[7] = 0x3FFFFFFFu;
        uint32_t k1[4], k2[4];
        bool n1, n2;
        REQUIRE(glv::split(k, k1, n1, k2, n2));          // both halves below 7 * 2^123 for every scalar below 2^254
        REQUIRE((k1[3] >> 27) < 7 && (k2[3] >> 27) < 7);
    }
    // the top window of a split plan (msmplan::glv_top_digit_bits): c = 16: 8 windows, magnitudes <= 14336 -> 14 of 15 index bits; c = 10:
    // 13 windows, <= 56 -> 6 of 9; c = 9: 15 windows, only the carry -> 0 of 8; unsigned c = 15: 9 windows, <= 55 -> 6 of 15
    REQUIRE(msmplan::glv_top_digit_bits(16, 8, true, 15) == 14 && msmplan::glv_top_digit_bits(10, 13, true, 9) == 6);
    REQUIRE(msmplan::glv_top_digit_bits(9, 15, true, 8) == 0 && msmplan::glv_top_digit_bits(15, 9, false, 15) == 6);
    REQUIRE(msmplan::glv_top_digit_bits(13, 10, true, 12) == 9 && msmplan::glv_top_digit_bits(17, 8, true, 16) == 7);
    {   // unsplit plans: the top window holds 254 - c*(W-1) bits
        msm_plan_t p;
        REQUIRE(msmplan::make_plan((size_t)1 << 20, 16, MSM_FLAG_NO_GLV, &p) == MSM_OK && p.num_windows == 16 && p.top_digit_bits == 14);
        REQUIRE(msmplan::make_plan((size_t)1 << 20, 20, MSM_FLAG_NO_GLV, &p) == MSM_OK && p.num_windows == 13 && p.top_digit_bits == 14);
        REQUIRE(msmplan::make_plan((size_t)1 << 20, 17, MSM_FLAG_NO_GLV, &p) == MSM_OK && p.num_windows == 15 && p.top_digit_bits == 16);
        REQUIRE(msmplan::make_plan((size_t)1 << 20, 9, MSM_FLAG_NO_GLV | MSM_FLAG_UNSIGNED_DIGITS, &p) == MSM_OK && p.num_windows == 29 && p.top_digit_bits == 2);
    }
    for (size_t n : {(size_t)1 << 10, (size_t)1 << 14, (size_t)1 << 20, (size_t)1 << 22}) {
        msm_plan_t p;
        REQUIRE(msmplan::make_plan(n, 0, 0, &p) == MSM_OK);
        uint32_t kb = 0;
        while ((1u << kb) < p.num_buckets) kb++;
        REQUIRE(p.glv ? p.top_digit_bits < kb : p.top_digit_bits == kb);
        REQUIRE(msmplan::make_plan(n, 0, MSM_FLAG_WINDOW_TABLE, &p) == MSM_OK);
        msmplan::table_knobs tk;
        REQUIRE(msmplan::make_table_plan(n, 0, MSM_FLAG_WINDOW_TABLE, &p, msmplan::GLV_MAX_POINTS, tk) == MSM_OK);
        kb = 0;
        while ((1u << kb) < p.num_buckets) kb++;
        REQUIRE(p.table_factor == 1 || p.top_digit_bits == kb);
    }
    return 0;
}

// the shard arithmetic of msm_multi (msm_multi.inc shard_bounds) and the uniform chunk schedule (msm_hip.hip stream_schedule)
static int check_partitions() {
    for (size_t n : {(size_t)1, (size_t)2, (size_t)7, (size_t)1000, ((size_t)1 << 20) + 3, (size_t)1 << 30})
        for (int G = 1; G <= 9; G++) {
            size_t prev = 0;
            for (int g = 0; g <= G; g++) {
                size_t lo = (size_t)((unsigned __int128)n * (unsigned)g / (unsigned)G);
                REQUIRE(lo >= prev && lo <= n);
                prev = lo;
            }
            REQUIRE(prev == n);
        }
    for (size_t chunk : {(size_t)1 << 8, (size_t)1 << 18})
        for (size_t n = 2 * chunk; n < 6 * chunk; n += chunk / 3 + 1) {
            std::vector<size_t> sizes;
            size_t left = n;
            while (left >= chunk + chunk / 2) sizes.push_back(chunk), left -= chunk;
            if (left > chunk) sizes.push_back((left / 2 + 63) & ~(size_t)63), left -= sizes.back();
            if (left) sizes.push_back(left);
            size_t sum = 0;
            for (size_t s : sizes) {
                REQUIRE(s > 0 && s < chunk + chunk / 2);
                sum += s;
            }
            REQUIRE(sum == n);
        }
    return 0;
}

static int check_piece_plan() {
    // the BASELINE shape: 8 windows x 2^21 entries over 8 x 32768 buckets (mean 64): whole buckets up to 80 entries, runs of 64 (a power of two, ~2^18 pieces)
    {
        const msmplan::piece_plan p = msmplan::make_piece_plan((size_t)8 << 21, 64, (size_t)8 << 15);
        REQUIRE(p.pmax == 80 && p.psplit == 64);
    }
    // tiny instance, forced lengths, a later chunk keeps its first chunk's lengths
    REQUIRE(msmplan::make_piece_plan(1000, 0, 512).pmax == 8 && msmplan::make_piece_plan(1000, 0, 512).psplit == 8);
    REQUIRE(msmplan::make_piece_plan((size_t)1 << 21, 8, (size_t)8 << 15).pmax == 16);  // 2^17 points, GLV: unchanged by the tiny-instance rule
    REQUIRE(msmplan::make_piece_plan((size_t)1 << 28, 600, (size_t)1 << 18).pmax == 648 && msmplan::make_piece_plan((size_t)1 << 28, 4000, (size_t)1 << 18).pmax == 1024 && msmplan::make_piece_plan((size_t)1 << 24, 600, 4096).pmax == 128);
    REQUIRE(msmplan::make_piece_plan((size_t)1 << 21, 64, (size_t)1 << 15).pmax == 16);  // 2^17 points, split window table: one array of 2^15 buckets
    REQUIRE(msmplan::make_piece_plan((size_t)1 << 20, 64, 4096, 7).pmax == 7 && msmplan::make_piece_plan((size_t)1 << 20, 64, 4096, 7).psplit == 7);
    {
        const msmplan::piece_plan a = msmplan::make_piece_plan((size_t)15 << 22, 64, (size_t)15 << 16);
        const msmplan::piece_plan b = msmplan::make_piece_plan((size_t)15 << 20, 16, (size_t)15 << 16, 0, &a);
        REQUIRE(b.pmax == a.pmax && b.psplit == a.psplit && b.max_pieces <= a.max_pieces && b.max_partials <= a.max_partials);
    }
    // the bounds hold for ANY distribution: simulate bucket sizes (uniform, one huge bucket, many medium ones, everything just above pmax)
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13, st ^= st >> 7, st ^= st << 17; return st; };
    for (int rep = 0; rep < 200; rep++) {
        const size_t tb = 1 + rnd() % 5000;
        std::vector<size_t> sz(tb, 0);
        const int kind = rep % 4;
        size_t pairs = 0;
        for (size_t k = 0; k < tb; k++) {
            size_t v = kind == 0 ? rnd() % 100 : kind == 1 ? (k == 0 ? 200000 : rnd() % 3) : kind == 2 ? (rnd() % 7 == 0 ? 500 + rnd() % 3000 : 0) : 0;
            sz[k] = v, pairs += v;
        }
        if (pairs == 0) continue;
        const uint32_t forced = rep % 5 == 0 ? 1 + (uint32_t)(rnd() % 40) : 0;
        const msmplan::piece_plan p = msmplan::make_piece_plan(pairs, pairs / tb, tb, forced);
        if (kind == 3) {  // worst case for the partial-sum bound: every bucket one entry above pmax
            pairs = 0;
            for (size_t k = 0; k < tb; k++) sz[k] = p.pmax + 1, pairs += sz[k];
        }
        const msmplan::piece_plan q = kind == 3 ? msmplan::make_piece_plan(pairs, pairs / tb, tb, forced) : p;
        size_t pieces = 0, partials = 0;
        for (size_t k = 0; k < tb; k++) {
            if (!sz[k]) continue;
            size_t m = 1;  // msmk::piece_split
            if (sz[k] > q.pmax) {
                const size_t run = sz[k] <= (size_t)8 * q.pmax ? q.pmax : q.psplit;
                m = (sz[k] + run - 1) / run;
            }
            pieces += m;
            if (m > 1) partials += m;
        }
        REQUIRE(q.psplit >= 1 && q.psplit <= q.pmax && q.pmax <= msmplan::PIECE_BINS_MAX);
        REQUIRE(pieces <= q.max_pieces && partials <= q.max_partials);
        // the same buckets inside an instance planned for up to 64 x as many entries (mostly-zero scalars): the kernels shorten the runs
        // (msmplan::effective_psplit) and the plan's bounds must still hold
        if (!forced) {
            const size_t pairs_max = pairs * (1 + rnd() % 64);
            const msmplan::piece_plan w = msmplan::make_piece_plan(pairs_max, pairs_max / tb, tb);
            const uint32_t run_eff = msmplan::effective_psplit(w.psplit, msmplan::SPLIT_ENTRIES_SHIFT, (uint32_t)pairs);
            REQUIRE(run_eff >= 8 || run_eff == w.psplit);
            REQUIRE(run_eff <= w.psplit);
            // ... and raise the longest whole bucket to the occupancy of the NON-EMPTY buckets (msmplan::effective_pmax); counted fully, partly (oversized regions are
            // left out of the count) or not at all
            size_t nonempty = 0;
            for (size_t k = 0; k < tb; k++) nonempty += sz[k] != 0;
            for (const size_t counted : {nonempty, nonempty / 3, (size_t)0}) {
                const uint32_t pmax_eff = msmplan::effective_pmax(w.pmax, (uint32_t)pairs, (uint32_t)counted);
                REQUIRE(pmax_eff >= w.pmax && pmax_eff <= msmplan::PIECE_BINS_MAX && (pmax_eff == w.pmax || pmax_eff <= std::max<size_t>(pairs >> 16, 32)));
                size_t pc = 0, pt = 0;
                for (size_t k = 0; k < tb; k++) {
                    if (!sz[k]) continue;
                    size_t m = 1;
                    if (sz[k] > pmax_eff) {
                        const size_t run = sz[k] <= (size_t)8 * pmax_eff ? pmax_eff : run_eff;
                        m = (sz[k] + run - 1) / run;
                    }
                    pc += m;
                    if (m > 1) pt += m;
                }
                REQUIRE(pc <= w.max_pieces && pt <= w.max_partials);
            }
        }
    }
    return 0;
}

int main() {
    if (check_planner() || check_table_planner() || check_piece_plan() || check_pool() || check_g1() || check_glv() || check_partitions()) return 1;
    std::puts("host runtime: planner, window-table planner, piece plan, pool, host_g1, glv split, partitions clean under ASan/UBSan");
    return 0;
}
