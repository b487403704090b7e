// msm_host_pool.hpp -- the host runtime's small persistent thread pool (CPU finish of an MSM, staging copies of pageable inputs).
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

// spin-wait hint of the host architecture (x86: pause; AArch64: yield; elsewhere: a compiler barrier)
static inline void msm_cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__) || defined(__arm__)
    __asm__ __volatile__("yield" ::: "memory");
#else
    __asm__ __volatile__("" ::: "memory");
#endif
}

// Small persistent host thread pool for the CPU finish (per-window Horner chains are independent).  The
// reference runs its CPU finish under rayon (metal_msm.rs:214-247); std::thread + a condition variable here.
// run() returns when every JOB is done, not when every worker has checked in: a worker the OS wakes late (seen as
// 3-10 ms outliers of the finish stage) simply finds nothing left, because the caller and the punctual workers pull jobs
// from one ticket counter.  The ticket carries the generation (its low 32 bits: a late worker would have to sleep through 2^32 runs
// to confuse two of them), so a late worker can never take a job of a later run().
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
            if ((uint32_t)(v >> 32) != (uint32_t)gen || (int)(uint32_t)v >= njobs) return;  // the ticket holds the LOW 32 bits of the generation
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
                for (uint32_t spins = 1; (uint32_t)(ticket_.load(std::memory_order_acquire) >> 32) == (uint32_t)seen; spins++) {
                    if ((spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
                    msm_cpu_relax();
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
