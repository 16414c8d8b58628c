// convert_helper.h -- host-only: the fp64 -> f32 conversion of a host batch (NeuralNet.gradientStep(double[] rows),
// NNT:83) and the helper threads that share it.  No HIP in here: tests/native/convert_helper_check.cpp builds it with
// g++ -fsanitize=thread.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <thread>

namespace gnn {
namespace host {

inline void rows_to_f32_plain(const double *src, float *dst, size_t n) {
    for (size_t i = 0; i < n; i++) dst[i] = (float)src[i];
}
// (the x86 specifics -- an AVX2 clone of the loop, the spin-wait hint -- sit behind the architecture test: the host side of the
//  library builds on any host the HIP toolchain targets)
#if defined(__x86_64__)
__attribute__((target("avx2"))) inline void rows_to_f32_avx2(const double *src, float *dst, size_t n) {
    for (size_t i = 0; i < n; i++) dst[i] = (float)src[i]; // (vectorised by the compiler: vcvtpd2ps)
}
inline void rows_to_f32(const double *src, float *dst, size_t n) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) rows_to_f32_avx2(src, dst, n);
    else rows_to_f32_plain(src, dst, n);
}
inline void cpu_relax() { __builtin_ia32_pause(); }
#else
inline void rows_to_f32(const double *src, float *dst, size_t n) { rows_to_f32_plain(src, dst, n); }
inline void cpu_relax() { std::atomic_signal_fence(std::memory_order_seq_cst); }
#endif

// The conversion is the call's largest host cost (0.8 MB read, 0.4 MB written per 128-row batch of 784 inputs: ~20 us on
// one core).  Up to three helper threads per process (GNN_MLP_CONVERT_THREADS = 0..3 overrides; one on small hosts) share
// a large batch with the calling thread.  Every thread, the caller included, owns one contiguous share (the same part of
// the batch call after call: its lines stay in that core's cache) and claims it piece by piece (kPiece values); a thread
// that has finished its own share claims pieces of the others' -- so the caller never waits for a helper that has not
// started (it converts that share itself), only for pieces already in other hands.  A caller that steps in a loop
// (NeuralNetTrainer.java:83) comes back within microseconds, so a helper polls for the next batch for a bounded time
// (~0.5 ms) before it goes to sleep on the condition variable: inside a training loop the hand-over costs no wake-up,
// outside one the helpers sleep.  If no thread can be created the caller converts alone.
class ConvertHelper {
  public:
    static constexpr int kMaxHelpers = 3;
    static constexpr size_t kPiece = 2048;
    ~ConvertHelper() {
        { std::lock_guard<std::mutex> lk(mu_); quit_.store(true); }
        cv_.notify_all();
        for (auto &th : th_) if (th.joinable()) th.join();
    }
    // converts [src, src + n) into dst; returns when ALL of it is done
    void run(const double *src, float *dst, size_t n) {
        std::unique_lock<std::mutex> call(call_mu_, std::try_to_lock); // one batch at a time; a second caller converts alone
        if (n < (size_t)1 << 15 || !call.owns_lock() || !start()) { rows_to_f32(src, dst, n); return; }
        const uint64_t job = seq_.load(std::memory_order_relaxed) + 1;
        const uint32_t pieces = (uint32_t)((n + kPiece - 1) / kPiece), threads = (uint32_t)n_helpers_ + 1;
        src_ = src; dst_ = dst; n_ = n; pieces_ = pieces;
        completed_.v.store(0, std::memory_order_relaxed);
        for (uint32_t i = 0; i < threads; i++) { // share i = pieces [i * pieces / threads, (i + 1) * pieces / threads)
            share_[i].end.store((uint32_t)((uint64_t)(i + 1) * pieces / threads), std::memory_order_relaxed);
            share_[i].next.store(job << 32 | (uint32_t)((uint64_t)i * pieces / threads)); // (an older job's claims now fail)
        }
        seq_.store(job);                      // (seq_cst: ordered against the sleepers' count below)
        if (sleepers_.load() > 0) {
            { std::lock_guard<std::mutex> lk(mu_); }
            cv_.notify_all();
        }
        claim_pieces(job, 0);
        // pieces in other hands: ~1 us each.  A helper that the scheduler took off its core mid-piece (a busy host: a JVM's own
        // threads) would keep a spinning caller waiting for a time slice, so after a short spin the caller yields its core
        for (unsigned spins = 0; completed_.v.load(std::memory_order_acquire) != pieces; spins++) {
            if (spins < kSpinsBeforeYield) cpu_relax();
            else std::this_thread::yield();
        }
    }

  private:
    // next = job << 32 | next piece of the share; end is atomic because a helper still leaving the previous job may read it
    // while the caller sets up the next one (its claim then fails on the job tag)
    struct alignas(64) Share { std::atomic<uint64_t> next{0}; std::atomic<uint32_t> end{0}; };
    struct alignas(64) Count { std::atomic<uint32_t> v{0}; };
    // a claim succeeds only while the share's counter still belongs to `job`, and the job's fields stay as they are until
    // every claimed piece is counted in completed_
    void claim_pieces(uint64_t job, int me) {
        const int threads = n_helpers_ + 1;
        for (int k = 0; k < threads; k++) {
            Share &sh = share_[(me + k) % threads];
            uint64_t v = sh.next.load();
            for (;;) {
                if ((v >> 32) != (job & 0xffffffffu)) return; // the job is over (and another may have begun)
                if ((uint32_t)v >= sh.end.load(std::memory_order_relaxed)) break;
                if (!sh.next.compare_exchange_weak(v, v + 1)) continue;
                const size_t begin = (size_t)(uint32_t)v * kPiece, len = (n_ - begin < kPiece) ? n_ - begin : kPiece;
                rows_to_f32(src_ + begin, dst_ + begin, len);
                completed_.v.fetch_add(1, std::memory_order_release);
                v = sh.next.load();
            }
        }
    }
    void work(int me) {
        uint64_t seen = 0;
        for (;;) {
            using clock = std::chrono::steady_clock;
            uint64_t s;
            clock::time_point t0 = clock::now();
            unsigned polls = 0;
            while ((s = seq_.load()) == seen && !quit_.load(std::memory_order_relaxed)) {
                // (the clock is read every 64th poll: ~25 ns per reading against ~40 ns per pause)
                if ((++polls & 63) != 0 || clock::now() - t0 < kPollWindow) { cpu_relax(); continue; }
                std::unique_lock<std::mutex> lk(mu_);
                sleepers_.fetch_add(1);
                cv_.wait(lk, [&] { return seq_.load() != seen || quit_.load(); });
                sleepers_.fetch_sub(1);
                t0 = clock::now();
            }
            if (quit_.load()) return;
            seen = s;
            claim_pieces(s, me);
        }
    }
    bool start() {
        if (started_) return n_helpers_ > 0;
        started_ = true;
        int want = (int)std::thread::hardware_concurrency() >= 8 ? kMaxHelpers : 1;
        if (const char *e = getenv("GNN_MLP_CONVERT_THREADS")) want = atoi(e);
        if (want > kMaxHelpers) want = kMaxHelpers;
        int made = 0;
        for (int i = 0; i < want; i++) {
            try {
                // (a helper reads n_helpers_ only after it has seen a job, and jobs are published after start() returns)
                th_[i] = std::thread([this, i]() { work(i + 1); });
                made = i + 1;
            } catch (...) {
                break;
            }
        }
        n_helpers_ = made;
        return made > 0;
    }
    // how long a helper polls for the next batch before it sleeps: several periods of the slowest loop that gains from helpers
    // (a window near the period itself lets the helpers fall asleep before every call) -- wall-clock time, whatever the host's clock rate
    static constexpr std::chrono::microseconds kPollWindow{500};
    static constexpr unsigned kSpinsBeforeYield = 4096; // ~150 us of pauses: twenty pieces' worth
    std::mutex mu_, call_mu_;
    std::condition_variable cv_;
    std::thread th_[kMaxHelpers];
    bool started_ = false;
    int n_helpers_ = 0;
    std::atomic<bool> quit_{false};
    std::atomic<int> sleepers_{0};
    const double *src_ = nullptr; float *dst_ = nullptr; size_t n_ = 0; uint32_t pieces_ = 0;
    alignas(64) std::atomic<uint64_t> seq_{0};
    Share share_[kMaxHelpers + 1];
    Count completed_;
};

} // namespace host
} // namespace gnn
