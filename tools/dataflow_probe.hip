// dataflow_probe.hip -- diagnostic: can the three kernels of a step be linked by device-side
// counters instead of kernel boundaries?  Three kernels per step (grids 152 / 32 / 294 like the
// real step) on three streams; each waits (bounded spin, one lane) until its producer's "done"
// counter reaches the step's target, reads what the producer wrote -- on other XCDs --, writes its
// own buffer, and releases its counter.  Compared with the same kernels serialised on ONE stream.
// Checks that the values really propagate (visibility across XCDs) and prints us/step.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)

constexpr unsigned SPIN_LIMIT = 1u << 22; // ~ seconds: a safety net, every wave exits

template <int SLEEP> __device__ inline void wait_ge(unsigned *ctr, unsigned target, unsigned *err) {
    if (threadIdx.x == 0) {
        unsigned spins = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? SPIN_LIMIT : 0; // after a timeout nobody waits
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(SLEEP);
            if (++spins > SPIN_LIMIT) { atomicAdd(err, 1u); break; }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
__device__ inline void signal(unsigned *ctr) {
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// Variant without fences: data moves with agent-scope RELAXED atomic loads/stores (cache-policy bits
// on plain loads/stores: they go past the XCD's non-coherent L2 lines), flags are relaxed too, and
// ordering comes from s_waitcnt + the workgroup barrier.
template <int SLEEP>
__global__ __launch_bounds__(256) void stage_sc(const float *in, float *out, int n, unsigned *wait_ctr, unsigned target,
                                                unsigned *done_ctr, unsigned *err) {
    if (threadIdx.x == 0) {
        unsigned spins = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? SPIN_LIMIT : 0;
        while (__hip_atomic_load(wait_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(SLEEP);
            if (++spins > SPIN_LIMIT) { atomicAdd(err, 1u); break; }
        }
    }
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float v = __hip_atomic_load(&in[(i * 7919) & (n - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&out[i], v + 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_s_waitcnt(0); // every store of this wave acknowledged
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(done_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// out[i] = in[(i * 7919) % n] + 1 : every workgroup reads what OTHER workgroups wrote
template <bool FLOW, int SLEEP>
__global__ __launch_bounds__(256) void stage(const float *in, float *out, int n, unsigned *wait_ctr, unsigned target,
                                             unsigned *done_ctr, unsigned *err) {
    if (FLOW) wait_ge<SLEEP>(wait_ctr, target, err);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = in[(i * 7919) & (n - 1)] + 1.f;
    if (FLOW) signal(done_ctr);
}

int main() {
    const int n = 1 << 18; // 1 MB per buffer
    const int g[3] = {152, 32, 294};
    float *X, *Y, *Z; unsigned *ctr;
    CK(hipMalloc(&X, n * 4)); CK(hipMalloc(&Y, n * 4)); CK(hipMalloc(&Z, n * 4)); CK(hipMalloc(&ctr, 64));
    hipStream_t s[3];
    for (auto &q : s) CK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    std::vector<float> host(n);
    const char *names[] = {"serial, 1 stream, no flags", "no flags, 3 streams (wrong results expected)", "flags, 1 stream", "flags, 3 streams, sleep 1",
                           "flags, 3 streams, sleep 16", "flags, 3 streams, sleep 100",
                           "sc1 data + relaxed flags, 1 stream", "sc1 data + relaxed flags, 3 streams, sleep 1", "sc1 data + relaxed flags, 3 streams, sleep 16"};
    for (int mode = 0; mode < 9; mode++) {
        for (int steps : {200, 2000}) {
            CK(hipMemset(X, 0, n * 4)); CK(hipMemset(Y, 0, n * 4)); CK(hipMemset(Z, 0, n * 4)); CK(hipMemset(ctr, 0, 64));
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int it = 0; it < steps; it++) {
                const unsigned t0_ = (unsigned)(it * g[2]), t1_ = (unsigned)((it + 1) * g[0]), t2_ = (unsigned)((it + 1) * g[1]);
#define STEP3(K, sa, sb, sc)                                                                                   \
    hipLaunchKernelGGL(K, dim3(g[0]), dim3(256), 0, sa, X, Y, n, ctr + 2, t0_, ctr + 0, ctr + 8);              \
    hipLaunchKernelGGL(K, dim3(g[1]), dim3(256), 0, sb, Y, Z, n, ctr + 0, t1_, ctr + 1, ctr + 8);              \
    hipLaunchKernelGGL(K, dim3(g[2]), dim3(256), 0, sc, Z, X, n, ctr + 1, t2_, ctr + 2, ctr + 8);
                switch (mode) {
                case 0: { STEP3((stage<false, 1>), s[0], s[0], s[0]) } break;
                case 1: { STEP3((stage<false, 1>), s[0], s[1], s[2]) } break;
                case 2: { STEP3((stage<true, 1>), s[0], s[0], s[0]) } break;
                case 3: { STEP3((stage<true, 1>), s[0], s[1], s[2]) } break;
                case 4: { STEP3((stage<true, 16>), s[0], s[1], s[2]) } break;
                case 5: { STEP3((stage<true, 100>), s[0], s[1], s[2]) } break;
                case 6: { STEP3((stage_sc<1>), s[0], s[0], s[0]) } break;
                case 7: { STEP3((stage_sc<1>), s[0], s[1], s[2]) } break;
                default: { STEP3((stage_sc<16>), s[0], s[1], s[2]) } break;
                }
            }
            CK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            CK(hipMemcpy(host.data(), X, n * 4, hipMemcpyDeviceToHost));
            unsigned c[16]; CK(hipMemcpy(c, ctr, 64, hipMemcpyDeviceToHost));
            int bad = 0;
            for (int i = 0; i < n; i++) bad += host[i] != 3.f * steps;
            printf("%-48s steps=%d: %.2f us/step, wrong values %d / %d, spin timeouts %u\n", names[mode], steps, us / steps, bad, n, c[8]);
            fflush(stdout);
            if (c[8]) { printf("spin timeouts: stopping\n"); return 2; }
        }
    }
    return 0;
}
