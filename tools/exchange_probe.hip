// exchange_probe.hip -- development harness (not shipped): what an exchange BETWEEN the workgroups of one launch costs.
// Round 4's structural experiment on rowblock_kernel: a row block of four batch rows shared by four
// workgroups, each owning a column slice of layer 2, the slices of A_2 (4 x 28 floats) exchanged through memory.  Before the
// kernel is rebuilt around it, the primitive alone: every workgroup of a cluster of four
//    writes its 448 bytes, publishes a flag, waits for its three peers' flags, reads their 448 bytes each,
// with per-workgroup cycle stamps, in three forms:
//    mode 0  agent-scope RELEASE on the flag store, ACQUIRE on the flag load (the memory model's answer for any placement:
//            on gfx942/gfx950 in multi-XCC mode this is buffer_wbl2 sc1 / buffer_inv sc1 around the accesses);
//    mode 1  every access a RELAXED agent-scope atomic (sc1: served by the XCD's L2, no write-back, no invalidate) --
//            correct only if the four workgroups share an XCD (one L2);
//    mode 2  as mode 0 with a device-wide __threadfence() instead of the release (the portable spelling).
// cluster c's workgroups: blockIdx = (c / 8) * 32 + slice * 8 + c % 8 -- the same blockIdx mod 8, i.e. the same XCD when the
// dispatcher deals workgroup i to XCD i mod 8 (tools/tile_probe measured that); `spread` = 1 uses blockIdx = 4 c + slice
// instead (four DIFFERENT XCDs): what the exchange costs across L2s, and whether mode 1 then still sees the data.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int SLICE_F = 112; // 4 rows x 28 columns
constexpr unsigned SPIN_LIMIT = 1u << 22;

struct Params {
    float *buf;              // [cluster][4 slices][112]
    unsigned *flags;         // [cluster][4]
    unsigned long long *stamps; // [wg][8]
    float *out;              // [wg]
    unsigned epoch;
    int spread, delay;
};

template <int MODE>
__global__ __launch_bounds__(256) void exchange_kernel(Params p) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int i = blockIdx.x;
    int c, s;
    if (p.spread) { c = i >> 2; s = i & 3; }
    else { c = (i >> 5) * 8 + (i & 7); s = (i >> 3) & 3; }
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    // `delay` > 0: the exchange happens MID-kernel, as it would in the row-block kernel -- the lines it will touch (flags, the peers'
    // slices, its own) are read once at the kernel's top (address translation and the L2 have seen them), then `delay` dependent
    // FMAs of other work (4 cycles each; + 25 % per slice index so that arrivals are a little staggered), then the stamps start.
    float *mine = p.buf + ((size_t)c * 4 + s) * SLICE_F;
    unsigned *flg = p.flags + (size_t)c * 4;
    float w = (float)t * 1e-3f;
    if (p.delay > 0) {
        if (t < 4) w += 1e-30f * (float)__hip_atomic_load(flg + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t >= 64 && t < 64 + 4 * 28) w += 1e-30f * p.buf[(size_t)c * 4 * SLICE_F + 4 * (t - 64)];
        for (int k = 0; k < p.delay + (p.delay / 4) * s; k++) w = __builtin_fmaf(w, 1.0001f, 1e-6f);
    }
    if (t == 0) t0 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
    if (wave == 0) {
        // ---- publish: 28 lanes x 16 B, then the flag
        if (MODE == 1) {
            for (int e = lane; e < SLICE_F; e += 64) __hip_atomic_store(mine + e, w + (float)(s * 1000 + e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0x0F70); // the wave's stores have left before the flag's store is issued (same queue, in order)
            if (lane == 0) __hip_atomic_store(flg + s, p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane < SLICE_F / 4) {
                f32x4 v;
                for (int j = 0; j < 4; j++) v[j] = w + (float)(s * 1000 + 4 * lane + j);
                *reinterpret_cast<f32x4 *>(mine + 4 * lane) = v;
            }
            if (MODE == 2) { __threadfence(); if (lane == 0) __hip_atomic_store(flg + s, p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            else if (lane == 0) __hip_atomic_store(flg + s, p.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            // (lane 0's release: s_waitcnt vmcnt(0) + buffer_wbl2 are wave-wide instructions -- they cover every lane's stores)
        }
        if (lane == 0) t1 = __builtin_amdgcn_s_memtime();
        // ---- wait for the three peers (lanes 0..2 poll one flag each; bounded: a peer that never comes ends the wait, out = -1)
        bool ok = true;
        if (lane < 3) {
            const int peer = lane + (lane >= s ? 1 : 0);
            unsigned spins = 0;
            while (__hip_atomic_load(flg + peer, MODE == 1 ? __ATOMIC_RELAXED : __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != p.epoch) {
                if (++spins > SPIN_LIMIT) { ok = false; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        ok = __all(ok);
        if (lane == 0) t2 = __builtin_amdgcn_s_memtime();
        // ---- read the peers' slices (84 float4: lanes 0..27 read one float4 of each peer)
        if (ok) { // (all three peers' slices requested before anything is used: one round trip)
            float va[3][2]; f32x4 vv[3];
            for (int q = 0; q < 3; q++) {
                const int peer = q + (q >= s ? 1 : 0);
                const float *theirs = p.buf + ((size_t)c * 4 + peer) * SLICE_F;
                if (MODE == 1) { for (int h = 0; h < 2; h++) va[q][h] = (lane + 64 * h < SLICE_F) ? __hip_atomic_load(theirs + lane + 64 * h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - w : 0.f; }
                else vv[q] = (lane < SLICE_F / 4) ? *reinterpret_cast<const f32x4 *>(theirs + 4 * lane) : (f32x4){w, w, w, w};
            }
            for (int q = 0; q < 3; q++) acc += MODE == 1 ? va[q][0] + va[q][1] : vv[q][0] + vv[q][1] + vv[q][2] + vv[q][3] - 4.f * w;
        } else acc = -1e30f;
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) t3 = __builtin_amdgcn_s_memtime();
    }
    if (t == 0) {
        p.out[i] = acc;
        unsigned long long *st = p.stamps + (size_t)i * 8;
        st[0] = t0; st[1] = t1; st[2] = t2; st[3] = t3;
        st[4] = (unsigned)__builtin_amdgcn_s_getreg(0xF814) & 15; // XCC id
        st[5] = (unsigned)__builtin_amdgcn_s_getreg(0xF804);      // HW_ID
    }
}

int main(int argc, char **argv) {
    const int clusters = argc > 1 ? atoi(argv[1]) : 32;
    const int n = clusters * 4;
    Params p{};
    CK(hipMalloc(&p.buf, (size_t)n * SLICE_F * 4)); CK(hipMemset(p.buf, 0, (size_t)n * SLICE_F * 4));
    CK(hipMalloc(&p.flags, (size_t)n * 4)); CK(hipMemset(p.flags, 0, (size_t)n * 4));
    CK(hipMalloc(&p.stamps, (size_t)n * 8 * 8));
    CK(hipMalloc(&p.out, (size_t)n * 4));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // expected sum for slice s: the three other slices' (1000 s' + e) over e = 0..111
    auto expect = [&](int sl) { double a = 0; for (int q = 0; q < 4; q++) if (q != sl) a += 1000.0 * q * SLICE_F + SLICE_F * (SLICE_F - 1) / 2.0; return a; };
    unsigned epoch = 0;
    auto run = [&](int mode, int spread, int delay, const char *name) {
        p.spread = spread; p.delay = delay;
        auto launch = [&]() {
            p.epoch = ++epoch;
            if (mode == 0) hipLaunchKernelGGL(exchange_kernel<0>, dim3(n), dim3(256), 0, s, p);
            else if (mode == 1) hipLaunchKernelGGL(exchange_kernel<1>, dim3(n), dim3(256), 0, s, p);
            else hipLaunchKernelGGL(exchange_kernel<2>, dim3(n), dim3(256), 0, s, p);
        };
        for (int it = 0; it < 20; it++) launch();
        CK(hipEventRecord(e0, s));
        const int reps = 500;
        for (int it = 0; it < reps; it++) launch();
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        launch(); CK(hipStreamSynchronize(s));
        std::vector<unsigned long long> st((size_t)n * 8); std::vector<float> out(n);
        CK(hipMemcpy(st.data(), p.stamps, st.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(out.data(), p.out, n * 4, hipMemcpyDeviceToHost));
        int wrong = 0, timeouts = 0, mixed = 0;
        std::vector<double> pub, wait, rd, total;
        unsigned long long first = ~0ull, last = 0;
        for (int i = 0; i < n; i++) {
            const int sl = spread ? (i & 3) : ((i >> 3) & 3);
            if (out[i] < -1e29f) timeouts++;
            else if (fabs(out[i] - expect(sl)) > 0.5 + 1e-4 * expect(sl)) wrong++;
            const unsigned long long *q = &st[(size_t)i * 8];
            pub.push_back((double)(q[1] - q[0])); wait.push_back((double)(q[2] - q[1])); rd.push_back((double)(q[3] - q[2])); total.push_back((double)(q[3] - q[0]));
            first = std::min(first, q[0]); last = std::max(last, q[3]);
        }
        for (int c = 0; c < clusters; c++) { // do the four workgroups of a cluster share an XCC?
            unsigned x[4];
            for (int sl = 0; sl < 4; sl++) { const int i = spread ? 4 * c + sl : (c / 8) * 32 + sl * 8 + c % 8; x[sl] = (unsigned)st[(size_t)i * 8 + 4]; }
            if (x[0] != x[1] || x[0] != x[2] || x[0] != x[3]) mixed++;
        }
        auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        auto mx = [](const std::vector<double> &v) { return *std::max_element(v.begin(), v.end()); };
        printf("%-46s %6.2f us/launch | stamps (s_memtime ticks, median / max): publish %5.0f / %5.0f  wait %5.0f / %5.0f  read %5.0f / %5.0f  total %5.0f / %5.0f | first start -> last end %llu | wrong %d timeouts %d | clusters over more than one XCC: %d of %d\n",
               name, ms * 1000.f / reps, med(pub), mx(pub), med(wait), mx(wait), med(rd), mx(rd), med(total), mx(total), last - first, wrong, timeouts, mixed, clusters);
    };
    run(0, 0, 0, "release/acquire, cluster in one XCD");
    run(2, 0, 0, "threadfence + acquire, cluster in one XCD");
    run(1, 0, 0, "relaxed sc1 only, cluster in one XCD");
    run(0, 0, 1000, "release/acquire, one XCD, mid-kernel (warm)");
    run(2, 0, 1000, "threadfence + acquire, one XCD, mid-kernel");
    run(1, 0, 1000, "relaxed sc1 only, one XCD, mid-kernel (warm)");
    run(1, 1, 1000, "relaxed sc1 only, 4 XCDs, mid-kernel (warm)");
    run(0, 1, 1000, "release/acquire, 4 XCDs, mid-kernel (warm)");
    run(0, 1, 0, "release/acquire, cluster over 4 XCDs");
    run(1, 1, 0, "relaxed sc1 only, cluster over 4 XCDs");
    // the same launch with no exchange at all would be ~2.5 us of launch cost: an empty-ish kernel for the base line
    return 0;
}
