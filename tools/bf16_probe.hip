// bf16_probe.hip -- would bf16-STORED operands (converted to f32 in registers, same f32 MFMA code)
// shorten fwd_first?  Times the f32 kernel against a copy that loads 2-byte operands.
#include "../graph-neural-net_amd/csrc/fused_kernels.h"
#include <cstdio>
#include <vector>
using namespace gnn;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)

__device__ __forceinline__ float bf2f(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }

template <int NW>
__global__ __launch_bounds__(NW * 64) void fwd_first_bf16(const unsigned short *A, int lda, const unsigned short *W, int ldw, float *C, int ldc,
                                                          int K, XcdTiling tiling) {
    constexpr int MAXC = 8, RLD = 20;
    __shared__ __attribute__((aligned(16))) float red[NW * 16 * RLD];
    int tm, tn;
    if (!tiling.tile_of(blockIdx.x, tm, tn)) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, fr = lane & 15, fq = lane >> 4;
    const int n0 = tn * 16, m0 = tm * 16, k16 = K / 16;
    const int c_begin = wave * k16 / NW, c_end = (wave + 1) * k16 / NW;
    const unsigned short *arow = A + (size_t)(m0 + fr) * lda + 4 * fq;
    const unsigned short *wcol = W + (size_t)(4 * fq) * ldw + n0 + fr;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    uint2 a[MAXC]; unsigned short b[MAXC][4];
#pragma unroll
    for (int i = 0; i < MAXC; i++) {
        const int c = c_begin + i;
        if (c < c_end) {
            a[i] = *reinterpret_cast<const uint2 *>(arow + c * 16);
            const unsigned short *w = wcol + (size_t)(c * 16) * ldw;
            b[i][0] = w[0]; b[i][1] = w[ldw]; b[i][2] = w[2 * ldw]; b[i][3] = w[3 * ldw];
        } else { a[i] = make_uint2(0, 0); b[i][0] = b[i][1] = b[i][2] = b[i][3] = 0; }
    }
#pragma unroll
    for (int i = 0; i < MAXC; i++) {
        const float ax = __builtin_bit_cast(float, a[i].x << 16), ay = __builtin_bit_cast(float, a[i].x & 0xffff0000u);
        const float az = __builtin_bit_cast(float, a[i].y << 16), aw = __builtin_bit_cast(float, a[i].y & 0xffff0000u);
        f32x4 &acc = (i & 1) ? acc1 : acc0;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ax, bf2f(b[i][0]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ay, bf2f(b[i][1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(az, bf2f(b[i][2]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, bf2f(b[i][3]), acc, 0, 0, 0);
    }
    const f32x4 acc = acc0 + acc1;
#pragma unroll
    for (int r = 0; r < 4; r++) red[(wave * 16 + fq * 4 + r) * RLD + fr] = acc[r];
    __syncthreads();
    if (t < 64) {
        const int m = t >> 2, q = t & 3;
        f32x4 s = {0, 0, 0, 0};
#pragma unroll
        for (int w = 0; w < NW; w++) s += *reinterpret_cast<const f32x4 *>(&red[(w * 16 + m) * RLD + q * 4]);
        *reinterpret_cast<f32x4 *>(C + (size_t)(m0 + m) * ldc + n0 + q * 4) = s;
    }
}

int main() {
    const int M = 128, K = 784, N = 304;
    float *A, *W, *C; unsigned short *Ab, *Wb;
    CK(hipMalloc(&A, M * K * 4)); CK(hipMalloc(&W, K * N * 4)); CK(hipMalloc(&C, M * N * 4));
    CK(hipMalloc(&Ab, M * K * 2)); CK(hipMalloc(&Wb, K * N * 2));
    CK(hipMemset(A, 0, M * K * 4)); CK(hipMemset(W, 0, K * N * 4)); CK(hipMemset(Ab, 0, M * K * 2)); CK(hipMemset(Wb, 0, K * N * 2));
    FwdFirstParams f{}; f.A = A; f.lda = K; f.W = W; f.ldw = N; f.C = C; f.ldc = N; f.M = M; f.N = N; f.K = K; f.m_true = M; f.n_true = 300; f.act = 0; f.apply_act = 1;
    f.tiling = make_xcd_tiling(M / 16, N / 16);
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        for (int v = 0; v < 2; v++) {
            for (int i = 0; i < 50; i++) { if (v) hipLaunchKernelGGL((fwd_first_bf16<8>), dim3(f.tiling.blocks()), dim3(512), 0, s, Ab, K, Wb, N, C, N, K, f.tiling); else hipLaunchKernelGGL((fwd_first_kernel<8, false, 0>), dim3(f.tiling.blocks()), dim3(512), 0, s, f); }
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < 500; i++) { if (v) hipLaunchKernelGGL((fwd_first_bf16<8>), dim3(f.tiling.blocks()), dim3(512), 0, s, Ab, K, Wb, N, C, N, K, f.tiling); else hipLaunchKernelGGL((fwd_first_kernel<8, false, 0>), dim3(f.tiling.blocks()), dim3(512), 0, s, f); }
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s: %.2f us per call\n", v ? "fwd_first, bf16-stored operands" : "fwd_first, f32 operands        ", ms * 2);
        }
    }
    return 0;
}
