// Developer microbenchmark: steady-state NT main loop WITH operand staging, two ways:
//   k3: global -> VGPR -> ds_write_b128 into a padded (stride 36) stage      (what gemm_nt.hip does)
//   k4: global_load_lds_dwordx4 straight into an XOR-swizzled unpadded stage (no VGPR hop, no ds_write)
// hipcc --offload-arch=gfx950 -O3 nt_staging.hip -o /tmp/nt_staging
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define mfma(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
#define BK 32

#define MFMA_BLOCK(a0, a1, b0, b1)                                                                   \
    {                                                                                                \
        const float x0[4] = {a0.x, a0.y, a0.z, a0.w}, x1[4] = {a1.x, a1.y, a1.z, a1.w};              \
        const float y0[4] = {b0.x, b0.y, b0.z, b0.w}, y1[4] = {b1.x, b1.y, b1.z, b1.w};              \
        _Pragma("unroll") for (int r = 0; r < 4; r++) {                                              \
            acc[0][0] = mfma(x0[r], y0[r], acc[0][0]); acc[0][1] = mfma(x0[r], y1[r], acc[0][1]);    \
            acc[1][0] = mfma(x1[r], y0[r], acc[1][0]); acc[1][1] = mfma(x1[r], y1[r], acc[1][1]);    \
        }                                                                                            \
    }

__global__ __launch_bounds__(512, 2) void k3(const float* A, const float* W, float* out, int K) {
    constexpr int LD = BK + 4;
    __shared__ __attribute__((aligned(16))) float smem[2][(128 + 256) * LD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w & 1, wn = w >> 1;
    const float* Ab = A + (size_t)blockIdx.x * 128 * K;
    f32x16 acc[2][2] = {};
    const int frag = (lane & 31) * LD + 4 * (lane >> 5);
    const int lr = tid >> 3, lc = (tid & 7) * 4;
    float4 ra[2], rw[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; i++) ra[i] = *(const float4*)(Ab + (size_t)(lr + 64 * i) * K + k0 + lc);
#pragma unroll
        for (int i = 0; i < 4; i++) rw[i] = *(const float4*)(W + (size_t)(lr + 64 * i) * K + k0 + lc);
    };
    auto lstore = [&](int st) {
#pragma unroll
        for (int i = 0; i < 2; i++) *(float4*)&smem[st][(lr + 64 * i) * LD + lc] = ra[i];
#pragma unroll
        for (int i = 0; i < 4; i++) *(float4*)&smem[st][(128 + lr + 64 * i) * LD + lc] = rw[i];
    };
    const int S = K / BK;
    gload(0); lstore(0); __syncthreads();
    for (int s = 0; s < S; s++) {
        if (s + 1 < S) gload((s + 1) * BK);
        const float* As = &smem[s & 1][wm * 64 * LD + frag];
        const float* Ws = &smem[s & 1][(128 + wn * 64) * LD + frag];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const float4 a0 = *(const float4*)(As + kk * 8), a1 = *(const float4*)(As + 32 * LD + kk * 8);
            const float4 b0 = *(const float4*)(Ws + kk * 8), b1 = *(const float4*)(Ws + 32 * LD + kk * 8);
            MFMA_BLOCK(a0, a1, b0, b1)
        }
        if (s + 1 < S) lstore((s + 1) & 1);
        __syncthreads();
    }
    float t = 0;
    for (int r = 0; r < 16; r++) t += acc[0][0][r] + acc[0][1][r] + acc[1][0][r] + acc[1][1][r];
    out[blockIdx.x * 512 + tid] = t;
}

// stage layout: row-major, 32 floats per row (8 chunks of 16 B), chunk c of row r stored at slot c ^ ((r >> 1) & 7)
__global__ __launch_bounds__(512, 2) void k4(const float* A, const float* W, float* out, int K, int shared_a) {
    __shared__ __attribute__((aligned(1024))) float smem[2][(128 + 256) * BK];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w & 1, wn = w >> 1;
    const float* Ab = A + (shared_a ? 0 : (size_t)blockIdx.x * 128 * K);
    f32x16 acc[2][2] = {};
    // direct loads: wave w covers stage rows {8*(w + 8*j) .. +8}, j = 0..5 (48 row-groups of 8: 16 of A then 32 of W)
    const int sub = lane >> 3, slot = lane & 7;
    const float* src[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const int row = 8 * (w + 8 * j) + sub;             // stage row 0..383
        const int chunk = slot ^ ((row >> 1) & 7);
        src[j] = (row < 128 ? Ab + (size_t)row * K : W + (size_t)(row - 128) * K) + chunk * 4;
    }
    auto gload = [&](int st, int k0) {
#pragma unroll
        for (int j = 0; j < 6; j++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + k0),
                                             (__attribute__((address_space(3))) void*)&smem[st][8 * (w + 8 * j) * BK],
                                             16, 0, 0);
    };
    const int fr = lane & 31, fh = lane >> 5;
    const int S = K / BK;
    gload(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s = 0; s < S; s++) {
        if (s + 1 < S) gload((s + 1) & 1, (s + 1) * BK);
        const float* st = smem[s & 1];
        const int ra0 = wm * 64 + fr, ra1 = ra0 + 32, rb0 = 128 + wn * 64 + fr, rb1 = rb0 + 32;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const int c = kk * 2 + fh;
            const float4 a0 = *(const float4*)(st + ra0 * BK + ((c ^ ((ra0 >> 1) & 7)) << 2));
            const float4 a1 = *(const float4*)(st + ra1 * BK + ((c ^ ((ra1 >> 1) & 7)) << 2));
            const float4 b0 = *(const float4*)(st + rb0 * BK + ((c ^ ((rb0 >> 1) & 7)) << 2));
            const float4 b1 = *(const float4*)(st + rb1 * BK + ((c ^ ((rb1 >> 1) & 7)) << 2));
            MFMA_BLOCK(a0, a1, b0, b1)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    float t = 0;
    for (int r = 0; r < 16; r++) t += acc[0][0][r] + acc[0][1][r] + acc[1][0][r] + acc[1][1][r];
    out[blockIdx.x * 512 + tid] = t;
}

// k5: as k4 but the two stages are SEPARATE __shared__ objects and the loop is unrolled by two, so the
// compiler's LDS-DMA alias tracking can tell "reading stage X" from "DMA into stage Y" (no vmcnt(0) after issue)
template <typename SA, typename SB>
__device__ __forceinline__ void k5_step(SA& cur, SB& nxt, const float* const (&src)[6], int w, int k_next, bool more,
                                        f32x16 (&acc)[2][2], int ra0, int rb0, int fh) {
    if (more) {
#pragma unroll
        for (int j = 0; j < 6; j++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + k_next),
                                             (__attribute__((address_space(3))) void*)&nxt[8 * (w + 8 * j) * BK], 16, 0, 0);
    }
    const int ra1 = ra0 + 32, rb1 = rb0 + 32;
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        const int c = kk * 2 + fh;
        const float4 a0 = *(const float4*)(&cur[ra0 * BK + ((c ^ ((ra0 >> 1) & 7)) << 2)]);
        const float4 a1 = *(const float4*)(&cur[ra1 * BK + ((c ^ ((ra1 >> 1) & 7)) << 2)]);
        const float4 b0 = *(const float4*)(&cur[rb0 * BK + ((c ^ ((rb0 >> 1) & 7)) << 2)]);
        const float4 b1 = *(const float4*)(&cur[rb1 * BK + ((c ^ ((rb1 >> 1) & 7)) << 2)]);
        MFMA_BLOCK(a0, a1, b0, b1)
    }
    __syncthreads();
}

__global__ __launch_bounds__(512, 2) void k5(const float* A, const float* W, float* out, int K, int shared_a) {
    __shared__ __attribute__((aligned(1024))) float st0[(128 + 256) * BK];
    __shared__ __attribute__((aligned(1024))) float st1[(128 + 256) * BK];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w & 1, wn = w >> 1;
    const float* Ab = A + (shared_a ? 0 : (size_t)blockIdx.x * 128 * K);
    f32x16 acc[2][2] = {};
    const int sub = lane >> 3, slot = lane & 7;
    const float* src[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const int row = 8 * (w + 8 * j) + sub;
        const int chunk = slot ^ ((row >> 1) & 7);
        src[j] = (row < 128 ? Ab + (size_t)row * K : W + (size_t)(row - 128) * K) + chunk * 4;
    }
    const int fr = lane & 31, fh = lane >> 5;
    const int S = K / BK;
#pragma unroll
    for (int j = 0; j < 6; j++)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j]),
                                         (__attribute__((address_space(3))) void*)&st0[8 * (w + 8 * j) * BK], 16, 0, 0);
    __syncthreads();
    const int ra0 = wm * 64 + fr, rb0 = 128 + wn * 64 + fr;
    for (int s = 0; s < S; s += 2) {
        k5_step(st0, st1, src, w, (s + 1) * BK, s + 1 < S, acc, ra0, rb0, fh);
        k5_step(st1, st0, src, w, (s + 2) * BK, s + 2 < S, acc, ra0, rb0, fh);
    }
    float t = 0;
    for (int r = 0; r < 16; r++) t += acc[0][0][r] + acc[0][1][r] + acc[1][0][r] + acc[1][1][r];
    out[blockIdx.x * 512 + tid] = t;
}

int main() {
    const int K = 2048, NB = 512, M = NB * 128;
    float *A, *W, *o3, *o4;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, 256 * K * 4); hipMalloc(&o3, NB * 512 * 4); hipMalloc(&o4, NB * 512 * 4);
    float* h = (float*)malloc((size_t)M * K * 4);
    for (size_t i = 0; i < (size_t)M * K; i++) h[i] = (float)((i * 2654435761u) % 1009) * 1e-4f - 0.05f;
    hipMemcpy(A, h, (size_t)M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, h + 12345, 256 * K * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int rep = 2; rep >= 0; rep--) {
        for (int i = 0; i < 100; i++) k3<<<NB, 512>>>(A, W, o3, K); hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 50; i++) k3<<<NB, 512>>>(A, W, o3, K); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("k3 register staging      : %.1f us  %.1f TFLOP/s\n", ms / 50 * 1e3, 2.0 * M * 256 * K / (ms / 50) / 1e9);
        for (int i = 0; i < 100; i++) k4<<<NB, 512>>>(A, W, o4, K, rep); hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 50; i++) k4<<<NB, 512>>>(A, W, o4, K, rep); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("k4 direct-to-LDS (shared_a=%d): %.1f us  %.1f TFLOP/s  (%s)\n", rep, ms / 50 * 1e3, 2.0 * M * 256 * K / (ms / 50) / 1e9, hipGetErrorString(hipGetLastError()));
    }
    for (int rep = 1; rep >= 0; rep--) {
        for (int i = 0; i < 100; i++) k5<<<NB, 512>>>(A, W, o4, K, rep); hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 50; i++) k5<<<NB, 512>>>(A, W, o4, K, rep); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("k5 direct, split stages (shared_a=%d): %.1f us  %.1f TFLOP/s  (%s)\n", rep, ms / 50 * 1e3, 2.0 * M * 256 * K / (ms / 50) / 1e9, hipGetErrorString(hipGetLastError()));
    }
    float *h3 = (float*)malloc(NB * 512 * 4), *h4 = (float*)malloc(NB * 512 * 4);
    hipMemcpy(h3, o3, NB * 512 * 4, hipMemcpyDeviceToHost); hipMemcpy(h4, o4, NB * 512 * 4, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < NB * 512; i++) if (h3[i] != h4[i]) bad++;
    printf("mismatches k3 vs k4: %d  (sample %g %g)\n", bad, h3[777], h4[777]);
    return 0;
}
