// Developer microbenchmark: the bf16 x 6 NT loop of gemm_nt.hip (BK = 16: one MFMA k group per stage, 128 x 256 tile,
// 8 waves of 2x2 blocks) with the fragments of stage s+1 read from LDS and split WHILE the 24 MFMAs of stage s run
// (register double buffering).  In the plain loop every wave of the workgroup reads + splits right after the barrier
// and multiplies afterwards, in lock step, so the matrix pipe idles through every split phase.
// MODE 0: plain loop.  MODE 1: fragments one stage ahead.  MODE 2: + sched_group_barrier interleave (1 MFMA : 7 VALU).
// hipcc --offload-arch=gfx950 -O3 nt_ahead.hip -o /tmp/nt_ahead
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define BK 16
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void dma16(const void* g, unsigned l) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(l) : "memory");
}
struct Split { bf16x8 p0, p1, p2; };
__device__ __forceinline__ Split split8t(const float4& lo, const float4& hi) {
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    unsigned u0[8], u1[8], u2[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u0[i] = __float_as_uint(v[i]);
        const float r1 = v[i] - __uint_as_float(u0[i] & 0xffff0000u);
        u1[i] = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(u1[i] & 0xffff0000u);
        u2[i] = __float_as_uint(r2);
    }
    u32x4 q0, q1, q2;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        q0[j] = __builtin_amdgcn_perm(u0[2 * j + 1], u0[2 * j], 0x07060302u);
        q1[j] = __builtin_amdgcn_perm(u1[2 * j + 1], u1[2 * j], 0x07060302u);
        q2[j] = __builtin_amdgcn_perm(u2[2 * j + 1], u2[2 * j], 0x07060302u);
    }
    Split s;
    s.p0 = __builtin_bit_cast(bf16x8, q0); s.p1 = __builtin_bit_cast(bf16x8, q1); s.p2 = __builtin_bit_cast(bf16x8, q2);
    return s;
}
#define MF(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define PRODUCTS(F) \
    _Pragma("unroll") for (int j = 0; j < 2; j++) { \
        MF(F[0].p2, F[2 + j].p0, acc[0][j]); MF(F[1].p2, F[2 + j].p0, acc[1][j]); \
        MF(F[0].p0, F[2 + j].p2, acc[0][j]); MF(F[1].p0, F[2 + j].p2, acc[1][j]); \
        MF(F[0].p1, F[2 + j].p1, acc[0][j]); MF(F[1].p1, F[2 + j].p1, acc[1][j]); \
        MF(F[0].p1, F[2 + j].p0, acc[0][j]); MF(F[1].p1, F[2 + j].p0, acc[1][j]); \
        MF(F[0].p0, F[2 + j].p1, acc[0][j]); MF(F[1].p0, F[2 + j].p1, acc[1][j]); \
        MF(F[0].p0, F[2 + j].p0, acc[0][j]); MF(F[1].p0, F[2 + j].p0, acc[1][j]); \
    }

template <int MODE>
__global__ __launch_bounds__(512, 2) void kern(const float* A, const float* W, float* C, int K, unsigned long long* clk) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    constexpr int STAGE = 384 * BK;                          // floats; rows of 64 B, 4 chunks, swizzle (row >> 2) & 3
    constexpr int NST = MODE == 3 ? 4 : 2;
    __shared__ __attribute__((aligned(1024))) float stages[NST * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w & 1, wn = w >> 1;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)&stages[0];
    const float* src[3];                                      // 24 instructions of 16 rows; wave w issues w, w+8, w+16
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int row = (w + 8 * j) * 16 + (lane >> 2);
        src[j] = (row < 128 ? A + ((size_t)(MODE == 4 ? blockIdx.x & 7 : blockIdx.x) * 128 + row) * K : W + (size_t)(row - 128) * K) + (((lane & 3) ^ ((row >> 2) & 3)) << 2);
    }
    auto issue = [&](int st, int k0) {
#pragma unroll
        for (int j = 0; j < 3; j++) dma16(src[j] + k0, lds0 + st * (STAGE * 4) + (w + 8 * j) * 1024);
    };
    const int fr = lane & 31, fh = lane >> 5;
    int foff[4];                                              // float offsets of this lane's first chunk: A blocks 0,1, W blocks 0,1
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const int row = (b < 2 ? wm * 64 + 32 * b : 128 + wn * 64 + 32 * (b - 2)) + fr;
        foff[b] = row * BK + (((2 * fh) ^ ((row >> 2) & 3)) << 2);
    }
    auto fetch = [&](const float* st, Split* f) {
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const float4 lo = *(const float4*)&st[foff[b]], hi = *(const float4*)&st[foff[b] ^ 4];
            f[b] = split8t(lo, hi);
        }
    };
    f32x16 acc[2][2] = {};
    const int S = K / BK;
    issue(0, 0);
    if (MODE == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int s = 0; s < S; s++) {
            if (s + 1 < S) issue((s + 1) & 1, (s + 1) * BK);
            Split f[4];
            fetch(stages + (s & 1) * STAGE, f);
            PRODUCTS(f)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else if (MODE == 3) {
        // ring of 4 stages: DMA three stages ahead of the MFMAs, fragments one stage ahead
        issue(1, BK); issue(2, 2 * BK);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __syncthreads();
        Split f[4];
        fetch(stages, f);
        for (int s = 0; s < S; s++) {
            // stage s+1 must have landed: stages s+1, s+2 (and not yet s+3) are in flight
            if (s + 2 < S) { asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); } else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            __syncthreads();
            if (s + 3 < S) issue((s + 3) & 3, (s + 3) * BK);  // slot of stage s-1: read during step s-2
            Split n[4];
            fetch(stages + ((s + 1) & 3) * STAGE, n);
            PRODUCTS(f)
#pragma unroll
            for (int b = 0; b < 4; b++) f[b] = n[b];
        }
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        Split f[4];
        fetch(stages, f);
        if (S > 1) issue(1, BK);
        for (int s = 0; s < S; s++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                  // stage s+1 landed; every wave holds stage s in registers
            if (s + 2 < S) issue(s & 1, (s + 2) * BK);        // slot of stage s: free
            Split n[4];
            fetch(stages + ((s + 1) & 1) * STAGE, n);      // unconditional (one basic block); the last one reads a stale stage
            PRODUCTS(f)
            if (MODE == 2) {
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);        // the 8 ds_read_b128 first
#pragma unroll
                for (int i = 0; i < 24; i++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);    // 8 VALU
                }
            }
#pragma unroll
            for (int b = 0; b < 4; b++) f[b] = n[b];
        }
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = blockIdx.x * 128 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                C[(size_t)row * 256 + wn * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
            }
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

template <int MODE>
float run(const float* A, const float* W, float* C, int NB, int K) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    static unsigned long long* clk = nullptr;
    if (!clk) (void)hipMalloc(&clk, 16);
    for (int i = 0; i < 60; i++) kern<MODE><<<NB, 512>>>(A, W, C, K, nullptr);
    (void)hipEventRecord(e0); for (int i = 0; i < 30; i++) kern<MODE><<<NB, 512>>>(A, W, C, K, clk); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc[2]; (void)hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    printf("[mode %d: shader clock %.0f MHz, workgroup 0 ran %.1f us] ", MODE, 100.0 * hc[0] / hc[1], hc[1] / 100.0);
    return ms / 30 * 1e3;
}

int main() {
    const int K = 2048, NB = 256, M = NB * 128;
    float *A, *W, *C[5];
    (void)hipMalloc(&A, (size_t)M * K * 4); (void)hipMalloc(&W, 256 * K * 4);
    for (int i = 0; i < 5; i++) (void)hipMalloc(&C[i], (size_t)M * 256 * 4);
    float* h = (float*)malloc((size_t)M * K * 4);
    srand(1);
    for (size_t i = 0; i < (size_t)M * K; i++) h[i] = ((float)rand() / RAND_MAX - 0.5f) * 2.0f;
    (void)hipMemcpy(A, h, (size_t)M * K * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(W, h + 31337, 256 * K * 4, hipMemcpyHostToDevice);
    const double fl = 2.0 * M * 256 * K;
    const float t0 = run<0>(A, W, C[0], NB, K), t1 = run<1>(A, W, C[1], NB, K), t2 = run<2>(A, W, C[2], NB, K);
    printf("plain loop                    : %8.1f us  %6.1f TFLOP/s fp32-equivalent (%s)\n", t0, fl / t0 / 1e6, hipGetErrorString(hipGetLastError()));
    printf("fragments one stage ahead     : %8.1f us  %6.1f\n", t1, fl / t1 / 1e6);
    printf("  + sched_group_barrier 1 : 8 : %8.1f us  %6.1f\n", t2, fl / t2 / 1e6);
    const float t3 = run<3>(A, W, C[3], NB, K), t4 = run<4>(A, W, C[4], NB, K);
    printf("ahead + 4-stage DMA ring      : %8.1f us  %6.1f\n", t3, fl / t3 / 1e6);
    printf("ahead, A rows L2-resident     : %8.1f us  %6.1f (diagnostic)\n", t4, fl / t4 / 1e6);
    const size_t n = (size_t)64 * 256;
    const float* hw = h + 31337;
    for (int m = 0; m < 4; m++) {
        float* c = (float*)malloc(n * 4);
        (void)hipMemcpy(c, C[m], n * 4, hipMemcpyDeviceToHost);
        double e = 0;
        for (int r = 0; r < 64; r++)
            for (int cc = 0; cc < 256; cc++) {
                double s = 0;
                for (int k = 0; k < K; k++) s += (double)h[(size_t)r * K + k] * (double)hw[(size_t)cc * K + k];
                e = fmax(e, fabs(c[r * 256 + cc] - s));
            }
        printf("mode %d max |error| vs fp64 %.3g\n", m, e);
    }
    return 0;
}
