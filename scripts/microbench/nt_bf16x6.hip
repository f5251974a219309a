// Developer microbenchmark: fp32-accurate NT product on the BF16 matrix cores.  Each fp32 operand value is split
// into three bf16 pieces (8 + 8 + 8 mantissa bits: a = a0 + a1 + a2 exactly up to the last piece's rounding) and
// the six significant cross products a0b0, a0b1, a1b0, a1b1, a0b2, a2b0 are accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16 (2.5 PFLOP/s dense against 157 TFLOP/s for v_mfma_f32_32x32x2_f32).
// Same staging as gemm_nt.hip (LDS-DMA, swizzled BK = 32 stage, 128 x 256 tile, 8 waves, one workgroup per CU);
// the split is done on the fragments in registers (worst case: no sharing of the split between waves).
// hipcc --offload-arch=gfx950 -O3 nt_bf16x6.hip -o /tmp/nt_bf16x6
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define BK 32
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void dma16(const float* g, unsigned l) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(l) : "memory");
}

struct Split { bf16x8 p0, p1, p2; };
// truncating variant (what the product kernels use: full-rate v_and / v_sub / v_perm only)
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
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
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
__device__ __forceinline__ Split split8(const float4& lo, const float4& hi) {
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    Split s;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const __bf16 a0 = (__bf16)v[i];
        const float r1 = v[i] - (float)a0;
        const __bf16 a1 = (__bf16)r1;
        const float r2 = r1 - (float)a1;
        s.p0[i] = a0; s.p1[i] = a1; s.p2[i] = (__bf16)r2;
    }
    return s;
}

// MODE 0: fp32 MFMA (reference)   MODE 1: bf16 x 6 products   MODE 2: bf16 x 3 products (a0b0 + a0b1 + a1b0)
// MODE 3: bf16 x 6 with the truncating split
template <int MODE>
__global__ __launch_bounds__(512, 2) void kern(const float* A, const float* W, float* C, int K) {
    constexpr int STAGE = 384 * BK;
    __shared__ __attribute__((aligned(1024))) float stages[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w & 1, wn = w >> 1;
    const int lrow = lane >> 3, lchunk = (lane & 7) ^ ((((w * 8) + lrow) >> 1) & 7);
    const unsigned lds_w = (unsigned)(uintptr_t)(lptr_t)&stages[0] + w * (8 * BK * 4);
    const float* src[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const int row = (w + 8 * j) * 8 + lrow;
        src[j] = (row < 128 ? A + ((size_t)blockIdx.x * 128 + row) * K : W + (size_t)(row - 128) * K) + lchunk * 4;
    }
    auto issue = [&](int st, int k0) {
#pragma unroll
        for (int j = 0; j < 6; j++) dma16(src[j] + k0, lds_w + st * (STAGE * 4) + j * (64 * BK * 4));
    };
    const int fr = lane & 31, fh = lane >> 5;
    const int sw = (fr >> 1) & 7;
    f32x16 acc[2][2] = {};
    auto chunk_at = [&](const float* cur, int row, int c) -> float4 { return *(const float4*)&cur[row * BK + ((c ^ sw) << 2)]; };
    const int S = K / BK;
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s = 0; s < S; s++) {
        if (s + 1 < S) issue((s + 1) & 1, (s + 1) * BK);
        const float* cur = stages + (s & 1) * STAGE;
        const int ra0 = wm * 64 + fr, rb0 = 128 + wn * 64 + fr;
        if (MODE == 0) {
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const int c = kk * 2 + fh;
                const float4 a0 = chunk_at(cur, ra0, c), a1 = chunk_at(cur, ra0 + 32, c);
                const float4 b0 = chunk_at(cur, rb0, c), b1 = chunk_at(cur, rb0 + 32, c);
                const float p0[4] = {a0.x, a0.y, a0.z, a0.w}, p1[4] = {a1.x, a1.y, a1.z, a1.w};
                const float q0[4] = {b0.x, b0.y, b0.z, b0.w}, q1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(p0[r], q0[r], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(p0[r], q1[r], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(p1[r], q0[r], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(p1[r], q1[r], acc[1][1], 0, 0, 0);
                }
            }
        } else {
            // two k groups of 16: lane (row fr, half fh) holds k = 16 g + 8 fh .. + 7 = chunks 4 g + 2 fh, + 1
#pragma unroll
            for (int g = 0; g < 2; g++) {
                const int c = 4 * g + 2 * fh;
                Split sa[2], sb[2];
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    if (MODE == 3) {
                        sa[i] = split8t(chunk_at(cur, ra0 + 32 * i, c), chunk_at(cur, ra0 + 32 * i, c + 1));
                        sb[i] = split8t(chunk_at(cur, rb0 + 32 * i, c), chunk_at(cur, rb0 + 32 * i, c + 1));
                    } else {
                        sa[i] = split8(chunk_at(cur, ra0 + 32 * i, c), chunk_at(cur, ra0 + 32 * i, c + 1));
                        sb[i] = split8(chunk_at(cur, rb0 + 32 * i, c), chunk_at(cur, rb0 + 32 * i, c + 1));
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) {
                        f32x16 t = acc[i][j];
                        if (MODE == 1 || MODE == 3) {
                            t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[i].p2, sb[j].p0, t, 0, 0, 0);
                            t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[i].p0, sb[j].p2, t, 0, 0, 0);
                            t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[i].p1, sb[j].p1, t, 0, 0, 0);
                        }
                        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[i].p1, sb[j].p0, t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[i].p0, sb[j].p1, t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[i].p0, sb[j].p0, t, 0, 0, 0);
                        acc[i][j] = t;
                    }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // C layout of the 32x32 MFMAs: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = blockIdx.x * 128 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                C[(size_t)row * 256 + wn * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
            }
}

template <int MODE>
float run(const float* A, const float* W, float* C, int NB, int K) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 60; i++) kern<MODE><<<NB, 512>>>(A, W, C, K);
    hipEventRecord(e0); for (int i = 0; i < 30; i++) kern<MODE><<<NB, 512>>>(A, W, C, K); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 30 * 1e3;
}

int main() {
    const int K = 2048, NB = 256, M = NB * 128;
    float *A, *W, *C0, *C1, *C2;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, 256 * K * 4);
    hipMalloc(&C0, (size_t)M * 256 * 4); hipMalloc(&C1, (size_t)M * 256 * 4); hipMalloc(&C2, (size_t)M * 256 * 4);
    float* h = (float*)malloc((size_t)M * K * 4);
    srand(1);
    for (size_t i = 0; i < (size_t)M * K; i++) h[i] = ((float)rand() / RAND_MAX - 0.5f) * 2.0f;
    hipMemcpy(A, h, (size_t)M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, h + 31337, 256 * K * 4, hipMemcpyHostToDevice);
    const double fl = 2.0 * M * 256 * K;
    float t3 = run<3>(A, W, C2, NB, K);
    float* c3 = (float*)malloc((size_t)4096 * 256 * 4); hipMemcpy(c3, C2, (size_t)4096 * 256 * 4, hipMemcpyDeviceToHost);
    float t0 = run<0>(A, W, C0, NB, K), t1 = run<1>(A, W, C1, NB, K), t2 = run<2>(A, W, C2, NB, K);
    printf("bf16 x 6, truncating : %8.1f us  %6.1f TFLOP/s fp32-equivalent\n", t3, 2.0 * M * 256 * K / t3 / 1e6);
    printf("fp32 MFMA            : %8.1f us  %6.1f TFLOP/s\n", t0, fl / t0 / 1e6);
    printf("bf16 x 6 (fp32 grade): %8.1f us  %6.1f TFLOP/s fp32-equivalent (%s)\n", t1, fl / t1 / 1e6, hipGetErrorString(hipGetLastError()));
    printf("bf16 x 3             : %8.1f us  %6.1f TFLOP/s fp32-equivalent\n", t2, fl / t2 / 1e6);
    const size_t n = (size_t)4096 * 256;          // compare the first 4096 rows against a double-precision host product
    float *c0 = (float*)malloc(n * 4), *c1 = (float*)malloc(n * 4), *c2 = (float*)malloc(n * 4);
    hipMemcpy(c0, C0, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c1, C1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c2, C2, n * 4, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0, e3 = 0, ref_max = 0;
    const float* hw = h + 31337;
    for (int r = 0; r < 64; r++)
        for (int c = 0; c < 256; c++) {
            double s = 0;
            for (int k = 0; k < K; k++) s += (double)h[(size_t)r * K + k] * (double)hw[(size_t)c * K + k];
            ref_max = fmax(ref_max, fabs(s));
            e0 = fmax(e0, fabs(c0[r * 256 + c] - s)); e1 = fmax(e1, fabs(c1[r * 256 + c] - s)); e2 = fmax(e2, fabs(c2[r * 256 + c] - s)); e3 = fmax(e3, fabs(c3[r * 256 + c] - s));
        }
    printf("max |error| vs fp64 (|ref| up to %.1f): fp32 MFMA %.3g, bf16x6 %.3g, bf16x6 truncating %.3g, bf16x3 %.3g\n", ref_max, e0, e1, e3, e2);
    return 0;
}
