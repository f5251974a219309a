// Does a wave's OWN instruction stream have to interleave the operand split with the matrix instructions?
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../p_companion_amd/csrc -I../../include nt_interleave.hip -o /tmp/nt_interleave
//
// The K-step of gemm_nt_kernel per wave: 8 ds_read_b128 (two A and two W blocks of 32 rows x 16 k), four three-way bf16
// splits (176 VALU) and 24 v_mfma_f32_32x32x16_bf16.  SQ counters say a SIMD spends the SUM of its waves' VALU and MFMA
// phases (removing the barriers or the DMA waits changes 0-7 %), i.e. co-resident waves do not fill each other's gaps.
// Modes, W waves per SIMD, no DMA and no barrier (the stage is static):
//   0  as the kernel is written: fragments of this step read + split, then its 24 MFMAs (compiler's order)
//   1  software pipeline: the 24 MFMAs of step k run on fragments split during step k-1 while the fragments of step k+1 are
//      read and split (compiler's order)
//   2  = 1 with sched_group_barrier: one MFMA, then 8 VALU, ...
//   3  = 1 with the dot2c split of 112 VALU (sched_group_barrier: one MFMA, 5 VALU)
//   6  the 24 MFMAs alone; 7  the reads and splits alone
//   4  = 0, 5 = 1 with the accumulators in AGPRs (inline-asm MFMA: the compiler cannot move those, so the stream is as written)
// Output: cycles per K-step per wave and per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "common.h"

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ Split3 split3_dot(const float4& lo, const float4& hi) {
    float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 q0, q1, q2;
    unsigned c_lo = 0x0000bf80u, c_hi = 0xbf800000u;
    asm volatile("" : "+s"(c_lo), "+s"(c_hi));
    const bf16x2 m_lo = __builtin_bit_cast(bf16x2, c_lo), m_hi = __builtin_bit_cast(bf16x2, c_hi);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float a = v[2 * j], b = v[2 * j + 1];
        const unsigned t0 = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
        a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, t0), m_lo, a, false);
        b = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, t0), m_hi, b, false);
        const unsigned t1 = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
        a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, t1), m_lo, a, false);
        b = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, t1), m_hi, b, false);
        q0[j] = t0; q1[j] = t1;
        q2[j] = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
    }
    Split3 s;
    s.p0 = __builtin_bit_cast(bf16x8, q0); s.p1 = __builtin_bit_cast(bf16x8, q1); s.p2 = __builtin_bit_cast(bf16x8, q2);
    return s;
}

#define BK 16
struct Frags { Split3 a[2], b[2]; };

template <bool DOT>
__device__ __forceinline__ Frags load_split(const float* st, int wm, int wn, int lane) {
    const int fr = lane & 31, gsw = (fr / 4) % 4;
    const int c0 = ((2 * (lane >> 5)) ^ gsw) << 2, c1 = ((2 * (lane >> 5) + 1) ^ gsw) << 2;
    const float* ar = st + (wm * 64 + fr) * BK;
    const float* br = st + (128 + wn * 64 + fr) * BK;
    Frags f;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const float4 lo = *reinterpret_cast<const float4*>(ar + i * 32 * BK + c0), hi = *reinterpret_cast<const float4*>(ar + i * 32 * BK + c1);
        f.a[i] = DOT ? split3_dot(lo, hi) : split3(lo, hi);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const float4 lo = *reinterpret_cast<const float4*>(br + j * 32 * BK + c0), hi = *reinterpret_cast<const float4*>(br + j * 32 * BK + c1);
        f.b[j] = DOT ? split3_dot(lo, hi) : split3(lo, hi);
    }
    return f;
}

// AGPR: the accumulators are pinned to the accumulation registers (inline asm, "+a"); otherwise the compiler keeps them in
// the ordinary vector registers whenever the kernel fits 256 of those
template <bool AGPR>
__device__ __forceinline__ f32x16 mm(const bf16x8& a, const bf16x8& b, f32x16 c) {
    if (AGPR) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
        return c;
    }
    return mfma_bf16(a, b, c);
}
template <bool AGPR>
__device__ __forceinline__ void mfmas(const Frags& f, f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
#define T(PA, PB) acc[0][j] = mm<AGPR>(f.a[0].PA, f.b[j].PB, acc[0][j]); acc[1][j] = mm<AGPR>(f.a[1].PA, f.b[j].PB, acc[1][j]);
        T(p2, p0) T(p0, p2) T(p1, p1) T(p1, p0) T(p0, p1) T(p0, p0)
#undef T
    }
}

template <int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void kern(const float* in, float* out, int iters, unsigned long long* cyc, int active_mod) {
    if (active_mod > 1 && ((blockIdx.x >> 3) % active_mod) != 0) { if (threadIdx.x == 0) cyc[blockIdx.x] = 0; return; }   // (whole groups of 8: one per XCD)
    __shared__ __attribute__((aligned(1024))) float stage[2][256 * BK];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w & 1, wn = w >> 1;
    for (int i = tid; i < 2 * 256 * BK; i += 256) (&stage[0][0])[i] = in[i];
    __syncthreads();
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 6) {                      // the 24 MFMAs alone, on fragments split once
        const Frags f = load_split<false>(stage[0], wm, wn, lane);
        for (int it = 0; it < iters; it++) {
            mfmas<false>(f, acc);
            asm volatile("" ::: "memory");
        }
    } else if (MODE == 7) {               // reads + splits alone (the pieces are kept alive by an empty asm)
        for (int it = 0; it < iters; it++) {
            const Frags f = load_split<false>(stage[it & 1], wm, wn, lane);
            asm volatile("" ::"v"(f.a[0].p0), "v"(f.a[0].p1), "v"(f.a[0].p2), "v"(f.a[1].p0), "v"(f.a[1].p1), "v"(f.a[1].p2),
                         "v"(f.b[0].p0), "v"(f.b[0].p1), "v"(f.b[0].p2), "v"(f.b[1].p0), "v"(f.b[1].p1), "v"(f.b[1].p2));
        }
    } else if (MODE == 0 || MODE == 4) {
        for (int it = 0; it < iters; it++) {
            const Frags f = load_split<false>(stage[it & 1], wm, wn, lane);
            mfmas<MODE == 4>(f, acc);
        }
    } else {
        Frags f = load_split<MODE == 3>(stage[0], wm, wn, lane);
        for (int it = 0; it < iters; it++) {
            const Frags n = load_split<MODE == 3>(stage[(it + 1) & 1], wm, wn, lane);
            mfmas<MODE == 5>(f, acc);
            if (MODE == 2 || MODE == 3) {
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);               // the 8 ds_reads first
#pragma unroll
                for (int g = 0; g < 24; g++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, MODE == 3 ? 5 : 8, 0);   // then VALU
                }
            }
            f = n;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int WPS>
static void run(const float* in, float* out, unsigned long long* cyc, int iters, int cus = 256, int active_mod = 1) {
    const int grid = cus * WPS;
    std::vector<unsigned long long> h(grid);
    double best = 1e30;
    float ms = 0.f;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((kern<MODE, WPS>), dim3(grid), dim3(256), 0, 0, in, out, iters, cyc, active_mod);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
        double m = 0;
        int live = 0;
        for (auto v : h) { m += (double)v; live += v != 0; }
        m /= live;
        if (m < best) best = m;
    }
    // s_memtime ticks at 100 MHz on this part: convert through the event time
    const double tflops = 2.0 * 128 * 128 * BK * (double)iters * grid / active_mod / (ms * 1e-3) / 1e12;
    if (active_mod > 1) printf("every %d-th group of 8 workgroups computes, the others exit: ", active_mod);
    printf("CUs %3d  mode %d  waves/SIMD %d  %8.1f memtime ticks/iter  %7.3f ms  %6.1f TFLOP/s-equivalent (fp32-grade), %6.1f us per 16 K-steps\n",
           cus, MODE, WPS, best / iters, ms, tflops, ms * 1e3 / iters * 16);
}

// ---- a 64 x 128 wave tile (2 x 4 blocks): six operand splits per eight block products instead of eight (the stage holds 128 A
// rows and 256 W rows; four waves as 2 x 2 own a 128 x 256 output tile).  Same reads-then-MFMAs order as mode 0.
template <int WPS>
__global__ __launch_bounds__(256, WPS) void kern_wide(const float* in, float* out, int iters, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(1024))) float stage[2][384 * BK];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w & 1, wn = w >> 1;
    for (int i = tid; i < 2 * 384 * BK; i += 256) (&stage[0][0])[i] = in[i % (2 * 256 * BK)];
    __syncthreads();
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const int fr = lane & 31, gsw = (fr / 4) % 4;
    const int c0 = ((2 * (lane >> 5)) ^ gsw) << 2, c1 = ((2 * (lane >> 5) + 1) ^ gsw) << 2;
    for (int it = 0; it < iters; it++) {
        const float* st = stage[it & 1];
        const float* ar = st + (wm * 64 + fr) * BK;
        const float* br = st + (128 + wn * 128 + fr) * BK;
        Split3 sa[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const float4 lo = *reinterpret_cast<const float4*>(ar + i * 32 * BK + c0), hi = *reinterpret_cast<const float4*>(ar + i * 32 * BK + c1);
            sa[i] = split3(lo, hi);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float4 lo = *reinterpret_cast<const float4*>(br + j * 32 * BK + c0), hi = *reinterpret_cast<const float4*>(br + j * 32 * BK + c1);
            const Split3 sb = split3(lo, hi);
#define T(PA, PB) acc[0][j] = mfma_bf16(sa[0].PA, sb.PB, acc[0][j]); acc[1][j] = mfma_bf16(sa[1].PA, sb.PB, acc[1][j]);
            T(p2, p0) T(p0, p2) T(p1, p1) T(p1, p0) T(p0, p1) T(p0, p0)
#undef T
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int WPS>
static void run_wide(const float* in, float* out, unsigned long long* cyc, int iters) {
    const int grid = 256 * WPS;
    float ms = 0.f;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((kern_wide<WPS>), dim3(grid), dim3(256), 0, 0, in, out, iters, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double tflops = 2.0 * 128 * 256 * BK * (double)iters * grid / (ms * 1e-3) / 1e12;
    printf("64x128 wave tiles (6 splits per 48 MFMAs)  waves/SIMD %d  %7.3f ms  %6.1f TFLOP/s-equivalent (fp32-grade)\n", WPS, ms, tflops);
}

int main() {
    float *in, *out;
    unsigned long long* cyc;
    hipMalloc(&in, 2 * 256 * BK * 4); hipMalloc(&out, 256 * 3 * 256 * 4); hipMalloc(&cyc, 256 * 3 * 8);
    std::vector<float> h(2 * 256 * BK);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int iters = 4096;
    run<0, 3>(in, out, cyc, iters);
    run<0, 2>(in, out, cyc, iters);
    run<0, 1>(in, out, cyc, iters);
    run<1, 2>(in, out, cyc, iters);
    run<1, 1>(in, out, cyc, iters);
    run<2, 2>(in, out, cyc, iters);
    run<2, 1>(in, out, cyc, iters);
    run<6, 3>(in, out, cyc, iters);
    run<3, 2>(in, out, cyc, iters);
    run<3, 1>(in, out, cyc, iters);
    run<4, 3>(in, out, cyc, iters);
    run<4, 2>(in, out, cyc, iters);
    run<4, 1>(in, out, cyc, iters);
    run<5, 2>(in, out, cyc, iters);
    run<5, 1>(in, out, cyc, iters);
    run<6, 3>(in, out, cyc, iters);
    run<6, 1>(in, out, cyc, iters);
    run<7, 3>(in, out, cyc, iters);
    run<7, 1>(in, out, cyc, iters);
    run_wide<2>(in, out, cyc, iters);
    run_wide<1>(in, out, cyc, iters);
    // chip-wide budget or per-CU limit?  One workgroup per CU, the MFMAs alone, on every CU / on every second / fourth CU's worth
    run<6, 1>(in, out, cyc, iters, 256, 2);
    run<6, 1>(in, out, cyc, iters, 256, 4);
    run<0, 1>(in, out, cyc, iters, 256, 2);
    run<0, 1>(in, out, cyc, iters, 256, 4);
    // is the wall the chip's power / current limit?  Fewer busy CUs (blocks b, b + 8, ... share an XCD; 64 blocks = 8 per XCD),
    // and all-zero operands (less switching): per-CU rate and the clock (ticks per ns) should both go up
    hipMemset(in, 0, 2 * 256 * BK * 4);
    run<0, 3>(in, out, cyc, iters);
    run<2, 1>(in, out, cyc, iters);
    run<6, 3>(in, out, cyc, iters);
    return 0;
}
