// Developer microbenchmark: sustained v_mfma_f32_32x32x16_bf16 rate, registers only, and how much of it survives
// when the wave also runs the three-way fp32 -> bf16 split (common.h split3: ~44 full-rate VALU ops per 8 values)
// between its MFMAs.  MODE 0: MFMA only, 4 independent accumulators.  MODE 1: one dependent chain.
// MODE 2: per 24 MFMAs (a 2x2-block wave tile, six products) 4 splits of live data.  MODE 3: 1 split per 24 MFMAs.
// hipcc --offload-arch=gfx950 -O3 mfma_bf16_peak.hip -o /tmp/mfma_bf16_peak && /tmp/mfma_bf16_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct Split3 { bf16x8 p0, p1, p2; };
__device__ __forceinline__ Split3 split3(const float* v) {
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
    Split3 s;
    s.p0 = __builtin_bit_cast(bf16x8, q0); s.p1 = __builtin_bit_cast(bf16x8, q1); s.p2 = __builtin_bit_cast(bf16x8, q2);
    return s;
}
#define MF(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
template <int WAVES, int MODE>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, int iters, unsigned long long* clk) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    float v[4][8];
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int i = 0; i < 8; i++) v[s][i] = threadIdx.x * 1e-3f + i + 0.37f * s;
    Split3 sp[4];
#pragma unroll
    for (int s = 0; s < 4; s++) sp[s] = split3(v[s]);
    for (int it = 0; it < iters; it++) {
        if (MODE == 2 || MODE == 3) {
#pragma unroll
            for (int s = 0; s < (MODE == 2 ? 4 : 1); s++) {
#pragma unroll
                for (int i = 0; i < 8; i++) { v[s][i] = v[s][i] * 1.0001f; asm volatile("" : "+v"(v[s][i])); }
                sp[s] = split3(v[s]);
            }
        }
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 24; j++) MF(sp[j & 3].p0, sp[(j + 1) & 3].p1, a0);
        } else {
#define TERM(PA, PB) MF(sp[0].PA, sp[2].PB, a0); MF(sp[0].PA, sp[3].PB, a1); MF(sp[1].PA, sp[2].PB, a2); MF(sp[1].PA, sp[3].PB, a3);
            TERM(p2, p0) TERM(p0, p2) TERM(p1, p1) TERM(p1, p0) TERM(p0, p1) TERM(p0, p0)
#undef TERM
        }
    }
    float s = 0;
    for (int r = 0; r < 16; r++) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}
template <int WAVES, int MODE>
void run(const char* name) {
    float* out; hipMalloc(&out, 256 * 8 * 64 * 16 * sizeof(float));
    const int iters = 2000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long* clk; hipMalloc(&clk, 16);
    k<WAVES, MODE><<<blocks, 64 * WAVES>>>(out, 10, nullptr);
    for (int i = 0; i < 20; i++) k<WAVES, MODE><<<blocks, 64 * WAVES>>>(out, iters, nullptr);     // reach the sustained clock
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<WAVES, MODE><<<blocks, 64 * WAVES>>>(out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * WAVES * iters * 24.0 * 32768.0;
    unsigned long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    // s_memtime counts shader clocks, s_memrealtime a constant 100 MHz
    printf("%-44s %.3f ms %7.1f TFLOP/s bf16 = %6.1f fp32-eq (x6)  shader clock %.0f MHz\n", name, ms, flops / ms / 1e9, flops / ms / 6e9, 100.0 * hc[0] / hc[1]);
    hipFree(out);
}
int main() {
    run<4, 0>("MFMA only, 1 wave/SIMD");
    run<8, 0>("MFMA only, 2 waves/SIMD");
    run<16, 0>("MFMA only, 4 waves/SIMD");
    run<8, 1>("one dependent chain, 2 waves/SIMD");
    run<4, 2>("4 splits per 24 MFMAs, 1 wave/SIMD");
    run<8, 2>("4 splits per 24 MFMAs, 2 waves/SIMD");
    run<16, 2>("4 splits per 24 MFMAs, 4 waves/SIMD");
    run<8, 3>("1 split per 24 MFMAs, 2 waves/SIMD");
    run<16, 3>("1 split per 24 MFMAs, 4 waves/SIMD");
    return 0;
}
