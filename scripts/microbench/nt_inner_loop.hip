// Developer microbenchmark: the NT kernel's inner loop alone (LDS-resident stage, no global traffic),
// in several source formulations.  hipcc --offload-arch=gfx950 -O3 nt_inner_loop.hip -o /tmp/nt_inner
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LD 36
#define mfma(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

template <int V>
__global__ __launch_bounds__(512, 2) void k(float* out, int steps) {
    __shared__ __attribute__((aligned(16))) float smem[2][(128 + 256) * LD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w & 1, wn = w >> 1;
    for (int i = tid; i < 2 * (128 + 256) * LD; i += 512) (&smem[0][0])[i] = (float)((i * 37) % 101) * 1e-3f;
    __syncthreads();
    f32x16 acc[2][2] = {};
    const int frag = (lane & 31) * LD + 4 * (lane >> 5);
    for (int s = 0; s < steps; s++) {
        const float* As = &smem[s & 1][wm * 64 * LD + frag];
        const float* Ws = &smem[s & 1][(128 + wn * 64) * LD + frag];
        if (V == 0) {
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const float4 a0 = *(const float4*)(As + kk * 8), a1 = *(const float4*)(As + 32 * LD + kk * 8);
                const float4 b0 = *(const float4*)(Ws + kk * 8), b1 = *(const float4*)(Ws + 32 * LD + kk * 8);
                const float x0[4] = {a0.x, a0.y, a0.z, a0.w}, x1[4] = {a1.x, a1.y, a1.z, a1.w};
                const float y0[4] = {b0.x, b0.y, b0.z, b0.w}, y1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    acc[0][0] = mfma(x0[r], y0[r], acc[0][0]); acc[0][1] = mfma(x0[r], y1[r], acc[0][1]);
                    acc[1][0] = mfma(x1[r], y0[r], acc[1][0]); acc[1][1] = mfma(x1[r], y1[r], acc[1][1]);
                }
            }
        } else if (V == 1) {   // all fragments of the step up front
            float4 a0[4], a1[4], b0[4], b1[4];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                a0[kk] = *(const float4*)(As + kk * 8); a1[kk] = *(const float4*)(As + 32 * LD + kk * 8);
                b0[kk] = *(const float4*)(Ws + kk * 8); b1[kk] = *(const float4*)(Ws + 32 * LD + kk * 8);
            }
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const float x0[4] = {a0[kk].x, a0[kk].y, a0[kk].z, a0[kk].w}, x1[4] = {a1[kk].x, a1[kk].y, a1[kk].z, a1[kk].w};
                const float y0[4] = {b0[kk].x, b0[kk].y, b0[kk].z, b0[kk].w}, y1[4] = {b1[kk].x, b1[kk].y, b1[kk].z, b1[kk].w};
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    acc[0][0] = mfma(x0[r], y0[r], acc[0][0]); acc[0][1] = mfma(x0[r], y1[r], acc[0][1]);
                    acc[1][0] = mfma(x1[r], y0[r], acc[1][0]); acc[1][1] = mfma(x1[r], y1[r], acc[1][1]);
                }
            }
        } else if (V == 2) {   // as V1 but the NEXT step's fragments are requested before this step's MFMAs (cross-step pipeline)
            // emulate: load fragments for stage (s+1)&1 first, then compute with registers from previous iteration
            static_assert(V != 2 || true, "");
        }
        if (V != 3) __syncthreads();
    }
    float t = 0;
    for (int r = 0; r < 16; r++) t += acc[0][0][r] + acc[0][1][r] + acc[1][0][r] + acc[1][1][r];
    out[blockIdx.x * 512 + tid] = t;
}

// V2: explicit cross-step software pipeline
__global__ __launch_bounds__(512, 2) void k2(float* out, int steps) {
    __shared__ __attribute__((aligned(16))) float smem[2][(128 + 256) * LD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w & 1, wn = w >> 1;
    for (int i = tid; i < 2 * (128 + 256) * LD; i += 512) (&smem[0][0])[i] = (float)((i * 37) % 101) * 1e-3f;
    __syncthreads();
    f32x16 acc[2][2] = {};
    const int frag = (lane & 31) * LD + 4 * (lane >> 5);
    float4 a0[4], a1[4], b0[4], b1[4];
    auto ld = [&](int st) {
        const float* As = &smem[st][wm * 64 * LD + frag];
        const float* Ws = &smem[st][(128 + wn * 64) * LD + frag];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            a0[kk] = *(const float4*)(As + kk * 8); a1[kk] = *(const float4*)(As + 32 * LD + kk * 8);
            b0[kk] = *(const float4*)(Ws + kk * 8); b1[kk] = *(const float4*)(Ws + 32 * LD + kk * 8);
        }
    };
    ld(0);
    for (int s = 0; s < steps; s++) {
        float4 c0[4], c1[4], d0[4], d1[4];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) { c0[kk] = a0[kk]; c1[kk] = a1[kk]; d0[kk] = b0[kk]; d1[kk] = b1[kk]; }
        __syncthreads();
        ld((s + 1) & 1);
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const float x0[4] = {c0[kk].x, c0[kk].y, c0[kk].z, c0[kk].w}, x1[4] = {c1[kk].x, c1[kk].y, c1[kk].z, c1[kk].w};
            const float y0[4] = {d0[kk].x, d0[kk].y, d0[kk].z, d0[kk].w}, y1[4] = {d1[kk].x, d1[kk].y, d1[kk].z, d1[kk].w};
#pragma unroll
            for (int r = 0; r < 4; r++) {
                acc[0][0] = mfma(x0[r], y0[r], acc[0][0]); acc[0][1] = mfma(x0[r], y1[r], acc[0][1]);
                acc[1][0] = mfma(x1[r], y0[r], acc[1][0]); acc[1][1] = mfma(x1[r], y1[r], acc[1][1]);
            }
        }
    }
    float t = 0;
    for (int r = 0; r < 16; r++) t += acc[0][0][r] + acc[0][1][r] + acc[1][0][r] + acc[1][1][r];
    out[blockIdx.x * 512 + tid] = t;
}

template <typename F>
void timeit(const char* name, F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(20); hipDeviceSynchronize();
    const int steps = 4000;
    hipEventRecord(e0); launch(steps); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 256.0 * 8 * steps * 64.0 * 4096.0;
    printf("%-40s %.3f ms  %.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    timeit("V0 per-kk fragments + barrier", [&](int s) { k<0><<<256, 512>>>(out, s); });
    timeit("V1 all fragments up front + barrier", [&](int s) { k<1><<<256, 512>>>(out, s); });
    timeit("V2 cross-step fragment pipeline", [&](int s) { k2<<<256, 512>>>(out, s); });
    timeit("V0 again", [&](int s) { k<0><<<256, 512>>>(out, s); });
    return 0;
}
