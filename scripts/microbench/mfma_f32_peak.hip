// Developer microbenchmark: sustained v_mfma_f32_32x32x2_f32 rate, registers only.
// hipcc --offload-arch=gfx950 -O3 mfma_f32_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, int iters) {
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-4f + 1.0f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
        }
    }
    float s = 0;
    for (int r = 0; r < 16; r++) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int WAVES>
void run(const char* name) {
    float* out; hipMalloc(&out, 256 * 4 * 64 * WAVES * sizeof(float));
    const int iters = 2000, blocks = 256 * (WAVES >= 8 ? 1 : 8 / WAVES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<WAVES><<<blocks, 64 * WAVES>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<WAVES><<<blocks, 64 * WAVES>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * WAVES * iters * 64.0 * 4096.0;
    printf("%s: %.3f ms  %.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
    hipFree(out);
}
int main() { run<4>("4 waves/block x2 per CU (2 waves/SIMD)"); run<8>("8 waves/block (2 waves/SIMD)"); run<4>("again"); return 0; }
