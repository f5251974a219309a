// Developer microbenchmark: does a busy MFMA pipe cost HBM bandwidth (shared power/fabric budget)?
// Half of the workgroups stream a copy (read 1 GiB, write 1 GiB), the other half run a register-resident
// fp32 MFMA loop (or exit at once).  Roles alternate by blockIdx so every CU hosts both kinds.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(const float4* src, float4* dst, size_t n4, int mfma_iters, int copy_on, float* sink,
                                         unsigned long long* tcopy) {
    const int role = blockIdx.x & 1, id = blockIdx.x >> 1, nb = gridDim.x >> 1;
    if (role == 0) {
        if (!copy_on) return;
        const unsigned long long t0 = wall_clock64();
        for (size_t i = (size_t)id * 256 + threadIdx.x; i < n4; i += (size_t)nb * 256) dst[i] = src[i];
        if (threadIdx.x == 0) atomicMax(tcopy, wall_clock64() - t0);
    } else {
        f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
        float x = threadIdx.x * 1e-3f, y = 0.5f;
        for (int i = 0; i < mfma_iters; i++) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
        }
        float t = 0; for (int r = 0; r < 16; r++) t += a0[r] + a1[r] + a2[r] + a3[r];
        sink[blockIdx.x * 256 + threadIdx.x] = t;
    }
}
int main() {
    const size_t bytes = 1ull << 30, n4 = bytes / 16;
    float4 *src, *dst; float* sink; unsigned long long* tc;
    hipMalloc(&src, bytes); hipMalloc(&dst, bytes); hipMalloc(&sink, 1024 * 256 * 4); hipMalloc(&tc, 8);
    hipMemset(src, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](int iters, int copy_on, const char* name) {
        float ms = 0; unsigned long long h = 0;
        for (int rep = 0; rep < 3; rep++) {
            hipMemset(tc, 0, 8);
            hipEventRecord(e0); k<<<1024, 256>>>(src, dst, n4, iters, copy_on, sink, tc); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(&h, tc, 8, hipMemcpyDeviceToHost);
        }
        const double tcopy = h * 1e-8;     // 100 MHz ticks
        printf("%-44s kernel %.3f ms", name, ms);
        if (copy_on) printf("  copy %.3f ms = %.2f TB/s (read+write)", tcopy * 1e3, 2.0 * bytes / tcopy / 1e12);
        if (iters) printf("  mfma %.1f TFLOP/s (over kernel time)", 512.0 * 4 * iters * 4 * 4096.0 / (ms * 1e-3) / 1e12);
        printf("\n");
    };
    run(0, 1, "copy alone");
    run(40000, 0, "MFMA alone (4 waves/CU-half)");
    run(40000, 1, "copy + MFMA together");
    run(0, 1, "copy alone (again)");
    run(80000, 1, "copy + longer MFMA");
    return 0;
}
