// Developer microbenchmark (follow-up of nt_deferred.hip): deferred epilogue, one slice per K-step of the
// NEXT tile, with the aux operand arriving through LDS-DMA (each lane parks its own 2 x 16 B in LDS a K-step
// ahead: no compiler-visible vector loads, so no compiler-placed vmcnt waits in the MFMA stream).
// Tile 128x256, K = 128 as 8 K-steps of 16, NST-stage ring, one 8-wave workgroup per CU.
// hipcc --offload-arch=gfx950 -O3 nt_deferred2.hip -o /tmp/nt_deferred2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define mfma(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
#define BK 16
#define KK 128
#define NK (KK / BK)
#define PLD 36
#define NST 3
#define AH 2          // a slice's aux is requested AH K-steps before the slice runs
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void dma16(const float* g, unsigned l) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(l) : "memory");
}

// MODE 0: epilogue after the K-loop, aux by plain loads.  MODE 1: deferred, aux through LDS-DMA.
template <int MODE, int ABL>
__global__ __launch_bounds__(512, 2) void kern(const float* A, const float* W, const float* aux, float* C, int ntiles) {
    constexpr int STAGE = 384 * BK;
    __shared__ __attribute__((aligned(1024))) float stages[NST * STAGE];
    __shared__ __attribute__((aligned(16))) float patch[8 * 16 * PLD];
    __shared__ __attribute__((aligned(1024))) float auxb[4 * 8 * 512];          // [buffer (3 live + 1 dump)][wave][2 x 1 KB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w & 1, wn = w >> 1;
    const int lrow = lane >> 2, lchunk = (lane & 3) ^ ((lrow >> 2) & 3);
    const unsigned lds_w = (unsigned)(uintptr_t)(lptr_t)&stages[0] + w * (16 * BK * 4);
    const unsigned lds_x = (unsigned)(uintptr_t)(lptr_t)&auxb[0] + w * 2048;
    const float* src[3];
    auto ptrs = [&](int tile) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int row = (w + 8 * j) * 16 + lrow;
            src[j] = (row < 128 ? A + ((size_t)tile * 128 + row) * KK : W + (size_t)(row - 128) * KK) + lchunk * 4;
        }
    };
    auto issue = [&](int st, int k0) {
#pragma unroll
        for (int j = 0; j < 3; j++) dma16(src[j] + k0, lds_w + st * (STAGE * 4) + j * (128 * BK * 4));
    };
    const int fr = lane & 31;
    const int fsw = ((lane >> 5) ^ ((fr >> 2) & 3)) << 2;
    const int fa = (wm * 64 + fr) * BK + fsw, fb = (128 + wn * 64 + fr) * BK + fsw;
    f32x16 acc[2][2], prv[2][2];
    auto zero = [&]() {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    };
    float* stg = patch + w * 16 * PLD;
    const int er = lane >> 3, ec = (lane & 7) * 4;
    auto slice_addr = [&](int t, int h, size_t& r0, int& col) {
        const int nt = h >> 2, mt = (h >> 1) & 1, half = h & 1;
        col = wn * 64 + nt * 32 + ec;
        r0 = (size_t)t * 128 + wm * 64 + mt * 32 + half * 16 + er;
    };
    auto aux_dma = [&](int t, int h, int buf) {      // this lane's two 16-B pieces of slice h -> LDS
        size_t r0; int col; slice_addr(t, h, r0, col);
        dma16(aux + r0 * 256 + col, lds_x + buf * 16384);
        dma16(aux + (r0 + 8) * 256 + col, lds_x + buf * 16384 + 1024);
    };
    auto slice = [&](f32x16 (&S)[2][2], int t, int h, float4 x0, float4 x1) {
        const int nt = h >> 2, mt = (h >> 1) & 1, half = h & 1;
        size_t r0; int col; slice_addr(t, h, r0, col);
#pragma unroll
        for (int q8 = 0; q8 < 8; q8++) stg[((q8 & 3) + 8 * (q8 >> 2) + 4 * (lane >> 5)) * PLD + (lane & 31)] = S[mt][nt][half * 8 + q8];
        __builtin_amdgcn_wave_barrier();
        float4 v0 = *(const float4*)&stg[er * PLD + ec], v1 = *(const float4*)&stg[(er + 8) * PLD + ec];
        v0.x *= 1.f - x0.x * x0.x; v0.y *= 1.f - x0.y * x0.y; v0.z *= 1.f - x0.z * x0.z; v0.w *= 1.f - x0.w * x0.w;
        v1.x *= 1.f - x1.x * x1.x; v1.y *= 1.f - x1.y * x1.y; v1.z *= 1.f - x1.z * x1.z; v1.w *= 1.f - x1.w * x1.w;
        if (!(ABL & 2) || v0.x == 1234.5f) {
            if (ABL & 4) {
                __builtin_nontemporal_store(v0.x, C + r0 * 256 + col); __builtin_nontemporal_store(v0.y, C + r0 * 256 + col + 1);
                __builtin_nontemporal_store(v0.z, C + r0 * 256 + col + 2); __builtin_nontemporal_store(v0.w, C + r0 * 256 + col + 3);
                __builtin_nontemporal_store(v1.x, C + (r0 + 8) * 256 + col); __builtin_nontemporal_store(v1.y, C + (r0 + 8) * 256 + col + 1);
                __builtin_nontemporal_store(v1.z, C + (r0 + 8) * 256 + col + 2); __builtin_nontemporal_store(v1.w, C + (r0 + 8) * 256 + col + 3);
            } else {
                *(float4*)(C + r0 * 256 + col) = v0;
                *(float4*)(C + (r0 + 8) * 256 + col) = v1;
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    auto slice_lds = [&](f32x16 (&S)[2][2], int t, int h, int buf) {
        const float* xb = auxb + buf * 4096 + w * 512 + lane * 4;
        slice(S, t, h, *(const float4*)xb, *(const float4*)(xb + 256));
    };
    auto slice_glb = [&](f32x16 (&S)[2][2], int t, int h) {
        size_t r0; int col; slice_addr(t, h, r0, col);
        slice(S, t, h, *(const float4*)(aux + r0 * 256 + col), *(const float4*)(aux + (r0 + 8) * 256 + col));
    };
    auto compute = [&](const float* cur) {
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            const int x = kk << 3;
            const float4 a0 = *(const float4*)&cur[fa ^ x], a1 = *(const float4*)&cur[(fa ^ x) + 32 * BK];
            const float4 b0 = *(const float4*)&cur[fb ^ x], b1 = *(const float4*)&cur[(fb ^ x) + 32 * BK];
            const float p0[4] = {a0.x, a0.y, a0.z, a0.w}, p1[4] = {a1.x, a1.y, a1.z, a1.w};
            const float q0[4] = {b0.x, b0.y, b0.z, b0.w}, q1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int r = 0; r < 4; r++) {
                acc[0][0] = mfma(p0[r], q0[r], acc[0][0]); acc[0][1] = mfma(p0[r], q1[r], acc[0][1]);
                acc[1][0] = mfma(p1[r], q0[r], acc[1][0]); acc[1][1] = mfma(p1[r], q1[r], acc[1][1]);
            }
        }
    };
    // the stage that lands LAST must be complete; everything older completes before it (in-order vmcnt)
    auto sync = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    // ring of NST stages: stage q of the block's K-step stream lives in slot q % NST; NST-1 steps are in flight
    ptrs(tile);
    int q_issue = 0;                      // next K-step (of the stream) to issue
    int itile = tile, ik = 0;             // its tile / k index
    auto issue_next = [&]() {
        if (itile >= ntiles) return;
        issue(q_issue % NST, ik * BK);
        q_issue++;
        if (++ik == NK) { ik = 0; itile += gridDim.x; if (itile < ntiles) ptrs(itile); }
    };
#pragma unroll
    for (int i = 0; i < NST - 1; i++) issue_next();
    zero();
    int q = 0, ptile = -1, abuf = 0;       // abuf: ring position of the aux buffer the NEXT request goes to
    const int my_tiles = (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;
    const int total_steps = my_tiles * NK;
    while (true) {
        const int ntile = tile + gridDim.x;
#pragma unroll
        for (int kt = 0; kt < NK; kt++) {
            // stage q and the aux requested AH steps ago must have landed; the younger stage + aux stay in flight
            if (MODE == 1 && q + NST - 2 < total_steps && q - 2 >= NK && !(ABL & 2)) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * 3 + (AH - 1) * 2 + 4) : "memory"); }   // + the 2 x 2 C stores of the last two slices
            else if (MODE == 1 && q + NST - 2 < total_steps) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * 3 + (AH - 1) * 2) : "memory"); }
            else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            __syncthreads();
            if (MODE == 1) {
                // request the aux of the slice that runs AH steps from now: slice kt+AH of the previous tile, or
                // (wrapping) slice kt+AH-8 of this tile, whose epilogue runs under the next tile's K-loop
                const int hh = kt + AH;
                if (ABL & 1) {}
                else if (hh < NK) { if (ptile >= 0) aux_dma(ptile, hh, abuf); else aux_dma(tile, 0, 3); }
                else aux_dma(tile, hh - NK, abuf);
                abuf = abuf == 2 ? 0 : abuf + 1;
            }
            issue_next();
            if (MODE == 1 && ptile >= 0) slice_lds(prv, ptile, kt, (abuf + 3 - 1 - AH) % 3);
            compute(stages + (q % NST) * STAGE);
            q++;
        }
        if (MODE == 0) {
#pragma unroll
            for (int h = 0; h < 8; h++) slice_glb(acc, tile, h);
        } else {
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) prv[i][j] = acc[i][j];
            ptile = tile;
        }
        if (ntile >= ntiles) break;
        zero();
        tile = ntile;
    }
    if (MODE == 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int h = 0; h < 8; h++) slice_glb(prv, ptile, h);
    }
}

template <int MODE, int ABL>
float run(const float* A, const float* W, const float* X, float* C, int ntiles) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 100; i++) kern<MODE, ABL><<<256, 512>>>(A, W, X, C, ntiles);
    hipEventRecord(e0); for (int i = 0; i < 50; i++) kern<MODE, ABL><<<256, 512>>>(A, W, X, C, ntiles); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 50 * 1e3;
}

int main() {
    const int ntiles = 918, M = ntiles * 128;
    float *A, *W, *X, *C0, *C1;
    hipMalloc(&A, (size_t)M * KK * 4); hipMalloc(&W, 256 * KK * 4); hipMalloc(&X, (size_t)M * 256 * 4);
    hipMalloc(&C0, (size_t)M * 256 * 4); hipMalloc(&C1, (size_t)M * 256 * 4);
    float* h = (float*)malloc((size_t)M * 256 * 4);
    for (size_t i = 0; i < (size_t)M * 256; i++) h[i] = (float)((i * 2654435761u) % 1009) * 1e-3f - 0.5f;
    hipMemcpy(A, h, (size_t)M * KK * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, h + 777, 256 * KK * 4, hipMemcpyHostToDevice);
    hipMemcpy(X, h, (size_t)M * 256 * 4, hipMemcpyHostToDevice);
    const double fl = 2.0 * M * 256 * KK;
    float t;
    t = run<0, 0>(A, W, X, C0, ntiles); printf("epilogue after the K-loop            : %.1f us  %.1f TFLOP/s\n", t, fl / t / 1e6);
    t = run<1, 0>(A, W, X, C1, ntiles); printf("deferred, aux via LDS-DMA            : %.1f us  %.1f TFLOP/s (%s)\n", t, fl / t / 1e6, hipGetErrorString(hipGetLastError()));
    float* c0 = (float*)malloc((size_t)M * 256 * 4); float* c1 = (float*)malloc((size_t)M * 256 * 4);
    hipMemcpy(c0, C0, (size_t)M * 256 * 4, hipMemcpyDeviceToHost); hipMemcpy(c1, C1, (size_t)M * 256 * 4, hipMemcpyDeviceToHost);
    size_t bad = 0; for (size_t i = 0; i < (size_t)M * 256; i++) if (c0[i] != c1[i]) bad++;
    printf("mismatches: %zu (sample %g %g)\n", bad, c0[12345], c1[12345]);
    t = run<1, 1>(A, W, X, C1, ntiles); printf("deferred, no aux traffic             : %.1f us\n", t);
    t = run<1, 2>(A, W, X, C1, ntiles); printf("deferred, no C stores                : %.1f us\n", t);
    t = run<1, 3>(A, W, X, C1, ntiles); printf("deferred, neither                    : %.1f us\n", t);
    t = run<1, 4>(A, W, X, C1, ntiles); printf("deferred, nontemporal C stores       : %.1f us\n", t);
    t = run<0, 4>(A, W, X, C1, ntiles); printf("after-loop, nontemporal C stores     : %.1f us\n", t);
    t = run<0, 0>(A, W, X, C0, ntiles); printf("epilogue after the K-loop (again)    : %.1f us\n", t);
    t = run<1, 0>(A, W, X, C1, ntiles); printf("deferred, aux via LDS-DMA (again)    : %.1f us\n", t);
    return 0;
}
