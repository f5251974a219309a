// Developer microbenchmark: fp32-grade NT product on the bf16 matrix cores with BOTH operands split once (planes in HBM).
// nt_bf16x6.hip splits every fragment in registers in every wave that uses it (4 splits per 24 MFMAs for a 2x2-block
// wave tile: mfma_bf16_peak.hip says that alone caps the loop at 226 of 318 TFLOP/s-equivalent).  The weight matrix is
// the same for every row tile, so here a tiny pre-pass writes it as three bf16 planes [3][N][K]; the K loop DMAs the
// planes into LDS (64 B per row and plane at BK = 32: the same XOR-swizzled 16-B chunks) and a W fragment is three
// ds_read_b128 with no VALU work.  Wave tile TA x TB blocks of 32x32: <2,2> 2 splits per 24 MFMAs, <1,4> 1 per 24.
// hipcc --offload-arch=gfx950 -O3 nt_wplanes.hip -o /tmp/nt_wplanes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define BK 32
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
// W [256][K] fp32 -> Wp [3][256][K] bf16 (truncating three-way split)
__global__ void split_w(const float* W, unsigned short* Wp, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = W[i];
    const unsigned u0 = __float_as_uint(v);
    const float r1 = v - __uint_as_float(u0 & 0xffff0000u);
    const unsigned u1 = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
    Wp[i] = u0 >> 16; Wp[n + i] = u1 >> 16; Wp[2 * n + i] = __float_as_uint(r2) >> 16;
}


// A [M][K] fp32 -> Ap [3][M][K] bf16 by the same pre-pass (stands for a producer epilogue that writes planes).
// stage: three A plane images of 128 rows x 32 bf16, then three W plane images of 256 rows x 32 bf16 (64 B rows, 4 chunks,
// swizzle (row >> 2) & 3): a fragment is three ds_read_b128, the K loop has no VALU work besides addresses.
template <int TA, int TB>
__global__ __launch_bounds__(512, 2) void kern(const unsigned short* Ap, const unsigned short* Wp, float* C, int K, int M) {
    constexpr int WGM = 128 / (32 * TA);
    constexpr int AP_BYTES = 128 * BK * 2, P_BYTES = 256 * BK * 2, A_BYTES = 3 * AP_BYTES, STAGE_BYTES = A_BYTES + 3 * P_BYTES;
    __shared__ __attribute__((aligned(1024))) unsigned char stages[2 * STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w % WGM, wn = w / WGM;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)&stages[0];
    const int lrow = lane >> 2;
    // A planes: 24 instructions of 16 rows (8 per plane); wave w issues w + 8 j, j < 3
    const unsigned short* asrc[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int g = w + 8 * j, plane = g >> 3, row = (g & 7) * 16 + lrow;
        asrc[j] = Ap + (size_t)plane * M * K + ((size_t)blockIdx.x * 128 + row) * K + (((lane & 3) ^ ((row >> 2) & 3)) << 3);
    }
    const unsigned short* wsrc[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const int g = w + 8 * j, plane = g >> 4, row = (g & 15) * 16 + lrow;
        wsrc[j] = Wp + (size_t)plane * 256 * K + (size_t)row * K + (((lane & 3) ^ ((row >> 2) & 3)) << 3);
    }
    auto issue = [&](int st, int k0) {
        const unsigned base = lds0 + st * STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < 3; j++) dma16(asrc[j] + k0, base + (w + 8 * j) * 1024);
#pragma unroll
        for (int j = 0; j < 6; j++) dma16(wsrc[j] + k0, base + A_BYTES + (w + 8 * j) * 1024);
    };
    const int fr = lane & 31, fh = lane >> 5;
    f32x16 acc[TA][TB] = {};
    const int S = K / BK;
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s = 0; s < S; s++) {
        if (s + 1 < S) issue((s + 1) & 1, (s + 1) * BK);
        const unsigned char* cur = stages + (s & 1) * STAGE_BYTES;
#pragma unroll
        for (int g = 0; g < 2; g++) {
            bf16x8 a0[TA], a1[TA], a2[TA];
#pragma unroll
            for (int i = 0; i < TA; i++) {
                const int row = wm * (32 * TA) + 32 * i + fr, c = 2 * g + fh, sw = (row >> 2) & 3;
                const unsigned char* p = cur + row * 64 + ((c ^ sw) << 4);
                a0[i] = *(const bf16x8*)p; a1[i] = *(const bf16x8*)(p + AP_BYTES); a2[i] = *(const bf16x8*)(p + 2 * AP_BYTES);
            }
#pragma unroll
            for (int j = 0; j < TB; j++) {
                const int row = wn * (32 * TB) + 32 * j + fr, c = 2 * g + fh, sw = (row >> 2) & 3;
                const unsigned char* p = cur + A_BYTES + row * 64 + ((c ^ sw) << 4);
                const bf16x8 b0 = *(const bf16x8*)p, b1 = *(const bf16x8*)(p + P_BYTES), b2 = *(const bf16x8*)(p + 2 * P_BYTES);
#define TERM(PA, PB) _Pragma("unroll") for (int i = 0; i < TA; i++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PA[i], PB, acc[i][j], 0, 0, 0);
                TERM(a2, b0) TERM(a0, b2) TERM(a1, b1) TERM(a1, b0) TERM(a0, b1) TERM(a0, b0)
#undef TERM
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TA; i++)
#pragma unroll
        for (int j = 0; j < TB; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = blockIdx.x * 128 + wm * (32 * TA) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                C[(size_t)row * 256 + wn * (32 * TB) + j * 32 + (lane & 31)] = acc[i][j][r];
            }
}

// 128 x 128 tile, 4 waves (2 x 2 of 64 x 64), BK = 16 planes (32 B rows, 2 chunks, swizzle (row >> 3) & 1):
// 24 KB per stage -> up to three workgroups per CU
__global__ __launch_bounds__(256, 3) void kern128(const unsigned short* Ap, const unsigned short* Wp, float* C, int K, int M) {
    constexpr int KB = 16;
    constexpr int P_BYTES = 128 * KB * 2, A_BYTES = 3 * P_BYTES, STAGE_BYTES = 2 * A_BYTES;
    __shared__ __attribute__((aligned(1024))) unsigned char stages[2 * STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w & 1, wn = w >> 1;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)&stages[0];
    const int lrow = lane >> 1;                                    // 32 rows per instruction
    const int tm = blockIdx.x >> 1, tn = blockIdx.x & 1;
    // per operand 12 instructions (4 per plane); wave w issues w + 4 j, j < 3, for A and for W
    const unsigned short *asrc[3], *wsrc[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int g = w + 4 * j, plane = g >> 2, row = (g & 3) * 32 + lrow;
        const int ch = ((lane & 1) ^ ((row >> 3) & 1)) << 3;
        asrc[j] = Ap + (size_t)plane * M * K + ((size_t)tm * 128 + row) * K + ch;
        wsrc[j] = Wp + (size_t)plane * 256 * K + (size_t)(tn * 128 + row) * K + ch;
    }
    auto issue = [&](int st, int k0) {
        const unsigned base = lds0 + st * STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < 3; j++) dma16(asrc[j] + k0, base + (w + 4 * j) * 1024);
#pragma unroll
        for (int j = 0; j < 3; j++) dma16(wsrc[j] + k0, base + A_BYTES + (w + 4 * j) * 1024);
    };
    const int fr = lane & 31, fh = lane >> 5;
    f32x16 acc[2][2] = {};
    const int S = K / KB;
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s = 0; s < S; s++) {
        if (s + 1 < S) issue((s + 1) & 1, (s + 1) * KB);
        const unsigned char* cur = stages + (s & 1) * STAGE_BYTES;
        bf16x8 a0[2], a1[2], a2[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int row = wm * 64 + 32 * i + fr, sw = (row >> 3) & 1;
            const unsigned char* p = cur + row * 32 + ((fh ^ sw) << 4);
            a0[i] = *(const bf16x8*)p; a1[i] = *(const bf16x8*)(p + P_BYTES); a2[i] = *(const bf16x8*)(p + 2 * P_BYTES);
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int row = wn * 64 + 32 * j + fr, sw = (row >> 3) & 1;
            const unsigned char* p = cur + A_BYTES + row * 32 + ((fh ^ sw) << 4);
            const bf16x8 b0 = *(const bf16x8*)p, b1 = *(const bf16x8*)(p + P_BYTES), b2 = *(const bf16x8*)(p + 2 * P_BYTES);
#define TERM(PA, PB) _Pragma("unroll") for (int i = 0; i < 2; i++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PA[i], PB, acc[i][j], 0, 0, 0);
            TERM(a2, b0) TERM(a0, b2) TERM(a1, b1) TERM(a1, b0) TERM(a0, b1) TERM(a0, b0)
#undef TERM
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = tm * 128 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                C[(size_t)row * 256 + tn * 128 + wn * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
            }
}

template <int TA, int TB>
float run(const unsigned short* Ap, const unsigned short* Wp, float* C, int NB, int K, int M) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 60; i++) kern<TA, TB><<<NB, 512>>>(Ap, Wp, C, K, M);
    (void)hipEventRecord(e0); for (int i = 0; i < 30; i++) kern<TA, TB><<<NB, 512>>>(Ap, Wp, C, K, M); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 30 * 1e3;
}
float run128(const unsigned short* Ap, const unsigned short* Wp, float* C, int NB, int K, int M) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 60; i++) kern128<<<2 * NB, 256>>>(Ap, Wp, C, K, M);
    (void)hipEventRecord(e0); for (int i = 0; i < 30; i++) kern128<<<2 * NB, 256>>>(Ap, Wp, C, K, M); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 30 * 1e3;
}

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 2048, NB = argc > 2 ? atoi(argv[2]) : 256, M = NB * 128;
    float *A, *W, *C0, *C1, *C2; unsigned short *Wp, *Ap;
    (void)hipMalloc(&A, (size_t)M * K * 4); (void)hipMalloc(&W, 256 * K * 4); (void)hipMalloc(&Wp, 3 * 256 * K * 2);
    (void)hipMalloc(&Ap, (size_t)3 * M * K * 2);
    (void)hipMalloc(&C0, (size_t)M * 256 * 4); (void)hipMalloc(&C1, (size_t)M * 256 * 4); (void)hipMalloc(&C2, (size_t)M * 256 * 4);
    float* h = (float*)malloc(((size_t)M * K + 31337 + 256 * K) * 4);
    srand(1);
    for (size_t i = 0; i < (size_t)M * K + 31337 + 256 * K; i++) h[i] = ((float)rand() / RAND_MAX - 0.5f) * 2.0f;
    (void)hipMemcpy(A, h, (size_t)M * K * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(W, h + 31337, 256 * K * 4, hipMemcpyHostToDevice);
    split_w<<<(256 * K + 255) / 256, 256>>>(W, Wp, 256 * K);
    split_w<<<(int)(((size_t)M * K + 255) / 256), 256>>>(A, Ap, M * K);
    const double fl = 2.0 * M * 256 * K;
    const float t22 = run<2, 2>(Ap, Wp, C0, NB, K, M), t14 = run<1, 4>(Ap, Wp, C1, NB, K, M), t128 = run128(Ap, Wp, C2, NB, K, M);
    printf("K=%d M=%d\n", K, M);
    printf("A+W planes, 128x256 tile, wave tile 2x2: %8.1f us  %6.1f TFLOP/s fp32-equivalent (%s)\n", t22, fl / t22 / 1e6, hipGetErrorString(hipGetLastError()));
    printf("A+W planes, 128x256 tile, wave tile 1x4: %8.1f us  %6.1f TFLOP/s fp32-equivalent\n", t14, fl / t14 / 1e6);
    printf("A+W planes, 128x128 tile x2, 4 waves, BK16, 3 wg/CU: %8.1f us  %6.1f TFLOP/s fp32-equivalent\n", t128, fl / t128 / 1e6);
    const size_t n = (size_t)64 * 256;
    float *c0 = (float*)malloc(n * 4), *c1 = (float*)malloc(n * 4), *c2 = (float*)malloc(n * 4);
    (void)hipMemcpy(c0, C0, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(c1, C1, n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(c2, C2, n * 4, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0, ref_max = 0;
    const float* hw = h + 31337;
    for (int r = 0; r < 64; r++)
        for (int c = 0; c < 256; c++) {
            double s = 0;
            for (int k = 0; k < K; k++) s += (double)h[(size_t)r * K + k] * (double)hw[(size_t)c * K + k];
            ref_max = fmax(ref_max, fabs(s));
            e0 = fmax(e0, fabs(c0[r * 256 + c] - s)); e1 = fmax(e1, fabs(c1[r * 256 + c] - s)); e2 = fmax(e2, fabs(c2[r * 256 + c] - s));
        }
    printf("max |error| vs fp64 (|ref| up to %.1f): 2x2 %.3g, 1x4 %.3g, 128 %.3g\n", ref_max, e0, e1, e2);
    return 0;
}
