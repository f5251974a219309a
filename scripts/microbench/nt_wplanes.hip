// Developer microbenchmark: fp32-grade NT product on the bf16 matrix cores with the WEIGHT operand split once.
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

// stage: A image 128 rows x 32 floats (128 B rows, 8 chunks, swizzle (row >> 1) & 7) then three W plane images of
// 256 rows x 32 bf16 (64 B rows, 4 chunks, swizzle (row >> 2) & 3)
template <int TA, int TB>
__global__ __launch_bounds__(512, 2) void kern(const float* A, const unsigned short* Wp, float* C, int K) {
    constexpr int WGM = 128 / (32 * TA);                     // wave grid rows; columns 8 / WGM
    constexpr int A_BYTES = 128 * BK * 4, P_BYTES = 256 * BK * 2, STAGE_BYTES = A_BYTES + 3 * P_BYTES;
    __shared__ __attribute__((aligned(1024))) unsigned char stages[2 * STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w % WGM, wn = w / WGM;
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)&stages[0];
    // A: 16 instructions of 8 rows; wave w issues instructions w and w + 8
    const float* asrc[2];
    {
        const int lrow = lane >> 3;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int row = (w + 8 * j) * 8 + lrow;
            asrc[j] = A + ((size_t)blockIdx.x * 128 + row) * K + (((lane & 7) ^ ((row >> 1) & 7)) << 2);
        }
    }
    // W planes: 48 instructions of 16 rows (16 per plane); wave w issues w + 8 j, j < 6
    const unsigned short* wsrc[6];
    {
        const int lrow = lane >> 2;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const int g = w + 8 * j, plane = g >> 4, row = (g & 15) * 16 + lrow;
            wsrc[j] = Wp + (size_t)plane * 256 * K + (size_t)row * K + (((lane & 3) ^ ((row >> 2) & 3)) << 3);
        }
    }
    auto issue = [&](int st, int k0) {
        const unsigned base = lds0 + st * STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < 2; j++) dma16(asrc[j] + k0, base + (w + 8 * j) * 1024);
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
            Split sa[TA];
#pragma unroll
            for (int i = 0; i < TA; i++) {
                const int row = wm * (32 * TA) + 32 * i + fr, c = 4 * g + 2 * fh, sw = (row >> 1) & 7;
                const float4 lo = *(const float4*)(cur + row * 128 + ((c ^ sw) << 4));
                const float4 hi = *(const float4*)(cur + row * 128 + (((c + 1) ^ sw) << 4));
                sa[i] = split8t(lo, hi);
            }
#pragma unroll
            for (int j = 0; j < TB; j++) {
                const int row = wn * (32 * TB) + 32 * j + fr, c = 2 * g + fh, sw = (row >> 2) & 3;
                const unsigned char* p = cur + A_BYTES + row * 64 + ((c ^ sw) << 4);
                const bf16x8 b0 = *(const bf16x8*)p, b1 = *(const bf16x8*)(p + P_BYTES), b2 = *(const bf16x8*)(p + 2 * P_BYTES);
#define TERM(PA, PB) _Pragma("unroll") for (int i = 0; i < TA; i++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa[i].PA, PB, acc[i][j], 0, 0, 0);
                TERM(p2, b0) TERM(p0, b2) TERM(p1, b1) TERM(p1, b0) TERM(p0, b1) TERM(p0, b0)
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

template <int TA, int TB>
float run(const float* A, const unsigned short* Wp, float* C, int NB, int K) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 60; i++) kern<TA, TB><<<NB, 512>>>(A, Wp, C, K);
    (void)hipEventRecord(e0); for (int i = 0; i < 30; i++) kern<TA, TB><<<NB, 512>>>(A, Wp, C, K); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 30 * 1e3;
}

int main() {
    const int K = 2048, NB = 256, M = NB * 128;
    float *A, *W, *C0, *C1; unsigned short* Wp;
    (void)hipMalloc(&A, (size_t)M * K * 4); (void)hipMalloc(&W, 256 * K * 4); (void)hipMalloc(&Wp, 3 * 256 * K * 2);
    (void)hipMalloc(&C0, (size_t)M * 256 * 4); (void)hipMalloc(&C1, (size_t)M * 256 * 4);
    float* h = (float*)malloc((size_t)M * K * 4);
    srand(1);
    for (size_t i = 0; i < (size_t)M * K; i++) h[i] = ((float)rand() / RAND_MAX - 0.5f) * 2.0f;
    (void)hipMemcpy(A, h, (size_t)M * K * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(W, h + 31337, 256 * K * 4, hipMemcpyHostToDevice);
    split_w<<<(256 * K + 255) / 256, 256>>>(W, Wp, 256 * K);
    const double fl = 2.0 * M * 256 * K;
    const float t22 = run<2, 2>(A, Wp, C0, NB, K), t14 = run<1, 4>(A, Wp, C1, NB, K);
    printf("W planes, wave tile 2x2 blocks: %8.1f us  %6.1f TFLOP/s fp32-equivalent (%s)\n", t22, fl / t22 / 1e6, hipGetErrorString(hipGetLastError()));
    printf("W planes, wave tile 1x4 blocks: %8.1f us  %6.1f TFLOP/s fp32-equivalent\n", t14, fl / t14 / 1e6);
    const size_t n = (size_t)64 * 256;
    float *c0 = (float*)malloc(n * 4), *c1 = (float*)malloc(n * 4);
    (void)hipMemcpy(c0, C0, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(c1, C1, n * 4, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, ref_max = 0;
    const float* hw = h + 31337;
    for (int r = 0; r < 64; r++)
        for (int c = 0; c < 256; c++) {
            double s = 0;
            for (int k = 0; k < K; k++) s += (double)h[(size_t)r * K + k] * (double)hw[(size_t)c * K + k];
            ref_max = fmax(ref_max, fabs(s));
            e0 = fmax(e0, fabs(c0[r * 256 + c] - s)); e1 = fmax(e1, fabs(c1[r * 256 + c] - s));
        }
    printf("max |error| vs fp64 (|ref| up to %.1f): 2x2 %.3g, 1x4 %.3g\n", ref_max, e0, e1);
    return 0;
}
