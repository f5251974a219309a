// 16 x 16 output blocks on v_mfma_f32_16x16x4_f32 (exact fp32 fma chains): the small products of the joint step's tile
// kernels and of the attention block's projection chains -- 16-row tiles put 256 workgroups on the chip at B = 4096.
#pragma once
#include "common.h"

typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4v mfma16(float a, float b, f32x4v c) {
#ifdef PC_EXP_NO_MFMA
    // developer build (scripts/dev/joint_mfma_knockout.sh; WRONG results): the product replaced by a keep-alive of its operands --
    // what is left is everything a faster matrix instruction could NOT shorten
    asm volatile("" ::"v"(a), "v"(b));
    return c;
#else
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}

// ---- 16 x 16 output blocks on v_mfma_f32_16x16x4_f32 ----------------------------------------------------------
// A operand: LDS rows (row stride lda floats), B operand: global weights.  k runs in chunks of 16; inside a chunk
// lane (i = lane & 15, h = lane >> 4) owns k = 16 c + 4 h + q, q = 0..3, for BOTH operands (a sum over k does not
// care about the order, so one 16-B read feeds four MFMAs).  MB row blocks share one B fragment set.
// NT: B[k][j] = W[(n0 + j) * ldw + k]   (y = x W^T, nn.Linear forward)
// NN: B[k][j] = W[k * ldw + n0 + j]     (dx = dy W)
// The B fragments of EVERY phase are requested at the top of the kernel (weights do not depend on the tile): the phases
// then wait for LDS hand-offs only, not for one L2 / HBM round trip each.
template <int NC> struct BFrag { float4 v[NC]; };

template <int K, bool NN>
__device__ __forceinline__ BFrag<K / 16> load_b(const float* W, int ldw, int n0, int nvalid, int lane) {
    const int i = lane & 15, h = lane >> 4;
    constexpr int NC = K / 16;
    BFrag<NC> b;
    const bool v = n0 + i < nvalid;
    if (NN) {
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const float* w = W + (size_t)(16 * c + 4 * h) * ldw + (v ? n0 + i : 0);
            b.v[c] = v ? make_float4(w[0], w[ldw], w[2 * ldw], w[3 * ldw]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else {
        const float* w = W + (size_t)(v ? n0 + i : 0) * ldw + 4 * h;
#pragma unroll
        for (int c = 0; c < NC; c++)
            b.v[c] = v ? *reinterpret_cast<const float4*>(w + 16 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return b;
}

template <int NC, int MB>
__device__ __forceinline__ void mul_b(const float* As, int lda, int mblocks, const BFrag<NC>& b, f32x4v (&acc)[MB], int lane) {
    const int i = lane & 15, h = lane >> 4;
#pragma unroll
    for (int c = 0; c < NC; c++) {
#pragma unroll
        for (int m = 0; m < MB; m++) {
            if (m < mblocks) {                       // (folds away when the caller passes a compile-time count == MB)
                const float4 a4 = *reinterpret_cast<const float4*>(As + (size_t)(16 * m + i) * lda + 4 * h + 16 * c);
                acc[m] = mfma16(a4.x, b.v[c].x, acc[m]);
                acc[m] = mfma16(a4.y, b.v[c].y, acc[m]);
                acc[m] = mfma16(a4.z, b.v[c].z, acc[m]);
                acc[m] = mfma16(a4.w, b.v[c].w, acc[m]);
            }
        }
    }
}
template <int K, int MB, bool NN>
__device__ __forceinline__ void block_product(const float* As, int lda, int mblocks, const float* W, int ldw, int n0,
                                              int nvalid, f32x4v (&acc)[MB], int lane) {
    const BFrag<K / 16> b = load_b<K, NN>(W, ldw, n0, nvalid, lane);
    mul_b<K / 16, MB>(As, lda, mblocks, b, acc, lane);
}
// result element (block m, register r) of lane: row 16 m + 4 (lane >> 4) + r, column n0 + (lane & 15)

