// dW[No,Ni] (+)= sum_r Z[r][o] * prologue(A)[r][i],  db[o] (+)= sum_r Z[r][o]
//
// The weight-gradient products of nn.Linear / in_proj / out_proj backward (what autograd
// derives for product2vec.py:14-28, type_transition.py:11-12, item_prediction.py:11-20):
// the contraction runs over ROWS (hundreds of thousands) and the output is tiny, so the row
// range is split over workgroups, each writes its fp32 partial tile to a slab and a second
// pass sums the slabs in a fixed order (bitwise reproducible, no float atomics).
//
// Both operands are read exactly as they lie in HBM (row-major, one 512-B row segment per
// 32 lanes), staged [32 rows][128] in LDS and consumed by v_mfma_f32_32x32x2_f32 with the
// ROW index as the MFMA k: lane l reads element [k0 + (l>>5)][tile + (l&31)] -- 32
// consecutive floats per half-wave, conflict-free ds_read_b32.
#include <atomic>
#include <mutex>
#include "common.h"

#define TM 128
#define TK 32

__device__ __forceinline__ void tn_small_body(const TnArgs& a, int tiles_i, int nsplit, int rows_per_split, int bid) {
    __shared__ __attribute__((aligned(16))) float smem[2][2][TK * TM];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wo = w >> 1, wi = w & 1;
    const int split = bid % nsplit;
    const int tile = bid / nsplit;
    const int o0 = (tile / tiles_i) * TM, i0 = (tile % tiles_i) * TM;
    const int r_begin = split * rows_per_split;
    const int r_end = min(a.R, r_begin + rows_per_split);

    const int lrow = tid >> 5, lcol = (tid & 31) * 4;   // 8 rows per pass, 4 passes
    const bool zcol_ok = o0 + lcol < a.No, acol_ok = i0 + lcol < a.Ni;   // No, Ni multiples of 4

    float4 rz[4], ra[4];
    float4 dbacc = make_float4(0.f, 0.f, 0.f, 0.f);
    // BN-backward coefficients of this thread's 4 columns, cached for one segment at a time
    int zseg = -1;
    float4 zmu, zis, zsc, zc1, zc2;
    auto gload = [&](int r0) {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int r = r0 + lrow + 8 * p;
            const bool rv = r < r_end;
            if (a.z_onehot) {
                const int t = rv ? a.z_onehot[r] : -1, c = o0 + lcol;
                rz[p] = make_float4(t == c ? 1.f : 0.f, t == c + 1 ? 1.f : 0.f, t == c + 2 ? 1.f : 0.f, t == c + 3 ? 1.f : 0.f);
            } else {
                rz[p] = (rv && zcol_ok) ? *reinterpret_cast<const float4*>(a.Z + (size_t)r * a.ldz + o0 + lcol)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (a.zaux && rv && zcol_ok) {
                const int s = seg_of_row(a.seg, r);
                if (s != zseg) {
                    const size_t o = (size_t)s * a.No + o0 + lcol;
                    zmu = *reinterpret_cast<const float4*>(a.z_mean + o);
                    zis = *reinterpret_cast<const float4*>(a.z_invstd + o);
                    zsc = *reinterpret_cast<const float4*>(a.z_scale + o);
                    zc1 = *reinterpret_cast<const float4*>(a.z_c1 + o);
                    zc2 = *reinterpret_cast<const float4*>(a.z_c2 + o);
                    zseg = s;
                }
                const float4 h = *reinterpret_cast<const float4*>(a.zaux + (size_t)r * a.ldzaux + o0 + lcol);
                const float zm = row_multiplicity(a.seg, r);
                rz[p].x = zsc.x * (rz[p].x - zm * (zc1.x + (h.x - zmu.x) * zis.x * zc2.x));
                rz[p].y = zsc.y * (rz[p].y - zm * (zc1.y + (h.y - zmu.y) * zis.y * zc2.y));
                rz[p].z = zsc.z * (rz[p].z - zm * (zc1.z + (h.z - zmu.z) * zis.z * zc2.z));
                rz[p].w = zsc.w * (rz[p].w - zm * (zc1.w + (h.w - zmu.w) * zis.w * zc2.w));
            }
            int src = r;
            bool av = rv && acol_ok;
            if (av && a.gather) { src = a.gather[r]; av = src >= 0; }
            ra[p] = av ? *reinterpret_cast<const float4*>(a.A + (size_t)src * a.lda + i0 + lcol)
                       : make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.prologue == NT_PRO_BNTANH && rv && acol_ok) {
                const int s = seg_of_row(a.seg, r);
                const float4 sc = *reinterpret_cast<const float4*>(a.pscale + (size_t)s * a.Ni + i0 + lcol);
                const float4 sh = *reinterpret_cast<const float4*>(a.pshift + (size_t)s * a.Ni + i0 + lcol);
                ra[p].x = fast_tanh(ra[p].x * sc.x + sh.x);
                ra[p].y = fast_tanh(ra[p].y * sc.y + sh.y);
                ra[p].z = fast_tanh(ra[p].z * sc.z + sh.z);
                ra[p].w = fast_tanh(ra[p].w * sc.w + sh.w);
            }
            dbacc.x += rz[p].x; dbacc.y += rz[p].y; dbacc.z += rz[p].z; dbacc.w += rz[p].w;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            *reinterpret_cast<float4*>(&smem[buf][0][(lrow + 8 * p) * TM + lcol]) = rz[p];
            *reinterpret_cast<float4*>(&smem[buf][1][(lrow + 8 * p) * TM + lcol]) = ra[p];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int nchunk = (r_end - r_begin + TK - 1) / TK;
    if (nchunk > 0) {
        gload(r_begin);
        lstore(0);
    }
    __syncthreads();
    const int fz = (lane >> 5) * TM + wo * 64 + (lane & 31);
    const int fa = (lane >> 5) * TM + wi * 64 + (lane & 31);
    // 32x32 blocks of this wave that lie inside [No, Ni] (wave-uniform): a small gradient (the joint step's 64x32 / 32x64
    // layers, its [T,64] one-hot table products) would otherwise spend most of the tile's MFMAs on zeros
    const bool vz0 = __builtin_amdgcn_readfirstlane(o0 + wo * 64 < a.No), vz1 = __builtin_amdgcn_readfirstlane(o0 + wo * 64 + 32 < a.No);
    const bool vx0 = __builtin_amdgcn_readfirstlane(i0 + wi * 64 < a.Ni), vx1 = __builtin_amdgcn_readfirstlane(i0 + wi * 64 + 32 < a.Ni);
    for (int c = 0; c < nchunk; c++) {
        const int cur = c & 1;
        if (c + 1 < nchunk) gload(r_begin + (c + 1) * TK);
        const float* Zs = &smem[cur][0][fz];
        const float* As = &smem[cur][1][fa];
#pragma unroll
        for (int k = 0; k < TK; k += 2) {
            const float z0 = Zs[k * TM], z1 = Zs[k * TM + 32];
            const float x0 = As[k * TM], x1 = As[k * TM + 32];
            if (vz0 && vx0) acc[0][0] = mfma32(z0, x0, acc[0][0]);
            if (vz0 && vx1) acc[0][1] = mfma32(z0, x1, acc[0][1]);
            if (vz1 && vx0) acc[1][0] = mfma32(z1, x0, acc[1][0]);
            if (vz1 && vx1) acc[1][1] = mfma32(z1, x1, acc[1][1]);
        }
        if (c + 1 < nchunk) lstore(cur ^ 1);
        __syncthreads();
    }

    float* slab = a.slabs + (size_t)split * ((size_t)a.No * a.Ni + a.No);
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            const int i = i0 + wi * 64 + nt * 32 + (lane & 31);
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const int o = o0 + wo * 64 + mt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                if (o < a.No && i < a.Ni) slab[(size_t)o * a.Ni + i] = acc[mt][nt][reg];
            }
        }
    if (a.db && i0 == 0) {
        // fold the 8 loader row-groups (tid>>5) holding the same 4 columns
        float* red = &smem[0][0][0];
        *reinterpret_cast<float4*>(&red[lrow * TM + lcol]) = dbacc;
        __syncthreads();
        if (tid < TM && o0 + tid < a.No) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < 8; g++) s += red[g * TM + tid];
            slab[(size_t)a.No * a.Ni + o0 + tid] = s;
        }
    }
}

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(TnArgs a, int tiles_i, int nsplit, int rows_per_split) {
    tn_small_body(a, tiles_i, nsplit, rows_per_split, blockIdx.x);
}

// Several small, mutually independent gradients in ONE launch (the four Linear layers of the joint step: each alone
// is 64 workgroups and ~13 us of pure latency): workgroup ranges [block0[j], block0[j+1]) run product j.
__global__ __launch_bounds__(256, 2) void gemm_tn_group_kernel(TnGroup g) {
    const int b = blockIdx.x;
    int j = 0;
#pragma unroll
    for (int i = 1; i < PC_TN_GROUP; i++) j += (i < g.n && b >= g.block0[i]) ? 1 : 0;
    tn_small_body(g.a[j], g.tiles_i[j], g.nsplit[j], g.rps[j], b - g.block0[j]);
}

// ---------------------------------------------------------------------------------------
// Full-tile variant for the FFN / attention weight gradients: ONE 8-wave workgroup per CU
// computes the WHOLE dW (up to 256x256) for its slice of rows, so each operand row is read
// from HBM exactly once (the 128x128-tile kernel above reads Z once per column tile and A
// once per row tile: 2x the traffic for a 256x256 gradient, and these products sit near the
// HBM roofline at the fp32 MFMA rate).  Wave grid WO x WI, each wave TO x TI MFMA tiles.
//   <2,4,4,2> 256x256 (dW3)   <2,4,2,2> 128x256 (dW5)   <4,2,2,2> 256x128 (dW0, dW_kv)
// The operands go global -> LDS directly (global_load_lds_dwordx4, see gemm_nt.hip): the stage is
// [TK rows][NO | NI] row-major exactly as the rows lie in HBM, so one wave instruction drops 1 KB
// of consecutive row floats and the MFMA fragments are conflict-free ds_read_b32 (32 consecutive
// floats per half-wave).  The fused loaders run IN PLACE on the landed stage:
//   APRO  A' = tanh(A * scale_s + shift_s)                       (dW3: A1 recomputed from H0)
//   ZPRO  Z' = scale_s * (Z - m (c1_s + (H - mean_s) invstd_s c2_s))  (dW0: BatchNorm backward of dZ1,
//         H = H0 staged beside Z)
// db is folded from the Z fragments the wi == 0 waves read anyway.
typedef __attribute__((address_space(3))) void* tn_lptr_t;
__device__ __attribute__((aligned(64))) float pc_tn_zero_chunk[16];
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void tn_dma16(const float* gsrc, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_addr) : "memory", "m0");
}
#pragma clang diagnostic pop
__device__ __forceinline__ void tn_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// halves == 2 (round 4: dW3): ONE launch computes the two NO-wide halves of a 2 NO-wide gradient over nsplit row slices each.
// Workgroups b and b + 8 -- the same XCD: blockIdx round-robins over the eight -- take the two halves of ONE row slice, so the
// second fetch of the slice's A rows is an L2 hit (two launches of 256 slices each read A twice from HBM and wrote twice the
// slabs).  Half h reads Z + h * NO columns and writes its slabs at a.slabs + h * slab_half_off.
template <int WO, int WI, int TO, int TI, int TKC, int NST, bool APRO, bool ZPRO>
__global__ __launch_bounds__(64 * WO * WI, WO * WI / 4) void gemm_tn8_kernel(TnArgs a, int nsplit, int rows_per_split, int halves,
                                                                             size_t slab_half_off) {
    constexpr int NWV = WO * WI, THREADS = 64 * NWV;            // 8 waves (2 per SIMD) or 16 (4 per SIMD)
    constexpr int NO = WO * TO * 32, NI = WI * TI * 32;
    // Image rows must not start on the same LDS bank two rows apart in the MFMA's k pair (lanes 0-31 read row k,
    // lanes 32-63 row k+1, same columns): a 256-wide image is stored with row stride 288 floats (each wave
    // instruction is one row: it is simply dropped 128 B further), a 128-wide one keeps stride 128 and swaps the
    // two 32-float halves of every 64 in its ODD rows (the lanes of the second row of an instruction fetch the
    // other half; the fragment / in-place readers of odd rows flip bit 5 of the column).
    // (bf16 products: lanes 0-31 read rows k .. k+7, lanes 32-63 rows k+8 .. k+15 of a 16-row group; the two
    // half-waves must sit on different banks: 256-wide images get row stride 260 floats (8 x 260 = 32 mod 64), 128-wide
    // ones swap the 32-float halves of every 64 in the rows whose bit 3 is set)
    constexpr int ZRS = NO == 256 ? 260 : NO, ARS = NI == 256 ? 260 : NI;     // row strides
    constexpr bool ZSW = NO == 128, ASW = NI == 128;                          // half swap of rows with bit 3 set
    static_assert((NO == 128 || NO == 256) && (NI == 128 || NI == 256), "image widths");
    constexpr int ZF = TKC * ZRS, AF = TKC * ARS;                // floats per stage image
    constexpr int STAGE = ZF * (ZPRO ? 2 : 1) + AF;              // Z [, H], A
    constexpr int JZ = TKC * NO / 256 / NWV, JA = TKC * NI / 256 / NWV;      // DMA instructions per wave and image
    static_assert((TKC * NO) % (256 * NWV) == 0 && (TKC * NI) % (256 * NWV) == 0, "images must split into whole wave instructions over the waves");
    constexpr int JALL = JZ * (ZPRO ? 2 : 1) + JA;              // DMA instructions per wave and stage
    __shared__ __attribute__((aligned(1024))) float smem[NST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wo = w % WO, wi = w / WO;
    const int half = halves == 2 ? ((int)blockIdx.x >> 3) & 1 : 0;
    const int split = halves == 2 ? (((int)blockIdx.x >> 4) << 3) + ((int)blockIdx.x & 7) : (int)blockIdx.x;
    if (halves == 2) {                                           // (uniform: scalar adds on the kernel arguments)
        a.Z += half * (WO * TO * 32);
        a.slabs += half * slab_half_off;
        if (a.db) a.db += half * (WO * TO * 32);
    }
    const int r_begin = split * rows_per_split;
    const int r_end = min(a.R, r_begin + rows_per_split);
    const float* const zsrc0 = pc_tn_zero_chunk;
    const unsigned lds0 = (unsigned)(uintptr_t)(tn_lptr_t)&smem[0];

    // DMA geometry: instruction g = w + NWV j of an image covers its floats [256 g, 256 g + 256)
    // (zc / ac: the GLOBAL column this lane fetches; zd / ad: byte offset of the instruction inside the image)
    int zr[JZ], zc[JZ], ar[JA], ac[JA];
    unsigned zd[JZ], ad[JA];
#pragma unroll
    for (int j = 0; j < JZ; j++) {
        const int g = w + NWV * j, f = g * 256 + lane * 4;
        zr[j] = f / NO; zc[j] = f % NO;
        if (ZSW && (zr[j] & 8)) zc[j] ^= 32;
        zd[j] = (NO == 256 ? g * ZRS : g * 256) * 4;
    }
#pragma unroll
    for (int j = 0; j < JA; j++) {
        const int g = w + NWV * j, f = g * 256 + lane * 4;
        ar[j] = f / NI; ac[j] = f % NI;
        if (ASW && (ar[j] & 8)) ac[j] ^= 32;
        ad[j] = (NI == 256 ? g * ARS : g * 256) * 4;
    }
    int gidx[JA];                                                // gather indices of the NEXT chunk to issue
    auto load_gather = [&](int r0) {
#pragma unroll
        for (int j = 0; j < JA; j++) {
            const int r = r0 + ar[j];
            gidx[j] = (a.gather && r < r_end) ? a.gather[r] : r;
        }
    };
    auto issue = [&](int st, int r0) {
        const unsigned base = lds0 + st * (STAGE * 4);
#pragma unroll
        for (int j = 0; j < JZ; j++) {
            const int r = r0 + zr[j];
            const bool v = r < r_end && zc[j] < a.No;
            tn_dma16(v ? a.Z + (size_t)r * a.ldz + zc[j] : zsrc0, base + zd[j]);
            if (ZPRO) tn_dma16(v ? a.zaux + (size_t)r * a.ldzaux + zc[j] : zsrc0, base + ZF * 4 + zd[j]);
        }
#pragma unroll
        for (int j = 0; j < JA; j++) {
            const int r = r0 + ar[j];
            const bool v = r < r_end && ac[j] < a.Ni && gidx[j] >= 0;
            tn_dma16(v ? a.A + (size_t)gidx[j] * a.lda + ac[j] : zsrc0, base + (ZPRO ? 2 : 1) * ZF * 4 + ad[j]);
        }
    };

    // in-place loaders: thread t owns 4 fixed columns and every (THREADS / (N/4))-th row of an image
    constexpr int ZT = NO / 4, AT = NI / 4;
    const int zrow = tid / ZT, zcol = (tid % ZT) * 4, arow = tid / AT, acol = (tid % AT) * 4;
    int zseg = -1;
    float4 zmu, zis, zsc, zc1, zc2;
    auto transform = [&](float* stage, int r0) {
        if (ZPRO && zcol < a.No) {
            float* Zs = stage;
            const float* Hs = stage + ZF;
#pragma unroll
            for (int p = 0; p < TKC / (THREADS / ZT); p++) {
                const int rr = zrow + (THREADS / ZT) * p, r = r0 + rr;
                if (r < r_end) {
                    const int s = seg_of_row(a.seg, r);
                    if (s != zseg) {
                        const size_t o = (size_t)s * a.No + zcol;
                        zmu = *reinterpret_cast<const float4*>(a.z_mean + o);
                        zis = *reinterpret_cast<const float4*>(a.z_invstd + o);
                        zsc = *reinterpret_cast<const float4*>(a.z_scale + o);
                        zc1 = *reinterpret_cast<const float4*>(a.z_c1 + o);
                        zc2 = *reinterpret_cast<const float4*>(a.z_c2 + o);
                        zseg = s;
                    }
                    const int zp = rr * ZRS + ((ZSW && (rr & 8)) ? zcol ^ 32 : zcol);
                    float4 z = *reinterpret_cast<const float4*>(&Zs[zp]);
                    const float4 h = *reinterpret_cast<const float4*>(&Hs[zp]);
                    const float zm = row_multiplicity(a.seg, r);
                    z.x = zsc.x * (z.x - zm * (zc1.x + (h.x - zmu.x) * zis.x * zc2.x));
                    z.y = zsc.y * (z.y - zm * (zc1.y + (h.y - zmu.y) * zis.y * zc2.y));
                    z.z = zsc.z * (z.z - zm * (zc1.z + (h.z - zmu.z) * zis.z * zc2.z));
                    z.w = zsc.w * (z.w - zm * (zc1.w + (h.w - zmu.w) * zis.w * zc2.w));
                    *reinterpret_cast<float4*>(&Zs[zp]) = z;
                }
            }
        }
        if (APRO && acol < a.Ni) {
            float* As = stage + (ZPRO ? 2 : 1) * ZF;
            // scale / shift are re-fetched per chunk (L1 hits) instead of living in 8 registers across the MFMA phase
            int aseg = -1;
            float4 asc, ash;
#pragma unroll
            for (int p = 0; p < TKC / (THREADS / AT); p++) {
                const int rr = arow + (THREADS / AT) * p, r = r0 + rr;
                if (r < r_end) {
                    const int s = seg_of_row(a.seg, r);
                    if (s != aseg) {
                        asc = *reinterpret_cast<const float4*>(a.pscale + (size_t)s * a.Ni + acol);
                        ash = *reinterpret_cast<const float4*>(a.pshift + (size_t)s * a.Ni + acol);
                        aseg = s;
                    }
                    const int ap = rr * ARS + ((ASW && (rr & 8)) ? acol ^ 32 : acol);
                    float4 x = *reinterpret_cast<const float4*>(&As[ap]);
                    x.x = fast_tanh(x.x * asc.x + ash.x);
                    x.y = fast_tanh(x.y * asc.y + ash.y);
                    x.z = fast_tanh(x.z * asc.z + ash.z);
                    x.w = fast_tanh(x.w * asc.w + ash.w);
                    *reinterpret_cast<float4*>(&As[ap]) = x;
                }
            }
        }
    };

    f32x16 acc[TO][TI];
    float zsum[TO];
#pragma unroll
    for (int i = 0; i < TO; i++) {
        zsum[i] = 0.f;
#pragma unroll
        for (int j = 0; j < TI; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    }

    // ring of NST stages: chunk c lives in slot c % NST and NST-1 chunks are in flight ahead of the one being
    // multiplied.  In-order vmcnt: "chunk c has landed" == at most the (NST-2) * JALL younger DMA instructions
    // are outstanding; near the end fewer are in flight and the wait is for everything.  (Measured: 4-6 stages of
    // 16 rows are no faster than 2 of 32 -- these kernels are not waiting for HBM; a row sweep shows a fixed
    // 28 / 62 us per launch for a 128x256 / 256x256 gradient, i.e. slab write + reduce + ramp, and 107-113
    // TFLOP/s in the limit of many rows, scripts/tn_rsweep.py.)
    const int nchunk = (r_end - r_begin + TKC - 1) / TKC;
#pragma unroll
    for (int i = 0; i < NST - 1; i++)
        if (i < nchunk) { load_gather(r_begin + i * TKC); issue(i, r_begin + i * TKC); }
    if (NST - 1 < nchunk) load_gather(r_begin + (NST - 1) * TKC);
    // lanes 32-63 read the odd row of each pair: in a half-swapped image their columns have bit 5 flipped
    // bf16 products (see common.h): one MFMA k group = 16 rows; lane (column lane & 31, half lane >> 5) holds the 8
    // rows 8 (lane >> 5) .. + 7 of its column, read one by one (row stride), split into three bf16 pieces once and
    // used by TI (Z blocks) / TO (A blocks) accumulators.
    static_assert(TKC % 16 == 0, "chunks are whole k groups");
    const int fz = 8 * (lane >> 5) * ZRS, fa = 8 * (lane >> 5) * ARS;
    int zo[TO], xo[TI];                                        // this lane's column of each 32-wide block
#pragma unroll
    for (int i = 0; i < TO; i++) zo[i] = (wo * (TO * 32) + 32 * i + (lane & 31)) ^ ((ZSW && lane >= 32) ? 32 : 0);
#pragma unroll
    for (int j = 0; j < TI; j++) xo[j] = (wi * (TI * 32) + 32 * j + (lane & 31)) ^ ((ASW && lane >= 32) ? 32 : 0);
    int cur = 0;
    for (int c = 0; c < nchunk; c++) {
#ifndef PC_EXP_NO_VMWAIT
        if (c + NST - 1 <= nchunk) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * JALL) : "memory"); }
        else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
#ifndef PC_EXP_NO_BARRIER
        __syncthreads();
#endif
                                          // chunk c visible to all; slot of chunk c-1 free again
#ifndef PC_EXP_NO_DMA
        if (c + NST - 1 < nchunk) {
            issue(cur == 0 ? NST - 1 : cur - 1, r_begin + (c + NST - 1) * TKC);
            if (c + NST < nchunk) load_gather(r_begin + (c + NST) * TKC);   // indices a whole chunk ahead of their use
        }
#endif
        float* stage = smem + cur * STAGE;
        if (APRO || ZPRO) { transform(stage, r_begin + c * TKC); tn_lds_sync(); }
        const float* Zs = stage + fz;
        const float* As = stage + (ZPRO ? 2 : 1) * ZF + fa;
#pragma unroll
        for (int g = 0; g < TKC / 16; g++) {
            // rows 16 g + 8 (lane >> 5) + t; in a half-swapped 128-wide image the swap follows bit 3 of the row, i.e.
            // the lane's half when g is even and its complement when... rows 16g+8h+t have bit 3 == h: constant per lane
            // developer builds (scripts/dev/nt_decompose.sh, KERNELS=tn; WRONG results): -DPC_EXP_NO_SPLIT, -DPC_EXP_NO_LDSREAD,
            // -DPC_EXP_NO_MFMA, -DPC_EXP_NO_DMA as in gemm_nt.hip; -DPC_EXP_NO_SLAB skips the slab store
#if defined(PC_EXP_NO_LDSREAD)
#define PC_TNV(PTR, OFF) __int_as_float(a.R + t)
#else
#define PC_TNV(PTR, OFF) (PTR)[OFF]
#endif
#if defined(PC_EXP_NO_SPLIT) || defined(PC_EXP_NO_LDSREAD)
#define PC_TNSPLIT(LO, HI) Split3{__builtin_bit_cast(bf16x8, LO), __builtin_bit_cast(bf16x8, HI), __builtin_bit_cast(bf16x8, LO)}
#else
#define PC_TNSPLIT(LO, HI) split3(LO, HI)
#endif
#if defined(PC_EXP_NO_MFMA)
#define PC_TNMFMA(A, B, C) ([&]() { asm volatile("" ::"v"(A), "v"(B)); return C; }())
#else
#define PC_TNMFMA(A, B, C) mfma_bf16(A, B, C)
#endif
            Split3 sx[TI];
#pragma unroll
            for (int j = 0; j < TI; j++) {
                float v[8];
#pragma unroll
                for (int t = 0; t < 8; t++) v[t] = PC_TNV(As, (16 * g + t) * ARS + xo[j]);
                sx[j] = PC_TNSPLIT(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
            }
#pragma unroll
            for (int i = 0; i < TO; i++) {
                float v[8];
#pragma unroll
                for (int t = 0; t < 8; t++) v[t] = PC_TNV(Zs, (16 * g + t) * ZRS + zo[i]);
                zsum[i] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                const Split3 sz = PC_TNSPLIT(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
#define PC_TERM(PZ, PX) _Pragma("unroll") for (int j = 0; j < TI; j++) acc[i][j] = PC_TNMFMA(sz.PZ, sx[j].PX, acc[i][j]);
                PC_PRIO_MFMA(PC_PRIO_TN_COND, 1);
                PC_TERM(p2, p0) PC_TERM(p0, p2) PC_TERM(p1, p1) PC_TERM(p1, p0) PC_TERM(p0, p1) PC_TERM(p0, p0)
                PC_PRIO_MFMA(PC_PRIO_TN_COND, 0);
#undef PC_TERM
            }
        }
        cur = cur + 1 == NST ? 0 : cur + 1;
    }

    float* slab = a.slabs + (size_t)split * ((size_t)a.No * a.Ni + a.No);
#pragma unroll
    for (int i = 0; i < TO; i++)
#pragma unroll
        for (int j = 0; j < TI; j++) {
            const int ci = wi * (TI * 32) + j * 32 + (lane & 31);
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const int o = wo * (TO * 32) + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
#ifdef PC_EXP_NO_SLAB
                if (o < a.No && ci < a.Ni && acc[i][j][reg] == 12345.678f) slab[(size_t)o * a.Ni + ci] = acc[i][j][reg];
#else
                if (o < a.No && ci < a.Ni) slab[(size_t)o * a.Ni + ci] = acc[i][j][reg];
#endif
            }
        }
    if (a.db && wi == 0) {
        // lanes l and l+32 hold the even / odd rows of column (l & 31)
#pragma unroll
        for (int i = 0; i < TO; i++) {
            const float s = zsum[i] + __shfl_xor(zsum[i], 32, 64);
            const int o = wo * (TO * 32) + i * 32 + (lane & 31);
            if (lane < 32 && o < a.No) slab[(size_t)a.No * a.Ni + o] = s;
        }
    }
}

// out[j] (+)= sum_s slabs[s][j], j over [No*Ni] then [No] (bias).  Eight lanes share one float4 of
// outputs: lane g sums slabs g, g+8, ... and the eight partial sums fold in a fixed xor order
// (bitwise reproducible), so a 256x256 gradient keeps ~500 workgroups streaming from HBM.
__device__ __forceinline__ void adam_rider4(const AdamRider& ad, size_t off, const float4 gv, float step_size, float bc2s) {
    float4 pv = *reinterpret_cast<float4*>(ad.p + off);
    float4 mv = *reinterpret_cast<float4*>(ad.m + off);
    float4 vv = *reinterpret_cast<float4*>(ad.v + off);
    if (pc_adam_dead(gv, mv, vv)) return;                   // (as adam_at_kernel: g = m = v = 0 changes nothing)
#define ADAM1(c) pc_adam_update(pv.c, mv.c, vv.c, gv.c, step_size, bc2s, ad.omb1, ad.beta2f, ad.omb2, ad.eps);
    ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
#undef ADAM1
    *reinterpret_cast<float4*>(ad.p + off) = pv;
    *reinterpret_cast<float4*>(ad.m + off) = mv;
    *reinterpret_cast<float4*>(ad.v + off) = vv;
}

__device__ __forceinline__ void tn_reduce_body(const float* slabs, int nsplit, int n_w, int n_b, float* dW, float* db,
                                               int accumulate, int bid, const AdamRider* ad = nullptr, float step_size = 0.f,
                                               float bc2s = 0.f) {
    const int total4 = (n_w + (db ? n_b : 0)) / 4;          // n_w, n_b multiples of 4
    const int t = bid * blockDim.x + threadIdx.x;
    const int j4 = t >> 3, g = t & 7;
    if (j4 >= total4) return;                               // whole 8-lane groups leave together
    const size_t stride = (size_t)n_w + n_b;
    const float* src = slabs + (size_t)j4 * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int k = g; k < nsplit; k += 8) {
        const float4 v = *reinterpret_cast<const float4*>(src + (size_t)k * stride);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        s.x += __shfl_xor(s.x, o, 64); s.y += __shfl_xor(s.y, o, 64);
        s.z += __shfl_xor(s.z, o, 64); s.w += __shfl_xor(s.w, o, 64);
    }
    if (g == 0) {
        const int j = j4 * 4;
        float* dst = j < n_w ? dW + j : db + (j - n_w);
        if (accumulate) {
            const float4 old = *reinterpret_cast<const float4*>(dst);
            s.x += old.x; s.y += old.y; s.z += old.z; s.w += old.w;
        }
        *reinterpret_cast<float4*>(dst) = s;
        if (ad) adam_rider4(*ad, (size_t)(dst - ad->g), s, step_size, bc2s);      // the parameter right behind its gradient
    }
}

__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* slabs, int nsplit, int n_w, int n_b, float* dW,
                                                        float* db, int accumulate) {
    tn_reduce_body(slabs, nsplit, n_w, n_b, dW, db, accumulate, blockIdx.x);
}
__global__ __launch_bounds__(256) void tn_reduce_group_kernel(TnReduceGroup g) {
    __shared__ float sc[2];
    const bool with_adam = g.ad.p != nullptr;
    // (the rider blocks come FIRST in the grid: dispatched first, they run beside the reduce blocks instead of behind them)
    const int n_rider = with_adam ? g.ad.rest_block0[g.ad.n_rest] : 0;
    const int b = (int)blockIdx.x - n_rider;
    if (with_adam) {
        if (threadIdx.x == 0) {                             // adam_at_kernel's scalars, the same fp64 expressions
            const double bc1 = 1.0 - pow(g.ad.beta1, (double)g.ad.t);
            const double bc2 = 1.0 - pow(g.ad.beta2, (double)g.ad.t);
            sc[0] = (float)(g.ad.lr / bc1);
            sc[1] = (float)sqrt(bc2);
            if (blockIdx.x == 0 && g.ad.step_count) *g.ad.step_count = g.ad.t;
        }
        __syncthreads();
        if (b < 0) {
            // rider: a gradient range no reduce job of this launch produces (finished by earlier kernels of the step)
            const int rb = (int)blockIdx.x;
            int r = 0;
            for (int i = 1; i < g.ad.n_rest; i++) r += rb >= g.ad.rest_block0[i] ? 1 : 0;
            const int off = g.ad.rest_lo[r] + ((rb - g.ad.rest_block0[r]) * 256 + (int)threadIdx.x) * 4;
            if (off < g.ad.rest_hi[r])
                adam_rider4(g.ad, (size_t)off, *reinterpret_cast<const float4*>(g.ad.g + off), sc[0], sc[1]);
            return;
        }
    }
    int j = 0;
#pragma unroll
    for (int i = 1; i < PC_TN_RGROUP; i++) j += (i < g.n && b >= g.block0[i]) ? 1 : 0;
    tn_reduce_body(g.slabs[j], g.nsplit[j], g.n_w[j], g.n_b[j], g.dW[j], g.db[j], g.accumulate[j], b - g.block0[j],
                   with_adam ? &g.ad : nullptr, with_adam ? sc[0] : 0.f, with_adam ? sc[1] : 0.f);
}

static bool tn_full_tile(int R, int No, int Ni) {
    return R >= 8192 && No <= 256 && Ni <= 256 && (No > 128 || Ni > 128) && No % 64 == 0 && Ni % 64 == 0;
}

static void tn_plan(int R, int No, int Ni, int* nsplit, int* rows_per_split) {
    int s;
    if (tn_full_tile(R, No, Ni)) {
        s = 256;                                       // one workgroup per CU owns the whole dW
    } else {
        const int tiles = ((No + TM - 1) / TM) * ((Ni + TM - 1) / TM);
        s = (512 + tiles - 1) / tiles;                 // ~2 workgroups per CU in total
    }
    const int max_s = (R + 63) / 64;                   // at least 2 chunks of 32 rows per split (few rows: spread them)
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    int rps = (R + s - 1) / s;
    rps = (rps + TK - 1) / TK * TK;
    s = (R + rps - 1) / rps;
    *nsplit = s;
    *rows_per_split = rps;
}

// An upper bound of the plan's slab count that is MONOTONE in R: workspaces are laid out for the largest row count a
// call may see and used with the actual one, and the plan itself is not monotone (R = 8192 splits 256 ways, R = 19969
// only 209 ways because rows_per_split is rounded up to whole chunks).
size_t gemm_tn_workspace_floats(int R, int No, int Ni) {
    const int tiles = ((No + TM - 1) / TM) * ((Ni + TM - 1) / TM);
    int cap = (512 + tiles - 1) / tiles;
    if (No <= 256 && Ni <= 256 && (No > 128 || Ni > 128) && No % 64 == 0 && Ni % 64 == 0 && cap < 256) cap = 256;
    int s = (R + 63) / 64;
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    return (size_t)s * ((size_t)No * Ni + No);
}

// appends one slab sum to a deferred list; the slab regions of a list must be pairwise disjoint (they are all read
// by the one reduce launch at the end)
static int tn_defer_push(TnDefer* d, const float* slabs, size_t slab_floats, int nsplit, int n_w, int n_b, float* dW,
                         float* db, int accumulate) {
    if (d->r.n >= PC_TN_RGROUP) return PC_EINVAL;
    const size_t used = (size_t)nsplit * ((size_t)n_w + n_b);
    if (used > slab_floats) return PC_EWORKSPACE;
    for (int k = 0; k < d->r.n; k++) {
        const float* lo = d->r.slabs[k];
        const float* hi = lo + (size_t)d->r.nsplit[k] * ((size_t)d->r.n_w[k] + d->r.n_b[k]);
        if (slabs < hi && lo < slabs + used) return PC_EINVAL;
    }
    const int k = d->r.n++;
    d->r.slabs[k] = slabs; d->r.nsplit[k] = nsplit; d->r.n_w[k] = n_w; d->r.n_b[k] = n_b; d->r.dW[k] = dW; d->r.db[k] = db;
    d->r.accumulate[k] = accumulate; d->r.block0[k] = d->rblocks;
    d->rblocks += ((n_w + (db ? n_b : 0)) / 4 * 8 + 255) / 256;
    return PC_OK;
}

int launch_gemm_tn(const TnArgs& a, hipStream_t st, TnDefer* defer) {
    if ((!a.Z && !a.z_onehot) || !a.A || !a.dW || !a.slabs || a.R <= 0 || a.No <= 0 || a.Ni <= 0) return PC_EINVAL;
    if (a.No % 4 || a.Ni % 4 || (a.Z && a.ldz % 4) || a.lda % 4 || a.lddw != a.Ni) return PC_ESHAPE;
    if (a.z_onehot && (a.zaux || tn_full_tile(a.R, a.No, a.Ni))) return PC_ESHAPE;     // few-row kernel only
    if (((uintptr_t)a.dW & 15) || (a.db && ((uintptr_t)a.db & 15)) || ((uintptr_t)a.slabs & 15)) return PC_ESHAPE;
    int nsplit, rps;
    tn_plan(a.R, a.No, a.Ni, &nsplit, &rps);
    if ((size_t)nsplit * ((size_t)a.No * a.Ni + a.No) > a.slab_floats) return PC_EWORKSPACE;
    const int tiles_o = (a.No + TM - 1) / TM, tiles_i = (a.Ni + TM - 1) / TM;
    const bool apro = a.prologue == NT_PRO_BNTANH, zpro = a.zaux != nullptr;
    const int pb = pc_prof_begin(PC_KIND_GEMM_TN, 2.0 * a.R * (double)a.No * a.Ni, st);
    if (tn_full_tile(a.R, a.No, a.Ni) && !(apro && zpro)) {
        // (TKC, NST): chunks of 32 rows in two stages, except the 256 x 256 tile -- 128 accumulator registers per lane: its DMA
        // geometry and gather indices for 32-row chunks do not fit beside them (7-14 VGPRs went to scratch), so it takes 16-row
        // chunks in three stages like the ZPRO forms (same steady-state rate: these kernels do not wait for HBM)
#define TN8(WO, WI, TO, TI, TKC, NST)                                                                         \
    do {                                                                                                      \
        if (apro) PC_LAUNCH((gemm_tn8_kernel<WO, WI, TO, TI, TKC, NST, true, false>), dim3(nsplit), dim3(64 * WO * WI), 0, st, a, nsplit, rps, 1, (size_t)0);    \
        else if (zpro && a.Ni > 128) PC_LAUNCH((gemm_tn8_kernel<2, 4, 4, 2, 16, 3, false, true>), dim3(nsplit), dim3(512), 0, st, a, nsplit, rps, 1, (size_t)0);  \
        else if (zpro) PC_LAUNCH((gemm_tn8_kernel<4, 2, 2, 2, 16, 3, false, true>), dim3(nsplit), dim3(512), 0, st, a, nsplit, rps, 1, (size_t)0);  \
        else PC_LAUNCH((gemm_tn8_kernel<WO, WI, TO, TI, TKC, NST, false, false>), dim3(nsplit), dim3(64 * WO * WI), 0, st, a, nsplit, rps, 1, (size_t)0);        \
    } while (0)
        if (a.No > 128 && a.Ni > 128) TN8(2, 4, 4, 2, 16, 3);
        else if (a.No > 128) TN8(4, 2, 2, 2, 32, 2);
        else TN8(2, 4, 2, 2, 32, 2);
#undef TN8
    } else {
        PC_LAUNCH(gemm_tn_kernel, dim3(tiles_o * tiles_i * nsplit), dim3(256), 0, st, a, tiles_i, nsplit, rps);
    }
    pc_prof_end(pb, st);
    PC_TRY(pc_launch_status());
    const int n_w = a.No * a.Ni, n_b = a.No;
    const int threads = (n_w + (a.db ? n_b : 0)) / 4 * 8;
    if (defer) return tn_defer_push(defer, a.slabs, a.slab_floats, nsplit, n_w, n_b, a.dW, a.db, a.accumulate);
    PC_LAUNCH(tn_reduce_kernel, dim3((threads + 255) / 256), dim3(256), 0, st, a.slabs, nsplit, n_w, n_b,
                       a.dW, a.db, a.accumulate);
    return pc_launch_status();
}

// dW[2 x 128, 256] = Z[:, 0:256]^T A as ONE launch of 2 x 128 row slices (see gemm_tn8_kernel, halves == 2); slabs0 / slabs1: the two
// halves' slab regions (each >= 128 x (128 x 256 + 128) floats), reduced as two jobs of the caller's deferred list
int launch_gemm_tn_halves(const TnArgs& a, float* slabs0, float* slabs1, size_t slab_floats, hipStream_t st, TnDefer* defer) {
    if (!a.Z || !a.A || !a.dW || !slabs0 || !slabs1 || !defer || a.R <= 0) return PC_EINVAL;
    if (a.No != 256 || a.Ni != 256 || a.ldz % 4 || a.lda % 4 || a.lddw != a.Ni || a.prologue != NT_PRO_NONE || a.zaux || a.z_onehot || a.gather)
        return PC_ESHAPE;
    if (slabs1 < slabs0) return PC_EINVAL;
    const int nsplit = 128;
    int rps = (a.R + nsplit - 1) / nsplit;
    rps = (rps + 31) / 32 * 32;                                  // whole 32-row chunks
    const int used = (a.R + rps - 1) / rps;                      // (<= 128: trailing slices own no rows and write zero slabs)
    const size_t per = (size_t)128 * a.Ni + 128;
    if ((size_t)nsplit * per > slab_floats) return PC_EWORKSPACE;
    (void)used;
    TnArgs h = a;
    h.No = 128; h.slabs = slabs0; h.slab_floats = slab_floats;
    const int pb = pc_prof_begin(PC_KIND_GEMM_TN, 2.0 * a.R * (double)a.No * a.Ni, st);
    PC_LAUNCH((gemm_tn8_kernel<2, 4, 2, 2, 32, 2, false, false>), dim3(2 * nsplit), dim3(512), 0, st, h, nsplit, rps, 2,
              (size_t)(slabs1 - slabs0));
    pc_prof_end(pb, st);
    PC_TRY(pc_launch_status());
    PC_TRY(tn_defer_push(defer, slabs0, slab_floats, nsplit, 128 * a.Ni, 128, a.dW, a.db, a.accumulate));
    return tn_defer_push(defer, slabs1, slab_floats, nsplit, 128 * a.Ni, 128, a.dW + (size_t)128 * a.Ni, a.db ? a.db + 128 : nullptr,
                         a.accumulate);
}

// One side queue per (device, main queue): two host threads stepping two models on two streams of one device each get their
// own.  A small fixed table; when it is full (or creation fails) the caller gets null and stays on its main queue.
// This table is the library's ONLY device state (include/pcompanion_hip.h "Library-owned device state"): created lazily by
// the unsplit fused Product2Vec step while PC_OPT_SIDE_QUEUE is on, destroyed by pc_release_device_state().
namespace {
struct ForkSlot { int dev; hipStream_t main_st; int state; PcFork f; };      // state: 0 free, 1 ready, -1 unavailable
ForkSlot g_fork_slots[32];
std::mutex g_fork_mu;
std::atomic<int> g_opt_side_queue{1};
std::atomic<int> g_opt_sorted_tables{1};
std::atomic<int> g_opt_bn_finalize_side{0};
std::atomic<int> g_opt_fused_loss{1};
std::atomic<int> g_opt_fused_out_chain{1};
std::atomic<int>* opt_slot(int option) {
    switch (option) {
        case PC_OPT_SIDE_QUEUE: return &g_opt_side_queue;
        case PC_OPT_SORTED_TABLE_GRADIENTS: return &g_opt_sorted_tables;
        case PC_OPT_BN_FINALIZE_SIDE: return &g_opt_bn_finalize_side;
        case PC_OPT_FUSED_LOSS: return &g_opt_fused_loss;
        case PC_OPT_FUSED_OUT_CHAIN: return &g_opt_fused_out_chain;
        default: return nullptr;
    }
}
}
int pc_opt_sorted_tables() { return g_opt_sorted_tables.load(std::memory_order_relaxed); }
int pc_opt_bn_finalize_side() { return g_opt_bn_finalize_side.load(std::memory_order_relaxed); }
int pc_opt_fused_loss() { return g_opt_fused_loss.load(std::memory_order_relaxed); }
int pc_opt_fused_out_chain() { return g_opt_fused_out_chain.load(std::memory_order_relaxed); }

extern "C" int pc_set_option(int option, int value) {
    std::atomic<int>* o = opt_slot(option);
    if (!o || (value != 0 && value != 1)) return PC_EINVAL;
    o->store(value, std::memory_order_relaxed);
    return PC_OK;
}
extern "C" int pc_get_option(int option, int* value) {
    std::atomic<int>* o = opt_slot(option);
    if (!o || !value) return PC_EINVAL;
    *value = o->load(std::memory_order_relaxed);
    return PC_OK;
}
extern "C" int pc_release_device_state(void) {
    std::lock_guard<std::mutex> lock(g_fork_mu);
    int cur = 0, rc = PC_OK;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (ForkSlot& c : g_fork_slots) {
        if (c.state == 1) {
            if (hipSetDevice(c.dev) != hipSuccess) { rc = (int)hipErrorInvalidDevice; continue; }
            (void)hipStreamSynchronize(c.f.side);                 // (nothing is pending between two steps: returns at once)
            for (int i = 0; i < PC_FORK_EVENTS; i++) { (void)hipEventDestroy(c.f.fork[i]); (void)hipEventDestroy(c.f.join[i]); }
            (void)hipStreamDestroy(c.f.side);
        }
        c = ForkSlot{};
    }
    if (have_cur) (void)hipSetDevice(cur);
    return rc;
}

PcFork* pc_fork_get(hipStream_t main_st) {
    int dev = 0;
    if (!g_opt_side_queue.load(std::memory_order_relaxed) || hipGetDevice(&dev) != hipSuccess) return nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;     // a stream being captured into a graph keeps the step on itself
    if (hipStreamIsCapturing(main_st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
    std::lock_guard<std::mutex> lock(g_fork_mu);
    ForkSlot* s = nullptr;
    for (ForkSlot& c : g_fork_slots) {
        if (c.state != 0 && c.dev == dev && c.main_st == main_st) { s = &c; break; }
        if (c.state == 0 && !s) s = &c;
    }
    if (!s) return nullptr;
    if (s->state == 0) {
        s->dev = dev; s->main_st = main_st;
        PcFork& f = s->f;
        // (the side queue runs at the calling stream's priority: its launches are part of the caller's step)
        int prio = 0;
        bool ok = (hipStreamGetPriority(main_st, &prio) == hipSuccess
                       ? hipStreamCreateWithPriority(&f.side, hipStreamNonBlocking, prio)
                       : hipStreamCreateWithFlags(&f.side, hipStreamNonBlocking)) == hipSuccess;
        for (int i = 0; ok && i < PC_FORK_EVENTS; i++)
            ok = hipEventCreateWithFlags(&f.fork[i], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&f.join[i], hipEventDisableTiming) == hipSuccess;
        f.pending = 0;
        s->state = ok ? 1 : -1;
    }
    return s->state == 1 ? &s->f : nullptr;
}
int pc_fork_begin(PcFork* f, int i, hipStream_t main_st) {
    PC_HIP_TRY(hipEventRecord(f->fork[i], main_st));
    PC_HIP_TRY(hipStreamWaitEvent(f->side, f->fork[i], 0));
    f->pending = 1;
    return PC_OK;
}
int pc_fork_mark(PcFork* f, int i) {
    PC_HIP_TRY(hipEventRecord(f->join[i], f->side));
    return PC_OK;
}
int pc_fork_wait(PcFork* f, int i, hipStream_t main_st) {
    PC_HIP_TRY(hipStreamWaitEvent(main_st, f->join[i], 0));
    return PC_OK;
}
int pc_fork_join(PcFork* f, int i, hipStream_t main_st) {
    PC_TRY(pc_fork_mark(f, i));
    PC_TRY(pc_fork_wait(f, i, main_st));
    f->pending = 0;
    return PC_OK;
}

int launch_tn_reduce_deferred(TnDefer* d, hipStream_t st) {
    if (!d) return PC_EINVAL;
    if (d->fork && d->fork->pending) PC_TRY(pc_fork_join(d->fork, 1, st));      // side-queue products: their slabs are summed here
    if (d->r.n == 0 && !d->adam) return PC_OK;
    for (int k = d->r.n; k <= PC_TN_RGROUP; k++) d->r.block0[k] = d->rblocks;
    int blocks = d->rblocks;
    if (d->adam) {
        // the optimizer rides: every reduce output must lie in the flat gradient buffer; the ranges between them get rider blocks
        const pc_adam_fused* a = d->adam;
        if (!a->param || !a->grad || !a->exp_avg || !a->exp_avg_sq || a->n == 0 || a->n % 4 || a->t < 1 || a->n > 0x7fffffffu) return PC_EINVAL;
        if (((uintptr_t)a->param | (uintptr_t)a->grad | (uintptr_t)a->exp_avg | (uintptr_t)a->exp_avg_sq) & 15) return PC_ESHAPE;
        AdamRider& ad = d->r.ad;
        ad.p = a->param; ad.m = a->exp_avg; ad.v = a->exp_avg_sq; ad.g = a->grad; ad.n = a->n;
        ad.step_count = a->step_count; ad.t = (long long)a->t; ad.lr = a->lr; ad.beta1 = a->beta1; ad.beta2 = a->beta2;
        ad.omb1 = (float)(1.0 - a->beta1); ad.beta2f = (float)a->beta2; ad.omb2 = (float)(1.0 - a->beta2); ad.eps = (float)a->eps;
        long lo[2 * PC_TN_RGROUP], hi[2 * PC_TN_RGROUP];
        int ni = 0;
        for (int k = 0; k < d->r.n; k++) {
            const float* outs[2] = {d->r.dW[k], d->r.db[k]};
            const int lens[2] = {d->r.n_w[k], d->r.db[k] ? d->r.n_b[k] : 0};
            for (int u = 0; u < 2; u++) {
                if (!outs[u] || lens[u] == 0) continue;
                const long o = outs[u] - a->grad;
                if (o < 0 || o + lens[u] > (long)a->n || (o & 3) || (lens[u] & 3)) return PC_EINVAL;
                lo[ni] = o; hi[ni] = o + lens[u]; ni++;
            }
        }
        for (int i = 1; i < ni; i++)                          // insertion sort by start (<= 32 intervals)
            for (int j2 = i; j2 > 0 && lo[j2] < lo[j2 - 1]; j2--) { long t0 = lo[j2]; lo[j2] = lo[j2 - 1]; lo[j2 - 1] = t0; t0 = hi[j2]; hi[j2] = hi[j2 - 1]; hi[j2 - 1] = t0; }
        long cur = 0;
        int nr = 0, rb = 0;
        for (int i = 0; i <= ni; i++) {
            const long stop = i < ni ? lo[i] : (long)a->n;
            if (stop > cur) {
                if (nr >= PC_ADAM_REST) return PC_EINVAL;
                ad.rest_lo[nr] = (int)cur; ad.rest_hi[nr] = (int)stop; ad.rest_block0[nr] = rb;
                rb += (int)((stop - cur + 1023) / 1024);
                nr++;
            }
            if (i < ni) {
                if (lo[i] < cur) return PC_EINVAL;            // two reduce jobs writing the same gradient range
                cur = hi[i];
            }
        }
        ad.n_rest = nr; ad.rest_block0[nr] = rb;
        ad.block0 = d->rblocks;
        blocks += rb;
    }
    if (blocks == 0) { tn_defer_init(d); return PC_OK; }
    PC_LAUNCH(tn_reduce_group_kernel, dim3(blocks), dim3(256), 0, st, d->r);
    tn_defer_init(d);
    return pc_launch_status();
}

// n <= PC_TN_GROUP independent products whose slab regions do not overlap.  Products that qualify for the full-tile
// kernels (many rows) are launched on their own.
int launch_gemm_tn_group(const TnArgs* args, int n, const TnReduceJob* extra, int n_extra, hipStream_t st, TnDefer* defer) {
    if (!args || n < 1 || n > PC_TN_GROUP || n_extra < 0 || n_extra > PC_TN_EXTRA || (n_extra && !extra)) return PC_EINVAL;
    TnGroup g = {};
    TnReduceGroup r = {};
    double flops = 0.0;
    int blocks = 0, rblocks = 0;
    // One round of the chip (2 workgroups of this kernel per CU): when the members' own plans add up to more than
    // 512 workgroups their splits are thinned in proportion (each then takes more 32-row chunks; fewer slabs).
    int planned = 0;
    for (int i = 0; i < n; i++) {
        const TnArgs& a = args[i];
        if (tn_full_tile(a.R, a.No, a.Ni) || a.prologue != NT_PRO_NONE || a.R <= 0 || a.No <= 0 || a.Ni <= 0) continue;
        int ns, rp;
        tn_plan(a.R, a.No, a.Ni, &ns, &rp);
        planned += ((a.No + TM - 1) / TM) * ((a.Ni + TM - 1) / TM) * ns;
    }
    const double thin = planned > 512 ? 512.0 / planned : 1.0;
    for (int i = 0; i < n; i++) {
        const TnArgs& a = args[i];
        if (tn_full_tile(a.R, a.No, a.Ni) || a.prologue != NT_PRO_NONE) { PC_TRY(launch_gemm_tn(a, st, defer)); continue; }
        if ((!a.Z && !a.z_onehot) || !a.A || !a.dW || !a.slabs || a.R <= 0 || a.No <= 0 || a.Ni <= 0) return PC_EINVAL;
        if (a.No % 4 || a.Ni % 4 || (a.Z && a.ldz % 4) || a.lda % 4 || a.lddw != a.Ni) return PC_ESHAPE;
        if (a.z_onehot && a.zaux) return PC_EINVAL;
        if (((uintptr_t)a.dW & 15) || (a.db && ((uintptr_t)a.db & 15)) || ((uintptr_t)a.slabs & 15)) return PC_ESHAPE;
        int nsplit, rps;
        tn_plan(a.R, a.No, a.Ni, &nsplit, &rps);
        if (thin < 1.0) {
            int s2 = (int)(nsplit * thin);
            if (s2 < 1) s2 = 1;
            rps = ((a.R + s2 - 1) / s2 + TK - 1) / TK * TK;
            nsplit = (a.R + rps - 1) / rps;
        }
        if ((size_t)nsplit * ((size_t)a.No * a.Ni + a.No) > a.slab_floats) return PC_EWORKSPACE;
        for (int k = 0; k < g.n; k++) {                          // slab regions must be disjoint
            const float* lo = g.a[k].slabs; const float* hi = lo + g.a[k].slab_floats;
            if (a.slabs < hi && lo < a.slabs + a.slab_floats) return PC_EINVAL;
        }
        const int tiles_o = (a.No + TM - 1) / TM, tiles_i = (a.Ni + TM - 1) / TM;
        const int k = g.n++;
        g.a[k] = a; g.tiles_i[k] = tiles_i; g.nsplit[k] = nsplit; g.rps[k] = rps; g.block0[k] = blocks;
        blocks += tiles_o * tiles_i * nsplit;
        const int n_w = a.No * a.Ni, n_b = a.No;
        r.slabs[k] = a.slabs; r.nsplit[k] = nsplit; r.n_w[k] = n_w; r.n_b[k] = n_b; r.dW[k] = a.dW; r.db[k] = a.db;
        r.accumulate[k] = a.accumulate; r.block0[k] = rblocks;
        rblocks += ((n_w + (a.db ? n_b : 0)) / 4 * 8 + 255) / 256;
        flops += 2.0 * a.R * (double)a.No * a.Ni;
    }
    r.n = g.n;
    for (int i = 0; i < n_extra; i++) {                          // slab sets filled by other kernels
        const TnReduceJob& e = extra[i];
        if (!e.slabs || !e.out || e.nsplit < 1 || e.n < 4 || e.n % 4 || ((uintptr_t)e.slabs & 15) || ((uintptr_t)e.out & 15))
            return PC_EINVAL;
        const int k = r.n++;
        r.slabs[k] = e.slabs; r.nsplit[k] = e.nsplit; r.n_w[k] = e.n; r.n_b[k] = 0; r.dW[k] = e.out; r.db[k] = nullptr;
        r.accumulate[k] = e.accumulate; r.block0[k] = rblocks;
        rblocks += (e.n / 4 * 8 + 255) / 256;
    }
    if (r.n == 0) return PC_OK;
    for (int k = g.n; k <= PC_TN_GROUP; k++) g.block0[k] = blocks;
    for (int k = r.n; k <= PC_TN_RGROUP; k++) r.block0[k] = rblocks;
    if (g.n) {
        const int pb = pc_prof_begin(PC_KIND_GEMM_TN, flops, st);
        PC_LAUNCH(gemm_tn_group_kernel, dim3(blocks), dim3(256), 0, st, g);
        pc_prof_end(pb, st);
        PC_TRY(pc_launch_status());
    }
    if (defer) {
        for (int k = 0; k < r.n; k++)
            PC_TRY(tn_defer_push(defer, r.slabs[k], (size_t)r.nsplit[k] * ((size_t)r.n_w[k] + r.n_b[k]), r.nsplit[k], r.n_w[k],
                                 r.n_b[k], r.dW[k], r.db[k], r.accumulate[k]));
        return PC_OK;
    }
    PC_LAUNCH(tn_reduce_group_kernel, dim3(rblocks), dim3(256), 0, st, r);
    return pc_launch_status();
}

