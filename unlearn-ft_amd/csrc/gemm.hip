// Implicit GEMM on the CDNA4 matrix cores: Linear / Conv3x3 (NHWC) forward, dgrad and wgrad in one template.
//   C[M,N] (+)= alpha * sum_k A(m,k) B(n,k) + bias[n] + rowvec[m/rpb][n] + R[m][n]
// Workgroup = 256 threads = 4 waves (2x2), tile 128x128, K-step 64 (bf16) / 32 (fp32); each wave owns a 64x64
// sub-tile as 4x4 MFMA 16x16 accumulators (32 bf16 MFMAs per barrier).  Operands are staged global -> registers -> LDS
// (double buffered, one barrier per K-step); the next tile's 8 x 16-byte loads per thread are issued BEFORE the current
// tile's MFMAs and written to LDS after them, so HBM/L2 latency hides under the matrix work.
// Every thread owns ONE tile row per operand and loads 64 contiguous bytes of it with four buffer_load_dwordx4 that share
// one 32-bit offset (+ immediates): conv padding, M/N/K tails and gather holes are expressed as an out-of-range offset,
// which the buffer unit zero-fills in hardware - no predicate registers, no selects, ~10 VALU per K-step for addressing.
// Gather index math is incremental (no integer division in the K loop): each thread keeps its (tap, ci) / pixel
// coordinates and advances them by one K-step.  Reduction-major operands (wgrad) stay in memory order in LDS and are
// transposed by ds_read_b64_tr_b16 on the way into the MFMA fragments.  The MFMA is issued with operands swapped
// (D = B.A^T) so a lane ends up with 4 consecutive n of one output row: 8/16-byte epilogue loads and stores.
// Blocks are remapped so that the n-tiles of one m-tile run back to back on one XCD (shared activation panel in L2).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "common.h"
#include <algorithm>

namespace {

#define PDMK_GEMM_NBUF 2
#ifndef PDMK_GEMM_PF2
#define PDMK_GEMM_PF2 0      // 1: two K-tiles in flight in registers (measured SLOWER on MI355X: +60-120 VGPRs), 0: one
#endif
constexpr int BM = 128, BN = 128;
// Three geometries are instantiated and chosen per launch (measured on MI355X, tools/gemm_bench.py):
//   KCH = 16-byte chunks per tile row:  8 -> K-step 64 bf16, 72 KiB LDS (2 blocks/CU);  4 -> K-step 32, 48 KiB (3/CU)
//   NW  = waves per workgroup:          8 -> 32x64 per wave (more waves per SIMD hide LDS/barrier latency: default)
//                                       4 -> 64x64 per wave (fewer LDS fragment reads per MFMA: wins on deep-K GEMMs
//                                            with more than 512 tiles, e.g. the 320->320 conv at 64x64)
template <typename T, int KCHUNKS, int NW> struct TileCfg {
    static constexpr int NTHREADS = 64 * NW;
    static constexpr int IM = 8 / NW * 2;               // 16-row MFMA tiles per wave (4 or 2)
    static constexpr int TPR = NTHREADS / 128;          // threads per tile row (2 or 4)
    static constexpr int NLD = KCHUNKS / TPR;           // 16-byte loads per thread and operand
    static constexpr int ROW_BYTES = KCHUNKS == 8 ? 144 : 96;               // padded LDS row of a rowk tile
    static constexpr int OPERAND_BYTES = 128 * ROW_BYTES;                   // >= (KCHUNKS*CH) k-rows x 144 elements
    static constexpr int CH = Mma<T>::CH;               // elements per 16-byte chunk
    static constexpr int BK = KCHUNKS * CH;             // 64 bf16 / 32 fp32 at 8 chunks per tile row
    static constexpr int RS_ROWK = ROW_BYTES / (int)sizeof(T);
    static constexpr int RS_COLK = 128 + 16;
    static constexpr int G4 = NLD * CH;                 // elements one thread loads per operand (64 / 32 bytes)
    static constexpr int GPR = 128 / G4;                // thread groups per 128-wide row (colk layouts)
    static_assert(BM * RS_ROWK * sizeof(T) <= OPERAND_BYTES, "rowk tile bytes");
    static_assert(BK * RS_COLK * sizeof(T) <= OPERAND_BYTES, "colk tile bytes");
    static_assert(NTHREADS / GPR == BK && TPR * G4 == BK && NLD >= 1, "one tile row per thread");
};

struct ConvGeom {
    int hi, wi, ci, ho, wo, ld, lg_wo, lg_howo;
};

// source pixel index for output pixel (b,oy,ox) and tap, or -1 when the tap reads zero padding
template <int CMODE>
__device__ __forceinline__ int conv_src_pixel(const ConvGeom& g, int b, int oy, int ox, int tap) {
    const int ky = (tap * 11) >> 5;          // tap / 3 for tap in [0, 9)
    const int kx = tap - 3 * ky;
    int vy, vx;
    if (CMODE == 1)      { vy = 2 * oy + ky - 1; vx = 2 * ox + kx - 1; }
    else if (CMODE == 4) { vy = 2 * oy + ky;     vx = 2 * ox + kx; }       // padding on the bottom / right only
    else                 { vy = oy + ky - 1;     vx = ox + kx - 1; }
    constexpr bool UP = CMODE == 2 || CMODE == 3;
    const int hv = UP ? 2 * g.hi : g.hi, wv = UP ? 2 * g.wi : g.wi;
    bool ok = (unsigned)vy < (unsigned)hv && (unsigned)vx < (unsigned)wv;
    if (CMODE == 3) ok = ok && (((vy | vx) & 1) == 0);
    const int iy = UP ? (vy >> 1) : vy, ix = UP ? (vx >> 1) : vx;
    return ok ? (b * g.hi + iy) * g.wi + ix : -1;
}

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
constexpr unsigned OOB = 0x80000000u;    // byte offset beyond any operand (operands are < 2 GiB): buffer loads return 0

template <typename T, int AMODE, int BMODE, int CMODE, int KCH, int NW>
__global__ __launch_bounds__(64 * NW) void igemm_kernel(pdmk_gemm_args g, int lg_wo, int lg_howo, unsigned a_bytes,
                                                         unsigned b_bytes) {
    typedef TileCfg<T, KCH, NW> TC;
    typedef Mma<T> MM;
    constexpr int CH = TC::CH, BK = TC::BK, NLD = TC::NLD, OPERAND_BYTES = TC::OPERAND_BYTES;
    constexpr int NTHREADS = TC::NTHREADS, IM = TC::IM;
    __shared__ __attribute__((aligned(16))) unsigned char smem[PDMK_GEMM_NBUF][2][OPERAND_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (g.N + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;

    const int nk_total = (g.K + BK - 1) / BK;
    const int per = (nk_total + gridDim.y - 1) / gridDim.y;
    const int kt0 = blockIdx.y * per;
    const int kt1 = min(nk_total, kt0 + per);
    if (kt0 >= kt1 && g.accumulate != 2) return;        // slab split-K: an empty split still writes its (zero) slab

    constexpr int ESZ = sizeof(T), G4 = TC::G4;
    const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), (short)0, (int)a_bytes, 0x00020000);
    const auto rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.B), (short)0, (int)b_bytes, 0x00020000);
    const ConvGeom cg{g.conv_hi, g.conv_wi, g.conv_ci, g.conv_ho, g.conv_wo, g.conv_ld, lg_wo, lg_howo};

    // ---------------------------------------------------------------- per-thread loader state (one row per operand)
    const int lr = tid / TC::TPR, lq = (tid % TC::TPR) * G4;   // rowk-type: tile row, element offset inside the k-tile
    const int ckr = tid / TC::GPR, cq = (tid % TC::GPR) * G4;  // colk-type: k-row inside the tile, column offset

    unsigned a_base = OOB;                 // rowk: byte offset of the row start
    int a_b = -1, a_oy = 0, a_ox = 0;      // conv: output pixel of this row
    int a_tap = 0, a_ci = 0;               // conv: (tap, ci) of this thread's 64 bytes in the CURRENT k-tile
    if (AMODE == PDMK_A_ROWK) {
        const int m = m0 + lr;
        if (m < g.M) a_base = (unsigned)m * (unsigned)g.lda * ESZ;
    } else if (AMODE == PDMK_A_CONV) {
        const int hw = cg.ho * cg.wo;
        const int m = m0 + lr;
        if (m < g.M) {
            a_b = m / hw;
            const int rem = m - a_b * hw;
            a_oy = rem / cg.wo;
            a_ox = rem - a_oy * cg.wo;
        }
        const int k = kt0 * BK + lq;
        a_tap = k / cg.ci;
        a_ci = k - a_tap * cg.ci;
    }
    unsigned b_base = OOB;
    int b_tap = 0, b_ci = 0;
    if (BMODE == PDMK_B_ROWK) {
        const int n = n0 + lr;
        if (n < g.N) b_base = (unsigned)n * (unsigned)g.ldb * ESZ;
    } else if (BMODE == PDMK_B_COLK_CONV) {
        const int col = n0 + cq;
        b_tap = col / cg.ci;
        b_ci = col - b_tap * cg.ci;
    }

    u32x4 ra[NLD], rb[NLD];
#if PDMK_GEMM_PF2
    u32x4 ra2[NLD], rb2[NLD];     // second register set: two tiles in flight
#endif
    // fused bias gradient (wgrad only): the blocks of the first n-tile also sum their A = dY tiles over the reduction
    // dim; every tile passes through store_tiles exactly once, which is where the registers are summed.
    const bool do_colsum = (AMODE == PDMK_A_COLK) && g.colsum_out != nullptr && n0 == 0;
    float csum[G4];
#pragma unroll
    for (int e = 0; e < G4; ++e) csum[e] = 0.f;

    auto load4 = [&](u32x4* r, const decltype(rsrcA)& rs, unsigned voff) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) r[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(voff + 16u * j), 0, 0);
    };

    auto load_tiles = [&](int kt, u32x4* ra, u32x4* rb) {
        const int k0 = kt * BK;
        // ---- A
        if (AMODE == PDMK_A_ROWK) {
            const int k = k0 + lq;
            load4(ra, rsrcA, (a_base != OOB && k < g.K) ? a_base + (unsigned)k * ESZ : OOB);
        } else if (AMODE == PDMK_A_CONV) {
            const int px = (k0 + lq < g.K && a_b >= 0) ? conv_src_pixel<CMODE>(cg, a_b, a_oy, a_ox, a_tap) : -1;
            load4(ra, rsrcA, px >= 0 ? ((unsigned)px * (unsigned)cg.ld + (unsigned)a_ci) * ESZ : OOB);
            a_ci += BK;                                   // advance this thread's (tap, ci) by one k-tile
            while (a_ci >= cg.ci) { a_ci -= cg.ci; ++a_tap; }
        } else {
            const int kr = k0 + ckr, col = m0 + cq;
            load4(ra, rsrcA, (kr < g.K && col < g.M) ? ((unsigned)kr * (unsigned)g.lda + (unsigned)col) * ESZ : OOB);
        }
        // ---- B
        if (BMODE == PDMK_B_ROWK) {
            const int k = k0 + lq;
            load4(rb, rsrcB, (b_base != OOB && k < g.K) ? b_base + (unsigned)k * ESZ : OOB);
        } else if (BMODE == PDMK_B_COLK) {
            const int kr = k0 + ckr, col = n0 + cq;
            load4(rb, rsrcB, (kr < g.K && col < g.N) ? ((unsigned)kr * (unsigned)g.ldb + (unsigned)col) * ESZ : OOB);
        } else {
            const int kr = k0 + ckr, col = n0 + cq;
            int px = -1;
            if (kr < g.K && col < g.N) {
                int b, oy, ox;
                if (cg.lg_wo >= 0) {          // power-of-two image: shifts instead of divisions
                    b = kr >> cg.lg_howo;
                    const int rem = kr & ((1 << cg.lg_howo) - 1);
                    oy = rem >> cg.lg_wo;
                    ox = rem & ((1 << cg.lg_wo) - 1);
                } else {
                    const int hw = cg.ho * cg.wo;
                    b = kr / hw;
                    const int rem = kr - b * hw;
                    oy = rem / cg.wo;
                    ox = rem - oy * cg.wo;
                }
                px = conv_src_pixel<CMODE>(cg, b, oy, ox, b_tap);
            }
            load4(rb, rsrcB, px >= 0 ? ((unsigned)px * (unsigned)cg.ld + (unsigned)b_ci) * ESZ : OOB);
        }
    };

    auto store_tiles = [&](int buf, const u32x4* ra, const u32x4* rb) {
        T* As = reinterpret_cast<T*>(smem[buf][0]);
        T* Bs = reinterpret_cast<T*>(smem[buf][1]);
        T* ad = (AMODE == PDMK_A_COLK) ? As + ckr * TC::RS_COLK + cq : As + lr * TC::RS_ROWK + lq;
        T* bd = (BMODE != PDMK_B_ROWK) ? Bs + ckr * TC::RS_COLK + cq : Bs + lr * TC::RS_ROWK + lq;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            *reinterpret_cast<u32x4*>(ad + j * CH) = ra[j];
            *reinterpret_cast<u32x4*>(bd + j * CH) = rb[j];
        }
        if (do_colsum) {
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                const T* v = reinterpret_cast<const T*>(&ra[j]);
#pragma unroll
                for (int e = 0; e < CH; ++e) csum[j * CH + e] += to_f32(v[e]);
            }
        }
    };

    f32x4 acc[IM][4];
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int cur) {
        const T* As = reinterpret_cast<const T*>(smem[cur][0]);
        const T* Bs = reinterpret_cast<const T*>(smem[cur][1]);
#pragma unroll
        for (int kk = 0; kk < BK; kk += MM::KS) {
            typename MM::frag af[IM], bf[4];
#pragma unroll
            for (int i = 0; i < IM; ++i)
                af[i] = (AMODE == PDMK_A_COLK) ? MM::load_colk(As, TC::RS_COLK, kk, wm * (16 * IM) + i * 16, lane)
                                               : MM::load_rowk(As, TC::RS_ROWK, wm * (16 * IM) + i * 16, kk, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                bf[j] = (BMODE != PDMK_B_ROWK) ? MM::load_colk(Bs, TC::RS_COLK, kk, wn * 64 + j * 16, lane)
                                               : MM::load_rowk(Bs, TC::RS_ROWK, wn * 64 + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < IM; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = MM::mma(bf[j], af[i], acc[i][j]);
        }
    };

#if PDMK_GEMM_PF2
    // two tiles in flight: while tile kt is multiplied out of LDS, tile kt+1 sits in one register set (written to LDS
    // after the MFMAs) and tile kt+2 is being fetched into the other one
    load_tiles(kt0, ra, rb);
    if (kt0 + 1 < kt1) load_tiles(kt0 + 1, ra2, rb2);
    store_tiles(0, ra, rb);
    __syncthreads();
    int cur = 0;
    for (int kt = kt0;;) {
        if (kt + 2 < kt1) load_tiles(kt + 2, ra, rb);
        compute(cur);
        if (kt + 1 < kt1) store_tiles(cur ^ 1, ra2, rb2);
        __syncthreads();
        cur ^= 1;
        if (++kt >= kt1) break;
        if (kt + 2 < kt1) load_tiles(kt + 2, ra2, rb2);
        compute(cur);
        if (kt + 1 < kt1) store_tiles(cur ^ 1, ra, rb);
        __syncthreads();
        cur ^= 1;
        if (++kt >= kt1) break;
    }
#else
    load_tiles(kt0, ra, rb);
    store_tiles(0, ra, rb);
    __syncthreads();
    int cur = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        const bool more = kt + 1 < kt1;
        if (more) load_tiles(kt + 1, ra, rb);
        compute(cur);
        if (more) store_tiles(cur ^ 1, ra, rb);
        __syncthreads();
        cur ^= 1;
    }
#endif

    if (do_colsum) {      // combine the BK k-rows through LDS (all tile reads are behind the last barrier)
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int e = 0; e < G4; ++e) red[ckr * 128 + cq + e] = csum[e];
        __syncthreads();
        if (tid < 128 && m0 + tid < g.M) {
            float s = 0.f;
            for (int j = 0; j < BK; ++j) s += red[j * 128 + tid];
            unsafeAtomicAdd(g.colsum_out + m0 + tid, s);
        }
    }

    // ---------------------------------------------------------------- epilogue (staged through LDS)
    // The accumulators go through an fp32 LDS image [64][132] in two passes (upper / lower 64 rows of the tile) so that
    // global traffic is row-contiguous: 16-byte stores / residual loads per lane on the plain path, and 256 contiguous
    // bytes per wave-instruction on the split-K float-atomic path (scattered 4-byte atomics run ~10x slower).
    const bool first = blockIdx.y == 0;
    const bool slab = g.accumulate == 2;                 // split-K partials to slab blockIdx.y of a [splitk][M][ldc] workspace
    const bool atomic = gridDim.y > 1 && !slab;
    const bool acc1 = g.accumulate == 1;
    const bool f32out = g.out_f32 || sizeof(T) == 4;
    const bool vec8 = ((g.N & 7) == 0) && ((g.ldc & 7) == 0) && (g.R == nullptr || (g.ldr & 7) == 0);
    constexpr int SROW = 132;
    float* stage = reinterpret_cast<float*>(smem);
    static_assert(64 * SROW * 4 <= PDMK_GEMM_NBUF * 2 * OPERAND_BYTES, "staging image must fit the tile buffers");
    float* Cf = reinterpret_cast<float*>(g.C) + (slab ? (long)blockIdx.y * g.M * g.ldc : 0L);
    T* Ct = reinterpret_cast<T*>(g.C);
    const T* Rp = reinterpret_cast<const T*>(g.R);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
        if ((wm * IM) / 4 == pass) {
            const int lrb = (wm * 16 * IM) % 64;
#pragma unroll
            for (int i = 0; i < IM; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v = acc[i][j];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= g.alpha;
                    *reinterpret_cast<f32x4*>(stage + (lrb + i * 16 + (lane & 15)) * SROW + wn * 64 + j * 16 + (lane >> 4) * 4) = v;
                }
        }
        __syncthreads();
        if (atomic) {
            for (int rr = 0; rr < 64 / NW; ++rr) {
                const int lrow = wave * (64 / NW) + rr, m = m0 + pass * 64 + lrow;
                if (m >= g.M) break;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int n = n0 + h * 64 + lane;
                    if (n < g.N) {
                        float v = stage[lrow * SROW + h * 64 + lane];
                        if (first) {
                            if (g.bias) v += g.bias[n];
                            if (g.rowvec) v += g.rowvec[(long)(m / g.rows_per_b) * (g.ldrv ? g.ldrv : g.N) + n];
                            if (Rp) v += to_f32(Rp[(long)m * g.ldr + n]);
                        }
                        unsafeAtomicAdd(Cf + (long)m * g.ldc + n, v);
                    }
                }
            }
        } else {
            constexpr int TPS = NTHREADS / 64, SEG = 128 / TPS;     // threads per staged row, columns per thread
            const int lrow = tid / TPS, seg = (tid % TPS) * SEG;
            const int m = m0 + pass * 64 + lrow;
            if (m < g.M) {
                const float* rv = g.rowvec ? g.rowvec + (long)(m / g.rows_per_b) * (g.ldrv ? g.ldrv : g.N) : nullptr;
#pragma unroll
                for (int c8 = 0; c8 < SEG / 8; ++c8) {
                    const int n = n0 + seg + c8 * 8;
                    if (n >= g.N) break;
                    float v[8];
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + lrow * SROW + seg + c8 * 8);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + lrow * SROW + seg + c8 * 8 + 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v[r] = lo[r]; v[4 + r] = hi[r]; }
                    const long off = (long)m * g.ldc + n;
                    if (vec8) {
                        if (g.bias) {
                            const float4 b0 = *reinterpret_cast<const float4*>(g.bias + n), b1 = *reinterpret_cast<const float4*>(g.bias + n + 4);
                            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                        }
                        if (rv) {
                            const float4 b0 = *reinterpret_cast<const float4*>(rv + n), b1 = *reinterpret_cast<const float4*>(rv + n + 4);
                            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                        }
                        if (Rp) {
                            if (sizeof(T) == 2) {
                                const bf16x8 rr = *reinterpret_cast<const bf16x8*>(Rp + (long)m * g.ldr + n);
#pragma unroll
                                for (int r = 0; r < 8; ++r) v[r] += (float)rr[r];
                            } else {
                                const float* rp = reinterpret_cast<const float*>(Rp) + (long)m * g.ldr + n;
                                const float4 b0 = *reinterpret_cast<const float4*>(rp), b1 = *reinterpret_cast<const float4*>(rp + 4);
                                v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                            }
                        }
                        if (f32out) {
                            float4 o0 = make_float4(v[0], v[1], v[2], v[3]), o1 = make_float4(v[4], v[5], v[6], v[7]);
                            if (acc1) {
                                const float4 c0 = *reinterpret_cast<const float4*>(Cf + off), c1 = *reinterpret_cast<const float4*>(Cf + off + 4);
                                o0.x += c0.x; o0.y += c0.y; o0.z += c0.z; o0.w += c0.w; o1.x += c1.x; o1.y += c1.y; o1.z += c1.z; o1.w += c1.w;
                            }
                            *reinterpret_cast<float4*>(Cf + off) = o0;
                            *reinterpret_cast<float4*>(Cf + off + 4) = o1;
                        } else {
                            if (acc1) {
                                const bf16x8 c = *reinterpret_cast<const bf16x8*>(Ct + off);
#pragma unroll
                                for (int r = 0; r < 8; ++r) v[r] += (float)c[r];
                            }
                            bf16x8 o;
#pragma unroll
                            for (int r = 0; r < 8; ++r) o[r] = (bf16)v[r];
                            st_stream(reinterpret_cast<bf16x8*>(Ct + off), o);
                        }
                    } else {
                        const int nv = min(8, g.N - n);
                        for (int r = 0; r < nv; ++r) {
                            float x = v[r];
                            if (g.bias) x += g.bias[n + r];
                            if (rv) x += rv[n + r];
                            if (Rp) x += to_f32(Rp[(long)m * g.ldr + n + r]);
                            if (f32out) Cf[off + r] = x + (acc1 ? Cf[off + r] : 0.f);
                            else Ct[off + r] = from_f32<T>(x + (acc1 ? to_f32(Ct[off + r]) : 0.f));
                        }
                    }
                }
            }
        }
    }
}

inline int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

template <typename T> int launch(const pdmk_gemm_args& g, hipStream_t st) {
    const int mt = (g.M + BM - 1) / BM, nt = (g.N + BN - 1) / BN;
    dim3 grid(mt * nt, g.splitk > 1 ? g.splitk : 1);
    int lg_wo = -1, lg_howo = -1;
    if (g.b_mode == PDMK_B_COLK_CONV) {
        lg_wo = ilog2_exact(g.conv_wo);
        lg_howo = ilog2_exact(g.conv_ho * g.conv_wo);
        if (lg_wo < 0 || lg_howo < 0) lg_wo = lg_howo = -1;
    }
    const long esz = sizeof(T);
    const long conv_bytes = (((long)g.conv_b * g.conv_hi * g.conv_wi - 1) * g.conv_ld + g.conv_ci) * esz;
    const long a_bytes = g.a_mode == PDMK_A_ROWK ? ((long)(g.M - 1) * g.lda + g.K) * esz
                       : g.a_mode == PDMK_A_CONV ? conv_bytes : ((long)(g.K - 1) * g.lda + g.M) * esz;
    const long b_bytes = g.b_mode == PDMK_B_ROWK ? ((long)(g.N - 1) * g.ldb + g.K) * esz
                       : g.b_mode == PDMK_B_COLK ? ((long)(g.K - 1) * g.ldb + g.N) * esz : conv_bytes;
    if (a_bytes >= (1L << 31) || b_bytes >= (1L << 31)) return -1;     // 32-bit buffer offsets
    // more blocks than the 2-per-CU geometry can hold at once -> the 3-per-CU geometry runs them in fewer rounds
    const bool many = (long)grid.x * grid.y > 512;            // more blocks than the 2-per-CU geometry holds at once
    // (the 4-wave 64x64-per-wave geometry wins isolated deep-K micro-benchmarks by up to 17 % but loses inside the real
    // step, where operands are not L2-warm: not dispatched)
    const bool deep4 = false;
#define PDMK_GO(AM, BMD, CM)                                                                                        \
    do {                                                                                                            \
        if (deep4)                                                                                                  \
            hipLaunchKernelGGL((igemm_kernel<T, AM, BMD, CM, 4, 4>), grid, dim3(256), 0, st, g, lg_wo, lg_howo,     \
                               (unsigned)a_bytes, (unsigned)b_bytes);                                               \
        else if (many)                                                                                              \
            hipLaunchKernelGGL((igemm_kernel<T, AM, BMD, CM, 4, 8>), grid, dim3(512), 0, st, g, lg_wo, lg_howo,     \
                               (unsigned)a_bytes, (unsigned)b_bytes);                                               \
        else                                                                                                        \
            hipLaunchKernelGGL((igemm_kernel<T, AM, BMD, CM, 8, 8>), grid, dim3(512), 0, st, g, lg_wo, lg_howo,     \
                               (unsigned)a_bytes, (unsigned)b_bytes);                                               \
    } while (0)
    if (g.a_mode == PDMK_A_ROWK && g.b_mode == PDMK_B_ROWK) PDMK_GO(PDMK_A_ROWK, PDMK_B_ROWK, 0);
    else if (g.a_mode == PDMK_A_CONV && g.b_mode == PDMK_B_ROWK) {
        switch (g.conv_mode) {
            case 0: PDMK_GO(PDMK_A_CONV, PDMK_B_ROWK, 0); break;
            case 1: PDMK_GO(PDMK_A_CONV, PDMK_B_ROWK, 1); break;
            case 2: PDMK_GO(PDMK_A_CONV, PDMK_B_ROWK, 2); break;
            case 3: PDMK_GO(PDMK_A_CONV, PDMK_B_ROWK, 3); break;
            default: PDMK_GO(PDMK_A_CONV, PDMK_B_ROWK, 4); break;
        }
    } else if (g.a_mode == PDMK_A_COLK && g.b_mode == PDMK_B_COLK) PDMK_GO(PDMK_A_COLK, PDMK_B_COLK, 0);
    else if (g.a_mode == PDMK_A_COLK && g.b_mode == PDMK_B_COLK_CONV) {
        switch (g.conv_mode) {
            case 0: PDMK_GO(PDMK_A_COLK, PDMK_B_COLK_CONV, 0); break;
            case 1: PDMK_GO(PDMK_A_COLK, PDMK_B_COLK_CONV, 1); break;
            case 2: PDMK_GO(PDMK_A_COLK, PDMK_B_COLK_CONV, 2); break;
            default: return -2;
        }
    } else return -2;
#undef PDMK_GO
    PDMK_CHECK_LAUNCH();
    return 0;
}

}  // namespace

int pdmk_gemm_dma_launch(const pdmk_gemm_args& g, hipStream_t st, long a_bytes, long b_bytes);   // gemm_dma.hip
int pdmk_gemm_ring_launch(const pdmk_gemm_args& g, hipStream_t st, long a_bytes, long b_bytes, int id);  // gemm_ring.hip
int pdmk_gemm_ring_num_configs();
int pdmk_gemm_ring_pick(const pdmk_gemm_args& g);
int pdmk_wgrad_ring_launch(const pdmk_gemm_args& g, hipStream_t st, long a_bytes, long b_bytes, int id);
int pdmk_wgrad_ring_num_configs();

static int env_int(const char* name, int def, int* cache) {     // cached unless PDMK_ENV_DYNAMIC is set (A/B tools)
    static int dynamic = -1;
    if (dynamic < 0) dynamic = getenv("PDMK_ENV_DYNAMIC") ? 1 : 0;
    if (*cache == INT32_MIN || dynamic) {
        const char* e = getenv(name);
        *cache = e ? atoi(e) : def;
    }
    return *cache;
}
static int dma_mode() { static int c = INT32_MIN; return env_int("PDMK_GEMM_DMA", 1, &c); }    // 0: register-staged only
static int ring_mode() { static int c = INT32_MIN; return env_int("PDMK_GEMM_RING", 1, &c); }  // 0: K-step-32 kernels only
static int tune_mode() { static int c = INT32_MIN; return env_int("PDMK_GEMM_TUNE", 1, &c); }  // 0: heuristics only
static int forced_cfg() { static int c = INT32_MIN; return env_int("PDMK_RING_CFG", -1, &c); } // >= 0: that candidate id
static int forced_wcfg() { static int c = INT32_MIN; return env_int("PDMK_WGRAD_CFG", -1, &c); } // same, weight gradients

// ---------------------------------------------------------------------------------------------------------------
// Plan cache.  The forward / dgrad GEMMs of the step come in ~200 shapes whose best tile shape, ring depth and split-K
// factor depend on how the grid quantises onto 256 CUs; instead of modelling that, every new shape is timed once on
// the device (candidates: the K-step-32 kernels and the LDS-DMA ring shapes of gemm_ring.hip; 3 launches each into a
// scratch output) the first time it is seen outside stream capture, and the winner is cached for the process.
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct PlanKey {
    int v[10];
    bool operator<(const PlanKey& o) const { return memcmp(v, o.v, sizeof v) < 0; }
};
std::mutex g_plan_mu;
std::map<PlanKey, int> g_plan_cfg;   // (shape, splitk) -> candidate id (0 = K-step-32 kernels, 1 + ring config id)
std::map<PlanKey, int> g_plan_sk;    // shape -> split-K factor
thread_local int g_last_candidate = -1;
bool g_plan_file_loaded = false;

// Optional persistence (PDMK_PLAN_CACHE=<file>): one line per entry "c|s v0 .. v9 value"; loaded on first use, appended
// to as shapes are tuned, so that a later process (a profiler run, the next training job) starts with the plans in place.
void plan_file_load() {
    if (g_plan_file_loaded) return;
    g_plan_file_loaded = true;
    const char* path = getenv("PDMK_PLAN_CACHE");
    if (!path) return;
    FILE* f = fopen(path, "r");
    if (!f) return;
    int ver = 0;
    if (fscanf(f, " pdmk-plan %d", &ver) != 1 || ver != pdmk_version()) {     // candidate numbering of another build
        fclose(f);
        return;
    }
    char kind;
    PlanKey k;
    int val;
    while (fscanf(f, " %c %d %d %d %d %d %d %d %d %d %d %d", &kind, &k.v[0], &k.v[1], &k.v[2], &k.v[3], &k.v[4], &k.v[5],
                  &k.v[6], &k.v[7], &k.v[8], &k.v[9], &val) == 12) {
        if (kind == 'c') g_plan_cfg[k] = val;
        else if (kind == 's') g_plan_sk[k] = val;
    }
    fclose(f);
}
void plan_file_append(char kind, const PlanKey& k, int val) {
    const char* path = getenv("PDMK_PLAN_CACHE");
    if (!path) return;
    FILE* probe = fopen(path, "r");
    int ver = 0;
    const bool fresh = !probe || fscanf(probe, " pdmk-plan %d", &ver) != 1 || ver != pdmk_version();
    if (probe) fclose(probe);
    FILE* f = fopen(path, fresh ? "w" : "a");
    if (!f) return;
    if (fresh) fprintf(f, "pdmk-plan %d\n", pdmk_version());
    fprintf(f, "%c %d %d %d %d %d %d %d %d %d %d %d\n", kind, k.v[0], k.v[1], k.v[2], k.v[3], k.v[4], k.v[5], k.v[6], k.v[7],
            k.v[8], k.v[9], val);
    fclose(f);
}
void* g_scratch = nullptr;
size_t g_scratch_bytes = 0;

PlanKey make_key(const pdmk_gemm_args& g, int sk) {
    PlanKey k;
    const bool cv = g.a_mode == PDMK_A_CONV || g.b_mode == PDMK_B_COLK_CONV;
    // weight gradients whose splits go to slabs (accumulate = 2: plain stores, summed later by pdmk_splitk_finish_group) are
    // planned apart from the ones that add with atomics: without the atomic traffic larger split factors win
    const int slabw = (g.a_mode == PDMK_A_COLK && g.accumulate == 2) ? 1000 : 0;
    const int v[10] = {g.M, g.N, g.K, g.a_mode, g.b_mode, (cv ? g.conv_mode : 100 * g.epilogue) + slabw, cv ? g.conv_hi : 0,
                       cv ? g.conv_wi : 0, cv ? g.conv_ci : 0, sk};
    memcpy(k.v, v, sizeof v);
    return k;
}

bool ring_eligible(const pdmk_gemm_args& g) {
    return g.dtype == PDMK_BF16 && g.b_mode == PDMK_B_ROWK && g.a_mode != PDMK_A_COLK;
}

bool wgrad_eligible(const pdmk_gemm_args& g) {
    return g.dtype == PDMK_BF16 && g.a_mode == PDMK_A_COLK && g.b_mode != PDMK_B_ROWK && g.out_f32 &&
           (g.b_mode == PDMK_B_COLK || g.conv_mode <= 2 || (g.conv_mode >= 5 && g.conv_mode <= 8));
}

bool operand_bytes(const pdmk_gemm_args& g, long* ab, long* bb) {
    // phase modes: the gradient phases (9..12) read a 2hi x 2wi image; the weight gradient of a phase (5..8) reads dY rows of
    // the 2hi x 2wi image
    const long in_px = (long)g.conv_b * g.conv_hi * g.conv_wi * ((g.a_mode == PDMK_A_CONV && g.conv_mode >= 9) ? 4 : 1);
    const long conv_bytes = ((in_px - 1) * g.conv_ld + g.conv_ci) * 2;
    const long krows = (g.a_mode == PDMK_A_COLK && g.b_mode == PDMK_B_COLK_CONV && g.conv_mode >= 5) ? 4L * g.K : (long)g.K;
    *ab = g.a_mode == PDMK_A_ROWK ? ((long)(g.M - 1) * g.lda + g.K) * 2
        : g.a_mode == PDMK_A_CONV ? conv_bytes : ((krows - 1) * g.lda + g.M) * 2;
    *bb = g.b_mode == PDMK_B_ROWK ? ((long)(g.N - 1) * g.ldb + g.K) * 2
        : g.b_mode == PDMK_B_COLK ? ((long)(g.K - 1) * g.ldb + g.N) * 2 : conv_bytes;
    return *ab < (1L << 31) && *bb < (1L << 31);
}

int launch_legacy(const pdmk_gemm_args& g, hipStream_t st) {
    if (g.colstat) return -2;                            // epilogue statistics: LDS-DMA kernels only
    if ((g.a_mode == PDMK_A_CONV || g.b_mode == PDMK_B_COLK_CONV) && g.conv_mode >= 5) return -2;   // phase convs: LDS-DMA kernels only
    const long nblocks = (long)((g.M + 127) / 128) * ((g.N + 127) / 128) * (g.splitk > 1 ? g.splitk : 1);
    long ab, bb;
    if (dma_mode() && (nblocks >= 384 || dma_mode() == 2) && ring_eligible(g) && (g.K % 32) == 0 && operand_bytes(g, &ab, &bb)) {
        const int rc = pdmk_gemm_dma_launch(g, st, ab, bb);
        if (rc <= 0) return rc;
    }
    return g.dtype == PDMK_BF16 ? launch<bf16>(g, st) : launch<float>(g, st);
}

int launch_candidate(const pdmk_gemm_args& g, hipStream_t st, int id) {
    if (id <= 0) return launch_legacy(g, st);
    long ab, bb;
    if (!operand_bytes(g, &ab, &bb)) return 1;
    if (g.a_mode == PDMK_A_COLK) return pdmk_wgrad_ring_launch(g, st, ab, bb, id - 1);
    return pdmk_gemm_ring_launch(g, st, ab, bb, id - 1);
}

// rows of C a problem may write: a forward phase of an upsampling conv (conv_mode 5..8) stores row m at a pixel of the
// 2hi x 2wi image, i.e. anywhere in 4 M rows
static int debug_scratch();
long out_rows(const pdmk_gemm_args& g) {
    // PDMK_DEBUG_SCRATCH=2 (the canary's own positive control, tests only): the round-3 bug on purpose - M rows - for problems
    // whose whole overrun (3 M N fp32) stays inside the canary band
    const bool phase_fwd = g.a_mode == PDMK_A_CONV && g.conv_mode >= 5 && g.conv_mode <= 8;
    if (phase_fwd && debug_scratch() == 2 && 3L * g.M * g.N * 4 + 4096 <= (1L << 20)) return (long)g.M;
    return phase_fwd ? 4L * g.M : (long)g.M;
}

bool can_tune(hipStream_t st) {
    if (!tune_mode() || !ring_mode()) return false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cs == hipStreamCaptureStatusNone;
}

// PDMK_DEBUG_SCRATCH=1 (tests): the scratch is re-allocated for EVERY tuning pass at exactly the size that pass computed, with a
// band of kCanaryBytes of 0xA5 right behind it; scratch_check() reads the band back after the timing launches.  A candidate
// that stores outside the rows the tuner sized the scratch for (round 3: the forward phase convs, which store 4 M rows - a silent
// out-of-bounds write in ordinary runs, a memory fault only under the profiler) turns the band dirty, the tuner returns an error
// and pdmk_debug_scratch_violations() counts it.
constexpr size_t kCanaryBytes = 1 << 20;
size_t g_scratch_req = 0;            // bytes the last pass asked for (the band starts there)
int g_scratch_violations = 0;
static int debug_scratch() { static int c = INT32_MIN; return env_int("PDMK_DEBUG_SCRATCH", 0, &c); }
static_assert(kCanaryBytes == (1 << 20), "out_rows() sizes its deliberate overrun against the band");

bool ensure_scratch(size_t bytes) {
    if (debug_scratch()) {
        if (g_scratch) (void)hipFree(g_scratch);
        g_scratch = nullptr;
        g_scratch_bytes = 0;
        bytes = (bytes + 255) & ~(size_t)255;
        if (hipMalloc(&g_scratch, bytes + kCanaryBytes) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (hipMemset(reinterpret_cast<char*>(g_scratch) + bytes, 0xA5, kCanaryBytes) != hipSuccess) { (void)hipGetLastError(); return false; }
        (void)hipDeviceSynchronize();
        g_scratch_bytes = bytes + kCanaryBytes;
        g_scratch_req = bytes;
        return true;
    }
    if (bytes <= g_scratch_bytes) return true;
    if (g_scratch) (void)hipFree(g_scratch);
    g_scratch = nullptr;
    g_scratch_bytes = 0;
    if (hipMalloc(&g_scratch, bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
    g_scratch_bytes = bytes;
    return true;
}

// true when the band behind the scratch is intact (or the debug mode is off)
bool scratch_check() {
    if (!debug_scratch() || !g_scratch) return true;
    (void)hipDeviceSynchronize();
    std::vector<unsigned char> host(kCanaryBytes);
    if (hipMemcpy(host.data(), reinterpret_cast<char*>(g_scratch) + g_scratch_req, kCanaryBytes, hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    for (size_t i = 0; i < kCanaryBytes; ++i)
        if (host[i] != 0xA5) {
            ++g_scratch_violations;
            fprintf(stderr, "[pdmk] tuner scratch overrun: byte %zu behind a %zu-byte scratch was overwritten\n", i, g_scratch_req);
            return false;
        }
    return true;
}

// time `reps` launches of candidate `id` (plus, for split-K, the workspace clear and the finish pass) in microseconds
float time_candidate(const pdmk_gemm_args& a, hipStream_t st, int id, float* ws, void* fin_out, hipEvent_t e0, hipEvent_t e1) {
    static const int reps = getenv("PDMK_TUNE_REPS") ? std::max(1, atoi(getenv("PDMK_TUNE_REPS"))) : 3;     // (A/B knob: 8 / 20 give the same step as 3)
    auto once = [&]() -> int {
        if (a.splitk > 1 && a.a_mode != PDMK_A_COLK) {    // slab split-K + the finish pass that adds the slabs
            const int rc = launch_candidate(a, st, id);
            if (rc) return rc;
            return pdmk_splitk_finish(ws, fin_out, nullptr, nullptr, nullptr, a.M, a.N, a.N, 0, 1, 0, a.splitk, 0, a.dtype,
                                      st);
        }
        return launch_candidate(a, st, id);
    };
    if (once() != 0) { (void)hipGetLastError(); return 1e30f; }
    float best = 1e30f;
    for (int round = 0; round < 2; ++round) {        // two rounds, keep the faster: one-off stalls must not pick the plan
        (void)hipEventRecord(e0, st);
        for (int r = 0; r < reps; ++r)
            if (once() != 0) { (void)hipGetLastError(); return 1e30f; }
        (void)hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess) { (void)hipGetLastError(); return 1e30f; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    float us = best * 1000.f / reps;
    // weight-gradient slabs: their sum is deferred to one grouped launch for many weights (pdmk_splitk_finish_group); it is
    // priced here by its traffic at 4 TB/s instead of being launched per candidate
    if (a.a_mode == PDMK_A_COLK && a.accumulate == 2) us += (float)(a.splitk + 1) * (float)a.M * (float)a.N * 4.f / 4.0e6f;
    return us;
}

// best candidate for (shape, sk); *t_out = its time.  Caller holds g_plan_mu and has checked can_tune().
int tune_cfg(const pdmk_gemm_args& g, hipStream_t st, int sk, float* t_out) {
    const size_t out_bytes = (size_t)out_rows(g) * g.N * 4;
    if (!ensure_scratch((size_t)(sk > 1 ? sk : 1) * out_bytes + out_bytes + (size_t)g.M * 4 + 256)) return -1;
    (void)hipDeviceSynchronize();                    // other streams (teacher branch) must not overlap the timings
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { (void)hipGetLastError(); return -1; }
    pdmk_gemm_args a = g;
    float* ws = reinterpret_cast<float*>(g_scratch);                               // [sk][M][N] slabs (or the output)
    void* out2 = reinterpret_cast<char*>(g_scratch) + (size_t)(sk > 1 ? sk : 1) * out_bytes;
    a.accumulate = 0;
    a.splitk = sk;
    a.ldc = g.N;
    a.colstat = nullptr;                             // timing launches must not add to the caller's GroupNorm accumulators
    if (g.epilogue == PDMK_EPI_GEGLU_BWD) a.ldc = 2 * g.N;    // output = gradient of the [M, 2N] pre-activation (C2 is an INPUT here)
    else a.C2 = nullptr;                             // tuning writes into scratch: the optional second output is left out
    if (a.colsum_out)
        a.colsum_out = reinterpret_cast<float*>(reinterpret_cast<char*>(out2) + out_bytes);      // bias gradient -> scratch
    if (g.a_mode == PDMK_A_COLK) {                                              // wgrad: fp32, atomics (or slabs) for sk > 1
        a.C = ws;
        if (g.accumulate == 2 && sk > 1) a.accumulate = 2;
    }
    else if (sk > 1) { a.C = ws; a.out_f32 = 1; a.accumulate = 2; a.bias = nullptr; a.rowvec = nullptr; a.R = nullptr; }
    else a.C = out2;
    int best = -1;
    float bt = 1e30f;
    const int ncand = 1 + (g.a_mode == PDMK_A_COLK ? pdmk_wgrad_ring_num_configs() : pdmk_gemm_ring_num_configs());
    for (int id = g.epilogue ? 1 : 0; id < ncand; ++id) {        // fused epilogues live in the ring kernels only
        const float t = time_candidate(a, st, id, ws, out2, e0, e1);
        if (t < bt) { bt = t; best = id; }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (getenv("PDMK_TUNE_LOG"))
        fprintf(stderr, "[pdmk tune] M=%d N=%d K=%d amode=%d cmode=%d sk=%d -> cand %d (%.1f us)\n", g.M, g.N, g.K, g.a_mode,
                g.conv_mode, sk, best, bt);
    if (!scratch_check()) return -1;
    *t_out = bt;
    return best;
}

int heuristic_cfg(const pdmk_gemm_args& g) {
    if (!ring_mode()) return 0;
    if (g.a_mode == PDMK_A_COLK) return (g.b_mode == PDMK_B_COLK_CONV || g.conv_mode >= 5) ? 2 : 0;   // what the tuned plans pick most often
    return 1 + pdmk_gemm_ring_pick(g);
}

int heuristic_wgrad_sk(const pdmk_gemm_args& g) {     // untuned default: ~512 workgroups, >= 16 K-steps per split
    const int tiles = ((g.M + 127) / 128) * ((g.N + 127) / 128);
    const int nk = g.K / 64 > 0 ? g.K / 64 : 1;
    int s = 512 / (tiles > 0 ? tiles : 1);
    if (s > nk / 16) s = nk / 16;
    if (s > 64) s = 64;
    return s < 1 ? 1 : s;
}

int heuristic_sk(const pdmk_gemm_args& g) {           // untuned default: split until ~256 workgroups, >= 8 K-steps each
    if (g.N & 3) return 1;
    const bool conv0 = g.a_mode == PDMK_A_CONV && g.conv_mode == 0;
    const int bm = conv0 ? (g.M >= 8192 ? 256 : 128) : (g.M >= 8192 ? 128 : 64);
    const int bn = conv0 ? 160 : 128;
    const long tiles = (long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn);
    const int nk = (g.K + 63) / 64;
    if (tiles >= 160 || nk < 30) return 1;
    long s = (256 + tiles - 1) / tiles;
    if (s > nk / 8) s = nk / 8;
    if (conv0 && s > (g.conv_ci + 63) / 64) s = (g.conv_ci + 63) / 64;
    return s < 1 ? 1 : (s > 16 ? 16 : (int)s);
}

}  // namespace

// argument validation shared by pdmk_gemm and pdmk_gemm_group (0 = ok)
static int validate_args(const pdmk_gemm_args& g) {
    if (!g.A || !g.B || !g.C) return -1;
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return -1;
    const int ch = g.dtype == PDMK_BF16 ? 8 : 4;     // elements per 16 bytes
    const int g4 = 4 * ch;                             // a thread moves up to 64 contiguous bytes: 32 bf16 / 16 fp32
    if (g.dtype != PDMK_BF16 && g.dtype != PDMK_F32) return -2;
    if (g.a_mode != PDMK_A_COLK && (g.K % g4)) return -1;   // reduction-major operands predicate every k-row
    if (((uintptr_t)g.A | (uintptr_t)g.B) & 15) return -1;
    if (g.a_mode == PDMK_A_ROWK && (g.lda % ch)) return -1;
    if (g.a_mode == PDMK_A_COLK && ((g.lda % ch) || (g.M % g4))) return -1;
    if (g.b_mode == PDMK_B_ROWK && (g.ldb % ch)) return -1;
    if (g.b_mode == PDMK_B_COLK && ((g.ldb % ch) || (g.N % g4))) return -1;
    if (g.a_mode == PDMK_A_CONV || g.b_mode == PDMK_B_COLK_CONV) {
        if (g.conv_ci <= 0 || (g.conv_ci % g4) || (g.conv_ld % ch) || g.conv_mode < 0 || g.conv_mode > 13) return -1;
        if (g.conv_mode >= 5) {     // 2x2 phase of a nearest-x2 upsample + 3x3 conv (pdmk.h): bf16 LDS-DMA kernels only
            if (g.dtype != PDMK_BF16 || g.conv_ho != g.conv_hi || g.conv_wo != g.conv_wi) return -2;
            if (g.b_mode == PDMK_B_COLK_CONV && g.conv_mode > 8) return -1;
        }
        if (g.conv_mode == 4 && (g.b_mode == PDMK_B_COLK_CONV || (g.conv_hi & 1) || (g.conv_wi & 1))) return -1;   // forward only
        if (g.conv_b <= 0 || g.conv_hi <= 0 || g.conv_wi <= 0 || g.conv_ho <= 0 || g.conv_wo <= 0) return -1;
        const long px = (long)g.conv_b * g.conv_ho * g.conv_wo;
        if (px >= (1L << 30) || (long)g.conv_b * g.conv_hi * g.conv_wi >= (1L << 30)) return -1;   // 32-bit pixel ids
        const int ntaps = g.conv_mode == 13 ? 16 : (g.conv_mode >= 5 ? 4 : 9);
        if (g.a_mode == PDMK_A_CONV && (g.M != px || g.K != ntaps * g.conv_ci)) return -1;
        if (g.b_mode == PDMK_B_COLK_CONV && (g.K != px || g.N != ntaps * g.conv_ci)) return -1;
        // gather geometry must be consistent with the source extent (out-of-image taps read as zero padding)
        if (g.conv_mode >= 5 && (long)g.conv_b * g.conv_hi * g.conv_wi * 4 >= (1L << 30)) return -1;
        const bool half = g.conv_mode == 1 || g.conv_mode == 4, twice = g.conv_mode == 2 || g.conv_mode == 3;
        const int eh = half ? (g.conv_hi + 1) / 2 : (twice ? 2 * g.conv_hi : g.conv_hi);
        const int ew = half ? (g.conv_wi + 1) / 2 : (twice ? 2 * g.conv_wi : g.conv_wi);
        if (g.conv_ho != eh || g.conv_wo != ew) return -1;
    }
    if (g.splitk > 1 && !(g.out_f32 || g.dtype == PDMK_F32)) return -1;
    if (g.accumulate == 2 && (g.splitk <= 1 || !(g.out_f32 || g.dtype == PDMK_F32) || g.bias || g.rowvec || g.R)) return -1;
    if (g.accumulate < 0 || g.accumulate > 2) return -1;
    if (g.rowvec && g.rows_per_b <= 0) return -1;
    if (g.epilogue != PDMK_EPI_NONE) {
        if (g.epilogue != PDMK_EPI_GEGLU && g.epilogue != PDMK_EPI_GEGLU_BWD) return -2;
        if (g.dtype != PDMK_BF16 || g.a_mode != PDMK_A_ROWK || g.b_mode != PDMK_B_ROWK || g.out_f32) return -2;
        if (g.epilogue == PDMK_EPI_GEGLU) {
            if ((g.N % 16) || (g.ldc % 8) || (g.C2 && (g.ldc2 % 8)) || g.splitk > 1 || g.accumulate || g.R || g.rowvec ||
                g.alpha != 1.0f || (g.K % 8))
                return -1;
        } else {        // GEGLU_BWD: C [M, 2N] = gradient of the pre-activation C2 [M, 2N]; plain GEMM otherwise
            if ((g.N % 8) || (g.ldc % 8) || !g.C2 || (g.ldc2 % 8) || g.splitk > 1 || g.accumulate || g.R || g.rowvec || g.bias ||
                g.alpha != 1.0f || (g.K % 8) || (((uintptr_t)g.C2 | (uintptr_t)g.C) & 15))
                return -1;
        }
        if (!ring_mode()) return -2;
    }
    if (g.colstat) {      // GroupNorm statistics of the output from the epilogue (pdmk.h): LDS-DMA kernels, plain bf16 epilogue
        if (g.dtype != PDMK_BF16 || g.a_mode == PDMK_A_COLK || g.b_mode != PDMK_B_ROWK || g.out_f32 || g.epilogue) return -2;
        if (g.rows_per_b <= 0 || (g.rows_per_b % 64) || (g.M % 64) || (g.N % 8) || (g.ldc % 8) || (g.R && (g.ldr % 8)) ||
            g.splitk > 1 || g.accumulate == 2 || g.cs_col0 < 0 || g.cs_ld < g.cs_col0 + g.N || (g.K % 8))
            return -1;
        if (g.a_mode == PDMK_A_CONV && g.conv_mode >= 9) return -1;
        if (!ring_mode()) return -2;
    }
    return 0;
}

int pdmk_gemm_rowblock_launch(const pdmk_gemm_args& g, hipStream_t st, long a_bytes, long b_bytes, int id, bool dry);   // gemm_rowblock.hip
int pdmk_gemm_rowblock_num_configs();
// LayerNorm prologue (pdmk_gemm_args.ln_gamma): the row-block candidate whose register image holds K, or -1
static int ln_candidate(const pdmk_gemm_args& g, long* ab, long* bb) {
    if (g.dtype != PDMK_BF16 || g.a_mode != PDMK_A_ROWK || g.b_mode != PDMK_B_ROWK || g.splitk > 1 || g.colstat || g.out_f32 ||
        !ring_mode() || !operand_bytes(g, ab, bb))
        return -1;
    for (int id = 0; id < pdmk_gemm_rowblock_num_configs(); ++id)
        if (pdmk_gemm_rowblock_launch(g, nullptr, *ab, *bb, id, true) == 0) return id;
    return -1;
}
extern "C" int pdmk_gemm_ln_supported(const pdmk_gemm_args* a) {
    if (!a || !a->ln_gamma || validate_args(*a)) return 0;
    long ab, bb;
    return ln_candidate(*a, &ab, &bb) >= 0 ? 1 : 0;
}

extern "C" int pdmk_gemm(const pdmk_gemm_args* a, pdmk_stream stream) {
    if (!a) return -1;
    const pdmk_gemm_args& g = *a;
    if (const int vrc = validate_args(g)) return vrc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    g_last_candidate = 0;
    if (g.ln_gamma) {                                   // one kernel family serves it: no plan, no tuning
        long ab, bb;
        const int id = ln_candidate(g, &ab, &bb);
        if (id < 0) return -2;
        const int rc = pdmk_gemm_rowblock_launch(g, st, ab, bb, id, false);
        g_last_candidate = 1 + pdmk_gemm_ring_num_configs() - pdmk_gemm_rowblock_num_configs() + id;
        return rc == 1 ? -2 : rc;
    }
    if (!(ring_eligible(g) && (g.K % 8) == 0) && !wgrad_eligible(g)) return launch_legacy(g, st);
    if ((g.a_mode == PDMK_A_COLK ? forced_wcfg() >= 0 : forced_cfg() >= 0) && !(g.epilogue && forced_cfg() == 0)) {
        const int rc = launch_candidate(g, st, g.a_mode == PDMK_A_COLK ? forced_wcfg() : forced_cfg());
        g_last_candidate = rc == 1 ? 0 : (g.a_mode == PDMK_A_COLK ? forced_wcfg() : forced_cfg());
        if (g.epilogue && rc == 1) return -2;
        const bool phase = (g.a_mode == PDMK_A_CONV || g.b_mode == PDMK_B_COLK_CONV) && g.conv_mode >= 5;
        if (!((phase || g.colstat) && rc == 1)) return rc == 1 ? launch_legacy(g, st) : rc;
        // a 2x2 phase conv the forced candidate does not serve has no K-step-32 form to fall back to: its own plan below
    }
    const int sk = g.splitk > 1 ? g.splitk : 1;
    int id;
    {
        std::lock_guard<std::mutex> lk(g_plan_mu);
        plan_file_load();
        const PlanKey key = make_key(g, sk);
        auto it = g_plan_cfg.find(key);
        if (it != g_plan_cfg.end()) id = it->second;
        else if (can_tune(st)) {
            float t;
            id = tune_cfg(g, st, sk, &t);
            if (id < 0) id = heuristic_cfg(g);
            g_plan_cfg[key] = id;
            plan_file_append('c', key, id);
        } else id = heuristic_cfg(g);                   // not cached: a later eager call may still tune it
    }
    if ((g.epilogue || g.colstat) && id <= 0) id = 1 + pdmk_gemm_ring_pick(g);
    int rc = launch_candidate(g, st, id);
    if (g.colstat && rc == 1) {                          // a plan the statistics epilogue does not exist for (row-block kernel)
        id = 1 + pdmk_gemm_ring_pick(g);
        rc = launch_candidate(g, st, id);
    }
    if (g.epilogue && rc == 1) return -2;               // no ring kernel takes this shape: the caller uses the two-pass form
    g_last_candidate = rc == 1 ? 0 : id;
    return rc == 1 ? launch_legacy(g, st) : rc;
}

// ---------------------------------------------------------------------------------------------------------------
// pdmk_gemm_group: several independent problems in one launch (include/pdmk.h).  A group is launched with ONE kernel shape;
// which one - or whether the members go out one by one after all - is measured once per group of shapes and cached.
// ---------------------------------------------------------------------------------------------------------------
int pdmk_gemm_ring_group_launch(const pdmk_gemm_args* gs, int n, hipStream_t st, const long* a_bytes, const long* b_bytes, int id);
int pdmk_wgrad_ring_group_launch(const pdmk_gemm_args* gs, int n, hipStream_t st, const long* a_bytes, const long* b_bytes, int id);
static int group_mode() { static int c = INT32_MIN; return env_int("PDMK_GEMM_GROUP", 1, &c); }   // 0: always one by one
static int forced_group() { static int c = INT32_MIN; return env_int("PDMK_GROUP_CFG", -1, &c); } // >= 1: group with that candidate (tests)

namespace {

std::map<std::vector<int>, int> g_group_plan;      // concatenated member keys -> candidate id of the grouped launch, -1 = separate

// planned candidate of one problem (the lookup / tune / heuristic part of pdmk_gemm)
int member_plan(const pdmk_gemm_args& g, hipStream_t st) {
    const int sk = g.splitk > 1 ? g.splitk : 1;
    std::lock_guard<std::mutex> lk(g_plan_mu);
    plan_file_load();
    const PlanKey key = make_key(g, sk);
    auto it = g_plan_cfg.find(key);
    if (it != g_plan_cfg.end()) return it->second;
    if (!can_tune(st)) return heuristic_cfg(g);
    float t;
    int id = tune_cfg(g, st, sk, &t);
    if (id < 0) id = heuristic_cfg(g);
    g_plan_cfg[key] = id;
    plan_file_append('c', key, id);
    return id;
}

int group_launch(const pdmk_gemm_args* a, int n, hipStream_t st, int id) {
    long ab[PDMK_GEMM_GROUP_MAX], bb[PDMK_GEMM_GROUP_MAX];
    for (int i = 0; i < n; ++i)
        if (!operand_bytes(a[i], &ab[i], &bb[i])) return 1;
    if (id <= 0) return 1;
    if (a[0].a_mode == PDMK_A_COLK) return pdmk_wgrad_ring_group_launch(a, n, st, ab, bb, id - 1);
    return pdmk_gemm_ring_group_launch(a, n, st, ab, bb, id - 1);
}

// microseconds per repetition of fn(), best of two rounds of `reps`
template <typename F> float time_us(F fn, hipStream_t st, hipEvent_t e0, hipEvent_t e1, int reps = 3) {
    if (fn() != 0) { (void)hipGetLastError(); return 1e30f; }
    float best = 1e30f;
    for (int round = 0; round < 2; ++round) {
        (void)hipEventRecord(e0, st);
        for (int r = 0; r < reps; ++r)
            if (fn() != 0) { (void)hipGetLastError(); return 1e30f; }
        (void)hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess) { (void)hipGetLastError(); return 1e30f; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best * 1000.f / reps;
}

// grouped candidate for these members (ids = their own plans), or -1: launch them one by one.  Caller checked can_tune().
int tune_group(const pdmk_gemm_args* a, int n, const int* ids, hipStream_t st) {
    // outputs go to scratch (an accumulating member must not be applied several times)
    pdmk_gemm_args t[PDMK_GEMM_GROUP_MAX];
    size_t off[PDMK_GEMM_GROUP_MAX + 1];
    off[0] = 0;
    for (int i = 0; i < n; ++i) {
        const size_t out = (size_t)out_rows(a[i]) * (size_t)a[i].N * 4;
        const size_t slabs = (a[i].accumulate == 2 && a[i].splitk > 1) ? (size_t)a[i].splitk : 1;
        off[i + 1] = off[i] + ((out * slabs + (size_t)a[i].M * 4 + 1023) & ~(size_t)1023);
    }
    {
        std::lock_guard<std::mutex> lk(g_plan_mu);
        if (!ensure_scratch(off[n] + 256)) return -1;
    }
    (void)hipDeviceSynchronize();
    for (int i = 0; i < n; ++i) {
        t[i] = a[i];
        char* base = reinterpret_cast<char*>(g_scratch) + off[i];
        t[i].C = base;
        t[i].ldc = a[i].N;
        t[i].colstat = nullptr;
        if (t[i].epilogue == PDMK_EPI_GEGLU_BWD) t[i].ldc = 2 * a[i].N;
        else t[i].C2 = nullptr;
        if (t[i].accumulate == 1) t[i].accumulate = 0;
        if (t[i].epilogue == PDMK_EPI_GEGLU) t[i].ldc = a[i].N / 2;
        if (t[i].colsum_out)
            t[i].colsum_out = reinterpret_cast<float*>(base + (off[i + 1] - off[i]) - (((size_t)a[i].M * 4 + 15) & ~(size_t)15));
    }
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { (void)hipGetLastError(); return -1; }
    const float t_sep = time_us([&]() -> int {
        for (int i = 0; i < n; ++i) {
            int rc = launch_candidate(t[i], st, ids[i]);
            if (rc == 1) rc = launch_legacy(t[i], st);
            if (rc) return rc;
        }
        return 0;
    }, st, e0, e1);
    int best = -1;
    float bt = t_sep * 0.98f;                           // a group must win by > 2 %
    // candidates: the members' own plans, the static heuristic's pick for the first member and - for convs - the halo shapes
    // (a shape that loses as a single 256-workgroup launch can win once four problems fill the grid)
    int cands[PDMK_GEMM_GROUP_MAX + 12], nc = 0;
    for (int i = 0; i < n; ++i) cands[nc++] = ids[i];
    cands[nc++] = heuristic_cfg(a[0]);
    if (a[0].a_mode == PDMK_A_CONV)
        for (int h = 0; h < 4; ++h) cands[nc++] = 1 + 12 + h;            // 1 + kNumBase + h (gemm_ring.hip)
    // Linear weight gradients (the block groups of pdm._pdmk.wgrad_group): every ring tile shape - a member's own plan was timed
    // with that member alone on the chip, where small tiles win by workgroup count; a full group has the workgroups
    if (a[0].a_mode == PDMK_A_COLK && a[0].b_mode == PDMK_B_COLK)
        for (int w = 0; w < 5; ++w) cands[nc++] = 1 + w;                 // kWCfgs (gemm_ring.hip)
    for (int i = 0; i < nc; ++i) {
        bool seen = cands[i] <= 0;
        for (int j = 0; j < i; ++j) seen = seen || cands[j] == cands[i];
        if (seen) continue;
        const float tg = time_us([&]() -> int { return group_launch(t, n, st, cands[i]); }, st, e0, e1);
        if (tg < bt) { bt = tg; best = cands[i]; }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (getenv("PDMK_TUNE_LOG")) {
        fprintf(stderr, "[pdmk group] n=%d", n);
        for (int i = 0; i < n; ++i) fprintf(stderr, " (%d,%d,%d|a%d sk%d id%d)", a[i].M, a[i].N, a[i].K, a[i].a_mode, a[i].splitk, ids[i]);
        fprintf(stderr, " separate %.1f us -> %s %d (%.1f us)\n", t_sep, best > 0 ? "grouped cand" : "separate", best, best > 0 ? bt : t_sep);
    }
    {
        std::lock_guard<std::mutex> lk(g_plan_mu);
        if (!scratch_check()) return -1;
    }
    return best;
}

}  // namespace

extern "C" int pdmk_gemm_group(const pdmk_gemm_args* a, int n, pdmk_stream stream, int32_t* grouped_out) {
    if (grouped_out) *grouped_out = 0;
    if (!a || n < 1 || n > PDMK_GEMM_GROUP_MAX) return -1;
    bool any_ln = false;
    for (int i = 0; i < n; ++i) {
        if (const int vrc = validate_args(a[i])) return vrc;
        any_ln = any_ln || a[i].ln_gamma;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    auto separate = [&]() -> int {
        for (int i = 0; i < n; ++i)
            if (const int rc = pdmk_gemm(&a[i], stream)) return rc;
        return 0;
    };
    if (n > 1 && forced_group() >= 1 && !any_ln) {                 // tests: this candidate, grouped, no timing
        const int rc = group_launch(a, n, st, forced_group());
        if (rc == 0) {
            g_last_candidate = forced_group();
            if (grouped_out) *grouped_out = n;
            return 0;
        }
        return rc == 1 ? separate() : rc;
    }
    if (n == 1 || any_ln || !group_mode() || !ring_mode() || forced_cfg() >= 0 || forced_wcfg() >= 0) return separate();
    bool fw = true, wg = true;
    for (int i = 0; i < n; ++i) {
        fw = fw && ring_eligible(a[i]) && (a[i].K % 8) == 0 && (a[i].a_mode == a[0].a_mode);
        wg = wg && wgrad_eligible(a[i]) && (a[i].b_mode == a[0].b_mode);
    }
    if (!fw && !wg) return separate();
    int ids[PDMK_GEMM_GROUP_MAX];
    std::vector<int> key;
    for (int i = 0; i < n; ++i) {
        ids[i] = member_plan(a[i], st);
        if ((a[i].epilogue || a[i].colstat) && ids[i] <= 0) ids[i] = 1 + pdmk_gemm_ring_pick(a[i]);
        const PlanKey k = make_key(a[i], a[i].splitk > 1 ? a[i].splitk : 1);
        key.insert(key.end(), k.v, k.v + 10);
        key.push_back(a[i].accumulate);
    }
    int id = -2;
    {
        std::lock_guard<std::mutex> lk(g_plan_mu);
        auto it = g_group_plan.find(key);
        if (it != g_group_plan.end()) id = it->second;
    }
    if (id == -2) {
        if (can_tune(st)) {
            id = tune_group(a, n, ids, st);
            std::lock_guard<std::mutex> lk(g_plan_mu);
            g_group_plan[key] = id;
        } else {                                          // unseen under capture: group only what needs no decision
            id = ids[0];
            for (int i = 1; i < n; ++i)
                if (ids[i] != ids[0]) id = -1;
        }
    }
    if (id > 0) {
        const int rc = group_launch(a, n, st, id);
        if (rc == 0) {
            g_last_candidate = id;
            if (grouped_out) *grouped_out = n;
            return 0;
        }
        if (rc != 1) return rc;
    }
    return separate();
}

/* Candidate the calling thread's last pdmk_gemm used (0 = K-step-32 kernels, 1.. = LDS-DMA ring shapes) and the kernel
 * symbol a profiler shows for it; measurement only (bench.py labels its HIP-event timings with these). */
extern "C" int pdmk_gemm_last_candidate(void) { return g_last_candidate; }
int pdmk_gemm_ring_name(int id, int conv, char* buf, int n);    // gemm_ring.hip
int pdmk_wgrad_ring_name(int id, int conv, char* buf, int n);
extern "C" int pdmk_gemm_candidate_name(int a_mode, int b_mode, int id, char* buf, int n) {
    if (!buf || n <= 0) return -1;
    if (id <= 0) {
        snprintf(buf, n, "igemm_kernel / pdmk_dma::igemm_dma_kernel (K-step-32)");
        return 0;
    }
    if (a_mode == PDMK_A_COLK) return pdmk_wgrad_ring_name(id - 1, b_mode == PDMK_B_COLK_CONV, buf, n);
    return pdmk_gemm_ring_name(id - 1, a_mode == PDMK_A_CONV, buf, n);
}

/* Split-K factor for a forward / dgrad GEMM (1 = do not split); see include/pdmk.h. */
extern "C" int pdmk_gemm_plan(const pdmk_gemm_args* a, pdmk_stream stream, int32_t* splitk_out) {
    if (!a || !splitk_out) return -1;
    const pdmk_gemm_args& g = *a;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    *splitk_out = 1;
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return -1;
    if ((g.N & 3) || g.epilogue) return 0;           // fused epilogues are never split
    const bool wg = wgrad_eligible(g);
    if (g.a_mode == PDMK_A_COLK && !wg) {
        *splitk_out = heuristic_wgrad_sk(g);
        return 0;
    }
    if (!wg && (!ring_eligible(g) || (g.K % 8) || forced_cfg() >= 0)) {
        *splitk_out = heuristic_sk(g);
        return 0;
    }
    std::lock_guard<std::mutex> lk(g_plan_mu);
    plan_file_load();
    const PlanKey key0 = make_key(g, 0);
    auto it = g_plan_sk.find(key0);
    if (it != g_plan_sk.end()) { *splitk_out = it->second; return 0; }
    if (!can_tune(st) || !a->A || !a->B) { *splitk_out = wg ? heuristic_wgrad_sk(g) : heuristic_sk(g); return 0; }
    const int tiles = ((g.M + 127) / 128) * ((g.N + 127) / 128);
    const int nk = (g.K + 63) / 64;
    const int cands[12] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64};
    int best_sk = 1;
    float bt = 1e30f;
    for (int c = 0; c < 12; ++c) {
        const int sk = cands[c];
        if (wg) {      // weight gradients: the reduction (pixels) is the long dimension, the output is small
            if (sk > 1 && (nk / sk < 4 || (long)tiles * sk > 1280)) continue;
            if ((long)tiles * sk * 4 < 128 && sk < 64 && nk / (2 * sk) >= 4) continue;      // far too few workgroups
        } else if (sk > 16 || (sk > 1 && (tiles > 200 || nk < 10 || nk / sk < 3 || (long)tiles * sk > 768))) continue;
        float t = 1e30f;
        const int id = tune_cfg(g, st, sk, &t);
        if (id < 0) continue;
        g_plan_cfg[make_key(g, sk)] = id;
        plan_file_append('c', make_key(g, sk), id);
        // a split must win by > 3 % (PDMK_SPLIT_MARGIN=<percent>: A/B knob - in the step the finish pass carries bias / residual /
        // statistics and runs from cold operands, which the back-to-back timing launches do not see)
        static const float margin = getenv("PDMK_SPLIT_MARGIN") ? 1.0f - 0.01f * (float)atof(getenv("PDMK_SPLIT_MARGIN")) : 0.97f;
        if (t < bt * (sk > 1 ? margin : 1.0f)) { bt = t; best_sk = sk; }
    }
    g_plan_sk[key0] = best_sk;
    plan_file_append('s', key0, best_sk);
    *splitk_out = best_sk;
    return 0;
}

/* Plan-cache control (include/pdmk.h): the cache is process-global state behind g_plan_mu. */
extern "C" int pdmk_plan_size(void) {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    return (int)(g_plan_cfg.size() + g_plan_sk.size());
}
extern "C" int pdmk_plan_export(const char* path) {
    if (!path) return -1;
    std::lock_guard<std::mutex> lk(g_plan_mu);
    FILE* f = fopen(path, "w");
    if (!f) return -1;
    fprintf(f, "pdmk-plan %d\n", pdmk_version());
    for (int pass = 0; pass < 2; ++pass)
        for (const auto& kv : (pass ? g_plan_sk : g_plan_cfg)) {
            const PlanKey& k = kv.first;
            fprintf(f, "%c %d %d %d %d %d %d %d %d %d %d %d\n", pass ? 's' : 'c', k.v[0], k.v[1], k.v[2], k.v[3], k.v[4], k.v[5],
                    k.v[6], k.v[7], k.v[8], k.v[9], kv.second);
        }
    return fclose(f) == 0 ? 0 : -1;
}
extern "C" int pdmk_plan_import(const char* path) {
    if (!path) return -1;
    std::lock_guard<std::mutex> lk(g_plan_mu);
    FILE* f = fopen(path, "r");
    if (!f) return -1;
    int ver = 0;
    if (fscanf(f, " pdmk-plan %d", &ver) != 1 || ver != pdmk_version()) {
        fclose(f);
        return -2;                                       // candidate numbering of another build
    }
    char kind;
    PlanKey k;
    int val, n = 0;
    while (fscanf(f, " %c %d %d %d %d %d %d %d %d %d %d %d", &kind, &k.v[0], &k.v[1], &k.v[2], &k.v[3], &k.v[4], &k.v[5],
                  &k.v[6], &k.v[7], &k.v[8], &k.v[9], &val) == 12) {
        if (kind == 'c') g_plan_cfg[k] = val;
        else if (kind == 's') g_plan_sk[k] = val;
        ++n;
    }
    fclose(f);
    return n;
}
extern "C" int pdmk_plan_clear(void) {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    g_plan_cfg.clear();
    g_plan_sk.clear();
    g_group_plan.clear();
    if (g_scratch) (void)hipFree(g_scratch);
    g_scratch = nullptr;
    g_scratch_bytes = 0;
    return 0;
}

// tuner-scratch overruns seen since the process started (PDMK_DEBUG_SCRATCH=1: a canary band behind the scratch, checked after
// every tuning pass); always 0 when the debug mode is off
extern "C" int pdmk_debug_scratch_violations(void) {
    std::lock_guard<std::mutex> lk(g_plan_mu);
    return g_scratch_violations;
}

extern "C" int pdmk_version(void) { return 121; }
