// Implicit GEMM, LDS-DMA ring (bf16; A = activation rows or 3x3 gather, B = weights [N][K]): forward / dgrad work-horse.
//   * K-step 64: every DMA wave-instruction (buffer_load_dwordx4 ... lds, 1 KiB) moves 8 tile rows x 128 B, i.e. FULL
//     128-byte lines of the source rows (half the texture-addresser work per byte of the 16 x 64 B pieces of a K-step-32
//     tile); out-of-range lanes (conv halo, M/N/K tails) are zero-filled into LDS by the buffer unit;
//   * one 512-thread workgroup per CU owns (almost) the whole 160 KiB LDS as a ring of STAGES (A BMx64 + B BNx64) bf16
//     tiles: STAGES-1 K-steps stay in flight behind a counted s_waitcnt vmcnt(N) + one raw s_barrier per K-step, so
//     L2 / HBM latency is covered by prefetch distance, not by occupancy;
//   * tile shapes BM in {64,128,256} x BN in {128,160}: the host picks the shape whose grid fills the 256 CUs evenly
//     (SD-2.1 widths are multiples of 160: 320 / 640 / 1280) - see pick_config();
//   * LDS image is lane-linear (what the DMA writes); ds_read_b128 bank conflicts are removed by XOR-ing the 16-byte
//     chunk index with (row>>1)&7 on the SOURCE address and on the fragment reads (same involution on both sides).
#include "common.h"

#include <stdio.h>
#include <stdlib.h>

#ifndef PDMK_WGRAD_XCD
#define PDMK_WGRAD_XCD 1      // weight gradients: all tiles of a split on one XCD (0: split = blockIdx.y, tiles round-robin)
#endif
#ifndef PDMK_HALO_PAIRS
#define PDMK_HALO_PAIRS 1     // halo conv, rings of >= 4 slots: two taps per barrier (0: one, as in round 1)
#endif
#ifndef PDMK_RING_SPLIT_ISSUE
#define PDMK_RING_SPLIT_ISSUE 0   // ring GEMM: a stage's DMA issue split by wave half (igemm_ring_body).  Measured (round 4): +-0 on the C x C shapes
                                  // in isolation, -0.5 % for the step (187.8 -> 186.9 images/s) - unlike the halo conv, whose events carry 6-8 pieces
                                  // per wave against 20-40 MFMAs, a ring stage is 3-5 pieces against 20 - so it is OFF
#endif
#ifndef PDMK_HALO_SPLIT_ISSUE
#define PDMK_HALO_SPLIT_ISSUE 1   // ... and the event's DMA issue split by wave half (conv_halo_body); 2 = four issue points (waves 2, 3 / 6, 7 between
                                  // a tap's two MFMA groups): measured 3-9 % SLOWER than 1 (64^2 320 -> 320: 71.2 -> 77.5 us), off
#endif

#ifndef PDMK_PRIO
#define PDMK_PRIO 0               // wave priority experiments (MI355X_MICROARCH.md "Two waves per SIMD"): 1 = ONE s_setprio 1 for the second-dispatched
                                  // half of the workgroup (waves 4-7) before the K-loop; 2 = s_setprio 1 / 0 around every MFMA cluster.
                                  // Measured (round 4, ring + halo + weight-gradient bodies, same box, two builds): 193.4 images/s (0) vs 192.8 (1)
                                  // vs 192.5 (2) - nothing to gain for these loops, OFF
#endif

namespace pdmk_ring {

constexpr int BK = 64, NT = 512;
constexpr unsigned OOB = 0x80000000u;
typedef __attribute__((address_space(3))) void lds_void;

struct ConvGeom {
    int hi, wi, ci, ho, wo, ld, mode;
};

// Launch coordinates of a workgroup inside ITS problem: (tile id, split id, tiles, splits).  A plain launch passes
// blockIdx / gridDim; a grouped launch (pdmk_gemm_group: up to PDMK_GEMM_GROUP_MAX problems of one kernel shape in one grid)
// maps the linear workgroup id to (problem, tile, split).  The kernel bodies below are written against LC only.
struct LC {
    unsigned bx, by, gx, gy;
};
__device__ __forceinline__ LC lc_plain() { return LC{blockIdx.x, blockIdx.y, gridDim.x, gridDim.y}; }
// Split launches (weight gradients: the reduction is cut into gridDim.y splits): ALL tiles of one split on ONE XCD.  The tiles
// of a split read the same k-rows of both operands (different column tiles); dealt round-robin over the 8 XCDs by their linear
// id they fetch those rows through up to 8 different L2s (PMC, round 2/3: 90 MB read per launch of the 320 x 320 Linear weight
// gradient against 42 MB of operands).  The linear id is mapped to an XCD-major logical id first (the workgroups of one XCD
// get consecutive ids), and that id is (split, tile) with the tile fastest - which split lands on which XCD is irrelevant,
// and split -> k-range is unchanged, so slab sums stay bit-identical.
__device__ __forceinline__ LC lc_split_xcd() {
    const unsigned gx = gridDim.x, gy = gridDim.y;
    if (gy == 1) return LC{blockIdx.x, 0u, gx, 1u};
    const unsigned t = (unsigned)xcd_remap((int)(blockIdx.y * gx + blockIdx.x), (int)(gx * gy));
    const unsigned z = t / gx;
    return LC{t - z * gx, z, gx, gy};
}
// problem of this workgroup in a grouped launch; false = padding workgroup (every problem starts on a multiple of 8 blocks,
// so that wgc.bx % 8 - the XCD under round-robin placement - equals the local id % 8 the tile remap assumes)
__device__ __forceinline__ bool lc_group(const pdmk_gemm_group_dev& gg, int& pi, LC& lc) {
    const unsigned b = blockIdx.x;
    pi = 0;
#pragma unroll
    for (int i = 1; i < PDMK_GEMM_GROUP_MAX; ++i)
        if (i < gg.n && b >= (unsigned)gg.start[i]) pi = i;
    const unsigned local = b - (unsigned)gg.start[pi];
    const unsigned gx = (unsigned)gg.gx[pi], gy = (unsigned)gg.gy[pi];
    if (local >= gx * gy) return false;
    const unsigned t = gy > 1 ? (unsigned)xcd_remap((int)local, (int)(gx * gy)) : local;     // all tiles of a split on one XCD (lc_split_xcd)
    const unsigned z = t / gx;
    lc = LC{t - z * gx, z, gx, gy};
    return true;
}

// source pixel of output pixel (b, oy, ox) under tap (0..8), or -1 (zero padding / zero-insertion hole / tap >= 9)
__device__ __forceinline__ int conv_src_pixel(const ConvGeom& g, int b, int oy, int ox, int tap) {
    if (g.mode >= 5) {            // 2x2 phase (a, b') of an upsampling conv: tap (dy, dx) reads pixel (oy + dy - 1 + a, ox + dx - 1 + b')
        const int ph = g.mode - 5, vy = oy + (tap >> 1) - 1 + (ph >> 1), vx = ox + (tap & 1) - 1 + (ph & 1);
        const bool ok = (unsigned)vy < (unsigned)g.hi && (unsigned)vx < (unsigned)g.wi && tap < 4;
        return ok ? (b * g.hi + vy) * g.wi + vx : -1;
    }
    const int ky = (tap * 11) >> 5;
    const int kx = tap - 3 * ky;
    const bool s2 = g.mode == 1 || g.mode == 4, up = g.mode == 2 || g.mode == 3;
    const int pad = g.mode == 4 ? 0 : 1;                  // mode 4: padding on the bottom / right only
    const int vy = (s2 ? 2 * oy : oy) + ky - pad, vx = (s2 ? 2 * ox : ox) + kx - pad;
    const int hv = up ? 2 * g.hi : g.hi, wv = up ? 2 * g.wi : g.wi;
    bool ok = (unsigned)vy < (unsigned)hv && (unsigned)vx < (unsigned)wv && tap < 9;
    if (g.mode == 3) ok = ok && (((vy | vx) & 1) == 0);
    const int iy = up ? (vy >> 1) : vy, ix = up ? (vx >> 1) : vx;
    return ok ? (b * g.hi + iy) * g.wi + ix : -1;
}

__device__ __forceinline__ void wait_vmcnt_dyn(int n) { pdmk_wait_vmcnt(n); }   // n is wave-uniform
__device__ __forceinline__ void prio_static(int wave) {       // wave is an SGPR value (readfirstlane): a scalar branch around one s_setprio
    if (PDMK_PRIO == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);
}
#define PDMK_PRIO_UP() do { if (PDMK_PRIO == 2) __builtin_amdgcn_s_setprio(1); } while (0)
#define PDMK_PRIO_DOWN() do { if (PDMK_PRIO == 2) __builtin_amdgcn_s_setprio(0); } while (0)

// Epilogue shared by the ring kernels (same contract as gemm.hip): accumulators -> LDS staging image (64 rows per pass) ->
// 16-byte rows of C with bias / rowvec / residual / accumulate fused, or fp32 atomics for split-K launches.
// TILE2D (halo conv on images wider than a tile): the tile is rows x tw output pixels of an image of width img_w, m0 is
// its first pixel and local row l is pixel m0 + (l / tw) * img_w + l % tw.
// OUTMAP (2x2 phase of an upsampling conv, conv_mode 5..8): row m is pixel (b, y, x) of the hi x wi grid the GEMM enumerates
// and is STORED at pixel (b, 2y + oa, 2x + ob) of the 2hi x 2wi output image (bounds, bias and rowvec still go by m).
template <int BM, int NJ, int LDS_BYTES, bool TILE2D = false, bool OUTMAP = false>
__device__ __forceinline__ void ring_epilogue(const pdmk_gemm_args& g, const LC wgc, f32x4 (&acc)[BM / 64][NJ], unsigned char* smem,
                                              int m0, int n0, int tw = 0, int img_w = 0) {
    auto row_of = [&](int l) { return TILE2D ? m0 + (l / tw) * img_w + (l % tw) : m0 + l; };
    auto out_row = [&](int m) -> long {
        if (!OUTMAP) return (long)m;
        const int hw = g.conv_hi * g.conv_wi, b = m / hw, rem = m - b * hw, y = rem / g.conv_wi, x = rem - y * g.conv_wi;
        const int ph = g.conv_mode - 5;
        return ((long)(b * 2 * g.conv_hi + 2 * y + (ph >> 1)) * (2 * g.conv_wi) + 2 * x + (ph & 1));
    };
    constexpr int BN = 32 * NJ, IM = BM / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: wave-dependent branches and counts stay in SGPRs
    const int wm = wave >> 1, wn = wave & 1;
    const bool first = wgc.by == 0;
    const bool slab = g.accumulate == 2;                 // split-K partials to slab wgc.by of a [splitk][M][ldc] workspace
    const bool atomic = wgc.gy > 1 && !slab;
    const bool acc1 = g.accumulate == 1;
    const bool f32out = g.out_f32 != 0;
    const bool vec8 = ((g.N & 7) == 0) && ((g.ldc & 7) == 0) && (g.R == nullptr || (g.ldr & 7) == 0);
    constexpr int SROW = BN + 4;
    constexpr int C8 = BN / 8;                                        // 8-column chunks per tile row
    constexpr int ITEMS = (64 * C8 + NT - 1) / NT;                    // (row, chunk) items per thread and pass
    constexpr int RS = NT / C8, CS_STRIDE = RS * C8, ITEMS_CS = (64 + RS - 1) / RS;   // the statistics epilogue's item mapping
    constexpr bool CS_OK = RS * 2 * BN * 4 <= LDS_BYTES;   // the partials fit the (drained) ring: every forward shape (NJ >= 2)
    static_assert(ITEMS_CS >= ITEMS && (CS_OK || NJ < 2), "statistics epilogue of the forward tile shapes");
    float* stage = reinterpret_cast<float*>(smem);
    static_assert(64 * SROW * 4 <= LDS_BYTES, "staging image must fit the ring");
    float* Cf = reinterpret_cast<float*>(g.C) + (slab ? (long)wgc.by * g.M * g.ldc : 0L);
    bf16* Ct = reinterpret_cast<bf16*>(g.C);
    const bf16* Rp = reinterpret_cast<const bf16*>(g.R);
    float cs1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, cs2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // colstat sums
#pragma unroll
    for (int pass = 0; pass < BM / 64; ++pass) {
        __syncthreads();
        if ((wm * 16 * IM) / 64 == pass) {
            const int lrb = (wm * 16 * IM) % 64;
#pragma unroll
            for (int i = 0; i < IM; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    f32x4 v = acc[i][j];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= g.alpha;
                    *reinterpret_cast<f32x4*>(stage + (lrb + i * 16 + (lane & 15)) * SROW + wn * (16 * NJ) + j * 16 + (lane >> 4) * 4) = v;
                }
        }
        if (atomic) {
            __syncthreads();
            for (int rr = 0; rr < 8; ++rr) {
                const int lr2 = wave * 8 + rr, m = row_of(pass * 64 + lr2);
                if (m >= g.M) break;
#pragma unroll
                for (int h = 0; h < (BN + 63) / 64; ++h) {
                    const int cl = h * 64 + lane, n = n0 + cl;
                    if (cl < BN && n < g.N) {
                        float v = stage[lr2 * SROW + cl];
                        if (first) {
                            if (g.bias) v += g.bias[n];
                            if (g.rowvec) v += g.rowvec[(long)(m / g.rows_per_b) * (g.ldrv ? g.ldrv : g.N) + n];
                            if (Rp) v += (float)Rp[(long)m * g.ldr + n];
                        }
                        unsafeAtomicAdd(Cf + (long)m * g.ldc + n, v);
                    }
                }
            }
        } else if (vec8 && g.epilogue == PDMK_EPI_GEGLU) {
            // GEGLU (blocks.py:44-59) on the staged fp32 tile: columns come as (hidden, gate) blocks of 8; a thread takes
            // one 16-column pair of a row, writes 8 outputs hidden * gelu(gate) and, for the backward, the pair itself
            constexpr int C16 = BN / 16;
            constexpr int ITEMS2 = (64 * C16 + NT - 1) / NT;
            bf16* C2 = reinterpret_cast<bf16*>(g.C2);
            __syncthreads();
#pragma unroll
            for (int it = 0; it < ITEMS2; ++it) {
                const int item = tid + it * NT;
                const int lr2 = item / C16, p = item - lr2 * C16;
                const int m = row_of(pass * 64 + lr2), n = n0 + p * 16;
                if (!(item < 64 * C16 && m < g.M && n < g.N)) continue;
                float v[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(stage + lr2 * SROW + p * 16 + q * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[q * 4 + r] = t4[r];
                }
                if (g.bias) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 b4 = *reinterpret_cast<const float4*>(g.bias + n + q * 4);
                        v[q * 4] += b4.x; v[q * 4 + 1] += b4.y; v[q * 4 + 2] += b4.z; v[q * 4 + 3] += b4.w;
                    }
                }
                bf16x8 hb, gb, o;
#pragma unroll
                for (int r = 0; r < 8; ++r) { hb[r] = (bf16)v[r]; gb[r] = (bf16)v[8 + r]; }
                if (C2) {
                    st_stream(reinterpret_cast<bf16x8*>(C2 + (long)m * g.ldc2 + n), hb);
                    st_stream(reinterpret_cast<bf16x8*>(C2 + (long)m * g.ldc2 + n + 8), gb);
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) o[r] = (bf16)((float)hb[r] * gelu_f((float)gb[r]));
                st_stream(reinterpret_cast<bf16x8*>(Ct + (long)m * g.ldc + (n >> 1)), o);
            }
        } else if (vec8 && g.epilogue == PDMK_EPI_GEGLU_BWD) {
            // input gradient THROUGH GEGLU (blocks.py:44-59 backward): the staged tile is d(hidden * gelu(gate))[m][n .. n+7];
            // C2 holds the forward pre-activation [M][2N] (hidden / gate interleaved in blocks of 8) and C receives its gradient in
            // the same layout: d hidden = d * gelu(gate), d gate = d * hidden * gelu'(gate).  d is rounded to bf16 first, so the
            // result is bit-identical to storing d and running pdmk_geglu_bwd(layout = 1) on it.
            const bf16* X2 = reinterpret_cast<const bf16*>(g.C2);
            bf16x8 hb[ITEMS], gb[ITEMS];
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int item = tid + it * NT;
                const int lr2 = item / C8, c8 = item - lr2 * C8;
                const int m = row_of(pass * 64 + lr2), n = n0 + c8 * 8;
                if (item < 64 * C8 && m < g.M && n < g.N) {
                    hb[it] = *reinterpret_cast<const bf16x8*>(X2 + (long)m * g.ldc2 + 2 * n);
                    gb[it] = *reinterpret_cast<const bf16x8*>(X2 + (long)m * g.ldc2 + 2 * n + 8);
                }
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int item = tid + it * NT;
                const int lr2 = item / C8, c8 = item - lr2 * C8;
                const int m = row_of(pass * 64 + lr2), n = n0 + c8 * 8;
                if (!(item < 64 * C8 && m < g.M && n < g.N)) continue;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + lr2 * SROW + c8 * 8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + lr2 * SROW + c8 * 8 + 4);
                bf16x8 dh, dg;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float d = (float)(bf16)(r < 4 ? lo[r] : hi[r - 4]);
                    const float hv = (float)hb[it][r], gv = (float)gb[it][r];
                    dh[r] = (bf16)(d * gelu_f(gv));
                    dg[r] = (bf16)(d * hv * gelu_grad_f(gv));
                }
                st_stream(reinterpret_cast<bf16x8*>(Ct + (long)m * g.ldc + 2 * n), dh);
                st_stream(reinterpret_cast<bf16x8*>(Ct + (long)m * g.ldc + 2 * n + 8), dg);
            }
        } else if (vec8) {
            // the residual / previous-output reads are issued BEFORE the barrier so that their latency overlaps it
            // with statistics (colstat) a thread keeps ONE 8-column chunk for the whole tile: sweeps of RS whole rows by the first
            // RS * C8 threads (the same mapping as the plain one whenever C8 divides the workgroup: every shape but the 160-wide)
            bf16x8 rres[ITEMS_CS], cprev[ITEMS_CS];
            const int istride = g.colstat ? CS_STRIDE : NT;
            const bool tid_ok = !g.colstat || tid < CS_STRIDE;
#pragma unroll
            for (int it = 0; it < ITEMS_CS; ++it) {
                const int item = tid + it * istride;
                const int lr2 = item / C8, c8 = item - lr2 * C8;
                const int m = row_of(pass * 64 + lr2), n = n0 + c8 * 8;
                const bool ok = tid_ok && item < 64 * C8 && m < g.M && n < g.N;
                const long off = out_row(m) * g.ldc + n;
                if (ok && Rp) rres[it] = *reinterpret_cast<const bf16x8*>(Rp + (long)m * g.ldr + n);
                if (ok && acc1 && !f32out) cprev[it] = *reinterpret_cast<const bf16x8*>(Ct + off);
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < ITEMS_CS; ++it) {
                const int item = tid + it * istride;
                const int lr2 = item / C8, c8 = item - lr2 * C8;
                const int m = row_of(pass * 64 + lr2), n = n0 + c8 * 8;
                if (!(tid_ok && item < 64 * C8 && m < g.M && n < g.N)) continue;
                float v[8];
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + lr2 * SROW + c8 * 8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + lr2 * SROW + c8 * 8 + 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = lo[r]; v[4 + r] = hi[r]; }
                if (g.bias) {
                    const float4 b0 = *reinterpret_cast<const float4*>(g.bias + n), b1 = *reinterpret_cast<const float4*>(g.bias + n + 4);
                    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                }
                if (g.rowvec) {
                    const float* rv = g.rowvec + (long)(m / g.rows_per_b) * (g.ldrv ? g.ldrv : g.N) + n;
                    const float4 b0 = *reinterpret_cast<const float4*>(rv), b1 = *reinterpret_cast<const float4*>(rv + 4);
                    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                }
                if (Rp) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += (float)rres[it][r];
                }
                const long off = out_row(m) * g.ldc + n;
                if (f32out) {
                    float4 o0 = make_float4(v[0], v[1], v[2], v[3]), o1 = make_float4(v[4], v[5], v[6], v[7]);
                    if (acc1) {       // fp32 accumulate (weight gradients without split-K): read in place
                        const float4 c0 = *reinterpret_cast<const float4*>(Cf + off), c1 = *reinterpret_cast<const float4*>(Cf + off + 4);
                        o0.x += c0.x; o0.y += c0.y; o0.z += c0.z; o0.w += c0.w;
                        o1.x += c1.x; o1.y += c1.y; o1.z += c1.z; o1.w += c1.w;
                    }
                    *reinterpret_cast<float4*>(Cf + off) = o0;
                    *reinterpret_cast<float4*>(Cf + off + 4) = o1;
                } else {
                    if (acc1) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] += (float)cprev[it][r];
                    }
                    bf16x8 o;
#pragma unroll
                    for (int r = 0; r < 8; ++r) o[r] = (bf16)v[r];
                    st_stream(reinterpret_cast<bf16x8*>(Ct + off), o);
                    if (g.colstat) {      // sums of the values AS STORED, per thread: its 8 columns over the rows it stores
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            const float x = (float)o[r];
                            cs1[r] += x;
                            cs2[r] += x * x;
                        }
                    }
                }
            }
            if (CS_OK && g.colstat) {
                // GroupNorm statistics (pdmk.h, colstat): the 64 rows of a pass belong to one image (rows_per_b % 64 == 0, whole
                // tiles); the per-thread sums run on over the passes of that image and are flushed when the image changes or the
                // tile ends: the RS threads of a chunk are added through the staging image (fixed order), the workgroups by 64-bit
                // fixed-point integer atomics (order-independent: bit-reproducible)
                const int mp0 = row_of(pass * 64);
                const int img = mp0 / g.rows_per_b;
                const bool flush = pass == BM / 64 - 1 || row_of((pass + 1) * 64) >= g.M || row_of((pass + 1) * 64) / g.rows_per_b != img;
                if (flush && mp0 < g.M) {
                    __syncthreads();
                    if (tid < CS_STRIDE) {
                        float* d = stage + (tid / C8) * (2 * BN) + (tid % C8) * 8;
                        *reinterpret_cast<f32x4*>(d) = f32x4{cs1[0], cs1[1], cs1[2], cs1[3]};
                        *reinterpret_cast<f32x4*>(d + 4) = f32x4{cs1[4], cs1[5], cs1[6], cs1[7]};
                        *reinterpret_cast<f32x4*>(d + BN) = f32x4{cs2[0], cs2[1], cs2[2], cs2[3]};
                        *reinterpret_cast<f32x4*>(d + BN + 4) = f32x4{cs2[4], cs2[5], cs2[6], cs2[7]};
                    }
                    __syncthreads();
                    for (int c = tid; c < 2 * BN; c += NT) {
                        const int col = c < BN ? c : c - BN;
                        if (n0 + col >= g.N) continue;
                        float sum = 0.f;
#pragma unroll
                        for (int r = 0; r < RS; ++r) sum += stage[r * (2 * BN) + c];
                        cs_add(g.colstat + ((long)img * 4 + (c < BN ? 0 : 2)) * g.cs_ld + g.cs_col0 + n0 + col, g.cs_ld, sum, c >= BN);
                    }
#pragma unroll
                    for (int r = 0; r < 8; ++r) cs1[r] = cs2[r] = 0.f;
                }
            }
        } else {
            __syncthreads();
            for (int item = tid; item < 64 * C8; item += NT) {
                const int lr2 = item / C8, c8 = item - lr2 * C8;
                const int m = row_of(pass * 64 + lr2), n = n0 + c8 * 8;
                if (m >= g.M || n >= g.N) continue;
                const float* rv = g.rowvec ? g.rowvec + (long)(m / g.rows_per_b) * (g.ldrv ? g.ldrv : g.N) : nullptr;
                const long off = out_row(m) * g.ldc + n;
                const int nv = min(8, g.N - n);
                for (int r = 0; r < nv; ++r) {
                    float x = stage[lr2 * SROW + c8 * 8 + r];
                    if (g.bias) x += g.bias[n + r];
                    if (rv) x += rv[n + r];
                    if (Rp) x += (float)Rp[(long)m * g.ldr + n + r];
                    if (f32out) Cf[off + r] = x + (acc1 ? Cf[off + r] : 0.f);
                    else Ct[off + r] = (bf16)(x + (acc1 ? (float)Ct[off + r] : 0.f));
                }
            }
        }
    }
}

constexpr int ring_smem_bytes(int BM, int NJ, int STAGES) { return STAGES * (BM * 128 + 32 * NJ * 128); }

template <bool CONV, int BM, int NJ, int STAGES, int OCC>
__device__ __forceinline__ void igemm_ring_body(const pdmk_gemm_args& g, unsigned a_bytes, unsigned b_bytes, const LC wgc,
                                                unsigned char* smem) {
    typedef Mma<bf16> MM;
    constexpr int BN = 32 * NJ;
    constexpr int IM = BM / 64;                            // 16-row MFMA tiles per wave (waves: 4 in M x 2 in N)
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, SLOT = A_BYTES + B_BYTES;
    constexpr int NA = BM / 64;                            // A DMA pieces per wave and stage (8 rows x 128 B each)
    constexpr int NBLK_B = BN / 8, NB = (NBLK_B + 7) / 8;  // B pieces: NBLK_B in all, wave w issues NBLK_B/8 + (w < NBLK_B%8)
    static_assert(STAGES >= 2 && STAGES * SLOT * (OCC / 2) <= 160 * 1024, "ring(s) must fit the 160 KiB LDS");
    static_assert(STAGES * SLOT == ring_smem_bytes(BM, NJ, STAGES), "smem size of the kernel wrappers");

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: wave-dependent branches and counts stay in SGPRs
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (g.N + BN - 1) / BN;
    const int tile = xcd_remap(wgc.bx, wgc.gx);
    // tile order: the workgroups of one XCD (consecutive logical tiles) share the operand panel that is worth more in its
    // L2 - the A rows (m-major) normally, the weight rows (n-major) for the 8x8-latent forward / dgrad layers, which stream
    // 15-60 MB of weights per launch for 512 output rows (n-major for every N > M measured 3 % slower for the step)
    const int ntm = (g.M + BM - 1) / BM;
    const bool nmajor = g.a_mode != PDMK_A_COLK && g.M <= 1024 && g.N > g.M;
    const int m0 = (nmajor ? tile % ntm : tile / ntn) * BM, n0 = (nmajor ? tile / ntm : tile % ntn) * BN;
    const int nk_total = (g.K + BK - 1) / BK;
    const int per = (nk_total + wgc.gy - 1) / wgc.gy;
    const int kt0 = wgc.by * per;
    const int kt1 = min(nk_total, kt0 + per);
    if (kt0 >= kt1 && g.accumulate != 2) return;        // slab split-K: an empty split still writes its (zero) slab

    const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), (short)0, (int)a_bytes, 0x00020000);
    const auto rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.B), (short)0, (int)b_bytes, 0x00020000);
    const ConvGeom cg{g.conv_hi, g.conv_wi, g.conv_ci, g.conv_ho, g.conv_wo, g.conv_ld, g.conv_mode};

    // ---- loader: lane -> (sub-row sr = lane>>3 of an 8-row piece, physical 16-byte chunk pc = lane&7).  Pieces of wave w
    // are blocks w, w+8, w+16, ... so every row this lane touches has row&15 == (w&1)*8 + sr: ONE swizzle per lane.
    const int sr = lane >> 3;
    const int lc = (lane & 7) ^ (((wave & 1) * 4) + (sr >> 1));      // logical chunk: k offset lc*8 inside the K-step
    unsigned a_base[NA], b_base[NB];
    int a_b[NA], a_oy[NA], a_ox[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        a_base[i] = OOB; a_b[i] = -1; a_oy[i] = 0; a_ox[i] = 0;
        const int m = m0 + (i * 8 + wave) * 8 + sr;
        if (m < g.M) {
            if (CONV) {
                const int hw = cg.ho * cg.wo;
                a_b[i] = m / hw;
                const int rem = m - a_b[i] * hw;
                a_oy[i] = rem / cg.wo;
                a_ox[i] = rem - a_oy[i] * cg.wo;
            } else {
                a_base[i] = (unsigned)m * (unsigned)g.lda * 2u;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int n = n0 + (i * 8 + wave) * 8 + sr;
        b_base[i] = (n < g.N) ? (unsigned)n * (unsigned)g.ldb * 2u : OOB;
    }
    const int nb_wave = NBLK_B / 8 + (wave < (NBLK_B % 8) ? 1 : 0);  // wave-uniform
    const int ndma = NA + nb_wave;                                   // this wave's DMA instructions per stage
    int kl = kt0 * BK + lc * 8;                                      // this lane's k of the NEXT stage to issue
    int tap = 0, cc = 0;
    if (CONV) {
        tap = kl / cg.ci;
        cc = kl - tap * cg.ci;
    }

    auto issue = [&](int slot) {
        unsigned char* sa = smem + slot * SLOT;
        const bool kok = kl < g.K;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            unsigned va;
            if (CONV) {
                const int px = a_b[i] >= 0 ? conv_src_pixel(cg, a_b[i], a_oy[i], a_ox[i], tap) : -1;
                va = px >= 0 ? ((unsigned)px * (unsigned)cg.ld + (unsigned)cc) * 2u : OOB;
            } else {
                va = (a_base[i] != OOB && kok) ? a_base[i] + (unsigned)kl * 2u : OOB;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void*)(sa + (i * 8 + wave) * 1024), 16, (int)va, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (i * 8 + wave < NBLK_B) {                             // wave-uniform
                const unsigned vb = (b_base[i] != OOB && kok) ? b_base[i] + (unsigned)kl * 2u : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_void*)(sa + A_BYTES + (i * 8 + wave) * 1024), 16, (int)vb, 0, 0, 0);
            }
        }
        kl += BK;
        if (CONV) {
            cc += BK;
            if (cc >= cg.ci) { cc -= cg.ci; ++tap; }
            if (cc >= cg.ci) { cc -= cg.ci; ++tap; }
        }
    };

    f32x4 acc[IM][NJ];
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addresses: row fr = lane&15 of a 16-row MFMA tile; logical chunk kk*4 + (lane>>4) -> physical ^ ((fr>>1)&7)
    const int fr = lane & 15;
    const int fsw = (fr >> 1) & 7;
    const unsigned fch0 = (unsigned)(((lane >> 4) ^ fsw) * 16), fch1 = (unsigned)(((4 + (lane >> 4)) ^ fsw) * 16);
    const unsigned a_row = (unsigned)(wm * (16 * IM) + fr) * 128u;
    const unsigned b_row = (unsigned)(wn * (16 * NJ) + fr) * 128u;

#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (kt0 + s < kt1) issue(s);

    int slot = 0;
    prio_static(wave);
    for (int kt = kt0; kt < kt1; ++kt) {
        // younger K-steps already in flight: STAGES - 2 except in the last steps.  Immediate waits where the count is known at
        // compile time (always for the two-slot rings, in the steady state for the deep ones, per wave class): the computed
        // jump of wait_vmcnt_dyn and its scalar set-up are ~80 cycles per K-step
        if (STAGES == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (kt + STAGES - 1 <= kt1) {
            if ((NBLK_B % 8) == 0 || wave < (NBLK_B % 8)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * (NA + NB)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * (NA + NB - 1)) : "memory");
        } else {
            wait_vmcnt_dyn(min(STAGES - 2, kt1 - 1 - kt) * ndma);
        }
        __builtin_amdgcn_s_barrier();
        // the stage's DMA instructions go into the slot read LAST step: anywhere inside this step will do.  Waves 0-3 issue them
        // first, their SIMD partners 4-7 between the two halves of the step's MFMAs (PDMK_RING_SPLIT_ISSUE; as in conv_halo_body:
        // eight waves in the memory pipe together right behind the barrier leave the matrix pipes idle)
        const bool late = PDMK_RING_SPLIT_ISSUE && wave >= 4;        // wave-uniform
        if (!late && kt + STAGES - 1 < kt1) issue(slot == 0 ? STAGES - 1 : slot - 1);
        const unsigned char* sa = smem + slot * SLOT;
        const unsigned char* sb = sa + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const unsigned fch = kk ? fch1 : fch0;
            bf16x8 af[IM], bf[NJ];
#pragma unroll
            for (int i = 0; i < IM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sa + a_row + fch + i * 2048);
#pragma unroll
            for (int j = 0; j < NJ; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(sb + b_row + fch + j * 2048);
            PDMK_PRIO_UP();
#pragma unroll
            for (int i = 0; i < IM; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = MM::mma(bf[j], af[i], acc[i][j]);
            PDMK_PRIO_DOWN();
            if (kk == 0 && late) {
                __builtin_amdgcn_sched_barrier(0);
                if (kt + STAGES - 1 < kt1) issue(slot == 0 ? STAGES - 1 : slot - 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        slot = slot + 1 == STAGES ? 0 : slot + 1;
    }

    if (PDMK_PRIO == 1) __builtin_amdgcn_s_setprio(0);
    ring_epilogue<BM, NJ, STAGES * SLOT>(g, wgc, acc, smem, m0, n0);
}

template <bool CONV, int BM, int NJ, int STAGES, int OCC>
__global__ __launch_bounds__(NT, OCC) void igemm_ring_kernel(pdmk_gemm_args g, unsigned a_bytes, unsigned b_bytes) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[ring_smem_bytes(BM, NJ, STAGES)];
    igemm_ring_body<CONV, BM, NJ, STAGES, OCC>(g, a_bytes, b_bytes, lc_plain(), smem);
}
// the same body for up to PDMK_GEMM_GROUP_MAX independent problems in one grid (teacher || student layer pairs, the weight
// gradients of several layers): longer launches, one boundary instead of several
template <bool CONV, int BM, int NJ, int STAGES, int OCC>
__global__ __launch_bounds__(NT, OCC) void igemm_ring_group_kernel(pdmk_gemm_group_dev gg) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[ring_smem_bytes(BM, NJ, STAGES)];
    int pi;
    LC wgc;
    if (!lc_group(gg, pi, wgc)) return;
    igemm_ring_body<CONV, BM, NJ, STAGES, OCC>(gg.p[pi], gg.a_bytes[pi], gg.b_bytes[pi], wgc, smem);
}

// ------------------------------------------------------------------------------------------------------------------
// Stride-1 3x3 convolution with the input halo staged ONCE per channel block ("halo conv"): an output tile is R image
// rows (or whole small images); for every 64-channel block its (R+2) x (W+2) input patch is DMA-ed into LDS once and
// the 9 taps run out of it as shifted fragment reads, instead of gathering the A tile 9 times from L2.  K order is
// (channel block, tap); weight tiles [BN][64] stream through a BSTAGES ring, the patch is double-buffered (the next
// block's pieces are issued one per tap).  Per 64 channels a 256x160 tile takes in 50 + 9x20 KiB instead of 9 x 52 KiB:
// half the operand intake per FLOP and a ninth of the activation traffic.  Patch swizzle: chunk ^= (patch row & 7),
// conflict-free for every tap shift; weight tiles keep the (row>>1)&7 swizzle of the ring kernel.
// Every loop event issues the same number of DMA instructions per wave (weight tile + one patch piece; surplus pieces
// re-issue piece 0 = identical bytes, or read out of bounds = zeros into the finished buffer), so the counted
// s_waitcnt is a constant.
constexpr int halo_smem_bytes(int NJ, int BSTAGES, int PMAX) { return 2 * PMAX * 128 + BSTAGES * 32 * NJ * 128; }

// NTAPS = 4: one 2x2 PHASE of a nearest-x2 upsample + 3x3 conv (conv_mode 5..12).  An output pixel (2y + a, 2x + b') of the
// upsampled conv reads a 2x2 neighbourhood of the LOW-resolution image with weights that are sums of the 3x3 taps falling on
// the same source pixel: 16 instead of 36 multiply-accumulates per low-resolution pixel.  In this kernel a phase is the 3x3
// conv restricted to the taps (a + dy, b' + dx), dy, dx in {0, 1}, of the SAME halo patch; modes 5..8 (forward) store row
// (b, y, x) at pixel (b, 2y + a, 2x + b') of the 2H x 2W output, modes 9..12 (input gradient) READ pixel (b, 2y + a, 2x + b')
// of the 2H x 2W gradient as pixel (b, y, x) and use the taps of phase (1 - a, 1 - b').  Two-taps-per-barrier path only.
template <int BM, int NJ, int BSTAGES, int PMAX, int NTAPS = 9>
__device__ __forceinline__ void conv_halo_body(const pdmk_gemm_args& g, unsigned a_bytes, unsigned b_bytes, int tw, const LC wgc,
                                               unsigned char* smem) {
    typedef Mma<bf16> MM;
    constexpr int BN = 32 * NJ, IM = BM / 64;
    constexpr int P_BYTES = PMAX * 128, B_BYTES = BN * 128;
    constexpr int NBLK_B = BN / 8, NB = (NBLK_B + 7) / 8;
    constexpr int NPW = (PMAX / 8 + 7) / 8;                          // patch pieces per wave (max)
    static_assert(PMAX % 8 == 0 && (NTAPS == 4 || NPW <= 9 - (BSTAGES - 1)), "whole pieces; next block's patch issued before its first weight tile");
    static_assert(NTAPS == 9 || (NTAPS == 4 && BSTAGES >= 4 && PDMK_HALO_PAIRS), "phase convs run the two-taps-per-barrier path");
    // phase geometry (NTAPS == 4): tap phase (ta, tb); strided input (in_s = 2, offsets ia, ib) for the gradient modes
    // conv_mode 13: the input gradient through ALL four phases as one problem - K = (phase, tap, channel), the channel-block
    // loop runs over 4 x ncb virtual blocks, each with the patch (pixel offset) and the taps of its phase
    const int phm = NTAPS == 4 ? g.conv_mode - 5 : 0;
    const bool merged = NTAPS == 4 && phm == 8;
    const bool dgr = NTAPS == 4 && phm >= 4;
    const int pa_ = merged ? 0 : (phm & 3) >> 1, pb_ = merged ? 0 : phm & 1;
    const int ta0 = dgr ? 1 - pa_ : pa_, tb0 = dgr ? 1 - pb_ : pb_;
    static_assert(2 * P_BYTES + BSTAGES * B_BYTES <= 160 * 1024, "LDS");
    static_assert(64 * (BN + 4) * 4 <= 2 * P_BYTES, "epilogue staging fits the patch buffers");
    static_assert(2 * P_BYTES + BSTAGES * B_BYTES == halo_smem_bytes(NJ, BSTAGES, PMAX), "smem size of the kernel wrappers");
    unsigned char* const bring = smem + 2 * P_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: wave-dependent branches and counts stay in SGPRs
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (g.N + BN - 1) / BN;
    const int tile = xcd_remap(wgc.bx, wgc.gx);
    // tile order: the workgroups of one XCD (consecutive logical tiles) share the operand panel that is worth more in its
    // L2 - the A rows (m-major) normally, the weight rows (n-major) for the 8x8-latent forward / dgrad layers, which stream
    // 15-60 MB of weights per launch for 512 output rows (n-major for every N > M measured 3 % slower for the step)
    const int ntm = (g.M + BM - 1) / BM;
    const bool nmajor = g.a_mode != PDMK_A_COLK && g.M <= 1024 && g.N > g.M;
    const int m0 = (nmajor ? tile % ntm : tile / ntn) * BM, n0 = (nmajor ? tile / ntm : tile % ntn) * BN;
    const int H = g.conv_hi, W = g.conv_wi, Ci = g.conv_ci, HW = H * W;
    const int ncb1 = (Ci + 63) / 64;
    const int ncb_total = (NTAPS == 4 && g.conv_mode == 13) ? 4 * ncb1 : ncb1;
    const int per = (ncb_total + wgc.gy - 1) / wgc.gy;
    const int cb0 = wgc.by * per, cb1 = min(ncb_total, cb0 + per);
    if (cb0 >= cb1 && g.accumulate != 2) return;        // slab split-K: an empty split still writes its (zero) slab

    // tile geometry: rimg rows x tw columns of one image (HW >= BM; tw = W when whole rows fit) or BM/HW whole images
    const int rimg = HW >= BM ? BM / tw : H;                          // rows of an image inside the tile
    const int W2 = tw + 2, pimg = (rimg + 2) * W2;                    // patch rows per image
    const int nimg = HW >= BM ? 1 : BM / HW;
    const int prows = nimg * pimg;
    int img0, y0, x0 = 0;                                             // first image / first output row / column of the tile
    if (tw == W) {
        img0 = m0 / HW;
        y0 = (m0 - img0 * HW) / W;
    } else {
        const int tpr = W / tw, tpi = (H / rimg) * tpr, tm = m0 / BM;
        img0 = tm / tpi;
        const int r = tm - img0 * tpi;
        y0 = (r / tpr) * rimg;
        x0 = (r - (r / tpr) * tpr) * tw;
    }
    const int mbase = tw == W ? m0 : (img0 * H + y0) * W + x0;        // first output pixel of the tile

    const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), (short)0, (int)a_bytes, 0x00020000);
    const auto rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.B), (short)0, (int)b_bytes, 0x00020000);

    // ---- patch loader: piece j of this wave = patch rows 8*(j*8+wave) .. +7; lane -> (row sr = lane>>3, physical chunk lane&7)
    const int sr = lane >> 3;
    const int lcp = (lane & 7) ^ sr;                                  // logical chunk (8 channels) of this lane, patch
    unsigned poff[NPW];                                               // byte offset of the source pixel, or OOB
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
        const int prow = (j * 8 + wave) * 8 + sr;
        poff[j] = OOB;
        if (prow < prows) {
            const int il = prow / pimg, rem = prow - il * pimg;
            const int py = rem / W2, px = rem - py * W2;
            const int b = img0 + il, y = y0 + py - 1, x = x0 + px - 1;
            if (b < g.conv_b && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
                poff[j] = (dgr ? (unsigned)((b * 2 * H + 2 * y + pa_) * (2 * W) + 2 * x + pb_)
                               : (unsigned)((b * H + y) * W + x)) * (unsigned)g.conv_ld * 2u;
        }
    }
    // ---- weight loader (as in the ring kernel)
    const int lcb = (lane & 7) ^ (((wave & 1) * 4) + (sr >> 1));
    unsigned b_base[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int n = n0 + (i * 8 + wave) * 8 + sr;
        b_base[i] = (n < g.N) ? (unsigned)n * (unsigned)g.ldb * 2u : OOB;
    }
    const int nb_wave = NBLK_B / 8 + (wave < (NBLK_B % 8) ? 1 : 0);

    // virtual block vcb -> (phase ph, channel block): ph = 0 and the block itself except in the merged gradient mode
    auto blk_ph = [&](int vcb) { return merged ? (vcb >= 2 * ncb1 ? (vcb >= 3 * ncb1 ? 3 : 2) : (vcb >= ncb1 ? 1 : 0)) : 0; };
    auto issue_piece = [&](int vcb, unsigned off, int jj) {           // piece jj of patch(vcb) from source offset `off`
        const int ph = blk_ph(vcb), cb = vcb - ph * ncb1;
        const int ch = cb * 64 + lcp * 8;
        const unsigned pho = merged ? (unsigned)((ph >> 1) * 2 * W + (ph & 1)) * (unsigned)g.conv_ld * 2u : 0u;
        const unsigned va = (off != OOB && ch < Ci && vcb < cb1) ? off + pho + (unsigned)ch * 2u : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void*)(smem + (vcb & 1) * P_BYTES + (jj * 8 + wave) * 1024), 16,
                                                 (int)va, 0, 0, 0);
    };
    auto issue_b = [&](int vcb, int tap, int slot) {
        unsigned char* sb = bring + slot * B_BYTES;
        const int ph = blk_ph(vcb), cb = vcb - ph * ncb1;
        const int ch = cb * 64 + lcb * 8;
        const unsigned koff = (unsigned)((ph * 4 + tap) * Ci + ch) * 2u;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (i * 8 + wave < NBLK_B) {
                const unsigned vb = (b_base[i] != OOB && ch < Ci && vcb < cb1) ? b_base[i] + koff : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_void*)(sb + (i * 8 + wave) * 1024), 16, (int)vb, 0, 0, 0);
            }
        }
    };

    f32x4 acc[IM][NJ];
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- fragment addressing.  A: patch row of output pixel (wm*16*IM + i*16 + fr) for tap (0,0); B: as the ring kernel
    const int fr = lane & 15, fg = lane >> 4;
    int prow0[IM];
#pragma unroll
    for (int i = 0; i < IM; ++i) {
        const int ml = wm * (16 * IM) + i * 16 + fr;                  // output pixel inside the tile
        const int il = ml / (rimg * tw), rem = ml - il * (rimg * tw);
        const int yl = rem / tw, x = rem - yl * tw;
        prow0[i] = il * pimg + yl * W2 + x;
    }
    const int fsw = (fr >> 1) & 7;
    const unsigned fch0 = (unsigned)((fg ^ fsw) * 16), fch1 = (unsigned)(((4 + fg) ^ fsw) * 16);
    const unsigned b_row = (unsigned)(wn * (16 * NJ) + fr) * 128u;

    // one tap: 2 x (IM x NJ) MFMAs out of the patch (shifted rows) and one weight tile
    int ta = ta0, tb = tb0;                                          // tap phase of the block being multiplied
    auto tap_compute = [&](const unsigned char* pbuf, const unsigned char* sb, int tap, auto&& mid) __attribute__((always_inline)) {
        const int toff = NTAPS == 4 ? (ta + (tap >> 1)) * W2 + (tb + (tap & 1)) : (tap / 3) * W2 + (tap % 3);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (kk == 1) mid();                      // (an issue point between the tap's two MFMA groups: conv_halo_body's event loop)
            bf16x8 af[IM], bf[NJ];
#pragma unroll
            for (int i = 0; i < IM; ++i) {
                const int prow = prow0[i] + toff;
                af[i] = *reinterpret_cast<const bf16x8*>(pbuf + prow * 128 + (((fg + 4 * kk) ^ (prow & 7)) * 16));
            }
            const unsigned fch = kk ? fch1 : fch0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(sb + b_row + fch + j * 2048);
            PDMK_PRIO_UP();
#pragma unroll
            for (int i = 0; i < IM; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = MM::mma(bf[j], af[i], acc[i][j]);
            PDMK_PRIO_DOWN();
        }
        __builtin_amdgcn_sched_barrier(0);            // no cross-tap code motion: keeps fragment live ranges to one tap
    };

    const int nsteps = cb1 > cb0 ? (cb1 - cb0) * NTAPS : 0;
    prio_static(wave);
    if (BSTAGES >= 4 && PDMK_HALO_PAIRS) {
        // ---- two taps per barrier (rings of >= 4 slots: the 128-row tiles).  A K-step of an 8-wave workgroup has ~480 cycles
        // of fixed cost (wait + barrier rendezvous + restart of the MFMA stream) next to 640 cycles of MFMA work
        // per SIMD and tap for these tiles; events (0,1) (2,3) (4,5) (6,7) (8) of a channel block halve the number of barriers.
        // An event waits for EVERYTHING in flight (its two weight tiles and the patch pieces issued with them, one event ago:
        // plain vmcnt(0), no counting), then issues the next event's tiles into the slots behind its own and two pieces of the
        // next block's patch; nothing in the loop depends on where in the sequence it is (tiles / pieces past the end read out
        // of range), so the body has no branches.
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const bool own = (j * 8 + wave) * 8 < prows;              // surplus pieces re-issue piece 0 (identical bytes, inside the buffer)
            issue_piece(nsteps > 0 ? cb0 : cb1, own ? poff[j] : poff[0], own ? j : 0);
        }
        issue_b(nsteps > 0 ? cb0 : cb1, 0, 0);
        issue_b(nsteps > 0 ? cb0 : cb1, 1, 1);
        int slot_c = 0;
        constexpr int NE = (NTAPS + 1) / 2;                       // events per channel block: (0,1)(2,3)(4,5)(6,7)(8) / (0,1)(2,3)
        constexpr int PPT = NTAPS == 9 ? 1 : (NPW + NTAPS - 1) / NTAPS;
        for (int cb = cb0; cb < cb1; ++cb) {
            const unsigned char* pbuf = smem + (cb & 1) * P_BYTES;
            if (merged) {                                             // gradient through phase ph uses the taps of phase (1 - a, 1 - b')
                const int ph = blk_ph(cb);
                ta = 1 - (ph >> 1);
                tb = 1 - (ph & 1);
            }
#pragma unroll
            for (int i = 0; i < IM; ++i) asm volatile("" : "+v"(prow0[i]));
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int t0 = 2 * e, ne = (t0 + 1 < NTAPS) ? 2 : 1;      // this event's taps
                const bool last = e == NE - 1;
                const int nn = last ? (NTAPS >= 2 ? 2 : 1) : ((t0 + 3 < NTAPS) ? 2 : 1);   // taps of the NEXT event
                const int cbn = last ? cb + 1 : cb, tn0 = last ? 0 : t0 + 2;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                // The event's DMA instructions (next event's weight tiles, pieces of the next block's patch: 6-8 per wave) go into
                // slots nobody reads during this event, so they can be issued anywhere inside it.  All eight waves issuing them
                // together right behind the barrier hold each other in the memory pipe (~140 cycles per piece) while
                // the matrix pipes idle: waves 0-3 issue first, their SIMD partners 4-7 multiply the first tap first and issue between
                // the taps (PDMK_HALO_SPLIT_ISSUE=0: all waves first)
                const int slot_c0 = slot_c;
                auto issue_all = [&]() __attribute__((always_inline)) {
                    int sl = slot_c0 + ne;
                    sl = sl >= BSTAGES ? sl - BSTAGES : sl;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        if (u < nn) {
                            issue_b(cbn, tn0 + u, sl);
                            sl = sl + 1 == BSTAGES ? 0 : sl + 1;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        if (u < ne) {
#pragma unroll
                            for (int q = 0; q < PPT; ++q) {               // PPT patch pieces of the next block per tap
                                const int t = (t0 + u) * PPT + q;
                                const bool own = t < NPW && (t * 8 + wave) * 8 < prows;
                                issue_piece(cb + 1, own ? poff[t < NPW ? t : 0] : poff[0], own ? t : 0);
                            }
                        }
                    }
                };
                // (measured, tools/conv_ab.py, B = 8: the 128 x 160 tiles gain 2.5-12 % - 64^2 320 -> 320 79.1 -> 70.5 us, 32^2 640 -> 640
                // 62.6 -> 59.4 - the 128 x 128 tiles LOSE 1-9 % - 960 -> 320 210.7 -> 229.7 us: with 16 MFMAs per tap and two pieces per
                // weight tile their partners finish the first tap before the early half is out of the memory pipe - so NJ = 5 only)
                // issue point of this wave: 0 = before the first tap, 2 = between the taps (PDMK_HALO_SPLIT_ISSUE >= 1, by wave half);
                // PDMK_HALO_SPLIT_ISSUE == 2 adds 1 / 3 = between the two MFMA groups of the first / second tap (waves 2, 3 / 6, 7)
                const int ip = (PDMK_HALO_SPLIT_ISSUE && NJ >= 5 && ne == 2)
                                   ? ((wave >= 4 ? 2 : 0) + ((PDMK_HALO_SPLIT_ISSUE == 2 && (wave & 2)) ? 1 : 0)) : 0;   // wave-uniform
                auto nomid = [&]() __attribute__((always_inline)) {};
                if (ip == 0) issue_all();
                if (PDMK_HALO_SPLIT_ISSUE == 2) tap_compute(pbuf, bring + slot_c * B_BYTES, t0, [&]() __attribute__((always_inline)) { if (ip == 1) issue_all(); });
                else tap_compute(pbuf, bring + slot_c * B_BYTES, t0, nomid);
                slot_c = slot_c + 1 == BSTAGES ? 0 : slot_c + 1;
                if (ip == 2) issue_all();
                if (ne == 2) {
                    if (PDMK_HALO_SPLIT_ISSUE == 2) tap_compute(pbuf, bring + slot_c * B_BYTES, t0 + 1, [&]() __attribute__((always_inline)) { if (ip == 3) issue_all(); });
                    else tap_compute(pbuf, bring + slot_c * B_BYTES, t0 + 1, nomid);
                    slot_c = slot_c + 1 == BSTAGES ? 0 : slot_c + 1;
                }
            }
        }
    } else {
        // ---- prologue: the first patch, then BSTAGES-1 weight tiles
    #pragma unroll
        for (int j = 0; j < NPW; ++j)
            if (nsteps > 0 && (j * 8 + wave) * 8 < prows) issue_piece(cb0, poff[j], j);
    #pragma unroll
        for (int s = 0; s < BSTAGES - 1; ++s)
            if (s < nsteps) issue_b(cb0 + s / 9, s % 9, s);

        const int n_ss = (BSTAGES - 2) * (nb_wave + 1) + 1;               // steady-state count of younger DMAs
        constexpr int N_SS_HI = (BSTAGES - 2) * (NB + 1) + 1;             // ... as immediates: waves with NB weight pieces per tile
        constexpr int N_SS_LO = (BSTAGES - 2) * NB + 1;                   //     and (NBLK_B % 8 != 0) waves with NB - 1
        int s = 0, slot_c = 0, slot_i = BSTAGES - 1;                      // step; ring slots of the tile read / issued next
        for (int cb = cb0; cb < cb1; ++cb) {
            const unsigned char* pbuf = smem + (cb & 1) * P_BYTES;
            // keep the 9 x IM x 2 fragment addresses from being hoisted out of this loop as invariants (72 VGPRs, spills)
    #pragma unroll
            for (int i = 0; i < IM; ++i) asm volatile("" : "+v"(prow0[i]));
            // interior channel blocks: every tap has a full ring behind it and a tile left to issue, so the event needs no
            // run-time case analysis - an immediate s_waitcnt, tap / block of the next tile known at compile time, the patch
            // piece chosen by a select.  (A scalar compare + branch costs ~40 cycles; the general path below has ~10 per tap,
            // against 640 cycles of MFMA work per SIMD and tap for the 128-row tiles.)
            const bool steady = cb > cb0 && cb + 1 < cb1;
    #pragma unroll
            for (int tap = 0; tap < 9; ++tap, ++s) {
                if (steady) {
                    if ((NBLK_B % 8) == 0 || wave < (NBLK_B % 8)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_SS_HI) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_SS_LO) : "memory");
                    __builtin_amdgcn_s_barrier();
                    constexpr int ahead = BSTAGES - 1;
                    issue_b(cb + ((tap + ahead) >= 9 ? 1 : 0), (tap + ahead) % 9, slot_i);
                    const bool own = tap < NPW && (tap * 8 + wave) * 8 < prows;
                    issue_piece(cb + 1, own ? poff[tap < NPW ? tap : 0] : poff[0], own ? tap : 0);
                    slot_i = slot_i + 1 == BSTAGES ? 0 : slot_i + 1;
                } else {
                    const bool tail = s + BSTAGES - 1 >= nsteps;
                    if (tail) wait_vmcnt_dyn(0);
                    else if (s <= BSTAGES - 2) wait_vmcnt_dyn((BSTAGES - 2 - s) * nb_wave + s * (nb_wave + 1));
                    else wait_vmcnt_dyn(n_ss);
                    __builtin_amdgcn_s_barrier();
                    if (!tail) {
                        const int sn = s + BSTAGES - 1;                   // weight tile to issue: step sn = (cbn, tapn)
                        const int cbn = cb0 + sn / 9, tapn = sn - (sn / 9) * 9;
                        issue_b(cbn, tapn, slot_i);
                        slot_i = slot_i + 1 == BSTAGES ? 0 : slot_i + 1;
                        // next block's patch, one piece per tap (surplus taps re-issue piece 0: identical bytes)
                        if (tap < NPW && (tap * 8 + wave) * 8 < prows) issue_piece(cb + 1, poff[tap < NPW ? tap : 0], tap);
                        else issue_piece(cb + 1, poff[0], 0);
                    }
                }
                tap_compute(pbuf, bring + slot_c * B_BYTES, tap, []() {});
                slot_c = slot_c + 1 == BSTAGES ? 0 : slot_c + 1;
            }
        }
    }
    // the last events of the paired / steady paths issue weight-tile and patch DMAs that nobody waits for (all out of range:
    // zeros into LDS); the epilogue reuses that LDS as its staging image, so drain them explicitly - a late zero write
    // must not land on a staged tile, and what __syncthreads() happens to emit is not a contract
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (PDMK_PRIO == 1) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    if (NTAPS == 4 && !dgr) {
        if (tw == W) ring_epilogue<BM, NJ, 2 * P_BYTES + BSTAGES * B_BYTES, false, NTAPS == 4>(g, wgc, acc, smem, m0, n0);
        else ring_epilogue<BM, NJ, 2 * P_BYTES + BSTAGES * B_BYTES, true, NTAPS == 4>(g, wgc, acc, smem, mbase, n0, tw, W);
        return;
    }
    if (tw == W) ring_epilogue<BM, NJ, 2 * P_BYTES + BSTAGES * B_BYTES>(g, wgc, acc, smem, m0, n0);
    else ring_epilogue<BM, NJ, 2 * P_BYTES + BSTAGES * B_BYTES, true>(g, wgc, acc, smem, mbase, n0, tw, W);
}

template <int BM, int NJ, int BSTAGES, int PMAX, int NTAPS = 9>
__global__ __launch_bounds__(NT, 2) void conv_halo_kernel(pdmk_gemm_args g, unsigned a_bytes, unsigned b_bytes, int tw) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[halo_smem_bytes(NJ, BSTAGES, PMAX)];
    conv_halo_body<BM, NJ, BSTAGES, PMAX, NTAPS>(g, a_bytes, b_bytes, tw, lc_plain(), smem);
}
template <int BM, int NJ, int BSTAGES, int PMAX, int NTAPS = 9>
__global__ __launch_bounds__(NT, 2) void conv_halo_group_kernel(pdmk_gemm_group_dev gg) {      // aux0 = tile width of the problem
    __shared__ __attribute__((aligned(1024))) unsigned char smem[halo_smem_bytes(NJ, BSTAGES, PMAX)];
    int pi;
    LC wgc;
    if (!lc_group(gg, pi, wgc)) return;
    conv_halo_body<BM, NJ, BSTAGES, PMAX, NTAPS>(gg.p[pi], gg.a_bytes[pi], gg.b_bytes[pi], gg.aux0[pi], wgc, smem);
}

// ------------------------------------------------------------------------------------------------------------------
// Weight-gradient GEMM: C[m][n] += sum_k A(k,m) B(k,n) with BOTH operands reduction-major (k = pixel rows):
// A = dY [P][M], B = X [P][N] (Linear) or the 3x3 gather of X (conv: column n = tap*Ci + c reads pixel src(p, tap)).
// Same ring as above with [64 k][128 col] tiles (256-byte rows, a DMA piece = 4 k-rows); the MFMA operands are read
// with ds_read_b64_tr_b16.  Swizzle: 16-byte chunk index ^= 2*f(k), f(k) = (k&3) | ((k>>3)&1)<<2, applied to the source
// column of each DMA lane and to the transposing reads: the 8 k-rows one 32-lane read group touches land in 8 different
// 32-byte bank groups.  The bias gradient (column sums of dY) is one extra MFMA per A fragment against a ones vector, in
// the workgroups of the first n-tile only.
constexpr int wgrad_smem_bytes(int BM, int NJ, int STAGES) { return STAGES * (64 * BM * 2 + 64 * 32 * NJ * 2); }

template <bool CONV, int BM, int NJ, int STAGES, int OCC>
__device__ __forceinline__ void wgrad_ring_body(const pdmk_gemm_args& g, int lg_wo, int lg_howo, unsigned a_bytes, unsigned b_bytes,
                                                const LC wgc, unsigned char* smem) {
    typedef Mma<bf16> MM;
    constexpr int BN = 32 * NJ, IM = BM / 64;
    constexpr int RA = BM * 2, RB = BN * 2;                     // row bytes of the [64 k][BM] / [64 k][BN] tiles: 128 or 256
    constexpr int A_BYTES = 64 * RA, B_BYTES = 64 * RB, SLOT = A_BYTES + B_BYTES;
    constexpr int PA = A_BYTES / 1024 / 8, PB = B_BYTES / 1024 / 8;   // DMA pieces per wave and stage (1 KiB each)
    static_assert((BM == 64 || BM == 128) && (BN == 64 || BN == 128), "tile");
    static_assert(STAGES >= 2 && STAGES * SLOT * (OCC / 2) <= 160 * 1024, "ring(s) must fit the 160 KiB LDS");
    static_assert(64 * (BN + 4) * 4 <= STAGES * SLOT, "epilogue staging");
    static_assert(STAGES * SLOT == wgrad_smem_bytes(BM, NJ, STAGES), "smem size of the kernel wrappers");

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: wave-dependent branches and counts stay in SGPRs
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (g.N + BN - 1) / BN;
    const int tile = xcd_remap(wgc.bx, wgc.gx);
    // tile order: the workgroups of one XCD (consecutive logical tiles) share the operand panel that is worth more in its
    // L2 - the A rows (m-major) normally, the weight rows (n-major) for the 8x8-latent forward / dgrad layers, which stream
    // 15-60 MB of weights per launch for 512 output rows (n-major for every N > M measured 3 % slower for the step)
    const int ntm = (g.M + BM - 1) / BM;
    const bool nmajor = g.a_mode != PDMK_A_COLK && g.M <= 1024 && g.N > g.M;
    const int m0 = (nmajor ? tile % ntm : tile / ntn) * BM, n0 = (nmajor ? tile / ntm : tile % ntn) * BN;
    const int nk_total = (g.K + BK - 1) / BK;
    const int per = (nk_total + wgc.gy - 1) / wgc.gy;
    const int kt0 = wgc.by * per;
    const int kt1 = min(nk_total, kt0 + per);
    if (kt0 >= kt1 && g.accumulate != 2) return;        // slab split-K: an empty split still writes its (zero) slab

    const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), (short)0, (int)a_bytes, 0x00020000);
    const auto rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.B), (short)0, (int)b_bytes, 0x00020000);
    const ConvGeom cg{g.conv_hi, g.conv_wi, g.conv_ci, g.conv_ho, g.conv_wo, g.conv_ld, g.conv_mode};

    // ---- loaders.  256-byte rows: a piece = 4 k-rows, lane -> (row lane>>4, chunk lane&15), swizzle 2*((k&3) | ((k>>3)&1)<<2).
    //               128-byte rows: a piece = 8 k-rows, lane -> (row lane>>3, chunk lane&7),  swizzle 2*(((k>>1)&1) | ((k>>3)&1)<<1).
    // Piece index = i*8 + wave, so the swizzle of every row a lane loads is the same (depends on lane and wave only).
    auto lane_row = [&](int rbytes) { return rbytes == 256 ? (lane >> 4) : (lane >> 3); };
    auto lane_lc = [&](int rbytes) {
        if (rbytes == 256) return (lane & 15) ^ (((lane >> 4) | (((wave >> 1) & 1) << 2)) << 1);
        return (lane & 7) ^ (((((lane >> 3) >> 1) & 1) | ((wave & 1) << 1)) << 1);
    };
    const int a_row = lane_row(RA), b_rowl = lane_row(RB);
    const int acol = m0 + lane_lc(RA) * 8, bcol = n0 + lane_lc(RB) * 8;
    const bool a_ok = acol < g.M, b_ok = bcol < g.N;
    int b_tap = 0, b_ci = 0;
    if (CONV) {
        b_tap = bcol / cg.ci;
        b_ci = bcol - b_tap * cg.ci;
    }

    auto issue = [&](int kt, int slot) {
        unsigned char* sa = smem + slot * SLOT;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int piece = i * 8 + wave;
            const int kr = kt * BK + piece * (1024 / RA) + a_row;    // pixel row
            unsigned krs = (unsigned)kr;
            if (CONV && cg.mode >= 5 && kr < g.K) {                  // phase (a, b') of an upsampled conv: dY row of pixel (b, 2y + a, 2x + b')
                const int hw = cg.ho * cg.wo, b = kr / hw, rem = kr - b * hw, oy = rem / cg.wo, ox = rem - oy * cg.wo;
                const int ph = cg.mode - 5;
                krs = (unsigned)((b * 2 * cg.ho + 2 * oy + (ph >> 1)) * (2 * cg.wo) + 2 * ox + (ph & 1));
            }
            const unsigned va = (kr < g.K && a_ok) ? (krs * (unsigned)g.lda + (unsigned)acol) * 2u : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void*)(sa + piece * 1024), 16, (int)va, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int piece = i * 8 + wave;
            const int kr = kt * BK + piece * (1024 / RB) + b_rowl;
            const bool kok = kr < g.K;
            unsigned vb = OOB;
            if (CONV) {
                if (kok && b_ok) {
                    int b, oy, ox;
                    if (lg_wo >= 0) {                                // power-of-two image: shifts instead of divisions
                        b = kr >> lg_howo;
                        const int rem = kr & ((1 << lg_howo) - 1);
                        oy = rem >> lg_wo;
                        ox = rem & ((1 << lg_wo) - 1);
                    } else {
                        const int hw = cg.ho * cg.wo;
                        b = kr / hw;
                        const int rem = kr - b * hw;
                        oy = rem / cg.wo;
                        ox = rem - oy * cg.wo;
                    }
                    const int px = conv_src_pixel(cg, b, oy, ox, b_tap);
                    if (px >= 0) vb = ((unsigned)px * (unsigned)cg.ld + (unsigned)b_ci) * 2u;
                }
            } else if (kok && b_ok) {
                vb = ((unsigned)kr * (unsigned)g.ldb + (unsigned)bcol) * 2u;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_void*)(sa + A_BYTES + piece * 1024), 16, (int)vb, 0, 0, 0);
        }
    };

    f32x4 acc[IM][NJ], acs[IM];
#pragma unroll
    for (int i = 0; i < IM; ++i) {
        acs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool do_colsum = g.colsum_out != nullptr && n0 == 0 && wn == 0;      // wave-uniform
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

    // transposing fragment reads: lane (g = lane>>4, q = (lane&15)>>2, p = lane&3) reads 8 bytes of k-row kk + 8g + q (and of
    // row + 4) at columns col0 + 4p; physical chunk = ((col0>>3) ^ 2 f(k)) | (p>>1)
    const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int f256 = (tq | ((tg & 1) << 2)) << 1, f128 = (((tq >> 1) & 1) | ((tg & 1) << 1)) << 1;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    auto tr_frag = [&](const unsigned char* base, int rbytes, int kk, int col0) -> bf16x8 {
        const unsigned ch = (unsigned)(((col0 >> 3) ^ (rbytes == 256 ? f256 : f128)) | (tp >> 1));
        const unsigned char* a0 = base + (unsigned)(kk + 8 * tg + tq) * (unsigned)rbytes + ch * 16u + (unsigned)(tp & 1) * 8u;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * rbytes));
        s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    };

#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (kt0 + s < kt1) issue(kt0 + s, s);

    int slot = 0;
    prio_static(wave);
    for (int kt = kt0; kt < kt1; ++kt) {
        if (STAGES == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (kt + STAGES - 1 <= kt1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * (PA + PB)) : "memory");
        else wait_vmcnt_dyn(min(STAGES - 2, kt1 - 1 - kt) * (PA + PB));
        __builtin_amdgcn_s_barrier();
        if (kt + STAGES - 1 < kt1) issue(kt + STAGES - 1, slot == 0 ? STAGES - 1 : slot - 1);
        const unsigned char* sa = smem + slot * SLOT;
        const unsigned char* sb = sa + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 32) {
            bf16x8 af[IM], bf[NJ];
#pragma unroll
            for (int i = 0; i < IM; ++i) af[i] = tr_frag(sa, RA, kk, wm * (16 * IM) + i * 16);
#pragma unroll
            for (int j = 0; j < NJ; ++j) bf[j] = tr_frag(sb, RB, kk, wn * (16 * NJ) + j * 16);
            PDMK_PRIO_UP();
#pragma unroll
            for (int i = 0; i < IM; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = MM::mma(bf[j], af[i], acc[i][j]);
            PDMK_PRIO_DOWN();
            if (do_colsum) {
#pragma unroll
                for (int i = 0; i < IM; ++i) acs[i] = MM::mma(ones, af[i], acs[i]);
            }
        }
        slot = slot + 1 == STAGES ? 0 : slot + 1;
    }
    if (PDMK_PRIO == 1) __builtin_amdgcn_s_setprio(0);
    if (do_colsum && (lane >> 4) == 0) {
#pragma unroll
        for (int i = 0; i < IM; ++i) {
            const int m = m0 + wm * (16 * IM) + i * 16 + (lane & 15);
            if (m < g.M) unsafeAtomicAdd(g.colsum_out + m, acs[i][0]);
        }
    }
    ring_epilogue<BM, NJ, STAGES * SLOT>(g, wgc, acc, smem, m0, n0);
}

template <bool CONV, int BM, int NJ, int STAGES, int OCC>
__global__ __launch_bounds__(NT, OCC) void wgrad_ring_kernel(pdmk_gemm_args g, int lg_wo, int lg_howo, unsigned a_bytes,
                                                             unsigned b_bytes) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[wgrad_smem_bytes(BM, NJ, STAGES)];
    wgrad_ring_body<CONV, BM, NJ, STAGES, OCC>(g, lg_wo, lg_howo, a_bytes, b_bytes, PDMK_WGRAD_XCD ? lc_split_xcd() : lc_plain(), smem);
}
template <bool CONV, int BM, int NJ, int STAGES, int OCC>
__global__ __launch_bounds__(NT, OCC) void wgrad_ring_group_kernel(pdmk_gemm_group_dev gg) {   // aux0 / aux1 = lg_wo / lg_howo
    __shared__ __attribute__((aligned(1024))) unsigned char smem[wgrad_smem_bytes(BM, NJ, STAGES)];
    int pi;
    LC wgc;
    if (!lc_group(gg, pi, wgc)) return;
    wgrad_ring_body<CONV, BM, NJ, STAGES, OCC>(gg.p[pi], gg.aux0[pi], gg.aux1[pi], gg.a_bytes[pi], gg.b_bytes[pi], wgc, smem);
}

struct Config {
    int bm, nj, stages, occ;      // occ 4 = "shallow": short ring, two workgroups per CU (many-round grids, few K-steps)
};
// candidate table: the autotuner in gemm.hip times these per GEMM shape; ids are stable (plan cache values)
static const Config kConfigs[] = {
    {256, 4, 3, 2}, {256, 5, 3, 2}, {128, 4, 4, 2}, {128, 5, 4, 2}, {64, 4, 6, 2}, {64, 5, 5, 2},
    {128, 4, 2, 4}, {64, 4, 3, 4},  {64, 5, 2, 4},  {128, 6, 3, 2}, {64, 6, 4, 2},  {128, 5, 2, 4},
    // small tiles for the small / short GEMMs of the 8x8 .. 32x32 levels: a workgroup's K-loop is bound by the ~70 GB/s a
    // CU takes in from L2, so a GEMM with fewer tiles than CUs runs at (tiles / 256) of the chip's intake - more, smaller
    // tiles (and split-K) spread it over the CUs
    {64, 2, 4, 4},  {128, 2, 3, 4}, {64, 3, 4, 4},
};
constexpr int kNumConfigs = sizeof(kConfigs) / sizeof(kConfigs[0]);
// candidate numbering (stable: plan files and tests name candidates by id): ring ids 0 .. kNumBase-1 = the first kNumBase
// ring shapes, kNumBase .. kNumBase+3 = the four halo-conv shapes, then the ring shapes added later
constexpr int kNumBase = 12;

// Halo-conv shapes (candidate ids kNumConfigs + h): tile rows, channel tiles, patch capacity; eligibility of a conv.
// Tile width for halo shape h on this conv: the image width when whole rows fit the patch buffer, else the widest
// power-of-two column block (images wider than a tile - the VAE encoder's 128^2..512^2 levels - are cut into
// rows x tw blocks, e.g. 16 x 16 output pixels + halo = 324 patch rows); 0 = not eligible.
static int halo_tile_w(const pdmk_gemm_args& g, int h, int splitk) {
    const bool phase = g.conv_mode >= 5 && g.conv_mode <= 13;        // 2x2 phase(s) of an upsampling conv: the 128-row shapes only
    if (g.a_mode != PDMK_A_CONV || !(g.conv_mode == 0 || phase) || g.conv_ho != g.conv_hi || g.conv_wo != g.conv_wi) return 0;
    if (phase && (h < 2 || !PDMK_HALO_PAIRS || g.R || (splitk > 1))) return 0;
    // (a phase's weights may be a column slice of the [N][16 ci] matrix of all four: ldb is then the full row)
    if ((g.conv_ci % 8) || g.conv_wi < 4) return 0;
    if (phase ? ((g.ldb % 8) || g.ldb < (g.conv_mode == 13 ? 16 : 4) * g.conv_ci) : g.ldb != 9 * g.conv_ci) return 0;
    if ((splitk > 1 ? splitk : 1) > (g.conv_ci + 63) / 64) return 0;
    const int bm = h < 2 ? 256 : 128, pmax = h < 2 ? 400 : 264;
    const int H = g.conv_hi, W = g.conv_wi, HW = H * W;
    if (HW < bm) return (bm % HW) == 0 && (bm / HW) * (H + 2) * (W + 2) <= pmax ? W : 0;
    if (HW % bm) return 0;
    static const int pref = getenv("PDMK_HALO_TW") ? atoi(getenv("PDMK_HALO_TW")) : 0;   // experiment knob: preferred block width
    if (pref > 0 && pref < W && (W % pref) == 0 && (bm % pref) == 0 && (H % (bm / pref)) == 0 &&
        (bm / pref + 2) * (pref + 2) <= pmax)
        return pref;
    if ((bm % W) == 0 && (bm / W + 2) * (W + 2) <= pmax) return W;
    for (int tw = 128; tw >= 8; tw >>= 1) {
        if (tw >= W || (W % tw) || (bm % tw)) continue;
        const int rows = bm / tw;
        if ((H % rows) == 0 && (rows + 2) * (tw + 2) <= pmax) return tw;
    }
    return 0;
}
static bool halo_ok(const pdmk_gemm_args& g, int h, int splitk) { return halo_tile_w(g, h, splitk) > 0; }

// Untuned default (graph capture of a shape never seen eagerly, PDMK_GEMM_TUNE=0): the decision tree the tuned plans of
// the SD-2.1 step condense to (tools/ring_sweep.py; plan files of bench.py).  Stride-1 convs: the halo kernel, 256-row
// tiles once there are enough of them; Linear and the strided convs: the two-workgroups-per-CU shallow rings, tile rows
// by M.  Returns the ring candidate index (halo shapes follow the ring shapes).
static int pick_config(const pdmk_gemm_args& g, int splitk) {
    enum { R256x128, R256x160, R128x128, R128x160, R64x128, R64x160, S128x128, S64x128, S64x160, R128x192, R64x192, S128x160 };
    const bool n160 = (g.N % 160) == 0 || (g.N > 256 && (g.N % 128) != 0);
    if (g.a_mode == PDMK_A_CONV && g.conv_mode >= 5) return kNumBase + (halo_ok(g, n160 ? 2 : 3, splitk) ? (n160 ? 2 : 3) : (n160 ? 3 : 2));
    if (g.a_mode == PDMK_A_CONV && g.conv_mode == 0) {
        const long t256 = (long)((g.M + 255) / 256) * ((g.N + 159) / 160);
        const int h = (g.M >= 8192 && t256 >= 128) ? (n160 ? 0 : 1) : (n160 ? 2 : 3);
        if (halo_ok(g, h, splitk)) return kNumBase + h;
        if (halo_ok(g, h | 2, splitk)) return kNumBase + (h | 2);
    }
    if (g.M >= 8192) return n160 ? S128x160 : S128x128;
    if (g.M >= 4096) return S64x160;
    if (g.M >= 2048 && n160 && g.N >= 1280) return S64x160;
    return S64x128;
}

}  // namespace pdmk_ring

constexpr int kNumHalo = 4;      // halo-conv candidates follow the ring shapes in the candidate numbering
// row-block Linear kernels (gemm_rowblock.hip): candidate ids after every ring / halo shape
int pdmk_gemm_rowblock_num_configs();
int pdmk_gemm_rowblock_name(int id, char* buf, int n);
int pdmk_gemm_rowblock_launch(const pdmk_gemm_args& g, hipStream_t st, long a_bytes, long b_bytes, int id, bool dry = false);
static int conv_halo_launch(const pdmk_gemm_args& g, hipStream_t st, long a_bytes, long b_bytes, int id) {
    using namespace pdmk_ring;
    const int tw = halo_tile_w(g, id, g.splitk);
    if (tw <= 0) return 1;
    const int bm = id < 2 ? 256 : 128, nj = (id & 1) ? 4 : 5;
    const int sk = g.splitk > 1 ? g.splitk : 1;
    const int bn = 32 * nj;
    dim3 grid(((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn), sk);
    if (g.conv_mode >= 5) {
        if (id == 2) hipLaunchKernelGGL((conv_halo_kernel<128, 5, 4, 264, 4>), grid, dim3(NT), 0, st, g, (unsigned)a_bytes, (unsigned)b_bytes, tw);
        else hipLaunchKernelGGL((conv_halo_kernel<128, 4, 5, 264, 4>), grid, dim3(NT), 0, st, g, (unsigned)a_bytes, (unsigned)b_bytes, tw);
        return hipGetLastError() == hipSuccess ? 0 : -1000;
    }
    switch (id) {
        case 0: hipLaunchKernelGGL((conv_halo_kernel<256, 5, 3, 400>), grid, dim3(NT), 0, st, g, (unsigned)a_bytes, (unsigned)b_bytes, tw); break;
        case 1: hipLaunchKernelGGL((conv_halo_kernel<256, 4, 3, 400>), grid, dim3(NT), 0, st, g, (unsigned)a_bytes, (unsigned)b_bytes, tw); break;
        case 2: hipLaunchKernelGGL((conv_halo_kernel<128, 5, 4, 264>), grid, dim3(NT), 0, st, g, (unsigned)a_bytes, (unsigned)b_bytes, tw); break;
        case 3: hipLaunchKernelGGL((conv_halo_kernel<128, 4, 5, 264>), grid, dim3(NT), 0, st, g, (unsigned)a_bytes, (unsigned)b_bytes, tw); break;
        default: return 1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

// pdmk.h: can the 2x2 phase form serve the nearest-x2 upsample + 3x3 conv of a [B, H, W, Ci] image to Co channels?
extern "C" int pdmk_conv_up2_supported(int B, int H, int W, int Ci, int Co, int dtype) {
    if (dtype != PDMK_BF16 || B <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0 || (Ci % 32) || (Co % 32)) return 0;
    pdmk_gemm_args g = {};
    g.a_mode = PDMK_A_CONV;
    g.b_mode = PDMK_B_ROWK;
    g.dtype = PDMK_BF16;
    g.conv_b = B;
    g.conv_hi = g.conv_ho = H;
    g.conv_wi = g.conv_wo = W;
    g.M = B * H * W;
    bool ok = true;
    for (int dir = 0; dir < 2; ++dir) {          // forward (Ci -> Co) and input gradient (Co -> Ci)
        g.conv_mode = dir ? 13 : 5;
        g.conv_ci = dir ? Co : Ci;
        g.N = dir ? Ci : Co;
        g.K = (dir ? 16 : 4) * g.conv_ci;
        g.ldb = g.K;
        ok = ok && (pdmk_ring::halo_tile_w(g, 2, 1) > 0 || pdmk_ring::halo_tile_w(g, 3, 1) > 0);
    }
    return ok ? 1 : 0;
}

int pdmk_gemm_ring_num_configs() { return pdmk_ring::kNumConfigs + kNumHalo + pdmk_gemm_rowblock_num_configs(); }
int pdmk_gemm_ring_name(int id, int conv, char* buf, int n) {      // the demangled symbol rocprofv3 reports
    using namespace pdmk_ring;
    if (id >= kNumConfigs + kNumHalo) return pdmk_gemm_rowblock_name(id - (kNumConfigs + kNumHalo), buf, n);
    if (id < 0) return -1;
    if (id >= kNumBase && id < kNumBase + kNumHalo) {
        const int h = id - kNumBase;
        snprintf(buf, n, "pdmk_ring::conv_halo_kernel<%d, %d, %d, %d, 9>", h < 2 ? 256 : 128, (h & 1) ? 4 : 5,
                 h < 2 ? 3 : ((h & 1) ? 5 : 4), h < 2 ? 400 : 264);          // (2x2 phase launches run the <..., 4> instantiation)
        return 0;
    }
    const Config c = kConfigs[id < kNumBase ? id : id - kNumHalo];
    snprintf(buf, n, "pdmk_ring::igemm_ring_kernel<%s, %d, %d, %d, %d>", conv ? "true" : "false", c.bm, c.nj, c.stages, c.occ);
    return 0;
}
int pdmk_gemm_ring_pick(const pdmk_gemm_args& g) { return pdmk_ring::pick_config(g, g.splitk); }

// called by pdmk_gemm (gemm.hip) after argument validation; returns 1 if the shape/config is not handled here
int pdmk_gemm_ring_launch(const pdmk_gemm_args& g, hipStream_t st, long a_bytes, long b_bytes, int id) {
    using namespace pdmk_ring;
    if (g.dtype != PDMK_BF16 || g.b_mode != PDMK_B_ROWK || g.a_mode == PDMK_A_COLK) return 1;
    if ((g.K % 8) || (g.a_mode == PDMK_A_CONV && (g.conv_ci % 8))) return 1;
    if (g.ln_gamma && id < kNumConfigs + kNumHalo) return 1;         // LayerNorm prologue: the row-block kernel only
    if (id >= kNumBase && id < kNumBase + kNumHalo) return conv_halo_launch(g, st, a_bytes, b_bytes, id - kNumBase);
    if (g.a_mode == PDMK_A_CONV && g.conv_mode >= 5) return 1;       // 2x2 phase convs: halo kernels only
    if (id >= kNumConfigs + kNumHalo) return pdmk_gemm_rowblock_launch(g, st, a_bytes, b_bytes, id - (kNumConfigs + kNumHalo));
    if (id < 0) return 1;
    const int sk = g.splitk > 1 ? g.splitk : 1;
    const Config c = kConfigs[id < kNumBase ? id : id - kNumHalo];
    const int bn = 32 * c.nj;
    dim3 grid(((g.M + c.bm - 1) / c.bm) * ((g.N + bn - 1) / bn), sk);
    const bool conv = g.a_mode == PDMK_A_CONV;
#define PDMK_RING_GO(BMv, NJv, STv, OCv)                                                                              \
    case (BMv * 1000 + NJv * 100 + STv * 10 + OCv):                                                                   \
        if (conv) hipLaunchKernelGGL((igemm_ring_kernel<true, BMv, NJv, STv, OCv>), grid, dim3(NT), 0, st, g,         \
                                     (unsigned)a_bytes, (unsigned)b_bytes);                                           \
        else hipLaunchKernelGGL((igemm_ring_kernel<false, BMv, NJv, STv, OCv>), grid, dim3(NT), 0, st, g,             \
                                (unsigned)a_bytes, (unsigned)b_bytes);                                                \
        break;
    switch (c.bm * 1000 + c.nj * 100 + c.stages * 10 + c.occ) {
        PDMK_RING_GO(256, 4, 3, 2) PDMK_RING_GO(256, 5, 3, 2) PDMK_RING_GO(128, 4, 4, 2) PDMK_RING_GO(128, 5, 4, 2)
        PDMK_RING_GO(64, 4, 6, 2) PDMK_RING_GO(64, 5, 5, 2) PDMK_RING_GO(128, 4, 2, 4) PDMK_RING_GO(64, 4, 3, 4)
        PDMK_RING_GO(64, 5, 2, 4) PDMK_RING_GO(128, 6, 3, 2) PDMK_RING_GO(64, 6, 4, 2) PDMK_RING_GO(128, 5, 2, 4)
        PDMK_RING_GO(64, 2, 4, 4) PDMK_RING_GO(128, 2, 3, 4) PDMK_RING_GO(64, 3, 4, 4)
        default: return 1;
    }
#undef PDMK_RING_GO
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

namespace pdmk_ring {

// ------------------------------------------------------------------------------------------------------------------
// Halo weight gradient of a stride-1 3x3 conv:  dW[co][tap][ci] (+)= sum_px dY[px][co] * X[src(px, tap)][ci].
// The ring kernel above gathers X once per tap (9 x the image through L2 -> LDS for one pass over dY).  Here a workgroup
// owns dW[64 co][9 taps][64 ci] and walks 128-pixel blocks (whole image rows, or whole small images): per block the dY
// tile [128 px][64 / 128 co] and the (rows+2) x (W+2) input patch [<= 288 px][64 ci] are LDS-DMA-ed ONCE and all nine taps
// read the patch at shifted rows with transposing ds_read_b64_tr_b16 (patch swizzle chunk ^= row & 7: conflict-free
// for every shift, as in conv_halo_kernel / the attention tiles).  49 KiB of intake per 9.4 MFLOP (192 FLOP/B) against
// 32 KiB per 2.1 MFLOP (65 FLOP/B) for the 128x128 ring tile.  8 waves = 2 (co halves) x 4 (16-channel ci slices);
// a wave keeps 2 x 9 accumulator tiles (72 VGPRs).  Pixel blocks are split over wgc.gy; partial tiles are combined
// with fp32 atomics in 256-byte rows (plain stores / read-add-store when there is a single split).
constexpr int wgrad_halo_smem_bytes(int IMW, int STAGES) { return STAGES * (128 * (32 * IMW) * 2 + 288 * 128); }

template <int IMW, int STAGES>      // IMW = 16-row co tiles per wave: 2 (64 co per workgroup) or 4 (128 co)
__device__ __forceinline__ void conv_wgrad_halo_body(const pdmk_gemm_args& g, unsigned a_bytes, unsigned b_bytes, const LC wgc,
                                                     unsigned char* smem) {
    typedef Mma<bf16> MM;
    constexpr int PX = 128, PMAX = 288, BM = 32 * IMW, RA = BM * 2;       // RA = bytes per pixel row of the dY tile
    constexpr int A_BYTES = PX * RA, P_BYTES = PMAX * 128, SLOT = A_BYTES + P_BYTES;
    constexpr int PA = A_BYTES / 1024 / 8;                            // dY pieces per wave and stage
    constexpr int NPW = (PMAX / 8 + 7) / 8;                           // patch pieces per wave and stage (constant count)
    constexpr int NDMA = PA + NPW;                                    // DMA instructions per wave and stage
    static_assert(STAGES >= 2 && STAGES * SLOT <= 160 * 1024, "LDS");
    static_assert((IMW == 2 || IMW == 4) && BM * 68 * 4 <= STAGES * SLOT, "epilogue staging");
    static_assert(STAGES * SLOT == wgrad_halo_smem_bytes(IMW, STAGES), "smem size of the kernel wrappers");

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: wave-dependent branches and counts stay in SGPRs
    const int wm = wave >> 2, wn = wave & 3;
    const int Ci = g.conv_ci, H = g.conv_hi, W = g.conv_wi, HW = H * W;
    const int nct = (Ci + 63) / 64;
    const int tile = xcd_remap(wgc.bx, wgc.gx);
    const int m0 = (tile / nct) * BM, c0 = (tile - (tile / nct) * nct) * 64;
    const int nblk = (g.K + PX - 1) / PX;
    const int per = (nblk + wgc.gy - 1) / wgc.gy;
    const int kb0 = wgc.by * per, kb1 = min(nblk, kb0 + per);
    if (kb0 >= kb1 && g.accumulate != 2) return;        // slab split-K: an empty split still writes its (zero) slab

    const int rimg = HW >= PX ? PX / W : H;                           // rows of an image inside a pixel block
    // patch row stride padded to a multiple of 8 pixels: (row + ky * W2) & 7 == row & 7, so the swizzle term of a tap
    // address does not depend on ky and the three ky reads of a kx share one address computation
    const int W2 = (W + 2 + 7) & ~7, pimg = (rimg + 2) * W2;
    const int nimg = HW >= PX ? 1 : PX / HW;
    const int prows = nimg * pimg;

    const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), (short)0, (int)a_bytes, 0x00020000);
    const auto rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.B), (short)0, (int)b_bytes, 0x00020000);

    // ---- dY loader (as wgrad_ring): 128-byte rows: piece = 8 pixel rows, lane -> (row lane>>3, chunk lane&7);
    //      256-byte rows: piece = 4 pixel rows, lane -> (row lane>>4, chunk lane&15); swizzles 2 f128(k) / 2 f256(k)
    const int a_row = RA == 256 ? (lane >> 4) : (lane >> 3);
    const int a_lc = RA == 256 ? ((lane & 15) ^ (((lane >> 4) | (((wave >> 1) & 1) << 2)) << 1))
                               : ((lane & 7) ^ (((((lane >> 3) >> 1) & 1) | ((wave & 1) << 1)) << 1));
    const int acol = m0 + a_lc * 8;
    const bool a_ok = acol < g.M;
    // ---- patch loader: piece = j*8 + wave, lane -> (row sr, physical chunk lane&7); logical chunk ^= row & 7
    const int sr = lane >> 3;
    const int pch = c0 + (((lane & 7) ^ sr) << 3);                    // first channel of this lane's chunk
    const bool p_ok = pch < Ci;
    int p_il[NPW], p_dy[NPW], p_dx[NPW];                              // patch row -> (image in block, y - y0, x); il < 0: none
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
        const int prow = (j * 8 + wave) * 8 + sr;
        p_il[j] = -1;
        p_dy[j] = p_dx[j] = 0;
        if (prow < prows) {
            const int il = prow / pimg, rem = prow - il * pimg;
            const int py = rem / W2;
            p_il[j] = il;
            p_dy[j] = py - 1;
            p_dx[j] = rem - py * W2 - 1;
        }
    }

    auto issue = [&](int kb, int slot) {
        unsigned char* sa = smem + slot * SLOT;
        const int p0 = kb * PX;
        const int img0 = p0 / HW, y0 = (p0 - img0 * HW) / W;
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int piece = i * 8 + wave;
            const int px = p0 + piece * (1024 / RA) + a_row;
            const unsigned va = (px < g.K && a_ok) ? ((unsigned)px * (unsigned)g.lda + (unsigned)acol) * 2u : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void*)(sa + piece * 1024), 16, (int)va, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            // surplus pieces (beyond the patch) re-issue piece j = 0 of this wave: identical bytes, constant DMA count
            const bool real = (j * 8 + wave) * 8 < prows;
            const int jj = real ? j : 0;
            const int il = real ? p_il[j] : p_il[0], dy = real ? p_dy[j] : p_dy[0], dx = real ? p_dx[j] : p_dx[0];
            unsigned vb = OOB;
            const int b = img0 + il, y = y0 + dy;
            if (il >= 0 && p_ok && b < g.conv_b && (unsigned)y < (unsigned)H && (unsigned)dx < (unsigned)W)
                vb = ((unsigned)((b * H + y) * W + dx) * (unsigned)g.conv_ld + (unsigned)pch) * 2u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_void*)(sa + A_BYTES + (jj * 8 + wave) * 1024), 16, (int)vb, 0, 0, 0);
        }
    };

    f32x4 acc[IMW][9], acs[IMW];
#pragma unroll
    for (int i = 0; i < IMW; ++i) {
        acs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool do_colsum = g.colsum_out != nullptr && c0 == 0 && wn == 0;      // wave-uniform
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

    // ---- transposing fragment reads: lane (tg, tq, tp) reads 8 bytes of k-row kk + 8 tg + tq (and of row + 4)
    const int tg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int fsw = RA == 256 ? ((tq | ((tg & 1) << 2)) << 1) : ((((tq >> 1) & 1) | ((tg & 1) << 1)) << 1);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    auto tr_a = [&](const unsigned char* base, int kk, int col0) -> bf16x8 {
        const unsigned ch = (unsigned)(((col0 >> 3) ^ fsw) | (tp >> 1));
        const unsigned char* a0 = base + (unsigned)(kk + 8 * tg + tq) * (unsigned)RA + ch * 16u + (unsigned)(tp & 1) * 8u;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * RA));
        s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    };
    // patch row of pixel k of a block for tap (0,0): k -> (image il, row yl, column x)
    int prk[4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = c * 32 + 8 * tg + tq + 4 * h;
            const int il = k / (rimg * W), rem = k - il * (rimg * W);
            const int yl = rem / W;
            prk[c][h] = il * pimg + yl * W2 + (rem - yl * W);
        }
    const unsigned bch = (unsigned)(2 * wn + (tp >> 1)), bsub = (unsigned)(tp & 1) * 8u;
    const int rowb = W2 * 128;                                       // bytes between patch rows ky and ky + 1

#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (kb0 + s < kb1) issue(kb0 + s, s);

    int slot = 0;
    for (int kb = kb0; kb < kb1; ++kb) {
        const int ahead = min(STAGES - 2, kb1 - 1 - kb);
        wait_vmcnt_dyn(ahead * NDMA);
        __builtin_amdgcn_s_barrier();
        if (kb + STAGES - 1 < kb1) issue(kb + STAGES - 1, slot == 0 ? STAGES - 1 : slot - 1);
        const unsigned char* sa = smem + slot * SLOT;
        const unsigned char* pb = sa + A_BYTES;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bf16x8 af[IMW];
#pragma unroll
            for (int i = 0; i < IMW; ++i) af[i] = tr_a(sa, c * 32, wm * (16 * IMW) + i * 16);
            asm volatile("" : "+v"(prk[c][0]), "+v"(prk[c][1]));     // keep the shifted addresses out of the loop invariants
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                const int r0 = prk[c][0] + tx, r1 = prk[c][1] + tx;
                const unsigned char* q0 = pb + (unsigned)r0 * 128u + ((bch ^ (unsigned)(r0 & 7)) << 4) + bsub;
                const unsigned char* q1 = pb + (unsigned)r1 * 128u + ((bch ^ (unsigned)(r1 & 7)) << 4) + bsub;
#pragma unroll
                for (int ty = 0; ty < 3; ++ty) {
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q0 + ty * rowb));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q1 + ty * rowb));
                    const bf16x8 bf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                    for (int i = 0; i < IMW; ++i) acc[i][ty * 3 + tx] = MM::mma(bf, af[i], acc[i][ty * 3 + tx]);
                }
            }
            if (do_colsum) {
#pragma unroll
                for (int i = 0; i < IMW; ++i) acs[i] = MM::mma(ones, af[i], acs[i]);
            }
        }
        slot = slot + 1 == STAGES ? 0 : slot + 1;
    }
    if (do_colsum && (lane >> 4) == 0) {
#pragma unroll
        for (int i = 0; i < IMW; ++i) {
            const int m = m0 + wm * (16 * IMW) + i * 16 + (lane & 15);
            if (m < g.M) unsafeAtomicAdd(g.colsum_out + m, acs[i][0]);
        }
    }
    // ---- epilogue: one tap at a time through a [BM][68] fp32 staging image; a wave writes BM/8 rows of 64 floats
    constexpr int SROW = 68;
    float* stage = reinterpret_cast<float*>(smem);
    // accumulate = 2: split z stores its partial tile into slab z of a [splits][M][ldc] workspace with plain stores (summed later
    // by pdmk_splitk_finish_group); else several splits add with fp32 atomics - ~180 KB per workgroup at ~5 GB/s per CU, which
    // is what bounded this kernel
    const bool slab = g.accumulate == 2;
    float* Cf = reinterpret_cast<float*>(g.C) + (slab ? (long)wgc.by * g.M * g.ldc : 0L);
    const bool atomic = wgc.gy > 1 && !slab, acc1 = g.accumulate == 1;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < IMW; ++i)
            *reinterpret_cast<f32x4*>(stage + (wm * (16 * IMW) + i * 16 + (lane & 15)) * SROW + wn * 16 + (lane >> 4) * 4) = acc[i][t];
        __syncthreads();
        const int ci = c0 + lane;
        if (ci < Ci) {
#pragma unroll
            for (int rr = 0; rr < BM / 8; ++rr) {
                const int row = wave * (BM / 8) + rr, m = m0 + row;
                if (m < g.M) {
                    const float v = stage[row * SROW + lane];
                    float* dst = Cf + (long)m * g.ldc + t * Ci + ci;
                    if (atomic) unsafeAtomicAdd(dst, v);
                    else *dst = acc1 ? *dst + v : v;
                }
            }
        }
    }
}

template <int IMW, int STAGES>
__global__ __launch_bounds__(NT, 1) void conv_wgrad_halo_kernel(pdmk_gemm_args g, unsigned a_bytes, unsigned b_bytes) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[wgrad_halo_smem_bytes(IMW, STAGES)];
    conv_wgrad_halo_body<IMW, STAGES>(g, a_bytes, b_bytes, PDMK_WGRAD_XCD ? lc_split_xcd() : lc_plain(), smem);
}
template <int IMW, int STAGES>
__global__ __launch_bounds__(NT, 1) void conv_wgrad_halo_group_kernel(pdmk_gemm_group_dev gg) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[wgrad_halo_smem_bytes(IMW, STAGES)];
    int pi;
    LC wgc;
    if (!lc_group(gg, pi, wgc)) return;
    conv_wgrad_halo_body<IMW, STAGES>(gg.p[pi], gg.a_bytes[pi], gg.b_bytes[pi], wgc, smem);
}

static bool wgrad_halo_ok(const pdmk_gemm_args& g) {
    if (g.dtype != PDMK_BF16 || g.a_mode != PDMK_A_COLK || g.b_mode != PDMK_B_COLK_CONV || !g.out_f32) return false;
    if (g.conv_mode != 0 || g.conv_ho != g.conv_hi || g.conv_wo != g.conv_wi || g.alpha != 1.0f) return false;
    if ((g.conv_ci % 8) || (g.conv_ld % 8) || (g.lda % 8) || (g.M % 8) || g.N != 9 * g.conv_ci || g.conv_wi < 4) return false;
    const int H = g.conv_hi, W = g.conv_wi, HW = H * W, W2 = (W + 2 + 7) & ~7;
    if (HW >= 128) return (HW % 128) == 0 && (128 % W) == 0 && (128 / W + 2) * W2 <= 288;
    return (128 % HW) == 0 && (128 / HW) * (H + 2) * W2 <= 288;
}

}  // namespace pdmk_ring

// ---- grouped launches (pdmk_gemm_group): candidate `id` (numbering of pdmk_gemm_ring_launch) for n problems in one grid.
// Returns 1 when a problem is not served by that candidate (the caller then launches the problems one by one).
namespace {
struct GroupGrid {
    pdmk_gemm_group_dev gg;
    unsigned total = 0;
    void add(int i, const pdmk_gemm_args& g, int tiles, int splits, long ab, long bb, int aux0 = 0, int aux1 = 0) {
        gg.start[i] = (int)total;
        gg.gx[i] = tiles;
        gg.gy[i] = splits;
        gg.a_bytes[i] = (unsigned)ab;
        gg.b_bytes[i] = (unsigned)bb;
        gg.aux0[i] = aux0;
        gg.aux1[i] = aux1;
        gg.p[i] = g;
        total += ((unsigned)tiles * (unsigned)splits + 7u) & ~7u;
    }
    void finish(int n) {
        gg.n = n;
        for (int i = n; i <= PDMK_GEMM_GROUP_MAX; ++i) gg.start[i] = (int)total;
        for (int i = n; i < PDMK_GEMM_GROUP_MAX; ++i) {
            gg.gx[i] = gg.gy[i] = 0;
            gg.a_bytes[i] = gg.b_bytes[i] = 0;
            gg.aux0[i] = gg.aux1[i] = 0;
            gg.p[i] = gg.p[0];
        }
    }
};
}  // namespace

int pdmk_gemm_ring_group_launch(const pdmk_gemm_args* gs, int n, hipStream_t st, const long* a_bytes, const long* b_bytes, int id) {
    using namespace pdmk_ring;
    if (n < 2 || n > PDMK_GEMM_GROUP_MAX || id < 0) return 1;
    const bool halo = id >= kNumBase && id < kNumBase + kNumHalo;
    if (id >= kNumConfigs + kNumHalo) return 1;                       // row-block kernels: not grouped
    const bool conv = gs[0].a_mode == PDMK_A_CONV;
    GroupGrid G;
    for (int i = 0; i < n; ++i) {
        const pdmk_gemm_args& g = gs[i];
        if (g.dtype != PDMK_BF16 || g.b_mode != PDMK_B_ROWK || g.a_mode == PDMK_A_COLK) return 1;
        if ((g.K % 8) || (g.a_mode == PDMK_A_CONV && (g.conv_ci % 8))) return 1;
        if ((g.a_mode == PDMK_A_CONV) != conv) return 1;              // one template instantiation per launch
        if (!halo && g.a_mode == PDMK_A_CONV && g.conv_mode >= 5) return 1;
        const int sk = g.splitk > 1 ? g.splitk : 1;
        if (halo) {
            const int h = id - kNumBase;
            const int tw = halo_tile_w(g, h, g.splitk);
            if (tw <= 0) return 1;
            const int bm = h < 2 ? 256 : 128, bn = 32 * ((h & 1) ? 4 : 5);
            G.add(i, g, ((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn), sk, a_bytes[i], b_bytes[i], tw);
        } else {
            const Config c = kConfigs[id < kNumBase ? id : id - kNumHalo];
            const int bn = 32 * c.nj;
            G.add(i, g, ((g.M + c.bm - 1) / c.bm) * ((g.N + bn - 1) / bn), sk, a_bytes[i], b_bytes[i]);
        }
    }
    G.finish(n);
    const dim3 grid(G.total);
    if (halo) {
        bool phase = gs[0].conv_mode >= 5;
        for (int i = 1; i < n; ++i)
            if ((gs[i].conv_mode >= 5) != phase) return 1;
        if (phase) {
            if (id - kNumBase == 2) hipLaunchKernelGGL((conv_halo_group_kernel<128, 5, 4, 264, 4>), grid, dim3(NT), 0, st, G.gg);
            else hipLaunchKernelGGL((conv_halo_group_kernel<128, 4, 5, 264, 4>), grid, dim3(NT), 0, st, G.gg);
            return hipGetLastError() == hipSuccess ? 0 : -1000;
        }
        switch (id - kNumBase) {
            case 0: hipLaunchKernelGGL((conv_halo_group_kernel<256, 5, 3, 400>), grid, dim3(NT), 0, st, G.gg); break;
            case 1: hipLaunchKernelGGL((conv_halo_group_kernel<256, 4, 3, 400>), grid, dim3(NT), 0, st, G.gg); break;
            case 2: hipLaunchKernelGGL((conv_halo_group_kernel<128, 5, 4, 264>), grid, dim3(NT), 0, st, G.gg); break;
            default: hipLaunchKernelGGL((conv_halo_group_kernel<128, 4, 5, 264>), grid, dim3(NT), 0, st, G.gg); break;
        }
        return hipGetLastError() == hipSuccess ? 0 : -1000;
    }
    const Config c = kConfigs[id < kNumBase ? id : id - kNumHalo];
#define PDMK_RING_GGO(BMv, NJv, STv, OCv)                                                                             \
    case (BMv * 1000 + NJv * 100 + STv * 10 + OCv):                                                                   \
        if (conv) hipLaunchKernelGGL((igemm_ring_group_kernel<true, BMv, NJv, STv, OCv>), grid, dim3(NT), 0, st, G.gg); \
        else hipLaunchKernelGGL((igemm_ring_group_kernel<false, BMv, NJv, STv, OCv>), grid, dim3(NT), 0, st, G.gg);   \
        break;
    switch (c.bm * 1000 + c.nj * 100 + c.stages * 10 + c.occ) {
        PDMK_RING_GGO(256, 4, 3, 2) PDMK_RING_GGO(256, 5, 3, 2) PDMK_RING_GGO(128, 4, 4, 2) PDMK_RING_GGO(128, 5, 4, 2)
        PDMK_RING_GGO(64, 4, 6, 2) PDMK_RING_GGO(64, 5, 5, 2) PDMK_RING_GGO(128, 4, 2, 4) PDMK_RING_GGO(64, 4, 3, 4)
        PDMK_RING_GGO(64, 5, 2, 4) PDMK_RING_GGO(128, 6, 3, 2) PDMK_RING_GGO(64, 6, 4, 2) PDMK_RING_GGO(128, 5, 2, 4)
        PDMK_RING_GGO(64, 2, 4, 4) PDMK_RING_GGO(128, 2, 3, 4) PDMK_RING_GGO(64, 3, 4, 4)
        default: return 1;
    }
#undef PDMK_RING_GGO
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

// ---- weight-gradient ring candidates: 128x128 deep / shallow rings, and the smaller tiles whose split-K epilogue moves a
// quarter to a half of the atomic bytes per workgroup (what bounds the small weights: ~5 GB/s of atomics per CU)
struct WCfg { int bm, nj, stages, occ; };
static const WCfg kWCfgs[] = {{128, 4, 4, 2}, {128, 4, 2, 4}, {64, 4, 3, 4}, {128, 2, 3, 4}, {64, 2, 4, 4}};
constexpr int kNumW = sizeof(kWCfgs) / sizeof(kWCfgs[0]);
int pdmk_wgrad_ring_num_configs() { return kNumW + 2; }      // + the halo conv weight gradients (64 / 128 output rows)
int pdmk_wgrad_ring_name(int id, int conv, char* buf, int n) {
    if (id == kNumW || id == kNumW + 1) {
        snprintf(buf, n, "pdmk_ring::conv_wgrad_halo_kernel<%d, %d>", id == kNumW ? 2 : 4, id == kNumW ? 3 : 2);
        return 0;
    }
    if (id < 0 || id >= kNumW) return -1;
    const WCfg c = kWCfgs[id];
    snprintf(buf, n, "pdmk_ring::wgrad_ring_kernel<%s, %d, %d, %d, %d>", conv ? "true" : "false", c.bm, c.nj, c.stages, c.occ);
    return 0;
}
int pdmk_wgrad_ring_launch(const pdmk_gemm_args& g, hipStream_t st, long a_bytes, long b_bytes, int id) {
    using namespace pdmk_ring;
    if (g.dtype != PDMK_BF16 || g.a_mode != PDMK_A_COLK || g.b_mode == PDMK_B_ROWK || !g.out_f32) return 1;
    const bool conv = g.b_mode == PDMK_B_COLK_CONV;
    if ((g.M % 8) || (g.N % 8) || (g.lda % 8) || (!conv && (g.ldb % 8))) return 1;
    if (conv && ((g.conv_ci % 8) || (g.conv_ld % 8) || g.conv_mode == 3 || g.conv_mode == 4 || g.conv_mode > 8)) return 1;
    if (id == kNumW || id == kNumW + 1) {
        if (!wgrad_halo_ok(g)) return 1;
        const int nblk = (g.K + 127) / 128, sk = g.splitk > 1 ? g.splitk : 1, bm = id == kNumW ? 64 : 128;
        if (sk > nblk) return 1;
        dim3 grid(((g.M + bm - 1) / bm) * ((g.conv_ci + 63) / 64), sk);
        if (bm == 64) hipLaunchKernelGGL((conv_wgrad_halo_kernel<2, 3>), grid, dim3(NT), 0, st, g, (unsigned)a_bytes, (unsigned)b_bytes);
        else hipLaunchKernelGGL((conv_wgrad_halo_kernel<4, 2>), grid, dim3(NT), 0, st, g, (unsigned)a_bytes, (unsigned)b_bytes);
        return hipGetLastError() == hipSuccess ? 0 : -1000;
    }
    if (id < 0 || id >= kNumW) return 1;
    int lg_wo = -1, lg_howo = -1;
    if (conv) {
        auto lg = [](int v) { int l = 0; if (v <= 0 || (v & (v - 1))) return -1; while ((1 << l) < v) ++l; return l; };
        lg_wo = lg(g.conv_wo);
        lg_howo = lg(g.conv_ho * g.conv_wo);
        if (lg_wo < 0 || lg_howo < 0) lg_wo = lg_howo = -1;
    }
    const WCfg c = kWCfgs[id];
    const int bn = 32 * c.nj;
    dim3 grid(((g.M + c.bm - 1) / c.bm) * ((g.N + bn - 1) / bn), g.splitk > 1 ? g.splitk : 1);
#define PDMK_WG_GO(BMv, NJv, STv, OCv)                                                                                \
    case (BMv * 1000 + NJv * 100 + STv * 10 + OCv):                                                                   \
        if (conv) hipLaunchKernelGGL((wgrad_ring_kernel<true, BMv, NJv, STv, OCv>), grid, dim3(NT), 0, st, g, lg_wo,  \
                                     lg_howo, (unsigned)a_bytes, (unsigned)b_bytes);                                  \
        else hipLaunchKernelGGL((wgrad_ring_kernel<false, BMv, NJv, STv, OCv>), grid, dim3(NT), 0, st, g, lg_wo,      \
                                lg_howo, (unsigned)a_bytes, (unsigned)b_bytes);                                       \
        break;
    switch (c.bm * 1000 + c.nj * 100 + c.stages * 10 + c.occ) {
        PDMK_WG_GO(128, 4, 4, 2) PDMK_WG_GO(128, 4, 2, 4) PDMK_WG_GO(64, 4, 3, 4) PDMK_WG_GO(128, 2, 3, 4) PDMK_WG_GO(64, 2, 4, 4)
        default: return 1;
    }
#undef PDMK_WG_GO
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}

// grouped weight gradients: candidate `id` (numbering of pdmk_wgrad_ring_launch) for n problems in one grid; 1 = not served
int pdmk_wgrad_ring_group_launch(const pdmk_gemm_args* gs, int n, hipStream_t st, const long* a_bytes, const long* b_bytes, int id) {
    using namespace pdmk_ring;
    if (n < 2 || n > PDMK_GEMM_GROUP_MAX || id < 0 || id > kNumW + 1) return 1;
    const bool conv = gs[0].b_mode == PDMK_B_COLK_CONV;
    const bool halo = id == kNumW || id == kNumW + 1;
    GroupGrid G;
    for (int i = 0; i < n; ++i) {
        const pdmk_gemm_args& g = gs[i];
        if (g.dtype != PDMK_BF16 || g.a_mode != PDMK_A_COLK || g.b_mode == PDMK_B_ROWK || !g.out_f32) return 1;
        if ((g.b_mode == PDMK_B_COLK_CONV) != conv) return 1;
        if ((g.M % 8) || (g.N % 8) || (g.lda % 8) || (!conv && (g.ldb % 8))) return 1;
        if (conv && ((g.conv_ci % 8) || (g.conv_ld % 8) || g.conv_mode == 3 || g.conv_mode == 4 || g.conv_mode > 8)) return 1;
        const int sk = g.splitk > 1 ? g.splitk : 1;
        if (halo) {
            if (!wgrad_halo_ok(g)) return 1;
            const int nblk = (g.K + 127) / 128, bm = id == kNumW ? 64 : 128;
            if (sk > nblk) return 1;
            G.add(i, g, ((g.M + bm - 1) / bm) * ((g.conv_ci + 63) / 64), sk, a_bytes[i], b_bytes[i]);
        } else {
            int lg_wo = -1, lg_howo = -1;
            if (conv) {
                auto lg = [](int v) { int l = 0; if (v <= 0 || (v & (v - 1))) return -1; while ((1 << l) < v) ++l; return l; };
                lg_wo = lg(g.conv_wo);
                lg_howo = lg(g.conv_ho * g.conv_wo);
                if (lg_wo < 0 || lg_howo < 0) lg_wo = lg_howo = -1;
            }
            const WCfg c = kWCfgs[id];
            const int bn = 32 * c.nj;
            G.add(i, g, ((g.M + c.bm - 1) / c.bm) * ((g.N + bn - 1) / bn), sk, a_bytes[i], b_bytes[i], lg_wo, lg_howo);
        }
    }
    G.finish(n);
    const dim3 grid(G.total);
    if (halo) {
        if (id == kNumW) hipLaunchKernelGGL((conv_wgrad_halo_group_kernel<2, 3>), grid, dim3(NT), 0, st, G.gg);
        else hipLaunchKernelGGL((conv_wgrad_halo_group_kernel<4, 2>), grid, dim3(NT), 0, st, G.gg);
        return hipGetLastError() == hipSuccess ? 0 : -1000;
    }
    const WCfg c = kWCfgs[id];
#define PDMK_WG_GGO(BMv, NJv, STv, OCv)                                                                               \
    case (BMv * 1000 + NJv * 100 + STv * 10 + OCv):                                                                   \
        if (conv) hipLaunchKernelGGL((wgrad_ring_group_kernel<true, BMv, NJv, STv, OCv>), grid, dim3(NT), 0, st, G.gg); \
        else hipLaunchKernelGGL((wgrad_ring_group_kernel<false, BMv, NJv, STv, OCv>), grid, dim3(NT), 0, st, G.gg);   \
        break;
    switch (c.bm * 1000 + c.nj * 100 + c.stages * 10 + c.occ) {
        PDMK_WG_GGO(128, 4, 4, 2) PDMK_WG_GGO(128, 4, 2, 4) PDMK_WG_GGO(64, 4, 3, 4) PDMK_WG_GGO(128, 2, 3, 4) PDMK_WG_GGO(64, 2, 4, 4)
        default: return 1;
    }
#undef PDMK_WG_GGO
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}
