// Implicit GEMM on the CDNA4 matrix cores: Linear / Conv3x3 (NHWC) forward, dgrad and wgrad in one template.
//   C[M,N] (+)= alpha * sum_k A(m,k) B(n,k) + bias[n] + rowvec[m/rpb][n] + R[m][n]
// Workgroup = 256 threads = 4 waves (2x2), tile 128x128, K-step 64 (bf16) / 32 (fp32); each wave owns a 64x64
// sub-tile as 4x4 MFMA 16x16 accumulators (32 bf16 MFMAs per barrier).  Operands are staged global -> registers -> LDS
// (double buffered, one barrier per K-step); the next tile's 8 x 16-byte loads per thread are issued BEFORE the current
// tile's MFMAs and consumed (zero-filled + written to LDS) after them, so HBM/L2 latency hides under the matrix work.
// The register hop is what makes zero-filled halos (conv padding), tails and the four gather geometries free.
// Gather index math is incremental (no integer division in the K loop): each thread keeps its (tap, ci) / pixel
// coordinates and advances them by one K-step.  Reduction-major operands (wgrad) stay in memory order in LDS and are
// transposed by ds_read_b64_tr_b16 on the way into the MFMA fragments.  The MFMA is issued with operands swapped
// (D = B.A^T) so a lane ends up with 4 consecutive n of one output row: 8/16-byte epilogue loads and stores.
// Blocks are remapped so that the n-tiles of one m-tile run back to back on one XCD (shared activation panel in L2).
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, NTHREADS = 256;
constexpr int OPERAND_BYTES = 18432;      // 128 x 144 B  ==  64(32) k-rows x 144 elements

template <typename T> struct TileCfg {
    static constexpr int CH = Mma<T>::CH;               // elements per 16-byte chunk
    static constexpr int BK = 8 * CH;                   // 64 bf16 / 32 fp32: 8 chunks per tile row
    static constexpr int RS_ROWK = BK + CH;             // 144-byte rows
    static constexpr int RS_COLK = 128 + 16;
    static constexpr int CPR = 128 / CH;                // chunks per 128-wide row (colk layouts)
    static constexpr int KROW_STEP = NTHREADS / CPR;    // 16 / 8
    static_assert(BM * RS_ROWK * sizeof(T) == OPERAND_BYTES, "rowk tile bytes");
    static_assert(BK * RS_COLK * sizeof(T) == OPERAND_BYTES, "colk tile bytes");
    static_assert(4 * KROW_STEP == BK, "4 k-rows per thread");
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
    if (CMODE == 1) { vy = 2 * oy + ky - 1; vx = 2 * ox + kx - 1; }
    else            { vy = oy + ky - 1;     vx = ox + kx - 1; }
    const int hv = (CMODE >= 2) ? 2 * g.hi : g.hi, wv = (CMODE >= 2) ? 2 * g.wi : g.wi;
    bool ok = (unsigned)vy < (unsigned)hv && (unsigned)vx < (unsigned)wv;
    if (CMODE == 3) ok = ok && (((vy | vx) & 1) == 0);
    const int iy = (CMODE >= 2) ? (vy >> 1) : vy, ix = (CMODE >= 2) ? (vx >> 1) : vx;
    return ok ? (b * g.hi + iy) * g.wi + ix : -1;
}

__device__ __forceinline__ uint4 ld16(const void* p) { return *reinterpret_cast<const uint4*>(p); }

template <typename T, int AMODE, int BMODE, int CMODE>
__global__ __launch_bounds__(NTHREADS) void igemm_kernel(pdmk_gemm_args g, int lg_wo, int lg_howo) {
    typedef TileCfg<T> TC;
    typedef Mma<T> MM;
    constexpr int CH = TC::CH, BK = TC::BK;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][OPERAND_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (g.N + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;

    const int nk_total = (g.K + BK - 1) / BK;
    const int per = (nk_total + gridDim.y - 1) / gridDim.y;
    const int kt0 = blockIdx.y * per;
    const int kt1 = min(nk_total, kt0 + per);
    if (kt0 >= kt1) return;

    const T* __restrict__ Ap = reinterpret_cast<const T*>(g.A);
    const T* __restrict__ Bp = reinterpret_cast<const T*>(g.B);
    const ConvGeom cg{g.conv_hi, g.conv_wi, g.conv_ci, g.conv_ho, g.conv_wo, g.conv_ld, lg_wo, lg_howo};

    // ---------------------------------------------------------------- per-thread loader state
    // rowk-type operands: 4 rows (r + 32 s), one fixed 16-byte k-chunk
    const int lr = tid >> 3, kc = (tid & 7) * CH;
    // colk-type operands: 4 k-rows (kr + KROW_STEP s), one fixed column chunk
    const int ckr = tid / TC::CPR, cmc = (tid % TC::CPR) * CH;

    long a_off[4] = {-1, -1, -1, -1};    // rowk: element offset of the row start (or -1)
    int a_b[4] = {-1, -1, -1, -1}, a_oy[4] = {0, 0, 0, 0}, a_ox[4] = {0, 0, 0, 0};   // conv: output pixel coords
    int a_tap = 0, a_ci = 0;             // conv: (tap, ci) of this thread's chunk in the CURRENT k-tile
    if (AMODE == PDMK_A_ROWK) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int m = m0 + lr + 32 * s;
            a_off[s] = m < g.M ? (long)m * g.lda : -1;
        }
    } else if (AMODE == PDMK_A_CONV) {
        const int hw = cg.ho * cg.wo;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int m = m0 + lr + 32 * s;
            if (m < g.M) {
                a_b[s] = m / hw;
                const int rem = m - a_b[s] * hw;
                a_oy[s] = rem / cg.wo;
                a_ox[s] = rem - a_oy[s] * cg.wo;
            }
        }
        const int k = kt0 * BK + kc;
        a_tap = k / cg.ci;
        a_ci = k - a_tap * cg.ci;
    }
    long b_off[4] = {-1, -1, -1, -1};
    int b_tap = 0, b_ci = 0;
    if (BMODE == PDMK_B_ROWK) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int n = n0 + lr + 32 * s;
            b_off[s] = n < g.N ? (long)n * g.ldb : -1;
        }
    } else if (BMODE == PDMK_B_COLK_CONV) {
        const int col = n0 + cmc;
        b_tap = col / cg.ci;
        b_ci = col - b_tap * cg.ci;
    }

    // Loads are always issued from a valid address (the operand base when predicated off); the zero-fill is applied
    // when the registers go to LDS, AFTER the MFMAs: a data select right behind the load would force vmcnt(0) early.
    uint4 ra[4], rb[4];
    unsigned pmask = 0;   // bit s: ra[s] valid, bit 4+s: rb[s] valid
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    // fused bias gradient (wgrad only): the blocks of the first n-tile also sum their A = dY tiles over the reduction
    // dim; every tile passes through store_tiles exactly once, which is where the (zero-filled) registers are summed.
    const bool do_colsum = (AMODE == PDMK_A_COLK) && g.colsum_out != nullptr && n0 == 0;
    float csum[CH];
#pragma unroll
    for (int e = 0; e < CH; ++e) csum[e] = 0.f;

    auto load_tiles = [&](int kt) {
        const int k0 = kt * BK;
        pmask = 0;
        // ---- A
        if (AMODE == PDMK_A_ROWK) {
            const int k = k0 + kc;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bool ok = a_off[s] >= 0 && k < g.K;
                pmask |= (unsigned)ok << s;
                ra[s] = ld16(ok ? Ap + a_off[s] + k : Ap);
            }
        } else if (AMODE == PDMK_A_CONV) {
            const bool kok = k0 + kc < g.K;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int px = (kok && a_b[s] >= 0) ? conv_src_pixel<CMODE>(cg, a_b[s], a_oy[s], a_ox[s], a_tap) : -1;
                pmask |= (unsigned)(px >= 0) << s;
                ra[s] = ld16(px >= 0 ? Ap + (long)px * cg.ld + a_ci : Ap);
            }
            a_ci += BK;                                   // advance this thread's (tap, ci) by one k-tile
            while (a_ci >= cg.ci) { a_ci -= cg.ci; ++a_tap; }
        } else {
            const int col = m0 + cmc;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int kr = k0 + ckr + TC::KROW_STEP * s;
                const bool ok = kr < g.K && col < g.M;
                pmask |= (unsigned)ok << s;
                ra[s] = ld16(ok ? Ap + (long)kr * g.lda + col : Ap);
            }
        }
        // ---- B
        if (BMODE == PDMK_B_ROWK) {
            const int k = k0 + kc;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bool ok = b_off[s] >= 0 && k < g.K;
                pmask |= (unsigned)ok << (4 + s);
                rb[s] = ld16(ok ? Bp + b_off[s] + k : Bp);
            }
        } else if (BMODE == PDMK_B_COLK) {
            const int col = n0 + cmc;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int kr = k0 + ckr + TC::KROW_STEP * s;
                const bool ok = kr < g.K && col < g.N;
                pmask |= (unsigned)ok << (4 + s);
                rb[s] = ld16(ok ? Bp + (long)kr * g.ldb + col : Bp);
            }
        } else {
            const int col = n0 + cmc;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int kr = k0 + ckr + TC::KROW_STEP * s;
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
                pmask |= (unsigned)(px >= 0) << (4 + s);
                rb[s] = ld16(px >= 0 ? Bp + (long)px * cg.ld + b_ci : Bp);
            }
        }
    };

    auto store_tiles = [&](int buf) {
        T* As = reinterpret_cast<T*>(smem[buf][0]);
        T* Bs = reinterpret_cast<T*>(smem[buf][1]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (!((pmask >> s) & 1)) ra[s] = zero4;
            if (!((pmask >> (4 + s)) & 1)) rb[s] = zero4;
        }
        if (AMODE == PDMK_A_COLK) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
                *reinterpret_cast<uint4*>(As + (ckr + TC::KROW_STEP * s) * TC::RS_COLK + cmc) = ra[s];
            if (do_colsum) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const T* v = reinterpret_cast<const T*>(&ra[s]);
#pragma unroll
                    for (int e = 0; e < CH; ++e) csum[e] += to_f32(v[e]);
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) *reinterpret_cast<uint4*>(As + (lr + 32 * s) * TC::RS_ROWK + kc) = ra[s];
        }
        if (BMODE == PDMK_B_ROWK) {
#pragma unroll
            for (int s = 0; s < 4; ++s) *reinterpret_cast<uint4*>(Bs + (lr + 32 * s) * TC::RS_ROWK + kc) = rb[s];
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s)
                *reinterpret_cast<uint4*>(Bs + (ckr + TC::KROW_STEP * s) * TC::RS_COLK + cmc) = rb[s];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_tiles(kt0);
    store_tiles(0);
    __syncthreads();
    int cur = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        const bool more = kt + 1 < kt1;
        if (more) load_tiles(kt + 1);
        const T* As = reinterpret_cast<const T*>(smem[cur][0]);
        const T* Bs = reinterpret_cast<const T*>(smem[cur][1]);
#pragma unroll
        for (int kk = 0; kk < BK; kk += MM::KS) {
            typename MM::frag af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[i] = (AMODE == PDMK_A_COLK) ? MM::load_colk(As, TC::RS_COLK, kk, wm * 64 + i * 16, lane)
                                               : MM::load_rowk(As, TC::RS_ROWK, wm * 64 + i * 16, kk, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                bf[j] = (BMODE != PDMK_B_ROWK) ? MM::load_colk(Bs, TC::RS_COLK, kk, wn * 64 + j * 16, lane)
                                               : MM::load_rowk(Bs, TC::RS_ROWK, wn * 64 + j * 16, kk, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = MM::mma(bf[j], af[i], acc[i][j]);
        }
        if (more) store_tiles(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    if (do_colsum) {      // combine the KROW_STEP k-row groups through LDS (all tile reads are behind the last barrier)
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int e = 0; e < CH; ++e) red[ckr * 128 + cmc + e] = csum[e];
        __syncthreads();
        if (tid < 128 && m0 + tid < g.M) {
            float s = 0.f;
            for (int j = 0; j < TC::KROW_STEP; ++j) s += red[j * 128 + tid];
            unsafeAtomicAdd(g.colsum_out + m0 + tid, s);
        }
    }

    // ---------------------------------------------------------------- epilogue
    const bool first = blockIdx.y == 0;
    const bool vec = ((g.N & 3) == 0) && ((g.ldc & 3) == 0) && (g.R == nullptr || (g.ldr & 3) == 0);
    const bool atomic = gridDim.y > 1;
    float* Cf = reinterpret_cast<float*>(g.C);
    T* Ct = reinterpret_cast<T*>(g.C);
    const T* Rp = reinterpret_cast<const T*>(g.R);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + (lane & 15);
        if (m >= g.M) continue;
        const float* rv = (g.rowvec && first) ? g.rowvec + (long)(m / g.rows_per_b) * g.N : nullptr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nb = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
            if (nb >= g.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * g.alpha;
            const int nv = min(4, g.N - nb);
            if (first) {
                if (vec) {
                    if (g.bias) {
                        const float4 bb = *reinterpret_cast<const float4*>(g.bias + nb);
                        v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
                    }
                    if (rv) {
                        const float4 bb = *reinterpret_cast<const float4*>(rv + nb);
                        v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
                    }
                    if (Rp) {
                        if (sizeof(T) == 2) {
                            const bf16x4 rr = *reinterpret_cast<const bf16x4*>(Rp + (long)m * g.ldr + nb);
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] += (float)rr[r];
                        } else {
                            const float4 rr = *reinterpret_cast<const float4*>(Rp + (long)m * g.ldr + nb);
                            v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
                        }
                    }
                } else {
                    for (int r = 0; r < nv; ++r) {
                        if (g.bias) v[r] += g.bias[nb + r];
                        if (rv) v[r] += rv[nb + r];
                        if (Rp) v[r] += to_f32(Rp[(long)m * g.ldr + nb + r]);
                    }
                }
            }
            const long off = (long)m * g.ldc + nb;
            if (g.out_f32 || sizeof(T) == 4) {
                if (atomic) {
                    for (int r = 0; r < nv; ++r) unsafeAtomicAdd(Cf + off + r, v[r]);
                } else if (vec) {
                    float4 o = make_float4(v[0], v[1], v[2], v[3]);
                    if (g.accumulate) {
                        const float4 c = *reinterpret_cast<const float4*>(Cf + off);
                        o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w;
                    }
                    *reinterpret_cast<float4*>(Cf + off) = o;
                } else {
                    for (int r = 0; r < nv; ++r) Cf[off + r] = v[r] + (g.accumulate ? Cf[off + r] : 0.f);
                }
            } else {
                if (vec) {
                    if (g.accumulate) {
                        const bf16x4 c = *reinterpret_cast<const bf16x4*>(Ct + off);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += (float)c[r];
                    }
                    bf16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (bf16)v[r];
                    *reinterpret_cast<bf16x4*>(Ct + off) = o;
                } else {
                    for (int r = 0; r < nv; ++r)
                        Ct[off + r] = from_f32<T>(v[r] + (g.accumulate ? to_f32(Ct[off + r]) : 0.f));
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
    dim3 grid(mt * nt, g.splitk > 1 ? g.splitk : 1), block(NTHREADS);
    int lg_wo = -1, lg_howo = -1;
    if (g.b_mode == PDMK_B_COLK_CONV) {
        lg_wo = ilog2_exact(g.conv_wo);
        lg_howo = ilog2_exact(g.conv_ho * g.conv_wo);
        if (lg_wo < 0 || lg_howo < 0) lg_wo = lg_howo = -1;
    }
#define PDMK_GO(AM, BMD, CM) hipLaunchKernelGGL((igemm_kernel<T, AM, BMD, CM>), grid, block, 0, st, g, lg_wo, lg_howo)
    if (g.a_mode == PDMK_A_ROWK && g.b_mode == PDMK_B_ROWK) PDMK_GO(PDMK_A_ROWK, PDMK_B_ROWK, 0);
    else if (g.a_mode == PDMK_A_CONV && g.b_mode == PDMK_B_ROWK) {
        switch (g.conv_mode) {
            case 0: PDMK_GO(PDMK_A_CONV, PDMK_B_ROWK, 0); break;
            case 1: PDMK_GO(PDMK_A_CONV, PDMK_B_ROWK, 1); break;
            case 2: PDMK_GO(PDMK_A_CONV, PDMK_B_ROWK, 2); break;
            default: PDMK_GO(PDMK_A_CONV, PDMK_B_ROWK, 3); break;
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

extern "C" int pdmk_gemm(const pdmk_gemm_args* a, pdmk_stream stream) {
    if (!a || !a->A || !a->B || !a->C) return -1;
    const pdmk_gemm_args& g = *a;
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return -1;
    const int ch = g.dtype == PDMK_BF16 ? 8 : 4;
    if (g.dtype != PDMK_BF16 && g.dtype != PDMK_F32) return -2;
    if (g.a_mode != PDMK_A_COLK && (g.K % ch)) return -1;   // reduction-major operands predicate every k-row
    if (((uintptr_t)g.A | (uintptr_t)g.B) & 15) return -1;
    if (g.a_mode == PDMK_A_ROWK && (g.lda % ch)) return -1;
    if (g.a_mode == PDMK_A_COLK && ((g.lda % ch) || (g.M % ch))) return -1;
    if (g.b_mode == PDMK_B_ROWK && (g.ldb % ch)) return -1;
    if (g.b_mode == PDMK_B_COLK && ((g.ldb % ch) || (g.N % ch))) return -1;
    if (g.a_mode == PDMK_A_CONV || g.b_mode == PDMK_B_COLK_CONV) {
        if (g.conv_ci <= 0 || (g.conv_ci % ch) || (g.conv_ld % ch) || g.conv_mode < 0 || g.conv_mode > 3) return -1;
        if (g.conv_b <= 0 || g.conv_hi <= 0 || g.conv_wi <= 0 || g.conv_ho <= 0 || g.conv_wo <= 0) return -1;
        const long px = (long)g.conv_b * g.conv_ho * g.conv_wo;
        if (px >= (1L << 30) || (long)g.conv_b * g.conv_hi * g.conv_wi >= (1L << 30)) return -1;   // 32-bit pixel ids
        if (g.a_mode == PDMK_A_CONV && (g.M != px || g.K != 9 * g.conv_ci)) return -1;
        if (g.b_mode == PDMK_B_COLK_CONV && (g.K != px || g.N != 9 * g.conv_ci)) return -1;
        // gather geometry must be consistent with the source extent (out-of-image taps read as zero padding)
        const int eh = g.conv_mode == 1 ? (g.conv_hi + 1) / 2 : (g.conv_mode >= 2 ? 2 * g.conv_hi : g.conv_hi);
        const int ew = g.conv_mode == 1 ? (g.conv_wi + 1) / 2 : (g.conv_mode >= 2 ? 2 * g.conv_wi : g.conv_wi);
        if (g.conv_ho != eh || g.conv_wo != ew) return -1;
    }
    if (g.splitk > 1 && !(g.out_f32 || g.dtype == PDMK_F32)) return -1;
    if (g.rowvec && g.rows_per_b <= 0) return -1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return g.dtype == PDMK_BF16 ? launch<bf16>(g, st) : launch<float>(g, st);
}

extern "C" int pdmk_version(void) { return 101; }
