// Implicit GEMM on the CDNA4 matrix cores: Linear / Conv3x3 (NHWC) forward, dgrad and wgrad in one template.
//   C[M,N] (+)= alpha * sum_k A(m,k) B(n,k) + bias[n] + rowvec[m/rpb][n] + R[m][n]
// Workgroup = 256 threads = 4 waves (2x2), tile 128x128, K-step 32 (bf16) / 16 (fp32); each wave owns a 64x64
// sub-tile as 4x4 MFMA 16x16 accumulators.  Operands are staged global -> registers -> LDS (double buffered, one
// barrier per K-step; the next tile's global loads are issued before the current tile's MFMAs so HBM/L2 latency
// hides under them).  The register hop is what makes zero-filled halos (conv padding), tails and the four gather
// geometries free.  Reduction-major operands (wgrad) stay in memory order in LDS and are transposed by
// ds_read_b64_tr_b16 on the way into the MFMA fragments.  The MFMA is issued with operands swapped (D = B.A^T) so a
// lane ends up with 4 consecutive n of one output row: 8/16-byte epilogue loads and stores.
// Blocks are remapped so that the n-tiles of one m-tile run back to back on one XCD (shared activation panel in L2).
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, NTHREADS = 256;
constexpr int OPERAND_BYTES = 12288;

template <typename T> struct TileCfg {
    static constexpr int CH = Mma<T>::CH;
    static constexpr int BK = 4 * CH;
    static constexpr int RS_ROWK = BK + (sizeof(T) == 2 ? 16 : 4);
    static constexpr int RS_COLK = 128 + 16;
    static constexpr int CPR = 128 / CH;          // chunks per 128-wide row (colk layouts)
    static constexpr int KROW_STEP = NTHREADS / CPR;
};

struct ConvGeom {
    int hi, wi, ci, ho, wo, mode, ld;
};

// source pixel offset (in pixels) for output pixel (b,oy,ox) and tap; returns -1 when the tap reads padding
__device__ __forceinline__ long conv_src_pixel(const ConvGeom& g, int b, int oy, int ox, int tap) {
    const int ky = tap / 3, kx = tap - 3 * ky;
    int iy, ix;
    if (g.mode == 1) {
        iy = 2 * oy + ky - 1;
        ix = 2 * ox + kx - 1;
        if (iy < 0 || ix < 0 || iy >= g.hi || ix >= g.wi) return -1;
    } else {
        const int vy = oy + ky - 1, vx = ox + kx - 1;
        if (vy < 0 || vx < 0) return -1;
        if (g.mode == 0) {
            iy = vy; ix = vx;
        } else if (g.mode == 2) {
            iy = vy >> 1; ix = vx >> 1;
        } else {
            if ((vy | vx) & 1) return -1;
            iy = vy >> 1; ix = vx >> 1;
        }
        if (iy >= g.hi || ix >= g.wi) return -1;
    }
    return ((long)b * g.hi + iy) * g.wi + ix;
}

__device__ __forceinline__ uint4 ld16(const void* p) { return *reinterpret_cast<const uint4*>(p); }

template <typename T, int AMODE, int BMODE>
__global__ __launch_bounds__(NTHREADS) void igemm_kernel(pdmk_gemm_args g) {
    typedef TileCfg<T> TC;
    typedef Mma<T> MM;
    constexpr int CH = TC::CH, BK = TC::BK;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][OPERAND_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (g.N + BN - 1) / BN;
    const int nblk = gridDim.x;
    const int tile = xcd_remap(blockIdx.x, nblk);
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;

    const int nk_total = (g.K + BK - 1) / BK;
    const int per = (nk_total + gridDim.y - 1) / gridDim.y;
    const int kt0 = blockIdx.y * per;
    const int kt1 = min(nk_total, kt0 + per);
    if (kt0 >= kt1) return;

    const T* __restrict__ Ap = reinterpret_cast<const T*>(g.A);
    const T* __restrict__ Bp = reinterpret_cast<const T*>(g.B);
    ConvGeom cg{g.conv_hi, g.conv_wi, g.conv_ci, g.conv_ho, g.conv_wo, g.conv_mode, g.conv_ld};

    // ---------------------------------------------------------------- per-thread loader state
    // rowk-type operands: two rows (r, r+64), fixed k-chunk kc
    const int lr = tid >> 2, kc = (tid & 3) * CH;
    // colk-type operands: two k-rows (kr, kr+KROW_STEP), fixed column chunk
    const int ckr = tid / TC::CPR, cmc = (tid % TC::CPR) * CH;

    const T* a_row[2] = {nullptr, nullptr};
    int a_b[2] = {0, 0}, a_oy[2] = {0, 0}, a_ox[2] = {0, 0};
    bool a_ok[2] = {false, false};
    if (AMODE == PDMK_A_ROWK) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int m = m0 + lr + 64 * s;
            a_ok[s] = m < g.M;
            a_row[s] = Ap + (long)(a_ok[s] ? m : 0) * g.lda;
        }
    } else if (AMODE == PDMK_A_CONV) {
        const int hw = cg.ho * cg.wo;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int m = m0 + lr + 64 * s;
            a_ok[s] = m < g.M;
            const int mm = a_ok[s] ? m : 0;
            a_b[s] = mm / hw;
            const int rem = mm - a_b[s] * hw;
            a_oy[s] = rem / cg.wo;
            a_ox[s] = rem - a_oy[s] * cg.wo;
        }
    }
    const T* b_row[2] = {nullptr, nullptr};
    bool b_ok[2] = {false, false};
    int b_tap = 0, b_ci = 0;
    if (BMODE == PDMK_B_ROWK) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int n = n0 + lr + 64 * s;
            b_ok[s] = n < g.N;
            b_row[s] = Bp + (long)(b_ok[s] ? n : 0) * g.ldb;
        }
    } else if (BMODE == PDMK_B_COLK_CONV) {
        const int col = n0 + cmc;
        b_tap = col / cg.ci;
        b_ci = col - b_tap * cg.ci;
    }

    // Loads are always issued from a valid address (the operand base when predicated off) and the zero-fill is applied
    // when the registers are written to LDS, AFTER the MFMAs of the current tile: a data select right behind the load
    // would make the compiler wait for the load (vmcnt(0)) before the MFMAs and expose the full memory latency.
    uint4 ra[2], rb[2];
    bool pa[2], pb[2];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);

    auto load_tiles = [&](int kt) {
        const int k0 = kt * BK;
        // ---- A
        if (AMODE == PDMK_A_ROWK) {
            const int k = k0 + kc;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                pa[s] = a_ok[s] && k < g.K;
                ra[s] = ld16(pa[s] ? a_row[s] + k : Ap);
            }
        } else if (AMODE == PDMK_A_CONV) {
            const int k = k0 + kc;
            const int tap = k / cg.ci, ci = k - tap * cg.ci;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const long px = (a_ok[s] && k < g.K) ? conv_src_pixel(cg, a_b[s], a_oy[s], a_ox[s], tap) : -1;
                pa[s] = px >= 0;
                ra[s] = ld16(pa[s] ? Ap + px * cg.ld + ci : Ap);
            }
        } else {
            const int col = m0 + cmc;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int kr = k0 + ckr + TC::KROW_STEP * s;
                pa[s] = kr < g.K && col < g.M;
                ra[s] = ld16(pa[s] ? Ap + (long)kr * g.lda + col : Ap);
            }
        }
        // ---- B
        if (BMODE == PDMK_B_ROWK) {
            const int k = k0 + kc;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                pb[s] = b_ok[s] && k < g.K;
                rb[s] = ld16(pb[s] ? b_row[s] + k : Bp);
            }
        } else if (BMODE == PDMK_B_COLK) {
            const int col = n0 + cmc;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int kr = k0 + ckr + TC::KROW_STEP * s;
                pb[s] = kr < g.K && col < g.N;
                rb[s] = ld16(pb[s] ? Bp + (long)kr * g.ldb + col : Bp);
            }
        } else {
            const int col = n0 + cmc;
            const int hw = cg.ho * cg.wo;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int kr = k0 + ckr + TC::KROW_STEP * s;
                long px = -1;
                if (kr < g.K && col < g.N) {
                    const int b = kr / hw;
                    const int rem = kr - b * hw;
                    const int oy = rem / cg.wo, ox = rem - oy * cg.wo;
                    px = conv_src_pixel(cg, b, oy, ox, b_tap);
                }
                pb[s] = px >= 0;
                rb[s] = ld16(pb[s] ? Bp + px * cg.ld + b_ci : Bp);
            }
        }
    };

    auto store_tiles = [&](int buf) {
        T* As = reinterpret_cast<T*>(smem[buf][0]);
        T* Bs = reinterpret_cast<T*>(smem[buf][1]);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (!pa[s]) ra[s] = zero4;
            if (!pb[s]) rb[s] = zero4;
        }
        if (AMODE == PDMK_A_COLK) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
                *reinterpret_cast<uint4*>(As + (ckr + TC::KROW_STEP * s) * TC::RS_COLK + cmc) = ra[s];
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s) *reinterpret_cast<uint4*>(As + (lr + 64 * s) * TC::RS_ROWK + kc) = ra[s];
        }
        if (BMODE == PDMK_B_ROWK) {
#pragma unroll
            for (int s = 0; s < 2; ++s) *reinterpret_cast<uint4*>(Bs + (lr + 64 * s) * TC::RS_ROWK + kc) = rb[s];
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s)
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
                for (int r = 0; r < nv; ++r) {
                    if (g.bias) v[r] += g.bias[nb + r];
                    if (rv) v[r] += rv[nb + r];
                    if (Rp) v[r] += to_f32(Rp[(long)m * g.ldr + nb + r]);
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

template <typename T> int launch(const pdmk_gemm_args& g, hipStream_t st) {
    const int mt = (g.M + BM - 1) / BM, nt = (g.N + BN - 1) / BN;
    dim3 grid(mt * nt, g.splitk > 1 ? g.splitk : 1), block(NTHREADS);
#define PDMK_GO(AM, BMD) hipLaunchKernelGGL((igemm_kernel<T, AM, BMD>), grid, block, 0, st, g)
    if (g.a_mode == PDMK_A_ROWK && g.b_mode == PDMK_B_ROWK) PDMK_GO(PDMK_A_ROWK, PDMK_B_ROWK);
    else if (g.a_mode == PDMK_A_CONV && g.b_mode == PDMK_B_ROWK) PDMK_GO(PDMK_A_CONV, PDMK_B_ROWK);
    else if (g.a_mode == PDMK_A_COLK && g.b_mode == PDMK_B_COLK) PDMK_GO(PDMK_A_COLK, PDMK_B_COLK);
    else if (g.a_mode == PDMK_A_COLK && g.b_mode == PDMK_B_COLK_CONV) PDMK_GO(PDMK_A_COLK, PDMK_B_COLK_CONV);
    else return -2;
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

extern "C" int pdmk_version(void) { return 100; }
