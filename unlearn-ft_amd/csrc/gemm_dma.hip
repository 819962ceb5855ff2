// Implicit GEMM, LDS-DMA variant (bf16, A = activation rows or 3x3 gather, B = weights [N][K]): the forward / dgrad
// work-horse.  Same math and epilogue as gemm.hip; what changes is how tiles reach the matrix cores:
//   * operands go global -> LDS directly (buffer_load_dwordx4 ... lds): no staging registers, no ds_write, and
//     out-of-range lanes (conv halo, M/N tails) are zero-filled into LDS by the buffer unit;
//   * a 4-slot ring of 128x32 (A) + 128x32 (B) bf16 tiles (16 KiB per slot, 64 KiB per workgroup -> 2 workgroups/CU,
//     16 waves/CU) keeps THREE K-steps in flight behind a counted s_waitcnt vmcnt(N): HBM / L2 latency is covered by
//     prefetch distance instead of by occupancy; one raw s_barrier per K-step;
//   * LDS image is lane-linear (a DMA wave-instruction writes 1 KiB = 16 rows x 64 B); bank conflicts of the
//     ds_read_b128 fragment reads are removed by XOR-swizzling the 16-byte chunk index with (row>>2)&3 on the SOURCE
//     side (per-lane global address) and on the read side.
#include "common.h"

#ifdef PDMK_STAMPS      // diagnostic build only: per-workgroup phase timestamps (s_memtime), read back by tools/
__device__ unsigned long long pdmk_stamps[8192 * 6];
extern "C" int pdmk_debug_read_stamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(pdmk_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#define PDMK_STAMP(i)                                                                                    \
    do {                                                                                                 \
        if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 8192)                                    \
            pdmk_stamps[blockIdx.x * 6 + (i)] = (i) >= 4 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define PDMK_STAMP(i)
#endif

namespace pdmk_dma {

constexpr int BN = 128, BK = 32, NT = 512;
constexpr int TILE_BYTES = 128 * BK * 2;               // 8 KiB per 128 tile rows per slot
constexpr unsigned OOB = 0x80000000u;
typedef __attribute__((address_space(3))) void lds_void;

struct ConvGeom {
    int hi, wi, ci, ho, wo, ld;
};

template <int CMODE>
__device__ __forceinline__ int conv_src_pixel(const ConvGeom& g, int b, int oy, int ox, int tap) {
    const int ky = (tap * 11) >> 5;
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

template <int N> __device__ __forceinline__ void wait_vmcnt();
template <> __device__ __forceinline__ void wait_vmcnt<0>() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<2>() { asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<3>() { asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<4>() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
template <> __device__ __forceinline__ void wait_vmcnt<6>() { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }

// BM = 128: 4-slot ring of 16 KiB (2 workgroups/CU, 32x64 outputs per wave);  BM = 256: 3-slot ring of 24 KiB (A 256x32 +
// B 128x32; 72 KiB -> still 2 workgroups/CU, 64x64 per wave): 33 % fewer operand bytes per FLOP into the CU, which is
// what bounds these kernels (~30 B/cycle/CU of L2->LDS intake), and twice the MFMAs per barrier.
template <bool CONV, int CMODE, int BM>
__global__ __launch_bounds__(NT, 4) void igemm_dma_kernel(pdmk_gemm_args g, unsigned a_bytes, unsigned b_bytes) {
    typedef Mma<bf16> MM;
    constexpr int AR = BM / 128;                           // A row-blocks of 128 per tile = A DMA instructions per stage
    constexpr int STAGES = BM == 128 ? 4 : 3;
    constexpr int SLOT_BYTES = (AR + 1) * TILE_BYTES;
    constexpr int IM = BM / 64;                            // 16-row MFMA tiles per wave (4 x 2 waves)
    constexpr int NDMA = AR + 1;                           // DMA instructions per thread and stage
    __shared__ __attribute__((aligned(1024))) unsigned char smem[STAGES * SLOT_BYTES];

    PDMK_STAMP(0);
    PDMK_STAMP(5);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;                 // 4 x 2 waves, (BM/4) x 64 outputs each
    const int ntn = (g.N + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;
    const int nk_total = g.K / BK;
    const int per = (nk_total + gridDim.y - 1) / gridDim.y;
    const int kt0 = blockIdx.y * per;
    const int kt1 = min(nk_total, kt0 + per);
    if (kt0 >= kt1 && g.accumulate != 2) return;        // slab split-K: an empty split still writes its (zero) slab

    const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), (short)0, (int)a_bytes, 0x00020000);
    const auto rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.B), (short)0, (int)b_bytes, 0x00020000);
    const ConvGeom cg{g.conv_hi, g.conv_wi, g.conv_ci, g.conv_ho, g.conv_wo, g.conv_ld};

    // ---- loader state: thread -> (tile row = tid>>2, physical 16-byte chunk = tid&3); logical chunk = phys ^ swizzle
    const int lrow = tid >> 2;
    const int lchunk = ((tid & 3) ^ ((lrow >> 2) & 3)) * 8;          // element offset inside the 32-wide k-tile
    unsigned a_base[AR], b_base = OOB;
    int a_b[AR], a_oy[AR], a_ox[AR];
#pragma unroll
    for (int s = 0; s < AR; ++s) {
        a_base[s] = OOB; a_b[s] = -1; a_oy[s] = 0; a_ox[s] = 0;
        const int m = m0 + s * 128 + lrow;
        if (m < g.M) {
            if (CONV) {
                const int hw = cg.ho * cg.wo;
                a_b[s] = m / hw;
                const int rem = m - a_b[s] * hw;
                a_oy[s] = rem / cg.wo;
                a_ox[s] = rem - a_oy[s] * cg.wo;
            } else {
                a_base[s] = (unsigned)m * (unsigned)g.lda * 2u;
            }
        }
    }
    {
        const int n = n0 + lrow;
        if (n < g.N) b_base = (unsigned)n * (unsigned)g.ldb * 2u;
    }
    int tap = 0, ci0 = 0;                                            // conv: block-uniform (tap, ci) of the NEXT tile to issue
    if (CONV) {
        const int k = kt0 * BK;
        tap = k / cg.ci;
        ci0 = k - tap * cg.ci;
    }
    const unsigned lds_wave = (unsigned)wave * 1024u;               // this wave's 1 KiB inside an 8 KiB tile

    auto issue = [&](int kt, int slot) {
        unsigned char* sa = smem + slot * SLOT_BYTES;
#pragma unroll
        for (int s = 0; s < AR; ++s) {
            unsigned va;
            if (CONV) {
                const int px = a_b[s] >= 0 ? conv_src_pixel<CMODE>(cg, a_b[s], a_oy[s], a_ox[s], tap) : -1;
                va = px >= 0 ? ((unsigned)px * (unsigned)cg.ld + (unsigned)(ci0 + lchunk)) * 2u : OOB;
            } else {
                va = a_base[s] != OOB ? a_base[s] + (unsigned)(kt * BK + lchunk) * 2u : OOB;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void*)(sa + s * TILE_BYTES + lds_wave), 16, (int)va, 0, 0, 0);
        }
        if (CONV) {
            ci0 += BK;
            if (ci0 >= cg.ci) { ci0 = 0; ++tap; }
        }
        const unsigned vb = b_base != OOB ? b_base + (unsigned)(kt * BK + lchunk) * 2u : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_void*)(sa + AR * TILE_BYTES + lds_wave), 16, (int)vb, 0, 0, 0);
    };

    f32x4 acc[IM][4];
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addresses: row r = lane&15 of a 16-row MFMA tile, logical chunk g = lane>>4 -> physical g ^ ((r>>2)&3)
    const int fr = lane & 15;
    const int fchunk = ((lane >> 4) ^ ((fr >> 2) & 3)) * 16;         // byte offset inside the 64-byte row
    const unsigned a_foff = (unsigned)(wm * (16 * IM) + fr) * 64u + fchunk;
    const unsigned b_foff = (unsigned)(wn * 64 + fr) * 64u + fchunk;

#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (kt0 + s < kt1) issue(kt0 + s, s);

    int slot = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        const int ahead = min(STAGES - 2, kt1 - 1 - kt);            // younger K-steps already in flight (NDMA each)
        if (ahead >= 2) wait_vmcnt<2 * NDMA>();
        else if (ahead == 1) wait_vmcnt<NDMA>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
#ifdef PDMK_STAMPS
        if (kt == kt0) PDMK_STAMP(1);
#endif
        if (kt + STAGES - 1 < kt1) issue(kt + STAGES - 1, (slot + STAGES - 1) % STAGES);
        const unsigned char* sa = smem + slot * SLOT_BYTES;
        const unsigned char* sb = sa + AR * TILE_BYTES;
        bf16x8 af[IM], bf[4];
#pragma unroll
        for (int i = 0; i < IM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sa + a_foff + i * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(sb + b_foff + j * 1024);
#pragma unroll
        for (int i = 0; i < IM; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = MM::mma(bf[j], af[i], acc[i][j]);
        slot = (slot + 1) % STAGES;
    }

    // ---------------------------------------------------------------- epilogue (same contract as gemm.hip)
    PDMK_STAMP(2);
    const bool first = blockIdx.y == 0;
    const bool slab = g.accumulate == 2;                 // split-K partials to slab blockIdx.y of a [splitk][M][ldc] workspace
    const bool atomic = gridDim.y > 1 && !slab;
    const bool acc1 = g.accumulate == 1;
    const bool f32out = g.out_f32 != 0;
    const bool vec8 = ((g.N & 7) == 0) && ((g.ldc & 7) == 0) && (g.R == nullptr || (g.ldr & 7) == 0);
    constexpr int SROW = 132;
    float* stage = reinterpret_cast<float*>(smem);
    static_assert(64 * SROW * 4 <= STAGES * SLOT_BYTES, "staging image must fit the ring");
    static_assert(2 * NDMA <= 4 || NDMA == 3, "wait_vmcnt instantiations");
    float* Cf = reinterpret_cast<float*>(g.C) + (slab ? (long)blockIdx.y * g.M * g.ldc : 0L);
    bf16* Ct = reinterpret_cast<bf16*>(g.C);
    const bf16* Rp = reinterpret_cast<const bf16*>(g.R);
#pragma unroll
    for (int pass = 0; pass < BM / 64; ++pass) {
        __syncthreads();
        if ((wm * 16 * IM) / 64 == pass) {
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
            for (int rr = 0; rr < 8; ++rr) {
                const int lr2 = wave * 8 + rr, m = m0 + pass * 64 + lr2;
                if (m >= g.M) break;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int n = n0 + h * 64 + lane;
                    if (n < g.N) {
                        float v = stage[lr2 * SROW + h * 64 + lane];
                        if (first) {
                            if (g.bias) v += g.bias[n];
                            if (g.rowvec) v += g.rowvec[(long)(m / g.rows_per_b) * (g.ldrv ? g.ldrv : g.N) + n];
                            if (Rp) v += (float)Rp[(long)m * g.ldr + n];
                        }
                        unsafeAtomicAdd(Cf + (long)m * g.ldc + n, v);
                    }
                }
            }
        } else {
            const int lr2 = tid >> 3, seg = (tid & 7) * 16;
            const int m = m0 + pass * 64 + lr2;
            if (m < g.M) {
                const float* rv = g.rowvec ? g.rowvec + (long)(m / g.rows_per_b) * (g.ldrv ? g.ldrv : g.N) : nullptr;
#pragma unroll
                for (int c8 = 0; c8 < 2; ++c8) {
                    const int n = n0 + seg + c8 * 8;
                    if (n >= g.N) break;
                    float v[8];
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + lr2 * SROW + seg + c8 * 8);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + lr2 * SROW + seg + c8 * 8 + 4);
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
                            const bf16x8 rr = *reinterpret_cast<const bf16x8*>(Rp + (long)m * g.ldr + n);
#pragma unroll
                            for (int r = 0; r < 8; ++r) v[r] += (float)rr[r];
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
                            if (Rp) x += (float)Rp[(long)m * g.ldr + n + r];
                            if (f32out) Cf[off + r] = x + (acc1 ? Cf[off + r] : 0.f);
                            else Ct[off + r] = (bf16)(x + (acc1 ? (float)Ct[off + r] : 0.f));
                        }
                    }
                }
            }
        }
    }
    PDMK_STAMP(3);
    PDMK_STAMP(4);
}

}  // namespace pdmk_dma

// called by pdmk_gemm (gemm.hip) after argument validation; returns 1 if the shape is not handled here
int pdmk_gemm_dma_launch(const pdmk_gemm_args& g, hipStream_t st, long a_bytes, long b_bytes) {
    using namespace pdmk_dma;
    if (g.dtype != PDMK_BF16 || g.b_mode != PDMK_B_ROWK || g.a_mode == PDMK_A_COLK) return 1;
    if ((g.K % BK) || (g.a_mode == PDMK_A_CONV && g.conv_mode > 3)) return 1;   // mode 4: ring / K-step-32 register kernels only
    const int nt = (g.N + BN - 1) / BN;
    // 256-row tiles when they still give every CU its two workgroups' worth of blocks (the 64x64-latent layers)
    const bool big = (long)((g.M + 255) / 256) * nt * (g.splitk > 1 ? g.splitk : 1) >= 320;
    const int mt = big ? (g.M + 255) / 256 : (g.M + 127) / 128;
    dim3 grid(mt * nt, g.splitk > 1 ? g.splitk : 1);
#define PDMK_DMA_GO(CV, CM)                                                                                           \
    do {                                                                                                              \
        if (big) hipLaunchKernelGGL((igemm_dma_kernel<CV, CM, 256>), grid, dim3(NT), 0, st, g, (unsigned)a_bytes,    \
                                    (unsigned)b_bytes);                                                               \
        else hipLaunchKernelGGL((igemm_dma_kernel<CV, CM, 128>), grid, dim3(NT), 0, st, g, (unsigned)a_bytes,        \
                                (unsigned)b_bytes);                                                                   \
    } while (0)
    if (g.a_mode == PDMK_A_ROWK) PDMK_DMA_GO(false, 0);
    else {
        switch (g.conv_mode) {
            case 0: PDMK_DMA_GO(true, 0); break;
            case 1: PDMK_DMA_GO(true, 1); break;
            case 2: PDMK_DMA_GO(true, 2); break;
            default: PDMK_DMA_GO(true, 3); break;
        }
    }
#undef PDMK_DMA_GO
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}
