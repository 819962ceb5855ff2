// Row-block Linear kernel (bf16, A = activation rows [M][K], B = weights [N][K], K <= 640): forward / dgrad of the attention and
// feed-forward projections of the 64x64 / 32x32 latent levels (blocks.py:203-295, 44-76), where M = 8192..32768 rows meet
// K = 320 / 640.  The ring kernel (gemm_ring.hip) gives every BM x BN output tile its own workgroup: with 5-10 K-steps per
// tile each workgroup is a cold start (first DMA from HBM), a short loop and an epilogue that do not overlap, the A rows
// of a row block are fetched once per n-tile (through different XCDs' L2s), and the L2 -> LDS intake per FLOP is
// (1/BM + 1/BN).  Here a workgroup OWNS a block of BM rows for the whole launch:
//   * its A rows x all of K are LDS-DMA-ed once and then live in REGISTERS as MFMA fragments (BM = 128, K = 320:
//     20 x bf16x8 = 80 VGPRs per lane), so A is read from HBM exactly once per launch and never again from LDS;
//   * it walks the n-tiles of its column group while the weight tiles [BN][64] stream through ONE ring of BST slots as a
//     continuous sequence of (n-tile, K-step) stages: the ring never drains between n-tiles, the epilogue of n-tile j runs
//     while the stages of n-tile j+1 are in flight, and the intake per FLOP is 1/BM (weights only);
//   * the epilogue is wave-private (no workgroup barrier): accumulators (+ bias from an LDS copy) -> a 16-row fp32 staging
//     image of the wave -> 16-byte row-contiguous stores, with the residual / previous output PREFETCHED one n-tile ahead
//     by buffer loads the compiler does not see (inline asm), so that no compiler-inserted s_waitcnt vmcnt(0) drains the
//     weight stream; every vector-memory instruction the wave issues is counted (`issued`), each ring slot remembers the
//     count at its issue (`mark`), and the wait for a slot is the exact s_waitcnt vmcnt(issued - mark).  Stores and
//     prefetches are buffer instructions whose inactive lanes point out of range (dropped by the buffer unit), never
//     branches, so the counts are exact;
//   * the K-step loop is branch-light (a scalar compare + branch costs ~40 cycles against 640 cycles of MFMA work per step
//     and SIMD): every wave issues the same number of DMA pieces per stage (surplus ones read out of range into a dump
//     area), stages past the last one are zero-filled instead of skipped, the run-time vmcnt goes through a computed jump
//     (common.h), and the kk = 0 fragments of a stage are read during the step before (BST slots carry BST-2 stages in
//     flight: a stage stays readable for two steps).
// What binds it (in-kernel timestamps): ~480 cycles of wait + barrier rendezvous per step next to those 640,
// ~1700 cycles of epilogue per n-tile, ~3.5 us of prologue (A rows from cold HBM).  It matches or beats the ring kernel by
// 0-20 % on the M = 32768, K = 320 shapes with N >= 640, reads A exactly once (PMC), and is bit-identical to it.
// Fused epilogues: bias, residual OR accumulate (bf16), PDMK_EPI_GEGLU (with the optional pre-activation copy).
#include "common.h"

#include <stdio.h>
#include <stdlib.h>

#ifndef PDMK_RB_STORE_AUX
#define PDMK_RB_STORE_AUX (PDMK_NT_STORES ? 2 : 0)   // cache policy of the output stores: 2 = nt (streaming; common.h st_stream), 16 = sc1 (write-through: measured slower)
#endif
#ifndef PDMK_RB_DEFER_GEGLU
#define PDMK_RB_DEFER_GEGLU 0   // 1: MODE 2 parks bf16 accumulators and defers its (VALU-heavy, ~8000 cycles per n-tile) epilogue instead of pipelining fragments: measured +-0 (107.3 vs 105.8 us at N = 2560)
#endif
#ifndef PDMK_RB_DEFER
#define PDMK_RB_DEFER 0      // deferred epilogue halves: 40 more live registers (spills beside the pipelined fragments) for ~5 %
#endif

namespace pdmk_rb {

constexpr int BK = 64, NT = 512;
constexpr unsigned OOB = 0x80000000u;
typedef __attribute__((address_space(3))) void lds_void;
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wait_vmcnt_dyn(int n) { pdmk_wait_vmcnt(n); }   // n is wave-uniform

#ifdef PDMK_RB_STAMPS   // diagnostic build only: per-workgroup phase timestamps of wave 0, read back by tools/rowblock_stamps.py
__device__ unsigned long long pdmk_rb_stamps[1024 * 24];
#define RB_STAMP(i)                                                                                               \
    do {                                                                                                          \
        if (wave == 0 && blockIdx.x < 1024) {                                                                     \
            const unsigned long long t__ = (i) >= 22 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); \
            if (lane == 0) pdmk_rb_stamps[blockIdx.x * 24 + (i)] = t__;                                           \
        }                                                                                                         \
    } while (0)
#else
#define RB_STAMP(i)
#endif

struct RbArgs {
    unsigned a_bytes, b_bytes, c_bytes, r_bytes, c2_bytes, ln_bytes, st_bytes;
    int ngrp;                     // column groups: workgroup (rb, grp) walks n-tiles [grp * per, (grp + 1) * per)
};

__device__ __forceinline__ i32x4 rsrc_words(const void* p, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    return i32x4{(int)(a & 0xffffffffull), (int)((a >> 32) & 0xffffull), (int)bytes, 0x00020000};
}

// MODE: 0 = bias only, 1 = residual or accumulate (prefetched addend), 2 = PDMK_EPI_GEGLU
// LNP: LayerNorm of the row block in the prologue (pdmk_gemm_args.ln_gamma) - its own instantiation, so that the plain kernels keep
// their register allocation (the prologue code in the same body cost them 15 spilled VGPRs)
template <int BM, int NJ, int NKA, int BST, int MODE, bool LNP = false>
__global__ __launch_bounds__(NT, 2) void rowblock_kernel(pdmk_gemm_args g, RbArgs x) {
    typedef Mma<bf16> MM;
    constexpr int BN = 32 * NJ, IM = BM / 64;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    constexpr int NA = BM / 64;                               // A DMA pieces per wave and K-step
    constexpr int NBLK_B = BN / 8, NB = (NBLK_B + 7) / 8;     // weight pieces per stage / per wave (max)
    constexpr int WCOLS = 16 * NJ;                            // columns of an n-tile owned by a wave (waves: 4 in M x 2 in N)
    constexpr int SROW = WCOLS + 4;                           // staging row pitch (floats)
    constexpr int STG_WAVE = 16 * SROW * 4;
    constexpr int BIAS_MAX = 4096;
    constexpr int RING = BST * B_BYTES, A_IMG = NKA * A_BYTES;
    constexpr int BODY = RING + 8 * STG_WAVE;                 // ring + staging; the A image overlays its END during the prologue
    constexpr int SPAN = BODY > A_IMG ? BODY : A_IMG;
    constexpr int A_OFF = SPAN - A_IMG;
    constexpr int DEP = BST - 2;                               // weight stages in flight ahead of the one being read by waves 0-3
    constexpr int PRO = (A_OFF / B_BYTES) < DEP ? (A_OFF / B_BYTES) : DEP;   // stages issued beside the A image
    constexpr int DUMP_OFF = SPAN + BIAS_MAX * 4;              // 1 KiB that surplus DMA pieces (zeros) land in
    constexpr int TOTAL = DUMP_OFF + 1024;
    static_assert(TOTAL <= 160 * 1024, "LDS");
    static_assert((A_OFF % 1024) == 0 && (RING % 1024) == 0, "DMA pieces are 1 KiB");
    static_assert(BST >= 3 && BST <= 6, "ring depth");
    constexpr int C8W = 2 * NJ;                               // 8-column chunks per staged row
    constexpr int ITEMS = (16 * C8W + 63) / 64;               // (row, chunk) items per lane and 16-row pass
    constexpr int ITEMS2 = (16 * NJ + 63) / 64;               // GEGLU: (row, 16-column pair) items
    __shared__ __attribute__((aligned(1024))) unsigned char smem[TOTAL];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // scalar: every count derived from it stays in SGPRs
    const int wm = wave >> 1, wn = wave & 1;
    // waves w and w + 4 share a SIMD.  (Running them half a step apart - waves 4-7 multiplying stage q-1 while waves 0-3 issue
    // their DMA pieces - was built and measured SLOWER: the two groups' epilogues then stall each other at two barriers per
    // n-tile.)  A stage stays in its slot for two steps: BST slots carry BST-2 stages in flight
    const bool early = ((wave ^ (wave >> 2)) & 1) == 0;        // DMA pieces in the first / second half of a step's MFMA stream
    const int ntn = (g.N + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int rb = tile / x.ngrp, grp = tile - rb * x.ngrp;
    const int per = (ntn + x.ngrp - 1) / x.ngrp;
    const int t0 = grp * per, t1 = min(ntn, t0 + per);
    if (t0 >= t1) return;
    RB_STAMP(0);
    RB_STAMP(22);
    const int m0 = rb * BM;
    const int Q = (t1 - t0) * NKA;                            // weight stages: always NKA per n-tile (zero-filled past K)

    unsigned char* const ring = smem;
    float* const stage = reinterpret_cast<float*>(smem + RING) + wave * (16 * SROW);
    float* const sbias = reinterpret_cast<float*>(smem + SPAN);

    const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.A), (short)0, (int)x.a_bytes, 0x00020000);
    const auto rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.B), (short)0, (int)x.b_bytes, 0x00020000);
    const auto rsrcC = __builtin_amdgcn_make_buffer_rsrc(g.C, (short)0, (int)x.c_bytes, 0x00020000);
    const auto rsrcC2 = __builtin_amdgcn_make_buffer_rsrc(g.C2 ? g.C2 : g.C, (short)0, (int)(g.C2 ? x.c2_bytes : 0u), 0x00020000);
    constexpr bool addend = MODE == 1;                        // g.R or g.accumulate == 1, never both (host)
    const i32x4 rsrcR = rsrc_words(g.R ? g.R : g.C, g.R ? x.r_bytes : x.c_bytes);
    const int ldadd = g.R ? g.ldr : g.ldc;
    constexpr bool geglu = MODE == 2;

    // ---- loader lane map (as in the ring kernel): lane -> (sub-row sr of an 8-row piece, physical 16-byte chunk lane&7)
    const int sr = lane >> 3;
    const int lc = (lane & 7) ^ (((wave & 1) * 4) + (sr >> 1));      // logical chunk: k offset lc*8 inside the K-step

    int issued = 0;                                                  // vector-memory instructions this wave has issued
    // ---- A image: NKA K-steps of the BM rows (zero-filled past K and past M)
#pragma unroll
    for (int kt = 0; kt < NKA; ++kt) {
        const int kl = kt * BK + lc * 8;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = m0 + (i * 8 + wave) * 8 + sr;
            const unsigned va = (m < g.M && kl < g.K) ? ((unsigned)m * (unsigned)g.lda + (unsigned)kl) * 2u : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_void*)(smem + A_OFF + kt * A_BYTES + (i * 8 + wave) * 1024), 16,
                                                     (int)va, 0, 0, 0);
        }
    }
    issued += NA * NKA;
    const int markA = issued;
    // LayerNorm prologue: gamma | beta (KP floats each, zero past K) requested NOW, one or two floats per thread, so that their
    // round trip hides behind the A image's; they go to LDS once the A image is dead (one ds_read per use instead of a dependent
    // global load per K-chunk: ten serialised L2 round trips made the fused launch no faster than LayerNorm + GEMM)
    constexpr int KP = NKA * BK, LNW = (2 * KP + NT - 1) / NT;
    float lnv[LNW];
    if constexpr (LNP) {
#pragma unroll
        for (int j = 0; j < LNW; ++j) {
            const int idx = tid + NT * j, kk_ = idx < KP ? idx : idx - KP;
            lnv[j] = (idx < 2 * KP && kk_ < g.K) ? (idx < KP ? g.ln_gamma[kk_] : g.ln_beta[kk_]) : 0.f;
        }
    }

    // ---- weight stream: stage qi = (n-tile ti, K-step ki) into ring slot si.  mk[] = `issued` right after the issue of
    // the (at most BST-1) stages in flight, oldest first: a shift register with constant indices only (a slot-indexed
    // array ends up in scratch memory, whose loads are vector-memory instructions themselves)
    int mk[DEP + 1];
#pragma unroll
    for (int s = 0; s <= DEP; ++s) mk[s] = 0;
    int qi = 0, ti = t0, ki = 0, si = 0;
    // Branch-free (a scalar compare + branch costs ~40 cycles here, and a K-step has ~640 cycles of MFMA work per SIMD): every
    // wave issues exactly NB pieces per stage; a piece past the wave's share, or of a stage past the last one, reads out of
    // range (zeros, no memory traffic) and lands in a 1 KiB dump area / a ring slot nobody reads any more
    auto issue_b = [&]() __attribute__((always_inline)) {
        unsigned char* sb = ring + si * B_BYTES;
        const int kl = ki * BK + lc * 8;
        const bool kok = kl < g.K && qi < Q;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const bool mine = i * 8 + wave < NBLK_B;                 // wave-uniform
            const int n = ti * BN + (i * 8 + wave) * 8 + sr;
            const unsigned vb = (mine && n < g.N && kok) ? ((unsigned)n * (unsigned)g.ldb + (unsigned)kl) * 2u : OOB;
            unsigned char* dst = mine ? sb + (i * 8 + wave) * 1024 : smem + DUMP_OFF;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_void*)dst, 16, (int)vb, 0, 0, 0);
        }
        issued += NB;
        ++qi;
        ki = ki + 1 == NKA ? 0 : ki + 1;
        ti += ki == 0 ? 1 : 0;
        si = si + 1 == BST ? 0 : si + 1;
    };
#pragma unroll
    for (int s = 0; s < PRO; ++s) { issue_b(); mk[s] = issued; }

    // bias -> LDS (fp32), columns of the whole GEMM (N <= BIAS_MAX - BN, host)
    if (g.bias)
        for (int n = tid; n < g.N; n += NT) sbias[n] = g.bias[n];

    wait_vmcnt_dyn(issued - markA);
    __builtin_amdgcn_s_barrier();
    RB_STAMP(2);

    // ---- A fragments -> registers: row fr = lane&15 of a 16-row MFMA tile; logical chunk kk*4 + (lane>>4) ^ ((fr>>1)&7)
    const int fr = lane & 15;
    const int fsw = (fr >> 1) & 7;
    const unsigned fch0 = (unsigned)(((lane >> 4) ^ fsw) * 16), fch1 = (unsigned)(((4 + (lane >> 4)) ^ fsw) * 16);
    const unsigned a_row = (unsigned)(wm * (16 * IM) + fr) * 128u;
    const unsigned b_row = (unsigned)(wn * WCOLS + fr) * 128u;
    bf16x8 af[NKA][IM][2];
#pragma unroll
    for (int kt = 0; kt < NKA; ++kt)
#pragma unroll
        for (int i = 0; i < IM; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                af[kt][i][kk] = *reinterpret_cast<const bf16x8*>(smem + A_OFF + kt * A_BYTES + a_row + (kk ? fch1 : fch0) + i * 2048);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                    // the A image may now be overwritten (ring slots, staging)
    RB_STAMP(3);
#pragma unroll
    for (int s = PRO; s < DEP; ++s) { issue_b(); mk[s] = issued; }

    // ---- LayerNorm in the prologue (pdmk_gemm_args.ln_gamma; BasicTransformerBlock norm1/2/3 -> the projection that reads it,
    // blocks.py:705-867): the row block IS in registers - row fr + 16 i of this wave's 16 IM rows spread over the 4 lanes with equal
    // lane & 15, (lane >> 4) selecting the 8-column chunk of every 32 - so the statistics are one register sweep and two
    // cross-lane adds per row, and the normalised values replace the fragments in place.  Same arithmetic as ln_fwd_kernel
    // (norm.hip): mean, variance of the deviations, (x - mean) * rstd * gamma + beta, one rounding to bf16.  Both waves of a row
    // group (wn = 0 / 1) hold the same rows; in column group 0 wn = 0 writes the statistics and each of the two the half of the
    // normalised rows' 32-column chunks with kk == wn - what the backward pass needs (counted buffer stores, out-of-range lanes
    // dropped by the buffer unit).
    if constexpr (LNP) {
        float* const lnp = reinterpret_cast<float*>(smem + RING);     // the waves' staging area: unused until the first epilogue
        static_assert(2 * KP * 4 <= 8 * STG_WAVE, "gamma | beta fit the staging area");
#pragma unroll
        for (int j = 0; j < LNW; ++j)
            if (tid + NT * j < 2 * KP) lnp[tid + NT * j] = lnv[j];
        const float invK = 1.0f / (float)g.K;
        const int kq = lane >> 4;
        const bool writer = wn == 0 && grp == 0;                     // wave-uniform
        const auto rsrcL = __builtin_amdgcn_make_buffer_rsrc(g.ln_out ? g.ln_out : g.C, (short)0, (int)(g.ln_out ? x.ln_bytes : 0u), 0x00020000);
        const auto rsrcS = __builtin_amdgcn_make_buffer_rsrc(g.ln_stats ? (void*)g.ln_stats : g.C, (short)0, (int)(g.ln_stats ? x.st_bytes : 0u), 0x00020000);
        float mean[IM], rstd[IM];
#pragma unroll
        for (int i = 0; i < IM; ++i) {
            float s1 = 0.f;
#pragma unroll
            for (int kt = 0; kt < NKA; ++kt)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int r = 0; r < 8; ++r) s1 += (float)af[kt][i][kk][r];          // zero-filled past K
            s1 += __shfl_xor(s1, 16, 64);
            s1 += __shfl_xor(s1, 32, 64);
            const float mu = s1 * invK;
            float sq = 0.f;
#pragma unroll
            for (int kt = 0; kt < NKA; ++kt)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const bool kv = kt * BK + (kk * 4 + kq) * 8 < g.K;                   // a chunk of 8 is inside K or padding
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        const float d = (float)af[kt][i][kk][r] - mu;
                        sq += kv ? d * d : 0.f;
                    }
                }
            sq += __shfl_xor(sq, 16, 64);
            sq += __shfl_xor(sq, 32, 64);
            mean[i] = mu;
            rstd[i] = rsqrtf(sq * invK + g.ln_eps);
            if (g.ln_stats && writer) {                              // wave-uniform
                const int m = m0 + wm * (16 * IM) + i * 16 + fr;
                const unsigned so = (kq == 0 && m < g.M) ? (unsigned)m * 8u : OOB;
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{mu, rstd[i]}), rsrcS, (int)so, 0, 0);
                issued += 1;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                // gamma | beta are in LDS for every wave
#pragma unroll
        for (int kt = 0; kt < NKA; ++kt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int k0 = kt * BK + (kk * 4 + kq) * 8;
                const bool kv = k0 < g.K;
                float gm[8], bt[8];
                {
                    const f32x4 g0 = *reinterpret_cast<const f32x4*>(lnp + k0), g1 = *reinterpret_cast<const f32x4*>(lnp + k0 + 4);
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(lnp + KP + k0), b1 = *reinterpret_cast<const f32x4*>(lnp + KP + k0 + 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { gm[r] = g0[r]; gm[4 + r] = g1[r]; bt[r] = b0[r]; bt[4 + r] = b1[r]; }
                }
#pragma unroll
                for (int i = 0; i < IM; ++i) {
                    bf16x8 o;
#pragma unroll
                    for (int r = 0; r < 8; ++r) o[r] = (bf16)(((float)af[kt][i][kk][r] - mean[i]) * rstd[i] * gm[r] + bt[r]);
                    af[kt][i][kk] = o;                               // (padding chunks: gamma = beta = 0 -> stays 0)
                    if (g.ln_out && grp == 0 && kk == wn) {          // wave-uniform: the two waves of a row group share the stores
                        const int m = m0 + wm * (16 * IM) + i * 16 + fr;
                        const unsigned oo = (kv && m < g.M) ? ((unsigned)m * (unsigned)g.ld_ln_out + (unsigned)k0) * 2u : OOB;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, o), rsrcL, (int)oo, 0, 0);
                        issued += 1;
                    }
                }
                if (kk) __builtin_amdgcn_sched_barrier(0);           // two slots' gamma / beta (32 registers) in flight at a time:
            }                                                        // hoisting all 2 NKA slots' reads to the top spills
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                // the staging area is the waves' own again (first epilogue)
    }

    // ---- per-lane epilogue geometry (row-contiguous read-back of the wave's 16 x WCOLS staging image)
    const int mrow0 = m0 + wm * (16 * IM);
    i32x4 rpre[IM][ITEMS];
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) rpre[i][it] = i32x4{0, 0, 0, 0};
    int markR = 0;
    f32x4 acc[IM][NJ];
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // residual / previous output of n-tile t, issued one tile ahead of its use by loads the compiler does not track
    auto prefetch = [&](int t) __attribute__((always_inline)) {
        const int n0 = t * BN + wn * WCOLS;
#pragma unroll
        for (int i = 0; i < IM; ++i)
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int item = lane + 64 * it;
                const int row = item / C8W, c8 = item - row * C8W;
                const int m = mrow0 + i * 16 + row, n = n0 + c8 * 8;
                const unsigned off = (item < 16 * C8W && m < g.M && n < g.N) ? ((unsigned)m * (unsigned)ldadd + (unsigned)n) * 2u : OOB;
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(rpre[i][it]) : "v"(off), "s"(rsrcR));
            }
        issued += IM * ITEMS;
        markR = issued;
    };

    // Epilogue of n-tile t in two halves per 16-row pass i (this wave's rows mrow0 + 16 i .., WCOLS columns), no workgroup
    // barrier: epi_write stages alpha * acc + bias in the wave's fp32 image, epi_read reads it back row-contiguously, adds
    // the prefetched addend / applies GEGLU and stores 16-byte rows.  The n-tiles before the last run them DEFERRED: the
    // accumulators are parked in `old` and the four halves ride at the end of the next n-tile's first K-steps, where the
    // wave's own MFMAs are draining and its SIMD partner has the matrix pipe (as one block at the end of an n-tile they
    // cost ~1700 cycles during which the workgroup's other waves sit at the next barrier)
    auto epi_write = [&](f32x4 (&a)[IM][NJ], int i, int t) __attribute__((always_inline)) {
        const int n0 = t * BN + wn * WCOLS;                          // first column of this wave in the n-tile
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            f32x4 v = a[i][j];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= g.alpha;
            if (g.bias) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + n0 + j * 16 + (lane >> 4) * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += b4[r];
            }
            *reinterpret_cast<f32x4*>(stage + (lane & 15) * SROW + j * 16 + (lane >> 4) * 4) = v;
        }
    };
    auto epi_read = [&](int i, int t) __attribute__((always_inline)) {
        const int n0 = t * BN + wn * WCOLS;
        if (geglu) {
            // GEGLU (blocks.py:44-59): staged columns come as (hidden, gate) blocks of 8; a lane takes one 16-column pair of
            // a row, writes 8 outputs hidden * gelu(gate) and, for the backward, the pair itself (same arithmetic as
            // ring_epilogue: both rounded to bf16 first)
#pragma unroll
            for (int it = 0; it < ITEMS2; ++it) {
                const int item = lane + 64 * it;
                const bool live = item < 16 * NJ;
                const int itc = live ? item : 0;
                const int row = itc / NJ, p = itc - row * NJ;
                const int m = mrow0 + i * 16 + row, n = n0 + p * 16;
                const bool ok = live && m < g.M && n < g.N;
                float v[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(stage + row * SROW + p * 16 + q * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[q * 4 + r] = t4[r];
                }
                bf16x8 hb, gb, o;
#pragma unroll
                for (int r = 0; r < 8; ++r) { hb[r] = (bf16)v[r]; gb[r] = (bf16)v[8 + r]; }
#pragma unroll
                for (int r = 0; r < 8; ++r) o[r] = (bf16)((float)hb[r] * gelu_f((float)gb[r]));
                if (g.C2) {                                          // wave-uniform
                    const unsigned o2 = ok ? ((unsigned)m * (unsigned)g.ldc2 + (unsigned)n) * 2u : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, hb), rsrcC2, (int)o2, 0, PDMK_RB_STORE_AUX);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, gb), rsrcC2, (int)(ok ? o2 + 16u : OOB), 0, PDMK_RB_STORE_AUX);
                }
                const unsigned oo = ok ? ((unsigned)m * (unsigned)g.ldc + (unsigned)(n >> 1)) * 2u : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, o), rsrcC, (int)oo, 0, PDMK_RB_STORE_AUX);
            }
            issued += ITEMS2 * (g.C2 ? 3 : 1);
        } else {
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int item = lane + 64 * it;
                const bool live = item < 16 * C8W;
                const int itc = live ? item : 0;
                const int row = itc / C8W, c8 = itc - row * C8W;
                const int m = mrow0 + i * 16 + row, n = n0 + c8 * 8;
                const bool ok = live && m < g.M && n < g.N;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + row * SROW + c8 * 8);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(stage + row * SROW + c8 * 8 + 4);
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = lo[r]; v[4 + r] = hi[r]; }
                if (addend) {
                    const bf16x8 rr = __builtin_bit_cast(bf16x8, rpre[i][it]);
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] += (float)rr[r];
                }
                bf16x8 o;
#pragma unroll
                for (int r = 0; r < 8; ++r) o[r] = (bf16)v[r];
                const unsigned oo = ok ? ((unsigned)m * (unsigned)g.ldc + (unsigned)n) * 2u : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, o), rsrcC, (int)oo, 0, PDMK_RB_STORE_AUX);
            }
            issued += ITEMS;
        }
    };
    auto epilogue_now = [&](int t) __attribute__((always_inline)) {
        if (addend) {
            wait_vmcnt_dyn(issued - markR);
#pragma unroll
            for (int i = 0; i < IM; ++i)
#pragma unroll
                for (int it = 0; it < ITEMS; ++it) asm volatile("" : "+v"(rpre[i][it]));
        }
#pragma unroll
        for (int i = 0; i < IM; ++i) {
            epi_write(acc, i, t);
            epi_read(i, t);
        }
    };

    // read + multiply one weight stage (ring slot `slot`) against the A fragments of K-step kt.  The wave's LDS-DMA pieces
    // of the stage DEP ahead are issued at one of four points of the MFMA stream, a different one for each wave of a SIMD
    // pair and for neighbouring SIMDs: eight waves issuing their pieces together right after the barrier hold each other
    // at ~140 cycles per piece
    // Software pipeline across the per-step barrier: the kk = 0 fragments of a stage are read during the second half of the
    // step before (the barrier of step q guarantees stage q + 1 has landed).  The wave's DMA pieces of the stage DEP ahead go
    // out either in the first or in the second half of the MFMA stream (EARLY: one uniform branch per step selects the body):
    // SIMD partners w / w + 4 and neighbouring SIMDs differ, so that not all eight waves sit in the memory pipe together
    // 40 registers go either to the pipelined kk = 0 fragments (MODE 0), to the prefetched addend (MODE 1, 24 of them) or to the
    // parked accumulators of a deferred epilogue (MODE 2: the GEGLU epilogue is ~8000 cycles of VALU work per n-tile - erf-GELU on
    // 10 240 elements - which rides on the next n-tile's steps beside the SIMD partner's MFMAs instead of stalling the barrier)
    constexpr bool PIPE = MODE == 0 || (MODE == 2 && !PDMK_RB_DEFER_GEGLU);
    bf16x8 f0[NJ];                                                   // kk = 0 fragments of the stage about to be multiplied
    auto compute = [&](int slot, int kt, bool EARLY) __attribute__((always_inline)) {
        const unsigned char* sb = ring + slot * B_BYTES;
        const unsigned char* sn = ring + (slot + 1 == BST ? 0 : slot + 1) * B_BYTES;
        bf16x8 f1[NJ];
        if (!PIPE) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) f0[j] = *reinterpret_cast<const bf16x8*>(sb + b_row + fch0 + j * 2048);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) f1[j] = *reinterpret_cast<const bf16x8*>(sb + b_row + fch1 + j * 2048);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int i = 0; i < IM; ++i) acc[i][j] = MM::mma(f0[j], af[kt][i][0], acc[i][j]);
            if (EARLY && j == 0) { issue_b(); mk[DEP - 1] = issued; }
        }
        if (PIPE) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) f0[j] = *reinterpret_cast<const bf16x8*>(sn + b_row + fch0 + j * 2048);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int i = 0; i < IM; ++i) acc[i][j] = MM::mma(f1[j], af[kt][i][1], acc[i][j]);
            if (!EARLY && j == 0) { issue_b(); mk[DEP - 1] = issued; }
        }
    };

    constexpr bool DEFER = (PDMK_RB_DEFER || (MODE == 2 && PDMK_RB_DEFER_GEGLU)) && MODE != 1 && 2 * IM <= NKA;   // MODE 1: the addend registers belong to one n-tile at a time
    // accumulators of the n-tile whose epilogue is deferred.  MODE 2 parks bf16(alpha * acc + bias) - exactly the values its
    // epilogue rounds to anyway - in half the registers
    typedef typename std::conditional<MODE == 2, bf16x4, f32x4>::type park_t;
    park_t old[IM][NJ];
#pragma unroll
    for (int i = 0; i < IM; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) old[i][j] = park_t{};
    auto park = [&](int t) __attribute__((always_inline)) {
        const int n0 = t * BN + wn * WCOLS;
#pragma unroll
        for (int i = 0; i < IM; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if constexpr (MODE == 2) {
                    f32x4 v = acc[i][j];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= g.alpha;
                    if (g.bias) {
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + n0 + j * 16 + (lane >> 4) * 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += b4[r];
                    }
                    bf16x4 h;
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[r] = (bf16)v[r];
                    old[i][j] = h;
                } else {
                    old[i][j] = acc[i][j];
                }
                acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
    };
    auto epi_write_parked = [&](int i, int t) __attribute__((always_inline)) {
        if constexpr (MODE == 2) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (float)old[i][j][r];
                *reinterpret_cast<f32x4*>(stage + (lane & 15) * SROW + j * 16 + (lane >> 4) * 4) = v;
            }
        } else {
            epi_write(old, i, t);
        }
    };
    bool pend = false;
    int sc = 0;                                                      // slot being consumed
    wait_vmcnt_dyn(issued - mk[0]);                                  // stage 0, and its kk = 0 fragments
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < NJ; ++j) f0[j] = PIPE ? *reinterpret_cast<const bf16x8*>(ring + b_row + fch0 + j * 2048) : bf16x8{};
    for (int t = t0; t < t1; ++t) {
#pragma unroll
        for (int kin = 0; kin < NKA; ++kin) {
            wait_vmcnt_dyn(issued - mk[1]);                          // stage qc + 1: read in the second half of this step
            __builtin_amdgcn_s_barrier();
#ifdef PDMK_RB_STAMPS
            { const int gs__ = (t - t0) * NKA + kin; if (gs__ < 16) RB_STAMP(4 + gs__); }
#endif
#pragma unroll
            for (int s = 0; s < DEP; ++s) mk[s] = mk[s + 1];
            if (kin == 0 && addend) prefetch(t);
            if (early) compute(sc, kin, true);
            else compute(sc, kin, false);
            sc = sc + 1 == BST ? 0 : sc + 1;
            if (DEFER && pend && kin < 2 * IM) {                     // a half of the previous n-tile's epilogue
                if (kin & 1) epi_read(kin >> 1, t - 1);
                else epi_write_parked(kin >> 1, t - 1);
            }
        }
        if (DEFER && t + 1 < t1) {
            park(t);
            pend = true;
        } else {
            epilogue_now(t);
#pragma unroll
            for (int i = 0; i < IM; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    RB_STAMP(21);
    RB_STAMP(23);
}

struct RbCfg {
    int bm, nj, nka, bst;
};
// candidate table (ids follow the ring and halo candidates of gemm_ring.hip; stable: plan cache values)
static const RbCfg kRb[] = {{128, 5, 5, 5}, {64, 5, 10, 5}};
constexpr int kNumRb = sizeof(kRb) / sizeof(kRb[0]);

static int rb_enabled() {      // PDMK_RB=0: the row-block candidates refuse every shape (same-box A/B against the ring kernels alone)
    static int v = -1;
    if (v < 0 || getenv("PDMK_ENV_DYNAMIC")) v = getenv("PDMK_RB") ? atoi(getenv("PDMK_RB")) : 1;
    return v;
}
static int grp_override() {
    static int v = -2;
    if (v == -2 || getenv("PDMK_ENV_DYNAMIC")) v = getenv("PDMK_RB_GRP") ? atoi(getenv("PDMK_RB_GRP")) : -1;
    return v;
}

}  // namespace pdmk_rb

#ifdef PDMK_RB_STAMPS
extern "C" int pdmk_debug_rb_read_stamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(pdmk_rb::pdmk_rb_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1;
}
#endif

int pdmk_gemm_rowblock_num_configs() { return pdmk_rb::kNumRb; }
int pdmk_gemm_rowblock_name(int id, char* buf, int n) {
    using namespace pdmk_rb;
    if (id < 0 || id >= kNumRb) return -1;
    snprintf(buf, n, "pdmk_rb::rowblock_kernel<%d, %d, %d, %d, ...>", kRb[id].bm, kRb[id].nj, kRb[id].nka, kRb[id].bst);
    return 0;
}

// returns 1 if the shape / epilogue is not handled here (the caller falls back), 0 on launch, < 0 on a launch error
// dry: answer only (0 = this candidate takes the problem), nothing is launched
int pdmk_gemm_rowblock_launch(const pdmk_gemm_args& g, hipStream_t st, long a_bytes, long b_bytes, int id, bool dry) {
    using namespace pdmk_rb;
    if (id < 0 || id >= kNumRb || !rb_enabled()) return 1;
    const RbCfg c = kRb[id];
    const int bn = 32 * c.nj;
    if (g.dtype != PDMK_BF16 || g.a_mode != PDMK_A_ROWK || g.b_mode != PDMK_B_ROWK) return 1;
    if (g.splitk > 1 || g.out_f32 || g.rowvec || g.colsum_out) return 1;
    if (g.accumulate != 0 && g.accumulate != 1) return 1;
    if (g.R && g.accumulate == 1) return 1;
    if ((g.K % 8) || g.K > 64 * c.nka || (g.N % 8) || (g.ldc % 8) || (g.R && (g.ldr % 8))) return 1;
    const int ntn = (g.N + bn - 1) / bn, ntm = (g.M + c.bm - 1) / c.bm;
    if (g.bias && (long)ntn * bn > 4096) return 1;
    const bool geglu = g.epilogue == PDMK_EPI_GEGLU;
    if (g.epilogue != PDMK_EPI_NONE && !geglu) return 1;
    if (g.colstat) return 1;                             // GroupNorm statistics epilogue: ring / halo kernels only
    if (g.ln_gamma && (!g.ln_beta || ((uintptr_t)g.ln_gamma & 15) || ((uintptr_t)g.ln_beta & 15) || (g.ln_out && (g.ld_ln_out % 8)) ||
                       ((uintptr_t)g.ln_out & 15) || ((uintptr_t)g.ln_stats & 7)))
        return 1;
    if (geglu && ((g.N % 16) || g.R || g.accumulate || (g.C2 && (g.ldc2 % 8)))) return 1;
    const long c_bytes = ((long)(g.M - 1) * g.ldc + (geglu ? g.N / 2 : g.N)) * 2;
    const long r_bytes = g.R ? ((long)(g.M - 1) * g.ldr + g.N) * 2 : 0;
    const long c2_bytes = g.C2 ? ((long)(g.M - 1) * g.ldc2 + g.N) * 2 : 0;
    const long ln_bytes = (g.ln_gamma && g.ln_out) ? ((long)(g.M - 1) * g.ld_ln_out + g.K) * 2 : 0;
    const long st_bytes = (g.ln_gamma && g.ln_stats) ? (long)g.M * 8 : 0;
    if (c_bytes >= (1L << 31) || r_bytes >= (1L << 31) || c2_bytes >= (1L << 31) || ln_bytes >= (1L << 31)) return 1;
    // column groups: fill the 256 CUs with (row blocks x groups) workgroups, at least one n-tile per group
    int ngrp = grp_override() > 0 ? grp_override() : (256 + ntm / 2) / ntm;
    if (ngrp < 1) ngrp = 1;
    if (ngrp > ntn) ngrp = ntn;
    const int per = (ntn + ngrp - 1) / ngrp;
    ngrp = (ntn + per - 1) / per;                                    // no empty groups
    RbArgs x{(unsigned)a_bytes, (unsigned)b_bytes, (unsigned)c_bytes, (unsigned)r_bytes, (unsigned)c2_bytes, (unsigned)ln_bytes,
             (unsigned)st_bytes, ngrp};
    dim3 grid(ntm * ngrp);
    if (dry) return 0;
    const int mode = geglu ? 2 : ((g.R || g.accumulate == 1) ? 1 : 0);
#define PDMK_RB_GO(BMv, NJv, NKv, STv)                                                                                  \
    switch (mode + (g.ln_gamma ? 3 : 0)) {                                                                              \
        case 0: hipLaunchKernelGGL((rowblock_kernel<BMv, NJv, NKv, STv, 0>), grid, dim3(NT), 0, st, g, x); break;       \
        case 1: hipLaunchKernelGGL((rowblock_kernel<BMv, NJv, NKv, STv, 1>), grid, dim3(NT), 0, st, g, x); break;       \
        case 2: hipLaunchKernelGGL((rowblock_kernel<BMv, NJv, NKv, STv, 2>), grid, dim3(NT), 0, st, g, x); break;       \
        case 3: hipLaunchKernelGGL((rowblock_kernel<BMv, NJv, NKv, STv, 0, true>), grid, dim3(NT), 0, st, g, x); break; \
        case 5: hipLaunchKernelGGL((rowblock_kernel<BMv, NJv, NKv, STv, 2, true>), grid, dim3(NT), 0, st, g, x); break; \
        default: return 1;                                                                                              \
    }
    switch (id) {
        case 0: PDMK_RB_GO(128, 5, 5, 5) break;
        case 1: PDMK_RB_GO(64, 5, 10, 5) break;
        default: return 1;
    }
#undef PDMK_RB_GO
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}
