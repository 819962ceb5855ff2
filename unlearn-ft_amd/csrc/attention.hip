// Flash-style scaled-dot-product attention for head dim 64 (self-attention over N = HW latent tokens, cross-attention
// over the 77 text tokens), forward + backward, bf16 (MFMA 16x16x32) and exact-fp32 (MFMA 16x16x4) instantiations.
// The N x N score matrix is never materialised: per 64-row query block the kernel walks key blocks with an online
// (base-2) softmax; fp32 statistics; LSE saved for the backward.  Q/K/V/O are addressed with explicit batch/row strides
// so the fused QKV projection output is consumed in place (no head transposes in HBM).
//
// Register-resident probabilities: every product is oriented so that the score tile comes out of the MFMA with the
// index that the NEXT product sums over in the accumulator REGISTERS (and the other index on the lanes).  The
// accumulators are then converted in place and fed back as the B operand (Mma<T>::acc_frag); the matching A operand is
// read transposed from the K / V / Q / dO tile that already sits in LDS in memory order (Mma<T>::load_colk_accs =
// ds_read_b64_tr_b16 with the same k permutation).  P, dS never touch LDS; softmax statistics are per-lane scalars.
//   forward : S^T = K Q^T  (key in regs, query on lanes)   O^T += V^T P^T
//   dK/dV   : S   = Q K^T  (query in regs, key on lanes)   dV^T += dO^T P,  dK^T += Q^T dS   (no atomics)
//   dQ      : S^T = K Q^T                                  dQ^T += K^T dS^T
// K/V (resp. Q/dO) tiles are double buffered in LDS; the next tile's buffer loads (hardware zero-fill past the sequence
// end) are issued before the current tile's MFMAs: one barrier per tile.
#include <cstdlib>

#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int D = 64;
constexpr float LOG2E = 1.4426950408889634f;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
constexpr unsigned OOB = 0x80000000u;

#ifndef PDMK_ATTN_KVB
#define PDMK_ATTN_KVB 64      // streamed K/V (forward, dq) and Q/dO (dk/dv) rows per block and barrier pair
#endif
template <typename T> struct ACfg;
template <> struct ACfg<bf16> {
    static constexpr int CH = 8, RS = 64, KVB = PDMK_ATTN_KVB;   // RS: LDS row stride of a [rows][64] tile; KVB: streamed rows
};
template <> struct ACfg<float> {
    static constexpr int CH = 4, RS = 68, KVB = 32;
};

// LDS tile addressing.  bf16: unpadded 128-byte rows with the 16-byte chunk index XOR-ed with (row & 7): conflict-free both
// for the row-wise ds_read_b128 fragments (Q / K / dO rows as MFMA operands) and for the transposing ds_read_b64_tr_b16
// reads of the same image in accumulator order (8 consecutive rows per 32-lane group) - the padded [rows][72] image it
// replaces spent 36-43 % of its LDS cycles on bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).  fp32: padded.
template <typename T> struct AT;
template <> struct AT<bf16> {
    typedef Mma<bf16> MM;
    static constexpr int RS = 64;
    static __device__ __forceinline__ int sw(int row) { return row & 7; }
    static __device__ __forceinline__ MM::frag rowk(const bf16* lds, int row0, int k0, int lane) {
        const int row = row0 + (lane & 15), ch = ((k0 >> 3) + (lane >> 4)) ^ (row & 7);
        return *reinterpret_cast<const bf16x8*>(lds + row * RS + ch * 8);
    }
    template <int S> static __device__ __forceinline__ MM::frag colk_accs(const bf16* lds, int k0, int col0, int lane) {
        const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
        const int row = k0 + 4 * g + q;                        // rows row and row + 16 share (row & 7)
        const bf16* a0 = lds + row * RS + ((((col0 >> 3) + (p >> 1)) ^ (row & 7)) * 8) + (p & 1) * 4;
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 16 * RS));
        s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    }
};
template <> struct AT<float> {
    typedef Mma<float> MM;
    static constexpr int RS = 68;
    static __device__ __forceinline__ int sw(int) { return 0; }
    static __device__ __forceinline__ MM::frag rowk(const float* lds, int row0, int k0, int lane) {
        return MM::load_rowk(lds, RS, row0, k0, lane);
    }
    template <int S> static __device__ __forceinline__ MM::frag colk_accs(const float* lds, int k0, int col0, int lane) {
        return MM::template load_colk_accs<S>(lds, RS, k0, col0, lane);
    }
};

// A [ROWS][64] tile of a strided [N][..] matrix: buffer loads into registers (rows past the end read as zero), LDS
// image [ROWS][RS].  NR = 16-byte chunks per thread.
template <typename T, int ROWS> struct TileIO {
    static constexpr int CH = ACfg<T>::CH, RS = ACfg<T>::RS, CPR = D / CH, NR = ROWS * CPR / NT;
    static_assert(ROWS * CPR % NT == 0, "tile chunks must divide the block");
    template <typename RS_T>
    static __device__ __forceinline__ void load(u32x4* r, const RS_T& rsrc, int ld, int row0, int nrows, int tid) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int c = tid + i * NT, row = row0 + c / CPR, dc = (c % CPR) * CH;
            const unsigned off = row < nrows ? ((unsigned)row * (unsigned)ld + (unsigned)dc) * (unsigned)sizeof(T) : OOB;
            r[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0);
        }
    }
    static __device__ __forceinline__ void store(T* lds, const u32x4* r, int tid) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int c = tid + i * NT, row = c / CPR;
            *reinterpret_cast<u32x4*>(lds + row * RS + (((c % CPR) ^ AT<T>::sw(row)) * CH)) = r[i];
        }
    }
};

template <typename T> __device__ __forceinline__ auto make_rsrc(const T* p, long nrows, int ld) {
    // upper bound of the bytes addressable from p within this (batch, head) slice
    const long bytes = ((nrows - 1) * (long)ld + D) * (long)sizeof(T);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(p), (short)0, (int)bytes, 0x00020000);
}

// wave-private LDS [16][RS] (cols 0..63) -> global rows, 16-byte stores; rows >= nrows skipped
template <typename T>
__device__ __forceinline__ void store_tile16(const T* lds, T* g, long ld, int row0, int nrows, int lane) {
    constexpr int CH = ACfg<T>::CH, RS = ACfg<T>::RS, CPR = D / CH;
    for (int c = lane; c < 16 * CPR; c += 64) {
        const int r = c / CPR, dc = (c % CPR) * CH;
        if (row0 + r < nrows)
            *reinterpret_cast<uint4*>(g + (long)(row0 + r) * ld + dc) = *reinterpret_cast<const uint4*>(lds + r * RS + dc);
    }
}
// accumulators X^T[d][c] (d = dt*16 + 4g + r in registers, c = lane&15) -> wave-private LDS [c][d] image
template <typename T>
__device__ __forceinline__ void stage_t(T* lds, const f32x4* acc, float scale, int lane) {
    constexpr int RS = ACfg<T>::RS;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            lds[(lane & 15) * RS + dt * 16 + (lane >> 4) * 4 + r] = from_f32<T>(acc[dt][r] * scale);
}

template <typename MM, int S, int NS> struct AccLoop {     // compile-time loop over the MFMA steps of one tile pair
    template <typename F> static __device__ __forceinline__ void run(F&& f) {
        f(std::integral_constant<int, S>{});
        AccLoop<MM, S + 1, NS>::run(f);
    }
};
template <typename MM, int NS> struct AccLoop<MM, NS, NS> {
    template <typename F> static __device__ __forceinline__ void run(F&&) {}
};

// ================================================================================================ forward
// one workgroup = 64*NQ queries of one (b,h); wave w owns NQ tiles of 16 queries (on the lanes) and sweeps the keys in
// tiles of KVB.  NQ = 2 halves the K / V fragment reads per MFMA (every fragment feeds both query tiles): the loop is
// LDS-read bound at NQ = 1 (24 reads per 16 MFMAs).
// CAUSAL (CLIP text tower): key j is visible to query i only for j <= i; key 0 is visible to every query, so
// the first key tile never leaves a row without a finite maximum.
template <typename T, int NQ, bool CAUSAL = false>
__global__ __launch_bounds__(NT) void attn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                      const T* __restrict__ v, T* __restrict__ o,
                                                      float* __restrict__ lse, int H, int Nq, int Nk, long q_bs,
                                                      int q_ld, long k_bs, int k_ld, long v_bs, int v_ld, long o_bs,
                                                      int o_ld, float scale2) {
    typedef Mma<T> MM;
    typedef ACfg<T> AC;
    constexpr int RS = AC::RS, KVB = AC::KVB, NKT = KVB / 16, QROWS = 64 * NQ;
    typedef TileIO<T, QROWS> QIO;
    typedef TileIO<T, KVB> KIO;
    __shared__ __attribute__((aligned(16))) T Qs[QROWS * RS];
    __shared__ __attribute__((aligned(16))) T Ks[2][KVB * RS];
    __shared__ __attribute__((aligned(16))) T Vs[2][KVB * RS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q0 = blockIdx.x * QROWS, h = blockIdx.y, b = blockIdx.z;
    const auto rq = make_rsrc(q + b * q_bs + h * D, Nq, q_ld);
    const auto rk = make_rsrc(k + b * k_bs + h * D, Nk, k_ld);
    const auto rv = make_rsrc(v + b * v_bs + h * D, Nk, v_ld);

    u32x4 rk_[KIO::NR], rv_[KIO::NR];
    {
        u32x4 rq_[QIO::NR];
        QIO::load(rq_, rq, q_ld, q0, Nq, tid);
        KIO::load(rk_, rk, k_ld, 0, Nk, tid);
        KIO::load(rv_, rv, v_ld, 0, Nk, tid);
        QIO::store(Qs, rq_, tid);
    }
    KIO::store(Ks[0], rk_, tid);
    KIO::store(Vs[0], rv_, tid);
    __syncthreads();
    typename MM::frag qf[NQ][D / MM::KS];  // B operand: B[k = d][col = query]
#pragma unroll
    for (int n = 0; n < NQ; ++n)
#pragma unroll
        for (int kk = 0; kk < D / MM::KS; ++kk) qf[n][kk] = AT<T>::rowk(Qs, (wave * NQ + n) * 16, kk * MM::KS, lane);

    f32x4 ot[NQ][4], ol[NQ];                // O^T[d = dt*16 + 4g + r][query = lane&15]; ol: softmax denominators
    float m_run[NQ];                        // running row maximum (scaled scores), per query column
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        m_run[n] = -INFINITY;
        ol[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) ot[n][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    typename MM::frag ones;
    if constexpr (std::is_same<T, bf16>::value) {
#pragma unroll
        for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;
    } else {
        ones = 1.0f;
    }

    // One key block.  TAIL (the last block when Nk is not a multiple of KVB) is the only one that masks keys past the end: it is
    // a second copy of the body, so that the full blocks carry no mask arithmetic at all (the compiler if-converted the mask of a
    // single body into ~50 compare / select instructions per block, a quarter of the loop's VALU work, which binds this kernel).
    int cur = 0;
    auto block = [&](auto tail_c, const int kb) {
        constexpr bool TAIL = decltype(tail_c)::value;
        const bool more = kb + KVB < Nk;
        if (more) {
            KIO::load(rk_, rk, k_ld, kb + KVB, Nk, tid);
            KIO::load(rv_, rv, v_ld, kb + KVB, Nk, tid);
        }
        f32x4 st[NQ][NKT];                  // S^T[key = 16t + 4g + r][query]
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
#pragma unroll
            for (int n = 0; n < NQ; ++n) st[n][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < D / MM::KS; ++kk) {
                const typename MM::frag kf = AT<T>::rowk(Ks[cur], t * 16, kk * MM::KS, lane);
#pragma unroll
                for (int n = 0; n < NQ; ++n) st[n][t] = MM::mma(kf, qf[n][kk], st[n][t]);
            }
        }
        // softmax on the raw scores (the loop is VALU-bound at d = 64: per score element a max, half a packed fma, an exp2 and
        // half a convert; the scale rides in the fma, the row sum is an MFMA against a ones fragment)
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            float mx = -INFINITY;
            if (TAIL) {
#pragma unroll
                for (int t = 0; t < NKT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kb + t * 16 + (lane >> 4) * 4 + r >= Nk) st[n][t][r] = -INFINITY;
            }
            if (CAUSAL) {
                const int qi = q0 + (wave * NQ + n) * 16 + (lane & 15);
#pragma unroll
                for (int t = 0; t < NKT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kb + t * 16 + (lane >> 4) * 4 + r > qi) st[n][t][r] = -INFINITY;
            }
#pragma unroll
            for (int t = 0; t < NKT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[n][t][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            // Lazy rescale: the running reference m_run only moves when the block maximum exceeds it by more than 2^8 - then
            // exp2(s - m_run) <= 256 in every later block (fp32 accumulators, bf16 probabilities: same relative precision), and
            // O, l and the saved LSE = m_run + log2(l) stay consistent with whatever reference was used.  After the first few
            // blocks no lane of the wave moves, and the 40 accumulator multiplies + the exp2 of alpha are skipped (wave-uniform)
            const float mxs = mx * scale2;
            const bool move = mxs > m_run[n] + 8.0f;              // m_run starts at -inf: the first block always moves
            if (__builtin_amdgcn_ballot_w64(move)) {
                const float mn = move ? mxs : m_run[n];
                const float alpha = __builtin_amdgcn_exp2f(m_run[n] - mn);
                m_run[n] = mn;
#pragma unroll
                for (int r = 0; r < 4; ++r) ol[n][r] *= alpha;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ot[n][dt][r] *= alpha;
            }
            const f32x2 sc2 = {scale2, scale2}, nm2 = {-m_run[n], -m_run[n]};
#pragma unroll
            for (int t = 0; t < NKT; ++t)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const f32x2 e2 = __builtin_elementwise_fma(f32x2{st[n][t][r], st[n][t][r + 1]}, sc2, nm2);   // v_pk_fma_f32
                    st[n][t][r] = __builtin_amdgcn_exp2f(e2[0]);
                    st[n][t][r + 1] = __builtin_amdgcn_exp2f(e2[1]);
                }
        }
#pragma unroll
        for (int pr = 0; pr < NKT / 2; ++pr) {
            AccLoop<MM, 0, MM::ACC_STEPS>::run([&](auto sc) {
                constexpr int S = decltype(sc)::value;
                typename MM::frag pf[NQ];
#pragma unroll
                for (int n = 0; n < NQ; ++n) {
                    pf[n] = MM::template acc_frag<S>(st[n][2 * pr], st[n][2 * pr + 1]);
                    ol[n] = MM::mma(ones, pf[n], ol[n]);             // row sums of P: every row of ol is l
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const typename MM::frag vf = AT<T>::template colk_accs<S>(Vs[cur], pr * 32, dt * 16, lane);
#pragma unroll
                    for (int n = 0; n < NQ; ++n) ot[n][dt] = MM::mma(vf, pf[n], ot[n][dt]);
                }
            });
        }
        if (more) {
            KIO::store(Ks[cur ^ 1], rk_, tid);
            KIO::store(Vs[cur ^ 1], rv_, tid);
        }
        __syncthreads();
        cur ^= 1;
    };
    const int nfull = Nk / KVB * KVB;
    for (int kb = 0; kb < nfull; kb += KVB) block(std::false_type{}, kb);
    if (nfull < Nk) block(std::true_type{}, nfull);
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        const float l = ol[n][0];
        T* Ow = Qs + (wave * NQ + n) * 16 * RS;        // this wave's own Q rows, free since qf was hoisted
        stage_t<T>(Ow, ot[n], 1.0f / l, lane);
        const int qi = q0 + (wave * NQ + n) * 16 + (lane & 15);
        if (lane < 16 && qi < Nq) lse[((long)b * H + h) * Nq + qi] = m_run[n] + log2f(l);
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NQ; ++n)
        store_tile16<T>(Qs + (wave * NQ + n) * 16 * RS, o + b * o_bs + h * D, o_ld, q0 + (wave * NQ + n) * 16, Nq, lane);
}

// ================================================================================================ backward
template <typename T>
__global__ void attn_delta_kernel(const T* __restrict__ o, const T* __restrict__ d_o, float* __restrict__ delta, int H,
                                  int Nq, long o_bs, int o_ld, long total) {
    constexpr int CH = ACfg<T>::CH;
    for (long i = blockIdx.x * (long)NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int qi = (int)(i % Nq);
        const long bh = i / Nq;
        const int h = (int)(bh % H);
        const long b = bh / H;
        const T* po = o + b * o_bs + (long)qi * o_ld + h * D;
        const T* pd = d_o + b * o_bs + (long)qi * o_ld + h * D;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < D; c += CH) {
            const uint4 a = *reinterpret_cast<const uint4*>(po + c), g = *reinterpret_cast<const uint4*>(pd + c);
            const T* ae = reinterpret_cast<const T*>(&a);
            const T* ge = reinterpret_cast<const T*>(&g);
#pragma unroll
            for (int e = 0; e < CH; ++e) s += to_f32(ae[e]) * to_f32(ge[e]);
        }
        delta[i] = s;
    }
}

// one workgroup = 64*NKW keys of one (b,h); wave w owns NKW tiles of 16 keys (on the lanes); sweeps query blocks of QB
// rows.  Every Q / dO fragment (row-wise for S and dP, transposed for dV and dK) feeds all NKW key tiles.
// SPLIT (cross-attention: 77 keys = 2 key blocks, a 32-workgroup grid otherwise; NKW = 1): blockIdx.x = split *
// key_blocks + key block, the workgroup sweeps only its share of the queries and stores fp32 partial dK / dV into slab
// `split` of `ws` ([nsplit][2][B][H][Nk][64]); attn_dkv_reduce_kernel adds the slabs.
template <typename T, bool SPLIT, int NKW>
__global__ __launch_bounds__(NT) void attn_bwd_dkv_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                          const T* __restrict__ v, const T* __restrict__ d_o,
                                                          const float* __restrict__ lse,
                                                          const float* __restrict__ delta, T* __restrict__ dk,
                                                          T* __restrict__ dv, int H, int Nq, int Nk, long q_bs,
                                                          int q_ld, long k_bs, int k_ld, long v_bs, int v_ld,
                                                          long o_bs, int o_ld, long dk_bs, int dk_ld, long dv_bs,
                                                          int dv_ld, float scale, float scale2, float* __restrict__ ws,
                                                          int nsplit) {
    typedef Mma<T> MM;
    typedef ACfg<T> AC;
    constexpr int RS = AC::RS, QB = AC::KVB, NQT = QB / 16, KROWS = 64 * NKW;
    typedef TileIO<T, KROWS> KIO;
    typedef TileIO<T, QB> QIO;
    static_assert(!SPLIT || NKW == 1, "the query split is for few-key grids");
    __shared__ __attribute__((aligned(16))) T Ks[KROWS * RS];  // K, V staging; reused for the dK / dV output staging
    __shared__ __attribute__((aligned(16))) T Vs[KROWS * RS];
    __shared__ __attribute__((aligned(16))) T Qs[2][QB * RS];
    __shared__ __attribute__((aligned(16))) T Os[2][QB * RS];
    __shared__ __attribute__((aligned(16))) float Ls[2][QB], Ds[2][QB];   // -lse, -delta of the query block

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kblocks = (Nk + KROWS - 1) / KROWS;
    const int split = SPLIT ? blockIdx.x / kblocks : 0;
    const int k0 = (SPLIT ? blockIdx.x - split * kblocks : blockIdx.x) * KROWS, h = blockIdx.y, b = blockIdx.z;
    int q_begin = 0, q_end = Nq;
    if (SPLIT) {
        const int nqb = (Nq + QB - 1) / QB, per = (nqb + nsplit - 1) / nsplit;
        q_begin = min(Nq, split * per * QB);
        q_end = min(Nq, (split + 1) * per * QB);
    }
    const auto rq = make_rsrc(q + b * q_bs + h * D, Nq, q_ld);
    const auto ro = make_rsrc(d_o + b * o_bs + h * D, Nq, o_ld);
    const auto rk = make_rsrc(k + b * k_bs + h * D, Nk, k_ld);
    const auto rv = make_rsrc(v + b * v_bs + h * D, Nk, v_ld);
    const float* lse_bh = lse + ((long)b * H + h) * Nq;
    const float* del_bh = delta + ((long)b * H + h) * Nq;

    u32x4 rq_[QIO::NR], ro_[QIO::NR];
    float rl = -INFINITY, rd = 0.f;
    {
        u32x4 rk_[KIO::NR], rv_[KIO::NR];
        KIO::load(rk_, rk, k_ld, k0, Nk, tid);
        KIO::load(rv_, rv, v_ld, k0, Nk, tid);
        QIO::load(rq_, rq, q_ld, q_begin, Nq, tid);
        QIO::load(ro_, ro, o_ld, q_begin, Nq, tid);
        if (tid < QB && q_begin + tid < Nq) { rl = -lse_bh[q_begin + tid]; rd = -del_bh[q_begin + tid]; }
        KIO::store(Ks, rk_, tid);
        KIO::store(Vs, rv_, tid);
    }
    QIO::store(Qs[0], rq_, tid);
    QIO::store(Os[0], ro_, tid);
    if (tid < QB) { Ls[0][tid] = rl; Ds[0][tid] = rd; }
    __syncthreads();
    typename MM::frag kf[NKW][D / MM::KS], vf[NKW][D / MM::KS];  // B operands: B[k = d][col = key]
    f32x4 dkt[NKW][4], dvt[NKW][4];        // dK^T / dV^T [d = dt*16 + 4g + r][key = lane&15]
#pragma unroll
    for (int j = 0; j < NKW; ++j) {
#pragma unroll
        for (int kk = 0; kk < D / MM::KS; ++kk) {
            kf[j][kk] = AT<T>::rowk(Ks, (wave * NKW + j) * 16, kk * MM::KS, lane);
            vf[j][kk] = AT<T>::rowk(Vs, (wave * NKW + j) * 16, kk * MM::KS, lane);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { dkt[j][i] = f32x4{0.f, 0.f, 0.f, 0.f}; dvt[j][i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }

    int cur = 0;
    for (int qb = q_begin; qb < q_end; qb += QB) {
        const bool more = qb + QB < q_end;
        if (more) {
            QIO::load(rq_, rq, q_ld, qb + QB, Nq, tid);
            QIO::load(ro_, ro, o_ld, qb + QB, Nq, tid);
            rl = -INFINITY; rd = 0.f;
            if (tid < QB && qb + QB + tid < Nq) { rl = -lse_bh[qb + QB + tid]; rd = -del_bh[qb + QB + tid]; }
        }
        f32x4 p[NKW][NQT], ds[NKW][NQT];   // P / dS [query = 16t + 4g + r][key]
#pragma unroll
        for (int t = 0; t < NQT; ++t) {
            f32x4 sacc[NKW], dp[NKW];
            // dP starts at -delta (dS = P * (dP - delta) without the subtraction)
            const f32x4 nd = *reinterpret_cast<const f32x4*>(&Ds[cur][t * 16 + (lane >> 4) * 4]);
            const f32x4 nlv = *reinterpret_cast<const f32x4*>(&Ls[cur][t * 16 + (lane >> 4) * 4]);
#pragma unroll
            for (int j = 0; j < NKW; ++j) { sacc[j] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[j] = nd; }
#pragma unroll
            for (int kk = 0; kk < D / MM::KS; ++kk) {
                const typename MM::frag qa = AT<T>::rowk(Qs[cur], t * 16, kk * MM::KS, lane);
                const typename MM::frag oa = AT<T>::rowk(Os[cur], t * 16, kk * MM::KS, lane);
#pragma unroll
                for (int j = 0; j < NKW; ++j) {
                    sacc[j] = MM::mma(qa, kf[j][kk], sacc[j]);
                    dp[j] = MM::mma(oa, vf[j][kk], dp[j]);
                }
            }
            // (keys past the end need no mask here: a key is an output COLUMN of both products, so whatever its probabilities are -
            // even inf when exp2(-lse) overflows - stays in dK / dV columns that are never stored; queries past the end have
            // -lse = -inf, i.e. p = 0)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
#pragma unroll
                for (int j = 0; j < NKW; ++j) {
                    const f32x2 e2 = __builtin_elementwise_fma(f32x2{sacc[j][r], sacc[j][r + 1]}, f32x2{scale2, scale2}, f32x2{nlv[r], nlv[r + 1]});
                    const f32x2 pv = {__builtin_amdgcn_exp2f(e2[0]), __builtin_amdgcn_exp2f(e2[1])};
                    const f32x2 d2 = pv * f32x2{dp[j][r], dp[j][r + 1]};
                    p[j][t][r] = pv[0];
                    p[j][t][r + 1] = pv[1];
                    ds[j][t][r] = d2[0];
                    ds[j][t][r + 1] = d2[1];
                }
            }
        }
#pragma unroll
        for (int pr = 0; pr < NQT / 2; ++pr) {
            AccLoop<MM, 0, MM::ACC_STEPS>::run([&](auto sc) {
                constexpr int S = decltype(sc)::value;
                typename MM::frag pf[NKW], sf[NKW];
#pragma unroll
                for (int j = 0; j < NKW; ++j) {
                    pf[j] = MM::template acc_frag<S>(p[j][2 * pr], p[j][2 * pr + 1]);
                    sf[j] = MM::template acc_frag<S>(ds[j][2 * pr], ds[j][2 * pr + 1]);
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const typename MM::frag oc = AT<T>::template colk_accs<S>(Os[cur], pr * 32, dt * 16, lane);
                    const typename MM::frag qc = AT<T>::template colk_accs<S>(Qs[cur], pr * 32, dt * 16, lane);
#pragma unroll
                    for (int j = 0; j < NKW; ++j) {
                        dvt[j][dt] = MM::mma(oc, pf[j], dvt[j][dt]);
                        dkt[j][dt] = MM::mma(qc, sf[j], dkt[j][dt]);
                    }
                }
            });
        }
        if (more) {
            QIO::store(Qs[cur ^ 1], rq_, tid);
            QIO::store(Os[cur ^ 1], ro_, tid);
            if (tid < QB) { Ls[cur ^ 1][tid] = rl; Ds[cur ^ 1][tid] = rd; }
        }
        __syncthreads();
        cur ^= 1;
    }
    if (SPLIT) {       // fp32 partials straight from the accumulators: lane = (d-group g, key), 4 consecutive d per store
        const int key = k0 + wave * 16 + (lane & 15);
        if (key < Nk) {
            const long slab = ((((long)split * 2) * gridDim.z + b) * H + h) * Nk + key;      // dK slab row
            const long dv_off = (long)gridDim.z * H * Nk;                                     // dV slabs follow dK's
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4 a = dkt[0][dt];
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] *= scale;
                *reinterpret_cast<f32x4*>(ws + slab * 64 + dt * 16 + (lane >> 4) * 4) = a;
                *reinterpret_cast<f32x4*>(ws + (slab + dv_off) * 64 + dt * 16 + (lane >> 4) * 4) = dvt[0][dt];
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < NKW; ++j) {        // K/V staging is free (fragments hoisted, all reads behind barriers)
        stage_t<T>(Ks + (wave * NKW + j) * 16 * RS, dkt[j], scale, lane);
        stage_t<T>(Vs + (wave * NKW + j) * 16 * RS, dvt[j], 1.0f, lane);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NKW; ++j) {
        store_tile16<T>(Ks + (wave * NKW + j) * 16 * RS, dk + b * dk_bs + h * D, dk_ld, k0 + (wave * NKW + j) * 16, Nk, lane);
        store_tile16<T>(Vs + (wave * NKW + j) * 16 * RS, dv + b * dv_bs + h * D, dv_ld, k0 + (wave * NKW + j) * 16, Nk, lane);
    }
}

// one workgroup = 64*NQ queries of one (b,h); wave w owns NQ tiles of 16 queries (on the lanes); sweeps key blocks.
// Every K / V fragment feeds all NQ query tiles (NQ = 2 halves the LDS reads per MFMA).
template <typename T, int NQ>
__global__ __launch_bounds__(NT) void attn_bwd_dq_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                         const T* __restrict__ v, const T* __restrict__ o,
                                                         const T* __restrict__ d_o,
                                                         const float* __restrict__ lse,
                                                         float* __restrict__ delta, T* __restrict__ dq, int H,
                                                         int Nq, int Nk, long q_bs, int q_ld, long k_bs, int k_ld,
                                                         long v_bs, int v_ld, long o_bs, int o_ld, long dq_bs,
                                                         int dq_ld, float scale, float scale2) {
    typedef Mma<T> MM;
    typedef ACfg<T> AC;
    constexpr int RS = AC::RS, KVB = AC::KVB, NKT = KVB / 16, QROWS = 64 * NQ;
    typedef TileIO<T, QROWS> QIO;
    typedef TileIO<T, KVB> KIO;
    __shared__ __attribute__((aligned(16))) T Qs[QROWS * RS];   // Q staging, then dQ output staging
    __shared__ __attribute__((aligned(16))) T Os[QROWS * RS];
    __shared__ __attribute__((aligned(16))) T Ks[2][KVB * RS];
    __shared__ __attribute__((aligned(16))) T Vs[2][KVB * RS];
    __shared__ float dls[QROWS];                                // delta of this block's queries

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q0 = blockIdx.x * QROWS, h = blockIdx.y, b = blockIdx.z;
    const auto rq = make_rsrc(q + b * q_bs + h * D, Nq, q_ld);
    const auto ro = make_rsrc(d_o + b * o_bs + h * D, Nq, o_ld);
    const auto rO = make_rsrc(o + b * o_bs + h * D, Nq, o_ld);
    const auto rk = make_rsrc(k + b * k_bs + h * D, Nk, k_ld);
    const auto rv = make_rsrc(v + b * v_bs + h * D, Nk, v_ld);

    u32x4 rk_[KIO::NR], rv_[KIO::NR];
    {
        u32x4 rq_[QIO::NR], ro_[QIO::NR], rO_[QIO::NR];
        QIO::load(rq_, rq, q_ld, q0, Nq, tid);
        QIO::load(ro_, ro, o_ld, q0, Nq, tid);
        QIO::load(rO_, rO, o_ld, q0, Nq, tid);
        KIO::load(rk_, rk, k_ld, 0, Nk, tid);
        KIO::load(rv_, rv, v_ld, 0, Nk, tid);
        QIO::store(Qs, rq_, tid);
        QIO::store(Os, ro_, tid);
        // delta[query] = rowsum(dO * O) (the flash backward's D term), formed here from the chunks this thread holds of both
        // tiles - chunk c = tid + i * NT is (row c / CPR, columns (c % CPR) * CH ..): the CPR threads of a row are CPR
        // consecutive lanes - and published for the dK / dV kernel, which is launched after this one
        constexpr int CPR = QIO::CPR, CH = QIO::CH;
#pragma unroll
        for (int i = 0; i < QIO::NR; ++i) {
            const T* a = reinterpret_cast<const T*>(&ro_[i]);
            const T* c_ = reinterpret_cast<const T*>(&rO_[i]);
            float s_ = 0.f;
#pragma unroll
            for (int e = 0; e < CH; ++e) s_ += to_f32(a[e]) * to_f32(c_[e]);
#pragma unroll
            for (int m = 1; m < CPR; m <<= 1) s_ += __shfl_xor(s_, m, 64);
            const int c = tid + i * NT, row = c / CPR;
            if ((c % CPR) == 0) {
                dls[row] = s_;
                if (q0 + row < Nq) delta[((long)b * H + h) * Nq + q0 + row] = s_;
            }
        }
    }
    KIO::store(Ks[0], rk_, tid);
    KIO::store(Vs[0], rv_, tid);
    __syncthreads();
    typename MM::frag qf[NQ][D / MM::KS], of[NQ][D / MM::KS];       // B operands: B[k = d][col = query]
    float nl2[NQ], ndl[NQ];                                         // -lse, -delta of this lane's query
    f32x4 dqt[NQ][4];                                               // dQ^T[d][query]
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
#pragma unroll
        for (int kk = 0; kk < D / MM::KS; ++kk) {
            qf[n][kk] = AT<T>::rowk(Qs, (wave * NQ + n) * 16, kk * MM::KS, lane);
            of[n][kk] = AT<T>::rowk(Os, (wave * NQ + n) * 16, kk * MM::KS, lane);
        }
        const int qi = q0 + (wave * NQ + n) * 16 + (lane & 15);
        nl2[n] = qi < Nq ? -lse[((long)b * H + h) * Nq + qi] : -INFINITY;
        ndl[n] = qi < Nq ? -dls[(wave * NQ + n) * 16 + (lane & 15)] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) dqt[n][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // One key block.  TAIL (the last block when Nk is not a multiple of KVB) zeroes the probabilities of keys past the end and is a
    // second copy of the body: the full blocks carry no mask arithmetic (the select per score element was half of this loop's VALU
    // instructions).  The mask itself cannot go: a missing key has K = 0, so s = 0 and p = exp2(-lse), which overflows when the
    // row's real scores are all below -128 (base 2) - inf * 0 in the dQ = dS K product would poison the whole row.
    int cur = 0;
    auto block = [&](auto tail_c, const int kb) {
        constexpr bool TAIL = decltype(tail_c)::value;
        const bool more = kb + KVB < Nk;
        if (more) {
            KIO::load(rk_, rk, k_ld, kb + KVB, Nk, tid);
            KIO::load(rv_, rv, v_ld, kb + KVB, Nk, tid);
        }
        f32x4 dst[NQ][NKT];                 // dS^T[key = 16t + 4g + r][query]
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
            f32x4 s[NQ], dp[NQ];
#pragma unroll
            for (int n = 0; n < NQ; ++n) {  // dP starts at -delta: dS = P * (dP - delta) needs no subtraction
                s[n] = f32x4{0.f, 0.f, 0.f, 0.f};
                dp[n] = f32x4{ndl[n], ndl[n], ndl[n], ndl[n]};
            }
#pragma unroll
            for (int kk = 0; kk < D / MM::KS; ++kk) {
                const typename MM::frag kf = AT<T>::rowk(Ks[cur], t * 16, kk * MM::KS, lane);
                const typename MM::frag vf = AT<T>::rowk(Vs[cur], t * 16, kk * MM::KS, lane);
#pragma unroll
                for (int n = 0; n < NQ; ++n) {
                    s[n] = MM::mma(kf, qf[n][kk], s[n]);
                    dp[n] = MM::mma(vf, of[n][kk], dp[n]);
                }
            }
#pragma unroll
            for (int n = 0; n < NQ; ++n)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {          // two score elements per v_pk_fma_f32 / v_pk_mul_f32
                    const f32x2 e2 = __builtin_elementwise_fma(f32x2{s[n][r], s[n][r + 1]}, f32x2{scale2, scale2}, f32x2{nl2[n], nl2[n]});
                    f32x2 pv = {__builtin_amdgcn_exp2f(e2[0]), __builtin_amdgcn_exp2f(e2[1])};
                    if (TAIL) {
                        if (kb + t * 16 + (lane >> 4) * 4 + r >= Nk) pv[0] = 0.f;
                        if (kb + t * 16 + (lane >> 4) * 4 + r + 1 >= Nk) pv[1] = 0.f;
                    }
                    const f32x2 d2 = pv * f32x2{dp[n][r], dp[n][r + 1]};
                    dst[n][t][r] = d2[0];
                    dst[n][t][r + 1] = d2[1];
                }
        }
#pragma unroll
        for (int pr = 0; pr < NKT / 2; ++pr) {
            AccLoop<MM, 0, MM::ACC_STEPS>::run([&](auto sc) {
                constexpr int S = decltype(sc)::value;
                typename MM::frag sf[NQ];
#pragma unroll
                for (int n = 0; n < NQ; ++n) sf[n] = MM::template acc_frag<S>(dst[n][2 * pr], dst[n][2 * pr + 1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const typename MM::frag kc = AT<T>::template colk_accs<S>(Ks[cur], pr * 32, dt * 16, lane);
#pragma unroll
                    for (int n = 0; n < NQ; ++n) dqt[n][dt] = MM::mma(kc, sf[n], dqt[n][dt]);
                }
            });
        }
        if (more) {
            KIO::store(Ks[cur ^ 1], rk_, tid);
            KIO::store(Vs[cur ^ 1], rv_, tid);
        }
        __syncthreads();
        cur ^= 1;
    };
    const int nfull = Nk / KVB * KVB;
    for (int kb = 0; kb < nfull; kb += KVB) block(std::false_type{}, kb);
    if (nfull < Nk) block(std::true_type{}, nfull);
#pragma unroll
    for (int n = 0; n < NQ; ++n) stage_t<T>(Qs + (wave * NQ + n) * 16 * RS, dqt[n], scale, lane);
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NQ; ++n)
        store_tile16<T>(Qs + (wave * NQ + n) * 16 * RS, dq + b * dq_bs + h * D, dq_ld, q0 + (wave * NQ + n) * 16, Nq, lane);
}

// dk / dv[b][key][h*64 + d] = sum over the query splits of the fp32 slabs written by attn_bwd_dkv_kernel<T, true>
template <typename T>
__global__ void attn_dkv_reduce_kernel(const float* __restrict__ ws, T* __restrict__ dk, T* __restrict__ dv, int B, int H,
                                       int Nk, int nsplit, long dk_bs, int dk_ld, long dv_bs, int dv_ld) {
    const long total = (long)2 * B * H * Nk * 16;                    // float4 items of one slab
    const long i = (long)blockIdx.x * NT + threadIdx.x;
    if (i >= total) return;
    float4 a = *reinterpret_cast<const float4*>(ws + i * 4);
    for (int s = 1; s < nsplit; ++s) {
        const float4 u = *reinterpret_cast<const float4*>(ws + ((long)s * total + i) * 4);
        a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
    }
    const int d4 = (int)(i & 15);
    long r = i >> 4;
    const int key = (int)(r % Nk); r /= Nk;
    const int h = (int)(r % H); r /= H;
    const int b = (int)(r % B);
    const bool is_v = r / B != 0;
    T* o = (is_v ? dv + b * dv_bs + (long)key * dv_ld : dk + b * dk_bs + (long)key * dk_ld) + h * D + d4 * 4;
    o[0] = from_f32<T>(a.x); o[1] = from_f32<T>(a.y); o[2] = from_f32<T>(a.z); o[3] = from_f32<T>(a.w);
}

template <typename T> bool aligned_ok(const void* p, long bs, int ld, long nrows) {
    constexpr int CH = ACfg<T>::CH;
    return (((uintptr_t)p) & 15) == 0 && (bs % CH) == 0 && (ld % CH) == 0 &&
           ((nrows - 1) * (long)ld + D) * (long)sizeof(T) < (1L << 31);
}

template <typename T>
int attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Nq, int Nk, long q_bs,
             int q_ld, long k_bs, int k_ld, long v_bs, int v_ld, long o_bs, int o_ld, float scale, int causal, hipStream_t st) {
    if (!aligned_ok<T>(q, q_bs, q_ld, Nq) || !aligned_ok<T>(k, k_bs, k_ld, Nk) || !aligned_ok<T>(v, v_bs, v_ld, Nk) ||
        !aligned_ok<T>(o, o_bs, o_ld, Nq))
        return -1;
    if (causal) {
        hipLaunchKernelGGL((attn_fwd_kernel<T, 1, true>), dim3((Nq + 63) / 64, H, B), dim3(NT), 0, st, (const T*)q, (const T*)k,
                           (const T*)v, (T*)o, lse, H, Nq, Nk, q_bs, q_ld, k_bs, k_ld, v_bs, v_ld, o_bs, o_ld, scale * LOG2E);
        PDMK_CHECK_LAUNCH();
        return 0;
    }
    // 32 queries per wave when there are enough query blocks to fill the chip that way, else 16 (PDMK_ATTN_NQ forces)
    static int forced = -1, dynamic = -1;
    if (dynamic < 0) dynamic = getenv("PDMK_ENV_DYNAMIC") ? 1 : 0;
    if (forced < 0 || dynamic) { const char* e = getenv("PDMK_ATTN_NQ"); forced = e ? atoi(e) : 0; }
    const bool wide = forced ? forced == 2 : ((long)((Nq + 127) / 128) * H * B >= 384 && Nk >= 256);
    if (wide) {
        hipLaunchKernelGGL((attn_fwd_kernel<T, 2>), dim3((Nq + 127) / 128, H, B), dim3(NT), 0, st, (const T*)q, (const T*)k,
                           (const T*)v, (T*)o, lse, H, Nq, Nk, q_bs, q_ld, k_bs, k_ld, v_bs, v_ld, o_bs, o_ld, scale * LOG2E);
    } else {
        hipLaunchKernelGGL((attn_fwd_kernel<T, 1>), dim3((Nq + 63) / 64, H, B), dim3(NT), 0, st, (const T*)q, (const T*)k,
                           (const T*)v, (T*)o, lse, H, Nq, Nk, q_bs, q_ld, k_bs, k_ld, v_bs, v_ld, o_bs, o_ld, scale * LOG2E);
    }
    PDMK_CHECK_LAUNCH();
    return 0;
}

template <typename T>
int attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
             float* delta, void* dq, void* dk, void* dv, int B, int H, int Nq, int Nk, long q_bs, int q_ld, long k_bs,
             int k_ld, long v_bs, int v_ld, long o_bs, int o_ld, long dq_bs, int dq_ld, long dk_bs, int dk_ld,
             long dv_bs, int dv_ld, float scale, float* ws, long ws_elems, hipStream_t st) {
    if (!aligned_ok<T>(q, q_bs, q_ld, Nq) || !aligned_ok<T>(k, k_bs, k_ld, Nk) || !aligned_ok<T>(v, v_bs, v_ld, Nk) ||
        !aligned_ok<T>(o, o_bs, o_ld, Nq) || !aligned_ok<T>(d_o, o_bs, o_ld, Nq) || !aligned_ok<T>(dq, dq_bs, dq_ld, Nq) ||
        !aligned_ok<T>(dk, dk_bs, dk_ld, Nk) || !aligned_ok<T>(dv, dv_bs, dv_ld, Nk))
        return -1;
    // (delta = rowsum(dO * O) is formed inside the dQ kernel's prologue since round 3 - it used to be a launch of its own, 32
    // per step - so the dQ kernel goes first and the dK / dV kernel reads what it published)
    // PDMK_ATTN_NQ (with PDMK_ENV_DYNAMIC) forces 16 (1) / 32 (2) rows per wave in all three kernels
    static int forced = -1, dynamic = -1;
    if (dynamic < 0) dynamic = getenv("PDMK_ENV_DYNAMIC") ? 1 : 0;
    if (forced < 0 || dynamic) { const char* e = getenv("PDMK_ATTN_NQ"); forced = e ? atoi(e) : 0; }
#define PDMK_DQ_ARGS (const T*)q, (const T*)k, (const T*)v, (const T*)o, (const T*)d_o, lse, delta, (T*)dq, H, Nq, Nk, q_bs, q_ld, k_bs, k_ld, \
                     v_bs, v_ld, o_bs, o_ld, dq_bs, dq_ld, scale, scale * LOG2E
    // 32 rows per wave in the backward kernels too, by the forward's rule (bf16 only: the fp32 forms need > 256 registers).  With
    // the MFMA results in ordinary VGPRs (Makefile) the wide forms are the faster ones at N = 4096 (one MI355X, B = 8, H = 5,
    // dQ + dK/dV: 791 -> 649 us; with AGPR accumulators they were the slower ones, 853 vs 874); PDMK_ATTN_NQ_DQ / _DKV force one
    static int f_dq = -1, f_dkv = -1;
    if (f_dq < 0 || dynamic) {
        const char* e = getenv("PDMK_ATTN_NQ_DQ");
        f_dq = e ? atoi(e) : 0;
        e = getenv("PDMK_ATTN_NQ_DKV");
        f_dkv = e ? atoi(e) : 0;
    }
    constexpr bool is_bf16 = std::is_same<T, bf16>::value;
    const int want_dq = f_dq ? f_dq : forced, want_dkv = f_dkv ? f_dkv : forced;
#ifdef PDMK_ATTN_AGPR      // built without -amdgpu-mfma-vgpr-form (csrc/Makefile probe): the narrow forms are the faster ones there
    const bool auto_wide = false;
#else
    const bool auto_wide = true;
#endif
    const bool wide_dq = want_dq ? want_dq == 2 : (auto_wide && is_bf16 && (long)((Nq + 127) / 128) * H * B >= 384 && Nk >= 256);
    const bool wide_dkv = want_dkv ? want_dkv == 2 : (auto_wide && is_bf16 && (long)((Nk + 127) / 128) * H * B >= 384 && Nq >= 256);
    if (wide_dq)
        hipLaunchKernelGGL((attn_bwd_dq_kernel<T, 2>), dim3((Nq + 127) / 128, H, B), dim3(NT), 0, st, PDMK_DQ_ARGS);
    else
        hipLaunchKernelGGL((attn_bwd_dq_kernel<T, 1>), dim3((Nq + 63) / 64, H, B), dim3(NT), 0, st, PDMK_DQ_ARGS);
#undef PDMK_DQ_ARGS
    // few keys (cross-attention): also split the query sweep, or the grid is only key_blocks*H*B workgroups
    const int kblocks = (Nk + 63) / 64, qblocks = (Nq + ACfg<T>::KVB - 1) / ACfg<T>::KVB;
    int nsplit = 1;
    if (ws && kblocks * H * B < 128 && qblocks >= 8) {
        nsplit = min(min(32, qblocks / 4), (256 + kblocks * H * B - 1) / (kblocks * H * B));
        const long per = (long)2 * B * H * Nk * 64;
        if (nsplit > ws_elems / per) nsplit = (int)(ws_elems / per);
    }
#define PDMK_DKV_ARGS (const T*)q, (const T*)k, (const T*)v, (const T*)d_o, lse, delta, (T*)dk, (T*)dv, H, Nq, Nk, q_bs, q_ld, \
                      k_bs, k_ld, v_bs, v_ld, o_bs, o_ld, dk_bs, dk_ld, dv_bs, dv_ld, scale, scale * LOG2E
    if (nsplit > 1) {
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, true, 1>), dim3(kblocks * nsplit, H, B), dim3(NT), 0, st, PDMK_DKV_ARGS, ws,
                           nsplit);
        const long items = (long)2 * B * H * Nk * 16;
        hipLaunchKernelGGL(attn_dkv_reduce_kernel<T>, dim3((unsigned)((items + NT - 1) / NT)), dim3(NT), 0, st, ws, (T*)dk,
                           (T*)dv, B, H, Nk, nsplit, dk_bs, dk_ld, dv_bs, dv_ld);
    } else if (wide_dkv) {         // 32 keys per wave (242 VGPRs, 65 KiB LDS)
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, false, 2>), dim3((Nk + 127) / 128, H, B), dim3(NT), 0, st, PDMK_DKV_ARGS,
                           nullptr, 1);
    } else {
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, false, 1>), dim3(kblocks, H, B), dim3(NT), 0, st, PDMK_DKV_ARGS, nullptr, 1);
    }
#undef PDMK_DKV_ARGS
    PDMK_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int pdmk_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Nq,
                             int Nk, int64_t q_bs, int q_ld, int64_t k_bs, int k_ld, int64_t v_bs, int v_ld,
                             int64_t o_bs, int o_ld, float scale, int dtype, pdmk_stream stream) {
    if (!q || !k || !v || !o || !lse || B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return -1;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PDMK_BF16)
        return attn_fwd<bf16>(q, k, v, o, lse, B, H, Nq, Nk, q_bs, q_ld, k_bs, k_ld, v_bs, v_ld, o_bs, o_ld, scale, 0, st);
    if (dtype == PDMK_F32)
        return attn_fwd<float>(q, k, v, o, lse, B, H, Nq, Nk, q_bs, q_ld, k_bs, k_ld, v_bs, v_ld, o_bs, o_ld, scale, 0, st);
    return -2;
}

extern "C" int pdmk_attn_fwd_causal(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int N,
                                    int64_t q_bs, int q_ld, int64_t k_bs, int k_ld, int64_t v_bs, int v_ld, int64_t o_bs,
                                    int o_ld, float scale, int dtype, pdmk_stream stream) {
    if (!q || !k || !v || !o || !lse || B <= 0 || H <= 0 || N <= 0) return -1;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PDMK_BF16)
        return attn_fwd<bf16>(q, k, v, o, lse, B, H, N, N, q_bs, q_ld, k_bs, k_ld, v_bs, v_ld, o_bs, o_ld, scale, 1, st);
    if (dtype == PDMK_F32)
        return attn_fwd<float>(q, k, v, o, lse, B, H, N, N, q_bs, q_ld, k_bs, k_ld, v_bs, v_ld, o_bs, o_ld, scale, 1, st);
    return -2;
}

extern "C" int pdmk_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                             const float* lse, float* delta, void* dq, void* dk, void* dv, int B, int H, int Nq,
                             int Nk, int64_t q_bs, int q_ld, int64_t k_bs, int k_ld, int64_t v_bs, int v_ld,
                             int64_t o_bs, int o_ld, int64_t dq_bs, int dq_ld, int64_t dk_bs, int dk_ld,
                             int64_t dv_bs, int dv_ld, float scale, float* ws, int64_t ws_elems, int dtype,
                             pdmk_stream stream) {
    if (!q || !k || !v || !o || !d_o || !lse || !delta || !dq || !dk || !dv || B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0)
        return -1;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PDMK_BF16)
        return attn_bwd<bf16>(q, k, v, o, d_o, lse, delta, dq, dk, dv, B, H, Nq, Nk, q_bs, q_ld, k_bs, k_ld, v_bs, v_ld,
                              o_bs, o_ld, dq_bs, dq_ld, dk_bs, dk_ld, dv_bs, dv_ld, scale, ws, (long)ws_elems, st);
    if (dtype == PDMK_F32)
        return attn_bwd<float>(q, k, v, o, d_o, lse, delta, dq, dk, dv, B, H, Nq, Nk, q_bs, q_ld, k_bs, k_ld, v_bs,
                               v_ld, o_bs, o_ld, dq_bs, dq_ld, dk_bs, dk_ld, dv_bs, dv_ld, scale, ws, (long)ws_elems, st);
    return -2;
}
