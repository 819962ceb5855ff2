// Flash-style scaled-dot-product attention for head dim 64 (self-attention over N = HW latent tokens, cross-attention
// over the 77 text tokens), forward + backward, bf16 (MFMA 16x16x32) and exact-fp32 (MFMA 16x16x4) instantiations.
// The N x N score matrix is never materialised: per 64-row query block the kernel walks key blocks with an online
// (base-2) softmax; fp32 statistics; LSE saved for the backward.  Q/K/V/O are addressed with explicit batch/row strides
// so the fused QKV projection output is consumed in place (no head transposes in HBM).
// Backward = delta pre-pass + a dK/dV kernel (one workgroup per 64 keys sweeps the queries) + a dQ kernel (one
// workgroup per 64 queries sweeps the keys): no atomics, bitwise reproducible.
// K / Q / dO tiles are held ONCE in LDS in memory order and read both k-contiguous (ds_read_b128) and transposed
// (ds_read_b64_tr_b16) depending on which product consumes them.
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int D = 64;
constexpr float LOG2E = 1.4426950408889634f;

template <typename T> struct ACfg;
template <> struct ACfg<bf16> {
    static constexpr int CH = 8, RS = 72, KVB = 64;   // RS: LDS row stride of a [rows][64] tile; KVB: streamed rows
};
template <> struct ACfg<float> {
    static constexpr int CH = 4, RS = 68, KVB = 32;
};

// global [rows, 64] (row stride ld) -> LDS [ROWS][RS]; rows >= nvalid are zero-filled
template <typename T, int ROWS>
__device__ __forceinline__ void load_tile(T* lds, const T* g, long ld, int row0, int nrows, int tid) {
    constexpr int CH = ACfg<T>::CH, RS = ACfg<T>::RS, CPR = D / CH;
    for (int c = tid; c < ROWS * CPR; c += NT) {
        const int r = c / CPR, dc = (c % CPR) * CH;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row0 + r < nrows) v = *reinterpret_cast<const uint4*>(g + (long)(row0 + r) * ld + dc);
        *reinterpret_cast<uint4*>(lds + r * RS + dc) = v;
    }
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

// ================================================================================================ forward
template <typename T>
__global__ __launch_bounds__(NT) void attn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                      const T* __restrict__ v, T* __restrict__ o,
                                                      float* __restrict__ lse, int H, int Nq, int Nk, long q_bs,
                                                      int q_ld, long k_bs, int k_ld, long v_bs, int v_ld, long o_bs,
                                                      int o_ld, float scale2) {
    typedef Mma<T> MM;
    typedef ACfg<T> AC;
    constexpr int RS = AC::RS, KVB = AC::KVB, NJ = KVB / 16, RSP = KVB + (sizeof(T) == 2 ? 8 : 4);
    __shared__ __attribute__((aligned(16))) T Qs[64 * RS];
    __shared__ __attribute__((aligned(16))) T Ks[KVB * RS];
    __shared__ __attribute__((aligned(16))) T Vs[KVB * RS];
    __shared__ __attribute__((aligned(16))) T Ps[4 * 16 * RSP];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const T* qg = q + b * q_bs + h * D;
    const T* kg = k + b * k_bs + h * D;
    const T* vg = v + b * v_bs + h * D;

    load_tile<T, 64>(Qs, qg, q_ld, q0, Nq, tid);
    __syncthreads();
    typename MM::frag qf[D / MM::KS];
#pragma unroll
    for (int kk = 0; kk < D / MM::KS; ++kk) qf[kk] = MM::load_rowk(Qs, RS, wave * 16, kk * MM::KS, lane);

    f32x4 oacc[4];
    float mrow[4], lrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { oacc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; mrow[i] = -INFINITY; lrow[i] = 0.f; }
    T* Pw = Ps + wave * 16 * RSP;

    for (int kb = 0; kb < Nk; kb += KVB) {
        __syncthreads();   // previous block's K/V/P reads done
        load_tile<T, KVB>(Ks, kg, k_ld, kb, Nk, tid);
        load_tile<T, KVB>(Vs, vg, v_ld, kb, Nk, tid);
        __syncthreads();
        f32x4 s[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            s[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < D / MM::KS; ++kk)
                s[j] = MM::mma(qf[kk], MM::load_rowk(Ks, RS, j * 16, kk * MM::KS, lane), s[j]);
        }
        // lane: rows q=(lane>>4)*4+r, cols key = kb + 16j + (lane&15)
        float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const bool valid = kb + j * 16 + (lane & 15) < Nk;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[j][r] = valid ? s[j][r] * scale2 : -INFINITY;
                mx[r] = fmaxf(mx[r], s[j][r]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) mx[r] = fmaxf(mx[r], __shfl_xor(mx[r], off, 64));
            const float mn = fmaxf(mrow[r], mx[r]);
            const float alpha = exp2f(mrow[r] - mn);
            mrow[r] = mn;
            lrow[r] *= alpha;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) oacc[dt][r] *= alpha;
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = exp2f(s[j][r] - mrow[r]);
                lrow[r] += p;
                Pw[((lane >> 4) * 4 + r) * RSP + j * 16 + (lane & 15)] = from_f32<T>(p);
            }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KVB / MM::KS; ++kk) {
            const typename MM::frag pf = MM::load_rowk(Pw, RSP, 0, kk * MM::KS, lane);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                oacc[dt] = MM::mma(pf, MM::load_colk(Vs, RS, kk * MM::KS, dt * 16, lane), oacc[dt]);
        }
    }
    // finalize: l over the 16 lanes of a row
    T* Ow = Qs + wave * 16 * RS;   // this wave's own Q rows, free since qf was hoisted
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float l = lrow[r];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) l += __shfl_xor(l, off, 64);
        const float inv = 1.0f / l;
        const int qr = (lane >> 4) * 4 + r;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) Ow[qr * RS + dt * 16 + (lane & 15)] = from_f32<T>(oacc[dt][r] * inv);
        const int qi = q0 + wave * 16 + qr;
        if ((lane & 15) == 0 && qi < Nq) lse[((long)b * H + h) * Nq + qi] = mrow[r] + log2f(l);
    }
    __syncthreads();
    store_tile16<T>(Ow, o + b * o_bs + h * D, o_ld, q0 + wave * 16, Nq, lane);
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

// one workgroup = 64 keys of one (b,h); wave w owns keys 16w..16w+15; sweeps query blocks of QB rows.
template <typename T>
__global__ __launch_bounds__(NT) void attn_bwd_dkv_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                          const T* __restrict__ v, const T* __restrict__ d_o,
                                                          const float* __restrict__ lse,
                                                          const float* __restrict__ delta, T* __restrict__ dk,
                                                          T* __restrict__ dv, int H, int Nq, int Nk, long q_bs,
                                                          int q_ld, long k_bs, int k_ld, long v_bs, int v_ld,
                                                          long o_bs, int o_ld, long dk_bs, int dk_ld, long dv_bs,
                                                          int dv_ld, float scale, float scale2) {
    typedef Mma<T> MM;
    typedef ACfg<T> AC;
    constexpr int RS = AC::RS, QB = AC::KVB, NJ = QB / 16, RSP = QB + (sizeof(T) == 2 ? 8 : 4);
    // region U: K,V staging (hoisted into registers), then reused for the wave-private P^T / dS^T tiles
    __shared__ __attribute__((aligned(16))) T U[2 * 64 * RS];
    __shared__ __attribute__((aligned(16))) T Qs[QB * RS];
    __shared__ __attribute__((aligned(16))) T Os[QB * RS];
    __shared__ float Ls[QB], Ds[QB];
    static_assert(2 * 4 * 16 * RSP <= 2 * 64 * RS, "P/dS tiles must fit the K/V staging region");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const T* qg = q + b * q_bs + h * D;
    const T* og = d_o + b * o_bs + h * D;

    load_tile<T, 64>(U, k + b * k_bs + h * D, k_ld, k0, Nk, tid);
    load_tile<T, 64>(U + 64 * RS, v + b * v_bs + h * D, v_ld, k0, Nk, tid);
    __syncthreads();
    typename MM::frag kf[D / MM::KS], vf[D / MM::KS];
#pragma unroll
    for (int kk = 0; kk < D / MM::KS; ++kk) {
        kf[kk] = MM::load_rowk(U, RS, wave * 16, kk * MM::KS, lane);
        vf[kk] = MM::load_rowk(U + 64 * RS, RS, wave * 16, kk * MM::KS, lane);
    }
    T* Pw = U + wave * 16 * RSP;
    T* Sw = U + 4 * 16 * RSP + wave * 16 * RSP;

    f32x4 dka[4], dva[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { dka[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dva[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // lane: rows key=(lane>>4)*4+r (this wave's 16 keys), cols q = qb + 16j + (lane&15)
    bool kvalid[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) kvalid[r] = k0 + wave * 16 + (lane >> 4) * 4 + r < Nk;

    for (int qb = 0; qb < Nq; qb += QB) {
        __syncthreads();
        load_tile<T, QB>(Qs, qg, q_ld, qb, Nq, tid);
        load_tile<T, QB>(Os, og, o_ld, qb, Nq, tid);
        if (tid < QB) {
            const bool ok = qb + tid < Nq;
            Ls[tid] = ok ? lse[((long)b * H + h) * Nq + qb + tid] : INFINITY;
            Ds[tid] = ok ? delta[((long)b * H + h) * Nq + qb + tid] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < D / MM::KS; ++kk) {
                st = MM::mma(kf[kk], MM::load_rowk(Qs, RS, j * 16, kk * MM::KS, lane), st);
                dp = MM::mma(vf[kk], MM::load_rowk(Os, RS, j * 16, kk * MM::KS, lane), dp);
            }
            const float l2 = Ls[j * 16 + (lane & 15)], dl = Ds[j * 16 + (lane & 15)];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = kvalid[r] ? exp2f(st[r] * scale2 - l2) : 0.f;
                const float ds = p * (dp[r] - dl);
                const int off = ((lane >> 4) * 4 + r) * RSP + j * 16 + (lane & 15);
                Pw[off] = from_f32<T>(p);
                Sw[off] = from_f32<T>(ds);
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < QB / MM::KS; ++kk) {
            const typename MM::frag pf = MM::load_rowk(Pw, RSP, 0, kk * MM::KS, lane);
            const typename MM::frag sf = MM::load_rowk(Sw, RSP, 0, kk * MM::KS, lane);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dva[dt] = MM::mma(pf, MM::load_colk(Os, RS, kk * MM::KS, dt * 16, lane), dva[dt]);
                dka[dt] = MM::mma(sf, MM::load_colk(Qs, RS, kk * MM::KS, dt * 16, lane), dka[dt]);
            }
        }
    }
    __syncthreads();
    // stage through this wave's rows of Qs / Os for 16-byte stores (QB*RS >= 4*16*RS only when QB==64; use U instead)
    T* Kw = U + wave * 16 * RS;
    T* Vw = U + 64 * RS + wave * 16 * RS;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int kr = (lane >> 4) * 4 + r;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            Kw[kr * RS + dt * 16 + (lane & 15)] = from_f32<T>(dka[dt][r] * scale);
            Vw[kr * RS + dt * 16 + (lane & 15)] = from_f32<T>(dva[dt][r]);
        }
    }
    __syncthreads();
    store_tile16<T>(Kw, dk + b * dk_bs + h * D, dk_ld, k0 + wave * 16, Nk, lane);
    store_tile16<T>(Vw, dv + b * dv_bs + h * D, dv_ld, k0 + wave * 16, Nk, lane);
}

// one workgroup = 64 queries of one (b,h); wave w owns 16 rows; sweeps key blocks.
template <typename T>
__global__ __launch_bounds__(NT) void attn_bwd_dq_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                         const T* __restrict__ v, const T* __restrict__ d_o,
                                                         const float* __restrict__ lse,
                                                         const float* __restrict__ delta, T* __restrict__ dq, int H,
                                                         int Nq, int Nk, long q_bs, int q_ld, long k_bs, int k_ld,
                                                         long v_bs, int v_ld, long o_bs, int o_ld, long dq_bs,
                                                         int dq_ld, float scale, float scale2) {
    typedef Mma<T> MM;
    typedef ACfg<T> AC;
    constexpr int RS = AC::RS, KVB = AC::KVB, NJ = KVB / 16, RSP = KVB + (sizeof(T) == 2 ? 8 : 4);
    __shared__ __attribute__((aligned(16))) T U[2 * 64 * RS];   // Q, dO staging -> dS tiles -> dQ staging
    __shared__ __attribute__((aligned(16))) T Ks[KVB * RS];
    __shared__ __attribute__((aligned(16))) T Vs[KVB * RS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const T* kg = k + b * k_bs + h * D;
    const T* vg = v + b * v_bs + h * D;

    load_tile<T, 64>(U, q + b * q_bs + h * D, q_ld, q0, Nq, tid);
    load_tile<T, 64>(U + 64 * RS, d_o + b * o_bs + h * D, o_ld, q0, Nq, tid);
    __syncthreads();
    typename MM::frag qf[D / MM::KS], of[D / MM::KS];
#pragma unroll
    for (int kk = 0; kk < D / MM::KS; ++kk) {
        qf[kk] = MM::load_rowk(U, RS, wave * 16, kk * MM::KS, lane);
        of[kk] = MM::load_rowk(U + 64 * RS, RS, wave * 16, kk * MM::KS, lane);
    }
    float l2[4], dl[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qi = q0 + wave * 16 + (lane >> 4) * 4 + r;
        l2[r] = qi < Nq ? lse[((long)b * H + h) * Nq + qi] : INFINITY;
        dl[r] = qi < Nq ? delta[((long)b * H + h) * Nq + qi] : 0.f;
    }
    T* Sw = U + wave * 16 * RSP;
    f32x4 dqa[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dqa[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kb = 0; kb < Nk; kb += KVB) {
        __syncthreads();
        load_tile<T, KVB>(Ks, kg, k_ld, kb, Nk, tid);
        load_tile<T, KVB>(Vs, vg, v_ld, kb, Nk, tid);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < D / MM::KS; ++kk) {
                s = MM::mma(qf[kk], MM::load_rowk(Ks, RS, j * 16, kk * MM::KS, lane), s);
                dp = MM::mma(of[kk], MM::load_rowk(Vs, RS, j * 16, kk * MM::KS, lane), dp);
            }
            const bool valid = kb + j * 16 + (lane & 15) < Nk;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = valid ? exp2f(s[r] * scale2 - l2[r]) : 0.f;
                Sw[((lane >> 4) * 4 + r) * RSP + j * 16 + (lane & 15)] = from_f32<T>(p * (dp[r] - dl[r]));
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < KVB / MM::KS; ++kk) {
            const typename MM::frag sf = MM::load_rowk(Sw, RSP, 0, kk * MM::KS, lane);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                dqa[dt] = MM::mma(sf, MM::load_colk(Ks, RS, kk * MM::KS, dt * 16, lane), dqa[dt]);
        }
    }
    __syncthreads();
    T* Qw = U + 64 * RS + wave * 16 * RS;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qr = (lane >> 4) * 4 + r;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) Qw[qr * RS + dt * 16 + (lane & 15)] = from_f32<T>(dqa[dt][r] * scale);
    }
    __syncthreads();
    store_tile16<T>(Qw, dq + b * dq_bs + h * D, dq_ld, q0 + wave * 16, Nq, lane);
}

template <typename T> bool aligned_ok(const void* p, long bs, int ld) {
    constexpr int CH = ACfg<T>::CH;
    return (((uintptr_t)p) & 15) == 0 && (bs % CH) == 0 && (ld % CH) == 0;
}

template <typename T>
int attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Nq, int Nk, long q_bs,
             int q_ld, long k_bs, int k_ld, long v_bs, int v_ld, long o_bs, int o_ld, float scale, hipStream_t st) {
    if (!aligned_ok<T>(q, q_bs, q_ld) || !aligned_ok<T>(k, k_bs, k_ld) || !aligned_ok<T>(v, v_bs, v_ld) ||
        !aligned_ok<T>(o, o_bs, o_ld))
        return -1;
    dim3 grid((Nq + 63) / 64, H, B);
    hipLaunchKernelGGL(attn_fwd_kernel<T>, grid, dim3(NT), 0, st, (const T*)q, (const T*)k, (const T*)v, (T*)o, lse, H,
                       Nq, Nk, q_bs, q_ld, k_bs, k_ld, v_bs, v_ld, o_bs, o_ld, scale * LOG2E);
    PDMK_CHECK_LAUNCH();
    return 0;
}

template <typename T>
int attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
             float* delta, void* dq, void* dk, void* dv, int B, int H, int Nq, int Nk, long q_bs, int q_ld, long k_bs,
             int k_ld, long v_bs, int v_ld, long o_bs, int o_ld, long dq_bs, int dq_ld, long dk_bs, int dk_ld,
             long dv_bs, int dv_ld, float scale, hipStream_t st) {
    if (!aligned_ok<T>(q, q_bs, q_ld) || !aligned_ok<T>(k, k_bs, k_ld) || !aligned_ok<T>(v, v_bs, v_ld) ||
        !aligned_ok<T>(o, o_bs, o_ld) || !aligned_ok<T>(d_o, o_bs, o_ld) || !aligned_ok<T>(dq, dq_bs, dq_ld) ||
        !aligned_ok<T>(dk, dk_bs, dk_ld) || !aligned_ok<T>(dv, dv_bs, dv_ld))
        return -1;
    const long total = (long)B * H * Nq;
    hipLaunchKernelGGL(attn_delta_kernel<T>, dim3((int)min(4096L, (total + NT - 1) / NT)), dim3(NT), 0, st,
                       (const T*)o, (const T*)d_o, delta, H, Nq, o_bs, o_ld, total);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<T>, dim3((Nk + 63) / 64, H, B), dim3(NT), 0, st, (const T*)q, (const T*)k,
                       (const T*)v, (const T*)d_o, lse, delta, (T*)dk, (T*)dv, H, Nq, Nk, q_bs, q_ld, k_bs, k_ld, v_bs,
                       v_ld, o_bs, o_ld, dk_bs, dk_ld, dv_bs, dv_ld, scale, scale * LOG2E);
    hipLaunchKernelGGL(attn_bwd_dq_kernel<T>, dim3((Nq + 63) / 64, H, B), dim3(NT), 0, st, (const T*)q, (const T*)k,
                       (const T*)v, (const T*)d_o, lse, delta, (T*)dq, H, Nq, Nk, q_bs, q_ld, k_bs, k_ld, v_bs, v_ld,
                       o_bs, o_ld, dq_bs, dq_ld, scale, scale * LOG2E);
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
        return attn_fwd<bf16>(q, k, v, o, lse, B, H, Nq, Nk, q_bs, q_ld, k_bs, k_ld, v_bs, v_ld, o_bs, o_ld, scale, st);
    if (dtype == PDMK_F32)
        return attn_fwd<float>(q, k, v, o, lse, B, H, Nq, Nk, q_bs, q_ld, k_bs, k_ld, v_bs, v_ld, o_bs, o_ld, scale, st);
    return -2;
}

extern "C" int pdmk_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                             const float* lse, float* delta, void* dq, void* dk, void* dv, int B, int H, int Nq,
                             int Nk, int64_t q_bs, int q_ld, int64_t k_bs, int k_ld, int64_t v_bs, int v_ld,
                             int64_t o_bs, int o_ld, int64_t dq_bs, int dq_ld, int64_t dk_bs, int dk_ld,
                             int64_t dv_bs, int dv_ld, float scale, int dtype, pdmk_stream stream) {
    if (!q || !k || !v || !o || !d_o || !lse || !delta || !dq || !dk || !dv || B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0)
        return -1;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PDMK_BF16)
        return attn_bwd<bf16>(q, k, v, o, d_o, lse, delta, dq, dk, dv, B, H, Nq, Nk, q_bs, q_ld, k_bs, k_ld, v_bs, v_ld,
                              o_bs, o_ld, dq_bs, dq_ld, dk_bs, dk_ld, dv_bs, dv_ld, scale, st);
    if (dtype == PDMK_F32)
        return attn_bwd<float>(q, k, v, o, d_o, lse, delta, dq, dk, dv, B, H, Nq, Nk, q_bs, q_ld, k_bs, k_ld, v_bs,
                               v_ld, o_bs, o_ld, dq_bs, dq_ld, dk_bs, dk_ld, dv_bs, dv_ld, scale, st);
    return -2;
}
