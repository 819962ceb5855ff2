// GroupNorm(+SiLU) and LayerNorm, forward and backward, NHWC / token-major, HBM-bandwidth kernels.
// All loads/stores are 16-byte chunks along the channel dim (rows are read fully coalesced); statistics are fp32
// per thread, fp32 LDS atomics per block, then double atomics across blocks so that E[x^2]-mean^2 is formed in double.
#include "common.h"
#include "vec.h"

namespace {

constexpr int NT = 256;
constexpr int MAXS = 3;   // chunk slots per thread: C/chunk <= 3*256

// Second stage of the per-channel parameter-gradient reductions: part[b][2][n] (one slab per first-stage block, plain
// stores) -> out0[n] += sum_b part[b][0][:], out1[n] += sum_b part[b][1][:].  Many blocks hammering the same few hundred
// addresses with float atomics serialise at the memory side (measured 130 us for a 10 us LayerNorm backward); two
// stages with at most 8 adds per address do not.
__global__ void reduce_partials_kernel(const float* __restrict__ part, int nblk, int n, float* __restrict__ out0,
                                       float* __restrict__ out1) {
    const int j = blockIdx.x * NT + threadIdx.x;
    if (j >= 2 * n) return;
    const int per = (nblk + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = b0;
    for (; b + 3 < b1; b += 4) {
        s0 += part[(long)b * 2 * n + j];
        s1 += part[(long)(b + 1) * 2 * n + j];
        s2 += part[(long)(b + 2) * 2 * n + j];
        s3 += part[(long)(b + 3) * 2 * n + j];
    }
    for (; b < b1; ++b) s0 += part[(long)b * 2 * n + j];
    const float s = (s0 + s1) + (s2 + s3);
    if (b1 > b0) unsafeAtomicAdd(j < n ? out0 + j : out1 + (j - n), s);
}
inline void launch_reduce_partials(const float* part, int nblk, int n, float* out0, float* out1, hipStream_t st) {
    if (!out0 && !out1) return;          // deferred: the caller reduces this slab later with pdmk_reduce_partials_group
    const int slices = nblk >= 256 ? 8 : (nblk >= 32 ? 4 : 1);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((2 * n + NT - 1) / NT, slices), dim3(NT), 0, st, part, nblk, n, out0,
                       out1);
}

// The same second stage for up to PDMK_PARTIAL_GROUP_MAX slabs in ONE launch (the per-layer launches are ~7 us each for
// ~1 us of traffic, and a training step has ~110 of them): blockIdx.y = item * 8 + slice.
struct PartialGroup {
    pdmk_partial_item it[PDMK_PARTIAL_GROUP_MAX];
};
__global__ void reduce_partials_group_kernel(PartialGroup gr) {
    const pdmk_partial_item it = gr.it[blockIdx.y >> 3];
    const int slice = blockIdx.y & 7, n = it.n, nblk = it.nblk;
    const int j = blockIdx.x * NT + threadIdx.x;
    if (j >= 2 * n) return;
    const int per = (nblk + 7) / 8;
    const int b0 = slice * per, b1 = min(nblk, b0 + per);
    const float* part = it.part;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = b0;
    for (; b + 3 < b1; b += 4) {
        s0 += part[(long)b * 2 * n + j];
        s1 += part[(long)(b + 1) * 2 * n + j];
        s2 += part[(long)(b + 2) * 2 * n + j];
        s3 += part[(long)(b + 3) * 2 * n + j];
    }
    for (; b < b1; ++b) s0 += part[(long)b * 2 * n + j];
    const float s = (s0 + s1) + (s2 + s3);
    if (b1 > b0) unsafeAtomicAdd(j < n ? it.out0 + j : it.out1 + (j - n), s);
}

// ------------------------------------------------------------------------------------------------ GroupNorm
// Thread map: a block sweeps rows [r0, r1) of one image; thread (cb, ro) owns 16-byte chunk cb (+ tpr*slot) of row
// ro (+ rif*k).  UNR rows are loaded per iteration before any of them is used (UNR independent 16-byte loads in flight
// per thread: these kernels are pure HBM streams, and a CU needs tens of KiB in flight to reach HBM bandwidth).
// Reductions never use atomics: per-block partials go to a slab with plain stores and the consumer kernel's prologue
// adds the (<= 64) slabs of its image in double (E[x^2] - mean^2 is formed in double).
struct GnMap2 {
    int tpr, rif;   // threads per row (= chunks per row, capped at 256), rows in flight
};
__host__ __device__ inline GnMap2 gn_map2(int nchunks) {
    const int tpr = nchunks < NT ? nchunks : NT;
    return GnMap2{tpr, NT / tpr};
}
constexpr int GN_MAXC = MAXS * NT * 8;   // channel capacity of the LDS reduction image (bf16 chunks)
constexpr int GN_MAXBLK = 64;            // stats blocks per image

template <typename T, int UNR, int SLOTS>
__global__ __launch_bounds__(NT) void gn_stats_kernel(const T* __restrict__ x, float* __restrict__ part, int HW, int C,
                                                      int ldx, int G, int gs, int rows_per_blk) {
    constexpr int V = Vec<T>::N;
    const int nchunks = C / V;
    const GnMap2 mp = gn_map2(nchunks);
    const int tid = threadIdx.x, cb = tid % mp.tpr, ro = tid / mp.tpr;
    const bool active = ro < mp.rif;
    const int b = blockIdx.y;
    const int r0 = blockIdx.x * rows_per_blk, r1 = min(HW, r0 + rows_per_blk);
    float s[SLOTS][V], q[SLOTS][V];
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl)
#pragma unroll
        for (int e = 0; e < V; ++e) s[sl][e] = q[sl][e] = 0.f;
    if (active) {
        for (int r = r0 + ro; r < r1; r += mp.rif * UNR) {
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                const int c = cb + mp.tpr * sl;
                if (c < nchunks) {
                    float f[UNR][V];
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const int rr = r + u * mp.rif;
                        if (rr < r1) Vec<T>::load(x + ((long)b * HW + rr) * ldx + c * V, f[u]);
                        else {
#pragma unroll
                            for (int e = 0; e < V; ++e) f[u][e] = 0.f;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < UNR; ++u)
#pragma unroll
                        for (int e = 0; e < V; ++e) { s[sl][e] += f[u][e]; q[sl][e] += f[u][e] * f[u][e]; }
                }
            }
        }
    }
    __shared__ float red[2][GN_MAXC / 8 * V];        // [rif][C] images of the per-thread sums
    if (active) {
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            const int c = cb + mp.tpr * sl;
            if (c < nchunks) {
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    red[0][ro * C + c * V + e] = s[sl][e];
                    red[1][ro * C + c * V + e] = q[sl][e];
                }
            }
        }
    }
    __syncthreads();
    // 256 threads: (group g = tid % 64, quarter = tid / 64) -> partial over a quarter of the group's (row, channel) cells
    const int g = tid & 63, qt = tid >> 6;
    float ps = 0.f, pq = 0.f;
    if (g < G) {
        const int cells = mp.rif * gs;
        for (int i = qt; i < cells; i += 4) {
            const int rr = i / gs, ch = g * gs + (i - rr * gs);
            ps += red[0][rr * C + ch];
            pq += red[1][rr * C + ch];
        }
    }
    __syncthreads();
    red[0][tid] = ps;
    red[1][tid] = pq;
    __syncthreads();
    if (tid < G) {
        float* slab = part + (((long)b * gridDim.x + blockIdx.x) * G + tid) * 2;
        slab[0] = (red[0][tid] + red[0][tid + 64]) + (red[0][tid + 128] + red[0][tid + 192]);
        slab[1] = (red[1][tid] + red[1][tid + 64]) + (red[1][tid + 128] + red[1][tid + 192]);
    }
}

// sums the nblk per-block (sum0, sum1) slabs of image b into LDS out0[g], out1[g] (double); all 256 threads take part
__device__ __forceinline__ void gn_sum_slabs(const float* __restrict__ part, int b, int nblk, int G, double* out0,
                                             double* out1, double (*scr)[NT]) {
    const int tid = threadIdx.x, g = tid & 63, qt = tid >> 6;
    double a0 = 0.0, a1 = 0.0;
    if (g < G)
        for (int k = qt; k < nblk; k += 4) {
            const float* slab = part + (((long)b * nblk + k) * G + g) * 2;
            a0 += (double)slab[0];
            a1 += (double)slab[1];
        }
    scr[0][tid] = a0;
    scr[1][tid] = a1;
    __syncthreads();
    if (tid < G) {
        out0[tid] = (scr[0][tid] + scr[0][tid + 64]) + (scr[0][tid + 128] + scr[0][tid + 192]);
        out1[tid] = (scr[1][tid] + scr[1][tid + 64]) + (scr[1][tid + 128] + scr[1][tid + 192]);
    }
    __syncthreads();
}

template <typename T, int UNR, int SLOTS>
__global__ __launch_bounds__(NT) void gn_apply_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ part, int nblk,
                                                      float* __restrict__ stats, int HW,
                                                      int C, int ldx, int ldy, int G, int gs, float eps, int silu,
                                                      int rows_per_blk, int cs_ld = 0) {
    constexpr int V = Vec<T>::N;
    typedef typename Vec<T>::raw Raw;
    const int nchunks = C / V;
    const GnMap2 mp = gn_map2(nchunks);
    const int tid = threadIdx.x, cb = tid % mp.tpr, ro = tid / mp.tpr;
    const int b = blockIdx.y;
    const int r0 = blockIdx.x * rows_per_blk, r1 = min(HW, r0 + rows_per_blk);
    const int cr = G * gs;
    __shared__ double scr[2][NT], d0[64], d1[64];
    __shared__ float lm[64], lr[64];
    // Everything that does not depend on the statistics is issued FIRST - this thread's gamma / beta chunks and its first sweep of
    // rows - so that those loads are in flight while the block works out mean / rstd (a chain of an L2 read, an LDS reduction and
    // a double-precision sqrt: 3-4 us per block, which every block of the grid pays at the same time; with the loads behind it the
    // kernel streamed at 2.5 TB/s where a plain copy of the same tensor reaches 6.4, tools/gn_bench.py).
    const bool active = ro < mp.rif;
    const bool vec_gb = ((reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15) == 0;
    // (the statistics' own loads are the very first: loads return in issue order, so behind the row sweep they would wait for it)
    long long l0 = 0, h0 = 0, l1 = 0, h1 = 0;
    int cs_bad = 0;
    if (nblk < 0 && (tid >> 2) < G) {
        // epilogue statistics (pdmk_gemm_args.colstat): `part` = this tensor's first column of the [B][4][cs_ld] per-(image, column)
        // fixed-point sums (two limbs each); group g = its gs columns.  4 threads per group (G <= 64): thread (g, p) adds the limbs of
        // columns p, p + 4, ... in registers, the partners are folded with two shuffles below.  Exact integer sums, no atomics
        // (LDS atomics here put 40 same-address adds of one group into one wave instruction: 8 us per block)
        const long long* cg = reinterpret_cast<const long long*>(part) + (long)b * 4 * cs_ld + (tid >> 2) * gs;
#pragma unroll 4
        for (int j = tid & 3; j < gs; j += 4) {
            const long long q3 = cg[3 * cs_ld + j];
            l0 += cg[j];
            h0 += cg[cs_ld + j];
            l1 += cg[2 * cs_ld + j];
            h1 += cs_strip(q3);
            cs_bad |= cs_flagged(q3) ? 1 : 0;                   // a producer's partial sum was NaN / inf (common.h cs_add)
        }
    }
    float gmv[SLOTS][V], btv[SLOTS][V];
    Raw pre[SLOTS][UNR];
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        const int c = cb + mp.tpr * sl;
        if (active && c < nchunks && c * V + V <= cr && vec_gb) {       // whole chunk inside the normalised channels: 16-byte loads
#pragma unroll
            for (int e = 0; e < V; e += 4) {
                const float4 g4 = *reinterpret_cast<const float4*>(gamma + c * V + e), b4 = *reinterpret_cast<const float4*>(beta + c * V + e);
                gmv[sl][e] = g4.x; gmv[sl][e + 1] = g4.y; gmv[sl][e + 2] = g4.z; gmv[sl][e + 3] = g4.w;
                btv[sl][e] = b4.x; btv[sl][e + 1] = b4.y; btv[sl][e + 2] = b4.z; btv[sl][e + 3] = b4.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int ch = c * V + e;
                const bool ok = active && c < nchunks && ch < cr;
                gmv[sl][e] = ok ? gamma[ch] : 0.f;
                btv[sl][e] = ok ? beta[ch] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int rr = r0 + ro + u * mp.rif;
            if (active && c < nchunks && rr < r1) pre[sl][u] = Vec<T>::load_raw(x + ((long)b * HW + rr) * ldx + c * V);
        }
    }
    if (nblk < 0) {
        const int g = tid >> 2, pth = tid & 3;
#pragma unroll
        for (int m = 1; m < 4; m <<= 1) {
            l0 += __shfl_xor(l0, m, 64);
            h0 += __shfl_xor(h0, m, 64);
            l1 += __shfl_xor(l1, m, 64);
            h1 += __shfl_xor(h1, m, 64);
        }
        cs_bad |= __shfl_xor(cs_bad, 1, 64);
        cs_bad |= __shfl_xor(cs_bad, 2, 64);
        if (g < G && pth == 0) {                                 // non-finite input -> non-finite statistics, as a statistics pass gives
            d0[g] = cs_bad ? __builtin_nan("") : cs_value(l0, h0);
            d1[g] = cs_bad ? __builtin_nan("") : cs_value(l1, h1);
        }
        __syncthreads();
    } else {
        gn_sum_slabs(part, b, nblk, G, d0, d1, scr);
    }
    if (tid < G) {
        // (double where the cancellation is: E[x^2] - mean^2 - INCLUDING 1 / n, whose error multiplies mean^2: the fp32 v_rcp seeds
        // it and two Newton steps in double bring it to ~1e-16 relative (a 6e-8 inv_n would leave 6e-8 * mean^2 / var in the
        // variance of a channel group with |mean| >> std - ADVICE r3).  Only the square root stays fp32 - v_rsq refined once: the
        // fp64 div + sqrt this replaces were ~1 us of every block's prologue, two double fma pairs are not)
        const double nd = (double)HW * (double)gs;
        double inv_n = (double)__frcp_rn((float)nd);
        inv_n = inv_n * (2.0 - nd * inv_n);
        inv_n = inv_n * (2.0 - nd * inv_n);
        const double mean = d0[tid] * inv_n;
        double var = d1[tid] * inv_n - mean * mean;
        if (var < 0) var = 0;
        const float ve = (float)(var + (double)eps);
        float rstd = __frsqrt_rn(ve);
        rstd = rstd * (1.5f - 0.5f * ve * rstd * rstd);
        lm[tid] = (float)mean;
        lr[tid] = rstd;
        if (blockIdx.x == 0) {
            stats[((long)b * G + tid) * 2] = (float)mean;
            stats[((long)b * G + tid) * 2 + 1] = rstd;
        }
    }
    __syncthreads();
    if (!active) return;
    float sc[SLOTS][V], sh[SLOTS][V];   // y = x*sc + sh  (pad channels: 0)
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        const int c = cb + mp.tpr * sl;
        int g = (c * V) / gs, rem = c * V - g * gs;          // one division per chunk; the group index then steps with the channel
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int ch = c * V + e;
            sc[sl][e] = sh[sl][e] = 0.f;
            if (c < nchunks && ch < cr) {
                const float a = lr[g] * gmv[sl][e];
                sc[sl][e] = a;
                sh[sl][e] = btv[sl][e] - lm[g] * a;
            }
            if (++rem == gs) { rem = 0; ++g; }
        }
    }
    const int step = mp.rif * UNR;
    for (int r = r0 + ro; r < r1; r += step) {
        Raw cur[SLOTS][UNR];
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            const int c = cb + mp.tpr * sl;
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                cur[sl][u] = pre[sl][u];
                const int rn = r + step + u * mp.rif;             // the next sweep's loads go out before this sweep's arithmetic
                if (c < nchunks && rn < r1) pre[sl][u] = Vec<T>::load_raw(x + ((long)b * HW + rn) * ldx + c * V);
            }
        }
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            const int c = cb + mp.tpr * sl;
            if (c < nchunks) {
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int rr = r + u * mp.rif;
                    if (rr < r1) {
                        float f[V];
                        Vec<T>::unpack(cur[sl][u], f);
#pragma unroll
                        for (int e = 0; e < V; ++e) {
                            const float z = f[e] * sc[sl][e] + sh[sl][e];
                            f[e] = silu ? silu_f(z) : z;
                        }
                        Vec<T>::store(y + ((long)b * HW + rr) * ldy + c * V, f);
                    }
                }
            }
        }
    }
}

// backward pass 1: per (b,g) s1 = sum gamma*dz, s2 = sum gamma*dz*xhat (slab `gpart`); per channel dgamma += sum
// dz*xhat, dbeta += sum dz (slab `part`, second stage reduce_partials_kernel)
template <typename T, int UNR, int SLOTS>
__global__ __launch_bounds__(NT) void gn_bwd_stats_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const float* __restrict__ stats, float* __restrict__ part,
                                                          float* __restrict__ gpart, int HW,
                                                          int C, int ldx, int lddy, int G, int gs, int silu,
                                                          int rows_per_blk) {
    constexpr int V = Vec<T>::N;
    const int nchunks = C / V;
    const GnMap2 mp = gn_map2(nchunks);
    const int tid = threadIdx.x, cb = tid % mp.tpr, ro = tid / mp.tpr;
    const bool active = ro < mp.rif;
    const int b = blockIdx.y;
    const int r0 = blockIdx.x * rows_per_blk, r1 = min(HW, r0 + rows_per_blk);
    const int cr = G * gs;
    typedef typename Vec<T>::raw Raw;
    float a1[SLOTS][V], a2[SLOTS][V], mean[SLOTS][V], rstd[SLOTS][V], gm[SLOTS][V], bt[SLOTS][V];
    // as gn_apply_kernel: 16-byte gamma / beta loads and the first sweep of rows go out first, (mean, rstd) through LDS (one
    // coalesced load per group instead of two scalar loads and a division per channel), then a register ping-pong over the sweeps
    __shared__ float sm[64], sr[64];
    const bool vec_gb = ((reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15) == 0;
    Raw px[SLOTS][UNR], pd[SLOTS][UNR];
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        const int c = cb + mp.tpr * sl;
        if (active && c < nchunks && c * V + V <= cr && vec_gb) {
#pragma unroll
            for (int e = 0; e < V; e += 4) {
                const float4 g4 = *reinterpret_cast<const float4*>(gamma + c * V + e), b4 = *reinterpret_cast<const float4*>(beta + c * V + e);
                gm[sl][e] = g4.x; gm[sl][e + 1] = g4.y; gm[sl][e + 2] = g4.z; gm[sl][e + 3] = g4.w;
                bt[sl][e] = b4.x; bt[sl][e + 1] = b4.y; bt[sl][e + 2] = b4.z; bt[sl][e + 3] = b4.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int ch = c * V + e;
                const bool ok = active && c < nchunks && ch < cr;
                gm[sl][e] = ok ? gamma[ch] : 0.f;
                bt[sl][e] = ok ? beta[ch] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int rr = r0 + ro + u * mp.rif;
            if (active && c < nchunks && rr < r1) {
                px[sl][u] = Vec<T>::load_raw(x + ((long)b * HW + rr) * ldx + c * V);
                pd[sl][u] = Vec<T>::load_raw(dy + ((long)b * HW + rr) * lddy + c * V);
            }
        }
    }
    if (tid < G) {
        sm[tid] = stats[((long)b * G + tid) * 2];
        sr[tid] = stats[((long)b * G + tid) * 2 + 1];
    }
    __syncthreads();
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        const int c = cb + mp.tpr * sl;
        int g = (c * V) / gs, rem = c * V - g * gs;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int ch = c * V + e;
            a1[sl][e] = a2[sl][e] = 0.f;
            mean[sl][e] = rstd[sl][e] = 0.f;
            if (c < nchunks && ch < cr) {
                mean[sl][e] = sm[g];
                rstd[sl][e] = sr[g];
            } else {
                gm[sl][e] = bt[sl][e] = 0.f;
            }
            if (++rem == gs) { rem = 0; ++g; }
        }
    }
    if (active) {
        const int step = mp.rif * UNR;
        for (int r = r0 + ro; r < r1; r += step) {
            Raw cx[SLOTS][UNR], cd[SLOTS][UNR];
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                const int c = cb + mp.tpr * sl;
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    cx[sl][u] = px[sl][u];
                    cd[sl][u] = pd[sl][u];
                    const int rn = r + step + u * mp.rif;
                    if (c < nchunks && rn < r1) {
                        px[sl][u] = Vec<T>::load_raw(x + ((long)b * HW + rn) * ldx + c * V);
                        pd[sl][u] = Vec<T>::load_raw(dy + ((long)b * HW + rn) * lddy + c * V);
                    }
                }
            }
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                const int c = cb + mp.tpr * sl;
                if (c < nchunks) {
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const int rr = r + u * mp.rif;
                        if (rr < r1) {
                            float f[V], d[V];
                            Vec<T>::unpack(cx[sl][u], f);
                            Vec<T>::unpack(cd[sl][u], d);
#pragma unroll
                            for (int e = 0; e < V; ++e) {
                                const float xh = (f[e] - mean[sl][e]) * rstd[sl][e];
                                float dz = d[e];
                                if (silu) dz *= silu_grad_f(xh * gm[sl][e] + bt[sl][e]);
                                a1[sl][e] += dz;
                                a2[sl][e] += dz * xh;
                            }
                        }
                    }
                }
            }
        }
    }
    __shared__ float red[2][GN_MAXC / 8 * V];        // [rif][C] images of the per-thread sums
    if (active) {
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            const int c = cb + mp.tpr * sl;
            if (c < nchunks) {
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    red[0][ro * C + c * V + e] = a1[sl][e];
                    red[1][ro * C + c * V + e] = a2[sl][e];
                }
            }
        }
    }
    __syncthreads();
    // per-channel sums over the rows in flight -> row 0 of the image (+ the dgamma/dbeta slab of this block)
    float* slab = part + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 2 * cr;                 // [dgamma | dbeta]
    for (int ch = tid; ch < cr; ch += NT) {
        float t1 = 0.f, t2 = 0.f;
        for (int rr = 0; rr < mp.rif; ++rr) { t1 += red[0][rr * C + ch]; t2 += red[1][rr * C + ch]; }
        slab[ch] = t2;
        slab[cr + ch] = t1;
        const float gmc = gamma[ch];
        red[0][ch] = gmc * t1;      // only row 0 is read below; rows >= 1 of other channels are not touched by this thread
        red[1][ch] = gmc * t2;
    }
    __syncthreads();
    const int g = tid & 63, qt = tid >> 6;
    float ps = 0.f, pq = 0.f;
    if (g < G)
        for (int i = qt; i < gs; i += 4) { ps += red[0][g * gs + i]; pq += red[1][g * gs + i]; }
    __syncthreads();
    red[0][tid] = ps;
    red[1][tid] = pq;
    __syncthreads();
    if (tid < G) {
        float* gs_ = gpart + (((long)b * gridDim.x + blockIdx.x) * G + tid) * 2;
        gs_[0] = (red[0][tid] + red[0][tid + 64]) + (red[0][tid + 128] + red[0][tid + 192]);
        gs_[1] = (red[1][tid] + red[1][tid + 64]) + (red[1][tid + 128] + red[1][tid + 192]);
    }
}

template <typename T, int UNR, int SLOTS>
__global__ __launch_bounds__(NT) void gn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                          T* __restrict__ dx, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const float* __restrict__ stats,
                                                          const float* __restrict__ gpart, int nblk, int HW, int C,
                                                          int ldx, int lddy, int lddx, int G, int gs, int silu,
                                                          int accumulate, int rows_per_blk, const T* __restrict__ add,
                                                          int ldadd) {
    constexpr int V = Vec<T>::N;
    typedef typename Vec<T>::raw Raw;
    const int nchunks = C / V;
    const GnMap2 mp = gn_map2(nchunks);
    const int tid = threadIdx.x, cb = tid % mp.tpr, ro = tid / mp.tpr;
    const int b = blockIdx.y;
    const int r0 = blockIdx.x * rows_per_blk, r1 = min(HW, r0 + rows_per_blk);
    const int cr = G * gs;
    __shared__ double scr[2][NT], d0[64], d1[64];
    __shared__ float sm[64], sr[64];
    // as gn_apply_kernel: what does not depend on the group sums goes out first (gamma / beta chunks, the (mean, rstd) pairs, the
    // first sweep of x / dy rows), so that it is in flight while the block adds up the slabs
    const bool active = ro < mp.rif;
    const bool vec_gb = ((reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15) == 0;
    float gm[SLOTS][V], bt[SLOTS][V];
    Raw px[SLOTS][UNR], pd[SLOTS][UNR];
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        const int c = cb + mp.tpr * sl;
        if (active && c < nchunks && c * V + V <= cr && vec_gb) {
#pragma unroll
            for (int e = 0; e < V; e += 4) {
                const float4 g4 = *reinterpret_cast<const float4*>(gamma + c * V + e), b4 = *reinterpret_cast<const float4*>(beta + c * V + e);
                gm[sl][e] = g4.x; gm[sl][e + 1] = g4.y; gm[sl][e + 2] = g4.z; gm[sl][e + 3] = g4.w;
                bt[sl][e] = b4.x; bt[sl][e + 1] = b4.y; bt[sl][e + 2] = b4.z; bt[sl][e + 3] = b4.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int ch = c * V + e;
                const bool ok = active && c < nchunks && ch < cr;
                gm[sl][e] = ok ? gamma[ch] : 0.f;
                bt[sl][e] = ok ? beta[ch] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int rr = r0 + ro + u * mp.rif;
            if (active && c < nchunks && rr < r1) {
                px[sl][u] = Vec<T>::load_raw(x + ((long)b * HW + rr) * ldx + c * V);
                pd[sl][u] = Vec<T>::load_raw(dy + ((long)b * HW + rr) * lddy + c * V);
            }
        }
    }
    if (tid < G) {
        sm[tid] = stats[((long)b * G + tid) * 2];
        sr[tid] = stats[((long)b * G + tid) * 2 + 1];
    }
    gn_sum_slabs(gpart, b, nblk, G, d0, d1, scr);          // (ends with a barrier: sm / sr are visible too)
    if (!active) return;
    const float invn = 1.0f / ((float)HW * (float)gs);
    float mean[SLOTS][V], rstd[SLOTS][V], c1[SLOTS][V], c2[SLOTS][V];
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        const int c = cb + mp.tpr * sl;
        int g = (c * V) / gs, rem = c * V - g * gs;          // one division per chunk; the group index then steps with the channel
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int ch = c * V + e;
            mean[sl][e] = rstd[sl][e] = c1[sl][e] = c2[sl][e] = 0.f;
            if (c < nchunks && ch < cr) {
                mean[sl][e] = sm[g];
                rstd[sl][e] = sr[g];
                c1[sl][e] = (float)d0[g] * invn;
                c2[sl][e] = (float)d1[g] * invn;
            }
            if (++rem == gs) { rem = 0; ++g; }
        }
    }
    const int step = mp.rif * UNR;
    for (int r = r0 + ro; r < r1; r += step) {
        Raw cx[SLOTS][UNR], cd[SLOTS][UNR];
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            const int c = cb + mp.tpr * sl;
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                cx[sl][u] = px[sl][u];
                cd[sl][u] = pd[sl][u];
                const int rn = r + step + u * mp.rif;             // the next sweep's loads go out before this sweep's arithmetic
                if (c < nchunks && rn < r1) {
                    px[sl][u] = Vec<T>::load_raw(x + ((long)b * HW + rn) * ldx + c * V);
                    pd[sl][u] = Vec<T>::load_raw(dy + ((long)b * HW + rn) * lddy + c * V);
                }
            }
        }
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            const int c = cb + mp.tpr * sl;
            if (c < nchunks) {
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int rr = r + u * mp.rif;
                    if (rr < r1) {
                        float f[V], d[V], o[V], a2[V];
                        Vec<T>::unpack(cx[sl][u], f);
                        Vec<T>::unpack(cd[sl][u], d);
                        if (accumulate) Vec<T>::load(dx + ((long)b * HW + rr) * lddx + c * V, o);
                        if (add) Vec<T>::load(add + ((long)b * HW + rr) * ldadd + c * V, a2);
#pragma unroll
                        for (int e = 0; e < V; ++e) {
                            const float xh = (f[e] - mean[sl][e]) * rstd[sl][e];
                            float dz = d[e];
                            if (silu) dz *= silu_grad_f(xh * gm[sl][e] + bt[sl][e]);
                            float v = rstd[sl][e] * (gm[sl][e] * dz - c1[sl][e] - xh * c2[sl][e]);
                            if (add) v += a2[e];           // a second finished gradient of x (residual fan-in) folded in
                            o[e] = accumulate ? o[e] + v : v;
                        }
                        Vec<T>::store(dx + ((long)b * HW + rr) * lddx + c * V, o);
                    }
                }
            }
        }
    }
}

// rows per block for ~target_blocks blocks chip-wide (at most max_blk per image, at least one sweep of rows per block)
inline int gn_rows_per_blk(int B, int HW, int sweep, int target_blocks, int max_blk) {
    int nb = (target_blocks + B - 1) / B;
    nb = max(1, min(min(nb, max_blk), HW / max(1, sweep)));
    int rpb = (HW + nb - 1) / nb;
    // whole sweeps (rows in flight x unroll) per block: a ragged last sweep runs with a fraction of the threads
    rpb = (rpb + sweep - 1) / sweep * sweep;
    return rpb;
}

template <typename T>
int gn_fwd(const void* x, void* y, const float* gamma, const float* beta, float* stats, double* ws, int B, int HW,
           int C, int ldx, int ldy, int G, int gs, float eps, int silu, hipStream_t st) {
    constexpr int V = Vec<T>::N;
    if (C % V || ldx % V || ldy % V || G > 64 || G * gs > C || C / V > MAXS * NT || C > GN_MAXC / 8 * V) return -1;
    const GnMap2 mp = gn_map2(C / V);
    float* part = reinterpret_cast<float*>(ws);                     // [B][nblk][G][2] floats <= B*G*64 doubles
    const int rpb_s = gn_rows_per_blk(B, HW, mp.rif * 8, 512, GN_MAXBLK);
    dim3 grid_s((HW + rpb_s - 1) / rpb_s, B);
    const int rpb = gn_rows_per_blk(B, HW, mp.rif * 4, 1024, 1 << 20);
    dim3 grid((HW + rpb - 1) / rpb, B);
    const int slots = (C / V + NT - 1) / NT;
#define PDMK_GNF(S)                                                                                                   \
    do {                                                                                                              \
        hipLaunchKernelGGL((gn_stats_kernel<T, 8, S>), grid_s, dim3(NT), 0, st, (const T*)x, part, HW, C, ldx, G, gs, \
                           rpb_s);                                                                                    \
        hipLaunchKernelGGL((gn_apply_kernel<T, 4, S>), grid, dim3(NT), 0, st, (const T*)x, (T*)y, gamma, beta, part,  \
                           (int)grid_s.x, stats, HW, C, ldx, ldy, G, gs, eps, silu, rpb);                             \
    } while (0)
    if (slots == 1) PDMK_GNF(1);
    else if (slots == 2) PDMK_GNF(2);
    else PDMK_GNF(3);
#undef PDMK_GNF
    PDMK_CHECK_LAUNCH();
    return 0;
}

template <typename T>
int gn_bwd(const void* x, const void* dy, void* dx, const float* gamma, const float* beta, const float* stats,
           float* dgamma, float* dbeta, double* ws, float* part, long part_elems, int B, int HW, int C, int ldx,
           int lddy, int lddx, int G, int gs, int silu, int acc, const void* add, int ldadd, hipStream_t st) {
    constexpr int V = Vec<T>::N;
    if (C % V || ldx % V || lddy % V || lddx % V || G > 64 || G * gs > C || C / V > MAXS * NT || C > GN_MAXC / 8 * V) return -1;
    if (add && ((ldadd % V) || ((uintptr_t)add & 15))) return -1;
    const GnMap2 mp = gn_map2(C / V);
    const int rpb_s = gn_rows_per_blk(B, HW, mp.rif * 4, 512, GN_MAXBLK);
    dim3 grid_s((HW + rpb_s - 1) / rpb_s, B);
    const int nblk = grid_s.x * grid_s.y, cr = G * gs;
    if (!part || (long)nblk * 2 * cr > part_elems) return -1;
    float* gpart = reinterpret_cast<float*>(ws);                    // [B][nblk][G][2] floats <= B*G*64 doubles
    const int rpb = gn_rows_per_blk(B, HW, mp.rif * 2, 1024, 1 << 20);
    dim3 grid((HW + rpb - 1) / rpb, B);
    const int slots = (C / V + NT - 1) / NT;
#define PDMK_GNB(S)                                                                                                   \
    do {                                                                                                              \
        hipLaunchKernelGGL((gn_bwd_stats_kernel<T, 4, S>), grid_s, dim3(NT), 0, st, (const T*)x, (const T*)dy, gamma, \
                           beta, stats, part, gpart, HW, C, ldx, lddy, G, gs, silu, rpb_s);                           \
        launch_reduce_partials(part, nblk, cr, dgamma, dbeta, st);                                                    \
        hipLaunchKernelGGL((gn_bwd_apply_kernel<T, 2, S>), grid, dim3(NT), 0, st, (const T*)x, (const T*)dy, (T*)dx,  \
                           gamma, beta, stats, gpart, (int)grid_s.x, HW, C, ldx, lddy, lddx, G, gs, silu, acc, rpb,   \
                           (const T*)add, ldadd);                                                                     \
    } while (0)
    if (slots == 1) PDMK_GNB(1);
    else if (slots == 2) PDMK_GNB(2);
    else PDMK_GNB(3);
#undef PDMK_GNB
    PDMK_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------ LayerNorm
constexpr int LN_MAXS = 5;   // per-lane chunk slots: C <= 5*64*chunk

// forward: a wave owns `rows_per_wave` consecutive rows.  gamma / beta stay in registers for all of them (16-byte loads, once per
// wave: the first version re-read them element by element for every row - 16 scalar loads per lane and row against one 16-byte
// load and one store of data, so the kernel was bound by VMEM instruction issue, 3.7 TB/s where a copy reaches 6.4) and the next
// row's load is in flight while the current row is reduced and written (the two wave reductions make one row a latency chain).
template <typename T, int SLOTS>
__global__ __launch_bounds__(NT) void ln_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float* __restrict__ stats, int M, int C, int ldx, int ldy,
                                                    float eps, int rows_per_wave) {
    constexpr int V = Vec<T>::N;
    typedef typename Vec<T>::raw Raw;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mbase = (blockIdx.x * 4 + wave) * rows_per_wave, mend = min(M, mbase + rows_per_wave);
    if (mbase >= M) return;
    const int nchunks = C / V;
    const bool vec_gb = ((reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15) == 0;
    Raw pre[SLOTS];
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        const int c = lane + 64 * sl;
        if (c < nchunks) pre[sl] = Vec<T>::load_raw(x + (long)mbase * ldx + c * V);
    }
    float gm[SLOTS][V], bt[SLOTS][V];
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        const int c = lane + 64 * sl;
        if (c < nchunks && vec_gb) {
#pragma unroll
            for (int e = 0; e < V; e += 4) {
                const float4 g4 = *reinterpret_cast<const float4*>(gamma + c * V + e), b4 = *reinterpret_cast<const float4*>(beta + c * V + e);
                gm[sl][e] = g4.x; gm[sl][e + 1] = g4.y; gm[sl][e + 2] = g4.z; gm[sl][e + 3] = g4.w;
                bt[sl][e] = b4.x; bt[sl][e + 1] = b4.y; bt[sl][e + 2] = b4.z; bt[sl][e + 3] = b4.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                gm[sl][e] = c < nchunks ? gamma[c * V + e] : 0.f;
                bt[sl][e] = c < nchunks ? beta[c * V + e] : 0.f;
            }
        }
    }
    const float invC = 1.0f / C;
    for (int m = mbase; m < mend; ++m) {
        float f[SLOTS][V];
        float sum = 0.f;
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            const int c = lane + 64 * sl;
            if (c < nchunks) {
                Vec<T>::unpack(pre[sl], f[sl]);
                if (m + 1 < mend) pre[sl] = Vec<T>::load_raw(x + (long)(m + 1) * ldx + c * V);
#pragma unroll
                for (int e = 0; e < V; ++e) sum += f[sl][e];
            }
        }
        const float mean = wave_sum(sum) * invC;
        float sq = 0.f;
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            const int c = lane + 64 * sl;
            if (c < nchunks) {
#pragma unroll
                for (int e = 0; e < V; ++e) { const float d = f[sl][e] - mean; sq += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(sq) * invC + eps);
        if (lane == 0) { stats[2 * (long)m] = mean; stats[2 * (long)m + 1] = rstd; }
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            const int c = lane + 64 * sl;
            if (c < nchunks) {
                float o[V];
#pragma unroll
                for (int e = 0; e < V; ++e) o[e] = (f[sl][e] - mean) * rstd * gm[sl][e] + bt[sl][e];
                Vec<T>::store(y + (long)m * ldy + c * V, o);
            }
        }
    }
}

// backward: each wave walks its rows R at a time (R independent row loads in flight: the row reductions make a single
// row latency-bound); SLOTS = 16-byte chunks per lane is a template parameter so small C does not pay registers for 1280.
template <typename T, int SLOTS, int R>
__global__ __launch_bounds__(NT) void ln_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                    T* __restrict__ dx, const float* __restrict__ gamma,
                                                    const float* __restrict__ stats, float* __restrict__ part,
                                                    int M, int C, int ldx, int lddy,
                                                    int lddx, int accumulate, int rows_per_wave,
                                                    const T* __restrict__ add, int ldadd) {
    // add (optional): a second finished gradient of the same tensor (the residual branch's, blocks.py:705-867) added in this
    // store - dx = (accumulate ? dx : 0) + add + this layer's input gradient - so that the fan-in needs neither a pass of its
    // own nor an in-place update of a buffer another kernel still reads (a deferred weight gradient's dY)
    constexpr int V = Vec<T>::N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunks = C / V;
    float gm[SLOTS][V], ag[SLOTS][V], ab[SLOTS][V];
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        const int c = lane + 64 * sl;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            ag[sl][e] = ab[sl][e] = 0.f;
            gm[sl][e] = (c < nchunks) ? gamma[c * V + e] : 0.f;
        }
    }
    const int mbase = (blockIdx.x * 4 + wave) * rows_per_wave;
    const int mend = min(M, mbase + rows_per_wave);
    const float invC = 1.0f / C;
    // register ping-pong: the next R rows (x, dy, their statistics) are requested before the current R rows are reduced - the two
    // wave reductions per row otherwise leave the memory pipe idle (36.9 us for 32 768 x 320 against 6.7 for a copy)
    typedef typename Vec<T>::raw Raw;
    Raw px[R][SLOTS], pd[R][SLOTS];
    float pm[R], pr[R];
    auto fetch = [&](int m0) {
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int m = min(m0 + q, mend - 1);          // clamp: duplicates of the last row are discarded below
            pm[q] = stats[2 * (long)m];
            pr[q] = stats[2 * (long)m + 1];
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                const int c = lane + 64 * sl;
                if (c < nchunks) {
                    px[q][sl] = Vec<T>::load_raw(x + (long)m * ldx + c * V);
                    pd[q][sl] = Vec<T>::load_raw(dy + (long)m * lddy + c * V);
                }
            }
        }
    };
    if (mbase < mend) fetch(mbase);
    for (int m0 = mbase; m0 < mend; m0 += R) {
        float xh[R][SLOTS][V], d[R][SLOTS][V], mean[R], rstd[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            mean[q] = pm[q];
            rstd[q] = pr[q];
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                const int c = lane + 64 * sl;
                if (c < nchunks) {
                    Vec<T>::unpack(px[q][sl], xh[q][sl]);
                    Vec<T>::unpack(pd[q][sl], d[q][sl]);
                } else {
#pragma unroll
                    for (int e = 0; e < V; ++e) xh[q][sl][e] = d[q][sl][e] = 0.f;
                }
            }
        }
        if (m0 + R < mend) fetch(m0 + R);
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const bool live = m0 + q < mend;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
                const int c = lane + 64 * sl;
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float h = (c < nchunks) ? (xh[q][sl][e] - mean[q]) * rstd[q] : 0.f;
                    xh[q][sl][e] = h;
                    if (live) { ab[sl][e] += d[q][sl][e]; ag[sl][e] += d[q][sl][e] * h; }
                    const float g = d[q][sl][e] * gm[sl][e];
                    s1 += g;
                    s2 += g * h;
                }
            }
            s1 = wave_sum(s1) * invC;
            s2 = wave_sum(s2) * invC;
            if (live) {
                const int m = m0 + q;
#pragma unroll
                for (int sl = 0; sl < SLOTS; ++sl) {
                    const int c = lane + 64 * sl;
                    if (c < nchunks) {
                        float o[V], a2[V];
                        if (accumulate) Vec<T>::load(dx + (long)m * lddx + c * V, o);
                        if (add) Vec<T>::load(add + (long)m * ldadd + c * V, a2);
#pragma unroll
                        for (int e = 0; e < V; ++e) {
                            float v = rstd[q] * (d[q][sl][e] * gm[sl][e] - s1 - xh[q][sl][e] * s2);
                            if (add) v += a2[e];
                            o[e] = accumulate ? o[e] + v : v;
                        }
                        Vec<T>::store(dx + (long)m * lddx + c * V, o);
                    }
                }
            }
        }
    }
    // combine the 4 waves in LDS, then one slab row per block (second stage: reduce_partials_kernel)
    __shared__ float red[2][3][64 * V];
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        const int c = lane + 64 * sl;
        __syncthreads();
        if (wave > 0) {
#pragma unroll
            for (int e = 0; e < V; ++e) { red[0][wave - 1][lane * V + e] = ag[sl][e]; red[1][wave - 1][lane * V + e] = ab[sl][e]; }
        }
        __syncthreads();
        if (wave == 0 && c < nchunks) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float g2 = ag[sl][e] + red[0][0][lane * V + e] + red[0][1][lane * V + e] + red[0][2][lane * V + e];
                const float b2 = ab[sl][e] + red[1][0][lane * V + e] + red[1][1][lane * V + e] + red[1][2][lane * V + e];
                float* slab = part + (long)blockIdx.x * 2 * C;      // [dgamma | dbeta] of this block
                slab[c * V + e] = g2;
                slab[C + c * V + e] = b2;
            }
        }
    }
}

template <typename T>
int ln_fwd(const void* x, void* y, const float* gamma, const float* beta, float* stats, int M, int C, int ldx, int ldy,
           float eps, hipStream_t st) {
    constexpr int V = Vec<T>::N;
    if (C % V || ldx % V || ldy % V || C / V > LN_MAXS * 64) return -1;
    int rpw = M / 8192;                              // rows per wave: 1 ... 8, >= ~2048 blocks of 4 waves (tools/ln_bench.py: 32 768 x 320
    rpw = rpw < 1 ? 1 : (rpw > 8 ? 8 : rpw);         // rows take 13.6 / 12.2 / 11.6 / 14.6 us at 1 / 2 / 4 / 8 rows per wave)
    const dim3 grid((M + 4 * rpw - 1) / (4 * rpw));
    const int slots = (C / V + 63) / 64;
#define PDMK_LNF(S) hipLaunchKernelGGL((ln_fwd_kernel<T, S>), grid, dim3(NT), 0, st, (const T*)x, (T*)y, gamma, beta, stats, M, \
                                       C, ldx, ldy, eps, rpw)
    if (slots == 1) PDMK_LNF(1);
    else if (slots == 2) PDMK_LNF(2);
    else if (slots == 3) PDMK_LNF(3);
    else PDMK_LNF(5);
#undef PDMK_LNF
    PDMK_CHECK_LAUNCH();
    return 0;
}
// rows per wave of the LayerNorm backward (about `blocks` blocks of 4 waves; also fixes the partial-slab count)
static int ln_bwd_rpw(int M) {
    // (tools/ln_bench.py, 32 768 x 320: 45 / 32 / 25.5 / 28 us at 2048 / 1024 / 512 / 256 blocks - the block's LDS combine and slab row
    // are worth ~16 rows of streaming; 8192 x 640: 19 us at 4 rows per wave, 16.6 at 8)
    int rpw = (M + 512 * 4 - 1) / (512 * 4);
    return rpw < 8 ? 8 : (rpw > 64 ? 64 : rpw);
}
template <typename T>
int ln_bwd(const void* x, const void* dy, void* dx, const float* gamma, const float* stats, float* dgamma,
           float* dbeta, float* part, long part_elems, int M, int C, int ldx, int lddy, int lddx, int acc,
           const void* add, int ldadd, hipStream_t st) {
    constexpr int V = Vec<T>::N;
    if (C % V || ldx % V || lddy % V || lddx % V || C / V > LN_MAXS * 64 || (add && (ldadd % V))) return -1;
    const int rpw = ln_bwd_rpw(M);
    const int rows_per_blk = 4 * rpw;
    const dim3 grid((M + rows_per_blk - 1) / rows_per_blk);
    if (!part || (long)grid.x * 2 * C > part_elems) return -1;
    const int slots = (C / V + 63) / 64;
#define PDMK_LNB(S, RR) hipLaunchKernelGGL((ln_bwd_kernel<T, S, RR>), grid, dim3(NT), 0, st, (const T*)x, (const T*)dy, \
                                           (T*)dx, gamma, stats, part, M, C, ldx, lddy, lddx, acc, rpw, (const T*)add, ldadd)
    if (slots <= 1) PDMK_LNB(1, 4);
    else if (slots == 2) PDMK_LNB(2, 2);
    else if (slots == 3) PDMK_LNB(3, 2);
    else PDMK_LNB(LN_MAXS, 1);
#undef PDMK_LNB
    launch_reduce_partials(part, grid.x, C, dgamma, dbeta, st);
    PDMK_CHECK_LAUNCH();
    return 0;
}

}  // namespace

template <typename T>
int gn_apply_colstat(const void* x, void* y, const float* gamma, const float* beta, float* stats, const int64_t* colstat, int cs_ld,
                     int B, int HW, int C, int ldx, int ldy, int G, int gs, float eps, int silu, hipStream_t st) {
    constexpr int V = Vec<T>::N;
    if (C % V || ldx % V || ldy % V || G > 64 || G * gs > C || C / V > MAXS * NT) return -1;
    const GnMap2 mp = gn_map2(C / V);
    const int rpb = gn_rows_per_blk(B, HW, mp.rif * 4, 1024, 1 << 20);     // (256 ... 1024 blocks and 2 / 4 / 8 rows in flight per thread
    dim3 grid((HW + rpb - 1) / rpb, B);                                    //  measure the same, tools/gn_dbg.py)
    const int slots = (C / V + NT - 1) / NT;
#define PDMK_GNA(S)                                                                                                   \
    hipLaunchKernelGGL((gn_apply_kernel<T, 4, S>), grid, dim3(NT), 0, st, (const T*)x, (T*)y, gamma, beta, reinterpret_cast<const float*>(colstat), -1, stats, \
                       HW, C, ldx, ldy, G, gs, eps, silu, rpb, cs_ld)
    if (slots == 1) PDMK_GNA(1);
    else if (slots == 2) PDMK_GNA(2);
    else PDMK_GNA(3);
#undef PDMK_GNA
    PDMK_CHECK_LAUNCH();
    return 0;
}

extern "C" int pdmk_groupnorm_apply_colstat(const void* x, void* y, const float* gamma, const float* beta, float* stats,
                                            const int64_t* colstat, int cs_ld, int cs_col0, int B, int HW, int C, int ldx, int ldy,
                                            int G, int gs, float eps, int silu, int dtype, pdmk_stream stream) {
    if (!x || !y || !gamma || !beta || !stats || !colstat || B <= 0 || HW <= 0 || C <= 0 || G <= 0 || gs <= 0 || cs_col0 < 0 ||
        cs_ld < cs_col0 + G * gs)
        return -1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == PDMK_BF16) return gn_apply_colstat<bf16>(x, y, gamma, beta, stats, colstat + cs_col0, cs_ld, B, HW, C, ldx, ldy, G, gs, eps, silu, st);
    if (dtype == PDMK_F32) return gn_apply_colstat<float>(x, y, gamma, beta, stats, colstat + cs_col0, cs_ld, B, HW, C, ldx, ldy, G, gs, eps, silu, st);
    return -2;
}

extern "C" int pdmk_groupnorm_fwd(const void* x, void* y, const float* gamma, const float* beta, float* stats,
                                  double* ws, int B, int HW, int C, int ldx, int ldy, int G, int gs, float eps,
                                  int silu, int dtype, pdmk_stream stream) {
    if (!x || !y || !gamma || !beta || !stats || !ws || B <= 0 || HW <= 0 || G <= 0 || gs <= 0) return -1;
    PDMK_DISPATCH(dtype, gn_fwd, x, y, gamma, beta, stats, ws, B, HW, C, ldx, ldy, G, gs, eps, silu,
                  (hipStream_t)stream);
}
extern "C" int pdmk_groupnorm_bwd(const void* x, const void* dy, void* dx, const float* gamma, const float* beta,
                                  const float* stats, float* dgamma, float* dbeta, double* ws, float* part_ws,
                                  int64_t part_ws_elems, int B, int HW, int C, int ldx, int lddy, int lddx, int G,
                                  int gs, int silu, int accumulate_dx, const void* add, int ldadd, int dtype,
                                  pdmk_stream stream) {
    if (!x || !dy || !dx || !gamma || !beta || !stats || (!dgamma != !dbeta) || !ws || B <= 0 || HW <= 0) return -1;
    PDMK_DISPATCH(dtype, gn_bwd, x, dy, dx, gamma, beta, stats, dgamma, dbeta, ws, part_ws, (long)part_ws_elems, B, HW,
                  C, ldx, lddy, lddx, G, gs, silu, accumulate_dx, add, ldadd, (hipStream_t)stream);
}
/* Slab geometry of the parameter-gradient partials the two backward passes leave in part_ws ([nblk][2][n] floats). */
extern "C" int pdmk_groupnorm_bwd_partial_dims(int B, int HW, int C, int G, int gs, int dtype, int32_t* nblk, int32_t* n) {
    if (!nblk || !n || B <= 0 || HW <= 0 || C <= 0) return -1;
    const int V = dtype == PDMK_BF16 ? 8 : 4;
    if (C % V) return -1;
    const GnMap2 mp = gn_map2(C / V);
    const int rpb_s = gn_rows_per_blk(B, HW, mp.rif * 4, 512, GN_MAXBLK);
    *nblk = ((HW + rpb_s - 1) / rpb_s) * B;
    *n = G * gs;
    return 0;
}
extern "C" int pdmk_layernorm_bwd_partial_dims(int M, int C, int32_t* nblk, int32_t* n) {
    if (!nblk || !n || M <= 0 || C <= 0) return -1;
    const int rpw = ln_bwd_rpw(M);
    *nblk = (M + 4 * rpw - 1) / (4 * rpw);
    *n = C;
    return 0;
}
extern "C" int pdmk_reduce_partials_group(const pdmk_partial_item* items, int n_items, pdmk_stream stream) {
    if (!items || n_items <= 0 || n_items > PDMK_PARTIAL_GROUP_MAX) return -1;
    PartialGroup gr;
    int nmax = 0;
    for (int i = 0; i < n_items; ++i) {
        if (!items[i].part || !items[i].out0 || !items[i].out1 || items[i].nblk <= 0 || items[i].n <= 0) return -1;
        gr.it[i] = items[i];
        nmax = items[i].n > nmax ? items[i].n : nmax;
    }
    for (int i = n_items; i < PDMK_PARTIAL_GROUP_MAX; ++i) gr.it[i] = items[0];
    hipLaunchKernelGGL(reduce_partials_group_kernel, dim3((2 * nmax + NT - 1) / NT, n_items * 8), dim3(NT), 0,
                       (hipStream_t)stream, gr);
    PDMK_CHECK_LAUNCH();
    return 0;
}
extern "C" int pdmk_layernorm_fwd(const void* x, void* y, const float* gamma, const float* beta, float* stats, int M,
                                  int C, int ldx, int ldy, float eps, int dtype, pdmk_stream stream) {
    if (!x || !y || !gamma || !beta || !stats || M <= 0) return -1;
    PDMK_DISPATCH(dtype, ln_fwd, x, y, gamma, beta, stats, M, C, ldx, ldy, eps, (hipStream_t)stream);
}
extern "C" int pdmk_layernorm_bwd(const void* x, const void* dy, void* dx, const float* gamma, const float* stats,
                                  float* dgamma, float* dbeta, float* part_ws, int64_t part_ws_elems, int M, int C,
                                  int ldx, int lddy, int lddx, int accumulate_dx, const void* add, int ldadd, int dtype,
                                  pdmk_stream stream) {
    if (!x || !dy || !dx || !gamma || !stats || (!dgamma != !dbeta) || M <= 0) return -1;
    PDMK_DISPATCH(dtype, ln_bwd, x, dy, dx, gamma, stats, dgamma, dbeta, part_ws, (long)part_ws_elems, M, C, ldx, lddy,
                  lddx, accumulate_dx, add, ldadd, (hipStream_t)stream);
}
