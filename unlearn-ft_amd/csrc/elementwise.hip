// Bandwidth-bound elementwise / reduction kernels of the step: GEGLU, SiLU, strided copies (skip concat), weight
// cast/permute, bias-gradient column sums, 2x2 sum-pool (upsample bwd), timestep embedding, forward diffusion,
// layout converters, loss heads, AXPBY, fused AdamW.  All bulk traffic moves as 16-byte chunks per lane.
#include "common.h"
#include "vec.h"

namespace {

constexpr int NT = 256;
inline int grid_for(long nwork, int cap = 4096) { return (int)max(1L, min((long)cap, (nwork + NT - 1) / NT)); }

// ------------------------------------------------------------------------------------------------ GEGLU
template <typename T>
// layout 1: hidden / gate interleaved in blocks of 8 columns (hidden feature c at 16 (c / 8) + c % 8, its gate 8 further)
__global__ void geglu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long M, int F, int ldx, int ldy, int layout) {
    constexpr int V = Vec<T>::N;
    const int fc = F / V;
    const long total = M * fc;
    for (long i = blockIdx.x * (long)NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long m = i / fc;
        const int c = (int)(i - m * fc) * V;
        const int ho = layout ? ((c >> 3) << 4) + (c & 7) : c, go = layout ? ho + 8 : F + c;
        float h[V], g[V];
        Vec<T>::load(x + m * ldx + ho, h);
        Vec<T>::load(x + m * ldx + go, g);
#pragma unroll
        for (int e = 0; e < V; ++e) h[e] *= gelu_f(g[e]);
        Vec<T>::store(y + m * ldy + c, h);
    }
}
template <typename T>
__global__ void geglu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, long M, int F,
                                 int ldx, int lddy, int lddx, int layout) {
    constexpr int V = Vec<T>::N;
    const int fc = F / V;
    const long total = M * fc;
    for (long i = blockIdx.x * (long)NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long m = i / fc;
        const int c = (int)(i - m * fc) * V;
        const int ho = layout ? ((c >> 3) << 4) + (c & 7) : c, go = layout ? ho + 8 : F + c;
        float h[V], g[V], d[V], dh[V], dg[V];
        Vec<T>::load(x + m * ldx + ho, h);
        Vec<T>::load(x + m * ldx + go, g);
        Vec<T>::load(dy + m * lddy + c, d);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            dh[e] = d[e] * gelu_f(g[e]);
            dg[e] = d[e] * h[e] * gelu_grad_f(g[e]);
        }
        Vec<T>::store(dx + m * lddx + ho, dh);
        Vec<T>::store(dx + m * lddx + go, dg);
    }
}
template <typename T> int geglu_fwd(const void* x, void* y, int M, int F, int ldx, int ldy, int layout, hipStream_t st) {
    constexpr int V = Vec<T>::N;
    if (F % V || ldx % V || ldy % V || (layout && (F % 8))) return -1;
    hipLaunchKernelGGL(geglu_fwd_kernel<T>, dim3(grid_for((long)M * F / V)), dim3(NT), 0, st, (const T*)x, (T*)y,
                       (long)M, F, ldx, ldy, layout);
    PDMK_CHECK_LAUNCH();
    return 0;
}
template <typename T>
int geglu_bwd(const void* x, const void* dy, void* dx, int M, int F, int ldx, int lddy, int lddx, int layout, hipStream_t st) {
    constexpr int V = Vec<T>::N;
    if (F % V || ldx % V || lddy % V || lddx % V || (layout && (F % 8))) return -1;
    hipLaunchKernelGGL(geglu_bwd_kernel<T>, dim3(grid_for((long)M * F / V)), dim3(NT), 0, st, (const T*)x,
                       (const T*)dy, (T*)dx, (long)M, F, ldx, lddy, lddx, layout);
    PDMK_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------ SiLU / AXPBY
template <typename T, int OP>   // 0 silu fwd, 1 silu bwd (y=dy in, out=dx), 2 axpby, 3 erf-gelu fwd, 4 round to e4m3 values
__global__ void ew_kernel(const T* __restrict__ x, const T* __restrict__ a, T* __restrict__ y, long n, float alpha,
                          float beta) {
    constexpr int V = Vec<T>::N;
    const long nv = n / V;
    for (long i = blockIdx.x * (long)NT + threadIdx.x; i < nv; i += (long)gridDim.x * NT) {
        float f[V], o[V];
        Vec<T>::load(x + i * V, f);
        if (OP == 0) {
#pragma unroll
            for (int e = 0; e < V; ++e) o[e] = silu_f(f[e]);
        } else if (OP == 1) {
            Vec<T>::load(a + i * V, o);
#pragma unroll
            for (int e = 0; e < V; ++e) o[e] *= silu_grad_f(f[e]);
        } else if (OP == 3) {
#pragma unroll
            for (int e = 0; e < V; ++e) o[e] = gelu_f(f[e]);
        } else if (OP == 4) {
#pragma unroll
            for (int e = 0; e < V; ++e) o[e] = quant_e4m3_f(f[e]);
        } else {
            Vec<T>::load(y + i * V, o);
#pragma unroll
            for (int e = 0; e < V; ++e) o[e] = alpha * f[e] + beta * o[e];
        }
        Vec<T>::store(y + i * V, o);
    }
    // scalar tail
    for (long i = nv * V + blockIdx.x * (long)NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        const float f = to_f32(x[i]);
        float o;
        if (OP == 0) o = silu_f(f);
        else if (OP == 1) o = to_f32(a[i]) * silu_grad_f(f);
        else if (OP == 3) o = gelu_f(f);
        else if (OP == 4) o = quant_e4m3_f(f);
        else o = alpha * f + beta * to_f32(y[i]);
        y[i] = from_f32<T>(o);
    }
}
template <typename T, int OP>
int ew(const void* x, const void* a, void* y, long n, float alpha, float beta, hipStream_t st) {
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)a) & 15) return -1;
    hipLaunchKernelGGL((ew_kernel<T, OP>), dim3(grid_for(n / Vec<T>::N + 1)), dim3(NT), 0, st, (const T*)x,
                       (const T*)a, (T*)y, n, alpha, beta);
    PDMK_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------ copy2d
template <typename T>
__global__ void copy2d_kernel(const T* __restrict__ src, T* __restrict__ dst, long rows, int cols, int lds, int ldd,
                              int acc) {
    constexpr int V = Vec<T>::N;
    const int cc = cols / V;
    const long total = rows * cc;
    for (long i = blockIdx.x * (long)NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long r = i / cc;
        const int c = (int)(i - r * cc) * V;
        float f[V];
        Vec<T>::load(src + r * lds + c, f);
        if (acc) {
            float o[V];
            Vec<T>::load(dst + r * ldd + c, o);
#pragma unroll
            for (int e = 0; e < V; ++e) f[e] += o[e];
        }
        Vec<T>::store(dst + r * ldd + c, f);
    }
}
template <typename T>
int copy2d(const void* src, void* dst, long rows, int cols, int lds, int ldd, int acc, hipStream_t st) {
    constexpr int V = Vec<T>::N;
    if (cols % V || lds % V || ldd % V || (((uintptr_t)src | (uintptr_t)dst) & 15)) return -1;
    hipLaunchKernelGGL(copy2d_kernel<T>, dim3(grid_for(rows * cols / V)), dim3(NT), 0, st, (const T*)src, (T*)dst,
                       rows, cols, lds, ldd, acc);
    PDMK_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------ cast / permute
// mode 0: dst[i] = src[i];  mode 1: [n0,n2] -> [n2,n0];  mode 2: conv [n0=Co,9,n2=Ci] -> [Ci, 9 (flipped), Co]
template <typename T>
__global__ void cast_permute_kernel(const float* __restrict__ src, T* __restrict__ dst, int n0, int n1, int n2,
                                    int mode) {
    const long total = (long)n0 * n1 * n2;
    for (long o = blockIdx.x * (long)NT + threadIdx.x; o < total; o += (long)gridDim.x * NT) {
        long s;
        if (mode == 0) {
            s = o;
        } else if (mode == 1) {   // o = i2*n0 + i0
            const long i2 = o / n0, i0 = o - i2 * n0;
            s = i0 * n2 + i2;
        } else {                  // o = (i2*9 + t')*n0 + i0,  t = 8 - t'
            const long i0 = o % n0, rest = o / n0;
            const long tp = rest % n1, i2 = rest / n1;
            s = (i0 * n1 + (n1 - 1 - tp)) * n2 + i2;
        }
        dst[o] = from_f32<T>(src[s]);
    }
}
template <typename T> int cast_permute(const float* src, void* dst, int n0, int n1, int n2, int mode, hipStream_t st) {
    hipLaunchKernelGGL(cast_permute_kernel<T>, dim3(grid_for((long)n0 * n1 * n2, 8192)), dim3(NT), 0, st, src, (T*)dst,
                       n0, n1, n2, mode);
    PDMK_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------ batched transpose
// One launch refreshes every dgrad weight copy: a table of 64x64 tiles, each a strided 2-D transpose
//   dst[dst_off + c*dst_ld + r] = src[src_off + r*src_ld + c],  r in [r0, r0+64) < rows, c in [c0, c0+64) < cols
// (Linear W^T; conv [Co][9][Ci] -> [Ci][9 flipped][Co] is nine such transposes).  Both sides coalesced through LDS.
struct TrTile { int src_off_lo, src_off_hi, dst_off_lo, dst_off_hi, rows, cols, src_ld, dst_ld, r0, c0, pad0, pad1; };
template <typename T>
__global__ __launch_bounds__(NT) void transpose_tiles_kernel(const T* __restrict__ src, T* __restrict__ dst,
                                                             const TrTile* __restrict__ tab) {
    constexpr int V = Vec<T>::N;                      // elements per 16 bytes
    __shared__ T tile[64][66];                        // 33 dwords (bf16) per row: the column reads below are at most 2-way conflicts
    const TrTile t = tab[blockIdx.x];
    const long so = ((long)t.src_off_hi << 32) | (unsigned)t.src_off_lo;
    const long dof = ((long)t.dst_off_hi << 32) | (unsigned)t.dst_off_lo;
    // 16-byte path (every weight of the step: rows of 8 k elements on 256-byte boundaries): each thread moves two 16-byte chunks in
    // and two out; the element-wise form this replaces issued 32 two-byte accesses per thread and ran at 1.7 TB/s
    const bool vec = ((so | dof | t.src_ld | t.dst_ld) & (V - 1)) == 0 &&
                     (((uintptr_t)(src + so) | (uintptr_t)(dst + dof)) & 15) == 0;
    if (vec) {
        constexpr int CPR = 64 / V;                   // chunks per tile row
#pragma unroll
        for (int j = 0; j < 64 * CPR / NT; ++j) {
            const int idx = threadIdx.x + NT * j, row = idx / CPR, ch = idx - row * CPR;
            const int r = t.r0 + row, c = t.c0 + ch * V;
            if (r < t.rows && c + V <= t.cols) {
                const typename Vec<T>::raw v = Vec<T>::load_raw(src + so + (long)r * t.src_ld + c);
#pragma unroll
                for (int e = 0; e < V; ++e) tile[row][ch * V + e] = v[e];
            } else if (r < t.rows) {
                for (int e = 0; e < V && c + e < t.cols; ++e) tile[row][ch * V + e] = src[so + (long)r * t.src_ld + c + e];
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 64 * CPR / NT; ++j) {
            const int idx = threadIdx.x + NT * j, cl = idx / CPR, rch = idx - cl * CPR;
            const int c = t.c0 + cl, r = t.r0 + rch * V;
            if (c < t.cols && r + V <= t.rows) {
                typename Vec<T>::raw v;
#pragma unroll
                for (int e = 0; e < V; ++e) v[e] = tile[rch * V + e][cl];
                *reinterpret_cast<typename Vec<T>::raw*>(dst + dof + (long)c * t.dst_ld + r) = v;
            } else if (c < t.cols) {
                for (int e = 0; e < V && r + e < t.rows; ++e) dst[dof + (long)c * t.dst_ld + r + e] = tile[rch * V + e][cl];
            }
        }
        return;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int r = t.r0 + ty + 4 * j, c = t.c0 + tx;
        if (r < t.rows && c < t.cols) tile[ty + 4 * j][tx] = src[so + (long)r * t.src_ld + c];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int c = t.c0 + ty + 4 * j, r = t.r0 + tx;
        if (r < t.rows && c < t.cols) dst[dof + (long)c * t.dst_ld + r] = tile[tx][ty + 4 * j];
    }
}

// ------------------------------------------------------------------------------------------------ colsum
// out[batch][n] (+)= sum over the rows of that batch of x[row][n].  Rows are read fully coalesced: a 256-thread block
// is (rows-in-flight x column chunks); partial sums are combined across the rows-in-flight in LDS, then ONE float
// atomic per column per block (blocks per batch are capped so same-address atomic contention stays low).
template <typename T>
__global__ __launch_bounds__(NT) void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, long rows, int N,
                                                    int ld, long rows_per_blk, int tpr, int ldo) {
    constexpr int V = Vec<T>::N;
    __shared__ float red[NT * V];
    const int nc = N / V;
    const int tid = threadIdx.x, ct = tid % tpr, ro = tid / tpr, rif = NT / tpr;
    const int c = blockIdx.x * tpr + ct;
    const long base = (long)blockIdx.z * rows;
    const long r0 = blockIdx.y * rows_per_blk, r1 = min(rows, r0 + rows_per_blk);
    float s[V];
#pragma unroll
    for (int e = 0; e < V; ++e) s[e] = 0.f;
    if (c < nc && ro < rif) {
        long r = r0 + ro;
        for (; r + 3L * rif < r1; r += 4L * rif) {     // 4 independent 16-byte loads in flight per lane
            float f0[V], f1[V], f2[V], f3[V];
            Vec<T>::load(x + (base + r) * ld + c * V, f0);
            Vec<T>::load(x + (base + r + rif) * ld + c * V, f1);
            Vec<T>::load(x + (base + r + 2L * rif) * ld + c * V, f2);
            Vec<T>::load(x + (base + r + 3L * rif) * ld + c * V, f3);
#pragma unroll
            for (int e = 0; e < V; ++e) s[e] += (f0[e] + f1[e]) + (f2[e] + f3[e]);
        }
        for (; r < r1; r += rif) {
            float f[V];
            Vec<T>::load(x + (base + r) * ld + c * V, f);
#pragma unroll
            for (int e = 0; e < V; ++e) s[e] += f[e];
        }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) red[tid * V + e] = s[e];
    __syncthreads();
    if (ro == 0 && c < nc) {
        for (int j = 1; j < rif; ++j)
#pragma unroll
            for (int e = 0; e < V; ++e) s[e] += red[(j * tpr + ct) * V + e];
#pragma unroll
        for (int e = 0; e < V; ++e) unsafeAtomicAdd(&out[(long)blockIdx.z * ldo + c * V + e], s[e]);
    }
}
__global__ void zero_f32_kernel(float* __restrict__ p, long n, int N, int ldo) {      // [n / N rows][N cols], row stride ldo
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[(i / N) * ldo + (i % N)] = 0.f;
}
template <typename T>
int colsum(const void* x, float* out, long rows, int N, int ld, int acc, int nbatch, int ldo, hipStream_t st) {
    if (ldo <= 0) ldo = N;
    constexpr int V = Vec<T>::N;
    if (N % V || ld % V) return -1;
    // a kernel, not hipMemsetAsync: memset nodes captured into the 2nd..nth hipGraph of a shared memory pool were seen to
    // leave the buffer unzeroed on replay (segmented backward graphs, ROCm 7.2)
    if (!acc) {
        const long n = (long)N * nbatch;
        hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, n, N, ldo);
    }
    const int nc = N / V;
    const int tpr = nc < NT ? nc : NT;              // threads per row = its 16-byte chunks (exact: a power of two left 24 of 64 idle at N = 320)
    const int rif = NT / tpr;
    const int gx = (nc + tpr - 1) / tpr;
    // blocks per batch: every block ends with one float atomic per column, and the blocks of a batch hit the SAME addresses - at 128
    // blocks per image the kernel took 30 us for a 21 MB tensor, at 8 it takes 8.9 (tools/colsum_bench.py; 16: 8.2, 4: 13)
    long gy = max(1L, min(rows / max(1, rif * 8), (long)max(1, 64 / (gx * nbatch))));
    const long rpb = (rows + gy - 1) / gy;
    gy = (rows + rpb - 1) / rpb;
    hipLaunchKernelGGL(colsum_kernel<T>, dim3(gx, (int)gy, nbatch), dim3(NT), 0, st, (const T*)x, out, rows, N, ld, rpb,
                       tpr, ldo);
    PDMK_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------ 2x2 sum pool
template <typename T>
__global__ void pool_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int H, int W, int C) {
    constexpr int V = Vec<T>::N;
    const int cc = C / V;
    const long total = (long)B * H * W * cc;
    for (long i = blockIdx.x * (long)NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int c = (int)(i % cc) * V;
        long p = i / cc;
        const int xw = (int)(p % W); p /= W;
        const int yh = (int)(p % H);
        const long b = p / H;
        const T* s = src + ((b * 2 * H + 2 * yh) * 2 * W + 2 * xw) * (long)C + c;
        float a[V], t[V];
        Vec<T>::load(s, a);
        Vec<T>::load(s + C, t);
#pragma unroll
        for (int e = 0; e < V; ++e) a[e] += t[e];
        Vec<T>::load(s + 2L * W * C, t);
#pragma unroll
        for (int e = 0; e < V; ++e) a[e] += t[e];
        Vec<T>::load(s + 2L * W * C + C, t);
#pragma unroll
        for (int e = 0; e < V; ++e) a[e] += t[e];
        Vec<T>::store(dst + ((b * H + yh) * W + xw) * (long)C + c, a);
    }
}
template <typename T> int pool(const void* src, void* dst, int B, int H, int W, int C, hipStream_t st) {
    if (C % Vec<T>::N) return -1;
    hipLaunchKernelGGL(pool_kernel<T>, dim3(grid_for((long)B * H * W * C / Vec<T>::N)), dim3(NT), 0, st, (const T*)src,
                       (T*)dst, B, H, W, C);
    PDMK_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------------ timestep embedding
template <typename T>
__global__ void temb_kernel(const int64_t* __restrict__ t, const float* __restrict__ freqs, T* __restrict__ out, int B,
                            int dim) {
    const int half = dim / 2;
    const int total = B * half;
    for (int i = blockIdx.x * NT + threadIdx.x; i < total; i += gridDim.x * NT) {
        const int b = i / half, j = i - b * half;
        const float arg = (float)t[b] * freqs[j];
        out[(long)b * dim + j] = from_f32<T>(cosf(arg));
        out[(long)b * dim + half + j] = from_f32<T>(sinf(arg));
    }
}

// ------------------------------------------------------------------------------------------------ forward diffusion
template <typename T>
__global__ void noise_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                             const int64_t* __restrict__ t, const float* __restrict__ sa, const float* __restrict__ sb,
                             T* __restrict__ noisy, float* __restrict__ target, int B, int C, int HW, int cpad) {
    const long total = (long)B * HW * cpad;
    for (long i = blockIdx.x * (long)NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int c = (int)(i % cpad);
        const long p = i / cpad;
        const int px = (int)(p % HW);
        const int b = (int)(p / HW);
        float xn = 0.f, v = 0.f;
        if (c < C) {
            const long s = ((long)b * C + c) * HW + px;
            const float a = sa[t[b]], sg = sb[t[b]];
            xn = a * x0[s] + sg * noise[s];
            v = a * noise[s] - sg * x0[s];
        }
        noisy[i] = from_f32<T>(xn);
        if (target) target[i] = v;
    }
}
template <typename T>
__global__ void nchw2nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int B, int C, int HW, int cpad) {
    const long total = (long)B * HW * cpad;
    for (long i = blockIdx.x * (long)NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int c = (int)(i % cpad);
        const long p = i / cpad;
        const int px = (int)(p % HW);
        const int b = (int)(p / HW);
        dst[i] = from_f32<T>(c < C ? src[((long)b * C + c) * HW + px] : 0.f);
    }
}
template <typename T>
__global__ void nhwc2nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int B, int C, int HW, int ld) {
    const long total = (long)B * C * HW;
    for (long i = blockIdx.x * (long)NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int px = (int)(i % HW);
        const long p = i / HW;
        const int c = (int)(p % C);
        const int b = (int)(p / C);
        dst[i] = to_f32(src[((long)b * HW + px) * ld + c]);
    }
}

// ------------------------------------------------------------------------------------------------ loss heads
__device__ __forceinline__ float ld_any(const void* p, int dtype, long i) {
    return dtype == PDMK_BF16 ? (float)reinterpret_cast<const bf16*>(p)[i] : reinterpret_cast<const float*>(p)[i];
}
__global__ void mse_fwd_kernel(const void* __restrict__ a, int adt, const void* __restrict__ b, int bdt,
                               const float* __restrict__ w, double* __restrict__ out, int slot, long rows_per_b,
                               int cols, int lda, int ldb, double scale) {
    const int bi = blockIdx.y;
    const long total = rows_per_b * cols;
    float s = 0.f;
    for (long i = blockIdx.x * (long)NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long r = i / cols;
        const int c = (int)(i - r * cols);
        const long row = (long)bi * rows_per_b + r;
        const float d = ld_any(a, adt, row * lda + c) - ld_any(b, bdt, row * ldb + c);
        s += d * d;
    }
    s = wave_sum(s);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = (double)red[0] + red[1] + red[2] + red[3];
        atomicAdd(&out[slot], tot * scale * (w ? (double)w[bi] : 1.0));
    }
}
__global__ void mse_bwd_kernel(const void* __restrict__ a, int adt, const void* __restrict__ b, int bdt,
                               const float* __restrict__ w, void* __restrict__ da, long rows_per_b, int cols, int lda,
                               int ldb, int ldda, float gscale, int acc) {
    const int bi = blockIdx.y;
    const long total = rows_per_b * cols;
    const float gs = gscale * (w ? w[bi] : 1.f);
    for (long i = blockIdx.x * (long)NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long r = i / cols;
        const int c = (int)(i - r * cols);
        const long row = (long)bi * rows_per_b + r;
        float g = gs * (ld_any(a, adt, row * lda + c) - ld_any(b, bdt, row * ldb + c));
        const long o = row * ldda + c;
        if (adt == PDMK_BF16) {
            bf16* p = reinterpret_cast<bf16*>(da);
            p[o] = (bf16)(g + (acc ? (float)p[o] : 0.f));
        } else {
            float* p = reinterpret_cast<float*>(da);
            p[o] = g + (acc ? p[o] : 0.f);
        }
    }
}

// Loss value AND its gradient seed in one pass, 8 elements (one or two 16-byte loads) per thread and iteration: the block
// feature head (trainer.py:2475-2481) reads 2 x 6.3 M activations per image; the scalar kernels above stay for the
// 4-channel prediction heads (cols not a multiple of 8).
template <typename T> __device__ __forceinline__ void ld8(const T* p, float* f);
template <> __device__ __forceinline__ void ld8<bf16>(const bf16* p, float* f) { Vec<bf16>::load(p, f); }
template <> __device__ __forceinline__ void ld8<float>(const float* p, float* f) { Vec<float>::load(p, f); Vec<float>::load(p + 4, f + 4); }
template <typename T> __device__ __forceinline__ void st8(T* p, const float* f);
template <> __device__ __forceinline__ void st8<bf16>(bf16* p, const float* f) { Vec<bf16>::store(p, f); }
template <> __device__ __forceinline__ void st8<float>(float* p, const float* f) { Vec<float>::store(p, f); Vec<float>::store(p + 4, f + 4); }

template <typename TA, typename TB>
__global__ __launch_bounds__(NT) void mse_vec_kernel(const TA* __restrict__ a, const TB* __restrict__ b,
                                                     const float* __restrict__ w, double* __restrict__ out, int slot,
                                                     TA* __restrict__ da, long rows_per_b, int cols, int lda, int ldb,
                                                     int ldda, double scale, float gscale, int acc) {
    const int bi = blockIdx.y, cc = cols >> 3;
    const long total = rows_per_b * cc;
    const float gs = gscale * (w ? w[bi] : 1.f);
    float s = 0.f;
    // 4 independent (row, chunk) items per thread and iteration: 8 sixteen-byte loads in flight before the first use (this is a
    // pure HBM stream).  The grid is small on purpose: every block ends in ONE double atomic on out[slot], and a few thousand
    // same-address atomics per launch (the round-2 grid) cost more than the stream itself
    const long step = (long)gridDim.x * NT;
    for (long i0 = blockIdx.x * (long)NT + threadIdx.x; i0 < total; i0 += 4 * step) {
        float x[4][8], y[4][8];
        long off[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = i0 + u * step;
            if (i < total) {
                const long r = i / cc;
                const int c = (int)(i - r * cc) << 3;
                const long row = (long)bi * rows_per_b + r;
                off[u] = row * ldda + c;
                ld8<TA>(a + row * lda + c, x[u]);
                ld8<TB>(b + row * ldb + c, y[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u * step >= total) continue;
#pragma unroll
            for (int e = 0; e < 8; ++e) { x[u][e] -= y[u][e]; s += x[u][e] * x[u][e]; }
            if (da) {
                if (acc) {
                    ld8<TA>(da + off[u], y[u]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[u][e] = gs * x[u][e] + y[u][e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[u][e] *= gs;
                }
                st8<TA>(da + off[u], x[u]);
            }
        }
    }
    s = wave_sum(s);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0 && out) {
        const double tot = (double)red[0] + red[1] + red[2] + red[3];
        atomicAdd(&out[slot], tot * scale * (w ? (double)w[bi] : 1.0));
    }
}

// ------------------------------------------------------------------------------------------------ split-K finish
// C[m][n] = (acc ? C : 0) + ws[m][n] + bias[n] + rowvec[m / rows_per_b][n] + R[m][n]   (ws: fp32 split-K partial sums)
template <typename T>
__global__ void splitk_finish_kernel(const float* __restrict__ ws, T* __restrict__ C, const float* __restrict__ bias,
                                     const float* __restrict__ rowvec, const T* __restrict__ R, long M, int N, int ldc,
                                     int ldr, int rows_per_b, int ldrv, int nslab, int acc) {
    const int nc = N / 4;
    const long total = M * nc;
    for (long i = blockIdx.x * (long)NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long m = i / nc;
        const int n = (int)(i - m * nc) * 4;
        float4 w = *reinterpret_cast<const float4*>(ws + m * N + n);
        int sl = 1;
        for (; sl + 3 <= nslab; sl += 3) {                         // split-K slabs [nslab][M][N], added in a fixed order; three
            const float4 u0 = *reinterpret_cast<const float4*>(ws + ((long)sl * M + m) * N + n);          // loads in flight
            const float4 u1 = *reinterpret_cast<const float4*>(ws + ((long)(sl + 1) * M + m) * N + n);
            const float4 u2 = *reinterpret_cast<const float4*>(ws + ((long)(sl + 2) * M + m) * N + n);
            w.x = ((w.x + u0.x) + u1.x) + u2.x; w.y = ((w.y + u0.y) + u1.y) + u2.y;
            w.z = ((w.z + u0.z) + u1.z) + u2.z; w.w = ((w.w + u0.w) + u1.w) + u2.w;
        }
        for (; sl < nslab; ++sl) {
            const float4 u = *reinterpret_cast<const float4*>(ws + ((long)sl * M + m) * N + n);
            w.x += u.x; w.y += u.y; w.z += u.z; w.w += u.w;
        }
        float v[4] = {w.x, w.y, w.z, w.w};
        if (bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bias[n + e];
        }
        if (rowvec) {
            const float* rv = rowvec + (m / rows_per_b) * ldrv + n;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += rv[e];
        }
        if (R) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += to_f32(R[m * ldr + n + e]);
        }
        T* c = C + m * ldc + n;
#pragma unroll
        for (int e = 0; e < 4; ++e) c[e] = from_f32<T>(v[e] + (acc ? to_f32(c[e]) : 0.f));
    }
}

// The same finish with the GroupNorm statistics of the stored output (pdmk_gemm_args.colstat, for split-K producers): a block
// owns 64 rows of one image x 64 columns; thread = (row % 16, 4-column chunk), four sweeps; the per-thread column sums of the
// values AS STORED are added over the 16 row-threads through LDS and leave as fixed-point integer atomics into
// colstat[image][0 / 1][column] (order-independent).
template <typename T>
__global__ __launch_bounds__(NT) void splitk_finish_cs_kernel(const float* __restrict__ ws, T* __restrict__ C, const float* __restrict__ bias,
                                                              const float* __restrict__ rowvec, const T* __restrict__ R, long M, int N,
                                                              int ldc, int ldr, int rows_per_b, int ldrv, int nslab, int acc,
                                                              int64_t* __restrict__ colstat, int cs_ld) {
    static_assert(NT == 256, "16 row-threads x 16 column chunks");
    __shared__ float part[16][2][64];
    const int tid = threadIdx.x, cc = tid & 15, rr = tid >> 4;
    const long m0 = (long)blockIdx.x * 64;
    const int n = blockIdx.y * 64 + cc * 4;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
        // (M % 64 == 0: all four sweeps are in range - no exits between them, so that their loads are in flight together)
        float4 w[4];
#pragma unroll
        for (int sw = 0; sw < 4; ++sw) w[sw] = *reinterpret_cast<const float4*>(ws + (m0 + sw * 16 + rr) * N + n);
        for (int sl = 1; sl < nslab; ++sl) {
#pragma unroll
            for (int sw = 0; sw < 4; ++sw) {
                const float4 u = *reinterpret_cast<const float4*>(ws + ((long)sl * M + m0 + sw * 16 + rr) * N + n);
                w[sw].x += u.x; w[sw].y += u.y; w[sw].z += u.z; w[sw].w += u.w;
            }
        }
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) b4 = make_float4(bias[n], bias[n + 1], bias[n + 2], bias[n + 3]);   // (no alignment assumed)
#pragma unroll
        for (int sw = 0; sw < 4; ++sw) {
            const long m = m0 + sw * 16 + rr;
            float v[4] = {w[sw].x + b4.x, w[sw].y + b4.y, w[sw].z + b4.z, w[sw].w + b4.w};
            if (rowvec) {
                const float* rv = rowvec + (m / rows_per_b) * ldrv + n;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += rv[e];
            }
            if (R) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += to_f32(R[m * ldr + n + e]);
            }
            T* c = C + m * ldc + n;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const T o = from_f32<T>(v[e] + (acc ? to_f32(c[e]) : 0.f));
                c[e] = o;
                const float x = to_f32(o);
                s1[e] += x;
                s2[e] += x * x;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        part[rr][0][cc * 4 + e] = s1[e];
        part[rr][1][cc * 4 + e] = s2[e];
    }
    __syncthreads();
    if (tid < 128) {
        const int which = tid >> 6, col = tid & 63, nn = blockIdx.y * 64 + col;
        if (nn < N && m0 < M) {
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += part[r][which][col];
            cs_add(colstat + ((m0 / rows_per_b) * 4 + which * 2) * cs_ld + nn, cs_ld, sum, which == 1);
        }
    }
}

// ------------------------------------------------------------------------------------------------ upsampling conv as 4 phases
// nearest-x2 upsample + 3x3 conv == four 2x2 convs on the low-resolution image (conv_mode 5..12, gemm_ring.hip): the 3x3 taps
// that fall on the same source pixel are summed.  For output phase a (row parity) source row offset dy in {0, 1} collects
// ky in S(a, dy): S(0,0) = {0}, S(0,1) = {1, 2}, S(1,0) = {0, 1}, S(1,1) = {2}; likewise for columns.
__device__ __forceinline__ int up2_lo(int a, int d) { return a == 0 ? (d == 0 ? 0 : 1) : (d == 0 ? 0 : 2); }
__device__ __forceinline__ int up2_hi(int a, int d) { return a == 0 ? (d == 0 ? 0 : 2) : (d == 0 ? 1 : 2); }
__device__ __forceinline__ int up2_d(int a, int k) { return a == 0 ? (k == 0 ? 0 : 1) : (k <= 1 ? 0 : 1); }   // dy of tap ky under phase a

// w3 [Co][9][Ci] fp32 master -> wp [4][Co][4][Ci] (forward) and wpt [Ci][4][4][Co] (input gradient, phase-major inside a
// row: wpt[ci][p][(e, f)][co] = wp[p][co][(1 - e, 1 - f)][ci]), both in T; the sums are formed in fp32 and rounded once
template <typename T>
__global__ void up2_pack_kernel(const float* __restrict__ w3, T* __restrict__ wp, T* __restrict__ wpt, int Co, int Ci) {
    // one thread = 8 consecutive input channels of one (phase, co, tap): 16-byte fp32 loads, one 16-byte store (bf16)
    const int c8n = Ci / 8;
    const long total = 4L * Co * 4 * c8n;
    for (long i = blockIdx.x * (long)NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int c8 = (int)(i % c8n);
        long r = i / c8n;
        const int j = (int)(r & 3);
        r >>= 2;
        const int co = (int)(r % Co), p = (int)(r / Co);
        const int a = p >> 1, b = p & 1, dy = j >> 1, dx = j & 1;
        float s[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = 0.f;
        for (int ky = up2_lo(a, dy); ky <= up2_hi(a, dy); ++ky)
            for (int kx = up2_lo(b, dx); kx <= up2_hi(b, dx); ++kx) {
                const float4* src = reinterpret_cast<const float4*>(w3 + ((long)co * 9 + ky * 3 + kx) * Ci + c8 * 8);
                const float4 v0 = src[0], v1 = src[1];
                s[0] += v0.x; s[1] += v0.y; s[2] += v0.z; s[3] += v0.w; s[4] += v1.x; s[5] += v1.y; s[6] += v1.z; s[7] += v1.w;
            }
        T* dst = wp + (((long)p * Co + co) * 4 + j) * Ci + c8 * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[e] = from_f32<T>(s[e]);
        if (wpt) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                wpt[(((long)(c8 * 8 + e) * 4 + p) * 4 + ((1 - dy) * 2 + (1 - dx))) * Co + co] = from_f32<T>(s[e]);
        }
    }
}
// dw3 [Co][9][Ci] += the phase gradients dwp [4][Co][4][Ci]: tap (ky, kx) belongs to exactly one (dy, dx) of every phase
__global__ void up2_combine_kernel(const float* __restrict__ dwp, float* __restrict__ dw3, int Co, int Ci) {
    const long total = (long)Co * 9 * Ci;
    for (long i = blockIdx.x * (long)NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int ci = (int)(i % Ci);
        const long r = i / Ci;
        const int t = (int)(r % 9), co = (int)(r / 9), ky = t / 3, kx = t - 3 * ky;
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            s += dwp[(((long)p * Co + co) * 4 + (up2_d(p >> 1, ky) * 2 + up2_d(p & 1, kx))) * Ci + ci];
        dw3[i] += s;
    }
}

// ------------------------------------------------------------------------------------------------ AdamW / sumsq
template <int U>
__global__ void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             long n, const float* __restrict__ lrp, float b1, float b2, float eps, float wd,
                             const float* __restrict__ bc, float gscale, int zero_grad, bf16* __restrict__ wout) {
    const float lr = lrp[0], bc1 = bc[0], bc2s = sqrtf(bc[1]);
    const long nv = n / 4;
    const long stride = (long)gridDim.x * NT;
    // U float4 per stream in flight per thread (U = 2: the four read streams of the next element set are requested before the
    // five write streams of this one go out)
    for (long i0 = blockIdx.x * (long)NT + threadIdx.x; i0 < nv; i0 += stride * U) {
        float4 P[U], G[U], Mv[U], Vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = i0 + u * stride;
            if (i < nv) {
                P[u] = reinterpret_cast<float4*>(p)[i];
                G[u] = reinterpret_cast<float4*>(g)[i];
                Mv[u] = reinterpret_cast<float4*>(m)[i];
                Vv[u] = reinterpret_cast<float4*>(v)[i];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = i0 + u * stride;
            if (i >= nv) break;
            float* pp = &P[u].x; float* gg = &G[u].x; float* mm = &Mv[u].x; float* vv = &Vv[u].x;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gr = gg[e] * gscale;
                pp[e] *= (1.f - lr * wd);
                mm[e] = b1 * mm[e] + (1.f - b1) * gr;
                vv[e] = b2 * vv[e] + (1.f - b2) * gr * gr;
                const float denom = sqrtf(vv[e]) / bc2s + eps;
                pp[e] -= (lr / bc1) * (mm[e] / denom);
            }
            reinterpret_cast<float4*>(p)[i] = P[u];
            if (wout) {      // refreshed bf16 compute copy in the same pass (saves re-reading the master arena)
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16)pp[e];
                reinterpret_cast<bf16x4*>(wout)[i] = o;
            }
            reinterpret_cast<float4*>(m)[i] = Mv[u];
            reinterpret_cast<float4*>(v)[i] = Vv[u];
            if (zero_grad) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}
// Deferred sum of weight-gradient split-K slabs for up to PDMK_SLAB_GROUP_MAX weights in one launch:
// dst[e] += sum_s ws[s * n + e], slabs added in a fixed order (bit-reproducible, no atomics); blockIdx.y = item.
struct SlabGroup {
    pdmk_slab_item it[PDMK_SLAB_GROUP_MAX];
};
__global__ void splitk_finish_group_kernel(SlabGroup gr) {
    const pdmk_slab_item it = gr.it[blockIdx.y];
    const long n4 = it.n / 4;
    const float4* ws = reinterpret_cast<const float4*>(it.ws);
    float4* dst = reinterpret_cast<float4*>(it.dst);
    for (long i = blockIdx.x * (long)NT + threadIdx.x; i < n4; i += (long)gridDim.x * NT) {
        float4 a = dst[i];
        int s = 0;
        for (; s + 4 <= it.nslab; s += 4) {           // four slabs' loads in flight; the sum order stays slab 0, 1, 2, ... (fixed)
            const float4 v0 = ws[(long)s * n4 + i], v1 = ws[(long)(s + 1) * n4 + i], v2 = ws[(long)(s + 2) * n4 + i],
                         v3 = ws[(long)(s + 3) * n4 + i];
            a.x = (((a.x + v0.x) + v1.x) + v2.x) + v3.x;
            a.y = (((a.y + v0.y) + v1.y) + v2.y) + v3.y;
            a.z = (((a.z + v0.z) + v1.z) + v2.z) + v3.z;
            a.w = (((a.w + v0.w) + v1.w) + v2.w) + v3.w;
        }
        for (; s < it.nslab; ++s) {
            const float4 v = ws[(long)s * n4 + i];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        dst[i] = a;
    }
}
__global__ void sumsq_kernel(const float* __restrict__ x, long n, double* __restrict__ out, int slot) {
    float s = 0.f;
    for (long i = blockIdx.x * (long)NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) s += x[i] * x[i];
    s = wave_sum(s);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&out[slot], (double)red[0] + red[1] + red[2] + red[3]);
}

}  // namespace

extern "C" int pdmk_geglu_fwd(const void* x, void* y, int M, int F, int ldx, int ldy, int layout, int dtype, pdmk_stream s) {
    if (!x || !y || M <= 0 || F <= 0) return -1;
    PDMK_DISPATCH(dtype, geglu_fwd, x, y, M, F, ldx, ldy, layout, (hipStream_t)s);
}
extern "C" int pdmk_geglu_bwd(const void* x, const void* dy, void* dx, int M, int F, int ldx, int lddy, int lddx,
                              int layout, int dtype, pdmk_stream s) {
    if (!x || !dy || !dx || M <= 0 || F <= 0) return -1;
    PDMK_DISPATCH(dtype, geglu_bwd, x, dy, dx, M, F, ldx, lddy, lddx, layout, (hipStream_t)s);
}
extern "C" int pdmk_silu_fwd(const void* x, void* y, int64_t n, int dtype, pdmk_stream s) {
    if (!x || !y || n <= 0) return -1;
    if (dtype == PDMK_BF16) return ew<bf16, 0>(x, x, y, n, 0, 0, (hipStream_t)s);
    if (dtype == PDMK_F32) return ew<float, 0>(x, x, y, n, 0, 0, (hipStream_t)s);
    return -2;
}
extern "C" int pdmk_quantize_e4m3(const void* x, void* y, int64_t n, int dtype, pdmk_stream s) {
    if (!x || !y || n <= 0) return -1;
    if (dtype == PDMK_BF16) return ew<bf16, 4>(x, x, y, n, 0, 0, (hipStream_t)s);
    if (dtype == PDMK_F32) return ew<float, 4>(x, x, y, n, 0, 0, (hipStream_t)s);
    return -2;
}
extern "C" int pdmk_silu_bwd(const void* x, const void* dy, void* dx, int64_t n, int dtype, pdmk_stream s) {
    if (!x || !dy || !dx || n <= 0) return -1;
    if (dtype == PDMK_BF16) return ew<bf16, 1>(x, dy, dx, n, 0, 0, (hipStream_t)s);
    if (dtype == PDMK_F32) return ew<float, 1>(x, dy, dx, n, 0, 0, (hipStream_t)s);
    return -2;
}
extern "C" int pdmk_gelu_fwd(const void* x, void* y, int64_t n, int dtype, pdmk_stream s) {
    if (!x || !y || n <= 0) return -1;
    if (dtype == PDMK_BF16) return ew<bf16, 3>(x, x, y, n, 0, 0, (hipStream_t)s);
    if (dtype == PDMK_F32) return ew<float, 3>(x, x, y, n, 0, 0, (hipStream_t)s);
    return -2;
}
extern "C" int pdmk_axpby(const void* x, void* y, float alpha, float beta, int64_t n, int dtype, pdmk_stream s) {
    if (!x || !y || n <= 0) return -1;
    if (dtype == PDMK_BF16) return ew<bf16, 2>(x, x, y, n, alpha, beta, (hipStream_t)s);
    if (dtype == PDMK_F32) return ew<float, 2>(x, x, y, n, alpha, beta, (hipStream_t)s);
    return -2;
}
extern "C" int pdmk_copy2d(const void* src, void* dst, int64_t rows, int cols, int lds, int ldd, int accumulate,
                           int dtype, pdmk_stream s) {
    if (!src || !dst || rows <= 0 || cols <= 0) return -1;
    PDMK_DISPATCH(dtype, copy2d, src, dst, (long)rows, cols, lds, ldd, accumulate, (hipStream_t)s);
}
extern "C" int pdmk_cast_permute(const float* src, void* dst, int n0, int n1, int n2, int mode, int dtype,
                                 pdmk_stream s) {
    if (!src || !dst || n0 <= 0 || n1 <= 0 || n2 <= 0 || mode < 0 || mode > 2) return -1;
    if (mode == 1 && n1 != 1) return -1;
    PDMK_DISPATCH(dtype, cast_permute, src, dst, n0, n1, n2, mode, (hipStream_t)s);
}
extern "C" int pdmk_transpose_tiles(const void* src, void* dst, const int32_t* table, int ntiles, int dtype,
                                    pdmk_stream s) {
    if (!src || !dst || !table || ntiles <= 0) return -1;
    if (dtype == PDMK_BF16)
        hipLaunchKernelGGL(transpose_tiles_kernel<bf16>, dim3(ntiles), dim3(NT), 0, (hipStream_t)s, (const bf16*)src,
                           (bf16*)dst, (const TrTile*)table);
    else if (dtype == PDMK_F32)
        hipLaunchKernelGGL(transpose_tiles_kernel<float>, dim3(ntiles), dim3(NT), 0, (hipStream_t)s, (const float*)src,
                           (float*)dst, (const TrTile*)table);
    else return -2;
    PDMK_CHECK_LAUNCH();
    return 0;
}
__global__ void zero_bytes_kernel(uint4* __restrict__ p, long n16) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
extern "C" int pdmk_zero(void* p, int64_t nbytes, pdmk_stream s) {
    if (!p || nbytes < 0 || (nbytes & 15) || ((uintptr_t)p & 15)) return -1;
    if (nbytes == 0) return 0;
    const long n16 = nbytes / 16;
    hipLaunchKernelGGL(zero_bytes_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, (hipStream_t)s, (uint4*)p, n16);
    PDMK_CHECK_LAUNCH();
    return 0;
}

extern "C" int pdmk_colsum(const void* x, float* out, int64_t rows, int N, int ld, int accumulate, int nbatch, int ldo,
                           int dtype, pdmk_stream s) {
    if (!x || !out || rows <= 0 || N <= 0 || nbatch <= 0) return -1;
    PDMK_DISPATCH(dtype, colsum, x, out, (long)rows, N, ld, accumulate, nbatch, ldo, (hipStream_t)s);
}
extern "C" int pdmk_pool2x2_sum(const void* src, void* dst, int B, int H, int W, int C, int dtype, pdmk_stream s) {
    if (!src || !dst || B <= 0 || H <= 0 || W <= 0 || C <= 0) return -1;
    PDMK_DISPATCH(dtype, pool, src, dst, B, H, W, C, (hipStream_t)s);
}
extern "C" int pdmk_timestep_embed(const int64_t* t, const float* freqs, void* out, int B, int dim, int dtype,
                                   pdmk_stream s) {
    if (!t || !freqs || !out || B <= 0 || dim <= 0 || (dim & 1)) return -1;
    dim3 grid(grid_for((long)B * dim / 2));
    if (dtype == PDMK_BF16) hipLaunchKernelGGL(temb_kernel<bf16>, grid, dim3(NT), 0, (hipStream_t)s, t, freqs, (bf16*)out, B, dim);
    else if (dtype == PDMK_F32) hipLaunchKernelGGL(temb_kernel<float>, grid, dim3(NT), 0, (hipStream_t)s, t, freqs, (float*)out, B, dim);
    else return -2;
    PDMK_CHECK_LAUNCH();
    return 0;
}
extern "C" int pdmk_add_noise_velocity(const float* x0, const float* noise, const int64_t* t, const float* sa,
                                       const float* sb, void* noisy, float* target, int B, int C, int HW, int cpad,
                                       int dtype, pdmk_stream s) {
    if (!x0 || !noise || !t || !sa || !sb || !noisy || B <= 0 || C <= 0 || HW <= 0 || cpad < C) return -1;
    dim3 grid(grid_for((long)B * HW * cpad));
    if (dtype == PDMK_BF16)
        hipLaunchKernelGGL(noise_kernel<bf16>, grid, dim3(NT), 0, (hipStream_t)s, x0, noise, t, sa, sb, (bf16*)noisy, target, B, C, HW, cpad);
    else if (dtype == PDMK_F32)
        hipLaunchKernelGGL(noise_kernel<float>, grid, dim3(NT), 0, (hipStream_t)s, x0, noise, t, sa, sb, (float*)noisy, target, B, C, HW, cpad);
    else return -2;
    PDMK_CHECK_LAUNCH();
    return 0;
}
extern "C" int pdmk_nchw_to_nhwc(const float* src, void* dst, int B, int C, int HW, int cpad, int dtype,
                                 pdmk_stream s) {
    if (!src || !dst || B <= 0 || C <= 0 || HW <= 0 || cpad < C) return -1;
    dim3 grid(grid_for((long)B * HW * cpad));
    if (dtype == PDMK_BF16) hipLaunchKernelGGL(nchw2nhwc_kernel<bf16>, grid, dim3(NT), 0, (hipStream_t)s, src, (bf16*)dst, B, C, HW, cpad);
    else if (dtype == PDMK_F32) hipLaunchKernelGGL(nchw2nhwc_kernel<float>, grid, dim3(NT), 0, (hipStream_t)s, src, (float*)dst, B, C, HW, cpad);
    else return -2;
    PDMK_CHECK_LAUNCH();
    return 0;
}
extern "C" int pdmk_nhwc_to_nchw(const void* src, float* dst, int B, int C, int HW, int ld, int dtype, pdmk_stream s) {
    if (!src || !dst || B <= 0 || C <= 0 || HW <= 0 || ld < C) return -1;
    dim3 grid(grid_for((long)B * HW * C));
    if (dtype == PDMK_BF16) hipLaunchKernelGGL(nhwc2nchw_kernel<bf16>, grid, dim3(NT), 0, (hipStream_t)s, (const bf16*)src, dst, B, C, HW, ld);
    else if (dtype == PDMK_F32) hipLaunchKernelGGL(nhwc2nchw_kernel<float>, grid, dim3(NT), 0, (hipStream_t)s, (const float*)src, dst, B, C, HW, ld);
    else return -2;
    PDMK_CHECK_LAUNCH();
    return 0;
}
extern "C" int pdmk_mse_fwd(const void* a, int a_dtype, const void* b, int b_dtype, const float* w, double* out,
                            int slot, int B, int64_t rows_per_b, int cols, int lda, int ldb, double scale,
                            pdmk_stream s) {
    if (!a || !b || !out || B <= 0 || rows_per_b <= 0 || cols <= 0) return -1;
    if ((a_dtype | b_dtype) & ~1) return -2;
    dim3 grid(grid_for(rows_per_b * cols, 256), B);
    hipLaunchKernelGGL(mse_fwd_kernel, grid, dim3(NT), 0, (hipStream_t)s, a, a_dtype, b, b_dtype, w, out, slot,
                       (long)rows_per_b, cols, lda, ldb, scale);
    PDMK_CHECK_LAUNCH();
    return 0;
}
extern "C" int pdmk_mse_fwd_bwd(const void* a, int a_dtype, const void* b, int b_dtype, const float* w, double* out,
                                int slot, void* da, int B, int64_t rows_per_b, int cols, int lda, int ldb, int ldda,
                                double scale, float gscale, int accumulate, pdmk_stream s) {
    if (!a || !b || (!out && !da) || B <= 0 || rows_per_b <= 0 || cols <= 0) return -1;
    if ((a_dtype | b_dtype) & ~1) return -2;
    const int va = a_dtype == PDMK_BF16 ? 8 : 4, vb = b_dtype == PDMK_BF16 ? 8 : 4;
    if ((cols & 7) || (lda % va) || (ldb % vb) || (da && (ldda % va)) || ((uintptr_t)a & 15) || ((uintptr_t)b & 15) ||
        ((uintptr_t)da & 15))
        return -1;
    dim3 grid(grid_for((rows_per_b * (cols >> 3) + 3) / 4, B >= 8 ? 96 : (B >= 2 ? 256 : 512)), B);
#define PDMK_MSEV(TA, TB)                                                                                              \
    hipLaunchKernelGGL((mse_vec_kernel<TA, TB>), grid, dim3(NT), 0, (hipStream_t)s, (const TA*)a, (const TB*)b, w, out, slot, \
                       (TA*)da, (long)rows_per_b, cols, lda, ldb, ldda, scale, gscale, accumulate)
    if (a_dtype == PDMK_BF16 && b_dtype == PDMK_BF16) PDMK_MSEV(bf16, bf16);
    else if (a_dtype == PDMK_BF16) PDMK_MSEV(bf16, float);
    else if (b_dtype == PDMK_BF16) PDMK_MSEV(float, bf16);
    else PDMK_MSEV(float, float);
#undef PDMK_MSEV
    PDMK_CHECK_LAUNCH();
    return 0;
}
extern "C" int pdmk_mse_bwd(const void* a, int a_dtype, const void* b, int b_dtype, const float* w, void* da, int B,
                            int64_t rows_per_b, int cols, int lda, int ldb, int ldda, float gscale, int accumulate,
                            pdmk_stream s) {
    if (!a || !b || !da || B <= 0 || rows_per_b <= 0 || cols <= 0) return -1;
    if ((a_dtype | b_dtype) & ~1) return -2;
    dim3 grid(grid_for(rows_per_b * cols, 256), B);
    hipLaunchKernelGGL(mse_bwd_kernel, grid, dim3(NT), 0, (hipStream_t)s, a, a_dtype, b, b_dtype, w, da,
                       (long)rows_per_b, cols, lda, ldb, ldda, gscale, accumulate);
    PDMK_CHECK_LAUNCH();
    return 0;
}
extern "C" int pdmk_adamw(float* p, float* g, float* m, float* v, int64_t n, const float* lr, float beta1, float beta2,
                          float eps, float weight_decay, const float* bias_corr, float grad_scale, int zero_grad,
                          void* w_bf16, pdmk_stream s) {
    if (!p || !g || !m || !v || !lr || !bias_corr || n <= 0 || (n & 3)) return -1;
    // (tools/adamw_bench.py, 77 M parameters: one float4 per stream and thread 458 us = 5.68 TB/s of the 34 B per parameter, two 437 us = 5.95;
    //  4096 or 16 384 blocks are slower)
    hipLaunchKernelGGL(adamw_kernel<2>, dim3(grid_for(n / 8, 8192)), dim3(NT), 0, (hipStream_t)s, p, g, m, v, (long)n, lr,
                       beta1, beta2, eps, weight_decay, bias_corr, grad_scale, zero_grad, (bf16*)w_bf16);
    PDMK_CHECK_LAUNCH();
    return 0;
}
extern "C" int pdmk_splitk_finish_group(const pdmk_slab_item* items, int n_items, pdmk_stream s) {
    if (!items || n_items <= 0 || n_items > PDMK_SLAB_GROUP_MAX) return -1;
    SlabGroup gr;
    long nmax = 0;
    for (int i = 0; i < n_items; ++i) {
        const pdmk_slab_item& t = items[i];
        if (!t.ws || !t.dst || t.n <= 0 || (t.n & 3) || t.nslab <= 0 || (((uintptr_t)t.ws | (uintptr_t)t.dst) & 15)) return -1;
        gr.it[i] = t;
        nmax = t.n > nmax ? t.n : nmax;
    }
    for (int i = n_items; i < PDMK_SLAB_GROUP_MAX; ++i) gr.it[i] = items[0];
    long bx = (nmax / 4 + NT - 1) / NT;
    bx = bx > 512 ? 512 : (bx < 1 ? 1 : bx);
    hipLaunchKernelGGL(splitk_finish_group_kernel, dim3((unsigned)bx, n_items), dim3(NT), 0, (hipStream_t)s, gr);
    PDMK_CHECK_LAUNCH();
    return 0;
}
extern "C" int pdmk_splitk_finish(const float* ws, void* C, const float* bias, const float* rowvec, const void* R,
                                  int64_t M, int N, int ldc, int ldr, int rows_per_b, int ldrv, int nslab, int accumulate,
                                  int dtype, pdmk_stream s) {
    if (!ws || !C || M <= 0 || N <= 0 || (N & 3) || (rowvec && rows_per_b <= 0) || nslab < 1) return -1;
    if (ldrv <= 0) ldrv = N;
    dim3 grid(grid_for(M * (N / 4)));
    if (dtype == PDMK_BF16)
        hipLaunchKernelGGL(splitk_finish_kernel<bf16>, grid, dim3(NT), 0, (hipStream_t)s, ws, (bf16*)C, bias, rowvec,
                           (const bf16*)R, (long)M, N, ldc, ldr, rows_per_b, ldrv, nslab, accumulate);
    else if (dtype == PDMK_F32)
        hipLaunchKernelGGL(splitk_finish_kernel<float>, grid, dim3(NT), 0, (hipStream_t)s, ws, (float*)C, bias, rowvec,
                           (const float*)R, (long)M, N, ldc, ldr, rows_per_b, ldrv, nslab, accumulate);
    else return -2;
    PDMK_CHECK_LAUNCH();
    return 0;
}
extern "C" int pdmk_splitk_finish_colstat(const float* ws, void* C, const float* bias, const float* rowvec, const void* R,
                                          int64_t M, int N, int ldc, int ldr, int rows_per_b, int ldrv, int nslab, int accumulate,
                                          int64_t* colstat, int cs_ld, int cs_col0, int dtype, pdmk_stream s) {
    if (!ws || !C || !colstat || M <= 0 || N <= 0 || (N & 3) || rows_per_b <= 0 || (rows_per_b % 64) || (M % 64) || nslab < 1 ||
        cs_col0 < 0 || cs_ld < cs_col0 + N)
        return -1;
    if (ldrv <= 0) ldrv = N;
    dim3 grid((unsigned)(M / 64), (unsigned)((N + 63) / 64));
    if (dtype == PDMK_BF16)
        hipLaunchKernelGGL(splitk_finish_cs_kernel<bf16>, grid, dim3(NT), 0, (hipStream_t)s, ws, (bf16*)C, bias, rowvec,
                           (const bf16*)R, (long)M, N, ldc, ldr, rows_per_b, ldrv, nslab, accumulate, colstat + cs_col0, cs_ld);
    else if (dtype == PDMK_F32)
        hipLaunchKernelGGL(splitk_finish_cs_kernel<float>, grid, dim3(NT), 0, (hipStream_t)s, ws, (float*)C, bias, rowvec,
                           (const float*)R, (long)M, N, ldc, ldr, rows_per_b, ldrv, nslab, accumulate, colstat + cs_col0, cs_ld);
    else return -2;
    PDMK_CHECK_LAUNCH();
    return 0;
}
extern "C" int pdmk_up2_pack_weights(const float* w3, void* wp, void* wpt, int Co, int Ci, int dtype, pdmk_stream s) {
    if (!w3 || !wp || Co <= 0 || Ci <= 0 || (Ci & 7) || ((uintptr_t)w3 & 15) || ((uintptr_t)wp & 15)) return -1;
    const dim3 grid(grid_for(4L * Co * 4 * (Ci / 8), 8192));
    if (dtype == PDMK_BF16)
        hipLaunchKernelGGL(up2_pack_kernel<bf16>, grid, dim3(NT), 0, (hipStream_t)s, w3, (bf16*)wp, (bf16*)wpt, Co, Ci);
    else if (dtype == PDMK_F32)
        hipLaunchKernelGGL(up2_pack_kernel<float>, grid, dim3(NT), 0, (hipStream_t)s, w3, (float*)wp, (float*)wpt, Co, Ci);
    else return -2;
    PDMK_CHECK_LAUNCH();
    return 0;
}
extern "C" int pdmk_up2_combine_wgrad(const float* dwp, float* dw3, int Co, int Ci, pdmk_stream s) {
    if (!dwp || !dw3 || Co <= 0 || Ci <= 0) return -1;
    hipLaunchKernelGGL(up2_combine_kernel, dim3(grid_for((long)Co * 9 * Ci, 4096)), dim3(NT), 0, (hipStream_t)s, dwp, dw3, Co, Ci);
    PDMK_CHECK_LAUNCH();
    return 0;
}
extern "C" int pdmk_sumsq(const float* x, int64_t n, double* out, int slot, pdmk_stream s) {
    if (!x || !out || n <= 0) return -1;
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n, 2048)), dim3(NT), 0, (hipStream_t)s, x, (long)n, out, slot);
    PDMK_CHECK_LAUNCH();
    return 0;
}
