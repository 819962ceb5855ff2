// Skinny linears (M <= 16 rows): the time-embedding MLP and the per-ResBlock time_emb_proj (blocks.py:334-341,
// unet_2d_conditional.py:1514-1533) have M = batch.  A 128x128 MFMA tile wastes 94 % of its rows there and serialises
// the whole reduction inside 5-10 workgroups; these kernels are plain weight-streaming dot products instead:
//   pdmk_skinny_gemm : y[m][n] (+)= sum_k x[m][k] w[n][k] (+ bias[n])     forward and (with w = W^T) dgrad
//   pdmk_skinny_wgrad: dw[n][k] += sum_m dy[m][n] x[m][k],  dbias[n] += sum_m dy[m][n]
// The skinny operand lives in LDS (<= 64 KiB), every weight byte is read exactly once with 16-byte loads: HBM/L2
// streaming bound (a 1280x1280 bf16 weight = 3.3 MB).
#include "vec.h"

namespace {

constexpr int MMAX = 16;
constexpr int NPW = 4;          // output columns per wave pass (x fragments from LDS are reused across them)

template <typename XT> __device__ __forceinline__ void lds_chunk(const XT* p, float* f, int n);
template <> __device__ __forceinline__ void lds_chunk<bf16>(const bf16* p, float* f, int n) {
    if (n == 8) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
    } else {
        const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = (float)v[i];
    }
}
template <> __device__ __forceinline__ void lds_chunk<float>(const float* p, float* f, int n) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w;
    if (n == 8) {
        const float4 b = *reinterpret_cast<const float4*>(p + 4);
        f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    }
}

// T: weight dtype (chunk CH = 8 bf16 / 4 fp32), XT: skinny operand dtype, MR: rows handled (8 or 16; rows >= M are zero)
template <typename T, typename XT, int MR>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(const XT* __restrict__ x, const T* __restrict__ w, void* __restrict__ y,
                                                          const float* __restrict__ bias, int M, int N, int K, int ldx, int ldw,
                                                          int ldy, int out_f32, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    XT* xs = reinterpret_cast<XT*>(smem_raw);                       // [MR][K], rows >= M zero
    constexpr int CH = Vec<T>::N;
    for (int i = threadIdx.x * 4; i < MR * K; i += 256 * 4) {       // K % 4 == 0
        const int m = i / K, k = i - m * K;
#pragma unroll
        for (int j = 0; j < 4; ++j) xs[i + j] = m < M ? x[(long)m * ldx + k + j] : (XT)0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * 4 + wave, nwaves = gridDim.x * 4;
    for (int n0 = gw * NPW; n0 < N; n0 += nwaves * NPW) {
        float acc[NPW][MR];
#pragma unroll
        for (int c = 0; c < NPW; ++c)
#pragma unroll
            for (int m = 0; m < MR; ++m) acc[c][m] = 0.f;
        for (int k0 = lane * CH; k0 < K; k0 += 64 * CH) {
            float wv[NPW][CH];
#pragma unroll
            for (int c = 0; c < NPW; ++c) {
                if (n0 + c < N) Vec<T>::load(w + (long)(n0 + c) * ldw + k0, wv[c]);
                else {
#pragma unroll
                    for (int j = 0; j < CH; ++j) wv[c][j] = 0.f;
                }
            }
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                float xv[CH];
                lds_chunk<XT>(xs + m * K + k0, xv, CH);
#pragma unroll
                for (int c = 0; c < NPW; ++c)
#pragma unroll
                    for (int j = 0; j < CH; ++j) acc[c][m] = fmaf(xv[j], wv[c][j], acc[c][m]);
            }
        }
#pragma unroll
        for (int c = 0; c < NPW; ++c)
#pragma unroll
            for (int m = 0; m < MR; ++m) acc[c][m] = wave_sum(acc[c][m]);
        // lane l < NPW*MR writes element (c = l / MR, m = l % MR)
        float mine = 0.f;
#pragma unroll
        for (int c = 0; c < NPW; ++c)
#pragma unroll
            for (int m = 0; m < MR; ++m)
                if (lane == c * MR + m) mine = acc[c][m];
        const int c = lane / MR, m = lane % MR, n = n0 + c;
        if (lane < NPW * MR && m < M && n < N) {
            float v = mine + (bias ? bias[n] : 0.f);
            const long off = (long)m * ldy + n;
            if (out_f32) {
                float* yf = reinterpret_cast<float*>(y);
                yf[off] = v + (accumulate ? yf[off] : 0.f);
            } else {
                T* yt = reinterpret_cast<T*>(y);
                yt[off] = from_f32<T>(v + (accumulate ? to_f32(yt[off]) : 0.f));
            }
        }
    }
}

// dw[n][k] += sum_m dy[m][n] x[m][k]; one block = 8 weight rows x all K; dy (DT) and x (T) staged in LDS
template <typename T, typename DT>
__global__ __launch_bounds__(256) void skinny_wgrad_kernel(const DT* __restrict__ dy, const T* __restrict__ x, float* __restrict__ dw,
                                                           float* __restrict__ dbias, int M, int N, int K, int lddy, int ldx,
                                                           int lddw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* dys = reinterpret_cast<float*>(smem_raw);                 // [8][MMAX]
    T* xs = reinterpret_cast<T*>(smem_raw + 8 * MMAX * 4);           // [M][K]
    const int nb = blockIdx.x * 8;
    for (int i = threadIdx.x; i < 8 * MMAX; i += 256) {
        const int r = i / MMAX, m = i % MMAX;
        dys[i] = (m < M && nb + r < N) ? to_f32(dy[(long)m * lddy + nb + r]) : 0.f;
    }
    for (int i = threadIdx.x * 4; i < M * K; i += 256 * 4) {
        const int m = i / K, k = i - m * K;
#pragma unroll
        for (int j = 0; j < 4; ++j) xs[i + j] = x[(long)m * ldx + k + j];
    }
    __syncthreads();
    const int kq = K / 4;
    for (int i = threadIdx.x; i < 8 * kq; i += 256) {
        const int r = i / kq, k = (i - r * kq) * 4;
        if (nb + r >= N) break;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        for (int m = 0; m < M; ++m) {
            const float d = dys[r * MMAX + m];
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = fmaf(d, to_f32(xs[m * K + k + j]), a[j]);
        }
        float4* p = reinterpret_cast<float4*>(dw + (long)(nb + r) * lddw + k);
        float4 o = *p;
        o.x += a[0]; o.y += a[1]; o.z += a[2]; o.w += a[3];
        *p = o;
    }
    if (dbias && threadIdx.x < 8 && nb + threadIdx.x < N) {
        float s = 0.f;
        for (int m = 0; m < M; ++m) s += dys[threadIdx.x * MMAX + m];
        dbias[nb + threadIdx.x] += s;
    }
}

template <typename T, typename XT>
int launch_gemm(const void* x, const void* w, void* y, const float* bias, int M, int N, int K, int ldx, int ldw, int ldy,
                int out_f32, int accumulate, hipStream_t st) {
    const int MR = M <= 8 ? 8 : 16;
    const size_t shm = (size_t)MR * K * sizeof(XT);
    if (shm > 64 * 1024) return -1;
    const int waves = (N + NPW - 1) / NPW;
    const int blocks = max(1, min((waves + 3) / 4, 1024));
    if (MR == 8)
        hipLaunchKernelGGL((skinny_gemm_kernel<T, XT, 8>), dim3(blocks), dim3(256), shm, st, (const XT*)x, (const T*)w, y, bias, M, N,
                           K, ldx, ldw, ldy, out_f32, accumulate);
    else
        hipLaunchKernelGGL((skinny_gemm_kernel<T, XT, 16>), dim3(blocks), dim3(256), shm, st, (const XT*)x, (const T*)w, y, bias, M,
                           N, K, ldx, ldw, ldy, out_f32, accumulate);
    PDMK_CHECK_LAUNCH();
    return 0;
}

template <typename T, typename DT>
int launch_wgrad(const void* dy, const void* x, float* dw, float* dbias, int M, int N, int K, int lddy, int ldx, int lddw,
                 hipStream_t st) {
    const size_t shm = 8 * MMAX * 4 + (size_t)M * K * sizeof(T);
    if (shm > 64 * 1024) return -1;
    hipLaunchKernelGGL((skinny_wgrad_kernel<T, DT>), dim3((N + 7) / 8), dim3(256), shm, st, (const DT*)dy, (const T*)x, dw, dbias, M,
                       N, K, lddy, ldx, lddw);
    PDMK_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int pdmk_skinny_gemm(const void* x, int x_dtype, const void* w, void* y, const float* bias, int M, int N, int K,
                                int ldx, int ldw, int ldy, int dtype, int out_f32, int accumulate, pdmk_stream stream) {
    if (!x || !w || !y || M <= 0 || M > MMAX || N <= 0 || K <= 0) return -1;
    const int ch = dtype == PDMK_BF16 ? 8 : 4;
    if ((K % ch) || (ldw % ch) || (ldx % 4) || ((uintptr_t)w & 15) || ((uintptr_t)x & 15)) return -1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == PDMK_BF16 && x_dtype == PDMK_BF16) return launch_gemm<bf16, bf16>(x, w, y, bias, M, N, K, ldx, ldw, ldy, out_f32, accumulate, st);
    if (dtype == PDMK_BF16 && x_dtype == PDMK_F32) return launch_gemm<bf16, float>(x, w, y, bias, M, N, K, ldx, ldw, ldy, out_f32, accumulate, st);
    if (dtype == PDMK_F32 && x_dtype == PDMK_F32) return launch_gemm<float, float>(x, w, y, bias, M, N, K, ldx, ldw, ldy, out_f32, accumulate, st);
    return -2;
}

extern "C" int pdmk_skinny_wgrad(const void* dy, int dy_dtype, const void* x, float* dw, float* dbias, int M, int N, int K,
                                 int lddy, int ldx, int lddw, int dtype, pdmk_stream stream) {
    if (!dy || !x || !dw || M <= 0 || M > MMAX || N <= 0 || K <= 0) return -1;
    if ((K % 4) || (lddw % 4) || (ldx % 4) || ((uintptr_t)dw & 15)) return -1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == PDMK_BF16 && dy_dtype == PDMK_BF16) return launch_wgrad<bf16, bf16>(dy, x, dw, dbias, M, N, K, lddy, ldx, lddw, st);
    if (dtype == PDMK_BF16 && dy_dtype == PDMK_F32) return launch_wgrad<bf16, float>(dy, x, dw, dbias, M, N, K, lddy, ldx, lddw, st);
    if (dtype == PDMK_F32 && dy_dtype == PDMK_F32) return launch_wgrad<float, float>(dy, x, dw, dbias, M, N, K, lddy, ldx, lddw, st);
    return -2;
}
