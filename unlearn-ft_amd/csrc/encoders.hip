// Kernels only the encoders in front of the step need (trainer.py:2405-2406, data_utils.py:155-191):
//   pdmk_softmax_rows  : row softmax of a materialised fp32 score matrix.  The encoder's one attention layer has a single
//                        512-wide head over 4096 tokens per image (diffusers AutoencoderKL mid block; CompVis twin
//                        ldm/modules/diffusionmodules/model.py:150-204): scores are two plain GEMMs per image around this
//                        pass (the U-Net's flash kernels are specialised for head dim 64).
//   pdmk_latent_sample : DiagonalGaussianDistribution.sample() * scaling_factor on NHWC moments -> NCHW fp32 latents.
//   pdmk_embed_tokens  : CLIP text embeddings, token row + position row.
#include "vec.h"

namespace {

constexpr int NT = 256;

// one workgroup per row; cols <= NT * 4 * PER (row cached in registers between the three sweeps)
template <typename T, int PER>
__global__ __launch_bounds__(NT) void softmax_rows_kernel(const float* __restrict__ s, T* __restrict__ p, int cols, long lds,
                                                          long ldp) {
    __shared__ float red[2][NT / 64];
    const float* srow = s + (long)blockIdx.x * lds;
    T* prow = p + (long)blockIdx.x * ldp;
    float v[PER][4];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = (i * NT + threadIdx.x) * 4;
        if (c < cols) {
            const float4 x = *reinterpret_cast<const float4*>(srow + c);
            v[i][0] = x.x; v[i][1] = x.y; v[i][2] = x.z; v[i][3] = x.w;
        } else {
            v[i][0] = v[i][1] = v[i][2] = v[i][3] = -INFINITY;
        }
        m = fmaxf(m, fmaxf(fmaxf(v[i][0], v[i][1]), fmaxf(v[i][2], v[i][3])));
    }
    m = wave_max(m);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) red[0][wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[i][j] = __builtin_amdgcn_exp2f((v[i][j] - m) * 1.44269504088896341f);
            sum += v[i][j];
        }
    sum = wave_sum(sum);
    if (lane == 0) red[1][wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[1][0] + red[1][1] + red[1][2] + red[1][3]);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = (i * NT + threadIdx.x) * 4;
        if (c < cols) {
#pragma unroll
            for (int j = 0; j < 4; ++j) prow[c + j] = from_f32<T>(v[i][j] * inv);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void latent_sample_kernel(const T* __restrict__ mom, int ld, const float* __restrict__ eps,
                                                           float* __restrict__ z, int B, int C, int HW, float scale) {
    const long n = (long)B * C * HW;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        const int px = (int)(i % HW);
        const int c = (int)((i / HW) % C);
        const long b = i / ((long)HW * C);
        const T* row = mom + (b * HW + px) * ld;
        const float mean = to_f32(row[c]);
        const float logvar = fminf(fmaxf(to_f32(row[C + c]), -30.0f), 20.0f);
        z[i] = (mean + expf(0.5f * logvar) * eps[i]) * scale;
    }
}

// out[i, :] = tok[clamp(ids[i]), :] + pos[i % T, :]   (one 16-byte chunk per thread)
template <typename T>
__global__ __launch_bounds__(NT) void embed_tokens_kernel(const int64_t* __restrict__ ids, const T* __restrict__ tok,
                                                          const T* __restrict__ pos, T* __restrict__ out, long ntok, int T_,
                                                          int D, int vocab, int ldt, int ldp, int ldo) {
    constexpr int V = Vec<T>::N;
    const int cpr = D / V;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < ntok * cpr; i += (long)gridDim.x * NT) {
        const long r = i / cpr;
        const int c = (int)(i - r * cpr) * V;
        long id = ids[r];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        float a[V], b[V];
        Vec<T>::load(tok + id * ldt + c, a);
        Vec<T>::load(pos + (r % T_) * ldp + c, b);
#pragma unroll
        for (int e = 0; e < V; ++e) a[e] += b[e];
        Vec<T>::store(out + r * ldo + c, a);
    }
}

template <typename T> int softmax_launch(const float* s, void* p, long rows, int cols, long lds, long ldp, hipStream_t st) {
    T* pt = reinterpret_cast<T*>(p);
    if (cols <= NT * 4) hipLaunchKernelGGL((softmax_rows_kernel<T, 1>), dim3(rows), dim3(NT), 0, st, s, pt, cols, lds, ldp);
    else if (cols <= NT * 8) hipLaunchKernelGGL((softmax_rows_kernel<T, 2>), dim3(rows), dim3(NT), 0, st, s, pt, cols, lds, ldp);
    else if (cols <= NT * 16) hipLaunchKernelGGL((softmax_rows_kernel<T, 4>), dim3(rows), dim3(NT), 0, st, s, pt, cols, lds, ldp);
    else if (cols <= NT * 64) hipLaunchKernelGGL((softmax_rows_kernel<T, 16>), dim3(rows), dim3(NT), 0, st, s, pt, cols, lds, ldp);
    else return -1;
    PDMK_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int pdmk_softmax_rows(const float* s, void* p, int64_t rows, int cols, int64_t lds, int64_t ldp, int dtype,
                                 pdmk_stream stream) {
    if (!s || !p || rows <= 0 || rows >= (1L << 31) || cols <= 0 || (cols & 3) || (lds & 3) || lds < cols || ldp < cols ||
        ((uintptr_t)s & 15))
        return -1;
    PDMK_DISPATCH(dtype, softmax_launch, s, p, (long)rows, cols, (long)lds, (long)ldp, (hipStream_t)stream);
}

extern "C" int pdmk_embed_tokens(const int64_t* ids, const void* tok, const void* pos, void* out, int64_t ntok, int T,
                                 int D, int vocab, int ldt, int ldp, int ldo, int dtype, pdmk_stream stream) {
    const int v = dtype == PDMK_BF16 ? 8 : 4;
    if (!ids || !tok || !pos || !out || ntok <= 0 || T <= 0 || D <= 0 || vocab <= 0 || (D % v) || (ldt % v) || (ldp % v) ||
        (ldo % v) || (((uintptr_t)tok | (uintptr_t)pos | (uintptr_t)out) & 15))
        return -1;
    const long items = ntok * (D / v);
    dim3 grid((unsigned)((items + NT - 1) / NT < 8192 ? (items + NT - 1) / NT : 8192));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PDMK_BF16) hipLaunchKernelGGL(embed_tokens_kernel<bf16>, grid, dim3(NT), 0, st, ids, (const bf16*)tok, (const bf16*)pos, (bf16*)out, (long)ntok, T, D, vocab, ldt, ldp, ldo);
    else if (dtype == PDMK_F32) hipLaunchKernelGGL(embed_tokens_kernel<float>, grid, dim3(NT), 0, st, ids, (const float*)tok, (const float*)pos, (float*)out, (long)ntok, T, D, vocab, ldt, ldp, ldo);
    else return -2;
    PDMK_CHECK_LAUNCH();
    return 0;
}

extern "C" int pdmk_latent_sample(const void* moments, int ld, const float* eps, float* latents, int B, int C, int HW,
                                  float scale, int dtype, pdmk_stream stream) {
    if (!moments || !eps || !latents || B <= 0 || C <= 0 || HW <= 0 || ld < 2 * C) return -1;
    const long n = (long)B * C * HW;
    dim3 grid((unsigned)((n + NT - 1) / NT < 4096 ? (n + NT - 1) / NT : 4096));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == PDMK_BF16) hipLaunchKernelGGL(latent_sample_kernel<bf16>, grid, dim3(NT), 0, st, (const bf16*)moments, ld, eps, latents, B, C, HW, scale);
    else if (dtype == PDMK_F32) hipLaunchKernelGGL(latent_sample_kernel<float>, grid, dim3(NT), 0, st, (const float*)moments, ld, eps, latents, B, C, HW, scale);
    else return -2;
    PDMK_CHECK_LAUNCH();
    return 0;
}
