// 16-byte vector access helpers: a "chunk" is 8 bf16 or 4 fp32 (one dwordx4 per lane — the coalescing sweet spot).
#pragma once
#include "common.h"

template <typename T> struct Vec;
template <> struct Vec<bf16> {
    static constexpr int N = 8;
    typedef bf16x8 raw;                                  // the 16-byte chunk as it is loaded (software-pipelined loops keep
    static __device__ __forceinline__ raw load_raw(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }   // these)
    static __device__ __forceinline__ void unpack(const raw& v, float* f) {
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
    }
    static __device__ __forceinline__ void load(const bf16* p, float* f) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
    }
    static __device__ __forceinline__ void store(bf16* p, const float* f) {
        bf16x8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (bf16)f[i];
        st_stream(reinterpret_cast<bf16x8*>(p), v);     // outputs of the streaming kernels: consumed by later launches
    }
};
template <> struct Vec<float> {
    static constexpr int N = 4;
    typedef f32x4 raw;
    static __device__ __forceinline__ raw load_raw(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void unpack(const raw& v, float* f) { f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3]; }
    static __device__ __forceinline__ void load(const float* p, float* f) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    }
    static __device__ __forceinline__ void store(float* p, const float* f) {
        st_stream(reinterpret_cast<f32x4*>(p), f32x4{f[0], f[1], f[2], f[3]});
    }
};

#define PDMK_DISPATCH(dtype, FN, ...)                      \
    do {                                                   \
        if ((dtype) == PDMK_BF16) return FN<bf16>(__VA_ARGS__); \
        if ((dtype) == PDMK_F32) return FN<float>(__VA_ARGS__); \
        return -2;                                         \
    } while (0)
