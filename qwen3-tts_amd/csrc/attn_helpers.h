// attn_helpers.h -- the KV-cache element helpers the decode attentions share (attention.hip; cp_layer.hip's attention stage).
#pragma once
#include "common.h"

namespace qtts {
template <typename KVT> __device__ inline KVT kv_cast(float v);
template <> __device__ inline float kv_cast<float>(float v) { return v; }
template <> __device__ inline bf16_t kv_cast<bf16_t>(float v) { return f32_to_bf16(v); }
__device__ inline float kv_load(const float* p) { return *p; }
__device__ inline float kv_load(const bf16_t* p) { return bf16_to_f32(*p); }
// softmax exponential of the decode attentions.  bf16 cache (the benchmarked mode): v_exp_f32 on x * log2(e), 2 instructions
// (~1e-7 relative: far inside what the bf16 K / V carry); fp32 cache (the parity mode): the library expf the goldens were taken with
// -- 10 instructions, and the key loop of attn_tk spends a fifth of its VALU time in them.
template <typename KVT>
__device__ inline float att_exp(float x) {
    if constexpr (sizeof(KVT) == 2) return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
    else return expf(x);
}
// cache element type of the fused code-predictor launches: bf16 engines bf16, fp32 engines (the exact parity mode) float
template <bool F32> struct CpaoKv { typedef bf16_t type; };
template <> struct CpaoKv<true> { typedef float type; };
}  // namespace qtts
