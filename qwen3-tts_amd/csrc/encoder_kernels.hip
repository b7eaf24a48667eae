// encoder_kernels.hip -- the few non-GEMM kernels of the codec ENCODER (Mimi encode) for gfx950.
// Everything heavy (SEANet convs incl. the strided ones as 2-tap GEMMs over "super-rows", the transformer's linears,
// the codebook scores) runs on gemm_tap.hip; attention on attn_rows.  STATUS round 1: compiled, not yet run on hardware.
#include "common.h"
#include "kernels.h"

namespace qtts {

__global__ __launch_bounds__(256) void elu_kernel(const float4* x, float4* y, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 v = x[i];
        v.x = v.x > 0.f ? v.x : expm1f(v.x);          // F.elu, alpha = 1
        v.y = v.y > 0.f ? v.y : expm1f(v.y);
        v.z = v.z > 0.f ? v.z : expm1f(v.z);
        v.w = v.w > 0.f ? v.w : expm1f(v.w);
        y[i] = v;
    }
}
void launch_elu(const float* x, float* y, int64_t n, hipStream_t st) {
    QTTS_REQUIRE(n % 4 == 0, QTTS_ERR_ARG, "elu: n % 4");
    const int64_t n4 = n / 4;
    const int grid = (int)std::min<int64_t>((n4 + 255) / 256, 65535);
    hipLaunchKernelGGL(elu_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<const float4*>(x),
                       reinterpret_cast<float4*>(y), n4);
    QTTS_CHECK_HIP(hipGetLastError());
}

// MimiConv1d(audio_channels = 1 -> C, k) causal: out[b*L + t][c] = bias[c] + sum_j w[c][j] * wav[b][t - (k-1) + j]
__global__ __launch_bounds__(256) void conv_in1_kernel(const float* wav, const float* w, const float* bias, float* out, int L,
                                                       int C, int k) {
    const int b = blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * (256 / 64) + (threadIdx.x >> 6);     // 4 positions per block, 64 lanes over channels
    if (t >= L) return;
    float xv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t s = t - (k - 1) + j;
        xv[j] = (j < k && s >= 0) ? wav[(size_t)b * L + s] : 0.f;
    }
    for (int c = threadIdx.x & 63; c < C; c += 64) {
        float acc = bias[c];
        for (int j = 0; j < k; ++j) acc += w[c * k + j] * xv[j];
        out[((size_t)b * L + t) * C + c] = acc;
    }
}
void launch_conv_in1(const float* wav, const float* w, const float* bias, float* out, int B, int L, int C, int k, hipStream_t st) {
    QTTS_REQUIRE(k >= 1 && k <= 8, QTTS_ERR_ARG, "conv_in1: kernel size 1..8");
    hipLaunchKernelGGL(conv_in1_kernel, dim3((L + 3) / 4, B), dim3(256), 0, st, wav, w, bias, out, L, C, k);
    QTTS_CHECK_HIP(hipGetLastError());
}

__device__ inline float block_sum_ln(float v, float* sm) {
    v = wave_sum64_dpp(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    __syncthreads();
    return r;
}
// nn.LayerNorm over the channel dim (biased variance, eps inside the sqrt)
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, int ldx, const float* w, const float* b, float eps,
                                                        float* y, int ldy, int C) {
    __shared__ float sm[4];
    const int row = blockIdx.x;
    const float* xr = x + (size_t)row * ldx;
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) s += xr[c];
    const float mean = block_sum_ln(s, sm) / (float)C;
    float q = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) { const float d = xr[c] - mean; q += d * d; }
    const float r = rsqrtf(block_sum_ln(q, sm) / (float)C + eps);
    for (int c = threadIdx.x; c < C; c += 256) y[(size_t)row * ldy + c] = (xr[c] - mean) * r * w[c] + b[c];
}
void launch_layernorm(const float* x, int ldx, const float* w, const float* b, float eps, float* y, int ldy, int rows, int C,
                      hipStream_t st) {
    hipLaunchKernelGGL(layernorm_kernel, dim3(rows), dim3(256), 0, st, x, ldx, w, b, eps, y, ldy, C);
    QTTS_CHECK_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void pad_rows_kernel(const float* src, int T, int left, int Tp, int replicate, float* dst,
                                                       int C4) {
    const int i = blockIdx.x, b = blockIdx.y;
    int s = i - left;
    const bool inside = s >= 0 && s < T;
    s = s < 0 ? 0 : (s >= T ? T - 1 : s);
    float4* to = reinterpret_cast<float4*>(dst) + ((size_t)b * Tp + i) * C4;
    const float4* from = reinterpret_cast<const float4*>(src) + ((size_t)b * T + s) * C4;
    for (int c = threadIdx.x; c < C4; c += 256) to[c] = (inside || replicate) ? from[c] : make_float4(0.f, 0.f, 0.f, 0.f);
}
void launch_pad_rows(const float* src, int T, int left, int right, int replicate, float* dst, int B, int C, hipStream_t st) {
    QTTS_REQUIRE(C % 4 == 0 && T >= 1 && left >= 0 && right >= 0, QTTS_ERR_ARG, "pad_rows: bad shape");
    hipLaunchKernelGGL(pad_rows_kernel, dim3(left + T + right, B), dim3(256), 0, st, src, T, left, left + T + right, replicate,
                       dst, C / 4);
    QTTS_CHECK_HIP(hipGetLastError());
}

// One workgroup per row: argmin over the codebook (MimiEuclideanCodebook.quantize: argmin of the Euclidean distance ==
// argmin of ||e||^2 - 2 r.e), lowest index on ties (torch.argmin), then the residual update of the RVQ loop.
__global__ __launch_bounds__(256) void vq_argmin_update_kernel(const float* scores, int bins, const float* enorm,
                                                               const float* table, int D, float* r, int64_t* codes_out,
                                                               int64_t stride_b, int T) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int row = blockIdx.x;
    const float* sr = scores + (size_t)row * bins;
    float bv = INFINITY;
    int bi = 0x7fffffff;
    for (int j = threadIdx.x; j < bins; j += 256) {
        const float d = enorm[j] - 2.0f * sr[j];
        if (d < bv || (d == bv && j < bi)) { bv = d; bi = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    bv = sv[0]; bi = si[0];
    for (int w2 = 1; w2 < 4; ++w2)
        if (sv[w2] < bv || (sv[w2] == bv && si[w2] < bi)) { bv = sv[w2]; bi = si[w2]; }
    if (bi < 0 || bi >= bins) bi = 0;                  // all-NaN scores must not become an out-of-range gather
    for (int c = threadIdx.x; c < D; c += 256) r[(size_t)row * D + c] -= table[(size_t)bi * D + c];
    if (threadIdx.x == 0) codes_out[(size_t)(row / T) * stride_b + (row % T)] = bi;
}
void launch_vq_argmin_update(const float* scores, int bins, const float* enorm, const float* table, int D, float* r,
                             int64_t* codes_out, int64_t stride_b, int B, int T, hipStream_t st) {
    hipLaunchKernelGGL(vq_argmin_update_kernel, dim3(B * T), dim3(256), 0, st, scores, bins, enorm, table, D, r, codes_out,
                       stride_b, T);
    QTTS_CHECK_HIP(hipGetLastError());
}

}  // namespace qtts
