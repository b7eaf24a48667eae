// elementwise.hip -- memory-bound helper kernels (channel-last activations, 16-B vector access).
#include "common.h"
#include "kernels.h"

namespace qtts {

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// block-wide sum for 256-thread blocks; `sm` = 4 floats of LDS
__device__ inline float block_sum(float v, float* sm) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

// ---------------------------------------------------------------------------------- RMSNorm
// Qwen3TTS(TokenizerV2Decoder)RMSNorm: y = w * (x * rsqrt(mean(x^2) + eps)), fp32 throughout.
template <bool OUT16>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* x, int ldx, const float* w, float eps, float* y,
                                                      int ldy, int C) {
    __shared__ float sm[4];
    const float* xr = x + (size_t)blockIdx.x * ldx;
    float s = 0.f;
    for (int c = threadIdx.x * 4; c < C; c += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(xr + c);
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    s = block_sum(s, sm);
    const float r = rsqrtf(s / (float)C + eps);
    for (int c = threadIdx.x * 4; c < C; c += 1024) {
        const float4 v = *reinterpret_cast<const float4*>(xr + c);
        const float4 g = *reinterpret_cast<const float4*>(w + c);
        float4 o;
        o.x = g.x * (v.x * r); o.y = g.y * (v.y * r); o.z = g.z * (v.z * r); o.w = g.w * (v.w * r);
        if constexpr (OUT16) {           // y is a bf16 [rows][ldy] buffer: the consumer is a bf16 GEMM that would round the same way
            uint2 h; h.x = pack_bf16(o.x, o.y); h.y = pack_bf16(o.z, o.w);
            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(y) + (size_t)blockIdx.x * ldy + c) = h;
        } else {
            *reinterpret_cast<float4*>(y + (size_t)blockIdx.x * ldy + c) = o;
        }
    }
}
void launch_rmsnorm(const float* x, int ldx, const float* w, float eps, float* y, int ldy, int rows, int C,
                    hipStream_t st) {
    QTTS_REQUIRE(C % 4 == 0, QTTS_ERR_ARG, "rmsnorm: C % 4");
    hipLaunchKernelGGL(rmsnorm_kernel<false>, dim3(rows), dim3(256), 0, st, x, ldx, w, eps, y, ldy, C);
    QTTS_CHECK_HIP(hipGetLastError());
}
void launch_rmsnorm16(const float* x, int ldx, const float* w, float eps, void* y16, int ldy, int rows, int C,
                      hipStream_t st) {
    QTTS_REQUIRE(C % 4 == 0 && ldy % 4 == 0, QTTS_ERR_ARG, "rmsnorm16: C % 4, ldy % 4");
    hipLaunchKernelGGL(rmsnorm_kernel<true>, dim3(rows), dim3(256), 0, st, x, ldx, w, eps, reinterpret_cast<float*>(y16), ldy, C);
    QTTS_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------- SnakeBeta
// y = x + ib[c] * sin(x * ea[c])^2 with ea = exp(alpha), ib = 1/(exp(beta)+1e-9) (tokenizer v2:602-616)
__global__ __launch_bounds__(256) void snake_kernel(const float* x, const float* ea, const float* ib, float* y,
                                                    int64_t n4, int C4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        const float4 a = *reinterpret_cast<const float4*>(ea + c);
        const float4 b = *reinterpret_cast<const float4*>(ib + c);
        float4 o;
        float s;
        s = sinf(v.x * a.x); o.x = v.x + b.x * (s * s);
        s = sinf(v.y * a.y); o.y = v.y + b.y * (s * s);
        s = sinf(v.z * a.z); o.z = v.z + b.z * (s * s);
        s = sinf(v.w * a.w); o.w = v.w + b.w * (s * s);
        reinterpret_cast<float4*>(y)[i] = o;
    }
}
void launch_snake(const float* x, const float* ea, const float* ib, float* y, int64_t rows, int C, hipStream_t st) {
    QTTS_REQUIRE(C % 4 == 0, QTTS_ERR_ARG, "snake: C % 4");
    const int64_t n4 = rows * C / 4;
    const int grid = (int)std::min<int64_t>((n4 + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(snake_kernel, dim3(grid), dim3(256), 0, st, x, ea, ib, y, n4, C / 4);
    QTTS_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------- ConvNeXt front half
// depthwise causal conv k=7 (groups = C) + LayerNorm(eps) over channels (tokenizer v2:227-232)
__global__ __launch_bounds__(256) void dwconv_ln_kernel(const float* x, const float* w7, const float* b,
                                                        const float* ln_w, const float* ln_b, float eps, float* y,
                                                        int T, int C) {
    __shared__ float sm[4];
    const int row = blockIdx.x;
    const int t = row % T;
    float v[8];
    float s = 0.f;
    int n = 0;
    for (int c = threadIdx.x; c < C; c += 256, ++n) {
        float acc = b[c];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const int tt = t - 6 + k;
            if (tt >= 0) acc += w7[c * 7 + k] * x[(size_t)(row - 6 + k) * C + c];
        }
        v[n] = acc;
        s += acc;
    }
    const float mean = block_sum(s, sm) / (float)C;
    float q = 0.f;
    for (int i = 0; i < n; ++i) { const float d = v[i] - mean; q += d * d; }
    const float var = block_sum(q, sm) / (float)C;
    const float r = rsqrtf(var + eps);
    n = 0;
    for (int c = threadIdx.x; c < C; c += 256, ++n) y[(size_t)row * C + c] = (v[n] - mean) * r * ln_w[c] + ln_b[c];
}
void launch_dwconv_ln(const float* x, const float* w7, const float* b, const float* ln_w, const float* ln_b,
                      float eps, float* y, int rows, int T, int C, hipStream_t st) {
    QTTS_REQUIRE(C <= 2048, QTTS_ERR_ARG, "dwconv_ln: C <= 2048");
    hipLaunchKernelGGL(dwconv_ln_kernel, dim3(rows), dim3(256), 0, st, x, w7, b, ln_w, ln_b, eps, y, T, C);
    QTTS_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------- RVQ dequant (gather-sum)
// tables [Q][bins][vq] already normalised (embedding_sum / clamp(cluster_usage, 1e-5), tokenizer v2:677).
// out row = [ table_0[code_0] | sum_{q=1..Q-1} table_q[code_q] ] (sequential sum order = v2:721-727).
__global__ __launch_bounds__(256) void rvq_gather_kernel(const int64_t* codes, int Q, int64_t sb, int64_t sq,
                                                         int64_t stt, int t0, int Tc, const float* tables,
                                                         int bins, int vq, float* out, int* err) {
    const int row = blockIdx.x;          // b * Tc + t
    const int b = row / Tc, t = row % Tc + t0;
    const int64_t* cp = codes + b * sb + t * stt;
    // an index past the codebook is an error (the reference's embedding lookup raises): flag it for the host, read row 0
    if (threadIdx.x < Q && cp[threadIdx.x * sq] >= bins) *err = 1;
    for (int j = threadIdx.x; j < 2 * vq; j += 256) {
        float acc;
        if (j < vq) {
            int64_t c = cp[0]; if (c < 0 || c >= bins) c = 0;
            acc = tables[((size_t)c) * vq + j];
        } else {
            acc = 0.f;
            for (int q = 1; q < Q; ++q) {
                int64_t c = cp[q * sq]; if (c < 0 || c >= bins) c = 0;
                const float e = tables[((size_t)q * bins + c) * vq + (j - vq)];
                acc = (q == 1) ? e : acc + e;
            }
        }
        out[(size_t)row * 2 * vq + j] = acc;
    }
}
void launch_rvq_gather(const int64_t* codes, int B, int Q, int T, int64_t stride_b, int64_t stride_q,
                       int64_t stride_t, int t0, int Tc, const float* tables, int bins, int vq, float* out,
                       int* err, hipStream_t st) {
    (void)T;
    QTTS_REQUIRE(Q <= 256 && err, QTTS_ERR_ARG, "rvq_gather: Q <= 256 and an error flag");
    hipLaunchKernelGGL(rvq_gather_kernel, dim3(B * Tc), dim3(256), 0, st, codes, Q, stride_b, stride_q, stride_t,
                       t0, Tc, tables, bins, vq, out, err);
    QTTS_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------- final conv C -> 1, k = 7
// x already SnakeBeta-activated, channel-last [B*T][C]; w [7][C]; out = clamp(conv + bias, -1, 1) (v2:884)
__global__ __launch_bounds__(256) void final_conv_kernel(const float* x, const float* w, float bias, float* wav,
                                                         float* pre, int64_t T, int C, int64_t out_stride_b,
                                                         int64_t skip) {
    extern __shared__ __attribute__((aligned(16))) float sm_fc[];
    const int LS = C + 1;
    float* xs = sm_fc;                 // [70][C+1]
    float* ws = sm_fc + 70 * LS;       // [7][C]
    const int b = blockIdx.y;
    const int64_t t0 = (int64_t)blockIdx.x * 64;
    const float* xb = x + (size_t)b * T * C;
    for (int i = threadIdx.x; i < 70 * C; i += 256) {
        const int r = i / C, c = i % C;
        const int64_t t = t0 - 6 + r;
        xs[r * LS + c] = (t >= 0 && t < T) ? xb[(size_t)t * C + c] : 0.f;
    }
    for (int i = threadIdx.x; i < 7 * C; i += 256) ws[i] = w[i];
    __syncthreads();
    const int tl = threadIdx.x >> 2, part = threadIdx.x & 3;
    float acc = 0.f;
    for (int k = 0; k < 7; ++k)
        for (int c = part; c < C; c += 4) acc += ws[k * C + c] * xs[(tl + k) * LS + c];
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    const int64_t t = t0 + tl;
    if (part == 0 && t < T && t >= skip) {
        const float v = acc + bias;
        const size_t o = (size_t)b * out_stride_b + (size_t)(t - skip);
        if (pre) pre[o] = v;
        wav[o] = fminf(fmaxf(v, -1.f), 1.f);
    }
}
void launch_final_conv(const float* x, const float* w, float bias, float* wav, float* pre_clamp, int64_t rows,
                       int64_t T, int C, int64_t out_stride_b, int64_t skip, hipStream_t st) {
    const int B = (int)(rows / T);
    const size_t lds = (70 * (C + 1) + 7 * C) * sizeof(float);
    QTTS_REQUIRE(lds <= 64 * 1024, QTTS_ERR_ARG, "final_conv: C too large");
    hipLaunchKernelGGL(final_conv_kernel, dim3((unsigned)((T + 63) / 64), B), dim3(256), lds, st, x, w, bias, wav,
                       pre_clamp, T, C, out_stride_b, skip);
    QTTS_CHECK_HIP(hipGetLastError());
}

// bf16 mode: x is the bf16 copy the last residual unit's epilogue leaves, the final SnakeBeta already applied ([B*T][C], C % 8 == 0).
// One output per thread, 256 per workgroup; the 262 input rows are staged once as bf16 pairs, row stride C / 2 + 1 words (odd for
// C = 96: a wave's column reads fall into 64 different banks); the taps are uniform per instruction and come through the scalar cache.
// HBM: 2 C bytes per output instead of the fp32 path's 4 C (fp32 tensor out of the unit) + 8 C (stand-alone SnakeBeta) + 4 C.
__global__ __launch_bounds__(256) void final_conv16_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w, float bias,
                                                           float* wav, float* pre, int64_t T, int C, int64_t out_stride_b,
                                                           int64_t skip) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float sm_fc[];
    unsigned* xs = reinterpret_cast<unsigned*>(sm_fc);           // [262][C / 2 + 1]
    const int LW = C / 2 + 1, upr = C / 8;
    const int b = blockIdx.y;
    const int64_t t0 = (int64_t)blockIdx.x * 256;
    const bf16_t* xb = x + (size_t)b * T * C;
    for (int i = threadIdx.x; i < 262 * upr; i += 256) {
        const int r = i / upr, u = i - r * upr;
        const int64_t t = t0 - 6 + r;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (t >= 0 && t < T) v = *reinterpret_cast<const u32x4*>(xb + (size_t)t * C + u * 8);
        unsigned* d = xs + r * LW + u * 4;
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    __syncthreads();
    const unsigned* xr = xs + threadIdx.x * LW;
    float acc0 = 0.f, acc1 = 0.f;
    for (int k = 0; k < 7; ++k) {
        const float* wk = w + k * C;
        const unsigned* xk = xr + k * LW;
#pragma unroll 8
        for (int c2 = 0; c2 < C / 2; ++c2) {
            const unsigned pr = xk[c2];
            acc0 += wk[2 * c2] * __uint_as_float(pr << 16);
            acc1 += wk[2 * c2 + 1] * __uint_as_float(pr & 0xffff0000u);
        }
    }
    const int64_t t = t0 + threadIdx.x;
    if (t < T && t >= skip) {
        const float v = acc0 + acc1 + bias;
        const size_t o = (size_t)b * out_stride_b + (size_t)(t - skip);
        if (pre) pre[o] = v;
        wav[o] = fminf(fmaxf(v, -1.f), 1.f);
    }
}
void launch_final_conv16(const bf16_t* x, const float* w, float bias, float* wav, float* pre_clamp, int64_t rows,
                         int64_t T, int C, int64_t out_stride_b, int64_t skip, hipStream_t st) {
    const int B = (int)(rows / T);
    QTTS_REQUIRE(C % 8 == 0, QTTS_ERR_ARG, "final_conv16: C % 8");
    const size_t lds = (size_t)262 * (C / 2 + 1) * 4;
    QTTS_REQUIRE(lds <= 64 * 1024, QTTS_ERR_ARG, "final_conv16: C too large");
    hipLaunchKernelGGL(final_conv16_kernel, dim3((unsigned)((T + 255) / 256), B), dim3(256), lds, st, x, w, bias, wav,
                       pre_clamp, T, C, out_stride_b, skip);
    QTTS_CHECK_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------------- RoPE in place (codec)
// rotate-half RoPE, position = row % T, on `nheads` consecutive heads of width hd starting at column 0
__global__ __launch_bounds__(256) void rope_inplace_kernel(float* qkv, int ld, int T, int nheads, int hd,
                                                           const float* inv_freq, int64_t total) {
    const int half = hd / 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int d = (int)(i % half);
        const int h = (int)((i / half) % nheads);
        const int64_t row = i / ((int64_t)half * nheads);
        const float ang = (float)(row % T) * inv_freq[d];
        const float c = cosf(ang), s = sinf(ang);
        float* p = qkv + row * ld + h * hd;
        const float x1 = p[d], x2 = p[d + half];
        p[d] = x1 * c - x2 * s;            // q*cos + rotate_half(q)*sin, first half: -x2
        p[d + half] = x2 * c + x1 * s;     //                                second half: +x1
    }
}
void launch_rope_inplace(float* qkv, int ld, int rows, int T, int n_heads_total, int hd, const float* inv_freq,
                         hipStream_t st) {
    const int64_t total = (int64_t)rows * n_heads_total * (hd / 2);
    const int grid = (int)std::min<int64_t>((total + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(rope_inplace_kernel, dim3(grid), dim3(256), 0, st, qkv, ld, T, n_heads_total, hd, inv_freq,
                       total);
    QTTS_CHECK_HIP(hipGetLastError());
}

}  // namespace qtts
