// speaker_kernels.hip -- the non-GEMM kernels of the speaker branch (ECAPA-TDNN + log-mel front end).
// The STFT is a 4-tap GEMM (n_fft / hop rows of `hop` samples per frame against a Hann-windowed DFT matrix), the mel
// projection and every TDNN / 1x1 conv are GEMMs too (gemm_tap.hip); what is left are paddings, activations and the
// per-channel time statistics.  STATUS round 1: compiled, orchestration executed on CPU stand-ins, no hardware run yet.
#include "common.h"
#include "kernels.h"

namespace qtts {

__device__ __forceinline__ int reflect_index(int i, int n) {     // torch 'reflect': -1 -> 1, n -> n-2
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

__global__ __launch_bounds__(256) void reflect_rows_1d_kernel(const float* wav, int S, int pad, int R, int hop, float* out) {
    const int r = blockIdx.x, b = blockIdx.y;
    for (int j = threadIdx.x; j < hop; j += 256)
        out[((size_t)b * R + r) * hop + j] = wav[(size_t)b * S + reflect_index(r * hop + j - pad, S)];
}
void launch_reflect_rows_1d(const float* wav, int S, int pad, int R, int hop, float* out, int B, hipStream_t st) {
    QTTS_REQUIRE(S > pad, QTTS_ERR_ARG, "mel: waveform shorter than the reflect padding");
    hipLaunchKernelGGL(reflect_rows_1d_kernel, dim3(R, B), dim3(256), 0, st, wav, S, pad, R, hop, out);
    QTTS_CHECK_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void magnitude_pad_kernel(const float* y, int ldy, int nb, float* out, int Kp) {
    const int64_t row = blockIdx.x;
    for (int f = threadIdx.x; f < Kp; f += 256) {
        float v = 0.f;
        if (f < nb) {
            const float re = y[row * ldy + f], im = y[row * ldy + nb + f];
            v = sqrtf(re * re + im * im + 1e-9f);
        }
        out[row * Kp + f] = v;
    }
}
void launch_magnitude_pad(const float* y, int ldy, int nb, float* out, int Kp, int64_t rows, hipStream_t st) {
    hipLaunchKernelGGL(magnitude_pad_kernel, dim3((unsigned)rows), dim3(256), 0, st, y, ldy, nb, out, Kp);
    QTTS_CHECK_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void reflect_pad_add_rows_kernel(const float* src1, int ld1, const float* src2, int ld2, int T,
                                                                   int p, int C, float* dst) {
    const int i = blockIdx.x, b = blockIdx.y;
    const int s = reflect_index(i - p, T);
    const float* a = src1 + ((size_t)b * T + s) * ld1;
    const float* a2 = src2 ? src2 + ((size_t)b * T + s) * ld2 : nullptr;
    float* to = dst + ((size_t)b * (T + 2 * p) + i) * C;
    for (int c = threadIdx.x; c < C; c += 256) to[c] = a[c] + (a2 ? a2[c] : 0.f);
}
void launch_reflect_pad_add_rows(const float* src1, int ld1, const float* src2, int ld2, int T, int p, int C, float* dst, int B,
                                 hipStream_t st) {
    QTTS_REQUIRE(T > p, QTTS_ERR_ARG, "reflect pad: sequence shorter than the padding");
    hipLaunchKernelGGL(reflect_pad_add_rows_kernel, dim3(T + 2 * p, B), dim3(256), 0, st, src1, ld1, src2, ld2, T, p, C, dst);
    QTTS_CHECK_HIP(hipGetLastError());
}

__device__ __forceinline__ float row_act(float v, int act) {
    switch (act) {
        case ROWACT_RELU: return fmaxf(v, 0.f);
        case ROWACT_RELU_TANH: return tanhf(fmaxf(v, 0.f));
        case ROWACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case ROWACT_LOG_CLAMP: return logf(fmaxf(v, 1e-5f));
        default: return v;
    }
}
__global__ __launch_bounds__(256) void copy_act_rows_kernel(const float* src, int lds, int Tsrc, int skip, int n, int C, int act,
                                                            float* dst, int ldd) {
    const int t = blockIdx.x, b = blockIdx.y;
    const float* from = src + ((size_t)b * Tsrc + skip + t) * lds;
    float* to = dst + ((size_t)b * n + t) * ldd;
    for (int c = threadIdx.x; c < C; c += 256) to[c] = row_act(from[c], act);
}
void launch_copy_act_rows(const float* src, int lds, int Tsrc, int skip, int n, int C, int act, float* dst, int ldd, int B,
                          hipStream_t st) {
    QTTS_REQUIRE(n >= 1 && skip >= 0 && skip + n <= Tsrc, QTTS_ERR_ARG, "copy_act_rows: bad shape");
    hipLaunchKernelGGL(copy_act_rows_kernel, dim3(n, B), dim3(256), 0, st, src, lds, Tsrc, skip, n, C, act, dst, ldd);
    QTTS_CHECK_HIP(hipGetLastError());
}

// one thread per (b, c) column walks the time axis twice (T is a few hundred frames; C >= 128 keeps the loads coalesced)
__global__ __launch_bounds__(256) void col_stats_kernel(const float* x, int ldx, const float* att, int T, int C, float* mean,
                                                        float* sd, int ld_out) {
    const int c = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (c >= C) return;
    const float* xb = x + (size_t)b * T * ldx + c;
    const float* ab = att ? att + (size_t)b * T * C + c : nullptr;
    const float wu = 1.f / (float)T;
    float m = 0.f;
    for (int t = 0; t < T; ++t) m += (ab ? ab[(size_t)t * C] : wu) * xb[(size_t)t * ldx];
    float q = 0.f;
    for (int t = 0; t < T; ++t) { const float d = xb[(size_t)t * ldx] - m; q += (ab ? ab[(size_t)t * C] : wu) * d * d; }
    mean[(size_t)b * ld_out + c] = m;
    if (sd) sd[(size_t)b * ld_out + c] = sqrtf(fmaxf(q, 1e-12f));
}
void launch_col_stats(const float* x, int ldx, const float* att, int T, int C, float* mean, float* sd, int ld_out, int B,
                      hipStream_t st) {
    hipLaunchKernelGGL(col_stats_kernel, dim3((C + 255) / 256, B), dim3(256), 0, st, x, ldx, att, T, C, mean, sd, ld_out);
    QTTS_CHECK_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void scale_add_rows_kernel(const float* h, int ldh, const float* gate, const float* r, int ldr,
                                                             float* out, int ldo, int T, int C) {
    const int t = blockIdx.x, b = blockIdx.y;
    const size_t row = (size_t)b * T + t;
    for (int c = threadIdx.x; c < C; c += 256) out[row * ldo + c] = h[row * ldh + c] * gate[(size_t)b * C + c] + r[row * ldr + c];
}
void launch_scale_add_rows(const float* h, int ldh, const float* gate, const float* r, int ldr, float* out, int ldo, int T, int C,
                           int B, hipStream_t st) {
    hipLaunchKernelGGL(scale_add_rows_kernel, dim3(T, B), dim3(256), 0, st, h, ldh, gate, r, ldr, out, ldo, T, C);
    QTTS_CHECK_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void concat_stats_kernel(const float* x, int ldx, const float* mean, const float* sd, int T, int C,
                                                           float* out) {
    const int t = blockIdx.x, b = blockIdx.y;
    const size_t row = (size_t)b * T + t;
    for (int c = threadIdx.x; c < C; c += 256) {
        out[row * 3 * C + c] = x[row * ldx + c];
        out[row * 3 * C + C + c] = mean[(size_t)b * C + c];
        out[row * 3 * C + 2 * C + c] = sd[(size_t)b * C + c];
    }
}
void launch_concat_stats(const float* x, int ldx, const float* mean, const float* sd, int T, int C, float* out, int B, hipStream_t st) {
    hipLaunchKernelGGL(concat_stats_kernel, dim3(T, B), dim3(256), 0, st, x, ldx, mean, sd, T, C, out);
    QTTS_CHECK_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void softmax_time_kernel(float* a, int T, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (c >= C) return;
    float* ab = a + (size_t)b * T * C + c;
    float m = -INFINITY;
    for (int t = 0; t < T; ++t) m = fmaxf(m, ab[(size_t)t * C]);
    float l = 0.f;
    for (int t = 0; t < T; ++t) { const float e = expf(ab[(size_t)t * C] - m); ab[(size_t)t * C] = e; l += e; }
    const float inv = 1.f / l;
    for (int t = 0; t < T; ++t) ab[(size_t)t * C] *= inv;
}
void launch_softmax_time(float* a, int T, int C, int B, hipStream_t st) {
    hipLaunchKernelGGL(softmax_time_kernel, dim3((C + 255) / 256, B), dim3(256), 0, st, a, T, C);
    QTTS_CHECK_HIP(hipGetLastError());
}

}  // namespace qtts
