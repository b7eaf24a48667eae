// stream_kernels.hip -- the three small kernels the state-carrying codec decode adds: everything else in
// that path reuses the validated kernels of the non-streaming decoder.  Thread-independent code (no LDS, no cross-lane
// ops): tests/hostemu also executes THESE SOURCES on the CPU through a sequential block/thread interpreter.
#include "common.h"
#include "kernels.h"

namespace qtts {

// ---------------------------------------------------------------------------------- streaming codec decode helpers
// Every layer of the decoder is causal, so a packet of new frames only needs, per stateful layer, the last
// (k-1)*dilation input rows of the previous packet (oracle/codec_stream_ref.py).  `stage_rows` builds
// [carried rows | new rows] so that the UNCHANGED conv / attention kernels run on it; `save_tail` refreshes the carry.
__global__ __launch_bounds__(256) void stage_rows_kernel(const float* src, int src_T, int skip, int n, const float* state,
                                                         int h, float* dst, int C4) {
    const int r = blockIdx.x, b = blockIdx.y;
    const float4* from = r < h ? reinterpret_cast<const float4*>(state) + ((size_t)b * h + r) * C4
                               : reinterpret_cast<const float4*>(src) + ((size_t)b * src_T + skip + (r - h)) * C4;
    float4* to = reinterpret_cast<float4*>(dst) + ((size_t)b * (h + n) + r) * C4;
    for (int c = threadIdx.x; c < C4; c += 256) to[c] = from[c];
}
void launch_stage_rows(const float* src, int src_T, int skip, int n, const float* state, int h, float* dst, int B, int C,
                       hipStream_t st) {
    QTTS_REQUIRE(C % 4 == 0 && n >= 1 && h >= 0 && skip >= 0 && skip + n <= src_T, QTTS_ERR_ARG, "stage_rows: bad shape");
    QTTS_REQUIRE(h == 0 || state, QTTS_ERR_ARG, "stage_rows: state missing");
    hipLaunchKernelGGL(stage_rows_kernel, dim3(h + n, B), dim3(256), 0, st, src, src_T, skip, n, state, h, dst, C / 4);
    QTTS_CHECK_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void save_tail_kernel(const float* buf, int Tp, float* state, int h, int C4) {
    const int j = blockIdx.x, b = blockIdx.y;
    const float4* from = reinterpret_cast<const float4*>(buf) + ((size_t)b * Tp + (Tp - h + j)) * C4;
    float4* to = reinterpret_cast<float4*>(state) + ((size_t)b * h + j) * C4;
    for (int c = threadIdx.x; c < C4; c += 256) to[c] = from[c];
}
void launch_save_tail(const float* buf, int Tp, float* state, int h, int B, int C, hipStream_t st) {
    if (h == 0) return;
    QTTS_REQUIRE(C % 4 == 0 && Tp >= h, QTTS_ERR_ARG, "save_tail: bad shape");
    hipLaunchKernelGGL(save_tail_kernel, dim3(h, B), dim3(256), 0, st, buf, Tp, state, h, C / 4);
    QTTS_CHECK_HIP(hipGetLastError());
}

// rotate-half RoPE at positions pos0 + (row % T) (a packet that starts at frame pos0 of its stream)
__global__ __launch_bounds__(256) void rope_offset_kernel(float* qkv, int ld, int T, int pos0, int nheads, int hd,
                                                          const float* inv_freq, int64_t total) {
    const int half = hd / 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int d = (int)(i % half);
        const int h = (int)((i / half) % nheads);
        const int64_t row = i / ((int64_t)half * nheads);
        const float ang = (float)(pos0 + (int)(row % T)) * inv_freq[d];
        const float c = cosf(ang), s = sinf(ang);
        float* p = qkv + row * ld + h * hd;
        const float x0 = p[d], x1 = p[d + half];
        p[d] = x0 * c - x1 * s;
        p[d + half] = x1 * c + x0 * s;
    }
}
void launch_rope_offset(float* qkv, int ld, int rows, int T, int pos0, int n_heads_total, int hd, const float* inv_freq,
                        hipStream_t st) {
    const int64_t total = (int64_t)rows * n_heads_total * (hd / 2);
    const int grid = (int)std::min<int64_t>((total + 255) / 256, 65535);
    hipLaunchKernelGGL(rope_offset_kernel, dim3(grid), dim3(256), 0, st, qkv, ld, T, pos0, n_heads_total, hd, inv_freq, total);
    QTTS_CHECK_HIP(hipGetLastError());
}

}  // namespace qtts
