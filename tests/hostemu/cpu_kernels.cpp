// TEST INFRASTRUCTURE -- CPU stand-ins for the launch_* interfaces of csrc/kernels.h whose HIP kernels are block-cooperative
// (LDS tiles, MFMA, cross-lane reductions): each kernel's documented SEMANTICS (not its tiling) in plain loops, fp32 only.
// The thread-independent kernels (stream_kernels.hip, speaker_kernels.hip, most of encoder_kernels.hip) are NOT restated
// here: tests/hostemu/build.py compiles their real sources against the sequential interpreter in hip/hip_runtime.h.  Linked with csrc/codec_engine.hip and csrc/encoder_engine.hip
// compiled as host C++ (tests/hostemu/build.py) so that the CPU test-suite runs the engines' real orchestration code:
// weight repacking in finalize(), buffer rotation, strides, the streaming carries, the C ABI.  Not a product path:
// libqtts_hostemu.so is only loaded by tests/test_hostemu.py.
#include <cmath>
#include <vector>
#include "common.h"
#include "kernels.h"

namespace qtts {

// gemm_tap.hip: C[m][n] = epi( sum_tap sum_k A[m + shift[tap]][k] * W[tap][n][k] ), zero row when (m % T) + shift < 0
void launch_gemm_tap(const GemmTapParams& p, bool bf16, hipStream_t) {
    QTTS_REQUIRE(!bf16, QTTS_ERR_ARG, "host emulation runs fp32 engines only");
    QTTS_REQUIRE(p.K % 32 == 0, QTTS_ERR_ARG, "gemm_tap: K must be a multiple of 32");
    QTTS_REQUIRE(p.taps >= 1 && p.taps <= 8, QTTS_ERR_ARG, "gemm_tap: 1..8 taps");
    QTTS_REQUIRE(p.M > 0 && p.N > 0, QTTS_ERR_ARG, "gemm_tap: empty problem");
    QTTS_REQUIRE(p.lda % 4 == 0, QTTS_ERR_ARG, "gemm_tap: lda must be a multiple of 4");
    if (p.act == ACT_SWIGLU) QTTS_REQUIRE(p.N % 32 == 0, QTTS_ERR_ARG, "gemm_tap: swiglu needs N % 32 == 0");
    const float* W = reinterpret_cast<const float*>(p.W);
    std::vector<float> acc(p.N);
    for (int m = 0; m < p.M; ++m) {
        const int t = m % p.T;
        for (int n = 0; n < p.N; ++n) acc[n] = 0.f;
        for (int tap = 0; tap < p.taps; ++tap) {
            const int sh = p.shift[tap];
            if (t + sh < 0) continue;
            const float* a = p.A + (size_t)(m + sh) * p.lda;
            const float* w = W + (size_t)tap * p.N * p.K;
            for (int n = 0; n < p.N; ++n) {
                const float* wr = w + (size_t)n * p.K;
                float s = 0.f;
                for (int k = 0; k < p.K; ++k) s += a[k] * wr[k];
                acc[n] += s;
            }
        }
        if (p.act == ACT_SWIGLU) {          // 16-row blocks alternate gate / up for the same 16 output features
            for (int n = 0; n < p.N; ++n) {
                if ((n / 16) % 2) continue;
                const float gt = acc[n], u = acc[n + 16];
                p.C[(size_t)m * p.ldc + (n / 32) * 16 + n % 16] = (gt / (1.f + expf(-gt))) * u;
            }
            continue;
        }
        for (int n = 0; n < p.N; ++n) {
            float v = acc[n] + (p.bias ? p.bias[n] : 0.f);
            if (p.act == ACT_GELU) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
            else if (p.act == ACT_SNAKE) { const float sn = sinf(v * p.snake_ea[n]); v = v + p.snake_ib[n] * (sn * sn); }
            else if (p.act == ACT_SILU) v = v / (1.f + expf(-v));
            v *= p.scale ? p.scale[n] : 1.f;
            if (p.res) v += p.res[(size_t)m * p.ldr + n];
            p.C[(size_t)m * p.ldc + n] = v;
        }
    }
}

void launch_rmsnorm(const float* x, int ldx, const float* w, float eps, float* y, int ldy, int rows, int C, hipStream_t) {
    for (int r = 0; r < rows; ++r) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += x[(size_t)r * ldx + c] * x[(size_t)r * ldx + c];
        const float rs = 1.f / sqrtf(s / (float)C + eps);
        for (int c = 0; c < C; ++c) y[(size_t)r * ldy + c] = w[c] * (x[(size_t)r * ldx + c] * rs);
    }
}

void launch_snake(const float* x, const float* ea, const float* ib, float* y, int64_t rows, int C, hipStream_t) {
    for (int64_t r = 0; r < rows; ++r)
        for (int c = 0; c < C; ++c) {
            const float v = x[r * C + c], s = sinf(v * ea[c]);
            y[r * C + c] = v + ib[c] * (s * s);
        }
}

void launch_dwconv_ln(const float* x, const float* w7, const float* b, const float* ln_w, const float* ln_b, float eps,
                      float* y, int rows, int T, int C, hipStream_t) {
    std::vector<float> v(C);
    for (int row = 0; row < rows; ++row) {
        const int t = row % T;
        float s = 0.f;
        for (int c = 0; c < C; ++c) {
            float acc = b[c];
            for (int k = 0; k < 7; ++k)
                if (t - 6 + k >= 0) acc += w7[c * 7 + k] * x[(size_t)(row - 6 + k) * C + c];
            v[c] = acc; s += acc;
        }
        const float mean = s / (float)C;
        float q = 0.f;
        for (int c = 0; c < C; ++c) q += (v[c] - mean) * (v[c] - mean);
        const float r = 1.f / sqrtf(q / (float)C + eps);
        for (int c = 0; c < C; ++c) y[(size_t)row * C + c] = (v[c] - mean) * r * ln_w[c] + ln_b[c];
    }
}

void launch_rvq_gather(const int64_t* codes, int B, int Q, int T, int64_t sb, int64_t sq, int64_t stt, int t0, int Tc,
                       const float* tables, int bins, int vq, float* out, hipStream_t) {
    (void)T;
    for (int row = 0; row < B * Tc; ++row) {
        const int b = row / Tc, t = row % Tc + t0;
        const int64_t* cp = codes + b * sb + t * stt;
        for (int j = 0; j < vq; ++j) {
            int64_t c0 = cp[0]; if (c0 < 0) c0 = 0;
            out[(size_t)row * 2 * vq + j] = tables[(size_t)c0 * vq + j];
            float acc = 0.f;
            for (int q = 1; q < Q; ++q) {
                int64_t c = cp[q * sq]; if (c < 0) c = 0;
                const float e = tables[((size_t)q * bins + c) * vq + j];
                acc = q == 1 ? e : acc + e;
            }
            out[(size_t)row * 2 * vq + vq + j] = acc;
        }
    }
}

void launch_final_conv(const float* x, const float* w, float bias, float* wav, float* pre, int64_t rows, int64_t T, int C,
                       int64_t out_stride_b, int64_t skip, hipStream_t) {
    const int64_t B = rows / T;
    for (int64_t b = 0; b < B; ++b)
        for (int64_t t = skip; t < T; ++t) {
            float acc = 0.f;
            for (int k = 0; k < 7; ++k) {
                const int64_t s = t - 6 + k;
                if (s < 0) continue;
                for (int c = 0; c < C; ++c) acc += w[k * C + c] * x[((size_t)b * T + s) * C + c];
            }
            const float v = acc + bias;
            const size_t o = (size_t)b * out_stride_b + (size_t)(t - skip);
            if (pre) pre[o] = v;
            wav[o] = fminf(fmaxf(v, -1.f), 1.f);
        }
}

static void rope_rows(float* qkv, int ld, int rows, int T, int pos0, int nheads, int hd, const float* inv_freq) {
    const int half = hd / 2;
    for (int r = 0; r < rows; ++r)
        for (int h = 0; h < nheads; ++h)
            for (int d = 0; d < half; ++d) {
                const float ang = (float)(pos0 + r % T) * inv_freq[d];
                const float c = cosf(ang), s = sinf(ang);
                float* p = qkv + (size_t)r * ld + h * hd;
                const float x0 = p[d], x1 = p[d + half];
                p[d] = x0 * c - x1 * s;
                p[d + half] = x1 * c + x0 * s;
            }
}
void launch_rope_inplace(float* qkv, int ld, int rows, int T, int n_heads_total, int hd, const float* inv_freq, hipStream_t) {
    rope_rows(qkv, ld, rows, T, 0, n_heads_total, hd, inv_freq);
}

}  // namespace qtts
