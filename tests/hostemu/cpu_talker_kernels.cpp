// TEST INFRASTRUCTURE -- CPU stand-ins for the launch interfaces csrc/talker_engine.hip uses (fp32 / exact mode only):
// the weight-streaming decode GEMM over its PACKED tile layout, fused q/k-norm + RoPE + paged-KV attention, the HF
// logits processors + greedy pick, and the device-resident loop-state glue.  Each restates the documented semantics of
// the HIP kernel it replaces (csrc/kernels.h, csrc/glue.h), so that the talker's real C++ -- weight packing in finalize(),
// prefill, the frame step, the stop bookkeeping, the C ABI -- runs in the CPU suite (tests/test_hostemu.py).
#include <cmath>
#include <vector>
#include "common.h"
#include "kernels.h"
#include "glue.h"

namespace qtts {

// ---- skinny.hip host helpers (same layout: [N/fs strips][K/KT k-tiles][4 k-slices][fs features][16 B])
bool skinny_can_stage(int, int, bool bf16) {
    QTTS_REQUIRE(!bf16, QTTS_ERR_ARG, "host emulation runs fp32 engines only");
    return false;
}
size_t skinny_packed_bytes(int N, int K, bool bf16) { return (size_t)N * K * (bf16 ? 2 : 4); }
void pack_skinny_weight(const float* W, int N, int K, bool bf16, void* out_host, const float* g, int fs) {
    QTTS_REQUIRE(!bf16, QTTS_ERR_ARG, "host emulation runs fp32 engines only");
    const int KT = 16, nkt = K / KT, strips = N / fs;
    float* out = reinterpret_cast<float*>(out_host);
    for (int s = 0; s < strips; ++s)
        for (int kt = 0; kt < nkt; ++kt)
            for (int q = 0; q < 4; ++q)
                for (int i = 0; i < fs; ++i) {
                    const size_t tile = ((size_t)s * nkt + kt) * (fs * 4) + q * fs + i;
                    const int k0 = kt * KT + q * 4;
                    const float* src = W + (size_t)(s * fs + i) * K + k0;
                    for (int e = 0; e < 4; ++e) out[tile * 4 + e] = g ? src[e] * g[k0 + e] : src[e];
                }
}
static inline float packed_w(const float* Wp, int K, int fs, int n, int k) {       // inverse of the layout above
    const int KT = 16, nkt = K / KT;
    const int s = n / fs, i = n % fs, kt = k / KT, q = (k % KT) / 4, e = k % 4;
    return Wp[(((size_t)s * nkt + kt) * (fs * 4) + q * fs + i) * 4 + e];
}

// skinny.hip: out[m][n] = epi( rstd[m] * sum_k x[m][k] * W'[n][k] ), rstd from ss_in when `norm`; SwiGLU pairs 16-row strips
void launch_skinny(const SkinnyParams& p, bool bf16, hipStream_t) {
    QTTS_REQUIRE(!bf16 && !p.x_bf16 && !p.out_bf16 && !p.out16, QTTS_ERR_ARG, "host emulation runs fp32 engines only");
    const int fs = p.fs ? p.fs : 16;
    QTTS_REQUIRE(fs == 16, QTTS_ERR_ARG, "skinny: narrow strips are only built for the staged bf16 kernel");
    QTTS_REQUIRE(p.N % 16 == 0 && p.K % 16 == 0 && p.M >= 1 && p.M <= 64, QTTS_ERR_ARG, "skinny: shape");
    QTTS_REQUIRE(!p.norm || p.ss_in, QTTS_ERR_ARG, "skinny: norm without LDS staging needs ss_in (row sums of squares)");
    if (p.done_flag && *p.done_flag) return;
    const float* Wp = reinterpret_cast<const float*>(p.Wp);
    std::vector<float> acc(p.N);
    for (int m = 0; m < p.M; ++m) {
        const float rstd = p.norm ? 1.f / sqrtf(p.ss_in[m] / (float)p.K + p.eps) : 1.f;
        for (int n = 0; n < p.N; ++n) {
            float s = 0.f;
            for (int k = 0; k < p.K; ++k) s += p.x[(size_t)m * p.ldx + k] * packed_w(Wp, p.K, fs, n, k);
            acc[n] = s * rstd + (p.bias ? p.bias[n] : 0.f);
        }
        if (p.act == ACT_SWIGLU) {
            for (int n = 0; n < p.N; ++n) {
                if ((n / 16) % 2) continue;
                const int col = (n / 32) * 16 + n % 16;
                float o = (acc[n] / (1.f + expf(-acc[n]))) * acc[n + 16];
                if (p.res) o += p.res[(size_t)m * p.ldr + col];
                p.out[(size_t)m * p.ldo + col] = o;
            }
        } else {
            for (int n = 0; n < p.N; ++n)
                p.out[(size_t)m * p.ldo + n] = acc[n] + (p.res ? p.res[(size_t)m * p.ldr + n] : 0.f);
        }
    }
}

}  // namespace qtts
