// TEST INFRASTRUCTURE -- the one CPU stand-in left in the host emulation: gemm_tap.hip's launch interface in plain loops
// (fp32 only).  Every kernel source of csrc/ -- gemm_tap.hip included -- is compiled for the SIMT emulator (simt.h); the
// tap GEMM's MFMA emulation is ~15x slower than these loops, so the large engine tests route launch_gemm_tap here by
// default and through the REAL kernel when asked:
//     hostemu_set_real_gemm(1)  /  QTTS_HOSTEMU_FULL=1     -> qtts::launch_gemm_tap_real (gemm_tap.hip, built with
//                                                            -Dlaunch_gemm_tap=launch_gemm_tap_real)
// (hostemu_set_real_gemm lives in test_entries.cpp.)  tests/test_hostemu.py runs the encoder / speaker engines and the
// kernel-level GEMM cases with the real kernel always,
// and the whole file with it under QTTS_HOSTEMU_FULL=1 (17 min; recorded in DESIGN.md).  Not a product path:
// libqtts_hostemu.so is only loaded by tests/.
#include <cmath>
#include <cstdlib>
#include <vector>
#include "common.h"
#include "kernels.h"

namespace qtts {

void launch_gemm_tap_real(const GemmTapParams& p, bool bf16, hipStream_t st);      // gemm_tap.hip on the emulator
int g_real_gemm = -1;              // set by hostemu_set_real_gemm (test_entries.cpp)
static bool real_gemm() {
    if (g_real_gemm < 0) { const char* e = getenv("QTTS_HOSTEMU_FULL"); g_real_gemm = (e && e[0] == '1') ? 1 : 0; }
    return g_real_gemm == 1;
}

// gemm_tap.hip: C[m][n] = epi( sum_tap sum_k A[m + shift[tap]][k] * W[tap][n][k] ), zero row when (m % T) + shift < 0
static void gemm_tap_loops(const GemmTapParams& p);
void launch_gemm_tap(const GemmTapParams& p, bool bf16, hipStream_t st) {
    if (real_gemm() || bf16) return launch_gemm_tap_real(p, bf16, st);     // bf16 engines: always the real kernels (small test dims only)
    // like a real launch: while a stream capture is open the work becomes a graph node (arguments by value) and runs at replay
    if (simt::capture().open) { const GemmTapParams q = p; simt::capture().open->nodes.push_back([q] { gemm_tap_loops(q); }); return; }
    gemm_tap_loops(p);
}
static void gemm_tap_loops(const GemmTapParams& p) {
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
                float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // 8 independent chains: vectorises (K % 32 == 0)
                for (int k = 0; k < p.K; k += 8)
                    for (int e = 0; e < 8; ++e) part[e] += a[k + e] * wr[k + e];
                acc[n] += ((part[0] + part[4]) + (part[1] + part[5])) + ((part[2] + part[6]) + (part[3] + part[7]));
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
            if (p.C) p.C[(size_t)m * p.ldc + n] = v;
        }
    }
}

}  // namespace qtts
