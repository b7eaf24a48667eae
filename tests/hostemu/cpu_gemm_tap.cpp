// TEST INFRASTRUCTURE -- the one CPU stand-in left in the host emulation: gemm_tap.hip's launch interface in plain loops
// (fp32 only).  Every kernel source of csrc/ -- gemm_tap.hip included -- is compiled for the SIMT emulator (simt.h); the
// tap GEMM's MFMA emulation is ~15x slower than these loops, so the large engine tests route launch_gemm_tap here by
// default and through the REAL kernel when asked:
//     hostemu_set_real_gemm(1)  /  QTTS_HOSTEMU_FULL=1     -> qtts::launch_gemm_tap_real (gemm_tap.hip, built with
//                                                            -Dlaunch_gemm_tap=launch_gemm_tap_real)
// tests/test_hostemu.py runs the encoder / speaker engines and the kernel-level GEMM cases with the real kernel always,
// and the whole file with it under QTTS_HOSTEMU_FULL=1 (17 min; recorded in DESIGN.md).  Not a product path:
// libqtts_hostemu.so is only loaded by tests/.
#include <cmath>
#include <cstdlib>
#include <vector>
#include "common.h"
#include "kernels.h"

namespace qtts {

void launch_gemm_tap_real(const GemmTapParams& p, bool bf16, hipStream_t st);      // gemm_tap.hip on the emulator
static int g_real_gemm = -1;
static bool real_gemm() {
    if (g_real_gemm < 0) { const char* e = getenv("QTTS_HOSTEMU_FULL"); g_real_gemm = (e && e[0] == '1') ? 1 : 0; }
    return g_real_gemm == 1;
}

// gemm_tap.hip: C[m][n] = epi( sum_tap sum_k A[m + shift[tap]][k] * W[tap][n][k] ), zero row when (m % T) + shift < 0
void launch_gemm_tap(const GemmTapParams& p, bool bf16, hipStream_t st) {
    if (real_gemm()) return launch_gemm_tap_real(p, bf16, st);
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
            p.C[(size_t)m * p.ldc + n] = v;
        }
    }
}

}  // namespace qtts

// ---- test-only entry points (host pointers): kernel-level cases for tests/test_hostemu.py
extern "C" void hostemu_set_real_gemm(int on) { qtts::g_real_gemm = on ? 1 : 0; }

// C[M][ldc] = epilogue(sum_tap A[m + shift[tap]] . W[tap]^T) through the REAL gemm_tap.hip kernels; returns 0 / QTTS_ERR_*
extern "C" int hostemu_gemm_tap(const float* A, int lda, int M, int T, const void* W, int N, int K, int taps, const int* shift,
                                const float* bias, const float* scale, const float* res, int ldr, const float* snake_ea,
                                const float* snake_ib, int act, float* C, int ldc, int bf16) {
    try {
        qtts::GemmTapParams p{};
        p.A = A; p.lda = lda; p.M = M; p.T = T; p.W = W; p.N = N; p.K = K; p.taps = taps;
        for (int i = 0; i < taps && i < 8; ++i) p.shift[i] = shift[i];
        p.bias = bias; p.scale = scale; p.res = res; p.ldr = ldr; p.snake_ea = snake_ea; p.snake_ib = snake_ib; p.act = act;
        p.C = C; p.ldc = ldc;
        qtts::launch_gemm_tap_real(p, bf16 != 0, nullptr);
        return 0;
    } catch (const qtts::Error& e) { return e.code; } catch (...) { return -1; }
}

// out[M][ldo] = skinny GEMM of x[M][K] with W[N][K] (row-major fp32, packed here exactly as the engines pack it)
extern "C" int hostemu_skinny(const float* x, int ldx, int M, const float* W, int N, int K, const float* g, int norm, float eps,
                              const float* bias, const float* res, int ldr, int act, float* out, int ldo, int bf16) {
    try {
        std::vector<unsigned char> wp(qtts::skinny_packed_bytes(N, K, bf16 != 0));
        qtts::pack_skinny_weight(W, N, K, bf16 != 0, wp.data(), g, 16);
        std::vector<float> ss(M, 0.f);
        for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) ss[m] += x[(size_t)m * ldx + k] * x[(size_t)m * ldx + k];
        qtts::SkinnyParams p{};
        p.x = x; p.ldx = ldx; p.M = M; p.Wp = wp.data(); p.N = N; p.K = K; p.fs = 16; p.norm = norm; p.ss_in = ss.data(); p.eps = eps;
        p.bias = bias; p.res = res; p.ldr = ldr; p.out = out; p.ldo = ldo; p.act = act;
        qtts::launch_skinny(p, bf16 != 0, nullptr);
        return 0;
    } catch (const qtts::Error& e) { return e.code; } catch (...) { return -1; }
}
