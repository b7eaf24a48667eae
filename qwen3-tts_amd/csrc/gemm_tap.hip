// gemm_tap.hip -- LDS-tiled MFMA "tap GEMM" for gfx950.
//
//   C[m, n] = epi( sum_{tap} sum_{k} A[row(m, tap), k] * W[tap][n][k] )
//
// with row(m, tap) = m + shift[tap] (shift <= 0) and a zero row whenever the shifted row would
// leave the sequence the output row belongs to ((m % T) + shift < 0): exactly the causal left
// padding of Qwen3TTSTokenizerV2CausalConvNet (tokenizer v2:189-192).  Activations are kept
// CHANNEL-LAST ([rows = batch*time][channels]) so that the contraction dimension is contiguous for
// both MFMA operands; in that layout
//   * nn.Linear                      = 1 tap, shift 0, W = weight (out,in) as stored
//   * causal Conv1d(k, dilation d)   = k taps, shift_j = -(k-1-j)*d, W[j] = weight[:, :, j]
//   * ConvTranspose1d(2r, stride r) then trim r (tokenizer v2:204-208)
//                                    = 2 taps (shift 0 / -1), N = r*Cout (output row m holds the r
//                                      upsampled positions m*r .. m*r+r-1), W[0][p*Cout+co] =
//                                      weight[:, co, p], W[1][p*Cout+co] = weight[:, co, p+r]
// so one kernel covers ~96 % of the codec decoder's FLOPs and all of the talker prefill GEMMs.
//
// Tiling: 128 x BN block tile, BK = 32, 256 threads = 4 waves (2x2), each wave owns 64 x BN/2 as
// 4 x BN/32 MFMA 16x16 tiles.  F32 mode uses v_mfma_f32_16x16x4_f32 (bit-exact fp32 fma chain,
// 157 TF peak), BF16 mode v_mfma_f32_16x16x32_bf16 (A converted fp32->bf16 while staging).
// Global -> registers -> LDS staging with the next tile's loads in flight during the MFMAs.
#include <map>
#include <mutex>
#include <type_traits>
#include "common.h"
#include "kernels.h"
#include "tstamp.h"

QTTS_TS_UNIT(tap)

namespace qtts {

typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {   // v_cvt_pk_bf16_f32 (RNE), == f32_to_bf16 for finite values
    hw_bf16x2 v = {(__bf16)a, (__bf16)b};
    return *reinterpret_cast<unsigned*>(&v);
}

// MFMA roles are swapped w.r.t. the textbook (A operand = the W fragment, B operand = the activation fragment; both have the same
// lane layout, so the swap is free): a lane then holds FOUR CONSECUTIVE output columns of one output row,
//     acc[i][j][r] = C[m0 + wm*64 + i*16 + li][n0 + wn*BN/2 + j*16 + lq*4 + r],
// and the epilogue moves 16-byte vectors: all residual / bias / parameter vectors of a lane are requested first (one latency
// round instead of one dependent round trip per element -- round 1's epilogue was a chain of 64 scalar load -> store pairs per
// lane and dominated the small-channel convolutions), then the activations run, then the stores go out.
// SnakeBeta y = v + ib * sin^2(v * ea).  FAST (bf16 mode): v_sin_f32 (input in revolutions) -- the result is rounded to bf16 or
// carries bf16-level noise anyway, and the library sinf (~60 instructions with its range reduction) would cost more than the
// MFMAs of a small-channel convolution tile; the exact-fp32 parity mode keeps sinf.
template <bool FAST>
__device__ __forceinline__ float snake1(float v, float ea, float ib) {
    float sn;
    if constexpr (FAST) sn = __builtin_amdgcn_sinf(v * ea * 0.15915494309189535f); else sn = sinf(v * ea);
    return v + ib * (sn * sn);
}

template <int BN, int TM, int TN, bool FAST>
__device__ __forceinline__ void tap_epilogue(const GemmTapParams& p, f32x4 (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                             int li, int lq) {
    if (p.act == ACT_SWIGLU) {
        // W rows come in 16-row blocks alternating gate / up for the same 16 features, so tiles (j, j+1) of one wave hold
        // gate / up of the same output columns in the same lane / register.
        if constexpr (TN % 2 == 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m0 + wm * (TM * 16) + i * 16 + li;
#pragma unroll
                for (int j = 0; j < TN; j += 2) {
                    const int n_packed = n0 + wn * (BN / 2) + j * 16 + lq * 4;
                    const int no = (n_packed / 32) * 16 + lq * 4;
                    if (m < p.M && n_packed < p.N) {
                        f32x4 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float g = acc[i][j][r], u = acc[i][j + 1][r]; o[r] = (g / (1.f + expf(-g))) * u; }
                        if (p.C) *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + no) = o;
                        if (p.C16) {     // bf16 copy for a bf16-GEMM consumer (ldc16 % 4 == 0, 8-byte aligned: checked at launch)
                            uint2 hv; hv.x = cvt_pk_bf16(o[0], o[1]); hv.y = cvt_pk_bf16(o[2], o[3]);
                            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C16) + (size_t)m * p.ldc16 + no) = hv;
                        }
                    }
                }
            }
        }
        return;
    }
    if (!p.vec4) {                       // ragged N or unaligned views (e.g. the speaker encoder's 1026-column DFT): element by element
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int m = m0 + wm * (TM * 16) + i * 16 + li;
#pragma unroll 1
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + wn * (BN / 2) + j * 16 + lq * 4 + r;
                    if (m >= p.M || n >= p.N) continue;
                    float v = acc[i][j][r] + (p.bias ? p.bias[n] : 0.f);
                    if (p.act == ACT_GELU) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
                    else if (p.act == ACT_SNAKE) v = snake1<FAST>(v, p.snake_ea[n], p.snake_ib[n]);
                    else if (p.act == ACT_SILU) v = v / (1.f + expf(-v));
                    v *= p.scale ? p.scale[n] : 1.f;
                    if (p.res) v += p.res[(size_t)m * p.ldr + n];
                    else if (p.res16) v += bf16_to_f32(reinterpret_cast<const bf16_t*>(p.res16)[(size_t)m * p.ldres16 + n]);
                    if (p.C) p.C[(size_t)m * p.ldc + n] = v;
                    if (p.R16) reinterpret_cast<bf16_t*>(p.R16)[(size_t)m * p.ldR16 + n] = f32_to_bf16(v);
                    if (p.C16) {
                        if (p.act16 == ACT_SNAKE) { const int n16 = p.snake16_period > 0 ? n % p.snake16_period : n; v = snake1<FAST>(v, p.snake16_ea[n16], p.snake16_ib[n16]); }
                        reinterpret_cast<bf16_t*>(p.C16)[(size_t)m * p.ldc16 + n] = f32_to_bf16(v);
                    }
                }
            }
        return;
    }
    // Everything a column quad needs from memory -- parameters, then the TM residual vectors -- is requested UNCONDITIONALLY and
    // back to back (absent operands read W, rows past M re-read row M - 1; selected afterwards).  With `ptr ? *ptr : 0` every one
    // of these loads was merged with a constant and waited for before the next one was issued: the epilogue of a 1x1 convolution
    // tile spent 17.6 us in 12 serialized residual round trips (profiles/r02_tstamp_codec_gemm.md).
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f}, one4 = {1.f, 1.f, 1.f, 1.f};
    const float* dummy = reinterpret_cast<const float*>(p.W);        // any readable, 16-byte aligned address
    const bool snake = p.act == ACT_SNAKE, snake16 = p.C16 && p.act16 == ACT_SNAKE;
#pragma unroll
    for (int j = 0; j < TN; ++j) {                                   // one column quad of this lane at a time
        const int n = n0 + wn * (BN / 2) + j * 16 + lq * 4;
        if (n >= p.N) continue;                                      // (N % 4 == 0: a quad is in or out as a whole)
        const int n16 = p.snake16_period > 0 ? n % p.snake16_period : n;       // (period % 4 == 0)
        f32x4 bias = *reinterpret_cast<const f32x4*>(p.bias ? p.bias + n : dummy);
        f32x4 scale = *reinterpret_cast<const f32x4*>(p.scale ? p.scale + n : dummy);
        const f32x4 ea = *reinterpret_cast<const f32x4*>(snake ? p.snake_ea + n : dummy);
        const f32x4 ib = *reinterpret_cast<const f32x4*>(snake ? p.snake_ib + n : dummy);
        const f32x4 ea16 = *reinterpret_cast<const f32x4*>(snake16 ? p.snake16_ea + n16 : dummy);
        const f32x4 ib16 = *reinterpret_cast<const f32x4*>(snake16 ? p.snake16_ib + n16 : dummy);
        f32x4 res[TM];
        uint2 rh[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * (TM * 16) + i * 16 + li, mc = m < p.M ? m : p.M - 1;
            res[i] = *reinterpret_cast<const f32x4*>(p.res ? p.res + (size_t)mc * p.ldr + n : dummy);
            rh[i] = *reinterpret_cast<const uint2*>(p.res16 ? reinterpret_cast<const bf16_t*>(p.res16) + (size_t)mc * p.ldres16 + n
                                                            : reinterpret_cast<const bf16_t*>(dummy));
        }
        if (!p.bias) bias = zero4;
        if (!p.scale) scale = one4;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * (TM * 16) + i * 16 + li;
            if (m >= p.M) continue;
            f32x4 r = zero4;
            if (p.res) r = res[i];
            else if (p.res16) r = (f32x4){__uint_as_float(rh[i].x << 16), __uint_as_float(rh[i].x & 0xffff0000u),
                                          __uint_as_float(rh[i].y << 16), __uint_as_float(rh[i].y & 0xffff0000u)};
            f32x4 v = acc[i][j] + bias;
            if (p.act == ACT_GELU) {
#pragma unroll
                for (int r2 = 0; r2 < 4; ++r2) v[r2] = 0.5f * v[r2] * (1.f + erff(v[r2] * 0.70710678118654752440f));
            } else if (p.act == ACT_SNAKE) {
#pragma unroll
                for (int r2 = 0; r2 < 4; ++r2) v[r2] = snake1<FAST>(v[r2], ea[r2], ib[r2]);
            } else if (p.act == ACT_SILU) {
#pragma unroll
                for (int r2 = 0; r2 < 4; ++r2) v[r2] = v[r2] / (1.f + expf(-v[r2]));
            }
            v = v * scale + r;
            if (p.C) *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + n) = v;
            if (p.R16) {               // the residual stream in bf16 (before the consumer's activation)
                uint2 h;
                h.x = cvt_pk_bf16(v[0], v[1]); h.y = cvt_pk_bf16(v[2], v[3]);
                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.R16) + (size_t)m * p.ldR16 + n) = h;
            }
            if (p.C16) {               // bf16 copy for a GEMM consumer, with that consumer's SnakeBeta folded in
                if (p.act16 == ACT_SNAKE) {
#pragma unroll
                    for (int r2 = 0; r2 < 4; ++r2) v[r2] = snake1<FAST>(v[r2], ea16[r2], ib16[r2]);
                }
                uint2 h;
                h.x = cvt_pk_bf16(v[0], v[1]); h.y = cvt_pk_bf16(v[2], v[3]);
                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C16) + (size_t)m * p.ldc16 + n) = h;
            }
        }
    }
}

template <int BN, bool BF16>
__global__ __launch_bounds__(256) void gemm_tap_kernel(GemmTapParams p) {
    constexpr int BM = 128, BK = 32;
    constexpr int TM = 4, TN = BN / 32;
    constexpr int LDS_STRIDE = BF16 ? 40 : 36;  // elements per LDS row (pad keeps 16-B alignment)
    using elem_t = typename std::conditional<BF16, bf16_t, float>::type;
    __shared__ __attribute__((aligned(16))) elem_t As[BM * LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) elem_t Ws[BN * LDS_STRIDE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lq = lane >> 4;

    // XCD-aware tile order: consecutive remapped ids share the A row-panel, and each XCD (bid % 8)
    // works on a contiguous range of tiles so the panel is fetched into ONE L2.
    const int n_tiles_n = (p.N + BN - 1) / BN;
    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (bid / n_tiles_n) * BM;
    const int n0 = (bid % n_tiles_n) * BN;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int ksteps = p.K / BK;
    const int nsteps = ksteps * p.taps;

    // staging registers
    float4 ra[4];
    constexpr int WREG = BF16 ? (BN * 4 + 255) / 256 : (BN * 8 + 255) / 256;
    uint4 rw[WREG];

    const int a_c4 = tid & 7;    // float4 column within the 32-wide k slab
    const int a_r = tid >> 3;    // 0..31 (+32*i)
    int a_t[4];                  // position of the row inside its sequence (for tap validity)
#pragma unroll
    for (int i = 0; i < 4; ++i) a_t[i] = (m0 + a_r + 32 * i) % p.T;

    auto load_tiles = [&](int s) {
        const int tap = s / ksteps;
        const int k0 = (s - tap * ksteps) * BK;
        const int sh = p.shift[tap];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + a_r + 32 * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < p.M && a_t[i] + sh >= 0)
                v = *reinterpret_cast<const float4*>(p.A + (size_t)(m + sh) * p.lda + k0 + a_c4 * 4);
            ra[i] = v;
        }
        if constexpr (BF16) {
            const bf16_t* W = reinterpret_cast<const bf16_t*>(p.W) + (size_t)tap * p.N * p.K;
#pragma unroll
            for (int i = 0; i < WREG; ++i) {
                const int idx = tid + 256 * i;
                const int r = idx >> 2, c = idx & 3;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (r < BN && n0 + r < p.N)
                    v = *reinterpret_cast<const uint4*>(W + (size_t)(n0 + r) * p.K + k0 + c * 8);
                rw[i] = v;
            }
        } else {
            const float* W = reinterpret_cast<const float*>(p.W) + (size_t)tap * p.N * p.K;
#pragma unroll
            for (int i = 0; i < WREG; ++i) {
                const int idx = tid + 256 * i;
                const int r = idx >> 3, c = idx & 7;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (r < BN && n0 + r < p.N)
                    v = *reinterpret_cast<const uint4*>(W + (size_t)(n0 + r) * p.K + k0 + c * 4);
                rw[i] = v;
            }
        }
    };
    auto store_tiles = [&]() {
        if constexpr (BF16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint2 h;
                h.x = cvt_pk_bf16(ra[i].x, ra[i].y);
                h.y = cvt_pk_bf16(ra[i].z, ra[i].w);
                *reinterpret_cast<uint2*>(&As[(a_r + 32 * i) * LDS_STRIDE + a_c4 * 4]) = h;
            }
#pragma unroll
            for (int i = 0; i < WREG; ++i) {
                const int idx = tid + 256 * i;
                const int r = idx >> 2, c = idx & 3;
                if (r < BN) *reinterpret_cast<uint4*>(&Ws[r * LDS_STRIDE + c * 8]) = rw[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<float4*>(&As[(a_r + 32 * i) * LDS_STRIDE + a_c4 * 4]) = ra[i];
#pragma unroll
            for (int i = 0; i < WREG; ++i) {
                const int idx = tid + 256 * i;
                const int r = idx >> 3, c = idx & 7;
                if (r < BN) *reinterpret_cast<uint4*>(&Ws[r * LDS_STRIDE + c * 4]) = rw[i];
            }
        }
    };

    load_tiles(0);
    store_tiles();
    __syncthreads();

    for (int s = 0; s < nsteps; ++s) {
        if (s + 1 < nsteps) load_tiles(s + 1);  // global loads stay in flight under the MFMAs
        if constexpr (BF16) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const bf16x8*>(&As[(wm * 64 + i * 16 + li) * LDS_STRIDE + lq * 8]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j] = *reinterpret_cast<const bf16x8*>(&Ws[(wn * (BN / 2) + j * 16 + li) * LDS_STRIDE + lq * 8]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                f32x4 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[i] = *reinterpret_cast<const f32x4*>(&As[(wm * 64 + i * 16 + li) * LDS_STRIDE + kk * 16 + lq * 4]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[j] = *reinterpret_cast<const f32x4*>(&Ws[(wn * (BN / 2) + j * 16 + li) * LDS_STRIDE + kk * 16 + lq * 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][e], a[i][e], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
        if (s + 1 < nsteps) {
            store_tiles();
            __syncthreads();
        }
    }

    tap_epilogue<BN, TM, TN, BF16>(p, acc, m0, n0, wm, wn, li, lq);
}

// ---- wide-K variant for SMALL grids (talker prefill, text projection, the codec's transformer): bf16, 1 tap, BK = 128.
// With fewer workgroups than CUs nothing hides the global-load latency of a k-step (~0.9 us measured per step with
// BK = 32), so the step is made 4x deeper instead: 4x fewer exposed round trips, 16 + 8 16-B loads in flight per
// thread.  BM x BN tile, 4 waves (2x2), LDS rows are 128 + 8 bf16.
// Round 3: what bounds these launches is what ONE CU can pull -- a k-step of a 128 x 128 tile moves 96 KB (64 KB of it the fp32
// activation tile) in ~2 us = 0.53 us + bytes / 65 GB/s, whatever the prefetch depth -- so the time of a launch is
// (bytes all workgroups pull) / (CUs with a workgroup x 65 GB/s).  Hence
//   * BM = 64 tiles where the 128-row grid leaves CUs idle (launch_gemm_tap picks the tile by that estimate), and
//   * A16: the activation tile read as bf16 when the producer left a bf16 copy (same rounding as converting here, half the bytes).
template <int BM, int BN, bool A16, int BK>
__global__ __launch_bounds__(256) void gemm_wide_kernel(GemmTapParams p) {
    constexpr int TM = BM / 32, TN = BN / 32;
    constexpr int STR = BK + 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_gw[];
    bf16_t* As = reinterpret_cast<bf16_t*>(smem_gw);
    bf16_t* Ws = As + BM * STR;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lq = lane >> 4;
    const int n_tiles_n = (p.N + BN - 1) / BN;
    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (bid / n_tiles_n) * BM;
    const int n0 = (bid % n_tiles_n) * BN;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nsteps = p.K / BK;
    constexpr int CH = BK / 8;                        // 16-B units of bf16 per row and k-step; fp32 rows have 2 CH
    constexpr int AREG = A16 ? BM * CH / 256 : BM * 2 * CH / 256;
    constexpr int WREG = BN * CH / 256;
    float4 ra[A16 ? 1 : AREG];
    uint4 ra16[A16 ? AREG : 1];
    uint4 rw[WREG];
    constexpr int AROWS = 256 / (2 * CH), WROWS = 256 / CH;
    const int a_c4 = tid % (2 * CH), a_r = tid / (2 * CH);      // fp32 A: rows a_r + AROWS * i
    const int w_c = tid % CH, w_r = tid / CH;                   // W (and bf16 A): rows w_r + WROWS * i
    const bf16_t* W = reinterpret_cast<const bf16_t*>(p.W);
    const bf16_t* Ah = reinterpret_cast<const bf16_t*>(p.A16);

    auto load_tiles = [&](int s) {
        const int k0 = s * BK;
        if constexpr (A16) {
#pragma unroll
            for (int i = 0; i < AREG; ++i) {
                const int m = m0 + w_r + WROWS * i;
                ra16[i] = m < p.M ? *reinterpret_cast<const uint4*>(Ah + (size_t)m * p.lda + k0 + w_c * 8) : make_uint4(0, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < AREG; ++i) {
                const int m = m0 + a_r + AROWS * i;
                ra[i] = m < p.M ? *reinterpret_cast<const float4*>(p.A + (size_t)m * p.lda + k0 + a_c4 * 4)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int i = 0; i < WREG; ++i) {
            const int r = w_r + WROWS * i;
            rw[i] = (n0 + r < p.N) ? *reinterpret_cast<const uint4*>(W + (size_t)(n0 + r) * p.K + k0 + w_c * 8)
                                   : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tiles = [&]() {
        if constexpr (A16) {
#pragma unroll
            for (int i = 0; i < AREG; ++i) *reinterpret_cast<uint4*>(&As[(w_r + WROWS * i) * STR + w_c * 8]) = ra16[i];
        } else {
#pragma unroll
            for (int i = 0; i < AREG; ++i) {
                uint2 h;
                h.x = cvt_pk_bf16(ra[i].x, ra[i].y);
                h.y = cvt_pk_bf16(ra[i].z, ra[i].w);
                *reinterpret_cast<uint2*>(&As[(a_r + AROWS * i) * STR + a_c4 * 4]) = h;
            }
        }
#pragma unroll
        for (int i = 0; i < WREG; ++i) *reinterpret_cast<uint4*>(&Ws[(w_r + WROWS * i) * STR + w_c * 8]) = rw[i];
    };

    load_tiles(0);
    store_tiles();
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        if (s + 1 < nsteps) load_tiles(s + 1);
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const bf16x8*>(&As[(wm * (BM / 2) + i * 16 + li) * STR + kk * 32 + lq * 8]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j] = *reinterpret_cast<const bf16x8*>(&Ws[(wn * (BN / 2) + j * 16 + li) * STR + kk * 32 + lq * 8]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (s + 1 < nsteps) {
            store_tiles();
            __syncthreads();
        }
    }
    tap_epilogue<BN, TM, TN, true>(p, acc, m0, n0, wm, wn, li, lq);
}

// ---- round 2: the codec decoder's GEMM in bf16 mode (gemm_tap2).  What the round-1 kernel above spent its time on
// (profiles/r01_pmc_mfma_codec.md: 8.5 % / 16.9 % MFMA-busy, 1.5 GB of fp32 activations moved for 248 GFLOP):
//   * a 7-tap causal conv re-read its input tile 7 times from global memory, converted fp32 -> bf16 every time, through
//     VGPRs into a single LDS buffer with two barriers per 32-wide k-step;
//   * SnakeBeta ran as its own full read + write pass in front of every conv.
// Here:
//   * TAP REUSE: per 32-wide k-slab the input tile is staged ONCE, with its causal halo (128 + max|shift| rows),
//     and all taps of that slab run from LDS with a row offset; a row that would come from before the start of its sequence is
//     zeroed in the operand registers (a tile may span two sequences, so this cannot be decided at staging time);
//   * bf16 activations in HBM wherever a tensor is only a GEMM input (A16 / C16): half the bytes, no conversion on the way in;
//   * two LDS buffers for both operands: the next step's global loads are in flight under this step's MFMAs and land in the
//     other buffer -- one barrier per step;
//   * the NEXT consumer's SnakeBeta is folded into the epilogue of the producer (act16): the residual stream stays fp32 (C),
//     the activated copy goes out as bf16 (C16), and the stand-alone snake passes disappear from the decoder blocks.
// Tile 128 x BN, 4 waves (2 x 2), each wave 64 x BN/2 as 4 x BN/32 MFMA 16x16x32 tiles per 32 of k.
template <int BN, int BK>
__global__ __launch_bounds__(256, BK == 32 ? 3 : 2) void gemm_tap2_kernel(GemmTapParams p, int halo, int cap /* rows of one A buffer: 128 + halo, rounded up */) {
    constexpr int BM = 128;
    constexpr int TM = 4, TN = BN / 32;
    constexpr int STR = BK + 8;                        // LDS row stride (bf16 elements): 16-B aligned, rows shift by 4 banks
    constexpr int CPR = BK / 8;                        // 16-B chunks per row
    constexpr int MAXROWS = BM + 56;                   // 7 taps x dilation 9 -> halo 54
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_t2[];
    // The staged input tile is needed in TWO buffers only when every step changes the k-slab (1 tap).  With several taps a slab
    // lives for `taps` steps: one buffer and one extra barrier per slab change, and the smaller LDS footprint keeps one more
    // workgroup per CU resident (C = 96, halo 54: 44.8 -> 30.1 KB) -- occupancy is what hides this kernel's latency.
    const int abufs = p.taps > 1 ? 1 : 2;
    bf16_t* As = reinterpret_cast<bf16_t*>(smem_t2);                       // [abufs][cap][STR]  (LDS is sized for the actual halo)
    bf16_t* Ws = As + abufs * cap * STR;                                   // [2][BN][STR]
    QTTS_TS_BEGIN();                       // (tstamp build: 1 = prologue done, 2 / 3 = after step 1 / step 1 + 2 taps, 4 = k-loop done, 5 = stored)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lq = lane >> 4;
    const int n_tiles_n = (p.N + BN - 1) / BN;
    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    {   // XCD-aware tile order (as gemm_tap_kernel)
        const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (bid / n_tiles_n) * BM;
    const int n0 = (bid % n_tiles_n) * BN;
    const int arows = BM + halo;                       // rows of the staged input tile: global rows m0 - halo .. m0 + 127
    const bf16_t* A16 = reinterpret_cast<const bf16_t*>(p.A16);
    const bf16_t* Wg = reinterpret_cast<const bf16_t*>(p.W);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int tpos[TM];                                      // position of this lane's output rows inside their sequence
#pragma unroll
    for (int i = 0; i < TM; ++i) tpos[i] = (m0 + wm * 64 + i * 16 + li) % p.T;

    const int kslabs = p.K / BK;
    const int nsteps = kslabs * p.taps;
    constexpr int AREG = (MAXROWS * CPR + 255) / 256;
    constexpr int WREG = (BN * CPR + 255) / 256;
    uint4 ra[AREG], rw[WREG];

    auto load_a = [&](int ks) {
#pragma unroll
        for (int i = 0; i < AREG; ++i) {
            const int idx = tid + 256 * i;
            const int r = idx / CPR, c = idx - r * CPR;
            const int gr = m0 - halo + r;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (r < arows && gr >= 0 && gr < p.M) v = *reinterpret_cast<const uint4*>(A16 + (size_t)gr * p.lda + ks * BK + c * 8);
            ra[i] = v;
        }
    };
    auto store_a = [&](int buf) {
        bf16_t* dst = As + (abufs == 2 ? buf : 0) * cap * STR;
#pragma unroll
        for (int i = 0; i < AREG; ++i) {
            const int idx = tid + 256 * i;
            const int r = idx / CPR, c = idx - r * CPR;
            if (r < arows) *reinterpret_cast<uint4*>(&dst[r * STR + c * 8]) = ra[i];
        }
    };
    auto load_w = [&](int ks, int tap) {
        const bf16_t* W = Wg + (size_t)tap * p.N * p.K;
#pragma unroll
        for (int i = 0; i < WREG; ++i) {
            const int idx = tid + 256 * i;
            const int r = idx / CPR, c = idx - r * CPR;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (r < BN && n0 + r < p.N) v = *reinterpret_cast<const uint4*>(W + (size_t)(n0 + r) * p.K + ks * BK + c * 8);
            rw[i] = v;
        }
    };
    auto store_w = [&](int buf) {
        bf16_t* dst = Ws + buf * BN * STR;
#pragma unroll
        for (int i = 0; i < WREG; ++i) {
            const int idx = tid + 256 * i;
            const int r = idx / CPR, c = idx - r * CPR;
            if (r < BN) *reinterpret_cast<uint4*>(&dst[r * STR + c * 8]) = rw[i];
        }
    };

    load_a(0); load_w(0, 0);
    store_a(0); store_w(0);
    __syncthreads();
    QTTS_TS(1);
    for (int s = 0; s < nsteps; ++s) {
        const int ks = s / p.taps, tap = s - ks * p.taps;
        const bool more = s + 1 < nsteps;
        const int ks2 = (s + 1) / p.taps, tap2 = (s + 1) - ks2 * p.taps;
        const bool new_slab = more && ks2 != ks;
        if (more) load_w(ks2, tap2);                   // global loads stay in flight under the MFMAs
        if (new_slab) load_a(ks2);
        {
            const bf16_t* Ab = As + (abufs == 2 ? (ks & 1) : 0) * cap * STR;
            const bf16_t* Wb = Ws + (s & 1) * BN * STR;
            const int sh = p.shift[tap];               // <= 0: output row m reads staged row (m - m0) + halo + sh
#pragma unroll
            for (int kk = 0; kk < BK / 32; ++kk) {
                bf16x8 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    a[i] = *reinterpret_cast<const bf16x8*>(&Ab[(wm * 64 + i * 16 + li + halo + sh) * STR + kk * 32 + lq * 8]);
                    if (tpos[i] + sh < 0) a[i] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};      // before the start of its own sequence
                }
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[j] = *reinterpret_cast<const bf16x8*>(&Wb[(wn * (BN / 2) + j * 16 + li) * STR + kk * 32 + lq * 8]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
            }
        }
        if (more) {                                    // the other buffers were last read one step (W) / one slab (A) ago
            store_w((s + 1) & 1);
            if (new_slab) {
                if (abufs == 1) __syncthreads();       // every wave is done with the slab that is about to be overwritten
                store_a(ks2 & 1);
            }
        }
        __syncthreads();
#if QTTS_TSTAMP
        if (s == 0) QTTS_TS(2);
        if (s == 2 * p.taps) QTTS_TS(3);
#endif
    }
    QTTS_TS(4);
    tap_epilogue<BN, TM, TN, true>(p, acc, m0, n0, wm, wn, li, lq);
    QTTS_TS_DRAINED(5);
    QTTS_TS_END(tap, 4, p.K * p.taps, p.N);
}

template <int BN, int BK>
static void launch_tap2(const GemmTapParams& p, int halo, hipStream_t st) {
    const int nb = cdiv(p.M, 128) * cdiv(p.N, BN);
    const int cap = (128 + halo + 7) & ~7;
    const size_t lds = ((size_t)(p.taps > 1 ? 1 : 2) * cap + 2 * BN) * (BK + 8) * 2;
    auto kern = gemm_tap2_kernel<BN, BK>;
    ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, st, p, halo, cap);
}

// ---- round 4: gemm_dma -- the bf16-activation tap GEMM with BOTH operands staged by LDS-DMA (global_load_lds, 16 B per lane).
// gemm_tap2 / gemm_wide move every tile global -> VGPRs -> ds_write -> LDS: the staging registers, the write pass and (gemm_wide) a
// second barrier per k-step are what the guide's "128 x 128, two barriers per step" structure spends its time on (~25 % MFMA-busy
// on the codec's C = 768 / 384 units and on the batch-32 prefill, profiles/r03_pmc_mfma_codec.md).  Here a k-step is
//     __syncthreads()            -- step s's tiles have landed (the barrier's fence drains the DMA queue), step s - 1 is consumed
//     request step s + 1's tiles -- W [BN][64] of (slab, tap), and the A tile [128 + halo][64] when the slab changes
//     32 MFMAs per wave from the tiles of step s
// -- one barrier, no staging registers, the next step's bytes in flight under this step's MFMAs.  LDS rows keep gemm_tap2's
// 144-byte stride (64 bf16 + 16 B: consecutive rows shift by 4 banks); an LDS-DMA writes lane-linear (wave-uniform base + 16 lane),
// so the padded image is produced by giving every lane the SOURCE address of the LDS byte it fills (resunit.hip's scheme): byte o
// of a buffer is (row, col) = divmod(o, 144), pad bytes and rows past the tile re-fetch a valid element that nobody reads.
// Sequence-start zeroing, tap offsets, epilogue: exactly gemm_tap2's.
__device__ __forceinline__ void gd_dma16(const void* src, void* lds_dst) {       // 64 lanes x 16 B -> 1 KiB of LDS, lane-linear
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
// The barrier alone does not order another wave's reads behind this wave's LDS-DMA: the data is in LDS when the REQUESTING wave's vmcnt
// says so.  The compiler's barrier fence emits the wait today (tools/isa_waits.py dma_barriers pins that for every kernel that uses
// global_load_lds); the kernel whose loop depends on it says it itself.
__device__ __forceinline__ void gd_dma_wait() {
#ifndef QTTS_HOST_EMU
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
template <int BN>
__global__ __launch_bounds__(256, 2) void gemm_dma_kernel(GemmTapParams p, int halo, int a_stride /* bytes of one A buffer, multiple of 1024 */) {
    constexpr int BM = 128, BK = 64;
    constexpr int TM = 4, TN = BN / 32;
    constexpr int STR = BK + 8, RS = STR * 2;          // LDS row: 72 bf16 = 144 B
    constexpr int W_BYTES = (BN * RS + 1023) / 1024 * 1024;
    constexpr int NWC = W_BYTES / 1024;                // 1-KiB chunks of a W buffer
    constexpr int WPW = (NWC + 3) / 4;                 // ... per wave
    constexpr int APW = ((BM + 56) * RS / 1024 + 1 + 3) / 4;      // A chunks per wave at the largest halo
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_gd[];
    unsigned char* Abuf = smem_gd;                     // [2][a_stride]
    unsigned char* Wbuf = smem_gd + 2 * a_stride;      // [2][W_BYTES]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lq = lane >> 4;
    const int n_tiles_n = (p.N + BN - 1) / BN;
    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    {   // XCD-aware tile order (as gemm_tap_kernel)
        const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (bid / n_tiles_n) * BM;
    const int n0 = (bid % n_tiles_n) * BN;
    const int arows = BM + halo;
    const int nac = (arows * RS + 1023) / 1024;        // chunks of an A buffer actually filled
    const unsigned char* A16 = reinterpret_cast<const unsigned char*>(p.A16);
    const unsigned char* Wg = reinterpret_cast<const unsigned char*>(p.W);

    // per-lane source offsets (bytes, without the k-slab / tap terms) of the chunks this wave requests: chunk c = wave + 4 i
    size_t a_off[APW], w_off[WPW];
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        const int o = (wave + 4 * i) * 1024 + lane * 16;
        int row = o / RS;
        const int col = o - row * RS;
        row = row < arows ? row : arows - 1;
        int gr = m0 - halo + row;
        gr = gr < 0 ? 0 : (gr >= p.M ? p.M - 1 : gr);
        a_off[i] = (size_t)gr * p.lda * 2 + (col < BK * 2 ? col : 0);
    }
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const int o = (wave + 4 * i) * 1024 + lane * 16;
        int row = o / RS;
        const int col = o - row * RS;
        row = row < BN ? row : BN - 1;
        int gn = n0 + row;
        gn = gn < p.N ? gn : p.N - 1;
        w_off[i] = (size_t)gn * p.K * 2 + (col < BK * 2 ? col : 0);
    }
    auto dma_a = [&](int ks, int buf) {
#pragma unroll
        for (int i = 0; i < APW; ++i)
            if (wave + 4 * i < nac) gd_dma16(A16 + a_off[i] + (size_t)ks * (BK * 2), Abuf + buf * a_stride + (wave + 4 * i) * 1024);
    };
    auto dma_w = [&](int ks, int tap, int buf) {
        const unsigned char* W = Wg + ((size_t)tap * p.N * p.K + (size_t)ks * BK) * 2;
#pragma unroll
        for (int i = 0; i < WPW; ++i)
            if (wave + 4 * i < NWC) gd_dma16(W + w_off[i], Wbuf + buf * W_BYTES + (wave + 4 * i) * 1024);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int tpos[TM];                                      // position of this lane's output rows inside their sequence
#pragma unroll
    for (int i = 0; i < TM; ++i) tpos[i] = (m0 + wm * 64 + i * 16 + li) % p.T;

    const int kslabs = p.K / BK;
    const int nsteps = kslabs * p.taps;
    dma_a(0, 0);
    dma_w(0, 0, 0);
    for (int s = 0; s < nsteps; ++s) {
        const int ks = s / p.taps, tap = s - ks * p.taps;
        gd_dma_wait();                                 // this wave's DMA requests of step s have landed (explicit: ADVICE r4) ...
        __syncthreads();                               // ... and every other wave's: step s's tiles are in LDS; every wave is done with step s - 1's
        if (s + 1 < nsteps) {
            const int ks2 = (s + 1) / p.taps, tap2 = (s + 1) - ks2 * p.taps;
            dma_w(ks2, tap2, (s + 1) & 1);             // (last read in step s - 1)
            if (ks2 != ks) dma_a(ks2, ks2 & 1);        // (last read in the last step of slab ks - 1)
        }
        const bf16_t* Ab = reinterpret_cast<const bf16_t*>(Abuf + (ks & 1) * a_stride);
        const bf16_t* Wb = reinterpret_cast<const bf16_t*>(Wbuf + (s & 1) * W_BYTES);
        const int sh = p.shift[tap];                   // <= 0: output row m reads staged row (m - m0) + halo + sh
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                a[i] = *reinterpret_cast<const bf16x8*>(&Ab[(wm * 64 + i * 16 + li + halo + sh) * STR + kk * 32 + lq * 8]);
                if (tpos[i] + sh < 0) a[i] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};      // before the start of its own sequence
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j] = *reinterpret_cast<const bf16x8*>(&Wb[(wn * (BN / 2) + j * 16 + li) * STR + kk * 32 + lq * 8]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
    }
    tap_epilogue<BN, TM, TN, true>(p, acc, m0, n0, wm, wn, li, lq);
}

template <int BN>
static void launch_dma(const GemmTapParams& p, int halo, hipStream_t st) {
    const int nb = cdiv(p.M, 128) * cdiv(p.N, BN);
    const int a_stride = ((128 + halo) * 144 + 1023) / 1024 * 1024;
    const size_t lds = 2 * (size_t)a_stride + 2 * (size_t)((BN * 144 + 1023) / 1024 * 1024);
    auto kern = gemm_dma_kernel<BN>;
    ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, st, p, halo, a_stride);
}

// ---- round 6: gemm_ring -- gemm_dma's arithmetic (same MFMA sequence per accumulator: results are BIT-IDENTICAL to gemm_dma_kernel's) with the
// three things profiles/r06_gemm_ring.md measured against it changed:
//   (1) LDS image without bank conflicts.  gemm_dma's 144-byte rows make every ds_read_b128 a 2-way conflict (SQ_LDS_BANK_CONFLICT = exactly half of
//       SQ_LDS_IDX_ACTIVE on every launch): the b128 lane groups of gfx950 are {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... -- a group reads k-chunk c
//       of 8 rows and chunk c ^ 1 of the 8 rows between them, and for the 16 rows of a fragment every row difference occurs, so no row padding is
//       conflict-free.  Here a tile row is 32 k = 64 bytes (four 16-byte chunks), rows are dense, and chunk c of row r sits at chunk position
//       c ^ (2 * ((r >> 2) & 1)): of the four rows of a group that share r % 4 (= the same quarter of a 256-byte bank row), the first and the last read
//       chunk c, the two between them chunk c ^ 1, and consecutive row quads alternate the XOR -- four distinct positions, WHATEVER the row offset
//       (the tap shift).  A wave-wide LDS-DMA fills 16 rows (1 KiB, lane l = row l / 4, position l % 4): every lane quad fetches one contiguous
//       64-byte segment.  (The first build stored [16-row band][chunk][row]: conflict-free too, but consecutive lanes then fetch consecutive ROWS --
//       64 cache lines per request -- and every shape but dilation 9 lost 15-25 % to gemm_dma.)  No padding: the dilation-9 launches (90 KB, one
//       workgroup per CU) fit two per CU again.
//   (2) a ring of NST weight tiles of 32 k (8 KiB each), requested D steps ahead, with COUNTED vmcnt: at the barrier of step t every wave has waited
//       for its own requests of the tiles <= t + 1 only; the requests of t + 2 .. t + D - 1 stay in flight across the barrier (gemm_dma: one step
//       deep, vmcnt(0) at every barrier -- a grid of <= 256 tiles spends ~1 us per 64-k step waiting for it).
//   (3) operand fragments double-buffered in registers: step t + 1's ds_reads are issued before step t's 16 MFMAs (tile t + 1 has landed by barrier t),
//       so the MFMAs of a step start right behind its barrier and tile t's buffer is free for tile t + D at barrier t (NST = D buffers suffice).
// Step order: (k-slab of 32 * AH, tap, half h < AH) with the weight tile (tap, k-slab, h) per step; the A slab (AH halves of [bands][1 KiB], with its causal
// halo) is staged once per slab and reused by every tap, two buffers, requested a whole slab ahead.  AH = 2 reproduces gemm_dma's 64-wide slabs
// (identical summation order); a plain Linear (taps == 1) runs AH = 1 with the A tile in the ring beside W.
// Ordering rules (the guide's: LDS-DMA data is visible to another wave's ds_read only behind the REQUESTING wave's vmcnt wait followed by a barrier the
// reader has passed): a tile is read one barrier after the wait that retires it (wait + barrier t, reads during step t for step t + 1); a buffer is
// re-requested behind the barrier that follows the lgkmcnt(0) of its last reads.  tests/test_host_logic.py pins the counted waits from the ISA.
__device__ __forceinline__ void ring_wait_vm(int n) {      // s_waitcnt vmcnt(n) takes an immediate; n is wave-uniform
#ifndef QTTS_HOST_EMU
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;     // (never less strict than asked)
    }
#else
    (void)n;
#endif
}
__device__ __forceinline__ void ring_barrier() {           // the reads of the step before are in registers, then the workgroup meets: no fence, no vmcnt drain
#ifndef QTTS_HOST_EMU
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#else
    __syncthreads();
#endif
}
#ifndef QTTS_RING_SGB_ALT
#define QTTS_RING_SGB_ALT 0
#endif
// "1 MFMA, then up to QTTS_RING_SGB others": 3 in the product (QTTS_RING_SGB_ALT 0); build variants: ALT 1 = no hints, 2 / 4 = that many
#define QTTS_RING_SGB (QTTS_RING_SGB_ALT == 0 ? 3 : (QTTS_RING_SGB_ALT == 1 ? 0 : QTTS_RING_SGB_ALT))
__device__ __forceinline__ void ring_sched_fence() {
#ifndef QTTS_HOST_EMU
    __builtin_amdgcn_sched_barrier(0);
#endif
}
template <int N> __device__ __forceinline__ void ring_wait_vm_c() {       // the steady-state wait: an immediate
#ifndef QTTS_HOST_EMU
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
// ABL != 0: measuring variants (tools/bench_gemm_ring.py --ablate; results are WRONG by construction): 1 = no requests in the steady steps, 2 = no
// fragment reads, 4 = no MFMAs, 8 = no barrier, 16 = no address arithmetic for the A fragments (tap shift ignored)
// TAPS = 7 | 2: the steady state runs whole slabs as straight-line code (`ustep`, version 4); 0: any tap count / the plain Linear (`fstep`)
template <int NST, int AH, bool RING_A, int TAPS = 0, int ABL = 0>
__global__ __launch_bounds__(256, 2) void gemm_ring_kernel(GemmTapParams p, int halo, int nbands /* 16-row bands of an A half-slab: ceil((128 + halo) / 16) */,
                                                                 int ks /* K splits per tile (1 = none) */, f32x4* ks_ws, unsigned* ks_cnt) {
    constexpr int BM = 128, BN = 128, TM = 4, TN = 4;
    constexpr int WT = 8192;                           // a weight tile: 128 rows x 32 k = 8 bands of 1 KiB
    constexpr int D = NST;                             // tiles requested ahead (launcher: NST <= steps per slab, or taps == 1)
    constexpr int LPS = RING_A ? 4 : 2;                // requests per wave and step that the counted waits rely on (the slab requests of taps > 1 are extra: over-waited, never under)
    static_assert(!RING_A || AH == 1, "a plain Linear runs 32-wide slabs");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_gr[];
    unsigned char* Wbuf = smem_gr;                     // [NST][WT]
    unsigned char* Abuf = smem_gr + NST * WT;          // taps > 1: [2][AH][nbands KiB]; taps == 1: [NST][WT]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lq = lane >> 4;
    const int n_tiles_n = (p.N + BN - 1) / BN;
    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    {   // XCD-aware tile order (as gemm_tap_kernel)
        const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // split-K (launcher: grids that leave CUs idle): the ks workgroups of a tile are neighbours in the XCD-aware order (same XCD: its L2 holds the partials)
    const int tile = bid / ks, split = bid - tile * ks;
    const int m0 = (tile / n_tiles_n) * BM;
    const int n0 = (tile % n_tiles_n) * BN;
    const int a_half = nbands * 1024, a_buf = AH * a_half;
    const unsigned char* A16 = reinterpret_cast<const unsigned char*>(p.A16);
    const unsigned char* Wg = reinterpret_cast<const unsigned char*>(p.W);

    // per-lane source offsets (bytes, without the k / tap terms) of the 16-row bands this wave requests: band = wave + 4 i; lane l fills position l % 4 of
    // row l / 4 of the band = chunk (l % 4) ^ (2 * ((row >> 2) & 1)) of that row (bands start at multiples of 16 rows: the row's bit 2 is the lane's bit 4)
    const int d_row = lane >> 2, d_chunk = (lane & 3) ^ ((lane >> 3) & 2);
    unsigned w_off[2], a_off[3];                       // (launcher: both operands < 4 GiB)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int gn = n0 + (wave + 4 * i) * 16 + d_row;
        gn = gn < p.N ? gn : p.N - 1;
        w_off[i] = ((unsigned)gn * (unsigned)p.K + d_chunk * 8) * 2u;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        int gr = m0 - halo + (wave + 4 * i) * 16 + d_row;
        gr = gr < 0 ? 0 : (gr >= p.M ? p.M - 1 : gr);
        a_off[i] = ((unsigned)gr * (unsigned)p.lda + d_chunk * 8) * 2u;
    }
    const int kslabs = p.K / (32 * AH) / ks;           // slabs of this workgroup (launcher: divisible)
    const unsigned k0 = (unsigned)split * (unsigned)kslabs * (AH * 64u);     // ... starting at this byte offset of a row
    const int S = p.taps * AH;                         // steps per slab
    const int nsteps = kslabs * S;
    // every cursor below advances by additions only: a step has 16 MFMAs per wave (256 clocks of the matrix pipe) and the scalar unit shares the issue slots
    const unsigned w_tap_bytes = (unsigned)p.N * (unsigned)p.K * 2u;   // (all byte offsets below are 32-bit: launcher)
    int sl_ks = 2;                                     // taps > 1: the next slab to request (0 and 1 go out in the prologue)
    unsigned sl_k = k0 + 2u * AH * 64u;                // ... its k offset in bytes
    auto req_a_slab = [&](unsigned kbytes, int buf) {    // taps > 1: AH halves of [nbands] bands
#pragma unroll
        for (int h = 0; h < AH; ++h)
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (wave + 4 * i < nbands) gd_dma16(A16 + (a_off[i] + kbytes + h * 64u), Abuf + buf * a_buf + h * a_half + (wave + 4 * i) * 1024);
    };
    // request cursor: tile rt = the next weight tile to request = (k offset r_k, tap r_tap) -> stage r_stage
    int rt = 0, r_tap = 0, r_h = 0, r_stage = 0;
    unsigned r_k = k0, r_wtap = 0;                     // bytes: 64 per half-slab; tap * N * K * 2
    auto request_next = [&]() {
        if constexpr (RING_A) {                        // the 128 x 32 A tile of the step, 8 bands, beside its weight tile
#pragma unroll
            for (int i = 0; i < 2; ++i) gd_dma16(A16 + (a_off[i] + r_k), Abuf + r_stage * WT + (wave + 4 * i) * 1024);
        }
        const unsigned wk = r_wtap + r_k;
#pragma unroll
        for (int i = 0; i < 2; ++i) gd_dma16(Wg + (w_off[i] + wk), Wbuf + r_stage * WT + (wave + 4 * i) * 1024);
        ++rt;
        r_stage = r_stage + 1 == NST ? 0 : r_stage + 1;
        // order: (slab, tap, half): the half moves k by 64 bytes, the tap rewinds the slab's halves, the slab keeps the advance
        if constexpr (RING_A) r_k += 64u;              // (one tap, 32-wide slabs: the tile index is the slab)
        else if (++r_h == AH) {
            r_h = 0;
            if (++r_tap == p.taps) { r_tap = 0; r_wtap = 0; r_k += 64u; } else { r_wtap += w_tap_bytes; r_k -= (AH - 1) * 64u; }
        } else r_k += 64u;
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int tpos[TM], arow[TM];                            // position of this lane's output rows inside their sequence; their staged row at shift 0
    bool near_start = false;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        tpos[i] = (m0 + wm * 64 + i * 16 + li) % p.T; arow[i] = wm * 64 + i * 16 + li + halo;
        near_start |= tpos[i] < halo;
    }
    // only a wave that holds rows within `halo` of a sequence start ever zeroes an operand fragment (one tile in T / 128): wave-uniform
    const bool zeroing = __ballot(near_start) != 0;
    const int wfrag = (wn * 64 + li) * 64 + ((lq ^ ((li >> 1) & 2)) << 4);      // + j KiB + stage * WT

    // the taps' shifts (<= 0, reach <= 56 rows) packed into one scalar pair: a scalar load from the argument block inside the loop shares lgkmcnt with the
    // ds_reads and would drain them before the MFMAs (seen in the ISA of the first build)
    unsigned long long shifts = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) shifts |= (unsigned long long)(unsigned)(i < p.taps ? -p.shift[i] : 0) << (8 * i);
    // read cursor: the tile whose fragments are read next: A buffer f_buf (taps > 1), half f_h, tap f_tap (its shift = byte f_tap of `shifts`), stage f_stage
    int f_tap = 0, f_h = 0, f_stage = 0, f_buf = 0;
    unsigned long long f_shifts = shifts;
    auto read_frags = [&](bf16x8 (&a)[TM], bf16x8 (&b)[TN], int& sh_out) {
        const unsigned char* Ab = RING_A ? Abuf + f_stage * WT : Abuf + f_buf * a_buf + f_h * a_half;
        const unsigned char* Wb = Wbuf + f_stage * WT + wfrag;
        const int sh = -(int)(f_shifts & 0xffu);       // <= 0
        sh_out = sh;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int R = arow[i] + sh;
            a[i] = *reinterpret_cast<const bf16x8*>(Ab + (R << 6) + ((lq ^ ((R >> 1) & 2)) << 4));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const bf16x8*>(Wb + j * 1024);
        f_stage = f_stage + 1 == NST ? 0 : f_stage + 1;
        if (++f_h == AH) {
            f_h = 0;
            if (++f_tap == p.taps) { f_tap = 0; f_shifts = shifts; f_buf ^= 1; } else f_shifts >>= 8;
        }
    };

    // prologue: slabs 0 and 1 (taps > 1), then the first D tiles of the ring
    if constexpr (!RING_A) { req_a_slab(k0, 0); if (kslabs > 1) req_a_slab(k0 + AH * 64u, 1); }
    for (int t = 0; t < D && t < nsteps; ++t) request_next();
    {   // tile 0 (and everything older: the slabs) has landed once at most the tiles 1 .. min(D, nsteps) - 1 are outstanding
        const int later = (D < nsteps ? D : nsteps) - 1;
        ring_wait_vm(LPS * later);
        ring_barrier();
    }
    bf16x8 fa[2][TM], fb[2][TN];
    int fsh[2];
    read_frags(fa[0], fb[0], fsh[0]);
    const int t_steady = nsteps - D;                   // steps t < t_steady have all of t + 2 .. t + D - 1 outstanding at their wait
    int sl_cnt = 0;                                    // steps of the current slab behind us (taps > 1)
    auto step = [&](auto steady, int t, bf16x8 (&ca)[TM], bf16x8 (&cb)[TN], int csh, bf16x8 (&na)[TM], bf16x8 (&nb)[TN], int& nsh) {
        // own requests of the tiles <= t + 1 are complete when only those of t + 2 .. min(t + D, nsteps) - 1 are outstanding
        if constexpr (decltype(steady)::value) ring_wait_vm_c<LPS * (D - 2)>();
        else { int later = nsteps - 2 - t; later = later < D - 2 ? later : D - 2; later = later > 0 ? later : 0; ring_wait_vm(LPS * later); }
        ring_barrier();                                // tiles <= t + 1 are in LDS for every wave; every wave holds step t's fragments: tile t's buffer is free
        if constexpr (!RING_A) {
            if (++sl_cnt == S) {                       // a new slab starts at the next step: the buffer of the slab before it (this step's, already in registers) is free
                sl_cnt = 0;
                if (sl_ks < kslabs) req_a_slab(sl_k, sl_ks & 1);
                ++sl_ks; sl_k += AH * 64u;
            }
        }
        if constexpr (decltype(steady)::value) request_next();       // tile t + D -> the buffer of tile t (t + D < nsteps in the steady state)
        else if (rt < nsteps) request_next();
        if (zeroing) {                                 // rows before the start of their own sequence
#pragma unroll
            for (int i = 0; i < TM; ++i)
                if (tpos[i] + csh < 0) ca[i] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
        // The first MFMAs go out BEFORE the next step's fragment reads: the compiler cannot see the lgkmcnt(0) of ring_barrier() and waits for this
        // step's fragments (read a step ago) with lgkmcnt(0) at their first use -- behind the new reads that wait would drain them too (seen in the ISA
        // of the second build: every step waited for its LDS reads with the matrix pipe idle); in front of them it costs nothing.
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cb[j], ca[0], acc[0][j], 0, 0, 0);
        ring_sched_fence();
        if constexpr (decltype(steady)::value) read_frags(na, nb, nsh);
        else if (t + 1 < nsteps) read_frags(na, nb, nsh);
        ring_sched_fence();
#pragma unroll
        for (int i = 1; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cb[j], ca[i], acc[i][j], 0, 0, 0);
        ring_sched_fence();                            // the step's MFMAs stay in front of the next wait (the first build's sank behind the barrier)
    };
    // ---- the steady state (version 3).  Versions 1-2 ran `step` above everywhere: correct, conflict-free, requests in flight across the barriers -- and no
    // faster than gemm_dma (profiles/r06_gemm_ring.md): a step is 16 MFMAs (256 clocks of the matrix pipe) plus ~100 scalar / vector / LDS / DMA
    // instructions that the wave issued one after the other AROUND the MFMA block (a lone workgroup took ~1 050 clocks per step, 24 % of a CU's rate).
    // `fstep` is the same step as straight-line code (no branch: the tap / slab cursors advance by selects, the tile's half is a template parameter,
    // the zeroing variant is its own loop, the slab request sits between two slabs) so that the whole step is ONE scheduling region, and asks the
    // scheduler for "1 MFMA, then up to 4 others" sixteen times: the requests, address arithmetic and fragment reads issue in the shadow of the MFMAs.
    int t = 0;
    auto fstep = [&](auto zero_tag, auto h_tag, bf16x8 (&ca)[TM], bf16x8 (&cb)[TN], int csh, bf16x8 (&na)[TM], bf16x8 (&nb)[TN], int& nsh) {
        constexpr bool ZERO = decltype(zero_tag)::value;
        constexpr int H = decltype(h_tag)::value;      // taps > 1: the half of this step's tile (= of the tile requested here: D is even); the next tile's is H ^ 1
        ring_wait_vm_c<LPS * (D - 2)>();
        if constexpr (ABL & 8) { ring_sched_fence(); } else ring_barrier();
        if constexpr (!(ABL & 1)) {   // tile t + D -> the buffer of tile t
            if constexpr (RING_A) {
#pragma unroll
                for (int i = 0; i < 2; ++i) gd_dma16(A16 + (a_off[i] + r_k), Abuf + r_stage * WT + (wave + 4 * i) * 1024);
            }
            const unsigned wk = r_wtap + r_k;
#pragma unroll
            for (int i = 0; i < 2; ++i) gd_dma16(Wg + (w_off[i] + wk), Wbuf + r_stage * WT + (wave + 4 * i) * 1024);
            r_stage = r_stage + 1 == NST ? 0 : r_stage + 1;
            if constexpr (RING_A || H == 0) r_k += 64u;
            else {
                const int nt = r_tap + 1;
                const bool wrap = nt == p.taps;
                r_tap = wrap ? 0 : nt; r_wtap = wrap ? 0u : r_wtap + w_tap_bytes; r_k += wrap ? 64u : 0u - 64u;
            }
        }
        if constexpr (ZERO) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                if (tpos[i] + csh < 0) ca[i] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
        {   // the next tile's fragments
            constexpr int HH = RING_A ? 0 : (H ^ 1);
            const unsigned char* Ab = RING_A ? Abuf + f_stage * WT : Abuf + f_buf * a_buf + HH * a_half;
            const unsigned char* Wb = Wbuf + f_stage * WT + wfrag;
            const int sh = -(int)(f_shifts & 0xffu);
            nsh = sh;
            if constexpr (ABL & 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) { na[i] = ca[i]; nb[i] = cb[i]; }
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int R = (RING_A || (ABL & 16)) ? arow[i] : arow[i] + sh;      // (a plain Linear has no shift: the address is loop-invariant)
                    na[i] = *reinterpret_cast<const bf16x8*>(Ab + (R << 6) + ((lq ^ ((R >> 1) & 2)) << 4));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) nb[j] = *reinterpret_cast<const bf16x8*>(Wb + j * 1024);
            }
            f_stage = f_stage + 1 == NST ? 0 : f_stage + 1;
            if constexpr (!RING_A && HH == 1) {        // (the tile after it starts the next tap)
                const int nt = f_tap + 1;
                const bool wrap = nt == p.taps;
                f_tap = wrap ? 0 : nt; f_shifts = wrap ? shifts : f_shifts >> 8; f_buf ^= wrap ? 1 : 0;
            }
        }
        if constexpr (ABL & 4) {
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][0][0] += __builtin_bit_cast(float, (int)ca[i][0] + (int)cb[i][1]);       // (keeps the fragments alive)
        } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cb[j], ca[i], acc[i][j], 0, 0, 0);
        }
#ifndef QTTS_HOST_EMU
#pragma unroll
        for (int k = 0; k < TM * TN; ++k) {            // (the first MFMA leads: the compiler's lgkmcnt(0) for this step's fragments then sits in front of the new reads)
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x096, 4, 0);       // up to four of VALU | SALU | VMEM | DS
        }
#endif
        ring_sched_fence();
    };
    // ---- version 4 (TAPS known at compile time): a slab's TAPS * AH steps unrolled.  The ablation of version 3 (profiles/r06_gemm_ring.md) says the step is
    // bound by instruction ISSUE: with requests, reads and MFMAs all removed the cursors, waits and the loop alone still take half of the step (a SIMD issues
    // about one instruction per 4-5 clocks whichever wave it comes from: two workgroups per CU do not hide each other's scalar work), i.e. ~90
    // instructions per 16 MFMAs must become ~30.  Unrolled, a step's tap and half are constants: the fragment addresses of all taps are computed ONCE
    // (apre: TAPS * 4 registers), the tap's weight offset is one of TAPS scalars, the shift is a constant byte of `shifts`, no cursor is left but the two
    // ring stages.
    unsigned apre[TAPS > 0 ? TAPS : 1];                // LDS byte offset of this lane's A fragment 0 at tap tp inside a half-slab (swizzle included); fragment i
                                                       // sits 16 i rows = i KiB further (a multiple of 16 rows leaves the swizzle bit alone): an immediate of the read
    unsigned wtap[TAPS > 0 ? TAPS : 1];                // tap * N * K * 2
    if constexpr (TAPS > 0) {
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) {
            wtap[tp] = tp * w_tap_bytes;
            const int R = arow[0] + p.shift[tp];
            apre[tp] = (unsigned)((R << 6) + ((lq ^ ((R >> 1) & 2)) << 4));
        }
    }
    unsigned u_sk = k0;                                // k offset (bytes) of the slab being computed
    unsigned u_ab[2][AH];                              // LDS byte offset of the halves of (this slab, the next slab)
#pragma unroll
    for (int h = 0; h < AH; ++h) { u_ab[0][h] = h * a_half; u_ab[1][h] = a_buf + h * a_half; }
    auto ustep = [&](auto zero_tag, auto last_tag, auto u_tag, bf16x8 (&ca)[TM], bf16x8 (&cb)[TN], bf16x8 (&na)[TM], bf16x8 (&nb)[TN]) {
        constexpr bool ZERO = decltype(zero_tag)::value;
        constexpr bool LAST = decltype(last_tag)::value;       // the last slab: the ring drains (no request past the slab, the waits follow, no read after the last step)
        constexpr int U = decltype(u_tag)::value;      // step inside the slab: tap U / AH, half U % AH
        constexpr int SS = (TAPS > 0 ? TAPS : 1) * AH;
        // own requests of the tiles <= t + 1 are complete when only those of t + 2 .. min(t + D, nsteps) - 1 are outstanding
        constexpr int LATER = !LAST ? D - 2 : (SS - 2 - U < D - 2 ? (SS - 2 - U > 0 ? SS - 2 - U : 0) : D - 2);
        ring_wait_vm_c<LPS * LATER>();
        if constexpr (ABL & 8) { ring_sched_fence(); } else ring_barrier();
        if constexpr (!(ABL & 1) && (!LAST || U + D < SS)) {     // tile U + D of this slab, or U + D - SS of the next -> the buffer of this step's tile
            constexpr int UR = (U + D) % SS;
            constexpr unsigned kofs = ((U + D) >= SS ? AH * 64u : 0u) + (UR % AH) * 64u;
            const unsigned wk = wtap[UR / AH] + u_sk + kofs;
#pragma unroll
            for (int i = 0; i < 2; ++i) gd_dma16(Wg + (w_off[i] + wk), Wbuf + r_stage * WT + (wave + 4 * i) * 1024);
            r_stage = r_stage + 1 == NST ? 0 : r_stage + 1;
        }
        if constexpr (ZERO) {
            const int csh = -(int)((shifts >> (8 * (U / AH))) & 0xffu);
#pragma unroll
            for (int i = 0; i < TM; ++i)
                if (tpos[i] + csh < 0) ca[i] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        }
        if constexpr (!LAST || U + 1 < SS) {   // the next tile's fragments: tile U + 1 of this slab or tile 0 of the next
            constexpr int UF = (U + 1) % SS;
            const unsigned ab = u_ab[(U + 1) >= SS ? 1 : 0][UF % AH];
            const unsigned char* Wb = Wbuf + f_stage * WT + wfrag;
            if constexpr (ABL & 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) { na[i] = ca[i]; nb[i] = cb[i]; }
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i) na[i] = *reinterpret_cast<const bf16x8*>(Abuf + (apre[UF / AH] + ab) + i * 1024);
#pragma unroll
                for (int j = 0; j < TN; ++j) nb[j] = *reinterpret_cast<const bf16x8*>(Wb + j * 1024);
            }
            f_stage = f_stage + 1 == NST ? 0 : f_stage + 1;
        }
        if constexpr (ABL & 4) {
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][0][0] += __builtin_bit_cast(float, (int)ca[i][0] + (int)cb[i][1]);
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cb[j], ca[i], acc[i][j], 0, 0, 0);
        }
#if !defined(QTTS_HOST_EMU) && QTTS_RING_SGB > 0
#pragma unroll
        for (int k = 0; k < TM * TN; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x096, QTTS_RING_SGB, 0);       // up to three of VALU | SALU | VMEM | DS (build variants ring_sgb0 / 2 / 4: A/B; 3: -10 % on the 192-tile grids, profiles/r06_gemm_ring.md)
        }
#endif
        ring_sched_fence();
    };
    auto uslab = [&](auto zero_tag, auto last_tag) {   // the TAPS * AH steps of one slab (an even count: the fragment registers end where they began)
        constexpr int SS = (TAPS > 0 ? TAPS : 1) * AH;
        static_assert(TAPS == 0 || SS % 2 == 0, "an odd number of steps per slab");
        auto pair = [&](auto u) {
            ustep(zero_tag, last_tag, std::integral_constant<int, decltype(u)::value>{}, fa[0], fb[0], fa[1], fb[1]);
            ustep(zero_tag, last_tag, std::integral_constant<int, decltype(u)::value + 1>{}, fa[1], fb[1], fa[0], fb[0]);
        };
        if constexpr (SS >= 2) pair(std::integral_constant<int, 0>{});
        if constexpr (SS >= 4) pair(std::integral_constant<int, 2>{});
        if constexpr (SS >= 6) pair(std::integral_constant<int, 4>{});
        if constexpr (SS >= 8) pair(std::integral_constant<int, 6>{});
        if constexpr (SS >= 10) pair(std::integral_constant<int, 8>{});
        if constexpr (SS >= 12) pair(std::integral_constant<int, 10>{});
        if constexpr (SS >= 14) pair(std::integral_constant<int, 12>{});
        static_assert(SS <= 14, "more taps than the unrolled slab covers");
    };
    auto run_steady = [&](auto zero_tag) {
        if constexpr (TAPS > 0) {
            // every slab but the last (all of their steps have t + D < nsteps: D <= S); between two slabs the buffer of the slab just finished takes slab + 2
            for (int ks = 0; ks + 1 < kslabs; ++ks) {
                uslab(zero_tag, std::false_type{});
                t += S;
                if (ks + 2 < kslabs) req_a_slab(sl_k, ks & 1);
                sl_k += AH * 64u;
                u_sk += AH * 64u;
#pragma unroll
                for (int h = 0; h < AH; ++h) { const unsigned x = u_ab[0][h]; u_ab[0][h] = u_ab[1][h]; u_ab[1][h] = x; }
            }
            uslab(zero_tag, std::true_type{});         // the last slab, its drain known at compile time too
            t += S;
        } else if constexpr (RING_A) {
            for (; t + 1 < t_steady; t += 2) {
                fstep(zero_tag, std::integral_constant<int, 0>{}, fa[0], fb[0], fsh[0], fa[1], fb[1], fsh[1]);
                fstep(zero_tag, std::integral_constant<int, 0>{}, fa[1], fb[1], fsh[1], fa[0], fb[0], fsh[0]);
            }
        } else {
            // every slab but the last: all of its steps have t + D < nsteps (D <= S); between two slabs the buffer of the slab just finished takes slab + 2
            for (int ks = 0; ks + 1 < kslabs; ++ks) {
                for (int u = 0; u < p.taps; ++u) {
                    fstep(zero_tag, std::integral_constant<int, 0>{}, fa[0], fb[0], fsh[0], fa[1], fb[1], fsh[1]);
                    fstep(zero_tag, std::integral_constant<int, 1>{}, fa[1], fb[1], fsh[1], fa[0], fb[0], fsh[0]);
                }
                t += S;
                if (ks + 2 < kslabs) req_a_slab(sl_k, ks & 1);
                sl_k += AH * 64u;
            }
            sl_ks = kslabs;                            // (nothing left to request for the generic steps of the last slab)
        }
    };
    if (zeroing) run_steady(std::true_type{}); else run_steady(std::false_type{});
    if constexpr (TAPS == 0) {                         // (with the taps known every slab ran above, the last one with its drain)
        rt = t + D < nsteps ? t + D : nsteps;          // the cursors `step` keeps and `fstep` knows statically (t is even here)
        r_h = 0; f_h = AH - 1; sl_cnt = 0;
#pragma unroll 1
        for (; t < nsteps; t += 2) {                   // the last slab / the last D (+ 1) steps: the ring drains, the waits follow it
            step(std::false_type{}, t, fa[0], fb[0], fsh[0], fa[1], fb[1], fsh[1]);
            if (t + 1 < nsteps) step(std::false_type{}, t + 1, fa[1], fb[1], fsh[1], fa[0], fb[0], fsh[0]);
        }
    }
    if (ks > 1) {
        // Split-K combine, deterministic and without a spin: every workgroup of the tile stores its partial accumulators (fragment order: one 16-byte vector
        // per lane and fragment), takes a ticket (ONE agent-scope acq_rel atomic per workgroup, by one lane between two barriers: its release writes the
        // workgroup's stores back, its acquire invalidates what this CU / XCD may hold of the others'), and the workgroup that draws the LAST ticket sums the
        // ks partials in split order (its own from registers at its own place) and runs the epilogue; it also re-arms the counter.  Nobody waits for anybody.
        f32x4* mine = ks_ws + ((size_t)(tile * ks + split) * (TM * TN)) * 256 + tid;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) mine[(i * TN + j) * 256] = acc[i][j];
        int* flag = reinterpret_cast<int*>(smem_gr);   // (the ring is drained: nothing of it is read any more)
        __syncthreads();
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(ks_cnt + tile, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == (unsigned)(ks - 1);
            if (last) __hip_atomic_store(ks_cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = last;
        }
        __syncthreads();
        if (*flag == 0) return;
        f32x4 tot[TM][TN];
        for (int sp = 0; sp < ks; ++sp) {
            const f32x4* src = ks_ws + ((size_t)(tile * ks + sp) * (TM * TN)) * 256 + tid;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const f32x4 v = sp == split ? acc[i][j] : src[(i * TN + j) * 256];
                    tot[i][j] = sp == 0 ? v : tot[i][j] + v;
                }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = tot[i][j];
    }
    tap_epilogue<BN, TM, TN, true>(p, acc, m0, n0, wm, wn, li, lq);
}

static int ring_lds_bytes(int nst, int ah, int taps, int halo) {
    const int nbands = (128 + halo + 15) / 16;
    return nst * 8192 + (taps == 1 ? nst * 8192 : 2 * ah * nbands * 1024);
}
// Split-K workspace of the ring kernel: per stream (launches of one stream are ordered; two engines or an engine and its codec run on their own streams),
// fixed size, allocated at the stream's first split launch (an eager call: every engine's first call is) and kept for the life of the library -- captured
// graphs hold its address.  512 partial tiles of 64 KiB + one ticket counter per tile.
constexpr int RING_KS_MAX_PARTS = 512;
struct RingKsWs { f32x4* ws = nullptr; unsigned* cnt = nullptr; };
static RingKsWs ring_ks_ws(hipStream_t st) {
    static std::mutex m;
    static std::map<std::pair<int, hipStream_t>, RingKsWs> tab;
    int dev = 0;
    QTTS_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(m);
    auto it = tab.find({dev, st});
    if (it != tab.end()) return it->second;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return RingKsWs{};      // never allocate under a capture: this launch runs unsplit
    RingKsWs w;
    void* a = nullptr; void* b = nullptr;
    QTTS_CHECK_HIP(hipMalloc(&a, (size_t)RING_KS_MAX_PARTS * 128 * 128 * 4));
    QTTS_CHECK_HIP(hipMalloc(&b, RING_KS_MAX_PARTS * sizeof(unsigned)));
    QTTS_CHECK_HIP(hipMemsetAsync(b, 0, RING_KS_MAX_PARTS * sizeof(unsigned), st));       // stream-ordered in front of the first launch (a hipMemset on the legacy
                                                                                         // stream does not order with a non-blocking stream: seen as a wrong first decode on the MI355X)
    w.ws = static_cast<f32x4*>(a); w.cnt = static_cast<unsigned*>(b);
    tab[{dev, st}] = w;
    return w;
}
// K splits of a launch: QTTS_GEMM_RING_KS = n > 1: n where the shape admits it (A/B only).  DEFAULT: NONE -- measured on the 192-tile grids it was built for
// (profiles/r06_gemm_ring.md): the batch-32 prefill's o projection 27.7 -> 48.4 us, down 63.8 -> 86.7, the C = 768 unit at 1 x 10 s 50.9 -> 56.7: the
// agent-scope release of the ticket writes back the whole L2 (the launch's own fp32 output tiles are dirty in it); only a 64-tile grid gains (batch-8 down:
// 61.4 -> 40.0 us at 4 splits, gemm_wide 44.2).
static int ring_splits(const GemmTapParams& p, int nst, int ah, int n_cu) {
    const int tiles = cdiv(p.M, 128) * cdiv(p.N, 128);
    const int slabs = p.K / (32 * ah), steps = slabs * p.taps * ah;
    int ks = QTTS_OPT_INT("QTTS_GEMM_RING_KS", 0);
    (void)n_cu;
    if (ks == 0) ks = 1;
    while (ks > 1 && (slabs % ks != 0 || tiles * ks > RING_KS_MAX_PARTS || steps / ks < 2 * nst || p.act == ACT_SWIGLU)) --ks;
    return ks < 1 ? 1 : ks;
}
template <int NST, int AH, bool RING_A, int TAPS = 0, int ABL = 0>
static void launch_ring_t(const GemmTapParams& p, int halo, hipStream_t st) {
    static const int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    const int nb = cdiv(p.M, 128) * cdiv(p.N, 128);
    const int nbands = (128 + halo + 15) / 16;
    int ks = ring_splits(p, NST, AH, n_cu);
    RingKsWs w{};
    if (ks > 1) { w = ring_ks_ws(st); if (!w.ws) ks = 1; }
    auto kern = gemm_ring_kernel<NST, AH, RING_A, TAPS, ABL>;
    ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(nb * ks), dim3(256), (size_t)ring_lds_bytes(NST, AH, p.taps, halo), st, p, halo, nbands, ks, w.ws, w.cnt);
}
// depth of the ring: QTTS_GEMM_RING_NST (4 | 6 | 8), default 4 (<= 80 KB with the widest halo: two workgroups per CU; the deeper rings measured no faster on
// any grid: the step is bound by instruction issue, not by the requests' latency).  A slab is requested a slab's steps (taps * 2) before its first read and
// must stay older than the tile whose wait covers it: NST <= taps * 2 (the two-tap transposed form: 4).  The unrolled kernels exist 4 and 8 deep.
static void launch_ring(const GemmTapParams& p, int halo, hipStream_t st) {
    QTTS_REQUIRE((size_t)p.M * p.lda * 2 < (1ull << 32) && (size_t)p.taps * p.N * p.K * 2 < (1ull << 32), QTTS_ERR_LIMIT, "gemm_ring: an operand of 4 GiB or more");
    int nst = QTTS_OPT_INT("QTTS_GEMM_RING_NST", 0);
    if (nst != 4 && nst != 6 && nst != 8) nst = 4;     // (measured: deeper rings lose the second workgroup per CU and gain nothing, profiles/r06_gemm_ring.md)
    const int unroll = QTTS_OPT_INT("QTTS_GEMM_RING_UNROLL", 1);       // 0: the generic steady step for every tap count (A/B)
    if (p.taps == 1) { if (nst == 8) launch_ring_t<8, 1, true>(p, halo, st); else launch_ring_t<4, 1, true>(p, halo, st); return; }
    while (nst > p.taps * 2) nst -= 2;
    QTTS_REQUIRE(nst >= 4, QTTS_ERR_ARG, "gemm_ring: ring deeper than a slab");
#if defined(QTTS_ABLATE) && !defined(QTTS_HOST_EMU)                    // (the `ablate` build variant only: measuring code is never linked into libqtts.so)
    if (const int abl = QTTS_OPT_INT("QTTS_GEMM_RING_ABLATE", 0); abl && p.taps == 7) {        // measuring variants of the 4-deep 7-tap kernel (wrong results by construction)
        switch (abl) {
            case 1: launch_ring_t<4, 2, false, 7, 1>(p, halo, st); return;
            case 2: launch_ring_t<4, 2, false, 7, 2>(p, halo, st); return;
            case 3: launch_ring_t<4, 2, false, 7, 3>(p, halo, st); return;
            case 4: launch_ring_t<4, 2, false, 7, 4>(p, halo, st); return;
            case 8: launch_ring_t<4, 2, false, 7, 8>(p, halo, st); return;
            case 7: launch_ring_t<4, 2, false, 7, 7>(p, halo, st); return;
            case 15: launch_ring_t<4, 2, false, 7, 15>(p, halo, st); return;
            default: break;
        }
    }
#endif
    if (unroll && p.taps == 7) { if (nst == 8) launch_ring_t<8, 2, false, 7>(p, halo, st); else launch_ring_t<4, 2, false, 7>(p, halo, st); return; }
    if (unroll && p.taps == 2) { launch_ring_t<4, 2, false, 2>(p, halo, st); return; }
    if (nst == 8) launch_ring_t<8, 2, false>(p, halo, st); else if (nst == 6) launch_ring_t<6, 2, false>(p, halo, st); else launch_ring_t<4, 2, false>(p, halo, st);
}

template <int BM, int BN, bool A16, int BK>
static void launch_wide_k(const GemmTapParams& p, hipStream_t st) {
    const int nb = cdiv(p.M, BM) * cdiv(p.N, BN);
    const size_t lds = (size_t)(BM + BN) * (BK + 8) * 2;
    auto kern = gemm_wide_kernel<BM, BN, A16, BK>;
    ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(nb), dim3(256), lds, st, p);
}
// Tile of the wide-K kernel.  Three bounds, all measured on the prefill's and the codec transformer's shapes
// (profiles/r03_gemm_small_tiles.md); the launch takes about the largest:
//   a workgroup's own chain: k-steps x (0.53 us + step bytes / 65 GB/s)      (one CU, one workgroup's loads in flight)
//   a CU's share:            workgroups per CU x the bytes each pulls / 80 GB/s
//   the operator itself:     row tiles x N K 2 bytes / 6 TB/s                 (every row tile streams the whole operator)
// 64-row tiles double the workgroups where 128 rows leave CUs idle; a 256-wide k-step (64-row tiles only: registers) halves the
// exposed round trips of a long K.
struct WideTile { int bm, bn, bk; };
static int g_wide_force = -1;                    // tests / A-B tooling: bm bn bk as digits (64064256), 0 = the chooser, -1 = QTTS_GEMM_WIDE_TILE
static WideTile wide_tile(const GemmTapParams& p, bool a16, int bn_max) {
    const int force_env = QTTS_OPT_INT("QTTS_GEMM_WIDE_TILE", 0);
    const int force = g_wide_force >= 0 ? g_wide_force : force_env;
    WideTile best{128, bn_max, 128};
    // CU count of the device the launch goes to (256 on MI355X; the three rates in the cost model above are MI355X measurements)
    static const int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    double tb = 1e30;
    const double ab = a16 ? 2.0 : 4.0;
    // Round 6 (last call): 32-row tiles for launches of at most 128 rows (the codec transformer at 1 x 10 s and in a 32-stream first packet, a
    // streaming push, a one-utterance prefill): four row tiles instead of two halve the bytes a workgroup pulls per k-step -- the chain bound above --
    // and put 2-4x the CUs to work; same MFMA sequence per output element (k ascending in 32s), so the results are bit-identical to any other tile.
    // 32 x 32 has no SwiGLU epilogue (gate / up pairs need two column tiles per wave).
    for (int cm : {128, 64, 32})
        for (int cn : {128, 64, 32})
            for (int ck : {128, 256}) {
                if (cn > bn_max || p.N % cn != 0 || p.K % ck != 0) continue;
                if (cm == 32 && (p.M > 128 || ck != 256 || cn > 64)) continue;
                if (cn == 32 && (cm != 32 || p.act == ACT_SWIGLU)) continue;
                if (ck == 256 && cm == 64 && !a16 && cn != 64) continue;             // (registers: 64 x 128 fp32 at 256 would spill)
                if (ck == 256 && cm == 128) continue;
                if (force && force != (cm * 1000 + cn) * 1000 + ck) continue;
                const int wgs = cdiv(p.M, cm) * cdiv(p.N, cn), steps = p.K / ck;
                const double sb = ck * (cm * ab + cn * 2.0) * 1e-3;                    // KB per k-step
                const double chain = steps * (0.53 + sb / 65.0);
                const double cu = cdiv(wgs, n_cu) * steps * sb / 80.0;
                const double op = (double)cdiv(p.M, cm) * p.N * p.K * 2.0 / 6e6;
                const double t = std::max(chain, std::max(cu, op)) * (1.0 + 0.01 * (cm * cn < 128 * 128));   // (ties go to the larger tile)
                if (t < tb) { tb = t; best = {cm, cn, ck}; }
            }
    return best;
}
template <bool A16>
static void launch_wide(const GemmTapParams& p, int bn_max, hipStream_t st) {
    const WideTile t = wide_tile(p, A16, bn_max);
    if (t.bm == 32) { if (t.bn == 64) launch_wide_k<32, 64, A16, 256>(p, st); else launch_wide_k<32, 32, A16, 256>(p, st); return; }
    if (t.bm == 128) { if (t.bn == 128) launch_wide_k<128, 128, A16, 128>(p, st); else launch_wide_k<128, 64, A16, 128>(p, st); }
    else if (t.bk == 128) { if (t.bn == 128) launch_wide_k<64, 128, A16, 128>(p, st); else launch_wide_k<64, 64, A16, 128>(p, st); }
    else if (t.bn == 128) { if constexpr (A16) launch_wide_k<64, 128, true, 256>(p, st); else launch_wide_k<64, 128, false, 128>(p, st); }
    else launch_wide_k<64, 64, A16, 256>(p, st);
}

template <int BN, bool BF16>
static void launch_t(const GemmTapParams& p, hipStream_t st) {
    const int nb = cdiv(p.M, 128) * cdiv(p.N, BN);
    hipLaunchKernelGGL((gemm_tap_kernel<BN, BF16>), dim3(nb), dim3(256), 0, st, p);
}

void launch_gemm_tap(const GemmTapParams& p_in, bool bf16, hipStream_t st) {
    GemmTapParams p = p_in;
    const int wide_max_tiles = [] { const char* e = QTTS_ENV("QTTS_GEMM_WIDE_MAX"); return e && atoi(e) > 0 ? atoi(e) : 2048; }();
    // (grids up to 2048 tiles of 128 x 128: the batch-32 prefill's gate|up GEMM, 1536 tiles, runs 150 us here, 187 us in the BK = 32 kernel)
    QTTS_REQUIRE(p.K % 32 == 0, QTTS_ERR_ARG, "gemm_tap: K must be a multiple of 32");
    QTTS_REQUIRE(p.taps >= 1 && p.taps <= 8, QTTS_ERR_ARG, "gemm_tap: 1..8 taps");
    QTTS_REQUIRE(p.M > 0 && p.N > 0, QTTS_ERR_ARG, "gemm_tap: empty problem");
    QTTS_REQUIRE(p.lda % 4 == 0, QTTS_ERR_ARG, "gemm_tap: lda must be a multiple of 4");
    // the epilogue moves 4-column vectors (16 B fp32 / 8 B bf16) whenever shapes and pointers allow it
    auto al16 = [](const void* q) { return reinterpret_cast<uintptr_t>(q) % 16 == 0; };
    p.vec4 = p.N % 4 == 0 && p.ldc % 4 == 0 && (!p.res || p.ldr % 4 == 0) && (!p.C16 || (p.ldc16 % 4 == 0 && reinterpret_cast<uintptr_t>(p.C16) % 8 == 0)) &&
             (!p.res16 || (p.ldres16 % 4 == 0 && reinterpret_cast<uintptr_t>(p.res16) % 8 == 0)) &&
             (!p.R16 || (p.ldR16 % 4 == 0 && reinterpret_cast<uintptr_t>(p.R16) % 8 == 0)) &&
             al16(p.C) && al16(p.res) && al16(p.bias) && al16(p.scale) && al16(p.snake_ea) && al16(p.snake_ib) && al16(p.snake16_ea) &&
             al16(p.snake16_ib) && (p.snake16_period % 4 == 0);
    if (p.act == ACT_SWIGLU) QTTS_REQUIRE(p.ldc % 4 == 0 && al16(p.C) && (!p.C16 || (p.ldc16 % 4 == 0 && reinterpret_cast<uintptr_t>(p.C16) % 8 == 0)),
                                          QTTS_ERR_ARG, "gemm_tap: swiglu outputs must be 16-byte (fp32) / 8-byte (bf16) aligned with ldc % 4 == 0");
    if (p.A16 && bf16 && p.taps == 1 && p.shift[0] == 0 && p.K % 128 == 0 && p.K >= 512 && p.N % 64 == 0 && p.lda % 8 == 0 && (p.C || p.C16) && !p.res16 &&
        !p.R16 && (cdiv(p.M, 128) * cdiv(p.N, 128) <= wide_max_tiles || p.act == ACT_SWIGLU)) {       // (the tap-reuse kernel has no SwiGLU epilogue)
        // bf16 activations into a SMALL grid (the talker prefill): the deep-k kernel reading the bf16 copy
        if (p.act == ACT_SWIGLU) QTTS_REQUIRE(p.N % 32 == 0, QTTS_ERR_ARG, "gemm_tap: swiglu needs N % 32 == 0");
        {   // QTTS_GEMM_DMA=1 | 2: the LDS-DMA kernel for the plain bf16 Linear too (2: only here) -- A/B only: on the prefill's shapes it wins
            // where N is wide (q|k|v 0.87x, 4096^3 0.90x) and loses where K is deep and the grid small (o 1.13x, down 1.11x); first packet
            // 31.9 -> 32.6 ms with it (profiles/r04_gemm_dma.md), so the tile chooser's kernel stays
            const int dma_env2 = [] { const char* e = QTTS_ENV("QTTS_GEMM_DMA"); return e ? atoi(e) : 0; }();
            // Round 6: the ring kernel for the plain bf16 Linear on grids of at least 128 tiles of 128 x 128 (the batch-32 prefill: o 35.1 -> 28.7 us,
            // down 83.8 -> 65.9, gate|up 130.8 -> 124.0, q|k|v 52.0 -> 51.0; batch 8: q|k|v 26.1 -> 25.3, gate|up 46.1 -> 42.2; a 64-tile grid
            // (batch-8 down) stays with gemm_wide's 64-row tiles: 44.6 vs 61.0 -- profiles/r06_gemm_ring.md).  QTTS_GEMM_RING=2: every grid; 0: none.
            const int ring_env = QTTS_OPT_INT("QTTS_GEMM_RING", 1);
            if (dma_env2 == 0 && (ring_env == 2 || (ring_env == 1 && cdiv(p.M, 128) * (p.N / 128) >= 128)) && p.N % 128 == 0 && p.K % 64 == 0) {
                launch_ring(p, 0, st);
                QTTS_CHECK_HIP(hipGetLastError());
                return;
            }
            if ((dma_env2 == 1 || dma_env2 == 2) && p.N % 128 == 0 && p.K % 64 == 0 && cdiv(p.M, 128) * (p.N / 128) >= 128) {
                launch_dma<128>(p, 0, st);
                QTTS_CHECK_HIP(hipGetLastError());
                return;
            }
        }
        launch_wide<true>(p, p.N % 128 == 0 ? 128 : 64, st);
        QTTS_CHECK_HIP(hipGetLastError());
        return;
    }
    if (p.A16) {                           // bf16 activations: the tap-reuse kernel (bf16 mode only)
        QTTS_REQUIRE(bf16, QTTS_ERR_ARG, "gemm_tap: A16 needs bf16 weights");
        QTTS_REQUIRE(p.act != ACT_SWIGLU, QTTS_ERR_ARG, "gemm_tap: the A16 kernel has no SwiGLU epilogue");
        QTTS_REQUIRE(p.lda % 8 == 0, QTTS_ERR_ARG, "gemm_tap: bf16 A needs lda % 8 == 0");
        QTTS_REQUIRE(p.C || p.C16 || p.R16, QTTS_ERR_ARG, "gemm_tap: no output");
        QTTS_REQUIRE(!(p.res && p.res16), QTTS_ERR_ARG, "gemm_tap: residual given twice");
        int halo = 0;
        for (int i = 0; i < p.taps; ++i) { QTTS_REQUIRE(p.shift[i] <= 0, QTTS_ERR_ARG, "gemm_tap: shift > 0"); halo = std::max(halo, -p.shift[i]); }
        QTTS_REQUIRE(halo <= 56, QTTS_ERR_LIMIT, "gemm_tap: tap reach > 56 rows");
        const int bn2 = (p.N % 128 == 0) ? 128 : (p.N % 96 == 0 ? 96 : (p.N <= 64 ? 64 : 128));
        // k-slab of 32: measured 7 % faster than 64 / 96 on the codec (14.8 vs 15.95 ms per 8 x 10 s,
        // profiles/r02_config2_codec_tap2*.json) -- the smaller LDS footprint (<= 50 KB) keeps 3 workgroups per CU resident
        // Round 4: a grid that leaves every CU at most one workgroup (the C = 768 / 384 units of a B = 1 x 10 s or 32 x 4-frame decode: 192
        // tiles, 168 k-steps each) is bound by its workgroups' own step chains, not by occupancy: 64-wide k-slabs halve the steps.
        // QTTS_TAP2_BK = 32 | 64 forces one side (A/B); default: 64 when the grid has at most `n_cu` tiles and K % 64 == 0.
        const int bk_env = QTTS_OPT_INT("QTTS_TAP2_BK", 0);
        static const int n_cu2 = [] {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
            return n;
        }();
        // The LDS-DMA kernel for every 128-column launch of this path (the codec's C = 768 / 384 units and transposed convolutions):
        // measured 2.70 -> 2.54 ms at B = 1 x 10 s and 10.23 -> 9.96 ms at 8 x 10 s (profiles/r04_gemm_dma.md).  QTTS_GEMM_DMA=0: gemm_tap2.
        const int dma_env = [] { const char* e = QTTS_ENV("QTTS_GEMM_DMA"); return e ? atoi(e) : -1; }();
        // Round 6: the ring kernel (band layout, counted waits, register-double-buffered fragments) wherever gemm_dma ran; QTTS_GEMM_RING=0: gemm_dma
        // (a 1x1 convolution of this path -- K <= 768, three to twelve 64-wide steps, its time is the residual / output traffic of the epilogue -- stays with
        // gemm_dma: 151.5 vs 167.1 us at C = 384, 8 x 10 s; QTTS_GEMM_RING=2 sends it through the ring as well)
        const int ring_env = QTTS_OPT_INT("QTTS_GEMM_RING", 1);
        if (dma_env != 0 && bn2 == 128 && p.K % 64 == 0 && (ring_env == 2 || (ring_env == 1 && p.taps > 1))) { launch_ring(p, halo, st); QTTS_CHECK_HIP(hipGetLastError()); return; }
        if (dma_env != 0 && bn2 == 128 && p.K % 64 == 0) { launch_dma<128>(p, halo, st); QTTS_CHECK_HIP(hipGetLastError()); return; }
        const int tiles = cdiv(p.M, 128) * cdiv(p.N, bn2);
        const bool bk64 = p.K % 64 == 0 && bn2 == 128 && (bk_env == 64 || (bk_env == 0 && tiles <= n_cu2));
        if (bk64) launch_tap2<128, 64>(p, halo, st);
        else if (bn2 == 128) launch_tap2<128, 32>(p, halo, st); else if (bn2 == 96) launch_tap2<96, 32>(p, halo, st); else launch_tap2<64, 32>(p, halo, st);
        QTTS_CHECK_HIP(hipGetLastError());
        return;
    }
    QTTS_REQUIRE(p.C || (p.C16 && p.act == ACT_SWIGLU), QTTS_ERR_ARG, "gemm_tap: null output");
    QTTS_REQUIRE(!p.res16 && !p.R16, QTTS_ERR_ARG, "gemm_tap: the bf16 residual stream needs the A16 kernel");
    int bn;
    if (p.act == ACT_SWIGLU) {
        QTTS_REQUIRE(p.N % 32 == 0, QTTS_ERR_ARG, "gemm_tap: swiglu needs N % 32 == 0");
        bn = (p.N % 128 == 0) ? 128 : 64;
    } else if (p.N % 128 == 0) bn = 128;
    else if (p.N % 96 == 0) bn = 96;
    else if (p.N <= 64) bn = 64;
    else bn = 128;
    // small grids are bound by what the CUs holding a workgroup can pull: the deep-k kernel (bf16, plain Linear), tile by wide_tile()
    if (bf16 && p.taps == 1 && p.shift[0] == 0 && p.K % 128 == 0 && p.K >= 512 && (bn == 128 || bn == 64) &&
        cdiv(p.M, 128) * cdiv(p.N, bn) <= wide_max_tiles) {
        launch_wide<false>(p, bn, st);
        QTTS_CHECK_HIP(hipGetLastError());
        return;
    }
    if (bf16) {
        if (bn == 128) launch_t<128, true>(p, st);
        else if (bn == 96) launch_t<96, true>(p, st);
        else launch_t<64, true>(p, st);
    } else {
        if (bn == 128) launch_t<128, false>(p, st);
        else if (bn == 96) launch_t<96, false>(p, st);
        else launch_t<64, false>(p, st);
    }
    QTTS_CHECK_HIP(hipGetLastError());
}

// DEBUG/test hook: force the tile of the wide-K kernel (a forced tile the shape does not admit falls back to 128 x 128).
extern "C" __attribute__((visibility("default"))) void qtts_debug_gemm_wide_tile(int32_t code) { g_wide_force = code; }
// DEBUG/test hook (host code only, no device): the tile wide_tile() picks for a plain Linear of this shape -- pins the chooser's
// documented behaviour in the CPU suite (tests/test_hostemu.py).
extern "C" __attribute__((visibility("default")))
void qtts_debug_gemm_wide_choice(int32_t M, int32_t N, int32_t K, int32_t a16, int32_t* bm, int32_t* bn, int32_t* bk) {
    GemmTapParams p{};
    p.M = M; p.N = N; p.K = K; p.taps = 1;
    const WideTile t = wide_tile(p, a16 != 0, N % 128 == 0 ? 128 : 64);
    *bm = t.bm; *bn = t.bn; *bk = t.bk;
}

// DEBUG/perf tooling (tools/bench_gemm_small.py; not part of the product surface): a hipGraph chain of `iters` identical
// launch_gemm_tap calls on an [M][K] activation (fp32, or bf16 when a16) and a bf16 [N][K] operator; microseconds per launch.
extern "C" __attribute__((visibility("default")))
int qtts_debug_gemm_tap(int32_t M, int32_t N, int32_t K, int32_t act, int32_t with_res, int32_t a16, int32_t iters, int32_t reps,
                        double* us_per_launch) {
    try {
        QTTS_REQUIRE(us_per_launch && iters > 0 && reps > 0 && M > 0, QTTS_ERR_ARG, "bad argument");
        DevBuf A, W, Cb, R;
        A.alloc((size_t)M * K * 4); W.alloc((size_t)N * K * 2); Cb.alloc((size_t)M * N * 4); R.alloc((size_t)M * N * 4);
        QTTS_CHECK_HIP(hipMemset(A.p, 0x3c, A.bytes)); QTTS_CHECK_HIP(hipMemset(W.p, 0x3c, W.bytes)); QTTS_CHECK_HIP(hipMemset(R.p, 0, R.bytes));
        GemmTapParams p{};
        if (a16) p.A16 = A.p; else p.A = A.as<float>();
        p.lda = K; p.M = M; p.T = M; p.W = W.p; p.N = N; p.K = K; p.taps = 1; p.act = act;
        const int No = act == ACT_SWIGLU ? N / 2 : N;
        if (with_res) { p.res = R.as<float>(); p.ldr = No; }
        p.C = Cb.as<float>(); p.ldc = No;
        hipStream_t st;
        QTTS_CHECK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        launch_gemm_tap(p, true, st);
        QTTS_CHECK_HIP(hipStreamSynchronize(st));
        hipGraph_t gr; hipGraphExec_t ge;
        QTTS_CHECK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < iters; ++i) launch_gemm_tap(p, true, st);
        QTTS_CHECK_HIP(hipStreamEndCapture(st, &gr));
        QTTS_CHECK_HIP(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
        QTTS_CHECK_HIP(hipGraphLaunch(ge, st)); QTTS_CHECK_HIP(hipStreamSynchronize(st));
        hipEvent_t a, b;
        QTTS_CHECK_HIP(hipEventCreate(&a)); QTTS_CHECK_HIP(hipEventCreate(&b));
        float best = 1e30f;
        for (int r = 0; r < reps; ++r) {
            QTTS_CHECK_HIP(hipEventRecord(a, st)); QTTS_CHECK_HIP(hipGraphLaunch(ge, st)); QTTS_CHECK_HIP(hipEventRecord(b, st));
            QTTS_CHECK_HIP(hipStreamSynchronize(st));
            float ms = 0; QTTS_CHECK_HIP(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms);
        }
        *us_per_launch = 1000.0 * best / iters;
        (void)hipEventDestroy(a); (void)hipEventDestroy(b); (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(gr); (void)hipStreamDestroy(st);
        return QTTS_OK;
    } catch (const qtts::Error& e) { qtts::set_last_error(e.what()); return e.code; }
    catch (const std::exception& e) { qtts::set_last_error(e.what()); return QTTS_ERR_ARG; }
}

// DEBUG/test hook (tests/test_gpu_parity.py, tools/bench_gemm_ring.py; not part of the product surface): the bf16-activation tap GEMM (the codec decoder's
// convolution form: A16 [M][lda] bf16 bits, W [taps][N][K] bf16 bits, shifts <= 0, sequences of T rows) on host data -> fp32 C [M][N] on the host;
// with iters > 0 also the time per launch of a hipGraph chain of `iters` launches (best of `reps`).  Which kernel runs follows the option table
// (QTTS_GEMM_RING / QTTS_GEMM_DMA / QTTS_GEMM_RING_NST) exactly as in the codec engine.
extern "C" __attribute__((visibility("default")))
int qtts_debug_gemm_tap16(const void* A16_host, int32_t lda, int32_t M, int32_t T, const void* W_host, int32_t N, int32_t K, int32_t taps,
                          const int32_t* shift, float* C_host, int32_t iters, int32_t reps, double* us_per_launch) {
    try {
        QTTS_REQUIRE(A16_host && W_host && shift && M > 0 && T > 0 && N > 0 && K > 0 && taps >= 1 && taps <= 8 && lda >= K, QTTS_ERR_ARG, "bad argument");
        DevBuf A, W, Cb;
        A.alloc((size_t)M * lda * 2); W.alloc((size_t)taps * N * K * 2); Cb.alloc((size_t)M * N * 4);
        QTTS_CHECK_HIP(hipMemcpy(A.p, A16_host, A.bytes, hipMemcpyHostToDevice));
        QTTS_CHECK_HIP(hipMemcpy(W.p, W_host, W.bytes, hipMemcpyHostToDevice));
        QTTS_CHECK_HIP(hipMemset(Cb.p, 0xff, Cb.bytes));
        GemmTapParams p{};
        p.A16 = A.p; p.lda = lda; p.M = M; p.T = T; p.W = W.p; p.N = N; p.K = K; p.taps = taps;
        for (int i = 0; i < taps; ++i) p.shift[i] = shift[i];
        p.C = Cb.as<float>(); p.ldc = N;
        hipStream_t st;
        QTTS_CHECK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        launch_gemm_tap(p, true, st);
        QTTS_CHECK_HIP(hipStreamSynchronize(st));
        if (C_host) QTTS_CHECK_HIP(hipMemcpy(C_host, Cb.p, Cb.bytes, hipMemcpyDeviceToHost));
        if (iters > 0 && reps > 0 && us_per_launch) {
            hipGraph_t gr; hipGraphExec_t ge;
            QTTS_CHECK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < iters; ++i) launch_gemm_tap(p, true, st);
            QTTS_CHECK_HIP(hipStreamEndCapture(st, &gr));
            QTTS_CHECK_HIP(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
            QTTS_CHECK_HIP(hipGraphLaunch(ge, st)); QTTS_CHECK_HIP(hipStreamSynchronize(st));
            hipEvent_t a, b;
            QTTS_CHECK_HIP(hipEventCreate(&a)); QTTS_CHECK_HIP(hipEventCreate(&b));
            float best = 1e30f;
            for (int r = 0; r < reps; ++r) {
                QTTS_CHECK_HIP(hipEventRecord(a, st)); QTTS_CHECK_HIP(hipGraphLaunch(ge, st)); QTTS_CHECK_HIP(hipEventRecord(b, st));
                QTTS_CHECK_HIP(hipStreamSynchronize(st));
                float ms = 0; QTTS_CHECK_HIP(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms);
            }
            *us_per_launch = 1000.0 * best / iters;
            (void)hipEventDestroy(a); (void)hipEventDestroy(b); (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(gr);
        }
        (void)hipStreamDestroy(st);
        return QTTS_OK;
    } catch (const qtts::Error& e) { qtts::set_last_error(e.what()); return e.code; }
    catch (const std::exception& e) { qtts::set_last_error(e.what()); return QTTS_ERR_ARG; }
}

}  // namespace qtts
