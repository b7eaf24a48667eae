// resunit.hip -- ONE kernel per residual unit of the codec decoder's blocks (gfx950, bf16 mode).
//
//   y = x + conv1x1( SnakeBeta_2( conv7_dil( SnakeBeta_1(x) ) ) )          Qwen3TTSTokenizerV2DecoderDecoderResidualUnit, V2:619-635
//
// 78 % of the codec decoder's FLOPs are these units (SURVEY.md 8a').  Until round 3 a unit was two tap-GEMM launches with the
// activated tile round-tripping through HBM in between, 128-row tiles, one barrier per 32-wide k-step and tap (21 barriers per
// conv7 tile at C = 96) and a 64 x 48 register tile per wave (two LDS operand reads per four MFMAs): 13.7 % MFMA-busy on the
// C = 96 / 192 instantiation, and the 1x1 convolution's time was its fp32 residual tile (profiles/r02_pmc_mfma_codec.md,
// r02_tstamp_codec_gemm.md).  Here, for C = 96 and C = 192 (the two blocks with the most rows):
//   * the unit's input tile -- SnakeBeta_1(x) as bf16, written by its producer -- is staged ONCE per workgroup with its causal halo
//     (6 x dilation rows) by LDS-DMA; all 7 taps read it with a row offset (zeroed in the operand registers where a row would come
//     from before the start of its sequence);
//   * weights are pre-packed at finalize into MFMA A-operand fragments (96 x 96 chunks of 18 KB) and streamed through a two-slot
//     LDS ring, one chunk per step, ONE barrier per step (7 per conv7 tile at C = 96); a chunk is requested into registers
//     three steps before its use, so no step waits for memory;
//   * a wave owns a 64-row x 96-column register tile (24 accumulators): per 32 of k it reads 4 activation + 6 weight fragments for
//     24 MFMAs -- 2.4 MFMAs per LDS operand read instead of 2, and 72 MFMAs between barriers instead of 12;
//   * conv7's accumulators never leave the registers: bias + SnakeBeta_2 are applied in place and the result IS the B operand of
//     the 1x1 convolution -- the output channels of conv7 are assigned to MFMA columns in a permuted order (tile j, column 4 q + r
//     = channel 32 (j / 2) + 8 q + 4 (j % 2) + r) so that lane (row, q) ends up holding channels 32 kk + 8 q .. + 8: exactly the
//     fragment the next MFMA wants from it.  No LDS round trip, no HBM round trip (C = 192: the two column halves of a row tile
//     live in two waves and swap their halves through LDS once);
//   * epilogue: + bias + residual (fp32, or bf16 inside a block) -> residual stream out (fp32 and / or bf16), and the NEXT consumer's
//     SnakeBeta folded in for its bf16 copy.
// Arithmetic: bf16 operands, fp32 accumulation, v_sin_f32 SnakeBeta -- the same as gemm_tap2 + tap_epilogue<FAST>.
#include <type_traits>
#include "common.h"
#include "kernels.h"
#include "tstamp.h"

QTTS_TS_UNIT(ru)

namespace qtts {

namespace {
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float ru_snake(float v, float ea, float ib) {      // y = v + ib * sin^2(v * ea), v_sin_f32 takes revolutions
    const float sn = __builtin_amdgcn_sinf(v * ea * 0.15915494309189535f);
    return v + ib * (sn * sn);
}
// MFMA column (tile j, feature f = 4 q + r) of conv7's output <-> channel inside the wave's 96-channel chunk
__host__ __device__ inline int ru_chan(int j, int f) { return 32 * (j >> 1) + 8 * (f >> 2) + 4 * (j & 1) + (f & 3); }

__device__ __forceinline__ void ru_dma16(const void* src, void* lds_dst) {       // 64 lanes x 16 B -> 1 KiB of LDS, lane-linear
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
}  // namespace

constexpr int RU_CH = 96;                          // channels per chunk
constexpr int RU_FRAG = 3 * 6 * 64 * 8;            // bf16 elements of one packed 96 x 96 chunk: [kk 3][j 6][lane 64][8]

// NC = C / 96, TM = 16-row tiles per wave.  Workgroup = 4 waves: NC = 1 -> 4 row groups; NC = 2 -> 2 row groups x 2 column halves.
// LDS: the activated input tile [(RW + halo)][C + 8] bf16 (rows 16 B apart from a bank-conflict-free stride), then the weight ring
// [2][NC][RU_FRAG] bf16.
// C = 96 runs TM = 2 (128-row tiles, 74 KB of LDS, TWO workgroups per CU): measured on the first build (TM = 4, 256-row tiles, one
// workgroup per CU) a tile took 24 us for 4 us of MFMA work, with or without the deeper weight prefetch -- a CU pulls its 300 KB of
// tile traffic (bf16 input with halo, fp32 residual in, fp32 + bf16 out) at the ~30 GB/s one CU gets, and with ONE resident
// workgroup that time adds to the compute time instead of hiding under another workgroup's MFMAs.
template <int NC, int TM, bool RES16>
__global__ __launch_bounds__(256, (TM <= 2 ? 2 : 1)) void resunit_kernel(ResUnitParams p, int halo, int a_bytes /* LDS bytes reserved for the input tile (multiple of 1024) */) {
    constexpr int C = RU_CH * NC, WR = 4 / NC, RPW = 16 * TM, RW = RPW * WR, STR = C + 8, RS = STR * 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_ru[];
    bf16_t* As = reinterpret_cast<bf16_t*>(smem_ru);
    bf16_t* Wst = reinterpret_cast<bf16_t*>(smem_ru + a_bytes);                // [2][NC][RU_FRAG]
    // per-channel parameters, staged once: b1 | ea2 | ib2 | b2 | ea16 | ib16 ([6][C] floats).  Read from global at their points of use
    // they cost a memory round trip each in the middle of the tile (in-kernel timestamps, profiles/r03_tstamp_codec_resunit.md:
    // 2.1 us for the 0.3 us 1x1 step, 8.7 us for the epilogue of a 19 us tile).
    float* Pst = reinterpret_cast<float*>(smem_ru + a_bytes + 2 * NC * RU_FRAG * 2);

    QTTS_TS_BEGIN();                                   // (tstamp build: 1 = tile staged, 2 = step 0 done, 3 = conv7 done, 4 = 1x1 done, 5 = stored)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave % WR, wc = wave / WR;
    const int li = lane & 15, lq = lane >> 4;
    const int m0 = blockIdx.x * RW;
    const unsigned char* A16 = reinterpret_cast<const unsigned char*>(p.A16);
    const bf16_t* W1p = reinterpret_cast<const bf16_t*>(p.W1p);
    const bf16_t* W2p = reinterpret_cast<const bf16_t*>(p.W2p);

    // ---- 0. weight stages 0..2 are requested first (global -> registers), then the input tile (rows m0 - halo .. m0 + RW - 1) by
    // LDS-DMA.  LDS byte o -> (row, col) = divmod(o, RS); the 16-B row pad and rows outside [0, M) fetch some valid address: such
    // rows only ever feed output rows that are not stored (rows >= M) or are zeroed at the operand (rows before a sequence start).
    //
    // The weight stream: stage k = 96 x 96 chunk(s) (tap, kc) of conv7 for k < 7 NC, then the NC chunks of the 1x1 convolution --
    // NC x 18 KB each, fragment order, so a stage is a linear copy.  A stage travels global -> registers THREE steps ahead of its
    // use and registers -> LDS ring slot one step ahead (first measurement of this kernel, profiles/r03_codec_*call3*: with the
    // stage requested by LDS-DMA at the top of the step before its use, a step -- 72 MFMAs per wave, 0.5 us -- waited ~2.4 us for
    // the DMA at its closing barrier: 16-20 % MFMA-busy.  Register staging lets the compiler wait for exactly the three-steps-old
    // loads; nothing is in flight towards LDS at a barrier).
    constexpr int TS = 8 * NC;                                   // stages = steps: 7 NC of conv7 + NC of the 1x1 convolution
    constexpr int NCHUNK = NC * RU_FRAG * 2 / 16;                // 16-B chunks per stage
    constexpr int NR = (NCHUNK + 255) / 256;
    u32x4 wreg[3][NR];
    auto stage_ptr = [&](int k) -> const u32x4* {
        return reinterpret_cast<const u32x4*>(k < 7 * NC ? W1p + (size_t)k * NC * RU_FRAG : W2p + (size_t)(k - 7 * NC) * NC * RU_FRAG);
    };
    auto load_stage = [&](u32x4 (&r)[NR], int k) {
        const u32x4* src = stage_ptr(k);
#pragma unroll
        for (int i = 0; i < NR; ++i) { const int idx = tid + 256 * i; r[i] = src[idx < NCHUNK ? idx : 0]; }
    };
    auto store_stage = [&](const u32x4 (&r)[NR], int slot) {
        u32x4* dst = reinterpret_cast<u32x4*>(Wst + (size_t)slot * NC * RU_FRAG);
#pragma unroll
        for (int i = 0; i < NR; ++i) { const int idx = tid + 256 * i; if (idx < NCHUNK) dst[idx] = r[i]; }
    };
    load_stage(wreg[0], 0); load_stage(wreg[1], 1); load_stage(wreg[2], 2);
    const int a_need = (RW + halo) * RS;
    for (int c = wave; c * 1024 < a_need; c += 4) {
        const int o = c * 1024 + lane * 16;
        const int row = o / RS, col = o - row * RS;
        int gr = m0 - halo + row;
        gr = gr < 0 ? 0 : (gr >= p.M ? p.M - 1 : gr);
        const unsigned char* src = A16 + (size_t)gr * p.lda * 2 + (col < C * 2 ? col : 0);
        ru_dma16(src, smem_ru + c * 1024);
    }
    {
        const float* srcs[6] = {p.b1, p.ea2, p.ib2, p.b2, p.ea16 ? p.ea16 : p.b2, p.ib16 ? p.ib16 : p.b2};
#pragma unroll
        for (int q = 0; q < 6; ++q)
            for (int c4 = tid; c4 < C / 4; c4 += 256) *reinterpret_cast<f32x4*>(&Pst[q * C + c4 * 4]) = *reinterpret_cast<const f32x4*>(srcs[q] + c4 * 4);
    }
    store_stage(wreg[0], 0);
    load_stage(wreg[0], 3);

    f32x4 acc[TM][6];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int tpos[TM];                                       // position of this lane's rows inside their sequence
#pragma unroll
    for (int i = 0; i < TM; ++i) tpos[i] = (m0 + wr * RPW + i * 16 + li) % p.T;
    u32x4 a2[TM][3];                                    // SnakeBeta_2(conv7) of this wave's rows x 96 channels: B operand of the 1x1
    // the residual tile of the epilogue, requested at the top of the LAST step so that it arrives under that step's MFMAs
    typename std::conditional<RES16, uint2, f32x4>::type resv[TM][6];
    __syncthreads();                                   // (the input tile's DMA has landed)
    QTTS_TS(1);

    // ---- 1..3. the steps, fully unrolled (register sets and ring slots are compile-time): conv7 (tap, kc), then the 1x1 (kc)
#pragma unroll
    for (int s = 0; s < TS; ++s) {
        if (s + 1 < TS) store_stage(wreg[(s + 1) % 3], (s + 1) & 1);       // stage s + 1: requested >= 2 steps ago; its slot was last read in step s - 1
        if (s + 4 < TS) load_stage(wreg[(s + 1) % 3], s + 4);
        const bf16_t* Wb = Wst + ((size_t)(s & 1) * NC + wc) * RU_FRAG;
        if (s == 7 * NC) {
            // ---- 2. bias + SnakeBeta_2 in place; the result is the B operand of the 1x1 convolution (this wave's 96 channels, 3 k-steps)
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                const int c0 = wc * RU_CH + 32 * kk + 8 * lq;              // channels c0 .. c0 + 7 of this lane
                const f32x4 bA = *reinterpret_cast<const f32x4*>(&Pst[c0]), bB = *reinterpret_cast<const f32x4*>(&Pst[c0 + 4]);
                const f32x4 eA = *reinterpret_cast<const f32x4*>(&Pst[C + c0]), eB = *reinterpret_cast<const f32x4*>(&Pst[C + c0 + 4]);
                const f32x4 iA = *reinterpret_cast<const f32x4*>(&Pst[2 * C + c0]), iB = *reinterpret_cast<const f32x4*>(&Pst[2 * C + c0 + 4]);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = ru_snake(acc[i][2 * kk][r] + bA[r], eA[r], iA[r]);
                        v[4 + r] = ru_snake(acc[i][2 * kk + 1][r] + bB[r], eB[r], iB[r]);
                    }
                    a2[i][kk][0] = pack_bf16(v[0], v[1]); a2[i][kk][1] = pack_bf16(v[2], v[3]);
                    a2[i][kk][2] = pack_bf16(v[4], v[5]); a2[i][kk][3] = pack_bf16(v[6], v[7]);
                }
            }
            if constexpr (NC == 2) {
                // the 1x1 convolution contracts over all 192 channels: the other half of this row tile lives in the wave with the same
                // row group and the other column half.  Every wave publishes its half as [row][channel] bf16 in the (now free: the
                // barrier that closed step s - 1 is behind every wave) input-tile area.
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int kk = 0; kk < 3; ++kk)
                        *reinterpret_cast<u32x4*>(&As[(wr * RPW + i * 16 + li) * STR + wc * RU_CH + 32 * kk + 8 * lq]) = a2[i][kk];
                __syncthreads();
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if (s == TS - 1) {
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int m = m0 + wr * RPW + i * 16 + li, mc = m < p.M ? m : p.M - 1;
                    const int n = wc * RU_CH + 16 * j + 4 * lq;
                    if constexpr (RES16) resv[i][j] = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(p.res16) + (size_t)mc * p.ldr + n);
                    else resv[i][j] = *reinterpret_cast<const f32x4*>(p.res + (size_t)mc * p.ldr + n);
                }
        }
        u32x4 wf[3][6];
#pragma unroll
        for (int kk = 0; kk < 3; ++kk)
#pragma unroll
            for (int j = 0; j < 6; ++j) wf[kk][j] = *reinterpret_cast<const u32x4*>(&Wb[((kk * 6 + j) * 64 + lane) * 8]);
        if (s < 7 * NC) {                              // conv7, step (tap, kc)
            const int tap = s / NC, kc = s - tap * NC;
            const int sh = -(6 - tap) * p.dil;         // output row m reads staged row (m - m0) + halo + sh
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const bf16_t* ar = &As[(wr * RPW + i * 16 + li + halo + sh) * STR + kc * RU_CH + lq * 8];
                const bool zero = tpos[i] + sh < 0;    // before the start of its own sequence: the causal left padding
#pragma unroll
                for (int kk = 0; kk < 3; ++kk) {
                    u32x4 a = *reinterpret_cast<const u32x4*>(ar + kk * 32);
                    if (zero) a = (u32x4){0u, 0u, 0u, 0u};
                    bf16x8 ab;
                    *reinterpret_cast<u32x4*>(&ab) = a;
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        bf16x8 wb;
                        *reinterpret_cast<u32x4*>(&wb) = wf[kk][j];
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, ab, acc[i][j], 0, 0, 0);
                    }
                }
            }
        } else {                                       // the 1x1 convolution, chunk kc of its contraction
            const int kc = s - 7 * NC;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int kk = 0; kk < 3; ++kk) {
                    u32x4 a = a2[i][kk];
                    if (NC == 2 && kc != wc) a = *reinterpret_cast<const u32x4*>(&As[(wr * RPW + i * 16 + li) * STR + kc * RU_CH + 32 * kk + 8 * lq]);
                    bf16x8 ab;
                    *reinterpret_cast<u32x4*>(&ab) = a;
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        bf16x8 wb;
                        *reinterpret_cast<u32x4*>(&wb) = wf[kk][j];
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, ab, acc[i][j], 0, 0, 0);
                    }
                }
        }
        if (s + 1 < TS) __syncthreads();               // slot (s + 1) & 1 is complete for step s + 1; slot s & 1 may be overwritten in step s + 1
#if QTTS_TSTAMP
        if (s == 0) QTTS_TS(2);
        if (s == 7 * NC - 1) QTTS_TS(3);
        if (s == TS - 1) QTTS_TS(4);
#endif
    }

    // ---- 4. epilogue: + bias + residual -> residual stream out (fp32 and / or bf16); the next consumer's SnakeBeta folded into its
    // bf16 copy.  Parameters come from LDS, the residual tile is already in registers: nothing here waits for memory but the stores.
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int n = wc * RU_CH + 16 * j + 4 * lq;
        const f32x4 b2 = *reinterpret_cast<const f32x4*>(&Pst[3 * C + n]);
        const f32x4 e16 = *reinterpret_cast<const f32x4*>(&Pst[4 * C + n]);
        const f32x4 i16 = *reinterpret_cast<const f32x4*>(&Pst[5 * C + n]);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wr * RPW + i * 16 + li;
            if (m >= p.M) continue;
            f32x4 r;
            if constexpr (RES16) r = (f32x4){__uint_as_float(resv[i][j].x << 16), __uint_as_float(resv[i][j].x & 0xffff0000u),
                                             __uint_as_float(resv[i][j].y << 16), __uint_as_float(resv[i][j].y & 0xffff0000u)};
            else r = resv[i][j];
            f32x4 v = acc[i][j] + b2 + r;
            if (p.C) *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + n) = v;
            if (p.R16) {
                uint2 h;
                h.x = pack_bf16(v[0], v[1]); h.y = pack_bf16(v[2], v[3]);
                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.R16) + (size_t)m * p.ldc + n) = h;
            }
            if (p.C16) {
                if (p.ea16) {
#pragma unroll
                    for (int r2 = 0; r2 < 4; ++r2) v[r2] = ru_snake(v[r2], e16[r2], i16[r2]);
                }
                uint2 h;
                h.x = pack_bf16(v[0], v[1]); h.y = pack_bf16(v[2], v[3]);
                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C16) + (size_t)m * p.ldc16 + n) = h;
            }
        }
    }
#if QTTS_TSTAMP
    ts_[5] = qtts::ts_drained();
    if (threadIdx.x == 0 && blockIdx.x % 499 == 7) {   // a sample of workgroups across the whole launch (the shared macro takes first / last only)
        const unsigned i_ = atomicAdd(&qtts::ts_cnt_ru, 1u);
        if (i_ < qtts::TS_CAP) {
            qtts::TsRec r_;
            for (int k_ = 0; k_ < 6; ++k_) r_.t[k_] = ts_[k_];
            r_.kind = 5; r_.a = NC * 100 + TM; r_.b = p.dil; r_.blk = (int)blockIdx.x;
            qtts::ts_log_ru[i_] = r_;
        }
    }
#endif
}

bool resunit_supported(int C) { return C == 96 || C == 192; }
size_t resunit_packed_elems(int C, int taps) { return (size_t)taps * (C / RU_CH) * (C / RU_CH) * RU_FRAG; }

// W[taps][N = C][K = C] (row-major f32) -> [tap][kc][wc][kk 3][j 6][lane 64][8] bf16: the fragment of lane (f = lane & 15, q = lane >> 4)
// is W[tap][96 wc + col(j, f)][96 kc + 32 kk + 8 q .. + 8], col = the permuted channel order for conv7's weights (permute_cols),
// the plain order 16 j + f for the 1x1 convolution.
void pack_resunit_weight(const float* W, int C, int taps, bool permute_cols, bf16_t* out) {
    const int NC = C / RU_CH;
    for (int tap = 0; tap < taps; ++tap)
        for (int kc = 0; kc < NC; ++kc)
            for (int wc = 0; wc < NC; ++wc)
                for (int kk = 0; kk < 3; ++kk)
                    for (int j = 0; j < 6; ++j)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int f = lane & 15, q = lane >> 4;
                            const int n = RU_CH * wc + (permute_cols ? ru_chan(j, f) : 16 * j + f);
                            const float* src = W + ((size_t)tap * C + n) * C + RU_CH * kc + 32 * kk + 8 * q;
                            bf16_t* dst = out + ((((((size_t)tap * NC + kc) * NC + wc) * 3 + kk) * 6 + j) * 64 + lane) * 8;
                            for (int e = 0; e < 8; ++e) dst[e] = f32_to_bf16(src[e]);
                        }
}

template <int NC, int TM, bool RES16>
static void launch_ru_r(const ResUnitParams& p, hipStream_t st) {
    constexpr int C = RU_CH * NC, RW = 16 * TM * (4 / NC);
    const int halo = 6 * p.dil;
    const int a_bytes = ((RW + halo) * (C + 8) * 2 + 1023) & ~1023;
    const size_t lds = (size_t)a_bytes + 2 * NC * RU_FRAG * 2 + 6 * C * sizeof(float);
    QTTS_REQUIRE(lds <= 160 * 1024, QTTS_ERR_LIMIT, "resunit: LDS budget exceeded");
    auto kern = resunit_kernel<NC, TM, RES16>;
    ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(cdiv(p.M, RW)), dim3(256), lds, st, p, halo, a_bytes);
}
template <int NC, int TM>
static void launch_ru(const ResUnitParams& p, hipStream_t st) {
    if (p.res16) launch_ru_r<NC, TM, true>(p, st); else launch_ru_r<NC, TM, false>(p, st);
}

void launch_resunit(const ResUnitParams& p, hipStream_t st) {
    QTTS_REQUIRE(resunit_supported(p.Cch), QTTS_ERR_ARG, "resunit: C must be 96 or 192");
    QTTS_REQUIRE(p.A16 && ((p.res != nullptr) != (p.res16 != nullptr)) && p.W1p && p.W2p && p.b1 && p.b2 && p.ea2 && p.ib2 && (p.C || p.C16 || p.R16),
                 QTTS_ERR_ARG, "resunit: null operand (exactly one of res / res16; at least one output)");
    QTTS_REQUIRE(p.M > 0 && p.T > 0 && p.dil >= 1 && 6 * p.dil <= 56, QTTS_ERR_ARG, "resunit: shape");
    QTTS_REQUIRE(p.lda % 8 == 0 && p.ldr % 4 == 0 && (!p.C || p.ldc % 4 == 0) && (!p.C16 || p.ldc16 % 4 == 0), QTTS_ERR_ARG, "resunit: leading dimensions");
    QTTS_REQUIRE((p.ea16 == nullptr) == (p.ib16 == nullptr), QTTS_ERR_ARG, "resunit: snake16 parameters go together");
    // rows per wave: QTTS_RESUNIT_TM = 2 | 4 overrides the C = 96 default of 2 (A/B runs; read per launch)
    const char* e = QTTS_ENV("QTTS_RESUNIT_TM");
    if (p.Cch == 96) { if (e && e[0] == '4') launch_ru<1, 4>(p, st); else launch_ru<1, 2>(p, st); }
    else launch_ru<2, 4>(p, st);
    QTTS_CHECK_HIP(hipGetLastError());
}

}  // namespace qtts
