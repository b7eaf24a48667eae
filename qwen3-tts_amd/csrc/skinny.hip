// skinny.hip -- weight-streaming "skinny" GEMM for the autoregressive decode step (gfx950).
//
//   out[M][N] = epi( rstd[m] * sum_k (x[m][k] * g[k]) * W[n][k] )          M = batch rows (<= 64)
//
// Decode is HBM-bound (8 flop/B at batch 8): the kernel's only job is to pull every
// weight byte across HBM exactly once at full rate.  Design:
//   * weights are re-packed at bind time into 1-KiB tiles that ARE the MFMA A-operand image
//     ([N/16 strips][K/KT k-tiles][64 lanes][16 B]); a wave's `global_load_dwordx4` therefore reads
//     1 KiB fully contiguous and the stream of a strip is one linear run -> perfectly coalesced,
//     no LDS round trip for the streamed operand (each byte is used once).
//   * MFMA roles are swapped w.r.t. the textbook: A = 16 output features x k, B = k x 16 batch rows
//     (batch padded to 16), so D holds 4 consecutive features per lane -> float4 epilogue stores.
//     bf16: v_mfma_f32_16x16x32_bf16 (x converted fp32->bf16 in registers);
//     f32 : v_mfma_f32_16x16x4_f32   (exact fp32 fma chain, parity mode).
//   * a workgroup = 8 (or 4) waves that split K of one strip (or a gate/up strip PAIR for SwiGLU) and
//     combine through LDS in a fixed order (deterministic, no atomics on the data path).
//   * the producing RMSNorm is folded in: rstd[m] factors out of the dot product, so the kernel takes
//     the per-row sum of squares (fixed-point integer accumulator -> order-independent, deterministic)
//     from the previous kernel's epilogue and applies g[k] to x on the fly; its own epilogue can emit
//     the sum of squares of what it writes (residual stream) for the next norm.
#include "common.h"
#include "kernels.h"

namespace qtts {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <bool BF16, int MT, int SPW, int NW>
__global__ __launch_bounds__(NW * 64) void skinny_kernel(SkinnyParams p) {
    if (!(p.ablate & 1) && p.done_flag && *p.done_flag) return;
    constexpr int KT = BF16 ? 32 : 16;     // k per tile
    constexpr int XV = BF16 ? 8 : 4;       // x values per lane per tile
    constexpr int U = 16 / SPW;            // k-tiles per chunk: 16 x 1-KiB weight loads in flight per wave
    __shared__ __attribute__((aligned(16))) f32x4 red[NW * SPW * MT * 64];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lj = lane & 15, lq = lane >> 4;
    const int nkt = p.K / KT;
    // k-tiles are dealt round-robin to the NW waves (tile = wave + NW*i): the workgroup's concurrent loads
    // cover one contiguous NW-KiB run of the strip's stream.
    const int my_tiles = (nkt - wave + NW - 1) / NW;
    const int nchunks = (my_tiles + U - 1) / U;
    const int strip0 = blockIdx.x * SPW;

    if (p.ss_zero && blockIdx.x == 0 && tid < 64) p.ss_zero[tid] = 0ull;

    f32x4 acc[SPW][MT];
#pragma unroll
    for (int s = 0; s < SPW; ++s)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[s][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const u32x4* wbase[SPW];
#pragma unroll
    for (int s = 0; s < SPW; ++s)
        wbase[s] = reinterpret_cast<const u32x4*>(p.Wp) + ((size_t)(strip0 + s) * nkt) * 64 + lane;

    auto load_chunk = [&](u32x4 (&w)[SPW][U], int c) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = c * U + u;
            const int kt = wave + NW * i;
#pragma unroll
            for (int s = 0; s < SPW; ++s)
                w[s][u] = (i < my_tiles && !(p.ablate & 8)) ? __builtin_nontemporal_load(wbase[s] + (size_t)kt * 64) : (u32x4){0u, 0u, 0u, 0u};
        }
    };
    auto compute_chunk = [&](u32x4 (&w)[SPW][U], int c) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = c * U + u;
            if (i >= my_tiles) break;
            const int k = (wave + NW * i) * KT + lq * XV;
            float gv[XV];
            if (p.g && !(p.ablate & 2)) {
#pragma unroll
                for (int e = 0; e < XV; e += 4) {
                    const float4 t = *reinterpret_cast<const float4*>(p.g + k + e);
                    gv[e] = t.x; gv[e + 1] = t.y; gv[e + 2] = t.z; gv[e + 3] = t.w;
                }
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int row = m * 16 + lj;
                float xv[XV];
                if (row < p.M && (p.ablate & 2)) {
#pragma unroll
                    for (int e = 0; e < XV; ++e) xv[e] = 1.0f + (float)(lane + e);
                } else if (row < p.M) {
#pragma unroll
                    for (int e = 0; e < XV; e += 4) {
                        const float4 t = *reinterpret_cast<const float4*>(p.x + (size_t)row * p.ldx + k + e);
                        xv[e] = t.x; xv[e + 1] = t.y; xv[e + 2] = t.z; xv[e + 3] = t.w;
                    }
                    if (p.g && !(p.ablate & 2)) {
#pragma unroll
                        for (int e = 0; e < XV; ++e) xv[e] *= gv[e];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < XV; ++e) xv[e] = 0.f;
                }
                if constexpr (BF16) {
                    bf16x8 xb;
#pragma unroll
                    for (int e = 0; e < 8; ++e) xb[e] = (short)f32_to_bf16(xv[e]);
#pragma unroll
                    for (int s = 0; s < SPW; ++s) {
                        bf16x8 wa;
                        *reinterpret_cast<u32x4*>(&wa) = w[s][u];
                        acc[s][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb, acc[s][m], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < SPW; ++s) {
                        f32x4 wa;
                        *reinterpret_cast<u32x4*>(&wa) = w[s][u];
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[s][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[e], xv[e], acc[s][m], 0, 0, 0);
                    }
                }
            }
        }
    };

    // ping-pong: the next chunk's 16 KiB of weight loads are in flight while the current chunk is consumed
    u32x4 wA[SPW][U], wB[SPW][U];
    load_chunk(wA, 0);
    for (int c = 0; c < nchunks; c += 2) {
        if (c + 1 < nchunks) load_chunk(wB, c + 1);
        compute_chunk(wA, c);
        if (c + 2 < nchunks) load_chunk(wA, c + 2);
        if (c + 1 < nchunks) compute_chunk(wB, c + 1);
    }

    // ---- cross-wave combine (fixed order) ----
#pragma unroll
    for (int s = 0; s < SPW; ++s)
#pragma unroll
        for (int m = 0; m < MT; ++m) red[((wave * SPW + s) * MT + m) * 64 + lane] = acc[s][m];
    __syncthreads();
    if (wave != 0) return;

    // acc[s][m][r] = partial of out[row = m*16 + lj][feature = (strip0+s)*16 + lq*4 + r]
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row = m * 16 + lj;
        f32x4 v[SPW];
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            f32x4 t = red[((0 * SPW + s) * MT + m) * 64 + lane];
#pragma unroll
            for (int w2 = 1; w2 < NW; ++w2) t += red[((w2 * SPW + s) * MT + m) * 64 + lane];
            v[s] = t;
        }
        float rstd = 1.f;
        if (p.ss_in && row < p.M && !(p.ablate & 4)) {
            const float ssum = (float)((double)p.ss_in[row] * (1.0 / SS_SCALE));
            rstd = rsqrtf(ssum / (float)p.K + p.eps);
        }
        float sq = 0.f;
        if (p.act == ACT_SWIGLU) {
            if constexpr (SPW == 2) {
                const int col = blockIdx.x * 16 + lq * 4;
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float gte = v[0][r] * rstd + (p.bias ? p.bias[(strip0)*16 + lq * 4 + r] : 0.f);
                    const float up = v[1][r] * rstd + (p.bias ? p.bias[(strip0 + 1) * 16 + lq * 4 + r] : 0.f);
                    o[r] = (gte / (1.f + expf(-gte))) * up;
                }
                if (row < p.M) {
                    if (p.res) o += *reinterpret_cast<const f32x4*>(p.res + (size_t)row * p.ldr + col);
                    *reinterpret_cast<f32x4*>(p.out + (size_t)row * p.ldo + col) = o;
                    sq = o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3];
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < SPW; ++s) {
                const int col = (strip0 + s) * 16 + lq * 4;
                f32x4 o = v[s] * rstd;
                if (p.bias && !(p.ablate & 4)) o += *reinterpret_cast<const f32x4*>(p.bias + col);
                if (row < p.M) {
                    if (p.res && !(p.ablate & 4)) o += *reinterpret_cast<const f32x4*>(p.res + (size_t)row * p.ldr + col);
                    *reinterpret_cast<f32x4*>(p.out + (size_t)row * p.ldo + col) = o;
                    sq += o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3];
                }
            }
        }
        if (p.ss_out) {
            sq += __shfl_xor(sq, 16);
            sq += __shfl_xor(sq, 32);
            if (lq == 0 && row < p.M)
                atomicAdd(p.ss_out + row, (unsigned long long)((double)sq * SS_SCALE + 0.5));
        }
    }
}

template <bool BF16, int MT>
static void launch_mt(const SkinnyParams& p, int spw, int nw, hipStream_t st) {
    const int strips = p.N / 16;
    if (nw == 8) {
        if (spw == 2) hipLaunchKernelGGL((skinny_kernel<BF16, MT, 2, 8>), dim3(strips / 2), dim3(512), 0, st, p);
        else hipLaunchKernelGGL((skinny_kernel<BF16, MT, 1, 8>), dim3(strips), dim3(512), 0, st, p);
    } else {
        if (spw == 2) hipLaunchKernelGGL((skinny_kernel<BF16, MT, 2, 4>), dim3(strips / 2), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((skinny_kernel<BF16, MT, 1, 4>), dim3(strips), dim3(256), 0, st, p);
    }
}

void launch_skinny(const SkinnyParams& p, bool bf16, hipStream_t st) {
    const int KT = bf16 ? 32 : 16;
    QTTS_REQUIRE(p.N % 16 == 0, QTTS_ERR_ARG, "skinny: N % 16");
    QTTS_REQUIRE(p.K % KT == 0, QTTS_ERR_ARG, "skinny: K must be a multiple of the k-tile");
    QTTS_REQUIRE(p.M >= 1 && p.M <= 64, QTTS_ERR_LIMIT, "skinny: 1 <= M <= 64");
    QTTS_REQUIRE(p.ldx % 4 == 0 && p.ldo % 4 == 0, QTTS_ERR_ARG, "skinny: ldx/ldo % 4");
    int spw = 1;
    if (p.act == ACT_SWIGLU) {
        QTTS_REQUIRE(p.N % 32 == 0, QTTS_ERR_ARG, "skinny: swiglu N % 32");
        spw = 2;
    } else if (p.N / 16 >= 1024 && (p.N / 16) % 2 == 0) spw = 2;
    const int nw = (p.K / KT >= 16) ? 8 : 4;     // 8 waves split K unless K is tiny
    const int mt = p.M <= 16 ? 1 : (p.M <= 32 ? 2 : 4);
    if (bf16) {
        if (mt == 1) launch_mt<true, 1>(p, spw, nw, st);
        else if (mt == 2) launch_mt<true, 2>(p, spw, nw, st);
        else launch_mt<true, 4>(p, spw, nw, st);
    } else {
        if (mt == 1) launch_mt<false, 1>(p, spw, nw, st);
        else if (mt == 2) launch_mt<false, 2>(p, spw, nw, st);
        else launch_mt<false, 4>(p, spw, nw, st);
    }
    QTTS_CHECK_HIP(hipGetLastError());
}

size_t skinny_packed_bytes(int N, int K, bool bf16) { return (size_t)N * K * (bf16 ? 2 : 4); }

void pack_skinny_weight(const float* W, int N, int K, bool bf16, void* out_host) {
    const int KT = bf16 ? 32 : 16;
    const int nkt = K / KT, strips = N / 16;
    parallel_for(strips, [&](int64_t s0, int64_t s1) {
        for (int64_t s = s0; s < s1; ++s)
            for (int kt = 0; kt < nkt; ++kt)
                for (int l = 0; l < 64; ++l) {
                    const int i = l & 15, q = l >> 4;
                    const size_t tile = ((size_t)s * nkt + kt) * 64 + l;
                    const float* src = W + (size_t)(s * 16 + i) * K + kt * KT + q * (bf16 ? 8 : 4);
                    if (bf16) {
                        bf16_t* d = reinterpret_cast<bf16_t*>(out_host) + tile * 8;
                        for (int e = 0; e < 8; ++e) d[e] = f32_to_bf16(src[e]);
                    } else {
                        float* d = reinterpret_cast<float*>(out_host) + tile * 4;
                        for (int e = 0; e < 4; ++e) d[e] = src[e];
                    }
                }
    });
}

}  // namespace qtts
