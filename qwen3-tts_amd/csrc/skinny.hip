// skinny.hip -- weight-streaming "skinny" GEMM for the autoregressive decode step (gfx950).
//
//   out[M][N] = epi( rstd[m] * sum_k x[m][k] * W'[n][k] )      M = batch rows (<= 64),  W' = W . diag(g)
//
// Decode is HBM-bound (8 flop/B at batch 8): the kernel's job is to pull every weight byte
// across HBM exactly once, with as little fixed latency around that stream as possible (a frame step is a
// chain of ~450 of these launches, so every serialized memory round trip inside the kernel is paid 450x).
//   * weights are re-packed at bind time into 1-KiB tiles that ARE the MFMA A-operand image
//     ([N/16 strips][K/KT k-tiles][64 lanes][16 B]); a wave's `global_load_dwordx4` reads 1 KiB fully
//     contiguous, non-temporal, straight to VGPRs (each byte is used once: no LDS round trip).
//     The RMSNorm weight g is folded into W at bind (W' = W.diag(g)); rstd[m] factors out of the dot product.
//   * MFMA roles are swapped w.r.t. the textbook: A = 16 output features x k, B = k x 16 batch rows
//     (batch padded to 16), so D holds 4 consecutive features per lane -> float4 epilogue stores.
//     bf16: v_mfma_f32_16x16x32_bf16; f32: v_mfma_f32_16x16x4_f32 (exact fp32 fma chain, parity mode).
//   * a workgroup = 8 (or 4) waves; k-tiles are dealt round-robin to the waves; every wave keeps two chunks
//     (16 KiB) of weight loads in flight; partial sums are combined through LDS in a fixed order
//     (deterministic, no atomics anywhere).
//   * bf16 mode, M <= 16: x (M x K fp32, produced by the previous kernel) is staged ONCE per workgroup through
//     LDS with fully coalesced 16-B loads and converted by v_cvt_pk_bf16_f32; because every workgroup then sees
//     the complete rows it computes rstd = rsqrt(mean(x^2)+eps) itself (fp32, fixed reduction order) -- no
//     cross-kernel reduction at all.  Other modes load x fragments directly and take row sums from `ss_in`.
//   * everything the epilogue needs (residual, bias) is loaded at kernel entry, under the weight stream.
#include "common.h"
#include "kernels.h"

namespace qtts {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 hwbf16x2 __attribute__((ext_vector_type(2)));

__device__ inline unsigned pack_bf16(float a, float b) {   // v_cvt_pk_bf16_f32 (RNE)
    hwbf16x2 v = {(__bf16)a, (__bf16)b};
    return *reinterpret_cast<unsigned*>(&v);
}

// Weight-tile load flavour.  Default (the measured one): non-temporal -- every byte is used once per launch.
// A/B variant (build.py VARIANTS, never the default build): QTTS_SKINNY_WLOAD=1 plain loads -- hypothesis: the code
// predictor's 157 MB of layer weights are re-read by all 15 passes of a frame and fit the 256 MB Infinity Cache, which
// streaming-hinted loads may decline to allocate in.  (A per-GEMM runtime choice between the two flavours does not
// survive the compiler: it merges `cond ? *p : nontemporal(*p)` into one plain load.)
#ifndef QTTS_SKINNY_WLOAD
#define QTTS_SKINNY_WLOAD 0
#endif
#ifndef QTTS_SKINNY_GU8
#define QTTS_SKINNY_GU8 0
#endif
#ifndef QTTS_SKINNY_LATE_NORM
#define QTTS_SKINNY_LATE_NORM 0
#endif
template <class T>
__device__ inline T skinny_wload(const T* ptr) {
#if QTTS_SKINNY_WLOAD == 1
    return *ptr;
#else
    return __builtin_nontemporal_load(ptr);
#endif
}

// FS = output features per strip (16 | 8 | 4).  Narrow strips put GEMMs with few output features on all 256 CUs
// (a CU pulls only ~24 GB/s); lanes with (lane & 15) >= FS carry no weights and their MFMA rows are ignored.
template <bool BF16, int MT, int SPW, int NW, bool STAGE, int FS>
__global__ __launch_bounds__(NW * 64) void skinny_kernel(SkinnyParams p) {
    constexpr int KT = BF16 ? 32 : 16;     // k per tile
    constexpr int XV = BF16 ? 8 : 4;       // x values per lane per tile
    constexpr int U = 8 / SPW;             // k-tiles per chunk; two chunks (16 x 1 KiB) in flight per wave
    constexpr int NT = NW * 64;
    // LDS: rsum [16*MT][NW] partial sum(x^2) | red [NW*SPW*MT*64] f32x4 | xs [M][K+8] bf16 (STAGE).  With more than one
    // m-tile the x image (up to 132 KB at M = 32, K = 2048) and `red` do not both fit, and `red` is only written after
    // the last read of xs -- so they share the space (one extra barrier, ALIAS kernels only).
    constexpr bool ALIAS = STAGE && MT > 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_sk[];
    float* rsum = reinterpret_cast<float*>(smem_sk);
    f32x4* red = reinterpret_cast<f32x4*>(smem_sk + (size_t)16 * MT * NW * 4);
    unsigned short* xs = reinterpret_cast<unsigned short*>(smem_sk + (size_t)16 * MT * NW * 4 +
                                                           (ALIAS ? 0 : (size_t)NW * SPW * MT * 64 * 16));

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int lj = lane & 15, lq = lane >> 4;
    const int nkt = p.K / KT;
    const int my_tiles = (nkt - wave + NW - 1) / NW;       // tile = wave + NW*i
    const int nchunks = (my_tiles + U - 1) / U;
    const int strip0 = blockIdx.x * SPW;
    const int XS = p.K + 8;                                // LDS row stride (elements): rows shift by 4 banks

    f32x4 acc[SPW][MT];
#pragma unroll
    for (int s = 0; s < SPW; ++s)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[s][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const u32x4* wbase[SPW];
#pragma unroll
    for (int s = 0; s < SPW; ++s)
        wbase[s] = reinterpret_cast<const u32x4*>(p.Wp) + ((size_t)(strip0 + s) * nkt) * (FS * 4) + lq * FS + lj;

    auto load_chunk = [&](u32x4 (&w)[SPW][U], int c) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = c * U + u;
            const int kt = wave + NW * i;
#pragma unroll
            for (int s = 0; s < SPW; ++s)
                w[s][u] = (i < my_tiles && lj < FS && !(p.ablate & 8)) ? skinny_wload(wbase[s] + (size_t)kt * (FS * 4))
                                                            : (u32x4){0u, 0u, 0u, 0u};
        }
    };

    // ---- 1. the weight stream starts first: two chunks per wave in flight
    u32x4 wA[SPW][U], wB[SPW][U];
    load_chunk(wA, 0);
    load_chunk(wB, 1);

    // ---- 2. epilogue operands of wave 0 are fetched now, under the weight stream
    f32x4 resv[SPW][MT], biasv[SPW];
    const bool epi_loads = wave == 0 && !(p.ablate & 4);
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
#if QTTS_SKINNY_GU8
        const int col = p.act == ACT_SWIGLU ? blockIdx.x * 8 + (lq & 1) * 4 : (strip0 + s) * FS + lq * 4;
#else
        const int col = (p.act == ACT_SWIGLU ? blockIdx.x * 16 : (strip0 + s) * FS) + lq * 4;
#endif
        biasv[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (epi_loads && p.bias && lq * 4 < FS) biasv[s] = *reinterpret_cast<const f32x4*>(p.bias + (strip0 + s) * FS + lq * 4);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            resv[s][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int row = m * 16 + lj;
            if (epi_loads && p.res && row < p.M && lq * 4 < FS && (p.act != ACT_SWIGLU || s == 0))
                resv[s][m] = *reinterpret_cast<const f32x4*>(p.res + (size_t)row * p.ldr + col);
        }
    }
    const int done = (p.done_flag && !(p.ablate & 1)) ? *p.done_flag : 0;

    // ---- 3. x: coalesced global -> bf16 -> LDS, with the per-row sum of squares on the way
    float rstd_l[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) rstd_l[m] = 1.f;
    if constexpr (STAGE) {
        if (p.x_bf16) {
            // x is already bf16: LDS-DMA (global_load_lds, 1 KiB per wave-instruction, no VGPRs, all requests in
            // flight at once).  The LDS image is lane-linear, so the row padding is applied on the SOURCE side:
            // LDS byte o -> (row, col) = divmod(o, RS); lanes that land in the 16-B row pad fetch a dummy.
            const int RS = XS * 2;
            const int nbytes = p.M * RS;
            const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.x);
            unsigned char* xs_b = reinterpret_cast<unsigned char*>(xs);
            for (int c = wave; c * 1024 < nbytes; c += NW) {
                const int o = c * 1024 + lane * 16;
                const int row = o / RS, col = o - row * RS;
                const bool ok = row < p.M && col < p.K * 2 && !(p.ablate & 2);
                const unsigned char* src = ok ? xb + (size_t)row * p.ldx * 2 + col : xb;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(xs_b + c * 1024), 16, 0, 0);
            }
        } else {
        const int k4 = p.K >> 2;                            // float4 per row; k4 % 64 == 0 (launcher checks)
        const int total = p.M * k4;
        for (int r = lane; r < 16 * MT; r += 64) rsum[r * NW + wave] = 0.f;        // this wave's private slots
        for (int i0 = 0; i0 < total; i0 += NT * 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = i0 + u * NT + tid;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx < total && !(p.ablate & 2)) {
                    const int row = idx / k4, c = idx - row * k4;
                    v[u] = *reinterpret_cast<const float4*>(p.x + (size_t)row * p.ldx + c * 4);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = i0 + u * NT + tid;
                const int row = idx / k4, c = idx - row * k4;   // wave-uniform row (64 consecutive idx, k4 % 64 == 0)
                if (idx < total) {
                    uint2 h;
                    h.x = pack_bf16(v[u].x, v[u].y);
                    h.y = pack_bf16(v[u].z, v[u].w);
                    *reinterpret_cast<uint2*>(&xs[row * XS + c * 4]) = h;
                }
                if (p.norm) {
                    float q = v[u].x * v[u].x + v[u].y * v[u].y + v[u].z * v[u].z + v[u].w * v[u].w;
                    q = wave_sum64_dpp(q);
                    if (lane == 0 && idx < total) rsum[row * NW + wave] += q;
                }
            }
        }
        }
    }
    if (done) return;
    if constexpr (STAGE) {
        __syncthreads();
#if QTTS_SKINNY_LATE_NORM
        if (p.norm && p.x_bf16) {
            // A/B variant (build.py VARIANTS): rstd is only needed by the epilogue, so the row variances are taken AFTER the
            // MFMA loop (below) and ride on the final barrier -- one workgroup barrier less in front of the first MFMA.
        } else
#endif
        if (p.norm && p.x_bf16) {
            // variance from the bf16 image (what the reference's bf16 path sees): wave w reduces rows w, w+NW, ...
            for (int row = wave; row < p.M; row += NW) {
                float q = 0.f;
                for (int c = lane * 8; c < p.K; c += 512) {
                    const u32x4 t = *reinterpret_cast<const u32x4*>(&xs[row * XS + c]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a = __uint_as_float(t[e] << 16), b2 = __uint_as_float(t[e] & 0xffff0000u);
                        q += a * a + b2 * b2;
                    }
                }
                q = wave_sum64_dpp(q);
                if (lane == 0) rsum[row * NW] = q;
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int row = m * 16 + lj;
                if (row < p.M) rstd_l[m] = rsqrtf(rsum[row * NW] / (float)p.K + p.eps);
            }
        } else if (p.norm) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int row = m * 16 + lj;
                if (row < p.M) {
                    float s = 0.f;
#pragma unroll
                    for (int w2 = 0; w2 < NW; ++w2) s += rsum[row * NW + w2];     // fixed order
                    rstd_l[m] = rsqrtf(s / (float)p.K + p.eps);
                }
            }
        }
    } else if (p.norm) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int row = m * 16 + lj;
            if (row < p.M) rstd_l[m] = rsqrtf(p.ss_in[row] / (float)p.K + p.eps);
        }
    }

    auto compute_chunk = [&](u32x4 (&w)[SPW][U], int c) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = c * U + u;
            if (i >= my_tiles) break;
            const int k = (wave + NW * i) * KT + lq * XV;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int row = m * 16 + lj;
                if constexpr (BF16) {
                    u32x4 t = (u32x4){0u, 0u, 0u, 0u};
                    if constexpr (STAGE) {
                        if (row < p.M) t = *reinterpret_cast<const u32x4*>(&xs[row * XS + k]);
                    } else {
                        if (row < p.M && !(p.ablate & 2)) {
                            const float4 a = *reinterpret_cast<const float4*>(p.x + (size_t)row * p.ldx + k);
                            const float4 b = *reinterpret_cast<const float4*>(p.x + (size_t)row * p.ldx + k + 4);
                            t[0] = pack_bf16(a.x, a.y); t[1] = pack_bf16(a.z, a.w);
                            t[2] = pack_bf16(b.x, b.y); t[3] = pack_bf16(b.z, b.w);
                        }
                    }
                    bf16x8 xb;
                    *reinterpret_cast<u32x4*>(&xb) = t;
#pragma unroll
                    for (int s = 0; s < SPW; ++s) {
                        bf16x8 wa;
                        *reinterpret_cast<u32x4*>(&wa) = w[s][u];
                        acc[s][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb, acc[s][m], 0, 0, 0);
                    }
                } else {
                    float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (row < p.M && !(p.ablate & 2)) xv = *reinterpret_cast<const float4*>(p.x + (size_t)row * p.ldx + k);
#pragma unroll
                    for (int s = 0; s < SPW; ++s) {
                        f32x4 wa;
                        *reinterpret_cast<u32x4*>(&wa) = w[s][u];
                        acc[s][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[0], xv.x, acc[s][m], 0, 0, 0);
                        acc[s][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[1], xv.y, acc[s][m], 0, 0, 0);
                        acc[s][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[2], xv.z, acc[s][m], 0, 0, 0);
                        acc[s][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[3], xv.w, acc[s][m], 0, 0, 0);
                    }
                }
            }
        }
    };

    // ---- 4. consume: ping-pong, the chunk after next is requested as soon as a buffer frees up
    for (int c = 0; c < nchunks; c += 2) {
        compute_chunk(wA, c);
        if (c + 2 < nchunks) load_chunk(wA, c + 2);
        if (c + 1 < nchunks) compute_chunk(wB, c + 1);
        if (c + 3 < nchunks) load_chunk(wB, c + 3);
    }

#if QTTS_SKINNY_LATE_NORM
    if constexpr (STAGE) {
        if (p.norm && p.x_bf16) {                  // same per-row reduction as above, same bits
            for (int row = wave; row < p.M; row += NW) {
                float q = 0.f;
                for (int c = lane * 8; c < p.K; c += 512) {
                    const u32x4 t = *reinterpret_cast<const u32x4*>(&xs[row * XS + c]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a = __uint_as_float(t[e] << 16), b2 = __uint_as_float(t[e] & 0xffff0000u);
                        q += a * a + b2 * b2;
                    }
                }
                q = wave_sum64_dpp(q);
                if (lane == 0) rsum[row * NW] = q;
            }
        }
    }
#endif
    // ---- 5. cross-wave combine (fixed order) and epilogue by wave 0
    if constexpr (ALIAS) __syncthreads();          // every wave is done reading xs before red overwrites it
#pragma unroll
    for (int s = 0; s < SPW; ++s)
#pragma unroll
        for (int m = 0; m < MT; ++m) red[((wave * SPW + s) * MT + m) * 64 + lane] = acc[s][m];
    __syncthreads();
    if (wave != 0) return;
#if QTTS_SKINNY_LATE_NORM
    if constexpr (STAGE) {
        if (p.norm && p.x_bf16) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int row = m * 16 + lj;
                if (row < p.M) rstd_l[m] = rsqrtf(rsum[row * NW] / (float)p.K + p.eps);
            }
        }
    }
#endif

    // v[s][r] = out[row = m*16 + lj][feature = (strip0+s)*16 + lq*4 + r]
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row = m * 16 + lj;
        f32x4 v[SPW];
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            f32x4 t = red[((0 * SPW + s) * MT + m) * 64 + lane];
#pragma unroll
            for (int w2 = 1; w2 < NW; ++w2) t += red[((w2 * SPW + s) * MT + m) * 64 + lane];
            v[s] = t * rstd_l[m] + biasv[s];
        }
        if (row >= p.M || lq * 4 >= FS) continue;
#if QTTS_SKINNY_GU8
        // A/B variant: a 16-feature strip = 8 gate + 8 up features (talker_engine.hip packs it so); lanes lq < 2 hold the gate
        // sums, lanes + 32 (same row, lq + 2) the matching up sums, fetched from the same LDS partials.
        if (p.act == ACT_SWIGLU) {
            if constexpr (SPW == 1 && FS == 16) {
                if (lq < 2) {
                    f32x4 tu = red[((0 * SPW + 0) * MT + m) * 64 + lane + 32];
#pragma unroll
                    for (int w2 = 1; w2 < NW; ++w2) tu += red[((w2 * SPW + 0) * MT + m) * 64 + lane + 32];
                    const f32x4 vu = tu * rstd_l[m] + biasv[0];
                    const int col = blockIdx.x * 8 + lq * 4;
                    f32x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (v[0][r] / (1.f + expf(-v[0][r]))) * vu[r];
                    o += resv[0][m];
                    if (p.out_bf16) {
                        uint2 h; h.x = pack_bf16(o[0], o[1]); h.y = pack_bf16(o[2], o[3]);
                        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.out) + (size_t)row * p.ldo + col) = h;
                    } else *reinterpret_cast<f32x4*>(p.out + (size_t)row * p.ldo + col) = o;
                }
            }
        } else
#endif
        if (p.act == ACT_SWIGLU) {
            if constexpr (SPW == 2) {
                const int col = blockIdx.x * 16 + lq * 4;
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (v[0][r] / (1.f + expf(-v[0][r]))) * v[1][r];
                o += resv[0][m];
                if (p.out_bf16) {
                    uint2 h; h.x = pack_bf16(o[0], o[1]); h.y = pack_bf16(o[2], o[3]);
                    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.out) + (size_t)row * p.ldo + col) = h;
                } else *reinterpret_cast<f32x4*>(p.out + (size_t)row * p.ldo + col) = o;
            }
        } else {
#pragma unroll
            for (int s = 0; s < SPW; ++s) {
                const int col = (strip0 + s) * FS + lq * 4;
                const f32x4 o = v[s] + resv[s][m];
                if (p.out16) {
                    uint2 h; h.x = pack_bf16(o[0], o[1]); h.y = pack_bf16(o[2], o[3]);
                    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.out16) + (size_t)row * p.ldo + col) = h;
                }
                if (p.out_bf16) {
                    uint2 h; h.x = pack_bf16(o[0], o[1]); h.y = pack_bf16(o[2], o[3]);
                    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.out) + (size_t)row * p.ldo + col) = h;
                } else *reinterpret_cast<f32x4*>(p.out + (size_t)row * p.ldo + col) = o;
            }
        }
    }
}

// The staged kernel needs the whole x (M x K bf16) in LDS: M <= 16 up to K = 7096, M <= 32 up to K = 2344 (red aliases xs).
bool skinny_can_stage(int M, int K, bool bf16) {
    if (!bf16 || K % 256 != 0) return false;
    if (M <= 16) return (size_t)M * (K + 8) * 2 <= 111 * 1024;
    return M <= 32 && (size_t)M * (K + 8) * 2 <= 150 * 1024;
}

template <bool BF16, int MT, int SPW, int NW, bool STAGE, int FS>
static void launch_one(const SkinnyParams& p, hipStream_t st) {
    const int grid = p.N / (FS * SPW);
    const size_t red_b = (size_t)NW * SPW * MT * 64 * 16, rs_b = (size_t)16 * MT * NW * sizeof(float);
    const size_t xs_b = STAGE ? (((size_t)p.M * (p.K + 8) * 2 + 1023) / 1024) * 1024 : 0;
    const size_t lds = rs_b + ((STAGE && MT > 1) ? std::max(red_b, xs_b) : red_b + xs_b);
    QTTS_REQUIRE(lds <= 160 * 1024, QTTS_ERR_LIMIT, "skinny: LDS budget exceeded");
    auto kern = skinny_kernel<BF16, MT, SPW, NW, STAGE, FS>;
    static bool attr_set = false;          // one flag per instantiation
    if (lds > 48 * 1024 && !attr_set) {
        QTTS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, p);
}

template <bool BF16, int MT>
static void launch_mt(const SkinnyParams& p, int spw, int nw, bool stage, int fs, hipStream_t st) {
    if constexpr (BF16 && MT == 1) {
        if (stage) {
            if (spw == 2) { if (nw == 8) launch_one<true, 1, 2, 8, true, 16>(p, st); else launch_one<true, 1, 2, 4, true, 16>(p, st); }
            else if (fs == 16) { if (nw == 8) launch_one<true, 1, 1, 8, true, 16>(p, st); else launch_one<true, 1, 1, 4, true, 16>(p, st); }
            else if (fs == 8) { if (nw == 8) launch_one<true, 1, 1, 8, true, 8>(p, st); else launch_one<true, 1, 1, 4, true, 8>(p, st); }
            else { if (nw == 8) launch_one<true, 1, 1, 8, true, 4>(p, st); else launch_one<true, 1, 1, 4, true, 4>(p, st); }
            return;
        }
    }
    if constexpr (BF16 && MT == 2) {
        if (stage) {
            QTTS_REQUIRE(fs == 16, QTTS_ERR_ARG, "skinny: narrow strips (fs < 16) are only built for the staged bf16 M<=16 kernel");
            if (spw == 2) { if (nw == 8) launch_one<true, 2, 2, 8, true, 16>(p, st); else launch_one<true, 2, 2, 4, true, 16>(p, st); }
            else { if (nw == 8) launch_one<true, 2, 1, 8, true, 16>(p, st); else launch_one<true, 2, 1, 4, true, 16>(p, st); }
            return;
        }
    }
    QTTS_REQUIRE(fs == 16, QTTS_ERR_ARG, "skinny: narrow strips (fs < 16) are only built for the staged bf16 M<=16 kernel");
    if (nw == 8) { if (spw == 2) launch_one<BF16, MT, 2, 8, false, 16>(p, st); else launch_one<BF16, MT, 1, 8, false, 16>(p, st); }
    else         { if (spw == 2) launch_one<BF16, MT, 2, 4, false, 16>(p, st); else launch_one<BF16, MT, 1, 4, false, 16>(p, st); }
}

void launch_skinny(const SkinnyParams& p, bool bf16, hipStream_t st) {
    const int KT = bf16 ? 32 : 16;
    const int fs = p.fs ? p.fs : 16;
    QTTS_REQUIRE(fs == 16 || fs == 8 || fs == 4, QTTS_ERR_ARG, "skinny: fs must be 16, 8 or 4");
    QTTS_REQUIRE(p.N % 16 == 0, QTTS_ERR_ARG, "skinny: N % 16");
    QTTS_REQUIRE(p.K % KT == 0, QTTS_ERR_ARG, "skinny: K must be a multiple of the k-tile");
    QTTS_REQUIRE(p.M >= 1 && p.M <= 64, QTTS_ERR_LIMIT, "skinny: 1 <= M <= 64");
    QTTS_REQUIRE(p.ldx % 4 == 0 && p.ldo % 4 == 0, QTTS_ERR_ARG, "skinny: ldx/ldo % 4");
    const bool stage = skinny_can_stage(p.M, p.K, bf16);
    QTTS_REQUIRE(!p.norm || stage || p.ss_in, QTTS_ERR_ARG, "skinny: norm without LDS staging needs ss_in (row sums of squares)");
    QTTS_REQUIRE(!p.x_bf16 || stage, QTTS_ERR_ARG, "skinny: bf16 x needs the staged kernel");
    QTTS_REQUIRE(!p.out_bf16 || bf16, QTTS_ERR_ARG, "skinny: bf16 output only in bf16 mode");
    int spw = 1;
    if (p.act == ACT_SWIGLU) {
        QTTS_REQUIRE(p.N % 32 == 0 && fs == 16, QTTS_ERR_ARG, "skinny: swiglu needs N % 32 and fs == 16");
#if QTTS_SKINNY_GU8
        QTTS_REQUIRE(!p.bias, QTTS_ERR_ARG, "skinny: the 8+8 swiglu variant has no bias path");
        spw = 1;
#else
        spw = 2;
#endif
    } else if (fs == 16 && p.N / 16 >= 1024 && (p.N / 16) % 2 == 0) spw = 2;
    const int nw = (p.K / KT >= 16) ? 8 : 4;     // 8 waves split K unless K is tiny
    const int mt = p.M <= 16 ? 1 : (p.M <= 32 ? 2 : 4);
    if (bf16) {
        if (mt == 1) launch_mt<true, 1>(p, spw, nw, stage, fs, st);
        else if (mt == 2) launch_mt<true, 2>(p, spw, nw, stage, fs, st);
        else launch_mt<true, 4>(p, spw, nw, false, fs, st);
    } else {
        if (mt == 1) launch_mt<false, 1>(p, spw, nw, false, fs, st);
        else if (mt == 2) launch_mt<false, 2>(p, spw, nw, false, fs, st);
        else launch_mt<false, 4>(p, spw, nw, false, fs, st);
    }
    QTTS_CHECK_HIP(hipGetLastError());
}

size_t skinny_packed_bytes(int N, int K, bool bf16) { return (size_t)N * K * (bf16 ? 2 : 4); }

// Pack W[N][K] (row-major f32), optionally scaled per column by g[K] (folded RMSNorm weight), into the tile layout
// [N/fs strips][K/KT k-tiles][4 k-slices][fs features][16 B].
void pack_skinny_weight(const float* W, int N, int K, bool bf16, void* out_host, const float* g, int fs) {
    const int KT = bf16 ? 32 : 16;
    const int nkt = K / KT, strips = N / fs;
    parallel_for(strips, [&](int64_t s0, int64_t s1) {
        for (int64_t s = s0; s < s1; ++s)
            for (int kt = 0; kt < nkt; ++kt)
                for (int q = 0; q < 4; ++q)
                    for (int i = 0; i < fs; ++i) {
                        const size_t tile = ((size_t)s * nkt + kt) * (fs * 4) + q * fs + i;
                        const int k0 = kt * KT + q * (bf16 ? 8 : 4);
                        const float* src = W + (size_t)(s * fs + i) * K + k0;
                        if (bf16) {
                            bf16_t* d = reinterpret_cast<bf16_t*>(out_host) + tile * 8;
                            for (int e = 0; e < 8; ++e) d[e] = f32_to_bf16(g ? src[e] * g[k0 + e] : src[e]);
                        } else {
                            float* d = reinterpret_cast<float*>(out_host) + tile * 4;
                            for (int e = 0; e < 4; ++e) d[e] = g ? src[e] * g[k0 + e] : src[e];
                        }
                    }
    });
}

}  // namespace qtts
