// kernels.h -- launch interfaces of the HIP kernels (all gfx950, wave64).
#pragma once
#include "common.h"

namespace qtts {

enum Act { ACT_NONE = 0, ACT_GELU = 1, ACT_SWIGLU = 2, ACT_SNAKE = 3, ACT_SILU = 4,
           ACT_SWIGLU8 = 5 };   // decode GEMM at batch <= 8 only: W rows interleaved [8 gate | 8 up] per 16-feature strip (see SkinnyParams::act)

// --------------------------------------------------------------------------------- gemm_tap.hip
struct GemmTapParams {
    const float* A; int lda;     // activations, channel-last [rows][lda]
    int M;                       // output rows
    int T;                       // rows per sequence (tap validity); M % T == 0
    const void* W;               // [taps][N][K], float or bf16
    int N, K, taps;
    int shift[8];                // per-tap row shift (<= 0)
    const float* bias;           // [N] or null
    const float* scale;          // [N] or null  (LayerScale / ConvNeXt gamma): v = scale * act(acc + bias)
    const float* res; int ldr;   // residual [M][ldr] or null, added last
    const float* snake_ea;       // ACT_SNAKE: exp(alpha)[N]
    const float* snake_ib;       // ACT_SNAKE: 1/(exp(beta)+1e-9)[N]
    int act;
    float* C; int ldc;           // ACT_SWIGLU writes N/2 columns
    // ---- bf16 mode, round 2 (gemm_tap2_kernel): activations that are ONLY a GEMM input travel as bf16
    const void* A16;             // bf16 [rows][lda] alternative to A (written by a producer's C16): selects the tap-reuse kernel
    void* C16;                   // optional bf16 output [M][ldc16] = bf16(act16(v)), v = the fp32 result after act / scale / res
    int ldc16;
    int act16;                   // ACT_NONE | ACT_SNAKE (with snake16_*): the NEXT consumer's SnakeBeta folded into this epilogue
    const float* snake16_ea; const float* snake16_ib;
    int snake16_period;          // > 0: the act16 parameters repeat with this period over the N columns (transposed conv: N = r * Cout)
    int vec4;                    // set by launch_gemm_tap: N / leading dimensions / pointers allow 4-column vector epilogue accesses
    // ---- bf16 residual stream (codec decoder blocks): the 1x1 convolutions of the residual units are HBM-bound and most of their
    // bytes were the fp32 residual tile read + written (profiles/r02_tstamp_codec_gemm.md)
    const void* res16; int ldres16;   // residual as bf16 [M][ldres16] (alternative to res)
    void* R16; int ldR16;             // optional bf16 output of v BEFORE act16 = the residual stream of the next unit
};
void launch_gemm_tap(const GemmTapParams& p, bool bf16, hipStream_t st);

// --------------------------------------------------------------------------------- resunit.hip
// One fused kernel per residual unit of the codec decoder's blocks (bf16 mode, C = 96 | 192):
//   out = x + conv1x1(SnakeBeta_2(conv7_dil(A16))) with A16 = bf16(SnakeBeta_1(x)) written by the unit's producer (V2:619-635).
struct ResUnitParams {
    const void* A16; int lda;            // bf16 [M][lda]: the unit's activated input
    const float* res; int ldr;           // fp32 [M][ldr]: x, the residual ...
    const void* res16;                   // ... or the same as bf16 [M][ldr] (the bf16 residual stream inside a decoder block)
    int M, T, dil, Cch;                  // rows, rows per sequence (causal left padding), dilation of the 7-tap conv, channels
    const void* W1p; const float* b1;    // conv7: weights packed by pack_resunit_weight(permute_cols = true), bias [C]
    const float* ea2; const float* ib2;  // SnakeBeta_2: exp(alpha)[C], 1 / (exp(beta) + 1e-9)[C]
    const void* W2p; const float* b2;    // conv1x1: packed (permute_cols = false), bias [C]
    float* C; int ldc;                   // fp32 output (the residual stream) or null
    void* R16;                           // bf16 output of the same values [M][ldc] (the bf16 residual stream) or null
    void* C16; int ldc16;                // bf16 output for the next GEMM consumer or null, ...
    const float* ea16; const float* ib16;    // ... with that consumer's SnakeBeta folded in (null: plain bf16 copy)
};
bool resunit_supported(int C);
size_t resunit_packed_elems(int C, int taps);
void pack_resunit_weight(const float* W /* [taps][C][C] */, int C, int taps, bool permute_cols, uint16_t* out);
void launch_resunit(const ResUnitParams& p, hipStream_t st);

// --------------------------------------------------------------------------------- skinny.hip
// Weight-streaming GEMM for decode: out[M <= 16*MT][N] = x[M][K] . W[N][K]^T, W pre-packed into
// 1-KiB MFMA-operand tiles (see pack_skinny_weight).
struct SkinnyParams {
    const float* x; int ldx; int M;
    int x_bf16;                  // 1: x points to bf16 [M][ldx] (the producer's bf16 copy); each wave loads its own MFMA B fragments
    int out_bf16;                // 1: out is written as bf16 [M][ldo] (feeds an x_bf16 consumer)
    void* out16;                 // optional second output: bf16 copy [M][ldo] of the fp32 `out` (hidden state for the next norm'd GEMM)
    const void* Wp;              // packed tiles [N/16][K/KT][64 lanes][16 B] (RMSNorm weight already folded in)
    int N, K;
    int fs;                      // output features per strip the weights were packed with: 16 (default, 0) | 8 | 4
    int norm;                    // 1: out = rstd[m] * (x . W'^T) with rstd = rsqrt(mean_k x^2 + eps)
    const float* ss_in;          // fp32 kernel only: sum_k x[m][k]^2 per row [M] (the bf16 kernel takes it on the matrix pipe)
    float eps;
    const float* bias;           // [N] or null
    const float* res; int ldr;   // residual [M][ldr] or null
    float* out; int ldo;
    int act;                     // ACT_NONE | ACT_SWIGLU (strip pairs gate/up -> N/2 output columns) | ACT_SWIGLU8 (round 3, skinny8_kernel:
                                 // ONE strip = 8 gate + 8 up rows of the same 8 output columns -> N/16 workgroups instead of N/32)
    // fp32 batch <= 8 kernel only (round 4): split-K with the combine in the consumer's prologue
    int ksplit;                  // 2: producer -- workgroup (strip, half) writes its sums to out + half * part_stride; half 0 adds `res` (plain GEMM only)
    size_t part_stride;          //    floats between the two halves' [M][ldo] partial buffers
    const float* xp;             // consumer: x is formed as xp[0] + xp[1] (= residual + half 0, half 1), at xp and xp + xp_stride, [M][ldx]; `x` is not read
    size_t xp_stride;
    float* x_out;                //    ... and workgroup 0 writes the combined rows here (outside the halves)
    // bf16, batch 17..32, plain GEMM without norm (round 6: skinny2_ks_kernel): K split over the workgroups of a 32-feature strip group, the
    // partial sums handed to the group's LAST workgroup as tagged granules (granule.h) and added there in k order
    float* ks_part;              // granule workspace (null: no split) ...
    size_t ks_part_bytes;        // ... and its size
    const int* ks_serial;        // device: the frame serial the tags derive from (null: serial 1 -- tests)
    int ks_slot;                 // this launch's slot among the frame's split launches, < 256: tag = serial << 8 | slot
    int ks_pause;                // x 64 clocks: the reducer's wait before its first read of the others' granules
    int* ks_err; int* ks_latch;  // raised by a reducer that gave up (the engine's flag, the generation's stop flag)
    const int* done_flag;        // optional device flag: when non-zero the kernel exits early
    int ablate;                  // `ablate` build variant only (-DQTTS_ABLATE; must be 0 in the product build): 1 no done check, 2 no x loads, 4 no epilogue loads, 8 no weight loads
};
void launch_skinny(const SkinnyParams& p, bool bf16, hipStream_t st);
void skinny_set_launch_events(hipEvent_t start, hipEvent_t stop);   // (bench.py's roofline leg: time the NEXT launch on its own; null = off)
bool skinny_takes_bf16_x(int M, int K, bool bf16);   // bf16 mode: any M <= 64, K % 32 == 0
bool skinny_f32_inline_norm(int M, int K);
bool skinny_f32_splitk_takes(int M, int K_producer, int K_consumer);   // fp32 mode: producer may split K in two, its consumer combines            // fp32 mode: the batch <= 8 kernel takes the RMSNorm row statistics itself (no ss_in)
bool skinny_ksplit_takes(int M, int N, int K, int fs);   // bf16, batch 17..32: the split-K kernel has an instantiation for this (strip width, shape)
bool skinny_swiglu8_takes(int K);                     // ACT_SWIGLU8 (batch <= 8 kernel): K = 1024 | 2048 | 3072 | 6144
size_t skinny_packed_bytes(int N, int K, bool bf16);
// Pack W[N][K] (row-major f32), optionally scaled per input column by g[K], into the streaming tile layout;
// gate/up interleaving is the caller's.
void pack_skinny_weight(const float* W, int N, int K, bool bf16, void* out_host, const float* g = nullptr, int fs = 16);

// --------------------------------------------------------------------------------- elementwise.hip
void launch_rmsnorm(const float* x, int ldx, const float* w, float eps, float* y, int ldy, int rows, int C,
                    hipStream_t st);
// the same with a bf16 output [rows][ldy] (the only consumer is a bf16 GEMM: identical rounding, half the bytes it reads)
void launch_rmsnorm16(const float* x, int ldx, const float* w, float eps, void* y16, int ldy, int rows, int C, hipStream_t st);
void launch_snake(const float* x, const float* ea, const float* ib, float* y, int64_t rows, int C, hipStream_t st);
// y[t][c] = LayerNorm_c(sum_k w[c][k] x[t-6+k][c] + b[c]) (ConvNeXt dwconv k=7 + LN eps)
void launch_dwconv_ln(const float* x, const float* w7, const float* b, const float* ln_w, const float* ln_b,
                      float eps, float* y, int rows, int T, int C, hipStream_t st);
// out[bt][0:vq] = tab0[code0], out[bt][vq:2vq] = sum_{q>=1} tab_q[code_q]  (codes (B,Q,T) or (B,T,Q))
void launch_rvq_gather(const int64_t* codes, int B, int Q, int T, int64_t stride_b, int64_t stride_q,
                       int64_t stride_t, int t0, int Tc, const float* tables, int bins, int vq, float* out,
                       int* err /* set to 1 when a code >= bins is seen */, hipStream_t st);
// final conv (C -> 1, k = 7, causal) + clamp; x already snake-activated, channel-last
void launch_final_conv(const float* x, const float* w /*[7][C]*/, float bias, float* wav, float* pre_clamp,
                       int64_t rows, int64_t T, int C, int64_t out_stride_b, int64_t skip, hipStream_t st);
// bf16 mode: x = the producer's bf16 copy with the final SnakeBeta applied
void launch_final_conv16(const bf16_t* x, const float* w /*[7][C]*/, float bias, float* wav, float* pre_clamp,
                         int64_t rows, int64_t T, int C, int64_t out_stride_b, int64_t skip, hipStream_t st);
// in-place rotate-half RoPE on the q and k parts of a fused qkv buffer (codec transformer: no q/k norm)
void launch_rope_inplace(float* qkv, int ld, int rows, int T, int n_heads_total /* q + k heads */, int hd,
                         const float* inv_freq, hipStream_t st);

// ---- streaming (state-carrying) codec decode helpers
// dst[b] = [state[b] (h rows) | src[b][skip : skip + n]] per sequence (channel-last rows of C floats, C % 4 == 0);
// src has src_T rows per sequence, dst h + n.  state may be null when h == 0 (plain compaction).
void launch_stage_rows(const float* src, int src_T, int skip, int n, const float* state, int h, float* dst, int B, int C,
                       hipStream_t st);
// state[b] = the last h rows of buf[b] (Tp rows per sequence)
void launch_save_tail(const float* buf, int Tp, float* state, int h, int B, int C, hipStream_t st);
// launch_rope_inplace with positions pos0 + (row % T)
void launch_rope_offset(float* qkv, int ld, int rows, int T, int pos0, int n_heads_total, int hd, const float* inv_freq,
                        hipStream_t st);

// ---- codec ENCODER helpers (Mimi encode) -- encoder_kernels.hip
void launch_elu(const float* x, float* y, int64_t n, hipStream_t st);
// first SEANet conv: wav (B, L) -> out channel-last [B*L][C], causal k taps, zero left pad (C_in = 1 cannot be a GEMM)
void launch_conv_in1(const float* wav, const float* w /*[C][k]*/, const float* bias, float* out, int B, int L, int C, int k,
                     hipStream_t st);
void launch_layernorm(const float* x, int ldx, const float* w, const float* b, float eps, float* y, int ldy, int rows, int C,
                      hipStream_t st);
// dst[b][i] = src[b][clamp(i - left, 0, T-1)] (replicate) or 0 outside [left, left+T) (zeros); rows of C floats
void launch_pad_rows(const float* src, int T, int left, int right, int replicate, float* dst, int B, int C, hipStream_t st);
// nearest codebook entry by argmin_j (enorm[j] - 2*scores[row][j]) (lowest index wins ties), then residual update
// r[row] -= table[idx]; codes_out[b*stride_b + t] = idx   (row = b*T + t)
void launch_vq_argmin_update(const float* scores, int bins, const float* enorm, const float* table, int D, float* r,
                             int64_t* codes_out, int64_t stride_b, int B, int T, hipStream_t st);

// ---- speaker encoder (ECAPA-TDNN + mel front end) -- speaker_kernels.hip
enum RowAct { ROWACT_NONE = 0, ROWACT_RELU = 1, ROWACT_RELU_TANH = 2, ROWACT_SIGMOID = 3, ROWACT_LOG_CLAMP = 4 };
// out[b][r][j] = wav[b][reflect(r*hop + j - pad)] for r < R (reflect = torch 'reflect', no edge repeat)
void launch_reflect_rows_1d(const float* wav, int S, int pad, int R, int hop, float* out, int B, hipStream_t st);
// out[row][f] = sqrt(re^2 + im^2 + 1e-9) for f < nb (re = y[row][f], im = y[row][nb + f]), 0 for nb <= f < Kp
void launch_magnitude_pad(const float* y, int ldy, int nb, float* out, int Kp, int64_t rows, hipStream_t st);
// dst[b][i][c] = src1[b][reflect(i - p)][c] (+ src2[...]) for i in [0, T + 2p): nn.Conv1d(padding="same", padding_mode="reflect")
void launch_reflect_pad_add_rows(const float* src1, int ld1, const float* src2, int ld2, int T, int p, int C, float* dst, int B,
                                 hipStream_t st);
// dst[(b*n + t)*ldd + c] = act(src[(b*Tsrc + skip + t)*lds + c]) for c < C
void launch_copy_act_rows(const float* src, int lds, int Tsrc, int skip, int n, int C, int act, float* dst, int ldd, int B,
                          hipStream_t st);
// mean[b][c] = sum_t w x, sd[b][c] = sqrt(clamp(sum_t w (x - mean)^2, 1e-12)); w = att[b][t][c] or 1/T when att == null
void launch_col_stats(const float* x, int ldx, const float* att, int T, int C, float* mean, float* sd, int ld_out, int B,
                      hipStream_t st);
// out[b][t][c] = h[b][t][c] * gate[b][c] + r[b][t][c]
void launch_scale_add_rows(const float* h, int ldh, const float* gate, const float* r, int ldr, float* out, int ldo, int T, int C,
                           int B, hipStream_t st);
// out[b][t] = [x[b][t] | mean[b] | sd[b]]  (3C columns)
void launch_concat_stats(const float* x, int ldx, const float* mean, const float* sd, int T, int C, float* out, int B, hipStream_t st);
// softmax over t for every (b, c), in place
void launch_softmax_time(float* a, int T, int C, int B, hipStream_t st);

// --------------------------------------------------------------------------------- attention.hip
// Generic row attention over a fused qkv buffer (prefill + codec transformer).
struct AttnRowsParams {
    const float* qkv; int ld; int q_off, k_off, v_off;   // column offsets of the q / k / v blocks
    int B, T, nh, nkv, hd;
    int window;                  // 0 = full causal; else keys in (tq - window, tq]
    const int* n_pad;            // optional [B]: keys < n_pad masked, query rows < n_pad skipped
    float* out; int ldo;
    void* out16 = nullptr;       // optional: write bf16 [rows][ldo] here INSTEAD of the fp32 `out` (the consumer is a bf16 GEMM)
};
void launch_attn_rows(const AttnRowsParams& p, hipStream_t st);

// Prefill: per-head RMSNorm + RoPE in place on q,k of the fused qkv buffer, and K/V append to the cache.
struct KvCache {
    void* k; void* v;            // pools [layer][page][kvh][16][hd] (float or bf16)
    const int* page_table;       // [B][pages_per_seq]
    int pages_per_seq, n_pages, nkv, hd;
    int bf16;
    int contig;                  // 1: page_table[b][i] == b*pages_per_seq + i (skip the indirection load)
    int vt;                      // 1: V pages are stored TRANSPOSED, [layer][page][kvh][hd][16] -- the A-operand image of the PV product on
                                 // the matrix pipe (attn_tk16_kernel, bf16 talker cache only); K pages stay [16][hd]
};
struct QkNormRopeParams {
    float* qkv; int ld; int B, T, nh, nkv, hd;
    const float* qw; const float* kw; float eps;
    const float* inv_freq;       // [hd/2]
    const int* n_pad;            // [B]
    KvCache kv; int layer;
};
void launch_qknorm_rope_store(const QkNormRopeParams& p, hipStream_t st);

// Decode: fused q/k RMSNorm + RoPE + KV append + attention for n_new (1|2) new tokens per sequence.
struct AttnDecodeParams {
    const float* qkv; int ld;    // rows t*B + b
    int B, n_new, nh, nkv, hd;
    const float* qw; const float* kw; float eps;
    const float* inv_freq;
    const int* n_pad;            // optional [B]
    const int* len_dev;          // device int: KV length before this step (talker) or null
    int len_static;              // used when len_dev == null (code predictor)
    KvCache kv; int layer;
    float* out; int ldo;         // rows t*B + b, columns nh*hd
    int out_bf16;                // 1: write bf16 (consumed by an x_bf16 GEMM)
    int max_len;                 // LDS score capacity
    const int* done_flag;
    // split-KV (talker, long sequences): nsplit > 1 workgroups per (sequence, kv head), each over max_len / nsplit keys, partial
    // results in `part` [B*nkv][nsplit][GQ][128 + 2] fp32 (numerator | max | denominator), merged by a second tiny kernel
    int nsplit; float* part;
    // optional [rope_cs_n][2][64]: cos | sin of pos * inv_freq for positions < rope_cs_n, filled by launch_rope_table with the very
    // expression the kernels evaluate (bit-identical); attn_cp (static position, known at launch) requests its row at kernel entry
    // instead of running sinf / cosf behind the arrival of the qkv row
    const float* rope_cs; int rope_cs_n;
};
void launch_attn_decode(const AttnDecodeParams& p, hipStream_t st);
void launch_rope_table(const float* inv_freq, int n_pos, float* out, hipStream_t st);
inline size_t attn_part_floats(int B, int nkv, int nsplit, int gq) { return (size_t)B * nkv * nsplit * gq * 130; }

// The code predictor's passes >= 1, attention + o-projection in one launch (attention.hip: cp_attn_o_kernel).  `a` as for
// launch_attn_decode (its out / ldo / out_bf16 are not used); the o-projection's operator packed by pack_skinny_weight(bf16, fs = 16).
struct CpAttnOParams {
    AttnDecodeParams a;
    const void* Wo;               // [H / 16 strips][nh * hd / 32 k-tiles][4][16][8] bf16
    const float* res;             // residual rows [B][H] (may be `out`)
    float* out;                   // hidden rows [B][H] fp32
    unsigned short* out16;        // optional bf16 copy [B][H]
    float* part;                  // scratch [nkv][8][H] granules of 8 B {fp32 partial sum, launch tag}: zero at engine creation (tag 0 is never used)
    const int* serial;            // device word that changes between two launches with the same `slot` (the engine: +1 per frame step)
    int slot;                     // < 128, different for launches that share a value of *serial (the engine: pass * layers + layer):
                                  // the launch tag = (*serial << 7) | slot differs from that of the launches before it
    int phase;                    // 2: the whole kernel.  0 / 1 (host emulator only, with Wqkv): the q|k|v strips only / everything after them
    int* err;                     // optional device flag, set if a reducer gave up waiting (never in a correct run)
    int* done_latch;              // optional: the generation's stop latch, SET by a consumer that gives up -- every later kernel of the frame
                                  // chain honours it, so a launch that lost its producers costs ONE give-up, not one per launch
    int first_pause, poll_step;   // x 64 clocks: a consumer's wait before its first read of other workgroups' granules, and between the
                                  // reads it keeps in flight after that (the engine passes 16 and 4: ~0.4 and ~0.1 us)
    // optional: the layer's q|k|v GEMM in front, in the same launch (workgroup i = 16-feature strip i of it; `a.qkv` is then unused)
    const void* Wqkv;             // packed by pack_skinny_weight(bf16, fs = 16, RMSNorm weight folded): [(nh + 2 nkv) * hd / 16][K / 32][4][16][8] bf16
    const unsigned short* x16;    // the layer's input rows [B][ldx16] bf16 (un-normalised: the kernel takes the row variances itself)
    int ldx16, K; float eps_in;
    float* qkv_gran;              // scratch [8][a.ld] granules of 8 B {fp32 q|k|v value, launch tag}: zero at engine creation
    int H;
};
bool cp_attn_o_takes(const AttnDecodeParams& a, int H);
void launch_cp_attn_o(const CpAttnOParams& P, hipStream_t st);
// Residency of the fused launch: workgroups of cp_attn_o_kernel (256 threads, its static LDS, its registers) one compute unit holds at
// once -- hipOccupancyMaxActiveBlocksPerMultiprocessor, the minimum over the instantiations an engine can launch.  A launch whose
// workgroups wait for each other's granules is only correct when ALL of them are resident: the engine admits a fused launch per
// device only while (fused engines on the device) x grid <= this x the device's compute units (talker_engine.hip: fused_admit).
int cp_attn_o_blocks_per_cu(bool f32 = false);
int cp_attn_o_grid(int H);                                        // workgroups of one launch
void cp_attn_o_set_launch_events(hipEvent_t start, hipEvent_t stop);   // bench.py's roofline leg: time the NEXT launch on its own (as skinny_set_launch_events)

// --------------------------------------------------------------------------------- cp_mlp.hip
// The code predictor's MLP of a layer (RMSNorm -> gate|up -> SwiGLU -> down -> + residual) as ONE launch, batch <= 8, bf16 (round 5).
struct CpMlpParams {
    int f32;                      // 1: the exact parity mode -- fp32 operators, `x16` points to fp32 rows, no bf16 copy (out16 null)
    const void* Wgu;              // pack_cp_mlp_gu: [8 J workgroups][H / 32][4][2 ACT rows][8] bf16, RMSNorm weight folded (J = H / 32, ACT = I / (8 J))
    const void* Wd;               // down-projection, pack_skinny_weight(bf16, fs = 16): [H / 16][I / 32][4][16][8]
    const unsigned short* x16;    // the MLP's input rows [B][ldx16] bf16 (the hidden state after attention, un-normalised)
    int ldx16; float eps;
    const float* res;             // residual rows [B][H] fp32 (may be `out`)
    float* out;                   // hidden rows [B][H] fp32
    unsigned short* out16;        // optional bf16 copy [B][H]
    float* act_gran;              // scratch [8 XCDs][8 rows][I / 16] granules {2 x bf16 act, tag}: zero at engine creation
    float* part;                  // scratch [8 XCDs][8 rows][H] granules {fp32 partial sum, tag}: zero at engine creation
    const int* serial; int slot;  // launch tag = (*serial << 7) | slot, as for cp_attn_o (its own buffers: its own slot numbering)
    int phase;                    // 3: the whole kernel.  0 / 1 / 2 (host emulator, or a test): phase A / B / C alone
    int* err; int* done_latch;    // give-up flag and the generation's stop latch (set by a consumer that gives up)
    const int* done_flag;         // optional: when non-zero the kernel exits early
    int first_pause, poll_step;   // x 64 clocks, as for cp_attn_o
    int pause_c;                  // x 64 clocks: the reducer's wait before its first read of the partial sums (first_pause: phase B's before the slice)
    int B, H, I;
};
bool cp_mlp_takes(int B, int H, int I);
bool cp_mlp_instantiated(int H, int I, bool bf16);
int cp_mlp_grid(int H);
int cp_mlp_blocks_per_cu(int H, int I, bool bf16);     // residency, as cp_attn_o_blocks_per_cu
size_t cp_mlp_gu_bytes(int H, int I, bool bf16);
void pack_cp_mlp_gu(const float* Wg, const float* Wu, const float* g, int H, int I, bool bf16, void* out_host);
void launch_cp_mlp(const CpMlpParams& P, hipStream_t st);
void cp_mlp_set_launch_events(hipEvent_t start, hipEvent_t stop);
// cp_mlp32.hip (round 6): the same launch for batch 9..32 (two 16-row tiles; bf16 engines).  Same parameter block; its granule buffers have 32
// rows per XCD: act_gran [8][32][I / 16], part [8][32][H] granules.
bool cp_mlp32_takes(int B, int H, int I);
bool cp_mlp32_instantiated(int H, int I);
int cp_mlp32_blocks_per_cu(int H, int I);
size_t cp_mlp32_act_bytes(int I);
size_t cp_mlp32_part_bytes(int H);
void launch_cp_mlp32(const CpMlpParams& P, hipStream_t st);
void cp_mlp32_set_launch_events(hipEvent_t start, hipEvent_t stop);

// --------------------------------------------------------------------------------- cp_layer.hip
// A whole decoder layer of the code predictor (passes >= 1, batch <= 8) as ONE launch (round 6): cp_attn_o's stages, then cp_mlp's, the
// launch boundary between them replaced by a granule hand-off of the hidden rows, the gate|up block requested at entry by LDS-DMA.
// `ao` / `mlp` as for the two launches (`mlp.x16`, `mlp.res`, `ao.out`, `ao.out16` are unused: the rows in between travel as granules
// and stay in the reducers' registers); one tag = (ao.serial, ao.slot) for all five granule buffers.
struct CpLayerParams {
    CpAttnOParams ao;
    CpMlpParams mlp;
    float* hid_gran;              // scratch [hid_slots][8 rows][H / 2] granules {2 x bf16 hidden, tag} (fp32 engines: [..][8][H] {fp32, tag}): zero at engine creation
    int hid_slot;                 // which region of hid_gran this launch uses (the engine: ao.slot -- a region is touched once per frame)
    int hid_mode;                 // 0: every wave polls its whole k quarter; 1: sentinel granules, then one sc1 read; 2: ... one read the L2 may serve
    int pause_h;                  // x 64 clocks: a workgroup's wait before its first read of the hidden rows
    int gu_pace;                  // x 64 clocks between the DMA requests of a workgroup that requests behind the attention stage (0: back to back)
    int gu_when;                  // the gate|up block's LDS-DMA: 2 behind the attention stage; 0 at kernel entry, 1 behind the o-projection operator's requests (A/B)
    int phase;                    // 8: the whole kernel.  0..4 (host emulator, or a test): one stage alone
};
bool cp_layer_takes(const AttnDecodeParams& a, int H, int I);
bool cp_layer_instantiated(int H, int I, bool bf16);
int cp_layer_grid(int H);
int cp_layer_lds_bytes(int H, int I, bool bf16);        // dynamic LDS of one workgroup (the admission account's second resource)
int cp_layer_blocks_per_cu(int H, int I, bool bf16);
void launch_cp_layer(const CpLayerParams& P, hipStream_t st);
void cp_layer_set_launch_events(hipEvent_t start, hipEvent_t stop);

// --------------------------------------------------------------------------------- sampling.hip
struct SampleParams {
    const float* logits; int ld; int V; int B;
    // processors
    const int* generated; int gen_stride;   // [B][gen_stride] tokens so far (talker) or null
    const int* n_generated_dev;              // device count of tokens generated so far (or null -> 0)
    float repetition_penalty;
    int eos; int min_new_tokens;             // eos < 0: no eos handling
    const unsigned char* suppress_mask;      // [V] or null
    int do_sample; int top_k; float top_p; float temperature;
    unsigned long long seed; unsigned int stream_id;   // Philox key / sub-stream (codebook index)
    const unsigned long long* seed_dev;                // if set, the key is read from device memory (graph replay across calls)
    const int* step_dev;                     // device step counter feeding the Philox offset
    // outputs
    int* tok_out; int tok_stride;            // token of row b -> tok_out[b * tok_stride]
    // talker-only bookkeeping (null for the code predictor)
    int* unfinished;                         // [B] in/out
    int* generated_out;                      // append position = n_generated
    int* n_generated_inc;                    // incremented by 1 (single block does it)
    int* done_flag;                          // set when all finished or n_generated >= max_new
    int* final_count;                        // n_generated at the moment done was set
    int max_new_tokens;
    const int* done_in;                      // early exit
    // optional fused gather (code predictor): gather_out[b][:] = gather_emb[token][:] -- the NEXT pass's input row
    // (codec_embedding[j](token), M:1281), so no separate gather kernel sits between sampler and GEMM
    const float* gather_emb; int gather_C; float* gather_out; unsigned short* gather_out16;
    // A/B variant (build.py VARIANTS): second fused gather -- the next pass's layer-0 q|k|v row, tabulated at finalize
    const float* gather2_emb; int gather2_C; float* gather2_out;
};
void launch_sample(const SampleParams& p, hipStream_t st);

}  // namespace qtts
