// TEST INFRASTRUCTURE -- kernel-level entry points of libqtts_hostemu.so (host pointers): tests/test_hostemu.py calls the
// launch_* interfaces of csrc/kernels.h directly, so that single kernels can be checked against numpy / the oracle on the
// SIMT emulator without an engine around them.  Not part of the product ABI (include/qtts.h).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "common.h"
#include "kernels.h"

// ---- test-only entry points (host pointers): kernel-level cases for tests/test_hostemu.py
namespace qtts { extern int g_real_gemm; void launch_gemm_tap_real(const GemmTapParams& p, bool bf16, hipStream_t st); }
// 0: fibers resume in ascending thread order, 1: descending, >= 2: seeded shuffle of the waves (see simt::run_block)
extern "C" void hostemu_set_fiber_order(int order) { simt::M().order = order; }

// 0: the workgroups of a launch run in ascending blockIdx.x order, 1: descending, >= 2: seeded shuffle (see simt::launch)
extern "C" void hostemu_set_block_order(int order) { simt::M().block_order = order; }

extern "C" void hostemu_set_real_gemm(int on) { qtts::g_real_gemm = on ? 1 : 0; }

// the codec's final convolution (C -> 1, k = 7, causal) + clamp: x fp32 [B*T][C] through final_conv_kernel, or x16 bf16 bits through
// final_conv16_kernel (round 3); wav / pre [B][out_stride], the first `skip` samples of every sequence dropped
extern "C" int hostemu_final_conv(const float* x, const unsigned short* x16, const float* w, float bias, float* wav, float* pre, int B, int T,
                                  int C, int out_stride, int skip) {
    try {
        if (x16) qtts::launch_final_conv16(x16, w, bias, wav, pre, (int64_t)B * T, T, C, out_stride, skip, nullptr);
        else qtts::launch_final_conv(x, w, bias, wav, pre, (int64_t)B * T, T, C, out_stride, skip, nullptr);
        return 0;
    } catch (const qtts::Error& e) { return e.code; } catch (...) { return -1; }
}

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

// The bf16-activation tap-reuse kernel (gemm_tap2): A given as bf16 bits [M][lda], optional fp32 output C and / or bf16 output
// C16 (with the consumer's SnakeBeta folded in when ea16 / ib16 are given).
extern "C" int hostemu_gemm_tap16(const unsigned short* A16, int lda, int M, int T, const void* W, int N, int K, int taps,
                                  const int* shift, const float* bias, const float* res, int ldr, const float* snake_ea,
                                  const float* snake_ib, int act, float* C, int ldc, unsigned short* C16, const float* ea16,
                                  const float* ib16) {
    try {
        qtts::GemmTapParams p{};
        p.A16 = A16; p.lda = lda; p.M = M; p.T = T; p.W = W; p.N = N; p.K = K; p.taps = taps;
        for (int i = 0; i < taps && i < 8; ++i) p.shift[i] = shift[i];
        p.bias = bias; p.res = res; p.ldr = ldr; p.snake_ea = snake_ea; p.snake_ib = snake_ib; p.act = act;
        p.C = C; p.ldc = ldc; p.C16 = C16; p.ldc16 = ldc; p.act16 = ea16 ? qtts::ACT_SNAKE : qtts::ACT_NONE;
        p.snake16_ea = ea16; p.snake16_ib = ib16;
        qtts::launch_gemm_tap_real(p, true, nullptr);
        return 0;
    } catch (const qtts::Error& e) { return e.code; } catch (...) { return -1; }
}

// the fused residual unit (resunit.hip): W1 [7][C][C], W2 [C][C] row-major fp32, packed here exactly as the codec engine packs them
extern "C" int hostemu_resunit(const unsigned short* A16, int lda, const float* res, int ldr, int M, int T, int dil, int C,
                               const float* W1, const float* b1, const float* ea2, const float* ib2, const float* W2, const float* b2,
                               float* out, int ldc, unsigned short* C16, const float* ea16, const float* ib16,
                               const unsigned short* res16, unsigned short* R16) {
    try {
        std::vector<uint16_t> w1(qtts::resunit_packed_elems(C, 7)), w2(qtts::resunit_packed_elems(C, 1));
        qtts::pack_resunit_weight(W1, C, 7, true, w1.data());
        qtts::pack_resunit_weight(W2, C, 1, false, w2.data());
        qtts::ResUnitParams p{};
        p.A16 = A16; p.lda = lda; p.res = res16 ? nullptr : res; p.res16 = res16; p.R16 = R16; p.ldr = ldr; p.M = M; p.T = T; p.dil = dil; p.Cch = C;
        p.W1p = w1.data(); p.b1 = b1; p.ea2 = ea2; p.ib2 = ib2; p.W2p = w2.data(); p.b2 = b2;
        p.C = out; p.ldc = ldc; p.C16 = C16; p.ldc16 = ldc; p.ea16 = ea16; p.ib16 = ib16;
        qtts::launch_resunit(p, nullptr);
        return 0;
    } catch (const qtts::Error& e) { qtts::set_last_error(e.what()); return e.code; } catch (...) { return -1; }
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
        p.x = x; p.ldx = ldx; p.M = M; p.Wp = wp.data(); p.N = N; p.K = K; p.fs = 16; p.norm = norm; p.eps = eps;
        p.ss_in = (!bf16 && qtts::skinny_f32_inline_norm(M, K)) ? nullptr : ss.data();     // (the batch <= 8 fp32 kernel must not need it)
        p.bias = bias; p.res = res; p.ldr = ldr; p.out = out; p.ldo = ldo; p.act = act;
        qtts::launch_skinny(p, bf16 != 0, nullptr);
        return 0;
    } catch (const qtts::Error& e) { return e.code; } catch (...) { return -1; }
}

// fp32 batch <= 8, round 4: y = x . W1^T as a split-K PRODUCER (parts[0] = res + first half of K, parts[1] = second half), then
// z = act( rmsnorm(parts[0] + parts[1]) . (g (.) W2)^T ) by the COMBINING consumer, which also writes the combined rows to x_out.
extern "C" int hostemu_skinny_splitk(const float* x, int M, const float* W1, int N1, int K1, const float* res, const float* W2, int N2,
                                     const float* g, float eps, int act, float* parts, float* x_out, float* z, int ldz) {
    try {
        std::vector<unsigned char> w1(qtts::skinny_packed_bytes(N1, K1, false)), w2(qtts::skinny_packed_bytes(N2, N1, false));
        qtts::pack_skinny_weight(W1, N1, K1, false, w1.data(), nullptr, 16);
        qtts::pack_skinny_weight(W2, N2, N1, false, w2.data(), g, 16);
        if (!qtts::skinny_f32_splitk_takes(M, K1, N1)) return -7;
        qtts::SkinnyParams a{};
        a.x = x; a.ldx = K1; a.M = M; a.Wp = w1.data(); a.N = N1; a.K = K1; a.fs = 16; a.out = parts; a.ldo = N1; a.act = qtts::ACT_NONE;
        a.ksplit = 2; a.part_stride = (size_t)8 * N1; a.res = res; a.ldr = N1;          // half 0 = residual + its sums
        qtts::launch_skinny(a, false, nullptr);
        qtts::SkinnyParams b{};
        b.x = parts; b.ldx = N1; b.M = M;                                            // (x is not read: the halves are)
        b.Wp = w2.data(); b.N = N2; b.K = N1; b.fs = 16; b.norm = 1; b.eps = eps; b.out = z; b.ldo = ldz; b.act = act;
        b.xp = parts; b.xp_stride = (size_t)8 * N1; b.x_out = x_out;
        qtts::launch_skinny(b, false, nullptr);
        return 0;
    } catch (const qtts::Error& e) { qtts::set_last_error(e.what()); return e.code; } catch (...) { return -1; }
}

// The bf16 decode GEMM as the frame step launches it: x handed over as the producer's bf16 copy (x_bf16), narrow strips
// (fs = 16 | 8 | 4), optional bf16 shadow output (out16, same leading dimension as out).
extern "C" int hostemu_skinny_bf16x(const float* x, int ldx, int M, const float* W, int N, int K, const float* g, int norm, float eps,
                                    const float* bias, const float* res, int ldr, int act, float* out, int ldo, int fs,
                                    unsigned short* out16) {
    try {
        std::vector<unsigned char> wp(qtts::skinny_packed_bytes(N, K, true));
        qtts::pack_skinny_weight(W, N, K, true, wp.data(), g, fs);
        std::vector<qtts::bf16_t> x16((size_t)M * ldx);
        for (size_t i = 0; i < x16.size(); ++i) x16[i] = qtts::f32_to_bf16(x[i]);
        qtts::SkinnyParams p{};
        p.x = reinterpret_cast<const float*>(x16.data()); p.x_bf16 = 1; p.ldx = ldx; p.M = M; p.Wp = wp.data(); p.N = N; p.K = K;
        p.fs = fs; p.norm = norm; p.eps = eps; p.bias = bias; p.res = res; p.ldr = ldr; p.out = out; p.ldo = ldo; p.act = act;
        p.out16 = out16;
        qtts::launch_skinny(p, true, nullptr);
        return 0;
    } catch (const qtts::Error& e) { return e.code; } catch (...) { return -1; }
}

// Round 6: the bf16 decode GEMM at batch 17..32 through the split-K kernel (skinny2_ks_kernel): `launches` launches in a row on ONE granule
// workspace (0xFF-filled: no tag matches), each with its own slot -- the second and later ones find the previous launch's granules in place.
// *took = 1 when the launcher has an instantiation for the shape (otherwise the launch ran skinny2_kernel).
extern "C" int hostemu_skinny_ksplit(const float* x, int ldx, int M, const float* W, int N, int K, const float* bias, const float* res, int ldr,
                                     float* out, int ldo, int fs, unsigned short* out16, int launches, int serial, int* took) {
    try {
        std::vector<unsigned char> wp(qtts::skinny_packed_bytes(N, K, true));
        qtts::pack_skinny_weight(W, N, K, true, wp.data(), nullptr, fs);
        std::vector<qtts::bf16_t> x16((size_t)M * ldx);
        for (size_t i = 0; i < x16.size(); ++i) x16[i] = qtts::f32_to_bf16(x[i]);
        std::vector<unsigned char> part((size_t)8 << 20, 0xFF);
        int err = 0, latch = 0;
        qtts::SkinnyParams p{};
        p.x = reinterpret_cast<const float*>(x16.data()); p.x_bf16 = 1; p.ldx = ldx; p.M = M; p.Wp = wp.data(); p.N = N; p.K = K;
        p.fs = fs; p.bias = bias; p.res = res; p.ldr = ldr; p.out = out; p.ldo = ldo; p.act = qtts::ACT_NONE; p.out16 = out16;
        p.ks_part = reinterpret_cast<float*>(part.data()); p.ks_part_bytes = part.size(); p.ks_serial = &serial; p.ks_err = &err; p.ks_latch = &latch;
        *took = qtts::skinny_ksplit_takes(M, N, K, fs) ? 1 : 0;
        for (int l = 0; l < launches; ++l) { p.ks_slot = 3 + 7 * l; qtts::launch_skinny(p, true, nullptr); }
        return err || latch ? -7 : 0;
    } catch (const qtts::Error& e) { return e.code; } catch (...) { return -1; }
}

// attn_rows (attention.hip): the prefill / codec-transformer attention over a fused fp32 q|k|v buffer [B][T][(nh + 2 nkv) hd]; out fp32 [B][T][nh hd]
// (or its bf16 image when out16 is given).  The launcher's choice (one query row per wave, or four: round 6) follows the option table.
extern "C" int hostemu_attn_rows(const float* qkv, int B, int T, int nh, int nkv, int hd, const int* n_pad, int window, float* out, unsigned short* out16) {
    try {
        qtts::AttnRowsParams p{};
        p.qkv = qkv; p.ld = (nh + 2 * nkv) * hd; p.q_off = 0; p.k_off = nh * hd; p.v_off = (nh + nkv) * hd;
        p.B = B; p.T = T; p.nh = nh; p.nkv = nkv; p.hd = hd; p.window = window; p.n_pad = n_pad; p.out = out; p.ldo = nh * hd; p.out16 = out16;
        qtts::launch_attn_rows(p, nullptr);
        return 0;
    } catch (const qtts::Error& e) { return e.code; } catch (...) { return -1; }
}

// One launch of sampling.hip's sample_kernel on B rows of logits: HF processors (repetition penalty over `generated`,
// min-new-tokens EOS block, suppress mask), temperature / top-k / top-p, Philox draw keyed by (seed, stream_id, step).
extern "C" int hostemu_sample(const float* logits, int ld, int V, int B, const int* generated, int gen_stride, int n_generated,
                              float repetition_penalty, int eos, int min_new_tokens, const unsigned char* suppress_mask,
                              int do_sample, int top_k, float top_p, float temperature, unsigned long long seed,
                              unsigned int stream_id, int step, int* tok_out) {
    try {
        qtts::SampleParams p{};
        p.logits = logits; p.ld = ld; p.V = V; p.B = B;
        p.generated = generated; p.gen_stride = gen_stride; p.n_generated_dev = generated ? &n_generated : nullptr;
        p.repetition_penalty = repetition_penalty; p.eos = eos; p.min_new_tokens = min_new_tokens; p.suppress_mask = suppress_mask;
        p.do_sample = do_sample; p.top_k = top_k; p.top_p = top_p; p.temperature = temperature;
        p.seed = seed; p.stream_id = stream_id; p.step_dev = &step;
        p.tok_out = tok_out; p.tok_stride = 1;
        qtts::launch_sample(p, nullptr);
        return 0;
    } catch (const qtts::Error& e) { return e.code; } catch (...) { return -1; }
}

// One launch of attention.hip's attn_decode_kernel: q/k RMSNorm + RoPE + KV append + GQA attention for n_new tokens per
// sequence on a paged cache of S0 keys (layer 0; contiguous page layout unless a page table is given).
extern "C" int hostemu_attn_decode(const float* qkv, int ld, int B, int n_new, int nh, int nkv, const float* qw, const float* kw,
                                   float eps, const float* inv_freq, const int* n_pad, int S0, void* kpool, void* vpool,
                                   const int* page_table, int pages_per_seq, int bf16, float* out, int ldo, int max_len) {
    try {
        qtts::AttnDecodeParams p{};
        p.qkv = qkv; p.ld = ld; p.B = B; p.n_new = n_new; p.nh = nh; p.nkv = nkv; p.hd = 128;
        p.qw = qw; p.kw = kw; p.eps = eps; p.inv_freq = inv_freq; p.n_pad = n_pad;
        // the code predictor calls with a static length and a 32-key score buffer; the talker with a device-side length
        p.len_dev = (max_len <= 32 && !n_pad) ? nullptr : &S0; p.len_static = S0;
        p.kv.k = kpool; p.kv.v = vpool; p.kv.page_table = page_table; p.kv.pages_per_seq = pages_per_seq;
        p.kv.n_pages = B * pages_per_seq; p.kv.nkv = nkv; p.kv.hd = 128; p.kv.bf16 = bf16; p.kv.contig = page_table ? 0 : 1;
        p.layer = 0; p.out = out; p.ldo = ldo; p.out_bf16 = 0; p.max_len = max_len; p.done_flag = nullptr;
        if (const char* e = QTTS_ENV("QTTS_DEBUG_ATTN_VT")) p.kv.vt = atoi(e) != 0;        // V pool given dim-major inside its pages (attn_tk16)
        std::vector<float> part;
        if (const char* e = QTTS_ENV("QTTS_DEBUG_ATTN_NSPLIT")) {      // split-KV variant of the talker call shape
            if (atoi(e) > 1 && n_new == 1 && p.len_dev) {
                p.nsplit = atoi(e);
                part.assign(qtts::attn_part_floats(B, nkv, p.nsplit, nh / nkv), NAN);
                p.part = part.data();
            }
        }
        qtts::launch_attn_decode(p, nullptr);
        return 0;
    } catch (const qtts::Error& e) { return e.code; } catch (...) { return -1; }
}

// The code predictor's attention + o-projection of one single-token pass (bf16 cache, 16 query / 8 kv heads of 128) on B rows:
// fused = 0: attn_cp (bf16 output) followed by the decode GEMM (bf16 x, strips of `fs_unfused` features, + residual), as the engine ran it
// through round 3; fused = 1: cp_attn_o_kernel.  Same inputs, K / V pools updated in place, hidden rows to out (fp32) and out16 (bf16).
extern "C" int hostemu_cp_attn_o(const float* qkv, int ld, int B, const float* qw, const float* kw, float eps, const float* inv_freq, int S0,
                                 void* kpool, void* vpool, const int* page_table, int pages_per_seq, const float* Wo, int H,
                                 const float* res, float* out, unsigned short* out16, int fused, int fs_unfused, const float* rope_cs, int rope_cs_n,
                                 unsigned epoch0) {
    try {
        const int nh = 16, nkv = 8, qd = nh * 128;
        qtts::AttnDecodeParams a{};
        a.qkv = qkv; a.ld = ld; a.B = B; a.n_new = 1; a.nh = nh; a.nkv = nkv; a.hd = 128;
        a.qw = qw; a.kw = kw; a.eps = eps; a.inv_freq = inv_freq; a.len_static = S0;
        a.kv.k = kpool; a.kv.v = vpool; a.kv.page_table = page_table; a.kv.pages_per_seq = pages_per_seq;
        a.kv.n_pages = B * pages_per_seq; a.kv.nkv = nkv; a.kv.hd = 128; a.kv.bf16 = 1; a.kv.contig = page_table ? 0 : 1;
        a.layer = 0; a.max_len = 32; a.rope_cs = rope_cs; a.rope_cs_n = rope_cs_n;
        std::vector<unsigned char> wp(qtts::skinny_packed_bytes(H, qd, true));
        for (int i = 0; i < B * H; ++i) out[i] = res[i];            // the engine's residual stream is updated in place
        if (fused) {
            qtts::pack_skinny_weight(Wo, H, qd, true, wp.data(), nullptr, 16);
            std::vector<float> part((size_t)8 * 8 * H * 2, 0.f);
            int serial = (int)epoch0, err = 0;
            qtts::CpAttnOParams f{};
            f.a = a; f.Wo = wp.data(); f.res = out; f.out = out; f.out16 = out16; f.part = part.data(); f.serial = &serial; f.slot = 3; f.phase = 2;
            f.err = &err; f.H = H; f.first_pause = 16; f.poll_step = 8;
            if (!qtts::cp_attn_o_takes(a, H)) return -2;
            for (int rep = 0; rep < 2; ++rep) {                     // twice on the same buffers: the second launch must not take the first one's granules
                if (rep) { for (int i = 0; i < B * H; ++i) out[i] = res[i]; f.a.kv.k = kpool; f.a.kv.v = vpool; f.slot = 4; }
                qtts::launch_cp_attn_o(f, nullptr);
                if (err) return -4;
            }
            return 0;
        }
        std::vector<qtts::bf16_t> att((size_t)B * qd, (qtts::bf16_t)0x7FC0);
        a.out = reinterpret_cast<float*>(att.data()); a.ldo = qd; a.out_bf16 = 1;
        qtts::launch_attn_decode(a, nullptr);
        qtts::pack_skinny_weight(Wo, H, qd, true, wp.data(), nullptr, fs_unfused);
        qtts::SkinnyParams p{};
        p.x = reinterpret_cast<const float*>(att.data()); p.x_bf16 = 1; p.ldx = qd; p.M = B; p.Wp = wp.data(); p.N = H; p.K = qd;
        p.fs = fs_unfused; p.res = out; p.ldr = H; p.out = out; p.ldo = H; p.act = qtts::ACT_NONE; p.out16 = out16;
        qtts::launch_skinny(p, true, nullptr);
        return 0;
    } catch (const qtts::Error& e) { return e.code; } catch (...) { return -1; }
}

// One code-predictor layer front of a single-token pass (K = hidden = 1024, 16 / 8 heads of 128) on B rows: q|k|v GEMM with the
// folded RMSNorm + attention + o-projection.  mode 0: three launches (decode GEMM, attn_cp, decode GEMM); mode 2: ONE launch
// (cp_attn_o_kernel with the q|k|v front: every workgroup produces a strip of q|k|v, hands it over as tagged granules, then attends).
// x [B][H] fp32 (f32 = 0: rounded to bf16 here, as the engine's bf16 copy of the hidden state), Wqkv [(nh + 2 nkv) * 128][H], gnorm [H].
// f32 = 1: the exact parity mode -- fp32 operators, rows and cache (kpool / vpool hold floats), cp_attn_o_kernel<.., .., true>.
extern "C" int hostemu_cp_layer_front(const float* x, int B, const float* Wqkv, const float* gnorm, float eps_in, const float* qw, const float* kw,
                                      float eps, const float* inv_freq, int S0, void* kpool, void* vpool, const int* page_table, int pages_per_seq,
                                      const float* Wo, int H, const float* res, float* out, unsigned short* out16, int mode, const float* rope_cs,
                                      int rope_cs_n, unsigned epoch0, int f32) {
    try {
        const bool bf = !f32;
        const int nh = 16, nkv = 8, qd = nh * 128, ld = (nh + 2 * nkv) * 128;
        std::vector<qtts::bf16_t> x16((size_t)B * H);
        for (size_t i = 0; i < x16.size(); ++i) x16[i] = qtts::f32_to_bf16(x[i]);
        std::vector<unsigned char> wq(qtts::skinny_packed_bytes(ld, H, bf));
        qtts::pack_skinny_weight(Wqkv, ld, H, bf, wq.data(), gnorm, 16);
        std::vector<float> qkv((size_t)B * ld, NAN);
        qtts::AttnDecodeParams a{};
        a.qkv = qkv.data(); a.ld = ld; a.B = B; a.n_new = 1; a.nh = nh; a.nkv = nkv; a.hd = 128;
        a.qw = qw; a.kw = kw; a.eps = eps; a.inv_freq = inv_freq; a.len_static = S0;
        a.kv.k = kpool; a.kv.v = vpool; a.kv.page_table = page_table; a.kv.pages_per_seq = pages_per_seq;
        a.kv.n_pages = B * pages_per_seq; a.kv.nkv = nkv; a.kv.hd = 128; a.kv.bf16 = bf ? 1 : 0; a.kv.contig = page_table ? 0 : 1;
        a.layer = 0; a.max_len = 32; a.rope_cs = rope_cs; a.rope_cs_n = rope_cs_n;
        std::vector<unsigned char> wp(qtts::skinny_packed_bytes(H, qd, bf));
        for (int i = 0; i < B * H; ++i) out[i] = res[i];
        if (mode == 2) {
            qtts::pack_skinny_weight(Wo, H, qd, bf, wp.data(), nullptr, 16);
            std::vector<float> part((size_t)8 * 8 * H * 2, 0.f), gran((size_t)8 * ld * 2, 0.f);
            int serial = (int)epoch0, err = 0;
            qtts::CpAttnOParams f{};
            f.a = a; f.a.qkv = nullptr; f.Wo = wp.data(); f.res = out; f.out = out; f.out16 = bf ? out16 : nullptr; f.part = part.data(); f.serial = &serial; f.slot = 9;
            f.phase = 2; f.err = &err; f.H = H; f.first_pause = 16; f.poll_step = 8;
            f.Wqkv = wq.data(); f.x16 = bf ? x16.data() : reinterpret_cast<const unsigned short*>(x); f.ldx16 = H; f.K = H; f.eps_in = eps_in; f.qkv_gran = gran.data();
            for (int rep = 0; rep < 2; ++rep) {                     // twice on the same granule buffers
                if (rep) { for (int i = 0; i < B * H; ++i) out[i] = res[i]; ++serial; }
                qtts::launch_cp_attn_o(f, nullptr);
                if (err) return -4;
            }
            // the consuming half ALONE on the buffers the launches above filled: under the same (serial, slot) it finds every granule; under
            // another slot or another serial every granule is a stale one -- the consumers must give up and say so, not take them
            std::vector<float> keep(out, out + (size_t)B * H);
            std::vector<unsigned short> keep16(out16, out16 + (size_t)B * H);
            for (int variant = 0; variant < 3; ++variant) {
                for (int i = 0; i < B * H; ++i) out[i] = res[i];
                qtts::CpAttnOParams c2 = f;
                c2.phase = 1;
                if (variant == 1) c2.slot = f.slot + 1;
                int serial2 = serial + 1;
                if (variant == 2) c2.serial = &serial2;
                err = 0;
                int latch = 0;
                c2.done_latch = &latch;
                qtts::launch_cp_attn_o(c2, nullptr);
                if ((variant == 0) != (err == 0)) return -5 - variant;
                if ((err != 0) != (latch != 0)) return -9;                         // a give-up latches the generation's stop flag: what this launch still writes is never consumed
                if (variant == 0 && memcmp(out, keep.data(), keep.size() * 4) != 0) return -8;

            }
            memcpy(out, keep.data(), keep.size() * 4);
            memcpy(out16, keep16.data(), keep16.size() * 2);
            return 0;
        }
        qtts::SkinnyParams q{};
        q.x = bf ? reinterpret_cast<const float*>(x16.data()) : x; q.x_bf16 = bf ? 1 : 0; q.ldx = H; q.M = B; q.Wp = wq.data(); q.N = ld; q.K = H;
        q.fs = 16; q.norm = 1; q.eps = eps_in; q.out = qkv.data(); q.ldo = ld; q.act = qtts::ACT_NONE;
        std::vector<float> ss(B, 0.f);
        if (!bf && !qtts::skinny_f32_inline_norm(B, H)) {
            for (int m = 0; m < B; ++m) { double acc = 0; for (int k = 0; k < H; ++k) acc += (double)x[(size_t)m * H + k] * x[(size_t)m * H + k]; ss[m] = (float)acc; }
            q.ss_in = ss.data();
        }
        qtts::launch_skinny(q, bf, nullptr);
        std::vector<qtts::bf16_t> att((size_t)B * qd, (qtts::bf16_t)0x7FC0);
        std::vector<float> att32((size_t)B * qd, NAN);
        a.out = bf ? reinterpret_cast<float*>(att.data()) : att32.data(); a.ldo = qd; a.out_bf16 = bf ? 1 : 0;
        qtts::launch_attn_decode(a, nullptr);
        qtts::pack_skinny_weight(Wo, H, qd, bf, wp.data(), nullptr, bf ? 8 : 16);
        qtts::SkinnyParams p{};
        p.x = bf ? reinterpret_cast<const float*>(att.data()) : att32.data(); p.x_bf16 = bf ? 1 : 0; p.ldx = qd; p.M = B; p.Wp = wp.data(); p.N = H; p.K = qd;
        p.fs = bf ? 8 : 16; p.res = out; p.ldr = H; p.out = out; p.ldo = H; p.act = qtts::ACT_NONE; p.out16 = bf ? out16 : nullptr;
        qtts::launch_skinny(p, bf, nullptr);
        return 0;
    } catch (const qtts::Error& e) { return e.code; } catch (...) { return -1; }
}

// cp_mlp32_kernel (cp_mlp32.hip, round 6): the fused MLP launch at batch <= 32 (two 16-row tiles), bf16: twice on the same granule buffers under two
// serials (the second finds the first's granules under another tag).
extern "C" int hostemu_cp_mlp32(const float* x, int B, const float* Wg, const float* Wu, const float* gnorm, float eps, const float* Wd, int H, int I,
                                const float* res, float* out, unsigned short* out16, unsigned epoch0) {
    try {
        std::vector<qtts::bf16_t> x16((size_t)B * H);
        for (size_t i = 0; i < x16.size(); ++i) x16[i] = qtts::f32_to_bf16(x[i]);
        std::vector<unsigned char> wgu(qtts::cp_mlp_gu_bytes(H, I, true)), wd(qtts::skinny_packed_bytes(H, I, true));
        qtts::pack_cp_mlp_gu(Wg, Wu, gnorm, H, I, true, wgu.data());
        qtts::pack_skinny_weight(Wd, H, I, true, wd.data(), nullptr, 16);
        std::vector<unsigned char> act(qtts::cp_mlp32_act_bytes(I), 0xFF), part(qtts::cp_mlp32_part_bytes(H), 0xFF);
        int serial = (int)epoch0, err = 0, latch = 0;
        qtts::CpMlpParams m{};
        m.Wgu = wgu.data(); m.Wd = wd.data(); m.x16 = x16.data(); m.ldx16 = H; m.eps = eps;
        m.res = out; m.out = out; m.out16 = out16;
        m.act_gran = reinterpret_cast<float*>(act.data()); m.part = reinterpret_cast<float*>(part.data()); m.serial = &serial; m.slot = 11; m.phase = 3;
        m.err = &err; m.done_latch = &latch; m.first_pause = 16; m.pause_c = 16; m.poll_step = 4; m.B = B; m.H = H; m.I = I;
        for (int rep = 0; rep < 2; ++rep) {
            for (int i = 0; i < B * H; ++i) out[i] = res[i];
            if (rep) ++serial;
            qtts::launch_cp_mlp32(m, nullptr);
            if (err || latch) return -4;
        }
        return 0;
    } catch (const qtts::Error& e) { return e.code; } catch (...) { return -1; }
}

// cp_mlp_kernel (cp_mlp.hip): the code predictor's MLP as one launch against the two decode-GEMM launches it replaces, bf16 (f32 = 0) or the
// exact fp32 mode (f32 = 1: fp32 operators, fp32 rows, fp32 intermediate vector).
// mode 0: the two launches (bf16: ACT_SWIGLU8 where K % 512 == 0; fp32: strip pairs); mode 3: the fused launch, twice on the same granule buffers
// under two serials, then phases B + C alone under the launch's own tag (bit-identical) and under another slot / serial (stale: give-up + latch).
extern "C" int hostemu_cp_mlp(const float* x, int B, const float* Wg, const float* Wu, const float* gnorm, float eps, const float* Wd, int H, int I,
                              const float* res, float* out, unsigned short* out16, int mode, unsigned epoch0, int f32) {
    try {
        const bool bf = !f32;
        std::vector<qtts::bf16_t> x16((size_t)B * H);
        for (size_t i = 0; i < x16.size(); ++i) x16[i] = qtts::f32_to_bf16(x[i]);
        for (int i = 0; i < B * H; ++i) out[i] = res[i];
        if (mode == 3) {
            std::vector<unsigned char> wgu(qtts::cp_mlp_gu_bytes(H, I, bf)), wd(qtts::skinny_packed_bytes(H, I, bf));
            qtts::pack_cp_mlp_gu(Wg, Wu, gnorm, H, I, bf, wgu.data());
            qtts::pack_skinny_weight(Wd, H, I, bf, wd.data(), nullptr, 16);
            std::vector<float> act((size_t)8 * 8 * (I / (bf ? 16 : 8)) * 2, 0.f), part((size_t)8 * 8 * H * 2, 0.f);
            int serial = (int)epoch0, err = 0, latch = 0;
            qtts::CpMlpParams m{};
            m.f32 = f32;
            m.Wgu = wgu.data(); m.Wd = wd.data(); m.x16 = bf ? x16.data() : reinterpret_cast<const unsigned short*>(x); m.ldx16 = H; m.eps = eps;
            m.res = out; m.out = out; m.out16 = bf ? out16 : nullptr;
            m.act_gran = act.data(); m.part = part.data(); m.serial = &serial; m.slot = 11; m.phase = 3; m.err = &err; m.done_latch = &latch;
            m.first_pause = 16; m.pause_c = 16; m.poll_step = 4; m.B = B; m.H = H; m.I = I;
            for (int rep = 0; rep < 2; ++rep) {
                if (rep) { for (int i = 0; i < B * H; ++i) out[i] = res[i]; ++serial; }
                qtts::launch_cp_mlp(m, nullptr);
                if (err || latch) return -4;
            }
            std::vector<float> keep(out, out + (size_t)B * H);
            std::vector<unsigned short> keep16(out16, out16 + (size_t)B * H);
            for (int variant = 0; variant < 3; ++variant) {
                for (int ph = 1; ph <= 2; ++ph) {
                    for (int i = 0; i < B * H; ++i) out[i] = res[i];
                    qtts::CpMlpParams c2 = m;
                    c2.phase = ph;
                    if (variant == 1) c2.slot = m.slot + 1;
                    int serial2 = serial + 1;
                    if (variant == 2) c2.serial = &serial2;
                    err = 0; latch = 0;
                    if (variant != 0 && ph == 2) {     // (phase B under a foreign tag re-published the partial sums under that tag: restore the launch's own)
                        qtts::CpMlpParams c1 = m; c1.phase = 1; int e2 = 0, l2 = 0; c1.err = &e2; c1.done_latch = &l2;
                        qtts::launch_cp_mlp(c1, nullptr);
                        for (int i = 0; i < B * H; ++i) out[i] = res[i];
                    }
                    qtts::launch_cp_mlp(c2, nullptr);
                    if ((variant == 0) != (err == 0)) return -5 - variant - 10 * ph;
                    if ((err != 0) != (latch != 0)) return -9;
                    if (variant == 0 && ph == 2 && memcmp(out, keep.data(), keep.size() * 4) != 0) return -8;
                }
            }
            memcpy(out, keep.data(), keep.size() * 4);
            memcpy(out16, keep16.data(), keep16.size() * 2);
            return 0;
        }
        // the two launches: gate|up interleaved (8-row blocks for ACT_SWIGLU8, 16-row strip pairs otherwise), then the down-projection
        std::vector<float> gu((size_t)2 * I * H);
        const int blk = (bf && H % 512 == 0 && B <= 8) ? 8 : 16;      // (ACT_SWIGLU8 is the batch <= 8 kernel's)
        for (int f = 0; f < I; ++f) {
            memcpy(&gu[((size_t)(f / blk) * 2 * blk + f % blk) * H], Wg + (size_t)f * H, (size_t)H * 4);
            memcpy(&gu[((size_t)(f / blk) * 2 * blk + blk + f % blk) * H], Wu + (size_t)f * H, (size_t)H * 4);
        }
        std::vector<unsigned char> wp(qtts::skinny_packed_bytes(2 * I, H, bf)), wdp(qtts::skinny_packed_bytes(H, I, bf));
        qtts::pack_skinny_weight(gu.data(), 2 * I, H, bf, wp.data(), gnorm, 16);
        qtts::pack_skinny_weight(Wd, H, I, bf, wdp.data(), nullptr, bf ? 8 : 16);
        std::vector<float> act32((size_t)B * I, NAN);
        std::vector<qtts::bf16_t> act((size_t)B * I, (qtts::bf16_t)0x7FC0);
        qtts::SkinnyParams g{};
        g.x = bf ? reinterpret_cast<const float*>(x16.data()) : x; g.x_bf16 = bf ? 1 : 0; g.ldx = H; g.M = B; g.Wp = wp.data(); g.N = 2 * I; g.K = H; g.fs = 16;
        std::vector<float> ss(B, 0.f);
        if (!bf && !qtts::skinny_f32_inline_norm(B, H)) {
            for (int m = 0; m < B; ++m) { double a = 0; for (int k = 0; k < H; ++k) a += (double)x[(size_t)m * H + k] * x[(size_t)m * H + k]; ss[m] = (float)a; }
            g.ss_in = ss.data();
        }
        g.norm = 1; g.eps = eps; g.out = bf ? reinterpret_cast<float*>(act.data()) : act32.data(); g.out_bf16 = bf ? 1 : 0; g.ldo = I;
        g.act = blk == 8 ? qtts::ACT_SWIGLU8 : qtts::ACT_SWIGLU;
        qtts::launch_skinny(g, bf, nullptr);
        qtts::SkinnyParams d{};
        d.x = bf ? reinterpret_cast<const float*>(act.data()) : act32.data(); d.x_bf16 = bf ? 1 : 0; d.ldx = I; d.M = B; d.Wp = wdp.data(); d.N = H; d.K = I;
        d.fs = bf ? 8 : 16; d.res = out; d.ldr = H; d.out = out; d.ldo = H; d.act = qtts::ACT_NONE; d.out16 = bf ? out16 : nullptr;
        qtts::launch_skinny(d, bf, nullptr);
        return 0;
    } catch (const qtts::Error& e) {
        return e.code;
    }
}
