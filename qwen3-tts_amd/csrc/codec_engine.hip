// codec_engine.hip -- host-side orchestration of the Qwen3-TTS-Tokenizer-12Hz decoder on gfx950.
//
// Data layout in HBM: every activation is CHANNEL-LAST fp32 [batch*time][channels]; weights are repacked
// once at finalize() into [tap][out][in] (fp32 or bf16) so that each causal conv / conv-transpose / linear
// is one `gemm_tap` launch (see gemm_tap.hip).  Stage order follows Qwen3TTSTokenizerV2Decoder.forward
// (tokenizer v2:869-884); chunking follows chunked_decode (v2:886-896).
#include <map>
#include <set>
#include <mutex>
#include <atomic>
#include <tuple>
#include <functional>
#include <algorithm>
#include "common.h"
#include "kernels.h"

namespace qtts {

struct Lin {                 // one gemm_tap operator
    DevBuf W, bias;
    int N = 0, K = 0, taps = 1;
    int shift[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool has_bias = false;
};
struct Snake { DevBuf ea, ib; };

}  // namespace qtts

using namespace qtts;

struct qtts_codec {
    qtts_codec_config cfg;
    bool bf16 = false, finalized = false;
    std::map<std::string, std::vector<float>> host;
    std::map<std::string, std::vector<int64_t>> shapes;

    int vq = 0, up_total = 1;
    DevBuf tables, inv_freq;
    Lin rvq_out, pre_conv, in_proj, out_proj, dec0;
    DevBuf t_norm;
    struct TLayer { Lin qkv, o, gu, down; DevBuf n1, n2, ls1, ls2; };
    std::vector<TLayer> tl;
    struct Up { Lin tconv, pw1, pw2; DevBuf dw_w, dw_b, ln_w, ln_b, gamma; };
    std::vector<Up> ups;
    struct Unit { Snake a1, a2; Lin c1, c2; DevBuf w1p, w2p; int dil = 1; bool fused = false; };   // w1p / w2p: resunit.hip's fragment-packed weights
    struct Block { Snake act; Lin tconv; Unit u[3]; int r, cin, cout; };
    std::vector<Block> blocks;
    Snake final_act;
    DevBuf final_w; float final_b = 0.f; int final_c = 0;

    DevBuf buf[4];
    DevBuf buf16[2];                    // bf16 mode: activations that are only a GEMM input (decoder blocks)
    size_t buf_elems = 0;
    DevBuf err_flag;                    // device int: a code index >= codebook_size was seen (checked by the entry points)

    // ---- hipGraph replay of whole decode calls (round 4).  A decode is ~140 launches whose arguments are fixed by (B, T, chunking) once
    // the codes and the waveform live at fixed addresses: the engine owns a staging buffer for each (`g_codes`, `g_wav`, `g_pre`); a
    // call copies its codes in (device to device, a few KB to a few hundred KB), replays the graph captured for its shape, and copies
    // the waveform out to the caller's buffer.  The SECOND call with a shape captures the launch sequence (on a private stream -- a capture
    // executes nothing, and the caller's stream may be the legacy default stream, which cannot be captured); this and every later call
    // replay it with one hipGraphLaunch on the caller's stream.  Keyed on the SHAPE only: a caller that allocates fresh tensors per
    // call (PyTorch does) still replays, and no call pays a capture because its addresses changed (first version of this round: keyed
    // on the caller's pointers -- a first-packet p99 of 37-58 ms where round 3 had 32.4, every new address pair a 5-25 ms
    // capture + instantiate inside a request).  QTTS_CODEC_GRAPH=0: always eager, no staging (A/B).
    struct GraphKey {
        int B, T, chunk, left, has_pre;
        bool operator<(const GraphKey& o) const {
            return std::tie(B, T, chunk, left, has_pre) < std::tie(o.B, o.T, o.chunk, o.left, o.has_pre);
        }
    };
    DevBuf g_codes, g_wav, g_pre;
    // staging sized for this call; growing a buffer moves it, so every captured graph (they bake the old address) is dropped first
    void ensure_staging(size_t codes_bytes, size_t wav_bytes, bool pre) {
        if (codes_bytes > g_codes.bytes || wav_bytes > g_wav.bytes || (pre && wav_bytes > g_pre.bytes)) {
            drop_graphs();
            if (codes_bytes > g_codes.bytes) g_codes.alloc(codes_bytes);
            if (wav_bytes > g_wav.bytes) g_wav.alloc(wav_bytes);
            if (pre && wav_bytes > g_pre.bytes) g_pre.alloc(wav_bytes);
        }
    }
    struct GraphSlot { hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr; uint64_t last_use = 0; int seen = 0; int nodes = 0; };
    std::map<GraphKey, GraphSlot> graphs;
    hipStream_t cap_stream = nullptr;
    uint64_t graph_clock = 0;
    int graph_replays = 0, graph_captures = 0, graph_nodes_replayed = 0;
    static constexpr size_t GRAPH_SLOTS = 8;
    static bool graph_enabled() { return QTTS_OPT_ON("QTTS_CODEC_GRAPH"); }
    void drop_graphs() {
        for (auto& kv : graphs) {
            if (kv.second.ge) (void)hipGraphExecDestroy(kv.second.ge);
            if (kv.second.g) (void)hipGraphDestroy(kv.second.g);
        }
        graphs.clear();
    }
    ~qtts_codec() {
        drop_graphs();
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
    }
    // run `body(stream)` -- a launch sequence without host synchronisation -- eagerly on `st`, or as a cached graph
    // `eager(stream)` runs the launch sequence on the caller's buffers; `staged(stream)` the same sequence on the staging buffers (that is
    // what a capture records); `copy_in` / `copy_out` move the caller's data to / from the staging buffers around a replay.
    template <class FE, class FS, class FI, class FO>
    void run_graphed(const GraphKey& key, hipStream_t st, FE&& eager, FS&& staged, FI&& copy_in, FO&& copy_out) {
        if (!graph_enabled()) { eager(st); return; }
        GraphSlot& slot = graphs[key];
        slot.last_use = ++graph_clock;
        auto body = [&](hipStream_t s) { staged(s); };
        if (!slot.ge) {
            if (++slot.seen < 2) {                                   // first sight: eager (also loads every code object lazily loaded)
                eager(st);
                evict();
                return;
            }
            if (!cap_stream) QTTS_CHECK_HIP(hipStreamCreateWithFlags(&cap_stream, hipStreamNonBlocking));
            QTTS_CHECK_HIP(hipStreamBeginCapture(cap_stream, hipStreamCaptureModeThreadLocal));
            try { body(cap_stream); }
            catch (...) {
                hipGraph_t gx = nullptr;
                (void)hipStreamEndCapture(cap_stream, &gx);
                if (gx) (void)hipGraphDestroy(gx);
                graphs.erase(key);
                throw;
            }
            QTTS_CHECK_HIP(hipStreamEndCapture(cap_stream, &slot.g));
            size_t nn = 0;
            QTTS_CHECK_HIP(hipGraphGetNodes(slot.g, nullptr, &nn));
            slot.nodes = (int)nn;
            if (hipGraphInstantiate(&slot.ge, slot.g, nullptr, nullptr, 0) != hipSuccess) {      // (no half-built slot survives a failure)
                (void)hipGraphDestroy(slot.g);
                graphs.erase(key);
                throw Error(QTTS_ERR_HIP, "codec: hipGraphInstantiate failed");
            }
            ++graph_captures;
        }
        copy_in(st);
        QTTS_CHECK_HIP(hipGraphLaunch(slot.ge, st));
        copy_out(st);
        ++graph_replays;
        graph_nodes_replayed = slot.nodes;
        evict();
    }
    // Captured graphs and shapes that were merely SEEN once are bounded separately: a run of one-off shapes (variable-length
    // non-streaming decode) can push out neither a captured graph nor -- before SEEN_SLOTS of them -- the memory of a shape's first sighting.
    static constexpr size_t SEEN_SLOTS = 64;
    void evict() {
        auto drop_lru = [&](bool captured, size_t cap) {
            for (;;) {
                size_t n = 0;
                auto lru = graphs.end();
                for (auto it = graphs.begin(); it != graphs.end(); ++it) {
                    if ((it->second.ge != nullptr) != captured) continue;
                    ++n;
                    if (lru == graphs.end() || it->second.last_use < lru->second.last_use) lru = it;
                }
                if (n <= cap) return;
                if (lru->second.ge) (void)hipGraphExecDestroy(lru->second.ge);
                if (lru->second.g) (void)hipGraphDestroy(lru->second.g);
                graphs.erase(lru);
            }
        };
        drop_lru(true, GRAPH_SLOTS);
        drop_lru(false, SEEN_SLOTS);
    }
    void check_codes_flag(hipStream_t st) {
        int e = 0;
        QTTS_CHECK_HIP(hipMemcpyAsync(&e, err_flag.p, 4, hipMemcpyDeviceToHost, st));
        QTTS_CHECK_HIP(hipStreamSynchronize(st));
        if (e) {
            QTTS_CHECK_HIP(hipMemset(err_flag.p, 0, 4));
            throw Error(QTTS_ERR_ARG, "codec: code index out of range (>= codebook_size)");
        }
    }

    // ---- streaming (state-carrying) decode: one session per handle, B sequences advancing in lockstep
    struct Carry { DevBuf d; int h = 0, C = 0; };
    std::vector<Carry> carry;           // in the order the stateful layers run
    DevBuf stream_npad;                 // [B] left-pad count of the staged KV window
    std::vector<int> stream_npad_host;
    int stream_B = 0;                   // 0 = no session
    int64_t stream_t = 0;               // frames decoded so far
    void stream_begin(int B);
    void stream_push(const int64_t* codes, int n, float* wav, hipStream_t st);

    std::vector<float>& P(const std::string& n) {
        auto it = host.find(n);
        if (it == host.end()) throw Error(QTTS_ERR_UNBOUND, "codec weight not bound: " + n);
        return it->second;
    }
    void expect(const std::string& n, std::initializer_list<int64_t> want) {
        P(n);
        auto& s = shapes[n];
        if (s.size() != want.size() || !std::equal(s.begin(), s.end(), want.begin())) {
            std::string a, b;
            for (auto d : s) a += std::to_string(d) + ",";
            for (auto d : want) b += std::to_string(d) + ",";
            throw Error(QTTS_ERR_ARG, "codec weight " + n + " has shape (" + a + ") but the config implies (" + b + ")");
        }
    }
    void upload_w(DevBuf& d, const std::vector<float>& w) {
        if (bf16) {
            std::vector<bf16_t> h(w.size());
            for (size_t i = 0; i < w.size(); ++i) h[i] = f32_to_bf16(w[i]);
            d.upload(h.data(), h.size() * 2);
        } else d.upload(w.data(), w.size() * 4);
    }
    void upload_f(DevBuf& d, const std::vector<float>& w) { d.upload(w.data(), w.size() * 4); }

    // nn.Linear weight (N,K) (+bias)
    void make_linear(Lin& l, const std::string& wname, const std::string& bname) {
        auto& w = P(wname); auto& s = shapes[wname];
        l.N = (int)s[0]; l.K = (int)s[1]; l.taps = 1; l.shift[0] = 0;
        upload_w(l.W, w);
        if (!bname.empty()) { upload_f(l.bias, P(bname)); l.has_bias = true; }
    }
    // causal Conv1d weight (Cout, Cin, k), dilation d
    void make_conv(Lin& l, const std::string& pfx, int dil) {
        auto& w = P(pfx + ".weight"); auto& s = shapes[pfx + ".weight"];
        const int Co = (int)s[0], Ci = (int)s[1], k = (int)s[2];
        QTTS_REQUIRE(k <= 8, QTTS_ERR_ARG, "conv kernel > 8");
        std::vector<float> r((size_t)k * Co * Ci);
        for (int j = 0; j < k; ++j)
            for (int n = 0; n < Co; ++n)
                for (int c = 0; c < Ci; ++c) r[((size_t)j * Co + n) * Ci + c] = w[((size_t)n * Ci + c) * k + j];
        l.N = Co; l.K = Ci; l.taps = k;
        for (int j = 0; j < k; ++j) l.shift[j] = -(k - 1 - j) * dil;
        upload_w(l.W, r);
        upload_f(l.bias, P(pfx + ".bias")); l.has_bias = true;
    }
    // ConvTranspose1d weight (Cin, Cout, k) with stride r, right-trim k - r  (k == r or k == 2r)
    void make_tconv(Lin& l, const std::string& pfx, int r) {
        auto& w = P(pfx + ".weight"); auto& s = shapes[pfx + ".weight"];
        const int Ci = (int)s[0], Co = (int)s[1], k = (int)s[2];
        QTTS_REQUIRE(k == r || k == 2 * r, QTTS_ERR_ARG, "transposed conv: kernel must be r or 2r");
        const int taps = k / r;
        std::vector<float> rp((size_t)taps * r * Co * Ci);
        for (int t = 0; t < taps; ++t)
            for (int p = 0; p < r; ++p)
                for (int co = 0; co < Co; ++co)
                    for (int ci = 0; ci < Ci; ++ci)
                        rp[(((size_t)t * r + p) * Co + co) * Ci + ci] = w[((size_t)ci * Co + co) * k + p + t * r];
        l.N = r * Co; l.K = Ci; l.taps = taps; l.shift[0] = 0; l.shift[1] = -1;
        upload_w(l.W, rp);
        auto& b = P(pfx + ".bias");
        std::vector<float> be((size_t)r * Co);
        for (int p = 0; p < r; ++p) for (int co = 0; co < Co; ++co) be[(size_t)p * Co + co] = b[co];
        upload_f(l.bias, be); l.has_bias = true;
    }
    void make_snake(Snake& s, const std::string& pfx) {
        auto& a = P(pfx + ".alpha"); auto& b = P(pfx + ".beta");
        std::vector<float> ea(a.size()), ib(b.size());
        for (size_t i = 0; i < a.size(); ++i) { ea[i] = expf(a[i]); ib[i] = 1.0f / (expf(b[i]) + 1e-9f); }
        upload_f(s.ea, ea); upload_f(s.ib, ib);
    }

    void gemm(const Lin& l, const float* A, int lda, int M, int T, float* C, int ldc, int act = ACT_NONE,
              const float* res = nullptr, int ldr = 0, const float* scale = nullptr, const Snake* sn = nullptr,
              hipStream_t st = nullptr) {
        GemmTapParams p{};
        p.A = A; p.lda = lda; p.M = M; p.T = T; p.W = l.W.p; p.N = l.N; p.K = l.K; p.taps = l.taps;
        for (int i = 0; i < 8; ++i) p.shift[i] = l.shift[i];
        p.bias = l.has_bias ? l.bias.as<float>() : nullptr;
        p.scale = scale; p.res = res; p.ldr = ldr;
        p.snake_ea = sn ? sn->ea.as<float>() : nullptr; p.snake_ib = sn ? sn->ib.as<float>() : nullptr;
        p.act = act; p.C = C; p.ldc = ldc;
        launch_gemm_tap(p, bf16, st);
    }
    // bf16 mode, decoder blocks: the same GEMM with a bf16 input (A16, selects gemm_tap2) and / or a bf16 copy of the result
    // (C16) that already carries the next consumer's SnakeBeta (sn16); C may be null when only the bf16 copy is consumed
    void gemm16(const Lin& l, const float* A, const bf16_t* A16, int lda, int M, int T, float* C, int ldc, int act, const float* res,
                int ldr, const Snake* sn, bf16_t* C16, const Snake* sn16, hipStream_t st, int sn16_period = 0,
                const bf16_t* res16 = nullptr, bf16_t* R16 = nullptr, const float* scale = nullptr) {
        GemmTapParams p{};
        p.scale = scale;
        p.A = A; p.A16 = A16; p.lda = lda; p.M = M; p.T = T; p.W = l.W.p; p.N = l.N; p.K = l.K; p.taps = l.taps;
        for (int i = 0; i < 8; ++i) p.shift[i] = l.shift[i];
        p.bias = l.has_bias ? l.bias.as<float>() : nullptr;
        p.res = res; p.ldr = ldr;
        p.snake_ea = sn ? sn->ea.as<float>() : nullptr; p.snake_ib = sn ? sn->ib.as<float>() : nullptr;
        p.act = act; p.C = C; p.ldc = ldc;
        p.C16 = C16; p.ldc16 = ldc; p.act16 = sn16 ? ACT_SNAKE : ACT_NONE;
        p.snake16_ea = sn16 ? sn16->ea.as<float>() : nullptr; p.snake16_ib = sn16 ? sn16->ib.as<float>() : nullptr;
        p.snake16_period = sn16_period;
        p.res16 = res16; p.ldres16 = ldr; p.R16 = R16; p.ldR16 = ldc;
        launch_gemm_tap(p, bf16, st);
    }

    void finalize();
    // one forward over frames [t0, t0+Tc) of `codes` (strides in elements).  If stage != null, stop after that
    // stage and copy it (channel-last) to stage_out.
    void forward(const int64_t* codes, int B, int64_t sb, int64_t sq, int64_t stt, int t0, int Tc, float* wav,
                 float* pre, int64_t wav_stride_b, int64_t skip_samples, const char* stage, float* stage_out,
                 int64_t cap, int64_t* L_out, int64_t* C_out, hipStream_t st);
};

void qtts_codec::finalize() {
    const auto& c = cfg;
    vq = c.codebook_dim / 2;
    up_total = 1;
    for (int i = 0; i < c.n_upsample_rates; ++i) up_total *= c.upsample_rates[i];
    for (int i = 0; i < c.n_upsampling_ratios; ++i) up_total *= c.upsampling_ratios[i];
    QTTS_REQUIRE(c.head_dim == 64 || c.head_dim == 128, QTTS_ERR_ARG, "codec head_dim must be 64 or 128");
    {   // every tensor whose shape the config determines is checked before anything is repacked
        const int64_t H = c.hidden_size, I = c.intermediate_size, Ld = c.latent_dim, D = c.decoder_dim;
        const int64_t qd = (int64_t)c.num_attention_heads * c.head_dim, kvd = (int64_t)c.num_key_value_heads * c.head_dim;
        for (int q = 0; q < c.num_quantizers; ++q) {
            const std::string p = q == 0 ? "quantizer.rvq_first.vq.layers.0._codebook."
                                         : "quantizer.rvq_rest.vq.layers." + std::to_string(q - 1) + "._codebook.";
            expect(p + "embedding_sum", {c.codebook_size, vq});
            expect(p + "cluster_usage", {c.codebook_size});
        }
        expect("quantizer.rvq_first.output_proj.weight", {c.codebook_dim, vq, 1});
        expect("quantizer.rvq_rest.output_proj.weight", {c.codebook_dim, vq, 1});
        expect("pre_conv.conv.weight", {Ld, c.codebook_dim, 3});
        expect("pre_conv.conv.bias", {Ld});
        expect("pre_transformer.input_proj.weight", {H, Ld});
        expect("pre_transformer.input_proj.bias", {H});
        expect("pre_transformer.output_proj.weight", {Ld, H});
        expect("pre_transformer.output_proj.bias", {Ld});
        expect("pre_transformer.norm.weight", {H});
        for (int l = 0; l < c.num_hidden_layers; ++l) {
            const std::string p = "pre_transformer.layers." + std::to_string(l) + ".";
            expect(p + "self_attn.q_proj.weight", {qd, H});
            expect(p + "self_attn.k_proj.weight", {kvd, H});
            expect(p + "self_attn.v_proj.weight", {kvd, H});
            expect(p + "self_attn.o_proj.weight", {H, qd});
            expect(p + "mlp.gate_proj.weight", {I, H});
            expect(p + "mlp.up_proj.weight", {I, H});
            expect(p + "mlp.down_proj.weight", {H, I});
            expect(p + "input_layernorm.weight", {H});
            expect(p + "post_attention_layernorm.weight", {H});
            expect(p + "self_attn_layer_scale.scale", {H});
            expect(p + "mlp_layer_scale.scale", {H});
        }
        for (int u = 0; u < c.n_upsampling_ratios; ++u) {
            const std::string p = "upsample." + std::to_string(u) + ".";
            expect(p + "0.conv.weight", {Ld, Ld, c.upsampling_ratios[u]});
            expect(p + "0.conv.bias", {Ld});
            expect(p + "1.dwconv.conv.weight", {Ld, 1, 7});
            expect(p + "1.dwconv.conv.bias", {Ld});
            expect(p + "1.norm.weight", {Ld});
            expect(p + "1.norm.bias", {Ld});
            expect(p + "1.pwconv1.weight", {4 * Ld, Ld});
            expect(p + "1.pwconv1.bias", {4 * Ld});
            expect(p + "1.pwconv2.weight", {Ld, 4 * Ld});
            expect(p + "1.pwconv2.bias", {Ld});
            expect(p + "1.gamma", {Ld});
        }
        expect("decoder.0.conv.weight", {D, Ld, 7});
        expect("decoder.0.conv.bias", {D});
        for (int i = 0; i < c.n_upsample_rates; ++i) {
            const std::string p = "decoder." + std::to_string(i + 1) + ".block.";
            const int64_t ci = D >> i, co = D >> (i + 1), r = c.upsample_rates[i];
            expect(p + "0.alpha", {ci}); expect(p + "0.beta", {ci});
            expect(p + "1.conv.weight", {ci, co, 2 * r}); expect(p + "1.conv.bias", {co});
            for (int j = 2; j <= 4; ++j) {
                const std::string up = p + std::to_string(j) + ".";
                expect(up + "act1.alpha", {co}); expect(up + "act1.beta", {co});
                expect(up + "act2.alpha", {co}); expect(up + "act2.beta", {co});
                expect(up + "conv1.conv.weight", {co, co, 7}); expect(up + "conv1.conv.bias", {co});
                expect(up + "conv2.conv.weight", {co, co, 1}); expect(up + "conv2.conv.bias", {co});
            }
        }
        const int n = c.n_upsample_rates;
        const int64_t cl = D >> n;
        expect("decoder." + std::to_string(n + 1) + ".alpha", {cl});
        expect("decoder." + std::to_string(n + 1) + ".beta", {cl});
        expect("decoder." + std::to_string(n + 2) + ".conv.weight", {1, cl, 7});
        expect("decoder." + std::to_string(n + 2) + ".conv.bias", {1});
    }
    // normalised codebooks: embedding_sum / clamp(cluster_usage, 1e-5) (v2:676-679), computed once
    {
        std::vector<float> t((size_t)c.num_quantizers * c.codebook_size * vq);
        for (int q = 0; q < c.num_quantizers; ++q) {
            const std::string p = q == 0 ? "quantizer.rvq_first.vq.layers.0._codebook."
                                         : "quantizer.rvq_rest.vq.layers." + std::to_string(q - 1) + "._codebook.";
            auto& es = P(p + "embedding_sum"); auto& cu = P(p + "cluster_usage");
            for (int i = 0; i < c.codebook_size; ++i) {
                const float d = std::max(cu[i], 1e-5f);
                for (int j = 0; j < vq; ++j)
                    t[((size_t)q * c.codebook_size + i) * vq + j] = es[(size_t)i * vq + j] / d;
            }
        }
        upload_f(tables, t);
    }
    {   // fused output_proj: [N = codebook_dim][K = 2*vq] = [first | rest] (v2:764-766, 815-821)
        auto& wf = P("quantizer.rvq_first.output_proj.weight"); auto& wr = P("quantizer.rvq_rest.output_proj.weight");
        std::vector<float> w((size_t)c.codebook_dim * 2 * vq);
        for (int n = 0; n < c.codebook_dim; ++n)
            for (int k = 0; k < vq; ++k) {
                w[(size_t)n * 2 * vq + k] = wf[(size_t)n * vq + k];
                w[(size_t)n * 2 * vq + vq + k] = wr[(size_t)n * vq + k];
            }
        rvq_out.N = c.codebook_dim; rvq_out.K = 2 * vq; rvq_out.taps = 1;
        upload_w(rvq_out.W, w);
    }
    make_conv(pre_conv, "pre_conv.conv", 1);
    make_linear(in_proj, "pre_transformer.input_proj.weight", "pre_transformer.input_proj.bias");
    make_linear(out_proj, "pre_transformer.output_proj.weight", "pre_transformer.output_proj.bias");
    upload_f(t_norm, P("pre_transformer.norm.weight"));
    if (host.count("pre_transformer.rotary_emb.inv_freq")) upload_f(inv_freq, P("pre_transformer.rotary_emb.inv_freq"));
    else {
        std::vector<float> f(c.head_dim / 2);
        for (int i = 0; i < c.head_dim / 2; ++i) f[i] = 1.0f / powf(c.rope_theta, (float)(2 * i) / (float)c.head_dim);
        upload_f(inv_freq, f);
    }
    const int H = c.hidden_size, I = c.intermediate_size;
    QTTS_REQUIRE(I % 16 == 0, QTTS_ERR_ARG, "codec intermediate_size % 16");
    tl.resize(c.num_hidden_layers);
    for (int l = 0; l < c.num_hidden_layers; ++l) {
        const std::string p = "pre_transformer.layers." + std::to_string(l) + ".";
        auto& L = tl[l];
        {   // fused q|k|v rows
            auto& q = P(p + "self_attn.q_proj.weight"); auto& k = P(p + "self_attn.k_proj.weight");
            auto& v = P(p + "self_attn.v_proj.weight");
            std::vector<float> w; w.reserve(q.size() + k.size() + v.size());
            w.insert(w.end(), q.begin(), q.end()); w.insert(w.end(), k.begin(), k.end()); w.insert(w.end(), v.begin(), v.end());
            L.qkv.N = (int)(w.size() / H); L.qkv.K = H; upload_w(L.qkv.W, w);
        }
        make_linear(L.o, p + "self_attn.o_proj.weight", "");
        {   // gate/up interleaved in 16-row blocks for the SwiGLU epilogue
            auto& g = P(p + "mlp.gate_proj.weight"); auto& u = P(p + "mlp.up_proj.weight");
            std::vector<float> w((size_t)2 * I * H);
            for (int f = 0; f < I; ++f) {
                memcpy(&w[((size_t)(f / 16) * 32 + f % 16) * H], &g[(size_t)f * H], H * 4);
                memcpy(&w[((size_t)(f / 16) * 32 + 16 + f % 16) * H], &u[(size_t)f * H], H * 4);
            }
            L.gu.N = 2 * I; L.gu.K = H; upload_w(L.gu.W, w);
        }
        make_linear(L.down, p + "mlp.down_proj.weight", "");
        upload_f(L.n1, P(p + "input_layernorm.weight"));
        upload_f(L.n2, P(p + "post_attention_layernorm.weight"));
        upload_f(L.ls1, P(p + "self_attn_layer_scale.scale"));
        upload_f(L.ls2, P(p + "mlp_layer_scale.scale"));
    }
    ups.resize(c.n_upsampling_ratios);
    for (int u = 0; u < c.n_upsampling_ratios; ++u) {
        const std::string p = "upsample." + std::to_string(u) + ".";
        make_tconv(ups[u].tconv, p + "0.conv", c.upsampling_ratios[u]);
        upload_f(ups[u].dw_w, P(p + "1.dwconv.conv.weight"));
        upload_f(ups[u].dw_b, P(p + "1.dwconv.conv.bias"));
        upload_f(ups[u].ln_w, P(p + "1.norm.weight"));
        upload_f(ups[u].ln_b, P(p + "1.norm.bias"));
        make_linear(ups[u].pw1, p + "1.pwconv1.weight", p + "1.pwconv1.bias");
        make_linear(ups[u].pw2, p + "1.pwconv2.weight", p + "1.pwconv2.bias");
        upload_f(ups[u].gamma, P(p + "1.gamma"));
    }
    make_conv(dec0, "decoder.0.conv", 1);
    blocks.resize(c.n_upsample_rates);
    const int dil[3] = {1, 3, 9};
    for (int i = 0; i < c.n_upsample_rates; ++i) {
        const std::string p = "decoder." + std::to_string(i + 1) + ".block.";
        auto& b = blocks[i];
        b.r = c.upsample_rates[i]; b.cin = c.decoder_dim >> i; b.cout = c.decoder_dim >> (i + 1);
        make_snake(b.act, p + "0");
        make_tconv(b.tconv, p + "1.conv", b.r);
        for (int j = 0; j < 3; ++j) {
            const std::string up = p + std::to_string(j + 2) + ".";
            make_snake(b.u[j].a1, up + "act1");
            make_snake(b.u[j].a2, up + "act2");
            make_conv(b.u[j].c1, up + "conv1.conv", dil[j]);
            make_conv(b.u[j].c2, up + "conv2.conv", 1);
            // bf16 mode, C = 96 | 192 (the two blocks with the most rows): the unit runs as ONE kernel (resunit.hip) on weights
            // packed into MFMA fragments here.  QTTS_CODEC_FUSED=0 keeps the two tap-GEMM launches (A/B runs).
            const bool fused_env = QTTS_OPT_ON("QTTS_CODEC_FUSED");
            b.u[j].dil = dil[j];
            if (bf16 && fused_env && resunit_supported(b.cout)) {
                const int Cc = b.cout;
                auto& w1 = P(up + "conv1.conv.weight");            // (Cout, Cin, 7) -> [tap][n][k]
                std::vector<float> r1((size_t)7 * Cc * Cc);
                for (int t = 0; t < 7; ++t)
                    for (int n2 = 0; n2 < Cc; ++n2)
                        for (int k = 0; k < Cc; ++k) r1[((size_t)t * Cc + n2) * Cc + k] = w1[((size_t)n2 * Cc + k) * 7 + t];
                std::vector<bf16_t> h1(resunit_packed_elems(Cc, 7)), h2(resunit_packed_elems(Cc, 1));
                pack_resunit_weight(r1.data(), Cc, 7, true, h1.data());
                pack_resunit_weight(P(up + "conv2.conv.weight").data(), Cc, 1, false, h2.data());     // (Cout, Cin, 1) == [n][k]
                b.u[j].w1p.upload(h1.data(), h1.size() * 2);
                b.u[j].w2p.upload(h2.data(), h2.size() * 2);
                b.u[j].fused = true;
            }
        }
    }
    const int n = c.n_upsample_rates;
    make_snake(final_act, "decoder." + std::to_string(n + 1));
    {
        const std::string p = "decoder." + std::to_string(n + 2) + ".conv";
        auto& w = P(p + ".weight"); auto& s = shapes[p + ".weight"];
        final_c = (int)s[1];
        QTTS_REQUIRE(s[0] == 1 && s[2] == 7, QTTS_ERR_ARG, "final conv must be (1, C, 7)");
        std::vector<float> r((size_t)7 * final_c);
        for (int k = 0; k < 7; ++k) for (int ch = 0; ch < final_c; ++ch) r[(size_t)k * final_c + ch] = w[(size_t)ch * 7 + k];
        upload_f(final_w, r);
        final_b = P(p + ".bias")[0];
    }
    // workspaces: the largest activation of any stage, per frame
    size_t per_frame = 0;
    {
        size_t L = 1;
        per_frame = std::max<size_t>({(size_t)2 * vq, (size_t)c.codebook_dim, (size_t)c.latent_dim,
                                      (size_t)(c.num_attention_heads + 2 * c.num_key_value_heads) * c.head_dim,
                                      (size_t)c.hidden_size, (size_t)c.intermediate_size});
        for (int u = 0; u < c.n_upsampling_ratios; ++u) { L *= c.upsampling_ratios[u]; per_frame = std::max(per_frame, L * 4 * c.latent_dim); }
        per_frame = std::max(per_frame, L * c.decoder_dim);
        for (int i = 0; i < c.n_upsample_rates; ++i) { L *= c.upsample_rates[i]; per_frame = std::max(per_frame, L * (size_t)(c.decoder_dim >> (i + 1))); }
    }
    buf_elems = per_frame * (size_t)std::max(1, c.max_batch) * (size_t)std::max(1, c.max_frames);
    for (auto& b : buf) b.alloc(buf_elems * sizeof(float));
    if (bf16) for (auto& b : buf16) b.alloc(buf_elems * sizeof(bf16_t));
    err_flag.alloc(4);
    QTTS_CHECK_HIP(hipMemset(err_flag.p, 0, 4));
    host.clear();  // host copies no longer needed
    finalized = true;
}

void qtts_codec::forward(const int64_t* codes, int B, int64_t sb, int64_t sq, int64_t stt, int t0, int Tc, float* wav,
                         float* pre, int64_t wav_stride_b, int64_t skip_samples, const char* stage, float* stage_out,
                         int64_t cap, int64_t* L_out, int64_t* C_out, hipStream_t st) {
    const auto& c = cfg;
    QTTS_REQUIRE(finalized, QTTS_ERR_STATE, "codec: finalize() first");
    QTTS_REQUIRE(B >= 1 && Tc >= 1, QTTS_ERR_ARG, "codec: empty input");
    QTTS_REQUIRE((size_t)B * Tc <= (size_t)c.max_batch * c.max_frames, QTTS_ERR_LIMIT,
                 "codec: B*T exceeds max_batch*max_frames given at create");
    float *x = buf[0].as<float>(), *s1 = buf[1].as<float>(), *s2 = buf[2].as<float>(), *s3 = buf[3].as<float>();
    int L = Tc;          // positions per sequence at the current stage
    int C = 0;           // channels of x
    auto want = [&](const char* name) { return stage && strcmp(stage, name) == 0; };
    auto emit = [&](const float* src) {
        const int64_t n = (int64_t)B * L * C;
        QTTS_REQUIRE(n <= cap, QTTS_ERR_ARG, "codec stage: output buffer too small");
        QTTS_CHECK_HIP(hipMemcpyAsync(stage_out, src, n * sizeof(float), hipMemcpyDeviceToDevice, st));
        if (L_out) *L_out = L;
        if (C_out) *C_out = C;
    };

    // ---- RVQ dequant: gather-sum then the two 1x1 output projections as one GEMM (v2:815-821)
    launch_rvq_gather(codes, B, c.num_quantizers, Tc, sb, sq, stt, t0, Tc, tables.as<float>(), c.codebook_size, vq, s1, err_flag.as<int>(), st);
    gemm(rvq_out, s1, 2 * vq, B * L, L, x, c.codebook_dim, ACT_NONE, nullptr, 0, nullptr, nullptr, st);
    C = c.codebook_dim;
    if (want("rvq")) { emit(x); return; }
    // ---- pre_conv (k=3 causal) (v2:874)
    gemm(pre_conv, x, C, B * L, L, s1, c.latent_dim, ACT_NONE, nullptr, 0, nullptr, nullptr, st);
    std::swap(x, s1); C = c.latent_dim;
    if (want("pre_conv")) { emit(x); return; }
    // ---- pre_transformer (v2:501-575)
    // (Round 4 ran this stage's GEMMs as weight-streaming strips for decodes of <= 128 rows -- the talker's decode GEMM with eight
    // m-tiles, norms and layer scales folded: 2.52 vs 2.44 ms at B = 1 x 10 s, i.e. slower than the tile GEMMs below; removed,
    // profiles/r04_codec_small_rows.md.)
    {
        const int H = c.hidden_size, I = c.intermediate_size;
        const int qd = c.num_attention_heads * c.head_dim, kvd = c.num_key_value_heads * c.head_dim;
        const int M = B * L;
        float* h = s1;
        gemm(in_proj, x, C, M, L, h, H, ACT_NONE, nullptr, 0, nullptr, nullptr, st);
        float *a = x, *b2 = s2, *b3 = s3;     // scratch
        for (auto& Ly : tl) {
            launch_rmsnorm(h, H, Ly.n1.as<float>(), c.rms_norm_eps, a, H, M, H, st);
            gemm(Ly.qkv, a, H, M, L, b2, qd + 2 * kvd, ACT_NONE, nullptr, 0, nullptr, nullptr, st);
            launch_rope_inplace(b2, qd + 2 * kvd, M, L, c.num_attention_heads + c.num_key_value_heads, c.head_dim,
                                inv_freq.as<float>(), st);
            AttnRowsParams ap{};
            ap.qkv = b2; ap.ld = qd + 2 * kvd; ap.q_off = 0; ap.k_off = qd; ap.v_off = qd + kvd;
            ap.B = B; ap.T = L; ap.nh = c.num_attention_heads; ap.nkv = c.num_key_value_heads; ap.hd = c.head_dim;
            ap.window = c.sliding_window; ap.n_pad = nullptr; ap.out = a; ap.ldo = qd;
            launch_attn_rows(ap, st);
            gemm(Ly.o, a, qd, M, L, h, H, ACT_NONE, h, H, Ly.ls1.as<float>(), nullptr, st);
            launch_rmsnorm(h, H, Ly.n2.as<float>(), c.rms_norm_eps, a, H, M, H, st);
            gemm(Ly.gu, a, H, M, L, b3, I, ACT_SWIGLU, nullptr, 0, nullptr, nullptr, st);
            gemm(Ly.down, b3, I, M, L, h, H, ACT_NONE, h, H, Ly.ls2.as<float>(), nullptr, st);
        }
        launch_rmsnorm(h, H, t_norm.as<float>(), c.rms_norm_eps, a, H, M, H, st);
        gemm(out_proj, a, H, M, L, b2, c.latent_dim, ACT_NONE, nullptr, 0, nullptr, nullptr, st);
        // b2 now holds the stage output; rotate so that x points at it
        float* old_x = x; x = b2; s2 = old_x; C = c.latent_dim;
    }
    if (want("pre_transformer")) { emit(x); return; }
    // ---- upsample: ConvTranspose(k = s = f) + ConvNeXt (v2:878-880)
    const bool dec0_a16_env = QTTS_OPT_ON("QTTS_CODEC_DEC0_A16");
    const bool dec0_a16 = bf16 && dec0_a16_env && !stage && c.n_upsampling_ratios > 0 && !blocks.empty() && c.latent_dim % 64 == 0 && c.decoder_dim % 32 == 0;
    for (int u = 0; u < c.n_upsampling_ratios; ++u) {
        const int f = c.upsampling_ratios[u];
        float *y = (x == s1 ? s2 : s1), *d = (x == s3 || y == s3) ? (x == s2 || y == s2 ? s1 : s2) : s3;
        // pick three distinct scratch buffers among {buf0..3} \ {x}
        float* all[4] = {buf[0].as<float>(), buf[1].as<float>(), buf[2].as<float>(), buf[3].as<float>()};
        float* fr[3]; int nf = 0;
        for (auto p : all) if (p != x) fr[nf++] = p;
        y = fr[0]; d = fr[1]; float* e = fr[2];
        gemm(ups[u].tconv, x, C, B * L, L, y, f * C, ACT_NONE, nullptr, 0, nullptr, nullptr, st);   // [B*L][f*C] == [B*L*f][C]
        L *= f;
        launch_dwconv_ln(y, ups[u].dw_w.as<float>(), ups[u].dw_b.as<float>(), ups[u].ln_w.as<float>(),
                         ups[u].ln_b.as<float>(), 1e-6f, d, B * L, L, C, st);
        gemm(ups[u].pw1, d, C, B * L, L, e, 4 * C, ACT_GELU, nullptr, 0, nullptr, nullptr, st);
        // Round 4: the last ConvNeXt block also leaves a bf16 copy of its output for decoder.0 -- a 7-tap convolution that then runs as
        // the bf16-activation tap GEMM (input tile staged once per k-slab for all taps, LDS-DMA) instead of re-reading and converting its
        // fp32 input per tap (245 us on 48 tiles at B = 1 x 10 s, profiles/r04_codec_kernel_trace_b1x125.md).  QTTS_CODEC_DEC0_A16=0: off.
        if (dec0_a16 && u + 1 == c.n_upsampling_ratios)
            gemm16(ups[u].pw2, e, nullptr, 4 * C, B * L, L, y, C, ACT_NONE, y, C, nullptr, buf16[1].as<bf16_t>(), nullptr, st, 0, nullptr, nullptr,
                   ups[u].gamma.as<float>());
        else gemm(ups[u].pw2, e, 4 * C, B * L, L, y, C, ACT_NONE, y, C, ups[u].gamma.as<float>(), nullptr, st);
        x = y;
        const std::string nm = "upsample" + std::to_string(u);
        if (want(nm.c_str())) { emit(x); return; }
    }
    auto scratch3 = [&](float*& a, float*& b, float*& cc) {
        float* all[4] = {buf[0].as<float>(), buf[1].as<float>(), buf[2].as<float>(), buf[3].as<float>()};
        float* fr[3]; int nf = 0;
        for (auto p : all) if (p != x) fr[nf++] = p;
        a = fr[0]; b = fr[1]; cc = fr[2];
    };
    // bf16 mode (round 2): inside the decoder blocks every tensor that is only a GEMM input travels as bf16 with the consumer's
    // SnakeBeta already applied by its producer's epilogue; the residual stream stays fp32.
    const bool fast16_env = QTTS_OPT_ON("QTTS_CODEC_FAST16");   // (=0: A/B)
    bool fast16 = bf16 && !blocks.empty() && fast16_env;
    for (auto& bk : blocks) fast16 = fast16 && bk.cin % 32 == 0 && bk.cout % 32 == 0;
    // QTTS_CODEC_FINAL16=0: fp32 tensor out of the last unit + stand-alone SnakeBeta + fp32 final conv (A/B runs)
    const bool final16_env = QTTS_OPT_ON("QTTS_CODEC_FINAL16");
    const bool final16 = fast16 && final16_env && !stage && wav && final_c % 8 == 0 && (size_t)262 * (final_c / 2 + 1) * 4 <= 64 * 1024;
    bf16_t* h16a = fast16 ? buf16[0].as<bf16_t>() : nullptr;
    bf16_t* h16b = fast16 ? buf16[1].as<bf16_t>() : nullptr;
    // (Residual stream: fp32 between the tap-GEMM launches of the C = 768 / 384 blocks; bf16 inside blocks whose units run fused
    // (round 3, `r16` below: on by default, relative RMS 0.0497 vs 0.0493 at real dims, pinned by the GPU test's 0.060 bar).)
    // ---- decoder.0: conv k=7 latent -> decoder_dim (v2:857)
    {
        float *a, *b, *cc; scratch3(a, b, cc);
        if (fast16 && dec0_a16) gemm16(dec0, nullptr, h16b, C, B * L, L, stage ? a : nullptr, c.decoder_dim, ACT_NONE, nullptr, 0, nullptr, h16a, &blocks[0].act, st);
        else if (fast16) gemm16(dec0, x, nullptr, C, B * L, L, a, c.decoder_dim, ACT_NONE, nullptr, 0, nullptr, h16a, &blocks[0].act, st);
        else gemm(dec0, x, C, B * L, L, a, c.decoder_dim, ACT_NONE, nullptr, 0, nullptr, nullptr, st);
        x = a; C = c.decoder_dim;
    }
    if (want("decoder0")) { emit(x); return; }
    // ---- 4 decoder blocks: SnakeBeta, ConvTranspose(2r, r), 3 residual units (v2:645-658, 628-635)
    for (size_t i = 0; i < blocks.size(); ++i) {
        auto& bk = blocks[i];
        float *a, *b, *cc; scratch3(a, b, cc);
        if (fast16) {
            // h16a = SnakeBeta_block(x) in bf16 (from the previous producer).  tconv -> b (fp32, the first unit's residual) and
            // h16a' = SnakeBeta_unit0.act1(b)
            // Round 3: inside a block whose units run fused (C = 96 / 192) the residual stream travels as bf16, as in the reference's
            // own bfloat16 mode: the fused unit is bound by its tile's HBM bytes (profiles/r03_pmc_mfma_codec.md), and 768 of its
            // 1236 B per row were the fp32 residual in and out.  It lives in the storage of the fp32 buffers `a` / `b`; fp32 comes
            // back where a tensor leaves the blocks (last unit of the last block) or a stage is asked for.
            // QTTS_CODEC_RES16=0: fp32 residual stream (A/B runs).
            const bool res16_env = QTTS_OPT_ON("QTTS_CODEC_RES16");
            const bool r16 = res16_env && !stage && bk.u[0].fused && bk.u[1].fused && bk.u[2].fused;
            gemm16(bk.tconv, nullptr, h16a, C, B * L, L, r16 ? nullptr : b, bk.r * bk.cout, ACT_NONE, nullptr, 0, nullptr, h16b, &bk.u[0].a1, st, bk.cout,
                   nullptr, r16 ? reinterpret_cast<bf16_t*>(b) : nullptr);
            std::swap(h16a, h16b);                 // h16a: activated input of unit 0; h16b: free
            L *= bk.r; C = bk.cout;
            float* cur = b; float* alt = a;        // (a was only the stand-alone snake's output in the fp32 path: free here)
            for (int j = 0; j < 3; ++j) {
                auto& un = bk.u[j];
                // the last unit of a block feeds only the next block's transposed conv (the activated bf16 copy): its residual-stream
                // output is written only where a stage was asked for.  The last unit of the last block feeds only the final
                // SnakeBeta + conv (C -> 1): the same bf16 copy with the final activation (final16; the fp32 tensor, the stand-alone
                // SnakeBeta pass over it and the fp32 read of the final conv were 16 C of the tail's 18 C bytes per output sample).
                const bool leaves = j == 2 && i + 1 == blocks.size();
                const bool to_final16 = leaves && final16;
                const Snake* next = j < 2 ? &bk.u[j + 1].a1 : (i + 1 < blocks.size() ? &blocks[i + 1].act : (to_final16 ? &final_act : nullptr));
                const bool dead = j == 2 && !stage && (!leaves || to_final16);
                if (un.fused) {                    // conv7 -> SnakeBeta_2 -> conv1x1 -> + residual in one kernel (resunit.hip)
                    ResUnitParams rp{};
                    rp.A16 = h16a; rp.lda = C; rp.ldr = C; rp.M = B * L; rp.T = L; rp.dil = un.dil; rp.Cch = C;
                    if (r16) rp.res16 = cur; else rp.res = cur;
                    rp.W1p = un.w1p.p; rp.b1 = un.c1.bias.as<float>(); rp.ea2 = un.a2.ea.as<float>(); rp.ib2 = un.a2.ib.as<float>();
                    rp.W2p = un.w2p.p; rp.b2 = un.c2.bias.as<float>();
                    rp.ldc = C;
                    if (r16 && !leaves) rp.R16 = dead ? nullptr : alt;       // bf16 stream on to the next unit ...
                    else rp.C = dead ? nullptr : alt;                        // ... fp32 where the tensor leaves the blocks as such
                    // (its bf16 output must not alias its bf16 input: a tile's halo rows are other tiles' output rows)
                    rp.C16 = next ? h16b : nullptr; rp.ldc16 = C;
                    rp.ea16 = next ? next->ea.as<float>() : nullptr; rp.ib16 = next ? next->ib.as<float>() : nullptr;
                    launch_resunit(rp, st);
                    std::swap(h16a, h16b);         // h16a: the next unit's (or next block's) activated input
                    std::swap(cur, alt);
                    continue;
                }
                gemm16(un.c1, nullptr, h16a, C, B * L, L, nullptr, C, ACT_SNAKE, nullptr, 0, &un.a2, h16b, nullptr, st);   // conv7 + act2 -> bf16
                (void)leaves;
                gemm16(un.c2, nullptr, h16b, C, B * L, L, dead ? nullptr : alt, C, ACT_NONE, cur, C, nullptr, next ? h16a : nullptr, next, st);  // 1x1 + residual
                std::swap(cur, alt);               // ping-pong between a and b: the residual input is never the output
            }
            x = cur;
        } else {
        launch_snake(x, bk.act.ea.as<float>(), bk.act.ib.as<float>(), a, (int64_t)B * L, C, st);
        gemm(bk.tconv, a, C, B * L, L, b, bk.r * bk.cout, ACT_NONE, nullptr, 0, nullptr, nullptr, st);
        L *= bk.r; C = bk.cout;
        float* cur = b;              // unit input / residual
        float* sA = a; float* sB = cc; float* alt = x;   // x's old buffer is free now
        for (int j = 0; j < 3; ++j) {
            auto& un = bk.u[j];
            launch_snake(cur, un.a1.ea.as<float>(), un.a1.ib.as<float>(), sA, (int64_t)B * L, C, st);
            gemm(un.c1, sA, C, B * L, L, sB, C, ACT_SNAKE, nullptr, 0, nullptr, &un.a2, st);        // conv7 + snake(act2)
            gemm(un.c2, sB, C, B * L, L, alt, C, ACT_NONE, cur, C, nullptr, nullptr, st);           // 1x1 + residual
            std::swap(cur, alt);
        }
        x = cur;
        }
        const std::string nm = "block" + std::to_string(i + 1);
        if (want(nm.c_str())) { emit(x); return; }
    }
    // ---- final SnakeBeta + conv(C -> 1, k=7) + clamp (v2:861-864, 884)
    QTTS_REQUIRE(C == final_c, QTTS_ERR_ARG, "final conv channel mismatch");
    if (final16) {                 // h16a = SnakeBeta_final(x) in bf16, written by the last unit
        launch_final_conv16(h16a, final_w.as<float>(), final_b, wav, pre, (int64_t)B * L, L, C, wav_stride_b, skip_samples, st);
    } else {
        float *a, *b, *cc; scratch3(a, b, cc);
        launch_snake(x, final_act.ea.as<float>(), final_act.ib.as<float>(), a, (int64_t)B * L, C, st);
        if (wav)
            launch_final_conv(a, final_w.as<float>(), final_b, wav, pre, (int64_t)B * L, L, C, wav_stride_b, skip_samples, st);
    }
    if (L_out) *L_out = L;
    if (C_out) *C_out = 1;
}

// ------------------------------------------------------------------------------------------ streaming decode
// The decoder is causal end to end (v2:159-208, 491), so a packet of n new frames needs, per stateful layer, only the
// last (k-1)*dilation input rows of what came before (one row for the k = 2r transposed convs, window-1 roped k/v rows
// per transformer layer): algorithm and its equality with the whole-sequence forward in oracle/codec_stream_ref.py.
// Every stateful layer is run by the UNCHANGED kernels of forward() on a staged buffer [carried rows | new rows]; the
// outputs of the carried rows are discarded by the next staging step (`skip`).
void qtts_codec::stream_begin(int B) {
    QTTS_REQUIRE(finalized, QTTS_ERR_STATE, "codec: finalize() first");
    QTTS_REQUIRE(B >= 1 && B <= cfg.max_batch, QTTS_ERR_LIMIT, "codec stream: batch exceeds max_batch");
    const auto& c = cfg;
    std::vector<std::pair<int, int>> want;            // (rows, channels) per stateful layer, in execution order
    want.push_back({2, c.codebook_dim});                                                          // pre_conv k=3
    const int qkvw = (c.num_attention_heads + 2 * c.num_key_value_heads) * c.head_dim;
    for (int l = 0; l < c.num_hidden_layers; ++l) want.push_back({c.sliding_window - 1, qkvw});   // roped q|k|v rows
    for (int u = 0; u < c.n_upsampling_ratios; ++u) want.push_back({6, c.latent_dim});            // ConvNeXt dwconv k=7
    want.push_back({6, c.latent_dim});                                                            // decoder.0 k=7
    const int dil[3] = {1, 3, 9};
    for (int i = 0; i < c.n_upsample_rates; ++i) {
        want.push_back({1, c.decoder_dim >> i});                                                  // transposed conv k=2r
        for (int j = 0; j < 3; ++j) want.push_back({6 * dil[j], c.decoder_dim >> (i + 1)});       // residual unit conv7
    }
    want.push_back({6, c.decoder_dim >> c.n_upsample_rates});                                     // final conv k=7
    carry.resize(want.size());
    for (size_t i = 0; i < want.size(); ++i) {
        carry[i].h = want[i].first; carry[i].C = want[i].second;
        const size_t bytes = (size_t)B * want[i].first * want[i].second * sizeof(float);
        carry[i].d.ensure(std::max<size_t>(bytes, 16));
        QTTS_CHECK_HIP(hipMemset(carry[i].d.p, 0, std::max<size_t>(bytes, 16)));   // zeros == the causal left padding
    }
    // hipMemset runs on the null stream and may return before it has executed; stream_push() launches on the caller's
    // stream, which does not order against the null stream when it is a non-blocking one (PyTorch's pool streams are).
    QTTS_CHECK_HIP(hipDeviceSynchronize());
    stream_npad.ensure((size_t)B * sizeof(int));
    stream_npad_host.assign(B, 0);
    stream_B = B;
    stream_t = 0;
}

void qtts_codec::stream_push(const int64_t* codes, int n, float* wav, hipStream_t st) {
    const auto& c = cfg;
    QTTS_REQUIRE(stream_B > 0, QTTS_ERR_STATE, "codec stream: stream_begin() first");
    QTTS_REQUIRE(n >= 1 && codes && wav, QTTS_ERR_ARG, "codec stream: empty packet");
    const int B = stream_B;
    float* pool[4] = {buf[0].as<float>(), buf[1].as<float>(), buf[2].as<float>(), buf[3].as<float>()};
    auto fits = [&](int64_t rows, int C) {
        QTTS_REQUIRE((size_t)rows * (size_t)C <= buf_elems, QTTS_ERR_LIMIT,
                     "codec stream: packet + carried rows exceed the workspace (raise max_frames at create)");
    };
    auto other = [&](std::initializer_list<const float*> busy) -> float* {     // a workspace buffer not in `busy`
        for (float* p : pool) {
            bool used = false;
            for (const float* q : busy) used = used || q == p;
            if (!used) return p;
        }
        throw Error(QTTS_ERR_STATE, "codec stream: out of workspace buffers");
    };
    size_t ci = 0;                                   // next carry slot
    // x: current activation, T rows per sequence of which the first `skip` are to be ignored, C channels
    float* x = nullptr; int T = n, skip = 0, C = 0;
    auto stage = [&](float* dst) {                   // dst = [carry | valid rows of x]; refresh the carry; x <- dst
        Carry& k = carry[ci++];
        QTTS_REQUIRE(k.C == C, QTTS_ERR_STATE, "codec stream: carry/channel mismatch");
        const int nv = T - skip;
        fits((int64_t)B * (k.h + nv), C);
        launch_stage_rows(x, T, skip, nv, k.d.as<float>(), k.h, dst, B, C, st);
        launch_save_tail(dst, k.h + nv, k.d.as<float>(), k.h, B, C, st);
        x = dst; T = k.h + nv; skip = k.h;
    };
    auto compact = [&](float* dst) {                 // drop the ignored rows
        const int nv = T - skip;
        launch_stage_rows(x, T, skip, nv, nullptr, 0, dst, B, C, st);
        x = dst; T = nv; skip = 0;
    };

    // ---- RVQ dequant (stateless)
    float* g = pool[1];
    x = pool[0]; C = c.codebook_dim;
    fits((int64_t)B * n, std::max(2 * vq, C));
    launch_rvq_gather(codes, B, c.num_quantizers, n, (int64_t)c.num_quantizers * n, n, 1, 0, n, tables.as<float>(),
                      c.codebook_size, vq, g, err_flag.as<int>(), st);
    gemm(rvq_out, g, 2 * vq, B * n, n, x, C, ACT_NONE, nullptr, 0, nullptr, nullptr, st);
    // ---- pre_conv k=3
    {
        float* s = other({x}); stage(s);
        float* y = other({x});
        fits((int64_t)B * T, c.latent_dim);
        gemm(pre_conv, x, C, B * T, T, y, c.latent_dim, ACT_NONE, nullptr, 0, nullptr, nullptr, st);
        x = y; C = c.latent_dim;
        compact(other({x}));
    }
    // ---- pre_transformer with a (window-1)-row KV carry per layer
    {
        const int H = c.hidden_size, I = c.intermediate_size, W1 = c.sliding_window - 1;
        const int qd = c.num_attention_heads * c.head_dim, kvd = c.num_key_value_heads * c.head_dim, qw = qd + 2 * kvd;
        const int M = B * n;
        const int pad = W1 - (int)std::min<int64_t>(stream_t, W1);          // carried rows that do not exist yet
        for (int b = 0; b < B; ++b) stream_npad_host[b] = pad;
        QTTS_CHECK_HIP(hipMemcpyAsync(stream_npad.p, stream_npad_host.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice, st));
        float* h = other({x});
        fits((int64_t)B * (W1 + n), std::max({qw, H, I}));
        gemm(in_proj, x, C, M, n, h, H, ACT_NONE, nullptr, 0, nullptr, nullptr, st);
        float* a = other({h});
        float* b2 = other({h, a});
        for (auto& Ly : tl) {
            launch_rmsnorm(h, H, Ly.n1.as<float>(), c.rms_norm_eps, a, H, M, H, st);
            gemm(Ly.qkv, a, H, M, n, b2, qw, ACT_NONE, nullptr, 0, nullptr, nullptr, st);
            launch_rope_offset(b2, qw, M, n, (int)stream_t, c.num_attention_heads + c.num_key_value_heads, c.head_dim,
                               inv_freq.as<float>(), st);
            // a <- [carried roped q|k|v rows | new rows]
            Carry& k = carry[ci++];
            QTTS_REQUIRE(k.C == qw && k.h == W1, QTTS_ERR_STATE, "codec stream: KV carry mismatch");
            launch_stage_rows(b2, n, 0, n, k.d.as<float>(), W1, a, B, qw, st);
            launch_save_tail(a, W1 + n, k.d.as<float>(), W1, B, qw, st);
            AttnRowsParams ap{};
            ap.qkv = a; ap.ld = qw; ap.q_off = 0; ap.k_off = qd; ap.v_off = qd + kvd;
            ap.B = B; ap.T = W1 + n; ap.nh = c.num_attention_heads; ap.nkv = c.num_key_value_heads; ap.hd = c.head_dim;
            ap.window = c.sliding_window; ap.n_pad = stream_npad.as<int>(); ap.out = b2; ap.ldo = qd;
            launch_attn_rows(ap, st);                                       // b2: [B][W1+n][qd], rows >= W1 are the new ones
            launch_stage_rows(b2, W1 + n, W1, n, nullptr, 0, a, B, qd, st);  // a: compact [B*n][qd]
            gemm(Ly.o, a, qd, M, n, h, H, ACT_NONE, h, H, Ly.ls1.as<float>(), nullptr, st);
            launch_rmsnorm(h, H, Ly.n2.as<float>(), c.rms_norm_eps, a, H, M, H, st);
            gemm(Ly.gu, a, H, M, n, b2, I, ACT_SWIGLU, nullptr, 0, nullptr, nullptr, st);
            gemm(Ly.down, b2, I, M, n, h, H, ACT_NONE, h, H, Ly.ls2.as<float>(), nullptr, st);
        }
        launch_rmsnorm(h, H, t_norm.as<float>(), c.rms_norm_eps, a, H, M, H, st);
        gemm(out_proj, a, H, M, n, b2, c.latent_dim, ACT_NONE, nullptr, 0, nullptr, nullptr, st);
        x = b2; T = n; skip = 0; C = c.latent_dim;
    }
    // ---- upsample: ConvTranspose(k = s = f, column-local) + ConvNeXt (dwconv k=7 carries 6 rows)
    for (int u = 0; u < c.n_upsampling_ratios; ++u) {
        const int f = c.upsampling_ratios[u];
        float* y = other({x});
        fits((int64_t)B * T * f, 4 * C);
        gemm(ups[u].tconv, x, C, B * T, T, y, f * C, ACT_NONE, nullptr, 0, nullptr, nullptr, st);
        x = y; T *= f; skip *= f;
        float* s = other({x}); stage(s);                                   // x = staged y (also the residual)
        float* d = other({x});
        float* e = other({x, d});
        fits((int64_t)B * T, 4 * C);
        launch_dwconv_ln(x, ups[u].dw_w.as<float>(), ups[u].dw_b.as<float>(), ups[u].ln_w.as<float>(),
                         ups[u].ln_b.as<float>(), 1e-6f, d, B * T, T, C, st);
        gemm(ups[u].pw1, d, C, B * T, T, e, 4 * C, ACT_GELU, nullptr, 0, nullptr, nullptr, st);
        gemm(ups[u].pw2, e, 4 * C, B * T, T, x, C, ACT_NONE, x, C, ups[u].gamma.as<float>(), nullptr, st);
    }
    // ---- decoder.0 conv k=7
    {
        float* s = other({x}); stage(s);
        float* y = other({x});
        fits((int64_t)B * T, c.decoder_dim);
        gemm(dec0, x, C, B * T, T, y, c.decoder_dim, ACT_NONE, nullptr, 0, nullptr, nullptr, st);
        x = y; C = c.decoder_dim;
    }
    // ---- decoder blocks
    for (size_t i = 0; i < blocks.size(); ++i) {
        auto& bk = blocks[i];
        {   // SnakeBeta (elementwise) + ConvTranspose(2r, r): output block t mixes input rows t and t-1
            float* s = other({x}); stage(s);                               // carry = 1 pre-activation row
            float* a = other({x});
            launch_snake(x, bk.act.ea.as<float>(), bk.act.ib.as<float>(), a, (int64_t)B * T, C, st);
            float* b = other({x, a});
            fits((int64_t)B * T * bk.r, bk.cout);
            gemm(bk.tconv, a, C, B * T, T, b, bk.r * bk.cout, ACT_NONE, nullptr, 0, nullptr, nullptr, st);
            x = b; T *= bk.r; skip *= bk.r; C = bk.cout;
        }
        for (int j = 0; j < 3; ++j) {
            auto& un = bk.u[j];
            float* s = other({x}); stage(s);                               // x = [carried unit inputs | new], also the residual
            float* sA = other({x});
            float* sB = other({x, sA});
            launch_snake(x, un.a1.ea.as<float>(), un.a1.ib.as<float>(), sA, (int64_t)B * T, C, st);
            gemm(un.c1, sA, C, B * T, T, sB, C, ACT_SNAKE, nullptr, 0, nullptr, &un.a2, st);
            gemm(un.c2, sB, C, B * T, T, sA, C, ACT_NONE, x, C, nullptr, nullptr, st);
            x = sA;
        }
    }
    // ---- final SnakeBeta + conv(C -> 1, k=7) + clamp
    {
        float* s = other({x}); stage(s);
        float* a = other({x});
        launch_snake(x, final_act.ea.as<float>(), final_act.ib.as<float>(), a, (int64_t)B * T, C, st);
        QTTS_REQUIRE(C == final_c, QTTS_ERR_ARG, "final conv channel mismatch");
        launch_final_conv(a, final_w.as<float>(), final_b, wav, nullptr, (int64_t)B * T, T, C, (int64_t)n * up_total, skip, st);
    }
    QTTS_REQUIRE(ci == carry.size(), QTTS_ERR_STATE, "codec stream: carry bookkeeping out of step");
    stream_t += n;
}

// ============================================================================================ C ABI
namespace qtts {
thread_local std::string g_last_error;
void set_last_error(const std::string& s) { g_last_error = s; }

// the A/B switch table (common.h: QTTS_ENV; include/qtts.h: qtts_set_option)
namespace {
struct OptTable {
    std::mutex m;
    std::map<std::string, std::string> kv;
    // the environment's value of a switch AS IT WAS WHEN THE LIBRARY FIRST LOOKED (include/qtts.h): one getenv per name for the life of the
    // process, under the table's lock -- later lookups never touch the environment again (a putenv from another thread cannot race a launch)
    std::map<std::string, std::pair<bool, std::string>> env;
    std::atomic<unsigned> gen{1};
    const std::pair<bool, std::string>& env_once(const char* var) {          // (caller holds m)
        auto it = env.find(var);
        if (it == env.end()) {
            const char* e = getenv(var);
            it = env.emplace(var, std::make_pair(e != nullptr, std::string(e ? e : ""))).first;
        }
        return it->second;
    }
};
OptTable& opt_table() { static OptTable t; return t; }
}  // namespace
void ensure_dynamic_lds(const void* kern, int bytes) {
    static std::mutex m;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    QTTS_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(m);
    if (done.count({kern, dev})) return;
    QTTS_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.insert({kern, dev});
}
const char* opt_lookup(const char* var, unsigned& gen_seen, std::string& cache, bool& has) {
    auto& t = opt_table();
    const unsigned cur = t.gen.load(std::memory_order_acquire);
    if (cur != gen_seen) {
        std::lock_guard<std::mutex> lk(t.m);
        auto it = t.kv.find(var);
        if (it != t.kv.end()) { cache = it->second; has = true; }
        else {
            const auto& e = t.env_once(var);
            if (e.first) { cache = e.second; has = true; }
            else { cache.clear(); has = false; }
        }
        gen_seen = t.gen.load(std::memory_order_relaxed);
    }
    return has ? cache.c_str() : nullptr;
}
}  // namespace qtts

#define QTTS_API_BEGIN try {
#define QTTS_API_END                                                        \
    }                                                                       \
    catch (const qtts::Error& e) { qtts::set_last_error(e.what()); return e.code; } \
    catch (const std::exception& e) { qtts::set_last_error(e.what()); return QTTS_ERR_ARG; } \
    return QTTS_OK;

extern "C" {

const char* qtts_last_error(void) { return qtts::g_last_error.c_str(); }
int qtts_abi_version(void) { return QTTS_ABI_VERSION; }

int qtts_set_option(const char* name, const char* value) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(name && strncmp(name, "QTTS_", 5) == 0 && strlen(name) < 64, QTTS_ERR_ARG, "set_option: the switch names start with QTTS_");
    auto& t = qtts::opt_table();
    std::lock_guard<std::mutex> lk(t.m);
    if (value) t.kv[name] = value; else t.kv.erase(name);
    t.gen.fetch_add(1, std::memory_order_release);
    QTTS_API_END
}
int qtts_get_option(const char* name, char* buf, int32_t cap) {
    bool set = false;
    try {                                    // (nothing may cross the C ABI: the map lookups allocate)
        if (!name || !buf || cap < 1) { qtts::set_last_error("get_option: null argument"); return QTTS_ERR_ARG; }
        auto& t = qtts::opt_table();
        std::lock_guard<std::mutex> lk(t.m);
        auto it = t.kv.find(name);
        const char* v = nullptr;
        if (it != t.kv.end()) v = it->second.c_str();
        else { const auto& e = t.env_once(name); if (e.first) v = e.second.c_str(); }
        snprintf(buf, (size_t)cap, "%s", v ? v : "");
        set = v != nullptr;
    }
    catch (const std::exception& e) { qtts::set_last_error(e.what()); return QTTS_ERR_ARG; }
    return set ? QTTS_OK : 1;
}

int qtts_codec_create(const qtts_codec_config* cfg, qtts_codec** out) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(cfg && out, QTTS_ERR_ARG, "null argument");
    QTTS_REQUIRE(cfg->compute_dtype == QTTS_F32 || cfg->compute_dtype == QTTS_BF16, QTTS_ERR_ARG, "compute_dtype");
    QTTS_REQUIRE(cfg->n_upsample_rates >= 1 && cfg->n_upsample_rates <= 8 && cfg->n_upsampling_ratios >= 0 &&
                     cfg->n_upsampling_ratios <= 8, QTTS_ERR_ARG, "upsample lists");
    QTTS_REQUIRE(cfg->codebook_dim % 64 == 0 && cfg->latent_dim % 32 == 0 && cfg->hidden_size % 32 == 0 &&
                     cfg->intermediate_size % 32 == 0 && (cfg->decoder_dim >> cfg->n_upsample_rates) % 32 == 0,
                 QTTS_ERR_ARG, "channel counts must be multiples of 32");
    int ndev = 0;
    QTTS_CHECK_HIP(hipGetDeviceCount(&ndev));
    QTTS_REQUIRE(ndev > 0, QTTS_ERR_HIP, "no HIP device");
    auto* c = new qtts_codec();
    c->cfg = *cfg;
    c->bf16 = cfg->compute_dtype == QTTS_BF16;
    *out = c;
    QTTS_API_END
}
void qtts_codec_destroy(qtts_codec* c) { delete c; }

int qtts_codec_bind(qtts_codec* c, const char* name, const void* host, int32_t src_dtype, int32_t ndim,
                    const int64_t* shape) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(c && name && host && shape, QTTS_ERR_ARG, "null argument");
    QTTS_REQUIRE(!c->finalized, QTTS_ERR_STATE, "bind after finalize");
    QTTS_REQUIRE(src_dtype == QTTS_F32 || src_dtype == QTTS_BF16, QTTS_ERR_ARG, "src_dtype");
    HostTensor t{host, src_dtype, std::vector<int64_t>(shape, shape + ndim)};
    c->host[name] = t.to_f32();
    c->shapes[name] = t.shape;
    QTTS_API_END
}
int qtts_codec_finalize(qtts_codec* c) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(c, QTTS_ERR_ARG, "null handle");
    c->finalize();
    QTTS_API_END
}
int qtts_codec_forward(qtts_codec* c, const int64_t* codes_dev, int32_t B, int32_t T, float* wav_dev,
                       float* pre_clamp_dev, void* stream) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(c && codes_dev && wav_dev, QTTS_ERR_ARG, "null argument");
    const int Q = c->cfg.num_quantizers;
    const size_t cbytes = (size_t)B * Q * T * sizeof(int64_t), wbytes = (size_t)B * T * c->up_total * sizeof(float);
    if (qtts_codec::graph_enabled()) c->ensure_staging(cbytes, wbytes, pre_clamp_dev != nullptr);
    auto run = [&](const int64_t* cd, float* wv, float* pr, hipStream_t s) {
        c->forward(cd, B, (int64_t)Q * T, T, 1, 0, T, wv, pr, (int64_t)T * c->up_total, 0, nullptr, nullptr, 0, nullptr, nullptr, s);
    };
    c->run_graphed({B, T, 0, -1, pre_clamp_dev ? 1 : 0}, (hipStream_t)stream,
        [&](hipStream_t s) { run(codes_dev, wav_dev, pre_clamp_dev, s); },
        [&](hipStream_t s) { run(c->g_codes.as<int64_t>(), c->g_wav.as<float>(), pre_clamp_dev ? c->g_pre.as<float>() : nullptr, s); },
        [&](hipStream_t s) { QTTS_CHECK_HIP(hipMemcpyAsync(c->g_codes.p, codes_dev, cbytes, hipMemcpyDeviceToDevice, s)); },
        [&](hipStream_t s) {
            QTTS_CHECK_HIP(hipMemcpyAsync(wav_dev, c->g_wav.p, wbytes, hipMemcpyDeviceToDevice, s));
            if (pre_clamp_dev) QTTS_CHECK_HIP(hipMemcpyAsync(pre_clamp_dev, c->g_pre.p, wbytes, hipMemcpyDeviceToDevice, s));
        });
    c->check_codes_flag((hipStream_t)stream);
    QTTS_API_END
}
int qtts_codec_forward_stage(qtts_codec* c, const int64_t* codes_dev, int32_t B, int32_t T, const char* stage,
                             float* out_dev, int64_t cap, int64_t* L, int64_t* C, void* stream) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(c && codes_dev && stage && out_dev, QTTS_ERR_ARG, "null argument");
    const int Q = c->cfg.num_quantizers;
    int64_t l = -1, ch = -1;
    c->forward(codes_dev, B, (int64_t)Q * T, T, 1, 0, T, nullptr, nullptr, 0, 0, stage, out_dev, cap, &l, &ch,
               (hipStream_t)stream);
    c->check_codes_flag((hipStream_t)stream);
    QTTS_REQUIRE(ch > 1 || strcmp(stage, "none") == 0, QTTS_ERR_NAME, std::string("unknown stage: ") + stage);
    if (L) *L = l;
    if (C) *C = ch;
    QTTS_API_END
}
int qtts_codec_decode(qtts_codec* c, const int64_t* codes_dev, int32_t B, int32_t T, int32_t chunk_size,
                      int32_t left_context, float* wav_dev, int64_t* lengths_host, void* stream) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(c && codes_dev && wav_dev, QTTS_ERR_ARG, "null argument");
    QTTS_REQUIRE(chunk_size >= 1 && left_context >= 0, QTTS_ERR_ARG, "chunk sizes");
    hipStream_t st = (hipStream_t)stream;
    const int Q = c->cfg.num_quantizers;
    const int64_t up = c->up_total;
    if (lengths_host) {  // audio_lengths = (codes[..., 0] > -1).sum(1) * decode_upsample_rate (v2:1012)
        std::vector<int64_t> h((size_t)B * T * Q);
        QTTS_CHECK_HIP(hipMemcpyAsync(h.data(), codes_dev, h.size() * 8, hipMemcpyDeviceToHost, st));
        QTTS_CHECK_HIP(hipStreamSynchronize(st));
        for (int b = 0; b < B; ++b) {
            int64_t n = 0;
            for (int t = 0; t < T; ++t) n += h[((size_t)b * T + t) * Q] > -1;
            lengths_host[b] = n * up;
        }
        for (int64_t v : h)   // the reference's embedding lookup raises on an index past the codebook
            QTTS_REQUIRE(v < c->cfg.codebook_size, QTTS_ERR_ARG, "codec: code index out of range (>= codebook_size)");
    }
    const size_t cbytes = (size_t)B * T * Q * sizeof(int64_t), wbytes = (size_t)B * T * up * sizeof(float);
    if (qtts_codec::graph_enabled()) c->ensure_staging(cbytes, wbytes, false);
    auto run = [&](const int64_t* cd, float* wv, hipStream_t s) {
        int start = 0;
        while (start < T) {  // chunked_decode (v2:886-896)
            const int end = std::min(start + chunk_size, T);
            const int ctx = (start - left_context > 0) ? left_context : start;
            c->forward(cd, B, (int64_t)T * Q, 1, Q, start - ctx, end - (start - ctx), wv + (int64_t)start * up, nullptr,
                       (int64_t)T * up, (int64_t)ctx * up, nullptr, nullptr, 0, nullptr, nullptr, s);
            start = end;
        }
    };
    c->run_graphed({B, T, chunk_size, left_context, 0}, st,
        [&](hipStream_t s) { run(codes_dev, wav_dev, s); },
        [&](hipStream_t s) { run(c->g_codes.as<int64_t>(), c->g_wav.as<float>(), s); },
        [&](hipStream_t s) { QTTS_CHECK_HIP(hipMemcpyAsync(c->g_codes.p, codes_dev, cbytes, hipMemcpyDeviceToDevice, s)); },
        [&](hipStream_t s) { QTTS_CHECK_HIP(hipMemcpyAsync(wav_dev, c->g_wav.p, wbytes, hipMemcpyDeviceToDevice, s)); });
    if (!lengths_host) c->check_codes_flag(st);      // (with lengths the codes were validated on the host above)
    QTTS_API_END
}

int qtts_codec_get_stats(qtts_codec* c, qtts_codec_stats* out) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(c && out, QTTS_ERR_ARG, "null argument");
    out->graph_captures = c->graph_captures; out->graph_replays = c->graph_replays;
    out->graphs_cached = 0; out->graph_nodes_last = c->graph_nodes_replayed;
    for (auto& kv : c->graphs) if (kv.second.ge) ++out->graphs_cached;
    QTTS_API_END
}

int qtts_codec_stream_begin(qtts_codec* c, int32_t B) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(c, QTTS_ERR_ARG, "null handle");
    c->stream_begin(B);
    QTTS_API_END
}
int qtts_codec_stream_push(qtts_codec* c, const int64_t* codes_dev, int32_t n_frames, float* wav_dev, void* stream) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(c && codes_dev && wav_dev, QTTS_ERR_ARG, "null argument");
    c->stream_push(codes_dev, n_frames, wav_dev, (hipStream_t)stream);
    c->check_codes_flag((hipStream_t)stream);      // a bad code must fail THIS call (and not poison the next one through a stale flag)
    QTTS_API_END
}

}  // extern "C"
