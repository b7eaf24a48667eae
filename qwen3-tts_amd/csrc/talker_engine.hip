// talker_engine.hip -- host-side orchestration of the autoregressive speech-token decoder on gfx950:
// talker prefill, the per-frame step (15-pass code predictor -> 16-way embedding sum -> 28-layer talker
// -> codec_head -> HF sampler) and its hipGraph capture.
//
// Everything that changes from frame to frame (KV length, positions, generation_step, finished flags,
// token history, RNG offset) lives in device memory and is advanced by kernels, so one captured graph
// is replayed for every frame; the host only polls a `done` flag every few frames.
//
// HBM layout:
//   weights   decode: packed 1-KiB MFMA tiles (skinny.hip), fp32 or bf16; prefill: row-major [N][K] copy
//   KV cache  paged, 16 tokens per page: pool[layer][page][kv_head][16][128] (fp32 or bf16), page table per
//             sequence reserved at create for max_seq tokens (no growth inside the captured step)
//   state     residual stream / qkv / mlp activations fp32 [rows <= 64][dim]; rows = t*B + b
#include <atomic>
#include <map>
#include <mutex>
#include <algorithm>
#include "common.h"
#include "kernels.h"
#include "glue.h"


using namespace qtts;

namespace {

struct LayerW {
    int fs_o = 16, fs_d = 16;         // features per strip of o_p / d_p
    DevBuf qkv_p, o_p, gu_p, d_p;     // packed (decode)
    DevBuf gu_p8;                     // gate|up packed with 8-row interleave (ACT_SWIGLU8: the batch <= 8 kernel, twice the workgroups); bf16 only
    DevBuf o_p16;                     // o-projection in 16-feature strips for the fused attention + o-projection launch (code predictor, bf16 only)
    DevBuf gu_mlp, d_p16;             // the fused MLP launch's operators (cp_mlp.hip: gate|up by workgroup, down in 16-feature strips; code predictor, bf16 only)
    DevBuf qkv_r, o_r, gu_r, d_r;     // row-major (prefill, talker only)
    DevBuf g1, g2, qn, kn;
};
struct StackDims { int H, I, nh, nkv, hd, qd, kvd; float eps; };

}  // namespace

struct qtts_talker {
    qtts_talker_config cfg;
    bool bf16 = false, finalized = false, prefilled = false;
    std::map<std::string, std::vector<float>> host;
    std::map<std::string, std::vector<int64_t>> shapes;

    StackDims td, cd;
    std::vector<LayerW> tl, cl;
    DevBuf t_norm, c_norm, head_p, emb_talker, emb_cp, proj_p, proj_b, inv_freq_t, inv_freq_c;
    DevBuf rope_cs_c;                   // code predictor: cos | sin of its 17 possible positions (launch_rope_table)
    DevBuf cp_qkv0_tab;   // [G-2][cp_vocab][q|k|v width] = layer-0 qkv GEMM (norm folded) of the pass input row of every token
    bool skip_qkv = false;
    DevBuf emb_cp_proj;   // [G-2][cp_vocab][cp H] = small_to_mtp_projection(codec_embedding[g](v)), built at finalize by the decode GEMM itself
    std::vector<DevBuf> lm_head_p;
    int fs_proj = 16, fs_lm = 16, fs_head = 16;
    DevBuf tp_fc1, tp_b1, tp_fc2, tp_b2;
    bool has_proj = false, has_text_proj = false, has_text_emb = false;
    DevBuf emb_text, tp_in, err_flag;          // text embedding table (prompt assembly), gathered rows, device error flag
    int64_t text_vocab = 0;
    double weight_bytes_frame = 0;

    // KV caches
    DevBuf kpool_t, vpool_t, kpool_c, vpool_c, ptab_t, ptab_c, attn_part;
    int attn_nsplit = 1;               // talker decode attention: workgroups per (sequence, kv head); > 1 when max_seq > 512
    KvCache kv_t, kv_c;
    // decode state / scratch
    DevBuf x, qkv, att, act, logits, past_hidden, cp_in, cp_x, cp_qkv, cp_att, cp_act, cp_logits, x16, cp_x16, ph16, cp_in16;
    DevBuf cur_tok, sub, generated, ss_rows, ints, n_pad_d, suppress, trailing, tts_pad;
    // prefill scratch
    DevBuf pf_x, pf_n, pf_qkv, pf_att, pf_act, tp_tmp;
    StepState ss;
    int B = 0, T0 = 0, Tt = 0, gen_cap = 0;
    // graph
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    // everything a captured frame step bakes into its kernel arguments and that can differ between generate() calls;
    // the graph is re-captured only when this changes (the Philox seed lives in device memory for the same reason)
    struct GraphKey {
        int B, Tt, eos, min_new, max_new, max_frames;
        int do_sample, top_k, sub_do_sample, sub_top_k;
        float top_p, temperature, rep, sub_top_p, sub_temperature;
        const void *codes, *hidden, *trailing, *tts_pad, *generated;
        bool operator==(const GraphKey& o) const { return memcmp(this, &o, sizeof(GraphKey)) == 0; }
    } graph_key;
    DevBuf seed_d;
    // resumable generation (qtts_talker_stream_*): everything a later call needs to keep stepping the same request
    struct StreamGen {
        bool active = false;
        qtts_sampling sp{};
        int eos = 0, min_new = 0, max_new = 0, max_frames = 0, launched = 0, done = 0;
        int64_t* codes = nullptr; float* hidden = nullptr;
    } sg;
    int graph_nodes = 0;
    // teacher forcing (diagnostic mode, eager only): qtts_talker_set_teacher
    struct Teacher { const int64_t* codes = nullptr; int F = 0; int* own = nullptr; const int* slots = nullptr; float* trace = nullptr; } tf;
    TeacherParams teacher_params() {
        TeacherParams p{};
        p.codes = tf.codes; p.F = tf.F; p.G = cfg.num_code_groups; p.B = B; p.V = cfg.vocab_size; p.own = tf.own; p.slots = tf.slots;
        p.trace = tf.trace; p.logits = logits.as<float>(); p.cur_tok = cur_tok.as<int>(); p.sub = sub.as<int>();
        p.sub_stride = cfg.num_code_groups; p.generated = generated.as<int>(); p.gen_stride = gen_cap; p.st = ss;
        return p;
    }
    // profiling of the dominant kernel (bench.py's roofline leg).  profile = 1: real frame steps run eagerly and EVERY decode-GEMM
    // launch of frames 1..PROF_FRAMES carries its own start / stop event pair (hipExtLaunchKernelGGL: the kernel's own begin / end
    // timestamps, what rocprofv3's kernel trace reports), aggregated per GEMM class afterwards.  profile = 2: round 2's measurement
    // -- only the decode-GEMM launches of one frame step re-captured as a hipGraph and replayed in isolation (kept for continuity).
    int profile = 0;
    bool timing_now = false, skinny_only = false;
    static constexpr int PROF_FRAMES = 6;
    struct LaunchEv { hipEvent_t a, b; int stack, N, K; double bytes = 0; };      // bytes: algorithmic bytes when N * K * element size is not it (the fused launch: two operators)
    std::vector<LaunchEv> ev;
    int cur_stack = 0;                 // which part of the frame step is being launched: 0 talker layers, 1 code predictor, 2 talker head
    std::vector<qtts_gemm_class> prof_classes;
    double prof_ms = 0; int64_t prof_launches = 0;
    int frames_run = 0;
    void release_events() {
        for (auto& e : ev) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
        ev.clear();
    }
    // after the stream has drained: per-class totals of the per-launch times
    void aggregate_profile() {
        prof_classes.clear();
        prof_ms = 0; prof_launches = 0;
        const double eb = bf16 ? 2.0 : 4.0;
        for (auto& e : ev) {
            float ms = 0.f;
            QTTS_CHECK_HIP(hipEventElapsedTime(&ms, e.a, e.b));
            qtts_gemm_class* c = nullptr;
            for (auto& q : prof_classes) if (q.stack == e.stack && q.N == e.N && q.K == e.K) { c = &q; break; }
            if (!c) {
                qtts_gemm_class q{};
                q.stack = e.stack; q.N = e.N; q.K = e.K; q.bytes_per_launch = e.bytes > 0 ? e.bytes : eb * (double)e.N * e.K;
                q.min_us = 1e30;
                prof_classes.push_back(q);
                c = &prof_classes.back();
            }
            c->launches += 1; c->total_ms += ms;
            c->min_us = std::min(c->min_us, 1000.0 * ms); c->max_us = std::max(c->max_us, 1000.0 * ms);
            prof_ms += ms; prof_launches += 1;
        }
        release_events();
    }

    std::vector<float>& P(const std::string& n) {
        auto it = host.find(n);
        if (it == host.end()) throw Error(QTTS_ERR_UNBOUND, "talker weight not bound: " + n);
        return it->second;
    }
    // bound tensor with its shape checked against the configuration (a mismatch must never reach a memcpy)
    std::vector<float>& PS(const std::string& n, std::initializer_list<int64_t> want) {
        auto& v = P(n);
        auto& s = shapes[n];
        if (s.size() != want.size() || !std::equal(s.begin(), s.end(), want.begin())) {
            std::string a, b;
            for (auto d : s) a += std::to_string(d) + ",";
            for (auto d : want) b += std::to_string(d) + ",";
            throw Error(QTTS_ERR_ARG, "talker weight " + n + " has shape (" + a + ") but the config implies (" + b + ")");
        }
        return v;
    }
    void upload_f(DevBuf& d, const std::vector<float>& w) { d.upload(w.data(), w.size() * 4); }
    void upload_rows(DevBuf& d, const std::vector<float>& w) {
        if (bf16) {
            std::vector<bf16_t> h(w.size());
            parallel_for((int64_t)w.size(), [&](int64_t a, int64_t b) { for (int64_t i = a; i < b; ++i) h[i] = f32_to_bf16(w[i]); });
            d.upload(h.data(), h.size() * 2);
        } else d.upload(w.data(), w.size() * 4);
    }
    void upload_packed(DevBuf& d, const std::vector<float>& w, int N, int K, const std::vector<float>* g = nullptr, int fs = 16) {
        std::vector<char> h(skinny_packed_bytes(N, K, bf16));
        pack_skinny_weight(w.data(), N, K, bf16, h.data(), g ? g->data() : nullptr, fs);     // g: folded RMSNorm weight
        d.upload(h.data(), h.size());
    }
    // Narrow strips when the GEMM would otherwise launch too few workgroups to pull its weights (bf16 kernel only).  Every
    // workgroup re-reads the whole x (M x K bf16) through its XCD's L2, so MORE workgroups also cost more (round-2 sweep,
    // profiles/r02_skinny_sweep_fs_M_temporal.txt).  Floor of 96 workgroups: measured 1.2 % faster per frame than 192 and than 48
    // (profiles/r02_ab_inproc_fs_floor.txt); QTTS_FS_MIN_WGS overrides it for A/B runs.
    int fs_min_wgs = [] { const char* e = QTTS_ENV("QTTS_FS_MIN_WGS"); return e && atoi(e) > 0 ? atoi(e) : 96; }();
    int choose_fs(int N, int K) const {
        if (!bf16) return 16;
        // a matrix of 16 MB or more is bound by what a CU can pull (~25 GB/s): it gets the full 192-workgroup floor (talker down,
        // 25 MB: 6.7 us in 256 workgroups vs 9.4 us in 128, profiles/r02_rocprofv3_kernel_trace_bench*.md); smaller ones are
        // latency-bound and pay for every extra workgroup's x re-read
        const int floor_wgs = (size_t)N * K * 2 >= ((size_t)16 << 20) ? std::max(192, fs_min_wgs) : fs_min_wgs;
        int fs = 16;
        while (fs > 4 && N / fs < floor_wgs) fs /= 2;
        return fs;
    }
    static std::vector<float> cat3(const std::vector<float>& a, const std::vector<float>& b, const std::vector<float>& c) {
        std::vector<float> w; w.reserve(a.size() + b.size() + c.size());
        w.insert(w.end(), a.begin(), a.end()); w.insert(w.end(), b.begin(), b.end()); w.insert(w.end(), c.begin(), c.end());
        return w;
    }
    static std::vector<float> interleave_gu(const std::vector<float>& g, const std::vector<float>& u, int I, int H) {
        std::vector<float> w((size_t)2 * I * H);
        for (int f = 0; f < I; ++f) {
            memcpy(&w[((size_t)(f / 16) * 32 + f % 16) * H], &g[(size_t)f * H], (size_t)H * 4);
            memcpy(&w[((size_t)(f / 16) * 32 + 16 + f % 16) * H], &u[(size_t)f * H], (size_t)H * 4);
        }
        return w;
    }
    // the same with 8-row blocks: strip j of the packed operator = gate rows 8 j .. 8 j + 7, then up rows 8 j .. 8 j + 7 (ACT_SWIGLU8)
    static std::vector<float> interleave_gu8(const std::vector<float>& g, const std::vector<float>& u, int I, int H) {
        std::vector<float> w((size_t)2 * I * H);
        for (int f = 0; f < I; ++f) {
            memcpy(&w[((size_t)(f / 8) * 16 + f % 8) * H], &g[(size_t)f * H], (size_t)H * 4);
            memcpy(&w[((size_t)(f / 8) * 16 + 8 + f % 8) * H], &u[(size_t)f * H], (size_t)H * 4);
        }
        return w;
    }
    // Round 3 (profiles/r03_ab_swiglu8.md): at batch <= 8 the gate|up GEMM runs as N / 16 workgroups of ONE strip (8 gate + 8 up rows)
    // instead of N / 32 strip pairs -- 768 instead of 384 for the talker: three per CU instead of 1.5, 9.5 vs 10.7 us streamed.
    // Bit-identical results (the same per-element accumulation), a second packed copy of the operator.  QTTS_SWIGLU8=0: strip pairs.
    bool swiglu8_env = QTTS_OPT_ON("QTTS_SWIGLU8");
    // The code predictor's attention + o-projection of passes >= 1 as ONE launch (attention.hip: cp_attn_o_kernel; bf16 engines, batch <= 8):
    // 2.60 vs 2.68 ms per frame on the MI355X in its fourth version (profiles/r04_cp_attn_o.md; the first three were slower than the two
    // launches).  QTTS_CP_ATTN_O=0 (read at engine creation): attn_cp + the decode GEMM.
    bool cp_attn_o_env = QTTS_OPT_ON("QTTS_CP_ATTN_O");
    int cp_attn_o_pause = [] { const char* e = QTTS_ENV("QTTS_CP_ATTN_O_PAUSE"); return e && atoi(e) >= 0 ? std::min(atoi(e), 200) : 16; }();   // (A/B: x 64 clocks)
    int cp_attn_o_step = [] { const char* e = QTTS_ENV("QTTS_CP_ATTN_O_STEP"); return e && atoi(e) >= 0 ? std::min(atoi(e), 200) : 4; }();
    DevBuf ao_part;                    // cp_attn_o: [8 kv heads][8 rows][H] granules {partial sum, tag}
    int64_t cp_attn_o_count = 0, cp_front_count = 0;
    // ... with the layer's own q|k|v GEMM in front of it in the same launch (layers >= 1).  QTTS_CP_FRONT=0: the decode GEMM, then cp_attn_o.
    bool cp_front_env = QTTS_OPT_ON("QTTS_CP_FRONT");
    DevBuf ao_qkv;                     // [8 rows][q|k|v width] granules {value, tag}
    // The code predictor's MLP of a layer as ONE launch (cp_mlp.hip; round 5).  QTTS_CP_MLP=0 (copied at engine creation): the two decode GEMMs.
    bool cp_mlp_env = QTTS_OPT_ON("QTTS_CP_MLP");
    // a consumer's wait before its first read of other workgroups' granules, x 64 clocks (A/B sweep profiles/r05_cp_mlp.md: a read that
    // leaves too early comes back stale and costs a round trip -- 8: 2.63, 16: 2.55, 24: 2.50, 32: 2.54 ms per frame with one value for both)
    int cp_mlp_pause_b = [] { const char* e = QTTS_ENV("QTTS_CP_MLP_PAUSE_B"); return e && atoi(e) >= 0 ? std::min(atoi(e), 200) : 24; }();
    int cp_mlp_pause_c = [] { const char* e = QTTS_ENV("QTTS_CP_MLP_PAUSE_C"); return e && atoi(e) >= 0 ? std::min(atoi(e), 200) : 24; }();
    int cp_mlp_step = [] { const char* e = QTTS_ENV("QTTS_CP_MLP_STEP"); return e && atoi(e) >= 0 ? std::min(atoi(e), 200) : 4; }();
    DevBuf mlp_act, mlp_part;          // granule buffers of the fused MLP launch
    // Round 6: the same launch at batch 9..32 (cp_mlp32.hip; BASELINE configs 4 / 5 run the frame step at batch 32): bf16 engines created for more
    // than 8 rows.  QTTS_CP_MLP32=0 (copied at engine creation): the two decode GEMMs.  Its launches count in cp_mlp_count / cp_mlp_per_step.
    bool cp_mlp32_env = QTTS_OPT_ON("QTTS_CP_MLP32");
    DevBuf mlp32_act, mlp32_part;      // its granule buffers (32 rows per XCD)
    int64_t cp_mlp_count = 0;
    int cp_mlp_per_step = 0;
    // Round 6: both fused launches of a layer as ONE (cp_layer.hip: the hidden rows between them travel as granules, the gate|up block is
    // requested at entry by LDS-DMA).  QTTS_CP_LAYER=0 (copied at engine creation): the two launches.  A launch of it counts in
    // cp_attn_o_count AND cp_mlp_count (both stages ran fused) and in cp_layer_count.
    bool cp_layer_env = QTTS_OPT_ON("QTTS_CP_LAYER");
    int cp_layer_pause_h = [] { const char* e = QTTS_ENV("QTTS_CP_LAYER_PAUSE_H"); return e && atoi(e) >= 0 ? std::min(atoi(e), 200) : 16; }();   // (A/B: x 64 clocks)
    int cp_layer_gu_when = QTTS_OPT_INT("QTTS_CP_LAYER_GU_WHEN", 2);
    int cp_layer_hid_mode = QTTS_OPT_INT("QTTS_CP_LAYER_HID_MODE", 1);
    int cp_layer_gu_pace = QTTS_OPT_INT("QTTS_CP_LAYER_GU_PACE", 0);
    static constexpr int CP_HID_SLOTS = 128;            // one region of the hidden-row granules per launch slot (position x layer)
    DevBuf cp_hid;                     // [8 rows][H / 2] granules {2 x bf16 hidden, tag} (fp32 engines: [8][H] {fp32, tag})
    int64_t cp_layer_count = 0;
    int cp_layer_per_step = 0;
    void build_layer(LayerW& L, const std::string& p, const StackDims& d, bool rows) {
        auto qkvw = cat3(PS(p + "self_attn.q_proj.weight", {d.qd, d.H}), PS(p + "self_attn.k_proj.weight", {d.kvd, d.H}),
                         PS(p + "self_attn.v_proj.weight", {d.kvd, d.H}));
        auto guw = interleave_gu(PS(p + "mlp.gate_proj.weight", {d.I, d.H}), PS(p + "mlp.up_proj.weight", {d.I, d.H}), d.I, d.H);
        auto& ow = PS(p + "self_attn.o_proj.weight", {d.H, d.qd});
        auto& dw = PS(p + "mlp.down_proj.weight", {d.H, d.I});
        upload_packed(L.qkv_p, qkvw, d.qd + 2 * d.kvd, d.H, &PS(p + "input_layernorm.weight", {d.H}));
        L.fs_o = choose_fs(d.H, d.qd); L.fs_d = choose_fs(d.H, d.I);
        upload_packed(L.o_p, ow, d.H, d.qd, nullptr, L.fs_o);
        upload_packed(L.gu_p, guw, 2 * d.I, d.H, &PS(p + "post_attention_layernorm.weight", {d.H}));
        // (only engines that can run at batch <= 8 at all pay for the second copy -- 1.46 GB at 1.7B dims; an engine created for waves
        // of 9..32 rows keeps the strip-pair kernel for a short last wave: ADVICE r3)
        if (bf16 && swiglu8_env && cfg.max_batch <= 8 && d.I % 8 == 0 && skinny_swiglu8_takes(d.H))
            upload_packed(L.gu_p8, interleave_gu8(PS(p + "mlp.gate_proj.weight", {d.I, d.H}), PS(p + "mlp.up_proj.weight", {d.I, d.H}), d.I, d.H),
                          2 * d.I, d.H, &PS(p + "post_attention_layernorm.weight", {d.H}));
        upload_packed(L.d_p, dw, d.H, d.I, nullptr, L.fs_d);
        // code predictor, bf16: passes >= 1 run attention + o-projection as ONE launch (attention.hip: cp_attn_o_kernel), whose waves own
        // 16-feature strips of the operator (a second packed copy: 4 MB per layer)
        if (bf16 && !rows && cp_attn_o_env && d.nh == 16 && d.nkv == 8 && d.hd == 128 && d.H % 128 == 0)
            upload_packed(L.o_p16, ow, d.H, d.qd, nullptr, 16);
        // ... and the MLP as ONE launch (cp_mlp.hip): gate|up packed by workgroup (XCD-major slices of the intermediate vector), down in 16-feature strips
        // (fp32 engines too, round 5: the exact parity mode runs through the same construction; their down operator is in 16-feature strips already)
        if (!rows && ((cp_mlp_env && cp_mlp_instantiated(d.H, d.I, bf16)) || (cp_mlp32_env && bf16 && cp_mlp32_instantiated(d.H, d.I)))) {
            std::vector<char> h(cp_mlp_gu_bytes(d.H, d.I, bf16));
            pack_cp_mlp_gu(PS(p + "mlp.gate_proj.weight", {d.I, d.H}).data(), PS(p + "mlp.up_proj.weight", {d.I, d.H}).data(),
                           PS(p + "post_attention_layernorm.weight", {d.H}).data(), d.H, d.I, bf16, h.data());
            L.gu_mlp.upload(h.data(), h.size());
            if (bf16) upload_packed(L.d_p16, dw, d.H, d.I, nullptr, 16);
        }
        if (rows) {
            upload_rows(L.qkv_r, qkvw);
            upload_rows(L.o_r, ow);
            upload_rows(L.gu_r, guw);
            upload_rows(L.d_r, dw);
        }
        upload_f(L.g1, PS(p + "input_layernorm.weight", {d.H}));
        upload_f(L.g2, PS(p + "post_attention_layernorm.weight", {d.H}));
        upload_f(L.qn, PS(p + "self_attn.q_norm.weight", {d.hd}));
        upload_f(L.kn, PS(p + "self_attn.k_norm.weight", {d.hd}));
    }
    void finalize();

    float* ssbuf() { return ss_rows.as<float>(); }     // row sums of squares for GEMMs that cannot stage x (M > 32 / fp32)

    void skinny(const SkinnyParams& p, hipStream_t st) {
        if (timing_now) {
            LaunchEv e{nullptr, nullptr, cur_stack, p.N, p.K, 0.0};
            QTTS_CHECK_HIP(hipEventCreate(&e.a)); QTTS_CHECK_HIP(hipEventCreate(&e.b));
            ev.push_back(e);
            skinny_set_launch_events(e.a, e.b);
            try { launch_skinny(p, bf16, st); } catch (...) { skinny_set_launch_events(nullptr, nullptr); throw; }
            skinny_set_launch_events(nullptr, nullptr);
        } else launch_skinny(p, bf16, st);
        ++skinny_count;
    }
    int64_t skinny_count = 0;
    static bool skinny_ablate_or_off() { const char* e = QTTS_ENV("QTTS_SKINNY8"); return e && e[0] == '0'; }   // (A/B switch of skinny.hip, through the option table)

    // x-side handling of a GEMM whose input is RMS-normalised: the bf16 kernel takes the row variances on the matrix pipe
    // (skinny.hip) from the producer's bf16 copy of x; the fp32 parity kernel gets the row sums of squares from one extra
    // tiny kernel.
    void norm_input(SkinnyParams& p, const StackDims& d, const void* x16v, hipStream_t st) {
        p.norm = 1; p.eps = d.eps;
        if (bf16) {
            if (x16v) { p.x = reinterpret_cast<const float*>(x16v); p.x_bf16 = 1; }
        } else {
            if (skinny_f32_inline_norm(p.M, p.K)) return;        // round 4: the batch <= 8 fp32 kernel sums x^2 from its own fragments
            if (!skinny_only) launch_row_ss(p.x, p.ldx, p.M, p.K, ssbuf(), ss.done, st);
            p.ss_in = ssbuf();
        }
    }
    // fp32 mode, batch <= 8 (round 4): the o- and down-projections split K over two workgroups per strip -- half 0 adds the residual to its
    // sums, half 1 writes raw sums, both into `sk_part` -- and the NEXT GEMM of the chain (gate|up, the next layer's q|k|v) forms its x as
    // half 0 + half 1 while it loads it and writes the combined rows back to the residual buffer (skinny.hip: KS / COMB; nobody else reads
    // the residual buffer during that launch).  `sk_pending` says that the halves of a down-projection still wait for the next q|k|v.  The
    // last layer of a stack does not split its down-projection: the stack's output is complete where the caller reads it.
    DevBuf sk_part;                    // [2][8][H_max] floats (fp32 engines only)
    bool sk_pending = false;
    // bf16 engines, batch 17..32 (round 6; BASELINE configs 4 and 5 run the frame step at batch 32): the o- and down-projections split K over the
    // workgroups of a 32-feature strip group and combine the partial sums inside the launch (skinny.hip: skinny2_ks_kernel) -- a workgroup then
    // pulls a quarter / an eighth of the x rows.  QTTS_SKINNY_KS=0 (copied at engine creation: captured graphs bake the choice in).  Launch slots
    // (tags): talker layer l: 2 l (o), 2 l + 1 (down); code predictor (position, layer): 2 L_talker + 2 (position x layers + layer) + {0, 1}.
    bool ks_split_env = QTTS_OPT_ON("QTTS_SKINNY_KS");
    int ks_pause = QTTS_OPT_INT("QTTS_SKINNY_KS_PAUSE", 8);
    DevBuf ks_part;                    // granule workspace of the split launches (one launch at a time uses it)
    int64_t ks_split_count = 0;
    int ks_split_per_step = 0;
    int ks_mink = QTTS_OPT_INT("QTTS_SKINNY_KS_MINK", 6144);
    void ks_arm(SkinnyParams& q, int slot) {
        if (!ks_split_env || !ks_part.p || !bf16 || q.M <= 16 || q.M > 32 || slot < 0 || slot >= 256 || skinny_only) return;
        if (q.K < ks_mink || !skinny_ksplit_takes(q.M, q.N, q.K, q.fs ? q.fs : 16)) return;
        q.ks_part = ks_part.as<float>(); q.ks_part_bytes = ks_part.bytes; q.ks_serial = ss.frame_serial; q.ks_slot = slot; q.ks_pause = ks_pause;
        q.ks_err = ss.n_generated + 5; q.ks_latch = ss.done;
        ++ks_split_count;
    }
    // one decoder layer on `M = n_new * B` rows of `xs` (in place)
    void decode_layer(const LayerW& L, const StackDims& d, float* xs, unsigned short* xs16, float* qkvb, float* attb,
                      float* actb, int M, int n_new, KvCache& kv, int layer, const int* len_dev, int len_static,
                      const int* npad, const float* inv_freq, int max_len, hipStream_t st, const float* rope_cs = nullptr, int rope_cs_n = 0,
                      bool last_layer = true) {
        // the MLP of this call as ONE launch (cp_mlp.hip): code predictor passes >= 1 at batch <= 8 of an engine that holds a place of its device's account
        const bool mlp_fusable = cp_mlp_env && cp_fused_slot && L.gu_mlp.p && mlp_act.p && !skinny_only && !len_dev && n_new == 1 && len_static >= 1 &&
                                 cp_mlp_takes(M, d.H, d.I) && layer < CP_FUSED_MAX_LAYERS && len_static * CP_FUSED_MAX_LAYERS + layer < 128 && (!bf16 || (xs16 && skinny_takes_bf16_x(M, d.H, true)));
        // (fp32: a layer whose MLP is fused keeps its o-projection whole -- the fused launch reads complete rows and writes complete rows)
        const bool splitk = !bf16 && !mlp_fusable && sk_part.p && M <= 8 && skinny_f32_splitk_takes(M, d.qd, d.H) && skinny_f32_splitk_takes(M, d.I, d.H);
        if (layer == 0 || !splitk) sk_pending = false;
        const size_t pstride = (size_t)8 * d.H;
        // xs16: bf16 copy of the hidden state kept in step with xs by every producer (bf16 mode, M <= 16), or null
        const bool h16 = xs16 && skinny_takes_bf16_x(M, d.H, bf16);
        SkinnyParams p{};
        p.done_flag = ss.done;
        p.x = xs; p.ldx = d.H; p.M = M; p.Wp = L.qkv_p.p; p.N = d.qd + 2 * d.kvd; p.K = d.H;
        p.out = qkvb; p.ldo = d.qd + 2 * d.kvd; p.act = ACT_NONE;
        AttnDecodeParams a{};
        a.qkv = qkvb; a.ld = d.qd + 2 * d.kvd; a.B = B; a.n_new = n_new; a.nh = d.nh; a.nkv = d.nkv; a.hd = d.hd;
        a.qw = L.qn.as<float>(); a.kw = L.kn.as<float>(); a.eps = d.eps; a.inv_freq = inv_freq; a.n_pad = npad;
        a.len_dev = len_dev; a.len_static = len_static; a.kv = kv; a.layer = layer; a.out = attb; a.ldo = d.qd;
        a.max_len = max_len; a.done_flag = ss.done;
        a.rope_cs = rope_cs; a.rope_cs_n = rope_cs_n;
        if (len_dev && attn_nsplit_active > 1) {     // talker, long sequences: split-KV over the LIVE length's bucket, not the capacity
            a.nsplit = attn_nsplit_active; a.part = attn_part.as<float>(); a.max_len = attn_span_active;
        }
        // bf16 mode: attention output and SwiGLU output travel as bf16 (as in the reference's bf16 path) and are
        // staged into the consuming GEMM by LDS-DMA
        const bool att16 = bf16 && skinny_takes_bf16_x(M, d.qd, true), act16 = bf16 && skinny_takes_bf16_x(M, d.I, true);
        a.out_bf16 = att16;
        // bf16 engines, code predictor passes >= 1 at batch <= 8: attention and o-projection in one launch (split over k by kv head, partial
        // sums handed over as tagged granules and added in kv-head order; profiles/r04_cp_attn_o.md) -- and, where the layer has a q|k|v
        // GEMM of its own (layers >= 1: layer 0's row comes from the table), that GEMM in front of them in the same launch
        // (fp32 engines, on request: the F32 instantiation on the fp32 decode GEMM's own packed operators, only in front of a fused MLP)
        const void* wo_fused = bf16 ? L.o_p16.p : (L.fs_o == 16 ? L.o_p.p : nullptr);
        const bool fuse_ao = cp_attn_o_env && wo_fused && ao_part.p && (bf16 ? att16 : mlp_fusable) && !skinny_only && cp_attn_o_takes(a, d.H) && layer < CP_FUSED_MAX_LAYERS &&
                             len_static * CP_FUSED_MAX_LAYERS + layer < 128;
        const bool front = fuse_ao && !skip_qkv && (h16 || !bf16) && cp_front_env && d.H == 1024 && a.ld == 4 * 8 * (d.H / 128) * 16;
        if (!skip_qkv && !front) {        // otherwise the previous pass's sampler gathered this row of qkvb from the table
            if (splitk && sk_pending) {    // the previous layer's down-projection left (residual + half 0, half 1): added on the way in
                p.xp = sk_part.as<float>(); p.xp_stride = pstride; p.x_out = xs;
                sk_pending = false;
            }
            norm_input(p, d, h16 ? xs16 : nullptr, st);
            skinny(p, st);
        }
        const bool fuse_layer = cp_layer_env && fuse_ao && mlp_fusable && cp_hid.p && cp_layer_takes(a, d.H, d.I);
        if (fuse_layer) {           // the whole layer as ONE launch (cp_layer.hip): both parameter blocks as for the two launches
            CpLayerParams cl{};
            CpAttnOParams& f = cl.ao;
            f.a = a; f.Wo = wo_fused; f.res = xs; f.out = xs; f.out16 = nullptr;
            f.part = ao_part.as<float>(); f.serial = ss.frame_serial; f.slot = len_static * CP_FUSED_MAX_LAYERS + layer; f.phase = 2;
            f.err = ss.n_generated + 5; f.done_latch = ss.done; f.H = d.H; f.first_pause = cp_attn_o_pause; f.poll_step = cp_attn_o_step;
            if (front) {
                f.Wqkv = L.qkv_p.p; f.x16 = bf16 ? xs16 : reinterpret_cast<const unsigned short*>(xs); f.ldx16 = d.H; f.K = d.H; f.eps_in = d.eps; f.qkv_gran = ao_qkv.as<float>();
                ++cp_front_count;
            }
            CpMlpParams& m = cl.mlp;
            m.f32 = bf16 ? 0 : 1;
            m.Wgu = L.gu_mlp.p; m.Wd = bf16 ? L.d_p16.p : L.d_p.p; m.eps = d.eps;
            m.out = xs; m.out16 = bf16 ? xs16 : nullptr;
            m.act_gran = mlp_act.as<float>(); m.part = mlp_part.as<float>(); m.serial = ss.frame_serial; m.slot = f.slot; m.phase = 3;
            m.err = f.err; m.done_latch = ss.done; m.done_flag = ss.done;
            m.B = M; m.H = d.H; m.I = d.I;
            m.first_pause = cp_mlp_pause_b; m.pause_c = cp_mlp_pause_c; m.poll_step = cp_mlp_step;
            cl.hid_gran = cp_hid.as<float>(); cl.hid_slot = f.slot; cl.hid_mode = cp_layer_hid_mode;
            cl.pause_h = cp_layer_pause_h; cl.gu_when = cp_layer_gu_when; cl.gu_pace = cp_layer_gu_pace; cl.phase = 8;
            if (timing_now) {          // bench.py's roofline leg: timed on its own (stack 5: the layer launch, every operator of the layer)
                const double eb = bf16 ? 2.0 : 4.0;
                LaunchEv e{nullptr, nullptr, 5, front ? a.ld + d.H + 3 * d.I : d.H + 3 * d.I, d.H,
                           eb * ((double)d.H * d.qd + (front ? (double)a.ld * d.H : 0.0) + 3.0 * (double)d.I * d.H)};
                QTTS_CHECK_HIP(hipEventCreate(&e.a)); QTTS_CHECK_HIP(hipEventCreate(&e.b));
                ev.push_back(e);
                cp_layer_set_launch_events(e.a, e.b);
                try { launch_cp_layer(cl, st); } catch (...) { cp_layer_set_launch_events(nullptr, nullptr); throw; }
                cp_layer_set_launch_events(nullptr, nullptr);
            } else launch_cp_layer(cl, st);
            ++cp_attn_o_count; ++cp_mlp_count; ++cp_layer_count;
            sk_pending = false;
            return;
        }
        if (fuse_ao) {
            CpAttnOParams f{};
            f.a = a; f.Wo = wo_fused; f.res = xs; f.out = xs; f.out16 = h16 ? xs16 : nullptr;
            f.part = ao_part.as<float>(); f.serial = ss.frame_serial; f.slot = len_static * CP_FUSED_MAX_LAYERS + layer; f.phase = 2;
            f.err = ss.n_generated + 5; f.done_latch = ss.done; f.H = d.H; f.first_pause = cp_attn_o_pause; f.poll_step = cp_attn_o_step;
            if (front) {
                f.Wqkv = L.qkv_p.p; f.x16 = bf16 ? xs16 : reinterpret_cast<const unsigned short*>(xs); f.ldx16 = d.H; f.K = d.H; f.eps_in = d.eps; f.qkv_gran = ao_qkv.as<float>();
                ++cp_front_count;
            }
            if (timing_now) {          // bench.py's roofline leg: this launch timed on its own, as the decode GEMM's (stack 3: the fused launch)
                const double eb = bf16 ? 2.0 : 4.0;
                LaunchEv e{nullptr, nullptr, 3, front ? a.ld + d.H : d.H, front ? d.H : d.qd,
                           eb * ((double)d.H * d.qd + (front ? (double)a.ld * d.H : 0.0))};
                QTTS_CHECK_HIP(hipEventCreate(&e.a)); QTTS_CHECK_HIP(hipEventCreate(&e.b));
                ev.push_back(e);
                cp_attn_o_set_launch_events(e.a, e.b);
                try { launch_cp_attn_o(f, st); } catch (...) { cp_attn_o_set_launch_events(nullptr, nullptr); throw; }
                cp_attn_o_set_launch_events(nullptr, nullptr);
            } else launch_cp_attn_o(f, st);
            ++cp_attn_o_count;
        } else {
        if (!skinny_only) launch_attn_decode(a, st);
        SkinnyParams o{};
        o.done_flag = ss.done;
        o.x_bf16 = att16;
        o.x = attb; o.ldx = d.qd; o.M = M; o.Wp = L.o_p.p; o.N = d.H; o.K = d.qd; o.res = xs; o.ldr = d.H;
        o.out = xs; o.ldo = d.H; o.act = ACT_NONE; o.out16 = h16 ? xs16 : nullptr; o.fs = L.fs_o;
        if (splitk) { o.out = sk_part.as<float>(); o.ksplit = 2; o.part_stride = pstride; }     // (o.res = xs: half 0 = residual + its sums)
        const int ks_slot0 = len_dev ? 2 * layer : 2 * (int)tl.size() + 2 * (len_static * (int)cl.size() + layer);
        if (n_new == 1) ks_arm(o, ks_slot0);
        skinny(o, st);
        }
        // bf16 engines, code predictor passes >= 1 at batch <= 8: the MLP as ONE launch (cp_mlp.hip) instead of the two decode GEMMs below
        // ... and at batch 9..32 the 32-row form of the same launch (cp_mlp32.hip)
        const bool mlp32_fusable = cp_mlp32_env && !mlp_fusable && bf16 && cp_fused_slot && L.gu_mlp.p && L.d_p16.p && mlp32_act.p && !skinny_only && !len_dev && n_new == 1 &&
                                   len_static >= 1 && M > 8 && cp_mlp32_takes(M, d.H, d.I) && layer < CP_FUSED_MAX_LAYERS && len_static * CP_FUSED_MAX_LAYERS + layer < 128 && h16;
        const bool fuse_mlp = mlp_fusable || mlp32_fusable;
        if (fuse_mlp) {
            CpMlpParams m{};
            m.f32 = bf16 ? 0 : 1;
            m.Wgu = L.gu_mlp.p; m.Wd = bf16 ? L.d_p16.p : L.d_p.p; m.x16 = bf16 ? xs16 : reinterpret_cast<const unsigned short*>(xs); m.ldx16 = d.H; m.eps = d.eps;
            m.res = xs; m.out = xs; m.out16 = bf16 ? xs16 : nullptr;
            m.act_gran = mlp_act.as<float>(); m.part = mlp_part.as<float>(); m.serial = ss.frame_serial; m.slot = len_static * CP_FUSED_MAX_LAYERS + layer; m.phase = 3;
            m.err = ss.n_generated + 5; m.done_latch = ss.done; m.done_flag = ss.done; m.first_pause = cp_attn_o_pause; m.poll_step = cp_attn_o_step;
            m.B = M; m.H = d.H; m.I = d.I;
            m.first_pause = cp_mlp_pause_b; m.pause_c = cp_mlp_pause_c; m.poll_step = cp_mlp_step;
            if (mlp32_fusable) {       // (its own granule buffers: 32 rows per XCD)
                m.act_gran = mlp32_act.as<float>(); m.part = mlp32_part.as<float>();
                launch_cp_mlp32(m, st);
                ++cp_mlp_count;
                sk_pending = false;
                return;
            }
            if (timing_now) {          // bench.py's roofline leg: timed on its own (stack 4: the fused MLP launch, three operators)
                LaunchEv e{nullptr, nullptr, 4, 3 * d.I, d.H, (bf16 ? 2.0 : 4.0) * 3.0 * (double)d.I * d.H};
                QTTS_CHECK_HIP(hipEventCreate(&e.a)); QTTS_CHECK_HIP(hipEventCreate(&e.b));
                ev.push_back(e);
                cp_mlp_set_launch_events(e.a, e.b);
                try { launch_cp_mlp(m, st); } catch (...) { cp_mlp_set_launch_events(nullptr, nullptr); throw; }
                cp_mlp_set_launch_events(nullptr, nullptr);
            } else launch_cp_mlp(m, st);
            ++cp_mlp_count;
            sk_pending = false;
            return;
        }
        SkinnyParams g{};
        g.done_flag = ss.done;
        g.x = xs; g.ldx = d.H; g.M = M; g.Wp = L.gu_p.p; g.N = 2 * d.I; g.K = d.H; g.out = actb; g.ldo = d.I; g.act = ACT_SWIGLU;
        g.out_bf16 = act16;
        if (splitk) { g.xp = sk_part.as<float>(); g.xp_stride = pstride; g.x_out = xs; }     // (residual + half 0) + half 1 of the o-projection
        norm_input(g, d, h16 ? xs16 : nullptr, st);
        if (L.gu_p8.p && g.x_bf16 && M <= 8 && !skinny_ablate_or_off()) { g.Wp = L.gu_p8.p; g.act = ACT_SWIGLU8; }
        skinny(g, st);
        SkinnyParams dn{};
        dn.done_flag = ss.done;
        dn.x_bf16 = act16;
        dn.x = actb; dn.ldx = d.I; dn.M = M; dn.Wp = L.d_p.p; dn.N = d.H; dn.K = d.I; dn.res = xs; dn.ldr = d.H;
        dn.out = xs; dn.ldo = d.H; dn.act = ACT_NONE; dn.out16 = h16 ? xs16 : nullptr; dn.fs = L.fs_d;
        if (splitk && !last_layer) { dn.out = sk_part.as<float>(); dn.ksplit = 2; dn.part_stride = pstride; sk_pending = true; }   // (dn.res = xs)
        if (n_new == 1) ks_arm(dn, (len_dev ? 2 * layer : 2 * (int)tl.size() + 2 * (len_static * (int)cl.size() + layer)) + 1);
        skinny(dn, st);
    }

    void prefill(const float* embeds, int B_, int T, const int32_t* n_pad_host, const float* trailing_dev, int Tt_,
                 const float* tts_pad_dev, hipStream_t st);
    void sample_talker(const qtts_sampling& sp, int eos, int min_new, int max_new, hipStream_t st);
    void frame_step(const qtts_sampling& sp, int eos, int min_new, int max_new, int64_t* codes, float* hidden,
                    int max_frames, hipStream_t st);
    // Two captured frame graphs per configuration: the short-sequence one (one attention workgroup per (sequence, kv head)) and,
    // for engines whose max_seq exceeds 512, the long-sequence one (split-KV attention + merge kernel, 28 more nodes): measured
    // on MI355X the split costs +0.29 ms per frame at 100-200 keys and saves 0.9 ms at 800 (profiles/r02_long_utterance_*),
    // so a generation switches graphs when its KV length passes SPLIT_FROM keys.
    // The key range is partitioned by the LIVE length (round 3): the host knows the KV length a burst of frame steps
    // will reach, picks the power-of-two bucket that holds it (512, 1024, 2048, ...) and launches the graph captured for that
    // bucket -- SPLIT_KEYS keys per workgroup, at most attn_nsplit workgroups per (sequence, kv head).  Partitioning the static
    // capacity instead (round 2) left an engine created with max_seq = 4096 / 9216 (attach() / from_pretrained defaults) with 1-2
    // non-empty splits at 800 keys: the merge nodes' cost without the split's gain.
    int SPLIT_FROM = [] { const char* e = QTTS_ENV("QTTS_ATTN_SPLIT_FROM"); return e && atoi(e) > 0 ? atoi(e) : 320; }();   // (env: tests)
    int SPLIT_KEYS = [] { const char* e = QTTS_ENV("QTTS_ATTN_SPLIT_KEYS"); return e && atoi(e) >= 64 ? atoi(e) / 64 * 64 : 256; }();   // (env: tests)
    std::map<int, std::pair<hipGraph_t, hipGraphExec_t>> graph_long;      // bucket (keys) -> captured long-sequence frame step
    int attn_nsplit_active = 1;        // what decode_layer launches (and what a capture in progress bakes in)
    int attn_span_active = 0;          // ... and the key span those workgroups partition (the bucket)
    bool long_mode(int kv_len_after) const { return attn_nsplit > 1 && kv_len_after > SPLIT_FROM; }
    // bucket of a KV length in long mode: the smallest power of two >= kv_len, at least 2 * SPLIT_KEYS, at most the capacity's bucket
    int span_bucket(int kv_len) const {
        int b = 2 * SPLIT_KEYS;
        while (b < kv_len) b *= 2;
        return b;
    }
    int nsplit_for(int bucket) const { return std::max(2, std::min(attn_nsplit, bucket / SPLIT_KEYS)); }
    void set_attn_mode(int kv_len_after) {
        if (long_mode(kv_len_after)) { attn_span_active = span_bucket(kv_len_after); attn_nsplit_active = nsplit_for(attn_span_active); }
        else { attn_span_active = 0; attn_nsplit_active = 1; }
    }
    bool any_graph() const { return graph_exec != nullptr || !graph_long.empty(); }
    // the captured graph for the mode a burst ending at `kv_len_after` keys runs in, capturing `step` (which is NOT executed by
    // the capture) on first use
    template <class F>
    hipGraphExec_t ensure_graph(int kv_len_after, hipStream_t st, F&& step) {
        set_attn_mode(kv_len_after);
        const bool lng = attn_nsplit_active > 1;
        std::pair<hipGraph_t, hipGraphExec_t> none{nullptr, nullptr};
        auto& slot = lng ? graph_long[attn_span_active] : none;
        hipGraph_t& g = lng ? slot.first : graph;
        hipGraphExec_t& ge = lng ? slot.second : graph_exec;
        if (!ge) {
            QTTS_CHECK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            try { step(); }
            catch (...) {
                hipGraph_t gx = nullptr;
                (void)hipStreamEndCapture(st, &gx);
                if (gx) (void)hipGraphDestroy(gx);
                throw;
            }
            QTTS_CHECK_HIP(hipStreamEndCapture(st, &g));
            size_t nn = 0;
            QTTS_CHECK_HIP(hipGraphGetNodes(g, nullptr, &nn));
            graph_nodes = (int)nn;
            QTTS_CHECK_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        }
        return ge;
    }
    void destroy_graph() {
        for (auto& kv : graph_long) {
            if (kv.second.second) (void)hipGraphExecDestroy(kv.second.second);
            if (kv.second.first) (void)hipGraphDestroy(kv.second.first);
        }
        graph_long.clear();
        if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
        if (graph) { (void)hipGraphDestroy(graph); graph = nullptr; }
    }
    ~qtts_talker() {
        destroy_graph(); release_events();
        fused_release();
    }
    // The fused launches (attention.hip: cp_attn_o_kernel; cp_mlp.hip: cp_mlp_kernel) are correct only when ALL workgroups of a launch are
    // resident at once: workgroups wait for granules that other workgroups of the same launch produce.  Residency is a property of the
    // DEVICE, so the admission is per device (round 5), and it is an account of the one resource that limits it here, the
    // register file: a compute unit has 512 registers per lane and SIMD; a launch of `grid` workgroups on `cus` compute units puts
    // ceil(grid / cus) workgroups on a compute unit, each -- 4 waves, one per SIMD, of <= 184 registers (bf16 engines; <= 272 for the fp32
    // instantiations of the two kernels: CP_SHARE_F32) -- with CP_SHARE = 184 of that budget
    // (the code objects' own numbers are pinned by tests/test_host_logic.py::test_fused_launches_fit_their_register_shares).  An engine's
    // share is that of its LARGEST fused launch (its launches run one after the other on one stream); engines are admitted while the shares
    // of the fused engines of a device add up to <= 512: two on a whole MI355X (bench.py --workload clone-shard at batch 8 runs two), none
    // on a CPX partition (32 compute units: 8 workgroups of a 256-workgroup launch per unit).  Kernels that do not wait for anybody (the
    // codec on another stream, another engine's GEMMs) only DELAY a fused launch: they drain, their places go to the launch's pending
    // workgroups (dispatch is in order), and the wait is bounded by their duration, far below the give-up limit.  What the rule cannot see
    // is another PROCESS on the same device: a consumer that loses its producers there gives up after ~0.3 s, latches the stop flag (one
    // give-up per generation, not per launch), the call fails with QTTS_ERR_STATE and the engine leaves the fused launches for good
    // (`fused_retire`): the caller's retry runs on the separate launches.
    // The launch tag's slot = position * CP_FUSED_MAX_LAYERS + layer must be unique among the launches of a frame that write one granule
    // buffer: a code predictor with more layers than this (the reference's default and every released checkpoint have 5;
    // configuration_qwen3_tts.py:370-454) keeps the separate launches -- finalize() switches both fused launches off (layer 5 of
    // pass L would otherwise share its tag with layer 0 of pass L + 1 and a consumer could take a stale granule as fresh).
    static constexpr int CP_FUSED_MAX_LAYERS = 5;
    static constexpr int CU_REG_BUDGET = 512, CP_SHARE = 184, CP_SHARE_F32 = 272;     // (fp32 engines: the F32 instantiations hold twice the operand registers -- one engine per device)
    // Round 6: the layer launch (cp_layer.hip) runs both stages in the two launches' register share + 8 (192 in bf16; 360 with the operators
    // in registers in fp32) and holds 65 KB of LDS per workgroup (the gate|up block requested by LDS-DMA): the account has a SECOND resource,
    // a compute unit's 160 KB of LDS (a register-only account over-admits the moment a fused launch stages operands in LDS).
    // cp_mlp32.hip (batch 9..32): 208 registers (184 + 24 accumulator), 33 KB of LDS -- two such engines per device inside 7/8 of the register file (the
    // headroom rule of the layer launch below: launches that fit EXACTLY starve their pending workgroups behind other streams' short waves)
    static constexpr int CP_SHARE_MLP32 = 208, CP_LDS_MLP32 = 34 * 1024;
    static constexpr int CP_SHARE_LAYER = 192, CP_SHARE_LAYER_F32 = 360, CU_LDS_BUDGET = 160 * 1024, CP_LDS_TWO_LAUNCHES = 17 * 1024;
    struct FusedDev { int regs = 0, lds = 0, engines = 0; };
    struct FusedRegistry { std::mutex m; std::map<int, FusedDev> dev; };        // device -> (register share, LDS bytes in use, fused engines)
    static FusedRegistry& fused_registry() { static FusedRegistry r; return r; }
    bool cp_fused_slot = false;
    int fused_device = -1, fused_capacity = 0, fused_share = 0, fused_lds = 0;     // capacity: engines with this engine's shares the device holds at once
    int cp_fused_per_step = 0;                         // fused launches in the frame step last launched / captured
    int cp_fused_giveups = 0;                          // generations of this engine that ended on the give-up flag
    // QTTS_CP_FUSED_MAX (A/B, tests): cap on fused engines per device below what residency allows
    // grid_cp: workgroups per launch of the engine's largest fused kernel; occ_ok: the occupancy API finds room for at least one workgroup of
    // every wanted kernel on a compute unit; share / lds: registers per lane and LDS bytes of one workgroup of the engine's largest fused launch
    // `budget_regs` / `budget_lds`: the part of a compute unit fused engines may take together.  The two-launch kernels (184 of 512 registers per
    // engine) leave room beside two engines for the short waves of other streams.  The first layer kernel (256 registers, 77 KB of LDS) fitted
    // twice EXACTLY -- and then any short wave that lands on a compute unit keeps a pending workgroup out until it has left, while the resident
    // workgroups of both launches wait for the pending ones: on the MI355X two such engines beside a codec stream and a third engine ran into
    // a give-up (profiles/r06_cp_layer.md).  Nor did the second version (192 registers, 65 KB): two layer engines alone ran clean, two beside a codec
    // stream and a third engine gave up again, a layer engine beside a two-launch engine and the same neighbours did not (tools/diag_layer_pair.py).
    // So layer engines may take HALF a compute unit's LDS together -- one per device -- and 7/8 of its registers; the next engine takes the two launches.
    void fused_admit(int grid_cp, bool occ_ok, int share, int lds, int budget_regs = CU_REG_BUDGET, int budget_lds = CU_LDS_BUDGET) {
        QTTS_CHECK_HIP(hipGetDevice(&fused_device));
        int cus = 0;
        QTTS_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, fused_device));
        const int per_cu = grid_cp > 0 ? cdiv(grid_cp, std::max(1, cus)) : 0;
        const int need = share * per_cu, need_lds = lds * per_cu;
        int max_engines = 1 << 20;
        if (const char* e = QTTS_ENV("QTTS_CP_FUSED_MAX")) max_engines = std::max(0, atoi(e));
        auto& r = fused_registry();
        std::lock_guard<std::mutex> lk(r.m);
        auto& d = r.dev[fused_device];
        fused_capacity = need > 0 && occ_ok ? std::min(max_engines, std::min(budget_regs / need, need_lds > 0 ? budget_lds / need_lds : 1 << 20)) : 0;
        if (!occ_ok || need <= 0 || d.engines + 1 > max_engines || d.regs + need > budget_regs || d.lds + need_lds > budget_lds) return;
        fused_share = need; fused_lds = need_lds; cp_fused_slot = true;
        d.regs += fused_share; d.lds += fused_lds; d.engines += 1;
    }
    void fused_release() {
        if (!cp_fused_slot) return;
        auto& r = fused_registry();
        std::lock_guard<std::mutex> lk(r.m);
        auto& d = r.dev[fused_device];
        d.regs -= fused_share; d.lds -= fused_lds; d.engines -= 1;
        cp_fused_slot = false; fused_share = 0; fused_lds = 0;
    }
    // after a give-up: this engine runs the separate launches from now on (graphs that baked the fused launch in are dropped)
    void fused_retire() {
        ++cp_fused_giveups;
        cp_attn_o_env = false; cp_mlp_env = false; cp_mlp32_env = false; cp_layer_env = false; ks_split_env = false;
        fused_release();
        destroy_graph();
        graph_nodes = 0;
    }
    void check_fused_flag(int flag, const char* where) {
        if (!flag) return;
        fused_retire();
        throw Error(QTTS_ERR_STATE, std::string(where) + ": a consumer of a fused launch gave up waiting for its producers' granules (the fused launch was not "
                                    "fully resident: another process on the device?); this engine now uses the separate launches -- retry the request");
    }
};

void qtts_talker::finalize() {
    const auto& c = cfg;
    td = {c.hidden_size, c.intermediate_size, c.num_attention_heads, c.num_key_value_heads, c.head_dim,
          c.num_attention_heads * c.head_dim, c.num_key_value_heads * c.head_dim, c.rms_norm_eps};
    cd = {c.cp_hidden_size, c.cp_intermediate_size, c.cp_num_attention_heads, c.cp_num_key_value_heads, c.cp_head_dim,
          c.cp_num_attention_heads * c.cp_head_dim, c.cp_num_key_value_heads * c.cp_head_dim, c.cp_rms_norm_eps};
    QTTS_REQUIRE(td.hd == 128 && cd.hd == 128, QTTS_ERR_ARG, "talker/code-predictor head_dim must be 128");
    QTTS_REQUIRE(td.nh / td.nkv <= 2 && cd.nh / cd.nkv <= 2, QTTS_ERR_ARG, "GQA group size must be <= 2");
    QTTS_REQUIRE(c.max_batch >= 1 && c.max_batch <= 32, QTTS_ERR_LIMIT, "max_batch must be 1..32");
    QTTS_REQUIRE(td.I % 16 == 0 && cd.I % 16 == 0, QTTS_ERR_ARG, "intermediate sizes % 16");
    const int G = c.num_code_groups;
    if (c.cp_num_hidden_layers > CP_FUSED_MAX_LAYERS) { cp_mlp_env = false; cp_attn_o_env = false; }     // (tag uniqueness, see CP_FUSED_MAX_LAYERS)
    if (!cp_mlp_instantiated(cd.H, cd.I, bf16) || c.max_batch > 8) cp_mlp_env = false;
    // the 32-row form: bf16 engines created for more than 8 rows, at most CP_FUSED_MAX_LAYERS layers (tag uniqueness)
    if (!bf16 || c.max_batch <= 8 || c.cp_num_hidden_layers > CP_FUSED_MAX_LAYERS || !cp_mlp32_instantiated(cd.H, cd.I)) cp_mlp32_env = false;
    // fp32 engines (the exact parity mode) take the fused MLP launch only on request (QTTS_CP_MLP_F32=1): cp_mlp_kernel<true, ...> is the fused
    // construction's bit-exact leg (tests/test_gpu_parity.py runs the reference goldens through it), but with fp32 operators the MLP is
    // 38 MB per layer and the round-4 plan -- o- and down-projection split over two workgroups per strip -- streams it faster than 256
    // workgroups of 256 registers do: 4.18 vs 4.30 ms per frame (1.7B, batch 8; profiles/r05_cp_mlp.md)
    if (!bf16 && !QTTS_OPT_SET("QTTS_CP_MLP_F32")) cp_mlp_env = false;
    // ... and the fused attention + o-projection launch likewise (QTTS_CP_ATTN_O_F32=1), together with the fused MLP only: both read and
    // write complete rows, so no half of a split-K projection is pending anywhere in passes >= 1
    const bool ao_f32 = !bf16 && cp_mlp_env && QTTS_OPT_SET("QTTS_CP_ATTN_O_F32");
    const bool want_ao = (bf16 || ao_f32) && cp_attn_o_env && cd.nh == 16 && cd.nkv == 8 && cd.H % 128 == 0;
    // the layer launch (cp_layer.hip) needs both fused stages, its instantiation, and a contiguous page table (the engine's always is)
    bool want_layer = cp_layer_env && want_ao && cp_mlp_env && cp_layer_instantiated(cd.H, cd.I, bf16);
    if (want_ao || cp_mlp_env || cp_mlp32_env) {      // one admission for the engine's fused launches
        int grid_cp = 0;
        bool occ_ok = true;
        if (want_ao) { grid_cp = std::max(grid_cp, cp_attn_o_grid(cd.H)); occ_ok = occ_ok && cp_attn_o_blocks_per_cu(!bf16) >= 1; }
        if (cp_mlp_env) { grid_cp = std::max(grid_cp, cp_mlp_grid(cd.H)); occ_ok = occ_ok && cp_mlp_blocks_per_cu(cd.H, cd.I, bf16) >= 1; }
        if (cp_mlp32_env) { grid_cp = std::max(grid_cp, cp_mlp_grid(cd.H)); occ_ok = occ_ok && cp_mlp32_blocks_per_cu(cd.H, cd.I) >= 1; }
        if (want_layer && cp_layer_blocks_per_cu(cd.H, cd.I, bf16) < 1) want_layer = false;      // (also sets the kernels' dynamic-LDS attribute, outside any capture)
        if (want_layer) fused_admit(grid_cp, occ_ok, bf16 ? CP_SHARE_LAYER : CP_SHARE_LAYER_F32, cp_layer_lds_bytes(cd.H, cd.I, bf16),
                                    CU_REG_BUDGET - CU_REG_BUDGET / 8, CU_LDS_BUDGET / 2);
        if (!cp_fused_slot) {         // no room for (or no) layer launch: the two launches' smaller shares
            want_layer = false;
            if (cp_mlp32_env) fused_admit(grid_cp, occ_ok, CP_SHARE_MLP32, CP_LDS_MLP32, CU_REG_BUDGET - CU_REG_BUDGET / 8);
            else fused_admit(grid_cp, occ_ok, bf16 ? CP_SHARE : CP_SHARE_F32, CP_LDS_TWO_LAUNCHES);
        }
    }
    if (!want_ao || !cp_fused_slot) cp_attn_o_env = false;
    if (!cp_fused_slot) { cp_mlp_env = false; cp_mlp32_env = false; }
    cp_layer_env = want_layer && cp_fused_slot;
    tl.resize(c.num_hidden_layers);
    for (int l = 0; l < c.num_hidden_layers; ++l) build_layer(tl[l], "model.layers." + std::to_string(l) + ".", td, true);
    cl.resize(c.cp_num_hidden_layers);
    for (int l = 0; l < c.cp_num_hidden_layers; ++l)
        build_layer(cl[l], "code_predictor.model.layers." + std::to_string(l) + ".", cd, false);
    upload_f(t_norm, PS("model.norm.weight", {td.H}));
    upload_f(c_norm, PS("code_predictor.model.norm.weight", {cd.H}));
    fs_head = choose_fs(c.vocab_size, td.H); fs_lm = choose_fs(c.cp_vocab_size, cd.H); fs_proj = choose_fs(cd.H, td.H);
    upload_packed(head_p, PS("codec_head.weight", {c.vocab_size, td.H}), c.vocab_size, td.H, nullptr, fs_head);
    upload_f(emb_talker, PS("model.codec_embedding.weight", {c.vocab_size, td.H}));
    {
        std::vector<float> e((size_t)(G - 1) * c.cp_vocab_size * td.H);
        for (int g = 0; g < G - 1; ++g) {
            auto& w = PS("code_predictor.model.codec_embedding." + std::to_string(g) + ".weight", {c.cp_vocab_size, td.H});
            memcpy(&e[(size_t)g * c.cp_vocab_size * td.H], w.data(), w.size() * 4);
        }
        upload_f(emb_cp, e);
    }
    lm_head_p.resize(G - 1);
    for (int g = 0; g < G - 1; ++g)
        upload_packed(lm_head_p[g], PS("code_predictor.lm_head." + std::to_string(g) + ".weight", {c.cp_vocab_size, cd.H}), c.cp_vocab_size, cd.H,
                      &PS("code_predictor.model.norm.weight", {cd.H}), fs_lm);
    has_proj = cd.H != td.H;
    if (has_proj) {
        upload_packed(proj_p, PS("code_predictor.small_to_mtp_projection.weight", {cd.H, td.H}), cd.H, td.H, nullptr, fs_proj);
        upload_f(proj_b, PS("code_predictor.small_to_mtp_projection.bias", {cd.H}));
    }
    // A/B variant (build.py VARIANTS): passes 1 .. G-2 of the code predictor feed small_to_mtp_projection with
    // codec_embedding[j-1](token) (M:1281-1282) -- a function of the token alone.  Tabulate it once, with the SAME decode
    // GEMM launches the frame step would make (16 rows at a time; a row's result does not depend on the batch around it), so
    // the sampler's fused gather can fetch the projected row and 14 of the 15 projection GEMMs leave the frame graph.
    if (has_proj && G > 2) {
        const int nt = G - 2;
        emb_cp_proj.alloc((size_t)nt * c.cp_vocab_size * cd.H * 4);
        for (int g = 0; g < nt; ++g)
            for (int v0 = 0; v0 < c.cp_vocab_size; v0 += 16) {
                SkinnyParams pj{};
                pj.x = emb_cp.as<float>() + ((size_t)g * c.cp_vocab_size + v0) * td.H; pj.ldx = td.H;
                pj.M = std::min(16, c.cp_vocab_size - v0); pj.Wp = proj_p.p; pj.N = cd.H; pj.K = td.H;
                pj.bias = proj_b.as<float>(); pj.out = emb_cp_proj.as<float>() + ((size_t)g * c.cp_vocab_size + v0) * cd.H;
                pj.ldo = cd.H; pj.act = ACT_NONE; pj.fs = fs_proj;
                launch_skinny(pj, bf16, nullptr);
            }
        QTTS_CHECK_HIP(hipDeviceSynchronize());
    }
    has_text_proj = host.count("text_projection.linear_fc1.weight") > 0;
    if (has_text_proj) {
        const int TH = c.text_hidden_size;
        upload_rows(tp_fc1, PS("text_projection.linear_fc1.weight", {TH, TH})); upload_f(tp_b1, PS("text_projection.linear_fc1.bias", {TH}));
        upload_rows(tp_fc2, PS("text_projection.linear_fc2.weight", {td.H, TH})); upload_f(tp_b2, PS("text_projection.linear_fc2.bias", {td.H}));
    }
    has_text_emb = host.count("model.text_embedding.weight") > 0;
    if (has_text_emb) {
        const auto& shp = shapes["model.text_embedding.weight"];
        QTTS_REQUIRE(shp.size() == 2 && shp[1] == c.text_hidden_size, QTTS_ERR_ARG,
                     "model.text_embedding.weight must be [text_vocab][text_hidden_size]");
        text_vocab = shp[0];
        upload_rows(emb_text, host["model.text_embedding.weight"]);
    }
    err_flag.alloc(4);
    QTTS_CHECK_HIP(hipMemset(err_flag.p, 0, 4));
    auto mk_freq = [&](DevBuf& d, const char* name, float theta, int hd) {
        if (host.count(name)) { upload_f(d, PS(name, {hd / 2})); return; }
        std::vector<float> f(hd / 2);
        for (int i = 0; i < hd / 2; ++i) f[i] = 1.0f / powf(theta, (float)(2 * i) / (float)hd);
        upload_f(d, f);
    };
    mk_freq(inv_freq_t, "model.rotary_emb.inv_freq", c.rope_theta, td.hd);
    mk_freq(inv_freq_c, "code_predictor.model.rotary_emb.inv_freq", c.cp_rope_theta, cd.hd);
    if (cd.hd == 128) {                 // (attn_cp's head_dim; positions 0 .. G)
        rope_cs_c.alloc((size_t)(G + 1) * 128 * sizeof(float));
        launch_rope_table(inv_freq_c.as<float>(), G + 1, rope_cs_c.as<float>(), nullptr);
        QTTS_CHECK_HIP(hipDeviceSynchronize());
    }

    // bytes of packed weights one frame step streams (bench.py's roofline numerator)
    const double eb = bf16 ? 2.0 : 4.0;
    auto layer_b = [&](const StackDims& d) { return eb * ((double)(d.qd + 2 * d.kvd) * d.H + (double)d.H * d.qd + 3.0 * d.I * d.H); };
    weight_bytes_frame = c.num_hidden_layers * layer_b(td) + eb * (double)c.vocab_size * td.H +
                         (G - 1) * (c.cp_num_hidden_layers * layer_b(cd) + eb * (double)c.cp_vocab_size * cd.H +
                                    (has_proj ? eb * (double)cd.H * td.H : 0.0));
    if (has_proj && G > 2) weight_bytes_frame -= (G - 2) * eb * (double)cd.H * td.H;   // only pass 0 still runs the projection
    if (G > 2) weight_bytes_frame -= (G - 2) * eb * (double)(cd.qd + 2 * cd.kvd) * cd.H;   // layer-0 qkv GEMM of passes >= 1

    // ---- KV caches (pages of 16 tokens, reserved up front)
    const size_t esz = bf16 ? 2 : 4;
    const int pps = cdiv(c.max_seq, 16);
    // bf16 talker cache: V pages transposed ([dim][16 keys]) -- the A-operand image of the PV product of attn_tk16_kernel, which runs
    // both attention products on the matrix pipe (QTTS_ATTN_MFMA=0: the VALU kernel attn_tk on row-major V pages, for A/B runs)
    const bool attn_mfma = bf16 && QTTS_OPT_ON("QTTS_ATTN_MFMA");
    kv_t = {nullptr, nullptr, nullptr, pps, pps * c.max_batch, td.nkv, td.hd, bf16 ? 1 : 0, 1, attn_mfma ? 1 : 0};
    const size_t tb = (size_t)c.num_hidden_layers * kv_t.n_pages * td.nkv * 16 * td.hd * esz;
    kpool_t.alloc(tb); vpool_t.alloc(tb);
    QTTS_CHECK_HIP(hipMemset(kpool_t.p, 0, tb)); QTTS_CHECK_HIP(hipMemset(vpool_t.p, 0, tb));
    const int cpps = cdiv(G + 1, 16);
    kv_c = {nullptr, nullptr, nullptr, cpps, cpps * c.max_batch, cd.nkv, cd.hd, bf16 ? 1 : 0, 1, 0};
    const size_t cb = (size_t)c.cp_num_hidden_layers * kv_c.n_pages * cd.nkv * 16 * cd.hd * esz;
    kpool_c.alloc(cb); vpool_c.alloc(cb);
    QTTS_CHECK_HIP(hipMemset(kpool_c.p, 0, cb)); QTTS_CHECK_HIP(hipMemset(vpool_c.p, 0, cb));
    {
        std::vector<int> t((size_t)c.max_batch * pps), u((size_t)c.max_batch * cpps);
        for (size_t i = 0; i < t.size(); ++i) t[i] = (int)i;
        for (size_t i = 0; i < u.size(); ++i) u[i] = (int)i;
        ptab_t.upload(t.data(), t.size() * 4); ptab_c.upload(u.data(), u.size() * 4);
    }
    // a sequence that can grow past 512 keys is read by several workgroups per (sequence, kv head): one CU pulls ~25 GB/s, and a
    // 60 s utterance has 0.4 MB of K / V per head and layer (measured 52 us per layer at 800 keys with one workgroup)
    if (const char* e = QTTS_ENV("QTTS_ATTN_NSPLIT")) attn_nsplit = std::max(1, std::min(16, atoi(e)));
    else attn_nsplit = c.max_seq > 512 ? std::min(16, cdiv(c.max_seq, SPLIT_KEYS)) : 1;
    if (attn_nsplit > 1) attn_part.alloc(attn_part_floats(c.max_batch, td.nkv, attn_nsplit, td.nh / td.nkv) * sizeof(float));
    kv_t.k = kpool_t.p; kv_t.v = vpool_t.p; kv_t.page_table = ptab_t.as<int>();
    kv_c.k = kpool_c.p; kv_c.v = vpool_c.p; kv_c.page_table = ptab_c.as<int>();

    // ---- decode scratch (rows <= 64)
    const int R = 64;
    x16.alloc((size_t)R * td.H * 2); cp_x16.alloc((size_t)R * cd.H * 2); ph16.alloc((size_t)R * td.H * 2); cp_in16.alloc((size_t)R * td.H * 2);
    x.alloc((size_t)R * td.H * 4); qkv.alloc((size_t)R * (td.qd + 2 * td.kvd) * 4); att.alloc((size_t)R * td.qd * 4);
    act.alloc((size_t)R * td.I * 4); logits.alloc((size_t)R * c.vocab_size * 4); past_hidden.alloc((size_t)R * td.H * 4);
    cp_in.alloc((size_t)R * td.H * 4); cp_x.alloc((size_t)R * cd.H * 4); cp_qkv.alloc((size_t)R * (cd.qd + 2 * cd.kvd) * 4);
    cp_att.alloc((size_t)R * cd.qd * 4); cp_act.alloc((size_t)R * cd.I * 4); cp_logits.alloc((size_t)(c.num_code_groups - 1) * R * c.cp_vocab_size * 4);     // [pass][row][cp vocab]: every pass keeps its own rows (qtts_talker_debug_cp_logits)
    cur_tok.alloc(R * 4); sub.alloc((size_t)R * G * 4); ss_rows.alloc(64 * 8); ints.alloc(64 * 4 + R * 4);
    if (!bf16) {                      // the two halves of a split-K o- / down-projection (decode_layer)
        const size_t hmax = (size_t)std::max(td.H, cd.H);
        sk_part.alloc(2 * 8 * hmax * 4);
        QTTS_CHECK_HIP(hipMemset(sk_part.p, 0, sk_part.bytes));
    }
    if (bf16 && c.max_batch > 16 && ks_split_env) {      // 256 workgroups x 8 (strip, m-tile) pairs x 64 lanes x 4 granules
        ks_part.alloc((size_t)8 << 20);
        QTTS_CHECK_HIP(hipMemset(ks_part.p, 0, ks_part.bytes));
    }
    if (!cl.empty() && cl[0].gu_mlp.p && cp_mlp32_env) {
        mlp32_act.alloc(cp_mlp32_act_bytes(cd.I)); mlp32_part.alloc(cp_mlp32_part_bytes(cd.H));
        QTTS_CHECK_HIP(hipMemset(mlp32_act.p, 0, mlp32_act.bytes));
        QTTS_CHECK_HIP(hipMemset(mlp32_part.p, 0, mlp32_part.bytes));
    }
    if (!cl.empty() && cl[0].gu_mlp.p) {
        mlp_act.alloc((size_t)8 * 8 * (cd.I / (bf16 ? 16 : 8)) * 8);
        mlp_part.alloc((size_t)8 * 8 * cd.H * 8);
        QTTS_CHECK_HIP(hipMemset(mlp_act.p, 0, mlp_act.bytes));
        QTTS_CHECK_HIP(hipMemset(mlp_part.p, 0, mlp_part.bytes));
    }
    if (!cl.empty() && cp_attn_o_env && (bf16 ? cl[0].o_p16.p != nullptr : true)) {
        ao_part.alloc((size_t)8 * 8 * cd.H * 8);
        ao_qkv.alloc((size_t)8 * (cd.qd + 2 * cd.kvd) * 8);
        QTTS_CHECK_HIP(hipMemset(ao_qkv.p, 0, ao_qkv.bytes));
        QTTS_CHECK_HIP(hipMemset(ao_part.p, 0, ao_part.bytes));
    }
    if (cp_layer_env && mlp_act.p && ao_part.p) {
        cp_hid.alloc((size_t)CP_HID_SLOTS * 8 * (bf16 ? cd.H / 2 : cd.H) * 8);
        QTTS_CHECK_HIP(hipMemset(cp_hid.p, 0, cp_hid.bytes));
    }
    n_pad_d.alloc(R * 4); suppress.alloc(c.vocab_size); seed_d.alloc(8);
    QTTS_CHECK_HIP(hipMemset(ss_rows.p, 0, ss_rows.bytes));
    int* ip = ints.as<int>();
    ss = {ip + 0, ip + 1, ip + 2, ip + 3, ip + 4, ip + 64, ip + 6};
    { const int one = 1; QTTS_CHECK_HIP(hipMemcpy(ss.frame_serial, &one, 4, hipMemcpyHostToDevice)); }       // (0 is the tag of a never-written granule)
    // A/B variant (build.py VARIANTS): in passes 1 .. G-2 the code predictor's layer-0 q|k|v GEMM sees only the pass input row,
    // a function of the previous token alone (has_proj: the projected embedding above; otherwise codec_embedding itself).
    // Tabulate it with the same launches the frame step makes -- bf16 mode: LDS-staged from the bf16 image of the row with the
    // variance taken from that image, as at run time for up to 32 rows; fp32 mode: row sums of squares from row_ss_kernel.
    if (G > 2) {
        const int nt = G - 2, QW = cd.qd + 2 * cd.kvd, Vc = c.cp_vocab_size;
        const float* xt = has_proj ? emb_cp_proj.as<float>() : emb_cp.as<float>();
        const bool staged = skinny_takes_bf16_x(16, cd.H, bf16);
        DevBuf x16t;
        if (staged) {
            std::vector<float> hx((size_t)nt * Vc * cd.H);
            QTTS_CHECK_HIP(hipMemcpy(hx.data(), xt, hx.size() * 4, hipMemcpyDeviceToHost));
            std::vector<bf16_t> h16(hx.size());
            parallel_for((int64_t)hx.size(), [&](int64_t a, int64_t b) { for (int64_t i = a; i < b; ++i) h16[i] = f32_to_bf16(hx[i]); });
            x16t.upload(h16.data(), h16.size() * 2);
        }
        cp_qkv0_tab.alloc((size_t)nt * Vc * QW * 4);
        for (int g = 0; g < nt; ++g)
            for (int v0 = 0; v0 < Vc; v0 += 16) {
                const size_t row0 = (size_t)g * Vc + v0;
                SkinnyParams p{};
                p.x = xt + row0 * cd.H; p.ldx = cd.H; p.M = std::min(16, Vc - v0); p.Wp = cl[0].qkv_p.p; p.N = QW; p.K = cd.H;
                p.out = cp_qkv0_tab.as<float>() + row0 * QW; p.ldo = QW; p.act = ACT_NONE; p.norm = 1; p.eps = cd.eps;
                if (staged) { p.x = reinterpret_cast<const float*>(x16t.as<bf16_t>() + row0 * cd.H); p.x_bf16 = 1; }
                else { launch_row_ss(p.x, cd.H, p.M, cd.H, ssbuf(), nullptr, nullptr); p.ss_in = ssbuf(); }
                launch_skinny(p, bf16, nullptr);
            }
        QTTS_CHECK_HIP(hipDeviceSynchronize());
    }
    host.clear();
    finalized = true;
}

// ------------------------------------------------------------------------------------------ prefill
void qtts_talker::prefill(const float* embeds, int B_, int T, const int32_t* n_pad_host, const float* trailing_dev,
                          int Tt_, const float* tts_pad_dev, hipStream_t st) {
    const auto& c = cfg;
    QTTS_REQUIRE(finalized, QTTS_ERR_STATE, "talker: finalize() first");
    QTTS_REQUIRE(B_ >= 1 && B_ <= c.max_batch, QTTS_ERR_LIMIT, "talker: batch exceeds max_batch");
    QTTS_REQUIRE(T >= 1 && T < c.max_seq, QTTS_ERR_LIMIT, "talker: prompt longer than max_seq");
    QTTS_REQUIRE(Tt_ >= 1, QTTS_ERR_ARG, "talker: trailing_text_hidden must have >= 1 row");
    B = B_; T0 = T; Tt = Tt_;
    for (int b = 0; b < B; ++b) QTTS_REQUIRE(n_pad_host[b] >= 0 && n_pad_host[b] < T, QTTS_ERR_ARG, "talker: n_pad out of range");
    const int M = B * T, H = td.H, W = td.qd + 2 * td.kvd;
    pf_x.ensure((size_t)M * H * 4); pf_n.ensure((size_t)M * std::max(H, td.qd) * 4); pf_qkv.ensure((size_t)M * W * 4);
    pf_att.ensure((size_t)M * td.qd * 4); pf_act.ensure((size_t)M * td.I * 4);
    trailing.ensure((size_t)B * Tt * H * 4); tts_pad.ensure((size_t)H * 4);
    QTTS_CHECK_HIP(hipMemcpyAsync(pf_x.p, embeds, (size_t)M * H * 4, hipMemcpyDeviceToDevice, st));
    QTTS_CHECK_HIP(hipMemcpyAsync(trailing.p, trailing_dev, (size_t)B * Tt * H * 4, hipMemcpyDeviceToDevice, st));
    QTTS_CHECK_HIP(hipMemcpyAsync(tts_pad.p, tts_pad_dev, (size_t)H * 4, hipMemcpyDeviceToDevice, st));
    QTTS_CHECK_HIP(hipMemcpyAsync(n_pad_d.p, n_pad_host, (size_t)B * 4, hipMemcpyHostToDevice, st));
    float *xs = pf_x.as<float>(), *nb = pf_n.as<float>(), *qb = pf_qkv.as<float>(), *ab = pf_att.as<float>(), *mb = pf_act.as<float>();
    // bf16 mode (round 3): a tensor whose only consumer is a GEMM -- the normed rows, the attention output, the SwiGLU product -- is
    // written as bf16 by its producer (the GEMM rounded it the same way while staging: bit-identical results) and read at half the
    // bytes; these small-grid GEMMs are bound by what a CU can pull (gemm_tap.hip, gemm_wide_kernel).  The bf16 tensors live in the
    // fp32 buffers' storage.  QTTS_PREFILL_A16=0: fp32 hand-over (A/B runs and the equality test).
    const bool a16_env = QTTS_OPT_ON("QTTS_PREFILL_A16");   // (read per call: the test runs both)
    // (the SwiGLU GEMM with bf16 input exists in the wide-K kernel only: K = H a multiple of 128, at least 512)
    const bool a16 = bf16 && a16_env && H >= 512 && H % 128 == 0 && td.qd % 8 == 0 && td.I % 32 == 0;
    auto gemm = [&](const DevBuf& Wr, int N, int K, const float* A, int lda, float* C, int ldc, int act_, const float* res, bool out16 = false) {
        GemmTapParams p{};
        if (a16) p.A16 = A; else p.A = A;
        p.lda = lda; p.M = M; p.T = T; p.W = Wr.p; p.N = N; p.K = K; p.taps = 1; p.act = act_;
        p.res = res; p.ldr = H;
        if (out16) { p.C16 = C; p.ldc16 = ldc; p.ldc = ldc; } else { p.C = C; p.ldc = ldc; }
        launch_gemm_tap(p, bf16, st);
    };
    for (int l = 0; l < c.num_hidden_layers; ++l) {
        auto& L = tl[l];
        if (a16) launch_rmsnorm16(xs, H, L.g1.as<float>(), td.eps, nb, H, M, H, st);
        else launch_rmsnorm(xs, H, L.g1.as<float>(), td.eps, nb, H, M, H, st);
        gemm(L.qkv_r, W, H, nb, H, qb, W, ACT_NONE, nullptr);
        QkNormRopeParams q{};
        q.qkv = qb; q.ld = W; q.B = B; q.T = T; q.nh = td.nh; q.nkv = td.nkv; q.hd = td.hd; q.qw = L.qn.as<float>();
        q.kw = L.kn.as<float>(); q.eps = td.eps; q.inv_freq = inv_freq_t.as<float>(); q.n_pad = n_pad_d.as<int>();
        q.kv = kv_t; q.layer = l;
        launch_qknorm_rope_store(q, st);
        AttnRowsParams a{};
        a.qkv = qb; a.ld = W; a.q_off = 0; a.k_off = td.qd; a.v_off = td.qd + td.kvd; a.B = B; a.T = T; a.nh = td.nh;
        a.nkv = td.nkv; a.hd = td.hd; a.window = 0; a.n_pad = n_pad_d.as<int>(); a.out = ab; a.ldo = td.qd;
        if (a16) a.out16 = ab;
        launch_attn_rows(a, st);
        gemm(L.o_r, H, td.qd, ab, td.qd, xs, H, ACT_NONE, xs);
        if (a16) launch_rmsnorm16(xs, H, L.g2.as<float>(), td.eps, nb, H, M, H, st);
        else launch_rmsnorm(xs, H, L.g2.as<float>(), td.eps, nb, H, M, H, st);
        gemm(L.gu_r, 2 * td.I, H, nb, H, mb, td.I, ACT_SWIGLU, nullptr, a16);
        gemm(L.d_r, H, td.I, mb, td.I, xs, H, ACT_NONE, xs);
    }
    // last position of every (left-padded) row -> final norm -> past_hidden, logits (M:1726-1740)
    for (int b = 0; b < B; ++b)
        QTTS_CHECK_HIP(hipMemcpyAsync(x.as<float>() + (size_t)b * H, xs + ((size_t)b * T + T - 1) * H, (size_t)H * 4,
                                      hipMemcpyDeviceToDevice, st));
    launch_rmsnorm(x.as<float>(), H, t_norm.as<float>(), td.eps, past_hidden.as<float>(), H, B, H, st);
    // loop state: the first sample+finish turns these into n_generated = 1, gen_step = 0, kv_len = T
    int init[6] = {0, -1, T - 1, 0, 0, 0};          // (slot 5: cp_attn_o's give-up flag, checked when the generation ends)
    QTTS_CHECK_HIP(hipMemcpyAsync(ss.n_generated, init, sizeof(init), hipMemcpyHostToDevice, st));
    std::vector<int> ones(B, 1);
    QTTS_CHECK_HIP(hipMemcpyAsync(ss.unfinished, ones.data(), (size_t)B * 4, hipMemcpyHostToDevice, st));
    SkinnyParams h{};
    h.x = past_hidden.as<float>(); h.ldx = H; h.M = B; h.Wp = head_p.p; h.N = c.vocab_size; h.K = H;
    h.out = logits.as<float>(); h.ldo = c.vocab_size; h.act = ACT_NONE; h.fs = fs_head;
    launch_skinny(h, bf16, st);
    QTTS_CHECK_HIP(hipStreamSynchronize(st));  // host buffers (n_pad, init, ones) must outlive the copies
    prefilled = true;
}

// ------------------------------------------------------------------------------------------ sampling of cb-0
void qtts_talker::sample_talker(const qtts_sampling& sp, int eos, int min_new, int max_new, hipStream_t st) {
    SampleParams p{};
    p.logits = logits.as<float>(); p.ld = cfg.vocab_size; p.V = cfg.vocab_size; p.B = B;
    p.generated = generated.as<int>(); p.gen_stride = gen_cap; p.n_generated_dev = ss.n_generated;
    p.repetition_penalty = sp.repetition_penalty; p.eos = eos; p.min_new_tokens = min_new;
    p.suppress_mask = suppress.as<unsigned char>();
    p.do_sample = sp.do_sample; p.top_k = sp.top_k; p.top_p = sp.top_p; p.temperature = sp.temperature;
    p.seed = sp.seed; p.seed_dev = seed_d.as<unsigned long long>(); p.stream_id = 0; p.step_dev = ss.n_generated;
    p.tok_out = cur_tok.as<int>(); p.tok_stride = 1; p.unfinished = ss.unfinished; p.generated_out = generated.as<int>();
    p.max_new_tokens = max_new; p.done_in = ss.done;
    if (tf.codes && tf.trace) launch_teacher(teacher_params(), 0, st);
    launch_sample(p, st);
    launch_sample_finish(ss, B, max_new, st);
    if (tf.codes) launch_teacher(teacher_params(), 1, st);
}

// ------------------------------------------------------------------------------------------ one frame
void qtts_talker::frame_step(const qtts_sampling& sp, int eos, int min_new, int max_new, int64_t* codes, float* hidden,
                             int max_frames, hipStream_t st) {
    const auto& c = cfg;
    const int G = c.num_code_groups;
    const int64_t fused_before = cp_attn_o_count, mlp_before = cp_mlp_count, layer_before = cp_layer_count, ks_before = ks_split_count;
    // ---- code predictor: G-1 dependent passes (M:1671-1680, 1250-1312)
    cur_stack = 1;
    for (int j = 0; j < G - 1; ++j) {
        const int n_new = j == 0 ? 2 : 1, M = n_new * B;
        CpGatherParams gp{};
        gp.pass = j; gp.B = B; gp.H = td.H; gp.past_hidden = past_hidden.as<float>(); gp.talker_emb = emb_talker.as<float>();
        gp.cur_tok = cur_tok.as<int>(); gp.cp_emb = emb_cp.as<float>(); gp.cp_vocab = c.cp_vocab_size;
        gp.sub = sub.as<int>(); gp.sub_stride = G; gp.done = ss.done;
        unsigned short* c16 = bf16 ? cp_x16.as<unsigned short>() : nullptr;
        // passes j >= 1 get their input row from the previous pass's sampler (fused gather): only pass 0 gathers here
        if (has_proj && j >= 1) {
            // the previous pass's sampler gathered the projected row straight into cp_x (and its bf16 shadow)
        } else
        if (has_proj) {
            gp.out = cp_in.as<float>(); gp.out16 = bf16 ? cp_in16.as<unsigned short>() : nullptr;
            if (!skinny_only && j == 0) launch_cp_gather(gp, st);
            SkinnyParams pj{};
            pj.done_flag = ss.done;
            pj.x = cp_in.as<float>(); pj.ldx = td.H; pj.M = M; pj.Wp = proj_p.p; pj.N = cd.H; pj.K = td.H;
            if (bf16) { pj.x = reinterpret_cast<const float*>(cp_in16.as<unsigned short>()); pj.x_bf16 = 1; }
            pj.bias = proj_b.as<float>(); pj.out = cp_x.as<float>(); pj.ldo = cd.H; pj.act = ACT_NONE; pj.fs = fs_proj;
            pj.out16 = (c16 && skinny_takes_bf16_x(M, cd.H, bf16)) ? c16 : nullptr;
            skinny(pj, st);
        } else {
            gp.out = cp_x.as<float>(); gp.out16 = c16;
            if (!skinny_only && j == 0) launch_cp_gather(gp, st);
        }
        for (int l = 0; l < c.cp_num_hidden_layers; ++l) {
            skip_qkv = j >= 1 && l == 0;
            decode_layer(cl[l], cd, cp_x.as<float>(), c16, cp_qkv.as<float>(), cp_att.as<float>(), cp_act.as<float>(), M, n_new,
                         kv_c, l, nullptr, j == 0 ? 0 : j + 1, nullptr, inv_freq_c.as<float>(), 32, st, rope_cs_c.as<float>(),
                         rope_cs_c.p ? G + 1 : 0, l + 1 == c.cp_num_hidden_layers);
        }
        skip_qkv = false;
        // final norm folded into lm_head[j]; only the LAST token's rows are needed (pass 0: rows [B, 2B))
        SkinnyParams lh{};
        lh.done_flag = ss.done;
        const int off = (n_new - 1) * B;
        lh.x = cp_x.as<float>() + (size_t)off * cd.H; lh.ldx = cd.H; lh.M = B; lh.Wp = lm_head_p[j].p; lh.N = c.cp_vocab_size;
        lh.K = cd.H; lh.out = cp_logits.as<float>() + (size_t)j * 64 * c.cp_vocab_size; lh.ldo = c.cp_vocab_size; lh.act = ACT_NONE; lh.fs = fs_lm;
        norm_input(lh, cd, (c16 && skinny_takes_bf16_x(M, cd.H, bf16)) ? c16 + (size_t)off * cd.H : nullptr, st);
        skinny(lh, st);
        SampleParams s{};
        s.logits = cp_logits.as<float>() + (size_t)j * 64 * c.cp_vocab_size; s.ld = c.cp_vocab_size; s.V = c.cp_vocab_size; s.B = B;
        s.repetition_penalty = 1.0f; s.eos = -1; s.do_sample = sp.subtalker_dosample; s.top_k = sp.subtalker_top_k;
        s.top_p = sp.subtalker_top_p; s.temperature = sp.subtalker_temperature; s.seed = sp.seed; s.stream_id = 1 + j;
        s.seed_dev = seed_d.as<unsigned long long>();
        s.step_dev = ss.n_generated; s.tok_out = sub.as<int>() + j; s.tok_stride = G; s.done_in = ss.done;
        if (j + 1 < G - 1) {       // next pass's input = codec_embedding[j](this token) (M:1281)
            s.gather_emb = emb_cp.as<float>() + (size_t)j * c.cp_vocab_size * td.H; s.gather_C = td.H;
            s.gather_out = has_proj ? cp_in.as<float>() : cp_x.as<float>();
            s.gather_out16 = has_proj ? nullptr : c16;
            if (has_proj) {
                s.gather_emb = emb_cp_proj.as<float>() + (size_t)j * c.cp_vocab_size * cd.H; s.gather_C = cd.H;
                s.gather_out = cp_x.as<float>();
                s.gather_out16 = (c16 && skinny_takes_bf16_x(B, cd.H, bf16)) ? c16 : nullptr;   // as the projection's out16
            }
            s.gather2_C = cd.qd + 2 * cd.kvd;
            s.gather2_emb = cp_qkv0_tab.as<float>() + (size_t)j * c.cp_vocab_size * s.gather2_C;
            s.gather2_out = cp_qkv.as<float>();
        }
        if (!skinny_only) launch_sample(s, st);
    }
    if (tf.codes && !skinny_only) launch_teacher(teacher_params(), 2, st);
    // ---- next talker input + frame outputs (M:1681-1692)
    EmbedSumParams e{};
    e.B = B; e.H = td.H; e.G = G; e.cp_vocab = c.cp_vocab_size; e.talker_emb = emb_talker.as<float>();
    e.cp_emb = emb_cp.as<float>(); e.cur_tok = cur_tok.as<int>(); e.sub = sub.as<int>(); e.sub_stride = G;
    e.trailing = trailing.as<float>(); e.Tt = Tt; e.tts_pad = tts_pad.as<float>(); e.past_hidden = past_hidden.as<float>();
    e.x_out = x.as<float>(); e.x_out16 = bf16 ? x16.as<unsigned short>() : nullptr; e.codes_out = codes; e.hidden_out = hidden; e.max_frames = max_frames; e.st = ss;
    if (!skinny_only) launch_embed_sum(e, st);
    // ---- talker decode forward (M:1706-1727)
    cur_stack = 0;
    for (int l = 0; l < c.num_hidden_layers; ++l)
        decode_layer(tl[l], td, x.as<float>(), bf16 ? x16.as<unsigned short>() : nullptr, qkv.as<float>(), att.as<float>(), act.as<float>(), B, 1, kv_t, l, ss.kv_len, 0,
                     n_pad_d.as<int>(), inv_freq_t.as<float>(), c.max_seq, st, nullptr, 0, l + 1 == c.num_hidden_layers);
    if (!skinny_only) launch_apply_norm(x.as<float>(), td.H, t_norm.as<float>(), td.eps, past_hidden.as<float>(), td.H, B, td.H, ss.done, st,
                                        bf16 ? ph16.as<unsigned short>() : nullptr);
    SkinnyParams h{};
    h.done_flag = ss.done;
    h.x = past_hidden.as<float>(); h.ldx = td.H; h.M = B; h.Wp = head_p.p; h.N = c.vocab_size; h.K = td.H;
    if (bf16) { h.x = reinterpret_cast<const float*>(ph16.as<unsigned short>()); h.x_bf16 = 1; }
    h.out = logits.as<float>(); h.ldo = c.vocab_size; h.act = ACT_NONE; h.fs = fs_head;
    cur_stack = 2;
    skinny(h, st);
    if (!skinny_only) sample_talker(sp, eos, min_new, max_new, st);
    cp_fused_per_step = (int)(cp_attn_o_count - fused_before);      // (a captured step replays exactly these launches)
    cp_mlp_per_step = (int)(cp_mlp_count - mlp_before);
    cp_layer_per_step = (int)(cp_layer_count - layer_before);
    ks_split_per_step = (int)(ks_split_count - ks_before);
}

// A blocking copy of the engine's run-time paths goes through the ENGINE'S stream, never the legacy stream: a `hipMemcpy` orders the legacy stream
// behind every blocking stream of the process, and while ANOTHER engine's thread captures its frame step (thread-local capture mode) HIP refuses it
// ("operation would make the legacy stream depend on a capturing blocking stream": bench.py --workload clone-shard with two engines per GPU hit it
// in 2 of 4 runs on the MI355X in round 6).
static void copy_on_stream(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t st) {
    QTTS_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, kind, st));
    QTTS_CHECK_HIP(hipStreamSynchronize(st));
}

// ============================================================================================ C ABI
#define QTTS_API_BEGIN try {
#define QTTS_API_END                                                        \
    }                                                                       \
    catch (const qtts::Error& e) { qtts::set_last_error(e.what()); return e.code; } \
    catch (const std::exception& e) { qtts::set_last_error(e.what()); return QTTS_ERR_ARG; } \
    return QTTS_OK;

extern "C" {

int qtts_talker_create(const qtts_talker_config* cfg, qtts_talker** out) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(cfg && out, QTTS_ERR_ARG, "null argument");
    QTTS_REQUIRE(cfg->weight_dtype == QTTS_F32 || cfg->weight_dtype == QTTS_BF16, QTTS_ERR_ARG, "weight_dtype");
    QTTS_REQUIRE(cfg->num_code_groups >= 2 && cfg->num_code_groups <= 32, QTTS_ERR_ARG, "num_code_groups");
    QTTS_REQUIRE(cfg->hidden_size % 128 == 0 && cfg->cp_hidden_size % 128 == 0 && cfg->intermediate_size % 128 == 0 &&
                     cfg->cp_intermediate_size % 128 == 0, QTTS_ERR_ARG, "hidden/intermediate sizes must be multiples of 128");
    QTTS_REQUIRE(cfg->vocab_size % 16 == 0 && cfg->cp_vocab_size % 16 == 0, QTTS_ERR_ARG, "vocab sizes % 16");
    int ndev = 0;
    if (!getenv("QTTS_DEBUG_NO_DEVICE")) {
        QTTS_CHECK_HIP(hipGetDeviceCount(&ndev));
        QTTS_REQUIRE(ndev > 0, QTTS_ERR_HIP, "no HIP device");
    }
    auto* t = new qtts_talker();
    t->cfg = *cfg;
    t->bf16 = cfg->weight_dtype == QTTS_BF16;
    *out = t;
    QTTS_API_END
}
void qtts_talker_destroy(qtts_talker* t) { delete t; }

int qtts_talker_bind(qtts_talker* t, const char* name, const void* host, int32_t src_dtype, int32_t ndim,
                     const int64_t* shape) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(t && name && host && shape, QTTS_ERR_ARG, "null argument");
    QTTS_REQUIRE(!t->finalized, QTTS_ERR_STATE, "bind after finalize");
    QTTS_REQUIRE(src_dtype == QTTS_F32 || src_dtype == QTTS_BF16, QTTS_ERR_ARG, "src_dtype");
    HostTensor ht{host, src_dtype, std::vector<int64_t>(shape, shape + ndim)};
    t->host[name] = ht.to_f32();
    t->shapes[name] = ht.shape;
    QTTS_API_END
}
int qtts_talker_finalize(qtts_talker* t) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(t, QTTS_ERR_ARG, "null handle");
    t->finalize();
    QTTS_API_END
}

static void text_project(qtts_talker* t, const float* x_dev, int rows, float* y_dev, hipStream_t st) {
    const int TH = t->cfg.text_hidden_size;
    QTTS_REQUIRE(TH % 32 == 0, QTTS_ERR_ARG, "text_hidden_size % 32");
    t->tp_tmp.ensure((size_t)rows * TH * 4);
    // fc1 + bias, SiLU, fc2 + bias (M:815-816)
    GemmTapParams p{};
    p.A = x_dev; p.lda = TH; p.M = rows; p.T = rows; p.W = t->tp_fc1.p; p.N = TH; p.K = TH; p.taps = 1;
    p.bias = t->tp_b1.as<float>(); p.act = ACT_SILU; p.C = t->tp_tmp.as<float>(); p.ldc = TH;
    launch_gemm_tap(p, t->bf16, st);
    GemmTapParams q{};
    q.A = t->tp_tmp.as<float>(); q.lda = TH; q.M = rows; q.T = rows; q.W = t->tp_fc2.p; q.N = t->td.H; q.K = TH; q.taps = 1;
    q.bias = t->tp_b2.as<float>(); q.act = ACT_NONE; q.C = y_dev; q.ldc = t->td.H;
    launch_gemm_tap(q, t->bf16, st);
}
static void check_err_flag(qtts_talker* t, hipStream_t st, const char* what) {
    int e = 0;
    QTTS_CHECK_HIP(hipMemcpyAsync(&e, t->err_flag.p, 4, hipMemcpyDeviceToHost, st));
    QTTS_CHECK_HIP(hipStreamSynchronize(st));
    if (e) {
        QTTS_CHECK_HIP(hipMemset(t->err_flag.p, 0, 4));
        throw Error(QTTS_ERR_ARG, what);
    }
}

int qtts_talker_text_projection(qtts_talker* t, const float* x_dev, int32_t rows, float* y_dev, void* stream) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(t && x_dev && y_dev && rows >= 1, QTTS_ERR_ARG, "bad argument");
    QTTS_REQUIRE(t->finalized && t->has_text_proj, QTTS_ERR_STATE, "text_projection weights were not bound");
    text_project(t, x_dev, rows, y_dev, (hipStream_t)stream);
    QTTS_API_END
}

int qtts_talker_text_embed(qtts_talker* t, const int64_t* ids_dev, int32_t rows, float* y_dev, void* stream) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(t && ids_dev && y_dev && rows >= 1, QTTS_ERR_ARG, "bad argument");
    QTTS_REQUIRE(t->finalized && t->has_text_proj && t->has_text_emb, QTTS_ERR_STATE,
                 "text_embed: model.text_embedding / text_projection weights were not bound");
    hipStream_t st = (hipStream_t)stream;
    const int TH = t->cfg.text_hidden_size;
    t->tp_in.ensure((size_t)rows * TH * 4);
    launch_gather_rows(t->emb_text.p, t->bf16, t->text_vocab, TH, ids_dev, rows, t->tp_in.as<float>(), t->err_flag.as<int>(), st);
    text_project(t, t->tp_in.as<float>(), rows, y_dev, st);
    check_err_flag(t, st, "text_embed: text token id out of range");
    QTTS_API_END
}

int qtts_talker_assemble_rows(qtts_talker* t, const int32_t* desc_dev, int32_t rows, const float* proj_dev, int32_t proj_rows,
                              const float* spk_dev, int32_t n_spk, const int64_t* ref_codes_dev, int32_t n_ref_frames,
                              float* out_dev, void* stream) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(t && desc_dev && out_dev && rows >= 1, QTTS_ERR_ARG, "bad argument");
    QTTS_REQUIRE(t->finalized, QTTS_ERR_STATE, "assemble_rows before finalize");
    QTTS_REQUIRE(proj_rows >= 0 && n_spk >= 0 && n_ref_frames >= 0 && (proj_rows == 0 || proj_dev) && (n_spk == 0 || spk_dev) &&
                     (n_ref_frames == 0 || ref_codes_dev), QTTS_ERR_ARG, "assemble_rows: table pointer / count mismatch");
    hipStream_t st = (hipStream_t)stream;
    AssembleParams p{};
    p.desc = desc_dev; p.rows = rows; p.H = t->td.H; p.G = t->cfg.num_code_groups; p.cp_vocab = t->cfg.cp_vocab_size;
    p.vocab = t->cfg.vocab_size; p.proj = proj_dev; p.proj_rows = proj_rows; p.talker_emb = t->emb_talker.as<float>();
    p.cp_emb = t->emb_cp.as<float>(); p.spk = spk_dev; p.n_spk = n_spk; p.ref_codes = ref_codes_dev; p.n_ref = n_ref_frames;
    p.out = out_dev; p.err = t->err_flag.as<int>();
    launch_assemble_rows(p, st);
    check_err_flag(t, st, "assemble_rows: descriptor index or reference code out of range");
    QTTS_API_END
}

int qtts_talker_prefill(qtts_talker* t, const float* embeds_dev, int32_t B, int32_t T, const int32_t* n_pad_host,
                        const float* trailing_dev, int32_t Tt, const float* tts_pad_dev, void* stream) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(t && embeds_dev && n_pad_host && trailing_dev && tts_pad_dev, QTTS_ERR_ARG, "null argument");
    t->prefill(embeds_dev, B, T, n_pad_host, trailing_dev, Tt, tts_pad_dev, (hipStream_t)stream);
    QTTS_API_END
}

int qtts_talker_generate(qtts_talker* t, const qtts_sampling* sp, int32_t max_new_tokens, int32_t min_new_tokens,
                         int32_t eos_token_id, const int32_t* suppress_host, int32_t n_suppress, int64_t* codes_dev,
                         float* hidden_dev, int64_t* tokens_dev, int32_t* n_frames_host, void* stream) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(t && sp && codes_dev && n_frames_host, QTTS_ERR_ARG, "null argument");
    QTTS_REQUIRE(t->prefilled, QTTS_ERR_STATE, "generate: prefill() first");
    QTTS_REQUIRE(max_new_tokens >= 1, QTTS_ERR_ARG, "max_new_tokens >= 1");
    QTTS_REQUIRE(t->T0 + max_new_tokens <= t->cfg.max_seq, QTTS_ERR_LIMIT, "prompt + max_new_tokens exceeds max_seq");
    QTTS_REQUIRE(eos_token_id >= 0 && eos_token_id < t->cfg.vocab_size, QTTS_ERR_ARG, "eos_token_id");
    hipStream_t st = (hipStream_t)stream;
    const int B = t->B, V = t->cfg.vocab_size;
    if (t->tf.codes)
        QTTS_REQUIRE(min_new_tokens >= max_new_tokens && t->tf.F == max_new_tokens - 1 && !sp->do_sample && !sp->subtalker_dosample,
                     QTTS_ERR_ARG, "teacher forcing: greedy, min_new_tokens == max_new_tokens == forced frames + 1");
    t->prefilled = false;  // the KV cache / loop state are consumed by this call
    {
        std::vector<unsigned char> m(V, 0);
        for (int i = 0; i < n_suppress; ++i) {
            QTTS_REQUIRE(suppress_host[i] >= 0 && suppress_host[i] < V, QTTS_ERR_ARG, "suppress token out of range");
            m[suppress_host[i]] = 1;
        }
        QTTS_CHECK_HIP(hipMemcpyAsync(t->suppress.p, m.data(), V, hipMemcpyHostToDevice, st));
        const unsigned long long seed = sp->seed;
        QTTS_CHECK_HIP(hipMemcpyAsync(t->seed_d.p, &seed, 8, hipMemcpyHostToDevice, st));
        QTTS_CHECK_HIP(hipStreamSynchronize(st));
    }
    t->gen_cap = max_new_tokens;
    t->generated.ensure((size_t)B * max_new_tokens * 4);
    const int max_frames = std::max(1, max_new_tokens - 1);
    t->frames_run = 0;
    if (!t->profile) { t->prof_ms = 0; t->prof_launches = 0; t->prof_classes.clear(); }
    t->release_events();

    t->sample_talker(*sp, eos_token_id, min_new_tokens, max_new_tokens, st);      // token 0
    int done = 0;
    auto poll = [&]() {
        QTTS_CHECK_HIP(hipMemcpyAsync(&done, t->ss.done, 4, hipMemcpyDeviceToHost, st));
        QTTS_CHECK_HIP(hipStreamSynchronize(st));
    };
    poll();
    const bool use_graph = t->cfg.use_graph && !t->profile && !t->tf.codes;     // (teacher forcing runs eagerly)
    const int total = max_new_tokens - 1;      // at most this many frame steps
    int f = 0;
    qtts_talker::GraphKey key;
    memset(&key, 0, sizeof(key));          // (padding bytes take part in the comparison)
    key.B = B; key.Tt = t->Tt; key.eos = eos_token_id; key.min_new = min_new_tokens; key.max_new = max_new_tokens;
    key.max_frames = max_frames; key.do_sample = sp->do_sample; key.top_k = sp->top_k; key.sub_do_sample = sp->subtalker_dosample;
    key.sub_top_k = sp->subtalker_top_k; key.top_p = sp->top_p; key.temperature = sp->temperature; key.rep = sp->repetition_penalty;
    key.sub_top_p = sp->subtalker_top_p; key.sub_temperature = sp->subtalker_temperature; key.codes = codes_dev;
    key.hidden = hidden_dev; key.trailing = t->trailing.p; key.tts_pad = t->tts_pad.p; key.generated = t->generated.p;
    if (!use_graph || !t->any_graph() || !(key == t->graph_key)) { t->destroy_graph(); t->graph_nodes = 0; }
    while (!done && f < total) {
        if (t->profile == 2 && f == 1) {
            // roofline leg: ONLY the dominant kernel (every skinny GEMM of one frame step, same shapes/order) as a
            // hipGraph, replayed back-to-back between two HIP events on this stream.
            const int REPS = 20;
            t->set_attn_mode(t->T0 + f + 1);
            t->skinny_only = true;
            hipGraph_t g2 = nullptr; hipGraphExec_t ge2 = nullptr;
            QTTS_CHECK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            t->skinny_count = 0;
            try {
                t->frame_step(*sp, eos_token_id, min_new_tokens, max_new_tokens, codes_dev, hidden_dev, max_frames, st);
            } catch (...) { t->skinny_only = false; hipGraph_t gx = nullptr; (void)hipStreamEndCapture(st, &gx); if (gx) (void)hipGraphDestroy(gx); throw; }
            t->skinny_only = false;
            QTTS_CHECK_HIP(hipStreamEndCapture(st, &g2));
            QTTS_CHECK_HIP(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
            QTTS_CHECK_HIP(hipGraphLaunch(ge2, st));
            QTTS_CHECK_HIP(hipStreamSynchronize(st));
            hipEvent_t ea, eb;
            QTTS_CHECK_HIP(hipEventCreate(&ea)); QTTS_CHECK_HIP(hipEventCreate(&eb));
            QTTS_CHECK_HIP(hipEventRecord(ea, st));
            for (int r = 0; r < REPS; ++r) QTTS_CHECK_HIP(hipGraphLaunch(ge2, st));
            QTTS_CHECK_HIP(hipEventRecord(eb, st));
            QTTS_CHECK_HIP(hipStreamSynchronize(st));
            float ms = 0; QTTS_CHECK_HIP(hipEventElapsedTime(&ms, ea, eb));
            t->prof_ms = ms; t->prof_launches = t->skinny_count * REPS; t->graph_nodes = (int)t->skinny_count;
            (void)hipEventDestroy(ea); (void)hipEventDestroy(eb); (void)hipGraphExecDestroy(ge2); (void)hipGraphDestroy(g2);
            // the loop state is no longer meaningful: latch `done` with the one complete frame and stop
            int fin2[2] = {1, 2};
            copy_on_stream(t->ss.done, fin2, sizeof(fin2), hipMemcpyHostToDevice, st);
            done = 1;
            break;
        }
        if (!use_graph) {
            t->set_attn_mode(t->T0 + f + 1);
            t->timing_now = t->profile == 1 && f >= 1 && f <= qtts_talker::PROF_FRAMES;     // (frame 0 warms the code up)
            try { t->frame_step(*sp, eos_token_id, min_new_tokens, max_new_tokens, codes_dev, hidden_dev, max_frames, st); }
            catch (...) { t->timing_now = false; throw; }
            t->timing_now = false;
            ++f;
            if (f % 8 == 0) poll();
            continue;
        }
        const int burst = std::min(8, total - f);
        hipGraphExec_t ge = t->ensure_graph(t->T0 + f + burst, st, [&] {
            t->frame_step(*sp, eos_token_id, min_new_tokens, max_new_tokens, codes_dev, hidden_dev, max_frames, st); });
        t->graph_key = key;
        for (int i = 0; i < burst; ++i) QTTS_CHECK_HIP(hipGraphLaunch(ge, st));
        f += burst;
        // The stop condition cannot latch while EOS is still blocked by MinNewTokensLength (frame step i samples token i + 1; EOS is
        // -inf until min_new_tokens tokens exist) and max_new_tokens is not reached: such bursts need no host round trip -- the
        // queue stays fed (each poll idles the GPU for the host's wake-up + the next launch, ~50 us).
        if (f + 1 >= min_new_tokens || f >= total) poll();
    }
    QTTS_CHECK_HIP(hipStreamSynchronize(st));
    t->frames_run = f;
    if (t->profile == 1) t->aggregate_profile();
    int fin[6];
    copy_on_stream(fin, t->ss.n_generated, sizeof(fin), hipMemcpyDeviceToHost, st);
    t->check_fused_flag(fin[5], "generate");
    QTTS_REQUIRE(fin[3] == 1, QTTS_ERR_STATE, "generate: loop ended without the stop condition being latched");
    *n_frames_host = fin[4] - 1;
    if (tokens_dev) {   // int32 history -> int64 (B, max_new_tokens)
        std::vector<int> h((size_t)B * max_new_tokens);
        copy_on_stream(h.data(), t->generated.p, h.size() * 4, hipMemcpyDeviceToHost, st);
        std::vector<int64_t> w(h.size(), -1);
        for (int b = 0; b < B; ++b)
            for (int i = 0; i < fin[4]; ++i) w[(size_t)b * max_new_tokens + i] = h[(size_t)b * max_new_tokens + i];
        copy_on_stream(tokens_dev, w.data(), w.size() * 8, hipMemcpyHostToDevice, st);
    }
    QTTS_API_END
}

// ------------------------------------------------------------------------------------------ resumable generation
// qtts_talker_generate in three pieces, so that a caller can take the frames as they are produced (streaming output,
// BASELINE config 4) instead of waiting for the whole utterance: begin = token 0, step = up to n more frame steps
// (same captured frame graph, same device-resident loop state, stops at the latch), end = the tail bookkeeping.
// After k frame steps codes[:, :k] (and hidden[:, :k]) are final: a frame step writes its own frame first.
static void stream_launch_frames(qtts_talker* t, int n, hipStream_t st) {
    auto& g = t->sg;
    const bool use_graph = t->cfg.use_graph != 0;
    auto poll = [&]() {
        QTTS_CHECK_HIP(hipMemcpyAsync(&g.done, t->ss.done, 4, hipMemcpyDeviceToHost, st));
        QTTS_CHECK_HIP(hipStreamSynchronize(st));
    };
    const int total = g.max_new - 1;
    int left = std::min(n, total - g.launched);
    while (!g.done && left > 0) {
        if (!use_graph) {
            t->set_attn_mode(t->T0 + g.launched + 1);
            t->frame_step(g.sp, g.eos, g.min_new, g.max_new, g.codes, g.hidden, g.max_frames, st);
            ++g.launched; --left;
            poll();
            continue;
        }
        const int burst = std::min(8, left);
        hipGraphExec_t ge = t->ensure_graph(t->T0 + g.launched + burst, st, [&] {
            t->frame_step(g.sp, g.eos, g.min_new, g.max_new, g.codes, g.hidden, g.max_frames, st); });
        for (int i = 0; i < burst; ++i) QTTS_CHECK_HIP(hipGraphLaunch(ge, st));
        g.launched += burst; left -= burst;
        poll();
    }
}

int qtts_talker_stream_begin(qtts_talker* t, const qtts_sampling* sp, int32_t max_new_tokens, int32_t min_new_tokens,
                             int32_t eos_token_id, const int32_t* suppress_host, int32_t n_suppress, int64_t* codes_dev,
                             float* hidden_dev, void* stream) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(t && sp && codes_dev, QTTS_ERR_ARG, "null argument");
    QTTS_REQUIRE(t->prefilled, QTTS_ERR_STATE, "stream_begin: prefill() first");
    QTTS_REQUIRE(!t->profile, QTTS_ERR_STATE, "stream_begin: not available in profile mode");
    QTTS_REQUIRE(!t->tf.codes, QTTS_ERR_STATE, "stream_begin: teacher forcing is a qtts_talker_generate mode (clear it with set_teacher(NULL))");
    QTTS_REQUIRE(max_new_tokens >= 1, QTTS_ERR_ARG, "max_new_tokens >= 1");
    QTTS_REQUIRE(t->T0 + max_new_tokens <= t->cfg.max_seq, QTTS_ERR_LIMIT, "prompt + max_new_tokens exceeds max_seq");
    QTTS_REQUIRE(eos_token_id >= 0 && eos_token_id < t->cfg.vocab_size, QTTS_ERR_ARG, "eos_token_id");
    hipStream_t st = (hipStream_t)stream;
    const int V = t->cfg.vocab_size;
    t->prefilled = false;
    {
        std::vector<unsigned char> m(V, 0);
        for (int i = 0; i < n_suppress; ++i) {
            QTTS_REQUIRE(suppress_host[i] >= 0 && suppress_host[i] < V, QTTS_ERR_ARG, "suppress token out of range");
            m[suppress_host[i]] = 1;
        }
        QTTS_CHECK_HIP(hipMemcpyAsync(t->suppress.p, m.data(), V, hipMemcpyHostToDevice, st));
        const unsigned long long seed = sp->seed;
        QTTS_CHECK_HIP(hipMemcpyAsync(t->seed_d.p, &seed, 8, hipMemcpyHostToDevice, st));
        QTTS_CHECK_HIP(hipStreamSynchronize(st));
    }
    t->gen_cap = max_new_tokens;
    t->generated.ensure((size_t)t->B * max_new_tokens * 4);
    auto& g = t->sg;
    g = qtts_talker::StreamGen{};
    g.sp = *sp; g.eos = eos_token_id; g.min_new = min_new_tokens; g.max_new = max_new_tokens;
    g.max_frames = std::max(1, max_new_tokens - 1); g.codes = codes_dev; g.hidden = hidden_dev;
    t->frames_run = 0;
    t->sample_talker(*sp, eos_token_id, min_new_tokens, max_new_tokens, st);      // token 0
    QTTS_CHECK_HIP(hipMemcpyAsync(&g.done, t->ss.done, 4, hipMemcpyDeviceToHost, st));
    QTTS_CHECK_HIP(hipStreamSynchronize(st));
    // the cached frame graph is reusable only when everything it baked in is unchanged (as in qtts_talker_generate)
    qtts_talker::GraphKey key;
    memset(&key, 0, sizeof(key));
    key.B = t->B; key.Tt = t->Tt; key.eos = g.eos; key.min_new = g.min_new; key.max_new = g.max_new; key.max_frames = g.max_frames;
    key.do_sample = sp->do_sample; key.top_k = sp->top_k; key.sub_do_sample = sp->subtalker_dosample;
    key.sub_top_k = sp->subtalker_top_k; key.top_p = sp->top_p; key.temperature = sp->temperature; key.rep = sp->repetition_penalty;
    key.sub_top_p = sp->subtalker_top_p; key.sub_temperature = sp->subtalker_temperature; key.codes = codes_dev;
    key.hidden = hidden_dev; key.trailing = t->trailing.p; key.tts_pad = t->tts_pad.p; key.generated = t->generated.p;
    if (!t->cfg.use_graph || !t->any_graph() || !(key == t->graph_key)) { t->destroy_graph(); t->graph_nodes = 0; }
    t->graph_key = key;
    g.active = true;
    QTTS_API_END
}

int qtts_talker_stream_step(qtts_talker* t, int32_t max_frames_now, int32_t* frames_total_host, int32_t* finished_host,
                            void* stream) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(t && frames_total_host && finished_host, QTTS_ERR_ARG, "null argument");
    QTTS_REQUIRE(t->sg.active, QTTS_ERR_STATE, "stream_step: stream_begin() first");
    QTTS_REQUIRE(max_frames_now >= 1, QTTS_ERR_ARG, "max_frames_now >= 1");
    hipStream_t st = (hipStream_t)stream;
    auto& g = t->sg;
    stream_launch_frames(t, max_frames_now, st);
    QTTS_CHECK_HIP(hipStreamSynchronize(st));
    int fin[6];
    copy_on_stream(fin, t->ss.n_generated, sizeof(fin), hipMemcpyDeviceToHost, st);
    if (fin[5]) g.active = false;                        // (no packet of a burst that lost a fused launch is handed out)
    t->check_fused_flag(fin[5], "stream_step");
    // frames whose codes are final: every launched step that ran before the latch; after the latch exactly final_count - 1
    const int valid = fin[3] ? fin[4] - 1 : g.launched;
    *frames_total_host = std::min(valid, g.launched);
    *finished_host = (fin[3] || g.launched >= g.max_new - 1) ? 1 : 0;
    QTTS_API_END
}

int qtts_talker_stream_end(qtts_talker* t, int64_t* tokens_dev, int32_t* n_frames_host, void* stream) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(t && n_frames_host, QTTS_ERR_ARG, "null argument");
    QTTS_REQUIRE(t->sg.active, QTTS_ERR_STATE, "stream_end: stream_begin() first");
    hipStream_t st = (hipStream_t)stream;
    auto& g = t->sg;
    QTTS_CHECK_HIP(hipStreamSynchronize(st));
    int fin[6];
    copy_on_stream(fin, t->ss.n_generated, sizeof(fin), hipMemcpyDeviceToHost, st);
    g.active = false;
    t->check_fused_flag(fin[5], "stream_end");
    t->frames_run = g.launched;
    // an abandoned stream (ended before the stop condition) reports the frames produced so far
    const int n_tok = fin[3] ? fin[4] : fin[0];
    *n_frames_host = std::min(std::max(0, n_tok - 1), g.launched);
    if (tokens_dev) {
        const int B = t->B, cap = g.max_new;
        std::vector<int> h((size_t)B * cap);
        copy_on_stream(h.data(), t->generated.p, h.size() * 4, hipMemcpyDeviceToHost, st);
        std::vector<int64_t> w(h.size(), -1);
        for (int b = 0; b < B; ++b)
            for (int i = 0; i < n_tok && i < cap; ++i) w[(size_t)b * cap + i] = h[(size_t)b * cap + i];
        copy_on_stream(tokens_dev, w.data(), w.size() * 8, hipMemcpyHostToDevice, st);
    }
    QTTS_API_END
}

int qtts_talker_debug_logits(qtts_talker* t, float* logits_dev, void* stream) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(t && logits_dev, QTTS_ERR_ARG, "null argument");
    QTTS_CHECK_HIP(hipMemcpyAsync(logits_dev, t->logits.p, (size_t)t->B * t->cfg.vocab_size * 4, hipMemcpyDeviceToDevice,
                                  (hipStream_t)stream));
    QTTS_API_END
}
int qtts_talker_debug_cp_logits(qtts_talker* t, float* logits_dev, void* stream) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(t && logits_dev, QTTS_ERR_ARG, "null argument");
    const size_t Vc = (size_t)t->cfg.cp_vocab_size;
    for (int j = 0; j < t->cfg.num_code_groups - 1; ++j)       // (the engine's rows are [pass][64 rows][cp vocab])
        QTTS_CHECK_HIP(hipMemcpyAsync(logits_dev + (size_t)j * t->B * Vc, t->cp_logits.as<float>() + (size_t)j * 64 * Vc, (size_t)t->B * Vc * 4,
                                      hipMemcpyDeviceToDevice, (hipStream_t)stream));
    QTTS_API_END
}
// DIAGNOSTIC (not in include/qtts.h): which XCD runs workgroup i of a launch?  (cp_mlp.hip slices its exchange by `blockIdx % 8`.)
__global__ void xcc_probe_kernel(int* out) {
#ifndef QTTS_HOST_EMU
    if (threadIdx.x == 0) out[blockIdx.x] = (int)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;       // HW_REG_XCC_ID
#else
    if (threadIdx.x == 0) out[blockIdx.x] = blockIdx.x & 7;
#endif
}
int qtts_debug_xcc_map(int32_t grid, int32_t* out_host, void* stream) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(grid >= 1 && grid <= 65536 && out_host, QTTS_ERR_ARG, "xcc_map: grid 1..65536");
    DevBuf d; d.alloc((size_t)grid * 4);
    hipLaunchKernelGGL(xcc_probe_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d.as<int>());
    QTTS_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    QTTS_CHECK_HIP(hipMemcpy(out_host, d.p, (size_t)grid * 4, hipMemcpyDeviceToHost));
    QTTS_API_END
}
// DIAGNOSTIC (not in include/qtts.h): workgroups of the layer launch one compute unit holds at once, by the occupancy API
int qtts_debug_cp_layer_occupancy(int32_t H, int32_t I, int32_t bf16) {
    try { return qtts::cp_layer_blocks_per_cu(H, I, bf16 != 0); } catch (...) { return -1; }
}
int qtts_talker_get_stats(qtts_talker* t, qtts_talker_stats* out) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(t && out, QTTS_ERR_ARG, "null argument");
    out->frames_run = t->frames_run; out->graph_nodes = t->graph_nodes; out->weight_bytes_per_frame = t->weight_bytes_frame;
    out->gemm_ms_last = t->prof_ms; out->gemm_launches_last = t->prof_launches;
    out->long_graphs = (int32_t)t->graph_long.size(); out->attn_nsplit_last = t->attn_nsplit_active; out->attn_span_last = t->attn_span_active;
    out->cp_fused_per_step = t->cp_fused_slot ? t->cp_fused_per_step : 0;
    out->cp_fused_launches_last = (int64_t)out->cp_fused_per_step * t->frames_run;
    out->cp_fused_giveups = t->cp_fused_giveups; out->cp_fused_capacity = t->fused_capacity; out->cp_fused_active = t->cp_fused_slot ? 1 : 0;
    out->cp_mlp_per_step = t->cp_fused_slot ? t->cp_mlp_per_step : 0;
    out->cp_layer_per_step = t->cp_fused_slot ? t->cp_layer_per_step : 0;
    out->ks_split_per_step = t->ks_split_env ? t->ks_split_per_step : 0; out->reserved0 = 0;
    QTTS_API_END
}
int qtts_talker_get_gemm_profile(qtts_talker* t, qtts_gemm_class* out, int32_t cap, int32_t* n) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(t && n && (out || cap == 0), QTTS_ERR_ARG, "null argument");
    *n = (int32_t)t->prof_classes.size();
    for (int i = 0; i < *n && i < cap; ++i) out[i] = t->prof_classes[i];
    QTTS_API_END
}
__global__ void null_kernel(int* p, int mode) {
    if (mode == 1) { if (*p) return; }
    if (mode == 2) { __shared__ int s[64]; s[threadIdx.x & 63] = threadIdx.x; __syncthreads(); if (threadIdx.x == 0 && s[5] == 12345) *p = 1; }
}
// DEBUG: hipGraph chain of trivial kernels (grid x block) -> us per launch: the dependent-kernel boundary here.
int qtts_debug_null_chain(int32_t grid, int32_t block, int32_t mode, int32_t iters, int32_t reps, double* us) {
    QTTS_API_BEGIN
    DevBuf d; d.alloc(64); QTTS_CHECK_HIP(hipMemset(d.p, 0, 64));
    hipStream_t st; QTTS_CHECK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipGraph_t gr; hipGraphExec_t ge;
    QTTS_CHECK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(null_kernel, dim3(grid), dim3(block), 0, st, d.as<int>(), mode);
    QTTS_CHECK_HIP(hipStreamEndCapture(st, &gr));
    QTTS_CHECK_HIP(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
    QTTS_CHECK_HIP(hipGraphLaunch(ge, st)); QTTS_CHECK_HIP(hipStreamSynchronize(st));
    hipEvent_t a, b; QTTS_CHECK_HIP(hipEventCreate(&a)); QTTS_CHECK_HIP(hipEventCreate(&b));
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        QTTS_CHECK_HIP(hipEventRecord(a, st)); QTTS_CHECK_HIP(hipGraphLaunch(ge, st)); QTTS_CHECK_HIP(hipEventRecord(b, st));
        QTTS_CHECK_HIP(hipStreamSynchronize(st));
        float ms = 0; QTTS_CHECK_HIP(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms);
    }
    *us = 1000.0 * best / iters;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b); (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(gr); (void)hipStreamDestroy(st);
    QTTS_API_END
}

// DEBUG/perf tooling (not part of the product surface): time a hipGraph chain of `iters` identical skinny GEMM
// launches (bf16, packed random-ish weights) and return the average microseconds per launch.
int qtts_debug_skinny_chain(int32_t N, int32_t K, int32_t M, int32_t act, int32_t with_norm, int32_t with_res,
                            int32_t ablate, int32_t iters, int32_t reps, double* us_per_launch) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(us_per_launch && iters > 0 && reps > 0, QTTS_ERR_ARG, "bad argument");
    DevBuf W, x, out, res, ssin, done;
    // QTTS_DEBUG_WBUFS=n: the launches of the chain rotate through n copies of the operator (n x bytes beyond the caches = every launch
    // streams from HBM / the Infinity Cache, as in the frame step; 1 = the same L2-resident operator every time)
    const int wbufs = [] { const char* e = QTTS_ENV("QTTS_DEBUG_WBUFS"); return e && atoi(e) > 0 ? atoi(e) : 1; }();
    const size_t wbytes = skinny_packed_bytes(N, K, true);
    W.alloc(wbytes * wbufs);
    {
        std::vector<uint16_t> h((size_t)N * K);
        uint32_t r = 12345;
        for (auto& v : h) { r = r * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 + ((r >> 16) & 0x3ff) - 0x200 + ((r >> 31) << 15)); }
        for (int i = 0; i < wbufs; ++i) QTTS_CHECK_HIP(hipMemcpy(static_cast<char*>(W.p) + i * wbytes, h.data(), wbytes, hipMemcpyHostToDevice));
    }
    const bool glu = act == ACT_SWIGLU || act == ACT_SWIGLU8;
    const int No = glu ? N / 2 : N;
    x.alloc((size_t)64 * K * 4); out.alloc((size_t)64 * No * 4); res.alloc((size_t)64 * No * 4);
    ssin.alloc(64 * 8); done.alloc(64);
    QTTS_CHECK_HIP(hipMemset(x.p, 0x3c, x.bytes));
    QTTS_CHECK_HIP(hipMemset(res.p, 0, res.bytes)); QTTS_CHECK_HIP(hipMemset(ssin.p, 0x3c, ssin.bytes));
    QTTS_CHECK_HIP(hipMemset(done.p, 0, done.bytes));
    SkinnyParams p{};
    p.x = x.as<float>(); p.ldx = K; p.M = M; p.Wp = W.p; p.N = N; p.K = K; p.eps = 1e-6f; p.act = act;
    p.x_bf16 = 1;                                   // as in the frame step: the producer's bf16 copy of x
    if (!glu) { int fs = 16; while (fs > 4 && N / fs < 192) fs /= 2; p.fs = fs; }     // the engine's choose_fs
    if (const char* e = QTTS_ENV("QTTS_DEBUG_FS")) { if (!glu && atoi(e) > 0) p.fs = atoi(e); }
    if (with_norm) { p.norm = 1; p.ss_in = ssin.as<float>(); }
    if (with_res) { p.res = res.as<float>(); p.ldr = No; }
    p.out = out.as<float>(); p.ldo = No; p.done_flag = done.as<int>(); p.ablate = ablate;
    hipStream_t st;
    QTTS_CHECK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    launch_skinny(p, true, st);
    QTTS_CHECK_HIP(hipStreamSynchronize(st));
    hipGraph_t gr; hipGraphExec_t ge;
    QTTS_CHECK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < iters; ++i) { p.Wp = static_cast<const char*>(W.p) + (size_t)(i % wbufs) * wbytes; launch_skinny(p, true, st); }
    QTTS_CHECK_HIP(hipStreamEndCapture(st, &gr));
    QTTS_CHECK_HIP(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
    QTTS_CHECK_HIP(hipGraphLaunch(ge, st));
    QTTS_CHECK_HIP(hipStreamSynchronize(st));
    hipEvent_t a, b;
    QTTS_CHECK_HIP(hipEventCreate(&a)); QTTS_CHECK_HIP(hipEventCreate(&b));
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        QTTS_CHECK_HIP(hipEventRecord(a, st));
        QTTS_CHECK_HIP(hipGraphLaunch(ge, st));
        QTTS_CHECK_HIP(hipEventRecord(b, st));
        QTTS_CHECK_HIP(hipStreamSynchronize(st));
        float ms = 0; QTTS_CHECK_HIP(hipEventElapsedTime(&ms, a, b));
        best = std::min(best, ms);
    }
    *us_per_launch = 1000.0 * best / iters;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b); (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(gr); (void)hipStreamDestroy(st);
    QTTS_API_END
}

int qtts_talker_set_teacher(qtts_talker* t, const int64_t* forced_codes_dev, int32_t n_frames, int32_t* own_dev,
                            const int32_t* logit_slots_dev, float* logits_trace_dev) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(t, QTTS_ERR_ARG, "null handle");
    QTTS_REQUIRE(!t->sg.active, QTTS_ERR_STATE, "set_teacher while a streaming generation is active");
    if (!forced_codes_dev) { t->tf = qtts_talker::Teacher{}; }
    else {
        QTTS_REQUIRE(n_frames >= 1 && own_dev, QTTS_ERR_ARG, "set_teacher: n_frames >= 1 and an output buffer for the own choices");
        QTTS_REQUIRE((logit_slots_dev == nullptr) == (logits_trace_dev == nullptr), QTTS_ERR_ARG, "set_teacher: slots and trace go together");
        t->tf.codes = forced_codes_dev; t->tf.F = n_frames; t->tf.own = own_dev; t->tf.slots = logit_slots_dev; t->tf.trace = logits_trace_dev;
    }
    QTTS_API_END
}

int qtts_talker_set_profile(qtts_talker* t, int32_t enable) {
    QTTS_API_BEGIN
    QTTS_REQUIRE(t, QTTS_ERR_ARG, "null handle");
    QTTS_REQUIRE(enable >= 0 && enable <= 2, QTTS_ERR_ARG, "set_profile: 0 off, 1 per-launch events on the real frame step, 2 isolated GEMM-only graph");
    t->profile = enable;
    QTTS_API_END
}

}  // extern "C"
