// glue.h -- device-resident loop state and the small frame-step kernels (sampling.hip).
#pragma once
#include "common.h"

namespace qtts {

// Everything that varies per decode step lives in device memory so the captured frame step is static.
struct StepState {
    int* n_generated;   // tokens sampled so far (HF cur_len of the generated part)
    int* gen_step;      // talker `generation_step` of the NEXT decode forward (= frame index)
    int* kv_len;        // talker KV length before the next decode forward
    int* done;          // latched when HF's stopping criteria would break
    int* final_count;   // n_generated at that moment
    int* unfinished;    // [B]
    int* frame_serial;  // optional: +1 per frame step, never reset (the hand-off tags of attention.hip's cp_attn_o_kernel derive from it)
};
void launch_sample_finish(const StepState& s, int B, int max_new_tokens, hipStream_t st);

// teacher forcing (diagnostic): record the engine's own greedy choices, then replace them by `codes` (sampling.hip)
struct TeacherParams {
    const int64_t* codes;       // forced codes [B][F][G]
    int F, G, B, V;
    int* own;                   // out: own choices [B][F+1][G] (entry [b][i][0] = own cb-0 token i; [b][f][1+j] = own sub-code j of frame f)
    const int* slots;           // [F+1]: logits-trace slot of token step i, or -1
    float* trace;               // [n_slots][B][V] raw cb-0 logits
    const float* logits;        // the engine's cb-0 logits buffer [B][V]
    int* cur_tok; int* sub; int sub_stride; int* generated; int gen_stride;
    StepState st;
};
void launch_teacher(const TeacherParams& p, int which /*0 trace, 1 token, 2 sub-codes*/, hipStream_t st);

struct CpGatherParams {
    int pass, B, H;
    const float* past_hidden;   // [B][H]
    const float* talker_emb;    // talker codec_embedding [vocab][H]
    const int* cur_tok;         // [B]
    const float* cp_emb;        // code-predictor codec_embedding [G-1][cp_vocab][H]
    int cp_vocab;
    const int* sub; int sub_stride;   // [B][G-1] sub-codes sampled so far
    float* out;                 // [rows][H]
    unsigned short* out16;      // optional bf16 copy [rows][H]
    const int* done;
};
void launch_cp_gather(const CpGatherParams& p, hipStream_t st);

struct EmbedSumParams {
    int B, H, G, cp_vocab;
    const float* talker_emb; const float* cp_emb;
    const int* cur_tok; const int* sub; int sub_stride;
    const float* trailing; int Tt; const float* tts_pad;
    const float* past_hidden;
    float* x_out; unsigned short* x_out16;   // fp32 hidden state (+ optional bf16 copy)
    int64_t* codes_out; float* hidden_out; int max_frames;
    StepState st;
};
void launch_embed_sum(const EmbedSumParams& p, hipStream_t st);

// y = g * (x * rsqrt(mean(x^2)+eps)) per row (final talker norm -> past_hidden), early-exit on *done
void launch_apply_norm(const float* x, int ldx, const float* g, float eps, float* y, int ldy, int rows, int C,
                       const int* done, hipStream_t st, unsigned short* y16 = nullptr /* optional bf16 copy [rows][ldy] */);
// ss[r] = sum_c x[r][c]^2 (only for GEMMs that cannot stage x through LDS)
void launch_row_ss(const float* x, int ldx, int rows, int C, float* ss, const int* done, hipStream_t st);

// ---- prompt assembly (M:2076-2269): every prompt / trailing row is a sum of at most one text-side row and one codec-side
// term; the host resolves WHICH (integers only), the device does all the floating point.
struct AssembleParams {
    const int32_t* desc;        // [rows][4] = {text_row, codec_id, spk_row, ref_frame}; -1 = absent
    int rows, H, G, cp_vocab, vocab;
    const float* proj; int proj_rows;           // projected text rows [proj_rows][H]
    const float* talker_emb; const float* cp_emb;
    const float* spk; int n_spk;                // speaker vectors [n_spk][H] (voice clone)
    const int64_t* ref_codes; int n_ref;        // ICL reference codes [n_ref][G]
    float* out;                                 // [rows][H]
    int* err;                                   // set to 1 on an out-of-range index (row is zeroed)
};
void launch_assemble_rows(const AssembleParams& p, hipStream_t st);
// out[r][:] = table[ids[r]][:] as fp32 (table fp32 or bf16); out-of-range ids set *err and give a zero row
void launch_gather_rows(const void* table, bool table_bf16, int64_t n_table, int C, const int64_t* ids, int rows, float* out,
                        int* err, hipStream_t st);

}  // namespace qtts
