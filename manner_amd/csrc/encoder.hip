// Encoder handle: packed PLM weights + the per-chunk layer schedule behind
// manner_hip_encode_cls (K1-K7: MannerTextEncoder.forward, reference
// manner/models/components/news_encoder.py:29-37 over HF BertModel.forward,
// transformers/models/bert/modeling_bert.py:623-686).
//
// Tokens of a chunk of news are PACKED (no padding rows): x[cu[n] + t] is token t of news n.
// Per layer:  qkv = x Wqkv^T + b          (one GEMM over the fused [3H, H] weight)
//             ctx = attention(qkv)        (per news, per head)
//             pre = ctx Wo^T + b + x      (f32)         x = LN(pre)
//             ffn = gelu(x W1^T + b)                    pre = ffn W2^T + b + x ; x = LN(pre)
// Activations x/qkv/ctx/ffn are bf16 / f16 (MANNER_HIP_PREC_BF16 / _F16) or f32 (MANNER_HIP_PREC_F32); LayerNorm statistics,
// softmax and all accumulation are f32 in both modes.
// bf16 with 256-tileable shapes runs the DEFERRED-LayerNorm schedule: no LayerNorm kernel and no f32 `pre`
// round trip — the residual stream keeps the pre-LayerNorm sums (bf16) plus {mean, rstd} per row, the
// normalisation is folded into the consuming GEMM (gamma into the weight, mean/rstd into its epilogue) and
// rebuilt on the fly where the normalised value is the residual (see DlnAux in gemm.hip).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "common.h"

namespace manner {

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

struct LayerWeights {
  void* wqkv = nullptr;   // [3H, H]
  void* wo = nullptr;     // [H, H]
  void* w1 = nullptr;     // [I, H]
  void* w2 = nullptr;     // [H, I]
  void* wqkv_f = nullptr; // 16-bit modes, deferred LayerNorm: gamma_in o Wqkv   (gamma_in: the LayerNorm feeding this layer)
  void* w1_f = nullptr;   // 16-bit modes, deferred LayerNorm: ln1g o W1
  float *cq1 = nullptr, *cf1 = nullptr;   // column sums of the ROUNDED folded weights (they differ between bf16 and f16)
};
struct LayerParams {      // f32, shared by both precisions
  float *bqkv, *bo, *ln1g, *ln1b, *b1, *b2, *ln2g, *ln2b;
  // deferred LayerNorm (see fold_layernorm): folded biases of wqkv_f and w1_f, residual biases
  float *cq2 = nullptr, *cf2 = nullptr, *bo_res = nullptr, *b2_res = nullptr;
};

}  // namespace manner

struct manner_hip_encoder {
  manner_hip_encoder_config cfg;
  uint32_t precisions = 0;
  float *word = nullptr, *pos = nullptr, *type0 = nullptr, *embg = nullptr, *embb = nullptr;
  std::vector<manner::LayerParams> params;
  std::vector<manner::LayerWeights> w[5];   // [MANNER_HIP_PREC_*]; BF16X3 / F16X3: 16-bit [out, 3*in] split weights
  int32_t* status = nullptr;                // device flag word
  // arrival counters of the GEMMs' in-launch row-statistics finalize (gemm.hip nres_fan_in): ARRIVE_CAP zeroed ints per stream lane,
  // one per row panel of the chunk in flight on that lane; the workgroup that completes a panel puts its counter back to 0
  static constexpr int ARRIVE_CAP = 4096;
  int32_t* arrive = nullptr;
  std::vector<void*> allocs;
  // chunks alternate between the caller's stream and a side stream (fork/join by events) so the
  // HBM-bound phases of one chunk overlap the MFMA-bound phases of the other (the side stream starts half
  // a layer late; in lockstep the two would just run the same kernels together).  Bit-identical results,
  // +4 % measured.  MANNER_HIP_STREAMS=1 disables
  int n_streams = 2;
  bool defer_ln = false;              // bf16 deferred-LayerNorm schedule available (folded weights packed); MANNER_HIP_DEFER_LN=0 disables
  static constexpr int MAX_STREAMS = 4;
  hipStream_t side[MAX_STREAMS] = {};      // side[0] unused (the caller's stream)
  hipEvent_t join_ev[MAX_STREAMS] = {};
  hipEvent_t phase_ev[MAX_STREAMS] = {};   // recorded mid-layer in the first chunk of stream i: stream i+1 starts there
  // opt-in per-launch timing (manner_hip_encoder_profile)
  bool profiling = false;
  struct Span { hipEvent_t a, b; int cls; };
  std::vector<Span> spans;                  // recorded, not yet read
  std::vector<hipEvent_t> free_events;
};

namespace manner {
namespace {

// activation dtype / element size of a precision mode (BF16X3 keeps f32 activations)
inline DType act_dtype(int prec) { return prec == MANNER_HIP_PREC_BF16 ? DT_BF16 : prec == MANNER_HIP_PREC_F16 ? DT_F16 : DT_F32; }
inline size_t act_bytes(int prec) { return (prec == MANNER_HIP_PREC_BF16 || prec == MANNER_HIP_PREC_F16) ? 2 : 4; }
// split-operand modes: f32 activations, every GEMM on the 16-bit MFMA over [hi | hi | lo] x [hi | lo | hi]
inline bool is_x3(int prec) { return prec == MANNER_HIP_PREC_BF16X3 || prec == MANNER_HIP_PREC_F16X3; }
inline DType x3_dtype(int prec) { return prec == MANNER_HIP_PREC_F16X3 ? DT_F16 : DT_BF16; }

int dev_alloc(manner_hip_encoder* e, size_t bytes, void** out) {
  MANNER_HIP_TRY(hipMalloc(out, bytes));
  e->allocs.push_back(*out);
  return MANNER_HIP_OK;
}
int dev_copy_f32(manner_hip_encoder* e, const float* src, size_t n, float** out, hipStream_t s) {
  int rc = dev_alloc(e, n * sizeof(float), (void**)out);
  if (rc) return rc;
  MANNER_HIP_TRY(hipMemcpyAsync(*out, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
  return MANNER_HIP_OK;
}
// place rows of an f32 [rows, cols] matrix at dst (+row offset) in the precision's element type
int pack_matrix(int prec, const float* src, size_t n, void* dst, size_t elem_off, hipStream_t s) {
  if (prec == MANNER_HIP_PREC_BF16 || prec == MANNER_HIP_PREC_F16)
    return convert_f32_to_16(prec == MANNER_HIP_PREC_F16 ? DT_F16 : DT_BF16, src, static_cast<char*>(dst) + elem_off * 2, (int64_t)n, s);
  MANNER_HIP_TRY(hipMemcpyAsync(static_cast<float*>(dst) + elem_off, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
  return MANNER_HIP_OK;
}

// brackets one launch with events when profiling is on
struct ProfScope {
  manner_hip_encoder* e; hipStream_t s; hipEvent_t a = nullptr, b = nullptr; int cls;
  static hipEvent_t get(manner_hip_encoder* e) {
    hipEvent_t ev = nullptr;
    if (!e->free_events.empty()) { ev = e->free_events.back(); e->free_events.pop_back(); }
    else if (hipEventCreate(&ev) != hipSuccess) ev = nullptr;
    return ev;
  }
  ProfScope(manner_hip_encoder* enc, hipStream_t st, int c) : e(enc), s(st), cls(c) {
    if (!e->profiling) return;
    a = get(e); b = get(e);
    if (a && b) (void)hipEventRecord(a, s);
  }
  ~ProfScope() {
    if (!a || !b) return;
    (void)hipEventRecord(b, s);
    e->spans.push_back({a, b, cls});
  }
};

struct Workspace {
  int32_t *lens, *cu, *m_total;      // m_total[0] = packed tokens, m_total[1] = news of the chunk
  float* pre;
  void *x, *qkv, *ctx, *ffn;
  void *xcls, *qcls;                 // compact [CLS] rows of the last layer
  void *mr_in, *mr_mid, *part;       // deferred LayerNorm: {mean, rstd} per row (layer input / after attention), partial sums
  int32_t* arrive = nullptr;         // this lane's arrival counters (handle-owned, not carved from the caller's workspace: they must start at 0)
  void* a3;                          // x3 modes: split copy [M, 3*max(H, I)] (16-bit) of an I-wide (or compact-row) f32 A operand
  void* a3h;                         // x3 modes: split copy [M, 3H] of the H-wide A operand (layer input / ctx / LN1 output), written by its producer
  // last layer, [CLS] rows only: the rows of several chunks are collected (acc_*) and the tail of the layer — output
  // projection, LayerNorm, FFN, LayerNorm — runs ONCE over them (run_cls_tail) instead of as 5 small launches per chunk
  void *acc_x, *acc_ctx, *acc_x1;    // [cls_cap + 256, H]: layer input (normalised) / attention output / LN1 output of the [CLS] rows
  int32_t* cls_total;                // device scalar: rows collected
  int64_t cls_cap;
};

constexpr int64_t CLS_CAP = 16384;   // [CLS] rows collected per stream before the tail runs (bounded by the chunk capacity)

size_t carve(const manner_hip_encoder* e, int64_t max_news, int64_t m_cap, int prec, char* base, Workspace* ws) {
  const size_t es = act_bytes(prec);
  const size_t H = e->cfg.hidden, I = e->cfg.intermediate;
  size_t off = 0;
  if (ws) { ws->a3 = nullptr; ws->a3h = nullptr; }
  auto take = [&](size_t bytes) { size_t o = off; off += (size_t)round_up((int64_t)bytes, 256); return base ? base + o : nullptr; };
  char* p;
  p = take((size_t)max_news * 4); if (ws) ws->lens = (int32_t*)p;
  p = take((size_t)(max_news + 1) * 4); if (ws) ws->cu = (int32_t*)p;
  p = take(256); if (ws) ws->m_total = (int32_t*)p;
  p = take((size_t)m_cap * H * 4); if (ws) ws->pre = (float*)p;
  p = take((size_t)m_cap * H * es); if (ws) ws->x = p;
  p = take((size_t)m_cap * 3 * H * es); if (ws) ws->qkv = p;
  p = take((size_t)m_cap * H * es); if (ws) ws->ctx = p;
  p = take((size_t)m_cap * I * es); if (ws) ws->ffn = p;
  const size_t n_pad = (size_t)round_up(max_news, 256);
  p = take(n_pad * H * es); if (ws) ws->xcls = p;
  p = take(n_pad * H * es); if (ws) ws->qcls = p;
  p = take((size_t)m_cap * 8); if (ws) ws->mr_in = p;
  p = take((size_t)m_cap * 8); if (ws) ws->mr_mid = p;
  p = take((size_t)m_cap * (H / 64) * 8); if (ws) ws->part = p;
  if (is_x3(prec)) {
    p = take((size_t)m_cap * 3 * (H > I ? H : I) * 2); if (ws) ws->a3 = p;
    p = take((size_t)m_cap * 3 * H * 2); if (ws) ws->a3h = p;
  }
  const int64_t cls_cap = m_cap < CLS_CAP ? m_cap : CLS_CAP;     // the tail borrows pre / ffn (m_cap rows) as scratch
  if (ws) ws->cls_cap = cls_cap;
  p = take((size_t)(cls_cap + 256) * H * es); if (ws) ws->acc_x = p;
  p = take((size_t)(cls_cap + 256) * H * es); if (ws) ws->acc_ctx = p;
  p = take((size_t)(cls_cap + 256) * H * es); if (ws) ws->acc_x1 = p;
  p = take(256); if (ws) ws->cls_total = (int32_t*)p;
  return off;
}

// `hidden_layers` < 0: the [CLS] path (out = f32 [n_news, H]).  Otherwise the first `hidden_layers` layers run in
// full and `out` receives their hidden states in the padded [n_news, lp, H] layout (dtype `hidden_dt`).
// `phase_mark` (two-stream mode) is recorded on `s` half a layer in — or, when no full layer runs (n_layers = 0, a
// one-layer model) or the chunk fails, when the function returns: whoever waits on it always waits on an event of
// THIS call, so a side stream is ordered after everything the caller's stream had queued before it.
struct PhaseGuard {
  hipEvent_t ev; hipStream_t s; bool done = false;
  void mark() { if (ev && !done) { (void)hipEventRecord(ev, s); done = true; } }
  ~PhaseGuard() { mark(); }
};

// [CLS] path: `cls_off` = rows already collected in ws.acc_* — this chunk's n_news rows are appended there and `out` is
// not written (run_cls_tail does, for all collected rows at once).
int encode_chunk(manner_hip_encoder* e, const int64_t* ids, const int64_t* mask, int64_t n_news, int64_t lp,
                 int64_t m_bound, int64_t expect_tokens, int prec, void* out, const Workspace& ws, hipStream_t s,
                 hipEvent_t phase_mark = nullptr, int hidden_layers = -1, DType hidden_dt = DT_F32, int64_t cls_off = 0) {
  PhaseGuard phase{phase_mark, s};
  const manner_hip_encoder_config& c = e->cfg;
  const DType dt = act_dtype(prec);
  const int H = c.hidden, I = c.intermediate;
  int rc;
  {
    ProfScope ps(e, s, MANNER_HIP_PROF_LENGTHS);
    if ((rc = lengths_and_offsets(mask, n_news, lp, ws.lens, ws.cu, ws.m_total, m_bound, expect_tokens, e->status, s))) return rc;
  }
  const int pos_offset = c.arch == MANNER_HIP_ARCH_ROBERTA ? c.pad_id + 1 : 0;
#define PROF_STEP(cls, call) { ProfScope ps(e, s, cls); if ((rc = (call))) return rc; }
  const size_t es = act_bytes(prec);
  if (is_16bit(dt) && e->defer_ln) {
    // ---- deferred-LayerNorm schedule: ws.x holds the pre-LayerNorm sums, mr_in / mr_mid their row statistics
    PROF_STEP(MANNER_HIP_PROF_EMBED, embed_raw(dt, ids, n_news, lp, ws.cu, e->word, e->pos, e->type0, H, c.ln_eps, pos_offset,
                                               c.vocab, c.max_pos, ws.x, ws.mr_in, e->status, s))
    const int groups = H / 64;
    for (int l = 0; l < c.layers; ++l) {
      const LayerWeights& w = e->w[prec][l];
      const LayerParams& p = e->params[l];
      const float* g_in = l == 0 ? e->embg : e->params[l - 1].ln2g;     // the LayerNorm that feeds this layer
      const float* b_in = l == 0 ? e->embb : e->params[l - 1].ln2b;
      if (l == hidden_layers) break;
      if (hidden_layers < 0 && l + 1 == c.layers) {
        // last layer: K|V for every token, everything else on the compact [CLS] rows (plain weights)
        const int64_t n_bound = round_up(n_news, 256);
        const int32_t* n_total = ws.m_total + 1;
        const char* wkv = static_cast<const char*>(w.wqkv_f) + (size_t)H * H * 2;
        PROF_STEP(MANNER_HIP_PROF_GEMM_QKV, gemm_tn_dln(dt, EPI_NORM, ws.x, wkv, p.cq2 + H, w.cq1 + H, ws.mr_in, nullptr, ws.qkv, m_bound, 2 * H, H, ws.m_total, s, expect_tokens))
        char* ax = static_cast<char*>(ws.acc_x) + (size_t)cls_off * H * es;
        char* actx = static_cast<char*>(ws.acc_ctx) + (size_t)cls_off * H * es;
        PROF_STEP(MANNER_HIP_PROF_GATHER, gather_cls_ln(dt, ws.x, ws.mr_in, ws.cu, n_news, H, g_in, b_in, ax, s))
        PROF_STEP(MANNER_HIP_PROF_CLS_TAIL, gemm_tn(dt, dt, EPI_BIAS, ax, w.wqkv, p.bqkv, nullptr, ws.qcls, n_bound, H, H, n_total, s))
        PROF_STEP(MANNER_HIP_PROF_ATTENTION, attention_cls(dt, ws.qcls, ws.qkv, actx, ws.cu, n_news, c.heads, H, s))
        break;
      }
      PROF_STEP(MANNER_HIP_PROF_GEMM_QKV, gemm_tn_dln(dt, EPI_NORM, ws.x, w.wqkv_f, p.cq2, w.cq1, ws.mr_in, nullptr, ws.qkv, m_bound, 3 * H, H, ws.m_total, s, expect_tokens))
      PROF_STEP(MANNER_HIP_PROF_ATTENTION, attention_varlen(dt, ws.qkv, ws.ctx, ws.cu, n_news, c.heads, H, (int)lp, s))
      // the next {mean, rstd} are finished inside the producing GEMM where its kernel carries the fan-in (fin.done), else by dln_finalize
      const bool fan_in = ws.arrive && m_bound / 192 + 1 <= manner_hip_encoder::ARRIVE_CAP;
      DlnFinalize fin_mid{ws.mr_mid, fan_in ? ws.arrive : nullptr, c.ln_eps, false}, fin_in{ws.mr_in, fan_in ? ws.arrive : nullptr, c.ln_eps, false};
      PROF_STEP(MANNER_HIP_PROF_GEMM_OUT, gemm_tn_dln(dt, EPI_NRES, ws.ctx, w.wo, p.bo_res, g_in, ws.mr_in, ws.part, ws.x, m_bound, H, H, ws.m_total, s, expect_tokens, &fin_mid))
      if (!fin_mid.done) PROF_STEP(MANNER_HIP_PROF_LAYERNORM, dln_finalize(ws.part, groups, H, c.ln_eps, ws.mr_mid, m_bound, ws.m_total, s))
      if (l == 0) phase.mark();   // two-stream mode: the other stream starts half a layer later
      PROF_STEP(MANNER_HIP_PROF_GEMM_FFN1, gemm_tn_dln(dt, EPI_NORM_GELU, ws.x, w.w1_f, p.cf2, w.cf1, ws.mr_mid, nullptr, ws.ffn, m_bound, I, H, ws.m_total, s, expect_tokens))
      PROF_STEP(MANNER_HIP_PROF_GEMM_FFN2, gemm_tn_dln(dt, EPI_NRES, ws.ffn, w.w2, p.b2_res, p.ln1g, ws.mr_mid, ws.part, ws.x, m_bound, H, I, ws.m_total, s, expect_tokens, &fin_in))
      if (!fin_in.done) PROF_STEP(MANNER_HIP_PROF_LAYERNORM, dln_finalize(ws.part, groups, H, c.ln_eps, ws.mr_in, m_bound, ws.m_total, s))
    }
    if (hidden_layers >= 0) {   // the residual stream is still un-normalised: apply the LayerNorm that closes layer hidden_layers-1
      const float* g = hidden_layers == 0 ? e->embg : e->params[hidden_layers - 1].ln2g;
      const float* b = hidden_layers == 0 ? e->embb : e->params[hidden_layers - 1].ln2b;
      PROF_STEP(MANNER_HIP_PROF_GATHER, scatter_hidden(dt, hidden_dt, ws.x, ws.mr_in, ws.cu, n_news, lp, H, g, b, out, s))
    }
    return MANNER_HIP_OK;
  }
  {
    ProfScope ps(e, s, MANNER_HIP_PROF_EMBED);
    if ((rc = embed_layernorm(dt, ids, n_news, lp, ws.cu, e->word, e->pos, e->type0, e->embg, e->embb, H, c.ln_eps,
                              pos_offset, c.vocab, c.max_pos, ws.x, e->status, s)))
      return rc;
  }
  // BF16X3: f32 activations everywhere; each GEMM first splits its A operand into [hi | hi | lo] bf16 (ws.a3) and
  // runs on the bf16 MFMA against the [hi | lo | hi] split weight, depth 3K, f32 residual / output
  const bool x3 = is_x3(prec);
  const DType xdt = x3_dtype(prec);
  auto gemm = [&](DType out_dt, Epilogue epi, const void* A, const void* Wm, const float* bias, const void* res, void* Y,
                  int64_t mb, int N, int K, const int* mt) -> int {
    if (!x3) return gemm_tn(dt, out_dt, epi, A, Wm, bias, res, Y, mb, N, K, mt, s);
    int r = split3_rows(xdt, false, static_cast<const float*>(A), ws.a3, K, mb, mt, s);
    if (r) return r;
    return gemm_tn(xdt, DT_F32, epi == EPI_BIAS_RES ? EPI_BIAS_RES_F32 : epi, ws.a3, Wm, bias, res, Y, mb, N, 3 * K, mt, s);
  };
  const int full_layers = hidden_layers >= 0 ? hidden_layers : c.layers - 1;
  // x3 modes: the [hi | hi | lo] operand of every GEMM is written by the kernel that PRODUCES the tensor — the LayerNorms emit the
  // split of their output next to the f32 row, FFN1's epilogue writes gelu(.) straight as FFN2's split operand (the f32
  // intermediate never exists), the f32 attention writes its rows as the out-projection's split operand — instead of by a
  // separate pass over the f32 tensor in front of each GEMM (17 % of the mode's time in round 2); only the embedding output
  // still goes through split3_rows, once per chunk.
  if (x3) PROF_STEP(MANNER_HIP_PROF_LAYERNORM, split3_rows(xdt, false, static_cast<const float*>(ws.x), ws.a3h, H, m_bound, ws.m_total, s))
  for (int l = 0; l < full_layers && x3; ++l) {
    const LayerWeights& w = e->w[prec][l];
    const LayerParams& p = e->params[l];
    float* xf = static_cast<float*>(ws.x);
    PROF_STEP(MANNER_HIP_PROF_GEMM_QKV, gemm_tn(xdt, DT_F32, EPI_BIAS, ws.a3h, w.wqkv, p.bqkv, nullptr, ws.qkv, m_bound, 3 * H, 3 * H, ws.m_total, s))
    static const bool attn_valu = getenv("MANNER_HIP_ATTN_F32_VALU") != nullptr;     // (the A/B VALU kernel writes f32 rows)
    if (attn_valu) {
      PROF_STEP(MANNER_HIP_PROF_ATTENTION, attention_varlen(dt, ws.qkv, ws.ctx, ws.cu, n_news, c.heads, H, (int)lp, s))
      PROF_STEP(MANNER_HIP_PROF_LAYERNORM, split3_rows(xdt, false, static_cast<const float*>(ws.ctx), ws.a3h, H, m_bound, ws.m_total, s))
    } else {
      PROF_STEP(MANNER_HIP_PROF_ATTENTION, attention_varlen(dt, ws.qkv, ws.ctx, ws.cu, n_news, c.heads, H, (int)lp, s, ws.a3h, xdt))
    }
    PROF_STEP(MANNER_HIP_PROF_GEMM_OUT, gemm_tn(xdt, DT_F32, EPI_BIAS_RES_F32, ws.a3h, w.wo, p.bo, ws.x, ws.pre, m_bound, H, 3 * H, ws.m_total, s))
    PROF_STEP(MANNER_HIP_PROF_LAYERNORM, layernorm_rows_split(xdt, ws.pre, p.ln1g, p.ln1b, H, c.ln_eps, xf, ws.a3h, m_bound, ws.m_total, s))
    if (l == 0) phase.mark();
    PROF_STEP(MANNER_HIP_PROF_GEMM_FFN1, gemm_tn(xdt, xdt, EPI_BIAS_GELU_SPLIT3, ws.a3h, w.w1, p.b1, nullptr, ws.a3, m_bound, I, 3 * H, ws.m_total, s))
    PROF_STEP(MANNER_HIP_PROF_GEMM_FFN2, gemm_tn(xdt, DT_F32, EPI_BIAS_RES_F32, ws.a3, w.w2, p.b2, ws.x, ws.pre, m_bound, H, 3 * I, ws.m_total, s))
    PROF_STEP(MANNER_HIP_PROF_LAYERNORM, layernorm_rows_split(xdt, ws.pre, p.ln2g, p.ln2b, H, c.ln_eps, xf, ws.a3h, m_bound, ws.m_total, s))
  }
  for (int l = 0; l < full_layers && !x3; ++l) {
    const LayerWeights& w = e->w[prec][l];
    const LayerParams& p = e->params[l];
    PROF_STEP(MANNER_HIP_PROF_GEMM_QKV, gemm(dt, EPI_BIAS, ws.x, w.wqkv, p.bqkv, nullptr, ws.qkv, m_bound, 3 * H, H, ws.m_total))
    PROF_STEP(MANNER_HIP_PROF_ATTENTION, attention_varlen(dt, ws.qkv, ws.ctx, ws.cu, n_news, c.heads, H, (int)lp, s))
    PROF_STEP(MANNER_HIP_PROF_GEMM_OUT, gemm(DT_F32, EPI_BIAS_RES, ws.ctx, w.wo, p.bo, ws.x, ws.pre, m_bound, H, H, ws.m_total))
    PROF_STEP(MANNER_HIP_PROF_LAYERNORM, layernorm_rows(dt, ws.pre, p.ln1g, p.ln1b, H, c.ln_eps, ws.x, m_bound, ws.m_total, s))
    if (l == 0) phase.mark();   // two-stream mode: the other stream starts half a layer later
    PROF_STEP(MANNER_HIP_PROF_GEMM_FFN1, gemm(dt, EPI_BIAS_GELU, ws.x, w.w1, p.b1, nullptr, ws.ffn, m_bound, I, H, ws.m_total))
    PROF_STEP(MANNER_HIP_PROF_GEMM_FFN2, gemm(DT_F32, EPI_BIAS_RES, ws.ffn, w.w2, p.b2, ws.x, ws.pre, m_bound, H, I, ws.m_total))
    PROF_STEP(MANNER_HIP_PROF_LAYERNORM, layernorm_rows(dt, ws.pre, p.ln2g, p.ln2b, H, c.ln_eps, ws.x, m_bound, ws.m_total, s))
  }
  if (hidden_layers >= 0) {
    PROF_STEP(MANNER_HIP_PROF_GATHER, scatter_hidden(dt, hidden_dt, ws.x, nullptr, ws.cu, n_news, lp, H, nullptr, nullptr, out, s))
    return MANNER_HIP_OK;
  }
  // Last layer: the reference keeps only last_hidden_state[:, 0, :] (news_encoder.py:30-34), so K and V
  // are needed for every token but Q, the output projection, both LayerNorms and the FFN only for the
  // [CLS] row of each news.  Those run on compact [n_news, *] buffers; the final LayerNorm writes the
  // f32 result straight into `out`.
  {
    const LayerWeights& w = e->w[prec][c.layers - 1];
    const LayerParams& p = e->params[c.layers - 1];
    const int64_t n_bound = round_up(n_news, 256);
    const int32_t* n_total = ws.m_total + 1;
    const char* wkv = static_cast<const char*>(w.wqkv) + (size_t)H * (x3 ? (size_t)3 * H * 2 : (size_t)H * es);
    if (x3)   // the split of the layer input is already in ws.a3h (written by the previous LayerNorm / the embedding split)
      PROF_STEP(MANNER_HIP_PROF_GEMM_QKV, gemm_tn(xdt, DT_F32, EPI_BIAS, ws.a3h, wkv, p.bqkv + H, nullptr, ws.qkv, m_bound, 2 * H, 3 * H, ws.m_total, s))
    else
      PROF_STEP(MANNER_HIP_PROF_GEMM_QKV, gemm(dt, EPI_BIAS, ws.x, wkv, p.bqkv + H, nullptr, ws.qkv, m_bound, 2 * H, H, ws.m_total))
    char* ax = static_cast<char*>(ws.acc_x) + (size_t)cls_off * H * es;
    char* actx = static_cast<char*>(ws.acc_ctx) + (size_t)cls_off * H * es;
    PROF_STEP(MANNER_HIP_PROF_GATHER, gather_cls_rows(dt, ws.x, ws.cu, n_news, H, ax, s))
    PROF_STEP(MANNER_HIP_PROF_CLS_TAIL, gemm(dt, EPI_BIAS, ax, w.wqkv, p.bqkv, nullptr, ws.qcls, n_bound, H, H, n_total))
    PROF_STEP(MANNER_HIP_PROF_ATTENTION, attention_cls(dt, ws.qcls, ws.qkv, actx, ws.cu, n_news, c.heads, H, s))
  }
  return MANNER_HIP_OK;
}

// Tail of the last layer over the `rows` [CLS] rows collected in ws.acc_* (K4-K6 on one row per news, then the CLS slice
// K7 is the result itself): pre = ctx Wo^T + b + x; x1 = LN(pre); pre = gelu(x1 W1^T + b) W2^T + b + x1; out = LN(pre).
// Scratch: ws.pre / ws.ffn of the (finished) chunk.  `out` receives f32 [rows, H].
int run_cls_tail(manner_hip_encoder* e, int prec, const Workspace& ws, int64_t rows, float* out, hipStream_t s) {
  if (rows <= 0) return MANNER_HIP_OK;
  const manner_hip_encoder_config& c = e->cfg;
  const DType dt = act_dtype(prec);
  const int H = c.hidden, I = c.intermediate;
  const bool x3 = is_x3(prec);
  const DType xdt = x3_dtype(prec);
  const LayerWeights& w = e->w[prec][c.layers - 1];
  const LayerParams& p = e->params[c.layers - 1];
  const int64_t bound = round_up(rows, 256);
  int rc;
  if ((rc = set_device_int(ws.cls_total, (int32_t)rows, s))) return rc;
  auto gemm = [&](DType out_dt, Epilogue epi, const void* A, const void* Wm, const float* bias, const void* res, void* Y, int N, int K) -> int {
    if (!x3) return gemm_tn(dt, out_dt, epi, A, Wm, bias, res, Y, bound, N, K, ws.cls_total, s);
    int r = split3_rows(xdt, false, static_cast<const float*>(A), ws.a3, K, bound, ws.cls_total, s);
    if (r) return r;
    return gemm_tn(xdt, DT_F32, epi == EPI_BIAS_RES ? EPI_BIAS_RES_F32 : epi, ws.a3, Wm, bias, res, Y, bound, N, 3 * K, ws.cls_total, s);
  };
  PROF_STEP(MANNER_HIP_PROF_CLS_TAIL, gemm(DT_F32, EPI_BIAS_RES, ws.acc_ctx, w.wo, p.bo, ws.acc_x, ws.pre, H, H))
  PROF_STEP(MANNER_HIP_PROF_CLS_TAIL, layernorm_rows(dt, ws.pre, p.ln1g, p.ln1b, H, c.ln_eps, ws.acc_x1, bound, ws.cls_total, s))
  PROF_STEP(MANNER_HIP_PROF_CLS_TAIL, gemm(dt, EPI_BIAS_GELU, ws.acc_x1, w.w1, p.b1, nullptr, ws.ffn, I, H))
  PROF_STEP(MANNER_HIP_PROF_CLS_TAIL, gemm(DT_F32, EPI_BIAS_RES, ws.ffn, w.w2, p.b2, ws.acc_x1, ws.pre, H, I))
  PROF_STEP(MANNER_HIP_PROF_CLS_TAIL, layernorm_rows(DT_F32, ws.pre, p.ln2g, p.ln2b, H, c.ln_eps, out, bound, ws.cls_total, s))
  return MANNER_HIP_OK;
}

// word t = order-independent sum of mixed samples of tensor t (see manner_hip_fingerprint in the header)
__global__ void __launch_bounds__(256) fingerprint_kernel(const uint32_t* const* __restrict__ tensors, const int64_t* __restrict__ counts, int samples,
                                                          uint32_t* __restrict__ out) {
  __shared__ uint32_t part[4];
  const int t = blockIdx.x;
  const uint32_t* p = tensors[t];
  const int64_t n = counts[t], m = n < samples ? n : (int64_t)samples;
  uint32_t h = 0;
  for (int64_t j = threadIdx.x; j < m; j += 256) {
    const int64_t idx = (m == n || m < 2) ? j : (j * (n - 1)) / (m - 1);
    uint32_t v = p[idx] ^ ((uint32_t)j * 0x9E3779B9u);
    v *= 0x85EBCA6Bu; v ^= v >> 13; v *= 0xC2B2AE35u; v ^= v >> 16;
    h += v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = h;
  __syncthreads();
  if (threadIdx.x == 0) out[t] = part[0] + part[1] + part[2] + part[3] + (uint32_t)n * 0x27D4EB2Fu;
}

}  // namespace
}  // namespace manner

using namespace manner;

extern "C" {

int manner_hip_abi_version(void) { return MANNER_HIP_ABI_VERSION; }

int manner_hip_fingerprint(const void* const* tensors, const int64_t* counts, int32_t n, int32_t samples, uint32_t* out, manner_hip_stream_t stream) {
  if (n == 0) return MANNER_HIP_OK;
  if (!tensors || !counts || !out || n < 0 || samples <= 0) return fail(MANNER_HIP_E_INVALID, "fingerprint: null table or n=%d samples=%d", n, samples);
  hipLaunchKernelGGL(fingerprint_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint32_t* const*>(tensors), counts,
                     samples, out);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}
const char* manner_hip_last_error(void) { return g_err; }

int manner_hip_encoder_profile(manner_hip_encoder_t enc, int32_t enable) {
  if (!enc) return fail(MANNER_HIP_E_INVALID, "encoder_profile: null handle");
  enc->profiling = enable != 0;
  return MANNER_HIP_OK;
}

int manner_hip_encoder_profile_read(manner_hip_encoder_t enc, manner_hip_stream_t stream, double* ms, int64_t* launches) {
  if (!enc || !ms || !launches) return fail(MANNER_HIP_E_INVALID, "encoder_profile_read: null argument");
  MANNER_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  for (auto& sp : enc->spans) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, sp.a, sp.b) == hipSuccess) { ms[sp.cls] += t; launches[sp.cls] += 1; }
    enc->free_events.push_back(sp.a);
    enc->free_events.push_back(sp.b);
  }
  enc->spans.clear();
  return MANNER_HIP_OK;
}

int manner_hip_encoder_destroy(manner_hip_encoder_t enc) {
  if (!enc) return MANNER_HIP_OK;
  for (auto& sp : enc->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
  for (hipEvent_t ev : enc->free_events) (void)hipEventDestroy(ev);
  for (int i = 0; i < manner_hip_encoder::MAX_STREAMS; ++i) {
    if (enc->side[i]) (void)hipStreamDestroy(enc->side[i]);
    if (enc->join_ev[i]) (void)hipEventDestroy(enc->join_ev[i]);
    if (enc->phase_ev[i]) (void)hipEventDestroy(enc->phase_ev[i]);
  }
  for (void* p : enc->allocs) (void)hipFree(p);
  delete enc;
  return MANNER_HIP_OK;
}

int manner_hip_encoder_create(const manner_hip_encoder_config* cfg, const float* const* weights, int32_t n_weights,
                              uint32_t precisions, manner_hip_stream_t stream, manner_hip_encoder_t* out) {
  if (!cfg || !weights || !out) return fail(MANNER_HIP_E_INVALID, "encoder_create: null argument");
  const int H = cfg->hidden, I = cfg->intermediate, L = cfg->layers;
  if (cfg->arch != MANNER_HIP_ARCH_BERT && cfg->arch != MANNER_HIP_ARCH_ROBERTA) return fail(MANNER_HIP_E_INVALID, "encoder_create: unknown arch %d", cfg->arch);
  if (H <= 0 || H % 128 || H > 1024 || cfg->heads <= 0 || H != cfg->heads * 64) return fail(MANNER_HIP_E_INVALID, "encoder_create: hidden=%d heads=%d (need H %% 128 == 0, H <= 1024, head_dim 64)", H, cfg->heads);
  if (I <= 0 || I % 128 || L <= 0) return fail(MANNER_HIP_E_INVALID, "encoder_create: intermediate=%d layers=%d unsupported", I, L);
  if (cfg->vocab <= 0 || cfg->max_pos <= 0 || cfg->type_vocab <= 0) return fail(MANNER_HIP_E_INVALID, "encoder_create: bad table sizes");
  if (n_weights != MANNER_HIP_W_EMB_COUNT + L * MANNER_HIP_WL_COUNT) return fail(MANNER_HIP_E_INVALID, "encoder_create: expected %d weight pointers, got %d", MANNER_HIP_W_EMB_COUNT + L * MANNER_HIP_WL_COUNT, n_weights);
  if (!(precisions & 31u) || (precisions & ~31u)) return fail(MANNER_HIP_E_INVALID, "encoder_create: precisions mask 0x%x", precisions);
  if ((precisions & ((1u << MANNER_HIP_PREC_BF16X3) | (1u << MANNER_HIP_PREC_F16X3))) && (H % 256 || I % 256)) return fail(MANNER_HIP_E_INVALID, "encoder_create: BF16X3 / F16X3 need hidden and intermediate sizes that are multiples of 256 (H=%d I=%d)", H, I);
  for (int i = 0; i < n_weights; ++i)
    if (!weights[i]) return fail(MANNER_HIP_E_INVALID, "encoder_create: weight pointer %d is null", i);

  hipStream_t s = (hipStream_t)stream;
  manner_hip_encoder* e = new manner_hip_encoder();
  e->cfg = *cfg;
  e->precisions = precisions;
  int rc = MANNER_HIP_OK;
  auto guard = [&](int r) { if (r && !rc) rc = r; return r == 0; };
  const size_t HH = (size_t)H * H, HI = (size_t)H * I;
  do {
    if (!guard(dev_copy_f32(e, weights[MANNER_HIP_W_WORD_EMB], (size_t)cfg->vocab * H, &e->word, s))) break;
    if (!guard(dev_copy_f32(e, weights[MANNER_HIP_W_POS_EMB], (size_t)cfg->max_pos * H, &e->pos, s))) break;
    if (!guard(dev_copy_f32(e, weights[MANNER_HIP_W_TYPE_EMB], (size_t)H, &e->type0, s))) break;   // row 0 only
    if (!guard(dev_copy_f32(e, weights[MANNER_HIP_W_EMB_LN_G], H, &e->embg, s))) break;
    if (!guard(dev_copy_f32(e, weights[MANNER_HIP_W_EMB_LN_B], H, &e->embb, s))) break;
    if (!guard(dev_alloc(e, 256, (void**)&e->status))) break;
    if (hipMemsetAsync(e->status, 0, 256, s) != hipSuccess) { rc = fail(MANNER_HIP_E_RUNTIME, "memset failed"); break; }
    const size_t arrive_bytes = (size_t)manner_hip_encoder::MAX_STREAMS * manner_hip_encoder::ARRIVE_CAP * sizeof(int32_t);
    if (!guard(dev_alloc(e, arrive_bytes, (void**)&e->arrive))) break;
    if (hipMemsetAsync(e->arrive, 0, arrive_bytes, s) != hipSuccess) { rc = fail(MANNER_HIP_E_RUNTIME, "memset failed"); break; }
    if (const char* ev = getenv("MANNER_HIP_STREAMS")) {
      const int v = atoi(ev);
      e->n_streams = v < 1 ? 1 : (v > manner_hip_encoder::MAX_STREAMS ? manner_hip_encoder::MAX_STREAMS : v);
    }
    // deferred LayerNorm needs every full-size GEMM on the 256x256 kernel: H, 3H, 2H and I multiples of 256
    e->defer_ln = (precisions & ((1u << MANNER_HIP_PREC_BF16) | (1u << MANNER_HIP_PREC_F16))) && H % 256 == 0 && I % 256 == 0;
    if (const char* ev = getenv("MANNER_HIP_DEFER_LN")) e->defer_ln = e->defer_ln && atoi(ev) != 0;
    for (int i = 0; i < e->n_streams && !rc; ++i) {
      if ((i > 0 && hipStreamCreateWithFlags(&e->side[i], hipStreamNonBlocking) != hipSuccess) ||
          hipEventCreateWithFlags(&e->join_ev[i], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&e->phase_ev[i], hipEventDisableTiming) != hipSuccess)
        rc = fail(MANNER_HIP_E_RUNTIME, "encoder_create: side stream/event creation failed");
    }
    if (rc) break;
    e->params.resize(L);
    for (int p = 0; p < 5; ++p) if (precisions & (1u << p)) e->w[p].resize(L);
    for (int l = 0; l < L && !rc; ++l) {
      const float* const* wl = weights + MANNER_HIP_W_EMB_COUNT + l * MANNER_HIP_WL_COUNT;
      LayerParams& P = e->params[l];
      if (!guard(dev_alloc(e, 3 * (size_t)H * 4, (void**)&P.bqkv))) break;
      for (int j = 0; j < 3; ++j)
        if (hipMemcpyAsync(P.bqkv + j * H, wl[MANNER_HIP_WL_Q_B + 2 * j], H * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) rc = fail(MANNER_HIP_E_RUNTIME, "bias copy failed");
      guard(dev_copy_f32(e, wl[MANNER_HIP_WL_AO_B], H, &P.bo, s));
      guard(dev_copy_f32(e, wl[MANNER_HIP_WL_ALN_G], H, &P.ln1g, s));
      guard(dev_copy_f32(e, wl[MANNER_HIP_WL_ALN_B], H, &P.ln1b, s));
      guard(dev_copy_f32(e, wl[MANNER_HIP_WL_FF1_B], I, &P.b1, s));
      guard(dev_copy_f32(e, wl[MANNER_HIP_WL_FF2_B], H, &P.b2, s));
      guard(dev_copy_f32(e, wl[MANNER_HIP_WL_OLN_G], H, &P.ln2g, s));
      guard(dev_copy_f32(e, wl[MANNER_HIP_WL_OLN_B], H, &P.ln2b, s));
      for (int p = 0; p < 4 && !rc; ++p) {
        if (is_x3(p) || !(precisions & (1u << p))) continue;
        const size_t es = act_bytes(p);
        LayerWeights& W = e->w[p][l];
        if (!guard(dev_alloc(e, 3 * HH * es, &W.wqkv))) break;
        if (!guard(dev_alloc(e, HH * es, &W.wo))) break;
        if (!guard(dev_alloc(e, HI * es, &W.w1))) break;
        if (!guard(dev_alloc(e, HI * es, &W.w2))) break;
        for (int j = 0; j < 3; ++j) guard(pack_matrix(p, wl[MANNER_HIP_WL_Q_W + 2 * j], HH, W.wqkv, j * HH, s));
        guard(pack_matrix(p, wl[MANNER_HIP_WL_AO_W], HH, W.wo, 0, s));
        guard(pack_matrix(p, wl[MANNER_HIP_WL_FF1_W], HI, W.w1, 0, s));
        guard(pack_matrix(p, wl[MANNER_HIP_WL_FF2_W], HI, W.w2, 0, s));
      }
      for (int p : {MANNER_HIP_PREC_BF16X3, MANNER_HIP_PREC_F16X3}) {     // [out, 3 in] split weights of the 16-bit type
        if (!(precisions & (1u << p)) || rc) continue;
        const DType sdt = x3_dtype(p);
        LayerWeights& W = e->w[p][l];
        if (!guard(dev_alloc(e, 3 * HH * 6, &W.wqkv)) || !guard(dev_alloc(e, HH * 6, &W.wo)) ||
            !guard(dev_alloc(e, HI * 6, &W.w1)) || !guard(dev_alloc(e, HI * 6, &W.w2))) break;
        for (int j = 0; j < 3; ++j)
          guard(split3_rows(sdt, true, wl[MANNER_HIP_WL_Q_W + 2 * j], static_cast<char*>(W.wqkv) + (size_t)j * HH * 6, H, H, nullptr, s));
        guard(split3_rows(sdt, true, wl[MANNER_HIP_WL_AO_W], W.wo, H, H, nullptr, s));
        guard(split3_rows(sdt, true, wl[MANNER_HIP_WL_FF1_W], W.w1, H, I, nullptr, s));
        guard(split3_rows(sdt, true, wl[MANNER_HIP_WL_FF2_W], W.w2, I, H, nullptr, s));
      }
      if (e->defer_ln && !rc) {
        // fold the LayerNorm that feeds each GEMM into its weight: the embedding LayerNorm (layer 0) or the
        // previous layer's output LayerNorm for Q|K|V, this layer's attention-output LayerNorm for FFN1
        const float* g_in = l == 0 ? e->embg : e->params[l - 1].ln2g;
        const float* b_in = l == 0 ? e->embb : e->params[l - 1].ln2b;
        if (!guard(dev_alloc(e, 3 * (size_t)H * 4, (void**)&P.cq2)) || !guard(dev_alloc(e, (size_t)I * 4, (void**)&P.cf2))) break;
        if (!guard(dev_alloc(e, (size_t)H * 4, (void**)&P.bo_res)) || !guard(dev_alloc(e, (size_t)H * 4, (void**)&P.b2_res))) break;
        for (int p : {MANNER_HIP_PREC_BF16, MANNER_HIP_PREC_F16}) {
          if (!(precisions & (1u << p)) || rc) continue;
          const DType fdt = act_dtype(p);
          LayerWeights& W = e->w[p][l];
          if (!guard(dev_alloc(e, 3 * HH * 2, &W.wqkv_f)) || !guard(dev_alloc(e, HI * 2, &W.w1_f))) break;
          if (!guard(dev_alloc(e, 3 * (size_t)H * 4, (void**)&W.cq1)) || !guard(dev_alloc(e, (size_t)I * 4, (void**)&W.cf1))) break;
          for (int j = 0; j < 3; ++j)
            guard(fold_layernorm(fdt, wl[MANNER_HIP_WL_Q_W + 2 * j], wl[MANNER_HIP_WL_Q_B + 2 * j], g_in, b_in, H, H,
                                 static_cast<char*>(W.wqkv_f) + j * HH * 2, W.cq1 + j * H, P.cq2 + j * H, s));
          guard(fold_layernorm(fdt, wl[MANNER_HIP_WL_FF1_W], wl[MANNER_HIP_WL_FF1_B], P.ln1g, P.ln1b, I, H, W.w1_f, W.cf1, P.cf2, s));
        }
        guard(add_vectors(P.bo, b_in, P.bo_res, H, s));
        guard(add_vectors(P.b2, P.ln1b, P.b2_res, H, s));
      }
    }
    if (rc) break;
    if (hipStreamSynchronize(s) != hipSuccess) { rc = fail(MANNER_HIP_E_RUNTIME, "encoder_create: stream sync failed: %s", hipGetErrorString(hipGetLastError())); break; }
  } while (0);
  if (rc) { manner_hip_encoder_destroy(e); return rc; }
  *out = e;
  return MANNER_HIP_OK;
}

size_t manner_hip_encoder_workspace_bytes(manner_hip_encoder_t enc, int64_t max_news, int64_t max_tokens, int32_t precision) {
  if (!enc || max_news <= 0 || max_tokens <= 0 || precision < 0 || precision > 4) return 0;
  return enc->n_streams * carve(enc, max_news, round_up(max_tokens, 256), precision, nullptr, nullptr);
}

// shared driver of encode_cls / encode_hidden: chunking, stream fork/join
static int encode_impl(manner_hip_encoder_t enc, const int64_t* ids, const int64_t* mask, const int32_t* host_lengths,
                       int64_t n_news, int64_t padded_len, int32_t precision, void* out, void* workspace,
                       size_t workspace_bytes, manner_hip_stream_t stream, int hidden_layers, DType hidden_dt) {
  if (!enc) return fail(MANNER_HIP_E_INVALID, "encode_cls: null handle");
  if (n_news == 0) return MANNER_HIP_OK;
  if (!ids || !mask || !out || !workspace || n_news < 0) return fail(MANNER_HIP_E_INVALID, "encode_cls: null pointer");
  if (precision < 0 || precision > 4 || !(enc->precisions & (1u << precision))) return fail(MANNER_HIP_E_INVALID, "encode_cls: precision %d was not requested at encoder_create", precision);
  if (padded_len < 1 || padded_len > MANNER_HIP_MAX_LEN) return fail(MANNER_HIP_E_INVALID, "encode_cls: padded_len %lld outside [1, %d]", (long long)padded_len, MANNER_HIP_MAX_LEN);
  if ((uintptr_t)workspace % 256) return fail(MANNER_HIP_E_INVALID, "encode_cls: workspace must be 256-byte aligned");
  const int H = enc->cfg.hidden;
  // largest (news, tokens) chunk capacity the workspace admits: tokens scale the big buffers.
  // With two streams the workspace is split into two independent halves.
  const int ns = enc->profiling ? 1 : enc->n_streams;   // per-kernel timing wants un-overlapped launches
  const size_t ws_each = (workspace_bytes / enc->n_streams) / 256 * 256;   // same chunk size with or without profiling
  const size_t es = act_bytes(precision);
  const size_t per_tok = (size_t)H * 4 + ((size_t)5 * H + enc->cfg.intermediate) * es +
                         (is_x3(precision) ? (size_t)6 * (H > enc->cfg.intermediate ? H : enc->cfg.intermediate) + (size_t)6 * H : 0);
  int64_t m_cap = (int64_t)(ws_each / per_tok) / 256 * 256;
  int64_t n_cap = 0;
  while (m_cap >= 256) {
    n_cap = m_cap;   // a news has >= 1 token, so a chunk never holds more news than tokens
    if (n_cap > n_news) n_cap = n_news;
    if (carve(enc, n_cap, m_cap, precision, nullptr, nullptr) <= ws_each) break;
    m_cap -= 256;
  }
  if (m_cap < 256 || m_cap < padded_len) return fail(MANNER_HIP_E_WORKSPACE, "encode_cls: workspace of %zu bytes cannot hold one 256-token tile per stream", workspace_bytes);
  Workspace ws[manner_hip_encoder::MAX_STREAMS];
  for (int i = 0; i < ns; ++i) {
    carve(enc, n_cap, m_cap, precision, static_cast<char*>(workspace) + i * ws_each, &ws[i]);
    ws[i].arrive = enc->arrive + (size_t)i * manner_hip_encoder::ARRIVE_CAP;
  }
  hipStream_t s0 = (hipStream_t)stream;
  int forked = 0;                                       // side streams in use so far
  // join: the caller's stream waits for every side stream — on EVERY exit path, so that work left on a side stream
  // by a failing call is still ordered before whatever the caller does next with the workspace
  auto join = [&]() -> int {
    for (int i = 1; i <= forked; ++i) {
      MANNER_HIP_TRY(hipEventRecord(enc->join_ev[i], enc->side[i]));
      MANNER_HIP_TRY(hipStreamWaitEvent(s0, enc->join_ev[i], 0));
    }
    forked = 0;
    return MANNER_HIP_OK;
  };
  // ---- plan: chunk boundaries (host arithmetic only), then CONTIGUOUS runs of chunks per stream, so that the [CLS] rows
  // a stream collects for its batched last-layer tail are consecutive rows of `out`
  struct Chunk { int64_t n0, cnt, m_bound, expect; };
  std::vector<Chunk> plan;
  for (int64_t n0 = 0; n0 < n_news;) {
    int64_t cnt = 0, m_bound, expect = -1;
    if (host_lengths) {
      int64_t tok = 0;
      while (n0 + cnt < n_news && cnt < n_cap) {
        const int64_t len = host_lengths[n0 + cnt];
        if (len < 1 || len > padded_len)
          return fail(MANNER_HIP_E_INVALID, "encode_cls: host_lengths[%lld]=%lld outside [1, padded_len]", (long long)(n0 + cnt), (long long)len);
        if (tok + len > m_cap) break;
        tok += len;
        ++cnt;
      }
      m_bound = round_up(tok, 256);
      expect = tok;
    } else {
      cnt = m_cap / padded_len;
      if (cnt > n_cap) cnt = n_cap;
      if (cnt > n_news - n0) cnt = n_news - n0;
      m_bound = round_up(cnt * padded_len, 256);
    }
    plan.push_back({n0, cnt, m_bound, expect});
    n0 += cnt;
  }
  const int n_chunks = (int)plan.size();
  const int lanes = n_chunks < ns ? n_chunks : ns;
  int first[manner_hip_encoder::MAX_STREAMS + 1];                 // lane l runs chunks [first[l], first[l+1])
  for (int l = 0; l <= lanes; ++l) first[l] = (int)((int64_t)n_chunks * l / lanes);
  int next[manner_hip_encoder::MAX_STREAMS];
  int64_t acc_rows[manner_hip_encoder::MAX_STREAMS] = {}, acc_first[manner_hip_encoder::MAX_STREAMS] = {};
  for (int l = 0; l < lanes; ++l) next[l] = first[l];
  const bool cls = hidden_layers < 0;
  const size_t news_bytes = cls ? (size_t)H * 4 : (size_t)padded_len * H * (hidden_dt == DT_F32 ? 4 : 2);
  auto flush = [&](int lane, hipStream_t s) -> int {              // the batched tail over the rows this lane has collected
    if (!cls || acc_rows[lane] == 0) return MANNER_HIP_OK;
    int rc = run_cls_tail(enc, precision, ws[lane], acc_rows[lane], static_cast<float*>(out) + acc_first[lane] * H, s);
    acc_rows[lane] = 0;
    return rc;
  };
  // chunks are enqueued round-robin over the lanes so that every stream always has work queued
  for (int left = n_chunks; left > 0;) {
    for (int lane = 0; lane < lanes; ++lane) {
      if (next[lane] >= first[lane + 1]) continue;
      const Chunk& ck = plan[next[lane]];
      const bool lane_first = next[lane] == first[lane];
      hipStream_t s = lane == 0 ? s0 : enc->side[lane];
      if (lane > 0 && lane_first) {
        // stream `lane` starts when the first chunk of stream lane-1 is half a layer in (its phase mark): from
        // then on the streams run out of phase, so the HBM-bound kernels of one (attention, GEMM epilogues) meet
        // the MFMA-bound main loops of another
        if (hipStreamWaitEvent(s, enc->phase_ev[lane - 1], 0) != hipSuccess) {
          (void)join();
          return fail(MANNER_HIP_E_RUNTIME, "encode_cls: hipStreamWaitEvent failed");
        }
        forked = lane;
      }
      int rc = MANNER_HIP_OK;
      if (cls && acc_rows[lane] + ck.cnt > ws[lane].cls_cap) rc = flush(lane, s);
      if (cls && acc_rows[lane] == 0) acc_first[lane] = ck.n0;
      if (!rc)
        rc = encode_chunk(enc, ids + ck.n0 * padded_len, mask + ck.n0 * padded_len, ck.cnt, padded_len, ck.m_bound, ck.expect,
                          precision, static_cast<char*>(out) + ck.n0 * news_bytes, ws[lane], s,
                          (lane_first && lane + 1 < lanes) ? enc->phase_ev[lane] : nullptr, hidden_layers, hidden_dt,
                          acc_rows[lane]);
      if (rc) { (void)join(); return rc; }
      if (cls) acc_rows[lane] += ck.cnt;
      ++next[lane];
      --left;
      if (next[lane] == first[lane + 1] && (rc = flush(lane, s))) { (void)join(); return rc; }
    }
  }
  return join();
}

int manner_hip_encode_cls(manner_hip_encoder_t enc, const int64_t* ids, const int64_t* mask, const int32_t* host_lengths,
                          int64_t n_news, int64_t padded_len, int32_t precision, float* out, void* workspace,
                          size_t workspace_bytes, manner_hip_stream_t stream) {
  return encode_impl(enc, ids, mask, host_lengths, n_news, padded_len, precision, out, workspace, workspace_bytes, stream, -1, DT_F32);
}

int manner_hip_encode_hidden(manner_hip_encoder_t enc, const int64_t* ids, const int64_t* mask, const int32_t* host_lengths,
                             int64_t n_news, int64_t padded_len, int32_t precision, int32_t n_layers, int32_t out_dtype,
                             void* out, void* workspace, size_t workspace_bytes, manner_hip_stream_t stream) {
  if (!enc) return fail(MANNER_HIP_E_INVALID, "encode_hidden: null handle");
  if (n_layers < 0 || n_layers > enc->cfg.layers) return fail(MANNER_HIP_E_INVALID, "encode_hidden: n_layers %d outside [0, %d]", n_layers, enc->cfg.layers);
  if (out_dtype != 0 && out_dtype != 1) return fail(MANNER_HIP_E_INVALID, "encode_hidden: out_dtype %d (0 = f32, 1 = bf16)", out_dtype);
  return encode_impl(enc, ids, mask, host_lengths, n_news, padded_len, precision, out, workspace, workspace_bytes, stream, n_layers,
                     out_dtype == 0 ? DT_F32 : DT_BF16);
}

int manner_hip_encoder_status(manner_hip_encoder_t enc, manner_hip_stream_t stream) {
  if (!enc) return fail(MANNER_HIP_E_INVALID, "encoder_status: null handle");
  int32_t flag = 0;
  hipStream_t s = (hipStream_t)stream;
  MANNER_HIP_TRY(hipMemcpyAsync(&flag, enc->status, sizeof(flag), hipMemcpyDeviceToHost, s));
  MANNER_HIP_TRY(hipMemsetAsync(enc->status, 0, sizeof(flag), s));
  MANNER_HIP_TRY(hipStreamSynchronize(s));
  if (flag & MANNER_HIP_STATUS_MASK) return fail(MANNER_HIP_E_INPUT, "attention_mask is not a right-padded 0/1 prefix mask with 1..%d real tokens per news", MANNER_HIP_MAX_LEN);
  if (flag & MANNER_HIP_STATUS_TOKEN) return fail(MANNER_HIP_E_INPUT, "input_ids or position index out of range of the embedding tables");
  if (flag & MANNER_HIP_STATUS_LENGTHS) return fail(MANNER_HIP_E_INPUT, "host_lengths disagree with the row sums of attention_mask (tokens beyond the chunk bound were dropped)");
  if (flag) return fail(MANNER_HIP_E_INPUT, "device status word 0x%x", flag);
  return MANNER_HIP_OK;
}

int manner_hip_encoder_status_async(manner_hip_encoder_t enc, int32_t* host_flag, manner_hip_stream_t stream) {
  if (!enc || !host_flag) return fail(MANNER_HIP_E_INVALID, "encoder_status_async: null argument");
  hipStream_t s = (hipStream_t)stream;
  MANNER_HIP_TRY(hipMemcpyAsync(host_flag, enc->status, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  MANNER_HIP_TRY(hipMemsetAsync(enc->status, 0, sizeof(int32_t), s));
  return MANNER_HIP_OK;
}

}  // extern "C"
