// Device-side collate: builds the tensors of a MINDRecBatch from a device-resident,
// pre-tokenised news store and the row indices of the batch's history / candidate news — what
// MINDCollate.__call__ (manner/data/components/mind_rec_dataset.py:114-137) assembles on the host with
// pd.concat + the tokenizer on every step.  Pure byte/index movement, HBM-bound: coalesced 16-byte stores of
// the int64 outputs, one wave per output row.
#include "common.h"

namespace manner {
namespace {

// _make_batch_assignees (mind_rec_dataset.py:171-174): seg[j] = i for off[i] <= j < off[i+1]
__global__ __launch_bounds__(256) void segments_kernel(const int64_t* __restrict__ off, int64_t B, int64_t total,
                                                      int64_t* __restrict__ seg) {
  for (int64_t j = blockIdx.x * 256ll + threadIdx.x; j < total; j += 256ll * gridDim.x) {
    int64_t lo = 0, hi = B;                       // largest i with off[i] <= j (empty segments are skipped)
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (off[mid] <= j) lo = mid; else hi = mid;
    }
    seg[j] = lo;
  }
}

typedef long long i64x2 __attribute__((ext_vector_type(2)));

// tokenizer(..., padding=True) restated on stored token rows (mind_rec_dataset.py:134-137): row r of the
// batch = the first min(len, Lp) stored ids of news rows[r], then pad_id; mask 1 / 0.
__global__ __launch_bounds__(256) void text_kernel(const int32_t* __restrict__ store_ids, const int32_t* __restrict__ store_len,
                                                  int64_t n_news, int Ls, const int32_t* __restrict__ rows, int64_t M,
                                                  int Lp, int pad_id, int64_t* __restrict__ ids, int64_t* __restrict__ mask) {
  const int lane = threadIdx.x & 63;
  const int64_t r = blockIdx.x * 4ll + (threadIdx.x >> 6);
  if (r >= M) return;
  int64_t src = rows[r];
  src = src < 0 ? 0 : (src >= n_news ? n_news - 1 : src);
  int len = store_len[src];
  len = len < 0 ? 0 : (len > Ls ? Ls : len);
  const int32_t* in = store_ids + src * Ls;
  int64_t* oi = ids + r * Lp;
  int64_t* om = mask + r * Lp;
  if ((Lp & 1) == 0) {                            // rows are 16-byte aligned: two int64 per lane per store
    for (int c = 2 * lane; c < Lp; c += 128) {
      i64x2 v, m;
      v[0] = c < len ? in[c] : pad_id;         m[0] = c < len;
      v[1] = c + 1 < len ? in[c + 1] : pad_id; m[1] = c + 1 < len;
      *reinterpret_cast<i64x2*>(oi + c) = v;
      *reinterpret_cast<i64x2*>(om + c) = m;
    }
  } else {
    for (int c = lane; c < Lp; c += 64) {
      oi[c] = c < len ? in[c] : pad_id;
      om[c] = c < len;
    }
  }
}

// _tokenize_entities (mind_rec_dataset.py:139-144): entity index lists right-padded with 0 to the batch max
__global__ __launch_bounds__(256) void entities_kernel(const int32_t* __restrict__ store_ent, const int32_t* __restrict__ store_cnt,
                                                      int64_t n_news, int Es, const int32_t* __restrict__ rows, int64_t M, int E,
                                                      int64_t* __restrict__ out) {
  const int64_t total = M * (int64_t)E;
  for (int64_t j = blockIdx.x * 256ll + threadIdx.x; j < total; j += 256ll * gridDim.x) {
    const int64_t r = j / E;
    const int c = (int)(j - r * E);
    int64_t src = rows[r];
    src = src < 0 ? 0 : (src >= n_news ? n_news - 1 : src);
    int cnt = store_cnt[src];
    cnt = cnt > Es ? Es : cnt;
    out[j] = c < cnt ? store_ent[src * Es + c] : 0;
  }
}

// category / sentiment labels and the sentiment score of every batch row (mind_rec_dataset.py:164-168)
__global__ __launch_bounds__(256) void aspects_kernel(const int32_t* __restrict__ cat, const int32_t* __restrict__ sent,
                                                     const float* __restrict__ score, int64_t n_news,
                                                     const int32_t* __restrict__ rows, int64_t M, int64_t* __restrict__ ocat,
                                                     int64_t* __restrict__ osent, float* __restrict__ oscore) {
  for (int64_t r = blockIdx.x * 256ll + threadIdx.x; r < M; r += 256ll * gridDim.x) {
    int64_t src = rows[r];
    src = src < 0 ? 0 : (src >= n_news ? n_news - 1 : src);
    if (ocat) ocat[r] = cat[src];
    if (osent) osent[r] = sent[src];
    if (oscore) oscore[r] = score[src];
  }
}

unsigned grid_for(int64_t n) { return (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096); }

}  // namespace
}  // namespace manner

using namespace manner;

extern "C" {

int manner_hip_collate_segments(const int64_t* off, int64_t B, int64_t total, int64_t* seg, manner_hip_stream_t stream) {
  if (B < 0 || total < 0 || (total && (!off || !seg || B == 0))) return fail(MANNER_HIP_E_INVALID, "collate_segments: bad argument");
  if (total == 0) return MANNER_HIP_OK;
  hipLaunchKernelGGL(segments_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, off, B, total, seg);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int manner_hip_collate_text(const int32_t* store_ids, const int32_t* store_len, int64_t n_news, int32_t Ls,
                            const int32_t* rows, int64_t M, int32_t Lp, int32_t pad_id, int64_t* ids, int64_t* mask,
                            manner_hip_stream_t stream) {
  if (M == 0 || Lp == 0) return MANNER_HIP_OK;
  if (!store_ids || !store_len || !rows || !ids || !mask || n_news <= 0 || Ls <= 0 || Lp < 0 || M < 0)
    return fail(MANNER_HIP_E_INVALID, "collate_text: bad argument");
  hipLaunchKernelGGL(text_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, store_ids, store_len, n_news,
                     (int)Ls, rows, M, (int)Lp, (int)pad_id, ids, mask);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int manner_hip_collate_entities(const int32_t* store_ent, const int32_t* store_cnt, int64_t n_news, int32_t Es,
                                const int32_t* rows, int64_t M, int32_t E, int64_t* out, manner_hip_stream_t stream) {
  if (M == 0 || E == 0) return MANNER_HIP_OK;
  if (!store_ent || !store_cnt || !rows || !out || n_news <= 0 || Es <= 0 || E < 0 || M < 0)
    return fail(MANNER_HIP_E_INVALID, "collate_entities: bad argument");
  hipLaunchKernelGGL(entities_kernel, dim3(grid_for(M * (int64_t)E)), dim3(256), 0, (hipStream_t)stream, store_ent, store_cnt,
                     n_news, (int)Es, rows, M, (int)E, out);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int manner_hip_collate_aspects(const int32_t* category, const int32_t* sentiment, const float* sentiment_score, int64_t n_news,
                               const int32_t* rows, int64_t M, int64_t* out_category, int64_t* out_sentiment,
                               float* out_score, manner_hip_stream_t stream) {
  if (M == 0) return MANNER_HIP_OK;
  if (!rows || n_news <= 0 || M < 0 || (out_category && !category) || (out_sentiment && !sentiment) ||
      (out_score && !sentiment_score))
    return fail(MANNER_HIP_E_INVALID, "collate_aspects: bad argument");
  hipLaunchKernelGGL(aspects_kernel, dim3(grid_for(M)), dim3(256), 0, (hipStream_t)stream, category, sentiment, sentiment_score,
                     n_news, rows, M, out_category, out_sentiment, out_score);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

}  // extern "C"
