// Content-addressed news-embedding cache (round 4): SURVEY.md §8(d) "mode T" — each unique news encoded once — behind the UNCHANGED
// drop-in call pattern.  The reference encodes every history and candidate occurrence of every impression again
// (manner/models/cr_module.py:107,113 -> news_encoder.py:29-37); a MIND dev set lists its 65 k news 4.2 M times.  In eval() under
// no_grad the text encoder is a pure function of (weights, real tokens of the row), row by row, so `MannerTextEncoder.forward` may look
// a row up by its tokens and encode only the rows it has not seen under the current weights.  Opt-in (MANNER_EMBED_CACHE_ROWS /
// MannerTextEncoder.embedding_cache_rows); bench.py's headline (mode R) and its `B8_eval` drop-in figure never use it.
//
//   manner_hip_news_key128        128-bit key of every row's REAL tokens (mask == 1 positions, any padded width)
//   manner_hip_news_cache_lookup  open-addressing table in caller-owned device memory: row -> {table row, state}
//
// HBM-trivial integer work (a batch is a few thousand rows of <= 512 tokens): one wave per row for the keys — coalesced 8-byte loads,
// a position-keyed 64-bit mix per token, two independent sums reduced over the wave — and one thread per row for the table, in three
// launches so that nobody reads a slot another thread of the same call is still filling.
#include "common.h"

namespace manner {
namespace {

__device__ __forceinline__ uint64_t mix64(uint64_t z) {          // splitmix64 finaliser: a bijection with full avalanche
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// key = (sum_j f0(j, id_j), sum_j f1(j, id_j)) over the real positions j, + a term of the token count: a sum of per-position random
// functions is a universal hash of the sequence (two rows collide only if the 64-bit sums do), position-keyed so that order matters,
// padding-width independent because padded positions add nothing.
__global__ __launch_bounds__(256) void news_key_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ mask, int64_t n_news,
                                                       int64_t padded_len, uint64_t* __restrict__ keys) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * 4 + wave;
  if (n >= n_news) return;
  const int64_t* ir = ids + n * padded_len;
  const int64_t* mr = mask + n * padded_len;
  uint64_t a = 0, b = 0, cnt = 0;
  for (int64_t j = lane; j < padded_len; j += 64) {
    if (mr[j] != 0) {
      const uint64_t t = (uint64_t)ir[j];
      a += mix64(t * 0x9e3779b97f4a7c15ull + (uint64_t)(j + 1) * 0xd1b54a32d192ed03ull);
      b += mix64((t + 0x632be59bd9b4e019ull) * 0xe7037ed1a0b428dbull ^ (uint64_t)(j + 1) * 0x8ebc6af09c88c6e3ull);
      ++cnt;
    }
  }
  a = wave_sum_u64(a); b = wave_sum_u64(b); cnt = wave_sum_u64(cnt);
  if (lane == 0) {
    a = mix64(a + cnt * 0xa0761d6478bd642full);
    b = mix64(b ^ (cnt + 1) * 0xe7037ed1a0b428dbull);
    keys[2 * n] = a ? a : 1;                         // 0 marks an empty slot
    keys[2 * n + 1] = b;
  }
}

// pass 1: find the slot holding k0 or claim an empty one.  scratch[n] = slot (-1: table full), scratch[n_news + n] = 1 when this
// thread's compare-and-swap put the key there (exactly one thread per new key, however often the key occurs in the call).
__global__ __launch_bounds__(256) void cache_claim_kernel(const uint64_t* __restrict__ keys, int64_t n_news, unsigned long long* slot_k0,
                                                          int64_t n_slots, int32_t* __restrict__ scratch) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= n_news) return;
  const unsigned long long k0 = keys[2 * n];
  const uint64_t maskb = (uint64_t)n_slots - 1;
  uint64_t i = mix64(k0) & maskb;
  int32_t slot = -1, won = 0;
  // bounded probing: insertion and search give up at the same distance, so a key is either within it or not in the table; a table
  // that has seen far more distinct keys than it has slots degrades to "encode, do not store" instead of to a long walk
  const int64_t max_probe = n_slots < 1024 ? n_slots : 1024;
  for (int64_t probe = 0; probe < max_probe; ++probe, i = (i + 1) & maskb) {
    unsigned long long cur = __hip_atomic_load(&slot_k0[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == 0) {
      cur = atomicCAS(&slot_k0[i], 0ull, k0);
      if (cur == 0) { slot = (int32_t)i; won = 1; break; }
    }
    if (cur == k0) { slot = (int32_t)i; break; }
  }
  scratch[n] = slot;
  scratch[n_news + n] = won;
}

// pass 2: the claimants complete their slots: second key word and a table row (-1 when the table is full: the key stays known and
// its rows are encoded on every call, never stored)
__global__ __launch_bounds__(256) void cache_fill_kernel(const uint64_t* __restrict__ keys, int64_t n_news, uint64_t* __restrict__ slot_k1,
                                                         int32_t* __restrict__ slot_row, int32_t* row_count, int32_t capacity_rows,
                                                         const int32_t* __restrict__ scratch) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= n_news || !scratch[n_news + n]) return;
  const int32_t slot = scratch[n];
  slot_k1[slot] = keys[2 * n + 1];
  const int32_t r = atomicAdd(row_count, 1);
  slot_row[slot] = r < capacity_rows ? r : -1;
}

// pass 3: state 0 = the row's embedding is (or, for a duplicate of a key new in this call, will be) in the table; 1 = new: encode and
// store at rows_out; 2 = encode, do not store (table or slots full, or — 2^-64 — another key with the same first word owns the slot)
__global__ __launch_bounds__(256) void cache_resolve_kernel(const uint64_t* __restrict__ keys, int64_t n_news, const uint64_t* __restrict__ slot_k1,
                                                            const int32_t* __restrict__ slot_row, const int32_t* __restrict__ scratch,
                                                            int32_t* __restrict__ rows_out, int32_t* __restrict__ state_out) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= n_news) return;
  const int32_t slot = scratch[n];
  int32_t row = -1, state = 2;
  if (slot >= 0 && slot_k1[slot] == keys[2 * n + 1]) {
    row = slot_row[slot];
    state = row < 0 ? 2 : (scratch[n_news + n] ? 1 : 0);
    if (row < 0) row = -1;
  }
  rows_out[n] = row;
  state_out[n] = state;
}

}  // namespace
}  // namespace manner

using namespace manner;

extern "C" int manner_hip_news_key128(const int64_t* ids, const int64_t* mask, int64_t n_news, int64_t padded_len, uint64_t* keys,
                                      manner_hip_stream_t stream) {
  if (n_news < 0 || padded_len < 0) return fail(MANNER_HIP_E_INVALID, "news_key128: n_news=%lld padded_len=%lld", (long long)n_news, (long long)padded_len);
  if (n_news == 0) return MANNER_HIP_OK;
  if (!ids || !mask || !keys) return fail(MANNER_HIP_E_INVALID, "news_key128: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(news_key_kernel, dim3((unsigned)((n_news + 3) / 4)), dim3(256), 0, s, ids, mask, n_news, padded_len, keys);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

extern "C" int manner_hip_news_cache_lookup(const uint64_t* keys, int64_t n_news, uint64_t* slot_keys, int32_t* slot_rows, int64_t n_slots,
                                            int32_t* row_count, int32_t capacity_rows, int32_t* rows_out, int32_t* state_out,
                                            int32_t* scratch, manner_hip_stream_t stream) {
  if (n_news < 0 || n_slots < 2 || (n_slots & (n_slots - 1)) || n_slots > 0x40000000 || capacity_rows < 0)
    return fail(MANNER_HIP_E_INVALID, "news_cache_lookup: n_news=%lld n_slots=%lld (a power of two <= 2^30) capacity=%d", (long long)n_news,
                (long long)n_slots, capacity_rows);
  if (n_news == 0) return MANNER_HIP_OK;
  if (!keys || !slot_keys || !slot_rows || !row_count || !rows_out || !state_out || !scratch)
    return fail(MANNER_HIP_E_INVALID, "news_cache_lookup: null pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 g((unsigned)((n_news + 255) / 256)), b(256);
  unsigned long long* k0 = reinterpret_cast<unsigned long long*>(slot_keys);
  uint64_t* k1 = slot_keys + n_slots;
  hipLaunchKernelGGL(cache_claim_kernel, g, b, 0, s, keys, n_news, k0, n_slots, scratch);
  MANNER_LAUNCH_CHECK();
  hipLaunchKernelGGL(cache_fill_kernel, g, b, 0, s, keys, n_news, k1, slot_rows, row_count, capacity_rows, scratch);
  MANNER_LAUNCH_CHECK();
  hipLaunchKernelGGL(cache_resolve_kernel, g, b, 0, s, keys, n_news, k1, slot_rows, scratch, rows_out, state_out);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}
