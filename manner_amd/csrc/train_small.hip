// SURVEY §8f-3, the small operators of the training step: backward passes of nn.Linear, AdditiveAttention
// (manner/models/components/attention.py:12-29), the axis-0 multi-head attention of the entity branch
// (news_encoder.py:60-72, quirk Q1), nn.Embedding with padding_idx, and dropout on a flat tensor.  The tensors are small
// (entity dim 100, query dim 200, a few thousand rows): plain f32 VALU kernels, each one the textbook derivative of the
// forward kernel in entity.hip / scoring.hip.  Composed into autograd Functions by manner_amd/train.py.
#include <math.h>

#include "common.h"

namespace manner {
namespace {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// same generator as train.hip (splitmix64 of (seed, site, index)); kept when bits >= thr
__device__ __forceinline__ uint32_t drop_bits(uint64_t seed, uint32_t site, uint64_t idx) {
  uint64_t z = (seed ^ ((uint64_t)site * 0xD6E8FEB86659FD93ull)) + (idx + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 32);
}
__global__ __launch_bounds__(256) void dropout_flat_kernel(const float* x, float* out, int64_t n, uint64_t seed, uint32_t site,
                                                           uint32_t thr, float scale) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = (thr == 0 || drop_bits(seed, site, (uint64_t)i) >= thr) ? x[i] * scale : 0.f;
}

// ---------------------------------------------------------------- nn.Linear backward: y = x W^T + b, x [R, K], W [O, K]
constexpr int LB_ROWS = 8;
// dx[r, k] = sum_o dy[r, o] W[o, k]  (+ add[r, k] when `add` != NULL)
constexpr int LB_OC = 2048;                          // output features staged per pass (8 rows x 2048 floats = 64 KiB of LDS)
__global__ __launch_bounds__(256) void lin_bwd_x_kernel(const float* __restrict__ dy, const float* __restrict__ W, int64_t R, int K,
                                                        int O, const float* __restrict__ add, float* __restrict__ dx) {
  extern __shared__ float ds[];                      // [LB_ROWS][min(O, LB_OC)]
  const int64_t r0 = (int64_t)blockIdx.x * LB_ROWS;
  const int nr = (int)min((int64_t)LB_ROWS, R - r0);
  const int oc = O < LB_OC ? O : LB_OC;
  // (K <= 256 * 4 per thread pass: the accumulators of one k live across the passes over O)
  for (int k0 = 0; k0 < K; k0 += 256) {
    const int k = k0 + threadIdx.x;
    float acc[LB_ROWS];
#pragma unroll
    for (int rr = 0; rr < LB_ROWS; ++rr) acc[rr] = 0.f;
    for (int o0 = 0; o0 < O; o0 += oc) {
      const int no = min(oc, O - o0);
      __syncthreads();
      for (int i = threadIdx.x; i < nr * no; i += 256) { const int rr = i / no, o = i - rr * no; ds[rr * oc + o] = dy[(r0 + rr) * O + o0 + o]; }
      __syncthreads();
      if (k < K)
        for (int o = 0; o < no; ++o) {
          const float w = W[(size_t)(o0 + o) * K + k];
#pragma unroll
          for (int rr = 0; rr < LB_ROWS; ++rr) acc[rr] = fmaf(ds[rr * oc + o], w, acc[rr]);
        }
    }
    if (k < K) {
#pragma unroll
      for (int rr = 0; rr < LB_ROWS; ++rr)
        if (rr < nr) dx[(r0 + rr) * K + k] = acc[rr] + (add ? add[(r0 + rr) * K + k] : 0.f);
    }
  }
}
// dW[o, k] = sum_r dy[r, o] x[r, k]: grid (ceil(K / 256), O); rows gathered through `gather` when x is an embedding table
__global__ __launch_bounds__(256) void lin_bwd_w_kernel(const float* __restrict__ dy, const float* __restrict__ x, int64_t R, int K,
                                                        int O, float* __restrict__ dW) {
  const int k = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y;
  if (k >= K) return;
  float a0 = 0.f, a1 = 0.f;
  int64_t r = 0;
  for (; r + 1 < R; r += 2) {
    a0 = fmaf(dy[r * O + o], x[r * K + k], a0);
    a1 = fmaf(dy[(r + 1) * O + o], x[(r + 1) * K + k], a1);
  }
  if (r < R) a0 = fmaf(dy[r * O + o], x[r * K + k], a0);
  dW[(size_t)o * K + k] = a0 + a1;
}
// db[o] = sum_r dy[r, o]
__global__ __launch_bounds__(256) void colsum_small_kernel(const float* __restrict__ dy, int64_t R, int O, float* __restrict__ db) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= O) return;
  float s = 0.f;
  for (int64_t r = 0; r < R; ++r) s += dy[r * O + o];
  db[o] = s;
}

// ---------------------------------------------------------------- AdditiveAttention backward (attention.py:21-27)
// pre [B, S, Q] = x W^T + b (from the linear kernel).  One workgroup per b: a = tanh(pre), w = softmax_s(a . q),
// out = sum_s w_s x_s.  Given dout [B, D]: dw_s = dout . x_s, dlogit_s = w_s (dw_s - sum w dw), dq += sum_s dlogit_s a_s,
// dpre = dlogit_s q (1 - a^2) (written over `pre`), w [B, S] kept for dx_s = w_s dout + dpre_s W.
constexpr int POOL_MAX_S = 1024;
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ x, float* __restrict__ pre, const float* __restrict__ query,
                                                       const float* __restrict__ dout, int S, int D, int Q, float* __restrict__ wout,
                                                       float* __restrict__ dq) {
  __shared__ float lg[POOL_MAX_S], dw[POOL_MAX_S];
  __shared__ float red[4];
  const int64_t b = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* pb = pre + (size_t)b * S * Q;
  const float* xb = x + (size_t)b * S * D;
  const float* gb = dout + (size_t)b * D;
  for (int s = wave; s < S; s += 4) {                      // logits and dw, one wave per position
    float l = 0.f, d = 0.f;
    for (int j = lane; j < Q; j += 64) l = fmaf(tanhf(pb[(size_t)s * Q + j]), query[j], l);
    for (int c = lane; c < D; c += 64) d = fmaf(gb[c], xb[(size_t)s * D + c], d);
    l = wsum(l); d = wsum(d);
    if (lane == 0) { lg[s] = l; dw[s] = d; }
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int s = threadIdx.x; s < S; s += 256) mx = fmaxf(mx, lg[s]);
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float se = 0.f;
  for (int s = threadIdx.x; s < S; s += 256) se += expf(lg[s] - mx);
  se = wsum(se);
  if (lane == 0) red[wave] = se;
  __syncthreads();
  const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
  __syncthreads();
  float c = 0.f;
  for (int s = threadIdx.x; s < S; s += 256) {
    const float w = expf(lg[s] - mx) * inv;
    lg[s] = w;                                             // lg now holds the weights
    c = fmaf(w, dw[s], c);
  }
  c = wsum(c);
  if (lane == 0) red[wave] = c;
  __syncthreads();
  c = red[0] + red[1] + red[2] + red[3];
  for (int s = threadIdx.x; s < S; s += 256) {
    wout[b * S + s] = lg[s];
    dw[s] = lg[s] * (dw[s] - c);                           // dlogit_s
  }
  __syncthreads();
  for (int j = threadIdx.x; j < Q; j += 256) {
    const float qj = query[j];
    float acc = 0.f;
    for (int s = 0; s < S; ++s) {
      const float a = tanhf(pb[(size_t)s * Q + j]);
      acc = fmaf(dw[s], a, acc);
      pb[(size_t)s * Q + j] = dw[s] * qj * (1.f - a * a);
    }
    atomicAdd(dq + j, acc);
  }
}
// add[b, s, :] = w[b, s] * dout[b, :]
__global__ __launch_bounds__(256) void pool_wdout_kernel(const float* __restrict__ w, const float* __restrict__ dout, int64_t R, int S,
                                                         int D, float* __restrict__ add) {
  const int64_t r = blockIdx.x;
  const int64_t b = r / S;
  const float ws = w[r];
  for (int c = threadIdx.x; c < D; c += 256) add[r * D + c] = ws * dout[b * D + c];
}

// ---------------------------------------------------------------- axis-0 attention backward
// qkv [N, E, 3D] (q | k | v), d_att [N, E, D] -> d_qkv [N, E, 3D]; attention along the N axis for each (slot e, head).
// Kernel 1 (thread = query row): softmax statistics {max, sum}, D_i = sum_j dP_ij P_ij, dq_i.  Kernel 2 (thread = key row):
// dk_j = sum_i dS_ij q_i (scaled), dv_j = sum_i P_ij dO_i.
template <int DH, int KT>
__global__ __launch_bounds__(256) void axis0_bwd_q_kernel(const float* __restrict__ qkv, const float* __restrict__ datt,
                                                          float* __restrict__ dqkv, float* __restrict__ stats /*[N, E*heads, 3]*/,
                                                          int64_t N, int E, int D, int heads) {
  __shared__ float ks[KT * DH], vs[KT * DH];
  const int eh = blockIdx.x, e = eh / heads, hh = eh - e * heads;
  const int64_t n = (int64_t)blockIdx.y * 256 + threadIdx.x;
  const float scale = 1.0f / sqrtf((float)DH);
  const size_t ld = (size_t)E * 3 * D;
  const float* base = qkv + (size_t)e * 3 * D + hh * DH;
  float q[DH], go[DH], dq[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) {
    q[d] = n < N ? base[n * ld + d] * scale : 0.f;
    go[d] = n < N ? datt[(size_t)n * E * D + (size_t)e * D + hh * DH + d] : 0.f;
    dq[d] = 0.f;
  }
  float mx = -INFINITY, sum = 0.f, Dn = 0.f;
  for (int pass = 0; pass < 3; ++pass) {
    for (int64_t t0 = 0; t0 < N; t0 += KT) {
      __syncthreads();
      for (int i = threadIdx.x; i < KT * DH; i += 256) {
        const int j = i / DH, d = i - j * DH;
        const int64_t kn = t0 + j;
        ks[i] = kn < N ? base[kn * ld + D + d] : 0.f;
        vs[i] = kn < N ? base[kn * ld + 2 * D + d] : 0.f;
      }
      __syncthreads();
      const int cnt = (int)min((int64_t)KT, N - t0);
      for (int j = 0; j < cnt; ++j) {
        float s = 0.f, gv = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) { s = fmaf(q[d], ks[j * DH + d], s); gv = fmaf(go[d], vs[j * DH + d], gv); }
        if (pass == 0) {
          if (s > mx) { sum *= expf(mx - s); mx = s; }
          sum += expf(s - mx);
        } else {
          const float p = expf(s - mx) / sum;
          if (pass == 1) Dn = fmaf(gv, p, Dn);
          else {
            const float ds = p * (gv - Dn) * scale;
#pragma unroll
            for (int d = 0; d < DH; ++d) dq[d] = fmaf(ds, ks[j * DH + d], dq[d]);
          }
        }
      }
    }
  }
  if (n < N) {
    float* dst = dqkv + n * ld + (size_t)e * 3 * D + hh * DH;
#pragma unroll
    for (int d = 0; d < DH; ++d) dst[d] = dq[d];
    float* st = stats + ((size_t)n * E * heads + eh) * 3;
    st[0] = mx; st[1] = sum; st[2] = Dn;
  }
}
template <int DH, int KT>
__global__ __launch_bounds__(256) void axis0_bwd_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ datt,
                                                           const float* __restrict__ stats, float* __restrict__ dqkv, int64_t N,
                                                           int E, int D, int heads) {
  __shared__ float qs[KT * DH], gs[KT * DH];
  __shared__ float sm[KT], sl[KT], sd[KT];
  const int eh = blockIdx.x, e = eh / heads, hh = eh - e * heads;
  const int64_t n = (int64_t)blockIdx.y * 256 + threadIdx.x;
  const float scale = 1.0f / sqrtf((float)DH);
  const size_t ld = (size_t)E * 3 * D;
  const float* base = qkv + (size_t)e * 3 * D + hh * DH;
  float k[DH], v[DH], dk[DH], dv[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) {
    k[d] = n < N ? base[n * ld + D + d] : 0.f;
    v[d] = n < N ? base[n * ld + 2 * D + d] : 0.f;
    dk[d] = dv[d] = 0.f;
  }
  for (int64_t t0 = 0; t0 < N; t0 += KT) {
    __syncthreads();
    for (int i = threadIdx.x; i < KT * DH; i += 256) {
      const int j = i / DH, d = i - j * DH;
      const int64_t qn = t0 + j;
      qs[i] = qn < N ? base[qn * ld + d] * scale : 0.f;
      gs[i] = qn < N ? datt[(size_t)qn * E * D + (size_t)e * D + hh * DH + d] : 0.f;
    }
    for (int j = threadIdx.x; j < KT; j += 256) {
      const int64_t qn = t0 + j;
      const float* st = stats + ((size_t)(qn < N ? qn : 0) * E * heads + eh) * 3;
      sm[j] = st[0]; sl[j] = 1.f / st[1]; sd[j] = st[2];
    }
    __syncthreads();
    const int cnt = (int)min((int64_t)KT, N - t0);
    for (int i = 0; i < cnt; ++i) {
      float s = 0.f, gv = 0.f;
#pragma unroll
      for (int d = 0; d < DH; ++d) { s = fmaf(qs[i * DH + d], k[d], s); gv = fmaf(gs[i * DH + d], v[d], gv); }
      const float p = expf(s - sm[i]) * sl[i];
      const float ds = p * (gv - sd[i]);                  // qs carries the 1/sqrt(dh)
#pragma unroll
      for (int d = 0; d < DH; ++d) { dk[d] = fmaf(ds, qs[i * DH + d], dk[d]); dv[d] = fmaf(p, gs[i * DH + d], dv[d]); }
    }
  }
  if (n < N) {
    float* dst = dqkv + n * ld + (size_t)e * 3 * D + hh * DH;
#pragma unroll
    for (int d = 0; d < DH; ++d) { dst[D + d] = dk[d]; dst[2 * D + d] = dv[d]; }
  }
}
// forward core only (projections are separate Linear ops in the training graph): same arithmetic as entity.hip's kernel
template <int DH, int KT>
__global__ __launch_bounds__(256) void axis0_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out, int64_t N, int E, int D,
                                                        int heads) {
  __shared__ float ks[KT * DH], vs[KT * DH];
  const int eh = blockIdx.x, e = eh / heads, hh = eh - e * heads;
  const int64_t n = (int64_t)blockIdx.y * 256 + threadIdx.x;
  const float scale = 1.0f / sqrtf((float)DH);
  const size_t ld = (size_t)E * 3 * D;
  const float* base = qkv + (size_t)e * 3 * D + hh * DH;
  float q[DH], o[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) { q[d] = n < N ? base[n * ld + d] * scale : 0.f; o[d] = 0.f; }
  float mx = -INFINITY, sum = 0.f;
  for (int64_t t0 = 0; t0 < N; t0 += KT) {
    __syncthreads();
    for (int i = threadIdx.x; i < KT * DH; i += 256) {
      const int j = i / DH, d = i - j * DH;
      const int64_t kn = t0 + j;
      ks[i] = kn < N ? base[kn * ld + D + d] : 0.f;
      vs[i] = kn < N ? base[kn * ld + 2 * D + d] : 0.f;
    }
    __syncthreads();
    const int cnt = (int)min((int64_t)KT, N - t0);
    for (int j = 0; j < cnt; ++j) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < DH; ++d) s = fmaf(q[d], ks[j * DH + d], s);
      if (s > mx) {
        const float f = expf(mx - s);
        sum *= f;
#pragma unroll
        for (int d = 0; d < DH; ++d) o[d] *= f;
        mx = s;
      }
      const float p = expf(s - mx);
      sum += p;
#pragma unroll
      for (int d = 0; d < DH; ++d) o[d] = fmaf(p, vs[j * DH + d], o[d]);
    }
  }
  if (n < N) {
    const float inv = 1.0f / sum;
    float* dst = out + (size_t)n * E * D + (size_t)e * D + hh * DH;
#pragma unroll
    for (int d = 0; d < DH; ++d) dst[d] = o[d] * inv;
  }
}

// ---------------------------------------------------------------- nn.Embedding
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ table,
                                                            int64_t n_rows, int D, float* __restrict__ out, int32_t* __restrict__ status) {
  const int64_t r = blockIdx.x;
  int64_t id = ids[r];
  if (id < 0 || id >= n_rows) { if (threadIdx.x == 0 && status) atomicOr(status, MANNER_HIP_STATUS_INDEX); id = 0; }
  for (int c = threadIdx.x; c < D; c += 256) out[r * D + c] = table[id * D + c];
}
// dtable[ids[r]] += dy[r] (f32 atomics); rows equal to padding_idx receive nothing (nn.Embedding(padding_idx=...))
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dy, int64_t n_rows,
                                                            int D, int64_t padding_idx, float* __restrict__ dtable) {
  const int64_t r = blockIdx.x;
  const int64_t id = ids[r];
  if (id < 0 || id >= n_rows || id == padding_idx) return;
  for (int c = threadIdx.x; c < D; c += 256) atomicAdd(dtable + id * D + c, dy[r * D + c]);
}

#define MANNER_AXIS0_DISPATCH(DH_, CALL)                                                                    \
  switch (DH_) {                                                                                            \
    case 4: CALL(4, 256); break;                                                                            \
    case 8: CALL(8, 256); break;                                                                            \
    case 10: CALL(10, 256); break;                                                                          \
    case 16: CALL(16, 128); break;                                                                          \
    case 32: CALL(32, 64); break;                                                                           \
    case 48: CALL(48, 32); break;                                                                           \
    case 64: CALL(64, 32); break;                                                                           \
    default: return fail(MANNER_HIP_E_INVALID, "axis-0 attention: head_dim %d unsupported (4, 8, 10, 16, 32, 48, 64)", DH_); \
  }

}  // namespace
}  // namespace manner

using namespace manner;

extern "C" {

int manner_hip_dropout(const float* x, float* out, int64_t n, uint64_t seed, uint32_t site, float p, manner_hip_stream_t stream) {
  if (n == 0) return MANNER_HIP_OK;
  if (!x || !out || n < 0 || !(p >= 0.f && p < 1.f)) return fail(MANNER_HIP_E_INVALID, "dropout: bad argument");
  const double t = (double)p * 4294967296.0;
  const uint32_t thr = p <= 0.f ? 0u : (t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t);
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(dropout_flat_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, x, out, n, seed,
                     site, thr, p <= 0.f ? 1.f : 1.f / (1.f - p));
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int manner_hip_linear_backward(const float* x, const float* weight, const float* grad_y, int64_t R, int32_t K, int32_t O,
                               const float* add_to_dx, float* grad_x, float* grad_w, float* grad_b, manner_hip_stream_t stream) {
  if (R < 0 || K <= 0 || O <= 0) return fail(MANNER_HIP_E_INVALID, "linear_backward: bad shape");
  hipStream_t s = (hipStream_t)stream;
  if (R == 0) {                                    // no rows: the parameter gradients are sums over nothing = 0 (grad_x is empty)
    if (grad_w) MANNER_HIP_TRY(hipMemsetAsync(grad_w, 0, (size_t)O * K * sizeof(float), s));
    if (grad_b) MANNER_HIP_TRY(hipMemsetAsync(grad_b, 0, (size_t)O * sizeof(float), s));
    return MANNER_HIP_OK;
  }
  if (!grad_y || (grad_x && !weight) || (grad_w && !x)) return fail(MANNER_HIP_E_INVALID, "linear_backward: null pointer");
  if (grad_x) {
    hipLaunchKernelGGL(lin_bwd_x_kernel, dim3((unsigned)((R + LB_ROWS - 1) / LB_ROWS)), dim3(256), LB_ROWS * (O < LB_OC ? O : LB_OC) * sizeof(float), s, grad_y,
                       weight, R, K, O, add_to_dx, grad_x);
    MANNER_LAUNCH_CHECK();
  }
  if (grad_w) {
    hipLaunchKernelGGL(lin_bwd_w_kernel, dim3((unsigned)((K + 255) / 256), (unsigned)O), dim3(256), 0, s, grad_y, x, R, K, O, grad_w);
    MANNER_LAUNCH_CHECK();
  }
  if (grad_b) {
    hipLaunchKernelGGL(colsum_small_kernel, dim3((unsigned)((O + 255) / 256)), dim3(256), 0, s, grad_y, R, O, grad_b);
    MANNER_LAUNCH_CHECK();
  }
  return MANNER_HIP_OK;
}

size_t manner_hip_additive_pool_backward_workspace_bytes(int64_t B, int64_t S, int32_t D, int32_t Q) {
  if (B <= 0 || S <= 0 || D <= 0 || Q <= 0) return 0;
  return (size_t)(B * S) * (size_t)(Q + D + 1) * sizeof(float) + 1024;
}

int manner_hip_additive_pool_backward(const float* x, const float* lin_w, const float* lin_b, const float* query, const float* grad_out,
                                      int64_t B, int64_t S, int32_t D, int32_t Q, float* grad_x, float* grad_w, float* grad_b,
                                      float* grad_q, void* workspace, size_t workspace_bytes, manner_hip_stream_t stream) {
  if (B < 0 || S < 0 || D <= 0 || Q <= 0) return fail(MANNER_HIP_E_INVALID, "additive_pool_backward: bad shape");
  if (B == 0 || S == 0) {                          // an empty batch / empty sequences: zero parameter gradients (grad_x is empty)
    hipStream_t s0 = (hipStream_t)stream;
    if (grad_w) MANNER_HIP_TRY(hipMemsetAsync(grad_w, 0, (size_t)Q * D * sizeof(float), s0));
    if (grad_b) MANNER_HIP_TRY(hipMemsetAsync(grad_b, 0, (size_t)Q * sizeof(float), s0));
    if (grad_q) MANNER_HIP_TRY(hipMemsetAsync(grad_q, 0, (size_t)Q * sizeof(float), s0));
    return MANNER_HIP_OK;
  }
  if (!x || !lin_w || !lin_b || !query || !grad_out || !grad_x || !grad_w || !grad_b || !grad_q || !workspace ||
      S > POOL_MAX_S || D <= 0 || Q <= 0)
    return fail(MANNER_HIP_E_INVALID, "additive_pool_backward: bad argument (S <= %d)", POOL_MAX_S);
  if (workspace_bytes < manner_hip_additive_pool_backward_workspace_bytes(B, S, D, Q))
    return fail(MANNER_HIP_E_WORKSPACE, "additive_pool_backward: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int64_t R = B * S;
  float* pre = static_cast<float*>(workspace);          // [R, Q]: pre-activations, then d pre
  float* add = pre + (size_t)R * Q;                     // [R, D]: w_s * dout
  float* w = add + (size_t)R * D;                       // [R]
  int rc;
  if ((rc = manner_hip_linear(x, lin_w, lin_b, R, D, Q, pre, stream))) return rc;
  MANNER_HIP_TRY(hipMemsetAsync(grad_q, 0, (size_t)Q * sizeof(float), s));
  hipLaunchKernelGGL(pool_bwd_kernel, dim3((unsigned)B), dim3(256), 0, s, x, pre, query, grad_out, (int)S, D, Q, w, grad_q);
  MANNER_LAUNCH_CHECK();
  hipLaunchKernelGGL(pool_wdout_kernel, dim3((unsigned)R), dim3(256), 0, s, w, grad_out, R, (int)S, D, add);
  MANNER_LAUNCH_CHECK();
  return manner_hip_linear_backward(x, lin_w, pre, R, D, Q, add, grad_x, grad_w, grad_b, stream);
}

int manner_hip_axis0_attention(const float* qkv, int64_t L0, int64_t B1, int32_t E, int32_t heads, float* out, manner_hip_stream_t stream) {
  if (L0 == 0 || B1 == 0) return MANNER_HIP_OK;
  if (!qkv || !out || L0 < 0 || B1 < 0 || E <= 0 || heads <= 0 || E % heads) return fail(MANNER_HIP_E_INVALID, "axis0_attention: bad argument");
  const dim3 g((unsigned)(B1 * heads), (unsigned)((L0 + 255) / 256)), b(256);
  hipStream_t s = (hipStream_t)stream;
#define MANNER_A0F(DH_, KT_) hipLaunchKernelGGL((axis0_fwd_kernel<DH_, KT_>), g, b, 0, s, qkv, out, L0, (int)B1, E, heads)
  MANNER_AXIS0_DISPATCH(E / heads, MANNER_A0F)
#undef MANNER_A0F
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int manner_hip_axis0_attention_backward(const float* qkv, const float* grad_out, int64_t L0, int64_t B1, int32_t E, int32_t heads,
                                        float* grad_qkv, float* stats /*[L0 * B1 * heads * 3]*/, manner_hip_stream_t stream) {
  if (L0 == 0 || B1 == 0) return MANNER_HIP_OK;
  if (!qkv || !grad_out || !grad_qkv || !stats || L0 < 0 || B1 < 0 || E <= 0 || heads <= 0 || E % heads)
    return fail(MANNER_HIP_E_INVALID, "axis0_attention_backward: bad argument");
  const dim3 g((unsigned)(B1 * heads), (unsigned)((L0 + 255) / 256)), b(256);
  hipStream_t s = (hipStream_t)stream;
#define MANNER_A0B(DH_, KT_)                                                                                                     \
  do {                                                                                                                           \
    hipLaunchKernelGGL((axis0_bwd_q_kernel<DH_, KT_>), g, b, 0, s, qkv, grad_out, grad_qkv, stats, L0, (int)B1, E, heads);       \
    hipLaunchKernelGGL((axis0_bwd_kv_kernel<DH_, KT_>), g, b, 0, s, qkv, grad_out, stats, grad_qkv, L0, (int)B1, E, heads);      \
  } while (0)
  MANNER_AXIS0_DISPATCH(E / heads, MANNER_A0B)
#undef MANNER_A0B
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int manner_hip_embedding(const int64_t* ids, int64_t R, const float* table, int64_t n_rows, int32_t D, float* out, int32_t* status,
                         manner_hip_stream_t stream) {
  if (R == 0) return MANNER_HIP_OK;
  if (!ids || !table || !out || R < 0 || n_rows <= 0 || D <= 0) return fail(MANNER_HIP_E_INVALID, "embedding: bad argument");
  hipLaunchKernelGGL(embedding_fwd_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, ids, table, n_rows, D, out, status);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int manner_hip_embedding_backward(const int64_t* ids, int64_t R, const float* grad_out, int64_t n_rows, int32_t D, int64_t padding_idx,
                                  float* grad_table, manner_hip_stream_t stream) {
  if (!grad_table || n_rows <= 0 || D <= 0) return fail(MANNER_HIP_E_INVALID, "embedding_backward: bad argument");
  hipStream_t s = (hipStream_t)stream;
  MANNER_HIP_TRY(hipMemsetAsync(grad_table, 0, (size_t)n_rows * D * sizeof(float), s));
  if (R == 0) return MANNER_HIP_OK;
  if (!ids || !grad_out || R < 0) return fail(MANNER_HIP_E_INVALID, "embedding_backward: bad argument");
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3((unsigned)R), dim3(256), 0, s, ids, grad_out, n_rows, D, padding_idx, grad_table);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

}  // extern "C"
