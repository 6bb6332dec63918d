// ppo_optim.hpp -- the tail of one PPO2 optimizer step (flex_gym/algo/ppo2/ppo2.py:182-197: tf.clip_by_global_norm, then
// tf.train.AdamOptimizer(learning_rate, epsilon=1e-5).apply_gradients) on FLAT buffers, gfx950.
//
// The learner keeps every parameter of a policy as a view of ONE persistent flat buffer, and the gradients and Adam moments
// likewise (ppo2.FlatParams): the data-parallel all-reduce is then one collective on the gradient buffer with nothing packed or
// unpacked around it (283 KB for the LSTM policy, 56 KB for the MLP), and clip + Adam is the single launch below instead of ~25
// small ones (per-tensor norms, the stack / norm / clamp of clip_grad_norm_, the multi-tensor Adam, 19 copies back).
//
//   irrl_clip_adam_kernel        ONE workgroup of 1024 lanes: sum g^2 in a fixed order (deterministic: every rank computes the
//                                same bits from the same all-reduced gradient, so replicas never drift), scale =
//                                grad_scale * max_norm / max(grad_scale * |g|, max_norm) (tf.clip_by_global_norm), then
//                                TensorFlow's Adam in place (epsilon outside the bias correction, see the kernel).  70 741
//                                parameters = 17 float4 per lane per array, L2-resident.
//   irrl_sum_rows_scatter_kernel out[map[m][c]] = add[m][c] + sum_r part[m][r][c]: the per-workgroup partial rows the MlpPolicy
//                                gradient kernels leave, summed in a fixed order and written STRAIGHT into the flat gradient
//                                buffer in parameter layout (map < 0: column not a parameter).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct ClipAdamArgs {
  int n;                    // parameters
  float *theta;             // [n] parameters, updated in place
  const float *g;           // [n] gradient (summed over ranks when grad_scale = 1 / world)
  float *m, *v;             // [n] Adam moments, updated in place
  float grad_scale;         // applied to g before everything else (1 / world: the mean over ranks of the all-reduced sum)
  float max_norm;           // <= 0: no clipping
  float beta1, beta2, eps;
  float step_size;          // lr / (1 - beta1^t)
  float inv_bc2_sqrt;       // 1 / sqrt(1 - beta2^t)
  float *norm_out;          // [1] or NULL: the (unclipped, scaled) global gradient norm
};

__global__ void __launch_bounds__(1024)
irrl_clip_adam_kernel(const ClipAdamArgs a) {
  __shared__ float red[16];
  __shared__ float bc;
  const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n4 = a.n >> 2;
  const float4 *g4 = (const float4 *)a.g;
  float acc = 0.0f;
  for (int i = tid; i < n4; i += 1024) {
    const float4 x = g4[i];
    acc += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
  }
  for (int i = (n4 << 2) + tid; i < a.n; i += 1024) { const float x = a.g[i]; acc += x * x; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) red[wv] = acc;
  __syncthreads();
  if (tid == 0) {
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += red[i];
    const float norm = a.grad_scale * sqrtf(s);
    float coef = 1.0f;
    if (a.max_norm > 0.0f) coef = a.max_norm / fmaxf(norm, a.max_norm);   // tf.clip_by_global_norm: t * clip_norm / max(global_norm, clip_norm)
    bc = a.grad_scale * coef;
    if (a.norm_out) *a.norm_out = norm;
  }
  __syncthreads();
  const float sc = bc, b1 = a.beta1, b2 = a.beta2, ob1 = 1.0f - a.beta1, ob2 = 1.0f - a.beta2;
  auto upd = [&](float g, float &m, float &v, float &th) {
    g *= sc;
    m = m + ob1 * (g - m);                 // lerp(m, g, 1 - beta1)
    v = b2 * v + ob2 * (g * g);
    // tf.train.AdamOptimizer: theta -= lr * sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps) -- epsilon sits OUTSIDE the bias correction
    // ("epsilon hat" of Kingma & Ba), i.e. an effective epsilon of eps / sqrt(1 - b2^t): 3e-4 at t = 1 for eps 1e-5, where
    // torch.optim.Adam (sqrt(v) / sqrt(1 - b2^t) + eps) would use 1e-5
    const float denom = (sqrtf(v) + a.eps) * a.inv_bc2_sqrt;
    th -= a.step_size * (m / denom);
    (void)b1;
  };
  float4 *m4 = (float4 *)a.m, *v4 = (float4 *)a.v, *t4 = (float4 *)a.theta;
  for (int i = tid; i < n4; i += 1024) {
    const float4 x = g4[i];
    float4 m = m4[i], v = v4[i], t = t4[i];
    upd(x.x, m.x, v.x, t.x); upd(x.y, m.y, v.y, t.y); upd(x.z, m.z, v.z, t.z); upd(x.w, m.w, v.w, t.w);
    m4[i] = m; v4[i] = v; t4[i] = t;
  }
  for (int i = (n4 << 2) + tid; i < a.n; i += 1024) upd(a.g[i], a.m[i], a.v[i], a.theta[i]);
}

// part [nmat, rows, cols]; map [nmat, cols] (int32; < 0: skip); add [nmat, cols] or NULL.  A workgroup owns 64 columns of one matrix:
// its four waves each add a quarter of the rows (coalesced 256-byte reads, eight loads in flight per lane), then add up through LDS.
__global__ void __launch_bounds__(256)
irrl_sum_rows_scatter_kernel(const float *__restrict__ part, int rows, int cols, const int *__restrict__ map, const float *__restrict__ add,
                             float *__restrict__ out) {
  __shared__ float red[4][64];
  const int cx = threadIdx.x & 63, rg = threadIdx.x >> 6, mat = (int)blockIdx.y;
  const int col = blockIdx.x * 64 + cx;
  float acc = 0.0f;
  if (col < cols) {
    const int per = (rows + 3) / 4, r0 = rg * per, r1 = min(rows, r0 + per);
    const float *p = part + ((size_t)mat * rows + r0) * cols + col;
    int r = r0;
    for (; r + 8 <= r1; r += 8) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; i++) v[i] = p[(size_t)i * cols];
#pragma unroll
      for (int i = 0; i < 8; i++) acc += v[i];
      p += (size_t)8 * cols;
    }
    for (; r < r1; r++) { acc += *p; p += cols; }
  }
  red[rg][cx] = acc;
  __syncthreads();
  if (rg == 0 && col < cols) {
    const int dst = map[(size_t)mat * cols + col];
    if (dst >= 0) {
      float s = ((red[0][cx] + red[1][cx]) + red[2][cx]) + red[3][cx];
      if (add) s += add[(size_t)mat * cols + col];
      out[dst] = s;
    }
  }
}

// ---- a random permutation of 0 .. n-1 without a sort (the shuffled sample order of an epoch, ppo2.py:364-380: np.random.shuffle(inds)) ----
// torch.randperm on the device is a radix sort of n random keys: 7 launches and 175 us for the 3.07 M samples of a 4096 x 750 rollout, ten
// times per update -- 6 % of the MlpPolicy update.  Here out[i] = E(i) with E a keyed bijection of [0, n): a 4-round Feistel network on the
// 2 b bits that hold n - 1 (round function: a 32-bit integer hash of the half and the round key), walked until the value falls below n
// (cycle walking: E restricted to [0, n) is again a bijection; 2^(2b) < 4 n, so fewer than 4 evaluations on average, 1.4 at n = 3.07 M).
// One launch, each element on its own: the order depends on (n, seed, counter) only -- every rank computes the same one, and the numpy
// twin (ppo2.feistel_permutation) gives the CPU path the same bits.
__device__ __host__ inline uint32_t irrl_perm_hash(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __host__ inline uint32_t irrl_perm_encrypt(uint32_t v, int half_bits, uint32_t seed, uint32_t counter) {
  const uint32_t mask = (1u << half_bits) - 1u;
  uint32_t L = v >> half_bits, R = v & mask;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const uint32_t k = irrl_perm_hash(seed + 0x9E3779B9u * (uint32_t)(r + 1)) ^ irrl_perm_hash(counter + 0x85EBCA6Bu * (uint32_t)(r + 1));
    const uint32_t F = irrl_perm_hash(R ^ k) & mask;
    const uint32_t nL = R, nR = L ^ F;
    L = nL; R = nR;
  }
  return (L << half_bits) | R;
}
__global__ void __launch_bounds__(256)
irrl_random_permutation_kernel(uint32_t n, int half_bits, uint32_t seed, uint32_t counter, long long *__restrict__ out) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  uint32_t v = irrl_perm_encrypt(i, half_bits, seed, counter);
  while (v >= n) v = irrl_perm_encrypt(v, half_bits, seed, counter);
  out[i] = (long long)v;
}
