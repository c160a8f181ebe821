// C2/K6 — gradient-norm clipping + AdamW over ONE flat f32 buffer of LoRA parameters.
//
// Replaces training_scripts/train_lora_dreambooth.py:878-888: the reference's
// clip_grad_norm_ walks every UNet parameter tensor (~860 M frozen elements) to
// find the 288 LoRA tensors, then AdamW runs per-tensor foreach kernels and
// zero_grad another pass.  The trainer keeps all A/B parameters, their grads and
// both Adam moments in flat buffers (the grad buffer is also the single RCCL
// all-reduce payload), so the whole optimiser step is two launches:
//   sumsq       : ||g||^2 -> device scalar (two-stage, deterministic)
//   clip_adamw  : g*=clip(norm); AdamW (torch.optim.AdamW maths); g=0
// No host synchronisation: the clip coefficient is derived on the device.
#include <algorithm>

#include "common.hpp"

namespace lora_amd {

constexpr int kOptThreads = 256;
constexpr int kSumsqMaxBlocks = 1024;

__device__ inline float block_sum(float v, float *s_buf) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s_buf[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < kOptThreads / 64; ++w) t += s_buf[w];
  return t;
}

__global__ __launch_bounds__(kOptThreads) void sumsq_stage1(const float *__restrict__ g, int64_t n,
                                                             float *__restrict__ partial) {
  __shared__ float s_buf[kOptThreads / 64];
  float acc = 0.f;
  const int64_t n4 = n >> 2;
  const float4 *g4 = reinterpret_cast<const float4 *>(g);
  for (int64_t i = (int64_t)blockIdx.x * kOptThreads + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kOptThreads) {
    float4 v = g4[i];
    acc = fmaf(v.x, v.x, acc); acc = fmaf(v.y, v.y, acc); acc = fmaf(v.z, v.z, acc); acc = fmaf(v.w, v.w, acc);
  }
  if (blockIdx.x == 0)
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += kOptThreads) acc = fmaf(g[i], g[i], acc);
  float t = block_sum(acc, s_buf);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(kOptThreads) void sumsq_stage2(const float *__restrict__ partial, int nblk,
                                                             float *__restrict__ out) {
  __shared__ float s_buf[kOptThreads / 64];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nblk; i += kOptThreads) acc += partial[i];
  float t = block_sum(acc, s_buf);
  if (threadIdx.x == 0) out[0] = t;
}

__global__ __launch_bounds__(kOptThreads) void clip_adamw_kernel(
    float *__restrict__ p, float *__restrict__ g, float *__restrict__ m, float *__restrict__ v, int64_t n,
    const lora_amd_adamw_group *__restrict__ groups, int n_groups, const float *__restrict__ sumsq,
    float grad_scale, float max_norm, float beta1, float beta2, float eps, int64_t step_host,
    const int64_t *__restrict__ step_dev, const float *__restrict__ scaler, int zero_grad) {
  // bias corrections in double, as torch's python scalars are; the step may live on the device so
  // that a captured hipGraph replays unchanged from step to step
  const double step = (double)(step_dev != nullptr ? step_dev[0] : step_host);
  const float bc1 = (float)(1.0 - pow((double)beta1, step));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, step));
  // fp16 loss scaling (lora_amd_loss_scale_update ran before this launch): scaler[2] = 1 / (scale the backward used),
  // scaler[3] = 0 when the reduced gradient holds inf/nan -> the update is skipped (GradScaler.step), grads still zeroed
  bool apply = true;
  if (scaler != nullptr) {
    grad_scale *= scaler[2];
    apply = scaler[3] != 0.f;
  }
  float coef = grad_scale;
  if (max_norm > 0.f && apply) {
    // torch.nn.utils.clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1
    const float total_norm = sqrtf(sumsq[0]) * fabsf(grad_scale);
    const float c = max_norm / (total_norm + 1e-6f);
    coef *= fminf(c, 1.0f);
  }
  for (int64_t i = (int64_t)blockIdx.x * kOptThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kOptThreads) {
    float lr = 0.f, wd = 0.f;
    bool owned = false;
    for (int k = 0; k < n_groups; ++k) {
      if (i >= groups[k].begin && i < groups[k].end) { lr = groups[k].lr; wd = groups[k].weight_decay; owned = true; }
    }
    if (!owned) continue;
    if (!apply) {
      if (zero_grad) g[i] = 0.f;
      continue;
    }
    const float grad = g[i] * coef;
    float pi = p[i];
    pi *= (1.0f - lr * wd);                      // param.mul_(1 - lr * weight_decay)
    float mi = m[i];
    mi = mi + (grad - mi) * (1.0f - beta1);      // exp_avg.lerp_(grad, 1 - beta1)
    float vi = v[i] * beta2;
    vi = fmaf((1.0f - beta2) * grad, grad, vi);  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * (mi / denom);             // param.addcdiv_(exp_avg, denom, value=-step_size)
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (zero_grad) g[i] = 0.f;
  }
}

}  // namespace lora_amd

using namespace lora_amd;

static int sumsq_blocks(int64_t n) {
  int64_t b = (n / 4 + kOptThreads - 1) / kOptThreads;
  if (b < 1) b = 1;
  return (int)std::min<int64_t>(b, kSumsqMaxBlocks);
}

extern "C" size_t lora_amd_sumsq_workspace(int64_t n) { return (size_t)kSumsqMaxBlocks * sizeof(float); }

extern "C" int lora_amd_sumsq(const float *g, int64_t n, float *out_sumsq, void *workspace, size_t workspace_bytes,
                              void *stream) {
  LORA_AMD_CHECK(g && out_sumsq && workspace, LORA_AMD_EINVAL, "sumsq: null pointer");
  LORA_AMD_CHECK(n >= 0, LORA_AMD_EINVAL, "sumsq: n=%lld", (long long)n);
  LORA_AMD_CHECK(((uintptr_t)g & 15u) == 0, LORA_AMD_EINVAL, "sumsq: g must be 16-byte aligned");
  LORA_AMD_CHECK(workspace_bytes >= lora_amd_sumsq_workspace(n), LORA_AMD_EWORKSPACE, "sumsq: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int nblk = sumsq_blocks(n);
  float *partial = reinterpret_cast<float *>(workspace);
  hipLaunchKernelGGL(sumsq_stage1, dim3(nblk), dim3(kOptThreads), 0, st, g, n, partial);
  hipLaunchKernelGGL(sumsq_stage2, dim3(1), dim3(kOptThreads), 0, st, partial, nblk, out_sumsq);
  return check_launch("lora_amd_sumsq");
}

static int clip_adamw_impl(float *p, float *g, float *exp_avg, float *exp_avg_sq, int64_t n,
                           const lora_amd_adamw_group *groups_dev, int32_t n_groups, const float *sumsq,
                           float grad_scale, float max_norm, float beta1, float beta2, float eps, int64_t step,
                           const int64_t *step_dev, const float *scaler, int32_t zero_grad, void *stream) {
  LORA_AMD_CHECK(p && g && exp_avg && exp_avg_sq && groups_dev, LORA_AMD_EINVAL, "clip_adamw: null pointer");
  LORA_AMD_CHECK(n >= 0 && n_groups >= 1 && n_groups <= 16, LORA_AMD_EINVAL, "clip_adamw: n=%lld n_groups=%d",
                 (long long)n, n_groups);
  LORA_AMD_CHECK(step_dev != nullptr || step >= 1, LORA_AMD_EINVAL, "clip_adamw: step must be >= 1 (got %lld)",
                 (long long)step);
  LORA_AMD_CHECK(max_norm <= 0.f || sumsq != nullptr, LORA_AMD_EINVAL, "clip_adamw: clipping needs sumsq");
  if (n == 0) return LORA_AMD_OK;
  hipStream_t st = (hipStream_t)stream;
  int grid = (int)std::min<int64_t>((n + kOptThreads - 1) / kOptThreads, 2048);
  hipLaunchKernelGGL(clip_adamw_kernel, dim3(grid), dim3(kOptThreads), 0, st, p, g, exp_avg, exp_avg_sq, n,
                     groups_dev, n_groups, sumsq, grad_scale, max_norm, beta1, beta2, eps, step, step_dev, scaler, zero_grad);
  return check_launch("lora_amd_clip_adamw");
}

extern "C" int lora_amd_clip_adamw(float *p, float *g, float *exp_avg, float *exp_avg_sq, int64_t n,
                                   const lora_amd_adamw_group *groups_dev, int32_t n_groups, const float *sumsq,
                                   float grad_scale, float max_norm, float beta1, float beta2, float eps,
                                   int64_t step, int32_t zero_grad, void *stream) {
  return clip_adamw_impl(p, g, exp_avg, exp_avg_sq, n, groups_dev, n_groups, sumsq, grad_scale, max_norm, beta1,
                         beta2, eps, step, nullptr, nullptr, zero_grad, stream);
}

extern "C" int lora_amd_clip_adamw_dev(float *p, float *g, float *exp_avg, float *exp_avg_sq, int64_t n,
                                       const lora_amd_adamw_group *groups_dev, int32_t n_groups,
                                       const float *sumsq, float grad_scale, float max_norm, float beta1,
                                       float beta2, float eps, const int64_t *step_dev, const float *scaler,
                                       int32_t zero_grad, void *stream) {
  LORA_AMD_CHECK(step_dev != nullptr, LORA_AMD_EINVAL, "clip_adamw_dev: null step pointer");
  LORA_AMD_CHECK(scaler == nullptr || sumsq != nullptr, LORA_AMD_EINVAL, "clip_adamw_dev: loss scaling needs sumsq");
  return clip_adamw_impl(p, g, exp_avg, exp_avg_sq, n, groups_dev, n_groups, sumsq, grad_scale, max_norm, beta1,
                         beta2, eps, 0, step_dev, scaler, zero_grad, stream);
}

// Dynamic loss scaling for fp16 training (what accelerate's GradScaler does around the reference's step,
// train_lora_dreambooth.py:489-494 mixed_precision="fp16"), entirely on the device so that a captured step replays:
//   state[0] scale the NEXT backward multiplies the loss by   state[1] consecutive finite steps
//   state[2] 1 / (scale of the step being applied)            state[3] 1 = gradient finite, 0 = skip this update
// Runs after sumsq (of the still-scaled, already all-reduced gradient) and before clip_adamw; advances the optimiser
// step counter only when the update will be applied (a skipped step does not count, as with GradScaler).
__global__ void loss_scale_update_kernel(float *state, const float *sumsq, int64_t *step_dev, float growth,
                                         float backoff, int interval) {
  const float scale = state[0];
  const bool finite = isfinite(sumsq[0]);
  state[2] = 1.0f / scale;
  state[3] = finite ? 1.0f : 0.0f;
  if (finite) {
    const float good = state[1] + 1.0f;
    if (good >= (float)interval) { state[0] = scale * growth; state[1] = 0.f; }
    else state[1] = good;
    if (step_dev != nullptr) step_dev[0] += 1;
  } else {
    state[0] = fmaxf(scale * backoff, 1.0f);
    state[1] = 0.f;
  }
}

extern "C" int lora_amd_loss_scale_update(float *state, const float *sumsq, int64_t *step_dev, float growth_factor,
                                          float backoff_factor, int32_t growth_interval, void *stream) {
  LORA_AMD_CHECK(state && sumsq, LORA_AMD_EINVAL, "loss_scale_update: null pointer");
  LORA_AMD_CHECK(growth_factor >= 1.f && backoff_factor > 0.f && backoff_factor <= 1.f && growth_interval >= 1,
                 LORA_AMD_EINVAL, "loss_scale_update: bad factors");
  hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, sumsq, step_dev,
                     growth_factor, backoff_factor, growth_interval);
  return check_launch("lora_amd_loss_scale_update");
}

// ---------------------------------------------------------------------------- textual-inversion rows (PTI phase 1)
// One workgroup per placeholder token: gather its gradient row from the embedding-table gradient, AdamW on the f32
// master row, pull the row norm toward 0.4 (ref cli_lora_pti.py:451-469), scatter the row back into the table.
// replaces: AdamW over the whole [vocab, hidden] table + normalisation + restoring every other row (:433-479).
template <class E>
__global__ __launch_bounds__(kOptThreads) void ti_rows_step_kernel(
    typename E::storage *__restrict__ table, const typename E::storage *__restrict__ table_grad,
    const int64_t *__restrict__ ids, float *__restrict__ rows, float *__restrict__ m, float *__restrict__ v, int hidden,
    float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale, int64_t step, float decay_lambda,
    float target_norm) {
  __shared__ float s_buf[kOptThreads / 64];
  const int64_t tok = ids[blockIdx.x];
  float *row = rows + (int64_t)blockIdx.x * hidden, *mr = m + (int64_t)blockIdx.x * hidden,
        *vr = v + (int64_t)blockIdx.x * hidden;
  const double st = (double)step;
  const float bc1 = (float)(1.0 - pow((double)beta1, st));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, st));
  float sq = 0.f;
  for (int i = threadIdx.x; i < hidden; i += kOptThreads) {
    const float grad = E::to_f(table_grad[tok * hidden + i]) * grad_scale;
    float pi = row[i] * (1.0f - lr * weight_decay);
    float mi = mr[i];
    mi = mi + (grad - mi) * (1.0f - beta1);
    float vi = vr[i] * beta2;
    vi = fmaf((1.0f - beta2) * grad, grad, vi);
    pi -= (lr / bc1) * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    row[i] = pi; mr[i] = mi; vr[i] = vi;
    sq = fmaf(pi, pi, sq);
  }
  const float norm = sqrtf(block_sum(sq, s_buf));
  // F.normalize(row) * (norm + lambda * (target - norm)); lambda < 0 disables the decay
  const float mult = decay_lambda >= 0.f ? (norm + decay_lambda * (target_norm - norm)) / fmaxf(norm, 1e-12f) : 1.0f;
  for (int i = threadIdx.x; i < hidden; i += kOptThreads) {
    const float pi = row[i] * mult;
    row[i] = pi;
    table[tok * hidden + i] = E::from_f(pi);
  }
}

extern "C" int lora_amd_ti_rows_step(void *table, const void *table_grad, const int64_t *ids_dev, int32_t n_tokens,
                                     int32_t hidden, int32_t table_dtype, float *rows, float *exp_avg,
                                     float *exp_avg_sq, float lr, float beta1, float beta2, float eps,
                                     float weight_decay, float grad_scale, int64_t step, float decay_lambda,
                                     float target_norm, void *stream) {
  LORA_AMD_CHECK(table && table_grad && ids_dev && rows && exp_avg && exp_avg_sq, LORA_AMD_EINVAL,
                 "ti_rows_step: null pointer");
  LORA_AMD_CHECK(dtype_ok(table_dtype) && hidden > 0 && n_tokens >= 0 && step >= 1, LORA_AMD_EINVAL,
                 "ti_rows_step: bad argument");
  if (n_tokens == 0) return LORA_AMD_OK;
  hipStream_t st = (hipStream_t)stream;
#define TI(E)                                                                                                   \
  hipLaunchKernelGGL((ti_rows_step_kernel<E>), dim3(n_tokens), dim3(kOptThreads), 0, st,                          \
                     reinterpret_cast<typename E::storage *>(table),                                             \
                     reinterpret_cast<const typename E::storage *>(table_grad), ids_dev, rows, exp_avg, exp_avg_sq, \
                     hidden, lr, beta1, beta2, eps, weight_decay, grad_scale, step, decay_lambda, target_norm)
  switch (table_dtype) {
    case LORA_AMD_F32: TI(f32_t); break;
    case LORA_AMD_F16: TI(f16_t); break;
    default: TI(bf16_t); break;
  }
#undef TI
  return check_launch("lora_amd_ti_rows_step");
}

__global__ void step_advance_kernel(int64_t *step) { step[0] += 1; }

extern "C" int lora_amd_step_advance(int64_t *step_dev, void *stream) {
  LORA_AMD_CHECK(step_dev != nullptr, LORA_AMD_EINVAL, "step_advance: null pointer");
  hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
  return check_launch("lora_amd_step_advance");
}
