// Preset-parameter losses and metrics on the device: the reference's SynthParamsLoss,
// QuantizedNumericalParamsLoss and CategoricalParamsAccuracy (model/loss.py:72-183, :187-261, :265-315) walk rows,
// parameters and one-hot groups in Python (one .item() per group).  Here a call is ONE launch:
//
//   params_loss_kernel     loss AND d loss / d u_out of SynthParamsLoss: a workgroup per row, a thread per item of the row
//                          (numerical column or one-hot group); the useless-parameter exclusion (data/preset.py:259-281)
//                          is a bit mask per row, the per-group row counts are recomputed by every workgroup from the
//                          masks in LDS (B x R compares), the loss leaves through per-workgroup float64 partials that the
//                          last-arriving workgroup adds up in a fixed order (deterministic, no clearing launch).
//   params_columns_kernel  the column pairs both metrics compare (quantised numerical values, arg-max classes) and the
//                          per-parameter match rate: a workgroup per VST parameter.
//
// These are tiny, latency-bound launches ([256, ~100..600] floats): the point is launch count and no host synchronisation,
// not bandwidth.
#include "pgv_common.h"

namespace {

constexpr int PL_THREADS = 256;

__global__ __launch_bounds__(PL_THREADS) void params_loss_kernel(const float* __restrict__ u_out, const float* __restrict__ u_in,
                                                                  int B, int L, pgv_params_tables t, int mode, float inv_t,
                                                                  float s_num, float s_cat, float* __restrict__ loss,
                                                                  float* __restrict__ grad, double* partial,
                                                                  unsigned* ticket) {
  extern __shared__ uint32_t sm[];
  uint32_t* offmask = sm;                               // [B]  bit r: rule r fires for the row (u_in[row][trig r] < 1e-3)
  float* wgt = reinterpret_cast<float*>(sm + B);        // [G]  weight of one row's term of group g
  __shared__ double red[16];
  __shared__ int is_last;
  const int tid = threadIdx.x;
  const int G = t.n_groups, K = t.K, R = t.n_rules;

  for (int b = tid; b < B; b += PL_THREADS) {
    uint32_t m = 0;
    for (int r = 0; r < R; ++r) m |= (u_in[(int64_t)b * L + t.rule_trig[r]] < 1e-3f ? 1u : 0u) << r;
    offmask[b] = m;
  }
  __syncthreads();
  double acc = 0.0;
  for (int g = tid; g < G; g += PL_THREADS) {
    const uint32_t gm = R ? t.cat_rules[g] : 0u;
    int n = B;
    if (gm) {
      n = 0;
      for (int b = 0; b < B; ++b) n += (offmask[b] & gm) == 0u;
    }
    float w = s_cat / (float)n;                         // loss.py:171: / (batch_size - len(rows_to_remove))
    if (mode == PGV_PARAMS_BCE) {                       // loss.py:174: mean over rows x classes, / 8
      int kg = 0;
      for (int k = 0; k < K; ++k) kg += t.cat_idx[g * K + k] >= 0;
      w /= 8.0f * (float)kg;
    }
    wgt[g] = w;
    // a group without a useful row: the reference divides 0 by 0 (sum over an empty selection / 0)
    if (n == 0 && blockIdx.x == 0) acc += (double)__builtin_nanf("");
  }
  __syncthreads();

  const int n_items = t.n_num + G;
  for (int row = blockIdx.x; row < B; row += gridDim.x) {
    const float* ro = u_out + (int64_t)row * L;
    const float* ri = u_in + (int64_t)row * L;
    float* rg = grad ? grad + (int64_t)row * L : nullptr;
    if (rg) {
      for (int c = tid; c < L; c += PL_THREADS) rg[c] = 0.f;
      __syncthreads();
    }
    const uint32_t m = offmask[row];
    for (int item = tid; item < n_items; item += PL_THREADS) {
      if (item < t.n_num) {                             // loss.py:128-136: both sides zeroed where useless
        const int c = t.num_idx[item];
        const bool useless = R && (m & t.num_rules[item]);
        const float d = useless ? 0.f : ro[c] - ri[c];
        acc += (double)(d * d) * (double)s_num;
        if (rg) rg[c] = 2.f * d * s_num;
        continue;
      }
      const int g = item - t.n_num;
      if (R && (m & t.cat_rules[g])) continue;          // loss.py:141-150: row removed for this group
      const int32_t* idx = t.cat_idx + g * K;
      const float w = wgt[g];
      if (mode == PGV_PARAMS_BCE) {                     // F.binary_cross_entropy (logs clamped at -100)
        for (int k = 0; k < K && idx[k] >= 0; ++k) {
          const float q = ro[idx[k]], p = ri[idx[k]];
          const float lq = fmaxf(logf(q), -100.f), l1q = fmaxf(log1pf(-q), -100.f);
          acc += (double)(-(p * lq + (1.f - p) * l1q)) * (double)w;
          if (rg) rg[idx[k]] = w * (q - p) / fmaxf((1.f - q) * q, 1e-12f);
        }
      } else if (mode == PGV_PARAMS_CCE_SOFTMAX) {      // loss.py:166-171 with the temperature softmax
        float mx = -__builtin_inff();
        for (int k = 0; k < K && idx[k] >= 0; ++k) mx = fmaxf(mx, ro[idx[k]] * inv_t);
        float sum = 0.f;
        for (int k = 0; k < K && idx[k] >= 0; ++k) sum += expf(ro[idx[k]] * inv_t - mx);
        int n_t = 0;
        for (int k = 0; k < K && idx[k] >= 0; ++k) {
          if (ri[idx[k]] != 0.f) {
            acc += (double)(-logf(expf(ro[idx[k]] * inv_t - mx) / sum)) * (double)w;
            ++n_t;
          }
        }
        if (rg)
          for (int k = 0; k < K && idx[k] >= 0; ++k) {
            const float s = expf(ro[idx[k]] * inv_t - mx) / sum;
            rg[idx[k]] = -w * inv_t * ((ri[idx[k]] != 0.f ? 1.f : 0.f) - (float)n_t * s);
          }
      } else {                                          // the network already outputs probabilities
        for (int k = 0; k < K && idx[k] >= 0; ++k) {
          if (ri[idx[k]] != 0.f) {
            const float q = ro[idx[k]];
            acc += (double)(-logf(q)) * (double)w;
            if (rg) rg[idx[k]] = -w / q;
          }
        }
      }
    }
  }

  const double tot = pgv_block_sum_d(acc, red);
  if (tid == 0) {
    __hip_atomic_store(&partial[blockIdx.x], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    is_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  double s = 0.0;
  for (int i = tid; i < (int)gridDim.x; i += PL_THREADS)
    s += __hip_atomic_load(&partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  s = pgv_block_sum_d(s, red);
  if (tid == 0) {
    loss[0] = (float)s;
    *ticket = 0u;                                       // ready for the next call / graph replay
  }
}

__device__ __forceinline__ int argmax_cols(const float* row, const int32_t* idx, int len) {
  int best = 0;
  float bv = row[idx[0]];
  for (int k = 1; k < len; ++k) {
    const float v = row[idx[k]];
    if (v > bv) bv = v, best = k;
  }
  return best;
}

__global__ __launch_bounds__(PL_THREADS) void params_columns_kernel(const float* __restrict__ u_out, const float* __restrict__ u_in,
                                                                     int B, int L, int n_items, const int32_t* __restrict__ kind,
                                                                     const int32_t* __restrict__ first, const int32_t* __restrict__ len,
                                                                     const float* __restrict__ card, const int32_t* __restrict__ idx,
                                                                     float* __restrict__ in_cols, float* __restrict__ out_cols,
                                                                     float* __restrict__ match) {
  __shared__ float red[16];
  const int item = blockIdx.x, k = kind[item], f = first[item], n = len[item];
  const float cm1 = card[item] - 1.0f;
  float eq = 0.f;
  for (int b = threadIdx.x; b < B; b += PL_THREADS) {
    const float* ri = u_in + (int64_t)b * L;
    const float* ro = u_out + (int64_t)b * L;
    float vi, vo;
    if (k == PGV_PARAMS_COL_QUANTIZED) {                // loss.py:231-240
      vi = ri[f];
      vo = ro[f];
      if (card[item] > 0.f) vo = rintf(vo * cm1) / cm1;
    } else if (k == PGV_PARAMS_COL_ONEHOT_VALUE) {      // loss.py:242-252
      const float d = (float)n - 1.0f;
      vi = (float)argmax_cols(ri, idx + f, n) / d;
      vo = (float)argmax_cols(ro, idx + f, n) / d;
    } else if (k == PGV_PARAMS_COL_CLASS) {             // loss.py:287-297
      vi = (float)(int)rintf(ri[f] * cm1);
      vo = (float)(int)rintf(ro[f] * cm1);
    } else {                                            // PGV_PARAMS_COL_ONEHOT_CLASS, loss.py:299-306
      vi = (float)argmax_cols(ri, idx + f, n);
      vo = (float)argmax_cols(ro, idx + f, n);
    }
    if (in_cols) in_cols[(int64_t)b * n_items + item] = vi;
    if (out_cols) out_cols[(int64_t)b * n_items + item] = vo;
    eq += vi == vo ? 1.f : 0.f;
  }
  const float tot = pgv_block_sum(eq, red);             // exact: counts <= 2^24
  if (threadIdx.x == 0 && match) match[item] = tot / (float)B;
}

}  // namespace

extern "C" {

int pgv_params_loss(const float* u_out, const float* u_in, int B, int L, const pgv_params_tables* t, int mode,
                    float softmax_t, int normalize, float cat_factor, float* loss, float* grad, void* workspace,
                    int64_t workspace_bytes, void* stream) {
  PGV_CHECK_ARG(u_out && u_in && t && loss && workspace && B > 0 && L > 0, "pgv_params_loss: bad argument");
  PGV_CHECK_ARG(mode >= PGV_PARAMS_CCE && mode <= PGV_PARAMS_BCE, "pgv_params_loss: unknown categorical mode %d", mode);
  PGV_CHECK_ARG(t->n_num >= 0 && t->n_groups >= 0 && t->n_rules >= 0 && t->n_rules <= 32 && (t->n_groups == 0 || t->K > 0),
                "pgv_params_loss: bad tables (at most 32 useless-parameter rules)");
  PGV_CHECK_ARG((t->n_num == 0 || t->num_idx) && (t->n_groups == 0 || t->cat_idx) &&
                    (t->n_rules == 0 || (t->rule_trig && (t->n_num == 0 || t->num_rules) && (t->n_groups == 0 || t->cat_rules))),
                "pgv_params_loss: null table");
  PGV_CHECK_ARG(mode != PGV_PARAMS_CCE_SOFTMAX || softmax_t > 0.f, "pgv_params_loss: softmax temperature must be > 0");
  const int grid = B < 1024 ? B : 1024;
  if (workspace_bytes < (int64_t)sizeof(double) * (1 + grid)) {
    pgv_set_error("pgv_params_loss: workspace of %lld bytes, %lld needed", (long long)workspace_bytes,
                  (long long)(sizeof(double) * (1 + grid)));
    return PGV_E_WORKSPACE;
  }
  const size_t lds = sizeof(uint32_t) * ((size_t)B + (size_t)t->n_groups);
  PGV_CHECK_ARG(lds <= 60 * 1024, "pgv_params_loss: B + groups = %d does not fit the LDS tables", B + t->n_groups);
  // loss.py:113-116,136: nn.MSELoss('mean') over [B, n_num] when normalised, else L2Loss = sum / B; :177-178: / groups
  const float s_num = t->n_num ? (normalize ? 1.0f / ((float)B * (float)t->n_num) : 1.0f / (float)B) : 0.f;
  const float s_cat = cat_factor * (normalize && t->n_groups ? 1.0f / (float)t->n_groups : 1.0f);
  unsigned* ticket = reinterpret_cast<unsigned*>(workspace);
  double* partial = reinterpret_cast<double*>(workspace) + 1;
  hipLaunchKernelGGL(params_loss_kernel, dim3(grid), dim3(PL_THREADS), lds, pgv_stream(stream), u_out, u_in, B, L, *t, mode,
                     mode == PGV_PARAMS_CCE_SOFTMAX ? 1.0f / softmax_t : 1.0f, s_num, s_cat, loss, grad, partial, ticket);
  PGV_CHECK_LAUNCH("pgv_params_loss");
  return PGV_OK;
}

int pgv_params_columns(const float* u_out, const float* u_in, int B, int L, int n_items, const int32_t* kind,
                       const int32_t* first, const int32_t* len, const float* card, const int32_t* idx, float* in_cols,
                       float* out_cols, float* match, void* stream) {
  PGV_CHECK_ARG(u_out && u_in && B > 0 && L > 0 && n_items >= 0, "pgv_params_columns: bad argument");
  if (n_items == 0) return PGV_OK;
  PGV_CHECK_ARG(kind && first && len && card, "pgv_params_columns: null item table");
  hipLaunchKernelGGL(params_columns_kernel, dim3(n_items), dim3(PL_THREADS), 0, pgv_stream(stream), u_out, u_in, B, L, n_items,
                     kind, first, len, card, idx, in_cols, out_cols, match);
  PGV_CHECK_LAUNCH("pgv_params_columns");
  return PGV_OK;
}

}  // extern "C"
