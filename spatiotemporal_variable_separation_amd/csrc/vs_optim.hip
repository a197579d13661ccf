// vs_optim.hip -- the Adam update of the training loop (reference: train.py:156-158 `optimizer.step()` with
// torch.optim.Adam(lr, betas), weight_decay = 0, amsgrad = False) for every parameter of the model in ONE launch.
//
// HBM-bound by construction: per parameter 16 B read (p, g, m, v) + 12 B written (p, m, v), optionally + 2 B for the bf16
// operand copy the next forward pass reads.  One launch walks a table of up to 64 tensors in 16 KiB chunks (a chunk never
// straddles two tensors), 16-byte vector accesses, every workgroup fully used whatever the mix of tensor sizes (the WaveEq
// model has two 24.6 M-element matrices next to 32-element biases).  The step count lives on the device, so the launch is
// recordable into a hipGraph.
#include "vs_common.h"
#include "vs_adam_math.h"
#include <math.h>

namespace {

constexpr int AD_MAXJ = 64;
constexpr int AD_CHUNK = 4096;          // elements per workgroup iteration

struct AdamJobs {
    float* p[AD_MAXJ];
    const void* g[AD_MAXJ];
    int g_bf16[AD_MAXJ];                // gradient stored as bf16 (the data-parallel wire image) instead of fp32
    float* m[AD_MAXJ];
    float* v[AD_MAXJ];
    unsigned short* shadow[AD_MAXJ];    // optional 16-bit (bf16 / fp16: `shadow_dtype`) copy of the updated parameter
    int shadow_dtype;
    long long n[AD_MAXJ];
    int aligned[AD_MAXJ];               // all four pointers 16-byte aligned (and the shadow 8-byte): vector path
    int skipped[AD_MAXJ];               // optimizer steps this tensor sat out (no gradient): its own step count lags the group's
    int chunk_off[AD_MAXJ + 1];         // prefix sum of chunk counts
    int nj;
};

int g_adam_max_blocks = 0;                 // vs_adam_set_max_blocks: grid cap of the next launches (0 = default)

// scale_state (optional, fp16 loss scaling): [0] = loss scale S (gradients arrive multiplied by S and are used as g / S),
// [1] = found_inf flag of this step (non-zero: leave everything untouched, GradScaler.step semantics)
__global__ __launch_bounds__(256) void adam_multi_kernel(AdamJobs J, const int* __restrict__ step, double lr_d, double beta1_d, double beta2_d,
                                                         float eps, const float* __restrict__ scale_state, const unsigned* __restrict__ guard) {
    if (guard && *guard != 0u) return;                        // a bounded exchange of this step timed out (vs_exchange_guard_set): no update
    float inv_scale = 1.f;
    if (scale_state) {
        if (scale_state[1] != 0.f) return;
        inv_scale = 1.f / scale_state[0];
    }
    // bias corrections from the device-side step count (the count is incremented by a separate 1-thread launch AFTER this one,
    // so every workgroup of this launch reads the same value)
    const int t_group = step[0] + 1;
    const int total = J.chunk_off[J.nj];
    for (int ch = blockIdx.x; ch < total; ch += gridDim.x) {
        int j = 0;                                            // uniform per workgroup: scalar search over <= 64 entries
        while (J.chunk_off[j + 1] <= ch) ++j;
        const AdamCoef c = vs_adam_coef(lr_d, beta1_d, beta2_d, eps, (double)(t_group - J.skipped[j]));
        const long long base = (long long)(ch - J.chunk_off[j]) * AD_CHUNK;
        const long long n = J.n[j];
        float* __restrict__ P = J.p[j];
        const float* __restrict__ G = (const float*)J.g[j];
        const __bf16* __restrict__ Gh = (const __bf16*)J.g[j];
        const bool gh = J.g_bf16[j] != 0;
        float* __restrict__ M = J.m[j];
        float* __restrict__ V = J.v[j];
        unsigned short* __restrict__ S = J.shadow[j];
        const int sdt = J.shadow_dtype;
#pragma unroll
        for (int u = 0; u < AD_CHUNK / 1024; ++u) {
            const long long i = base + (long long)u * 1024 + threadIdx.x * 4;
            if (i + 3 < n && J.aligned[j]) {
                f32x4 p = *reinterpret_cast<const f32x4*>(P + i), g;
                if (gh) {
                    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4g;
                    const bf16x4g t4 = *reinterpret_cast<const bf16x4g*>(Gh + i);
                    g = f32x4{(float)t4[0], (float)t4[1], (float)t4[2], (float)t4[3]};
                } else {
                    g = *reinterpret_cast<const f32x4*>(G + i);
                }
                f32x4 m = *reinterpret_cast<const f32x4*>(M + i), v = *reinterpret_cast<const f32x4*>(V + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (scale_state) g[e] *= inv_scale;                     // scaler.unscale_: grad * (1 / scale)
                    float pe = p[e], me = m[e], ve = v[e];
                    vs_adam_elem(c, g[e], pe, me, ve);
                    p[e] = pe; m[e] = me; v[e] = ve;
                }
                *reinterpret_cast<f32x4*>(P + i) = p;
                *reinterpret_cast<f32x4*>(M + i) = m;
                *reinterpret_cast<f32x4*>(V + i) = v;
                if (S) {
                    const u16x4 s = {vs_f2h(p[0], sdt), vs_f2h(p[1], sdt), vs_f2h(p[2], sdt), vs_f2h(p[3], sdt)};
                    *reinterpret_cast<u16x4*>(S + i) = s;
                }
            } else {
                for (long long k = i; k < n && k < i + 4; ++k) {
                    float g = gh ? (float)Gh[k] : G[k];
                    if (scale_state) g *= inv_scale;
                    float pe = P[k], me = M[k], ve = V[k];
                    vs_adam_elem(c, g, pe, me, ve);
                    P[k] = pe; M[k] = me; V[k] = ve;
                    if (S) S[k] = vs_f2h(pe, sdt);
                }
            }
        }
    }
}

__global__ void step_increment_kernel(int* step, const float* scale_state, const unsigned* guard, unsigned* skips) {
    if (guard && *guard != 0u) {                              // exchange time-out: the step did not happen -- counted, so that the host can say how many
        if (skips) skips[0] += 1u;
        return;
    }
    if (scale_state && scale_state[1] != 0.f) return;         // overflow step: skipped, the step count does not advance
    step[0] += 1;
}

// ---- fp16 loss scaling (reference train.py:96-97, 151-155: torch.cuda.amp.GradScaler) on the device ---------------------------
// found_inf |= any non-finite element of any listed gradient tensor (fp32 or a 16-bit wire image)
constexpr int CF_MAXJ = 64;
struct FiniteJobs {
    const void* g[CF_MAXJ]; int dt[CF_MAXJ]; long long n[CF_MAXJ]; int chunk_off[CF_MAXJ + 1]; int nj;
};
__global__ __launch_bounds__(256) void check_finite_kernel(FiniteJobs J, float* found_inf) {
    const int total = J.chunk_off[J.nj];
    bool bad = false;
    for (int ch = blockIdx.x; ch < total; ch += gridDim.x) {
        int j = 0;
        while (J.chunk_off[j + 1] <= ch) ++j;
        const long long base = (long long)(ch - J.chunk_off[j]) * AD_CHUNK, n = J.n[j];
        const int dt = J.dt[j];
        const bool vec = ((uintptr_t)J.g[j] % 16) == 0;
#pragma unroll
        for (int u = 0; u < AD_CHUNK / 1024; ++u) {
            const long long i = base + (long long)u * 1024 + threadIdx.x * 4;
            if (i + 3 < n && vec && dt == VS_F32) {
                const u32x4 v = *reinterpret_cast<const u32x4*>((const float*)J.g[j] + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) bad |= (v[e] & 0x7f800000u) == 0x7f800000u;      // exponent all ones: inf or NaN
            } else {
                for (long long k = i; k < n && k < i + 4; ++k) {
                    const float f = vs_ld(J.g[j], dt, k);
                    bad |= (__float_as_uint(f) & 0x7f800000u) == 0x7f800000u;
                }
            }
        }
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) found_inf[0] = 1.f;       // benign race: every writer stores the same value
}

// state = [scale, found_inf, growth_tracker, skipped_steps]: GradScaler.update()
__global__ void loss_scale_update_kernel(float* state, float growth, float backoff, int interval) {
    if (state[1] != 0.f) {
        state[0] *= backoff;
        state[2] = 0.f;
        state[3] += 1.f;
    } else {
        const float t = state[2] + 1.f;
        if (t >= (float)interval) { state[0] *= growth; state[2] = 0.f; }
        else state[2] = t;
    }
    state[1] = 0.f;
}

}  // namespace

extern "C" int vs_adam_multi_scaled(int n_tensors, float* const* params, const void* const* grads, const int32_t* grad_dtype, float* const* exp_avg,
                                    float* const* exp_avg_sq, void* const* shadow, int shadow_dtype, const int64_t* numel, const int32_t* skipped,
                                    int32_t* step, double lr, double beta1, double beta2, double eps, const float* scale_state, void* stream) {
    VS_CHECK_ARG(!shadow || vs_is16(shadow_dtype), "vs_adam_multi: the operand copies are 16-bit (VS_BF16 | VS_F16)");
    VS_CHECK_ARG(n_tensors >= 1 && n_tensors <= AD_MAXJ && params && grads && exp_avg && exp_avg_sq && numel && step,
                 "vs_adam_multi: bad argument (1..%d tensors)", AD_MAXJ);
    VS_CHECK_ARG(lr > 0.0 && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps > 0.0, "vs_adam_multi: bad hyper-parameter");
    AdamJobs J;
    J.nj = n_tensors;
    J.chunk_off[0] = 0;
    for (int j = 0; j < n_tensors; ++j) {
        VS_CHECK_ARG(params[j] && grads[j] && exp_avg[j] && exp_avg_sq[j] && numel[j] > 0, "vs_adam_multi: bad tensor %d", j);
        const int gd = grad_dtype ? grad_dtype[j] : VS_F32;
        VS_CHECK_ARG(gd == VS_F32 || gd == VS_BF16, "vs_adam_multi: bad gradient dtype of tensor %d", j);
        VS_CHECK_ARG(((uintptr_t)params[j] | (uintptr_t)exp_avg[j] | (uintptr_t)exp_avg_sq[j]) % 4 == 0 && (uintptr_t)grads[j] % (gd == VS_F32 ? 4 : 2) == 0,
                     "vs_adam_multi: tensor %d is not aligned to its element size", j);
        J.g_bf16[j] = gd == VS_BF16;
        J.p[j] = params[j]; J.g[j] = grads[j]; J.m[j] = exp_avg[j]; J.v[j] = exp_avg_sq[j];
        J.shadow[j] = shadow ? (unsigned short*)shadow[j] : nullptr;
        J.aligned[j] = ((uintptr_t)params[j] | (uintptr_t)exp_avg[j] | (uintptr_t)exp_avg_sq[j]) % 16 == 0 &&
                       (uintptr_t)grads[j] % (gd == VS_F32 ? 16 : 8) == 0 && (!J.shadow[j] || (uintptr_t)J.shadow[j] % 8 == 0);
        J.n[j] = numel[j];
        J.skipped[j] = skipped ? skipped[j] : 0;
        const int64_t chunks = (numel[j] + AD_CHUNK - 1) / AD_CHUNK;
        VS_CHECK_ARG(J.chunk_off[j] + chunks < (1ll << 30), "vs_adam_multi: too many elements");
        J.chunk_off[j + 1] = J.chunk_off[j] + (int)chunks;
    }
    J.shadow_dtype = shadow ? shadow_dtype : VS_BF16;
    int blocks = J.chunk_off[n_tensors];
    const int cap = g_adam_max_blocks > 0 ? g_adam_max_blocks : 256 * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, J, step, lr, beta1, beta2, (float)eps, scale_state,
                       (const unsigned*)vs_g_exchange_guard);
    VS_CHECK_LAUNCH("vs_adam_multi");
    return VS_OK;
}

extern "C" int vs_adam_set_max_blocks(int blocks) {
    const int prev = g_adam_max_blocks;
    g_adam_max_blocks = blocks > 0 ? blocks : 0;
    return prev;
}

extern "C" int vs_adam_multi(int n_tensors, float* const* params, const void* const* grads, const int32_t* grad_dtype, float* const* exp_avg,
                             float* const* exp_avg_sq, void* const* shadow_bf16, const int64_t* numel, const int32_t* skipped, int32_t* step, double lr,
                             double beta1, double beta2, double eps, void* stream) {
    return vs_adam_multi_scaled(n_tensors, params, grads, grad_dtype, exp_avg, exp_avg_sq, shadow_bf16, VS_BF16, numel, skipped, step, lr, beta1, beta2,
                                eps, nullptr, stream);
}

extern "C" int vs_adam_step_increment(int32_t* step, void* stream) {
    VS_CHECK_ARG(step, "vs_adam_step_increment: null pointer");
    hipLaunchKernelGGL(step_increment_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step, (const float*)nullptr, (const unsigned*)vs_g_exchange_guard, vs_g_exchange_skips);
    VS_CHECK_LAUNCH("vs_adam_step_increment");
    return VS_OK;
}

extern "C" int vs_adam_step_increment_scaled(int32_t* step, const float* scale_state, void* stream) {
    VS_CHECK_ARG(step, "vs_adam_step_increment_scaled: null pointer");
    hipLaunchKernelGGL(step_increment_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step, scale_state, (const unsigned*)vs_g_exchange_guard, vs_g_exchange_skips);
    VS_CHECK_LAUNCH("vs_adam_step_increment_scaled");
    return VS_OK;
}

extern "C" int vs_check_finite_multi(int n_tensors, const void* const* grads, const int32_t* grad_dtype, const int64_t* numel, float* found_inf,
                                     void* stream) {
    VS_CHECK_ARG(n_tensors >= 1 && n_tensors <= CF_MAXJ && grads && numel && found_inf, "vs_check_finite_multi: bad argument (1..%d tensors)", CF_MAXJ);
    FiniteJobs J;
    J.nj = n_tensors;
    J.chunk_off[0] = 0;
    for (int j = 0; j < n_tensors; ++j) {
        const int gd = grad_dtype ? grad_dtype[j] : VS_F32;
        VS_CHECK_ARG(grads[j] && numel[j] > 0 && vs_dtype_ok(gd), "vs_check_finite_multi: bad tensor %d", j);
        J.g[j] = grads[j]; J.dt[j] = gd; J.n[j] = numel[j];
        const int64_t chunks = (numel[j] + AD_CHUNK - 1) / AD_CHUNK;
        VS_CHECK_ARG(J.chunk_off[j] + chunks < (1ll << 30), "vs_check_finite_multi: too many elements");
        J.chunk_off[j + 1] = J.chunk_off[j] + (int)chunks;
    }
    int blocks = J.chunk_off[n_tensors];
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(check_finite_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, J, found_inf);
    VS_CHECK_LAUNCH("vs_check_finite_multi");
    return VS_OK;
}

extern "C" int vs_loss_scale_update(float* state, float growth_factor, float backoff_factor, int growth_interval, void* stream) {
    VS_CHECK_ARG(state && growth_factor >= 1.f && backoff_factor > 0.f && backoff_factor <= 1.f && growth_interval >= 1, "vs_loss_scale_update: bad argument");
    hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, growth_factor, backoff_factor, growth_interval);
    VS_CHECK_LAUNCH("vs_loss_scale_update");
    return VS_OK;
}
