// vs_loss.h -- what the fused training losses (vs_eltwise.hip: vs_train_losses_*) and the decoder GEMM with the frame losses in its
// epilogue (vs_gemm.hip: vs_gemm_frame_loss) share: the argument block, the target-frame rule, the scalar assembly.
// Reference: train.py:117-149 (the four terms), train.py:85-86 / :139 (the two frame MSEs).
#pragma once
#include "vs_common.h"

namespace {

struct LossArgs {
    const float* frames; const float* full; const int* idx;
    const int* t_dev; int ae_shift, first_forecast;           // idx == NULL: frame 0 <-> full[:, t_dev[0] - ae_shift], frame g <-> first_forecast + g - 1
    int64_t rows, D; int G, T;
    const float* s_old; const float* s_new; int64_t n_s;      // spatial codes (n_s = 0: no spatial term)
    const float* t0; int64_t Bt, Ct;                           // initial temporal code [Bt, Ct]
    float l_ae, l_s, l_pred, l_t;
    float inv_ae, inv_pred, inv_s, inv_t;                      // 1/N of each mean
};

__device__ __forceinline__ int loss_target_frame(const LossArgs& a, int g) {
    if (a.idx) return a.idx[g];
    return g == 0 ? a.t_dev[0] - a.ae_shift : a.first_forecast + g - 1;
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

constexpr int VS_LOSS_MAX_PARTIALS = 4096;          // vs_train_losses_fwd_grad: `out` holds 16 + 2 * 4096 floats
struct LossGrads { const float* g; void* dz; int dz_dtype; int act; float* ds_old; float* ds_new; float* dt0; };


// The two code terms (train.py:120-122 zero-order loss, :141-149 t_reg) and their gradients: tiny, one workgroup, fixed order.
// Returns (sum (s_old - s_new)^2, sum t0^2) on every thread.
template <bool GRAD>
__device__ __forceinline__ void loss_code_terms(const LossArgs& a, const LossGrads& gr, float up, float* red, float& ss_out, float& st_out) {
    float ss = 0.f, st = 0.f;
    const float cs = up * a.l_s * 2.f * a.inv_s, ct = up * a.l_t * a.inv_t;
    for (int64_t i = threadIdx.x; i < a.n_s; i += 256) {
        const float d = a.s_old[i] - a.s_new[i];
        ss += d * d;
        if constexpr (GRAD) { gr.ds_old[i] = cs * d; gr.ds_new[i] = -(cs * d); }
    }
    for (int64_t i = threadIdx.x; i < a.Bt * a.Ct; i += 256) {
        const float v = a.t0[i];
        st += v * v;
        if constexpr (GRAD) gr.dt0[i] = ct * v;
    }
    ss_out = block_sum_256(ss, red);
    st_out = block_sum_256(st, red);
}

// A device-scope release fence per workgroup (the "last ticket assembles" pattern) costs more than this whole kernel on
// gfx950 (each fence writes back the XCD's L2: +45 us at 1024 workgroups), so the scalars are assembled by a 1-thread launch.
// (partials > 0: the frame sums arrive as that many per-workgroup pairs at out[16 ..])
__global__ __launch_bounds__(256) void train_losses_finalize_kernel(LossArgs a, float* out, int partials) {
    if (partials > 0) {
        __shared__ float red[4];
        float s0 = 0.f, s1 = 0.f;
        for (int i = threadIdx.x; i < partials; i += 256) { s0 += out[16 + 2 * i]; s1 += out[17 + 2 * i]; }
        s0 = block_sum_256(s0, red);
        s1 = block_sum_256(s1, red);
        if (threadIdx.x == 0) { out[0] = s0; out[1] = s1; }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const float ae = out[0] * a.inv_ae, pred = out[1] * a.inv_pred, zero = out[2] * a.inv_s, treg = 0.5f * out[3] * a.inv_t;
    out[5] = ae; out[6] = zero; out[7] = pred; out[8] = treg;
    out[4] = a.l_ae * ae + a.l_s * zero + a.l_pred * pred + a.l_t * treg;          // same association as train.py:146-149
}


// Scalar assembly for vs_gemm_frame_loss: the frame sums arrive as `partials` per-workgroup pairs at out[16 ..] (written by the GEMM's
// epilogue), the two code terms and their gradients are computed HERE (one workgroup) -- no zero fill, no atomics, reproducible.
__global__ __launch_bounds__(256) void frame_loss_finish_kernel(LossArgs a, float* out, int partials, LossGrads gr) {
    __shared__ float red[4];
    float s0 = 0.f, s1 = 0.f;
    for (int i = threadIdx.x; i < partials; i += 256) { s0 += out[16 + 2 * i]; s1 += out[17 + 2 * i]; }
    s0 = block_sum_256(s0, red);
    s1 = block_sum_256(s1, red);
    float ss, st;
    loss_code_terms<true>(a, gr, gr.g[0], red, ss, st);
    if (threadIdx.x != 0) return;
    out[0] = s0; out[1] = s1; out[2] = ss; out[3] = st;
    const float ae = s0 * a.inv_ae, pred = s1 * a.inv_pred, zero = ss * a.inv_s, treg = 0.5f * st * a.inv_t;
    out[5] = ae; out[6] = zero; out[7] = pred; out[8] = treg;
    out[4] = a.l_ae * ae + a.l_s * zero + a.l_pred * pred + a.l_t * treg;          // same association as train.py:146-149
}

// `frames` may be NULL for vs_gemm_frame_loss (they are never stored)
int fill_loss_args(LossArgs& a, const float* frames, const float* full, const int32_t* idx, const int32_t* t_dev, int ae_shift, int first_forecast,
                   int64_t B, int G, int T, int64_t D, const float* s_old, const float* s_new, int64_t n_s, const float* t0, int64_t Bt, int64_t Ct,
                   int average_tloss, const float* lambdas) {
    a.t_dev = t_dev; a.ae_shift = ae_shift; a.first_forecast = first_forecast;
    VS_CHECK_ARG(full && (idx || t_dev) && t0 && lambdas && B > 0 && G >= 1 && T > 0 && D > 0 && Bt > 0 && Ct > 0 && n_s >= 0,
                 "vs_train_losses: bad argument");
    VS_CHECK_ARG(n_s == 0 || (s_old && s_new), "vs_train_losses: spatial codes missing");
    a.frames = frames; a.full = full; a.idx = idx; a.rows = B * G; a.D = D; a.G = G; a.T = T;
    a.s_old = s_old; a.s_new = s_new; a.n_s = n_s; a.t0 = t0; a.Bt = Bt; a.Ct = Ct;
    a.l_ae = lambdas[0]; a.l_s = lambdas[1]; a.l_t = lambdas[2]; a.l_pred = lambdas[3];
    a.inv_ae = (float)(1.0 / ((double)B * D));
    a.inv_pred = (float)(1.0 / ((double)B * (G > 1 ? G - 1 : 1) * D));
    a.inv_s = n_s > 0 ? (float)(1.0 / (double)n_s) : 0.f;
    a.inv_t = (float)(1.0 / (average_tloss ? (double)Bt * Ct : (double)Bt));
    return VS_OK;
}
}  // namespace
