// vs_norm.hip -- the HBM-bound layers between the convolutions: BatchNorm2d (training statistics, fused affine +
// activation, backward), per-channel sums (conv bias gradient), MaxPool2d(2,2), nearest Upsample(x2).
//
// Reference: conv.py:41-60 (conv -> BatchNorm2d -> activation blocks), MaxPool2d conv.py:151-169,330-335,
// Upsample conv.py:296-314,371-377,406-413.  BatchNorm statistics are per CALL (training mode, eps 1e-5, momentum 0.1,
// biased variance for normalisation, unbiased for the running estimate), exactly nn.BatchNorm2d.
//
// Kernel shapes: statistics/reductions use one 256-thread workgroup per channel (C = 64..512 workgroups; the batch*HW
// extent of a channel is streamed with coalesced reads, reduced in registers -> LDS, no atomics -> reproducible);
// element-wise passes are grid-stride with the channel recovered from the flat NCHW index.
#include "vs_common.h"
#include <stdlib.h>

namespace {

// per-channel reductions accumulate in fp64 (like ATen's CPU BatchNorm): they are HBM-bound, the adds are free, and the
// statistics feed every element of the layer, so their rounding noise is amplified by deep BatchNorm stacks
__device__ __forceinline__ double block_sum(double v, double* red) {
    // blockDim.x = 256 or 1024 threads = 4 or 16 waves of 64; `red` holds 16 doubles
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < nw; ++w) t += red[w];          // fixed order: reproducible
    return t;
}

// derivative of the activation from its PRE-activation input z
__device__ __forceinline__ float act_grad_from_pre(float z, int act) {
    switch (act) {
        case VS_ACT_RELU: return z > 0.f ? 1.f : 0.f;
        case VS_ACT_LEAKY: return z > 0.f ? 1.f : 0.2f;
        case VS_ACT_SIGMOID: { const float s = 1.f / (1.f + expf(-z)); return s * (1.f - s); }
        case VS_ACT_TANH: { const float t = tanhf(z); return 1.f - t * t; }
        case VS_ACT_ELU: return z > 0.f ? 1.f : expf(z);
        default: return 1.f;
    }
}

// Statistics per (group, channel): the batch is G consecutive groups of Bg samples, each group normalised on its own --
// one reference CALL per group (the decoder is invoked once per frame with per-call statistics; batching
// the calls over time is exact when every call keeps its own mean/variance and the running estimates are folded in
// call order, which bn_running_kernel does).
// 8 (bf16) / 4 (fp32) consecutive elements of one (sample, channel) plane per 16-byte load
__device__ __forceinline__ int ld_vec(const void* x, int xd, int64_t idx, float (&v)[8]) {
    if (xd == VS_F32) {
        const f32x4 t = *reinterpret_cast<const f32x4*>((const float*)x + idx);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
        return 4;
    }
    const u16x8 t = *reinterpret_cast<const u16x8*>((const unsigned short*)x + idx);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = vs_h2f(t[j], xd);
    return 8;
}
__device__ __forceinline__ void st_vec(void* y, int yd, int64_t idx, const float (&v)[8], int n) {
    if (yd == VS_F32) {
        for (int o = 0; o < n; o += 4) {
            f32x4 t = {v[o], v[o + 1], v[o + 2], v[o + 3]};
            *reinterpret_cast<f32x4*>((float*)y + idx + o) = t;
        }
    } else if (n == 8) {
        u16x8 t;
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = vs_f2h(v[j], yd);
        *reinterpret_cast<u16x8*>((unsigned short*)y + idx) = t;
    } else {
        const u16x4 t = {vs_f2h(v[0], yd), vs_f2h(v[1], yd), vs_f2h(v[2], yd), vs_f2h(v[3], yd)};
        *reinterpret_cast<u16x4*>((unsigned short*)y + idx) = t;
    }
}

// channel c and (group, channel) slot of the flat NCHW element index i; 32-bit divisions whenever the tensor has < 2^31
// elements (three 64-bit divisions per 16-byte vector cost as much as the payload arithmetic)
__device__ __forceinline__ void chan_of(int64_t i, int64_t HW, int C, int64_t group_elems, bool small, int& c, int& gc) {
    if (small) {
        const uint32_t u = (uint32_t)i;
        c = (int)((u / (uint32_t)HW) % (uint32_t)C);
        gc = (int)(u / (uint32_t)group_elems) * C + c;
    } else {
        c = (int)((i / HW) % C);
        gc = (int)(i / group_elems) * C + c;
    }
}

// ---- planes whose size is not a multiple of the vector width (17x17, 9x9, 33x33 maps of the chairs ResNet18) ----------------------------
// vec == 2: a plane is cut into ceil(HW / w) units of w consecutive elements; all but the last are 16-byte accesses at element
// (not 16-byte) alignment -- global memory takes them unaligned --, the last one is a short scalar tail.  Unit u of a tensor with
// P planes: plane = u / upp, first element = (u % upp) * w.
// w = elements per unit: 8 when every tensor of the pass is bf16, 4 as soon as one is fp32 (a 4-element bf16 access is 8 bytes)
__device__ __forceinline__ int ld_unit(const void* x, int xd, int64_t idx, int left, int w, float (&v)[8]) {
    if (left >= w) {
        if (xd == VS_F32) {
            const f32x4 t = *reinterpret_cast<const f32x4*>((const float*)x + idx);
            v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
        } else if (w == 8) {
            const u16x8 t = *reinterpret_cast<const u16x8*>((const unsigned short*)x + idx);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = vs_h2f(t[j], xd);
        } else {
            const u16x4 t = *reinterpret_cast<const u16x4*>((const unsigned short*)x + idx);
            v[0] = vs_h2f(t[0], xd); v[1] = vs_h2f(t[1], xd); v[2] = vs_h2f(t[2], xd); v[3] = vs_h2f(t[3], xd);
        }
        return w;
    }
    for (int j = 0; j < left; ++j) v[j] = vs_ld(x, xd, idx + j);
    return left;
}
__device__ __forceinline__ void st_unit(void* y, int yd, int64_t idx, const float (&v)[8], int cnt, int w) {
    if (cnt == w) { st_vec(y, yd, idx, v, cnt); return; }
    for (int j = 0; j < cnt; ++j) vs_st(y, yd, idx + j, v[j]);
}
__device__ __forceinline__ int unit_width(int d0, int d1, int d2) { return (d0 == VS_F32 || d1 == VS_F32 || d2 == VS_F32) ? 4 : 8; }

// ---- hot loops of the 16-byte vector paths, element count per vector as a template constant --------------------------------------
template <int W>
__device__ __forceinline__ void ld_vecw(const void* x, int xd, int64_t idx, float (&v)[W]) {
    if constexpr (W == 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>((const float*)x + idx);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    } else {
        const u16x8 t = *reinterpret_cast<const u16x8*>((const unsigned short*)x + idx);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = vs_h2f(t[j], xd);
    }
}
template <int W>
__device__ __forceinline__ void st_vecw(void* y, int yd, int64_t idx, const float (&v)[W]) {
    if (yd == VS_F32) {
#pragma unroll
        for (int o = 0; o < W; o += 4) {
            f32x4 t = {v[o], v[o + 1], v[o + 2], v[o + 3]};
            *reinterpret_cast<f32x4*>((float*)y + idx + o) = t;
        }
    } else if constexpr (W == 8) {
        u16x8 t;
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = vs_f2h(v[j], yd);
        *reinterpret_cast<u16x8*>((unsigned short*)y + idx) = t;
    } else {
        const u16x4 t = {vs_f2h(v[0], yd), vs_f2h(v[1], yd), vs_f2h(v[2], yd), vs_f2h(v[3], yd)};
        *reinterpret_cast<u16x4*>((unsigned short*)y + idx) = t;
    }
}

template <int W>
__device__ __forceinline__ void bn_fwd_vec_loop(const void* x, int xd, void* y, int yd, const float* mean, const float* invstd, const float* gamma,
                                                const float* beta, int act, int C, int64_t HW, int64_t nv, int64_t group_elems, bool small) {
    for (int64_t iv0 = (int64_t)blockIdx.x * 1024 + threadIdx.x; iv0 < nv; iv0 += (int64_t)gridDim.x * 1024) {
        float v[4][W];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (iv0 + k * 256 < nv) ld_vecw<W>(x, xd, (iv0 + k * 256) * W, v[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t i = (iv0 + k * 256) * W;
            if (iv0 + k * 256 < nv) {
                int c, gc;
                chan_of(i, HW, C, group_elems, small, c, gc);
                const float mu = mean[gc], is = invstd[gc], g = gamma[c], bt = beta[c];
#pragma unroll
                for (int j = 0; j < W; ++j) v[k][j] = vs_act((v[k][j] - mu) * is * g + bt, act);
                st_vecw<W>(y, yd, i, v[k]);
            }
        }
    }
}

template <int W>
__device__ __forceinline__ void bn_bwd_apply_vec_loop(const void* dy, const void* x, int xd, const float* mean, const float* invstd, const float* gamma,
                                                      const float* beta, int act, const float* sum_dz, const float* sum_dz_xhat, void* dx, int dxd,
                                                      int C, int64_t HW, int64_t nv, int64_t group_elems, bool small, float inv_n, int training) {
    for (int64_t iv0 = (int64_t)blockIdx.x * 512 + threadIdx.x; iv0 < nv; iv0 += (int64_t)gridDim.x * 512) {
        float xv[2][W], gv[2][W];                                    // two vectors of each tensor in flight per thread
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (iv0 + k * 256 < nv) { ld_vecw<W>(x, xd, (iv0 + k * 256) * W, xv[k]); ld_vecw<W>(dy, xd, (iv0 + k * 256) * W, gv[k]); }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int64_t i = (iv0 + k * 256) * W;
            if (iv0 + k * 256 < nv) {
                int c, gc;
                chan_of(i, HW, C, group_elems, small, c, gc);
                const float mu = mean[gc], is = invstd[gc], g = gamma[c], bt = beta[c];
                const float k1 = sum_dz[gc] * inv_n, k2 = sum_dz_xhat[gc] * inv_n;
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    const float xh = (xv[k][j] - mu) * is;
                    const float dz = gv[k][j] * act_grad_from_pre(xh * g + bt, act);
                    xv[k][j] = training ? g * is * (dz - k1 - xh * k2) : g * is * dz;
                }
                st_vecw<W>(dx, dxd, i, xv[k]);
            }
        }
    }
}

template <int W>
__device__ __forceinline__ void bn_bwd_reduce_vec_loop(const void* dy, const void* x, int xd, float mu, float is, float g, float bt, int act, int64_t b0,
                                                       int c, int C, int64_t HW, uint32_t per32, uint32_t nvec, double& s1, double& s2) {
    for (uint32_t i0 = threadIdx.x; i0 < nvec; i0 += 2 * blockDim.x) {
        float xv[2][W], gv[2][W];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t i = i0 + k * blockDim.x;
            if (i < nvec) {
                const uint32_t b = i / per32, p = (i - b * per32) * W;
                const int64_t idx = ((b0 + b) * C + c) * HW + p;
                ld_vecw<W>(x, xd, idx, xv[k]);
                ld_vecw<W>(dy, xd, idx, gv[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (i0 + k * blockDim.x < nvec) {
                float a1 = 0.f, a2 = 0.f;                            // fp32 over the elements of a vector, fp64 across vectors
#pragma unroll
                for (int j = 0; j < W; ++j) {
                    const float xh = (xv[k][j] - mu) * is;
                    const float dz = gv[k][j] * act_grad_from_pre(xh * g + bt, act);
                    a1 += dz;
                    a2 += dz * xh;
                }
                s1 += (double)a1;
                s2 += (double)a2;
            }
        }
    }
}

__global__ __launch_bounds__(1024) void bn_stats_kernel(const void* x, int xd, int Bg, int C, int64_t HW, float* mean, float* invstd,
                                                       float* ubvar, float eps, int vec) {
    __shared__ double red[16];
    const int c = blockIdx.x, g = blockIdx.y;
    const int64_t n = (int64_t)Bg * HW;
    const int64_t b0 = (int64_t)g * Bg;
    // single pass, fp64 sum and sum of squares: exact enough (53-bit accumulation of fp32/bf16 data) and half the HBM traffic
    double s = 0.0, q = 0.0;
    if (vec == 2) {
        const int w = xd == VS_F32 ? 4 : 8;
        const uint32_t upp = (uint32_t)((HW + w - 1) / w), nunit = (uint32_t)Bg * upp;
        for (uint32_t i = threadIdx.x; i < nunit; i += blockDim.x) {
            const uint32_t b = i / upp, e0 = (i - b * upp) * w;
            float v[8];
            const int cnt = ld_unit(x, xd, ((b0 + b) * C + c) * HW + e0, (int)(HW - e0), w, v);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < cnt) { s += (double)v[j]; q += (double)v[j] * (double)v[j]; }
        }
    } else if (vec) {
        const int w = xd == VS_F32 ? 4 : 8;
        const int64_t per = HW / w;                                  // vectors per plane
        const uint32_t per32 = (uint32_t)per, nvec = (uint32_t)Bg * per32;           // < 2^31: a channel of one call group
        for (uint32_t i = threadIdx.x; i < nvec; i += blockDim.x) {
            const uint32_t b = i / per32, p = (i - b * per32) * w;
            float v[8];
            const int cnt = ld_vec(x, xd, ((b0 + b) * C + c) * HW + p, v);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < cnt) { s += (double)v[j]; q += (double)v[j] * (double)v[j]; }
        }
    } else {
        for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
            const int64_t b = i / HW, p = i - b * HW;
            const double v = (double)vs_ld(x, xd, ((b0 + b) * C + c) * HW + p);
            s += v; q += v * v;
        }
    }
    const double ts = block_sum(s, red);
    const double tq = block_sum(q, red);
    if (threadIdx.x == 0) {
        const double mu = ts / (double)n;
        double ss = tq - ts * mu;                                    // sum (x - mu)^2
        if (ss < 0.0) ss = 0.0;
        const double var = ss / (double)n;
        mean[g * C + c] = (float)mu;
        invstd[g * C + c] = (float)(1.0 / sqrt(var + (double)eps));
        if (ubvar) ubvar[g * C + c] = (float)(n > 1 ? ss / (double)(n - 1) : var);
    }
}

__global__ __launch_bounds__(256) void bn_running_kernel(const float* mean, const float* ubvar, int G, int C, float* rmean, float* rvar,
                                                         float momentum) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double rm = rmean[c], rv = rvar[c];
    for (int g = 0; g < G; ++g) {                       // sequential, in call order
        rm = (1.0 - momentum) * rm + momentum * (double)mean[g * C + c];
        rm = (double)(float)rm;                         // the reference rounds to fp32 after every call
        rv = (1.0 - momentum) * rv + momentum * (double)ubvar[g * C + c];
        rv = (double)(float)rv;
    }
    rmean[c] = (float)rm;
    rvar[c] = (float)rv;
}

__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const void* x, int xd, void* y, int yd, const float* mean, const float* invstd,
                                                         const float* gamma, const float* beta, int act, int C, int64_t HW, int64_t total,
                                                         int64_t group_elems, int vec, const float* ubvar = nullptr, float* rmean = nullptr,
                                                         float* rvar = nullptr, float momentum = 0.f, int G = 1) {
    if (rmean && blockIdx.x == gridDim.x - 1) {
        // the running estimates of the statistics this pass applies, folded in call order (bn_running_kernel's arithmetic) by the last
        // workgroup: one launch less per BatchNorm call (a one-workgroup kernel lasts ~5 us inside the replayed step)
        for (int c = threadIdx.x; c < C; c += 256) {
            double rm = rmean[c], rv = rvar[c];
            for (int g = 0; g < G; ++g) {
                rm = (double)(float)((1.0 - momentum) * rm + momentum * (double)mean[g * C + c]);
                rv = (double)(float)((1.0 - momentum) * rv + momentum * (double)ubvar[g * C + c]);
            }
            rmean[c] = (float)rm;
            rvar[c] = (float)rv;
        }
    }
    if (vec == 2) {
        const int w = unit_width(xd, yd, VS_BF16);
        const uint32_t upp = (uint32_t)((HW + w - 1) / w);
        const int64_t planes = total / HW, nunit = planes * upp;
        const uint32_t ppg = (uint32_t)(group_elems / HW);           // planes per call group
        for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < nunit; u += (int64_t)gridDim.x * 256) {
            const uint32_t plane = (uint32_t)(u / upp), e0 = (uint32_t)(u - (int64_t)plane * upp) * w;
            const int c = (int)(plane % (uint32_t)C), gc = (int)(plane / ppg) * C + c;
            const float mu = mean[gc], is = invstd[gc], g = gamma[c], bt = beta[c];
            const int64_t i = (int64_t)plane * HW + e0;
            float v[8];
            const int cnt = ld_unit(x, xd, i, (int)(HW - e0), w, v);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < cnt) v[j] = vs_act((v[j] - mu) * is * g + bt, act);
            st_unit(y, yd, i, v, cnt, w);
        }
        return;
    }
    if (vec) {                                                       // 16-byte chunks never straddle a (sample, channel) plane
        const int w = xd == VS_F32 ? 4 : 8;
        const int64_t nv = total / w;
        const bool small = total < (int64_t)1 << 31;
        // four 16-byte vectors in flight per thread before any of the per-vector index arithmetic (two divisions, four table
        // look-ups): with one vector per iteration the pass ran at 2.3-3.5 TB/s.  (W is a template constant: with a run-time element
        // count the register arrays go to scratch.)
        if (xd == VS_F32) bn_fwd_vec_loop<4>(x, xd, y, yd, mean, invstd, gamma, beta, act, C, HW, nv, group_elems, small);
        else bn_fwd_vec_loop<8>(x, xd, y, yd, mean, invstd, gamma, beta, act, C, HW, nv, group_elems, small);
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)((i / HW) % C);
        const int gc = (int)(i / group_elems) * C + c;
        const float xh = (vs_ld(x, xd, i) - mean[gc]) * invstd[gc];
        vs_st(y, yd, i, vs_act(xh * gamma[c] + beta[c], act));
    }
}

__global__ __launch_bounds__(1024) void bn_bwd_reduce_kernel(const void* dy, int dyd, const void* x, int xd, const float* mean,
                                                            const float* invstd, const float* gamma, const float* beta, int act, int Bg, int C,
                                                            int64_t HW, float* sum_dz, float* sum_dz_xhat, int vec) {
    __shared__ double red[16];
    const int c = blockIdx.x, grp = blockIdx.y;
    const int64_t n = (int64_t)Bg * HW;
    const int64_t b0 = (int64_t)grp * Bg;
    const float mu = mean[grp * C + c], is = invstd[grp * C + c], g = gamma[c], bt = beta[c];
    double s1 = 0.0, s2 = 0.0;
    if (vec == 2) {
        const int w = unit_width(xd, dyd, VS_BF16);
        const uint32_t upp = (uint32_t)((HW + w - 1) / w), nunit = (uint32_t)Bg * upp;
        for (uint32_t i = threadIdx.x; i < nunit; i += blockDim.x) {
            const uint32_t b = i / upp, e0 = (i - b * upp) * w;
            const int64_t idx = ((b0 + b) * C + c) * HW + e0;
            float xv[8], gv[8];
            const int cnt = ld_unit(x, xd, idx, (int)(HW - e0), w, xv);
            ld_unit(dy, dyd, idx, (int)(HW - e0), w, gv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j < cnt) {
                    const float xh = (xv[j] - mu) * is;
                    const float dz = gv[j] * act_grad_from_pre(xh * g + bt, act);
                    s1 += (double)dz;
                    s2 += (double)dz * (double)xh;
                }
            }
        }
    } else if (vec && xd == dyd) {
        // one 16-byte load per tensor and thread (8 bf16 / 4 fp32 of one plane); fp32 partial sums over the vector, fp64 across
        const int w = xd == VS_F32 ? 4 : 8;
        const int64_t per = HW / w;
        const uint32_t per32 = (uint32_t)per, nvec = (uint32_t)Bg * per32;
        if (xd == VS_F32) bn_bwd_reduce_vec_loop<4>(dy, x, xd, mu, is, g, bt, act, b0, c, C, HW, per32, nvec, s1, s2);
        else bn_bwd_reduce_vec_loop<8>(dy, x, xd, mu, is, g, bt, act, b0, c, C, HW, per32, nvec, s1, s2);
    } else if (vec) {
        const int w = (xd == VS_F32 || dyd == VS_F32) ? 4 : 8;
        const int64_t per = HW / w;
        for (int64_t i = threadIdx.x; i < (int64_t)Bg * per; i += blockDim.x) {
            const int64_t b = i / per, p = (i - b * per) * w;
            const int64_t idx = ((b0 + b) * C + c) * HW + p;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j < w) {
                    const float xh = (vs_ld(x, xd, idx + j) - mu) * is;
                    const float dz = vs_ld(dy, dyd, idx + j) * act_grad_from_pre(xh * g + bt, act);
                    s1 += (double)dz;
                    s2 += (double)dz * (double)xh;
                }
            }
        }
    } else
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const int64_t b = i / HW, p = i - b * HW;
        const int64_t idx = ((b0 + b) * C + c) * HW + p;
        const float xh = (vs_ld(x, xd, idx) - mu) * is;
        const float dz = vs_ld(dy, dyd, idx) * act_grad_from_pre(xh * g + bt, act);
        s1 += (double)dz;
        s2 += (double)dz * (double)xh;
    }
    const double t1 = block_sum(s1, red);
    const double t2 = block_sum(s2, red);
    if (threadIdx.x == 0) { sum_dz[grp * C + c] = (float)t1; sum_dz_xhat[grp * C + c] = (float)t2; }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const void* dy, int dyd, const void* x, int xd, const float* mean,
                                                           const float* invstd, const float* gamma, const float* beta, int act,
                                                           const float* sum_dz, const float* sum_dz_xhat, void* dx, int dxd, int Bg, int C,
                                                           int64_t HW, int64_t total, int training, int vec, float* gsum_dbeta, float* gsum_dgamma,
                                                           int groups) {
    if (gsum_dbeta && blockIdx.x == 0) {
        // d beta / d gamma of the PARAMETER = the per-call-group sums added up (fixed order); a by-product of workgroup 0
        for (int c = threadIdx.x; c < C; c += 256) {
            float a = 0.f, b = 0.f;
            for (int g = 0; g < groups; ++g) { a += sum_dz[g * C + c]; b += sum_dz_xhat[g * C + c]; }
            gsum_dbeta[c] = a;
            gsum_dgamma[c] = b;
        }
    }
    const float inv_n = 1.f / (float)((int64_t)Bg * HW);
    const int64_t group_elems = (int64_t)Bg * C * HW;
    if (vec == 2) {
        const int w = unit_width(xd, dyd, dxd);
        const uint32_t upp = (uint32_t)((HW + w - 1) / w);
        const int64_t planes = total / HW, nunit = planes * upp;
        const uint32_t ppg = (uint32_t)Bg * (uint32_t)C;
        for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < nunit; u += (int64_t)gridDim.x * 256) {
            const uint32_t plane = (uint32_t)(u / upp), e0 = (uint32_t)(u - (int64_t)plane * upp) * w;
            const int c = (int)(plane % (uint32_t)C), gc = (int)(plane / ppg) * C + c;
            const float mu = mean[gc], is = invstd[gc], g = gamma[c], bt = beta[c];
            const float k1 = sum_dz[gc] * inv_n, k2 = sum_dz_xhat[gc] * inv_n;
            const int64_t i = (int64_t)plane * HW + e0;
            float xv[8], gv[8];
            const int cnt = ld_unit(x, xd, i, (int)(HW - e0), w, xv);
            ld_unit(dy, dyd, i, (int)(HW - e0), w, gv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j < cnt) {
                    const float xh = (xv[j] - mu) * is;
                    const float dz = gv[j] * act_grad_from_pre(xh * g + bt, act);
                    xv[j] = training ? g * is * (dz - k1 - xh * k2) : g * is * dz;
                }
            }
            st_unit(dx, dxd, i, xv, cnt, w);
        }
        return;
    }
    if (vec && xd == dyd) {                                          // 16-byte loads of both tensors, one 16-byte (or 2 x 16) store
        const int w = xd == VS_F32 ? 4 : 8;
        const int64_t nv = total / w;
        const bool small = total < (int64_t)1 << 31;
        if (xd == VS_F32) bn_bwd_apply_vec_loop<4>(dy, x, xd, mean, invstd, gamma, beta, act, sum_dz, sum_dz_xhat, dx, dxd, C, HW, nv, group_elems, small, inv_n, training);
        else bn_bwd_apply_vec_loop<8>(dy, x, xd, mean, invstd, gamma, beta, act, sum_dz, sum_dz_xhat, dx, dxd, C, HW, nv, group_elems, small, inv_n, training);
        return;
    }
    if (vec) {
        const int w = 4;                                             // 4 consecutive elements of one plane per thread
        const int64_t nv = total / w;
        for (int64_t iv = (int64_t)blockIdx.x * 256 + threadIdx.x; iv < nv; iv += (int64_t)gridDim.x * 256) {
            const int64_t i = iv * w;
            const int c = (int)((i / HW) % C);
            const int gc = (int)(i / group_elems) * C + c;
            const float mu = mean[gc], is = invstd[gc], g = gamma[c], bt = beta[c];
            const float k1 = sum_dz[gc] * inv_n, k2 = sum_dz_xhat[gc] * inv_n;
#pragma unroll
            for (int j = 0; j < w; ++j) {
                const float xh = (vs_ld(x, xd, i + j) - mu) * is;
                const float dz = vs_ld(dy, dyd, i + j) * act_grad_from_pre(xh * g + bt, act);
                vs_st(dx, dxd, i + j, training ? g * is * (dz - k1 - xh * k2) : g * is * dz);
            }
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)((i / HW) % C);
        const int gc = (int)(i / group_elems) * C + c;
        const float xh = (vs_ld(x, xd, i) - mean[gc]) * invstd[gc];
        const float dz = vs_ld(dy, dyd, i) * act_grad_from_pre(xh * gamma[c] + beta[c], act);
        float v;
        if (training) v = gamma[c] * invstd[gc] * (dz - sum_dz[gc] * inv_n - xh * sum_dz_xhat[gc] * inv_n);
        else v = gamma[c] * invstd[gc] * dz;                // eval mode: statistics are constants
        vs_st(dx, dxd, i, v);
    }
}

// Small slabs (one (call group, channel) slab of <= 8192 elements: the SST integrator's 8 x 16 x 16 maps, 276 BatchNorm calls per
// step): reduce AND apply in one launch.  A 256-thread workgroup owns the slab, keeps dz and x-hat of its <= 32 elements per thread in
// registers between the two phases (same-dtype 16-byte vector path only), so dy and x are read once and the second launch
// (8-9 us each at these sizes, pure launch latency) disappears.
// out_a[c] = sum_g a[g][c], out_b[c] = sum_g b[g][c] (the per-call-group d beta / d gamma of the one-launch form, added up)
__global__ __launch_bounds__(256) void group_sum2_kernel(const float* a, const float* b, int groups, int C, float* out_a, float* out_b) {
    for (int c = blockIdx.x * 256 + threadIdx.x; c < C; c += gridDim.x * 256) {
        float sa = 0.f, sb = 0.f;
        for (int g = 0; g < groups; ++g) { sa += a[g * C + c]; sb += b[g * C + c]; }
        out_a[c] = sa;
        out_b[c] = sb;
    }
}


// ---- per-call-group parameter gradients summed inside the launch that produces them ---------------------------------------------------
// The one-launch backward kernels below run one workgroup per (channel, call group) and leave d beta / d gamma per group; the sums over the
// groups (what the optimizer wants) used to be a launch of their own (group_sum2_kernel: 30 launches of ~4.8 us per TaxiBJ step, 29 per SST
// step -- pure kernel-boundary cost).  Now the LAST workgroup of a channel to arrive adds the groups' values in group order (bit for bit what
// group_sum2_kernel computes).  Hand-off: thread 0 stores its two partials write-through (relaxed agent-scope stores = sc1), drains them
// (s_waitcnt vmcnt(0)) and bumps the channel's counter (relaxed agent-scope atomic); the workgroup whose add returns groups - 1 reads every
// partial with agent-scope (sc1) loads, which do not hit a stale line of its own L1, writes the sums and clears the counter for the next
// launch.  Placement-independent; the counters come from a static pool (no allocation behind the ABI), dealt out in rotation.
struct BnGroupSums { int groups; float* out_a; float* out_b; unsigned* cnt; };
constexpr unsigned BN_ARRIVE_POOL = 1u << 15;
}  // namespace
__device__ unsigned vs_bn_arrive_pool[1u << 15];  // (external linkage: hipGetSymbolAddress does not find a symbol of an unnamed namespace)
namespace {

__device__ __forceinline__ void bn_group_sums_arrive(const BnGroupSums& gs, float* a, float* b, int grp, int C, int c, float v1, float v2) {
    // thread 0 of workgroup (c, grp); a / b: the per-group arrays [groups][C]
    if (!gs.out_a) { a[grp * C + c] = v1; b[grp * C + c] = v2; return; }
    if (gs.groups == 1) { a[c] = v1; b[c] = v2; gs.out_a[c] = v1; gs.out_b[c] = v2; return; }
    __hip_atomic_store(a + grp * C + c, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(b + grp * C + c, v2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned old = __hip_atomic_fetch_add(gs.cnt + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (unsigned)(gs.groups - 1)) {
        float sa = 0.f, sb = 0.f;
        for (int g = 0; g < gs.groups; ++g) {
            sa += __hip_atomic_load(a + g * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sb += __hip_atomic_load(b + g * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        gs.out_a[c] = sa;
        gs.out_b[c] = sb;
        __hip_atomic_store(gs.cnt + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int NV>                    // 16-byte vectors per thread
__global__ __launch_bounds__(256) void bn_bwd_small_kernel(const void* dy, const void* x, int xd, const float* mean, const float* invstd,
                                                          const float* gamma, const float* beta, int act, float* sum_dz, float* sum_dz_xhat,
                                                          void* dx, int dxd, int Bg, int C, int HW, int training, BnGroupSums gs) {
    __shared__ double red[16];
    const int c = blockIdx.x, grp = blockIdx.y;
    const int w = xd == VS_F32 ? 4 : 8;
    const int per = HW / w, nvec = Bg * per;
    const int64_t b0 = (int64_t)grp * Bg;
    const float mu = mean[grp * C + c], is = invstd[grp * C + c], g = gamma[c], bt = beta[c];
    float dzv[NV][8], xhv[NV][8];
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int r = 0; r < NV; ++r) {
        const int i = threadIdx.x + r * 256;
        if (i < nvec) {
            const int b = i / per, p = (i - b * per) * w;
            const int64_t idx = ((b0 + b) * C + c) * (int64_t)HW + p;
            float xv[8], gv[8];
            const int cnt = ld_vec(x, xd, idx, xv);
            ld_vec(dy, xd, idx, gv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j < cnt) {
                    const float xh = (xv[j] - mu) * is;
                    const float dz = gv[j] * act_grad_from_pre(xh * g + bt, act);
                    xhv[r][j] = xh; dzv[r][j] = dz;
                    a1 += dz; a2 += dz * xh;
                }
            }
        }
    }
    const double t1 = block_sum((double)a1, red);
    const double t2 = block_sum((double)a2, red);
    const float inv_n = 1.f / (float)((int64_t)Bg * HW);
    const float k1 = (float)t1 * inv_n, k2 = (float)t2 * inv_n;
#pragma unroll
    for (int r = 0; r < NV; ++r) {
        const int i = threadIdx.x + r * 256;
        if (i < nvec) {
            const int b = i / per, p = (i - b * per) * w;
            const int64_t idx = ((b0 + b) * C + c) * (int64_t)HW + p;
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = training ? g * is * (dzv[r][j] - k1 - xhv[r][j] * k2) : g * is * dzv[r][j];
            st_vec(dx, dxd, idx, o, w);
        }
    }
    if (threadIdx.x == 0) bn_group_sums_arrive(gs, sum_dz, sum_dz_xhat, grp, C, c, (float)t1, (float)t2);      // (last: the drain waits for this thread's stores too)
}

// The same for the fused residual block (functional.ConvResBlockFn): training mode, one call group, z in a 16-bit type; the upstream
// gradient is EITHER the split slabs a vs_conv3_img16 input-gradient launch left (summed here in order) OR one or two tensors (dy_a in
// any type, dy_b fp32: the gradients of the block's two outputs); d gamma / d beta are written or ADDED to a pending gradient.
template <int NV>
__global__ __launch_bounds__(256) void bn_bwd_small_ex_kernel(const void* dy_a, int dad, const float* dy_b, const float* slabs, int nslabs, const void* x,
                                                             int xd, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                                             int act, float* dbeta, float* dgamma, int accumulate, void* dx, int dxd, int B, int C,
                                                             int HW) {
    __shared__ double red[16];
    const int c = blockIdx.x;
    const int per = HW / 8, nvec = B * per;
    const float mu = mean[c], is = invstd[c], g = gamma[c], bt = beta[c];
    const int64_t stride = (int64_t)B * C * HW;
    float dzv[NV][8], xhv[NV][8];
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int r = 0; r < NV; ++r) {
        const int i = threadIdx.x + r * 256;
        if (i < nvec) {
            const int b = i / per, p = (i - b * per) * 8;
            const int64_t idx = ((int64_t)b * C + c) * (int64_t)HW + p;
            float xv[8], gv[8];
            ld_vec(x, xd, idx, xv);
            if (slabs) {
                f32x4 lo = *reinterpret_cast<const f32x4*>(slabs + idx), hi = *reinterpret_cast<const f32x4*>(slabs + idx + 4);
#pragma unroll 4
                for (int k = 1; k < nslabs; ++k) {
                    const f32x4 l2 = *reinterpret_cast<const f32x4*>(slabs + k * stride + idx), h2 = *reinterpret_cast<const f32x4*>(slabs + k * stride + idx + 4);
                    lo[0] += l2[0]; lo[1] += l2[1]; lo[2] += l2[2]; lo[3] += l2[3];
                    hi[0] += h2[0]; hi[1] += h2[1]; hi[2] += h2[2]; hi[3] += h2[3];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) { gv[j] = lo[j]; gv[j + 4] = hi[j]; }
            } else {
                if (dad == VS_F32) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>((const float*)dy_a + idx), hi = *reinterpret_cast<const f32x4*>((const float*)dy_a + idx + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { gv[j] = lo[j]; gv[j + 4] = hi[j]; }
                } else {
                    ld_vec(dy_a, dad, idx, gv);
                }
                if (dy_b) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(dy_b + idx), hi = *reinterpret_cast<const f32x4*>(dy_b + idx + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { gv[j] += lo[j]; gv[j + 4] += hi[j]; }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = (xv[j] - mu) * is;
                const float dz = gv[j] * act_grad_from_pre(xh * g + bt, act);
                xhv[r][j] = xh; dzv[r][j] = dz;
                a1 += dz; a2 += dz * xh;
            }
        }
    }
    const double t1 = block_sum((double)a1, red);
    const double t2 = block_sum((double)a2, red);
    if (threadIdx.x == 0) {
        if (accumulate) { dbeta[c] += (float)t1; dgamma[c] += (float)t2; }
        else { dbeta[c] = (float)t1; dgamma[c] = (float)t2; }
    }
    const float inv_n = 1.f / (float)((int64_t)B * HW);
    const float k1 = (float)t1 * inv_n, k2 = (float)t2 * inv_n;
#pragma unroll
    for (int r = 0; r < NV; ++r) {
        const int i = threadIdx.x + r * 256;
        if (i < nvec) {
            const int b = i / per, p = (i - b * per) * 8;
            const int64_t idx = ((int64_t)b * C + c) * (int64_t)HW + p;
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = g * is * (dzv[r][j] - k1 - xhv[r][j] * k2);
            st_vec(dx, dxd, idx, o, 8);
        }
    }
}

// The forward counterpart for ONE call group: batch statistics, running-statistics update, affine + activation in one launch
// (vs_bn_stats is two launches -- statistics, running update -- and vs_bn_act_fwd a third).  Two-pass variance from registers.
template <int NV>
__global__ __launch_bounds__(256) void bn_fwd_small_kernel(const void* x, int xd, void* y, int yd, const float* gamma, const float* beta, int act,
                                                          float* mean, float* invstd, float* rmean, float* rvar, float momentum, float eps, int B,
                                                          int C, int HW, const float* slabs = nullptr, int nslabs = 0, const float* bias = nullptr,
                                                          void* z = nullptr, const float* skip = nullptr, float* xnew = nullptr, void* xnew16 = nullptr,
                                                          float* ubvar = nullptr) {
    // blockIdx.y = call group (B = the maps of ONE group; groups lie one after the other along the batch axis): statistics per group;
    // with several groups the running estimates are folded in call order by bn_running_kernel from mean / ubvar (ubvar != NULL)
    __shared__ double red[16];
    const int c = blockIdx.x, grp = blockIdx.y;
    const int64_t b0 = (int64_t)grp * B;
    mean += grp * C;
    invstd += grp * C;
    const int w = xd == VS_F32 ? 4 : 8;
    const int per = HW / w, nvec = B * per;
    const int64_t n = (int64_t)B * HW;
    float xv[NV][8];
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < NV; ++r) {
        const int i = threadIdx.x + r * 256;
        if (i < nvec) {
            const int b = i / per, p = (i - b * per) * w;
            const int64_t idx = ((b0 + b) * C + c) * (int64_t)HW + p;
            int cnt = 8;
            if (slabs) {
                // the convolution left split partial sums (fp32 slabs of the whole tensor): z = round16(sum of the slabs in order + bias) is
                // stored for backward and normalised here (xd is the 16-bit type of z: 8 elements per thread and round)
                const int64_t stride = (int64_t)B * C * HW;
                f32x4 lo = *reinterpret_cast<const f32x4*>(slabs + idx), hi = *reinterpret_cast<const f32x4*>(slabs + idx + 4);
#pragma unroll 4
                for (int k = 1; k < nslabs; ++k) {
                    const f32x4 l2 = *reinterpret_cast<const f32x4*>(slabs + k * stride + idx), h2 = *reinterpret_cast<const f32x4*>(slabs + k * stride + idx + 4);
                    lo[0] += l2[0]; lo[1] += l2[1]; lo[2] += l2[2]; lo[3] += l2[3];
                    hi[0] += h2[0]; hi[1] += h2[1]; hi[2] += h2[2]; hi[3] += h2[3];
                }
                const float bv = bias ? bias[c] : 0.f;
                u16x8 zb;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    zb[j] = vs_f2h((j < 4 ? lo[j] : hi[j - 4]) + bv, xd);
                    xv[r][j] = vs_h2f(zb[j], xd);
                }
                *reinterpret_cast<u16x8*>((unsigned short*)z + idx) = zb;
            } else {
                cnt = ld_vec(x, xd, idx, xv[r]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < cnt) a += xv[r][j];
        }
    }
    const double mu_d = block_sum((double)a, red) / (double)n;
    const float mu = (float)mu_d;
    float q = 0.f;
#pragma unroll
    for (int r = 0; r < NV; ++r) {
        const int i = threadIdx.x + r * 256;
        if (i < nvec) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < w) { const float d = xv[r][j] - mu; q += d * d; }
        }
    }
    double ss = block_sum((double)q, red);
    // (sum (x - fl(mu))^2 = sum (x - mu)^2 + n (mu - fl(mu))^2: remove the rounding of the mean)
    ss -= (double)n * (mu_d - (double)mu) * (mu_d - (double)mu);
    if (ss < 0.0) ss = 0.0;
    const double var = ss / (double)n;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) {
        mean[c] = mu;
        invstd[c] = is;
        if (ubvar) {
            ubvar[grp * C + c] = (float)(n > 1 ? ss / (double)(n - 1) : var);
        } else if (rmean) {
            const double ub = n > 1 ? ss / (double)(n - 1) : var;
            rmean[c] = (float)((1.0 - momentum) * (double)rmean[c] + momentum * (double)mu);
            rvar[c] = (float)((1.0 - momentum) * (double)rvar[c] + momentum * (double)(float)ub);
        }
    }
    const float g = gamma[c], bt = beta[c];
#pragma unroll
    for (int r = 0; r < NV; ++r) {
        const int i = threadIdx.x + r * 256;
        if (i < nvec) {
            const int b = i / per, p = (i - b * per) * w;
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = vs_act((xv[r][j] - mu) * is * g + bt, act);
            const int64_t idx = ((b0 + b) * C + c) * (int64_t)HW + p;
            st_vec(y, yd, idx, o, w);
            if (skip) {
                // residual block tail (resnet.py:66-70): the block output skip + y in fp32 and as the next block's 16-bit operand (w == 8)
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(skip + idx), s1 = *reinterpret_cast<const f32x4*>(skip + idx + 4);
                f32x4 n0, n1;
                u16x8 nb;
#pragma unroll
                for (int j = 0; j < 4; ++j) { n0[j] = s0[j] + o[j]; n1[j] = s1[j] + o[j + 4]; nb[j] = vs_f2h(n0[j], xd); nb[j + 4] = vs_f2h(n1[j], xd); }
                *reinterpret_cast<f32x4*>(xnew + idx) = n0;
                *reinterpret_cast<f32x4*>(xnew + idx + 4) = n1;
                if (xnew16) *reinterpret_cast<u16x8*>((unsigned short*)xnew16 + idx) = nb;
            }
        }
    }
}

// ---- BatchNorm of LARGE slabs with the slab resident in registers (round 4) -------------------------------------------------------------
// vs_bn_stats + vs_bn_act_fwd read the pre-BatchNorm tensor twice (statistics, then apply) and the backward pair reads dy and x twice each:
// HBM passes, 14-17 % of the Moving-MNIST / TaxiBJ steps.  One (call group, channel) slab of the batched decoder / encoder calls is 1.6 K ..
// 131 K 16-bit elements: a 1024-thread workgroup holds it in registers as packed 16-byte vectors (<= 16 per thread), so the tensor is read ONCE
// -- statistics in two passes over the registers (mean, then centred squares: the arithmetic of bn_fwd_small_kernel), apply from the registers.
// Backward: x and dy resident (<= 8 + 8 vectors: slabs up to 65 536 elements).  grid = (C, call groups); running estimates of several call
// groups are folded in call order by bn_running_kernel from mean / the unbiased variances.  ACT: VS_ACT_NONE / VS_ACT_LEAKY compile-time
// (anything else goes through the run-time switch, ACT = -1).
template <int ACT>
__device__ __forceinline__ float slab_act(float v, int act) {
    if constexpr (ACT == VS_ACT_NONE) return v;
    else if constexpr (ACT == VS_ACT_LEAKY) return v > 0.f ? v : 0.2f * v;
    else return vs_act(v, act);
}
template <int ACT>
__device__ __forceinline__ float slab_act_grad(float pre, int act) {
    if constexpr (ACT == VS_ACT_NONE) return 1.f;
    else if constexpr (ACT == VS_ACT_LEAKY) return pre > 0.f ? 1.f : 0.2f;
    else return act_grad_from_pre(pre, act);
}

// element j of a packed 16-byte vector as fp32 (XD: VS_BF16 / VS_F16 compile-time)
template <int XD>
__device__ __forceinline__ float slab_get(const u32x4& v, int j) {
    const unsigned w = v[j >> 1];
    if constexpr (XD == VS_BF16) return __uint_as_float((j & 1) ? (w & 0xffff0000u) : (w << 16));
    else return vs_h2f((unsigned short)((j & 1) ? (w >> 16) : (w & 0xffffu)), VS_F16);
}
// the resident vectors must stay PACKED between the phases: without this fence the compiler keeps the fp32 conversions of phase one alive for
// the later phases (8 registers per vector instead of 4) and spills
template <int NV>
__device__ __forceinline__ void slab_fence(u32x4 (&v)[NV]) {
#pragma unroll
    for (int r = 0; r < NV; ++r) asm volatile("" : "+v"(v[r]));
}

template <int NV, int ACT, int XD>
__global__ __launch_bounds__(1024) void bn_fwd_slab_kernel(const unsigned short* __restrict__ x, void* __restrict__ y, int yd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, int act, float* mean,
                                                           float* invstd, float* ubvar, float* rmean, float* rvar, float momentum, float eps, int Bg, int C,
                                                           int HW) {
    __shared__ double red[16];
    const int c = blockIdx.x, grp = blockIdx.y;
    // a plane is HW / 8 vectors, a power of two <= 1024: the workgroup covers 1024 / per maps per round, a thread's vectors are `bstep` maps apart
    const int per = HW >> 3, bstep = 1024 / per;
    const int bt0 = threadIdx.x / per;
    const int64_t n = (int64_t)Bg * HW;
    const int64_t off0 = (((int64_t)grp * Bg + bt0) * C + c) * (int64_t)HW + ((threadIdx.x - bt0 * per) << 3);
    const int64_t rstride = (int64_t)bstep * C * HW;
    u32x4 xv[NV];
#pragma unroll
    for (int r = 0; r < NV; ++r) {
        xv[r] = u32x4{0u, 0u, 0u, 0u};
        if (bt0 + r * bstep < Bg) xv[r] = *reinterpret_cast<const u32x4*>(x + off0 + r * rstride);
    }
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < NV; ++r) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a += slab_get<XD>(xv[r], j);          // (absent vectors are zeros)
    }
    slab_fence<NV>(xv);
    const double mu_d = block_sum((double)a, red) / (double)n;
    const float mu = (float)mu_d;
    float q = 0.f;
#pragma unroll
    for (int r = 0; r < NV; ++r) {
        if (bt0 + r * bstep < Bg) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = slab_get<XD>(xv[r], j) - mu; q += d * d; }
        }
    }
    slab_fence<NV>(xv);
    double ss = block_sum((double)q, red);
    ss -= (double)n * (mu_d - (double)mu) * (mu_d - (double)mu);          // (the rounding of the mean: as bn_fwd_small_kernel)
    if (ss < 0.0) ss = 0.0;
    const double var = ss / (double)n;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    if (threadIdx.x == 0) {
        mean[grp * C + c] = mu;
        invstd[grp * C + c] = is;
        const double ub = n > 1 ? ss / (double)(n - 1) : var;
        if (ubvar) {
            ubvar[grp * C + c] = (float)ub;
        } else if (rmean) {
            rmean[c] = (float)((1.0 - momentum) * (double)rmean[c] + momentum * (double)mu);
            rvar[c] = (float)((1.0 - momentum) * (double)rvar[c] + momentum * (double)(float)ub);
        }
    }
    const float g = gamma[c], bt = beta[c];
    const float sc = is * g, sh = bt - mu * is * g;
    (void)sh;
#pragma unroll
    for (int r = 0; r < NV; ++r) {
        if (bt0 + r * bstep < Bg) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = slab_act<ACT>((slab_get<XD>(xv[r], j) - mu) * sc + bt, act);
            st_vec(y, yd, off0 + r * rstride, o, 8);
        }
    }
}

template <int NV, int ACT, int XD>
__global__ __launch_bounds__(1024) void bn_bwd_slab_kernel(const unsigned short* __restrict__ dy, const unsigned short* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, int act, float* sum_dz,
                                                           float* sum_dz_xhat, void* __restrict__ dx, int dxd, int Bg, int C, int HW, BnGroupSums gs) {
    __shared__ double red[16];
    const int c = blockIdx.x, grp = blockIdx.y;
    const int per = HW >> 3, bstep = 1024 / per;                           // (the thread -> vector map of bn_fwd_slab_kernel)
    const int bt0 = threadIdx.x / per;
    const int64_t off0 = (((int64_t)grp * Bg + bt0) * C + c) * (int64_t)HW + ((threadIdx.x - bt0 * per) << 3);
    const int64_t rstride = (int64_t)bstep * C * HW;
    const float mu = mean[grp * C + c], is = invstd[grp * C + c], g = gamma[c], bt = beta[c];
    u32x4 xv[NV], gv[NV];
#pragma unroll
    for (int r = 0; r < NV; ++r) {
        xv[r] = u32x4{0u, 0u, 0u, 0u};
        gv[r] = u32x4{0u, 0u, 0u, 0u};
        if (bt0 + r * bstep < Bg) {
            xv[r] = *reinterpret_cast<const u32x4*>(x + off0 + r * rstride);
            gv[r] = *reinterpret_cast<const u32x4*>(dy + off0 + r * rstride);
        }
    }
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int r = 0; r < NV; ++r) {
        asm volatile("" : "+v"(xv[r]), "+v"(gv[r]), "+v"(a1), "+v"(a2));   // (one vector pair at a time: the scheduler would otherwise convert them all up front and spill)
#pragma unroll
        for (int j = 0; j < 8; ++j) {                                      // (absent vectors: dy = 0 contributes nothing)
            const float xh = (slab_get<XD>(xv[r], j) - mu) * is;
            const float dz = slab_get<XD>(gv[r], j) * slab_act_grad<ACT>(xh * g + bt, act);
            a1 += dz;
            a2 += dz * xh;
        }
    }
    slab_fence<NV>(xv);
    slab_fence<NV>(gv);
    const double t1 = block_sum((double)a1, red);
    const double t2 = block_sum((double)a2, red);
    if (threadIdx.x == 0) bn_group_sums_arrive(gs, sum_dz, sum_dz_xhat, grp, C, c, (float)t1, (float)t2);
    const float inv_n = 1.f / (float)((int64_t)Bg * HW);
    const float k1 = (float)t1 * inv_n, k2 = (float)t2 * inv_n;
#pragma unroll
    for (int r = 0; r < NV; ++r) {
        asm volatile("" : "+v"(xv[r]), "+v"(gv[r]) : : "memory");
        if (bt0 + r * bstep < Bg) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = (slab_get<XD>(xv[r], j) - mu) * is;
                const float dz = slab_get<XD>(gv[r], j) * slab_act_grad<ACT>(xh * g + bt, act);
                o[j] = g * is * (dz - k1 - xh * k2);
            }
            st_vec(dx, dxd, off0 + r * rstride, o, 8);
        }
    }
}

// Slabs of 65 537 .. 131 072 elements (the 32 x 32 maps of a 100-128 sample call: the widest BatchNorm layers of the Moving-MNIST and TaxiBJ
// networks): x stays packed in registers (<= 16 vectors), the first NL vectors of dy go global -> LDS by LDS-DMA (144 KB, no register pass) and
// stay there, the remaining ones are read again in the apply phase (from L2 / the Infinity Cache: a workgroup's tail is <= 112 KB).  Traffic
// 3 + (NV - NL) / NV tensor passes instead of 5 (bn_bwd_reduce_kernel + bn_bwd_apply_kernel).
__device__ __attribute__((aligned(16))) const uint32_t bn_zero16[4] = {0u, 0u, 0u, 0u};

template <int NV, int NL, int ACT, int XD>
__global__ __launch_bounds__(1024) void bn_bwd_slab_lds_kernel(const unsigned short* __restrict__ dy, const unsigned short* __restrict__ x,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta, int act, float* sum_dz,
                                                               float* sum_dz_xhat, void* __restrict__ dx, int dxd, int Bg, int C, int HW, BnGroupSums gs) {
    extern __shared__ __attribute__((aligned(16))) char gl_raw[];            // dy vectors [NL][1024] x 16 bytes (the ONLY dynamic LDS object)
    __shared__ double red[16];
    u32x4* gl = reinterpret_cast<u32x4*>(gl_raw);
    const int c = blockIdx.x, grp = blockIdx.y;
    const int per = HW >> 3, bstep = 1024 / per;                           // (the thread -> vector map of bn_fwd_slab_kernel)
    const int bt0 = threadIdx.x / per;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t off0 = (((int64_t)grp * Bg + bt0) * C + c) * (int64_t)HW + ((threadIdx.x - bt0 * per) << 3);
    const int64_t rstride = (int64_t)bstep * C * HW;
    const float mu = mean[grp * C + c], is = invstd[grp * C + c], g = gamma[c], bt = beta[c];
    // dy vectors 0 .. NL-1: global -> LDS, lane-linear (a thread later reads exactly the 16 bytes its own lane's DMA wrote)
#pragma unroll
    for (int r = 0; r < NL; ++r) {
        const void* src = (bt0 + r * bstep < Bg) ? (const void*)(dy + off0 + r * rstride) : (const void*)bn_zero16;
        const uint32_t dst = (uint32_t)(uintptr_t)(gl_raw + ((size_t)r * 1024 + wave * 64) * 16);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src) : "memory", "m0");
    }
    u32x4 xv[NV], gt[NV - NL];
#pragma unroll
    for (int r = 0; r < NV; ++r) {
        xv[r] = u32x4{0u, 0u, 0u, 0u};
        if (bt0 + r * bstep < Bg) xv[r] = *reinterpret_cast<const u32x4*>(x + off0 + r * rstride);
    }
#pragma unroll
    for (int r = NL; r < NV; ++r) {
        gt[r - NL] = u32x4{0u, 0u, 0u, 0u};
        if (bt0 + r * bstep < Bg) gt[r - NL] = *reinterpret_cast<const u32x4*>(dy + off0 + r * rstride);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // (the DMA is counted with the loads)
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int r = 0; r < NV; ++r) {
        u32x4 gv;
        if (r < NL) gv = gl[r * 1024 + threadIdx.x];
        else gv = gt[r < NL ? 0 : r - NL];
        asm volatile("" : "+v"(xv[r]), "+v"(gv), "+v"(a1), "+v"(a2));       // (one vector pair at a time, see bn_bwd_slab_kernel)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = (slab_get<XD>(xv[r], j) - mu) * is;
            const float dz = slab_get<XD>(gv, j) * slab_act_grad<ACT>(xh * g + bt, act);
            a1 += dz;
            a2 += dz * xh;
        }
    }
    slab_fence<NV>(xv);
    const double t1 = block_sum((double)a1, red);
    const double t2 = block_sum((double)a2, red);
    if (threadIdx.x == 0) bn_group_sums_arrive(gs, sum_dz, sum_dz_xhat, grp, C, c, (float)t1, (float)t2);
    const float inv_n = 1.f / (float)((int64_t)Bg * HW);
    const float k1 = (float)t1 * inv_n, k2 = (float)t2 * inv_n;
    // the tail of dy again (its registers were given up after the sums)
#pragma unroll
    for (int r = NL; r < NV; ++r) {
        gt[r - NL] = u32x4{0u, 0u, 0u, 0u};
        if (bt0 + r * bstep < Bg) gt[r - NL] = *reinterpret_cast<const u32x4*>(dy + off0 + r * rstride);
    }
#pragma unroll
    for (int r = 0; r < NV; ++r) {
        u32x4 gv;
        if (r < NL) gv = gl[r * 1024 + threadIdx.x];
        else gv = gt[r < NL ? 0 : r - NL];
        asm volatile("" : "+v"(xv[r]), "+v"(gv) : : "memory");
        if (bt0 + r * bstep < Bg) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = (slab_get<XD>(xv[r], j) - mu) * is;
                const float dz = slab_get<XD>(gv, j) * slab_act_grad<ACT>(xh * g + bt, act);
                o[j] = g * is * (dz - k1 - xh * k2);
            }
            st_vec(dx, dxd, off0 + r * rstride, o, 8);
        }
    }
}

template <int NV, int ACT, int XD>
int launch_bn_bwd_slab_lds(dim3 grid, hipStream_t st, const void* dy, const void* x, const float* mean, const float* invstd, const float* gamma,
                           const float* beta, int act, float* s1, float* s2, void* dx, int dxd, int Bg, int C, int HW, BnGroupSums gs) {
    constexpr int NL = 9;
    auto kfn = bn_bwd_slab_lds_kernel<NV, NL, ACT, XD>;
    constexpr int lds = NL * 1024 * 16;
    static bool attr_set = false;
    if (!attr_set) {                                    // above the 64 KiB default limit of dynamic LDS
        if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return vs_fail(VS_ERR_LAUNCH, "cannot raise the dynamic LDS limit to %d bytes", lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(kfn, grid, dim3(1024), lds, st, (const unsigned short*)dy, (const unsigned short*)x, mean, invstd, gamma, beta, act, s1, s2, dx, dxd,
                       Bg, C, HW, gs);
    return VS_OK;
}

// per-channel sum; blockIdx.y splits the (batch x pixel) extent so that few-channel tensors (the 1-channel frames of the
// last decoder layer) still fill the chip; partial sums meet in one float atomic per workgroup
__global__ __launch_bounds__(256) void chan_sum_kernel(const void* x, int xd, int B, int C, int64_t HW, float* out) {
    __shared__ double red[16];
    const int c = blockIdx.x;
    const int64_t n = (int64_t)B * HW;
    const int64_t per = (n + gridDim.y - 1) / gridDim.y;
    const int64_t i0 = (int64_t)blockIdx.y * per;
    int64_t i1 = i0 + per;
    if (i1 > n) i1 = n;
    double s = 0.0;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
        const int64_t b = i / HW, p = i - b * HW;
        s += (double)vs_ld(x, xd, (b * C + c) * HW + p);
    }
    const double t = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(out + c, (float)t);
}

// the same sum without atomics (1024 float atomics on ONE address -- a 1-channel frame gradient -- serialise: 57 us for 1 MB): chunk sums
// of 16-byte loads into a workspace [C][chunks] (fp64), a second launch adds every channel's chunks in a fixed order
__global__ __launch_bounds__(256) void chan_sum_part_kernel(const void* x, int xd, int B, int C, int64_t HW, double* part) {
    __shared__ double red[16];
    const int c = blockIdx.x, chunks = gridDim.y;
    double s = 0.0;
    if (xd != VS_F32 && HW % 8 == 0) {
        const int64_t hw8 = HW / 8, n = (int64_t)B * hw8;
        const int64_t per = (n + chunks - 1) / chunks, i0 = (int64_t)blockIdx.y * per;
        const int64_t i1 = i0 + per < n ? i0 + per : n;
        for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
            const int64_t b = i / hw8, p = i - b * hw8;
            const u32x4 v = *reinterpret_cast<const u32x4*>((const unsigned short*)x + (b * C + c) * HW + p * 8);
            float t = 0.f;                                             // eight values in fp32, then into the fp64 running sum
#pragma unroll
            for (int d = 0; d < 4; ++d) t += vs_h2f((unsigned short)(v[d] & 0xffffu), xd) + vs_h2f((unsigned short)(v[d] >> 16), xd);
            s += (double)t;
        }
    } else {
        const int64_t n = (int64_t)B * HW;
        const int64_t per = (n + chunks - 1) / chunks, i0 = (int64_t)blockIdx.y * per;
        const int64_t i1 = i0 + per < n ? i0 + per : n;
        for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
            const int64_t b = i / HW, p = i - b * HW;
            s += (double)vs_ld(x, xd, (b * C + c) * HW + p);
        }
    }
    const double t = block_sum(s, red);
    if (threadIdx.x == 0) part[(int64_t)c * chunks + blockIdx.y] = t;
}

__global__ __launch_bounds__(256) void chan_sum_finish_kernel(const double* part, int chunks, float* out) {
    __shared__ double red[16];
    const int c = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < chunks; i += 256) s += part[(int64_t)c * chunks + i];
    const double t = block_sum(s, red);
    if (threadIdx.x == 0) out[c] = (float)t;
}

// MaxPool2d(2,2): planes = B*C, input H x W (even), output H/2 x W/2
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const void* x, int xd, void* y, int yd, int64_t planes, int H, int W) {
    const int OH = H / 2, OW = W / 2;
    const int64_t total = planes * OH * OW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ox = (int)(i % OW), oy = (int)((i / OW) % OH);
        const int64_t pl = i / ((int64_t)OW * OH);
        const int64_t base = (pl * H + 2 * oy) * W + 2 * ox;
        const float a = vs_ld(x, xd, base), b = vs_ld(x, xd, base + 1), c = vs_ld(x, xd, base + W), d = vs_ld(x, xd, base + W + 1);
        vs_st(y, yd, i, fmaxf(fmaxf(a, b), fmaxf(c, d)));
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const void* x, int xd, const void* dy, int dyd, void* dx, int dxd, int64_t planes,
                                                          int H, int W) {
    const int OH = H / 2, OW = W / 2;
    const int64_t total = planes * OH * OW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ox = (int)(i % OW), oy = (int)((i / OW) % OH);
        const int64_t pl = i / ((int64_t)OW * OH);
        const int64_t base = (pl * H + 2 * oy) * W + 2 * ox;
        const float v[4] = {vs_ld(x, xd, base), vs_ld(x, xd, base + 1), vs_ld(x, xd, base + W), vs_ld(x, xd, base + W + 1)};
        int arg = 0;                                        // first maximum in window scan order, like ATen
        float best = v[0];
#pragma unroll
        for (int k = 1; k < 4; ++k)
            if (v[k] > best) { best = v[k]; arg = k; }
        const float g = vs_ld(dy, dyd, i);
        vs_st(dx, dxd, base, arg == 0 ? g : 0.f);
        vs_st(dx, dxd, base + 1, arg == 1 ? g : 0.f);
        vs_st(dx, dxd, base + W, arg == 2 ? g : 0.f);
        vs_st(dx, dxd, base + W + 1, arg == 3 ? g : 0.f);
    }
}

// nearest Upsample x2: planes x (H x W) -> planes x (2H x 2W)
// nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (the chairs ResNet18 stem, conv.py:517): windows overlap, so backward is a
// gather -- every input pixel visits the (at most four) windows that contain it, recomputes each window's arg-max (first
// maximum in row-major window order, like ATen) and adds that window's gradient when it is the winner: no atomics
__device__ __forceinline__ int pool3_argmax(const void* x, int xd, int64_t plane_base, int H, int W, int oy, int ox, float& best) {
    const int y0 = oy * 2 - 1 < 0 ? 0 : oy * 2 - 1, x0 = ox * 2 - 1 < 0 ? 0 : ox * 2 - 1;
    const int y1 = oy * 2 + 2 > H ? H : oy * 2 + 2, x1 = ox * 2 + 2 > W ? W : ox * 2 + 2;
    int arg = y0 * W + x0;                                           // ATen: max_pool2d starts from the first in-range position
    best = -INFINITY;
    for (int iy = y0; iy < y1; ++iy)
        for (int ix = x0; ix < x1; ++ix) {
            const float v = vs_ld(x, xd, plane_base + (int64_t)iy * W + ix);
            if (v > best || v != v) { best = v; arg = iy * W + ix; }  // a NaN always takes over, as in ATen
        }
    return arg;
}

__global__ __launch_bounds__(256) void maxpool3s2_fwd_kernel(const void* x, int xd, void* y, int yd, int64_t planes, int H, int W, int OH, int OW) {
    const int64_t total = planes * OH * OW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ox = (int)(i % OW), oy = (int)((i / OW) % OH);
        const int64_t plane = i / ((int64_t)OW * OH);
        float best;
        pool3_argmax(x, xd, plane * H * W, H, W, oy, ox, best);
        vs_st(y, yd, i, best);
    }
}

__global__ __launch_bounds__(256) void maxpool3s2_bwd_kernel(const void* x, int xd, const void* dy, int dyd, void* dx, int dxd, int64_t planes, int H,
                                                             int W, int OH, int OW) {
    const int64_t total = planes * H * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ix = (int)(i % W), iy = (int)((i / W) % H);
        const int64_t plane = i / ((int64_t)W * H);
        float acc = 0.f;
        // windows (oy, ox) with 2 oy - 1 <= iy <= 2 oy + 1
        for (int oy = iy / 2; oy <= (iy + 1) / 2; ++oy) {
            if (oy < 0 || oy >= OH) continue;
            for (int ox = ix / 2; ox <= (ix + 1) / 2; ++ox) {
                if (ox < 0 || ox >= OW) continue;
                float best;
                if (pool3_argmax(x, xd, plane * H * W, H, W, oy, ox, best) == iy * W + ix) acc += vs_ld(dy, dyd, (plane * OH + oy) * OW + ox);
            }
        }
        vs_st(dx, dxd, i, acc);
    }
}

__global__ __launch_bounds__(256) void upsample_fwd_kernel(const void* x, int xd, void* y, int yd, int64_t planes, int H, int W) {
    const int OH = 2 * H, OW = 2 * W;
    const int64_t total = planes * OH * OW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ox = (int)(i % OW), oy = (int)((i / OW) % OH);
        const int64_t pl = i / ((int64_t)OW * OH);
        vs_st(y, yd, i, vs_ld(x, xd, (pl * H + oy / 2) * W + ox / 2));
    }
}

__global__ __launch_bounds__(256) void upsample_bwd_kernel(const void* dy, int dyd, void* dx, int dxd, int64_t planes, int H, int W) {
    const int OW = 2 * W;
    const int64_t total = planes * H * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int x0 = (int)(i % W), y0 = (int)((i / W) % H);
        const int64_t pl = i / ((int64_t)W * H);
        const int64_t base = (pl * 2 * H + 2 * y0) * OW + 2 * x0;
        vs_st(dx, dxd, i, vs_ld(dy, dyd, base) + vs_ld(dy, dyd, base + 1) + vs_ld(dy, dyd, base + OW) + vs_ld(dy, dyd, base + OW + 1));
    }
}

// ---- 16-byte forms of the 2 x 2 pooling / nearest upsampling kernels (16-bit tensors, rows of whole 16-byte pieces) ----------------------------
// The scalar kernels above move 2 bytes per lane and instruction and spend three 64-bit divisions per element: 46 us for the 20 MB the TaxiBJ
// decoder upsamples (6 launches, 0.53 ms of the 8.1 ms step with the pooling kernels).  Here a thread owns 8 pixels of the SMALL map of a row
// pair: one division per 8 (16) elements, 16-byte loads and stores.  Rows are numbered through all planes (row r of the small map <-> rows 2r,
// 2r + 1 of the large one), so no plane index is needed.  Same arithmetic as the scalar kernels, value for value.
__device__ __forceinline__ float h16(unsigned w, int hi, int dt) { return vs_h2f((unsigned short)(hi ? (w >> 16) : (w & 0xffffu)), dt); }

// nearest x2: small [rows][W] -> large [2 rows][2 W]; W % 8 == 0
__global__ __launch_bounds__(256) void upsample_fwd_v8_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, int64_t rows, int W8) {
    const int64_t total = rows * W8;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (int64_t)gridDim.x * 256) {
        const int64_t r = u / W8;
        const int c = (int)(u - r * W8);
        const u32x4 s = *reinterpret_cast<const u32x4*>(x + (r * W8 + c) * 8);
        u32x4 lo, hi;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            lo[2 * k] = (s[k] & 0xffffu) * 0x10001u;
            lo[2 * k + 1] = (s[k] >> 16) * 0x10001u;
            hi[2 * k] = (s[2 + k] & 0xffffu) * 0x10001u;
            hi[2 * k + 1] = (s[2 + k] >> 16) * 0x10001u;
        }
        unsigned short* o = y + (2 * r * W8 * 2 + 2 * c) * 8;               // large rows hold 2 W8 pieces
        *reinterpret_cast<u32x4*>(o) = lo;
        *reinterpret_cast<u32x4*>(o + 8) = hi;
        *reinterpret_cast<u32x4*>(o + (int64_t)W8 * 16) = lo;
        *reinterpret_cast<u32x4*>(o + (int64_t)W8 * 16 + 8) = hi;
    }
}

// the two large rows of a small row's 8 pixels: a[0..1] = row 2r (16 pixels), b[0..1] = row 2r + 1
__device__ __forceinline__ void load_pair_rows(const unsigned short* big, int64_t r, int W8, int c, u32x4 (&a)[2], u32x4 (&b)[2]) {
    const unsigned short* p = big + (2 * r * W8 * 2 + 2 * c) * 8;
    a[0] = *reinterpret_cast<const u32x4*>(p);
    a[1] = *reinterpret_cast<const u32x4*>(p + 8);
    b[0] = *reinterpret_cast<const u32x4*>(p + (int64_t)W8 * 16);
    b[1] = *reinterpret_cast<const u32x4*>(p + (int64_t)W8 * 16 + 8);
}

// gradient of nearest x2: dx = ((dy00 + dy01) + dy10) + dy11 in fp32, like the scalar kernel
__global__ __launch_bounds__(256) void upsample_bwd_v8_kernel(const unsigned short* __restrict__ dy, unsigned short* __restrict__ dx, int dt, int64_t rows, int W8) {
    const int64_t total = rows * W8;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (int64_t)gridDim.x * 256) {
        const int64_t r = u / W8;
        const int c = (int)(u - r * W8);
        u32x4 a[2], b[2];
        load_pair_rows(dy, r, W8, c, a, b);
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {                                        // small pixel k <- large pixels 2k, 2k + 1 = word k of the 8 words
            const unsigned wa = a[k >> 2][k & 3], wb = b[k >> 2][k & 3];
            const float v = h16(wa, 0, dt) + h16(wa, 1, dt) + h16(wb, 0, dt) + h16(wb, 1, dt);
            const unsigned bits = vs_f2h(v, dt);
            if (k & 1) o[k >> 1] |= bits << 16;
            else o[k >> 1] = bits;
        }
        *reinterpret_cast<u32x4*>(dx + (r * W8 + c) * 8) = o;
    }
}

// MaxPool2d(2, 2): large [2 rows][2 W] -> small [rows][W]
__global__ __launch_bounds__(256) void maxpool_fwd_v8_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, int dt, int64_t rows, int W8) {
    const int64_t total = rows * W8;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (int64_t)gridDim.x * 256) {
        const int64_t r = u / W8;
        const int c = (int)(u - r * W8);
        u32x4 a[2], b[2];
        load_pair_rows(x, r, W8, c, a, b);
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned wa = a[k >> 2][k & 3], wb = b[k >> 2][k & 3];
            const float v = fmaxf(fmaxf(h16(wa, 0, dt), h16(wa, 1, dt)), fmaxf(h16(wb, 0, dt), h16(wb, 1, dt)));
            const unsigned bits = vs_f2h(v, dt);
            if (k & 1) o[k >> 1] |= bits << 16;
            else o[k >> 1] = bits;
        }
        *reinterpret_cast<u32x4*>(y + (r * W8 + c) * 8) = o;
    }
}

// its gradient: dy goes to the first maximum of the window in scan order, zeros to the other three pixels
__global__ __launch_bounds__(256) void maxpool_bwd_v8_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ dy,
                                                             unsigned short* __restrict__ dx, int dt, int64_t rows, int W8) {
    const int64_t total = rows * W8;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (int64_t)gridDim.x * 256) {
        const int64_t r = u / W8;
        const int c = (int)(u - r * W8);
        u32x4 a[2], b[2];
        load_pair_rows(x, r, W8, c, a, b);
        const u32x4 g = *reinterpret_cast<const u32x4*>(dy + (r * W8 + c) * 8);
        u32x4 oa[2], ob[2];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned wa = a[k >> 2][k & 3], wb = b[k >> 2][k & 3];
            const float v[4] = {h16(wa, 0, dt), h16(wa, 1, dt), h16(wb, 0, dt), h16(wb, 1, dt)};
            int arg = 0;
            float best = v[0];
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (v[q] > best) { best = v[q]; arg = q; }
            const unsigned gb = (k & 1) ? (g[k >> 1] >> 16) : (g[k >> 1] & 0xffffu);
            oa[k >> 2][k & 3] = arg == 0 ? gb : (arg == 1 ? gb << 16 : 0u);
            ob[k >> 2][k & 3] = arg == 2 ? gb : (arg == 3 ? gb << 16 : 0u);
        }
        unsigned short* p = dx + (2 * r * W8 * 2 + 2 * c) * 8;
        *reinterpret_cast<u32x4*>(p) = oa[0];
        *reinterpret_cast<u32x4*>(p + 8) = oa[1];
        *reinterpret_cast<u32x4*>(p + (int64_t)W8 * 16) = ob[0];
        *reinterpret_cast<u32x4*>(p + (int64_t)W8 * 16 + 8) = ob[1];
    }
}

// the 16-byte forms apply: one 16-bit type throughout, small-map rows of whole pieces, 16-byte aligned tensors
inline bool v8_ok(int d0, int d1, int Wsmall, const void* p0, const void* p1, const void* p2 = nullptr) {
    static const int mode = getenv("VS_POOL_V8") ? atoi(getenv("VS_POOL_V8")) : 1;
    return mode && vs_is16(d0) && d0 == d1 && Wsmall % 8 == 0 && ((uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2) % 16 == 0;
}

inline unsigned ew_grid(int64_t total) {
    int64_t b = (total + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int vs_bn_stats(const void* x, int x_dtype, int B, int C, int64_t HW, int groups, float* mean, float* invstd, float* var_scratch,
                           float* running_mean, float* running_var, float momentum, float eps, void* stream) {
    VS_CHECK_ARG(x && mean && invstd && B > 0 && C > 0 && HW > 0 && groups >= 1 && B % groups == 0, "vs_bn_stats: bad argument");
    VS_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "vs_bn_stats: running_mean/var must come together");
    VS_CHECK_ARG(!running_mean || var_scratch, "vs_bn_stats: var_scratch [groups*C] is needed to update running statistics");
    const int w_ = x_dtype == VS_F32 ? 4 : 8;
    int vec = (HW % 8 == 0) && ((uintptr_t)x % 16 == 0);
    if (!vec && HW % w_ != 0 && HW >= w_ && (int64_t)B * C * HW < ((int64_t)1 << 31) && (uintptr_t)x % 16 == 0) vec = 2;
    // few workgroups with a long reduction each (encoders: C x groups = 64-512): the loop is latency bound, 1024 threads keep four
    // times as many loads in flight per workgroup
    const unsigned nt_ = ((int64_t)C * groups <= 1024 && (int64_t)(B / groups) * HW >= 8192) ? 1024u : 256u;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(C, groups), dim3(nt_), 0, (hipStream_t)stream, x, x_dtype, B / groups, C, HW, mean, invstd,
                       running_mean ? var_scratch : nullptr, eps, vec);
    VS_CHECK_LAUNCH("vs_bn_stats");
    if (running_mean) {
        hipLaunchKernelGGL(bn_running_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, mean, var_scratch, groups, C, running_mean,
                           running_var, momentum);
        VS_CHECK_LAUNCH("vs_bn_stats running update");
    }
    return VS_OK;
}

// vs_bn_stats without the running update: the unbiased variances go to ubvar [groups][C] for vs_bn_act_fwd_running, which folds them
extern "C" int vs_bn_stats_ub(const void* x, int x_dtype, int B, int C, int64_t HW, int groups, float* mean, float* invstd, float* ubvar, float eps,
                              void* stream) {
    VS_CHECK_ARG(x && mean && invstd && ubvar && B > 0 && C > 0 && HW > 0 && groups >= 1 && B % groups == 0, "vs_bn_stats_ub: bad argument");
    const int w_ = x_dtype == VS_F32 ? 4 : 8;
    int vec = (HW % 8 == 0) && ((uintptr_t)x % 16 == 0);
    if (!vec && HW % w_ != 0 && HW >= w_ && (int64_t)B * C * HW < ((int64_t)1 << 31) && (uintptr_t)x % 16 == 0) vec = 2;
    const unsigned nt_ = ((int64_t)C * groups <= 1024 && (int64_t)(B / groups) * HW >= 8192) ? 1024u : 256u;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(C, groups), dim3(nt_), 0, (hipStream_t)stream, x, x_dtype, B / groups, C, HW, mean, invstd, ubvar, eps, vec);
    VS_CHECK_LAUNCH("vs_bn_stats_ub");
    return VS_OK;
}

// 1 when vs_bn_train_fwd_small serves the tensor: one call group, slabs of <= 8192 elements, 16-byte vectors
extern "C" int vs_bn_train_fwd_small_supported(int x_dtype, int B, int C, int64_t HW) {
    static const int small_mode = getenv("VS_BN_SMALL") ? atoi(getenv("VS_BN_SMALL")) : 1;
    const int w = x_dtype == VS_F32 ? 4 : 8;
    return small_mode && vs_dtype_ok(x_dtype) && B > 0 && C >= 32 && HW > 0 && HW % 8 == 0 && (int64_t)B * HW <= 8192 && HW % w == 0;
}

// Training-mode BatchNorm2d forward of ONE reference call on a small tensor in one launch: mean / invstd [C] (kept for backward),
// running statistics updated in place (NULL: not tracked), y = act(gamma * x_hat + beta).  Same results as vs_bn_stats +
// vs_bn_act_fwd up to summation order.
extern "C" int vs_bn_train_fwd_small(const void* x, int x_dtype, void* y, int y_dtype, const float* gamma, const float* beta, int act, float* mean,
                                     float* invstd, float* running_mean, float* running_var, float momentum, float eps, int B, int C, int64_t HW,
                                     void* stream) {
    VS_CHECK_ARG(x && y && gamma && beta && mean && invstd && vs_dtype_ok(y_dtype), "vs_bn_train_fwd_small: bad argument");
    VS_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "vs_bn_train_fwd_small: running_mean/var must come together");
    if (!vs_bn_train_fwd_small_supported(x_dtype, B, C, HW) || ((uintptr_t)x | (uintptr_t)y) % 16 != 0)
        return vs_fail(VS_ERR_UNSUPPORTED, "vs_bn_train_fwd_small: tensor not served (use vs_bn_stats + vs_bn_act_fwd)");
    const int64_t nvec = (int64_t)B * HW / (x_dtype == VS_F32 ? 4 : 8);
#define VS_BN_SMALL(NV)                                                                                                                 \
    hipLaunchKernelGGL(bn_fwd_small_kernel<NV>, dim3(C), dim3(256), 0, (hipStream_t)stream, x, x_dtype, y, y_dtype, gamma, beta, act, mean, invstd, \
                       running_mean, running_var, momentum, eps, B, C, (int)HW)
    if (nvec <= 256) VS_BN_SMALL(1);
    else if (nvec <= 512) VS_BN_SMALL(2);
    else if (nvec <= 1024) VS_BN_SMALL(4);
    else VS_BN_SMALL(8);
#undef VS_BN_SMALL
    VS_CHECK_LAUNCH("vs_bn_train_fwd_small");
    return VS_OK;
}

// The same for `groups` reference calls stacked along the batch axis (B maps in all, B / groups per call, each normalised with its own
// statistics): mean / invstd [groups][C]; the running estimates are folded in call order by a second (tiny) launch from mean and the
// unbiased variances left in var_scratch [groups][C].  Served when one call's slab is (vs_bn_train_fwd_small_supported(x_dtype, B / groups, ..)).
extern "C" int vs_bn_train_fwd_small_groups(const void* x, int x_dtype, void* y, int y_dtype, const float* gamma, const float* beta, int act, float* mean,
                                            float* invstd, float* var_scratch, float* running_mean, float* running_var, float momentum, float eps,
                                            int B, int C, int64_t HW, int groups, void* stream) {
    VS_CHECK_ARG(x && y && gamma && beta && mean && invstd && vs_dtype_ok(y_dtype) && groups >= 1 && B > 0 && B % groups == 0,
                 "vs_bn_train_fwd_small_groups: bad argument");
    VS_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr) && (!running_mean || var_scratch), "vs_bn_train_fwd_small_groups: running_mean / running_var / var_scratch come together");
    const int Bg = B / groups;
    if (!vs_bn_train_fwd_small_supported(x_dtype, Bg, C, HW) || ((uintptr_t)x | (uintptr_t)y) % 16 != 0)
        return vs_fail(VS_ERR_UNSUPPORTED, "vs_bn_train_fwd_small_groups: tensor not served (use vs_bn_stats + vs_bn_act_fwd)");
    const int64_t nvec = (int64_t)Bg * HW / (x_dtype == VS_F32 ? 4 : 8);
    const dim3 grid(C, groups);
#define VS_BN_SMALL(NV)                                                                                                                 \
    hipLaunchKernelGGL(bn_fwd_small_kernel<NV>, grid, dim3(256), 0, (hipStream_t)stream, x, x_dtype, y, y_dtype, gamma, beta, act, mean, invstd, \
                       (float*)nullptr, (float*)nullptr, momentum, eps, Bg, C, (int)HW, (const float*)nullptr, 0, (const float*)nullptr, (void*)nullptr,  \
                       (const float*)nullptr, (float*)nullptr, (void*)nullptr, running_mean ? var_scratch : (float*)nullptr)
    if (nvec <= 256) VS_BN_SMALL(1);
    else if (nvec <= 512) VS_BN_SMALL(2);
    else if (nvec <= 1024) VS_BN_SMALL(4);
    else VS_BN_SMALL(8);
#undef VS_BN_SMALL
    VS_CHECK_LAUNCH("vs_bn_train_fwd_small_groups");
    if (running_mean) {
        hipLaunchKernelGGL(bn_running_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, mean, var_scratch, groups, C, running_mean,
                           running_var, momentum);
        VS_CHECK_LAUNCH("vs_bn_train_fwd_small_groups running update");
    }
    return VS_OK;
}

// The same on the fp32 split slabs [nslabs][B][C][HW] a convolution left (vs_conv3_img16): z = round(sum of the slabs + bias) in z_dtype
// (16-bit) is written for backward, statistics / running update / affine / activation as vs_bn_train_fwd_small -- the slab reduction,
// the bias and the BatchNorm forward in one launch.
extern "C" int vs_bn_train_fwd_small_slabs(const float* slabs, int nslabs, const float* bias, void* z, int z_dtype, void* y, int y_dtype, const float* gamma,
                                           const float* beta, int act, float* mean, float* invstd, float* running_mean, float* running_var,
                                           float momentum, float eps, const float* skip, float* xnew, void* xnew16, int B, int C, int64_t HW,
                                           void* stream) {
    VS_CHECK_ARG(slabs && nslabs >= 1 && z && y && gamma && beta && mean && invstd && vs_is16(z_dtype) && vs_dtype_ok(y_dtype), "vs_bn_train_fwd_small_slabs: bad argument");
    VS_CHECK_ARG((skip == nullptr) == (xnew == nullptr) && (skip || !xnew16), "vs_bn_train_fwd_small_slabs: skip and xnew come together (xnew16 optional)");
    VS_CHECK_ARG(((uintptr_t)skip | (uintptr_t)xnew | (uintptr_t)xnew16) % 16 == 0, "vs_bn_train_fwd_small_slabs: skip / xnew must be 16-byte aligned");
    VS_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "vs_bn_train_fwd_small_slabs: running_mean/var must come together");
    if (!vs_bn_train_fwd_small_supported(z_dtype, B, C, HW) || ((uintptr_t)slabs | (uintptr_t)z | (uintptr_t)y) % 16 != 0)
        return vs_fail(VS_ERR_UNSUPPORTED, "vs_bn_train_fwd_small_slabs: tensor not served (vs_slab_sum + vs_bn_stats + vs_bn_act_fwd)");
    const int64_t nvec = (int64_t)B * HW / 8;
#define VS_BN_SMALL(NV)                                                                                                                 \
    hipLaunchKernelGGL(bn_fwd_small_kernel<NV>, dim3(C), dim3(256), 0, (hipStream_t)stream, (const void*)nullptr, z_dtype, y, y_dtype, gamma, beta, act, mean, \
                       invstd, running_mean, running_var, momentum, eps, B, C, (int)HW, slabs, nslabs, bias, z, skip, xnew, xnew16)
    if (nvec <= 256) VS_BN_SMALL(1);
    else if (nvec <= 512) VS_BN_SMALL(2);
    else if (nvec <= 1024) VS_BN_SMALL(4);
    else VS_BN_SMALL(8);
#undef VS_BN_SMALL
    VS_CHECK_LAUNCH("vs_bn_train_fwd_small_slabs");
    return VS_OK;
}

// One (call group, channel) slab of 8 193 .. 131 072 16-bit elements: read once, held in the registers of a 1024-thread workgroup
extern "C" int vs_bn_train_fwd_slab_supported(int x_dtype, int Bg, int C, int64_t HW) {
    static const int slab_mode = getenv("VS_BN_SLAB") ? atoi(getenv("VS_BN_SLAB")) : 1;
    const int64_t n = (int64_t)Bg * HW;
    return slab_mode && vs_is16(x_dtype) && Bg > 0 && C > 0 && HW >= 8 && HW <= 8192 && (HW & (HW - 1)) == 0 && n > 8192 && n <= 131072;
}

// Training-mode BatchNorm2d (+ activation) forward of `groups` reference calls stacked along the batch axis, every (call, channel) slab read
// ONCE: mean / invstd [groups][C]; running estimates updated (groups > 1: in call order by a second small launch from var_scratch [groups][C]).
extern "C" int vs_bn_train_fwd_slab(const void* x, int x_dtype, void* y, int y_dtype, const float* gamma, const float* beta, int act, float* mean,
                                    float* invstd, float* var_scratch, float* running_mean, float* running_var, float momentum, float eps, int B, int C,
                                    int64_t HW, int groups, void* stream) {
    VS_CHECK_ARG(x && y && gamma && beta && mean && invstd && vs_dtype_ok(y_dtype) && groups >= 1 && B > 0 && B % groups == 0, "vs_bn_train_fwd_slab: bad argument");
    VS_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr) && (!running_mean || groups == 1 || var_scratch),
                 "vs_bn_train_fwd_slab: running_mean / running_var come together; several call groups need var_scratch");
    const int Bg = B / groups;
    if (!vs_bn_train_fwd_slab_supported(x_dtype, Bg, C, HW) || ((uintptr_t)x | (uintptr_t)y) % 16 != 0)
        return vs_fail(VS_ERR_UNSUPPORTED, "vs_bn_train_fwd_slab: tensor not served (vs_bn_stats + vs_bn_act_fwd)");
    const int64_t nvec = (int64_t)Bg * HW / 8;
    const dim3 grid(C, groups);
    float* ub = (groups > 1 && running_mean) ? var_scratch : nullptr;
    float* rm = groups == 1 ? running_mean : nullptr;
    float* rv = groups == 1 ? running_var : nullptr;
    hipStream_t st = (hipStream_t)stream;
#define VS_BN_SLAB(NV, AV)                                                                                                              \
    do {                                                                                                                                  \
        if (x_dtype == VS_BF16)                                                                                                           \
            hipLaunchKernelGGL((bn_fwd_slab_kernel<NV, AV, VS_BF16>), grid, dim3(1024), 0, st, (const unsigned short*)x, y, y_dtype, gamma, beta, act, mean, \
                               invstd, ub, rm, rv, momentum, eps, Bg, C, (int)HW);                                                        \
        else                                                                                                                              \
            hipLaunchKernelGGL((bn_fwd_slab_kernel<NV, AV, VS_F16>), grid, dim3(1024), 0, st, (const unsigned short*)x, y, y_dtype, gamma, beta, act, mean,  \
                               invstd, ub, rm, rv, momentum, eps, Bg, C, (int)HW);                                                        \
    } while (0)
#define VS_BN_SLAB_NV(AV)                      \
    do {                                       \
        if (nvec <= 2048) VS_BN_SLAB(2, AV);   \
        else if (nvec <= 4096) VS_BN_SLAB(4, AV);  \
        else if (nvec <= 8192) VS_BN_SLAB(8, AV);  \
        else VS_BN_SLAB(16, AV);               \
    } while (0)
    if (act == VS_ACT_LEAKY) VS_BN_SLAB_NV(VS_ACT_LEAKY);
    else if (act == VS_ACT_NONE) VS_BN_SLAB_NV(VS_ACT_NONE);
    else VS_BN_SLAB_NV(-1);
#undef VS_BN_SLAB_NV
#undef VS_BN_SLAB
    VS_CHECK_LAUNCH("vs_bn_train_fwd_slab");
    if (ub) {
        hipLaunchKernelGGL(bn_running_kernel, dim3((C + 255) / 256), dim3(256), 0, st, mean, var_scratch, groups, C, running_mean, running_var, momentum);
        VS_CHECK_LAUNCH("vs_bn_train_fwd_slab running update");
    }
    return VS_OK;
}

extern "C" int vs_bn_act_fwd(const void* x, int x_dtype, void* y, int y_dtype, const float* mean, const float* invstd, const float* gamma,
                             const float* beta, int act, int B, int C, int64_t HW, int groups, void* stream) {
    VS_CHECK_ARG(x && y && mean && invstd && gamma && beta && B > 0 && C > 0 && HW > 0 && groups >= 1 && B % groups == 0, "vs_bn_act_fwd: bad argument");
    const int64_t total = (int64_t)B * C * HW;
    const int w_ = x_dtype == VS_F32 ? 4 : 8;
    int vec = (HW % 8 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0);
    if (!vec && HW >= 8 && total < ((int64_t)1 << 31) && (uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0) vec = 2;   // also mixed dtypes
    (void)w_;
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(ew_grid(vec == 1 ? total / 16 : total / 4)), dim3(256), 0, (hipStream_t)stream, x, x_dtype, y, y_dtype, mean, invstd,
                       gamma, beta, act, C, HW, total, (int64_t)(B / groups) * C * HW, vec);
    VS_CHECK_LAUNCH("vs_bn_act_fwd");
    return VS_OK;
}

// vs_bn_act_fwd that also folds the running estimates (mean / ubvar [groups][C] from vs_bn_stats_ub) in call order: one launch less per call
extern "C" int vs_bn_act_fwd_running(const void* x, int x_dtype, void* y, int y_dtype, const float* mean, const float* invstd, const float* gamma,
                                     const float* beta, int act, int B, int C, int64_t HW, int groups, const float* ubvar, float* running_mean,
                                     float* running_var, float momentum, void* stream) {
    VS_CHECK_ARG(x && y && mean && invstd && gamma && beta && B > 0 && C > 0 && HW > 0 && groups >= 1 && B % groups == 0, "vs_bn_act_fwd_running: bad argument");
    VS_CHECK_ARG(ubvar && running_mean && running_var, "vs_bn_act_fwd_running: ubvar / running_mean / running_var are required");
    const int64_t total = (int64_t)B * C * HW;
    int vec = (HW % 8 == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0);
    if (!vec && HW >= 8 && total < ((int64_t)1 << 31) && (uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0) vec = 2;
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(ew_grid(vec == 1 ? total / 16 : total / 4)), dim3(256), 0, (hipStream_t)stream, x, x_dtype, y, y_dtype, mean, invstd,
                       gamma, beta, act, C, HW, total, (int64_t)(B / groups) * C * HW, vec, ubvar, running_mean, running_var, momentum, groups);
    VS_CHECK_LAUNCH("vs_bn_act_fwd_running");
    return VS_OK;
}

// dgamma_sum / dbeta_sum (both or neither; [C]): additionally the parameter gradients summed over the call groups
extern "C" int vs_bn_act_bwd_gsum(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean, const float* invstd,
                                  const float* gamma, const float* beta, int act, int training, int groups, float* dgamma, float* dbeta, void* dx,
                                  int dx_dtype, int B, int C, int64_t HW, float* dgamma_sum, float* dbeta_sum, void* stream);

extern "C" int vs_bn_act_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean, const float* invstd,
                             const float* gamma, const float* beta, int act, int training, int groups, float* dgamma, float* dbeta, void* dx,
                             int dx_dtype, int B, int C, int64_t HW, void* stream) {
    return vs_bn_act_bwd_gsum(dy, dy_dtype, x, x_dtype, mean, invstd, gamma, beta, act, training, groups, dgamma, dbeta, dx, dx_dtype, B, C, HW, nullptr,
                              nullptr, stream);
}

extern "C" int vs_bn_act_bwd_gsum(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean, const float* invstd,
                                  const float* gamma, const float* beta, int act, int training, int groups, float* dgamma, float* dbeta, void* dx,
                                  int dx_dtype, int B, int C, int64_t HW, float* dgamma_sum, float* dbeta_sum, void* stream) {
    VS_CHECK_ARG((dgamma_sum == nullptr) == (dbeta_sum == nullptr), "vs_bn_act_bwd_gsum: pass both sums or neither");
    VS_CHECK_ARG(dy && x && mean && invstd && gamma && beta && dgamma && dbeta && dx && B > 0 && C > 0 && HW > 0 && groups >= 1 &&
                     B % groups == 0, "vs_bn_act_bwd: bad argument");
    const int w_ = x_dtype == VS_F32 ? 4 : 8;
    int vec = (HW % 8 == 0);
    if ((!vec || x_dtype != dy_dtype) && HW >= 8 && (int64_t)B * C * HW < ((int64_t)1 << 31) && ((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) % 16 == 0)
        vec = 2;                                                         // ragged planes and / or mixed dtypes: unit-per-plane path
    (void)w_;
    // the group sums inside the producing launch (bn_group_sums_arrive); VS_BN_GSUM_FUSED=0: the separate group_sum2_kernel launch
    BnGroupSums gs{groups, nullptr, nullptr, nullptr};
    {
        const char* env = getenv("VS_BN_GSUM_FUSED");                 // read per call: tests switch it
        if (dbeta_sum && !(env && atoi(env) == 0) && C <= 4096) {
            static unsigned* pool = nullptr;
            static unsigned next = 0;
            if (!pool && hipGetSymbolAddress((void**)&pool, HIP_SYMBOL(vs_bn_arrive_pool)) != hipSuccess) pool = nullptr;
            if (pool) {
                if (next + (unsigned)C > BN_ARRIVE_POOL) next = 0;
                gs.out_a = dbeta_sum; gs.out_b = dgamma_sum; gs.cnt = pool + next;
                next += (unsigned)C;
            }
        }
    }
    {
        // one launch for small slabs (see bn_bwd_small_kernel); enough workgroups to be worth it
        static const int small_mode = getenv("VS_BN_SMALL") ? atoi(getenv("VS_BN_SMALL")) : 1;
        const int64_t nslab = (int64_t)(B / groups) * HW;
        const int wv = x_dtype == VS_F32 ? 4 : 8;
        if (small_mode && vec == 1 && x_dtype == dy_dtype && nslab <= 8192 && HW < (1 << 20) &&
            ((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) % 16 == 0 && (int64_t)C * groups >= 32) {
            const int64_t nvec = nslab / wv;
            const dim3 grid(C, groups);
#define VS_BN_SMALL(NV)                                                                                                              \
            hipLaunchKernelGGL(bn_bwd_small_kernel<NV>, grid, dim3(256), 0, (hipStream_t)stream, dy, x, x_dtype, mean, invstd, gamma, beta, act,  \
                               dbeta, dgamma, dx, dx_dtype, B / groups, C, (int)HW, training, gs)
            if (nvec <= 256) VS_BN_SMALL(1);
            else if (nvec <= 512) VS_BN_SMALL(2);
            else if (nvec <= 1024) VS_BN_SMALL(4);
            else VS_BN_SMALL(8);
#undef VS_BN_SMALL
            VS_CHECK_LAUNCH("vs_bn_act_bwd (small slabs)");
            if (dbeta_sum && !gs.out_a) {
                hipLaunchKernelGGL(group_sum2_kernel, dim3((unsigned)vs_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, dbeta, dgamma, groups, C, dbeta_sum,
                                   dgamma_sum);
                VS_CHECK_LAUNCH("vs_bn_act_bwd (group sums)");
            }
            return VS_OK;
        }
    }
    {
        // slabs of 8 193 .. 65 536 16-bit elements: x and dy resident in the registers of a 1024-thread workgroup, both read ONCE (bn_bwd_slab_kernel)
        static const int slab_mode = getenv("VS_BN_SLAB") ? atoi(getenv("VS_BN_SLAB")) : 1;
        const int64_t nslab = (int64_t)(B / groups) * HW;
        if (slab_mode && training && vec == 1 && vs_is16(x_dtype) && x_dtype == dy_dtype && nslab > 8192 && nslab <= 65536 && HW <= 8192 && (HW & (HW - 1)) == 0 &&
            ((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) % 16 == 0) {
            const int64_t nvec = nslab / 8;
            const dim3 grid(C, groups);
            hipStream_t st = (hipStream_t)stream;
#define VS_BN_SLAB(NV, AV)                                                                                                              \
            do {                                                                                                                          \
                if (x_dtype == VS_BF16)                                                                                                   \
                    hipLaunchKernelGGL((bn_bwd_slab_kernel<NV, AV, VS_BF16>), grid, dim3(1024), 0, st, (const unsigned short*)dy, (const unsigned short*)x, mean, \
                                       invstd, gamma, beta, act, dbeta, dgamma, dx, dx_dtype, B / groups, C, (int)HW, gs);                \
                else                                                                                                                      \
                    hipLaunchKernelGGL((bn_bwd_slab_kernel<NV, AV, VS_F16>), grid, dim3(1024), 0, st, (const unsigned short*)dy, (const unsigned short*)x, mean,  \
                                       invstd, gamma, beta, act, dbeta, dgamma, dx, dx_dtype, B / groups, C, (int)HW, gs);                \
            } while (0)
#define VS_BN_SLAB_NV(AV)                          \
            do {                                   \
                if (nvec <= 2048) VS_BN_SLAB(2, AV);   \
                else if (nvec <= 4096) VS_BN_SLAB(4, AV);  \
                else VS_BN_SLAB(8, AV);            \
            } while (0)
            if (act == VS_ACT_LEAKY) VS_BN_SLAB_NV(VS_ACT_LEAKY);
            else if (act == VS_ACT_NONE) VS_BN_SLAB_NV(VS_ACT_NONE);
            else VS_BN_SLAB_NV(-1);
#undef VS_BN_SLAB_NV
#undef VS_BN_SLAB
            VS_CHECK_LAUNCH("vs_bn_act_bwd (resident slabs)");
            if (dbeta_sum && !gs.out_a) {
                hipLaunchKernelGGL(group_sum2_kernel, dim3((unsigned)vs_cdiv(C, 256)), dim3(256), 0, st, dbeta, dgamma, groups, C, dbeta_sum, dgamma_sum);
                VS_CHECK_LAUNCH("vs_bn_act_bwd (group sums)");
            }
            return VS_OK;
        }
    }
    {
        // slabs of 65 537 .. 131 072 elements: x in registers, dy in LDS + a re-read tail (bn_bwd_slab_lds_kernel).  VS_BN_SLAB=2: not these.
        static const int slab_mode = getenv("VS_BN_SLAB") ? atoi(getenv("VS_BN_SLAB")) : 1;
        const int64_t nslab = (int64_t)(B / groups) * HW;
        if (slab_mode == 1 && training && vec == 1 && vs_is16(x_dtype) && x_dtype == dy_dtype && nslab > 65536 && nslab <= 131072 && HW <= 8192 &&
            (HW & (HW - 1)) == 0 && ((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) % 16 == 0) {
            const int64_t nvec = nslab / 8;
            const dim3 grid(C, groups);
            hipStream_t st = (hipStream_t)stream;
            int rc;
#define VS_BN_SLAB(NV, AV)                                                                                                                            \
            rc = x_dtype == VS_BF16 ? launch_bn_bwd_slab_lds<NV, AV, VS_BF16>(grid, st, dy, x, mean, invstd, gamma, beta, act, dbeta, dgamma, dx, dx_dtype,   \
                                                                              B / groups, C, (int)HW, gs)                                              \
                                    : launch_bn_bwd_slab_lds<NV, AV, VS_F16>(grid, st, dy, x, mean, invstd, gamma, beta, act, dbeta, dgamma, dx, dx_dtype,    \
                                                                             B / groups, C, (int)HW, gs)
#define VS_BN_SLAB_NV(AV)                              \
            do {                                       \
                if (nvec <= 12288) VS_BN_SLAB(12, AV); \
                else if (nvec <= 14336) VS_BN_SLAB(14, AV); \
                else VS_BN_SLAB(16, AV);               \
            } while (0)
            if (act == VS_ACT_LEAKY) VS_BN_SLAB_NV(VS_ACT_LEAKY);
            else if (act == VS_ACT_NONE) VS_BN_SLAB_NV(VS_ACT_NONE);
            else VS_BN_SLAB_NV(-1);
#undef VS_BN_SLAB_NV
#undef VS_BN_SLAB
            if (rc != VS_OK) return rc;
            VS_CHECK_LAUNCH("vs_bn_act_bwd (resident slabs, LDS)");
            if (dbeta_sum && !gs.out_a) {
                hipLaunchKernelGGL(group_sum2_kernel, dim3((unsigned)vs_cdiv(C, 256)), dim3(256), 0, st, dbeta, dgamma, groups, C, dbeta_sum, dgamma_sum);
                VS_CHECK_LAUNCH("vs_bn_act_bwd (group sums)");
            }
            return VS_OK;
        }
    }
    const unsigned nt_ = ((int64_t)C * groups <= 1024 && (int64_t)(B / groups) * HW >= 8192) ? 1024u : 256u;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C, groups), dim3(nt_), 0, (hipStream_t)stream, dy, dy_dtype, x, x_dtype, mean, invstd, gamma,
                       beta, act, B / groups, C, HW, dbeta, dgamma, vec);
    VS_CHECK_LAUNCH("vs_bn_act_bwd reduce");
    const int64_t total = (int64_t)B * C * HW;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(vec == 1 && x_dtype == dy_dtype ? total / 8 : total / 4)), dim3(256), 0, (hipStream_t)stream, dy, dy_dtype, x, x_dtype, mean, invstd,
                       gamma, beta, act, dbeta, dgamma, dx, dx_dtype, B / groups, C, HW, total, training, vec, dbeta_sum, dgamma_sum, groups);
    VS_CHECK_LAUNCH("vs_bn_act_bwd apply");
    return VS_OK;
}

// Training-mode BatchNorm2d + activation backward of one call on a small 16-bit tensor z (vs_bn_train_fwd_small_supported) in one
// launch, with the upstream gradient taken from split slabs (slabs != NULL: [nslabs][B][C][HW] fp32, dy_a / dy_b ignored) or from
// dy_a (+ dy_b, fp32, optional); accumulate != 0 ADDS d gamma / d beta to the given vectors (a pending gradient) instead of storing.
extern "C" int vs_bn_act_bwd_small_ex(const void* dy_a, int dy_a_dtype, const float* dy_b, const float* slabs, int nslabs, const void* z, int z_dtype,
                                      const float* mean, const float* invstd, const float* gamma, const float* beta, int act, float* dgamma,
                                      float* dbeta, int accumulate, void* dx, int dx_dtype, int B, int C, int64_t HW, void* stream) {
    VS_CHECK_ARG((slabs ? nslabs >= 1 : dy_a != nullptr) && z && mean && invstd && gamma && beta && dgamma && dbeta && dx && vs_is16(z_dtype) &&
                     vs_dtype_ok(dx_dtype) && (slabs || vs_dtype_ok(dy_a_dtype)), "vs_bn_act_bwd_small_ex: bad argument");
    if (!vs_bn_train_fwd_small_supported(z_dtype, B, C, HW) ||
        ((uintptr_t)dy_a | (uintptr_t)dy_b | (uintptr_t)slabs | (uintptr_t)z | (uintptr_t)dx) % 16 != 0)
        return vs_fail(VS_ERR_UNSUPPORTED, "vs_bn_act_bwd_small_ex: tensor not served (vs_bn_act_bwd)");
    const int64_t nvec = (int64_t)B * HW / 8;
#define VS_BN_SMALL(NV)                                                                                                                  \
    hipLaunchKernelGGL(bn_bwd_small_ex_kernel<NV>, dim3(C), dim3(256), 0, (hipStream_t)stream, dy_a, dy_a_dtype, dy_b, slabs, nslabs, z, z_dtype, mean, \
                       invstd, gamma, beta, act, dbeta, dgamma, accumulate, dx, dx_dtype, B, C, (int)HW)
    if (nvec <= 256) VS_BN_SMALL(1);
    else if (nvec <= 512) VS_BN_SMALL(2);
    else if (nvec <= 1024) VS_BN_SMALL(4);
    else VS_BN_SMALL(8);
#undef VS_BN_SMALL
    VS_CHECK_LAUNCH("vs_bn_act_bwd_small_ex");
    return VS_OK;
}

extern "C" int vs_chan_sum(const void* x, int x_dtype, int B, int C, int64_t HW, float* out, void* stream) {
    VS_CHECK_ARG(x && out && B > 0 && C > 0 && HW > 0, "vs_chan_sum: bad argument");
    if (vs_zero_async(out, (size_t)C * sizeof(float), (hipStream_t)stream) != hipSuccess) return vs_fail(VS_ERR_LAUNCH, "vs_chan_sum: memset failed");
    int64_t chunks = ((int64_t)B * HW + 32767) / 32768;
    const int64_t want = (1024 + C - 1) / C;                     // aim at ~1024 workgroups in total
    if (chunks > want) chunks = want;
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL(chan_sum_kernel, dim3(C, (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, x, x_dtype, B, C, HW, out);
    VS_CHECK_LAUNCH("vs_chan_sum");
    return VS_OK;
}

namespace {
int chan_sum_chunks(int B, int C, int64_t HW) {
    int64_t chunks = ((int64_t)B * HW + 8191) / 8192;            // >= 8 k elements (32 per thread) per workgroup
    const int64_t want = (2048 + C - 1) / C;                     // ~2048 workgroups in total
    if (chunks > want) chunks = want;
    return chunks < 1 ? 1 : (int)chunks;
}
}  // namespace

extern "C" size_t vs_chan_sum_workspace_bytes(int B, int C, int64_t HW) {
    return (B > 0 && C > 0 && HW > 0) ? (size_t)C * chan_sum_chunks(B, C, HW) * sizeof(double) : 0;
}

// vs_chan_sum with the chunk sums in a caller-provided workspace: no atomics, a fixed summation order
extern "C" int vs_chan_sum_ws(const void* x, int x_dtype, int B, int C, int64_t HW, void* ws, size_t ws_bytes, float* out, void* stream) {
    VS_CHECK_ARG(x && out && ws && B > 0 && C > 0 && HW > 0 && vs_dtype_ok(x_dtype), "vs_chan_sum_ws: bad argument");
    VS_CHECK_ARG(ws_bytes >= vs_chan_sum_workspace_bytes(B, C, HW) && (uintptr_t)ws % 8 == 0, "vs_chan_sum_ws: workspace too small (vs_chan_sum_workspace_bytes)");
    VS_CHECK_ARG(x_dtype == VS_F32 || HW % 8 != 0 || (uintptr_t)x % 16 == 0, "vs_chan_sum_ws: a 16-bit operand must be 16-byte aligned");
    const int chunks = chan_sum_chunks(B, C, HW);
    hipLaunchKernelGGL(chan_sum_part_kernel, dim3(C, (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, x, x_dtype, B, C, HW, (double*)ws);
    VS_CHECK_LAUNCH("vs_chan_sum_ws");
    hipLaunchKernelGGL(chan_sum_finish_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, (const double*)ws, chunks, out);
    VS_CHECK_LAUNCH("vs_chan_sum_ws (finish)");
    return VS_OK;
}

extern "C" int vs_maxpool2_fwd(const void* x, int x_dtype, void* y, int y_dtype, int64_t planes, int H, int W, void* stream) {
    VS_CHECK_ARG(x && y && planes > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "vs_maxpool2_fwd: bad argument (H, W must be even)");
    if (v8_ok(x_dtype, y_dtype, W / 2, x, y)) {
        hipLaunchKernelGGL(maxpool_fwd_v8_kernel, dim3(ew_grid(planes * (H / 2) * (W / 16))), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x,
                           (unsigned short*)y, x_dtype, planes * (H / 2), W / 16);
        VS_CHECK_LAUNCH("vs_maxpool2_fwd");
        return VS_OK;
    }
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_grid(planes * (H / 2) * (W / 2))), dim3(256), 0, (hipStream_t)stream, x, x_dtype, y, y_dtype,
                       planes, H, W);
    VS_CHECK_LAUNCH("vs_maxpool2_fwd");
    return VS_OK;
}

extern "C" int vs_maxpool2_bwd(const void* x, int x_dtype, const void* dy, int dy_dtype, void* dx, int dx_dtype, int64_t planes, int H, int W,
                               void* stream) {
    VS_CHECK_ARG(x && dy && dx && planes > 0 && H % 2 == 0 && W % 2 == 0, "vs_maxpool2_bwd: bad argument");
    if (x_dtype == dx_dtype && v8_ok(x_dtype, dy_dtype, W / 2, x, dy, dx)) {
        hipLaunchKernelGGL(maxpool_bwd_v8_kernel, dim3(ew_grid(planes * (H / 2) * (W / 16))), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x,
                           (const unsigned short*)dy, (unsigned short*)dx, x_dtype, planes * (H / 2), W / 16);
        VS_CHECK_LAUNCH("vs_maxpool2_bwd");
        return VS_OK;
    }
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_grid(planes * (H / 2) * (W / 2))), dim3(256), 0, (hipStream_t)stream, x, x_dtype, dy, dy_dtype,
                       dx, dx_dtype, planes, H, W);
    VS_CHECK_LAUNCH("vs_maxpool2_bwd");
    return VS_OK;
}

extern "C" int vs_maxpool3s2_fwd(const void* x, int x_dtype, void* y, int y_dtype, int64_t planes, int H, int W, void* stream) {
    VS_CHECK_ARG(x && y && planes > 0 && H > 0 && W > 0, "vs_maxpool3s2_fwd: bad argument");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool3s2_fwd_kernel, dim3(ew_grid(planes * OH * OW)), dim3(256), 0, (hipStream_t)stream, x, x_dtype, y, y_dtype, planes, H,
                       W, OH, OW);
    VS_CHECK_LAUNCH("vs_maxpool3s2_fwd");
    return VS_OK;
}

extern "C" int vs_maxpool3s2_bwd(const void* x, int x_dtype, const void* dy, int dy_dtype, void* dx, int dx_dtype, int64_t planes, int H, int W,
                                 void* stream) {
    VS_CHECK_ARG(x && dy && dx && planes > 0 && H > 0 && W > 0, "vs_maxpool3s2_bwd: bad argument");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3(ew_grid(planes * H * W)), dim3(256), 0, (hipStream_t)stream, x, x_dtype, dy, dy_dtype, dx,
                       dx_dtype, planes, H, W, OH, OW);
    VS_CHECK_LAUNCH("vs_maxpool3s2_bwd");
    return VS_OK;
}

extern "C" int vs_upsample2_fwd(const void* x, int x_dtype, void* y, int y_dtype, int64_t planes, int H, int W, void* stream) {
    VS_CHECK_ARG(x && y && planes > 0 && H > 0 && W > 0, "vs_upsample2_fwd: bad argument");
    if (v8_ok(x_dtype, y_dtype, W, x, y)) {
        hipLaunchKernelGGL(upsample_fwd_v8_kernel, dim3(ew_grid(planes * H * (W / 8))), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x,
                           (unsigned short*)y, planes * H, W / 8);
        VS_CHECK_LAUNCH("vs_upsample2_fwd");
        return VS_OK;
    }
    hipLaunchKernelGGL(upsample_fwd_kernel, dim3(ew_grid(planes * 4 * H * W)), dim3(256), 0, (hipStream_t)stream, x, x_dtype, y, y_dtype, planes,
                       H, W);
    VS_CHECK_LAUNCH("vs_upsample2_fwd");
    return VS_OK;
}

extern "C" int vs_upsample2_bwd(const void* dy, int dy_dtype, void* dx, int dx_dtype, int64_t planes, int H, int W, void* stream) {
    VS_CHECK_ARG(dy && dx && planes > 0 && H > 0 && W > 0, "vs_upsample2_bwd: bad argument");
    if (v8_ok(dy_dtype, dx_dtype, W, dy, dx)) {
        hipLaunchKernelGGL(upsample_bwd_v8_kernel, dim3(ew_grid(planes * H * (W / 8))), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)dy,
                           (unsigned short*)dx, dy_dtype, planes * H, W / 8);
        VS_CHECK_LAUNCH("vs_upsample2_bwd");
        return VS_OK;
    }
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3(ew_grid(planes * H * W)), dim3(256), 0, (hipStream_t)stream, dy, dy_dtype, dx, dx_dtype, planes,
                       H, W);
    VS_CHECK_LAUNCH("vs_upsample2_bwd");
    return VS_OK;
}
