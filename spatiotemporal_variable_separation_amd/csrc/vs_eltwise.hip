// vs_eltwise.hip -- HBM-bound helpers around the GEMMs: casts, strided window copy, bias-gradient column
// sums, activation forward/backward.  All are grid-stride kernels with 16-byte accesses on the contiguous
// path (8 bf16 / 4 fp32 per lane), capped at 2048 workgroups of 256 threads (8 per CU on 256 CUs).
#include <stdarg.h>
#include "vs_common.h"

thread_local char vs_err_buf[256] = "";
unsigned* vs_g_exchange_guard = nullptr;

extern "C" int vs_exchange_guard_set(void* word) {
    if ((uintptr_t)word % 4 != 0) return vs_fail(VS_ERR_ARG, "vs_exchange_guard_set: the guard word must be 4-byte aligned");
    vs_g_exchange_guard = (unsigned*)word;
    return VS_OK;
}
extern "C" void* vs_exchange_guard_get(void) { return vs_g_exchange_guard; }
unsigned* vs_g_exchange_skips = nullptr;
extern "C" int vs_exchange_skip_counter_set(void* word) {
    if ((uintptr_t)word % 4 != 0) return vs_fail(VS_ERR_ARG, "vs_exchange_skip_counter_set: the counter must be 4-byte aligned");
    vs_g_exchange_skips = (unsigned*)word;
    return VS_OK;
}

int vs_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(vs_err_buf, sizeof(vs_err_buf), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* vs_last_error(void) { return vs_err_buf; }
extern "C" const char* vs_version(void) { return "varsep_hip 0.2 (gfx950)"; }

namespace {

inline unsigned grid_for(int64_t work_items) {
    int64_t b = vs_cdiv(work_items, 256);
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// 4 consecutive elements per thread per iteration
template <class F>
__device__ __forceinline__ void map4(const void* x, int xd, void* y, int yd, int64_t n, F f) {
    const int64_t n4 = n / 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float v[4];
        if (xd == VS_F32) { f32x4 t = ((const f32x4*)x)[i]; v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
        else { const u16x4 t = ((const u16x4*)x)[i]; v[0] = vs_h2f(t[0], xd); v[1] = vs_h2f(t[1], xd); v[2] = vs_h2f(t[2], xd); v[3] = vs_h2f(t[3], xd); }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = f(v[j], i * 4 + j);
        if (yd == VS_F32) { f32x4 t = {v[0], v[1], v[2], v[3]}; ((f32x4*)y)[i] = t; }
        else { const u16x4 t = {vs_f2h(v[0], yd), vs_f2h(v[1], yd), vs_f2h(v[2], yd), vs_f2h(v[3], yd)}; ((u16x4*)y)[i] = t; }
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        vs_st(y, yd, i, f(vs_ld(x, xd, i), i));
}

__global__ __launch_bounds__(256) void cast_kernel(const void* src, int sd, void* dst, int dd, int64_t n) {
    map4(src, sd, dst, dd, n, [](float v, int64_t) { return v; });
}

__global__ __launch_bounds__(256) void act_fwd_kernel(const void* x, int xd, void* y, int yd, int act, int64_t n) {
    map4(x, xd, y, yd, n, [act](float v, int64_t) { return vs_act(v, act); });
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const void* dy, int dyd, const void* y, int yd, void* dz, int dzd, int act,
                                                      int64_t n) {
    map4(dy, dyd, dz, dzd, n, [=](float g, int64_t i) { return g * vs_act_grad_from_out(vs_ld(y, yd, i), act); });
}

__global__ __launch_bounds__(256) void copy2d_kernel(const void* src, int sd, int64_t lds, void* dst, int dd, int64_t ldd,
                                                     int64_t rows, int64_t cols, const int32_t* off_dev, int64_t off_scale) {
    const int64_t off = off_dev ? (int64_t)(*off_dev) * off_scale : 0;
    const int64_t total = rows * cols;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / cols, c = idx % cols;
        vs_st(dst, dd, r * ldd + c, vs_ld(src, sd, r * lds + off + c));
    }
}

// Two windows of the same [rows, lds] source stacked into one [2 rows, ldd] operand in ONE launch: destination rows [0, rows) come from
// src + elem_a + *off_dev * off_scale (a window that moves on the device), rows [rows, 2 rows) from src + elem_b (a fixed one).  The E_t
// input of the recorded MLP step ([random window; conditioning window], train.py:45-88 of the reference) -- two copy launches in front of
// the critical path before.  VEC: four elements per thread (everything a multiple of 4 elements).
template <bool VEC>
__global__ __launch_bounds__(256) void copy2d_pair_kernel(const void* src, int sd, int64_t lds, void* dst, int dd, int64_t ldd, int64_t rows, int64_t cols,
                                                          const int32_t* off_dev, int64_t off_scale, int64_t elem_a, int64_t elem_b) {
    const int64_t off_a = elem_a + (off_dev ? (int64_t)(*off_dev) * off_scale : 0);
    constexpr int V = VEC ? 4 : 1;
    const int64_t cv = cols / V, total = 2 * rows * cv;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / cv, c = (idx - r * cv) * V;
        const int64_t s = (r < rows ? r * lds + off_a : (r - rows) * lds + elem_b) + c;
        if (VEC) {
            float v[4];
            if (sd == VS_F32) { const f32x4 t = *reinterpret_cast<const f32x4*>((const float*)src + s); v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
            else { const u16x4 t = *reinterpret_cast<const u16x4*>((const unsigned short*)src + s); for (int j = 0; j < 4; ++j) v[j] = vs_h2f(t[j], sd); }
            if (dd == VS_F32) *reinterpret_cast<f32x4*>((float*)dst + r * ldd + c) = f32x4{v[0], v[1], v[2], v[3]};
            else *reinterpret_cast<u16x4*>((unsigned short*)dst + r * ldd + c) = u16x4{vs_f2h(v[0], dd), vs_f2h(v[1], dd), vs_f2h(v[2], dd), vs_f2h(v[3], dd)};
        } else {
            vs_st(dst, dd, r * ldd + c, vs_ld(src, sd, s));
        }
    }
}

// column sums: block (bx, by) reduces rows [by*RB, by*RB+RB) of columns [bx*64, bx*64+64) and adds with one atomic per column
constexpr int CS_ROWS = 256;
__global__ __launch_bounds__(256) void colsum_kernel(const void* X, int xd, int64_t ldx, int64_t M, int64_t N, float* out) {
    __shared__ float part[4][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int64_t n = (int64_t)blockIdx.x * 64 + c;
    const int64_t r0 = (int64_t)blockIdx.y * CS_ROWS;
    int64_t r1 = r0 + CS_ROWS;
    if (r1 > M) r1 = M;
    float s = 0.f;
    if (n < N)
        for (int64_t r = r0 + g; r < r1; r += 4) s += vs_ld(X, xd, r * ldx + n);
    part[g][c] = s;
    __syncthreads();
    if (g == 0 && n < N) atomicAdd(out + n, part[0][c] + part[1][c] + part[2][c] + part[3][c]);
}

// several column-sum jobs in one launch (all bias gradients of a Linear chain): blockIdx.x walks the jobs' column blocks
constexpr int CS_MAXJ = 12;
struct ColsumJobs {
    int nj;
    const void* x[CS_MAXJ]; int dt[CS_MAXJ];
    int64_t ld[CS_MAXJ], M[CS_MAXJ], N[CS_MAXJ];
    float* out[CS_MAXJ];
    int blk_off[CS_MAXJ + 1];
    int direct;                          // one row block per column: plain stores (nothing to meet, so nothing to clear first)
};

// A workgroup reduces CS_ROWS rows of a 256-column block: 32 lanes x 8 consecutive columns (one 16-byte load per row for bf16,
// two for fp32), 8 row groups, 4 rows in flight per thread; the 8 row groups meet in LDS, one atomic per column.
constexpr int CSM_COLS = 256;
__global__ __launch_bounds__(256) void colsum_multi_kernel(ColsumJobs J) {
    __shared__ float part[8][CSM_COLS + 8];
    int j = 0;
    while (j + 1 < J.nj && (int)blockIdx.x >= J.blk_off[j + 1]) ++j;
    const int64_t M = J.M[j], N = J.N[j], ldx = J.ld[j];
    const int64_t r0 = (int64_t)blockIdx.y * CS_ROWS;
    if (r0 >= M) return;
    const int c8 = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int64_t n = (int64_t)(blockIdx.x - J.blk_off[j]) * CSM_COLS + 8 * c8;
    int64_t r1 = r0 + CS_ROWS;
    if (r1 > M) r1 = M;
    const void* X = J.x[j];
    const int xd = J.dt[j];
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    const bool vec = n + 8 <= N && ldx % 8 == 0 && (uintptr_t)X % 16 == 0;
    if (vec && vs_is16(xd)) {
        const unsigned short* p = (const unsigned short*)X + n;
        int64_t r = r0 + g;
        for (; r + 24 < r1; r += 32) {
            u32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const u32x4*>(p + (r + 8 * u) * ldx);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned short* h = reinterpret_cast<const unsigned short*>(&v[u]);
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += vs_h2f(h[e], xd);
            }
        }
        for (; r < r1; r += 8) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(p + r * ldx);
            const unsigned short* h = reinterpret_cast<const unsigned short*>(&v);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += vs_h2f(h[e], xd);
        }
    } else if (vec && xd == VS_F32) {
        const float* p = (const float*)X + n;
        for (int64_t r = r0 + g; r < r1; r += 8) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(p + r * ldx), b = *reinterpret_cast<const f32x4*>(p + r * ldx + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[e] += a[e]; s[4 + e] += b[e]; }
        }
    } else if (n < N) {
        for (int64_t r = r0 + g; r < r1; r += 8)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (n + e < N) s[e] += vs_ld(X, xd, r * ldx + n + e);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[g][8 * c8 + e] = s[e];
    __syncthreads();
    const int col = threadIdx.x;
    const int64_t nc = (int64_t)(blockIdx.x - J.blk_off[j]) * CSM_COLS + col;
    if (nc < N) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += part[q][col];
        if (J.direct) J.out[j][nc] = t;
        else atomicAdd(J.out[j] + nc, t);
    }
}

// ---- fused frame losses: sum of squared errors of every decoded frame against its target frame ------------------------------
// frames [B, G, D] fp32 (G frames per sample: the auto-encoding reconstruction followed by the n forecasts), full [B, T, D]
// fp32 (all observed frames), idx[g] = index of the frame of `full` that frame g is compared with (device array, so the
// random auto-encoding target can change without re-recording a graph).  sums[0] = SSE of frame 0 (ae_loss, train.py:85-86),
// sums[1] = SSE of frames 1.. (forecast loss, train.py:139).
__global__ __launch_bounds__(256) void frames_sse_kernel(const float* frames, const float* full, const int* idx, int64_t rows, int G, int T,
                                                         int64_t D, float* sums) {
    __shared__ float red[2][4];
    float s0 = 0.f, s1 = 0.f;                      // frame 0 / frames 1..
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {       // each workgroup walks many (sample, frame) rows:
        const int g = (int)(r % G);                                 // two atomics per workgroup, not per row
        const int64_t b = r / G;
        const float* f = frames + r * D;
        const float* t = full + (b * T + idx[g]) * D;
        float s = 0.f;
        for (int64_t i = (int64_t)threadIdx.x * 4; i + 3 < D; i += 1024) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(f + i), c = *reinterpret_cast<const f32x4*>(t + i);
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = a[j] - c[j]; s += d * d; }
        }
        if (threadIdx.x == 0)
            for (int64_t i = D & ~(int64_t)3; i < D; ++i) { const float d = f[i] - t[i]; s += d * d; }
        if (g == 0) s0 += s; else s1 += s;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_down(s0, o, 64); s1 += __shfl_down(s1, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s0; red[1][threadIdx.x >> 6] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(sums, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(sums + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

// dframes[b,g,:] = coef[g == 0 ? 0 : 1] * (frames[b,g,:] - full[b,idx[g],:]);  coef on the device (2/N times the upstream gradient)
__global__ __launch_bounds__(256) void frames_sse_bwd_kernel(const float* frames, const float* full, const int* idx, int G, int T, int64_t D,
                                                             const float* coef, float* dframes) {
    const int g = blockIdx.y % G;
    const int64_t b = blockIdx.y / G;
    const float k = coef[g == 0 ? 0 : 1];
    const float* f = frames + (b * G + g) * D;
    const float* t = full + (b * T + idx[g]) * D;
    float* o = dframes + (b * G + g) * D;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i + 3 < D; i += (int64_t)gridDim.x * 1024) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(f + i), c = *reinterpret_cast<const f32x4*>(t + i);
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = k * (a[j] - c[j]);
        *reinterpret_cast<f32x4*>(o + i) = r;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = D & ~(int64_t)3; i < D; ++i) o[i] = k * (f[i] - t[i]);
}

}  // namespace

#include "vs_loss.h"
// ---- all four training losses in one pass (train.py:117-149) -------------------------------------------------------------
//   total = l_ae * mse(frame 0) + l_s * mean((s_old - s_new)^2) + l_pred * mse(frames 1..) + l_t * t_reg
//   t_reg = 0.5 * mean_b sum_c t0^2   (average_tloss: 0.5 * mean_{b,c} t0^2)
// Composed from torch ops this tail is ~30 launches of 2-5 us forward and as many backward -- 12 % of a WaveEq step that is
// otherwise MFMA/HBM work.  Forward: a zero fill, ONE pass (frame rows walked by all workgroups, the two small code terms by
// workgroup 0) and a 1-thread launch assembling the scalars; backward: ONE kernel writing every gradient.
namespace {
// out: [0..3] raw sums (SSE frame 0, SSE frames 1.., sum (s_old - s_new)^2, sum t0^2), [4] total, [5] ae, [6] zero, [7] pred,
//      [8] t_reg
// GRAD: the same pass also writes the gradients of `total` for an upstream gradient *g that is known when the forward runs (a
// recorded step passes its resident 1.0 / loss scale): dz = k (y - target) act'(y) in the compute dtype for the producing chain and
// the three small code gradients -- the frames and targets are read ONCE per step instead of twice (D % 4 == 0).
template <bool GRAD>
__global__ __launch_bounds__(256) void train_losses_fwd_kernel(LossArgs a, float* out, LossGrads gr) {
    __shared__ float red[4];
    float s0 = 0.f, s1 = 0.f;
    float up = 1.f;
    if constexpr (GRAD) up = gr.g[0];
    for (int64_t r = blockIdx.x; r < a.rows; r += gridDim.x) {
        const int g = (int)(r % a.G);
        const int64_t b = r / a.G;
        const float* f = a.frames + r * a.D;
        const float* t = a.full + (b * a.T + loss_target_frame(a, g)) * a.D;
        const float k = g == 0 ? up * a.l_ae * 2.f * a.inv_ae : up * a.l_pred * 2.f * a.inv_pred;
        float s = 0.f;
        int64_t i = (int64_t)threadIdx.x * 4;
        for (; i + 3 * 1024 + 3 < a.D; i += 4096) {            // four 16-byte loads of each stream in flight per thread
            f32x4 x[4], y[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { x[u] = *reinterpret_cast<const f32x4*>(f + i + u * 1024); y[u] = *reinterpret_cast<const f32x4*>(t + i + u * 1024); }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float rr[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = x[u][j] - y[u][j];
                    s += d * d;
                    if constexpr (GRAD) rr[j] = (k * d) * vs_act_grad_from_out(x[u][j], gr.act);
                }
                if constexpr (GRAD) {
                    const int64_t o = r * a.D + i + u * 1024;
                    if (gr.dz_dtype == VS_F32) *reinterpret_cast<f32x4*>((float*)gr.dz + o) = f32x4{rr[0], rr[1], rr[2], rr[3]};
                    else {
                        const u16x4 w = {vs_f2h(rr[0], gr.dz_dtype), vs_f2h(rr[1], gr.dz_dtype), vs_f2h(rr[2], gr.dz_dtype), vs_f2h(rr[3], gr.dz_dtype)};
                        *reinterpret_cast<u16x4*>((unsigned short*)gr.dz + o) = w;
                    }
                }
            }
        }
        for (; i + 3 < a.D; i += 1024) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(f + i), y = *reinterpret_cast<const f32x4*>(t + i);
            float rr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = x[j] - y[j];
                s += d * d;
                if constexpr (GRAD) rr[j] = (k * d) * vs_act_grad_from_out(x[j], gr.act);
            }
            if constexpr (GRAD) {
                const int64_t o = r * a.D + i;
                if (gr.dz_dtype == VS_F32) *reinterpret_cast<f32x4*>((float*)gr.dz + o) = f32x4{rr[0], rr[1], rr[2], rr[3]};
                else {
                    const u16x4 w = {vs_f2h(rr[0], gr.dz_dtype), vs_f2h(rr[1], gr.dz_dtype), vs_f2h(rr[2], gr.dz_dtype), vs_f2h(rr[3], gr.dz_dtype)};
                    *reinterpret_cast<u16x4*>((unsigned short*)gr.dz + o) = w;
                }
            }
        }
        if (threadIdx.x == 0)
            for (int64_t kk = a.D & ~(int64_t)3; kk < a.D; ++kk) { const float d = f[kk] - t[kk]; s += d * d; }      // (GRAD requires D % 4 == 0)
        if (g == 0) s0 += s; else s1 += s;
    }
    s0 = block_sum_256(s0, red);
    s1 = block_sum_256(s1, red);
    if constexpr (GRAD) {
        // per-workgroup partials (out[16 + 2 wg ..]), summed in workgroup order by the finalize launch: no float atomics (thousands of
        // them on two addresses serialise in L2) and a reproducible sum
        if (threadIdx.x == 0) { out[16 + 2 * blockIdx.x] = s0; out[17 + 2 * blockIdx.x] = s1; }
    } else {
        if (threadIdx.x == 0) { atomicAdd(out, s0); atomicAdd(out + 1, s1); }
    }
    if (blockIdx.x == 0) {                                     // the two code terms are tiny: one workgroup, fixed order
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
        ss = block_sum_256(ss, red);
        st = block_sum_256(st, red);
        if (threadIdx.x == 0) {
            if constexpr (GRAD) { out[2] = ss; out[3] = st; }     // one writer, and the frame sums travel as per-workgroup partials: nothing to clear first
            else { atomicAdd(out + 2, ss); atomicAdd(out + 3, st); }
        }
    }
}

// gradients of `total` times the upstream scalar *g: dframes, ds_old (= -ds_new), dt0
// dz != NULL: the frames are the output y of an activation (the decoder's last one) and the kernel writes the gradient of its
// INPUT, k * (y - target) * act'(y), in dz's dtype -- what ops.act_bwd would compute from dframes in a second pass
__global__ __launch_bounds__(256) void train_losses_bwd_kernel(LossArgs a, const float* g, float* dframes, float* ds_old, float* ds_new, float* dt0,
                                                               int act, void* dz, int dz_dtype) {
    const float up = g[0];
    if (blockIdx.y == (unsigned)a.rows) {                       // one extra grid row: the code gradients
        const float cs = up * a.l_s * 2.f * a.inv_s, ct = up * a.l_t * a.inv_t;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n_s; i += (int64_t)gridDim.x * 256) {
            const float d = cs * (a.s_old[i] - a.s_new[i]);
            ds_old[i] = d; ds_new[i] = -d;
        }
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.Bt * a.Ct; i += (int64_t)gridDim.x * 256) dt0[i] = ct * a.t0[i];
        return;
    }
    const int gidx = blockIdx.y % a.G;
    const int64_t b = blockIdx.y / a.G;
    const float k = gidx == 0 ? up * a.l_ae * 2.f * a.inv_ae : up * a.l_pred * 2.f * a.inv_pred;
    const float* f = a.frames + (b * a.G + gidx) * a.D;
    const float* t = a.full + (b * a.T + loss_target_frame(a, gidx)) * a.D;
    const int64_t row0 = (b * a.G + gidx) * a.D;
    if (dz) {
        for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i + 3 < a.D; i += (int64_t)gridDim.x * 1024) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(f + i), y = *reinterpret_cast<const f32x4*>(t + i);
            float r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = (k * (x[j] - y[j])) * vs_act_grad_from_out(x[j], act);
            if (dz_dtype == VS_F32) {
                *reinterpret_cast<f32x4*>((float*)dz + row0 + i) = f32x4{r[0], r[1], r[2], r[3]};
            } else {
                u16x4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = vs_f2h(r[j], dz_dtype);
                *reinterpret_cast<u16x4*>((unsigned short*)dz + row0 + i) = w;            // D % 4 tail handled below; row0 % 4 == 0 is checked by the host
            }
        }
        if (blockIdx.x == 0 && threadIdx.x == 0)
            for (int64_t i = a.D & ~(int64_t)3; i < a.D; ++i) vs_st(dz, dz_dtype, row0 + i, (k * (f[i] - t[i])) * vs_act_grad_from_out(f[i], act));
        return;
    }
    float* o = dframes + row0;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i + 3 < a.D; i += (int64_t)gridDim.x * 1024) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(f + i), y = *reinterpret_cast<const f32x4*>(t + i);
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = k * (x[j] - y[j]);
        *reinterpret_cast<f32x4*>(o + i) = r;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = a.D & ~(int64_t)3; i < a.D; ++i) o[i] = k * (f[i] - t[i]);
}

}  // namespace

extern "C" int vs_train_losses_fwd(const float* frames, const float* full, const int32_t* idx, const int32_t* t_random_dev, int ae_shift,
                                   int first_forecast, int64_t B, int G, int T, int64_t D,
                                   const float* s_old, const float* s_new, int64_t n_s, const float* t0, int64_t Bt, int64_t Ct,
                                   int average_tloss, const float* lambdas, float* out, void* stream) {
    LossArgs a;
    int rc = fill_loss_args(a, frames, full, idx, t_random_dev, ae_shift, first_forecast, B, G, T, D, s_old, s_new, n_s, t0, Bt, Ct, average_tloss,
                            lambdas);
    if (rc != VS_OK) return rc;
    VS_CHECK_ARG(out && frames, "vs_train_losses_fwd: null pointer");
    if (vs_zero_async(out, 10 * sizeof(float), (hipStream_t)stream) != hipSuccess) return vs_fail(VS_ERR_LAUNCH, "vs_train_losses_fwd: zero fill failed");
    int64_t wgs = B * G;
    if (wgs > 1024) wgs = 1024;
    hipLaunchKernelGGL(train_losses_fwd_kernel<false>, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, a, out, LossGrads{});
    hipLaunchKernelGGL(train_losses_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a, out, 0);
    VS_CHECK_LAUNCH("vs_train_losses_fwd");
    return VS_OK;
}

// vs_train_losses_fwd + vs_train_losses_bwd in ONE pass over the frames, for an upstream gradient that is known up front
// (`grad_total`: one float on the device, read by the kernel): out as vs_train_losses_fwd, dz / ds_old / ds_new / dt0 as
// vs_train_losses_bwd writes them (dz form only: the frames are the outputs of activation `frames_act`).
extern "C" int vs_train_losses_fwd_grad(const float* frames, const float* full, const int32_t* idx, const int32_t* t_random_dev, int ae_shift,
                                        int first_forecast, int64_t B, int G, int T, int64_t D, const float* s_old, const float* s_new, int64_t n_s,
                                        const float* t0, int64_t Bt, int64_t Ct, int average_tloss, const float* lambdas, float* out,
                                        const float* grad_total, float* ds_old, float* ds_new, float* dt0, int frames_act, void* dz, int dz_dtype,
                                        void* stream) {
    LossArgs a;
    int rc = fill_loss_args(a, frames, full, idx, t_random_dev, ae_shift, first_forecast, B, G, T, D, s_old, s_new, n_s, t0, Bt, Ct, average_tloss,
                            lambdas);
    if (rc != VS_OK) return rc;
    VS_CHECK_ARG(frames && out && grad_total && dz && dt0 && (n_s == 0 || (ds_old && ds_new)), "vs_train_losses_fwd_grad: null pointer");
    VS_CHECK_ARG(vs_dtype_ok(dz_dtype) && frames_act >= VS_ACT_NONE && frames_act <= VS_ACT_ELU && D % 4 == 0,
                 "vs_train_losses_fwd_grad: needs a valid dz dtype / activation and D %% 4 == 0");
    // (no clearing launch: out[0..8] are all written -- [2], [3] by workgroup 0, the rest by the finalize launch from the partials at out[16..])
    int64_t wgs = B * G;                                       // one row per workgroup: every load of the pass is issued at once
    if (wgs > VS_LOSS_MAX_PARTIALS) wgs = VS_LOSS_MAX_PARTIALS;
    LossGrads gr{grad_total, dz, dz_dtype, frames_act, ds_old, ds_new, dt0};
    hipLaunchKernelGGL(train_losses_fwd_kernel<true>, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, a, out, gr);
    hipLaunchKernelGGL(train_losses_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a, out, (int)wgs);
    VS_CHECK_LAUNCH("vs_train_losses_fwd_grad");
    return VS_OK;
}

extern "C" int vs_train_losses_bwd(const float* frames, const float* full, const int32_t* idx, const int32_t* t_random_dev, int ae_shift,
                                   int first_forecast, int64_t B, int G, int T, int64_t D,
                                   const float* s_old, const float* s_new, int64_t n_s, const float* t0, int64_t Bt, int64_t Ct,
                                   int average_tloss, const float* lambdas, const float* grad_total, float* dframes, float* ds_old,
                                   float* ds_new, float* dt0, int frames_act, void* dz, int dz_dtype, void* stream) {
    LossArgs a;
    int rc = fill_loss_args(a, frames, full, idx, t_random_dev, ae_shift, first_forecast, B, G, T, D, s_old, s_new, n_s, t0, Bt, Ct, average_tloss,
                            lambdas);
    if (rc != VS_OK) return rc;
    VS_CHECK_ARG(frames && grad_total && (dframes || dz) && dt0 && (n_s == 0 || (ds_old && ds_new)), "vs_train_losses_bwd: null pointer");
    VS_CHECK_ARG(!dz || (vs_dtype_ok(dz_dtype) && frames_act >= VS_ACT_NONE && frames_act <= VS_ACT_ELU && D % 4 == 0),
                 "vs_train_losses_bwd: dz needs a valid dtype / activation and D %% 4 == 0");
    unsigned gx = (unsigned)((D / 4 + 255) / 256);
    if (gx > 8) gx = 8;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(train_losses_bwd_kernel, dim3(gx, (unsigned)(B * G + 1)), dim3(256), 0, (hipStream_t)stream, a, grad_total, dframes, ds_old,
                       ds_new, dt0, frames_act, dz, dz_dtype);
    VS_CHECK_LAUNCH("vs_train_losses_bwd");
    return VS_OK;
}

// ---- batch assembly from an HBM-resident simulation set -------------------------------------------------------------------------
// Reference: data/wave_eq.py:67-72 (`WaveEq.__getitem__`: item idx -> sequence idx / per, first frame idx % per, window of
// seq_len frames) and :86-90 (`WaveEqPartial`: the same window restricted to n_pixels fixed pixels), applied to a whole batch
// of item indices.  At the step rates of this path (WaveEq: 128 x 25 frames per 1.8 ms = 29 GB/s of fp32 frames) a host
// DataLoader cannot feed the GPU; the whole normalised set stays in HBM and a batch is one gather launch driven by the
// sampler's indices.
namespace {
__global__ __launch_bounds__(256) void gather_windows_kernel(const float* data, int64_t nt, int64_t frame, const int32_t* item, int per,
                                                             int seq_len, const int32_t* pix, int n_pix, void* out, int od) {
    const int b = blockIdx.y;
    const int idx = item[b];
    const int64_t seq = idx / per, first = idx - seq * per;
    const float* src = data + (seq * nt + first) * frame;            // seq_len consecutive frames
    const int64_t width = pix ? n_pix : frame;
    const int64_t total = (int64_t)seq_len * width;
    const int64_t obase = (int64_t)b * total;
    if (!pix && (frame & 3) == 0) {                                  // whole frames: a contiguous run of seq_len * frame floats
        for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (int64_t)gridDim.x * 1024) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + i);
            if (od == VS_F32) *reinterpret_cast<f32x4*>((float*)out + obase + i) = v;
            else { const u16x4 w = {vs_f2h(v[0], od), vs_f2h(v[1], od), vs_f2h(v[2], od), vs_f2h(v[3], od)}; *reinterpret_cast<u16x4*>((unsigned short*)out + obase + i) = w; }
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t t = i / width, p = i - t * width;
        vs_st(out, od, obase + i, src[t * frame + (pix ? pix[p] : p)]);
    }
}
}  // namespace

extern "C" int vs_gather_windows(const float* data, int64_t n_seq, int64_t nt, int64_t frame_elems, const int32_t* item_idx, int batch,
                                 int windows_per_seq, int seq_len, const int32_t* pixel_idx, int n_pixels, void* out, int out_dtype,
                                 void* stream) {
    VS_CHECK_ARG(data && item_idx && out && n_seq > 0 && nt > 0 && frame_elems > 0 && batch > 0 && windows_per_seq > 0 && seq_len > 0 &&
                     seq_len <= nt, "vs_gather_windows: bad argument");
    VS_CHECK_ARG(windows_per_seq + seq_len - 1 <= nt, "vs_gather_windows: a window would run past the end of its sequence");
    VS_CHECK_ARG(!pixel_idx || n_pixels > 0, "vs_gather_windows: pixel table without a count");
    VS_CHECK_ARG(vs_dtype_ok(out_dtype), "vs_gather_windows: bad out_dtype");
    const int64_t total = (int64_t)seq_len * (pixel_idx ? n_pixels : frame_elems);
    int64_t gx = (total / 4 + 255) / 256;
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(gather_windows_kernel, dim3((unsigned)gx, (unsigned)batch), dim3(256), 0, (hipStream_t)stream, data, nt, frame_elems,
                       item_idx, windows_per_seq, seq_len, pixel_idx, n_pixels, out, out_dtype);
    VS_CHECK_LAUNCH("vs_gather_windows");
    return VS_OK;
}

// ---- decoder input of a whole rollout: z[b, g, :] = mix(s[b, :], t[b, g, :]) with t = [t_rand ; t_codes] -----------------------
// Reference: mlp_encdec.py:43-48 (`torch.cat([z1, z2], dim=1)` or `z1 * z2`) applied to the auto-encoding pair and to every
// rollout step (model.py:74-83).  From torch ops this is expand + cat + mul + cast forward and mul, mul, sum-over-frames, slice
// and cat gradients backward (~14 launches between the decoder and the integrator); here one launch each way.
namespace {
// mixing 0: concat (Cz = Cs + Ct), 1: mul (Cz = Cs = Ct)
__global__ __launch_bounds__(256) void mix_codes_fwd_kernel(const float* s, const float* t_rand, const float* t_codes, int64_t B, int n, int Cs,
                                                            int Ct, int mixing, float* out, unsigned short* out_lp, int lp_dtype) {
    const int G = n + 1, Cz = mixing ? Cs : Cs + Ct;
    const int64_t total = B * G * Cz;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cz);
        const int64_t row = i / Cz;
        const int g = (int)(row % G);
        const int64_t b = row / G;
        float v;
        if (mixing) {
            const float tv = g == 0 ? t_rand[b * Ct + c] : t_codes[(b * n + g - 1) * Ct + c];
            v = s[b * Cs + c] * tv;
        } else if (c < Cs) {
            v = s[b * Cs + c];
        } else {
            v = g == 0 ? t_rand[b * Ct + c - Cs] : t_codes[(b * n + g - 1) * Ct + c - Cs];
        }
        out[i] = v;
        if (out_lp) out_lp[i] = vs_f2h(v, lp_dtype);
    }
}

// one workgroup per sample: thread c walks the G frames of column c (ds needs the sum over the frames, in frame order)
__global__ __launch_bounds__(256) void mix_codes_bwd_kernel(const float* dz, const float* s, const float* t_rand, const float* t_codes, int64_t B,
                                                            int n, int Cs, int Ct, int mixing, float* ds, float* dt_rand, float* dt_codes) {
    const int G = n + 1, Cz = mixing ? Cs : Cs + Ct;
    const int64_t b = blockIdx.x;
    for (int c = threadIdx.x; c < Cz; c += 256) {
        const float* dzp = dz + (b * G) * Cz + c;
        if (mixing) {
            const float sv = s[b * Cs + c];
            float acc = 0.f;
            for (int g = 0; g < G; ++g) {
                const float d = dzp[(int64_t)g * Cz];
                const float tv = g == 0 ? t_rand[b * Ct + c] : t_codes[(b * n + g - 1) * Ct + c];
                acc += d * tv;
                if (g == 0) dt_rand[b * Ct + c] = d * sv; else dt_codes[(b * n + g - 1) * Ct + c] = d * sv;
            }
            ds[b * Cs + c] = acc;
        } else if (c < Cs) {
            float acc = 0.f;
            for (int g = 0; g < G; ++g) acc += dzp[(int64_t)g * Cz];
            ds[b * Cs + c] = acc;
        } else {
            for (int g = 0; g < G; ++g) {
                const float d = dzp[(int64_t)g * Cz];
                if (g == 0) dt_rand[b * Ct + c - Cs] = d; else dt_codes[(b * n + g - 1) * Ct + c - Cs] = d;
            }
        }
    }
}
}  // namespace

extern "C" int vs_mix_codes_fwd(const float* s, const float* t_rand, const float* t_codes, int64_t B, int n, int Cs, int Ct, int mixing, float* out,
                                void* out_lowp, int lowp_dtype, void* stream) {
    VS_CHECK_ARG(!out_lowp || vs_is16(lowp_dtype), "vs_mix_codes_fwd: the second output is a 16-bit copy (VS_BF16 | VS_F16)");
    VS_CHECK_ARG(s && t_rand && (t_codes || n == 0) && out && B > 0 && n >= 0 && Cs > 0 && Ct > 0, "vs_mix_codes_fwd: bad argument");
    VS_CHECK_ARG((mixing == 0 || mixing == 1) && (mixing == 0 || Cs == Ct), "vs_mix_codes_fwd: mixing 1 (mul) needs Cs == Ct");
    const int64_t total = B * (n + 1) * (mixing ? Cs : Cs + Ct);
    hipLaunchKernelGGL(mix_codes_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, s, t_rand, t_codes, B, n, Cs, Ct, mixing, out,
                       (unsigned short*)out_lowp, lowp_dtype);
    VS_CHECK_LAUNCH("vs_mix_codes_fwd");
    return VS_OK;
}

extern "C" int vs_mix_codes_bwd(const float* dz, const float* s, const float* t_rand, const float* t_codes, int64_t B, int n, int Cs, int Ct,
                                int mixing, float* ds, float* dt_rand, float* dt_codes, void* stream) {
    VS_CHECK_ARG(dz && s && t_rand && (t_codes || n == 0) && ds && dt_rand && (dt_codes || n == 0) && B > 0 && n >= 0 && Cs > 0 && Ct > 0,
                 "vs_mix_codes_bwd: bad argument");
    VS_CHECK_ARG((mixing == 0 || mixing == 1) && (mixing == 0 || Cs == Ct), "vs_mix_codes_bwd: mixing 1 (mul) needs Cs == Ct");
    hipLaunchKernelGGL(mix_codes_bwd_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, dz, s, t_rand, t_codes, B, n, Cs, Ct, mixing, ds,
                       dt_rand, dt_codes);
    VS_CHECK_LAUNCH("vs_mix_codes_bwd");
    return VS_OK;
}

extern "C" int vs_frames_sse_fwd(const float* frames, const float* full, const int32_t* idx, int64_t B, int G, int T, int64_t D, float* sums,
                                 void* stream) {
    VS_CHECK_ARG(frames && full && idx && sums && B > 0 && G > 0 && T > 0 && D > 0, "vs_frames_sse_fwd: bad argument");
    if (vs_zero_async(sums, 2 * sizeof(float), (hipStream_t)stream) != hipSuccess) return vs_fail(VS_ERR_LAUNCH, "vs_frames_sse_fwd: memset failed");
    int64_t wgs = B * G;
    if (wgs > 1024) wgs = 1024;
    hipLaunchKernelGGL(frames_sse_kernel, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, frames, full, idx, B * G, G, T, D, sums);
    VS_CHECK_LAUNCH("vs_frames_sse_fwd");
    return VS_OK;
}

extern "C" int vs_frames_sse_bwd(const float* frames, const float* full, const int32_t* idx, int64_t B, int G, int T, int64_t D, const float* coef,
                                 float* dframes, void* stream) {
    VS_CHECK_ARG(frames && full && idx && coef && dframes && B > 0 && G > 0 && T > 0 && D > 0, "vs_frames_sse_bwd: bad argument");
    unsigned gx = (unsigned)((D / 4 + 255) / 256);
    if (gx > 8) gx = 8;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(frames_sse_bwd_kernel, dim3(gx, (unsigned)(B * G)), dim3(256), 0, (hipStream_t)stream, frames, full, idx, G, T, D, coef,
                       dframes);
    VS_CHECK_LAUNCH("vs_frames_sse_bwd");
    return VS_OK;
}

// ---- the conv families' code losses and the weighted total in one pass (reference train.py:38-42, 141-149) ---------------------------------
// zero-order loss mean((s_old - s_new)^2) -- with skip connections over the code AND every skip tensor (element-count weighted, i.e. one mean
// over their concatenation) --, the temporal regulariser 0.5 sum t0^2 / N_t and total = l_ae ae + l_s zero + l_pred pred + l_t t_reg, with ae /
// pred taken from the frame kernels' raw sums.  As torch ops that is two concatenations of all skips (SST: 2 x 16 MB), sub / pow / mean, a dozen
// scalar launches and as many again backward -- ~45 launches of 4-50 us per step.  Here: the pairs are read where they lie (fp32 or 16-bit),
// one workgroup per 4096-element chunk leaves a partial sum, a one-block launch finishes (fixed order: reproducible); backward is ONE launch
// that writes d s_old / d s_new in their own types, d t0, and the two coefficients vs_frames_sse_bwd takes.
namespace {
constexpr int CL_MAX_PAIRS = 10, CL_CHUNK = 4096;
struct CodeLossJobs {
    const void* a[CL_MAX_PAIRS];
    const void* b[CL_MAX_PAIRS];
    void* da[CL_MAX_PAIRS];
    void* db[CL_MAX_PAIRS];
    int dtype[CL_MAX_PAIRS];
    int64_t count[CL_MAX_PAIRS];
    int first_chunk[CL_MAX_PAIRS + 1];          // chunks [first_chunk[j], first_chunk[j + 1]) belong to pair j
    int n_pairs;
    const float* t0;
    float* dt0;
    int64_t t_count;
    int n_chunks;                               // pair chunks, then the chunks of t0
};

__device__ __forceinline__ void cl_load8(const void* p, int dtype, int64_t i, float (&v)[8]) {
    if (dtype == VS_F32) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>((const float*)p + i), hi = *reinterpret_cast<const f32x4*>((const float*)p + i + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[j + 4] = hi[j]; }
    } else {
        const u16x8 r = *reinterpret_cast<const u16x8*>((const unsigned short*)p + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = vs_h2f(r[j], dtype);
    }
}
__device__ __forceinline__ void cl_store8(void* p, int dtype, int64_t i, const float (&v)[8]) {
    if (dtype == VS_F32) {
        f32x4 lo, hi;
#pragma unroll
        for (int j = 0; j < 4; ++j) { lo[j] = v[j]; hi[j] = v[j + 4]; }
        *reinterpret_cast<f32x4*>((float*)p + i) = lo;
        *reinterpret_cast<f32x4*>((float*)p + i + 4) = hi;
    } else {
        u16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = vs_f2h(v[j], dtype);
        *reinterpret_cast<u16x8*>((unsigned short*)p + i) = r;
    }
}

__global__ __launch_bounds__(256) void code_losses_partial_kernel(CodeLossJobs J, float* __restrict__ partial) {
    __shared__ float red[4];
    const int chunk = blockIdx.x;
    float s = 0.f;
    if (chunk < J.first_chunk[J.n_pairs]) {
        int j = 0;
        while (chunk >= J.first_chunk[j + 1]) ++j;
        const int64_t base = (int64_t)(chunk - J.first_chunk[j]) * CL_CHUNK;
#pragma unroll
        for (int r = 0; r < CL_CHUNK / (256 * 8); ++r) {
            const int64_t i = base + (int64_t)(r * 256 + threadIdx.x) * 8;
            if (i < J.count[j]) {                                            // (count is a multiple of 8)
                float a[8], b[8];
                cl_load8(J.a[j], J.dtype[j], i, a);
                cl_load8(J.b[j], J.dtype[j], i, b);
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float d = a[k] - b[k]; s += d * d; }
            }
        }
    } else {
        const int64_t base = (int64_t)(chunk - J.first_chunk[J.n_pairs]) * CL_CHUNK;
        for (int e = threadIdx.x; e < CL_CHUNK; e += 256) {
            const int64_t i = base + e;
            if (i < J.t_count) { const float t = J.t0[i]; s += t * t; }
        }
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) partial[chunk] = s;
}

struct CodeLossScal { float l_ae, l_s, l_pred, l_t, scale_ae, scale_pred, inv_s, inv_t; };

// out[0..4] = total, ae, zero, pred, t_reg
__global__ __launch_bounds__(256) void code_losses_finish_kernel(const float* __restrict__ partial, int n_pair_chunks, int n_chunks, const float* sse_ae,
                                                                 const float* sse_pred, CodeLossScal c, float* __restrict__ out) {
    __shared__ double red[2][256];
    double z = 0.0, t = 0.0;
    for (int i = threadIdx.x; i < n_pair_chunks; i += 256) z += (double)partial[i];
    for (int i = n_pair_chunks + threadIdx.x; i < n_chunks; i += 256) t += (double)partial[i];
    red[0][threadIdx.x] = z;
    red[1][threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 256; ++i) { z += red[0][i]; t += red[1][i]; }     // fixed order
        const float ae = (sse_ae[0] + sse_ae[1]) * c.scale_ae, pred = (sse_pred[0] + sse_pred[1]) * c.scale_pred;
        const float zero = (float)(z * (double)c.inv_s), treg = (float)(0.5 * t * (double)c.inv_t);
        out[0] = c.l_ae * ae + c.l_s * zero + c.l_pred * pred + c.l_t * treg;
        out[1] = ae;
        out[2] = zero;
        out[3] = pred;
        out[4] = treg;
    }
}

// coefs[0..1] = g l_ae 2 scale_ae (the auto-encoding frame stack), coefs[2..3] = g l_pred 2 scale_pred: what vs_frames_sse_bwd multiplies (y - target) by
__global__ __launch_bounds__(256) void code_losses_bwd_kernel(CodeLossJobs J, const float* __restrict__ g, CodeLossScal c, float* __restrict__ coefs) {
    const int chunk = blockIdx.x;
    const float up = g[0];
    if (chunk == 0 && threadIdx.x == 0) {
        coefs[0] = coefs[1] = up * c.l_ae * 2.f * c.scale_ae;
        coefs[2] = coefs[3] = up * c.l_pred * 2.f * c.scale_pred;
    }
    if (chunk < J.first_chunk[J.n_pairs]) {
        int j = 0;
        while (chunk >= J.first_chunk[j + 1]) ++j;
        const float k = up * c.l_s * 2.f * c.inv_s;
        const int64_t base = (int64_t)(chunk - J.first_chunk[j]) * CL_CHUNK;
#pragma unroll
        for (int r = 0; r < CL_CHUNK / (256 * 8); ++r) {
            const int64_t i = base + (int64_t)(r * 256 + threadIdx.x) * 8;
            if (i < J.count[j]) {
                float a[8], b[8], d[8], e[8];
                cl_load8(J.a[j], J.dtype[j], i, a);
                cl_load8(J.b[j], J.dtype[j], i, b);
#pragma unroll
                for (int q = 0; q < 8; ++q) { d[q] = k * (a[q] - b[q]); e[q] = -d[q]; }
                if (J.da[j]) cl_store8(J.da[j], J.dtype[j], i, d);
                if (J.db[j]) cl_store8(J.db[j], J.dtype[j], i, e);
            }
        }
    } else if (J.dt0) {
        const float k = up * c.l_t * c.inv_t;                                    // d/dt of 0.5 t^2 / N_t
        const int64_t base = (int64_t)(chunk - J.first_chunk[J.n_pairs]) * CL_CHUNK;
        for (int e = threadIdx.x; e < CL_CHUNK; e += 256) {
            const int64_t i = base + e;
            if (i < J.t_count) J.dt0[i] = k * J.t0[i];
        }
    }
}

int cl_fill(CodeLossJobs& J, int n_pairs, const void* const* a, const void* const* b, void* const* da, void* const* db, const int* dtype, const int64_t* count,
            const float* t0, float* dt0, int64_t t_count) {
    if (n_pairs < 0 || n_pairs > CL_MAX_PAIRS || t_count < 0 || (t_count > 0 && !t0)) return -1;
    J.n_pairs = n_pairs;
    int64_t chunks = 0;
    for (int j = 0; j < n_pairs; ++j) {
        if (!a[j] || !b[j] || !vs_dtype_ok(dtype[j]) || count[j] <= 0 || count[j] % 8 != 0 || ((uintptr_t)a[j] | (uintptr_t)b[j]) % 16 != 0) return -1;
        if (da && da[j] && (uintptr_t)da[j] % 16 != 0) return -1;
        if (db && db[j] && (uintptr_t)db[j] % 16 != 0) return -1;
        J.a[j] = a[j]; J.b[j] = b[j]; J.da[j] = da ? da[j] : nullptr; J.db[j] = db ? db[j] : nullptr;
        J.dtype[j] = dtype[j]; J.count[j] = count[j];
        J.first_chunk[j] = (int)chunks;
        chunks += vs_cdiv(count[j], CL_CHUNK);
    }
    J.first_chunk[n_pairs] = (int)chunks;
    J.t0 = t0; J.dt0 = dt0; J.t_count = t_count;
    chunks += vs_cdiv(t_count, CL_CHUNK);
    if (chunks < 1 || chunks > (1 << 22)) return -1;
    J.n_chunks = (int)chunks;
    return 0;
}
}  // namespace

// ---- decoder inputs of a batched rollout: cat([a repeated over the n frames, x], dim 1) in one pass (reference conv.py:228, 388-394) ----------
// The SST decoder runs its n frame calls as one batch of n B maps; every skip tensor (and the spatial code) of the B sequences is the same for
// all frames.  torch builds `a.repeat(n, 1, 1, 1)` (n copies written) and then `cat` (read again, written again): 0.6 GB per SST step.  Here
// out[f B + b][c] = c < Ca ? a[b][c] : x[f B + b][c - Ca], converted to out's type, one write.  Backward: da[b][c] = sum over the frames of
// dout[f B + b][c] (fp32 accumulation, frame order), dx = the other channels -- one pass over dout.
namespace {
__global__ __launch_bounds__(256) void cat_bcast_fwd_kernel(const void* __restrict__ a, int ad, const void* __restrict__ x, int xd, void* __restrict__ out,
                                                            int od, int B, int Ca, int Cb, int64_t HW8, int64_t total8) {
    // one 8-element piece per thread (HW a multiple of 8)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (int64_t)gridDim.x * 256) {
        const int64_t p = i % HW8, mc = i / HW8;
        const int C = Ca + Cb;
        const int c = (int)(mc % C);
        const int64_t m = mc / C;                                                // map f B + b
        float v[8];
        if (c < Ca) cl_load8(a, ad, (((m % B) * Ca + c) * HW8 + p) * 8, v);
        else cl_load8(x, xd, ((m * Cb + (c - Ca)) * HW8 + p) * 8, v);
        cl_store8(out, od, i * 8, v);
    }
}

__global__ __launch_bounds__(256) void cat_bcast_bwd_kernel(const void* __restrict__ dout, int dd, void* __restrict__ da, int ad, void* __restrict__ dx, int xd,
                                                            int B, int n, int Ca, int Cb, int64_t HW8, int64_t total8) {
    // thread space: [x part: n B Cb HW8 pieces] then [a part: B Ca HW8 pieces]
    const int C = Ca + Cb;
    const int64_t x8 = (int64_t)n * B * Cb * HW8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (int64_t)gridDim.x * 256) {
        float v[8];
        if (i < x8) {
            if (!dx) continue;
            const int64_t p = i % HW8, mc = i / HW8;
            const int c = (int)(mc % Cb);
            const int64_t m = mc / Cb;
            cl_load8(dout, dd, ((m * C + Ca + c) * HW8 + p) * 8, v);
            cl_store8(dx, xd, i * 8, v);
        } else {
            if (!da) continue;
            const int64_t j = i - x8, p = j % HW8, bc = j / HW8;
            const int c = (int)(bc % Ca);
            const int64_t b = bc / Ca;
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int f = 0; f < n; ++f) {                                         // frame order: reproducible
                cl_load8(dout, dd, ((((int64_t)f * B + b) * C + c) * HW8 + p) * 8, v);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += v[k];
            }
            cl_store8(da, ad, j * 8, acc);
        }
    }
}
}  // namespace

extern "C" int vs_cat_bcast_fwd(const void* a, int a_dtype, const void* x, int x_dtype, void* out, int out_dtype, int B, int n, int Ca, int Cb, int64_t HW,
                                void* stream) {
    VS_CHECK_ARG(a && x && out && vs_dtype_ok(a_dtype) && vs_dtype_ok(x_dtype) && vs_dtype_ok(out_dtype) && B > 0 && n > 0 && Ca > 0 && Cb > 0 && HW > 0 &&
                     HW % 8 == 0 && ((uintptr_t)a | (uintptr_t)x | (uintptr_t)out) % 16 == 0,
                 "vs_cat_bcast_fwd: bad argument (planes of a multiple of 8 elements, 16-byte aligned tensors)");
    const int64_t total8 = (int64_t)n * B * (Ca + Cb) * (HW / 8);
    hipLaunchKernelGGL(cat_bcast_fwd_kernel, dim3(grid_for(total8)), dim3(256), 0, (hipStream_t)stream, a, a_dtype, x, x_dtype, out, out_dtype, B, Ca, Cb, HW / 8,
                       total8);
    VS_CHECK_LAUNCH("vs_cat_bcast_fwd");
    return VS_OK;
}

extern "C" int vs_cat_bcast_bwd(const void* dout, int dout_dtype, void* da, int a_dtype, void* dx, int x_dtype, int B, int n, int Ca, int Cb, int64_t HW,
                                void* stream) {
    VS_CHECK_ARG(dout && (da || dx) && vs_dtype_ok(dout_dtype) && vs_dtype_ok(a_dtype) && vs_dtype_ok(x_dtype) && B > 0 && n > 0 && Ca > 0 && Cb > 0 && HW > 0 &&
                     HW % 8 == 0 && ((uintptr_t)dout | (uintptr_t)da | (uintptr_t)dx) % 16 == 0,
                 "vs_cat_bcast_bwd: bad argument (planes of a multiple of 8 elements, 16-byte aligned tensors)");
    const int64_t total8 = ((int64_t)n * B * Cb + (int64_t)B * Ca) * (HW / 8);
    hipLaunchKernelGGL(cat_bcast_bwd_kernel, dim3(grid_for(total8)), dim3(256), 0, (hipStream_t)stream, dout, dout_dtype, da, a_dtype, dx, x_dtype, B, n, Ca, Cb,
                       HW / 8, total8);
    VS_CHECK_LAUNCH("vs_cat_bcast_bwd");
    return VS_OK;
}

// chunks = rows of `partial` vs_code_losses_fwd needs
extern "C" int64_t vs_code_losses_chunks(int n_pairs, const int64_t* count, int64_t t_count) {
    int64_t c = vs_cdiv(t_count > 0 ? t_count : 0, CL_CHUNK);
    for (int j = 0; j < n_pairs; ++j) c += vs_cdiv(count[j], CL_CHUNK);
    return c;
}

// out[0..4] = total, ae, zero, pred, t_reg.  Pairs (a[j], b[j]): count[j] elements (a multiple of 8) of dtype[j], 16-byte aligned; t0: t_count
// fp32 elements; sse_ae / sse_pred: the [2] raw sums of vs_frames_sse_fwd; scale_* = 1 / elements of each frame mean; inv_s = 1 / sum of the
// pair counts; inv_t = 1 / N_t (train.py:141-146: B x positions, or all elements with average_tloss).
extern "C" int vs_code_losses_fwd(int n_pairs, const void* const* a, const void* const* b, const int* dtype, const int64_t* count, const float* t0,
                                  int64_t t_count, const float* sse_ae, const float* sse_pred, float scale_ae, float scale_pred, float l_ae, float l_s,
                                  float l_pred, float l_t, float inv_s, float inv_t, float* partial, float* out, void* stream) {
    CodeLossJobs J = {};
    VS_CHECK_ARG(sse_ae && sse_pred && partial && out && cl_fill(J, n_pairs, a, b, nullptr, nullptr, dtype, count, t0, nullptr, t_count) == 0,
                 "vs_code_losses_fwd: bad argument (<= 10 pairs, counts multiples of 8, 16-byte aligned)");
    const CodeLossScal c = {l_ae, l_s, l_pred, l_t, scale_ae, scale_pred, inv_s, inv_t};
    hipLaunchKernelGGL(code_losses_partial_kernel, dim3((unsigned)J.n_chunks), dim3(256), 0, (hipStream_t)stream, J, partial);
    hipLaunchKernelGGL(code_losses_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, J.first_chunk[n_pairs], J.n_chunks, sse_ae, sse_pred, c,
                       out);
    VS_CHECK_LAUNCH("vs_code_losses_fwd");
    return VS_OK;
}

// Gradients of `total` for the upstream gradient g[0] (device): da[j] / db[j] (either may be NULL) in dtype[j], dt0 (fp32, may be NULL), and
// coefs[4] = the coefficient pairs vs_frames_sse_bwd takes for the auto-encoding / forecast frame stacks.
extern "C" int vs_code_losses_bwd(int n_pairs, const void* const* a, const void* const* b, void* const* da, void* const* db, const int* dtype,
                                  const int64_t* count, const float* t0, float* dt0, int64_t t_count, const float* g, float scale_ae, float scale_pred,
                                  float l_ae, float l_s, float l_pred, float l_t, float inv_s, float inv_t, float* coefs, void* stream) {
    CodeLossJobs J = {};
    VS_CHECK_ARG(g && coefs && cl_fill(J, n_pairs, a, b, da, db, dtype, count, t0, dt0, t_count) == 0,
                 "vs_code_losses_bwd: bad argument (<= 10 pairs, counts multiples of 8, 16-byte aligned)");
    const CodeLossScal c = {l_ae, l_s, l_pred, l_t, scale_ae, scale_pred, inv_s, inv_t};
    hipLaunchKernelGGL(code_losses_bwd_kernel, dim3((unsigned)J.n_chunks), dim3(256), 0, (hipStream_t)stream, J, g, c, coefs);
    VS_CHECK_LAUNCH("vs_code_losses_bwd");
    return VS_OK;
}

extern "C" int vs_colsum_multi(int n_jobs, const void* const* X, const int* x_dtype, const int64_t* ldx, const int64_t* M, const int64_t* N,
                               float* const* out, float* zero_base, int64_t zero_count, void* stream) {
    VS_CHECK_ARG(n_jobs >= 1 && n_jobs <= CS_MAXJ && X && x_dtype && ldx && M && N && out, "vs_colsum_multi: bad argument (1..%d jobs)", CS_MAXJ);
    ColsumJobs J;
    J.nj = n_jobs;
    int64_t max_m = 0;
    J.blk_off[0] = 0;
    for (int j = 0; j < n_jobs; ++j) {
        VS_CHECK_ARG(X[j] && out[j] && M[j] > 0 && N[j] > 0 && ldx[j] >= N[j], "vs_colsum_multi: bad job %d", j);
        J.x[j] = X[j]; J.dt[j] = x_dtype[j]; J.ld[j] = ldx[j]; J.M[j] = M[j]; J.N[j] = N[j]; J.out[j] = out[j];
        J.blk_off[j + 1] = J.blk_off[j] + (int)vs_cdiv(N[j], CSM_COLS);
        if (M[j] > max_m) max_m = M[j];
    }
    dim3 grid((unsigned)J.blk_off[n_jobs], (unsigned)vs_cdiv(max_m, CS_ROWS));
    // Every job within ONE row block (the encoders' bias gradients: 256 rows) and the outputs tiling exactly the area this call is asked to
    // clear: each column sum has one writer, so it is stored instead of added and the clearing launch is not issued (one kernel boundary
    // less on the critical path of the WaveEq step's tail).
    J.direct = 0;
    if (zero_base && zero_count > 0 && grid.y == 1) {
        int64_t cols = 0;
        bool inside = true;
        for (int j = 0; j < n_jobs; ++j) {
            cols += N[j];
            if (out[j] < zero_base || out[j] + N[j] > zero_base + zero_count) inside = false;
        }
        J.direct = inside && cols == zero_count;
    }
    if (zero_base && zero_count > 0 && !J.direct) {
        if (vs_zero_async(zero_base, (size_t)zero_count * sizeof(float), (hipStream_t)stream) != hipSuccess)
            return vs_fail(VS_ERR_LAUNCH, "vs_colsum_multi: memset failed");
    }
    hipLaunchKernelGGL(colsum_multi_kernel, grid, dim3(256), 0, (hipStream_t)stream, J);
    VS_CHECK_LAUNCH("vs_colsum_multi");
    return VS_OK;
}

extern "C" int vs_cast(const void* src, int sd, void* dst, int dd, int64_t n, void* stream) {
    VS_CHECK_ARG(src && dst && n >= 0, "vs_cast: bad argument");
    VS_CHECK_ARG(vs_dtype_ok(sd) && vs_dtype_ok(dd), "vs_cast: bad dtype");
    if (n == 0) return VS_OK;
    hipLaunchKernelGGL(cast_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, src, sd, dst, dd, n);
    VS_CHECK_LAUNCH("vs_cast");
    return VS_OK;
}

extern "C" int vs_act_fwd(const void* x, int xd, void* y, int yd, int act, int64_t n, void* stream) {
    VS_CHECK_ARG(x && y && n >= 0, "vs_act_fwd: bad argument");
    if (n == 0) return VS_OK;
    hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, x, xd, y, yd, act, n);
    VS_CHECK_LAUNCH("vs_act_fwd");
    return VS_OK;
}

extern "C" int vs_act_bwd(const void* dy, int dyd, const void* y, int yd, void* dz, int dzd, int act, int64_t n, void* stream) {
    VS_CHECK_ARG(dy && y && dz && n >= 0, "vs_act_bwd: bad argument");
    if (n == 0) return VS_OK;
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, (hipStream_t)stream, dy, dyd, y, yd, dz, dzd, act, n);
    VS_CHECK_LAUNCH("vs_act_bwd");
    return VS_OK;
}

extern "C" int vs_copy2d(const void* src, int sd, int64_t lds, void* dst, int dd, int64_t ldd, int64_t rows, int64_t cols,
                         const int32_t* col_offset_dev, int64_t col_offset_scale, void* stream) {
    VS_CHECK_ARG(src && dst && rows >= 0 && cols >= 0, "vs_copy2d: bad argument");
    if (rows * cols == 0) return VS_OK;
    hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, (hipStream_t)stream, src, sd, lds, dst, dd, ldd,
                       rows, cols, col_offset_dev, col_offset_scale);
    VS_CHECK_LAUNCH("vs_copy2d");
    return VS_OK;
}

extern "C" int vs_copy2d_pair(const void* src, int sd, int64_t lds, void* dst, int dd, int64_t ldd, int64_t rows, int64_t cols,
                              const int32_t* col_offset_dev, int64_t col_offset_scale, int64_t elem_offset_a, int64_t elem_offset_b, void* stream) {
    VS_CHECK_ARG(src && dst && rows >= 0 && cols >= 0 && vs_dtype_ok(sd) && vs_dtype_ok(dd), "vs_copy2d_pair: bad argument");
    if (rows * cols == 0) return VS_OK;
    const int se = vs_esize(sd), de = vs_esize(dd);
    const bool vec = cols % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0 && col_offset_scale % 4 == 0 && elem_offset_a % 4 == 0 && elem_offset_b % 4 == 0 &&
                     (uintptr_t)src % (4 * se) == 0 && (uintptr_t)dst % (4 * de) == 0;
    if (vec)
        hipLaunchKernelGGL(copy2d_pair_kernel<true>, dim3(grid_for(2 * rows * cols / 4)), dim3(256), 0, (hipStream_t)stream, src, sd, lds, dst, dd, ldd, rows, cols,
                           col_offset_dev, col_offset_scale, elem_offset_a, elem_offset_b);
    else
        hipLaunchKernelGGL(copy2d_pair_kernel<false>, dim3(grid_for(2 * rows * cols)), dim3(256), 0, (hipStream_t)stream, src, sd, lds, dst, dd, ldd, rows, cols,
                           col_offset_dev, col_offset_scale, elem_offset_a, elem_offset_b);
    VS_CHECK_LAUNCH("vs_copy2d_pair");
    return VS_OK;
}

extern "C" int vs_colsum(const void* X, int xd, int64_t ldx, int64_t M, int64_t N, float* out, int accumulate, void* stream) {
    VS_CHECK_ARG(X && out && M > 0 && N > 0 && ldx >= N, "vs_colsum: bad argument");
    if (!accumulate) {
        if (vs_zero_async(out, (size_t)N * sizeof(float), (hipStream_t)stream) != hipSuccess)
            return vs_fail(VS_ERR_LAUNCH, "vs_colsum: memset failed");
    }
    dim3 grid((unsigned)vs_cdiv(N, 64), (unsigned)vs_cdiv(M, CS_ROWS));
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, X, xd, ldx, M, N, out);
    VS_CHECK_LAUNCH("vs_colsum");
    return VS_OK;
}
