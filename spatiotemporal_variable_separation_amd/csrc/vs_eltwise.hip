// vs_eltwise.hip -- HBM-bound helpers around the GEMMs: casts, strided window copy, bias-gradient column
// sums, activation forward/backward.  All are grid-stride kernels with 16-byte accesses on the contiguous
// path (8 bf16 / 4 fp32 per lane), capped at 2048 workgroups of 256 threads (8 per CU on 256 CUs).
#include <stdarg.h>
#include "vs_common.h"

thread_local char vs_err_buf[256] = "";

int vs_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(vs_err_buf, sizeof(vs_err_buf), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* vs_last_error(void) { return vs_err_buf; }
extern "C" const char* vs_version(void) { return "varsep_hip 0.1 (gfx950)"; }

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
        else { bf16x4 t = ((const bf16x4*)x)[i]; v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = f(v[j], i * 4 + j);
        if (yd == VS_F32) { f32x4 t = {v[0], v[1], v[2], v[3]}; ((f32x4*)y)[i] = t; }
        else { bf16x4 t = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]}; ((bf16x4*)y)[i] = t; }
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
};

__global__ __launch_bounds__(256) void colsum_multi_kernel(ColsumJobs J) {
    __shared__ float part[4][64];
    int j = 0;
    while (j + 1 < J.nj && (int)blockIdx.x >= J.blk_off[j + 1]) ++j;
    const int64_t M = J.M[j], N = J.N[j], ldx = J.ld[j];
    const int64_t r0 = (int64_t)blockIdx.y * CS_ROWS;
    if (r0 >= M) return;
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int64_t n = (int64_t)(blockIdx.x - J.blk_off[j]) * 64 + c;
    int64_t r1 = r0 + CS_ROWS;
    if (r1 > M) r1 = M;
    const void* X = J.x[j];
    const int xd = J.dt[j];
    float s = 0.f;
    if (n < N)
        for (int64_t r = r0 + g; r < r1; r += 4) s += vs_ld(X, xd, r * ldx + n);
    part[g][c] = s;
    __syncthreads();
    if (g == 0 && n < N) atomicAdd(J.out[j] + n, part[0][c] + part[1][c] + part[2][c] + part[3][c]);
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
        J.blk_off[j + 1] = J.blk_off[j] + (int)vs_cdiv(N[j], 64);
        if (M[j] > max_m) max_m = M[j];
    }
    if (zero_base && zero_count > 0) {
        if (vs_zero_async(zero_base, (size_t)zero_count * sizeof(float), (hipStream_t)stream) != hipSuccess)
            return vs_fail(VS_ERR_LAUNCH, "vs_colsum_multi: memset failed");
    }
    dim3 grid((unsigned)J.blk_off[n_jobs], (unsigned)vs_cdiv(max_m, CS_ROWS));
    hipLaunchKernelGGL(colsum_multi_kernel, grid, dim3(256), 0, (hipStream_t)stream, J);
    VS_CHECK_LAUNCH("vs_colsum_multi");
    return VS_OK;
}

extern "C" int vs_cast(const void* src, int sd, void* dst, int dd, int64_t n, void* stream) {
    VS_CHECK_ARG(src && dst && n >= 0, "vs_cast: bad argument");
    VS_CHECK_ARG((sd == VS_F32 || sd == VS_BF16) && (dd == VS_F32 || dd == VS_BF16), "vs_cast: bad dtype");
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
