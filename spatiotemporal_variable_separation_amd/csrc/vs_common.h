// Shared device/host helpers for libvarsep_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/varsep_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;

#define VS_WAVE 64

extern thread_local char vs_err_buf[256];
int vs_fail(int code, const char* fmt, ...);

// The process-wide exchange guard word (vs_exchange_guard_set, vs_eltwise.hip): when registered, every kernel with a bounded in-launch
// exchange (the MLP integrator, the one-launch ConvResBlock layer) raises THIS device word on a time-out instead of the word inside its own
// workspace, and every optimizer launch (vs_adam_multi*, vs_gemm_adam, the step counter) reads it first and leaves parameters, moments and
// the step count untouched while it is non-zero: an update is never computed from the results of a timed-out exchange.
extern unsigned* vs_g_exchange_guard;
// optional companion (vs_exchange_skip_counter_set): bumped by the step-count kernel every time the guard made it skip a step
extern unsigned* vs_g_exchange_skips;

#define VS_CHECK_ARG(cond, ...)                        \
    do {                                               \
        if (!(cond)) return vs_fail(VS_ERR_ARG, __VA_ARGS__); \
    } while (0)

#define VS_CHECK_LAUNCH(what)                                                            \
    do {                                                                                 \
        hipError_t e_ = hipGetLastError();                                               \
        if (e_ != hipSuccess) return vs_fail(VS_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e_)); \
    } while (0)

// ---- zero fill as a kernel ----------------------------------------------------------------------------------------
// hipMemsetAsync recorded into a hipGraph (stream capture) does not reliably re-execute in order on ROCm 7.0: replays of a
// single-stream capture left accumulator buffers un-zeroed.  A fill kernel is an ordinary kernel node.  `bytes` must be a
// multiple of 4 and `ptr` 4-byte aligned (every caller zeroes fp32 / 8-byte granule areas).
static __global__ void vs_zero_kernel(uint32_t* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}

static inline hipError_t vs_zero_async(void* ptr, size_t bytes, hipStream_t stream) {
    const size_t n = bytes / 4;
    if (n == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(vs_zero_kernel, dim3(blocks), dim3(256), 0, stream, (uint32_t*)ptr, n);
    return hipGetLastError();
}

// ---- 16-bit storage types: VS_BF16 and VS_F16 share every data-movement path; only the conversions differ -----------------
__host__ __device__ __forceinline__ constexpr bool vs_is16(int dtype) { return dtype == VS_BF16 || dtype == VS_F16; }
__host__ __device__ __forceinline__ constexpr bool vs_dtype_ok(int dtype) { return dtype == VS_F32 || dtype == VS_BF16 || dtype == VS_F16; }
__host__ __device__ __forceinline__ constexpr int vs_esize(int dtype) { return dtype == VS_F32 ? 4 : 2; }

// bits of a 16-bit element -> float / float -> bits (round to nearest even; a NaN stays a NaN: plain casts, see the guide)
__device__ __forceinline__ float vs_h2f(unsigned short b, int dtype) {
    if (dtype == VS_BF16) return __uint_as_float((unsigned)b << 16);
    _Float16 h;
    __builtin_memcpy(&h, &b, 2);
    return (float)h;
}
__device__ __forceinline__ unsigned short vs_f2h(float v, int dtype) {
    unsigned short b;
    if (dtype == VS_BF16) { const __bf16 h = (__bf16)v; __builtin_memcpy(&b, &h, 2); }
    else { const _Float16 h = (_Float16)v; __builtin_memcpy(&b, &h, 2); }
    return b;
}

// ---- scalar load/store with dtype dispatch (dtype is wave-uniform) -------------------------------
__device__ __forceinline__ float vs_ld(const void* p, int dtype, int64_t i) {
    return dtype == VS_F32 ? ((const float*)p)[i] : vs_h2f(((const unsigned short*)p)[i], dtype);
}
__device__ __forceinline__ void vs_st(void* p, int dtype, int64_t i, float v) {
    if (dtype == VS_F32) ((float*)p)[i] = v;
    else ((unsigned short*)p)[i] = vs_f2h(v, dtype);
}

// ---- activations (networks/utils.py:50-72) ---------------------------------------------------------
__device__ __forceinline__ float vs_act(float v, int act) {
    switch (act) {
        case VS_ACT_RELU: return v > 0.f ? v : 0.f;
        case VS_ACT_LEAKY: return v > 0.f ? v : 0.2f * v;
        case VS_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case VS_ACT_TANH: return tanhf(v);
        case VS_ACT_ELU: return v > 0.f ? v : (expf(v) - 1.f);
        default: return v;
    }
}
// derivative of the activation expressed through its OUTPUT y (what the forward pass kept)
__device__ __forceinline__ float vs_act_grad_from_out(float y, int act) {
    switch (act) {
        case VS_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case VS_ACT_LEAKY: return y > 0.f ? 1.f : 0.2f;
        case VS_ACT_SIGMOID: return y * (1.f - y);
        case VS_ACT_TANH: return 1.f - y * y;
        case VS_ACT_ELU: return y > 0.f ? 1.f : (y + 1.f);
        default: return 1.f;
    }
}

static inline int64_t vs_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
