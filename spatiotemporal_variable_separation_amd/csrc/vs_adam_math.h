// vs_adam_math.h -- the Adam update of one element, shared by the optimizer kernel (vs_optim.hip) and the weight-gradient GEMM
// whose epilogue applies it directly (vs_gemm_adam): both produce bitwise the same parameters from the same gradient.
// Arithmetic follows torch.optim.Adam's single-tensor path (betas / lr arrive as Python doubles; 1 - beta and beta^t in double).
#pragma once
#include <hip/hip_runtime.h>

namespace {

struct AdamCoef {
    float w1, w2, beta2, bc2_sqrt, step_size, eps;
};

// t = number of steps this parameter will have taken after the update
__device__ __forceinline__ AdamCoef vs_adam_coef(double lr_d, double beta1_d, double beta2_d, float eps, double t) {
#pragma clang fp contract(off)
    AdamCoef c;
    c.w1 = (float)(1.0 - beta1_d);
    c.w2 = (float)(1.0 - beta2_d);
    c.beta2 = (float)beta2_d;
    const float bc1 = (float)(1.0 - pow(beta1_d, t));
    c.bc2_sqrt = (float)sqrt(1.0 - pow(beta2_d, t));
    c.step_size = (float)lr_d / bc1;
    c.eps = eps;
    return c;
}

// (no FMA contraction: the compiler would otherwise fuse differently in the two kernels that inline this and they would differ in
// the last bit; torch's single-tensor path also rounds every product)
__device__ __forceinline__ void vs_adam_elem(const AdamCoef& c, float g, float& p, float& m, float& v) {
#pragma clang fp contract(off)
    m = m + c.w1 * (g - m);                                  // exp_avg.lerp_(grad, 1 - beta1)
    v = v * c.beta2 + (c.w2 * g) * g;                        // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
    p = p - c.step_size * (m / denom);                       // param.addcdiv_(exp_avg, denom, -step_size)
}

}  // namespace
