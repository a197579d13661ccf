// vs_metrics.hip -- the frame metrics of the reference's evaluation scripts on the device: per-plane mean squared error (PSNR is
// 10 log10(1 / mse), test/mnist/test.py:138-141) and per-plane mean SSIM (utils/ssim.py:81-111 through test/utils.py:19-24
// `_ssim_wrapper`: 11 x 11 Gaussian window, sigma 1.5, "valid" windows, k1 = 0.01, k2 = 0.03).
//
// The reference runs five depthwise 11 x 11 conv2d launches per call (mu_x, mu_y, E[x^2], E[y^2], E[xy]) plus ~15 elementwise
// ones.  Here ONE workgroup owns a plane pair: both planes go to LDS once, the Gaussian is applied separably (the 2-D window is the
// outer product of the normalised 1-D window: softmax of a sum = product of softmaxes) -- a horizontal pass of the five maps into
// LDS, then the vertical pass, the SSIM formula per window position and the plane mean in registers.  121 -> 22 multiply-adds per
// window and map, every input byte read from HBM once.
#include "vs_common.h"

namespace {

constexpr int SS_WIN = 11;

struct SsimWindow { float g[SS_WIN]; };

// planes of up to 64 x 64 (32 KiB of inputs + 5 x H x (W - 10) floats of row-filtered maps <= 69 KiB)
__global__ __launch_bounds__(256) void frame_metrics_kernel(const float* pred, const float* target, int H, int W, SsimWindow win, float c1, float c2,
                                                           float* mse, float* ssim) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int64_t plane = blockIdx.x;
    const int hw = H * W, OW = W - SS_WIN + 1, OH = H - SS_WIN + 1;
    float* sx = smem;                      // [H][W]
    float* sy = smem + hw;                 // [H][W]
    float* hm = smem + 2 * hw;             // [5][H][OW]
    const float* px = pred + plane * hw;
    const float* py = target + plane * hw;
    float se = 0.f;
    for (int i = threadIdx.x; i < hw; i += 256) {
        const float a = px[i], b = py[i];
        sx[i] = a; sy[i] = b;
        const float d = a - b;
        se += d * d;
    }
    __syncthreads();
    // horizontal pass
    const int nh = H * OW;
    for (int i = threadIdx.x; i < nh; i += 256) {
        const int r = i / OW, c = i - r * OW;
        float m1 = 0.f, m2 = 0.f, q1 = 0.f, q2 = 0.f, q12 = 0.f;
#pragma unroll
        for (int j = 0; j < SS_WIN; ++j) {
            const float a = sx[r * W + c + j], b = sy[r * W + c + j], g = win.g[j];
            m1 += g * a; m2 += g * b; q1 += g * (a * a); q2 += g * (b * b); q12 += g * (a * b);
        }
        hm[i] = m1; hm[nh + i] = m2; hm[2 * nh + i] = q1; hm[3 * nh + i] = q2; hm[4 * nh + i] = q12;
    }
    __syncthreads();
    // vertical pass + SSIM per window position
    float acc = 0.f;
    const int no = OH * OW;
    for (int i = threadIdx.x; i < no; i += 256) {
        const int r = i / OW, c = i - r * OW;
        float m1 = 0.f, m2 = 0.f, q1 = 0.f, q2 = 0.f, q12 = 0.f;
#pragma unroll
        for (int j = 0; j < SS_WIN; ++j) {
            const int k = (r + j) * OW + c;
            const float g = win.g[j];
            m1 += g * hm[k]; m2 += g * hm[nh + k]; q1 += g * hm[2 * nh + k]; q2 += g * hm[3 * nh + k]; q12 += g * hm[4 * nh + k];
        }
        const float mu1_sq = m1 * m1, mu2_sq = m2 * m2, mu12 = m1 * m2;
        const float v1 = 2.f * (q12 - mu12) + c2, v2 = (q1 - mu1_sq) + (q2 - mu2_sq) + c2;
        acc += ((2.f * mu12 + c1) * v1) / ((mu1_sq + mu2_sq + c1) * v2);
    }
    // block sums (4 waves)
    __shared__ float red[2][4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { se += __shfl_down(se, o, 64); acc += __shfl_down(acc, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = se; red[1][threadIdx.x >> 6] = acc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (mse) mse[plane] = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)hw;
        if (ssim) ssim[plane] = no > 0 ? (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (float)no : 0.f;
    }
}

}  // namespace

extern "C" int vs_frame_metrics(const float* pred, const float* target, int64_t planes, int H, int W, float max_val, float k1, float k2, float sigma,
                                float* mse, float* ssim, void* stream) {
    VS_CHECK_ARG(pred && target && planes > 0 && planes < (1ll << 31) && H >= SS_WIN && W >= SS_WIN && (mse || ssim), "vs_frame_metrics: bad argument");
    const size_t lds = ((size_t)2 * H * W + (size_t)5 * H * (W - SS_WIN + 1)) * sizeof(float);
    if (lds > 150 * 1024) return vs_fail(VS_ERR_UNSUPPORTED, "vs_frame_metrics: planes of %d x %d do not fit the LDS (<= 64 x 64 .. 80 x 80)", H, W);
    // the reference's window: softmax over the 2-D grid of -(x^2 + y^2) / (2 sigma^2) == outer product of the normalised 1-D windows
    SsimWindow win;
    double sum = 0.0, e[SS_WIN];
    for (int j = 0; j < SS_WIN; ++j) {
        const double x = (double)j - (SS_WIN - 1) / 2.0;
        e[j] = exp(-x * x / (2.0 * (double)sigma * (double)sigma));
        sum += e[j];
    }
    for (int j = 0; j < SS_WIN; ++j) win.g[j] = (float)(e[j] / sum);
    const float c1 = (k1 * max_val) * (k1 * max_val), c2 = (k2 * max_val) * (k2 * max_val);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)frame_metrics_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
            return vs_fail(VS_ERR_LAUNCH, "vs_frame_metrics: cannot raise the dynamic LDS limit");
        attr_set = true;
    }
    hipLaunchKernelGGL(frame_metrics_kernel, dim3((unsigned)planes), dim3(256), lds, (hipStream_t)stream, pred, target, H, W, win, c1, c2, mse, ssim);
    VS_CHECK_LAUNCH("vs_frame_metrics");
    return VS_OK;
}
