// vs_data.hip -- input pipeline on the device: Moving-MNIST sequence generation (reference: data/moving_mnist.py:112-253).
//
// The reference builds every training sequence on the host: per digit five draws from the global NumPy stream (digit index, start
// position, speed), a trajectory of `seq_len` positions with elastic bounces off the frame borders computed in Python floats
// (= IEEE double), and a compositing loop that adds the 28x28 digit into a 64x64 frame per time step, clips at 255 and divides by
// 255.  At MI355X step rates (Moving-MNIST B=128: > 100 k frames/s) a 4-worker host generator is the bottleneck, so here only the
// five integers per digit cross PCIe (drawn by the host from the same NumPy stream, in the reference's order) and one launch
// renders the whole batch:
//   * trajectory: thread d < num_digits of every workgroup replays moving_mnist.py:154-237 for its digit in double arithmetic,
//     operation for operation (deterministic mode: no draws inside the bounce loop), and leaves the rounded positions
//     (Python round() = round-half-even = rint) of frame t in LDS;
//   * compositing: the workgroup of (sample b, frame t) writes the 64x64 frame: sum of the digits' pixels at their positions,
//     min(., 255), / 255 (IEEE division: same bits as NumPy's float32 division), as fp32 or a 16-bit type.
// Digits live in HBM as uint8 [n_digits_total, h, w] (MNIST: 47 MB).
#include "vs_common.h"

namespace {

constexpr int MM_MAXD = 8;          // digits per video (reference default 2)

struct Traj { double sx, sy; int dx, dy; };

// moving_mnist.py:257-299: intersection of the line y = a x + b with a vertical / horizontal border
__device__ __forceinline__ bool inter_x(double a, double b, double x_lim, double lo, double hi, double eps, double& cx, double& cy) {
    const double y = a * x_lim + b;
    cx = x_lim; cy = y;
    return (y >= lo - eps) && (y <= hi + eps);
}
__device__ __forceinline__ bool inter_y(double a, double b, double y_lim, double lo, double hi, double eps, double& cx, double& cy) {
    const double x = (y_lim - b) / a;
    cx = x; cy = y_lim;
    return (x >= lo - eps) && (x <= hi + eps);
}

// moving_mnist.py:177-255 (`_process_collision`, deterministic: the speed is mirrored, never redrawn)
__device__ void process_collision(Traj& s, double x_min, double x_max, double y_min, double y_max) {
    const double eps = 1e-8;
    bool left = s.sx < x_min - eps, upper = s.sy < y_min - eps, right = s.sx > x_max + eps, bottom = s.sy > y_max + eps;
    int guard = 0;
    while ((left || right || upper || bottom) && ++guard < 64) {
        double cx = 0.0, cy = 0.0;
        if (s.dx == 0) {
            cx = s.sx; cy = upper ? y_min : y_max;
        } else if (s.dy == 0) {
            cx = left ? x_min : x_max; cy = s.sy;
        } else {
            const double a = (double)s.dy / (double)s.dx;
            const double b = s.sy - a * s.sx;
            double tx, ty;
            if (left) { left = inter_x(a, b, x_min, y_min, y_max, eps, tx, ty); if (left) { cx = tx; cy = ty; } }
            if (right) { right = inter_x(a, b, x_max, y_min, y_max, eps, tx, ty); if (right) { cx = tx; cy = ty; } }
            if (upper) { upper = inter_y(a, b, y_min, x_min, x_max, eps, tx, ty); if (upper) { cx = tx; cy = ty; } }
            if (bottom) { bottom = inter_y(a, b, y_max, x_min, x_max, eps, tx, ty); if (bottom) { cx = tx; cy = ty; } }
        }
        const double p = s.dx != 0 ? (s.sx - cx) / (double)s.dx : (s.sy - cy) / (double)s.dy;
        if (left) s.dx = abs(s.dx);
        if (right) s.dx = -abs(s.dx);
        if (upper) s.dy = abs(s.dy);
        if (bottom) s.dy = -abs(s.dy);
        s.sx = cx + (double)s.dx * p;
        s.sy = cy + (double)s.dy * p;
        left = s.sx < x_min - eps; upper = s.sy < y_min - eps; right = s.sx > x_max + eps; bottom = s.sy > y_max + eps;
    }
}

// grid (T, B): one workgroup per frame.  init[b][d] = (digit index, sx, sy, dx, dy)
__global__ __launch_bounds__(256) void moving_mnist_kernel(const unsigned char* __restrict__ digits, int dh, int dw, const int* __restrict__ init,
                                                           int nd, int T, int F, void* out, int od) {
    __shared__ int pos[MM_MAXD][3];            // digit index, row offset, column offset of this frame
    const int t = blockIdx.x, b = blockIdx.y;
    if ((int)threadIdx.x < nd) {
        const int* q = init + ((int64_t)b * nd + threadIdx.x) * 5;
        Traj s{(double)q[1], (double)q[2], q[3], q[4]};
        const double x_max = (double)(F - dh), y_max = (double)(F - dw);
        int px = 0, py = 0;
        for (int tt = 0; tt <= t; ++tt) {      // moving_mnist.py:166-175: bounce, record the rounded position, then move
            process_collision(s, 0.0, x_max, 0.0, y_max);
            px = (int)rint(s.sx); py = (int)rint(s.sy);
            s.sy += (double)s.dy;
            s.sx += (double)s.dx;
        }
        pos[threadIdx.x][0] = q[0]; pos[threadIdx.x][1] = px; pos[threadIdx.x][2] = py;
    }
    __syncthreads();
    const int64_t base = ((int64_t)b * T + t) * F * F;
    for (int i = threadIdx.x; i < F * F; i += 256) {
        const int r = i / F, c = i - r * F;
        float v = 0.f;
        for (int d = 0; d < nd; ++d) {
            const int rr = r - pos[d][1], cc = c - pos[d][2];
            if (rr >= 0 && rr < dh && cc >= 0 && cc < dw) v += (float)digits[((int64_t)pos[d][0] * dh + rr) * dw + cc];
        }
        v = v > 255.f ? 255.f : v;
        vs_st(out, od, base + i, v / 255.f);
    }
}

}  // namespace

extern "C" int vs_moving_mnist_batch(const uint8_t* digits, int64_t n_digits_total, int digit_h, int digit_w, const int32_t* init, int batch,
                                     int num_digits, int seq_len, int frame_size, void* out, int out_dtype, void* stream) {
    VS_CHECK_ARG(digits && init && out && n_digits_total > 0 && digit_h > 0 && digit_w > 0 && batch > 0 && seq_len > 0, "vs_moving_mnist_batch: bad argument");
    VS_CHECK_ARG(num_digits >= 1 && num_digits <= MM_MAXD, "vs_moving_mnist_batch: 1..%d digits per video", MM_MAXD);
    VS_CHECK_ARG(frame_size >= digit_h && frame_size >= digit_w, "vs_moving_mnist_batch: the digit does not fit the frame");
    VS_CHECK_ARG(vs_dtype_ok(out_dtype), "vs_moving_mnist_batch: bad out_dtype");
    hipLaunchKernelGGL(moving_mnist_kernel, dim3((unsigned)seq_len, (unsigned)batch), dim3(256), 0, (hipStream_t)stream, digits, digit_h, digit_w, init,
                       num_digits, seq_len, frame_size, out, out_dtype);
    VS_CHECK_LAUNCH("vs_moving_mnist_batch");
    return VS_OK;
}
