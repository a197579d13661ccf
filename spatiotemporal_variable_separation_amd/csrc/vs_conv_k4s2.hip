// vs_conv_k4s2.hip -- the 4x4 stride-2 pad-1 convolution family of the DCGAN encoder / decoder WITHOUT a column matrix (gfx950 only).
//
// Reference layers: Conv2d(c, 2c, 4, 2, 1) of DCGAN64Encoder (networks/conv.py:119-122) and ConvTranspose2d(2c, c, 4, 2, 1) of DCGAN64Decoder
// (conv.py:260-263).  The transposed convolution's forward and the convolution's input gradient are scatter-type and run on the tap
// kernel (vs_conv_tap.hip).  The other four operations are gather-type,
//
//     out[m][oy][ox] = sum_{c, ky, kx} W[m][c][ky][kx] * in[c][2 oy + ky - 1][2 ox + kx - 1]                       (Conv2d forward, ConvT dgrad)
//     dW[m][c][ky][kx] = sum_{maps, oy, ox} small[m][oy][ox] * big[c][2 oy + ky - 1][2 ox + kx - 1]               (both weight gradients)
//
// and used to go through a [c x 16 taps][pixels] column matrix in HBM (written once, read by a GEMM: 4x the tensor, PMC: 147 MB written per
// gather launch and 480-560 MB fetched per GEMM launch at Moving-MNIST size).
//
// A stride-2 tap never mixes the parities of the input grid: tap ky reads input rows of parity (ky + 1) & 1, tap kx columns of parity
// (kx + 1) & 1.  Splitting the input into its four parity planes P[py][px][c][y][x] = in[c][2 y + py][2 x + px] ("space to depth") turns the
// 4x4 stride-2 window into four 2x2 stride-1 windows, one per plane, on maps of the OUTPUT's size:
//
//     ky = 0 -> plane row y - 1 of the odd plane,  ky = 1 -> row y of the even plane,  ky = 2 -> row y of the odd plane,  ky = 3 -> row y + 1 (even)
//
// i.e. the operation IS a 3x3 stride-1 pad-1 convolution over 4 c channels (plane-major) whose 3x3 weight holds, per plane, the 2x2 taps that
// plane sees (the other five are zero) -- including the zero padding: row -1 of the odd plane and row H/2 of the even plane are the
// out-of-range rows of the original.  So the row-band kernels of vs_conv_img.hip do the work (LDS-DMA staged row bands, x shift on the result,
// MFMA fragments streamed from a pre-pack, no column matrix); this file supplies the three small kernels around them:
//   * vs_space_to_depth2: in [B][C][H][W] (16-bit) -> planes [B][4 C][H/2][W/2] (16-byte loads, 16-byte stores; one read + one write of the
//     tensor instead of a 4x column matrix; the planes of a gradient map serve BOTH the input gradient and the weight gradient of a
//     transposed convolution);
//   * vs_conv_k4s2_pack_weight: fp32 [M][K][4][4] (Conv2d weight, or a ConvTranspose2d weight read as [in = M][out = K] for its input
//     gradient) -> the row-band kernels' MFMA-fragment pre-pack over 4 K plane channels;
//   * vs_conv_k4s2_wgrad_finish: the row-band weight gradient's fp32 slabs [slab][9 taps][M][4 K] -> dW [M][K][4][4] (+ a pending gradient),
//     reading only the 16 of 36 (plane, tap) positions that are taps of the 4x4 window.
#include "vs_common.h"

namespace {

typedef u32x4 k4_u32x4;

// 4x4 tap index t in 0..3 along one axis -> (parity plane, 3x3 tap of that plane): t = 0 -> (1, 0), 1 -> (0, 1), 2 -> (1, 1), 3 -> (0, 2)
__host__ __device__ __forceinline__ int k4_plane(int t) { return (t + 1) & 1; }
__host__ __device__ __forceinline__ int k4_tap3(int t) { return (t + 1) >> 1; }
// inverse: (plane parity p, 3x3 tap t3) -> 4x4 tap or -1:  p = 1: t3 0 -> 0, 1 -> 2;  p = 0: t3 1 -> 1, 2 -> 3
__host__ __device__ __forceinline__ int k4_tap4(int p, int t3) {
    const int t = 2 * t3 + p - 1;
    return (t >= 0 && t <= 3) ? t : -1;
}

// one thread: 16 consecutive input pixels of one row (two 16-byte loads) -> 8 even + 8 odd pixels (one 16-byte store into each of two planes)
__global__ __launch_bounds__(256) void space_to_depth2_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, int C, int H, int W,
                                                              int64_t units) {
    const int wu = W >> 4, H2 = H >> 1, W2 = W >> 1;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < units; u += (int64_t)gridDim.x * 256) {
        const int j = (int)(u % wu);
        int64_t t = u / wu;
        const int r = (int)(t % H);
        t /= H;
        const int c = (int)(t % C);
        const int64_t b = t / C;
        const unsigned short* src = x + ((b * C + c) * H + r) * (int64_t)W + 16 * j;
        const k4_u32x4 a = *reinterpret_cast<const k4_u32x4*>(src), bq = *reinterpret_cast<const k4_u32x4*>(src + 8);
        // dword d of a = pixels (2 d, 2 d + 1): low half even, high half odd
        k4_u32x4 ev, od;
        ev[0] = (a[0] & 0xffffu) | (a[1] << 16);
        ev[1] = (a[2] & 0xffffu) | (a[3] << 16);
        ev[2] = (bq[0] & 0xffffu) | (bq[1] << 16);
        ev[3] = (bq[2] & 0xffffu) | (bq[3] << 16);
        od[0] = (a[0] >> 16) | (a[1] & 0xffff0000u);
        od[1] = (a[2] >> 16) | (a[3] & 0xffff0000u);
        od[2] = (bq[0] >> 16) | (bq[1] & 0xffff0000u);
        od[3] = (bq[2] >> 16) | (bq[3] & 0xffff0000u);
        const int py = r & 1, yy = r >> 1;
        unsigned short* d0 = y + (((b * 4 + (py * 2 + 0)) * C + c) * H2 + yy) * (int64_t)W2 + 8 * j;
        unsigned short* d1 = y + (((b * 4 + (py * 2 + 1)) * C + c) * H2 + yy) * (int64_t)W2 + 8 * j;
        *reinterpret_cast<k4_u32x4*>(d0) = ev;
        *reinterpret_cast<k4_u32x4*>(d1) = od;
    }
}

// rows of 8 pixels (8 x 8 maps -> 4 x 4 planes): one thread = one input row (16 bytes) -> 4 even + 4 odd pixels, an 8-byte store into each of
// the row parity's two planes
__global__ __launch_bounds__(256) void space_to_depth2_w8_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, int C, int H,
                                                                 int64_t units) {
    const int H2 = H >> 1;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < units; u += (int64_t)gridDim.x * 256) {
        const int r = (int)(u % H);
        int64_t t = u / H;
        const int c = (int)(t % C);
        const int64_t b = t / C;
        const k4_u32x4 a = *reinterpret_cast<const k4_u32x4*>(x + u * 8);
        const u32x2 ev = {(a[0] & 0xffffu) | (a[1] << 16), (a[2] & 0xffffu) | (a[3] << 16)};
        const u32x2 od = {(a[0] >> 16) | (a[1] & 0xffff0000u), (a[2] >> 16) | (a[3] & 0xffff0000u)};
        const int py = r & 1, yy = r >> 1;
        *reinterpret_cast<u32x2*>(y + (((b * 4 + (py * 2 + 0)) * C + c) * H2 + yy) * 4) = ev;
        *reinterpret_cast<u32x2*>(y + (((b * 4 + (py * 2 + 1)) * C + c) * H2 + yy) * 4) = od;
    }
}

// Wp[mt][chunk][ky3][kx3][lane][8] (the layout of conv3_img16_pack_kernel, flip = 0) over 4 K plane channels c' = plane K + c
template <int CT>
__global__ __launch_bounds__(256) void k4s2_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst, int M, int K, int64_t total) {
    const int chunks = (4 * K) >> 4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int jj = (int)(e & 7), lane = (int)((e >> 3) & 63);
        int64_t t = e >> 9;
        const int kx3 = (int)(t % 3);
        t /= 3;
        const int ky3 = (int)(t % 3);
        t /= 3;
        const int chunk = (int)(t % chunks), mt = (int)(t / chunks);
        const int m = mt * 32 + (lane & 31), cp = chunk * 16 + 8 * (lane >> 5) + jj;
        const int plane = cp / K, c = cp - plane * K;
        const int ky = k4_tap4(plane >> 1, ky3), kx = k4_tap4(plane & 1, kx3);
        float v = 0.f;
        if (m < M && ky >= 0 && kx >= 0) v = w[(((int64_t)m * K + c) * 4 + ky) * 4 + kx];
        dst[e] = vs_f2h(v, CT);
    }
}

// the same for K a multiple of 64 (vs_conv_k4s2_skip_form): only the 2 x 2 taps a plane sees, Wp[mt][chunk][kyi][kxi][lane][8] with
// ky3 = ky0(plane) + kyi, kx3 = kx0(plane) + kxi (ky0 = 0 for the odd-row planes, 1 for the even ones; likewise kx0) -- the order the K4 form of
// conv3_band_kernel streams
template <int CT>
__global__ __launch_bounds__(256) void k4s2_pack_skip_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst, int M, int K, int64_t total) {
    const int chunks = (4 * K) >> 4;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int jj = (int)(e & 7), lane = (int)((e >> 3) & 63);
        int64_t t = e >> 9;
        const int kxi = (int)(t & 1);
        t >>= 1;
        const int kyi = (int)(t & 1);
        t >>= 1;
        const int chunk = (int)(t % chunks), mt = (int)(t / chunks);
        const int m = mt * 32 + (lane & 31), cp = chunk * 16 + 8 * (lane >> 5) + jj;
        const int plane = cp / K, c = cp - plane * K;
        const int py = plane >> 1, px = plane & 1;
        const int ky = k4_tap4(py, (py ? 0 : 1) + kyi), kx = k4_tap4(px, (px ? 0 : 1) + kxi);
        float v = 0.f;
        if (m < M) v = w[(((int64_t)m * K + c) * 4 + ky) * 4 + kx];
        dst[e] = vs_f2h(v, CT);
    }
}

// dW[m][c][ky][kx] = sum over the n slabs [tap3][m][c'] (+ the pending gradient); one thread per (m, c): 16 coalesced reads per slab along c
__global__ __launch_bounds__(256) void k4s2_wgrad_finish_kernel(const float* __restrict__ src, int n, int M, int K, const float* __restrict__ addend,
                                                                float* __restrict__ out) {
    const int64_t mk = (int64_t)M * K, mc3 = (int64_t)M * 4 * K, slab = mc3 * 9;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < mk; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / K;
        const int c = (int)(i - m * K);
        float s[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) s[t] = addend ? addend[i * 16 + t] : 0.f;
        for (int k = 0; k < n; ++k)                                                // fixed order: reproducible
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int ky = t >> 2, kx = t & 3;
                const int plane = k4_plane(ky) * 2 + k4_plane(kx), tap3 = k4_tap3(ky) * 3 + k4_tap3(kx);
                s[t] += src[(int64_t)k * slab + (int64_t)tap3 * mc3 + m * (4 * (int64_t)K) + (int64_t)plane * K + c];
            }
#pragma unroll
        for (int t = 0; t < 16; ++t) out[i * 16 + t] = s[t];
    }
}

}  // namespace

extern "C" int vs_space_to_depth2_supported(int compute, int B, int C, int H, int W) {
    return vs_is16(compute) && B > 0 && C > 0 && H >= 2 && H % 2 == 0 && ((W >= 16 && W % 16 == 0) || W == 8) && (int64_t)B * C * H * W < ((int64_t)1 << 40);
}

extern "C" int vs_space_to_depth2(int compute, const void* x, void* y, int B, int C, int H, int W, void* stream) {
    VS_CHECK_ARG(x && y, "vs_space_to_depth2: bad argument");
    VS_CHECK_ARG(vs_space_to_depth2_supported(compute, B, C, H, W), "vs_space_to_depth2: 16-bit tensors with even H and W = 8 or a multiple of 16 only");
    VS_CHECK_ARG(((uintptr_t)x | (uintptr_t)y) % 16 == 0, "vs_space_to_depth2: operands must be 16-byte aligned");
    if (W == 8) {
        const int64_t rows = (int64_t)B * C * H;
        int64_t blocks8 = vs_cdiv(rows, 256);
        if (blocks8 > 16384) blocks8 = 16384;
        hipLaunchKernelGGL(space_to_depth2_w8_kernel, dim3((unsigned)blocks8), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, (unsigned short*)y, C, H, rows);
        VS_CHECK_LAUNCH("vs_space_to_depth2");
        return VS_OK;
    }
    const int64_t units = (int64_t)B * C * H * (W >> 4);
    int64_t blocks = vs_cdiv(units, 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(space_to_depth2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, (unsigned short*)y, C, H, W, units);
    VS_CHECK_LAUNCH("vs_space_to_depth2");
    return VS_OK;
}

extern "C" int vs_conv_k4s2_skip_form(int K);          // vs_conv_img.hip

extern "C" size_t vs_conv_k4s2_packed_elems(int K, int M) { return (size_t)vs_cdiv(M, 32) * 32 * (size_t)(4 * K) * (vs_conv_k4s2_skip_form(K) ? 4 : 9); }

extern "C" int vs_conv_k4s2_pack_weight(int compute, const float* w, int K, int M, void* dst, void* stream) {
    VS_CHECK_ARG(w && dst && vs_is16(compute) && K > 0 && M > 0 && (4 * K) % 16 == 0, "vs_conv_k4s2_pack_weight: bad argument (16-bit, K a multiple of 4)");
    const int64_t total = (int64_t)vs_conv_k4s2_packed_elems(K, M);
    int64_t blocks = vs_cdiv(total, 256);
    if (blocks > 4096) blocks = 4096;
    if (vs_conv_k4s2_skip_form(K)) {
        if (compute == VS_BF16)
            hipLaunchKernelGGL(k4s2_pack_skip_kernel<VS_BF16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)dst, M, K, total);
        else
            hipLaunchKernelGGL(k4s2_pack_skip_kernel<VS_F16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)dst, M, K, total);
    } else if (compute == VS_BF16)
        hipLaunchKernelGGL(k4s2_pack_kernel<VS_BF16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)dst, M, K, total);
    else
        hipLaunchKernelGGL(k4s2_pack_kernel<VS_F16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)dst, M, K, total);
    VS_CHECK_LAUNCH("vs_conv_k4s2_pack_weight");
    return VS_OK;
}

extern "C" int vs_conv_k4s2_wgrad_finish(const float* slabs, int nslabs, const float* addend, float* out, int M, int K, void* stream) {
    VS_CHECK_ARG(slabs && out && nslabs >= 1 && M > 0 && K > 0, "vs_conv_k4s2_wgrad_finish: bad argument");
    int64_t blocks = vs_cdiv((int64_t)M * K, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k4s2_wgrad_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, slabs, nslabs, M, K, addend, out);
    VS_CHECK_LAUNCH("vs_conv_k4s2_wgrad_finish");
    return VS_OK;
}
