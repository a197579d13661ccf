// vs_conv_tap.hip -- im2col-free ConvTranspose2d k4 s2 p1 forward (the DCGAN decoder layers, conv.py:233-264: 79 % of the
// Moving-MNIST step's FLOPs) with the input tile staged in LDS and the BatchNorm statistics taken in the epilogue.
//
// Formulation ("tap GEMM + col2im in the epilogue").  A transposed convolution is, per kernel tap t = (ky, kx), a 1x1 convolution
//     G_t[m][y][x] = sum_c W[c][m][ky][kx] * X[c][y][x]                                  (no shifts, no padding, no gather)
// followed by a scatter-add  Y[m][2y - 1 + ky][2x - 1 + kx] += G_t[m][y][x].  The 16 maps G_t of a block of 16 output channels
// and a tile of 256 input pixels (whole images: 16 samples of 4x4, 4 of 8x8, one of 16x16) are ONE 256 x 256 GEMM tile over
// K = Cin: rows (t, m_local), columns pixels.  So the main loop is the 256x256 LDS-DMA ring kernel of vs_gemm_big.h verbatim --
//   A = packed weights [Cout/16][16 taps x 16 channels][Cin]      (R layout, dense rows)
//   B = X itself, NCHW: for one channel the pixels of a tile are contiguous  (S layout [32 k][128 px], read with
//       ds_read_b64_tr_b16: no operand is ever shifted, so the 8-byte alignment rule of the transposing read always holds)
// and the input is read from HBM exactly once per 16 output channels (the previous form gathered it into four per-phase column
// matrices: 4 x (2 B written + read) x taps per input element, then scattered the result with stride-2 two-byte stores).
// Epilogue: the accumulators go to LDS as fp32 (128 KiB ring, two passes of 8 channels), every output pixel sums its <= 4
// contributions G_t[y + dy][x + dx] (zero padding = a skipped term: whole images are in the tile, no halo), adds the bias, rounds
// ONCE to the storage type, is written as 16-byte runs of 8 consecutive output pixels, and its value / square enter the per-
// (call group, channel) sums the following BatchNorm needs (fp64 atomics, one pair per wave and channel) -- the separate
// statistics pass over the output disappears.
#include "vs_gemm_big.h"

namespace {

constexpr int TAP_MB = 16;          // output channels per tile (x 16 taps = 256 GEMM rows)

// B operand: pixels n0 .. n0+255 of X [B][Cin][HW] for channels k0 .. k0+31, as two S half-tiles [32 k][128 px]
struct TapPixels {
    const unsigned short* src[2];
    int kofs;
    bool ok[2];
    int64_t step;
    __device__ __forceinline__ void prepare(const unsigned short* X, int Cin, int HW, int64_t npix, int64_t n0) {
        const int u = (int)threadIdx.x, k = u >> 4, piece = (u & 15) ^ ((k & 3) << 2);
        kofs = k;
        step = (int64_t)BIG_BK * HW;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t n = n0 + 128 * h + piece * 8;           // HW % 8 == 0: the 8 pixels of a piece share their sample
            ok[h] = n < npix;
            const int64_t b = n / HW, pix = n - b * HW;
            src[h] = X + (b * Cin + k) * HW + pix;
        }
    }
    __device__ __forceinline__ void stage(int h, char* lds, int64_t k0, int64_t K, bool live) {
        const int wave = threadIdx.x >> 6;
        const void* g = (live && ok[h] && k0 + kofs < K) ? (const void*)src[h] : (const void*)vs_glds_zero;
        __builtin_amdgcn_global_load_lds((glds_glb_ptr*)g, (glds_lds_ptr*)(lds + wave * 1024), 16, 0, 0);
        src[h] += step;
    }
};

// The shared main loop: acc (256 rows x 256 pixels, 128 x 64 per wave) = At_tile [256][Cin] * X[:, pixels n0 .. n0+255].
// The 4-deep LDS-DMA ring of gemm_big_kernel (tiles t+1 .. t+3 in flight, counted vmcnt, one barrier per tile) in its plain form:
// both fragment sets of a tile are read, then multiplied.  (The cross-tile fragment pipeline of the GEMM needs a third fragment
// set; with the epilogues' state that is past 256 VGPRs and the compiler spilled accumulators INSIDE the loop.)  Ends with the
// ring drained and every wave past its last LDS read: the ring memory is free for the epilogue.
template <int CT>
__device__ __forceinline__ void tap_k_loop(f32x16 (&acc)[4][2], const unsigned short* At_tile, const unsigned short* X, int Cin, int HW, int64_t npix,
                                           int64_t n0, char* smem) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wr = wave >> 2, wc = wave & 3;
    const int64_t K = Cin, kt_end = (K + BIG_BK - 1) / BIG_BK;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    BigOperand<LR> ga;
    TapPixels gb;
    ga.prepare(At_tile, Cin, 256, 0, 0);
    gb.prepare(X, Cin, HW, npix, n0);
    auto stage_tile = [&](int slot, int64_t kt) {
        char* base = smem + slot * BIG_TILE_BYTES;
        const bool live = kt < kt_end;
        ga.stage_checked(0, base + wave * 1024, kt * BIG_BK, K, live);
        ga.stage_checked(1, base + 8192 + wave * 1024, kt * BIG_BK, K, live);
        ga.advance();
        gb.stage(0, base + 16384, kt * BIG_BK, K, live);
        gb.stage(1, base + 24576, kt * BIG_BK, K, live);
    };
    const int bcol = (wc & 1) * 64;
    stage_tile(0, 0);
    stage_tile(1, 1);
    stage_tile(2, 2);
    int slot = 0;
    for (int64_t kt = 0; kt < kt_end; ++kt) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stage_tile((slot + 3) & 3, kt + 3);
        const unsigned short* pa = reinterpret_cast<const unsigned short*>(smem + slot * BIG_TILE_BYTES + wr * 8192);
        const unsigned short* pb = reinterpret_cast<const unsigned short*>(smem + slot * BIG_TILE_BYTES + 16384 + (wc >> 1) * 8192);
        u32x4 a0[4], b0[2], a1[4], b1[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) a0[i] = big_frag<LR>(pa, 32 * i, 0, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) b0[j] = big_frag<LS>(pb, bcol + 32 * j, 0, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) a1[i] = big_frag<LR>(pa, 32 * i, 16, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) b1[j] = big_frag<LS>(pb, bcol + 32 * j, 16, lane);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mfma16_32<CT>(a0[i], b0[j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mfma16_32<CT>(a1[i], b1[j], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
        slot = (slot + 1) & 3;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// wave-level sums of (value, square) of the stored outputs -> fp64 atomics on the (group, channel) slot
__device__ __forceinline__ void tap_bn_accumulate(double* sums, float s1, float s2, int lane, bool valid, int64_t slot_index) {
    // <= 1024 stored values per lane group: fp32 partial sums inside the wave (relative error ~1e-6), fp64 across workgroups
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_down(s1, off, 64); s2 += __shfl_down(s2, off, 64); }
    if (lane == 0 && valid) {
        atomicAdd(sums + slot_index * 2, (double)s1);
        atomicAdd(sums + slot_index * 2 + 1, (double)s2);
    }
}

template <int CT>
__global__ __launch_bounds__(512) void convt_k4s2_tap_kernel(const unsigned short* X, const unsigned short* At, const float* bias, unsigned short* Y,
                                                             double* sums, int B, int Cin, int H, int W, int Cout, int Bg, int tiles_m, int diag, int y_f32) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // ring: 4 x [A0 | A1 | B0 | B1] x 8 KiB; then the fp32 staging area
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int HW = H * W;
    const int64_t npix = (int64_t)B * HW;
    const unsigned tile = big_tile_of(blockIdx.x, gridDim.x);
    const int tm = (int)(tile % (unsigned)tiles_m);
    const int64_t n0 = (int64_t)(tile / (unsigned)tiles_m) * 256;
    f32x16 acc[4][2];
    tap_k_loop<CT>(acc, At + (int64_t)tm * 256 * (diag & 2 ? 32 : Cin), X, diag & 2 ? 32 : Cin, HW, npix, n0, smem);
    if (diag & 4) {                                               // timing diagnostics: no epilogue at all
        if (acc[0][0][0] == 123.f) Y[0] = 1;
        return;
    }

    // ---- epilogue: col2im through LDS.  Staging image Gs[tap 16][channel 8][pixel 256] fp32 = 128 KiB, two passes ------------
    float* Gs = reinterpret_cast<float*>(smem);
    const int OH = 2 * H, OW = 2 * W, OHW = OH * OW;
    const int ipr = OW / 8;                          // 8-pixel output items per output row
    const int items_per_sample = OH * ipr;           // = HW / 2
    const int lg_ipr = 31 - __builtin_clz(ipr), lg_ips = 31 - __builtin_clz(items_per_sample);
    const int cj = lane & 31, rh = 4 * (lane >> 5);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        // rows of this wave's 32-row block i: tap = 2 (4 wr + i) + (row >> 4), channel = row & 15; this pass takes channels
        // 8 pass .. 8 pass + 7, i.e. the accumulator registers v with ((v >> 2) & 1) == pass
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int vv = 0; vv < 8; ++vv) {
                    const int v = (vv & 3) + 4 * pass + 8 * (vv >> 2);
                    const int row = (v & 3) + 8 * (v >> 2) + rh;                 // 0..31 inside the block
                    const int tap = 2 * (4 * wr + i) + (row >> 4), ch = row & 7;
                    Gs[(tap * 8 + ch) * 256 + wc * 64 + 32 * j + cj] = acc[i][j][v];
                }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // 8 channels x S samples x OH x OW outputs = 8 x 128 items of 8 consecutive pixels of one output row.  Wave w owns channel
        // w of the pass (two items per lane): the BatchNorm partial sums stay in registers over both items and are reduced once
        // per wave and pass.  (First version: items dealt round-robin, a 64-lane fp64 shuffle reduction per item and 16
        // conditional scalar LDS reads per tap row: 575 of the 814 us of the 128 -> 64 @16x16 layer.)
        if (!(diag & 1)) {
            const int ch = wave;
            const int m = tm * TAP_MB + 8 * pass + ch;
            const float bv = (bias && m < Cout) ? bias[m] : 0.f;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll 1
            for (int rep = 0; rep < 2; ++rep) {
                const int rem = rep * 64 + lane;                 // H = W in {4, 8, 16}: every divisor below is a power of two
                const int s = rem >> lg_ips, r2 = rem & (items_per_sample - 1);
                const int oy = r2 >> lg_ipr, ox0 = (r2 & (ipr - 1)) * 8, x0 = ox0 >> 1;
                const int64_t b = n0 / HW + s;
                float z[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) z[e] = bv;
                // rows: oy even <- (ky 1, y = oy/2), (ky 3, y = oy/2 - 1); oy odd <- (ky 0, y = (oy+1)/2), (ky 2, y = (oy-1)/2)
#pragma unroll
                for (int jy = 0; jy < 2; ++jy) {
                    const int ky = (oy & 1) ? 2 * jy : 1 + 2 * jy;
                    const int y = (oy + 1 - ky) >> 1;
                    if (y < 0 || y >= H) continue;
                    const float* g = Gs + ((ky * 4) * 8 + ch) * 256 + s * HW + y * W + x0;      // + kx * 8 * 256 per tap column
                    // even outputs ox0 + 2q <- (kx 1, x = x0 + q), (kx 3, x = x0 + q - 1); odd ox0 + 2q + 1 <- (kx 0, x0 + q + 1), (kx 2, x0 + q):
                    // four aligned 16-byte reads at x0 and the two neighbours across the item's ends
                    const f32x4 k0 = *reinterpret_cast<const f32x4*>(g), k1 = *reinterpret_cast<const f32x4*>(g + 2048);
                    const f32x4 k2 = *reinterpret_cast<const f32x4*>(g + 2 * 2048), k3 = *reinterpret_cast<const f32x4*>(g + 3 * 2048);
                    const float left = x0 > 0 ? g[3 * 2048 - 1] : 0.f, right = x0 + 4 < W ? g[4] : 0.f;
                    z[0] += k1[0] + left;   z[1] += k2[0] + k0[1];
                    z[2] += k1[1] + k3[0];  z[3] += k2[1] + k0[2];
                    z[4] += k1[2] + k3[1];  z[5] += k2[2] + k0[3];
                    z[6] += k1[3] + k3[2];  z[7] += k2[3] + right;
                }
                if (m < Cout && b < B) {
                    if (y_f32) {
                        // fp32 output (vs_convt_k4s2_tap_fwd_f32: the fp32 parity mode assembles an fp32 convolution from bf16 pieces, ops._tap_split)
                        float* yo = reinterpret_cast<float*>(Y) + (b * Cout + m) * OHW + oy * OW + ox0;
                        *reinterpret_cast<f32x4*>(yo) = f32x4{z[0], z[1], z[2], z[3]};
                        *reinterpret_cast<f32x4*>(yo + 4) = f32x4{z[4], z[5], z[6], z[7]};
#pragma unroll
                        for (int e = 0; e < 8; ++e) { s1 += z[e]; s2 += z[e] * z[e]; }
                    } else {
                        u16x8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            o[e] = vs_f2h(z[e], CT);
                            const float zr = vs_h2f(o[e], CT);                       // statistics of the STORED values, like vs_bn_stats
                            s1 += zr; s2 += zr * zr;
                        }
                        *reinterpret_cast<u16x8*>(Y + (b * Cout + m) * OHW + oy * OW + ox0) = o;
                    }
                }
            }
            // the tile's samples belong to ONE call group (Bg % S == 0 is checked on the host)
            if (sums) tap_bn_accumulate(sums, s1, s2, lane, m < Cout && n0 / HW < B, ((n0 / HW) / Bg) * Cout + m);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                            // staging area free for the next pass
    }
}

// At[tm][tap * 16 + ml][c] = W[c][tm * 16 + ml][tap]   (W: ConvTranspose2d weight [Cin][Cout][4][4], fp32 master)
template <int CT>
__global__ __launch_bounds__(256) void tap_pack_kernel(const float* W, unsigned short* At, int Cin, int Cout, int tiles_m) {
    const int64_t total = (int64_t)tiles_m * 256 * Cin;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cin);
        const int64_t r = i / Cin;
        const int row = (int)(r & 255), tm = (int)(r >> 8);
        const int tap = row >> 4, m = tm * TAP_MB + (row & 15);
        At[i] = m < Cout ? vs_f2h(W[((int64_t)c * Cout + m) * 16 + tap], CT) : (unsigned short)0;
    }
}


// ---- Conv2d k3 s1 p1 (forward, and -- with flipped, transposed weights -- its input gradient) on whole images of <= 256 pixels ----
// Same tap GEMM: rows (tap 0..8, channel 0..27) = 252 of the 256 tile rows, out[m][y][x] = sum_t G_t[m][y + ky - 1][x + kx - 1].
// Serves the 16x16 / 8x8 / 4x4 layers of the SST encoder / decoder / ConvResnet integrator (conv.py:323-426, resnet.py:53-88: half of
// the SST step's FLOPs) and of the VGG encoder / decoder (conv.py:127-171, 267-320).
constexpr int K3_MB = 28, K3_HALF = 14;

template <int CT>
__global__ __launch_bounds__(512) void conv_k3s1_tap_kernel(const unsigned short* X, const unsigned short* At, const float* bias, void* Y, int y_dtype,
                                                            double* sums, int B, int Cin, int H, int W, int Cout, int Bg, int tiles_m) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int HW = H * W;
    const int64_t npix = (int64_t)B * HW;
    const unsigned tile = big_tile_of(blockIdx.x, gridDim.x);
    const int tm = (int)(tile % (unsigned)tiles_m);
    const int64_t n0 = (int64_t)(tile / (unsigned)tiles_m) * 256;
    f32x16 acc[4][2];
    tap_k_loop<CT>(acc, At + (int64_t)tm * 256 * Cin, X, Cin, HW, npix, n0, smem);

    // staging image Gs[tap 9][channel 14][pixel 256] fp32 = 126 KiB, two passes of 14 channels.  Tile row R = tap * 28 + channel
    // is not affine in the accumulator's (block, register) index, so a 256-entry table (in the 2 KiB the image leaves free) maps
    // R -> pass << 28 | float offset of its staging row; 64 divisions by 28 per thread and pass, or 64 precomputed addresses held
    // in registers across the passes, both cost more than 64 LDS reads.
    float* Gs = reinterpret_cast<float*>(smem);
    int* lut = reinterpret_cast<int*>(smem + 126 * 1024);
    if (threadIdx.x < 256) {
        const int R = threadIdx.x, tap = R / K3_MB, ml = R - tap * K3_MB;
        lut[R] = R < 9 * K3_MB ? ((ml / K3_HALF) << 28) | ((tap * K3_HALF + ml % K3_HALF) * 256) : (3 << 28);
    }
    __builtin_amdgcn_s_barrier();
    const int P = W >= 8 ? 8 : 4;                    // output pixels per work item (one run inside an image row)
    const int ipr = W / P, items_per_sample = H * ipr;
    const int items = K3_HALF * (256 / P);           // 448 (P = 8) or 896 (P = 4) per pass
    const int cj = lane & 31, rh = 4 * (lane >> 5);
    const int* lrow = lut + 128 * wr + rh;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int e = lrow[32 * i + (v & 3) + 8 * (v >> 2)];
                if ((e >> 28) == pass) {
                    float* dst = Gs + (e & 0xfffffff) + wc * 64 + cj;
                    dst[0] = acc[i][0][v];
                    dst[32] = acc[i][1][v];
                }
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int item = (int)threadIdx.x; item < items; item += 512) {
            const int per_ch = 256 / P;
            const int ch = item / per_ch, rem = item - ch * per_ch;
            const int s = rem / items_per_sample, r2 = rem - s * items_per_sample;
            const int y = r2 / ipr, x0 = (r2 - y * ipr) * P;
            const int m = tm * K3_MB + pass * K3_HALF + ch;
            const int64_t b = n0 / HW + s;
            float z[8];
            const float bv = (bias && m < Cout) ? bias[m] : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = bv;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int yy = y + ky - 1;
                if (yy < 0 || yy >= H) continue;
                const float* g = Gs + ((ky * 3) * K3_HALF + ch) * 256 + s * HW + yy * W + x0;     // + kx * 14 * 256 per tap column
                const float* g0 = g, * g1 = g + K3_HALF * 256, * g2 = g + 2 * K3_HALF * 256;
                // out[x0 + e] += G_kx1[x0 + e] + G_kx0[x0 + e - 1] + G_kx2[x0 + e + 1]: aligned 16-byte reads at x0 of the three tap
                // rows plus the two neighbours across the item's ends
                float a[8], c[8], r[8];
                *reinterpret_cast<f32x4*>(a) = *reinterpret_cast<const f32x4*>(g0);
                *reinterpret_cast<f32x4*>(c) = *reinterpret_cast<const f32x4*>(g1);
                *reinterpret_cast<f32x4*>(r) = *reinterpret_cast<const f32x4*>(g2);
                if (P == 8) {
                    *reinterpret_cast<f32x4*>(a + 4) = *reinterpret_cast<const f32x4*>(g0 + 4);
                    *reinterpret_cast<f32x4*>(c + 4) = *reinterpret_cast<const f32x4*>(g1 + 4);
                    *reinterpret_cast<f32x4*>(r + 4) = *reinterpret_cast<const f32x4*>(g2 + 4);
                }
                const float left = x0 > 0 ? g0[-1] : 0.f, right = x0 + P < W ? g2[P] : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (e >= P) break;
                    z[e] += c[e] + (e ? a[e - 1] : left) + (e + 1 < P ? r[e + 1] : right);
                }
            }
            float s1 = 0.f, s2 = 0.f;
            const bool live = m < Cout && b < B;
            if (live) {
                const int64_t o = (b * Cout + m) * HW + y * W + x0;
                if (y_dtype == VS_F32) {
                    float* yo = (float*)Y + o;
                    for (int e = 0; e < P; ++e) { yo[e] = z[e]; s1 += z[e]; s2 += z[e] * z[e]; }
                } else {
                    unsigned short h16[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        h16[e] = vs_f2h(z[e], y_dtype);
                        if (e < P) { const float zr = vs_h2f(h16[e], y_dtype); s1 += zr; s2 += zr * zr; }
                    }
                    if (P == 8) *reinterpret_cast<u16x8*>((unsigned short*)Y + o) = *reinterpret_cast<u16x8*>(h16);
                    else *reinterpret_cast<u16x4*>((unsigned short*)Y + o) = *reinterpret_cast<u16x4*>(h16);
                }
            }
            if (sums) {
                // lanes of one wave may work on two channels here (448 items per pass do not split evenly): reduce per lane group
                // of equal channel with a segmented pattern -- items of a channel are consecutive, so compare with the wave's first lane
                const int ch0 = __builtin_amdgcn_readfirstlane(ch);
                const int chl = __builtin_amdgcn_readlane(ch, 63);
                tap_bn_accumulate(sums, ch == ch0 ? s1 : 0.f, ch == ch0 ? s2 : 0.f, lane,
                                  tm * K3_MB + pass * K3_HALF + ch0 < Cout && n0 / HW < B && item - lane < items,
                                  ((n0 / HW) / Bg) * Cout + tm * K3_MB + pass * K3_HALF + ch0);
                if (chl != ch0)
                    tap_bn_accumulate(sums, ch == chl ? s1 : 0.f, ch == chl ? s2 : 0.f, lane,
                                      tm * K3_MB + pass * K3_HALF + chl < Cout && n0 / HW < B && chl < K3_HALF,
                                      ((n0 / HW) / Bg) * Cout + tm * K3_MB + pass * K3_HALF + chl);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

// At[tm][tap * 28 + ml][c]: forward W[m][c][ky][kx] (tap = ky * 3 + kx); input gradient (flip = 1, W is [c][m][3][3] seen from the
// gradient's side: rows are the conv's INPUT channels, K its output channels) W[c][m][2 - ky][2 - kx]
template <int CT>
__global__ __launch_bounds__(256) void tap_pack_k3_kernel(const float* W, unsigned short* At, int Cin, int Cout, int tiles_m, int flip) {
    const int64_t total = (int64_t)tiles_m * 256 * Cin;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cin);
        const int64_t r = i / Cin;
        const int row = (int)(r & 255), tm = (int)(r >> 8);
        const int tap = row / K3_MB, m = tm * K3_MB + (row - tap * K3_MB);
        float v = 0.f;
        if (row < 9 * K3_MB && m < Cout)
            v = flip ? W[((int64_t)c * Cout + m) * 9 + (8 - tap)] : W[((int64_t)m * Cin + c) * 9 + tap];
        At[i] = vs_f2h(v, CT);
    }
}

// mean / invstd / unbiased variance of every (group, channel) from the fp64 sums of the conv epilogue
__global__ __launch_bounds__(256) void bn_from_sums_kernel(const double* sums, int GC, double n, float eps, float* mean, float* invstd, float* ubvar) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= GC) return;
    const double ts = sums[2 * i], tq = sums[2 * i + 1];
    const double mu = ts / n;
    double ss = tq - ts * mu;
    if (ss < 0.0) ss = 0.0;
    const double var = ss / n;
    mean[i] = (float)mu;
    invstd[i] = (float)(1.0 / sqrt(var + (double)eps));
    if (ubvar) ubvar[i] = (float)(n > 1.0 ? ss / (n - 1.0) : var);
}

__global__ __launch_bounds__(256) void bn_running_from_groups_kernel(const float* mean, const float* ubvar, int G, int C, float* rmean, float* rvar,
                                                                     float momentum) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double rm = rmean[c], rv = rvar[c];
    for (int g = 0; g < G; ++g) {                       // sequential, in call order; fp32 rounding after every call like the reference
        rm = (double)(float)((1.0 - momentum) * rm + momentum * (double)mean[g * C + c]);
        rv = (double)(float)((1.0 - momentum) * rv + momentum * (double)ubvar[g * C + c]);
    }
    rmean[c] = (float)rm;
    rvar[c] = (float)rv;
}

// mean / invstd of every (call group, channel) from the epilogue sums AND the running estimates folded in call order, one thread per channel;
// reset: the sums are left at zero for the next step (a persistent buffer then needs no fill launch)
__global__ __launch_bounds__(256) void bn_from_sums_fold_kernel(double* sums, int G, int C, double n, float eps, float* mean, float* invstd, float* rmean,
                                                                float* rvar, float momentum, int reset) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double rm = rmean ? (double)rmean[c] : 0.0, rv = rvar ? (double)rvar[c] : 0.0;
    for (int g = 0; g < G; ++g) {
        const int i = g * C + c;
        const double ts = sums[2 * i], tq = sums[2 * i + 1];
        const double mu = ts / n;
        double ss = tq - ts * mu;
        if (ss < 0.0) ss = 0.0;
        const double var = ss / n;
        mean[i] = (float)mu;
        invstd[i] = (float)(1.0 / sqrt(var + (double)eps));
        const float ub = (float)(n > 1.0 ? ss / (n - 1.0) : var);
        rm = (double)(float)((1.0 - momentum) * rm + momentum * (double)(float)mu);
        rv = (double)(float)((1.0 - momentum) * rv + momentum * (double)ub);
        if (reset) { sums[2 * i] = 0.0; sums[2 * i + 1] = 0.0; }
    }
    if (rmean) { rmean[c] = (float)rm; rvar[c] = (float)rv; }
}

// The same from a table of per-workgroup partial sums parts[rows][C][2] (fp32; the rows of call group g are g * rpg .. (g + 1) * rpg - 1).  Grid =
// (channel octets, call groups), block = 8 channels x 32 row lanes: a lane adds every 32nd row of its group in fp64, the 32 lanes of a channel
// meet in LDS in a fixed order (reproducible launch to launch); mean / invstd / the unbiased variance per (group, channel).  The running
// estimates are folded in call order by bn_running_fold_kernel behind it (a first form with ONE block per 32 channels that walked the groups
// itself left the chip to 2-16 workgroups: 15 us per layer, TaxiBJ 8.1 -> 8.7 ms).
__global__ __launch_bounds__(256) void bn_from_parts_kernel(const float* __restrict__ parts, int rpg, int C, double n, float eps, float* __restrict__ mean,
                                                            float* __restrict__ invstd, float* __restrict__ ubvar) {
    __shared__ double red[32][8][2];
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3, c = blockIdx.x * 8 + cl, g = blockIdx.y;
    double ts = 0.0, tq = 0.0;
    if (c < C) {
        const float* p = parts + ((int64_t)g * rpg * C + c) * 2;
        for (int r = rl; r < rpg; r += 32) {
            const float2 v = *reinterpret_cast<const float2*>(p + (int64_t)r * C * 2);
            ts += (double)v.x;
            tq += (double)v.y;
        }
    }
    red[rl][cl][0] = ts;
    red[rl][cl][1] = tq;
    __syncthreads();
    if (rl == 0 && c < C) {
        for (int k = 1; k < 32; ++k) { ts += red[k][cl][0]; tq += red[k][cl][1]; }
        const int i = g * C + c;
        const double mu = ts / n;
        double ss = tq - ts * mu;
        if (ss < 0.0) ss = 0.0;
        const double var = ss / n;
        mean[i] = (float)mu;
        invstd[i] = (float)(1.0 / sqrt(var + (double)eps));
        if (ubvar) ubvar[i] = (float)(n > 1.0 ? ss / (n - 1.0) : var);
    }
}

__global__ __launch_bounds__(256) void bn_running_fold_kernel(const float* __restrict__ mean, const float* __restrict__ ubvar, int G, int C, float* rmean,
                                                              float* rvar, float momentum) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double rm = rmean[c], rv = rvar[c];
    for (int g = 0; g < G; ++g) {                       // sequential, in call order; fp32 rounding after every call like the reference
        rm = (double)(float)((1.0 - momentum) * rm + momentum * (double)mean[g * C + c]);
        rv = (double)(float)((1.0 - momentum) * rv + momentum * (double)ubvar[g * C + c]);
    }
    rmean[c] = (float)rm;
    rvar[c] = (float)rv;
}

}  // namespace

extern "C" int vs_bn_stats_from_parts_fold(const float* parts, int rows_per_group, int groups, int C, int64_t n_per_group, float* mean, float* invstd,
                                           float* var_scratch, float* running_mean, float* running_var, float momentum, float eps, void* stream) {
    VS_CHECK_ARG(parts && mean && invstd && groups >= 1 && rows_per_group >= 1 && C > 0 && n_per_group > 0, "vs_bn_stats_from_parts_fold: bad argument");
    VS_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "vs_bn_stats_from_parts_fold: running_mean/var must come together");
    VS_CHECK_ARG(!running_mean || var_scratch, "vs_bn_stats_from_parts_fold: the running update needs var_scratch [groups][C]");
    VS_CHECK_ARG((uintptr_t)parts % 8 == 0, "vs_bn_stats_from_parts_fold: the table must be 8-byte aligned");
    hipLaunchKernelGGL(bn_from_parts_kernel, dim3((unsigned)vs_cdiv(C, 8), (unsigned)groups), dim3(256), 0, (hipStream_t)stream, parts, rows_per_group, C,
                       (double)n_per_group, eps, mean, invstd, running_mean ? var_scratch : nullptr);
    if (running_mean)
        hipLaunchKernelGGL(bn_running_fold_kernel, dim3((unsigned)vs_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, mean, var_scratch, groups, C,
                           running_mean, running_var, momentum);
    VS_CHECK_LAUNCH("vs_bn_stats_from_parts_fold");
    return VS_OK;
}

extern "C" int vs_convt_tap_supported(int compute, int B, int Cin, int H, int W, int Cout, int groups) {
    if (!vs_is16(compute)) return 0;
    const int HW = H * W;
    if (H != W || (HW != 16 && HW != 64 && HW != 256)) return 0;          // whole images in a 256-pixel tile
    if (Cin % 8 != 0 || Cin < 32 || Cout < 8) return 0;
    if (groups < 1 || B % groups != 0) return 0;
    const int S = 256 / HW;
    if ((B / groups) % S != 0) return 0;                                    // a tile never straddles two BatchNorm call groups
    return 1;
}

extern "C" size_t vs_convt_tap_packed_elems(int Cin, int Cout) { return (size_t)vs_cdiv(Cout, TAP_MB) * 256 * (size_t)Cin; }

extern "C" int vs_convt_tap_pack_weight(int compute, const float* w, int Cin, int Cout, void* dst, void* stream) {
    VS_CHECK_ARG(vs_is16(compute) && w && dst && Cin > 0 && Cout > 0, "vs_convt_tap_pack_weight: bad argument");
    const int tiles_m = (int)vs_cdiv(Cout, TAP_MB);
    int64_t blocks = vs_cdiv((int64_t)tiles_m * 256 * Cin, 256);
    if (blocks > 2048) blocks = 2048;
    if (compute == VS_BF16) hipLaunchKernelGGL(tap_pack_kernel<VS_BF16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)dst, Cin, Cout, tiles_m);
    else hipLaunchKernelGGL(tap_pack_kernel<VS_F16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)dst, Cin, Cout, tiles_m);
    VS_CHECK_LAUNCH("vs_convt_tap_pack_weight");
    return VS_OK;
}

namespace {
int convt_tap_go(int compute, const void* x, const void* w_tap, const float* bias, void* y, int y_f32, double* bn_sums, int B, int Cin, int H,
                 int W, int Cout, int groups, void* stream) {
    VS_CHECK_ARG(x && w_tap && y, "vs_convt_k4s2_tap_fwd: null pointer");
    VS_CHECK_ARG(vs_convt_tap_supported(compute, B, Cin, H, W, Cout, groups), "vs_convt_k4s2_tap_fwd: unsupported geometry (query vs_convt_tap_supported)");
    VS_CHECK_ARG(((uintptr_t)x | (uintptr_t)w_tap | (uintptr_t)y) % 16 == 0, "vs_convt_k4s2_tap_fwd: operands must be 16-byte aligned");
    const int tiles_m = (int)vs_cdiv(Cout, TAP_MB);
    const int64_t tiles_px = vs_cdiv((int64_t)B * H * W, 256);
    VS_CHECK_ARG(tiles_px * tiles_m < (1ll << 31), "vs_convt_k4s2_tap_fwd: too many tiles");
    if (bn_sums) {
        if (vs_zero_async(bn_sums, (size_t)groups * Cout * 2 * sizeof(double), (hipStream_t)stream) != hipSuccess)
            return vs_fail(VS_ERR_LAUNCH, "vs_convt_k4s2_tap_fwd: zero fill failed");
    }
    auto kb = convt_k4s2_tap_kernel<VS_BF16>;
    auto kh = convt_k4s2_tap_kernel<VS_F16>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, BIG_STAGES * BIG_TILE_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void*)kh, hipFuncAttributeMaxDynamicSharedMemorySize, BIG_STAGES * BIG_TILE_BYTES) != hipSuccess)
            return vs_fail(VS_ERR_LAUNCH, "vs_convt_k4s2_tap_fwd: cannot raise the dynamic LDS limit");
        attr_set = true;
    }
    dim3 grid((unsigned)(tiles_px * tiles_m));
    const char* denv = getenv("VS_TAP_DIAG");                     // timing diagnostics only (wrong results)
    const int diag = denv ? atoi(denv) : 0;
    if (compute == VS_BF16)
        hipLaunchKernelGGL(kb, grid, dim3(512), BIG_STAGES * BIG_TILE_BYTES, (hipStream_t)stream, (const unsigned short*)x, (const unsigned short*)w_tap, bias,
                           (unsigned short*)y, bn_sums, B, Cin, H, W, Cout, B / groups, tiles_m, diag, y_f32);
    else
        hipLaunchKernelGGL(kh, grid, dim3(512), BIG_STAGES * BIG_TILE_BYTES, (hipStream_t)stream, (const unsigned short*)x, (const unsigned short*)w_tap, bias,
                           (unsigned short*)y, bn_sums, B, Cin, H, W, Cout, B / groups, tiles_m, diag, y_f32);
    VS_CHECK_LAUNCH("vs_convt_k4s2_tap_fwd");
    return VS_OK;
}
}  // namespace

extern "C" int vs_convt_k4s2_tap_fwd(int compute, const void* x, const void* w_tap, const float* bias, void* y, double* bn_sums, int B, int Cin, int H,
                                     int W, int Cout, int groups, void* stream) {
    return convt_tap_go(compute, x, w_tap, bias, y, 0, bn_sums, B, Cin, H, W, Cout, groups, stream);
}

// The same launch with an fp32 output tensor y [B, Cout, 2H, 2W] (no BatchNorm sums): used by the fp32 parity mode that assembles an fp32
// transposed convolution from bf16 pieces of its operands (bilinearity; reference layers conv.py:260-263 and the input gradient of :119-122).
extern "C" int vs_convt_k4s2_tap_fwd_f32(int compute, const void* x, const void* w_tap, const float* bias, float* y, int B, int Cin, int H, int W,
                                         int Cout, void* stream) {
    return convt_tap_go(compute, x, w_tap, bias, y, 1, nullptr, B, Cin, H, W, Cout, 1, stream);
}


extern "C" int vs_conv_k3_tap_supported(int compute, int B, int Cin, int H, int W, int Cout, int groups) {
    if (!vs_is16(compute)) return 0;
    const int HW = H * W;
    if (H != W || (HW != 16 && HW != 64 && HW != 256)) return 0;
    if (Cin % 8 != 0 || Cin < 32 || Cout < 8) return 0;
    if (groups < 1 || B % groups != 0) return 0;
    if ((B / groups) % (256 / HW) != 0) return 0;
    return 1;
}

extern "C" size_t vs_conv_k3_tap_packed_elems(int Cin, int Cout) { return (size_t)vs_cdiv(Cout, K3_MB) * 256 * (size_t)Cin; }

// flip = 0: w is the Conv2d weight [Cout][Cin][3][3] (forward).  flip = 1: input gradient -- pass the SAME weight tensor with
// Cin := the conv's Cout (the contraction) and Cout := the conv's Cin (the rows): w is then read as [Cin][Cout][3][3], taps flipped.
extern "C" int vs_conv_k3_tap_pack_weight(int compute, const float* w, int Cin, int Cout, int flip, void* dst, void* stream) {
    VS_CHECK_ARG(vs_is16(compute) && w && dst && Cin > 0 && Cout > 0, "vs_conv_k3_tap_pack_weight: bad argument");
    const int tiles_m = (int)vs_cdiv(Cout, K3_MB);
    int64_t blocks = vs_cdiv((int64_t)tiles_m * 256 * Cin, 256);
    if (blocks > 2048) blocks = 2048;
    if (compute == VS_BF16) hipLaunchKernelGGL(tap_pack_k3_kernel<VS_BF16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)dst, Cin, Cout, tiles_m, flip);
    else hipLaunchKernelGGL(tap_pack_k3_kernel<VS_F16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)dst, Cin, Cout, tiles_m, flip);
    VS_CHECK_LAUNCH("vs_conv_k3_tap_pack_weight");
    return VS_OK;
}

extern "C" int vs_conv_k3s1_tap_fwd(int compute, const void* x, const void* w_tap, const float* bias, void* y, int y_dtype, double* bn_sums, int B,
                                    int Cin, int H, int W, int Cout, int groups, void* stream) {
    VS_CHECK_ARG(x && w_tap && y && vs_dtype_ok(y_dtype), "vs_conv_k3s1_tap_fwd: bad argument");
    VS_CHECK_ARG(vs_conv_k3_tap_supported(compute, B, Cin, H, W, Cout, groups), "vs_conv_k3s1_tap_fwd: unsupported geometry (query vs_conv_k3_tap_supported)");
    VS_CHECK_ARG(((uintptr_t)x | (uintptr_t)w_tap | (uintptr_t)y) % 16 == 0, "vs_conv_k3s1_tap_fwd: operands must be 16-byte aligned");
    const int tiles_m = (int)vs_cdiv(Cout, K3_MB);
    const int64_t tiles_px = vs_cdiv((int64_t)B * H * W, 256);
    VS_CHECK_ARG(tiles_px * tiles_m < (1ll << 31), "vs_conv_k3s1_tap_fwd: too many tiles");
    if (bn_sums) {
        if (vs_zero_async(bn_sums, (size_t)groups * Cout * 2 * sizeof(double), (hipStream_t)stream) != hipSuccess)
            return vs_fail(VS_ERR_LAUNCH, "vs_conv_k3s1_tap_fwd: zero fill failed");
    }
    auto kb = conv_k3s1_tap_kernel<VS_BF16>;
    auto kh = conv_k3s1_tap_kernel<VS_F16>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, BIG_STAGES * BIG_TILE_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void*)kh, hipFuncAttributeMaxDynamicSharedMemorySize, BIG_STAGES * BIG_TILE_BYTES) != hipSuccess)
            return vs_fail(VS_ERR_LAUNCH, "vs_conv_k3s1_tap_fwd: cannot raise the dynamic LDS limit");
        attr_set = true;
    }
    dim3 grid((unsigned)(tiles_px * tiles_m));
    if (compute == VS_BF16)
        hipLaunchKernelGGL(kb, grid, dim3(512), BIG_STAGES * BIG_TILE_BYTES, (hipStream_t)stream, (const unsigned short*)x, (const unsigned short*)w_tap, bias,
                           y, y_dtype, bn_sums, B, Cin, H, W, Cout, B / groups, tiles_m);
    else
        hipLaunchKernelGGL(kh, grid, dim3(512), BIG_STAGES * BIG_TILE_BYTES, (hipStream_t)stream, (const unsigned short*)x, (const unsigned short*)w_tap, bias,
                           y, y_dtype, bn_sums, B, Cin, H, W, Cout, B / groups, tiles_m);
    VS_CHECK_LAUNCH("vs_conv_k3s1_tap_fwd");
    return VS_OK;
}

// vs_bn_stats_from_sums in ONE launch (statistics of every call group and the running estimates folded in call order); reset != 0 leaves the
// sums at zero, so a persistent sums buffer needs no fill launch before the next convolution adds into it
extern "C" int vs_bn_stats_from_sums_fold(double* sums, int groups, int C, int64_t n_per_group, float* mean, float* invstd, float* running_mean,
                                          float* running_var, float momentum, float eps, int reset, void* stream) {
    VS_CHECK_ARG(sums && mean && invstd && groups >= 1 && C > 0 && n_per_group > 0, "vs_bn_stats_from_sums_fold: bad argument");
    VS_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "vs_bn_stats_from_sums_fold: running_mean/var must come together");
    hipLaunchKernelGGL(bn_from_sums_fold_kernel, dim3((unsigned)vs_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, sums, groups, C, (double)n_per_group, eps,
                       mean, invstd, running_mean, running_var, momentum, reset);
    VS_CHECK_LAUNCH("vs_bn_stats_from_sums_fold");
    return VS_OK;
}

extern "C" int vs_bn_stats_from_sums(const double* sums, int groups, int C, int64_t n_per_group, float* mean, float* invstd, float* var_scratch,
                                     float* running_mean, float* running_var, float momentum, float eps, void* stream) {
    VS_CHECK_ARG(sums && mean && invstd && groups >= 1 && C > 0 && n_per_group > 0, "vs_bn_stats_from_sums: bad argument");
    VS_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "vs_bn_stats_from_sums: running_mean/var must come together");
    VS_CHECK_ARG(!running_mean || var_scratch, "vs_bn_stats_from_sums: var_scratch [groups*C] is needed to update running statistics");
    hipLaunchKernelGGL(bn_from_sums_kernel, dim3((unsigned)vs_cdiv((int64_t)groups * C, 256)), dim3(256), 0, (hipStream_t)stream, sums, groups * C,
                       (double)n_per_group, eps, mean, invstd, running_mean ? var_scratch : nullptr);
    VS_CHECK_LAUNCH("vs_bn_stats_from_sums");
    if (running_mean) {
        hipLaunchKernelGGL(bn_running_from_groups_kernel, dim3((unsigned)vs_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, mean, var_scratch, groups, C,
                           running_mean, running_var, momentum);
        VS_CHECK_LAUNCH("vs_bn_stats_from_sums running update");
    }
    return VS_OK;
}
