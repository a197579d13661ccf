// vs_conv_thin.hip -- convolutions between a map with MANY channels and the image side with a FEW (1..8) (gfx950 only).
//
// Reference layers: the first encoder layer and the last decoder layer of every convolutional family --
//   DCGAN64Encoder.c1 = Conv2d(nc, 64, 4, 2, 1) (networks/conv.py:119), DCGAN64Decoder.upc5 = ConvTranspose2d(64, nc, 4, 2, 1) (conv.py:264-267),
//   VGG64Encoder.c1[0] = Conv2d(nc, 64, 3, 1, 1) (conv.py:130-133), VGG64Decoder's last ConvTranspose2d(64, nc, 3, 1, 1) (conv.py:300-303),
//   EncoderSST's first Conv2d(nc, 64, 3, 1, 1) and DecoderSST's last Conv2d(64, nc, 3, 1, 1) (conv.py:345-426).
// As GEMMs these have one dimension of 1..8 (x 9 or 16 taps): the MFMA tiles of the general kernels are empty (the weight gradient of
// SST's last layer, one output channel against 64 x 9 and 1.3 M pixels, took 630 us per launch for 1.5 GFLOP) and the column-matrix forms move 9..16 x the
// tensor.  All of them are bound by ONE pass over the many-channel map; the kernels here make exactly that pass, on the VALU.
//
// Geometry (pad = 1 everywhere): big [B][C][H][W], thin [B][M][S H][S W] with (k, S) = (3, 1) or (4, 2); a big pixel (y, x) meets the thin
// pixels (S y + ty - 1, S x + tx - 1), ty, tx < k.  With Wc[c][m][t] the weight seen from that side,
//     expand : big[c][p]     = bias[c] + sum_{m, t} thin[m][S p + t - 1] Wc[c][m][t]      Conv2d forward (few inputs); ConvT / Conv2d(s1) input gradient (few outputs)
//     reduce : thin[m][q]    = bias[m] + sum_{c, t : S p + t - 1 = q} big[c][p] Wc[c][m][t]  ConvT forward / Conv2d(s1) forward with few outputs
//     wgrad  : G[c][m][t]    = sum_{maps, p} big[c][p] thin[m][S p + t - 1]                 both weight gradients of both
// The caller passes the weight tensor with the two strides (c, m) and a flip flag (t -> k^2 - 1 - t: a stride-1 Conv2d seen from its output).
//
// Work unit = a BAND of 512 consecutive pixels of one big map (512 / W whole rows): lane l of every wave owns the 16-byte piece l of the
// band, the four waves of a workgroup own different channels.  expand / wgrad stage the thin rows the band meets in LDS (zero rows / columns
// outside the map) and read per piece a k x (8 S + k - S) window from there; reduce reads the three big rows of a piece directly (16-byte
// loads, the +-1 pixel from the neighbour lane by DPP) and adds the four waves' partial sums through LDS.
#include "vs_common.h"

namespace {

typedef unsigned short u16;

constexpr int TH_BAND = 512;       // pixels of the big map per band
constexpr int TH_PAD = 8;          // zero elements left and right of a staged thin row (keeps the 16-byte reads aligned)

template <int CT>
__device__ __forceinline__ void th_cvt8(const u32x4 raw, float* v) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        if constexpr (CT == VS_BF16) {
            v[2 * d] = __uint_as_float(raw[d] << 16);
            v[2 * d + 1] = __uint_as_float(raw[d] & 0xffff0000u);
        } else {
            v[2 * d] = vs_h2f((u16)(raw[d] & 0xffffu), VS_F16);
            v[2 * d + 1] = vs_h2f((u16)(raw[d] >> 16), VS_F16);
        }
    }
}

__device__ __forceinline__ unsigned th_dpp_shr1(unsigned v) {   // lane i <- lane i - 1 inside its 16-lane row, 0 into lane 0 of the row
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
}
__device__ __forceinline__ unsigned th_dpp_shl1(unsigned v) {   // lane i <- lane i + 1, 0 into lane 15
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true);
}

struct ThinGeo {
    int B, C, H, W, M;             // big [B][C][H][W]; thin [B][M][S H][S W]
    int bands_per_map;             // H * W / 512
};

// thin rows of one band in LDS: [m][row][TH_PAD + S W + TH_PAD], rows outside the map are zero.
// the window of lane `lane`'s piece for thin channel m: win[ty][j] = thin[S y + ty - 1][S x0 - 1 + j], j < 8 S + KK - S
template <int CT, int S, int KK>
__device__ __forceinline__ void th_window(const u16* __restrict__ lds_m, int W, int lane, float (*win)[8 * S + KK - S]) {
    const int ppr = W >> 3, rl = lane / ppr, x0 = (lane % ppr) * 8, pitch = S * W + 2 * TH_PAD;
#pragma unroll
    for (int ty = 0; ty < KK; ++ty) {
        const u16* row = lds_m + (S * rl + ty) * pitch + TH_PAD + S * x0;
        win[ty][0] = vs_h2f(row[-1], CT);
#pragma unroll
        for (int h = 0; h < S; ++h) th_cvt8<CT>(*reinterpret_cast<const u32x4*>(row + 8 * h), &win[ty][1 + 8 * h]);
        win[ty][1 + 8 * S] = vs_h2f(row[8 * S], CT);
    }
}

// ================================================================ expand ==========================================================================
template <int CT, int S, int KK>
__global__ __launch_bounds__(256) void thin_expand_kernel(const u16* __restrict__ thin, const u16* __restrict__ w, int64_t w_sc, int64_t w_sm, int flip,
                                                          const float* __restrict__ bias, void* __restrict__ out, int od, ThinGeo g) {
    constexpr int WN = 8 * S + KK - S, T = KK * KK, CH = S == 1 ? 8 : 4;        // CH channels share one read of the window
    extern __shared__ __attribute__((aligned(16))) unsigned char th_smem[];
    const int W = g.W, M = g.M, C = g.C, Ht = S * g.H, Wt = S * W;
    const int NR = S * (TH_BAND / W) + KK - S, pitch = Wt + 2 * TH_PAD, ppr_t = Wt >> 3;
    u16* tb = reinterpret_cast<u16*>(th_smem);                                   // [M][NR][pitch]
    float* wl = reinterpret_cast<float*>(th_smem + (((size_t)M * NR * pitch * 2 + 15) & ~(size_t)15));   // [C][M][T]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int band = blockIdx.x % g.bands_per_map, b = blockIdx.x / g.bands_per_map;
    const int y0 = band * (TH_BAND / W);
    for (int i = tid; i < C * M * T; i += 256) {
        const int t = i % T, m = (i / T) % M, c = i / (T * M);
        wl[i] = vs_h2f(w[c * w_sc + m * w_sm + (flip ? T - 1 - t : t)], CT);
    }
    for (int i = tid; i < M * NR * 2; i += 256) {                                // the pads: 16 bytes each side of every row
        u16* row = tb + (size_t)(i >> 1) * pitch + ((i & 1) ? TH_PAD + Wt : 0);
        *reinterpret_cast<u32x4*>(row) = u32x4{0u, 0u, 0u, 0u};
    }
    for (int i = tid; i < M * NR * ppr_t; i += 256) {
        const int pc = i % ppr_t, r = (i / ppr_t) % NR, m = i / (ppr_t * NR);
        const int ty = S * y0 - 1 + r;
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (ty >= 0 && ty < Ht) v = *reinterpret_cast<const u32x4*>(thin + (((int64_t)b * M + m) * Ht + ty) * Wt + pc * 8);
        *reinterpret_cast<u32x4*>(tb + ((size_t)m * NR + r) * pitch + TH_PAD + pc * 8) = v;
    }
    __syncthreads();
    const int cpw = C >> 2;                                                      // channels of this wave (C a multiple of 32)
    const int64_t obase = ((int64_t)b * C) * g.H * W + (int64_t)band * TH_BAND + lane * 8;
    for (int c0 = wave * cpw; c0 < (wave + 1) * cpw; c0 += CH) {
        float acc[CH][8];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const float bv = bias ? bias[c0 + j] : 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[j][i] = bv;
        }
        for (int m = 0; m < M; ++m) {
            float win[KK][WN];
            th_window<CT, S, KK>(tb + (size_t)m * NR * pitch, W, lane, win);
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const float* wq = wl + ((c0 + j) * M + m) * T;                   // wave-uniform address: an LDS broadcast
#pragma unroll
                for (int ty = 0; ty < KK; ++ty)
#pragma unroll
                    for (int tx = 0; tx < KK; ++tx) {
                        const float wv = wq[ty * KK + tx];
#pragma unroll
                        for (int i = 0; i < 8; ++i) acc[j][i] += wv * win[ty][S * i + tx];
                    }
            }
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int64_t o = obase + (int64_t)(c0 + j) * g.H * W;
            if (od == VS_F32) {
                float* op = (float*)out + o;
                *reinterpret_cast<f32x4*>(op) = f32x4{acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
                *reinterpret_cast<f32x4*>(op + 4) = f32x4{acc[j][4], acc[j][5], acc[j][6], acc[j][7]};
            } else {
                u32x4 pk;
#pragma unroll
                for (int d = 0; d < 4; ++d) pk[d] = (unsigned)vs_f2h(acc[j][2 * d], od) | ((unsigned)vs_f2h(acc[j][2 * d + 1], od) << 16);
                *reinterpret_cast<u32x4*>((u16*)out + o) = pk;
            }
        }
    }
}

// ================================================================ wgrad ===========================================================================
// Workgroup = (worker, channel group of 4 CPW channels, thin-channel group of MT): wave w owns channels 4 CPW g + CPW w .. of the band the
// workgroup is on; the bands worker, worker + NW, ... of the batch are walked with the next band's global loads issued before the FMAs of
// the current one.  Every lane keeps CPW x MT x k^2 sums over ITS pieces; at the end a wave reduction and one fp32 partial per worker:
// partials [NW][C][Mp][k^2] (Mp = M rounded up to MT), summed in a fixed order by thin_wgrad_finish_kernel (no atomics: bit-reproducible).
template <int CT, int S, int KK, int MT, int CPW>
__global__ __launch_bounds__(256) void thin_wgrad_kernel(const u16* __restrict__ big, const u16* __restrict__ thin, float* __restrict__ partials, ThinGeo g,
                                                         int cgroups, int mgroups, int NW) {
    constexpr int WN = 8 * S + KK - S, T = KK * KK;
    constexpr int NTP = KK == 3 ? 1 : (MT == 1 ? 2 : 3);                         // staged 16-byte pieces per thread (W <= 128)
    extern __shared__ __attribute__((aligned(16))) unsigned char th_smem[];
    u16* tb = reinterpret_cast<u16*>(th_smem);                                   // [MT][NR][pitch]
    const int W = g.W, M = g.M, C = g.C, Ht = S * g.H, Wt = S * W;
    const int NR = S * (TH_BAND / W) + KK - S, pitch = Wt + 2 * TH_PAD, ppr_t = Wt >> 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int id = blockIdx.x;
    const int mg = id % mgroups;
    id /= mgroups;
    const int cg = id % cgroups, worker = id / cgroups;
    const int c_base = (cg * 4 + wave) * CPW, m_base = mg * MT;
    const int64_t nbands = (int64_t)g.B * g.bands_per_map;
    const int HW = g.H * W;

    for (int i = tid; i < MT * NR * 2; i += 256) {
        u16* row = tb + (size_t)(i >> 1) * pitch + ((i & 1) ? TH_PAD + Wt : 0);
        *reinterpret_cast<u32x4*>(row) = u32x4{0u, 0u, 0u, 0u};
    }

    float acc[CPW][MT][T];
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < T; ++t) acc[c][m][t] = 0.f;

    u32x4 nb[CPW], nt[NTP];
    auto fetch = [&](int64_t bd) {
        const int band = (int)(bd % g.bands_per_map);
        const int64_t b = bd / g.bands_per_map;
        const int y0 = band * (TH_BAND / W);
#pragma unroll
        for (int c = 0; c < CPW; ++c) nb[c] = *reinterpret_cast<const u32x4*>(big + (b * C + c_base + c) * HW + (int64_t)band * TH_BAND + lane * 8);
#pragma unroll
        for (int r = 0; r < NTP; ++r) {
            const int i = r * 256 + tid;
            const int pc = i % ppr_t, rr = (i / ppr_t) % NR, m = i / (ppr_t * NR);
            const int ty = S * y0 - 1 + rr;
            nt[r] = u32x4{0u, 0u, 0u, 0u};
            if (m < MT && m_base + m < M && ty >= 0 && ty < Ht)
                nt[r] = *reinterpret_cast<const u32x4*>(thin + ((b * M + m_base + m) * Ht + ty) * Wt + pc * 8);
        }
    };
    if (worker < nbands) fetch(worker);
    for (int64_t bd = worker; bd < nbands; bd += NW) {
        __syncthreads();                                                         // nobody reads the previous band's rows any more
#pragma unroll
        for (int r = 0; r < NTP; ++r) {
            const int i = r * 256 + tid;
            const int pc = i % ppr_t, rr = (i / ppr_t) % NR, m = i / (ppr_t * NR);
            if (m < MT) *reinterpret_cast<u32x4*>(tb + ((size_t)m * NR + rr) * pitch + TH_PAD + pc * 8) = nt[r];
        }
        u32x4 cur[CPW];
#pragma unroll
        for (int c = 0; c < CPW; ++c) cur[c] = nb[c];
        __syncthreads();
        if (bd + NW < nbands) fetch(bd + NW);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float win[KK][WN];
            th_window<CT, S, KK>(tb + (size_t)m * NR * pitch, W, lane, win);
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
                float bv[8];
                th_cvt8<CT>(cur[c], bv);
#pragma unroll
                for (int ty = 0; ty < KK; ++ty)
#pragma unroll
                    for (int tx = 0; tx < KK; ++tx) {
                        float s = 0.f;
#pragma unroll
                        for (int i = 0; i < 8; ++i) s += bv[i] * win[ty][S * i + tx];
                        acc[c][m][ty * KK + tx] += s;
                    }
            }
        }
    }
    const int Mp = mgroups * MT;
    float* dst = partials + (((int64_t)worker * C + c_base) * Mp + m_base) * T;
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < T; ++t) {
                float v = acc[c][m][t];
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
                if (lane == 0) dst[((int64_t)c * Mp + m) * T + t] = v;
            }
}

// out[c o_sc + m o_sm + (flip ? T - 1 - t : t)] = (addend ? addend[same] : 0) + sum_w partials[w][c][m][t]; 64 elements x 4 worker slices per workgroup
__global__ __launch_bounds__(256) void thin_wgrad_finish_kernel(const float* __restrict__ partials, int NW, int C, int Mp, int M, int T,
                                                                const float* __restrict__ addend, float* __restrict__ out, int64_t o_sc, int64_t o_sm,
                                                                int flip) {
    __shared__ float red[4][64];
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6, total = C * Mp * T;
    float s = 0.f;
    if (e < total)
#pragma unroll 8
        for (int w = sl; w < NW; w += 4) s += partials[(int64_t)w * total + e];
    red[sl][threadIdx.x & 63] = s;
    __syncthreads();
    if (sl == 0 && e < total) {
        const int t = e % T, m = (e / T) % Mp, c = e / (T * Mp);
        if (m < M) {
            const int64_t o = c * o_sc + m * o_sm + (flip ? T - 1 - t : t);
            const float v = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
            out[o] = (addend ? addend[o] : 0.f) + v;
        }
    }
}

// ================================================================ reduce ==========================================================================
// thin output from the many-channel map: lane = piece of the band, wave w = channels w C / 4 .. ; per channel the three rows y - 1, y, y + 1 of the
// piece (16-byte loads; the +-1 pixel from the neighbour lane by DPP -- a map row is W / 8 <= 16 lanes and never straddles a DPP row) and
//   (3, 1): acc[m][i]          += Wc[c][m][2 - ty][2 - tx] * row[ty][i + tx - 1]                          -> thin[m][y][x0 + i]
//   (4, 2): acc[m][py][px][i]  += Wc[c][m][ky][kx] * row[dy][i + dx - 1], py = (ky + 1) & 1, dy = (py + 1 - ky) / 2 + 1 (same for x) -> thin[m][2 y + py][2 (x0 + i) + px]
// then the four waves' sums through LDS, bias, typed 16-byte stores.
template <int CT, int S, int KK, int MT>
__global__ __launch_bounds__(256) void thin_reduce_kernel(const u16* __restrict__ big, const u16* __restrict__ w, int64_t w_sc, int64_t w_sm, int flip,
                                                          const float* __restrict__ bias, void* __restrict__ out, int od, ThinGeo g) {
    constexpr int T = KK * KK, NACC = MT * S * S * 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char th_smem[];
    const int W = g.W, M = g.M, C = g.C, H = g.H, HW = H * W;
    float* wl = reinterpret_cast<float*>(th_smem);                               // [C][MT][T]
    float* red = wl + ((C * MT * T + 3) & ~3);                                   // [3 waves][NACC][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int band = blockIdx.x % g.bands_per_map, b = blockIdx.x / g.bands_per_map;
    for (int i = tid; i < C * MT * T; i += 256) {
        const int t = i % T, m = (i / T) % MT, c = i / (T * MT);
        wl[i] = m < M ? vs_h2f(w[c * w_sc + m * w_sm + (flip ? T - 1 - t : t)], CT) : 0.f;
    }
    __syncthreads();
    const int ppr = W >> 3, R = TH_BAND / W, rl = lane / ppr, x0 = (lane % ppr) * 8, y = band * R + rl;
    const bool has_l = x0 > 0, has_r = x0 + 8 < W, up = y > 0, dn = y + 1 < H;
    float acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
    const int cpw = C >> 2;
    const u16* src = big + ((int64_t)b * C + wave * cpw) * HW + (int64_t)y * W + x0;
    for (int cc = 0; cc < cpw; cc += 2) {
        u32x4 raw[2][3];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const u16* p = src + (int64_t)(cc + u) * HW;
            raw[u][1] = *reinterpret_cast<const u32x4*>(p);
            raw[u][0] = up ? *reinterpret_cast<const u32x4*>(p - W) : u32x4{0u, 0u, 0u, 0u};
            raw[u][2] = dn ? *reinterpret_cast<const u32x4*>(p + W) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float xs[3][10];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                th_cvt8<CT>(raw[u][d], &xs[d][1]);
                const unsigned lw = th_dpp_shr1(raw[u][d][3]), rw = th_dpp_shl1(raw[u][d][0]);
                xs[d][0] = has_l ? vs_h2f((u16)(lw >> 16), CT) : 0.f;
                xs[d][9] = has_r ? vs_h2f((u16)(rw & 0xffffu), CT) : 0.f;
            }
            const float* wq = wl + (wave * cpw + cc + u) * MT * T;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if constexpr (S == 1) {
#pragma unroll
                    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                        for (int tx = 0; tx < 3; ++tx) {
                            const float wv = wq[m * T + (2 - ty) * 3 + (2 - tx)];      // the big pixel at offset (ty - 1, tx - 1) reaches q with tap 2 - t
#pragma unroll
                            for (int i = 0; i < 8; ++i) acc[m * 8 + i] += wv * xs[ty][i + tx];
                        }
                } else {
#pragma unroll
                    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 4; ++kx) {
                            const int py = (ky + 1) & 1, px = (kx + 1) & 1, dy = (py + 1 - ky) / 2 + 1, dx = (px + 1 - kx) / 2 + 1;
                            const float wv = wq[m * T + ky * 4 + kx];
#pragma unroll
                            for (int i = 0; i < 8; ++i) acc[((m * 2 + py) * 2 + px) * 8 + i] += wv * xs[dy][i + dx];
                        }
                }
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) red[((wave - 1) * NACC + i) * 64 + lane] = acc[i];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = ((acc[i] + red[i * 64 + lane]) + red[(NACC + i) * 64 + lane]) + red[(2 * NACC + i) * 64 + lane];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (m >= M) break;
            const float bv = bias ? bias[m] : 0.f;
#pragma unroll
            for (int py = 0; py < S; ++py) {
                float v[8 * S];
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int px = 0; px < S; ++px) v[S * i + px] = acc[S == 1 ? m * 8 + i : ((m * 2 + py) * 2 + px) * 8 + i] + bv;
                const int64_t o = (((int64_t)b * M + m) * (S * H) + S * y + py) * (S * W) + S * x0;
                if (od == VS_F32) {
                    float* op = (float*)out + o;
#pragma unroll
                    for (int q = 0; q < 2 * S; ++q) *reinterpret_cast<f32x4*>(op + 4 * q) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
                } else {
#pragma unroll
                    for (int q = 0; q < S; ++q) {
                        u32x4 pk;
#pragma unroll
                        for (int d = 0; d < 4; ++d) pk[d] = (unsigned)vs_f2h(v[8 * q + 2 * d], od) | ((unsigned)vs_f2h(v[8 * q + 2 * d + 1], od) << 16);
                        *reinterpret_cast<u32x4*>((u16*)out + o + 8 * q) = pk;
                    }
                }
            }
        }
    }
}

bool thin_geo_ok(int compute, int B, int C, int H, int W, int M, int k, int stride, int pad) {
    if (!vs_is16(compute) || pad != 1 || !((k == 3 && stride == 1) || (k == 4 && stride == 2))) return false;
    if (B < 1 || M < 1 || M > 8 || C < 32 || C % 32 != 0 || C > 1024) return false;
    if (W < 8 || W > 128 || TH_BAND % W != 0 || H < 1 || ((int64_t)H * W) % TH_BAND != 0) return false;
    if ((int64_t)B * C * H * W * stride * stride >= (1ll << 40)) return false;
    if ((int64_t)B * (H * W / TH_BAND) >= (1ll << 30)) return false;
    // LDS budgets of the expand / reduce launches (their weight tables are C x M x k^2 fp32: 512 x 8 x 16 would be 256 KiB): a geometry is
    // "supported" only if every kernel of the family fits, so that the dispatcher's predicate and the launchers agree
    const int NR = stride * (TH_BAND / W) + k - stride, pitch = stride * W + 2 * TH_PAD;
    const size_t lds_expand = (((size_t)M * NR * pitch * 2 + 15) & ~(size_t)15) + (size_t)C * M * k * k * 4;
    const int MT = M == 1 ? 1 : (M == 2 ? 2 : 4);
    const size_t lds_reduce = (size_t)((C * MT * k * k + 3) & ~3) * 4 + (size_t)3 * MT * stride * stride * 8 * 64 * 4;
    if (lds_expand > 160 * 1024 || lds_reduce > 160 * 1024) return false;
    return true;
}

ThinGeo thin_geo(int B, int C, int H, int W, int M) { return ThinGeo{B, C, H, W, M, H * W / TH_BAND}; }

template <typename K>
int thin_lds(K kernel, size_t lds, const char* what) {
    if (lds > 160 * 1024) return vs_fail(VS_ERR_ARG, "%s: the staged rows need %zu bytes of LDS", what, lds);
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return vs_fail(VS_ERR_LAUNCH, "%s: cannot raise the dynamic LDS limit", what);
    return VS_OK;
}

}  // namespace

extern "C" int vs_conv_thin_supported(int compute, int B, int C, int H, int W, int M, int k, int stride, int pad) {
    return thin_geo_ok(compute, B, C, H, W, M, k, stride, pad) ? 1 : 0;
}

// the forward / input-gradient kernels additionally want their weights (C x M x k^2 fp32) and rows in LDS
extern "C" int vs_conv_thin_expand(int compute, const void* thin, const void* w, int64_t w_sc, int64_t w_sm, int flip, const float* bias, void* out,
                                   int out_dtype, int B, int C, int H, int W, int M, int k, int stride, void* stream) {
    VS_CHECK_ARG(thin && w && out && vs_dtype_ok(out_dtype), "vs_conv_thin_expand: bad argument");
    VS_CHECK_ARG(thin_geo_ok(compute, B, C, H, W, M, k, stride, 1), "vs_conv_thin_expand: unsupported geometry (query vs_conv_thin_supported)");
    VS_CHECK_ARG(((uintptr_t)thin | (uintptr_t)out) % 16 == 0, "vs_conv_thin_expand: operands must be 16-byte aligned");
    const ThinGeo g = thin_geo(B, C, H, W, M);
    const int NR = stride * (TH_BAND / W) + k - stride, pitch = stride * W + 2 * TH_PAD;
    const size_t lds = (((size_t)M * NR * pitch * 2 + 15) & ~(size_t)15) + (size_t)C * M * k * k * 4;
    const dim3 grid((unsigned)((int64_t)B * g.bands_per_map));
    hipStream_t st = (hipStream_t)stream;
#define VS_THIN_EXPAND(CTV, SV, KV)                                                                                                     \
    do {                                                                                                                                \
        auto kern = thin_expand_kernel<CTV, SV, KV>;                                                                                    \
        const int rc = thin_lds(kern, lds, "vs_conv_thin_expand");                                                                      \
        if (rc != VS_OK) return rc;                                                                                                     \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const u16*)thin, (const u16*)w, w_sc, w_sm, flip, bias, out, out_dtype, g); \
    } while (0)
    if (compute == VS_BF16) {
        if (k == 3) VS_THIN_EXPAND(VS_BF16, 1, 3);
        else VS_THIN_EXPAND(VS_BF16, 2, 4);
    } else {
        if (k == 3) VS_THIN_EXPAND(VS_F16, 1, 3);
        else VS_THIN_EXPAND(VS_F16, 2, 4);
    }
#undef VS_THIN_EXPAND
    VS_CHECK_LAUNCH("vs_conv_thin_expand");
    return VS_OK;
}

extern "C" int vs_conv_thin_reduce(int compute, const void* big, const void* w, int64_t w_sc, int64_t w_sm, int flip, const float* bias, void* out,
                                   int out_dtype, int B, int C, int H, int W, int M, int k, int stride, void* stream) {
    VS_CHECK_ARG(big && w && out && vs_dtype_ok(out_dtype), "vs_conv_thin_reduce: bad argument");
    VS_CHECK_ARG(thin_geo_ok(compute, B, C, H, W, M, k, stride, 1) && M <= (k == 3 ? 4 : 2),
                 "vs_conv_thin_reduce: unsupported geometry (query vs_conv_thin_supported; at most 4 (k = 3) / 2 (k = 4) thin channels)");
    VS_CHECK_ARG(((uintptr_t)big | (uintptr_t)out) % 16 == 0, "vs_conv_thin_reduce: operands must be 16-byte aligned");
    const ThinGeo g = thin_geo(B, C, H, W, M);
    const int MT = M == 1 ? 1 : (M == 2 ? 2 : 4), T = k * k, NACC = MT * stride * stride * 8;
    const size_t lds = (size_t)((C * MT * T + 3) & ~3) * 4 + (size_t)3 * NACC * 64 * 4;
    const dim3 grid((unsigned)((int64_t)B * g.bands_per_map));
    hipStream_t st = (hipStream_t)stream;
#define VS_THIN_REDUCE(CTV, SV, KV, MV)                                                                                                \
    do {                                                                                                                                \
        auto kern = thin_reduce_kernel<CTV, SV, KV, MV>;                                                                                \
        const int rc = thin_lds(kern, lds, "vs_conv_thin_reduce");                                                                      \
        if (rc != VS_OK) return rc;                                                                                                     \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const u16*)big, (const u16*)w, w_sc, w_sm, flip, bias, out, out_dtype, g);  \
    } while (0)
#define VS_THIN_REDUCE_CT(CTV)                                 \
    do {                                                       \
        if (k == 3) {                                          \
            if (MT == 1) VS_THIN_REDUCE(CTV, 1, 3, 1);         \
            else if (MT == 2) VS_THIN_REDUCE(CTV, 1, 3, 2);    \
            else VS_THIN_REDUCE(CTV, 1, 3, 4);                 \
        } else {                                               \
            if (MT == 1) VS_THIN_REDUCE(CTV, 2, 4, 1);         \
            else VS_THIN_REDUCE(CTV, 2, 4, 2);                 \
        }                                                      \
    } while (0)
    if (compute == VS_BF16) VS_THIN_REDUCE_CT(VS_BF16);
    else VS_THIN_REDUCE_CT(VS_F16);
#undef VS_THIN_REDUCE_CT
#undef VS_THIN_REDUCE
    VS_CHECK_LAUNCH("vs_conv_thin_reduce");
    return VS_OK;
}

namespace {
struct ThinWgradPlan { int MT, CPW, cgroups, mgroups, NW; };
ThinWgradPlan thin_wgrad_plan(int B, int C, int H, int W, int M, int k) {
    ThinWgradPlan p;
    p.MT = M == 1 ? 1 : 2;
    p.CPW = (k == 4 && p.MT == 2) ? 2 : 4;
    p.cgroups = C / (4 * p.CPW);
    p.mgroups = (M + p.MT - 1) / p.MT;
    const int64_t nbands = (int64_t)B * (H * W / TH_BAND);
    int64_t nw = 512 / ((int64_t)p.cgroups * p.mgroups);           // 512 workgroups: two per CU are resident (200 VGPRs), each walks >= 4 bands
    if (nw < 1) nw = 1;
    if (nw > nbands) nw = nbands;
    p.NW = (int)nw;
    return p;
}
}  // namespace

extern "C" size_t vs_conv_thin_wgrad_workspace_bytes(int B, int C, int H, int W, int M, int k) {
    if (B < 1 || C < 32 || W < 8 || TH_BAND % W != 0 || ((int64_t)H * W) % TH_BAND != 0 || M < 1 || (k != 3 && k != 4)) return 0;
    const ThinWgradPlan p = thin_wgrad_plan(B, C, H, W, M, k);
    return (size_t)p.NW * C * (p.mgroups * p.MT) * k * k * 4;
}

// out[c o_sc + m o_sm + (flip ? k^2 - 1 - t : t)] = (addend ? addend[..] : 0) + sum_{maps, pixels} big[c][p] thin[m][S p + t - 1]; `addend` may be `out`
extern "C" int vs_conv_thin_wgrad(int compute, const void* big, const void* thin, float* ws, size_t ws_bytes, const float* addend, float* out, int64_t o_sc,
                                  int64_t o_sm, int flip, int B, int C, int H, int W, int M, int k, int stride, void* stream) {
    VS_CHECK_ARG(big && thin && ws && out, "vs_conv_thin_wgrad: bad argument");
    VS_CHECK_ARG(thin_geo_ok(compute, B, C, H, W, M, k, stride, 1), "vs_conv_thin_wgrad: unsupported geometry (query vs_conv_thin_supported)");
    VS_CHECK_ARG(((uintptr_t)big | (uintptr_t)thin | (uintptr_t)ws) % 16 == 0, "vs_conv_thin_wgrad: operands must be 16-byte aligned");
    VS_CHECK_ARG(ws_bytes >= vs_conv_thin_wgrad_workspace_bytes(B, C, H, W, M, k), "vs_conv_thin_wgrad: workspace too small (vs_conv_thin_wgrad_workspace_bytes)");
    const ThinGeo g = thin_geo(B, C, H, W, M);
    const ThinWgradPlan p = thin_wgrad_plan(B, C, H, W, M, k);
    const int NR = stride * (TH_BAND / W) + k - stride, pitch = stride * W + 2 * TH_PAD;
    const size_t lds = (size_t)p.MT * NR * pitch * 2;
    const dim3 grid((unsigned)((int64_t)p.NW * p.cgroups * p.mgroups));
    hipStream_t st = (hipStream_t)stream;
#define VS_THIN_WGRAD(CTV, SV, KV, MV, CV)                                                                                              \
    do {                                                                                                                                \
        auto kern = thin_wgrad_kernel<CTV, SV, KV, MV, CV>;                                                                             \
        const int rc = thin_lds(kern, lds, "vs_conv_thin_wgrad");                                                                       \
        if (rc != VS_OK) return rc;                                                                                                     \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const u16*)big, (const u16*)thin, ws, g, p.cgroups, p.mgroups, p.NW);       \
    } while (0)
#define VS_THIN_WGRAD_CT(CTV)                                  \
    do {                                                       \
        if (k == 3) {                                          \
            if (p.MT == 1) VS_THIN_WGRAD(CTV, 1, 3, 1, 4);     \
            else VS_THIN_WGRAD(CTV, 1, 3, 2, 4);               \
        } else {                                               \
            if (p.MT == 1) VS_THIN_WGRAD(CTV, 2, 4, 1, 4);     \
            else VS_THIN_WGRAD(CTV, 2, 4, 2, 2);               \
        }                                                      \
    } while (0)
    if (compute == VS_BF16) VS_THIN_WGRAD_CT(VS_BF16);
    else VS_THIN_WGRAD_CT(VS_F16);
#undef VS_THIN_WGRAD_CT
#undef VS_THIN_WGRAD
    VS_CHECK_LAUNCH("vs_conv_thin_wgrad");
    const int Mp = p.mgroups * p.MT, total = C * Mp * k * k;
    hipLaunchKernelGGL(thin_wgrad_finish_kernel, dim3((unsigned)vs_cdiv(total, 64)), dim3(256), 0, st, (const float*)ws, p.NW, C, Mp, M, k * k, addend, out, o_sc,
                       o_sm, flip);
    VS_CHECK_LAUNCH("vs_conv_thin_wgrad (finish)");
    return VS_OK;
}
