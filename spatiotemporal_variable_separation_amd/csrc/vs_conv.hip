// vs_conv.hip -- Conv2d / ConvTranspose2d (forward, input gradient, weight gradient) as im2col-free implicit GEMMs
// on the shared MFMA contraction kernel (vs_gemm_core.h).  Activations stay NCHW in HBM; there is no im2col buffer
// and no activation re-layout: operand loaders gather straight from the NCHW tensors into LDS tiles.
//
// Orientation: output CHANNELS are the GEMM rows, output PIXELS (b, y, x) the GEMM columns, so that in the 32x32 MFMA
// accumulator (column on the lane) consecutive lanes hold consecutive pixels of one channel: NCHW stores are
// contiguous runs per register, and the per-channel bias is a row constant.
//
// Every contraction is   out[m, pixel] = sum_{c, tap} Wd[m][(c, tap)] * src[b, c, gy*s + dy(tap), gx*s + dx(tap)]
//   * the pixel operand is a TAP GATHER: for one (c, tap) a 16-byte LDS unit is 8 (bf16) / 4 (fp32) consecutive pixels
//     of one image row, i.e. consecutive (s = 1) or every-other (s = 2) source elements -> one or two UNALIGNED 16-byte
//     global loads (gfx950 runs in unaligned-access mode) instead of per-element loads; only units that touch the zero
//     padding or straddle rows take the per-element path;
//   * the weight operand is always a dense row-major matrix Wd[M][C*ntap]: the reference layout itself for Conv2d
//     forward and ConvTranspose2d input-gradient, a pre-packed copy (vs_conv_pack_weight: channel dims swapped, taps
//     selected per phase, converted to the compute type, refreshed once per optimizer step) for the others;
//   * stride-2 transposed convolutions (ConvTranspose2d forward, Conv2d input-gradient, k4 s2 p1) are split into the
//     4 output-parity PHASES: each phase is a dense 2x2-tap stride-1 contraction over the small grid that scatters to
//     every other output pixel -- no multiplications by structural zeros (4x fewer MACs than the gather form);
//   * weight gradients reduce over pixels: dy / x enter as channel-major rows (vector loads), the other tensor as a
//     tap gather with the pixel on the reduction axis; long reductions use split-K.
//
//   op                    M      N (grid)          K              weights                  gathered tensor
//   conv   forward        Cout   B*OH*OW           Cin*kh*kw      w (reference layout)     x,  s = stride
//   conv   dgrad          Cin    B*H*W | phases    Cout*ntap      packed (swap)            dy, s = 1
//   conv   wgrad          Cout   Cin*kh*kw         B*OH*OW        --                       x,  s = stride (pixel = reduction)
//   convT  forward        Cout   B*OH*OW | phases  Cin*ntap       packed (swap)            x,  s = 1
//   convT  dgrad          Cin    B*H*W             Cout*kh*kw     w (reference layout)     dy, s = stride
//   convT  wgrad          Cin    Cout*kh*kw        B*H*W          --                       dy, s = stride (pixel = reduction)
//
// Reference call sites: every nn.Conv2d / nn.ConvTranspose2d of networks/conv.py:119-122,147-170,258-263,294-318,
// 326-343,362-382,402-417 and networks/resnet.py:57-59, plus their autograd.
#include <type_traits>
#include "vs_gemm_core.h"
#include "vs_gemm_mid.h"
#include <stdlib.h>

namespace {

constexpr int MAXTAP = 32;       // 5x5 kernels (the chairs ResNet18 stem, conv.py:514) have 25 taps

struct TapGeo {
    int B, C, H, W;        // the NCHW tensor gathered from
    int GH, GW;            // iteration grid (one GEMM column / reduction index per (b, gy, gx))
    int s;                 // source coordinate = grid coordinate * s + tap offset
    int ntap;
    signed char dy[MAXTAP], dx[MAXTAP];   // host-side tap list; the device reads the nibble-packed copies below
    // index-math helpers filled by finish(): shifts when the grid is a power of two (-1 otherwise), multiply-high magic for / ntap
    int gw_shift, ghw_shift;
    unsigned ntap_magic;
    // tap t -> ((tab[t / 16] >> 4 (t % 16)) & 15) - 8   (offsets are within [-8, 7]); four scalars, 16 taps per 64-bit word
    unsigned long long dy_tab[2], dx_tab[2];
    __device__ __forceinline__ int tap_dy(int t) const { return (int)(((t < 16 ? dy_tab[0] : dy_tab[1]) >> (4 * (t & 15))) & 15ull) - 8; }
    __device__ __forceinline__ int tap_dx(int t) const { return (int)(((t < 16 ? dx_tab[0] : dx_tab[1]) >> (4 * (t & 15))) & 15ull) - 8; }
    void finish() {
        dy_tab[0] = dy_tab[1] = dx_tab[0] = dx_tab[1] = 0;
        for (int t = 0; t < ntap; ++t) {
            dy_tab[t >> 4] |= (unsigned long long)((dy[t] + 8) & 15) << (4 * (t & 15));
            dx_tab[t >> 4] |= (unsigned long long)((dx[t] + 8) & 15) << (4 * (t & 15));
        }
        gw_shift = ghw_shift = -1;
        for (int k = 0; k < 31; ++k) {
            if ((1 << k) == GW) gw_shift = k;
            if ((1 << k) == GH * GW) ghw_shift = k;
        }
        ntap_magic = (unsigned)((0x100000000ull + ntap - 1) / ntap);      // floor(q / ntap) = umulhi(q, magic) for q < 2^16
    }
    __device__ __forceinline__ void split_q(int q, int& c, int& t) const {
        if (ntap == 1) { c = q; t = 0; return; }                        // the magic for 1 would be 2^32: does not fit 32 bits
        c = (q < 65536) ? (int)__umulhi((unsigned)q, ntap_magic) : q / ntap;
        t = q - c * ntap;
    }
    __device__ __forceinline__ void split_pix(int64_t pix, int& b, int& gy, int& gx) const {
        int rem;
        if (ghw_shift >= 0) { b = (int)(pix >> ghw_shift); rem = (int)(pix & ((1 << ghw_shift) - 1)); }
        else { const int ghw = GH * GW; b = (int)(pix / ghw); rem = (int)(pix - (int64_t)b * ghw); }
        if (gw_shift >= 0) { gy = rem >> gw_shift; gx = rem & ((1 << gw_shift) - 1); }
        else { gy = rem / GW; gx = rem - gy * GW; }
    }
};

// src[b, c, gy*s + dy[t], gx*s + dx[t]] for U consecutive grid pixels starting at pix0 and one q = c * ntap + t
template <int CT>
struct TapGather {
    typedef typename CTraits<CT>::T T;
    static constexpr int U = CTraits<CT>::U;
    const T* src; TapGeo g; int64_t npix, nq;

    // LEN consecutive grid pixels of ONE grid row, written to dst[0..LEN)
    template <int LEN>
    __device__ __forceinline__ void segment(T* dst, int b, int c, int gy, int gx0, int dyv, int dxv) const {
        const int iy = gy * g.s + dyv;
#pragma unroll
        for (int j = 0; j < LEN; ++j) dst[j] = (T)0.f;
        if (iy < 0 || iy >= g.H) return;
        const T* row = src + (((int64_t)b * g.C + c) * g.H + iy) * g.W;
        const int ix0 = gx0 * g.s + dxv, last = ix0 + (LEN - 1) * g.s;
        if (ix0 >= 0 && last < g.W) {
            struct __attribute__((packed, aligned(sizeof(T)))) Vec { T v[LEN]; };
            if (g.s == 1) {
                const Vec t = *reinterpret_cast<const Vec*>(row + ix0);          // one unaligned vector load
#pragma unroll
                for (int j = 0; j < LEN; ++j) dst[j] = t.v[j];
            } else if (g.s == 2) {
                // elements ix0, ix0+2, ...: first half from [ix0, ix0+LEN), second half from [ix0+LEN-1, ix0+2LEN-1)
                const Vec t0 = *reinterpret_cast<const Vec*>(row + ix0);
                const Vec t1 = *reinterpret_cast<const Vec*>(row + ix0 + LEN - 1);
#pragma unroll
                for (int j = 0; j < LEN / 2; ++j) { dst[j] = t0.v[2 * j]; dst[LEN / 2 + j] = t1.v[2 * j + 1]; }
            } else {
#pragma unroll
                for (int j = 0; j < LEN; ++j) dst[j] = row[ix0 + j * g.s];
            }
        } else if (g.s == 1 && g.W >= LEN && ix0 >= -2 && last <= g.W + 1) {
            // border unit of a stride-1 gather (zero padding on one side): ONE vector load of the nearest in-range window,
            // then a register shift -- small feature maps (4..16 pixels per row) are almost all border units
            struct __attribute__((packed, aligned(sizeof(T)))) Vec { T v[LEN]; };
            const int base = ix0 < 0 ? 0 : g.W - LEN;
            const Vec t = *reinterpret_cast<const Vec*>(row + base);
            const int sh = ix0 - base;                                           // -2, -1 (left border) or +1, +2 (right border)
#pragma unroll
            for (int j = 0; j < LEN; ++j) {
                T v = (T)0.f;
                if (sh == -1) { if (j >= 1) v = t.v[j - 1]; }
                else if (sh == 1) { if (j + 1 < LEN) v = t.v[j + 1]; }
                else if (sh == -2) { if (j >= 2) v = t.v[j - 2]; }
                else if (sh == 2) { if (j + 2 < LEN) v = t.v[j + 2]; }
                dst[j] = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < LEN; ++j) {
                const int ix = ix0 + j * g.s;
                if (ix >= 0 && ix < g.W) dst[j] = row[ix];
            }
        }
    }

    // U consecutive grid pixels starting at (b, gy, gx) [= flat pix0] for channel c and tap offset (dyv, dxv)
    __device__ __forceinline__ u32x4 unit_at(int64_t pix0, int b, int gy, int gx, int c, int dyv, int dxv) const {
        T tmp[U];
        if ((g.GW % U) == 0 && pix0 + U <= npix) {
            segment<U>(tmp, b, c, gy, gx, dyv, dxv);                             // whole unit inside one grid row
        } else if (g.GW * 2 == U && gx == 0 && pix0 + U <= npix) {
            segment<U / 2>(tmp, b, c, gy, 0, dyv, dxv);                          // 4-wide grids (bf16): two rows per unit
            int gy2 = gy + 1, b2 = b;
            if (gy2 == g.GH) { gy2 = 0; ++b2; }
            segment<U / 2>(tmp + U / 2, b2, c, gy2, 0, dyv, dxv);
        } else {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                tmp[j] = (T)0.f;
                if (pix0 + j < npix) {
                    const int iy = gy * g.s + dyv, ix = gx * g.s + dxv;
                    if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) tmp[j] = src[(((int64_t)b * g.C + c) * g.H + iy) * g.W + ix];
                }
                if (++gx == g.GW) { gx = 0; if (++gy == g.GH) { gy = 0; ++b; } }
            }
        }
        return *reinterpret_cast<u32x4*>(tmp);
    }
};

// GEMM operand views of a gather: PIX_IS_ROW -> element(i = pixel, k = q), LDS layout S (unit along i);
//                                 otherwise   -> element(i = q, k = pixel), LDS layout R (unit along k).
template <int CT, bool PIX_IS_ROW>
struct GatherOp {
    __device__ __forceinline__ void shift(int64_t) {}
    typedef typename CTraits<CT>::T T;
    static constexpr int U = CTraits<CT>::U;
    static constexpr int layout = PIX_IS_ROW ? LS : LR;
    TapGather<CT> gather;
    // The index i of a unit slot is fixed along K: decode it once.
    //   PIX_IS_ROW : (a,b,c) = (b, gy, gx), `base` = element offset of source pixel (b, channel 0, gy*s, gx*s), and
    //                `interior` has bit t set when tap t reads U in-range pixels of one row (single vector load, no checks)
    //   otherwise  : (a,b,c) = (channel, dy, dx) of the fixed q
    struct State { int a, b, c; int ok; int64_t base; unsigned interior; };
    __device__ __forceinline__ State prepare(int64_t i) const {
        State st = {0, 0, 0, 0, 0, 0u};
        const TapGeo& g = gather.g;
        if (PIX_IS_ROW) {
            st.ok = i < gather.npix;
            if (st.ok) {
                g.split_pix(i, st.a, st.b, st.c);                                // (b, gy, gx)
                st.base = (((int64_t)st.a * g.C) * g.H + (int64_t)st.b * g.s) * g.W + (int64_t)st.c * g.s;
                if ((g.GW % U) == 0 && i + U <= gather.npix && g.s <= 2) {
                    for (int t = 0; t < g.ntap; ++t) {
                        const int iy = st.b * g.s + g.tap_dy(t), ix0 = st.c * g.s + g.tap_dx(t);
                        if (iy >= 0 && iy < g.H && ix0 >= 0 && ix0 + (U - 1) * g.s < g.W) st.interior |= 1u << t;
                    }
                }
            }
        } else {
            st.ok = i < gather.nq;
            if (st.ok) {
                int t;
                g.split_q((int)i, st.a, t);                                      // (c, dy, dx)
                st.b = g.tap_dy(t); st.c = g.tap_dx(t);
            }
        }
        return st;
    }
    __device__ __forceinline__ u32x4 load(const State& st, int64_t i, int64_t k) const {
        u32x4 z = {0u, 0u, 0u, 0u};
        if (!st.ok) return z;
        const TapGeo& g = gather.g;
        if (PIX_IS_ROW) {
            if (k >= gather.nq) return z;
            int c, t;
            g.split_q((int)k, c, t);
            const int dyv = g.tap_dy(t), dxv = g.tap_dx(t);
            if ((st.interior >> t) & 1u) {                                       // fast path: no bounds logic at all
                const T* p = gather.src + st.base + ((int64_t)c * g.H + dyv) * g.W + dxv;
                struct __attribute__((packed, aligned(sizeof(T)))) Vec { T v[U]; };
                if (g.s == 1) return *reinterpret_cast<const u32x4*>(reinterpret_cast<const Vec*>(p));
                const Vec t0 = *reinterpret_cast<const Vec*>(p);
                const Vec t1 = *reinterpret_cast<const Vec*>(p + U - 1);
                T tmp[U];
#pragma unroll
                for (int j = 0; j < U / 2; ++j) { tmp[j] = t0.v[2 * j]; tmp[U / 2 + j] = t1.v[2 * j + 1]; }
                return *reinterpret_cast<u32x4*>(tmp);
            }
            return gather.unit_at(i, st.a, st.b, st.c, c, dyv, dxv);
        } else {
            if (k >= gather.npix) return z;
            int b, gy, gx;
            g.split_pix(k, b, gy, gx);
            return gather.unit_at(k, b, gy, gx, st.a, st.b, st.c);
        }
    }
};

// element(m = channel, k = pixel (b, pix)) = src[(b * C + m) * HW + pix]
template <int CT>
struct ChanRows {
    __device__ __forceinline__ void shift(int64_t) {}
    typedef typename CTraits<CT>::T T;
    static constexpr int U = CTraits<CT>::U;
    static constexpr int layout = LR;
    const T* src; int64_t C, HW, K; int vec_ok; int hw_shift;
    struct State { int ok; };
    __device__ __forceinline__ State prepare(int64_t m) const { return State{m < C}; }
    __device__ __forceinline__ u32x4 load(const State& st, int64_t m, int64_t k) const {
        u32x4 z = {0u, 0u, 0u, 0u};
        if (!st.ok || k >= K) return z;
        int64_t b, r;
        if (hw_shift >= 0) { b = k >> hw_shift; r = k & (((int64_t)1 << hw_shift) - 1); }
        else { b = k / HW; r = k - b * HW; }
        if (vec_ok && r + U <= HW) return *reinterpret_cast<const u32x4*>(src + (b * C + m) * HW + r);
        T tmp[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            tmp[j] = (k + j < K) ? src[(b * C + m) * HW + r] : (T)0.f;
            if (++r == HW) { r = 0; ++b; }
        }
        return *reinterpret_cast<u32x4*>(tmp);
    }
};

template <int CT, class OpA, class OpB>
int run(const OpA& a, const OpB& b, int64_t M, int64_t N, int64_t K, const Epi& epi, void* ws, size_t ws_bytes, hipStream_t stream,
        const char* what) {
    typedef typename CTraits<CT>::T T;
    constexpr int BK = CT != VS_F32 ? 64 : 16;
    Plan plan = make_plan(CT, M, N, K);
    constexpr bool both_dense = std::is_same<OpA, Dense<CT, OpA::layout>>::value && std::is_same<OpB, Dense<CT, OpB::layout>>::value;
    if (!both_dense && plan.bm == 128 && plan.bn == 128) plan.bn = 64;   // gather operands are register hungry: 128x128 drops to 2 waves/SIMD
    if ((!both_dense || CT == VS_F32) && plan.bm == 64 && plan.bn == 128) plan.bn = 64;
    float* slabs = nullptr;
    if (plan.splits > 1) {
        const size_t need = (size_t)plan.splits * (size_t)M * (size_t)N * sizeof(float);
        if (!ws || ws_bytes < need) { plan.splits = 1; plan.k_tiles_per_split = vs_cdiv(K, BK); }   // fall back: no split
        else slabs = (float*)ws;
    }
    dim3 grid, block(256);
#define VS_LAUNCH(BM_, BN_)                                                                                               \
    {                                                                                                                     \
        constexpr size_t smem = (TileGeom<CT, OpA::layout, BM_, BK>::ELEMS + TileGeom<CT, OpB::layout, BN_, BK>::ELEMS) * sizeof(T); \
        grid = dim3((unsigned)vs_cdiv(N, BN_), (unsigned)vs_cdiv(M, BM_), (unsigned)plan.splits);                         \
        if (epi.nchw_hw > 0)                                                                                              \
            hipLaunchKernelGGL((gemm_kernel<CT, OpA, OpB, BM_, BN_, BK, true>), grid, block, smem, stream, a, b, M, N, K, \
                               (int)plan.k_tiles_per_split, epi, slabs);                                                  \
        else                                                                                                              \
            hipLaunchKernelGGL((gemm_kernel<CT, OpA, OpB, BM_, BN_, BK, false>), grid, block, smem, stream, a, b, M, N, K,\
                               (int)plan.k_tiles_per_split, epi, slabs);                                                  \
    }
    if (plan.bm == 128 && plan.bn == 128) VS_LAUNCH(128, 128)
    else if (plan.bm == 128) VS_LAUNCH(128, 64)
    else if (plan.bn == 128) { if constexpr (both_dense && CT != VS_F32) VS_LAUNCH(64, 128) }
    else VS_LAUNCH(64, 64)
#undef VS_LAUNCH
    VS_CHECK_LAUNCH(what);
    if (slabs) {
        int64_t blocks = vs_cdiv(M * N, 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, slabs, plan.splits, M, N, epi);
        VS_CHECK_LAUNCH(what);
    }
    return VS_OK;
}

inline Epi nchw_epi(void* out, int out_dtype, const float* bias, int64_t plane, int64_t channels) {
    Epi e{out, 0, out_dtype, 1.f, bias, VS_ACT_NONE, nullptr, 0, 0, VS_ACT_NONE, 0, plane, channels, 0, 0, 0, 0, 0, 0, 0};
    return e;
}
inline Epi rowmajor_epi(void* out, int64_t ldc) {
    Epi e{out, ldc, VS_F32, 1.f, nullptr, VS_ACT_NONE, nullptr, 0, 0, VS_ACT_NONE, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    return e;
}

int check_conv(const char* what, int compute, const void* a, const void* b, const void* c, int B, int Cin, int H, int W, int Cout,
               int kh, int kw, int stride, int pad) {
    VS_CHECK_ARG(vs_dtype_ok(compute), "%s: compute type %d", what, compute);
    VS_CHECK_ARG(a && b && c, "%s: null pointer", what);
    VS_CHECK_ARG(B > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0, "%s: bad geometry", what);
    VS_CHECK_ARG(kh * kw <= MAXTAP, "%s: kernel %dx%d has more than %d taps", what, kh, kw, MAXTAP);
    return VS_OK;
}

// Which (k, stride, pad) transposed geometries decompose into equal-size parity phases: every output pixel (s*i + py)
// exists for i in [0, H): needs OH == s*H, i.e. kh - 2*pad == s.  (k4 s2 p1 yes; stride 1 is the single trivial phase.)
// stride-2 transposed forms are run as 4 output-parity phases; tap offsets (py + p - ky) / 2 must fit the [-8, 7] tap tables
inline bool phase_ok(int kh, int kw, int s, int p) { return s == 2 && kh <= 9 && kw <= 9 && p <= 8 && kh * kw <= MAXTAP; }
// the geometry the original phase kernels were written for: every phase covers the whole source grid (out = 2 x in)
inline bool phase_uniform(int kh, int kw, int s, int p) { return s == 2 && kh - 2 * p == s && kw - 2 * p == s; }

// taps of output parity (py, px) of a transposed convolution: ky with (py + p - ky) % s == 0, source offset (py+p-ky)/s
inline int phase_taps(int kh, int kw, int s, int p, int py, int px, int* kidx, signed char* dy, signed char* dx) {
    int n = 0;
    for (int ky = 0; ky < kh; ++ky) {
        if ((py + p - ky) % s != 0) continue;
        for (int kx = 0; kx < kw; ++kx) {
            if ((px + p - kx) % s != 0) continue;
            kidx[n] = ky * kw + kx;
            dy[n] = (signed char)((py + p - ky) / s);
            dx[n] = (signed char)((px + p - kx) / s);
            ++n;
        }
    }
    return n;
}

// ---- tiny-M path (Cout = 1..4: the image-producing last decoder layer): one thread per 16-byte pixel unit, all K in
// registers-free streaming fashion.  HBM/L2 bound (the MFMA tile would waste 60/64 rows); weights are wave-uniform loads.
template <int CT, int MR>
__global__ __launch_bounds__(256) void gather_rowdot_kernel(Dense<CT, LR> A, GatherOp<CT, true> B, int M, int64_t N, int K, Epi epi) {
    typedef typename CTraits<CT>::T T;
    constexpr int U = CTraits<CT>::U;
    const int64_t units = (N + U - 1) / U;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < units; u += (int64_t)gridDim.x * 256) {
        const int64_t pix0 = u * U;
        const typename GatherOp<CT, true>::State st = B.prepare(pix0);
        float acc[MR][U];
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int j = 0; j < U; ++j) acc[m][j] = 0.f;
        for (int q0 = 0; q0 < K; q0 += 8) {                          // 8 independent gathers in flight per thread
            u32x4 raw[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) raw[e] = B.load(st, pix0, q0 + e);          // q >= K returns zeros
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const T* v = reinterpret_cast<const T*>(&raw[e]);
                const int q = q0 + e < K ? q0 + e : K - 1;
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    if (m < M) {
                        const float w = (float)A.p[(int64_t)m * A.ld + q];
#pragma unroll
                        for (int j = 0; j < U; ++j) acc[m][j] += w * (float)v[j];
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int64_t n = pix0 + j;
            if (n >= N) break;
            const int64_t cb = nchw_col_base(epi, n);
#pragma unroll
            for (int m = 0; m < MR; ++m)
                if (m < M) epi_store_nchw(epi, m, cb, acc[m][j]);
        }
    }
}

// ---- materialised gather ("im2col" into a transient column matrix) -------------------------------------------------------------
// cols[q][pix] = src[b, c, gy*s + dy(t), gx*s + dx(t)]  (q = c * ntap + t, pixels contiguous, row pitch ld = npix rounded up to U).
// The implicit form above pays the gather's index math and border handling inside the MFMA loop (50-75 TFLOP/s); for
// contractions with >= 64 output channels it is cheaper to run the gather ONCE as a streaming kernel (HBM-bound: 2 bytes written
// and later read per (q, pixel)) and to contract with the dense operand loaders (350-430 TFLOP/s).  The matrix lives in the
// caller's workspace and is dead after the call.
template <int CT>
__global__ __launch_bounds__(256) void im2col_kernel(TapGather<CT> gth, typename CTraits<CT>::T* cols, int64_t ld, int64_t units_per_q) {
    constexpr int U = CTraits<CT>::U;
    const int q = blockIdx.y;
    int c, t;
    gth.g.split_q(q, c, t);
    const int dyv = gth.g.tap_dy(t), dxv = gth.g.tap_dx(t);
    typename CTraits<CT>::T* row = cols + (int64_t)q * ld;
    for (int64_t pu = (int64_t)blockIdx.x * 256 + threadIdx.x; pu < units_per_q; pu += (int64_t)gridDim.x * 256) {
        const int64_t pix0 = pu * U;
        int b, gy, gx;
        gth.g.split_pix(pix0, b, gy, gx);
        *reinterpret_cast<u32x4*>(row + pix0) = gth.unit_at(pix0, b, gy, gx, c, dyv, dxv);
    }
}

// Stride-1, same-size grids with taps inside the 3x3 neighbourhood (the parity phases of the stride-2 transposed convolutions,
// 3x3 pad-1 convolutions), bf16, rows a multiple of 8 pixels: one thread owns (channel c, 8-pixel unit) and produces the units
// of ALL taps from at most three source rows, each fetched once as (left neighbour, aligned 16-byte unit, right neighbour) --
// nine independent loads, no per-tap address math, shifts by v_alignbyte.  The generic kernel above spends one dependent,
// branchy gather per (tap, unit) and reaches ~1 TB/s of column-matrix writes; this form is bound by the writes.
template <int CT>
__global__ __launch_bounds__(256) void im2col_s1_bf16_kernel(TapGather<CT> gth, typename CTraits<CT>::T* cols, int64_t ld, int64_t units_per_c) {
    typedef typename CTraits<CT>::T __bf16_t;
    const TapGeo& g = gth.g;
    const int c = blockIdx.y, ntap = g.ntap, W = g.W, H = g.H;
    for (int64_t pu = (int64_t)blockIdx.x * 256 + threadIdx.x; pu < units_per_c; pu += (int64_t)gridDim.x * 256) {
        const int64_t pix0 = pu * 8;
        int b, gy, gx;
        g.split_pix(pix0, b, gy, gx);
        const __bf16_t* plane = gth.src + ((int64_t)b * g.C + c) * H * W;
        u32x4 u[3];
        unsigned l[3], r[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {                                   // source rows gy - 1, gy, gy + 1
            const int iy = gy + d - 1;
            const bool ok = iy >= 0 && iy < H;
            const __bf16_t* row = plane + (int64_t)(ok ? iy : gy) * W + gx;
            const u32x4 v = *reinterpret_cast<const u32x4*>(row);
            const unsigned short lv = *reinterpret_cast<const unsigned short*>(row + (gx > 0 ? -1 : 0));
            const unsigned short rv = *reinterpret_cast<const unsigned short*>(row + (gx + 8 < W ? 8 : 7));
#pragma unroll
            for (int j = 0; j < 4; ++j) u[d][j] = ok ? v[j] : 0u;
            l[d] = (ok && gx > 0) ? (unsigned)lv : 0u;
            r[d] = (ok && gx + 8 < W) ? (unsigned)rv : 0u;
        }
        for (int t = 0; t < ntap; ++t) {
            const int dyv = g.tap_dy(t), dxv = g.tap_dx(t);
            const u32x4 uu = dyv < 0 ? u[0] : (dyv == 0 ? u[1] : u[2]);
            const unsigned ll = dyv < 0 ? l[0] : (dyv == 0 ? l[1] : l[2]);
            const unsigned rr = dyv < 0 ? r[0] : (dyv == 0 ? r[1] : r[2]);
            u32x4 o;
            if (dxv == 0) {
                o = uu;
            } else if (dxv < 0) {                                       // [l, u0 .. u6]
                o[0] = (uu[0] << 16) | ll;
#pragma unroll
                for (int j = 1; j < 4; ++j) o[j] = __builtin_amdgcn_alignbyte(uu[j], uu[j - 1], 2);
            } else {                                                    // [u1 .. u7, r]
#pragma unroll
                for (int j = 0; j < 3; ++j) o[j] = __builtin_amdgcn_alignbyte(uu[j + 1], uu[j], 2);
                o[3] = (uu[3] >> 16) | (rr << 16);
            }
            *reinterpret_cast<u32x4*>(cols + ((int64_t)c * ntap + t) * ld + pix0) = o;
        }
    }
}

// Stride-2 4x4 pad-1 gathers (Conv2d k4 s2 p1 forward / weight gradient, ConvTranspose2d k4 s2 p1 input / weight gradient), bf16,
// output-grid rows a multiple of 8 pixels: one thread owns (channel c, 8 grid pixels of row gy) and produces the units of all
// 16 taps from the four source rows 2 gy - 1 .. 2 gy + 2, each fetched once as (left neighbour, two aligned 16-byte units,
// right neighbour) -- 16 independent loads.  Source column of grid pixel j under tap kx is 2 (gx + j) - 1 + kx: the even /
// odd halves of the 16-element window, shifted by one element for kx = 0 and kx = 3.
template <int CT>
__global__ __launch_bounds__(256) void im2col_k4s2_bf16_kernel(TapGather<CT> gth, typename CTraits<CT>::T* cols, int64_t ld, int64_t units_per_c) {
    typedef typename CTraits<CT>::T __bf16_t;
    const TapGeo& g = gth.g;
    const int c = blockIdx.y, W = g.W, H = g.H;
    for (int64_t pu = (int64_t)blockIdx.x * 256 + threadIdx.x; pu < units_per_c; pu += (int64_t)gridDim.x * 256) {
        const int64_t pix0 = pu * 8;
        int b, gy, gx;
        g.split_pix(pix0, b, gy, gx);
        const __bf16_t* plane = gth.src + ((int64_t)b * g.C + c) * H * W;
        const int x0 = 2 * gx;                                            // window = source columns x0 - 1 .. x0 + 16
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const int iy = 2 * gy - 1 + ky;
            const bool ok = iy >= 0 && iy < H;
            const __bf16_t* row = plane + (int64_t)(ok ? iy : 0) * W + x0;
            const u32x4 a = *reinterpret_cast<const u32x4*>(row), bq = *reinterpret_cast<const u32x4*>(row + 8);
            const unsigned short lv = *reinterpret_cast<const unsigned short*>(row + (x0 > 0 ? -1 : 0));
            const unsigned short rv = *reinterpret_cast<const unsigned short*>(row + (x0 + 16 < W ? 16 : 15));
            unsigned w[8];                                                // dword i = (r[2i], r[2i+1])
#pragma unroll
            for (int j = 0; j < 4; ++j) { w[j] = ok ? a[j] : 0u; w[4 + j] = ok ? bq[j] : 0u; }
            const unsigned L = (ok && x0 > 0) ? (unsigned)lv : 0u, R = (ok && x0 + 16 < W) ? (unsigned)rv : 0u;
            u32x4 ev, od, evs, ods;                                       // evens r0,r2,..,r14 | odds r1,..,r15 | r2,..,r14,R | L,r1,..,r13
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ev[j] = (w[2 * j] & 0xffffu) | (w[2 * j + 1] << 16);
                od[j] = (w[2 * j] >> 16) | (w[2 * j + 1] & 0xffff0000u);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) evs[j] = __builtin_amdgcn_alignbyte(ev[j + 1], ev[j], 2);
            evs[3] = (ev[3] >> 16) | (R << 16);
            ods[0] = (od[0] << 16) | L;
#pragma unroll
            for (int j = 1; j < 4; ++j) ods[j] = __builtin_amdgcn_alignbyte(od[j], od[j - 1], 2);
            __bf16_t* dst = cols + ((int64_t)c * 16 + ky * 4) * ld + pix0;          // tap t = ky * 4 + kx  (natural_taps order)
            *reinterpret_cast<u32x4*>(dst) = ods;                                  // kx = 0: columns 2 j - 1
            *reinterpret_cast<u32x4*>(dst + ld) = ev;                              // kx = 1: columns 2 j
            *reinterpret_cast<u32x4*>(dst + 2 * ld) = od;                          // kx = 2: columns 2 j + 1
            *reinterpret_cast<u32x4*>(dst + 3 * ld) = evs;                         // kx = 3: columns 2 j + 2
        }
    }
}

// Stride-1 taps inside the 3x3 neighbourhood on 4 x 4 maps (the 512-channel layers of the VGG encoders / decoders: their weight gradients
// keep a column matrix): one thread owns a whole map (32 bytes, two 16-byte loads) and writes the 16 pixels of every tap -- a row shift picks
// the source row (or zeros), a column shift is a 16-bit shift of the row's 64 bits; lanes = consecutive maps, so every store instruction of a
// wave is 2 KiB contiguous.  The generic kernel issues one gather per (tap, 8 pixels): 28 us per launch at B = 200, 15 launches per TaxiBJ step.
template <int CT>
__global__ __launch_bounds__(256) void im2col_s1_w4_kernel(TapGather<CT> gth, typename CTraits<CT>::T* cols, int64_t ld, int64_t maps) {
    typedef typename CTraits<CT>::T T;
    const TapGeo& g = gth.g;
    const int c = blockIdx.y, ntap = g.ntap;
    for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < maps; b += (int64_t)gridDim.x * 256) {
        const T* plane = gth.src + (b * g.C + c) * 16;
        const u32x4 lo = *reinterpret_cast<const u32x4*>(plane), hi = *reinterpret_cast<const u32x4*>(plane + 8);
        const uint64_t r[4] = {(uint64_t)lo[0] | ((uint64_t)lo[1] << 32), (uint64_t)lo[2] | ((uint64_t)lo[3] << 32),
                               (uint64_t)hi[0] | ((uint64_t)hi[1] << 32), (uint64_t)hi[2] | ((uint64_t)hi[3] << 32)};
        for (int t = 0; t < ntap; ++t) {
            const int dyv = g.tap_dy(t), dxv = g.tap_dx(t);
            uint64_t o[4];
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                const int sy = y + dyv;
                uint64_t v = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) v = sy == k ? r[k] : v;                 // (no dynamic register indexing)
                o[y] = dxv == 0 ? v : (dxv < 0 ? v << 16 : v >> 16);
            }
            T* dst = cols + ((int64_t)c * ntap + t) * ld + b * 16;
            *reinterpret_cast<u32x4*>(dst) = u32x4{(unsigned)o[0], (unsigned)(o[0] >> 32), (unsigned)o[1], (unsigned)(o[1] >> 32)};
            *reinterpret_cast<u32x4*>(dst + 8) = u32x4{(unsigned)o[2], (unsigned)(o[2] >> 32), (unsigned)o[3], (unsigned)(o[3] >> 32)};
        }
    }
}

// k4 s2 p1 taps from 8 x 8 maps onto their 4 x 4 grid (the DCGAN encoder's c4 / decoder's upc2, conv.py:122, 260): one thread owns a map (eight
// 16-byte rows), splits every row once into its even and odd columns (tap kx reads columns 2 gx - 1 + kx: odds shifted, evens, odds, evens
// shifted) and writes the 16 grid pixels of each of the 16 taps as 32 contiguous bytes.
template <int CT>
__global__ __launch_bounds__(256) void im2col_k4s2_w8_kernel(TapGather<CT> gth, typename CTraits<CT>::T* cols, int64_t ld, int64_t maps) {
    typedef typename CTraits<CT>::T T;
    const TapGeo& g = gth.g;
    const int c = blockIdx.y;
    for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < maps; b += (int64_t)gridDim.x * 256) {
        const T* plane = gth.src + (b * g.C + c) * 64;
        uint64_t ev[8], od[8];
#pragma unroll
        for (int y = 0; y < 8; ++y) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(plane + y * 8);       // dword j = columns (2 j, 2 j + 1)
            ev[y] = (uint64_t)((v[0] & 0xffffu) | (v[1] << 16)) | ((uint64_t)((v[2] & 0xffffu) | (v[3] << 16)) << 32);
            od[y] = (uint64_t)((v[0] >> 16) | (v[1] & 0xffff0000u)) | ((uint64_t)((v[2] >> 16) | (v[3] & 0xffff0000u)) << 32);
        }
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            uint64_t e4[4], o4[4];                                                // source rows 2 gy - 1 + ky, gy = 0 .. 3 (zeros outside)
#pragma unroll
            for (int gy = 0; gy < 4; ++gy) {
                const int sy = 2 * gy - 1 + ky;                                    // compile-time after unrolling
                e4[gy] = (sy >= 0 && sy < 8) ? ev[sy < 0 ? 0 : (sy > 7 ? 7 : sy)] : 0ull;
                o4[gy] = (sy >= 0 && sy < 8) ? od[sy < 0 ? 0 : (sy > 7 ? 7 : sy)] : 0ull;
            }
            T* dst = cols + ((int64_t)c * 16 + ky * 4) * ld + b * 16;              // tap t = ky * 4 + kx (natural_taps order)
            auto put = [&](T* d, const uint64_t* q) {
                *reinterpret_cast<u32x4*>(d) = u32x4{(unsigned)q[0], (unsigned)(q[0] >> 32), (unsigned)q[1], (unsigned)(q[1] >> 32)};
                *reinterpret_cast<u32x4*>(d + 8) = u32x4{(unsigned)q[2], (unsigned)(q[2] >> 32), (unsigned)q[3], (unsigned)(q[3] >> 32)};
            };
            uint64_t os[4], es[4];
#pragma unroll
            for (int gy = 0; gy < 4; ++gy) { os[gy] = o4[gy] << 16; es[gy] = e4[gy] >> 16; }
            put(dst, os);                                                          // kx = 0: columns 2 gx - 1
            put(dst + ld, e4);                                                     // kx = 1: columns 2 gx
            put(dst + 2 * ld, o4);                                                 // kx = 2: columns 2 gx + 1
            put(dst + 3 * ld, es);                                                 // kx = 3: columns 2 gx + 2
        }
    }
}

inline bool k4s2_w8_ok(const TapGeo& g, int64_t npix, const void* src) {
    if (g.s != 2 || g.ntap != 16 || g.H != 8 || g.W != 8 || g.GH != 4 || g.GW != 4 || npix % 16 != 0 || (uintptr_t)src % 16 != 0) return false;
    for (int t = 0; t < 16; ++t)
        if (g.dy[t] != t / 4 - 1 || g.dx[t] != t % 4 - 1) return false;              // natural (unflipped) k4 p1 taps
    return true;
}

inline bool s1_w4_ok(const TapGeo& g, int64_t npix, const void* src) {
    if (g.s != 1 || g.H != 4 || g.W != 4 || g.GH != 4 || g.GW != 4 || npix % 16 != 0 || (uintptr_t)src % 16 != 0) return false;
    for (int t = 0; t < g.ntap; ++t)
        if (g.dy[t] < -1 || g.dy[t] > 1 || g.dx[t] < -1 || g.dx[t] > 1) return false;
    return true;
}

inline bool k4s2_fast_ok(const TapGeo& g, int64_t npix, const void* src) {
    if (g.s != 2 || g.ntap != 16 || g.H != 2 * g.GH || g.W != 2 * g.GW || g.GW % 8 != 0 || npix % 8 != 0 || (uintptr_t)src % 16 != 0) return false;
    for (int t = 0; t < 16; ++t)
        if (g.dy[t] != t / 4 - 1 || g.dx[t] != t % 4 - 1) return false;              // natural (unflipped) k4 p1 taps
    return true;
}

inline bool s1_fast_ok(const TapGeo& g, int64_t npix, const void* src) {
    if (g.s != 1 || g.GH != g.H || g.GW != g.W || g.W % 8 != 0 || npix % 8 != 0 || (uintptr_t)src % 16 != 0) return false;
    for (int t = 0; t < g.ntap; ++t)
        if (g.dy[t] < -1 || g.dy[t] > 1 || g.dx[t] < -1 || g.dx[t] > 1) return false;
    return true;
}

template <int CT>
inline int64_t cols_pitch(int64_t npix) { return (npix + CTraits<CT>::U - 1) / CTraits<CT>::U * CTraits<CT>::U; }

// when the column matrix pays: enough output channels that the contraction dominates the extra HBM pass
inline bool cols_worthwhile(int64_t M, int64_t npix, int64_t nq) {
    if (getenv("VS_CONV_COLS_FORCE")) return M > 4 && nq <= 65535;      // test aid: the column-matrix form at any size
    static const int64_t min_work = getenv("VS_CONV_COLS_MIN") ? atoll(getenv("VS_CONV_COLS_MIN")) : ((int64_t)1 << 20);
    return M >= 64 && npix * nq >= min_work && nq <= 65535;
}

template <int CT>
int materialise(const TapGather<CT>& gth, void* ws, hipStream_t st, const char* what) {
    constexpr int U = CTraits<CT>::U;
    const int64_t ld = cols_pitch<CT>(gth.npix), units = ld / U;
    int64_t bx = (units + 255) / 256;
    if (bx > 1024) bx = 1024;
    if constexpr (CT != VS_F32) {
        if (s1_w4_ok(gth.g, gth.npix, gth.src) && getenv("VS_IM2COL_W4") == nullptr) {
            const int64_t maps = gth.npix / 16;
            int64_t bm = (maps + 255) / 256;
            if (bm > 1024) bm = 1024;
            hipLaunchKernelGGL(im2col_s1_w4_kernel<CT>, dim3((unsigned)bm, (unsigned)gth.g.C), dim3(256), 0, st, gth, (typename CTraits<CT>::T*)ws, ld, maps);
            VS_CHECK_LAUNCH(what);
            return VS_OK;
        }
        if (k4s2_w8_ok(gth.g, gth.npix, gth.src) && getenv("VS_IM2COL_W4") == nullptr) {
            const int64_t maps = gth.npix / 16;
            int64_t bm = (maps + 255) / 256;
            if (bm > 1024) bm = 1024;
            hipLaunchKernelGGL(im2col_k4s2_w8_kernel<CT>, dim3((unsigned)bm, (unsigned)gth.g.C), dim3(256), 0, st, gth, (typename CTraits<CT>::T*)ws, ld, maps);
            VS_CHECK_LAUNCH(what);
            return VS_OK;
        }
        if (s1_fast_ok(gth.g, gth.npix, gth.src)) {
            hipLaunchKernelGGL(im2col_s1_bf16_kernel<CT>, dim3((unsigned)bx, (unsigned)gth.g.C), dim3(256), 0, st, gth, (typename CTraits<CT>::T*)ws, ld, units);
            VS_CHECK_LAUNCH(what);
            return VS_OK;
        }
        if (k4s2_fast_ok(gth.g, gth.npix, gth.src)) {
            hipLaunchKernelGGL(im2col_k4s2_bf16_kernel<CT>, dim3((unsigned)bx, (unsigned)gth.g.C), dim3(256), 0, st, gth, (typename CTraits<CT>::T*)ws, ld, units);
            VS_CHECK_LAUNCH(what);
            return VS_OK;
        }
    }
    hipLaunchKernelGGL(im2col_kernel<CT>, dim3((unsigned)bx, (unsigned)gth.nq), dim3(256), 0, st, gth, (typename CTraits<CT>::T*)ws, ld, units);
    VS_CHECK_LAUNCH(what);
    return VS_OK;
}

// ---- tiny-M, stride-1 gathers inside the 3x3 neighbourhood (the 1-2 channel image layer of the VGG / SST decoders, single
// phases of a transposed convolution): a thread owns (sample, row, 8 pixels), walks the channels, fetches the three source rows
// once per channel (left neighbour, 16-byte unit, right neighbour) and applies all taps from registers.  The row-dot kernel
// above issues one dependent gather per (channel, tap) and reaches 0.2 TB/s on these layers.
struct SmallTaps { int idx[3][3]; };               // tap index of offset (dy, dx) in [-1, 1]^2, or -1

template <int CT, int MR>
__global__ __launch_bounds__(256) void gather_small_s1_kernel(Dense<CT, LR> A, TapGather<CT> gth, SmallTaps tp, int M, int64_t N, Epi epi) {
    typedef typename CTraits<CT>::T T;
    const TapGeo& g = gth.g;
    const int W = g.W, H = g.H, C = g.C, ntap = g.ntap;
    const int64_t units = N / 8;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < units; u += (int64_t)gridDim.x * 256) {
        const int64_t pix0 = u * 8;
        int b, gy, gx;
        g.split_pix(pix0, b, gy, gx);
        float acc[MR][8];
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[m][i] = 0.f;
        for (int c = 0; c < C; ++c) {
            const T* plane = gth.src + ((int64_t)b * C + c) * H * W;
            float xs[3][10];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int iy = gy + d - 1;
                const bool ok = iy >= 0 && iy < H;
                const T* row = plane + (int64_t)(ok ? iy : gy) * W + gx;
                T v[8];
                *reinterpret_cast<u32x4*>(v) = *reinterpret_cast<const u32x4*>(row);
                if constexpr (CT == VS_F32) *reinterpret_cast<u32x4*>(v + 4) = *reinterpret_cast<const u32x4*>(row + 4);
                const T lv = row[gx > 0 ? -1 : 0], rv = row[gx + 8 < W ? 8 : 7];
                xs[d][0] = (ok && gx > 0) ? (float)lv : 0.f;
                xs[d][9] = (ok && gx + 8 < W) ? (float)rv : 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) xs[d][1 + i] = ok ? (float)v[i] : 0.f;
            }
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                if (m >= M) break;
                const T* wrow = A.p + (int64_t)m * A.ld + (int64_t)c * ntap;
#pragma unroll
                for (int d = 0; d < 3; ++d)
#pragma unroll
                    for (int e = 0; e < 3; ++e) {
                        const int t = tp.idx[d][e];                     // uniform
                        if (t < 0) continue;
                        const float wv = (float)wrow[t];
#pragma unroll
                        for (int i = 0; i < 8; ++i) acc[m][i] += wv * xs[d][i + e];
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t cb = nchw_col_base(epi, pix0 + i);
#pragma unroll
            for (int m = 0; m < MR; ++m)
                if (m < M) epi_store_nchw(epi, m, cb, acc[m][i]);
        }
    }
}

// ---- forward-like contraction:  out[b, m, gy*S+oy, gx*S+ox] = bias[m] + sum Wd[m][(c,t)] src[b, c, gy*s+dy, gx*s+dx] ----
template <int CT>
int gather_gemm(const void* src, const void* wd, const float* bias, void* out, int out_dtype, int M, const TapGeo& g, int OH, int OW, int scat,
                int oy, int ox, void* ws, size_t ws_bytes, hipStream_t st, const char* what, bool cols_ready = false) {
    typedef typename CTraits<CT>::T T;
    const int64_t N = (int64_t)g.B * g.GH * g.GW, K = (int64_t)g.C * g.ntap;
    Dense<CT, LR> a{(const T*)wd, K, M, K, ((uintptr_t)wd % 16 == 0) && (K % CTraits<CT>::U == 0)};
    GatherOp<CT, true> b;
    b.gather.src = (const T*)src; b.gather.g = g; b.gather.g.finish(); b.gather.npix = N; b.gather.nq = K;
    Epi e = nchw_epi(out, out_dtype, bias, (int64_t)OH * OW, M);
    if (scat != 1 || g.GH != OH || g.GW != OW) {
        e.g_w = g.GW; e.g_hw = g.GH * g.GW; e.o_w = OW; e.sy = scat; e.sx = scat; e.oy = oy; e.ox = ox;
    }
    if (M <= 2 && s1_fast_ok(b.gather.g, N, src) && getenv("VS_CONV_SMALL") == nullptr) {
        SmallTaps tp;
        for (int d = 0; d < 3; ++d)
            for (int e2 = 0; e2 < 3; ++e2) tp.idx[d][e2] = -1;
        for (int t = 0; t < g.ntap; ++t) tp.idx[g.dy[t] + 1][g.dx[t] + 1] = t;
        int64_t blocks = vs_cdiv(N / 8, 256);
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL((gather_small_s1_kernel<CT, 2>), dim3((unsigned)blocks), dim3(256), 0, st, a, b.gather, tp, M, N, e);
        VS_CHECK_LAUNCH(what);
        return VS_OK;
    }
    if (M <= 4 && K <= 65536) {
        int64_t blocks = vs_cdiv(vs_cdiv(N, CTraits<CT>::U), 256);
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL((gather_rowdot_kernel<CT, 4>), dim3((unsigned)blocks), dim3(256), 0, st, a, b, M, N, (int)K, e);
        VS_CHECK_LAUNCH(what);
        return VS_OK;
    }
    const size_t cols_bytes = (size_t)K * cols_pitch<CT>(N) * sizeof(T);
    if (ws && ws_bytes >= cols_bytes && cols_worthwhile(M, N, K)) {
        if (!cols_ready) {
            int rc = materialise<CT>(b.gather, ws, st, what);
            if (rc != VS_OK) return rc;
        }
        const int64_t ld = cols_pitch<CT>(N);
        Dense<CT, LS> bd{(const T*)ws, ld, N, K, 1};               // element (pixel n, k = q) at cols[q * ld + n]
        if constexpr (CT != VS_F32) {
            // many 128-wide pixel tiles, >= 96 output channels: the 128x128 LDS-DMA ring tile (weights R, column matrix S, NCHW epilogue
            // with 8-byte / 16-byte stores along the pixels)
            static const int mid_mode = getenv("VS_CONV_FWD_MID") ? atoi(getenv("VS_CONV_FWD_MID")) : 1;
            const int64_t kt = vs_cdiv(K, BIG_BK), tiles = vs_cdiv((int64_t)M, 128) * vs_cdiv(N, 128);
            if (mid_mode && e.g_hw == 0 && K % 8 == 0 && N % 8 == 0 && (uintptr_t)wd % 16 == 0 && K < (1ll << 23) && ld < (1ll << 23) &&
                kt >= 8 && tiles >= 192 && (M >= 96 || mid_mode == 2)) {
                int rc2 = mid_launch<CT, LR, LS, true>(wd, K, ws, ld, M, N, K, 1, kt, 5, 1, e, nullptr, st, 0);
                if (rc2 != VS_OK) return rc2;
                VS_CHECK_LAUNCH(what);
                return VS_OK;
            }
        }
        return run<CT>(a, bd, M, N, K, e, (char*)ws + cols_bytes, ws_bytes - cols_bytes, st, what);
    }
    return run<CT>(a, b, M, N, K, e, ws, ws_bytes, st, what);     // implicit gather; the workspace (if any) serves split-K
}

// ---- ConvTranspose2d k4 s2 p1 with 1..4 output channels (the image-producing last decoder layer), all four parity phases in ONE
// pass: a thread owns (sample, source row y, 8 source pixels) and walks the input channels; per channel it fetches the rows
// y-1, y, y+1 once (left neighbour, 16-byte unit, right neighbour) and accumulates the 2 x 2 output parities of its 8 pixels,
// i.e. two output rows of 16 consecutive pixels.  The four per-phase launches of the row-dot kernel read the input four times
// (0.7 TB/s effective); here it is read once and the kernel is bound by that read.
// Packed weights (vs_conv_pack_weight, stride 2): Wp[phase = 2 py + px][m][c][j], j = the phase's taps in (ky, kx) ascending order.
template <int CT, int MR>
__global__ __launch_bounds__(256) void convt_k4s2_small_kernel(const typename CTraits<CT>::T* x, const typename CTraits<CT>::T* wp, const float* bias,
                                                               void* out, int od, int B, int C, int H, int W, int M) {
    typedef typename CTraits<CT>::T T;
    const int upr = W / 8;                                              // 8-pixel units per source row
    const int64_t units = (int64_t)B * H * upr;
    // phase taps: py = 0 -> (ky, dy) = (1, 0), (3, -1);  py = 1 -> (0, 1), (2, 0)   (same for x); j = 2 * jy + jx
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < units; u += (int64_t)gridDim.x * 256) {
        const int ux = (int)(u % upr);
        const int64_t by = u / upr;
        const int y = (int)(by % H), b = (int)(by / H), x0 = ux * 8;
        float acc[MR][2][2][8];
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int px = 0; px < 2; ++px)
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[m][py][px][i] = 0.f;
        for (int c = 0; c < C; ++c) {
            const T* plane = x + ((int64_t)b * C + c) * H * W;
            float xs[3][10];                                            // rows y-1, y, y+1; columns x0-1 .. x0+8
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int iy = y + d - 1;
                const bool ok = iy >= 0 && iy < H;
                const T* row = plane + (int64_t)(ok ? iy : y) * W + x0;
                T v[8];
                if constexpr (CT != VS_F32) {
                    *reinterpret_cast<u32x4*>(v) = *reinterpret_cast<const u32x4*>(row);
                } else {
                    *reinterpret_cast<u32x4*>(v) = *reinterpret_cast<const u32x4*>(row);
                    *reinterpret_cast<u32x4*>(v + 4) = *reinterpret_cast<const u32x4*>(row + 4);
                }
                const T lv = row[x0 > 0 ? -1 : 0], rv = row[x0 + 8 < W ? 8 : 7];
                xs[d][0] = (ok && x0 > 0) ? (float)lv : 0.f;
                xs[d][9] = (ok && x0 + 8 < W) ? (float)rv : 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) xs[d][1 + i] = ok ? (float)v[i] : 0.f;
            }
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                if (m >= M) break;
#pragma unroll
                for (int py = 0; py < 2; ++py)
#pragma unroll
                    for (int px = 0; px < 2; ++px) {
                        const T* wq = wp + (((int64_t)(2 * py + px) * M + m) * C + c) * 4;
#pragma unroll
                        for (int jy = 0; jy < 2; ++jy)
#pragma unroll
                            for (int jx = 0; jx < 2; ++jx) {
                                const float wv = (float)wq[2 * jy + jx];
                                const int dy = py == 0 ? (jy == 0 ? 0 : -1) : (jy == 0 ? 1 : 0);
                                const int dx = px == 0 ? (jx == 0 ? 0 : -1) : (jx == 0 ? 1 : 0);
#pragma unroll
                                for (int i = 0; i < 8; ++i) acc[m][py][px][i] += wv * xs[dy + 1][i + dx + 1];
                            }
                    }
            }
        }
        const int OW = 2 * W;
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            if (m >= M) break;
            const float bv = bias ? bias[m] : 0.f;
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const int64_t o = (((int64_t)b * M + m) * (2 * H) + 2 * y + py) * OW + 2 * x0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    vs_st(out, od, o + 2 * i, acc[m][py][0][i] + bv);
                    vs_st(out, od, o + 2 * i + 1, acc[m][py][1][i] + bv);
                }
            }
        }
    }
}

inline void natural_taps(TapGeo& g, int kh, int kw, int pad, bool flipped) {
    g.ntap = kh * kw;
    for (int ky = 0; ky < kh; ++ky)
        for (int kx = 0; kx < kw; ++kx) {
            g.dy[ky * kw + kx] = (signed char)(flipped ? pad - ky : ky - pad);
            g.dx[ky * kw + kx] = (signed char)(flipped ? pad - kx : kx - pad);
        }
}

// plain (strided) convolution form: Conv2d forward and ConvTranspose2d input gradient; weights in reference layout
template <int CT>
int conv_form(const void* src, const void* w, const float* bias, void* out, int out_dtype, int B, int Csrc, int H, int W, int M, int kh, int kw,
              int s, int p, int OH, int OW, void* ws, size_t ws_bytes, hipStream_t st, const char* what, bool cols_ready = false) {
    TapGeo g{B, Csrc, H, W, OH, OW, s, 0, {}, {}, 0, 0, 0, 0, 0};
    natural_taps(g, kh, kw, p, false);
    return gather_gemm<CT>(src, w, bias, out, out_dtype, M, g, OH, OW, 1, 0, 0, ws, ws_bytes, st, what, cols_ready);
}

// out[plane, gy * s + oy, gx * s + ox] = 0 over a phase grid (a phase of a strided input gradient that no tap reaches)
__global__ __launch_bounds__(256) void zero_phase_kernel(void* out, int od, int64_t planes, int GH, int GW, int OH, int OW, int s, int oy, int ox) {
    const int64_t total = planes * GH * GW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int gx = (int)(i % GW), gy = (int)((i / GW) % GH);
        const int64_t plane = i / ((int64_t)GW * GH);
        vs_st(out, od, (plane * OH + gy * s + oy) * OW + gx * s + ox, 0.f);
    }
}

// transposed form: ConvTranspose2d forward and Conv2d input gradient; weights PACKED by vs_conv_pack_weight
//   stride 1      : Wp[M][Csrc][kh*kw], taps flipped (offset p - k)
//   stride 2 phase: Wp[phase][M][Csrc][ntap_phase]
template <int CT>
int transposed_form(const void* src, const void* wp, const float* bias, void* out, int out_dtype, int B, int Csrc, int H, int W, int M, int kh,
                    int kw, int s, int p, int OH, int OW, void* ws, size_t ws_bytes, hipStream_t st, const char* what) {
    typedef typename CTraits<CT>::T T;
    if (s == 1) {
        TapGeo g{B, Csrc, H, W, OH, OW, 1, 0, {}, {}, 0, 0, 0, 0, 0};
        natural_taps(g, kh, kw, p, true);
        return gather_gemm<CT>(src, wp, bias, out, out_dtype, M, g, OH, OW, 1, 0, 0, ws, ws_bytes, st, what);
    }
    if (!phase_ok(kh, kw, s, p)) return vs_fail(VS_ERR_UNSUPPORTED, "%s: transposed geometry k%d s%d p%d is not supported", what, kh, s, p);
    if (kh == 4 && kw == 4 && p == 1 && M <= 4 && W % 8 == 0 && OH == 2 * H && OW == 2 * W && (uintptr_t)src % 16 == 0 &&
        getenv("VS_CONVT_SMALL") == nullptr) {
        const int64_t units = (int64_t)B * H * (W / 8);
        int64_t blocks = (units + 255) / 256;
        if (blocks > 16384) blocks = 16384;
        if (M <= 2)
            hipLaunchKernelGGL((convt_k4s2_small_kernel<CT, 2>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)src, (const T*)wp, bias, out,
                               out_dtype, B, Csrc, H, W, M);
        else            // 3-channel frames (chairs): 128 accumulators per thread, still one pass over the input
            hipLaunchKernelGGL((convt_k4s2_small_kernel<CT, 4>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)src, (const T*)wp, bias, out, out_dtype,
                           B, Csrc, H, W, M);
        VS_CHECK_LAUNCH(what);
        return VS_OK;
    }
    // Output pixel (2 gy + py, 2 gx + px) of phase (py, px) reads source pixels (gy + dy_t, gx + dx_t): the phase grid is the set
    // of output pixels of that parity, ceil((OH - py) / 2) x ceil((OW - px) / 2) -- the whole source grid when out = 2 x in
    // (k4 s2 p1), smaller and phase-dependent for the input gradients of k3 s2 p1 / k1 s2 p0 convolutions on odd sizes
    // (conv.py:440, 531 of the reference's ResNet18).  A phase without taps (k1: only even pixels receive anything) is zeros.
    const T* wph = (const T*)wp;
    for (int py = 0; py < s; ++py)
        for (int px = 0; px < s; ++px) {
            const int GH = (OH - py + s - 1) / s, GW = (OW - px + s - 1) / s;
            TapGeo g{B, Csrc, H, W, GH, GW, 1, 0, {}, {}, 0, 0, 0, 0, 0};
            int kidx[MAXTAP];
            g.ntap = phase_taps(kh, kw, s, p, py, px, kidx, g.dy, g.dx);
            if (GH <= 0 || GW <= 0) { wph += (int64_t)M * Csrc * g.ntap; continue; }
            if (g.ntap == 0) {
                if (bias) return vs_fail(VS_ERR_UNSUPPORTED, "%s: a tap-free phase with a bias is not supported (k%d s%d p%d)", what, kh, s, p);
                const int64_t total = (int64_t)B * M * GH * GW;
                int64_t blocks = (total + 255) / 256;
                if (blocks > 4096) blocks = 4096;
                hipLaunchKernelGGL(zero_phase_kernel, dim3((unsigned)blocks), dim3(256), 0, st, out, out_dtype, (int64_t)B * M, GH, GW, OH, OW, s, py, px);
                VS_CHECK_LAUNCH(what);
                continue;
            }
            int rc = gather_gemm<CT>(src, wph, bias, out, out_dtype, M, g, OH, OW, s, py, px, ws, ws_bytes, st, what);
            if (rc != VS_OK) return rc;
            wph += (int64_t)M * Csrc * g.ntap;
        }
    return VS_OK;
}

// does wgrad_form(M, N = channels * taps, K = pixels) materialise its column matrix into this workspace?  (also asked by the
// ConvTranspose2d input gradient, which can reuse that very matrix)
template <int CT>
inline bool wgrad_uses_cols(int64_t M, int64_t N, int64_t K, const void* ws, size_t ws_bytes) {
    const size_t cols_bytes = (size_t)N * cols_pitch<CT>(K) * sizeof(typename CTraits<CT>::T);
    return ws && ws_bytes >= cols_bytes + vs_gemm_workspace_bytes(M, N, K) && cols_worthwhile(M, K, N);
}

// R [B][M][hw] (16-bit) -> dense rows [M][B hw]: the channel-rows operand of a weight gradient on maps of fewer than 32 pixels (4 x 4), where a
// 32-wide K tile of the LDS-DMA ring kernel would straddle two maps; one 16-byte piece per thread and step
__global__ __launch_bounds__(256) void chan_rows_to_dense_kernel(const unsigned short* __restrict__ r, unsigned short* __restrict__ d, int M, int hw8, int64_t K8,
                                                                 int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / K8, k8 = i - m * K8, b = k8 / hw8, p8 = k8 - b * hw8;
        *reinterpret_cast<u32x4*>(d + i * 8) = *reinterpret_cast<const u32x4*>(r + ((b * M + m) * hw8 + p8) * 8);
    }
}

// weight gradient: dW[m][(c,t)] = sum_pix R[b,m,pix] * G[b,c,py*s-p+ky,px*s-p+kx]; R has Cr channels on the (PH,PW) pixel
// grid, G has Cg channels of size (GH,GW)
template <int CT>
int wgrad_form(const void* r, const void* gsrc, float* dw, int B, int Cr, int PH, int PW, int Cg, int GH, int GW, int kh, int kw, int s, int p,
               void* ws, size_t ws_bytes, hipStream_t st, const char* what, int accumulate = 0) {
    typedef typename CTraits<CT>::T T;
    const int64_t M = Cr, N = (int64_t)Cg * kh * kw, K = (int64_t)B * PH * PW;
    const int64_t hw = (int64_t)PH * PW;
    int hw_shift = -1;
    for (int k = 0; k < 31; ++k) if (((int64_t)1 << k) == hw) hw_shift = k;
    ChanRows<CT> a{(const T*)r, Cr, hw, K, ((uintptr_t)r % 16 == 0) && (hw % CTraits<CT>::U == 0), hw_shift};
    GatherOp<CT, false> b;
    TapGeo g{B, Cg, GH, GW, PH, PW, s, 0, {}, {}, 0, 0, 0, 0, 0};
    natural_taps(g, kh, kw, p, false);
    b.gather.src = (const T*)gsrc; b.gather.g = g; b.gather.g.finish(); b.gather.npix = K; b.gather.nq = N;
    const size_t cols_bytes = (size_t)N * cols_pitch<CT>(K) * sizeof(T);
    if (wgrad_uses_cols<CT>(M, N, K, ws, ws_bytes)) {
        int rc = materialise<CT>(b.gather, ws, st, what);
        if (rc != VS_OK) return rc;
        const int64_t ld = cols_pitch<CT>(K);
        Dense<CT, LR> bd{(const T*)ws, ld, N, K, 1};               // element (q, k = pixel) at cols[q * ld + k]
        Epi e = rowmajor_epi(dw, N);
        e.accumulate = accumulate;
        if constexpr (CT != VS_F32) {
            // the 128x128 LDS-DMA ring tile (vs_gemm_mid.h) with dy as a channel-rows operand: planes of a multiple of 32 pixels (a K
            // tile never straddles two images), split-K over the batch x pixel axis into slabs; 2x the register-staged tile on the
            // DCGAN / VGG weight gradients (K = 10^4 .. 10^5, 16-64 output tiles)
            static const int mid_mode = getenv("VS_CONV_WGRAD_MID") ? atoi(getenv("VS_CONV_WGRAD_MID")) : 1;
            const int64_t kt = K / BIG_BK, tiles = vs_cdiv(M, 128) * vs_cdiv(N, 128);
            // maps of 8 / 16 pixels (hw not a multiple of the K tile): the same kernel on a dense [M][K] copy of R (one small transposing pass)
            const bool via_dense = hw % BIG_BK != 0 && hw % 8 == 0;
            const size_t dense_bytes = via_dense ? (size_t)M * (size_t)K * sizeof(T) : 0;
            if (mid_mode && (hw % BIG_BK == 0 || via_dense) && K % BIG_BK == 0 && M % 8 == 0 && N % 8 == 0 && (uintptr_t)r % 16 == 0 && hw < (1ll << 23) &&
                ld < (1ll << 23) && K < (1ll << 23) && kt >= 32 && tiles <= 512 && (M >= 64 || mid_mode == 2)) {
                int splits = 1;
                if (tiles < 448) {
                    splits = (int)(448 / tiles);
                    const int64_t max_by_k = kt / 16;
                    if (splits > max_by_k) splits = (int)max_by_k;
                    if (splits > 64) splits = 64;
                    if (splits < 1) splits = 1;
                }
                const int64_t ktps = vs_cdiv(kt, splits);
                splits = (int)vs_cdiv(kt, ktps);
                float* slabs = nullptr;
                const size_t need = splits > 1 ? (size_t)splits * (size_t)M * (size_t)N * sizeof(float) : 0;
                if (need + dense_bytes <= ws_bytes - cols_bytes) {
                    if (splits > 1) slabs = (float*)((char*)ws + cols_bytes);
                    const int stages = tiles * splits <= 256 ? 10 : 5;
                    const void* a_ptr = r;
                    if (via_dense) {
                        unsigned short* dense = (unsigned short*)((char*)ws + cols_bytes + need);
                        const int64_t total = M * K / 8;
                        int64_t blocks = vs_cdiv(total, 256);
                        if (blocks > 4096) blocks = 4096;
                        hipLaunchKernelGGL(chan_rows_to_dense_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const unsigned short*)r, dense, (int)M,
                                           (int)(hw / 8), K / 8, total);
                        VS_CHECK_LAUNCH(what);
                        a_ptr = dense;
                    }
                    int rc2 = mid_launch<CT, LR, LR>(a_ptr, via_dense ? K : hw, ws, ld, M, N, K, splits, ktps, stages, 1, e, slabs, st, via_dense ? 0 : hw);
                    if (rc2 != VS_OK) return rc2;
                    VS_CHECK_LAUNCH(what);
                    if (slabs) {
                        int64_t blocks = vs_cdiv(M * N, 256);
                        if (blocks > 2048) blocks = 2048;
                        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, slabs, splits, M, N, e);
                        VS_CHECK_LAUNCH(what);
                    }
                    return VS_OK;
                }
            }
        }
        return run<CT>(a, bd, M, N, K, e, (char*)ws + cols_bytes, ws_bytes - cols_bytes, st, what);
    }
    Epi e = rowmajor_epi(dw, N);
    e.accumulate = accumulate;
    return run<CT>(a, b, M, N, K, e, ws, ws_bytes, st, what);
}

// ---- weight pre-pack for the transposed form ---------------------------------------------------------------------------
// src fp32 [D0][D1][kh*kw] (Conv2d: D0 = Cout, D1 = Cin; ConvTranspose2d: D0 = Cin, D1 = Cout).  The transposed form has
// its GEMM rows on D1 and reduces over D0.  stride 1: dst[m][c][t] = src[c][m][t];  stride 2: for each phase,
// dst[phase][m][c][j] = src[c][m][kidx_phase[j]].
struct PackTable { int nphase; int ntap[4]; int kidx[4][MAXTAP]; };

template <int CT>
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* src, typename CTraits<CT>::T* dst, int D0, int D1, int khw, PackTable tb) {
    int64_t base = 0;
    for (int ph = 0; ph < tb.nphase; ++ph) {
        const int nt = tb.ntap[ph];
        const int64_t count = (int64_t)D1 * D0 * nt;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
            const int j = (int)(i % nt);
            const int64_t mc = i / nt;
            const int c = (int)(mc % D0), m = (int)(mc / D0);
            dst[base + i] = (typename CTraits<CT>::T)src[((int64_t)c * D1 + m) * khw + tb.kidx[ph][j]];
        }
        base += count;
    }
}

#define VS_DISPATCH(compute, fn, ...) \
    ((compute) == VS_BF16 ? fn<VS_BF16>(__VA_ARGS__) : (compute) == VS_F16 ? fn<VS_F16>(__VA_ARGS__) : fn<VS_F32>(__VA_ARGS__))

}  // namespace

extern "C" size_t vs_conv_packed_elems(int D0, int D1, int kh, int kw, int stride, int pad) {
    // stride 1: D0*D1*kh*kw; stride 2 phases: the phases partition the taps, so the total is the same
    (void)stride; (void)pad;
    return (size_t)D0 * D1 * kh * kw;
}

extern "C" int vs_conv_pack_weight(int compute, const float* w, int D0, int D1, int kh, int kw, int stride, int pad, void* dst, void* stream) {
    VS_CHECK_ARG(vs_dtype_ok(compute), "vs_conv_pack_weight: compute type %d", compute);
    VS_CHECK_ARG(w && dst && D0 > 0 && D1 > 0 && kh * kw <= MAXTAP && kh > 0 && kw > 0, "vs_conv_pack_weight: bad argument");
    PackTable tb;
    if (stride == 1) {
        tb.nphase = 1; tb.ntap[0] = kh * kw;
        for (int t = 0; t < kh * kw; ++t) tb.kidx[0][t] = t;
    } else {
        if (!phase_ok(kh, kw, stride, pad)) return vs_fail(VS_ERR_UNSUPPORTED, "vs_conv_pack_weight: k%d s%d p%d is not supported", kh, stride, pad);
        tb.nphase = 4;
        signed char dy[MAXTAP], dx[MAXTAP];
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) tb.ntap[py * 2 + px] = phase_taps(kh, kw, stride, pad, py, px, tb.kidx[py * 2 + px], dy, dx);
    }
    int64_t total = (int64_t)D0 * D1 * kh * kw;
    int64_t blocks = (total / 4 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    if (compute == VS_BF16)
        hipLaunchKernelGGL(pack_weight_kernel<VS_BF16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, (__bf16*)dst, D0, D1, kh * kw, tb);
    else if (compute == VS_F16)
        hipLaunchKernelGGL(pack_weight_kernel<VS_F16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, (_Float16*)dst, D0, D1, kh * kw, tb);
    else
        hipLaunchKernelGGL(pack_weight_kernel<VS_F32>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, (float*)dst, D0, D1, kh * kw, tb);
    VS_CHECK_LAUNCH("vs_conv_pack_weight");
    return VS_OK;
}

extern "C" size_t vs_conv_wgrad_workspace_bytes(int B, int Cin, int OH, int OW, int Cout, int kh, int kw) {
    // upper bound over both weight-gradient orientations (conv: M=Cout,N=Cin*khw ; convT: M=Cin,N=Cout*khw)
    const int64_t K = (int64_t)B * OH * OW;
    size_t a = vs_gemm_workspace_bytes(Cout, (int64_t)Cin * kh * kw, K);
    size_t b = vs_gemm_workspace_bytes(Cin, (int64_t)Cout * kh * kw, K);
    return a > b ? a : b;
}

extern "C" size_t vs_conv_workspace_bytes(int compute, int B, int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad) {
    // covers forward, input gradient and weight gradient of Conv2d(Cin, Cout) on [B, Cin, H, W] AND of the ConvTranspose2d with the
    // same numbers: the largest column matrix any of them materialises + the split-K slabs of the contraction behind it
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || stride <= 0) return 0;
    const int64_t e = vs_esize(compute), U = 16 / e;
    const int64_t OHc = (H + 2 * pad - kh) / stride + 1, OWc = (W + 2 * pad - kw) / stride + 1;       // Conv2d output
    const int64_t OHt = (int64_t)(H - 1) * stride - 2 * pad + kh, OWt = (int64_t)(W - 1) * stride - 2 * pad + kw;   // ConvTranspose2d output
    const int64_t khw = (int64_t)kh * kw;
    const int64_t tt = (stride == 2 && phase_ok(kh, kw, stride, pad)) ? (int64_t)((kh + 1) / 2) * ((kw + 1) / 2) : khw;   // most taps of a phase
    auto cols = [&](int64_t npix, int64_t nq) { return (size_t)(nq * ((npix + U - 1) / U * U) * e); };
    size_t worst = 0;
    auto take = [&](size_t c, int64_t M, int64_t N, int64_t K) { const size_t t = c + vs_gemm_workspace_bytes(M, N, K); if (t > worst) worst = t; };
    if (OHc > 0 && OWc > 0) {
        take(cols(B * OHc * OWc, Cin * khw), Cout, B * OHc * OWc, Cin * khw);                         // conv forward
        if (stride == 2) take(cols(B * OHc * OWc, Cout * tt), Cin, B * OHc * OWc, Cout * tt);        // conv dgrad: per phase on dy's grid
        else take(cols((int64_t)B * H * W, Cout * khw), Cin, (int64_t)B * H * W, Cout * khw);         //   stride 1: on dx's grid
        take(cols(B * OHc * OWc, Cin * khw) + (size_t)(Cout * B * OHc * OWc * e), Cout, Cin * khw, B * OHc * OWc);   // conv wgrad (+ a dense copy of dy)
    }
    if (OHt > 0 && OWt > 0) {
        if (stride == 2) take(cols((int64_t)B * H * W, Cin * tt), Cout, (int64_t)B * H * W, Cin * tt); // convT forward: per phase on x's grid
        else take(cols(B * OHt * OWt, Cin * khw), Cout, B * OHt * OWt, Cin * khw);                     //   stride 1: on the output grid
        take(cols((int64_t)B * H * W, Cout * khw), Cin, (int64_t)B * H * W, Cout * khw);              // convT dgrad
        take(cols((int64_t)B * H * W, Cout * khw) + (size_t)((int64_t)Cin * B * H * W * e), Cin, Cout * khw, (int64_t)B * H * W);   // convT wgrad (+ a dense copy of x)
    }
    return worst;
}

extern "C" int vs_conv2d_fwd(int compute, const void* x, const void* w, const float* bias, void* y, int y_dtype, int B, int Cin, int H,
                             int W, int Cout, int kh, int kw, int stride, int pad, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_conv("vs_conv2d_fwd", compute, x, w, y, B, Cin, H, W, Cout, kh, kw, stride, pad);
    if (rc) return rc;
    const int OH = (H + 2 * pad - kh) / stride + 1, OW = (W + 2 * pad - kw) / stride + 1;
    VS_CHECK_ARG(OH > 0 && OW > 0, "vs_conv2d_fwd: empty output");
    return VS_DISPATCH(compute, conv_form, x, w, bias, y, y_dtype, B, Cin, H, W, Cout, kh, kw, stride, pad, OH, OW, workspace, workspace_bytes,
                       (hipStream_t)stream, "vs_conv2d_fwd");
}

extern "C" int vs_conv2d_dgrad(int compute, const void* dy, const void* w_packed, void* dx, int dx_dtype, int B, int Cin, int H, int W,
                               int Cout, int kh, int kw, int stride, int pad, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_conv("vs_conv2d_dgrad", compute, dy, w_packed, dx, B, Cin, H, W, Cout, kh, kw, stride, pad);
    if (rc) return rc;
    const int OH = (H + 2 * pad - kh) / stride + 1, OW = (W + 2 * pad - kw) / stride + 1;
    // dx[b,ci,y,x] = sum dy[b,co,(y+p-ky)/s,(x+p-kx)/s] W[co][ci][ky,kx]: transposed form gathering from dy
    return VS_DISPATCH(compute, transposed_form, dy, w_packed, nullptr, dx, dx_dtype, B, Cout, OH, OW, Cin, kh, kw, stride, pad, H, W,
                       workspace, workspace_bytes, (hipStream_t)stream, "vs_conv2d_dgrad");
}

extern "C" int vs_conv2d_wgrad(int compute, const void* dy, const void* x, float* dw, int B, int Cin, int H, int W, int Cout, int kh, int kw,
                               int stride, int pad, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_conv("vs_conv2d_wgrad", compute, dy, x, dw, B, Cin, H, W, Cout, kh, kw, stride, pad);
    if (rc) return rc;
    const int OH = (H + 2 * pad - kh) / stride + 1, OW = (W + 2 * pad - kw) / stride + 1;
    return VS_DISPATCH(compute, wgrad_form, dy, x, dw, B, Cout, OH, OW, Cin, H, W, kh, kw, stride, pad, workspace, workspace_bytes,
                       (hipStream_t)stream, "vs_conv2d_wgrad");
}

// dw += ... instead of dw = ...: a module applied many times per step (the SST integrator's convolutions: 39 calls) adds each call's
// weight gradient straight into the accumulated one (GEMM / split-K reduce epilogue) instead of storing it and adding it with a
// second launch
extern "C" int vs_conv2d_wgrad_acc(int compute, const void* dy, const void* x, float* dw, int B, int Cin, int H, int W, int Cout, int kh,
                                   int kw, int stride, int pad, void* workspace, size_t workspace_bytes, int accumulate, void* stream) {
    int rc = check_conv("vs_conv2d_wgrad_acc", compute, dy, x, dw, B, Cin, H, W, Cout, kh, kw, stride, pad);
    if (rc) return rc;
    const int OH = (H + 2 * pad - kh) / stride + 1, OW = (W + 2 * pad - kw) / stride + 1;
    return VS_DISPATCH(compute, wgrad_form, dy, x, dw, B, Cout, OH, OW, Cin, H, W, kh, kw, stride, pad, workspace, workspace_bytes,
                       (hipStream_t)stream, "vs_conv2d_wgrad_acc", accumulate ? 1 : 0);
}

extern "C" int vs_conv_transpose2d_wgrad_acc(int compute, const void* dy, const void* x, float* dw, int B, int Cin, int H, int W, int Cout,
                                             int kh, int kw, int stride, int pad, void* workspace, size_t workspace_bytes, int accumulate,
                                             void* stream) {
    int rc = check_conv("vs_conv_transpose2d_wgrad_acc", compute, dy, x, dw, B, Cin, H, W, Cout, kh, kw, stride, pad);
    if (rc) return rc;
    const int OH = (H - 1) * stride - 2 * pad + kh, OW = (W - 1) * stride - 2 * pad + kw;
    return VS_DISPATCH(compute, wgrad_form, x, dy, dw, B, Cin, H, W, Cout, OH, OW, kh, kw, stride, pad, workspace, workspace_bytes,
                       (hipStream_t)stream, "vs_conv_transpose2d_wgrad_acc", accumulate ? 1 : 0);
}

extern "C" int vs_conv_transpose2d_fwd(int compute, const void* x, const void* w_packed, const float* bias, void* y, int y_dtype, int B,
                                       int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    int rc = check_conv("vs_conv_transpose2d_fwd", compute, x, w_packed, y, B, Cin, H, W, Cout, kh, kw, stride, pad);
    if (rc) return rc;
    const int OH = (H - 1) * stride - 2 * pad + kh, OW = (W - 1) * stride - 2 * pad + kw;
    VS_CHECK_ARG(OH > 0 && OW > 0, "vs_conv_transpose2d_fwd: empty output");
    return VS_DISPATCH(compute, transposed_form, x, w_packed, bias, y, y_dtype, B, Cin, H, W, Cout, kh, kw, stride, pad, OH, OW,
                       workspace, workspace_bytes, (hipStream_t)stream, "vs_conv_transpose2d_fwd");
}

extern "C" int vs_conv_transpose2d_dgrad(int compute, const void* dy, const void* w, void* dx, int dx_dtype, int B, int Cin, int H, int W,
                                         int Cout, int kh, int kw, int stride, int pad, void* workspace, size_t workspace_bytes,
                                         int cols_from_wgrad, void* stream) {
    int rc = check_conv("vs_conv_transpose2d_dgrad", compute, dy, w, dx, B, Cin, H, W, Cout, kh, kw, stride, pad);
    if (rc) return rc;
    const int OH = (H - 1) * stride - 2 * pad + kh, OW = (W - 1) * stride - 2 * pad + kw;
    // dx[b,ci,y,x] = sum dy[b,co,y*s-p+ky,x*s-p+kx] W[ci][(co,ky,kx)]: a plain convolution of dy with W read as dense [Cin, Cout*khw]
    // the weight gradient gathers dy through the same taps onto the same pixel grid: when it ran just before on this stream with
    // this workspace (and did materialise), its column matrix is still there
    bool ready = false;
    if (cols_from_wgrad) {
        const int64_t Mw = Cin, Nw = (int64_t)Cout * kh * kw, Kw = (int64_t)B * H * W;
        ready = compute != VS_F32 ? wgrad_uses_cols<VS_BF16>(Mw, Nw, Kw, workspace, workspace_bytes)      // sizes only: bf16 == fp16
                                  : wgrad_uses_cols<VS_F32>(Mw, Nw, Kw, workspace, workspace_bytes);
    }
    return VS_DISPATCH(compute, conv_form, dy, w, nullptr, dx, dx_dtype, B, Cout, OH, OW, Cin, kh, kw, stride, pad, H, W, workspace,
                       workspace_bytes, (hipStream_t)stream, "vs_conv_transpose2d_dgrad", ready);
}

extern "C" int vs_conv_transpose2d_wgrad(int compute, const void* dy, const void* x, float* dw, int B, int Cin, int H, int W, int Cout,
                                         int kh, int kw, int stride, int pad, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_conv("vs_conv_transpose2d_wgrad", compute, dy, x, dw, B, Cin, H, W, Cout, kh, kw, stride, pad);
    if (rc) return rc;
    const int OH = (H - 1) * stride - 2 * pad + kh, OW = (W - 1) * stride - 2 * pad + kw;
    // dW[ci][(co,ky,kx)] = sum_pix x[b,ci,pix] dy[b,co,y*s-p+ky,x*s-p+kx]
    return VS_DISPATCH(compute, wgrad_form, x, dy, dw, B, Cin, H, W, Cout, OH, OW, kh, kw, stride, pad, workspace, workspace_bytes,
                       (hipStream_t)stream, "vs_conv_transpose2d_wgrad");
}
