// vs_conv.hip -- Conv2d / ConvTranspose2d (forward, input gradient, weight gradient) as im2col-free implicit GEMMs
// on the shared MFMA contraction kernel (vs_gemm_core.h).  Activations stay NCHW in HBM; no im2col buffer, no
// layout change, no weight re-packing: the operand loaders below gather straight from the NCHW tensors / the
// [Cout,Cin,kh,kw] (Conv2d) or [Cin,Cout,kh,kw] (ConvTranspose2d) weights into LDS tiles.
//
// Orientation: output CHANNELS are the GEMM rows, output PIXELS (b, y, x) the GEMM columns, so that in the 32x32 MFMA
// accumulator (column on the lane) consecutive lanes hold consecutive pixels of one channel: NCHW stores are
// contiguous 128-byte runs per register, and the per-channel bias is a row constant.
//
//   op                    M        N            K            A (rows m, reduction k)        B (rows n, reduction k)
//   conv   forward        Cout     B*OH*OW      Cin*kh*kw    W[m][k]              (dense)   x[b,c,oy*s-p+ky,ox*s-p+kx]   (im2col)
//   conv   dgrad          Cin      B*H*W        Cout*kh*kw   W[co][m][ky,kx]      (WeightT) dy[b,co,(y+p-ky)/s,(x+p-kx)/s] (col2im)
//   conv   wgrad          Cout     Cin*kh*kw    B*OH*OW      dy[b,m,pix]          (ChanRows) x gathered as im2col, pixel = reduction
//   convT  forward        Cout     B*OH*OW      Cin*kh*kw    W[ci][m][ky,kx]      (WeightT) x[b,ci,(oy+p-ky)/s,(ox+p-kx)/s] (col2im)
//   convT  dgrad          Cin      B*H*W        Cout*kh*kw   W[m][k]              (dense)   dy[b,co,y*s-p+ky,x*s-p+kx]   (im2col)
//   convT  wgrad          Cin      Cout*kh*kw   B*H*W        x[b,m,pix]           (ChanRows) dy gathered as im2col, pixel = reduction
//
// Reference call sites: every nn.Conv2d / nn.ConvTranspose2d of networks/conv.py:119-122,147-170,258-263,294-318,
// 326-343,362-382,402-417 and networks/resnet.py:57-59, plus their autograd.
#include "vs_gemm_core.h"

namespace {

struct Geo {
    int B, C, H, W;       // the NCHW tensor the loader gathers from
    int kh, kw, s, p;
    int OH, OW;           // the pixel grid enumerated by the GEMM index (conv: output grid; col2im: the larger grid)
};

template <int CT>
struct GatherBase {
    typedef typename CTraits<CT>::T T;
    static constexpr int U = CTraits<CT>::U;
    const T* src; Geo g; int64_t npix, nq;
};

// src[b, c, py*s - p + ky, px*s - p + kx] for U consecutive pixels starting at pix0 and one q = (c, ky, kx)
template <int CT>
struct Im2col : GatherBase<CT> {
    typedef typename CTraits<CT>::T T;
    static constexpr int U = CTraits<CT>::U;
    __device__ __forceinline__ u32x4 unit(int64_t pix0, int64_t q) const {
        T tmp[U];
#pragma unroll
        for (int j = 0; j < U; ++j) tmp[j] = (T)0.f;
        const Geo& g = this->g;
        if (q < this->nq && pix0 < this->npix) {
            const int khw = g.kh * g.kw;
            const int c = (int)(q / khw), kk = (int)(q % khw);
            const int ky = kk / g.kw, kx = kk % g.kw;
            const int ohw = g.OH * g.OW;
            int b = (int)(pix0 / ohw);
            const int rem = (int)(pix0 % ohw);
            int py = rem / g.OW, px = rem % g.OW;
#pragma unroll
            for (int j = 0; j < U; ++j) {
                if (pix0 + j < this->npix) {
                    const int iy = py * g.s - g.p + ky, ix = px * g.s - g.p + kx;
                    if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
                        tmp[j] = this->src[(((int64_t)b * g.C + c) * g.H + iy) * g.W + ix];
                }
                if (++px == g.OW) { px = 0; if (++py == g.OH) { py = 0; ++b; } }
            }
        }
        return *reinterpret_cast<u32x4*>(tmp);
    }
};

// src[b, c, (py + p - ky)/s, (px + p - kx)/s] where divisible and in range (transposed-convolution gather)
template <int CT>
struct Col2im : GatherBase<CT> {
    typedef typename CTraits<CT>::T T;
    static constexpr int U = CTraits<CT>::U;
    __device__ __forceinline__ u32x4 unit(int64_t pix0, int64_t q) const {
        T tmp[U];
#pragma unroll
        for (int j = 0; j < U; ++j) tmp[j] = (T)0.f;
        const Geo& g = this->g;
        if (q < this->nq && pix0 < this->npix) {
            const int khw = g.kh * g.kw;
            const int c = (int)(q / khw), kk = (int)(q % khw);
            const int ky = kk / g.kw, kx = kk % g.kw;
            const int ohw = g.OH * g.OW;
            int b = (int)(pix0 / ohw);
            const int rem = (int)(pix0 % ohw);
            int py = rem / g.OW, px = rem % g.OW;
#pragma unroll
            for (int j = 0; j < U; ++j) {
                if (pix0 + j < this->npix) {
                    const int ty = py + g.p - ky, tx = px + g.p - kx;
                    if (ty >= 0 && tx >= 0 && (ty % g.s) == 0 && (tx % g.s) == 0) {
                        const int iy = ty / g.s, ix = tx / g.s;
                        if (iy < g.H && ix < g.W) tmp[j] = this->src[(((int64_t)b * g.C + c) * g.H + iy) * g.W + ix];
                    }
                }
                if (++px == g.OW) { px = 0; if (++py == g.OH) { py = 0; ++b; } }
            }
        }
        return *reinterpret_cast<u32x4*>(tmp);
    }
};

// GEMM operand views of a gather: PIX_IS_ROW -> element(i = pixel, k = q), LDS layout S (unit along i);
//                                 otherwise   -> element(i = q, k = pixel), LDS layout R (unit along k).
template <int CT, class G, bool PIX_IS_ROW>
struct GatherOp {
    static constexpr int layout = PIX_IS_ROW ? LS : LR;
    G gather;
    __device__ __forceinline__ u32x4 load(int64_t i, int64_t k) const { return PIX_IS_ROW ? gather.unit(i, k) : gather.unit(k, i); }
};

// element(m, k = (o, kk)) = w[(o * Mtot + m) * khw + kk]  (weight with the GEMM row as its SECOND dimension)
template <int CT>
struct WeightT {
    typedef typename CTraits<CT>::T T;
    static constexpr int U = CTraits<CT>::U;
    static constexpr int layout = LR;
    const T* w; int64_t Mtot, K; int khw;
    __device__ __forceinline__ u32x4 load(int64_t m, int64_t k) const {
        T tmp[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int64_t kq = k + j;
            T v = (T)0.f;
            if (m < Mtot && kq < K) {
                const int64_t o = kq / khw;
                v = w[(o * Mtot + m) * khw + (kq - o * khw)];
            }
            tmp[j] = v;
        }
        return *reinterpret_cast<u32x4*>(tmp);
    }
};

// element(m = channel, k = pixel (b, pix)) = src[(b * C + m) * HW + pix]
template <int CT>
struct ChanRows {
    typedef typename CTraits<CT>::T T;
    static constexpr int U = CTraits<CT>::U;
    static constexpr int layout = LR;
    const T* src; int64_t C, HW, K; int vec_ok;
    __device__ __forceinline__ u32x4 load(int64_t m, int64_t k) const {
        u32x4 z = {0u, 0u, 0u, 0u};
        if (m >= C || k >= K) return z;
        int64_t b = k / HW, r = k - b * HW;
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
    constexpr int BK = CT == VS_BF16 ? 64 : 16;
    Plan plan = make_plan(CT, M, N, K);
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

inline Epi nchw_epi(void* out, int out_dtype, const float* bias, int64_t hw, int64_t channels, int accumulate) {
    Epi e{out, 0, out_dtype, 1.f, bias, VS_ACT_NONE, nullptr, 0, 0, VS_ACT_NONE, accumulate, hw, channels};
    return e;
}
inline Epi rowmajor_epi(void* out, int64_t ldc) {
    Epi e{out, ldc, VS_F32, 1.f, nullptr, VS_ACT_NONE, nullptr, 0, 0, VS_ACT_NONE, 0, 0, 0};
    return e;
}

int check_conv(const char* what, int compute, const void* a, const void* b, const void* c, int B, int Cin, int H, int W, int Cout,
               int kh, int kw, int stride, int pad) {
    VS_CHECK_ARG(compute == VS_F32 || compute == VS_BF16, "%s: compute type %d", what, compute);
    VS_CHECK_ARG(a && b && c, "%s: null pointer", what);
    VS_CHECK_ARG(B > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0, "%s: bad geometry", what);
    return VS_OK;
}

// ---- the six contractions, templated on the compute type ------------------------------------------------------------
// conv-like forward: y[b,co,oy,ox] = bias + sum x[b,ci,oy*s-p+ky,ox*s-p+kx] Wd[co][(ci,ky,kx)]   (Wd dense [Cout, Cin*khw])
template <int CT>
int conv_like_fwd(const void* x, const void* wd, const float* bias, void* y, int y_dtype, int accumulate, int B, int Cin, int H, int W,
                  int Cout, int kh, int kw, int s, int p, int OH, int OW, hipStream_t st, const char* what) {
    typedef typename CTraits<CT>::T T;
    const int64_t M = Cout, N = (int64_t)B * OH * OW, K = (int64_t)Cin * kh * kw;
    Dense<CT, LR> a{(const T*)wd, K, M, K, ((uintptr_t)wd % 16 == 0) && (K % CTraits<CT>::U == 0)};
    GatherOp<CT, Im2col<CT>, true> b;
    b.gather.src = (const T*)x; b.gather.g = Geo{B, Cin, H, W, kh, kw, s, p, OH, OW}; b.gather.npix = N; b.gather.nq = K;
    return run<CT>(a, b, M, N, K, nchw_epi(y, y_dtype, bias, (int64_t)OH * OW, Cout, accumulate), nullptr, 0, st, what);
}

// transposed-conv-like forward: y[b,co,oy,ox] = bias + sum x[b,ci,(oy+p-ky)/s,(ox+p-kx)/s] Wt[ci][co][ky,kx]
template <int CT>
int convT_like_fwd(const void* x, const void* wt, const float* bias, void* y, int y_dtype, int accumulate, int B, int Cin, int H, int W,
                   int Cout, int kh, int kw, int s, int p, int OH, int OW, hipStream_t st, const char* what) {
    typedef typename CTraits<CT>::T T;
    const int64_t M = Cout, N = (int64_t)B * OH * OW, K = (int64_t)Cin * kh * kw;
    WeightT<CT> a{(const T*)wt, M, K, kh * kw};
    GatherOp<CT, Col2im<CT>, true> b;
    b.gather.src = (const T*)x; b.gather.g = Geo{B, Cin, H, W, kh, kw, s, p, OH, OW}; b.gather.npix = N; b.gather.nq = K;
    return run<CT>(a, b, M, N, K, nchw_epi(y, y_dtype, bias, (int64_t)OH * OW, Cout, accumulate), nullptr, 0, st, what);
}

// weight gradient: dW[m][(c,ky,kx)] = sum_pix R[b,m,pix] * G[b,c,py*s-p+ky,px*s-p+kx]; R has Cr channels on the (PH,PW) pixel
// grid, G has Cg channels of size (GH,GW)
template <int CT>
int wgrad_like(const void* r, const void* gsrc, float* dw, int B, int Cr, int PH, int PW, int Cg, int GH, int GW, int kh, int kw, int s,
               int p, void* ws, size_t ws_bytes, hipStream_t st, const char* what) {
    typedef typename CTraits<CT>::T T;
    const int64_t M = Cr, N = (int64_t)Cg * kh * kw, K = (int64_t)B * PH * PW;
    const int64_t hw = (int64_t)PH * PW;
    ChanRows<CT> a{(const T*)r, Cr, hw, K, ((uintptr_t)r % 16 == 0) && (hw % CTraits<CT>::U == 0)};
    GatherOp<CT, Im2col<CT>, false> b;
    b.gather.src = (const T*)gsrc; b.gather.g = Geo{B, Cg, GH, GW, kh, kw, s, p, PH, PW}; b.gather.npix = K; b.gather.nq = N;
    return run<CT>(a, b, M, N, K, rowmajor_epi(dw, N), ws, ws_bytes, st, what);
}

#define VS_DISPATCH(compute, fn, ...) ((compute) == VS_BF16 ? fn<VS_BF16>(__VA_ARGS__) : fn<VS_F32>(__VA_ARGS__))

}  // namespace

extern "C" size_t vs_conv_wgrad_workspace_bytes(int B, int Cin, int OH, int OW, int Cout, int kh, int kw) {
    // upper bound over both weight-gradient orientations (conv: M=Cout,N=Cin*khw ; convT: M=Cin,N=Cout*khw)
    const int64_t K = (int64_t)B * OH * OW;
    size_t a = vs_gemm_workspace_bytes(Cout, (int64_t)Cin * kh * kw, K);
    size_t b = vs_gemm_workspace_bytes(Cin, (int64_t)Cout * kh * kw, K);
    return a > b ? a : b;
}

extern "C" int vs_conv2d_fwd(int compute, const void* x, const void* w, const float* bias, void* y, int y_dtype, int B, int Cin, int H,
                             int W, int Cout, int kh, int kw, int stride, int pad, void* stream) {
    int rc = check_conv("vs_conv2d_fwd", compute, x, w, y, B, Cin, H, W, Cout, kh, kw, stride, pad);
    if (rc) return rc;
    const int OH = (H + 2 * pad - kh) / stride + 1, OW = (W + 2 * pad - kw) / stride + 1;
    VS_CHECK_ARG(OH > 0 && OW > 0, "vs_conv2d_fwd: empty output");
    return VS_DISPATCH(compute, conv_like_fwd, x, w, bias, y, y_dtype, 0, B, Cin, H, W, Cout, kh, kw, stride, pad, OH, OW,
                       (hipStream_t)stream, "vs_conv2d_fwd");
}

extern "C" int vs_conv2d_dgrad(int compute, const void* dy, const void* w, void* dx, int dx_dtype, int B, int Cin, int H, int W, int Cout,
                               int kh, int kw, int stride, int pad, void* stream) {
    int rc = check_conv("vs_conv2d_dgrad", compute, dy, w, dx, B, Cin, H, W, Cout, kh, kw, stride, pad);
    if (rc) return rc;
    const int OH = (H + 2 * pad - kh) / stride + 1, OW = (W + 2 * pad - kw) / stride + 1;
    // dx[b,ci,y,x] = sum dy[b,co,(y+p-ky)/s,(x+p-kx)/s] W[co][ci][ky,kx]: the transposed-conv gather with W read as [Cout][Cin][khw]
    return VS_DISPATCH(compute, convT_like_fwd, dy, w, nullptr, dx, dx_dtype, 0, B, Cout, OH, OW, Cin, kh, kw, stride, pad, H, W,
                       (hipStream_t)stream, "vs_conv2d_dgrad");
}

extern "C" int vs_conv2d_wgrad(int compute, const void* dy, const void* x, float* dw, int B, int Cin, int H, int W, int Cout, int kh, int kw,
                               int stride, int pad, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_conv("vs_conv2d_wgrad", compute, dy, x, dw, B, Cin, H, W, Cout, kh, kw, stride, pad);
    if (rc) return rc;
    const int OH = (H + 2 * pad - kh) / stride + 1, OW = (W + 2 * pad - kw) / stride + 1;
    return VS_DISPATCH(compute, wgrad_like, dy, x, dw, B, Cout, OH, OW, Cin, H, W, kh, kw, stride, pad, workspace, workspace_bytes,
                       (hipStream_t)stream, "vs_conv2d_wgrad");
}

extern "C" int vs_conv_transpose2d_fwd(int compute, const void* x, const void* w, const float* bias, void* y, int y_dtype, int B, int Cin,
                                       int H, int W, int Cout, int kh, int kw, int stride, int pad, void* stream) {
    int rc = check_conv("vs_conv_transpose2d_fwd", compute, x, w, y, B, Cin, H, W, Cout, kh, kw, stride, pad);
    if (rc) return rc;
    const int OH = (H - 1) * stride - 2 * pad + kh, OW = (W - 1) * stride - 2 * pad + kw;
    VS_CHECK_ARG(OH > 0 && OW > 0, "vs_conv_transpose2d_fwd: empty output");
    return VS_DISPATCH(compute, convT_like_fwd, x, w, bias, y, y_dtype, 0, B, Cin, H, W, Cout, kh, kw, stride, pad, OH, OW,
                       (hipStream_t)stream, "vs_conv_transpose2d_fwd");
}

extern "C" int vs_conv_transpose2d_dgrad(int compute, const void* dy, const void* w, void* dx, int dx_dtype, int B, int Cin, int H, int W,
                                         int Cout, int kh, int kw, int stride, int pad, void* stream) {
    int rc = check_conv("vs_conv_transpose2d_dgrad", compute, dy, w, dx, B, Cin, H, W, Cout, kh, kw, stride, pad);
    if (rc) return rc;
    const int OH = (H - 1) * stride - 2 * pad + kh, OW = (W - 1) * stride - 2 * pad + kw;
    // dx[b,ci,y,x] = sum dy[b,co,y*s-p+ky,x*s-p+kx] W[ci][(co,ky,kx)]: a plain convolution of dy with W read as dense [Cin, Cout*khw]
    return VS_DISPATCH(compute, conv_like_fwd, dy, w, nullptr, dx, dx_dtype, 0, B, Cout, OH, OW, Cin, kh, kw, stride, pad, H, W,
                       (hipStream_t)stream, "vs_conv_transpose2d_dgrad");
}

extern "C" int vs_conv_transpose2d_wgrad(int compute, const void* dy, const void* x, float* dw, int B, int Cin, int H, int W, int Cout,
                                         int kh, int kw, int stride, int pad, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_conv("vs_conv_transpose2d_wgrad", compute, dy, x, dw, B, Cin, H, W, Cout, kh, kw, stride, pad);
    if (rc) return rc;
    const int OH = (H - 1) * stride - 2 * pad + kh, OW = (W - 1) * stride - 2 * pad + kw;
    // dW[ci][(co,ky,kx)] = sum_pix x[b,ci,pix] dy[b,co,y*s-p+ky,x*s-p+kx]
    return VS_DISPATCH(compute, wgrad_like, x, dy, dw, B, Cin, H, W, Cout, OH, OW, kh, kw, stride, pad, workspace, workspace_bytes,
                       (hipStream_t)stream, "vs_conv_transpose2d_wgrad");
}
