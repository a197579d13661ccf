// vs_rollout.hip -- the residual latent integrator as ONE persistent launch per direction.
//
// Reference: networks/model.py:78-83 (python loop over t calling t_resnet) and networks/resnet.py:22-50
// (x <- x + W3 relu(W2 relu(W1 x + b1) + b2) + b3, n_blocks in series).  At the README recipes this is
// (n-1) * n_blocks * 3 dependent GEMMs with 16..128 rows and K, N <= 512 (WaveEq: 216 of them): pure
// launch/latency cost as separate kernels.  Here one workgroup owns a 16-row slab of the batch for the WHOLE
// rollout (rows are independent), keeps the running code x in LDS in fp32, and walks time x blocks x layers with
// workgroup barriers only -- no grid-wide synchronisation, no inter-workgroup traffic, any placement is correct.
//
// MI355X mapping:
//   * 8 waves (512 threads) per slab; a layer's output [16, N] is cut into 16-column MFMA tiles x K chunks so that
//     all 8 waves have work even when N is the tiny code size (K-split partials are summed in a fixed order in
//     LDS -> bitwise reproducible, no float atomics).
//   * activations (A operand, 16 rows) live in LDS, padded so ds_read_b128 is conflict free; weights (B operand)
//     are streamed straight from L2 into MFMA fragments with 16-byte loads, several k-steps in flight per wave
//     (the GEMV / M<=16 regime: an LDS round trip for an operand nobody shares would be pure overhead).  All
//     blocks' weights (WaveEq: 1.8 MB bf16) stay resident in the XCD L2 for the whole launch.
//   * bf16: v_mfma_f32_16x16x32_bf16; fp32 (parity mode): v_mfma_f32_16x16x4_f32 with 4 k per lane per 16-wide
//     group (lane q holds k = 4q..4q+3, MFMA j consumes element j on both operands).
//   * everything the backward pass needs (block inputs, both hidden activations) is written once, in the
//     [block][step][row][feature] order the batched weight-gradient GEMMs read directly.
// Backward-through-time runs the same structure in reverse with transposed weight copies, producing dr/dh2/dh1
// for every (block, step); the weight gradients are then three large vs_gemm calls per block (K = (n-1)*B).
#include "vs_common.h"

namespace {

constexpr int MAXB = 8;          // max residual blocks
constexpr int NW = 8;            // waves per workgroup (2 per SIMD: 256 VGPRs each for the weight prefetch ring)
constexpr int NT = NW * 64;

struct RollParams {
    int B, C, H, nb, n;          // batch, code size, hidden size, blocks, steps (n_forecast)
    const void* W[3 * MAXB];     // PACKED (vs_pack_rollout_weight). fwd: W1 [H,C], W2 [H,H], W3 [C,H];  bwd: W3^T [H,C], W2^T [H,H], W1^T [C,H]
    const float* bias[3 * MAXB]; // fwd only
    const float* x0;             // [B, C]
    float* t_codes;              // [B, n, C]
    float* residuals;            // [n-1, nb, B, C] or null
    void* xin_save;              // [nb, n-1, B, C]  compute dtype
    void* h1_save;               // [nb, n-1, B, H]
    void* h2_save;               // [nb, n-1, B, H]
    unsigned* m1_save;           // [nb, n-1, B, 32] ReLU sign bits of h1: thread (row, j) owns columns j + 32 u, bit u
    unsigned* m2_save;           // same for h2
    // backward
    const float* g;              // [B, n, C] gradient wrt every t_code
    float* dx0;                  // [B, C]
    void* dr_save;               // [nb, n-1, B, C]
    void* dh2_save;              // [nb, n-1, B, H]
    void* dh1_save;              // [nb, n-1, B, H]
};

template <int CT> struct RT;
template <> struct RT<VS_BF16> { typedef __bf16 T; static constexpr int KS = 32; static constexpr int U = 8; };
template <> struct RT<VS_F32> { typedef float T; static constexpr int KS = 16; static constexpr int U = 4; };

__host__ __device__ inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// partial[s][16][Np] (fp32) = in[16, K(chunk s)] * W[N, K]^T ; `in` is an LDS tile [16][KP] zero padded to a
// multiple of KS; W is global [N][K] row-major.  All NW waves cooperate: item = (n-tile, k-chunk).
template <int CT>
__device__ __forceinline__ f32x4 mma16(const u32x4& av, const u32x4& bv, f32x4 acc) {
    if constexpr (CT == VS_BF16) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&av), *reinterpret_cast<const bf16x8*>(&bv), acc,
                                                       0, 0, 0);
    } else {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(&av);
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(&bv);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], b4[j], acc, 0, 0, 0);
        return acc;
    }
}

constexpr int RING = 16;         // 16-byte weight loads kept in flight per lane in the streaming layer

// Heavy (H x H) layer: every work item is one 16-column tile over the full K = SPC k-steps (no K split), SPC a
// multiple of RING.  Branch-free software pipeline with static register indices: while the MFMA of piece u runs, the
// load of the piece RING steps ahead (same tile, or the wave's next tile) is already in flight; loads past the
// wave's last tile are clamped to its last piece (harmless re-reads) so no load sits under a condition.
template <int CT, int SPC>
__device__ __forceinline__ void layer_stream(const typename RT<CT>::T* in, int KP, const typename RT<CT>::T* Wp, int N, float* part, int Np) {
    typedef typename RT<CT>::T T;
    constexpr int KS = RT<CT>::KS, U = RT<CT>::U;
    static_assert(SPC % RING == 0, "SPC must be a multiple of RING");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ntiles = (N + 15) >> 4;
    if (wave >= ntiles) return;
    const int c = lane & 15, g = lane >> 4;
    const T* arow = in + c * KP + g * U;
    const int last_tile = wave + ((ntiles - 1 - wave) / NW) * NW;
    const int64_t tile_stride = (int64_t)SPC * 64 * U;
    const T* wl = Wp + (int64_t)lane * U;
    u32x4 ring[RING];
#pragma unroll
    for (int u = 0; u < RING; ++u) ring[u] = *reinterpret_cast<const u32x4*>(wl + wave * tile_stride + (int64_t)u * 64 * U);
    for (int nt = wave; nt < ntiles; nt += NW) {
        const int nxt = (nt + NW <= last_tile) ? nt + NW : last_tile;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int gq = 0; gq < SPC / RING; ++gq) {
#pragma unroll
            for (int u = 0; u < RING; ++u) {
                const int sidx = gq * RING + u;
                acc = mma16<CT>(*reinterpret_cast<const u32x4*>(arow + sidx * KS), ring[u], acc);
                const int ahead = sidx + RING;                            // static: same tile or the next one
                const T* src = (ahead < SPC) ? wl + nt * tile_stride + (int64_t)ahead * 64 * U
                                             : wl + nxt * tile_stride + (int64_t)(ahead - SPC) * 64 * U;
                ring[u] = *reinterpret_cast<const u32x4*>(src);
            }
        }
        float* dst = part + nt * 16 + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(4 * g + r) * Np] = acc[r];
    }
}

// Generic layer (small K or small N: the code-size layers, odd sizes): items = (16-column tile, K chunk), batches of 4
// k-steps.  partial[s][16][Np] (fp32) = in[16, K(chunk s)] * W[N, K]^T.
template <int CT>
__device__ __forceinline__ void layer_generic(const typename RT<CT>::T* in, int KP, int K, const typename RT<CT>::T* Wp, int N,
                                              float* part, int Np, int ksplit) {
    typedef typename RT<CT>::T T;
    constexpr int KS = RT<CT>::KS, U = RT<CT>::U;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ntiles = (N + 15) >> 4;
    const int ksteps = (K + KS - 1) / KS;
    const int spc = (ksteps + ksplit - 1) / ksplit;
    const int c = lane & 15, g = lane >> 4;
    const T* arow = in + c * KP + g * U;
    for (int item = wave; item < ntiles * ksplit; item += NW) {
        const int nt = item / ksplit, ks = item % ksplit;
        const int s0 = ks * spc;
        int s1 = s0 + spc;
        if (s1 > ksteps) s1 = ksteps;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const T* wt = Wp + ((int64_t)nt * ksteps * 64 + lane) * U;
        int sb = s0;
        for (; sb + 4 <= s1; sb += 4) {                 // full batches: unconditional loads
            u32x4 w4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) w4[u] = *reinterpret_cast<const u32x4*>(wt + (int64_t)(sb + u) * 64 * U);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = mma16<CT>(*reinterpret_cast<const u32x4*>(arow + (sb + u) * KS), w4[u], acc);
        }
        for (; sb < s1; ++sb)
            acc = mma16<CT>(*reinterpret_cast<const u32x4*>(arow + sb * KS), *reinterpret_cast<const u32x4*>(wt + (int64_t)sb * 64 * U), acc);
        float* dst = part + (ks * 16) * Np + nt * 16 + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(4 * g + r) * Np] = acc[r];
    }
}

// All NW waves cooperate on out[16, N] = in[16, K] W^T; the result lands in `part` as ksplit slabs of [16][Np].
template <int CT>
__device__ __forceinline__ void layer_partial(const typename RT<CT>::T* in, int KP, int K, const typename RT<CT>::T* Wp, int N,
                                              float* part, int Np, int ksplit) {
    constexpr int KS = RT<CT>::KS;
    const int ksteps = (K + KS - 1) / KS;
    const int ntiles = (N + 15) >> 4;
    if (ksplit == 1 && ntiles >= NW) {
        if (ksteps == 16) { layer_stream<CT, 16>(in, KP, Wp, N, part, Np); return; }
        if (ksteps == 32) { layer_stream<CT, 32>(in, KP, Wp, N, part, Np); return; }
        if (ksteps == 64) { layer_stream<CT, 64>(in, KP, Wp, N, part, Np); return; }
    }
    layer_generic<CT>(in, KP, K, Wp, N, part, Np, ksplit);
}

__device__ __forceinline__ float part_sum(const float* part, int Np, int ksplit, int row, int col) {
    float v = 0.f;
    for (int s = 0; s < ksplit; ++s) v += part[(s * 16 + row) * Np + col];
    return v;
}

__device__ __forceinline__ int ksplit_for(int N) {
    const int ntiles = (N + 15) >> 4;
    int ks = 1;
    while (ks * ntiles < NW && ks < 8) ks <<= 1;
    return ks;
}

struct Lds {
    float* xs;      // [16][Cf]   running code / running gradient, fp32
    void* a_c;      // [16][Ck]   C-wide MFMA operand (block input / dr)
    void* a_h1;     // [16][Hk]
    void* a_h2;     // [16][Hk]
    float* part;    // [ksplit*16][pitch of the layer]
    float* bias;    // fwd: [nb][2H + C] all biases, staged once
    int Cf, Ck, Hk, NpH, NpC;
};

template <int CT>
__device__ __forceinline__ Lds carve(char* smem, int C, int H, int nb) {
    typedef typename RT<CT>::T T;
    constexpr int KS = RT<CT>::KS;
    Lds L;
    L.Cf = round_up(C, 4);
    L.Ck = round_up(C, KS) + RT<CT>::U;            // + one 16-byte unit: conflict-free ds_read_b128
    L.Hk = round_up(H, KS) + RT<CT>::U;
    L.NpH = round_up(H, 16) + 4;
    L.NpC = round_up(C, 16) + 4;
    size_t off = 0;
    L.xs = reinterpret_cast<float*>(smem + off); off += (size_t)16 * L.Cf * 4;
    off = (off + 15) & ~(size_t)15;
    L.a_c = smem + off; off += (size_t)16 * L.Ck * sizeof(T);
    off = (off + 15) & ~(size_t)15;
    L.a_h1 = smem + off; off += (size_t)16 * L.Hk * sizeof(T);
    off = (off + 15) & ~(size_t)15;
    L.a_h2 = smem + off; off += (size_t)16 * L.Hk * sizeof(T);
    off = (off + 15) & ~(size_t)15;
    L.bias = reinterpret_cast<float*>(smem + off); off += (size_t)nb * (2 * H + C) * 4;
    off = (off + 15) & ~(size_t)15;
    L.part = reinterpret_cast<float*>(smem + off);
    return L;
}

template <int CT>
size_t lds_bytes(int C, int H, int nb) {
    typedef typename RT<CT>::T T;
    constexpr int KS = RT<CT>::KS;
    const int Cf = round_up(C, 4), Ck = round_up(C, KS) + RT<CT>::U, Hk = round_up(H, KS) + RT<CT>::U;
    const int NpH = round_up(H, 16) + 4, NpC = round_up(C, 16) + 4;
    size_t off = (size_t)16 * Cf * 4;
    off = (off + 15) & ~(size_t)15; off += (size_t)16 * Ck * sizeof(T);
    off = (off + 15) & ~(size_t)15; off += (size_t)16 * Hk * sizeof(T);
    off = (off + 15) & ~(size_t)15; off += (size_t)16 * Hk * sizeof(T);
    off = (off + 15) & ~(size_t)15; off += (size_t)nb * (2 * H + C) * 4;
    off = (off + 15) & ~(size_t)15;
    // partials: a layer with N outputs uses ksplit_for(N) * 16 rows of pitch round_up(N,16)+4
    int ksH = 1, ksC = 1;
    { int nt = (H + 15) / 16; while (ksH * nt < NW && ksH < 8) ksH <<= 1; }
    { int nt = (C + 15) / 16; while (ksC * nt < NW && ksC < 8) ksC <<= 1; }
    const size_t pH = (size_t)16 * ksH * NpH * 4, pC = (size_t)16 * ksC * NpC * 4;
    off += pH > pC ? pH : pC;
    return off;
}

template <int CT>
__device__ __forceinline__ void zero_tile(typename RT<CT>::T* t, int elems) {
    for (int i = threadIdx.x; i < elems; i += NT) t[i] = (typename RT<CT>::T)0.f;
}

template <int CT>
__global__ __launch_bounds__(NT) void rollout_fwd_kernel(RollParams p) {
    typedef typename RT<CT>::T T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Lds L = carve<CT>(smem, p.C, p.H, p.nb);
    T* a_c = (T*)L.a_c; T* a_h1 = (T*)L.a_h1; T* a_h2 = (T*)L.a_h2;
    const int row0 = blockIdx.x * 16;
    const int B = p.B, C = p.C, H = p.H, n = p.n, nb = p.nb;
    const int ksH = ksplit_for(H), ksC = ksplit_for(C);

    zero_tile<CT>(a_c, 16 * L.Ck); zero_tile<CT>(a_h1, 16 * L.Hk); zero_tile<CT>(a_h2, 16 * L.Hk);
    for (int b = 0; b < nb; ++b) {
        float* bb = L.bias + b * (2 * H + C);
        for (int i = threadIdx.x; i < H; i += NT) { bb[i] = p.bias[3 * b][i]; bb[H + i] = p.bias[3 * b + 1][i]; }
        for (int i = threadIdx.x; i < C; i += NT) bb[2 * H + i] = p.bias[3 * b + 2][i];
    }
    for (int c = threadIdx.x & 31, r = threadIdx.x >> 5; c < C; c += 32) {
        const float v = (row0 + r < B) ? p.x0[(int64_t)(row0 + r) * C + c] : 0.f;
        L.xs[r * L.Cf + c] = v;
        if (row0 + r < B) p.t_codes[((int64_t)(row0 + r) * n) * C + c] = v;
    }
    __syncthreads();

    for (int t = 1; t < n; ++t) {
        for (int b = 0; b < nb; ++b) {
            const T* W1 = (const T*)p.W[3 * b]; const T* W2 = (const T*)p.W[3 * b + 1]; const T* W3 = (const T*)p.W[3 * b + 2];
            const float* b1 = L.bias + b * (2 * H + C); const float* b2 = b1 + H; const float* b3 = b2 + H;
            const int64_t sbase = ((int64_t)b * (n - 1) + (t - 1)) * B + row0;     // row index into the [nb][n-1][B][.] saves
            // block input -> MFMA operand (+ saved for dW1)
            for (int c = threadIdx.x & 31, r = threadIdx.x >> 5; c < C; c += 32) {
                const T v = (T)L.xs[r * L.Cf + c];
                a_c[r * L.Ck + c] = v;
                if (row0 + r < B) ((T*)p.xin_save)[(sbase + r) * C + c] = v;
            }
            __syncthreads();
            layer_partial<CT>(a_c, L.Ck, C, W1, H, L.part, L.NpH, ksH);
            __syncthreads();
            {
                unsigned bits = 0u;
                int u = 0;
                for (int c = threadIdx.x & 31, r = threadIdx.x >> 5; c < H; c += 32, ++u) {
                    float v = part_sum(L.part, L.NpH, ksH, r, c) + b1[c];
                    v = v > 0.f ? v : 0.f;
                    const T hv = (T)v;
                    if ((float)hv > 0.f && u < 32) bits |= 1u << u;
                    a_h1[r * L.Hk + c] = hv;
                    if (row0 + r < B) ((T*)p.h1_save)[(sbase + r) * H + c] = hv;
                }
                if (row0 + (threadIdx.x >> 5) < B) p.m1_save[(sbase + (threadIdx.x >> 5)) * 32 + (threadIdx.x & 31)] = bits;
            }
            __syncthreads();
            layer_partial<CT>(a_h1, L.Hk, H, W2, H, L.part, L.NpH, ksH);
            __syncthreads();
            {
                unsigned bits = 0u;
                int u = 0;
                for (int c = threadIdx.x & 31, r = threadIdx.x >> 5; c < H; c += 32, ++u) {
                    float v = part_sum(L.part, L.NpH, ksH, r, c) + b2[c];
                    v = v > 0.f ? v : 0.f;
                    const T hv = (T)v;
                    if ((float)hv > 0.f && u < 32) bits |= 1u << u;
                    a_h2[r * L.Hk + c] = hv;
                    if (row0 + r < B) ((T*)p.h2_save)[(sbase + r) * H + c] = hv;
                }
                if (row0 + (threadIdx.x >> 5) < B) p.m2_save[(sbase + (threadIdx.x >> 5)) * 32 + (threadIdx.x & 31)] = bits;
            }
            __syncthreads();
            layer_partial<CT>(a_h2, L.Hk, H, W3, C, L.part, L.NpC, ksC);
            __syncthreads();
            for (int c = threadIdx.x & 31, r = threadIdx.x >> 5; c < C; c += 32) {
                const float res = part_sum(L.part, L.NpC, ksC, r, c) + b3[c];
                L.xs[r * L.Cf + c] += res;
                if (p.residuals && row0 + r < B)
                    p.residuals[(((int64_t)(t - 1) * nb + b) * B + row0 + r) * C + c] = res;
            }
            __syncthreads();
        }
        for (int c = threadIdx.x & 31, r = threadIdx.x >> 5; c < C; c += 32) {
            if (row0 + r < B) p.t_codes[((int64_t)(row0 + r) * n + t) * C + c] = L.xs[r * L.Cf + c];
        }
    }
}

template <int CT>
__global__ __launch_bounds__(NT) void rollout_bwd_kernel(RollParams p) {
    typedef typename RT<CT>::T T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Lds L = carve<CT>(smem, p.C, p.H, p.nb);
    T* a_c = (T*)L.a_c; T* a_h1 = (T*)L.a_h1; T* a_h2 = (T*)L.a_h2;
    const int row0 = blockIdx.x * 16;
    const int B = p.B, C = p.C, H = p.H, n = p.n, nb = p.nb;
    const int ksH = ksplit_for(H), ksC = ksplit_for(C);

    zero_tile<CT>(a_c, 16 * L.Ck); zero_tile<CT>(a_h1, 16 * L.Hk); zero_tile<CT>(a_h2, 16 * L.Hk);
    for (int i = threadIdx.x; i < 16 * L.Cf; i += NT) L.xs[i] = 0.f;
    __syncthreads();

    unsigned nx2 = 0u, nx1 = 0u;
    if (n > 1 && row0 + (threadIdx.x >> 5) < B) {
        const int64_t nbase = ((int64_t)(nb - 1) * (n - 1) + (n - 2)) * B + row0;
        nx2 = p.m2_save[(nbase + (threadIdx.x >> 5)) * 32 + (threadIdx.x & 31)];
        nx1 = p.m1_save[(nbase + (threadIdx.x >> 5)) * 32 + (threadIdx.x & 31)];
    }
    for (int t = n - 1; t >= 1; --t) {
        for (int c = threadIdx.x & 31, r = threadIdx.x >> 5; c < C; c += 32) {
            if (row0 + r < B) L.xs[r * L.Cf + c] += p.g[((int64_t)(row0 + r) * n + t) * C + c];
        }
        __syncthreads();
        for (int b = nb - 1; b >= 0; --b) {
            const T* W3T = (const T*)p.W[3 * b]; const T* W2T = (const T*)p.W[3 * b + 1]; const T* W1T = (const T*)p.W[3 * b + 2];
            const int64_t sbase = ((int64_t)b * (n - 1) + (t - 1)) * B + row0;
            // dr = gradient wrt the residual = running gradient (rounded to the compute type for the MFMA and dW3)
            for (int c = threadIdx.x & 31, r = threadIdx.x >> 5; c < C; c += 32) {
                const T v = (T)L.xs[r * L.Cf + c];
                a_c[r * L.Ck + c] = v;
                if (row0 + r < B) ((T*)p.dr_save)[(sbase + r) * C + c] = v;
            }
            __syncthreads();
            // ReLU sign bits of this block-step were fetched one block-step ahead; fetch the next ones now
            const int mr = threadIdx.x >> 5, mc = threadIdx.x & 31;
            const bool mrow = row0 + mr < B;
            const unsigned bits2 = nx2, bits1 = nx1;
            {
                int tb = t, bb = b - 1;
                if (bb < 0) { bb = nb - 1; tb = t - 1; }
                if (tb >= 1 && mrow) {
                    const int64_t nbase = ((int64_t)bb * (n - 1) + (tb - 1)) * B + row0;
                    nx2 = p.m2_save[(nbase + mr) * 32 + mc];
                    nx1 = p.m1_save[(nbase + mr) * 32 + mc];
                }
            }
            const T* h2row = (const T*)p.h2_save + (sbase + mr) * H;
            const T* h1row = (const T*)p.h1_save + (sbase + mr) * H;
            layer_partial<CT>(a_c, L.Ck, C, W3T, H, L.part, L.NpH, ksH);          // dh2 = (dr W3) * relu'(h2)
            __syncthreads();
            {
                int u = 0;
                for (int c = mc; c < H; c += 32, ++u) {
                    const bool on = u < 32 ? ((bits2 >> u) & 1u) != 0u : (mrow && (float)h2row[c] > 0.f);
                    const T dv = (T)(on ? part_sum(L.part, L.NpH, ksH, mr, c) : 0.f);
                    a_h2[mr * L.Hk + c] = dv;
                    if (mrow) ((T*)p.dh2_save)[(sbase + mr) * H + c] = dv;
                }
            }
            __syncthreads();
            layer_partial<CT>(a_h2, L.Hk, H, W2T, H, L.part, L.NpH, ksH);         // dh1 = (dh2 W2) * relu'(h1)
            __syncthreads();
            {
                int u = 0;
                for (int c = mc; c < H; c += 32, ++u) {
                    const bool on = u < 32 ? ((bits1 >> u) & 1u) != 0u : (mrow && (float)h1row[c] > 0.f);
                    const T dv = (T)(on ? part_sum(L.part, L.NpH, ksH, mr, c) : 0.f);
                    a_h1[mr * L.Hk + c] = dv;
                    if (mrow) ((T*)p.dh1_save)[(sbase + mr) * H + c] = dv;
                }
            }
            __syncthreads();
            layer_partial<CT>(a_h1, L.Hk, H, W1T, C, L.part, L.NpC, ksC);         // dx_in = dx_out + dh1 W1
            __syncthreads();
            for (int c = threadIdx.x & 31, r = threadIdx.x >> 5; c < C; c += 32) {
                L.xs[r * L.Cf + c] += part_sum(L.part, L.NpC, ksC, r, c);
            }
            __syncthreads();
        }
    }
    for (int c = threadIdx.x & 31, r = threadIdx.x >> 5; c < C; c += 32) {
        if (row0 + r < B) p.dx0[(int64_t)(row0 + r) * C + c] = L.xs[r * L.Cf + c] + p.g[((int64_t)(row0 + r) * n) * C + c];
    }
}

template <int CT>
int launch_roll(bool fwd, const RollParams& p, hipStream_t stream) {
    const size_t smem = lds_bytes<CT>(p.C, p.H, p.nb);
    if (smem > 160 * 1024) return vs_fail(VS_ERR_UNSUPPORTED, "vs_mlp_rollout: C=%d H=%d needs %zu B of LDS (> 160 KiB)", p.C, p.H, smem);
    const void* kfn = fwd ? (const void*)rollout_fwd_kernel<CT> : (const void*)rollout_bwd_kernel<CT>;
    if (smem > 64 * 1024) {
        if (hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return vs_fail(VS_ERR_LAUNCH, "vs_mlp_rollout: cannot raise dynamic LDS limit to %zu", smem);
    }
    dim3 grid((unsigned)((p.B + 15) / 16));
    if (fwd) hipLaunchKernelGGL(rollout_fwd_kernel<CT>, grid, dim3(NT), smem, stream, p);
    else hipLaunchKernelGGL(rollout_bwd_kernel<CT>, grid, dim3(NT), smem, stream, p);
    VS_CHECK_LAUNCH("vs_mlp_rollout");
    return VS_OK;
}

int check_common(int compute, int B, int C, int H, int nb, int n) {
    VS_CHECK_ARG(compute == VS_F32 || compute == VS_BF16, "vs_mlp_rollout: compute type %d", compute);
    VS_CHECK_ARG(B > 0 && C > 0 && H > 0 && n >= 1, "vs_mlp_rollout: bad sizes B=%d C=%d H=%d n=%d", B, C, H, n);
    VS_CHECK_ARG(nb >= 1 && nb <= MAXB, "vs_mlp_rollout: n_blocks=%d (supported: 1..%d)", nb, MAXB);
    return VS_OK;
}

}  // namespace

extern "C" int vs_mlp_rollout_fwd(int compute, int B, int C, int H, int n_blocks, int n_steps, const float* x0,
                                  const void* const* weights, const float* const* biases, float* t_codes, float* residuals,
                                  void* xin_save, void* h1_save, void* h2_save, uint32_t* m1_save, uint32_t* m2_save, void* stream) {
    int rc = check_common(compute, B, C, H, n_blocks, n_steps);
    if (rc != VS_OK) return rc;
    VS_CHECK_ARG(x0 && weights && biases && t_codes && xin_save && h1_save && h2_save && m1_save && m2_save,
                 "vs_mlp_rollout_fwd: null pointer");
    RollParams p = {};
    p.B = B; p.C = C; p.H = H; p.nb = n_blocks; p.n = n_steps;
    for (int i = 0; i < 3 * n_blocks; ++i) { p.W[i] = weights[i]; p.bias[i] = biases[i]; }
    p.x0 = x0; p.t_codes = t_codes; p.residuals = residuals;
    p.xin_save = xin_save; p.h1_save = h1_save; p.h2_save = h2_save; p.m1_save = m1_save; p.m2_save = m2_save;
    return compute == VS_BF16 ? launch_roll<VS_BF16>(true, p, (hipStream_t)stream) : launch_roll<VS_F32>(true, p, (hipStream_t)stream);
}

extern "C" int vs_mlp_rollout_bwd(int compute, int B, int C, int H, int n_blocks, int n_steps, const float* grad_t_codes,
                                  const void* const* weights_t, const void* h1_save, const void* h2_save,
                                  const uint32_t* m1_save, const uint32_t* m2_save, float* dx0, void* dr_save, void* dh2_save,
                                  void* dh1_save, void* stream) {
    int rc = check_common(compute, B, C, H, n_blocks, n_steps);
    if (rc != VS_OK) return rc;
    VS_CHECK_ARG(grad_t_codes && weights_t && h1_save && h2_save && m1_save && m2_save && dx0 && dr_save && dh2_save && dh1_save,
                 "vs_mlp_rollout_bwd: null pointer");
    RollParams p = {};
    p.B = B; p.C = C; p.H = H; p.nb = n_blocks; p.n = n_steps;
    for (int i = 0; i < 3 * n_blocks; ++i) p.W[i] = weights_t[i];
    p.g = grad_t_codes; p.dx0 = dx0;
    p.h1_save = const_cast<void*>(h1_save); p.h2_save = const_cast<void*>(h2_save);
    p.m1_save = const_cast<uint32_t*>(m1_save); p.m2_save = const_cast<uint32_t*>(m2_save);
    p.dr_save = dr_save; p.dh2_save = dh2_save; p.dh1_save = dh1_save;
    return compute == VS_BF16 ? launch_roll<VS_BF16>(false, p, (hipStream_t)stream) : launch_roll<VS_F32>(false, p, (hipStream_t)stream);
}

// dst[c, r] = (dst_dtype) src[r, c]   (src row-major [rows, cols]); transposed bf16/fp32 weight copies for the backward rollout
namespace {
__global__ __launch_bounds__(256) void transpose_cast_kernel(const void* src, int sd, void* dst, int dd, int rows, int cols) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int r = by + j, c = bx + tx;
        tile[j][tx] = (r < rows && c < cols) ? vs_ld(src, sd, (int64_t)r * cols + c) : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = bx + j, r = by + tx;
        if (r < rows && c < cols) vs_st(dst, dd, (int64_t)c * rows + r, tile[tx][j]);
    }
}
}  // namespace

extern "C" int vs_transpose_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int rows, int cols, void* stream) {
    VS_CHECK_ARG(src && dst && rows > 0 && cols > 0, "vs_transpose_cast: bad argument");
    dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
    hipLaunchKernelGGL(transpose_cast_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, src_dtype, dst, dst_dtype, rows, cols);
    VS_CHECK_LAUNCH("vs_transpose_cast");
    return VS_OK;
}

// ---- weight pre-pack: logical L[N][K] -> MFMA-fragment order, one contiguous 1 KiB piece per (n-tile, k-step) ---------
// piece (nt, s), lane l = 16 g + c holds L[16 nt + c][s*KS + g*U .. +U-1] (zero padded).  Done once per optimizer
// step together with the fp32 -> compute-type conversion of the master weights (SURVEY.md 8b allows a pre-pack
// that is invalidated by the optimizer step).
namespace {
template <int CT>
__global__ __launch_bounds__(256) void pack_kernel(const float* src, int transpose, int N, int K, typename RT<CT>::T* dst) {
    typedef typename RT<CT>::T T;
    constexpr int KS = RT<CT>::KS, U = RT<CT>::U;
    const int ntiles = (N + 15) / 16, ksteps = (K + KS - 1) / KS;
    const int64_t units = (int64_t)ntiles * ksteps * 64;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < units; u += (int64_t)gridDim.x * 256) {
        const int lane = (int)(u & 63);
        const int64_t piece = u >> 6;
        const int s = (int)(piece % ksteps), nt = (int)(piece / ksteps);
        const int n = nt * 16 + (lane & 15), k0 = s * KS + (lane >> 4) * U;
        T tmp[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int k = k0 + j;
            float v = 0.f;
            if (n < N && k < K) v = transpose ? src[(int64_t)k * N + n] : src[(int64_t)n * K + k];
            tmp[j] = (T)v;
        }
        *reinterpret_cast<u32x4*>(dst + u * U) = *reinterpret_cast<u32x4*>(tmp);
    }
}
}  // namespace

extern "C" size_t vs_rollout_packed_elems(int compute, int N, int K) {
    const int KS = compute == VS_BF16 ? 32 : 16;
    return (size_t)((N + 15) / 16) * 16 * (size_t)((K + KS - 1) / KS) * KS;
}

extern "C" int vs_pack_rollout_weight(int compute, const float* src, int transpose, int N, int K, void* dst, void* stream) {
    VS_CHECK_ARG(compute == VS_F32 || compute == VS_BF16, "vs_pack_rollout_weight: compute type %d", compute);
    VS_CHECK_ARG(src && dst && N > 0 && K > 0, "vs_pack_rollout_weight: bad argument");
    const int KS = compute == VS_BF16 ? 32 : 16;
    int64_t units = (int64_t)((N + 15) / 16) * ((K + KS - 1) / KS) * 64;
    int64_t blocks = (units + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (compute == VS_BF16)
        hipLaunchKernelGGL(pack_kernel<VS_BF16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, transpose, N, K, (__bf16*)dst);
    else
        hipLaunchKernelGGL(pack_kernel<VS_F32>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, transpose, N, K, (float*)dst);
    VS_CHECK_LAUNCH("vs_pack_rollout_weight");
    return VS_OK;
}
