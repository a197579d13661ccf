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
    unsigned* m2_save;           // same for h2  (both [nb, n-1, B, P, 32]: one word per part)
    int P;                       // workgroups per slab (hidden dimension split)
    unsigned long long* xbuf;    // exchange area [2][nslabs][P][16*Cf] of {epoch, value} granules
    unsigned* xerr;              // timeout flag
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

constexpr int RING = 8;          // 16-byte weight loads kept in flight per lane in the streaming layer

// One layer of one slab, as seen by one workgroup:  out[16, tiles tile0..tile0+ntl) = in[16, k-steps) * Wp^T.
//   in     LDS tile [16][KP] (compute type), its k-step a_s0 + i pairs with weight k-step w_s0 + i, i < nsteps
//   Wp     packed weight (vs_pack_rollout_weight) of a matrix with `ksteps_total` k-steps per column tile
//   part   fp32 LDS result: ksplit slabs of [16][Np]; column index is LOCAL (0 = first column of tile0)
struct LayerArgs {
    int KP, a_s0, w_s0, nsteps, ksteps_total, tile0, ntl, Np, ksplit;
};

// Streaming form: every item is one column tile over SPC k-steps (no K split), SPC a multiple of RING.  Branch-free
// software pipeline with static register indices: while the MFMA of piece u runs, the load of the piece RING steps ahead
// (same tile, or the wave's next tile) is already in flight; loads past the wave's last tile are clamped to its last
// piece (harmless re-reads) so no load sits under a condition.
template <int CT>
struct Ring { u32x4 r[RING]; };

template <int CT>
__device__ __forceinline__ void stream_issue(const typename RT<CT>::T* Wp, const LayerArgs& a, Ring<CT>& ring) {
    typedef typename RT<CT>::T T;
    constexpr int U = RT<CT>::U;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = wave < a.ntl ? wave : a.ntl - 1;                     // idle waves fetch a valid piece (discarded)
    const T* wl = Wp + ((int64_t)(a.tile0 + tile) * a.ksteps_total + a.w_s0) * 64 * U + (int64_t)lane * U;
#pragma unroll
    for (int u = 0; u < RING; ++u) ring.r[u] = *reinterpret_cast<const u32x4*>(wl + (int64_t)u * 64 * U);
}

template <int CT, int SPC, bool REFILL>
__device__ __forceinline__ f32x4 stream_tile(const typename RT<CT>::T* arow, const typename RT<CT>::T* wcur, const typename RT<CT>::T* wnext,
                                             Ring<CT>& ring) {
    typedef typename RT<CT>::T T;
    constexpr int KS = RT<CT>::KS, U = RT<CT>::U;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int gq = 0; gq < SPC / RING; ++gq) {
#pragma unroll
        for (int u = 0; u < RING; ++u) {
            const int sidx = gq * RING + u;
            acc = mma16<CT>(*reinterpret_cast<const u32x4*>(arow + sidx * KS), ring.r[u], acc);
            const int ahead = sidx + RING;                                // static: same tile or the next one
            if (ahead < SPC) ring.r[u] = *reinterpret_cast<const u32x4*>(wcur + (int64_t)ahead * 64 * U);
            else if (REFILL) ring.r[u] = *reinterpret_cast<const u32x4*>(wnext + (int64_t)(ahead - SPC) * 64 * U);
        }
    }
    return acc;
}

// Streaming form: every item is one column tile over SPC k-steps (no K split), SPC a multiple of RING.  Branch-free
// software pipeline with static register indices: the ring was filled by stream_issue() BEFORE the previous layer's
// epilogue and barrier (weights do not depend on activations), and while the MFMA of piece u runs the load of the
// piece RING steps ahead (same tile, or the wave's next tile) is in flight.  The last tile of a wave refills nothing.
template <int CT, int SPC>
__device__ __forceinline__ void stream_run(const typename RT<CT>::T* in, const typename RT<CT>::T* Wp, float* part, const LayerArgs& a,
                                           Ring<CT>& ring) {
    typedef typename RT<CT>::T T;
    constexpr int KS = RT<CT>::KS, U = RT<CT>::U;
    static_assert(SPC % RING == 0, "SPC must be a multiple of RING");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= a.ntl) return;
    const int c = lane & 15, g = lane >> 4;
    const T* arow = in + c * a.KP + g * U + a.a_s0 * KS;
    const int64_t tile_stride = (int64_t)a.ksteps_total * 64 * U;
    const T* wl = Wp + (int64_t)a.tile0 * tile_stride + ((int64_t)a.w_s0 * 64 + lane) * U;
    int nt = wave;
    for (; nt + NW < a.ntl; nt += NW) {
        const f32x4 acc = stream_tile<CT, SPC, true>(arow, wl + nt * tile_stride, wl + (nt + NW) * tile_stride, ring);
        float* dst = part + nt * 16 + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(4 * g + r) * a.Np] = acc[r];
    }
    const f32x4 acc = stream_tile<CT, SPC, false>(arow, wl + nt * tile_stride, wl, ring);
    float* dst = part + nt * 16 + c;
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(4 * g + r) * a.Np] = acc[r];
}

__device__ __forceinline__ bool streams(const LayerArgs& a) {
    return a.ksplit == 1 && a.ntl >= NW && (a.nsteps == 16 || a.nsteps == 32 || a.nsteps == 64);
}

// Small form: a wave's (tile, K chunk) items need at most 8 weight pieces in total (the code-size layers: K <= one or
// two k-steps, or a handful of tiles).  All pieces are fetched by small_issue() ahead of the previous epilogue/barrier.
constexpr int SMALLN = 4;
template <int CT>
struct Small { u32x4 w[SMALLN]; };

__device__ __forceinline__ bool small_fits(const LayerArgs& a) {
    // one k-step per item and at most SMALLN items per wave
    const int spc = (a.nsteps + a.ksplit - 1) / a.ksplit;
    return spc == 1 && (a.ntl * a.ksplit + NW - 1) / NW <= SMALLN;
}

template <int CT>
__device__ __forceinline__ void small_issue(const typename RT<CT>::T* Wp, const LayerArgs& a, Small<CT>& sm) {
    constexpr int U = RT<CT>::U;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nitems = a.ntl * a.ksplit;
#pragma unroll
    for (int q = 0; q < SMALLN; ++q) {
        int item = wave + NW * q;
        if (item >= nitems) item = nitems - 1;                             // clamp: always a valid address, value unused
        const int nt = a.ksplit == 1 ? item : item / a.ksplit;
        const int ks = a.ksplit == 1 ? 0 : item - nt * a.ksplit;
        sm.w[q] = *reinterpret_cast<const u32x4*>(Wp + (((int64_t)(a.tile0 + nt) * a.ksteps_total + a.w_s0 + ks) * 64 + lane) * U);
    }
}

template <int CT>
__device__ __forceinline__ void small_run(const typename RT<CT>::T* in, float* part, const LayerArgs& a, const Small<CT>& sm) {
    typedef typename RT<CT>::T T;
    constexpr int KS = RT<CT>::KS, U = RT<CT>::U;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nitems = a.ntl * a.ksplit;
    const int c = lane & 15, g = lane >> 4;
    const T* arow = in + c * a.KP + g * U + a.a_s0 * KS;
#pragma unroll
    for (int q = 0; q < SMALLN; ++q) {
        const int item = wave + NW * q;
        if (item < nitems) {
            const int nt = a.ksplit == 1 ? item : item / a.ksplit;
            const int ks = a.ksplit == 1 ? 0 : item - nt * a.ksplit;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = mma16<CT>(*reinterpret_cast<const u32x4*>(arow + ks * KS), sm.w[q], acc);
            float* dst = part + (ks * 16) * a.Np + nt * 16 + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(4 * g + r) * a.Np] = acc[r];
        }
    }
}

// Generic form (small K or few tiles: the code-size layers, odd sizes): items = (column tile, K chunk), batches of 4.
template <int CT>
__device__ __forceinline__ void layer_generic(const typename RT<CT>::T* in, const typename RT<CT>::T* Wp, float* part, const LayerArgs& a) {
    typedef typename RT<CT>::T T;
    constexpr int KS = RT<CT>::KS, U = RT<CT>::U;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int spc = (a.nsteps + a.ksplit - 1) / a.ksplit;
    const int c = lane & 15, g = lane >> 4;
    const T* arow = in + c * a.KP + g * U + a.a_s0 * KS;
    for (int item = wave; item < a.ntl * a.ksplit; item += NW) {
        const int nt = item / a.ksplit, ks = item % a.ksplit;
        const int s0 = ks * spc;
        int s1 = s0 + spc;
        if (s1 > a.nsteps) s1 = a.nsteps;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const T* wt = Wp + (((int64_t)(a.tile0 + nt) * a.ksteps_total + a.w_s0) * 64 + lane) * U;
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
        float* dst = part + (ks * 16) * a.Np + nt * 16 + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(4 * g + r) * a.Np] = acc[r];
    }
}

// A layer in two halves: issue() starts the first weight loads (call it before the previous layer's epilogue and
// barrier), run() does the MFMAs once the activations are in LDS.  Two flavours so that only one register ring is
// ever live: the hidden x hidden layer (streaming, else generic) and the code-size layers (small, else generic).
template <int CT>
struct StagedStream {
    Ring<CT> ring;
    bool on;
    __device__ __forceinline__ void issue(const typename RT<CT>::T* Wp, const LayerArgs& a) {
        on = streams(a);
        if (on) stream_issue<CT>(Wp, a, ring);
    }
    __device__ __forceinline__ void run(const typename RT<CT>::T* in, const typename RT<CT>::T* Wp, float* part, const LayerArgs& a) {
        if (on) {
            if (a.nsteps == 16) stream_run<CT, 16>(in, Wp, part, a, ring);
            else if (a.nsteps == 32) stream_run<CT, 32>(in, Wp, part, a, ring);
            else stream_run<CT, 64>(in, Wp, part, a, ring);
        } else layer_generic<CT>(in, Wp, part, a);
    }
};

template <int CT>
struct StagedSmall {
    Small<CT> sm;
    bool on;
    __device__ __forceinline__ void issue(const typename RT<CT>::T* Wp, const LayerArgs& a) {
        on = small_fits(a);
        if (on) small_issue<CT>(Wp, a, sm);
    }
    __device__ __forceinline__ void run(const typename RT<CT>::T* in, const typename RT<CT>::T* Wp, float* part, const LayerArgs& a) {
        if (on) small_run<CT>(in, part, a, sm);
        else layer_generic<CT>(in, Wp, part, a);
    }
};

__device__ __forceinline__ float part_sum(const float* part, int Np, int ksplit, int row, int col) {
    float v = 0.f;
    for (int s = 0; s < ksplit; ++s) v += part[(s * 16 + row) * Np + col];
    return v;
}

__host__ __device__ inline int ksplit_for(int ntiles, int nsteps) {
    int ks = 1;
    while (ks * ntiles < NW && ks < 8 && ks * 2 <= nsteps) ks <<= 1;
    return ks;
}

// ---- inter-workgroup all-reduce of a [16, C] fp32 partial among the P workgroups of one slab --------------------------
// Data-tagged 8-byte granules {epoch, value}: each workgroup publishes its partial with one relaxed agent-scope 8-byte
// store per element (write-through, untorn) into ITS slot of a double-buffered exchange area and reads every peer's
// granule with relaxed agent-scope loads until the tag equals the epoch -- the data is the flag, no fence, no
// dependence on placement.  Every workgroup sums the P partials in part order, so all P hold bitwise identical
// results.  Double buffering is sufficient: a workgroup can only publish epoch e+2 (same buffer as e) after it has
// finished epoch e+1, which needs every peer's e+1 granules, which each peer publishes only after reading epoch e.
// The area is zeroed by a fill kernel before every launch (epochs restart at 1).  Spins are bounded: on a timeout
// the error word is set and the kernel runs on (wrong result, no hang).
typedef unsigned long long u64;

struct Exchange {
    u64* base;            // [2][nslabs][P][16 * Cx]
    unsigned* err;
    int nslabs, P, Cx, slab, part;
    __device__ __forceinline__ u64* slot(int buf, int who) const {
        return base + (((int64_t)buf * nslabs + slab) * P + who) * (16 * Cx);
    }
};

__device__ __forceinline__ float exchange_sum(const Exchange& x, unsigned epoch, int idx, float mine) {
    const int buf = epoch & 1;
    u64 g = ((u64)epoch << 32) | (u64)__float_as_uint(mine);
    __hip_atomic_store(x.slot(buf, x.part) + idx, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float total = 0.f;
    for (int q = 0; q < x.P; ++q) {
        float v = mine;
        if (q != x.part) {
            const u64* src = x.slot(buf, q) + idx;
            u64 got = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while ((unsigned)(got >> 32) != epoch) {
                if (++spins > (1u << 22)) { atomicOr(x.err, 1u); break; }
                __builtin_amdgcn_s_sleep(1);
                got = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            v = __uint_as_float((unsigned)got);
        }
        total += v;
    }
    return total;
}

struct Lds {
    float* xs;      // [16][Cf]   running code / running gradient, fp32
    void* a_c;      // [16][Ck]   C-wide MFMA operand (block input / dr)
    void* a_h1;     // [16][Hk]
    void* a_h2;     // [16][Hk]
    float* part;    // partial sums of the current layer
    float* bias;    // fwd: [nb][2H + C] all biases, staged once
    int Cf, Ck, Hk, NpH, NpC;
};

__host__ __device__ inline size_t lds_layout(int C, int H, int nb, int esize, int KS, int U, int* Cf, int* Ck, int* Hk, int* NpH, int* NpC,
                                             size_t* off_c, size_t* off_h1, size_t* off_h2, size_t* off_bias, size_t* off_part) {
    *Cf = round_up(C, 4);
    *Ck = round_up(C, KS) + U;            // + one 16-byte unit: conflict-free ds_read_b128
    *Hk = round_up(H, KS) + U;
    *NpH = round_up(H, 16) + 4;
    *NpC = round_up(C, 16) + 4;
    size_t off = (size_t)16 * (*Cf) * 4;
    off = (off + 15) & ~(size_t)15; *off_c = off; off += (size_t)16 * (*Ck) * esize;
    off = (off + 15) & ~(size_t)15; *off_h1 = off; off += (size_t)16 * (*Hk) * esize;
    off = (off + 15) & ~(size_t)15; *off_h2 = off; off += (size_t)16 * (*Hk) * esize;
    off = (off + 15) & ~(size_t)15; *off_bias = off; off += (size_t)nb * (2 * H + C) * 4;
    off = (off + 15) & ~(size_t)15; *off_part = off;
    // partials: worst case over the layer shapes used (full-H output unsplit; C output with ksplit 8)
    const size_t pH = (size_t)16 * 8 * (*NpH) * 4 / ((H + 15) / 16 >= NW ? 8 : 1);
    const size_t pC = (size_t)16 * 8 * (*NpC) * 4;
    off += pH > pC ? pH : pC;
    return off;
}

template <int CT>
__device__ __forceinline__ Lds carve(char* smem, int C, int H, int nb) {
    Lds L;
    size_t oc, o1, o2, ob, op;
    lds_layout(C, H, nb, (int)sizeof(typename RT<CT>::T), RT<CT>::KS, RT<CT>::U, &L.Cf, &L.Ck, &L.Hk, &L.NpH, &L.NpC, &oc, &o1, &o2, &ob, &op);
    L.xs = reinterpret_cast<float*>(smem);
    L.a_c = smem + oc; L.a_h1 = smem + o1; L.a_h2 = smem + o2;
    L.bias = reinterpret_cast<float*>(smem + ob);
    L.part = reinterpret_cast<float*>(smem + op);
    return L;
}

template <int CT>
__device__ __forceinline__ void zero_tile(typename RT<CT>::T* t, int elems) {
    for (int i = threadIdx.x; i < elems; i += NT) t[i] = (typename RT<CT>::T)0.f;
}

// Workgroup -> (slab, part): part-major so that the P workgroups of a slab are nslabs apart in blockIdx (under the
// observed round-robin dispatch over the 8 XCDs they share an XCD when nslabs % 8 == 0: speed only, never correctness).
template <int CT>
__global__ __launch_bounds__(NT) void rollout_fwd_kernel(RollParams p) {
    typedef typename RT<CT>::T T;
    constexpr int KS = RT<CT>::KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Lds L = carve<CT>(smem, p.C, p.H, p.nb);
    T* a_c = (T*)L.a_c; T* a_h1 = (T*)L.a_h1; T* a_h2 = (T*)L.a_h2;
    const int B = p.B, C = p.C, H = p.H, n = p.n, nb = p.nb, P = p.P;
    const int nslabs = (B + 15) / 16;
    const int slab = blockIdx.x % nslabs, part = blockIdx.x / nslabs;
    const int row0 = slab * 16;
    const int Hs = H / P, c_lo = part * Hs;                 // this workgroup's slice of the hidden dimension
    const int tilesH = (H + 15) / 16, tilesC = (C + 15) / 16, tilesS = (Hs + 15) / 16;
    const int stepsC = (C + KS - 1) / KS, stepsH = (H + KS - 1) / KS, stepsS = (Hs + KS - 1) / KS;
    Exchange X{p.xbuf, p.xerr, nslabs, P, L.Cf, slab, part};

    // layer descriptors
    const LayerArgs A1{L.Ck, 0, 0, stepsC, stepsC, 0, tilesH, L.NpH, ksplit_for(tilesH, stepsC)};                     // h1 (all columns)
    const LayerArgs A2{L.Hk, 0, 0, stepsH, stepsH, P > 1 ? c_lo / 16 : 0, P > 1 ? tilesS : tilesH, L.NpH,
                       ksplit_for(P > 1 ? tilesS : tilesH, stepsH)};                                                   // h2 (own slice)
    const LayerArgs A3{L.Hk, P > 1 ? c_lo / KS : 0, P > 1 ? c_lo / KS : 0, P > 1 ? stepsS : stepsH, stepsH, 0, tilesC, L.NpC,
                       ksplit_for(tilesC, P > 1 ? stepsS : stepsH)};                                                   // partial residual

    zero_tile<CT>(a_c, 16 * L.Ck); zero_tile<CT>(a_h1, 16 * L.Hk); zero_tile<CT>(a_h2, 16 * L.Hk);
    for (int b = 0; b < nb; ++b) {
        float* bb = L.bias + b * (2 * H + C);
        for (int i = threadIdx.x; i < H; i += NT) { bb[i] = p.bias[3 * b][i]; bb[H + i] = p.bias[3 * b + 1][i]; }
        for (int i = threadIdx.x; i < C; i += NT) bb[2 * H + i] = p.bias[3 * b + 2][i];
    }
    const int er = threadIdx.x >> 5, ec = threadIdx.x & 31;           // epilogue mapping: 32 threads per slab row
    const bool vrow = row0 + er < B;
    for (int c = ec; c < C; c += 32) {
        const float v = vrow ? p.x0[(int64_t)(row0 + er) * C + c] : 0.f;
        L.xs[er * L.Cf + c] = v;
        if (vrow && part == 0) p.t_codes[((int64_t)(row0 + er) * n) * C + c] = v;
    }
    __syncthreads();

    unsigned epoch = 0;
    for (int t = 1; t < n; ++t) {
        for (int b = 0; b < nb; ++b) {
            const T* W1 = (const T*)p.W[3 * b]; const T* W2 = (const T*)p.W[3 * b + 1]; const T* W3 = (const T*)p.W[3 * b + 2];
            const float* b1 = L.bias + b * (2 * H + C); const float* b2 = b1 + H; const float* b3 = b2 + H;
            const int64_t sbase = ((int64_t)b * (n - 1) + (t - 1)) * B + row0;     // row index into the [nb][n-1][B][.] saves
            StagedSmall<CT> S1, S3;
            StagedStream<CT> S2;
            S1.issue(W1, A1);
            for (int c = ec; c < C; c += 32) {                                      // block input -> MFMA operand (+ dW1 save)
                const T v = (T)L.xs[er * L.Cf + c];
                a_c[er * L.Ck + c] = v;
                if (vrow && part == 0) ((T*)p.xin_save)[(sbase + er) * C + c] = v;
            }
            __syncthreads();
            S1.run(a_c, W1, L.part, A1);                                            // h1: every workgroup computes all of it
            S2.issue(W2, A2);                                                       // weights of the next layer fly over the epilogue
            __syncthreads();
            {
                unsigned bits = 0u;
                for (int c = ec; c < H; c += 32) {
                    float v = part_sum(L.part, L.NpH, A1.ksplit, er, c) + b1[c];
                    v = v > 0.f ? v : 0.f;
                    const T hv = (T)v;
                    a_h1[er * L.Hk + c] = hv;
                    if (c >= c_lo && c < c_lo + Hs) {                               // own slice: save + sign bits
                        const int u = (c - c_lo - ec) >> 5;
                        if ((float)hv > 0.f && u < 32) bits |= 1u << u;
                        if (vrow) ((T*)p.h1_save)[(sbase + er) * H + c] = hv;
                    }
                }
                if (vrow) p.m1_save[((sbase + er) * P + part) * 32 + ec] = bits;
            }
            __syncthreads();
            S2.run(a_h1, W2, L.part, A2);                                           // h2: own column slice (weight stream / P)
            S3.issue(W3, A3);
            __syncthreads();
            {
                unsigned bits = 0u;
                int u = 0;
                for (int cl = ec; cl < Hs; cl += 32, ++u) {
                    float v = part_sum(L.part, L.NpH, A2.ksplit, er, cl) + b2[c_lo + cl];
                    v = v > 0.f ? v : 0.f;
                    const T hv = (T)v;
                    if ((float)hv > 0.f && u < 32) bits |= 1u << u;
                    a_h2[er * L.Hk + c_lo + cl] = hv;
                    if (vrow) ((T*)p.h2_save)[(sbase + er) * H + c_lo + cl] = hv;
                }
                if (vrow) p.m2_save[((sbase + er) * P + part) * 32 + ec] = bits;
            }
            __syncthreads();
            S3.run(a_h2, W3, L.part, A3);                                           // partial residual over the own K slice
            __syncthreads();
            ++epoch;
            for (int c = ec; c < C; c += 32) {
                float res = part_sum(L.part, L.NpC, A3.ksplit, er, c);
                if (P > 1) res = exchange_sum(X, epoch, er * L.Cf + c, res);
                res += b3[c];
                L.xs[er * L.Cf + c] += res;
                if (p.residuals && vrow && part == 0) p.residuals[(((int64_t)(t - 1) * nb + b) * B + row0 + er) * C + c] = res;
            }
            __syncthreads();
        }
        if (part == 0)
            for (int c = ec; c < C; c += 32)
                if (vrow) p.t_codes[((int64_t)(row0 + er) * n + t) * C + c] = L.xs[er * L.Cf + c];
    }
}

template <int CT>
__global__ __launch_bounds__(NT) void rollout_bwd_kernel(RollParams p) {
    typedef typename RT<CT>::T T;
    constexpr int KS = RT<CT>::KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Lds L = carve<CT>(smem, p.C, p.H, p.nb);
    T* a_c = (T*)L.a_c; T* a_h1 = (T*)L.a_h1; T* a_h2 = (T*)L.a_h2;
    const int B = p.B, C = p.C, H = p.H, n = p.n, nb = p.nb, P = p.P;
    const int nslabs = (B + 15) / 16;
    const int slab = blockIdx.x % nslabs, part = blockIdx.x / nslabs;
    const int row0 = slab * 16;
    const int Hs = H / P, c_lo = part * Hs;
    const int tilesH = (H + 15) / 16, tilesC = (C + 15) / 16, tilesS = (Hs + 15) / 16;
    const int stepsC = (C + KS - 1) / KS, stepsH = (H + KS - 1) / KS, stepsS = (Hs + KS - 1) / KS;
    Exchange X{p.xbuf, p.xerr, nslabs, P, L.Cf, slab, part};

    const LayerArgs A3{L.Ck, 0, 0, stepsC, stepsC, 0, tilesH, L.NpH, ksplit_for(tilesH, stepsC)};                     // dh2 (all columns)
    const LayerArgs A2{L.Hk, 0, 0, stepsH, stepsH, P > 1 ? c_lo / 16 : 0, P > 1 ? tilesS : tilesH, L.NpH,
                       ksplit_for(P > 1 ? tilesS : tilesH, stepsH)};                                                   // dh1 (own slice)
    const LayerArgs A1{L.Hk, P > 1 ? c_lo / KS : 0, P > 1 ? c_lo / KS : 0, P > 1 ? stepsS : stepsH, stepsH, 0, tilesC, L.NpC,
                       ksplit_for(tilesC, P > 1 ? stepsS : stepsH)};                                                   // partial dx

    zero_tile<CT>(a_c, 16 * L.Ck); zero_tile<CT>(a_h1, 16 * L.Hk); zero_tile<CT>(a_h2, 16 * L.Hk);
    for (int i = threadIdx.x; i < 16 * L.Cf; i += NT) L.xs[i] = 0.f;
    const int er = threadIdx.x >> 5, ec = threadIdx.x & 31;
    const bool vrow = row0 + er < B;
    __syncthreads();

    unsigned epoch = 0;
    for (int t = n - 1; t >= 1; --t) {
        for (int c = ec; c < C; c += 32)
            if (vrow) L.xs[er * L.Cf + c] += p.g[((int64_t)(row0 + er) * n + t) * C + c];
        __syncthreads();
        for (int b = nb - 1; b >= 0; --b) {
            const T* W3T = (const T*)p.W[3 * b]; const T* W2T = (const T*)p.W[3 * b + 1]; const T* W1T = (const T*)p.W[3 * b + 2];
            const int64_t sbase = ((int64_t)b * (n - 1) + (t - 1)) * B + row0;
            // sign bits: h2 for ALL columns (word q of part q), h1 for the own slice; issued before the GEMMs that hide them
            unsigned w2[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) w2[q] = (vrow && q < P) ? p.m2_save[((sbase + er) * P + q) * 32 + ec] : 0u;
            const unsigned w1 = vrow ? p.m1_save[((sbase + er) * P + part) * 32 + ec] : 0u;
            StagedSmall<CT> S3, S1;
            StagedStream<CT> S2;
            S3.issue(W3T, A3);
            // dr = gradient wrt the residual = running gradient (rounded to the compute type for the MFMA and dW3)
            for (int c = ec; c < C; c += 32) {
                const T v = (T)L.xs[er * L.Cf + c];
                a_c[er * L.Ck + c] = v;
                if (vrow && part == 0) ((T*)p.dr_save)[(sbase + er) * C + c] = v;
            }
            __syncthreads();
            S3.run(a_c, W3T, L.part, A3);                                           // dh2 = (dr W3) * relu'(h2): all columns
            S2.issue(W2T, A2);
            __syncthreads();
            for (int c = ec; c < H; c += 32) {
                const int q = c / Hs, u = (c - q * Hs - ec) >> 5;
                bool on;
                if (u < 32) {
                    unsigned word = w2[0];
#pragma unroll
                    for (int qq = 1; qq < 8; ++qq) word = (q == qq) ? w2[qq] : word;
                    on = ((word >> u) & 1u) != 0u;
                } else {
                    on = vrow && (float)((const T*)p.h2_save)[(sbase + er) * H + c] > 0.f;
                }
                const T dv = (T)(on ? part_sum(L.part, L.NpH, A3.ksplit, er, c) : 0.f);
                a_h2[er * L.Hk + c] = dv;
                if (vrow && q == part) ((T*)p.dh2_save)[(sbase + er) * H + c] = dv;
            }
            __syncthreads();
            S2.run(a_h2, W2T, L.part, A2);                                          // dh1 = (dh2 W2) * relu'(h1): own slice
            S1.issue(W1T, A1);
            __syncthreads();
            {
                int u = 0;
                for (int cl = ec; cl < Hs; cl += 32, ++u) {
                    const bool on = u < 32 ? ((w1 >> u) & 1u) != 0u
                                           : (vrow && (float)((const T*)p.h1_save)[(sbase + er) * H + c_lo + cl] > 0.f);
                    const T dv = (T)(on ? part_sum(L.part, L.NpH, A2.ksplit, er, cl) : 0.f);
                    a_h1[er * L.Hk + c_lo + cl] = dv;
                    if (vrow) ((T*)p.dh1_save)[(sbase + er) * H + c_lo + cl] = dv;
                }
            }
            __syncthreads();
            S1.run(a_h1, W1T, L.part, A1);                                          // partial dx_in over the own K slice
            __syncthreads();
            ++epoch;
            for (int c = ec; c < C; c += 32) {
                float v = part_sum(L.part, L.NpC, A1.ksplit, er, c);
                if (P > 1) v = exchange_sum(X, epoch, er * L.Cf + c, v);
                L.xs[er * L.Cf + c] += v;
            }
            __syncthreads();
        }
    }
    if (part == 0)
        for (int c = ec; c < C; c += 32)
            if (vrow) p.dx0[(int64_t)(row0 + er) * C + c] = L.xs[er * L.Cf + c] + p.g[((int64_t)(row0 + er) * n) * C + c];
}

int pick_parts(int compute, int B, int C, int H) {
    // split the hidden dimension over P workgroups per slab when slices stay MFMA/k-step aligned and LDS-friendly
    const int KS = compute == VS_BF16 ? 32 : 16;
    const int nslabs = (B + 15) / 16;
    int P = 1;
    for (int cand = 2; cand <= 8; cand *= 2) {
        if (H % cand) break;
        const int Hs = H / cand;
        if (Hs % KS || Hs % 16 || Hs < 64) break;
        if (nslabs * cand > 128) break;                 // keep every workgroup resident with room to spare (256 CUs)
        P = cand;
    }
    if (P > 4) P = 4;
    return P;
}

template <int CT>
int launch_roll(bool fwd, const RollParams& p, hipStream_t stream) {
    int Cf, Ck, Hk, NpH, NpC;
    size_t oc, o1, o2, ob, op;
    const size_t smem = lds_layout(p.C, p.H, p.nb, (int)sizeof(typename RT<CT>::T), RT<CT>::KS, RT<CT>::U, &Cf, &Ck, &Hk, &NpH, &NpC, &oc, &o1,
                                   &o2, &ob, &op);
    if (smem > 160 * 1024) return vs_fail(VS_ERR_UNSUPPORTED, "vs_mlp_rollout: C=%d H=%d needs %zu B of LDS (> 160 KiB)", p.C, p.H, smem);
    const void* kfn = fwd ? (const void*)rollout_fwd_kernel<CT> : (const void*)rollout_bwd_kernel<CT>;
    if (smem > 64 * 1024) {
        if (hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return vs_fail(VS_ERR_LAUNCH, "vs_mlp_rollout: cannot raise dynamic LDS limit to %zu", smem);
    }
    const int nslabs = (p.B + 15) / 16;
    if (p.P > 1) {
        const size_t xbytes = (size_t)2 * nslabs * p.P * 16 * Cf * sizeof(u64);
        if (vs_zero_async(p.xbuf, xbytes + 16, stream) != hipSuccess) return vs_fail(VS_ERR_LAUNCH, "vs_mlp_rollout: memset failed");
    }
    dim3 grid((unsigned)(nslabs * p.P));
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

int setup_exchange(RollParams& p, int compute, void* workspace, size_t workspace_bytes) {
    p.P = pick_parts(compute, p.B, p.C, p.H);
    const int nslabs = (p.B + 15) / 16, Cf = round_up(p.C, 4);
    const size_t need = (size_t)2 * nslabs * p.P * 16 * Cf * sizeof(u64) + 16;
    if (p.P > 1 && (!workspace || workspace_bytes < need)) p.P = 1;         // no exchange area: run unsplit
    p.xbuf = (u64*)workspace;
    p.xerr = p.P > 1 ? (unsigned*)((char*)workspace + need - 16) : nullptr;
    return VS_OK;
}

}  // namespace

extern "C" int vs_mlp_rollout_parts(int compute, int B, int C, int H) { return pick_parts(compute, B, C, H); }

extern "C" size_t vs_mlp_rollout_workspace_bytes(int compute, int B, int C, int H) {
    const int P = pick_parts(compute, B, C, H);
    if (P <= 1) return 0;
    return (size_t)2 * ((B + 15) / 16) * P * 16 * round_up(C, 4) * sizeof(u64) + 16;
}

extern "C" int vs_mlp_rollout_fwd(int compute, int B, int C, int H, int n_blocks, int n_steps, const float* x0,
                                  const void* const* weights, const float* const* biases, float* t_codes, float* residuals,
                                  void* xin_save, void* h1_save, void* h2_save, uint32_t* m1_save, uint32_t* m2_save, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    int rc = check_common(compute, B, C, H, n_blocks, n_steps);
    if (rc != VS_OK) return rc;
    VS_CHECK_ARG(x0 && weights && biases && t_codes && xin_save && h1_save && h2_save && m1_save && m2_save,
                 "vs_mlp_rollout_fwd: null pointer");
    RollParams p = {};
    p.B = B; p.C = C; p.H = H; p.nb = n_blocks; p.n = n_steps;
    for (int i = 0; i < 3 * n_blocks; ++i) { p.W[i] = weights[i]; p.bias[i] = biases[i]; }
    p.x0 = x0; p.t_codes = t_codes; p.residuals = residuals;
    p.xin_save = xin_save; p.h1_save = h1_save; p.h2_save = h2_save; p.m1_save = m1_save; p.m2_save = m2_save;
    setup_exchange(p, compute, workspace, workspace_bytes);
    return compute == VS_BF16 ? launch_roll<VS_BF16>(true, p, (hipStream_t)stream) : launch_roll<VS_F32>(true, p, (hipStream_t)stream);
}

extern "C" int vs_mlp_rollout_bwd(int compute, int B, int C, int H, int n_blocks, int n_steps, const float* grad_t_codes,
                                  const void* const* weights_t, const void* h1_save, const void* h2_save,
                                  const uint32_t* m1_save, const uint32_t* m2_save, float* dx0, void* dr_save, void* dh2_save,
                                  void* dh1_save, void* workspace, size_t workspace_bytes, void* stream) {
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
    setup_exchange(p, compute, workspace, workspace_bytes);
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
