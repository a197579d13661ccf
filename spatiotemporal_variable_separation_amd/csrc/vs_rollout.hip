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
#include <stdlib.h>

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
    unsigned* xerr;              // STICKY timeout flag (last 16 bytes of the workspace: never zeroed by the library, the caller
                                 // reads and clears it -- ops.rollout_exchange_error / train.check_rollout_exchange)
    size_t xtotal;               // workspace bytes (all but the last 16 are zero-filled before every launch)
    unsigned spin_limit;         // rounds a wait may take before it gives up (VS_ROLLOUT_SPIN_LIMIT, default 2^22)
    int nap;                     // weight-stationary form: s_sleep units (64 clocks) between seeing the producer's input and polling its partials
    unsigned* ebase;             // weight-stationary form: per-slab epoch base words (even; advanced by the slab's last workgroup of every launch)
    int ring_skew;               // weight-stationary form, TEST AID (VS_ROLLOUT_XCD_LOCAL=2): deal the slabs out so that the ring of a slab is
                                 // spread over the XCDs -- the placement the XCD-local exchange must never meet
    // backward
    const float* g;              // [B, n, C] gradient wrt every t_code
    float* dx0;                  // [B, C]
    void* dr_save;               // [nb, n-1, B, C]
    void* dh2_save;              // [nb, n-1, B, H]
    void* dh1_save;              // [nb, n-1, B, H]
};

template <int CT> struct RT;
template <> struct RT<VS_BF16> { typedef __bf16 T; static constexpr int KS = 32; static constexpr int U = 8; };
template <> struct RT<VS_F16> { typedef _Float16 T; static constexpr int KS = 32; static constexpr int U = 8; };
template <> struct RT<VS_F32> { typedef float T; static constexpr int KS = 16; static constexpr int U = 4; };

__host__ __device__ inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// partial[s][16][Np] (fp32) = in[16, K(chunk s)] * W[N, K]^T ; `in` is an LDS tile [16][KP] zero padded to a
// multiple of KS; W is global [N][K] row-major.  All NW waves cooperate: item = (n-tile, k-chunk).
template <int CT>
__device__ __forceinline__ f32x4 mma16(const u32x4& av, const u32x4& bv, f32x4 acc) {
    if constexpr (CT == VS_BF16) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&av), *reinterpret_cast<const bf16x8*>(&bv), acc,
                                                       0, 0, 0);
    } else if constexpr (CT == VS_F16) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(&av), *reinterpret_cast<const f16x8*>(&bv), acc, 0, 0, 0);
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
    unsigned spin_limit;
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
                if (++spins > x.spin_limit) { atomicOr(x.err, 1u); break; }
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

__host__ __device__ inline int ksplit_for(int ntiles, int nsteps);

__host__ __device__ inline size_t lds_layout(int C, int H, int nb, int P, int esize, int KS, int U, int* Cf, int* Ck, int* Hk, int* NpH, int* NpC,
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
    // partials: ksplit slabs of [16][Np] for each of the three layer shapes a workgroup runs (same set in both directions)
    const int Hs = H / P;
    const int tilesH = (H + 15) / 16, tilesC = (C + 15) / 16, tilesS = (Hs + 15) / 16;
    const int stepsC = (C + KS - 1) / KS, stepsH = (H + KS - 1) / KS, stepsS = (Hs + KS - 1) / KS;
    const size_t p1 = (size_t)ksplit_for(tilesH, stepsC) * 16 * (*NpH) * 4;
    const size_t p2 = (size_t)ksplit_for(P > 1 ? tilesS : tilesH, stepsH) * 16 * (*NpH) * 4;
    const size_t p3 = (size_t)ksplit_for(tilesC, P > 1 ? stepsS : stepsH) * 16 * (*NpC) * 4;
    size_t pm = p1 > p2 ? p1 : p2;
    if (p3 > pm) pm = p3;
    off += pm;
    return off;
}

template <int CT>
__device__ __forceinline__ Lds carve(char* smem, int C, int H, int nb, int P) {
    Lds L;
    size_t oc, o1, o2, ob, op;
    lds_layout(C, H, nb, P, (int)sizeof(typename RT<CT>::T), RT<CT>::KS, RT<CT>::U, &L.Cf, &L.Ck, &L.Hk, &L.NpH, &L.NpC, &oc, &o1, &o2, &ob, &op);
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
    const Lds L = carve<CT>(smem, p.C, p.H, p.nb, p.P);
    T* a_c = (T*)L.a_c; T* a_h1 = (T*)L.a_h1; T* a_h2 = (T*)L.a_h2;
    const int B = p.B, C = p.C, H = p.H, n = p.n, nb = p.nb, P = p.P;
    const int nslabs = (B + 15) / 16;
    const int slab = blockIdx.x % nslabs, part = blockIdx.x / nslabs;
    const int row0 = slab * 16;
    const int Hs = H / P, c_lo = part * Hs;                 // this workgroup's slice of the hidden dimension
    const int tilesH = (H + 15) / 16, tilesC = (C + 15) / 16, tilesS = (Hs + 15) / 16;
    const int stepsC = (C + KS - 1) / KS, stepsH = (H + KS - 1) / KS, stepsS = (Hs + KS - 1) / KS;
    Exchange X{p.xbuf, p.xerr, nslabs, P, L.Cf, slab, part, p.spin_limit};

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
            const int64_t mbase = ((int64_t)b * (n - 1) + (t - 1)) * (nslabs * 16) + row0;   // sign-bit arrays have rows padded to 16
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
                if (vrow) p.m1_save[((mbase + er) * P + part) * 32 + ec] = bits;
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
                if (vrow) p.m2_save[((mbase + er) * P + part) * 32 + ec] = bits;
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
    const Lds L = carve<CT>(smem, p.C, p.H, p.nb, p.P);
    T* a_c = (T*)L.a_c; T* a_h1 = (T*)L.a_h1; T* a_h2 = (T*)L.a_h2;
    const int B = p.B, C = p.C, H = p.H, n = p.n, nb = p.nb, P = p.P;
    const int nslabs = (B + 15) / 16;
    const int slab = blockIdx.x % nslabs, part = blockIdx.x / nslabs;
    const int row0 = slab * 16;
    const int Hs = H / P, c_lo = part * Hs;
    const int tilesH = (H + 15) / 16, tilesC = (C + 15) / 16, tilesS = (Hs + 15) / 16;
    const int stepsC = (C + KS - 1) / KS, stepsH = (H + KS - 1) / KS, stepsS = (Hs + KS - 1) / KS;
    Exchange X{p.xbuf, p.xerr, nslabs, P, L.Cf, slab, part, p.spin_limit};

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
            const int64_t mbase = ((int64_t)b * (n - 1) + (t - 1)) * (nslabs * 16) + row0;
            // sign bits: h2 for ALL columns (word q of part q), h1 for the own slice; issued before the GEMMs that hide them
            unsigned w2[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) w2[q] = (vrow && q < P) ? p.m2_save[((mbase + er) * P + q) * 32 + ec] : 0u;
            const unsigned w1 = vrow ? p.m1_save[((mbase + er) * P + part) * 32 + ec] : 0u;
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

// =====================================================================================================================
// Weight-stationary pipelined form (bf16, C <= 32, H in {128, 256, 512}).
//
// The slab-per-workgroup form above re-streams every block's weights from L2 at every time step and pays ~7 barriers and
// LDS round trips per block-step (7.7 us fwd / 10.7 us bwd at WaveEq size) although the MFMA work of a block-step is a
// few hundred cycles.  Here every (slab, block, part) triple gets its OWN workgroup whose weight fragments are loaded
// into registers ONCE (H = 512: 104 VGPRs per lane) and never move again: a block-step is then three short MFMA bursts
// out of registers, three barriers, and one hop of the granule exchange to the workgroups of the next block.  The
// workgroups of one slab form a ring of nb groups x P parts; group b sleeps while the other blocks work (the chip has
// 256 CUs and the recurrence can use at most nslabs * P of them at a time anyway).
//
//   part p of block b owns hidden columns [64 p, 64 p + 64):  P = H / 64, 4 waves, wave w owns the 16-column tile w of it
//   layer a (code -> hidden, N = H, K = C): computed in full by every part (K is one k-step), wave w: tiles w, w+4, ...
//   layer b (hidden -> hidden slice, N = 64, K = H): wave w: tile 4p + w, KH k-steps, two accumulators
//   layer c (hidden slice -> code, N = C, K = 64): waves 0/1: code tile w, two chained k-steps -> partial [16, 32]
//   exchange: epoch e = block-step index + 1; slots [2][nslabs][P + 1][16 x 32] granules {epoch, fp32}: P partials plus the
//   block input x (published by part 0 as soon as it is known).  The consumer (every part of the next block) computes
//   x_out = x_in + ((p_0 + ... + p_{P-1}) + b3) in that fixed order: all parts hold bitwise identical codes.
//   Double buffering is sufficient for the same reason as above (a group publishes e + 2 only after receiving all of e + 1,
//   whose producers had finished reading e).
// Backward-through-time is the same skeleton with the transposed weights, ReLU masks instead of bias + ReLU and the
// block-steps walked in reverse.  Sign bits are stored one bit per hidden column, row-major (16-bit unit per 16-column tile).
namespace wsr {

constexpr int WT = 256;
constexpr int XP = 40;            // LDS row pitch of the code operand (32 + 8: conflict-free ds_read_b128)
constexpr int SP = 72;            // LDS row pitch of the hidden-slice operand (64 + 8)
#ifndef WS_NAP
#define WS_NAP 8                  // s_sleep units (64 clocks) between seeing the producer's input and polling its partials
#endif

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

__device__ __forceinline__ u64 gload(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// XL = false: agent-scope store (sc1: written through to the fabric, visible to every XCD).  XL = true: the ring of a slab lives on ONE XCD
// (slab = blockIdx % nslabs with nslabs % 8 == 0 under the round-robin dispatch over the 8 XCDs), so a granule only has to reach that XCD's
// L2, where the consumers' sc1 loads find it: a plain store.  Measured (tools/probes/xcd_pingpong.hip): 210 ns per hop against 385 ns for
// sc1 / sc1 -- and NOT visible from another XCD until the line is evicted.  The dispatcher's round robin may start a launch on any XCD (blocks
// are NOT pinned to XCD blockIdx % 8), but blocks 8 apart meet on one XCD; should that ever not hold, a consumer keeps polling an old epoch
// and runs into the spin limit (the sticky error flag) -- a granule carries its epoch, so a stale line can never pass for the new value.
template <bool XL>
__device__ __forceinline__ void gstore(u64* p, u64 v) {
    if constexpr (XL) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Two adjacent granules per load.  A relaxed agent-scope 8-byte atomic load is `global_load_dwordx2 ... sc1`; the exchange is
// bound by the per-CU load issue rate, and a 16-byte `global_load_dwordx4 ... sc1` moves two granules per lane and issue slot
// (each 8-byte half is still read untorn -- every granule carries its own tag, both are checked).  All loads of a round are
// issued back to back inside one asm statement, which also waits for them (the compiler cannot track asm-issued loads).
template <int N> struct GranulePairs;
template <> struct GranulePairs<3> {
    static __device__ __forceinline__ void load(u32x4* v, const u64* const* a) {
        asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\tglobal_load_dwordx4 %2, %5, off sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2])
                     : "memory");
    }
};
template <> struct GranulePairs<5> {
    static __device__ __forceinline__ void load(u32x4* v, const u64* const* a) {
        asm volatile("global_load_dwordx4 %0, %5, off sc1\n\tglobal_load_dwordx4 %1, %6, off sc1\n\tglobal_load_dwordx4 %2, %7, off sc1\n\t"
                     "global_load_dwordx4 %3, %8, off sc1\n\tglobal_load_dwordx4 %4, %9, off sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4])
                     : "memory");
    }
};
template <> struct GranulePairs<9> {
    static __device__ __forceinline__ void load(u32x4* v, const u64* const* a) {
        asm volatile("global_load_dwordx4 %0, %9, off sc1\n\tglobal_load_dwordx4 %1, %10, off sc1\n\tglobal_load_dwordx4 %2, %11, off sc1\n\t"
                     "global_load_dwordx4 %3, %12, off sc1\n\tglobal_load_dwordx4 %4, %13, off sc1\n\tglobal_load_dwordx4 %5, %14, off sc1\n\t"
                     "global_load_dwordx4 %6, %15, off sc1\n\tglobal_load_dwordx4 %7, %16, off sc1\n\tglobal_load_dwordx4 %8, %17, off sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8])
                     : "memory");
    }
};
// LDS-only workgroup barrier: __syncthreads() also drains the vector-memory counter (its release fence waits for every
// outstanding global store -- the activation saves and the exchange granules -- to be acknowledged by L2); the barriers of a
// block-step only order LDS traffic, so they wait for the LDS counter alone.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int CT>
__device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, f32x4 acc) {
    if constexpr (CT == VS_BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), acc, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8*>(&a), *reinterpret_cast<const f16x8*>(&b), acc, 0, 0, 0);
}
// two fp32 -> one dword of two bf16 / fp16 (round to nearest even, hardware convert)
template <int CT>
__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
    if constexpr (CT == VS_BF16) {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        bf16x2 v = {(__bf16)lo, (__bf16)hi};
        return *reinterpret_cast<unsigned*>(&v);
    } else {
        typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
        f16x2 v = {(_Float16)lo, (_Float16)hi};
        return *reinterpret_cast<unsigned*>(&v);
    }
}
// ReLU on two packed bf16 / fp16: as signed 16-bit integers negative floats are negative, so max(x, 0) clears them
__device__ __forceinline__ unsigned pk_relu(unsigned v) {
    unsigned r;
    asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(v));
    return r;
}
// x where the lane's bit of a stored v_cmp result is set, else 0.  `half` is the 32-bit half of the 64-lane mask that holds
// this lane's bit, `sh` = 31 - (lane & 31) moves that bit to the sign position.
__device__ __forceinline__ float keep_if(float x, unsigned half, int sh) { return (int)(half << sh) < 0 ? x : 0.f; }
// sign bits of one 16-feature tile: words [0..3] = low halves of the four v_cmp masks (r = 0..3), [4..7] = high halves, so a
// lane fetches its four bits with one 16-byte load
__device__ __forceinline__ void store_tile_masks(unsigned* dst, const u64* bal) {
    *reinterpret_cast<u32x4*>(dst) = u32x4{(unsigned)bal[0], (unsigned)bal[1], (unsigned)bal[2], (unsigned)bal[3]};
    *reinterpret_cast<u32x4*>(dst + 4) = u32x4{(unsigned)(bal[0] >> 32), (unsigned)(bal[1] >> 32), (unsigned)(bal[2] >> 32), (unsigned)(bal[3] >> 32)};
}

#ifdef VS_WS_TIMING
#define WS_STAMP(k) do { if (dbg && slab == 0 && part == 0 && tid == 0) dbg[((int64_t)q * 8 + (k))] = wall_clock64(); } while (0)
#else
#define WS_STAMP(k) do { } while (0)
#endif

// Transposed chain: every layer is computed as (weights) x (activations)^T, so an MFMA result -- lane (c, g) holds 4
// consecutive FEATURES 4g..4g+3 of batch row c -- is, after the bf16 convert, already the B-operand fragment of the next
// layer (8 k-values per lane = the results of two feature tiles; the k order inside a k-step is a fixed permutation that the
// next layer's weight fragments are loaded with).  Layer a -> layer b therefore never touches LDS: wave w computes the
// feature tiles T = 4 j + w of layer a and contracts exactly those features in layer b (K split over the 4 waves, all 64
// output features of the part); the four K-partials meet in LDS (16 KB, one barrier), wave w finishes output tile w.
template <int KH, bool FWD, int CT, bool XL = false>
__global__ __launch_bounds__(WT) void rollout_ws_kernel(RollParams p, int mask_pitch /* words per row of m1/m2 */, long long* dbg) {
    typedef typename RT<CT>::T T;
    constexpr int H = 32 * KH, P = KH / 2, NJ = KH / 2, NKB = NJ / 2 > 0 ? NJ / 2 : 1;
    static_assert(NJ % 2 == 0, "two layer-a tiles form one k-step of layer b");
    __shared__ __attribute__((aligned(16))) T xa[16 * XP];
    __shared__ __attribute__((aligned(16))) T ah2[16 * SP];
    __shared__ __attribute__((aligned(16))) float bias_a[H];
    __shared__ __attribute__((aligned(16))) f32x4 pb[4 * 4 * 64];          // layer b K-partials: [wave][output tile][lane]

    const int B = p.B, C = p.C, n = p.n, nb = p.nb;
    const int nslabs = (B + 15) / 16, Bp = nslabs * 16;
    int id = blockIdx.x;
    const int slab = p.ring_skew ? (id + id / nslabs) % nslabs : id % nslabs;     // (skew: a bijection inside every group of nslabs ids)
    id /= nslabs;
    const int part = id % P;
    const int blk = id / P;
    const int pos = FWD ? blk : nb - 1 - blk;                 // position of this block inside a time step, in execution order
    const int row0 = slab * 16;
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (told to the compiler: scalar mask loads)
    const int er = tid >> 4, ec = 2 * (tid & 15);             // exchange mapping: the adjacent elements (er, ec) and (er, ec + 1)
    const bool erow = row0 + er < B;
    const int own = 4 * part + w;                             // the 16-feature tile of the hidden dimension this wave finishes / saves
    const bool vrow = row0 + c < B;                           // MFMA results: lane (c, g) belongs to batch row c

    // ---- weights: loaded once, resident in registers for the whole rollout ------------------------------------------
    const T* Wa = (const T*)p.W[3 * blk];
    const T* Wb = (const T*)p.W[3 * blk + 1];
    const T* Wc = (const T*)p.W[3 * blk + 2];
    u32x4 wa[NJ], wa_own, wb[4][NKB], wc[2];
#pragma unroll
    for (int j = 0; j < NJ; ++j) wa[j] = *reinterpret_cast<const u32x4*>(Wa + ((int64_t)(4 * j + w) * 64 + lane) * 8);
    wa_own = *reinterpret_cast<const u32x4*>(Wa + ((int64_t)own * 64 + lane) * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int jj = 0; jj < NKB; ++jj) {
            // k-step jj of this wave = layer-a tiles 4 (2 jj) + w and 4 (2 jj + 1) + w: lane (c, g) contracts the features
            // 16 (8 jj + w) + 4 g + {0..3} and the same + 64.  In the standard pack they sit in k-step 4 jj + w / 2 (resp. + 2),
            // lane group 2 (w & 1) + g / 2, 8-byte half g & 1.
            const T* piece = Wb + (((int64_t)(4 * part + i) * KH + 4 * jj + (w >> 1)) * 64 + 16 * (2 * (w & 1) + (g >> 1)) + c) * 8 + (g & 1) * 4;
            const u32x2 lo = *reinterpret_cast<const u32x2*>(piece);
            const u32x2 hi = *reinterpret_cast<const u32x2*>(piece + 2 * 64 * 8);
            wb[i][jj] = u32x4{lo.x, lo.y, hi.x, hi.y};
        }
    }
    const bool code_tile = (w & 1) * 16 < C;                  // C <= 16: the packed code layer has one column tile, the other is zero
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        wc[s] = u32x4{0u, 0u, 0u, 0u};
        if (code_tile) wc[s] = *reinterpret_cast<const u32x4*>(Wc + (((int64_t)(w & 1) * KH + 2 * part + s) * 64 + lane) * 8);
    }
    f32x4 bias_b = {0.f, 0.f, 0.f, 0.f};
    float bc[2] = {0.f, 0.f};
    if (FWD) {
        for (int i = tid; i < H; i += WT) bias_a[i] = p.bias[3 * blk][i];
#pragma unroll
        for (int r = 0; r < 4; ++r) bias_b[r] = p.bias[3 * blk + 1][16 * own + 4 * g + r];
        const int prev = (blk + nb - 1) % nb;                  // the block whose output this workgroup consumes
#pragma unroll
        for (int i = 0; i < 2; ++i) bc[i] = ec + i < C ? p.bias[3 * prev + 2][ec + i] : 0.f;
    }
    __syncthreads();

    // Epochs of this launch = ebase + block-step index: the slab's base word is even, was left by the previous launch (forward, backward,
    // or the previous replay) above every tag that launch used, and is advanced again below -- so the exchange area never has to be
    // cleared between launches (a 5 us fill launch in front of each of the two integrator kernels of a step).
    const unsigned ebase = p.ebase[slab];
    u64* const xbase = p.xbuf;
    auto slot = [&](unsigned epoch, int who) -> u64* {
        return xbase + ((((int64_t)(epoch & 1u) * nslabs + slab) * (P + 1) + who) << 9);
    };

    const int iters = (n - 1) + (pos == 0 ? 1 : 0);            // position 0 also receives the very last output
    for (int it = 0; it < iters; ++it) {
        const int q = it * nb + pos;                            // block-step index in execution order
        const bool extra = it == n - 1;
        const int t = FWD ? it + 1 : n - 1 - it;                // time step of this block-step
        const int64_t sbase = ((int64_t)blk * (n - 1) + (t - 1)) * B + row0;
        // sign bits of this (block, step, slab): lane masks (v_cmp results: bit 16 g + c <-> feature 4 g + r of the tile, batch
        // row c), 32 bytes per tile (store_tile_masks); the tile T = 4 j + w is stored at index NJ w + j: a wave reads one contiguous run
        const int64_t mofs = (((int64_t)blk * (n - 1) + (t - 1)) * Bp + row0) * mask_pitch;
        const int mown = (NJ * w + part) * 8;                   // word index of this wave's own tile (T = 4 part + w)
        // bwd: this lane's sign bits of the block-step do not depend on the recurrence: fetched before the wait
        u32x4 mk_a[NJ], mk_ao = {0u, 0u, 0u, 0u}, mk_b = {0u, 0u, 0u, 0u};
        const int msh = 31 - (lane & 31);
        if (!FWD && !extra) {
            const unsigned* m2w = p.m2_save + mofs + 4 * (lane >> 5);
#pragma unroll
            for (int j = 0; j < NJ; ++j) mk_a[j] = *reinterpret_cast<const u32x4*>(m2w + (NJ * w + j) * 8);
            mk_ao = *reinterpret_cast<const u32x4*>(m2w + mown);
            mk_b = *reinterpret_cast<const u32x4*>(p.m1_save + mofs + 4 * (lane >> 5) + mown);
        }
        WS_STAMP(0);

        // ---- 1. receive the block input (fwd: code x, bwd: running gradient) --------------------------------------
        float xv[2];
        if (q == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float v = 0.f;
                if (erow && ec + i < C) v = FWD ? p.x0[(int64_t)(row0 + er) * C + ec + i] : p.g[((int64_t)(row0 + er) * n + (n - 1)) * C + ec + i];
                xv[i] = v;
            }
        } else {
            const unsigned epoch = ebase + (unsigned)q;
            unsigned spins = 0;
            // Phase 1: cheap wait (one 8-byte load per thread and round) for the producer's block INPUT, which part 0 publishes a
            // whole block-step of compute before the partials.  Phase 2: the partials are now ~1 us away -- after a fixed nap
            // every round fetches all granules, so the round that finds them complete IS the fetch.
            while ((unsigned)(gload(slot(epoch, P) + er * 32 + ec) >> 32) != epoch) {
                if (++spins > p.spin_limit) break;
                __builtin_amdgcn_s_sleep(1);
            }
            for (int z = 0; z < p.nap; ++z) __builtin_amdgcn_s_sleep(1);
            u32x4 v[P + 1];
            const u64* addr[P + 1];
#pragma unroll
            for (int s = 0; s <= P; ++s) addr[s] = slot(epoch, s) + er * 32 + ec;
            for (;;) {
                GranulePairs<P + 1>::load(v, addr);
                bool ok = true;
#pragma unroll
                for (int s = 0; s <= P; ++s) ok = ok && v[s][1] == epoch && v[s][3] == epoch;
                if (ok) break;
                if (++spins > p.spin_limit) { atomicOr(p.xerr, 1u); break; }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float res = __uint_as_float(v[0][2 * i]);
#pragma unroll
                for (int s = 1; s < P; ++s) res += __uint_as_float(v[s][2 * i]);
                if (FWD) res += bc[i];
                float x = __uint_as_float(v[P][2 * i]) + res;
                const bool valid = erow && ec + i < C;
                if (FWD) {
                    if (p.residuals && part == 0 && valid) p.residuals[((int64_t)(q - 1) * B + row0 + er) * C + ec + i] = res;
                } else if (pos == 0 && valid) {
                    x += p.g[((int64_t)(row0 + er) * n + t) * C + ec + i];      // entering time step t: add its upstream gradient
                }
                xv[i] = x;
            }
        }
        WS_STAMP(1);
        if (extra) {
            if (part == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (erow && ec + i < C) {
                        if (FWD) p.t_codes[((int64_t)(row0 + er) * n + it) * C + ec + i] = xv[i];
                        else p.dx0[(int64_t)(row0 + er) * C + ec + i] = xv[i];
                    }
                }
                // the slab's last block-step: every workgroup of its ring has long read the base; the next launch starts above this one's tags
                if (tid == 0) p.ebase[slab] = ebase + (((unsigned)q + 2u) & ~1u);
            }
            break;
        }
        const unsigned epoch_out = ebase + (unsigned)q + 1u;
        *reinterpret_cast<unsigned*>(xa + er * XP + ec) = pk_bf16<CT>(xv[0], xv[1]);
        WS_STAMP(2);
        lds_barrier();
        WS_STAMP(3);
        // everything below the barrier that is not on the critical path of the ring: block-input saves, the code output
        if (part == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                gstore<XL>(slot(epoch_out, P) + er * 32 + ec + i, ((u64)epoch_out << 32) | (u64)__float_as_uint(xv[i]));
                if (erow && ec + i < C) {
                    if (FWD) {
                        ((T*)p.xin_save)[(sbase + er) * C + ec + i] = (T)xv[i];
                        if (pos == 0) p.t_codes[((int64_t)(row0 + er) * n + it) * C + ec + i] = xv[i];   // input of block 0 = code of time it
                    } else {
                        ((T*)p.dr_save)[(sbase + er) * C + ec + i] = (T)xv[i];
                    }
                }
            }
        }

        // ---- 2. layer a (code -> the features this wave contracts in layer b), results stay in registers ---------------
        const u32x4 xb = *reinterpret_cast<const u32x4*>(xa + c * XP + g * 8);
        f32x4 aa[NJ], ao = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            aa[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (FWD) aa[j] = *reinterpret_cast<const f32x4*>(bias_a + 16 * (4 * j + w) + 4 * g);
        }
        if (FWD) ao = *reinterpret_cast<const f32x4*>(bias_a + 16 * own + 4 * g);
#pragma unroll
        for (int j = 0; j < NJ; ++j) aa[j] = mma<CT>(wa[j], xb, aa[j]);
        ao = mma<CT>(wa_own, xb, ao);           // this wave's own tile once more, for the activation save (and the sign bits)
        u32x4 hf[NKB];
#pragma unroll
        for (int jj = 0; jj < NKB; ++jj) {
            f32x4 a0 = aa[2 * jj], a1 = aa[2 * jj + 1];
            if (FWD) {
                hf[jj] = u32x4{pk_relu(pk_bf16<CT>(a0[0], a0[1])), pk_relu(pk_bf16<CT>(a0[2], a0[3])), pk_relu(pk_bf16<CT>(a1[0], a1[1])),
                               pk_relu(pk_bf16<CT>(a1[2], a1[3]))};
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    a0[r] = keep_if(a0[r], mk_a[2 * jj][r], msh);          // ReLU mask of h2 gates dh2
                    a1[r] = keep_if(a1[r], mk_a[2 * jj + 1][r], msh);
                }
                hf[jj] = u32x4{pk_bf16<CT>(a0[0], a0[1]), pk_bf16<CT>(a0[2], a0[3]), pk_bf16<CT>(a1[0], a1[1]), pk_bf16<CT>(a1[2], a1[3])};
            }
        }

        // ---- 3. layer b: K-partial over this wave's features, all 64 output features of the part ----------------------------
        {
            f32x4 bb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) bb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jj = 0; jj < NKB; ++jj) {
#pragma unroll
                for (int i = 0; i < 4; ++i) bb[i] = mma<CT>(wb[i][jj], hf[jj], bb[i]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) pb[(w * 4 + i) * 64 + lane] = bb[i];
        }
        {   // own tile of layer a: save (4 consecutive features of row c = 8 bytes) + sign bits
            u32x2 pk;
            if (FWD) {
                u64 bal[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) bal[r] = __ballot(ao[r] > 0.f);
                pk = u32x2{pk_relu(pk_bf16<CT>(ao[0], ao[1])), pk_relu(pk_bf16<CT>(ao[2], ao[3]))};
                if (lane == 0) store_tile_masks(p.m1_save + mofs + mown, bal);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) ao[r] = keep_if(ao[r], mk_ao[r], msh);
                pk = u32x2{pk_bf16<CT>(ao[0], ao[1]), pk_bf16<CT>(ao[2], ao[3])};
            }
            if (vrow) {
                T* dst = (FWD ? (T*)p.h1_save : (T*)p.dh2_save) + (sbase + c) * H + 16 * own + 4 * g;
                *reinterpret_cast<u32x2*>(dst) = pk;
            }
        }
        WS_STAMP(4);
        lds_barrier();
        {   // finish output tile w of the part: fixed-order sum of the four K-partials
            f32x4 sum = pb[(0 * 4 + w) * 64 + lane];
#pragma unroll
            for (int ww = 1; ww < 4; ++ww) {
                const f32x4 o = pb[(ww * 4 + w) * 64 + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r) sum[r] += o[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) sum[r] += bias_b[r];
            u32x2 pk;
            if (FWD) {
                u64 bal[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) bal[r] = __ballot(sum[r] > 0.f);
                pk = u32x2{pk_relu(pk_bf16<CT>(sum[0], sum[1])), pk_relu(pk_bf16<CT>(sum[2], sum[3]))};
                if (lane == 0) store_tile_masks(p.m2_save + mofs + mown, bal);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) sum[r] = keep_if(sum[r], mk_b[r], msh);            // ReLU mask of h1 gates dh1
                pk = u32x2{pk_bf16<CT>(sum[0], sum[1]), pk_bf16<CT>(sum[2], sum[3])};
            }
            *reinterpret_cast<u32x2*>(ah2 + c * SP + 16 * w + 4 * g) = pk;
            if (vrow) {
                T* dst = (FWD ? (T*)p.h2_save : (T*)p.dh1_save) + (sbase + c) * H + 16 * own + 4 * g;
                *reinterpret_cast<u32x2*>(dst) = pk;
            }
        }
        WS_STAMP(5);
        lds_barrier();
        WS_STAMP(6);

        // ---- 4. layer c: [16, own 64] x [64, C]: partial of the block output, published to the next block ---------------
        if (w < 2) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const T* arow = ah2 + c * SP + g * 8;
            acc = mma<CT>(*reinterpret_cast<const u32x4*>(arow), wc[0], acc);
            acc = mma<CT>(*reinterpret_cast<const u32x4*>(arow + 32), wc[1], acc);
            u64* dst = slot(epoch_out, part) + 16 * w + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) gstore<XL>(dst + (4 * g + r) * 32, ((u64)epoch_out << 32) | (u64)__float_as_uint(acc[r]));
        }
        WS_STAMP(7);
        // the next block-step of this workgroup overwrites xa / ah2 / pb only after its receive, i.e. after the other blocks of
        // the ring (or, for n_blocks == 1, its own peers) consumed this output: the LDS reads above are long finished
    }
}

// usable when the register-resident fragments and the one-workgroup-per-(slab, block, part) grid fit
inline bool usable(int compute, int B, int C, int H, int nb) {
    if (!vs_is16(compute) || C > 32 || (H != 128 && H != 256 && H != 512)) return false;
    const int nslabs = (B + 15) / 16, P = H / 64;
    // all workgroups must be co-resident (they wait for each other): one per CU, with headroom for whatever else is running
    static int cus = -1;
    if (cus < 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        cus = n;
    }
    if ((int64_t)nslabs * nb * P > (int64_t)cus - cus / 8) return false;      // MI355X: 256 CUs -> at most 224 workgroups
    const char* e = getenv("VS_ROLLOUT_WS");
    return !(e && e[0] == '0');
}

int g_xcd_local_allowed = 1;          // vs_mlp_rollout_xcd_local_set: 0 after a failed placement probe (process-wide)
inline int xcd_local_env() {
    const char* e = getenv("VS_ROLLOUT_XCD_LOCAL");                           // (read per call: tests switch it)
    return e ? atoi(e) : 1;
}
inline bool xcd_local_wanted(int nslabs) { return g_xcd_local_allowed && xcd_local_env() != 0 && nslabs % 8 == 0; }

inline size_t exchange_bytes(int B, int H) {
    const int nslabs = (B + 15) / 16, P = H / 64;
    return (size_t)2 * nslabs * (P + 1) * 512 * sizeof(u64);
}
// + the per-slab epoch base words behind the granules (padded to whole 16-byte units)
inline size_t base_words_bytes(int B) { return (size_t)((((B + 15) / 16) * 4 + 15) / 16) * 16; }

}  // namespace wsr

int pick_parts(int compute, int B, int C, int H) {
    // split the hidden dimension over P workgroups per slab when slices stay MFMA/k-step aligned and LDS-friendly
    const int KS = compute != VS_F32 ? 32 : 16;
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

int launch_ws(int compute, bool fwd, const RollParams& p, int mask_pitch, size_t workspace_bytes, hipStream_t stream) {
    // No fill: the workspace is zero when the caller creates it (the contract of vs_mlp_rollout_workspace_bytes) and every launch tags its
    // granules above the previous launch's (RollParams::ebase).  VS_ROLLOUT_ZERO=1: clear it anyway (all-zero is always a valid state).
    static const int zero_mode = getenv("VS_ROLLOUT_ZERO") ? atoi(getenv("VS_ROLLOUT_ZERO")) : 0;
    if (zero_mode && vs_zero_async(p.xbuf, p.xtotal - 16, stream) != hipSuccess) return vs_fail(VS_ERR_LAUNCH, "vs_mlp_rollout: zero fill failed");
    (void)workspace_bytes;
    const int nslabs = (p.B + 15) / 16, P = p.H / 64;
    dim3 grid((unsigned)(nslabs * p.nb * P)), block(wsr::WT);
    long long* dbg = nullptr;
#ifdef VS_WS_TIMING
    static long long* dbg_buf = nullptr;
    const int Q = (p.n - 1) * p.nb;
    if (!dbg_buf && hipMalloc(&dbg_buf, 8 * 8 * 4096) != hipSuccess) dbg_buf = nullptr;
    dbg = Q <= 4096 ? dbg_buf : nullptr;
#endif
    // rings on one XCD each (see wsr::gstore): 8, 16, ... slabs.  VS_ROLLOUT_XCD_LOCAL=0: agent-scope stores always; =2 (test aid): the
    // XCD-local stores with the rings deliberately spread over the XCDs.  vs_mlp_rollout_xcd_local_set(0) (the start-up probe's verdict,
    // ops.mlp_rollout_fwd) overrides both for the rest of the process.
    const bool xl = wsr::xcd_local_wanted(nslabs);
#define VS_WS_LAUNCH_X(KH, XLV)                                                                                       \
    do {                                                                                                              \
        if (compute == VS_BF16) {                                                                                     \
            if (fwd) hipLaunchKernelGGL((wsr::rollout_ws_kernel<KH, true, VS_BF16, XLV>), grid, block, 0, stream, p, mask_pitch, dbg);  \
            else hipLaunchKernelGGL((wsr::rollout_ws_kernel<KH, false, VS_BF16, XLV>), grid, block, 0, stream, p, mask_pitch, dbg);     \
        } else {                                                                                                      \
            if (fwd) hipLaunchKernelGGL((wsr::rollout_ws_kernel<KH, true, VS_F16, XLV>), grid, block, 0, stream, p, mask_pitch, dbg);   \
            else hipLaunchKernelGGL((wsr::rollout_ws_kernel<KH, false, VS_F16, XLV>), grid, block, 0, stream, p, mask_pitch, dbg);      \
        }                                                                                                             \
    } while (0)
#define VS_WS_LAUNCH(KH)                    \
    do {                                    \
        if (xl) VS_WS_LAUNCH_X(KH, true);   \
        else VS_WS_LAUNCH_X(KH, false);     \
    } while (0)
    if (p.H == 512) VS_WS_LAUNCH(16);
    else if (p.H == 256) VS_WS_LAUNCH(8);
    else VS_WS_LAUNCH(4);
#undef VS_WS_LAUNCH
#undef VS_WS_LAUNCH_X
    VS_CHECK_LAUNCH("vs_mlp_rollout (weight-stationary)");
#ifdef VS_WS_TIMING
    if (dbg) {          // debug build only: phase timestamps of (slab 0, part 0) of every block, 10 ns ticks
        if (hipStreamSynchronize(stream) != hipSuccess) return VS_OK;
        static long long h[8 * 4096];
        if (hipMemcpy(h, dbg, sizeof(long long) * 8 * Q, hipMemcpyDeviceToHost) != hipSuccess) return VS_OK;
        double sum[8] = {0}, hop = 0;
        int cnt = 0;
        for (int q = 2; q < Q - 1; ++q) {
            for (int k = 1; k < 8; ++k) sum[k] += (double)(h[q * 8 + k] - h[q * 8 + k - 1]);
            hop += (double)(h[(q + 1) * 8 + 1] - h[q * 8 + 7]);
            ++cnt;
        }
        fprintf(stderr, "[ws %s] ticks of 10 ns per block-step: wait %.1f | post-recv %.1f | barrier %.1f | layer a %.1f | layer b %.1f | barrier %.1f | "
                "layer c + publish %.1f | hop publish->received %.1f | total %.1f (%d block-steps)\n", fwd ? "fwd" : "bwd", sum[1] / cnt, sum[2] / cnt,
                sum[3] / cnt, sum[4] / cnt, sum[5] / cnt, sum[6] / cnt, sum[7] / cnt, hop / cnt, (double)(h[(Q - 1) * 8 + 7] - h[0]) / (Q - 1), Q);
    }
#endif
    return VS_OK;
}

template <int CT>
int launch_roll(bool fwd, const RollParams& p, hipStream_t stream) {
    int Cf, Ck, Hk, NpH, NpC;
    size_t oc, o1, o2, ob, op;
    const size_t smem = lds_layout(p.C, p.H, p.nb, p.P, (int)sizeof(typename RT<CT>::T), RT<CT>::KS, RT<CT>::U, &Cf, &Ck, &Hk, &NpH, &NpC, &oc, &o1,
                                   &o2, &ob, &op);
    if (smem > 160 * 1024) return vs_fail(VS_ERR_UNSUPPORTED, "vs_mlp_rollout: C=%d H=%d needs %zu B of LDS (> 160 KiB)", p.C, p.H, smem);
    const void* kfn = fwd ? (const void*)rollout_fwd_kernel<CT> : (const void*)rollout_bwd_kernel<CT>;
    if (smem > 64 * 1024) {
        if (hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return vs_fail(VS_ERR_LAUNCH, "vs_mlp_rollout: cannot raise dynamic LDS limit to %zu", smem);
    }
    const int nslabs = (p.B + 15) / 16;
    if (p.P > 1) {
        if (vs_zero_async(p.xbuf, p.xtotal - 16, stream) != hipSuccess) return vs_fail(VS_ERR_LAUNCH, "vs_mlp_rollout: zero fill failed");
    }
    dim3 grid((unsigned)(nslabs * p.P));
    if (fwd) hipLaunchKernelGGL(rollout_fwd_kernel<CT>, grid, dim3(NT), smem, stream, p);
    else hipLaunchKernelGGL(rollout_bwd_kernel<CT>, grid, dim3(NT), smem, stream, p);
    VS_CHECK_LAUNCH("vs_mlp_rollout");
    return VS_OK;
}

int check_common(int compute, int B, int C, int H, int nb, int n) {
    VS_CHECK_ARG(vs_dtype_ok(compute), "vs_mlp_rollout: compute type %d", compute);
    VS_CHECK_ARG(B > 0 && C > 0 && H > 0 && n >= 1, "vs_mlp_rollout: bad sizes B=%d C=%d H=%d n=%d", B, C, H, n);
    VS_CHECK_ARG(nb >= 1 && nb <= MAXB, "vs_mlp_rollout: n_blocks=%d (supported: 1..%d)", nb, MAXB);
    return VS_OK;
}

inline unsigned spin_limit_from_env() {
    const char* e = getenv("VS_ROLLOUT_SPIN_LIMIT");            // test aid: a tiny limit forces the time-out path
    const long v = e ? atol(e) : 0;
    return v > 0 ? (unsigned)v : (1u << 22);
}

int setup_exchange(RollParams& p, int compute, void* workspace, size_t workspace_bytes) {
    p.spin_limit = spin_limit_from_env();
    p.P = pick_parts(compute, p.B, p.C, p.H);
    const int nslabs = (p.B + 15) / 16, Cf = round_up(p.C, 4);
    const size_t need = (size_t)2 * nslabs * p.P * 16 * Cf * sizeof(u64) + 16;
    if (p.P > 1 && (!workspace || workspace_bytes < need)) p.P = 1;         // no exchange area: run unsplit
    p.xbuf = (u64*)workspace;
    p.xtotal = workspace_bytes & ~(size_t)15;
    p.xerr = p.P > 1 ? (vs_g_exchange_guard ? vs_g_exchange_guard : (unsigned*)((char*)workspace + p.xtotal - 16)) : nullptr;
    return VS_OK;
}

// weight-stationary form: same decision in both directions (the sign-bit layout differs from the slab form)
bool setup_ws(RollParams& p, int compute, void* workspace, size_t workspace_bytes, int* mask_pitch) {
    p.spin_limit = spin_limit_from_env();
    { const char* e = getenv("VS_ROLLOUT_NAP"); p.nap = e ? atoi(e) : WS_NAP; if (p.nap < 0 || p.nap > 64) p.nap = WS_NAP; }
    if (!wsr::usable(compute, p.B, p.C, p.H, p.nb) || p.n < 2) return false;
    if (!workspace || workspace_bytes < wsr::exchange_bytes(p.B, p.H) + wsr::base_words_bytes(p.B) + 16) return false;
    p.xbuf = (u64*)workspace;
    p.ebase = (unsigned*)((char*)workspace + wsr::exchange_bytes(p.B, p.H));
    p.xtotal = workspace_bytes & ~(size_t)15;
    p.xerr = vs_g_exchange_guard ? vs_g_exchange_guard : (unsigned*)((char*)workspace + p.xtotal - 16);
    p.ring_skew = wsr::xcd_local_env() == 2 ? 1 : 0;
    *mask_pitch = pick_parts(compute, p.B, p.C, p.H) * 32;
    return true;
}

}  // namespace

extern "C" int vs_mlp_rollout_parts(int compute, int B, int C, int H) { return pick_parts(compute, B, C, H); }

// The XCD-local exchange (wsr::gstore<true>) leans on a dispatch property HIP does not promise (workgroups 8 apart share an XCD).
// _get: would a launch of this geometry take it now?  _set(0): never again in this process (the caller's start-up probe failed).
extern "C" int vs_mlp_rollout_xcd_local_get(int compute, int B, int C, int H, int n_blocks) {
    return wsr::usable(compute, B, C, H, n_blocks) && wsr::xcd_local_wanted((B + 15) / 16) ? 1 : 0;
}
extern "C" int vs_mlp_rollout_xcd_local_set(int allowed) {
    wsr::g_xcd_local_allowed = allowed ? 1 : 0;
    return VS_OK;
}

extern "C" size_t vs_mlp_rollout_workspace_bytes(int compute, int B, int C, int H) {
    const int P = pick_parts(compute, B, C, H);
    size_t need = P > 1 ? (size_t)2 * ((B + 15) / 16) * P * 16 * round_up(C, 4) * sizeof(u64) + 16 : 0;
    if (vs_is16(compute) && C <= 32 && (H == 128 || H == 256 || H == 512)) {       // weight-stationary form (any n_blocks)
        const size_t ws = wsr::exchange_bytes(B, H) + wsr::base_words_bytes(B) + 16;
        if (ws > need) need = ws;
    }
    return need;
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
    int mask_pitch = 0;
    if (setup_ws(p, compute, workspace, workspace_bytes, &mask_pitch)) return launch_ws(compute, true, p, mask_pitch, p.xtotal, (hipStream_t)stream);
    setup_exchange(p, compute, workspace, workspace_bytes);
    return compute == VS_BF16 ? launch_roll<VS_BF16>(true, p, (hipStream_t)stream)
           : compute == VS_F16 ? launch_roll<VS_F16>(true, p, (hipStream_t)stream) : launch_roll<VS_F32>(true, p, (hipStream_t)stream);
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
    int mask_pitch = 0;
    if (setup_ws(p, compute, workspace, workspace_bytes, &mask_pitch)) return launch_ws(compute, false, p, mask_pitch, p.xtotal, (hipStream_t)stream);
    setup_exchange(p, compute, workspace, workspace_bytes);
    return compute == VS_BF16 ? launch_roll<VS_BF16>(false, p, (hipStream_t)stream)
           : compute == VS_F16 ? launch_roll<VS_F16>(false, p, (hipStream_t)stream) : launch_roll<VS_F32>(false, p, (hipStream_t)stream);
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

// Several weights in one launch (the integrator has 3 x n_blocks matrices, each needed as L and L^T): one workgroup-stride
// loop over the concatenated 16-byte units of all jobs.
namespace {
constexpr int PK_MAXJ = 48;
struct PackJobs {
    const float* src[PK_MAXJ];
    void* dst[PK_MAXJ];
    int N[PK_MAXJ], K[PK_MAXJ], transpose[PK_MAXJ];
    long long unit_off[PK_MAXJ + 1];
    int nj;
};
template <int CT>
__global__ __launch_bounds__(256) void pack_multi_kernel(PackJobs J) {
    typedef typename RT<CT>::T T;
    constexpr int KS = RT<CT>::KS, U = RT<CT>::U;
    const long long total = J.unit_off[J.nj];
    for (long long gu = (long long)blockIdx.x * 256 + threadIdx.x; gu < total; gu += (long long)gridDim.x * 256) {
        int j = 0;
        while (J.unit_off[j + 1] <= gu) ++j;
        const long long u = gu - J.unit_off[j];
        const int N = J.N[j], K = J.K[j], transpose = J.transpose[j];
        const float* src = J.src[j];
        const int ksteps = (K + KS - 1) / KS;
        const int lane = (int)(u & 63);
        const long long piece = u >> 6;
        const int s = (int)(piece % ksteps), nt = (int)(piece / ksteps);
        const int n = nt * 16 + (lane & 15), k0 = s * KS + (lane >> 4) * U;
        T tmp[U];
#pragma unroll
        for (int e = 0; e < U; ++e) {
            const int k = k0 + e;
            float v = 0.f;
            if (n < N && k < K) v = transpose ? src[(int64_t)k * N + n] : src[(int64_t)n * K + k];
            tmp[e] = (T)v;
        }
        *reinterpret_cast<u32x4*>((T*)J.dst[j] + u * U) = *reinterpret_cast<u32x4*>(tmp);
    }
}
}  // namespace

extern "C" int vs_pack_rollout_weights(int compute, int n_jobs, const float* const* src, const int* transpose, const int* N, const int* K,
                                       void* const* dst, void* stream) {
    VS_CHECK_ARG(vs_dtype_ok(compute), "vs_pack_rollout_weights: compute type %d", compute);
    VS_CHECK_ARG(n_jobs >= 1 && n_jobs <= PK_MAXJ && src && transpose && N && K && dst, "vs_pack_rollout_weights: bad argument (1..%d jobs)", PK_MAXJ);
    const int KS = compute != VS_F32 ? 32 : 16;
    PackJobs J;
    J.nj = n_jobs;
    J.unit_off[0] = 0;
    for (int j = 0; j < n_jobs; ++j) {
        VS_CHECK_ARG(src[j] && dst[j] && N[j] > 0 && K[j] > 0, "vs_pack_rollout_weights: bad job %d", j);
        J.src[j] = src[j]; J.dst[j] = dst[j]; J.N[j] = N[j]; J.K[j] = K[j]; J.transpose[j] = transpose[j];
        J.unit_off[j + 1] = J.unit_off[j] + (long long)((N[j] + 15) / 16) * ((K[j] + KS - 1) / KS) * 64;
    }
    long long blocks = (J.unit_off[n_jobs] + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (compute == VS_BF16) hipLaunchKernelGGL(pack_multi_kernel<VS_BF16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, J);
    else if (compute == VS_F16) hipLaunchKernelGGL(pack_multi_kernel<VS_F16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, J);
    else hipLaunchKernelGGL(pack_multi_kernel<VS_F32>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, J);
    VS_CHECK_LAUNCH("vs_pack_rollout_weights");
    return VS_OK;
}

extern "C" size_t vs_rollout_packed_elems(int compute, int N, int K) {
    const int KS = compute != VS_F32 ? 32 : 16;
    return (size_t)((N + 15) / 16) * 16 * (size_t)((K + KS - 1) / KS) * KS;
}

extern "C" int vs_pack_rollout_weight(int compute, const float* src, int transpose, int N, int K, void* dst, void* stream) {
    VS_CHECK_ARG(vs_dtype_ok(compute), "vs_pack_rollout_weight: compute type %d", compute);
    VS_CHECK_ARG(src && dst && N > 0 && K > 0, "vs_pack_rollout_weight: bad argument");
    const int KS = compute != VS_F32 ? 32 : 16;
    int64_t units = (int64_t)((N + 15) / 16) * ((K + KS - 1) / KS) * 64;
    int64_t blocks = (units + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (compute == VS_BF16)
        hipLaunchKernelGGL(pack_kernel<VS_BF16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, transpose, N, K, (__bf16*)dst);
    else if (compute == VS_F16)
        hipLaunchKernelGGL(pack_kernel<VS_F16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, transpose, N, K, (_Float16*)dst);
    else
        hipLaunchKernelGGL(pack_kernel<VS_F32>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, transpose, N, K, (float*)dst);
    VS_CHECK_LAUNCH("vs_pack_rollout_weight");
    return VS_OK;
}
