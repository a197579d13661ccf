// vs_gemm_core.h -- the LDS-tiled MFMA contraction kernel shared by vs_gemm.hip (dense operands) and vs_conv.hip
// (implicit-GEMM convolution operands).  See vs_gemm.hip for the design notes.
#pragma once
#include <stdlib.h>
#include <string.h>
#include "vs_common.h"

namespace {

enum { LR = VS_LAYOUT_R, LS = VS_LAYOUT_S };

template <int CT> struct CTraits;
template <> struct CTraits<VS_F32> { typedef float T; static constexpr int U = 4; static constexpr int KSTEP = 8; };
template <> struct CTraits<VS_BF16> { typedef __bf16 T; static constexpr int U = 8; static constexpr int KSTEP = 16; };
template <> struct CTraits<VS_F16> { typedef _Float16 T; static constexpr int U = 8; static constexpr int KSTEP = 16; };

// one 32x32x16 MFMA on 16-bit operands given as raw fragment bits (8 elements per lane): bf16 or IEEE half, fp32 accumulate
template <int CT>
__device__ __forceinline__ f32x16 mfma16_32(const u32x4& a, const u32x4& b, const f32x16& c) {
    if constexpr (CT == VS_BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

#include "vs_adam_math.h"

struct Epi {
    void* C; int64_t ldc; int c_dtype;
    float alpha; const float* bias; int act;
    const void* mask; int64_t ldmask; int mask_dtype; int mask_act;
    int accumulate;
    // nchw_hw > 0: rows m are channels, columns n are pixels (b * hw + pix) of an NCHW tensor with `nchw_c` channels:
    // element (m, n) lives at ((n / hw) * nchw_c + m) * hw + n % hw and the bias is per ROW (channel).
    int64_t nchw_hw, nchw_c;
    // optional sub-grid scatter (parity phases of a stride-2 transposed convolution): column n enumerates an iteration
    // grid [B][g_hw / g_w][g_w] and lands on output pixel (gy*sy + oy, gx*sx + ox) of a plane with rows of o_w pixels
    int g_w, g_hw, o_w, sy, sx, oy, ox;
    // batched dense GEMM (vs_gemm_batched): gridDim.z = batch * splits; operand / output element strides between problems
    int splits_per_batch;
    int64_t batch_a, batch_b, batch_c;
    // fused optimizer (vs_gemm_adam, ring-tile kernels only): the result is the GRADIENT of the fp32 parameter C [M, ldc]; the
    // epilogue applies the Adam update to C / adam_m / adam_v (same layout) and refreshes the 16-bit operand copy, the gradient
    // is never stored
    float* adam_m; float* adam_v; unsigned short* adam_shadow; int adam_shadow_dtype;
    const int* adam_step; int adam_skipped;
    const unsigned* adam_guard;          // vs_exchange_guard_set: non-zero word = leave the parameter block untouched
    double adam_lr, adam_beta1, adam_beta2; float adam_eps;
    // workgroup -> tile map: 1 = the workgroups of one XCD (b, b + 8, ... under round-robin dispatch) take a contiguous run of the
    // (split, tile row, tile column) order, so an XCD's L2 holds a few A row panels and one K chunk instead of a share of everything
    int xcd_runs;
    // split-K finished inside the launch (vs_gemm / vs_gemm_batched): every split stores its fp32 slab, bumps its tile's counter and the LAST
    // workgroup to arrive adds the slabs in split order (bitwise what splitk_reduce_kernel computes), applies the epilogue and clears the counter
    unsigned* sk_counters; int sk_splits; int64_t sk_bytes;      // sk_bytes: size of the whole slab area (< 2 GiB)
    // frame losses in the epilogue (vs_gemm_frame_loss, 256x256 tile kernel only): the result act(alpha * acc + bias) is row r = (b, g) of the
    // decoded frame stack [B, G, N]; it is compared with full[b, target(g)] (fp32 [B, T, N]) and NOT stored: the squared errors go to two
    // per-workgroup partial sums (frame 0 | frames 1..: fl_partials[2 * workgroup ..]) and k * (y - target) * act'(y) to fl_dz in fl_dz_dtype
    const float* fl_full; const int* fl_tdev; int fl_ae_shift, fl_first, fl_G, fl_T;
    const float* fl_up; float fl_l_ae, fl_l_pred, fl_inv_ae, fl_inv_pred;        // k = *fl_up * l * 2 * (1 / count), in this order
    void* fl_dz; int fl_dz_dtype; float* fl_partials;
    int adam_pipe;                       // fused optimizer: state of four row pieces requested ahead (VS_ADAM_PIPE=0: one piece at a time)
    int p8_plain;                        // VS_GEMM_P8_QUICK=0 (measurements): interior tiles of the staggered tile take the general stores too
};

// XCD-aware, bijective block -> tile index (blocks b and b + 8 share an XCD under round-robin dispatch: speed only)
__device__ __forceinline__ unsigned big_tile_of(unsigned b, unsigned nwg) {
    const unsigned xcd = b & 7u, q = nwg >> 3, r = nwg & 7u;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

__device__ __forceinline__ Epi epi_for_batch(const Epi& e, int64_t batch) {
    Epi r = e;
    if (batch) r.C = (char*)e.C + batch * e.batch_c * vs_esize(e.c_dtype);
    return r;
}

// ---- split-K finished inside the launch ---------------------------------------------------------------------------------------
// No fences: an agent-scope fence is a write-back + invalidate of the XCD's whole L2 (measured in the WaveEq step: the split launches 4x
// slower and the integrator's XCD-local exchange, which lives in L2, 1.7x).  Instead the slabs are written THROUGH to the fabric (sc1
// stores = relaxed agent-scope stores), a workgroup waits for its own stores (vmcnt 0), then bumps the tile's counter (relaxed agent-scope
// atomic), and the last workgroup to arrive reads every slab with sc1 loads, which do not hit in a stale line of its own L2.
// The slab area of one launch is < 2 GiB (the host falls back to the reduce launch otherwise): one buffer descriptor, 32-bit byte offsets,
// aux = 16 = sc1 on every access (compiler-tracked loads and stores of 4 and 16 bytes).
typedef __attribute__((address_space(8))) void* sk_rsrc_t;
__device__ __forceinline__ auto sk_rsrc(const float* base, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}
template <class R> __device__ __forceinline__ void sk_store(R r, int64_t elem, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)(elem * 4), 0, 16);
}
template <class R> __device__ __forceinline__ void sk_store4(R r, int64_t elem, const f32x4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)(elem * 4), 0, 16);
}
template <class R> __device__ __forceinline__ float sk_load(R r, int64_t elem) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)(elem * 4), 0, 16));
}
template <class R> __device__ __forceinline__ f32x4 sk_load4(R r, int64_t elem) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(elem * 4), 0, 16));
}
// Arrival of one split of output tile `tile_id` (all threads of the workgroup call it after their slab stores): true for the workgroup
// that arrives last.  `lds_word`: any LDS word nobody reads or writes any more (the kernels' only LDS object is their dynamic tile area).
__device__ __forceinline__ bool sk_arrive_last(const Epi& e, unsigned tile_id, volatile unsigned* lds_word) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // my slab stores have reached the fabric
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = atomicAdd(e.sk_counters + tile_id, 1u);
        const unsigned last = old == (unsigned)(e.sk_splits - 1);
        if (last) atomicExch(e.sk_counters + tile_id, 0u);      // everybody has arrived: the counter is ready for the next launch
        *lds_word = last;
    }
    __syncthreads();
    return *lds_word != 0u;
}

// row-major form: element (m, n) at C[m*ldc + n], bias per column
__device__ __forceinline__ void epi_store(const Epi& e, int64_t m, int64_t n, float v) {
    v *= e.alpha;
    if (e.bias) v += e.bias[n];
    v = vs_act(v, e.act);
    if (e.mask) v *= vs_act_grad_from_out(vs_ld(e.mask, e.mask_dtype, m * e.ldmask + n), e.mask_act);
    const int64_t idx = m * e.ldc + n;
    if (e.accumulate) v += vs_ld(e.C, e.c_dtype, idx);
    vs_st(e.C, e.c_dtype, idx, v);
}

// NCHW form: `col_base` = (n / hw) * C * hw + n % hw is computed once per column by the caller; bias per row (channel)
__device__ __forceinline__ void epi_store_nchw(const Epi& e, int64_t m, int64_t col_base, float v) {
    v *= e.alpha;
    if (e.bias) v += e.bias[m];
    v = vs_act(v, e.act);
    const int64_t idx = col_base + m * e.nchw_hw;
    if (e.mask) v *= vs_act_grad_from_out(vs_ld(e.mask, e.mask_dtype, idx), e.mask_act);
    if (e.accumulate) v += vs_ld(e.C, e.c_dtype, idx);
    vs_st(e.C, e.c_dtype, idx, v);
}

__device__ __forceinline__ int64_t nchw_col_base(const Epi& e, int64_t n) {
    if (e.g_hw == 0) {
        const int64_t b = n / e.nchw_hw;
        return b * e.nchw_c * e.nchw_hw + (n - b * e.nchw_hw);
    }
    const int64_t b = n / e.g_hw;
    const int r = (int)(n - b * e.g_hw);
    const int gy = r / e.g_w, gx = r - gy * e.g_w;
    return b * e.nchw_c * e.nchw_hw + (int64_t)(gy * e.sy + e.oy) * e.o_w + gx * e.sx + e.ox;
}

// ---- dense operand: element (i,k) at p[i*ld+k] (R) or p[k*ld+i] (S); T = storage = compute type -------
template <int CT, int LAYOUT>
struct Dense {
    typedef typename CTraits<CT>::T T;
    static constexpr int U = CTraits<CT>::U;
    static constexpr int layout = LAYOUT;
    const T* p; int64_t ld; int64_t rows; int64_t K; int vec_ok;
    __device__ __forceinline__ void shift(int64_t elems) { p += elems; }       // batched GEMM: move to this problem's operand

    // Per-thread, per-unit-slot state that does not change along K (the row a unit belongs to is fixed for the whole K loop)
    struct State { const T* base; int ok; };
    __device__ __forceinline__ State prepare(int64_t i) const {
        State st;
        if (LAYOUT == LR) { st.ok = i < rows; st.base = p + i * ld; }
        else { st.ok = i < rows ? (i + U <= rows ? 2 : 1) : 0; st.base = p + i; }
        return st;
    }
    // 16-byte unit: R -> elements (i, k..k+U-1);  S -> elements (i..i+U-1, k)
    __device__ __forceinline__ u32x4 load(const State& st, int64_t i, int64_t k) const {
        u32x4 r = {0u, 0u, 0u, 0u};
        if (!st.ok || k >= K) return r;
        if (LAYOUT == LR) {
            const T* q = st.base + k;
            if (vec_ok && k + U <= K) return *reinterpret_cast<const u32x4*>(q);
            T tmp[U];
#pragma unroll
            for (int j = 0; j < U; ++j) tmp[j] = (k + j < K) ? q[j] : (T)0.f;
            return *reinterpret_cast<u32x4*>(tmp);
        } else {
            const T* q = st.base + k * ld;
            if (vec_ok && st.ok == 2) return *reinterpret_cast<const u32x4*>(q);
            T tmp[U];
#pragma unroll
            for (int j = 0; j < U; ++j) tmp[j] = (i + j < rows) ? q[j] : (T)0.f;
            return *reinterpret_cast<u32x4*>(tmp);
        }
    }
};

// LDS geometry of one operand tile
template <int CT, int LAYOUT, int ROWS, int BK>
struct TileGeom {
    static constexpr int U = CTraits<CT>::U;
    // R: [ROWS][BK+U]   (pitch odd multiple of 16 B -> ds_read_b128 conflict free)
    // S: [BK][ROWS+pad] (bf16: pad 32 -> the four k-rows of a tr_b16 read land in distinct 64-byte bank groups)
    static constexpr int PITCH = LAYOUT == LR ? BK + U : (CT != VS_F32 ? ROWS + 32 : ROWS + 4);
    static constexpr int ELEMS = LAYOUT == LR ? ROWS * PITCH : BK * PITCH;
    static constexpr int UNITS = ROWS * BK / U;          // 16-byte units per tile
    static constexpr int PER_THREAD = UNITS / 256;
    static_assert(UNITS % 256 == 0, "tile must split evenly over 256 threads");
    // unit u -> (row offset, k offset) and LDS element offset
    __device__ static __forceinline__ void map(int u, int& di, int& dk, int& lds_off) {
        if (LAYOUT == LR) {
            constexpr int CH = BK / U;
            di = u / CH; dk = (u % CH) * U; lds_off = di * PITCH + dk;
        } else {
            constexpr int CH = ROWS / U;
            dk = u / CH; di = (u % CH) * U; lds_off = dk * PITCH + di;
        }
    }
};

template <int CT, class Op, int ROWS, int BK>
__device__ __forceinline__ void tile_prepare(const Op& op, int64_t i0, typename Op::State (&st)[TileGeom<CT, Op::layout, ROWS, BK>::PER_THREAD]) {
    typedef TileGeom<CT, Op::layout, ROWS, BK> G;
#pragma unroll
    for (int it = 0; it < G::PER_THREAD; ++it) {
        int di, dk, off;
        G::map(threadIdx.x + it * 256, di, dk, off);
        st[it] = op.prepare(i0 + di);
    }
}

template <int CT, class Op, int ROWS, int BK>
__device__ __forceinline__ void tile_fetch(const Op& op, const typename Op::State (&st)[TileGeom<CT, Op::layout, ROWS, BK>::PER_THREAD], int64_t i0,
                                           int64_t k0, u32x4 (&regs)[TileGeom<CT, Op::layout, ROWS, BK>::PER_THREAD]) {
    typedef TileGeom<CT, Op::layout, ROWS, BK> G;
#pragma unroll
    for (int it = 0; it < G::PER_THREAD; ++it) {
        int di, dk, off;
        G::map(threadIdx.x + it * 256, di, dk, off);
        regs[it] = op.load(st[it], i0 + di, k0 + dk);
    }
}

template <int CT, int LAYOUT, int ROWS, int BK>
__device__ __forceinline__ void tile_commit(typename CTraits<CT>::T* lds,
                                            const u32x4 (&regs)[TileGeom<CT, LAYOUT, ROWS, BK>::PER_THREAD]) {
    typedef TileGeom<CT, LAYOUT, ROWS, BK> G;
#pragma unroll
    for (int it = 0; it < G::PER_THREAD; ++it) {
        int di, dk, off;
        G::map(threadIdx.x + it * 256, di, dk, off);
        *reinterpret_cast<u32x4*>(lds + off) = regs[it];
    }
}

// ---- fragment fetch: 16-bit types, one 32-row block, one 16-deep k step -> 8 elements as raw bits (lane r=l&31, h=l>>5 holds
// k=8h..8h+7)
typedef __attribute__((ext_vector_type(4))) short vs_i16x4;
__device__ __forceinline__ u32x4 vs_tr16_pair(const unsigned short* a, int second_offset_elems) {
    // two ds_read_b64_tr_b16 (k = 8h..8h+3 and 8h+4..8h+7): the transposing read moves 16-bit elements, whatever they encode
    typedef __attribute__((address_space(3))) vs_i16x4 lds_i16x4;
    const vs_i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4*)(a));
    const vs_i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4*)(a + second_offset_elems));
    typedef __attribute__((ext_vector_type(8))) short i16x8;
    const i16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(u32x4, both);
}

template <int LAYOUT, int PITCH>
__device__ __forceinline__ u32x4 frag_bf16(const unsigned short* tile, int row0, int kk, int lane) {
    if (LAYOUT == LR) {
        const int r = lane & 31, h = lane >> 5;
        return *reinterpret_cast<const u32x4*>(tile + (row0 + r) * PITCH + kk + 8 * h);
    } else {
        // ds_read_b64_tr_b16: per 16-lane group a 4(k) x 16(row) block; lane 4q+p supplies the address of k-row q,
        // rows 4p..4p+3; lane i receives row i of the four k-rows.  Two reads cover k = 8h..8h+3 and 8h+4..8h+7.
        const int li = lane & 15, q = li >> 2, p = li & 3, cb = (lane >> 4) & 1, h = lane >> 5;
        const unsigned short* a = tile + (kk + 8 * h + q) * PITCH + row0 + 16 * cb + 4 * p;
        return vs_tr16_pair(a, 4 * PITCH);
    }
}

// ---- fragment fetch: f32, one 32-row block, one 8-deep k group -> 4 floats; element j belongs to k = kk+4q+j
template <int LAYOUT, int PITCH>
__device__ __forceinline__ f32x4 frag_f32(const float* tile, int row0, int kk, int lane) {
    const int r = lane & 31, q = lane >> 5;
    if (LAYOUT == LR) {
        return *reinterpret_cast<const f32x4*>(tile + (row0 + r) * PITCH + kk + 4 * q);
    } else {
        const float* a = tile + (kk + 4 * q) * PITCH + row0 + r;
        f32x4 v;
        v[0] = a[0]; v[1] = a[PITCH]; v[2] = a[2 * PITCH]; v[3] = a[3 * PITCH];
        return v;
    }
}

template <int CT, class OpA, class OpB, int BM, int BN, int BK, bool NCHW = false>
__global__ __launch_bounds__(256) void gemm_kernel(OpA A, OpB B, int64_t M, int64_t N, int64_t K,
                                                   int k_tiles_per_split, Epi epi_in, float* slabs) {
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (epi_in.xcd_runs) {
        const unsigned gx = gridDim.x, gy = gridDim.y;
        const unsigned t = big_tile_of(bx + gx * (by + gy * bz), gx * gy * gridDim.z);
        bz = t / (gx * gy);
        by = (t - bz * gx * gy) / gx;
        bx = t - (bz * gy + by) * gx;
    }
    int zsplit = bz;
    if (epi_in.splits_per_batch > 0) {                  // batched: z = batch * splits + split
        const int batch = bz / epi_in.splits_per_batch;
        zsplit = bz - batch * epi_in.splits_per_batch;
        A.shift(batch * epi_in.batch_a);
        B.shift(batch * epi_in.batch_b);
    }
    const Epi epi = epi_for_batch(epi_in, epi_in.splits_per_batch > 0 ? bz / epi_in.splits_per_batch : 0);
    typedef typename CTraits<CT>::T T;
    typedef TileGeom<CT, OpA::layout, BM, BK> GA;
    typedef TileGeom<CT, OpB::layout, BN, BK> GB;
    constexpr int WM = BM / 2, WN = BN / 2;          // per-wave C block (waves 2x2)
    constexpr int TM = WM / 32, TN = WN / 32;
    static_assert(TM >= 1 && TN >= 1, "tile too small");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* sA = reinterpret_cast<T*>(smem);
    T* sB = sA + GA::ELEMS;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;
    const int64_t m0 = (int64_t)by * BM, n0 = (int64_t)bx * BN;
    const int64_t kt_total = (K + BK - 1) / BK;
    const int64_t kt_begin = (int64_t)zsplit * k_tiles_per_split;
    int64_t kt_end = kt_begin + k_tiles_per_split;
    if (kt_end > kt_total) kt_end = kt_total;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    u32x4 ra[GA::PER_THREAD], rb[GB::PER_THREAD];
    typename OpA::State sta[GA::PER_THREAD];
    typename OpB::State stb[GB::PER_THREAD];
    tile_prepare<CT, OpA, BM, BK>(A, m0, sta);
    tile_prepare<CT, OpB, BN, BK>(B, n0, stb);
    if (kt_begin < kt_end) {
        tile_fetch<CT, OpA, BM, BK>(A, sta, m0, kt_begin * BK, ra);
        tile_fetch<CT, OpB, BN, BK>(B, stb, n0, kt_begin * BK, rb);
    }
    for (int64_t kt = kt_begin; kt < kt_end; ++kt) {
        tile_commit<CT, OpA::layout, BM, BK>(sA, ra);
        tile_commit<CT, OpB::layout, BN, BK>(sB, rb);
        __syncthreads();
#ifndef VS_DIAG_NO_GLOBAL
        if (kt + 1 < kt_end) {       // prefetch next tile into registers; lands while the MFMAs below run
            tile_fetch<CT, OpA, BM, BK>(A, sta, m0, (kt + 1) * BK, ra);
            tile_fetch<CT, OpB, BN, BK>(B, stb, n0, (kt + 1) * BK, rb);
        }
#endif
        if constexpr (CT != VS_F32) {
#pragma unroll
            for (int kk = 0; kk < BK; kk += 16) {
                u32x4 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[i] = frag_bf16<OpA::layout, GA::PITCH>(reinterpret_cast<const unsigned short*>(sA), wm + 32 * i, kk, lane);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[j] = frag_bf16<OpB::layout, GB::PITCH>(reinterpret_cast<const unsigned short*>(sB), wn + 32 * j, kk, lane);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
#ifdef VS_DIAG_NO_MFMA
                        acc[i][j][0] += (float)fa[i][0] + (float)fb[j][0];          // keeps the LDS reads alive, no matrix work (diagnostic)
#else
                        acc[i][j] = mfma16_32<CT>(fa[i], fb[j], acc[i][j]);
#endif
                    }
            }
        } else {
            // parity mode: two-level summation.  Each K tile (16 products) is accumulated in a fresh MFMA chain and then
            // added to the running total, so the rounding error grows like sqrt(16) + sqrt(K/16) instead of sqrt(K) of a
            // single k-ordered fmaf chain (K reaches 4608 in the 512-channel 3x3 convolutions) -- comparable to the
            // blocked accumulation of CPU BLAS/oneDNN that the oracle runs on.
            f32x16 part[TM][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int v = 0; v < 16; ++v) part[i][j][v] = 0.f;
#pragma unroll
            for (int kk = 0; kk < BK; kk += 8) {
                f32x4 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[i] = frag_f32<OpA::layout, GA::PITCH>(reinterpret_cast<const float*>(sA), wm + 32 * i, kk, lane);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[j] = frag_f32<OpB::layout, GB::PITCH>(reinterpret_cast<const float*>(sB), wn + 32 * j, kk, lane);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            part[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], part[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] += part[i][j];
        }
        __syncthreads();
    }

    // C/D map of the 32x32 MFMA shape: column = lane & 31, row = (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5)
    const int cj = lane & 31, rh = 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int64_t n = n0 + wn + 32 * j + cj;
            if (n >= N) continue;
            int64_t col_base = 0;
            if constexpr (NCHW) col_base = nchw_col_base(epi, n);
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int64_t m = m0 + wm + 32 * i + (v & 3) + 8 * (v >> 2) + rh;
                if (m >= M) continue;
                if (slabs) {
                    if (epi_in.sk_counters) sk_store(sk_rsrc(slabs, epi_in.sk_bytes), ((int64_t)bz * M + m) * N + n, acc[i][j][v]);
                    else slabs[((int64_t)bz * M + m) * N + n] = acc[i][j][v];
                }
                else if constexpr (NCHW) epi_store_nchw(epi, m, col_base, acc[i][j][v]);
                else epi_store(epi, m, n, acc[i][j][v]);
            }
        }
    if (slabs && epi_in.sk_counters) {
        const int splits = epi_in.sk_splits;
        const unsigned pb = (unsigned)(bz / (unsigned)splits);                 // problem of a batched launch (0 otherwise)
        if (!sk_arrive_last(epi_in, (pb * gridDim.y + by) * gridDim.x + bx, reinterpret_cast<volatile unsigned*>(smem))) return;
        const int64_t total = M * N;
        const auto rs = sk_rsrc(slabs, epi_in.sk_bytes);
        const int64_t first = (int64_t)pb * splits * total;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int64_t n = n0 + wn + 32 * j + cj;
                if (n >= N) continue;
                int64_t col_base = 0;
                if constexpr (NCHW) col_base = nchw_col_base(epi, n);
#pragma unroll 4
                for (int v = 0; v < 16; ++v) {
                    const int64_t m = m0 + wm + 32 * i + (v & 3) + 8 * (v >> 2) + rh;
                    if (m >= M) continue;
                    const int64_t q = first + m * N + n;
                    float sum = 0.f;
#pragma unroll 4
                    for (int s = 0; s < splits; ++s) sum += sk_load(rs, q + (int64_t)s * total);        // split order: reproducible
                    if constexpr (NCHW) epi_store_nchw(epi, m, col_base, sum);
                    else epi_store(epi, m, n, sum);
                }
            }
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* slabs, int splits, int64_t M, int64_t N, Epi epi_in) {
    const int64_t total = M * N;
    slabs += (int64_t)blockIdx.y * splits * total;      // blockIdx.y = problem of a batched GEMM (0 otherwise)
    const Epi epi = epi_for_batch(epi_in, blockIdx.y);
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        float v = 0.f;
        for (int s = 0; s < splits; ++s) v += slabs[(int64_t)s * total + idx];      // fixed order: reproducible
        const int64_t m = idx / N, n = idx - m * N;
        if (epi.nchw_hw > 0) epi_store_nchw(epi, m, nchw_col_base(epi, n), v);
        else epi_store(epi, m, n, v);
    }
}

struct Plan { int bm, bn, splits; int64_t k_tiles_per_split; int batch = 1; };

template <int CT> constexpr int bk_of() { return CT != VS_F32 ? 64 : 16; }

Plan make_plan(int compute, int64_t M, int64_t N, int64_t K, int64_t batch = 1, bool forward_layout = false) {
    const int bk = compute != VS_F32 ? 64 : 16;
    Plan p;
    // Tile choice (measured on the config-2 shapes, tools/gemm_bench.py): the kernel keeps ~3 workgroups (12 waves) per CU
    // busy; with fewer than ~4 tiles of 128x128 per CU most SIMDs hold a single wave that cannot overlap its LDS reads with
    // MFMA and 128x64 wins (421 vs 312 TF/s at 3328x4096x1200); 64x64 wins when K is short (310 vs 172 TF/s at K = 256);
    // few-tile problems go to split-K, where larger tiles mean fewer fp32 slabs.
    const int64_t t128 = vs_cdiv(M, 128) * vs_cdiv(N, 128);
    const int64_t t12864 = vs_cdiv(M, 128) * vs_cdiv(N, 64);
    // 128x64 only from ~2.3 tiles per CU upwards.  Below that the step is faster with 64x64 tiles although the isolated kernel is
    // not (WaveEq B=128, whole recorded step: 1.51 -> 1.43 ms; 3328x1200 outputs are 494 tiles of 128x64 but 988 of 64x64, and
    // the 256x1200 encoder outputs 38 against 76): the launches overlap with the gradient branches, where more and lighter
    // workgroups fill the CUs the neighbours leave.  VS_GEMM_T64_BELOW moves the threshold (0 = the round-1 rule).
    static const int64_t t64_below_any = getenv("VS_GEMM_T64_BELOW") ? atoll(getenv("VS_GEMM_T64_BELOW")) : 600;
    // R x R operands = a Linear layer's FORWARD launch: nothing runs beside the decoder's forward chain (the gradient branches that made the
    // light tiles win exist in backward only), so the threshold may differ there (VS_GEMM_T64_BELOW_RR)
    static const int64_t t64_below_rr = getenv("VS_GEMM_T64_BELOW_RR") ? atoll(getenv("VS_GEMM_T64_BELOW_RR")) : t64_below_any;
    const int64_t t64_below = forward_layout ? t64_below_rr : t64_below_any;
    if (const char* f = getenv("VS_GEMM_TILE")) {                       // debugging aid: force a tile ("128x128", "128x64", "64x64")
        p.bm = atoi(f); const char* x = strchr(f, 'x'); p.bn = x ? atoi(x + 1) : p.bm;
    } else if (K <= 512 && t128 >= 256) { p.bm = 64; p.bn = 64; }      // short K: prologue/epilogue bound, many small tiles win
    else if (t128 >= 1024) { p.bm = 128; p.bn = 128; }                 // >= 4 big tiles per CU: best LDS reuse
    else if ((t12864 >= 160 || vs_cdiv(M, 64) * vs_cdiv(N, 64) < 256) && M > 64 && t12864 >= t64_below) { p.bm = 128; p.bn = 64; }
    else { p.bm = 64; p.bn = 64; }
    if (M <= 64) p.bm = 64;
    if (N <= 64) p.bn = 64;
    if (p.bm == 64) p.bn = 64;
    // <= 64 output rows x very many columns (64-channel convolution layers over a whole batch of pixels): a 64x128 tile gives
    // every wave two accumulators per A fragment (1.5 LDS fragment reads per MFMA instead of 2)
    if (compute != VS_F32 && !getenv("VS_GEMM_TILE") && M <= 64 && M > 32 && vs_cdiv(N, 128) >= 1024 && K >= 128) { p.bm = 64; p.bn = 128; }
    const int64_t tiles = vs_cdiv(M, p.bm) * vs_cdiv(N, p.bn) * batch;
    const int64_t kt = vs_cdiv(K, bk);
    int splits = 1;
    if (tiles < 192 && kt >= 8) {
        splits = (int)((512 + tiles - 1) / tiles);
        const int64_t max_by_k = kt / 4;            // keep >= 4 K tiles per split
        if (splits > max_by_k) splits = (int)max_by_k;
        if (splits > 64) splits = 64;
        if (splits < 1) splits = 1;
    } else if (tiles < 1024 && kt >= 128) {
        // long reductions over few tiles (convolution weight gradients: K = batch x pixels up to ~10^6): one workgroup per CU
        // walking thousands of K tiles is latency bound; aim at ~1024 workgroups, >= 32 K tiles each
        splits = (int)((1024 + tiles - 1) / tiles);
        const int64_t max_by_k = kt / 32;
        if (splits > max_by_k) splits = (int)max_by_k;
        if (splits > 64) splits = 64;
        if (splits < 1) splits = 1;
    }
    p.k_tiles_per_split = vs_cdiv(kt, splits);
    p.splits = (int)vs_cdiv(kt, p.k_tiles_per_split);
    return p;
}

}  // namespace
