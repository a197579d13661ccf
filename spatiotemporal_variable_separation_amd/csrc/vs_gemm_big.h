// vs_gemm_big.h -- 256x256x64 tile, 8 waves, LDS-DMA double buffered: the 16-bit GEMM kernel for problems whose 256x256 tiles
// (x split-K) fit ONE round of the 256 CUs (decoder layers of the WaveEq model: 3328 x 4096 x 1200 = 208 tiles).
//
// Why a second kernel: the 128x64 / 128x128 tiles of gemm_kernel / gemm_glds_kernel need 1.0-1.5 KiB of LDS fragment reads per
// 32x32x16 MFMA and saturate the CU's LDS port (256 B/clk) long before the matrix pipes; they also quantise badly on this model
// (832 / 494 / 260 tiles on 256 CUs leave a mostly empty last round).  Here every wave owns a 128 x 64 block of C (4 x 2
// accumulators of the 32x32 shape = 128 registers): 6 fragment reads feed 8 MFMAs (0.75 KiB per MFMA, 37 % of the LDS port), and
// one workgroup per CU holds the whole 128 KiB double buffer.
//   * 512 threads = 8 waves as 2 (M) x 4 (N), two waves per SIMD: while one wave of a SIMD waits for its fragments the other
//     issues MFMAs;
//   * a K tile (256 x 64 of A and of B) is four 16 KiB half-tiles [A rows 0-127 | A rows 128-255 | B rows 0-127 | B rows 128-255],
//     each filled by two global_load_lds_dwordx4 per thread (1 KiB per wave instruction, no VGPRs); the half-tile images are the
//     swizzled, unpadded ones of vs_gemm_glds.h (R: piece ^ ((row >> 1) & 7); S: piece ^ ((k & 3) << 2), read with
//     ds_read_b64_tr_b16), so all four operand layouts are served and every LDS read is conflict free;
//   * K loop: one barrier per K tile.  The four half-tiles of tile t+1 are requested right after the barrier that releases their
//     buffer, so their DMA is in flight while tile t is multiplied; `s_waitcnt vmcnt(0)` + raw `s_barrier` ends the tile;
//     fragments of k-step s+1 are requested before the MFMAs of k-step s (two register sets);
//   * blockIdx -> tile map groups the tiles of one XCD (blocks b, b+8, ...) into a contiguous run of the row-major tile order,
//     so the A row panel of a run stays in that XCD's L2;
//   * split-K, the epilogue (bias / activation / mask / NCHW) and the slab reduction are those of gemm_kernel.
#pragma once
#include "vs_gemm_glds.h"

namespace {

// Per-thread source pointers of one operand's two half-tiles (2 DMA pieces per half): advanced by one K tile per stage call.
template <int LAYOUT>
struct BigOperand {
    const unsigned short* src[2][2];     // [half][round]
    int kofs[2];                         // k of the piece inside the tile (same for both halves)
    bool ok[2][2];
    int64_t step;

    __device__ __forceinline__ void prepare(const unsigned short* p, int64_t ld, int64_t rows, int64_t i0, int64_t k_begin) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int u = r * 512 + (int)threadIdx.x;               // linear 16-byte slot of the 16 KiB half-tile image
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int64_t base = i0 + 128 * h;
                if (LAYOUT == LR) {
                    const int row = u >> 3, piece = (u & 7) ^ ((row >> 1) & 7);
                    ok[h][r] = base + row < rows;
                    kofs[r] = piece * 8;
                    src[h][r] = p + (base + row) * ld + k_begin + piece * 8;
                    step = 64;
                } else {
                    const int k = u >> 4, piece = (u & 15) ^ ((k & 3) << 2);
                    ok[h][r] = base + piece * 8 < rows;
                    kofs[r] = k;
                    src[h][r] = p + (k_begin + k) * ld + base + piece * 8;
                    step = 64 * ld;
                }
            }
        }
    }
    // issue the 2 DMA pieces of half `h` of the K tile starting at k0 into `lds` (16 KiB, wave-linear) and advance
    __device__ __forceinline__ void stage(int h, char* lds, int64_t k0, int64_t K) {
        const int wave = threadIdx.x >> 6;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const void* g = (ok[h][r] && k0 + kofs[r] < K) ? (const void*)src[h][r] : (const void*)vs_glds_zero;
            __builtin_amdgcn_global_load_lds((glds_glb_ptr*)g, (glds_lds_ptr*)(lds + (r * 512 + wave * 64) * 16), 16, 0, 0);
            src[h][r] += step;
        }
    }
};

// XCD-aware, bijective block -> tile index (blocks b and b + 8 share an XCD under round-robin dispatch: speed only)
__device__ __forceinline__ unsigned big_tile_of(unsigned b, unsigned nwg) {
    const unsigned xcd = b & 7u, q = nwg >> 3, r = nwg & 7u;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

template <int CT, int LA, int LB, bool NCHW>
__global__ __launch_bounds__(512) void gemm_big_kernel(const unsigned short* Ap, int64_t lda, const unsigned short* Bp, int64_t ldb, int64_t M, int64_t N,
                                                       int64_t K, int k_tiles_per_split, int tiles_n, Epi epi_in, float* slabs) {
    int zsplit = blockIdx.z;
    int batch = 0;
    if (epi_in.splits_per_batch > 0) {
        batch = blockIdx.z / epi_in.splits_per_batch;
        zsplit = blockIdx.z - batch * epi_in.splits_per_batch;
        Ap += batch * epi_in.batch_a;
        Bp += batch * epi_in.batch_b;
    }
    const Epi epi = epi_for_batch(epi_in, batch);
    extern __shared__ __attribute__((aligned(16))) char smem[];      // the ONLY LDS object: 2 x [A0 | A1 | B0 | B1] x 16 KiB

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const unsigned tile = big_tile_of(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(tile / (unsigned)tiles_n) * 256, n0 = (int64_t)(tile % (unsigned)tiles_n) * 256;
    const int64_t kt_total = (K + 63) / 64;
    const int64_t kt_begin = (int64_t)zsplit * k_tiles_per_split;
    int64_t kt_end = kt_begin + k_tiles_per_split;
    if (kt_end > kt_total) kt_end = kt_total;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    BigOperand<LA> ga;
    BigOperand<LB> gb;
    ga.prepare(Ap, lda, M, m0, kt_begin * 64);
    gb.prepare(Bp, ldb, N, n0, kt_begin * 64);

    // half-tile q of buffer `buf`: q = 0, 1 -> A rows 0-127 / 128-255; q = 2, 3 -> B rows 0-127 / 128-255
    auto half_ptr = [&](int buf, int q) -> char* { return smem + buf * 65536 + q * 16384; };
    auto stage_half = [&](int buf, int q, int64_t kt) {
        if (q < 2) ga.stage(q, half_ptr(buf, q), kt * 64, K);
        else gb.stage(q - 2, half_ptr(buf, q), kt * 64, K);
    };

    if (kt_begin < kt_end) {
#pragma unroll
        for (int q = 0; q < 4; ++q) stage_half(0, q, kt_begin);
    }
    int cur = 0;
    for (int64_t kt = kt_begin; kt < kt_end; ++kt) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // tile kt landed for every wave; every wave is done reading tile kt - 1
        const unsigned short* pa = reinterpret_cast<const unsigned short*>(half_ptr(cur, wr));
        const unsigned short* pb = reinterpret_cast<const unsigned short*>(half_ptr(cur, 2 + (wc >> 1)));
        const int bcol = (wc & 1) * 64;
        const bool more = kt + 1 < kt_end;
        u32x4 a0[4], b0[2], a1[4], b1[2];
#define VS_BIG_LOAD(fa, fb, kk)                                                      \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) fa[i] = glds_frag<LA>(pa, 32 * i, kk, lane); \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) fb[j] = glds_frag<LB>(pb, bcol + 32 * j, kk, lane);
#define VS_BIG_MFMA(fa, fb)                                                          \
        __builtin_amdgcn_s_setprio(1);                                               \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = mfma16_32<CT>(fa[i], fb[j], acc[i][j]); \
        __builtin_amdgcn_s_setprio(0);
        // the whole next tile is requested NOW: its DMA has this tile's 32 MFMAs per wave (x 2 waves per SIMD) to land.  (Issued
        // one half-tile per k-step, the last half had only one k-step of cover and every tile ended in a ~1 us vmcnt stall:
        // 108 us instead of 76 us at 3328 x 4096 x 1200.)
        if (more) {
            stage_half(cur ^ 1, 0, kt + 1);
            stage_half(cur ^ 1, 1, kt + 1);
        }
        VS_BIG_LOAD(a0, b0, 0)
        if (more) {
            stage_half(cur ^ 1, 2, kt + 1);
            stage_half(cur ^ 1, 3, kt + 1);
        }
        VS_BIG_LOAD(a1, b1, 16)
        VS_BIG_MFMA(a0, b0)
        VS_BIG_LOAD(a0, b0, 32)
        VS_BIG_MFMA(a1, b1)
        VS_BIG_LOAD(a1, b1, 48)
        VS_BIG_MFMA(a0, b0)
        VS_BIG_MFMA(a1, b1)
#undef VS_BIG_LOAD
#undef VS_BIG_MFMA
        cur ^= 1;
    }

    // C/D map of the 32x32 MFMA shape: column = lane & 31, row = (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5)
    const int cj = lane & 31, rh = 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t n = n0 + wc * 64 + 32 * j + cj;
            if (n >= N) continue;
            int64_t col_base = 0;
            if constexpr (NCHW) col_base = nchw_col_base(epi, n);
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int64_t m = m0 + wr * 128 + 32 * i + (v & 3) + 8 * (v >> 2) + rh;
                if (m >= M) continue;
                if (slabs) slabs[((int64_t)blockIdx.z * M + m) * N + n] = acc[i][j][v];
                else if constexpr (NCHW) epi_store_nchw(epi, m, col_base, acc[i][j][v]);
                else epi_store(epi, m, n, acc[i][j][v]);
            }
        }
}

// ---- when to take it ---------------------------------------------------------------------------------------------------------
// One workgroup per CU.  The plan asks for at most ONE round of 256 workgroups: tiles x splits <= 256, >= 6 K tiles per split
// (amortises prologue + epilogue), and enough work that the 256-wide tile is not mostly padding.  VS_GEMM_BIG=0 disables,
// =2 takes it whenever the operands allow (tests).
struct BigPlan { bool use; int splits; int64_t k_tiles_per_split; int tiles_m, tiles_n; };

inline BigPlan make_big_plan(int compute, int64_t M, int64_t N, int64_t K, int64_t batch) {
    BigPlan p{false, 1, 0, (int)vs_cdiv(M, 256), (int)vs_cdiv(N, 256)};
    static const int mode = getenv("VS_GEMM_BIG") ? atoi(getenv("VS_GEMM_BIG")) : 1;
    if (compute == VS_F32 || mode == 0) return p;
    const int64_t kt = vs_cdiv(K, 64);
    const int64_t tiles = (int64_t)p.tiles_m * p.tiles_n * batch;
    p.k_tiles_per_split = kt;
    if (mode == 2) { p.use = tiles <= 65535; return p; }
    if (M < 512 || N < 512 || kt < 6 || tiles > 256) return p;
    // padding waste of the 256-wide tiles
    const double fill = (double)M * (double)N / ((double)p.tiles_m * 256.0 * (double)p.tiles_n * 256.0);
    if (fill < 0.8) return p;
    int splits = (int)(256 / tiles);
    const int64_t max_by_k = kt / 6;
    if (splits > max_by_k) splits = (int)max_by_k;
    if (splits < 1) splits = 1;
    if (splits > 16) splits = 16;
    // a split costs a slab round trip (4 B written + read per output element and split): only split when the CUs would
    // otherwise idle for more than that costs
    if (tiles >= 160) splits = 1;
    p.k_tiles_per_split = vs_cdiv(kt, splits);
    p.splits = (int)vs_cdiv(kt, p.k_tiles_per_split);
    p.use = true;
    return p;
}

}  // namespace
