// vs_gemm_glds.h -- bf16 128x128 GEMM tile staged by LDS-DMA (global_load_lds_dwordx4), gfx950 only.
//
// Same contraction, operand layouts, split-K and epilogues as gemm_kernel (vs_gemm_core.h); what differs is how a K tile
// reaches LDS.  The register-staged loop moves every 16-byte unit global -> VGPR -> ds_write_b128 and needs its pads for
// conflict-free reads; here each wave instruction copies 1 KiB straight into LDS (no VGPRs, no ds_write pass), which frees
// the registers for a 64x64 accumulator block per wave (1.0 fragment reads per MFMA instead of 1.5 at 128x64) at the same
// occupancy.  The DMA destination is lane-linear (wave base + lane*16), so tiles are UNPADDED and bank conflicts are
// avoided by permuting 16-byte pieces: the permutation is applied to the per-lane SOURCE address and again to the read
// address (cdna_hip_programming.md rule 21).
//   R tile [128 rows][64 k]  (128-byte rows, two rows per 256-byte bank row): piece p of row r sits in slot p ^ ((r >> 1) & 7);
//                            a 16-lane group of a ds_read_b128 (16 consecutive rows, one k piece) covers all 16 slots.
//   S tile [64 k][128 rows]  (256-byte k-rows = one bank row each): piece p of k-row k sits in slot p ^ ((k & 3) << 2);
//                            the four k-rows of a ds_read_b64_tr_b16 land in four different 64-byte bank groups.
// Pieces outside the matrix (row tail, K tail) are fetched from a 16-byte block of zeros instead: the per-lane source
// address is the only thing LDS-DMA lets a lane choose.  Requirements (checked by the launcher, else the register-staged
// kernel runs): 16-byte aligned base, leading dimension % 8 == 0, contiguous extent % 8 == 0 (K for R, rows for S).
// STAGES = 1: one LDS buffer, two barriers per K tile, 3 workgroups per CU overlap each other (best when the tiles of a
// problem fit one round of 768 slots); STAGES = 2: two buffers, prefetch across the barrier, 2 workgroups per CU (best
// for >= 1024 tiles: 711 vs 588 TF/s at 4096^3).
#pragma once
#include "vs_gemm_core.h"

namespace {

__device__ __attribute__((aligned(16))) const uint32_t vs_glds_zero[4] = {0u, 0u, 0u, 0u};

typedef __attribute__((address_space(3))) void glds_lds_ptr;
typedef __attribute__((address_space(1))) const void glds_glb_ptr;

// Per-thread source pointers of one operand tile: 4 DMA rounds of 256 lanes x 16 B = 16 KiB.
template <int LAYOUT>
struct GldsOperand {
    const __bf16* src[4];     // piece of K tile 0 (advanced by one K tile per step)
    int kofs[4];              // k of the piece inside the tile (R: first of its 8 k; S: its k-row)
    bool ok[4];               // row(s) inside the matrix
    int64_t step;             // elements to advance per K tile

    __device__ __forceinline__ void prepare(const __bf16* p, int64_t ld, int64_t rows, int64_t i0, int64_t k_begin) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int u = r * 256 + (int)threadIdx.x;              // linear 16-byte slot of the LDS image
            if (LAYOUT == LR) {
                const int row = u >> 3, piece = (u & 7) ^ ((row >> 1) & 7);
                ok[r] = i0 + row < rows;
                kofs[r] = piece * 8;
                src[r] = p + (i0 + row) * ld + k_begin + piece * 8;
                step = 64;
            } else {
                const int k = u >> 4, piece = (u & 15) ^ ((k & 3) << 2);
                ok[r] = i0 + piece * 8 < rows;                     // rows % 8 == 0: a piece is inside or outside as a whole
                kofs[r] = k;
                src[r] = p + (k_begin + k) * ld + i0 + piece * 8;
                step = 64 * ld;
            }
        }
    }
    // issue the 4 DMA pieces of the K tile starting at k0 (K = reduction extent) into `lds` (16 KiB, wave-linear)
    __device__ __forceinline__ void stage(char* lds, int64_t k0, int64_t K) {
        const int wave = threadIdx.x >> 6;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const void* g = (ok[r] && k0 + kofs[r] < K) ? (const void*)src[r] : (const void*)vs_glds_zero;
            __builtin_amdgcn_global_load_lds((glds_glb_ptr*)g, (glds_lds_ptr*)(lds + (r * 256 + wave * 64) * 16), 16, 0, 0);
            src[r] += step;
        }
    }
};

template <int LAYOUT>
__device__ __forceinline__ u32x4 glds_frag(const unsigned short* tile, int row0, int kk, int lane) {
    if (LAYOUT == LR) {
        const int row = row0 + (lane & 31), q = (kk >> 3) + (lane >> 5);
        return *reinterpret_cast<const u32x4*>(tile + row * 64 + ((q ^ ((row >> 1) & 7)) << 3));
    } else {
        const int li = lane & 15, q = li >> 2, p = li & 3, cb = (lane >> 4) & 1, h = lane >> 5;
        const int k1 = kk + 8 * h + q;                             // k1 + 4 has the same (k & 3): same permutation
        const int rowoff = row0 + 16 * cb + 4 * p;
        const int slot = (rowoff >> 3) ^ ((k1 & 3) << 2);
        const unsigned short* a = tile + k1 * 128 + slot * 8 + (rowoff & 7);
        return vs_tr16_pair(a, 4 * 128);
    }
}

template <int LA, int LB, bool NCHW, int STAGES = 1, int CT = VS_BF16>
__global__ __launch_bounds__(256) void gemm_glds_kernel(const __bf16* Ap, int64_t lda, const __bf16* Bp, int64_t ldb, int64_t M, int64_t N,
                                                        int64_t K, int k_tiles_per_split, Epi epi_in, float* slabs) {
    int zsplit = blockIdx.z;
    int batch = 0;
    if (epi_in.splits_per_batch > 0) {
        batch = blockIdx.z / epi_in.splits_per_batch;
        zsplit = blockIdx.z - batch * epi_in.splits_per_batch;
        Ap += batch * epi_in.batch_a;
        Bp += batch * epi_in.batch_b;
    }
    const Epi epi = epi_for_batch(epi_in, batch);
    extern __shared__ __attribute__((aligned(16))) char smem[];      // the ONLY LDS object: [A tile 16 KiB][B tile 16 KiB]
    char* sA = smem;
    char* sB = smem + 16384;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int64_t m0 = (int64_t)blockIdx.y * 128, n0 = (int64_t)blockIdx.x * 128;
    const int64_t kt_total = (K + 63) / 64;
    const int64_t kt_begin = (int64_t)zsplit * k_tiles_per_split;
    int64_t kt_end = kt_begin + k_tiles_per_split;
    if (kt_end > kt_total) kt_end = kt_total;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    GldsOperand<LA> ga;
    GldsOperand<LB> gb;
    ga.prepare(Ap, lda, M, m0, kt_begin * 64);
    gb.prepare(Bp, ldb, N, n0, kt_begin * 64);

    // fragments of k-step s+1 are requested before the MFMAs of k-step s are issued (two register sets, static names): with
    // one or two waves per SIMD nothing else hides the ds_read latency
    auto compute = [&](const char* tA, const char* tB) {
        const unsigned short* pa = reinterpret_cast<const unsigned short*>(tA);
        const unsigned short* pb = reinterpret_cast<const unsigned short*>(tB);
        u32x4 a0[2], b0[2], a1[2], b1[2];
#define VS_GLDS_LOAD(fa, fb, kk)                                               \
        fa[0] = glds_frag<LA>(pa, wm, kk, lane); fa[1] = glds_frag<LA>(pa, wm + 32, kk, lane); \
        fb[0] = glds_frag<LB>(pb, wn, kk, lane); fb[1] = glds_frag<LB>(pb, wn + 32, kk, lane);
#define VS_GLDS_MFMA(fa, fb)                                                   \
        acc[0][0] = mfma16_32<CT>(fa[0], fb[0], acc[0][0]); \
        acc[0][1] = mfma16_32<CT>(fa[0], fb[1], acc[0][1]); \
        acc[1][0] = mfma16_32<CT>(fa[1], fb[0], acc[1][0]); \
        acc[1][1] = mfma16_32<CT>(fa[1], fb[1], acc[1][1]);
        VS_GLDS_LOAD(a0, b0, 0)
        VS_GLDS_LOAD(a1, b1, 16)
        VS_GLDS_MFMA(a0, b0)
        VS_GLDS_LOAD(a0, b0, 32)
        VS_GLDS_MFMA(a1, b1)
        VS_GLDS_LOAD(a1, b1, 48)
        VS_GLDS_MFMA(a0, b0)
        VS_GLDS_MFMA(a1, b1)
#undef VS_GLDS_LOAD
#undef VS_GLDS_MFMA
    };
    if constexpr (STAGES == 1) {
        for (int64_t kt = kt_begin; kt < kt_end; ++kt) {
            ga.stage(sA, kt * 64, K);
            gb.stage(sB, kt * 64, K);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            compute(sA, sB);
            __syncthreads();
        }
    } else {
        // two LDS buffers: the DMA of K tile t+1 stays in flight across the barrier while tile t is multiplied; one barrier per
        // K tile; tile t+1 goes into the buffer tile t-1 was read from, which the barrier has just released.  (A deeper ring
        // -- 4 buffers, counted vmcnt(16) -- was measured SLOWER: 128 KiB of LDS leaves one wave per SIMD and the
        // ds_read -> MFMA chain of a single wave is then the bottleneck: 483 vs 711 TF/s at 4096^3.)
        if (kt_begin < kt_end) {
            ga.stage(smem, kt_begin * 64, K);
            gb.stage(smem + 16384, kt_begin * 64, K);
        }
        int cur = 0;
        for (int64_t kt = kt_begin; kt < kt_end; ++kt) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();              // tile kt landed for every wave; every wave is done reading tile kt-1
            if (kt + 1 < kt_end) {
                ga.stage(smem + (cur ^ 1) * 32768, (kt + 1) * 64, K);
                gb.stage(smem + (cur ^ 1) * 32768 + 16384, (kt + 1) * 64, K);
            }
            compute(smem + cur * 32768, smem + cur * 32768 + 16384);
            cur ^= 1;
        }
    }

    const int cj = lane & 31, rh = 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t n = n0 + wn + 32 * j + cj;
            if (n >= N) continue;
            int64_t col_base = 0;
            if constexpr (NCHW) col_base = nchw_col_base(epi, n);
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int64_t m = m0 + wm + 32 * i + (v & 3) + 8 * (v >> 2) + rh;
                if (m >= M) continue;
                if (slabs) slabs[((int64_t)blockIdx.z * M + m) * N + n] = acc[i][j][v];
                else if constexpr (NCHW) epi_store_nchw(epi, m, col_base, acc[i][j][v]);
                else epi_store(epi, m, n, acc[i][j][v]);
            }
        }
}

// operand fits the LDS-DMA tile loader?
inline bool glds_operand_ok(const void* p, int64_t ld, int layout, int64_t rows, int64_t K, int64_t batch_stride) {
    return (uintptr_t)p % 16 == 0 && ld % 8 == 0 && batch_stride % 8 == 0 && (layout == LR ? K % 8 == 0 : rows % 8 == 0);
}

}  // namespace
