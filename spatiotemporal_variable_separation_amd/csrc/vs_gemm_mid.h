// vs_gemm_mid.h -- 128x128 output tile, 4 waves, the 4-deep LDS-DMA ring and fenced instruction order of vs_gemm_big.h at half
// the tile edge: the 16-bit GEMM kernel for problems whose 256x256 tiles would leave most of the chip idle (the 1200-wide layers
// of the WaveEq model: 3328 x 1200 is 65 tiles of 256x256 but 260 of 128x128) and, with split-K, for the long reductions over few
// tiles (256 x 1200 x 20480, the encoders' first layer).
//   * 256 threads = 4 waves as 2 (M) x 2 (N); a wave owns 64 x 64 of C (2 x 2 accumulators of the 32x32 shape, 64 registers).
//   * ring slot = [A rows 0-127 | B rows 0-127] x 32 k = 16 KiB, four slots = 64 KiB: two workgroups per CU, which are not coupled
//     by a barrier and fill each other's DMA / barrier gaps.
//   * a half-tile image is 512 pieces of 16 bytes, so every thread requests two pieces per operand and K tile (4 DMA requests per
//     thread and tile, as in the 256-wide kernel: the same vmcnt arithmetic), through the scalar-base form of the request.
//   * images, swizzles and fragment reads are those of vs_gemm_big.h (big_frag); the epilogue stages a wave's 64 x 64 block through
//     its quarter of the ring in one pass.
#pragma once
#include "vs_gemm_big.h"

namespace {

constexpr int MID_TILE_BYTES = 2 * 8192;          // A | B

// counted wait with the count as a template constant (s_waitcnt takes immediates)
template <int N>
__device__ __forceinline__ void vs_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N>
__device__ __forceinline__ void vs_wait_vm_lgkm0() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

// Per-thread offsets of one operand's two pieces (u = tid and tid + 256 of the 8 KiB half-tile image); rows past the end are
// clamped (see BigOperand), k past K goes through stage_checked().
template <int LAYOUT>
struct MidOperand {
    const char* base;                    // UNIFORM: first element of the next K tile
    uint32_t voff[2];
    int kofs[2];
    int64_t step;

    __device__ __forceinline__ void prepare(const unsigned short* p, int64_t ld, int64_t rows, int64_t i0, int64_t k_begin) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int u = (int)threadIdx.x + 256 * q;
            if (LAYOUT == LR) {
                const int row = u >> 2, piece = (u & 3) ^ ((row >> 2) & 3);
                int64_t r = i0 + row;
                if (r > rows - 1) r = rows - 1;
                kofs[q] = piece * 8;
                voff[q] = (uint32_t)(((r - i0) * ld + piece * 8) * 2);
            } else {
                const int k = u >> 4, piece = (u & 15) ^ ((k & 3) << 2);
                int64_t c = i0 + piece * 8;
                if (c > rows - 8) c = rows - 8;
                kofs[q] = k;
                voff[q] = (uint32_t)((k * ld + (c - i0)) * 2);
            }
        }
        if (LAYOUT == LR) {
            base = reinterpret_cast<const char*>(p + i0 * ld + k_begin);
            step = BIG_BK * 2;
        } else {
            base = reinterpret_cast<const char*>(p + k_begin * ld + i0);
            step = BIG_BK * ld * 2;
        }
        run_tiles = 0; t_in_run = 0; run_jump = 0;
    }
    // "channel rows" (LAYOUT R only): element (m = channel, k = (image b, pixel)) of an NCHW tensor at (b * C + m) * HW + pixel, i.e. an
    // R operand with leading dimension HW whose K axis comes in runs of HW (one image) with a jump of (C - 1) * HW between them --
    // the dy operand of a convolution's weight gradient.  HW % 32 == 0: a K tile never straddles two images.
    int run_tiles, t_in_run;             // K tiles per run (0: plain dense operand), position inside the current run
    int64_t run_jump;                    // extra bytes at the end of a run

    __device__ __forceinline__ void prepare_runs(const unsigned short* p, int64_t C, int64_t HW, int64_t i0, int64_t k_begin) {
        prepare(p, HW, C, i0, 0);
        const int64_t b = k_begin / HW, r = k_begin - b * HW;
        base = reinterpret_cast<const char*>(p + (b * C + i0) * HW + r);
        run_tiles = (int)(HW / BIG_BK);
        t_in_run = (int)(r / BIG_BK);
        run_jump = (C * HW - HW) * 2;
    }
    __device__ __forceinline__ void advance() {
        base += step;
        if (run_tiles && ++t_in_run == run_tiles) { t_in_run = 0; base += run_jump; }
    }
    __device__ __forceinline__ void stage(int q, char* lds) const {
        const uint32_t dst = (uint32_t)(uintptr_t)lds;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff[q]), "s"(base) : "memory", "m0");
    }
    __device__ __forceinline__ void stage_checked(int q, char* lds, int64_t k0, int64_t K, bool live) const {
        const void* g = (live && k0 + kofs[q] < K) ? (const void*)(base + voff[q]) : (const void*)vs_glds_zero;
        const uint32_t dst = (uint32_t)(uintptr_t)lds;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(g) : "memory", "m0");
    }
};

// ST ring slots of 16 KiB: 5 (80 KiB) where two workgroups share a CU, 10 (all 160 KiB) where the launch has at most one per CU --
// what a CU pulls through LDS-DMA is (bytes in flight) / (1.3-2 us), so the ring is as deep as the LDS allows.
template <int CT, int LA, int LB, bool NCHW, int ST, bool ADAM>
__global__ __launch_bounds__(256) void gemm_mid_kernel(const unsigned short* Ap, int64_t lda, const unsigned short* Bp, int64_t ldb, int64_t M, int64_t N,
                                                          int64_t K, int k_tiles_per_split, int tiles_n, Epi epi_in, float* slabs, int64_t a_chan_hw) {
    // XCD runs over (split, tile): the workgroups of one XCD share K chunks, so a chunk of A and of B is fetched into ONE L2
    unsigned tile, bz = blockIdx.z;
    if (epi_in.xcd_runs) {
        const unsigned t = big_tile_of(blockIdx.x + gridDim.x * blockIdx.z, gridDim.x * gridDim.z);
        bz = t / gridDim.x;
        tile = t - bz * gridDim.x;
    } else {
        tile = big_tile_of(blockIdx.x, gridDim.x);
    }
    int zsplit = bz;
    int batch = 0;
    if (epi_in.splits_per_batch > 0) {
        batch = bz / epi_in.splits_per_batch;
        zsplit = bz - batch * epi_in.splits_per_batch;
        Ap += batch * epi_in.batch_a;
        Bp += batch * epi_in.batch_b;
    }
    const Epi epi = epi_for_batch(epi_in, batch);
    extern __shared__ __attribute__((aligned(16))) char smem[];      // the ONLY LDS object: ST x [A | B] x 8 KiB

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wr = wave >> 1, wc = wave & 1;
    const int64_t m0 = (int64_t)(tile / (unsigned)tiles_n) * 128, n0 = (int64_t)(tile % (unsigned)tiles_n) * 128;
    const int64_t kt_total = (K + BIG_BK - 1) / BIG_BK;
    const int64_t kt_begin = (int64_t)zsplit * k_tiles_per_split;
    int64_t kt_end = kt_begin + k_tiles_per_split;
    if (kt_end > kt_total) kt_end = kt_total;
    int64_t kt_full = K / BIG_BK;
    if (kt_full > kt_end) kt_full = kt_end;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    MidOperand<LA> ga;
    MidOperand<LB> gb;
    if (LA == LR && a_chan_hw > 0) ga.prepare_runs(Ap, M, a_chan_hw, m0, kt_begin * BIG_BK);      // A = channel rows of an NCHW tensor (M channels)
    else ga.prepare(Ap, lda, M, m0, kt_begin * BIG_BK);
    gb.prepare(Bp, ldb, N, n0, kt_begin * BIG_BK);

    // piece q of a tile: 0, 1 = A pieces tid / tid + 256, 2, 3 = B; destination = image byte u * 16 = wave * 1024 (+ 4096 for q odd)
    char* const my_piece = smem + wave * 1024;
    auto dst_of = [&](int slot, int q) { return my_piece + slot * MID_TILE_BYTES + (q >> 1) * 8192 + (q & 1) * 4096; };
    auto stage_full = [&](int slot, int q) {
        if (q < 2) ga.stage(q, dst_of(slot, q)); else gb.stage(q - 2, dst_of(slot, q));
        if (q == 1) ga.advance();
        if (q == 3) gb.advance();
    };
    int64_t kt = kt_begin;
    auto stage_any = [&](int slot, int q) {
        const int64_t k4 = kt + ST;
        const bool live = k4 < kt_end;
        if (q < 2) ga.stage_checked(q, dst_of(slot, q), k4 * BIG_BK, K, live); else gb.stage_checked(q - 2, dst_of(slot, q), k4 * BIG_BK, K, live);
        if (q == 1) ga.advance();
        if (q == 3) gb.advance();
    };

    struct Frags { u32x4 a[2], b[2]; };
    Frags f0, f1, f2;
    auto rd_a = [&](Frags& f, int slot_, int kk, int i) {
        f.a[i] = big_frag<LA>(reinterpret_cast<const unsigned short*>(smem + slot_ * MID_TILE_BYTES), wr * 64 + 32 * i, kk, lane);
    };
    auto rd_b = [&](Frags& f, int slot_, int kk, int j) {
        f.b[j] = big_frag<LB>(reinterpret_cast<const unsigned short*>(smem + slot_ * MID_TILE_BYTES + 8192), wc * 64 + 32 * j, kk, lane);
    };
    auto mf = [&](const Frags& f, int i, int j) { acc[i][j] = mfma16_32<CT>(f.a[i], f.b[j], acc[i][j]); };

    kt = kt_begin - ST;
    for (int s4 = 0; s4 < ST; ++s4) {
        for (int q = 0; q < 4; ++q) stage_any(s4, q);
        ++kt;
    }
    vs_wait_vm<4 * (ST - 1)>();
    __builtin_amdgcn_s_barrier();
    for (int i = 0; i < 2; ++i) { rd_a(f0, 0, 0, i); rd_a(f1, 0, 16, i); }
    for (int j = 0; j < 2; ++j) { rd_b(f0, 0, 0, j); rd_b(f1, 0, 16, j); }
    int slot = 0;
    // X, Y: fragments of tile kt (k-steps 0 / 1), Z: free set.  One DMA request and one fragment read per MFMA.
#define VS_MID_BODY(X, Y, Z, STAGE)                                                                   \
    {                                                                                                 \
        mf(X, 0, 0); VS_FENCE;                                                                        \
        vs_wait_vm_lgkm0<4 * (ST - 2)>();           /* tile kt+1 landed; my reads of tile kt are done */  \
        __builtin_amdgcn_s_barrier();                                                                 \
        VS_FENCE;                                                                                     \
        const int nslot = slot + 1 == ST ? 0 : slot + 1;                                              \
        STAGE(slot, 0); rd_a(Z, nslot, 0, 0); VS_FENCE;                                               \
        mf(X, 0, 1); VS_FENCE; rd_a(Z, nslot, 0, 1); VS_FENCE;                                        \
        mf(X, 1, 0); VS_FENCE; STAGE(slot, 1); rd_b(Z, nslot, 0, 0); VS_FENCE;                        \
        mf(X, 1, 1); VS_FENCE; rd_b(Z, nslot, 0, 1); VS_FENCE;                                        \
        mf(Y, 0, 0); VS_FENCE; STAGE(slot, 2); rd_a(X, nslot, 16, 0); VS_FENCE;                       \
        mf(Y, 0, 1); VS_FENCE; rd_a(X, nslot, 16, 1); VS_FENCE;                                       \
        mf(Y, 1, 0); VS_FENCE; STAGE(slot, 3); rd_b(X, nslot, 16, 0); VS_FENCE;                       \
        mf(Y, 1, 1); VS_FENCE; rd_b(X, nslot, 16, 1); VS_FENCE;                                       \
        slot = nslot;                                                                                 \
        ++kt;                                                                                         \
    }
    {
        int64_t n3 = (kt_full - ST - kt_begin) / 3;
        for (; n3 > 0; --n3) {
            VS_MID_BODY(f0, f1, f2, stage_full)
            VS_MID_BODY(f2, f0, f1, stage_full)
            VS_MID_BODY(f1, f2, f0, stage_full)
        }
    }
    while (kt < kt_end) {
        VS_MID_BODY(f0, f1, f2, stage_any)
        if (kt >= kt_end) break;
        VS_MID_BODY(f2, f0, f1, stage_any)
        if (kt >= kt_end) break;
        VS_MID_BODY(f1, f2, f0, stage_any)
    }
#undef VS_MID_BODY
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: accumulators -> LDS (row-major 64 x 64 fp32 per wave) -> coalesced row stores ----
    float* stg = reinterpret_cast<float*>(smem) + wave * (64 * 64);          // 16 KiB per wave, the whole 64 KiB ring
    const int cj = lane & 31, rh = 4 * (lane >> 5);
    float* slab_base = slabs ? slabs + (int64_t)bz * M * N : nullptr;
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v)
                stg[(32 * ii + (v & 3) + 8 * (v >> 2) + rh) * 64 + 32 * j + cj] = acc[ii][j][v];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // own writes only: a wave reads back what it wrote itself
    const int64_t nn = n0 + wc * 64 + (lane & 15) * 4;
    if constexpr (ADAM) {
        // fused optimizer (vs_gemm_adam; an instantiation of its own, so that profiles tell it from the plain GEMM): the tile is
        // a block of the parameter's gradient
        if (epi.adam_guard && *epi.adam_guard != 0u) return;          // an exchange of this step timed out: no update (vs_common.h)
        const AdamCoef coef = vs_adam_coef(epi.adam_lr, epi.adam_beta1, epi.adam_beta2, epi.adam_eps, (double)(epi.adam_step[0] + 1 - epi.adam_skipped));
        // The update is a pure HBM stream (24 B in, 26 B out per parameter) and a lane's 16 row pieces are independent: the state of FOUR
        // pieces (12 x 16 B per lane) is requested before the previous four are updated and stored, so a wave keeps 12 KiB of reads in
        // flight instead of 3 (one piece at a time ran the 640 MB launches of the WaveEq encoders at 3.3 TB/s: latency x bytes in flight).
        // (whole tiles only -- no per-row branch: a branch around a request makes the compiler wait for EVERY outstanding load, vmcnt(0),
        // at the next use, which serialises the groups again)
        if (epi.adam_pipe && m0 + 128 <= M && nn + 3 < N && ((epi.ldc | nn) & 3) == 0 &&
            (((uintptr_t)epi.C | (uintptr_t)epi.adam_m | (uintptr_t)epi.adam_v) & 15) == 0 && (!epi.adam_shadow || ((uintptr_t)epi.adam_shadow & 7) == 0)) {
            constexpr int G = 4;
            f32x4 p[2][G], mm[2][G], vv[2][G];
            const int64_t idx0 = (m0 + wr * 64 + (lane >> 4)) * epi.ldc + nn;           // row piece (g, t) lies (g * G + t) * 4 rows further
            const int64_t rstep = 4 * epi.ldc;
            const float* __restrict__ Pp = (const float*)epi.C + idx0;
            const float* __restrict__ Mp = epi.adam_m + idx0;
            const float* __restrict__ Vp = epi.adam_v + idx0;
            const int sdt = epi.adam_shadow_dtype;
            // adam_pipe & 2: the fp32 master and the two moments are touched once per step -- non-temporal loads and stores (they do not displace
            // what the step re-reads from the L2s / the Infinity Cache); the 16-bit copy, read by the next forward pass, keeps the default policy
            const bool nt = (epi.adam_pipe & 2) != 0;
            auto ld4 = [&](const float* q) { return nt ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q)) : *reinterpret_cast<const f32x4*>(q); };
            auto st4 = [&](float* q, const f32x4& v) { if (nt) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(q)); else *reinterpret_cast<f32x4*>(q) = v; };
#pragma unroll
            for (int t = 0; t < G; ++t) {
                p[0][t] = ld4(Pp + t * rstep);
                mm[0][t] = ld4(Mp + t * rstep);
                vv[0][t] = ld4(Vp + t * rstep);
            }
#pragma unroll
            for (int g = 0; g < 16 / G; ++g) {
                if (g + 1 < 16 / G) {
#pragma unroll
                    for (int t = 0; t < G; ++t) {
                        const int64_t o = ((g + 1) * G + t) * rstep;
                        p[(g + 1) & 1][t] = ld4(Pp + o);
                        mm[(g + 1) & 1][t] = ld4(Mp + o);
                        vv[(g + 1) & 1][t] = ld4(Vp + o);
                    }
                }
#pragma unroll
                for (int t = 0; t < G; ++t) {
                    const int r = (g * G + t) * 4 + (lane >> 4);
                    const f32x4 g4 = *reinterpret_cast<const f32x4*>(stg + r * 64 + (lane & 15) * 4);
                    f32x4 pe = p[g & 1][t], me = mm[g & 1][t], ve = vv[g & 1][t];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float a = pe[e], b = me[e], c = ve[e];
                        vs_adam_elem(coef, g4[e] * epi.alpha, a, b, c);
                        pe[e] = a; me[e] = b; ve[e] = c;
                    }
                    const int64_t idx = idx0 + (g * G + t) * rstep;
                    st4((float*)epi.C + idx, pe);
                    st4(epi.adam_m + idx, me);
                    st4(epi.adam_v + idx, ve);
                    if (epi.adam_shadow) {
                        const u16x4 h = {vs_f2h(pe[0], sdt), vs_f2h(pe[1], sdt), vs_f2h(pe[2], sdt), vs_f2h(pe[3], sdt)};
                        *reinterpret_cast<u16x4*>(epi.adam_shadow + idx) = h;
                    }
                }
            }
            return;
        }
        for (int it = 0; it < 16; ++it) {
            const int r = it * 4 + (lane >> 4);
            const int64_t m = m0 + wr * 64 + r;
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(stg + r * 64 + (lane & 15) * 4);
            if (m < M && nn < N) big_adam4(epi, coef, m, nn, N, v4);
        }
        return;
    }
    for (int it = 0; it < 16; ++it) {
        const int r = it * 4 + (lane >> 4);
        const int64_t m = m0 + wr * 64 + r;
        const f32x4 v4 = *reinterpret_cast<const f32x4*>(stg + r * 64 + (lane & 15) * 4);
        if (m < M && nn < N) big_store4<NCHW>(epi, m, nn, N, v4, slab_base ? slab_base + m * N : nullptr, epi_in.sk_counters ? slabs : nullptr, epi_in.sk_bytes);
    }
    if (slab_base && epi_in.sk_counters) {
        // split-K finished in the launch: the last workgroup of this tile to arrive adds the slabs (split order) and applies the epilogue
        const int splits = epi_in.sk_splits;
        if (!sk_arrive_last(epi_in, (unsigned)batch * gridDim.x + tile, reinterpret_cast<volatile unsigned*>(smem))) return;
        const int64_t total = M * N;
        const auto rs = sk_rsrc(slabs, epi_in.sk_bytes);
        const int64_t first = (int64_t)batch * splits * total;
        const bool vec = nn + 3 < N && (((uintptr_t)slabs | (uintptr_t)(total * 4)) & 15) == 0 && (N & 3) == 0;
        for (int it = 0; it < 16; ++it) {
            const int64_t m = m0 + wr * 64 + it * 4 + (lane >> 4);
            if (m >= M || nn >= N) continue;
            const int64_t q = first + m * N + nn;
            f32x4 sum = {0.f, 0.f, 0.f, 0.f};
            if (vec) {
#pragma unroll 8
                for (int s = 0; s < splits; ++s) sum += sk_load4(rs, q + (int64_t)s * total);
            } else {
                for (int s = 0; s < splits; ++s)
                    for (int t = 0; t < 4; ++t)
                        if (nn + t < N) sum[t] += sk_load(rs, q + (int64_t)s * total + t);
            }
            big_store4<NCHW>(epi, m, nn, N, sum, nullptr);
        }
    }
}

// launch (shared by vs_gemm.hip and the convolution weight gradients of vs_conv.hip); a_chan_hw > 0: A is the channel-rows view of an
// NCHW tensor with M channels and planes of a_chan_hw elements (lda is ignored)
template <int CT, int LA, int LB, bool NCHW = false>
int mid_launch(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, int splits, int64_t k_tiles_per_split,
               int stages, int batch, const Epi& epi, float* slabs, hipStream_t stream, int64_t a_chan_hw = 0) {
    if constexpr (CT == VS_F32) {
        return vs_fail(VS_ERR_UNSUPPORTED, "the 128x128 LDS-DMA ring tile is a 16-bit kernel");
    } else {
        const int tiles_m = (int)vs_cdiv(M, 128), tiles_n = (int)vs_cdiv(N, 128);
        dim3 grid((unsigned)(tiles_m * tiles_n), 1, (unsigned)(splits * batch));
        // several splits: the XCD runs go over (split, tile) -- the 256 x 1200 x 20480 encoder layer (20 tiles x 22 splits) fetched 147 MB for
        // 60 MB of operands with the splits of one K chunk spread over all eight L2s.  VS_GEMM_XCD=0: runs over the tiles of each split only.
        static const int xcd_mode = getenv("VS_GEMM_XCD") ? atoi(getenv("VS_GEMM_XCD")) : 1;
        Epi e_runs = epi;
        e_runs.xcd_runs = xcd_mode && grid.z > 1;
        const Epi& epi_l = e_runs;
        auto go = [&](auto kfn, int st_, bool& attr_set) -> int {
            const int lds = st_ * MID_TILE_BYTES;
            if (!attr_set) {                           // above the 64 KiB default limit of dynamic LDS
                if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
                    return vs_fail(VS_ERR_LAUNCH, "cannot raise the dynamic LDS limit to %d bytes", lds);
                attr_set = true;
            }
            hipLaunchKernelGGL(kfn, grid, dim3(256), lds, stream, (const unsigned short*)A, lda, (const unsigned short*)B, ldb, M, N, K,
                               (int)k_tiles_per_split, tiles_n, epi_l, slabs, a_chan_hw);
            return VS_OK;
        };
        static bool set5 = false, set10 = false, set_adam = false;
        if constexpr (NCHW) {
            return stages == 10 ? go(gemm_mid_kernel<CT, LA, LB, true, 10, false>, 10, set10) : go(gemm_mid_kernel<CT, LA, LB, true, 5, false>, 5, set5);
        } else {
            return epi.adam_m ? go(gemm_mid_kernel<CT, LA, LB, false, 5, true>, 5, set_adam)
                   : stages == 10 ? go(gemm_mid_kernel<CT, LA, LB, false, 10, false>, 10, set10)
                                  : go(gemm_mid_kernel<CT, LA, LB, false, 5, false>, 5, set5);
        }
    }
}

// ---- when to take it ---------------------------------------------------------------------------------------------------------
// k_tiles_per_split counts K tiles of BIG_BK.  VS_GEMM_MID=0 disables, =2 takes it whenever the operands allow (tests).
struct MidPlan { bool use; int splits; int64_t k_tiles_per_split; int tiles_m, tiles_n; int stages; };

inline MidPlan make_mid_plan(int compute, int64_t M, int64_t N, int64_t K, int64_t batch) {
    MidPlan p{false, 1, 0, (int)vs_cdiv(M, 128), (int)vs_cdiv(N, 128), 5};
    const char* env = getenv("VS_GEMM_MID");                      // read per call: tests switch it
    const int mode = env ? atoi(env) : 1;
    if (compute == VS_F32 || mode == 0) return p;
    const int64_t kt = vs_cdiv(K, BIG_BK);
    const int64_t tiles = (int64_t)p.tiles_m * p.tiles_n * batch;
    p.k_tiles_per_split = kt;
    if (mode == 2) { p.use = tiles <= 65535 * 4; p.stages = tiles <= 256 ? 10 : 5; return p; }
    if (M < 128 || N < 128 || kt < 16) return p;                   // short K: prologue / epilogue bound, the small tiles win
    const double fill = (double)M * (double)N / ((double)p.tiles_m * 128.0 * (double)p.tiles_n * 128.0);
    if (fill < 0.75) return p;
    // two workgroups per CU are resident: up to 512 in one round.  Few tiles and a long K: split so that ~one round is filled
    // and every split keeps >= 12 K tiles (prologue + epilogue + the slab round trip cost ~8 tiles' worth).
    int splits = 1;
    if (tiles < 200 && kt >= 24) {
        splits = (int)(448 / tiles);
        const int64_t max_by_k = kt / 12;
        if (splits > max_by_k) splits = (int)max_by_k;
        if (splits > 32) splits = 32;
        if (splits < 1) splits = 1;
    }
    // Taken where it wins INSIDE the WaveEq step, not only alone (same-box A/B of the replayed step): problems of 280..520
    // workgroups -- 4096 x 1200 x 3328 (320 tiles: 54 vs 81 us alone, 70 vs 140 us under the integrator's backward kernel),
    // 1200 x 1200 x 3328 split in 4, 256 x 1200 x 20480 split in 22.  At 260 workgroups (3328 x 1200 x K: one per CU) it is
    // 33.8 vs 35.3 us alone but 54 vs 45 us in the step, where its 80 KiB of LDS per workgroup keeps neighbours off the CU.
    if (tiles * splits < 280 || tiles * splits > 520) return p;
    p.k_tiles_per_split = vs_cdiv(kt, splits);
    p.splits = (int)vs_cdiv(kt, p.k_tiles_per_split);
    p.stages = tiles * p.splits <= 256 ? 10 : 5;
    p.use = true;
    return p;
}

}  // namespace
