// vs_gemm_big.h -- 256x256 output tile, 8 waves, K in steps of 32 through a 4-deep LDS-DMA ring: the 16-bit GEMM kernel for
// problems whose 256x256 tiles (x split-K) fit ONE round of the 256 CUs (decoder layers of the WaveEq model: 3328 x 4096 x 1200
// = 208 tiles).
//
// Why a second kernel: the 128x64 / 128x128 tiles of gemm_kernel / gemm_glds_kernel need 1.0-1.5 KiB of LDS fragment reads per
// 32x32x16 MFMA and saturate the CU's LDS port (256 B/clk) long before the matrix pipes; they also quantise badly on this model
// (832 / 494 / 260 tiles on 256 CUs leave a mostly empty last round).  Here every wave owns a 128 x 64 block of C (4 x 2
// accumulators of the 32x32 shape = 128 registers): 6 fragment reads feed 8 MFMAs (0.75 KiB per MFMA, 37 % of the LDS port), and
// one workgroup per CU owns the whole 128 KiB ring.
//   * 512 threads = 8 waves as 2 (M) x 4 (N), two waves per SIMD.
//   * Operand movement is what bounds a one-workgroup-per-CU GEMM on MI355X: a CU pulls ~25-70 GB/s through LDS-DMA depending on
//     where the bytes sit and how many are in flight (MI355X_MICROARCH: ldsdma-fill, indexed rows), and a first version with two
//     64-deep buffers (ONE tile in flight, drained to vmcnt(0) every tile) ran at the latency of one 64 KiB burst per tile:
//     3.1 us per K tile, 300 TFLOP/s.  So: K tiles of 32 (32 KiB: A 256 x 32 + B 256 x 32), FOUR ring slots, the DMA of tiles
//     t+1 .. t+3 in flight while tile t is multiplied, counted `s_waitcnt vmcnt(8)` (never 0 inside the loop) + raw `s_barrier`
//     once per tile; a tile is requested right after the barrier that frees its slot.  Tiles past the end of the split are
//     requested from a block of zeros so the count stays uniform.
//   * a K tile is four 8 KiB half-tiles [A rows 0-127 | A rows 128-255 | B rows 0-127 | B rows 128-255], one
//     global_load_lds_dwordx4 per thread each (1 KiB per wave instruction, no VGPRs).  Unpadded, swizzled images (the DMA
//     destination is lane-linear, so the permutation sits on the SOURCE address and on the read address):
//       R [128 rows][32 k]: 64-byte rows, piece p (8 k) of row r in slot p ^ ((r >> 2) & 3): the 16 lanes of a ds_read_b128 group
//                           (rows r..r+3, r+12..15, r+20..27 of one piece) cover all 16 slots of the 256-byte bank row;
//       S [32 k][128 rows]: 256-byte k-rows, piece p (8 rows) of k-row k in slot p ^ ((k & 3) << 2), read with ds_read_b64_tr_b16
//                           (the four k-rows of a transposing read sit in four different 64-byte bank groups).
//   * epilogue through LDS: the accumulators (column on the lane) are laid out row-major in the (now idle) ring, 64 rows per wave
//     at a time, and read back as float4 along the rows: every store instruction writes whole 256-byte row segments, and the
//     epilogue arithmetic (bias / activation / mask / accumulate / 16-bit convert) exists once, in a loop, instead of 128
//     unrolled scalar copies (which the compiler answered by spilling the accumulators to scratch).
//   * blockIdx -> tile map groups the tiles of one XCD (blocks b, b+8, ...) into a contiguous run of the row-major tile order,
//     so the A row panel of a run stays in that XCD's L2.
//   * split-K and the slab reduction are those of gemm_kernel.
#pragma once
#include <type_traits>
#include "vs_gemm_glds.h"

namespace {

constexpr int BIG_BK = 32;
constexpr int BIG_STAGES = 4;                     // the tap convolutions' ring (vs_conv_tap.hip)
// the GEMM's ring: five slots = all 160 KiB of the CU.  What a CU pulls through LDS-DMA is (bytes in flight) / (latency), and the
// operands of a large GEMM stream from HBM / the Infinity Cache at 1.3-2 us per request under load: four tiles (128 KiB) in flight
// instead of three.
#define VS_GB_ST 5
#define VS_GB_WAIT_FIRST 16                       // 4 * (ST - 1): tile 0 has landed
#define VS_GB_WAIT_LOOP 12                        // 4 * (ST - 2): tile kt + 1 has landed
#define VS_STR_(x) #x
#define VS_STR(x) VS_STR_(x)
static_assert(VS_GB_WAIT_FIRST == 4 * (VS_GB_ST - 1) && VS_GB_WAIT_LOOP == 4 * (VS_GB_ST - 2), "vmcnt immediates follow the ring depth");
constexpr int GEMM_BIG_STAGES = VS_GB_ST;
constexpr int BIG_TILE_BYTES = 4 * 8192;          // A0 | A1 | B0 | B1

// Per-thread source pointer of one operand's two half-tiles (one 16-byte DMA piece per half and K tile).  Rows past the end of
// the operand are CLAMPED to its last row / last 8-row piece (valid memory): what they contribute lands in output rows or
// columns that are never stored, and the request needs no per-lane select.  Only k past K must read zeros (it meets valid rows
// of the other operand): the one partial K tile of a problem goes through stage_checked().
template <int LAYOUT>
struct BigOperand {
    const char* base;                    // UNIFORM: first element of the next K tile (advances by `step` bytes per tile)
    uint32_t voff[2];                    // [half]: this lane's byte offset from `base` (fixed for the whole launch)
    int kofs;                            // k of the piece inside the tile
    int64_t step;

    __device__ __forceinline__ void prepare(const unsigned short* p, int64_t ld, int64_t rows, int64_t i0, int64_t k_begin) {
        const int u = (int)threadIdx.x;                         // linear 16-byte slot of the 8 KiB half-tile image
        // offsets are taken from the tile's first row / column so that they fit 32 bits (256 rows x ld x 2 B < 4 GiB: ld < 2^23)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t first = i0 + 128 * h;
            if (LAYOUT == LR) {
                const int row = u >> 2, piece = (u & 3) ^ ((row >> 2) & 3);
                int64_t r = first + row;
                if (r > rows - 1) r = rows - 1;
                kofs = piece * 8;
                voff[h] = (uint32_t)(((r - i0) * ld + piece * 8) * 2);
            } else {
                const int k = u >> 4, piece = (u & 15) ^ ((k & 3) << 2);
                int64_t c = first + piece * 8;                  // rows % 8 == 0: a piece is inside or outside as a whole
                if (c > rows - 8) c = rows - 8;
                kofs = k;
                voff[h] = (uint32_t)((k * ld + (c - i0)) * 2);
            }
        }
        if (LAYOUT == LR) {
            base = reinterpret_cast<const char*>(p + i0 * ld + k_begin);
            step = BIG_BK * 2;
        } else {
            base = reinterpret_cast<const char*>(p + k_begin * ld + i0);
            step = BIG_BK * ld * 2;
        }
    }
    __device__ __forceinline__ void advance() { base += step; }
    // request half `h` of a FULL K tile into `lds` (this wave's 1 KiB of the 8 KiB image): scalar base + 32-bit lane offset.
    // Written out (the builtin takes a flat pointer and costs two 64-bit VALU adds per request): M0 = LDS address of the wave's
    // piece, `global_load_lds_dwordx4 voffset, sbase`.  The request counts on vmcnt like any load; the loop waits by hand.
    __device__ __forceinline__ void stage(int h, char* lds) const {
        const uint32_t dst = (uint32_t)(uintptr_t)lds;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff[h]), "s"(base) : "memory", "m0");
    }
    // the same for a tile that may be partial (k0 + 32 > K) or past the end of the split (`live` false): zeros there
    __device__ __forceinline__ void stage_checked(int h, char* lds, int64_t k0, int64_t K, bool live) const {
        const void* g = (live && k0 + kofs < K) ? (const void*)(base + voff[h]) : (const void*)vs_glds_zero;
        const uint32_t dst = (uint32_t)(uintptr_t)lds;           // (asm as well: M0 has ONE writer in this kernel)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(g) : "memory", "m0");
    }
};

// fragment of one 32-row block and one 16-deep k-step (kk = 0 or 16) of a half-tile image
template <int LAYOUT>
__device__ __forceinline__ u32x4 big_frag(const unsigned short* tile, int row0, int kk, int lane) {
    if (LAYOUT == LR) {
        const int row = row0 + (lane & 31), q = (kk >> 3) + (lane >> 5);
        return *reinterpret_cast<const u32x4*>(tile + row * 32 + ((q ^ ((row >> 2) & 3)) << 3));
    } else {
        const int li = lane & 15, q = li >> 2, p = li & 3, cb = (lane >> 4) & 1, h = lane >> 5;
        const int k1 = kk + 8 * h + q;                             // k1 + 4 has the same (k & 3): same permutation
        const int rowoff = row0 + 16 * cb + 4 * p;
        const int slot = (rowoff >> 3) ^ ((k1 & 3) << 2);
        return vs_tr16_pair(tile + k1 * 128 + slot * 8 + (rowoff & 7), 4 * 128);
    }
}

// (big_tile_of, the XCD-aware block -> tile index: vs_gemm_core.h)

// four consecutive columns n..n+3 of row m; vector I/O when the row-major addresses allow it
// the Adam update of four consecutive parameters (row m, columns n..n+3) with the accumulators as their gradient
__device__ __forceinline__ void big_adam4(const Epi& e, const AdamCoef& c, int64_t m, int64_t n, int64_t N, const f32x4& g) {
    const int64_t idx = m * e.ldc + n;
    float* P = (float*)e.C + idx;
    float* Mo = e.adam_m + idx;
    float* Vo = e.adam_v + idx;
    if (n + 3 < N && (e.ldc & 3) == 0 && (n & 3) == 0) {
        f32x4 p = *reinterpret_cast<const f32x4*>(P), mm = *reinterpret_cast<const f32x4*>(Mo), vv = *reinterpret_cast<const f32x4*>(Vo);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float pe = p[t], me = mm[t], ve = vv[t];
            vs_adam_elem(c, g[t] * e.alpha, pe, me, ve);
            p[t] = pe; mm[t] = me; vv[t] = ve;
        }
        *reinterpret_cast<f32x4*>(P) = p;
        *reinterpret_cast<f32x4*>(Mo) = mm;
        *reinterpret_cast<f32x4*>(Vo) = vv;
        if (e.adam_shadow) {
            const u16x4 h = {vs_f2h(p[0], e.adam_shadow_dtype), vs_f2h(p[1], e.adam_shadow_dtype), vs_f2h(p[2], e.adam_shadow_dtype),
                             vs_f2h(p[3], e.adam_shadow_dtype)};
            *reinterpret_cast<u16x4*>(e.adam_shadow + idx) = h;
        }
    } else {
        for (int t = 0; t < 4; ++t)
            if (n + t < N) {
                float pe = P[t], me = Mo[t], ve = Vo[t];
                vs_adam_elem(c, g[t] * e.alpha, pe, me, ve);
                P[t] = pe; Mo[t] = me; Vo[t] = ve;
                if (e.adam_shadow) e.adam_shadow[idx + t] = vs_f2h(pe, e.adam_shadow_dtype);
            }
    }
}

template <bool NCHW>
__device__ __forceinline__ void big_store4(const Epi& e, int64_t m, int64_t n, int64_t N, const f32x4& v, float* slab_row, const float* through = nullptr,
                                           int64_t through_bytes = 0) {
    if (slab_row) {                                               // split-K partial: raw fp32, reduced (with the epilogue) later
        if (through) {                                            // ... by the last workgroup of this launch: written through (sk_store)
            const auto rs = sk_rsrc(through, through_bytes);      // `through` = start of the whole slab area
            const int64_t e = (slab_row + n) - through;
            if (n + 3 < N && ((uintptr_t)(slab_row + n) & 15) == 0) sk_store4(rs, e, v);
            else
                for (int t = 0; t < 4; ++t)
                    if (n + t < N) sk_store(rs, e + t, v[t]);
            return;
        }
        if (n + 3 < N && ((uintptr_t)(slab_row + n) & 15) == 0) *reinterpret_cast<f32x4*>(slab_row + n) = v;
        else
            for (int t = 0; t < 4; ++t)
                if (n + t < N) slab_row[n + t] = v[t];
        return;
    }
    if constexpr (NCHW) {
        // rows = channels, columns = pixels (b * hw + pix): four consecutive pixels of one plane are contiguous in NCHW
        if (n + 3 < N && e.g_hw == 0 && (e.nchw_hw & 3) == 0 && (n & 3) == 0 && !e.mask && !e.accumulate && ((uintptr_t)e.C & 15) == 0) {
            const int64_t idx = nchw_col_base(e, n) + m * e.nchw_hw;
            const float bsv = e.bias ? e.bias[m] : 0.f;
            f32x4 r;
#pragma unroll
            for (int t = 0; t < 4; ++t) r[t] = vs_act(v[t] * e.alpha + bsv, e.act);
            if (e.c_dtype == VS_F32) {
                *reinterpret_cast<f32x4*>((float*)e.C + idx) = r;
            } else {
                const u16x4 h = {vs_f2h(r[0], e.c_dtype), vs_f2h(r[1], e.c_dtype), vs_f2h(r[2], e.c_dtype), vs_f2h(r[3], e.c_dtype)};
                *reinterpret_cast<u16x4*>((unsigned short*)e.C + idx) = h;
            }
        } else {
            for (int t = 0; t < 4; ++t)
                if (n + t < N) epi_store_nchw(e, m, nchw_col_base(e, n + t), v[t]);
        }
    } else {
        const bool vec = n + 3 < N && ((e.ldc | n) & 3) == 0 && !e.mask && !e.accumulate && ((uintptr_t)e.C & 15) == 0;
        if (vec) {
            f32x4 r;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float x = v[t] * e.alpha;
                if (e.bias) x += e.bias[n + t];
                r[t] = vs_act(x, e.act);
            }
            if (e.c_dtype == VS_F32) {
                *reinterpret_cast<f32x4*>((float*)e.C + m * e.ldc + n) = r;
            } else {
                const u16x4 h = {vs_f2h(r[0], e.c_dtype), vs_f2h(r[1], e.c_dtype), vs_f2h(r[2], e.c_dtype), vs_f2h(r[3], e.c_dtype)};
                *reinterpret_cast<u16x4*>((unsigned short*)e.C + m * e.ldc + n) = h;
            }
        } else {
            for (int t = 0; t < 4; ++t)
                if (n + t < N) epi_store(e, m, n + t, v[t]);
        }
    }
}

// frame-loss epilogue (Epi::fl_*): address of the target of frame row m, columns n.. (m < M, n + 3 < N)
__device__ __forceinline__ const float* big_loss_target(const Epi& e, int64_t m, int64_t n, int64_t N, int t_rand, int& g_out) {
    const unsigned b = (unsigned)m / (unsigned)e.fl_G;                 // (rows < 2^31: the host checks; a 64-bit division is ~190 instructions)
    const int g = (int)((unsigned)m - b * (unsigned)e.fl_G);
    const int frame = g == 0 ? t_rand - e.fl_ae_shift : e.fl_first + g - 1;
    g_out = g;
    return e.fl_full + ((int64_t)b * e.fl_T + frame) * N + n;
}
// ... and what is done with four consecutive columns of the result once their targets `t` have arrived
__device__ __forceinline__ void big_loss4(const Epi& e, int64_t m, int64_t n, int64_t N, int g, const f32x4& v, const f32x4& t, const f32x4& bias4, float up,
                                          float& s0, float& s1) {
    const float k = g == 0 ? up * e.fl_l_ae * 2.f * e.fl_inv_ae : up * e.fl_l_pred * 2.f * e.fl_inv_pred;      // (as train_losses_fwd_kernel)
    float rr[4], s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float x = vs_act(v[j] * e.alpha + bias4[j], e.act);
        const float d = x - t[j];
        s += d * d;
        rr[j] = (k * d) * vs_act_grad_from_out(x, e.act);
    }
    if (g == 0) s0 += s; else s1 += s;
    const int64_t o = m * N + n;
    if (e.fl_dz_dtype == VS_F32) *reinterpret_cast<f32x4*>((float*)e.fl_dz + o) = f32x4{rr[0], rr[1], rr[2], rr[3]};
    else {
        const u16x4 w = {vs_f2h(rr[0], e.fl_dz_dtype), vs_f2h(rr[1], e.fl_dz_dtype), vs_f2h(rr[2], e.fl_dz_dtype), vs_f2h(rr[3], e.fl_dz_dtype)};
        *reinterpret_cast<u16x4*>((unsigned short*)e.fl_dz + o) = w;
    }
}

// Instruction order inside one K tile ("block") of the main loop is pinned in the source (sched_barrier(0) fences between the
// groups): an MFMA holds the SIMD's issue port for 8 of its 32 cycles, so the 12 fragment reads, the 4 DMA requests and their
// address arithmetic of a block are dealt out one or two per MFMA instead of ahead of them.  (All waves in step and everything
// issued ahead: DMA issue ~500 cycles with the matrix pipes idle, then the two waves of a SIMD queueing for the pipe -- 1550
// cycles per tile against the 1024 the 32 MFMAs of a SIMD need.  The scheduler left to itself, or steered with
// sched_group_barrier, clumps the reads behind the MFMAs.)
#define VS_FENCE __builtin_amdgcn_sched_barrier(0)

// LOSS: the frame-loss epilogue (Epi::fl_*, vs_gemm_frame_loss) -- an instantiation of its own: its registers must not cost the GEMM its
// spill-free allocation (236 VGPRs)
template <int CT, int LA, int LB, bool NCHW, bool LOSS = false>
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
    extern __shared__ __attribute__((aligned(16))) char smem[];      // the ONLY LDS object: STAGES x [A0 | A1 | B0 | B1] x 8 KiB

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // scalar: DMA destinations need no v_readfirstlane
    const int wr = wave >> 2, wc = wave & 3;
    const unsigned tile = big_tile_of(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(tile / (unsigned)tiles_n) * 256, n0 = (int64_t)(tile % (unsigned)tiles_n) * 256;
    const int64_t kt_total = (K + BIG_BK - 1) / BIG_BK;
    const int64_t kt_begin = (int64_t)zsplit * k_tiles_per_split;
    int64_t kt_end = kt_begin + k_tiles_per_split;
    if (kt_end > kt_total) kt_end = kt_total;
    int64_t kt_full = K / BIG_BK;                                    // tiles [kt_begin, kt_full) are full and live
    if (kt_full > kt_end) kt_full = kt_end;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;

    BigOperand<LA> ga;
    BigOperand<LB> gb;
    ga.prepare(Ap, lda, M, m0, kt_begin * BIG_BK);
    gb.prepare(Bp, ldb, N, n0, kt_begin * BIG_BK);

    // request piece `q` (A0 | A1 | B0 | B1) of K tile kt + 4 into ring slot `slot`: 4 DMA instructions per thread and tile, always
    // (the vmcnt arithmetic relies on it)
    char* const my_piece = smem + wave * 1024;
    auto stage_full = [&](int slot, int q) {
        char* base = my_piece + slot * BIG_TILE_BYTES + q * 8192;
        if (q < 2) ga.stage(q, base); else gb.stage(q - 2, base);
        if (q == 1) ga.advance();
        if (q == 3) gb.advance();
    };
    int64_t kt = kt_begin;                                             // the tile being multiplied
    auto stage_any = [&](int slot, int q) {
        char* base = my_piece + slot * BIG_TILE_BYTES + q * 8192;
        const int64_t k4 = kt + GEMM_BIG_STAGES;
        const bool live = k4 < kt_end;
        if (q < 2) ga.stage_checked(q, base, k4 * BIG_BK, K, live); else gb.stage_checked(q - 2, base, k4 * BIG_BK, K, live);
        if (q == 1) ga.advance();
        if (q == 3) gb.advance();
    };

    // Software pipeline across K tiles: the fragments of tile t+1 are read (and tile t+4 requested) WHILE the MFMAs of tile t
    // run.  Three fragment sets rotate: X = k-step 0, Y = k-step 1 of tile t; k-step 0 of tile t+1 goes to Z, k-step 1 to X once
    // the MFMAs of X are issued.  Ring: tile t+1 is being read, t+2 and t+3 are in flight, t+4 takes the slot of tile t, whose
    // fragments are in registers.
    struct Frags { u32x4 a[4], b[2]; };
    Frags f0, f1, f2;
    const int bcol = (wc & 1) * 64;
    const int a_off = wr * 8192, b_off = 16384 + (wc >> 1) * 8192;
    auto rd_a = [&](Frags& f, int slot_, int kk, int i) {
        f.a[i] = big_frag<LA>(reinterpret_cast<const unsigned short*>(smem + slot_ * BIG_TILE_BYTES + a_off), 32 * i, kk, lane);
    };
    auto rd_b = [&](Frags& f, int slot_, int kk, int j) {
        f.b[j] = big_frag<LB>(reinterpret_cast<const unsigned short*>(smem + slot_ * BIG_TILE_BYTES + b_off), bcol + 32 * j, kk, lane);
    };
    auto mf = [&](const Frags& f, int i, int j) { acc[i][j] = mfma16_32<CT>(f.a[i], f.b[j], acc[i][j]); };
    // prologue: tiles 0 .. 3 requested (zeros past the end), tile 0 read
    kt = kt_begin - GEMM_BIG_STAGES;                                   // stage_any requests tile kt + STAGES
    for (int s4 = 0; s4 < GEMM_BIG_STAGES; ++s4) {
        for (int q = 0; q < 4; ++q) stage_any(s4, q);
        ++kt;
    }
    asm volatile("s_waitcnt vmcnt(" VS_STR(VS_GB_WAIT_FIRST) ")" ::: "memory");   // tile 0 (this wave's pieces) ...
    __builtin_amdgcn_s_barrier();                                      // ... and everybody else's
    for (int i = 0; i < 4; ++i) { rd_a(f0, 0, 0, i); rd_a(f1, 0, 16, i); }
    for (int j = 0; j < 2; ++j) { rd_b(f0, 0, 0, j); rd_b(f1, 0, 16, j); }
    int slot = 0;                                                      // ring slot of tile kt
    // X, Y: fragments of tile kt (k-steps 0 / 1), Z: free set.  STAGE(q) requests piece q of tile kt + 4 into the slot of tile kt.
#define VS_BIG_BODY(X, Y, Z, STAGE)                                                                   \
    {                                                                                                 \
        /* two MFMAs are queued ahead of the barrier: a wave that arrives early leaves the pipe busy */ \
        mf(X, 0, 0); mf(X, 0, 1); VS_FENCE;                                                           \
        asm volatile("s_waitcnt vmcnt(" VS_STR(VS_GB_WAIT_LOOP) ") lgkmcnt(0)" ::: "memory"); /* tile kt+1 landed; my reads of tile kt done */ \
        __builtin_amdgcn_s_barrier();                                                                 \
        VS_FENCE;                                                                                     \
        const int nslot = slot + 1 == GEMM_BIG_STAGES ? 0 : slot + 1;                                 \
        STAGE(slot, 0); rd_a(Z, nslot, 0, 0); VS_FENCE;                                               \
        mf(X, 1, 0); VS_FENCE; rd_a(Z, nslot, 0, 1); VS_FENCE;                                        \
        mf(X, 1, 1); VS_FENCE; rd_a(Z, nslot, 0, 2); VS_FENCE;                                        \
        mf(X, 2, 0); VS_FENCE; STAGE(slot, 1); rd_a(Z, nslot, 0, 3); VS_FENCE;                        \
        mf(X, 2, 1); VS_FENCE; rd_b(Z, nslot, 0, 0); VS_FENCE;                                        \
        mf(X, 3, 0); VS_FENCE; rd_b(Z, nslot, 0, 1); VS_FENCE;                                        \
        mf(X, 3, 1); VS_FENCE;                                                                        \
        mf(Y, 0, 0); VS_FENCE; STAGE(slot, 2); rd_a(X, nslot, 16, 0); VS_FENCE;                       \
        mf(Y, 0, 1); VS_FENCE; rd_a(X, nslot, 16, 1); VS_FENCE;                                       \
        mf(Y, 1, 0); VS_FENCE; rd_a(X, nslot, 16, 2); VS_FENCE;                                       \
        mf(Y, 1, 1); VS_FENCE; STAGE(slot, 3); rd_a(X, nslot, 16, 3); VS_FENCE;                       \
        mf(Y, 2, 0); VS_FENCE; rd_b(X, nslot, 16, 0); VS_FENCE;                                       \
        mf(Y, 2, 1); VS_FENCE; rd_b(X, nslot, 16, 1); VS_FENCE;                                       \
        mf(Y, 3, 0);                                                                                  \
        mf(Y, 3, 1); VS_FENCE;                                                                        \
        slot = nslot;                                                                                 \
        ++kt;                                                                                         \
    }
    // main loop: a multiple of three blocks whose tile kt + 4 is full and live (no per-lane selects, no branch in the body)
    {
        int64_t n3 = (kt_full - GEMM_BIG_STAGES - kt_begin) / 3;
        for (; n3 > 0; --n3) {
            VS_BIG_BODY(f0, f1, f2, stage_full)
            VS_BIG_BODY(f2, f0, f1, stage_full)
            VS_BIG_BODY(f1, f2, f0, stage_full)
        }
    }
    // the remaining (>= 4) blocks: the partial K tile, if any, and the zero-source requests that keep the count uniform
    while (kt < kt_end) {
        VS_BIG_BODY(f0, f1, f2, stage_any)
        if (kt >= kt_end) break;
        VS_BIG_BODY(f2, f0, f1, stage_any)
        if (kt >= kt_end) break;
        VS_BIG_BODY(f1, f2, f0, stage_any)
    }
#undef VS_BIG_BODY
    // drain the (zero-source) requests still in flight before the ring is reused as the epilogue's staging area
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: accumulators -> LDS (row-major, 64 rows x 64 columns of fp32 per wave and pass) -> coalesced row stores ----
    float* stg = reinterpret_cast<float*>(smem) + wave * (64 * 64);          // 16 KiB per wave, 128 KiB in all
    const int cj = lane & 31, rh = 4 * (lane >> 5);
    float* slab_base = slabs ? slabs + (int64_t)blockIdx.z * M * N : nullptr;
    float fl_s0 = 0.f, fl_s1 = 0.f;
    float fl_up = 0.f;
    int fl_t = 0;
    if constexpr (LOSS) { fl_up = epi.fl_up[0]; fl_t = epi.fl_tdev[0]; }
    // (the two passes as calls with a COMPILE-TIME pass number: an index of `acc` the compiler cannot resolve would move all 128
    // accumulators to scratch memory -- seen with the frame-loss variant's inner lambda inside an unrolled `for`)
    auto epilogue_pass = [&](auto pass_c) {
        constexpr int pass = decltype(pass_c)::value;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    stg[(32 * ii + (v & 3) + 8 * (v >> 2) + rh) * 64 + 32 * j + cj] = acc[2 * pass + ii][j][v];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // own writes only: a wave reads back what it wrote itself
        const int64_t nn = n0 + wc * 64 + (lane & 15) * 4;
        if constexpr (LOSS) {
            // frame losses: a row piece needs 16 bytes of its TARGET frame from HBM; the requests of four pieces are in flight while the previous
            // four are worked off (one piece at a time the epilogue waited ~1 us per piece: +25 us on the WaveEq step).  Rows / columns past
            // the end read a clamped (valid) address -- no branch around a request (see gemm_mid_kernel's optimizer epilogue) -- and are skipped.
            f32x4 tg[2][4];
            int gg[2][4];
            const int64_t nn_c = nn + 3 < N ? nn : N - 4;
            // (a lane's four columns are the same for all its row pieces: the bias is loaded ONCE -- a load inside the loop is waited for
            // with vmcnt(0), i.e. together with every target request in flight)
            const f32x4 bias4 = epi.bias ? *reinterpret_cast<const f32x4*>(epi.bias + nn_c) : f32x4{0.f, 0.f, 0.f, 0.f};
            auto request = [&](int buf, int q) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    int64_t m = m0 + wr * 128 + 64 * pass + (q * 4 + t) * 4 + (lane >> 4);
                    if (m > M - 1) m = M - 1;
                    tg[buf][t] = *reinterpret_cast<const f32x4*>(big_loss_target(epi, m, nn_c, N, fl_t, gg[buf][t]));
                }
            };
            request(0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q + 1 < 4) request((q + 1) & 1, q + 1);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int r = (q * 4 + t) * 4 + (lane >> 4);
                    const int64_t m = m0 + wr * 128 + 64 * pass + r;
                    const f32x4 v4 = *reinterpret_cast<const f32x4*>(stg + r * 64 + (lane & 15) * 4);
                    if (m < M && nn + 3 < N) big_loss4(epi, m, nn, N, gg[q & 1][t], v4, tg[q & 1][t], bias4, fl_up, fl_s0, fl_s1);
                }
            }
        } else {
            for (int it = 0; it < 16; ++it) {
                const int r = it * 4 + (lane >> 4);
                const int64_t m = m0 + wr * 128 + 64 * pass + r;
                const f32x4 v4 = *reinterpret_cast<const f32x4*>(stg + r * 64 + (lane & 15) * 4);
                if (m < M && nn < N) big_store4<NCHW>(epi, m, nn, N, v4, slab_base ? slab_base + m * N : nullptr);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // reads done before the next pass overwrites the area
    };
    epilogue_pass(std::integral_constant<int, 0>{});
    epilogue_pass(std::integral_constant<int, 1>{});
    if constexpr (LOSS) {
        // the workgroup's two partial sums: lanes -> waves (shuffles) -> LDS (the staging area is idle after the barrier) -> one pair per tile
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { fl_s0 += __shfl_down(fl_s0, o, 64); fl_s1 += __shfl_down(fl_s1, o, 64); }
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
        if (lane == 0) { red[2 * wave] = fl_s0; red[2 * wave + 1] = fl_s1; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float a0 = 0.f, a1 = 0.f;
            for (int w = 0; w < 8; ++w) { a0 += red[2 * w]; a1 += red[2 * w + 1]; }
            epi.fl_partials[2 * blockIdx.x] = a0;
            epi.fl_partials[2 * blockIdx.x + 1] = a1;
        }
    }
}

// ---- when to take it ---------------------------------------------------------------------------------------------------------
// One workgroup per CU.  The plan asks for ONE round of 160..256 workgroups, >= 12 K tiles (of 32) (amortises prologue + epilogue:
// ~26 us of a 63 us launch at 3328 x 4096 x 1200 are launch, ring fill and the 54 MB of fp32 output), and enough work that the
// 256-wide tile is not mostly padding.  VS_GEMM_BIG=0 disables, =2 takes it whenever the operands allow (tests; split-K off).
// k_tiles_per_split counts K tiles of BIG_BK.  Measured (MI355X, bf16, random operands): 3328 x 4096 x 1200 63 us (128x64 tile:
// 76 us), 4096^3 158 us = 870 TFLOP/s (128x128 LDS-DMA tile: 184 us).  Timing-only variants of the loop (VS_BIG_DIAG): without
// the MFMAs and fragment reads the DMA ring alone runs at 0.57 us per 32 KiB tile (57 GB/s per CU), without real DMA traffic
// the multiply alone at 0.74 us per tile (MFMA-issue bound at the clock the chip holds under load), both together at 1.03 us.
struct BigPlan { bool use; int splits; int64_t k_tiles_per_split; int tiles_m, tiles_n; };

inline BigPlan make_big_plan(int compute, int64_t M, int64_t N, int64_t K, int64_t batch) {
    BigPlan p{false, 1, 0, (int)vs_cdiv(M, 256), (int)vs_cdiv(N, 256)};
    const char* env = getenv("VS_GEMM_BIG");                      // read per call: tests switch it
    const int mode = env ? atoi(env) : 1;
    if (compute == VS_F32 || mode == 0) return p;
    const int64_t kt = vs_cdiv(K, BIG_BK);
    const int64_t tiles = (int64_t)p.tiles_m * p.tiles_n * batch;
    p.k_tiles_per_split = kt;
    if (mode == 2) { p.use = tiles <= 65535; return p; }
    if (M < 512 || N < 512 || kt < 12 || tiles > 256) return p;
    // padding waste of the 256-wide tiles
    const double fill = (double)M * (double)N / ((double)p.tiles_m * 256.0 * (double)p.tiles_n * 256.0);
    if (fill < 0.8) return p;
    // Split-K is NOT planned: measured on the WaveEq decoder shapes the slab round trip of 256-wide tiles (4 B written + read per
    // output element and split) costs more than it buys -- 3328x1200x4096: 90 us split in 3 vs 78 us on the 128x64 tile,
    // 3328x1200x1200: 53 vs 35 us -- so the tile is taken where its tiles alone fill most of the chip (>= 160 of 256 CUs).
    int splits = 1;
    if (tiles < 160) return p;
    p.k_tiles_per_split = vs_cdiv(kt, splits);
    p.splits = (int)vs_cdiv(kt, p.k_tiles_per_split);
    p.use = true;
    return p;
}

}  // namespace
