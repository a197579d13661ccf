// vs_gemm_p8.h -- 256 x (128 NI) output tile, 8 waves in two staggered groups, K in steps of 64 through two LDS-DMA buffers with counted
// waits: the 16-bit GEMM main loop of the "8 phases per two K tiles" class (cdna_hip_programming.md section 5, T2-T5) for the
// Linear layers of the MLP family (reference: networks/mlp.py:24-41, mlp_encdec.py:43-50).
//
// Geometry (NI = 2: 256 x 256; NI = 1: 256 x 128):
//   * 512 threads = 8 waves as 2 (M) x 4 (N); wave (wr, wc) owns the contiguous 128 x (32 NI) block of C at rows wr*128, columns wc*32*NI,
//     as 2 x 2 quadrants Q(i, j) of 64 x (16 NI): 4 x NI accumulators of v_mfma_f32_16x16x32 each (NI = 2: 128 registers).
//   * one K tile (64 deep) in LDS = four half-tile images  HA0 | HA1 | HB0 | HB1.  HAi holds, for every wave row wr, the 64 rows of
//     quadrant row i (LDS row wr*64 + x  <->  tile row wr*128 + i*64 + x); HBj the 16 NI columns of quadrant column j of every wave column
//     (LDS row wc*16NI + y  <->  tile column wc*32NI + j*16NI + y).  So a phase that multiplies Q(i, j) reads exactly HAi and HBj, every
//     image is read in ONE phase by all waves, and can be re-requested two phases later.  Two K-tile buffers: 2 x 64 KiB (NI = 2).
//   * images are filled by global_load_lds_dwordx4 (1 KiB per wave instruction, no VGPRs), so they are lane-linear; the bank swizzle sits on
//     the SOURCE address and on the read address (the same involution):
//       R operand (reduction index contiguous): [rows / 8][8 rows][8 pieces of 8 k], piece s of row r in slot s ^ ((r >> 1) & 7): a wave
//         instruction fetches 8 whole 128-byte lines, and the 16 lanes of every ds_read_b128 group land on the 16 different 16-byte
//         slots of the 256-byte bank row (conflict-free; the plain [128][64] image of the guide's template is 4-way);
//       S operand (output index contiguous): [64 k][rows] with 32-byte pieces (16 rows) of k-row k in slot p ^ ((k & 3) | ((k >> 3) & 1) << 2),
//         read with ds_read_b64_tr_b16 (the 8 k-rows of a 32-lane half sit in 8 different 32-byte slots).
//   * schedule of one K tile u (buffer u & 1), four phases, each [LDS reads + one half-tile request | barrier | 8 NI MFMAs | barrier]:
//         phase 1: read HB0, HA0;  request HA1 of tile u+1;  multiply Q(0,0)
//         phase 2: read HB1;       request HB0 of tile u+2;  multiply Q(0,1)
//         phase 3: read HA1;       request HA0 of tile u+2;  multiply Q(1,1)
//         phase 4: --              request HB1 of tile u+2;  s_waitcnt vmcnt(three requests);  multiply Q(1,0)
//     waves 4-7 run one barrier behind waves 0-3 (one extra s_barrier up front), so on every SIMD one wave multiplies while its partner
//     reads and requests.  The wait of phase 4 leaves the three youngest half-tile requests in flight across the barriers and retires
//     tile u+1, which is read from the next phase on (one barrier after every wave's wait).  An image is re-requested two phases after
//     the phase that read it -- HB0 one phase after, its reads being retired by a counted lgkmcnt BEFORE the reading phase's barrier.
//   * operands are multiplied swapped (the B fragment is the MFMA's first operand): a lane then holds four CONSECUTIVE columns of one row
//     of C, and the epilogue stores 16-byte vectors straight from the accumulators (no LDS pass).
#pragma once
#include <type_traits>
#include "vs_gemm_big.h"

namespace {

constexpr int P8_BK = 64;
#ifndef P8_LOSS_DEPTH
#define P8_LOSS_DEPTH 12       // target requests in flight per lane in the frame-loss epilogue (16: the sigmoid form spills 376 bytes per lane)
#endif

// tools/probes/p8_bench.hip -DP8_STAMPS: wall-clock stamps (100 MHz) of workgroup 0 at the seams of the kernel; nothing in the library build
#ifdef P8_STAMPS
__device__ long long p8_stamps[8];
#define P8_STAMP(k) long long p8_t##k = wall_clock64()
#define P8_STAMPS_OUT() do { if (blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x == 0) { p8_stamps[0] = p8_t0; p8_stamps[1] = p8_t1; p8_stamps[2] = p8_t2; \
                                  p8_stamps[3] = p8_t3; p8_stamps[4] = p8_t4; p8_stamps[5] = wall_clock64(); } } while (0)
#else
#define P8_STAMP(k) do { } while (0)
#define P8_STAMPS_OUT() do { } while (0)
#endif

template <int CT>
__device__ __forceinline__ f32x4 mfma16x32(const u32x4& a, const u32x4& b, const f32x4& c) {
    if constexpr (CT == VS_BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ const char* p8_uniform64(const char* p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const char*>((uintptr_t)(((uint64_t)hi << 32) | lo));
}

// S image: the 32-byte slot (16 output indices) p of k-row kr lives in slot p ^ key(kr); the 8 k-rows a 32-lane half of ds_read_b64_tr_b16
// touches (kr = 8 g + q, g of two values, q = 0..3; the second read of a fragment: + 4) land on 8 different 32-byte pieces of the bank row
template <int ROWS>
__device__ __forceinline__ constexpr int p8_s_slot_key(int kr) {
    return ROWS == 128 ? ((kr & 3) | (((kr >> 3) & 1) << 2))          // 256-byte k-rows: one k-row = one bank row, 8 slots
                       : (((kr >> 1) & 1) | (((kr >> 3) & 1) << 1));  // 128-byte k-rows: two k-rows per bank row (kr & 1), 4 slots each
}

// LDS row -> tile row / column of half-tile image h; RUN = consecutive indices a wave owns per quadrant (A: 64, B: 16 NI).
// A: 2 wave rows x 64 = 128 LDS rows; B: 4 wave columns x 16 NI = 64 NI LDS rows
__device__ __forceinline__ constexpr int p8_tile_index(int lds_row, int h, int run) { return (lds_row / run) * (2 * run) + h * run + (lds_row % run); }

// One operand's loader.  ROWS = LDS rows of a half-tile image (128 or 64), RUN as above.  Requests per thread and half-tile: ROWS / 64.
template <int LAYOUT, int ROWS, int RUN>
struct P8Operand {
    static constexpr int NJ = ROWS / 64;                 // 1 KiB pieces per wave and half-tile
    static constexpr int HALF_BYTES = ROWS * P8_BK * 2;
    const char* base[2];                                 // UNIFORM [h]: first element of the next K tile of half h (each half advances alone)
    uint32_t voff[2][NJ];                                // [h][j]: this lane's byte offset from base[h]
    int kofs[NJ];                                        // k of this lane's piece inside a K tile
    int64_t step;

    __device__ __forceinline__ void prepare(const unsigned short* p, int64_t ld, int64_t rows, int64_t i0, int64_t k_begin, int wave) {
        const int i = (int)(threadIdx.x & 63);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (LAYOUT == LR) {
                    const int r = 8 * wave + (i >> 3) + 64 * j;                      // LDS row of this lane's piece
                    const int s = (i & 7) ^ ((r >> 1) & 7);                          // the 8-k piece that lives in slot i & 7 of that row
                    int64_t g = i0 + p8_tile_index(r, h, RUN);
                    if (g > rows - 1) g = rows - 1;                                  // clamped rows land in output rows never stored
                    kofs[j] = 8 * s;
                    voff[h][j] = (uint32_t)(((g - i0) * ld + 8 * s) * 2);
                } else {
                    // k-row of this lane's piece and the LDS column (output index) of the 32-byte slot it fills: 256-byte k-rows (ROWS 128, 4 per
                    // wave instruction) or 128-byte k-rows (ROWS 64, 8 per wave instruction)
                    const int kr = ROWS == 128 ? 4 * wave + 32 * j + (i >> 4) : 8 * wave + (i >> 3);
                    const int f = p8_s_slot_key<ROWS>(kr);
                    const int nl = ROWS == 128 ? ((((i & 15) >> 1) ^ f) << 4) + (i & 1) * 8 : ((((i & 7) >> 1) ^ f) << 4) + (i & 1) * 8;
                    int64_t g = i0 + p8_tile_index(nl, h, RUN);
                    if (g > rows - 8) g = rows - 8;                                  // rows % 8 == 0: a piece is inside or outside as a whole
                    kofs[j] = kr;
                    voff[h][j] = (uint32_t)((kr * ld + (g - i0)) * 2);
                }
            }
        // base and step are wave-uniform by construction; the 64-bit multiplies above are VALU work, so say it: an "s" operand of the request
        // below must be an SGPR pair (hipcc puts a VGPR pair into the instruction text otherwise: "invalid operand")
        const char* b0 = LAYOUT == LR ? reinterpret_cast<const char*>(p + i0 * ld + k_begin) : reinterpret_cast<const char*>(p + k_begin * ld + i0);
        base[0] = base[1] = p8_uniform64(b0);
        step = (int64_t)p8_uniform64(reinterpret_cast<const char*>((uintptr_t)(LAYOUT == LR ? (int64_t)P8_BK * 2 : (int64_t)P8_BK * ld * 2)));
    }
    // request half `h` of K tile `kt` into the image at byte OFF of the LDS area (`wbase` = LDS address of the area + wave * 1024, uniform);
    // CHECKED: the tile may be partial or past the end of the split (zeros then: k past K meets valid rows of the other operand).
    // M0 = LDS address of this wave's piece, written in the same statement that uses it.
    template <bool CHECKED, int OFF>
    __device__ __forceinline__ void stage(int h, uint32_t wbase, int64_t kt, int64_t kt_end, int64_t K) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if constexpr (!CHECKED) {
                if (j == 0)
                    asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(wbase), "v"(voff[h][0]), "s"(base[h]), "n"(OFF) : "memory", "m0", "scc");
                else
                    asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(wbase), "v"(voff[h][NJ - 1]), "s"(base[h]), "n"(OFF + 8192) : "memory", "m0", "scc");
            } else {
                const void* g = (kt < kt_end && kt * P8_BK + kofs[j] < K) ? (const void*)(base[h] + voff[h][j]) : (const void*)vs_glds_zero;
                if (j == 0)
                    asm volatile("s_add_u32 m0, %0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(wbase), "v"(g), "n"(OFF) : "memory", "m0", "scc");
                else
                    asm volatile("s_add_u32 m0, %0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(wbase), "v"(g), "n"(OFF + 8192) : "memory", "m0", "scc");
            }
        }
        base[h] += step;
    }
};

// ---- fragment reads ---------------------------------------------------------------------------------------------------------------
// A fragment = 8 elements of k-step ks (0 / 1) for output index R0 + (lane & 15), R0 a multiple of 16.  The lane-dependent part of its
// LDS address is kept in a few base registers per K-tile buffer (NB per operand), everything else is an immediate offset below 64 KiB
// (buffer 1 starts 64 KiB in: folded into the address by the compiler, every read of it would want a register of its own).
//   R image: base[ks]   = lane part with the k-step's slot permutation;      + image + R0 * 128
//   S image: base[rep]  = lane part with fragment repeat rep's slot (4 wr + mi, or NI wc + ni) permuted in;   + image + ks * 32 k-rows (+ 4 k-rows)
typedef __attribute__((address_space(3))) const u32x4 p8_lds_u32x4;
typedef __attribute__((address_space(3))) vs_i16x4 p8_lds_i16x4;

template <int LAYOUT, int REPS>
struct P8Reader {
    static constexpr int NB = LAYOUT == LR ? 2 : REPS;
    uint32_t base[2][NB];                                             // [buffer][...]: LDS byte addresses

    // `area`: LDS address of the operand's first image in buffer 0; `tile_bytes`: distance of buffer 1; `first_rep`: 4 wr (A) / NI wc (B) = R0 / 16 of repeat 0
    __device__ __forceinline__ void prepare(uint32_t area, int tile_bytes, int first_rep, int rows, int lane) {
#pragma unroll
        for (int P = 0; P < 2; ++P)
#pragma unroll
            for (int x = 0; x < NB; ++x) {
                uint32_t a;
                if (LAYOUT == LR) {
                    const int r = lane & 15;
                    a = area + P * tile_bytes + first_rep * 2048 + (r >> 3) * 1024 + (r & 7) * 128 + ((((lane >> 4) + 4 * x) ^ (r >> 1)) << 4);
                } else {
                    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
                    const int f = rows == 128 ? p8_s_slot_key<128>(8 * g + q) : p8_s_slot_key<64>(8 * g + q);
                    a = area + P * tile_bytes + (8 * g + q) * (rows * 2) + (((first_rep + x) ^ f) << 5) + 8 * pp;
                }
                asm volatile("" : "+v"(a));                            // opaque: keeps 64 KiB out of the immediate offsets
                base[P][x] = a;
            }
    }
    // repeat `rep` (0 .. REPS-1), k-step ks, of the image `img_off` bytes into the operand's area; P, rep, ks, img_off compile-time after unrolling
    __device__ __forceinline__ u32x4 frag(int P, int img_off, int rep, int ks, int rows) const {
        if (LAYOUT == LR) {
            return *reinterpret_cast<p8_lds_u32x4*>(base[P][ks] + img_off + rep * 2048);
        } else {
            const uint32_t a = base[P][rep] + img_off + ks * 32 * (rows * 2);
            const vs_i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<p8_lds_i16x4*>(a));
            const vs_i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<p8_lds_i16x4*>(a + 4 * (rows * 2)));
            typedef __attribute__((ext_vector_type(8))) short i16x8;
            return __builtin_bit_cast(u32x4, (i16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
    }
};

#define P8_FENCE __builtin_amdgcn_sched_barrier(0)
#define P8_BAR()                                \
    do {                                        \
        asm volatile("s_barrier" ::: "memory"); \
        P8_FENCE;                               \
    } while (0)

// MI: 16-row accumulator repeats per quadrant (4: 256 rows per tile; 2: 128 rows -- 64 KiB of LDS and <= 128 registers, two workgroups per CU)
// LOSS: 0 = a GEMM; 1 / 2 = the frame-loss epilogue (vs_gemm_frame_loss) whose straight-line form on interior tiles is compiled for no activation /
// for a sigmoid behind the last Linear (one form per instantiation: two in one kernel cost it its spill-free register allocation)
template <int CT, int LA, int LB, int NI, bool NCHW, int LOSS = 0, int MI = 4>
__global__ __launch_bounds__(512, MI == 4 ? 2 : 4) void gemm_p8_kernel(const unsigned short* Ap, int64_t lda, const unsigned short* Bp, int64_t ldb, int64_t M, int64_t N,
                                                      int64_t K, int k_tiles_per_split, int tiles_n, Epi epi_in, float* slabs) {
    constexpr int BN = 128 * NI, BM = 64 * MI;
    constexpr int RUN_A = 16 * MI, ROWS_A = 32 * MI;
    constexpr int RUN_B = 16 * NI, ROWS_B = 64 * NI;
    typedef P8Operand<LA, ROWS_A, RUN_A> OpA;
    typedef P8Operand<LB, ROWS_B, RUN_B> OpB;
    constexpr int HA = OpA::HALF_BYTES, HB = OpB::HALF_BYTES;
    constexpr int TILE = 2 * HA + 2 * HB;                             // one K tile: HA0 | HA1 | HB0 | HB1
    constexpr int WAIT = OpB::NJ * 2 + OpA::NJ;                       // requests of HB0, HA0, HB1: what the phase-4 wait leaves in flight

    P8_STAMP(0);
    int zsplit = blockIdx.z;
    int batch = 0;
    if (epi_in.splits_per_batch > 0) {
        batch = blockIdx.z / epi_in.splits_per_batch;
        zsplit = blockIdx.z - batch * epi_in.splits_per_batch;
        Ap += batch * epi_in.batch_a;
        Bp += batch * epi_in.batch_b;
    }
    const Epi epi = epi_for_batch(epi_in, batch);
    extern __shared__ __attribute__((aligned(16))) char smem[];      // the ONLY LDS object: 2 x TILE

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wr = wave >> 2, wc = wave & 3;
    const unsigned tile = big_tile_of(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(tile / (unsigned)tiles_n) * BM, n0 = (int64_t)(tile % (unsigned)tiles_n) * BN;
    const int64_t kt_total = (K + P8_BK - 1) / P8_BK;
    const int64_t kt_begin = (int64_t)zsplit * k_tiles_per_split;
    int64_t kt_end = kt_begin + k_tiles_per_split;
    if (kt_end > kt_total) kt_end = kt_total;
    int64_t kt_full = K / P8_BK;                                     // tiles [kt_begin, kt_full) are full and live
    if (kt_full > kt_end) kt_full = kt_end;

    f32x4 acc[2][2][MI][NI];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[i][j][mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    const uint32_t wbase = (uint32_t)(uintptr_t)smem + (uint32_t)wave * 1024u;      // LDS address of this wave's 1 KiB piece of image 0
    OpA ga;
    OpB gb;
    ga.prepare(Ap, lda, M, m0, kt_begin * P8_BK, wave);
    gb.prepare(Bp, ldb, N, n0, kt_begin * P8_BK, wave);

    u32x4 fa[MI][2], fb0[NI][2], fb1[NI][2];                           // [repeat][k-step]
    P8Reader<LA, MI> ra;
    P8Reader<LB, NI> rb;
    ra.prepare((uint32_t)(uintptr_t)smem, TILE, MI * wr, ROWS_A, lane);
    rb.prepare((uint32_t)(uintptr_t)smem + 2 * HA, TILE, NI * wc, ROWS_B, lane);
    auto rd_a = [&](int P, int i) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa[mi][ks] = ra.frag(P, i * HA, mi, ks, ROWS_A);
    };
    auto rd_b = [&](u32x4(&fb)[NI][2], int P, int j) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fb[ni][ks] = rb.frag(P, j * HB, ni, ks, ROWS_B);
    };
    auto mm = [&](auto ic, auto jc, const u32x4(&fb)[NI][2]) {
        constexpr int i = decltype(ic)::value, j = decltype(jc)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[i][j][mi][ni] = mfma16x32<CT>(fb[ni][ks], fa[mi][ks], acc[i][j][mi][ni]);
        __builtin_amdgcn_s_setprio(0);
        P8_FENCE;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    constexpr int LGKM_B0 = (LA == LR ? 2 * MI : (4 * MI > 15 ? 15 : 4 * MI));                      // reads of HA0 that may still be out when HB0's are done (4-bit counter)

    // one K tile `u` out of buffer P (see the schedule in the header)
    auto k_tile = [&](auto pc, auto chk, int64_t u) {
        constexpr int P = decltype(pc)::value;
        constexpr bool CHK = decltype(chk)::value;
        constexpr int CUR = P * TILE, NXT = (P ^ 1) * TILE;
        // phase 1
        rd_b(fb0, P, 0);
        P8_FENCE;
        rd_a(P, 0);
        P8_FENCE;
        ga.template stage<CHK, NXT + HA>(1, wbase, u + 1, kt_end, K);
        // HB0 is re-requested in phase 2: its reads (issued first) are retired here, ahead of this phase's barrier
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(LGKM_B0) : "memory");
        P8_BAR();
        mm(I0{}, I0{}, fb0);
        P8_BAR();
        // phase 2
        rd_b(fb1, P, 1);
        P8_FENCE;
        gb.template stage<CHK, CUR + 2 * HA>(0, wbase, u + 2, kt_end, K);
        P8_BAR();
        mm(I0{}, I1{}, fb1);
        P8_BAR();
        // phase 3
        rd_a(P, 1);
        P8_FENCE;
        ga.template stage<CHK, CUR>(0, wbase, u + 2, kt_end, K);
        P8_BAR();
        mm(I1{}, I1{}, fb1);
        P8_BAR();
        // phase 4
        gb.template stage<CHK, CUR + 2 * HA + HB>(1, wbase, u + 2, kt_end, K);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT) : "memory");
        P8_BAR();
        mm(I1{}, I0{}, fb0);
        P8_BAR();
    };
    using T = std::true_type;
    using F = std::false_type;

    // prologue: tile 0 whole, HB0 | HA0 | HB1 of tile 1 (HA1 of tile 1 is phase 1's request)
    P8_STAMP(1);
    {
        ga.template stage<true, 0>(0, wbase, kt_begin, kt_end, K);
        ga.template stage<true, HA>(1, wbase, kt_begin, kt_end, K);
        gb.template stage<true, 2 * HA>(0, wbase, kt_begin, kt_end, K);
        gb.template stage<true, 2 * HA + HB>(1, wbase, kt_begin, kt_end, K);
        gb.template stage<true, TILE + 2 * HA>(0, wbase, kt_begin + 1, kt_end, K);
        ga.template stage<true, TILE>(0, wbase, kt_begin + 1, kt_end, K);
        gb.template stage<true, TILE + 2 * HA + HB>(1, wbase, kt_begin + 1, kt_end, K);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT) : "memory");
        P8_BAR();
        if (wr == 1) P8_BAR();                                        // waves 4-7 run one barrier behind
    }
    P8_STAMP(2);
    int64_t u = kt_begin;
    // main loop: pairs of K tiles whose requests (tiles u+1 .. u+3) are full and live
    while (u + 3 < kt_full) {
        k_tile(I0{}, F{}, u);
        k_tile(I1{}, F{}, u + 1);
        u += 2;
    }
    while (u < kt_end) {
        k_tile(I0{}, T{}, u);
        if (++u >= kt_end) break;
        k_tile(I1{}, T{}, u);
        ++u;
    }
    if (wr == 0) P8_BAR();                                            // (the barrier waves 4-7 took up front)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // zero-source requests past the end must not outlive the workgroup's LDS
    P8_STAMP(3);

    // ---- epilogue: a lane holds C[m][n .. n+3] per accumulator ----
    float* slab_base = slabs ? slabs + (int64_t)blockIdx.z * M * N : nullptr;
    const int64_t mw = m0 + wr * (2 * RUN_A) + (lane & 15), nw = n0 + wc * (2 * RUN_B) + (lane >> 4) * 4;
    if constexpr (LOSS) {
        // frame losses (vs_gemm_frame_loss): compare with the target frames instead of storing; partial sums per workgroup
        float fl_s0 = 0.f, fl_s1 = 0.f;
        const float fl_up = epi.fl_up[0];
        const int fl_t = epi.fl_tdev[0];
        // Sixteen target requests (one quadrant row: 2 x NI x 4 accumulators' worth) are in flight before the first is consumed: a lane's 16 bytes
        // of a target frame come from HBM at ~2 us under load, and four at a time (the first form) left the epilogue waiting 8 times per wave
        // (fused losses then LOST 40 us on the WaveEq step).  Rows / columns past the end read a clamped (valid) address and are skipped.
        f32x4 bias4[2][NI];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int64_t n = nw + j * RUN_B + ni * 16;
                const int64_t n_c = n + 3 < N ? n : N - 4;
                bias4[j][ni] = epi.bias ? *reinterpret_cast<const f32x4*>(epi.bias + n_c) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        // A tile that lies inside the matrix with the common settings (no activation or a sigmoid behind the last Linear, gradient stored in the compute
        // type) takes straight-line code: no branch per element (the run-time activation / dtype switches are ~100 scalar branches per four
        // elements, ~36 us of a 64 us launch at 3328 x 4096 x 1200 by the stamps of tools/probes/p8_bench.hip), one rolling window of DEPTH
        // target requests that is refilled as it is consumed (the hardware counter waits are the compiler's, exact in straight-line code), and
        // the four stores of a row back to back (a full 128-byte line of the gradient).  Same arithmetic, same order of the partial sums.
        const bool quick = !epi.p8_plain && (m0 + BM <= M) && (n0 + BN <= N) && epi.act == (LOSS == 2 ? VS_ACT_SIGMOID : VS_ACT_NONE) && epi.fl_dz_dtype == CT;
        auto loss_quick = [&]() {
            constexpr int ACT = LOSS == 2 ? VS_ACT_SIGMOID : VS_ACT_NONE;
            constexpr int RQ = 2 * NI, ROWS = 2 * MI, Q = ROWS * RQ, DEPTH = P8_LOSS_DEPTH < Q ? P8_LOSS_DEPTH : Q;
            const float k_ae = fl_up * epi.fl_l_ae * 2.f * epi.fl_inv_ae, k_pred = fl_up * epi.fl_l_pred * 2.f * epi.fl_inv_pred;
            const float* tp[ROWS];
            unsigned short* dp[ROWS];
            int g_of[ROWS];
            f32x4 tg[DEPTH];
            auto request = [&](int q) {
                const int r = q / RQ, c = q % RQ;
                if (c == 0) {
                    const int64_t m = mw + (r / MI) * RUN_A + (r % MI) * 16;
                    tp[r] = big_loss_target(epi, m, nw, N, fl_t, g_of[r]);
                    dp[r] = (unsigned short*)epi.fl_dz + m * N + nw;
                }
                tg[q % DEPTH] = *reinterpret_cast<const f32x4*>(tp[r] + (c / NI) * RUN_B + (c % NI) * 16);
            };
#pragma unroll
            for (int q = 0; q < DEPTH; ++q) request(q);
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int r = q / RQ, c = q % RQ, i = r / MI, mi = r % MI, j = c / NI, ni = c % NI;
                const float k = g_of[r] == 0 ? k_ae : k_pred;
                const f32x4 t4 = tg[q % DEPTH];
                float rr[4], sq = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float x = vs_act(acc[i][j][mi][ni][t] * epi.alpha + bias4[j][ni][t], ACT);
                    const float d = x - t4[t];
                    sq += d * d;
                    rr[t] = (k * d) * vs_act_grad_from_out(x, ACT);
                }
                fl_s0 += g_of[r] == 0 ? sq : 0.f;
                fl_s1 += g_of[r] == 0 ? 0.f : sq;
                const u16x4 w = {vs_f2h(rr[0], CT), vs_f2h(rr[1], CT), vs_f2h(rr[2], CT), vs_f2h(rr[3], CT)};
                *reinterpret_cast<u16x4*>(dp[r] + j * RUN_B + ni * 16) = w;
                if (q + DEPTH < Q) request(q + DEPTH);
                P8_FENCE;                                  // (left to itself the scheduler hoists requests and row set-ups until registers spill)
            }
        };
        if (quick) loss_quick();
        auto loss_rows = [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            f32x4 tg[2][NI][MI];
            int gg[MI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                int64_t m = mw + i * RUN_A + mi * 16;
                if (m > M - 1) m = M - 1;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const int64_t n = nw + j * RUN_B + ni * 16;
                        const int64_t n_c = n + 3 < N ? n : N - 4;
                        tg[j][ni][mi] = *reinterpret_cast<const f32x4*>(big_loss_target(epi, m, n_c, N, fl_t, gg[mi]));
                    }
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int64_t m = mw + i * RUN_A + mi * 16;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const int64_t n = nw + j * RUN_B + ni * 16;
                        if (m < M && n + 3 < N) big_loss4(epi, m, n, N, gg[mi], acc[i][j][mi][ni], tg[j][ni][mi], bias4[j][ni], fl_up, fl_s0, fl_s1);
                    }
            }
        };
        if (!quick) {
            loss_rows(I0{});
            loss_rows(I1{});
        }
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
    } else {
        // The stores that matter (Linear forward: bias + none / relu / leaky; input gradient: activation mask of the layer below) are
        // straight-line code with the activation as a compile-time constant, accumulators in registers.  Everything else (sigmoid / tanh / elu,
        // accumulate, NCHW scatter, odd alignments) takes the general store in a run-time loop: compact code, accumulators through scratch
        // memory once (32 unrolled copies of the general store were 30 000 instructions per instantiation).
        const bool mask_ok = !epi.mask || ((epi.mask_act == VS_ACT_RELU || epi.mask_act == VS_ACT_LEAKY) && (epi.ldmask & 3) == 0 && ((uintptr_t)epi.mask & 15) == 0);
        const bool fast = !NCHW && mask_ok && !epi.accumulate && !epi.adam_m && (epi.ldc & 3) == 0 && (N & 3) == 0 && ((uintptr_t)epi.C & 15) == 0 &&
                          (!epi.bias || ((uintptr_t)epi.bias & 15) == 0) && (epi.act == VS_ACT_NONE || epi.act == VS_ACT_RELU || epi.act == VS_ACT_LEAKY);
        // Interior tiles of the stores of a training step -- Linear forward (bias, none / relu, 16-bit result) and input gradient (16-bit relu /
        // leaky mask of the layer below) -- take straight-line code: the general fast form below waits for EVERY store to be acknowledged before
        // the next one (its bounds branches leave the compiler a `s_waitcnt vmcnt(0)` in front of each: 32 x ~250 ns = 8 us of a 36 us launch
        // at 3328 x 4096 x 1200, stamps of tools/probes/p8_bench.hip).  Here the mask bits come through a rolling window of DEPTH requests, the
        // test `mask > 0` is made on the 16 bits themselves (positive, non-zero, not a NaN: what the float comparison says) and the stores of a
        // row follow each other (full 128-byte lines).
        const bool quick = !epi.p8_plain && fast && !slab_base && (m0 + BM <= M) && (n0 + BN <= N) && epi.c_dtype == CT && (epi.act == VS_ACT_NONE || epi.act == VS_ACT_RELU) &&
                           (!epi.mask || epi.mask_dtype == VS_BF16 || epi.mask_dtype == VS_F16);
        if (quick) {
            auto store_quick = [&](auto actc) {
                constexpr int ACT = decltype(actc)::value;
                constexpr int RQ = 2 * NI, ROWS = 2 * MI, Q = ROWS * RQ, DEPTH = 16 < Q ? 16 : Q;
                const bool has_mask = epi.mask != nullptr;
                const float mask_slope = epi.mask_act == VS_ACT_LEAKY ? 0.2f : 0.f;
                const int top = epi.mask_dtype == VS_F16 ? 0x7C00 : 0x7F80;          // +inf: the largest bit pattern that is "> 0"
                f32x4 bias4[2][NI];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        bias4[j][ni] = epi.bias ? *reinterpret_cast<const f32x4*>(epi.bias + nw + j * RUN_B + ni * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
                const unsigned short* yp[ROWS];
                unsigned short* cp[ROWS];
                u16x4 yh[DEPTH];
                auto request = [&](int q) {
                    const int r = q / RQ, c = q % RQ;
                    if (c == 0) {
                        const int64_t m = mw + (r / MI) * RUN_A + (r % MI) * 16;
                        yp[r] = (const unsigned short*)epi.mask + m * epi.ldmask + nw;
                        cp[r] = (unsigned short*)epi.C + m * epi.ldc + nw;
                    }
                    if (has_mask) yh[q % DEPTH] = *reinterpret_cast<const u16x4*>(yp[r] + (c / NI) * RUN_B + (c % NI) * 16);
                    else yh[q % DEPTH] = u16x4{1, 1, 1, 1};
                };
#pragma unroll
                for (int q = 0; q < DEPTH; ++q) request(q);
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const int r = q / RQ, c = q % RQ, i = r / MI, mi = r % MI, j = c / NI, ni = c % NI;
                    const u16x4 y4 = yh[q % DEPTH];
                    float rr[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float x = vs_act(acc[i][j][mi][ni][t] * epi.alpha + bias4[j][ni][t], ACT);
                        const int yb = (int)(short)y4[t];
                        rr[t] = x * ((yb > 0 && yb <= top) ? 1.f : mask_slope);
                    }
                    const u16x4 w = {vs_f2h(rr[0], CT), vs_f2h(rr[1], CT), vs_f2h(rr[2], CT), vs_f2h(rr[3], CT)};
                    *reinterpret_cast<u16x4*>(cp[r] + j * RUN_B + ni * 16) = w;
                    if (q + DEPTH < Q) request(q + DEPTH);
                }
            };
            if (epi.act == VS_ACT_RELU) store_quick(std::integral_constant<int, VS_ACT_RELU>{});
            else store_quick(std::integral_constant<int, VS_ACT_NONE>{});
        } else if (slab_base && (N & 3) == 0) {
            // split-K partial: raw fp32, reduced (with the epilogue) by splitk_reduce_kernel
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            const int64_t m = mw + i * RUN_A + mi * 16, n = nw + j * RUN_B + ni * 16;
                            if (m < M && n < N) *reinterpret_cast<f32x4*>(slab_base + m * N + n) = acc[i][j][mi][ni];
                        }
        } else if (fast && !slab_base) {
            const float mask_slope = epi.mask_act == VS_ACT_LEAKY ? 0.2f : 0.f;
            auto store_all = [&](auto actc, auto wide) {
                constexpr int ACT = decltype(actc)::value;
                constexpr bool F32OUT = decltype(wide)::value;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) {
                        const int64_t n = nw + j * RUN_B + ni * 16;
                        const int64_t n_c = n < N ? n : 0;
                        const f32x4 bias4 = epi.bias ? *reinterpret_cast<const f32x4*>(epi.bias + n_c) : f32x4{0.f, 0.f, 0.f, 0.f};
                        // the mask values of the column's eight accumulators are requested TOGETHER, from clamped (always valid) addresses and ahead of
                        // the bounds branches: a load behind each branch would be waited for one at a time (8 x ~1 us per column group)
                        // (raw bits until they are used: a conversion next to the load would wait for it)
                        f32x4 yf[2][MI];
                        u16x4 yh[2][MI];
                        if (epi.mask) {
#pragma unroll
                            for (int i = 0; i < 2; ++i)
#pragma unroll
                                for (int mi = 0; mi < MI; ++mi) {
                                    int64_t m = mw + i * RUN_A + mi * 16;
                                    if (m > M - 1) m = M - 1;
                                    if (epi.mask_dtype == VS_F32) yf[i][mi] = *reinterpret_cast<const f32x4*>((const float*)epi.mask + m * epi.ldmask + n_c);
                                    else yh[i][mi] = *reinterpret_cast<const u16x4*>((const unsigned short*)epi.mask + m * epi.ldmask + n_c);
                                }
                        }
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int mi = 0; mi < MI; ++mi) {
                                const int64_t m = mw + i * RUN_A + mi * 16;
                                if (!(m < M && n < N)) continue;
                                f32x4 r;
#pragma unroll
                                for (int t = 0; t < 4; ++t) r[t] = vs_act(acc[i][j][mi][ni][t] * epi.alpha + bias4[t], ACT);
                                if (epi.mask) {
#pragma unroll
                                    for (int t = 0; t < 4; ++t) {
                                        const float yv = epi.mask_dtype == VS_F32 ? yf[i][mi][t] : vs_h2f(yh[i][mi][t], epi.mask_dtype);
                                        r[t] *= yv > 0.f ? 1.f : mask_slope;
                                    }
                                }
                                if constexpr (F32OUT) {
                                    *reinterpret_cast<f32x4*>((float*)epi.C + m * epi.ldc + n) = r;
                                } else {
                                    const u16x4 hv = {vs_f2h(r[0], epi.c_dtype), vs_f2h(r[1], epi.c_dtype), vs_f2h(r[2], epi.c_dtype), vs_f2h(r[3], epi.c_dtype)};
                                    *reinterpret_cast<u16x4*>((unsigned short*)epi.C + m * epi.ldc + n) = hv;
                                }
                            }
                    }
            };
            auto by_type = [&](auto actc) {
                if (epi.c_dtype == VS_F32) store_all(actc, std::true_type{});
                else store_all(actc, std::false_type{});
            };
            if (epi.act == VS_ACT_RELU) by_type(std::integral_constant<int, VS_ACT_RELU>{});
            else if (epi.act == VS_ACT_LEAKY) by_type(std::integral_constant<int, VS_ACT_LEAKY>{});
            else by_type(std::integral_constant<int, VS_ACT_NONE>{});
        } else {
            // accumulators -> LDS (idle now; compile-time indices) -> a run-time loop over them: a lane reads back what it wrote itself
            P8_BAR();                                                  // every wave's requests have landed (vmcnt(0) above): the buffers are free
            f32x4* const stg = reinterpret_cast<f32x4*>(smem) + wave * (2 * MI * NI * 64) + lane;
            auto pass = [&](auto ic) {
                constexpr int i = decltype(ic)::value;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) stg[((j * MI + mi) * NI + ni) * 64] = acc[i][j][mi][ni];
#pragma unroll 1
                for (int q = 0; q < 2 * MI * NI; ++q) {
                    const int ni = q % NI, mi = (q / NI) % MI, j = q / (MI * NI);
                    const int64_t m = mw + i * RUN_A + mi * 16, n = nw + j * RUN_B + ni * 16;
                    const f32x4 v = stg[q * 64];
                    if (m < M && n < N) big_store4<NCHW>(epi, m, n, N, v, slab_base ? slab_base + m * N : nullptr);
                }
            };
            pass(I0{});
            pass(I1{});
        }
    }
    P8_STAMP(4);
#ifdef P8_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    P8_STAMPS_OUT();
}

// ---- when to take it ---------------------------------------------------------------------------------------------------------
// One workgroup per CU (128 / 96 KiB of LDS).  VS_GEMM_P8: 0 = never, 1 = by plan (default), 2 = whenever the operands allow (tests).
struct P8Plan { bool use; int ni; int splits; int64_t k_tiles_per_split; int tiles_m, tiles_n; int mi = 4; };

inline P8Plan make_p8_plan(int compute, int64_t M, int64_t N, int64_t K, int64_t batch, int lb) {
    P8Plan p{false, 2, 1, vs_cdiv(K, P8_BK), (int)vs_cdiv(M, 256), (int)vs_cdiv(N, 256)};
    const char* env = getenv("VS_GEMM_P8");                       // read per call: tests switch it
    const int mode = env ? atoi(env) : 1;
    if (compute == VS_F32 || mode == 0) return p;
    const char* env_ni = getenv("VS_GEMM_P8_NI");
    const int force_ni = env_ni ? atoi(env_ni) : 0;
    auto fill_of = [&](int bn) { return (double)M * (double)N / ((double)vs_cdiv(M, 256) * 256.0 * (double)vs_cdiv(N, bn) * (double)bn); };
    const char* env_mi = getenv("VS_GEMM_P8_MI");
    const int force_mi = env_mi ? atoi(env_mi) : 0;
    if (mode == 2) {
        p.ni = force_ni == 1 ? 1 : 2;
        if (force_mi == 2 && p.ni == 1) { p.mi = 2; p.tiles_m = (int)vs_cdiv(M, 128); }
        p.tiles_n = (int)vs_cdiv(N, 128 * p.ni);
        p.use = (int64_t)p.tiles_m * p.tiles_n * batch <= 65535;
        return p;
    }
    // One tile row, long K (the encoders' first layer, 256 x 1200 x 20480: the weight matrix is streamed once, HBM-bound): 256 x 128 tiles, K split
    // over ~one round of CUs into fp32 slabs (reduced by splitk_reduce_kernel).  Measured against the 128 x 128 ring tile (20 tiles x 22 splits):
    // alone, cold operands 43.5 -> 38.1 us; replayed WaveEq step, two interleaved pairs 1.2414 / 1.2337 -> 1.2114 / 1.2079 ms.  VS_GEMM_P8_SPLIT=0: off.
    {
        const char* env_sp = getenv("VS_GEMM_P8_SPLIT");
        // VS_GEMM_P8_SPLIT_NI=2: 256 x 256 tiles (the activation panel is read by half as many tile columns, twice the splits and slab bytes)
        const char* env_sni = getenv("VS_GEMM_P8_SPLIT_NI");
        const int sni = env_sni && atoi(env_sni) == 2 ? 2 : 1;
        const int64_t kt = p.k_tiles_per_split, tn = vs_cdiv(N, 128 * sni);
        if (!(env_sp && atoi(env_sp) == 0) && M > 128 && M <= 256 && N >= 512 && kt >= 64 && tn * batch <= 64 && (double)N / (tn * 128.0 * sni) >= 0.85) {
            // VS_GEMM_P8_SPLIT_MI=2: 128 x 128 tiles (100 registers per lane, two workgroups per CU: a workgroup fits beside a workgroup of the integrator's
            // forward kernel, under which E_s's first layer runs in the WaveEq step)
            const char* env_smi = getenv("VS_GEMM_P8_SPLIT_MI");
            const int tm = env_smi && atoi(env_smi) == 2 ? 2 : 1;
            int64_t splits = 250 / (tn * tm * batch);
            if (splits > kt / 8) splits = kt / 8;
            if (splits >= 2) {
                p.ni = sni;
                if (tm == 2 && sni == 1) { p.mi = 2; p.tiles_m = (int)vs_cdiv(M, 128); }
                p.tiles_n = (int)tn;
                p.k_tiles_per_split = vs_cdiv(kt, splits);
                p.splits = (int)vs_cdiv(kt, p.k_tiles_per_split);
                p.use = true;
                return p;
            }
        }
    }
    if (M < 512 || N < 512 || p.k_tiles_per_split < 6) return p;
    const int64_t t256 = (int64_t)p.tiles_m * p.tiles_n * batch;
    if (t256 >= 160 && fill_of(256) >= 0.8) { p.use = true; return p; }
    // 256 x 128 where the 256-wide tiles leave most CUs idle (decoder layers of the WaveEq model, 3328 x 1200: 65 -> 130 tiles; measured in the
    // replayed WaveEq step against the 64 x 64 tile, two interleaved pairs: 1.2309 / 1.2380 vs 1.2424 / 1.2459 ms).  VS_GEMM_P8_NI=2: never.
    // 128 x 128 (MI = 2, two workgroups per CU) where that fills the chip once: 3328 x 1200 -> 260 workgroups on 512 slots.  VS_GEMM_P8_MI=4: never.
    const int64_t t128sq = (int64_t)vs_cdiv(M, 128) * vs_cdiv(N, 128) * batch;
    const double fill128sq = (double)M * (double)N / ((double)vs_cdiv(M, 128) * 128.0 * (double)vs_cdiv(N, 128) * 128.0);
    if (force_ni != 2 && force_mi != 4 && force_mi == 2 && t128sq >= 200 && t128sq <= 512 && fill128sq >= 0.85 && p.k_tiles_per_split >= 8) {
        p.ni = 1;
        p.mi = 2;
        p.tiles_m = (int)vs_cdiv(M, 128);
        p.tiles_n = (int)vs_cdiv(N, 128);
        p.use = true;
        return p;
    }
    const int64_t t128 = (int64_t)p.tiles_m * vs_cdiv(N, 128) * batch;
    if (force_ni != 2 && t128 >= 96 && t128 <= 256 && fill_of(128) >= 0.85 && p.k_tiles_per_split >= 8) {
        p.ni = 1;
        p.tiles_n = (int)vs_cdiv(N, 128);
        p.use = true;
    }
    return p;
}

}  // namespace
