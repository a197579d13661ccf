// vs_conv_band2.hip -- round 5: Conv2d k3 s1 p1 (and the k4 s2 p1 family on parity planes) on row bands with BOTH operands staged through
// LDS by PERSISTENT workgroups (gfx950 only).  Reference layers: every 3x3 block of the VGG / SST encoders and decoders (conv.py:127-171,
// 267-426), the DCGAN k4 s2 p1 layers (conv.py:119-122, 260-263), forward and input gradient.
//
// What was measured on the round-2 kernel (conv3_band_kernel, vs_conv_img.hip) and on the first form of this one:
//   * counters (profiles/r05_pre_*_mfma_util.md): matrix pipes busy 21-44 % of the CU-busy cycles, texture-address units 40-62 %, LDS bank
//     conflicts 21-69 % of the LDS cycles for every map width but 16: four waves fetch the SAME weight fragments from global memory (3 KiB per
//     wave per six MFMAs = the whole vector-memory path) and the pixel image has a channel pitch of 640 / 768 bytes;
//   * ablation (every phase's DMA, fragment reads, MFMAs and barriers switched off one by one): the main loop of a 256 -> 256 layer on 8 x 8
//     maps is 35 of 81 us -- ~90 % of the MFMA rate while it runs -- and 46 us are per-WORKGROUP fixed cost: ~2 us of launch + address setup
//     and an epilogue whose stores reach 1.3-2.2 TB/s because they come in short bursts at the end of 2-10 us workgroups (64 -> 64 on
//     64 x 64 maps: main loop 34 us of 302).
// So this kernel
//   * stages the weights ONCE per workgroup by LDS-DMA (the pre-pack is in MFMA fragment order: a fragment is 1 KiB, lane-linear, read back
//     with one conflict-free ds_read_b128) and the pixels with their 16-byte pieces permuted per channel (piece ^ f(channel & 3): the
//     permutation sits on the DMA's per-lane SOURCE address and on the read address) -- conflict-free for every width;
//   * stages maps of 4 x 4 / 8 x 8 pixels DENSE (sixteen / four whole maps per tile, no zero rows); rows outside a map or the image are read
//     from a clamped address and the fragment is zeroed on the result lane (a lane of the transposed fragment IS one pixel);
//   * is PERSISTENT: the grid is what the chip holds (two workgroups per CU), a workgroup walks tiles L, L + grid, ... -- per-lane address
//     setup once, the DMA of the next tile's first phase rides in the current tile's last phase, the stores of a tile drain while the next
//     one is multiplied;
//   * stores through LDS: the accumulator holds one pixel per lane, so the tile is transposed in the stage it has just consumed
//     (wave-private) and leaves in 16-byte pieces, 1 KiB per wave instruction;
//   * phases of KC = 32 (16 for 64-wide maps) input channels, two LDS stages, one barrier per phase (+ one per tile), fragments double
//     buffered in registers, the x shift of the taps on the RESULT (three accumulators per column tile).
// WM = 1: 256 threads, a wave owns 32 output channels x 64 pixels, two workgroups per CU; WM = 2: 512 threads = 2 x 4 waves, 64 output
// channels x 256 pixels per workgroup (the pixel image is staged once for both channel tiles), one workgroup per CU.
#include "vs_gemm_glds.h"
#include <stdlib.h>

namespace {

template <int W, int KC>
struct B2Geo {
    static constexpr int IPB = W == 8 ? 4 : (W == 4 ? 16 : 1);          // maps per tile
    static constexpr int RI = W == 8 ? 8 : (W == 4 ? 4 : 256 / W);      // rows of a map (W <= 8) or of the band (W >= 16)
    static constexpr bool HALO = IPB == 1;                               // W >= 16: the band's neighbour rows are staged (R + 2 rows)
    static constexpr int RPI = RI + (HALO ? 2 : 0);
    static constexpr int CE = IPB * RPI * W;                             // elements of one channel's image
    static constexpr int PPC = CE / 8;                                   // its 16-byte pieces: 48 / 40 / 36 / 32 / 32
    static constexpr int XE = KC * CE;                                   // x elements per stage
    static constexpr int XPIECES = KC * PPC;
};

// permutation of a channel's 16-byte pieces (an involution on the piece index): the four channels a transposing read touches per 32-lane
// half land in four different 16-word bank groups
template <int W>
__device__ __forceinline__ int b2_swz(int c) {
    if constexpr (W == 32) return (c & 2) << 1;
    else if constexpr (W == 16) return 0;
    else return (c & 3) << 2;
}

__device__ __forceinline__ float b2_gather(int byte_index, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_index, __builtin_bit_cast(int, v)));
}

// one LDS-DMA request of this wave: 1 KiB to `lds_dst` (wave-uniform), lane i supplies sbase + voff (M0 has this one writer)
__device__ __forceinline__ void b2_dma(uint32_t lds_dst, const void* sbase, uint32_t voff) {
    // (the base is wave-uniform by construction; told to the compiler, which otherwise may keep it in vector registers)
    const uint64_t a = (uint64_t)(uintptr_t)sbase;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    sbase = reinterpret_cast<const void*>((uintptr_t)(((uint64_t)hi << 32) | lo));
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst), "v"(voff), "s"(sbase) : "memory", "m0");
}

// CT compute type, W map width, WM 32-channel output tiles per workgroup (1: 4 waves, 2: 8 waves), K4: the k4 s2 p1 family on parity planes
// (X holds [4 planes][K] channels, a phase lies in ONE plane, which sees 2 x 2 of the 3 x 3 taps; pack of vs_conv_k4s2_pack_weight), KC
// input channels per phase
template <int CT, int W, int WM, int K4, int KC>
__global__ __launch_bounds__(256 * WM, 2) void conv3_band2_kernel(const unsigned short* __restrict__ X, const unsigned short* __restrict__ Wp,
                                                                 const float* __restrict__ bias, void* __restrict__ Y, int yd, int B, int Creal, int H, int Cout,
                                                                 int mgroups, int bands, int ntiles, int chunks_total, int nph, int stagger) {
    typedef B2Geo<W, KC> G;
    // VS_BAND2_STAGGER=k (an experiment, default 0): the second half of the grid -- under round-robin placement the second workgroup of every CU -- starts
    // k x 1024 cycles late, so that the two co-resident workgroups do not meet at the matrix pipe and at the LDS in lockstep
    if (stagger > 0 && (int)blockIdx.x >= ((int)gridDim.x >> 1))
        for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(16);
    constexpr int NTHR = 256 * WM;
    constexpr int NKY = K4 ? 2 : 3, NF = K4 ? 2 : 3, NT = NKY * NF;               // tap rows, tap columns (fragments per group), taps per chunk
    constexpr int NJ = 2;                                                          // 32-pixel column tiles per wave
    constexpr int NCH = KC / 16, NG = NCH * NKY;                                   // 16-channel chunks and (chunk, tap row) groups per phase
    constexpr int WE = WM * NCH * NT * 512;                                        // weight elements per stage
    constexpr int WPIECES = WE / 8;
    constexpr int XR = (G::XPIECES + NTHR - 1) / NTHR, WR = (WPIECES + NTHR - 1) / NTHR;
    constexpr int STAGE = G::XE + WE;                                              // elements per stage
    constexpr int ROWB = W * 2;                                                    // bytes of an image row
    // channels per pass of the epilogue's transposition (three fp32 planes of CPP x 64 pixels per wave must fit the stage just consumed)
    constexpr int CPP = (size_t)STAGE * 2 >= (size_t)(4 * WM) * 3 * 8 * 64 * 4 ? 8 : 4;
    static_assert((size_t)STAGE * 2 >= (size_t)(4 * WM) * 3 * CPP * 64 * 4, "the epilogue's planes must fit a stage");
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];          // [2][x image | weight fragments]

    // logical workgroup id: blocks are dealt round-robin over the 8 XCDs (observed, speed only), so give every XCD a contiguous run of logical
    // ids: the output-channel groups of one band -- consecutive tiles -- then meet in one L2
    int L = blockIdx.x;
    const int nwg = (int)gridDim.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = L & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                                      // the wave's output-channel tile and 64-pixel range
    const int px0 = wn * 64;
    const int HW = H * W;
    const int mtiles = (Cout + 31) >> 5;

    // ---- DMA sources -----------------------------------------------------------------------------------------------------------------
    // x: destination piece u = r * NTHR + tid (linear in LDS) takes source piece (u % PPC) ^ swz(channel) of channel u / PPC.  Byte offsets from
    // the tile's origin -- W >= 16: row (first row of the band - 1) of channel 0 of map b; W <= 8: channel 0 of the tile's first map -- are the
    // same for every tile and phase; pieces that would leave the image (the row above the first band, below the last; maps past the batch)
    // are redirected to a row / map inside it: their values never reach an accumulator (the read side zeroes them)
    auto xoff = [&](int r, int cmax, int mpmax) -> uint32_t {
        const int u = r * NTHR + tid;
        int cl = u / G::PPC;
        const int sp = (u - cl * G::PPC) ^ b2_swz<W>(cl);
        if (cl > cmax) cl = cmax;
        if constexpr (G::HALO) {
            constexpr int PW = W / 8;
            const int rr = sp / PW, pc = sp - rr * PW;
            return (uint32_t)(((cl * H + rr) * W + pc * 8) * 2);
        } else {
            constexpr int PM = W * W / 8;                                          // pieces per map: 8 / 2
            int mp = sp / PM;
            const int pi = sp - mp * PM;
            if (mp > mpmax) mp = mpmax;
            return (uint32_t)(((mp * Creal + cl) * HW + pi * 8) * 2);
        }
    };
    uint32_t xvo[XR];
    unsigned rowtop = 0, rowbot = 0;                                               // bit r: piece r lies in the halo row above / below the band
#pragma unroll
    for (int r = 0; r < XR; ++r) {
        xvo[r] = xoff(r, KC - 1, G::IPB - 1);
        if constexpr (G::HALO) {
            const int u = r * NTHR + tid, cl = u / G::PPC, sp = (u - cl * G::PPC) ^ b2_swz<W>(cl), rr = sp / (W / 8);
            if (rr == 0) rowtop |= 1u << r;
            if (rr == G::RI + 1) rowbot |= 1u << r;
        }
    }
    const uint32_t lane16 = lane * 16;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

    struct Tile { int mg, band, b; };
    auto decode = [&](int t) -> Tile {
        Tile c;
        c.mg = t % mgroups;
        t /= mgroups;
        c.band = t % bands;
        c.b = t / bands;
        return c;
    };
    // request i (compile-time) of phase `ph` of tile `c` into stage `stg`
    auto dma_piece = [&](int i, const Tile& c, int ph, int stg) {
        if (i < XR) {
            const int r = i;
            if (G::XPIECES % NTHR != 0 && r == XR - 1 && wave * 64 >= G::XPIECES - (XR - 1) * NTHR) return;           // a partial last round
            uint32_t vo;
            const char* base;
            const bool partial = (ph + 1) * KC > Creal;
            if constexpr (G::HALO) {
                base = reinterpret_cast<const char*>(X) + ((((int64_t)c.b * Creal + ph * KC) * H + c.band * G::RI - 1) * W) * 2;
                vo = partial ? xoff(r, Creal - 1 - ph * KC, 0) : xvo[r];
                if (c.band == 0 && ((rowtop >> r) & 1u)) vo += ROWB;
                if (c.band == bands - 1 && ((rowbot >> r) & 1u)) vo -= ROWB;
            } else {
                const int img0 = c.band * G::IPB;
                base = reinterpret_cast<const char*>(X) + (((int64_t)img0 * Creal + ph * KC) * HW) * 2;
                const bool ragged = img0 + G::IPB > B;
                vo = (partial || ragged) ? xoff(r, partial ? Creal - 1 - ph * KC : KC - 1, ragged ? B - 1 - img0 : G::IPB - 1) : xvo[r];
            }
            b2_dma(lds0 + (uint32_t)(stg * STAGE * 2 + (r * NTHR + wave * 64) * 16), base, vo);
        } else {
            const int r = i - XR;
            if (WPIECES % NTHR != 0 && r == WR - 1 && wave * 64 >= WPIECES - (WR - 1) * NTHR) return;
            // weights: piece u of a stage belongs to output tile u / (NCH * NT * 64) of the workgroup, linear inside it (pack: [tile][chunk][tap][lane][8])
            const int u = r * NTHR + wave * 64;                                    // (wave-uniform)
            const int ml = u / (NCH * NT * 64), within = u - ml * (NCH * NT * 64);
            int mt = c.mg * WM + ml;
            if (mt > mtiles - 1) mt = mtiles - 1;                                  // (a tile past the last one: computed, never stored)
            const char* base = reinterpret_cast<const char*>(Wp) + (((int64_t)mt * chunks_total + ph * NCH) * NT) * 1024 + (int64_t)within * 16;
            b2_dma(lds0 + (uint32_t)((stg * STAGE + G::XE) * 2 + (r * NTHR + wave * 64) * 16), base, lane16);
        }
    };
    constexpr int NDMA = XR + WR;

    int t = L;
    if (t >= ntiles) return;
    Tile cur = decode(t);
#pragma unroll
    for (int i = 0; i < NDMA; ++i) dma_piece(i, cur, 0, 0);

    // ---- read side: byte offsets of this lane's 4-pixel piece inside a stage for every column tile and tap row -------------------------
    // transposing read: lane 4 q + p of a 16-lane group supplies k-row q, pixels 4 p .. 4 p + 3; cb = 16-pixel half of the tile, h = k half
    const int li = lane & 15, q = li >> 2, p = li & 3, cb = (lane >> 4) & 1, h = lane >> 5;
    const int c_lane = 8 * h + q;
    int xrd[NJ][3];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = px0 + j * 32 + 16 * cb + 4 * p;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            int inner;
            if constexpr (G::HALO) {
                inner = (n / W + ky) * W + n % W;
            } else if constexpr (W == 8) {
                int rr = ((n & 63) >> 3) + ky - 1;
                rr = rr < 0 ? 0 : (rr > 7 ? 7 : rr);
                inner = ((n >> 6) * 8 + rr) * 8 + (n & 7);
            } else {
                int rr = ((n & 15) >> 2) + ky - 1;
                rr = rr < 0 ? 0 : (rr > 3 ? 3 : rr);
                inner = ((n >> 4) * 4 + rr) * 4;
            }
            xrd[j][ky] = (c_lane * G::CE + (((inner >> 3) ^ b2_swz<W>(c_lane)) << 3) + (inner & 7)) * 2;
        }
    }
    // which of this lane's RESULT pixels (pixel l31 of tile j) sit in the first / last row of the band (W >= 16) or of their map: bit j
    const int l31 = lane & 31;
    unsigned top_l = 0, bot_l = 0;
    int col[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = px0 + j * 32 + l31;
        col[j] = n % W;
        const int row = G::HALO ? n / W : (n % (W * W)) / W;
        if (row == 0) top_l |= 1u << j;
        if (row == G::RI - 1) bot_l |= 1u << j;
    }

    u32x4 fa[2][3], fb[2][NJ];
    // fragments of group g (chunk g / NKY, tap row g % NKY) of the stage at `st`
    auto load_group = [&](int g, int set, const char* st, int ky0) {
        const int ch = g / NKY, kyi = g % NKY;
        const char* wfr = st + G::XE * 2 + ((wm * NCH + ch) * NT + kyi * NF) * 1024 + lane16;
#pragma unroll
        for (int kx = 0; kx < NF; ++kx) fa[set][kx] = *reinterpret_cast<const u32x4*>(wfr + kx * 1024);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            // (K4: tap rows {ky0, ky0 + 1} with a run-time ky0 -- selected between two registers, never a run-time register index)
            const int xo = K4 ? (ky0 ? xrd[j][kyi + 1] : xrd[j][kyi]) : xrd[j][kyi];
            fb[set][j] = vs_tr16_pair(reinterpret_cast<const unsigned short*>(st + xo + ch * 16 * G::CE * 2), 4 * G::CE);
        }
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int gp = 0;                                                                    // phases done so far: phase gp lives in stage gp & 1
    for (;;) {
        const int tn = t + nwg;
        const bool has_next = tn < ntiles;
        Tile nxt = cur;
        if (has_next) nxt = decode(tn);
        // the bias values this lane adds on the way out, fetched now.  16-bit output: pass (q4, sub) -> channel 8 q4 + CPP sub + (lane & (8 CPP - 1)) / 8;
        // fp32 output: read `rd` of pass (q4, sub) -> channel 8 q4 + CPP sub + 4 rd + lane / 16, i.e. slot 2 q4 + (sub | rd) of eight.
        // (Fetched inside the epilogue's passes, or fetched here but waited for by an `asm volatile` s_waitcnt the compiler cannot see, every use
        //  carried a compiler-made `s_waitcnt vmcnt(0)` that also drained the previous pass's stores: 15 acknowledged stores in a row per tile,
        //  the 6 100-cycle epilogue of round 5's phase stamps.  The waits at the phase end and at the head of the epilogue are the BUILTIN, which
        //  clears the compiler's scoreboard: with a value possibly still on its way and a conditional store in between, the only safe counted
        //  wait is vmcnt(0).)
        float bpre[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int mb = (cur.mg * WM + wm) * 32;
            const int m16 = mb + (k / (8 / CPP)) * 8 + (k % (8 / CPP)) * CPP + ((lane & (CPP * 8 - 1)) >> 3);
            const int m32 = mb + (k / 2) * 8 + (k % 2) * 4 + (lane >> 4);
            const int m = yd != VS_F32 ? m16 : m32;
            const bool used = yd != VS_F32 ? k < 4 * (8 / CPP) : true;
            bpre[k] = (bias && used && m < Cout) ? bias[m] : 0.f;
        }
        f32x16 acc[3][NJ];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[kx][j][v] = 0.f;
        // the tap row above / below is outside the image for the first / last band (W >= 16) or outside the map (W <= 8, every tile)
        const bool tile_top = G::HALO ? cur.band == 0 : true, tile_bot = G::HALO ? cur.band == bands - 1 : true;

        for (int ph = 0; ph < nph; ++ph) {
            const int stg = gp & 1;
            const char* st = reinterpret_cast<const char*>(smem) + stg * STAGE * 2;
            // K4: the plane of this phase (wave-uniform): odd rows (plane >> 1) see tap rows {0, 1}, even rows {1, 2}; likewise the columns
            const int plane = K4 ? (ph * KC) / (Creal >> 2) : 0;
            const int ky0 = K4 ? ((plane >> 1) ? 0 : 1) : 0;
            const bool odd_cols = K4 && (plane & 1);
            const bool last = ph + 1 == nph;
            const bool more = !last || has_next;
            const Tile dt = last ? nxt : cur;                                       // whose phase is requested during this one
            const int dph = last ? 0 : ph + 1;
            load_group(0, 0, st, ky0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int set = g & 1;
                if (g + 1 < NG) load_group(g + 1, set ^ 1, st, ky0);
                __builtin_amdgcn_sched_barrier(0);
                // zero the pixels whose tap row lies outside the map / image (a lane of the transposed fragment is ONE pixel: 16 cb + (lane & 15))
                const int ky = ky0 + g % NKY;
                if (ky == 0 && tile_top) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        if ((top_l >> j) & 1u) fb[set][j] = u32x4{0u, 0u, 0u, 0u};
                }
                if (ky == 2 && tile_bot) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        if ((bot_l >> j) & 1u) fb[set][j] = u32x4{0u, 0u, 0u, 0u};
                }
                if constexpr (K4) {
                    if (odd_cols) {                                                  // column taps {0, 1}: x[col - 1], x[col]
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            acc[0][j] = mfma16_32<CT>(fa[set][0], fb[set][j], acc[0][j]);
                            acc[1][j] = mfma16_32<CT>(fa[set][1], fb[set][j], acc[1][j]);
                        }
                    } else {                                                         // column taps {1, 2}: x[col], x[col + 1]
#pragma unroll
                        for (int j = 0; j < NJ; ++j) {
                            acc[1][j] = mfma16_32<CT>(fa[set][0], fb[set][j], acc[1][j]);
                            acc[2][j] = mfma16_32<CT>(fa[set][1], fb[set][j], acc[2][j]);
                        }
                    }
                } else {
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) acc[kx][j] = mfma16_32<CT>(fa[set][kx], fb[set][j], acc[kx][j]);
                }
                // the next phase's requests ride behind the first groups' MFMAs (they have the rest of the phase to land)
                if (more) {
                    constexpr int SPREAD = NG >= 2 ? 2 : NG;
                    constexpr int PER = (NDMA + SPREAD - 1) / SPREAD;
#pragma unroll
                    for (int i = g * PER; i < (g + 1) * PER && i < NDMA; ++i)
                        if (g < SPREAD) dma_piece(i, dt, dph, stg ^ 1);
                }
            }
            // the next stage has landed (this wave's share; behind the barrier everybody's) and nobody reads this stage any more
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0x0070);                                    // vmcnt(0) lgkmcnt(0), as an instruction the compiler accounts for
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            ++gp;
        }

        // ---- out[x] = G_1[x] + G_0[x - 1] + G_2[x + 1] inside the image row, + bias; the tile leaves through the stage just consumed -----
        // The accumulator holds ONE pixel per lane (32 consecutive pixels of a channel across 32 lanes).  Phase stamps of two earlier forms of
        // this epilogue: cross-lane rotations (64 ds_bpermute round trips of ~80 cycles, waited for one by one) + bias
        // loads = 8 900 of a tile's 29 500 cycles; LDS float adds (ds_add_f32 one pixel to the right / left) = 43 000 cycles -- an LDS atomic
        // costs ~450 cycles per wave instruction.  Now the x shift is a matter of WHERE a value is written: three plain ds_write_b32 per value
        // into wave-private planes [3][CPP channels][64 px] -- G_1 in place, G_0 one pixel to the right, G_2 one to the left -- nothing waits
        // for a cross-lane result; a lane then reads 8 (fp32 output: 4) pixels of each plane, masks the positions no neighbour wrote (row
        // ends), adds in the order of the sums above, adds the bias (fetched at the start of the tile), converts and stores 16 bytes.
        {
            // (nothing is outstanding here -- the last phase ended with the wait above -- but the compiler cannot know that the phase loop ran at
            //  least once: said again where it dominates the passes, for its scoreboard)
            __builtin_amdgcn_s_waitcnt(0x0070);
            const int mbase = (cur.mg * WM + wm) * 32;
            float* of = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + ((gp - 1) & 1) * STAGE * 2 + wave * (3 * CPP * 64 * 4));
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {                                        // channels 8 q4 .. 8 q4 + 7 of the wave's 32: registers v = 4 q4 .. 4 q4 + 3
#pragma unroll
                for (int sub = 0; sub < 8 / CPP; ++sub) {                           // (CPP = 4: the lanes of one 32-lane half at a time)
                    const bool mine = CPP == 8 || (lane >> 5) == sub;
                    if (mine) {
#pragma unroll
                        for (int vv = 0; vv < 4; ++vv) {
                            const int v = q4 * 4 + vv;
                            const int chp = (CPP == 8 ? 4 * (lane >> 5) : 0) + vv;      // channel inside the pass
#pragma unroll
                            for (int j = 0; j < NJ; ++j) {
                                const int e = chp * 64 + j * 32 + l31;
                                of[CPP * 64 + e] = acc[1][j][v];
                                if (col[j] != W - 1) of[e + 1] = acc[0][j][v];                   // G_0[x] is the left neighbour's share of out[x + 1]
                                if (col[j] != 0) of[2 * CPP * 64 + e - 1] = acc[2][j][v];        // G_2[x] the right neighbour's share of out[x - 1]
                            }
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // (a wave reads back only what it wrote itself)
                    if (yd != VS_F32) {
                        // 16-bit output: a lane takes 8 pixels of one channel: CPP * 8 tasks per pass
                        const int c = lane & (CPP * 8 - 1), chp = c >> 3, px = (c & 7) * 8;
                        const int m = mbase + q4 * 8 + sub * CPP + chp, tpx = px0 + px;   // pixel inside the workgroup's 256-pixel tile
                        int64_t off;
                        bool live = m < Cout && (CPP == 8 || lane < 32);
                        if constexpr (G::HALO) {
                            off = ((int64_t)cur.b * Cout + m) * HW + cur.band * G::RI * W + tpx;
                        } else {
                            const int img = cur.band * G::IPB + tpx / (W * W);
                            off = ((int64_t)img * Cout + m) * HW + tpx % (W * W);
                            live = live && img < B;
                        }
                        const float* r0 = of + chp * 64 + px;
                        float g0[8], g1[8], g2[8];
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const f32x4 a0 = *reinterpret_cast<const f32x4*>(r0 + 4 * k), a1 = *reinterpret_cast<const f32x4*>(r0 + CPP * 64 + 4 * k),
                                        a2 = *reinterpret_cast<const f32x4*>(r0 + 2 * CPP * 64 + 4 * k);
#pragma unroll
                            for (int i = 0; i < 4; ++i) { g0[4 * k + i] = a0[i]; g1[4 * k + i] = a1[i]; g2[4 * k + i] = a2[i]; }
                        }
                        // positions nobody wrote: the first pixel of a row has no left neighbour, the last no right one (rows are W pixels; a piece of
                        // 8 pixels starts at a multiple of 8, so only its pixels 0 / 4 and 3 / 7 can be row ends)
                        if (tpx % W == 0) g0[0] = 0.f;
                        if (W == 4) g0[4] = 0.f;
                        if ((tpx + 8) % W == 0) g2[7] = 0.f;
                        if (W == 4) g2[3] = 0.f;
                        const float bv = bpre[q4 * (8 / CPP) + sub];
                        u32x4 o;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float lo = g1[2 * k] + g0[2 * k] + g2[2 * k] + bv, hi = g1[2 * k + 1] + g0[2 * k + 1] + g2[2 * k + 1] + bv;
                            o[k] = (unsigned)vs_f2h(lo, yd) | ((unsigned)vs_f2h(hi, yd) << 16);
                        }
                        if (live) *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(Y) + off) = o;
                    } else {
                        // fp32 output (the last layer of a stack): 4 pixels per lane and store, CPP * 16 tasks per pass
#pragma unroll
                        for (int rd = 0; rd < (CPP * 16 + 63) / 64; ++rd) {
                            const int c = rd * 64 + lane, chp = c >> 4, px = (c & 15) * 4;
                            const int m = mbase + q4 * 8 + sub * CPP + chp, tpx = px0 + px;
                            int64_t off;
                            bool live = m < Cout && c < CPP * 16;
                            if constexpr (G::HALO) {
                                off = ((int64_t)cur.b * Cout + m) * HW + cur.band * G::RI * W + tpx;
                            } else {
                                const int img = cur.band * G::IPB + tpx / (W * W);
                                off = ((int64_t)img * Cout + m) * HW + tpx % (W * W);
                                live = live && img < B;
                            }
                            const int cc = c < CPP * 16 ? c : 0;                       // (idle lanes read a valid address)
                            const float* r0 = of + (cc >> 4) * 64 + (cc & 15) * 4;
                            f32x4 a0 = *reinterpret_cast<const f32x4*>(r0), a1 = *reinterpret_cast<const f32x4*>(r0 + CPP * 64),
                                  a2 = *reinterpret_cast<const f32x4*>(r0 + 2 * CPP * 64);
                            if (tpx % W == 0) a0[0] = 0.f;
                            if ((tpx + 4) % W == 0) a2[3] = 0.f;
                            const float bv = bpre[q4 * 2 + (CPP == 8 ? rd : sub)];
                            const f32x4 o = a1 + a0 + a2 + bv;
                            if (live) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(Y) + off) = o;
                        }
                    }
                    // (the next pass overwrites the planes: the reads above are back once their values have been used by the stores)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
        }
        if (!has_next) break;
        // every wave has read its transposed block back before anybody requests the next tile's second phase into this stage
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur = nxt;
        t = tn;
    }
}

template <int W, int WM, int K4, int KC>
int b2_launch(int compute, const void* x, const void* w_packed, const float* bias, void* y, int y_dtype, int B, int Cin, int H, int Cout, hipStream_t stream) {
    typedef B2Geo<W, KC> G;
    constexpr int NT = K4 ? 4 : 9;
    constexpr size_t lds = (size_t)2 * (G::XE + WM * (KC / 16) * NT * 512) * 2;
    static_assert(lds <= 160 * 1024, "two stages must fit the CU's LDS");
    auto kb = conv3_band2_kernel<VS_BF16, W, WM, K4, KC>;
    auto kh = conv3_band2_kernel<VS_F16, W, WM, K4, KC>;
    static bool attr_set = false;
    static int cus = 0;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute((const void*)kh, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return vs_fail(VS_ERR_LAUNCH, "vs_conv3_band (v2): cannot raise the dynamic LDS limit");
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        attr_set = true;
    }
    const int mtiles = (int)vs_cdiv(Cout, 32), mgroups = (int)vs_cdiv(mtiles, WM);
    const int bands = G::IPB > 1 ? (int)vs_cdiv(B, G::IPB) : H / G::RI;
    const int64_t ntiles = (int64_t)(G::IPB > 1 ? 1 : B) * bands * mgroups;
    if (ntiles >= (1ll << 31)) return vs_fail(VS_ERR_UNSUPPORTED, "vs_conv3_band (v2): too many tiles");
    // persistent: what the chip holds at once (workgroups per CU by LDS: two where they fit, and never more than 2 waves per SIMD)
    const int per_cu = WM == 2 ? 1 : ((size_t)160 * 1024 / lds >= 2 ? 2 : 1);
    const char* ge = getenv("VS_BAND2_GRID");                                       // (diagnosis: workgroups per CU, 0 = one workgroup per tile)
    int64_t slots = (int64_t)cus * (ge && atoi(ge) > 0 ? atoi(ge) : per_cu);
    if (ge && atoi(ge) == 0) slots = ntiles;
    const dim3 grid((unsigned)(ntiles < slots ? ntiles : slots));
    const int chunks_total = (int)vs_cdiv(Cin, 64) * 4;                             // the pack holds whole 64-channel phases (zeros beyond Cin)
    const int nph = (int)vs_cdiv(Cin, KC);
    const char* se = getenv("VS_BAND2_STAGGER");                                    // read per call: A/B runs switch it
    const int stagger = (se && per_cu == 2) ? atoi(se) : 0;
    if (compute == VS_BF16)
        hipLaunchKernelGGL(kb, grid, dim3(256 * WM), lds, stream, (const unsigned short*)x, (const unsigned short*)w_packed, bias, y, y_dtype, B, Cin, H, Cout,
                           mgroups, bands, (int)ntiles, chunks_total, nph, stagger);
    else
        hipLaunchKernelGGL(kh, grid, dim3(256 * WM), lds, stream, (const unsigned short*)x, (const unsigned short*)w_packed, bias, y, y_dtype, B, Cin, H, Cout,
                           mgroups, bands, (int)ntiles, chunks_total, nph, stagger);
    return VS_OK;
}

// WM = 2 (512 threads, 64 output channels: the pixel image staged once for both channel tiles) vs WM = 1 (two independent workgroups per
// CU).  VS_BAND2_WM = 1 / 2 forces a form (read per call: tools/band_bench.py A/B).
template <int W, int K4>
int b2_pick(int compute, const void* x, const void* w_packed, const float* bias, void* y, int y_dtype, int B, int Cin, int H, int Cout, hipStream_t stream) {
    constexpr int KC = W == 64 ? 16 : 32;                                           // (64-wide maps: 24 KiB of pixels per 32 channels -- half phases keep two workgroups per CU)
    const char* fe = getenv("VS_BAND2_WM");
    const int force = fe ? atoi(fe) : 0;
    // measured (tools/band_bench.py, profiles/r05_band_bench.txt): the 8-wave form wins wherever it still has a tile per CU -- 64-wide maps by
    // 30 %, the decoders' 16 / 32-wide layers by 5-10 % -- and loses on the short launches of the encoders (fewer, heavier workgroups)
    constexpr int IPB = W == 8 ? 4 : (W == 4 ? 16 : 1), RI = W == 8 ? 8 : (W == 4 ? 4 : 256 / W);
    const int64_t tiles2 = (IPB > 1 ? vs_cdiv(B, IPB) : (int64_t)B * (H / RI)) * vs_cdiv(vs_cdiv(Cout, 32), 2);
    bool two = Cout > 32 && tiles2 >= 256;
    if (force == 1) two = false;
    if (force == 2) two = Cout > 32;
    if (two) return b2_launch<W, 2, K4, 32>(compute, x, w_packed, bias, y, y_dtype, B, Cin, H, Cout, stream);
    return b2_launch<W, 1, K4, KC>(compute, x, w_packed, bias, y, y_dtype, B, Cin, H, Cout, stream);
}

}  // namespace

// Called by the entry points of vs_conv_img.hip (vs_conv3_band*, vs_conv_k4s2_band*) when no BatchNorm sums are wanted from the epilogue.
// k4 != 0: x = parity planes [B][Cin = 4 K][H][W], pack of vs_conv_k4s2_pack_weight (skip form: K a multiple of 64).
int vs_band2_go(int compute, const void* x, const void* w_packed, const float* bias, void* y, int y_dtype, int B, int Cin, int H, int W, int Cout, int k4,
                hipStream_t stream) {
#define VS_B2_CASE(WV)                                                                                                   \
    case WV:                                                                                                             \
        return k4 ? b2_pick<WV, 1>(compute, x, w_packed, bias, y, y_dtype, B, Cin, H, Cout, stream)                        \
                  : b2_pick<WV, 0>(compute, x, w_packed, bias, y, y_dtype, B, Cin, H, Cout, stream);
    switch (W) {
        VS_B2_CASE(64)
        VS_B2_CASE(32)
        VS_B2_CASE(16)
        VS_B2_CASE(8)
        VS_B2_CASE(4)
        default: return vs_fail(VS_ERR_UNSUPPORTED, "vs_conv3_band (v2): map width %d", W);
    }
#undef VS_B2_CASE
}
