// vs_conv_band2.hip -- round 5: Conv2d k3 s1 p1 (and the k4 s2 p1 family on parity planes) on row bands with BOTH operands staged through
// LDS (gfx950 only).  Reference layers: every 3x3 block of the VGG / SST encoders and decoders (conv.py:127-171, 267-426), the DCGAN
// k4 s2 p1 layers (conv.py:119-122, 260-263), forward and input gradient.
//
// What the counters said about the round-2 kernel (conv3_band_kernel, vs_conv_img.hip; profiles/r05_pre_*_mfma_util.md): matrix pipes
// busy 21-44 % of the CU-busy cycles, texture-address units busy 40-62 %, LDS bank conflicts 21-69 % of the LDS cycles for every map
// width but 16.  Its four waves each fetch the SAME weight fragments from global memory (3 KiB per wave per six MFMAs: 64 B/clk/CU, the
// whole vector-memory path) and its pixel image has a channel pitch of 640 / 768 bytes (2- / 4-way conflicts of the transposing reads).
// This kernel:
//   * weights reach LDS ONCE per workgroup by LDS-DMA (the pre-pack is already in MFMA fragment order: a fragment is 1 KiB, lane-linear,
//     read back with one conflict-free ds_read_b128) -- the vector-memory traffic of a workgroup drops 4x (WM = 1) to 8x (WM = 2) per MFMA;
//   * a workgroup owns 256 pixels x 32 WM output channels (WM = 2: waves 2 x 2, a wave 32 channels x 128 pixels = four 32-pixel column
//     tiles, one weight fragment feeds four MFMAs; WM = 1: a wave 32 channels x 64 pixels, two workgroups per CU for short launches);
//   * the pixel image is stored with its 16-byte pieces permuted per channel (piece ^ f(channel & 3): the permutation sits on the DMA's
//     per-lane SOURCE address and on the read address) -- conflict-free for every width (tools: /tmp model in DESIGN section 4e);
//   * maps of 4 x 4 / 8 x 8 pixels are staged DENSE (no zero rows: sixteen / four whole maps per tile), rows outside a map or outside the
//     image are read from a clamped address and the fragment is zeroed by a select on the result lane (a lane of the transposed fragment
//     IS one pixel) -- nothing is fetched from a block of zeros any more;
//   * phases of 32 input channels, two LDS stages (x 16-24 KiB + weights 18 WM KiB each), one barrier per phase; fragments double
//     buffered in registers; the x shift of the taps on the RESULT (three accumulators per column tile) and the epilogue as before.
#include "vs_gemm_glds.h"
#include <stdlib.h>

namespace {

constexpr int B2_KC = 32;                    // input channels per phase

template <int W>
struct B2Geo {
    static constexpr int IPB = W == 8 ? 4 : (W == 4 ? 16 : 1);          // maps per tile
    static constexpr int RI = W == 8 ? 8 : (W == 4 ? 4 : 256 / W);      // rows of a map (W <= 8) or of the band (W >= 16)
    static constexpr bool HALO = IPB == 1;                               // W >= 16: the band's neighbour rows are staged (R + 2 rows)
    static constexpr int RPI = RI + (HALO ? 2 : 0);
    static constexpr int CE = IPB * RPI * W;                             // elements of one channel's image
    static constexpr int PPC = CE / 8;                                   // its 16-byte pieces: 48 / 40 / 36 / 32 / 32
    static constexpr int XE = B2_KC * CE;                                // x elements per stage
    static constexpr int XPIECES = B2_KC * PPC;
    static constexpr int XR = (XPIECES + 255) / 256;                     // DMA rounds (pieces per thread): 6 / 5 / 5 / 4 / 4
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
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst), "v"(voff), "s"(sbase) : "memory", "m0");
}

// CT compute type, W map width, WM 32-channel output tiles per workgroup (1 or 2), K4: the k4 s2 p1 family on parity planes (X holds
// [4 planes][K] channels, a 32-channel phase lies in ONE plane, which sees 2 x 2 of the 3 x 3 taps; pack of vs_conv_k4s2_pack_weight)
template <int CT, int W, int WM, int K4>
__global__ __launch_bounds__(256, WM == 2 ? 1 : 2) void conv3_band2_kernel(const unsigned short* __restrict__ X, const unsigned short* __restrict__ Wp,
                                                                          const float* __restrict__ bias, void* __restrict__ Y, int yd, int B, int Creal, int H,
                                                                          int Cout, int mgroups, int bands, int xcd_remap, int chunks_total, int nph, int ablate) {
    // ablate (VS_BAND2_ABLATE, diagnosis only -- results are wrong): 1 no x DMA after phase 0, 2 no weight DMA after phase 0, 4 fragments read once,
    // 8 no MFMAs, 16 no barrier / wait at the end of a phase, 32 no DMA of phase 0 either, 64 no epilogue, 128 return at once
    if (ablate & 128) return;
    typedef B2Geo<W> G;
    constexpr int NKY = K4 ? 2 : 3, NF = K4 ? 2 : 3, NT = NKY * NF;               // tap rows, tap columns (fragments per group), taps per chunk
    constexpr int NJ = 2 * WM;                                                     // 32-pixel column tiles per wave
    constexpr int NG = 2 * NKY;                                                    // (chunk, tap row) groups per phase
    constexpr int WE = WM * 2 * NT * 512;                                          // weight elements per stage
    constexpr int WPIECES = WE / 8, WR = (WPIECES + 255) / 256;
    constexpr int STAGE = G::XE + WE;                                              // elements per stage
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];          // [2][x image | weight fragments]

    int id = blockIdx.x;
    if (xcd_remap) {       // the output-channel groups of one band stage the same rows: give every XCD a contiguous run of logical ids
        const int nwg = (int)gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = id & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    const int mg = id % mgroups;
    id /= mgroups;
    const int band = id % bands, b = id / bands;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = WM == 2 ? wave >> 1 : 0, wn = WM == 2 ? wave & 1 : wave;       // the wave's output-channel tile and pixel range
    const int px0 = wn * (NJ * 32);                                                // first pixel of the wave inside the 256-pixel tile
    const int HW = H * W;

    // ---- DMA sources -----------------------------------------------------------------------------------------------------------------
    // x: destination piece u = r * 256 + tid (linear in LDS) takes source piece (u % PPC) ^ swz(channel) of channel u / PPC.  Byte offsets
    // from the phase's base (channel 32 ph of the tile's first image) are the same in every phase; rows / maps outside the image are CLAMPED
    // (their values never reach an accumulator: the read side zeroes them)
    const int img0 = G::IPB > 1 ? band * G::IPB : b;
    auto xoff = [&](int r, int cmax) -> uint32_t {
        const int u = r * 256 + tid;
        int cl = u / G::PPC;
        const int sp = (u - cl * G::PPC) ^ b2_swz<W>(cl);
        if (cl > cmax) cl = cmax;
        if constexpr (G::HALO) {
            constexpr int PW = W / 8;
            const int rr = sp / PW, pc = sp - rr * PW;
            int y = band * G::RI + rr - 1;
            y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
            return (uint32_t)(((cl * H + y) * W + pc * 8) * 2);
        } else {
            constexpr int PM = W * W / 8;                                          // pieces per map: 8 / 2
            const int mp = sp / PM, pi = sp - mp * PM;
            int img = img0 + mp;
            if (img > B - 1) img = B - 1;
            return (uint32_t)((((img - img0) * Creal + cl) * HW + pi * 8) * 2);
        }
    };
    uint32_t xvo[G::XR];
#pragma unroll
    for (int r = 0; r < G::XR; ++r) xvo[r] = xoff(r, B2_KC - 1);
    const char* xbase = reinterpret_cast<const char*>(X + (int64_t)img0 * Creal * HW);
    const int64_t xstep = (int64_t)B2_KC * HW * 2;
    // weights: piece u of a stage belongs to output tile u / (NT * 128) of the workgroup, linear inside it (the pack is [tile][chunk][tap][lane][8])
    const uint32_t lane16 = lane * 16;
    const int mtiles = (Cout + 31) >> 5;
    auto wsrc = [&](int r, int ph) -> const char* {
        const int u = r * 256 + wave * 64;                                         // (wave-uniform)
        const int ml = u / (NT * 128), within = u - ml * (NT * 128);
        int mt = mg * WM + ml;
        if (mt > mtiles - 1) mt = mtiles - 1;                                      // (a tile past the last one: computed, never stored)
        return reinterpret_cast<const char*>(Wp) + (((int64_t)mt * chunks_total + ph * 2) * NT) * 1024 + (int64_t)within * 16;
    };
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

    auto dma_x = [&](int r, int ph, bool clamp_ch) {
        if (G::XPIECES % 256 != 0 && r == G::XR - 1 && wave * 64 >= G::XPIECES - (G::XR - 1) * 256) return;       // (W = 16: a half round)
        const uint32_t vo = clamp_ch ? xoff(r, Creal - 1 - ph * B2_KC) : xvo[r];
        b2_dma(lds0 + (uint32_t)((ph & 1) * STAGE * 2 + (r * 256 + wave * 64) * 16), xbase + ph * xstep, vo);
    };
    auto dma_w = [&](int r, int ph) {
        if (WPIECES % 256 != 0 && r == WR - 1 && wave * 64 >= WPIECES - (WR - 1) * 256) return;
        b2_dma(lds0 + (uint32_t)(((ph & 1) * STAGE + G::XE) * 2 + (r * 256 + wave * 64) * 16), wsrc(r, ph), lane16);
    };
    constexpr int NDMA = G::XR + WR;
    auto dma_piece = [&](int i, int ph) {                                          // request i of a phase (compile-time i)
        const bool partial = (ph + 1) * B2_KC > Creal;
        if (i < G::XR) { if (!(ablate & 1) || ph == 0) dma_x(i, ph, partial); }
        else if (!(ablate & 2) || ph == 0) dma_w(i - G::XR, ph);
    };

    if (!(ablate & 32)) {
#pragma unroll
        for (int i = 0; i < NDMA; ++i) dma_piece(i, 0);
    }

    // ---- read side: element offsets of this lane's 4-pixel piece for every column tile and tap row ------------------------------------
    // transposing read: lane 4 q + p of a 16-lane group supplies k-row q, pixels 4 p .. 4 p + 3; cb = 16-pixel half of the tile, h = k half
    const int li = lane & 15, q = li >> 2, p = li & 3, cb = (lane >> 4) & 1, h = lane >> 5;
    const int c_lane = 8 * h + q;
    int xrd[NJ][3];                                                                // byte offsets inside a stage (k-row c_lane, chunk 0)
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
    // which of this lane's RESULT pixels (pixel l31 of tile j) have their upper / lower neighbour row outside the map or image: bit j
    const int l31 = lane & 31;
    unsigned top = 0, bot = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = px0 + j * 32 + l31;
        int row, rows;
        if constexpr (G::HALO) { row = band * G::RI + n / W; rows = H; }
        else { row = (n % (W * W)) / W; rows = W; }
        if (row == 0) top |= 1u << j;
        if (row == rows - 1) bot |= 1u << j;
    }
    const bool any_top = __builtin_amdgcn_readfirstlane(__any(top != 0)), any_bot = __builtin_amdgcn_readfirstlane(__any(bot != 0));

    f32x16 acc[3][NJ];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[kx][j][v] = 0.f;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    u32x4 fa[2][3], fb[2][NJ];
    // fragments of group g (chunk g / NKY, tap row g % NKY) of the stage at `st`
    auto load_group = [&](int g, int set, const char* st, int ky0) {
        const int ch = g / NKY, kyi = g % NKY;
        const char* wfr = st + G::XE * 2 + ((wm * 2 + ch) * NT + kyi * NF) * 1024 + lane16;
#pragma unroll
        for (int kx = 0; kx < NF; ++kx) fa[set][kx] = *reinterpret_cast<const u32x4*>(wfr + kx * 1024);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            // (K4: tap rows {ky0, ky0 + 1} with a run-time ky0 -- selected between two registers, never a run-time register index)
            const int xo = K4 ? (ky0 ? xrd[j][kyi + 1] : xrd[j][kyi]) : xrd[j][kyi];
            fb[set][j] = vs_tr16_pair(reinterpret_cast<const unsigned short*>(st + xo + ch * 16 * G::CE * 2), 4 * G::CE);
        }
    };

    for (int ph = 0; ph < nph; ++ph) {
        const char* st = reinterpret_cast<const char*>(smem) + (ph & 1) * STAGE * 2;
        // K4: the plane of this phase (wave-uniform): odd rows (plane >> 1) see tap rows {0, 1}, even rows {1, 2}; likewise the columns
        const int plane = K4 ? (ph * B2_KC) / (Creal >> 2) : 0;
        const int ky0 = K4 ? ((plane >> 1) ? 0 : 1) : 0;
        const bool odd_cols = K4 && (plane & 1);
        const bool more = ph + 1 < nph;
        if (!(ablate & 4) || ph == 0) load_group(0, 0, st, ky0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int set = g & 1;
            if (g + 1 < NG && (!(ablate & 4) || (ph == 0 && g == 0))) load_group(g + 1, set ^ 1, st, ky0);
            __builtin_amdgcn_sched_barrier(0);
            // zero the pixels whose tap row lies outside the map / image (a lane of the transposed fragment is ONE pixel: 16 cb + (lane & 15))
            const int ky = ky0 + g % NKY;
            if (ky == 0 && any_top) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    if ((top >> j) & 1u) fb[set][j] = u32x4{0u, 0u, 0u, 0u};
            }
            if (ky == 2 && any_bot) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    if ((bot >> j) & 1u) fb[set][j] = u32x4{0u, 0u, 0u, 0u};
            }
            if (ablate & 8) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(fa[set][0]), "+v"(fb[set][j]));
            } else if constexpr (K4) {
                if (odd_cols) {                                                      // column taps {0, 1}: x[col - 1], x[col]
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        acc[0][j] = mfma16_32<CT>(fa[set][0], fb[set][j], acc[0][j]);
                        acc[1][j] = mfma16_32<CT>(fa[set][1], fb[set][j], acc[1][j]);
                    }
                } else {                                                             // column taps {1, 2}: x[col], x[col + 1]
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
                constexpr int PER = (NDMA + 2) / 3;                                  // spread over the first three groups
#pragma unroll
                for (int i = g * PER; i < (g + 1) * PER && i < NDMA; ++i)
                    if (g < 3) dma_piece(i, ph + 1);
            }
        }
        // the next stage has landed (this wave's share; behind the barrier everybody's) and nobody reads this stage any more
        if (!(ablate & 16)) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    if (ablate & 16) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (ablate & 64) {
        if (acc[1][0][0] == 12345.f) reinterpret_cast<float*>(Y)[0] = acc[0][0][1] + acc[2][0][2];      // (keeps the loop alive)
        return;
    }

    // ---- out[x] = G_1[x] + G_0[x - 1] + G_2[x + 1] inside the image row, + bias, typed store ------------------------------------------
    const int idx_l = ((lane & 32) | ((lane - 1) & 31)) * 4, idx_r = ((lane & 32) | ((lane + 1) & 31)) * 4;
    // The accumulator holds ONE pixel per lane (32 consecutive pixels of a channel across 32 lanes): stored from there, a wave instruction
    // moves two 64-byte runs (16-bit output) -- the ablation of round 5 (every phase's DMA, fragment reads, MFMAs and
    // barriers switched off) showed 49 of the 81 us of a 256 -> 256 layer on 8 x 8 maps, 273 of 319 us of 64 -> 64 on 64 x 64 maps, in the
    // prologue + THIS epilogue: ~50 ns per 128-byte store instruction, 900-5000 of them per CU.  So the tile is transposed through the
    // (now idle) LDS stages, wave-private: one ds_write per value, then 16-byte reads along the pixels and 1 KiB global stores.
    const int mbase = (mg * WM + wm) * 32;
    constexpr int PXW = NJ * 32;                                                   // pixels of the wave's tile
    char* ost = reinterpret_cast<char*>(smem) + wave * (32 * PXW * 4);             // [32 channels][PXW] in the output type (<= 16 KiB per wave)
    int col[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) col[j] = (px0 + j * 32 + l31) % W;
    float bvs[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int m = mbase + 4 * (lane >> 5) + (v & 3) + 8 * (v >> 2);
        bvs[v] = (bias && m < Cout) ? bias[m] : 0.f;
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int chl = 4 * (lane >> 5) + (v & 3) + 8 * (v >> 2);
        float rl[NJ], rr[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const float gl = acc[0][j][v], gr = acc[2][j][v];
            rl[j] = b2_gather(idx_l, gl);                                           // lane i <- lane i - 1 of the tile (lane 0 <- lane 31)
            rr[j] = b2_gather(idx_r, gr);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            // the left neighbour of a tile's first pixel is the last pixel of the wave's previous tile (same row only when W = 64)
            float left = l31 != 0 ? rl[j] : (j > 0 ? rl[j - 1] : 0.f);
            float right = l31 != 31 ? rr[j] : (j + 1 < NJ ? rr[j + 1] : 0.f);
            if (col[j] == 0) left = 0.f;
            if (col[j] == W - 1) right = 0.f;
            const float o = acc[1][j][v] + left + right + bvs[v];
            const int e = chl * PXW + j * 32 + l31;
            if (yd == VS_F32) reinterpret_cast<float*>(ost)[e] = o;
            else reinterpret_cast<unsigned short*>(ost)[e] = vs_f2h(o, yd);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                             // (a wave reads back only what it wrote itself)
    const int es = yd == VS_F32 ? 4 : 2, per = 16 / es, cpr = PXW / per;            // pixels per 16-byte piece, pieces per channel row
    for (int c = lane; c < 32 * cpr; c += 64) {
        const int chl = c / cpr, px = (c - chl * cpr) * per;
        const int m = mbase + chl, t = px0 + px;                                   // pixel inside the workgroup's 256-pixel tile
        int64_t off;
        bool live = m < Cout;
        if constexpr (G::HALO) {
            off = ((int64_t)b * Cout + m) * HW + band * G::RI * W + t;
        } else {
            const int img = img0 + t / (W * W);
            off = ((int64_t)img * Cout + m) * HW + t % (W * W);
            live = live && img < B;
        }
        const u32x4 val = *reinterpret_cast<const u32x4*>(ost + (chl * PXW + px) * es);
        if (live) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(Y) + off * es) = val;
    }
}

template <int W, int WM, int K4>
int b2_launch(int compute, const void* x, const void* w_packed, const float* bias, void* y, int y_dtype, int B, int Cin, int H, int Cout, hipStream_t stream) {
    typedef B2Geo<W> G;
    constexpr int NT = K4 ? 4 : 9;
    constexpr size_t lds = (size_t)2 * (G::XE + WM * 2 * NT * 512) * 2;
    static_assert(lds <= 160 * 1024, "two stages must fit the CU's LDS");
    auto kb = conv3_band2_kernel<VS_BF16, W, WM, K4>;
    auto kh = conv3_band2_kernel<VS_F16, W, WM, K4>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute((const void*)kh, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return vs_fail(VS_ERR_LAUNCH, "vs_conv3_band (v2): cannot raise the dynamic LDS limit");
        attr_set = true;
    }
    const int mtiles = (int)vs_cdiv(Cout, 32), mgroups = (int)vs_cdiv(mtiles, WM);
    const int bands = G::IPB > 1 ? (int)vs_cdiv(B, G::IPB) : H / G::RI;
    const int64_t nwg = (int64_t)(G::IPB > 1 ? 1 : B) * bands * mgroups;
    if (nwg >= (1ll << 31)) return vs_fail(VS_ERR_UNSUPPORTED, "vs_conv3_band (v2): grid too large");
    const dim3 grid((unsigned)nwg);
    static const int xcd_remap = getenv("VS_BAND_XCD") ? atoi(getenv("VS_BAND_XCD")) : 1;
    const int remap = xcd_remap && mgroups > 1 && grid.x >= 64;
    const int chunks_total = (int)vs_cdiv(Cin, 64) * 4;                             // the pack holds whole 64-channel phases (zeros beyond Cin)
    const int nph = (int)vs_cdiv(Cin, B2_KC);
    const char* ab = getenv("VS_BAND2_ABLATE");
    const int ablate = ab ? atoi(ab) : 0;
    if (compute == VS_BF16)
        hipLaunchKernelGGL(kb, grid, dim3(256), lds, stream, (const unsigned short*)x, (const unsigned short*)w_packed, bias, y, y_dtype, B, Cin, H, Cout, mgroups,
                           bands, remap, chunks_total, nph, ablate);
    else
        hipLaunchKernelGGL(kh, grid, dim3(256), lds, stream, (const unsigned short*)x, (const unsigned short*)w_packed, bias, y, y_dtype, B, Cin, H, Cout, mgroups,
                           bands, remap, chunks_total, nph, ablate);
    return VS_OK;
}

// WM = 2 (one workgroup per CU, 64 output channels) pays when there is enough work per workgroup to cover its prologue and enough
// workgroups to fill the chip; otherwise WM = 1 (two workgroups per CU cover each other).  VS_BAND2_WM = 1 / 2 forces a form.
template <int W, int K4>
int b2_pick(int compute, const void* x, const void* w_packed, const float* bias, void* y, int y_dtype, int B, int Cin, int H, int Cout, hipStream_t stream) {
    typedef B2Geo<W> G;
    const char* fe = getenv("VS_BAND2_WM");                                          // (read per call: tools/band_bench.py A/B)
    const int force = fe ? atoi(fe) : 0;
    const int64_t tiles = (int64_t)(G::IPB > 1 ? vs_cdiv(B, G::IPB) : (int64_t)B * (H / G::RI));
    const int64_t wg2 = tiles * vs_cdiv(vs_cdiv(Cout, 32), 2);
    bool two = Cout > 32 && Cin >= 128 && wg2 >= 384;
    if (force == 1) two = false;
    if (force == 2) two = Cout > 32;
    if (two) return b2_launch<W, 2, K4>(compute, x, w_packed, bias, y, y_dtype, B, Cin, H, Cout, stream);
    return b2_launch<W, 1, K4>(compute, x, w_packed, bias, y, y_dtype, B, Cin, H, Cout, stream);
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
