// vs_conv_wgrad2.hip -- round 5: weight gradient of Conv2d k3 s1 p1 on row bands with both operands staged by LDS-DMA (gfx950 only).
// Reference: the backward pass of every 3x3 block of the VGG / SST encoders and decoders (conv.py:127-171, 267-426):
//     dW[m][c][ky][kx] = sum over maps and pixels of dz[m][y][x] * x[c][y + ky - 1][x + kx - 1].
//
// The round-2 kernel (wgrad3_band_kernel, vs_conv_img.hip) stages THREE column-shifted copies of the x band through registers (the tap
// shift sits on the contraction axis, and an MFMA fragment read must be 16-byte aligned), single-buffered, two barriers per band, one
// workgroup of four waves per CU: the counters of round 5 (profiles/r05_pre_*_mfma_util.md) show its matrix pipes busy 19-31 % of the
// CU-busy cycles with the waves parked at waits / barriers 40-47 % of the time; a k-step costs ten ds_read_b128 for nine MFMAs.
// Here
//   * x is staged ONCE, unshifted, and dz beside it, both by LDS-DMA into two stages (the next band travels while this one is multiplied;
//     one barrier per band; nothing passes through registers);
//   * the column shift happens in REGISTERS: a lane's B fragment for the centre tap is one aligned ds_read_b128 (8 consecutive pixels of its
//     channel); the fragments of the left / right taps are that register shifted by one pixel (v_alignbyte) with the neighbour pixel
//     from one ds_read_b32 each side -- 1 + 3 x (1 + 2 small) LDS reads per nine MFMAs instead of ten wide ones, ~30 VALU under the MFMAs;
//   * rows outside the image (W >= 16: the band's halo rows at the top / bottom of a map) are a wave-uniform property of a k-step: their
//     MFMAs are skipped; rows outside a small map (W <= 8: maps are staged dense) are zeroed on the lane;
//   * a workgroup is eight waves = 2 output-channel sub-tiles (64 m) x 4 quarters of the band's sixteen k-steps, 32 input channels; the four
//     partial sums of a sub-tile meet in LDS at the very end: ONE slab [tap][m][c] per workgroup share (vs_conv3_wgrad_band_finish adds them).
#include "vs_gemm_glds.h"
#include <stdlib.h>

namespace {

constexpr int WG2_MAX_PIECES = 64;
struct Wg2Pieces {
    const unsigned short* x[WG2_MAX_PIECES];
    const unsigned short* dz[WG2_MAX_PIECES];
    int maps_per_piece;
};

template <int W>
struct Wg2Geo {
    static constexpr int IPB = W == 8 ? 4 : (W == 4 ? 16 : 1);          // maps per item
    static constexpr int R = W == 8 ? 8 : (W == 4 ? 4 : 256 / W);       // rows of a map (W <= 8) or of the band
    static constexpr bool HALO = IPB == 1;
    static constexpr int ROWS = IPB * (R + (HALO ? 2 : 0));             // staged rows per channel
    static constexpr int PPC = ROWS * W / 8;                            // 16-byte pieces of a channel: 48 / 40 / 36 / 32 / 32
    static constexpr int PPCP = PPC + ((PPC & 1) ? 0 : 1);              // + one idle piece: an ODD pitch keeps the 32 channels of a fragment read apart
    static constexpr int XB = (32 * PPCP + 63) / 64 * 64 * 16;          // bytes of the x tile (whole 1 KiB wave requests: the last one is half idle)
    static constexpr int ZPP = 33;                                      // pieces per dz row (32 + 1 idle)
    static constexpr int ZB = 64 * ZPP * 16;                            // bytes of the dz tile [64 m][256 px]
    static constexpr int STAGE = XB + ZB;
    static constexpr int XPIECES = 32 * PPCP, ZPIECES = 64 * ZPP;
    static constexpr int XR = (XPIECES + 511) / 512, ZR = (ZPIECES + 511) / 512;
};

__device__ __forceinline__ void wg2_dma(uint32_t lds_dst, const void* sbase, uint32_t voff) {
    const uint64_t a = (uint64_t)(uintptr_t)sbase;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    sbase = reinterpret_cast<const void*>((uintptr_t)(((uint64_t)hi << 32) | lo));
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst), "v"(voff), "s"(sbase) : "memory", "m0");
}

// K4 = 1: the k4 s2 p1 family on parity planes (csrc/vs_conv_k4s2.hip): Cin = 4 K plane channels, K a multiple of 32, so the 32 channels of a workgroup
// lie in ONE plane, which sees 2 x 2 of the 3 x 3 taps: four MFMAs per k-step, four slab positions written (vs_conv_k4s2_wgrad_finish reads no others)
template <int CT, int W, int K4>
__global__ __launch_bounds__(512, 2) void wgrad2_band_kernel(Wg2Pieces pieces, float* __restrict__ slabs, int B, int Cin, int H, int Cout, int ctiles, int ksplit, int lds_neighbours) {
    typedef Wg2Geo<W> G;
    constexpr int ROWB = W * 2;
    constexpr int NKY = K4 ? 2 : 3, NT = K4 ? 4 : 9;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];         // [2][x tile | dz tile]
    int id = blockIdx.x;
    const int ks = id % ksplit;
    id /= ksplit;
    const int ct = id % ctiles, mt = id / ctiles;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int msub = wave & 1, kpart = wave >> 1;                                // 32-row sub-tile of dz, quarter of the band's k-steps
    const int bands = G::HALO ? H / G::R : 1;
    const int HW = H * W;
    const int64_t items = G::HALO ? (int64_t)B * bands : (int64_t)((B + G::IPB - 1) / G::IPB);
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const int cvalid = Cin - ct * 32 < 32 ? Cin - ct * 32 : 32, mvalid = Cout - mt * 64 < 64 ? Cout - mt * 64 : 64;

    // ---- DMA sources: destination piece u = r * 512 + tid, linear in LDS -------------------------------------------------------------------
    // x: piece pp of channel u / PPCP (the idle piece of a channel fetches piece 0 again); rows / maps / channels past the operand are
    // redirected inside it -- what they hold never reaches a stored result (skipped k-steps, zeroed lanes, rows / columns that are not stored)
    auto xoff = [&](int r, int mpmax) -> uint32_t {
        const int u = r * 512 + tid;
        int cl = u / G::PPCP, pp = u - cl * G::PPCP;
        if (pp >= G::PPC) pp = 0;
        if (cl > cvalid - 1) cl = cvalid - 1;
        if (cl < 0) cl = 0;
        if constexpr (G::HALO) {
            constexpr int PW = W / 8;
            const int rr = pp / PW, pc = pp - rr * PW;
            return (uint32_t)(((cl * H + rr) * W + pc * 8) * 2);                  // from row (first row of the band - 1) of the tile's first channel
        } else {
            constexpr int PM = W * W / 8;
            int mp = pp / PM;
            const int pi = pp - mp * PM;
            if (mp > mpmax) mp = mpmax;
            return (uint32_t)(((mp * Cin + cl) * HW + pi * 8) * 2);
        }
    };
    auto zoff = [&](int r, int mpmax) -> uint32_t {
        const int u = r * 512 + tid;
        int ml = u / G::ZPP, pp = u - ml * G::ZPP;
        if (pp >= 32) pp = 0;
        if (ml > mvalid - 1) ml = mvalid - 1;
        if constexpr (G::HALO) {
            return (uint32_t)((ml * HW + pp * 8) * 2);                            // from the band's first row of the tile's first output channel
        } else {
            constexpr int PM = W * W / 8;
            int mp = pp / PM;
            const int pi = pp - mp * PM;
            if (mp > mpmax) mp = mpmax;
            return (uint32_t)(((mp * Cout + ml) * HW + pi * 8) * 2);
        }
    };
    uint32_t xvo[G::XR], zvo[G::ZR];
    unsigned rowtop = 0, rowbot = 0;
#pragma unroll
    for (int r = 0; r < G::XR; ++r) {
        xvo[r] = xoff(r, G::IPB - 1);
        if constexpr (G::HALO) {
            const int u = r * 512 + tid, cl = u / G::PPCP;
            int pp = u - cl * G::PPCP;
            if (pp >= G::PPC) pp = 0;
            const int rr = pp / (W / 8);
            if (rr == 0) rowtop |= 1u << r;
            if (rr == G::R + 1) rowbot |= 1u << r;
        }
    }
#pragma unroll
    for (int r = 0; r < G::ZR; ++r) zvo[r] = zoff(r, G::IPB - 1);

    auto dma_item = [&](int64_t it, int stg) {
        const int bg = G::HALO ? (int)(it / bands) : (int)it * G::IPB, band = G::HALO ? (int)(it % bands) : 0;
        const int piece = bg / pieces.maps_per_piece, b = bg - piece * pieces.maps_per_piece;       // (an item never straddles pieces)
        const bool ragged = !G::HALO && bg + G::IPB > B;
        const char* xb;
        const char* zb;
        if constexpr (G::HALO) {
            xb = reinterpret_cast<const char*>(pieces.x[piece]) + ((((int64_t)b * Cin + ct * 32) * H + band * G::R - 1) * W) * 2;
            zb = reinterpret_cast<const char*>(pieces.dz[piece]) + ((((int64_t)b * Cout + mt * 64) * H + band * G::R) * W) * 2;
        } else {
            xb = reinterpret_cast<const char*>(pieces.x[piece]) + (((int64_t)b * Cin + ct * 32) * HW) * 2;
            zb = reinterpret_cast<const char*>(pieces.dz[piece]) + (((int64_t)b * Cout + mt * 64) * HW) * 2;
        }
#pragma unroll
        for (int r = 0; r < G::XR; ++r) {
            if (G::XPIECES % 512 != 0 && r == G::XR - 1 && wave * 64 >= G::XPIECES - (G::XR - 1) * 512) continue;
            uint32_t vo = ragged ? xoff(r, B - 1 - bg) : xvo[r];
            if constexpr (G::HALO) {
                if (band == 0 && ((rowtop >> r) & 1u)) vo += ROWB;
                if (band == bands - 1 && ((rowbot >> r) & 1u)) vo -= ROWB;
            }
            wg2_dma(lds0 + (uint32_t)(stg * G::STAGE + (r * 512 + wave * 64) * 16), xb, vo);
        }
#pragma unroll
        for (int r = 0; r < G::ZR; ++r) {
            if (G::ZPIECES % 512 != 0 && r == G::ZR - 1 && wave * 64 >= G::ZPIECES - (G::ZR - 1) * 512) continue;
            const uint32_t vo = ragged ? zoff(r, B - 1 - bg) : zvo[r];
            wg2_dma(lds0 + (uint32_t)(stg * G::STAGE + G::XB + (r * 512 + wave * 64) * 16), zb, vo);
        }
    };

    // K4: first tap row / column this plane sees (odd planes: {0, 1}, even planes: {1, 2}); wave-uniform
    const int plane = K4 ? (ct * 32) / (Cin >> 2) : 0;
    const int ky_lo = K4 ? ((plane >> 1) ? 0 : 1) : 0, kx_lo = K4 ? ((plane & 1) ? 0 : 1) : 0;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;

    const int rl = lane & 31, h = lane >> 5;
    int64_t it = ks;
    if (it < items) dma_item(it, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int stg = 0;
    for (; it < items; it += ksplit) {
        const int64_t nx = it + ksplit;
        if (nx < items) dma_item(nx, stg ^ 1);
        const unsigned char* xs = smem + stg * G::STAGE;
        const unsigned char* zs = xs + G::XB;
        const int bg = G::HALO ? (int)(it / bands) : (int)it * G::IPB, band = G::HALO ? (int)(it % bands) : 0;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int t = kpart * 4 + tt, p0 = t * 16;                           // k-step: pixels p0 .. p0 + 15 of the item
            if constexpr (!G::HALO) {
                // the k-step's map: past the batch in a ragged last item -> nothing to add (wave-uniform)
                const int mp = W == 4 ? t : t / 4;
                if (bg + mp >= B) continue;
            }
            const u32x4 af = *reinterpret_cast<const u32x4*>(zs + (msub * 32 + rl) * (G::ZPP * 16) + (p0 + 8 * h) * 2);
#pragma unroll
            for (int kyi = 0; kyi < NKY; ++kyi) {
                const int ky = ky_lo + kyi;
                u32x4 own;
                unsigned lft = 0, rgt = 0;                                        // dwords holding the pixel before / after the lane's eight
                if constexpr (G::HALO) {
                    // the sixteen pixels lie in ONE row of the band; tap row ky reads staged row (row + ky) (row 0 = the halo above the band)
                    const int row = p0 / W, q = p0 % W + 8 * h;
                    const int yimg = band * G::R + row + ky - 1;                  // wave-uniform
                    if (yimg < 0 || yimg >= H) continue;                          // a row outside the image contributes nothing
                    const unsigned char* src = xs + rl * (G::PPCP * 16) + ((row + ky) * W + q) * 2;
                    own = *reinterpret_cast<const u32x4*>(src);
                    // The sixteen pixels of a k-step are split over the two lane halves (h): the pixel AFTER the lower half's eight is the upper half's
                    // first, the pixel BEFORE the upper half's eight is the lower half's last -- one v_permlane32_swap instead of two LDS reads.  What
                    // is left for LDS is the pixel outside the sixteen (one ds_read_b32 per lane; none at W = 16, where a k-step is a whole row): the
                    // channel pitch is a multiple of 16 bytes, so a 4-byte read of 32 channels is 4-way bank-conflicted whatever the pad, and round 5's
                    // counters had these reads at 46-52 % of the kernel's LDS cycles (profiles/r05_*_mfma_util.md).
                    if (lds_neighbours) {                                        // VS_WGRAD2_NB=1: the round-5 form (both neighbours from LDS), kept for A/B runs
                        const unsigned lv = *reinterpret_cast<const unsigned*>(src - (q > 0 ? 4 : 0)), rv = *reinterpret_cast<const unsigned*>(src + (q + 8 < W ? 16 : 0));
                        lft = q > 0 ? lv : 0u;
                        rgt = q + 8 < W ? rv : 0u;
                    } else {
                        const auto sw = __builtin_amdgcn_permlane32_swap(own[0], own[3], false, false);
                        unsigned outer = 0u;
                        if constexpr (W > 16) {
                            // (unconditional load, the select on the address and on the value: a conditional load makes the compiler wait for each one)
                            const bool need = h ? (q + 8 < W) : (q > 0);
                            const unsigned ov = *reinterpret_cast<const unsigned*>(src + (need ? (h ? 16 : -4) : 0));
                            outer = need ? ov : 0u;
                        }
                        lft = h ? sw[0] : outer;                                 // upper half: the lower half's last dword (its HIGH half is the pixel before)
                        rgt = h ? outer : sw[1];                                 // lower half: the upper half's first dword (its LOW half is the pixel after)
                    }
                } else if constexpr (W == 8) {
                    // a lane's eight pixels are one whole row of an 8 x 8 map: no column neighbours; the tap row may leave the map (per lane half)
                    const int mp = t / 4, r8 = 2 * (t % 4) + h + ky - 1;
                    const bool ok = r8 >= 0 && r8 < 8;
                    const int rc = r8 < 0 ? 0 : (r8 > 7 ? 7 : r8);
                    own = *reinterpret_cast<const u32x4*>(xs + rl * (G::PPCP * 16) + ((mp * 8 + rc) * 8) * 2);
                    if (!ok) own = u32x4{0u, 0u, 0u, 0u};
                } else {
                    // W = 4: a lane's eight pixels are rows 2 h, 2 h + 1 of the k-step's 4 x 4 map; tap row ky reads rows 2 h + ky - 1, 2 h + ky
                    const int ra = 2 * h + ky - 1, rb = ra + 1;
                    const unsigned char* mb = xs + rl * (G::PPCP * 16) + (t * 16) * 2;
                    const u32x2 a = *reinterpret_cast<const u32x2*>(mb + (ra < 0 ? 0 : ra) * 8), bb = *reinterpret_cast<const u32x2*>(mb + (rb > 3 ? 3 : rb) * 8);
                    own[0] = ra < 0 ? 0u : a[0]; own[1] = ra < 0 ? 0u : a[1];
                    own[2] = rb > 3 ? 0u : bb[0]; own[3] = rb > 3 ? 0u : bb[1];
                }
                u32x4 fl, fr;                                                      // the lane's pixels shifted: x[col - 1], x[col + 1]
                if constexpr (W == 4) {
                    fl[0] = own[0] << 16;                                          // rows of four pixels: the shift stops at the row's ends
                    fl[1] = __builtin_amdgcn_alignbyte(own[1], own[0], 2);
                    fl[2] = own[2] << 16;
                    fl[3] = __builtin_amdgcn_alignbyte(own[3], own[2], 2);
                    fr[0] = __builtin_amdgcn_alignbyte(own[1], own[0], 2);
                    fr[1] = own[1] >> 16;
                    fr[2] = __builtin_amdgcn_alignbyte(own[3], own[2], 2);
                    fr[3] = own[3] >> 16;
                } else {
                    const unsigned m01 = __builtin_amdgcn_alignbyte(own[1], own[0], 2), m12 = __builtin_amdgcn_alignbyte(own[2], own[1], 2),
                                   m23 = __builtin_amdgcn_alignbyte(own[3], own[2], 2);
                    fl[0] = __builtin_amdgcn_alignbyte(own[0], lft, 2);            // (pixel before : pixels 0 ..): lft's HIGH half is the pixel before
                    fl[1] = m01; fl[2] = m12; fl[3] = m23;
                    fr[0] = m01; fr[1] = m12; fr[2] = m23;
                    fr[3] = __builtin_amdgcn_alignbyte(rgt, own[3], 2);            // (.. pixel 7 : pixel after): rgt's LOW half
                }
                if constexpr (K4) {
                    if (kx_lo == 0) {                                              // column taps {0, 1}: x[col - 1], x[col]
                        acc[kyi * 2 + 0] = mfma16_32<CT>(af, fl, acc[kyi * 2 + 0]);
                        acc[kyi * 2 + 1] = mfma16_32<CT>(af, own, acc[kyi * 2 + 1]);
                    } else {                                                       // column taps {1, 2}: x[col], x[col + 1]
                        acc[kyi * 2 + 0] = mfma16_32<CT>(af, own, acc[kyi * 2 + 0]);
                        acc[kyi * 2 + 1] = mfma16_32<CT>(af, fr, acc[kyi * 2 + 1]);
                    }
                } else {
                    acc[ky * 3 + 0] = mfma16_32<CT>(af, fl, acc[ky * 3 + 0]);
                    acc[ky * 3 + 1] = mfma16_32<CT>(af, own, acc[ky * 3 + 1]);
                    acc[ky * 3 + 2] = mfma16_32<CT>(af, fr, acc[ky * 3 + 2]);
                }
            }
        }
        // the next item has landed (this wave's share; behind the barrier everybody's) and nobody reads this stage any more
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stg ^= 1;
    }

    // ---- the four k-quarters of a sub-tile meet in LDS; quarter 0 stores the slab of this share: [tap][m][c], 128-byte runs along c ------
    float* red = reinterpret_cast<float*>(smem);                                  // [8 waves][16][64] per tap round = 32 KiB
    float* out = slabs + (int64_t)ks * ((int64_t)Cout * Cin * 9);
    const int c = ct * 32 + rl;
#pragma unroll
    for (int tp = 0; tp < NT; ++tp) {
        const int t3 = K4 ? (ky_lo + (tp >> 1)) * 3 + kx_lo + (tp & 1) : tp;        // slab position [tap of the 3 x 3 form][m][c]
        if (kpart != 0) {
#pragma unroll
            for (int v = 0; v < 16; ++v) red[(wave * 16 + v) * 64 + lane] = acc[tp][v];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kpart == 0) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                float s = acc[tp][v];
#pragma unroll
                for (int k = 1; k < 4; ++k) s += red[((k * 2 + msub) * 16 + v) * 64 + lane];          // fixed order: reproducible
                const int m = mt * 64 + msub * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
                if (m < Cout && c < Cin) out[((int64_t)t3 * Cout + m) * Cin + c] = s;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

int wg2_ksplit(int B, int Cin, int H, int W, int Cout) {
    const int64_t tiles = vs_cdiv(Cout, 64) * vs_cdiv(Cin, 32);
    const int64_t items = W == 8 ? vs_cdiv(B, 4) : (W == 4 ? vs_cdiv(B, 16) : (int64_t)B * (H / (256 / W)));
    static const int target_wgs = getenv("VS_WGRAD2_WGS") ? atoi(getenv("VS_WGRAD2_WGS")) : 256;      // one workgroup of eight waves per CU
    // ONE round of workgroups: a workgroup's two stages fill a CU, so the 257th workgroup waits for a whole share of the bands to finish
    // (260 -> 256 channels: 36 tiles x 8 shares = 288 workgroups ran 1.8 x as long as 36 x 7 = 252)
    int64_t ks = target_wgs / tiles;
    if (ks < 1) ks = 1;
    if (ks > items) ks = items;
    const int64_t slab_bytes = (int64_t)Cout * Cin * 9 * 4;
    while (ks > 1 && ks * slab_bytes > ((int64_t)96 << 20)) --ks;                // at most 96 MiB of slabs
    return (int)(ks < 1 ? 1 : ks);
}

template <int W, int K4>
int wg2_launch(int compute, const Wg2Pieces& pieces, float* slabs, int B, int Cin, int H, int Cout, hipStream_t stream) {
    typedef Wg2Geo<W> G;
    constexpr size_t lds = (size_t)2 * G::STAGE;
    static_assert(lds <= 160 * 1024 && lds >= 32 * 1024, "two stages fit the CU; the final reduction needs 32 KiB");
    auto kb = wgrad2_band_kernel<VS_BF16, W, K4>;
    auto kh = wgrad2_band_kernel<VS_F16, W, K4>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
            hipFuncSetAttribute((const void*)kh, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return vs_fail(VS_ERR_LAUNCH, "vs_conv3_wgrad_band (v2): cannot raise the dynamic LDS limit");
        attr_set = true;
    }
    const int ks = wg2_ksplit(B, Cin, H, W, Cout);
    const int mtiles = (int)vs_cdiv(Cout, 64), ctiles = (int)vs_cdiv(Cin, 32);
    const dim3 grid((unsigned)((int64_t)mtiles * ctiles * ks));
    // the column neighbours of a lane's eight pixels: from LDS (1) or from the other lane half by v_permlane32_swap (0).  Same-box A/B of
    // round 6 (tools/band_bench.py wgrad, VS_BAND_BENCH_VARIANTS="VS_WGRAD2_NB=0,1"): the swap form wins 4-9 % at W = 16 (no LDS read left:
    // 88.7 vs 93.3, 123.5 vs 135.3, 107.6 vs 113.7 us) and loses 2-5 % at W = 32 / 64 (one conflicted read still needed + the selects).
    const char* env = getenv("VS_WGRAD2_NB");                     // read per call: A/B runs switch it
    const int nb = env ? atoi(env) : (W > 16 ? 1 : 0);
    if (compute == VS_BF16)
        hipLaunchKernelGGL(kb, grid, dim3(512), lds, stream, pieces, slabs, B, Cin, H, Cout, ctiles, ks, nb);
    else
        hipLaunchKernelGGL(kh, grid, dim3(512), lds, stream, pieces, slabs, B, Cin, H, Cout, ctiles, ks, nb);
    return VS_OK;
}

}  // namespace

// slabs the v2 kernel writes for this geometry (one per share of the bands)
int vs_wgrad2_slabs(int B, int Cin, int H, int W, int Cout) { return wg2_ksplit(B, Cin, H, W, Cout); }

// x / dz: up to 64 equal pieces of `maps_per_piece` maps each (vs_conv3_wgrad_band_pieces); slabs [vs_wgrad2_slabs][9][Cout][Cin] fp32
// k4 != 0: x = parity planes [B][Cin = 4 K][H][W] (K a multiple of 32), dz = the small map; slabs as above (four of nine positions written)
int vs_wgrad2_go(int compute, int npieces, const void* const* x, const void* const* dz, int maps_per_piece, float* slabs, int B, int Cin, int H, int W, int Cout,
                 int k4, hipStream_t stream) {
    if (npieces < 1 || npieces > WG2_MAX_PIECES) return vs_fail(VS_ERR_UNSUPPORTED, "vs_conv3_wgrad_band (v2): %d pieces", npieces);
    Wg2Pieces p = {};
    for (int i = 0; i < npieces; ++i) {
        p.x[i] = (const unsigned short*)x[i];
        p.dz[i] = (const unsigned short*)dz[i];
    }
    p.maps_per_piece = maps_per_piece;
#define VS_WG2_CASE(WV) \
    case WV: return k4 ? wg2_launch<WV, 1>(compute, p, slabs, B, Cin, H, Cout, stream) : wg2_launch<WV, 0>(compute, p, slabs, B, Cin, H, Cout, stream);
    switch (W) {
        VS_WG2_CASE(64)
        VS_WG2_CASE(32)
        VS_WG2_CASE(16)
        VS_WG2_CASE(8)
        VS_WG2_CASE(4)
        default: return vs_fail(VS_ERR_UNSUPPORTED, "vs_conv3_wgrad_band (v2): map width %d", W);
    }
}
