// vs_conv_img.hip -- Conv2d k3 s1 p1 on a FEW 16x16 images with many channels (gfx950 only).
//
// The SST recipe's ConvResnet integrator (resnet.py:53-88 of the reference) applies 64 -> 512 -> 512 -> 64 convolutions to a
// batch of 8 maps of 16 x 16 pixels, 78 times per training step each way.  One such convolution is 1-10 GFLOP on 2048 pixels:
// as "column matrix + GEMM" or as the tap GEMM of vs_conv_tap.hip it is a chain of 10-45 us launches that fill a fraction of
// the chip.  This kernel is shaped for that case:
//   * workgroup = (image, 32 output channels, split of the input channels); 4 waves, wave w owns the image rows 4w .. 4w+3
//     (two 32-pixel MFMA column tiles).  8 images x 16 channel tiles x 2 splits = 256 workgroups for 512 -> 512.
//   * the split's input channels of the image go to LDS ONCE as [channel][18 rows][16 px] (zero rows above and below): up to
//     256 channels = 144 KiB, no ring, one barrier.
//   * the x shift of a tap is taken on the OUTPUT side: G_kx[m][y][x] = sum_{c, ky} W[m][c][ky][kx] X[c][y + ky - 1][x] needs
//     the pixel operand unshifted in x, so one transposing LDS read (ds_read_b64_tr_b16 pair) of the rows y + ky - 1 serves the
//     three kx taps, and out[y][x] = G_0[x - 1] + G_1[x] + G_2[x + 1] is two DPP row shifts per accumulator register in the
//     epilogue (a 16-lane DPP row IS an image row of the MFMA result; shifted-in lanes read 0 = the zero padding).
//   * the weights never touch LDS: they are pre-packed in MFMA fragment order (vs_conv3_img16_pack_weight), every lane loads
//     its 16 bytes of a fragment straight from global memory, twelve (chunk, ky) groups = 36 fragments ahead of their use (four in the two-workgroups-per-CU form)
//     (the four waves of a workgroup read the same fragments: L1 hits).
//   * split partial sums go to fp32 slabs [split][B][Cout][256]; the consumer adds them: vs_bn_train_fwd_small_slabs (bias +
//     16-bit rounding + BatchNorm statistics + running update + affine + activation in one launch, vs_norm.hip) or vs_slab_sum.
// The input gradient is the same kernel on dz with the weight packed transposed and flipped (flip = 1).
#include "vs_gemm_glds.h"

namespace {

constexpr int IMG_CPITCH = 288;          // LDS elements per channel: 18 rows x 16 pixels

__device__ __forceinline__ float dpp_row_shr1(float v) {      // lane i <- lane i - 1 inside its 16-lane row, 0 into lane 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_row_shl1(float v) {      // lane i <- lane i + 1, 0 into lane 15
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true));
}

// derivative of the activation from its PRE-activation input (as vs_norm.hip's BatchNorm backward kernels take it)
__device__ __forceinline__ float img_act_grad(float z, int act) {
    switch (act) {
        case VS_ACT_RELU: return z > 0.f ? 1.f : 0.f;
        case VS_ACT_LEAKY: return z > 0.f ? 1.f : 0.2f;
        case VS_ACT_SIGMOID: { const float sg = 1.f / (1.f + expf(-z)); return sg * (1.f - sg); }
        case VS_ACT_TANH: { const float t = tanhf(z); return 1.f - t * t; }
        case VS_ACT_ELU: return z > 0.f ? 1.f : expf(z);
        default: return 1.f;
    }
}

__device__ __forceinline__ float lane_gather(int byte_index, float v) {       // value of lane byte_index / 4
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_index, __builtin_bit_cast(int, v)));
}

// Weight fragments: three 1 KiB wave loads (the kx taps of one (chunk, ky) group) written as asm so that they STAY where they are
// issued -- AHEAD (12 or 4) groups ahead of their use; left to the compiler the loads sink to just in front of the MFMAs that read them
// and every group waits for a full L2 round trip.  The loads are invisible to the compiler's counter tracking, so the matching wait is
// written out too: at the use of a group exactly N = (AHEAD - 1) * 3 younger fragment loads are outstanding (33 or 9).  The wait names
// the registers as in/out operands: the MFMAs that read them cannot be scheduled above it.  Loads still in flight when the loop ends
// MUST be drained before the epilogue: their destination registers are dead to the compiler, which re-uses them (for store addresses).
__device__ __forceinline__ void img_load3(u32x4& a0, u32x4& a1, u32x4& a2, const void* sbase, unsigned voff) {
    asm volatile("global_load_dwordx4 %0, %3, %4\n\tglobal_load_dwordx4 %1, %3, %4 offset:1024\n\tglobal_load_dwordx4 %2, %3, %4 offset:2048"
                 : "=&v"(a0), "=&v"(a1), "=&v"(a2)
                 : "v"(voff), "s"(sbase)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void img_wait3n(u32x4& a0, u32x4& a1, u32x4& a2) {   // N younger loads may stay in flight
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a0), "+v"(a1), "+v"(a2) : "n"(N) : "memory");
}
// the same for groups of TWO fragments (the k4 s2 p1 form on parity planes: a plane sees two of the three column taps)
__device__ __forceinline__ void img_load2(u32x4& a0, u32x4& a1, const void* sbase, unsigned voff) {
    asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024" : "=&v"(a0), "=&v"(a1) : "v"(voff), "s"(sbase) : "memory");
}
template <int N>
__device__ __forceinline__ void img_wait2n(u32x4& a0, u32x4& a1) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a0), "+v"(a1) : "n"(N) : "memory");
}

// NP 16-byte pieces per thread of the staged image: piece u = channel (u >> 5), 8 pixels (u & 31) -> the channel's rows 1 .. 16 in LDS
template <int NP>
__device__ __forceinline__ void img_stage(unsigned short* xs, const unsigned short* xb, int u0) {
    u32x4 v[NP];
#pragma unroll
    for (int r = 0; r < NP; ++r) v[r] = *reinterpret_cast<const u32x4*>(xb + (int64_t)(u0 + r * 256) * 8);
#pragma unroll
    for (int r = 0; r < NP; ++r) {
        const int u = u0 + r * 256;
        *reinterpret_cast<u32x4*>(xs + (u >> 5) * IMG_CPITCH + 16 + (u & 31) * 8) = v[r];
    }
}

// AHEAD = 12, MINB = 1: the deep prefetch, one workgroup per CU.  AHEAD = 4, MINB = 2: two workgroups per CU (splits of <= 128 channels =
// 72 KiB of LDS each) that cover each other's staging and epilogue -- taken when that doubles the number of resident workgroups.
// The tile of one workgroup -- image b, output channels 32 mt .., input channels c0 .. c0 + cs - 1 -- up to the x shift on the result:
// o[j][v] = partial sum (over this split's channels) of output channel 32 mt + (v & 3) + 8 (v >> 2) + 4 (lane >> 5) at pixel
// (4 wave + 2 j) * 16 + (lane & 31).  The two kernels below differ in what they do with it.
template <int CT, int AHEAD>
__device__ __forceinline__ void img16_tile(const unsigned short* __restrict__ X, const u32x4* __restrict__ Wp, int Cin, int cs, int mt, int b, int split,
                                           unsigned short* xs, float (&o)[2][16]) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c0 = split * cs;

    // ---- weight fragments of the first twelve (chunk, ky) groups: requested first, they travel while the image is staged -----
    // group g, tap column kx: 1 KiB of fragments at wbase + g * 3072 + kx * 1024 (+ lane * 16)
    const int chunks_total = Cin >> 4;
    const char* wbase = reinterpret_cast<const char*>(Wp + ((int64_t)(mt * chunks_total + (c0 >> 4)) * 3) * 192);
    const unsigned voff = lane * 16;
    u32x4 a[AHEAD][3];
#pragma unroll
    for (int gg = 0; gg < AHEAD; ++gg) img_load3(a[gg][0], a[gg][1], a[gg][2], wbase + gg * 3072, voff);

    // ---- the split's channels of image b -> LDS (16-byte pieces: 32 per channel), halo rows zeroed ----------------------
    const unsigned short* xb = X + ((int64_t)b * Cin + c0) * 256;
    const int pieces = cs * 32;
    int base = 0;
    for (; base + 4096 <= pieces; base += 4096) img_stage<16>(xs, xb, base + tid);          // pieces is a multiple of 2048 (cs % 64 == 0)
    if (base < pieces) img_stage<8>(xs, xb, base + tid);
    for (int u = tid; u < cs * 4; u += 256) {
        const int cl = u >> 2, h = u & 3;
        *reinterpret_cast<u32x4*>(xs + cl * IMG_CPITCH + (h >> 1) * 17 * 16 + (h & 1) * 8) = u32x4{0u, 0u, 0u, 0u};
    }
    __syncthreads();

    f32x16 acc[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[kx][j][v] = 0.f;

    // transposing read: lane 4q + p of a 16-lane group supplies k-row q, pixels 4p .. 4p + 3; cb = image row of the pair, h = k half
    const int li = lane & 15, q = li >> 2, p = li & 3, cb = (lane >> 4) & 1, h = lane >> 5;
    const unsigned short* lb = xs + (8 * h + q) * IMG_CPITCH + (4 * wave + cb) * 16 + 4 * p;

    const int ngroups = (cs >> 4) * 3;                                            // (chunk, ky) groups of this split

    for (int g0 = 0; g0 < ngroups; g0 += 12) {
        const unsigned short* lc = lb + (g0 / 3) * 16 * IMG_CPITCH;
#pragma unroll
        for (int gg = 0; gg < 12; ++gg) {
            const int ch = gg / 3, ky = gg % 3;
            u32x4 bf[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = vs_tr16_pair(lc + ch * 16 * IMG_CPITCH + (2 * j + ky) * 16, 4 * IMG_CPITCH);
            const int sl = gg % AHEAD;                                           // register set of this group (12 % AHEAD == 0)
            img_wait3n<(AHEAD - 1) * 3>(a[sl][0], a[sl][1], a[sl][2]);           // all but the (AHEAD - 1) * 3 youngest fragment loads have landed
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[kx][j] = mfma16_32<CT>(a[sl][kx], bf[j], acc[kx][j]);
            int gn = g0 + AHEAD + gg;                                         // the group this register set serves next
            if (gn > ngroups - 1) gn = ngroups - 1;                               // past the end: a harmless repeat of the last group
            img_load3(a[sl][0], a[sl][1], a[sl][2], wbase + (int64_t)gn * 3072, voff);
        }
    }

    // the repeated fragment loads of the last groups are still in flight and their destination registers are dead to the compiler, which
    // would re-use them for whatever it schedules next (round 4: the x-shift sums below, hoisted above a bare wait, came back with the
    // values of three registers overwritten by a landing load).  The wait therefore NAMES every fragment register as an in/out operand:
    // they stay allocated until the loads have landed, and nothing that follows can move above it.
#pragma unroll
    for (int gg = 0; gg < AHEAD; ++gg) {
        if (gg == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[gg][0]), "+v"(a[gg][1]), "+v"(a[gg][2]) : : "memory");
        else asm volatile("" : "+v"(a[gg][0]), "+v"(a[gg][1]), "+v"(a[gg][2]) : : "memory");
    }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(acc[kx][j]));         // (the sums below are taken after the wait)

    // ---- out[y][x] = G_1[x] + G_0[x - 1] + G_2[x + 1] ------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) o[j][v] = acc[1][j][v] + dpp_row_shr1(acc[0][j][v]) + dpp_row_shl1(acc[2][j][v]);
}

// AHEAD = 12, MINB = 1: the deep prefetch, one workgroup per CU.  AHEAD = 4, MINB = 2: two workgroups per CU (splits of <= 128 channels =
// 72 KiB of LDS each) that cover each other's staging and epilogue -- taken when that doubles the number of resident workgroups.
template <int CT, int AHEAD, int MINB>
__global__ __launch_bounds__(256, MINB) void conv3_img16_kernel(const unsigned short* __restrict__ X, const u32x4* __restrict__ Wp, float* __restrict__ slabs,
                                                          int B, int Cin, int Cout, int cs, int splits, int mtiles, int wide_store) {
    extern __shared__ __attribute__((aligned(16))) unsigned short xs[];          // [cs][18][16]
    int id = blockIdx.x;
    const int split = id % splits;
    id /= splits;
    const int mt = id % mtiles, b = id / mtiles;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float o[2][16];
    img16_tile<CT, AHEAD>(X, Wp, Cin, cs, mt, b, split, xs, o);

    // ---- fp32 slab of this split ----------------------------------------------------------------------------------------------------
    float* out = slabs + (((int64_t)split * B + b) * Cout) * 256;
    if (wide_store) {
        // round 5: the accumulator holds ONE pixel per lane, so a store instruction from there moves two 128-byte runs (32 of them per wave).
        // The tile is transposed through the image's LDS (nobody reads it any more behind the barrier): wave-private [32 channels][64 px],
        // read back 4 pixels per lane: 8 store instructions of 1 KiB per wave.
        __syncthreads();
        float* of = reinterpret_cast<float*>(xs) + wave * (32 * 64);
        const int chb = 4 * (lane >> 5), l31 = lane & 31;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) of[(chb + (v & 3) + 8 * (v >> 2)) * 64 + j * 32 + l31] = o[j][v];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      // (a wave reads back only what it wrote itself)
#pragma unroll
        for (int rd = 0; rd < 8; ++rd) {
            const int c = rd * 64 + lane, chl = c >> 4, px = (c & 15) * 4;
            const int m = mt * 32 + chl;
            const f32x4 val = *reinterpret_cast<const f32x4*>(of + chl * 64 + px);
            if (m < Cout) *reinterpret_cast<f32x4*>(out + (int64_t)m * 256 + wave * 64 + px) = val;
        }
        return;
    }
    const int mrow = mt * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pix = (4 * wave + 2 * j) * 16 + (lane & 31);
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int m = mrow + (v & 3) + 8 * (v >> 2);
            if (m < Cout) out[(int64_t)m * 256 + pix] = o[j][v];
        }
    }
}

// ---- convolution + BatchNorm of one residual-block layer in ONE launch ---------------------------------------------------------------
// The ConvResnet integrator (resnet.py:53-88) is 78 block calls each way per SST step, every layer of a block a convolution launch
// (conv3_img16_kernel, split partial sums to fp32 slabs) and a BatchNorm launch (slab sum + statistics over the 8 maps + affine): 936
// launches of 6-14 us that sit at the kernel-boundary floor.  Here the BatchNorm of a layer rides in the epilogue of its convolution; what
// the second launch got from the kernel boundary -- the other splits' partial sums, the other maps' statistics -- is exchanged INSIDE the
// launch through epoch-tagged 8-byte granules {epoch, fp32} (the mechanism of the MLP integrator, vs_rollout.hip: relaxed agent-scope
// 8-byte stores / loads, the data is the flag, no fence, bounded spins):
//   1. the tile's partial sums go to LDS as [32 channels][256 pixels]; thread t then works on PIXEL t;
//   2. reduce-scatter over the S input-channel splits of (image, channel tile): split s keeps the channels 32 s / S .. and receives their
//      partial sums from the other S - 1 workgroups (area A: [workgroup][source split][32 / S channels][256 px]), added in split order --
//      the slab order of the two-launch path, so z = round16(sum + bias) is bit-identical to it;
//   3. per-channel statistics of the workgroup's image (sum, centred sum of squares: two LDS rounds), all-gather over the B images of
//      (channel tile, split) through area B ([channel][image][2]), combined with the parallel-variance formula in fp64 and image order;
//   4. affine + activation (+ the block's skip), stores of z, y (and x + r, its 16-bit copy), running estimates by the image-0 workgroups.
// MODE 1 is the backward layer: the convolution is the input-gradient of the FOLLOWING layer (flipped pack), the epilogue the BatchNorm
// backward of this one -- dy' = dy act'(.), (sum dy', sum dy' xhat) over all images by the same all-gather, dz = gamma invstd (dy' - k1 - xhat k2).
// Every workgroup of the launch must be resident (grid <= one per CU: the launcher checks); a partner that does not answer within
// spin_limit polls raises the sticky error word instead of hanging.  epoch = *epoch_base + call_idx: the base is a device word advanced
// once per training step by vs_exchange_epoch_advance (recordable), call_idx the launch's number within the step (a launch argument).
typedef unsigned long long xg64;

struct ImgBnArgs {
    const unsigned short* X;
    const u32x4* Wp;
    int B, Cin, Cout, cs, splits, mtiles;
    xg64* xa;
    xg64* xb;
    const unsigned* epoch_base;
    unsigned call_idx, spin_limit;
    unsigned* xerr;
    unsigned xerr_bit;          // 1 in the workspace's own word, 2 in the process-wide guard word (vs_exchange_guard_set)
    const float* bias;
    const float* gamma;
    const float* beta;
    int act;
    float eps, momentum;
    float* rmean;
    float* rvar;
    float* mean;
    float* invstd;
    unsigned short* z;          // MODE 0: out, MODE 1: in
    void* y;
    int yd;
    const float* skip;
    float* xnew;
    unsigned short* xnew16;
    unsigned short* dz;         // MODE 1: out
    float* dgamma;
    float* dbeta;
    int accumulate;
};

__device__ __forceinline__ xg64 xg_load(const xg64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void xg_store(xg64* p, xg64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ xg64 xg_pack(unsigned epoch, float v) { return ((xg64)epoch << 32) | (xg64)__float_as_uint(v); }
__device__ __forceinline__ float xg_val(xg64 g) { return __uint_as_float((unsigned)g); }

template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// sum over the 64 lanes of a wave, in every lane: four DPP steps inside the 16-lane rows, two cross-row exchanges
__device__ __forceinline__ float wave_sum64(float v) {
    v += dpp_get<0xB1>(v);          // quad_perm [1, 0, 3, 2]
    v += dpp_get<0x4E>(v);          // quad_perm [2, 3, 0, 1]
    v += dpp_get<0x141>(v);         // row_half_mirror
    v += dpp_get<0x140>(v);         // row_mirror
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// -DVS_IMGBN_STAMP (tools/imgbn_timing.py): workgroup 0 leaves wall-clock stamps of its phases in the last 128 bytes of area B
#ifdef VS_IMGBN_STAMP
#define IMGBN_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<unsigned long long*>(p.xa)[-16 + (k)] = wall_clock64(); } while (0)
#else
#define IMGBN_STAMP(k) do { } while (0)
#endif

// ACT (VS_ACT_NONE / VS_ACT_LEAKY: what the ConvResBlock uses) and YF (y in fp32) are compile-time: with the activation and the store type as
// run-time switches every one of the N channel iterations of the store loops carried a ten-way branch -- 0.2 us per channel, 6.5 us of a 15 us launch.
template <int CT, int S, int MODE, int ACT, int YF>
__global__ __launch_bounds__(256, 1) void conv3_img16_bn_kernel(ImgBnArgs p) {
    constexpr int N = 32 / S;                                                    // channels this workgroup owns after the reduce-scatter
    extern __shared__ __attribute__((aligned(16))) unsigned short xs[];          // [cs][18][16], then the partial-sum tile + reduction rows
    int id = blockIdx.x;
    const int me = id;
    const int split = id % S;
    id /= S;
    const int mt = id % p.mtiles, b = id / p.mtiles;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned epoch = *p.epoch_base + p.call_idx;

    IMGBN_STAMP(0);
    float o[2][16];
    img16_tile<CT, 12>(p.X, p.Wp, p.Cin, p.cs, mt, b, split, xs, o);
    IMGBN_STAMP(1);

    // ---- 1. partial sums -> LDS [32 channels][256 pixels] -------------------------------------------------------------------------------
    float* P = reinterpret_cast<float*>(xs);
    float* redA = P + 32 * 256;                                                  // [4 waves][32]
    float* redB = redA + 128;
    float* st = redB + 128;                                                      // [32][2]
    float* par = st + 64;                                                        // [5][32]: bias, gamma, beta, mean, invstd of the owned channels
    __syncthreads();                                                             // everybody has read its last fragments of the image
    // the per-channel parameters of the channels this workgroup will own go to LDS once: read one by one inside the store loops below they
    // cost a global-load latency per channel (the compiler keeps a load behind the stores it may alias): 28 us for a 7 us convolution
    if (tid < N) {
        const int m = mt * 32 + split * N + tid;
        const bool live = m < p.Cout;
        par[tid] = (live && p.bias) ? p.bias[m] : 0.f;
        par[32 + tid] = live ? p.gamma[m] : 0.f;
        par[64 + tid] = live ? p.beta[m] : 0.f;
        if constexpr (MODE == 1) {
            par[96 + tid] = live ? p.mean[m] : 0.f;
            par[128 + tid] = live ? p.invstd[m] : 0.f;
        }
    }
    {
        const int h = lane >> 5, l31 = lane & 31;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) P[((v & 3) + 8 * (v >> 2) + 4 * h) * 256 + wave * 64 + j * 32 + l31] = o[j][v];
    }
    __syncthreads();

    // ---- 2. reduce-scatter over the splits ----------------------------------------------------------------------------------------------
    float R[N];
    bool timed_out = false;
    if constexpr (S == 1) {
#pragma unroll
        for (int i = 0; i < N; ++i) R[i] = P[i * 256 + tid];
    } else {
        const int wg0 = me - split;                                              // workgroup of split 0 of this (image, channel tile)
#pragma unroll
        for (int cc = 0; cc < 32; ++cc) {
            const int owner = cc / N;
            if (owner != split) xg_store(p.xa + (((int64_t)(wg0 + owner) * S + split) * N + (cc % N)) * 256 + tid, xg_pack(epoch, P[cc * 256 + tid]));
        }
        xg64 got[(S - 1) * N];
        const xg64* mine = p.xa + ((int64_t)me * S) * N * 256 + tid;
#pragma unroll
        for (int k = 0; k < S - 1; ++k) {
            const int src = k < split ? k : k + 1;
#pragma unroll
            for (int i = 0; i < N; ++i) got[k * N + i] = xg_load(mine + ((int64_t)src * N + i) * 256);
        }
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int k = 0; k < S - 1; ++k) {
                const int src = k < split ? k : k + 1;
#pragma unroll
                for (int i = 0; i < N; ++i)
                    if ((unsigned)(got[k * N + i] >> 32) != epoch) {
                        got[k * N + i] = xg_load(mine + ((int64_t)src * N + i) * 256);
                        ok = false;
                    }
            }
            if (ok) break;
            if (++spins > p.spin_limit) { timed_out = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float own = P[(split * N + i) * 256 + tid];
            float acc = 0.f;
#pragma unroll
            for (int src = 0; src < S; ++src) {                                  // split order = the slab order of the two-launch path
                // got[] holds the sources in order without this split: source src sits at row src (src < split) or src - 1 (src > split);
                // both candidates are named with compile-time indices (a run-time index would send the array to scratch)
                const float lo = xg_val(got[(src < S - 1 ? src : 0) * N + i]), hi = xg_val(got[(src > 0 ? src - 1 : 0) * N + i]);
                const float v = src == split ? own : (src < split ? lo : hi);
                acc = src == 0 ? v : acc + v;
            }
            R[i] = acc;
        }
    }

    IMGBN_STAMP(2);
    // ---- 3. / 4. the BatchNorm of the layer over the B images ---------------------------------------------------------------------------
    const int m0 = mt * 32 + split * N;                                          // first channel this workgroup owns
    const int64_t pix0 = ((int64_t)b * p.Cout + m0) * 256 + tid;
    float e1[N], e2[N];                                                          // MODE 0: (z, -)   MODE 1: (dy', xhat)
    // The two per-channel sums over the image's 256 pixels: the values go to LDS pixel-major [channel][256 + T] and T = 256 / N threads per
    // channel add N pixels each (stride T: conflict-free with that pitch), then meet by 3 .. 6 butterfly steps inside their T lanes.  (A first
    // form reduced every channel across the wave from the pixel-major registers: 2 N dependent cross-lane reductions per thread, 0.4 us per
    // channel -- 18 us behind a 7 us convolution.)
    constexpr int T = 256 / N, ZP = 256 + T;
    float* Z1 = st + 64 + 160;                                                   // [N][ZP]
    float* Z2 = Z1 + N * ZP;
    const int zc = tid / T, zpart = tid % T;                                     // the channel / pixel residue this thread sums
    auto group_sum = [&](float v) {                                              // sum over the T lanes of the channel, in every one of them
        v += dpp_get<0xB1>(v);          // quad_perm [1, 0, 3, 2]
        v += dpp_get<0x4E>(v);          // quad_perm [2, 3, 0, 1]
        v += dpp_get<0x141>(v);         // row_half_mirror: 8 lanes
        if constexpr (T >= 16) v += dpp_get<0x140>(v);                           // row_mirror: 16 lanes
        if constexpr (T >= 64) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); }
        return v;
    };
    float ga = 0.f, gq = 0.f;                                                    // the channel's two numbers for this image (in its T threads)
    if constexpr (MODE == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const bool live = m0 + i < p.Cout;
            const unsigned short zb = vs_f2h(R[i] + par[i], CT);
            e1[i] = vs_h2f(zb, CT);
            if (live) p.z[pix0 + (int64_t)i * 256] = zb;
            Z1[i * ZP + tid] = e1[i];
        }
        __syncthreads();
        float zv[N];
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < N; ++k) { zv[k] = Z1[zc * ZP + zpart + T * k]; a += zv[k]; }
        ga = group_sum(a);                                                       // sum over this image's 256 pixels
        const float mb = ga * (1.f / 256.f);
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < N; ++k) { const float d = zv[k] - mb; q += d * d; }
        gq = group_sum(q);                                                       // centred sum of squares
    } else {
        // (32-bit destinations: two 16-bit loads packed into one register are a d16 / d16_hi pair, the second of which MERGES into the
        //  register the first one is still writing -- the compiler then waits for every load: 32 serial HBM round trips, 15 us)
        unsigned zin[N];
#pragma unroll
        for (int i = 0; i < N; ++i) zin[i] = m0 + i < p.Cout ? (unsigned)p.z[pix0 + (int64_t)i * 256] : 0u;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const bool live = m0 + i < p.Cout;
            const float xv = vs_h2f((unsigned short)zin[i], CT);
            const float xh = (xv - par[96 + i]) * par[128 + i];
            const float pre = xh * par[32 + i] + par[64 + i];
            const float dzp = live ? R[i] * (ACT == VS_ACT_LEAKY ? (pre > 0.f ? 1.f : 0.2f) : 1.f) : 0.f;
            e1[i] = dzp;
            e2[i] = xh;
            Z1[i * ZP + tid] = dzp;
            Z2[i * ZP + tid] = dzp * xh;
        }
        __syncthreads();
        float a = 0.f, q = 0.f;
#pragma unroll
        for (int k = 0; k < N; ++k) { a += Z1[zc * ZP + zpart + T * k]; q += Z2[zc * ZP + zpart + T * k]; }
        ga = group_sum(a);
        gq = group_sum(q);
    }
    // all-gather over the images: thread i < N publishes its channel's two numbers for image b; then thread (i, bb) fetches image bb's pair of
    // channel i -- ONE round trip for the whole workgroup (a first form in which thread i polled the B images one after the other cost 2 B
    // dependent round trips per layer: SST 20.0 -> 24.4 ms) -- and parks it in LDS, where thread i combines them in image order
    IMGBN_STAMP(3);
    float* gat = P;                                                              // [B][N][2]: the partial-sum tile is no longer needed
    if (zpart == 0) {                                                            // the first of the channel's T threads publishes
        xg64* row = p.xb + ((int64_t)(m0 + zc) * p.B) * 2;
        xg_store(row + 2 * b, xg_pack(epoch, ga));
        xg_store(row + 2 * b + 1, xg_pack(epoch, gq));
    }
    {
        constexpr int PER = 256 / N;                                             // images fetched per round
        const int i = tid % N;
        const xg64* row = p.xb + ((int64_t)(m0 + i) * p.B) * 2;
        for (int base = 0; base < p.B; base += PER) {
            const int bb = base + tid / N;
            if (bb < p.B) {
                xg64 g = xg_load(row + 2 * bb), g2 = xg_load(row + 2 * bb + 1);
                unsigned spins = 0;
                while ((unsigned)(g2 >> 32) != epoch || (unsigned)(g >> 32) != epoch) {
                    if (++spins > p.spin_limit) { timed_out = true; break; }
                    __builtin_amdgcn_s_sleep(2);
                    g = xg_load(row + 2 * bb);
                    g2 = xg_load(row + 2 * bb + 1);
                }
                gat[(bb * N + i) * 2] = xg_val(g);
                gat[(bb * N + i) * 2 + 1] = xg_val(g2);
            }
        }
    }
    __syncthreads();
    IMGBN_STAMP(4);
    if (tid < N) {
        const int i = tid;
        if constexpr (MODE == 0) {
            // first the mean over all images, then the centred sums: M2 = sum_b [M2_b + 256 (mean_b - mean)^2]
            double tot = 0.0;
            for (int bb = 0; bb < p.B; ++bb) tot += (double)gat[(bb * N + i) * 2];
            const double n = 256.0 * (double)p.B, mu = tot / n;
            double m2 = 0.0;
            for (int bb = 0; bb < p.B; ++bb) {
                const double d = (double)gat[(bb * N + i) * 2] * (1.0 / 256.0) - mu;
                m2 += (double)gat[(bb * N + i) * 2 + 1] + 256.0 * d * d;
            }
            const double var = m2 / n;
            const float muf = (float)mu, is = (float)(1.0 / sqrt(var + (double)p.eps));
            st[2 * i] = muf;
            st[2 * i + 1] = is;
            if (b == 0 && m0 + i < p.Cout) {
                p.mean[m0 + i] = muf;
                p.invstd[m0 + i] = is;
                if (p.rmean) {
                    const double ub = n > 1.0 ? m2 / (n - 1.0) : var;
                    p.rmean[m0 + i] = (float)((1.0 - p.momentum) * (double)p.rmean[m0 + i] + p.momentum * (double)muf);
                    p.rvar[m0 + i] = (float)((1.0 - p.momentum) * (double)p.rvar[m0 + i] + p.momentum * (double)(float)ub);
                }
            }
        } else {
            double t1 = 0.0, t2 = 0.0;
            for (int bb = 0; bb < p.B; ++bb) {
                t1 += (double)gat[(bb * N + i) * 2];
                t2 += (double)gat[(bb * N + i) * 2 + 1];
            }
            const float inv_n = 1.f / (256.f * (float)p.B);
            st[2 * i] = (float)t1 * inv_n;
            st[2 * i + 1] = (float)t2 * inv_n;
            if (b == 0 && m0 + i < p.Cout) {
                if (p.accumulate) { p.dbeta[m0 + i] += (float)t1; p.dgamma[m0 + i] += (float)t2; }
                else { p.dbeta[m0 + i] = (float)t1; p.dgamma[m0 + i] = (float)t2; }
            }
        }
    }
    __syncthreads();
    IMGBN_STAMP(5);
    if (timed_out) atomicOr(p.xerr, p.xerr_bit);

    if constexpr (MODE == 0) {
        float sk[N];
        if (p.skip) {
#pragma unroll
            for (int i = 0; i < N; ++i) sk[i] = m0 + i < p.Cout ? p.skip[pix0 + (int64_t)i * 256] : 0.f;               // (in flight together, ahead of the stores)
        }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (m0 + i >= p.Cout) continue;
            const int64_t idx = pix0 + (int64_t)i * 256;
            const float pre = (e1[i] - st[2 * i]) * st[2 * i + 1] * par[32 + i] + par[64 + i];
            const float yv = ACT == VS_ACT_LEAKY ? (pre > 0.f ? pre : 0.2f * pre) : pre;
            if constexpr (YF) reinterpret_cast<float*>(p.y)[idx] = yv;
            else reinterpret_cast<unsigned short*>(p.y)[idx] = vs_f2h(yv, CT);
            if (p.skip) {                                                        // block tail (resnet.py:66-70): x + r in fp32 and as the next operand
                const float xn = sk[i] + yv;
                p.xnew[idx] = xn;
                if (p.xnew16) p.xnew16[idx] = vs_f2h(xn, CT);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (m0 + i >= p.Cout) continue;
            p.dz[pix0 + (int64_t)i * 256] = vs_f2h(par[32 + i] * par[128 + i] * (e1[i] - st[2 * i] - e2[i] * st[2 * i + 1]), CT);
        }
    }
    IMGBN_STAMP(6);
}

// base += 65536: one training step's worth of launch numbers (the caller's call_idx restarts at 1)
__global__ void exchange_epoch_advance_kernel(unsigned* base) { *base += 65536u; }

// ---- many maps of 16 / 32 / 64 pixels width: the same contraction on row BANDS -------------------------------------------------
// Every 3x3 block of the SST / VGG encoders and decoders (conv.py:127-171, 267-426 of the reference) on hundreds of maps: as "column
// matrix + GEMM" the gather writes and the GEMM re-reads 9 x the input (64 -> 64 channels on 352 maps of 64 x 64: 1.7 GB each way, 0.8 +
// 0.5 ms for 0.1 TFLOP).  Here a workgroup owns 256 consecutive pixels of one map (R = 256 / W whole rows) and 32 output channels:
//   * the band's R + 2 input rows of 64 channels at a time go to LDS by LDS-DMA (global_load_lds_dwordx4; the [channel][row][pixel]
//     image is lane-linear in the 16-byte piece index, rows outside the map come from a block of zeros), double buffered: the DMA of
//     channels 64 (p + 1) .. is in flight while channels 64 p .. are multiplied.  It is covered by the same counted wait as the weight
//     stream: it is older than the 33 fragment loads the last group of a phase leaves outstanding.
//   * weights, fragment stream and the unshifted pixel operand exactly as above (same pre-pack); wave w owns pixels 64 w .. 64 w + 63.
//   * x shift on the result: a column tile is 32 consecutive pixels of a row, so the neighbour of its first / last lane is the last /
//     first lane of the wave's other tile (W = 64) or the zero padding: one ds_bpermute rotation per tile and side, a select per value.
//   * output in the caller's type with the bias, straight from the registers (128-byte runs along the pixels).
//   * two forms: AHEAD = 12 fragment groups in flight and one workgroup per CU (the registers of 36 fragments), or AHEAD = 4 and TWO
//     workgroups per CU where their LDS fits (one 64-channel phase, or W = 16): a workgroup's launch, first DMA and epilogue are then
//     covered by its neighbour's MFMA phase instead of by a deep prefetch.
// K4 = 1: the k4 s2 p1 convolution family on the four parity planes of its large operand (csrc/vs_conv_k4s2.hip): X holds [4 planes][K]
// channels (plane = row parity * 2 + column parity, K a multiple of 64, so a 64-channel phase lies in ONE plane) and a plane sees only 2 x 2
// of the 3 x 3 taps -- rows {-1, 0} and columns {-1, 0} for the odd planes, {0, +1} for the even ones.  The fragment stream then holds
// 4 chunks x 2 tap rows x 2 tap columns per phase (vs_conv_k4s2_pack_weight, skip form) and the zero taps cost neither fragment loads nor
// MFMAs: 32 instead of 72 MFMAs per phase and wave.
template <int CT, int W, int AHEAD, int MINB, int K4 = 0>
__global__ __launch_bounds__(256, MINB) void conv3_band_kernel(const unsigned short* __restrict__ X, const u32x4* __restrict__ Wp, const float* __restrict__ bias,
                                                         void* __restrict__ Y, int yd, int B, int Cin, int H, int Cout, int mtiles, int bands,
                                                         int band_xcd_remap, double* __restrict__ bn_sums, int maps_per_group, int Creal) {
    // Creal = the channels X really has (its per-map stride); the channels Creal .. Cin - 1 of the last phase are staged as zeros
    constexpr int NKY = K4 ? 2 : 3, NF = K4 ? 2 : 3, GPP = 4 * NKY;              // tap rows and fragments per group, groups per 64-channel phase
    static_assert(GPP % AHEAD == 0, "the fragment ring must divide the groups of a phase");
    // W >= 16: the tile is R = 256 / W rows of ONE map (`band` = which rows); W = 8: FOUR whole 8 x 8 maps (`band` = which four, b = 0);
    // W = 4: SIXTEEN whole 4 x 4 maps.  A 16-byte piece then holds two rows, so a map is laid out as [2 zero rows][4 rows] (three pieces):
    // the row above a map is its own zero piece, the row below it the zero piece of the NEXT map (one more zero piece closes each buffer)
    constexpr int IPB = W == 8 ? 4 : (W == 4 ? 16 : 1), RI = W == 8 ? 8 : (W == 4 ? 4 : 256 / W), RPI = RI + 2, RP = IPB * RPI;
    constexpr int ROW0 = W == 4 ? 1 : 0;                                         // rows between the start of a map's block and its row -1
    constexpr int NP = 64 * RP * W / 8 / 256;                                    // 16-byte pieces per thread and phase: 12 / 12 / 10 / 9 / 10
    constexpr int BUF = 64 * RP * W + (W == 4 ? 8 : 0);                          // elements per buffer
    extern __shared__ __attribute__((aligned(16))) unsigned short xs[];          // [2][64 channels][IPB maps][RPI rows][W]
    int id = blockIdx.x;
    if (band_xcd_remap) {
        // the output-channel tiles of one band stage the SAME input rows: blocks are dealt round-robin over the 8 XCDs (observed, speed
        // only), so give every XCD a contiguous range of logical ids (bijective for any grid) and the tiles of a band meet in one L2
        const int nwg = (int)gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = id & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    }
    const int mt = id % mtiles;
    id /= mtiles;
    const int band = id % bands, b = id / bands;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nph = Cin >> 6;                                                    // (Cin = the channel count padded to whole 64-channel phases)

    auto dma = [&](int ph) {
        char* dst = reinterpret_cast<char*>(xs + (ph & 1) * BUF);
#pragma unroll
        for (int r = 0; r < NP; ++r) {
            const void* g;
            if constexpr (W == 4) {
                const int u = r * 256 + tid, pp = u % 3, mp = (u / 3) % IPB, cl = u / (3 * IPB);     // piece 0 of a map: its two zero rows
                const int img = band * IPB + mp;
                g = (pp != 0 && img < B && ph * 64 + cl < Creal) ? (const void*)(X + (((int64_t)img * Creal + ph * 64 + cl) * 16 + (pp - 1) * 8))
                                                               : (const void*)vs_glds_zero;
            } else {
                constexpr int PW = W / 8;
                const int u = r * 256 + tid, pc = u % PW, rr = (u / PW) % RP, cl = u / (PW * RP);
                const int y = (IPB == 1 ? band * RI : 0) + rr % RPI - 1;
                const int img = IPB == 1 ? b : band * IPB + rr / RPI;
                g = (y >= 0 && y < H && img < B && ph * 64 + cl < Creal) ? (const void*)(X + ((((int64_t)img * Creal + ph * 64 + cl) * H + y) * W + pc * 8))
                                                                       : (const void*)vs_glds_zero;
            }
            // asm: the compiler orders a builtin LDS-DMA against every later LDS read with s_waitcnt vmcnt(0), which would also drain the
            // weight stream at every phase; the DMA is covered by the counted waits below instead (M0 has this one writer)
            const uint32_t d = (uint32_t)(uintptr_t)(dst + (r * 256 + wave * 64) * 16);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(d), "v"(g) : "memory", "m0");
        }
    };
    if constexpr (W == 4) {
        if (tid < (nph > 1 ? 2 : 1)) *reinterpret_cast<u32x4*>(xs + tid * BUF + 64 * RP * W) = u32x4{0u, 0u, 0u, 0u};     // the closing zero piece of each buffer
    }
    dma(0);

    const int chunks_total = Cin >> 4;
    const char* wbase = reinterpret_cast<const char*>(Wp + ((int64_t)mt * chunks_total * NKY) * (NF * 64));
    const unsigned voff = lane * 16;
    u32x4 a[AHEAD][3];                                                            // (K4: the third slot is unused)
#pragma unroll
    for (int gg = 0; gg < AHEAD; ++gg) {
        if constexpr (K4) img_load2(a[gg][0], a[gg][1], wbase + gg * (NF * 1024), voff);
        else img_load3(a[gg][0], a[gg][1], a[gg][2], wbase + gg * 3072, voff);
    }

    f32x16 acc[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[kx][j][v] = 0.f;

    const int li = lane & 15, q = li >> 2, p = li & 3, cb = (lane >> 4) & 1, h = lane >> 5;
    int lofs[2];                                                                  // element offset of this lane's piece of tile j, k-row 0, tap row 0
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q0 = wave * 64 + j * 32 + 16 * cb + 4 * p, qi = q0 % (RI * W);
        lofs[j] = ((8 * h + q) * RP + (q0 / (RI * W)) * RPI + qi / W + ROW0) * W + qi % W;
    }
    const int ngroups = chunks_total * NKY;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                             // phase 0 (and the first fragments) have landed
    __builtin_amdgcn_s_barrier();

    for (int ph = 0; ph < nph; ++ph) {
        if (ph + 1 < nph) dma(ph + 1);
        const unsigned short* buf = xs + (ph & 1) * BUF;
        const int g0 = ph * GPP;
        // K4: the plane of this phase (wave-uniform): odd rows (plane >> 1) see tap rows {0, 1}, even rows {1, 2}; likewise the columns
        const int plane = K4 ? ph / (nph >> 2) : 0;
        const int ky0 = K4 ? ((plane >> 1) ? 0 : 1) : 0;
        const bool odd_cols = K4 && (plane & 1);
#pragma unroll
        for (int gg = 0; gg < GPP; ++gg) {
            const int ch = gg / NKY, ky = ky0 + gg % NKY;
            u32x4 bf[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = vs_tr16_pair(buf + lofs[j] + (ch * 16 * RP + ky) * W, 4 * RP * W);
            const int sl = gg % AHEAD;                                           // register set of this group (GPP % AHEAD == 0; a constant once unrolled)
            if constexpr (K4) {
                img_wait2n<(AHEAD - 1) * 2>(a[sl][0], a[sl][1]);
                if (odd_cols) {                                                  // column taps {0, 1}: x[col - 1], x[col]
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[0][j] = mfma16_32<CT>(a[sl][0], bf[j], acc[0][j]);
                        acc[1][j] = mfma16_32<CT>(a[sl][1], bf[j], acc[1][j]);
                    }
                } else {                                                         // column taps {1, 2}: x[col], x[col + 1]
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[1][j] = mfma16_32<CT>(a[sl][0], bf[j], acc[1][j]);
                        acc[2][j] = mfma16_32<CT>(a[sl][1], bf[j], acc[2][j]);
                    }
                }
            } else {
                img_wait3n<(AHEAD - 1) * 3>(a[sl][0], a[sl][1], a[sl][2]);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[kx][j] = mfma16_32<CT>(a[sl][kx], bf[j], acc[kx][j]);
            }
            int gn = g0 + AHEAD + gg;
            if (gn > ngroups - 1) gn = ngroups - 1;
            if constexpr (K4) img_load2(a[sl][0], a[sl][1], wbase + (int64_t)gn * (NF * 1024), voff);
            else img_load3(a[sl][0], a[sl][1], a[sl][2], wbase + (int64_t)gn * 3072, voff);
        }
        // the last wait of the phase left <= (AHEAD - 1) * 3 loads outstanding, all younger than the DMA of the next phase: it has landed for this
        // wave; behind the barrier for all of them, and nobody reads this phase's buffer any more
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    // the repeated fragment loads of the last groups are still in flight and their destination registers are dead to the compiler:
    // they must land before it re-uses those registers (for the store addresses below)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- out[x] = G_1[x] + G_0[x - 1] + G_2[x + 1] inside the image row, + bias, typed store -------------------------------------------
    const int l31 = lane & 31;
    const int idx_l = ((lane & 32) | ((lane - 1) & 31)) * 4, idx_r = ((lane & 32) | ((lane + 1) & 31)) * 4;
    const int mrow = mt * 32 + 4 * (lane >> 5);
    // W = 8: wave w owns map w of the four; W = 4: maps 4 w .. 4 w + 3 of the sixteen -- tile j holds two of them, lanes 0-15 / 16-31 one each
    const int oimg = IPB == 1 ? b : (W == 4 ? band * IPB + wave * 4 + (l31 >> 4) : band * IPB + wave);
    const int64_t obase = W == 4 ? (int64_t)oimg * Cout * 16 + (l31 & 15)
                                 : (int64_t)oimg * Cout * H * W + (IPB == 1 ? band * RI * W + wave * 64 : 0) + l31;
    const int col0 = (wave * 64 + l31) % W, col1 = (wave * 64 + 32 + l31) % W;
    float bvs[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int m = mrow + (v & 3) + 8 * (v >> 2);
        bvs[v] = (bias && m < Cout) ? bias[m] : 0.f;
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int m = mrow + (v & 3) + 8 * (v >> 2);
        float rl[2], rr[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float gl = acc[0][j][v], gr = acc[2][j][v];          // (a bit_cast applied to the vector element itself read element 0)
            rl[j] = lane_gather(idx_l, gl);
            rr[j] = lane_gather(idx_r, gr);
        }
        const float left0 = (l31 != 0 && col0 != 0) ? rl[0] : 0.f;
        const float left1 = col1 == 0 ? 0.f : (l31 != 0 ? rl[1] : rl[0]);
        const float right0 = col0 == W - 1 ? 0.f : (l31 != 31 ? rr[0] : rr[1]);
        const float right1 = (l31 != 31 && col1 != W - 1) ? rr[1] : 0.f;
        if constexpr (W == 4) {
            const float bv = bvs[v];
            const int64_t o = obase + (int64_t)m * 16;
            if (m < Cout && oimg < B) vs_st(Y, yd, o, acc[1][0][v] + left0 + right0 + bv);
            if (m < Cout && oimg + 2 < B) vs_st(Y, yd, o + (int64_t)2 * Cout * 16, acc[1][1][v] + left1 + right1 + bv);
        } else {
            const float bv = bvs[v];
            const float o0 = acc[1][0][v] + left0 + right0 + bv, o1 = acc[1][1][v] + left1 + right1 + bv;
            const bool live = m < Cout && oimg < B;
            if (live) {
                const int64_t o = obase + (int64_t)m * H * W;
                vs_st(Y, yd, o, o0);
                vs_st(Y, yd, o + 32, o1);
            }
            if (bn_sums) {
                // sum and sum of squares of the STORED values (rounded as stored: what a statistics pass over y would read) of this wave's 64
                // pixels of channel m -> the (call group, channel) slot the following BatchNorm reads: fp32 over the 32 lanes, fp64 atomics
                // across waves and workgroups (all pixels of a wave lie in one map here: W >= 8)
                const float r0 = yd == VS_F32 ? o0 : vs_h2f(vs_f2h(o0, yd), yd), r1 = yd == VS_F32 ? o1 : vs_h2f(vs_f2h(o1, yd), yd);
                float s1 = live ? r0 + r1 : 0.f, s2 = live ? r0 * r0 + r1 * r1 : 0.f;
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
                if (maps_per_group > 0) {
                    if (l31 == 0 && live) {
                        double* slot = bn_sums + ((int64_t)(oimg / maps_per_group) * Cout + m) * 2;
                        atomicAdd(slot, (double)s1);
                        atomicAdd(slot + 1, (double)s2);
                    }
                } else if (IPB == 1) {
                    // partial-table form (maps_per_group == 0, bn_sums = float parts[B * bands][Cout][2]): no atomics -- the four waves' sums of
                    // this channel meet in LDS (free by now: everybody is past the last phase's barrier) and ONE thread per channel writes the
                    // workgroup's (sum, sum of squares) to its own row of the table; vs_bn_stats_from_parts_fold adds the rows in a fixed order
                    if (l31 == 0) {
                        float* red = reinterpret_cast<float*>(xs);
                        const int cl = 4 * (lane >> 5) + (v & 3) + 8 * (v >> 2);
                        red[(wave * 32 + cl) * 2] = s1;
                        red[(wave * 32 + cl) * 2 + 1] = s2;
                    }
                } else if (l31 == 0 && live) {                                       // W = 8: a wave's 64 pixels ARE one map -> row = the map
                    float* slot = reinterpret_cast<float*>(bn_sums) + ((int64_t)oimg * Cout + m) * 2;
                    slot[0] = s1;
                    slot[1] = s2;
                }
            }
        }
    }
    if constexpr (W != 4 && IPB == 1) {
        if (bn_sums && maps_per_group == 0) {
            __syncthreads();
            if (tid < 32) {
                const float* red = reinterpret_cast<const float*>(xs);
                const int m = mt * 32 + tid;
                const float s1 = (red[tid * 2] + red[(32 + tid) * 2]) + (red[(64 + tid) * 2] + red[(96 + tid) * 2]);
                const float s2 = (red[tid * 2 + 1] + red[(32 + tid) * 2 + 1]) + (red[(64 + tid) * 2 + 1] + red[(96 + tid) * 2 + 1]);
                if (m < Cout) {
                    float* slot = reinterpret_cast<float*>(bn_sums) + (((int64_t)b * bands + band) * Cout + m) * 2;
                    slot[0] = s1;
                    slot[1] = s2;
                }
            }
        }
    }
}

// ---- weight gradient of the same convolution without a column matrix -------------------------------------------------------------------
// dW[m][c][ky][kx] = sum over maps and pixels of dz[m][y][x] * x[c][y + ky - 1][x + kx - 1]: the contraction runs over PIXELS, so the tap
// shift sits on the contraction axis and cannot be taken on the result.  Workgroup = (32 output channels, 32 input channels, share ks of
// the row bands of the batch); per band (256 consecutive pixels of one map = 256 / W rows) it stages
//     dz  [32 m][256 px]                          rows of 264 elements (odd multiple of 16 bytes: conflict-free ds_read_b128 over rows)
//     x   [3 column shifts][32 c][R + 2 rows][W]  channel pitch an odd multiple of 16 bytes; copy kx holds x[..][col + kx - 1], zero beyond
// through registers (the +-1 pixel copies are built with v_alignbyte from the piece and one neighbour pixel each side), so that EVERY
// fragment read is an aligned ds_read_b128 (an unaligned one is replayed at 64 cycles).  The band's 16 k-steps of 16 pixels are split
// over the four waves; a k-step is one dz fragment and nine x fragments (copy kx, row + ky) -> nine MFMAs into nine 32x32 accumulators.
// The next band's global loads are issued before the MFMA phase of the current one.  Each wave leaves its nine accumulators in an fp32
// slab ([ksplit x 4 / MW][9][Cout][Cin]); vs_conv3_wgrad_band_finish adds the slabs (and the pending gradient, if any) into [Cout][Cin][3][3].
// MW = 32-row output-channel sub-tiles per workgroup (4, 2 or 1): wave w owns sub-tile w % MW and the k-steps of part w / MW of 4 / MW.
// With MW = 4 every wave walks all 16 k-steps of a band for its own 32 rows (144 MFMAs per staged band, one slab per workgroup share).
// The batch may come in up to 64 equal pieces (the remembered (dz, x) pairs of a convolution that is applied once per predicted frame:
// functional.flush_deferred_wgrads) -- map b lives in piece b / maps_per_piece; the pointers travel as kernel arguments.
constexpr int WG_MAX_PIECES = 64;
struct WgradPieces {
    const unsigned short* x[WG_MAX_PIECES];
    const unsigned short* dz[WG_MAX_PIECES];
    int maps_per_piece;
};

// K4 = 1: the weight gradient of a k4 s2 p1 (transposed) convolution on the parity planes of its large operand (csrc/vs_conv_k4s2.hip): Cin = 4 K
// plane channels, K a multiple of 32, so the 32 channels of a workgroup lie in ONE plane, which sees 2 x 2 of the 3 x 3 taps: the other five
// MFMAs per k-step (and their slab stores: vs_conv_k4s2_wgrad_finish never reads them) are skipped.
template <int W>
constexpr int wgrad_cpitch() { return W == 4 ? 388 : (W == 8 ? 4 * 10 : 256 / W + 2) * W + 8; }

// CTW (K4 form only) = 32-channel tiles of ONE plane a workgroup walks per item, one after the other against the SAME staged dz tile: the dz
// tile [32 MW][256 px] is what every channel tile of a layer re-reads (L2 -> LDS: 1.07 GB per launch on the Moving-MNIST 8 x 8 <-> 4 x 4 layers,
// 9 TB/s -- the kernel was L2-bound), and with four accumulators per tile instead of nine two tiles fit the register file.
template <int CT, int W, int MW, int K4 = 0, int CTW = 1>
__global__ __launch_bounds__(256) void wgrad3_band_kernel(WgradPieces pieces, float* __restrict__ slabs, int B, int Cin, int H, int Cout, int ctiles, int ksplit) {
    // W >= 16: an item is a band of R = 256 / W rows of one map; W = 8: FOUR whole 8 x 8 maps (rows of one 16-byte piece, no column neighbours);
    // W = 4: SIXTEEN whole 4 x 4 maps (the 512-channel layers of the VGG encoders / decoders, the parity planes of 8 x 8 maps): a 16-byte
    // piece is TWO rows, a k-step of 16 pixels is one map; a fragment (8 pixels = two rows, shifted by ky rows = 8 bytes) is read as two
    // aligned ds_read_b64, the channel pitch of 388 elements (194 words = 2 mod 64) keeps the 32 rows of such a read on 64 different banks;
    // the zero rows above / below every map are written once at kernel start
    constexpr int IPB = W == 8 ? 4 : (W == 4 ? 16 : 1), R = W == 8 ? 8 : (W == 4 ? 4 : 256 / W), RPI = R + 2, RP = IPB * RPI, PW = W >= 8 ? W / 8 : 1;
    constexpr int PPM = R * W / 8;                                               // 16-byte pieces of one map of a multi-map item
    constexpr int CPITCH = wgrad_cpitch<W>();                                    // 392 / 328 / 296 / 328 elements: (CPITCH / 8) odd; W = 4: 388
    constexpr int ZPITCH = 264;
    constexpr int NX = W == 4 ? 32 * IPB * PPM : 32 * RP * PW;                   // 16-byte pieces of the x tile: 1536 / 1280 / 1152; W = 4: 1024
    constexpr int NXT = (NX + 255) / 256;                                        // per thread: 6 / 5 / 5 (the last partly) / 4
    extern __shared__ __attribute__((aligned(16))) unsigned short xs[];          // [3][32][CPITCH] then dz [32 MW][ZPITCH]
    constexpr int KW = 4 / MW, KSTEPS = 16 / KW;                                 // k parts per band, k-steps per wave and band
    unsigned short* zs = xs + 3 * 32 * CPITCH;
    int id = blockIdx.x;
    const int ks = id % ksplit;
    id /= ksplit;
    const int ct0 = (id % ctiles) * CTW, mt = id / ctiles;                          // ctiles = groups of CTW channel tiles
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bands = H / R;
    const int64_t items = IPB == 1 ? (int64_t)B * bands : (int64_t)((B + IPB - 1) / IPB);

    static_assert(CTW == 1 || K4, "several channel tiles per workgroup: K4 form only (four accumulators per tile)");
    constexpr int NT = K4 ? 4 : 9;                                               // accumulators per channel tile: the taps a plane sees / all nine
    f32x16 acc[CTW][NT];
#pragma unroll
    for (int s2 = 0; s2 < CTW; ++s2)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[s2][t][v] = 0.f;

    u32x4 xr[NXT], zr[4 * MW];
    unsigned xe[NXT];                                                            // neighbour pixels: low half = the one before the piece, high half = after
    auto load_x = [&](int64_t it, int sub) {
        const int ct = ct0 + sub;
        const int bg = IPB == 1 ? (int)(it / bands) : (int)it * IPB, band = IPB == 1 ? (int)(it % bands) : 0;       // first map of the item
        const int piece = bg / pieces.maps_per_piece, b = bg - piece * pieces.maps_per_piece;                             // (an item never straddles pieces)
        const unsigned short* X = pieces.x[piece];
        if constexpr (W == 4) {
#pragma unroll
            for (int r = 0; r < NXT; ++r) {
                const int u = r * 256 + tid, hf = u & 1, im = (u >> 1) & 15, c = ct * 32 + (u >> 5);
                const bool ok = c < Cin && bg + im < B;
                xr[r] = *reinterpret_cast<const u32x4*>(ok ? X + ((int64_t)(b + im) * Cin + c) * 16 + hf * 8 : reinterpret_cast<const unsigned short*>(vs_glds_zero));
            }
        } else
#pragma unroll
        for (int r = 0; r < NXT; ++r) {
            const int u = r * 256 + tid, pc = u % PW, rr = (u / PW) % RP, cl = u / (PW * RP);
            const int im = rr / RPI, y = band * R + rr % RPI - 1, c = ct * 32 + cl;
            // every load is unconditional and the SELECT sits on the address (a block of zeros for what lies outside): a conditional load
            // makes the compiler wait for each one where the two paths merge -- ten serial round trips per band
            const bool ok = u < NX && y >= 0 && y < H && c < Cin && bg + im < B;
            const unsigned short* zero = reinterpret_cast<const unsigned short*>(vs_glds_zero);
            const unsigned short* src = ok ? X + (((int64_t)(b + im) * Cin + c) * H + y) * W + pc * 8 : zero;
            const unsigned short* pb = (ok && pc > 0) ? src - 1 : zero;
            const unsigned short* pa = (ok && pc + 1 < PW) ? src + 8 : zero;
            xr[r] = *reinterpret_cast<const u32x4*>(src);
            xe[r] = (unsigned)*pb | ((unsigned)*pa << 16);
        }
    };
    auto load_z = [&](int64_t it) {
        const int bg = IPB == 1 ? (int)(it / bands) : (int)it * IPB, band = IPB == 1 ? (int)(it % bands) : 0;
        const int piece = bg / pieces.maps_per_piece, b = bg - piece * pieces.maps_per_piece;
        const unsigned short* DZ = pieces.dz[piece];
#pragma unroll
        for (int r = 0; r < 4 * MW; ++r) {
            const int u = r * 256 + tid, ml = u >> 5, pc = u & 31, m = mt * (32 * MW) + ml;
            const int im = IPB == 1 ? 0 : pc / PPM, pcm = IPB == 1 ? pc : pc % PPM;                                    // map of the item, piece inside it
            const unsigned short* src = (m < Cout && bg + im < B) ? DZ + (((int64_t)(b + im) * Cout + m) * H + band * R) * W + pcm * 8
                                                                  : reinterpret_cast<const unsigned short*>(vs_glds_zero);
            zr[r] = *reinterpret_cast<const u32x4*>(src);
        }
    };
    auto store_x = [&]() {
        if constexpr (W == 4) {
#pragma unroll
            for (int r = 0; r < NXT; ++r) {
                const int u = r * 256 + tid, hf = u & 1, im = (u >> 1) & 15, cl = u >> 5;
                unsigned short* dst = xs + cl * CPITCH + im * (RPI * 4) + 4 + hf * 8;      // rows 1 + 2 hf, 2 + 2 hf of the map's six
                const u32x4 d = xr[r];                                                   // (d0 d1) = one row of four pixels, (d2 d3) the next
                u32x2 a, bb;
                a[0] = __builtin_amdgcn_alignbyte(d[0], 0u, 2);  a[1] = __builtin_amdgcn_alignbyte(d[1], d[0], 2);
                bb[0] = __builtin_amdgcn_alignbyte(d[2], 0u, 2); bb[1] = __builtin_amdgcn_alignbyte(d[3], d[2], 2);
                *reinterpret_cast<u32x2*>(dst) = a;                                      // copy 0: x[col - 1]
                *reinterpret_cast<u32x2*>(dst + 4) = bb;
                a[0] = d[0]; a[1] = d[1]; bb[0] = d[2]; bb[1] = d[3];
                *reinterpret_cast<u32x2*>(dst + 32 * CPITCH) = a;                        // copy 1: x[col]
                *reinterpret_cast<u32x2*>(dst + 32 * CPITCH + 4) = bb;
                a[0] = __builtin_amdgcn_alignbyte(d[1], d[0], 2);  a[1] = __builtin_amdgcn_alignbyte(0u, d[1], 2);
                bb[0] = __builtin_amdgcn_alignbyte(d[3], d[2], 2); bb[1] = __builtin_amdgcn_alignbyte(0u, d[3], 2);
                *reinterpret_cast<u32x2*>(dst + 64 * CPITCH) = a;                        // copy 2: x[col + 1]
                *reinterpret_cast<u32x2*>(dst + 64 * CPITCH + 4) = bb;
            }
        } else
#pragma unroll
        for (int r = 0; r < NXT; ++r) {
            const int u = r * 256 + tid, pc = u % PW, rr = (u / PW) % RP, cl = u / (PW * RP);
            if (u < NX) {
                unsigned short* dst = xs + cl * CPITCH + rr * W + pc * 8;
                const u32x4 d = xr[r];
                const unsigned before = xe[r] << 16, after = xe[r] >> 16;           // `before` in the HIGH half: (d0 : before) >> 16 starts with it
                u32x4 lft, rgt;
                lft[0] = __builtin_amdgcn_alignbyte(d[0], before, 2);
                lft[1] = __builtin_amdgcn_alignbyte(d[1], d[0], 2);
                lft[2] = __builtin_amdgcn_alignbyte(d[2], d[1], 2);
                lft[3] = __builtin_amdgcn_alignbyte(d[3], d[2], 2);
                rgt[0] = __builtin_amdgcn_alignbyte(d[1], d[0], 2);
                rgt[1] = __builtin_amdgcn_alignbyte(d[2], d[1], 2);
                rgt[2] = __builtin_amdgcn_alignbyte(d[3], d[2], 2);
                rgt[3] = __builtin_amdgcn_alignbyte(after, d[3], 2);
                *reinterpret_cast<u32x4*>(dst) = lft;                                  // copy 0: x[col - 1]
                *reinterpret_cast<u32x4*>(dst + 32 * CPITCH) = d;                      // copy 1: x[col]
                *reinterpret_cast<u32x4*>(dst + 64 * CPITCH) = rgt;                    // copy 2: x[col + 1]
            }
        }
    };
    auto store_z = [&]() {
#pragma unroll
        for (int r = 0; r < 4 * MW; ++r) {
            const int u = r * 256 + tid, ml = u >> 5, pc = u & 31;
            *reinterpret_cast<u32x4*>(zs + ml * ZPITCH + pc * 8) = zr[r];
        }
    };

    const int rl = lane & 31, h = lane >> 5;
    const int msub = wave % MW, kpart = wave / MW;
    // K4: first tap row / column this plane sees (odd planes: {0, 1}, even planes: {1, 2}); wave-uniform (the CTW tiles lie in one plane)
    const int plane = K4 ? (ct0 * 32) / (Cin >> 2) : 0;
    const int ky_lo = K4 ? ((plane >> 1) ? 0 : 1) : 0, kx_lo = K4 ? ((plane & 1) ? 0 : 1) : 0;
    if constexpr (W == 4) {                                                      // the zero rows (and the pad) of the x image: written once
        for (int i = tid; i < 3 * 32 * CPITCH / 2; i += 256) reinterpret_cast<unsigned*>(xs)[i] = 0u;
    }
    int64_t it = ks;
    if (it < items) { load_z(it); load_x(it, 0); }
    for (; it < items; it += ksplit) {
#pragma unroll
        for (int sub = 0; sub < CTW; ++sub) {
            __syncthreads();                                                     // the previous stage's fragments have been read
            if (sub == 0) store_z();                                             // (the dz tile stays for all CTW channel tiles of the item)
            store_x();
            __syncthreads();
            // the next stage's global loads travel during this MFMA phase
            if (sub + 1 < CTW) load_x(it, sub + 1);
            else if (it + ksplit < items) { load_z(it + ksplit); load_x(it + ksplit, 0); }
#pragma unroll 4
            for (int tt = 0; tt < KSTEPS; ++tt) {
                const int t = kpart * KSTEPS + tt, p0 = t * 16, row = (p0 / (R * W)) * RPI + (p0 % (R * W)) / W, x0 = p0 % W;
                const u32x4 af = *reinterpret_cast<const u32x4*>(zs + (msub * 32 + rl) * ZPITCH + p0 + 8 * h);
#pragma unroll
                for (int ti = 0; ti < NT; ++ti) {
                    const int ky = K4 ? ky_lo + (ti >> 1) : ti / 3, kx = K4 ? kx_lo + (ti & 1) : ti % 3;
                    u32x4 bf;
                    if constexpr (W == 4) {
                        const unsigned short* src = xs + (kx * 32 + rl) * CPITCH + (row + ky) * W + 8 * h;
                        const u32x2 lo = *reinterpret_cast<const u32x2*>(src), hi = *reinterpret_cast<const u32x2*>(src + 4);
                        bf[0] = lo[0]; bf[1] = lo[1]; bf[2] = hi[0]; bf[3] = hi[1];
                    } else {
                        bf = *reinterpret_cast<const u32x4*>(xs + (kx * 32 + rl) * CPITCH + (row + ky) * W + x0 + 8 * h);
                    }
                    acc[sub][ti] = mfma16_32<CT>(af, bf, acc[sub][ti]);
                }
            }
        }
    }

    // ---- this wave's nine 32 x 32 partial sums -> slab (ks, kpart), laid out [tap][m][c]: the lanes of a store are 32 consecutive c (128-byte
    // runs).  In the weight's own [m][c][tap] order a wave store is 64 words 36 bytes apart, every one a sector of its own: the PMC write
    // counter showed 415-545 MB per launch for 37 MB of slabs; vs_conv3_wgrad_band_finish transposes once while it adds the slabs.
    float* out = slabs + ((int64_t)ks * KW + kpart) * ((int64_t)Cout * Cin * 9);
#pragma unroll
    for (int sub = 0; sub < CTW; ++sub) {
        const int c = (ct0 + sub) * 32 + rl;
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            const int t = K4 ? (ky_lo + (ti >> 1)) * 3 + kx_lo + (ti & 1) : ti;     // slab position [tap of the 3 x 3 form][m][c]
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = mt * (32 * MW) + msub * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
                if (m < Cout && c < Cin) out[((int64_t)t * Cout + m) * Cin + c] = acc[sub][ti][v];
            }
        }
    }
}

// dW[m][c][tap] = sum over the n slabs [tap][m][c] (+ the pending gradient): coalesced reads along c, nine consecutive words written per thread
__global__ __launch_bounds__(256) void wgrad_slab_finish_kernel(const float* __restrict__ src, int n, int Cout, int Cin, const float* __restrict__ addend,
                                                                float* __restrict__ out) {
    const int64_t mc = (int64_t)Cout * Cin, total = mc * 9;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < mc; i += (int64_t)gridDim.x * 256) {
        float s[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) s[t] = addend ? addend[i * 9 + t] : 0.f;
#pragma unroll 2
        for (int k = 0; k < n; ++k)                                                // fixed order: reproducible
#pragma unroll
            for (int t = 0; t < 9; ++t) s[t] += src[(int64_t)k * total + (int64_t)t * mc + i];
#pragma unroll
        for (int t = 0; t < 9; ++t) out[i * 9 + t] = s[t];
    }
}

// Wp[mt][chunk][ky][kx][lane][8]: lane (r = lane & 31, h = lane >> 5) holds W[m = 32 mt + r][c = 16 chunk + 8 h .. + 7][ky][kx]
// (flip = 0, w = [M][K][3][3]) or, for the input gradient (flip = 1, w = the conv's [K][M][3][3]), w[c][m][2 - ky][2 - kx]
template <int CT>
__global__ __launch_bounds__(256) void conv3_img16_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ dst, int M, int K, int flip, int64_t total) {
    const int chunks = ((K + 63) >> 6) * 4;                                         // contraction channels padded to whole 64-channel phases (zeros)
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int jj = (int)(e & 7), lane = (int)((e >> 3) & 63);
        int64_t t = e >> 9;
        const int kx = (int)(t % 3); t /= 3;
        const int ky = (int)(t % 3); t /= 3;
        const int chunk = (int)(t % chunks), mt = (int)(t / chunks);
        const int m = mt * 32 + (lane & 31), c = chunk * 16 + 8 * (lane >> 5) + jj;
        float v = 0.f;
        if (m < M && c < K)
            v = flip ? w[(((int64_t)c * M + m) * 3 + (2 - ky)) * 3 + (2 - kx)] : w[(((int64_t)m * K + c) * 3 + ky) * 3 + kx];
        dst[e] = vs_f2h(v, CT);
    }
}

// All the pre-packs of a step in ONE launch (every weight changes once per optimizer step, so every pack is stale once per step: 36-67
// launches of ~5 us at TaxiBJ / SST size otherwise).  A job = (w, dst, M, K, flip); 16-byte units (8 packed elements) are numbered across jobs.
constexpr int IPK_MAXJ = 96;
struct ImgPackJobs {
    const float* w[IPK_MAXJ];
    unsigned short* dst[IPK_MAXJ];
    int M[IPK_MAXJ], K[IPK_MAXJ], flip[IPK_MAXJ];
    long long unit_off[IPK_MAXJ + 1];
    int nj;
};
template <int CT>
__global__ __launch_bounds__(256) void conv3_img16_pack_multi_kernel(ImgPackJobs J) {
    // a TILE = (32 packed rows m, 16 contraction channels c) = 9 taps x 64 lanes x 8 elements, contiguous in the pack (9 KiB).  Its source
    // is 32 runs of 144 floats (flip = 0: w[m][c0 .. c0 + 15][9]) or 16 runs of 288 floats (flip = 1: w[c][m0 .. m0 + 31][9]): read
    // coalesced into LDS as [m][c][tap], written as 16-byte pieces (thread-per-element gathers read one 4-byte word per cache line:
    // 139 us for the 56 packs of a TaxiBJ step)
    __shared__ float tile[32 * 16 * 9];
    const long long total = J.unit_off[J.nj];                                       // (unit = tile here)
    for (long long gt = blockIdx.x; gt < total; gt += gridDim.x) {
        int lo = 0, hi = J.nj;                                                      // job of this tile: unit_off[lo] <= gt < unit_off[lo + 1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (J.unit_off[mid] <= gt) lo = mid; else hi = mid;
        }
        const int j = lo;
        const int M = J.M[j], K = J.K[j], flip = J.flip[j], chunks = ((K + 63) >> 6) * 4;
        const long long tl = gt - J.unit_off[j];
        const int chunk = (int)(tl % chunks), mt = (int)(tl / chunks);
        const float* w = J.w[j];
        if (!flip) {
            for (int idx = threadIdx.x; idx < 32 * 144; idx += 256) {
                const int mi = idx / 144, r = idx - mi * 144, m = mt * 32 + mi;
                tile[idx] = (m < M && chunk * 16 + r / 9 < K) ? w[((int64_t)m * K + chunk * 16) * 9 + r] : 0.f;
            }
        } else {
            for (int idx = threadIdx.x; idx < 16 * 288; idx += 256) {
                const int ci = idx / 288, r = idx - ci * 288, mi = r / 9, t = 8 - (r - mi * 9), m = mt * 32 + mi;
                tile[(mi * 16 + ci) * 9 + t] = (m < M && chunk * 16 + ci < K) ? w[((int64_t)(chunk * 16 + ci) * M + mt * 32) * 9 + r] : 0.f;
            }
        }
        __syncthreads();
        unsigned short* dst = J.dst[j] + tl * 4608;
        for (int u = threadIdx.x; u < 576; u += 256) {
            const int lane = u & 63, t = u >> 6, r = lane & 31, h = lane >> 5;
            unsigned short o[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) o[jj] = vs_f2h(tile[(r * 16 + 8 * h + jj) * 9 + t], CT);
            *reinterpret_cast<u32x4*>(dst + u * 8) = *reinterpret_cast<const u32x4*>(o);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ slabs, int nslabs, int64_t total, const float* __restrict__ bias,
                                                       const float* __restrict__ addend, const float* __restrict__ addend2, int C, int HW,
                                                       void* __restrict__ out, int od) {
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (int64_t)gridDim.x * 1024) {
        f32x4 s = *reinterpret_cast<const f32x4*>(slabs + i);
#pragma unroll 4
        for (int k = 1; k < nslabs; ++k) {                                         // fixed order: reproducible
            const f32x4 t = *reinterpret_cast<const f32x4*>(slabs + (int64_t)k * total + i);
            s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
        }
        if (bias) {
            const float bv = bias[(i / HW) % C];
            s[0] += bv; s[1] += bv; s[2] += bv; s[3] += bv;
        }
        if (addend) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(addend + i);
            s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
        }
        if (addend2) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(addend2 + i);
            s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
        }
        if (od == VS_F32) *reinterpret_cast<f32x4*>((float*)out + i) = s;
        else *reinterpret_cast<u16x4*>((unsigned short*)out + i) = u16x4{vs_f2h(s[0], od), vs_f2h(s[1], od), vs_f2h(s[2], od), vs_f2h(s[3], od)};
    }
}

// partial[g][i] = sum of the slabs of group g (consecutive runs of `per` slabs), fixed order: the first pass over MANY slabs of a small
// tensor (hundreds of weight-gradient slabs of a 64 x 64 x 3 x 3 weight: one pass would be 36 workgroups adding 512 values each)
__global__ __launch_bounds__(256) void slab_sum_grouped_kernel(const float* __restrict__ slabs, int nslabs, int per, int64_t total, float* __restrict__ partial) {
    const int g = blockIdx.y, s0 = g * per, s1 = s0 + per < nslabs ? s0 + per : nslabs;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (int64_t)gridDim.x * 1024) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int k = s0; k < s1; ++k) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(slabs + (int64_t)k * total + i);
            s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
        }
        *reinterpret_cast<f32x4*>(partial + (int64_t)g * total + i) = s;
    }
}

// input-channel splits: enough workgroups to cover the chip once, at most 256 channels (144 KiB of LDS) per split, >= 64 per split
int img16_splits(int B, int Cin, int Cout) {
    const int64_t base = (int64_t)B * vs_cdiv(Cout, 32);
    int s = 1;
    while (Cin / s > 256 || (base * s < 256 && Cin / (2 * s) >= 64)) {
        if ((Cin / s) % 128 != 0) break;                                           // halving would leave a split that is not a multiple of 64
        s *= 2;
    }
    // VS_CONV_IMG_PAIR=1: one full round of one workgroup per CU with splits of > 128 channels is halved and run as TWO workgroups per CU.
    // Measured on the SST step and NOT the default: 25.66 vs 25.52 ms (twice the slabs for the BatchNorm launch to add, no faster launch).
    static const int pair_mode = getenv("VS_CONV_IMG_PAIR") ? atoi(getenv("VS_CONV_IMG_PAIR")) : 0;
    if (pair_mode && base * s >= 256 && base * s < 512 && Cin / s > 128 && (Cin / s) % 128 == 0) s *= 2;
    return s;
}

}  // namespace

extern "C" int vs_conv3_img16_supported(int compute, int B, int Cin, int H, int W, int Cout) {
    if (!vs_is16(compute) || H != 16 || W != 16 || B < 1 || Cout < 1 || Cin < 64 || Cin % 64 != 0) return 0;
    const int s = img16_splits(B, Cin, Cout);
    if (Cin % s != 0 || (Cin / s) % 64 != 0 || Cin / s > 256) return 0;
    if ((int64_t)B * vs_cdiv(Cout, 32) * s >= (1ll << 31)) return 0;
    return 1;
}

extern "C" int vs_conv3_img16_splits(int B, int Cin, int Cout) { return img16_splits(B, Cin, Cout); }

// (the contraction is padded to whole 64-channel phases with zeros: vs_conv3_band stages zeros for the channels a last phase lacks)
extern "C" size_t vs_conv3_img16_packed_elems(int Cin, int Cout) { return (size_t)vs_cdiv(Cout, 32) * (size_t)(vs_cdiv(Cin, 64) * 4) * 9 * 512; }

// flip = 0: w is the Conv2d weight [Cout][Cin][3][3].  flip = 1 (input gradient): pass the SAME tensor with Cin := the conv's Cout
// (the contraction) and Cout := the conv's Cin (the rows).
extern "C" int vs_conv3_img16_pack_weight(int compute, const float* w, int Cin, int Cout, int flip, void* dst, void* stream) {
    VS_CHECK_ARG(vs_is16(compute) && w && dst && Cin >= 1 && Cout > 0, "vs_conv3_img16_pack_weight: bad argument");
    const int64_t total = (int64_t)vs_conv3_img16_packed_elems(Cin, Cout);
    int64_t blocks = vs_cdiv(total, 256);
    if (blocks > 2048) blocks = 2048;
    if (compute == VS_BF16)
        hipLaunchKernelGGL(conv3_img16_pack_kernel<VS_BF16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)dst, Cout, Cin, flip, total);
    else
        hipLaunchKernelGGL(conv3_img16_pack_kernel<VS_F16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)dst, Cout, Cin, flip, total);
    VS_CHECK_LAUNCH("vs_conv3_img16_pack_weight");
    return VS_OK;
}

// n_jobs (<= 96) pre-packs at once: job j = vs_conv3_img16_pack_weight(compute, w[j], K[j], M[j], flip[j], dst[j]) (K = contraction channels, M = rows)
extern "C" int vs_conv3_img16_pack_weights(int compute, int n_jobs, const float* const* w, const int* K, const int* M, const int* flip, void* const* dst,
                                           void* stream) {
    VS_CHECK_ARG(vs_is16(compute) && n_jobs >= 1 && n_jobs <= IPK_MAXJ && w && K && M && flip && dst, "vs_conv3_img16_pack_weights: bad argument (1..%d jobs)",
                 IPK_MAXJ);
    ImgPackJobs J;
    J.nj = n_jobs;
    J.unit_off[0] = 0;
    for (int j = 0; j < n_jobs; ++j) {
        VS_CHECK_ARG(w[j] && dst[j] && M[j] > 0 && K[j] >= 1 && (uintptr_t)dst[j] % 16 == 0, "vs_conv3_img16_pack_weights: bad job %d", j);
        J.w[j] = w[j]; J.dst[j] = (unsigned short*)dst[j]; J.M[j] = M[j]; J.K[j] = K[j]; J.flip[j] = flip[j];
        J.unit_off[j + 1] = J.unit_off[j] + (long long)(vs_conv3_img16_packed_elems(K[j], M[j]) / 4608);       // tiles of (32 rows, 16 channels)
    }
    long long blocks = J.unit_off[n_jobs];
    if (blocks > 4096) blocks = 4096;
    if (compute == VS_BF16) hipLaunchKernelGGL(conv3_img16_pack_multi_kernel<VS_BF16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, J);
    else hipLaunchKernelGGL(conv3_img16_pack_multi_kernel<VS_F16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, J);
    VS_CHECK_LAUNCH("vs_conv3_img16_pack_weights");
    return VS_OK;
}

// x [B][Cin][16][16] (16-bit), w_packed from vs_conv3_img16_pack_weight -> slabs [vs_conv3_img16_splits][B][Cout][256] fp32 (no bias)
extern "C" int vs_conv3_img16(int compute, const void* x, const void* w_packed, float* slabs, int B, int Cin, int Cout, void* stream) {
    VS_CHECK_ARG(x && w_packed && slabs, "vs_conv3_img16: bad argument");
    VS_CHECK_ARG(vs_conv3_img16_supported(compute, B, Cin, 16, 16, Cout), "vs_conv3_img16: unsupported geometry (query vs_conv3_img16_supported)");
    VS_CHECK_ARG(((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)slabs) % 16 == 0, "vs_conv3_img16: operands must be 16-byte aligned");
    const int splits = img16_splits(B, Cin, Cout), cs = Cin / splits, mtiles = (int)vs_cdiv(Cout, 32);
    const size_t lds = (size_t)cs * IMG_CPITCH * 2;
    const dim3 grid((unsigned)((int64_t)B * mtiles * splits));
    const bool pair = cs <= 128 && grid.x > 256;                                 // two workgroups per CU fit (72 KiB each) and there are enough of them
    auto kb = pair ? conv3_img16_kernel<VS_BF16, 4, 2> : conv3_img16_kernel<VS_BF16, 12, 1>;
    auto kh = pair ? conv3_img16_kernel<VS_F16, 4, 2> : conv3_img16_kernel<VS_F16, 12, 1>;
    static bool attr_set[2] = {false, false};
    if (!attr_set[pair]) {
        if (hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * IMG_CPITCH * 2) != hipSuccess ||
            hipFuncSetAttribute((const void*)kh, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * IMG_CPITCH * 2) != hipSuccess)
            return vs_fail(VS_ERR_LAUNCH, "vs_conv3_img16: cannot raise the dynamic LDS limit");
        attr_set[pair] = true;
    }
    const char* wse = getenv("VS_IMG16_WIDE_STORE");                              // (read per call: A/B in one process; the LDS image holds >= 32 KiB: cs >= 64)
    const int wide = !(wse && wse[0] == '0') && lds >= (size_t)32 * 1024;
    if (compute == VS_BF16)
        hipLaunchKernelGGL(kb, grid, dim3(256), lds, (hipStream_t)stream, (const unsigned short*)x, (const u32x4*)w_packed, slabs, B, Cin, Cout, cs, splits, mtiles, wide);
    else
        hipLaunchKernelGGL(kh, grid, dim3(256), lds, (hipStream_t)stream, (const unsigned short*)x, (const u32x4*)w_packed, slabs, B, Cin, Cout, cs, splits, mtiles, wide);
    VS_CHECK_LAUNCH("vs_conv3_img16");
    return VS_OK;
}

// ---- the fused layer (conv3_img16_bn_kernel): workspace = [epoch word][error word] .. 256 | area B (1 MiB) | area A (16 MiB), zero-filled by the caller
// once and kept; one workspace per stream (launches on a stream follow each other, so they can share the areas)
constexpr size_t IMGBN_B_OFF = 256, IMGBN_B_BYTES = (size_t)1 << 20, IMGBN_A_OFF = IMGBN_B_OFF + IMGBN_B_BYTES, IMGBN_A_BYTES = (size_t)16 << 20;

extern "C" size_t vs_conv3_img16_bn_workspace_bytes(void) { return IMGBN_A_OFF + IMGBN_A_BYTES; }

extern "C" int vs_conv3_img16_bn_supported(int compute, int B, int Cin, int Cout) {
    if (!vs_conv3_img16_supported(compute, B, Cin, 16, 16, Cout)) return 0;
    const int s = img16_splits(B, Cin, Cout), mtiles = (int)vs_cdiv(Cout, 32);
    if (s != 1 && s != 2 && s != 8) return 0;
    static const int cus = getenv("VS_IMG_BN_MAX_WGS") ? atoi(getenv("VS_IMG_BN_MAX_WGS")) : 256;       // every workgroup must be resident: one per CU
    if ((int64_t)B * mtiles * s > cus) return 0;
    if (B > 64 || (int64_t)mtiles * 32 * B * 16 > (int64_t)IMGBN_B_BYTES) return 0;
    return 1;
}

// the (activation, y type) pairs the one-launch layer is built for: forward LeakyReLU with a 16-bit y, forward no activation with an fp32 y;
// backward LeakyReLU or none
extern "C" int vs_conv3_img16_bn_form_supported(int backward, int act, int y_dtype, int compute) {
    if (act != VS_ACT_NONE && act != VS_ACT_LEAKY) return 0;
    if (backward) return 1;
    return (act == VS_ACT_LEAKY && y_dtype == compute) || (act == VS_ACT_NONE && y_dtype == VS_F32);
}

extern "C" int vs_exchange_epoch_advance(void* ws, void* stream) {
    VS_CHECK_ARG(ws && (uintptr_t)ws % 16 == 0, "vs_exchange_epoch_advance: bad argument");
    hipLaunchKernelGGL(exchange_epoch_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned*)ws);
    VS_CHECK_LAUNCH("vs_exchange_epoch_advance");
    return VS_OK;
}

static int imgbn_launch(int compute, int mode, ImgBnArgs& a, void* ws, unsigned call_idx, hipStream_t stream) {
    static const unsigned spin = getenv("VS_IMG_BN_SPIN") ? (unsigned)atol(getenv("VS_IMG_BN_SPIN")) : 2000000u;
    a.splits = img16_splits(a.B, a.Cin, a.Cout);
    a.cs = a.Cin / a.splits;
    a.mtiles = (int)vs_cdiv(a.Cout, 32);
    char* base = (char*)ws;
    a.epoch_base = (const unsigned*)base;
    a.xerr = vs_g_exchange_guard ? vs_g_exchange_guard : (unsigned*)(base + 4);
    a.xerr_bit = vs_g_exchange_guard ? 2u : 1u;
    a.xb = (xg64*)(base + IMGBN_B_OFF);
    a.xa = (xg64*)(base + IMGBN_A_OFF);
    a.call_idx = call_idx;
    a.spin_limit = spin;
    // the image (cs channels) or the epilogue's tiles -- partial sums [32][256], two pixel-major tiles of <= [32][264] and the small rows --
    // whichever is larger
    size_t lds = (size_t)a.cs * IMG_CPITCH * 2;
    if (lds < (size_t)104 * 1024) lds = (size_t)104 * 1024;
    const dim3 grid((unsigned)((int64_t)a.B * a.mtiles * a.splits));
#define VS_IMGBN_GO(CTV, SV, MV, AV, YV)                                                                                                  \
    do {                                                                                                                                  \
        auto kern = conv3_img16_bn_kernel<CTV, SV, MV, AV, YV>;                                                                           \
        static bool attr_set = false;                                                                                                     \
        if (!attr_set) {                                                                                                                  \
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * IMG_CPITCH * 2) != hipSuccess)   \
                return vs_fail(VS_ERR_LAUNCH, "vs_conv3_img16_bn: cannot raise the dynamic LDS limit");                                   \
            attr_set = true;                                                                                                              \
        }                                                                                                                                 \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);                                                                        \
    } while (0)
#define VS_IMGBN_S(CTV, MV, AV, YV)                                  \
    do {                                                             \
        if (a.splits == 1) VS_IMGBN_GO(CTV, 1, MV, AV, YV);          \
        else if (a.splits == 2) VS_IMGBN_GO(CTV, 2, MV, AV, YV);     \
        else VS_IMGBN_GO(CTV, 8, MV, AV, YV);                        \
    } while (0)
    // the forms the ConvResBlock needs: forward LeakyReLU -> 16-bit y (inner layers), forward no activation -> fp32 y (the block's last layer),
    // backward through LeakyReLU / no activation
    const bool leaky = a.act == VS_ACT_LEAKY, yf = a.yd == VS_F32;
#define VS_IMGBN_CT(CTV)                                                                  \
    do {                                                                                  \
        if (mode == 0 && leaky && !yf) VS_IMGBN_S(CTV, 0, VS_ACT_LEAKY, 0);               \
        else if (mode == 0 && !leaky && yf) VS_IMGBN_S(CTV, 0, VS_ACT_NONE, 1);           \
        else if (mode == 1 && leaky) VS_IMGBN_S(CTV, 1, VS_ACT_LEAKY, 0);                 \
        else if (mode == 1) VS_IMGBN_S(CTV, 1, VS_ACT_NONE, 0);                           \
        else return vs_fail(VS_ERR_UNSUPPORTED, "vs_conv3_img16_bn: this (activation, output type) pair is not built (query vs_conv3_img16_bn_form_supported)"); \
    } while (0)
    if (compute == VS_BF16) VS_IMGBN_CT(VS_BF16);
    else VS_IMGBN_CT(VS_F16);
#undef VS_IMGBN_CT
#undef VS_IMGBN_S
#undef VS_IMGBN_GO
    VS_CHECK_LAUNCH("vs_conv3_img16_bn");
    return VS_OK;
}

// y = act(BatchNorm_train(conv3x3(x) + bias)) of ONE reference call (all B maps are the call's batch) in one launch: z [B][Cout][256]
// (the 16-bit pre-BatchNorm output, kept for backward), y in y_dtype, mean / invstd [Cout], running estimates updated; skip != NULL (the
// block's last layer): xnew = skip + y (fp32) and its 16-bit copy xnew16.  call_idx: 1 .. 65535, distinct for every launch since the last
// vs_exchange_epoch_advance on this workspace.
extern "C" int vs_conv3_img16_bn_fwd(int compute, const void* x, const void* w_packed, void* ws, unsigned call_idx, const float* bias, const float* gamma,
                                     const float* beta, int act, float* running_mean, float* running_var, float momentum, float eps, void* z, void* y,
                                     int y_dtype, float* mean, float* invstd, const float* skip, float* xnew, void* xnew16, int B, int Cin, int Cout,
                                     void* stream) {
    VS_CHECK_ARG(x && w_packed && ws && gamma && beta && z && y && mean && invstd && vs_dtype_ok(y_dtype), "vs_conv3_img16_bn_fwd: bad argument");
    VS_CHECK_ARG(vs_conv3_img16_bn_supported(compute, B, Cin, Cout), "vs_conv3_img16_bn_fwd: unsupported geometry (query vs_conv3_img16_bn_supported)");
    VS_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr) && (!skip || xnew), "vs_conv3_img16_bn_fwd: running_mean/var, skip/xnew come together");
    VS_CHECK_ARG(vs_conv3_img16_bn_form_supported(0, act, y_dtype, compute), "vs_conv3_img16_bn_fwd: (activation, y type) not built (query vs_conv3_img16_bn_form_supported)");
    VS_CHECK_ARG(call_idx >= 1 && call_idx < 65536, "vs_conv3_img16_bn_fwd: call_idx out of range (advance the epoch base)");
    VS_CHECK_ARG(((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)ws) % 16 == 0, "vs_conv3_img16_bn_fwd: operands must be 16-byte aligned");
    ImgBnArgs a = {};
    a.X = (const unsigned short*)x; a.Wp = (const u32x4*)w_packed; a.B = B; a.Cin = Cin; a.Cout = Cout;
    a.bias = bias; a.gamma = gamma; a.beta = beta; a.act = act; a.eps = eps; a.momentum = momentum;
    a.rmean = running_mean; a.rvar = running_var; a.mean = mean; a.invstd = invstd;
    a.z = (unsigned short*)z; a.y = y; a.yd = y_dtype; a.skip = skip; a.xnew = xnew; a.xnew16 = (unsigned short*)xnew16;
    return imgbn_launch(compute, 0, a, ws, call_idx, (hipStream_t)stream);
}

// The backward layer: dy = conv3x3_input_gradient(dz_next) (w_packed: the FOLLOWING layer's weight packed with flip = 1; Cin = its output
// channels, Cout = this layer's channels), then this layer's BatchNorm + activation backward from the stored z, mean, invstd:
// dz [B][Cout][256] (16-bit); d gamma / d beta written or (accumulate) ADDED to dgamma / dbeta.
extern "C" int vs_conv3_img16_bn_bwd(int compute, const void* dz_next, const void* w_packed, void* ws, unsigned call_idx, const void* z, const float* mean,
                                     const float* invstd, const float* gamma, const float* beta, int act, void* dz, float* dgamma, float* dbeta,
                                     int accumulate, int B, int Cin, int Cout, void* stream) {
    VS_CHECK_ARG(dz_next && w_packed && ws && z && mean && invstd && gamma && beta && dz && dgamma && dbeta, "vs_conv3_img16_bn_bwd: bad argument");
    VS_CHECK_ARG(vs_conv3_img16_bn_supported(compute, B, Cin, Cout), "vs_conv3_img16_bn_bwd: unsupported geometry (query vs_conv3_img16_bn_supported)");
    VS_CHECK_ARG(call_idx >= 1 && call_idx < 65536, "vs_conv3_img16_bn_bwd: call_idx out of range (advance the epoch base)");
    VS_CHECK_ARG(vs_conv3_img16_bn_form_supported(1, act, 0, compute), "vs_conv3_img16_bn_bwd: activation not built (query vs_conv3_img16_bn_form_supported)");
    VS_CHECK_ARG(((uintptr_t)dz_next | (uintptr_t)w_packed | (uintptr_t)ws) % 16 == 0, "vs_conv3_img16_bn_bwd: operands must be 16-byte aligned");
    ImgBnArgs a = {};
    a.X = (const unsigned short*)dz_next; a.Wp = (const u32x4*)w_packed; a.B = B; a.Cin = Cin; a.Cout = Cout;
    a.gamma = gamma; a.beta = beta; a.act = act; a.mean = (float*)mean; a.invstd = (float*)invstd;
    a.z = (unsigned short*)z; a.dz = (unsigned short*)dz; a.dgamma = dgamma; a.dbeta = dbeta; a.accumulate = accumulate;
    return imgbn_launch(compute, 1, a, ws, call_idx, (hipStream_t)stream);
}

// out[b][c][p] = sum_s slabs[s][b][c][p] (+ bias[c]) (+ addend[b][c][p], fp32) in out_dtype; HW a multiple of 4
extern "C" int vs_slab_sum2(const float* slabs, int nslabs, const float* bias, const float* addend, const float* addend2, void* out, int out_dtype, int B,
                            int C, int64_t HW, void* stream) {
    VS_CHECK_ARG(slabs && out && nslabs >= 1 && B > 0 && C > 0 && HW > 0 && HW % 4 == 0 && vs_dtype_ok(out_dtype), "vs_slab_sum: bad argument");
    VS_CHECK_ARG(((uintptr_t)slabs | (uintptr_t)out | (uintptr_t)addend | (uintptr_t)addend2) % 16 == 0, "vs_slab_sum: operands must be 16-byte aligned");
    const int64_t total = (int64_t)B * C * HW;
    int64_t blocks = vs_cdiv(total, 1024);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, slabs, nslabs, total, bias, addend, addend2, C, (int)HW, out,
                       out_dtype);
    VS_CHECK_LAUNCH("vs_slab_sum");
    return VS_OK;
}

extern "C" int vs_slab_sum(const float* slabs, int nslabs, const float* bias, const float* addend, void* out, int out_dtype, int B, int C, int64_t HW,
                           void* stream) {
    return vs_slab_sum2(slabs, nslabs, bias, addend, nullptr, out, out_dtype, B, C, HW, stream);
}

// ---- row-band form: many maps, W in {16, 32, 64} with H a multiple of 256 / W, or 8 x 8 / 4 x 4 maps; Cin a multiple of 64; y in any type, bias added --
extern "C" int vs_conv3_band_supported(int compute, int B, int Cin, int H, int W, int Cout) {
    // (Cin need not be a multiple of 64: the last phase stages zeros for the channels it lacks and the pre-pack holds zero weights there)
    if (!vs_is16(compute) || (W != 4 && W != 8 && W != 16 && W != 32 && W != 64) || B < 1 || Cout < 1 || Cin < 1) return 0;
    if (W == 4) return H == 4 && (int64_t)vs_cdiv(B, 16) * vs_cdiv(Cout, 32) < (1ll << 31);      // whole 4 x 4 maps, sixteen per workgroup
    if (W == 8) return H == 8 && (int64_t)vs_cdiv(B, 4) * vs_cdiv(Cout, 32) < (1ll << 31);       // whole 8 x 8 maps, four per workgroup
    const int R = 256 / W;
    if (H < R || H % R != 0) return 0;
    if ((int64_t)B * (H / R) * vs_cdiv(Cout, 32) >= (1ll << 31)) return 0;
    return 1;
}

// bytes of one staged 64-channel phase (conv3_band_kernel's BUF)
template <int W>
constexpr size_t band_buf_bytes() {
    constexpr int IPB = W == 8 ? 4 : (W == 4 ? 16 : 1), RI = W == 8 ? 8 : (W == 4 ? 4 : 256 / W);
    return (size_t)(64 * IPB * (RI + 2) * W + (W == 4 ? 8 : 0)) * 2;
}

template <int W, int AHEAD, int MINB, int K4 = 0>
static int launch_band(int compute, const void* x, const void* w_packed, const float* bias, void* y, int y_dtype, int B, int Cin, int H, int Cout,
                       hipStream_t stream, double* bn_sums = nullptr, int maps_per_group = 1) {
    constexpr int R = W == 8 ? 8 : (W == 4 ? 4 : 256 / W), IPB = W == 8 ? 4 : (W == 4 ? 16 : 1);
    const size_t lds = (size_t)(Cin > 64 ? 2 : 1) * band_buf_bytes<W>();         // one buffer when there is a single 64-channel phase
    auto kb = conv3_band_kernel<VS_BF16, W, AHEAD, MINB, K4>;
    auto kh = conv3_band_kernel<VS_F16, W, AHEAD, MINB, K4>;
    static bool attr_set = false;
    if (!attr_set) {
        const int most = (int)(2 * band_buf_bytes<W>());
        if (hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, most) != hipSuccess ||
            hipFuncSetAttribute((const void*)kh, hipFuncAttributeMaxDynamicSharedMemorySize, most) != hipSuccess)
            return vs_fail(VS_ERR_LAUNCH, "vs_conv3_band: cannot raise the dynamic LDS limit");
        attr_set = true;
    }
    const int Cpad = (int)vs_cdiv(Cin, 64) * 64;                                    // the kernel walks whole 64-channel phases
    const int mtiles = (int)vs_cdiv(Cout, 32), bands = IPB > 1 ? (int)vs_cdiv(B, IPB) : H / R;
    const dim3 grid((unsigned)((int64_t)(IPB > 1 ? 1 : B) * bands * mtiles));
    static const int xcd_remap = getenv("VS_BAND_XCD") ? atoi(getenv("VS_BAND_XCD")) : 1;
    const int remap = xcd_remap && mtiles > 1 && grid.x >= 64;
    if (compute == VS_BF16)
        hipLaunchKernelGGL(kb, grid, dim3(256), lds, stream, (const unsigned short*)x, (const u32x4*)w_packed, bias, y, y_dtype, B, Cpad, H, Cout, mtiles, bands,
                           remap, bn_sums, maps_per_group, Cin);
    else
        hipLaunchKernelGGL(kh, grid, dim3(256), lds, stream, (const unsigned short*)x, (const u32x4*)w_packed, bias, y, y_dtype, B, Cpad, H, Cout, mtiles, bands,
                           remap, bn_sums, maps_per_group, Cin);
    return VS_OK;
}

// k4 s2 p1 on parity planes (K4 form of the kernel): 8 fragment groups per phase -> rings of 4 (two workgroups per CU) or 8
template <int W>
static int launch_band_k4_w(int compute, const void* x, const void* w_packed, const float* bias, void* y, int y_dtype, int B, int Cin, int H, int Cout,
                            hipStream_t stream, double* bn_sums = nullptr, int mpg = 1) {
    static const int pair_mode = getenv("VS_CONV_BAND_PAIR") ? atoi(getenv("VS_CONV_BAND_PAIR")) : 1;
    const size_t lds = 2 * band_buf_bytes<W>();                                  // Cin = 4 K >= 256: always two buffers
    if (pair_mode && 2 * lds <= 160 * 1024) return launch_band<W, 4, 2, 1>(compute, x, w_packed, bias, y, y_dtype, B, Cin, H, Cout, stream, bn_sums, mpg);
    return launch_band<W, 8, 1, 1>(compute, x, w_packed, bias, y, y_dtype, B, Cin, H, Cout, stream, bn_sums, mpg);
}

template <int W>
static int launch_band_w(int compute, const void* x, const void* w_packed, const float* bias, void* y, int y_dtype, int B, int Cin, int H, int Cout,
                         hipStream_t stream, double* bn_sums = nullptr, int mpg = 1) {
    // two workgroups per CU where 2 x LDS fits (VS_CONV_BAND_PAIR=0: always the deep-prefetch form)
    static const int pair_mode = getenv("VS_CONV_BAND_PAIR") ? atoi(getenv("VS_CONV_BAND_PAIR")) : 1;
    const size_t lds = (size_t)(Cin > 64 ? 2 : 1) * band_buf_bytes<W>();
    if (pair_mode && 2 * lds <= 160 * 1024) return launch_band<W, 4, 2>(compute, x, w_packed, bias, y, y_dtype, B, Cin, H, Cout, stream, bn_sums, mpg);
    return launch_band<W, 12, 1>(compute, x, w_packed, bias, y, y_dtype, B, Cin, H, Cout, stream, bn_sums, mpg);
}

// round 5: the form with both operands through LDS (csrc/vs_conv_band2.hip) serves every call that wants no BatchNorm sums from the epilogue;
// VS_BAND_V2=0 (read per call: tools/band_bench.py runs A/B pairs in one process) restores the round-2 kernel
int vs_band2_go(int compute, const void* x, const void* w_packed, const float* bias, void* y, int y_dtype, int B, int Cin, int H, int W, int Cout, int k4,
                hipStream_t stream);
static inline bool band2_enabled() {
    const char* e = getenv("VS_BAND_V2");
    return !(e && e[0] == '0');
}

// x [B][Cin][H][W] (16-bit), w_packed from vs_conv3_img16_pack_weight (same pre-pack) -> y [B][Cout][H][W] in y_dtype, bias added
// bn_sums != NULL: the (sum, sum of squares) of the stored outputs are ADDED to bn_sums[group][Cout][2] (fp64; group = map / (B / groups)) -- what the
// BatchNorm behind the convolution needs (vs_bn_stats_from_sums_fold), without a statistics pass over y.  Maps of >= 8 x 8 pixels only
// (vs_conv3_band_bn_supported).
extern "C" int vs_conv3_band_bn_supported(int compute, int B, int Cin, int H, int W, int Cout, int groups) {
    return vs_conv3_band_supported(compute, B, Cin, H, W, Cout) && W >= 8 && groups >= 1 && B % groups == 0;     // (a wave's 64 pixels lie in ONE map)
}

extern "C" int vs_conv3_band_bn(int compute, const void* x, const void* w_packed, const float* bias, void* y, int y_dtype, int B, int Cin, int H, int W,
                                int Cout, double* bn_sums, int groups, void* stream) {
    VS_CHECK_ARG(x && w_packed && y && vs_dtype_ok(y_dtype), "vs_conv3_band: bad argument");
    VS_CHECK_ARG(vs_conv3_band_supported(compute, B, Cin, H, W, Cout), "vs_conv3_band: unsupported geometry (query vs_conv3_band_supported)");
    VS_CHECK_ARG(!bn_sums || vs_conv3_band_bn_supported(compute, B, Cin, H, W, Cout, groups), "vs_conv3_band_bn: statistics not served for this geometry");
    VS_CHECK_ARG(((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)y) % 16 == 0 && (uintptr_t)bn_sums % 8 == 0, "vs_conv3_band: operands must be 16-byte aligned");
    const int mpg = bn_sums ? B / groups : 1;
    int rc;
    if (!bn_sums && band2_enabled()) {
        rc = vs_band2_go(compute, x, w_packed, bias, y, y_dtype, B, Cin, H, W, Cout, 0, (hipStream_t)stream);
        if (rc != VS_OK) return rc;
        VS_CHECK_LAUNCH("vs_conv3_band (v2)");
        return VS_OK;
    }
    if (W == 64) rc = launch_band_w<64>(compute, x, w_packed, bias, y, y_dtype, B, Cin, H, Cout, (hipStream_t)stream, bn_sums, mpg);
    else if (W == 32) rc = launch_band_w<32>(compute, x, w_packed, bias, y, y_dtype, B, Cin, H, Cout, (hipStream_t)stream, bn_sums, mpg);
    else if (W == 16) rc = launch_band_w<16>(compute, x, w_packed, bias, y, y_dtype, B, Cin, H, Cout, (hipStream_t)stream, bn_sums, mpg);
    else if (W == 8) rc = launch_band_w<8>(compute, x, w_packed, bias, y, y_dtype, B, Cin, H, Cout, (hipStream_t)stream, bn_sums, mpg);
    else rc = launch_band_w<4>(compute, x, w_packed, bias, y, y_dtype, B, Cin, H, Cout, (hipStream_t)stream);
    if (rc != VS_OK) return rc;
    VS_CHECK_LAUNCH("vs_conv3_band");
    return VS_OK;
}

extern "C" int vs_conv3_band(int compute, const void* x, const void* w_packed, const float* bias, void* y, int y_dtype, int B, int Cin, int H, int W,
                             int Cout, void* stream) {
    return vs_conv3_band_bn(compute, x, w_packed, bias, y, y_dtype, B, Cin, H, W, Cout, nullptr, 1, stream);
}

// The same statistics WITHOUT atomics: every workgroup (W >= 16: one row band of one map; W = 8: every map) writes the (sum, sum of squares) of
// the stored values of its 32 channels to its own row of parts [vs_conv3_band_bn_parts_rows(B, H, W)][Cout][2] (fp32; row = map * bands + band,
// so the rows of a BatchNorm call group are consecutive); vs_bn_stats_from_parts_fold adds a group's rows in a fixed order (fp64) and does what
// vs_bn_stats_from_sums_fold does.  Reproducible launch to launch, no zero fill.  k4 != 0: x are parity planes [B][4 K][H][W], Cin = 4 K, the
// pack of vs_conv_k4s2_pack_weight (the gather of csrc/vs_conv_k4s2.hip).
extern "C" int vs_conv3_band_bn_parts_rows(int B, int H, int W) {
    if (W == 8) return B;
    if (W != 16 && W != 32 && W != 64) return 0;
    return B * (H / (256 / W));
}

extern "C" int vs_conv3_band_bn_parts(int compute, const void* x, const void* w_packed, const float* bias, void* y, int y_dtype, int B, int Cin, int H, int W,
                                      int Cout, float* parts, int k4, void* stream) {
    VS_CHECK_ARG(x && w_packed && y && parts && vs_dtype_ok(y_dtype), "vs_conv3_band_bn_parts: bad argument");
    VS_CHECK_ARG(vs_conv3_band_bn_supported(compute, B, Cin, H, W, Cout, 1), "vs_conv3_band_bn_parts: statistics not served for this geometry");
    VS_CHECK_ARG(((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)y) % 16 == 0 && (uintptr_t)parts % 8 == 0, "vs_conv3_band_bn_parts: operands must be 16-byte aligned");
    VS_CHECK_ARG(!k4 || Cin % 4 == 0, "vs_conv3_band_bn_parts: the planes form needs Cin = 4 K");
    double* tag = reinterpret_cast<double*>(parts);                               // (maps_per_group = 0 selects the table form in the kernel)
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (k4 && vs_conv_k4s2_skip_form(Cin / 4)) {
        if (W == 64) rc = launch_band_k4_w<64>(compute, x, w_packed, bias, y, y_dtype, B, Cin, H, Cout, st, tag, 0);
        else if (W == 32) rc = launch_band_k4_w<32>(compute, x, w_packed, bias, y, y_dtype, B, Cin, H, Cout, st, tag, 0);
        else if (W == 16) rc = launch_band_k4_w<16>(compute, x, w_packed, bias, y, y_dtype, B, Cin, H, Cout, st, tag, 0);
        else rc = launch_band_k4_w<8>(compute, x, w_packed, bias, y, y_dtype, B, Cin, H, Cout, st, tag, 0);
    } else {
        if (W == 64) rc = launch_band_w<64>(compute, x, w_packed, bias, y, y_dtype, B, Cin, H, Cout, st, tag, 0);
        else if (W == 32) rc = launch_band_w<32>(compute, x, w_packed, bias, y, y_dtype, B, Cin, H, Cout, st, tag, 0);
        else if (W == 16) rc = launch_band_w<16>(compute, x, w_packed, bias, y, y_dtype, B, Cin, H, Cout, st, tag, 0);
        else rc = launch_band_w<8>(compute, x, w_packed, bias, y, y_dtype, B, Cin, H, Cout, st, tag, 0);
    }
    if (rc != VS_OK) return rc;
    VS_CHECK_LAUNCH("vs_conv3_band_bn_parts");
    return VS_OK;
}

// ---- weight gradient on row bands: x [B][Cin][H][W], dz [B][Cout][H][W] (16-bit) -> fp32 slabs [vs_conv3_wgrad_band_slabs][Cout][Cin][3][3] ----
extern "C" int vs_conv3_wgrad_band_supported(int compute, int B, int Cin, int H, int W, int Cout) {
    if (!vs_is16(compute) || (W != 4 && W != 8 && W != 16 && W != 32 && W != 64) || B < 1 || Cout < 8 || Cin < 8) return 0;
    if (W == 4) return H == 4;                                                   // whole 4 x 4 maps, sixteen per item
    if (W == 8) return H == 8;                                                   // whole 8 x 8 maps, four per item
    const int R = 256 / W;
    if (H < R || H % R != 0) return 0;
    return 1;
}

static int wgrad_band_mw(int Cout) { return Cout > 64 ? 4 : (Cout > 32 ? 2 : 1); }      // 32-row sub-tiles per workgroup

static int wgrad_band_ksplit(int B, int Cin, int H, int W, int Cout, int ctw = 1) {
    const int mw = wgrad_band_mw(Cout);
    const int64_t tiles = vs_cdiv(Cout, 32 * mw) * vs_cdiv(Cin, 32 * ctw), items = W == 8 ? vs_cdiv(B, 4) : (W == 4 ? vs_cdiv(B, 16) : (int64_t)B * (H / (256 / W)));
    // ONE round of workgroups (a workgroup's LDS fills a CU): every share of the bands costs a slab of the weight's size, written and read
    // again by the finish pass -- with two rounds (512) the slabs of a TaxiBJ step were 3.9 GB of traffic: 9.80 -> 9.37 ms with 256; 128, 192
    // and 384 are slower (idle CUs / a partial second round)
    static const int target_wgs = getenv("VS_WGRAD_BAND_WGS") ? atoi(getenv("VS_WGRAD_BAND_WGS")) : 256;
    int64_t ks = vs_cdiv(target_wgs, tiles);
    if (ks > items) ks = items;
    const int64_t slab_bytes = (int64_t)Cout * Cin * 9 * 4;
    while (ks > 1 && ks * (4 / mw) * slab_bytes > ((int64_t)96 << 20)) --ks;    // at most 96 MiB of slabs
    return (int)(ks < 1 ? 1 : ks);
}

// round 5: the form with both operands staged by LDS-DMA and the column shift taken in registers (csrc/vs_conv_wgrad2.hip); VS_WGRAD_V2=0
// (read per call: the slab query and the launch must see the same value) restores the round-2 kernel.  The k4 s2 family keeps the round-2 kernel.
int vs_wgrad2_slabs(int B, int Cin, int H, int W, int Cout);
int vs_wgrad2_go(int compute, int npieces, const void* const* x, const void* const* dz, int maps_per_piece, float* slabs, int B, int Cin, int H, int W, int Cout,
                 int k4, hipStream_t stream);
static inline bool wgrad2_enabled() {
    const char* e = getenv("VS_WGRAD_V2");
    return !(e && e[0] == '0');
}
// the k4 s2 family: built and correct (tools/band_bench.py k4 --check), but a plane's 2 x 2 taps leave four MFMAs per k-step against the same
// staged tiles -- the launch is bound by the LDS-DMA (52 KiB per 1 150 cycles) and loses to the round-2 kernel, which walks two channel tiles
// per staged dz tile: 256 -> 349 us, 192 -> 263 us on the Moving-MNIST decoder layers, the step 6.11 -> 6.32 ms.  Opt-in: VS_WGRAD_V2_K4=1.
static inline bool wgrad2_k4_enabled() {
    const char* e = getenv("VS_WGRAD_V2_K4");
    return wgrad2_enabled() && e && e[0] == '1';
}

extern "C" int vs_conv3_wgrad_band_slabs(int B, int Cin, int H, int W, int Cout) {
    if (wgrad2_enabled()) return vs_wgrad2_slabs(B, Cin, H, W, Cout);
    return (4 / wgrad_band_mw(Cout)) * wgrad_band_ksplit(B, Cin, H, W, Cout);
}

// K4 form: two 32-channel tiles of a plane per workgroup against one staged dz tile.  Four (VS_WGRAD_K4_CTW4=1, where a plane holds a multiple of
// 128 channels) is built and measured SLOWER: half as many tiles means twice the batch shares to fill the chip, i.e. twice the slabs, and the
// fourth stage of an item no longer hides its loads -- Moving-MNIST 6.45 (two) vs 6.71 ms (four).
static int wgrad_k4_ctw(int Cin) {
    static const int allow4 = getenv("VS_WGRAD_K4_CTW4") ? atoi(getenv("VS_WGRAD_K4_CTW4")) : 0;
    return (allow4 && (Cin >> 2) % 128 == 0) ? 4 : 2;
}

template <int W, int MW, int K4, int CTW>
static void launch_wgrad_band_ctw(int compute, const WgradPieces& pieces, float* slabs, int B, int Cin, int H, int Cout, int ksplit, hipStream_t stream) {
    const size_t lds = (size_t)(3 * 32 * wgrad_cpitch<W>() + 32 * MW * 264) * 2;
    auto kb = wgrad3_band_kernel<VS_BF16, W, MW, K4, CTW>;
    auto kh = wgrad3_band_kernel<VS_F16, W, MW, K4, CTW>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)kh, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int mtiles = (int)vs_cdiv(Cout, 32 * MW), ctiles = (int)vs_cdiv(Cin, 32 * CTW);
    const dim3 grid((unsigned)((int64_t)mtiles * ctiles * ksplit));
    if (compute == VS_BF16)
        hipLaunchKernelGGL(kb, grid, dim3(256), lds, stream, pieces, slabs, B, Cin, H, Cout, ctiles, ksplit);
    else
        hipLaunchKernelGGL(kh, grid, dim3(256), lds, stream, pieces, slabs, B, Cin, H, Cout, ctiles, ksplit);
}

template <int W, int MW, int K4 = 0>
static void launch_wgrad_band(int compute, const WgradPieces& pieces, float* slabs, int B, int Cin, int H, int Cout, int ksplit, hipStream_t stream) {
    if constexpr (K4) {
        if (wgrad_k4_ctw(Cin) == 4) launch_wgrad_band_ctw<W, MW, K4, 4>(compute, pieces, slabs, B, Cin, H, Cout, ksplit, stream);
        else launch_wgrad_band_ctw<W, MW, K4, 2>(compute, pieces, slabs, B, Cin, H, Cout, ksplit, stream);
    } else {
        launch_wgrad_band_ctw<W, MW, K4, 1>(compute, pieces, slabs, B, Cin, H, Cout, ksplit, stream);
    }
}

template <int W, int K4 = 0>
static void launch_wgrad_band_w(int compute, const WgradPieces& pieces, float* slabs, int B, int Cin, int H, int Cout, int ksplit, hipStream_t stream) {
    const int mw = wgrad_band_mw(Cout);
    if (mw == 4) launch_wgrad_band<W, 4, K4>(compute, pieces, slabs, B, Cin, H, Cout, ksplit, stream);
    else if (mw == 2) launch_wgrad_band<W, 2, K4>(compute, pieces, slabs, B, Cin, H, Cout, ksplit, stream);
    else launch_wgrad_band<W, 1, K4>(compute, pieces, slabs, B, Cin, H, Cout, ksplit, stream);
}

template <int K4 = 0>
static int wgrad_band_go(int compute, const WgradPieces& pieces, float* slabs, int B, int Cin, int H, int W, int Cout, hipStream_t stream) {
    const int ks = wgrad_band_ksplit(B, Cin, H, W, Cout, K4 ? wgrad_k4_ctw(Cin) : 1);
    if (W == 64) launch_wgrad_band_w<64, K4>(compute, pieces, slabs, B, Cin, H, Cout, ks, stream);
    else if (W == 32) launch_wgrad_band_w<32, K4>(compute, pieces, slabs, B, Cin, H, Cout, ks, stream);
    else if (W == 16) launch_wgrad_band_w<16, K4>(compute, pieces, slabs, B, Cin, H, Cout, ks, stream);
    else if (W == 8) launch_wgrad_band_w<8, K4>(compute, pieces, slabs, B, Cin, H, Cout, ks, stream);
    else launch_wgrad_band_w<4, K4>(compute, pieces, slabs, B, Cin, H, Cout, ks, stream);
    VS_CHECK_LAUNCH("vs_conv3_wgrad_band");
    return VS_OK;
}

extern "C" int vs_conv3_wgrad_band(int compute, const void* x, const void* dz, float* slabs, int B, int Cin, int H, int W, int Cout, void* stream) {
    VS_CHECK_ARG(x && dz && slabs, "vs_conv3_wgrad_band: bad argument");
    VS_CHECK_ARG(vs_conv3_wgrad_band_supported(compute, B, Cin, H, W, Cout), "vs_conv3_wgrad_band: unsupported geometry (query vs_conv3_wgrad_band_supported)");
    VS_CHECK_ARG(((uintptr_t)x | (uintptr_t)dz | (uintptr_t)slabs) % 16 == 0, "vs_conv3_wgrad_band: operands must be 16-byte aligned");
    if (wgrad2_enabled()) {
        const int rc = vs_wgrad2_go(compute, 1, &x, &dz, B, slabs, B, Cin, H, W, Cout, 0, (hipStream_t)stream);
        if (rc != VS_OK) return rc;
        VS_CHECK_LAUNCH("vs_conv3_wgrad_band (v2)");
        return VS_OK;
    }
    WgradPieces pieces = {};
    pieces.x[0] = (const unsigned short*)x;
    pieces.dz[0] = (const unsigned short*)dz;
    pieces.maps_per_piece = B;
    return wgrad_band_go<0>(compute, pieces, slabs, B, Cin, H, W, Cout, (hipStream_t)stream);
}

// ---- k4 s2 p1 on parity planes (csrc/vs_conv_k4s2.hip): planes [B][4 K][H][W] ----------------------------------------------------------
// K a multiple of 64: the forms of the kernels that skip the five zero taps of every plane (weights packed by vs_conv_k4s2_pack_weight in
// the matching 2 x 2 form); otherwise the plain 3 x 3 kernels on a zero-padded 3 x 3 pack.
extern "C" int vs_conv_k4s2_skip_form(int K) { return K >= 64 && K % 64 == 0; }

extern "C" int vs_conv_k4s2_band_bn(int compute, const void* planes, const void* w_packed, const float* bias, void* y, int y_dtype, int B, int K, int H, int W,
                                    int M, double* bn_sums, int groups, void* stream) {
    VS_CHECK_ARG(planes && w_packed && y && vs_dtype_ok(y_dtype), "vs_conv_k4s2_band: bad argument");
    VS_CHECK_ARG(vs_conv3_band_supported(compute, B, 4 * K, H, W, M), "vs_conv_k4s2_band: unsupported geometry (query vs_conv3_band_supported on the planes)");
    VS_CHECK_ARG(!bn_sums || vs_conv3_band_bn_supported(compute, B, 4 * K, H, W, M, groups), "vs_conv_k4s2_band_bn: statistics not served for this geometry");
    VS_CHECK_ARG(((uintptr_t)planes | (uintptr_t)w_packed | (uintptr_t)y) % 16 == 0 && (uintptr_t)bn_sums % 8 == 0, "vs_conv_k4s2_band: operands must be 16-byte aligned");
    if (!vs_conv_k4s2_skip_form(K)) return vs_conv3_band_bn(compute, planes, w_packed, bias, y, y_dtype, B, 4 * K, H, W, M, bn_sums, groups, stream);
    const int mpg = bn_sums ? B / groups : 1;
    int rc;
    if (!bn_sums && band2_enabled()) {
        rc = vs_band2_go(compute, planes, w_packed, bias, y, y_dtype, B, 4 * K, H, W, M, 1, (hipStream_t)stream);
        if (rc != VS_OK) return rc;
        VS_CHECK_LAUNCH("vs_conv_k4s2_band (v2)");
        return VS_OK;
    }
    if (W == 64) rc = launch_band_k4_w<64>(compute, planes, w_packed, bias, y, y_dtype, B, 4 * K, H, M, (hipStream_t)stream, bn_sums, mpg);
    else if (W == 32) rc = launch_band_k4_w<32>(compute, planes, w_packed, bias, y, y_dtype, B, 4 * K, H, M, (hipStream_t)stream, bn_sums, mpg);
    else if (W == 16) rc = launch_band_k4_w<16>(compute, planes, w_packed, bias, y, y_dtype, B, 4 * K, H, M, (hipStream_t)stream, bn_sums, mpg);
    else if (W == 8) rc = launch_band_k4_w<8>(compute, planes, w_packed, bias, y, y_dtype, B, 4 * K, H, M, (hipStream_t)stream, bn_sums, mpg);
    else rc = launch_band_k4_w<4>(compute, planes, w_packed, bias, y, y_dtype, B, 4 * K, H, M, (hipStream_t)stream);
    if (rc != VS_OK) return rc;
    VS_CHECK_LAUNCH("vs_conv_k4s2_band");
    return VS_OK;
}

extern "C" int vs_conv_k4s2_band(int compute, const void* planes, const void* w_packed, const float* bias, void* y, int y_dtype, int B, int K, int H, int W,
                                 int M, void* stream) {
    return vs_conv_k4s2_band_bn(compute, planes, w_packed, bias, y, y_dtype, B, K, H, W, M, nullptr, 1, stream);
}

// slabs vs_conv_k4s2_wgrad_band writes (the skip form shares the batch among fewer, heavier workgroups than vs_conv3_wgrad_band on the same planes)
extern "C" int vs_conv_k4s2_wgrad_band_slabs(int B, int K, int H, int W, int M) {
    if (wgrad2_k4_enabled() && vs_conv_k4s2_skip_form(K)) return vs_wgrad2_slabs(B, 4 * K, H, W, M);
    return (4 / wgrad_band_mw(M)) * wgrad_band_ksplit(B, 4 * K, H, W, M, vs_conv_k4s2_skip_form(K) ? wgrad_k4_ctw(4 * K) : 1);
}

extern "C" int vs_conv_k4s2_wgrad_band(int compute, const void* planes, const void* small, float* slabs, int B, int K, int H, int W, int M, void* stream) {
    VS_CHECK_ARG(planes && small && slabs, "vs_conv_k4s2_wgrad_band: bad argument");
    VS_CHECK_ARG(vs_conv3_wgrad_band_supported(compute, B, 4 * K, H, W, M), "vs_conv_k4s2_wgrad_band: unsupported geometry");
    VS_CHECK_ARG(((uintptr_t)planes | (uintptr_t)small | (uintptr_t)slabs) % 16 == 0, "vs_conv_k4s2_wgrad_band: operands must be 16-byte aligned");
    if (wgrad2_k4_enabled() && vs_conv_k4s2_skip_form(K)) {
        const int rc = vs_wgrad2_go(compute, 1, &planes, &small, B, slabs, B, 4 * K, H, W, M, 1, (hipStream_t)stream);
        if (rc != VS_OK) return rc;
        VS_CHECK_LAUNCH("vs_conv_k4s2_wgrad_band (v2)");
        return VS_OK;
    }
    WgradPieces pieces = {};
    pieces.x[0] = (const unsigned short*)planes;
    pieces.dz[0] = (const unsigned short*)small;
    pieces.maps_per_piece = B;
    if (vs_conv_k4s2_skip_form(K)) return wgrad_band_go<1>(compute, pieces, slabs, B, 4 * K, H, W, M, (hipStream_t)stream);
    return wgrad_band_go<0>(compute, pieces, slabs, B, 4 * K, H, W, M, (hipStream_t)stream);
}

// The same over a batch that lies in `npieces` (<= 64) separate tensors of `maps_per_piece` maps each (x[i] [maps][Cin][H][W], dz[i]
// [maps][Cout][H][W]): the gradient over their concatenation without building it.  Slabs as for B = npieces * maps_per_piece.
extern "C" int vs_conv3_wgrad_band_pieces(int compute, int npieces, const void* const* x, const void* const* dz, int maps_per_piece, float* slabs, int Cin,
                                          int H, int W, int Cout, void* stream) {
    VS_CHECK_ARG(x && dz && slabs && npieces >= 1 && npieces <= WG_MAX_PIECES && maps_per_piece >= 1, "vs_conv3_wgrad_band_pieces: bad argument");
    const int B = npieces * maps_per_piece;
    VS_CHECK_ARG(vs_conv3_wgrad_band_supported(compute, B, Cin, H, W, Cout), "vs_conv3_wgrad_band_pieces: unsupported geometry");
    VS_CHECK_ARG(W != 8 || npieces == 1 || maps_per_piece % 4 == 0, "vs_conv3_wgrad_band_pieces: 8 x 8 maps go four at a time: maps_per_piece must be a multiple of 4");
    VS_CHECK_ARG(W != 4 || npieces == 1 || maps_per_piece % 16 == 0, "vs_conv3_wgrad_band_pieces: 4 x 4 maps go sixteen at a time: maps_per_piece must be a multiple of 16");
    if (wgrad2_enabled()) {
        for (int i = 0; i < npieces; ++i)
            VS_CHECK_ARG(x[i] && dz[i] && ((uintptr_t)x[i] | (uintptr_t)dz[i]) % 16 == 0, "vs_conv3_wgrad_band_pieces: every piece must be a 16-byte aligned tensor");
        VS_CHECK_ARG((uintptr_t)slabs % 16 == 0, "vs_conv3_wgrad_band_pieces: slabs must be 16-byte aligned");
        const int rc = vs_wgrad2_go(compute, npieces, x, dz, maps_per_piece, slabs, B, Cin, H, W, Cout, 0, (hipStream_t)stream);
        if (rc != VS_OK) return rc;
        VS_CHECK_LAUNCH("vs_conv3_wgrad_band_pieces (v2)");
        return VS_OK;
    }
    WgradPieces pieces = {};
    for (int i = 0; i < npieces; ++i) {
        VS_CHECK_ARG(x[i] && dz[i] && ((uintptr_t)x[i] | (uintptr_t)dz[i]) % 16 == 0, "vs_conv3_wgrad_band_pieces: every piece must be a 16-byte aligned tensor");
        pieces.x[i] = (const unsigned short*)x[i];
        pieces.dz[i] = (const unsigned short*)dz[i];
    }
    pieces.maps_per_piece = maps_per_piece;
    VS_CHECK_ARG((uintptr_t)slabs % 16 == 0, "vs_conv3_wgrad_band_pieces: slabs must be 16-byte aligned");
    return wgrad_band_go<0>(compute, pieces, slabs, B, Cin, H, W, Cout, (hipStream_t)stream);
}

// partial[g][i] = sum over the slabs g * per .. g * per + per - 1 (per = ceil(nslabs / groups)) of slabs[s][i], i < total (a multiple of 4):
// first pass over many slabs of a small tensor; vs_slab_sum over the `groups` partials finishes.
extern "C" int vs_slab_sum_grouped(const float* slabs, int nslabs, int groups, float* partial, int64_t total, void* stream) {
    VS_CHECK_ARG(slabs && partial && nslabs >= 1 && groups >= 1 && groups <= nslabs && total > 0 && total % 4 == 0, "vs_slab_sum_grouped: bad argument");
    VS_CHECK_ARG(((uintptr_t)slabs | (uintptr_t)partial) % 16 == 0, "vs_slab_sum_grouped: operands must be 16-byte aligned");
    const int per = (nslabs + groups - 1) / groups;
    int64_t blocks = vs_cdiv(total, 1024);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(slab_sum_grouped_kernel, dim3((unsigned)blocks, (unsigned)groups), dim3(256), 0, (hipStream_t)stream, slabs, nslabs, per, total, partial);
    VS_CHECK_LAUNCH("vs_slab_sum_grouped");
    return VS_OK;
}

// dW [Cout][Cin][3][3] = sum of the nslabs partial gradients vs_conv3_wgrad_band left ([tap][Cout][Cin] each; for many slabs of a small weight
// run vs_slab_sum_grouped over them first: it is layout-blind) + addend (the pending gradient, or NULL); out may be addend.
extern "C" int vs_conv3_wgrad_band_finish(const float* slabs, int nslabs, const float* addend, float* out, int Cout, int Cin, void* stream) {
    VS_CHECK_ARG(slabs && out && nslabs >= 1 && Cout > 0 && Cin > 0, "vs_conv3_wgrad_band_finish: bad argument");
    int64_t blocks = vs_cdiv((int64_t)Cout * Cin, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wgrad_slab_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, slabs, nslabs, Cout, Cin, addend, out);
    VS_CHECK_LAUNCH("vs_conv3_wgrad_band_finish");
    return VS_OK;
}
