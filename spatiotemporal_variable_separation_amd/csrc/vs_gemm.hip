// vs_gemm.hip -- LDS-tiled MFMA GEMM for gfx950:  C[m,n] = epi( sum_k A(m,k) * B(n,k) ).
//
// One kernel template serves every dense contraction of the var_sep hot path (Linear forward, input
// gradient, weight gradient; see include/varsep_hip.h).  Design points, all MI355X-specific:
//   * 64-wide wavefronts, 4 waves per workgroup arranged 2x2; every wave owns a (BM/2)x(BN/2) block of C as
//     TMxTN accumulators of the 32x32 MFMA shape (16 fp32 registers per lane each).
//   * compute type bf16 -> v_mfma_f32_32x32x16_bf16, fp32 accumulate;
//     compute type f32  -> v_mfma_f32_32x32x2_f32 (bit-exact fp32 FMA chain, the parity path).
//   * operands may be "R" (reduction index contiguous) or "S" (output index contiguous) in memory.  Tiles
//     are staged global -> registers -> LDS with 16-byte vectors along the contiguous axis and kept in LDS
//     in their NATIVE orientation:  R tiles as [row][BK+pad] read back with ds_read_b128;  S tiles as
//     [k][rows+pad] read back with ds_read_b64_tr_b16 (bf16: the hardware transposing read delivers the
//     k-contiguous fragment the MFMA wants) or plain ds_read_b32 (f32: one element per lane).  No operand is
//     ever transposed in HBM, so the weight-gradient GEMM (both operands "S") costs the same as the forward.
//   * pads are chosen so every LDS read is bank-conflict free (MI355X: 64 banks x 4 B, b128 reads served
//     in 16-lane groups, tr_b16/b64 reads in 32-lane halves).
//   * register prefetch of tile t+1 is issued before the MFMAs of tile t (global latency hides under the
//     matrix pipe); one LDS buffer, two barriers per K tile; 2-4 workgroups per CU overlap each other.
//   * few-tile/long-K problems (encoder first layer: M=128..256, K=20480) are split along K over
//     gridDim.z into fp32 slabs and combined by a second kernel that also applies the epilogue, so the
//     result is bitwise reproducible (no float atomics).
#include "vs_gemm_core.h"
#include "vs_gemm_glds.h"
#include "vs_gemm_big.h"
#include "vs_gemm_mid.h"
#include "vs_gemm_p8.h"
#include "vs_loss.h"

namespace {

template <int CT, int LA, int LB, int BM, int BN>
int launch(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, const Plan& plan,
           const Epi& epi, float* slabs, hipStream_t stream) {
    typedef typename CTraits<CT>::T T;
    constexpr int BK = bk_of<CT>();
    constexpr int U = CTraits<CT>::U;
    // 16-byte vector loads need every problem of a batch to start 16-byte aligned too
    Dense<CT, LA> a{(const T*)A, lda, M, K, ((uintptr_t)A % 16 == 0) && (lda % U == 0) && (epi.batch_a % U == 0)};
    Dense<CT, LB> b{(const T*)B, ldb, N, K, ((uintptr_t)B % 16 == 0) && (ldb % U == 0) && (epi.batch_b % U == 0)};
    constexpr size_t smem = (TileGeom<CT, LA, BM, BK>::ELEMS + TileGeom<CT, LB, BN, BK>::ELEMS) * sizeof(T);
    const int batch = epi.splits_per_batch > 0 ? plan.batch : 1;
    dim3 grid((unsigned)vs_cdiv(N, BN), (unsigned)vs_cdiv(M, BM), (unsigned)(plan.splits * batch));
    // XCD runs (Epi::xcd_runs) from 64 workgroups upwards: PMC on the WaveEq decoder layers (3328 x 1200 x 1200, 988 tiles of 64 x 64) showed
    // 77 MB fetched per launch for 11 MB of operands -- each of the eight L2s pulled the whole of A and B.  VS_GEMM_XCD=0: the plain map.
    static const int xcd_mode = getenv("VS_GEMM_XCD") ? atoi(getenv("VS_GEMM_XCD")) : 1;
    Epi e = epi;
    e.xcd_runs = xcd_mode && (int64_t)grid.x * grid.y * grid.z >= 64;
    hipLaunchKernelGGL((gemm_kernel<CT, Dense<CT, LA>, Dense<CT, LB>, BM, BN, BK>), grid, dim3(256), smem, stream, a, b, M, N,
                       K, (int)plan.k_tiles_per_split, e, slabs);
    VS_CHECK_LAUNCH("vs_gemm");
    return VS_OK;
}

template <int CT, int LA, int LB>
int launch_glds(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, const Plan& plan, const Epi& epi,
                float* slabs, hipStream_t stream) {
    const int batch = epi.splits_per_batch > 0 ? plan.batch : 1;
    dim3 grid((unsigned)vs_cdiv(N, 128), (unsigned)vs_cdiv(M, 128), (unsigned)(plan.splits * batch));
    static const int forced = getenv("VS_GEMM_GLDS_STAGES") ? atoi(getenv("VS_GEMM_GLDS_STAGES")) : 0;
    const int stages = forced ? forced : ((int64_t)grid.x * grid.y * grid.z >= 1024 ? 2 : 1);
    if (stages == 2)
        hipLaunchKernelGGL((gemm_glds_kernel<LA, LB, false, 2, CT>), grid, dim3(256), 65536, stream, (const __bf16*)A, lda, (const __bf16*)B, ldb, M, N,
                           K, (int)plan.k_tiles_per_split, epi, slabs);
    else
        hipLaunchKernelGGL((gemm_glds_kernel<LA, LB, false, 1, CT>), grid, dim3(256), 32768, stream, (const __bf16*)A, lda, (const __bf16*)B, ldb, M, N,
                           K, (int)plan.k_tiles_per_split, epi, slabs);
    VS_CHECK_LAUNCH("vs_gemm (LDS-DMA tile)");
    return VS_OK;
}

template <int CT, int LA, int LB, bool LOSS = false>
int launch_big(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, const BigPlan& bp, int batch, const Epi& epi,
               float* slabs, hipStream_t stream) {
    if constexpr (CT == VS_F32) {
        return vs_fail(VS_ERR_UNSUPPORTED, "vs_gemm: the 256x256 tile is a 16-bit kernel");
    } else {
        auto kfn = gemm_big_kernel<CT, LA, LB, false, LOSS>;
        static bool attr_set = false;                  // 128 KiB of dynamic LDS: above the 64 KiB default limit
        if (!attr_set) {
            if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_BIG_STAGES * BIG_TILE_BYTES) != hipSuccess)
                return vs_fail(VS_ERR_LAUNCH, "vs_gemm: cannot raise the dynamic LDS limit to 160 KiB");
            attr_set = true;
        }
        dim3 grid((unsigned)(bp.tiles_m * bp.tiles_n), 1, (unsigned)(bp.splits * batch));
        hipLaunchKernelGGL(kfn, grid, dim3(512), GEMM_BIG_STAGES * BIG_TILE_BYTES, stream, (const unsigned short*)A, lda, (const unsigned short*)B, ldb, M, N, K,
                           (int)bp.k_tiles_per_split, bp.tiles_n, epi, slabs);
        VS_CHECK_LAUNCH("vs_gemm (256x256 tile)");
        return VS_OK;
    }
}

template <int CT>
int launch_big_layout(int la, int lb, const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, const BigPlan& bp,
                      int batch, const Epi& epi, float* slabs, hipStream_t stream) {
    if (la == LR && lb == LR) return launch_big<CT, LR, LR>(A, lda, B, ldb, M, N, K, bp, batch, epi, slabs, stream);
    if (la == LR && lb == LS) return launch_big<CT, LR, LS>(A, lda, B, ldb, M, N, K, bp, batch, epi, slabs, stream);
    if (la == LS && lb == LR) return launch_big<CT, LS, LR>(A, lda, B, ldb, M, N, K, bp, batch, epi, slabs, stream);
    return launch_big<CT, LS, LS>(A, lda, B, ldb, M, N, K, bp, batch, epi, slabs, stream);
}

// the 256 x 256 / 256 x 128 tile with two staggered wave groups (vs_gemm_p8.h)
template <int CT, int LA, int LB, int NI, int LOSS = 0, int MI = 4>
int launch_p8(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, const P8Plan& pp, int batch, const Epi& epi, float* slabs,
              hipStream_t stream) {
    if constexpr (CT == VS_F32) {
        return vs_fail(VS_ERR_UNSUPPORTED, "vs_gemm: the staggered 256-row tile is a 16-bit kernel");
    } else {
        auto kfn = gemm_p8_kernel<CT, LA, LB, NI, false, LOSS, MI>;
        constexpr int lds = 2 * (2 * 32 * MI * P8_BK * 2 + 2 * 64 * NI * P8_BK * 2);
        static bool attr_set = false;                  // 96 / 128 KiB of dynamic LDS: above the 64 KiB default limit
        if (!attr_set) {
            if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
                return vs_fail(VS_ERR_LAUNCH, "vs_gemm: cannot raise the dynamic LDS limit to %d bytes", lds);
            attr_set = true;
        }
        dim3 grid((unsigned)(pp.tiles_m * pp.tiles_n), 1, (unsigned)(pp.splits * batch));
        static const int plain = getenv("VS_GEMM_P8_QUICK") && atoi(getenv("VS_GEMM_P8_QUICK")) == 0;
        Epi e = epi;
        e.p8_plain = plain;
        hipLaunchKernelGGL(kfn, grid, dim3(512), lds, stream, (const unsigned short*)A, lda, (const unsigned short*)B, ldb, M, N, K, (int)pp.k_tiles_per_split,
                           pp.tiles_n, e, slabs);
        VS_CHECK_LAUNCH("vs_gemm (staggered 256-row tile)");
        return VS_OK;
    }
}

template <int CT>
int launch_p8_layout(int la, int lb, const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, const P8Plan& pp, int batch,
                     const Epi& epi, float* slabs, hipStream_t stream) {
    if (pp.ni == 1 && pp.mi == 2) {
        if (la == LR && lb == LR) return launch_p8<CT, LR, LR, 1, 0, 2>(A, lda, B, ldb, M, N, K, pp, batch, epi, slabs, stream);
        if (la == LR && lb == LS) return launch_p8<CT, LR, LS, 1, 0, 2>(A, lda, B, ldb, M, N, K, pp, batch, epi, slabs, stream);
        if (la == LS && lb == LR) return launch_p8<CT, LS, LR, 1, 0, 2>(A, lda, B, ldb, M, N, K, pp, batch, epi, slabs, stream);
        return launch_p8<CT, LS, LS, 1, 0, 2>(A, lda, B, ldb, M, N, K, pp, batch, epi, slabs, stream);
    }
    if (pp.ni == 1) {
        if (la == LR && lb == LR) return launch_p8<CT, LR, LR, 1>(A, lda, B, ldb, M, N, K, pp, batch, epi, slabs, stream);
        if (la == LR && lb == LS) return launch_p8<CT, LR, LS, 1>(A, lda, B, ldb, M, N, K, pp, batch, epi, slabs, stream);
        if (la == LS && lb == LR) return launch_p8<CT, LS, LR, 1>(A, lda, B, ldb, M, N, K, pp, batch, epi, slabs, stream);
        return launch_p8<CT, LS, LS, 1>(A, lda, B, ldb, M, N, K, pp, batch, epi, slabs, stream);
    }
    if (la == LR && lb == LR) return launch_p8<CT, LR, LR, 2>(A, lda, B, ldb, M, N, K, pp, batch, epi, slabs, stream);
    if (la == LR && lb == LS) return launch_p8<CT, LR, LS, 2>(A, lda, B, ldb, M, N, K, pp, batch, epi, slabs, stream);
    if (la == LS && lb == LR) return launch_p8<CT, LS, LR, 2>(A, lda, B, ldb, M, N, K, pp, batch, epi, slabs, stream);
    return launch_p8<CT, LS, LS, 2>(A, lda, B, ldb, M, N, K, pp, batch, epi, slabs, stream);
}

inline P8Plan p8_plan_for(int compute, int64_t M, int64_t N, int64_t K, int64_t batch, const void* A, int64_t lda, int la, const void* B, int64_t ldb, int lb,
                          int64_t stride_a, int64_t stride_b) {
    P8Plan pp = make_p8_plan(compute, M, N, K, batch, lb);
    if (pp.use && !(glds_operand_ok(A, lda, la, M, K, stride_a) && glds_operand_ok(B, ldb, lb, N, K, stride_b))) pp.use = false;
    if (pp.use && (lda >= (1ll << 22) || ldb >= (1ll << 22))) pp.use = false;       // 32-bit lane offsets inside a tile
    return pp;
}

template <int CT, int LA, int LB>
int launch_mid(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, const MidPlan& mp, int batch, const Epi& epi,
               float* slabs, hipStream_t stream) {
    const int rc = mid_launch<CT, LA, LB>(A, lda, B, ldb, M, N, K, mp.splits, mp.k_tiles_per_split, mp.stages, batch, epi, slabs, stream);
    if (rc != VS_OK) return rc;
    VS_CHECK_LAUNCH("vs_gemm (128x128 ring tile)");
    return VS_OK;
}

template <int CT>
int launch_mid_layout(int la, int lb, const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, const MidPlan& mp,
                      int batch, const Epi& epi, float* slabs, hipStream_t stream) {
    if (la == LR && lb == LR) return launch_mid<CT, LR, LR>(A, lda, B, ldb, M, N, K, mp, batch, epi, slabs, stream);
    if (la == LR && lb == LS) return launch_mid<CT, LR, LS>(A, lda, B, ldb, M, N, K, mp, batch, epi, slabs, stream);
    if (la == LS && lb == LR) return launch_mid<CT, LS, LR>(A, lda, B, ldb, M, N, K, mp, batch, epi, slabs, stream);
    return launch_mid<CT, LS, LS>(A, lda, B, ldb, M, N, K, mp, batch, epi, slabs, stream);
}

inline MidPlan mid_plan_for(int compute, int64_t M, int64_t N, int64_t K, int64_t batch, const void* A, int64_t lda, int la, const void* B,
                            int64_t ldb, int lb, int64_t stride_a, int64_t stride_b) {
    MidPlan mp = make_mid_plan(compute, M, N, K, batch);
    if (mp.use && !(glds_operand_ok(A, lda, la, M, K, stride_a) && glds_operand_ok(B, ldb, lb, N, K, stride_b))) mp.use = false;
    if (mp.use && (lda >= (1ll << 23) || ldb >= (1ll << 23))) mp.use = false;       // 32-bit lane offsets inside a tile
    return mp;
}

// the 256x256 tile is taken when its plan says so and both operands fit the LDS-DMA loader (alignment, multiples of 8)
inline BigPlan big_plan_for(int compute, int64_t M, int64_t N, int64_t K, int64_t batch, const void* A, int64_t lda, int la, const void* B,
                            int64_t ldb, int lb, int64_t stride_a, int64_t stride_b) {
    BigPlan bp = make_big_plan(compute, M, N, K, batch);
    if (bp.use && !(glds_operand_ok(A, lda, la, M, K, stride_a) && glds_operand_ok(B, ldb, lb, N, K, stride_b))) bp.use = false;
    if (bp.use && (lda >= (1ll << 22) || ldb >= (1ll << 22))) bp.use = false;       // 32-bit lane offsets inside a tile
    return bp;
}

template <int CT, int LA, int LB>
int launch_tile(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, const Plan& plan,
                const Epi& epi, float* slabs, hipStream_t stream) {
    if constexpr (CT != VS_F32 && LA == LR && LB == LR) {
        // LDS-DMA staged tile (vs_gemm_glds.h) wherever the plan picks 128x128 (>= 1024 tiles).  Measured on MI355X
        // (tools/gemm_bench.py): 732 vs 588 TF/s at 4096^3 with two LDS buffers; with S operands it is not faster than the
        // register-staged tile yet (598 vs 593), so only R x R takes this path.  At 512-1023 tiles (the decoder's
        // 3328x4096x1200: 832 tiles on 768 / 512 resident slots) 128x64 register staging stays ahead: 75 vs 89-98 us.
        // VS_GEMM_GLDS=0 disables, =2 forces all layouts.
        static const int glds_mode = getenv("VS_GEMM_GLDS") ? atoi(getenv("VS_GEMM_GLDS")) : 1;
        if (plan.bm == 128 && plan.bn == 128 && glds_mode && glds_operand_ok(A, lda, LA, M, K, epi.batch_a) &&
            glds_operand_ok(B, ldb, LB, N, K, epi.batch_b))
            return launch_glds<CT, LA, LB>(A, lda, B, ldb, M, N, K, plan, epi, slabs, stream);
    } else if constexpr (CT != VS_F32) {
        static const int glds_mode = getenv("VS_GEMM_GLDS") ? atoi(getenv("VS_GEMM_GLDS")) : 1;
        if (plan.bm == 128 && plan.bn == 128 && glds_mode == 2 && glds_operand_ok(A, lda, LA, M, K, epi.batch_a) &&
            glds_operand_ok(B, ldb, LB, N, K, epi.batch_b))
            return launch_glds<CT, LA, LB>(A, lda, B, ldb, M, N, K, plan, epi, slabs, stream);
    }
    if (plan.bm == 128 && plan.bn == 128) return launch<CT, LA, LB, 128, 128>(A, lda, B, ldb, M, N, K, plan, epi, slabs, stream);
    if (plan.bm == 128 && plan.bn == 64) return launch<CT, LA, LB, 128, 64>(A, lda, B, ldb, M, N, K, plan, epi, slabs, stream);
    if constexpr (CT != VS_F32)
        if (plan.bm == 64 && plan.bn == 128) return launch<CT, LA, LB, 64, 128>(A, lda, B, ldb, M, N, K, plan, epi, slabs, stream);
    return launch<CT, LA, LB, 64, 64>(A, lda, B, ldb, M, N, K, plan, epi, slabs, stream);
}

template <int CT>
int launch_layout(int la, int lb, const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K,
                  const Plan& plan, const Epi& epi, float* slabs, hipStream_t stream) {
    if (la == LR && lb == LR) return launch_tile<CT, LR, LR>(A, lda, B, ldb, M, N, K, plan, epi, slabs, stream);
    if (la == LR && lb == LS) return launch_tile<CT, LR, LS>(A, lda, B, ldb, M, N, K, plan, epi, slabs, stream);
    if (la == LS && lb == LR) return launch_tile<CT, LS, LR>(A, lda, B, ldb, M, N, K, plan, epi, slabs, stream);
    return launch_tile<CT, LS, LS>(A, lda, B, ldb, M, N, K, plan, epi, slabs, stream);
}

// ---- split-K arrival counters -------------------------------------------------------------------------------------------------
// One word per output tile of a split launch, zero between launches (the last workgroup to arrive clears it).  The words come from a static
// pool in device memory (no allocation behind the ABI), dealt out round robin: two launches share a word only if more than SK_POOL words
// were taken between them AND they are in flight at the same time; a recorded graph keeps the ranges it was captured with.
constexpr unsigned SK_POOL = 1u << 20;
}  // namespace
__device__ unsigned vs_sk_pool[1u << 20];         // (external linkage: hipGetSymbolAddress does not find a symbol of an unnamed namespace)
namespace {

// `tile_slab_bytes`: what the last workgroup of a tile has to read back (splits x tile x 4 B).  It reads at ~16-64 KB per us (one workgroup,
// dependent on the loads it keeps in flight), the reduce launch at TB/s: the fix-up pays for small tiles x few splits only (measured in the
// WaveEq step: 256 x 1200 x 20480 in 22 splits of 128 x 128 tiles, 1.4 MB per tile: 41 -> 115 us; 64 x 64 tiles x 4 splits: even or ahead).
// With the threshold the WaveEq step is still 15-25 us SLOWER (1.284 / 1.286 / 1.299 vs 1.268 / 1.268 / 1.275 ms, three interleaved pairs: the
// 64 x 64 kernel's slabs leave as 4-byte sc1 stores, one fabric write each), so the fix-up is OPT-IN: VS_GEMM_SPLITK_FUSED: 0 = never (default),
// 1 = up to SK_FUSED_MAX_BYTES per tile, 2 = always.
constexpr int64_t SK_FUSED_MAX_BYTES = 96 << 10;
unsigned* sk_take(int64_t words, int64_t tile_slab_bytes) {
    static unsigned* base = nullptr;
    static unsigned next = 0;
    const char* env = getenv("VS_GEMM_SPLITK_FUSED");              // read per call: tests switch it
    const int fused = env ? atoi(env) : 0;
    if (!fused || words <= 0 || words > SK_POOL / 4 || (fused == 1 && tile_slab_bytes > SK_FUSED_MAX_BYTES)) return nullptr;
    if (!base && hipGetSymbolAddress((void**)&base, HIP_SYMBOL(vs_sk_pool)) != hipSuccess) { base = nullptr; return nullptr; }
    if (next + words > SK_POOL) next = 0;
    unsigned* p = base + next;
    next += (unsigned)words;
    return p;
}

}  // namespace

extern "C" size_t vs_gemm_batched_workspace_bytes(int batch, int64_t M, int64_t N, int64_t K) {
    if (batch <= 0 || M <= 0 || N <= 0 || K <= 0) return 0;
    size_t worst = 0;
    for (int c = 0; c < 2; ++c) {                        // fp32 and the 16-bit types (bf16 and fp16 plan alike)
        Plan p = make_plan(c, M, N, K, batch);
        if (p.splits > 1) {
            size_t b = (size_t)batch * p.splits * (size_t)M * (size_t)N * sizeof(float);
            if (b > worst) worst = b;
        }
        const BigPlan bp = make_big_plan(c, M, N, K, batch);
        if (bp.use && bp.splits > 1) {
            size_t b = (size_t)batch * bp.splits * (size_t)M * (size_t)N * sizeof(float);
            if (b > worst) worst = b;
        }
        const MidPlan mp = make_mid_plan(c, M, N, K, batch);
        if (mp.use && mp.splits > 1) {
            size_t b = (size_t)batch * mp.splits * (size_t)M * (size_t)N * sizeof(float);
            if (b > worst) worst = b;
        }
        for (int lb = 0; lb < 2; ++lb) {
            const P8Plan pp = make_p8_plan(c, M, N, K, batch, lb);
            if (pp.use && pp.splits > 1) {
                size_t b = (size_t)batch * pp.splits * (size_t)M * (size_t)N * sizeof(float);
                if (b > worst) worst = b;
            }
        }
    }
    return worst;
}

// `batch` independent problems of one shape in one launch (the weight gradients of the integrator's blocks: the same three
// small GEMMs per block, each too small to fill the chip): problem i uses A + i*stride_a, B + i*stride_b, C + i*stride_c.
extern "C" int vs_gemm_batched(int compute, int batch, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, int64_t stride_a,
                               int layout_a, const void* B, int64_t ldb, int64_t stride_b, int layout_b, void* C, int64_t ldc,
                               int64_t stride_c, int c_dtype, float alpha, int accumulate, void* workspace, size_t workspace_bytes,
                               void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK_ARG(vs_dtype_ok(compute), "vs_gemm_batched: compute type %d", compute);
    VS_CHECK_ARG(batch >= 1 && batch <= 1024 && M > 0 && N > 0 && K > 0, "vs_gemm_batched: bad sizes");
    VS_CHECK_ARG(A && B && C, "vs_gemm_batched: null operand");
    VS_CHECK_ARG((layout_a == LR || layout_a == LS) && (layout_b == LR || layout_b == LS), "vs_gemm_batched: bad layout");
    VS_CHECK_ARG(vs_dtype_ok(c_dtype), "vs_gemm_batched: bad c_dtype");
    VS_CHECK_ARG(lda >= (layout_a == LR ? K : M) && ldb >= (layout_b == LR ? K : N) && ldc >= N, "vs_gemm_batched: leading dimension too small");
    Plan plan = make_plan(compute, M, N, K, batch);
    plan.batch = batch;
    const P8Plan pp = p8_plan_for(compute, M, N, K, batch, A, lda, layout_a, B, ldb, layout_b, stride_a, stride_b);
    BigPlan bp = big_plan_for(compute, M, N, K, batch, A, lda, layout_a, B, ldb, layout_b, stride_a, stride_b);
    if (pp.use) bp.use = false;
    MidPlan mp{false, 1, 0, 0, 0, 5};
    if (!bp.use && !pp.use) mp = mid_plan_for(compute, M, N, K, batch, A, lda, layout_a, B, ldb, layout_b, stride_a, stride_b);
    if (pp.use) { plan.splits = pp.splits; plan.k_tiles_per_split = pp.k_tiles_per_split; }
    if (bp.use) { plan.splits = bp.splits; plan.k_tiles_per_split = bp.k_tiles_per_split; }
    if (mp.use) { plan.splits = mp.splits; plan.k_tiles_per_split = mp.k_tiles_per_split; }
    Epi epi{C, ldc, c_dtype, alpha, nullptr, VS_ACT_NONE, nullptr, 0, 0, 0, accumulate, 0, 0, 0, 0, 0, 0, 0, 0, 0, plan.splits, stride_a, stride_b, stride_c};
    float* slabs = nullptr;
    if (plan.splits > 1) {
        const size_t need = (size_t)batch * plan.splits * (size_t)M * (size_t)N * sizeof(float);
        if (!workspace || workspace_bytes < need)
            return vs_fail(VS_ERR_WORKSPACE, "vs_gemm_batched: split-K needs %zu workspace bytes, got %zu", need, workspace_bytes);
        slabs = (float*)workspace;
        // one counter per (problem, tile) of the kernel that runs: 128-wide tiles on the ring kernel, plan.bm x plan.bn otherwise
        if (!bp.use && !pp.use && (mp.use || !(plan.bm == 128 && plan.bn == 128))) {      // (the 128x128 LDS-DMA tile never splits in practice: no fix-up there)
            if (need < (1ull << 31)) epi.sk_counters = sk_take(batch * (mp.use ? (int64_t)mp.tiles_m * mp.tiles_n : vs_cdiv(M, plan.bm) * vs_cdiv(N, plan.bn)),
                                                             (int64_t)plan.splits * (mp.use ? 128 * 128 : plan.bm * plan.bn) * 4);
            epi.sk_splits = plan.splits;
            epi.sk_bytes = (int64_t)need;
        }
    }
    int rc;
    if (pp.use)
        rc = compute == VS_BF16 ? launch_p8_layout<VS_BF16>(layout_a, layout_b, A, lda, B, ldb, M, N, K, pp, batch, epi, slabs, stream)
                                : launch_p8_layout<VS_F16>(layout_a, layout_b, A, lda, B, ldb, M, N, K, pp, batch, epi, slabs, stream);
    else if (bp.use)
        rc = compute == VS_BF16 ? launch_big_layout<VS_BF16>(layout_a, layout_b, A, lda, B, ldb, M, N, K, bp, batch, epi, slabs, stream)
                                : launch_big_layout<VS_F16>(layout_a, layout_b, A, lda, B, ldb, M, N, K, bp, batch, epi, slabs, stream);
    else if (mp.use)
        rc = compute == VS_BF16 ? launch_mid_layout<VS_BF16>(layout_a, layout_b, A, lda, B, ldb, M, N, K, mp, batch, epi, slabs, stream)
                                : launch_mid_layout<VS_F16>(layout_a, layout_b, A, lda, B, ldb, M, N, K, mp, batch, epi, slabs, stream);
    else
        rc = compute == VS_BF16  ? launch_layout<VS_BF16>(layout_a, layout_b, A, lda, B, ldb, M, N, K, plan, epi, slabs, stream)
             : compute == VS_F16 ? launch_layout<VS_F16>(layout_a, layout_b, A, lda, B, ldb, M, N, K, plan, epi, slabs, stream)
                                 : launch_layout<VS_F32>(layout_a, layout_b, A, lda, B, ldb, M, N, K, plan, epi, slabs, stream);
    if (rc != VS_OK) return rc;
    if (slabs && !epi.sk_counters) {
        int64_t blocks = vs_cdiv(M * N, 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks, (unsigned)batch), dim3(256), 0, stream, slabs, plan.splits, M, N, epi);
        VS_CHECK_LAUNCH("vs_gemm_batched split-K reduce");
    }
    return VS_OK;
}

// Weight gradient + optimizer in one launch: G = A * B^T (as vs_gemm) is the gradient of the fp32 parameter `param` [M, N]; the
// epilogue applies the Adam update (vs_adam_math.h: bitwise what vs_adam_multi computes from a stored G) to param / exp_avg /
// exp_avg_sq and rewrites the 16-bit operand copy.  The gradient never reaches HBM: per parameter 12 B read + 14 B written instead
// of 4 B (gradient store) + 30 B (optimizer pass).  16-bit operands that fit the LDS-DMA loader only (VS_ERR_UNSUPPORTED otherwise:
// the caller falls back to vs_gemm + vs_adam_multi); no split-K, whatever K.
extern "C" int vs_gemm_adam(int compute, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, int layout_a, const void* B, int64_t ldb,
                            int layout_b, float alpha, float* param, float* exp_avg, float* exp_avg_sq, void* shadow, int shadow_dtype,
                            const int32_t* step, int32_t skipped, double lr, double beta1, double beta2, double eps, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK_ARG(compute == VS_BF16 || compute == VS_F16, "vs_gemm_adam: 16-bit compute types only (%d)", compute);
    VS_CHECK_ARG(M > 0 && N > 0 && K > 0, "vs_gemm_adam: M, N, K must be positive");
    VS_CHECK_ARG(A && B && param && exp_avg && exp_avg_sq && step, "vs_gemm_adam: null pointer");
    VS_CHECK_ARG((layout_a == LR || layout_a == LS) && (layout_b == LR || layout_b == LS), "vs_gemm_adam: bad layout");
    VS_CHECK_ARG(lda >= (layout_a == LR ? K : M) && ldb >= (layout_b == LR ? K : N), "vs_gemm_adam: leading dimension too small");
    VS_CHECK_ARG(!shadow || shadow_dtype == VS_BF16 || shadow_dtype == VS_F16, "vs_gemm_adam: bad shadow dtype");
    if (!(glds_operand_ok(A, lda, layout_a, M, K, 0) && glds_operand_ok(B, ldb, layout_b, N, K, 0)) || lda >= (1ll << 23) || ldb >= (1ll << 23))
        return vs_fail(VS_ERR_UNSUPPORTED, "vs_gemm_adam: operands do not fit the LDS-DMA loader (16-byte alignment, multiples of 8)");
    Epi epi{param, N, VS_F32, alpha, nullptr, VS_ACT_NONE, nullptr, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    epi.adam_m = exp_avg; epi.adam_v = exp_avg_sq; epi.adam_shadow = (unsigned short*)shadow; epi.adam_shadow_dtype = shadow_dtype;
    epi.adam_step = step; epi.adam_skipped = skipped; epi.adam_guard = vs_g_exchange_guard;
    static const int adam_pipe = getenv("VS_ADAM_PIPE") ? atoi(getenv("VS_ADAM_PIPE")) : 1;      // 0: one row piece at a time; 1: four ahead; 3: + non-temporal state
    epi.adam_pipe = adam_pipe;
    epi.adam_lr = lr; epi.adam_beta1 = beta1; epi.adam_beta2 = beta2; epi.adam_eps = (float)eps;
    MidPlan mp{true, 1, vs_cdiv(K, BIG_BK), (int)vs_cdiv(M, 128), (int)vs_cdiv(N, 128), 5};
    if ((int64_t)mp.tiles_m * mp.tiles_n > 0x7fffffffll) return vs_fail(VS_ERR_UNSUPPORTED, "vs_gemm_adam: too many tiles");
    return compute == VS_BF16 ? launch_mid_layout<VS_BF16>(layout_a, layout_b, A, lda, B, ldb, M, N, K, mp, 1, epi, nullptr, stream)
                              : launch_mid_layout<VS_F16>(layout_a, layout_b, A, lda, B, ldb, M, N, K, mp, 1, epi, nullptr, stream);
}

// The decoder's last Linear layer with the frame losses in its epilogue (recorded MLP-family step): frames = act(A W^T + bias) is row
// r = (b, g) of the decoded stack [B, G, N = D] (reference: networks/mlp_encdec.py:43-50 + train.py:85-86, 139); instead of storing the
// 54 MB of fp32 frames and reading them back with the targets (vs_train_losses_fwd_grad), the 256 x 256 tile kernel compares its result
// with full[b, target(g)] while it is in registers: squared errors into one partial-sum pair per workgroup, k (y - target) act'(y) into
// `dz`.  A one-workgroup launch then adds the partials, computes the two code terms and their gradients and assembles `out` (layout of
// vs_train_losses_fwd_grad: [4] total, [5] ae, [6] zero-order, [7] pred, [8] t_reg).  VS_ERR_UNSUPPORTED when the problem does not run on
// the 256 x 256 tile (the caller stores the frames and calls vs_train_losses_fwd_grad instead).
extern "C" int vs_gemm_frame_loss(int compute, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* W, int64_t ldw,
                                  const float* bias, int act, const float* full, const int32_t* t_random_dev, int ae_shift, int first_forecast,
                                  int G, int T, const float* s_old, const float* s_new, int64_t n_s, const float* t0, int64_t Bt, int64_t Ct,
                                  int average_tloss, const float* lambdas, const float* grad_total, void* dz, int dz_dtype, float* ds_old,
                                  float* ds_new, float* dt0, float* out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK_ARG(compute == VS_BF16 || compute == VS_F16, "vs_gemm_frame_loss: 16-bit compute types only (%d)", compute);
    VS_CHECK_ARG(M > 0 && M < (1ll << 31) && N > 0 && K > 0 && G >= 1 && M % G == 0 && N % 4 == 0, "vs_gemm_frame_loss: bad sizes (rows = B * G, N %% 4 == 0)");
    VS_CHECK_ARG(!bias || ((uintptr_t)bias & 15) == 0, "vs_gemm_frame_loss: the bias must be 16-byte aligned");
    VS_CHECK_ARG(A && W && full && t_random_dev && grad_total && dz && dt0 && out && (n_s == 0 || (ds_old && ds_new)), "vs_gemm_frame_loss: null pointer");
    VS_CHECK_ARG(lda >= K && ldw >= K, "vs_gemm_frame_loss: leading dimension too small");
    VS_CHECK_ARG(vs_dtype_ok(dz_dtype) && act >= VS_ACT_NONE && act <= VS_ACT_ELU, "vs_gemm_frame_loss: bad dz dtype / activation");
    LossArgs a;
    int rc = fill_loss_args(a, nullptr, full, nullptr, t_random_dev, ae_shift, first_forecast, M / G, G, T, N, s_old, s_new, n_s, t0, Bt, Ct,
                            average_tloss, lambdas);
    if (rc != VS_OK) return rc;
    const P8Plan pp = p8_plan_for(compute, M, N, K, 1, A, lda, LR, W, ldw, LR, 0, 0);
    const bool p8 = pp.use && pp.ni == 2 && pp.splits == 1 && (int64_t)pp.tiles_m * pp.tiles_n <= VS_LOSS_MAX_PARTIALS;
    const BigPlan bp = big_plan_for(compute, M, N, K, 1, A, lda, LR, W, ldw, LR, 0, 0);
    if (!p8 && (!bp.use || bp.splits != 1 || (int64_t)bp.tiles_m * bp.tiles_n > VS_LOSS_MAX_PARTIALS))
        return vs_fail(VS_ERR_UNSUPPORTED, "vs_gemm_frame_loss: the problem does not run on the 256x256 tile kernels");
    Epi epi{nullptr, N, VS_F32, 1.f, bias, act, nullptr, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    epi.fl_full = full; epi.fl_tdev = t_random_dev; epi.fl_ae_shift = ae_shift; epi.fl_first = first_forecast; epi.fl_G = G; epi.fl_T = T;
    epi.fl_up = grad_total; epi.fl_l_ae = a.l_ae; epi.fl_l_pred = a.l_pred; epi.fl_inv_ae = a.inv_ae; epi.fl_inv_pred = a.inv_pred;
    epi.fl_dz = dz; epi.fl_dz_dtype = dz_dtype; epi.fl_partials = out + 16;
    if (p8 && act == VS_ACT_SIGMOID)
        rc = compute == VS_BF16 ? launch_p8<VS_BF16, LR, LR, 2, 2>(A, lda, W, ldw, M, N, K, pp, 1, epi, nullptr, stream)
                                : launch_p8<VS_F16, LR, LR, 2, 2>(A, lda, W, ldw, M, N, K, pp, 1, epi, nullptr, stream);
    else if (p8)
        rc = compute == VS_BF16 ? launch_p8<VS_BF16, LR, LR, 2, 1>(A, lda, W, ldw, M, N, K, pp, 1, epi, nullptr, stream)
                                : launch_p8<VS_F16, LR, LR, 2, 1>(A, lda, W, ldw, M, N, K, pp, 1, epi, nullptr, stream);
    else
        rc = compute == VS_BF16 ? launch_big<VS_BF16, LR, LR, true>(A, lda, W, ldw, M, N, K, bp, 1, epi, nullptr, stream)
                                : launch_big<VS_F16, LR, LR, true>(A, lda, W, ldw, M, N, K, bp, 1, epi, nullptr, stream);
    if (rc != VS_OK) return rc;
    LossGrads gr{grad_total, dz, dz_dtype, act, ds_old, ds_new, dt0};
    hipLaunchKernelGGL(frame_loss_finish_kernel, dim3(1), dim3(256), 0, stream, a, out, p8 ? pp.tiles_m * pp.tiles_n : bp.tiles_m * bp.tiles_n, gr);
    VS_CHECK_LAUNCH("vs_gemm_frame_loss");
    return VS_OK;
}

extern "C" size_t vs_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    size_t worst = 0;
    for (int c = 0; c < 2; ++c) {
        Plan p = make_plan(c, M, N, K);
        if (p.splits > 1) {
            size_t b = (size_t)p.splits * (size_t)M * (size_t)N * sizeof(float);
            if (b > worst) worst = b;
        }
        const BigPlan bp = make_big_plan(c, M, N, K, 1);
        if (bp.use && bp.splits > 1) {
            size_t b = (size_t)bp.splits * (size_t)M * (size_t)N * sizeof(float);
            if (b > worst) worst = b;
        }
        const MidPlan mp = make_mid_plan(c, M, N, K, 1);
        if (mp.use && mp.splits > 1) {
            size_t b = (size_t)mp.splits * (size_t)M * (size_t)N * sizeof(float);
            if (b > worst) worst = b;
        }
        for (int lb = 0; lb < 2; ++lb) {
            const P8Plan pp = make_p8_plan(c, M, N, K, 1, lb);
            if (pp.use && pp.splits > 1) {
                size_t b = (size_t)pp.splits * (size_t)M * (size_t)N * sizeof(float);
                if (b > worst) worst = b;
            }
        }
    }
    return worst;
}

extern "C" int vs_gemm(int compute, int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, int layout_a, const void* B,
                       int64_t ldb, int layout_b, void* C, int64_t ldc, int c_dtype, float alpha, const float* bias, int act,
                       const void* mask, int64_t ldmask, int mask_dtype, int mask_act, int accumulate, void* workspace,
                       size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VS_CHECK_ARG(vs_dtype_ok(compute), "vs_gemm: compute type %d", compute);
    VS_CHECK_ARG(M > 0 && N > 0 && K > 0, "vs_gemm: M, N, K must be positive (%lld %lld %lld)", (long long)M, (long long)N, (long long)K);
    VS_CHECK_ARG(A && B && C, "vs_gemm: null operand");
    VS_CHECK_ARG((layout_a == LR || layout_a == LS) && (layout_b == LR || layout_b == LS), "vs_gemm: bad layout");
    VS_CHECK_ARG(vs_dtype_ok(c_dtype) && (!mask || vs_dtype_ok(mask_dtype)), "vs_gemm: bad c_dtype / mask_dtype");
    VS_CHECK_ARG(lda >= (layout_a == LR ? K : M) && ldb >= (layout_b == LR ? K : N) && ldc >= N, "vs_gemm: leading dimension too small");
    VS_CHECK_ARG(!mask || ldmask >= N, "vs_gemm: ldmask too small");
    VS_CHECK_ARG(act >= VS_ACT_NONE && act <= VS_ACT_ELU, "vs_gemm: bad activation");
    Epi epi{C, ldc, c_dtype, alpha, bias, act, mask, ldmask, mask_dtype, mask_act, accumulate, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    Plan plan = make_plan(compute, M, N, K, 1, layout_a == LR && layout_b == LR);
    const P8Plan pp = p8_plan_for(compute, M, N, K, 1, A, lda, layout_a, B, ldb, layout_b, 0, 0);
    BigPlan bp = big_plan_for(compute, M, N, K, 1, A, lda, layout_a, B, ldb, layout_b, 0, 0);
    if (pp.use) bp.use = false;
    MidPlan mp{false, 1, 0, 0, 0, 5};
    if (!bp.use && !pp.use) mp = mid_plan_for(compute, M, N, K, 1, A, lda, layout_a, B, ldb, layout_b, 0, 0);
    if (pp.use) { plan.splits = pp.splits; plan.k_tiles_per_split = pp.k_tiles_per_split; }
    if (bp.use) { plan.splits = bp.splits; plan.k_tiles_per_split = bp.k_tiles_per_split; }
    if (mp.use) { plan.splits = mp.splits; plan.k_tiles_per_split = mp.k_tiles_per_split; }
    float* slabs = nullptr;
    if (plan.splits > 1) {
        const size_t need = (size_t)plan.splits * (size_t)M * (size_t)N * sizeof(float);
        if (!workspace || workspace_bytes < need)
            return vs_fail(VS_ERR_WORKSPACE, "vs_gemm: split-K needs %zu workspace bytes, got %zu", need, workspace_bytes);
        slabs = (float*)workspace;
        if (!bp.use && !pp.use && (mp.use || !(plan.bm == 128 && plan.bn == 128))) {
            if (need < (1ull << 31)) epi.sk_counters = sk_take(mp.use ? (int64_t)mp.tiles_m * mp.tiles_n : vs_cdiv(M, plan.bm) * vs_cdiv(N, plan.bn),
                                                             (int64_t)plan.splits * (mp.use ? 128 * 128 : plan.bm * plan.bn) * 4);
            epi.sk_splits = plan.splits;
            epi.sk_bytes = (int64_t)need;
        }
    }
    int rc;
    if (pp.use)
        rc = compute == VS_BF16 ? launch_p8_layout<VS_BF16>(layout_a, layout_b, A, lda, B, ldb, M, N, K, pp, 1, epi, slabs, stream)
                                : launch_p8_layout<VS_F16>(layout_a, layout_b, A, lda, B, ldb, M, N, K, pp, 1, epi, slabs, stream);
    else if (bp.use)
        rc = compute == VS_BF16 ? launch_big_layout<VS_BF16>(layout_a, layout_b, A, lda, B, ldb, M, N, K, bp, 1, epi, slabs, stream)
                                : launch_big_layout<VS_F16>(layout_a, layout_b, A, lda, B, ldb, M, N, K, bp, 1, epi, slabs, stream);
    else if (mp.use)
        rc = compute == VS_BF16 ? launch_mid_layout<VS_BF16>(layout_a, layout_b, A, lda, B, ldb, M, N, K, mp, 1, epi, slabs, stream)
                                : launch_mid_layout<VS_F16>(layout_a, layout_b, A, lda, B, ldb, M, N, K, mp, 1, epi, slabs, stream);
    else
        rc = compute == VS_BF16  ? launch_layout<VS_BF16>(layout_a, layout_b, A, lda, B, ldb, M, N, K, plan, epi, slabs, stream)
             : compute == VS_F16 ? launch_layout<VS_F16>(layout_a, layout_b, A, lda, B, ldb, M, N, K, plan, epi, slabs, stream)
                                 : launch_layout<VS_F32>(layout_a, layout_b, A, lda, B, ldb, M, N, K, plan, epi, slabs, stream);
    if (rc != VS_OK) return rc;
    if (slabs && !epi.sk_counters) {
        int64_t blocks = vs_cdiv(M * N, 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, slabs, plan.splits, M, N, epi);
        VS_CHECK_LAUNCH("vs_gemm split-K reduce");
    }
    return VS_OK;
}
