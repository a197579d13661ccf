// p8_bench.hip -- standalone correctness + timing harness of the 256 x (128 NI) "8 phases" GEMM tile (csrc/vs_gemm_p8.h).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/p8_bench.hip -o tools/probes/p8_bench
//   p8_bench M N K la lb ni [mode]      la / lb: 0 = R, 1 = S; ni: 1 | 2 | 12 (256x128 | 256x256 | 128x128); mode: int = exact small-integer operands
//                                        (bit-exact against a naive kernel, repeated), rand = uniform [-1,1) operands (timing + tolerance),
//                                        cold = rand with operand sets rotating through > 256 MB
// Prints one line per run; exit code 1 on a mismatch.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../spatiotemporal_variable_separation_amd/csrc/vs_gemm_p8.h"

static int g_out16 = 0;
thread_local char vs_err_buf[256];
int vs_fail(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); return code; }
unsigned* vs_g_exchange_guard = nullptr;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }

__global__ void ref_kernel(const unsigned short* A, int64_t lda, int la, const unsigned short* B, int64_t ldb, int lb, float* C, int64_t M, int64_t N, int64_t K) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N || m >= M) return;
    float s = 0.f;
    for (int64_t k = 0; k < K; ++k) {
        const float a = __uint_as_float((unsigned)(la == 0 ? A[m * lda + k] : A[k * lda + m]) << 16);
        const float b = __uint_as_float((unsigned)(lb == 0 ? B[n * ldb + k] : B[k * ldb + n]) << 16);
        s = fmaf(a, b, s);
    }
    C[m * N + n] = s;
}

template <int LA, int LB, int NI, int MI = 4>
static void launch(const unsigned short* A, int64_t lda, const unsigned short* B, int64_t ldb, float* C, int64_t M, int64_t N, int64_t K, hipStream_t st) {
    auto kfn = gemm_p8_kernel<VS_BF16, LA, LB, NI, false, 0, MI>;
    constexpr int lds = 2 * (2 * 32 * MI * 64 * 2 + 2 * 64 * NI * 64 * 2);
    static bool set = false;
    if (!set) { CK(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); set = true; }
    Epi epi{};
    epi.C = C; epi.ldc = N; epi.c_dtype = VS_F32; epi.alpha = 1.f;
    if (g_out16) {                          // timing of the 16-bit stores (C is large enough for either); 2: with a 16-bit relu mask
        epi.c_dtype = VS_BF16;
        if (g_out16 == 2) {
            static unsigned short* mask = nullptr;
            if (!mask) { CK(hipMalloc(&mask, (size_t)M * N * 2)); CK(hipMemset(mask, 0x3f, (size_t)M * N * 2)); }
            epi.mask = mask; epi.ldmask = N; epi.mask_dtype = VS_BF16; epi.mask_act = VS_ACT_RELU;
        }
    }
    const int tm = (int)((M + 64 * MI - 1) / (64 * MI)), tn = (int)((N + 128 * NI - 1) / (128 * NI));
    hipLaunchKernelGGL(kfn, dim3(tm * tn), dim3(512), lds, st, A, lda, B, ldb, M, N, K, (int)((K + 63) / 64), tn, epi, (float*)nullptr);
}

// the frame-loss epilogue (vs_gemm_frame_loss) on a linear frame map: row m = (b, g) of [M / 26, 26, N] against target row m; nothing is checked
// but the time (the library's tests hold its values)
static void launch_loss(const unsigned short* A, int64_t lda, const unsigned short* B, int64_t ldb, float* C, int64_t M, int64_t N, int64_t K, hipStream_t st) {
    auto kfn = g_out16 == 3 ? gemm_p8_kernel<VS_BF16, LR, LR, 2, false, 2, 4> : gemm_p8_kernel<VS_BF16, LR, LR, 2, false, 1, 4>;
    constexpr int lds = 2 * (2 * 32 * 4 * 64 * 2 + 2 * 64 * 2 * 64 * 2);
    static int set = 0;
    static float *target, *partials, *up;
    static int* tdev;
    static unsigned short* dz;
    const int tm = (int)((M + 255) / 256), tn = (int)((N + 255) / 256);
    if (!set) {
        CK(hipFuncSetAttribute((const void*)gemm_p8_kernel<VS_BF16, LR, LR, 2, false, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        CK(hipFuncSetAttribute((const void*)gemm_p8_kernel<VS_BF16, LR, LR, 2, false, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        CK(hipMalloc(&target, (size_t)M * N * 4)); CK(hipMemset(target, 0, (size_t)M * N * 4));
        CK(hipMalloc(&dz, (size_t)M * N * 2));
        CK(hipMalloc(&partials, (size_t)tm * tn * 8));
        CK(hipMalloc(&up, 4)); const float one = 1.f; CK(hipMemcpy(up, &one, 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&tdev, 4)); CK(hipMemset(tdev, 0, 4));
        set = 1;
    }
    Epi epi{};
    epi.C = C; epi.ldc = N; epi.c_dtype = VS_F32; epi.alpha = 1.f;
    epi.fl_full = target; epi.fl_tdev = tdev; epi.fl_ae_shift = 0; epi.fl_first = 1; epi.fl_G = 26; epi.fl_T = 26;
    epi.fl_up = up; epi.fl_l_ae = 1.f; epi.fl_l_pred = 1.f; epi.fl_inv_ae = 1.f; epi.fl_inv_pred = 1.f;
    epi.fl_dz = dz; epi.fl_dz_dtype = VS_BF16; epi.fl_partials = partials;
    if (g_out16 == 3) epi.act = VS_ACT_SIGMOID;
    hipLaunchKernelGGL(kfn, dim3(tm * tn), dim3(512), lds, st, A, lda, B, ldb, M, N, K, (int)((K + 63) / 64), tn, epi, (float*)nullptr);
}

typedef void (*launch_fn)(const unsigned short*, int64_t, const unsigned short*, int64_t, float*, int64_t, int64_t, int64_t, hipStream_t);

int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage: p8_bench M N K la lb ni [int|rand|cold] [iters]\n"); return 2; }
    const int64_t M = atoll(argv[1]), N = atoll(argv[2]), K = atoll(argv[3]);
    const int la = atoi(argv[4]), lb = atoi(argv[5]), ni = atoi(argv[6]);
    const char* mode = argc > 7 ? argv[7] : "rand";
    const int iters = argc > 8 ? atoi(argv[8]) : 20;
    launch_fn fn = nullptr;
    if (ni == 2) {
        if (la == 0 && lb == 0) fn = launch<LR, LR, 2>;
        if (la == 0 && lb == 1) fn = launch<LR, LS, 2>;
        if (la == 1 && lb == 0) fn = launch<LS, LR, 2>;
        if (la == 1 && lb == 1) fn = launch<LS, LS, 2>;
    } else if (ni == 12) {                        // 128 x 128 (MI = 2, NI = 1)
        if (la == 0 && lb == 0) fn = launch<LR, LR, 1, 2>;
        if (la == 0 && lb == 1) fn = launch<LR, LS, 1, 2>;
        if (la == 1 && lb == 0) fn = launch<LS, LR, 1, 2>;
        if (la == 1 && lb == 1) fn = launch<LS, LS, 1, 2>;
    } else {
        if (la == 0 && lb == 0) fn = launch<LR, LR, 1>;
        if (la == 0 && lb == 1) fn = launch<LR, LS, 1>;
        if (la == 1 && lb == 0) fn = launch<LS, LR, 1>;
        if (la == 1 && lb == 1) fn = launch<LS, LS, 1>;
    }
    if (!strcmp(mode, "h16")) g_out16 = 1;
    if (!strcmp(mode, "h16m")) g_out16 = 2;
    if (!strcmp(mode, "loss_sigmoid")) g_out16 = 3;
    const bool loss = !strcmp(mode, "loss") || g_out16;          // (time only)
    if (!strcmp(mode, "loss") || g_out16 == 3)          // ni = 2, R x R, M a multiple of 26: time only
        fn = launch_loss;
    if (!fn) { fprintf(stderr, "unsupported layout / ni\n"); return 2; }
    const bool exact = !strcmp(mode, "int");
    const bool cold = !strcmp(mode, "cold");
    const int64_t lda = la == 0 ? K : M, ldb = lb == 0 ? K : N;
    const size_t ea = (size_t)M * K, eb = (size_t)N * K, ec = (size_t)M * N;
    const size_t per_set = (ea + eb) * 2 + ec * 4;
    const int nset = cold ? (int)std::max<size_t>(1, std::min<size_t>(64, 600000000ull / per_set)) : 1;
    std::vector<unsigned short> ha(ea), hb(eb);
    unsigned long long s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (auto& v : ha) v = exact ? f2bf((float)((int)(rnd() % 5) - 2)) : f2bf((float)(rnd() % 65536) / 32768.f - 1.f);
    for (auto& v : hb) v = exact ? f2bf((float)((int)(rnd() % 5) - 2)) : f2bf((float)(rnd() % 65536) / 32768.f - 1.f);
    std::vector<unsigned short*> dA(nset), dB(nset);
    std::vector<float*> dC(nset);
    for (int i = 0; i < nset; ++i) {
        CK(hipMalloc(&dA[i], ea * 2)); CK(hipMalloc(&dB[i], eb * 2)); CK(hipMalloc(&dC[i], ec * 4));
        CK(hipMemcpy(dA[i], ha.data(), ea * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB[i], hb.data(), eb * 2, hipMemcpyHostToDevice));
        CK(hipMemset(dC[i], 0xff, ec * 4));
    }
    float* dR;
    CK(hipMalloc(&dR, ec * 4));
    hipLaunchKernelGGL(ref_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)M), dim3(256), 0, 0, dA[0], lda, la, dB[0], ldb, lb, dR, M, N, K);
    CK(hipDeviceSynchronize());
    std::vector<float> hr(ec), hc(ec);
    CK(hipMemcpy(hr.data(), dR, ec * 4, hipMemcpyDeviceToHost));
    int bad_runs = 0;
    const int checks = exact ? iters : 1;
    for (int c = 0; c < checks; ++c) {
        CK(hipMemset(dC[0], 0xff, ec * 4));
        fn(dA[0], lda, dB[0], ldb, dC[0], M, N, K, 0);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hc.data(), dC[0], ec * 4, hipMemcpyDeviceToHost));
        size_t bad = 0, first = 0;
        double worst = 0;
        for (size_t i = 0; i < ec; ++i) {
            const double d = fabs((double)hc[i] - (double)hr[i]);
            const bool b = exact ? (hc[i] != hr[i]) : !(d <= 1e-3 * (1.0 + fabs((double)hr[i])) + 2e-5 * K);
            if (d > worst) worst = d;
            if (b && !bad++) first = i;
        }
        if (loss) break;
        if (bad) {
            ++bad_runs;
            printf("MISMATCH run %d: %zu of %zu elements, first at (%zu, %zu): got %g want %g, worst |d| %g\n", c, bad, ec, first / N, first % N, hc[first], hr[first], worst);
            if (bad_runs >= 3) break;
        } else if (c == 0) printf("check ok (%s): worst |d| %g\n", exact ? "bit-exact integers" : "tolerance", worst);
    }
#ifdef P8_STAMPS
    {   // seams of workgroup 0, in us since its first instruction: prepare done | prologue landed | K loop done | epilogue issued | stores retired
        for (int rep = 0; rep < 4; ++rep) {
            if (rep == 0 || rep == 2)      // something else through the instruction caches first
                hipLaunchKernelGGL(ref_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)std::min<int64_t>(M, 64)), dim3(256), 0, 0, dA[0], lda, la, dB[0], ldb, lb, dR, std::min<int64_t>(M, 64), N, K);
            CK(hipDeviceSynchronize());
            fn(dA[0], lda, dB[0], ldb, dC[0], M, N, K, 0);
            CK(hipDeviceSynchronize());
            long long st[8];
            CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(p8_stamps), sizeof(st)));
            printf("stamps%s: prepare %.2f  prologue %.2f  k-loop %.2f  epilogue-issue %.2f  stores-retired %.2f us\n", (rep == 0 || rep == 2) ? " (after another kernel)" : " (repeat)          ",
                   (st[1] - st[0]) * 0.01, (st[2] - st[0]) * 0.01, (st[3] - st[0]) * 0.01, (st[4] - st[0]) * 0.01, (st[5] - st[0]) * 0.01);
        }
    }
#endif
    // timing
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < std::max(3, nset); ++i) fn(dA[i % nset], lda, dB[i % nset], ldb, dC[i % nset], M, N, K, 0);
    CK(hipDeviceSynchronize());
    const int reps = std::max(iters, nset);
    float best = 1e30f, tot = 0;
    for (int round = 0; round < 5; ++round) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) fn(dA[i % nset], lda, dB[i % nset], ldb, dC[i % nset], M, N, K, 0);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        if (ms < best) best = ms;
        tot += ms;
    }
    const double fl = 2.0 * M * N * K;
    printf("p8 M=%lld N=%lld K=%lld la=%d lb=%d ni=%d %s: best %.1f us %.1f TF/s, mean %.1f us %.1f TF/s%s\n", (long long)M, (long long)N, (long long)K, la, lb, ni, mode,
           best * 1e3, fl / best / 1e9, tot / 5 * 1e3, fl / (tot / 5) / 1e9, bad_runs ? "  ** WRONG **" : "");
    return bad_runs ? 1 : 0;
}
