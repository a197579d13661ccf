// Throughput/latency of agent-scope 8-byte atomic loads as the rollout's consumers issue them:
// nwg workgroups x 256 threads, each thread loads `per` granules per round from its slab's exchange area.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned long long u64;
__device__ inline u64 get(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int PER>
__global__ void xload(const u64* buf, int nslabs, int rounds, u64* sink, long long* cyc) {
    const int slab = blockIdx.x % nslabs;
    const u64* base = buf + (size_t)slab * PER * 512;
    u64 acc = 0;
    long long t0 = wall_clock64();
    for (int r = 0; r < rounds; ++r) {
        u64 v[PER];
#pragma unroll
        for (int s = 0; s < PER; ++s) v[s] = get(base + s * 512 + ((threadIdx.x + r) & 511));
#pragma unroll
        for (int s = 0; s < PER; ++s) acc += v[s];
    }
    long long t1 = wall_clock64();
    if (acc == 12345) sink[0] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int PER>
void run(int nwg, int nslabs, const u64* buf, u64* sink, long long* cyc) {
    const int rounds = 2000;
    hipLaunchKernelGGL(xload<PER>, dim3(nwg), dim3(256), 0, 0, buf, nslabs, 10, sink, cyc);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(xload<PER>, dim3(nwg), dim3(256), 0, 0, buf, nslabs, rounds, sink, cyc);
    (void)hipDeviceSynchronize();
    long long h[512];
    (void)hipMemcpy(h, cyc, sizeof(long long) * nwg, hipMemcpyDeviceToHost);
    double avg = 0;
    for (int i = 0; i < nwg; ++i) avg += (double)h[i];
    avg /= nwg;
    printf("%3d WGs x 256 thr, %2d loads/thread/round (%5.1f KB/WG/round): %.3f us per round, aggregate %.2f TB/s\n", nwg, PER, PER * 256 * 8 / 1024.0,
           avg * 0.01 / rounds, (double)nwg * PER * 256 * 8 / (avg * 1e-8 / rounds) / 1e12);
}
int main() {
    u64 *buf, *sink; long long* cyc;
    (void)hipMalloc(&buf, 8 * 18 * 512 * 8 * 2);
    (void)hipMemset(buf, 0, 8 * 18 * 512 * 8 * 2);
    (void)hipMalloc(&sink, 8); (void)hipMalloc(&cyc, 8 * 512);
    for (int nwg : {1, 8, 32, 64, 128}) {
        run<2>(nwg, 8, buf, sink, cyc);
        run<10>(nwg, 8, buf, sink, cyc);
        run<18>(nwg, 8, buf, sink, cyc);
    }
    return 0;
}
