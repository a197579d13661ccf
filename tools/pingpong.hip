// Latency of the epoch-tagged granule exchange between workgroups (agent-scope relaxed 8-byte atomics through L2/fabric).
// hipcc --offload-arch=gfx950 -O3 tools/pingpong.hip -o /tmp/pingpong && /tmp/pingpong
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned long long u64;

__device__ inline void put(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline u64 get(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// `nwg` workgroups launched; only wg `a` and wg `b` take part.  Each hop: the sender writes 512 granules (16x32 values, as
// the rollout publishes), the receiver polls all of them.  `hops` round trips.
__global__ void pingpong(u64* buf, int a, int b, int hops, int ngran, int* xcc, long long* cycles) {
    const int wg = blockIdx.x;
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[wg] = (int)(id & 0xf);
    }
    if (wg != a && wg != b) return;
    const int me = wg == a ? 0 : 1;
    u64* mine = buf + (size_t)me * 2 * ngran;        // double buffered
    const u64* theirs = buf + (size_t)(1 - me) * 2 * ngran;
    long long t0 = wall_clock64();
    for (int e = 1; e <= hops; ++e) {
        const int par = e & 1;
        if (me == 0) {
            for (int i = threadIdx.x; i < ngran; i += blockDim.x) put(mine + par * ngran + i, ((u64)e << 32) | (unsigned)i);
            for (int i = threadIdx.x; i < ngran; i += blockDim.x) {
                u64 v; int spin = 0;
                do { v = get(theirs + par * ngran + i); } while ((v >> 32) != (u64)e && ++spin < (1 << 22));
            }
        } else {
            for (int i = threadIdx.x; i < ngran; i += blockDim.x) {
                u64 v; int spin = 0;
                do { v = get(theirs + par * ngran + i); } while ((v >> 32) != (u64)e && ++spin < (1 << 22));
            }
            for (int i = threadIdx.x; i < ngran; i += blockDim.x) put(mine + par * ngran + i, ((u64)e << 32) | (unsigned)i);
        }
        __syncthreads();
    }
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) cycles[me] = t1 - t0;
}

int main() {
    const int nwg = 64, ngran = 512, hops = 2000;
    u64* buf; int* xcc; long long* cyc;
    hipMalloc(&buf, sizeof(u64) * 4 * ngran);
    hipMalloc(&xcc, sizeof(int) * nwg);
    hipMalloc(&cyc, sizeof(long long) * 2);
    int rate_khz = 0;
    hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    int pairs[][2] = {{0, 1}, {0, 8}, {0, 16}, {0, 2}, {1, 9}, {0, 32}};
    for (auto& pr : pairs) {
        for (int threads : {64, 256}) {
            hipMemset(buf, 0, sizeof(u64) * 4 * ngran);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(pingpong, dim3(nwg), dim3(threads), 0, 0, buf, pr[0], pr[1], hops, ngran, xcc, cyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            int hx[nwg]; hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost);
            printf("wg %d (xcc %d) <-> wg %d (xcc %d), %d threads, %d granules: %.3f us per one-way hop (kernel %.3f ms)\n", pr[0], hx[pr[0]], pr[1],
                   hx[pr[1]], threads, ngran, ms * 1e3 / hops / 2, ms);
        }
    }
    // small payload: 64 granules
    for (auto& pr : pairs) {
        hipMemset(buf, 0, sizeof(u64) * 4 * ngran);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(pingpong, dim3(nwg), dim3(64), 0, 0, buf, pr[0], pr[1], hops, 64, xcc, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("wg %d <-> wg %d, 64 threads, 64 granules: %.3f us per one-way hop\n", pr[0], pr[1], ms * 1e3 / hops / 2);
    }
    return 0;
}
