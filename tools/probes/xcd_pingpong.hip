// Ping-pong latency between two workgroups of ONE XCD (blocks 0 and 8 of a 16-block launch) and of two XCDs (blocks 0 and 1) for the cache
// policies an in-kernel exchange could use on gfx950: which (store, load) pairs make a value written by one CU visible to a polling CU, and
// how long a hop takes.  Every wait is bounded; a policy that never becomes visible reports "stuck".  Build + run:
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/xcd_pingpong tools/probes/xcd_pingpong.hip && /tmp/xcd_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef unsigned long long u64;

template <int LP>
__device__ __forceinline__ u64 ld(const u64* p) {
    u64 r;
    if constexpr (LP == 0) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    else if constexpr (LP == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    else if constexpr (LP == 2) asm volatile("buffer_inv sc0\n\tglobal_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    else if constexpr (LP == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 nt\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    else if constexpr (LP == 4) asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    else if constexpr (LP == 5) asm volatile("buffer_inv sc1\n\tglobal_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    return r;
}
template <int SP>
__device__ __forceinline__ void st(u64* p, u64 v) {
    if constexpr (SP == 0) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if constexpr (SP == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    else if constexpr (SP == 2) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// blocks a and b bounce a counter `rounds` times; out[0] = clocks (100 MHz) of block a, out[1] = 1 if a wait ran into the limit,
// out[2], out[3] = XCC ids of the two blocks
template <int SP, int LP>
__global__ void pingpong(u64* flag, int a, int b, int rounds, u64* out) {
    const int me = (int)blockIdx.x;
    if (me != a && me != b) return;
    if (threadIdx.x != 0) return;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[me == a ? 2 : 3] = xcc & 0xf;
    u64* mine = flag + (me == a ? 0 : 16);          // (separate 128-byte lines)
    u64* theirs = flag + (me == a ? 16 : 0);
    const u64 t0 = wall_clock64();
    bool stuck = false;
    for (int r = 1; r <= rounds && !stuck; ++r) {
        if (me == a) st<SP>(theirs, (u64)r);
        int spins = 0;
        while (ld<LP>(mine) < (u64)r) {
            if (++spins > 200000) { stuck = true; break; }
        }
        if (me == b) st<SP>(theirs, (u64)r);
    }
    if (stuck) {                                      // release the partner through the always-visible path
        st<0>(theirs, (u64)1 << 40);
        out[1] = 1;
    }
    if (me == a) out[0] = wall_clock64() - t0;
}

template <int SP, int LP>
void run(const char* name, u64* flag, u64* out, int a, int b) {
    const int rounds = 2000;
    hipMemset(flag, 0, 256);
    hipMemset(out, 0, 64);
    hipLaunchKernelGGL((pingpong<SP, LP>), dim3(16), dim3(64), 0, 0, flag, a, b, rounds, out);
    hipDeviceSynchronize();
    u64 h[4];
    hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    if (h[1]) printf("%-34s blocks %d,%d (xcc %llu,%llu): stuck (never visible)\n", name, a, b, h[2], h[3]);
    else printf("%-34s blocks %d,%d (xcc %llu,%llu): %.0f ns per hop\n", name, a, b, h[2], h[3], (double)h[0] * 10.0 / (2.0 * rounds));
}

int main() {
    u64 *flag, *out;
    hipMalloc(&flag, 256);
    hipMalloc(&out, 64);
    for (int pair = 0; pair < 2; ++pair) {
        const int a = 0, b = pair == 0 ? 8 : 1;
        run<0, 0>("st sc1 / ld sc1 (agent)", flag, out, a, b);
        run<3, 6>("st sc0 sc1 / ld sc0 sc1 (system)", flag, out, a, b);
        run<1, 1>("st sc0 / ld sc0", flag, out, a, b);
        run<1, 2>("st sc0 / inv sc0 + ld sc0", flag, out, a, b);
        run<2, 2>("st plain / inv sc0 + ld sc0", flag, out, a, b);
        run<1, 3>("st sc0 / ld sc0 nt", flag, out, a, b);
        run<2, 4>("st plain / ld nt", flag, out, a, b);
        run<2, 5>("st plain / inv sc1 + ld plain", flag, out, a, b);
        run<1, 0>("st sc0 / ld sc1", flag, out, a, b);
        run<2, 0>("st plain / ld sc1", flag, out, a, b);
        run<0, 2>("st sc1 / inv sc0 + ld sc0", flag, out, a, b);
    }
    return 0;
}
