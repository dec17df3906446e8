// mb_xwg.hip — cross-workgroup (cross-XCD) hand-off through global memory inside one launch:
// is data written with agent-scope relaxed atomic stores + a flag seen by agent-scope loads of another
// workgroup that has an older copy of the same lines in its own L2?  And what does one hand-off cost?
// WG 0 and WG 1 (different XCDs: workgroups are dealt round-robin to the 8 XCDs) ping-pong an 8 KB tile.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ROUNDS 200
__device__ __forceinline__ double ld_dev(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int wait_flag(const int* f, int want)
{
    int spins = 0;
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && spins < (1 << 22)) { __builtin_amdgcn_s_sleep(2); spins++; }
    return spins < (1 << 22);
}
__global__ __launch_bounds__(256) void k_pingpong(double* tile, int* flag, int* bad, unsigned long long* cyc, int partner)
{
    const int me = blockIdx.x;
    if (me != 0 && me != partner) return;
    const int who = (me == 0) ? 0 : 1;
    const int tid = threadIdx.x;
    __shared__ int ok;
    unsigned long long t0 = 0;
    int nbad = 0;
    for (int r = 0; r < ROUNDS; r++) {
        // round r: writer = r & 1; value = r*1000 + index
        if ((r & 1) == who) {
            for (int i = tid; i < 1024; i += 256) st_dev(&tile[i], (double)(r * 1000 + i));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flag, r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (tid == 0) ok = wait_flag(flag, r + 1);
            __syncthreads();
            if (!ok) { if (tid == 0) atomicAdd(bad, 1000000); return; }
            for (int i = tid; i < 1024; i += 256) { const double v = ld_dev(&tile[i]); if (v != (double)(r * 1000 + i)) nbad++; }
        }
        if (r == 10 && tid == 0) t0 = __builtin_readcyclecounter();
    }
    if (tid == 0 && who == 0) cyc[0] = __builtin_readcyclecounter() - t0;
    if (nbad) atomicAdd(bad, nbad);
}
// claim latency: one thread, dependent device-scope fetch_add chain
__global__ void k_claim(int* ctr, unsigned long long* cyc)
{
    unsigned long long t0 = __builtin_readcyclecounter();
    int s = 0;
    for (int i = 0; i < 100; i++) s += __hip_atomic_fetch_add(ctr, 1 + (s & 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    cyc[1] = __builtin_readcyclecounter() - t0; cyc[2] = s;
}
int main()
{
    double* tile; int* flag; int* bad; unsigned long long* cyc;
    hipMalloc(&tile, 8192); hipMalloc(&flag, 64); hipMalloc(&bad, 64); hipMalloc(&cyc, 64);
    for (int partner : {1, 2, 8, 9, 255}) {
        hipMemset(flag, 0, 64); hipMemset(bad, 0, 64); hipMemset(cyc, 0, 64); hipMemset(tile, 0, 8192);
        hipLaunchKernelGGL(k_pingpong, dim3(256), dim3(256), 0, 0, tile, flag, bad, cyc, partner);
        hipDeviceSynchronize();
        int hb; unsigned long long hc[4];
        hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(hc, cyc, 32, hipMemcpyDeviceToHost);
        printf("partner WG %3d: bad=%d  cycles/hand-off (8 KB tile + flag, s_memtime 100 MHz ticks) = %.1f\n", partner, hb, (double)hc[0] / (ROUNDS - 11));
    }
    hipMemset(flag, 0, 64);
    hipLaunchKernelGGL(k_claim, dim3(1), dim3(1), 0, 0, flag, cyc);
    hipDeviceSynchronize();
    unsigned long long hc[4]; hipMemcpy(hc, cyc, 32, hipMemcpyDeviceToHost);
    printf("dependent agent-scope fetch_add: %.1f ticks each\n", (double)hc[1] / 100);
    return 0;
}
