// diagnostic: dependent-issue latencies of the primitives on the GMW pivot chain (one wave, gfx950).  scratch tool
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <class T> __device__ __forceinline__ unsigned long long now(T& dep) { unsigned long long t; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(dep) :: "memory"); return t; }
__device__ __forceinline__ double readlane_d(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
#define REP 64
__global__ void k(double* out, unsigned long long* ts, double x0)
{
    double x = x0 + threadIdx.x * 1e-9;
    unsigned long long t0, t1;
    int i = 0;
    // 1. dependent fma chain
    t0 = now(x);
#pragma unroll
    for (int r = 0; r < REP; r++) x = fma(x, 1.0000001, 1e-9);
    asm volatile("" : "+v"(x));
    t1 = now(x); ts[i++] = t1 - t0;
    // 2. dependent rcp chain
    t0 = now(x);
#pragma unroll
    for (int r = 0; r < REP; r++) x = __builtin_amdgcn_rcp(x);
    asm volatile("" : "+v"(x));
    t1 = now(x); ts[i++] = t1 - t0;
    // 3. dependent sqrt chain (v_sqrt_f64)
    t0 = now(x);
#pragma unroll
    for (int r = 0; r < REP; r++) x = __builtin_amdgcn_sqrt(x);
    asm volatile("" : "+v"(x));
    t1 = now(x); ts[i++] = t1 - t0;
    // 4. readlane -> valu -> readlane chain
    t0 = now(x);
#pragma unroll
    for (int r = 0; r < REP; r++) { double s = readlane_d(x, r & 63); x = fma(x, s, 1e-9); }
    asm volatile("" : "+v"(x));
    t1 = now(x); ts[i++] = t1 - t0;
    // 5. dependent MFMA chain (C = prev D)
    d4 acc = { x, x, x, x };
    t0 = now(acc);
#pragma unroll
    for (int r = 0; r < REP; r++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
    asm volatile("" : "+v"(acc));
    t1 = now(acc); ts[i++] = t1 - t0;
    // 6. MFMA whose A operand depends on the previous result (D -> VALU -> A)
    t0 = now(acc);
#pragma unroll
    for (int r = 0; r < REP; r++) { double a = acc[0] * 1.0000001; acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, x, acc, 0, 0, 0); }
    asm volatile("" : "+v"(acc));
    t1 = now(acc); ts[i++] = t1 - t0;
    // 7. independent MFMAs (4 accumulators)
    d4 b0 = acc, b1 = acc, b2 = acc, b3 = acc;
    t0 = now(x);
#pragma unroll
    for (int r = 0; r < REP / 4; r++) {
        b0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, b0, 0, 0, 0);
        b1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, b1, 0, 0, 0);
        b2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, b2, 0, 0, 0);
        b3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, b3, 0, 0, 0);
    }
    asm volatile("" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
    t1 = now(b3); ts[i++] = t1 - t0;
    // 8. 4x4x4 MFMA dependent chain
    double y = x;
    t0 = now(y);
#pragma unroll
    for (int r = 0; r < REP; r++) y = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, y, 0, 0, 0);
    asm volatile("" : "+v"(y));
    t1 = now(y); ts[i++] = t1 - t0;
    // 9. independent fma (throughput), 8 chains
    double z[8];
#pragma unroll
    for (int q = 0; q < 8; q++) z[q] = x + q;
    t0 = now(z[0]);
#pragma unroll
    for (int r = 0; r < REP / 8; r++)
#pragma unroll
        for (int q = 0; q < 8; q++) z[q] = fma(z[q], 1.0000001, 1e-9);
#pragma unroll
    for (int q = 0; q < 8; q++) asm volatile("" : "+v"(z[q]));
    t1 = now(z[7]); ts[i++] = t1 - t0;
    // 10. LDS write -> barrier-less read round trip (same wave)
    __shared__ double sh[64];
    t0 = now(x);
#pragma unroll
    for (int r = 0; r < 16; r++) { sh[threadIdx.x] = x; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); x = sh[(threadIdx.x + 1) & 63]; asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x) :: "memory"); }
    t1 = now(x); ts[i++] = (t1 - t0) * 4;
    // 11. readlane throughput: 32 independent readlane_d + fma using them
    t0 = now(y);
    double accu = 0;
#pragma unroll
    for (int r = 0; r < REP; r++) { double s = readlane_d(y, r & 63); accu = fma(s, z[r & 7], accu); }
    asm volatile("" : "+v"(accu));
    t1 = now(accu); ts[i++] = t1 - t0;
    // 12. ds_bpermute dependent chain
    int iv = threadIdx.x;
    t0 = now(iv);
#pragma unroll
    for (int r = 0; r < 16; r++) iv = __builtin_amdgcn_ds_bpermute(((threadIdx.x + 1) & 63) << 2, iv);
    asm volatile("" : "+v"(iv));
    t1 = now(iv); ts[i++] = (t1 - t0) * 4;
    out[threadIdx.x] = x + acc[0] + b0[0] + b1[1] + b2[2] + b3[3] + y + z[0] + z[1] + z[2] + z[3] + z[4] + z[5] + z[6] + z[7] + accu + iv;
}
int main()
{
    double* out; unsigned long long* ts;
    hipMalloc(&out, 64 * 8); hipMalloc(&ts, 16 * 8);
    unsigned long long h[16];
    const char* nm[] = { "fma dep", "rcp dep", "sqrt dep", "readlane_d+fma dep", "mfma16 dep (C chain)", "mfma16 dep via VALU->A", "mfma16 indep x4", "mfma4x4x4 dep", "fma indep x8", "LDS write->read", "readlane_d+fma indep", "ds_bpermute dep" };
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, ts, 1.37);
        hipDeviceSynchronize();
        hipMemcpy(h, ts, sizeof h, hipMemcpyDeviceToHost);
    }
    for (int i = 0; i < 12; i++) printf("%-26s %7.1f cycles/op\n", nm[i], h[i] / (double)REP);
    return 0;
}
