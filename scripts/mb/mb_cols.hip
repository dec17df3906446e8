// micro-benchmark + check of the column-layout (VALU/DPP) 32x32 GMW block factor.  scratch tool
#define SRUKF_STAMPS 1
#include "../../cv-monoslam_amd/csrc/srukf_gmw_cols.h"
#include <cstdio>
#include <cmath>
#include <vector>
__device__ __forceinline__ unsigned long long now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
__global__ __launch_bounds__(256) void k(const double* G, int ld, double eps, double* Tt, double* pD, double* psq, double* prD, double* Dall, double* S, unsigned long long* ts)
{
    __shared__ double region[GMW_XM_DOUBLES + GMW_LM_DOUBLES + 32];
    const GmwColsLds w = gmw_cols_carve(region);
    __shared__ double Tl[32 * 33];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int e = tid; e < 1024; e += 256) w.Xm[e >> 5][e & 31] = G[(size_t)(e >> 5) * ld + (e & 31)];
    if (tid < 32) w.Dv[tid] = 0.0;
    __syncthreads();
    unsigned long long t0 = now();
    if (tid == 64) srukf_stamps[19] = t0;
    if (wv == 0) { gmw_cols_pivot_wave(w, eps, lane); ts[0] = now() - t0; }
    else if (wv == 2) { gmw_cols_t_wave<true>(w, lane, Tt, Tl); ts[1] = now() - t0; }
    else if (wv == 1) { gmw_cols_out_wave(w, 0, lane, 32, ld, 0, pD, psq, prD, Dall, S); if (lane == 0) ts[2] = now() - t0; }
    else { gmw_cols_out_wave(w, 1, lane, 32, ld, 0, pD, psq, prD, Dall, S); }
}
int main()
{
    const int ld = 32;
    std::vector<double> h(1024), B(1024);
    srand(3);
    for (auto& x : B) x = rand() / (double)RAND_MAX - 0.5;
    for (int r = 0; r < 32; r++) for (int c = 0; c < 32; c++) { double s = (r == c) ? 0.5 : 0.0; for (int k = 0; k < 32; k++) s += B[k * 32 + r] * B[k * 32 + c]; h[r * 32 + c] = s; }
    // host reference: LDL^T with D = max(eps, |c_jj|), T = L^{-1}
    const double eps = 1e-13;
    std::vector<double> C = h, Sref(1024, 0.0), L(1024, 0.0), Dr(32);
    for (int j = 0; j < 32; j++) {
        Dr[j] = fmax(eps, fabs(C[j * 32 + j]));
        for (int c = j; c < 32; c++) Sref[j * 32 + c] = (c == j) ? sqrt(Dr[j]) : C[j * 32 + c] / sqrt(Dr[j]);
        for (int r = j + 1; r < 32; r++) L[j * 32 + r] = C[j * 32 + r] / Dr[j];
        for (int r = j + 1; r < 32; r++) for (int c = r; c < 32; c++) C[r * 32 + c] -= L[j * 32 + r] * C[j * 32 + c];
    }
    // T[r][c]: rows: T = inverse of unit lower M^T+I where (I+M^T)[r][k] = L[k][r]
    std::vector<double> T(1024, 0.0);
    for (int c = 0; c < 32; c++) { for (int r = 0; r < 32; r++) { double s = (r == c) ? 1.0 : 0.0; for (int k = 0; k < r; k++) s -= L[k * 32 + r] * T[k * 32 + c]; T[r * 32 + c] = s; } }
    double *G, *Tt, *pD, *psq, *prD, *Dall, *S; unsigned long long* ts;
    hipMalloc(&G, 8192); hipMalloc(&Tt, 8192); hipMalloc(&S, 8192); hipMalloc(&pD, 256); hipMalloc(&psq, 256); hipMalloc(&prD, 256); hipMalloc(&Dall, 256); hipMalloc(&ts, 64);
    hipMemcpy(G, h.data(), 8192, hipMemcpyHostToDevice); hipMemset(S, 0, 8192); hipMemset(Tt, 0, 8192);
    unsigned long long hts[3];
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, G, ld, eps, Tt, pD, psq, prD, Dall, S, ts);
        hipDeviceSynchronize();
        hipMemcpy(hts, ts, sizeof hts, hipMemcpyDeviceToHost);
        printf("pivot wave %llu cycles, T wave %llu, incl. outputs %llu\n", hts[0], hts[1], hts[2]);
    }
    std::vector<double> hS(1024), hT(1024), hD(32);
    hipMemcpy(hS.data(), S, 8192, hipMemcpyDeviceToHost); hipMemcpy(hT.data(), Tt, 8192, hipMemcpyDeviceToHost); hipMemcpy(hD.data(), pD, 256, hipMemcpyDeviceToHost);
    double eS = 0, eT = 0, eD = 0;
    for (int r = 0; r < 32; r++) { eD = fmax(eD, fabs(hD[r] - Dr[r])); for (int c = 0; c < 32; c++) { if (c >= r) eS = fmax(eS, fabs(hS[r * 32 + c] - Sref[r * 32 + c])); eT = fmax(eT, fabs(hT[c * 32 + r] - T[r * 32 + c])); } }
    { unsigned long long hs[32]; hipMemcpyFromSymbol(hs, HIP_SYMBOL(srukf_stamps), sizeof hs); printf("pivot wave at pivot 0/8/16/24: %llu %llu %llu %llu\n", hs[12]-hs[19], hs[13]-hs[19], hs[14]-hs[19], hs[15]-hs[19]); printf("T wave polls passed at: %llu %llu %llu %llu %llu\n", hs[21]-hs[19], hs[22]-hs[19], hs[23]-hs[19], hs[24]-hs[19], hs[25]-hs[19]); printf("out wave polls passed at: %llu %llu %llu %llu %llu\n", hs[20]-hs[19], hs[22]-hs[19], hs[24]-hs[19], hs[26]-hs[19], hs[27]-hs[19]); }
    printf("max |dS| %.3e  |dT| %.3e  |dD| %.3e\n", eS, eT, eD);
    return 0;
}
