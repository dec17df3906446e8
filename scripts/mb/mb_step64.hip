// diagnostic: where does one k_gmw_step64 spend its cycles (block (0,0), wave 0)?  scratch tool
#define SRUKF_STAMPS 1
#include "../../cv-monoslam_amd/csrc/srukf_factor.hip"
#include <cstdio>
#include <vector>
int main()
{
    const int n = 1204, ld = 1216;
    std::vector<double> h((size_t)ld * ld);
    for (int r = 0; r < ld; r++) for (int c = 0; c < ld; c++) h[(size_t)r * ld + c] = (r == c) ? 2.0 + 0.001 * r : 0.3 / (1 + abs(r - c));
    double *G, *D, *S; void* pan[2];
    hipMalloc(&G, sizeof(double) * ld * ld); hipMalloc(&S, sizeof(double) * ld * ld); hipMalloc(&D, sizeof(double) * ld);
    hipMalloc(&pan[0], sizeof(GmwPanel64)); hipMalloc(&pan[1], sizeof(GmwPanel64));
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; rep++) {
        hipMemcpy(G, h.data(), sizeof(double) * ld * ld, hipMemcpyHostToDevice);
        hipEventRecord(a, st);
        int pb = 0, nl = 0;
        for (int j0 = -64; j0 + 64 < ld; j0 += 64, pb ^= 1, nl++) srukf_launch_gmw_step64(st, n, ld, j0, 1e-13, G, pan[pb ^ 1], pan[pb], D, S, nullptr);
        hipEventRecord(b, st); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("whole factorisation: %.1f us (%d launches)\n", ms * 1000, nl);
    }
    hipMemcpy(G, h.data(), sizeof(double) * ld * ld, hipMemcpyHostToDevice);
    int pb = 0;
    for (int j0 = -64; j0 <= 512; j0 += 64, pb ^= 1) srukf_launch_gmw_step64(st, n, ld, j0, 1e-13, G, pan[pb ^ 1], pan[pb], D, S, nullptr);
    hipStreamSynchronize(st);
    unsigned long long hs[16];
    hipMemcpyFromSymbol(hs, HIP_SYMBOL(srukf_stamps), sizeof hs);
    const char* nm[] = { "start->prefetch issued", "A: 3-stage slab -> LDS", "barrier", "B: quarter K=64 + exchange", "factor 1 (wave 0)", "barrier (waits for waves 1-3)", "C1 + C2", "factor 2 (wave 0)" };
    for (int i = 0; i < 8; i++) printf("%-32s %6llu ticks\n", nm[i], hs[i + 1] - hs[i]);
    printf("after barrier B: wave 0 done %llu, wave 1 done %llu, wave 2 (T1) done %llu, wave 3 done %llu\n", hs[5] - hs[4], hs[12] - hs[4], hs[13] - hs[4], hs[14] - hs[4]);
    printf("total in-kernel %llu ticks\n", hs[8] - hs[0]);
    return 0;
}
