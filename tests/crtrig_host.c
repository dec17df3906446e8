/* Host build of cv-monoslam_amd/csrc/srukf_crtrig.h (the header the warp kernel includes) for tests/test_host.py:
 * counts the arguments on which its sin / cos differ from binary128 libquadmath values rounded once to double
 * (the oracle's definition in orc_warp_patch), and from this host's libm.
 *   gcc -O2 -ffp-contract=off -mfma -shared -fPIC crtrig_host.c -lquadmath -lm */
#include <math.h>
#include <quadmath.h>
#include <stdint.h>
#include "../cv-monoslam_amd/csrc/srukf_crtrig.h"

void crt_sincos_host(double x, double* s, double* c) { crt_sincos(x, s, c); }

/* n pseudo-random arguments in [-range, range]: out = { mismatches vs quad (sin), (cos), vs libm (sin), (cos) } */
void crt_sweep(long n, double range, uint64_t seed, long out[4])
{
    uint64_t s = seed ? seed : 88172645463325252ull;
    out[0] = out[1] = out[2] = out[3] = 0;
    for (long i = 0; i < n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double x = (2.0 * ((double)(s >> 11) / 9007199254740992.0) - 1.0) * range;
        double a, b;
        crt_sincos(x, &a, &b);
        if (a != (double)sinq((__float128)x)) out[0]++;
        if (b != (double)cosq((__float128)x)) out[1]++;
        if (a != sin(x)) out[2]++;
        if (b != cos(x)) out[3]++;
    }
}
