// srukf_crtrig.h — sin / cos of a double, evaluated in double-double arithmetic and rounded once.
//
// Why: wrapPatch (SLAM.cpp:1803-1906) ends in floor / ceil / a truncating uchar cast, so the bytes of matchPatch
// depend on the LAST BIT of cos(theta), sin(theta) of the robot heading (getTransferMatrix, 1031-1037).  The
// reference takes them from the host C runtime.  ocml's device sin / cos are accurate to an ulp or two, which is not
// enough to land on the same byte as a host libm whenever the warped coordinate sits next to an integer (it does
// whenever the robot has barely moved since the landmark was created).  The result below carries ~100 bits before
// the single final rounding, i.e. it is the correctly rounded value except when the true value lies within 2^-45 ulp
// of a rounding boundary — the same value a correctly rounding host libm returns (tests/test_host.py sweeps it
// against this container's libm; compiled for the host by tests/crtrig_host.c).
//
// Only +, -, *, fma on doubles: bit-reproducible on any IEEE-754 machine as long as the compiler neither contracts
// nor reassociates (srukf_assoc.hip is built with -ffp-contract=off).  Valid for |x| <= 2^20.
#ifndef SRUKF_CRTRIG_H_
#define SRUKF_CRTRIG_H_

#if defined(__HIPCC__)
#define CRT_FN __device__ __host__ static inline
#define CRT_FMA(a, b, c) __builtin_fma((a), (b), (c))
#else
#define CRT_FN static inline
#define CRT_FMA(a, b, c) __builtin_fma((a), (b), (c))
#endif

typedef struct crt_dd { double hi, lo; } crt_dd;

CRT_FN crt_dd crt_two_sum(double a, double b)
{
    crt_dd r; r.hi = a + b;
    const double bb = r.hi - a;
    r.lo = (a - (r.hi - bb)) + (b - bb);
    return r;
}
CRT_FN crt_dd crt_fast_two_sum(double a, double b)     // |a| >= |b|
{
    crt_dd r; r.hi = a + b; r.lo = b - (r.hi - a);
    return r;
}
CRT_FN crt_dd crt_two_prod(double a, double b)
{
    crt_dd r; r.hi = a * b; r.lo = CRT_FMA(a, b, -r.hi);
    return r;
}
CRT_FN crt_dd crt_add(crt_dd a, crt_dd b)
{
    crt_dd s = crt_two_sum(a.hi, b.hi);
    const crt_dd t = crt_two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = crt_fast_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return crt_fast_two_sum(s.hi, s.lo);
}
CRT_FN crt_dd crt_neg(crt_dd a) { crt_dd r; r.hi = -a.hi; r.lo = -a.lo; return r; }
CRT_FN crt_dd crt_mul(crt_dd a, crt_dd b)
{
    crt_dd p = crt_two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return crt_fast_two_sum(p.hi, p.lo);
}
CRT_FN crt_dd crt_div_d(crt_dd a, double b)            // b: a small exact integer
{
    const double q1 = a.hi / b;
    const crt_dd p = crt_two_prod(q1, b);
    crt_dd r = crt_two_sum(a.hi, -p.hi);
    r.lo = (r.lo - p.lo) + a.lo;
    const double q2 = (r.hi + r.lo) / b;
    return crt_fast_two_sum(q1, q2);
}

// sin and cos of x, each rounded to nearest
CRT_FN void crt_sincos(double x, double* s_out, double* c_out)
{
    // pi/2 = P1 + P2 + P3 + ..., three non-overlapping doubles (161 bits)
    const double P1 = 1.5707963267948966, P2 = 6.123233995736766e-17, P3 = -1.4973849048591698e-33;
    const double two_over_pi = 0.6366197723675814;
    // k = nearest integer to x * 2/pi (ties do not matter: either neighbour leaves |r| <= pi/4 + tiny)
    double k = x * two_over_pi;
    k = (k >= 0.0) ? (double)(long long)(k + 0.5) : -(double)(long long)(0.5 - k);
    // r = x - k * pi/2 in double-double
    crt_dd r; r.hi = x; r.lo = 0.0;
    r = crt_add(r, crt_neg(crt_two_prod(k, P1)));
    r = crt_add(r, crt_neg(crt_two_prod(k, P2)));
    r = crt_add(r, crt_neg(crt_two_prod(k, P3)));
    const crt_dd r2 = crt_mul(r, r);
    // Taylor series with the term recurrence t_{i+1} = -t_i r^2 / ((m+1)(m+2)); |r| <= 0.79: 15 terms reach 2^-110
    crt_dd ts = r, ss = r;                              // sin: r - r^3/3! + ...
    crt_dd tc; tc.hi = 1.0; tc.lo = 0.0;
    crt_dd cs = tc;                                     // cos: 1 - r^2/2! + ...
    for (int i = 1; i <= 15; i++) {
        tc = crt_neg(crt_div_d(crt_mul(tc, r2), (double)((2 * i - 1) * (2 * i))));
        cs = crt_add(cs, tc);
        ts = crt_neg(crt_div_d(crt_mul(ts, r2), (double)((2 * i) * (2 * i + 1))));
        ss = crt_add(ss, ts);
    }
    const double sr = ss.hi + ss.lo, cr = cs.hi + cs.lo;
    const long long q = (long long)k & 3;              // two's complement: also right for negative k
    double s, c;
    switch (q) {
    case 0:  s = sr;  c = cr;  break;
    case 1:  s = cr;  c = -sr; break;
    case 2:  s = -sr; c = -cr; break;
    default: s = -cr; c = sr;  break;
    }
    *s_out = s; *c_out = c;
}

#endif /* SRUKF_CRTRIG_H_ */
