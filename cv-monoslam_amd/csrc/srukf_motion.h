// srukf_motion.h — the robot part of one sigma point through the odometry model: shared by the motion step (srukf_predict.hip)
// and by the frame tail of the rank-aware replay, which prepares the next frame's table of robot poses (srukf_rank.hip).
#pragma once
#include "srukf_device.h"

// One sigma point's robot rows through the odometry model (generateSigmaPoints 1159-1160 restricted to the robot and
// control-noise rows, passSigmaThroughMotionFunction 1492-1523): r[] = propagated pose, (c2, s2) = cos / sin of its heading.
struct MotionCtl { double rot1, trans, rot2, crot2, srot2; };
__device__ __forceinline__ void srukf_motion_point(const MotionCtl& m, const double (&xr)[4], const double (&srow)[4], const double (&mnoise)[3],
                                                   double gs, double (&r)[4], double& c2, double& s2)
{
#pragma clang fp contract(off)
    // (every fused multiply-add written out, contraction off: see srukf_project — the same point evaluated by two kernels must give the same bits)
    double q[3];
#pragma unroll
    for (int e = 0; e < 4; e++) r[e] = fma(srow[e], gs, xr[e] * 1) + 0;      // generateSigmaPoints, 1159-1160: xr * 1 + srow * gs + 0
#pragma unroll
    for (int e = 0; e < 3; e++) q[e] = fma(mnoise[e], gs, 0.0 * 1) + 0;
    const double r1 = m.rot1 - q[0], tr = m.trans - q[1], r2 = m.rot2 - q[2];     // 1492-1494
    double sn, cs;
    sincos(r[3] + r1, &sn, &cs);
    r[0] = fma(tr, cs, r[0]);                                                // 1518-1523
    r[1] = fma(tr, sn, r[1]);
    r[2] += 0.0;
    r[3] += r1 + r2;
    // cos/sin of the final heading by angle addition; the rot2-noise columns evaluate it directly
    if (q[2] == 0.0) { c2 = fma(cs, m.crot2, -(sn * m.srot2)); s2 = fma(sn, m.crot2, cs * m.srot2); }
    else sincos(r[3], &s2, &c2);
}
__device__ __forceinline__ void srukf_motion_centre(const MotionCtl& m, const double (&xr)[4], double (&s0)[4], double& c0s, double& s0s)
{
#pragma clang fp contract(off)
    double sn, cs;
    sincos(xr[3] + m.rot1, &sn, &cs);
    s0[0] = fma(m.trans, cs, xr[0]); s0[1] = fma(m.trans, sn, xr[1]); s0[2] = xr[2] + 0.0; s0[3] = xr[3] + (m.rot1 + m.rot2);
    c0s = fma(cs, m.crot2, -(sn * m.srot2)); s0s = fma(sn, m.crot2, cs * m.srot2);
}
