/* TEST INFRASTRUCTURE -- part of the CPU oracle, not of the product.
 *
 * Single-precision sine with the exact results of the libm the reference links
 * against in the build container: glibc 2.35, x86-64, FMA-capable CPU (ifunc
 * variant __sinf_fma).  The reference calls std::sin(float) inside process loops
 * whose result is truncated to an integer delay (src/oalsfxpp.cpp:4273, 7454-7465,
 * and 5725 for the ring modulator), so a 1-ulp difference can move a delay tap by a
 * whole sample.  The oracle therefore does not call the host's libm; it restates the
 * published algorithm glibc uses (sysdeps/ieee754/flt-32/s_sinf.c + sincosf.h,
 * from ARM optimized-routines: double-precision polynomial on a quadrant-reduced
 * argument) with every a*b+c written as an explicit fused multiply-add, which is
 * what the FMA build of that file executes.
 *
 * Pinned by oracle/sinf_check.c: bit-identical to sinf() of this container's glibc
 * for all 2,240,806,914 floats with |x| <= 100 (the non-FMA evaluation differs on 8).
 */
#ifndef OALSFX_ORACLE_REF_SINF_H
#define OALSFX_ORACLE_REF_SINF_H

#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct {
    double sign[4];
    double hpi_inv; /* 2/pi * 2^24 */
    double hpi;     /* pi/2 */
    double c0, c1, c2, c3, c4;
    double s1, s2, s3;
} oracle_sincos_t;

static const oracle_sincos_t oracle_sincos_table[2] = {
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2,
     0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3,
     0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2,
     -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3,
     0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};

static inline uint32_t oracle_abstop12(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    return (u >> 20) & 0x7ff;
}

static inline float oracle_sinf_poly(double x, double x2, const oracle_sincos_t* p, int n)
{
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = fma(x2, p->s3, p->s2);
        const double x7 = x3 * x2;
        const double s = fma(x3, p->s1, x);
        return (float)fma(x7, s1, s);
    } else {
        const double x4 = x2 * x2;
        const double c2 = fma(x2, p->c4, p->c3);
        const double c1 = fma(x2, p->c1, p->c0);
        const double x6 = x4 * x2;
        const double c = fma(x4, p->c2, c1);
        return (float)fma(x6, c2, c);
    }
}

/* Valid for |y| < 120 (every argument the process path produces lies in [-pi, 2*pi]). */
static inline float oracle_sinf(float y)
{
    double x = y;
    const oracle_sincos_t* p = &oracle_sincos_table[0];
    if (oracle_abstop12(y) < oracle_abstop12(0x1.921FB6p-1f)) {
        if (oracle_abstop12(y) < oracle_abstop12(0x1p-12f)) return y;
        return oracle_sinf_poly(x, x * x, p, 0);
    }
    const double r = x * p->hpi_inv;
    const int n = ((int32_t)r + 0x800000) >> 24;
    x = fma(-(double)n, p->hpi, x);
    const double s = p->sign[n & 3];
    if (n & 2) p = &oracle_sincos_table[1];
    return oracle_sinf_poly(x * s, x * x, p, n);
}

#endif
