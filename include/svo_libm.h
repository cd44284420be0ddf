/*
 * svo_libm.h — the two libm-class functions on the hot path, written out so that every
 * build of them (host gcc, hipcc for gfx950) returns the same bits.
 *
 * The reference reaches sin / cos through cv::Rodrigues (src/lib/pose_manager.cpp:15-16 and
 * inside cv::projectPoints, src/lib/transform_keypoints.cpp:45) and a hypot through the
 * Jacobi SVD behind Matx66f::inv(DECOMP_SVD) (src/lib/pose_estimator.cpp:405,
 * src/lib/pose_refinement.cpp:398). Which libm the reference links is not pinned; any
 * conforming one is within an ulp. The Gauss-Newton loops stop on cost differences at float
 * rounding level, so a last-bit difference between the host libm and the device math library
 * changes an iteration trace every few hundred frames. Both the HIP kernels and the CPU oracle
 * therefore evaluate
 *   - svo_sincos: the fdlibm kernels (k_sin.c / k_cos.c polynomials, Cody-Waite reduction by
 *     pi/2 in three parts), < 1 ulp on the range a rotation vector can have;
 *   - svo_hypot: OpenCV's own formula of modules/core/src/lapack.cpp (|a| sqrt(1 + (b/a)^2)),
 *     which is what its JacobiSVDImpl_ calls;
 * built from + - * / sqrt only (IEEE, no contraction: -ffp-contract=off on both sides).
 */
#ifndef SVO_LIBM_H
#define SVO_LIBM_H

#include <math.h>

#if defined(__HIPCC__)
#define SVO_HD __host__ __device__
#else
#define SVO_HD
#endif

/* sin on [-pi/4, pi/4]; y is the tail of x (iy != 0) */
SVO_HD static inline double svo_k_sin(double x, double y, int iy)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x, w = z * z;
    const double r = S2 + z * (S3 + z * S4) + z * w * (S5 + z * S6);
    const double v = z * x;
    if (iy == 0) return x + v * (S1 + z * r);
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

/* cos on [-pi/4, pi/4] with tail y */
SVO_HD static inline double svo_k_cos(double x, double y)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x;
    double w = z * z;
    const double r = z * (C1 + z * (C2 + z * C3)) + (w * w) * (C4 + z * (C5 + z * C6));
    const double hz = 0.5 * z;
    w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * y));
}

/* sin and cos of x, |x| < ~1e5 (beyond that the reduction loses accuracy, never determinism) */
SVO_HD static inline void svo_sincos(double x, double *s, double *c)
{
    const double pio4 = 7.85398163397448278999e-01;
    if (fabs(x) <= pio4) {
        *s = svo_k_sin(x, 0.0, 0);
        *c = svo_k_cos(x, 0.0);
        return;
    }
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079632673412561417e+00;   /* first 33 bits of pi/2 */
    const double pio2_2 = 6.07710050630396597660e-11;   /* second 33 bits */
    const double pio2_2t = 2.02226624879595063154e-21;  /* pi/2 - (pio2_1 + pio2_2) */
    const double fn = rint(x * invpio2);
    const int n = (int)fn;
    double r = x - fn * pio2_1;
    const double t = r;
    const double w2 = fn * pio2_2;
    r = t - w2;
    const double w = fn * pio2_2t - ((t - r) - w2);
    const double y0 = r - w;
    const double y1 = (r - y0) - w;
    const double ks = svo_k_sin(y0, y1, 1), kc = svo_k_cos(y0, y1);
    switch (n & 3) {
        case 0: *s = ks; *c = kc; break;
        case 1: *s = kc; *c = -ks; break;
        case 2: *s = -ks; *c = -kc; break;
        default: *s = -kc; *c = ks; break;
    }
}

/* hypot of OpenCV's lapack.cpp */
SVO_HD static inline double svo_hypot(double a, double b)
{
    a = fabs(a);
    b = fabs(b);
    if (a > b) {
        b /= a;
        return a * sqrt(1 + b * b);
    }
    if (b > 0) {
        a /= b;
        return b * sqrt(1 + a * a);
    }
    return 0;
}

#endif
