/*
 * cv_prims.c — CPU ORACLE, part 1: the OpenCV 4.x primitives the reference's
 * hot path calls, restated from OpenCV's published algorithms.
 * TEST INFRASTRUCTURE ONLY (see svo_oracle.h). PARITY UNPINNED against a real
 * OpenCV build: none is installed and the reference pins no version
 * (src/Makefile:8-9). Call sites in the reference are cited per function.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "svo_oracle.h"
#include "../include/svo_libm.h"

/* ------------------------------------------------------------------------ */
/* cv::Rodrigues (vector -> matrix), called from PoseManager::set_pose,
 * src/lib/pose_manager.cpp:15-16, and inside cv::projectPoints
 * (src/lib/transform_keypoints.cpp:45). OpenCV works in double:
 *   theta = |r|; theta < DBL_EPSILON -> I
 *   R = cos(theta) I + (1-cos(theta)) k k^T + sin(theta) [k]x,  k = r/theta  */
void svo_o_rodrigues(const float r[3], double R[9])
{
    double rx = r[0], ry = r[1], rz = r[2];
    double theta = sqrt(rx * rx + ry * ry + rz * rz);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = 0.0;
        R[0] = R[4] = R[8] = 1.0;
        return;
    }
    /* sin / cos from include/svo_libm.h (fdlibm kernels): the reference's libm is not pinned, and
     * this way the oracle and the device evaluate the same polynomial */
    double c, s;
    svo_sincos(theta, &s, &c);
    const double c1 = 1.0 - c;
    double itheta = 1.0 / theta;
    rx *= itheta; ry *= itheta; rz *= itheta;
    const double rrt[9] = { rx * rx, rx * ry, rx * rz,
                            rx * ry, ry * ry, ry * rz,
                            rx * rz, ry * rz, rz * rz };
    const double r_x[9] = { 0, -rz, ry,
                            rz, 0, -rx,
                            -ry, rx, 0 };
    const double eye[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    for (int k = 0; k < 9; k++) {
        double t = c * eye[k];      /* c*Matx33d::eye()      */
        t = t + c1 * rrt[k];        /* + c1*rrt              */
        R[k] = t + s * r_x[k];      /* + s*r_x               */
    }
}

void svo_o_rodrigues_f(const float r[3], float R[9])
{
    double Rd[9];
    svo_o_rodrigues(r, Rd);
    for (int k = 0; k < 9; k++) R[k] = (float)Rd[k];
}

void svo_o_pose_matrices(const float pose[6], float rot[9], float inv_rot[9])
{
    float a[3] = { pose[3], pose[4], pose[5] };
    float na[3] = { -pose[3], -pose[4], -pose[5] };
    svo_o_rodrigues_f(a, rot);
    svo_o_rodrigues_f(na, inv_rot);
}

/* ------------------------------------------------------------------------ */
/* JacobiSVDImpl_<float> of OpenCV modules/core/src/lapack.cpp (one-sided
 * Hestenes Jacobi on the rows of At), reached through
 *   Matx66f::inv(DECOMP_SVD)  src/lib/pose_estimator.cpp:405, pose_refinement.cpp:398
 *   cv::solve(.., DECOMP_SVD) src/lib/depth_filter.cpp:200
 *   cv::KalmanFilter::correct src/lib/depth_filter.cpp:215, stereo_slam.cpp:344
 * The random re-orthogonalisation OpenCV applies to rows with a zero singular
 * value is not restated: those rows only ever get multiplied by 0 afterwards. */
void svo_o_jacobi_svd(float *At, int astep, float *Wout, float *Vt, int vstep,
                      int m, int n, int n1)
{
    const double minval = FLT_MIN;
    const float eps = FLT_EPSILON * 2;
    double W[32];
    int i, j, k, iter, max_iter = m > 30 ? m : 30;
    float c, s;
    double sd;

    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) {
            float t = At[i * astep + k];
            sd += (double)t * t;
        }
        W[i] = sd;
        if (Vt) {
            for (k = 0; k < n; k++) Vt[i * vstep + k] = 0;
            Vt[i * vstep + i] = 1;
        }
    }

    for (iter = 0; iter < max_iter; iter++) {
        int changed = 0;
        for (i = 0; i < n - 1; i++)
            for (j = i + 1; j < n; j++) {
                float *Ai = At + i * astep, *Aj = At + j * astep;
                double a = W[i], p = 0, b = W[j];

                for (k = 0; k < m; k++) p += (double)Ai[k] * Aj[k];

                if (fabs(p) <= eps * sqrt((double)a * b)) continue;

                p *= 2;
                double beta = a - b, gamma = svo_hypot((double)p, beta);   /* OpenCV lapack.cpp's own hypot */
                if (beta < 0) {
                    double delta = (gamma - beta) * 0.5;
                    s = (float)sqrt(delta / gamma);
                    c = (float)(p / (gamma * s * 2));
                } else {
                    c = (float)sqrt((gamma + beta) / (gamma * 2));
                    s = (float)(p / (gamma * c * 2));
                }

                a = b = 0;
                for (k = 0; k < m; k++) {
                    float t0 = c * Ai[k] + s * Aj[k];
                    float t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0; Aj[k] = t1;
                    a += (double)t0 * t0; b += (double)t1 * t1;
                }
                W[i] = a; W[j] = b;
                changed = 1;

                if (Vt) {
                    float *Vi = Vt + i * vstep, *Vj = Vt + j * vstep;
                    for (k = 0; k < n; k++) {
                        float t0 = c * Vi[k] + s * Vj[k];
                        float t1 = -s * Vi[k] + c * Vj[k];
                        Vi[k] = t0; Vj[k] = t1;
                    }
                }
            }
        if (!changed) break;
    }

    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) {
            float t = At[i * astep + k];
            sd += (double)t * t;
        }
        W[i] = sqrt(sd);
    }

    for (i = 0; i < n - 1; i++) {
        j = i;
        for (k = i + 1; k < n; k++)
            if (W[j] < W[k]) j = k;
        if (i != j) {
            double tw = W[i]; W[i] = W[j]; W[j] = tw;
            if (Vt) {
                for (k = 0; k < m; k++) {
                    float t = At[i * astep + k]; At[i * astep + k] = At[j * astep + k]; At[j * astep + k] = t;
                }
                for (k = 0; k < n; k++) {
                    float t = Vt[i * vstep + k]; Vt[i * vstep + k] = Vt[j * vstep + k]; Vt[j * vstep + k] = t;
                }
            }
        }
    }

    for (i = 0; i < n; i++) Wout[i] = (float)W[i];
    if (!Vt) return;

    for (i = 0; i < n1; i++) {
        sd = i < n ? W[i] : 0;
        s = (float)(sd > minval ? 1 / sd : 0.);
        for (k = 0; k < m; k++) At[i * astep + k] *= s;
    }
}

/* threshold factor of SVBkSb for float data: OpenCV passes (float)(DBL_EPSILON*2)
 * for BOTH element types (lapack.cpp); restated as remembered. */
static const float SVBKSB_EPS = (float)(DBL_EPSILON * 2);

/* cv::invert(src, dst, DECOMP_SVD) as used by Matx::inv: SVD::compute then
 * SVD::backSubst with an identity right-hand side; Matx::inv returns zeros
 * when cv::invert reports 0 (sigma_max < FLT_EPSILON or sigma_min/sigma_max == 0). */
int svo_o_inv_svd(const float *A, int n, float *Ainv)
{
    float At[16 * 16], Vt[16 * 16], W[16];
    int i, j, r;
    /* temp_a = transpose(src): rows of At are columns of A */
    for (i = 0; i < n; i++)
        for (j = 0; j < n; j++) At[i * n + j] = A[j * n + i];
    svo_o_jacobi_svd(At, n, W, Vt, n, n, n, n);
    /* rows of At are now the left singular vectors u_i, rows of Vt the v_i */
    for (i = 0; i < n * n; i++) Ainv[i] = 0;
    double threshold = 0;
    for (i = 0; i < n; i++) threshold += W[i];
    threshold *= SVBKSB_EPS;
    for (i = 0; i < n; i++) {
        double wi = W[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        double buffer[16];
        for (j = 0; j < n; j++) buffer[j] = At[i * n + j] * wi;      /* u_i[j] / w_i */
        for (r = 0; r < n; r++) {                                    /* MatrAXPY    */
            float sv = Vt[i * n + r];
            for (j = 0; j < n; j++)
                Ainv[r * n + j] = (float)(Ainv[r * n + j] + sv * buffer[j]);
        }
    }
    int ok = W[0] >= FLT_EPSILON ? (W[n - 1] / W[0] != 0) : 0;
    if (!ok)
        for (i = 0; i < n * n; i++) Ainv[i] = 0;
    return ok;
}

/* cv::solve(A, b, x, DECOMP_SVD) for m >= n, one right-hand side
 * (src/lib/depth_filter.cpp:194-200: 3x2). */
void svo_o_solve_svd(const float *A, int m, int n, const float *b, float *x)
{
    float At[16 * 16], Vt[16 * 16], W[16];
    int i, j;
    for (i = 0; i < n; i++)
        for (j = 0; j < m; j++) At[i * m + j] = A[j * n + i];
    svo_o_jacobi_svd(At, m, W, Vt, n, m, n, n);
    for (j = 0; j < n; j++) x[j] = 0;
    double threshold = 0;
    for (i = 0; i < n; i++) threshold += W[i];
    threshold *= SVBKSB_EPS;
    for (i = 0; i < n; i++) {
        double wi = W[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        double s = 0;
        for (j = 0; j < m; j++) s += At[i * m + j] * b[j];   /* float product, double sum */
        s *= wi;
        for (j = 0; j < n; j++) x[j] = (float)(x[j] + s * Vt[i * n + j]);
    }
}

/* ------------------------------------------------------------------------ */
/* exponential_map, src/include/exponential_map.hpp:12-37. The norm is hard
 * set to 1; `cos`/`sin` resolve to the double overloads (the header is
 * included before any using-directive), so the two scale factors are doubles
 * that multiply a Matx33f (result rounded to float per element).
 * Known answer: src/test/test_exponential_map.cpp:35-48. */
static void mat33f_mul(const float *a, const float *b, float *out)
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            float s = 0;
            for (int k = 0; k < 3; k++) s += a[i * 3 + k] * b[k * 3 + j];
            out[i * 3 + j] = s;
        }
}

static void mat33f_vec(const float *a, const float *v, float *out)
{
    float t[3];
    for (int i = 0; i < 3; i++) {
        float s = 0;
        for (int k = 0; k < 3; k++) s += a[i * 3 + k] * v[k];
        t[i] = s;
    }
    out[0] = t[0]; out[1] = t[1]; out[2] = t[2];
}

void svo_o_exponential_map(const float twist[6], float out[6])
{
    const float v[3] = { twist[0], twist[1], twist[2] };
    const float w[3] = { twist[3], twist[4], twist[5] };
    const float K[9] = { 0, -w[2], w[1],
                         w[2], 0, -w[0],
                         -w[1], w[0], 0 };
    float K2[9], M[9];
    const float norm = 1.0f;
    const double c1 = 1 - cos((double)norm);
    const double c2 = norm - sin((double)norm);
    mat33f_mul(K, K, K2);
    for (int k = 0; k < 9; k++) {
        float e = (k == 0 || k == 4 || k == 8) ? 1.0f : 0.0f;
        float t0 = e * norm;              /* _eye*_norm                 */
        float t1 = (float)(K[k] * c1);    /* (1-cos)*w_skew             */
        float t2 = (float)(K2[k] * c2);   /* (norm-sin)*(w_skew*w_skew) */
        M[k] = (t0 + t1) + t2;
    }
    float t[3];
    mat33f_vec(M, v, t);
    out[0] = t[0]; out[1] = t[1]; out[2] = t[2];
    out[3] = w[0]; out[4] = w[1]; out[5] = w[2];
}

/* ------------------------------------------------------------------------ */
/* project_keypoints, src/lib/transform_keypoints.cpp:11-48: points are
 * translated by -t in float, then cv::projectPoints(rvec = -r, tvec = 0,
 * K, dist = (k1,k2,p1,p2,k3)) which works in double and stores float. */
void svo_o_project_keypoints(const float pose[6], const svo_kp3d *in, int n,
                             const svo_camera_settings *cam, svo_kp2d *out)
{
    const float nr[3] = { -pose[3], -pose[4], -pose[5] };
    double R[9];
    svo_o_rodrigues(nr, R);
    const double fx = cam->fx, fy = cam->fy, cx = cam->cx, cy = cam->cy;
    const double k1 = cam->k1, k2 = cam->k2, p1 = cam->p1, p2 = cam->p2, k3 = cam->k3;
    for (int i = 0; i < n; i++) {
        const float Xf = in[i].x - pose[0], Yf = in[i].y - pose[1], Zf = in[i].z - pose[2];
        const double X = Xf, Y = Yf, Z = Zf;
        double x = R[0] * X + R[1] * Y + R[2] * Z + 0.0;
        double y = R[3] * X + R[4] * Y + R[5] * Z + 0.0;
        double z = R[6] * X + R[7] * Y + R[8] * Z + 0.0;
        z = z ? 1. / z : 1;
        x *= z; y *= z;
        const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
        const double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
        const double cdist = 1 + k1 * r2 + k2 * r4 + k3 * r6;
        const double xd = x * cdist + p1 * a1 + p2 * a2;
        const double yd = y * cdist + p1 * a3 + p2 * a1;
        out[i].x = (float)(xd * fx + cx);
        out[i].y = (float)(yd * fy + cy);
    }
}

/* ------------------------------------------------------------------------ */
static int reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

/* cv::pyrDown for CV_8U (separable [1 4 6 4 1], BORDER_REFLECT_101, rounding
 * (v+128)>>8), used by cv::buildOpticalFlowPyramid, src/lib/stereo_slam.cpp:139. */
void svo_o_pyr_down(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride)
{
    const int dw = (w + 1) / 2, dh = (h + 1) / 2;
    int *rows = (int *)malloc(sizeof(int) * (size_t)dw * 5);
    for (int y = 0; y < dh; y++) {
        for (int k = 0; k < 5; k++) {
            const int sy = reflect101(2 * y + k - 2, h);
            const uint8_t *s = src + (size_t)sy * sstride;
            int *row = rows + (size_t)k * dw;
            for (int x = 0; x < dw; x++) {
                const int x0 = reflect101(2 * x - 2, w), x1 = reflect101(2 * x - 1, w);
                const int x2 = 2 * x, x3 = reflect101(2 * x + 1, w), x4 = reflect101(2 * x + 2, w);
                row[x] = s[x2] * 6 + (s[x1] + s[x3]) * 4 + s[x0] + s[x4];
            }
        }
        for (int x = 0; x < dw; x++) {
            const int v = rows[2 * dw + x] * 6 + (rows[dw + x] + rows[3 * dw + x]) * 4 +
                          rows[x] + rows[4 * dw + x];
            dst[(size_t)y * dstride + x] = (uint8_t)((v + 128) >> 8);
        }
    }
    free(rows);
}

/* calcSharrDeriv of OpenCV's lkpyramid.cpp: dx = [3 10 3]^T (x) [-1 0 1],
 * dy = [-1 0 1]^T (x) [3 10 3], image borders by reflection-101, int16. */
void svo_o_scharr(const uint8_t *src, int w, int h, int sstride, int16_t *dst)
{
    int16_t *trow0 = (int16_t *)malloc(sizeof(int16_t) * (size_t)(w + 2) * 2);
    int16_t *trow1 = trow0 + (w + 2);
    for (int y = 0; y < h; y++) {
        const uint8_t *srow0 = src + (size_t)(y > 0 ? y - 1 : h > 1 ? 1 : 0) * sstride;
        const uint8_t *srow1 = src + (size_t)y * sstride;
        const uint8_t *srow2 = src + (size_t)(y < h - 1 ? y + 1 : h > 1 ? h - 2 : 0) * sstride;
        int16_t *t0 = trow0 + 1, *t1 = trow1 + 1;
        for (int x = 0; x < w; x++) {
            t0[x] = (int16_t)((srow0[x] + srow2[x]) * 3 + srow1[x] * 10);
            t1[x] = (int16_t)(srow2[x] - srow0[x]);
        }
        const int x0 = w > 1 ? 1 : 0, x1 = w > 1 ? w - 2 : 0;
        t0[-1] = t0[x0]; t0[w] = t0[x1];
        t1[-1] = t1[x0]; t1[w] = t1[x1];
        int16_t *drow = dst + (size_t)y * w * 2;
        for (int x = 0; x < w; x++) {
            drow[x * 2] = (int16_t)(t0[x + 1] - t0[x - 1]);
            drow[x * 2 + 1] = (int16_t)((t1[x + 1] + t1[x - 1]) * 3 + t1[x] * 10);
        }
    }
    free(trow0);
}

/* ------------------------------------------------------------------------ */
/* cv::KalmanFilter with one state and one measurement, A = H = 1, as driven
 * by DepthFilter::update_kps3d (src/lib/depth_filter.cpp:202-215) and set up
 * in DepthCalculator::calculate_depth (src/lib/depth_calculator.cpp:277-289):
 * predict() then correct(meas). cv::gemm on float data accumulates in double
 * and stores float; the 1x1 gain comes out of cv::solve(DECOMP_SVD). */
void svo_o_kf1_update(float *x, float *P, float Q, float R, float meas)
{
    /* predict */
    float statePre = (float)((double)1.0f * (double)*x);
    float temp1 = (float)((double)1.0f * (double)*P);
    float errorCovPre = (float)((double)temp1 * (double)1.0f + (double)Q);
    /* correct */
    float temp2 = (float)((double)1.0f * (double)errorCovPre);
    float temp3 = (float)((double)temp2 * (double)1.0f + (double)R);
    float gain;
    svo_o_solve_svd(&temp3, 1, 1, &temp2, &gain);
    float temp5 = meas - (float)((double)1.0f * (double)statePre);
    *x = (float)((double)gain * (double)temp5 + (double)statePre);
    *P = (float)(-((double)gain * (double)temp2) + (double)errorCovPre);
}
