/*
 * hot_path.c — CPU ORACLE, part 2: restatement of the reference's hot path
 * (pyramids, sparse image alignment, KLT refinement, reprojection GN, stereo
 * depth filter). TEST INFRASTRUCTURE ONLY (see svo_oracle.h).
 * Every function cites the reference lines it follows (paths relative to the
 * reference repository root).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "svo_oracle.h"
#include "oracle_internal.h"

/* ======================================================================== */
/* P1  halfSample / createImgPyramid — src/lib/stereo_slam.cpp:93-121        */
void svo_o_build_pyramid(const svo_image *lvl0, int n_levels, svo_image *levels)
{
    levels[0] = *lvl0;
    for (int l = 1; l < n_levels; l++) {
        const svo_image *in = &levels[l - 1];
        svo_image *out = &levels[l];
        out->width = in->width / 2;
        out->height = in->height / 2;
        out->stride = out->width;
        uint8_t *o = (uint8_t *)out->data;
        for (int j = 0; j < out->height; j++) {
            const uint8_t *up = in->data + (size_t)(2 * j) * in->stride;
            const uint8_t *lo = in->data + (size_t)(2 * j + 1) * in->stride;
            for (int i = 0, x = 0; i < out->width; i++, x += 2)
                o[(size_t)j * out->stride + i] =
                    (uint8_t)((up[x] + up[x + 1] + lo[x] + lo[x + 1]) / 4);
        }
    }
}

/* P2  image part of cv::buildOpticalFlowPyramid(left, pyr, Size(w,w), 2) —
 * src/lib/stereo_slam.cpp:137-139. Level l+1 = pyrDown(level l); OpenCV stops
 * early when the next level is not larger than the window. */
int svo_o_build_lk_pyramid(const svo_image *lvl0, int max_levels, int win, svo_image *levels)
{
    levels[0] = *lvl0;
    int w = lvl0->width, h = lvl0->height;
    for (int l = 0; l < max_levels; l++) {
        if (l != 0) {
            levels[l].width = w;
            levels[l].height = h;
            levels[l].stride = w;
            svo_o_pyr_down(levels[l - 1].data, levels[l - 1].width, levels[l - 1].height,
                           levels[l - 1].stride, (uint8_t *)levels[l].data, w);
        }
        w = (w + 1) / 2;
        h = (h + 1) / 2;
        if (w <= win || h <= win) return l + 1;
    }
    return max_levels;
}

/* ======================================================================== */
/* A4  _get_intensity_diff / get_total_intensity_diff —
 * src/lib/image_comparison.cpp:9-91,103-120                                 */
static float intensity_diff(const svo_image *im1, const svo_image *im2,
                            svo_kp2d c1, svo_kp2d c2, int patch_size)
{
    const float half_size = ((float)patch_size - 1.0f) / 2.0f;
    float s1x = c1.x - half_size, s1y = c1.y - half_size;
    float s2x = c2.x - half_size, s2y = c2.y - half_size;
    const int ip1x = (int)floor(s1x), ip1y = (int)floor(s1y);
    const int ip2x = (int)floor(s2x), ip2y = (int)floor(s2y);
    const float x12 = s1x - ip1x, y12 = s1y - ip1y;
    const float x22 = s2x - ip2x, y22 = s2y - ip2y;
    const float x11 = (float)(1.0 - x12), y11 = (float)(1.0 - y12);
    const float x21 = (float)(1.0 - x22), y21 = (float)(1.0 - y22);
    const float m1[4] = { x11 * y11, x12 * y11, x11 * y12, x12 * y12 };
    const float m2[4] = { x21 * y21, x22 * y21, x21 * y22, x22 * y22 };
    float intensity = 0;
    if (ip1y >= 0 && ip1y + patch_size < im1->height &&
        ip2y >= 0 && ip2y + patch_size < im2->height &&
        ip1x >= 0 && ip1x + patch_size < im1->width &&
        ip2x >= 0 && ip2x + patch_size < im2->width) {
        for (int i = 0; i < patch_size; i++) {
            const uint8_t *src11 = im1->data + (size_t)(i + ip1y) * im1->stride + ip1x;
            const uint8_t *src12 = im1->data + (size_t)(i + ip1y + 1) * im1->stride + ip1x;
            const uint8_t *src21 = im2->data + (size_t)(i + ip2y) * im2->stride + ip2x;
            const uint8_t *src22 = im2->data + (size_t)(i + ip2y + 1) * im2->stride + ip2x;
            for (int j = 0; j < patch_size; j++) {
                const float px1[4] = { src11[0], src11[1], src12[0], src12[1] };
                const float px2[4] = { src21[0], src21[1], src22[0], src22[1] };
                float i1 = 0, i2 = 0;
                for (int k = 0; k < 4; k++) i1 += m1[k] * px1[k];
                for (int k = 0; k < 4; k++) i2 += m2[k] * px2[k];
                intensity += fabsf(i1 - i2);
                src11++; src12++; src21++; src22++;
            }
        }
    }
    return intensity;
}

float svo_o_total_intensity_diff(const svo_image *img1, const svo_image *img2,
                                 const svo_kp2d *kps1, const svo_kp2d *kps2, int n, int patch)
{
    float diff = 0;
    for (int i = 0; i < n; i++) diff += intensity_diff(img1, img2, kps1[i], kps2[i], patch);
    return diff;
}

/* A5  get_patch_sum — src/lib/pose_estimator.cpp:82-112                     */
static float patch_sum(const svo_image *im, float cx, float cy)
{
    const float sx = cx - 0.5f, sy = cy - 0.5f;
    const int ipx = (int)floor(sx), ipy = (int)floor(sy);
    const float x2 = sx - ipx, y2 = sy - ipy;
    const float x1 = (float)(1.0 - x2), y1 = (float)(1.0 - y2);
    const uint8_t *src1 = im->data + (size_t)ipy * im->stride + ipx;
    const uint8_t *src2 = src1 + im->stride;
    const uint8_t *src3 = src2 + im->stride;
    float intensity = x1 * y1 * src1[0] + y1 * src1[1] + x2 * y1 * src1[2] +
                      x1 * src2[0] + src2[1] + x2 * src2[2] +
                      x1 * y2 * src3[0] + y2 * src3[1] + x2 * y2 * src3[2];
    return intensity;
}

/* the 2x6 Jacobian of src/lib/pose_estimator.cpp:343-344 (and
 * src/lib/pose_refinement.cpp:380-381)                                      */
static void pose_jacobian(float fx, float fy, float x, float y, float z, float J[12])
{
    J[0] = -fx / z;  J[1] = 0;        J[2] = fx * x / (z * z);
    J[3] = fx * x * y / (z * z);      J[4] = -fx * (1 + (x * x) / (z * z)); J[5] = fx * y / z;
    J[6] = 0;        J[7] = -fy / z;  J[8] = fy * y / (z * z);
    J[9] = fy * (1 + (y * y) / (z * z)); J[10] = -fy * x * y / (z * z);     J[11] = -fy * x / z;
}

static void mat33f_vec3(const float *a, const float *v, float *out)
{
    float t[3];
    for (int i = 0; i < 3; i++) {
        float s = 0;
        for (int k = 0; k < 3; k++) s += a[i * 3 + k] * v[k];
        t[i] = s;
    }
    out[0] = t[0]; out[1] = t[1]; out[2] = t[2];
}

typedef struct sia_state {
    const svo_image *prev_pyr, *cur_pyr;
    const svo_camera_settings *cam;
    svo_camera_settings lcam;
    int level, n;
    svo_kp2d *kps2d;      /* active, full resolution      */
    svo_kp3d *kps3d;      /* active                       */
    svo_kp2d *lkps;       /* active, level resolution     */
    svo_kp2d *proj;       /* scratch                      */
    float *gtj;           /* n*16*6 gradient_times_jacobians */
    float *diffs;         /* n*16                         */
    float hessian[36], inv_hessian[36], residual[6];
    int n_gradient, n_cost;
} sia_state;

/* A2  setLevel — src/lib/pose_estimator.cpp:541-562 */
static void sia_set_level(sia_state *s, int level)
{
    s->level = level;
    const int divider = 1 << level;
    s->lcam = *s->cam;
    s->lcam.fx /= divider; s->lcam.fy /= divider;
    s->lcam.cx /= divider; s->lcam.cy /= divider;
    s->lcam.baseline /= divider;
    memcpy(s->lkps, s->kps2d, sizeof(svo_kp2d) * (size_t)s->n);
    if (level == 0) return;
    for (int i = 0; i < s->n; i++) { s->lkps[i].x /= divider; s->lkps[i].y /= divider; }
}

/* do_calc — src/lib/pose_estimator.cpp:275-300 */
static float sia_do_calc(sia_state *s, const float pose[6])
{
    s->n_cost++;
    svo_o_project_keypoints(pose, s->kps3d, s->n, &s->lcam, s->proj);
    return svo_o_total_intensity_diff(&s->prev_pyr[s->level], &s->cur_pyr[s->level],
                                      s->lkps, s->proj, s->n,
                                      s->lcam.window_size_pose_estimator);
}

/* A6  calculate_hessian — src/lib/pose_estimator.cpp:312-416 */
static void sia_calculate_hessian(sia_state *s, const float pose[6])
{
    const svo_image *prev = &s->prev_pyr[s->level];
    const int PATCH_SIZE = 4;
    float rot[9], inv_rot[9];
    svo_o_pose_matrices(pose, rot, inv_rot);
    const float fx = s->lcam.fx, fy = s->lcam.fy;
    for (int i = 0; i < s->n; i++) {
        float kx = s->lkps[i].x, ky = s->lkps[i].y;
        kx -= PATCH_SIZE / 2; ky -= PATCH_SIZE / 2;
        float kp[3] = { s->kps3d[i].x - pose[0], s->kps3d[i].y - pose[1], s->kps3d[i].z - pose[2] };
        mat33f_vec3(inv_rot, kp, kp);
        float J[12];
        pose_jacobian(fx, fy, kp[0], kp[1], kp[2], J);
        float *it = s->gtj + (size_t)i * 16 * 6;
        for (int r = 0; r < PATCH_SIZE; r++) {
            for (int c = 0; c < PATCH_SIZE; c++, it += 6) {
                if ((kx - 2.0) < 0 || (ky - 2.0) < 0 || (kx + 3.0) >= prev->width ||
                    (ky + 3.0) >= prev->height) {
                    for (int k = 0; k < 6; k++) it[k] = 0;
                    kx++;
                    continue;
                }
                const float int1 = patch_sum(prev, kx + 1, ky);
                const float int2 = patch_sum(prev, kx - 1, ky);
                const float int3 = patch_sum(prev, kx, ky + 1);
                const float int4 = patch_sum(prev, kx, ky - 1);
                const float g0 = int1 - int2, g1 = int3 - int4;
                for (int k = 0; k < 6; k++) {
                    float sum = 0;
                    sum += g0 * J[k];
                    sum += g1 * J[6 + k];
                    it[k] = sum;
                }
                kx++;
            }
            kx -= PATCH_SIZE;
            ky++;
        }
    }
    for (int k = 0; k < 36; k++) s->hessian[k] = 0;
    const float *it = s->gtj;
    for (int i = 0; i < s->n * 16; i++, it += 6)
        for (int a = 0; a < 6; a++)
            for (int b = 0; b < 6; b++) s->hessian[a * 6 + b] += it[a] * it[b];
    svo_o_inv_svd(s->hessian, 6, s->inv_hessian);
}

/* A7  get_gradient — src/lib/pose_estimator.cpp:418-539. The member `hessian`
 * is never assigned (the local of :399 shadows it), so calculate_hessian runs
 * on every call with the current pose. */
static void sia_get_gradient(sia_state *s, const float pose[6], float grad[6])
{
    const int PATCH_SIZE = 4;
    const svo_image *cur = &s->cur_pyr[s->level];
    const svo_image *prev = &s->prev_pyr[s->level];
    s->n_gradient++;
    sia_calculate_hessian(s, pose);
    svo_o_project_keypoints(pose, s->kps3d, s->n, &s->lcam, s->proj);

    const int half_patch_size = PATCH_SIZE / 2;
    float *diff = s->diffs;
    for (int i = 0; i < s->n; i++) {
        float kx = s->proj[i].x - half_patch_size, ky = s->proj[i].y - half_patch_size;
        float rx = s->lkps[i].x - half_patch_size, ry = s->lkps[i].y - half_patch_size;
        for (int r = 0; r < PATCH_SIZE; r++) {
            for (int c = 0; c < PATCH_SIZE; c++, kx++, rx++, diff++) {
                if (!((rx - 1.0) < 0 || (kx - 1.0) < 0 ||
                      (ry - 1.0) < 0 || (ky - 1.0) < 0 ||
                      (rx + 2.0) > prev->width || (kx + 2.0) > cur->width ||
                      (ry + 2.0) > prev->height || (ky + 2.0) > cur->height)) {
                    const float int1 = patch_sum(prev, rx, ry);
                    const float int2 = patch_sum(cur, kx, ky);
                    *diff = int2 - int1;
                } else
                    *diff = 0;
            }
            ky++; ry++;
            kx -= PATCH_SIZE; rx -= PATCH_SIZE;
        }
    }

    float res[6] = { 0, 0, 0, 0, 0, 0 };
    for (int i = 0; i < s->n * 16; i++) {
        const float *g = s->gtj + (size_t)i * 6;
        for (int k = 0; k < 6; k++) res[k] -= g[k] * s->diffs[i];
    }
    memcpy(s->residual, res, sizeof(res));

    float delta[6];
    for (int a = 0; a < 6; a++) {
        float sum = 0;
        for (int b = 0; b < 6; b++) sum += s->inv_hessian[a * 6 + b] * res[b];
        delta[a] = sum;
    }
    float pg[6];
    svo_o_exponential_map(delta, pg);
    float rot[9], inv_rot[9];
    svo_o_pose_matrices(pose, rot, inv_rot);
    mat33f_vec3(rot, pg, grad);
    mat33f_vec3(rot, pg + 3, grad + 3);
}

/* A9  estimate_pose_at_level — src/lib/pose_estimator.cpp:166-222.
 * NB the loop counter i is shared by the outer and the inner loop. */
static float sia_estimate_level(sia_state *s, const float guess[6], float out[6], int level,
                                svo_gn_trace *tr)
{
    const int maxIter = 50;
    sia_set_level(s, level);
    float x0[6];
    memcpy(x0, guess, sizeof(x0));
    const int g0 = s->n_gradient, c0 = s->n_cost;
    float prev_cost = sia_do_calc(s, x0);
    const float initial = prev_cost;
    int accepted = 0, exit_small = 0;
    int i;
    for (i = 0; i < maxIter; i++) {
        float gradient[6];
        sia_get_gradient(s, x0, gradient);
        float k = 1.0f;
        for (; i < maxIter; i++) {
            float x[6];
            for (int j = 0; j < 6; j++) x[j] = x0[j] + k * gradient[j];
            const float new_cost = sia_do_calc(s, x);
            if (new_cost < prev_cost) {
                memcpy(x0, x, sizeof(x0));
                prev_cost = new_cost;
                accepted++;
                break;
            } else if (fabs(new_cost - prev_cost) < 1.0) {
                i = maxIter;
                exit_small = 1;
                break;
            } else
                k /= 2;
        }
    }
    memcpy(out, x0, sizeof(x0));
    if (tr) {
        tr->level = level;
        tr->n_gradient = s->n_gradient - g0;
        tr->n_cost = s->n_cost - c0;
        tr->n_accepted = accepted;
        tr->exit_small = exit_small;
        tr->initial_cost = initial;
        tr->final_cost = prev_cost;
        memcpy(tr->pose, x0, sizeof(x0));
    }
    return prev_cost;
}

static void sia_init(sia_state *s, const svo_image *prev_pyr, const svo_image *cur_pyr,
                     const svo_kp2d *kps2d, const svo_kp3d *kps3d, const uint32_t *flags,
                     int n, const svo_camera_settings *cam)
{
    memset(s, 0, sizeof(*s));
    s->prev_pyr = prev_pyr; s->cur_pyr = cur_pyr; s->cam = cam;
    const size_t cap = (size_t)(n > 0 ? n : 1);
    s->kps2d = (svo_kp2d *)malloc(sizeof(svo_kp2d) * cap);
    s->kps3d = (svo_kp3d *)malloc(sizeof(svo_kp3d) * cap);
    s->lkps = (svo_kp2d *)malloc(sizeof(svo_kp2d) * cap);
    s->proj = (svo_kp2d *)malloc(sizeof(svo_kp2d) * cap);
    s->gtj = (float *)malloc(sizeof(float) * cap * 16 * 6);
    s->diffs = (float *)malloc(sizeof(float) * cap * 16);
    /* A1 PoseEstimatorCallback ctor — src/lib/pose_estimator.cpp:226-259 */
    int m = 0;
    for (int i = 0; i < n; i++) {
        if (flags && (flags[i] & SVO_IGNORE_TEMPORARY)) continue;
        s->kps2d[m] = kps2d[i];
        s->kps3d[m] = kps3d[i];
        m++;
    }
    s->n = m;
}

static void sia_free(sia_state *s)
{
    free(s->kps2d); free(s->kps3d); free(s->lkps); free(s->proj); free(s->gtj); free(s->diffs);
}

/* PoseEstimator::estimate_pose — src/lib/pose_estimator.cpp:115-130 */
float svo_o_sparse_align(const svo_image *prev_pyr, const svo_image *cur_pyr,
                         const svo_kp2d *kps2d, const svo_kp3d *kps3d,
                         const uint32_t *flags, int n, const svo_camera_settings *cam,
                         const float pose_guess[6], float pose_out[6], svo_gn_trace *trace)
{
    sia_state s;
    sia_init(&s, prev_pyr, cur_pyr, kps2d, kps3d, flags, n, cam);
    float est[6], err = 0;
    memcpy(est, pose_guess, sizeof(est));
    if (trace) memset(trace, 0, sizeof(svo_gn_trace) * (size_t)cam->max_pyramid_levels);
    for (int i = cam->max_pyramid_levels; i > cam->min_pyramid_level_pose_estimation; i--) {
        float ne[6];
        const int level = i - 1;
        err = sia_estimate_level(&s, est, ne, level, trace ? &trace[level] : NULL);
        memcpy(est, ne, sizeof(est));
    }
    memcpy(pose_out, est, sizeof(est));
    sia_free(&s);
    return err;
}

void svo_o_sia_gradient(const svo_image *prev, const svo_image *cur, int level,
                        const svo_kp2d *kps2d, const svo_kp3d *kps3d,
                        const uint32_t *flags, int n, const svo_camera_settings *cam,
                        const float pose[6], float H[36], float b[6], float step[6])
{
    /* prev/cur are the level images themselves; build 1-entry "pyramids"
     * indexed at [level] */
    svo_image pp[SVO_MAX_PYRAMID_LEVELS], cp[SVO_MAX_PYRAMID_LEVELS];
    memset(pp, 0, sizeof(pp)); memset(cp, 0, sizeof(cp));
    pp[level] = *prev; cp[level] = *cur;
    sia_state s;
    sia_init(&s, pp, cp, kps2d, kps3d, flags, n, cam);
    sia_set_level(&s, level);
    sia_get_gradient(&s, pose, step);
    memcpy(H, s.hessian, sizeof(float) * 36);
    memcpy(b, s.residual, sizeof(float) * 6);
    sia_free(&s);
}

/* ======================================================================== */
/* B2  cv::calcOpticalFlowPyrLK (OpenCV modules/video/src/lkpyramid.cpp,
 * LKTrackerInvoker) as called from src/lib/optical_flow.cpp:41-44.
 * Restated choices (PARITY UNPINNED):
 *  - the sums A11/A12/A22/b1/b2 are accumulated as exact 64-bit integers and
 *    converted to float once (OpenCV accumulates the integer products in
 *    float lanes, so its low bits depend on the SIMD width of the build);
 *  - everything else (14-bit bilinear weights, 5 fractional bits of the
 *    interpolated image, 2^-20 scale, min-eigenvalue test, 30 iterations,
 *    eps 0.01, oscillation back-off, error = mean |diff|/32) as in OpenCV.   */

void svo_oi_lkpyr_build(svo_oi_lkpyr *p, const svo_image *levels, int n_levels, int win)
{
    p->n_levels = n_levels;
    p->win = win;
    for (int l = 0; l < n_levels; l++) {
        const svo_image *im = &levels[l];
        const int w = im->width, h = im->height, pw = w + 2 * win, ph = h + 2 * win;
        p->w[l] = w; p->h[l] = h; p->pstride[l] = pw;
        p->img[l] = (uint8_t *)malloc((size_t)pw * ph);
        p->deriv[l] = (int16_t *)calloc((size_t)pw * ph * 2, sizeof(int16_t));
        /* copyMakeBorder(.., BORDER_REFLECT_101) */
        for (int y = 0; y < ph; y++) {
            int sy = y - win;
            if (sy < 0 || sy >= h) {
                if (h == 1) sy = 0;
                else do { if (sy < 0) sy = -sy; else sy = 2 * h - 2 - sy; } while (sy < 0 || sy >= h);
            }
            for (int x = 0; x < pw; x++) {
                int sx = x - win;
                if (sx < 0 || sx >= w) {
                    if (w == 1) sx = 0;
                    else do { if (sx < 0) sx = -sx; else sx = 2 * w - 2 - sx; } while (sx < 0 || sx >= w);
                }
                p->img[l][(size_t)y * pw + x] = im->data[(size_t)sy * im->stride + sx];
            }
        }
        /* calcSharrDeriv on the image interior, constant-0 border */
        int16_t *d = (int16_t *)malloc(sizeof(int16_t) * (size_t)w * h * 2);
        svo_o_scharr(im->data, w, h, im->stride, d);
        for (int y = 0; y < h; y++)
            memcpy(p->deriv[l] + ((size_t)(y + win) * pw + win) * 2, d + (size_t)y * w * 2,
                   sizeof(int16_t) * (size_t)w * 2);
        free(d);
    }
}

void svo_oi_lkpyr_free(svo_oi_lkpyr *p)
{
    for (int l = 0; l < p->n_levels; l++) { free(p->img[l]); free(p->deriv[l]); }
    memset(p, 0, sizeof(*p));
}

#define LK_DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

/* diagnostics: LK iterations executed since the last reset (not part of any result) */
static long long g_lk_iterations = 0, g_lk_points = 0;
static int g_lk_hist[4096]; static int g_lk_cur = 0;
int svo_o_lk_hist(int *out, int cap) { int n = g_lk_cur < cap ? g_lk_cur : cap; for (int i = 0; i < n; i++) out[i] = g_lk_hist[i]; g_lk_cur = 0; memset(g_lk_hist, 0, sizeof(g_lk_hist)); return n; }
long long svo_o_lk_iterations(int reset)
{
    long long v = g_lk_iterations;
    if (reset) { g_lk_iterations = 0; g_lk_points = 0; }
    return v;
}

static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_floor_f(float v) { return (int)floorf(v); }

void svo_oi_klt_track(const svo_oi_lkpyr *P, const svo_oi_lkpyr *N, const svo_kp2d *prev_pts,
                      svo_kp2d *next_pts, int npts, uint8_t *status, float *err)
{
    const int win = P->win;
    const int maxLevel = (P->n_levels < N->n_levels ? P->n_levels : N->n_levels) - 1;
    const int maxCount = 30;
    double epsilon = 0.01;
    epsilon *= epsilon;
    const double minEigThreshold = 1e-4;
    const float halfWin = (win - 1) * 0.5f;
    const int W_BITS = 14, W_BITS1 = 14;
    const float FLT_SCALE = 1.f / (1 << 20);
    int16_t *IWinBuf = (int16_t *)malloc(sizeof(int16_t) * (size_t)win * win * 3);
    int16_t *derivIWinBuf = IWinBuf + win * win;

    for (int i = 0; i < npts; i++) { status[i] = 1; err[i] = 0; }

    for (int level = maxLevel; level >= 0; level--) {
        const int cols = P->w[level], rows = P->h[level];
        const int Jcols = N->w[level], Jrows = N->h[level];
        const int stepI = P->pstride[level], stepJ = N->pstride[level], dstep = stepI * 2;
        const uint8_t *I0 = P->img[level] + (size_t)win * stepI + win;
        const uint8_t *J0 = N->img[level] + (size_t)win * stepJ + win;
        const int16_t *D0 = P->deriv[level] + ((size_t)win * stepI + win) * 2;

        for (int ptidx = 0; ptidx < npts; ptidx++) {
            const float lscale = (float)(1. / (1 << level));
            float prevx = prev_pts[ptidx].x * lscale, prevy = prev_pts[ptidx].y * lscale;
            float nextx, nexty;
            if (level == maxLevel) {
                nextx = next_pts[ptidx].x * lscale;    /* OPTFLOW_USE_INITIAL_FLOW */
                nexty = next_pts[ptidx].y * lscale;
            } else {
                nextx = next_pts[ptidx].x * 2.f;
                nexty = next_pts[ptidx].y * 2.f;
            }
            next_pts[ptidx].x = nextx; next_pts[ptidx].y = nexty;

            prevx -= halfWin; prevy -= halfWin;
            const int iprevx = cv_floor_f(prevx), iprevy = cv_floor_f(prevy);
            if (iprevx < -win || iprevx >= cols || iprevy < -win || iprevy >= rows) {
                if (level == 0) { status[ptidx] = 0; err[ptidx] = 0; }
                continue;
            }
            float a = prevx - iprevx, b = prevy - iprevy;
            int iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
            int iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
            int iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
            int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            int64_t iA11 = 0, iA12 = 0, iA22 = 0;

            for (int y = 0; y < win; y++) {
                const uint8_t *src = I0 + (ptrdiff_t)(y + iprevy) * stepI + iprevx;
                const int16_t *dsrc = D0 + (ptrdiff_t)(y + iprevy) * dstep + iprevx * 2;
                int16_t *Iptr = IWinBuf + y * win;
                int16_t *dIptr = derivIWinBuf + y * win * 2;
                for (int x = 0; x < win; x++, dsrc += 2, dIptr += 2) {
                    const int ival = LK_DESCALE(src[x] * iw00 + src[x + 1] * iw01 +
                                                src[x + stepI] * iw10 + src[x + stepI + 1] * iw11,
                                                W_BITS1 - 5);
                    const int ixval = LK_DESCALE(dsrc[0] * iw00 + dsrc[2] * iw01 +
                                                 dsrc[dstep] * iw10 + dsrc[dstep + 2] * iw11, W_BITS1);
                    const int iyval = LK_DESCALE(dsrc[1] * iw00 + dsrc[3] * iw01 +
                                                 dsrc[dstep + 1] * iw10 + dsrc[dstep + 3] * iw11, W_BITS1);
                    Iptr[x] = (int16_t)ival;
                    dIptr[0] = (int16_t)ixval;
                    dIptr[1] = (int16_t)iyval;
                    iA11 += (int64_t)ixval * ixval;
                    iA12 += (int64_t)ixval * iyval;
                    iA22 += (int64_t)iyval * iyval;
                }
            }
            float A11 = (float)iA11 * FLT_SCALE, A12 = (float)iA12 * FLT_SCALE,
                  A22 = (float)iA22 * FLT_SCALE;
            float D = A11 * A22 - A12 * A12;
            const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                                 (2 * win * win);
            if (minEig < minEigThreshold || D < FLT_EPSILON) {
                if (level == 0) status[ptidx] = 0;
                continue;
            }
            D = 1.f / D;
            nextx -= halfWin; nexty -= halfWin;
            float prevDx = 0, prevDy = 0;

            for (int j = 0; j < maxCount; j++) {
                g_lk_iterations++;
                if (ptidx < 4096) { g_lk_hist[ptidx]++; if (ptidx + 1 > g_lk_cur) g_lk_cur = ptidx + 1; }
                const int inextx = cv_floor_f(nextx), inexty = cv_floor_f(nexty);
                if (inextx < -win || inextx >= Jcols || inexty < -win || inexty >= Jrows) {
                    if (level == 0) status[ptidx] = 0;
                    break;
                }
                a = nextx - inextx; b = nexty - inexty;
                iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
                iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
                iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                int64_t ib1 = 0, ib2 = 0;
                for (int y = 0; y < win; y++) {
                    const uint8_t *Jptr = J0 + (ptrdiff_t)(y + inexty) * stepJ + inextx;
                    const int16_t *Iptr = IWinBuf + y * win;
                    const int16_t *dIptr = derivIWinBuf + y * win * 2;
                    for (int x = 0; x < win; x++, dIptr += 2) {
                        const int diff = LK_DESCALE(Jptr[x] * iw00 + Jptr[x + 1] * iw01 +
                                                    Jptr[x + stepJ] * iw10 + Jptr[x + stepJ + 1] * iw11,
                                                    W_BITS1 - 5) - Iptr[x];
                        ib1 += (int64_t)diff * dIptr[0];
                        ib2 += (int64_t)diff * dIptr[1];
                    }
                }
                const float b1 = (float)ib1 * FLT_SCALE, b2 = (float)ib2 * FLT_SCALE;
                const float dx = (float)((A12 * b2 - A22 * b1) * D);
                const float dy = (float)((A12 * b1 - A11 * b2) * D);
                nextx += dx; nexty += dy;
                next_pts[ptidx].x = nextx + halfWin;
                next_pts[ptidx].y = nexty + halfWin;
                if ((double)dx * dx + (double)dy * dy <= epsilon) break;
                if (j > 0 && fabsf(dx + prevDx) < 0.01 && fabsf(dy + prevDy) < 0.01) {
                    next_pts[ptidx].x -= dx * 0.5f;
                    next_pts[ptidx].y -= dy * 0.5f;
                    break;
                }
                prevDx = dx; prevDy = dy;
            }

            if (status[ptidx] && level == 0) {
                const float npx = next_pts[ptidx].x - halfWin, npy = next_pts[ptidx].y - halfWin;
                const int inx = cv_floor_f(npx), iny = cv_floor_f(npy);
                if (inx < -win || inx >= Jcols || iny < -win || iny >= Jrows) {
                    status[ptidx] = 0;
                    continue;
                }
                const float aa = npx - inx, bb = npy - iny;
                iw00 = cv_round_f((1.f - aa) * (1.f - bb) * (1 << W_BITS));
                iw01 = cv_round_f(aa * (1.f - bb) * (1 << W_BITS));
                iw10 = cv_round_f((1.f - aa) * bb * (1 << W_BITS));
                iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
                float errval = 0.f;
                for (int y = 0; y < win; y++) {
                    const uint8_t *Jptr = J0 + (ptrdiff_t)(y + iny) * stepJ + inx;
                    const int16_t *Iptr = IWinBuf + y * win;
                    for (int x = 0; x < win; x++) {
                        const int diff = LK_DESCALE(Jptr[x] * iw00 + Jptr[x + 1] * iw01 +
                                                    Jptr[x + stepJ] * iw10 + Jptr[x + stepJ + 1] * iw11,
                                                    W_BITS1 - 5) - Iptr[x];
                        errval += fabsf((float)diff);
                    }
                }
                err[ptidx] = errval * 1.f / (32 * win * win);
            }
        }
    }
    /* src/lib/optical_flow.cpp:46-50 */
    for (int i = 0; i < npts; i++)
        if (status[i] == 0) err[i] = INFINITY;
    free(IWinBuf);
}

void svo_o_klt_track(const svo_image *prev_lk, const svo_image *cur_lk, int n_levels,
                     const svo_kp2d *prev_pts, svo_kp2d *cur_pts, int n, int win,
                     uint8_t *status, float *err)
{
    svo_oi_lkpyr P, N;
    svo_oi_lkpyr_build(&P, prev_lk, n_levels, win);
    svo_oi_lkpyr_build(&N, cur_lk, n_levels, win);
    svo_oi_klt_track(&P, &N, prev_pts, cur_pts, n, status, err);
    svo_oi_lkpyr_free(&P);
    svo_oi_lkpyr_free(&N);
}

/* B1  merge step of PoseRefiner::refine_pose — src/lib/pose_refinement.cpp:125-150
 * (the reverse iteration / pop_back there is bookkeeping: entry i of the
 * frame always meets its own tracked position and error). */
void svo_o_refine_merge(svo_kp2d *kps2d, uint32_t *flags, const svo_kp2d *tracked,
                        const float *err, int n)
{
    for (int i = n; i > 0; i--) {
        const int j = i - 1;
        const float dx = kps2d[j].x - tracked[j].x, dy = kps2d[j].y - tracked[j].y;
        const float diff = dx * dx + dy * dy;
        if (err[j] > 20)
            flags[j] |= SVO_IGNORE_COMPLETELY;
        else if (diff > 81)
            flags[j] |= SVO_IGNORE_DURING_REFINEMENT;
        else {
            flags[j] &= ~(uint32_t)SVO_IGNORE_DURING_REFINEMENT;
            kps2d[j] = tracked[j];
        }
    }
}

/* B3  PoseRefinerCallback::do_calc — src/lib/pose_refinement.cpp:321-348 */
static float reproj_do_calc(const svo_kp2d *kps2d, const svo_kp3d *kps3d, const uint32_t *flags,
                            int n, const svo_camera_settings *cam, const float pose[6],
                            svo_kp2d *proj)
{
    svo_o_project_keypoints(pose, kps3d, n, cam, proj);
    float tot = 0;
    for (int i = 0; i < n; i++) {
        if (flags[i] & (SVO_IGNORE_DURING_REFINEMENT | SVO_IGNORE_COMPLETELY | SVO_IGNORE_TEMPORARY))
            continue;
        const float d0 = fabsf(proj[i].x - kps2d[i].x), d1 = fabsf(proj[i].y - kps2d[i].y);
        tot += d0 + d1;
    }
    return tot;
}

/* PoseRefinerCallback::get_gradient — src/lib/pose_refinement.cpp:350-412 */
static void reproj_get_gradient(const svo_kp2d *kps2d, const svo_kp3d *kps3d,
                                const uint32_t *flags, int n, const svo_camera_settings *cam,
                                const float pose[6], svo_kp2d *proj, float grad[6])
{
    svo_o_project_keypoints(pose, kps3d, n, cam, proj);
    float err[6] = { 0 }, H[36] = { 0 };
    float rot[9], inv_rot[9];
    svo_o_pose_matrices(pose, rot, inv_rot);
    const float fx = cam->fx, fy = cam->fy;
    for (int i = 0; i < n; i++) {
        float kp[3] = { kps3d[i].x - pose[0], kps3d[i].y - pose[1], kps3d[i].z - pose[2] };
        mat33f_vec3(inv_rot, kp, kp);
        if (flags[i] & (SVO_IGNORE_DURING_REFINEMENT | SVO_IGNORE_COMPLETELY | SVO_IGNORE_TEMPORARY))
            continue;
        float J[12];
        pose_jacobian(fx, fy, kp[0], kp[1], kp[2], J);
        const float d0 = kps2d[i].x - proj[i].x, d1 = kps2d[i].y - proj[i].y;
        if ((fabs(d0) > 3.0) || (fabs(d1) > 3.0)) continue;
        for (int a = 0; a < 6; a++)
            for (int b = 0; b < 6; b++) {
                float s = 0;
                s += J[a] * J[b];
                s += J[6 + a] * J[6 + b];
                H[a * 6 + b] += s;
            }
        for (int a = 0; a < 6; a++) {
            float s = 0;
            s += J[a] * d0;
            s += J[6 + a] * d1;
            err[a] += s;
        }
    }
    float Hinv[36], twist[6];
    svo_o_inv_svd(H, 6, Hinv);
    for (int a = 0; a < 6; a++) {
        float s = 0;
        for (int b = 0; b < 6; b++) s += Hinv[a * 6 + b] * err[b];
        twist[a] = s;
    }
    svo_o_exponential_map(twist, grad);
}

/* PoseRefiner::update_pose — src/lib/pose_refinement.cpp:236-290 */
float svo_o_reproj_gn(const svo_kp2d *kps2d, const svo_kp3d *kps3d, const uint32_t *flags,
                      int n, const svo_camera_settings *cam, const float pose_in[6],
                      float pose_out[6], svo_gn_trace *tr)
{
    const int maxIter = 50;
    svo_kp2d *proj = (svo_kp2d *)malloc(sizeof(svo_kp2d) * (size_t)(n > 0 ? n : 1));
    float x0[6];
    memcpy(x0, pose_in, sizeof(x0));
    int n_grad = 0, n_cost = 1, accepted = 0, exit_small = 0;
    float prev_cost = reproj_do_calc(kps2d, kps3d, flags, n, cam, x0, proj);
    const float initial = prev_cost;
    int i;
    for (i = 0; i < maxIter; i++) {
        float gradient[6];
        reproj_get_gradient(kps2d, kps3d, flags, n, cam, x0, proj, gradient);
        n_grad++;
        float k = 1.0f;
        for (; i < maxIter; i++) {
            float x[6];
            for (int j = 0; j < 6; j++) x[j] = x0[j] + k * gradient[j];
            const float new_cost = reproj_do_calc(kps2d, kps3d, flags, n, cam, x, proj);
            n_cost++;
            if (new_cost < prev_cost) {
                memcpy(x0, x, sizeof(x0));
                prev_cost = new_cost;
                accepted++;
                break;
            } else if (fabs(new_cost - prev_cost) < 0.0001) {
                i = maxIter;
                exit_small = 1;
                break;
            } else
                k /= 2;
        }
    }
    memcpy(pose_out, x0, sizeof(x0));
    if (tr) {
        tr->level = 0; tr->n_gradient = n_grad; tr->n_cost = n_cost; tr->n_accepted = accepted;
        tr->exit_small = exit_small; tr->initial_cost = initial; tr->final_cost = prev_cost;
        memcpy(tr->pose, x0, sizeof(x0));
    }
    free(proj);
    return prev_cost;
}

/* ======================================================================== */
/* C1  DepthFilter::calculate_disparities — src/lib/depth_filter.cpp:259-327;
 * same loop in DepthCalculator::calculate_depth — src/lib/depth_calculator.cpp:200-240.
 * cv::matchTemplate(TM_SQDIFF) is restated as the exact integer SSD rounded
 * once to float (OpenCV goes through a DFT cross-correlation whose rounding
 * noise is not reproducible; PARITY UNPINNED). cv::minMaxLoc returns the
 * first minimum in row-major order. */
void svo_o_ssd_disparity(const svo_image *left, const svo_image *right,
                         const svo_kp2d *kps2d, int n, int win, int search_x, int search_y,
                         int clamp_half, float *disparity)
{
    const int window_before = win / 2, window_after = (win + 1) / 2;
    const int cols = left->width, rows = left->height;
    const int mw_max = search_x + 1 + win, mh_max = 2 * search_y + 1 + win;
    int32_t *match = (int32_t *)malloc(sizeof(int32_t) * (size_t)mw_max * mh_max);
    for (int i = 0; i < n; i++) {
        disparity[i] = -1;
        const int x = (int)kps2d[i].x, y = (int)kps2d[i].y;
        const int x11 = x - window_before > 0 ? x - window_before : 0;
        const int x12 = cols - 1 < x + window_after ? cols - 1 : x + window_after;
        const int y11 = y - window_before > 0 ? y - window_before : 0;
        const int y12 = rows < y + window_after ? rows : y + window_after;
        if (clamp_half && (x12 <= 0 || y12 <= 0 || x11 >= cols - 1 || y11 >= rows - 1)) continue;
        const int x21 = x11;
        const int x22 = cols - 1 < x + window_after + search_x ? cols - 1 : x + window_after + search_x;
        const int y21 = y - window_before - search_y > 0 ? y - window_before - search_y : 0;
        const int y22 = rows - 1 < y + window_after + search_y ? rows - 1 : y + window_after + search_y;
        if (clamp_half && (x22 <= 0 || y22 <= 0 || x21 >= cols - 1 || y21 >= rows - 1)) continue;
        const int tw = x12 - x11, th = y12 - y11, rw = x22 - x21, rh = y22 - y21;
        const int mw = rw - tw + 1, mh = rh - th + 1;
        if (tw <= 0 || th <= 0 || mw <= 0 || mh <= 0) continue; /* cv would throw */
        for (int k = 0; k < mh; k++)
            for (int j = 0; j < mw; j++) match[k * mw + j] = 0;
        /* exact integer SSD; loop order chosen so that the innermost loop runs over
         * the contiguous offsets j and vectorises (the order of integer adds is free) */
        for (int k = 0; k < mh; k++)
            for (int r = 0; r < th; r++) {
                const uint8_t *t = left->data + (size_t)(y11 + r) * left->stride + x11;
                const uint8_t *q = right->data + (size_t)(y21 + k + r) * right->stride + x21;
                int32_t *restrict m = match + k * mw;
                for (int c = 0; c < tw; c++) {
                    const int tv = t[c];
                    const uint8_t *restrict qq = q + c;
                    for (int j = 0; j < mw; j++) {
                        const int d = (int)qq[j] - tv;
                        m[j] += d * d;
                    }
                }
            }
        /* minMaxLoc on the float map */
        float minVal = (float)match[0];
        int minx = 0, miny = 0;
        for (int k = 0; k < mh; k++)
            for (int j = 0; j < mw; j++) {
                const float v = (float)match[k * mw + j];
                if (v < minVal) { minVal = v; minx = j; miny = k; }
            }
        float minPos = 0;
        int matches = 0;
        for (int j = minx; j < mw; j++)
            for (int k = miny; k < mh; k++)
                if ((float)match[k * mw + j] <= (double)minVal) { minPos += j; matches++; }
        minPos = minPos / matches;
        disparity[i] = clamp_half ? (0.5f > minPos ? 0.5f : minPos) : minPos;
    }
    free(match);
}

/* C2  DepthFilter::outlier_check — src/lib/depth_filter.cpp:52-128 */
void svo_o_outlier_check(const svo_kp2d *kps2d, const float *disparity, int n,
                         const svo_camera_settings *cam, const float frame_pose[6],
                         const svo_kp3d *ref3d, const float *kf_pose,
                         int32_t *outlier_count, int32_t *inlier_count)
{
    const float fx = cam->fx, fy = cam->fy, cx = cam->cx, cy = cam->cy, baseline = cam->baseline;
    float rot[9], inv_rot[9];
    svo_o_pose_matrices(frame_pose, rot, inv_rot);
    for (int i = 0; i < n; i++) {
        const float d = disparity[i];
        const float _z = baseline / (d > 0.5f ? d : 0.5f);
        const float _x = (kps2d[i].x - cx) / fx * _z;
        const float _y = (kps2d[i].y - cy) / fy * _z;
        float p[3] = { _x, _y, _z };
        mat33f_vec3(rot, p, p);
        p[0] += frame_pose[0]; p[1] += frame_pose[1]; p[2] += frame_pose[2];

        float krot[9], kinv[9];
        svo_o_pose_matrices(kf_pose + (size_t)i * 6, krot, kinv);
        const float *t = kf_pose + (size_t)i * 6;
        float a[3] = { p[0] - t[0], p[1] - t[1], p[2] - t[2] };
        mat33f_vec3(kinv, a, a);
        float r[3] = { ref3d[i].x - t[0], ref3d[i].y - t[1], ref3d[i].z - t[2] };
        mat33f_vec3(kinv, r, r);
        const float disp_ref = baseline / r[2];
        const float disp = baseline / a[2];
        const float pixel_distance = disp - disp_ref;
        const float deviation = 0.5f;
        if (fabsf(pixel_distance) > 5 * deviation) outlier_count[i]++;
        else inlier_count[i]++;
    }
}

/* D1  DepthFilter::update_kps3d — src/lib/depth_filter.cpp:130-257 */
void svo_o_update_kps3d(const svo_kp2d *kps2d, svo_kp3d *kps3d, const uint32_t *flags, int n,
                        const svo_camera_settings *cam, const float frame_pose[6],
                        const svo_kp2d *ref2d, const float *kf_pose,
                        int32_t *outlier_count, float *kf_inv_depth, float *kf_variance)
{
    const float fx = cam->fx, fy = cam->fy, cx = cam->cx, cy = cam->cy;
    float frot[9], finv[9];
    svo_o_pose_matrices(frame_pose, frot, finv);
    for (int i = 0; i < n; i++) {
        const float *kp = kf_pose + (size_t)i * 6;
        float krot[9], kinv[9];
        svo_o_pose_matrices(kp, krot, kinv);
        const float c1[3] = { kp[0], kp[1], kp[2] };
        const float c2[3] = { frame_pose[0], frame_pose[1], frame_pose[2] };
        float diff[3] = { fabsf(c1[0] - c2[0]), fabsf(c1[1] - c2[1]), fabsf(c1[2] - c2[2]) };
        mat33f_vec3(kinv, diff, diff);

        if (flags[i] & (SVO_IGNORE_COMPLETELY | SVO_IGNORE_DURING_REFINEMENT)) {
            outlier_count[i]++;
            continue;
        }
        if (diff[0] < 0.1 && diff[1] < 0.1) continue;

        float p1[3] = { ref2d[i].x - cx, ref2d[i].y - cy, fx };
        mat33f_vec3(krot, p1, p1);
        float p2[3] = { kps2d[i].x - cx, kps2d[i].y - cy, fx };
        mat33f_vec3(frot, p2, p2);

        const float A[6] = { p1[0], -p2[0], p1[1], -p2[1], p1[2], -p2[2] };
        const float yv[3] = { c2[0] - c1[0], c2[1] - c1[1], c2[2] - c1[2] };
        float l[2];
        svo_o_solve_svd(A, 3, 2, yv, l);

        const float deviation = (float)(0.5 / (double)sqrtf(diff[0] * diff[0] + diff[1] * diff[1]));
        const float Rm = deviation * deviation;

        /* inv_rotation_kf*l(0)*(p1-c1): (Matx33f*float) then *Vec3f */
        float M[9];
        for (int k = 0; k < 9; k++) M[k] = kinv[k] * l[0];
        float pc[3] = { p1[0] - c1[0], p1[1] - c1[1], p1[2] - c1[2] };
        float new_p[3];
        mat33f_vec3(M, pc, new_p);
        float _z = new_p[2];
        svo_o_kf1_update(&kf_inv_depth[i], &kf_variance[i], 0.0001f, Rm, 1 / _z);
        _z = (float)(1.0 / (double)kf_inv_depth[i]);

        const float _x = (ref2d[i].x - cx) / fx * _z;
        const float _y = (ref2d[i].y - cy) / fy * _z;
        float cp[3] = { _x, _y, _z };
        mat33f_vec3(krot, cp, cp);
        kps3d[i].x = c1[0] + cp[0];
        kps3d[i].y = c1[1] + cp[1];
        kps3d[i].z = c1[2] + cp[2];
    }
}
