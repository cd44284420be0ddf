/*
 * tracker.c — CPU ORACLE, part 3: restatement of the orchestrator
 * StereoSlam::new_image (src/lib/stereo_slam.cpp:123-271) with keyframe
 * creation (src/lib/keyframe_manager.cpp, depth_calculator.cpp,
 * corner_detector.cpp) and the 12-state pose Kalman filter (:296-359).
 * TEST INFRASTRUCTURE ONLY (see svo_oracle.h).
 *
 * Deliberate deviations from the reference, all outside the arithmetic:
 *  - keyframe counters are per instance (the reference uses process-global
 *    statics: keyframe_manager.cpp:8, depth_calculator.cpp:135);
 *  - debug colours come from a fixed LCG instead of rand() (depth_calculator.cpp:258);
 *  - no stdout tracing.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "svo_oracle.h"
#include "oracle_internal.h"

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* ---------------------------------------------------------------- keypoints */
typedef struct kps_t {
    int n, cap;
    svo_kp2d *kps2d;
    svo_kp3d *kps3d;
    svo_kp_info *info;
} kps_t;

static void kps_reserve(kps_t *k, int cap)
{
    if (cap <= k->cap) return;
    int nc = k->cap ? k->cap : 64;
    while (nc < cap) nc *= 2;
    k->kps2d = (svo_kp2d *)realloc(k->kps2d, sizeof(svo_kp2d) * (size_t)nc);
    k->kps3d = (svo_kp3d *)realloc(k->kps3d, sizeof(svo_kp3d) * (size_t)nc);
    k->info = (svo_kp_info *)realloc(k->info, sizeof(svo_kp_info) * (size_t)nc);
    k->cap = nc;
}

static void kps_copy(kps_t *dst, const kps_t *src)
{
    kps_reserve(dst, src->n);
    dst->n = src->n;
    memcpy(dst->kps2d, src->kps2d, sizeof(svo_kp2d) * (size_t)src->n);
    memcpy(dst->kps3d, src->kps3d, sizeof(svo_kp3d) * (size_t)src->n);
    memcpy(dst->info, src->info, sizeof(svo_kp_info) * (size_t)src->n);
}

static void kps_free(kps_t *k)
{
    free(k->kps2d); free(k->kps3d); free(k->info);
    memset(k, 0, sizeof(*k));
}

static void kps_erase(kps_t *k, int i)
{
    memmove(k->kps2d + i, k->kps2d + i + 1, sizeof(svo_kp2d) * (size_t)(k->n - i - 1));
    memmove(k->kps3d + i, k->kps3d + i + 1, sizeof(svo_kp3d) * (size_t)(k->n - i - 1));
    memmove(k->info + i, k->info + i + 1, sizeof(svo_kp_info) * (size_t)(k->n - i - 1));
    k->n--;
}

static uint32_t info_flags(const svo_kp_info *f)
{
    return (f->ignore_during_refinement ? SVO_IGNORE_DURING_REFINEMENT : 0) |
           (f->ignore_completely ? SVO_IGNORE_COMPLETELY : 0) |
           (f->ignore_temporary ? SVO_IGNORE_TEMPORARY : 0);
}

/* ------------------------------------------------------------------ images */
typedef struct images_t {
    int refs;
    int n_left;
    svo_image left[SVO_MAX_PYRAMID_LEVELS]; /* halfSample pyramid          */
    uint8_t *left_buf[SVO_MAX_PYRAMID_LEVELS];
    svo_image right0;
    uint8_t *right_buf;
    int n_lk;
    svo_image lk[SVO_LK_LEVELS];            /* Gaussian pyramid (unpadded) */
    uint8_t *lk_buf[SVO_LK_LEVELS];
    svo_oi_lkpyr lkpyr;                     /* padded + derivatives        */
} images_t;

static void images_release(images_t *im)
{
    if (!im) return;
    if (--im->refs > 0) return;
    for (int l = 0; l < SVO_MAX_PYRAMID_LEVELS; l++) free(im->left_buf[l]);
    for (int l = 0; l < SVO_LK_LEVELS; l++) free(im->lk_buf[l]);
    free(im->right_buf);
    svo_oi_lkpyr_free(&im->lkpyr);
    free(im);
}

typedef struct frame_t {
    int id;
    float pose[6];
    kps_t kps;
    double time_stamp;
    images_t *im;
} frame_t;

typedef struct keyframe_t {
    int id;
    float pose[6];
    kps_t kps;
    images_t *im;
} keyframe_t;

/* ------------------------------------------------------ 12-state Kalman filter */
#define KF_N 12
typedef struct kf12_t {
    float statePre[KF_N], statePost[KF_N];
    float A[KF_N * KF_N], Hm[KF_N * KF_N], Q[KF_N * KF_N], R[KF_N * KF_N];
    float errorCovPre[KF_N * KF_N], errorCovPost[KF_N * KF_N], gain[KF_N * KF_N];
} kf12_t;

static void set_identity(float *m, int n, float v)
{
    memset(m, 0, sizeof(float) * (size_t)n * n);
    for (int i = 0; i < n; i++) m[i * n + i] = v;
}

/* d = alpha * a * op(b) + beta * c, double accumulation, float store (cv::gemm) */
static void gemm_f(const float *a, const float *b, int bt, double alpha, const float *c,
                   double beta, float *d, int m, int k, int n)
{
    float tmp[KF_N * KF_N];
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int p = 0; p < k; p++)
                s += (double)a[i * k + p] * (double)(bt ? b[j * k + p] : b[p * n + j]);
            s *= alpha;
            if (c) s += (double)c[i * n + j] * beta;
            tmp[i * n + j] = (float)s;
        }
    memcpy(d, tmp, sizeof(float) * (size_t)m * n);
}

/* cv::solve(A, B, X, DECOMP_SVD) with nb right-hand sides (square A) */
static void solve_svd_multi(const float *A, int n, const float *B, int nb, float *X)
{
    float At[16 * 16], Vt[16 * 16], W[16];
    const float eps = (float)(DBL_EPSILON * 2);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) At[i * n + j] = A[j * n + i];
    svo_o_jacobi_svd(At, n, W, Vt, n, n, n, n);
    for (int i = 0; i < n * nb; i++) X[i] = 0;
    double threshold = 0;
    for (int i = 0; i < n; i++) threshold += W[i];
    threshold *= eps;
    for (int i = 0; i < n; i++) {
        double wi = W[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        double buffer[16];
        for (int j = 0; j < nb; j++) buffer[j] = 0;
        for (int r = 0; r < n; r++) {
            const float s = At[i * n + r];
            for (int j = 0; j < nb; j++) buffer[j] = buffer[j] + (double)(s * B[r * nb + j]);
        }
        for (int j = 0; j < nb; j++) buffer[j] *= wi;
        for (int r = 0; r < n; r++) {
            const float s = Vt[i * n + r];
            for (int j = 0; j < nb; j++) X[r * nb + j] = (float)(X[r * nb + j] + s * buffer[j]);
        }
    }
}

/* StereoSlam ctor — src/lib/stereo_slam.cpp:29-41 */
static void kf12_init(kf12_t *kf)
{
    memset(kf, 0, sizeof(*kf));
    set_identity(kf->A, KF_N, 1.0f);
    set_identity(kf->Hm, KF_N, 1.0f);
    set_identity(kf->Q, KF_N, 100.0f);
    set_identity(kf->R, KF_N, 1.0f);
    set_identity(kf->errorCovPost, KF_N, 1.0f);
}

static void kf12_predict(kf12_t *kf)
{
    float temp1[KF_N * KF_N];
    gemm_f(kf->A, kf->statePost, 0, 1, NULL, 0, kf->statePre, KF_N, KF_N, 1);
    gemm_f(kf->A, kf->errorCovPost, 0, 1, NULL, 0, temp1, KF_N, KF_N, KF_N);
    gemm_f(temp1, kf->A, 1, 1, kf->Q, 1, kf->errorCovPre, KF_N, KF_N, KF_N);
    memcpy(kf->statePost, kf->statePre, sizeof(kf->statePre));
    memcpy(kf->errorCovPost, kf->errorCovPre, sizeof(kf->errorCovPre));
}

static void kf12_correct(kf12_t *kf, const float *z)
{
    float temp2[KF_N * KF_N], temp3[KF_N * KF_N], temp4[KF_N * KF_N], temp5[KF_N], hx[KF_N];
    gemm_f(kf->Hm, kf->errorCovPre, 0, 1, NULL, 0, temp2, KF_N, KF_N, KF_N);
    gemm_f(temp2, kf->Hm, 1, 1, kf->R, 1, temp3, KF_N, KF_N, KF_N);
    solve_svd_multi(temp3, KF_N, temp2, KF_N, temp4);
    for (int i = 0; i < KF_N; i++)
        for (int j = 0; j < KF_N; j++) kf->gain[i * KF_N + j] = temp4[j * KF_N + i];
    gemm_f(kf->Hm, kf->statePre, 0, 1, NULL, 0, hx, KF_N, KF_N, 1);
    for (int i = 0; i < KF_N; i++) temp5[i] = z[i] - hx[i];
    gemm_f(kf->gain, temp5, 0, 1, kf->statePre, 1, kf->statePost, KF_N, KF_N, 1);
    gemm_f(kf->gain, temp2, 0, -1, kf->errorCovPre, 1, kf->errorCovPost, KF_N, KF_N, KF_N);
}

/* ------------------------------------------------------------------- slam */
struct svo_o_slam {
    svo_camera_settings cam;
    keyframe_t *keyframes;
    int n_keyframes, cap_keyframes;
    frame_t *frame;
    float motion[6];
    kf12_t kf;
    uint32_t keyframe_counter; /* KeyFrameManager::keyframe_counter */
    uint32_t keyframe_count;   /* static in DepthCalculator::calculate_depth */
    uint32_t color_lcg;
    svo_pose *trajectory;
    int n_traj, cap_traj;
    svo_o_frame_stats stats;
};

svo_o_slam *svo_o_slam_create(const svo_camera_settings *cam)
{
    svo_o_slam *s = (svo_o_slam *)calloc(1, sizeof(*s));
    s->cam = *cam;
    kf12_init(&s->kf);
    s->color_lcg = 12345u;
    return s;
}

static void frame_free(frame_t *f)
{
    if (!f) return;
    kps_free(&f->kps);
    images_release(f->im);
    free(f);
}

void svo_o_slam_destroy(svo_o_slam *s)
{
    if (!s) return;
    frame_free(s->frame);
    for (int i = 0; i < s->n_keyframes; i++) {
        kps_free(&s->keyframes[i].kps);
        images_release(s->keyframes[i].im);
    }
    free(s->keyframes);
    free(s->trajectory);
    free(s);
}

/* StereoSlam::update_pose — src/lib/stereo_slam.cpp:296-359 */
void svo_o_slam_update_pose(svo_o_slam *s, const float pose[6], const float speed[6],
                            const float pose_var[6], const float speed_var[6], double dt,
                            float filtered[6])
{
    kf12_t *kf = &s->kf;
    for (int i = 0; i < 6; i++) kf->A[i * KF_N + 6 + i] = (float)dt;
    kf12_predict(kf);
    for (int i = 0; i < 6; i++) {
        kf->R[i * KF_N + i] = pose_var[i];
        kf->R[(6 + i) * KF_N + 6 + i] = speed_var[i];
    }
    float z[KF_N];
    for (int i = 0; i < 6; i++) { z[i] = pose[i]; z[6 + i] = speed[i]; }
    kf12_correct(kf, z);
    for (int i = 0; i < 6; i++) filtered[i] = kf->statePost[i];
}

/* ----------------------------------------------------- keyframe creation */
/* cv::FAST (TYPE_9_16, nonmaxSuppression) restated from OpenCV
 * modules/features2d/src/fast.cpp + fast_score.cpp: a pixel is a corner if 9
 * contiguous ring pixels are all darker than v-t or all brighter than v+t;
 * score = largest t for which it still is one; kept if its score is strictly
 * larger than the scores of its 8 neighbours. PARITY UNPINNED. */
static const int ring_dx[16] = { 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1 };
static const int ring_dy[16] = { 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3 };

static int fast_corner_score(const uint8_t *p, int stride, int threshold)
{
    int d[25];
    const int v = p[0];
    for (int k = 0; k < 25; k++) d[k] = v - p[ring_dy[k & 15] * stride + ring_dx[k & 15]];
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        a = a < d[k + 3] ? a : d[k + 3];
        if (a <= a0) continue;
        for (int q = 4; q <= 8; q++) a = a < d[k + q] ? a : d[k + q];
        int t = a < d[k] ? a : d[k];
        a0 = a0 > t ? a0 : t;
        t = a < d[k + 9] ? a : d[k + 9];
        a0 = a0 > t ? a0 : t;
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        for (int q = 3; q <= 5; q++) b = b > d[k + q] ? b : d[k + q];
        if (b >= b0) continue;
        for (int q = 6; q <= 8; q++) b = b > d[k + q] ? b : d[k + q];
        int t = b > d[k] ? b : d[k];
        b0 = b0 < t ? b0 : t;
        t = b > d[k + 9] ? b : d[k + 9];
        b0 = b0 < t ? b0 : t;
    }
    return -b0 - 1;
}

static int fast_is_corner(const uint8_t *p, int stride, int threshold)
{
    const int v = p[0], K = 8, N = 25;
    int count = 0;
    for (int k = 0; k < N; k++) {
        const int x = p[ring_dy[k & 15] * stride + ring_dx[k & 15]];
        if (x < v - threshold) { if (++count > K) return 1; } else count = 0;
    }
    count = 0;
    for (int k = 0; k < N; k++) {
        const int x = p[ring_dy[k & 15] * stride + ring_dx[k & 15]];
        if (x > v + threshold) { if (++count > K) return 1; } else count = 0;
    }
    return 0;
}

void svo_o_fast_score_nms(const uint8_t *img, int w, int h, int stride, int threshold,
                          uint8_t *score)
{
    uint8_t *raw = (uint8_t *)calloc((size_t)w * h, 1);
    memset(score, 0, (size_t)w * h);
    for (int i = 3; i < h - 3; i++)
        for (int j = 3; j < w - 3; j++) {
            const uint8_t *p = img + (size_t)i * stride + j;
            if (fast_is_corner(p, stride, threshold))
                raw[(size_t)i * w + j] = (uint8_t)fast_corner_score(p, stride, threshold);
        }
    for (int i = 3; i < h - 3; i++)
        for (int j = 3; j < w - 3; j++) {
            const int sc = raw[(size_t)i * w + j];
            if (!sc) continue; /* a detected corner always scores >= threshold - 1 > 0 */
            const uint8_t *r0 = raw + (size_t)(i - 1) * w + j, *r1 = raw + (size_t)i * w + j,
                          *r2 = raw + (size_t)(i + 1) * w + j;
            if (sc > r1[1] && sc > r1[-1] && sc > r0[-1] && sc > r0[0] && sc > r0[1] &&
                sc > r2[-1] && sc > r2[0] && sc > r2[1])
                score[(size_t)i * w + j] = (uint8_t)sc;
        }
    free(raw);
}

static int reflect101i(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}

/* cv::Sobel(image, edge, -1, 1, 0): 3x3 [-1 0 1; -2 0 2; -1 0 1], BORDER_REFLECT_101,
 * saturate_cast<uchar> (src/lib/corner_detector.cpp:24-25). */
void svo_o_sobel_x_u8(const uint8_t *img, int w, int h, int stride, uint8_t *dst)
{
    for (int y = 0; y < h; y++) {
        const uint8_t *r0 = img + (size_t)reflect101i(y - 1, h) * stride;
        const uint8_t *r1 = img + (size_t)y * stride;
        const uint8_t *r2 = img + (size_t)reflect101i(y + 1, h) * stride;
        for (int x = 0; x < w; x++) {
            const int xm = reflect101i(x - 1, w), xp = reflect101i(x + 1, w);
            const int v = (r0[xp] - r0[xm]) + 2 * (r1[xp] - r1[xm]) + (r2[xp] - r2[xm]);
            dst[(size_t)y * w + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
}

/* CornerDetector::detect_keypoints — src/lib/corner_detector.cpp:13-79.
 * FAST keypoints come in OpenCV's order (row-major), the first best wins. */
int svo_o_detect_keypoints(const uint8_t *img, int w, int h, int stride, int grid_w, int grid_h,
                           int level, svo_kp2d *kps, float *score, int32_t *type, int cap)
{
    (void)level;
    uint8_t *fast = (uint8_t *)malloc((size_t)w * h);
    uint8_t *edge = (uint8_t *)malloc((size_t)w * h);
    svo_o_fast_score_nms(img, w, h, stride, 6, fast);
    svo_o_sobel_x_u8(img, w, h, stride, edge);
    int n = 0;
    int top = 0, bottom = grid_h;
    while (1) {
        int left = 0, right = grid_w;
        while (1) {
            if (right > w) break;
            float best = -1, bx = 0, by = 0;
            int btype = SVO_KP_FAST;
            for (int y = top > 0 ? top : 0; y < bottom && y < h; y++)
                for (int x = left; x < right; x++) {
                    const uint8_t sc = fast[(size_t)y * w + x];
                    if (!sc) continue; /* not a keypoint (a kept corner has score >= threshold) */
                    if (best < sc) { best = sc; bx = (float)x; by = (float)y; btype = SVO_KP_FAST; }
                }
            if (best < 0) {
                for (int k = left; k < right; k++)
                    for (int l = top; l < bottom; l++) {
                        const uint8_t response = edge[(size_t)l * w + k];
                        if (best < response) { best = response; bx = (float)k; by = (float)l; btype = SVO_KP_EDGELET; }
                    }
            }
            if (n < cap) { kps[n].x = bx; kps[n].y = by; score[n] = best; type[n] = btype; }
            n++;
            left += grid_w; right += grid_w;
        }
        bottom += grid_h;
        if (bottom > h) break;
        top += grid_h;
    }
    free(fast); free(edge);
    return n;
}

/* find_bad_keypoints — src/lib/depth_calculator.cpp:67-86 */
static void find_bad_keypoints(frame_t *f)
{
    const int width = f->im->left[0].width, height = f->im->left[0].height;
    for (int i = 0; i < f->kps.n; i++) {
        const svo_kp2d kp = f->kps.kps2d[i];
        if (kp.x < 0 || kp.y < 0 || kp.x > width || kp.y > height ||
            f->kps.info[i].ignore_completely || f->kps.info[i].ignore_during_refinement) {
            kps_erase(&f->kps, i);
            i--;
        }
    }
}

/* DepthCalculator::calculate_depth — src/lib/depth_calculator.cpp:132-392 */
static void calculate_depth(svo_o_slam *s, frame_t *f)
{
    const svo_camera_settings *cam = &s->cam;
    const float fx = cam->fx, fy = cam->fy, cx = cam->cx, cy = cam->cy, baseline = cam->baseline;

    find_bad_keypoints(f);

    /* detect_keypoints_on_each_level — :11-35 */
    const int nlev = f->im->n_left / 2;
    svo_kp2d *pk[SVO_MAX_PYRAMID_LEVELS];
    float *ps[SVO_MAX_PYRAMID_LEVELS];
    int32_t *pt[SVO_MAX_PYRAMID_LEVELS];
    int pn[SVO_MAX_PYRAMID_LEVELS];
    int gw = cam->grid_width, gh = cam->grid_height;
    for (int l = 0; l < nlev; l++) {
        const svo_image *im = &f->im->left[l];
        const int cap = (im->width / (gw > 0 ? gw : 1) + 1) * (im->height / (gh > 0 ? gh : 1) + 1) + 1;
        pk[l] = (svo_kp2d *)malloc(sizeof(svo_kp2d) * (size_t)cap);
        ps[l] = (float *)malloc(sizeof(float) * (size_t)cap);
        pt[l] = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
        pn[l] = svo_o_detect_keypoints(im->data, im->width, im->height, im->stride, gw, gh, l,
                                       pk[l], ps[l], pt[l], cap);
        gw /= 2; gh /= 2;
    }
    /* select_best_keypoints — :37-65 (index j is used on every level as is) */
    const int nsel = nlev > 0 ? pn[0] : 0;
    svo_kp2d *sk = (svo_kp2d *)malloc(sizeof(svo_kp2d) * (size_t)(nsel + 1));
    float *ss = (float *)malloc(sizeof(float) * (size_t)(nsel + 1));
    int32_t *st = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nsel + 1));
    int32_t *sl = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nsel + 1));
    for (int j = 0; j < nsel; j++) { sk[j] = pk[0][j]; ss[j] = ps[0][j]; st[j] = pt[0][j]; sl[j] = 0; }
    for (int i = 1; i < nlev; i++)
        for (int j = 0; j < nsel; j++) {
            if (j >= pn[i]) continue; /* the reference would read out of bounds here */
            if (st[j] == SVO_KP_FAST && pt[i][j] == SVO_KP_EDGELET) continue;
            if (st[j] == pt[i][j] && ss[j] > ps[i][j]) continue;
            sk[j] = pk[i][j];
            sk[j].x *= (1 << i);
            sk[j].y *= (1 << i);
            ss[j] = ps[i][j]; st[j] = pt[i][j]; sl[j] = i;
        }

    const int old_count = f->kps.n;

    /* merge_keypoints — :88-130, called with grid_width/grid_height swapped (:179-180) */
    {
        const int m_grid_height = cam->grid_width, m_grid_width = cam->grid_height;
        const int image_width = f->im->left[0].width, image_height = f->im->left[0].height;
        for (int x = 0; x < image_width; x += m_grid_width) {
            const int left = x, right = left + m_grid_width;
            for (int y = 0; y < image_height; y += m_grid_height) {
                const int top = y, bottom = y + m_grid_height;
                int match = 0;
                for (int i = 0; i < f->kps.n; i++) {
                    const svo_kp2d kp = f->kps.kps2d[i];
                    if (kp.x > left && kp.x < right && kp.y > top && kp.y < bottom) { match = 1; break; }
                }
                if (!match)
                    for (int i = 0; i < nsel; i++) {
                        const svo_kp2d kp = sk[i];
                        if (kp.x > left && kp.x < right && kp.y > top && kp.y < bottom) {
                            kps_reserve(&f->kps, f->kps.n + 1);
                            const int q = f->kps.n++;
                            f->kps.kps2d[q] = kp;
                            memset(&f->kps.info[q], 0, sizeof(svo_kp_info));
                            f->kps.info[q].score = ss[i];
                            f->kps.info[q].type = st[i];
                            f->kps.info[q].level = sl[i];
                            f->kps.kps3d[q].x = f->kps.kps3d[q].y = f->kps.kps3d[q].z = 0;
                        }
                    }
            }
        }
    }

    float rot[9], inv_rot[9];
    svo_o_pose_matrices(f->pose, rot, inv_rot);

    const int n_new = f->kps.n - old_count;
    float *disp = (float *)malloc(sizeof(float) * (size_t)(n_new + 1));
    svo_o_ssd_disparity(&f->im->left[0], &f->im->right0, f->kps.kps2d + old_count, n_new,
                        cam->window_size_depth_calculator, cam->search_x, cam->search_y, 0, disp);
    for (int i = old_count; i < f->kps.n; i++) {
        const svo_kp2d kp = f->kps.kps2d[i];
        const float disparity = disp[i - old_count];
        const float _z = baseline / (0.5f > disparity ? 0.5f : disparity);
        const float _x = (kp.x - cx) / fx * _z;
        const float _y = (kp.y - cy) / fy * _z;
        const float loc[3] = { _x, _y, _z };
        float g[3];
        for (int r = 0; r < 3; r++) {
            float sum = 0;
            for (int k = 0; k < 3; k++) sum += rot[r * 3 + k] * loc[k];
            g[r] = sum;
        }
        f->kps.kps3d[i].x = g[0] + f->pose[0];
        f->kps.kps3d[i].y = g[1] + f->pose[1];
        f->kps.kps3d[i].z = g[2] + f->pose[2];

        s->color_lcg = s->color_lcg * 1664525u + 1013904223u;
        const uint32_t color = s->color_lcg >> 8;
        svo_kp_info *in = &f->kps.info[i];
        in->color[0] = (color >> 0) & 0xFF;
        in->color[1] = (color >> 8) & 0xFF;
        in->color[2] = (color >> 16) & 0xFF;
        in->keyframe_id = (int32_t)s->keyframe_count;
        in->keypoint_index = i;
        in->ignore_completely = 0;
        in->ignore_temporary = 1;
        in->ignore_during_refinement = 0;
        in->inlier_count = 0;
        in->outlier_count = 0;
        const float deviation = (float)(0.5 / (double)(baseline / fx));
        in->kf_variance = deviation * deviation;
        in->kf_inv_depth = 1 / _z;
    }
    free(disp);
    for (int l = 0; l < nlev; l++) { free(pk[l]); free(ps[l]); free(pt[l]); }
    free(sk); free(ss); free(st); free(sl);
    s->keyframe_count++;
}

/* KeyFrameManager::create_keyframe — src/lib/keyframe_manager.cpp:15-32 */
static keyframe_t *create_keyframe(svo_o_slam *s, frame_t *f)
{
    if (s->n_keyframes == s->cap_keyframes) {
        s->cap_keyframes = s->cap_keyframes ? s->cap_keyframes * 2 : 16;
        s->keyframes = (keyframe_t *)realloc(s->keyframes, sizeof(keyframe_t) * (size_t)s->cap_keyframes);
    }
    keyframe_t *kf = &s->keyframes[s->n_keyframes++];
    memset(kf, 0, sizeof(*kf));
    kf->id = (int)s->keyframe_counter++;
    calculate_depth(s, f);
    kps_copy(&kf->kps, &f->kps);
    memcpy(kf->pose, f->pose, sizeof(kf->pose));
    kf->im = f->im;
    f->im->refs++;
    return kf;
}

/* KeyFrameManager::keyframe_needed — src/lib/keyframe_manager.cpp:47-74 */
static int keyframe_needed(const svo_o_slam *s, const frame_t *f)
{
    const int image_width = f->im->left[0].width, image_height = f->im->left[0].height;
    int inside = 0;
    for (int i = 0; i < f->kps.n; i++) {
        const svo_kp2d kp = f->kps.kps2d[i];
        if (kp.x > 0 && kp.y > 0 && kp.x < image_width && kp.y < image_height &&
            !f->kps.info[i].ignore_completely)
            inside++;
    }
    const int max_keypoints = (image_width / s->cam.grid_width) * (image_height / s->cam.grid_height);
    return inside < 0.66 * max_keypoints;
}

static images_t *build_images(const svo_camera_settings *cam, const uint8_t *left,
                              const uint8_t *right, int w, int h)
{
    images_t *im = (images_t *)calloc(1, sizeof(*im));
    im->refs = 1;
    im->n_left = cam->max_pyramid_levels;
    for (int l = 0, lw = w, lh = h; l < im->n_left; l++, lw /= 2, lh /= 2) {
        im->left_buf[l] = (uint8_t *)malloc((size_t)(lw > 0 ? lw : 1) * (lh > 0 ? lh : 1));
        im->left[l].data = im->left_buf[l];
    }
    memcpy(im->left_buf[0], left, (size_t)w * h);
    svo_image l0 = { im->left_buf[0], w, h, w };
    svo_o_build_pyramid(&l0, im->n_left, im->left);
    im->right_buf = (uint8_t *)malloc((size_t)w * h);
    memcpy(im->right_buf, right, (size_t)w * h);
    im->right0.data = im->right_buf; im->right0.width = w; im->right0.height = h; im->right0.stride = w;
    for (int l = 1, lw = w, lh = h; l < SVO_LK_LEVELS; l++) {
        lw = (lw + 1) / 2; lh = (lh + 1) / 2;
        im->lk_buf[l] = (uint8_t *)malloc((size_t)lw * lh);
        im->lk[l].data = im->lk_buf[l];
    }
    im->n_lk = svo_o_build_lk_pyramid(&l0, SVO_LK_LEVELS, cam->window_size_opt_flow, im->lk);
    svo_oi_lkpyr_build(&im->lkpyr, im->lk, im->n_lk, cam->window_size_opt_flow);
    return im;
}

static void traj_push(svo_o_slam *s, const float pose[6])
{
    if (s->n_traj == s->cap_traj) {
        s->cap_traj = s->cap_traj ? s->cap_traj * 2 : 256;
        s->trajectory = (svo_pose *)realloc(s->trajectory, sizeof(svo_pose) * (size_t)s->cap_traj);
    }
    memcpy(&s->trajectory[s->n_traj++], pose, sizeof(svo_pose));
}

/* StereoSlam::new_image — src/lib/stereo_slam.cpp:123-271 */
int svo_o_slam_new_image(svo_o_slam *s, const uint8_t *left, const uint8_t *right, int width,
                         int height, float time_stamp)
{
    const svo_camera_settings *cam = &s->cam;
    svo_o_frame_stats *stx = &s->stats;
    memset(stx, 0, sizeof(*stx));
    const double t_begin = now_s();
    int made_keyframe = 0;

    frame_t *prev = s->frame;
    frame_t *f = (frame_t *)calloc(1, sizeof(*f));
    s->frame = f;
    f->time_stamp = time_stamp;

    double t0 = now_s();
    f->im = build_images(cam, left, right, width, height);
    stx->t_pyramid = now_s() - t0;

    if (!prev) {
        f->id = 0;
        memset(f->pose, 0, sizeof(f->pose));
        t0 = now_s();
        create_keyframe(s, f);
        stx->t_keyframe = now_s() - t0;
        made_keyframe = 1;
        for (int i = 0; i < f->kps.n; i++) f->kps.info[i].ignore_temporary = 0;
    } else {
        f->id = prev->id + 1;
        for (int i = 0; i < 6; i++) f->pose[i] = s->kf.statePre[i];

        /* remove_outliers — :43-56 */
        {
            int m = 0;
            for (int i = 0; i < prev->kps.n; i++) {
                if (prev->kps.info[i].ignore_completely) continue;
                prev->kps.kps2d[m] = prev->kps.kps2d[i];
                prev->kps.kps3d[m] = prev->kps.kps3d[i];
                prev->kps.info[m] = prev->kps.info[i];
                m++;
            }
            prev->kps.n = m;
        }
        const int n = prev->kps.n;
        stx->n_tracked = n;
        uint32_t *flags = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n + 1));
        int n_active = 0;
        for (int i = 0; i < n; i++) {
            flags[i] = info_flags(&prev->kps.info[i]);
            if (!(flags[i] & SVO_IGNORE_TEMPORARY)) n_active++;
        }
        stx->n_active = n_active;

        /* estimate_pose — :58-90 */
        t0 = now_s();
        float estimated[6];
        svo_o_sparse_align(prev->im->left, f->im->left, prev->kps.kps2d, prev->kps.kps3d, flags, n,
                           cam, f->pose, estimated, stx->sia_trace);
        stx->t_sia = now_s() - t0;
        for (int l = 0; l < cam->max_pyramid_levels; l++) {
            stx->sia_gradient_calls += stx->sia_trace[l].n_gradient;
            stx->sia_cost_calls += stx->sia_trace[l].n_cost;
        }
        memcpy(stx->pose_sia, estimated, sizeof(estimated));
        memcpy(f->pose, estimated, sizeof(estimated));
        kps_copy(&f->kps, &prev->kps);
        svo_o_project_keypoints(estimated, prev->kps.kps3d, n, cam, f->kps.kps2d);

        /* PoseRefiner::refine_pose — src/lib/pose_refinement.cpp:62-177.
         * Keypoints are grouped by origin keyframe (std::map: ascending id) and
         * each group is tracked from that keyframe's LK pyramid. */
        t0 = now_s();
        svo_kp2d *tracked = (svo_kp2d *)malloc(sizeof(svo_kp2d) * (size_t)(n + 1));
        svo_kp2d *refpts = (svo_kp2d *)malloc(sizeof(svo_kp2d) * (size_t)(n + 1));
        svo_kp2d *gref = (svo_kp2d *)malloc(sizeof(svo_kp2d) * (size_t)(n + 1));
        svo_kp2d *gcur = (svo_kp2d *)malloc(sizeof(svo_kp2d) * (size_t)(n + 1));
        int *gidx = (int *)malloc(sizeof(int) * (size_t)(n + 1));
        float *err = (float *)malloc(sizeof(float) * (size_t)(n + 1));
        float *gerr = (float *)malloc(sizeof(float) * (size_t)(n + 1));
        uint8_t *gstatus = (uint8_t *)malloc((size_t)(n + 1));
        for (int i = 0; i < n; i++) {
            const svo_kp_info *in = &f->kps.info[i];
            refpts[i] = s->keyframes[in->keyframe_id].kps.kps2d[in->keypoint_index];
        }
        for (int kfid = 0; kfid < s->n_keyframes; kfid++) {
            int m = 0;
            for (int i = 0; i < n; i++)
                if (f->kps.info[i].keyframe_id == kfid) {
                    gref[m] = refpts[i]; gcur[m] = f->kps.kps2d[i]; gidx[m] = i; m++;
                }
            if (!m) continue;
            svo_oi_klt_track(&s->keyframes[kfid].im->lkpyr, &f->im->lkpyr, gref, gcur, m, gstatus, gerr);
            for (int q = 0; q < m; q++) { tracked[gidx[q]] = gcur[q]; err[gidx[q]] = gerr[q]; }
        }
        stx->t_klt = now_s() - t0;
        t0 = now_s();
        for (int i = 0; i < n; i++) flags[i] = info_flags(&f->kps.info[i]);
        svo_o_refine_merge(f->kps.kps2d, flags, tracked, err, n);
        float refined[6];
        svo_o_reproj_gn(f->kps.kps2d, f->kps.kps3d, flags, n, cam, f->pose, refined, &stx->reproj_trace);
        memcpy(f->pose, refined, sizeof(refined));
        memcpy(stx->pose_refined, refined, sizeof(refined));
        stx->t_reproj = now_s() - t0;

        /* DepthFilter::update_depth — src/lib/depth_filter.cpp:40-50 */
        t0 = now_s();
        float *disp = (float *)malloc(sizeof(float) * (size_t)(n + 1));
        svo_o_ssd_disparity(&f->im->left[0], &f->im->right0, f->kps.kps2d, n,
                            cam->window_size_depth_calculator, cam->search_x, cam->search_y, 1, disp);
        stx->t_disparity = now_s() - t0;
        t0 = now_s();
        svo_kp3d *ref3d = (svo_kp3d *)malloc(sizeof(svo_kp3d) * (size_t)(n + 1));
        float *kfpose = (float *)malloc(sizeof(float) * 6 * (size_t)(n + 1));
        int32_t *outl = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
        int32_t *inl = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
        float *kx = (float *)malloc(sizeof(float) * (size_t)(n + 1));
        float *kP = (float *)malloc(sizeof(float) * (size_t)(n + 1));
        for (int i = 0; i < n; i++) {
            const svo_kp_info *in = &f->kps.info[i];
            const keyframe_t *kf = &s->keyframes[in->keyframe_id];
            ref3d[i] = kf->kps.kps3d[in->keypoint_index];
            memcpy(kfpose + (size_t)i * 6, kf->pose, sizeof(float) * 6);
            outl[i] = in->outlier_count; inl[i] = in->inlier_count;
            kx[i] = in->kf_inv_depth; kP[i] = in->kf_variance;
        }
        svo_o_outlier_check(f->kps.kps2d, disp, n, cam, f->pose, ref3d, kfpose, outl, inl);
        svo_kp3d *updated = (svo_kp3d *)malloc(sizeof(svo_kp3d) * (size_t)(n + 1));
        memcpy(updated, f->kps.kps3d, sizeof(svo_kp3d) * (size_t)n);
        svo_o_update_kps3d(f->kps.kps2d, updated, flags, n, cam, f->pose, refpts, kfpose, outl, kx, kP);

        /* write back — src/lib/stereo_slam.cpp:205-226 */
        for (int i = 0; i < n; i++) {
            svo_kp_info *in = &f->kps.info[i];
            in->ignore_during_refinement = (flags[i] & SVO_IGNORE_DURING_REFINEMENT) != 0;
            in->ignore_completely = (flags[i] & SVO_IGNORE_COMPLETELY) != 0;
            in->outlier_count = outl[i]; in->inlier_count = inl[i];
            in->kf_inv_depth = kx[i]; in->kf_variance = kP[i];
            keyframe_t *kf = &s->keyframes[in->keyframe_id];
            if (in->outlier_count > in->inlier_count) in->ignore_completely = 1;
            if (in->inlier_count > in->outlier_count) in->ignore_temporary = 0;
            kf->kps.kps3d[in->keypoint_index] = updated[i];
            f->kps.kps3d[i] = updated[i];
            svo_kp_info *ki = &kf->kps.info[in->keypoint_index];
            ki->ignore_temporary = in->ignore_temporary;
            ki->ignore_completely = in->ignore_completely;
            ki->inlier_count = in->inlier_count;
            ki->outlier_count = in->outlier_count;
        }
        stx->t_filter = now_s() - t0;
        svo_o_project_keypoints(f->pose, f->kps.kps3d, n, cam, f->kps.kps2d);

        if (keyframe_needed(s, f)) {
            t0 = now_s();
            create_keyframe(s, f);
            stx->t_keyframe = now_s() - t0;
            made_keyframe = 1;
            int cnt = 0;
            for (int i = 0; i < f->kps.n; i++)
                if (!f->kps.info[i].ignore_temporary) cnt++;
            if ((size_t)cnt < (size_t)f->kps.n / 4)
                for (int i = 0; i < f->kps.n; i++) f->kps.info[i].ignore_temporary = 0;
        }
        free(flags); free(tracked); free(refpts); free(gref); free(gcur); free(gidx);
        free(err); free(gerr); free(gstatus); free(disp); free(ref3d); free(kfpose);
        free(outl); free(inl); free(kx); free(kP); free(updated);
    }

    if (prev) {
        const double dt = f->time_stamp - prev->time_stamp;
        const double inv = 1. / dt;
        for (int i = 0; i < 6; i++) s->motion[i] = (float)((f->pose[i] - prev->pose[i]) * inv);
        const float pv[6] = { 0.1f, 0.1f, 0.1f, 0.1f, 0.1f, 0.1f };
        const float mv[6] = { 1, 1, 1, 1, 1, 1 };
        float filtered[6];
        svo_o_slam_update_pose(s, f->pose, s->motion, pv, mv, 0.0, filtered);
        memcpy(f->pose, filtered, sizeof(filtered));
        frame_free(prev);
    }
    traj_push(s, f->pose);
    stx->t_total = now_s() - t_begin;
    return made_keyframe;
}

void svo_o_slam_get_pose(const svo_o_slam *s, float pose[6])
{
    if (s->frame) memcpy(pose, s->frame->pose, sizeof(float) * 6);
    else memset(pose, 0, sizeof(float) * 6);
}

int svo_o_slam_num_keypoints(const svo_o_slam *s) { return s->frame ? s->frame->kps.n : 0; }
int svo_o_slam_num_keyframes(const svo_o_slam *s) { return s->n_keyframes; }

static int copy_kps(const kps_t *k, svo_kp2d *kps2d, svo_kp3d *kps3d, svo_kp_info *info, int cap)
{
    const int n = k->n < cap ? k->n : cap;
    if (kps2d) memcpy(kps2d, k->kps2d, sizeof(svo_kp2d) * (size_t)n);
    if (kps3d) memcpy(kps3d, k->kps3d, sizeof(svo_kp3d) * (size_t)n);
    if (info) memcpy(info, k->info, sizeof(svo_kp_info) * (size_t)n);
    return k->n;
}

int svo_o_slam_get_keypoints(const svo_o_slam *s, svo_kp2d *kps2d, svo_kp3d *kps3d,
                             svo_kp_info *info, int cap)
{
    if (!s->frame) return 0;
    return copy_kps(&s->frame->kps, kps2d, kps3d, info, cap);
}

int svo_o_slam_get_keyframe_keypoints(const svo_o_slam *s, int id, svo_kp2d *kps2d,
                                      svo_kp3d *kps3d, svo_kp_info *info, float pose[6], int cap)
{
    if (id < 0 || id >= s->n_keyframes) return -1;
    if (pose) memcpy(pose, s->keyframes[id].pose, sizeof(float) * 6);
    return copy_kps(&s->keyframes[id].kps, kps2d, kps3d, info, cap);
}

void svo_o_slam_get_stats(const svo_o_slam *s, svo_o_frame_stats *out) { *out = s->stats; }
