/* oracle_internal.h — CPU ORACLE internals shared by hot_path.c and tracker.c.
 * TEST INFRASTRUCTURE ONLY (see svo_oracle.h). */
#ifndef SVO_ORACLE_INTERNAL_H
#define SVO_ORACLE_INTERNAL_H

#include <stddef.h>
#include "svo_oracle.h"

/* What cv::buildOpticalFlowPyramid(img, pyr, Size(win,win), maxLevel,
 * withDerivatives=true, BORDER_REFLECT_101, BORDER_CONSTANT) stores per level:
 * the image padded by `win` (reflect-101) and the interleaved Scharr
 * derivatives padded by `win` zeros. */
typedef struct svo_oi_lkpyr {
    int n_levels, win;
    int w[SVO_LK_LEVELS], h[SVO_LK_LEVELS], pstride[SVO_LK_LEVELS];
    uint8_t *img[SVO_LK_LEVELS];
    int16_t *deriv[SVO_LK_LEVELS];
} svo_oi_lkpyr;

void svo_oi_lkpyr_build(svo_oi_lkpyr *p, const svo_image *levels, int n_levels, int win);
void svo_oi_lkpyr_free(svo_oi_lkpyr *p);
void svo_oi_klt_track(const svo_oi_lkpyr *P, const svo_oi_lkpyr *N, const svo_kp2d *prev_pts,
                      svo_kp2d *next_pts, int npts, uint8_t *status, float *err);

#endif
