// depth.hip — rows C1, C2, D1, D3 of SURVEY §8a: the stereo depth filter
// DepthFilter::update_depth (src/lib/depth_filter.cpp:40-50).
//
//  * ssd_disparity_kernel (C1): calculate_disparities (depth_filter.cpp:259-327)
//    and the identical loop of DepthCalculator::calculate_depth
//    (depth_calculator.cpp:200-240). One workgroup per keypoint; template and
//    search region live in LDS; exact int32 SSD (cv::matchTemplate TM_SQDIFF),
//    first-minimum argmin in row-major order (cv::minMaxLoc) and the
//    tie-averaged column of :313-323.
//  * filter_update_kernel (C2, D1, D3): outlier_check (:52-128),
//    update_kps3d (:130-257), the flag/write-back loop of
//    StereoSlam::new_image (src/lib/stereo_slam.cpp:205-229) and the counter
//    of KeyFrameManager::keyframe_needed (keyframe_manager.cpp:47-74).
#include "svo_kernels.hpp"

namespace svo {

constexpr int SSD_MAX_WIN = 36;
constexpr int SSD_MAX_ROI_W = 36 + 64 + 4;   // win + search_x, padded
constexpr int SSD_MAX_ROI_H = 36 + 2 * 8;    // win + 2*search_y
constexpr int SSD_MAX_MATCH = 65 * 17;

__global__ __launch_bounds__(256) void ssd_disparity_kernel(const SsdArgs* __restrict__ args) {
    const SsdArgs& a = args[blockIdx.y];
    if (a.enable && !*a.enable) return;
    const int n = *a.n_ptr;
    const int kp = a.first + (a.first_ptr ? *a.first_ptr : 0) + (int)blockIdx.x;
    if (kp >= n) return;
    const int tid = threadIdx.x;

    __shared__ uint8_t s_t[SSD_MAX_WIN * SSD_MAX_WIN];
    __shared__ uint8_t s_r[SSD_MAX_ROI_H * SSD_MAX_ROI_W];
    __shared__ int s_m[SSD_MAX_MATCH];
    __shared__ unsigned long long s_key[4];
    __shared__ int s_sum[4], s_cnt[4];

    const int window_before = a.win / 2, window_after = (a.win + 1) / 2;
    const int cols = a.left.w, rows = a.left.h;
    const svo_kp2d p = a.kps2d[kp];
    const int x = (int)p.x, y = (int)p.y;
    const int x11 = max(0, x - window_before);
    const int x12 = min(cols - 1, x + window_after);
    const int y11 = max(0, y - window_before);
    const int y12 = min(rows, y + window_after);
    const int x21 = x11;
    const int x22 = min(cols - 1, x + window_after + a.search_x);
    const int y21 = max(0, y - window_before - a.search_y);
    const int y22 = min(rows - 1, y + window_after + a.search_y);
    const int tw = x12 - x11, th = y12 - y11, rw = x22 - x21, rh = y22 - y21;
    const int mw = rw - tw + 1, mh = rh - th + 1;
    bool skip = false;
    if (a.clamp_half && (x12 <= 0 || y12 <= 0 || x11 >= cols - 1 || y11 >= rows - 1)) skip = true;
    if (a.clamp_half && (x22 <= 0 || y22 <= 0 || x21 >= cols - 1 || y21 >= rows - 1)) skip = true;
    if (tw <= 0 || th <= 0 || mw <= 0 || mh <= 0) skip = true;
    if (tw > SSD_MAX_WIN || th > SSD_MAX_WIN || rw > SSD_MAX_ROI_W || rh > SSD_MAX_ROI_H ||
        mw * mh > SSD_MAX_MATCH)
        skip = true;  // host validates window sizes; never taken with valid settings
    if (skip) {
        if (tid == 0) a.disparity[kp] = -1.0f;
        return;
    }

    for (int i = tid; i < tw * th; i += 256) {
        const int r = i / tw, c = i % tw;
        s_t[r * SSD_MAX_WIN + c] = a.left.data[(size_t)(y11 + r) * a.left.stride + x11 + c];
    }
    for (int i = tid; i < rw * rh; i += 256) {
        const int r = i / rw, c = i % rw;
        s_r[r * SSD_MAX_ROI_W + c] = a.right.data[(size_t)(y21 + r) * a.right.stride + x21 + c];
    }
    __syncthreads();

    const int nm = mw * mh;
    unsigned long long best = ~0ull;
    for (int o = tid; o < nm; o += 256) {
        const int k = o / mw, j = o % mw;
        int acc = 0;
        for (int r = 0; r < th; r++) {
            const uint8_t* t = &s_t[r * SSD_MAX_WIN];
            const uint8_t* q = &s_r[(k + r) * SSD_MAX_ROI_W + j];
            for (int c = 0; c < tw; c++) {
                const int d = (int)q[c] - (int)t[c];
                acc += d * d;
            }
        }
        s_m[o] = acc;
        const unsigned long long key = ((unsigned long long)(unsigned)acc << 32) | (unsigned)o;
        best = key < best ? key : best;
    }
    // first minimum in row-major order == smallest (value, index) key
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned long long other = __shfl_xor(best, o, 64);
        best = other < best ? other : best;
    }
    if ((tid & 63) == 0) s_key[tid >> 6] = best;
    __syncthreads();
    best = s_key[0];
    for (int w = 1; w < 4; w++) best = s_key[w] < best ? s_key[w] : best;
    const int min_int = (int)(best >> 32);
    const int min_o = (int)(best & 0xffffffffu);
    const int minx = min_o % mw, miny = min_o / mw;
    const float minVal = (float)min_int;

    int sumj = 0, cnt = 0;
    for (int o = tid; o < nm; o += 256) {
        const int k = o / mw, j = o % mw;
        if (j >= minx && k >= miny && (double)(float)s_m[o] <= (double)minVal) { sumj += j; cnt++; }
    }
    sumj = wave_sum_i(sumj);
    cnt = wave_sum_i(cnt);
    if ((tid & 63) == 0) { s_sum[tid >> 6] = sumj; s_cnt[tid >> 6] = cnt; }
    __syncthreads();
    if (tid == 0) {
        const int ts = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
        const int tc = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        float minPos = (float)ts;      // sum of small ints: exact in float in any order
        minPos = minPos / tc;
        a.disparity[kp] = a.clamp_half ? fmaxf(0.5f, minPos) : minPos;
    }
}

void launch_ssd(const SsdArgs* d_args, int batch, int max_n, hipStream_t stream) {
    if (max_n <= 0) return;
    hipLaunchKernelGGL(ssd_disparity_kernel, dim3(max_n, batch), dim3(256), 0, stream, d_args);
}

// -------------------------------------------------------------------------
__global__ __launch_bounds__(256) void filter_update_kernel(const FilterArgs* __restrict__ args) {
    const FilterArgs& a = args[blockIdx.x];
    const int n = *a.n_ptr;
    const int tid = threadIdx.x;
    __shared__ PoseMats s_frame;
    __shared__ int s_inside[4];
    if (tid == 0) {
        float pose[6];
        for (int i = 0; i < 6; i++) pose[i] = a.frame_pose[i];
        pose_mats(pose, s_frame);
    }
    __syncthreads();
    const float fx = a.cam.fx, fy = a.cam.fy, cx = a.cam.cx, cy = a.cam.cy, baseline = a.cam.baseline;
    const CamD camd = make_camd(fx, fy, cx, cy, a.cam);
    int inside = 0;

    for (int i = tid; i < n; i += 256) {
        // references: explicit arrays (stage API) or the keyframe table (tracker)
        float kfp[6];
        svo_kp3d r3;
        svo_kp2d r2;
        KfDev* kf = nullptr;
        int kidx = 0;
        if (a.kfs) {
            kf = &a.kfs[a.kf_id[i]];
            kidx = a.kp_index[i];
            for (int q = 0; q < 6; q++) kfp[q] = kf->pose[q];
            r3 = kf->kps3d[kidx];
            r2 = kf->kps2d[kidx];
        } else {
            for (int q = 0; q < 6; q++) kfp[q] = a.kf_pose[(size_t)i * 6 + q];
            r3 = a.ref3d ? a.ref3d[i] : svo_kp3d{0, 0, 0};
            r2 = a.ref2d ? a.ref2d[i] : svo_kp2d{0, 0};
        }
        PoseMats km;
        pose_mats(kfp, km);
        const svo_kp2d kp2 = a.kps2d[i];
        uint32_t flags = a.flags[i];
        int outl = a.outlier_count[i], inl = a.inlier_count[i];

        if (a.do_outlier_check) {   // depth_filter.cpp:52-128
            const float d = a.disparity[i];
            const float _z = baseline / fmaxf(d, 0.5f);
            const float _x = (kp2.x - cx) / fx * _z;
            const float _y = (kp2.y - cy) / fy * _z;
            float pw[3] = {_x, _y, _z};
            mat33f_vec(s_frame.R, pw, pw);
            pw[0] += s_frame.t[0]; pw[1] += s_frame.t[1]; pw[2] += s_frame.t[2];
            float av[3] = {pw[0] - kfp[0], pw[1] - kfp[1], pw[2] - kfp[2]};
            mat33f_vec(km.Ri, av, av);
            float rv[3] = {r3.x - kfp[0], r3.y - kfp[1], r3.z - kfp[2]};
            mat33f_vec(km.Ri, rv, rv);
            const float disp_ref = baseline / rv[2];
            const float disp = baseline / av[2];
            const float pixel_distance = disp - disp_ref;
            if (fabsf(pixel_distance) > 5 * 0.5f) outl++;
            else inl++;
        }

        svo_kp3d p3 = a.kps3d[i];
        if (a.do_update) {          // depth_filter.cpp:130-257
            const float c1[3] = {kfp[0], kfp[1], kfp[2]};
            const float c2[3] = {s_frame.t[0], s_frame.t[1], s_frame.t[2]};
            float diff[3] = {fabsf(c1[0] - c2[0]), fabsf(c1[1] - c2[1]), fabsf(c1[2] - c2[2])};
            mat33f_vec(km.Ri, diff, diff);
            if (flags & (SVO_IGNORE_COMPLETELY | SVO_IGNORE_DURING_REFINEMENT)) {
                outl++;
            } else if (!((double)diff[0] < 0.1 && (double)diff[1] < 0.1)) {
                float p1[3] = {r2.x - cx, r2.y - cy, fx};
                mat33f_vec(km.R, p1, p1);
                float p2[3] = {kp2.x - cx, kp2.y - cy, fx};
                mat33f_vec(s_frame.R, p2, p2);
                const float A[6] = {p1[0], -p2[0], p1[1], -p2[1], p1[2], -p2[2]};
                const float yv[3] = {c2[0] - c1[0], c2[1] - c1[1], c2[2] - c1[2]};
                float l[2];
                solve_svd_3x2(A, yv, l);
                const float deviation =
                    (float)(0.5 / (double)sqrtf(diff[0] * diff[0] + diff[1] * diff[1]));
                const float Rm = deviation * deviation;
                float Mx[9];
#pragma unroll
                for (int k = 0; k < 9; k++) Mx[k] = km.Ri[k] * l[0];
                const float pc[3] = {p1[0] - c1[0], p1[1] - c1[1], p1[2] - c1[2]};
                float new_p[3];
                mat33f_vec(Mx, pc, new_p);
                float _z = new_p[2];
                float kx = a.kf_inv_depth[i], kP = a.kf_variance[i];
                kf1_update(kx, kP, 0.0001f, Rm, 1 / _z);
                a.kf_inv_depth[i] = kx;
                a.kf_variance[i] = kP;
                _z = (float)(1.0 / (double)kx);
                const float _x = (r2.x - cx) / fx * _z;
                const float _y = (r2.y - cy) / fy * _z;
                float cp[3] = {_x, _y, _z};
                mat33f_vec(km.R, cp, cp);
                p3.x = c1[0] + cp[0];
                p3.y = c1[1] + cp[1];
                p3.z = c1[2] + cp[2];
            }
        }

        if (a.do_flags) {           // stereo_slam.cpp:205-226
            if (outl > inl) flags |= SVO_IGNORE_COMPLETELY;
            if (inl > outl) flags &= ~(uint32_t)SVO_IGNORE_TEMPORARY;
            if (kf) {
                kf->kps3d[kidx] = p3;
                const uint32_t keep = kf->flags[kidx] & SVO_IGNORE_DURING_REFINEMENT;
                kf->flags[kidx] = keep | (flags & (SVO_IGNORE_TEMPORARY | SVO_IGNORE_COMPLETELY));
                kf->inlier_count[kidx] = inl;
                kf->outlier_count[kidx] = outl;
            }
        }
        a.kps3d[i] = p3;
        if (a.do_flags) a.flags[i] = flags;
        a.outlier_count[i] = outl;
        a.inlier_count[i] = inl;

        if (a.do_reproject) {       // stereo_slam.cpp:228-229 + keyframe_manager.cpp:55-64
            const svo_kp2d q = project_point(s_frame.Rd, s_frame.t, camd, p3);
            a.kps2d[i] = q;
            if (q.x > 0 && q.y > 0 && q.x < a.width && q.y < a.height && !(flags & SVO_IGNORE_COMPLETELY))
                inside++;
        }
    }
    if (a.inside_count) {
        inside = wave_sum_i(inside);
        if ((tid & 63) == 0) s_inside[tid >> 6] = inside;
        __syncthreads();
        if (tid == 0) *a.inside_count = s_inside[0] + s_inside[1] + s_inside[2] + s_inside[3];
    }
}

void launch_filter(const FilterArgs* d_args, int batch, hipStream_t stream) {
    hipLaunchKernelGGL(filter_update_kernel, dim3(batch), dim3(256), 0, stream, d_args);
}

}  // namespace svo
