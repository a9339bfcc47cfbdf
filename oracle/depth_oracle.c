/* TEST INFRASTRUCTURE ONLY.  CPU restatement of FrameKDMap::ProcessDepth (depth image -> world-frame
 * obstacle cloud), SURVEY.md section 8 row f2.
 *
 * Only tests/ may load this.  The product (avoid_mpc_amd/csrc/depth.hip) never does.
 *
 * Restated (AM = /root/reference/roswrapper/ros/src/avoid_mpc):
 *   inv_depth        FrameKDMap::GetInvDepthImg<T>                 AM/src/FrameKDMap.cpp:76-89
 *   deptho_process   FrameKDMap::ProcessDepth (obstacle cloud)     :90-130
 *   uv2camera        FrameKDMap::UV2Camera                         :131-138
 *   scaled intrinsics: FrameKDMap constructor                      :21-24
 *
 * PARITY UNPINNED for two third-party pieces that are neither under /root/reference nor in this image:
 *   - cv::resize (OpenCV, version set by the ROS distribution; README.md of the reference names none).  The call
 *     `cv::resize(inv, inv, newSize, cv::INTER_MAX)` (:109) passes INTER_MAX as the `fx` argument, so the
 *     interpolation is the default INTER_LINEAR.  Restated from OpenCV's published algorithm for CV_32F
 *     (imgproc/resize.cpp, resizeGeneric_/HResizeLinear/VResizeLinear): source coordinate
 *     (float)((d + 0.5) * scale - 0.5) with scale = 1 / (dsize / ssize), floor, float weights (1 - f, f), border
 *     clamp with f = 0, horizontal pass then vertical pass in float.  OpenCV's SIMD build may fuse the vertical
 *     pass into FMAs; this restatement does not (1-ulp differences in the inverse depth are possible).
 *   - Eigen's 4x4 products under the reference's -march=native (FMA contraction, accumulation order).
 * The contract used for GPU parity is the arithmetic written here (compiled with -ffp-contract=off).
 */
#include <math.h>
#include <stddef.h>

typedef struct {
    double pixel2meter, depth_min, depth_max, resize_scale, fx, fy, cx, cy, Tbc[16];
} deptho_params; /* same layout as amk_depth_params */

static float inv_depth_raw(double raw_as_float, const deptho_params *p) {
    const float depth = (float)(raw_as_float * p->pixel2meter);       /* :80-81 (static_cast<float>(pixel) * double -> float) */
    if ((double)depth < p->depth_min || (double)depth > p->depth_max) return 0.f;
    return (float)(1.0 / (double)depth);                               /* :85 */
}
static float inv_at(const void *img, int type, int r, int c, int cols, const deptho_params *p) {
    const size_t i = (size_t)r * cols + c;
    const float v = type == 0 ? (float)((const unsigned short *)img)[i] : ((const float *)img)[i];
    return inv_depth_raw((double)v, p);
}
static void linear_tap(int d, double scale, int n, int *s0, int *s1, float *w1) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= n - 1) { f = 0.f; s = n - 1; }
    *s0 = s;
    *s1 = s + 1 < n ? s + 1 : n - 1;
    *w1 = f;
}

/* One scene.  Returns the number of points written to cloud ([W*H][stride] float32, row-major pixel order). */
int deptho_process(const void *depth, int type, int rows, int cols, const deptho_params *p, const double *Twb,
                   float *cloud, int stride, float *inv_small /* optional [H][W] */) {
    const int W = (int)((double)cols / p->resize_scale), H = (int)((double)rows / p->resize_scale);
    const double sx = 1.0 / ((double)W / (double)cols), sy = 1.0 / ((double)H / (double)rows);
    const double fx = p->fx / p->resize_scale, fy = p->fy / p->resize_scale;
    const double cx = p->cx / p->resize_scale, cy = p->cy / p->resize_scale;
    double M[12];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            double acc = Twb[4 * i + 0] * p->Tbc[0 + j];
            acc = acc + Twb[4 * i + 1] * p->Tbc[4 + j];
            acc = acc + Twb[4 * i + 2] * p->Tbc[8 + j];
            acc = acc + Twb[4 * i + 3] * p->Tbc[12 + j];
            M[4 * i + j] = acc;
        }
    int n = 0;
    for (int row = 0; row < H; ++row)
        for (int col = 0; col < W; ++col) {
            int x0, x1, y0, y1;
            float ax, ay;
            linear_tap(col, sx, cols, &x0, &x1, &ax);
            linear_tap(row, sy, rows, &y0, &y1, &ay);
            const float a0 = 1.f - ax, b0 = 1.f - ay;
            const float t0 = inv_at(depth, type, y0, x0, cols, p) * a0 + inv_at(depth, type, y0, x1, cols, p) * ax;
            const float t1 = inv_at(depth, type, y1, x0, cols, p) * a0 + inv_at(depth, type, y1, x1, cols, p) * ax;
            const float inv = t0 * b0 + t1 * ay;
            if (inv_small) inv_small[(size_t)row * W + col] = inv;
            const double invd = (double)inv;
            if (invd < 1e-2) continue;                                  /* :113-115 */
            const double d = 1.0 / invd;                                /* :116 */
            if (!(d > p->depth_min && d < p->depth_max)) continue;      /* :117 */
            const double xc = ((double)col - cx) * d / fx;              /* :133 */
            const double yc = ((double)row - cy) * d / fy;              /* :134 */
            float *o = cloud + (size_t)n * stride;
            o[0] = (float)(((M[0] * xc + M[1] * yc) + M[2] * d) + M[3]);
            o[1] = (float)(((M[4] * xc + M[5] * yc) + M[6] * d) + M[7]);
            o[2] = (float)(((M[8] * xc + M[9] * yc) + M[10] * d) + M[11]);
            ++n;
        }
    return n;
}
