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
 *   deptho_edge      FrameKDMap::BuildEdgeCloud (row f3)           :176-214
 *                    = 8-bit quantisation (:180-193), cv::erode 3x3 (:194), cv::Canny(img, 0.1, 0.3) (:196),
 *                      back-projection of the edge pixels through mCurFrame.Twc * Tbc (:197-211)
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
 *   - cv::erode / cv::Canny (row f3).  Restated from OpenCV's published algorithms: erode with a 3x3 ones kernel
 *     and the default border (BORDER_CONSTANT, +inf: out-of-image taps are ignored); Canny with aperture 3 and the
 *     L1 norm: Sobel 3x3 with BORDER_REPLICATE, magnitude |dx| + |dy|, thresholds cvFloor(0.1) = cvFloor(0.3) = 0,
 *     non-maximum suppression with the fixed-point tangent test (TG22 = 13573, shift 15; horizontal: m > left &&
 *     m >= right, vertical: m > up && m >= down, diagonal: both strict; magnitudes outside the image are 0).  With
 *     both thresholds 0 every surviving pixel is a strong edge, so the hysteresis pass has nothing to decide.
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

static void small_inverse_depth(const void *depth, int type, int rows, int cols, const deptho_params *p, int W, int H,
                                float *inv_small) {
    const double sx = 1.0 / ((double)W / (double)cols), sy = 1.0 / ((double)H / (double)rows);
    for (int row = 0; row < H; ++row)
        for (int col = 0; col < W; ++col) {
            int x0, x1, y0, y1;
            float ax, ay;
            linear_tap(col, sx, cols, &x0, &x1, &ax);
            linear_tap(row, sy, rows, &y0, &y1, &ay);
            const float a0 = 1.f - ax, b0 = 1.f - ay;
            const float t0 = inv_at(depth, type, y0, x0, cols, p) * a0 + inv_at(depth, type, y0, x1, cols, p) * ax;
            const float t1 = inv_at(depth, type, y1, x0, cols, p) * a0 + inv_at(depth, type, y1, x1, cols, p) * ax;
            inv_small[(size_t)row * W + col] = t0 * b0 + t1 * ay;
        }
}

/* BuildEdgeCloud for one scene.  Twc = mCurFrame.Twc as the reference holds it at :209 (the PREVIOUS frame's
 * Twb * Tbc -- it is only updated after ProcessDepth, :50 -- and Tbc is applied once more).
 * quant / eroded / edges: optional [H][W] uint8 intermediates (edges: 255 where Canny fires).  work: [H*W] floats
 * + [H*W] shorts of scratch.  Returns the number of points written (0 when the obstacle cloud would be empty,
 * ProcessDepth :126-128). */
int deptho_edge(const void *depth, int type, int rows, int cols, const deptho_params *p, const double *Twc, float *cloud,
                int stride, unsigned char *quant, unsigned char *eroded, unsigned char *edges, float *work_inv,
                short *work_mag) {
    const int W = (int)((double)cols / p->resize_scale), H = (int)((double)rows / p->resize_scale);
    const double fx = p->fx / p->resize_scale, fy = p->fy / p->resize_scale;
    const double cx = p->cx / p->resize_scale, cy = p->cy / p->resize_scale;
    const double range = p->depth_max - p->depth_min;
    small_inverse_depth(depth, type, rows, cols, p, W, H, work_inv);
    int any = 0;
    for (int i = 0; i < W * H && !any; ++i) {
        const double invd = (double)work_inv[i];
        if (invd < 1e-2) continue;
        const double d = 1.0 / invd;
        any = d > p->depth_min && d < p->depth_max;
    }
    if (!any) return 0;
    /* :180-193 */
    for (int i = 0; i < W * H; ++i) {
        const float inv = work_inv[i];
        if ((double)inv > 1e-2) quant[i] = (unsigned char)((double)(1 / inv) / range * (double)200.0f);
        else quant[i] = 255;
    }
    /* cv::erode, 3x3 ones */
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            unsigned char m = 255;
            for (int dr = -1; dr <= 1; ++dr)
                for (int dc = -1; dc <= 1; ++dc) {
                    const int rr = r + dr, cc = c + dc;
                    if (rr < 0 || rr >= H || cc < 0 || cc >= W) continue;
                    if (quant[rr * W + cc] < m) m = quant[rr * W + cc];
                }
            eroded[r * W + c] = m;
        }
    /* cv::Canny: Sobel (replicated border), L1 magnitude */
#define EPX(r, c) ((int)eroded[((r) < 0 ? 0 : ((r) >= H ? H - 1 : (r))) * W + ((c) < 0 ? 0 : ((c) >= W ? W - 1 : (c)))])
#define SOBX(r, c) ((EPX(r - 1, c + 1) + 2 * EPX(r, c + 1) + EPX(r + 1, c + 1)) - (EPX(r - 1, c - 1) + 2 * EPX(r, c - 1) + EPX(r + 1, c - 1)))
#define SOBY(r, c) ((EPX(r + 1, c - 1) + 2 * EPX(r + 1, c) + EPX(r + 1, c + 1)) - (EPX(r - 1, c - 1) + 2 * EPX(r - 1, c) + EPX(r - 1, c + 1)))
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            const int dx = SOBX(r, c), dy = SOBY(r, c);
            work_mag[r * W + c] = (short)((dx < 0 ? -dx : dx) + (dy < 0 ? -dy : dy));
        }
#define MAG(r, c) (((r) < 0 || (r) >= H || (c) < 0 || (c) >= W) ? 0 : (int)work_mag[(r) * W + (c)])
    double M[12];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            double acc = Twc[4 * i + 0] * p->Tbc[0 + j];
            acc = acc + Twc[4 * i + 1] * p->Tbc[4 + j];
            acc = acc + Twc[4 * i + 2] * p->Tbc[8 + j];
            acc = acc + Twc[4 * i + 3] * p->Tbc[12 + j];
            M[4 * i + j] = acc;
        }
    int n = 0;
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            const int m = MAG(r, c);
            int edge = 0;
            if (m > 0) {
                const int xs = SOBX(r, c), ys = SOBY(r, c);
                const int x = xs < 0 ? -xs : xs, y = (ys < 0 ? -ys : ys) << 15;
                const int tg22x = x * 13573;
                if (y < tg22x) edge = m > MAG(r, c - 1) && m >= MAG(r, c + 1);
                else {
                    const int tg67x = tg22x + (x << 16);
                    if (y > tg67x) edge = m > MAG(r - 1, c) && m >= MAG(r + 1, c);
                    else {
                        const int s = ((xs ^ ys) < 0) ? -1 : 1;
                        edge = m > MAG(r - 1, c - s) && m > MAG(r + 1, c + s);
                    }
                }
            }
            if (edges) edges[r * W + c] = edge ? 255 : 0;
            if (!edge) continue;
            double d = (double)(float)eroded[r * W + c];        /* :199 */
            d = d * range / 200.0;                              /* :200 */
            if (d > p->depth_max || d < p->depth_min) continue; /* :201-203 */
            const double xc = ((double)c - cx) * d / fx, yc = ((double)r - cy) * d / fy;
            float *o = cloud + (size_t)n * stride;
            o[0] = (float)(((M[0] * xc + M[1] * yc) + M[2] * d) + M[3]);
            o[1] = (float)(((M[4] * xc + M[5] * yc) + M[6] * d) + M[7]);
            o[2] = (float)(((M[8] * xc + M[9] * yc) + M[10] * d) + M[11]);
            ++n;
        }
    return n;
#undef EPX
#undef SOBX
#undef SOBY
#undef MAG
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
