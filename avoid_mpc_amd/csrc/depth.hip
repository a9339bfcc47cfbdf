// Depth image -> world-frame obstacle cloud for gfx950: FrameKDMap::ProcessDepth
// (AM/src/FrameKDMap.cpp:90-130 with GetInvDepthImg :76-89 and UV2Camera :131-138; AM = roswrapper/ros/src/avoid_mpc
// in the reference tree).  SURVEY.md section 8 row f2: the per-pixel streaming step right before the tree build.
//
// One block per scene.  Only the 2x2 raw pixels under each down-scaled pixel are read (4 % of a 640x480 image at
// resize_scale 10): the reference inverts the whole image and then lets cv::resize pick the same four taps.
// The kept pixels are appended in row-major order (ballot + popcount prefix per wave, LDS combine per block),
// which is the order the reference's emplace_back produces and therefore the index space of the KD tree.
#include <cstring>

#include "amk_common.h"

namespace {

constexpr int kDepthThreads = 256;

struct DepthGeom {
    int rows, cols, W, H;
    double scale_x, scale_y;       // source pixels per destination pixel (cv::resize: 1 / (dsize / ssize))
    double pixel2meter, dmin, dmax;
    double fx, fy, cx, cy;         // already divided by resize_scale
    double Tbc[16];
};

#pragma clang fp contract(off)
template <typename T>
__device__ __forceinline__ float inv_depth(const T *img, int r, int c, int cols, const DepthGeom &g) {
    const float depth = (float)((double)(float)img[(size_t)r * cols + c] * g.pixel2meter);   // :80-81
    if ((double)depth < g.dmin || (double)depth > g.dmax) return 0.f;                        // :82-83
    return (float)(1.0 / (double)depth);                                                      // :85
}

// cv::resize INTER_LINEAR tap of destination index d along an axis of n source samples: source index and the
// weight of the second tap, in float as OpenCV computes them for CV_32F (imgproc/resize.cpp, resizeGeneric)
__device__ __forceinline__ void linear_tap(int d, double scale, int n, int &s0, int &s1, float &w1) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= n - 1) { f = 0.f; s = n - 1; }
    s0 = s;
    s1 = s + 1 < n ? s + 1 : n - 1;
    w1 = f;
}

// bilinear inverse depth of destination pixel (row, col): cv::resize's two float passes (horizontal, then vertical)
template <typename T>
__device__ __forceinline__ float small_inv(const T *img, int row, int col, const DepthGeom &g) {
    int x0, x1, y0, y1;
    float ax, ay;
    linear_tap(col, g.scale_x, g.cols, x0, x1, ax);
    linear_tap(row, g.scale_y, g.rows, y0, y1, ay);
    const float a0 = 1.f - ax, b0 = 1.f - ay;
    const float t0 = inv_depth(img, y0, x0, g.cols, g) * a0 + inv_depth(img, y0, x1, g.cols, g) * ax;
    const float t1 = inv_depth(img, y1, x0, g.cols, g) * a0 + inv_depth(img, y1, x1, g.cols, g) * ax;
    return t0 * b0 + t1 * ay;
}

// rows 0..2 of A * Tbc into M[12] (threads 0..11), ordered sums without FMA
__device__ __forceinline__ void pose_times_tbc(const double *A, const DepthGeom &g, double *M) {
    const int tid = threadIdx.x;
    if (tid < 12) {
        const int i = tid / 4, j = tid % 4;
        double acc = A[4 * i + 0] * g.Tbc[0 + j];
        acc = acc + A[4 * i + 1] * g.Tbc[4 + j];
        acc = acc + A[4 * i + 2] * g.Tbc[8 + j];
        acc = acc + A[4 * i + 3] * g.Tbc[12 + j];
        M[tid] = acc;
    }
}

// appends the kept points of one round in thread order; returns the new base (every thread)
__device__ __forceinline__ int append_round(bool keep, float px, float py, float pz, float *out, int point_stride, int base,
                                            int *wave_tot) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned long long m = __ballot(keep);
    if (lane == 0) wave_tot[w] = __popcll(m);
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int j = 0; j < kDepthThreads / 64; ++j) {
        const int t = wave_tot[j];
        woff += j < w ? t : 0;
        tot += t;
    }
    if (keep) {
        float *o = out + (size_t)(base + woff + __popcll(m & ((1ull << lane) - 1ull))) * point_stride;
        o[0] = px; o[1] = py; o[2] = pz;
    }
    __syncthreads();
    return base + tot;
}

// FrameKDMap::BuildEdgeCloud (FrameKDMap.cpp:176-214) for one scene per block; the whole down-scaled frame lives in
// LDS: quantised depth (u8), its 3x3 erosion (u8), Sobel L1 magnitude (i16).  cv::Canny(., 0.1, 0.3) on 8-bit data =
// non-maximum suppression of the non-zero magnitudes (both thresholds floor to 0: every survivor is a strong edge).
template <typename T>
__global__ __launch_bounds__(kDepthThreads) void depth_to_edge_kernel(const T *__restrict__ depth, long long scene_stride,
                                                                      DepthGeom g, const double *__restrict__ Twc,
                                                                      float *__restrict__ cloud, int point_stride,
                                                                      long long cloud_scene_stride,
                                                                      int *__restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char edge_smem[];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int W = g.W, H = g.H, npix = W * H;
    unsigned char *quant = edge_smem, *ero = edge_smem + npix;
    short *mag = reinterpret_cast<short *>(edge_smem + 2 * ((npix + 1) & ~1));
    const T *img = depth + (long long)s * scene_stride;
    float *out = cloud + (long long)s * cloud_scene_stride;
    __shared__ double M[12];
    __shared__ int wave_tot[kDepthThreads / 64];
    __shared__ int any_obstacle;
    if (tid == 0) any_obstacle = 0;
    pose_times_tbc(Twc + 16 * s, g, M);   // (mCurFrame.Twc * mParamTbc), :209
    __syncthreads();
    const double range = g.dmax - g.dmin;
    bool any = false;
    for (int p = tid; p < npix; p += kDepthThreads) {
        const float inv = small_inv(img, p / W, p % W, g);
        const double invd = (double)inv;
        if (!(invd < 1e-2)) {               // would ProcessDepth keep this pixel? (:113-117)
            const double d = 1.0 / invd;
            any = any || (d > g.dmin && d < g.dmax);
        }
        quant[p] = invd > 1e-2 ? (unsigned char)((double)(1 / inv) / range * (double)200.0f) : (unsigned char)255;  // :183-191
    }
    if (any) any_obstacle = 1;
    __syncthreads();
    if (!any_obstacle) {                    // cloud->empty(): BuildEdgeCloud is not reached (:126-128)
        if (tid == 0) counts[s] = 0;
        return;
    }
    for (int p = tid; p < npix; p += kDepthThreads) {   // cv::erode 3x3, out-of-image taps ignored
        const int r = p / W, c = p % W;
        int m = 255;
        for (int dr = -1; dr <= 1; ++dr)
            for (int dc = -1; dc <= 1; ++dc) {
                const int rr = r + dr, cc = c + dc;
                if (rr >= 0 && rr < H && cc >= 0 && cc < W) m = min(m, (int)quant[rr * W + cc]);
            }
        ero[p] = (unsigned char)m;
    }
    __syncthreads();
    auto epx = [&](int r, int c) { return (int)ero[min(max(r, 0), H - 1) * W + min(max(c, 0), W - 1)]; };  // BORDER_REPLICATE
    auto sobx = [&](int r, int c) {
        return (epx(r - 1, c + 1) + 2 * epx(r, c + 1) + epx(r + 1, c + 1)) - (epx(r - 1, c - 1) + 2 * epx(r, c - 1) + epx(r + 1, c - 1));
    };
    auto soby = [&](int r, int c) {
        return (epx(r + 1, c - 1) + 2 * epx(r + 1, c) + epx(r + 1, c + 1)) - (epx(r - 1, c - 1) + 2 * epx(r - 1, c) + epx(r - 1, c + 1));
    };
    for (int p = tid; p < npix; p += kDepthThreads) mag[p] = (short)(abs(sobx(p / W, p % W)) + abs(soby(p / W, p % W)));
    __syncthreads();
    auto mg = [&](int r, int c) { return (r < 0 || r >= H || c < 0 || c >= W) ? 0 : (int)mag[r * W + c]; };
    int base = 0;
    for (int p0 = 0; p0 < npix; p0 += kDepthThreads) {
        const int p = p0 + tid;
        bool keep = false;
        float px = 0.f, py = 0.f, pz = 0.f;
        if (p < npix) {
            const int r = p / W, c = p % W, m = mag[p];
            bool edge = false;
            if (m > 0) {
                const int xs = sobx(r, c), ys = soby(r, c);
                const int x = abs(xs), y = abs(ys) << 15, tg22x = x * 13573;
                if (y < tg22x) edge = m > mg(r, c - 1) && m >= mg(r, c + 1);
                else if (y > tg22x + (x << 16)) edge = m > mg(r - 1, c) && m >= mg(r + 1, c);
                else {
                    const int sg = ((xs ^ ys) < 0) ? -1 : 1;
                    edge = m > mg(r - 1, c - sg) && m > mg(r + 1, c + sg);
                }
            }
            if (edge) {
                double d = (double)(float)ero[p];   // :199
                d = d * range / 200.0;              // :200
                if (!(d > g.dmax || d < g.dmin)) {  // :201-203
                    const double xc = ((double)c - g.cx) * d / g.fx, yc = ((double)r - g.cy) * d / g.fy;
                    px = (float)(((M[0] * xc + M[1] * yc) + M[2] * d) + M[3]);
                    py = (float)(((M[4] * xc + M[5] * yc) + M[6] * d) + M[7]);
                    pz = (float)(((M[8] * xc + M[9] * yc) + M[10] * d) + M[11]);
                    keep = true;
                }
            }
        }
        base = append_round(keep, px, py, pz, out, point_stride, base, wave_tot);
    }
    if (tid == 0) counts[s] = base;
}

template <typename T>
__global__ __launch_bounds__(kDepthThreads) void depth_to_cloud_kernel(const T *__restrict__ depth, long long scene_stride,
                                                                       DepthGeom g, const double *__restrict__ Twb,
                                                                       float *__restrict__ cloud, int point_stride,
                                                                       long long cloud_scene_stride,
                                                                       int *__restrict__ counts) {
    const int s = blockIdx.x, tid = threadIdx.x;
    const T *img = depth + (long long)s * scene_stride;
    float *out = cloud + (long long)s * cloud_scene_stride;
    __shared__ double M[12];                      // rows 0..2 of Twb * Tbc
    __shared__ int wave_tot[kDepthThreads / 64];
    pose_times_tbc(Twb + 16 * s, g, M);           // (mat4Twb * mParamTbc), evaluated once (:119-120)
    __syncthreads();
    const int npix = g.W * g.H;
    int base = 0;
    for (int p0 = 0; p0 < npix; p0 += kDepthThreads) {
        const int p = p0 + tid;
        bool keep = false;
        float px = 0.f, py = 0.f, pz = 0.f;
        if (p < npix) {
            const int row = p / g.W, col = p % g.W;
            const double invd = (double)small_inv(img, row, col, g);
            if (!(invd < 1e-2)) {                                       // :113-115
                const double d = 1.0 / invd;                            // :116
                if (d > g.dmin && d < g.dmax) {                         // :117
                    const double xc = ((double)col - g.cx) * d / g.fx;  // UV2Camera :133-136
                    const double yc = ((double)row - g.cy) * d / g.fy;
                    px = (float)(((M[0] * xc + M[1] * yc) + M[2] * d) + M[3]);   // pcl::PointXYZ is float32 (:122)
                    py = (float)(((M[4] * xc + M[5] * yc) + M[6] * d) + M[7]);
                    pz = (float)(((M[8] * xc + M[9] * yc) + M[10] * d) + M[11]);
                    keep = true;
                }
            }
        }
        base = append_round(keep, px, py, pz, out, point_stride, base, wave_tot);
    }
    if (tid == 0) counts[s] = base;
}

int make_geom(int rows, int cols, const amk_depth_params *p, DepthGeom &g) {
    if (!p || rows <= 0 || cols <= 0 || !(p->resize_scale > 0)) return AMK_ERR_INVALID_ARG;
    g.rows = rows; g.cols = cols;
    g.W = (int)((double)cols / p->resize_scale);   // mParamWidth = cols / mParamDepthScale (int <- double, :106)
    g.H = (int)((double)rows / p->resize_scale);
    if (g.W <= 0 || g.H <= 0) return AMK_ERR_INVALID_ARG;
    g.scale_x = 1.0 / ((double)g.W / (double)cols);  // cv::resize: inv_scale = dsize / ssize; scale = 1 / inv_scale
    g.scale_y = 1.0 / ((double)g.H / (double)rows);
    g.pixel2meter = p->pixel2meter; g.dmin = p->depth_min; g.dmax = p->depth_max;
    g.fx = p->fx / p->resize_scale; g.fy = p->fy / p->resize_scale;   // FrameKDMap.cpp:21-24
    g.cx = p->cx / p->resize_scale; g.cy = p->cy / p->resize_scale;
    std::memcpy(g.Tbc, p->Tbc, sizeof g.Tbc);
    return AMK_OK;
}

}  // namespace

extern "C" {

int amk_depth_out_size(int rows, int cols, double resize_scale, int *out_w, int *out_h) {
    if (rows <= 0 || cols <= 0 || !(resize_scale > 0) || !out_w || !out_h) return AMK_ERR_INVALID_ARG;
    *out_w = (int)((double)cols / resize_scale);
    *out_h = (int)((double)rows / resize_scale);
    return (*out_w > 0 && *out_h > 0) ? AMK_OK : AMK_ERR_INVALID_ARG;
}

int amk_depth_to_cloud(const void *d_depth, int depth_type, int rows, int cols, long long scene_stride, int n_scenes,
                       const amk_depth_params *params, const double *d_Twb, float *d_cloud, int point_stride,
                       long long cloud_scene_stride, int *d_counts, void *stream) {
    if (!d_depth || !d_Twb || !d_cloud || !d_counts || n_scenes <= 0 || point_stride < 3) return AMK_ERR_INVALID_ARG;
    if (depth_type != AMK_DEPTH_U16 && depth_type != AMK_DEPTH_F32) return AMK_ERR_UNSUPPORTED;
    DepthGeom g;
    int st = make_geom(rows, cols, params, g);
    if (st != AMK_OK) return st;
    if (scene_stride < (long long)rows * cols || cloud_scene_stride < (long long)g.W * g.H * point_stride)
        return AMK_ERR_INVALID_ARG;
    if (amk_device_count() <= 0) return AMK_ERR_NO_DEVICE;
    if (depth_type == AMK_DEPTH_U16)
        hipLaunchKernelGGL(depth_to_cloud_kernel<unsigned short>, dim3(n_scenes), dim3(kDepthThreads), 0, (hipStream_t)stream,
                           (const unsigned short *)d_depth, scene_stride, g, d_Twb, d_cloud, point_stride, cloud_scene_stride,
                           d_counts);
    else
        hipLaunchKernelGGL(depth_to_cloud_kernel<float>, dim3(n_scenes), dim3(kDepthThreads), 0, (hipStream_t)stream,
                           (const float *)d_depth, scene_stride, g, d_Twb, d_cloud, point_stride, cloud_scene_stride, d_counts);
    AMK_HIP(hipGetLastError());
    return AMK_OK;
}

int amk_depth_to_edge_cloud(const void *d_depth, int depth_type, int rows, int cols, long long scene_stride, int n_scenes,
                            const amk_depth_params *params, const double *d_Twc, float *d_cloud, int point_stride,
                            long long cloud_scene_stride, int *d_counts, void *stream) {
    if (!d_depth || !d_Twc || !d_cloud || !d_counts || n_scenes <= 0 || point_stride < 3) return AMK_ERR_INVALID_ARG;
    if (depth_type != AMK_DEPTH_U16 && depth_type != AMK_DEPTH_F32) return AMK_ERR_UNSUPPORTED;
    DepthGeom g;
    int st = make_geom(rows, cols, params, g);
    if (st != AMK_OK) return st;
    if (scene_stride < (long long)rows * cols || cloud_scene_stride < (long long)g.W * g.H * point_stride)
        return AMK_ERR_INVALID_ARG;
    if ((long long)g.W * g.H > AMK_EDGE_MAX_PIXELS) return AMK_ERR_UNSUPPORTED;
    if (amk_device_count() <= 0) return AMK_ERR_NO_DEVICE;
    const size_t npix = (size_t)g.W * g.H, lds = 2 * ((npix + 1) & ~(size_t)1) + 2 * npix;  // quant, eroded (u8), magnitude (i16)
    if (depth_type == AMK_DEPTH_U16)
        hipLaunchKernelGGL(depth_to_edge_kernel<unsigned short>, dim3(n_scenes), dim3(kDepthThreads), lds, (hipStream_t)stream,
                           (const unsigned short *)d_depth, scene_stride, g, d_Twc, d_cloud, point_stride, cloud_scene_stride,
                           d_counts);
    else
        hipLaunchKernelGGL(depth_to_edge_kernel<float>, dim3(n_scenes), dim3(kDepthThreads), lds, (hipStream_t)stream,
                           (const float *)d_depth, scene_stride, g, d_Twc, d_cloud, point_stride, cloud_scene_stride, d_counts);
    AMK_HIP(hipGetLastError());
    return AMK_OK;
}

static int depth_host_common(int edge, const void *h_depth, int depth_type, int rows, int cols, long long scene_stride,
                             int n_scenes, const amk_depth_params *params, const double *h_T, float *h_cloud,
                             int point_stride, long long cloud_scene_stride, int *h_counts) {
    if (!h_depth || !h_T || !h_cloud || !h_counts || n_scenes <= 0) return AMK_ERR_INVALID_ARG;
    if (depth_type != AMK_DEPTH_U16 && depth_type != AMK_DEPTH_F32) return AMK_ERR_UNSUPPORTED;
    if (amk_device_count() <= 0) return AMK_ERR_NO_DEVICE;
    const size_t esz = depth_type == AMK_DEPTH_U16 ? 2 : 4;
    amk::DevBuf<unsigned char> img;
    amk::DevBuf<double> pose;
    amk::DevBuf<float> cloud;
    amk::DevBuf<int> cnt;
    AMK_HIP(img.alloc((size_t)scene_stride * n_scenes * esz));
    AMK_HIP(pose.alloc((size_t)16 * n_scenes));
    AMK_HIP(cloud.alloc((size_t)cloud_scene_stride * n_scenes));
    AMK_HIP(cnt.alloc(n_scenes));
    AMK_HIP(hipMemcpy(img.p, h_depth, (size_t)scene_stride * n_scenes * esz, hipMemcpyHostToDevice));
    AMK_HIP(hipMemcpy(pose.p, h_T, sizeof(double) * 16 * n_scenes, hipMemcpyHostToDevice));
    int st = edge ? amk_depth_to_edge_cloud(img.p, depth_type, rows, cols, scene_stride, n_scenes, params, pose.p, cloud.p,
                                            point_stride, cloud_scene_stride, cnt.p, nullptr)
                  : amk_depth_to_cloud(img.p, depth_type, rows, cols, scene_stride, n_scenes, params, pose.p, cloud.p,
                                       point_stride, cloud_scene_stride, cnt.p, nullptr);
    if (st != AMK_OK) return st;
    AMK_HIP(hipDeviceSynchronize());
    AMK_HIP(hipMemcpy(h_counts, cnt.p, sizeof(int) * n_scenes, hipMemcpyDeviceToHost));
    AMK_HIP(hipMemcpy(h_cloud, cloud.p, sizeof(float) * (size_t)cloud_scene_stride * n_scenes, hipMemcpyDeviceToHost));
    return AMK_OK;
}

int amk_depth_to_edge_cloud_host(const void *h_depth, int depth_type, int rows, int cols, long long scene_stride,
                                 int n_scenes, const amk_depth_params *params, const double *h_Twc, float *h_cloud,
                                 int point_stride, long long cloud_scene_stride, int *h_counts) {
    return depth_host_common(1, h_depth, depth_type, rows, cols, scene_stride, n_scenes, params, h_Twc, h_cloud,
                             point_stride, cloud_scene_stride, h_counts);
}

int amk_depth_to_cloud_host(const void *h_depth, int depth_type, int rows, int cols, long long scene_stride, int n_scenes,
                            const amk_depth_params *params, const double *h_Twb, float *h_cloud, int point_stride,
                            long long cloud_scene_stride, int *h_counts) {
    return depth_host_common(0, h_depth, depth_type, rows, cols, scene_stride, n_scenes, params, h_Twb, h_cloud,
                             point_stride, cloud_scene_stride, h_counts);
}

}  // extern "C"
