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

template <typename T>
__global__ __launch_bounds__(kDepthThreads) void depth_to_cloud_kernel(const T *__restrict__ depth, long long scene_stride,
                                                                       DepthGeom g, const double *__restrict__ Twb,
                                                                       float *__restrict__ cloud, int point_stride,
                                                                       long long cloud_scene_stride,
                                                                       int *__restrict__ counts) {
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const T *img = depth + (long long)s * scene_stride;
    float *out = cloud + (long long)s * cloud_scene_stride;
    __shared__ double M[12];                      // rows 0..2 of Twb * Tbc
    __shared__ int wave_tot[kDepthThreads / 64];
    if (tid < 12) {                               // (mat4Twb * mParamTbc), evaluated once (:119-120)
        const int i = tid / 4, j = tid % 4;
        const double *A = Twb + 16 * s;
        double acc = A[4 * i + 0] * g.Tbc[0 + j];
        acc = acc + A[4 * i + 1] * g.Tbc[4 + j];
        acc = acc + A[4 * i + 2] * g.Tbc[8 + j];
        acc = acc + A[4 * i + 3] * g.Tbc[12 + j];
        M[tid] = acc;
    }
    __syncthreads();
    const int npix = g.W * g.H;
    int base = 0;
    for (int p0 = 0; p0 < npix; p0 += kDepthThreads) {
        const int p = p0 + tid;
        bool keep = false;
        float px = 0.f, py = 0.f, pz = 0.f;
        if (p < npix) {
            const int row = p / g.W, col = p % g.W;
            int x0, x1, y0, y1;
            float ax, ay;
            linear_tap(col, g.scale_x, g.cols, x0, x1, ax);
            linear_tap(row, g.scale_y, g.rows, y0, y1, ay);
            const float a0 = 1.f - ax, b0 = 1.f - ay;
            const float t0 = inv_depth(img, y0, x0, g.cols, g) * a0 + inv_depth(img, y0, x1, g.cols, g) * ax;  // horizontal pass
            const float t1 = inv_depth(img, y1, x0, g.cols, g) * a0 + inv_depth(img, y1, x1, g.cols, g) * ax;
            const float inv = t0 * b0 + t1 * ay;                                                                // vertical pass
            const double invd = (double)inv;
            if (!(invd < 1e-2)) {                                       // :113-115
                const double d = 1.0 / invd;                            // :116
                if (d > g.dmin && d < g.dmax) {                         // :117
                    const double xc = ((double)col - g.cx) * d / g.fx;  // UV2Camera :133-136
                    const double yc = ((double)row - g.cy) * d / g.fy;
                    const double X = ((M[0] * xc + M[1] * yc) + M[2] * d) + M[3];
                    const double Y = ((M[4] * xc + M[5] * yc) + M[6] * d) + M[7];
                    const double Z = ((M[8] * xc + M[9] * yc) + M[10] * d) + M[11];
                    px = (float)X; py = (float)Y; pz = (float)Z;        // pcl::PointXYZ is float32 (:122)
                    keep = true;
                }
            }
        }
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
        base += tot;
        __syncthreads();
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

int amk_depth_to_cloud_host(const void *h_depth, int depth_type, int rows, int cols, long long scene_stride, int n_scenes,
                            const amk_depth_params *params, const double *h_Twb, float *h_cloud, int point_stride,
                            long long cloud_scene_stride, int *h_counts) {
    if (!h_depth || !h_Twb || !h_cloud || !h_counts || n_scenes <= 0) return AMK_ERR_INVALID_ARG;
    if (depth_type != AMK_DEPTH_U16 && depth_type != AMK_DEPTH_F32) return AMK_ERR_UNSUPPORTED;
    if (amk_device_count() <= 0) return AMK_ERR_NO_DEVICE;
    const size_t esz = depth_type == AMK_DEPTH_U16 ? 2 : 4;
    amk::DevBuf<unsigned char> img;
    amk::DevBuf<double> twb;
    amk::DevBuf<float> cloud;
    amk::DevBuf<int> cnt;
    AMK_HIP(img.alloc((size_t)scene_stride * n_scenes * esz));
    AMK_HIP(twb.alloc((size_t)16 * n_scenes));
    AMK_HIP(cloud.alloc((size_t)cloud_scene_stride * n_scenes));
    AMK_HIP(cnt.alloc(n_scenes));
    AMK_HIP(hipMemcpy(img.p, h_depth, (size_t)scene_stride * n_scenes * esz, hipMemcpyHostToDevice));
    AMK_HIP(hipMemcpy(twb.p, h_Twb, sizeof(double) * 16 * n_scenes, hipMemcpyHostToDevice));
    int st = amk_depth_to_cloud(img.p, depth_type, rows, cols, scene_stride, n_scenes, params, twb.p, cloud.p, point_stride,
                                cloud_scene_stride, cnt.p, nullptr);
    if (st != AMK_OK) return st;
    AMK_HIP(hipDeviceSynchronize());
    AMK_HIP(hipMemcpy(h_counts, cnt.p, sizeof(int) * n_scenes, hipMemcpyDeviceToHost));
    AMK_HIP(hipMemcpy(h_cloud, cloud.p, sizeof(float) * (size_t)cloud_scene_stride * n_scenes, hipMemcpyDeviceToHost));
    return AMK_OK;
}

}  // extern "C"
