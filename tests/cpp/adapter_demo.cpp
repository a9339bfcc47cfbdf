// Exercises the reference-shaped C++ adapters (include/avoid_mpc_amd/*.hpp) the way the reference's
// own code uses the classes they replace; driven by tests/test_cpp_adapters_gpu.py, which compares
// the numbers written here with the oracle.
//   adapter_demo <in.bin> <out.bin>
// in.bin : int32 n, ne, N, K, max_iter, nq ; double params[6+25+4+4+5] ; float cloud[n*3], edge[ne*3] ;
//          double odom[10] (pos vel acc yaw) ; double ref_path[N*10] ; double queries[nq*3]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "avoid_mpc_amd/avoidance_step.hpp"

using namespace avoid_mpc_amd;

struct Cloud {  // stands where pcl::PointCloud<pcl::PointXYZ> stands in the reference
    std::vector<PointXYZ> points;
};

template <class T>
static void rd(FILE *f, T *p, size_t n) {
    if (fread(p, sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
}
template <class T>
static void wr(FILE *f, const T *p, size_t n) { fwrite(p, sizeof(T), n, f); }

int main(int argc, char **argv) {
    if (argc < 3) return 1;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    int hdr[6];
    rd(f, hdr, 6);
    const int n = hdr[0], ne = hdr[1], N = hdr[2], K = hdr[3], max_iter = hdr[4], nq = hdr[5];
    double sc[6];  // T, dt, speed, safety, decay, height
    rd(f, sc, 6);
    std::vector<double> weights(25), tau(4), gains(4), lim(5);  // lim: aMinZ aMaxZ aMaxXy aMaxYawDot radius
    rd(f, weights.data(), 25); rd(f, tau.data(), 4); rd(f, gains.data(), 4); rd(f, lim.data(), 5);
    std::vector<float> cl((size_t)n * 3), ed((size_t)ne * 3);
    rd(f, cl.data(), cl.size()); rd(f, ed.data(), ed.size());
    double od[10];
    rd(f, od, 10);
    std::vector<double> ref((size_t)N * 10), qs((size_t)nq * 3);
    rd(f, ref.data(), ref.size()); rd(f, qs.data(), qs.size());
    fclose(f);

    auto cloud = std::make_shared<Cloud>(), edge = std::make_shared<Cloud>();
    for (int i = 0; i < n; ++i) cloud->points.emplace_back(cl[3 * i], cl[3 * i + 1], cl[3 * i + 2]);
    for (int i = 0; i < ne; ++i) edge->points.emplace_back(ed[3 * i], ed[3 * i + 1], ed[3 * i + 2]);

    FILE *o = fopen(argv[2], "wb");
    // 1. KDTreeTwo used directly, as FrameKDMap.cpp:45,264 does
    KDTreeTwo<double> tree;
    tree.InitializeNew(cloud);
    for (int i = 0; i < nq; ++i) {
        tree.SearchForNearest(qs[3 * i], qs[3 * i + 1], qs[3 * i + 2], K);
        int c = (int)tree.indices.size();
        wr(o, &c, 1);
        wr(o, tree.indices.data(), c);
        wr(o, tree.squared_distances.data(), c);
        for (int j = 0; j < c; ++j) { float p[3] = {tree.closest_pts[j].x, tree.closest_pts[j].y, tree.closest_pts[j].z}; wr(o, p, 3); }
    }
    {   // 1b. the same queries in one device round trip: identical to the one-at-a-time results
        std::vector<std::vector<int>> bi; std::vector<std::vector<double>> bd;
        tree.SearchForNearestBatch(qs.data(), nq, K, bi, bd);
        int same = 1;
        for (int i = 0; i < nq; ++i) {
            tree.SearchForNearest(qs[3 * i], qs[3 * i + 1], qs[3 * i + 2], K);
            if (bi[i] != tree.indices || bd[i] != tree.squared_distances) same = 0;
        }
        wr(o, &same, 1);
    }
    {   // 1c. SetNanoflannTieOrder: cloud and queries on a lattice (equal distances are the rule); indices as the reference's
        auto qc = std::make_shared<Cloud>();
        for (int i = 0; i < n; ++i)
            qc->points.emplace_back(std::round(cl[3 * i] * 4) / 4, std::round(cl[3 * i + 1] * 4) / 4, std::round(cl[3 * i + 2] * 4) / 4);
        KDTreeTwo<double> lat;
        lat.SetNanoflannTieOrder(true);
        lat.InitializeNew(qc);
        for (int i = 0; i < nq; ++i) {
            lat.SearchForNearest(std::round(qs[3 * i] * 8) / 8, std::round(qs[3 * i + 1] * 8) / 8, std::round(qs[3 * i + 2] * 8) / 8, K);
            int c = (int)lat.indices.size();
            wr(o, &c, 1);
            wr(o, lat.indices.data(), c);
            wr(o, lat.squared_distances.data(), c);
        }
    }
    // 2. FrameKDMap front end, as AvoidanceStateMachine.cpp:214,264,270 does
    FrameKDMap map;
    map.AddVertex(cloud, edge);
    for (int i = 0; i < nq; ++i) {
        Vector3d p(qs[3 * i], qs[3 * i + 1], qs[3 * i + 2]);
        std::vector<Vector3d> pts; std::vector<double> d2;
        map.QueryNearest(p, K, pts, d2);
        double nd = map.GetNearestDistance(p);
        std::vector<Vector3d> ep; std::vector<double> ed2;
        map.QueryNearest(p, 1, ep, ed2, true);
        int c = (int)d2.size(), ce = (int)ed2.size();
        wr(o, &c, 1); wr(o, d2.data(), c); wr(o, &nd, 1); wr(o, &ce, 1); wr(o, ed2.data(), ce);
    }
    // 3. the TASK step (SetupMPC + Step), twice: second call carries the warm start
    AvoidanceTaskStep task(sc[0], sc[1], K, max_iter, sc[2], sc[3], sc[4], sc[5]);
    amk_mpc_setup_weights(task.mpc(), weights.data());
    amk_mpc_setup_tau(task.mpc(), tau.data());
    amk_mpc_setup_gains(task.mpc(), gains.data());
    amk_mpc_set_drone_accel_limits(task.mpc(), lim[0], lim[1], lim[2], lim[3]);
    amk_mpc_set_drone_radius(task.mpc(), lim[4]);
    task.RefPath() = ref;
    OdomState os;
    for (int i = 0; i < 3; ++i) { os.pos[i] = od[i]; os.vel[i] = od[3 + i]; os.acc[i] = od[6 + i]; }
    os.yaw = od[9];
    for (int rep = 0; rep < 2; ++rep) {
        std::vector<double> u; std::vector<std::vector<double>> x0;
        bool safe = task.Step(map, os, u, x0);
        int s = safe ? 1 : 0;
        wr(o, &s, 1); wr(o, task.Flags(), 4); wr(o, u.data(), 4);
        for (auto &row : x0) wr(o, row.data(), 14);
        wr(o, task.RefPath().data(), (size_t)N * 10);
    }
    // 4. ObstacleAvoidanceMPC on its own (HighLvlMpc.cpp), K inferred from the vector length
    {
        ObstacleAvoidanceMPC mpc(sc[0], sc[1], "so/mpc_obstacle_v2.so");
        mpc.SetupWeights(weights); mpc.SetupTau(tau); mpc.SetupGains(gains);
        mpc.SetDroneAccelLimits(lim[0], lim[1], lim[2], lim[3]); mpc.SetDroneRadius(lim[4]);
        std::vector<double> vecRef(20 + 10 * N + 3 * K * N, 0.0);
        for (int i = 0; i < 10; ++i) vecRef[i] = (i < 3) ? od[i] : (i >= 4 && i < 7 ? od[3 + i - 4] : 0.0);
        for (int i = 0; i < 10 * N; ++i) vecRef[10 + i] = ref[i];
        for (int i = 0; i < 3 * K * N; ++i) vecRef[10 + 10 * N + i] = 10000.0;  // no obstacles
        for (int i = 0; i < 10; ++i) vecRef[10 + 10 * N + 3 * K * N + i] = ref[10 * (N - 1) + i];
        std::vector<double> u; std::vector<std::vector<double>> x0;
        mpc.Solve(vecRef, u, x0, true);
        wr(o, vecRef.data(), vecRef.size()); wr(o, u.data(), 4); wr(o, mpc.LastSolveInfo(), 4);
    }
    // 5. keyframes (FrameKDMap.cpp:437-488) + multi-frame QueryNearest (:347-375): three frames = the same cloud
    //    seen from a camera that moves 1.5 m per frame; the second and third update sweep + rebuild on the GPU
    {
        FrameKDMap kmap;
        kmap.ptIsInCurFrame = [](const Vector3d &) { return false; };  // force the multi-frame path
        int th_count = 10;
        double th_dist = 0.1;
        for (int f = 0; f < 3; ++f) {
            auto cf = std::make_shared<Cloud>(), ef = std::make_shared<Cloud>();
            for (int i = 0; i < n; ++i)
                if (cl[3 * i] >= 1.5f * f && cl[3 * i] < 1.5f * f + 12.f) cf->points.emplace_back(cl[3 * i], cl[3 * i + 1], cl[3 * i + 2]);
            for (int i = 0; i < ne; ++i) ef->points.emplace_back(ed[3 * i], ed[3 * i + 1], ed[3 * i + 2]);
            kmap.AddVertex(cf, ef);
            kmap.KeyframeUpdate(Vector3d(1.5 * f - 30.0, 0, 1.5), Vector3d(1, 0, 0), 0.1, th_dist, th_count, 100);
            int kc = (int)kmap.KeyFrameCount(), lo = kmap.LastSweepOutliers();
            wr(o, &kc, 1); wr(o, &lo, 1);
        }
        for (int i = 0; i < nq; ++i) {
            Vector3d p(qs[3 * i], qs[3 * i + 1], qs[3 * i + 2]);
            std::vector<Vector3d> pts; std::vector<double> d2;
            kmap.QueryNearest(p, K, pts, d2);
            double nd = kmap.GetNearestDistance(p);
            int c = (int)d2.size();
            wr(o, &c, 1); wr(o, d2.data(), c); wr(o, &nd, 1);
        }
    }
    // 6. ProcessDepth (FrameKDMap.cpp:90-130): a synthetic 16UC1 depth frame (millimetres) made here from a closed
    //    formula the test repeats; yaml intrinsics at 120 x 160, resize_scale 4, a yawed body pose
    {
        const int rows = 120, cols = 160;
        std::vector<unsigned short> img((size_t)rows * cols);
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c) {
                unsigned v = 1500u + (unsigned)((r * 37 + c * 91) % 4000);
                if ((r * 7 + c * 13) % 11 == 0) v = 0;  // holes
                img[(size_t)r * cols + c] = (unsigned short)v;
            }
        FrameKDMap dmap;
        dmap.depthParams.pixel2meter = 1e-3; dmap.depthParams.depth_min = 0.1; dmap.depthParams.depth_max = 100.0;
        dmap.depthParams.resize_scale = 4.0;
        dmap.depthParams.fx = 80.0; dmap.depthParams.fy = 80.0; dmap.depthParams.cx = 80.0; dmap.depthParams.cy = 60.0;
        const double Tbc[16] = {0, 0, 1, 0.1, -1, 0, 0, 0.0, 0, -1, 0, 0.05, 0, 0, 0, 1};
        for (int i = 0; i < 16; ++i) dmap.depthParams.Tbc[i] = Tbc[i];
        const double Twb[16] = {0.8, -0.6, 0, 2.0, 0.6, 0.8, 0, -1.0, 0, 0, 1, 1.5, 0, 0, 0, 1};
        Cloud dc;
        dmap.ProcessDepth(img.data(), AMK_DEPTH_U16, rows, cols, Twb, dc);
        int cnt = (int)dc.points.size();
        wr(o, &cnt, 1);
        for (auto &p : dc.points) { float v[3] = {p.x, p.y, p.z}; wr(o, v, 3); }
        // 7. the reference's AddVertex(Twb, depth) twice (FrameKDMap.cpp:34-51): the second frame's edge cloud is
        //    back-projected with the FIRST frame's Twb * Tbc (:209); then the Edge-KD-tree query of PlanWapionts
        const double Twb2[16] = {1, 0, 0, 2.5, 0, 1, 0, -1.0, 0, 0, 1, 1.5, 0, 0, 0, 1};
        dmap.AddVertex(Twb, img.data(), AMK_DEPTH_U16, rows, cols);
        dmap.AddVertex(Twb2, img.data(), AMK_DEPTH_U16, rows, cols);
        int ecnt = (int)dmap.CurFrame().edgeCloud->GetPointCloud().pts.size();
        wr(o, &ecnt, 1);
        std::vector<Vector3d> epts; std::vector<double> ed2;
        dmap.QueryNearest(Vector3d(3.0, -1.0, 1.5), 1, epts, ed2, true);
        int ec = (int)ed2.size();
        wr(o, &ec, 1); wr(o, ed2.data(), ec);
        // 8. PtIsInFrame (FrameKDMap.cpp:215-231) with the current frame's Twc and the down-scaled camera model
        for (int i = 0; i < nq; ++i) {
            int in = dmap.PtIsInFrame(Vector3d(qs[3 * i], qs[3 * i + 1], qs[3 * i + 2]), dmap.CurTwc()) ? 1 : 0;
            wr(o, &in, 1);
        }
    }
    fclose(o);
    return 0;
}
