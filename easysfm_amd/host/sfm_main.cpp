// ./bin/sfm_native -- the reference executable's command line (cpp_code/test/sfm.cpp:32-50, cpp_code/script/run_fountain_small.sh:22-24)
// as a native C++ program over the C ABI of libesfm_hip.so, through the host mirror of the reference's classes
// (esfm_host.hpp).  Same thirteen positional arguments in the same order, same ASCII .ply of PointXYZRGB at argv[5], exit
// status 1 on success like the reference (sfm.cpp:339).  The control flow is the one of
// easysfm_amd/pipeline.py (the Python twin of this file); the (i, j < i) pair loop of sfm.cpp:140-170 is ONE batched call per stage
// (every frame's descriptors uploaded once, one match launch sequence, one RANSAC / pose / depth batch -- INTEGRATION.md section 2;
// ESFM_PAIR_BY_PAIR=1 in the environment runs it pair by pair through host pointers as the reference does, same results):
//   import -> undistort -> SURF or ORB -> all-pairs match + 5-point RANSAC + depth -> track ids -> initial pair -> triangulate -> BA
//   -> (next frame by PnP -> triangulate against every registered frame -> periodic BA)* -> final BA -> SOR -> .ply
// Differences from the reference: ORB's intensity tests use this library's own point pairs (cv::ORB's learned table ships only
// inside OpenCV); the viewer arguments are accepted and ignored (no display); PNG and baseline JPEG images are read.
//   bin/sfm_native --dump-image in.png out.raw   writes rows, cols (int32) and the BGR bytes: the decoder's test hook
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <atomic>
#include <string>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "esfm_host.hpp"

using namespace p3dv;

namespace {

struct frame_pair_t {  // utility.h:57-78
    std::vector<DMatch> matches;
    Matrix4f T_21 = Matrix4f::Identity();
    double appro_depth = 1.0;
};

Matrix4f mul(const Matrix4f &a, const Matrix4f &b)
{
    Matrix4f c;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = 0.f;
            for (int k = 0; k < 4; ++k) s += a(i, k) * b(k, j);
            c(i, j) = s;
        }
    return c;
}

// wall-clock of the stages the reference times with its `... cost = ... seconds` lines (feature_matching.cpp:141, ba.cpp:285), one
// summary line at the end (`stage seconds: ...`) for bench.py's config-1 / config-3 legs
struct StageClock {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double lap()
    {
        const auto t1 = std::chrono::steady_clock::now();
        const double s = std::chrono::duration<double>(t1 - t0).count();
        t0 = t1;
        return s;
    }
};

int dump_image(const char *in, const char *out)
{
    int rows = 0, cols = 0;
    std::vector<uint8_t> bgr;
    const std::string err = read_image_bgr(in, rows, cols, bgr);
    if (!err.empty()) { std::cerr << err << std::endl; return 3; }
    std::ofstream f(out, std::ios::binary);
    const int32_t hdr[2] = {rows, cols};
    f.write(reinterpret_cast<const char *>(hdr), sizeof(hdr));
    f.write(reinterpret_cast<const char *>(bgr.data()), std::streamsize(bgr.size()));
    return f ? 0 : 3;
}

}  // namespace

int main(int argc, char **argv)
{
    if (argc == 4 && std::string(argv[1]) == "--dump-image") return dump_image(argv[2], argv[3]);
    if (argc != 14) {
        std::cerr << "usage: sfm_native image_folder image_list calib_K_file calib_distort_file output.ply feature_type(S) feature_parameter "
                     "repro_dis_ransac find_init_frames ba_calib_change_tolerance ba_frequency launch_viewer view_sphere" << std::endl;
        return 2;
    }
    const std::string image_data_path = argv[1], image_list_path = argv[2], calib_file_path = argv[3], distort_file_path = argv[4],
                      output_file_path = argv[5];
    char using_feature = argv[6][0];
    const int feature_extract_parameter = std::atoi(argv[7]);
    const double ransac_reproj_distance = std::atof(argv[8]);
    const bool use_track_frames_as_init = std::atoi(argv[9]) != 0;
    const double fix_calib_tolerance_BA = std::atof(argv[10]);
    const int frequency_BA = std::max(1, std::atoi(argv[11]));
    if (using_feature != 'S' && using_feature != 'O') { std::cout << "Wrong feature input. Use SURF as default feature." << std::endl; using_feature = 'S'; }

    try {
        DataIO io;
        FeatureMatching fm;
        MotionEstimator ee;
        std::vector<frame_t> frames;
        double t_import = 0, t_detect = 0, t_match = 0, t_verify = 0, t_tracks = 0, t_ba = 0, t_register = 0, t_sor = 0, t_pnp = 0, t_next = 0;
        int n_ba = 0;
        StageClock total_clock, clk;
        if (!io.importImageFilenames(image_list_path, image_data_path, frames) || frames.size() < 2) { std::cerr << "need at least two images" << std::endl; return 3; }
        const int frame_number = int(frames.size());
        Matrix3f K_mat = Matrix3f::Identity();
        if (!io.importCalib(calib_file_path, K_mat)) { std::cerr << "cannot read the calibration file " << calib_file_path << std::endl; return 3; }
        DistortMat distort_coeff;
        if (!io.importDistort(distort_file_path, distort_coeff)) std::cout << "No distortion coefficients imported. Use defualt one (0)." << std::endl;

        // The image files are decoded by a few host threads running ahead of the loop below (a 768 x 512 PNG is ~10 ms of inflate; the
        // first frame's import otherwise also waits for the GPU's first-touch initialisation): frame i is taken when its decode is done,
        // messages and failures appear where the reference's sequential loop has them.
        std::vector<std::string> decode_err((size_t)frame_number);
        std::vector<char> decoded((size_t)frame_number, 0);
        std::mutex decode_mu;
        std::condition_variable decode_cv;
        std::atomic<int> next_decode{0};
        std::vector<std::thread> decoders;
        {
            const int n_thr = std::max(1, std::min<int>({frame_number, 8, (int)std::thread::hardware_concurrency()}));
            for (int w = 0; w < n_thr; ++w)
                decoders.emplace_back([&] {
                    for (int i = next_decode++; i < frame_number; i = next_decode++) {
                        ImageMat &img = frames[size_t(i)].rgb_image;
                        img.channels = 3;
                        std::string err;
                        // a malformed or huge header must end as this frame's "No more images", not as std::terminate in a worker
                        try { err = read_image_bgr(frames[size_t(i)].image_file_path, img.rows, img.cols, img.data); }
                        catch (const std::exception &e) { err = std::string("decoder failed: ") + e.what(); }
                        catch (...) { err = "decoder failed"; }
                        {
                            std::lock_guard<std::mutex> lk(decode_mu);
                            decode_err[size_t(i)] = err;
                            decoded[size_t(i)] = 1;
                        }
                        decode_cv.notify_all();
                    }
                });
        }
        struct Joiner { std::vector<std::thread> &t; ~Joiner() { for (auto &x : t) if (x.joinable()) x.join(); } } joiner{decoders};
        auto import_decoded = [&](int i) {
            {
                std::unique_lock<std::mutex> lk(decode_mu);
                decode_cv.wait(lk, [&] { return decoded[size_t(i)] != 0; });
            }
            if (!decode_err[size_t(i)].empty()) {          // (DataIO::importImages' message)
                frames[size_t(i)].rgb_image = ImageMat();
                std::cout << "No more images" << " (" << decode_err[size_t(i)] << ")" << std::endl;
                return false;
            }
            return true;
        };

        // ---- per frame: import, undistort, SURF (sfm.cpp:84-126)
        std::cout << "Begin feature extraction" << std::endl;
        const bool frame_trace = std::getenv("ESFM_FRAME_TRACE") != nullptr;       // per-frame milliseconds on stderr (diagnosis of first-touch costs)
        for (int i = 0; i < frame_number; ++i) {
            clk.lap();
            if (!import_decoded(i)) return 3;
            const double t_wait = clk.lap();
            frames[size_t(i)].K_cam = K_mat;
            if (!ee.doUnDistort(frames[size_t(i)], distort_coeff)) return 3;
            const double t_und = clk.lap();
            t_import += t_wait + t_und;
            std::cout << "Feature extraction of Frame [ " << i << " ]" << std::endl;
            if (using_feature == 'O' ? !fm.detectFeaturesORB(frames[size_t(i)], feature_extract_parameter)          // sfm.cpp:112-117
                                     : !fm.detectFeaturesSURF(frames[size_t(i)], feature_extract_parameter)) return 3;
            frames[size_t(i)].init_pixel_ids();
            const double t_det = clk.lap();
            t_detect += t_det;
            if (frame_trace)
                std::cerr << "[frame " << i << "] decode wait " << t_wait * 1e3 << " ms, undistort " << t_und * 1e3 << " ms, detect " << t_det * 1e3 << " ms" << std::endl;
        }
        std::cout << "Feature extraction done" << std::endl;

        // ---- all pairs (i, j < i): match, verify, relative depth (sfm.cpp:140-167)
        const int num_min_pair = 20;
        const size_t nf = size_t(frame_number);
        std::vector<std::vector<frame_pair_t>> graph(nf, std::vector<frame_pair_t>(nf));
        const bool pair_by_pair = std::getenv("ESFM_PAIR_BY_PAIR") != nullptr && std::atoi(std::getenv("ESFM_PAIR_BY_PAIR")) != 0;
        if (pair_by_pair) {
            for (int i = 0; i < frame_number; ++i)
                for (int j = 0; j < i; ++j) {
                    frame_pair_t &g = graph[size_t(i)][size_t(j)];
                    std::vector<DMatch> temp_matches, inlier_matches;
                    StageClock pc;
                    if (using_feature == 'O') fm.matchFeaturesORB(frames[size_t(i)], frames[size_t(j)], temp_matches);   // sfm.cpp:153-160
                    else fm.matchFeaturesSURF(frames[size_t(i)], frames[size_t(j)], temp_matches);
                    t_match += pc.lap();
                    if (int(temp_matches.size()) > num_min_pair) {
                        Matrix4f T = Matrix4f::Identity();
                        if (ee.estimate2D2D_E5P_RANSAC(frames[size_t(i)], frames[size_t(j)], temp_matches, inlier_matches, T, ransac_reproj_distance)) {
                            g.T_21 = T;
                            double depth = 1.0;
                            if (ee.getDepthFast(frames[size_t(i)], frames[size_t(j)], T, inlier_matches, depth)) g.appro_depth = depth;
                            g.matches.swap(inlier_matches);
                            if (!g.matches.empty()) std::cout << "Pair ( " << i << " , " << j << " ): [" << g.matches.size() << "] verified matches." << std::endl;
                        }
                    }
                    t_verify += pc.lap();
                }
        } else {
            // the same loop, batched: one match call for the whole (i, j < i) list, one RANSAC + pose + depth batch for the pairs with
            // more than num_min_pair matches (none of them depends on another pair's result)
            StageClock pc;
            std::vector<std::pair<int, int>> pairs;
            for (int i = 0; i < frame_number; ++i) for (int j = 0; j < i; ++j) pairs.emplace_back(i, j);
            std::vector<std::vector<DMatch>> temp_matches;
            if (!fm.matchFeaturesAllPairs(frames, pairs, using_feature == 'O', temp_matches)) return 3;
            t_match += pc.lap();
            std::vector<std::pair<int, int>> jobs;
            std::vector<std::vector<DMatch>> job_matches;
            for (size_t p = 0; p < pairs.size(); ++p)
                if (int(temp_matches[p].size()) > num_min_pair) { jobs.push_back(pairs[p]); job_matches.push_back(std::move(temp_matches[p])); }
            std::vector<std::vector<DMatch>> inliers;
            std::vector<Matrix4f> Ts;
            std::vector<double> depths;
            std::vector<char> ok;
            if (!ee.estimate2D2D_E5P_RANSAC_pairs(frames, jobs, job_matches, inliers, Ts, depths, ok, ransac_reproj_distance)) {
                // A batch entry point fails as a whole when ONE pair is unusable (a non-finite essential matrix, say); the reference's
                // loop -- and ESFM_PAIR_BY_PAIR=1 above -- only loses that pair (estimate_motion.cpp:27-97 returns false, sfm.cpp:165
                // carries on).  Same here: the batch's pairs are verified one by one, a failing pair is skipped.
                std::cout << "batched verification failed (" << esfm_last_error() << "): verifying pair by pair" << std::endl;
                inliers.assign(jobs.size(), {}); Ts.assign(jobs.size(), Matrix4f::Identity()); depths.assign(jobs.size(), 1.0); ok.assign(jobs.size(), 0);
                for (size_t p = 0; p < jobs.size(); ++p) {
                    frame_t &fi = frames[size_t(jobs[p].first)], &fj = frames[size_t(jobs[p].second)];
                    Matrix4f T = Matrix4f::Identity();
                    if (!ee.estimate2D2D_E5P_RANSAC(fi, fj, job_matches[p], inliers[p], T, ransac_reproj_distance)) continue;
                    double depth = 1.0;
                    Ts[p] = T;
                    if (ee.getDepthFast(fi, fj, T, inliers[p], depth)) depths[p] = depth;
                    ok[p] = 1;
                }
            }
            for (size_t p = 0; p < jobs.size(); ++p) {
                if (!ok[p]) continue;
                frame_pair_t &g = graph[size_t(jobs[p].first)][size_t(jobs[p].second)];
                g.T_21 = Ts[p];
                g.appro_depth = depths[p];
                g.matches.swap(inliers[p]);
                if (!g.matches.empty()) std::cout << "Pair ( " << jobs[p].first << " , " << jobs[p].second << " ): [" << g.matches.size() << "] verified matches." << std::endl;
            }
            t_verify += pc.lap();
        }

        // ---- track ids (sfm.cpp:173-216): a keypoint takes the id of its verified match in an earlier frame unless the frame
        // already uses that id; the rest get fresh ids
        clk.lap();
        size_t total_kp = 0;
        for (const frame_t &f : frames) total_kp += f.keypoints.size();
        std::vector<std::vector<bool>> track(nf, std::vector<bool>(std::max<size_t>(total_kp, 1), false));
        int cur_id = 0;
        std::vector<char> id_in_frame(std::max<size_t>(total_kp, 1), 0);     // which ids the current frame's keypoints hold (ids < total_kp)
        for (int i = 0; i < frame_number; ++i) {
            frame_t &fi = frames[size_t(i)];
            // the reference walks the frame's whole id list for every match ("is this id used already?", sfm.cpp:190-199):
            // O(matches x keypoints) per frame.  The same question from a flag per id that follows every assignment.
            std::vector<int> held;
            for (int j = 0; j < i; ++j) {
                const frame_t &fj = frames[size_t(j)];
                for (const DMatch &m : graph[size_t(i)][size_t(j)].matches) {
                    const int tid = fj.unique_pixel_ids[size_t(m.trainIdx)];
                    int &mine = fi.unique_pixel_ids[size_t(m.queryIdx)];
                    if (mine >= 0 && mine == tid) continue;
                    if (!id_in_frame[size_t(tid)]) {
                        if (mine >= 0) id_in_frame[size_t(mine)] = 0;      // (this keypoint was the only holder of its old id)
                        mine = tid; id_in_frame[size_t(tid)] = 1; held.push_back(tid);
                        fi.unique_pixel_has_match[size_t(m.queryIdx)] = true;
                    }
                }
            }
            for (int v : held) id_in_frame[size_t(v)] = 0;
            int fresh = 0;
            for (size_t k = 0; k < fi.unique_pixel_ids.size(); ++k) {
                if (fi.unique_pixel_ids[k] < 0) fi.unique_pixel_ids[k] = cur_id + fresh++;
                track[size_t(i)][size_t(fi.unique_pixel_ids[k])] = true;
            }
            cur_id += fresh;
        }
        std::cout << "The total unique feature point number is " << cur_id << std::endl;
        t_tracks += clk.lap();

        // ---- initial pair (sfm.cpp:218-247)
        int init_1 = 1, init_2 = 0;
        double depth_init = 10.0;
        if (use_track_frames_as_init) {
            std::vector<std::vector<double>> depth(nf, std::vector<double>(nf, 0.0));
            for (int i = 0; i < frame_number; ++i) for (int j = 0; j < i; ++j) depth[size_t(i)][size_t(j)] = graph[size_t(i)][size_t(j)].appro_depth;
            fm.findInitializeFramePair(track, frames, depth, init_1, init_2, depth_init);
        }
        std::cout << "Initialization frames: [ " << init_1 << " ] and [ " << init_2 << " ]" << std::endl;
        pointcloud_sparse_t cloud;
        frames[size_t(init_1)].pose_cam = Matrix4f::Identity();
        frames[size_t(init_2)].pose_cam = mul(graph[size_t(init_1)][size_t(init_2)].T_21, frames[size_t(init_1)].pose_cam);
        ee.doTriangulation(frames[size_t(init_1)], frames[size_t(init_2)], graph[size_t(init_1)][size_t(init_2)].matches, cloud);
        std::vector<bool> todo(nf, true);
        todo[size_t(init_1)] = todo[size_t(init_2)] = false;
        BundleAdjustment ba;
        t_register += clk.lap();
        ba.doSFMBA(frames, todo, cloud, fix_calib_tolerance_BA);
        t_ba += clk.lap(); ++n_ba;

        // ---- register the remaining frames (sfm.cpp:262-321)
        int remaining = frame_number - 2;
        double t_next_prev = 0;
        double reproj = ransac_reproj_distance;
        while (remaining > 0) {
            int nxt = -1;
            fm.findNextFrame(track, todo, cloud.unique_point_ids, nxt);
            if (nxt < 0) break;                      // no unregistered frame sees the map (the reference would index with an uninitialised value)
            t_next += clk.lap(); t_register += t_next - t_next_prev; t_next_prev = t_next;
            const bool ok = ee.estimate2D3D_P3P_RANSAC(frames[size_t(nxt)], cloud, reproj);
            { const double dt = clk.lap(); t_pnp += dt; t_register += dt; }
            reproj += 1.0;
            for (int i = 0; i < frame_number; ++i) {
                if (todo[size_t(i)]) continue;
                if (nxt > i) ee.doTriangulation(frames[size_t(nxt)], frames[size_t(i)], graph[size_t(nxt)][size_t(i)].matches, cloud);
                else ee.doTriangulation(frames[size_t(i)], frames[size_t(nxt)], graph[size_t(i)][size_t(nxt)].matches, cloud);
            }
            if (!ok) ee.outlierFilter(cloud);
            todo[size_t(nxt)] = false;
            --remaining;
            t_register += clk.lap();
            if (remaining % frequency_BA == 0) {
                ba.initBA();
                ba.doSFMBA(frames, todo, cloud, fix_calib_tolerance_BA);
                reproj = ransac_reproj_distance;
                t_ba += clk.lap(); ++n_ba;
            }
            std::cout << "Progress: [ " << frame_number - remaining << " / " << frame_number << " ]" << std::endl;
        }
        clk.lap();
        ba.doSFMBA(frames, todo, cloud);
        t_ba += clk.lap(); ++n_ba;

        // ---- final cloud: SOR filter, .ply (sfm.cpp:329-337)
        std::vector<PointXYZRGB> filtered;
        CProceesing<PointXYZRGB> cp;
        if (!cp.SORFilter(cloud.points, filtered)) return 3;
        const std::filesystem::path out_dir = std::filesystem::path(output_file_path).parent_path();
        if (!out_dir.empty()) std::filesystem::create_directories(out_dir);
        t_sor += clk.lap();
        if (!io.writePlyFile(output_file_path, filtered)) return 3;
        std::cout << "stage seconds: import+undistort " << t_import << " detect " << t_detect << " match " << t_match << " verify " << t_verify << " tracks " << t_tracks
                  << " register " << t_register << " register_next_frame " << t_next << " register_pnp " << t_pnp << " ba " << t_ba << " ba_calls " << n_ba << " sor " << t_sor << " total " << total_clock.lap()
                  << " frames " << frame_number << " pairs " << frame_number * (frame_number - 1) / 2 << " batched " << (pair_by_pair ? 0 : 1) << std::endl;
    } catch (const std::exception &e) {       // 1 is the reference's SUCCESS status: a failure must not look like one
        std::cerr << "sfm_native: " << e.what() << std::endl;
        return 3;
    }
    return 1;
}
