// C++ host mirror of the reference's hot-path interface over the C ABI (include/esfm.h).
//
// Same class and member names, argument meaning, defaults and error behaviour as
//   p3dv::FeatureMatching::matchFeaturesORB / matchFeaturesSURF   cpp_code/include/feature_matching.h:17-21
//   p3dv::BundleAdjustment::{initBA,setBAProblem,solveBA,doSFMBA}  cpp_code/include/ba.h:59-84
//   p3dv::MotionEstimator::{estimate2D2D_E5P_RANSAC,getDepthFast,doTriangulation,estimate2D3D_P3P_RANSAC,outlierFilter}
//                                                                  cpp_code/include/estimate_motion.h:17-35
// with the OpenCV / PCL / Eigen value types of cpp_code/include/utility.h:21-102 replaced by the
// plain structs below (the image has none of those libraries).  Header-only; link libesfm_hip.so.
// All compute happens on the GPU behind the C ABI; there is no CPU fallback here.
#pragma once

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <memory>
#include <vector>

#include "../../include/esfm.h"
#include "esfm_jpeg.hpp"   // (pulls in esfm_png.hpp and read_image_bgr)

namespace p3dv {

// ---- value types (utility.h) ------------------------------------------------------------------
struct DMatch {  // cv::DMatch
    int queryIdx = -1, trainIdx = -1, imgIdx = 0;
    float distance = 0.f;
    DMatch() = default;
    DMatch(int q, int t, int img, float d) : queryIdx(q), trainIdx(t), imgIdx(img), distance(d) {}
};

struct Point2f { float x = 0.f, y = 0.f; };
struct KeyPoint {  // cv::KeyPoint (only .pt is read downstream, ba.cpp:37)
    Point2f pt;
    float size = 0.f, angle = -1.f, response = 0.f;
    int octave = 0, class_id = -1;
};

// cv::Mat restricted to what descriptors need: CV_32F (SURF, N x 64) or CV_8U (ORB, N x 32), continuous
struct DescMat {
    enum Type { F32 = 0, U8 = 1 };
    int rows = 0, cols = 0;
    Type type = F32;
    std::vector<uint8_t> bytes;
    void create(int r, int c, Type t) { rows = r; cols = c; type = t; bytes.assign(size_t(r) * c * (t == F32 ? 4 : 1), 0); }
    template <class T> T *ptr(int r = 0) { return reinterpret_cast<T *>(bytes.data()) + size_t(r) * cols; }
    template <class T> const T *ptr(int r = 0) const { return reinterpret_cast<const T *>(bytes.data()) + size_t(r) * cols; }
};

template <int R, int C> struct Matf {  // Eigen::Matrix<float,R,C>, (r,c) access
    float v[R * C];
    Matf() { for (auto &x : v) x = 0.f; }
    float &operator()(int r, int c) { return v[r * C + c]; }
    float operator()(int r, int c) const { return v[r * C + c]; }
    static Matf Identity() { Matf m; for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = 1.f; return m; }
};
using Matrix3f = Matf<3, 3>;
using Matrix4f = Matf<4, 4>;

struct ImageMat {  // cv::Mat, CV_8UC1 / CV_8UC3 (BGR, interleaved), row-major and continuous
    int rows = 0, cols = 0, channels = 3;
    std::vector<uint8_t> data;
    bool empty() const { return data.empty(); }
};
struct DistortMat {  // the 1 x 4 CV_64FC1 matrix sfm.cpp:78 creates: zeros
    double v[4] = {0, 0, 0, 0};
};

struct frame_t {  // utility.h:21-55
    unsigned int frame_id = 0;
    std::string image_file_path;
    ImageMat rgb_image;
    std::vector<KeyPoint> keypoints;
    DescMat descriptors;
    std::vector<int> unique_pixel_ids;
    std::vector<bool> unique_pixel_has_match;
    Matrix4f pose_cam = Matrix4f::Identity();
    Matrix3f K_cam = Matrix3f::Identity();
    frame_t() = default;
    frame_t(unsigned int id, const std::string &path) : frame_id(id), image_file_path(path) {}
    bool init_pixel_ids()
    {
        unique_pixel_ids.assign(keypoints.size(), -1);
        unique_pixel_has_match.assign(keypoints.size(), false);
        return true;
    }
};

struct PointXYZRGB { float x = 0, y = 0, z = 0; uint8_t r = 0, g = 0, b = 0; };  // pcl::PointXYZRGB

struct pointcloud_sparse_t {  // utility.h:88-102 (rgb_pointcloud->points flattened)
    std::vector<PointXYZRGB> points;
    std::vector<int> unique_point_ids;
    std::vector<int> is_inlier;
};

// ---- library context ----------------------------------------------------------------------------
class EsfmError : public std::runtime_error {
public:
    int status;
    EsfmError(int s, const std::string &m) : std::runtime_error(m), status(s) {}
};

inline esfm_ctx *default_ctx()
{
    static thread_local esfm_ctx *ctx = nullptr;
    if (!ctx) {
        int rc = esfm_ctx_create(0, nullptr, &ctx);
        if (rc != ESFM_OK) throw EsfmError(rc, esfm_last_error());
    }
    return ctx;
}

// ---- matching (feature_matching.h:17-21) ----------------------------------------------------------
class FeatureMatching {
public:
    // feature_matching.cpp:43-69: SURF::create(minHessian)->detect + SURF::create()->compute on cur_frame.rgb_image (BGR)
    bool detectFeaturesSURF(frame_t &cur_frame, int minHessian = 400, bool show = false)
    {
        (void)show;
        const ImageMat &img = cur_frame.rgb_image;
        if (img.empty()) { std::cerr << "frame has no image" << std::endl; return false; }
        // room for every possible maximum (a quarter of the pixels), NOT zero-filled (the library writes the n rows it returns and
        // nothing else: a value-initialised std::vector of these 28 MB cost 7 - 8 ms per 768 x 512 image in page faults) and KEPT
        // between frames: the device-to-host copies land here, the runtime pins what it copies into, and unmapping pinned pages at the
        // end of every call showed up as 15 - 35 ms stalls in the next frames' calls
        const int cap = img.rows * img.cols / 4 + 1024;
        float *kp = scratch_f32(kp_buf_, kp_cap_, size_t(7) * size_t(cap)), *desc = scratch_f32(desc_buf_, desc_cap_, size_t(64) * size_t(cap));
        int32_t n = 0;
        if (esfm_surf_detect_and_compute(default_ctx(), img.data.data(), img.rows, img.cols, img.channels, double(minHessian), cap, kp, desc, &n) != ESFM_OK) {
            std::cerr << esfm_last_error() << std::endl;
            return false;
        }
        cur_frame.keypoints.resize(size_t(n));
        for (int k = 0; k < n; ++k) {
            KeyPoint &q = cur_frame.keypoints[size_t(k)];
            const float *v = &kp[size_t(7) * size_t(k)];
            q.pt.x = v[0]; q.pt.y = v[1]; q.size = v[2]; q.angle = v[3]; q.response = v[4]; q.octave = int(v[5]); q.class_id = int(v[6]);
        }
        cur_frame.descriptors.create(n, 64, DescMat::F32);
        if (n) std::memcpy(cur_frame.descriptors.ptr<float>(), desc, sizeof(float) * size_t(64) * size_t(n));
        if (!quiet) std::cout << "Found " << n << " features." << std::endl;
        return true;
    }

    // feature_matching.cpp:14-41: ORB::create(max_num)->detect + ->compute on cur_frame.rgb_image (BGR)
    bool detectFeaturesORB(frame_t &cur_frame, int max_num = 5000, bool show = false)
    {
        (void)show;
        const ImageMat &img = cur_frame.rgb_image;
        if (img.empty()) { std::cerr << "frame has no image" << std::endl; return false; }
        const int cap = 2 * max_num + 4096;       // retainBest keeps ties
        float *kp = scratch_f32(kp_buf_, kp_cap_, size_t(7) * size_t(cap));                       // (kept between frames: see detectFeaturesSURF)
        uint8_t *desc = reinterpret_cast<uint8_t *>(scratch_f32(desc_buf_, desc_cap_, size_t(8) * size_t(cap)));
        int32_t n = 0;
        if (esfm_orb_detect_and_compute(default_ctx(), img.data.data(), img.rows, img.cols, img.channels, max_num, cap, kp, desc, &n) != ESFM_OK) {
            std::cerr << esfm_last_error() << std::endl;
            return false;
        }
        cur_frame.keypoints.resize(size_t(n));
        for (int k = 0; k < n; ++k) {
            KeyPoint &q = cur_frame.keypoints[size_t(k)];
            const float *v = &kp[size_t(7) * size_t(k)];
            q.pt.x = v[0]; q.pt.y = v[1]; q.size = v[2]; q.angle = v[3]; q.response = v[4]; q.octave = int(v[5]); q.class_id = int(v[6]);
        }
        cur_frame.descriptors.create(n, 32, DescMat::U8);
        if (n) std::memcpy(cur_frame.descriptors.ptr<uint8_t>(), desc, size_t(32) * size_t(n));
        if (!quiet) std::cout << "Found " << n << " features" << std::endl;
        return true;
    }

    bool matchFeaturesORB(frame_t &cur_frame_1, frame_t &cur_frame_2, std::vector<DMatch> &matches, double ratio_thre = 0.8,
                          bool show = false)
    {
        return run(cur_frame_1, cur_frame_2, matches, ratio_thre, true, "ORB");
    }

    bool matchFeaturesSURF(frame_t &cur_frame_1, frame_t &cur_frame_2, std::vector<DMatch> &matches, double ratio_thre = 0.5,
                           bool show = false)
    {
        return run(cur_frame_1, cur_frame_2, matches, ratio_thre, false, "SURF");
    }

    // Which two frames start the reconstruction (the contract of feature_matching.cpp:160-229, defaults feature_matching.h:26): every
    // track weighs as many frames as see it; a pair (i, j < i) scores the summed weight of the tracks BOTH frames see; pairs whose
    // depth / baseline ratio exceeds the limit are out; the best score of at least min_track_num_init wins, the LATER pair on ties.
    // appro_depth[i][j] = img_match_graph[i][j].appro_depth.  Written over per-frame track lists: a frame's weights are laid out once
    // and every earlier frame sums over its own list -- O(frames x observations) where the reference walks frames^2 x tracks.
    bool findInitializeFramePair(std::vector<std::vector<bool>> &feature_track_matrix, std::vector<frame_t> &frames,
                                 const std::vector<std::vector<double>> &appro_depth, int &initialization_frame_1,
                                 int &initialization_frame_2, double &depth_init, int min_track_num_init = 100,
                                 double max_depth_baseline_ratio_init = 50.0)
    {
        const size_t n_frames = frames.size(), n_tracks = feature_track_matrix.empty() ? 0 : feature_track_matrix[0].size();
        std::vector<int> weight(n_tracks, 0);
        std::vector<std::vector<int>> seen(n_frames);            // track ids per frame
        for (size_t f = 0; f < n_frames; ++f) {
            feature_track_matrix[f].resize(n_tracks);
            for (size_t t = 0; t < n_tracks; ++t)
                if (feature_track_matrix[f][t]) { ++weight[t]; seen[f].push_back(int(t)); }
        }
        long long best = min_track_num_init;
        int first = 0, second = 0;
        double best_ratio = 0.0;
        std::vector<int> laid(n_tracks, 0);                        // the weights of the tracks frame i sees, 0 elsewhere
        for (size_t i = 0; i < n_frames; ++i) {
            for (int t : seen[i]) laid[size_t(t)] = weight[size_t(t)];
            for (size_t j = 0; j < i; ++j) {
                const double ratio = appro_depth[i][j];
                if (ratio > max_depth_baseline_ratio_init) continue;
                long long score = 0;
                for (int t : seen[j]) score += laid[size_t(t)];
                if (score >= best) { best = score; best_ratio = ratio; first = int(i); second = int(j); }
            }
            for (int t : seen[i]) laid[size_t(t)] = 0;
        }
        if (first == second) {
            if (!quiet) std::cout << "Failed to find proper frame pair for initialization. Use default frame [1] and frame [0]" << std::endl;
            initialization_frame_1 = 1; initialization_frame_2 = 0;
            return false;                                           // depth_init keeps the caller's value, as in the reference (:217-223)
        }
        initialization_frame_1 = first; initialization_frame_2 = second;
        depth_init = best_ratio;
        return true;
    }

    // The frame to register next (the contract of feature_matching.cpp:231-268): among the frames still to process the one that sees
    // most of the tracks that already have a 3-D point; the FIRST such frame on ties; `next_frame` untouched when none sees any.
    bool findNextFrame(std::vector<std::vector<bool>> &feature_track_matrix, std::vector<bool> &frames_to_process,
                       std::vector<int> &unique_3d_point_ids, int &next_frame)
    {
        int most = 0;
        for (size_t f = 0; f < frames_to_process.size(); ++f) {
            if (!frames_to_process[f]) continue;
            const std::vector<bool> &row = feature_track_matrix[f];
            const int shared = int(std::count_if(unique_3d_point_ids.begin(), unique_3d_point_ids.end(), [&](int id) { return bool(row[size_t(id)]); }));
            if (shared > most) { most = shared; next_frame = int(f); }
        }
        return true;
    }

    bool quiet = false;  // the reference prints timing lines (feature_matching.cpp:96-97,141-142)

    // The pair loop of test/sfm.cpp:140-161 in ONE call: matchFeatures{SURF,ORB}(frames[pairs[p].first], frames[pairs[p].second],
    // matches[p]) for every listed pair -- every frame's descriptors uploaded once, one launch sequence for the whole list
    // (esfm_match_pairs), the per-pair lines of feature_matching.cpp:96-97 / :141-142 printed with the call's time shared out
    // evenly.  matches[p] is appended to, like the single-pair members.
    bool matchFeaturesAllPairs(std::vector<frame_t> &frames, const std::vector<std::pair<int, int>> &pairs, bool hamming,
                               std::vector<std::vector<DMatch>> &matches, double ratio_thre = -1.0)
    {
        if (ratio_thre < 0) ratio_thre = hamming ? 0.8 : 0.5;               // the members' defaults (feature_matching.h:17-21)
        matches.resize(pairs.size());
        if (pairs.empty()) return true;
        int width = 0;
        for (const frame_t &f : frames) if (f.descriptors.rows > 0) { width = f.descriptors.cols; break; }
        if (width == 0) return true;                                       // no frame has a descriptor: nothing can match
        const size_t row_bytes = hamming ? size_t(width) : size_t(width) * 4;
        std::vector<int32_t> off(frames.size() + 1, 0);
        for (size_t i = 0; i < frames.size(); ++i) {
            const DescMat &d = frames[i].descriptors;
            if (d.rows > 0 && (d.cols != width || d.type != (hamming ? DescMat::U8 : DescMat::F32))) { std::cerr << "descriptor shapes differ\n"; return false; }
            off[i + 1] = off[i] + d.rows;
        }
        std::vector<uint8_t> bank(size_t(off.back()) * row_bytes + 16);
        for (size_t i = 0; i < frames.size(); ++i)
            if (frames[i].descriptors.rows > 0) std::memcpy(bank.data() + size_t(off[i]) * row_bytes, frames[i].descriptors.bytes.data(), size_t(frames[i].descriptors.rows) * row_bytes);
        std::vector<int32_t> pl(2 * pairs.size());
        size_t total = 0;
        for (size_t p = 0; p < pairs.size(); ++p) { pl[2 * p] = pairs[p].first; pl[2 * p + 1] = pairs[p].second; total += size_t(frames[size_t(pairs[p].first)].descriptors.rows); }
        std::vector<int32_t> qi(std::max<size_t>(total, 1)), ti(std::max<size_t>(total, 1)), n_out(pairs.size());
        std::vector<float> d(std::max<size_t>(total, 1));
        std::vector<int64_t> out_off(pairs.size() + 1);
        auto tic = std::chrono::steady_clock::now();
        const int rc = esfm_match_pairs(default_ctx(), hamming ? ESFM_HAMMING : ESFM_L2_F32, bank.data(), off.data(), int(frames.size()), width, pl.data(),
                                        int(pairs.size()), ratio_thre, qi.data(), ti.data(), d.data(), n_out.data(), out_off.data());
        if (rc != ESFM_OK) { std::cerr << esfm_last_error() << std::endl; return false; }
        const std::chrono::duration<double> dt = std::chrono::steady_clock::now() - tic;
        for (size_t p = 0; p < pairs.size(); ++p) {
            const size_t o = size_t(out_off[p]);
            for (int k = 0; k < n_out[p]; ++k) matches[p].push_back(DMatch(qi[o + size_t(k)], ti[o + size_t(k)], 0, d[o + size_t(k)]));
            if (!quiet) {
                std::cout << "match " << (hamming ? "ORB" : "SURF") << " cost = " << dt.count() / double(pairs.size()) << " seconds. " << std::endl;
                std::cout << "# Correspondence: Initial [ " << frames[size_t(pairs[p].first)].descriptors.rows << " ]  Filtered by Lowe ratio test [ " << n_out[p]
                          << " ]" << std::endl;
            }
        }
        return true;
    }

private:
    // detection outputs' landing buffers, uninitialised, grown on demand, kept for the object's life
    std::unique_ptr<float[]> kp_buf_, desc_buf_;
    size_t kp_cap_ = 0, desc_cap_ = 0;
    static float *scratch_f32(std::unique_ptr<float[]> &buf, size_t &cap, size_t n)
    {
        if (n > cap) { buf.reset(new float[n]); cap = n; }
        return buf.get();
    }

    bool run(frame_t &f1, frame_t &f2, std::vector<DMatch> &matches, double ratio, bool hamming, const char *tag)
    {
        const DescMat &q = f1.descriptors, &t = f2.descriptors;
        if (q.rows > 0 && t.rows > 0 && (q.cols != t.cols || q.type != t.type)) { std::cerr << "descriptor shapes differ\n"; return false; }
        if ((hamming && q.type != DescMat::U8) || (!hamming && q.type != DescMat::F32)) { std::cerr << "wrong descriptor type\n"; return false; }
        auto tic = std::chrono::steady_clock::now();
        std::vector<int32_t> qi(size_t(std::max(q.rows, 1))), ti(size_t(std::max(q.rows, 1)));
        std::vector<float> d(size_t(std::max(q.rows, 1)));
        int32_t n = 0;
        int rc = hamming ? esfm_match_hamming(default_ctx(), q.ptr<uint8_t>(), q.rows, t.ptr<uint8_t>(), t.rows, q.cols, ratio, qi.data(),
                                              ti.data(), d.data(), &n)
                         : esfm_match_l2_f32(default_ctx(), q.ptr<float>(), q.rows, t.ptr<float>(), t.rows, q.cols, ratio, qi.data(),
                                             ti.data(), d.data(), &n);
        if (rc != ESFM_OK) { std::cerr << esfm_last_error() << std::endl; return false; }
        const size_t before = matches.size();
        for (int k = 0; k < n; ++k) matches.push_back(DMatch(qi[size_t(k)], ti[size_t(k)], 0, d[size_t(k)]));  // appended (:90, :135)
        if (!quiet) {
            std::chrono::duration<double> dt = std::chrono::steady_clock::now() - tic;
            std::cout << "match " << tag << " cost = " << dt.count() << " seconds. " << std::endl;
            std::cout << "# Correspondence: Initial [ " << q.rows << " ]  Filtered by Lowe ratio test [ " << matches.size() - before
                      << " ]" << std::endl;
        }
        return true;
    }
};

// ---- sparse-cloud post-processing (cloudprocessing.hpp:20-72, data_io.cpp:147-165) ---------------------
// CProceesing<PointT>::SORFilter = pcl::StatisticalOutlierRemoval (MeanK 50, StddevMulThresh 2.0) -> esfm_sor_filter.
template <typename PointT = PointXYZRGB> class CProceesing {
public:
    bool SORFilter(const std::vector<PointT> &incloud, std::vector<PointT> &outcloud, int MeanK = 50, double std = 2.0)
    {
        const int n = int(incloud.size());
        std::vector<float> xyz(size_t(3) * size_t(std::max(n, 1)));
        for (int i = 0; i < n; ++i) { xyz[size_t(3 * i)] = incloud[size_t(i)].x; xyz[size_t(3 * i + 1)] = incloud[size_t(i)].y; xyz[size_t(3 * i + 2)] = incloud[size_t(i)].z; }
        std::vector<uint8_t> keep(size_t(std::max(n, 1)));
        int32_t n_keep = 0;
        const int rc = esfm_sor_filter(default_ctx(), xyz.data(), n, 3, MeanK, std, nullptr, keep.data(), &n_keep, nullptr);
        if (rc != ESFM_OK) { std::cerr << esfm_last_error() << std::endl; return false; }
        outcloud.clear();
        for (int i = 0; i < n; ++i) if (keep[size_t(i)]) outcloud.push_back(incloud[size_t(i)]);
        std::cout << "apply SOR filter: [ " << n << " ] points before filtering, [ " << outcloud.size() << " ] points after filtering." << std::endl;
        return true;
    }
};

// DataIO::writePlyFile (data_io.cpp:147-165): width 1, height N, pcl::io::savePLYFile (ASCII, camera element) [upstream
// pcl/io/ply_io.cpp, restated from memory -- the reference ships no example file].
class DataIO {
public:
    // data_io.cpp:17-46: whitespace-separated file names, each joined with the folder; frame ids count up from 0
    bool importImageFilenames(const std::string image_list_path, const std::string image_data_path, std::vector<frame_t> &frames)
    {
        std::ifstream image_list_file(image_list_path.c_str(), std::ios::in);
        if (!image_list_file.is_open()) { std::cout << "open image_list_file failed, file is: " << image_list_path << std::endl; return 0; }
        int count = 0;
        std::string cur_file;
        while (image_list_file >> cur_file) {
            frames.push_back(frame_t(unsigned(count), image_data_path + "/" + cur_file));
            std::cout << count << ": " << frames.back().image_file_path << std::endl;
            ++count;
        }
        std::cout << "Frame number is " << frames.size() << std::endl;
        return 1;
    }

    // data_io.cpp:48-72: cv::imread(path, CV_LOAD_IMAGE_COLOR) -> 8-bit BGR (PNG files; esfm_png.hpp)
    bool importImages(frame_t &cur_frame, bool show = false)
    {
        (void)show;
        ImageMat &img = cur_frame.rgb_image;
        img.channels = 3;
        const std::string err = read_image_bgr(cur_frame.image_file_path, img.rows, img.cols, img.data);
        if (!err.empty()) { img = ImageMat(); std::cout << "No more images" << " (" << err << ")" << std::endl; return false; }
        return true;
    }

    // data_io.cpp:74-95: up to three rows of three numbers into K (later rows of a longer file are ignored)
    bool importCalib(const std::string &fileName, Matrix3f &K_mat)
    {
        std::ifstream in(fileName, std::ios::in);
        if (!in) return false;
        for (int i = 0; i < 3; ++i) {
            float a, b, c;
            if (!(in >> a >> b >> c)) break;
            K_mat(i, 0) = a; K_mat(i, 1) = b; K_mat(i, 2) = c;
        }
        std::cout << "Import camera calibration file done." << std::endl;
        return true;
    }

    bool writePlyFile(const std::string &fileName, const std::vector<PointXYZRGB> &pointCloud)
    {
        std::ofstream fs(fileName);
        if (!fs) { std::cerr << "Couldn't write file " << std::endl; return false; }
        const size_t n = pointCloud.size();
        fs << "ply\nformat ascii 1.0\ncomment PCL generated\nelement vertex " << n
           << "\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue"
              "\nelement camera 1\nproperty float view_px\nproperty float view_py\nproperty float view_pz"
              "\nproperty float x_axisx\nproperty float x_axisy\nproperty float x_axisz"
              "\nproperty float y_axisx\nproperty float y_axisy\nproperty float y_axisz"
              "\nproperty float z_axisx\nproperty float z_axisy\nproperty float z_axisz"
              "\nproperty float focal\nproperty float scalex\nproperty float scaley\nproperty float centerx\nproperty float centery"
              "\nproperty int viewportx\nproperty int viewporty\nproperty float k1\nproperty float k2\nend_header\n";
        fs << std::setprecision(8);
        for (const PointXYZRGB &p : pointCloud)
            fs << p.x << " " << p.y << " " << p.z << " " << int(p.r) << " " << int(p.g) << " " << int(p.b) << "\n";
        fs << "0 0 0 1 0 0 0 1 0 0 0 1 0 0 0 0 0 1 " << n << " 0 0\n";
        std::cout << "Output [ " << n << " ] points." << std::endl << "Output ply file done." << std::endl;
        return bool(fs);
    }

    // DataIO::importDistort (data_io.cpp:97-125): up to three groups of k1 k2 p1 p2 are extracted as floats, then stored with
    // at<float>(0, i) into the CV_64FC1 matrix -- i.e. into its first 16 bytes.  Reproduced: cv::undistort
    // sees v[0] = the double made of the bits of (k1, k2), v[1] = that of (p1, p2), v[2] = v[3] = 0.
    bool importDistort(const std::string &fileName, DistortMat &distort_coeff)
    {
        std::ifstream in(fileName, std::ios::in);
        if (!in) return false;
        int i = 0;
        float k[4] = {0.f, 0.f, 0.f, 0.f};
        while (!in.eof() && i < 3) {       // the reference's loop spins forever on a fourth group; three are read at most
            in >> k[0] >> k[1] >> k[2] >> k[3];
            if (in.fail()) break;
            ++i;
        }
        in.close();
        std::memcpy(distort_coeff.v, k, sizeof(k));
        std::cout << "Import camera distortion coefficients file done." << std::endl;
        return true;
    }
};

inline void angle_axis_to_rotation(const double aa[3], double R[9]);

// ---- two-view / 3-D/2-D geometry (estimate_motion.h) ------------------------------------------------------
class MotionEstimator {
public:
    // estimate_motion.cpp:27-97: cv::findEssentialMat(RANSAC) on the matched pixels with frame 1's K, inliers appended,
    // cv::recoverPose with the RANSAC mask -> T = [R | t; 0 0 0 1]
    bool estimate2D2D_E5P_RANSAC(frame_t &cur_frame_1, frame_t &cur_frame_2, std::vector<DMatch> &matches, std::vector<DMatch> &inlier_matches,
                                 Matrix4f &T, double ransac_thre = 1.0, double ransac_prob = 0.99, bool show = false)
    {
        std::vector<float> p1, p2;
        gather(cur_frame_1, cur_frame_2, matches, 1, p1, p2);
        const int n = int(matches.size());
        float K4[4]; k4_of(cur_frame_1.K_cam, K4);
        std::vector<uint8_t> mask(size_t(std::max(n, 1)));
        double E[9], R[9], t[3];
        int rc = esfm_find_essential_mat(default_ctx(), p1.data(), p2.data(), n, K4, ransac_prob, ransac_thre, E, mask.data(), nullptr);
        if (rc != ESFM_OK) { std::cerr << esfm_last_error() << std::endl; return false; }
        for (int i = 0; i < n; ++i) if (mask[size_t(i)]) inlier_matches.push_back(matches[size_t(i)]);   // :55-61
        rc = esfm_recover_pose(default_ctx(), E, p1.data(), p2.data(), n, K4, R, t, mask.data(), nullptr);   // :67
        if (rc != ESFM_OK) { std::cerr << esfm_last_error() << std::endl; return false; }
        T = Matrix4f::Identity();
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T(r, c) = float(R[3 * r + c]); T(r, 3) = float(t[r]); }   // :76-85
        if (!quiet) std::cout << "Find [" << inlier_matches.size() << "] inlier matches from [" << matches.size() << "] total matches." << std::endl;
        return true;
    }

    // estimate2D2D_E5P_RANSAC + getDepthFast (estimate_motion.cpp:27-97, :234-283; sfm.cpp:163-170) for MANY pairs in shared launches:
    // job p = (frame_1[p], frame_2[p], matches[p]).  ok[p] / inlier_matches[p] / T[p] / appro_depth[p] are what the per-pair members
    // return for that job (appro_depth stays untouched where the per-pair getDepthFast is not reached or fails).
    bool estimate2D2D_E5P_RANSAC_pairs(std::vector<frame_t> &frames, const std::vector<std::pair<int, int>> &jobs, const std::vector<std::vector<DMatch>> &matches,
                                       std::vector<std::vector<DMatch>> &inlier_matches, std::vector<Matrix4f> &T, std::vector<double> &appro_depth,
                                       std::vector<char> &ok, double ransac_thre = 1.0, double ransac_prob = 0.99, int random_rate = 20)
    {
        const size_t nj = jobs.size();
        inlier_matches.assign(nj, {}); T.assign(nj, Matrix4f::Identity()); ok.assign(nj, 0);
        appro_depth.resize(nj, 1.0);
        if (nj == 0) return true;
        std::vector<int32_t> off(nj + 1, 0);
        for (size_t p = 0; p < nj; ++p) off[p + 1] = off[p] + int32_t(matches[p].size());
        std::vector<float> p1(size_t(2) * size_t(std::max(off[nj], 1))), p2(p1.size()), K4(4 * nj);
        for (size_t p = 0; p < nj; ++p) {
            std::vector<float> a, b;
            gather(frames[size_t(jobs[p].first)], frames[size_t(jobs[p].second)], matches[p], 1, a, b);
            std::copy(a.begin(), a.end(), p1.begin() + 2 * off[p]); std::copy(b.begin(), b.end(), p2.begin() + 2 * off[p]);
            k4_of(frames[size_t(jobs[p].first)].K_cam, &K4[4 * p]);                           // frame 1's K for both images (:43-44)
        }
        std::vector<double> E(9 * nj), R(9 * nj), t(3 * nj);
        std::vector<uint8_t> mask(size_t(std::max(off[nj], 1)));
        std::vector<int32_t> status(nj, 0);
        int rc = esfm_find_essential_pairs(default_ctx(), int(nj), off.data(), p1.data(), p2.data(), K4.data(), ransac_prob, ransac_thre, E.data(), mask.data(),
                                           status.data(), nullptr);
        if (rc != ESFM_OK) { std::cerr << esfm_last_error() << std::endl; return false; }
        for (size_t p = 0; p < nj; ++p) {
            if (!status[p]) { for (int k = off[p]; k < off[p + 1]; ++k) mask[size_t(k)] = 0; continue; }
            for (int k = off[p]; k < off[p + 1]; ++k) if (mask[size_t(k)]) inlier_matches[p].push_back(matches[p][size_t(k - off[p])]);   // :55-61
        }
        rc = esfm_recover_pose_pairs(default_ctx(), int(nj), off.data(), p1.data(), p2.data(), K4.data(), E.data(), mask.data(), R.data(), t.data(), nullptr);   // :67
        if (rc != ESFM_OK) { std::cerr << esfm_last_error() << std::endl; return false; }
        // getDepthFast: every random_rate-th inlier match between [I | 0] and T_21, one triangulation launch for all jobs
        std::vector<int32_t> doff(1, 0);
        std::vector<float> P1s, P2s, da, db;
        std::vector<size_t> dj;
        for (size_t p = 0; p < nj; ++p) {
            if (!status[p]) { if (!quiet) std::cerr << "no essential matrix for pair ( " << jobs[p].first << " , " << jobs[p].second << " )" << std::endl; continue; }
            ok[p] = 1;
            for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T[p](r, c) = float(R[9 * p + size_t(3 * r + c)]); T[p](r, 3) = float(t[3 * p + size_t(r)]); }   // :76-85
            if (!quiet) std::cout << "Find [" << inlier_matches[p].size() << "] inlier matches from [" << matches[p].size() << "] total matches." << std::endl;
            std::vector<float> a, b;
            frame_t &f1 = frames[size_t(jobs[p].first)];
            gather(f1, frames[size_t(jobs[p].second)], inlier_matches[p], random_rate, a, b);
            if (a.empty()) { appro_depth[p] = std::nan(""); continue; }                  // 0 / 0 in the reference (:280)
            pixel2cam(a, f1.K_cam); pixel2cam(b, f1.K_cam);                               // frame 1's K for both (:245)
            const float I34[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
            P1s.insert(P1s.end(), I34, I34 + 12);
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) P2s.push_back(T[p](r, c));
            da.insert(da.end(), a.begin(), a.end()); db.insert(db.end(), b.begin(), b.end());
            doff.push_back(doff.back() + int32_t(a.size() / 2));
            dj.push_back(p);
        }
        if (!dj.empty()) {
            std::vector<float> hh(size_t(4) * size_t(doff.back()));
            rc = esfm_triangulate_pairs(default_ctx(), int(dj.size()), P1s.data(), P2s.data(), doff.data(), da.data(), db.data(), hh.data());
            if (rc != ESFM_OK) { std::cerr << esfm_last_error() << std::endl; return false; }
            for (size_t s = 0; s < dj.size(); ++s) {
                double depth_sum = 0;
                for (int i = doff[s]; i < doff[s + 1]; ++i) {
                    const float x = hh[size_t(4 * i)] / hh[size_t(4 * i + 3)], y = hh[size_t(4 * i + 1)] / hh[size_t(4 * i + 3)], z = hh[size_t(4 * i + 2)] / hh[size_t(4 * i + 3)];
                    depth_sum += double(std::sqrt(x * x + y * y + z * z));   // Eigen::Vector3f::norm()
                }
                appro_depth[dj[s]] = depth_sum / double(doff[s + 1] - doff[s]);
            }
        }
        return true;
    }

    // estimate_motion.cpp:234-283: every random_rate-th match triangulated between [I|0] and T_21, mean distance from camera 1
    bool getDepthFast(frame_t &cur_frame_1, frame_t &cur_frame_2, Matrix4f &T_21, const std::vector<DMatch> &matches, double &appro_depth,
                      int random_rate = 20)
    {
        std::vector<float> a, b;
        gather(cur_frame_1, cur_frame_2, matches, random_rate, a, b);
        const int n = int(a.size() / 2);
        pixel2cam(a, cur_frame_1.K_cam); pixel2cam(b, cur_frame_1.K_cam);   // frame 1's K for both (:245)
        float P1[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}, P2[12];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) P2[4 * r + c] = T_21(r, c);
        std::vector<float> h(size_t(4) * size_t(std::max(n, 1)));
        if (n > 0 && esfm_triangulate_points(default_ctx(), P1, P2, a.data(), b.data(), n, h.data()) != ESFM_OK) { std::cerr << esfm_last_error() << std::endl; return false; }
        double depth_sum = 0;
        for (int i = 0; i < n; ++i) {
            const float x = h[size_t(4 * i)] / h[size_t(4 * i + 3)], y = h[size_t(4 * i + 1)] / h[size_t(4 * i + 3)], z = h[size_t(4 * i + 2)] / h[size_t(4 * i + 3)];
            depth_sum += double(std::sqrt(x * x + y * y + z * z));   // Eigen::Vector3f::norm()
        }
        appro_depth = depth_sum / n;   // 0 / 0 for an empty sample, like the reference
        return true;
    }

    // estimate_motion.cpp:285-367 (without the colour lookup: frame_t carries no image here)
    bool doTriangulation(frame_t &cur_frame_1, frame_t &cur_frame_2, const std::vector<DMatch> &matches, pointcloud_sparse_t &sparse_pointcloud,
                         bool show = false)
    {
        std::unordered_map<int, int> known;
        for (int id : sparse_pointcloud.unique_point_ids) known.emplace(id, 1);
        std::vector<float> a, b;
        int count_new = 0;
        for (const DMatch &m : matches) {
            const int uid = cur_frame_1.unique_pixel_ids[size_t(m.queryIdx)];
            if (known.count(uid)) continue;
            known.emplace(uid, 1);
            sparse_pointcloud.unique_point_ids.push_back(uid);
            sparse_pointcloud.is_inlier.push_back(1);
            a.push_back(cur_frame_1.keypoints[size_t(m.queryIdx)].pt.x); a.push_back(cur_frame_1.keypoints[size_t(m.queryIdx)].pt.y);
            b.push_back(cur_frame_2.keypoints[size_t(m.trainIdx)].pt.x); b.push_back(cur_frame_2.keypoints[size_t(m.trainIdx)].pt.y);
            ++count_new;
        }
        pixel2cam(a, cur_frame_1.K_cam); pixel2cam(b, cur_frame_1.K_cam);
        float P1[12], P2[12];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) { P1[4 * r + c] = cur_frame_1.pose_cam(r, c); P2[4 * r + c] = cur_frame_2.pose_cam(r, c); }
        std::vector<float> h(size_t(4) * size_t(std::max(count_new, 1)));
        if (count_new > 0 && esfm_triangulate_points(default_ctx(), P1, P2, a.data(), b.data(), count_new, h.data()) != ESFM_OK) { std::cerr << esfm_last_error() << std::endl; return false; }
        const ImageMat &img = cur_frame_1.rgb_image;
        for (int i = 0; i < count_new; ++i) {
            PointXYZRGB p;
            p.x = h[size_t(4 * i)] / h[size_t(4 * i + 3)]; p.y = h[size_t(4 * i + 1)] / h[size_t(4 * i + 3)]; p.z = h[size_t(4 * i + 2)] / h[size_t(4 * i + 3)];
            if (!img.empty() && img.channels == 3) {
                // colour from frame 1's image at the keypoint of matches[i] -- the reference indexes the match list with the
                // index of the i-th NEW point (estimate_motion.cpp:345), which is another match once any was skipped
                const Point2f &px = cur_frame_1.keypoints[size_t(matches[size_t(i)].queryIdx)].pt;
                const int y = int(px.y), x = int(px.x);
                if (y >= 0 && y < img.rows && x >= 0 && x < img.cols) {
                    const uint8_t *bgr = &img.data[(size_t(y) * size_t(img.cols) + size_t(x)) * 3];
                    p.b = bgr[0]; p.g = bgr[1]; p.r = bgr[2];
                }
            }
            sparse_pointcloud.points.push_back(p);
        }
        if (!quiet) std::cout << "Triangulate [ " << count_new << " ] new points, [ " << sparse_pointcloud.points.size() << " ] points in total." << std::endl;
        return true;
    }

    // estimate_motion.cpp:99-232: id join (keypoint-major, +-300 gate), solvePnPRansac(EPnP), pose write-back, the reference's
    // inlier bookkeeping (the int index matrix read as float always addresses correspondence 0)
    bool estimate2D3D_P3P_RANSAC(frame_t &cur_frame, pointcloud_sparse_t &cur_map_3d, double ransac_thre = 2.5, int iterationsCount = 50000,
                                 double ransac_prob = 0.99, bool show = false)
    {
        std::unordered_map<int, std::vector<int>> where;
        for (size_t j = 0; j < cur_map_3d.unique_point_ids.size(); ++j) where[cur_map_3d.unique_point_ids[j]].push_back(int(j));
        std::vector<float> p2, p3;
        std::vector<int> index;
        const float dist_thre = 300;
        for (size_t i = 0; i < cur_frame.unique_pixel_ids.size(); ++i) {
            auto it = where.find(cur_frame.unique_pixel_ids[i]);
            if (it == where.end()) continue;
            for (int j : it->second) {
                const PointXYZRGB &q = cur_map_3d.points[size_t(j)];
                if (std::abs(q.x) < dist_thre && std::abs(q.y) < dist_thre && std::abs(q.z) < dist_thre) {
                    p2.push_back(cur_frame.keypoints[i].pt.x); p2.push_back(cur_frame.keypoints[i].pt.y);
                    p3.push_back(q.x); p3.push_back(q.y); p3.push_back(q.z);
                    index.push_back(j);
                }
            }
        }
        const int count = int(index.size());
        float K4[4]; k4_of(cur_frame.K_cam, K4);
        double rv[3], tv[3], R[9];
        int32_t n_inl = 0;
        std::vector<uint8_t> mask(size_t(std::max(count, 1)));
        int rc = esfm_solve_pnp_ransac(default_ctx(), p3.data(), p2.data(), count, K4, iterationsCount, ransac_thre, ransac_prob, rv, tv, R, mask.data(),
                                       &n_inl, nullptr);
        if (rc != ESFM_OK) { std::cerr << esfm_last_error() << std::endl; return false; }
        if (n_inl > 0 && !index.empty()) index[0] = -1;   // inliers.at<float>(i, 0) on CV_32S (:169)
        for (int j : index) if (j >= 0) cur_map_3d.is_inlier[size_t(j)] = 0;
        double Rr[9];
        angle_axis_to_rotation(rv, Rr);   // cv::Rodrigues(r_vec, R_mat) (:184)
        cur_frame.pose_cam = Matrix4f::Identity();
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) cur_frame.pose_cam(r, c) = float(Rr[3 * r + c]); cur_frame.pose_cam(r, 3) = float(tv[r]); }
        float reproj_err = 0.f;
        for (int i = 0; i < count; ++i) {
            const double X = p3[size_t(3 * i)], Y = p3[size_t(3 * i + 1)], Z = p3[size_t(3 * i + 2)];
            const double xc = Rr[0] * X + Rr[1] * Y + Rr[2] * Z + tv[0], yc = Rr[3] * X + Rr[4] * Y + Rr[5] * Z + tv[1], zc = Rr[6] * X + Rr[7] * Y + Rr[8] * Z + tv[2];
            const float u = float(xc / zc * double(K4[0]) + double(K4[1])), v = float(yc / zc * double(K4[2]) + double(K4[3]));
            const float dx = u - p2[size_t(2 * i)], dy = v - p2[size_t(2 * i + 1)];
            reproj_err += std::sqrt(dx * dx + dy * dy);
        }
        reproj_err /= float(count);
        const double inlier_ratio = 1.0 * n_inl / count;
        if (!quiet) std::cout << "Inlier count: " << n_inl << std::endl << "Mean reprojection error: " << reproj_err << std::endl;
        return !(reproj_err > 10 && inlier_ratio < 0.5);
    }

    // estimate_motion.cpp:431-441: cv::undistort(cur_frame.rgb_image, undistorted_img, K_cam, distort_coeff)
    bool doUnDistort(frame_t &cur_frame, const DistortMat &distort_coeff)
    {
        ImageMat &img = cur_frame.rgb_image;
        if (img.empty()) { std::cerr << "frame has no image" << std::endl; return false; }
        const double K4[4] = {cur_frame.K_cam(0, 0), cur_frame.K_cam(0, 2), cur_frame.K_cam(1, 1), cur_frame.K_cam(1, 2)};
        std::vector<uint8_t> undistorted_img(img.data.size());
        if (esfm_undistort(default_ctx(), img.data.data(), img.rows, img.cols, img.channels, K4, distort_coeff.v, undistorted_img.data()) != ESFM_OK) {
            std::cerr << esfm_last_error() << std::endl;
            return false;
        }
        img.data.swap(undistorted_img);
        if (!quiet) std::cout << "Undistort the image done." << std::endl;
        return true;
    }

    // estimate_motion.cpp:476-505
    bool outlierFilter(pointcloud_sparse_t &sparse_pointcloud, int MeanK = 40, double std = 2.5)
    {
        const int n = int(sparse_pointcloud.points.size());
        std::vector<float> xyz(size_t(3) * size_t(std::max(n, 1)));
        for (int i = 0; i < n; ++i) { xyz[size_t(3 * i)] = sparse_pointcloud.points[size_t(i)].x; xyz[size_t(3 * i + 1)] = sparse_pointcloud.points[size_t(i)].y; xyz[size_t(3 * i + 2)] = sparse_pointcloud.points[size_t(i)].z; }
        std::vector<uint8_t> keep(size_t(std::max(n, 1)));
        if (esfm_sor_filter(default_ctx(), xyz.data(), n, 3, MeanK, std, nullptr, keep.data(), nullptr, nullptr) != ESFM_OK) { std::cerr << esfm_last_error() << std::endl; return false; }
        pointcloud_sparse_t out;
        for (int i = 0; i < n; ++i) if (keep[size_t(i)]) {
            out.points.push_back(sparse_pointcloud.points[size_t(i)]);
            out.unique_point_ids.push_back(sparse_pointcloud.unique_point_ids[size_t(i)]);
            out.is_inlier.push_back(sparse_pointcloud.is_inlier[size_t(i)]);
        }
        sparse_pointcloud.points.swap(out.points); sparse_pointcloud.unique_point_ids.swap(out.unique_point_ids); sparse_pointcloud.is_inlier.swap(out.is_inlier);
        return true;
    }

    bool quiet = false;

private:
    static void k4_of(const Matrix3f &K, float K4[4]) { K4[0] = K(0, 0); K4[1] = K(0, 2); K4[2] = K(1, 1); K4[3] = K(1, 2); }
    static void gather(const frame_t &f1, const frame_t &f2, const std::vector<DMatch> &matches, int rate, std::vector<float> &a, std::vector<float> &b)
    {
        for (size_t i = 0; i < matches.size(); ++i) {
            if (int(i) % rate != 0) continue;
            a.push_back(f1.keypoints[size_t(matches[i].queryIdx)].pt.x); a.push_back(f1.keypoints[size_t(matches[i].queryIdx)].pt.y);
            b.push_back(f2.keypoints[size_t(matches[i].trainIdx)].pt.x); b.push_back(f2.keypoints[size_t(matches[i].trainIdx)].pt.y);
        }
    }
    // estimate_motion.h:41-46, float arithmetic
    static void pixel2cam(std::vector<float> &p, const Matrix3f &K)
    {
        for (size_t i = 0; i + 1 < p.size(); i += 2) { p[i] = (p[i] - K(0, 2)) / K(0, 0); p[i + 1] = (p[i + 1] - K(1, 2)) / K(1, 1); }
    }
};

// ---- cv::Rodrigues as used at ba.cpp:82 (matrix -> vector) and :239 (vector -> matrix) -------------
inline void rotation_to_angle_axis(const double R[9], double aa[3])
{
    const double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = std::sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1.0) * 0.5;
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
    const double theta = std::acos(c);
    if (s < 1e-5) {
        if (c > 0) { aa[0] = aa[1] = aa[2] = 0.0; return; }
        double v[3] = {std::sqrt(std::max((R[0] + 1) * 0.5, 0.0)), std::sqrt(std::max((R[4] + 1) * 0.5, 0.0)),
                       std::sqrt(std::max((R[8] + 1) * 0.5, 0.0))};
        if (R[1] < 0) v[1] = -v[1];
        if (R[2] < 0) v[2] = -v[2];
        if ((R[5] > 0) != (v[1] * v[2] > 0)) v[2] = -v[2];
        const double nrm = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        for (int i = 0; i < 3; ++i) aa[i] = v[i] * (theta / std::max(nrm, 1e-300));
        return;
    }
    const double k = 0.5 * theta / s;
    aa[0] = rx * k; aa[1] = ry * k; aa[2] = rz * k;
}

inline void angle_axis_to_rotation(const double aa[3], double R[9])
{
    const double theta = std::sqrt(aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2]);
    if (theta < 2.2204460492503131e-16) { for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0; return; }
    const double k[3] = {aa[0] / theta, aa[1] / theta, aa[2] / theta};
    const double c = std::cos(theta), s = std::sin(theta), c1 = 1.0 - c;
    R[0] = c + c1 * k[0] * k[0]; R[1] = c1 * k[0] * k[1] - s * k[2]; R[2] = c1 * k[0] * k[2] + s * k[1];
    R[3] = c1 * k[1] * k[0] + s * k[2]; R[4] = c + c1 * k[1] * k[1]; R[5] = c1 * k[1] * k[2] - s * k[0];
    R[6] = c1 * k[2] * k[0] - s * k[1]; R[7] = c1 * k[2] * k[1] + s * k[0]; R[8] = c + c1 * k[2] * k[2];
}

// ---- bundle adjustment (ba.h:28-106) -----------------------------------------------------------------
class BundleAdjustment {
public:
    BundleAdjustment() { initBA(); esfm_ba_options_default(&options_); }

    int num_observations() const { return num_observations_; }
    double *mutable_cameras() { return parameters_.data(); }
    double *mutable_points() { return parameters_.data() + 6 * num_cameras_; }

    bool initBA()  // ba.h:59-75
    {
        point_index_.clear(); camera_index_.clear(); points_2d_.clear(); calibs_.clear(); parameters_.clear();
        num_cameras_ = num_points_ = num_parameters_ = num_observations_ = 0;
        ref_process_camera_id_ = -1;
        if (verbose) std::cout << "Bundle Ajustment parameters initialization done." << std::endl;
        return true;
    }

    // ba.cpp:12-130.  Same observation list in the same order (camera-major, point index ascending; for a
    // point the FIRST keypoint with has_match and a matching id, the `break` at :44) as the reference's
    // O(Ncam*Npts*Nkp) triple loop, derived with one hash map per frame.
    bool setBAProblem(std::vector<frame_t> &frames, std::vector<bool> &process_frame_id, pointcloud_sparse_t &sfm_sparse_points,
                      double fix_calib_tolerance_BA, int reference_frame_id)
    {
        num_observations_ = 0; num_cameras_ = 0;
        num_points_ = int(sfm_sparse_points.unique_point_ids.size());
        for (size_t i = 0; i < frames.size(); ++i) {
            if (process_frame_id[i]) continue;  // 0 = registered
            calibs_.push_back(frames[i].K_cam);
            std::unordered_map<int, int> first_kp;  // track id -> lowest keypoint index with has_match
            const frame_t &fr = frames[i];
            for (size_t j = 0; j < fr.unique_pixel_ids.size(); ++j)
                if (fr.unique_pixel_has_match[j]) first_kp.emplace(fr.unique_pixel_ids[j], int(j));  // emplace keeps the first
            for (int k = 0; k < num_points_; ++k) {
                auto it = first_kp.find(sfm_sparse_points.unique_point_ids[size_t(k)]);
                if (it == first_kp.end()) continue;
                points_2d_.push_back(fr.keypoints[size_t(it->second)].pt);
                point_index_.push_back(k);
                camera_index_.push_back(num_cameras_);
                ++num_observations_;
            }
            if (int(i) == reference_frame_id) ref_process_camera_id_ = num_cameras_;
            ++num_cameras_;
        }
        num_parameters_ = 6 * num_cameras_ + 3 * num_points_ + (fix_calib_tolerance_BA != 0 ? 4 : 0);
        parameters_.assign(size_t(num_parameters_), 0.0);
        int k = 0;
        for (size_t i = 0; i < frames.size(); ++i) {
            if (process_frame_id[i]) continue;
            double R[9], aa[3];
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R[3 * r + c] = double(frames[i].pose_cam(r, c));
            rotation_to_angle_axis(R, aa);
            for (int a = 0; a < 3; ++a) {
                parameters_[size_t(6 * k + a)] = double(float(aa[a]));  // rot_vec is CV_32F (:86-88)
                parameters_[size_t(6 * k + 3 + a)] = double(frames[i].pose_cam(a, 3));
            }
            ++k;
        }
        for (int i = 0; i < num_points_; ++i) {
            const PointXYZRGB &p = sfm_sparse_points.points[size_t(i)];
            parameters_[size_t(6 * num_cameras_ + 3 * i)] = p.x;
            parameters_[size_t(6 * num_cameras_ + 3 * i + 1)] = p.y;
            parameters_[size_t(6 * num_cameras_ + 3 * i + 2)] = p.z;
        }
        if (fix_calib_tolerance_BA != 0 && !calibs_.empty()) {
            parameters_[size_t(num_parameters_ - 4)] = calibs_[0](0, 0); parameters_[size_t(num_parameters_ - 3)] = calibs_[0](0, 2);
            parameters_[size_t(num_parameters_ - 2)] = calibs_[0](1, 1); parameters_[size_t(num_parameters_ - 1)] = calibs_[0](1, 2);
        }
        if (verbose) {
            std::cout << "Find [ " << num_observations_ << " ] Observations in total." << std::endl;
            std::cout << "There are [ " << num_cameras_ << " ] cameras and [ " << num_points_ << " ] 3D points." << std::endl;
        }
        return 2 * num_observations_ > num_parameters_;  // "Ready to solve" (:120-129)
    }

    double *mutable_calib() { return parameters_.data() + 6 * num_cameras_ + 3 * num_points_; }   // ba.h:45

    // ba.cpp:132-212: ceres::Solve -> esfm_ba_solve_ex.  Fixed intrinsics (:142-164) or the shared free block bounded to
    // +- tolerance (:167-196); a reference frame named in setBAProblem is bounded to +-1e-10 (:134, :155-162).
    bool solveBA(double fix_calib_tolerance_BA)
    {
        const double fixed_threshold = 1e-10;  // :134
        std::vector<float> K4(size_t(4 * num_cameras_));
        for (int c = 0; c < num_cameras_; ++c) {
            K4[size_t(4 * c)] = calibs_[size_t(c)](0, 0); K4[size_t(4 * c + 1)] = calibs_[size_t(c)](0, 2);
            K4[size_t(4 * c + 2)] = calibs_[size_t(c)](1, 1); K4[size_t(4 * c + 3)] = calibs_[size_t(c)](1, 2);
        }
        options_.verbose = verbose ? 1 : 0;  // minimizer_progress_to_stdout (:204)
        int rc = esfm_ba_solve_ex(default_ctx(), num_cameras_, num_points_, num_observations_, camera_index_.data(), point_index_.data(),
                                  reinterpret_cast<const float *>(points_2d_.data()), K4.data(), mutable_cameras(), mutable_points(),
                                  fix_calib_tolerance_BA != 0 ? mutable_calib() : nullptr, fix_calib_tolerance_BA,
                                  ref_process_camera_id_, fixed_threshold, &options_, nullptr, nullptr, &summary_);
        if (rc != ESFM_OK) { std::cerr << esfm_last_error() << std::endl; return false; }
        return true;
    }

    // ba.cpp:214-288
    bool doSFMBA(std::vector<frame_t> &frames, std::vector<bool> &process_frame_id, pointcloud_sparse_t &sfm_sparse_points,
                 double fix_calib_tolerance_BA = 0.0, int reference_frame_id = -1)
    {
        auto tic = std::chrono::steady_clock::now();
        initBA();
        setBAProblem(frames, process_frame_id, sfm_sparse_points, fix_calib_tolerance_BA, reference_frame_id);
        const bool ok = solveBA(fix_calib_tolerance_BA);
        int k = 0;
        for (size_t i = 0; i < frames.size(); ++i) {
            if (process_frame_id[i]) continue;
            const double aa[3] = {double(float(parameters_[size_t(6 * k)])), double(float(parameters_[size_t(6 * k + 1)])),
                                  double(float(parameters_[size_t(6 * k + 2)]))};  // rot_vec float (:235-237)
            double R[9];
            angle_axis_to_rotation(aa, R);
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) frames[i].pose_cam(r, c) = float(R[3 * r + c]);
                frames[i].pose_cam(r, 3) = float(parameters_[size_t(6 * k + 3 + r)]);
            }
            ++k;
            if (fix_calib_tolerance_BA != 0) {  // :250-256
                frames[i].K_cam(0, 0) = float(parameters_[size_t(num_parameters_ - 4)]);
                frames[i].K_cam(0, 2) = float(parameters_[size_t(num_parameters_ - 3)]);
                frames[i].K_cam(1, 1) = float(parameters_[size_t(num_parameters_ - 2)]);
                frames[i].K_cam(1, 2) = float(parameters_[size_t(num_parameters_ - 1)]);
            }
        }
        if (verbose && fix_calib_tolerance_BA != 0)
            std::cout << "Calib intrinsic parameters after BA:" << std::endl
                      << "[fx:" << parameters_[size_t(num_parameters_ - 4)] << " ,cx:" << parameters_[size_t(num_parameters_ - 3)]
                      << " ,fy:" << parameters_[size_t(num_parameters_ - 2)] << " ,cy:" << parameters_[size_t(num_parameters_ - 1)] << " ]"
                      << std::endl;
        for (int i = 0; i < num_points_; ++i) {  // :277-279, float truncation
            sfm_sparse_points.points[size_t(i)].x = float(parameters_[size_t(6 * num_cameras_ + 3 * i)]);
            sfm_sparse_points.points[size_t(i)].y = float(parameters_[size_t(6 * num_cameras_ + 3 * i + 1)]);
            sfm_sparse_points.points[size_t(i)].z = float(parameters_[size_t(6 * num_cameras_ + 3 * i + 2)]);
        }
        if (verbose) {
            std::chrono::duration<double> dt = std::chrono::steady_clock::now() - tic;
            std::cout << "Bundle Ajustment cost = " << dt.count() << " seconds. " << std::endl << "Bundle Ajustment done." << std::endl;
        }
        return ok;
    }

    bool verbose = false;
    esfm_ba_options options_;
    esfm_ba_summary summary_;
    // the reference's private state (ba.h:86-105), public here for the tests
    int num_cameras_ = 0, num_points_ = 0, num_observations_ = 0, num_parameters_ = 0;
    std::vector<int> point_index_, camera_index_;
    std::vector<double> parameters_;  // [6 per camera (rot, tran) | 3 per point | optional fx,cx,fy,cy]
    std::vector<Point2f> points_2d_;
    std::vector<Matrix3f> calibs_;
    int ref_process_camera_id_ = -1;
};

}  // namespace p3dv
