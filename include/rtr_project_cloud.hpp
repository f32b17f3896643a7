// rtr_project_cloud.hpp -- header-only C++ facade with the reference's ProjectCloud surface
// (reference: src/RTRenderer/include/project_cloud.h:11-19) over the C ABI of rtr.h.
//
// The reference's header hard-wires cv::Mat, cv::Matx44d, CameraCalibration and
// std::unordered_map<int, OctreeGrid::Block>.  None of OpenCV / glm is in this build image,
// so the facade is written against the *members the reference actually uses* and accepts any
// types that provide them -- the reference's own types satisfy every requirement, so
// `rtr::ProjectCloud` is a source-level drop-in for `::ProjectCloud` in
// example/render_trajectory/main.cpp:87-96 and cloudreader.cpp:233-246:
//   Grid        : iterable of pairs whose .second has .positions (elements with .x .y .z) and
//                 .colors (elements indexable [0..2])            (Octreegrid.h:16-21,162-180)
//   Calibration : getWidth(), getHeight(), getIntrinsicsMatrix()(r,c)   (CameraCalibration.h)
//   Extrinsics  : operator()(r,c) -> double, world->camera              (cv::Matx44d)
//   Image       : template ptr<T>() -> T*  (cv::Mat::ptr<uint8_t>() / ptr<float>()),
//                 caller-allocated CV_8UC3 / CV_32F of size W x H      (main.cpp:93-94)
// Return codes as in project_cloud.cu:268-312: 1 on success, -1 when both outputs are null.
// Unlike the reference (exit(1) on CUDA errors, project_cloud.cu:13-17) failures throw.
//
// computeFull (project_cloud.h:17-18, project_cloud.cu:437-493) needs libtorch: define RTR_WITH_TORCH
// before including this header (and link libtorch); without it the class has the two projection
// methods and tensor(), the device pointer computeFull hands to the U-Net.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#ifdef RTR_WITH_TORCH
#include <torch/script.h>

#include <cstring>
#include <filesystem>
#endif

#include "rtr.h"

namespace rtr {

class ProjectCloud {
public:
    template <class Grid>
    explicit ProjectCloud(const Grid& grid, const std::string& modelFilename = std::string(""), int device = 0)
        : model_filename_(modelFilename) {
        // OctreeGrid::getVertexPositions / getVertexColors (Octreegrid.h:162-180)
        std::vector<float> xyzw;
        std::vector<uint8_t> rgba;
        for (const auto& pair : grid) {
            for (const auto& p : pair.second.positions) {
                xyzw.push_back(p.x); xyzw.push_back(p.y); xyzw.push_back(p.z); xyzw.push_back(1.0f);
            }
            for (const auto& c : pair.second.colors) {
                rgba.push_back(c[0]); rgba.push_back(c[1]); rgba.push_back(c[2]); rgba.push_back(255);
            }
        }
        if (rtr_abi_version() != RTR_ABI_VERSION)  // (a library built from another rtr.h: struct sizes may differ)
            throw std::runtime_error("librtr_hip.so has ABI version " + std::to_string(rtr_abi_version()) +
                                     ", this header is version " + std::to_string(RTR_ABI_VERSION));
        check(nullptr, rtr_create(&ctx_, device));
        // (the library's default upload policy: the point order is measured and the cloud Morton-sorted once when its
        // 256-point chunks are not compact -- the grid's 0.25 m blocks are unordered inside -- then packed losslessly)
        check(ctx_, rtr_upload_points(ctx_, xyzw.data(), 16, rgba.data(), 4, xyzw.size() / 4));
#ifdef RTR_WITH_TORCH
        load_model(device);
#endif
    }
    ProjectCloud(const ProjectCloud&) = delete;  // owns device buffers (the reference forgets this)
    ProjectCloud& operator=(const ProjectCloud&) = delete;
    ~ProjectCloud() { rtr_destroy(ctx_); }

    template <class Calibration, class Extrinsics, class Image>
    int computeRGBD(const Calibration& calibration, const Extrinsics& extrinsics, Image* color, Image* depth) {
        return frame(calibration, extrinsics, color, depth, false);
    }
    template <class Calibration, class Extrinsics, class Image>
    int computeFilteredRGBD(const Calibration& calibration, const Extrinsics& extrinsics, Image* color, Image* depth) {
        return frame(calibration, extrinsics, color, depth, true);
    }
    // literal nullptr for one output, as in cloudreader.cpp:246 `computeRGBD(calib, pose, nullptr, &depth)`
    template <class Calibration, class Extrinsics, class Image>
    int computeRGBD(const Calibration& c, const Extrinsics& e, std::nullptr_t, Image* depth) {
        return frame(c, e, static_cast<Image*>(nullptr), depth, false);
    }
    template <class Calibration, class Extrinsics, class Image>
    int computeRGBD(const Calibration& c, const Extrinsics& e, Image* color, std::nullptr_t) {
        return frame(c, e, color, static_cast<Image*>(nullptr), false);
    }
    template <class Calibration, class Extrinsics>
    int computeRGBD(const Calibration&, const Extrinsics&, std::nullptr_t, std::nullptr_t) { return -1; }
    template <class Calibration, class Extrinsics, class Image>
    int computeFilteredRGBD(const Calibration& c, const Extrinsics& e, std::nullptr_t, Image* depth) {
        return frame(c, e, static_cast<Image*>(nullptr), depth, true);
    }
    template <class Calibration, class Extrinsics, class Image>
    int computeFilteredRGBD(const Calibration& c, const Extrinsics& e, Image* color, std::nullptr_t) {
        return frame(c, e, color, static_cast<Image*>(nullptr), true);
    }
    template <class Calibration, class Extrinsics>
    int computeFilteredRGBD(const Calibration&, const Extrinsics&, std::nullptr_t, std::nullptr_t) { return -1; }
#ifdef RTR_WITH_TORCH
    // computeFull (project_cloud.cu:437-493): projection + prefilter, then the TorchScript model on the
    // RESIDENT fp16 {1,5,H,W} tensor (torch::from_blob, no copy: :471), output[0] permuted to H x W x 3
    // (:475) and converted like cv::Mat::convertTo(CV_8UC3, 255.0) (:480: scale, round half to even,
    // saturate); depth <- the prefiltered depth buffer (:485).  Either output may be null; returns 1.
    // The projector runs on HIP's default stream here, which is torch's current stream unless the caller
    // changed it: kernels and the model are ordered without a host synchronisation.
    template <class Calibration, class Extrinsics, class Image>
    int computeFull(const Calibration& calibration, const Extrinsics& extrinsics, Image* color, Image* depth) {
        if (!has_model_) throw std::runtime_error("rtr: No model file name given, computeFull will not work");  // :247
        const int W = calibration.getWidth(), H = calibration.getHeight();
        float P[16];
        projection(calibration, extrinsics, P);
        check(ctx_, rtr_set_resolution(ctx_, W, H));
        check(ctx_, rtr_render(ctx_, P, 1));
        torch::Tensor input = torch::from_blob(tensor(), {1, 5, H, W},
                                               torch::TensorOptions().dtype(torch::kFloat16).device(torch::kCUDA, device_));
        torch::NoGradGuard no_grad;
        torch::Tensor output = model_.forward({input}).toTensor();
        output = output[0].permute({1, 2, 0}).contiguous();
        if (color != nullptr) {
            torch::Tensor u8 = output.to(torch::kFloat32).mul(255.0).round().clamp(0.0, 255.0).to(torch::kUInt8).cpu();
            std::memcpy(color->template ptr<uint8_t>(), u8.data_ptr<uint8_t>(), (size_t)W * H * 3);
        }
        if (depth != nullptr)
            check(ctx_, rtr_download_buffer(ctx_, RTR_BUF_DEPTH, depth->template ptr<float>(), (size_t)W * H * 4));
        return 1;
    }
    template <class Calibration, class Extrinsics, class Image>
    int computeFull(const Calibration& c, const Extrinsics& e, Image* color, std::nullptr_t) {
        return computeFull(c, e, color, static_cast<Image*>(nullptr));
    }
    template <class Calibration, class Extrinsics, class Image>
    int computeFull(const Calibration& c, const Extrinsics& e, std::nullptr_t, Image* depth) {
        return computeFull(c, e, static_cast<Image*>(nullptr), depth);
    }
    // a model object instead of a file under $HOME/.render_cache (tests, callers that build their own)
    void set_model(torch::jit::Module m) {
        model_ = std::move(m);
        model_.to(torch::Device(torch::kCUDA, device_));
        has_model_ = true;
        check(ctx_, rtr_set_stream(ctx_, nullptr));
    }
#endif
    // computeFull (project_cloud.cu:437-493) = computeFilteredRGBD + the caller's U-Net on this
    // device pointer: torch::from_blob(tensor(), {1,5,H,W}, fp16, kCUDA)   (project_cloud.cu:471)
    void* tensor() const {
        void* p = nullptr;
        check(ctx_, rtr_device_buffer(ctx_, RTR_BUF_TENSOR, &p, nullptr));
        return p;
    }
    rtr_ctx* context() const { return ctx_; }

private:
    template <class Calibration, class Extrinsics>
    static void projection(const Calibration& calibration, const Extrinsics& extrinsics, float P[16]) {
        double K[9], E[16];
        const auto Km = calibration.getIntrinsicsMatrix();
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) K[3 * r + c] = Km(r, c);
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) E[4 * r + c] = extrinsics(r, c);
        rtr_compose_projection(K, E, P);  // project_cloud.cu:318
    }
#ifdef RTR_WITH_TORCH
    void load_model(int device) {  // project_cloud.cu:225-250 (throws where the reference exits)
        device_ = device;
        if (model_filename_.empty()) return;  // "No model file name given, computeFull will not work."
        const char* home = std::getenv("HOME");
        const std::string path = (std::filesystem::path(home ? home : "") / ".render_cache" / model_filename_).string();
        if (!std::filesystem::exists(path))
            throw std::runtime_error("rtr: Model file does not exist: " + path + " (export a TorchScript model for this "
                                     "camera resolution first)");
        set_model(torch::jit::load(path));
    }
    torch::jit::Module model_;
    bool has_model_ = false;
    int device_ = 0;
#endif
    template <class Calibration, class Extrinsics, class Image>
    int frame(const Calibration& calibration, const Extrinsics& extrinsics, Image* color, Image* depth, bool filtered) {
        if (color == nullptr && depth == nullptr) return -1;  // project_cloud.cu:270-273
        float P[16];
        projection(calibration, extrinsics, P);
        check(ctx_, rtr_set_resolution(ctx_, calibration.getWidth(), calibration.getHeight()));  // :275-298
        uint8_t* c8 = color ? color->template ptr<uint8_t>() : nullptr;
        float* d32 = depth ? depth->template ptr<float>() : nullptr;
        check(ctx_, filtered ? rtr_project_filtered(ctx_, P, c8, d32) : rtr_project(ctx_, P, c8, d32));
        return 1;
    }
    static void check(const rtr_ctx* c, int rc) {
        if (rc != RTR_OK) throw std::runtime_error(std::string("rtr: ") + rtr_last_error(c));
    }
    rtr_ctx* ctx_ = nullptr;
    std::string model_filename_;
};

}  // namespace rtr
