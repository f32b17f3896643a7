// C++ computeFull of include/rtr_project_cloud.hpp (RTR_WITH_TORCH) on ROCm libtorch: the TorchScript
// model (a file under $HOME/.render_cache, like project_cloud.cu:225-246) must receive the library's
// resident fp16 {1,5,H,W} tensor -- checked by pointer through a second model that returns its input --
// and its output goes through the convertTo(CV_8UC3, 255.0) step.  Stand-in input types as in
// facade_check.cpp (TEST INPUT TYPES, not a build of the reference).
//   compute_full_check <cloud.bin> <W> <H> <K9+E16 doubles .bin> <model.pt name> <out_prefix>
#define RTR_WITH_TORCH
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#include "rtr_project_cloud.hpp"

struct P3 { float x, y, z; };
struct C3 { unsigned char v[3]; unsigned char operator[](int i) const { return v[i]; } };
struct Block { std::vector<P3> positions; std::vector<C3> colors; };
struct K33 { double m[9]; double operator()(int r, int c) const { return m[3 * r + c]; } };
struct M44 { double m[16]; double operator()(int r, int c) const { return m[4 * r + c]; } };
struct Calib {
    K33 K; int w, h;
    int getWidth() const { return w; }
    int getHeight() const { return h; }
    K33 getIntrinsicsMatrix() const { return K; }
};
struct Img {
    std::vector<unsigned char> bytes;
    template <class T> T* ptr() { return reinterpret_cast<T*>(bytes.data()); }
};

static void dump(const std::string& path, const void* p, size_t n) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f || fwrite(p, 1, n, f) != n) { perror(path.c_str()); exit(2); }
    fclose(f);
}

int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    unsigned long long n = 0;
    if (!f || fread(&n, 8, 1, f) != 1) return 2;
    std::map<int, Block> grid;
    std::vector<P3> pts(n); std::vector<C3> cols(n);
    if (fread(pts.data(), 12, n, f) != n || fread(cols.data(), 3, n, f) != n) return 2;
    fclose(f);
    for (unsigned long long i = 0; i < n; ++i) {
        Block& b = grid[i < n / 2 ? 0 : 1];
        b.positions.push_back(pts[i]); b.colors.push_back(cols[i]);
    }
    const int W = atoi(argv[2]), H = atoi(argv[3]);
    Calib cal; M44 E;
    f = fopen(argv[4], "rb");
    if (!f || fread(cal.K.m, 8, 9, f) != 9 || fread(E.m, 8, 16, f) != 16) return 2;
    fclose(f);
    cal.w = W; cal.h = H;
    const std::string out = argv[6];
    try {
        {   // the reference prints an error and exits when the file is missing: here it throws
            bool threw = false;
            try { rtr::ProjectCloud bad(grid, "no_such_model.pt"); } catch (const std::exception&) { threw = true; }
            if (!threw) return 7;
        }
        {   // no model file name: the projection methods work, computeFull does not
            rtr::ProjectCloud plain(grid, "");
            Img c0, d0;
            c0.bytes.resize((size_t)W * H * 3); d0.bytes.resize((size_t)W * H * 4);
            bool threw = false;
            try { plain.computeFull(cal, E, &c0, &d0); } catch (const std::exception&) { threw = true; }
            if (!threw) return 8;
        }
        rtr::ProjectCloud pc(grid, argv[5]);
        Img rgb, depth;
        rgb.bytes.resize((size_t)W * H * 3); depth.bytes.resize((size_t)W * H * 4);
        if (pc.computeFull(cal, E, &rgb, &depth) != 1) return 3;
        dump(out + ".rgb", rgb.bytes.data(), rgb.bytes.size());
        dump(out + ".depth", depth.bytes.data(), depth.bytes.size());
        Img conly;
        conly.bytes.resize((size_t)W * H * 3);
        if (pc.computeFull(cal, E, &conly, nullptr) != 1 || conly.bytes != rgb.bytes) return 6;
        // what the model saw: run it once more by hand on the resident tensor and compare storage
        torch::Tensor in = torch::from_blob(pc.tensor(), {1, 5, H, W},
                                            torch::TensorOptions().dtype(torch::kFloat16).device(torch::kCUDA, 0));
        torch::Tensor host = in.cpu().contiguous();
        dump(out + ".tensor", host.data_ptr(), (size_t)W * H * 10);
    } catch (const std::exception& e) {
        fprintf(stderr, "%s\n", e.what());
        return 5;
    }
    return 0;
}
