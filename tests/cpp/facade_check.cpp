// Host-side C++ check of include/rtr_project_cloud.hpp over the C ABI: builds with plain g++
// (no HIP headers), links librtr_hip.so.  The stand-in types below are TEST INPUT TYPES with
// the members the facade uses (the image has no OpenCV); they are not a build of the reference.
//   facade_check <cloud.bin> <W> <H> <K9+E16 doubles .bin> <out_prefix>
// cloud.bin: u64 n, n*(3 f32), n*(3 u8).  Writes <out>.rgb/.depth/.frgb/.fdepth/.tensor.
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
    if (argc < 6) { fprintf(stderr, "usage\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    unsigned long long n = 0;
    if (!f || fread(&n, 8, 1, f) != 1) return 2;
    std::map<int, Block> grid;  // two blocks, like a (tiny) OctreeGrid
    std::vector<P3> pts(n); std::vector<C3> cols(n);
    if (fread(pts.data(), 12, n, f) != n || fread(cols.data(), 3, n, f) != n) return 2;
    fclose(f);
    for (unsigned long long i = 0; i < n; ++i) {
        Block& b = grid[i < n / 2 ? 0 : 1];
        b.positions.push_back(pts[i]); b.colors.push_back(cols[i]);
    }
    int W = atoi(argv[2]), H = atoi(argv[3]);
    Calib cal; M44 E;
    f = fopen(argv[4], "rb");
    if (!f || fread(cal.K.m, 8, 9, f) != 9 || fread(E.m, 8, 16, f) != 16) return 2;
    fclose(f);
    cal.w = W; cal.h = H;
    std::string out = argv[5];
    try {
        rtr::ProjectCloud pc(grid, "");
        Img rgb, depth;
        rgb.bytes.resize((size_t)W * H * 3); depth.bytes.resize((size_t)W * H * 4);
        if (pc.computeRGBD(cal, E, (Img*)nullptr, (Img*)nullptr) != -1) return 3;
        if (pc.computeRGBD(cal, E, nullptr, nullptr) != -1) return 3;
        Img donly;
        donly.bytes.resize((size_t)W * H * 4);
        if (pc.computeRGBD(cal, E, nullptr, &donly) != 1) return 3;  // depth-only call of cloudreader.cpp:246
        if (pc.computeRGBD(cal, E, &rgb, &depth) != 1) return 3;
        if (donly.bytes != depth.bytes) return 6;
        dump(out + ".rgb", rgb.bytes.data(), rgb.bytes.size());
        dump(out + ".depth", depth.bytes.data(), depth.bytes.size());
        if (pc.computeFilteredRGBD(cal, E, &rgb, &depth) != 1) return 3;
        dump(out + ".frgb", rgb.bytes.data(), rgb.bytes.size());
        dump(out + ".fdepth", depth.bytes.data(), depth.bytes.size());
        std::vector<unsigned char> t((size_t)W * H * 10);
        if (rtr_download_buffer(pc.context(), RTR_BUF_TENSOR, t.data(), t.size()) != RTR_OK) return 4;
        dump(out + ".tensor", t.data(), t.size());
    } catch (const std::exception& e) {
        fprintf(stderr, "%s\n", e.what());
        return 5;
    }
    return 0;
}
