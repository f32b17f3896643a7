// render_trajectory -- the reference's example application on top of rtr::ProjectCloud
// (reference: example/render_trajectory/main.cpp:67-101): load a cloud, a calibration and a
// trajectory, replay the trajectory through computeRGBD / computeFilteredRGBD.
//
//   render_trajectory <pcl_path> <trajectory_path> <calibration_file> [--filtered] [--out DIR] [--every N]
//
// Plain C++17 + the C ABI (no OpenCV / glm / tinyply in this image), so the small host-side
// pieces the reference takes from those libraries are restated here:
//   * binary little-endian PLY with float x,y,z + uchar red,green,blue, colours stored B,G,R
//     (cloudreader.cpp:122-177)
//   * the loader's grid cache `pcd.oct` (OctreeGrid::readOctreeBinary, Octreegrid.h:83-114; what
//     CloudReader::loadCloud reads instead of the cloud when ~/.pcl_cache holds one, cloudreader.cpp:182-190):
//     a <pcl_path> ending in .oct is read as such, blocks and their order kept
//   * calibration: COLMAP cameras.txt (OPENCV model, floats) or the 6-line txt
//     (CameraCalibration.cpp:101-209)
//   * trajectory: COLMAP images.txt (README.md:92, world->camera) when the file is named
//     images.txt, else `timestamp tx ty tz qx qy qz qw` camera-to-world (main.cpp:20-65),
//     inverted here as main.cpp:96 does with cv::Matx44d::inv (rigid inverse [R^T, -R^T t]).
// Output instead of cv::imshow: optional frame_<k>.ppm / .pfm files and an fps line.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "rtr_project_cloud.hpp"

namespace {

struct P3 { float x, y, z; };
struct C3 { unsigned char v[3]; unsigned char operator[](int i) const { return v[i]; } };
struct Block { std::vector<P3> positions; std::vector<C3> colors; };
struct K33 { double m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; double operator()(int r, int c) const { return m[3 * r + c]; } };
struct M44 { double m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; double operator()(int r, int c) const { return m[4 * r + c]; } };
struct Calibration {
    K33 K; int width = 640, height = 480;  // CameraCalibration.cpp:5-10
    int getWidth() const { return width; }
    int getHeight() const { return height; }
    K33 getIntrinsicsMatrix() const { return K; }
};
struct Image {
    std::vector<unsigned char> bytes;
    template <class T> T* ptr() { return reinterpret_cast<T*>(bytes.data()); }
};

bool ends_with(const std::string& s, const std::string& t) { return s.size() >= t.size() && s.compare(s.size() - t.size(), t.size(), t) == 0; }

bool load_calibration(const std::string& file, Calibration& cal) {
    std::ifstream ifs(file);
    if (!ifs) { std::cerr << "Failed to open calibration file " << file << "\n"; return false; }
    if (ends_with(file, "cameras.txt")) {  // CameraCalibration.cpp:103-158
        std::string line;
        while (std::getline(ifs, line)) {
            if (line.empty() || line[0] == '#') continue;
            std::istringstream iss(line);
            int id; std::string model; float fx, fy, cx, cy;
            iss >> id >> model >> cal.width >> cal.height;
            if (model != "OPENCV" && model != "OPENCV_FISHEYE") { std::cerr << "Unsupported camera model: " << model << "\n"; return false; }
            iss >> fx >> fy >> cx >> cy;  // parsed as float, like the reference
            cal.K = K33();
            cal.K.m[0] = fx; cal.K.m[4] = fy; cal.K.m[2] = cx; cal.K.m[5] = cy;
            return true;
        }
        return false;
    }
    ifs >> cal.width >> cal.height;  // CameraCalibration.cpp:160-170
    for (double& v : cal.K.m) ifs >> v;
    return static_cast<bool>(ifs);
}

M44 pose_from_quat(double qw, double qx, double qy, double qz, double tx, double ty, double tz) {
    double n = std::sqrt(qw * qw + qx * qx + qy * qy + qz * qz);  // cv::Quatd::normalize (main.cpp:38)
    double w = qw / n, x = qx / n, y = qy / n, z = qz / n;
    M44 M;
    const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                         2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                         2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M.m[4 * r + c] = R[3 * r + c];
    M.m[3] = tx; M.m[7] = ty; M.m[11] = tz;
    return M;
}

M44 rigid_inverse(const M44& A) {
    M44 B;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) B.m[4 * r + c] = A.m[4 * c + r];
    for (int r = 0; r < 3; ++r) B.m[4 * r + 3] = -(B.m[4 * r] * A.m[3] + B.m[4 * r + 1] * A.m[7] + B.m[4 * r + 2] * A.m[11]);
    return B;
}

std::vector<M44> load_trajectory(const std::string& file) {
    std::vector<M44> out;
    std::ifstream ifs(file);
    const bool colmap = ends_with(file, "images.txt");
    std::string line;
    while (std::getline(ifs, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream iss(line);
        if (colmap) {  // IMAGE_ID QW QX QY QZ TX TY TZ CAMERA_ID NAME, then a POINTS2D line
            long id; double qw, qx, qy, qz, tx, ty, tz; int cam; std::string name;
            if (!(iss >> id >> qw >> qx >> qy >> qz >> tx >> ty >> tz >> cam >> name)) continue;
            out.push_back(pose_from_quat(qw, qx, qy, qz, tx, ty, tz));
            std::getline(ifs, line);  // skip the (possibly empty) POINTS2D line
        } else {       // timestamp tx ty tz qx qy qz qw, camera-to-world (main.cpp:32)
            double ts, tx, ty, tz, qx, qy, qz, qw;
            if (!(iss >> ts >> tx >> ty >> tz >> qx >> qy >> qz >> qw)) continue;
            out.push_back(rigid_inverse(pose_from_quat(qw, qx, qy, qz, tx, ty, tz)));  // main.cpp:96 pose.inv()
        }
    }
    return out;
}

bool load_ply(const std::string& file, std::map<int, Block>& grid) {
    std::ifstream ss(file, std::ios::binary);
    if (!ss) { std::cerr << "Failed to open file: " << file << "\n"; return false; }
    std::string line; size_t n = 0; bool le = false; std::vector<std::string> props;
    std::getline(ss, line);
    if (line.rfind("ply", 0) != 0) return false;
    while (std::getline(ss, line)) {
        std::istringstream iss(line); std::string a, b, c;
        iss >> a;
        if (a == "format") { iss >> b; le = b == "binary_little_endian"; }
        else if (a == "element") { iss >> b >> n; if (b != "vertex") n = n; }
        else if (a == "property") { iss >> b >> c; props.push_back(b + " " + c); }
        else if (a == "end_header") break;
    }
    const std::vector<std::string> want = {"float x", "float y", "float z", "uchar red", "uchar green", "uchar blue"};
    if (!le || props != want) { std::cerr << "only binary_little_endian float x,y,z + uchar red,green,blue is supported here\n"; return false; }
    std::vector<unsigned char> buf(n * 15);
    ss.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)buf.size());
    if ((size_t)ss.gcount() != buf.size()) return false;
    Block& b = grid[0];  // one block: the projector does not depend on the block structure
    b.positions.resize(n); b.colors.resize(n);
    for (size_t i = 0; i < n; ++i) {
        std::memcpy(&b.positions[i], &buf[15 * i], 12);
        b.colors[i] = C3{{buf[15 * i + 14], buf[15 * i + 13], buf[15 * i + 12]}};  // B,G,R (cloudreader.cpp:168)
    }
    return true;
}

// OctreeGrid::readOctreeBinary (Octreegrid.h:83-114): numBlocksX/Y/Z and the block count as four ints, then per block
// its key (int), the point count (size_t), n x 3 floats, n x 3 colour bytes (as stored: B,G,R) and the block's
// bbMin / bbMax (3 floats each; the projector never reads them).  Blocks keep the file's order.
bool load_oct(const std::string& file, std::map<int, Block>& grid, size_t& points) {
    std::ifstream ss(file, std::ios::binary);
    if (!ss) { std::cerr << "Failed to open file: " << file << "\n"; return false; }
    int dims[4];
    ss.read(reinterpret_cast<char*>(dims), sizeof dims);
    if (!ss || dims[3] < 0) return false;
    points = 0;
    for (int b = 0; b < dims[3]; ++b) {
        int key; unsigned long long n;
        ss.read(reinterpret_cast<char*>(&key), 4);
        ss.read(reinterpret_cast<char*>(&n), 8);
        if (!ss || n > (1ull << 32)) return false;
        Block& blk = grid[key];  // (a std::map: key order; the frame does not depend on the point order)
        const size_t at = blk.positions.size();
        blk.positions.resize(at + n); blk.colors.resize(at + n);
        ss.read(reinterpret_cast<char*>(blk.positions.data() + at), (std::streamsize)(n * 12));
        ss.read(reinterpret_cast<char*>(blk.colors.data() + at), (std::streamsize)(n * 3));
        float bb[6];
        ss.read(reinterpret_cast<char*>(bb), sizeof bb);
        if (!ss) return false;
        points += n;
    }
    return true;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 4) {  // main.cpp:69-74
        std::cerr << "Missing required parameter:\nrender_trajectory pcl_path trajectory_path calibration_file"
                     " [--filtered] [--out DIR] [--every N]\n";
        return -1;
    }
    bool filtered = false; std::string out; int every = 100;
    for (int i = 4; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--filtered") filtered = true;
        else if (a == "--out" && i + 1 < argc) out = argv[++i];
        else if (a == "--every" && i + 1 < argc) every = std::atoi(argv[++i]);
    }
    Calibration calibration;
    if (!load_calibration(argv[3], calibration)) return -1;
    std::map<int, Block> grid;
    size_t points = 0;
    if (ends_with(argv[1], ".oct")) {
        if (!load_oct(argv[1], grid, points)) return -1;
    } else {
        if (!load_ply(argv[1], grid)) return -1;
        points = grid[0].positions.size();
    }
    std::cout << "Loaded " << points << " points" << std::endl;  // main.cpp:86
    std::vector<M44> trajectory = load_trajectory(argv[2]);
    const int W = calibration.getWidth(), H = calibration.getHeight();
    try {
        rtr::ProjectCloud projector(grid, "");  // main.cpp:87
        Image rgb, depth;
        rgb.bytes.resize((size_t)W * H * 3); depth.bytes.resize((size_t)W * H * 4);  // main.cpp:93-94
        auto t0 = std::chrono::steady_clock::now();
        int k = 0;
        for (const M44& pose : trajectory) {
            int rc = filtered ? projector.computeFilteredRGBD(calibration, pose, &rgb, &depth)
                              : projector.computeRGBD(calibration, pose, &rgb, &depth);  // main.cpp:96
            if (rc != 1) return -2;
            if (!out.empty() && k % every == 0) {
                std::ofstream f(out + "/frame_" + std::to_string(k + 1) + ".ppm", std::ios::binary);
                f << "P6\n" << W << " " << H << "\n255\n";
                for (size_t p = 0; p < (size_t)W * H; ++p) { const unsigned char* c = &rgb.bytes[3 * p]; const char o[3] = {(char)c[2], (char)c[1], (char)c[0]}; f.write(o, 3); }
                std::ofstream g(out + "/frame_" + std::to_string(k + 1) + ".pfm", std::ios::binary);
                g << "Pf\n" << W << " " << H << "\n-1.0\n";
                for (int y = H - 1; y >= 0; --y) g.write(reinterpret_cast<const char*>(depth.ptr<float>() + (size_t)y * W), W * 4);
            }
            ++k;
        }
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("frames %d  time %.3f s  %.1f fps  (%dx%d, %s, with D2H)\n", k, dt, k / dt, W, H, filtered ? "filtered" : "projection");
    } catch (const std::exception& e) {
        std::cerr << e.what() << "\n";
        return -3;
    }
    return 0;
}
