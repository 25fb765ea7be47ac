// host_io.cpp -- file formats either side of the hot path:
//   PLY mesh ingest/egress  (the reference's `ply` crate is 100 % commented out, so the
//                            contract is the public PLY 1.0 format, not that crate)
//   OBJ ingest               rayrs-lib/src/wavefront_obj.rs:15-64
//   Radiance .hdr read/write what image::hdr::{HdrDecoder, HDREncoder} do at rayrs/src/main.rs:36-41, :113-121
//   PPM / PNG write          rayrs-lib/src/image.rs:193-254, rayrs/src/main.rs:104-110
// Host only; nothing here runs on the GPU.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/rayrs_hip.h"

namespace {

thread_local std::string g_io_error;

int io_fail(const std::string& msg) {
    g_io_error = msg;
    return RAYRS_IO_ERROR;
}

// ------------------------------------------------------------------- PLY

struct PlyProp {
    std::string name;
    bool is_list = false;
    int type = 0;        // scalar type (or list item type)
    int count_type = 0;  // list count type
};
struct PlyElement {
    std::string name;
    uint64_t count = 0;
    std::vector<PlyProp> props;
};

// type ids and sizes
enum { T_I8, T_U8, T_I16, T_U16, T_I32, T_U32, T_F32, T_F64, T_BAD };
int ply_type(const std::string& s) {
    if (s == "char" || s == "int8") return T_I8;
    if (s == "uchar" || s == "uint8") return T_U8;
    if (s == "short" || s == "int16") return T_I16;
    if (s == "ushort" || s == "uint16") return T_U16;
    if (s == "int" || s == "int32") return T_I32;
    if (s == "uint" || s == "uint32") return T_U32;
    if (s == "float" || s == "float32") return T_F32;
    if (s == "double" || s == "float64") return T_F64;
    return T_BAD;
}
const int TYPE_SIZE[] = {1, 1, 2, 2, 4, 4, 4, 8, 0};

double read_binary_scalar(const uint8_t* p, int type) {
    switch (type) {
        case T_I8: return (double)*reinterpret_cast<const int8_t*>(p);
        case T_U8: return (double)*p;
        case T_I16: { int16_t v; std::memcpy(&v, p, 2); return v; }
        case T_U16: { uint16_t v; std::memcpy(&v, p, 2); return v; }
        case T_I32: { int32_t v; std::memcpy(&v, p, 4); return v; }
        case T_U32: { uint32_t v; std::memcpy(&v, p, 4); return v; }
        case T_F32: { float v; std::memcpy(&v, p, 4); return v; }
        default: { double v; std::memcpy(&v, p, 8); return v; }
    }
}

}  // namespace

// Writers report a short write, a full disk or a failed flush instead of returning OK on a truncated file.
static int finish_write(FILE* f, bool ok, const char* path) {
    ok = ok && std::ferror(f) == 0;
    ok = (std::fclose(f) == 0) && ok;
    return ok ? RAYRS_OK : io_fail(std::string("write failed: ") + path);
}

// No exception crosses the C boundary: allocation failures of the loaders and writers become statuses.
#define IO_GUARDED(call)                     \
    try {                                    \
        return call;                         \
    } catch (const std::bad_alloc&) {        \
        g_io_error = "out of memory";        \
        return RAYRS_OOM;                    \
    } catch (const std::exception& e) {      \
        return io_fail(e.what());            \
    }

extern "C" {

const char* rayrs_io_last_error(void) { return g_io_error.c_str(); }

void rayrs_buffer_free(void* p) { std::free(p); }

static int ply_load_impl(const char* path, float** verts_out, uint32_t* nverts_out, uint32_t** idx_out, uint32_t* ntris_out) {
    if (!path || !verts_out || !nverts_out || !idx_out || !ntris_out) return RAYRS_INVALID_ARG;
    *verts_out = nullptr;
    *idx_out = nullptr;
    *nverts_out = *ntris_out = 0;
    std::ifstream f(path, std::ios::binary);
    if (!f) return io_fail(std::string("cannot open ") + path);
    std::string line;
    if (!std::getline(f, line) || line.substr(0, 3) != "ply") return io_fail("not a PLY file");
    int format = -1;  // 0 ascii, 1 binary little endian
    std::vector<PlyElement> elements;
    bool header_done = false;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ls(line);
        std::string word;
        ls >> word;
        if (word == "format") {
            std::string fmt;
            ls >> fmt;
            if (fmt == "ascii") format = 0;
            else if (fmt == "binary_little_endian") format = 1;
            else return io_fail("unsupported PLY format " + fmt);
        } else if (word == "element") {
            PlyElement e;
            ls >> e.name >> e.count;
            elements.push_back(e);
        } else if (word == "property") {
            if (elements.empty()) return io_fail("property before element");
            PlyProp p;
            std::string t;
            ls >> t;
            if (t == "list") {
                std::string ct, it;
                ls >> ct >> it >> p.name;
                p.is_list = true;
                p.count_type = ply_type(ct);
                p.type = ply_type(it);
                if (p.count_type == T_BAD || p.type == T_BAD) return io_fail("bad list property type");
            } else {
                p.type = ply_type(t);
                ls >> p.name;
                if (p.type == T_BAD) return io_fail("bad property type " + t);
            }
            elements.back().props.push_back(p);
        } else if (word == "end_header") {
            header_done = true;
            break;
        }  // comment / obj_info: ignored
    }
    if (!header_done || format < 0) return io_fail("truncated PLY header");

    std::vector<float> verts;
    std::vector<uint32_t> idx;
    // the rest of the file
    std::vector<uint8_t> body((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    size_t pos = 0;
    std::istringstream ascii;
    if (format == 0) ascii.str(std::string(body.begin(), body.end()));

    auto next_scalar = [&](int type, double& out) -> bool {
        if (format == 0) {
            return static_cast<bool>(ascii >> out);
        }
        const int sz = TYPE_SIZE[type];
        if (pos + (size_t)sz > body.size()) return false;
        out = read_binary_scalar(body.data() + pos, type);
        pos += (size_t)sz;
        return true;
    };

    for (const PlyElement& e : elements) {
        const bool is_vertex = e.name == "vertex";
        const bool is_face = e.name == "face";
        int ix = -1, iy = -1, iz = -1;
        if (is_vertex) {
            for (size_t k = 0; k < e.props.size(); k++) {
                if (e.props[k].name == "x") ix = (int)k;
                if (e.props[k].name == "y") iy = (int)k;
                if (e.props[k].name == "z") iz = (int)k;
            }
            if (ix < 0 || iy < 0 || iz < 0) return io_fail("vertex element without x/y/z");
            if (e.count >= (1ull << 32)) return io_fail("too many vertices");
        }
        // the header is not trusted: an element of `count` rows needs at least one byte per property and row
        // (two in ascii: a digit and a separator), so a count the body cannot hold is refused before any reserve
        {
            const uint64_t min_row = std::max<uint64_t>(e.props.size(), 1) * (format == 0 ? 2u : 1u);
            const uint64_t left = body.size() - std::min(body.size(), format == 0 ? (size_t)0 : pos);
            if (e.count > left / min_row + 1) return io_fail("element count exceeds the file size");
        }
        if (is_vertex) verts.reserve((size_t)e.count * 3);
        std::vector<double> poly;
        for (uint64_t r = 0; r < e.count; r++) {
            double xyz[3] = {0, 0, 0};
            for (size_t k = 0; k < e.props.size(); k++) {
                const PlyProp& p = e.props[k];
                if (p.is_list) {
                    double n;
                    if (!next_scalar(p.count_type, n)) return io_fail("truncated PLY body");
                    if (!(n >= 0.0 && n <= 1e6)) return io_fail("bad list length");
                    poly.clear();
                    for (int j = 0; j < (int)n; j++) {
                        double v;
                        if (!next_scalar(p.type, v)) return io_fail("truncated PLY body");
                        poly.push_back(v);
                    }
                    if (is_face && (p.name == "vertex_indices" || p.name == "vertex_index")) {
                        for (double v : poly)  // validated as read, before any narrowing cast
                            if (!(v >= 0.0 && v < 4294967296.0 && v == std::floor(v))) return io_fail("face index out of range");
                        for (size_t j = 2; j < poly.size(); j++) {  // fan triangulation, winding kept
                            idx.push_back((uint32_t)poly[0]);
                            idx.push_back((uint32_t)poly[j - 1]);
                            idx.push_back((uint32_t)poly[j]);
                        }
                    }
                } else {
                    double v;
                    if (!next_scalar(p.type, v)) return io_fail("truncated PLY body");
                    if (is_vertex) {
                        // `property double x`: the caller gets f32 vertices, so only values an f32 holds exactly
                        // pass (anything else would silently move the geometry; NaN fails the comparison too).
                        // A `float` property is an f32 by declaration: its decimal text is rounded to it.
                        const bool coord = (int)k == ix || (int)k == iy || (int)k == iz;
                        if (coord && p.type == T_F64 && (double)(float)v != v)
                            return io_fail("double vertex coordinate is not representable in f32");
                        if ((int)k == ix) xyz[0] = v;
                        if ((int)k == iy) xyz[1] = v;
                        if ((int)k == iz) xyz[2] = v;
                    }
                }
            }
            if (is_vertex) {
                for (double c : xyz) verts.push_back((float)c);
            }
        }
    }
    const uint32_t nv = (uint32_t)(verts.size() / 3);
    for (uint32_t i : idx)
        if (i >= nv) return io_fail("face index out of range");
    *verts_out = static_cast<float*>(std::malloc(std::max<size_t>(verts.size(), 1) * sizeof(float)));
    *idx_out = static_cast<uint32_t*>(std::malloc(std::max<size_t>(idx.size(), 1) * sizeof(uint32_t)));
    if (!*verts_out || !*idx_out) {
        std::free(*verts_out);
        std::free(*idx_out);
        *verts_out = nullptr;
        *idx_out = nullptr;
        return RAYRS_OOM;
    }
    std::memcpy(*verts_out, verts.data(), verts.size() * sizeof(float));
    std::memcpy(*idx_out, idx.data(), idx.size() * sizeof(uint32_t));
    *nverts_out = nv;
    *ntris_out = (uint32_t)(idx.size() / 3);
    return RAYRS_OK;
}

int rayrs_ply_load(const char* path, float** verts_out, uint32_t* nverts_out, uint32_t** idx_out, uint32_t* ntris_out) {
    try {  // nothing unwinds across the C boundary
        return ply_load_impl(path, verts_out, nverts_out, idx_out, ntris_out);
    } catch (const std::bad_alloc&) {
        g_io_error = "out of memory";
        return RAYRS_OOM;
    } catch (const std::exception& e) {
        return io_fail(e.what());
    }
}

static int rayrs_ply_save_impl(const char* path, const float* verts, uint32_t nverts, const uint32_t* idx, uint32_t ntris,
                   int binary) {
    if (!path || (!verts && nverts) || (!idx && ntris)) return RAYRS_INVALID_ARG;
    std::ofstream f(path, std::ios::binary);
    if (!f) return io_fail(std::string("cannot create ") + path);
    f << "ply\nformat " << (binary ? "binary_little_endian" : "ascii") << " 1.0\ncomment rayrs-mi355x\n"
      << "element vertex " << nverts << "\nproperty float x\nproperty float y\nproperty float z\n"
      << "element face " << ntris << "\nproperty list uchar int vertex_indices\nend_header\n";
    if (binary) {
        f.write(reinterpret_cast<const char*>(verts), (std::streamsize)nverts * 12);
        std::vector<uint8_t> rec((size_t)ntris * 13);
        for (uint32_t t = 0; t < ntris; t++) {
            rec[(size_t)t * 13] = 3;
            std::memcpy(&rec[(size_t)t * 13 + 1], idx + 3 * (size_t)t, 12);
        }
        f.write(reinterpret_cast<const char*>(rec.data()), (std::streamsize)rec.size());
    } else {
        char buf[128];
        for (uint32_t v = 0; v < nverts; v++) {
            std::snprintf(buf, sizeof buf, "%.9g %.9g %.9g\n", verts[3 * v], verts[3 * v + 1], verts[3 * v + 2]);
            f << buf;  // 9 significant digits round-trip an f32 exactly
        }
        for (uint32_t t = 0; t < ntris; t++) f << "3 " << idx[3 * t] << ' ' << idx[3 * t + 1] << ' ' << idx[3 * t + 2] << '\n';
    }
    return f ? RAYRS_OK : io_fail("write failed");
}

// ------------------------------------------------------------------- OBJ

// load_obj_file, wavefront_obj.rs:15-45: `v x y z` and `f i j k` lines only, fields split
// on single spaces, 1-based plain indices, triangles only.  Vertices stay f64.
// wavefront_obj::load_obj_file_spheres (wavefront_obj.rs:46-64): the `v` lines only, one sphere centre each; every other
// line -- `f` lines too, which load_obj_file would parse -- is skipped.  The radius is the caller's (rayrs_object_from_spheres).
static int rayrs_obj_load_spheres_impl(const char* path, double** centers_out, uint32_t* n_out) {
    if (!path || !centers_out || !n_out) return RAYRS_INVALID_ARG;
    std::ifstream f(path);
    if (!f) return io_fail(std::string("cannot open ") + path);
    std::vector<double> c;
    std::string line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::vector<std::string> v;
        size_t start = 0;
        for (;;) {  // text.split(' ')
            const size_t sp = line.find(' ', start);
            v.push_back(line.substr(start, sp == std::string::npos ? std::string::npos : sp - start));
            if (sp == std::string::npos) break;
            start = sp + 1;
        }
        if (v[0] != "v") continue;
        if (v.size() < 4) return io_fail("bad v line");  // the reference .unwrap()s
        for (int k = 1; k <= 3; k++) {
            char* end = nullptr;
            const double x = std::strtod(v[(size_t)k].c_str(), &end);
            if (end == v[(size_t)k].c_str()) return io_fail("bad number in v line");
            c.push_back(x);
        }
    }
    *centers_out = static_cast<double*>(std::malloc(std::max<size_t>(c.size(), 1) * sizeof(double)));
    if (!*centers_out) return RAYRS_OOM;
    std::memcpy(*centers_out, c.data(), c.size() * sizeof(double));
    *n_out = (uint32_t)(c.size() / 3);
    return RAYRS_OK;
}

static int rayrs_obj_load_impl(const char* path, double** verts_out, uint32_t* nverts_out, uint32_t** idx_out,
                   uint32_t* ntris_out) {
    if (!path || !verts_out || !nverts_out || !idx_out || !ntris_out) return RAYRS_INVALID_ARG;
    std::ifstream f(path);
    if (!f) return io_fail(std::string("cannot open ") + path);
    std::vector<double> verts;
    std::vector<uint32_t> idx;
    std::string line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::vector<std::string> v;
        size_t start = 0;
        for (;;) {  // text.split(' ')
            const size_t sp = line.find(' ', start);
            v.push_back(line.substr(start, sp == std::string::npos ? std::string::npos : sp - start));
            if (sp == std::string::npos) break;
            start = sp + 1;
        }
        if (v[0] == "v") {
            if (v.size() < 4) return io_fail("bad v line");  // the reference .unwrap()s
            for (int k = 1; k <= 3; k++) {
                char* end = nullptr;
                const double x = std::strtod(v[(size_t)k].c_str(), &end);
                if (end == v[(size_t)k].c_str()) return io_fail("bad number in v line");
                verts.push_back(x);
            }
        } else if (v[0] == "f") {
            if (v.size() < 4) return io_fail("bad f line");
            for (int k = 1; k <= 3; k++) {
                char* end = nullptr;
                const unsigned long i = std::strtoul(v[(size_t)k].c_str(), &end, 10);
                if (end == v[(size_t)k].c_str() || *end != '\0' || i == 0) return io_fail("bad index in f line");
                idx.push_back((uint32_t)(i - 1));
            }
        }
    }
    const uint32_t nv = (uint32_t)(verts.size() / 3);
    for (uint32_t i : idx)
        if (i >= nv) return io_fail("face index out of range");
    *verts_out = static_cast<double*>(std::malloc(std::max<size_t>(verts.size(), 1) * sizeof(double)));
    *idx_out = static_cast<uint32_t*>(std::malloc(std::max<size_t>(idx.size(), 1) * sizeof(uint32_t)));
    if (!*verts_out || !*idx_out) return RAYRS_OOM;
    std::memcpy(*verts_out, verts.data(), verts.size() * sizeof(double));
    std::memcpy(*idx_out, idx.data(), idx.size() * sizeof(uint32_t));
    *nverts_out = nv;
    *ntris_out = (uint32_t)(idx.size() / 3);
    return RAYRS_OK;
}

// --------------------------------------------------------- Radiance .hdr

// Reads a Radiance RGBE picture (-Y h +X w, flat or new-style RLE scanlines) into w*h RGB f32,
// row-major, top row first: what HdrDecoder::read_image_hdr yields (main.rs:36-41).
static int rayrs_hdr_load_impl(const char* path, float** rgb_out, uint32_t* w_out, uint32_t* h_out) {
    if (!path || !rgb_out || !w_out || !h_out) return RAYRS_INVALID_ARG;
    *rgb_out = nullptr;
    FILE* f = std::fopen(path, "rb");
    if (!f) return io_fail(std::string("cannot open ") + path);
    std::vector<uint8_t> data;
    {
        uint8_t buf[65536];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + n);
        std::fclose(f);
    }
    size_t pos = 0;
    auto getline = [&](std::string& out) -> bool {
        out.clear();
        while (pos < data.size()) {
            const char c = (char)data[pos++];
            if (c == '\n') return true;
            out.push_back(c);
        }
        return !out.empty();
    };
    std::string line;
    if (!getline(line) || (line.substr(0, 2) != "#?")) return io_fail("not a Radiance file");
    bool fmt_ok = false;
    while (getline(line)) {
        if (line.empty()) break;
        if (line.find("FORMAT=32-bit_rle_rgbe") != std::string::npos) fmt_ok = true;
    }
    if (!fmt_ok) return io_fail("unsupported Radiance FORMAT");
    if (!getline(line)) return io_fail("missing resolution line");
    int h = 0, w = 0;
    if (std::sscanf(line.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0) return io_fail("unsupported orientation");
    float* rgb = static_cast<float*>(std::malloc((size_t)w * h * 3 * sizeof(float)));
    if (!rgb) return RAYRS_OOM;
    std::vector<uint8_t> scan((size_t)w * 4);
    for (int y = 0; y < h; y++) {
        if (pos + 4 > data.size()) { std::free(rgb); return io_fail("truncated pixel data"); }
        if (w >= 8 && w < 32768 && data[pos] == 2 && data[pos + 1] == 2 && ((data[pos + 2] << 8) | data[pos + 3]) == w) {
            pos += 4;
            for (int c = 0; c < 4; c++) {
                int x = 0;
                while (x < w) {
                    if (pos >= data.size()) { std::free(rgb); return io_fail("truncated RLE data"); }
                    int n = data[pos++];
                    if (n > 128) {
                        n -= 128;
                        if (pos >= data.size() || x + n > w) { std::free(rgb); return io_fail("bad RLE run"); }
                        const uint8_t v = data[pos++];
                        for (int k = 0; k < n; k++) scan[(size_t)(x++) * 4 + c] = v;
                    } else {
                        if (n == 0 || pos + (size_t)n > data.size() || x + n > w) { std::free(rgb); return io_fail("bad RLE literal"); }
                        for (int k = 0; k < n; k++) scan[(size_t)(x++) * 4 + c] = data[pos++];
                    }
                }
            }
        } else {
            if (pos + (size_t)w * 4 > data.size()) { std::free(rgb); return io_fail("truncated flat data"); }
            std::memcpy(scan.data(), &data[pos], (size_t)w * 4);
            pos += (size_t)w * 4;
        }
        for (int x = 0; x < w; x++) {
            const uint8_t* p = &scan[(size_t)x * 4];
            float* o = rgb + ((size_t)y * w + x) * 3;
            if (p[3] == 0) {
                o[0] = o[1] = o[2] = 0.f;
            } else {
                const float s = std::ldexp(1.0f, (int)p[3] - (128 + 8));
                o[0] = p[0] * s, o[1] = p[1] * s, o[2] = p[2] * s;
            }
        }
    }
    *rgb_out = rgb;
    *w_out = (uint32_t)w;
    *h_out = (uint32_t)h;
    return RAYRS_OK;
}

// HDREncoder::encode (main.rs:113-121): flat (non-RLE) RGBE scanlines.
static int rayrs_hdr_save_impl(const char* path, const float* rgb, uint32_t w, uint32_t h) {
    if (!path || !rgb || !w || !h) return RAYRS_INVALID_ARG;
    FILE* f = std::fopen(path, "wb");
    if (!f) return io_fail(std::string("cannot create ") + path);
    bool ok = std::fprintf(f, "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %u +X %u\n", h, w) > 0;
    std::vector<uint8_t> scan((size_t)w * 4);
    for (uint32_t y = 0; y < h; y++) {
        for (uint32_t x = 0; x < w; x++) {
            const float* p = rgb + ((size_t)y * w + x) * 3;
            float m = std::fmax(p[0], std::fmax(p[1], p[2]));
            uint8_t* o = &scan[(size_t)x * 4];
            if (!(m > 1e-32f)) {
                o[0] = o[1] = o[2] = o[3] = 0;
            } else {
                int e;
                const float s = std::frexp(m, &e) * 256.0f / m;
                o[0] = (uint8_t)(p[0] > 0 ? p[0] * s : 0);
                o[1] = (uint8_t)(p[1] > 0 ? p[1] * s : 0);
                o[2] = (uint8_t)(p[2] > 0 ? p[2] * s : 0);
                o[3] = (uint8_t)(e + 128);
            }
        }
        ok = ok && std::fwrite(scan.data(), 1, scan.size(), f) == scan.size();
    }
    return finish_write(f, ok, path);
}

// ------------------------------------------------------------- LDR output

// Image::to_raw_bytes(gamma), image.rs:193-222: clip(0,1).powf(gamma), (255.99 * x) as u8;
// counts clamped / NaN / negative pixels as the reference prints them.
int rayrs_image_to_bytes(const float* rgb, uint32_t w, uint32_t h, double gamma, uint8_t* out, uint64_t counts[3]) {
    if (!rgb || !out) return RAYRS_INVALID_ARG;
    uint64_t bright = 0, nans = 0, neg = 0;
    for (size_t i = 0; i < (size_t)w * h; i++) {
        const double v[3] = {rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]};
        if (std::isnan(v[0]) || std::isnan(v[1]) || std::isnan(v[2])) nans++;
        if (v[0] < 0. || v[1] < 0. || v[2] < 0.) neg++;
        if (v[0] > 1. || v[1] > 1. || v[2] > 1.) bright++;
        for (int c = 0; c < 3; c++) {
            const double clipped = std::fmax(std::fmin(v[c], 1.0), 0.0);  // x.min(max).max(min), vecmath.rs:388-396
            const double b = 255.99 * std::pow(clipped, gamma);
            out[3 * i + (size_t)c] = std::isnan(b) ? 0 : (b >= 255.0 ? 255 : (b <= 0.0 ? 0 : (uint8_t)b));  // `as u8` saturates
        }
    }
    if (counts) counts[0] = bright, counts[1] = nans, counts[2] = neg;
    return RAYRS_OK;
}

// Image::save(PpmBinary), image.rs:248-251
int rayrs_ppm_save(const char* path, const uint8_t* bytes, uint32_t w, uint32_t h) {
    if (!path || !bytes) return RAYRS_INVALID_ARG;
    FILE* f = std::fopen(path, "wb");
    if (!f) return io_fail(std::string("cannot create ") + path);
    bool ok = std::fprintf(f, "P6\n%u %u\n255\n", w, h) > 0;
    ok = ok && std::fwrite(bytes, 1, (size_t)w * h * 3, f) == (size_t)w * h * 3;
    return finish_write(f, ok, path);
}

// image::save_buffer(.., ColorType::Rgb8) to PNG (main.rs:104-110): 8-bit RGB, zlib "stored" blocks.
static int rayrs_png_save_impl(const char* path, const uint8_t* bytes, uint32_t w, uint32_t h) {
    if (!path || !bytes || !w || !h) return RAYRS_INVALID_ARG;
    static uint32_t crc_table[256];
    static bool have = false;
    if (!have) {
        for (uint32_t n = 0; n < 256; n++) {
            uint32_t c = n;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
            crc_table[n] = c;
        }
        have = true;
    }
    auto crc = [&](const uint8_t* p, size_t n, uint32_t c) {
        for (size_t i = 0; i < n; i++) c = crc_table[(c ^ p[i]) & 0xffu] ^ (c >> 8);
        return c;
    };
    FILE* f = std::fopen(path, "wb");
    if (!f) return io_fail(std::string("cannot create ") + path);
    auto be32 = [](uint8_t* p, uint32_t v) { p[0] = v >> 24, p[1] = v >> 16, p[2] = v >> 8, p[3] = v; };
    auto chunk = [&](const char* type, const std::vector<uint8_t>& payload) {
        uint8_t len[4];
        be32(len, (uint32_t)payload.size());
        std::fwrite(len, 1, 4, f);
        std::fwrite(type, 1, 4, f);
        if (!payload.empty()) std::fwrite(payload.data(), 1, payload.size(), f);
        uint32_t c = crc(reinterpret_cast<const uint8_t*>(type), 4, 0xffffffffu);
        c = crc(payload.data(), payload.size(), c) ^ 0xffffffffu;
        uint8_t cb[4];
        be32(cb, c);
        std::fwrite(cb, 1, 4, f);
    };
    const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    std::fwrite(sig, 1, 8, f);
    std::vector<uint8_t> ihdr(13);
    be32(&ihdr[0], w);
    be32(&ihdr[4], h);
    ihdr[8] = 8, ihdr[9] = 2, ihdr[10] = 0, ihdr[11] = 0, ihdr[12] = 0;
    chunk("IHDR", ihdr);
    // raw scanlines with filter byte 0
    std::vector<uint8_t> raw((size_t)h * ((size_t)w * 3 + 1));
    for (uint32_t y = 0; y < h; y++) {
        raw[(size_t)y * ((size_t)w * 3 + 1)] = 0;
        std::memcpy(&raw[(size_t)y * ((size_t)w * 3 + 1) + 1], bytes + (size_t)y * w * 3, (size_t)w * 3);
    }
    std::vector<uint8_t> z;
    z.push_back(0x78);
    z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t off = 0; off < raw.size();) {
        const size_t n = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + n == raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xff));
        z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xff));
        z.push_back((uint8_t)((~n >> 8) & 0xff));
        z.insert(z.end(), raw.begin() + (std::ptrdiff_t)off, raw.begin() + (std::ptrdiff_t)(off + n));
        for (size_t i = 0; i < n; i++) {
            a = (a + raw[off + i]) % 65521u;
            b = (b + a) % 65521u;
        }
        off += n;
    }
    uint8_t ad[4];
    be32(ad, (b << 16) | a);
    z.insert(z.end(), ad, ad + 4);
    chunk("IDAT", z);
    chunk("IEND", {});
    return finish_write(f, true, path);
}

int rayrs_ply_save(const char* path, const float* verts, uint32_t nverts, const uint32_t* idx, uint32_t ntris,
                   int binary) { IO_GUARDED(rayrs_ply_save_impl(path, verts, nverts, idx, ntris, binary)); }
int rayrs_obj_load(const char* path, double** verts_out, uint32_t* nverts_out, uint32_t** idx_out,
                   uint32_t* ntris_out) { IO_GUARDED(rayrs_obj_load_impl(path, verts_out, nverts_out, idx_out, ntris_out)); }
int rayrs_obj_load_spheres(const char* path, double** centers_out, uint32_t* n_out) { IO_GUARDED(rayrs_obj_load_spheres_impl(path, centers_out, n_out)); }
int rayrs_hdr_load(const char* path, float** rgb_out, uint32_t* w_out, uint32_t* h_out) { IO_GUARDED(rayrs_hdr_load_impl(path, rgb_out, w_out, h_out)); }
int rayrs_hdr_save(const char* path, const float* rgb, uint32_t w, uint32_t h) { IO_GUARDED(rayrs_hdr_save_impl(path, rgb, w, h)); }
int rayrs_png_save(const char* path, const uint8_t* bytes, uint32_t w, uint32_t h) { IO_GUARDED(rayrs_png_save_impl(path, bytes, w, h)); }

}  // extern "C"
