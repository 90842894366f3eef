#include <cstdio>
#include <cstdint>
#include <algorithm>
#include "file_manager.h"

#include <fstream>
#include <sstream>

namespace ptmi {

std::map<std::string, Material> loadMTL(const std::string& filename) {
    std::map<std::string, Material> table;
    std::ifstream in(filename);
    if (!in.is_open()) return table;            // reference warns and carries on with no materials (:42-45)

    std::string name, line;
    Material mat;
    auto flush = [&] { if (!name.empty()) table[name] = mat; };
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string key;
        ls >> key;
        if (key == "newmtl") {
            flush();
            ls >> name;
            mat = Material();
        } else if (key == "Kd" || key == "Ke") {
            float r = 0, g = 0, b = 0;
            ls >> r >> g >> b;
            (key == "Kd" ? mat.bsdf : mat.Le) = mk3(r, g, b);
        }
        // Ks / Ns / Ni / d / illum / Ka are not read by the reference
    }
    flush();
    return table;
}

namespace {

// One face-vertex token: "v", "v/vt", "v//vn" or "v/vt/vn" (file_manager.h:162-190).
// Returns false when the token does not start with an integer (e.g. the "#" and
// "Top" of a trailing comment): the reference warns and skips just that token.
bool parseFaceToken(const std::string& token, size_t& v, size_t& vn) {
    std::stringstream ts(token);
    v = 0; vn = 0;
    size_t vt = 0;
    char slash;
    if (!(ts >> v)) return false;
    if (ts.peek() == '/') {
        ts >> slash;
        if (ts.peek() == '/') {
            ts >> slash;
            ts >> vn;
        } else if (ts >> vt) {
            if (ts.peek() == '/') { ts >> slash; ts >> vn; }
        }
    }
    if (ts.fail() && vn != 0) vn = 0;
    return true;
}

}  // namespace

bool loadOBJ(const std::string& obj_filename, std::vector<Primitive>& out, std::string* warnings) {
    out.clear();
    std::ifstream in(obj_filename);
    if (!in.is_open()) return false;

    std::string dir;
    const size_t cut = obj_filename.find_last_of("/\\");
    if (cut != std::string::npos) dir = obj_filename.substr(0, cut + 1);

    std::vector<f3> positions, normals;
    std::map<std::string, Material> materials;
    Material active;
    std::ostringstream warn;

    std::string line;
    for (int line_no = 1; std::getline(in, line); ++line_no) {
        // comment, object and smoothing-group lines are dropped on their FIRST character (:120)
        if (line.empty() || line[0] == '#' || line[0] == 'o' || line[0] == 's') continue;
        std::istringstream ls(line);
        std::string key;
        ls >> key;

        if (key == "v" || key == "vn") {
            float x, y, z;
            if (!(ls >> x >> y >> z)) { warn << "line " << line_no << ": malformed " << key << "\n"; continue; }
            if (key == "v") positions.push_back(mk3(x, y, z));
            else normals.push_back(unit_vector(mk3(x, y, z)));          // normals are normalised on read (:141)
        } else if (key == "mtllib") {
            std::string mtl;
            ls >> mtl;
            materials = loadMTL(dir + mtl);
        } else if (key == "usemtl") {
            std::string mtl;
            ls >> mtl;
            auto it = materials.find(mtl);
            if (it != materials.end()) active = it->second;
            else { warn << "material '" << mtl << "' not found, using default\n"; active = Material(); }
        } else if (key == "f") {
            std::vector<size_t> vi, ni;
            std::string token;
            while (ls >> token) {
                size_t v, vn;
                if (!parseFaceToken(token, v, vn)) { warn << "line " << line_no << ": skipped token '" << token << "'\n"; continue; }
                vi.push_back(v); ni.push_back(vn);
            }
            const size_t arity = vi.size();
            if (arity != 3 && arity != 4) { warn << "line " << line_no << ": face with " << arity << " vertices dropped\n"; continue; }
            bool ok = true;
            for (size_t k = 0; k < arity; k++) ok = ok && vi[k] != 0 && vi[k] <= positions.size();   // 1-based, no relative indices
            if (!ok) { warn << "line " << line_no << ": invalid vertex index\n"; continue; }
            const bool has_vn = ni[0] != 0 && ni[0] <= normals.size();   // only the FIRST corner's vn is used (:207-211, :236-238)
            Primitive p;
            if (arity == 3) {
                const f3 a = positions[vi[0] - 1], b = positions[vi[1] - 1], c = positions[vi[2] - 1];
                p = has_vn ? Primitive::triangle(a, b, c, active.bsdf, normals[ni[0] - 1])
                           : Primitive::triangle(a, b, c, active.bsdf);
            } else {
                p = Primitive::quad(positions[vi[0] - 1], positions[vi[1] - 1], positions[vi[2] - 1], positions[vi[3] - 1], active.bsdf);
                if (has_vn) p.normal = normals[ni[0] - 1];
            }
            p.Le = active.Le;
            out.push_back(p);
        }
        // vt, g and anything else: ignored
    }
    if (warnings) *warnings = warn.str();
    return !out.empty();
}

std::vector<Primitive> convertQuadsToTriangles(const std::vector<Primitive>& primitives) {
    std::vector<Primitive> tris;
    tris.reserve(primitives.size() * 2);
    for (const Primitive& p : primitives) {
        if (p.type != PRIM_QUAD) { tris.push_back(p); continue; }
        // split along the v00-v11 diagonal; the 4-argument Triangle constructor recomputes a GEOMETRIC
        // normal, so an OBJ vn override on the quad is lost here (application_state.h:337, 347)
        Primitive t1 = Primitive::triangle(p.v[0], p.v[1], p.v[2], p.bsdf);
        Primitive t2 = Primitive::triangle(p.v[0], p.v[2], p.v[3], p.bsdf);
        t1.Le = t2.Le = p.Le;
        tris.push_back(t1); tris.push_back(t2);
    }
    return tris;
}

std::vector<Primitive> subdivide_primitives(const std::vector<Primitive>& prims, int num_subdivisions) {
    std::vector<Primitive> cur = prims;
    auto mid = [](f3 a, f3 b) { return 0.5f * (a + b); };
    for (int level = 0; level < num_subdivisions; ++level) {
        std::vector<Primitive> next;
        next.reserve(cur.size() * 4);
        for (const Primitive& p : cur) {
            Primitive kids[4];
            if (p.type == PRIM_TRIANGLE) {             // form_factors.h:479-499
                const f3 m0 = mid(p.v[0], p.v[1]), m1 = mid(p.v[1], p.v[2]), m2 = mid(p.v[2], p.v[0]);
                kids[0] = Primitive::triangle(p.v[0], m0, m2, p.bsdf);
                kids[1] = Primitive::triangle(m0, p.v[1], m1, p.bsdf);
                kids[2] = Primitive::triangle(m1, p.v[2], m2, p.bsdf);
                kids[3] = Primitive::triangle(m0, m1, m2, p.bsdf);
            } else {                                   // form_factors.h:501-522
                const f3 m01 = mid(p.v[0], p.v[1]), m12 = mid(p.v[1], p.v[2]);
                const f3 m23 = mid(p.v[2], p.v[3]), m30 = mid(p.v[3], p.v[0]);
                const f3 c = 0.25f * (p.v[0] + p.v[1] + p.v[2] + p.v[3]);
                kids[0] = Primitive::quad(p.v[0], m01, c, m30, p.bsdf);
                kids[1] = Primitive::quad(m01, p.v[1], m12, c, p.bsdf);
                kids[2] = Primitive::quad(c, m12, p.v[2], m23, p.bsdf);
                kids[3] = Primitive::quad(m30, c, m23, p.v[3], p.bsdf);
            }
            for (Primitive& k : kids) { k.Le = p.Le; next.push_back(k); }
        }
        cur.swap(next);
    }
    return cur;
}

namespace {
uint32_t crc32_update(uint32_t crc, const unsigned char* p, size_t n) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1; table[i] = c; }
        ready = true;
    }
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
    return crc;
}
void put_be32(std::vector<unsigned char>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
void put_chunk(std::vector<unsigned char>& png, const char type[4], const std::vector<unsigned char>& data) {
    put_be32(png, (uint32_t)data.size());
    const size_t start = png.size();
    png.insert(png.end(), type, type + 4);
    png.insert(png.end(), data.begin(), data.end());
    put_be32(png, crc32_update(0xffffffffu, png.data() + start, png.size() - start) ^ 0xffffffffu);
}
}  // namespace

bool writePNG(const std::string& path, int width, int height, const unsigned char* rgb) {
    if (width <= 0 || height <= 0 || !rgb) return false;
    const size_t stride = (size_t)width * 3;
    std::vector<unsigned char> raw;                             // filter byte 0 + pixels, top row first
    raw.reserve((stride + 1) * (size_t)height);
    for (int y = height - 1; y >= 0; y--) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb + (size_t)y * stride, rgb + (size_t)(y + 1) * stride);
    }
    std::vector<unsigned char> z;                               // zlib: header, stored deflate blocks, adler32
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t pos = 0; pos < raw.size();) {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back(n & 0xff); z.push_back(n >> 8); z.push_back(~n & 0xff); z.push_back((~n >> 8) & 0xff);
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        for (size_t i = 0; i < n; i++) { a = (a + raw[pos + i]) % 65521u; b = (b + a) % 65521u; }
        pos += n;
    }
    put_be32(z, (b << 16) | a);
    std::vector<unsigned char> png = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<unsigned char> ihdr;
    put_be32(ihdr, (uint32_t)width); put_be32(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8 bit, RGB
    put_chunk(png, "IHDR", ihdr);
    put_chunk(png, "IDAT", z);
    put_chunk(png, "IEND", {});
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(png.data(), 1, png.size(), f) == png.size();
    return (std::fclose(f) == 0) && ok;
}

}  // namespace ptmi
