// pbrt_loader.cpp — loadPBRT (utils/pbrt_loader.h:178-422) without the vendored parser library.
//
// The reference hands the file to pbrtParser (ext/pbrtparser, a third-party library vendored in the reference tree):
// pbrt::importPBRT -> Scene::makeSingleLevel -> walk of world->instances / world->shapes.  This file restates the part of
// that library the import path runs through - its lexer and statement grammar (impl/syntactic/Lexer.inl, Parser.inl), the
// scoping of materials / textures / area lights (impl/syntactic/Scene.h: Attributes), the extraction of triangle meshes,
// materials and area lights (impl/semantic/Geometry.cpp, Materials.cpp, Textures.cpp), the single-level flattening
// (impl/semantic/Scene.cpp:372-456) and its float affine algebra (include/pbrtParser/math.h) - and then the loader's own
// conversion to primitives, expression by expression, so that the primitive arrays come out bit-identical to the compiled
// reference (tests/test_pbrt_loader.py against oracle/_ref/libptmi_ref_pbrt.so and the committed goldens).
//
// "plymesh" shapes: the .ply reader the library drives (impl/3rdParty/rply.c through impl/semantic/Geometry.cpp:79-208) is
// restated too - header grammar, ASCII and both binary byte orders, every scalar type with rply's range checks, lists -
// for what the import uses: vertex x/y/z (+ nx/ny/nz), the face list vertex_indices / vertex_index, triangles only.
//
// Not restated (the load fails with a message naming the construct): shape
// types the reference importer crashes on (it dereferences the null shape of every type but trianglemesh / plymesh / curve /
// sphere / disk).  Rotate calls the host libm's sinf / cosf exactly as the library does (the numerics contract's sincos differs
// from glibc's in the last bit on 2.6 % of angles, which would show in every rotated vertex).
#include "pbrt_loader.h"

#include <cctype>
#include <cerrno>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>

namespace ptmi {
namespace {

struct PbrtError : std::runtime_error { using std::runtime_error::runtime_error; };

// ---------------------------------------------------------------------------------------------
// float algebra of include/pbrtParser/math.h (evaluation order as written there; build uses -ffp-contract=off)
// ---------------------------------------------------------------------------------------------
struct V3 { float x = 0, y = 0, z = 0; };
inline V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, float b) { return v3(a.x * b, a.y * b, a.z * b); }
inline float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross3(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline V3 normalize3(V3 a) { return a * (1 / sqrtf(dot3(a, a))); }
struct M3 { V3 vx = v3(1, 0, 0), vy = v3(0, 1, 0), vz = v3(0, 0, 1); };
inline M3 m3(V3 x, V3 y, V3 z) { M3 m; m.vx = x; m.vy = y; m.vz = z; return m; }
inline V3 operator*(const M3& a, V3 b) { return a.vx * b.x + a.vy * b.y + a.vz * b.z; }
inline M3 operator*(const M3& a, const M3& b) { return m3(a * b.vx, a * b.vy, a * b.vz); }
inline M3 operator*(const M3& a, float b) { return m3(a.vx * b, a.vy * b, a.vz * b); }
inline M3 transpose3(const M3& a) { return m3(v3(a.vx.x, a.vy.x, a.vz.x), v3(a.vx.y, a.vy.y, a.vz.y), v3(a.vx.z, a.vy.z, a.vz.z)); }
inline float determinant3(const M3& a) { return dot3(a.vx, cross3(a.vy, a.vz)); }
inline M3 inverse_transpose3(const M3& a) { return m3(cross3(a.vy, a.vz), cross3(a.vz, a.vx), cross3(a.vx, a.vy)) * (1 / determinant3(a)); }
// DiffuseAreaLightBB::LinRGB (impl/semantic/Lights.cpp:233-318): the colour of a blackbody of the given temperature - Planck's
// law at 400, 401, ... 700 nm relative to its peak, through the analytic fits of the CIE 1931 matching functions (Wyman, Sloan,
// Shirley 2013) and the XYZ -> linear sRGB matrix; float arithmetic in the library's order, powf / expf from the host libm as
// there (the compiled reference importer returns the same bits: tests/test_pbrt_loader.py).
inline float cie_fit(float lambda, const float (*lobe)[4], int n) {                 // lobe: {amplitude, centre, slope below, slope above}
    float s = 0.0f;
    for (int k = 0; k < n; k++) {
        const float t = (lambda - lobe[k][1]) * ((lambda < lobe[k][1]) ? lobe[k][2] : lobe[k][3]);
        const float term = lobe[k][0] * expf(-0.5f * t * t);
        s = k == 0 ? term : s + term;
    }
    return s;
}
inline V3 blackbodyLinRGB(float temperature) {
    static const float k = 1.3806488E-23f, h = 6.62606957E-34f, c = 2.99792458E8f;
    static const float X[3][4] = {{0.362f, 442.0f, 0.0624f, 0.0374f}, {1.056f, 599.8f, 0.0264f, 0.0323f}, {-0.065f, 501.1f, 0.0490f, 0.0382f}};
    static const float Y[2][4] = {{0.821f, 568.8f, 0.0213f, 0.0247f}, {0.286f, 530.9f, 0.0613f, 0.0322f}};
    static const float Z[2][4] = {{1.217f, 437.0f, 0.0845f, 0.0278f}, {0.681f, 459.0f, 0.0385f, 0.0725f}};
    auto bb = [temperature](float lambda) {
        lambda *= 1E-3f;                                                           // nm to microns
        return float(((2.0f * 1E24f * h * c * c) / powf(lambda, 5.0f)) * (1.0f / (expf((1E6f * h * c) / (lambda * k * temperature)) - 1.0f)));
    };
    const float lambda_max_radiance = 2.8977721e-3f / temperature * 1e9f;
    const float max_radiance = bb(lambda_max_radiance);
    float x = 0.0f, y = 0.0f, z = 0.0f, n = 0.0f;
    for (float lambda = 400.0f; lambda <= 700.0f; lambda += 1.0f) {
        const float p = bb(lambda) / max_radiance;
        x += p * cie_fit(lambda, X, 3);
        y += p * cie_fit(lambda, Y, 2);
        z += p * cie_fit(lambda, Z, 2);
        n += cie_fit(lambda, Y, 2);
    }
    return m3(v3(3.2404542f, -0.9692660f, 0.0556434f), v3(-1.5371385f, 1.8760108f, -0.2040259f), v3(-0.4985314f, 0.0415560f, 1.0572252f)) *
           v3(x / n, y / n, z / n);
}
struct Affine { M3 l; V3 p; };
inline Affine affine(const M3& l, V3 p) { Affine a; a.l = l; a.p = p; return a; }
inline Affine operator*(const Affine& a, const Affine& b) { return affine(a.l * b.l, a.l * b.p + a.p); }
inline Affine affine_scale(V3 u) { return affine(m3(v3(u.x, 0, 0), v3(0, u.y, 0), v3(0, 0, u.z)), v3(0, 0, 0)); }
inline Affine affine_translate(V3 u) { return affine(M3(), u); }
inline Affine affine_inverse(const Affine& a) { const M3 il = transpose3(inverse_transpose3(a.l)); return affine(il, -(il * a.p)); }
inline Affine affine_rotate(V3 axis, float r) {                                   // math.h: affine3f::rotate
    const V3 u = normalize3(axis);
    const float s = sinf(r), c = cosf(r);      // the host libm, as the library: glibc evaluates both in double and rounds once
    return affine(m3(v3(u.x * u.x + (1 - u.x * u.x) * c, u.x * u.y * (1 - c) + u.z * s, u.x * u.z * (1 - c) - u.y * s),
                     v3(u.x * u.y * (1 - c) - u.z * s, u.y * u.y + (1 - u.y * u.y) * c, u.y * u.z * (1 - c) + u.x * s),
                     v3(u.x * u.z * (1 - c) + u.y * s, u.y * u.z * (1 - c) - u.x * s, u.z * u.z + (1 - u.z * u.z) * c)),
                  v3(0, 0, 0));
}
inline V3 xfmPoint(const Affine& m, V3 p) { return m.l * p + m.p; }
inline V3 xfmNormal(const Affine& m, V3 n) { return inverse_transpose3(m.l) * n; }

// ---------------------------------------------------------------------------------------------
// lexer (impl/syntactic/Lexer.inl) with the Include splice of Parser.inl:peek
// ---------------------------------------------------------------------------------------------
struct Token {
    enum Type { NONE, STRING, SPECIAL, LITERAL } type = NONE;
    std::string text;
    explicit operator bool() const { return type != NONE; }
};
// The token grammar of impl/syntactic/Lexer.inl as a table-driven scanner over the whole file text: every byte belongs to one of
// five classes, and a token is a RUN - blanks and comments are skipped with one search each, a string is the span up to the next
// quote, a literal the maximal run of "word" bytes.  Tokens: "..." (no escapes), one of [ , ] alone, or a literal that ends before
// the next blank, #, quote or bracket.  A comment runs to the end of its line; an unterminated one ends the input.
struct Lexer {
    enum Class : unsigned char { WORD = 0, BLANK, HASH, QUOTE, BRACKET };
    struct Table { unsigned char cls[256]; };
    static const Table& table() {
        static const Table t = [] {
            Table x{};
            for (unsigned char c : {' ', '\n', '\t', '\r'}) x.cls[c] = BLANK;
            x.cls[(unsigned char)'#'] = HASH; x.cls[(unsigned char)'"'] = QUOTE;
            for (unsigned char c : {'[', ',', ']'}) x.cls[c] = BRACKET;
            return x;
        }();
        return t;
    }
    std::string buf; size_t pos = 0;
    explicit Lexer(const std::string& file) {
        std::ifstream in(file, std::ios::binary);
        if (!in) throw PbrtError("could not open '" + file + "'");
        std::ostringstream ss; ss << in.rdbuf(); buf = ss.str();
    }
    Class at(size_t i) const { return (Class)table().cls[(unsigned char)buf[i]]; }
    Token next() {
        const size_t n = buf.size();
        while (pos < n) {                                  // between tokens: blanks and whole comment lines
            if (at(pos) == BLANK) { pos++; continue; }
            if (at(pos) != HASH) break;
            const size_t eol = buf.find('\n', pos);
            pos = eol == std::string::npos ? n : eol + 1;
            if (eol == std::string::npos) return Token();
        }
        if (pos >= n) return Token();
        Token t;
        switch (at(pos)) {
            case QUOTE: {
                const size_t close = buf.find('"', pos + 1);
                if (close == std::string::npos) throw PbrtError("string literal without a closing quote");
                t.type = Token::STRING; t.text.assign(buf, pos + 1, close - pos - 1); pos = close + 1;
                return t;
            }
            case BRACKET:
                t.type = Token::SPECIAL; t.text.assign(1, buf[pos++]);
                return t;
            default: {
                size_t end = pos + 1;
                while (end < n && at(end) == WORD) end++;
                t.type = Token::LITERAL; t.text.assign(buf, pos, end - pos); pos = end;
                return t;
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------
// syntactic scene (impl/syntactic/Scene.h)
// ---------------------------------------------------------------------------------------------
struct TextureDecl;
struct Param {
    enum Kind { FLOAT, INT, BOOL, STRING, TEXTURE } kind = FLOAT;
    std::string type;
    std::vector<float> f; std::vector<int> i; std::vector<bool> b; std::vector<std::string> s;
    std::shared_ptr<TextureDecl> texture;
    size_t size() const { return kind == FLOAT ? f.size() : kind == INT ? i.size() : kind == BOOL ? b.size() : kind == STRING ? s.size() : 1; }
};
struct ParamSet {
    std::map<std::string, Param> param;
    const Param* find(const std::string& n) const { auto it = param.find(n); return it == param.end() ? nullptr : &it->second; }
    const Param* findKind(const std::string& n, Param::Kind k) const { const Param* p = find(n); return p && p->kind == k ? p : nullptr; }
    bool hasTexture(const std::string& n) const { return findKind(n, Param::TEXTURE) != nullptr; }
    bool hasNf(const std::string& n, size_t N) const { const Param* p = findKind(n, Param::FLOAT); return p && p->f.size() == N; }
    // The accessors of ParamSet (impl/syntactic/Scene.cpp:141-330) share one rule: an absent parameter is "not there" (the caller's
    // fallback), a parameter of another kind or of the wrong length fails the load.  `count` 0 = any length.
    const Param* expect(const std::string& n, Param::Kind kind, size_t count) const {
        static const char* const names[] = {"float", "integer", "bool", "string", "texture"};
        const Param* p = find(n);
        if (!p) return nullptr;
        if (p->kind != kind) throw PbrtError("parameter '" + n + "' is a " + names[p->kind] + ", a " + names[kind] + " was asked for");
        if (count && p->size() != count)
            throw PbrtError("parameter '" + n + "' holds " + std::to_string(p->size()) + " values where " + std::to_string(count) + " are required");
        return p;
    }
    bool get3f(float* out, const std::string& n) const {
        const Param* p = expect(n, Param::FLOAT, 3);
        if (p) std::copy(p->f.begin(), p->f.end(), out);
        return p != nullptr;
    }
    bool get2f(float* out, const std::string& n) const {
        const Param* p = expect(n, Param::FLOAT, 2);
        if (p) std::copy(p->f.begin(), p->f.end(), out);
        return p != nullptr;
    }
    float get1f(const std::string& n, float fallback = 0) const { const Param* p = expect(n, Param::FLOAT, 1); return p ? p->f[0] : fallback; }
    int get1i(const std::string& n, int fallback = 0) const { const Param* p = expect(n, Param::INT, 1); return p ? p->i[0] : fallback; }
    bool getBool(const std::string& n, bool fallback = false) const { const Param* p = expect(n, Param::BOOL, 1); return p ? (bool)p->b[0] : fallback; }
    std::string getString(const std::string& n) const { const Param* p = expect(n, Param::STRING, 1); return p ? p->s[0] : std::string(); }
    std::shared_ptr<TextureDecl> getTexture(const std::string& n) const { const Param* p = expect(n, Param::TEXTURE, 0); return p ? p->texture : nullptr; }
    void getPairNf(const std::string& n) const {                      // a list of (x, y) pairs: an odd number of floats fails the load
        const Param* p = expect(n, Param::FLOAT, 0);
        if (p && p->f.size() % 2 != 0) throw PbrtError("parameter '" + n + "' holds an odd number of floats where pairs are required");
    }
};
struct Scope;
struct TextureDecl : ParamSet { std::string name, texelType, mapType; bool checked = false; };
struct MaterialDecl : ParamSet { std::string type, name; std::shared_ptr<Scope> scope; bool checked = false; };
struct AreaLightDecl : ParamSet { std::string type; };
// Attributes (impl/syntactic/Scene.h:102-199): area lights are COPIED into a new scope and appended to, named items are looked
// up through the parent chain
struct Scope {
    std::shared_ptr<Scope> parent;
    std::vector<std::shared_ptr<AreaLightDecl>> areaLights;
    std::map<std::string, std::shared_ptr<MaterialDecl>> namedMaterial;
    std::map<std::string, std::shared_ptr<TextureDecl>> namedTexture;
    std::shared_ptr<MaterialDecl> findMaterial(const std::string& n) const {
        for (const Scope* s = this; s; s = s->parent.get()) { auto it = s->namedMaterial.find(n); if (it != s->namedMaterial.end()) return it->second; }
        return nullptr;
    }
    std::shared_ptr<TextureDecl> findTexture(const std::string& n) const {
        for (const Scope* s = this; s; s = s->parent.get()) { auto it = s->namedTexture.find(n); if (it != s->namedTexture.end()) return it->second; }
        return nullptr;
    }
};
struct ObjectDecl;
struct ShapeDecl : ParamSet {
    std::string type;
    std::shared_ptr<MaterialDecl> material;
    std::vector<std::shared_ptr<AreaLightDecl>> areaLights;      // the attribute clone's list at the Shape statement
    Affine xfm;
};
struct ObjectDecl {
    std::string name;
    std::vector<std::shared_ptr<ShapeDecl>> shapes;
    std::vector<std::pair<std::shared_ptr<ObjectDecl>, Affine>> instances;
};

// ---------------------------------------------------------------------------------------------
// parser (impl/syntactic/Parser.inl)
// ---------------------------------------------------------------------------------------------
struct Parser {
    struct CTM { Affine atStart; bool startActive = true, endActive = true; };
    std::string rootNamePath;
    std::vector<std::shared_ptr<Lexer>> lexerStack;
    std::shared_ptr<Lexer> tokens;
    std::deque<Token> peekQueue;
    CTM ctm;
    std::vector<CTM> transformStack;
    std::vector<std::shared_ptr<MaterialDecl>> materialStack;
    std::vector<std::shared_ptr<ObjectDecl>> objectStack;
    std::map<std::string, std::shared_ptr<ObjectDecl>> namedObjects;
    std::shared_ptr<Scope> scope = std::make_shared<Scope>();
    std::shared_ptr<MaterialDecl> currentMaterial;
    std::shared_ptr<ObjectDecl> world = std::make_shared<ObjectDecl>();

    static std::string pathOf(std::string fn) {                                    // Parser.inl:892-901
        for (char& c : fn) if (c == '\\') c = '/';
        const size_t pos = fn.find_last_of('/');
        return pos == std::string::npos ? std::string() : fn.substr(0, pos + 1);
    }
    explicit Parser(const std::string& fn) {
        rootNamePath = pathOf(fn);
        tokens = std::make_shared<Lexer>(fn);
        objectStack.push_back(world);
    }
    Token peek() {
        while (peekQueue.empty()) {
            Token t = tokens->next();
            if (t && t.text == "Include" ) {                                        // any token type, as the library compares the text only
                Token name = tokens->next();
                std::string file = name.text;
                if (file.empty() || file[0] != '/') file = rootNamePath + "/" + file;
                lexerStack.push_back(tokens);
                tokens = std::make_shared<Lexer>(file);
                continue;
            }
            if (t) { peekQueue.push_back(t); continue; }
            if (lexerStack.empty()) return Token();
            tokens = lexerStack.back(); lexerStack.pop_back();
        }
        return peekQueue.front();
    }
    Token next() {
        Token t = peek();
        if (!t) throw PbrtError("unexpected end of file ...");
        peekQueue.pop_front();
        return t;
    }
    static float stodf(const std::string& s) {
        try { return (float)std::stod(s); } catch (const std::exception&) { throw PbrtError("not a number: '" + s + "'"); }
    }
    float parseFloat() { return stodf(next().text); }
    V3 parseVec3f() { const float x = parseFloat(), y = parseFloat(), z = parseFloat(); return v3(x, y, z); }
    Affine parseMatrix() {                                                          // Parser.inl:101-143 (asserts elided)
        next();
        Affine m; float v[16];
        for (float& e : v) e = stodf(next().text);
        m.l.vx = v3(v[0], v[1], v[2]); m.l.vy = v3(v[4], v[5], v[6]); m.l.vz = v3(v[8], v[9], v[10]); m.p = v3(v[12], v[13], v[14]);
        next();
        return m;
    }
    void addTransform(const Affine& x) { if (ctm.startActive) ctm.atStart = ctm.atStart * x; }
    void setTransform(const Affine& x) { if (ctm.startActive) ctm.atStart = x; }

    static std::vector<std::string> splitWords(const std::string& in) {
        std::vector<std::string> out; size_t pos = 0;
        while (true) {
            const size_t b = in.find_first_not_of(" \n\t", pos);
            if (b == std::string::npos) return out;
            const size_t e = in.find_first_of(" \n\t", b);
            out.push_back(in.substr(b, e == std::string::npos ? std::string::npos : e - b));
            pos = e;
            if (e == std::string::npos) return out;
        }
    }
    bool parseParam(ParamSet& into) {                                               // Parser.inl:146-241
        Token t = peek();
        if (t.type != Token::STRING) return false;
        const std::vector<std::string> comp = splitWords(next().text);
        if (comp.size() != 2) throw PbrtError("parameter declaration '" + t.text + "' is not \"type name\"");
        const std::string& type = comp[0];
        Param p; p.type = type;
        if (type == "float" || type == "color" || type == "blackbody" || type == "rgb" || type == "spectrum" || type == "normal" ||
            type == "point" || type == "point2" || type == "point3" || type == "point4" || type == "vector") p.kind = Param::FLOAT;
        else if (type == "integer") p.kind = Param::INT;
        else if (type == "bool") p.kind = Param::BOOL;
        else if (type == "texture") p.kind = Param::TEXTURE;
        else if (type == "string") p.kind = Param::STRING;
        else throw PbrtError("unknown parameter type '" + type + "'");
        auto add = [&](const std::string& text) {
            try {
                switch (p.kind) {
                    case Param::FLOAT: p.f.push_back(std::stof(text)); break;
                    case Param::INT: p.i.push_back(atoi(text.c_str())); break;
                    case Param::BOOL:
                        if (text == "true") p.b.push_back(true); else if (text == "false") p.b.push_back(false);
                        else throw PbrtError("invalid value '" + text + "' for bool parameter");
                        break;
                    case Param::STRING: p.s.push_back(text); break;
                    case Param::TEXTURE: p.texture = scope->findTexture(text); break;       // unknown names stay null (a warning there)
                }
            } catch (const PbrtError&) { throw; } catch (const std::exception&) { throw PbrtError("not a number: '" + text + "'"); }
        };
        const std::string value = next().text;
        if (value == "[") {
            std::string v = next().text;
            while (v != "]") { add(v); v = next().text; }
        } else {
            if (type == "spectrum") {
                // a measured spectrum: the named file (relative to the scene's directory unless absolute) is tokenised by the same
                // lexer and EVERY token joins the parameter as a float (Parser.inl:208-224) - (wavelength, value) pairs for the
                // material to take with getParamPairNf (Materials.cpp:164-183); a file that is not there, or a token that is not
                // a number, fails the load as it does in the library
                std::string file = value;
                if (file.empty() || file[0] != '/') file = rootNamePath + "/" + file;
                Lexer spd(file);
                for (Token x = spd.next(); x; x = spd.next()) add(x.text);
            } else add(value);
        }
        into.param[comp[1]] = p;
        return true;
    }
    void parseParams(ParamSet& into) { while (parseParam(into)) {} }

    void pushAttributes() {
        auto s = std::make_shared<Scope>(); s->parent = scope; s->areaLights = scope->areaLights; scope = s;
        materialStack.push_back(currentMaterial);
        transformStack.push_back(ctm);
    }
    void popAttributes() {
        if (transformStack.empty() || materialStack.empty() || !scope->parent) throw PbrtError("AttributeEnd without AttributeBegin");
        ctm = transformStack.back(); transformStack.pop_back();
        scope = scope->parent;
        currentMaterial = materialStack.back(); materialStack.pop_back();
    }
    std::shared_ptr<ObjectDecl> findNamedObject(const std::string& name) {
        auto it = namedObjects.find(name);
        if (it != namedObjects.end()) return it->second;
        auto o = std::make_shared<ObjectDecl>(); o->name = name;
        namedObjects[name] = o;
        return o;
    }
    bool parseTransform(const Token& t) {                                           // Parser.inl:311-395
        const std::string& s = t.text;
        if (s == "ActiveTransform") {
            const std::string which = next().text;
            if (which == "All") { ctm.startActive = true; ctm.endActive = true; }
            else if (which == "StartTime") { ctm.startActive = true; ctm.endActive = false; }
            else if (which == "EndTime") { ctm.startActive = false; ctm.endActive = true; }
            else throw PbrtError("unknown argument '" + which + "' to 'ActiveTransform' command");
            return true;
        }
        if (s == "TransformBegin") { transformStack.push_back(ctm); return true; }
        if (s == "TransformEnd") {
            if (transformStack.empty()) throw PbrtError("TransformEnd without TransformBegin");
            ctm = transformStack.back(); transformStack.pop_back(); return true;
        }
        if (s == "Scale") { addTransform(affine_scale(parseVec3f())); return true; }
        if (s == "Translate") { addTransform(affine_translate(parseVec3f())); return true; }
        if (s == "ConcatTransform") { addTransform(parseMatrix()); return true; }
        if (s == "Rotate") {
            const float angle = parseFloat();
            const V3 axis = parseVec3f();
            addTransform(affine_rotate(axis, angle * (float)M_PI / 180.f));
            return true;
        }
        if (s == "Transform") {
            next();
            Affine m;
            m.l.vx = parseVec3f(); next();
            m.l.vy = parseVec3f(); next();
            m.l.vz = parseVec3f(); next();
            m.p = parseVec3f(); next();
            next();
            addTransform(m);                                                        // the library CONCATENATES here too
            return true;
        }
        if (s == "Identity") { setTransform(Affine()); return true; }
        if (s == "ReverseOrientation") return true;                                 // not read by the importer
        if (s == "CoordSysTransform") { next(); return true; }
        return false;
    }
    void makeNamedMedium() {
        next();
        ParamSet p; parseParams(p);
        const Param* type = p.find("type");
        if (!type) throw PbrtError("named medium that does not specify a 'type' parameter!?");
        if (type->kind != Param::STRING) throw PbrtError("named medium has a type, but not a string!?");
    }
    void parseWorld() {                                                             // Parser.inl:398-660
        while (true) {
            const Token t = next();
            const std::string& s = t.text;
            if (s == "WorldEnd") break;
            if (s == "LightSource") { next(); ParamSet p; parseParams(p); continue; }        // not read by the importer
            if (s == "AreaLightSource") {
                auto a = std::make_shared<AreaLightDecl>(); a->type = next().text;
                parseParams(*a);
                scope->areaLights.push_back(a);
                continue;
            }
            if (s == "Material") {
                auto m = std::make_shared<MaterialDecl>(); m->type = next().text;
                parseParams(*m);
                m->scope = scope;
                currentMaterial = m;
                continue;
            }
            if (s == "Texture") {
                auto x = std::make_shared<TextureDecl>();
                x->name = next().text; x->texelType = next().text; x->mapType = next().text;
                scope->namedTexture[x->name] = x;
                parseParams(*x);
                continue;
            }
            if (s == "MakeNamedMaterial") {
                auto m = std::make_shared<MaterialDecl>(); m->type = "<implicit>";
                const std::string name = next().text;
                scope->namedMaterial[name] = m;
                parseParams(*m);
                m->scope = scope;
                const Param* type = m->find("type");
                if (!type) throw PbrtError("named material that does not specify a 'type' parameter!?");
                if (type->kind != Param::STRING) throw PbrtError("named material has a type, but not a string!?");
                if (type->s.empty()) throw PbrtError("named material with an empty 'type' parameter");
                m->type = type->s[0]; m->name = name;
                continue;
            }
            if (s == "MakeNamedMedium") { makeNamedMedium(); continue; }
            if (s == "NamedMaterial") { currentMaterial = scope->findMaterial(next().text); continue; }
            if (s == "MediumInterface") { next(); next(); continue; }
            if (s == "AttributeBegin") { pushAttributes(); continue; }
            if (s == "AttributeEnd") { popAttributes(); continue; }
            if (s == "Shape") {
                auto sh = std::make_shared<ShapeDecl>();
                sh->type = next().text; sh->material = currentMaterial; sh->areaLights = scope->areaLights; sh->xfm = ctm.atStart;
                parseParams(*sh);
                objectStack.back()->shapes.push_back(sh);
                continue;
            }
            if (s == "Volume") { next(); ParamSet p; parseParams(p); continue; }
            if (parseTransform(t)) continue;
            if (s == "ObjectBegin") { objectStack.push_back(findNamedObject(next().text)); continue; }
            if (s == "ObjectEnd") {
                if (objectStack.size() < 2) throw PbrtError("ObjectEnd without ObjectBegin");
                objectStack.pop_back(); continue;
            }
            if (s == "ObjectInstance") {
                objectStack.back()->instances.emplace_back(findNamedObject(next().text), ctm.atStart);
                continue;
            }
            throw PbrtError("unexpected token '" + s + "'");
        }
    }
    void parseScene() {                                                             // Parser.inl:709-884
        while (peek()) {
            const Token t = next();
            const std::string& s = t.text;
            if (parseTransform(t)) continue;
            if (s == "LookAt") {
                const V3 v0 = parseVec3f(), v1 = parseVec3f(), v2 = parseVec3f();
                Affine m;
                m.l.vz = normalize3(v1 - v0);
                m.l.vx = normalize3(cross3(v2, m.l.vz));
                m.l.vy = cross3(m.l.vz, m.l.vx);
                m.p = v0;
                addTransform(affine_inverse(m));
                continue;
            }
            if (s == "Camera" || s == "Sampler" || s == "Integrator" || s == "SurfaceIntegrator" || s == "VolumeIntegrator" ||
                s == "PixelFilter" || s == "Accelerator" || s == "Film" || s == "Renderer") {
                next(); ParamSet p; parseParams(p); continue;
            }
            if (s == "WorldBegin") { ctm = CTM(); parseWorld(); continue; }
            if (s == "MediumInterface") { next(); next(); continue; }
            if (s == "MakeNamedMedium") { makeNamedMedium(); continue; }
            if (s == "Material") throw PbrtError("'Material' field not within a WorldBegin/End context.");
            throw PbrtError("unexpected token '" + s + "'");
        }
    }
};

// ---------------------------------------------------------------------------------------------
// semantic pass (impl/semantic/Geometry.cpp, Materials.cpp, Textures.cpp): what the importer reads, and the errors it raises
// ---------------------------------------------------------------------------------------------
struct Mesh {
    std::vector<V3> vertex, normal;
    std::vector<int> index;                              // 3 per face
    std::shared_ptr<MaterialDecl> material;
    bool hasLight = false; V3 L;
};
struct SemObject {
    std::string name;
    std::vector<std::shared_ptr<Mesh>> shapes;           // nullptr: a shape that is not a triangle mesh (skipped by the importer)
    std::vector<std::pair<std::shared_ptr<SemObject>, Affine>> instances;
};

// ---------------------------------------------------------------------------------------------
// PLY files of "plymesh" shapes: pbrt::ply::parse (impl/semantic/Geometry.cpp:79-208) over rply (impl/3rdParty/rply.c)
// ---------------------------------------------------------------------------------------------
namespace ply {
enum { T_I8, T_U8, T_I16, T_U16, T_I32, T_U32, T_F32, T_F64, T_LIST };
int typeOf(const std::string& w) {                                                  // ply_type_list, rply.c:77-83
    static const char* const names[17] = {"int8", "uint8", "int16", "uint16", "int32", "uint32", "float32", "float64",
                                          "char", "uchar", "short", "ushort", "int", "uint", "float", "double", "list"};
    for (int i = 0; i < 17; i++) if (w == names[i]) return i == 16 ? (int)T_LIST : i % 8;
    return -1;
}
struct Property { std::string name; int type = 0, lenType = 0, valType = 0; };
struct Element { std::string name; long count = 0; std::vector<Property> props; };

struct Reader {
    std::string buf; size_t pos = 0;
    int mode = 2;                                                                   // 0 big endian, 1 little endian, 2 ascii
    bool rn = false;
    std::string word;
    static bool blank(char c) { return c == ' ' || c == '\n' || c == '\r' || c == '\t'; }
    // ply_read_word (rply.c:956-1003): skip blanks, take the word, consume the ONE character behind it
    bool readWord() {
        while (pos < buf.size() && blank(buf[pos])) pos++;
        if (pos >= buf.size()) return false;                                        // "Unexpected end of file"
        const size_t b = pos;
        while (pos < buf.size() && !blank(buf[pos])) pos++;
        word.assign(buf, b, pos - b);
        if (pos < buf.size()) pos++;
        const size_t nul = word.find('\0');                                         // the library compares C strings
        if (nul != std::string::npos) word.resize(nul);
        return !word.empty() && word.size() < 256;                                  // WORDSIZE
    }
    // ply_read_line (rply.c:1013-1046): up to and including the next newline
    bool readLine() {
        const size_t e = buf.find('\n', pos);
        if (e == std::string::npos) return false;
        const size_t len = e - pos;
        pos = e + 1;
        return len < 1024;                                                          // LINESIZE
    }
    bool chunk(void* out, size_t n) {                                               // ply_read_chunk / _reverse
        if (buf.size() - pos < n) return false;
        unsigned char* o = static_cast<unsigned char*>(out);
        if (mode == 1) std::memcpy(o, buf.data() + pos, n);
        else for (size_t i = 0; i < n; i++) o[i] = (unsigned char)buf[pos + n - 1 - i];
        pos += n;
        return true;
    }
    // the idriver handlers (rply.c:1415-1530); the value travels as a double, as in the library
    bool value(int type, double& v) {
        if (mode == 2) {
            if (!readWord()) return false;
            char* end = nullptr;
            if (type == T_F32 || type == T_F64) {
                v = std::strtod(word.c_str(), &end);
                if (*end) return false;
                return type == T_F32 ? !(v < -(double)FLT_MAX || v > (double)FLT_MAX) : !(v < -DBL_MAX || v > DBL_MAX);
            }
            v = (double)std::strtol(word.c_str(), &end, 10);
            if (*end) return false;
            static const double lo[6] = {-128.0, 0.0, -32768.0, 0.0, -2147483648.0, 0.0};
            static const double hi[6] = {127.0, 255.0, 32767.0, 65535.0, 2147483647.0, 4294967295.0};
            return !(v > hi[type] || v < lo[type]);
        }
        switch (type) {
            case T_I8: { int8_t x; if (!chunk(&x, 1)) return false; v = x; return true; }
            case T_U8: { uint8_t x; if (!chunk(&x, 1)) return false; v = x; return true; }
            case T_I16: { int16_t x; if (!chunk(&x, 2)) return false; v = x; return true; }
            case T_U16: { uint16_t x; if (!chunk(&x, 2)) return false; v = x; return true; }
            case T_I32: { int32_t x; if (!chunk(&x, 4)) return false; v = x; return true; }
            case T_U32: { uint32_t x; if (!chunk(&x, 4)) return false; v = x; return true; }
            case T_F32: { float x; if (!chunk(&x, 4)) return false; v = x; return true; }
            default: return chunk(&v, 8);
        }
    }
};

// ply_read_header (rply.c:396-424) with its sub-parsers (:1194-1279)
void readHeader(Reader& r, std::vector<Element>& elements, const std::string& fileName) {
    const PbrtError bad("Unable to read the header of PLY file " + fileName);
    const std::string& b = r.buf;
    if (b.size() < 4 || b[0] != 'p' || b[1] != 'l' || b[2] != 'y' || !std::isspace((unsigned char)b[3])) throw bad;
    r.rn = b[3] == '\r' && b.size() > 4 && b[4] == '\n';
    r.pos = 3;
    if (!r.readWord() || r.word != "format" || !r.readWord()) throw bad;
    if (r.word == "binary_big_endian") r.mode = 0; else if (r.word == "binary_little_endian") r.mode = 1; else if (r.word == "ascii") r.mode = 2; else throw bad;
    if (!r.readWord() || r.word != "1.0" || !r.readWord()) throw bad;
    auto comment = [&]() {                                                          // comment / obj_info: the rest of the line
        if (r.word != "comment" && r.word != "obj_info") return false;
        if (!r.readLine() || !r.readWord()) throw bad;
        return true;
    };
    auto property = [&]() {
        if (r.word != "property") return false;
        if (elements.empty()) throw bad;
        Property p;
        if (!r.readWord()) throw bad;
        p.type = typeOf(r.word);
        if (p.type < 0) throw bad;
        if (p.type == T_LIST) {
            if (!r.readWord()) throw bad;
            p.lenType = typeOf(r.word);
            if (p.lenType < 0 || !r.readWord()) throw bad;
            p.valType = typeOf(r.word);
            if (p.valType < 0) throw bad;
            if (p.lenType == T_LIST || p.valType == T_LIST) throw bad;             // the library would index its handler table with it
        }
        if (!r.readWord()) throw bad;
        p.name = r.word;
        if (!r.readWord()) throw bad;
        elements.back().props.push_back(p);
        return true;
    };
    while (r.word != "end_header") {
        if (comment()) continue;
        if (r.word != "element") throw bad;                                         // "Unexpected token"
        Element e;
        if (!r.readWord()) throw bad;
        e.name = r.word;
        if (!r.readWord()) throw bad;
        if (std::sscanf(r.word.c_str(), "%ld", &e.count) != 1) throw bad;
        if (e.count < 0) throw bad;
        if (!r.readWord()) throw bad;
        elements.push_back(e);
        while (property() || comment()) {}
    }
    if (r.rn) { if (r.pos >= b.size()) throw bad; r.pos++; }
}

// pbrt::ply::parse: positions, optional normals, triangle indices; texture coordinates are read past (the loader has no use for them)
void parse(const std::string& fileName, std::vector<V3>& pos, std::vector<V3>& nor, std::vector<int>& idx) {
    Reader r;
    {
        std::ifstream f(fileName, std::ios::binary);
        if (!f) throw PbrtError("Couldn't open PLY file " + fileName);
        std::ostringstream ss; ss << f.rdbuf(); r.buf = ss.str();
    }
    std::vector<Element> elements;
    readHeader(r, elements, fileName);
    long vertex_count = 0, face_count = 0;
    bool has_normals = false, has_indices = false;
    std::string indices_name;
    int n_vertex_elements = 0, n_face_elements = 0;
    for (const Element& e : elements) {
        auto has = [&](const char* n) { for (const Property& p : e.props) if (p.name == n) return true; return false; };
        if (e.name == "vertex") {
            n_vertex_elements++;
            vertex_count = e.count;
            if (!(has("x") && has("y") && has("z"))) throw PbrtError(fileName + ": Vertex coordinate property not found!");
            has_normals = has("nx") && has("ny") && has("nz");
        } else if (e.name == "face") {
            n_face_elements++;
            face_count = e.count;
            for (const Property& p : e.props) if (p.name == "vertex_index" || p.name == "vertex_indices") { has_indices = true; indices_name = p.name; }
        }
    }
    if (n_vertex_elements > 1 || n_face_elements > 1)                               // the library sizes by the last and fills the first
        throw PbrtError(fileName + ": more than one vertex / face element is not supported");
    if (vertex_count == 0 || face_count == 0) throw PbrtError(fileName + ": PLY file is invalid! No face/vertex elements found!");
    // every instance with a property takes at least one byte of data: a count beyond the file size cannot be read (the
    // library fails at that point of the data too - after allocating for it)
    if ((size_t)vertex_count > r.buf.size() || (size_t)face_count > r.buf.size()) throw PbrtError(fileName + ": unable to read the contents of PLY file");
    pos.assign((size_t)vertex_count, V3());
    if (has_normals) nor.assign((size_t)vertex_count, V3());
    if (has_indices) idx.assign((size_t)face_count * 3, 0);
    const PbrtError unreadable(fileName + ": unable to read the contents of PLY file");
    for (const Element& e : elements) {                                             // ply_read (rply.c:441-454)
        const bool is_vertex = e.name == "vertex", is_face = e.name == "face";
        // a callback is bound to the FIRST property of its name (ply_find_property)
        auto first = [&](const std::string& n) { for (size_t k = 0; k < e.props.size(); k++) if (e.props[k].name == n) return (long)k; return -1L; };
        const long kx = is_vertex ? first("x") : -1, ky = is_vertex ? first("y") : -1, kz = is_vertex ? first("z") : -1;
        const long knx = is_vertex && has_normals ? first("nx") : -1, kny = is_vertex && has_normals ? first("ny") : -1, knz = is_vertex && has_normals ? first("nz") : -1;
        const long kidx = is_face && has_indices ? first(indices_name) : -1;
        for (long j = 0; j < e.count; j++) {
            for (long k = 0; k < (long)e.props.size(); k++) {
                const Property& p = e.props[k];
                double v = 0.0;
                if (p.type != T_LIST) {
                    if (!r.value(p.type, v)) throw unreadable;
                    const float fv = (float)v;
                    if (k == kx) pos[j].x = fv; else if (k == ky) pos[j].y = fv; else if (k == kz) pos[j].z = fv;
                    else if (k == knx) nor[j].x = fv; else if (k == kny) nor[j].y = fv; else if (k == knz) nor[j].z = fv;
                    continue;
                }
                double length = 0.0;
                if (!r.value(p.lenType, length)) throw unreadable;
                const long n = (long)length;
                if (k == kidx && n != 3) throw PbrtError("Found face with vertex count different from 3, only triangles are supported");
                for (long l = 0; l < n; l++) {
                    if (!r.value(p.valType, v)) throw unreadable;
                    if (k == kidx) idx[(size_t)j * 3 + l] = (int)v;
                }
            }
        }
    }
}
}  // namespace ply

void checkTexture(const std::shared_ptr<TextureDecl>& t);
void useTexture(const ParamSet& ps, const std::string& name) { checkTexture(ps.getTexture(name)); }
void checkTexture(const std::shared_ptr<TextureDecl>& t) {                          // Textures.cpp:151-197
    if (!t || t->checked) return;
    t->checked = true;
    const std::string& m = t->mapType;
    auto scaleOrTex = [&](const std::string& n) {
        if (t->hasTexture(n)) useTexture(*t, n);
        else if (t->hasNf(n, 3)) { float v[3]; t->get3f(v, n); }
        else t->get1f(n);
    };
    if (m == "imagemap" || m == "ptex") { t->getString("filename"); if (m == "imagemap") { if (t->hasNf("uscale", 1)) t->get1f("uscale"); if (t->hasNf("vscale", 1)) t->get1f("vscale"); } }
    else if (m == "scale") { scaleOrTex("tex1"); scaleOrTex("tex2"); }
    else if (m == "mix") {
        if (t->hasNf("amount", 3)) { float v[3]; t->get3f(v, "amount"); } else if (t->hasNf("amount", 1)) t->get1f("amount"); else useTexture(*t, "amount");
        scaleOrTex("tex1"); scaleOrTex("tex2");
    }
    else if (m == "constant") { if (t->hasNf("value", 1)) t->get1f("value"); else { float v[3]; t->get3f(v, "value"); } }
    else if (m == "checkerboard") {
        for (const auto& it : t->param) {
            const std::string& n = it.first; float v[3];
            if (n == "uscale" || n == "vscale") t->get1f(n);
            else if (n == "tex1" || n == "tex2") t->get3f(v, n);
            else throw PbrtError("unknown checker texture param '" + n + "'");
        }
    }
    else if (m == "marble") { if (t->hasNf("scale", 1)) t->get1f("scale"); }
    else if (m == "fbm" || m == "windy" || m == "wrinkled") {}
    else throw PbrtError("un-handled pbrt texture type '" + m + "'");
}

void checkMaterial(const std::shared_ptr<MaterialDecl>& m);
std::string materialType(const MaterialDecl& m) { return m.type == "" ? m.getString("type") : m.type; }
void checkMaterial(const std::shared_ptr<MaterialDecl>& m) {                        // Materials.cpp:424-510 + the per-type extractors
    if (!m || m->checked) return;
    m->checked = true;
    const std::string type = materialType(*m);
    float v[3];
    auto rgbOrTex = [&](const std::string& n) { if (m->hasTexture(n)) useTexture(*m, n); else m->get3f(v, n); };
    auto f1OrTex = [&](const std::string& n) { if (m->hasTexture(n)) useTexture(*m, n); else m->get1f(n); };
    auto each = [&](const char* what, const std::function<bool(const std::string&)>& known) {
        for (const auto& it : m->param) if (it.first != "type" && !known(it.first)) throw PbrtError(std::string("un-handled ") + what + "-material parameter '" + it.first + "'");
    };
    if (type == "matte") each("matte", [&](const std::string& n) {
        if (n == "Kd") rgbOrTex(n);
        else if (n == "sigma") { if (!m->hasNf(n, 1)) useTexture(*m, n); }
        else if (n == "bumpmap") useTexture(*m, n);
        else return false;
        return true; });
    else if (type == "plastic") each("plastic", [&](const std::string& n) {
        if (n == "Kd" || n == "Ks") rgbOrTex(n);
        else if (n == "roughness") f1OrTex(n);
        else if (n == "remaproughness") m->getBool(n);
        else if (n == "bumpmap") useTexture(*m, n);
        else return false;
        return true; });
    else if (type == "metal") each("metal", [&](const std::string& n) {
        if (n == "roughness" || n == "uroughness" || n == "vroughness") f1OrTex(n);
        else if (n == "remaproughness") m->getBool(n);
        else if (n == "eta" || n == "k") { if (m->hasNf(n, 3)) m->get3f(v, n); else m->getPairNf(n); }
        else if (n == "bumpmap") useTexture(*m, n);
        else return false;
        return true; });
    else if (type == "mirror") each("mirror", [&](const std::string& n) {
        if (n == "Kr") { if (m->hasTexture(n)) throw PbrtError("mapping Kr for mirror materials not implemented"); m->get3f(v, n); }
        else if (n == "bumpmap") useTexture(*m, n);
        else return false;
        return true; });
    else if (type == "substrate") each("substrate", [&](const std::string& n) {
        if (n == "Kd" || n == "Ks") rgbOrTex(n);
        else if (n == "uroughness" || n == "vroughness") f1OrTex(n);
        else if (n == "remaproughness") m->getBool(n);
        else if (n == "bumpmap") useTexture(*m, n);
        else return false;
        return true; });
    else if (type == "uber") each("uber", [&](const std::string& n) {
        if (n == "Kd" || n == "Kr" || n == "Kt" || n == "Ks" || n == "opacity") rgbOrTex(n);
        else if (n == "alpha" || n == "shadowalpha") f1OrTex(n);
        else if (n == "index" || n == "uroughness" || n == "vroughness") m->get1f(n);
        else if (n == "roughness") { if (m->hasTexture(n)) useTexture(*m, n); else if (m->hasNf(n, 1)) m->get1f(n); else throw PbrtError("uber::roughness in un-recognized format..."); }
        else if (n == "bumpmap") useTexture(*m, n);
        else return false;
        return true; });
    else if (type == "disney") {
        m->get3f(v, "color");
        for (const char* n : {"anisotropic", "clearcoat", "clearcoatgloss", "difftrans", "eta", "flatness", "metallic", "roughness", "sheen",
                              "sheentint", "spectrans", "speculartint"}) m->get1f(n);
        m->getBool("thin");
    }
    else if (type == "mix") {
        if (m->hasTexture("amount")) useTexture(*m, "amount"); else m->get3f(v, "amount");
        const std::string n0 = m->getString("namedmaterial1"), n1 = m->getString("namedmaterial2");
        if (n0 == "") throw PbrtError("mix material w/o 'namedmaterial1' parameter");
        if (n1 == "") throw PbrtError("mix material w/o 'namedmaterial2' parameter");
        const auto m0 = m->scope ? m->scope->findMaterial(n0) : nullptr, m1 = m->scope ? m->scope->findMaterial(n1) : nullptr;
        if (!m0 || !m1) throw PbrtError("mix material naming a material that is not in scope (the reference asserts)");
        checkMaterial(m0); checkMaterial(m1);
    }
    else if (type == "translucent") { m->get3f(v, "transmit"); m->get3f(v, "reflect"); if (m->hasTexture("Kd")) useTexture(*m, "Kd"); else m->get3f(v, "Kd"); }
    else if (type == "glass") { m->get3f(v, "Kr"); m->get3f(v, "Kt"); m->get1f("index"); }
    else if (type == "hair") {
        for (const auto& it : m->param) {
            if (it.first == "eumelanin" || it.first == "alpha" || it.first == "beta_m") m->get1f(it.first);
            else if (it.first != "type") throw PbrtError("as-yet-unhandled hair-material parameter '" + it.first + "'");
        }
    }
    else if (type == "fourier") {
        for (const auto& it : m->param) {
            if (it.first == "bsdffile") m->getString(it.first);
            else if (it.first != "type") throw PbrtError("un-handled fourier-material parameter '" + it.first + "'");
        }
    }
    // "", and every other type: a plain Material (the importer's "Unknown material type")
}

struct Semantic {
    std::string basePath;                                                           // Scene::makeGlobalFileName: basePath + relative name
    std::map<ObjectDecl*, std::shared_ptr<SemObject>> emitted;
    std::shared_ptr<SemObject> emitObject(const std::shared_ptr<ObjectDecl>& o) {   // Geometry.cpp:446-483
        auto it = emitted.find(o.get());
        if (it != emitted.end()) return it->second;
        auto ours = std::make_shared<SemObject>();
        emitted[o.get()] = ours;
        ours->name = o->name;
        for (const auto& sh : o->shapes) ours->shapes.push_back(emitShape(*sh));
        for (const auto& in : o->instances) ours->instances.emplace_back(emitObject(in.first), in.second);
        return ours;
    }
    std::shared_ptr<Mesh> emitShape(const ShapeDecl& sh) {                          // Geometry.cpp:343-444
        const std::string& t = sh.type;
        if (t != "trianglemesh" && t != "plymesh" && t != "curve" && t != "sphere" && t != "disk")
            throw PbrtError("shape type '" + t + "' is not handled by the reference's parser (the reference importer crashes on it)");
        checkMaterial(sh.material);
        for (const auto& it : sh.param) if (it.second.kind == Param::TEXTURE) checkTexture(it.second.texture);
        std::shared_ptr<Mesh> mesh;
        if (t == "trianglemesh") {
            mesh = std::make_shared<Mesh>();
            mesh->material = sh.material;
            auto vecs = [&](const std::string& n) {
                std::vector<V3> out;
                if (const Param* p = sh.findKind(n, Param::FLOAT)) for (size_t k = 0; k + 3 <= p->f.size(); k += 3) out.push_back(v3(p->f[k], p->f[k + 1], p->f[k + 2]));
                return out;
            };
            mesh->vertex = vecs("P");
            mesh->normal = vecs("N");
            if (const Param* p = sh.findKind("indices", Param::INT)) mesh->index.assign(p->i.begin(), p->i.begin() + (p->i.size() / 3) * 3);
            for (V3& v : mesh->vertex) v = xfmPoint(sh.xfm, v);
            for (V3& n : mesh->normal) n = xfmNormal(sh.xfm, n);
        } else if (t == "plymesh") {                                                // emitPlyMesh, Geometry.cpp:222-236
            mesh = std::make_shared<Mesh>();
            mesh->material = sh.material;
            ply::parse(basePath + sh.getString("filename"), mesh->vertex, mesh->normal, mesh->index);
            for (V3& v : mesh->vertex) v = xfmPoint(sh.xfm, v);
            for (V3& n : mesh->normal) n = xfmNormal(sh.xfm, n);
        } else if (t == "sphere" || t == "disk") {
            sh.get1f("radius");
            if (t == "disk" && sh.hasNf("height", 1)) sh.get1f("height");
        } else {                                                                    // curve
            if (sh.findKind("type", Param::STRING)) sh.getString("type");
            if (sh.findKind("basis", Param::STRING)) sh.getString("basis");
            for (const char* n : {"width", "width0", "width1"}) if (sh.hasNf(n, 1)) sh.get1f(n);
            if (const Param* d = sh.findKind("degree", Param::INT)) if (d->i.size() == 1) sh.get1i("degree");
        }
        // area light: the FIRST one of the scope (Geometry.cpp:426-438, 378-403)
        if (!sh.areaLights.empty()) {
            const AreaLightDecl& a = *sh.areaLights[0];
            if (a.type == "diffuse") {
                if (a.hasNf("L", 2)) {
                    a.get1i("nsamples", 1);
                    float TS[2]; a.get2f(TS, "L");
                    // DiffuseAreaLightBB: loadPBRT takes light->LinRGB() (utils/pbrt_loader.h:311-312) - the temperature's colour,
                    // normalised; the second float ("scale") is parsed and never used
                    if (mesh) { mesh->hasLight = true; mesh->L = blackbodyLinRGB(TS[0]); }
                } else if (a.hasNf("L", 3)) {
                    float L[3]; a.get3f(L, "L"); a.get1i("nsamples", 1);
                    if (mesh) { mesh->hasLight = true; mesh->L = v3(L[0], L[1], L[2]); }
                }
            }
        }
        return mesh;
    }
};

// Scene::makeSingleLevel (impl/semantic/Scene.cpp:372-456)
struct Flat { std::shared_ptr<SemObject> object; Affine xfm; };
bool isSingleLevel(const SemObject& w) {
    if (!w.shapes.empty()) return false;
    for (const auto& in : w.instances) if (in.first && !in.first->instances.empty()) return false;
    return true;
}
void flatten(const std::shared_ptr<SemObject>& o, const Affine& xfm, std::vector<Flat>& out, int depth = 0) {
    if (!o) return;
    if (depth > 256) throw PbrtError("object instancing recurses (the reference overflows its stack on such a scene)");
    if (!o->shapes.empty()) out.push_back(Flat{o, xfm});
    for (const auto& in : o->instances) flatten(in.first, xfm * in.second, out, depth + 1);
}

// ---------------------------------------------------------------------------------------------
// the loader proper (utils/pbrt_loader.h)
// ---------------------------------------------------------------------------------------------
struct PBRTMaterial {                                                               // pbrt_loader.h:34-49
    f3 diffuse = mk3(0.8f, 0.8f, 0.8f), specular = mk3(0, 0, 0);
    float metallic = 0.0f;
    f3 getBSDF() const {
        const float k = 1.0f - metallic;
        const f3 a = mk3(diffuse.x * k, diffuse.y * k, diffuse.z * k), b = mk3(specular.x * metallic, specular.y * metallic, specular.z * metallic);
        return a + b;
    }
};
f3 toF3(const float* v) { return mk3(v[0], v[1], v[2]); }
PBRTMaterial convertMaterial(const MaterialDecl& m) {                               // pbrt_loader.h:85-164 on the library's defaults
    PBRTMaterial r;
    const std::string type = materialType(m);
    auto rgb = [&](const std::string& n, float d0, float d1, float d2) {          // field default, 1 1 1 under a texture, else the parameter
        float v[3] = {d0, d1, d2};
        if (m.hasTexture(n)) { v[0] = v[1] = v[2] = 1.0f; } else m.get3f(v, n);
        return toF3(v);
    };
    auto plain = [&](const std::string& n, float d0, float d1, float d2) { float v[3] = {d0, d1, d2}; m.get3f(v, n); return toF3(v); };
    if (type == "disney") {
        r.diffuse = plain("color", 0.5f, 0.5f, 0.5f);
        r.metallic = m.get1f("metallic", 0.f);
        r.specular = mk3(r.diffuse.x * r.metallic, r.diffuse.y * r.metallic, r.diffuse.z * r.metallic);
    } else if (type == "matte") { r.diffuse = rgb("Kd", .5f, .5f, .5f); r.metallic = 0.0f; }
    else if (type == "plastic") { r.diffuse = rgb("Kd", .25f, .25f, .25f); r.specular = rgb("Ks", .25f, .25f, .25f); }
    else if (type == "metal") {
        float eta[3] = {0.21221054f, 0.91804785f, 1.1000715f}, k[3] = {3.9132357f, 2.4519274f, 2.1321275f};
        if (m.hasNf("eta", 3)) m.get3f(eta, "eta");
        if (m.hasNf("k", 3)) m.get3f(k, "k");
        float rr[3];
        for (int i = 0; i < 3; i++) { const float n = eta[i], kv = k[i]; rr[i] = ((n - 1) * (n - 1) + kv * kv) / ((n + 1) * (n + 1) + kv * kv); }
        r.diffuse = toF3(rr); r.metallic = 1.0f;
    }
    else if (type == "mirror") { r.diffuse = mk3(0, 0, 0); r.specular = plain("Kr", .9f, .9f, .9f); r.metallic = 1.0f; }
    else if (type == "glass") { r.diffuse = plain("Kt", 1.f, 1.f, 1.f); }
    else if (type == "substrate") { r.diffuse = rgb("Kd", .5f, .5f, .5f); r.specular = rgb("Ks", .5f, .5f, .5f); }
    else if (type == "uber") { r.diffuse = rgb("Kd", .25f, .25f, .25f); r.specular = rgb("Ks", .25f, .25f, .25f); }
    else if (type == "translucent") { float v[3] = {0.25f, 0.25f, 0.25f}; if (!m.hasTexture("Kd")) m.get3f(v, "Kd"); r.diffuse = toF3(v); }
    return r;                                                                       // every other type: the defaults ("Unknown material type")
}
f3 transformPoint(const Affine& x, V3 p) {                                          // pbrt_loader.h:64-70
    return mk3(x.l.vx.x * p.x + x.l.vy.x * p.y + x.l.vz.x * p.z + x.p.x,
               x.l.vx.y * p.x + x.l.vy.y * p.y + x.l.vz.y * p.z + x.p.y,
               x.l.vx.z * p.x + x.l.vy.z * p.y + x.l.vz.z * p.z + x.p.z);
}
f3 transformNormal(const Affine& x, V3 n) {                                         // pbrt_loader.h:73-81
    return unit_vector(mk3(x.l.vx.x * n.x + x.l.vy.x * n.y + x.l.vz.x * n.z,
                           x.l.vx.y * n.x + x.l.vy.y * n.y + x.l.vz.y * n.z,
                           x.l.vx.z * n.x + x.l.vy.z * n.y + x.l.vz.z * n.z));
}
Primitive makeTriangle(f3 a, f3 b, f3 c, f3 bsdf, f3 normal, f3 Le) {
    Primitive p; p.type = PRIM_TRIANGLE;
    p.v[0] = a; p.v[1] = b; p.v[2] = c; p.v[3] = mk3(0, 0, 0);
    p.bsdf = bsdf; p.normal = normal; p.Le = Le;
    return p;
}

// box3f / getBounds of the flattened scene, for the proxy of oversized scenes (Scene.cpp:34-279, pbrt_loader.h:225-270)
struct Box { V3 lo = v3(FLT_MAX, FLT_MAX, FLT_MAX), hi = v3(-FLT_MAX, -FLT_MAX, -FLT_MAX);
             bool empty() const { return hi.x < lo.x || hi.y < lo.y || hi.z < lo.z; }
             void extend(V3 p) { lo = v3(lo.x > p.x ? p.x : lo.x, lo.y > p.y ? p.y : lo.y, lo.z > p.z ? p.z : lo.z);
                                 hi = v3(hi.x < p.x ? p.x : hi.x, hi.y < p.y ? p.y : hi.y, hi.z < p.z ? p.z : hi.z); } };

}  // namespace

bool loadPBRT(const std::string& filename, std::vector<Primitive>& out, std::string* error) {
    out.clear();
    try {
        Parser parser(filename);
        parser.parseScene();
        Semantic sem;
        sem.basePath = parser.rootNamePath;                                         // Parser.inl:906-913
        const std::shared_ptr<SemObject> world = sem.emitObject(parser.world);
        std::vector<Flat> instances;                                                // world->instances after makeSingleLevel()
        if (isSingleLevel(*world)) for (const auto& in : world->instances) instances.push_back(Flat{in.first, in.second});
        else flatten(world, Affine(), instances);

        const size_t kMaxTriangles = 2000000;                                       // pbrt_loader.h:205-226
        size_t estimated = 0;
        for (const Flat& in : instances) {
            for (const auto& mesh : in.object->shapes) if (mesh) { estimated += mesh->index.size() / 3; if (estimated > kMaxTriangles) break; }
            if (estimated > kMaxTriangles) break;
        }
        if (estimated > kMaxTriangles) {                                            // bounding-box proxy, pbrt_loader.h:228-270
            Box b;
            for (const Flat& in : instances) {                                      // Object::getBounds over Instance::getBounds
                Box ob;
                for (const auto& mesh : in.object->shapes) if (mesh) { Box g; for (const V3& v : mesh->vertex) g.extend(v); if (!g.empty()) { ob.extend(g.lo); ob.extend(g.hi); } }
                if (ob.empty()) continue;
                Box ib;
                for (int k = 0; k < 8; k++) ib.extend(xfmPoint(in.xfm, v3(k & 4 ? ob.hi.x : ob.lo.x, k & 2 ? ob.hi.y : ob.lo.y, k & 1 ? ob.hi.z : ob.lo.z)));
                if (!ib.empty()) { b.extend(ib.lo); b.extend(ib.hi); }
            }
            const f3 lo = mk3(b.lo.x, b.lo.y, b.lo.z), hi = mk3(b.hi.x, b.hi.y, b.hi.z);
            const f3 v000 = mk3(lo.x, lo.y, lo.z), v001 = mk3(lo.x, lo.y, hi.z), v010 = mk3(lo.x, hi.y, lo.z), v011 = mk3(lo.x, hi.y, hi.z),
                     v100 = mk3(hi.x, lo.y, lo.z), v101 = mk3(hi.x, lo.y, hi.z), v110 = mk3(hi.x, hi.y, lo.z), v111 = mk3(hi.x, hi.y, hi.z);
            auto addQuad = [&](f3 a, f3 bb, f3 c, f3 d) {
                const f3 n = unit_vector(cross(bb - a, c - a));
                out.push_back(makeTriangle(a, bb, c, mk3(0.8f, 0.2f, 0.2f), n, mk3(0, 0, 0)));
                out.push_back(makeTriangle(a, c, d, mk3(0.8f, 0.2f, 0.2f), n, mk3(0, 0, 0)));
            };
            addQuad(v000, v001, v011, v010); addQuad(v100, v110, v111, v101); addQuad(v000, v100, v101, v001);
            addQuad(v010, v011, v111, v110); addQuad(v000, v010, v110, v100); addQuad(v001, v101, v111, v011);
            return true;
        }

        for (const Flat& in : instances) {                                          // pbrt_loader.h:279-336
            for (const auto& mesh : in.object->shapes) {
                if (!mesh) continue;                                                // "Skipping non-triangle shape"
                const PBRTMaterial mat = mesh->material ? convertMaterial(*mesh->material) : PBRTMaterial();
                const f3 emission = mesh->hasLight ? mk3(mesh->L.x, mesh->L.y, mesh->L.z) : mk3(0, 0, 0);
                const bool has_normals = mesh->normal.size() >= mesh->vertex.size();
                const f3 bsdf = mat.getBSDF();
                for (size_t f = 0; f + 2 < mesh->index.size(); f += 3) {
                    const int i0 = mesh->index[f], i1 = mesh->index[f + 1], i2 = mesh->index[f + 2];
                    const int nv = (int)mesh->vertex.size();
                    if (i0 < 0 || i1 < 0 || i2 < 0 || i0 >= nv || i1 >= nv || i2 >= nv) throw PbrtError("triangle index outside the vertex array");
                    const f3 a = transformPoint(in.xfm, mesh->vertex[i0]), b = transformPoint(in.xfm, mesh->vertex[i1]), c = transformPoint(in.xfm, mesh->vertex[i2]);
                    const f3 normal = has_normals ? transformNormal(in.xfm, mesh->normal[i0]) : unit_vector(cross(b - a, c - a));
                    out.push_back(makeTriangle(a, b, c, bsdf, normal, emission));
                }
            }
        }
        if (out.empty()) throw PbrtError("No triangles found in PBRT scene");
        return true;
    } catch (const std::exception& e) {
        out.clear();
        if (error) *error = e.what();
        return false;
    }
}

}  // namespace ptmi
