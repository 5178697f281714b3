// scene.cpp — scene loaders (JSON, PLY), camera set-up, pixel resolve and PPM output.
#include "scene.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <thread>
#include <map>
#include <memory>
#include <sstream>

namespace rth {

// ------------------------------------------------------------------------------------------------
// A small JSON reader: enough for the scene files serde_json writes (main.rs:60-63).
namespace {
struct JVal {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    double num = 0; bool b = false; std::string str;
    std::vector<JVal> arr; std::vector<std::pair<std::string, JVal>> obj;
    const JVal* get(const char* key) const {
        for (const auto& kv : obj) if (kv.first == key) return &kv.second;
        return nullptr;
    }
};
struct JParser {
    const char* p; const char* end; std::string err;
    void ws() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p; }
    bool fail(const char* m) { if (err.empty()) err = m; return false; }
    bool value(JVal& v, int depth = 0) {
        if (depth > 64) return fail("JSON nested too deeply");
        ws();
        if (p >= end) return fail("unexpected end of JSON");
        char c = *p;
        if (c == '{') {
            ++p; v.kind = JVal::Obj; ws();
            if (p < end && *p == '}') { ++p; return true; }
            for (;;) {
                ws(); JVal k;
                if (p >= end || *p != '"' || !string(k.str)) return fail("expected object key");
                ws(); if (p >= end || *p != ':') return fail("expected ':'"); ++p;
                JVal x; if (!value(x, depth + 1)) return false;
                v.obj.emplace_back(std::move(k.str), std::move(x));
                ws(); if (p >= end) return fail("unterminated object");
                if (*p == ',') { ++p; continue; }
                if (*p == '}') { ++p; return true; }
                return fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            ++p; v.kind = JVal::Arr; ws();
            if (p < end && *p == ']') { ++p; return true; }
            for (;;) {
                JVal x; if (!value(x, depth + 1)) return false;
                v.arr.push_back(std::move(x));
                ws(); if (p >= end) return fail("unterminated array");
                if (*p == ',') { ++p; continue; }
                if (*p == ']') { ++p; return true; }
                return fail("expected ',' or ']'");
            }
        }
        if (c == '"') { v.kind = JVal::Str; return string(v.str); }
        if (!strncmp(p, "true", 4 < end - p ? 4 : end - p) && end - p >= 4) { p += 4; v.kind = JVal::Bool; v.b = true; return true; }
        if (end - p >= 5 && !strncmp(p, "false", 5)) { p += 5; v.kind = JVal::Bool; v.b = false; return true; }
        if (end - p >= 4 && !strncmp(p, "null", 4)) { p += 4; v.kind = JVal::Null; return true; }
        // number: strtod is correctly rounded (glibc); serde_json's default parser may differ by 1 ulp on
        // >15-digit mantissas — unverifiable without the Rust binary
        char* e = nullptr;
        std::string tmp(p, (size_t)std::min<ptrdiff_t>(end - p, 64));
        double d = strtod(tmp.c_str(), &e);
        if (e == tmp.c_str()) return fail("bad JSON value");
        p += (e - tmp.c_str()); v.kind = JVal::Num; v.num = d;
        return true;
    }
    bool string(std::string& out) {
        ++p;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) return fail("bad escape");
                switch (*p) {
                case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
                case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                case 'u': if (end - p < 5) return fail("bad \\u escape"); out += '?'; p += 4; break;
                default: out += *p;
                }
                ++p;
            } else out += *p++;
        }
        if (p >= end) return fail("unterminated string");
        ++p; return true;
    }
};
bool vec3_of(const JVal* v, double out[3]) {           // Vec3 { e: [f64; 3] } (vec3.rs:6-9)
    if (!v || v->kind != JVal::Obj) return false;
    const JVal* e = v->get("e");
    if (!e || e->kind != JVal::Arr || e->arr.size() != 3) return false;
    for (int i = 0; i < 3; ++i) { if (e->arr[i].kind != JVal::Num) return false; out[i] = e->arr[i].num; }
    return true;
}
bool num_of(const JVal* v, double& out) { if (!v || v->kind != JVal::Num) return false; out = v->num; return true; }
} // namespace

bool load_json(const std::string& path, Scene& out, std::string& err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = "Could not open scene file: " + path; return false; }
    std::stringstream ss; ss << f.rdbuf();
    std::string text = ss.str();
    JParser jp{text.data(), text.data() + text.size(), {}};
    JVal root;
    if (!jp.value(root) || root.kind != JVal::Obj) { err = "Could not read scene file: " + (jp.err.empty() ? std::string("not an object") : jp.err); return false; }
    const JVal* objs = root.get("objects");
    if (!objs || objs->kind != JVal::Arr) { err = "Could not read scene file: missing \"objects\""; return false; }
    out = Scene(); out.filename = path;
    for (size_t i = 0; i < objs->arr.size(); ++i) {
        const JVal& o = objs->arr[i];
        auto bad = [&](const char* what) { err = "Could not read scene file: object " + std::to_string(i) + ": " + what; return false; };
        if (o.kind != JVal::Obj) return bad("not an object");
        const JVal* type = o.get("type");
        if (!type || type->kind != JVal::Str) return bad("missing type tag");
        // material (material.rs:22-25, 53-57, 81-84); one Rc<dyn Material> per object in the file
        const JVal* m = o.get("mat");
        if (!m || m->kind != JVal::Obj) return bad("missing mat");
        const JVal* mt = m->get("type");
        if (!mt || mt->kind != JVal::Str) return bad("mat has no type tag");
        rt_material mat; memset(&mat, 0, sizeof mat);
        if (mt->str == "Lambertian") { mat.type = RT_LAMBERTIAN; if (!vec3_of(m->get("albedo"), mat.albedo)) return bad("Lambertian.albedo"); }
        else if (mt->str == "Metal") { mat.type = RT_METAL; if (!vec3_of(m->get("albedo"), mat.albedo) || !num_of(m->get("fuzz"), mat.param)) return bad("Metal fields"); }
        else if (mt->str == "Dielectric") { mat.type = RT_DIELECTRIC; if (!num_of(m->get("refraction_index"), mat.param)) return bad("Dielectric.refraction_index"); }
        else return bad("unknown material type");
        rt_prim p; memset(&p, 0, sizeof p);
        p.material = (uint32_t)out.mats.size();
        double c[3];
        if (type->str == "Sphere") {                    // sphere.rs:12-17 (deserialisation bypasses Sphere::new)
            p.type = RT_SPHERE;
            if (!vec3_of(o.get("center"), p.v) || !num_of(o.get("radius"), p.v[3])) return bad("Sphere fields");
            c[0] = p.v[0]; c[1] = p.v[1]; c[2] = p.v[2];                 // sphere.rs:69-71
        } else if (type->str == "Triangle") {           // triangle.rs:11-18: centroid is read verbatim
            p.type = RT_TRIANGLE;
            if (!vec3_of(o.get("v0"), p.v) || !vec3_of(o.get("v1"), p.v + 3) || !vec3_of(o.get("v2"), p.v + 6) || !vec3_of(o.get("centroid"), c))
                return bad("Triangle fields");
        } else return bad("unknown object type");
        out.mats.push_back(mat);
        out.prims.push_back(p);
        out.centroids.insert(out.centroids.end(), c, c + 3);
    }
    return true;
}

// ------------------------------------------------------------------------------------------------
// PLY (ASCII and binary) → triangles, utils/parser.rs:15-81
namespace {
struct PlyProp { std::string name; bool is_list = false; std::string type, count_type; };
struct PlyElem { std::string name; size_t count = 0; std::vector<PlyProp> props; };
size_t ply_size(const std::string& t) {
    if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
    if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
    if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
    if (t == "double" || t == "float64") return 8;
    return 0;
}
bool is_float_type(const std::string& t) { return t == "float" || t == "float32"; }
// Decimal token -> f32, correctly rounded like strtof / Rust's str::parse::<f32>() (ply-rs).  Fast path (Clinger): up to 15
// significant digits and a power of ten up to 10^22 are exact in f64, so one f64 multiply or divide is the correctly
// rounded f64; narrowing it is the correctly rounded f32 unless that f64 sits exactly on an f32 rounding boundary, which
// (like every other shape: hex, inf, nan, long mantissas) goes to strtof.
float parse_f32(const char* b) {
    static const double P10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const char* p = b; bool neg = false;
    if (*p == '-') { neg = true; ++p; } else if (*p == '+') ++p;
    uint64_t mant = 0; int digits = 0, exp10 = 0; bool any = false;
    while (*p >= '0' && *p <= '9') { any = true; if (mant || *p != '0') { mant = mant * 10 + (uint64_t)(*p - '0'); ++digits; } ++p; if (digits > 15) return strtof(b, nullptr); }
    if (*p == '.') { ++p; while (*p >= '0' && *p <= '9') { any = true; if (mant || *p != '0') { mant = mant * 10 + (uint64_t)(*p - '0'); ++digits; } --exp10; ++p; if (digits > 15) return strtof(b, nullptr); } }
    if (!any) return strtof(b, nullptr);
    if (*p == 'e' || *p == 'E') {
        ++p; bool eneg = false; if (*p == '-') { eneg = true; ++p; } else if (*p == '+') ++p;
        if (!(*p >= '0' && *p <= '9')) return strtof(b, nullptr);
        int e = 0; while (*p >= '0' && *p <= '9') { e = e * 10 + (*p - '0'); if (e > 10000) return strtof(b, nullptr); ++p; }
        exp10 += eneg ? -e : e;
    }
    if (!(*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r' || *p == 0)) return strtof(b, nullptr);
    if (exp10 < -22 || exp10 > 22) return strtof(b, nullptr);
    double d = (double)mant; d = exp10 < 0 ? d / P10[-exp10] : d * P10[exp10];
    uint64_t bits; memcpy(&bits, &d, 8);
    if ((bits & 0x1fffffffull) == 0x10000000ull) return strtof(b, nullptr);      // exactly half an f32 ulp in f64: decide on the decimal
    float f = (float)d;
    if (!(f - f == 0.0f) || (f != 0.0f && fabsf(f) < 1.17549435e-38f)) return strtof(b, nullptr);   // overflow / subnormal results
    return neg ? -f : f;
}
inline bool parse_int(const char* b, int64_t& v) {
    const char* p = b; bool neg = false; if (*p == '-') { neg = true; ++p; } else if (*p == '+') ++p;
    if (!(*p >= '0' && *p <= '9')) return false;
    int64_t x = 0; int n = 0; while (*p >= '0' && *p <= '9') { x = x * 10 + (*p - '0'); ++p; if (++n > 17) return false; }
    if (!(*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r' || *p == 0)) return false;
    v = neg ? -x : x; return true;
}
double read_bin(const unsigned char* p, const std::string& t, bool big) {
    unsigned char b[8]; size_t n = ply_size(t);
    for (size_t i = 0; i < n; ++i) b[i] = big ? p[n - 1 - i] : p[i];
    if (t == "char" || t == "int8") { int8_t v; memcpy(&v, b, 1); return v; }
    if (t == "uchar" || t == "uint8") { uint8_t v; memcpy(&v, b, 1); return v; }
    if (t == "short" || t == "int16") { int16_t v; memcpy(&v, b, 2); return v; }
    if (t == "ushort" || t == "uint16") { uint16_t v; memcpy(&v, b, 2); return v; }
    if (t == "int" || t == "int32") { int32_t v; memcpy(&v, b, 4); return v; }
    if (t == "uint" || t == "uint32") { uint32_t v; memcpy(&v, b, 4); return v; }
    if (t == "float" || t == "float32") { float v; memcpy(&v, b, 4); return v; }
    double v; memcpy(&v, b, 8); return v;
}

struct Fact { int coord; bool f32, is_idx; };        // what a PLY property means to the loader: x/y/z, float typed, the index list

size_t ply_parallel_min() {                          // rows from which an ASCII element is parsed by several threads (RT_PLY_PAR_MIN)
    const char* e = getenv("RT_PLY_PAR_MIN");
    return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)65536;
}

// Rows [0, count) of an ASCII element, one row per line, parsed by up to 16 threads.  Vertex rows fill `verts` (appended),
// face rows append triangles to `out`.  Returns false (and appends nothing) unless every row parsed cleanly.
bool parse_rows_parallel(const char* cur, const char* end, size_t count, const Fact* facts, const char* is_list, size_t n_props, bool is_v,
                         std::vector<double>& verts, Scene& out, const char*& after) {
    std::vector<const char*> line(count + 1);
    const char* p = cur;
    for (size_t r = 0; r < count; ++r) {
        line[r] = p;
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        if (!nl) { if (r + 1 != count || p >= end) return false; p = end; } else p = nl + 1;
    }
    line[count] = p;
    unsigned nt = std::thread::hardware_concurrency(); if (nt == 0) nt = 1; if (nt > 16) nt = 16;
    if (const char* e = getenv("RT_PLY_THREADS")) { int v = atoi(e); if (v >= 1 && v <= 64) nt = (unsigned)v; }
    const size_t v0 = verts.size(), n_verts = v0 / 3, p0 = out.prims.size();
    if (is_v) verts.resize(v0 + count * 3); else { out.prims.resize(p0 + count); out.centroids.resize((p0 + count) * 3); }
    std::vector<char> bad(nt, 0);
    auto work = [&](unsigned t) {
        const size_t a = count * t / nt, b = count * (t + 1) / nt;
        for (size_t r = a; r < b; ++r) {
            const char* c = line[r]; const char* e = line[r + 1];
            auto tok = [&](const char*& tb) { while (c < e && (*c == ' ' || *c == '\t' || *c == '\r' || *c == '\n')) ++c; if (c >= e) return false; tb = c; while (c < e && !(*c == ' ' || *c == '\t' || *c == '\r' || *c == '\n')) ++c; return true; };
            double xyz[3] = {0, 0, 0}; bool got[3] = {false, false, false}; int64_t idx[3] = {0, 0, 0}; bool got_idx = false;
            for (size_t pi = 0; pi < n_props; ++pi) {
                size_t n = 1; const char* tb;
                if (is_list[pi]) { int64_t nv; if (!tok(tb) || !parse_int(tb, nv) || nv < 0 || nv > 64) { bad[t] = 1; return; } n = (size_t)nv; }
                if (!is_v && is_list[pi] && facts[pi].is_idx && n != 3) { bad[t] = 1; return; }
                for (size_t k = 0; k < n; ++k) {
                    if (!tok(tb)) { bad[t] = 1; return; }
                    if (is_v && !is_list[pi] && facts[pi].coord >= 0) {
                        if (!facts[pi].f32) { bad[t] = 1; return; }
                        xyz[facts[pi].coord] = (double)parse_f32(tb) * 100.0; got[facts[pi].coord] = true;       // parser.rs:67-73
                    } else if (!is_v && is_list[pi] && facts[pi].is_idx) {
                        int64_t iv; if (facts[pi].f32 || !parse_int(tb, iv)) { bad[t] = 1; return; }
                        idx[k] = iv; got_idx = true;
                    }
                }
            }
            const char* tb; if (tok(tb)) { bad[t] = 1; return; }                    // more tokens on the line than the header declares
            if (is_v) {
                if (!got[0] || !got[1] || !got[2]) { bad[t] = 1; return; }
                for (int k = 0; k < 3; ++k) verts[v0 + 3 * r + k] = xyz[k];
            } else {
                if (!got_idx) { bad[t] = 1; return; }
                rt_prim q; memset(&q, 0, sizeof q); q.type = RT_TRIANGLE; q.material = 0;
                for (int k = 0; k < 3; ++k) {
                    if (idx[k] < 0 || (size_t)idx[k] >= n_verts) { bad[t] = 1; return; }
                    for (int cc = 0; cc < 3; ++cc) q.v[3 * k + cc] = verts[(size_t)idx[k] * 3 + cc];
                }
                out.prims[p0 + r] = q;
                for (int k = 0; k < 3; ++k) out.centroids[3 * (p0 + r) + k] = (q.v[k] + q.v[3 + k] + q.v[6 + k]) / 3.0;   // Triangle::new triangle.rs:21-28
            }
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    bool ok = true; for (char b2 : bad) ok = ok && !b2;
    if (!ok) { if (is_v) verts.resize(v0); else { out.prims.resize(p0); out.centroids.resize(p0 * 3); } return false; }
    after = line[count];
    return true;
}
} // namespace

bool load_ply(const std::string& path, const rt_material& mat, Scene& out, std::string& err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = "Failed to open PLY scene: " + path; return false; }
    std::stringstream ss; ss << f.rdbuf();
    const std::string data = ss.str();
    size_t pos = 0;
    auto next_line = [&](std::string& line) {
        if (pos >= data.size()) return false;
        size_t e = data.find('\n', pos);
        if (e == std::string::npos) e = data.size();
        line = data.substr(pos, e - pos);
        if (!line.empty() && line.back() == '\r') line.pop_back();
        pos = e + 1; return true;
    };
    std::string line;
    if (!next_line(line) || line != "ply") { err = "Parse error: not a PLY file"; return false; }
    std::vector<PlyElem> elems; int fmt = -1;   // 0 ascii, 1 little, 2 big
    for (;;) {
        if (!next_line(line)) { err = "Parse error: PLY header not terminated"; return false; }
        std::istringstream ls(line); std::string tok; ls >> tok;
        if (tok == "end_header") break;
        if (tok == "format") { std::string f2; ls >> f2; fmt = f2 == "ascii" ? 0 : (f2 == "binary_little_endian" ? 1 : (f2 == "binary_big_endian" ? 2 : -1)); }
        else if (tok == "element") { PlyElem e; ls >> e.name >> e.count; elems.push_back(e); }
        else if (tok == "property") {
            if (elems.empty()) { err = "Parse error: property before element"; return false; }
            PlyProp p; std::string t; ls >> t;
            if (t == "list") { p.is_list = true; ls >> p.count_type >> p.type >> p.name; } else { p.type = t; ls >> p.name; }
            if (!ply_size(p.type) || (p.is_list && !ply_size(p.count_type))) { err = "Parse error: unknown PLY type " + p.type; return false; }
            elems.back().props.push_back(p);
        }
    }
    if (fmt < 0) { err = "Parse error: unknown PLY format"; return false; }
    bool have_v = false, have_f = false;
    std::vector<double> verts;               // x,y,z per vertex after f32→f64 ×100 (parser.rs:67-73)
    out = Scene(); out.filename = path;
    out.mats.push_back(mat);
    const char* cur = data.data() + pos; const char* end = data.data() + data.size();
    auto ascii_tok = [&](const char*& b, const char*& e2) {
        while (cur < end && (*cur == ' ' || *cur == '\t' || *cur == '\n' || *cur == '\r')) ++cur;
        if (cur >= end) return false;
        b = cur; while (cur < end && !(*cur == ' ' || *cur == '\t' || *cur == '\n' || *cur == '\r')) ++cur;
        e2 = cur; return true;
    };
    std::vector<int64_t> idx;
    for (const PlyElem& el : elems) {
        const bool is_v = el.name == "vertex", is_f = el.name == "face";
        // per-property facts, looked up once per element instead of once per value
        std::vector<Fact> facts;
        for (const PlyProp& pr : el.props)
            facts.push_back({pr.name == "x" ? 0 : (pr.name == "y" ? 1 : (pr.name == "z" ? 2 : -1)), is_float_type(pr.type), pr.name == "vertex_indices"});
        if (is_v) { have_v = true; verts.reserve(el.count * 3); }
        if (is_f) { have_f = true; if (!have_v) { err = "Parse error No vertices in PLY file"; return false; } out.prims.reserve(el.count); out.centroids.reserve(el.count * 3); }
        size_t r_begin = 0;
        if (fmt == 0 && (is_v || is_f) && el.count >= ply_parallel_min()) {
            // Big ASCII elements (the 871 k-triangle dragon is 1.3 M lines): rows are split over threads at line starts.
            // Anything unusual (a row that is not exactly one line, a bad token, an index out of range) makes this
            // return false with nothing consumed, and the token-by-token loop below parses — and reports — as it always did.
            std::vector<Fact> fv(facts.begin(), facts.end());
            std::vector<char> is_list; for (const PlyProp& pr : el.props) is_list.push_back(pr.is_list ? 1 : 0);
            const char* after = nullptr;
            if (parse_rows_parallel(cur, end, el.count, fv.data(), is_list.data(), fv.size(), is_v, verts, out, after)) { cur = after; r_begin = el.count; }
        }
        for (size_t r = r_begin; r < el.count; ++r) {
            double xyz[3] = {0, 0, 0}; bool got[3] = {false, false, false};
            idx.clear(); bool got_idx = false;
            for (size_t pi = 0; pi < el.props.size(); ++pi) {
                const PlyProp& pr = el.props[pi]; const Fact& fa = facts[pi];
                size_t n = 1;
                if (pr.is_list) {
                    if (fmt == 0) { const char *b, *e2; if (!ascii_tok(b, e2)) { err = "Parse error: truncated PLY"; return false; } n = (size_t)strtoll(b, nullptr, 10); }   /* the buffer is NUL-terminated and tokens end at white space */
                    else { size_t sz = ply_size(pr.count_type); if ((size_t)(end - cur) < sz) { err = "Parse error: truncated PLY"; return false; } n = (size_t)read_bin((const unsigned char*)cur, pr.count_type, fmt == 2); cur += sz; }
                }
                for (size_t k = 0; k < n; ++k) {
                    double val; float fval = 0; const bool is_f32 = fa.f32;
                    if (fmt == 0) {
                        const char *b, *e2; if (!ascii_tok(b, e2)) { err = "Parse error: truncated PLY"; return false; }
                        int64_t iv;
                        if (is_f32) { fval = parse_f32(b); val = fval; } else if (parse_int(b, iv)) val = (double)iv; else val = strtod(b, nullptr);
                    } else {
                        size_t sz = ply_size(pr.type); if ((size_t)(end - cur) < sz) { err = "Parse error: truncated PLY"; return false; }
                        val = read_bin((const unsigned char*)cur, pr.type, fmt == 2); cur += sz;
                    }
                    if (is_v && !pr.is_list) {
                        const int c = fa.coord;
                        if (c >= 0) {
                            if (!is_f32) { err = "Parse error Property is not a double value."; return false; }   // parser.rs:67-73 accepts Property::Float only
                            xyz[c] = (double)(float)val * 100.0; got[c] = true;
                        }
                    }
                    if (is_f && pr.is_list && fa.is_idx) { idx.push_back((int64_t)val); got_idx = true; }
                }
            }
            if (is_v) {
                if (!got[0] || !got[1] || !got[2]) { err = "Parse error: Vertex has no x/y/z value"; return false; }
                verts.insert(verts.end(), xyz, xyz + 3);
            }
            if (is_f) {
                if (!got_idx) { err = "Parse error: Face has no vertex_indices value"; return false; }
                if (idx.size() != 3) { err = "Parse error: Face should have 3 vertices"; return false; }
                rt_prim p; memset(&p, 0, sizeof p); p.type = RT_TRIANGLE; p.material = 0;
                for (int k = 0; k < 3; ++k) {
                    if (idx[k] < 0 || (size_t)idx[k] * 3 + 2 >= verts.size()) { err = "Parse error: Vertex " + std::to_string(idx[k]) + " not found"; return false; }
                    for (int c = 0; c < 3; ++c) p.v[3 * k + c] = verts[(size_t)idx[k] * 3 + c];
                }
                double c[3];                            // Triangle::new triangle.rs:21-28
                for (int k = 0; k < 3; ++k) c[k] = (p.v[k] + p.v[3 + k] + p.v[6 + k]) / 3.0;
                out.prims.push_back(p); out.centroids.insert(out.centroids.end(), c, c + 3);
            }
        }
    }
    if (!have_v) { err = "Parse error No vertices in PLY file"; return false; }
    if (!have_f) { err = "Parse error No faces in PLY file"; return false; }
    out.vertex_lattice = 100.0;
    return true;
}

void subdivide_triangles(Scene& sc, int levels) {
    const double S = sc.vertex_lattice;
    auto mid = [S](double x, double y) {
        double m = 0.5 * (x + y);
        return S > 0.0 ? (double)(float)(m / S) * S : m;
    };
    for (int l = 0; l < levels; ++l) {
        std::vector<rt_prim> np; std::vector<double> nc;
        np.reserve(sc.prims.size() * 4); nc.reserve(sc.centroids.size() * 4);
        for (size_t i = 0; i < sc.prims.size(); ++i) {
            const rt_prim& p = sc.prims[i];
            if (p.type != RT_TRIANGLE) { np.push_back(p); nc.insert(nc.end(), &sc.centroids[3 * i], &sc.centroids[3 * i] + 3); continue; }
            double a[3], b[3], c[3], ab[3], bc[3], ca[3];
            for (int k = 0; k < 3; ++k) { a[k] = p.v[k]; b[k] = p.v[3 + k]; c[k] = p.v[6 + k]; ab[k] = mid(a[k], b[k]); bc[k] = mid(b[k], c[k]); ca[k] = mid(c[k], a[k]); }
            const double* tri[4][3] = {{a, ab, ca}, {ab, b, bc}, {ca, bc, c}, {ab, bc, ca}};   // same winding as the parent
            for (auto& t : tri) {
                rt_prim q = p; double ce[3];
                for (int k = 0; k < 3; ++k) { q.v[k] = t[0][k]; q.v[3 + k] = t[1][k]; q.v[6 + k] = t[2][k]; ce[k] = (q.v[k] + q.v[3 + k] + q.v[6 + k]) / 3.0; }
                np.push_back(q); nc.insert(nc.end(), ce, ce + 3);
            }
        }
        sc.prims.swap(np); sc.centroids.swap(nc);
    }
}

// ------------------------------------------------------------------------------------------------
// camera
namespace {
struct V { double x, y, z; };
V operator+(V a, V b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
V operator-(V a, V b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
V operator-(V a) { return {-a.x, -a.y, -a.z}; }
V operator*(V a, double t) { return {a.x * t, a.y * t, a.z * t}; }
V operator*(double t, V a) { return a * t; }
V operator/(V a, double t) { return (1.0 / t) * a; }                  // vec3.rs:203-209
double len(V a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
V unitv(V a) { return a / len(a); }
V crossv(V a, V b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
double deg2rad(double d) { return d * 3.14159265358979323846264338327950288 / 180.0; }   // rtweekend.rs:162-164
} // namespace

bool camera_preset(const std::string& filename, int camera_index, CameraIn& cam, std::string& err) {
    cam = CameraIn();                                   // Camera::new camera.rs:39-51
    cam.aspect_ratio = 16.0 / 9.0;                      // main.rs:73-78
    cam.image_width = 900; cam.samples_per_pixel = 50; cam.max_depth = 50;
    cam.defocus_angle = 0.1; cam.focus_dist = 1.0;
    auto set = [](double* d, double a, double b, double c) { d[0] = a; d[1] = b; d[2] = c; };
    if (filename.rfind("scenes/weekend", 0) == 0) {     // main.rs:81-84
        cam.vfov = 20.0; set(cam.look_from, 13.0, 2.0, 3.0); set(cam.look_at, 0.0, 0.0, 0.0);
    } else if (filename.rfind("scenes/dragon", 0) == 0) {   // main.rs:85-97
        cam.vfov = 20.0; cam.focus_dist = 50.0; set(cam.look_at, 0.0, 12.0, 0.0);
        switch (camera_index) {
        case -1: case 0: set(cam.look_from, 0.0, 15.0, 50.0); break;
        case 1: set(cam.look_from, -50.0, 15.0, 20.0); break;
        case 2: set(cam.look_from, 80.0, 15.0, 10.0); break;
        case 3: set(cam.look_from, 10.0, 50.0, 25.0); break;
        case 4: set(cam.look_from, -10.0, 30.0, 25.0); break;
        default: err = "Camera position does not exist"; return false;
        }
    } else {                                            // main.rs:98-103
        cam.vfov = 90.0; set(cam.look_from, 0.0, 0.0, 0.0); set(cam.look_at, 0.0, 0.0, -1.0);
    }
    set(cam.v_up, 0.0, 1.0, 0.0);                       // main.rs:104
    return true;
}

void camera_initialise(const CameraIn& in, rt_camera& out) {    // camera.rs:84-128
    double hf = (double)in.image_width / in.aspect_ratio;
    uint32_t image_height = !(hf > 0.0) ? 0u : (hf >= 4294967295.0 ? 0xffffffffu : (uint32_t)hf);
    if (image_height < 1) image_height = 1;
    V look_from = {in.look_from[0], in.look_from[1], in.look_from[2]}, look_at = {in.look_at[0], in.look_at[1], in.look_at[2]};
    V v_up = {in.v_up[0], in.v_up[1], in.v_up[2]};
    V center = look_from;
    double theta = deg2rad(in.vfov);
    double h = std::tan(theta / 2.0);
    double viewport_height = 2.0 * h * in.focus_dist;
    double viewport_width = viewport_height * ((double)in.image_width / (double)image_height);
    V w = unitv(look_from - look_at);
    V u = unitv(crossv(v_up, w));
    V v = crossv(w, u);
    V viewport_u = viewport_width * u;
    V viewport_v = viewport_height * -v;
    V du = viewport_u / (double)in.image_width;
    V dv = viewport_v / (double)image_height;
    V upper_left = center - (in.focus_dist * w) - (viewport_u / 2.0) - (viewport_v / 2.0);
    V pixel00 = upper_left + 0.5 * (du + dv);
    double defocus_radius = in.focus_dist * std::tan(deg2rad(in.defocus_angle / 2.0));
    V ddu = u * defocus_radius, ddv = v * defocus_radius;
    out.image_width = in.image_width; out.image_height = image_height;
    auto st = [](double* d, V a) { d[0] = a.x; d[1] = a.y; d[2] = a.z; };
    st(out.center, center); st(out.pixel00, pixel00); st(out.du, du); st(out.dv, dv); st(out.ddu, ddu); st(out.ddv, ddv);
    out.defocus_angle = in.defocus_angle;
}

void resolve_pixels(const double* sum, uint64_t n_pixels, uint32_t spp_total, uint8_t* rgb) {
    const double scale = 1.0 / (double)spp_total;        // camera.rs:93
    for (uint64_t k = 0; k < n_pixels * 3; ++k) {
        double lin = sum[k] * scale;                     // camera.rs:75
        double g = lin > 0.0 ? std::sqrt(lin) : 0.0;     // color.rs:24-30
        double c = g < 0.0 ? 0.0 : (g > 0.999 ? 0.999 : g);   // color.rs:16-19, interval.rs:36-44
        rgb[k] = (uint8_t)(uint32_t)(256.0 * c);
    }
}

bool write_ppm_p3(const std::string& path, uint32_t w, uint32_t h, const uint8_t* rgb, std::string& err) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { err = "Could not open image file: " + path; return false; }
    // "r g b\n" per pixel (color.rs:21); the 256 possible numbers are formatted once (a 4K frame is 25 M of them)
    char num[256][4]; uint8_t len[256];
    for (int v = 0; v < 256; ++v) len[v] = (uint8_t)snprintf(num[v], 4, "%d", v);
    const std::string head = "P3\n" + std::to_string(w) + " " + std::to_string(h) + "\n255\n";   // camera.rs:61
    const uint64_t n = (uint64_t)w * h;
    std::vector<char> buf(head.size() + n * 12);
    char* p = buf.data();
    memcpy(p, head.data(), head.size()); p += head.size();
    for (uint64_t i = 0; i < n; ++i) {
        for (int c = 0; c < 3; ++c) {
            const uint8_t v = rgb[3 * i + c];
            memcpy(p, num[v], len[v]); p += len[v];
            *p++ = c == 2 ? '\n' : ' ';
        }
    }
    const size_t bytes = (size_t)(p - buf.data());
    bool ok = fwrite(buf.data(), 1, bytes, f) == bytes;
    ok = (fclose(f) == 0) && ok;
    if (!ok) err = "Could not write to image file: " + path;
    return ok;
}

// Binary PPM (P6): the same 8-bit pixels as the contractual P3 text in a twelfth of the bytes
bool write_ppm_p6(const std::string& path, uint32_t w, uint32_t h, const uint8_t* rgb, std::string& err) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { err = "Could not open image file: " + path; return false; }
    const std::string head = "P6\n" + std::to_string(w) + " " + std::to_string(h) + "\n255\n";
    bool ok = fwrite(head.data(), 1, head.size(), f) == head.size();
    const size_t n = (size_t)w * h * 3;
    ok = ok && fwrite(rgb, 1, n, f) == n;
    ok = (fclose(f) == 0) && ok;
    if (!ok) err = "Could not write to image file: " + path;
    return ok;
}

} // namespace rth
