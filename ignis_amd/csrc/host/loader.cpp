// Scene loader: Ignis JSON scene -> SceneDatabase-compatible tables + POD
// lowering of materials / lights / camera / technique (include/ig_tables.h).
//
// Follows (restated, not shared) the reference's loader path:
//   Parser (properties, transforms)      src/runtime/loader/Parser.cpp:110-300
//   shapes -> "shapes" / "trimesh_primbvh" src/runtime/shape/TriMeshProvider.cpp:478-615
//   entities -> "entities" + scene BVH    src/runtime/loader/LoaderEntity.cpp:32-205
//   lights                                src/runtime/light/{AreaLight,PointLight}.cpp, loader/LoaderLight.cpp
//   camera                                src/runtime/camera/PerspectiveCamera.cpp:7-76
//   technique                             src/runtime/technique/PathTechnique.cpp:8-33
//   film                                  src/runtime/Runtime.cpp:38-54
//
// Documented deviation: the reference iterates std::unordered_map's and loads
// shapes in parallel, so its entity / shape / material ids are not a function
// of the scene file. Here ids follow declaration
// order, entities grouped by first-seen material as the reference groups them.
#include "bvh.h"
#include "igh_host.h"
#include "json.h"
#include "mesh.h"
#include "png.h"
#include "exr.h"
#include "floatimage.h"
#include "jpeg.h"
#include "pexpr.h"
#include "hosek.h"

#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <set>
#include <sstream>
#include <string>

namespace igh {

static constexpr float Pi      = 3.14159265358979323846f;
static constexpr float Deg2Rad = Pi / 180.0f;

[[noreturn]] static void fail(const std::string& msg) { throw std::runtime_error(msg); }

// ---------------------------------------------------------------- property helpers

static V3 getVector3(const JsonValue& v, const char* what)
{
    if (!v.isArray() || v.arr.size() != 3)
        fail(std::string("Expected vector of length 3 for ") + what);
    for (const auto& e : v.arr)
        if (!e.isNumber())
            fail(std::string("Given vector is not only numbers: ") + what);
    return V3((float)v.arr[0].num, (float)v.arr[1].num, (float)v.arr[2].num);
}

static Affine fromMatrixArray(const JsonValue& v)
{
    const size_t len = v.arr.size();
    for (const auto& e : v.arr)
        if (!e.isNumber())
            fail("Given matrix is not only numbers");
    Affine t;
    if (len == 9) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                t.L.m[i][j] = (float)v.arr[(size_t)(i * 3 + j)].num;
    } else if (len == 12 || len == 16) {
        // Row major input (Parser.cpp:127-140)
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j)
                t.L.m[i][j] = (float)v.arr[(size_t)(i * 4 + j)].num;
            t.t[i] = (float)v.arr[(size_t)(i * 4 + 3)].num;
        }
    } else {
        fail("Expected transform property to be an array of size 9, 12 or 16");
    }
    return t;
}

static M3 angleAxis(float angle, V3 axis)
{
    // Eigen::AngleAxis::toRotationMatrix
    const float s = std::sin(angle), c = std::cos(angle);
    const V3 sin_axis = axis * s;
    const V3 cos1_axis = axis * (1 - c);
    M3 r;
    float tmp;
    tmp       = cos1_axis.x * axis.y;
    r.m[0][1] = tmp - sin_axis.z;
    r.m[1][0] = tmp + sin_axis.z;
    tmp       = cos1_axis.x * axis.z;
    r.m[0][2] = tmp + sin_axis.y;
    r.m[2][0] = tmp - sin_axis.y;
    tmp       = cos1_axis.y * axis.z;
    r.m[1][2] = tmp - sin_axis.x;
    r.m[2][1] = tmp + sin_axis.x;
    r.m[0][0] = cos1_axis.x * axis.x + c;
    r.m[1][1] = cos1_axis.y * axis.y + c;
    r.m[2][2] = cos1_axis.z * axis.z + c;
    return r;
}

static Affine lookAt(V3 eye, V3 center, V3 up)
{
    // Parser.cpp:142-170
    V3 f = normalized(center - eye);
    if (dot(f, f) <= 1.1920928955e-07f)
        f = V3(0, 0, 1);
    V3 u = normalized(up);
    V3 s = normalized(cross(f, u));
    u    = cross(s, f);
    Affine m;
    for (int i = 0; i < 3; ++i) {
        m.L.m[i][0] = s[i];
        m.L.m[i][1] = u[i];
        m.L.m[i][2] = f[i];
        m.t[i]      = eye[i];
    }
    return m;
}

static void applyTransformOps(Affine& transform, const JsonValue& obj)
{
    for (const auto& kv : obj.obj) {
        const std::string& name = kv.first;
        const JsonValue& val    = kv.second;
        if (name == "translate") {
            Affine t;
            t.t       = getVector3(val, "translate");
            transform = transform * t;
        } else if (name == "scale") {
            Affine s;
            if (val.isNumber()) {
                s.L.m[0][0] = s.L.m[1][1] = s.L.m[2][2] = (float)val.num;
            } else {
                const V3 v  = getVector3(val, "scale");
                s.L.m[0][0] = v.x;
                s.L.m[1][1] = v.y;
                s.L.m[2][2] = v.z;
            }
            transform = transform * s;
        } else if (name == "rotate") {
            const V3 a = getVector3(val, "rotate");
            Affine r;
            r.L       = angleAxis(Deg2Rad * a.x, V3(1, 0, 0)) * angleAxis(Deg2Rad * a.y, V3(0, 1, 0)) * angleAxis(Deg2Rad * a.z, V3(0, 0, 1));
            transform = transform * r;
        } else if (name == "lookat") {
            if (!val.isObject())
                fail("Expected transform lookat property to be an object");
            V3 origin(0, 0, 0), target(0, 1, 0), up(0, 0, 1);
            bool has_dir = false;
            V3 direction;
            for (const auto& kv2 : val.obj) {
                if (kv2.first == "origin")
                    origin = getVector3(kv2.second, "origin");
                else if (kv2.first == "target")
                    target = getVector3(kv2.second, "target");
                else if (kv2.first == "up")
                    up = getVector3(kv2.second, "up");
                else if (kv2.first == "direction") {
                    direction = getVector3(kv2.second, "direction");
                    has_dir   = true;
                }
            }
            transform = transform * lookAt(origin, has_dir ? direction + origin : target, up);
        } else if (name == "matrix") {
            if (!val.isArray())
                fail("Expected transform matrix to be an array");
            transform = transform * fromMatrixArray(val);
        } else if (name == "qrotate") {
            // a quaternion (w, x, y, z), taken as a rotation the way Eigen's Transform *= Quaternion does (Parser.cpp:99-108,195-196)
            if (!val.isArray() || val.arr.size() != 4)
                fail("Expected vector of length 4");
            float q[4];
            for (int i = 0; i < 4; ++i) {
                if (!val.arr[i].isNumber())
                    fail("Given vector is not only numbers");
                q[i] = (float)val.arr[i].num;
            }
            const float w = q[0], x = q[1], y = q[2], z = q[3];
            Affine r;
            r.L.m[0][0] = 1 - 2 * (y * y + z * z), r.L.m[0][1] = 2 * (x * y - w * z), r.L.m[0][2] = 2 * (x * z + w * y);
            r.L.m[1][0] = 2 * (x * y + w * z), r.L.m[1][1] = 1 - 2 * (x * x + z * z), r.L.m[1][2] = 2 * (y * z - w * x);
            r.L.m[2][0] = 2 * (x * z - w * y), r.L.m[2][1] = 2 * (y * z + w * x), r.L.m[2][2] = 1 - 2 * (x * x + y * y);
            transform = transform * r;
        } else {
            fail("Transform property got unknown entry type '" + name + "'");
        }
    }
}

static Affine getTransform(const JsonValue& parent, const char* key = "transform")
{
    const JsonValue* v = parent.find(key);
    if (!v)
        return Affine();
    if (v->isObject()) {
        Affine t;
        applyTransformOps(t, *v);
        return t;
    }
    if (!v->isArray())
        fail("Transform property is neither an array nor an object");
    if (!v->arr.empty() && v->arr[0].isObject()) {
        Affine t;
        for (const auto& e : v->arr) {
            if (!e.isObject())
                fail("Given transform property does not consist of transform operations only");
            applyTransformOps(t, e);
        }
        return t;
    }
    return fromMatrixArray(*v);
}

// The reference evaluates colour properties through PExpr (ShadingTree). Only
// constants are lowered here: a number, [r,g,b], or the literal "color(r,g,b)"
// form the in-tree scenes use; anything else is refused.
// Constant PExpr expressions as the scene exporters write them ("color(r, g, b, a)", "(color(...) * 100.0)", "2 * 0.5"):
// numbers, color(...) / vec3(...), + - * /, unary minus, parentheses; a scalar broadcasts. Anything with variables, textures
// or other functions is not constant and is refused by the caller.
struct ConstExpr {
    const std::string& s;
    size_t pos = 0;
    bool ok    = true;
    struct Val {
        V3 v;
        bool scalar;
    };
    explicit ConstExpr(const std::string& text)
        : s(text)
    {
    }
    bool eat(char c)
    {
        if (pos < s.size() && s[pos] == c) {
            ++pos;
            return true;
        }
        return false;
    }
    Val primary()
    {
        if (eat('(')) {
            const Val v = sum();
            ok &= eat(')');
            return v;
        }
        for (const char* fn : { "color(", "vec3(" })
            if (s.compare(pos, std::strlen(fn), fn) == 0) {
                pos += std::strlen(fn);
                float c[4] = { 0, 0, 0, 1 };
                int n      = 0;
                do {
                    const Val a = sum();
                    ok &= a.scalar && n < 4;
                    if (n < 4)
                        c[n] = a.v.x;
                    ++n;
                } while (ok && eat(','));
                ok &= eat(')') && n >= 3;
                return Val{ V3(c[0], c[1], c[2]), false };
            }
        // the constants and a few of the scalar functions of the expression language (Transpiler.cpp:338-372,644-700)
        struct Named {
            const char* name;
            float value;
        };
        for (const Named& c : { Named{ "Pi", Pi }, Named{ "Eps", 1.1920928955e-07f }, Named{ "E", 2.718281828459045f } }) {
            const size_t n = std::strlen(c.name);
            if (s.compare(pos, n, c.name) == 0 && (pos + n >= s.size() || !(std::isalnum((unsigned char)s[pos + n]) || s[pos + n] == '_' || s[pos + n] == '('))) {
                pos += n;
                return Val{ V3(c.value, c.value, c.value), true };
            }
        }
        for (const char* fn : { "sqrt(", "abs(", "sin(", "cos(", "tan(", "exp(", "log(", "rad(", "deg(", "min(", "max(", "pow(", "clamp(" })
            if (s.compare(pos, std::strlen(fn), fn) == 0) {
                const std::string name(fn, std::strlen(fn) - 1);
                pos += std::strlen(fn);
                float a[3] = { 0, 0, 0 };
                int n      = 0;
                do {
                    const Val v = sum();
                    ok &= v.scalar && n < 3;
                    if (n < 3)
                        a[n] = v.v.x;
                    ++n;
                } while (ok && eat(','));
                const int want = (name == "min" || name == "max" || name == "pow") ? 2 : (name == "clamp" ? 3 : 1);
                ok &= eat(')') && n == want;
                float r = 0;
                if (name == "sqrt") r = std::sqrt(a[0]);
                else if (name == "abs") r = std::fabs(a[0]);
                else if (name == "sin") r = std::sin(a[0]);
                else if (name == "cos") r = std::cos(a[0]);
                else if (name == "tan") r = std::tan(a[0]);
                else if (name == "exp") r = std::exp(a[0]);
                else if (name == "log") r = std::log(a[0]);
                else if (name == "rad") r = a[0] * Deg2Rad;
                else if (name == "deg") r = a[0] / Deg2Rad;
                else if (name == "min") r = std::min(a[0], a[1]);
                else if (name == "max") r = std::max(a[0], a[1]);
                else if (name == "pow") r = std::pow(a[0], a[1]);
                else r = std::min(std::max(a[0], a[1]), a[2]);
                return Val{ V3(r, r, r), true };
            }
        const char* begin = s.c_str() + pos;
        char* end         = nullptr;
        const float f     = std::strtof(begin, &end);
        if (end == begin || !(std::isdigit((unsigned char)*begin) || *begin == '.')) {
            ok = false;
            return Val{ V3(0, 0, 0), true };
        }
        pos += (size_t)(end - begin);
        return Val{ V3(f, f, f), true };
    }
    // `^` is PExpr's power operator: tighter than a sign, right-associative
    Val power()
    {
        const Val a = primary();
        if (ok && eat('^')) {
            const Val b = unary();
            return Val{ V3(std::pow(a.v.x, b.v.x), std::pow(a.v.y, b.v.y), std::pow(a.v.z, b.v.z)), a.scalar && b.scalar };
        }
        return a;
    }
    Val unary()
    {
        if (eat('-')) {
            const Val v = unary();
            return Val{ V3(-v.v.x, -v.v.y, -v.v.z), v.scalar };
        }
        eat('+');
        return power();
    }
    Val product()
    {
        Val a = unary();
        while (ok && pos < s.size() && (s[pos] == '*' || s[pos] == '/')) {
            const char op = s[pos++];
            const Val b   = unary();
            a = op == '*' ? Val{ V3(a.v.x * b.v.x, a.v.y * b.v.y, a.v.z * b.v.z), a.scalar && b.scalar }
                          : Val{ V3(a.v.x / b.v.x, a.v.y / b.v.y, a.v.z / b.v.z), a.scalar && b.scalar };
        }
        return a;
    }
    Val sum()
    {
        Val a = product();
        while (ok && pos < s.size() && (s[pos] == '+' || s[pos] == '-')) {
            const char op = s[pos++];
            const Val b   = product();
            a = op == '+' ? Val{ V3(a.v.x + b.v.x, a.v.y + b.v.y, a.v.z + b.v.z), a.scalar && b.scalar }
                          : Val{ V3(a.v.x - b.v.x, a.v.y - b.v.y, a.v.z - b.v.z), a.scalar && b.scalar };
        }
        return a;
    }
    static bool evaluate(const std::string& text, V3& out)
    {
        std::string compact;
        for (char c : text)
            if (!std::isspace((unsigned char)c))
                compact += c;
        if (compact.empty())
            return false;
        ConstExpr e(compact);
        const Val v = e.sum();
        if (!e.ok || e.pos != compact.size())
            return false;
        out = v.v;
        return true;
    }
};

// The scene's "parameters" while a scene is being built: an expression over them and constants only ("color_point * scale_point",
// scenes/parameter_plane.json) is a constant to this backend, which has no run-time parameter registry
// (registry::get_global_parameter_*, ShadingTree.cpp:30-60).
static thread_local const std::map<std::string, igh::pexpr::Param>* g_scene_params = nullptr;

static bool evaluateWithParameters(const std::string& src, V3& out)
{
    if (!g_scene_params)
        return false;
    try {
        igh::pexpr::Env env;
        env.params                    = *g_scene_params;
        const igh::pexpr::Program prog = igh::pexpr::compile(src, env);
        if (!prog.is_const || prog.type == igh::pexpr::Type::Bool || prog.type == igh::pexpr::Type::Vec2)
            return false;
        out = V3(prog.value[0], prog.value[1], prog.value[2]);
        return true;
    } catch (const std::runtime_error&) {
        return false;
    }
}

static bool parseConstColor(const JsonValue& v, V3& out)
{
    if (v.isNumber()) {
        out = V3((float)v.num, (float)v.num, (float)v.num);
        return true;
    }
    if (v.isArray() && v.arr.size() == 3 && v.arr[0].isNumber() && v.arr[1].isNumber() && v.arr[2].isNumber()) {
        out = V3((float)v.arr[0].num, (float)v.arr[1].num, (float)v.arr[2].num);
        return true;
    }
    if (v.isString())
        return ConstExpr::evaluate(v.str, out) || evaluateWithParameters(v.str, out);
    return false;
}

static V3 getColor(const JsonValue& obj, const std::string& key, V3 def, const std::string& owner)
{
    const JsonValue* v = obj.find(key);
    if (!v)
        return def;
    V3 c;
    if (!parseConstColor(*v, c))
        fail("'" + owner + "': property '" + key + "' is not a constant colour; expressions and textures here are not supported by the HIP backend");
    return c;
}

static float getConstNumber(const JsonValue& obj, const std::string& key, float def, const std::string& owner)
{
    const JsonValue* v = obj.find(key);
    if (!v)
        return def;
    if (v->isString()) { // a constant expression such as "(0.175)^2" (the Blender exporter writes roughness that way)
        V3 c;
        if ((ConstExpr::evaluate(v->str, c) || evaluateWithParameters(v->str, c)) && c.x == c.y && c.y == c.z)
            return c.x;
    }
    if (!v->isNumber())
        fail("'" + owner + "': property '" + key + "' is not a constant number; expressions and textures here are not supported by the HIP backend");
    return (float)v->num;
}

// ---------------------------------------------------------------- scene container

struct ShapeRec {
    std::string name;
    BBox bbox;
    int32_t user1 = 0, user2 = 0; // prim-BVH offset in floats, split (TriMeshProvider.cpp:598)
    std::optional<PlaneShape> plane;
    float area = 0;
    // analytic sphere ("sphere" shapes, SphereProvider.cpp) or a mesh that tessellates one (TriMesh::getAsSphere)
    std::optional<SphereShape> sphere;
    bool analytic = false; // no triangles: intersected as a sphere, lives in the sphere scene BVH
};

struct Scene {
    std::vector<float> entities;
    std::vector<ig_lookup_entry> shape_lookups;
    std::vector<uint8_t> shape_data;
    std::vector<uint8_t> primbvh;
    std::vector<std::pair<size_t, uint32_t>> primbvh_nodes; // where each shape's Node8[] sits inside primbvh (byte offset, count)
    std::vector<ig_node8> scene_nodes;
    std::vector<ig_entity_leaf1> scene_leaves;
    std::vector<ig_node8> sphere_nodes;
    std::vector<ig_entity_leaf1> sphere_leaves;
    std::vector<ig_material> materials;
    std::vector<int32_t> entity_per_material;
    std::vector<ig_light> lights;
    std::vector<float> light_hierarchy;
    std::vector<uint32_t> light_codes;
    std::vector<float> light_cdf;
    std::vector<ig_medium> media;
    std::vector<std::string> entity_names;
    std::vector<std::string> material_names;
    std::vector<ig_texture> textures;
    std::vector<uint8_t> texture_data;
    std::vector<float> cdf_data;
    std::vector<uint32_t> expr_code; // programs of the shading expressions (include/ig_expr.h)
    igd_scene tables{};
};

template <typename T>
static void appendBytes(std::vector<uint8_t>& dst, const T* src, size_t count)
{
    const uint8_t* p = reinterpret_cast<const uint8_t*>(src);
    dst.insert(dst.end(), p, p + sizeof(T) * count);
}

static void padTo(std::vector<uint8_t>& dst, size_t alignment)
{
    // FixTable/DynTable::addEntry semantics (src/runtime/table/FixTable.h:14-23)
    if (alignment != 0 && !dst.empty()) {
        const size_t defect = alignment - dst.size() % alignment;
        dst.resize(dst.size() + defect);
    }
}

static constexpr size_t Pack4Alignment = 16;

static TriMesh loadShapeMesh(const std::string& name, const JsonValue& elem, const std::string& base_dir)
{
    const std::string type = elem.getString("type");
    if (type == "triangle") {
        const V3 p0 = elem.has("p0") ? getVector3(*elem.find("p0"), "p0") : V3(0, 0, 0);
        const V3 p1 = elem.has("p1") ? getVector3(*elem.find("p1"), "p1") : V3(1, 0, 0);
        const V3 p2 = elem.has("p2") ? getVector3(*elem.find("p2"), "p2") : V3(0, 1, 0);
        return TriMesh::MakeTriangle(p0, p1, p2);
    } else if (type == "rectangle") {
        if (!elem.has("p0")) {
            const float width  = elem.getNumber("width", 2.0f);
            const float height = elem.getNumber("height", 2.0f);
            const V3 origin    = elem.has("origin") ? getVector3(*elem.find("origin"), "origin") : V3(-width / 2, -height / 2, 0);
            return TriMesh::MakePlane(origin, V3(1, 0, 0) * width, V3(0, 1, 0) * height);
        }
        const V3 p0 = getVector3(*elem.find("p0"), "p0");
        const V3 p1 = elem.has("p1") ? getVector3(*elem.find("p1"), "p1") : V3(1, -1, 0);
        const V3 p2 = elem.has("p2") ? getVector3(*elem.find("p2"), "p2") : V3(1, 1, 0);
        const V3 p3 = elem.has("p3") ? getVector3(*elem.find("p3"), "p3") : V3(-1, 1, 0);
        return TriMesh::MakeRectangle(p0, p1, p2, p3);
    } else if (type == "cube" || type == "box") {
        const float width  = elem.getNumber("width", 2.0f);
        const float height = elem.getNumber("height", 2.0f);
        const float depth  = elem.getNumber("depth", 2.0f);
        const V3 origin    = elem.has("origin") ? getVector3(*elem.find("origin"), "origin") : V3(-width / 2, -height / 2, -depth / 2);
        return TriMesh::MakeBox(origin, V3(1, 0, 0) * width, V3(0, 1, 0) * height, V3(0, 0, 1) * depth);
    } else if (type == "ply" || type == "obj" || type == "mitsuba" || type == "external") {
        const std::string filename = elem.getString("filename");
        if (filename.empty())
            fail("Shape '" + name + "': No filename given");
        const std::string path = (filename[0] == '/' || base_dir.empty()) ? filename : base_dir + "/" + filename;
        const size_t dot       = path.rfind('.');
        std::string ext        = dot == std::string::npos ? "" : path.substr(dot);
        for (auto& c : ext)
            c = (char)std::tolower((unsigned char)c);
        // ExternalShape dispatch by extension (TriMeshProvider.cpp:41-75)
        if (ext == ".obj" || (type == "obj" && ext != ".ply"))
            return load_obj(path);
        if (ext == ".mts" || ext == ".serialized" || type == "mitsuba")
            return load_serialized(path, (size_t)std::max(0, elem.getInt("shape_index", 0)));
        if (ext != ".ply")
            fail("Shape '" + name + "': only .ply, .obj and Mitsuba serialized external meshes are supported by this loader (got '" + filename + "')");
        return load_ply(path);
    }
    // procedural meshes (TriMeshProvider.cpp:48-103)
    auto vec3Or = [&](const char* key, V3 def) { return elem.has(key) ? getVector3(*elem.find(key), key) : def; };
    if (type == "icosphere")
        return TriMesh::MakeIcoSphere(vec3Or("center", V3(0, 0, 0)), elem.getNumber("radius", 1.0f), (uint32_t)std::max(0, elem.getInt("subdivisions", 4)));
    if (type == "uvsphere")
        return TriMesh::MakeUVSphere(vec3Or("center", V3(0, 0, 0)), elem.getNumber("radius", 1.0f), (uint32_t)std::max(0, elem.getInt("stacks", 32)),
                                     (uint32_t)std::max(0, elem.getInt("slices", 16)));
    if (type == "cylinder") {
        float base_radius, top_radius;
        if (elem.has("radius")) {
            base_radius = top_radius = elem.getNumber("radius", 1.0f);
        } else {
            base_radius = elem.getNumber("bottom_radius", 1.0f);
            top_radius  = elem.getNumber("top_radius", base_radius);
        }
        return TriMesh::MakeCylinder(vec3Or("p0", V3(0, 0, 0)), base_radius, vec3Or("p1", V3(0, 0, 1)), top_radius,
                                     (uint32_t)std::max(0, elem.getInt("sections", 32)), elem.getBool("filled", true));
    }
    if (type == "cone")
        return TriMesh::MakeCone(vec3Or("p0", V3(0, 0, 0)), elem.getNumber("radius", 1.0f), vec3Or("p1", V3(0, 0, 1)),
                                 (uint32_t)std::max(0, elem.getInt("sections", 32)), elem.getBool("filled", true));
    if (type == "disk")
        return TriMesh::MakeDisk(vec3Or("origin", V3(0, 0, 0)), vec3Or("normal", V3(0, 0, 1)), elem.getNumber("radius", 1.0f),
                                 (uint32_t)std::max(0, elem.getInt("sections", 32)));
    if (type == "inline") {
        // TriMeshProvider.cpp:196-292: flat arrays "indices" (3 per face), "vertices" (3 per point), optional "normals", "texcoords"
        auto numbers = [&](const char* key, std::vector<double>& out) {
            const JsonValue* a = elem.find(key);
            if (a && a->isObject()) // typed array property { "type": "integer" | "number", "values": [...] } (Parser.cpp)
                a = a->find("values");
            if (!a || !a->isArray())
                return false;
            for (const auto& v : a->arr) {
                if (v.type != JsonValue::Number)
                    fail("Shape '" + name + "': '" + key + "' must be an array of numbers");
                out.push_back(v.num);
            }
            return true;
        };
        std::vector<double> ind, pos, nrm, tex;
        if (!numbers("indices", ind))
            fail("Shape '" + name + "': No indices given");
        if (ind.size() % 3 != 0)
            fail("Shape '" + name + "': Number of indices not multiple of 3. Only triangular faces are accepted");
        if (!numbers("vertices", pos))
            fail("Shape '" + name + "': No vertices given");
        if (pos.size() % 3 != 0)
            fail("Shape '" + name + "': Number of vertices not multiple of 3");
        const bool has_normals = numbers("normals", nrm), has_tex = numbers("texcoords", tex);
        if (has_normals && nrm.size() != pos.size())
            fail("Shape '" + name + "': Number of normals does not match number of vertices");
        if (has_tex && (tex.size() % 2 != 0 || tex.size() / 2 != pos.size() / 3))
            fail("Shape '" + name + "': Number of texcoords does not match number of vertices");
        TriMesh mesh;
        const size_t points = pos.size() / 3;
        for (size_t i = 0; i < ind.size(); i += 3) {
            for (int k = 0; k < 3; ++k)
                if (ind[i + k] < 0 || ind[i + k] >= (double)points)
                    fail("Shape '" + name + "': vertex index out of range");
            mesh.indices.insert(mesh.indices.end(), { (uint32_t)ind[i], (uint32_t)ind[i + 1], (uint32_t)ind[i + 2], 0u });
        }
        for (size_t i = 0; i < points; ++i)
            mesh.vertices.push_back(V3((float)pos[3 * i], (float)pos[3 * i + 1], (float)pos[3 * i + 2]));
        if (has_normals)
            for (size_t i = 0; i < points; ++i)
                mesh.normals.push_back(V3((float)nrm[3 * i], (float)nrm[3 * i + 1], (float)nrm[3 * i + 2]));
        else
            mesh.computeVertexNormals();
        if (has_tex)
            for (size_t i = 0; i < points; ++i)
                mesh.texcoords.push_back(V2{ (float)tex[2 * i], (float)tex[2 * i + 1] });
        else
            mesh.makeTexCoordsNormalized();
        return mesh;
    }
    fail("Shape '" + name + "': Can not load shape type '" + type + "'");
}

// SphereProvider::handle (src/runtime/shape/SphereProvider.cpp:10-53): centre + radius as four floats in the "shapes" table, a
// bounding box through the six axis extremes, no triangles
static void handleSphere(Scene& sc, std::vector<ShapeRec>& shapes, const std::string& name, const JsonValue& elem)
{
    const V3 origin    = elem.has("center") ? getVector3(*elem.find("center"), "center") : V3(0, 0, 0);
    const float radius = elem.getNumber("radius", 1.0f);
    if (radius <= 0)
        fail("Shape '" + name + "': While loading shape type 'sphere' an invalid radius of " + std::to_string(radius) + " was given");
    BBox bbox;
    for (const V3 axis : { V3(1, 0, 0), V3(0, 1, 0), V3(0, 0, 1) }) {
        bbox.extend(origin + axis * radius);
        bbox.extend(origin - axis * radius);
    }
    bbox.inflate(1e-5f);
    padTo(sc.shape_data, Pack4Alignment);
    sc.shape_lookups.push_back(ig_lookup_entry{ IG_SHAPE_SPHERE, 0, (uint64_t)sc.shape_data.size() });
    const float rec4[4] = { origin.x, origin.y, origin.z, radius };
    appendBytes(sc.shape_data, rec4, 4);
    ShapeRec rec;
    rec.name     = name;
    rec.bbox     = bbox;
    rec.sphere   = SphereShape{ origin, radius };
    rec.analytic = true;
    shapes.push_back(rec);
}

static void handleShape(Scene& sc, std::vector<ShapeRec>& shapes, const std::string& name, const JsonValue& elem, const std::string& base_dir)
{
    if (elem.getString("type") == "sphere")
        return handleSphere(sc, shapes, name, elem);
    TriMesh mesh = loadShapeMesh(name, elem, base_dir);
    if (mesh.vertices.empty())
        fail("Shape '" + name + "': no vertices were generated");
    if (mesh.faceCount() == 0)
        fail("Shape '" + name + "': no indices were generated");

    // User options, TriMeshProvider.cpp:526-545
    if (elem.getBool("flip_normals", false))
        mesh.flipNormals();
    if (elem.getBool("face_normals", false))
        mesh.setupFaceNormalsAsVertexNormals();
    else if (elem.getBool("smooth_normals", false))
        mesh.computeVertexNormals();
    if (elem.getBool("generic_uv", false))
        mesh.makeTexCoordsNormalized();
    const Affine shapeT = getTransform(elem);
    if (!shapeT.isIdentity())
        mesh.transform(shapeT);
    for (const char* unsupported : { "displacement", "subdivision", "refinement" })
        if (elem.has(unsupported))
            fail("Shape '" + name + "': option '" + unsupported + "' is not supported by this loader");

    BBox bbox = mesh.computeBBox();
    bbox.inflate(1e-5f);

    // Prim BVH -> fix table "trimesh_primbvh" (TriMeshProvider.cpp:305-362)
    std::vector<ig_node8> nodes;
    std::vector<ig_tri4> tris;
    build_tri_bvh8(mesh, nodes, tris);

    padTo(sc.primbvh, Pack4Alignment);
    const uint64_t bvh_offset = sc.primbvh.size() / sizeof(float);
    const uint32_t header[4]  = { (uint32_t)nodes.size(), (uint32_t)tris.size(), 0, 0 };
    appendBytes(sc.primbvh, header, 4);
    sc.primbvh_nodes.emplace_back(sc.primbvh.size(), (uint32_t)nodes.size());
    appendBytes(sc.primbvh, nodes.data(), nodes.size());
    appendBytes(sc.primbvh, tris.data(), tris.size());

    // Mesh -> dyn table "shapes" (TriMeshProvider.cpp:575-596)
    padTo(sc.shape_data, Pack4Alignment);
    sc.shape_lookups.push_back(ig_lookup_entry{ IG_SHAPE_TRIMESH, 0, (uint64_t)sc.shape_data.size() });
    const uint32_t mheader[4] = { (uint32_t)mesh.faceCount(), (uint32_t)mesh.vertices.size(), (uint32_t)mesh.normals.size(), (uint32_t)mesh.texcoords.size() };
    appendBytes(sc.shape_data, mheader, 4);
    const float bb[8] = { bbox.min.x, bbox.min.y, bbox.min.z, 0, bbox.max.x, bbox.max.y, bbox.max.z, 0 };
    appendBytes(sc.shape_data, bb, 8);
    for (const auto& v : mesh.vertices) {
        const float f[4] = { v.x, v.y, v.z, 0 };
        appendBytes(sc.shape_data, f, 4);
    }
    for (const auto& n : mesh.normals) {
        const float f[4] = { n.x, n.y, n.z, 0 };
        appendBytes(sc.shape_data, f, 4);
    }
    appendBytes(sc.shape_data, mesh.indices.data(), mesh.indices.size());
    for (const auto& t : mesh.texcoords) {
        const float f[2] = { t.x, t.y };
        appendBytes(sc.shape_data, f, 2);
    }

    ShapeRec rec;
    rec.name  = name;
    rec.bbox  = bbox;
    rec.user1 = (int32_t)(uint32_t)(bvh_offset & 0xFFFFFFFFu);
    rec.user2 = (int32_t)(uint32_t)((bvh_offset >> 32) & 0xFFFFFFFFu);
    rec.plane = mesh.getAsPlane();
    rec.area  = mesh.computeArea();
    if (!rec.plane.has_value())
        rec.sphere = mesh.getAsSphere(); // TriMeshProvider.cpp:561: "if not a plane, it might be a simple sphere"
    shapes.push_back(rec);
}

static void textureCoordinates(const JsonValue& tex, std::string& u, std::string& v);

// A colour property that names a checkerboard texture (src/runtime/pattern/CheckerBoardPattern.cpp:13-33,
// src/artic/texture/checkerboard.art) is lowered into the material record; other textures are refused.
static bool lowerCheckerboard(const JsonValue& prop, const JsonValue& textures, ig_material& m, const std::string& owner)
{
    if (!prop.isString())
        return false;
    {
        // The exporters' inline form "select(checkerboard(uvw * S) == 1, A, B)": node_checkerboard3 (texture/checkerboard.art:2)
        // with w = 0 is 1 exactly where the parities of floor(S u) and floor(S v) differ, i.e. make_checkerboard_texture with
        // scale (S, S), color0 = A, color1 = B.
        std::string e;
        for (char c : prop.str)
            if (!std::isspace((unsigned char)c))
                e += c;
        const std::string head = "select(checkerboard(uvw*";
        if (e.compare(0, head.size(), head) == 0 && e.back() == ')') {
            char* end         = nullptr;
            const float scale = std::strtof(e.c_str() + head.size(), &end);
            const std::string mid = ")==1,";
            const size_t at       = (size_t)(end - e.c_str());
            if (end != e.c_str() + head.size() && e.compare(at, mid.size(), mid) == 0) {
                const std::string args = e.substr(at + mid.size(), e.size() - at - mid.size() - 1);
                int depth = 0;
                size_t comma = std::string::npos;
                for (size_t i = 0; i < args.size(); ++i) {
                    depth += args[i] == '(' ? 1 : (args[i] == ')' ? -1 : 0);
                    if (args[i] == ',' && depth == 0) {
                        comma = i;
                        break;
                    }
                }
                V3 a, b;
                if (comma != std::string::npos && ConstExpr::evaluate(args.substr(0, comma), a) && ConstExpr::evaluate(args.substr(comma + 1), b)) {
                    m.flags |= IG_MAT_CHECKER;
                    m.q[0] = a.x, m.q[1] = a.y, m.q[2] = a.z;
                    m.q[3] = b.x, m.q[4] = b.y, m.q[5] = b.z;
                    m.q[6] = m.q[7] = scale;
                    return true;
                }
            }
        }
    }
    for (const auto& t : textures.arr) {
        if (t.getString("name") != prop.str)
            continue;
        if (t.getString("type") != "checkerboard")
            fail("'" + owner + "': texture '" + prop.str + "' of type '" + t.getString("type") + "' is not supported by the HIP backend here");
        if (t.has("transform")) {
            std::string tu, tv;
            textureCoordinates(t, tu, tv);
            if (tu != "uv.x" || tv != "uv.y") // (colour properties take the expression route, lowerColor)
                fail("'" + owner + "': a checkerboard under a transform is not supported by this loader here");
        }
        const V3 c0 = getColor(t, "color0", V3(0, 0, 0), prop.str);
        const V3 c1 = getColor(t, "color1", V3(1, 1, 1), prop.str);
        m.flags |= IG_MAT_CHECKER;
        m.q[0] = c0.x, m.q[1] = c0.y, m.q[2] = c0.z;
        m.q[3] = c1.x, m.q[4] = c1.y, m.q[5] = c1.z;
        m.q[6] = getConstNumber(t, "scale_x", 2.0f, prop.str);
        m.q[7] = getConstNumber(t, "scale_y", 2.0f, prop.str);
        return true;
    }
    return false;
}

// Bitmap textures ("image" / "bitmap", LoaderTexture.cpp:65-66) loaded on first use into the packed form the
// reference uploads for 8-bit files (ImagePattern.cpp:17-75, Image::loadAsPacked Image.cpp:714-808).
struct TextureBank {
    const JsonValue& defs;
    std::string base_dir;
    std::vector<ig_texture>& recs;
    std::vector<uint8_t>& data;
    std::map<std::string, int> ids;
    // shading expressions: compiled once per use into Scene::expr_code (pexpr.h)
    std::vector<uint32_t>* expr_code = nullptr;
    std::map<std::string, igh::pexpr::Param> params; // the scene's "parameters" section

    bool has(const std::string& tex_name) const
    {
        for (const auto& t : defs.arr)
            if (t.getString("name") == tex_name)
                return true;
        return false;
    }
    std::map<std::string, igh::pexpr::Param> local_params; // the custom variables of the "expr" texture being lowered (lowerColor)
    igh::pexpr::Program compileExpr(const std::string& src, const std::string& owner)
    {
        igh::pexpr::Env env;
        env.params  = params;
        for (const auto& lp : local_params)
            env.params[lp.first] = lp.second;
        env.texture = [&](const std::string& n) { return has(n) ? get(n, owner) : -1; };
        try {
            return igh::pexpr::compile(src, env);
        } catch (const std::runtime_error& e) {
            fail("'" + owner + "': expression \"" + src + "\": " + e.what());
        }
    }
    int32_t addProgram(const igh::pexpr::Program& prog)
    {
        const int32_t at = (int32_t)expr_code->size();
        expr_code->insert(expr_code->end(), prog.code.begin(), prog.code.end());
        return at;
    }

    static uint8_t toLinear(uint8_t c)
    {
        // byte_color_to_linear (Image.cpp:40-51)
        const float v = c / 255.0f;
        const float l = v <= 0.04045f ? v / 12.92f : std::pow((v + 0.055f) / 1.055f, 2.4f);
        return (uint8_t)std::min<uint16_t>(255, (uint16_t)std::floor(l * 255));
    }
    static uint32_t wrapMode(const std::string& s) { return s == "mirror" ? IG_WRAP_MIRROR : (s == "clamp" ? IG_WRAP_CLAMP : IG_WRAP_REPEAT); }

    int get(const std::string& tex_name, const std::string& owner)
    {
        auto it = ids.find(tex_name);
        if (it != ids.end())
            return it->second;
        const JsonValue* def = nullptr;
        for (const auto& t : defs.arr)
            if (t.getString("name") == tex_name)
                def = &t;
        if (!def)
            fail("'" + owner + "': unknown texture '" + tex_name + "'");
        const std::string type = def->getString("type");
        if (type != "image" && type != "bitmap")
            fail("'" + owner + "': texture '" + tex_name + "' of type '" + type + "' is not supported by the HIP backend here");
        if (def->has("transform"))
            fail("Texture '" + tex_name + "': image transforms are not supported by this loader");
        if (def->getBool("force_unpacked", false))
            fail("Texture '" + tex_name + "': force_unpacked is not supported by this loader");
        const std::string filename = def->getString("filename");
        if (filename.empty())
            fail("Texture '" + tex_name + "': no filename");
        const std::string path = (filename[0] == '/' || base_dir.empty()) ? filename : base_dir + "/" + filename;
        const bool is_float = isFloatImagePath(path);
        std::string lower = path;
        for (char& ch : lower)
            ch = (char)std::tolower((unsigned char)ch);
        auto ends_with = [&](const char* e) { return lower.size() >= std::strlen(e) && lower.compare(lower.size() - std::strlen(e), std::strlen(e), e) == 0; };
        const bool is_jpeg = ends_with(".jpg") || ends_with(".jpeg");
        if (!is_float && !is_jpeg && !ends_with(".png"))
            fail("Texture '" + tex_name + "': only PNG, JPEG, OpenEXR and Radiance HDR files are supported by this loader");
        PngImage png;
        FloatImage fimg;
        if (is_float) {
            fimg = readFloatImage(path); // Image::load (Image.cpp:497-712): kept as floats, never packed (Image.cpp:717-721)
        } else if (is_jpeg) {
            JpegImage j  = readJpeg(path); // packed like any other 8-bit file (Image::loadAsPacked, Image.cpp:714-808)
            png.width    = j.width;
            png.height   = j.height;
            png.channels = j.channels;
            png.data     = std::move(j.data);
        } else {
            png = readPng(path);
        }
        const bool linear = def->getBool("linear", false);

        ig_texture rec{};
        rec.width  = is_float ? fimg.width : png.width;
        rec.height = is_float ? fimg.height : png.height;
        // ImagePattern.cpp:25-29: anything but "bilinear" / "nearest" (e.g. the scenes' "trilinear") is bicubic
        const std::string filter = def->getString("filter_type", "bicubic");
        rec.filter               = filter == "bilinear" ? IG_TEX_BILINEAR : (filter == "nearest" ? IG_TEX_NEAREST : IG_TEX_BICUBIC);
        if (def->has("wrap_mode_u")) {
            rec.wrap_u = wrapMode(def->getString("wrap_mode_u", "repeat"));
            rec.wrap_v = wrapMode(def->getString("wrap_mode_v", "repeat"));
        } else {
            rec.wrap_u = rec.wrap_v = wrapMode(def->getString("wrap_mode", "repeat"));
        }
        data.resize((data.size() + 15) & ~(size_t)15);
        rec.offset = data.size();
        // rows bottom to top (stbi_set_flip_vertically_on_load, Image.cpp:724)
        const size_t n = (size_t)rec.width * rec.height;
        if (is_float) {
            // 32-bit floats, one or four per texel, no colour-space conversion; Image::flipY (Image.cpp:108-121,709)
            rec.channels = IG_TEX_FLOAT_BIT | fimg.channels;
            const size_t row = (size_t)rec.width * fimg.channels * sizeof(float);
            data.resize(rec.offset + n * fimg.channels * sizeof(float));
            for (uint32_t y = 0; y < rec.height; ++y)
                std::memcpy(&data[rec.offset + (size_t)y * row], &fimg.pixels[(size_t)(rec.height - 1 - y) * rec.width * fimg.channels], row);
        } else if (png.channels == 1) {
            rec.channels = 1;
            data.resize(rec.offset + n);
            for (uint32_t y = 0; y < png.height; ++y)
                for (uint32_t x = 0; x < png.width; ++x) {
                    const uint8_t v = png.data[(size_t)(png.height - 1 - y) * png.width + x];
                    data[rec.offset + (size_t)y * png.width + x] = linear ? v : toLinear(v);
                }
        } else {
            // 3 channels get alpha 255; 2 (gray + alpha) is re-requested as RGBA from stb (Image.cpp:731-736)
            rec.channels = 4;
            data.resize(rec.offset + n * 4);
            for (uint32_t y = 0; y < png.height; ++y)
                for (uint32_t x = 0; x < png.width; ++x) {
                    const uint8_t* p = &png.data[((size_t)(png.height - 1 - y) * png.width + x) * png.channels];
                    uint8_t r, g, b, a;
                    if (png.channels == 2)
                        r = g = b = p[0], a = p[1];
                    else
                        r = p[0], g = p[1], b = p[2], a = png.channels == 4 ? p[3] : 255;
                    uint8_t* o = &data[rec.offset + ((size_t)y * png.width + x) * 4];
                    o[0] = linear ? r : toLinear(r), o[1] = linear ? g : toLinear(g), o[2] = linear ? b : toLinear(b), o[3] = a;
                }
        }
        const int id = (int)recs.size();
        recs.push_back(rec);
        ids[tex_name] = id;
        return id;
    }
};

// ---- sun position (LoaderUtils::getEA, src/runtime/loader/LoaderUtils.cpp:66-105)
struct SunAngles {
    float elevation, azimuth; // radians; azimuth west of south (skysun/ElevationAzimuth.h)
    V3 direction() const      // toDirectionYUp (ElevationAzimuth.h:22-30)
    {
        return V3(-std::cos(elevation) * std::sin(azimuth), std::sin(elevation), -std::cos(elevation) * std::cos(azimuth));
    }
};

// The PSA solar position algorithm (Blanco-Muriel et al., "Computing the Solar Vector", Solar Energy 70(5), 2001) as the
// reference evaluates it (skysun/SunLocation.cpp:20-37,58-125): single-precision Julian date and decimal hours, the rest
// in double; local time = UTC - timezone, longitude in degrees west.
static SunAngles sunFromTimeAndPlace(int year, int month, int day, int hour, int minute, float seconds, float latitude, float longitude, float timezone)
{
    const float dec_hours = hour + timezone + (minute + seconds / 60.0f) / 60.0f;
    const int m14         = (month - 14) / 12;
    const int jdn = (1461 * (year + 4800 + m14)) / 4 + (367 * (month - 2 - 12 * m14)) / 12 - (3 * ((year + 4900 + m14) / 100)) / 4 + day - 32075;
    const float julian = (float)jdn - 0.5f + dec_hours / 24.0f;
    const double n     = julian - 2451545.0f; // days since noon, 1 January 2000
    const double hours = dec_hours;

    const double omega     = 2.1429 - 0.0010394594 * n;
    const double mean_long = 4.8950630 + 0.017202791698 * n;
    const double anomaly   = 6.2400600 + 0.0172019699 * n;
    const double ecl_long  = mean_long + 0.03341607 * std::sin(anomaly) + 0.00034894 * std::sin(2 * anomaly) - 0.0001134 - 0.0000203 * std::sin(omega);
    const double obliquity = 0.4090928 - 6.2140e-9 * n + 0.0000396 * std::cos(omega);

    const double sin_long = std::sin(ecl_long);
    double right_asc      = std::atan2(std::cos(obliquity) * sin_long, std::cos(ecl_long));
    if (right_asc < 0)
        right_asc += 2 * Pi;
    const double declination = std::asin(std::sin(obliquity) * sin_long);

    const double gmst       = 6.6974243242 + 0.0657098283 * n + hours;
    const double lmst       = Deg2Rad * ((float)(gmst * 15 - longitude));
    const double lat        = Deg2Rad * latitude;
    const double hour_angle = lmst - right_asc;
    double zenith           = std::acos(std::cos(lat) * std::cos(hour_angle) * std::cos(declination) + std::sin(declination) * std::sin(lat));
    double azimuth          = std::atan2(-std::sin(hour_angle), std::tan(declination) * std::cos(lat) - std::sin(lat) * std::cos(hour_angle));
    if (azimuth < 0)
        azimuth += 2 * Pi;
    zenith += (6371.01 / 149597890.0) * std::sin(zenith); // parallax
    return SunAngles{ Pi / 2 - (float)zenith, std::fmod((float)azimuth + Pi, 2 * Pi) };
}

// direction / sun_direction (through elevation and azimuth, as the reference does), elevation + azimuth, or time and place
static SunAngles getSunAngles(const JsonValue& l)
{
    const char* key = l.has("direction") ? "direction" : (l.has("sun_direction") ? "sun_direction" : nullptr);
    if (key) {
        V3 d           = getVector3(*l.find(key), key);
        const float dl = std::sqrt(dot(d, d));
        d              = dl > 0 ? d * (1 / dl) : V3(0, 0, 1);
        float azimuth  = std::atan2(-d.x, -d.z); // fromDirectionYUp (ElevationAzimuth.h:15-20)
        if (azimuth < 0)
            azimuth += 2 * Pi;
        return SunAngles{ Pi / 2 - std::acos(d.y), azimuth };
    }
    if (l.has("elevation") || l.has("azimuth"))
        return SunAngles{ l.getNumber("elevation", 0.0f), l.getNumber("azimuth", 0.0f) };
    return sunFromTimeAndPlace(l.getInt("year", 2020), l.getInt("month", 5), l.getInt("day", 6), l.getInt("hour", 12), l.getInt("minute", 0),
                               l.getNumber("seconds", 0.0f), l.getNumber("latitude", 49.235422f), l.getNumber("longitude", -6.9965744f),
                               l.getNumber("timezone", -2.0f));
}

// ---- Perez all-weather sky (src/artic/light/perez.art): the part the reference evaluates once per light — the model from
// (brightness, clearness) or irradiances, the explicit parameters (a, b, c, d, e), the normalisation against the integral over
// Radiance's 145 sky patches and the ground / sun radiances of make_perez_light_from_model. Coefficients: R. Perez, R. Seals,
// J. Michalsky, "All-weather model for sky luminance distribution", Solar Energy 50(3), 1993 (the tables of gendaylit).
namespace perez {
constexpr float SolarE = 1367, SolarL = 127500, PreciWater = 2;
constexpr int Bins     = 8;
const float Ranges[9]  = { 1.000f, 1.065f, 1.230f, 1.500f, 1.950f, 2.800f, 4.500f, 6.200f, 12.01f };
const float PA[32] = { 1.3525f, -0.2576f, -0.2690f, -1.4366f, -1.2219f, -0.7730f, 1.4148f, 1.1016f, -1.1000f, -0.2515f, 0.8952f, 0.0156f,
                       -0.5484f, -0.6654f, -0.2672f, 0.7117f, -0.6000f, -0.3566f, -2.5000f, 2.3250f, -1.0156f, -0.3670f, 1.0078f, 1.4051f,
                       -1.0000f, 0.0211f, 0.5025f, -0.5119f, -1.0500f, 0.0289f, 0.4260f, 0.3590f };
const float PB[32] = { -0.7670f, 0.0007f, 1.2734f, -0.1233f, -0.2054f, 0.0367f, -3.9128f, 0.9156f, 0.2782f, -0.1812f, -4.5000f, 1.1766f,
                       0.7234f, -0.6219f, -5.6812f, 2.6297f, 0.2937f, 0.0496f, -5.6812f, 1.8415f, 0.2875f, -0.5328f, -3.8500f, 3.3750f,
                       -0.3000f, 0.1922f, 0.7023f, -1.6317f, -0.3250f, 0.1156f, 0.7781f, 0.0025f };
const float PC[32] = { 2.8000f, 0.6004f, 1.2375f, 1.0000f, 6.9750f, 0.1774f, 6.4477f, -0.1239f, 24.7219f, -13.0812f, -37.7000f, 34.8438f,
                       33.3389f, -18.3000f, -62.2500f, 52.0781f, 21.0000f, -4.7656f, -21.5906f, 7.2492f, 14.0000f, -0.9999f, -7.1406f, 7.5469f,
                       19.0000f, -5.0000f, 1.2438f, -1.9094f, 31.0625f, -14.5000f, -46.1148f, 55.3750f };
const float PD[32] = { 1.8734f, 0.6297f, 0.9738f, 0.2809f, -1.5798f, -0.5081f, -1.7812f, 0.1080f, -5.0000f, 1.5218f, 3.9229f, -2.6204f,
                       -3.5000f, 0.0016f, 1.1477f, 0.1062f, -3.5000f, -0.1554f, 1.4062f, 0.3988f, -3.4000f, -0.1078f, -1.0750f, 1.5702f,
                       -4.0000f, 0.0250f, 0.3844f, 0.2656f, -7.2312f, 0.4050f, 13.3500f, 0.6234f };
const float PE[32] = { 0.0356f, -0.1246f, -0.5718f, 0.9938f, 0.2624f, 0.0672f, -0.2190f, -0.4285f, -0.0156f, 0.1597f, 0.4199f, -0.5562f,
                       0.4659f, -0.3296f, -0.0876f, -0.0329f, 0.0032f, 0.0766f, -0.0656f, -0.1294f, -0.0672f, 0.4016f, 0.3017f, -0.4844f,
                       1.0468f, -0.3788f, -2.4517f, 1.4656f, 1.5000f, -0.6426f, 1.8564f, 0.5636f };
const float DiffuseEff[4][8] = { { 97.24f, 107.22f, 104.97f, 102.39f, 100.71f, 106.42f, 141.88f, 152.23f },
                                 { -0.46f, 1.15f, 2.96f, 5.59f, 5.94f, 3.83f, 1.90f, 0.35f },
                                 { 12.00f, 0.59f, -5.53f, -13.95f, -22.75f, -36.15f, -53.24f, -45.27f },
                                 { -8.91f, -3.95f, -8.77f, -13.90f, -23.74f, -28.83f, -14.03f, -7.98f } };
const float DirectEff[4][8]  = { { 57.20f, 98.99f, 109.83f, 110.34f, 106.36f, 107.19f, 105.75f, 101.18f },
                                 { -4.55f, -3.46f, -4.90f, -5.84f, -3.97f, -1.25f, 0.77f, 1.58f },
                                 { -2.98f, -1.21f, -1.71f, -1.99f, -1.75f, -1.51f, -1.26f, -1.10f },
                                 { 117.12f, 12.38f, -8.81f, -4.56f, -6.16f, -26.73f, -34.44f, -8.29f } };

struct Model {
    float brightness, clearness, direct_irrad, diffuse_irrad, direct_illum, diffuse_illum;
    float params[5];
};

static float clampf(float v, float lo, float hi) { return std::min(std::max(v, lo), hi); }
static float eccentricity(float day)
{
    const float a = 2 * Pi * clampf(day / 365, 0.0f, 1.0f); // perez.art:168-171
    return 1.00011f + 0.034221f * std::cos(a) + 0.00128f * std::sin(a) + 0.000719f * std::cos(2 * a) + 0.000077f * std::sin(2 * a);
}
static float airMass(float zenith) { return 1 / (std::cos(zenith) + 0.15f * std::exp(std::log(93.885f - zenith / Deg2Rad) * -1.253f)); } // :173
static int category(float clearness)                                                                                                     // :175-182
{
    for (int bin = 0; bin < Bins; ++bin)
        if (clearness >= Ranges[bin] && clearness < Ranges[bin + 1])
            return bin;
    return Bins - 1;
}
static void explicitParameters(float brightness, float clearness, float zenith, float out[5]) // :198-233
{
    if (clearness > 1.065f && clearness < 2.8f && brightness < 0.2f)
        brightness = 0.2f;
    auto std4 = [&](const float* p) { return p[0] + p[1] * zenith + brightness * (p[2] + p[3] * zenith); };
    const int bin = category(clearness);
    out[0] = std4(&PA[bin * 4]);
    out[1] = std4(&PB[bin * 4]);
    out[4] = std4(&PE[bin * 4]);
    if (bin == 0) {
        out[2] = std::exp(std::pow(brightness * (PC[0] + PC[1] * zenith), PC[2])) - PC[3];
        out[3] = -std::exp(brightness * (PD[0] + PD[1] * zenith)) + PD[2] + brightness * PD[3];
    } else {
        out[2] = std4(&PC[bin * 4]);
        out[3] = std4(&PD[bin * 4]);
    }
}
static Model finish(float brightness, float clearness, float direct_irrad, float diffuse_irrad, float zenith)
{
    Model m{ brightness, clearness, direct_irrad, diffuse_irrad, 0, 0, {} };
    const int bin = category(clearness);
    // compute_direct_efficacy / compute_diffuse_efficacy (:184-196)
    m.direct_illum  = direct_irrad * std::max(0.0f, DirectEff[0][bin] + DirectEff[1][bin] * PreciWater + DirectEff[2][bin] * std::exp(5.73f * zenith - 5) + DirectEff[3][bin] * brightness);
    m.diffuse_illum = diffuse_irrad * (DiffuseEff[0][bin] + DiffuseEff[1][bin] * PreciWater + DiffuseEff[2][bin] * std::cos(zenith) + DiffuseEff[3][bin] * std::log(brightness));
    explicitParameters(brightness, clearness, zenith, m.params);
    return m;
}
// make_model_from_brightness_clearness (gendaylit -P, :94-111)
static Model fromBrightnessClearness(float brightness, float clearness, float zenith, float day)
{
    brightness            = clampf(brightness, 0.01f, 0.6f);
    clearness             = clampf(clearness, 1.0f, 12.0f - 0.001f);
    const float diffuse0  = brightness * SolarE * eccentricity(day) / airMass(zenith);
    const float c         = 1.041f * zenith * zenith * zenith;
    const float direct    = clampf((clearness * (1 + c) - c) * diffuse0 - diffuse0, 0.0f, SolarE);
    return finish(brightness, clearness, direct, std::max(diffuse0, 0.0f), zenith);
}
// make_model_from_irradiance (gendaylit -W, :113-130); the two check_ functions are applied crosswise, as written there
static Model fromIrradiance(float diffuse_irrad, float direct_irrad, float zenith, float day)
{
    diffuse_irrad          = clampf(diffuse_irrad, 0.0f, SolarE);
    direct_irrad           = std::max(direct_irrad, 0.0f);
    const float brightness = clampf(diffuse_irrad * airMass(zenith) / (SolarE * eccentricity(day)), 0.01f, 0.6f);
    const float c          = 1.041f * zenith * zenith * zenith;
    const float clearness  = clampf(((diffuse_irrad + direct_irrad) / diffuse_irrad + c) / (1 + c), 1.0f, 12.0f - 0.001f);
    return finish(brightness, clearness, direct_irrad, diffuse_irrad, zenith);
}
// perez::eval (:235-242)
static float eval(float cos_sun, float cos_theta, const float p[5])
{
    const float sun_a = std::acos(cos_sun);
    const float A     = 1 + p[0] * std::exp(p[1] / std::max(1e-5f, cos_theta));
    const float B     = 1 + p[2] * std::exp(p[3] * sun_a) + p[4] * cos_sun * cos_sun;
    return A * B;
}
// perez::integrate (:267-290) over the Tregenza patch centres (:244-265): 30, 30, 24, 24, 18, 12, 6 patches on rings
// 12 degrees apart, then the zenith patch
static float integrate(float zenith, const float p[5])
{
    const float cs = std::cos(zenith), sn = std::sin(zenith);
    const int ring_count[8] = { 30, 30, 24, 24, 18, 12, 6, 1 };
    float sum               = 0;
    for (int r = 0; r < 8; ++r) {
        const float theta = Deg2Rad * (float)(84 - 12 * r);
        const float ct = std::cos(theta), st = std::sin(theta);
        for (int k = 0; k < ring_count[r]; ++k) {
            const float phi     = Deg2Rad * (float)(k * (360 / ring_count[r]));
            const float cos_sun = std::min(1.0f, cs * ct + sn * st * std::cos(phi));
            sum += eval(cos_sun, ct, p) * ct;
        }
    }
    return 2 * Pi * sum / 145;
}
static int dayOfTheYear(int year, int month, int day) // TimePoint::dayOfTheYear (skysun/SunLocation.cpp:7-18): tm_yday, 0-based
{
    static const int before[12] = { 0, 31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334 };
    const bool leap             = (year % 4 == 0 && year % 100 != 0) || year % 400 == 0;
    const int m                 = std::min(std::max(month, 1), 12);
    return before[m - 1] + (leap && m > 2 ? 1 : 0) + day - 1;
}
} // namespace perez

// ---- environment maps: texture baking and the sampling tables
// One lookup of a packed bitmap texture the way the device does it (src/artic/texture/image.art:9-156 with the identity
// transform): used to bake the radiance of an environment light (src/artic/entrypoints/bake.art:1-26).
static V3 lookupTexture(const ig_texture& t, const std::vector<uint8_t>& data, float u, float v)
{
    const int W = (int)t.width, H = (int)t.height;
    auto border = [](uint32_t mode, int x, int w) {
        if (mode == IG_WRAP_CLAMP)
            return std::min(std::max(x, 0), w - 1);
        if (mode == IG_WRAP_MIRROR) {
            const int a = x < 0 ? -1 - x : x;
            const int k = a % w;
            return ((a / w) & 1) == 0 ? w - 1 - k : k;
        }
        const int r = x % w;
        return r < 0 ? r + w : r;
    };
    auto texel = [&](int x, int y) {
        x = border(t.wrap_u, x, W);
        y = border(t.wrap_v, y, H);
        const uint8_t* p = &data[t.offset];
        if (t.channels & IG_TEX_FLOAT_BIT) {
            const uint32_t nc = t.channels & 0xFFu;
            float c[4];
            std::memcpy(c, p + ((size_t)y * W + x) * nc * sizeof(float), nc * sizeof(float));
            return nc == 1 ? V3(c[0], c[0], c[0]) : V3(c[0], c[1], c[2]);
        }
        if (t.channels == 1) {
            const float g = (float)p[(size_t)y * W + x] / 255;
            return V3(g, g, g);
        }
        p += ((size_t)y * W + x) * 4;
        return V3((float)p[0] / 255, (float)p[1] / 255, (float)p[2] / 255);
    };
    if (t.filter == IG_TEX_NEAREST)
        return texel((int)std::floor(u * (float)W), (int)std::floor(v * (float)H));
    const float pu = u * (float)W - 0.5f, pv = v * (float)H - 0.5f;
    const int ix = (int)std::floor(pu), iy = (int)std::floor(pv);
    const float fx = pu - std::floor(pu), fy = pv - std::floor(pv);
    auto mix = [](V3 a, V3 b, float k) { return a * (1 - k) + b * k; };
    if (t.filter == IG_TEX_BILINEAR)
        return mix(mix(texel(ix, iy), texel(ix + 1, iy), fx), mix(texel(ix, iy + 1), texel(ix + 1, iy + 1), fx), fy);
    // bicubic B-spline from four bilinear-like taps
    auto w0 = [](float a) { return (a * (a * (-a + 3) - 3) + 1) / 6; };
    auto w1 = [](float a) { return (a * a * (3 * a - 6) + 4) / 6; };
    auto w2 = [](float a) { return (a * (a * (-3 * a + 3) + 3) + 1) / 6; };
    auto w3 = [](float a) { return (a * a * a) / 6; };
    const float g0x = w0(fx) + w1(fx), g1x = w2(fx) + w3(fx), g0y = w0(fy) + w1(fy), g1y = w2(fy) + w3(fy);
    const int x0 = (int)std::floor((float)ix + (w1(fx) / g0x - 1) + 0.5f), x1 = (int)std::floor((float)ix + (w3(fx) / g1x + 1) + 0.5f);
    const int y0 = (int)std::floor((float)iy + (w1(fy) / g0y - 1) + 0.5f), y1 = (int)std::floor((float)iy + (w3(fy) / g1y + 1) + 0.5f);
    return (texel(x0, y0) * (g0x * g0y) + texel(x1, y0) * (g1x * g0y)) + (texel(x0, y1) * (g0x * g1y) + texel(x1, y1) * (g1x * g1y));
}

// CDF::computeForImage (src/runtime/CDF.cpp:43-150): marginal over rows (weighted by sin(theta)), one conditional per
// row, MIS compensation (the mean response is subtracted unless the image is constant). Appends height + width * height
// floats to `out` and returns the offset of the table.
static uint32_t appendEnvironmentCdf(std::vector<float>& out, const std::vector<float>& rgb, size_t width, size_t height, bool compensate)
{
    constexpr float MinEps = 1e-5f;
    auto response = [](float r, float g, float b) { return (std::max(r, 0.0f) + std::max(g, 0.0f) + std::max(b, 0.0f)) / 3; };
    float defect = 0;
    if (compensate) {
        float lowest = std::numeric_limits<float>::infinity();
        for (size_t i = 0; i < width * height; ++i) {
            const float r = response(rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2]);
            lowest        = std::min(lowest, r);
            defect += r / (float)width;
        }
        defect /= (float)height;
        if (std::abs(lowest - defect) < 1e-4f)
            defect = 0;
    }
    const size_t offset = out.size();
    out.resize(offset + height + width * height);
    float* marginal    = &out[offset];
    float* conditional = &out[offset + height];
    for (size_t y = 0; y < height; ++y) {
        const float* p = &rgb[y * width * 3];
        float* cond    = &conditional[y * width];
        cond[0]        = response(p[0] - defect, p[1] - defect, p[2] - defect);
        for (size_t x = 1; x < width; ++x)
            cond[x] = cond[x - 1] + response(p[x * 3] - defect, p[x * 3 + 1] - defect, p[x * 3 + 2] - defect);
        const float sum = cond[width - 1];
        marginal[y]     = sum * std::sin(Pi * ((float)y + 0.5f) / (float)height);
        if (sum > MinEps) {
            const float n = 1.0f / sum;
            for (size_t x = 0; x < width; ++x)
                cond[x] *= n;
        } else {
            const float n = 1.0f / (float)width;
            for (size_t x = 1; x < width; ++x)
                cond[x - 1] = (float)x * n;
        }
        cond[width - 1] = 1;
    }
    for (size_t y = 1; y < height; ++y)
        marginal[y] += marginal[y - 1];
    if (marginal[height - 1] > MinEps) {
        const float n = 1.0f / marginal[height - 1];
        for (size_t y = 0; y < height; ++y)
            marginal[y] *= n;
    } else {
        const float n = 1.0f / (float)height;
        for (size_t y = 1; y < height; ++y)
            marginal[y - 1] = (float)y * n;
    }
    marginal[height - 1] = 1;
    return (uint32_t)offset;
}

// The "transform" of a procedural texture as the reference applies it to the surface coordinates: LoaderUtils::inlineTransformAs2d
// (src/runtime/loader/LoaderUtils.cpp:40-64) keeps the upper-left 2 x 2 block and the x / y translation of the 3D transform, WRITES the nine
// numbers into the generated shader with a stream's default precision (six significant digits: restated by printing and re-reading), and
// mat3x3_transform_point_affine (src/artic/core/matrix.art:237-240) is vec3_dot(row, (u, v, 1)) per coordinate — `dot` of the expression
// language is that fma chain (include/ig_expr.h IGE_DOT). An identity matrix (Eigen's isIdentity, precision 1e-5) is left out.
static void textureCoordinates(const JsonValue& tex, std::string& u, std::string& v)
{
    u = "uv.x", v = "uv.y";
    if (!tex.has("transform"))
        return;
    const Affine t = getTransform(tex);
    float m[2][3]  = { { t.L.m[0][0], t.L.m[0][1], t.t[0] }, { t.L.m[1][0], t.L.m[1][1], t.t[1] } };
    bool identity  = true;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j) {
            char buf[48];
            std::snprintf(buf, sizeof buf, "%g", (double)m[i][j]);
            m[i][j] = std::strtof(buf, nullptr);
            identity = identity && std::fabs(m[i][j] - (i == j ? 1.0f : 0.0f)) <= 1e-5f;
        }
    if (identity)
        return;
    const auto row = [&](int i) {
        char buf[160];
        std::snprintf(buf, sizeof buf, "dot(vec3((%.9g), (%.9g), (%.9g)), vec3(uv.x, uv.y, 1.0))", (double)m[i][0], (double)m[i][1], (double)m[i][2]);
        return std::string(buf);
    };
    u = row(0), v = row(1);
}

// A colour property that names a brick texture (src/runtime/pattern/BrickPattern.cpp:13-38: color0 / color1 / scale_x,y / gap_x,y with
// the defaults below) as the shading expression make_brick_texture is (src/artic/texture/brick.art:1-19, identity transform):
//   suv = uv * scale;  x = fract(select(fract(suv.y * 0.5) > 0.5, suv.x + 0.5, suv.x));  y = fract(suv.y)
//   step(edge, v) = select(v < edge, 0, 1);  color_lerp(color0, color1, step(x, 1 - gap.x) * step(y, 1 - gap.y))
// The interpreter shared by the kernels and the oracle (include/ig_expr.h) evaluates it; the factor is 0 or 1, so the lerp returns one of
// the two colours exactly.
static bool brickExpression(const JsonValue& prop, const JsonValue& textures, std::string& expr, const std::string& owner)
{
    if (!prop.isString())
        return false;
    for (const auto& t : textures.arr) {
        if (t.getString("name") != prop.str)
            continue;
        const std::string type = t.getString("type");
        if (type == "noise" || type == "cellnoise" || type == "pnoise" || type == "perlin" || type == "voronoi" || type == "fbm") {
            // NoisePattern.cpp:33-57 -> make_[c]<type>_texture (src/artic/texture/noise.art:250-275): color * func(transform(uv) * scale, seed),
            // "colored": the three-channel form of the function
            const V3 c     = getColor(t, "color", V3(1, 1, 1), prop.str);
            const float sd = getConstNumber(t, "seed", 36326639.0f, prop.str);
            const float sx = getConstNumber(t, "scale_x", 10.0f, prop.str), sy = getConstNumber(t, "scale_y", 10.0f, prop.str);
            const JsonValue* colored = t.find("colored");
            std::string tu, tv;
            textureCoordinates(t, tu, tv);
            char buf[768];
            std::snprintf(buf, sizeof buf, "color((%.9g), (%.9g), (%.9g)) * %s%s(vec2(%s * (%.9g), %s * (%.9g)), (%.9g))", (double)c.x, (double)c.y, (double)c.z,
                          colored && colored->isBool() && colored->b ? "c" : "", type.c_str(), tu.c_str(), (double)sx, tv.c_str(), (double)sy, (double)sd);
            expr = buf;
            return true;
        }
        if (type != "brick")
            continue;
        const V3 c0 = getColor(t, "color0", V3(0, 0, 0), prop.str), c1 = getColor(t, "color1", V3(1, 1, 1), prop.str);
        const float sx = getConstNumber(t, "scale_x", 3.0f, prop.str), sy = getConstNumber(t, "scale_y", 6.0f, prop.str);
        const float gx = getConstNumber(t, "gap_x", 0.05f, prop.str), gy = getConstNumber(t, "gap_y", 0.1f, prop.str);
        const auto num = [](float v) { // (nine significant digits name a float exactly)
            char buf[48];
            std::snprintf(buf, sizeof buf, "(%.9g)", (double)v);
            return std::string(buf);
        };
        const auto col = [&](const V3& c) { return "color(" + num(c.x) + ", " + num(c.y) + ", " + num(c.z) + ")"; };
        std::string tu, tv;
        textureCoordinates(t, tu, tv);
        const std::string su = tu + " * " + num(sx), sv = tv + " * " + num(sy);
        expr = "mix(" + col(c0) + ", " + col(c1) + ", select((1 - " + num(gx) + ") < fract(select(fract(" + sv + " * 0.5) > 0.5, " + su + " + 0.5, " + su + ")), 0.0, 1.0)"
               + " * select((1 - " + num(gy) + ") < fract(" + sv + "), 0.0, 1.0))";
        return true;
    }
    return false;
}

// A colour property of a BSDF (ShadingTree::addColor, src/runtime/loader/ShadingTree.cpp): a constant, a bitmap, checkerboard or brick
// texture by name, or a PExpr string, which becomes a program of the expression table unless it folds to a constant.
static void lowerColor(const JsonValue& bsdf, const char* key, V3 def, const JsonValue& textures, TextureBank& bank, ig_material& m, const std::string& name)
{
    const JsonValue* col = bsdf.find(key);
    if (col && col->isString())
        for (const auto& t : textures.arr)
            if (t.getString("name") == col->str && (t.getString("type") == "image" || t.getString("type") == "bitmap")) {
                m.flags |= IG_MAT_IMAGE;
                m.tex_refl = bank.get(col->str, name);
                return;
            }
    std::string brick;
    bool is_brick = col && brickExpression(*col, textures, brick, name);
    bank.local_params.clear();
    if (col && !is_brick && col->isString())
        for (const auto& t : textures.arr)
            if (t.getString("name") == col->str && t.getString("type") == "expr") {
                // ExprPattern.cpp:14-75: the texture IS its "expr" string; properties num_<x> / color_<x> / vec_<x> / bool_<x> are variables <x> of
                // it (constants here; the reference allows textures and expressions there too)
                brick    = t.getString("expr");
                is_brick = true;
                if (brick.empty())
                    fail("Texture '" + col->str + "' requires an expression");
                for (const auto& kv : t.obj) {
                    const std::string& k = kv.first;
                    igh::pexpr::Param p{};
                    V3 c;
                    if (k.rfind("num_", 0) == 0 && k.size() > 4) {
                        const float v = getConstNumber(t, k, 0.0f, col->str);
                        p.type = igh::pexpr::Type::Num, p.value = { v, v, v, v };
                        bank.local_params[k.substr(4)] = p;
                    } else if ((k.rfind("color_", 0) == 0 && k.size() > 6) || (k.rfind("vec_", 0) == 0 && k.size() > 4)) {
                        const bool is_col = k[0] == 'c';
                        if (!parseConstColor(kv.second, c))
                            fail("Texture '" + col->str + "': property '" + k + "' is not a constant; textures and expressions there are not supported by the HIP backend");
                        p.type = is_col ? igh::pexpr::Type::Vec4 : igh::pexpr::Type::Vec3, p.value = { c.x, c.y, c.z, is_col ? 1.0f : 0.0f };
                        bank.local_params[k.substr(is_col ? 6 : 4)] = p;
                    } else if (k.rfind("bool_", 0) == 0 && k.size() > 5 && kv.second.isBool()) {
                        const float v = kv.second.b ? 1.0f : 0.0f;
                        p.type = igh::pexpr::Type::Bool, p.value = { v, v, v, v };
                        bank.local_params[k.substr(5)] = p;
                    }
                }
                break;
            }
    if (col && !is_brick && col->isString())
        for (const auto& t : textures.arr)
            if (t.getString("name") == col->str && t.getString("type") == "checkerboard" && t.has("transform")) {
                // a checkerboard under a transform: make_checkerboard_texture (texture/checkerboard.art:4-13) written out — color0 where the
                // parities of the two scaled coordinates differ, i.e. where node_checkerboard2 (the expression language's checkerboard(vec2)) is 0
                std::string tu, tv;
                textureCoordinates(t, tu, tv);
                if (tu == "uv.x" && tv == "uv.y")
                    break; // (an identity: the record form below)
                const V3 c0 = getColor(t, "color0", V3(0, 0, 0), col->str), c1 = getColor(t, "color1", V3(1, 1, 1), col->str);
                char buf[512];
                std::snprintf(buf, sizeof buf, "select(checkerboard(vec2(%s * (%.9g), %s * (%.9g))) == 1, color((%.9g), (%.9g), (%.9g)), color((%.9g), (%.9g), (%.9g)))", tu.c_str(),
                              (double)getConstNumber(t, "scale_x", 2.0f, col->str), tv.c_str(), (double)getConstNumber(t, "scale_y", 2.0f, col->str), (double)c1.x, (double)c1.y,
                              (double)c1.z, (double)c0.x, (double)c0.y, (double)c0.z);
                brick    = buf;
                is_brick = true;
                break;
            }
    if (col && !is_brick && lowerCheckerboard(*col, textures, m, name))
        return;
    V3 c = def;
    if (col && (is_brick || !parseConstColor(*col, c))) {
        if (!col->isString())
            fail("'" + name + "': property '" + key + "' is neither a colour nor an expression");
        const igh::pexpr::Program prog = bank.compileExpr(is_brick ? brick : col->str, name);
        using igh::pexpr::Type;
        if (prog.type == Type::Bool || prog.type == Type::Vec2) // "Expression does not return a number or color" (Transpiler.cpp:1303-1306)
            fail("'" + name + "': expression of property '" + key + "' is a " + igh::pexpr::typeName(prog.type) + ", not a number or colour");
        bank.local_params.clear();
        if (!prog.is_const) {
            m.flags |= IG_MAT_EXPR_COLOR;
            m.tex_refl = bank.addProgram(prog);
            return;
        }
        c = V3(prog.value[0], prog.value[1], prog.value[2]);
    }
    m.p[0] = c.x, m.p[1] = c.y, m.p[2] = c.z;
}

// the "parameters" section (docs/src/scene/pexpr.rst "Scene Parameters"): number / vector / color constants expressions may name
static void loadParameters(const JsonValue& doc, std::map<std::string, igh::pexpr::Param>& out)
{
    const JsonValue* params = doc.find("parameters");
    if (!params || !params->isArray())
        return;
    for (const auto& p : params->arr) {
        const std::string pname = p.getString("name"), type = p.getString("type");
        const JsonValue* v      = p.find("value");
        if (pname.empty() || !v)
            continue;
        // ShadingTree::handleGlobalParameterNumber / Vector / Color (ShadingTree.cpp:63-106): "int" and "integer" are numbers here,
        // a single number fills a vector or colour, a colour's alpha is 1
        igh::pexpr::Param out_p{};
        const bool triple = v->isArray() && v->arr.size() == 3 && v->arr[0].isNumber() && v->arr[1].isNumber() && v->arr[2].isNumber();
        if ((type == "number" || type == "int" || type == "integer") && v->isNumber()) {
            out_p.type  = igh::pexpr::Type::Num;
            out_p.value = { (float)v->num, (float)v->num, (float)v->num, (float)v->num };
        } else if ((type == "vector" || type == "color") && (triple || v->isNumber())) {
            out_p.type  = type == "vector" ? igh::pexpr::Type::Vec3 : igh::pexpr::Type::Vec4;
            const float x = triple ? (float)v->arr[0].num : (float)v->num, y = triple ? (float)v->arr[1].num : x, z = triple ? (float)v->arr[2].num : x;
            out_p.value = { x, y, z, type == "color" ? 1.0f : 0.0f };
        } else if (type == "string") {
            continue; // no expression can name it
        } else {
            fail("Parameter '" + pname + "': expected a number, or three numbers for a vector / color");
        }
        out[pname] = out_p;
    }
}

// The weight of a blend / mask (ShadingTree::addNumber): a constant, or a number expression that becomes a program of the expression
// table; cutoff: compared with the "cutoff" threshold into 0 / 1.
static void lowerWeight(const JsonValue& bsdf, TextureBank& bank, ig_material& m, const std::string& name, bool cutoff)
{
    const JsonValue* w = bsdf.find("weight");
    const float threshold = cutoff ? getConstNumber(bsdf, "cutoff", 0.5f, name) : 0.0f;
    if (!w || w->isNumber()) {
        const float weight = w ? (float)w->num : 0.5f;
        m.p[0]             = cutoff ? (weight < threshold ? 0.0f : 1.0f) : weight;
        return;
    }
    if (!w->isString())
        fail("'" + name + "': property 'weight' is neither a number nor an expression");
    std::string src = w->str;
    if (cutoff) { // select(weight < cutoff, 0, 1) (MaskBSDF.cpp:51-54)
        char num[48];
        std::snprintf(num, sizeof(num), "%.9g", (double)threshold);
        src = "select((" + src + ") < " + num + ", 0.0, 1.0)";
    }
    const igh::pexpr::Program prog = bank.compileExpr(src, name);
    if (!igh::pexpr::isScalar(prog.type))
        fail("'" + name + "': expression of property 'weight' is a " + igh::pexpr::typeName(prog.type) + ", not a number");
    if (prog.is_const) {
        m.p[0] = prog.value[0];
        return;
    }
    m.flags |= IG_MAT_EXPR_WEIGHT;
    m.tex_id = bank.addProgram(prog);
}

// Inner materials of blends: appended to the material table after the entity-bound ones (index = aux_base + position)
struct AuxMaterials {
    std::vector<ig_material> list;
    std::vector<std::string> names;
    int32_t base = 0;
};

// Number properties of a BSDF that are shading expressions (ShadingTree::addNumber / handlePropertyNumber / handleTexture,
// src/runtime/loader/ShadingTree.cpp:211-251,811-840): a string that does not fold to a constant is compiled like a colour
// expression — a colour-valued one counts as its average — and becomes an entry of the material's number list (ig_tables.h,
// IG_MAT_EXPR_NUMBERS), evaluated per hit. The record keeps the property's default.
struct NumberEntries {
    std::vector<uint32_t> words; // {kind | slot_a << 8 | slot_b << 16, bits(aspect), program offset} per entry
};
static float lowerNumber(const JsonValue& bsdf, const std::string& key, float def, TextureBank& bank, const std::string& owner, NumberEntries& out,
                         uint32_t kind, int slot_a, int slot_b = 0, float aspect = 1.0f)
{
    const JsonValue* v = bsdf.find(key);
    if (!v)
        return def;
    if (v->isNumber())
        return (float)v->num;
    if (!v->isString())
        fail("'" + owner + "': property '" + key + "' is neither a number nor an expression");
    V3 c;
    if ((ConstExpr::evaluate(v->str, c) || evaluateWithParameters(v->str, c)) && c.x == c.y && c.y == c.z)
        return c.x;
    using igh::pexpr::Type;
    igh::pexpr::Program prog = bank.compileExpr(v->str, owner);
    if (prog.type == Type::Bool || prog.type == Type::Vec2 || prog.type == Type::Str)
        fail("'" + owner + "': expression of property '" + key + "' is a " + igh::pexpr::typeName(prog.type) + ", not a number");
    if (!igh::pexpr::isScalar(prog.type)) // "Using average instead" (ShadingTree.cpp:835-836): color_average = (r + g + b) / 3
        prog = bank.compileExpr("avg((" + v->str + ").xyz)", owner);
    if (prog.is_const)
        return prog.value[0];
    uint32_t abits;
    std::memcpy(&abits, &aspect, 4);
    out.words.push_back(kind | (uint32_t)slot_a << 8 | (uint32_t)slot_b << 16);
    out.words.push_back(abits);
    out.words.push_back((uint32_t)bank.addProgram(prog));
    return def;
}
// roughness (or alpha) + a constant anisotropy -> (alpha_u, alpha_v) (BSDF::setupRoughness, BSDF.cpp:53-99; microfacet::compute_explicit);
// `dynamic`: the roughness is an expression, the pair is written per hit
static void lowerRoughnessPair(const JsonValue& bsdf, const std::string& rname, float def, TextureBank& bank, const std::string& owner, NumberEntries& out, uint32_t kind,
                               int slot_u, int slot_v, float& au, float& av, bool& dynamic)
{
    const float an     = getConstNumber(bsdf, "anisotropic", 0.0f, owner);
    const float aspect = an == 0 ? 1.0f : std::sqrt(1 - std::min(std::max(an, 0.0f), 1.0f) * 0.99f);
    const size_t before = out.words.size();
    const float r       = lowerNumber(bsdf, rname, def, bank, owner, out, kind, slot_u, slot_v, aspect);
    dynamic             = out.words.size() != before;
    au = r / aspect, av = r * aspect;
}

static ig_material lowerBsdf(const std::string& name, const JsonValue& scene_bsdfs, const JsonValue& textures, TextureBank& bank, AuxMaterials& aux, int depth = 0)
{
    NumberEntries numbers;
    const JsonValue* bsdf = nullptr;
    for (const auto& b : scene_bsdfs.arr)
        if (b.getString("name") == name)
            bsdf = &b;
    if (!bsdf)
        fail("Unknown bsdf '" + name + "'");
    if (depth > 4)
        fail("BSDF '" + name + "': nesting too deep");

    ig_material m;
    std::memset(&m, 0, sizeof(m));
    m.light_id = -1;
    m.tex_id   = -1;
    m.tex_refl = -1;

    const std::string type = bsdf->getString("type");
    if (type == "diffuse" || type == "roughdiffuse") {
        m.bsdf_type = IG_BSDF_DIFFUSE;
        lowerColor(*bsdf, "reflectance", V3(0.8f, 0.8f, 0.8f), textures, bank, m, name);
        m.p[3] = lowerNumber(*bsdf, bsdf->has("alpha") ? "alpha" : "roughness", 0.0f, bank, name, numbers, IG_NUM_PLAIN, 3);
    } else if (type == "dielectric" || type == "glass" || type == "roughdielectric" || type == "thindielectric") {
        // DielectricBSDF.cpp:13-41; IOR table BSDF.cpp:7-30 (vacuum 1.0, bk7 1.5046)
        if (bsdf->has("distribution") || bsdf->has("roughness_u") || bsdf->has("roughness_v") || bsdf->has("alpha_u") || bsdf->has("alpha_v"))
            fail("BSDF '" + name + "': only the default isotropic/anisotropic VNDF-GGX roughness form is supported");
        if (bsdf->has("ext_ior_material") || bsdf->has("int_ior_material"))
            fail("BSDF '" + name + "': named IOR materials are not supported by this loader");
        m.bsdf_type = IG_BSDF_DIELECTRIC;
        m.p[0]      = lowerNumber(*bsdf, "ext_ior", 1.0f, bank, name, numbers, IG_NUM_PLAIN, 0);
        m.p[1]      = lowerNumber(*bsdf, "int_ior", 1.5046f, bank, name, numbers, IG_NUM_PLAIN, 1);
        const V3 ks = getColor(*bsdf, "specular_reflectance", V3(1, 1, 1), name);
        const V3 kt = getColor(*bsdf, "specular_transmittance", V3(1, 1, 1), name);
        m.p[2] = ks.x, m.p[3] = ks.y, m.p[4] = ks.z;
        m.p[5] = kt.x, m.p[6] = kt.y, m.p[7] = kt.z;
        {
            // BSDF::setupRoughness (BSDF.cpp:53-99) -> make_dielectric_bsdf (dielectric.art:193-206): a rough interface unless the
            // distribution is a delta one (no roughness given, or alpha <= 1e-4); rough + thin is rough (dielectric.art:202)
            const std::string rname = bsdf->has("alpha") ? "alpha" : "roughness";
            float au, av;
            bool dynamic;
            lowerRoughnessPair(*bsdf, rname, 0.1f, bank, name, numbers, IG_NUM_ROUGHNESS_DIELECTRIC, 9, 10, au, av, dynamic);
            if (dynamic && bsdf->getBool("thin", false))
                m.flags |= IG_MAT_THIN; // (what the interface is where the expression yields a delta distribution)
            if (bsdf->has(rname) && au > 1e-4f && av > 1e-4f) {
                m.bsdf_type       = IG_BSDF_ROUGH_DIELECTRIC;
                const float alpha = au < av ? au : av;
                m.p[8]            = alpha <= 0.01f ? 1e-3f : (alpha <= 0.1f ? 1e-4f : 1e-5f); // pdf_eps (dielectric.art:67-82)
                m.p[9]            = au;
                m.p[10]           = av;
            } else if (bsdf->getBool("thin", false)) {
                m.flags |= IG_MAT_THIN;
            }
        }
    } else if (type == "transparent" || type == "passthrough") {
        // TransparentBSDF.cpp:13-22 (tinted), PassthroughBSDF.cpp (white): the ray goes straight on
        m.bsdf_type = IG_BSDF_TRANSPARENT;
        const V3 c  = type == "transparent" ? getColor(*bsdf, "color", V3(1, 1, 1), name) : V3(1, 1, 1);
        m.p[0] = c.x, m.p[1] = c.y, m.p[2] = c.z;
    } else if (type == "conductor" || type == "roughconductor" || type == "mirror") {
        // ConductorBSDF.cpp:13-34 (defaults: material "none" = eta 0, k 1, BSDF.cpp:41), roughness via
        // BSDF::setupRoughness (BSDF.cpp:53-99): VNDF-GGX, compute_explicit(roughness, anisotropic)
        // (src/artic/core/microfacet.art:427-432)
        if (bsdf->has("material"))
            fail("BSDF '" + name + "': named conductor materials are not supported by this loader");
        if (bsdf->has("distribution") || bsdf->has("roughness_u") || bsdf->has("roughness_v") || bsdf->has("alpha_u") || bsdf->has("alpha_v"))
            fail("BSDF '" + name + "': only the default isotropic/anisotropic VNDF-GGX roughness form is supported");
        const std::string rname = bsdf->has("alpha") ? "alpha" : "roughness";
        const bool smooth       = !bsdf->has(rname); // BSDF::setupRoughness (BSDF.cpp:58-63): no roughness -> delta distribution
        m.bsdf_type    = IG_BSDF_CONDUCTOR;
        const V3 eta   = getColor(*bsdf, "eta", V3(0, 0, 0), name);
        const V3 k     = getColor(*bsdf, "k", V3(1, 1, 1), name);
        const V3 ks    = getColor(*bsdf, "specular_reflectance", V3(1, 1, 1), name);
        bool dynamic;
        lowerRoughnessPair(*bsdf, rname, 0.1f, bank, name, numbers, IG_NUM_ROUGHNESS_DELTA, 9, 10, m.p[9], m.p[10], dynamic);
        m.p[0] = eta.x, m.p[1] = eta.y, m.p[2] = eta.z;
        m.p[3] = k.x, m.p[4] = k.y, m.p[5] = k.z;
        m.p[6] = ks.x, m.p[7] = ks.y, m.p[8] = ks.z;
        if (smooth || m.p[9] <= 1e-4f || m.p[10] <= 1e-4f) // check_if_delta_distribution (microfacet.art:298)
            m.flags |= IG_MAT_SMOOTH;
    } else if (type == "plastic" || type == "roughplastic") {
        // PlasticBSDF.cpp:13-44: a diffuse base under a (rough) mirror-like coating, mixed by the dielectric Fresnel term
        if (bsdf->has("ext_ior_material") || bsdf->has("int_ior_material"))
            fail("BSDF '" + name + "': named IOR materials are not supported by this loader");
        if (bsdf->has("distribution") || bsdf->has("roughness_u") || bsdf->has("roughness_v") || bsdf->has("alpha_u") || bsdf->has("alpha_v"))
            fail("BSDF '" + name + "': only the default isotropic/anisotropic VNDF-GGX roughness form is supported");
        m.bsdf_type = IG_BSDF_PLASTIC;
        lowerColor(*bsdf, "diffuse_reflectance", V3(0.8f, 0.8f, 0.8f), textures, bank, m, name);
        m.p[3]      = lowerNumber(*bsdf, "ext_ior", 1.0f, bank, name, numbers, IG_NUM_PLAIN, 3);  // vacuum (BSDF.cpp:8)
        m.p[4]      = lowerNumber(*bsdf, "int_ior", 1.49f, bank, name, numbers, IG_NUM_PLAIN, 4); // polypropylene (BSDF.cpp:17)
        const V3 ks = getColor(*bsdf, "specular_reflectance", V3(1, 1, 1), name);
        m.p[6] = ks.x, m.p[7] = ks.y, m.p[8] = ks.z;
        // BSDF::setupRoughness (BSDF.cpp:53-99), as for the conductor
        const std::string rname = bsdf->has("alpha") ? "alpha" : "roughness";
        bool dynamic;
        lowerRoughnessPair(*bsdf, rname, 0.1f, bank, name, numbers, IG_NUM_ROUGHNESS_DELTA, 9, 10, m.p[9], m.p[10], dynamic);
        if (!bsdf->has(rname) || m.p[9] <= 1e-4f || m.p[10] <= 1e-4f)
            m.flags |= IG_MAT_SMOOTH;
    } else if (type == "principled") {
        // PrincipledBSDF.cpp:14-98: every number may be an expression (the number list)
        for (const char* key : { "reflective_ior_spec", "refractive_ior_spec", "ior_spec" })
            if (bsdf->has(key))
                fail("BSDF '" + name + "': named IOR materials are not supported by this loader");
        m.bsdf_type = IG_BSDF_PRINCIPLED;
        lowerColor(*bsdf, "base_color", V3(0.8f, 0.8f, 0.8f), textures, bank, m, name);
        const float bk7 = 1.5046f; // BSDF.cpp:9
        // every number may be an expression or a texture (PrincipledBSDF.cpp:22-54: tree.addNumber)
        if (bsdf->has("reflective_ior") || bsdf->has("refractive_ior")) {
            m.p[3] = lowerNumber(*bsdf, "reflective_ior", bk7, bank, name, numbers, IG_NUM_PLAIN, 3);
            m.p[4] = lowerNumber(*bsdf, "refractive_ior", bk7, bank, name, numbers, IG_NUM_PLAIN, 4);
        } else {
            m.p[3] = lowerNumber(*bsdf, "ior", bk7, bank, name, numbers, IG_NUM_PLAIN, 3);
            m.p[4] = lowerNumber(*bsdf, "ior", bk7, bank, name, numbers, IG_NUM_PLAIN, 4);
        }
        m.p[5] = lowerNumber(*bsdf, "diffuse_transmission", 0.0f, bank, name, numbers, IG_NUM_PLAIN, 5);
        m.p[6] = lowerNumber(*bsdf, "specular_transmission", 0.0f, bank, name, numbers, IG_NUM_PLAIN, 6);
        m.p[7] = lowerNumber(*bsdf, "specular_tint", 0.0f, bank, name, numbers, IG_NUM_PLAIN, 7);
        if (bsdf->has("roughness_u") || bsdf->has("roughness_v")) {
            m.p[8] = lowerNumber(*bsdf, "roughness_u", 0.5f, bank, name, numbers, IG_NUM_PLAIN, 8);
            m.p[9] = lowerNumber(*bsdf, "roughness_v", 0.5f, bank, name, numbers, IG_NUM_PLAIN, 9);
        } else {
            // microfacet::compute_explicit (src/artic/core/microfacet.art:427-432)
            bool dynamic;
            lowerRoughnessPair(*bsdf, "roughness", 0.5f, bank, name, numbers, IG_NUM_ROUGHNESS, 8, 9, m.p[8], m.p[9], dynamic);
        }
        m.p[10] = lowerNumber(*bsdf, "flatness", 0.0f, bank, name, numbers, IG_NUM_PLAIN, 10);
        m.r[0]  = lowerNumber(*bsdf, "metallic", 0.0f, bank, name, numbers, IG_NUM_PLAIN, 20);
        m.r[1]  = lowerNumber(*bsdf, "sheen", 0.0f, bank, name, numbers, IG_NUM_PLAIN, 21);
        m.r[2]  = lowerNumber(*bsdf, "sheen_tint", 0.0f, bank, name, numbers, IG_NUM_PLAIN, 22);
        m.r[3]  = lowerNumber(*bsdf, "clearcoat", 0.0f, bank, name, numbers, IG_NUM_PLAIN, 23);
        m.r[4]  = lowerNumber(*bsdf, "clearcoat_gloss", 0.0f, bank, name, numbers, IG_NUM_PLAIN, 24);
        m.r[5]  = lowerNumber(*bsdf, "clearcoat_roughness", 0.1f, bank, name, numbers, IG_NUM_PLAIN, 25);
        if (bsdf->getBool("thin", false))
            m.flags |= IG_MAT_THIN;
        if (!bsdf->getBool("clearcoat_top_only", true))
            m.flags |= IG_MAT_CLEARCOAT_ALL;
    } else if (type == "blend" || type == "mix") {
        // BlendBSDF.cpp:14-56: make_mix_bsdf(first, second, weight); the same bsdf twice is that bsdf
        const std::string first = bsdf->getString("first"), second = bsdf->getString("second");
        if (first.empty() || second.empty())
            fail("BSDF '" + name + "': has no inner bsdfs given");
        if (first == second)
            return lowerBsdf(first, scene_bsdfs, textures, bank, aux, depth + 1);
        m.bsdf_type = IG_BSDF_BLEND;
        lowerWeight(*bsdf, bank, m, name, false);
        int slot    = 0;
        for (const std::string& inner : { first, second }) {
            const ig_material im = lowerBsdf(inner, scene_bsdfs, textures, bank, aux, depth + 1);
            if (im.bsdf_type == IG_BSDF_BLEND || (im.flags & (IG_MAT_BUMP | IG_MAT_NORMALMAP | IG_MAT_EXPR_NORMAL)))
                fail("BSDF '" + name + "': nested blends and bump / normal maps inside a blend are not supported by the HIP backend");
            if ((im.flags & (IG_MAT_EXPR_COLOR | IG_MAT_EXPR_NUMBERS)) || im.bsdf_type == IG_BSDF_RAD_BRTD || im.bsdf_type == IG_BSDF_RAD_ROOS)
                fail("BSDF '" + name + "': expressions and Radiance BSDFs inside a blend are not supported by the HIP backend");
            m.pad[slot++] = aux.base + (int32_t)aux.list.size();
            aux.list.push_back(im);
            aux.names.push_back(inner);
        }
    } else if (type == "twosided" || type == "doublesided") {
        // DoubleSidedBSDF.cpp:16-35: make_doublesided_bsdf around the inner BSDF (the outermost wrapper only)
        const std::string inner = bsdf->getString("bsdf");
        if (inner.empty())
            fail("BSDF '" + name + "': has no inner bsdf given");
        if (depth > 0)
            fail("BSDF '" + name + "': a two-sided BSDF inside another BSDF is not supported by the HIP backend");
        m = lowerBsdf(inner, scene_bsdfs, textures, bank, aux, depth + 1);
        m.flags |= IG_MAT_DOUBLESIDED;
    } else if (type == "mask" || type == "cutoff") {
        // MaskBSDF.cpp:17-57: make_mix_bsdf(masked, passthrough, weight) ("inverted": the other way round); "cutoff" turns the
        // weight into 0 / 1 against a threshold. Constant weights only (the reference's scenes drive it with noise expressions).
        const std::string masked = bsdf->getString("bsdf");
        if (masked.empty())
            fail("BSDF '" + name + "': has no inner bsdf given");
        const bool inverted = bsdf->getBool("inverted", false);
        ig_material inner   = lowerBsdf(masked, scene_bsdfs, textures, bank, aux, depth + 1);
        if (inner.bsdf_type == IG_BSDF_BLEND || (inner.flags & (IG_MAT_BUMP | IG_MAT_NORMALMAP | IG_MAT_EXPR_NORMAL)))
            fail("BSDF '" + name + "': blends and bump / normal maps inside a mask are not supported by the HIP backend");
        if ((inner.flags & (IG_MAT_EXPR_COLOR | IG_MAT_EXPR_NUMBERS)) || inner.bsdf_type == IG_BSDF_RAD_BRTD || inner.bsdf_type == IG_BSDF_RAD_ROOS)
            fail("BSDF '" + name + "': expressions and Radiance BSDFs inside a mask are not supported by the HIP backend");
        ig_material through{};
        through.bsdf_type = IG_BSDF_TRANSPARENT; // make_passthrough_bsdf = white perfect refraction
        through.light_id  = -1;
        through.tex_id = through.tex_refl = -1;
        through.p[0] = through.p[1] = through.p[2] = 1;
        m.bsdf_type = IG_BSDF_BLEND;
        lowerWeight(*bsdf, bank, m, name, type == "cutoff"); // "cutoff" turns the weight into 0 / 1 against a threshold
        for (int slot = 0; slot < 2; ++slot) {
            const bool is_inner = (slot == 0) != inverted;
            m.pad[slot]         = aux.base + (int32_t)aux.list.size();
            aux.list.push_back(is_inner ? inner : through);
            aux.names.push_back(is_inner ? masked : "passthrough");
        }
    } else if (type == "phong") {
        // PhongBSDF.cpp:16-17
        m.bsdf_type = IG_BSDF_PHONG;
        const V3 ks = getColor(*bsdf, "specular_reflectance", V3(1, 1, 1), name);
        m.p[0] = ks.x, m.p[1] = ks.y, m.p[2] = ks.z;
        m.p[3] = lowerNumber(*bsdf, "exponent", 30.0f, bank, name, numbers, IG_NUM_PLAIN, 3);
    } else if (type == "bumpmap" || type == "normalmap") {
        // MapBSDF.cpp:17-52: make_bumpmap(ctx, inner, texture_dx(map).r, texture_dy(map).r, strength) /
        // make_normalmap(ctx, inner, map colour, strength)
        const std::string inner = bsdf->getString("bsdf");
        if (inner.empty())
            fail("BSDF '" + name + "': has no inner bsdf given");
        m = lowerBsdf(inner, scene_bsdfs, textures, bank, aux, depth + 1);
        if (m.flags & (IG_MAT_BUMP | IG_MAT_NORMALMAP | IG_MAT_EXPR_NORMAL | IG_MAT_EXPR_WEIGHT))
            fail("BSDF '" + name + "': nested bump / normal maps and maps around a blend with an expression weight are not supported by the HIP backend");
        const JsonValue* map = bsdf->find("map");
        if (!map || !map->isString())
            fail("BSDF '" + name + "': 'map' must name a bitmap texture");
        m.flags |= type == "bumpmap" ? IG_MAT_BUMP : IG_MAT_NORMALMAP;
        m.tex_id = bank.get(map->str, name);
        m.p[11]  = getConstNumber(*bsdf, "strength", 1.0f, name);
    } else if (type == "rad_brtdfunc") {
        // RadBRTDFuncBSDF.cpp:15-20: six colours, constants here
        m.bsdf_type    = IG_BSDF_RAD_BRTD;
        const V3 rs    = getColor(*bsdf, "reflection_specular", V3(1, 1, 1), name);
        const V3 ts    = getColor(*bsdf, "transmission_specular", V3(0, 0, 0), name);
        const V3 dd    = getColor(*bsdf, "direct_diffuse", V3(0, 0, 0), name);
        const V3 rf    = getColor(*bsdf, "reflection_front_diffuse", V3(0, 0, 0), name);
        const V3 rb    = getColor(*bsdf, "reflection_back_diffuse", V3(0, 0, 0), name);
        const V3 td    = getColor(*bsdf, "transmission_diffuse", V3(0, 0, 0), name);
        const V3 rfd = rf + dd, rbd = rb + dd; // color_add(select(is_entering, front, back), dir_diff) (bsdf/rad.art:13)
        const float v[15] = { rs.x, rs.y, rs.z, ts.x, ts.y, ts.z, rfd.x, rfd.y, rfd.z, rbd.x, rbd.y, rbd.z, td.x, td.y, td.z };
        std::memcpy(m.p, v, 12 * sizeof(float));
        std::memcpy(m.q, v + 12, 3 * sizeof(float));
    } else if (type == "rad_roos") {
        // RadRoosBSDF.cpp:15-38. The C++ side hands (refl_w, refl_p, refl_q, trns_w, trns_p, trns_q) to make_rad_roos_bsdf(surf, cosN,
        // trns_w, trns_p, trns_q, refl_w, refl_p, refl_q, ...) (bsdf/rad.art:36-39): the "refl_*" properties drive the transmission
        // and vice versa. Restated as written: p[0..2] is what the Artic function calls trns_*, p[3..5] what it calls refl_*.
        m.bsdf_type = IG_BSDF_RAD_ROOS;
        m.p[0] = getConstNumber(*bsdf, "refl_w", 0.0f, name), m.p[1] = getConstNumber(*bsdf, "refl_p", 0.0f, name), m.p[2] = getConstNumber(*bsdf, "refl_q", 0.0f, name);
        m.p[3] = getConstNumber(*bsdf, "trns_w", 0.0f, name), m.p[4] = getConstNumber(*bsdf, "trns_p", 0.0f, name), m.p[5] = getConstNumber(*bsdf, "trns_q", 0.0f, name);
        const V3 rf = getColor(*bsdf, "reflection_front_diffuse", V3(0, 0, 0), name);
        const V3 rb = getColor(*bsdf, "reflection_back_diffuse", V3(0, 0, 0), name);
        const V3 td = getColor(*bsdf, "transmission_diffuse", V3(0, 0, 0), name);
        m.p[6] = rf.x, m.p[7] = rf.y, m.p[8] = rf.z, m.p[9] = rb.x, m.p[10] = rb.y, m.p[11] = rb.z;
        m.q[0] = td.x, m.q[1] = td.y, m.q[2] = td.z;
    } else if (type == "transform") {
        // TransformBSDF.cpp:17-49: make_normal_set(ctx, inner, normal) with the "normal" vector property (default +Z, as written);
        // the "tangent" form (make_normal_tangent_set) is not carried
        const std::string inner = bsdf->getString("bsdf");
        if (inner.empty())
            fail("BSDF '" + name + "': has no inner bsdf given");
        if (bsdf->has("tangent"))
            fail("BSDF '" + name + "': the 'tangent' form of the transform BSDF is not supported by the HIP backend");
        m = lowerBsdf(inner, scene_bsdfs, textures, bank, aux, depth + 1);
        if (m.bsdf_type == IG_BSDF_BLEND || (m.flags & (IG_MAT_BUMP | IG_MAT_NORMALMAP | IG_MAT_EXPR_NORMAL | IG_MAT_DOUBLESIDED)))
            fail("BSDF '" + name + "': blends, two-sided BSDFs and nested normal changes inside a transform BSDF are not supported by the HIP backend");
        std::string src = "vec3(0, 0, 1)";
        if (const JsonValue* nv = bsdf->find("normal")) {
            if (nv->isString())
                src = nv->str;
            else if (nv->isArray() && nv->arr.size() == 3 && nv->arr[0].isNumber() && nv->arr[1].isNumber() && nv->arr[2].isNumber())
            {
                char buf[160];
                std::snprintf(buf, sizeof(buf), "vec3(%.9g, %.9g, %.9g)", (double)(float)nv->arr[0].num, (double)(float)nv->arr[1].num, (double)(float)nv->arr[2].num);
                src = buf;
            }
            else
                fail("BSDF '" + name + "': 'normal' is neither a vector nor an expression");
        }
        const igh::pexpr::Program prog = bank.compileExpr(src, name);
        if (prog.type != igh::pexpr::Type::Vec3)
            fail("BSDF '" + name + "': expression of property 'normal' is a " + igh::pexpr::typeName(prog.type) + ", not a vec3");
        m.flags |= IG_MAT_EXPR_NORMAL;
        m.tex_id = bank.addProgram(prog);
    } else {
        fail("BSDF '" + name + "': type '" + type + "' is not supported by the HIP backend");
    }
    if (!numbers.words.empty()) {
        // the material's number list: count, then the entries; its offset travels as the bits of r[7]
        const uint32_t at = (uint32_t)bank.expr_code->size();
        bank.expr_code->push_back((uint32_t)(numbers.words.size() / 3));
        bank.expr_code->insert(bank.expr_code->end(), numbers.words.begin(), numbers.words.end());
        m.flags |= IG_MAT_EXPR_NUMBERS;
        std::memcpy(&m.r[7], &at, 4);
    }
    return m;
}

// ---------------------------------------------------------------- light hierarchy
// LightHierarchy::setup (src/runtime/light/LightHierarchy.cpp:47-125) over PointBvh
// (src/runtime/container/PointBvh.inl:23-96), restated including its quirks (the stored split value
// `mid` is half the node's largest extent, not a coordinate).
struct LightEntry {
    V3 position, direction;
    float flux; // negative: no direction (delta light)
    int32_t id;
};

struct PointBvhNode {
    size_t index;
    BBox bbox;
    float mid;
    int axis; // < 0: leaf
};

static void buildLightHierarchy(const std::vector<LightEntry>& lights, std::vector<float>& out_nodes, std::vector<uint32_t>& out_codes)
{
    std::vector<PointBvhNode> inner;
    std::vector<LightEntry> leaves;
    auto makeLeaf = [](const BBox& b) { return PointBvhNode{ 0, b, 0, -1 }; };

    for (const LightEntry& elem : lights) {
        const V3 p = elem.position;
        leaves.push_back(elem);
        if (inner.empty()) {
            BBox b;
            b.extend(p);
            inner.push_back(makeLeaf(b));
            continue;
        }
        // getForPointExtend: extend boxes along the path, descend by the box centre
        size_t cur = 0;
        for (;;) {
            inner[cur].bbox.extend(p);
            if (inner[cur].axis < 0)
                break;
            const float mid = inner[cur].bbox.center()[inner[cur].axis];
            cur             = p[inner[cur].axis] < mid ? inner[cur].index : inner[cur].index + 1;
        }
        const size_t leafIdx    = leaves.size() - 1;
        const size_t oldLeafIdx = inner[cur].index;
        const BBox nodeBox      = inner[cur].bbox;
        const V3 diam           = nodeBox.diameter();
        int axis                = 0;
        float maxc              = diam.x;
        if (diam.y > maxc) { maxc = diam.y; axis = 1; }
        if (diam.z > maxc) { maxc = diam.z; axis = 2; }
        const float mid      = maxc / 2;
        const size_t leftIdx = inner.size();
        inner[cur].index = leftIdx;
        inner[cur].axis  = axis;
        inner[cur].mid   = mid;
        // computeSplit(left, right, axis, 0.5) (math/BoundingBox.h:74-82)
        BBox leftBox = nodeBox, rightBox = nodeBox;
        const float off = (nodeBox.max[axis] - nodeBox.min[axis]) * 0.5f;
        leftBox.max[axis] -= off;
        rightBox.min[axis] += off;
        inner.push_back(makeLeaf(leftBox));
        inner.push_back(makeLeaf(rightBox));
        if (p[axis] < mid) {
            inner[leftIdx].index     = leafIdx;
            inner[leftIdx + 1].index = oldLeafIdx;
        } else {
            inner[leftIdx].index     = oldLeafIdx;
            inner[leftIdx + 1].index = leafIdx;
        }
    }

    std::vector<LightEntry> entries(inner.size());
    out_codes.assign(lights.size(), 0);
    std::function<LightEntry(size_t, uint32_t, uint32_t)> populate = [&](size_t id, uint32_t code, uint32_t depth) -> LightEntry {
        const PointBvhNode& node = inner.at(id);
        LightEntry& entry        = entries[id];
        if (node.axis < 0) {
            entry                        = leaves.at(node.index);
            out_codes[(size_t)entry.id] = code;
        } else {
            const LightEntry left  = populate(node.index, code, depth + 1);
            const LightEntry right = populate(node.index + 1, code | (0x1u << depth), depth + 1);
            entry.position         = node.bbox.center();
            entry.id               = -(int32_t)(node.index + 1);
            if (left.flux < 0 && right.flux < 0) {
                entry.direction = V3(0, 0, 1);
                entry.flux      = left.flux + right.flux;
            } else if (left.flux < 0) {
                entry.direction = V3(0, 0, 1);
                entry.flux      = -(-left.flux + right.flux);
            } else if (right.flux < 0) {
                entry.direction = V3(0, 0, 1);
                entry.flux      = -(left.flux - right.flux);
            } else {
                entry.direction = normalized(left.direction + right.direction);
                entry.flux      = left.flux + right.flux;
            }
        }
        return entry;
    };
    populate(0, 0, 0);

    out_nodes.clear();
    for (const LightEntry& e : entries) {
        float idf;
        std::memcpy(&idf, &e.id, 4);
        const float rec[8] = { e.position.x, e.position.y, e.position.z, e.flux, e.direction.x, e.direction.y, e.direction.z, idf };
        out_nodes.insert(out_nodes.end(), rec, rec + 8);
    }
}

static void writeEntity(std::vector<float>& tbl, const Affine& toLocal, const Affine& toGlobal, const M3& normalMat, uint32_t shapeID, uint32_t materialID)
{
    // Column-major 3x4, 3x4, 3x3, then ids (LoaderEntity.cpp:150-162)
    auto write34 = [&](const Affine& a) {
        for (int c = 0; c < 3; ++c)
            for (int r = 0; r < 3; ++r)
                tbl.push_back(a.L.m[r][c]);
        tbl.push_back(a.t.x);
        tbl.push_back(a.t.y);
        tbl.push_back(a.t.z);
    };
    write34(toLocal);
    write34(toGlobal);
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r)
            tbl.push_back(normalMat.m[r][c]);
    float f;
    std::memcpy(&f, &shapeID, 4);
    tbl.push_back(f);
    std::memcpy(&f, &materialID, 4);
    tbl.push_back(f);
    tbl.push_back(0.0f);
}

static std::unique_ptr<Scene> buildScene(const JsonValue& doc, const std::string& base_dir, const igh_options* opts)
{
    if (!doc.isObject())
        fail("Expected scene root to be an object");

    auto sc = std::make_unique<Scene>();

    std::map<std::string, igh::pexpr::Param> parameters;
    loadParameters(doc, parameters);
    struct ParamScope {
        ParamScope(const std::map<std::string, igh::pexpr::Param>* p) { g_scene_params = p; }
        ~ParamScope() { g_scene_params = nullptr; }
    } param_scope(&parameters);

    // ---- technique (Runtime.cpp:20-36, PathTechnique.cpp:8-18)
    ig_technique tech{ 64, 2, 0.0f, 1, IG_SELECTOR_UNIFORM, IG_TECHNIQUE_PATH, 0, 0, 0, 0, 0.0f, 0 };
    std::string selector;
    if (const JsonValue* t = doc.find("technique")) {
        const std::string type = t->getString("type", "path");
        if (type == "ao")
            tech.type = IG_TECHNIQUE_AO; // AOTechnique.cpp: no parameters
        else if (type == "volpath")
            tech.type = IG_TECHNIQUE_VOLPATH; // VolumePathTechnique.cpp: the parameters of the path tracer
        else if (type == "debug") {
            // DebugTechnique.cpp:6-14, DebugMode.cpp:5-34: the mode by name (case-insensitive), "normal" when unknown
            tech.type = IG_TECHNIQUE_DEBUG;
            static const char* const modes[] = { "normal", "tangent", "bitangent", "geometric normal", "local normal", "local tangent", "local bitangent",
                                                 "local geometric normal", "texture coords", "prim coords", "point", "local point", "generated coords",
                                                 "hit distance", "area", "raw prim id", "prim id", "raw entity id", "entity id", "raw material id",
                                                 "material id", "is emissive", "is specular", "is entering", "check bsdf", "albedo", "medium inner",
                                                 "medium outer" };
            std::string mode = t->getString("mode", "");
            for (char& ch : mode)
                ch = (char)std::tolower((unsigned char)ch);
            for (int i = 0; i < 28; ++i)
                if (mode == modes[i])
                    tech.debug_mode = i;
        } else if (type == "lt" || type == "lighttracer")
            tech.type = IG_TECHNIQUE_LIGHTTRACER; // LightTracerTechnique.cpp:10-17
        else if (type == "wireframe")
            tech.type = IG_TECHNIQUE_WIREFRAME; // WireframeTechnique.cpp: no parameters
        else if (type == "ppm" || type == "photonmapper") {
            // PhotonMappingTechnique.cpp:11-22 (the merge radius becomes absolute once the scene's bounding box is known, below)
            tech.type            = IG_TECHNIQUE_PPM;
            tech.photon_count    = std::max(100, t->getInt("photons", 1000000));
            tech.max_light_depth = t->getInt("max_light_depth", 8);
            tech.merge_radius    = t->getNumber("radius", 0.01f);
        } else if (type != "path")
            fail("Technique '" + type + "' is not supported by the HIP backend (only 'path', 'volpath', 'ao', 'debug', 'lt', 'ppm' and 'wireframe')");
        tech.max_depth = t->getInt("max_depth", 64);
        tech.min_depth = t->getInt("min_depth", 2);
        if (tech.type == IG_TECHNIQUE_PPM) { // "max_depth" else "max_camera_depth", "min_depth" else "min_camera_depth"
            tech.max_depth = t->has("max_depth") ? tech.max_depth : t->getInt("max_camera_depth", 64);
            tech.min_depth = t->has("min_depth") ? tech.min_depth : t->getInt("min_camera_depth", 2);
        }
        if (tech.type == IG_TECHNIQUE_LIGHTTRACER) { // "max_depth" else "max_light_depth", "min_depth" else "min_light_depth"
            tech.max_depth = t->has("max_depth") ? tech.max_depth : t->getInt("max_light_depth", 64);
            tech.min_depth = t->has("min_depth") ? tech.min_depth : t->getInt("min_light_depth", 2);
        }
        tech.clamp     = t->getNumber("clamp", 0.0f);
        tech.nee       = t->getBool("nee", true) ? 1 : 0;
        selector       = t->getString("light_selector");
        tech.aov_mis = (tech.type == IG_TECHNIQUE_PATH && t->getBool("aov_mis", false)) ? 1 : 0; // PathTechnique.cpp:16
    }

    // ---- film (Runtime.cpp:38-54)
    float film_w = 800, film_h = 600;
    if (const JsonValue* film = doc.find("film")) {
        if (const JsonValue* size = film->find("size")) {
            if (!size->isArray() || size->arr.size() != 2 || !size->arr[0].isNumber() || !size->arr[1].isNumber())
                fail("Expected film size to be a vector of length 2");
            film_w = (float)size->arr[0].num;
            film_h = (float)size->arr[1].num;
        }
    }
    int width  = (opts && opts->film_width > 0) ? opts->film_width : (int)film_w;
    int height = (opts && opts->film_height > 0) ? opts->film_height : (int)film_h;
    width      = std::max(1, width);
    height     = std::max(1, height);

    // ---- camera (PerspectiveCamera.cpp:7-24,69-76; Camera.cpp:5-15)
    ig_camera cam;
    std::memset(&cam, 0, sizeof(cam));
    cam.eye[2] = 0;
    cam.dir[2] = 1;
    cam.up[1]  = 1;
    cam.fov    = 60 * Deg2Rad;
    cam.near_clip    = 0;
    cam.far_clip     = std::numeric_limits<float>::max();
    cam.aspect_ratio = -1;
    cam.scale        = 1;
    cam.focal_length = 1;
    if (const JsonValue* film = doc.find("film")) {
        // Runtime.cpp:51-53 + RayGenerationShader.cpp:40-48: "halton", "mjitt", everything else is the uniform sampler
        std::string ps = film->getString("sampler", "independent");
        for (char& ch : ps)
            ch = (char)std::tolower((unsigned char)ch);
        cam.pixel_sampler = ps == "halton" ? IG_PIXEL_SAMPLER_HALTON : (ps == "mjitt" ? IG_PIXEL_SAMPLER_MJITT : IG_PIXEL_SAMPLER_UNIFORM);
    }
    bool cam_has_transform = false;
    if (const JsonValue* c = doc.find("camera")) {
        // LoaderCamera.cpp:31-36
        const std::string type = c->getString("type", "perspective");
        if (type == "perspective")
            cam.type = IG_CAMERA_PERSPECTIVE;
        else if (type == "orthogonal")
            cam.type = IG_CAMERA_ORTHOGONAL;
        else if (type == "fishlens" || type == "fisheye")
            cam.type = IG_CAMERA_FISHLENS;
        else
            fail("Camera '" + type + "' is not supported by the HIP backend (perspective, orthogonal, fishlens)");
        if (cam.type == IG_CAMERA_PERSPECTIVE) {
            // PerspectiveCamera.cpp:19-20,50: a lens when the aperture radius exceeds FltEps
            cam.focal_length    = c->getNumber("focal_length", 1.0f);
            cam.aperture_radius = c->getNumber("aperture_radius", 0.0f);
        } else if (cam.type == IG_CAMERA_ORTHOGONAL) {
            cam.scale = c->getNumber("scale", 1.0f); // OrthogonalCamera.cpp:16
        } else {
            // FishLensCamera.cpp:17-28
            cam.fisheye_mask       = c->getBool("mask", false) ? 1 : 0;
            std::string mode       = c->getString("mode", "circular");
            for (char& ch : mode)
                ch = (char)std::tolower((unsigned char)ch);
            cam.fisheye_mode = mode == "cropped" ? IG_FISHEYE_CROPPED : (mode == "full" ? IG_FISHEYE_FULL : IG_FISHEYE_CIRCULAR);
        }
        // the orthogonal camera always takes its transform, identity when absent (OrthogonalCamera.cpp:10,54-62)
        cam_has_transform = c->has("transform") || cam.type == IG_CAMERA_ORTHOGONAL;
        if (c->has("transform")) {
            const Affine T = getTransform(*c);
            const V3 eye   = T.point(V3(0, 0, 0));
            for (int i = 0; i < 3; ++i) {
                cam.eye[i] = eye[i];
                cam.dir[i] = T.L.m[i][2];
                cam.up[i]  = T.L.m[i][1];
            }
        }
        if (c->has("vfov")) {
            cam.fov             = c->getNumber("vfov", 60) * Deg2Rad;
            cam.fov_is_vertical = 1;
        } else if (c->has("hfov")) {
            cam.fov = c->getNumber("hfov", 60) * Deg2Rad;
        } else {
            cam.fov = c->getNumber("fov", 60) * Deg2Rad;
        }
        cam.near_clip = c->getNumber("near_clip", 0.0f);
        cam.far_clip  = c->getNumber("far_clip", std::numeric_limits<float>::max());
        if (cam.far_clip < cam.near_clip)
            std::swap(cam.near_clip, cam.far_clip);
        if (c->has("aspect_ratio"))
            cam.aspect_ratio = c->getNumber("aspect_ratio", 1);
    }

    // ---- shapes
    std::vector<ShapeRec> shapes;
    std::map<std::string, uint32_t> shape_ids;
    if (const JsonValue* arr = doc.find("shapes")) {
        if (!arr->isArray())
            fail("Expected 'shapes' to be an array");
        for (const auto& s : arr->arr) {
            const std::string name = s.getString("name");
            if (name.empty())
                fail("Shape without a name");
            if (shape_ids.count(name))
                fail("Shape name '" + name + "' is used twice");
            shape_ids[name] = (uint32_t)shapes.size();
            handleShape(*sc, shapes, name, s, base_dir);
        }
    }

    static const JsonValue emptyArray = [] { JsonValue v; v.type = JsonValue::Array; return v; }();
    const JsonValue& bsdfs    = doc.find("bsdfs") ? *doc.find("bsdfs") : emptyArray;
    const JsonValue& jlights  = doc.find("lights") ? *doc.find("lights") : emptyArray;
    const JsonValue& entities = doc.find("entities") ? *doc.find("entities") : emptyArray;
    const JsonValue& textures = doc.find("textures") ? *doc.find("textures") : emptyArray;
    if (!textures.isArray())
        fail("Expected 'textures' to be an array");
    if (!bsdfs.isArray() || !jlights.isArray() || !entities.isArray())
        fail("Expected 'bsdfs', 'lights' and 'entities' to be arrays");
    if (doc.has("media") && !doc.find("media")->isArray())
        fail("Expected 'media' to be an array");

    // ---- which entities are area lights (LoaderLight::setupAreaLights)
    std::map<std::string, std::string> area_light_of_entity; // entity -> light name
    for (const auto& l : jlights.arr)
        if (l.getString("type") == "area")
            area_light_of_entity[l.getString("entity")] = l.getString("name");

    // ---- group entities by material (LoaderEntity.cpp:43-97)
    struct MatKey {
        std::string bsdf, light_entity;
        int medium_inner = -1, medium_outer = -1;
    };
    // media are numbered in the order entities name them (LoaderMedium::acquire, LoaderMedium.cpp:113-121)
    const JsonValue no_media;
    const JsonValue& media_defs = doc.find("media") ? *doc.find("media") : no_media;
    std::vector<std::string> acquired_media;
    auto acquireMedium = [&](const std::string& ename, const std::string& mname) -> int {
        if (mname.empty())
            return -1;
        const JsonValue* def = nullptr;
        if (media_defs.isArray())
            for (const auto& m : media_defs.arr)
                if (m.getString("name") == mname)
                    def = &m;
        if (!def)
            fail("Entity " + ename + " has unknown medium " + mname);
        for (size_t i = 0; i < acquired_media.size(); ++i)
            if (acquired_media[i] == mname)
                return (int)i;
        const std::string mtype = def->getString("type");
        ig_medium med{};
        if (mtype == "homogeneous" || mtype == "constant") {
            const V3 sa = getColor(*def, "sigma_a", V3(0, 0, 0), mname), ss = getColor(*def, "sigma_s", V3(0, 0, 0), mname);
            med.sigma_a[0] = sa.x, med.sigma_a[1] = sa.y, med.sigma_a[2] = sa.z;
            med.sigma_s[0] = ss.x, med.sigma_s[1] = ss.y, med.sigma_s[2] = ss.z;
            med.g    = getConstNumber(*def, "g", 0.0f, mname);
            med.type = IG_MEDIUM_HOMOGENEOUS;
        } else if (mtype == "vacuum") {
            med.type = IG_MEDIUM_VACUUM;
        } else {
            fail("No medium type '" + mtype + "' available");
        }
        acquired_media.push_back(mname);
        sc->media.push_back(med);
        return (int)acquired_media.size() - 1;
    };
    std::vector<MatKey> mat_keys;
    std::vector<std::vector<const JsonValue*>> groups;
    for (const auto& e : entities.arr) {
        const std::string ename = e.getString("name");
        const std::string bname = e.getString("bsdf");
        // LoaderEntity.cpp:47-55,108-113: an entity without a (known) bsdf or shape is reported and left out; the scene still loads
        bool known = !bname.empty() && shape_ids.count(e.getString("shape")) != 0;
        if (known) {
            known = false;
            for (const auto& b : bsdfs.arr)
                known |= b.getString("name") == bname;
        }
        if (!known)
            continue;
        // the medium interface is part of the material (LoaderEntity.cpp:57-96)
        const int m_in  = acquireMedium(ename, e.getString("inner_medium"));
        const int m_out = acquireMedium(ename, e.getString("outer_medium"));
        if (area_light_of_entity.count(ename)) {
            mat_keys.push_back(MatKey{ bname, ename, m_in, m_out });
            groups.emplace_back().push_back(&e);
        } else {
            size_t id = 0;
            for (; id < mat_keys.size(); ++id)
                if (mat_keys[id].bsdf == bname && mat_keys[id].light_entity.empty() && mat_keys[id].medium_inner == m_in && mat_keys[id].medium_outer == m_out)
                    break;
            if (id == mat_keys.size()) {
                mat_keys.push_back(MatKey{ bname, "", m_in, m_out });
                groups.emplace_back();
            }
            groups[id].push_back(&e);
        }
    }

    // ---- entities in material order (LoaderEntity.cpp:99-175)
    struct EmissiveEntity {
        uint32_t id, shape_id, mat_id;
        Affine transform;
    };
    std::map<std::string, EmissiveEntity> emissive;
    std::vector<EntityObject> objs, sphere_objs; // one scene BVH per shape provider (SceneBVHAdapter.h:110-129 per provider)
    BBox sceneBBox;
    uint32_t entityCount = 0;
    for (size_t materialID = 0; materialID < groups.size(); ++materialID) {
        for (const JsonValue* e : groups[materialID]) {
            const std::string ename = e->getString("name");
            const std::string sname = e->getString("shape");
            if (sname.empty())
                fail("Entity " + ename + " has no shape");
            auto sit = shape_ids.find(sname);
            if (sit == shape_ids.end())
                fail("Entity " + ename + " has unknown shape " + sname);
            const uint32_t shapeID = sit->second;
            const ShapeRec& shape  = shapes[shapeID];

            uint32_t flags = 0;
            if (e->getBool("camera_visible", true))
                flags |= 0x1;
            if (e->getBool("light_visible", true))
                flags |= 0x2;
            if (e->getBool("bounce_visible", true))
                flags |= 0x4;
            if (e->getBool("shadow_visible", true))
                flags |= 0x8;

            const Affine transform    = getTransform(*e);
            const Affine invTransform = inverse(transform);
            const BBox entityBox      = shape.bbox.transformed(transform);
            sceneBBox.extend(entityBox);

            if (area_light_of_entity.count(ename))
                emissive[ename] = EmissiveEntity{ entityCount, shapeID, (uint32_t)materialID, transform };

            const M3 toGlobalNormal = transpose(inverse(transform.L));
            writeEntity(sc->entities, invTransform, transform, toGlobalNormal, shapeID, (uint32_t)materialID);

            EntityObject obj;
            obj.bbox        = entityBox;
            obj.entity_id   = (int32_t)entityCount;
            obj.shape_id    = (int32_t)shapeID;
            obj.material_id = (int32_t)materialID;
            obj.user1       = shape.user1;
            obj.user2       = shape.user2;
            obj.flags       = flags;
            for (int c = 0; c < 3; ++c)
                for (int r = 0; r < 3; ++r)
                    obj.local[c * 3 + r] = invTransform.L.m[r][c];
            obj.local[9]  = invTransform.t.x;
            obj.local[10] = invTransform.t.y;
            obj.local[11] = invTransform.t.z;
            (shape.analytic ? sphere_objs : objs).push_back(obj);

            sc->entity_names.push_back(ename);
            ++entityCount;
        }
        sc->entity_per_material.push_back((int32_t)groups[materialID].size());
    }

    if (!objs.empty())
        build_scene_bvh8(objs, sc->scene_nodes, sc->scene_leaves);
    if (!sphere_objs.empty())
        build_scene_bvh8(sphere_objs, sc->sphere_nodes, sc->sphere_leaves);

    // ---- lights: infinite first, then finite (light_selector.art:26-46 id convention)
    std::vector<ig_light> infinite, finite;
    std::vector<LightEntry> hier_entries; // position / direction / flux per finite light (Light::position, direction, computeFlux)
    std::map<std::string, int32_t> finite_index_of_entity;
    TextureBank bank{ textures, base_dir, sc->textures, sc->texture_data, {}, &sc->expr_code, {} };
    bank.params = parameters;
    for (const auto& l : jlights.arr) {
        const std::string lname = l.getString("name");
        const std::string type  = l.getString("type");
        ig_light out;
        std::memset(&out, 0, sizeof(out));
        out.entity_id = -1;
        if (type == "area") {
            const std::string ent = l.getString("entity");
            auto it               = emissive.find(ent);
            if (it == emissive.end())
                fail("No entity named '" + ent + "' exists for area light '" + lname + "'");
            const ShapeRec& shape = shapes[it->second.shape_id];
            // AreaLight.cpp:50-62: the specialised samplers are used unless "optimize" is off (always for non-triangle shapes)
            const bool opt = l.getBool("optimize", true) || shape.analytic;
            if (opt && !shape.plane.has_value() && shape.sphere.has_value()) {
                // make_sphere_area_emitter (light/area.art:259-317). Area: compute_ellipsoid_area (shapes/sphere.art:21-28) over the
                // entity's toGlobal matrix — vec3_len2, i.e. SQUARED lengths, go into Knud Thomsen's formula there with p = 1.6,
                // which still gives 4 pi r^2 for a uniform scale; evaluated here once instead of per sample on the device.
                const Affine& T    = it->second.transform;
                const float r      = shape.sphere->radius;
                auto len2          = [&](V3 axis) { const V3 d = T.direction(axis * r); return dot(d, d); };
                const float l1 = len2(V3(1, 0, 0)), l2 = len2(V3(0, 1, 0)), l3 = len2(V3(0, 0, 1));
                const float P       = 1.6f;
                const float area    = 4 * Pi * std::pow((std::pow(l1 * l2, P / 2) + std::pow(l1 * l3, P / 2) + std::pow(l2 * l3, P / 2)) / 3, 1 / P);
                // the host's own area (flux, "power" -> radiance): approximate_ellipsoid_area, AreaLight.cpp:26-35
                const float w = norm(T.direction(V3(1, 0, 0) * r)), h = norm(T.direction(V3(0, 1, 0) * r)), dd = norm(T.direction(V3(0, 0, 1) * r));
                const float PH     = 1.6075f;
                const float m_area = 4 * Pi * std::pow((std::pow(w * h, PH) + std::pow(w * dd, PH) + std::pow(h * dd, PH)) / 3, 1 / PH);
                V3 radiance, cache;
                if (l.has("power")) {
                    cache    = getColor(l, "power", V3(m_area * Pi, m_area * Pi, m_area * Pi), lname);
                    radiance = cache * ((1 / Pi) / m_area);
                } else {
                    radiance = getColor(l, "radiance", V3(1, 1, 1), lname);
                    cache    = radiance * (m_area * Pi);
                }
                out.type      = IG_LIGHT_SPHERE;
                out.entity_id = (int32_t)it->second.id;
                out.d[0] = shape.sphere->origin.x, out.d[1] = shape.sphere->origin.y, out.d[2] = shape.sphere->origin.z, out.d[3] = r;
                out.d[4] = radiance.x, out.d[5] = radiance.y, out.d[6] = radiance.z, out.d[7] = area;
                finite_index_of_entity[ent] = (int32_t)finite.size();
                // AreaLight.cpp:73-82: position = the transformed centre, no direction
                hier_entries.push_back(LightEntry{ T.point(shape.sphere->origin), V3(0, 0, 1), -((cache.x + cache.y + cache.z) / 3), (int32_t)finite.size() });
                finite.push_back(out);
                continue;
            }
            if (!shape.plane.has_value() || !opt) {
                if (shape.analytic)
                    fail("Area light '" + lname + "': Given entity '" + ent + "' primitive type is not triangular"); // AreaLight.cpp:93
                // AreaLight.cpp:60-62,84-92,206-227: no specialised sampler -> make_shape_area_emitter over the entity's triangles.
                // Position = centre of the bounding box, no direction (negative flux in the hierarchy), area = mesh area times
                // the transform's approximate area scale (AreaLight.cpp:12-24).
                const Affine& T = it->second.transform;
                const V3 ls     = shape.bbox.diameter();
                const float w   = norm(T.direction(V3(1, 0, 0) * ls.x));
                const float h   = norm(T.direction(V3(0, 1, 0) * ls.y));
                const float dd  = norm(T.direction(V3(0, 0, 1) * ls.z));
                const float half_area = ls.x * ls.y + ls.x * ls.z + ls.y * ls.z; // BoundingBox::halfArea
                const float area      = shape.area * ((w * h + w * dd + h * dd) / half_area);
                V3 radiance, cache;
                if (l.has("power")) {
                    cache    = getColor(l, "power", V3(area * Pi, area * Pi, area * Pi), lname); // mColor_Cache (AreaLight.cpp:98-107)
                    radiance = cache * ((1 / Pi) / area);                                          // color_mulf(power, flt_inv_pi / mArea)
                } else {
                    radiance = getColor(l, "radiance", V3(1, 1, 1), lname);
                    cache    = radiance * (area * Pi);
                }
                out.type      = IG_LIGHT_MESH_AREA;
                out.entity_id = (int32_t)it->second.id;
                out.d[0] = radiance.x, out.d[1] = radiance.y, out.d[2] = radiance.z;
                finite_index_of_entity[ent] = (int32_t)finite.size();
                hier_entries.push_back(LightEntry{ T.point(shape.bbox.center()), V3(0, 0, 1), -((cache.x + cache.y + cache.z) / 3), (int32_t)finite.size() });
                finite.push_back(out);
                continue;
            }
            // AreaLight.cpp:138-170 + "SimplePlaneLight" layout (light/area.art:416-440)
            const Affine& T    = it->second.transform;
            const V3 origin    = T.point(shape.plane->origin);
            const V3 x_axis    = T.direction(shape.plane->x_axis);
            const V3 y_axis    = T.direction(shape.plane->y_axis);
            const V3 cr        = cross(x_axis, y_axis);
            const V3 normal    = normalized(cr);
            const float area   = norm(cr);
            // "power" -> color_mulf(power, flt_inv_pi / mArea) (AreaLight.cpp:222-225)
            const V3 radiance  = l.has("power") ? getColor(l, "power", V3(area * Pi, area * Pi, area * Pi), lname) * ((1 / Pi) / area) : getColor(l, "radiance", V3(1, 1, 1), lname);
            const auto& tc     = shape.plane->texcoords;
            const float d[24]  = { origin.x, origin.y, origin.z, normal.x,
                                   x_axis.x, x_axis.y, x_axis.z, normal.y,
                                   y_axis.x, y_axis.y, y_axis.z, normal.z,
                                   tc[0].x, tc[0].y, tc[1].x, tc[1].y,
                                   tc[2].x, tc[2].y, tc[3].x, tc[3].y,
                                   radiance.x, radiance.y, radiance.z, area };
            out.type      = IG_LIGHT_PLANE;
            out.entity_id = (int32_t)it->second.id;
            std::memcpy(out.d, d, sizeof(d));
            finite_index_of_entity[ent] = (int32_t)finite.size();
            // AreaLight.cpp:68-70,96-107: centre of the plane, its normal, flux = mean(radiance * area * pi)
            const V3 cache = radiance * (area * Pi);
            hier_entries.push_back(LightEntry{ origin + x_axis * 0.5f + y_axis * 0.5f, normal, (cache.x + cache.y + cache.z) / 3, (int32_t)finite.size() });
            finite.push_back(out);
        } else if (type == "point") {
            // PointLight.cpp:16-71 ("SimplePointLight": pos, 0, intensity, 0)
            const V3 pos = l.has("position") ? getVector3(*l.find("position"), "position") : V3(0, 0, 0);
            V3 intensity;
            if (l.has("power")) {
                const V3 power = getColor(l, "power", V3(4 * Pi, 4 * Pi, 4 * Pi), lname);
                intensity      = V3(power.x / (4 * Pi), power.y / (4 * Pi), power.z / (4 * Pi));
            } else {
                const V3 cache = getColor(l, "intensity", V3(1, 1, 1), lname) * (4 * Pi);
                intensity      = V3(cache.x / (4 * Pi), cache.y / (4 * Pi), cache.z / (4 * Pi));
            }
            out.type = IG_LIGHT_POINT;
            out.d[0] = pos.x, out.d[1] = pos.y, out.d[2] = pos.z;
            out.d[4] = intensity.x, out.d[5] = intensity.y, out.d[6] = intensity.z;
            // PointLight.cpp:17-31: no direction (stored as negative flux), flux = mean(intensity * 4 pi)
            const V3 cache = intensity * (4 * Pi);
            hier_entries.push_back(LightEntry{ pos, V3(0, 0, 1), -((cache.x + cache.y + cache.z) / 3), (int32_t)finite.size() });
            finite.push_back(out);
        } else if (type == "spot") {
            // SpotLight.cpp:13-97, light/spot.art: position, direction, cutoff / falloff (degrees), intensity or power
            const V3 pos     = l.has("position") ? getVector3(*l.find("position"), "position") : V3(0, 0, 0);
            V3 dir           = l.has("direction") ? getVector3(*l.find("direction"), "direction") : getSunAngles(l).direction();
            const float dl   = std::sqrt(dot(dir, dir));
            dir              = dl > 0 ? dir * (1 / dl) : V3(0, 0, 1);
            const float cutoff  = getConstNumber(l, "cutoff", 30.0f, lname) * Deg2Rad;
            const float falloff = getConstNumber(l, "falloff", 20.0f, lname) * Deg2Rad;
            const float factor  = 2 * Pi * (1 - 0.5f * (std::cos(cutoff) + std::cos(falloff))); // power_factor
            V3 cache; // mColor_Cache: power
            if (l.has("power"))
                cache = getColor(l, "power", V3(factor, factor, factor), lname);
            else
                cache = getColor(l, "intensity", V3(1, 1, 1), lname) * factor;
            const V3 intensity = cache * (1 / factor);
            out.type = IG_LIGHT_SPOT;
            out.d[0] = pos.x, out.d[1] = pos.y, out.d[2] = pos.z, out.d[3] = std::cos(cutoff);
            out.d[4] = dir.x, out.d[5] = dir.y, out.d[6] = dir.z, out.d[7] = std::cos(falloff);
            out.d[8] = intensity.x, out.d[9] = intensity.y, out.d[10] = intensity.z;
            hier_entries.push_back(LightEntry{ pos, dir, (cache.x + cache.y + cache.z) / 3, (int32_t)finite.size() });
            finite.push_back(out);
        } else if (type == "directional" || type == "direction" || type == "distant") {
            // DirectionalLight.cpp, light/directional.art: an infinite delta light
            // LoaderUtils::getDirection: an explicit vector, or elevation / azimuth, or the sun of a date, time and place
            V3 dir         = l.has("direction") ? getVector3(*l.find("direction"), "direction") : getSunAngles(l).direction();
            const float dl = std::sqrt(dot(dir, dir));
            dir            = dl > 0 ? dir * (1 / dl) : V3(0, 0, 1);
            const V3 irr   = getColor(l, "irradiance", V3(1, 1, 1), lname);
            out.type       = IG_LIGHT_DIRECTIONAL;
            out.d[0] = dir.x, out.d[1] = dir.y, out.d[2] = dir.z;
            out.d[4] = irr.x, out.d[5] = irr.y, out.d[6] = irr.z;
            infinite.push_back(out);
        } else if (type == "env" || type == "constant") {
            // EnvironmentLight.cpp:40-98: a constant radiance bakes to a 1x1 texture, so the reference
            // builds make_environment_light (uniform sphere sampling), not the CDF-sampled variant.
            const JsonValue* rad = l.find("radiance");
            bool textured        = false;
            if (rad && rad->isString())
                for (const auto& t : textures.arr)
                    if (t.getString("name") == rad->str && (t.getString("type") == "image" || t.getString("type") == "bitmap"))
                        textured = true;
            if (textured) {
                // EnvironmentLight.cpp:14-98: the radiance texture is baked at >= 1024 x 512 (bake.art:5-6: uv = pixel / (size - 1))
                // and sampled through the marginal / conditional CDF of the baked image; "cdf": "none" keeps uniform sampling
                std::string method = l.getString("cdf", "conditional");
                for (char& ch : method)
                    ch = (char)std::tolower((unsigned char)ch);
                if (method != "conditional" && method != "" && method != "none")
                    fail("Environment light '" + lname + "': cdf method '" + method + "' is not supported by the HIP backend (only 'conditional' and 'none')");
                const int tex_id    = bank.get(rad->str, lname);
                const ig_texture tx = sc->textures[tex_id];
                size_t bw = std::max<size_t>(1024, tx.width), bh = std::max<size_t>(512, tx.height);
                uint32_t cdf_off = 0;
                if (method == "none") {
                    bw = bh = 0; // no table: the device samples directions uniformly (make_environment_light, env.art:161-164)
                } else {
                    std::vector<float> baked(bw * bh * 3);
                    for (size_t y = 0; y < bh; ++y)
                        for (size_t x = 0; x < bw; ++x) {
                            const V3 c = lookupTexture(tx, sc->texture_data, (float)x / (float)(bw - 1), (float)y / (float)(bh - 1));
                            baked[(y * bw + x) * 3] = c.x, baked[(y * bw + x) * 3 + 1] = c.y, baked[(y * bw + x) * 3 + 2] = c.z;
                        }
                    cdf_off = appendEnvironmentCdf(sc->cdf_data, baked, bw, bh, l.getBool("compensate", true));
                }
                const V3 scale         = getColor(l, "scale", V3(1, 1, 1), lname);
                // "_transform" = transform.linear().transpose().inverse() (EnvironmentLight.cpp:56)
                const M3 T   = l.has("transform") ? inverse(transpose(getTransform(l).L)) : M3{};
                out.type     = IG_LIGHT_ENV_TEXTURED;
                out.d[0] = scale.x, out.d[1] = scale.y, out.d[2] = scale.z;
                for (int c = 0; c < 3; ++c)
                    for (int r = 0; r < 3; ++r)
                        out.d[3 + c * 3 + r] = T.m[r][c];
                const uint32_t ints[4] = { (uint32_t)tex_id, cdf_off, (uint32_t)bw, (uint32_t)bh };
                std::memcpy(&out.d[12], ints, sizeof(ints));
                infinite.push_back(out);
                continue;
            }
            if (rad && rad->isString() && rad->str.rfind("color(", 0) != 0)
                fail("Environment light '" + lname + "': only constant colours and bitmap textures are supported as radiance by the HIP backend");
            // a "transform" only rotates the lookup direction of make_environment_light_function_spherical (env.art:74-96:
            // the sampled direction itself is not transformed), so it has no effect on a constant radiance
            const V3 radiance = getColor(l, "radiance", V3(1, 1, 1), lname);
            const V3 scale    = getColor(l, "scale", V3(1, 1, 1), lname);
            out.type          = IG_LIGHT_ENV;
            out.d[0] = scale.x * radiance.x, out.d[1] = scale.y * radiance.y, out.d[2] = scale.z * radiance.z; // color_mul(scale, tex), env.art:163
            infinite.push_back(out);
        } else if (type.rfind("cie", 0) == 0) {
            // CIELight.cpp:6-107, LoaderLight.cpp:55-97
            int kind;
            if (type == "cie_uniform" || type == "cieuniform")
                kind = IG_CIE_UNIFORM;
            else if (type == "cie_cloudy" || type == "ciecloudy")
                kind = IG_CIE_CLOUDY;
            else if (type == "cie_clear" || type == "cieclear")
                kind = IG_CIE_CLEAR;
            else if (type == "cie_intermediate" || type == "cieintermediate")
                kind = IG_CIE_INTERMEDIATE;
            else
                fail("Light '" + lname + "': type '" + type + "' is not supported by the HIP backend");
            const V3 scale  = getColor(l, "scale", V3(1, 1, 1), lname);
            const V3 zenith = getColor(l, "zenith", V3(1, 1, 1), lname);
            const V3 ground = getColor(l, "ground", V3(1, 1, 1), lname);
            const float gb  = getConstNumber(l, "ground_brightness", 0.2f, lname);
            out.type   = IG_LIGHT_CIE;
            out.pad[0] = kind;
            out.pad[1] = l.getBool("has_ground", true) ? 1 : 0;
            out.d[0] = zenith.x, out.d[1] = zenith.y, out.d[2] = zenith.z;
            out.d[3] = ground.x, out.d[4] = ground.y, out.d[5] = ground.z;
            out.d[6] = gb;
            V3 sun(0, 0, 1); // LoaderUtils::getEA default direction
            if (kind == IG_CIE_CLEAR || kind == IG_CIE_INTERMEDIATE) {
                const SunAngles ea = getSunAngles(l);
                float elevation    = ea.elevation;
                const V3 sun_dir   = ea.direction();
                if (elevation > 87 * Deg2Rad)
                    elevation = 87 * Deg2Rad;
                const bool clear      = kind == IG_CIE_CLEAR;
                const float turbidity = l.getNumber("turbidity", 2.45f);
                const float SkyIllum  = 203;
                float zb              = (1.376f * turbidity - 1.81f) * std::tan(elevation) + 0.38f;
                if (!clear)
                    zb = (zb + 8.6f * sun_dir.y + 0.123f) / 2;
                zb = std::max(0.0f, zb * 1000 / SkyIllum);
                float factor;
                if (clear)
                    factor = 0.274f * (0.91f + 10 * std::exp(-3 * (Pi / 2 - elevation)) + 0.45f * sun_dir.y * sun_dir.y);
                else
                    factor = (2.739f + 0.9891f * std::sin(0.3119f + 2.6f * elevation)) * std::exp(-(Pi / 2 - elevation) * (0.4441f + 1.48f * elevation));
                // skylight_normalization_factor (CIELight.cpp:25-36)
                const float cl[5] = { 2.766521f, 0.547665f, -0.369832f, 0.009237f, 0.059229f };
                const float im[5] = { 3.5556f, -2.7152f, -1.3081f, 1.0660f, 0.60227f };
                const float* arr  = clear ? cl : im;
                const float x     = (elevation - Pi / 4) / (Pi / 4);
                float nf          = arr[4];
                for (int i = 3; i >= 0; --i)
                    nf = nf * x + arr[i];
                const float norm_factor     = nf * (1 / Pi) / factor;
                const float SunIllum        = 208;
                const float solarbrightness = 1.5e9f / SunIllum * (1.147f - 0.147f / std::max(sun_dir.y, 0.16f));
                const float additive        = 6e-5f * (1 / Pi) * solarbrightness * sun_dir.y * (clear ? 1.0f : 0.15f);
                out.d[7] = zb / factor;
                out.d[8] = zb * norm_factor + additive;
                sun      = sun_dir;
            }
            out.d[9] = sun.x, out.d[10] = sun.y, out.d[11] = sun.z;
            out.d[12] = scale.x, out.d[13] = scale.y, out.d[14] = scale.z;
            const M3 T = l.has("transform") ? inverse(transpose(getTransform(l).L)) : M3{};
            for (int c = 0; c < 3; ++c)
                for (int r = 0; r < 3; ++r)
                    out.d[15 + c * 3 + r] = T.m[r][c];
            infinite.push_back(out);
        } else if (type == "perez") {
            // PerezLight.cpp:10-136 + make_perez_light_from_model / make_perez_light_raw (light/perez.art:292-388)
            const bool has_ground = l.getBool("has_ground", true);
            const bool has_sun    = l.getBool("has_sun", true);
            std::string mode      = l.getString("output", "visibleradiance");
            for (char& ch : mode)
                ch = (char)std::tolower((unsigned char)ch);
            const int output = mode == "visibleradiance" ? 0 : (mode == "solarradiance" ? 1 : 2);
            const V3 ground  = getColor(l, "ground", V3(0.2f, 0.2f, 0.2f), lname);
            const V3 tint    = getColor(l, "color", V3(1, 1, 1), lname);
            V3 sun_dir       = getSunAngles(l).direction(); // "Scene to Light (Local)"
            {
                const float dl = std::sqrt(dot(sun_dir, sun_dir));
                sun_dir        = dl > 0 ? sun_dir * (1 / dl) : V3(0, 0, 1);
            }
            const float day = l.has("day_of_the_year") ? getConstNumber(l, "day_of_the_year", 0.0f, lname)
                                                       : (float)perez::dayOfTheYear(l.getInt("year", 2020), l.getInt("month", 5), l.getInt("day", 6));
            const float sin_altitude   = std::min(std::max(sun_dir.y, -1.0f), 1.0f);
            const float solar_altitude = std::asin(sin_altitude);
            const float solar_zenith   = Pi / 2 - solar_altitude;
            perez::Model model;
            if (l.has("direct_irradiance") || l.has("diffuse_irradiance")) {
                const float diffuse = getConstNumber(l, "diffuse_irradiance", 1.0f, lname);
                if (l.has("direct_horizontal_irradiance") && !l.has("direct_irradiance"))
                    model = perez::fromIrradiance(diffuse, getConstNumber(l, "direct_horizontal_irradiance", 1.0f, lname) / std::cos(solar_zenith), solar_zenith, day);
                else
                    model = perez::fromIrradiance(diffuse, getConstNumber(l, "direct_irradiance", 1.0f, lname), solar_zenith, day);
            } else {
                model = perez::fromBrightnessClearness(getConstNumber(l, "brightness", 0.2f, lname), getConstNumber(l, "clearness", 1.0f, lname), solar_zenith, day);
            }
            const float WhiteEfficiency = 179; // color_builtins::white_efficiency (core/color.art:78)
            auto safeDiv                = [](float a, float b) { return std::fabs(b) <= 1.1920928955e-07f ? 0.0f : a / b; }; // safe_div, core/common.art:263
            const float integrand       = perez::integrate(solar_zenith, model.params);
            const float diffnorm        = safeDiv(output == 0 ? model.diffuse_illum / WhiteEfficiency : (output == 1 ? model.diffuse_irrad : model.diffuse_illum), integrand);
            const float half_angle      = Deg2Rad * (0.533f / 2); // flt_sun_radius_deg
            const float sun_factor      = 2 * Pi * (1 - std::cos(half_angle));
            const V3 sun_color          = tint * (output == 0 ? model.direct_illum / WhiteEfficiency : (output == 1 ? model.direct_irrad : model.direct_illum));
            const V3 sky_color          = tint * diffnorm;
            const V3 zenith             = sky_color * perez::eval(sin_altitude, 1, model.params);
            float normfactor;
            if (model.clearness == 1) {
                normfactor = 0.777778f;
            } else if (model.clearness < 6) {
                const float f2 = (2.739f + 0.9891f * std::sin(0.3119f + 2.6f * solar_altitude)) * std::exp(-solar_zenith * (0.4441f + 1.48f * solar_altitude));
                const float x  = solar_altitude / (Pi / 4) - 1;
                const float nc = (((0.60227f * x + 1.0660f) * x - 1.3081f) * x - 2.7152f) * x + 3.5556f;
                normfactor     = safeDiv(nc, f2) / Pi;
            } else {
                const float f2 = 0.274f * (0.91f + 10 * std::exp(-3 * solar_zenith) + 0.45f * sin_altitude * sin_altitude);
                const float x  = solar_altitude / (Pi / 4) - 1;
                const float nc = (((0.059229f * x + 0.009237f) * x - 0.369832f) * x + 0.547665f) * x + 2.766521f;
                normfactor     = safeDiv(nc, f2) / Pi;
            }
            const bool sun_on      = has_sun && model.clearness > 1;
            const V3 sun_part      = sun_on ? sun_color * (std::fabs(sin_altitude) / Pi) : V3(0, 0, 0);
            const V3 sum           = sun_part + zenith * normfactor;
            const V3 actual_ground = V3(ground.x * sum.x, ground.y * sum.y, ground.z * sum.z);
            const V3 actual_sun    = sun_on ? sun_color * (1 / sun_factor) : V3(0, 0, 0);
            // "_transform" (PerezLight.cpp:48-51) or make_cie_sky_transform(up) = make_orthonormal_mat3x3_y (core/matrix.art:34-42)
            M3 T;
            if (l.has("transform")) {
                T = inverse(transpose(getTransform(l).L));
            } else {
                V3 n = l.has("up") ? getVector3(*l.find("up"), "up") : V3(0, 1, 0);
                n    = n * (1 / std::sqrt(dot(n, n)));
                const float sign = std::copysign(1.0f, n.y);
                const float a    = -1 / (sign + n.y);
                const float b    = n.x * n.z * a;
                const V3 t(1 + sign * n.x * n.x * a, -sign * n.x, sign * b);
                const V3 bt(b, -n.z, sign + n.z * n.z * a);
                const V3 cols[3] = { t, n, bt };
                for (int c = 0; c < 3; ++c)
                    T.m[0][c] = cols[c].x, T.m[1][c] = cols[c].y, T.m[2][c] = cols[c].z;
            }
            // the record of a function environment (IG_LIGHT_CIE, kind IG_CIE_PEREZ); with a sun the same plus the sun terms
            out.type   = has_sun ? IG_LIGHT_PEREZ : IG_LIGHT_CIE;
            out.pad[0] = IG_CIE_PEREZ;
            out.pad[1] = has_ground ? 1 : 0;
            out.d[0] = sky_color.x, out.d[1] = sky_color.y, out.d[2] = sky_color.z;
            out.d[3] = actual_ground.x, out.d[4] = actual_ground.y, out.d[5] = actual_ground.z;
            out.d[6] = model.params[0], out.d[7] = model.params[1], out.d[8] = model.params[2];
            out.d[9] = sun_dir.x, out.d[10] = sun_dir.y, out.d[11] = sun_dir.z;
            out.d[12] = model.params[3], out.d[13] = model.params[4];
            out.d[14] = std::cos(half_angle);
            for (int c = 0; c < 3; ++c)
                for (int r = 0; r < 3; ++r)
                    out.d[15 + c * 3 + r] = T.m[r][c];
            if (has_sun) {
                const V3 g = T * sun_dir; // mat3x3_mul(transform, l_sun_dir): the sun model works in scene space
                out.d[24] = actual_sun.x, out.d[25] = actual_sun.y, out.d[26] = actual_sun.z;
                out.d[27] = g.x, out.d[28] = g.y, out.d[29] = g.z;
            }
            infinite.push_back(out);
        } else if (type == "sun") {
            // SunLight.cpp:11-57, light/sun.art:1-48: an infinite cone light; direction = from the scene towards the sun
            V3 dir           = getSunAngles(l).direction();
            const float dl   = std::sqrt(dot(dir, dir));
            dir              = dl > 0 ? dir * (1 / dl) : V3(0, 0, 1);
            const float angle = getConstNumber(l, "angle", 0.533f, lname); // flt_sun_radius_deg
            const float half  = angle / 2 * Deg2Rad;
            V3 radiance;
            if (l.has("radiance")) {
                radiance = getColor(l, "radiance", V3(1, 1, 1), lname);
            } else {
                // irradiance / sun_area_from_srad(rad(angle / 2)) (SunLight.cpp:51-52, sun.art:5)
                radiance = getColor(l, "irradiance", V3(1, 1, 1), lname) * (1 / (Pi * half * half));
            }
            out.type = IG_LIGHT_SUN;
            out.d[0] = dir.x, out.d[1] = dir.y, out.d[2] = dir.z;
            out.d[3] = std::cos(half);
            out.d[4] = radiance.x, out.d[5] = radiance.y, out.d[6] = radiance.z;
            infinite.push_back(out);
        } else if (type == "sky") {
            // SkyLight.cpp:30-75 + skysun/SkyModel.cpp:9-54: the Hosek-Wilkie model (hosek.h) evaluated on the upper hemisphere into a
            // 512 x 256 RGBA float image (row = angle from the zenith, column = azimuth shifted by pi / 4), which then is an ordinary
            // textured environment light: bilinear filter, repeat border, marginal / conditional CDF of the image itself (no
            // baking, no MIS compensation: setup_cdf2d(ctx, path, true, false)). As written there, what the model is handed as
            // "solar elevation" is pi / 2 - elevation.
            const SunAngles ea    = getSunAngles(l);
            const V3 ground       = l.has("ground") ? getVector3(*l.find("ground"), "ground") : V3(0.8f, 0.8f, 0.8f);
            const float turbidity = getConstNumber(l, "turbidity", 3.0f, lname);
            constexpr size_t ResAz = 512, ResEl = 256; // skysun/SkySunConfig.h
            const float solar_elevation = Pi / 2 - ea.elevation;
            const float sun_se = std::sin(solar_elevation), sun_ce = std::cos(solar_elevation);
            const hosek::ChannelState state[3] = { hosek::init(0, turbidity, ground.x, solar_elevation), hosek::init(1, turbidity, ground.y, solar_elevation),
                                                   hosek::init(2, turbidity, ground.z, solar_elevation) };
            // The reference writes the image to an EXR file (rows top to bottom, alpha dropped) and loads it back through Image::load,
            // which gives RGB files an alpha of 1 and flips the rows (Image.cpp:108-121,709): row y of the model is row ResEl - 1 - y of
            // the texture and of the image the CDF is computed from.
            std::vector<float> rgba(ResAz * ResEl * 4), rgb(ResAz * ResEl * 3);
            for (size_t y = 0; y < ResEl; ++y) {
                const float theta = (Pi / 2) * y / (float)ResEl; // ELEVATION_RANGE * y / count
                const float st = std::sin(theta), ct = std::cos(theta);
                for (size_t x = 0; x < ResAz; ++x) {
                    float azimuth = (2 * Pi) * x / (float)ResAz - Pi / 4;
                    if (azimuth < 0)
                        azimuth += 2 * Pi;
                    const float cos_gamma = ct * sun_ce + st * sun_se * std::cos(azimuth - ea.azimuth);
                    const float gamma     = std::acos(std::min(1.0f, std::max(-1.0f, cos_gamma)));
                    const size_t at       = (ResEl - 1 - y) * ResAz + x;
                    for (int k = 0; k < 3; ++k) {
                        constexpr float CIEYSum = 106.856980f;
                        const float radiance    = (float)hosek::radiance(state[k], theta, gamma) / CIEYSum;
                        rgba[at * 4 + k] = rgb[at * 3 + k] = std::max(0.0f, radiance);
                    }
                    rgba[at * 4 + 3] = 1;
                }
            }
            ig_texture rec{};
            rec.width = (uint32_t)ResAz, rec.height = (uint32_t)ResEl;
            rec.channels = IG_TEX_FLOAT_BIT | 4;
            rec.filter   = IG_TEX_BILINEAR;
            rec.wrap_u = rec.wrap_v = IG_WRAP_REPEAT;
            sc->texture_data.resize((sc->texture_data.size() + 15) & ~(size_t)15);
            rec.offset = sc->texture_data.size();
            sc->texture_data.resize(rec.offset + rgba.size() * sizeof(float));
            std::memcpy(&sc->texture_data[rec.offset], rgba.data(), rgba.size() * sizeof(float));
            const int tex_id = (int)sc->textures.size();
            sc->textures.push_back(rec);
            const uint32_t cdf_off = appendEnvironmentCdf(sc->cdf_data, rgb, ResAz, ResEl, false);
            const V3 scale         = getColor(l, "scale", V3(1, 1, 1), lname);
            const M3 T             = l.has("transform") ? inverse(transpose(getTransform(l).L)) : M3{}; // SkyLight.cpp:58
            out.type               = IG_LIGHT_ENV_TEXTURED;
            out.d[0] = scale.x, out.d[1] = scale.y, out.d[2] = scale.z;
            for (int c = 0; c < 3; ++c)
                for (int r = 0; r < 3; ++r)
                    out.d[3 + c * 3 + r] = T.m[r][c];
            const uint32_t ints[4] = { (uint32_t)tex_id, cdf_off, (uint32_t)ResAz, (uint32_t)ResEl };
            std::memcpy(&out.d[12], ints, sizeof(ints));
            infinite.push_back(out);
        } else {
            fail("Light '" + lname + "': type '" + type + "' is not supported by the HIP backend");
        }
    }
    sc->lights = infinite;
    sc->lights.insert(sc->lights.end(), finite.begin(), finite.end());

    // ---- materials
    AuxMaterials aux;
    aux.base = (int32_t)mat_keys.size();
    for (size_t m = 0; m < mat_keys.size(); ++m) {
        ig_material mat = lowerBsdf(mat_keys[m].bsdf, bsdfs, textures, bank, aux);
        if (!mat_keys[m].light_entity.empty())
            mat.light_id = (int32_t)infinite.size() + finite_index_of_entity.at(mat_keys[m].light_entity);
        mat.pad[2] = ((mat_keys[m].medium_inner + 1) & 0xFFFF) | ((mat_keys[m].medium_outer + 1) << 16);
        sc->materials.push_back(mat);
        sc->material_names.push_back(mat_keys[m].bsdf);
    }
    for (size_t i = 0; i < aux.list.size(); ++i) {
        sc->materials.push_back(aux.list[i]);
        sc->material_names.push_back(aux.names[i]);
        sc->entity_per_material.push_back(0);
    }

    // ---- light selector (LoaderLight.cpp:423-460: <= 1 light -> uniform)
    tech.light_selector = IG_SELECTOR_UNIFORM;
    if (sc->lights.size() > 1 && !selector.empty() && selector != "uniform") {
        if (selector == "simple") {
            // LoaderLight.cpp:440-447,455-478: CDF::computeForArray over computeFlux of the finite lights (CDF.cpp:14-44). Without
            // finite lights there is nothing to build a CDF from: uniform.
            if (!finite.empty()) {
                tech.light_selector = IG_SELECTOR_SIMPLE;
                std::vector<float>& cdf = sc->light_cdf;
                cdf.resize(finite.size());
                for (size_t i = 0; i < finite.size(); ++i)
                    cdf[i] = (i ? cdf[i - 1] : 0.0f) + std::abs(hier_entries[i].flux); // (a light without direction carries its flux negated)
                const float sum = cdf.back();
                if (sum > 1e-5f) {
                    const float n = 1.0f / sum;
                    for (float& v : cdf)
                        v *= n;
                } else {
                    const float n = 1.0f / (float)cdf.size();
                    for (size_t x = 1; x < cdf.size(); ++x)
                        cdf[x - 1] = (float)x * n;
                }
                cdf.back() = 1;
            }
        } else if (selector != "hierarchy")
            fail("Light selector '" + selector + "' is not supported by the HIP backend (only 'uniform', 'simple' and 'hierarchy')");
        // make_hierarchy_light_selector falls back to uniform without finite lights (light_selector.art:81-83)
        else if (!finite.empty()) {
            tech.light_selector = IG_SELECTOR_HIERARCHY;
            buildLightHierarchy(hier_entries, sc->light_hierarchy, sc->light_codes);
        }
    }

    // ---- large BVHs: child boxes on per-node 8-bit grids, so that the HIP device can keep a node in one 128-byte line
    // (bvh.cpp quantise_node8; IGH_NODE_QUANT=0 never, 1 always, default: from 64 MB of Node8 records on — what the GPU's L2s no longer hold)
    {
        size_t node_bytes = (sc->scene_nodes.size() + sc->sphere_nodes.size()) * sizeof(ig_node8);
        for (const auto& r : sc->primbvh_nodes)
            node_bytes += (size_t)r.second * sizeof(ig_node8);
        const char* e    = std::getenv("IGH_NODE_QUANT");
        const bool quant = e && *e ? std::atoi(e) != 0 : node_bytes > ((size_t)64 << 20);
        if (quant) {
            for (const auto& r : sc->primbvh_nodes)
                quantise_node8(reinterpret_cast<ig_node8*>(sc->primbvh.data() + r.first), r.second);
            quantise_node8(sc->scene_nodes.data(), sc->scene_nodes.size());
            quantise_node8(sc->sphere_nodes.data(), sc->sphere_nodes.size());
        }
    }

    // ---- publish
    igd_scene& t         = sc->tables;
    t.entities           = sc->entities.data();
    t.entity_count       = entityCount;
    t.shape_lookups      = sc->shape_lookups.data();
    t.shape_count        = (uint32_t)sc->shape_lookups.size();
    t.shape_data         = sc->shape_data.data();
    t.shape_data_size    = sc->shape_data.size();
    t.primbvh            = sc->primbvh.data();
    t.primbvh_size       = sc->primbvh.size();
    t.scene_nodes        = sc->scene_nodes.data();
    t.scene_node_count   = (uint32_t)sc->scene_nodes.size();
    t.scene_leaves       = sc->scene_leaves.data();
    t.scene_leaf_count   = (uint32_t)sc->scene_leaves.size();
    t.sphere_nodes       = sc->sphere_nodes.data();
    t.sphere_node_count  = (uint32_t)sc->sphere_nodes.size();
    t.sphere_leaves      = sc->sphere_leaves.data();
    t.sphere_leaf_count  = (uint32_t)sc->sphere_leaves.size();
    t.light_cdf          = sc->light_cdf.data();
    t.light_cdf_count    = (uint32_t)sc->light_cdf.size();
    t.media              = sc->media.data();
    t.media_count        = (uint32_t)sc->media.size();
    t.expr_code          = sc->expr_code.empty() ? nullptr : sc->expr_code.data();
    t.expr_code_count    = (uint32_t)sc->expr_code.size();
    t.materials          = sc->materials.data();
    t.material_count     = (uint32_t)sc->materials.size();
    t.entity_per_material = sc->entity_per_material.data();
    t.lights             = sc->lights.data();
    t.light_count        = (uint32_t)sc->lights.size();
    t.infinite_light_count = (uint32_t)infinite.size();
    t.light_hierarchy    = sc->light_hierarchy.empty() ? nullptr : sc->light_hierarchy.data();
    t.light_hierarchy_nodes = (uint32_t)(sc->light_hierarchy.size() / 8);
    t.light_codes        = sc->light_codes.empty() ? nullptr : sc->light_codes.data();
    sc->texture_data.resize(sc->texture_data.size() + 16); // vector loads never run past the allocation
    t.textures           = sc->textures.empty() ? nullptr : sc->textures.data();
    t.texture_count      = (uint32_t)sc->textures.size();
    t.texture_data       = sc->texture_data.data();
    t.texture_data_size  = sc->texture_data.size();
    t.cdf_data           = sc->cdf_data.empty() ? nullptr : sc->cdf_data.data();
    t.cdf_data_count     = sc->cdf_data.size();
    if (!cam_has_transform) {
        // no transform: a view over the whole scene (PerspectiveCamera.cpp:77-101, FishLensCamera.cpp:76-101)
        cam.dir[0] = 0, cam.dir[1] = 0, cam.dir[2] = -1;
        cam.up[0] = 0, cam.up[1] = 1, cam.up[2] = 0;
        cam.eye[0] = cam.eye[1] = cam.eye[2] = 0;
        if (entityCount) {
            const V3 diam = sceneBBox.diameter();
            float a = diam.x / 2, b = diam.y / 2, half = 60.0f * Deg2Rad / 2;
            if (cam.type == IG_CAMERA_PERSPECTIVE) {
                const float aspect = cam.aspect_ratio > 0 ? cam.aspect_ratio : (float)width / (float)height;
                a    = diam.x / (2 * (cam.fov_is_vertical ? aspect : 1));
                b    = diam.y / (2 * (!cam.fov_is_vertical ? aspect : 1));
                half = cam.fov / 2;
            }
            const float sn = std::sin(half);
            const float d  = std::abs(sn) <= 1.1920928955e-07f ? 0 : std::max(a, b) * std::sqrt(1 / (sn * sn) - 1);
            cam.eye[0]     = sceneBBox.center().x;
            cam.eye[1]     = sceneBBox.center().y;
            cam.eye[2]     = sceneBBox.max.z + d;
        }
    }
    t.camera             = cam;
    t.technique          = tech;
    if (tech.type == IG_TECHNIQUE_WIREFRAME && cam.type != IG_CAMERA_PERSPECTIVE && cam.type != IG_CAMERA_ORTHOGONAL)
        fail("Technique 'wireframe': the camera differential of this camera type is not supported by the HIP backend");
    if (tech.type == IG_TECHNIQUE_LIGHTTRACER) {
        // the light tracer needs Light::sample_emission and Camera::sample_pixel: lowered for these light types and the pinhole camera
        // (Light::sample_emission is lowered for every light type: lt_core.h)
        if (cam.type != IG_CAMERA_PERSPECTIVE || cam.aperture_radius > 0)
            fail("Technique 'lt': only the perspective camera without depth of field is supported by the HIP backend");
    }
    if (tech.type == IG_TECHNIQUE_PPM) {
        // the light pass needs Light::sample_emission like the light tracer's emitter
        // __tech_radius = radius * SceneDiameter, SceneDiameter = |bbox.max - bbox.min| (PhotonMappingTechnique.cpp:100, LoaderEntity.cpp:187)
        const V3 size = entityCount ? sceneBBox.diameter() : V3(0, 0, 0);
        t.technique.merge_radius = tech.merge_radius * std::sqrt(size.x * size.x + size.y * size.y + size.z * size.z);
    }
    for (int i = 0; i < 3; ++i) {
        t.bbox_min[i] = entityCount ? sceneBBox.min[i] : 0;
        t.bbox_max[i] = entityCount ? sceneBBox.max[i] : 0;
    }
    t.film_width  = width;
    t.film_height = height;
    {
        // bbox_radius(scene_bbox) * 1.01 (src/artic/core/bbox.art:24, light/env.art:88)
        const V3 size  = entityCount ? sceneBBox.diameter() : V3(0, 0, 0);
        t.scene_radius = std::sqrt(size.x * size.x + size.y * size.y + size.z * size.z) / 2 * 1.01f;
    }
    return sc;
}

} // namespace igh

// ---------------------------------------------------------------- C ABI

struct igh_scene {
    std::unique_ptr<igh::Scene> scene;
};

static thread_local std::string g_last_error;

extern "C" {

namespace igh {
// "externals" (src/runtime/loader/Parser.cpp:395-463): other Ignis scene files are loaded first and the including file
// then adds to / replaces what they define — named objects by name, camera / technique / film as a whole. File names of
// an included object stay relative to the file that declared it, so they are made absolute while merging.
static JsonValue mergeExternals(const JsonValue& doc, const std::string& base_dir, int depth)
{
    const JsonValue* exts = doc.find("externals");
    if (!exts)
        return doc;
    if (!exts->isArray())
        fail("Expected 'external' elements to be an array");
    if (depth > 8)
        fail("'externals' are nested too deeply");

    JsonValue merged;
    merged.type = JsonValue::Object;
    auto member = [&](const std::string& key) -> JsonValue& {
        for (auto& p : merged.obj)
            if (p.first == key)
                return p.second;
        merged.obj.emplace_back(key, JsonValue{});
        return merged.obj.back().second;
    };
    auto absorb = [&](const JsonValue& from, const std::string& dir, bool fix_paths) {
        for (const auto& p : from.obj) {
            if (p.first == "externals")
                continue;
            if (!p.second.isArray()) { // camera, technique, film, ...: the later definition wins as a whole
                member(p.first) = p.second;
                continue;
            }
            JsonValue& list = member(p.first);
            list.type       = JsonValue::Array;
            for (JsonValue item : p.second.arr) {
                if (fix_paths && item.isObject())
                    for (auto& q : item.obj)
                        if (q.first == "filename" && q.second.isString() && !q.second.str.empty() && q.second.str[0] != '/')
                            q.second.str = dir + "/" + q.second.str;
                const std::string name = item.getString("name");
                bool replaced          = false;
                if (!name.empty())
                    for (auto& existing : list.arr)
                        if (existing.getString("name") == name) {
                            existing = item;
                            replaced = true;
                            break;
                        }
                if (!replaced)
                    list.arr.push_back(std::move(item));
            }
        }
    };

    for (const auto& e : exts->arr) {
        if (!e.isObject())
            fail("Expected 'external' element to be an object");
        const std::string filename = e.getString("filename");
        if (filename.empty())
            fail("Expected a path for externals");
        const std::string path = (filename[0] == '/' || base_dir.empty()) ? filename : base_dir + "/" + filename;
        std::string type       = e.getString("type");
        const size_t dot       = path.rfind('.');
        std::string ext        = dot == std::string::npos ? "" : path.substr(dot);
        for (auto& c : ext)
            c = (char)std::tolower((unsigned char)c);
        if (type.empty())
            type = ext == ".json" ? "ignis" : "";
        if (type != "ignis")
            fail("External '" + filename + "': only Ignis scene files (.json) can be included by this loader");
        std::ifstream f(path, std::ios::in | std::ios::binary);
        if (!f)
            fail("Could not find path '" + filename + "'");
        std::stringstream ss;
        ss << f.rdbuf();
        const size_t slash    = path.find_last_of('/');
        const std::string dir = slash == std::string::npos ? "." : path.substr(0, slash);
        const JsonValue inner = mergeExternals(JsonParser(ss.str()).parse(), dir, depth + 1);
        absorb(inner, dir, dir != base_dir);
    }
    absorb(doc, base_dir, false);
    return merged;
}
} // namespace igh

igh_scene* igh_load_string(const char* json, const char* base_dir, const igh_options* opts)
{
    g_last_error.clear();
    if (!json) {
        g_last_error = "igh_load_string: json is NULL";
        return nullptr;
    }
    try {
        const std::string text(json);
        igh::JsonParser parser(text);
        const std::string dir    = base_dir ? base_dir : "";
        const igh::JsonValue doc = igh::mergeExternals(parser.parse(), dir, 0);
        auto sc                  = igh::buildScene(doc, dir, opts);
        return new igh_scene{ std::move(sc) };
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return nullptr;
    }
}

igh_scene* igh_load_file(const char* path, const igh_options* opts)
{
    g_last_error.clear();
    if (!path) {
        g_last_error = "igh_load_file: path is NULL";
        return nullptr;
    }
    std::ifstream f(path, std::ios::in | std::ios::binary);
    if (!f) {
        g_last_error = std::string("Could not open file '") + path + "'";
        return nullptr;
    }
    std::stringstream ss;
    ss << f.rdbuf();
    std::string p(path);
    const size_t slash     = p.find_last_of('/');
    const std::string base = slash == std::string::npos ? "." : p.substr(0, slash);
    return igh_load_string(ss.str().c_str(), base.c_str(), opts);
}

const igd_scene* igh_tables(const igh_scene* scene) { return scene ? &scene->scene->tables : nullptr; }

const char* igh_entity_name(const igh_scene* scene, uint32_t id)
{
    if (!scene || id >= scene->scene->entity_names.size())
        return nullptr;
    return scene->scene->entity_names[id].c_str();
}

const char* igh_material_name(const igh_scene* scene, uint32_t id)
{
    if (!scene || id >= scene->scene->material_names.size())
        return nullptr;
    return scene->scene->material_names[id].c_str();
}

void igh_free(igh_scene* scene) { delete scene; }

int32_t igh_save_exr(const char* path, const float* rgb, int32_t width, int32_t height, float scale, const char* const* meta)
{
    g_last_error.clear();
    try {
        if (!path)
            throw std::runtime_error("igh_save_exr: path is NULL");
        std::vector<std::pair<std::string, std::string>> attrs;
        for (const char* const* m = meta; m && m[0] && m[1]; m += 2)
            attrs.emplace_back(m[0], m[1]);
        igh::writeExr(path, rgb, width, height, scale, attrs);
        return 0;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return -1;
    }
}

int32_t igh_read_float_image(const char* path, uint32_t* width, uint32_t* height, uint32_t* channels, float* pixels, uint64_t capacity)
{
    g_last_error.clear();
    try {
        if (!path || !width || !height || !channels)
            throw std::runtime_error("igh_read_float_image: NULL argument");
        const igh::FloatImage img = igh::readFloatImage(path);
        *width = img.width, *height = img.height, *channels = img.channels;
        if (pixels) {
            if (capacity < img.pixels.size())
                throw std::runtime_error("igh_read_float_image: buffer too small");
            std::memcpy(pixels, img.pixels.data(), img.pixels.size() * sizeof(float));
        }
        return 0;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return -1;
    }
}

int32_t igh_read_image8(const char* path, uint32_t* width, uint32_t* height, uint32_t* channels, uint8_t* pixels, uint64_t capacity)
{
    g_last_error.clear();
    try {
        if (!path || !width || !height || !channels)
            throw std::runtime_error("igh_read_image8: NULL argument");
        std::string lower = path;
        for (char& ch : lower)
            ch = (char)std::tolower((unsigned char)ch);
        igh::PngImage img;
        if (lower.size() >= 4 && lower.compare(lower.size() - 4, 4, ".png") == 0) {
            img = igh::readPng(path);
        } else {
            igh::JpegImage j = igh::readJpeg(path);
            img.width = j.width, img.height = j.height, img.channels = j.channels;
            img.data = std::move(j.data);
        }
        *width = img.width, *height = img.height, *channels = img.channels;
        if (pixels) {
            if (capacity < img.data.size())
                throw std::runtime_error("igh_read_image8: buffer too small");
            std::memcpy(pixels, img.data.data(), img.data.size());
        }
        return 0;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return -1;
    }
}

int32_t igh_eval_expression(const char* source, const float* vars, float result[4], int32_t* type, uint32_t* words)
{
    g_last_error.clear();
    try {
        if (!source || !result)
            throw std::runtime_error("igh_eval_expression: NULL argument");
        struct Vars {
            const float* v;
            ige_v4 var(int id) const { return v ? ige_v4{ { v[id * 4], v[id * 4 + 1], v[id * 4 + 2], v[id * 4 + 3] } } : ige_v4{}; }
            ige_v4 tex(uint32_t, float, float) const { return ige_v4{}; }
            ige_v4 evr(ige_v4, ige_v4, ige_v4 n) const { return n; } // (ensure_valid_reflection lives with the shading code)
        };
        igh::pexpr::Env env;
        const igh::pexpr::Program prog = igh::pexpr::compile(source, env);
        const ige_v4 r                 = ige_run(prog.code.data(), Vars{ vars });
        std::memcpy(result, r.v, sizeof(r.v));
        if (type)
            *type = (int32_t)prog.type;
        if (words)
            *words = (uint32_t)prog.code.size();
        return 0;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return -1;
    }
}

double igh_eval_sky(int32_t channel, double turbidity, double albedo, double elevation, double theta, double gamma)
{
    if (channel < 0 || channel > 2 || !(turbidity >= 1 && turbidity <= 10))
        return 0;
    return igh::hosek::radiance(igh::hosek::init(channel, turbidity, albedo, elevation), theta, gamma);
}

int32_t igh_test_collapse_plan(const float* boxes, uint32_t count, float reinsert_ratio, int32_t reinsert_iterations, double out[5])
{
    if (!boxes || !out || count == 0)
        return -1;
    std::vector<igh::BBox> bb(count);
    for (uint32_t i = 0; i < count; ++i) {
        bb[i].min = igh::V3(boxes[6 * i], boxes[6 * i + 1], boxes[6 * i + 2]);
        bb[i].max = igh::V3(boxes[6 * i + 3], boxes[6 * i + 4], boxes[6 * i + 5]);
    }
    igh::collapse_plan_check(bb, reinsert_ratio, reinsert_iterations, out);
    return 0;
}

int32_t igh_test_mesh_edges(int32_t which, float radius, uint32_t subdivisions, uint64_t out[6])
{
    if (!out)
        return -1;
    const igh::TriMesh mesh = which == 0 ? igh::TriMesh::MakeIcoSphere(igh::V3(0, 0, 0), radius, subdivisions)
                                         : igh::TriMesh::MakeTriangle(igh::V3(0, 0, 0), igh::V3(1, 0, 0), igh::V3(0, 1, 0));
    // the directed edges of the faces in face order: edge 3 f + k runs from corner k to corner k + 1 of face f — the half edges of
    // TriMesh::computeHalfEdges (TriMesh.cpp:733-768), whose twin is the edge that runs the other way in another face
    const size_t faces = mesh.faceCount();
    std::map<std::pair<uint32_t, uint32_t>, uint32_t> edge_id;
    for (size_t f = 0; f < faces; ++f)
        for (uint32_t k = 0; k < 3; ++k)
            edge_id[{ mesh.indices[4 * f + k], mesh.indices[4 * f + (k + 1) % 3] }] = (uint32_t)(3 * f + k);
    auto twin_of = [&](uint32_t id) -> int64_t {
        const size_t f = id / 3, k = id % 3;
        const auto it  = edge_id.find({ mesh.indices[4 * f + (k + 1) % 3], mesh.indices[4 * f + k] });
        return it == edge_id.end() ? -1 : (int64_t)it->second;
    };
    uint64_t with_twin = 0, mutual = 0, around_vertex = 0;
    for (uint32_t id = 0; id < 3 * faces; ++id) {
        const int64_t t = twin_of(id);
        if (t < 0)
            continue;
        ++with_twin;
        mutual += twin_of((uint32_t)t) == (int64_t)id ? 1 : 0;
        // "the twin of the previous half edge has the same vertex": previous = the edge into this edge's start corner
        const uint32_t prev = (uint32_t)(3 * (id / 3) + (id % 3 + 2) % 3);
        const int64_t pt    = twin_of(prev);
        if (pt >= 0)
            around_vertex += mesh.indices[4 * ((size_t)pt / 3) + (size_t)pt % 3] == mesh.indices[4 * (id / 3) + id % 3] ? 1 : 0;
    }
    out[0] = faces, out[1] = 3 * faces, out[2] = edge_id.size(), out[3] = with_twin, out[4] = mutual, out[5] = around_vertex;
    return 0;
}

int32_t igh_test_quantise_nodes(float* bounds, const int32_t* child, uint32_t count, int32_t* pad)
{
    if (!bounds || !child || !pad)
        return -1;
    std::vector<ig_node8> nodes(count);
    for (uint32_t n = 0; n < count; ++n) {
        std::memset(&nodes[n], 0, sizeof(ig_node8));
        std::memcpy(nodes[n].bounds, bounds + (size_t)n * 48, sizeof(nodes[n].bounds));
        std::memcpy(nodes[n].child, child + (size_t)n * 8, sizeof(nodes[n].child));
    }
    igh::quantise_node8(nodes.data(), nodes.size());
    for (uint32_t n = 0; n < count; ++n) {
        std::memcpy(bounds + (size_t)n * 48, nodes[n].bounds, sizeof(nodes[n].bounds));
        std::memcpy(pad + (size_t)n * 4, nodes[n].pad, 4 * sizeof(int32_t));
    }
    return 0;
}

const char* igh_last_error(void) { return g_last_error.c_str(); }

} // extern "C"
