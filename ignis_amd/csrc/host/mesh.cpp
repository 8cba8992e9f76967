#include "mesh.h"

#include <zlib.h>

#include <iterator>
#include <map>
#include <sstream>
#include <tuple>

#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace igh {

static constexpr float FltEps = 1.1920928955e-07f;

static inline V3 computeTriangleNormal(V3 v0, V3 v1, V3 v2) { return cross(v1 - v0, v2 - v0); }

static inline bool isApprox(V3 a, V3 b, float prec)
{
    // Eigen isApprox: |a-b|^2 <= prec^2 * min(|a|^2, |b|^2)
    const V3 d = a - b;
    return dot(d, d) <= prec * prec * std::min(dot(a, a), dot(b, b));
}

void TriMesh::flipNormals()
{
    for (size_t i = 0; i < indices.size(); i += 4)
        std::swap(indices[i + 1], indices[i + 2]);
    for (auto& n : normals)
        n = -n;
}

void TriMesh::computeVertexNormals()
{
    normals.assign(vertices.size(), V3(0, 0, 0));
    for (size_t i = 0; i < indices.size(); i += 4) {
        const V3 N = normalized(computeTriangleNormal(vertices[indices[i + 0]], vertices[indices[i + 1]], vertices[indices[i + 2]]));
        normals[indices[i + 0]] = normals[indices[i + 0]] + N;
        normals[indices[i + 1]] = normals[indices[i + 1]] + N;
        normals[indices[i + 2]] = normals[indices[i + 2]] + N;
    }
    for (auto& n : normals)
        n = normalized(n);
}

void TriMesh::makeTexCoordsNormalized()
{
    texcoords.resize(vertices.size());
    const BBox bbox = computeBBox();
    for (size_t i = 0; i < vertices.size(); ++i) {
        const V3 d = bbox.diameter();
        const V3 t = vertices[i] - bbox.min;
        V2 p;
        if (d.x > FltEps)
            p.x = t.x / d.x;
        if (d.y > FltEps)
            p.y = t.y / d.y;
        texcoords[i] = p;
    }
}

void TriMesh::setupFaceNormalsAsVertexNormals()
{
    const size_t fc = faceCount();
    std::vector<V3> nv(fc * 3), nn(fc * 3);
    for (size_t f = 0; f < fc; ++f)
        for (int k = 0; k < 3; ++k)
            nv[3 * f + k] = vertices[indices[4 * f + k]];
    for (size_t f = 0; f < fc; ++f) {
        const V3 N = normalized(computeTriangleNormal(nv[3 * f], nv[3 * f + 1], nv[3 * f + 2]));
        nn[3 * f] = nn[3 * f + 1] = nn[3 * f + 2] = N;
    }
    if (!texcoords.empty()) {
        std::vector<V2> nt(fc * 3);
        for (size_t f = 0; f < fc; ++f)
            for (int k = 0; k < 3; ++k)
                nt[3 * f + k] = texcoords[indices[4 * f + k]];
        texcoords = std::move(nt);
    }
    vertices = std::move(nv);
    normals  = std::move(nn);
    for (uint32_t f = 0; f < (uint32_t)fc; ++f)
        for (uint32_t k = 0; k < 3; ++k)
            indices[4 * f + k] = 3 * f + k;
}

void TriMesh::transform(const Affine& t)
{
    if (t.isIdentity())
        return;
    const M3 normalMat = inverse(transpose(t.L));
    for (auto& v : vertices)
        v = t.point(v);
    for (auto& n : normals)
        n = normalized(normalMat * n);
}

BBox TriMesh::computeBBox() const
{
    BBox b;
    for (const auto& v : vertices)
        b.extend(v);
    return b;
}

float TriMesh::computeArea() const
{
    float area = 0;
    for (size_t f = 0; f < faceCount(); ++f)
        area += 0.5f * norm(computeTriangleNormal(vertices[indices[4 * f]], vertices[indices[4 * f + 1]], vertices[indices[4 * f + 2]]));
    return area;
}

std::optional<PlaneShape> TriMesh::getAsPlane() const
{
    constexpr float PlaneEPS = 1e-5f;
    if (faceCount() != 2)
        return std::nullopt;

    std::array<V3, 4> unique_verts;
    std::array<uint32_t, 4> unique_ids{};
    if (vertices.size() != 4) {
        // The reference also accepts 5-6 vertices with duplicates; its dedup loop reads
        // uninitialised slots (TriMesh.cpp:534-557), so only the exact 4-vertex case is kept.
        return std::nullopt;
    }
    for (size_t i = 0; i < 4; ++i) {
        unique_verts[i] = vertices[i];
        unique_ids[i]   = (uint32_t)i;
    }

    const V3 fn0 = normalized(computeTriangleNormal(vertices[indices[0]], vertices[indices[1]], vertices[indices[2]]));
    const V3 fn1 = normalized(computeTriangleNormal(vertices[indices[4]], vertices[indices[5]], vertices[indices[6]]));
    if (!isApprox(fn0, fn1, PlaneEPS))
        return std::nullopt;

    auto sq = [&](uint32_t a, uint32_t b) {
        const V3 d = vertices[indices[a]] - vertices[indices[b]];
        return dot(d, d);
    };
    const float e1 = sq(0, 1), e2 = sq(1, 2), e3 = sq(2, 0);
    const float e4 = sq(4, 5), e5 = sq(5, 6), e6 = sq(6, 4);
    const auto safeCheck = [=](float a, float b) { return std::abs(a - b) <= PlaneEPS; };
    if (!safeCheck(e1, e4) && !safeCheck(e2, e4) && !safeCheck(e3, e4))
        return std::nullopt;
    if (!safeCheck(e1, e5) && !safeCheck(e2, e5) && !safeCheck(e3, e5))
        return std::nullopt;
    if (!safeCheck(e1, e6) && !safeCheck(e2, e6) && !safeCheck(e3, e6))
        return std::nullopt;

    const V3 origin   = unique_verts[0];
    auto computeAngle = [&](size_t start) {
        const V3 x = normalized(unique_verts[(start + 0) % 3 + 1] - origin);
        const V3 y = normalized(unique_verts[(start + 1) % 3 + 1] - origin);
        return std::acos(dot(x, y));
    };
    const float a12 = std::abs(computeAngle(0));
    const float a23 = std::abs(computeAngle(1));
    const float a31 = std::abs(computeAngle(2));
    int sel         = 2;
    if (a12 >= a23 && a12 >= a31)
        sel = 0;
    else if (a23 >= a31 && a23 >= a12)
        sel = 1;

    PlaneShape shape;
    shape.origin = origin;
    shape.x_axis = unique_verts[(sel + 0) % 3 + 1] - origin;
    shape.y_axis = unique_verts[(sel + 1) % 3 + 1] - origin;

    const V3 normal = normalized(cross(shape.x_axis, shape.y_axis));
    if (dot(fn0, normal) < 0) {
        std::swap(shape.x_axis, shape.y_axis);
        std::swap(unique_verts[1], unique_verts[2]);
        std::swap(unique_ids[1], unique_ids[2]);
    }

    if (!texcoords.empty()) {
        shape.texcoords[0]                 = texcoords[unique_ids[0]];
        shape.texcoords[(0 + sel) % 3 + 1] = texcoords[unique_ids[1]];
        shape.texcoords[(1 + sel) % 3 + 1] = texcoords[unique_ids[2]];
        shape.texcoords[(2 + sel) % 3 + 1] = texcoords[unique_ids[3]];
    } else {
        shape.texcoords[0] = V2{ 0, 0 };
        shape.texcoords[1] = V2{ 1, 0 };
        shape.texcoords[2] = V2{ 0, 1 };
        shape.texcoords[3] = V2{ 1, 1 };
    }
    return shape;
}

static void addTriangle(TriMesh& mesh, V3 origin, V3 xAxis, V3 yAxis)
{
    const V3 N         = normalized(cross(xAxis, yAxis));
    const uint32_t off = (uint32_t)mesh.vertices.size();
    mesh.vertices.insert(mesh.vertices.end(), { origin, origin + xAxis, origin + yAxis });
    mesh.normals.insert(mesh.normals.end(), { N, N, N });
    mesh.texcoords.insert(mesh.texcoords.end(), { V2{ 0, 0 }, V2{ 1, 0 }, V2{ 0, 1 } });
    mesh.indices.insert(mesh.indices.end(), { 0 + off, 1 + off, 2 + off, 0 });
}

// addGrid with count 1x1 (TriMesh.cpp:783-817)
static void addPlane(TriMesh& mesh, V3 origin, V3 xAxis, V3 yAxis)
{
    const V3 N         = normalized(cross(xAxis, yAxis));
    const uint32_t off = (uint32_t)mesh.vertices.size();
    for (uint32_t j = 0; j <= 1; ++j) {
        for (uint32_t i = 0; i <= 1; ++i) {
            const float u = (float)i, v = (float)j;
            mesh.vertices.push_back(origin + xAxis * u + yAxis * v);
            mesh.normals.push_back(N);
            mesh.texcoords.push_back(V2{ u, v });
        }
    }
    const uint32_t ind1 = off, ind2 = 2 + off;
    mesh.indices.insert(mesh.indices.end(), { ind1, ind1 + 1, ind2 + 1, 0, ind1, ind2 + 1, ind2, 0 });
}

TriMesh TriMesh::MakePlane(V3 origin, V3 x_axis, V3 y_axis)
{
    TriMesh m;
    addPlane(m, origin, x_axis, y_axis);
    return m;
}

TriMesh TriMesh::MakeTriangle(V3 p0, V3 p1, V3 p2)
{
    TriMesh m;
    addTriangle(m, p0, p1 - p0, p2 - p0);
    return m;
}

TriMesh TriMesh::MakeRectangle(V3 p0, V3 p1, V3 p2, V3 p3)
{
    TriMesh m;
    addTriangle(m, p0, p1 - p0, p3 - p0);
    addTriangle(m, p1, p2 - p1, p3 - p1);
    return m;
}

TriMesh TriMesh::MakeBox(V3 origin, V3 xAxis, V3 yAxis, V3 zAxis)
{
    const V3 lll = origin;
    const V3 hhh = origin + xAxis + yAxis + zAxis;
    TriMesh m;
    addPlane(m, lll, yAxis, xAxis);
    addPlane(m, lll, xAxis, zAxis);
    addPlane(m, lll, zAxis, yAxis);
    addPlane(m, hhh, -xAxis, -yAxis);
    addPlane(m, hhh, -zAxis, -xAxis);
    addPlane(m, hhh, -yAxis, -zAxis);
    return m;
}

// ---------------------------------------------------------------- procedural shapes

namespace {
constexpr float kPiF = 3.14159265358979323846f;

// Tangent::frame (src/runtime/math/Tangent.h:50-73): the branch-free basis of Duff et al., normalised
void tangentFrame(V3 n, V3& tx, V3& ty)
{
    const float sg = std::copysign(1.0f, n.z);
    const float a  = -1.0f / (sg + n.z);
    const float b  = n.x * n.y * a;
    tx             = normalized(V3(1.0f + sg * n.x * n.x * a, sg * b, -sg * n.x));
    ty             = normalized(V3(b, sg + n.y * n.y * a, -n.y));
}

void pushTri(TriMesh& m, uint32_t a, uint32_t b, uint32_t c) { m.indices.insert(m.indices.end(), { a, b, c, 0u }); }

// A ring of `count` vertices around `center` in the (tx, ty) plane, optionally preceded by the centre vertex and closed
// by a triangle fan (TriMesh.cpp:819-852). Returns the index of the first ring vertex.
uint32_t appendRing(TriMesh& m, V3 center, V3 n, V3 tx, V3 ty, float radius, uint32_t count, bool with_fan, bool reversed)
{
    const uint32_t hub = (uint32_t)m.vertices.size();
    if (with_fan) {
        m.vertices.push_back(center);
        m.normals.push_back(n);
        m.texcoords.push_back(V2{ 0, 0 });
    }
    const uint32_t first = (uint32_t)m.vertices.size();
    const float step     = 1.0f / count;
    for (uint32_t k = 0; k < count; ++k) {
        const float c = std::cos(2 * kPiF * step * k);
        const float s = std::sin(2 * kPiF * step * k);
        m.vertices.push_back(tx * radius * c + ty * radius * s + center);
        m.normals.push_back(n);
        m.texcoords.push_back(V2{ 0.5f * (c + 1), 0.5f * (s + 1) });
    }
    if (with_fan)
        for (uint32_t k = 0; k < count; ++k) {
            const uint32_t cur = first + k, nxt = first + (k + 1) % count;
            if (reversed)
                pushTri(m, hub, nxt, cur);
            else
                pushTri(m, hub, cur, nxt);
        }
    return first;
}
} // namespace

TriMesh TriMesh::MakeDisk(V3 center, V3 normal, float radius, uint32_t sections)
{
    TriMesh m;
    V3 tx, ty;
    tangentFrame(normal, tx, ty);
    appendRing(m, center, normal, tx, ty, radius, std::max(3u, sections), true, false);
    return m;
}

TriMesh TriMesh::MakeCone(V3 base_center, float base_radius, V3 tip, uint32_t sections, bool fill_cap)
{
    sections      = std::max(3u, sections);
    const V3 axis = normalized(base_center - tip);
    V3 tx, ty;
    tangentFrame(axis, tx, ty);
    TriMesh m;
    const uint32_t ring = appendRing(m, base_center, axis, tx, ty, base_radius, sections, fill_cap, false);
    const uint32_t apex = (uint32_t)m.vertices.size();
    m.vertices.push_back(tip);
    m.normals.push_back(axis);
    m.texcoords.push_back(V2{ 0, 0 });
    for (uint32_t k = 0; k < sections; ++k)
        pushTri(m, ring + k, apex, ring + (k + 1) % sections);
    m.computeVertexNormals(); // cap and mantle share the ring vertices (TriMesh.cpp:1101)
    return m;
}

TriMesh TriMesh::MakeCylinder(V3 base_center, float base_radius, V3 top_center, float top_radius, uint32_t sections, bool fill_cap)
{
    sections      = std::max(3u, sections);
    const V3 axis = normalized(base_center - top_center);
    V3 tx, ty;
    tangentFrame(axis, tx, ty);
    TriMesh m;
    const uint32_t lo = appendRing(m, base_center, axis, tx, ty, base_radius, sections, fill_cap, false);
    const uint32_t hi = appendRing(m, top_center, axis, tx, ty, top_radius, sections, fill_cap, true);
    for (uint32_t k = 0; k < sections; ++k) {
        const uint32_t k1 = (k + 1) % sections;
        pushTri(m, lo + k, hi + k, lo + k1);
        pushTri(m, hi + k, hi + k1, lo + k1);
    }
    m.computeVertexNormals();
    return m;
}

// Latitude / longitude sphere: (stacks + 1) rows of `slices` vertices, poles duplicated per slice (TriMesh.cpp:854-907)
TriMesh TriMesh::MakeUVSphere(V3 center, float radius, uint32_t stacks, uint32_t slices)
{
    stacks = std::max(2u, stacks);
    slices = std::max(2u, slices);
    TriMesh m;
    const float drho = kPiF / (float)stacks, dtheta = 2 * kPiF / (float)slices;
    for (uint32_t row = 0; row <= stacks; ++row) {
        const float rho = (float)row * drho;
        const float sr = std::sin(rho), cr = std::cos(rho);
        for (uint32_t col = 0; col < slices; ++col) {
            const float theta = col * dtheta;
            const V3 n(-std::sin(theta) * sr, std::cos(theta) * sr, cr);
            m.vertices.push_back(n * radius + center);
            m.normals.push_back(n);
            m.texcoords.push_back(V2{ (float)(0.5 * theta / kPiF), rho / kPiF });
        }
    }
    for (uint32_t row = 0; row < stacks; ++row) {
        const uint32_t up = row * slices, dn = (row + 1) * slices;
        for (uint32_t col = 0; col < slices; ++col) {
            const uint32_t nc = (col + 1) % slices;
            pushTri(m, dn + col, dn + nc, up + nc);
            pushTri(m, dn + col, up + nc, up + col);
        }
    }
    return m;
}

// Icosahedron from three golden rectangles, refined by splitting every edge at its spherical midpoint
// (TriMesh.cpp:910-1025). Vertex numbering and face order follow the reference so the tessellation is the same.
TriMesh TriMesh::MakeIcoSphere(V3 center, float radius, uint32_t subdivisions)
{
    TriMesh m;
    const float phi = 1.618033989f;
    // vertex (axis d, sign a, sign b) has phi * a on axis d + 1 and b on axis d + 2
    auto corner = [](int d, int a, int b) { return (uint32_t)(d * 4 + (a + 1) + ((b + 1) >> 1)); };
    for (int d = 0; d < 3; ++d)
        for (int a = -1; a <= 1; a += 2)
            for (int b = -1; b <= 1; b += 2) {
                float v[3] = { 0, 0, 0 };
                v[(d + 1) % 3] = phi * a;
                v[(d + 2) % 3] = 1.0f * b;
                m.vertices.push_back(normalized(V3(v[0], v[1], v[2])));
            }
    // 8 faces with one corner on each rectangle, then 12 faces with an edge on one rectangle
    for (int a = -1; a <= 1; a += 2)
        for (int b = -1; b <= 1; b += 2)
            for (int c = -1; c <= 1; c += 2) {
                const uint32_t i1 = corner(0, a, b), i2 = corner(1, b, c), i3 = corner(2, c, a);
                if (a * b * c == -1)
                    pushTri(m, i1, i3, i2);
                else
                    pushTri(m, i1, i2, i3);
            }
    for (int d = 0; d < 3; ++d)
        for (int a = -1; a <= 1; a += 2)
            for (int b = -1; b <= 1; b += 2) {
                const uint32_t i1 = corner(d, a, +1), i2 = corner(d, a, -1), i3 = corner((d + 2) % 3, b, a);
                if (a * b == 1)
                    pushTri(m, i1, i3, i2);
                else
                    pushTri(m, i1, i2, i3);
            }
    for (uint32_t level = 0; level < subdivisions; ++level) {
        std::map<std::pair<uint32_t, uint32_t>, uint32_t> midpoint;
        const size_t faces = m.indices.size() / 4;
        // midpoints are numbered in the order the directed edges (low -> high index) are met
        for (size_t f = 0; f < faces; ++f)
            for (int e = 0; e < 3; ++e) {
                const uint32_t p = m.indices[4 * f + e], q = m.indices[4 * f + (e + 1) % 3];
                if (p >= q)
                    continue;
                midpoint[{ p, q }] = (uint32_t)m.vertices.size();
                m.vertices.push_back(normalized(m.vertices[p] + m.vertices[q]));
            }
        std::vector<uint32_t> finer;
        finer.reserve(faces * 16);
        for (size_t f = 0; f < faces; ++f) {
            uint32_t corner_id[3], mid[3];
            for (int e = 0; e < 3; ++e) {
                corner_id[e]     = m.indices[4 * f + e];
                const uint32_t q = m.indices[4 * f + (e + 1) % 3];
                mid[e]           = midpoint.at({ std::min(corner_id[e], q), std::max(corner_id[e], q) });
            }
            finer.insert(finer.end(), { mid[0], mid[1], mid[2], 0u });
            for (int e = 0; e < 3; ++e)
                finer.insert(finer.end(), { corner_id[e], mid[e], mid[(e + 2) % 3], 0u });
        }
        m.indices.swap(finer);
    }
    m.normals.resize(m.vertices.size());
    m.texcoords.resize(m.vertices.size());
    for (size_t i = 0; i < m.vertices.size(); ++i) {
        const V3 n   = normalized(m.vertices[i]);
        m.normals[i] = n;
        float lon    = std::atan2(-n.x, n.y);
        if (lon < 0)
            lon += 2 * kPiF;
        m.texcoords[i] = V2{ lon / (2 * kPiF), std::acos(n.z) / kPiF };
    }
    // translate(center) * scale(radius); normals are unaffected by a uniform scale
    for (auto& v : m.vertices)
        v = v * radius + center;
    return m;
}

// ---------------------------------------------------------------- PLY

namespace {
struct PlyHeader {
    int VertexCount = 0, FaceCount = 0;
    int XElem = -1, YElem = -1, ZElem = -1, NXElem = -1, NYElem = -1, NZElem = -1, UElem = -1, VElem = -1;
    int VertexPropCount = 0;
    int IndElem         = -1;
    bool SwitchEndianness = false;
    bool idxIsByteCount   = true;
    bool hasVertices() const { return XElem >= 0 && YElem >= 0 && ZElem >= 0; }
    bool hasNormals() const { return NXElem >= 0 && NYElem >= 0 && NZElem >= 0; }
    bool hasUVs() const { return UElem >= 0 && VElem >= 0; }
};

template <typename T>
T swap_endian(T u)
{
    unsigned char b[sizeof(T)];
    std::memcpy(b, &u, sizeof(T));
    for (size_t k = 0; k < sizeof(T) / 2; ++k)
        std::swap(b[k], b[sizeof(T) - k - 1]);
    std::memcpy(&u, b, sizeof(T));
    return u;
}

// Faces with 3 vertices are kept, 4 become a fan; larger polygons use the
// reference's convex-fan fallback (PlyFile.cpp:30-66; its ear-clipping path is
// only reached by meshes outside the hot-path configs).
void triangulate(const std::vector<uint32_t>& g, std::vector<uint32_t>& out)
{
    if (g.size() < 3)
        return;
    for (uint32_t j = 2; j < (uint32_t)g.size(); ++j)
        out.insert(out.end(), { g[0], g[j - 1], g[j], 0 });
}
} // namespace

TriMesh load_ply(const std::string& path)
{
    std::ifstream stream(path, std::ios::in | std::ios::binary);
    if (!stream)
        throw std::runtime_error("PLY file '" + path + "' can not be opened");

    std::string magic;
    stream >> magic;
    if (magic != "ply")
        throw std::runtime_error("'" + path + "' is not a ply file");

    std::string method;
    PlyHeader header;
    int facePropCounter = 0;
    for (std::string line; std::getline(stream, line);) {
        std::stringstream ss(line);
        std::string action;
        ss >> action;
        if (action == "comment")
            continue;
        else if (action == "format")
            ss >> method;
        else if (action == "element") {
            std::string type;
            ss >> type;
            if (type == "vertex")
                ss >> header.VertexCount;
            else if (type == "face")
                ss >> header.FaceCount;
        } else if (action == "property") {
            std::string type;
            ss >> type;
            if (type == "float") {
                std::string name;
                ss >> name;
                const int c = header.VertexPropCount;
                if (name == "x") header.XElem = c;
                else if (name == "y") header.YElem = c;
                else if (name == "z") header.ZElem = c;
                else if (name == "nx") header.NXElem = c;
                else if (name == "ny") header.NYElem = c;
                else if (name == "nz") header.NZElem = c;
                else if (name == "u" || name == "s") header.UElem = c;
                else if (name == "v" || name == "t") header.VElem = c;
                ++header.VertexPropCount;
            } else if (type == "list") {
                ++facePropCounter;
                std::string countType, indType, name;
                ss >> countType >> indType >> name;
                if (name == "vertex_indices" || name == "vertex_index")
                    header.IndElem = facePropCounter - 1;
            } else {
                ++header.VertexPropCount;
            }
        } else if (action == "end_header")
            break;
    }

    if (!header.hasVertices() || header.IndElem < 0 || header.VertexCount <= 0 || header.FaceCount <= 0)
        throw std::runtime_error("PLY file '" + path + "' does not contain valid mesh data");

    header.SwitchEndianness = (method == "binary_big_endian");
    const bool ascii        = (method == "ascii");

    auto readFloat = [&]() {
        float v = 0;
        stream.read(reinterpret_cast<char*>(&v), sizeof(v));
        return header.SwitchEndianness ? swap_endian(v) : v;
    };
    auto readIdx = [&]() {
        uint32_t v = 0;
        stream.read(reinterpret_cast<char*>(&v), sizeof(v));
        return header.SwitchEndianness ? swap_endian(v) : v;
    };

    TriMesh mesh;
    mesh.vertices.reserve(header.VertexCount);
    for (int i = 0; i < header.VertexCount; ++i) {
        float vals[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }; // x y z nx ny nz u v
        auto assign   = [&](int elem, float val) {
            if (header.XElem == elem) vals[0] = val;
            else if (header.YElem == elem) vals[1] = val;
            else if (header.ZElem == elem) vals[2] = val;
            else if (header.NXElem == elem) vals[3] = val;
            else if (header.NYElem == elem) vals[4] = val;
            else if (header.NZElem == elem) vals[5] = val;
            else if (header.UElem == elem) vals[6] = val;
            else if (header.VElem == elem) vals[7] = val;
        };
        if (ascii) {
            std::string line;
            if (!std::getline(stream, line))
                throw std::runtime_error("PLY file '" + path + "': not enough vertices");
            std::stringstream ss(line);
            int elem = 0;
            float val;
            while (ss >> val)
                assign(elem++, val);
        } else {
            for (int elem = 0; elem < header.VertexPropCount; ++elem)
                assign(elem, readFloat());
        }
        mesh.vertices.emplace_back(vals[0], vals[1], vals[2]);
        if (header.hasNormals()) {
            float n = std::sqrt(vals[3] * vals[3] + vals[4] * vals[4] + vals[5] * vals[5]);
            if (n == 0.0f)
                n = 1.0f;
            mesh.normals.emplace_back(vals[3] / n, vals[4] / n, vals[5] / n);
        }
        if (header.hasUVs())
            mesh.texcoords.push_back(V2{ vals[6], vals[7] });
    }
    if (!stream && !ascii)
        throw std::runtime_error("PLY file '" + path + "': truncated vertex data");

    mesh.indices.reserve((size_t)header.FaceCount * 4);
    std::vector<uint32_t> tmp;
    for (int i = 0; i < header.FaceCount; ++i) {
        tmp.clear();
        if (ascii) {
            std::string line;
            if (!std::getline(stream, line))
                throw std::runtime_error("PLY file '" + path + "': not enough faces");
            std::stringstream ss(line);
            uint32_t elems = 0;
            ss >> elems;
            for (uint32_t e = 0; e < elems; ++e) {
                uint32_t idx = 0;
                ss >> idx;
                tmp.push_back(idx);
            }
        } else {
            uint8_t elems = 0;
            stream.read(reinterpret_cast<char*>(&elems), sizeof(elems));
            for (uint32_t e = 0; e < elems; ++e)
                tmp.push_back(readIdx());
            if (!stream)
                throw std::runtime_error("PLY file '" + path + "': truncated face data");
        }
        for (uint32_t idx : tmp)
            if (idx >= mesh.vertices.size())
                throw std::runtime_error("PLY file '" + path + "': face index out of range");
        triangulate(tmp, mesh.indices);
    }

    if (mesh.normals.empty()) {
        mesh.computeVertexNormals();
    } else {
        // fixNormals, TriMesh.cpp:17-32
        for (auto& n : mesh.normals) {
            const float len2 = dot(n, n);
            if (len2 <= FltEps || std::isnan(len2))
                n = V3(0, 1, 0);
            else
                { const float l = std::sqrt(len2); n = V3(n.x / l, n.y / l, n.z / l); }
        }
    }
    if (mesh.texcoords.empty())
        mesh.makeTexCoordsNormalized();
    return mesh;
}

// Wavefront OBJ (src/runtime/mesh/ObjFile.cpp:25-213 over tinyobjloader): v / vt / vn / f records, 1-based and
// negative (relative) indices, `v`, `v/vt`, `v//vn`, `v/vt/vn` corners. A mesh vertex is created per distinct
// (v, vn, vt) triple in first-use order (ObjFile.cpp:125-139); polygons are split as a fan (tinyobjloader does the
// same for quads and ear-clips larger polygons; the fan equals it for convex faces). Groups / objects / materials
// are merged into one mesh like the reference's default (no shape_index).
TriMesh load_obj(const std::string& path)
{
    std::ifstream stream(path);
    if (!stream)
        throw std::runtime_error("OBJ file '" + path + "' can not be opened");
    std::vector<V3> pos, nrm;
    std::vector<V2> tex;
    struct Corner {
        int v, n, t;
    };
    std::vector<std::vector<Corner>> faces;
    std::string line;
    while (std::getline(stream, line)) {
        std::istringstream ls(line);
        std::string tag;
        if (!(ls >> tag) || tag[0] == '#')
            continue;
        if (tag == "v") {
            float x = 0, y = 0, z = 0;
            ls >> x >> y >> z;
            pos.emplace_back(x, y, z);
        } else if (tag == "vn") {
            float x = 0, y = 0, z = 0;
            ls >> x >> y >> z;
            nrm.emplace_back(x, y, z);
        } else if (tag == "vt") {
            float u = 0, v = 0;
            ls >> u >> v;
            tex.push_back(V2{ u, v });
        } else if (tag == "f") {
            std::vector<Corner> f;
            std::string c;
            while (ls >> c) {
                int idx[3] = { 0, 0, 0 }; // v, vt, vn as written (0 = absent)
                int k = 0;
                size_t start = 0;
                while (k < 3 && start <= c.size()) {
                    const size_t slash = c.find('/', start);
                    const std::string part = c.substr(start, slash == std::string::npos ? std::string::npos : slash - start);
                    if (!part.empty())
                        idx[k] = std::atoi(part.c_str());
                    ++k;
                    if (slash == std::string::npos)
                        break;
                    start = slash + 1;
                }
                auto fix = [&](int i, size_t count) -> int { // 1-based or negative-relative -> 0-based, -1 = absent
                    if (i > 0)
                        return i - 1;
                    if (i < 0)
                        return (int)count + i;
                    return -1;
                };
                Corner cr{ fix(idx[0], pos.size()), fix(idx[2], nrm.size()), fix(idx[1], tex.size()) };
                if (cr.v < 0 || cr.v >= (int)pos.size())
                    throw std::runtime_error("OBJ file '" + path + "': face references a missing vertex");
                if (cr.n >= (int)nrm.size() || cr.t >= (int)tex.size())
                    throw std::runtime_error("OBJ file '" + path + "': face references a missing normal / texcoord");
                f.push_back(cr);
            }
            if (f.size() >= 3)
                faces.push_back(std::move(f));
        }
    }
    if (pos.empty())
        throw std::runtime_error("OBJ file '" + path + "': No vertices given!");

    bool has_norms = false, has_tex = false; // ObjFile.cpp:62-106: used by at least one corner
    for (const auto& f : faces)
        for (const auto& c : f) {
            has_norms |= c.n >= 0;
            has_tex |= c.t >= 0;
        }

    TriMesh mesh;
    std::map<std::tuple<int, int, int>, uint32_t> index_map;
    auto embed = [&](const Corner& c) -> uint32_t {
        const auto key = std::make_tuple(c.v, c.n, c.t);
        const auto it  = index_map.find(key);
        if (it != index_map.end())
            return it->second;
        const uint32_t id = (uint32_t)index_map.size();
        index_map[key]    = id;
        mesh.vertices.push_back(pos[c.v]);
        if (has_norms)
            mesh.normals.push_back(c.n >= 0 ? nrm[c.n] : V3(0, 0, 1));
        if (has_tex)
            mesh.texcoords.push_back(c.t >= 0 ? tex[c.t] : V2{ 0.0f, 0.0f });
        return id;
    };
    for (const auto& f : faces)
        for (size_t k = 1; k + 1 < f.size(); ++k) {
            mesh.indices.push_back(embed(f[0]));
            mesh.indices.push_back(embed(f[k]));
            mesh.indices.push_back(embed(f[k + 1]));
            mesh.indices.push_back(0);
        }
    if (!has_norms)
        mesh.computeVertexNormals();
    if (!has_tex)
        mesh.makeTexCoordsNormalized();
    return mesh;
}

// Mitsuba "serialized" meshes (.serialized / .mts; src/runtime/mesh/MtsSerializedFile.cpp:167-317): a file holds several
// sub-meshes, each {u16 0x041C, u16 version (3 or 4)} followed by one zlib stream {u32 flags, [v4: UTF-8 name, 0],
// u64 vertices, u64 triangles, positions, [normals], [texture coordinates], [colours], indices}; the file ends with the
// offsets of the sub-meshes (u64 each in v4, u32 in v3) and their u32 count.
TriMesh load_serialized(const std::string& path, size_t shape_index)
{
    auto bad = [&](const std::string& why) { return std::runtime_error("Serialized mesh '" + path + "': " + why); };
    std::ifstream in(path, std::ios::in | std::ios::binary);
    if (!in)
        throw bad("cannot open file");
    const std::vector<uint8_t> file((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    auto rd = [&](size_t off, void* dst, size_t n) {
        if (off + n > file.size())
            throw bad("truncated file");
        std::memcpy(dst, &file[off], n);
    };
    uint16_t ident = 0, version = 0;
    rd(0, &ident, 2);
    rd(2, &version, 2);
    if (ident != 0x041C)
        throw bad("not a Mitsuba serialized file");
    if (version < 3)
        throw bad("insufficient version number " + std::to_string(version) + " < 3");
    uint32_t count = 0;
    rd(file.size() - 4, &count, 4);
    if (shape_index >= count)
        throw bad("shape index " + std::to_string(shape_index) + " out of range (" + std::to_string(count) + " shapes)");
    const size_t entry = version >= 4 ? 8 : 4;
    auto offsetOf      = [&](size_t i) {
        uint64_t v = 0;
        rd(file.size() - 4 - entry * (count - i), &v, entry);
        return (size_t)v;
    };
    const size_t begin = offsetOf(shape_index);
    const size_t end   = shape_index + 1 == count ? file.size() - 4 - entry * count : offsetOf(shape_index + 1);
    if (begin + 4 > end || end > file.size())
        throw bad("corrupt shape dictionary");

    // inflate the sub-mesh in one go, growing the output as needed
    std::vector<uint8_t> raw((end - begin) * 4 + 1024);
    {
        z_stream zs;
        std::memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, 15) != Z_OK)
            throw bad("zlib initialisation failed");
        zs.next_in  = const_cast<Bytef*>(&file[begin + 4]);
        zs.avail_in = (uInt)(end - begin - 4);
        size_t have = 0;
        int rc      = Z_OK;
        while (rc != Z_STREAM_END) {
            if (have == raw.size())
                raw.resize(raw.size() * 2);
            const size_t chunk = std::min<size_t>(raw.size() - have, 1u << 30);
            zs.next_out        = &raw[have];
            zs.avail_out       = (uInt)chunk;
            rc                 = inflate(&zs, Z_NO_FLUSH);
            have += chunk - zs.avail_out;
            if (rc != Z_OK && rc != Z_STREAM_END) {
                inflateEnd(&zs);
                throw bad("corrupt compressed data");
            }
            if (rc == Z_OK && zs.avail_in == 0 && zs.avail_out != 0)
                break; // input exhausted without an end marker: use what there is
        }
        inflateEnd(&zs);
        raw.resize(have);
    }
    size_t pos = 0;
    auto take  = [&](void* dst, size_t n) {
        if (pos + n > raw.size())
            throw bad("attempting to read past the end of the stream");
        std::memcpy(dst, &raw[pos], n);
        pos += n;
    };
    enum : uint32_t { HasNormals = 0x0001, HasTexCoords = 0x0002, HasColors = 0x0008, Double = 0x2000 };
    uint32_t flags = 0;
    take(&flags, 4);
    if (version >= 4) {
        uint8_t ch = 1;
        while (ch != 0)
            take(&ch, 1); // shape name
    }
    uint64_t n_vertices = 0, n_triangles = 0;
    take(&n_vertices, 8);
    take(&n_triangles, 8);
    if (n_vertices == 0 || n_triangles == 0)
        throw bad("has no valid mesh");
    auto number = [&]() {
        if (flags & Double) {
            double d;
            take(&d, 8);
            return (float)d;
        }
        float f;
        take(&f, 4);
        return f;
    };
    TriMesh mesh;
    mesh.vertices.resize(n_vertices);
    for (auto& v : mesh.vertices) {
        const float x = number(), y = number(), z = number();
        v             = V3(x, y, z);
    }
    if (flags & HasNormals) {
        mesh.normals.resize(n_vertices);
        for (auto& n : mesh.normals) {
            const float x = number(), y = number(), z = number();
            n             = V3(x, y, z);
        }
    }
    if (flags & HasTexCoords) {
        mesh.texcoords.resize(n_vertices);
        for (auto& t : mesh.texcoords) {
            const float u = number(), v = number();
            t             = V2{ u, v };
        }
    }
    if (flags & HasColors)
        for (uint64_t i = 0; i < n_vertices * 3; ++i)
            (void)number();
    mesh.indices.resize(n_triangles * 4);
    for (uint64_t f = 0; f < n_triangles; ++f) {
        for (int k = 0; k < 3; ++k) {
            uint64_t id = 0;
            take(&id, n_vertices > 0xFFFFFFFFull ? 8 : 4);
            if (id >= n_vertices)
                throw bad("vertex index out of range");
            mesh.indices[f * 4 + k] = (uint32_t)id;
        }
        mesh.indices[f * 4 + 3] = 0;
    }
    if (!(flags & HasNormals)) {
        mesh.computeVertexNormals();
    } else {
        // fixNormals (TriMesh.cpp:17-32)
        for (auto& n : mesh.normals) {
            const float len2 = dot(n, n);
            if (len2 <= FltEps || std::isnan(len2))
                n = V3(0, 1, 0);
            else
                n = n * (1 / std::sqrt(len2));
        }
    }
    if (!(flags & HasTexCoords))
        mesh.makeTexCoordsNormalized();
    return mesh;
}

} // namespace igh
