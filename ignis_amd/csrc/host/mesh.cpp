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

// Is this mesh one planar quad (two congruent triangles over four vertices)? Then area lights on it use the analytic plane
// sampler (src/runtime/light/AreaLight.cpp:50-72). The decision and the frame must be the reference's
// (src/runtime/mesh/TriMesh.cpp:521-620) — which corner becomes the origin and which edges the axes decides what the light's
// random numbers mean — but the code is this file's own:
//   1. exactly two faces over exactly four vertices. (The reference has a second branch for five or six vertices with
//      duplicates; its counter can reach at most three, so that branch rejects every mesh, and so does this.)
//   2. both triangles face the same way: unit normals equal in Eigen's isApprox sense, |a - b|^2 <= eps^2 min(|a|^2, |b|^2)
//   3. congruent triangles: every squared edge length of the second one is among the first one's (+- eps)
//   4. frame: origin = vertex 0; of its three neighbours the two that enclose the widest angle are the quad's edges, the third
//      is the opposite corner. Widest angle = smallest cosine (acos is monotonic), ties resolved in the reference's order of
//      pairs (1,2), (2,3), (3,1). x / y follow that cyclic order, swapped if their cross product opposes the face normal.
//   5. texture coordinates of the corners in the order origin, x end, y end, opposite — or the unit square.
namespace {
constexpr float kPlaneEps = 1e-5f;

bool sameDirection(V3 a, V3 b)
{
    const V3 d = a - b;
    return dot(d, d) <= kPlaneEps * kPlaneEps * std::min(dot(a, a), dot(b, b));
}

// squared edge lengths of triangle `t` of a face-index list with four entries per face
std::array<float, 3> edgeLengths2(const TriMesh& m, size_t t)
{
    std::array<float, 3> e{};
    for (int k = 0; k < 3; ++k) {
        const V3 d = m.vertices[m.indices[4 * t + k]] - m.vertices[m.indices[4 * t + (k + 1) % 3]];
        e[k]       = dot(d, d);
    }
    return e;
}
} // namespace

std::optional<PlaneShape> TriMesh::getAsPlane() const
{
    if (faceCount() != 2 || vertices.size() != 4)
        return std::nullopt;

    V3 face_normal[2];
    for (size_t t = 0; t < 2; ++t)
        face_normal[t] = normalized(computeTriangleNormal(vertices[indices[4 * t]], vertices[indices[4 * t + 1]], vertices[indices[4 * t + 2]]));
    if (!sameDirection(face_normal[0], face_normal[1]))
        return std::nullopt;

    const auto first = edgeLengths2(*this, 0), second = edgeLengths2(*this, 1);
    for (float e : second) {
        bool found = false;
        for (float f : first)
            found |= std::abs(f - e) <= kPlaneEps;
        if (!found)
            return std::nullopt;
    }

    // neighbours of the origin as unit directions; cosine between the cyclic pairs (1,2), (2,3), (3,1).
    // The reference compares acos() of these in float: do the same so that near-ties fall the same way.
    const V3 origin = vertices[0];
    V3 dir[3];
    for (int k = 0; k < 3; ++k)
        dir[k] = normalized(vertices[k + 1] - origin);
    float angle[3];
    for (int k = 0; k < 3; ++k)
        angle[k] = std::abs(std::acos(dot(dir[k], dir[(k + 1) % 3])));
    int widest = 2;
    if (angle[0] >= angle[1] && angle[0] >= angle[2])
        widest = 0;
    else if (angle[1] >= angle[2] && angle[1] >= angle[0])
        widest = 1;

    // corner ids: [origin, x end, y end]; the remaining vertex is the opposite corner
    uint32_t x_end = (uint32_t)(widest % 3 + 1), y_end = (uint32_t)((widest + 1) % 3 + 1);
    PlaneShape shape;
    shape.origin = origin;
    shape.x_axis = vertices[x_end] - origin;
    shape.y_axis = vertices[y_end] - origin;
    const bool flipped = dot(face_normal[0], normalized(cross(shape.x_axis, shape.y_axis))) < 0;
    if (flipped)
        std::swap(shape.x_axis, shape.y_axis);

    if (texcoords.empty()) {
        shape.texcoords[0] = V2{ 0, 0 };
        shape.texcoords[1] = V2{ 1, 0 };
        shape.texcoords[2] = V2{ 0, 1 };
        shape.texcoords[3] = V2{ 1, 1 };
    } else {
        // The reference fills slot (k + widest) % 3 + 1 from vertex k + 1 (k = 0, 1, 2) after exchanging vertices 1 and 2 when
        // the frame was flipped — an assignment by vertex number, not by corner role; kept as it is (TriMesh.cpp:606-611).
        uint32_t id[4] = { 0, 1, 2, 3 };
        if (flipped)
            std::swap(id[1], id[2]);
        shape.texcoords[0] = texcoords[id[0]];
        for (int k = 0; k < 3; ++k)
            shape.texcoords[(k + widest) % 3 + 1] = texcoords[id[k + 1]];
    }
    return shape;
}

// Does this mesh tessellate a sphere? Then an area light on it is sampled as the analytic sphere (AreaLight.cpp:60-62). The
// criteria are the reference's (TriMesh.cpp:637-731), checked in this order:
//   at least 32 faces; a bounding box with volume whose three extents agree to 1e-4 (relative); all vertices at one squared
//   distance from the box centre (mean, +- 1e-4 absolute); no two faces that share an edge are perpendicular and none of those
//   is degenerate (this is what keeps the default cylinder out); vertices in all eight octants around the centre.
std::optional<SphereShape> TriMesh::getAsSphere() const
{
    constexpr float kSphereEps = 1e-4f, kPerpendicularEps = 1e-7f;
    if (faceCount() < 32)
        return std::nullopt;
    const BBox box  = computeBBox();
    const V3 extent = box.diameter();
    const V3 centre = box.center();
    if (extent.x * extent.y * extent.z <= kSphereEps)
        return std::nullopt;
    auto differ = [&](float a, float b) { return std::abs((a - b) / std::max(1e-5f, a + b)) > kSphereEps; };
    if (differ(extent.x, extent.y) || differ(extent.x, extent.z) || differ(extent.y, extent.z))
        return std::nullopt;

    float radius2 = 0;
    for (const V3& v : vertices)
        radius2 += dot(centre - v, centre - v);
    radius2 /= (float)vertices.size();
    if (radius2 <= kSphereEps)
        return std::nullopt;
    bool octant[8] = {};
    for (const V3& v : vertices) {
        const V3 d = centre - v;
        if (std::abs(dot(d, d) - radius2) > kSphereEps)
            return std::nullopt;
        octant[(d.x < 0 ? 1 : 0) | (d.y < 0 ? 2 : 0) | (d.z < 0 ? 4 : 0)] = true;
    }

    // faces across every shared edge: directed edge (a, b) of one face and (b, a) of another
    std::map<std::pair<uint32_t, uint32_t>, size_t> face_of_edge;
    for (size_t f = 0; f < faceCount(); ++f)
        for (int k = 0; k < 3; ++k)
            face_of_edge[{ indices[4 * f + k], indices[4 * f + (k + 1) % 3] }] = f;
    auto unit_normal = [&](size_t f) { return normalized(computeTriangleNormal(vertices[indices[4 * f]], vertices[indices[4 * f + 1]], vertices[indices[4 * f + 2]])); };
    auto has_nan     = [](V3 n) { return std::isnan(n.x) || std::isnan(n.y) || std::isnan(n.z); };
    for (size_t f = 0; f < faceCount(); ++f)
        for (int k = 0; k < 3; ++k) {
            const auto twin = face_of_edge.find({ indices[4 * f + (k + 1) % 3], indices[4 * f + k] });
            if (twin == face_of_edge.end())
                continue;
            const V3 n1 = unit_normal(f), n2 = unit_normal(twin->second);
            if (has_nan(n1) || has_nan(n2) || std::abs(dot(n1, n2)) <= kPerpendicularEps)
                return std::nullopt;
        }
    for (bool seen : octant)
        if (!seen)
            return std::nullopt;
    return SphereShape{ centre, std::sqrt(radius2) };
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
//
// A table-driven reader: the header is parsed into elements and typed properties, and one generic decoder walks the body
// in file order. What a mesh needs is picked by NAME (x y z, nx ny nz, s t / u v; the `vertex_indices` / `vertex_index`
// list), everything else — other elements, extra properties, lists — is decoded and dropped. Every scalar type of the
// format is understood (char … double and the int8 … float64 spellings, ascii / little / big endian); the reference's reader
// (src/runtime/mesh/PlyFile.cpp:70-290) handles float properties and uchar / uint lists only, for which this one yields the
// same mesh: normals renormalised, quads and larger polygons split as a fan around their first corner (the reference
// ear-clips polygons of more than four corners and falls back to that fan, :30-66), four indices per face.

namespace {
namespace ply {

enum class Scalar { I8, U8, I16, U16, I32, U32, F32, F64 };

struct ScalarName {
    const char* name;
    Scalar type;
};
constexpr ScalarName kScalarNames[] = {
    { "char", Scalar::I8 },    { "int8", Scalar::I8 },     { "uchar", Scalar::U8 },  { "uint8", Scalar::U8 },
    { "short", Scalar::I16 },  { "int16", Scalar::I16 },   { "ushort", Scalar::U16 }, { "uint16", Scalar::U16 },
    { "int", Scalar::I32 },    { "int32", Scalar::I32 },   { "uint", Scalar::U32 },  { "uint32", Scalar::U32 },
    { "float", Scalar::F32 },  { "float32", Scalar::F32 }, { "double", Scalar::F64 }, { "float64", Scalar::F64 },
};
constexpr size_t kScalarSize[] = { 1, 1, 2, 2, 4, 4, 4, 8 };

struct Property {
    std::string name;
    Scalar type       = Scalar::F32; // value type (element type of a list)
    bool is_list      = false;
    Scalar count_type = Scalar::U8;
};
struct Element {
    std::string name;
    size_t count = 0;
    std::vector<Property> props;
};
enum class Format { Ascii, Little, Big };

// Body of the file: whitespace-separated tokens (ascii) or packed scalars (binary)
class Body {
public:
    Body(const std::vector<char>& bytes, size_t offset, Format format, const std::string& path)
        : mBytes(bytes)
        , mPos(offset)
        , mFormat(format)
        , mPath(path)
    {
    }
    double next(Scalar type)
    {
        if (mFormat == Format::Ascii) {
            while (mPos < mBytes.size() && std::isspace((unsigned char)mBytes[mPos]))
                ++mPos;
            const size_t start = mPos;
            while (mPos < mBytes.size() && !std::isspace((unsigned char)mBytes[mPos]))
                ++mPos;
            if (start == mPos)
                throw std::runtime_error("PLY file '" + mPath + "': unexpected end of data");
            return std::strtod(std::string(&mBytes[start], mPos - start).c_str(), nullptr);
        }
        const size_t n = kScalarSize[(int)type];
        if (mPos + n > mBytes.size())
            throw std::runtime_error("PLY file '" + mPath + "': truncated data");
        unsigned char raw[8];
        for (size_t k = 0; k < n; ++k) // to host (little endian) byte order
            raw[k] = (unsigned char)mBytes[mPos + (mFormat == Format::Big ? n - 1 - k : k)];
        mPos += n;
        switch (type) {
        case Scalar::I8: return (double)(int8_t)raw[0];
        case Scalar::U8: return (double)raw[0];
        case Scalar::I16: { int16_t v; std::memcpy(&v, raw, 2); return v; }
        case Scalar::U16: { uint16_t v; std::memcpy(&v, raw, 2); return v; }
        case Scalar::I32: { int32_t v; std::memcpy(&v, raw, 4); return v; }
        case Scalar::U32: { uint32_t v; std::memcpy(&v, raw, 4); return v; }
        case Scalar::F32: { float v; std::memcpy(&v, raw, 4); return v; }
        default: { double v; std::memcpy(&v, raw, 8); return v; }
        }
    }

private:
    const std::vector<char>& mBytes;
    size_t mPos;
    Format mFormat;
    const std::string& mPath;
};

Scalar scalarOf(const std::string& word, const std::string& path)
{
    for (const auto& s : kScalarNames)
        if (word == s.name)
            return s.type;
    throw std::runtime_error("PLY file '" + path + "': unknown property type '" + word + "'");
}

} // namespace ply
} // namespace

TriMesh load_ply(const std::string& path)
{
    using namespace ply;
    std::vector<char> bytes;
    {
        std::ifstream stream(path, std::ios::in | std::ios::binary);
        if (!stream)
            throw std::runtime_error("PLY file '" + path + "' can not be opened");
        bytes.assign(std::istreambuf_iterator<char>(stream), std::istreambuf_iterator<char>());
    }

    // ---- header: lines up to "end_header"
    Format format = Format::Ascii;
    std::vector<Element> elements;
    size_t pos     = 0;
    bool first     = true, ended = false;
    while (pos < bytes.size() && !ended) {
        size_t eol = pos;
        while (eol < bytes.size() && bytes[eol] != '\n')
            ++eol;
        std::stringstream line(std::string(&bytes[pos], eol - pos));
        pos = eol + 1;
        std::string word;
        line >> word;
        if (first) {
            if (word != "ply")
                throw std::runtime_error("'" + path + "' is not a ply file");
            first = false;
        } else if (word == "format") {
            line >> word;
            if (word == "binary_little_endian")
                format = Format::Little;
            else if (word == "binary_big_endian")
                format = Format::Big;
            else if (word != "ascii")
                throw std::runtime_error("PLY file '" + path + "': unknown format '" + word + "'");
        } else if (word == "element") {
            Element e;
            line >> e.name >> e.count;
            elements.push_back(e);
        } else if (word == "property") {
            if (elements.empty())
                throw std::runtime_error("PLY file '" + path + "': property outside of an element");
            Property p;
            line >> word;
            if (word == "list") {
                std::string count_word, value_word;
                line >> count_word >> value_word;
                p.is_list    = true;
                p.count_type = scalarOf(count_word, path);
                p.type       = scalarOf(value_word, path);
            } else {
                p.type = scalarOf(word, path);
            }
            line >> p.name;
            elements.back().props.push_back(p);
        } else if (word == "end_header") {
            ended = true;
        } // comment, obj_info: skipped
    }
    if (!ended)
        throw std::runtime_error("PLY file '" + path + "': header has no end");

    // ---- body, element after element in file order
    enum Slot { X, Y, Z, NX, NY, NZ, U, V, SlotCount };
    static const struct {
        const char* name;
        Slot slot;
    } kSlots[] = { { "x", X }, { "y", Y }, { "z", Z }, { "nx", NX }, { "ny", NY }, { "nz", NZ }, { "u", U }, { "s", U }, { "v", V }, { "t", V } };

    TriMesh mesh;
    bool have_positions = false, have_faces = false;
    Body body(bytes, pos, format, path);
    std::vector<uint32_t> corners;
    for (const Element& e : elements) {
        if (e.name == "vertex") {
            std::vector<int> slot_of(e.props.size(), -1);
            bool present[SlotCount] = {};
            for (size_t k = 0; k < e.props.size(); ++k)
                for (const auto& s : kSlots)
                    if (!e.props[k].is_list && e.props[k].name == s.name) {
                        slot_of[k]       = s.slot;
                        present[s.slot]  = true;
                    }
            have_positions         = present[X] && present[Y] && present[Z];
            const bool has_normals = present[NX] && present[NY] && present[NZ];
            const bool has_uvs     = present[U] && present[V];
            mesh.vertices.reserve(e.count);
            for (size_t i = 0; i < e.count; ++i) {
                float vals[SlotCount] = {};
                for (size_t k = 0; k < e.props.size(); ++k) {
                    const Property& p = e.props[k];
                    if (p.is_list) {
                        for (size_t n = (size_t)body.next(p.count_type); n > 0; --n)
                            body.next(p.type);
                    } else {
                        const float v = (float)body.next(p.type);
                        if (slot_of[k] >= 0)
                            vals[slot_of[k]] = v;
                    }
                }
                mesh.vertices.emplace_back(vals[X], vals[Y], vals[Z]);
                if (has_normals) {
                    float n = std::sqrt(vals[NX] * vals[NX] + vals[NY] * vals[NY] + vals[NZ] * vals[NZ]);
                    if (n == 0.0f)
                        n = 1.0f;
                    mesh.normals.emplace_back(vals[NX] / n, vals[NY] / n, vals[NZ] / n);
                }
                if (has_uvs)
                    mesh.texcoords.push_back(V2{ vals[U], vals[V] });
            }
        } else if (e.name == "face") {
            mesh.indices.reserve(e.count * 4);
            for (size_t i = 0; i < e.count; ++i) {
                for (const Property& p : e.props) {
                    if (!p.is_list) {
                        body.next(p.type);
                        continue;
                    }
                    const bool is_corners = p.name == "vertex_indices" || p.name == "vertex_index";
                    have_faces |= is_corners;
                    corners.clear();
                    for (size_t n = (size_t)body.next(p.count_type); n > 0; --n) {
                        const double idx = body.next(p.type);
                        if (is_corners) {
                            if (idx < 0 || idx >= (double)mesh.vertices.size())
                                throw std::runtime_error("PLY file '" + path + "': face index out of range");
                            corners.push_back((uint32_t)idx);
                        }
                    }
                    // a fan around the first corner; four indices per triangle (the fourth is padding)
                    for (size_t j = 2; is_corners && j < corners.size(); ++j)
                        mesh.indices.insert(mesh.indices.end(), { corners[0], corners[j - 1], corners[j], 0u });
                }
            }
        } else {
            for (size_t i = 0; i < e.count; ++i)
                for (const Property& p : e.props)
                    for (size_t n = p.is_list ? (size_t)body.next(p.count_type) : 1; n > 0; --n)
                        body.next(p.type);
        }
    }
    if (!have_positions || !have_faces || mesh.vertices.empty() || mesh.indices.empty())
        throw std::runtime_error("PLY file '" + path + "' does not contain valid mesh data");

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
