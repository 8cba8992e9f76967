// Triangle mesh container + the mesh operations the scene loader needs.
// Follows src/runtime/mesh/TriMesh.{h,cpp} and src/runtime/mesh/PlyFile.cpp.
#pragma once

#include "hostmath.h"

#include <array>
#include <cstdint>
#include <optional>
#include <string>
#include <vector>

namespace igh {

struct PlaneShape {
    V3 origin, x_axis, y_axis;
    std::array<V2, 4> texcoords;
};

struct SphereShape {
    V3 origin;
    float radius = 0;
};

struct TriMesh {
    std::vector<V3> vertices;
    std::vector<V3> normals;
    std::vector<V2> texcoords;
    std::vector<uint32_t> indices; // 4 per face, 4th is padding (PlyFile.cpp:224-244)

    size_t faceCount() const { return indices.size() / 4; }

    void flipNormals();                 // TriMesh.cpp:34-43
    void computeVertexNormals();        // TriMesh.cpp:96-116
    void makeTexCoordsNormalized();     // TriMesh.cpp:123-142
    void setupFaceNormalsAsVertexNormals(); // TriMesh.cpp:152-197
    void transform(const Affine& t);    // TriMesh.cpp:241-273
    BBox computeBBox() const;           // TriMesh.cpp:144-150
    float computeArea() const;          // TriMesh.cpp:199-209
    std::optional<PlaneShape> getAsPlane() const; // TriMesh.cpp:521-633
    std::optional<SphereShape> getAsSphere() const; // TriMesh.cpp:637-731

    static TriMesh MakePlane(V3 origin, V3 x_axis, V3 y_axis);     // TriMesh.cpp:783-817,1039-1044
    static TriMesh MakeRectangle(V3 p0, V3 p1, V3 p2, V3 p3);      // TriMesh.cpp:1053-1059
    static TriMesh MakeTriangle(V3 p0, V3 p1, V3 p2);              // TriMesh.cpp:1046-1051
    static TriMesh MakeBox(V3 origin, V3 x_axis, V3 y_axis, V3 z_axis);
    // procedural shapes of the "uvsphere", "icosphere", "disk", "cone" and "cylinder" plugins: the tessellations
    // (triangle sets, shared vertices, normals, texture coordinates) of TriMesh.cpp:819-1131
    static TriMesh MakeUVSphere(V3 center, float radius, uint32_t stacks, uint32_t slices);
    static TriMesh MakeIcoSphere(V3 center, float radius, uint32_t subdivisions);
    static TriMesh MakeDisk(V3 center, V3 normal, float radius, uint32_t sections);
    static TriMesh MakeCone(V3 base_center, float base_radius, V3 tip, uint32_t sections, bool fill_cap);
    static TriMesh MakeCylinder(V3 base_center, float base_radius, V3 top_center, float top_radius, uint32_t sections, bool fill_cap);
};

// Throws std::runtime_error with a message on malformed input.
TriMesh load_ply(const std::string& path);
TriMesh load_obj(const std::string& path); // src/runtime/mesh/ObjFile.cpp
TriMesh load_serialized(const std::string& path, size_t shape_index); // src/runtime/mesh/MtsSerializedFile.cpp

} // namespace igh
