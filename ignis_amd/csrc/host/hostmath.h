// Small float vector / affine-matrix helpers for the host loader (the reference
// uses Eigen; only what the loader path needs is restated here).
#pragma once

#include <algorithm>
#include <cmath>
#include <limits>

namespace igh {

struct V2 {
    float x = 0, y = 0;
};

struct V3 {
    float x = 0, y = 0, z = 0;
    V3() = default;
    V3(float x_, float y_, float z_)
        : x(x_)
        , y(y_)
        , z(z_)
    {
    }
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};

inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return a * s; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float norm(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalized(V3 a)
{
    const float n = norm(a);
    return n > 0 ? V3(a.x / n, a.y / n, a.z / n) : a; // Eigen: *this / norm()
}
inline V3 vmin(V3 a, V3 b) { return V3(std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)); }
inline V3 vmax(V3 a, V3 b) { return V3(std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)); }

// Row-major 3x3
struct M3 {
    float m[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
};

inline V3 operator*(const M3& a, V3 v)
{
    return V3(a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z,
              a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
              a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z);
}

inline M3 operator*(const M3& a, const M3& b)
{
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return r;
}

inline M3 transpose(const M3& a)
{
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            r.m[i][j] = a.m[j][i];
    return r;
}

// Cofactor inverse (what Eigen does for fixed 3x3)
inline M3 inverse(const M3& a)
{
    M3 c;
    c.m[0][0] = a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1];
    c.m[0][1] = a.m[0][2] * a.m[2][1] - a.m[0][1] * a.m[2][2];
    c.m[0][2] = a.m[0][1] * a.m[1][2] - a.m[0][2] * a.m[1][1];
    c.m[1][0] = a.m[1][2] * a.m[2][0] - a.m[1][0] * a.m[2][2];
    c.m[1][1] = a.m[0][0] * a.m[2][2] - a.m[0][2] * a.m[2][0];
    c.m[1][2] = a.m[0][2] * a.m[1][0] - a.m[0][0] * a.m[1][2];
    c.m[2][0] = a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0];
    c.m[2][1] = a.m[0][1] * a.m[2][0] - a.m[0][0] * a.m[2][1];
    c.m[2][2] = a.m[0][0] * a.m[1][1] - a.m[0][1] * a.m[1][0];
    const float det    = a.m[0][0] * c.m[0][0] + a.m[0][1] * c.m[1][0] + a.m[0][2] * c.m[2][0];
    const float invdet = 1.0f / det;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            c.m[i][j] *= invdet;
    return c;
}

// Affine transform: linear 3x3 + translation (Eigen Transform<float,3,Affine>)
struct Affine {
    M3 L;
    V3 t;

    V3 point(V3 p) const { return L * p + t; }
    V3 direction(V3 d) const { return L * d; }
    bool isIdentity() const
    {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                if (L.m[i][j] != (i == j ? 1.0f : 0.0f))
                    return false;
        return t.x == 0 && t.y == 0 && t.z == 0;
    }
};

inline Affine operator*(const Affine& a, const Affine& b)
{
    Affine r;
    r.L = a.L * b.L;
    r.t = a.L * b.t + a.t;
    return r;
}

inline Affine inverse(const Affine& a)
{
    Affine r;
    r.L = inverse(a.L);
    r.t = -(r.L * a.t);
    return r;
}

struct BBox {
    V3 min = V3(std::numeric_limits<float>::infinity(), std::numeric_limits<float>::infinity(), std::numeric_limits<float>::infinity());
    V3 max = V3(-std::numeric_limits<float>::infinity(), -std::numeric_limits<float>::infinity(), -std::numeric_limits<float>::infinity());

    void extend(V3 p)
    {
        min = vmin(min, p);
        max = vmax(max, p);
    }
    void extend(const BBox& b)
    {
        min = vmin(min, b.min);
        max = vmax(max, b.max);
    }
    V3 center() const { return (max + min) * 0.5f; }
    V3 diameter() const { return max - min; }
    // libbvh: (d0 + d1) * d2 + d0 * d1
    float halfArea() const
    {
        const V3 d = max - min;
        return (d.x + d.y) * d.z + d.x * d.y;
    }
    // src/runtime/math/BoundingBox.h:56-65
    void inflate(float eps)
    {
        const V3 len = max - min;
        for (int i = 0; i < 3; ++i) {
            if (len[i] < eps) {
                max[i] += eps / 2;
                min[i] -= eps / 2;
            }
        }
    }
    // src/runtime/math/BoundingBox.h:84-95
    BBox transformed(const Affine& T) const
    {
        BBox b;
        b.extend(T.point(min));
        b.extend(T.point(V3(max.x, min.y, min.z)));
        b.extend(T.point(V3(min.x, max.y, min.z)));
        b.extend(T.point(V3(max.x, max.y, min.z)));
        b.extend(T.point(V3(min.x, min.y, max.z)));
        b.extend(T.point(V3(max.x, min.y, max.z)));
        b.extend(T.point(V3(min.x, max.y, max.z)));
        b.extend(T.point(max));
        return b;
    }
};

} // namespace igh
