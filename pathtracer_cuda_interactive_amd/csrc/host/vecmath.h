// Host-side fp32 vector / matrix helpers for the scene pipeline.
// Semantics follow the reference's host math: float3/float is a reciprocal
// multiply (cutil_math.h:349-353), normalize = v * (1/sqrt(dot)) (cutil_math.h:401-405,51-54).
#pragma once
#include <cmath>
#include <cstdint>

namespace pth {

struct f3 { float x, y, z; };
struct i3 { int32_t x, y, z; };

inline f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
inline f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
inline f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline f3 operator*(float s, f3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline f3 operator/(f3 a, float s) { float inv = 1.0f / s; return a * inv; }
inline float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline f3 cross(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(f3 v) { return sqrtf(dot(v, v)); }
inline f3 normalize(f3 v) { float inv_len = 1.0f / sqrtf(dot(v, v)); return v * inv_len; }
inline float tmin(float a, float b) { return a < b ? a : b; }   // torrey.cuh:75-78
inline float tmax(float a, float b) { return a > b ? a : b; }   // torrey.cuh:70-73

constexpr float kPi = float(3.14159265358979323846);
inline float radians(float deg) { return (kPi / float(180)) * deg; }
inline float degrees(float rad) { return (float(180) / kPi) * rad; }

// Row-major 4x4 (matrix.h:6-66)
struct Mat4 {
    float m[4][4];
    static Mat4 zero() { Mat4 r; for (auto& row : r.m) for (float& v : row) v = 0.0f; return r; }
    static Mat4 identity() { Mat4 r = zero(); for (int i = 0; i < 4; i++) r.m[i][i] = 1.0f; return r; }
    float& operator()(int i, int j) { return m[i][j]; }
    const float& operator()(int i, int j) const { return m[i][j]; }
};

inline Mat4 operator*(const Mat4& a, const Mat4& b) {   // matrix.h:213-224
    Mat4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            float acc = 0.0f;
            for (int k = 0; k < 4; k++) acc += a(i, k) * b(k, j);
            r(i, j) = acc;
        }
    return r;
}

Mat4 inverse(const Mat4& m);                       // matrix.h:71-211 (cofactor expansion)
Mat4 translate(f3 d);                              // transform.cpp:6-11
Mat4 scale(f3 s);                                  // transform.cpp:13-18
Mat4 rotate(float angle_deg, f3 axis);             // transform.cpp:20-46
Mat4 look_at(f3 pos, f3 look, f3 up);              // transform.cpp:48-70
f3 xform_point(const Mat4& x, f3 p);               // transform.cpp:80-88
f3 xform_normal(const Mat4& inv_x, f3 n);          // transform.cpp:96-101

}  // namespace pth
