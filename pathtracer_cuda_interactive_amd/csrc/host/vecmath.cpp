#include "vecmath.h"

namespace pth {

// 3x3 minor determinant of m with row r and column c removed.
static float minor3(const Mat4& m, int r, int c) {
    int rr[3], cc[3];
    for (int i = 0, k = 0; i < 4; i++) if (i != r) rr[k++] = i;
    for (int j = 0, k = 0; j < 4; j++) if (j != c) cc[k++] = j;
    auto a = [&](int i, int j) { return m(rr[i], cc[j]); };
    return a(0, 0) * a(1, 1) * a(2, 2) - a(0, 0) * a(1, 2) * a(2, 1) - a(1, 0) * a(0, 1) * a(2, 2) +
           a(1, 0) * a(0, 2) * a(2, 1) + a(2, 0) * a(0, 1) * a(1, 2) - a(2, 0) * a(0, 2) * a(1, 1);
}

// Adjugate / determinant, as matrix.h:71-211 does (inv_det is a double there: matrix.h:204).
Mat4 inverse(const Mat4& m) {
    Mat4 inv;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            float cof = minor3(m, j, i);
            inv(i, j) = ((i + j) & 1) ? -cof : cof;
        }
    float det = m(0, 0) * inv(0, 0) + m(0, 1) * inv(1, 0) + m(0, 2) * inv(2, 0) + m(0, 3) * inv(3, 0);
    if (det == 0) return Mat4::zero();
    double inv_det = 1.0 / det;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) inv(i, j) = float(inv(i, j) * inv_det);
    return inv;
}

Mat4 translate(f3 d) {
    Mat4 r = Mat4::identity();
    r(0, 3) = d.x; r(1, 3) = d.y; r(2, 3) = d.z;
    return r;
}

Mat4 scale(f3 s) {
    Mat4 r = Mat4::identity();
    r(0, 0) = s.x; r(1, 1) = s.y; r(2, 2) = s.z;
    return r;
}

Mat4 rotate(float angle_deg, f3 axis) {
    f3 a = normalize(axis);
    float s = sinf(radians(angle_deg));
    float c = cosf(radians(angle_deg));
    Mat4 m = Mat4::zero();
    m(0, 0) = a.x * a.x + (1 - a.x * a.x) * c;
    m(0, 1) = a.x * a.y * (1 - c) - a.z * s;
    m(0, 2) = a.x * a.z * (1 - c) + a.y * s;
    m(1, 0) = a.x * a.y * (1 - c) + a.z * s;
    m(1, 1) = a.y * a.y + (1 - a.y * a.y) * c;
    m(1, 2) = a.y * a.z * (1 - c) - a.x * s;
    m(2, 0) = a.x * a.z * (1 - c) - a.y * s;
    m(2, 1) = a.y * a.z * (1 - c) + a.x * s;
    m(2, 2) = a.z * a.z + (1 - a.z * a.z) * c;
    m(3, 3) = 1;
    return m;
}

Mat4 look_at(f3 pos, f3 look, f3 up) {
    Mat4 m = Mat4::zero();
    f3 dir = normalize(look - pos);
    f3 left = normalize(cross(normalize(up), dir));
    f3 new_up = cross(dir, left);
    m(0, 0) = left.x;   m(1, 0) = left.y;   m(2, 0) = left.z;
    m(0, 1) = new_up.x; m(1, 1) = new_up.y; m(2, 1) = new_up.z;
    m(0, 2) = dir.x;    m(1, 2) = dir.y;    m(2, 2) = dir.z;
    m(0, 3) = pos.x;    m(1, 3) = pos.y;    m(2, 3) = pos.z;
    m(3, 3) = 1;
    return m;
}

f3 xform_point(const Mat4& x, f3 p) {
    float tx = x(0, 0) * p.x + x(0, 1) * p.y + x(0, 2) * p.z + x(0, 3);
    float ty = x(1, 0) * p.x + x(1, 1) * p.y + x(1, 2) * p.z + x(1, 3);
    float tz = x(2, 0) * p.x + x(2, 1) * p.y + x(2, 2) * p.z + x(2, 3);
    float tw = x(3, 0) * p.x + x(3, 1) * p.y + x(3, 2) * p.z + x(3, 3);
    float inv_w = float(1) / tw;
    return {tx * inv_w, ty * inv_w, tz * inv_w};
}

f3 xform_normal(const Mat4& ix, f3 n) {
    return normalize(f3{ix(0, 0) * n.x + ix(1, 0) * n.y + ix(2, 0) * n.z,
                        ix(0, 1) * n.x + ix(1, 1) * n.y + ix(2, 1) * n.z,
                        ix(0, 2) * n.x + ix(1, 2) * n.y + ix(2, 2) * n.z});
}

}  // namespace pth
