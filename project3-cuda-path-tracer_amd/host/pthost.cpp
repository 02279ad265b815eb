// pthost.cpp -- see pthost.h.  Host-only C++ (no GPU code): scene text -> reference-layout structs
// with GLM 0.9.6.3's exact arithmetic, camera set-up, image output.
#include "pthost.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace {

char g_err[512] = "";

struct V3 { float x, y, z; };
struct M4 { float m[4][4]; };          // m[col][row], glm::mat4

V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
V3 scl(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
V3 neg(V3 a) { return v3(-a.x, -a.y, -a.z); }
float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }                 // func_geometric.inl:64-72
V3 cross(V3 x, V3 y) { return v3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
V3 normalize(V3 a) { return scl(a, 1.0f / std::sqrt(dot(a, a))); }                    // :153-159
float length(V3 a) { return std::sqrt(dot(a, a)); }

M4 identity() { M4 r; memset(&r, 0, sizeof r); for (int i = 0; i < 4; ++i) r.m[i][i] = 1.0f; return r; }

// column helpers: c = a*s, c = a + b on vec4 columns
void col_scale(const float a[4], float s, float o[4]) { for (int i = 0; i < 4; ++i) o[i] = a[i] * s; }
void col_add(const float a[4], const float b[4], float o[4]) { for (int i = 0; i < 4; ++i) o[i] = a[i] + b[i]; }

// glm::translate (gtc/matrix_transform.inl:40-49): Result[3] = m[0]*v[0] + m[1]*v[1] + m[2]*v[2] + m[3]
M4 translate(const M4 &m, V3 v) {
    M4 r = m;
    float a[4], b[4], c[4], t[4];
    col_scale(m.m[0], v.x, a); col_scale(m.m[1], v.y, b); col_scale(m.m[2], v.z, c);
    col_add(a, b, t); col_add(t, c, t); col_add(t, m.m[3], t);
    memcpy(r.m[3], t, sizeof t);
    return r;
}

// glm::rotate (gtc/matrix_transform.inl:52-85)
M4 rotate(const M4 &m, float angle, V3 v) {
    const float a = angle;
    const float c = std::cos(a);
    const float s = std::sin(a);
    V3 axis = normalize(v);
    V3 temp = scl(axis, (1.0f - c));                 // (T(1) - c) * axis
    float R[3][3];
    const float ax[3] = {axis.x, axis.y, axis.z}, tp[3] = {temp.x, temp.y, temp.z};
    R[0][0] = c + tp[0] * ax[0];
    R[0][1] = 0 + tp[0] * ax[1] + s * ax[2];
    R[0][2] = 0 + tp[0] * ax[2] - s * ax[1];
    R[1][0] = 0 + tp[1] * ax[0] - s * ax[2];
    R[1][1] = c + tp[1] * ax[1];
    R[1][2] = 0 + tp[1] * ax[2] + s * ax[0];
    R[2][0] = 0 + tp[2] * ax[0] + s * ax[1];
    R[2][1] = 0 + tp[2] * ax[1] - s * ax[0];
    R[2][2] = c + tp[2] * ax[2];
    M4 r;
    for (int k = 0; k < 3; ++k) {
        float a0[4], a1[4], a2[4], t[4];
        col_scale(m.m[0], R[k][0], a0); col_scale(m.m[1], R[k][1], a1); col_scale(m.m[2], R[k][2], a2);
        col_add(a0, a1, t); col_add(t, a2, t);
        memcpy(r.m[k], t, sizeof t);
    }
    memcpy(r.m[3], m.m[3], sizeof r.m[3]);
    return r;
}

// glm::scale (gtc/matrix_transform.inl:122-134)
M4 scale(const M4 &m, V3 v) {
    M4 r;
    col_scale(m.m[0], v.x, r.m[0]); col_scale(m.m[1], v.y, r.m[1]); col_scale(m.m[2], v.z, r.m[2]);
    memcpy(r.m[3], m.m[3], sizeof r.m[3]);
    return r;
}

// tmat4x4 operator* (type_mat4x4.inl:686-704): Result[j] = A0*B[j][0] + A1*B[j][1] + A2*B[j][2] + A3*B[j][3]
M4 mul(const M4 &A, const M4 &B) {
    M4 r;
    for (int j = 0; j < 4; ++j) {
        float a0[4], a1[4], a2[4], a3[4], t[4];
        col_scale(A.m[0], B.m[j][0], a0); col_scale(A.m[1], B.m[j][1], a1);
        col_scale(A.m[2], B.m[j][2], a2); col_scale(A.m[3], B.m[j][3], a3);
        col_add(a0, a1, t); col_add(t, a2, t); col_add(t, a3, t);
        memcpy(r.m[j], t, sizeof t);
    }
    return r;
}

// detail::compute_inverse(tmat4x4) (type_mat4x4.inl:36-92)
M4 inverse(const M4 &mm) {
    const float (*m)[4] = mm.m;
    float Coef00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float Coef02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float Coef03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float Coef04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float Coef06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float Coef07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float Coef08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float Coef10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float Coef11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float Coef12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float Coef14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float Coef15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float Coef16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float Coef18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float Coef19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float Coef20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float Coef22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float Coef23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    const float Fac0[4] = {Coef00, Coef00, Coef02, Coef03}, Fac1[4] = {Coef04, Coef04, Coef06, Coef07};
    const float Fac2[4] = {Coef08, Coef08, Coef10, Coef11}, Fac3[4] = {Coef12, Coef12, Coef14, Coef15};
    const float Fac4[4] = {Coef16, Coef16, Coef18, Coef19}, Fac5[4] = {Coef20, Coef20, Coef22, Coef23};
    const float Vec0[4] = {m[1][0], m[0][0], m[0][0], m[0][0]}, Vec1[4] = {m[1][1], m[0][1], m[0][1], m[0][1]};
    const float Vec2[4] = {m[1][2], m[0][2], m[0][2], m[0][2]}, Vec3[4] = {m[1][3], m[0][3], m[0][3], m[0][3]};
    float Inv[4][4];
    const float SignA[4] = {+1, -1, +1, -1}, SignB[4] = {-1, +1, -1, +1};
    for (int i = 0; i < 4; ++i) {
        Inv[0][i] = (Vec1[i] * Fac0[i] - Vec2[i] * Fac1[i] + Vec3[i] * Fac2[i]) * SignA[i];
        Inv[1][i] = (Vec0[i] * Fac0[i] - Vec2[i] * Fac3[i] + Vec3[i] * Fac4[i]) * SignB[i];
        Inv[2][i] = (Vec0[i] * Fac1[i] - Vec1[i] * Fac3[i] + Vec3[i] * Fac5[i]) * SignA[i];
        Inv[3][i] = (Vec0[i] * Fac2[i] - Vec1[i] * Fac4[i] + Vec2[i] * Fac5[i]) * SignB[i];
    }
    const float Row0[4] = {Inv[0][0], Inv[1][0], Inv[2][0], Inv[3][0]};
    float Dot0[4];
    for (int i = 0; i < 4; ++i) Dot0[i] = m[0][i] * Row0[i];
    const float Dot1 = (Dot0[0] + Dot0[1]) + (Dot0[2] + Dot0[3]);
    const float OneOverDeterminant = 1.0f / Dot1;
    M4 r;
    for (int c = 0; c < 4; ++c) for (int i = 0; i < 4; ++i) r.m[c][i] = Inv[c][i] * OneOverDeterminant;
    return r;
}

// glm::inverseTranspose(tmat4x4) (gtc/matrix_inverse.inl:95-147)
M4 inverse_transpose(const M4 &mm) {
    const float (*m)[4] = mm.m;
    float S00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float S01 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float S02 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float S03 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float S04 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float S05 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float S06 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float S07 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float S08 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float S09 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float S10 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float S11 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float S12 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float S13 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float S14 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float S15 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float S16 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float S17 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float S18 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    M4 I;
    I.m[0][0] = +(m[1][1] * S00 - m[1][2] * S01 + m[1][3] * S02);
    I.m[0][1] = -(m[1][0] * S00 - m[1][2] * S03 + m[1][3] * S04);
    I.m[0][2] = +(m[1][0] * S01 - m[1][1] * S03 + m[1][3] * S05);
    I.m[0][3] = -(m[1][0] * S02 - m[1][1] * S04 + m[1][2] * S05);
    I.m[1][0] = -(m[0][1] * S00 - m[0][2] * S01 + m[0][3] * S02);
    I.m[1][1] = +(m[0][0] * S00 - m[0][2] * S03 + m[0][3] * S04);
    I.m[1][2] = -(m[0][0] * S01 - m[0][1] * S03 + m[0][3] * S05);
    I.m[1][3] = +(m[0][0] * S02 - m[0][1] * S04 + m[0][2] * S05);
    I.m[2][0] = +(m[0][1] * S06 - m[0][2] * S07 + m[0][3] * S08);
    I.m[2][1] = -(m[0][0] * S06 - m[0][2] * S09 + m[0][3] * S10);
    I.m[2][2] = +(m[0][0] * S11 - m[0][1] * S09 + m[0][3] * S12);
    I.m[2][3] = -(m[0][0] * S08 - m[0][1] * S10 + m[0][2] * S12);
    I.m[3][0] = -(m[0][1] * S13 - m[0][2] * S14 + m[0][3] * S15);
    I.m[3][1] = +(m[0][0] * S13 - m[0][2] * S16 + m[0][3] * S17);
    I.m[3][2] = -(m[0][0] * S14 - m[0][1] * S16 + m[0][3] * S18);
    I.m[3][3] = +(m[0][0] * S15 - m[0][1] * S17 + m[0][2] * S18);
    // the four terms are summed left to right: ((+a) + b) + c) + d
    float Determinant = +m[0][0] * I.m[0][0] + m[0][1] * I.m[0][1] + m[0][2] * I.m[0][2] + m[0][3] * I.m[0][3];
    for (int c = 0; c < 4; ++c) for (int i = 0; i < 4; ++i) I.m[c][i] /= Determinant;
    return I;
}

// The scene text as a sequence of lines.  The whole file is read once and cut at every line end -- "\n", "\r\n" or a
// lone "\r", so scene files written on any platform load alike (what utilities.cpp:84-112 provides for the reference's
// loader); a last line without a terminator counts, a terminator at the very end opens no further line.  `next` hands out
// the lines in order; asking for one when none is left yields "" and ends `good()`, which is the stream state the
// reference's loops test (scene.cpp:21,66,124).
class Lines {
public:
    explicit Lines(const std::string &text) {
        size_t from = 0;
        const size_t n = text.size();
        while (from < n) {
            size_t to = text.find_first_of("\r\n", from);
            if (to == std::string::npos) to = n;
            rows_.emplace_back(text, from, to - from);
            from = to + ((to + 1 < n && text[to] == '\r' && text[to + 1] == '\n') ? 2 : 1);
        }
    }
    void next(std::string &line) {
        if (at_ < rows_.size()) line = rows_[at_++];
        else { line.clear(); spent_ = true; }
    }
    bool good() const { return !spent_; }
private:
    std::vector<std::string> rows_;
    size_t at_ = 0;
    bool spent_ = false;
};

std::vector<std::string> tokens_of(const std::string &s) {       // utilityCore::tokenizeString
    std::stringstream ss(s);
    std::vector<std::string> r;
    std::string w;
    while (ss >> w) r.push_back(w);
    return r;
}

V3 vec3_of(const std::vector<std::string> &t) {                  // glm::vec3(atof, atof, atof)
    return v3((float)atof(t[1].c_str()), (float)atof(t[2].c_str()), (float)atof(t[3].c_str()));
}

pt_vec3 P(V3 v) { pt_vec3 r = {v.x, v.y, v.z}; return r; }
V3 U(pt_vec3 v) { return v3(v.x, v.y, v.z); }

const float PI_F = 3.1415926535897932384626422832795028841971f;   // utilities.h:12

// main.cpp:53-67 (orbit state from the loaded camera) + main.cpp:102-120 (runCuda, camchanged)
void orbit_recompute(pt_camera &cam) {
    V3 view = U(cam.view);
    V3 viewXZ = v3(view.x, 0.0f, view.z);
    V3 viewZY = v3(0.0f, view.y, view.z);
    float phi = std::acos(dot(normalize(viewXZ), v3(0, 0, -1)));
    float theta = std::acos(dot(normalize(viewZY), v3(0, 1, 0)));
    V3 ogLookAt = U(cam.lookAt);
    float zoom = length(sub(U(cam.position), ogLookAt));
    V3 cameraPosition;
    cameraPosition.x = zoom * std::sin(phi) * std::sin(theta);
    cameraPosition.y = zoom * std::cos(theta);
    cameraPosition.z = zoom * std::cos(phi) * std::sin(theta);
    V3 v = neg(normalize(cameraPosition));
    V3 u = v3(0, 1, 0);
    V3 r = cross(v, u);
    cam.view = P(v);
    cam.up = P(cross(r, v));
    cam.right = P(r);
    cameraPosition = add(cameraPosition, U(cam.lookAt));
    cam.position = P(cameraPosition);
}

struct Builder {
    std::vector<pt_geom> geoms;
    std::vector<pt_material> mats;
    std::vector<pt_triangle> tris;
    std::vector<pt_mesh> meshes;
};

bool load_obj(const std::string &path, const M4 &T, std::vector<pt_triangle> &out, std::string &err) {
    std::ifstream f(path.c_str());
    if (!f.is_open()) { err = "cannot open mesh file " + path; return false; }
    std::vector<V3> vs;
    std::string line;
    while (std::getline(f, line)) {
        std::vector<std::string> t = tokens_of(line);
        if (t.empty()) continue;
        if (t[0] == "v" && t.size() >= 4) {
            V3 p = v3((float)atof(t[1].c_str()), (float)atof(t[2].c_str()), (float)atof(t[3].c_str()));
            // world space = vec3(transform * vec4(p, 1)), GLM order (type_mat4x4.inl:617-628)
            float w[3];
            for (int r = 0; r < 3; ++r)
                w[r] = (T.m[0][r] * p.x + T.m[1][r] * p.y) + (T.m[2][r] * p.z + T.m[3][r] * 1.0f);
            vs.push_back(v3(w[0], w[1], w[2]));
        } else if (t[0] == "f" && t.size() >= 4) {
            std::vector<int> idx;
            for (size_t k = 1; k < t.size(); ++k) {
                int i = atoi(t[k].c_str());                       // "i", "i/j", "i//k"
                if (i < 0) i = (int)vs.size() + 1 + i;
                if (i < 1 || i > (int)vs.size()) { err = "bad face index in " + path; return false; }
                idx.push_back(i - 1);
            }
            for (size_t k = 1; k + 1 < idx.size(); ++k) {         // fan triangulation
                pt_triangle tr;
                tr.v0 = P(vs[idx[0]]); tr.v1 = P(vs[idx[k]]); tr.v2 = P(vs[idx[k + 1]]);
                out.push_back(tr);
            }
        }
    }
    return true;
}

}  // namespace

extern "C" {

const char *pth_last_error(void) { return g_err; }

void pth_build_geom_matrices(pt_geom *g) {
    // The object-to-world matrix the reference's loader builds (utilities.cpp:65-72): translate * (rotX * rotY * rotZ) *
    // scale, each factor GLM's, angles given in degrees and converted as (angle * pi) / 180 in binary32, products taken
    // left to right -- the order of the roundings is what the golden Geom fixtures pin.
    const float euler[3] = {g->rotation.x, g->rotation.y, g->rotation.z};
    M4 spin = identity();
    for (int axis = 0; axis < 3; ++axis) {
        const V3 unit = v3(axis == 0 ? 1.0f : 0.0f, axis == 1 ? 1.0f : 0.0f, axis == 2 ? 1.0f : 0.0f);
        const M4 turn = rotate(identity(), euler[axis] * PI_F / 180, unit);
        spin = axis == 0 ? turn : mul(spin, turn);
    }
    const M4 to_world = mul(mul(translate(identity(), U(g->translation)), spin), scale(identity(), U(g->scale)));
    const M4 to_object = inverse(to_world), normals = inverse_transpose(to_world);
    memcpy(&g->transform, &to_world, 64);
    memcpy(&g->inverseTransform, &to_object, 64);
    memcpy(&g->invTranspose, &normals, 64);
}

pth_scene *pth_load_scene(const char *path) {
    std::ifstream file(path, std::ios::binary);
    if (!file.is_open()) { snprintf(g_err, sizeof g_err, "Error reading from file %s", path); return NULL; }
    std::stringstream whole;
    whole << file.rdbuf();
    Lines fp(whole.str());
    Builder b;
    pth_scene *s = (pth_scene *)calloc(1, sizeof(pth_scene));
    std::string dir(path);
    size_t slash = dir.find_last_of('/');
    dir = slash == std::string::npos ? std::string(".") : dir.substr(0, slash);
    bool have_cam = false;
    std::string line;
    while (fp.good()) {
        fp.next(line);
        if (line.empty()) continue;
        std::vector<std::string> tok = tokens_of(line);
        if (tok.empty()) continue;
        if (tok[0] == "MATERIAL") {                                   // Scene::loadMaterial, scene.cpp:153-188
            if (tok.size() < 2 || atoi(tok[1].c_str()) != (int)b.mats.size()) {
                snprintf(g_err, sizeof g_err, "MATERIAL ID does not match expected number of materials"); goto fail;
            }
            pt_material m; memset(&m, 0, sizeof m);
            for (int i = 0; i < 7; ++i) {                             // exactly seven property lines
                fp.next(line);
                std::vector<std::string> t = tokens_of(line);
                if (t.empty()) continue;
                if (t[0] == "RGB" && t.size() >= 4) m.color = P(vec3_of(t));
                else if (t[0] == "SPECEX" && t.size() >= 2) m.specular.exponent = (float)atof(t[1].c_str());
                else if (t[0] == "SPECRGB" && t.size() >= 4) m.specular.color = P(vec3_of(t));
                else if (t[0] == "REFL" && t.size() >= 2) m.hasReflective = (float)atof(t[1].c_str());
                else if (t[0] == "REFR" && t.size() >= 2) m.hasRefractive = (float)atof(t[1].c_str());
                else if (t[0] == "REFRIOR" && t.size() >= 2) m.indexOfRefraction = (float)atof(t[1].c_str());
                else if (t[0] == "EMITTANCE" && t.size() >= 2) m.emittance = (float)atof(t[1].c_str());
            }
            b.mats.push_back(m);
        } else if (tok[0] == "OBJECT") {                              // Scene::loadGeom, scene.cpp:35-90
            if (tok.size() < 2 || atoi(tok[1].c_str()) != (int)b.geoms.size()) {
                snprintf(g_err, sizeof g_err, "OBJECT ID does not match expected number of geoms"); goto fail;
            }
            pt_geom g; memset(&g, 0, sizeof g);
            std::string mesh_file;
            fp.next(line);
            if (!line.empty() && fp.good()) {
                std::vector<std::string> t = tokens_of(line);
                if (line == "sphere") g.type = PT_SPHERE;
                else if (line == "cube") g.type = PT_CUBE;
                else if (!t.empty() && t[0] == "mesh" && t.size() >= 2) { g.type = PT_TRIANGLE_MESH; mesh_file = t[1]; }
            }
            fp.next(line);
            if (!line.empty() && fp.good()) {
                std::vector<std::string> t = tokens_of(line);
                if (t.size() >= 2) g.materialid = atoi(t[1].c_str());
            }
            fp.next(line);
            while (!line.empty() && fp.good()) {
                std::vector<std::string> t = tokens_of(line);
                if (t.size() >= 4) {
                    if (t[0] == "TRANS") g.translation = P(vec3_of(t));
                    else if (t[0] == "ROTAT") g.rotation = P(vec3_of(t));
                    else if (t[0] == "SCALE") g.scale = P(vec3_of(t));
                }
                fp.next(line);
            }
            pth_build_geom_matrices(&g);
            if (g.type == PT_TRIANGLE_MESH) {
                pt_mesh me; me.geom_index = (int32_t)b.geoms.size(); me.first_triangle = (int32_t)b.tris.size();
                M4 T; memcpy(&T, &g.transform, 64);
                std::string err;
                std::string full = mesh_file.size() && mesh_file[0] == '/' ? mesh_file : dir + "/" + mesh_file;
                if (!load_obj(full, T, b.tris, err)) { snprintf(g_err, sizeof g_err, "%s", err.c_str()); goto fail; }
                me.triangle_count = (int32_t)b.tris.size() - me.first_triangle;
                b.meshes.push_back(me);
            }
            b.geoms.push_back(g);
        } else if (tok[0] == "CAMERA") {                              // Scene::loadCamera, scene.cpp:92-151
            pt_camera &camera = s->camera_loaded;
            memset(&camera, 0, sizeof camera);
            float fovy = 0.0f;
            for (int i = 0; i < 5; ++i) {
                fp.next(line);
                std::vector<std::string> t = tokens_of(line);
                if (t.empty()) continue;
                if (t[0] == "RES" && t.size() >= 3) { camera.resolution[0] = atoi(t[1].c_str()); camera.resolution[1] = atoi(t[2].c_str()); }
                else if (t[0] == "FOVY" && t.size() >= 2) fovy = (float)atof(t[1].c_str());
                else if (t[0] == "ITERATIONS" && t.size() >= 2) s->iterations = atoi(t[1].c_str());
                else if (t[0] == "DEPTH" && t.size() >= 2) s->trace_depth = atoi(t[1].c_str());
                else if (t[0] == "FILE" && t.size() >= 2) snprintf(s->image_name, sizeof s->image_name, "%s", t[1].c_str());
            }
            fp.next(line);
            while (!line.empty() && fp.good()) {
                std::vector<std::string> t = tokens_of(line);
                if (t.size() >= 4) {
                    if (t[0] == "EYE") camera.position = P(vec3_of(t));
                    else if (t[0] == "LOOKAT") camera.lookAt = P(vec3_of(t));
                    else if (t[0] == "UP") camera.up = P(vec3_of(t));
                }
                fp.next(line);
            }
            float yscaled = std::tan(fovy * (PI_F / 180));
            float xscaled = (yscaled * camera.resolution[0]) / camera.resolution[1];
            float fovx = (std::atan(xscaled) * 180) / PI_F;
            camera.fov[0] = fovx; camera.fov[1] = fovy;
            // scene.cpp:138 computes `right` from `view` BEFORE view is set (:142): cross(0, up) normalised = NaN
            camera.right = P(normalize(cross(U(camera.view), U(camera.up))));
            camera.pixelLength[0] = 2 * xscaled / (float)camera.resolution[0];
            camera.pixelLength[1] = 2 * yscaled / (float)camera.resolution[1];
            camera.view = P(normalize(sub(U(camera.lookAt), U(camera.position))));
            have_cam = true;
        }
    }
    if (!have_cam || b.mats.empty()) { snprintf(g_err, sizeof g_err, "scene %s has no CAMERA or no MATERIAL", path); goto fail; }
    for (size_t i = 0; i < b.geoms.size(); ++i)
        if (b.geoms[i].materialid < 0 || b.geoms[i].materialid >= (int)b.mats.size()) {
            snprintf(g_err, sizeof g_err, "OBJECT %zu links material %d of %zu", i, b.geoms[i].materialid, b.mats.size()); goto fail;
        }
    s->camera = s->camera_loaded;
    orbit_recompute(s->camera);
    s->num_geoms = (int32_t)b.geoms.size(); s->num_materials = (int32_t)b.mats.size();
    s->num_triangles = (int32_t)b.tris.size(); s->num_meshes = (int32_t)b.meshes.size();
    s->geoms = (pt_geom *)malloc(sizeof(pt_geom) * (b.geoms.size() + 1));
    s->materials = (pt_material *)malloc(sizeof(pt_material) * (b.mats.size() + 1));
    s->triangles = (pt_triangle *)malloc(sizeof(pt_triangle) * (b.tris.size() + 1));
    s->meshes = (pt_mesh *)malloc(sizeof(pt_mesh) * (b.meshes.size() + 1));
    if (!b.geoms.empty()) memcpy(s->geoms, b.geoms.data(), sizeof(pt_geom) * b.geoms.size());
    memcpy(s->materials, b.mats.data(), sizeof(pt_material) * b.mats.size());
    if (!b.tris.empty()) memcpy(s->triangles, b.tris.data(), sizeof(pt_triangle) * b.tris.size());
    if (!b.meshes.empty()) memcpy(s->meshes, b.meshes.data(), sizeof(pt_mesh) * b.meshes.size());
    g_err[0] = 0;
    return s;
fail:
    free(s);
    return NULL;
}

void pth_free_scene(pth_scene *s) {
    if (!s) return;
    free(s->geoms); free(s->materials); free(s->triangles); free(s->meshes); free(s);
}

// main.cpp:78-99 + image.cpp:22-39
void pth_image_to_rgb8(const float *image_sum, int w, int h, float samples, uint8_t *rgb) {
    for (int x = 0; x < w; ++x)
        for (int y = 0; y < h; ++y) {
            const int index = x + (y * w);
            const int i = y * w + (w - 1 - x);                     // img.setPixel(width - 1 - x, y, pix / samples)
            for (int c = 0; c < 3; ++c) {
                float v = image_sum[3 * index + c] / samples;
                v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);          // glm::clamp(x, 0, 1) = min(max(x, 0), 1)
                rgb[3 * i + c] = (unsigned char)(v * 255.f);
            }
        }
}

static uint32_t crc_table[256];
static void crc_init(void) {
    for (uint32_t n = 0; n < 256; ++n) {
        uint32_t c = n;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
        crc_table[n] = c;
    }
}
static uint32_t crc_update(uint32_t c, const uint8_t *p, size_t n) {
    for (size_t i = 0; i < n; ++i) c = crc_table[(c ^ p[i]) & 0xff] ^ (c >> 8);
    return c;
}
static void put32(std::vector<uint8_t> &v, uint32_t x) { for (int s = 24; s >= 0; s -= 8) v.push_back((uint8_t)(x >> s)); }
static void chunk(std::vector<uint8_t> &out, const char *type, const std::vector<uint8_t> &data) {
    put32(out, (uint32_t)data.size());
    std::vector<uint8_t> td(type, type + 4);
    td.insert(td.end(), data.begin(), data.end());
    out.insert(out.end(), td.begin(), td.end());
    put32(out, crc_update(0xffffffffu, td.data(), td.size()) ^ 0xffffffffu);
}

int pth_write_png(const char *path, const uint8_t *rgb, int w, int h) {
    crc_init();
    std::vector<uint8_t> raw;                                       // filter byte 0 + scanline
    for (int y = 0; y < h; ++y) { raw.push_back(0); raw.insert(raw.end(), rgb + (size_t)3 * w * y, rgb + (size_t)3 * w * (y + 1)); }
    std::vector<uint8_t> z;                                         // zlib stream of stored blocks
    z.push_back(0x78); z.push_back(0x01);
    size_t pos = 0;
    while (pos < raw.size()) {
        size_t n = raw.size() - pos; if (n > 65535) n = 65535;
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xff)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xff)); z.push_back((uint8_t)((~n >> 8) & 0xff));
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        pos += n;
    }
    uint32_t a = 1, b2 = 0;
    for (size_t i = 0; i < raw.size(); ++i) { a = (a + raw[i]) % 65521u; b2 = (b2 + a) % 65521u; }
    put32(z, (b2 << 16) | a);
    std::vector<uint8_t> out;
    const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    out.insert(out.end(), sig, sig + 8);
    std::vector<uint8_t> ihdr;
    put32(ihdr, (uint32_t)w); put32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr); chunk(out, "IDAT", z); chunk(out, "IEND", std::vector<uint8_t>());
    FILE *f = fopen(path, "wb");
    if (!f) { snprintf(g_err, sizeof g_err, "cannot write %s", path); return -1; }
    fwrite(out.data(), 1, out.size(), f);
    fclose(f);
    return 0;
}

int pth_write_pfm(const char *path, const float *image_sum, int w, int h, float samples) {
    FILE *f = fopen(path, "wb");
    if (!f) { snprintf(g_err, sizeof g_err, "cannot write %s", path); return -1; }
    fprintf(f, "PF\n%d %d\n-1.0\n", w, h);
    std::vector<float> row((size_t)3 * w);
    for (int y = h - 1; y >= 0; --y) {                              // PFM rows run bottom to top
        for (int x = 0; x < 3 * w; ++x) row[x] = image_sum[(size_t)3 * w * y + x] / samples;
        fwrite(row.data(), sizeof(float), row.size(), f);
    }
    fclose(f);
    return 0;
}

// the raw running sum back from a PFM written with samples = 1 (ptbench --save-sum): what pt_set_image resumes from
int pth_read_pfm(const char *path, float *image_sum, int w, int h) {
    FILE *f = fopen(path, "rb");
    if (!f) { snprintf(g_err, sizeof g_err, "cannot read %s", path); return -1; }
    char magic[3] = {0, 0, 0};
    int fw = 0, fh = 0;
    float scale = 0.0f;
    if (fscanf(f, "%2s %d %d %f", magic, &fw, &fh, &scale) != 4 || strcmp(magic, "PF") != 0 || fgetc(f) != '\n') {
        fclose(f); snprintf(g_err, sizeof g_err, "%s: not a colour PFM", path); return -1;
    }
    if (fw != w || fh != h || !(scale < 0.0f)) {
        fclose(f); snprintf(g_err, sizeof g_err, "%s: %dx%d scale %g, expected %dx%d little-endian", path, fw, fh, scale, w, h); return -1;
    }
    for (int y = h - 1; y >= 0; --y)
        if (fread(image_sum + (size_t)3 * w * y, sizeof(float), (size_t)3 * w, f) != (size_t)3 * w) {
            fclose(f); snprintf(g_err, sizeof g_err, "%s: truncated", path); return -1;
        }
    fclose(f);
    if (scale != -1.0f)
        for (size_t k = 0; k < (size_t)3 * w * h; ++k) image_sum[k] *= -scale;
    return 0;
}

}  // extern "C"
