// pt_h_scene.hpp -- scene upload: cull records, per-triangle spheres, hierarchies, camera masks
// (one of the host-side headers of libptmi355.so, included by ptmi355.hip -- the only translation unit -- in dependency order)
#pragma once

namespace {

int upload_tri_bounds(const pt_scene_desc *d, double Rorigin);

// per-primitive cull boxes (pt_cull.hpp) for the scene of R.desc as seen from camera `cam`: the |origin|_1 bound
// they are derived for covers the scene and the camera; a camera that later moves beyond it gets new boxes
int upload_cull(const pt_scene_desc *d, const pt_camera &cam) {
    const int n = d->num_geoms;
    std::vector<const float *> inv((size_t)std::max(1, n));
    std::vector<char> sph((size_t)std::max(1, n)), skip((size_t)std::max(1, n));
    for (int i = 0; i < n; ++i) {
        inv[(size_t)i] = &d->geoms[i].inverseTransform.m[0][0];
        sph[(size_t)i] = d->geoms[i].type == PT_SPHERE;
        skip[(size_t)i] = d->geoms[i].type == PT_TRIANGLE_MESH;
    }
    const double eye[3] = {(double)cam.position.x, (double)cam.position.y, (double)cam.position.z};
    std::vector<ptcull::Box> boxes;
    std::vector<double> pts(eye, eye + 3);
    // triangle meshes are world-space soups: their vertices bound where rays can start as well
    for (int t = 0; t < d->num_triangles; ++t) {
        const pt_vec3 *v = &d->triangles[t].v0;
        double m = 0.0;
        for (int k = 0; k < 3; ++k) m = std::max(m, (double)std::fabs(v[k].x) + std::fabs(v[k].y) + std::fabs(v[k].z));
        if (t == 0 || m > pts[3]) { if (pts.size() < 6) pts.resize(6, 0.0); pts[3] = m; pts[4] = 0.0; pts[5] = 0.0; }
    }
    R.scene.rmax = ptcull::make_boxes(inv.data(), reinterpret_cast<const bool *>(sph.data()),
                                      reinterpret_cast<const bool *>(skip.data()), n, pts.data(), (int)(pts.size() / 3), boxes);
    std::vector<float> rec((size_t)std::max(1, n) * CULL_WORDS, 0.0f);
    for (int i = 0; i < n; ++i) {
        float *r = rec.data() + (size_t)i * CULL_WORDS;
        for (int k = 0; k < 3; ++k) ptcull::centre_half(boxes[(size_t)i].lo[k], boxes[(size_t)i].hi[k], r[2 * k], r[2 * k + 1]);
        int ax = 3;
        if (d->geoms[i].type == PT_CUBE && !pt_experiment("PTMI355_NO_AXIS_REJECT"))
            ax = ptcull::reject_row(&d->geoms[i].inverseTransform.m[0][0], &r[7]);       // words 7..10: the row
        if (ax == 4 && pt_experiment("PTMI355_NO_ROW_REJECT")) ax = 3;
        const int tw = d->geoms[i].type | (ax << 8);
        memcpy(&r[6], &tw, 4);
        const uint32_t ent = ((uint32_t)d->geoms[i].type << 7) | ((uint32_t)i << 9);     // the candidate ring's entry for this geom
        memcpy(&r[11], &ent, 4);
    }
    if (!R.d_cull) HIPCHK(hipMalloc(&R.d_cull, rec.size() * 4));
    HIPCHK(hipMemcpyAsync(R.d_cull, rec.data(), rec.size() * 4, hipMemcpyHostToDevice, R.stream));
    HIPCHK(hipStreamSynchronize(R.stream));            // `rec` is pageable host memory about to go out of scope
    R.scene.cull = R.d_cull;
    R.cull_eye_reach = std::fabs(eye[0]) + std::fabs(eye[1]) + std::fabs(eye[2]);
    return upload_tri_bounds(d, (double)R.scene.rmax);       // the bounding spheres hold for origins within the same bound
}

// Every-triangle loop (MESH_TILES), stage 1 (pt_kernels.hpp: mesh_sweep): per triangle a sphere {c, Rs} such that a
// ray whose line passes c at more than Rs cannot be ACCEPTED for the triangle.  Derivation (u = 2^-24):
//   * the spec accepts a hit only if the point it reports, P_k = fl(o_k + fl(d_k tz)), lies in the triangle's box
//     [fl(lo_k - pad), fl(hi_k + pad)] (tri_point_ok, evaluated on these very floats): |P - c| <= R0 = half diagonal
//     of that box, c its centre;
//   * P_k differs from the line's point o_k + d_k tz by at most u |d_k tz| + u |P_k| <= 2u (|o_k| + |P_k|): the line
//     passes P within delta = 2 sqrt3 u (R + |c|_inf + R0)  (non-wild rays: |o|_1 <= R);
//   * the kernel's q = c x d' - fl(o x d') (fused multiply-adds, d' = d / |d| up to 2^-20) is off by at most
//     2^-21 (|o|_inf + |c|_inf) |d'|_inf per component, and its |q|^2 and the compare lose another 2^-20 relative;
//     rounding c to float moves it by u |c|_inf.
//   Rs = (R0 + 4 (delta + 2^-19 (R + |c|_inf + R0))) (1 + 2^-10), squared and rounded up.  Non-finite triangles get
//   Rs^2 = +inf (always a candidate: the exact test decides, and it never accepts them).  Each mesh's entries are
//   padded to a multiple of four with Rs^2 = -1 (no ray is a candidate: |q|^2 > -1).
// one mesh: `count` triangles -> ((count + 3) & ~3) x {cx, cy, cz, Rs^2}
void make_tri_bounds(const pt_triangle *tris, int count, double Rorigin, float *out) {
    const double u = 0x1p-24;
    const float pad = ptbvh::spec_pad(reinterpret_cast<const float *>(tris), count);
    const int n4 = (count + 3) & ~3;
    for (int i = 0; i < n4; ++i) {
        float *o = out + (size_t)i * 4;
        if (i >= count) { o[0] = o[1] = o[2] = 0.0f; o[3] = -1.0f; continue; }
        const pt_triangle &t = tris[i];
        const float v0[3] = {t.v0.x, t.v0.y, t.v0.z};
        const float e1[3] = {t.v1.x - t.v0.x, t.v1.y - t.v0.y, t.v1.z - t.v0.z};      // the device record's e1, e2
        const float e2[3] = {t.v2.x - t.v0.x, t.v2.y - t.v0.y, t.v2.z - t.v0.z};
        double c[3], h2 = 0.0, cinf = 0.0;
        bool fin = true;
        for (int a = 0; a < 3; ++a) {
            const float x1 = v0[a] + e1[a], x2 = v0[a] + e2[a];                       // tri_point_ok's own floats
            const float lo = std::fmin(v0[a], std::fmin(x1, x2)) - pad, hi = std::fmax(v0[a], std::fmax(x1, x2)) + pad;
            if (!std::isfinite(lo) || !std::isfinite(hi)) fin = false;
            c[a] = 0.5 * ((double)lo + (double)hi);
            const double h = 0.5 * ((double)hi - (double)lo);
            h2 += h * h;
            cinf = std::fmax(cinf, std::fabs(c[a]));
        }
        if (!fin || !std::isfinite(Rorigin)) { o[0] = o[1] = o[2] = 0.0f; o[3] = INFINITY; continue; }
        const double R0 = std::sqrt(h2);
        const double reach = Rorigin + cinf + R0;
        const double delta = 2.0 * 1.7320508075688772 * u * reach;
        const double Rs = (R0 + 4.0 * (delta + 0x1p-19 * reach)) * (1.0 + 0x1p-10);
        for (int a = 0; a < 3; ++a) o[a] = (float)c[a];
        o[3] = ptcull::round_up(Rs * Rs);
        if (!std::isfinite(o[3])) o[3] = INFINITY;
    }
}

// The same spheres as the matrix pipe reads them (pt_k_trisweep.hpp: mesh_sweep, stage 1): in the mesh's own frame -- centre g
// of the finite spheres' bounding box, scaled by 1 / Rm with Rm = max (|c - g| + Rs), so that every sphere lies in the unit
// ball -- a triangle is the 32 binary16 K-slots of
//     [-cx^2 -cy^2 -cz^2 -2cxcy -2cxcz -2cycz | cx cy cz | K = |c|^2 - Rs^2 | 1]          (a term's three slots: hi, hi, lo)
// evaluated in binary64 from the sphere's floats and split into (hi, lo) pairs.  spheres: n x {cx, cy, cz, Rs^2} as
// make_tri_bounds writes them (Rs^2 = -1: padding, +inf: a non-finite triangle); records: n x 32 binary16; frame: {g, 1 / Rm}.
void make_tri_records(const float *spheres, int n, _Float16 *records, float frame[4]) {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < n; ++i) {
        const float *sp = spheres + (size_t)i * 4;
        if (!(sp[3] >= 0.0f) || !std::isfinite(sp[3])) continue;
        for (int a = 0; a < 3; ++a) { lo[a] = std::fmin(lo[a], (double)sp[a]); hi[a] = std::fmax(hi[a], (double)sp[a]); }
    }
    double g[3] = {0.0, 0.0, 0.0}, Rm = 0.0;
    if (lo[0] <= hi[0]) for (int a = 0; a < 3; ++a) g[a] = (double)(float)(0.5 * (lo[a] + hi[a]));       // (the kernel subtracts the float)
    for (int i = 0; i < n; ++i) {
        const float *sp = spheres + (size_t)i * 4;
        if (!(sp[3] >= 0.0f) || !std::isfinite(sp[3])) continue;
        const double dx = sp[0] - g[0], dy = sp[1] - g[1], dz = sp[2] - g[2];
        Rm = std::fmax(Rm, std::sqrt(dx * dx + dy * dy + dz * dz) + std::sqrt((double)sp[3]));
    }
    Rm = Rm > 0.0 ? Rm * (1.0 + 0x1p-20) : 1.0;
    float ir = (float)(1.0 / Rm);
    if (!(ir > 0.0f) || !std::isfinite(ir)) ir = 1.0f;
    frame[0] = (float)g[0]; frame[1] = (float)g[1]; frame[2] = (float)g[2]; frame[3] = ir;
    auto pair = [](double v, _Float16 &h, _Float16 &l) { h = (_Float16)(float)v; l = (_Float16)(float)(v - (double)(float)h); };
    for (int i = 0; i < n; ++i) {
        const float *sp = spheres + (size_t)i * 4;
        _Float16 *a = records + (size_t)i * 32;
        for (int k = 0; k < 32; ++k) a[k] = (_Float16)0.0f;
        a[29] = (_Float16)1.0f; a[30] = (_Float16)1.0f;                       // x M (hi, lo)
        if (!(sp[3] >= 0.0f)) { a[27] = (_Float16)30000.0f; continue; }       // padding: v = 30000 + M > 0 for every ray, wild ones included
        if (!std::isfinite(sp[3])) { a[27] = (_Float16)-30000.0f; continue; } // a non-finite triangle: everybody's candidate (never accepted)
        // (the kernel's scale is the FLOAT ir applied to float differences: the record uses the same scale)
        const double c[3] = {(sp[0] - g[0]) * (double)ir, (sp[1] - g[1]) * (double)ir, (sp[2] - g[2]) * (double)ir};
        const double v[9] = {-c[0] * c[0], -c[1] * c[1], -c[2] * c[2], -2.0 * c[0] * c[1], -2.0 * c[0] * c[2], -2.0 * c[1] * c[2], c[0], c[1], c[2]};
        for (int t = 0; t < 9; ++t) { _Float16 h, l; pair(v[t], h, l); a[3 * t] = h; a[3 * t + 1] = h; a[3 * t + 2] = l; }
        _Float16 h, l;
        pair((c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) - (double)sp[3] * (double)ir * (double)ir, h, l);
        a[27] = h; a[28] = l;
    }
}

int upload_tri_bounds(const pt_scene_desc *d, double Rorigin) {
    if (R.mesh_mode != MESH_TILES || d->num_meshes <= 0) return PT_OK;
    size_t recs = 0;
    for (int k = 0; k < d->num_meshes; ++k) recs += (size_t)((d->meshes[k].triangle_count + 63) & ~63);
    std::vector<_Float16> tr(std::max<size_t>(recs, 64) * 32, (_Float16)0.0f);
    std::vector<float> sph;
    size_t off = 0;
    for (int k = 0; k < d->num_meshes; ++k) {
        const pt_mesh &m = d->meshes[k];
        const size_t n64 = (size_t)((m.triangle_count + 63) & ~63);
        sph.assign(n64 * 4, 0.0f);
        make_tri_bounds(d->triangles + m.first_triangle, m.triangle_count, Rorigin, sph.data());
        for (size_t i = (size_t)((m.triangle_count + 3) & ~3); i < n64; ++i) sph[i * 4 + 3] = -1.0f;
        float frame[4];
        make_tri_records(sph.data(), (int)n64, tr.data() + off * 32, frame);
        // the mesh's frame rides in its geom record (words G_INV + 7 .. + 10; a mesh's matrices are never read)
        float *r = R.grec_frames.data() + (size_t)m.geom_index * 4;
        for (int a = 0; a < 4; ++a) r[a] = frame[a];
        off += n64;
    }
    if (!R.d_tri_bound || R.tri_bound_words < tr.size() / 2) {
        if (R.d_tri_bound) { HIPCHK(hipStreamSynchronize(R.stream)); (void)hipFree(R.d_tri_bound); R.d_tri_bound = nullptr; }
        HIPCHK(hipMalloc(&R.d_tri_bound, tr.size() * 2));
        R.tri_bound_words = tr.size() / 2;
    }
    HIPCHK(hipMemcpyAsync(R.d_tri_bound, tr.data(), tr.size() * 2, hipMemcpyHostToDevice, R.stream));
    // ... and the frames into the device's geom records
    for (int k = 0; k < d->num_meshes; ++k) {
        const int gi = d->meshes[k].geom_index;
        HIPCHK(hipMemcpyAsync(R.d_geoms + (size_t)gi * ptd::GEOM_WORDS + ptd::G_INV + 7, R.grec_frames.data() + (size_t)gi * 4, 16, hipMemcpyHostToDevice, R.stream));
    }
    HIPCHK(hipStreamSynchronize(R.stream));            // `tr` is pageable host memory about to go out of scope
    R.scene.tri_rec = R.d_tri_bound;
    return PT_OK;
}

}  // namespace

namespace one {

// PT_MESH_BVH: one tree per mesh (pt_bvh.hpp), all trees in one node buffer; the leaf-ordered copies
// of the triangle records carry the original index in word 9.  Geom record words 2/3 of a mesh
// become (root node, triangle count).
static int upload_bvh(const pt_scene_desc *d, std::vector<float> &grec) {
    std::vector<float> nodes, btris, tops;
    std::vector<int32_t> mesh_list;                          // {geom, root record, triangles, 0} in geom order
    float prune = 0.0f;
    int guard = 1;
    R.bvh_info = pt_bvh_info{};
    R.mesh_grids.clear();
    std::vector<int> by_geom((size_t)d->num_meshes);
    for (int k = 0; k < d->num_meshes; ++k) by_geom[(size_t)k] = k;
    std::sort(by_geom.begin(), by_geom.end(), [&](int x, int y) { return d->meshes[x].geom_index < d->meshes[y].geom_index; });
    for (int kk = 0; kk < d->num_meshes; ++kk) {
        const int k = by_geom[(size_t)kk];
        const pt_mesh &m = d->meshes[k];
        if (kk > 0 && d->meshes[by_geom[(size_t)kk - 1]].geom_index == m.geom_index)
            return fail(PT_ERR_INVALID, "pt_init: geom %d owns more than one mesh", m.geom_index);
        ptbvh::Tree tree;
        ptbvh::build(reinterpret_cast<const float *>(d->triangles + m.first_triangle), m.triangle_count, tree, (double)R.scene.rmax);
        const int root = (int)(nodes.size() / BVH_NODE_WORDS);
        const int slot0 = (int)(btris.size() / TRI_WORDS);
        if ((int64_t)slot0 + m.triangle_count >= (1 << ptbvh::LINK_BITS) || tree.num_nodes() >= (1 << ptbvh::LINK_BITS))
            return fail(PT_ERR_INVALID, "pt_init: PT_MESH_BVH holds at most 2^24 triangles (record links are 24 bits)");
        for (int n = 0; n < tree.num_nodes(); ++n) {           // leaf children: slot in the tree -> slot in the shared buffer
            float *w = &tree.nodes[(size_t)n * BVH_NODE_WORDS];
            for (int c = 0; c < 2; ++c) {
                uint32_t link;
                memcpy(&link, &w[6 + c], 4);
                if ((link >> ptbvh::LINK_BITS) & ptbvh::INFO_LEAF) { link += (uint32_t)slot0; memcpy(&w[6 + c], &link, 4); }
            }
        }
        nodes.insert(nodes.end(), tree.nodes.begin(), tree.nodes.end());
        for (int s = 0; s < m.triangle_count; ++s) {
            const int32_t orig = m.first_triangle + tree.order[(size_t)s];
            const pt_triangle &t = d->triangles[orig];
            float r[TRI_WORDS] = {t.v0.x, t.v0.y, t.v0.z,
                                  t.v1.x - t.v0.x, t.v1.y - t.v0.y, t.v1.z - t.v0.z,
                                  t.v2.x - t.v0.x, t.v2.y - t.v0.y, t.v2.z - t.v0.z, 0.0f, 0.0f, 0.0f};
            memcpy(&r[9], &orig, 4);
            r[10] = tree.spec_pad;
            btris.insert(btris.end(), r, r + TRI_WORDS);
        }
        float *g = grec.data() + (size_t)m.geom_index * ptd::GEOM_WORDS;
        memcpy(&g[2], &root, 4); memcpy(&g[3], &m.triangle_count, 4);
        for (int a = 0; a < 3; ++a) { g[ptd::G_INV + a] = tree.origin[a]; g[ptd::G_INV + 3 + a] = tree.step[a]; }   // the mesh's grid
        for (int a = 0; a < 3; ++a) R.mesh_grids.push_back(tree.origin[a] - tree.step[a]);
        for (int a = 0; a < 3; ++a) R.mesh_grids.push_back(tree.origin[a] + (float)(ptbvh::GRID_MAX + 1) * tree.step[a]);
        // the first records of this tree (its most visited ones, pt_bvh.hpp: number) go into the LDS copy k_mesh keeps
        const int share = d->num_meshes <= BVH_TOP ? BVH_TOP / d->num_meshes : 0;
        const int top_cnt = std::min(share, tree.num_nodes()), top_off = (int)(tops.size() / BVH_NODE_WORDS);
        tops.insert(tops.end(), tree.nodes.begin(), tree.nodes.begin() + (size_t)top_cnt * BVH_NODE_WORDS);
        const int32_t entry[4] = {m.geom_index, root, m.triangle_count, top_off | (top_cnt << 16)};
        mesh_list.insert(mesh_list.end(), entry, entry + 4);
        prune = std::max(prune, tree.prune);
        guard = std::max(guard, tree.num_nodes() + 1);
        R.bvh_info.nodes += tree.num_nodes();
        R.bvh_info.triangles += m.triangle_count;
        R.bvh_info.depth = std::max(R.bvh_info.depth, tree.depth);
        R.bvh_info.pad = std::max(R.bvh_info.pad, tree.pad);
    }
    R.bvh_info.prune = prune;
    if (nodes.size() * 4 >= ((size_t)1 << 32))
        return fail(PT_ERR_INVALID, "pt_init: PT_MESH_BVH holds at most 4 GiB of hierarchy records (k_mesh addresses them with 32-bit offsets)");
    if (nodes.empty()) nodes.assign(BVH_NODE_WORDS, 0.0f);
    if (btris.empty()) btris.assign(TRI_WORDS, 0.0f);
    HIPCHK(hipMalloc(&R.d_bvh_nodes, nodes.size() * 4));
    HIPCHK(hipMalloc(&R.d_bvh_tris, btris.size() * 4));
    HIPCHK(hipMemcpy(R.d_bvh_nodes, nodes.data(), nodes.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(R.d_bvh_tris, btris.data(), btris.size() * 4, hipMemcpyHostToDevice));
    if (mesh_list.empty()) mesh_list.assign(4, 0);
    HIPCHK(hipMalloc((void **)&R.d_bvh_meshes, mesh_list.size() * 4));
    HIPCHK(hipMemcpy(R.d_bvh_meshes, mesh_list.data(), mesh_list.size() * 4, hipMemcpyHostToDevice));
    R.scene.bvh_meshes = R.d_bvh_meshes; R.scene.bvh_nmesh = d->num_meshes;
    R.scene.bvh_nodes = R.d_bvh_nodes; R.scene.bvh_tris = R.d_bvh_tris;
    R.scene.bvh_top_n = (int)(tops.size() / BVH_NODE_WORDS);
    if (tops.empty()) tops.assign(BVH_NODE_WORDS, 0.0f);
    HIPCHK(hipMalloc(&R.d_bvh_top, tops.size() * 4));
    HIPCHK(hipMemcpy(R.d_bvh_top, tops.data(), tops.size() * 4, hipMemcpyHostToDevice));
    R.scene.bvh_top = R.d_bvh_top;
    R.scene.bvh_prune = prune; R.scene.bvh_guard = guard;
    return PT_OK;
}

// Bounce 0, pinhole camera: which 64-pixel tiles of the local frame can see a mesh at all.  A camera ray is
// d = view - right * alpha - up * beta with alpha = pixelLength.x * (fx - W/2), beta likewise (pathtrace.cu:136-139),
// fx within half a pixel of the pixel's x.  A ray whose triangle hit the spec accepts reports a point inside that
// mesh's box grid (the hit-point test, pt_bvh.hpp), so the pixel lies inside the perspective image of the grid's
// eight corners -- computed here in double, widened by two pixels -- and every other tile can skip ray generation,
// root tests and walks in k_mesh.  No mask (nullptr) when a corner is not in front of the camera, the frame does
// not tile by 64 pixels, or a thin lens is on (then rays do not start at the eye).
// (re)build the bounce-0 candidate masks for the current camera and cull boxes: one launch on the stream, ordered
// behind whatever still reads the old masks and ahead of everything enqueued later (the buffer never moves, so
// captured graphs stay valid)
static int update_cull0() {
    if (!R.cull0_tiles) return PT_OK;
    hipLaunchKernelGGL(k_cull0_mask, dim3((R.cull0_tiles + WAVES - 1) / WAVES), dim3(BLOCK), 0, R.stream, R.scene, R.cam,
                       R.map, R.trace_depth, R.d_cull0, R.cull0_tiles);
    HIPCHK(hipGetLastError());
    return PT_OK;
}

static int update_cam_mask() {
    R.cam_mask_valid = false;
    if (R.mesh_mode != MESH_BVH || R.map.tile_pixels % TILE != 0 || R.mesh_grids.empty()) return PT_OK;
    const pt_camera &c = R.cam;
    const double M[3][3] = {{c.view.x, -c.right.x, -c.up.x}, {c.view.y, -c.right.y, -c.up.y}, {c.view.z, -c.right.z, -c.up.z}};
    const double det = M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
                       M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
    if (!(std::fabs(det) > 1e-9) || !(c.pixelLength[0] != 0.0f) || !(c.pixelLength[1] != 0.0f)) return PT_OK;
    double x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
    for (size_t m = 0; m + 6 <= R.mesh_grids.size(); m += 6) {
        for (int corner = 0; corner < 8; ++corner) {
            const double v[3] = {(double)R.mesh_grids[m + ((corner & 1) ? 3 : 0)] - c.position.x,
                                 (double)R.mesh_grids[m + 1 + ((corner & 2) ? 3 : 0)] - c.position.y,
                                 (double)R.mesh_grids[m + 2 + ((corner & 4) ? 3 : 0)] - c.position.z};
            // Cramer: (s, s*alpha, s*beta) = M^-1 v
            auto det3 = [](const double A[3][3]) {
                return A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                       A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
            };
            double sol[3];
            for (int k = 0; k < 3; ++k) {
                double A[3][3];
                for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i][j] = (j == k) ? v[i] : M[i][j];
                sol[k] = det3(A) / det;
            }
            const double reach = std::fabs(v[0]) + std::fabs(v[1]) + std::fabs(v[2]);
            if (!(sol[0] > 1e-6 * (reach + 1.0))) return PT_OK;                  // at or behind the eye: no mask
            const double fx = 0.5 * c.resolution[0] + sol[1] / sol[0] / (double)c.pixelLength[0];
            const double fy = 0.5 * c.resolution[1] + sol[2] / sol[0] / (double)c.pixelLength[1];
            if (!std::isfinite(fx) || !std::isfinite(fy)) return PT_OK;
            x0 = std::min(x0, fx); x1 = std::max(x1, fx); y0 = std::min(y0, fy); y1 = std::max(y1, fy);
        }
    }
    x0 -= 2.0; x1 += 2.0; y0 -= 2.0; y1 += 2.0;
    const uint32_t tps = (uint32_t)R.map.tile_pixels / TILE;
    std::vector<unsigned long long> mask((tps + 63) / 64, 0ull);
    for (uint32_t t = 0; t < tps; ++t) {
        bool any = false;
        for (int k = 0; k < TILE && !any; ++k) {
            const int j = (int)t * TILE + k;
            int pix = j;
            if (R.map.tile_count != 1) {                                          // pt_types.hpp: local_to_pixel
                const int ly = j / R.map.W, x = j - ly * R.map.W, ls = ly / R.map.strip_rows;
                pix = x + ((ls * R.map.tile_count + R.map.tile_index) * R.map.strip_rows + (ly - ls * R.map.strip_rows)) * R.map.W;
            }
            const int y = pix / R.map.W, x = pix - y * R.map.W;
            any = x >= x0 && x <= x1 && y >= y0 && y <= y1;
        }
        if (any) mask[t >> 6] |= 1ull << (t & 63u);
    }
    if (!R.d_cam_mask) HIPCHK(hipMalloc((void **)&R.d_cam_mask, mask.size() * sizeof(unsigned long long)));
    HIPCHK(hipMemcpy(R.d_cam_mask, mask.data(), mask.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
    R.cam_mask_valid = true;
    return PT_OK;
}

}  // namespace one
