// mesh.cpp — native host implementations of the two mesh-extraction utilities reconstruct.py reaches
// (SURVEY.md 8(f-1)), behind a C ABI (libslice3d_mesh.so, built with g++):
//
//   MISE  — multiresolution iso-surface extraction scheduler, replaces
//           reg_slices/src_convonet/utils/libmise/mise.pyx:33-368 (query / update / to_dense)
//   marching cubes — replaces libmcubes.marching_cubes (libmcubes/pywrapper.cpp:90-127 driving
//           marchingcubes.h:23-193): same cell traversal, vertex ownership and interpolation
//           orientation, so vertices and faces come out in the reference's order.
//
// Host C++ for now; the device version (active-voxel refinement + classify/scan/emit) is a later row.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <unordered_map>
#include <vector>

extern "C" {

// =============================================================================================
// MISE
// =============================================================================================
struct MiseVoxel {
    int x, y, z;
    int level;
    bool leaf;
    int64_t child[8];
};
struct MisePoint {
    int x, y, z;
    double value;
    bool known;
};
struct Mise {
    int res0, depth, vs0, res;
    double thr;
    std::vector<MiseVoxel> vox;
    std::vector<MisePoint> pts;
    // (x,y,z) at full resolution -> pts index: a dense (res+1)^3 table when that is small enough, else a hash map
    std::vector<int32_t> dense;
    std::unordered_map<int64_t, int64_t> index;
    // Refinement state.  The reference recomputes, on every update(), for every leaf the OR of
    //   f(p) = (value >= thr) | (value <= thr) << 1   over the known points p in the leaf's closed box
    // and splits the leaves that hold both bits (mise.pyx:182-232): O(rounds x points).  The same flags are kept
    // incrementally here: a newly known point is OR-ed into the <= 8 leaves around it, a new child starts from the known
    // points inside its box, and a leaf becomes a split candidate the moment its flags reach 3.  Splits still happen in
    // ascending voxel order among the leaves that existed when update() was called, so voxel / point numbering — and
    // with it query() order — is the reference's.
    std::vector<uint8_t> flags;
    std::vector<int64_t> cand;       // leaves whose flags reached 3 since the last refine
    bool need_full = false;          // a point was updated twice: rebuild the flags from scratch
    size_t scan_from = 0;            // every point before this index is known

    int64_t key(int x, int y, int z) const {
        const int64_t r = res + 1;
        return (r * x + y) * r + z;
    }
    int64_t find_point(int x, int y, int z) const {
        if (!dense.empty()) return dense[(size_t)key(x, y, z)];
        auto it = index.find(key(x, y, z));
        return it == index.end() ? -1 : it->second;
    }
    void add_point(int x, int y, int z) {
        if (!dense.empty()) dense[(size_t)key(x, y, z)] = (int32_t)pts.size();
        else index[key(x, y, z)] = (int64_t)pts.size();
        pts.push_back(MisePoint{x, y, z, 0.0, false});
    }
    // leaf voxel containing integer location (x,y,z), or -1 outside the grid (mise.pyx:283-348)
    int64_t leaf_at(int x, int y, int z) const {
        if (x < 0 || y < 0 || z < 0 || x >= res || y >= res || z >= res) return -1;
        int64_t idx = ((int64_t)(x >> depth) * res0 + (y >> depth)) * res0 + (z >> depth);
        int rx = x & (vs0 - 1), ry = y & (vs0 - 1), rz = z & (vs0 - 1);
        int size = vs0;
        while (!vox[idx].leaf) {
            size >>= 1;
            const int ox = rx >= size, oy = ry >= size, oz = rz >= size;
            idx = vox[idx].child[ox * 4 + oy * 2 + oz];
            rx -= ox * size; ry -= oy * size; rz -= oz * size;
        }
        return idx;
    }
    uint8_t point_flag(const MisePoint& p) const { return (p.value >= thr ? 1 : 0) | (p.value <= thr ? 2 : 0); }
    void touch(const MisePoint& p) {   // OR a known point into the leaves around it
        const uint8_t f = point_flag(p);
        for (int i = -1; i < 1; ++i)
            for (int j = -1; j < 1; ++j)
                for (int k = -1; k < 1; ++k) {
                    const int64_t v = leaf_at(p.x + i, p.y + j, p.z + k);
                    if (v < 0) continue;
                    const uint8_t old = flags[v];
                    flags[v] = old | f;
                    if (old != 3 && flags[v] == 3 && vox[v].level != depth) cand.push_back(v);
                }
    }
    void split(int64_t idx) {
        const MiseVoxel v = vox[idx];
        const int lvl = v.level + 1, size = 1 << (depth - lvl);
        vox[idx].leaf = false;
        const int64_t first = (int64_t)vox.size();
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 2; ++k) {
                    vox[idx].child[i * 4 + j * 2 + k] = (int64_t)vox.size();
                    MiseVoxel c{v.x + i * size, v.y + j * size, v.z + k * size, lvl, true, {0, 0, 0, 0, 0, 0, 0, 0}};
                    vox.push_back(c);
                }
        // the children start from the known points already inside their boxes (parent corners, hanging points of finer
        // neighbours); the points this split creates below are unknown and do not count
        flags.resize(vox.size(), 0);
        for (int64_t c = first; c < first + 8; ++c) {
            const MiseVoxel& cv = vox[c];
            uint8_t f = 0;
            for (int i = 0; i <= size && f != 3; ++i)
                for (int j = 0; j <= size && f != 3; ++j)
                    for (int k = 0; k <= size; ++k) {
                        const int64_t pi = find_point(cv.x + i, cv.y + j, cv.z + k);
                        if (pi >= 0 && pts[pi].known) f |= point_flag(pts[pi]);
                    }
            flags[c] = f;
            if (f == 3 && lvl != depth) cand.push_back(c);   // split on the NEXT update, as the reference would
        }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                for (int k = 0; k < 3; ++k) {
                    const int x = v.x + i * size, y = v.y + j * size, z = v.z + k * size;
                    if (find_point(x, y, z) < 0) add_point(x, y, z);
                }
    }
    void refine() {
        if (need_full) {   // rebuild from scratch, the reference's way
            flags.assign(vox.size(), 0);
            cand.clear();
            for (const MisePoint& p : pts)
                if (p.known) {
                    const uint8_t f = point_flag(p);
                    for (int i = -1; i < 1; ++i)
                        for (int j = -1; j < 1; ++j)
                            for (int k = -1; k < 1; ++k) {
                                const int64_t v = leaf_at(p.x + i, p.y + j, p.z + k);
                                if (v >= 0) flags[v] |= f;
                            }
                }
            for (size_t v = 0; v < vox.size(); ++v)
                if (vox[v].leaf && vox[v].level != depth && flags[v] == 3) cand.push_back((int64_t)v);
            need_full = false;
        }
        std::vector<int64_t> todo;
        todo.swap(cand);   // split() refills cand with next round's candidates
        std::sort(todo.begin(), todo.end());
        todo.erase(std::unique(todo.begin(), todo.end()), todo.end());
        for (const int64_t v : todo)
            if (vox[v].leaf && vox[v].level != depth && flags[v] == 3) split(v);
    }
};

void* s3d_mise_create(int resolution0, int depth, double threshold) {
    if (resolution0 < 1 || depth < 0 || depth > 12) return nullptr;
    Mise* m = new Mise();
    m->res0 = resolution0; m->depth = depth; m->thr = threshold;
    m->vs0 = 1 << depth; m->res = resolution0 * m->vs0;
    {
        const int64_t r = (int64_t)m->res + 1;
        if (r * r * r <= (int64_t)300 * 1000 * 1000) m->dense.assign((size_t)(r * r * r), -1);
    }
    m->vox.reserve((size_t)resolution0 * resolution0 * resolution0);
    for (int i = 0; i < resolution0; ++i)
        for (int j = 0; j < resolution0; ++j)
            for (int k = 0; k < resolution0; ++k)
                m->vox.push_back(MiseVoxel{i * m->vs0, j * m->vs0, k * m->vs0, 0, true, {0, 0, 0, 0, 0, 0, 0, 0}});
    for (int i = 0; i <= resolution0; ++i)
        for (int j = 0; j <= resolution0; ++j)
            for (int k = 0; k <= resolution0; ++k) m->add_point(i * m->vs0, j * m->vs0, k * m->vs0);
    m->flags.assign(m->vox.size(), 0);
    return m;
}
void s3d_mise_destroy(void* h) { delete (Mise*)h; }
int s3d_mise_resolution(void* h) { return ((Mise*)h)->res; }
static void mise_skip_known(Mise* m) {
    while (m->scan_from < m->pts.size() && m->pts[m->scan_from].known) ++m->scan_from;
}
long s3d_mise_query_count(void* h) {
    Mise* m = (Mise*)h;
    mise_skip_known(m);
    long n = 0;
    for (size_t i = m->scan_from; i < m->pts.size(); ++i) n += !m->pts[i].known;
    return n;
}
// out: (n,3) int64, unknown points in insertion order (mise.pyx:106-129)
void s3d_mise_query(void* h, int64_t* out) {
    Mise* m = (Mise*)h;
    mise_skip_known(m);
    for (size_t i = m->scan_from; i < m->pts.size(); ++i) {
        const MisePoint& p = m->pts[i];
        if (!p.known) {
            *out++ = p.x; *out++ = p.y; *out++ = p.z;
        }
    }
}
// returns 0, or -1 if a point is not a grid point (the reference raises ValueError)
int s3d_mise_update(void* h, const int64_t* points, const double* values, long n) {
    Mise* m = (Mise*)h;
    const int64_t r = m->res;
    for (long i = 0; i < n; ++i) {
        const int64_t x = points[3 * i], y = points[3 * i + 1], z = points[3 * i + 2];
        if (x < 0 || y < 0 || z < 0 || x > r || y > r || z > r) return -1;
        const int64_t idx = m->find_point((int)x, (int)y, (int)z);
        if (idx < 0) return -1;
        MisePoint& p = m->pts[idx];
        if (p.known) m->need_full = true;   // re-valued point: its old flag bits may no longer hold
        p.value = values[i];
        p.known = true;
        if (!m->need_full) m->touch(p);
    }
    m->refine();
    return 0;
}
// out: (res+1)^3 doubles; unknown entries forward-filled along x, then y, then z (mise.pyx:131-163)
void s3d_mise_to_dense(void* h, double* out) {
    Mise* m = (Mise*)h;
    const int64_t r = m->res + 1, total = r * r * r;
    for (int64_t i = 0; i < total; ++i) out[i] = NAN;
    for (const MisePoint& p : m->pts) out[(r * p.x + p.y) * r + p.z] = p.value;
    for (int64_t i = 1; i < r; ++i)
        for (int64_t j = 0; j < r; ++j)
            for (int64_t k = 0; k < r; ++k)
                if (isnan(out[(r * i + j) * r + k])) out[(r * i + j) * r + k] = out[(r * (i - 1) + j) * r + k];
    for (int64_t i = 0; i < r; ++i)
        for (int64_t j = 1; j < r; ++j)
            for (int64_t k = 0; k < r; ++k)
                if (isnan(out[(r * i + j) * r + k])) out[(r * i + j) * r + k] = out[(r * i + j - 1) * r + k];
    for (int64_t i = 0; i < r; ++i)
        for (int64_t j = 0; j < r; ++j)
            for (int64_t k = 1; k < r; ++k)
                if (isnan(out[(r * i + j) * r + k])) out[(r * i + j) * r + k] = out[(r * i + j) * r + k - 1];
}
long s3d_mise_num_points(void* h) { return (long)((Mise*)h)->pts.size(); }

// =============================================================================================
// marching cubes
// =============================================================================================
static const char* const kCases[256] = {
#include "mc_cases.inc"
};

// cube corners in units of one cell
static const int kCorner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
// edge e runs from corner a to corner b along `axis` (interpolation keeps this orientation);
// a cell OWNS edges 6, 5, 10 (the three meeting at its far corner).  Any other edge was created by the
// neighbour cell at offset (di,dj,dk) in its slot `slot`, unless this cell lies on the matching boundary.
struct McEdge {
    int a, b, axis;
    int di, dj, dk, slot;   // owner cell = (i+di, j+dj, k+dk); owned edges have di=dj=dk=0
};
static const McEdge kEdge[12] = {
    {0, 1, 0, 0, -1, -1, 0},  {1, 2, 1, 0, 0, -1, 1},  {2, 3, 0, 0, 0, -1, 0},  {3, 0, 1, -1, 0, -1, 1},
    {4, 5, 0, 0, -1, 0, 0},   {5, 6, 1, 0, 0, 0, 1},   {6, 7, 0, 0, 0, 0, 0},   {7, 4, 1, -1, 0, 0, 1},
    {0, 4, 2, -1, -1, 0, 2},  {1, 5, 2, 0, -1, 0, 2},  {2, 6, 2, 0, 0, 0, 2},   {3, 7, 2, -1, 0, 0, 2}};
static const int kVisit[12] = {6, 5, 10, 0, 1, 2, 3, 4, 7, 8, 9, 11};   // vertex creation order per cell

struct McResult {
    std::vector<double> verts;
    std::vector<int64_t> tris;
};

// grid: (nx,ny,nz) doubles, C order.  Vertex coordinates are in index units + 0.5 (cell-centre
// convention of libmcubes, undone by the caller: reconstruct.py:199-201).
void* s3d_mc_run(const double* grid, int nx, int ny, int nz, double iso) {
    McResult* R = new McResult();
    if (nx < 2 || ny < 2 || nz < 2) return R;
    const int cx = nx - 1, cy = ny - 1, cz = nz - 1;
    uint16_t edge_mask[256];
    for (int c = 0; c < 256; ++c) {
        uint16_t m = 0;
        for (const char* p = kCases[c]; *p; ++p) m |= (uint16_t)(1u << (*p <= '9' ? *p - '0' : *p - 'a' + 10));
        edge_mask[c] = m;
    }
    std::vector<int64_t> shared((size_t)2 * cy * cz * 3, -1);   // two i-slabs of owned-edge vertex ids
    auto slot = [&](int i, int j, int k, int s) -> int64_t& { return shared[(((size_t)(i & 1) * cy + j) * cz + k) * 3 + s]; };
    auto at = [&](int x, int y, int z) { return grid[((size_t)x * ny + y) * nz + z]; };
    for (int i = 0; i < cx; ++i)
        for (int j = 0; j < cy; ++j)
            for (int k = 0; k < cz; ++k) {
                double v[8];
                unsigned cube = 0;
                for (int c = 0; c < 8; ++c) {
                    v[c] = at(i + kCorner[c][0], j + kCorner[c][1], k + kCorner[c][2]);
                    if (v[c] <= iso) cube |= 1u << c;
                }
                const uint16_t em = edge_mask[cube];
                if (!em) continue;
                int64_t id[12];
                for (int t = 0; t < 12; ++t) {
                    const int e = kVisit[t];
                    if (!(em & (1u << e))) continue;
                    const McEdge& E = kEdge[e];
                    const bool boundary = (E.di && i == 0) || (E.dj && j == 0) || (E.dk && k == 0);
                    const bool owned = !E.di && !E.dj && !E.dk;
                    if (!owned && !boundary) {
                        id[e] = slot(i + E.di, j + E.dj, k + E.dk, E.slot);
                        continue;
                    }
                    id[e] = (int64_t)(R->verts.size() / 3);
                    if (owned) slot(i, j, k, E.slot) = id[e];
                    double p[3] = {i + kCorner[E.a][0] + 0.5, j + kCorner[E.a][1] + 0.5, k + kCorner[E.a][2] + 0.5};
                    const double x1 = p[E.axis], x2 = (E.axis == 0 ? i : E.axis == 1 ? j : k) + kCorner[E.b][E.axis] + 0.5;
                    const double f1 = v[E.a], f2 = v[E.b];
                    p[E.axis] = f2 == f1 ? (x2 + x1) / 2 : (x2 - x1) * (iso - f1) / (f2 - f1) + x1;
                    R->verts.push_back(p[0]); R->verts.push_back(p[1]); R->verts.push_back(p[2]);
                }
                for (const char* p = kCases[cube]; *p; ++p) R->tris.push_back(id[*p <= '9' ? *p - '0' : *p - 'a' + 10]);
            }
    return R;
}
long s3d_mc_num_vertices(void* h) { return (long)(((McResult*)h)->verts.size() / 3); }
long s3d_mc_num_triangles(void* h) { return (long)(((McResult*)h)->tris.size() / 3); }
void s3d_mc_copy(void* h, double* verts, int64_t* tris) {
    McResult* R = (McResult*)h;
    if (!R->verts.empty()) memcpy(verts, R->verts.data(), R->verts.size() * sizeof(double));
    if (!R->tris.empty()) memcpy(tris, R->tris.data(), R->tris.size() * sizeof(int64_t));
}
void s3d_mc_free(void* h) { delete (McResult*)h; }

}  // extern "C"
