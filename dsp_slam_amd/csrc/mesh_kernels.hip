// Mesh extraction on the device: voxel grid -> decoder forward (mlp_kernel) -> marching cubes, without the SDF volume ever
// leaving HBM.  Replaces MeshExtractor.extract_mesh_from_code's grid decode + scikit-image call
// (reference reconstruct/optimizer.py:206-223, reconstruct/utils.py:97-140).
//
// Marching cubes here is HBM/latency-bound byte work (4 B read per grid point and a few neighbours from L2, 12 B written
// per vertex and per face): one thread per grid point, three small kernels around one block-sum scan:
//   k_mc_count   per block: number of vertices (sign-changing owned edges) and triangles (cell case table)
//   k_mc_scan    exclusive scan of the block sums (one block)
//   k_mc_verts   vertex positions + the (grid point, axis) -> vertex id map
//   k_mc_faces   triangles as vertex ids through that map
// Output order is fixed (vertices by (grid point, axis), faces by (cell, table order)) so it can be compared bit for bit
// with oracle/mc_oracle.py.  The 256-case table is generated on the host at first use (mc_build_tables) by the same
// published construction the oracle restates: per cube face, join each entering edge to the next leaving edge counter-
// clockwise; chain the segments into loops; triangulate each loop avoiding diagonals that lie inside a cube face.
#include "dsp_internal.h"

#include <algorithm>
#include <array>
#include <map>
#include <vector>

namespace dsp {

// ------------------------------------------------------------------------------------------------
// host: case table
// ------------------------------------------------------------------------------------------------
namespace {

struct Off3 { int o[3]; };

inline Off3 corner_offset(int c) { return Off3{{c & 1, (c >> 1) & 1, (c >> 2) & 1}}; }
inline int corner_index(const Off3& o) { return o.o[0] | (o.o[1] << 1) | (o.o[2] << 2); }

// cube edge between adjacent corners: 4 * axis + u + 2 v, (u, v) = its offsets along the two other axes (increasing order)
int edge_id(int c0, int c1) {
    const Off3 a = corner_offset(c0), b = corner_offset(c1);
    int axis = -1;
    for (int i = 0; i < 3; ++i) if (a.o[i] != b.o[i]) axis = i;
    int rest[2], n = 0;
    for (int i = 0; i < 3; ++i) if (i != axis) rest[n++] = i;
    return 4 * axis + a.o[rest[0]] + 2 * a.o[rest[1]];
}

void edge_owner(int e, Off3& off, int& axis) {
    axis = e / 4;
    const int uv = e % 4;
    int rest[2], n = 0;
    for (int i = 0; i < 3; ++i) if (i != axis) rest[n++] = i;
    off = Off3{{0, 0, 0}};
    off.o[rest[0]] = uv & 1;
    off.o[rest[1]] = uv >> 1;
}

bool edges_share_face(int e0, int e1) {
    Off3 o0, o1;
    int a0, a1;
    edge_owner(e0, o0, a0);
    edge_owner(e1, o1, a1);
    for (int a = 0; a < 3; ++a)
        if (a != a0 && a != a1 && o0.o[a] == o1.o[a]) return true;
    return false;
}

std::vector<std::array<int, 4>> cube_faces() {   // corners counter-clockwise seen from outside
    static const int other[3][2] = {{1, 2}, {2, 0}, {0, 1}};   // (b, c) with e_a = e_b x e_c
    std::vector<std::array<int, 4>> faces;
    for (int a = 0; a < 3; ++a)
        for (int side = 0; side < 2; ++side) {
            int ring[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
            if (side == 0) { std::swap(ring[0][0], ring[3][0]); std::swap(ring[0][1], ring[3][1]); std::swap(ring[1][0], ring[2][0]); std::swap(ring[1][1], ring[2][1]); }
            std::array<int, 4> f;
            for (int i = 0; i < 4; ++i) {
                Off3 o{{0, 0, 0}};
                o.o[a] = side; o.o[other[a][0]] = ring[i][0]; o.o[other[a][1]] = ring[i][1];
                f[i] = corner_index(o);
            }
            faces.push_back(f);
        }
    return faces;
}

std::vector<std::vector<int>> case_loops(int cfg, bool flip) {
    std::map<int, int> nxt;
    for (const auto& ring : cube_faces()) {
        int ins[4];
        for (int i = 0; i < 4; ++i) ins[i] = (cfg >> ring[i]) & 1;
        for (int i = 0; i < 4; ++i) {
            if (!ins[i] && ins[(i + 1) % 4]) {                                   // entering edge
                int j = (i + 1) % 4;
                while (!(ins[j] && !ins[(j + 1) % 4])) j = (j + 1) % 4;         // next leaving edge counter-clockwise
                const int e_in = edge_id(ring[i], ring[(i + 1) % 4]), e_out = edge_id(ring[j], ring[(j + 1) % 4]);
                if (flip) nxt[e_out] = e_in; else nxt[e_in] = e_out;
            }
        }
    }
    std::vector<std::vector<int>> loops;
    bool seen[12] = {false};
    for (const auto& kv : nxt) {
        const int e = kv.first;
        if (seen[e]) continue;
        std::vector<int> loop{e};
        seen[e] = true;
        while (nxt[loop.back()] != e) { loop.push_back(nxt[loop.back()]); seen[loop.back()] = true; }
        loops.push_back(loop);
    }
    return loops;
}

typedef std::vector<std::array<int, 3>> TriList;

std::vector<TriList> triangulations(const std::vector<int>& loop) {
    const int n = (int)loop.size();
    if (n < 3) return {TriList()};
    std::vector<TriList> out;
    for (int k = 1; k < n - 1; ++k) {   // the polygon edge (first, last) belongs to exactly one triangle (first, k, last)
        const std::vector<int> l(loop.begin(), loop.begin() + k + 1), r(loop.begin() + k, loop.end());
        for (const auto& left : triangulations(l))
            for (const auto& right : triangulations(r)) {
                TriList t = left;
                t.push_back({loop[0], loop[k], loop[n - 1]});
                t.insert(t.end(), right.begin(), right.end());
                out.push_back(t);
            }
    }
    return out;
}

TriList triangulate(const std::vector<int>& loop) {
    const int n = (int)loop.size();
    auto pos = [&](int e) { return (int)(std::find(loop.begin(), loop.end(), e) - loop.begin()); };
    bool have = false;
    int best_bad = 0;
    TriList best;
    for (const auto& tri : triangulations(loop)) {
        int bad = 0;
        for (const auto& t : tri)
            for (int s = 0; s < 3; ++s) {
                const int a = t[s], b = t[(s + 1) % 3];
                const int d = ((pos(a) - pos(b)) % n + n) % n;
                if (d != 1 && d != n - 1 && edges_share_face(a, b)) ++bad;
            }
        if (!have || bad < best_bad || (bad == best_bad && tri < best)) { have = true; best_bad = bad; best = tri; }
    }
    return best;
}

}  // namespace

void mc_build_tables(McTables& t) {
    memset(&t, 0xff, sizeof t);
    // winding: a lone inside corner 0 must give a triangle whose normal points away from it (towards +gradient)
    bool flip = false;
    {
        const std::vector<int> l = case_loops(1, false)[0];
        double p[3][3];
        for (int i = 0; i < 3; ++i) {
            Off3 o; int ax;
            edge_owner(l[i], o, ax);
            for (int c = 0; c < 3; ++c) p[i][c] = o.o[c];
            p[i][ax] = 0.5;
        }
        double u[3], v[3];
        for (int c = 0; c < 3; ++c) { u[c] = p[1][c] - p[0][c]; v[c] = p[2][c] - p[0][c]; }
        const double nsum = (u[1] * v[2] - u[2] * v[1]) + (u[2] * v[0] - u[0] * v[2]) + (u[0] * v[1] - u[1] * v[0]);
        flip = nsum < 0;
    }
    for (int cfg = 0; cfg < 256; ++cfg) {
        int n = 0;
        for (const auto& loop : case_loops(cfg, flip))
            for (const auto& tri : triangulate(loop)) {
                if (n >= MC_MAX_TRI) throw std::logic_error("marching-cubes case table overflow");
                for (int s = 0; s < 3; ++s) t.tri[cfg][3 * n + s] = (unsigned char)tri[s];
                ++n;
            }
        t.n_tri[cfg] = (unsigned char)n;
    }
    for (int e = 0; e < 12; ++e) {
        Off3 o; int ax;
        edge_owner(e, o, ax);
        t.edge_off[e][0] = (unsigned char)o.o[0]; t.edge_off[e][1] = (unsigned char)o.o[1]; t.edge_off[e][2] = (unsigned char)o.o[2];
        t.edge_off[e][3] = (unsigned char)ax;
    }
}

// ------------------------------------------------------------------------------------------------
// device
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int MC_BLOCK = 256;

struct Dims { int n0, n1, n2; };

__device__ __forceinline__ void unravel(int g, Dims d, int& i, int& j, int& k) {
    k = g % d.n2;
    j = (g / d.n2) % d.n1;
    i = g / (d.n2 * d.n1);
}

// vertices owned by grid point g (sign changes along +axis) and triangles of the cell whose lowest corner is g
__device__ __forceinline__ void mc_classify(const float* __restrict__ vol, Dims d, float level, int g, int n_pts,
                                            const McTables* __restrict__ tab, int& nv, int& cfg, int& nt, bool cross[3]) {
    nv = 0; cfg = 0; nt = 0;
    cross[0] = cross[1] = cross[2] = false;
    if (g >= n_pts) return;
    int i, j, k;
    unravel(g, d, i, j, k);
    const int s1 = d.n2, s0 = d.n1 * d.n2;
    const bool in0 = vol[g] < level;
    const bool h0 = i + 1 < d.n0, h1 = j + 1 < d.n1, h2 = k + 1 < d.n2;
    if (h0) cross[0] = in0 != (vol[g + s0] < level);
    if (h1) cross[1] = in0 != (vol[g + s1] < level);
    if (h2) cross[2] = in0 != (vol[g + 1] < level);
    nv = (int)cross[0] + (int)cross[1] + (int)cross[2];
    if (h0 && h1 && h2) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int gi = g + (c & 1) * s0 + ((c >> 1) & 1) * s1 + ((c >> 2) & 1);
            cfg |= (vol[gi] < level ? 1 : 0) << c;
        }
        nt = tab->n_tri[cfg];
    }
}

// exclusive scan of one int2 per thread over the block; returns this thread's offsets, block totals in `total`
__device__ __forceinline__ int2 block_scan(int2 v, int2* sh, int2& total) {
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int d = 1; d < MC_BLOCK; d <<= 1) {
        int2 add = make_int2(0, 0);
        if (t >= d) add = sh[t - d];
        __syncthreads();
        sh[t].x += add.x; sh[t].y += add.y;
        __syncthreads();
    }
    total = sh[MC_BLOCK - 1];
    return make_int2(sh[t].x - v.x, sh[t].y - v.y);
}

__global__ __launch_bounds__(MC_BLOCK) void k_mc_count(const float* vol, Dims d, float level, int n_pts, const McTables* tab, int2* block_sums) {
    __shared__ int2 sh[MC_BLOCK];
    const int g = blockIdx.x * MC_BLOCK + threadIdx.x;
    int nv, cfg, nt;
    bool cross[3];
    mc_classify(vol, d, level, g, n_pts, tab, nv, cfg, nt, cross);
    int2 total;
    block_scan(make_int2(nv, nt), sh, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void k_mc_scan(int2* block_sums, int n_blocks, long long* totals) {
    __shared__ long long shx[1024], shy[1024];
    const int t = threadIdx.x;
    const int per = (n_blocks + 1023) / 1024;
    const int b0 = t * per, b1 = min(n_blocks, b0 + per);
    long long sx = 0, sy = 0;
    for (int b = b0; b < b1; ++b) { sx += block_sums[b].x; sy += block_sums[b].y; }
    shx[t] = sx; shy[t] = sy;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        long long ax = 0, ay = 0;
        if (t >= d) { ax = shx[t - d]; ay = shy[t - d]; }
        __syncthreads();
        shx[t] += ax; shy[t] += ay;
        __syncthreads();
    }
    long long ox = shx[t] - sx, oy = shy[t] - sy;
    if (t == 1023) { totals[0] = shx[t]; totals[1] = shy[t]; }
    for (int b = b0; b < b1; ++b) {     // totals stay below 2^31 (checked on the host before the emit kernels run)
        const int2 v = block_sums[b];
        block_sums[b] = make_int2((int)ox, (int)oy);
        ox += v.x; oy += v.y;
    }
}

__global__ __launch_bounds__(MC_BLOCK) void k_mc_verts(const float* vol, Dims d, float level, int n_pts, const McTables* tab,
                                                       const int2* block_off, float spacing, float origin, float* verts, int* vidmap) {
    __shared__ int2 sh[MC_BLOCK];
    const int g = blockIdx.x * MC_BLOCK + threadIdx.x;
    int nv, cfg, nt;
    bool cross[3];
    mc_classify(vol, d, level, g, n_pts, tab, nv, cfg, nt, cross);
    int2 total;
    const int2 off = block_scan(make_int2(nv, nt), sh, total);
    if (nv == 0) return;
    int i, j, k;
    unravel(g, d, i, j, k);
    const int stride[3] = {d.n1 * d.n2, d.n2, 1};
    const float s0 = vol[g];
    int vid = block_off[blockIdx.x].x + off.x;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (!cross[a]) continue;
        const float s1 = vol[g + stride[a]];
        const float t = __fdiv_rn(__fsub_rn(level, s0), __fsub_rn(s1, s0));
        float p[3] = {(float)i, (float)j, (float)k};
        p[a] = __fadd_rn(p[a], t);
#pragma unroll
        for (int c = 0; c < 3; ++c) verts[3 * (size_t)vid + c] = __fadd_rn(__fmul_rn(p[c], spacing), origin);
        vidmap[3 * (size_t)g + a] = vid;
        ++vid;
    }
}

__global__ __launch_bounds__(MC_BLOCK) void k_mc_faces(const float* vol, Dims d, float level, int n_pts, const McTables* tab,
                                                       const int2* block_off, const int* vidmap, int* faces) {
    __shared__ int2 sh[MC_BLOCK];
    const int g = blockIdx.x * MC_BLOCK + threadIdx.x;
    int nv, cfg, nt;
    bool cross[3];
    mc_classify(vol, d, level, g, n_pts, tab, nv, cfg, nt, cross);
    int2 total;
    const int2 off = block_scan(make_int2(nv, nt), sh, total);
    if (nt == 0) return;
    const int s1 = d.n2, s0 = d.n1 * d.n2;
    const size_t f0 = (size_t)block_off[blockIdx.x].y + off.y;
    for (int t = 0; t < nt; ++t)
        for (int c = 0; c < 3; ++c) {
            const int e = tab->tri[cfg][3 * t + c];
            const int go = g + tab->edge_off[e][0] * s0 + tab->edge_off[e][1] * s1 + tab->edge_off[e][2];
            faces[3 * (f0 + t) + c] = vidmap[3 * (size_t)go + tab->edge_off[e][3]];
        }
}

// voxel grid of MeshExtractor (reference utils.py:97-116), x slowest.  regular: point g = (i, j, k) * voxel_size + (-1).
// Otherwise the grid the reference really samples: `overall_index.long() / vol_dim` is true division under torch >= 1.6, so in
// float32 q1 = g / n, y index = fmod(q1, n), x index = fmod(q1 / n, n) (sheared by up to one voxel; golden_voxel_grid.npz).
__global__ void k_grid_points(float4* pts, int n, float voxel_size, int regular) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n * n * n) return;
    float fi, fj;
    const float fk = (float)(g % n);
    if (regular) {
        fj = (float)((g / n) % n);
        fi = (float)(g / (n * n));
    } else {
        const float fn = (float)n;
        const float q1 = __fdiv_rn((float)g, fn);
        fj = fmodf(q1, fn);
        fi = fmodf(__fdiv_rn(q1, fn), fn);
    }
    pts[g] = make_float4(__fadd_rn(__fmul_rn(fi, voxel_size), -1.f), __fadd_rn(__fmul_rn(fj, voxel_size), -1.f),
                         __fadd_rn(__fmul_rn(fk, voxel_size), -1.f), 0.f);
}

}  // namespace

int mc_num_blocks(int n_pts) { return (n_pts + MC_BLOCK - 1) / MC_BLOCK; }

hipError_t launch_grid_points(float4* pts, int n, float voxel_size, int regular, hipStream_t s) {
    const int total = n * n * n;
    hipLaunchKernelGGL(k_grid_points, dim3((total + 255) / 256), dim3(256), 0, s, pts, n, voxel_size, regular);
    return hipGetLastError();
}

hipError_t launch_mc_count(const float* vol, int n0, int n1, int n2, float level, const McTables* tab, int2* block_sums, long long* totals,
                           hipStream_t s) {
    const int n_pts = n0 * n1 * n2, nb = mc_num_blocks(n_pts);
    const Dims d{n0, n1, n2};
    hipLaunchKernelGGL(k_mc_count, dim3(nb), dim3(MC_BLOCK), 0, s, vol, d, level, n_pts, tab, block_sums);
    hipLaunchKernelGGL(k_mc_scan, dim3(1), dim3(1024), 0, s, block_sums, nb, totals);
    return hipGetLastError();
}

hipError_t launch_mc_emit(const float* vol, int n0, int n1, int n2, float level, const McTables* tab, const int2* block_off, float spacing,
                          float origin, float* verts, int* vidmap, int* faces, hipStream_t s) {
    const int n_pts = n0 * n1 * n2, nb = mc_num_blocks(n_pts);
    const Dims d{n0, n1, n2};
    hipLaunchKernelGGL(k_mc_verts, dim3(nb), dim3(MC_BLOCK), 0, s, vol, d, level, n_pts, tab, block_off, spacing, origin, verts, vidmap);
    hipLaunchKernelGGL(k_mc_faces, dim3(nb), dim3(MC_BLOCK), 0, s, vol, d, level, n_pts, tab, block_off, vidmap, faces);
    return hipGetLastError();
}

}  // namespace dsp
