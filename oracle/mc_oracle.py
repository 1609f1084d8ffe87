"""CPU oracle for the mesh-extraction step that follows every successful reconstruction
(reference reconstruct/optimizer.py:206-223 -> reconstruct/utils.py:119-140).

TEST INFRASTRUCTURE ONLY -- never imported by the product path.

PARITY UNPINNED.  The reference calls scikit-image's `measure.marching_cubes_lewiner(volume, level=0.0,
spacing=[voxel_size]*3)` (utils.py:130) and shifts the vertices by the grid origin (-1,-1,-1) (utils.py:133-138).
scikit-image is a third-party dependency that is NOT in /root/reference and NOT installed in this image (the
reference pins scikit-image 0.18 in environment_cuda113.yml), and the reference holds no golden meshes, so there is
nothing to pin a restatement of Lewiner's 33-case tables against.  What is restated here instead is the published
marching-cubes construction those tables refine (Lorensen & Cline 1987: one vertex per sign-changing grid edge by
linear interpolation, one polygon per connected surface patch of a cell), with the case table GENERATED rather than
transcribed:

  * corners c = (c&1, c>>1&1, c>>2&1) offsets along volume axes (0, 1, 2); inside = value < level;
  * on every cube face (corners listed counter-clockwise seen from outside the cube) each "entering" edge
    (outside -> inside corner) is joined to the next "leaving" edge counter-clockwise.  On an ambiguous face (two
    diagonal inside corners) this always cuts the two inside corners apart; the rule only looks at the face's own
    four signs, so the two cells sharing the face agree and the mesh is watertight;
  * the face segments chain into closed loops; each loop gets the triangulation with the fewest diagonals lying inside
    a cube face (such a diagonal could coincide with the neighbouring cell's and give an edge shared by four triangles;
    ties: lexicographically smallest triangle list); winding gives normals along +grad (outward for a signed distance).

The mesh agrees with Lewiner's in vertex positions (same edge interpolation) and in every non-ambiguous cell; it can
differ in how an ambiguous cell is triangulated.  Ordering is fixed so that an implementation can be compared
bit-for-bit: vertices by (grid point index, axis), faces by (cell index, table order).
"""
import numpy as np

AXIS_OTHER = {0: (1, 2), 1: (2, 0), 2: (0, 1)}     # (b, c) with e_a = e_b x e_c


def corner_offset(c):
    return (c & 1, (c >> 1) & 1, (c >> 2) & 1)


def corner_index(off):
    return off[0] | (off[1] << 1) | (off[2] << 2)


def edge_id(c0, c1):
    """Cube edge between two adjacent corners: id = 4*axis + u + 2*v, (u, v) = offsets of the edge in the two other
    axes taken in increasing axis order; the edge is owned by its corner with offset 0 along `axis`."""
    o0, o1 = corner_offset(c0), corner_offset(c1)
    axis = [i for i in range(3) if o0[i] != o1[i]]
    assert len(axis) == 1
    axis = axis[0]
    rest = [i for i in range(3) if i != axis]
    return 4 * axis + o0[rest[0]] + 2 * o0[rest[1]]


def edge_owner(e):
    """(corner offset of the owning corner, axis) of cube edge e."""
    axis, uv = e // 4, e % 4
    rest = [i for i in range(3) if i != axis]
    off = [0, 0, 0]
    off[rest[0]] = uv & 1
    off[rest[1]] = uv >> 1
    return tuple(off), axis


def cube_faces():
    """Six faces, corners counter-clockwise as seen from outside the cube."""
    faces = []
    for a in range(3):
        b, c = AXIS_OTHER[a]
        for side in (0, 1):
            ring = [(0, 0), (1, 0), (1, 1), (0, 1)]
            if side == 0:
                ring = ring[::-1]
            corners = []
            for (ub, uc) in ring:
                off = [0, 0, 0]
                off[a], off[b], off[c] = side, ub, uc
                corners.append(corner_index(off))
            faces.append(corners)
    return faces


def _loops(cfg, flip, alt=False):
    """alt=True resolves an AMBIGUOUS face (two diagonally opposite inside corners) the other way: each entering edge is joined
    to the second leaving edge counter-clockwise, so the face separates its two OUTSIDE corners.  Also a rule of the face's
    own four signs, hence also watertight -- a different, equally valid triangulation over the same vertices (test use only)."""
    nxt = {}
    for ring in cube_faces():
        ins = [(cfg >> c) & 1 for c in ring]
        ambiguous = ins == [ins[0], 1 - ins[0], ins[0], 1 - ins[0]]
        for i in range(4):
            if not ins[i] and ins[(i + 1) % 4]:                      # entering edge
                j = (i + 1) % 4
                while not (ins[j] and not ins[(j + 1) % 4]):         # next leaving edge counter-clockwise
                    j = (j + 1) % 4
                if alt and ambiguous:                                # ... or the one after it
                    j = (j + 1) % 4
                    while not (ins[j] and not ins[(j + 1) % 4]):
                        j = (j + 1) % 4
                e_in = edge_id(ring[i], ring[(i + 1) % 4])
                e_out = edge_id(ring[j], ring[(j + 1) % 4])
                if flip:
                    assert e_out not in nxt
                    nxt[e_out] = e_in
                else:
                    assert e_in not in nxt
                    nxt[e_in] = e_out
    loops, seen = [], set()
    for e in sorted(nxt):
        if e in seen:
            continue
        loop = [e]
        seen.add(e)
        while nxt[loop[-1]] != e:
            loop.append(nxt[loop[-1]])
            assert loop[-1] not in seen
            seen.add(loop[-1])
        loops.append(loop)
    return loops


def _edge_midpoint(e):
    off, axis = edge_owner(e)
    p = np.array(off, np.float64)
    p[axis] = 0.5
    return p


def _edges_share_face(e0, e1):
    """True when cube edges e0 and e1 lie on a common cube face."""
    def faces_of(e):
        off, axis = edge_owner(e)
        return {(a, off[a]) for a in range(3) if a != axis}
    return bool(faces_of(e0) & faces_of(e1))


def _triangulations(loop):
    """All triangulations of the polygon `loop` (list of vertices), each a list of index triples in loop order."""
    n = len(loop)
    if n < 3:
        return [[]]
    out = []
    # the polygon edge (loop[0], loop[-1]) belongs to exactly one triangle (0, k, n-1)
    for k in range(1, n - 1):
        for left in _triangulations(loop[:k + 1]):
            for right in _triangulations(loop[k:]):
                out.append(left + [(loop[0], loop[k], loop[-1])] + right)
    return out


def _triangulate(loop):
    """Triangulation of one surface loop.  A diagonal joining two crossings that lie on the same cube face would run
    inside that face, where the neighbouring cell may place the same diagonal (an edge shared by four triangles), so the
    triangulation with the fewest such diagonals is taken; ties go to the lexicographically smallest triangle list."""
    best = None
    for tri in _triangulations(loop):
        bad = 0
        for t in tri:
            for a, b in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])):
                ia, ib = loop.index(a), loop.index(b)
                consecutive = (ia - ib) % len(loop) in (1, len(loop) - 1)
                if not consecutive and _edges_share_face(a, b):
                    bad += 1
        key = (bad, tri)
        if best is None or key < best:
            best = key
    return best[1]


def build_tables(alt=False):
    """(n_tri[256] uint8, tri[256, MAX_TRI*3] uint8 of cube edge ids, 255-padded).  alt: see _loops."""
    # winding: a lone inside corner 0 must give a triangle whose normal points away from it
    flip = False
    l = _loops(1, False)[0]
    p = [_edge_midpoint(e) for e in l]
    if np.dot(np.cross(p[1] - p[0], p[2] - p[0]), np.ones(3)) < 0:
        flip = True
    tris = []
    for cfg in range(256):
        t = []
        for loop in _loops(cfg, flip, alt):
            assert len(loop) >= 3
            for tri in _triangulate(loop):
                t += list(tri)
        tris.append(t)
    max_tri = max(len(t) for t in tris) // 3
    n_tri = np.array([len(t) // 3 for t in tris], np.uint8)
    tab = np.full((256, max_tri * 3), 255, np.uint8)
    for cfg, t in enumerate(tris):
        tab[cfg, :len(t)] = t
    return n_tri, tab


_TABLES = None


def tables():
    global _TABLES
    if _TABLES is None:
        _TABLES = build_tables()
    return _TABLES


def marching_cubes(volume, level=0.0, table=None):
    """volume (n0, n1, n2) float32 -> (vertices (V,3) float32 in INDEX coordinates, faces (F,3) int32).
    table: (n_tri, tri) to use instead of the library's construction (tests: build_tables(alt=True)).

    vertex on the edge from grid point g along axis a:  g + t * e_a,  t = (level - s0) / (s1 - s0)  (float32)."""
    vol = np.ascontiguousarray(volume, np.float32)
    n0, n1, n2 = vol.shape
    level = np.float32(level)
    inside = vol < level
    # --- vertices: one per sign-changing edge, ordered by (grid point, axis)
    cross = np.zeros(vol.shape + (3,), bool)
    tpar = np.zeros(vol.shape + (3,), np.float32)
    for a in range(3):
        sl0 = [slice(None)] * 3
        sl1 = [slice(None)] * 3
        sl0[a] = slice(0, -1)
        sl1[a] = slice(1, None)
        sl0, sl1 = tuple(sl0), tuple(sl1)
        c = inside[sl0] != inside[sl1]
        cross[sl0 + (a,)] = c
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (level - vol[sl0]) / (vol[sl1] - vol[sl0])
        tpar[sl0 + (a,)] = np.where(c, t, np.float32(0))
    flat = cross.reshape(-1)
    vid = np.cumsum(flat, dtype=np.int64) - 1            # vertex id of (grid point, axis) where flat
    gidx, axis = np.nonzero(cross.reshape(-1, 3))
    ijk = np.stack(np.unravel_index(gidx, vol.shape), 1).astype(np.float32)
    verts = ijk.copy()
    verts[np.arange(len(axis)), axis] = ijk[np.arange(len(axis)), axis] + tpar.reshape(-1, 3)[gidx, axis]
    # --- faces: per cell (ordered by the index of its lowest corner), table order
    n_tri, tab = tables() if table is None else table
    cfg = np.zeros((n0 - 1, n1 - 1, n2 - 1), np.int32)
    for c in range(8):
        o = corner_offset(c)
        cfg |= inside[o[0]:n0 - 1 + o[0], o[1]:n1 - 1 + o[1], o[2]:n2 - 1 + o[2]].astype(np.int32) << c
    cells = np.nonzero(n_tri[cfg.reshape(-1)])[0]
    ci, cj, ck = np.unravel_index(cells, cfg.shape)
    faces = []
    vid3 = vid.reshape(n0, n1, n2, 3)
    ccfg = cfg.reshape(-1)[cells]
    own = [edge_owner(e) for e in range(12)]
    own_off = np.array([o for o, _ in own])
    own_axis = np.array([a for _, a in own])
    max_t = tab.shape[1] // 3
    for t in range(max_t):
        sel = n_tri[ccfg] > t
        if not sel.any():
            break
        e = tab[ccfg[sel], 3 * t:3 * t + 3].astype(np.int64)         # (n, 3) edge ids
        i = ci[sel, None] + own_off[e, 0]
        j = cj[sel, None] + own_off[e, 1]
        k = ck[sel, None] + own_off[e, 2]
        f = vid3[i, j, k, own_axis[e]]
        faces.append((cells[sel], np.full(sel.sum(), t), f))
    if faces:
        cell_id = np.concatenate([f[0] for f in faces])
        tri_id = np.concatenate([f[1] for f in faces])
        fv = np.concatenate([f[2] for f in faces])
        order = np.lexsort((tri_id, cell_id))
        fv = fv[order]
    else:
        fv = np.zeros((0, 3), np.int64)
    return verts.astype(np.float32), fv.astype(np.int32)


def to_object_frame(verts_index, vol_dim):
    """Index coordinates -> the decoder's [-1, 1]^3 frame: v * voxel_size + (-1)  (utils.py:127,133-138), float32."""
    vs = np.float32(2.0 / (vol_dim - 1))
    return (verts_index.astype(np.float32) * vs + np.float32(-1.0)).astype(np.float32)


def convert_sdf_voxels_to_mesh(sdf_volume):
    """Restates reference utils.py:119-140 with the marching cubes above."""
    vol = np.asarray(sdf_volume, np.float32)
    if not (vol.min() <= 0.0 <= vol.max()):
        raise ValueError("Surface level must be within volume data range.")      # what scikit-image raises
    v, f = marching_cubes(vol, 0.0)
    return to_object_frame(v, vol.shape[0]), f


def mesh_report(verts, faces):
    """Topology facts used by the tests: (n_boundary_edges, n_nonmanifold_edges, euler_characteristic, signed_volume)."""
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]]).astype(np.int64)
    key = np.minimum(e[:, 0], e[:, 1]) * (len(verts) + 1) + np.maximum(e[:, 0], e[:, 1])
    sign = np.where(e[:, 0] < e[:, 1], 1, -1)
    uniq, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
    bal = np.zeros(len(uniq), np.int64)
    np.add.at(bal, inv, sign)
    boundary = int((cnt == 1).sum())
    nonmanifold = int(((cnt > 2) | ((cnt == 2) & (bal != 0))).sum())
    used = np.unique(faces)
    euler = len(used) - len(uniq) + len(faces)
    p = verts.astype(np.float64)
    a, b, c = p[faces[:, 0]], p[faces[:, 1]], p[faces[:, 2]]
    vol = float(np.einsum("ij,ij->i", a, np.cross(b, c)).sum() / 6.0)
    return boundary, nonmanifold, euler, vol
