"""UV atlas generator for meshes without texture coordinates — the stand-in for the xatlas call of
src/models/textured_mesh.py:392-404 (`xatlas.parametrize(v, f)` -> vmapping, indices, uvs -> `vt`, `ft`), which is a C++
extension that is absent offline.  Layout parity with xatlas is impossible; what is kept is its contract: every face gets its own
triangle in [0,1]^2, triangles do not overlap, faces that are neighbours on a smooth piece of surface stay neighbours in the atlas
(so bilinear sampling bleeds only across chart borders, which are separated by a gutter), deterministic output, and the
`cache/<stem>/{vt,ft}.pth` round trip of the caller.

Method (host-side numpy; runs once per mesh and is cached):
  1. charts   faces in descending area order seed charts that grow breadth-first over shared edges while the face normal stays
              within `max_angle_deg` of the SEED's normal (so every triangle of a chart faces the projection plane: no flipped
              triangles, stretch bounded by 1 / cos(max_angle)).
  2. project  orthographic projection onto the plane of the seed normal, rotated to the minimum-area bounding box.
  3. untangle a chart whose surface winds over itself inside the angle bound overlaps itself in projection: every chart is
              rasterised on its own and the later of two overlapping faces moves out into a new chart, until no texel centre is
              claimed twice.
  4. pack     "skyline with profiles": charts in descending height order; each is described by its bottom / top outline per texel
              column (widened by half the gutter), tried in its four right-angle poses and dropped at the x where it comes to rest
              lowest on the current skyline; small charts first look for a free box in the holes under the skyline (summed-area
              test on a 4-texel cell grid).  One global texel density for all charts, found by bisection so that the stack just fits
              the square.
  5. verify   the packed atlas is rasterised once more; any texel still claimed twice ejects its later face and packs again.
Measured on the bundled meshes at 1024^2, gutter 2: nascar 217 charts / 0.73 of the texels used / 1 445 seam edges of 11 250;
bunny 74 / 0.65 / 1 066 of 9 897; blub 93 / 0.63 / 1 500 of 21 312 (grid_atlas: a seam on every edge, 0.41 used).
`grid_atlas` (textured_mesh.py) — one cell per triangle, a seam on every edge — stays available as a fallback."""
import numpy as np


def _face_normals_areas(v, f):
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    n = np.cross(b - a, c - a)
    l = np.linalg.norm(n, axis=1)
    return n / np.maximum(l, 1e-30)[:, None], 0.5 * l


def _adjacency(f):
    """-> list of neighbour faces per face (faces sharing an edge; non-manifold edges link all their faces), in a fixed order."""
    F = f.shape[0]
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0)
    e.sort(axis=1)
    fid = np.tile(np.arange(F), 3)
    key = e[:, 0].astype(np.int64) * (int(f.max()) + 1) + e[:, 1]
    order = np.lexsort((fid, key))
    key, fid = key[order], fid[order]
    nbr = [[] for _ in range(F)]
    start = np.flatnonzero(np.r_[True, key[1:] != key[:-1]])
    end = np.r_[start[1:], len(key)]
    for s, t in zip(start, end):
        if t - s < 2:
            continue
        g = fid[s:t]
        for i in g:
            for j in g:
                if i != j:
                    nbr[i].append(int(j))
    return nbr


def _grow_charts(normals, areas, nbr, cos_min, todo=None, chart_of=None, next_id=0):
    """Breadth-first region growing.  todo: faces to (re)assign (default all).  -> chart_of [F], seeds {chart: seed face}."""
    F = normals.shape[0]
    if chart_of is None:
        chart_of = np.full(F, -1, np.int64)
    pool = np.arange(F) if todo is None else np.asarray(todo)
    free = np.zeros(F, bool)
    free[pool] = True
    seeds = {}
    for s in pool[np.lexsort((pool, -areas[pool]))]:                 # descending area, ties by face id
        if not free[s]:
            continue
        cid = next_id
        next_id += 1
        seeds[cid] = int(s)
        ns = normals[s]
        chart_of[s] = cid
        free[s] = False
        queue, head = [int(s)], 0
        while head < len(queue):
            cur = queue[head]
            head += 1
            for nb in nbr[cur]:
                if free[nb] and normals[nb] @ ns >= cos_min:
                    free[nb] = False
                    chart_of[nb] = cid
                    queue.append(nb)
    return chart_of, seeds, next_id


def _project(v, f, faces_of, n):
    """Chart faces -> (local vertex ids, 2-D coordinates of the chart's vertices in mesh units, rotated to the min-area box)."""
    vid, inv = np.unique(f[faces_of].reshape(-1), return_inverse=True)
    p = v[vid]
    t = np.cross(n, [0.0, 0.0, 1.0] if abs(n[2]) < 0.9 else [1.0, 0.0, 0.0])
    t /= np.linalg.norm(t)
    b = np.cross(n, t)
    q = np.stack([p @ t, p @ b], 1)
    best = None
    for deg in range(0, 90, 5):                                      # the box of angle a + 90 is the box of angle a
        a = np.deg2rad(deg)
        r = q @ np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
        ext = r.max(0) - r.min(0)
        if best is None or ext[0] * ext[1] < best[0] - 1e-12:
            best = (ext[0] * ext[1], r, ext)
    _, r, ext = best
    if ext[1] > ext[0]:                                              # lie flat: wide charts settle better on a skyline
        r = np.stack([r[:, 1], -r[:, 0]], 1)
    r = r - r.min(0)
    return vid, inv.reshape(-1, 3), r


def _profiles(q, tri, scale, gutter):
    """Bottom / top outline of a chart per texel column, in texels, widened by the gutter.  q [n,2] chart coordinates (mesh units),
    tri [m,3] local vertex ids.  -> (bottom [w], top [w]) for columns 0..w-1 of the chart's box placed at x = gutter."""
    p = q * scale
    w = int(np.ceil(p[:, 0].max() + 1e-9)) + 1
    e = np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]], 0)
    a, b = p[e[:, 0]], p[e[:, 1]]
    sw = a[:, 0] > b[:, 0]
    a2 = np.where(sw[:, None], b, a)
    b2 = np.where(sw[:, None], a, b)
    c0 = np.floor(a2[:, 0]).astype(np.int64)
    c1 = np.minimum(np.floor(b2[:, 0]).astype(np.int64), w - 1)
    cnt = c1 - c0 + 1
    eid = np.repeat(np.arange(len(cnt)), cnt)
    col = np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt) + c0[eid]
    ax, ay, bx, by = a2[eid, 0], a2[eid, 1], b2[eid, 0], b2[eid, 1]
    dx = np.maximum(bx - ax, 1e-12)
    xl = np.clip(col, ax, bx)
    xr = np.clip(col + 1.0, ax, bx)
    yl = ay + (by - ay) * (xl - ax) / dx
    yr = ay + (by - ay) * (xr - ax) / dx
    lo = np.full(w, np.inf)
    hi = np.full(w, -np.inf)
    np.minimum.at(lo, col, np.minimum(yl, yr))
    np.maximum.at(hi, col, np.maximum(yl, yr))
    ok = np.isfinite(lo)
    lo = np.where(ok, lo, np.inf)
    hi = np.where(ok, hi, -np.inf)
    g = (int(gutter) + 1) // 2                                       # half the gutter on every side of every chart: `gutter` texels between two charts
    lo_p = np.pad(lo, g, constant_values=np.inf)
    hi_p = np.pad(hi, g, constant_values=-np.inf)
    if g > 0:                                                        # widen by the gutter: min / max filter over +-g columns
        win_lo = np.lib.stride_tricks.sliding_window_view(np.pad(lo_p, g, constant_values=np.inf), 2 * g + 1)
        win_hi = np.lib.stride_tricks.sliding_window_view(np.pad(hi_p, g, constant_values=-np.inf), 2 * g + 1)
        lo_p, hi_p = win_lo.min(1), win_hi.max(1)
    return np.floor(lo_p) - g, np.ceil(hi_p) + g


def _quarter(q):
    """The chart turned by 90 degrees (counter-clockwise), moved back into the positive quadrant."""
    return np.stack([q[:, 1].max() - q[:, 1], q[:, 0]], 1)


def _pack(charts, scale, R, gutter, holes=False):
    """Skyline packing at `scale` texels per mesh unit.  Every chart is tried in its four right-angle poses and keeps the one that
    comes to rest lowest; with `holes`, small charts first look for a free box under the skyline.  -> (placements {k: (x, y, quarter, turned, w_box, h_box)} in texels, total height) or None if a chart
    fits in no pose."""
    g = (int(gutter) + 1) // 2
    prof = []
    for k, (q, tri) in enumerate(charts):
        pr = [_profiles(q, tri, scale, gutter), _profiles(_quarter(q), tri, scale, gutter)]
        pr = [p if len(p[0]) <= R else None for p in pr]
        if pr[0] is None and pr[1] is None:
            return None
        prof.append(pr)

    def box_h(p):
        return float(np.max(p[1][np.isfinite(p[1])]))
    height = [min(box_h(p) for p in pr if p is not None) for pr in prof]
    order = sorted(range(len(charts)), key=lambda k: (-height[k], k))
    sky = np.zeros(R)
    CS = 4                                                           # the hole search works on cells of CS x CS texels
    Rc = -(-R // CS)
    occ = np.zeros((Rc + 1, Rc), bool)                               # a cell is taken when any of its texels is (gutter included)
    sat = None
    pos = {}

    def mark(x, y0, l2, h2):
        for i in np.flatnonzero(np.isfinite(h2)):
            a, b = int(max(y0 + l2[i], 0)), int(min(y0 + h2[i], R))
            if b > a:
                occ[a // CS:(b - 1) // CS + 1, (x + i) // CS] = True

    for k in order:
        best = None
        small = holes and min(len(p[0]) * box_h(p) for p in prof[k] if p is not None) <= 0.02 * R * R
        if small and pos:
            # a small chart first looks for a hole under the skyline that takes its whole box: summed-area test of every
            # cell-aligned position at once
            if sat is None:
                sat = np.zeros((Rc + 2, Rc + 1), np.int32)
                sat[1:, 1:] = occ.cumsum(0, dtype=np.int32).cumsum(1, dtype=np.int32)
                sky_c = np.pad(sky, (0, Rc * CS - R), constant_values=0.0).reshape(Rc, CS).min(1)
            for quarter in (0, 1):
                if prof[k][quarter] is None:
                    continue
                lo, hi = prof[k][quarter]
                w, H = len(lo), box_h((lo, hi))
                wc, hc = -(-w // CS), -(-int(H + g) // CS)           # box in cells: rows -g .. H of the profile frame
                if hc >= Rc or wc > Rc:
                    continue
                s4 = sat[hc:Rc + 1, wc:] - sat[:Rc + 1 - hc, wc:] - sat[hc:Rc + 1, :Rc + 1 - wc] + sat[:Rc + 1 - hc, :Rc + 1 - wc]
                # the box must also stay below the skyline over its columns (the skyline's bookkeeping is then untouched)
                skymin = np.lib.stride_tricks.sliding_window_view(sky_c, wc).min(1)
                ys = np.arange(s4.shape[0])[:, None]
                ok = (s4 == 0) & ((ys + hc) * CS <= skymin[None, :]) & (np.arange(s4.shape[1])[None, :] * CS + w <= R)
                if ok.any():
                    yy, xx = np.nonzero(ok)
                    j = int(np.lexsort((xx, yy))[0])
                    cand = (float((yy[j] + hc) * CS), float(yy[j] * CS + g), quarter, 0, int(xx[j]) * CS, hi, w, H, lo, True)
                    if best is None or cand[:4] < best[:4]:
                        best = cand
        if best is None:
            for quarter in (0, 1):
                if prof[k][quarter] is None:
                    continue
                lo, hi = prof[k][quarter]
                w = len(lo)
                win = np.lib.stride_tricks.sliding_window_view(sky, w)   # [R-w+1, w]
                H = box_h((lo, hi))                                  # box height incl. the gutter row on top (the bottom one is at -g)
                for turned in (0, 1):
                    # turned by 180 degrees about the box centre: column i <- column w-1-i, bottom = top', top = bottom'
                    l2, h2 = (lo, hi) if not turned else ((H - g) - hi[::-1], (H - g) - lo[::-1])
                    y = (win - np.where(np.isfinite(l2), l2, np.inf)[None, :]).max(1)
                    top = y + np.max(h2[np.isfinite(h2)])
                    x = int(np.lexsort((np.arange(len(y)), y, top))[0])  # lowest resulting top, then lowest origin, then leftmost
                    cand = (float(top[x]), float(np.ceil(y[x])), quarter, turned, x, h2, w, H, l2, False)
                    if best is None or cand[:4] < best[:4]:
                        best = cand
        _, y0, quarter, turned, x, h2, w, H, l2, in_hole = best
        pos[k] = (x + g, y0, quarter, turned, w - 2 * g, H - g)      # profile column 0 is the gutter column left of the chart
        if not in_hole:
            seg = sky[x:x + w]
            sky[x:x + w] = np.where(np.isfinite(h2), np.maximum(seg, y0 + h2), seg)
        mark(x, y0, l2, h2)
        sat = None
    return pos, float(sky.max())


def rasterize_uv_counts(vt, ft, R):
    """How many triangles cover each texel centre of an R x R atlas (strict interior).  The non-overlap check of the generator and
    of its test.  -> (count [R,R] int32, owner face of the last hit [R,R] int64)."""
    cnt = np.zeros((R, R), np.int32)
    own = np.full((R, R), -1, np.int64)
    p = vt[ft] * R                                                   # [F,3,2] texel coordinates
    for k in range(ft.shape[0]):
        a, b, c = p[k]
        x0, x1 = int(np.floor(min(a[0], b[0], c[0]))), int(np.ceil(max(a[0], b[0], c[0])))
        y0, y1 = int(np.floor(min(a[1], b[1], c[1]))), int(np.ceil(max(a[1], b[1], c[1])))
        x0, y0, x1, y1 = max(x0, 0), max(y0, 0), min(x1, R), min(y1, R)
        if x1 <= x0 or y1 <= y0:
            continue
        xs = np.arange(x0, x1) + 0.5
        ys = np.arange(y0, y1) + 0.5
        X, Y = np.meshgrid(xs, ys)
        d = (b[0] - a[0]) * (c[1] - a[1]) - (c[0] - a[0]) * (b[1] - a[1])
        if abs(d) < 1e-12:
            continue
        w1 = ((X - a[0]) * (c[1] - a[1]) - (c[0] - a[0]) * (Y - a[1])) / d
        w2 = ((b[0] - a[0]) * (Y - a[1]) - (X - a[0]) * (b[1] - a[1])) / d
        eps = 1e-6
        inside = (w1 > eps) & (w2 > eps) & (w1 + w2 < 1 - eps)
        cnt[y0:y1, x0:x1] += inside
        own[y0:y1, x0:x1][inside] = k
    return cnt, own


def chart_atlas(vertices, faces, resolution=1024, gutter=2, max_angle_deg=50.0, return_info=False):
    """vertices [V,3] float, faces [F,3] int -> vt [N,2] float32 in [0,1]^2, ft [F,3] int64 (and an info dict).
    Deterministic: the same mesh gives the same atlas on every run and rank."""
    v = np.asarray(vertices, np.float64)
    f = np.asarray(faces, np.int64)
    F = f.shape[0]
    R = int(resolution)
    normals, areas = _face_normals_areas(v, f)
    nbr = _adjacency(f)
    cos_min = float(np.cos(np.deg2rad(max_angle_deg)))
    chart_of, seeds, next_id = _grow_charts(normals, areas, nbr, cos_min)
    s0 = 0.75 * R / max(float(np.sqrt(areas.sum())), 1e-12)           # about the density the packer will settle on

    def eject(faces_out):
        """Offending faces leave their chart and form new charts among themselves (same angle bound)."""
        nonlocal chart_of, next_id
        fo = np.asarray(sorted(faces_out), np.int64)
        for c in set(int(x) for x in chart_of[fo]):
            if seeds[c] in faces_out:                                    # keep a seed inside every chart that stays
                rest = np.flatnonzero((chart_of == c) & ~np.isin(np.arange(F), fo))
                if len(rest):
                    seeds[c] = int(rest[np.lexsort((rest, -areas[rest]))[0]])
                else:
                    del seeds[c]
        chart_of[fo] = -1
        chart_of, new_seeds, next_id = _grow_charts(normals, areas, nbr, cos_min, todo=fo, chart_of=chart_of, next_id=next_id)
        seeds.update(new_seeds)

    # a chart whose surface winds over itself inside the angle bound overlaps in projection: rasterise each chart on its own and
    # move the later of two overlapping faces out, until no texel centre is claimed twice
    pending = sorted(seeds)
    for _round in range(12):
        out = set()
        for c in pending:
            fo = np.flatnonzero(chart_of == c)
            if len(fo) < 2:
                continue
            _, tri, q = _project(v, f, fo, normals[seeds[c]])
            L = int(np.ceil(q.max() * s0)) + 2
            cnt, own = rasterize_uv_counts((q * s0 + 1.0) / L, tri, L)
            if cnt.max() > 1:
                out.update(int(fo[k]) for k in np.unique(own[cnt > 1]))
        if not out:
            break
        before = set(seeds)
        touched = set(int(x) for x in chart_of[np.asarray(sorted(out))])
        eject(out)
        pending = sorted((set(seeds) - before) | (touched & set(seeds)))
    for _round in range(4):
        ids = sorted(seeds)
        charts, members = [], []
        for c in ids:
            fo = np.flatnonzero(chart_of == c)
            vid, tri, q = _project(v, f, fo, normals[seeds[c]])
            charts.append((q, tri))
            members.append((fo, vid))
        # one texel density for all charts: the largest scale whose packing fits the square — bisection with the plain skyline,
        # then a short ladder above it with the hole search on (small charts under overhangs buy 5-15 % of density)
        hi_s, lo_s, best = 1.1 * R / max(float(np.sqrt(areas.sum())), 1e-12), 0.0, None
        for _ in range(10):
            mid = 0.5 * (lo_s + hi_s)
            res = _pack(charts, mid, R, gutter)
            if res is not None and res[1] <= R:
                best, lo_s = (mid, res[0]), mid
            else:
                hi_s = mid
        if best is None:
            raise RuntimeError("chart_atlas: the charts do not fit the atlas at any texel density")
        base_s, lo_m, hi_m = best[0], None, 1.2
        res = _pack(charts, base_s, R, gutter, holes=True)
        if res is not None and res[1] <= R:
            best, lo_m = (base_s, res[0]), 1.0
            for _ in range(5):
                mid = 0.5 * (lo_m + hi_m)
                res = _pack(charts, base_s * mid, R, gutter, holes=True)
                if res is not None and res[1] <= R:
                    best, lo_m = (base_s * mid, res[0]), mid
                else:
                    hi_m = mid
        scale, pos = best
        vts, ft = [], np.zeros((F, 3), np.int64)
        base = 0
        for k, ((q, tri), (fo, vid)) in enumerate(zip(charts, members)):
            x, y, quarter, turned, wb, hb = pos[k]
            p2 = (_quarter(q) if quarter else q) * scale
            if turned:                                                   # 180 degrees about the centre of the chart's texel box
                p2 = np.array([float(wb), hb]) - p2
            vts.append((p2 + np.array([x, y])) / R)
            ft[fo] = tri + base
            base += q.shape[0]
        vt = np.concatenate(vts, 0)
        cnt, own = rasterize_uv_counts(vt, ft, R)
        if cnt.max() <= 1:
            break
        eject(set(int(k) for k in np.unique(own[cnt > 1])))           # the final density differs a little from s0: rare
    vt = np.clip(vt, 0.0, 1.0).astype(np.float32)
    if not return_info:
        return vt, ft
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0)
    e.sort(axis=1)
    fid = np.tile(np.arange(F), 3)
    key = e[:, 0] * (int(f.max()) + 1) + e[:, 1]
    o = np.argsort(key, kind='stable')
    key, fid = key[o], fid[o]
    same = key[1:] == key[:-1]
    seam = int(np.sum(same & (chart_of[fid[1:]] != chart_of[fid[:-1]])))
    info = dict(charts=len(seeds), seam_edges=seam, interior_edges=int(np.sum(same)) - seam, texels_per_unit=float(scale),
                utilisation=float((cnt > 0).mean()), overlap_texels=int((cnt > 1).sum()), resolution=R, gutter=int(gutter))
    return vt, ft, info
