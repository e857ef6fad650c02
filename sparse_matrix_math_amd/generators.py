"""Deterministic synthetic matrices of the benchmark configurations (BASELINE.json configs).

Host (numpy) implementations of the same laws the device generators in csrc/smm_gen.hip follow; tests check the
two agree bit for bit.  All return the three arrays of the reference's CSRMatrix layout
(include/sparse_matrix_math.h:1243-1259): start int32[rows+1], positions int32[nnz] ascending per row,
values T[nnz].
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = 0x9E3779B97F4A7C15


def _mix64(z):
    """splitmix64 finalizer on a uint64 array (wrapping arithmetic)."""
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def band_offsets(n, k=25, seed=0x5EED, max_offset=1 << 20):
    """K distinct offsets in [1, min(max_offset, n)) drawn from splitmix64(seed), ascending."""
    m = min(int(max_offset), int(n))
    if m < 2:
        return np.zeros(0, dtype=np.int64)
    want = min(int(k), 32, m - 1)
    state = int(seed) & 0xFFFFFFFFFFFFFFFF
    out = []
    while len(out) < want:
        state = (state + _GOLD) & 0xFFFFFFFFFFFFFFFF
        cand = 1 + int(_mix64(np.uint64(state))) % (m - 1)
        if cand not in out:
            out.append(cand)
    return np.array(sorted(out), dtype=np.int64)


def _band_value(seed, r, k, dtype):
    """-(0.02 + 0.98 u(hash(r,k))), u in [0,1) with 24 random bits; r = min(i,j), k = offset index."""
    with np.errstate(over="ignore"):
        key = np.uint64(seed) + (r.astype(np.uint64) * np.uint64(64) + np.uint64(k + 1)) * np.uint64(_GOLD)
    h = _mix64(key)
    u = ((h >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)).astype(dtype)
    scaled = dtype(0.98) * u
    return -(dtype(0.02) + scaled)


def banded_random_spd(n, k=25, seed=0x5EED, max_offset=1 << 20, dtype=np.float32, diag_shift=1.0):
    """Banded-random symmetric strictly diagonally dominant (hence SPD) matrix, ~2k+1 nonzeros per row.

    For every row i and offset d_k: A[i][i-d_k] and A[i][i+d_k] (when inside the matrix) hold
    -(0.02+0.98u(hash(min(i,j),k))); A[i][i] = diag_shift + sum |offdiag| accumulated in ascending column order
    (diag_shift = 1 is SURVEY.md's law).  Every row sums to diag_shift, so the all-ones vector is the eigenvector of
    the smallest eigenvalue diag_shift and the condition number is at most (diag_shift + 2 sum|offdiag|) / diag_shift.
    """
    dtype = np.dtype(dtype).type
    offs = band_offsets(n, k, seed, max_offset)
    K = len(offs)
    rows = np.arange(n, dtype=np.int64)
    width = 2 * K + 1
    cols = np.empty((n, width), dtype=np.int64)
    vals = np.zeros((n, width), dtype=dtype)
    valid = np.zeros((n, width), dtype=bool)
    diag = np.full(n, dtype(diag_shift))
    slot = 0
    for kk in range(K - 1, -1, -1):  # columns i - d_k ascending
        j = rows - offs[kk]
        ok = j >= 0
        v = _band_value(seed, np.where(ok, j, 0), kk, dtype)
        cols[:, slot] = j
        vals[:, slot] = v
        valid[:, slot] = ok
        diag = np.where(ok, diag + (-v), diag)
        slot += 1
    diag_slot = slot
    slot += 1
    for kk in range(K):  # columns i + d_k ascending
        j = rows + offs[kk]
        ok = j < n
        v = _band_value(seed, rows, kk, dtype)
        cols[:, slot] = j
        vals[:, slot] = v
        valid[:, slot] = ok
        diag = np.where(ok, diag + (-v), diag)
        slot += 1
    cols[:, diag_slot] = rows
    vals[:, diag_slot] = diag
    valid[:, diag_slot] = True
    counts = valid.sum(axis=1)
    start = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=start[1:])
    return start.astype(np.int32), cols[valid].astype(np.int32), vals[valid].astype(dtype)


def stencil3d(nx, ny, nz, diag=6.0, lo=-1.0, hi=-1.0, dtype=np.float64):
    """7-point stencil on an nx*ny*nz grid (x fastest): `lo` on the three lower neighbours, `hi` on the upper ones.

    (6,-1,-1) is the 3-D Laplacian of config 4; (6,-1-c,-1+c) is the convection-diffusion stand-in for atmosmodd.
    """
    dtype = np.dtype(dtype).type
    n = nx * ny * nz
    i = np.arange(n, dtype=np.int64)
    ix = i % nx
    iy = (i // nx) % ny
    iz = i // (nx * ny)
    plane = nx * ny
    cand = [
        (i - plane, iz > 0, lo),
        (i - nx, iy > 0, lo),
        (i - 1, ix > 0, lo),
        (i, np.ones(n, dtype=bool), diag),
        (i + 1, ix < nx - 1, hi),
        (i + nx, iy < ny - 1, hi),
        (i + plane, iz < nz - 1, hi),
    ]
    cols = np.stack([c for c, _, _ in cand], axis=1)
    valid = np.stack([m for _, m, _ in cand], axis=1)
    vals = np.broadcast_to(np.array([v for _, _, v in cand], dtype=dtype), (n, 7))
    start = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(valid.sum(axis=1), out=start[1:])
    return start.astype(np.int32), cols[valid].astype(np.int32), vals[valid].astype(dtype)


def stencil3d_wide(nx, ny, nz, points=27, dtype=np.float64):
    """19- or 27-point stencil on an nx*ny*nz grid (x fastest, Dirichlet truncation): every neighbour (dx, dy, dz) in {-1, 0, 1}^3 -- for 19
    points without the eight corners.  27 points with diag 26 and -1 elsewhere is HPCG's matrix; here each of the three neighbour classes
    (face / edge / corner) has its own constant so that a wrong pairing of value and column shows.  Its far offsets come in CLUSTERS around
    -nx*ny and +nx*ny: what the three-window march kernel (csrc/smm_spmv_march.hip, spmvPatternConstMarch3Kernel) is for."""
    assert points in (19, 27)
    dtype = np.dtype(dtype).type
    n = nx * ny * nz
    i = np.arange(n, dtype=np.int64)
    ix, iy, iz = i % nx, (i // nx) % ny, i // (nx * ny)
    cols, valid, vals = [], [], []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                order = abs(dx) + abs(dy) + abs(dz)
                if points == 19 and order == 3:
                    continue
                ok = (ix + dx >= 0) & (ix + dx < nx) & (iy + dy >= 0) & (iy + dy < ny) & (iz + dz >= 0) & (iz + dz < nz)
                cols.append(i + dz * nx * ny + dy * nx + dx)
                valid.append(ok)
                vals.append({0: float(points - 1), 1: -1.0, 2: -0.5, 3: -0.25}[order])
    cols = np.stack(cols, axis=1)
    valid = np.stack(valid, axis=1)
    vals = np.broadcast_to(np.array(vals, dtype=dtype), cols.shape)
    start = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(valid.sum(axis=1), out=start[1:])
    return start.astype(np.int32), cols[valid].astype(np.int32), vals[valid].astype(dtype)


def poisson3d(n, dtype=np.float64):
    return stencil3d(n, n, n, 6.0, -1.0, -1.0, dtype)


def convdiff3d(n, c=0.3, dtype=np.float64):
    """Non-symmetric 7-point convection-diffusion operator (config 5 stand-in): diag 6, lower -1-c, upper -1+c."""
    return stencil3d(n, n, n, 6.0, -1.0 - c, -1.0 + c, dtype)


def convdiff3d_varying(n, c=0.3, dtype=np.float64):
    """The 7-point convection-diffusion operator of config 5's stand-in with SPATIALLY VARYING coefficients, as a real atmosmodd-like
    matrix has them: diffusion 1, a smooth rotating velocity field (u, v, w)(x, y, z) of amplitude c discretised with central
    differences (west / east = -1 -+ u, south / north = -1 -+ v, down / up = -1 -+ w) and a reaction term 0 <= r < 0.1 on the diagonal
    (6 + r).  Non-symmetric, every diagonal varies from row to row: the SpMV needs values[] (PATTERN / MASKS, not CONST)."""
    dtype = np.dtype(dtype).type
    N = n * n * n
    i = np.arange(N, dtype=np.int64)
    ix, iy, iz = i % n, (i // n) % n, i // (n * n)
    fx, fy, fz = (ix + 0.5) / n, (iy + 0.5) / n, (iz + 0.5) / n
    two_pi = 2.0 * np.pi
    u = c * np.sin(two_pi * fy) * np.cos(two_pi * fz)
    v = c * np.sin(two_pi * fz) * np.cos(two_pi * fx)
    w = c * np.sin(two_pi * fx) * np.cos(two_pi * fy)
    r = 0.1 * fx * fy * fz
    cand = [
        (i - n * n, iz > 0, -1.0 - w),
        (i - n, iy > 0, -1.0 - v),
        (i - 1, ix > 0, -1.0 - u),
        (i, np.ones(N, dtype=bool), 6.0 + r),
        (i + 1, ix < n - 1, -1.0 + u),
        (i + n, iy < n - 1, -1.0 + v),
        (i + n * n, iz < n - 1, -1.0 + w),
    ]
    cols = np.stack([cc for cc, _, _ in cand], axis=1)
    valid = np.stack([m for _, m, _ in cand], axis=1)
    vals = np.stack([vv for _, _, vv in cand], axis=1)
    start = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(valid.sum(axis=1), out=start[1:])
    return start.astype(np.int32), cols[valid].astype(np.int32), vals[valid].astype(dtype)


def poisson2d(nx, ny=None, dtype=np.float64):
    """5-point Laplacian (diag 4, off-diagonals -1, Dirichlet truncation) -- configs 1 and 2 with nx = ny = 1000."""
    ny = nx if ny is None else ny
    dtype = np.dtype(dtype).type
    n = nx * ny
    i = np.arange(n, dtype=np.int64)
    ix = i % nx
    iy = i // nx
    cand = [
        (i - nx, iy > 0, -1.0),
        (i - 1, ix > 0, -1.0),
        (i, np.ones(n, dtype=bool), 4.0),
        (i + 1, ix < nx - 1, -1.0),
        (i + nx, iy < ny - 1, -1.0),
    ]
    cols = np.stack([c for c, _, _ in cand], axis=1)
    valid = np.stack([m for _, m, _ in cand], axis=1)
    vals = np.broadcast_to(np.array([v for _, _, v in cand], dtype=dtype), (n, 5))
    start = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(valid.sum(axis=1), out=start[1:])
    return start.astype(np.int32), cols[valid].astype(np.int32), vals[valid].astype(dtype)


def random_rows(rows, cols, min_len, max_len, seed=1, dtype=np.float64, empty_every=0, diag_dominant=False):
    """Ragged random CSR for edge-case tests: row lengths uniform in [min_len, max_len], sorted distinct columns,
    optionally every `empty_every`-th row empty.  With diag_dominant the matrix is square with a dominant diagonal."""
    rng = np.random.default_rng(seed)
    dtype = np.dtype(dtype).type
    lens = rng.integers(min_len, max_len + 1, size=rows)
    if empty_every:
        lens[::empty_every] = 0
    lens = np.minimum(lens, cols)
    start = np.zeros(rows + 1, dtype=np.int64)
    np.cumsum(lens, out=start[1:])
    positions = np.empty(start[-1], dtype=np.int32)
    values = np.empty(start[-1], dtype=dtype)
    for r in range(rows):
        ln = lens[r]
        if ln == 0:
            continue
        c = np.sort(rng.choice(cols, size=ln, replace=False))
        if diag_dominant and r < cols and r not in c:
            c[rng.integers(0, ln)] = r
            c = np.unique(c)
            if len(c) < ln:  # collision removed an entry; refill deterministically
                extra = np.setdiff1d(np.arange(cols), c)[: ln - len(c)]
                c = np.sort(np.concatenate([c, extra]))
        v = rng.uniform(-1.0, 1.0, size=ln).astype(dtype)
        if diag_dominant and r < cols:
            v[np.searchsorted(c, r)] = dtype(np.abs(v).sum() + 1.0)
        positions[start[r]:start[r + 1]] = c
        values[start[r]:start[r + 1]] = v
    return start.astype(np.int32), positions, values


def row_sums(start, values):
    """RHS convention of the reference's solver tests: b = row sums of A, so the exact solution is all ones
    (sumColumsPerRow, test/include/test_common.h:13-21) -- accumulated left to right in the matrix dtype."""
    rows = len(start) - 1
    out = np.zeros(rows, dtype=values.dtype)
    lens = np.diff(start)
    maxlen = int(lens.max()) if rows else 0
    for j in range(maxlen):
        m = lens > j
        out[m] = out[m] + values[start[:-1][m] + j]
    return out
