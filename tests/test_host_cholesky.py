"""Host sparse Cholesky (multical_amd/csrc/host/block_cholesky.hpp, the stand-in for
sparse_block_matrix::LinearSolverCholmod, linear_solver_cholmod.h:70-112) through its C entry points.

The reference pins its solvers only against each other and against dense algebra (sparse_block_matrix/test/solver_tests.cpp;
aslam_backend/test/LinearSolverTests.cpp:42-57: dx equal to 1e-6); the same is done here against numpy / scipy on matrices
of the shape the hot path produces: block-banded spline part, a border of global blocks every row touches, blocks of
dimension 6/3/2/1, upper-triangular scalar CCS in fillCCS order."""
import ctypes as C
import os

import numpy as np
import pytest
import scipy.sparse as sp

from multical_amd import _lib


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Chol(object):
    def __init__(self):
        self.lib = _lib.load()
        self.h = self.lib.mc_cholesky_create()

    def __del__(self):
        self.lib.mc_cholesky_destroy(self.h)

    def analyze(self, U, base=None):
        n = U.shape[0]
        self.n = n
        Cp, Ci = U.indptr.astype(np.int32), U.indices.astype(np.int32)
        b = None if base is None else np.ascontiguousarray(base, np.int32)
        rc = self.lib.mc_cholesky_analyze(self.h, n, _ip(Cp), _ip(Ci), 0 if b is None else len(b) - 1, None if b is None else _ip(b))
        assert rc == 0, self.lib.mc_cholesky_last_error(self.h)

    def factorize(self, Cx):
        return self.lib.mc_cholesky_factorize(self.h, _dp(np.ascontiguousarray(Cx, np.float64)))

    def solve(self, b):
        x = np.zeros(self.n)
        assert self.lib.mc_cholesky_solve(self.h, _dp(np.ascontiguousarray(b, np.float64)), _dp(x)) == 0
        return x

    def stats(self):
        nd, nnz, fl = C.c_int(), C.c_longlong(), C.c_double()
        self.lib.mc_cholesky_stats(self.h, C.byref(nd), C.byref(nnz), C.byref(fl))
        return nd.value, nnz.value, fl.value


def arrowhead_problem(rng, n_band=300, band=7, dims=(6, 3), n_global=5, global_dims=(3, 3, 1, 2, 4), extra_pairs=40):
    """Random J (rows = 'terms' touching a window of consecutive band blocks + some globals) -> H = J^T J + I, with the
    block partition; returns (upper CCS as scipy csc with sorted indices, block bases, dense H)."""
    bd = [dims[i % len(dims)] for i in range(n_band)] + list(global_dims[:n_global])
    base = np.concatenate([[0], np.cumsum(bd)]).astype(np.int32)
    nb, n = len(bd), int(base[-1])
    pat = np.zeros((nb, nb), bool)
    H = np.eye(n)
    for t in range(4 * n_band):
        s = rng.integers(0, n_band - band)
        blocks = list(range(s, s + rng.integers(2, band + 1)))
        blocks += [n_band + g for g in range(n_global) if rng.random() < 0.5]
        cols = np.concatenate([np.arange(base[b], base[b + 1]) for b in blocks])
        Jr = rng.standard_normal((2, len(cols)))
        H[np.ix_(cols, cols)] += Jr.T @ Jr
        for a in blocks:
            for b in blocks:
                pat[a, b] = True
    for _ in range(extra_pairs):        # structurally present but numerically zero blocks (padded spline coefficients)
        a, b = rng.integers(0, nb, 2)
        pat[a, b] = pat[b, a] = True
    np.fill_diagonal(pat, True)
    mask = np.zeros((n, n), bool)
    for a in range(nb):
        for b in range(a, nb):
            if pat[a, b]:
                mask[base[a]:base[a + 1], base[b]:base[b + 1]] = True
    mask = np.triu(mask)
    rows, cols = np.nonzero(mask.T)      # column-major enumeration: (col, row) sorted
    U = sp.csc_matrix((H[cols, rows], (cols, rows)), shape=(n, n))   # keeps explicit zeros
    U.sort_indices()
    return U, base, H


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_solution_matches_dense(seed):
    rng = np.random.default_rng(seed)
    U, base, H = arrowhead_problem(rng)
    c = Chol()
    c.analyze(U, base)
    assert c.factorize(U.data) == 0
    b = rng.standard_normal(U.shape[0])
    x = c.solve(b)
    xr = np.linalg.solve(H, b)
    assert np.abs(x - xr).max() <= 1e-9 * np.abs(xr).max()
    # a second factorisation on new values with the same analysis (what every LM iteration does)
    H2 = H + np.diag(rng.random(H.shape[0]) * 10.0)
    U2 = U.copy()
    U2.data = H2[U.indices, np.repeat(np.arange(U.shape[0]), np.diff(U.indptr))]
    assert c.factorize(U2.data) == 0
    x2 = c.solve(b)
    assert np.abs(x2 - np.linalg.solve(H2, b)).max() <= 1e-9 * np.abs(x2).max()


def test_scalar_partition_and_block_partition_agree():
    rng = np.random.default_rng(5)
    U, base, H = arrowhead_problem(rng, n_band=120)
    b = rng.standard_normal(U.shape[0])
    c1, c2 = Chol(), Chol()
    c1.analyze(U, base)
    c2.analyze(U, None)
    assert c1.factorize(U.data) == 0 and c2.factorize(U.data) == 0
    x1, x2 = c1.solve(b), c2.solve(b)
    assert np.abs(x1 - x2).max() <= 1e-10 * np.abs(x1).max()


def test_not_positive_definite_is_reported_not_raised():
    rng = np.random.default_rng(7)
    U, base, H = arrowhead_problem(rng, n_band=60)
    c = Chol()
    c.analyze(U, base)
    bad = U.data.copy()
    diag_pos = U.indptr[1:] - 1
    bad[diag_pos[17]] = -1.0
    assert c.factorize(bad) == 1           # LinearSolverCholmod::solve returns false (linear_solver_cholmod.h:91-100)
    bad[diag_pos[17]] = np.nan
    assert c.factorize(bad) == 1
    assert c.factorize(U.data) == 0        # and the solver object stays usable
    b = rng.standard_normal(U.shape[0])
    assert np.abs(c.solve(b) - np.linalg.solve(H, b)).max() <= 1e-9


def test_fill_stays_near_the_band_with_dense_border():
    """The ordering must not let the global blocks (touched by every row) fill the band: factor size stays within a small
    multiple of band + border."""
    rng = np.random.default_rng(11)
    U, base, H = arrowhead_problem(rng, n_band=1500, band=7, n_global=5, extra_pairs=0)
    c = Chol()
    c.analyze(U, base)
    nd, nnz, flops = c.stats()
    n = U.shape[0]
    assert nnz < n * (7 * 6 + 13 + 20) * 2.0, (nnz, n)
    assert c.factorize(U.data) == 0
    b = rng.standard_normal(n)
    x = c.solve(b)
    A = sp.csc_matrix(U + sp.triu(U, 1).T)
    assert np.abs(A @ x - b).max() <= 1e-8 * np.abs(b).max()


def test_rejects_a_lower_triangular_pattern():
    c = Chol()
    L = sp.csc_matrix(np.tril(np.ones((4, 4))))
    Cp, Ci = L.indptr.astype(np.int32), L.indices.astype(np.int32)
    rc = c.lib.mc_cholesky_analyze(c.h, 4, _ip(Cp), _ip(Ci), 0, None)
    assert rc == -1 and b"upper" in c.lib.mc_cholesky_last_error(c.h)


def test_covariance_blocks_match_the_dense_inverse():
    """LinearSolverCholmod::solvePattern: selected blocks of H^-1 (diagonal and off-diagonal, band and border)."""
    rng = np.random.default_rng(21)
    U, base, H = arrowhead_problem(rng, n_band=80)
    c = Chol()
    c.analyze(U, base)
    assert c.factorize(U.data) == 0
    nb = len(base) - 1
    pairs = [(nb - 1, nb - 1), (nb - 2, nb - 1), (3, 3), (3, nb - 1), (10, 4), (nb - 3, 7)]
    bi, bj = np.array([p[0] for p in pairs], np.int32), np.array([p[1] for p in pairs], np.int32)
    sizes = (base[bi + 1] - base[bi]).astype(np.int64) * (base[bj + 1] - base[bj])
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    out = np.zeros(int(off[-1]))
    rc = c.lib.mc_cholesky_covariance_blocks(c.h, len(pairs), _ip(bi), _ip(bj), off.ctypes.data_as(C.POINTER(C.c_longlong)), _dp(out))
    assert rc == 0, c.lib.mc_cholesky_last_error(c.h)
    Pinv = np.linalg.inv(H)
    for k, (i, j) in enumerate(pairs):
        blk = out[off[k]:off[k + 1]].reshape(base[i + 1] - base[i], base[j + 1] - base[j])
        want = Pinv[base[i]:base[i + 1], base[j]:base[j + 1]]
        assert np.abs(blk - want).max() <= 1e-9 * np.abs(Pinv).max()


# ---- time-axis dissection (independent parts, one thread each) and the numerical pattern ------------------------------
def _coords(base, n_band):
    """band blocks get their position, the global blocks none"""
    t = np.full(len(base) - 1, np.nan)
    t[:n_band] = np.arange(n_band, dtype=float)
    return t


@pytest.mark.parametrize("threads", [1, 2, 4, 7])
def test_dissection_matches_dense(threads):
    rng = np.random.default_rng(31 + threads)
    n_band = 700
    U, base, H = arrowhead_problem(rng, n_band=n_band)
    c = Chol()
    c.analyze(U, base)
    assert c.lib.mc_cholesky_set_block_coordinates(c.h, _dp(_coords(base, n_band)), threads) == 0
    assert c.factorize(U.data) == 0
    b = rng.standard_normal(U.shape[0])
    xr = np.linalg.solve(H, b)
    assert np.abs(c.solve(b) - xr).max() <= 1e-9 * np.abs(xr).max()
    perm = np.zeros(len(base) - 1, np.int32)
    assert c.lib.mc_cholesky_block_permutation(c.h, _ip(perm)) == len(perm)
    assert sorted(perm.tolist()) == list(range(len(perm)))
    # new values, same structure; then a matrix that is not positive definite; then a good one again
    H2 = H + np.diag(rng.random(H.shape[0]) * 3.0)
    v2 = H2[U.indices, np.repeat(np.arange(U.shape[0]), np.diff(U.indptr))]
    assert c.factorize(v2) == 0
    assert np.abs(c.solve(b) - np.linalg.solve(H2, b)).max() <= 1e-9 * np.abs(xr).max()
    bad = v2.copy()
    bad[U.indptr[1:][U.shape[0] // 2] - 1] = -5.0
    assert c.factorize(bad) == 1
    assert c.factorize(v2) == 0


def test_dissection_covariance_blocks():
    rng = np.random.default_rng(77)
    n_band = 400
    U, base, H = arrowhead_problem(rng, n_band=n_band)
    c = Chol()
    c.analyze(U, base)
    assert c.lib.mc_cholesky_set_block_coordinates(c.h, _dp(_coords(base, n_band)), 4) == 0
    assert c.factorize(U.data) == 0
    nb = len(base) - 1
    pairs = [(nb - 1, nb - 1), (5, 5), (200, nb - 2), (399, 398)]
    bi, bj = np.array([p[0] for p in pairs], np.int32), np.array([p[1] for p in pairs], np.int32)
    sizes = (base[bi + 1] - base[bi]).astype(np.int64) * (base[bj + 1] - base[bj])
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    out = np.zeros(int(off[-1]))
    assert c.lib.mc_cholesky_covariance_blocks(c.h, len(pairs), _ip(bi), _ip(bj), off.ctypes.data_as(C.POINTER(C.c_longlong)), _dp(out)) == 0
    Pinv = np.linalg.inv(H)
    for k, (i, j) in enumerate(pairs):
        blk = out[off[k]:off[k + 1]].reshape(base[i + 1] - base[i], base[j + 1] - base[j])
        assert np.abs(blk - Pinv[base[i]:base[i + 1], base[j]:base[j + 1]]).max() <= 1e-9 * np.abs(Pinv).max()


@pytest.mark.parametrize("threads", [1, 3])
def test_numerically_zero_blocks_are_left_out_and_come_back(threads):
    """The structurally present, numerically zero blocks (padded spline coefficients) do not enter the factor; when one of
    them becomes non-zero in a later matrix the structure grows and the answer is still right."""
    rng = np.random.default_rng(41)
    n_band = 500
    U, base, H = arrowhead_problem(rng, n_band=n_band, extra_pairs=300)
    full = Chol()
    full.analyze(U, base)
    nnz_full = full.stats()[1]                      # before any factorize: the structure of the whole pattern
    c = Chol()
    c.analyze(U, base)
    assert c.lib.mc_cholesky_set_block_coordinates(c.h, _dp(_coords(base, n_band)), threads) == 0
    assert c.factorize(U.data) == 0
    assert c.stats()[1] < nnz_full                  # the zero blocks were dropped
    b = rng.standard_normal(U.shape[0])
    xr = np.linalg.solve(H, b)
    assert np.abs(c.solve(b) - xr).max() <= 1e-9 * np.abs(xr).max()
    # switch on a stored-but-zero block far from the diagonal
    cols = np.repeat(np.arange(U.shape[0]), np.diff(U.indptr))
    zero = np.nonzero((U.data == 0.0) & (cols - U.indices > 40))[0]
    assert zero.size > 0
    k = zero[zero.size // 2]
    i, j = U.indices[k], cols[k]
    H3 = H.copy()
    H3[i, j] = H3[j, i] = 0.25
    H3[i, i] += 1.0; H3[j, j] += 1.0
    v3 = H3[U.indices, cols]
    assert c.factorize(v3) == 0
    x3 = np.linalg.solve(H3, b)
    assert np.abs(c.solve(b) - x3).max() <= 1e-9 * np.abs(x3).max()
    # and back on the first matrix: blocks are never dropped again, the result is unchanged
    assert c.factorize(U.data) == 0
    assert np.abs(c.solve(b) - xr).max() <= 1e-9 * np.abs(xr).max()


def test_linked_update_kernel_is_the_vector_one():
    """The factorisation is header-only and two translation units instantiate it (mc_host.cpp with g++, mc_driver.hip
    with hipcc); the linker keeps one copy.  That copy must be the AVX2 / FMA one: the one compiled without -march
    gives the same results three times slower, which no other test notices."""
    import shutil
    import subprocess
    from multical_amd import _lib
    if not (shutil.which("nm") and shutil.which("objdump")):
        pytest.skip("binutils not available")
    lib = _lib.LIB_PATH
    syms = [l.split()[-1] for l in subprocess.check_output(["nm", lib], text=True).splitlines()
            if "BlockSparseCholesky10updateTileILi6ELi2" in l]
    assert syms, "update kernel not found in " + lib
    dis = subprocess.check_output(["objdump", "-d", "--no-show-raw-insn", "--disassemble=" + syms[0], lib], text=True)
    assert "vfnmadd" in dis and "ymm" in dis


def test_both_vector_widths_of_the_update_kernel():
    """The panel update has a 256-bit and a 512-bit form, chosen once per process from the CPU (MC_CHOLESKY_NO_AVX512
    forces the narrow one).  Whatever this host picks, the other one is run too: the dense-solve and dissection tests
    again in a fresh process with the switch set, and the 512-bit tile must be in the library for hosts that have it."""
    import shutil
    import subprocess
    import sys
    from multical_amd import _lib
    env = dict(os.environ, MC_CHOLESKY_NO_AVX512="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", __file__, "-k",
                        "solution_matches_dense or dissection_matches_dense or come_back"], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    if shutil.which("nm") and shutil.which("objdump"):
        syms = [l.split()[-1] for l in subprocess.check_output(["nm", _lib.LIB_PATH], text=True).splitlines()
                if "BlockSparseCholesky13updateTile512ILi6ELi3" in l]
        assert syms, "512-bit update kernel not found in " + _lib.LIB_PATH
        dis = subprocess.check_output(["objdump", "-d", "--no-show-raw-insn", "--disassemble=" + syms[0], _lib.LIB_PATH], text=True)
        assert "vfnmadd" in dis and "zmm" in dis


@pytest.mark.parametrize("threads", [1, 3])
def test_zero_block_inside_the_fill_needs_no_new_structure(threads):
    """A stored-but-zero block whose place exists in the factor anyway (it lies inside the fill) is mapped from the start:
    when it becomes non-zero the factor structure stays as it is and the answer is right."""
    rng = np.random.default_rng(41)
    n_band = 500
    U, base, H = arrowhead_problem(rng, n_band=n_band, extra_pairs=300)
    cols = np.repeat(np.arange(U.shape[0]), np.diff(U.indptr))
    near = np.nonzero((U.data == 0.0) & (cols > U.indices) & (cols - U.indices <= 40))[0]
    b = rng.standard_normal(U.shape[0])
    absorbed = 0
    for k in near[:: max(1, near.size // 10)]:
        c = Chol()
        c.analyze(U, base)
        assert c.lib.mc_cholesky_set_block_coordinates(c.h, _dp(_coords(base, n_band)), threads) == 0
        assert c.factorize(U.data) == 0
        nnz0 = c.stats()[1]
        i, j = U.indices[k], cols[k]
        H3 = H.copy()
        H3[i, j] = H3[j, i] = 0.25
        H3[i, i] += 1.0; H3[j, j] += 1.0
        assert c.factorize(H3[U.indices, cols]) == 0
        x3 = np.linalg.solve(H3, b)
        assert np.abs(c.solve(b) - x3).max() <= 1e-9 * np.abs(x3).max()
        absorbed += c.stats()[1] == nnz0
    assert absorbed > 0
