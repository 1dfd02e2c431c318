"""The product's closed-form term math (mc_terms.h, compiled with g++ by tests/hostprobe) against the DAG-walking oracle,
term by term, plus bit-exact pattern / indexing.  Runs without a GPU."""
import numpy as np
import pytest

from probe_api import ProbeApi


@pytest.fixture(scope="module")
def probe(small_problem):
    p = ProbeApi()
    small_problem["spec"].emit(p)
    p.finalize()
    return p


def test_dimensions_and_block_indices(probe, small_oracle):
    o, ids = small_oracle
    assert (probe.j_rows, probe.j_cols, probe.n_blocks, probe.n_terms) == (o.j_rows, o.j_cols, o.n_blocks, o.n_terms)


def test_pattern_is_bit_exact(probe, small_oracle):
    o, _ = small_oracle
    o.evaluate_error(True)
    Cp, Ci, Cx, rhs = o.build_system(True)
    Cp2, Ci2 = probe.pattern()
    assert np.array_equal(Cp, Cp2)
    assert np.array_equal(Ci, Ci2)


@pytest.mark.parametrize("use_mest", [True, False])
def test_every_term_matches_oracle(probe, small_oracle, use_mest):
    o, _ = small_oracle
    o.evaluate_error(use_mest)
    chi = o.term_chi2()
    worst_j = worst_e = worst_c = 0.0
    for t in range(o.n_terms):
        if probe.term_kind(t) == 5:
            continue
        Ja, ea = o.term_weighted_jacobian(t, use_mest)
        Jp, ep, c = probe.term(t, use_mest)
        worst_j = max(worst_j, np.abs(Ja - Jp).max() / max(np.abs(Ja).max(), 1e-300))
        worst_e = max(worst_e, np.abs(ea - ep).max() / max(1e-9, np.abs(ea).max()))
        worst_c = max(worst_c, abs(c - chi[t]) / max(1e-9, chi[t]))
    assert worst_j < 1e-9 and worst_e < 1e-9 and worst_c < 1e-9, (worst_j, worst_e, worst_c)


def test_sharding_partitions_terms(small_problem):
    flags = []
    for r in range(3):
        p = ProbeApi()
        p.set_shard(r, 3)
        small_problem["spec"].emit(p)
        p.finalize()
        flags.append(p.my_terms())
    total = np.sum(flags, 0)
    assert np.all(total == 1)
    sizes = [int(f.sum()) for f in flags]
    assert min(sizes) > 0.2 * max(sizes), sizes


@pytest.mark.parametrize("threads", [1, 4])
def test_cpu_optimised_baseline_equals_the_oracle(small_problem, small_oracle, threads):
    """bench.py's `cpu_optimised` baseline (the product's formulation compiled for the host, OpenMP over work items) must
    compute what the oracle computes: same cost, same H and rhs (1e-9), on 1 and on 4 threads."""
    from probe_api import ProbeApi
    o, _ = small_oracle
    p = ProbeApi()
    small_problem["spec"].emit(p)
    p.finalize()
    Jo, _ = o.evaluate_error(True)
    Cp, Ci, Cx, rhs = o.build_system(True)
    J, Cxp, rhsp = p.cpu_optimised(True, True, threads)
    assert abs(J - Jo) <= 1e-9 * Jo
    assert np.abs(Cxp - Cx).max() <= 1e-9 * np.abs(Cx).max()
    assert np.abs(rhsp - rhs).max() <= 1e-9 * np.abs(rhs).max()
    assert np.all(Cxp[Cx == 0.0] == 0.0)
    J2, _, _ = p.cpu_optimised(True, False, threads)
    assert abs(J2 - Jo) <= 1e-9 * Jo


def test_motion_terms_are_split_over_the_shards_by_coefficient(small_problem):
    """A BSplineMotionError term covers a whole spline; a rank takes the band columns, the rows of Q c and the cost share
    of the coefficients in its window (Problem::motionRange).  Three shards of the small problem (motion priors on every
    bias spline): the partial systems sum to the unsharded one, entries outside the shared set have one writer, and the
    band of the motion terms does not end up in the shared set (it did while rank 0 took the whole term)."""
    whole = ProbeApi()
    small_problem["spec"].emit(whole)
    whole.finalize()
    J0, Cx0, rhs0 = whole.cpu_optimised(True, True, 2)
    J = 0.0
    Cx = np.zeros_like(Cx0)
    rhs = np.zeros_like(rhs0)
    writers = np.zeros(Cx0.size, int)
    shared = None
    for r in range(3):
        p = ProbeApi()
        p.set_shard(r, 3)
        small_problem["spec"].emit(p)
        p.finalize()
        Jr, Cxr, rhsr = p.cpu_optimised(True, True, 2)
        J += Jr
        Cx += Cxr
        rhs += rhsr
        writers += Cxr != 0.0
        rhs_writers = (rhs_writers if r else 0) + (rhsr != 0.0)
        # mc_set_clear_mode's table: every value this rank writes lies inside its runs, the runs are ascending and disjoint,
        # and they are a part of the matrix, not all of it
        runs, erows = p.written_runs()
        inside = np.zeros(Cx0.size, bool)
        for first, count in runs:
            assert count > 0 and not inside[first:first + count].any()
            inside[first:first + count] = True
        assert (np.diff(runs[:, 0]) > 0).all() and not (Cxr != 0.0)[~inside].any()
        assert inside.sum() < 0.75 * Cx0.size, (r, inside.sum(), Cx0.size)
        rows_in = np.zeros(p.j_rows, bool)
        for first, count in erows:
            rows_in[first:first + count] = True
        assert rows_in.sum() < p.j_rows
        written_rows = (written_rows if r else 0) + rows_in
        srhs = p.shared_rhs_indices()
        s = p.shared_indices()
        assert shared is None or np.array_equal(shared, s)      # the same list on every rank
        shared = s
    assert abs(J - J0) <= 1e-12 * J0
    assert np.abs(Cx - Cx0).max() <= 1e-12 * np.abs(Cx0).max()
    assert np.abs(rhs - rhs0).max() <= 1e-12 * np.abs(rhs0).max()
    assert (written_rows >= 1).all()         # every residual row is in some rank's runs
    private = np.ones(Cx0.size, bool)
    private[shared] = False
    assert writers[private].max() <= 1
    assert shared.size < 0.2 * Cx0.size, (shared.size, Cx0.size)
    # the same for rhs: outside the shared columns (calibration parameters, window boundaries) an entry has one writer
    private_rhs = np.ones(rhs0.size, bool)
    private_rhs[srhs] = False
    assert 0 < srhs.size < 0.5 * rhs0.size and rhs_writers[private_rhs].max() <= 1
