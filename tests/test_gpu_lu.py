"""GPU tests of the batched LU / Newton-solve consumer (csrc/pj_lu.h through the C ABI: pj_lu_factor_dev,
pj_lu_solve_dev, pj_newton_solve_dev) against LAPACK on the host (scipy.linalg.lu_factor / numpy.linalg.solve).
The reference has no batched solver -- its users hand one state's Jacobian to a dense solver
(docs/examples.rst:106-170) -- so the checker is that solver: same pivot rows, factors to rounding, residuals."""
import numpy as np
import pytest

from conftest import MECHS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    torch.cuda.set_device(0)
    return torch


def _blocks(rng, n, nsp, dominant=False):
    a = rng.standard_normal((n, nsp, nsp)) * 10.0 ** rng.uniform(-3, 3, (n, 1, 1))
    if dominant:
        a += np.eye(nsp) * nsp * 10.0
    return a


def _to_aos(a):
    # (n, r, c) -> pyJac's per-state layout a[s, r + NSP*c]
    return np.ascontiguousarray(a.transpose(0, 2, 1).reshape(a.shape[0], -1))


def _from_aos(x, nsp):
    return x.reshape(-1, nsp, nsp).transpose(0, 2, 1)


def _scipy_perm(piv):
    perm = np.arange(piv.size)
    for k, p in enumerate(piv):
        perm[k], perm[p] = perm[p], perm[k]
    return perm


@pytest.mark.parametrize('nsp', [1, 2, 7, 8, 9, 10, 16, 17, 24, 32, 33, 48, 53, 54, 55, 56, 57, 64, 65, 80, 81, 96, 97, 100, 111, 112, 113, 128, 129, 140])
def test_lu_factor_matches_lapack(nsp, torch_cuda):
    """P A = L U with LAPACK's pivot rows; factors agree to rounding; also through I - gamma A."""
    import scipy.linalg
    from pyjac_amd import linsolve
    torch = torch_cuda
    rng = np.random.default_rng(100 + nsp)
    n = 300 if nsp > 16 else 1000          # (65 .. 128 rows: k_lu4, four wavefronts per block; beyond: the LDS-resident kernel)
    a = _blocks(rng, n, nsp)
    for gamma in (0.0, 0.37):
        m = a if gamma == 0.0 else np.eye(nsp) - gamma * a
        lu, perm = linsolve.lu_factor(torch.from_numpy(_to_aos(a)).cuda(), gamma)
        lu, perm = _from_aos(lu.cpu().numpy(), nsp), perm.cpu().numpy()
        assert np.isfinite(lu).all()
        worst = 0.0
        for s in range(n):
            L = np.tril(lu[s], -1) + np.eye(nsp)
            U = np.triu(lu[s])
            assert sorted(perm[s]) == list(range(nsp))
            err = np.abs(L @ U - m[s][perm[s]]).max() / np.abs(m[s]).max()
            worst = max(worst, err)
            if s < 40:
                ref_lu, piv = scipy.linalg.lu_factor(m[s])
                # same pivot rows (ties have probability zero for random data) and the same factors to rounding
                assert np.array_equal(_scipy_perm(piv), perm[s]), (nsp, s)
                assert np.abs(lu[s] - ref_lu).max() <= 1e-10 * max(1.0, np.abs(ref_lu).max()), (nsp, s)
        assert worst < 1e-13 * nsp, (nsp, gamma, worst)


@pytest.mark.parametrize('nsp', [1, 3, 10, 17, 24, 32, 53, 64, 65, 111, 140])
def test_solves_match_lapack(nsp, torch_cuda):
    """pj_lu_solve_dev on stored factors and the fused factor + solve agree with numpy.linalg.solve."""
    from pyjac_amd import linsolve
    torch = torch_cuda
    rng = np.random.default_rng(200 + nsp)
    n = 500
    a = _blocks(rng, n, nsp, dominant=True)
    b = rng.standard_normal((n, nsp))
    ref = np.linalg.solve(a, b[:, :, None])[:, :, 0]
    d_a, d_b = torch.from_numpy(_to_aos(a)).cuda(), torch.from_numpy(b).cuda()
    lu, perm = linsolve.lu_factor(d_a)
    x1 = linsolve.lu_solve(lu, perm, d_b).cpu().numpy()
    x2 = linsolve.newton_solve(d_a, d_b).cpu().numpy()
    x3, lu3, perm3 = linsolve.newton_solve(d_a, d_b, keep_factors=True)
    # forward error bounded through the condition number (LAPACK's own answer carries cond * eps), and the residual
    # at rounding level whatever the conditioning (backward stability of partial pivoting)
    scale = np.abs(ref).max(axis=1, keepdims=True)
    cond = np.linalg.cond(a)[:, None]
    absA = np.abs(a)
    for x in (x1, x2, x3.cpu().numpy()):
        assert (np.abs(x - ref) <= 1e-13 * nsp * cond * scale).all()
        res = np.einsum('sij,sj->si', a, x) - b
        bound = np.einsum('sij,sj->si', absA, np.abs(x)) + np.abs(b)
        assert (np.abs(res) <= 1e-14 * nsp * bound.max(axis=1, keepdims=True)).all()
    assert torch.equal(lu3, lu) and torch.equal(perm3, perm)
    # in place: factors over the blocks, solution over the right-hand sides
    d_a2, d_b2 = d_a.clone(), d_b.clone()
    lu_ip, perm_ip = linsolve.lu_factor(d_a2, overwrite=True)
    assert lu_ip.data_ptr() == d_a2.data_ptr() and torch.equal(lu_ip, lu)
    linsolve.lu_solve(lu_ip, perm_ip, d_b2, out=d_b2)
    assert np.array_equal(d_b2.cpu().numpy(), x1)
    # Newton matrix I - gamma A
    gamma = 0.0037         # (0.01 would cancel the dominant diagonal 10 NSP of these blocks exactly: M ill-defined)
    refg = np.linalg.solve(np.eye(nsp) - gamma * a, b[:, :, None])[:, :, 0]
    xg = linsolve.newton_solve(d_a, d_b, gamma=gamma).cpu().numpy()
    mg = np.eye(nsp) - gamma * a
    condg = np.linalg.cond(mg)[:, None]
    assert (np.abs(xg - refg) <= 1e-13 * nsp * condg * np.abs(refg).max(axis=1, keepdims=True)).all()
    res = np.einsum('sij,sj->si', mg, xg) - b
    bound = np.einsum('sij,sj->si', np.abs(mg), np.abs(xg)) + np.abs(b)
    assert (np.abs(res) <= 1e-14 * nsp * bound.max(axis=1, keepdims=True)).all()


@pytest.mark.parametrize('nsp', [3, 10, 16, 24, 32, 53, 64, 72, 111])
def test_batch_layout_inputs(nsp, torch_cuda):
    """The state-fastest batch layout (what the Jacobian kernels write at full speed): factors and solutions are the
    same as from the per-state layout, bit for bit; n is not a multiple of anything."""
    import pyjac_amd
    from pyjac_amd import linsolve
    torch = torch_cuda
    rng = np.random.default_rng(300 + nsp)
    n = 997 if nsp <= 64 else 203
    a = _blocks(rng, n, nsp, dominant=True)
    b = rng.standard_normal((n, nsp))
    d_a, d_b = torch.from_numpy(_to_aos(a)).cuda(), torch.from_numpy(b).cuda()
    d_as, d_bs = d_a.T.contiguous(), d_b.T.contiguous()           # (NSP*NSP, n), (NSP, n)
    S = pyjac_amd.LAYOUT_SOA
    lu, perm = linsolve.lu_factor(d_a, 0.0037)
    lus, perms = linsolve.lu_factor(d_as, 0.0037, layout=S)
    assert torch.equal(lu, lus) and torch.equal(perm, perms)
    x = linsolve.newton_solve(d_a, d_b, gamma=0.0037)
    xs, lu2, perm2 = linsolve.newton_solve(d_as, d_bs, gamma=0.0037, layout=S, keep_factors=True)
    assert xs.shape == (nsp, n) and torch.equal(xs.T.contiguous(), x) and torch.equal(lu2, lu) and torch.equal(perm2, perm)
    x2 = linsolve.lu_solve(lu, perm, d_bs, layout=S)
    assert torch.equal(x2.T.contiguous(), linsolve.lu_solve(lu, perm, d_b))
    with pytest.raises(ValueError):
        linsolve.lu_factor(d_as, overwrite=True, layout=S)


def test_edge_cases(torch_cuda):
    from pyjac_amd import PyjacError, linsolve
    torch = torch_cuda
    # empty batch
    e = torch.empty((0, 100), dtype=torch.float64, device='cuda')
    lu, perm = linsolve.lu_factor(e)
    assert lu.shape == (0, 100) and perm.shape == (0, 10)
    # more rows than the LDS holds
    with pytest.raises(PyjacError):
        linsolve.lu_factor(torch.zeros((2, 141 * 141), dtype=torch.float64, device='cuda'))
    # a column of NaNs must not corrupt memory: permutation stays a permutation, neighbours untouched
    rng = np.random.default_rng(5)
    a = _blocks(rng, 3, 10)
    a[1, :, 4] = np.nan
    lu, perm = linsolve.lu_factor(torch.from_numpy(_to_aos(a)).cuda())
    perm = perm.cpu().numpy()
    assert all(sorted(p) == list(range(10)) for p in perm)
    lu = lu.cpu().numpy()
    assert np.isfinite(lu[0]).all() and np.isfinite(lu[2]).all()
    # pivoting is needed: zero diagonal
    z = np.array([[0.0, 2.0], [3.0, 1.0]])[None]
    x = linsolve.newton_solve(torch.from_numpy(_to_aos(z)).cuda(), torch.tensor([[2.0, 4.0]], dtype=torch.float64, device='cuda'))
    assert np.allclose(x.cpu().numpy(), np.linalg.solve(z[0], [2.0, 4.0]))


@pytest.mark.parametrize('name,n', [('h2o2_n2', 20000), ('gri30_shaped', 3000), ('usc2_shaped', 600)])
def test_newton_step_on_jacobians(name, n, torch_cuda):
    """The consumer on real input: Jacobians of the mechanism in pyJac's per-state layout (AoS), Newton matrix
    I - gamma J with an implicit-step-sized gamma, residual of the solve against the matrix rebuilt on the host."""
    import pyjac_amd
    from pyjac_amd import linsolve, synth
    torch = torch_cuda
    ev = pyjac_amd.Evaluator(MECHS[name])
    pres, y = (synth.dist_a if name == 'h2o2_n2' else synth.dist_b)(n, ev.nsp)
    d_p, d_y = torch.from_numpy(pres).cuda(), torch.from_numpy(np.ascontiguousarray(y.T)).cuda()
    ev.use_spec(2)
    jac = ev.jacobian(d_p, d_y, y_layout=pyjac_amd.LAYOUT_AOS, jac_layout=pyjac_amd.LAYOUT_AOS)
    rng = np.random.default_rng(3)
    b = rng.standard_normal((n, ev.nsp))
    gamma = 1e-7
    x = linsolve.newton_solve(jac, torch.from_numpy(b).cuda(), gamma=gamma).cpu().numpy()
    assert np.isfinite(x).all()
    J = _from_aos(jac.cpu().numpy()[:400], ev.nsp)
    M = np.eye(ev.nsp) - gamma * J
    res = np.einsum('sij,sj->si', M, x[:400]) - b[:400]
    # backward-stable: residual small against |M| |x| + |b|
    bound = np.einsum('sij,sj->si', np.abs(M), np.abs(x[:400])) + np.abs(b[:400])
    assert (np.abs(res) <= 1e-13 * ev.nsp * bound.max(axis=1, keepdims=True)).all()
    ref = np.linalg.solve(M, b[:400, :, None])[:, :, 0]
    cond = np.linalg.cond(M)[:, None]
    assert (np.abs(x[:400] - ref) <= 1e-13 * ev.nsp * cond * np.abs(ref).max(axis=1, keepdims=True)).all()
    # the same step straight from the batch layout (SoA Jacobians, SoA vectors): no transposed copy
    ev.use_spec(1)
    jac_soa = ev.jacobian(d_p, torch.from_numpy(y).cuda())
    xs = linsolve.newton_solve(jac_soa, torch.from_numpy(np.ascontiguousarray(b.T)).cuda(), gamma=gamma,
                               layout=pyjac_amd.LAYOUT_SOA).cpu().numpy().T
    assert (np.abs(xs[:400] - ref) <= 1e-13 * ev.nsp * cond * np.abs(ref).max(axis=1, keepdims=True)).all()
    # and as one call on the evaluator
    xe = ev.newton_solve(d_p, torch.from_numpy(y).cuda(), torch.from_numpy(np.ascontiguousarray(b.T)).cuda(), gamma)
    assert np.array_equal(xe.cpu().numpy().T, xs)


@pytest.mark.parametrize('nsp', [10, 24, 53, 64, 100, 111])
def test_pivot_ties(nsp, torch_cuda):
    """Exact ties of the column maximum (structured Newton matrices have them).  Every kernel picks A row of maximum
    magnitude (partial pivoting holds: |L| <= 1, P A = L U exactly on small-integer data).  Which one: the LDS-resident
    kernel (129 .. 140 rows) exchanges rows physically and takes the first row of maximum magnitude in the current
    order -- LAPACK dgetf2's choice, pivot for pivot (a lane scans rows 32 apart there, so the lowest LANE is not the
    lowest row: ADVICE round 3).  The register-resident kernels (<= 128 rows) never exchange rows: among tied rows
    they take the lowest ORIGINAL row, which is dgetf2's choice unless the tie involves a row that an earlier step
    displaced (include/pyjac_amd.h says so)."""
    import scipy.linalg
    from pyjac_amd import linsolve
    torch = torch_cuda
    rng = np.random.default_rng(7 + nsp)
    n = 64
    # small integers: every elimination step is exact in the first steps, so ties stay ties
    a = rng.integers(-2, 3, (n, nsp, nsp)).astype(np.float64)
    a[:, :, 0] = rng.choice([-3.0, 3.0], (n, nsp))          # column 0: every row ties
    if nsp > 40:
        a[:, 1, 0] = 1.0
        a[:, 0, 0] = 2.0                                     # the first maximum is row 2, its twins sit 32 rows further down
    lu, perm = linsolve.lu_factor(torch.from_numpy(_to_aos(a)).cuda())
    lu, perm = _from_aos(lu.cpu().numpy(), nsp), perm.cpu().numpy()
    checked = 0
    for s in range(n):
        with np.errstate(all='ignore'):
            ref_lu, piv = scipy.linalg.lu_factor(a[s], check_finite=False)
        if not np.isfinite(ref_lu).all() or np.abs(np.diag(ref_lu)).min() < 1e-9 or not np.isfinite(lu[s]).all():
            continue                                         # singular draw
        assert sorted(perm[s]) == list(range(nsp))
        L = np.tril(lu[s], -1) + np.eye(nsp)
        U = np.triu(lu[s])
        assert np.abs(L).max() <= 1.0 + 1e-12               # a row of maximum magnitude was the pivot, every step
        assert np.abs(L @ U - a[s][perm[s]]).max() <= 1e-9 * max(1.0, np.abs(U).max())
        assert perm[s][0] == (2 if nsp > 40 else 0)          # step 0: the first of the tied rows
        # (later steps divide by 3: the tied values are no longer exact ties, and which implementation's rounding breaks
        # them how is not a property of the pivot rule -- only the exact steps are compared with LAPACK)
        assert perm[s][0] == _scipy_perm(piv)[0]
        checked += 1
    assert checked > n // 2


def test_newton_solve_rejects_aliased_buffers(torch_cuda):
    """pj_newton_solve_dev: storing the factors over batch-layout blocks, or the solution over the blocks, is refused
    (PJ_EINVAL) instead of corrupting other states' blocks."""
    import ctypes
    from pyjac_amd import _lib
    torch = torch_cuda
    nsp, n = 10, 256
    a = torch.randn((nsp * nsp, n), dtype=torch.float64, device='cuda')
    b = torch.randn((nsp, n), dtype=torch.float64, device='cuda')
    perm = torch.empty((n, nsp), dtype=torch.int32, device='cuda')
    L = _lib.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = L.pj_newton_solve_dev(nsp, n, a.data_ptr(), _lib.LAYOUT_SOA, 0.0, b.data_ptr(), b.data_ptr(), _lib.LAYOUT_SOA,
                               a.data_ptr(), perm.data_ptr(), st)
    assert rc != 0                                           # d_lu == d_a in the batch layout
    rc = L.pj_newton_solve_dev(nsp, n, a.data_ptr(), _lib.LAYOUT_SOA, 0.0, b.data_ptr(), a.data_ptr(), _lib.LAYOUT_SOA,
                               None, None, st)
    assert rc != 0                                           # d_x inside d_a
    x = torch.empty_like(b)
    rc = L.pj_newton_solve_dev(nsp, n, a.data_ptr(), _lib.LAYOUT_SOA, 0.0, b.data_ptr(), x.data_ptr(), _lib.LAYOUT_SOA,
                               None, None, st)
    assert rc == 0
    torch.cuda.synchronize()
