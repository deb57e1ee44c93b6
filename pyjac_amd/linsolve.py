"""Batched LU / Newton solves on per-state Jacobian blocks (include/pyjac_amd.h: pj_lu_factor_dev,
pj_lu_solve_dev, pj_newton_solve_dev; kernel: csrc/pj_lu.h).

The consumer an implicit integrator puts behind ``eval_jacobian``: the reference hands one state's Jacobian
to the caller's dense solver (docs/examples.rst:106-170) and has no batched form.  Blocks are pyJac's
per-state C layout -- ``a[s, r + NSP*c]`` -- i.e. what ``Evaluator.jacobian(..., jac_layout=LAYOUT_AOS)``
returns.  Blocks of up to 64 rows are factored in registers (a lane per row), up to 140 rows in LDS (a workgroup
per block).  torch tensors carry the device memory; the work is the HIP kernel's.
"""
import ctypes

from . import _lib
from ._lib import check


def _stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _blocks(a, name='a'):
    import torch
    if not (isinstance(a, torch.Tensor) and a.is_cuda and a.dtype == torch.float64 and a.is_contiguous() and a.dim() == 2):
        raise ValueError('%s: expected a contiguous float64 CUDA tensor of shape (n, NSP*NSP)' % name)
    nsp = int(round(a.shape[1] ** 0.5))
    if nsp * nsp != a.shape[1]:
        raise ValueError('%s: second dimension is not a square' % name)
    return a.shape[0], nsp


def _vectors(b, n, nsp, device, name):
    import torch
    if not (isinstance(b, torch.Tensor) and b.is_cuda and b.dtype == torch.float64 and b.is_contiguous() and
            tuple(b.shape) == (n, nsp) and b.device == device):
        raise ValueError('%s: expected a contiguous float64 CUDA tensor of shape (%d, %d) on %s' % (name, n, nsp, device))


def lu_factor(a, gamma: float = 0.0, overwrite: bool = False):
    """P A = L U (or of I - gamma A) for every block; returns (lu, perm): lu like a (L unit lower below the
    diagonal, U on and above), perm (n, NSP) int32 with perm[s, k] = the row of A that became row k."""
    import torch
    n, nsp = _blocks(a)
    lu = a if overwrite else torch.empty_like(a)
    perm = torch.empty((n, nsp), dtype=torch.int32, device=a.device)
    check(_lib.lib().pj_lu_factor_dev(nsp, n, a.data_ptr(), float(gamma), lu.data_ptr(), perm.data_ptr(), _stream()))
    return lu, perm


def lu_solve(lu, perm, b, out=None):
    """x_s = A_s^-1 b_s from lu_factor's result; b: (n, NSP)."""
    import torch
    n, nsp = _blocks(lu, 'lu')
    _vectors(b, n, nsp, lu.device, 'b')
    if not (perm.is_cuda and perm.dtype == torch.int32 and perm.is_contiguous() and tuple(perm.shape) == (n, nsp)):
        raise ValueError('perm: expected lu_factor\'s int32 (n, NSP) tensor')
    x = torch.empty_like(b) if out is None else out
    _vectors(x, n, nsp, lu.device, 'out')
    check(_lib.lib().pj_lu_solve_dev(nsp, n, lu.data_ptr(), perm.data_ptr(), b.data_ptr(), x.data_ptr(), _stream()))
    return x


def newton_solve(a, b, gamma: float = 0.0, out=None, keep_factors: bool = False):
    """x_s = (I - gamma A_s)^-1 b_s (gamma = 0: A_s^-1 b_s) in one pass over the blocks; the factors stay in
    registers unless keep_factors (then returns (x, lu, perm))."""
    import torch
    n, nsp = _blocks(a)
    _vectors(b, n, nsp, a.device, 'b')
    x = torch.empty_like(b) if out is None else out
    _vectors(x, n, nsp, a.device, 'out')
    lu = torch.empty_like(a) if keep_factors else None
    perm = torch.empty((n, nsp), dtype=torch.int32, device=a.device) if keep_factors else None
    check(_lib.lib().pj_newton_solve_dev(nsp, n, a.data_ptr(), float(gamma), b.data_ptr(), x.data_ptr(),
                                         lu.data_ptr() if keep_factors else None,
                                         perm.data_ptr() if keep_factors else None, _stream()))
    return (x, lu, perm) if keep_factors else x
