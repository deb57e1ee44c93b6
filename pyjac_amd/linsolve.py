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


def _blocks(a, name='a', layout=None):
    """(n, NSP) of a batch of blocks: (n, NSP*NSP) per state (LAYOUT_AOS) or (NSP*NSP, n) state-fastest (LAYOUT_SOA)."""
    import torch
    from ._lib import LAYOUT_SOA
    if not (isinstance(a, torch.Tensor) and a.is_cuda and a.dtype == torch.float64 and a.is_contiguous() and a.dim() == 2):
        raise ValueError('%s: expected a contiguous 2-D float64 CUDA tensor' % name)
    n, ne = (a.shape[1], a.shape[0]) if layout == LAYOUT_SOA else (a.shape[0], a.shape[1])
    nsp = int(round(ne ** 0.5))
    if nsp * nsp != ne:
        raise ValueError('%s: the block dimension is not a square' % name)
    # the launch goes to torch's current stream of the CURRENT device: the tensors have to live there
    if a.device.index != torch.cuda.current_device():
        raise ValueError('%s lives on %s but the current device is cuda:%d (use torch.cuda.device(%s.device))'
                         % (name, a.device, torch.cuda.current_device(), name))
    return n, nsp


def _vectors(b, n, nsp, device, name, layout=None):
    import torch
    from ._lib import LAYOUT_SOA
    shape = (nsp, n) if layout == LAYOUT_SOA else (n, nsp)
    if not (isinstance(b, torch.Tensor) and b.is_cuda and b.dtype == torch.float64 and b.is_contiguous() and
            tuple(b.shape) == shape and b.device == device):
        raise ValueError('%s: expected a contiguous float64 CUDA tensor of shape %s on %s' % (name, shape, device))


def lu_factor(a, gamma: float = 0.0, overwrite: bool = False, layout: int = None):
    """P A = L U (or of I - gamma A) for every block; a: (n, NSP*NSP) per state, or (NSP*NSP, n) with
    layout=LAYOUT_SOA.  Returns (lu, perm): lu (n, NSP*NSP) per state (L unit lower below the diagonal, U on and
    above), perm (n, NSP) int32 with perm[s, k] = the row of A that became row k."""
    import torch
    from ._lib import LAYOUT_AOS, LAYOUT_SOA
    layout = LAYOUT_AOS if layout is None else layout
    n, nsp = _blocks(a, 'a', layout)
    if overwrite and layout == LAYOUT_SOA:
        raise ValueError('in-place factorisation needs the per-state layout')
    lu = a if overwrite else torch.empty((n, nsp * nsp), dtype=torch.float64, device=a.device)
    perm = torch.empty((n, nsp), dtype=torch.int32, device=a.device)
    check(_lib.lib().pj_lu_factor_dev(nsp, n, a.data_ptr(), layout, float(gamma), lu.data_ptr(), perm.data_ptr(), _stream()))
    return lu, perm


def lu_solve(lu, perm, b, out=None, layout: int = None):
    """x_s = A_s^-1 b_s from lu_factor's result; b: (n, NSP), or (NSP, n) with layout=LAYOUT_SOA."""
    import torch
    from ._lib import LAYOUT_AOS
    layout = LAYOUT_AOS if layout is None else layout
    n, nsp = _blocks(lu, 'lu')
    _vectors(b, n, nsp, lu.device, 'b', layout)
    if not (perm.is_cuda and perm.dtype == torch.int32 and perm.is_contiguous() and tuple(perm.shape) == (n, nsp)):
        raise ValueError('perm: expected lu_factor\'s int32 (n, NSP) tensor')
    x = torch.empty_like(b) if out is None else out
    _vectors(x, n, nsp, lu.device, 'out', layout)
    check(_lib.lib().pj_lu_solve_dev(nsp, n, lu.data_ptr(), perm.data_ptr(), b.data_ptr(), x.data_ptr(), layout, _stream()))
    return x


def newton_solve(a, b, gamma: float = 0.0, out=None, keep_factors: bool = False, layout: int = None):
    """x_s = (I - gamma A_s)^-1 b_s (gamma = 0: A_s^-1 b_s) in one pass over the blocks; the factors stay in
    registers unless keep_factors (then returns (x, lu, perm)).  layout=LAYOUT_SOA: a is (NSP*NSP, n) and b, x are
    (NSP, n) -- what Evaluator.jacobian returns by default: the Newton step without a transposed copy."""
    import torch
    from ._lib import LAYOUT_AOS
    layout = LAYOUT_AOS if layout is None else layout
    n, nsp = _blocks(a, 'a', layout)
    _vectors(b, n, nsp, a.device, 'b', layout)
    x = torch.empty_like(b) if out is None else out
    _vectors(x, n, nsp, a.device, 'out', layout)
    lu = torch.empty((n, nsp * nsp), dtype=torch.float64, device=a.device) if keep_factors else None
    perm = torch.empty((n, nsp), dtype=torch.int32, device=a.device) if keep_factors else None
    check(_lib.lib().pj_newton_solve_dev(nsp, n, a.data_ptr(), layout, float(gamma), b.data_ptr(), x.data_ptr(), layout,
                                         lu.data_ptr() if keep_factors else None,
                                         perm.data_ptr() if keep_factors else None, _stream()))
    return (x, lu, perm) if keep_factors else x
