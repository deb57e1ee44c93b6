"""Mechanism handle over the C ABI (host arrays and torch device tensors)."""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import _lib
from ._lib import LAYOUT_AOS, LAYOUT_SOA, check, dptr
from .mechanism import read_mech
from .tables import MechTables, build_tables


class Evaluator:
    """One mechanism on the current HIP device.

    Replaces "one compiled pyjacob module per mechanism"
    (pyjac/pywrap/pywrap_gen.py:66-128): the mechanism is loaded as tables.
    """

    def __init__(self, mech, therm: str = None, last_spec: str = None, specialize: str = 'auto'):
        """specialize: 'auto' attaches the prebuilt mechanism-specific kernels if
        pyjac_amd/spec/ holds them; 'build' also compiles them when missing (hipcc:
        seconds for H2-size mechanisms, about a minute on 8 cores for GRI-size);
        'off' never (table-driven kernel)."""
        if isinstance(mech, MechTables):
            self.tables = mech
            self.mechanism = None
        elif isinstance(mech, str) and mech.endswith('.pjtab'):
            self.tables = MechTables.load(mech)
            self.mechanism = None
        else:
            self.mechanism = read_mech(mech, therm, last_spec)
            self.tables = build_tables(self.mechanism)
        L = _lib.lib()
        I = np.ascontiguousarray(self.tables.I, dtype=np.int32)
        D = np.ascontiguousarray(self.tables.D, dtype=np.float64)
        h = ctypes.c_void_p()
        check(L.pj_mech_create(I.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), I.size,
                               dptr(D), D.size, ctypes.byref(h)))
        self._h = h
        self.nsp = L.pj_mech_nsp(h)
        self.n_fwd = L.pj_mech_fwd_rates(h)
        self.n_rev = L.pj_mech_rev_rates(h)
        self.n_pres_mod = L.pj_mech_pres_mod_rates(h)
        self.attached_spec = None
        # bumped by every setter that can change results or the kernel that produces them: consumers that cache
        # results per state (pyjacob.py) key on it
        self.settings_generation = 0
        if specialize != 'off':
            self.specialize(build=(specialize == 'build'))

    # ---- mechanism-specific state-per-lane kernels ----
    # csrc/pj_lane.hip: the whole Jacobian in one lane's registers (small mechanisms);
    # csrc/pj_rblk.hip: row-block kernels that rebuild the rates they need + one-pass rate-output kernels (the rest)
    SPEC_MAX_NSP, SPEC_MAX_RXN = 16, 64

    def spec_kind(self) -> str:
        """Which specialised kernel family specialize(build=True) compiles for this mechanism."""
        from .tables import DA_PROD_NU, DA_REAC_NU, F_CHEB, F_SRI, IA_FLAGS, IA_PROD_PTR, IA_REAC_PTR
        I, D = self.tables.I, self.tables.D
        flags = I[I[16 + IA_FLAGS]:I[16 + IA_FLAGS] + self.n_fwd]
        # (pj_lane.hip carries no SRI / Chebyshev code and no general stoichiometry -- fractional coefficients,
        # more than three molecules on a side: those mechanisms take the row-block family at any size)
        small = self.nsp <= self.SPEC_MAX_NSP and self.n_fwd <= self.SPEC_MAX_RXN
        general = False
        for ptr, nu in ((IA_REAC_PTR, DA_REAC_NU), (IA_PROD_PTR, DA_PROD_NU)):
            pp = I[I[16 + ptr]:I[16 + ptr] + self.n_fwd + 1]
            nn = D[I[48 + nu]:I[48 + nu] + int(pp[-1])]
            general |= any(float(x) != int(x) for x in nn) or any(nn[pp[i]:pp[i + 1]].sum() > 3 for i in range(self.n_fwd))
        return 'lane' if small and not general and not any(int(f) & (F_SRI | F_CHEB) for f in flags) else 'rblk'

    def spec_path(self, kind: str = None, **opts) -> str:
        """File name of a specialised library: mechanism hash + a digest of everything else that shapes the
        binary (specbuild.library_path).  None: no sources installed and no prebuilt library of this kind."""
        from . import specbuild
        return specbuild.library_path(kind or self.spec_kind(), _lib.lib().pj_mech_spec_hash(self._h), opts)

    def specialize(self, build: bool = False, kind: str = None, **opts) -> bool:
        """Attach (and with build=True compile if missing) the mechanism-specific kernels.
        kind: 'lane' | 'rblk' | None (whatever is there, else the default for the size)."""
        from . import specbuild
        L = _lib.lib()
        if kind not in (None, 'lane', 'rblk'):
            raise ValueError("kind: 'lane' or 'rblk'")
        kinds = [kind] if kind else [self.spec_kind()] + [k for k in ('lane', 'rblk') if k != self.spec_kind()]
        paths = [self.spec_path(k, **opts) for k in kinds]
        so = next((q for q in paths if q and os.path.exists(q)), None)
        if so is None:
            if not build or paths[0] is None:
                return False
            so = paths[0]
            if kinds[0] == 'lane':
                specbuild.build_lane(L, self._h, so)
            else:
                from .kcfactors import kc_factor_rows
                specbuild.build_rblk(L, self._h, self.nsp, so, kcf_rows=kc_factor_rows(self.tables), nkc=int(self.tables.I[10]),
                                     nrxn=self.n_fwd, **opts)
        check(L.pj_mech_attach_spec(self._h, so.encode()))
        self.settings_generation += 1
        self.attached_spec = so
        return True

    @property
    def spec_kernel(self) -> str:
        """'pj_lane' / 'pj_rblk' for the attached specialisation, '' if none."""
        so = os.path.basename(self.attached_spec or '')
        return 'pj_lane' if so.startswith('libpj_spec_') else 'pj_rblk' if so.startswith('libpj_rblk_') else ''

    @property
    def has_spec(self) -> bool:
        return bool(_lib.lib().pj_mech_has_spec(self._h))

    def use_spec(self, on):
        """False/0: table-driven kernel; True/1: attached kernels for SoA Jacobians (default);
        2: attached kernels for every layout."""
        check(_lib.lib().pj_mech_use_spec(self._h, int(on)))
        self.settings_generation += 1

    def set_spec_launch(self, streams: int = -1, chunk_states: int = -1, split_tail: int = -1, aos_direct: int = -1):
        """Launch settings of the attached row-block library, per evaluator (include/pyjac_amd.h:
        pj_mech_set_spec_launch); -1 leaves a setting unchanged."""
        check(_lib.lib().pj_mech_set_spec_launch(self._h, int(streams), int(chunk_states), int(split_tail), int(aos_direct)))
        self.settings_generation += 1

    def close(self):
        if getattr(self, '_h', None):
            _lib.lib().pj_mech_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- tuning / options ----
    def set_launch(self, tile_states: int = 0, threads: int = 0):
        check(_lib.lib().pj_mech_set_launch(self._h, tile_states, threads))
        self.settings_generation += 1

    def get_launch(self):
        a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(_lib.lib().pj_mech_get_launch(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return dict(tile_states=a.value, threads=b.value, lds_bytes=c.value)

    def set_generic_kernel(self, name: str):
        """'auto' (default: k_tab for SoA Jacobians, k_eval for AoS ones), 'k_tab' (table-driven state-per-lane
        row blocks) or 'k_eval' (cooperative kernel) for Jacobians evaluated without an attached library."""
        check(_lib.lib().pj_mech_set_generic_kernel(self._h, {'k_eval': 0, 'auto': 1, 'k_tab': 2}[name]))
        self.settings_generation += 1

    def set_check_inputs(self, on: bool):
        """Verify T > 0, p > 0 and finite inputs before every device evaluation (one extra pass + a sync)."""
        check(_lib.lib().pj_mech_set_check_inputs(self._h, int(on)))

    def set_sum_last_species(self, on: bool):
        check(_lib.lib().pj_mech_set_sum_last_species(self._h, int(on)))
        self.settings_generation += 1

    # ---- bytes per unit of work ----
    @property
    def jacobian_bytes_per_state(self) -> int:
        return 8 * (self.nsp + 1) + 8 * self.nsp * self.nsp

    @property
    def rates_bytes_per_state(self) -> int:
        return 8 * (self.nsp + 1) + 8 * self.nsp

    # ---- device tensors (torch is plumbing: memory + streams) ----
    @staticmethod
    def _stream():
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _chk(name, t, numel, device=None):
        """The C ABI takes raw pointers: a wrong size, dtype, device or a strided view would be an
        out-of-bounds access on the device, so refuse it here."""
        import torch
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
            raise ValueError('%s: expected a contiguous float64 CUDA tensor' % name)
        if t.numel() != numel:
            raise ValueError('%s: expected %d elements, got %d' % (name, numel, t.numel()))
        if device is not None and t.device != device:
            raise ValueError('%s: on %s, expected %s' % (name, t.device, device))

    def jacobian(self, pres, y, y_layout=LAYOUT_SOA, out=None, jac_layout=LAYOUT_SOA):
        """pres: (n,) cuda f64; y: SoA (NSP, n) or AoS (n, NSP) cuda f64.
        Returns jac as SoA (NSP*NSP, n) or AoS (n, NSP*NSP)."""
        import torch
        n = pres.numel()
        self._chk('pres', pres, n)
        self._chk('y', y, n * self.nsp, pres.device)
        if out is None:
            shape = (self.nsp * self.nsp, n) if jac_layout == LAYOUT_SOA else (n, self.nsp * self.nsp)
            out = torch.empty(shape, dtype=torch.float64, device=pres.device)
        self._chk('out', out, n * self.nsp * self.nsp, pres.device)
        check(_lib.lib().pj_eval_jacobian_dev(self._h, n, pres.data_ptr(), y.data_ptr(), y_layout,
                                              out.data_ptr(), jac_layout, self._stream()))
        return out

    def newton_solve(self, pres, y, rhs, gamma: float, layout=LAYOUT_SOA, out=None):
        """One Newton update per state: dx_s = (I - gamma J(Phi_s))^-1 rhs_s -- the Jacobian kernel followed by the
        batched LU / solve of csrc/pj_lu.h on the blocks where the kernel left them (no transposed copy; the
        per-state integrator loop of docs/examples.rst:106-170 as two launches).  y, rhs, dx: SoA (NSP, n) or AoS
        (n, NSP) according to `layout`."""
        from . import linsolve
        jac = self.jacobian(pres, y, y_layout=layout, jac_layout=layout)
        return linsolve.newton_solve(jac, rhs, gamma=gamma, out=out, layout=layout)

    def jacobian_vec(self, pres, y, v, layout=LAYOUT_SOA, out=None):
        """w_s = J(Phi_s) v_s per state (the consumer of pyJac's sparse_multiplier fused into the
        Jacobian kernel when a pj_lane library is attached).  y, v, w: SoA (NSP, n) or AoS (n, NSP)."""
        import torch
        n = pres.numel()
        self._chk('pres', pres, n)
        self._chk('y', y, n * self.nsp, pres.device)
        self._chk('v', v, n * self.nsp, pres.device)
        if out is None:
            out = torch.empty_like(v)
        self._chk('out', out, n * self.nsp, pres.device)
        check(_lib.lib().pj_eval_jacobian_vec_dev(self._h, n, pres.data_ptr(), y.data_ptr(), layout,
                                                  v.data_ptr(), out.data_ptr(), layout, self._stream()))
        return out

    def rates(self, pres, y, y_layout=LAYOUT_SOA, want=('conc', 'fwd', 'rev', 'pres_mod',
                                                          'spec_rates', 'dydt')):
        """All SoA outputs of pyjacob.cu's k_dydt pass as a dict of (rows, n) tensors."""
        import torch
        n = pres.numel()
        self._chk('pres', pres, n)
        self._chk('y', y, n * self.nsp, pres.device)
        rows = dict(conc=self.nsp, fwd=self.n_fwd, rev=max(self.n_rev, 1),
                    pres_mod=max(self.n_pres_mod, 1), spec_rates=self.nsp, dydt=self.nsp)
        outs = {k: torch.zeros((rows[k], n), dtype=torch.float64, device=pres.device) for k in want}
        p = lambda k: outs[k].data_ptr() if k in outs else None
        check(_lib.lib().pj_eval_rates_dev(self._h, n, pres.data_ptr(), y.data_ptr(), y_layout,
                                           p('conc'), p('fwd'), p('rev'), p('pres_mod'),
                                           p('spec_rates'), p('dydt'), self._stream()))
        return outs

    def fd_jacobian(self, pres, y, out=None, jac_layout=LAYOUT_SOA):
        """Finite-difference Jacobian of dydt (the reference's comparison arm,
        performance_tester/fd_jacob.c); y SoA (NSP, n)."""
        import torch
        n = pres.numel()
        self._chk('pres', pres, n)
        self._chk('y', y, n * self.nsp, pres.device)
        if out is None:
            shape = (self.nsp * self.nsp, n) if jac_layout == LAYOUT_SOA else (n, self.nsp * self.nsp)
            out = torch.empty(shape, dtype=torch.float64, device=pres.device)
        self._chk('out', out, n * self.nsp * self.nsp, pres.device)
        check(_lib.lib().pj_eval_fd_jacobian_dev(self._h, n, pres.data_ptr(), y.data_ptr(), out.data_ptr(),
                                                 jac_layout, self._stream()))
        return out

    def time_jacobian(self, pres, y, out, iters: int, y_layout=LAYOUT_SOA, jac_layout=LAYOUT_SOA):
        """Average kernel time (ms) over `iters` launches, HIP events on the
        launch stream (pj_time_jacobian_dev)."""
        self._chk('pres', pres, pres.numel())
        self._chk('y', y, pres.numel() * self.nsp, pres.device)
        self._chk('out', out, pres.numel() * self.nsp * self.nsp, pres.device)
        ms = ctypes.c_double()
        check(_lib.lib().pj_time_jacobian_dev(self._h, pres.numel(), pres.data_ptr(), y.data_ptr(),
                                              y_layout, out.data_ptr(), jac_layout, self._stream(),
                                              iters, ctypes.byref(ms)))
        return ms.value

    # ---- host batch driver: pyjacob.cu init / run / cleanup ----
    def init(self, num: int) -> int:
        return check(_lib.lib().pj_init(self._h, int(num)))

    def run(self, num, padded, pres, y, conc, fwd, rev, pres_mod, spec_rates, dy, jac):
        check(_lib.lib().pj_run(self._h, int(num), int(padded), dptr(pres), dptr(y), dptr(conc),
                                dptr(fwd), dptr(rev), dptr(pres_mod), dptr(spec_rates), dptr(dy),
                                dptr(jac)))

    def cleanup(self):
        check(_lib.lib().pj_cleanup(self._h))
