#!/usr/bin/env python3
"""bench.py -- headline benchmark: fp64 analytical Jacobians per second.

A "step" is one pass of the hot path (the Jacobian kernel behind
pj_eval_jacobian_dev) over one batch of synthetic states that is already
resident in HBM.  One process per GPU; the batch shards trivially, so per-GPU
work is fixed as N grows ("weak" scaling) and there is no collective in the
timed region.  Rank 0 prints ONE JSON line.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workloads (BASELINE.json configs):
    gri  GRI-3.0-shaped synthetic, 53 sp / 325 rxn, 1e6 states/GPU, Dist-B (configs 3-4; default:
         the configuration BASELINE.json's metric is quoted on)
    h2   H2/O2+N2, 10 sp / 28 rxn, 1e6 states/GPU, Dist-A "PaSR-tiled"     (config 2)
    usc  USC-II-shaped synthetic, 111 sp / 784 rxn w/ PLOG, 2e5 states     (config 5)
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md chip table (spec); 6290 measured copy
# Second roof (SURVEY.md 8(d) "report both"): fp64 vector issue.  A wavefront issues one VALU instruction per
# 4 shader clocks at best (64 lanes over a 16-lane SIMD); measured with every CU busy on fp64 FMAs: 2.34 ns per
# instruction and wavefront (profiles/r02_micro_fma_latency.txt, 1.85 GHz sustained) x 256 CUs x 4 SIMDs
VALU_PEAK_WAVE_INSTR_PER_S = 256 * 4 / 2.341e-9

WORKLOADS = {
    'h2': dict(mech=os.path.join(ROOT, 'pyjac_amd', 'data', 'h2o2_n2.inp'), ref='h2o2_n2',
               n=1_000_000, dist='A', label='H2/O2+N2 10sp/28rxn, 1e6 synthetic PaSR-tiled states per GPU'),
    'gri': dict(mech=os.path.join(ROOT, 'pyjac_amd', 'data', 'gri30_shaped.inp'), ref='gri30_shaped',
                n=1_000_000, dist='B', label='GRI-Mech-3.0-shaped synthetic 53sp/325rxn, 1e6 uniform states per GPU'),
    'usc': dict(mech=os.path.join(ROOT, 'pyjac_amd', 'data', 'usc2_shaped.inp'), ref='usc2_shaped',
                n=200_000, dist='B', label='USC-Mech-II-shaped synthetic 111sp/784rxn PLOG, 2e5 uniform states per GPU'),
}


def make_states(w, nsp, n, seed):
    from pyjac_amd import synth
    if w['dist'] == 'A':
        return synth.dist_a(n, nsp, seed=seed)
    return synth.dist_b(n, nsp, seed=seed)


def usable_cpus():
    """Threads the host side may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def cpu_baseline(w, tables, target_seconds=16.0):
    """Reference's generated C (oracle/_ref, kind "reference") when that library
    travelled with the snapshot, else the table-driven port (kind "port"), timed
    on this box's host cores on a bounded sample of the same workload with the
    protocol of pyjac/performance_tester/tester.c.in:23-31 (OpenMP parallel-for over
    states): at one thread and at every usable core.  `value` is the all-cores rate."""
    import numpy as np
    from oracle.oracle import Oracle, Reference
    cores = usable_cpus()
    if Reference.available(w['ref']):
        impl, kind = Reference(w['ref']), 'reference'
    else:
        impl, kind = Oracle(tables, native=True), 'port'
    nsp = tables.nsp
    pres, y = make_states(w, nsp, 200_000 if nsp <= 16 else 20_000 if nsp <= 64 else 4_000, seed=99)
    y_aos = np.ascontiguousarray(y.T)
    n = pres.size

    def rate(threads, seconds, nn):
        impl.batch_jacob(pres[:nn], y_aos[:nn], threads)          # warm-up pass (page faults, thread pool)
        passes, dt = 0, 0.0
        t0 = time.perf_counter()
        while dt < seconds:
            impl.batch_jacob(pres[:nn], y_aos[:nn], threads)
            passes += 1
            dt = time.perf_counter() - t0
        return passes * nn / dt, passes, dt
    n1 = max(256, n // 16)
    r1, p1, t1 = rate(1, target_seconds * 0.3, n1)
    rN, pN, tN = rate(cores, target_seconds * 0.45, n)
    # the reference's CPU arm sweeps OpenMP thread counts 1, 2, 4 ... ncpu (performance_tester.py:276-283): the counts
    # between 1 and all cores, on what is left of the time budget
    mid, t = [], 2
    while t < cores:
        mid.append(t)
        t *= 2
    sweep = {1: r1, cores: rN}
    for t in mid:
        sweep[t] = rate(t, target_seconds * 0.25 / max(len(mid), 1), max(n1, min(n, n1 * t)))[0]
    return dict(value=rN, unit='Jacobians/s', cores=cores, kind=kind, one_thread=r1, cpu=cpu_model(),
                thread_sweep={str(k): sweep[k] for k in sorted(sweep)},
                sample='%d passes over %d states on %d threads (%.1f s) and %d passes over %d states on 1 thread '
                       '(%.1f s) of the same synthetic distribution, OpenMP parallel-for over states'
                       % (pN, n, cores, tN, p1, n1, t1))


def end_to_end(ev, w, n_sample, np):
    """PCIe-inclusive time of the reference's CUDA harness (tester.cu.in:109-156: H2D of the states,
    kernel, D2H of the Jacobians, pageable host memory) on a bounded sample; secondary figure."""
    from pyjac_amd.performance_tester import speedtest
    pres, y = make_states(w, ev.nsp, n_sample, seed=7)
    r = speedtest(ev, pres, y, repeats=2, quiet=True)
    nbytes = n_sample * 8 * (ev.nsp + 1 + ev.nsp * ev.nsp + 3 * ev.nsp + ev.n_fwd + max(ev.n_rev, 1) + max(ev.n_pres_mod, 1))
    return dict(states=n_sample, ms=r['end_to_end_ms'], jacobians_per_s=n_sample / r['end_to_end_ms'] * 1e3,
                host_GBps=nbytes / r['end_to_end_ms'] / 1e6,
                note='H2D + kernels + D2H through pj_run (tester.cu.in:109-156 protocol), every output array of the '
                     'reference harness copied back; caller buffers are pageable numpy arrays')


class LibraryMissing(RuntimeError):
    pass


def open_mechanism(pyjac_amd, mech, dist=None, local_rank=0, build_rblk=False, world=1, dev='cuda'):
    """Evaluator with its mechanism-specific kernels attached.  Prebuilt libraries
    (__graft_entry__.build()) are used as they are; a missing register-resident (pj_lane)
    library is compiled here (seconds).  The row-block (pj_rblk) library of a larger
    mechanism takes minutes to build: only the HEADLINE workload of a ONE-process run compiles a missing one
    (build_rblk; said on stderr), so that the metric is never quoted on the no-compile path by accident -- a
    library whose sources changed since it was built is "missing".  Without a compiler the table-driven kernels
    run and the line's `config.kernel` says so.

    Several ranks (a scaling run): nothing is compiled.  Every rank looks for the library FIRST, before any
    barrier, the ranks agree on the outcome (one MIN all-reduce of a flag), and if any of them has no up-to-date
    library ALL of them stop with one message -- no rank sits in a barrier for the ten minutes another one
    compiles (the collective watchdog's default time-out)."""
    ev = pyjac_amd.Evaluator(mech, specialize='auto')
    if dist is not None and dist.is_initialized() and world > 1:
        want = ev.spec_kind() == 'lane' or build_rblk
        agree_on_library(dist, ev.has_spec or not want, dev, '%s (expected %s)' % (os.path.basename(mech), ev.spec_path()))
        return ev
    if not ev.has_spec and (ev.spec_kind() == 'lane' or build_rblk):
        try:
            if ev.spec_kind() != 'lane':
                sys.stderr.write('bench: no up-to-date pj_rblk library for %s: compiling it (minutes)\n' % os.path.basename(mech))
            ev.specialize(build=True)
        except Exception as e:      # no compiler on this box: the table-driven kernels serve the mechanism
            sys.stderr.write('bench: cannot build the mechanism-specific kernels (%s): table-driven path\n' % e)
    return ev


def agree_on_library(dist, have: bool, dev, what: str):
    """One MIN all-reduce of "this rank has its library": raises LibraryMissing on EVERY rank if any rank has none."""
    import torch
    flag = torch.tensor([1 if have else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        raise LibraryMissing('no up-to-date mechanism-specific library for %s on at least one rank (this rank: %s): run '
                             '`python -c "import __graft_entry__ as g; g.build()"` once before a multi-GPU run -- nothing is '
                             'compiled inside one' % (what, 'present' if have else 'MISSING'))


def profile_of(kind, wl, library):
    """(json, matches) of a committed counter profile (profiles/traffic_<wl>.json, profiles/valu_<wl>.json): `matches` says
    whether it was collected on the library that is attached NOW (tools/r06_prof.sh stores the library's file name, which
    carries the digest of the kernel sources and build options, next to the counters) -- None if the profile does not say."""
    path = os.path.join(ROOT, 'profiles', '%s_%s.json' % (kind, wl))
    if not os.path.exists(path):
        return None, None
    j = json.load(open(path))
    lib = j.get('library')
    return j, (None if lib is None else bool(library) and lib == library)


def kernel_label(ev, inject=None):
    if inject:
        # PJ_BENCH_EVALUATOR: not the HIP path at all (tests/test_bench_gloo.py); the line must say so
        return 'INJECTED evaluator %s (PJ_BENCH_EVALUATOR): not a measurement of the HIP path' % inject
    return {'pj_lane': 'pj_lane (register-resident state-per-lane kernel)',
            'pj_rblk': 'pj_rblk (state-per-lane row-block kernels that rebuild their rates + falloff/PLOG pre-pass; up to 53 '
                       'species: one row kernel, four lane groups on 64 states, K_c from per-species factor columns in LDS)',
            }.get(
                ev.spec_kernel if ev.has_spec else '', 'k_tab / k_eval (table-driven: NO mechanism-specific library attached)')


def also_workloads(primary, pyjac_amd, torch, np):
    """Short kernel-only measurements of the other BASELINE.json configurations on
    this GPU (reported next to the headline, never part of `value`)."""
    out = {}
    for key, n in (('h2', 1_000_000), ('gri', 1_000_000), ('usc', 200_000)):
        if key == primary or not os.path.exists(WORKLOADS[key]['mech']):
            continue
        try:
            w = WORKLOADS[key]
            ev = open_mechanism(pyjac_amd, w['mech'])
            pres, y = make_states(w, ev.nsp, n, seed=20240901)
            soa = ev.has_spec or ev.get_launch()['tile_states'] >= 16
            L = pyjac_amd.LAYOUT_SOA if soa else pyjac_amd.LAYOUT_AOS
            d_p = torch.from_numpy(pres).cuda()
            d_y = torch.from_numpy(y if soa else np.ascontiguousarray(y.T)).cuda()
            jac = torch.empty(ev.nsp * ev.nsp * n, dtype=torch.float64, device='cuda')
            ev.time_jacobian(d_p, d_y, jac, 2, L, L)
            ms = ev.time_jacobian(d_p, d_y, jac, 5, L, L)
            gbs = n * ev.jacobian_bytes_per_state / ms / 1e6
            out[key] = dict(workload=w['label'].replace('1e6', '%g' % n).replace('2e5', '%g' % n),
                            states=n, kernel_ms=ms, jacobians_per_s=n / ms * 1e3,
                            achieved_GBps=gbs, frac=gbs / HBM_PEAK_GBPS,
                            kernel=kernel_label(ev), layout='soa' if soa else 'aos')
            del jac, d_y, d_p, ev
            torch.cuda.empty_cache()
        except Exception as ex:
            out[key] = {'error': repr(ex)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # defaults sized so that launch ramp-up (clocks, first-touch) and the final synchronisation are
    # amortised: 200 steps are 45 ms (H2), 3.5 s (GRI-shaped), 4 s (USC-shaped) of GPU time
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default=os.environ.get('PJ_WORKLOAD', 'auto'))
    ap.add_argument('--states', type=int, default=0, help='states per GPU (default: workload size)')
    ap.add_argument('--layout', default='auto', choices=['auto', 'soa', 'aos'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-also', action='store_true',
                    help='skip the short extra measurements of the other workloads (N=1 only)')
    ap.add_argument('--validate-states', default='4096',
                    help='states per rank in the multi-GPU validation all-gather (outside the timed region): a number, or '
                         '"all" for the whole batch (GRI-shaped, 1e6 states per rank: 22.5 GB per rank, gathered a chunk at a time)')
    a = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import pyjac_amd
    from pyjac_amd.dist import iter_gathered, shard_checksums

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # PJ_BENCH_EVALUATOR=module:factory (tests/test_bench_gloo.py only): THIS function -- partition, barriers,
    # MAX-reduce of the elapsed time, validation gather, cross-rank recomputation, checksums -- on CPU tensors
    # over gloo with a stand-in for the evaluator that the test supplies; never a measurement
    inject = os.environ.get('PJ_BENCH_EVALUATOR')
    backend = os.environ.get('PJ_DIST_BACKEND', 'gloo' if inject else 'nccl')
    dev = 'cpu' if inject else 'cuda'
    sync = (lambda: None) if inject else torch.cuda.synchronize
    if not inject:
        assert torch.cuda.is_available(), 'bench.py needs a GPU (no CPU fallback)'
        torch.cuda.set_device(local_rank)
    # a process group whenever the launcher set one up (torch.distributed.run exports WORLD_SIZE also for ONE process):
    # the N = 1 leg of a scaling run then goes through RCCL, the chunked validation gather and the recomputation too
    # (peer = this rank itself)
    grouped = world > 1 or ('WORLD_SIZE' in os.environ and 'MASTER_PORT' in os.environ)
    if grouped:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        import datetime
        kw = {} if inject else {'device_id': torch.device('cuda', local_rank)}
        # (an explicit time-out: the default of the collective watchdog is ten minutes, less than one validation pass over
        # 8 x 22.5 GB of Jacobians may take on a loaded node; PJ_DIST_TIMEOUT_S overrides)
        kw['timeout'] = datetime.timedelta(seconds=int(os.environ.get('PJ_DIST_TIMEOUT_S', 1800)))
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    assert world == a.gpus, 'launch with --nproc-per-node equal to --gpus'

    wl = a.workload
    if wl == 'auto':
        # BASELINE.json `metric` is quoted on the GRI-Mech 3.0 batch (configs[2], fits one GPU: 22.5 GB of
        # Jacobian); GRI-Mech 3.0 itself is not available offline: same-shape synthetic mechanism
        wl = 'gri'
    w = WORKLOADS[wl]
    try:
        if inject:
            import importlib
            mod, fn = inject.split(':')
            ev = getattr(importlib.import_module(mod), fn)(w['mech'])
            if grouped and world > 1:
                agree_on_library(dist, bool(ev.has_spec), dev, os.path.basename(w['mech']))
        else:
            ev = open_mechanism(pyjac_amd, w['mech'], dist if grouped else None, local_rank, build_rblk=True, world=world, dev=dev)
    except LibraryMissing as ex:
        # every rank is here (the ranks agreed): one message, one exit code, no rank left in a collective
        if rank == 0:
            sys.stderr.write('bench: %s\n' % ex)
        dist.destroy_process_group()
        sys.exit(3)
    n = a.states or w['n']
    # every rank owns n states (weak scaling); global batch = world * n
    pres, y = make_states(w, ev.nsp, n, seed=20240901 + rank)
    lay = a.layout
    if lay == 'auto':
        lay = 'soa' if (ev.has_spec or ev.get_launch()['tile_states'] >= 16) else 'aos'
    L = pyjac_amd.LAYOUT_SOA if lay == 'soa' else pyjac_amd.LAYOUT_AOS
    d_p = torch.from_numpy(pres).to(dev)
    d_y = torch.from_numpy(y if L == pyjac_amd.LAYOUT_SOA else np.ascontiguousarray(y.T)).to(dev)
    shape = (ev.nsp * ev.nsp, n) if L == pyjac_amd.LAYOUT_SOA else (n, ev.nsp * ev.nsp)
    jac = torch.empty(shape, dtype=torch.float64, device=dev)

    def step():
        ev.jacobian(d_p, d_y, y_layout=L, out=jac, jac_layout=L)

    # untimed: bring clocks and caches to steady state (>= 0.2 s of launches), then the W warm-up steps
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < (0.0 if inject else 0.2):
        step()
        sync()
    for _ in range(a.warmup):
        step()
    sync()
    if grouped:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    if grouped:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if grouped:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) >= elapsed
        elapsed = float(t.item())

    # kernel-only duration, HIP events on the launch stream (roofline.achieved)
    ms_kernel = ev.time_jacobian(d_p, d_y, jac, min(max(a.steps, 5), 100), L, L)

    finite = bool(torch.isfinite(jac[:, ::997] if L == pyjac_amd.LAYOUT_SOA else jac[::997]).all())
    validation = None
    if grouped:
        # The single RCCL all-gather of the path (outside the timed region): reassemble a validation batch of the
        # first nv states of every rank, a chunk at a time through one receive buffer (dist.iter_gathered).
        #  * every chunk: what arrived from rank r is finite, this rank's own piece came back bit-identical, and
        #    the sums of the received pieces add up to the checksums the ranks computed locally;
        #  * a strided sample of the states of the NEXT rank (a remote one) is evaluated again HERE, from that
        #    rank's seeded inputs, and compared with the gathered columns: the bytes that crossed
        #    xGMI are the Jacobians of those states, not merely self-consistent.
        nv = n if str(a.validate_states).lower() == 'all' else min(int(a.validate_states), n)
        soa = L == pyjac_amd.LAYOUT_SOA
        shard = (jac[:, :nv] if soa else jac[:nv].T).contiguous()
        t0 = time.perf_counter()
        cs = shard_checksums(shard)
        sums = torch.zeros(world, dtype=torch.float64, device=dev)
        ok, nbytes = True, 0
        peer = (rank + 1) % world
        p_pres, p_y = make_states(w, ev.nsp, n, seed=20240901 + peer)
        # the first min(nv, 512) states of the peer's shard, as one contiguous batch: whole workgroups, i.e. the same
        # kernel variant that produced the peer's shard (a strided handful would take the small-batch kernels)
        ns = min(nv, 512)
        sample = torch.arange(0, ns)
        pj = ev.jacobian(torch.from_numpy(np.ascontiguousarray(p_pres[:ns])).to(dev),
                         torch.from_numpy(np.ascontiguousarray(p_y[:, :ns] if soa else p_y[:, :ns].T)).to(dev),
                         y_layout=L, jac_layout=L)
        pj = pj if soa else pj.T
        sync()
        remote_err = 0.0
        nsp_ = ev.nsp
        for c0, g in iter_gathered(shard, int(os.environ.get('PJ_VALIDATE_CHUNK', 1024))):
            cols = g.shape[2]
            ok &= bool(torch.equal(g[rank], shard[:, c0:c0 + cols])) and bool(torch.isfinite(g).all())
            sums += g.sum(dim=(1, 2))
            nbytes += int(g.numel() * 8)
            inside = (sample >= c0) & (sample < c0 + cols)
            if bool(inside.any()):
                got = g[peer][:, (sample[inside] - c0).to(g.device)]
                ref = pj[:, inside.to(pj.device)]
                # entry-wise |d| <= 1e-6 |J| + 1e-12 max(row scale, column scale): the metric of tests/conftest.py
                # (identical kernels give 0; another kernel variant differs at rounding level on entries that are
                # 1e-13 of their row scale, where a plain relative error means nothing)
                ab = ref.abs().reshape(nsp_, nsp_, -1)                     # [col][row][state]
                scale = torch.maximum(ab.amax(dim=0, keepdim=True), ab.amax(dim=1, keepdim=True)).expand_as(ab).reshape(ref.shape)
                tol = 1e-6 * ref.abs() + 1e-12 * scale + 1e-300
                remote_err = max(remote_err, float(((got - ref).abs() / tol).max()))
        for r in range(world):
            ok &= bool(torch.allclose(cs[r, 0], sums[r], rtol=1e-9))
        ok &= remote_err <= 1.0
        validation = dict(states_per_rank=nv, gathered_bytes=nbytes, ok=bool(ok), remote_rank_checked=peer,
                          remote_states_recomputed=int(sample.numel()), remote_max_err_over_tolerance=remote_err,
                          seconds=round(time.perf_counter() - t0, 4))

    if rank == 0:
        total = world * n * a.steps
        value = total / elapsed
        bj = ev.jacobian_bytes_per_state
        achieved = n * bj / (ms_kernel * 1e-3) / 1e9
        library = os.path.basename(getattr(ev, 'attached_spec', None) or '') or None
        traffic = traffic_source = None
        tj, traffic_matches = profile_of('traffic', wl, library)   # rocprofv3 PMC passes of this round
        if tj:
            # PMC-measured HBM bytes of this kernel (profiles/README.md); a streaming map, so
            # a launch over n states moves n / states_per_launch times the profiled bytes
            traffic = tj['hbm_bytes_per_launch'] * n / tj.get('states_per_launch', n)
            # NOT measured in this run (PMC counters need rocprofv3 around the process): say where it comes from
            traffic_source = ('committed profile profiles/traffic_%s.json (rocprofv3 --pmc passes over tools/one_step.py, %d states '
                              'per launch, library %s), not measured in this run' % (wl, tj.get('states_per_launch', 0), tj.get('library')))
        # the instruction roof of the same step: VALU instructions per state and the share of wave-cycles that
        # issue, from the committed SQ-counter summary of this workload (profiles/valu_<wl>.json, tools/valu_roof.py)
        valu = None
        vj, valu_matches = profile_of('valu', wl, library)
        if vj:
            wave_instr_per_s = vj['valu_instr_per_state'] * n / 64.0 / (ms_kernel * 1e-3)
            valu = {'instr_per_state': vj['valu_instr_per_state'], 'issue_frac': vj['issue_frac'],
                    'wait_frac': vj.get('wait_frac'), 'achieved_wave_instr_per_s': wave_instr_per_s,
                    'fp64_peak_wave_instr_per_s': VALU_PEAK_WAVE_INSTR_PER_S,
                    'frac': wave_instr_per_s / VALU_PEAK_WAVE_INSTR_PER_S, 'source': vj.get('source'),
                    'profile_matches_library': valu_matches,
                    'source_kind': 'committed profile profiles/valu_%s.json (SQ counters of an earlier rocprofv3 run), '
                                   'not measured in this run; only achieved_wave_instr_per_s uses this run\'s kernel_ms' % wl}
        hbm_frac = achieved / HBM_PEAK_GBPS
        line = {
            'metric': 'fp64 analytical Jacobians/s', 'value': value, 'unit': 'Jacobians/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': elapsed / a.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': w['label'], 'key': wl, 'states_per_gpu': n, 'nsp': ev.nsp,
                       'n_rxn': ev.n_fwd, 'layout': lay, 'launch': ev.get_launch(),
                       'parallelism': 'states sharded over %d GPU(s), no data-path collective' % world,
                       'finite': finite},
            # `frac` is the HBM fraction the metric asks for; `bound` names the roof the kernel is closer to
            'roofline': {'bound': 'valu' if (valu and valu['frac'] > hbm_frac) else 'hbm', 'achieved': achieved,
                         'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': hbm_frac, 'traffic': traffic,
                         'traffic_source': traffic_source,
                         # do the committed counter profiles belong to the library that ran?  (its file name carries the digest
                         # of kernel sources + build options; None: a profile that does not name its library)
                         'library': library, 'profile_matches_library': traffic_matches,
                         'bytes_per_state': bj, 'kernel_ms': ms_kernel, 'valu': valu},
        }
        if validation:
            line['validation_allgather'] = validation
        line['config']['kernel'] = kernel_label(ev, inject)
        line['evaluator'] = ('injected:' + inject) if inject else 'native (pyjac_amd HIP path through the C ABI)'
        if world == 1 and not a.no_also:
            line['also'] = also_workloads(wl, pyjac_amd, torch, np)
            if ev.spec_kernel in ('pj_lane', 'pj_rblk') and L == pyjac_amd.LAYOUT_SOA:
                # the Jacobian consumed in registers (w = J v), nothing but T, p, Y, v read
                # and w written -- reported next to the headline, never part of `value`
                try:
                    d_v = torch.randn_like(d_y)
                    d_w = torch.empty_like(d_y)
                    for _ in range(3):
                        ev.jacobian_vec(d_p, d_y, d_v, out=d_w)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        ev.jacobian_vec(d_p, d_y, d_v, out=d_w)
                    e1.record()
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / 20
                    bjv = 8 * (2 * ev.nsp + 1) + 8 * ev.nsp
                    line['also']['fused_jacobian_vector_product'] = dict(
                        states=n, kernel_ms=ms, products_per_s=n / ms * 1e3, bytes_per_state=bjv,
                        achieved_GBps=n * bjv / ms / 1e6, finite=bool(torch.isfinite(d_w[:, ::997]).all()),
                        kernel='k_lane with w = J v fused in' if ev.spec_kernel == 'pj_lane' else
                               'k_jvd (every reaction once: its derivative row times v, scattered like its rate)',
                        note='no Jacobian in memory: %d instead of %d bytes per state' % (bjv, bj))
                except Exception as ex:
                    line['also']['fused_jacobian_vector_product'] = {'error': repr(ex)}
                # the path a user without a compiler gets: the same batch through the table-driven kernels (no
                # mechanism-specific library: k_tab for this SoA batch) -- reported, never part of `value`
                try:
                    ev0 = pyjac_amd.Evaluator(w['mech'], specialize='off')
                    ev0.time_jacobian(d_p, d_y, jac, 1, L, L)
                    ms0 = ev0.time_jacobian(d_p, d_y, jac, 3, L, L)
                    line['also']['no_compile_path'] = dict(
                        kernel='k_tab + k_tab_fin (table-driven, state per lane)', states=n, kernel_ms=ms0,
                        jacobians_per_s=n / ms0 * 1e3, frac=n * bj / ms0 / 1e6 / HBM_PEAK_GBPS,
                        finite=bool(torch.isfinite(jac[:, ::997]).all()))
                    ev0.close()
                    step()          # leave the headline kernel's output in `jac`
                except Exception as ex:
                    line['also']['no_compile_path'] = {'error': repr(ex)}
                # the consumer of the Jacobians: one Newton update (I - gamma J_s) dx_s = r_s per state,
                # factor + solve fused (csrc/pj_lu.h), straight from the batch-layout Jacobians of this step
                try:
                    from pyjac_amd import linsolve
                    lay = pyjac_amd.LAYOUT_SOA if L == pyjac_amd.LAYOUT_SOA else pyjac_amd.LAYOUT_AOS
                    rhs = torch.ones((ev.nsp, n) if lay == pyjac_amd.LAYOUT_SOA else (n, ev.nsp), dtype=torch.float64, device='cuda')
                    dx = torch.empty_like(rhs)
                    linsolve.newton_solve(jac, rhs, gamma=1e-7, out=dx, layout=lay)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(3):
                        linsolve.newton_solve(jac, rhs, gamma=1e-7, out=dx, layout=lay)
                    e1.record()
                    torch.cuda.synchronize()
                    ms_lu = e0.elapsed_time(e1) / 3
                    line['also']['newton_step'] = dict(
                        states=n, solve_ms=ms_lu, jacobian_ms=ms_kernel, total_ms=ms_kernel + ms_lu,
                        steps_per_s=n / (ms_kernel + ms_lu) * 1e3, finite=bool(torch.isfinite(dx[:, ::997] if lay == pyjac_amd.LAYOUT_SOA else dx[::997]).all()),
                        note='batched LU with partial pivoting + solve of (I - 1e-7 J) dx = 1, one pass over the %d x %d '
                             'blocks (pj_newton_solve_dev); jacobian_ms = the headline kernel' % (ev.nsp, ev.nsp))
                    del rhs, dx
                except Exception as ex:
                    line['also']['newton_step'] = {'error': repr(ex)}
                # the reference's own comparison arm (performance_tester/fd_jacob.c: NSP + 1 dydt evaluations per
                # state) on a bounded sample of the same batch -- the "analytical vs finite difference" ratio of the pyJac
                # paper, reported, never part of `value`
                try:
                    # (bounded by bytes: NSP^2 doubles per state next to the live Jacobian -- at most 2 GB; SoA inputs)
                    nf = max(64, min(n, 131072, (2 << 30) // (8 * ev.nsp * ev.nsp)) // 64 * 64)
                    fd_p = d_p[:nf].contiguous()
                    fd_y = (d_y[:, :nf] if L == pyjac_amd.LAYOUT_SOA else d_y[:nf].T).contiguous()
                    fd_out = torch.empty((ev.nsp * ev.nsp, nf), dtype=torch.float64, device=d_p.device)
                    ev.fd_jacobian(fd_p, fd_y, out=fd_out)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(3):
                        ev.fd_jacobian(fd_p, fd_y, out=fd_out)
                    e1.record()
                    torch.cuda.synchronize()
                    ms_fd = e0.elapsed_time(e1) / 3
                    line['also']['finite_difference_arm'] = dict(
                        states=nf, kernel_ms=ms_fd, jacobians_per_s=nf / ms_fd * 1e3,
                        analytical_over_fd=(n / ms_kernel) / (nf / ms_fd),
                        note='first-order differences of the GPU dydt with fd_jacob.c\'s increment: %d rate passes per state' % (ev.nsp + 1))
                    del fd_out, fd_p, fd_y
                except Exception as ex:
                    line['also']['finite_difference_arm'] = {'error': repr(ex)}
                # configs[1] "spec_rates + Jacobian": the rate pass (pyjacob.cu k_dydt) on the same batch,
                # with every intermediate array written (conc, fwd, rev, pres_mod, spec_rates, dy) and
                # with dy only
                try:
                    from pyjac_amd import _lib
                    rows = dict(conc=ev.nsp, fwd=ev.n_fwd, rev=max(ev.n_rev, 1), pres_mod=max(ev.n_pres_mod, 1),
                                spec_rates=ev.nsp, dy=ev.nsp)
                    bufs = {k: torch.empty((r, n), dtype=torch.float64, device='cuda') for k, r in rows.items()}
                    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

                    def rates(which):
                        p = lambda k: bufs[k].data_ptr() if k in which else None
                        _lib.check(_lib.lib().pj_eval_rates_dev(ev._h, n, d_p.data_ptr(), d_y.data_ptr(), L, p('conc'),
                                                                p('fwd'), p('rev'), p('pres_mod'), p('spec_rates'),
                                                                p('dy'), stream))
                    res = {}
                    for label, which in (('all_arrays', tuple(rows)), ('dydt_only', ('dy',))):
                        for _ in range(3):
                            rates(which)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(20):
                            rates(which)
                        e1.record()
                        torch.cuda.synchronize()
                        ms = e0.elapsed_time(e1) / 20
                        by = 8 * (ev.nsp + 1) + 8 * sum(rows[k] for k in which)
                        res[label] = dict(kernel_ms=ms, states_per_s=n / ms * 1e3, bytes_per_state=by,
                                          achieved_GBps=n * by / ms / 1e6, frac=n * by / ms / 1e6 / HBM_PEAK_GBPS)
                    line['also']['rate_pass'] = res
                except Exception as ex:
                    line['also']['rate_pass'] = {'error': repr(ex)}
        if world == 1 and not a.no_cpu_baseline:
            try:
                line['end_to_end'] = end_to_end(ev, w, 65536 if ev.nsp <= 64 else 16384, np)
            except Exception as ex:
                line['end_to_end'] = {'error': repr(ex)}
            try:
                line['cpu_baseline'] = cpu_baseline(w, ev.tables)
            except Exception as ex:   # the baseline is reported, never required
                line['cpu_baseline'] = {'error': repr(ex)}
        print(json.dumps(line), flush=True)
    if grouped:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
