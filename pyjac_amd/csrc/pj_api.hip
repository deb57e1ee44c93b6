// pj_api.hip -- HIP kernels + C ABI (include/pyjac_amd.h) for gfx950.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#define PJ_DEV __device__ __forceinline__
#include "pj_kernel.h"
#include "pj_tab.h"
#define PJ_LU_DECL_ONLY 1       // (the kernels: csrc/pj_lu.hip, translation units of their own)
#include "pj_lu.h"
#include "../../include/pyjac_amd.h"

using namespace pj;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

#define HIPCHK(call)                                                                   \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess)                                                          \
            return fail(e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice ? PJ_ENODEV \
                                                                              : PJ_EHIP, \
                        std::string(#call) + ": " + hipGetErrorString(e_));            \
    } while (0)

constexpr int MODE_JAC = 1, MODE_CONC_IN = 2;
// timing ablations (wrong results; tools/gpu_probe.py --ablate): skip a phase
constexpr int ABL_P2 = 16, ABL_P3 = 32, ABL_P4 = 64, ABL_P0 = 128;

// conc-input variant of phase 0: the caller supplies concentrations
// (eval_rxn_rates / get_rxn_pres_mod prototypes, pyjacob_wrapper.pyx:10-13)
template <int TS>
PJ_DEV void phase0c(const DevMech& M, const Batch& B, const double* cin, const double* Tin, double* V,
                    int tid, int NT, long tile, Lane& L)
{
    const int s = tid % TS, u = tid / TS, NU = NT / TS;
    long gs = tile * TS + s;
    L.valid = gs < B.n;
    if (!L.valid) gs = B.n - 1;
    L.gs = gs;
    const double T = Tin[gs], p = B.pres[gs];
    L.T = T; L.p = p; L.logT = log(T); L.invT = 1.0 / T; L.logp = log(p);
    L.Wbar = 1.0; L.rho = 1.0; L.invrho = 1.0; L.m = p / (RU_ * T); L.yN = 0.0;
    L.cpavg = 1.0; L.dcp = 0.0;
    for (int k = u; k < M.nsp; k += NU) {
        V[(M.v.C + k) * TS + s] = cin[k * B.o_ld + gs];
        V[(M.v.HW + k) * TS + s] = 0.0;
        V[(M.v.CP + k) * TS + s] = 0.0;
        V[(M.v.YC + k) * TS + s] = 0.0;
        V[(M.v.YD + k) * TS + s] = 0.0;
    }
    if (u == 0) V[M.v.ONE * TS + s] = 1.0;
}

#ifndef PJ_KEVAL_WAVES
#define PJ_KEVAL_WAVES 3      // waves per SIMD the register allocation is bounded for (<= 168 VGPRs)
#endif
template <int TS>
__global__ void __launch_bounds__(256, PJ_KEVAL_WAVES)
k_eval(DevMech M, Batch B, int mode, const double* cin, const double* Tin, double* aux)
{
    extern __shared__ __attribute__((aligned(16))) double V[];
    const int tid = threadIdx.x, NT = blockDim.x;
    const long ntiles = (B.n + TS - 1) / TS;
#ifdef PJ_TIMING
    long long tacc[10] = {0}, tprev = clock64();
#define PJ_TICK(i) { __syncthreads(); const long long tn = clock64(); tacc[i] += tn - tprev; tprev = tn; }
#else
#define PJ_TICK(i)
#endif
    stage_consts<TS>(M, V, tid, NT);
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        Lane L;
        if (mode & MODE_CONC_IN) {
            phase0c<TS>(M, B, cin, Tin, V, tid, NT, tile, L);
        } else {
            phase0a<TS>(M, B, V, tid, NT, tile, L);
            PJ_TICK(0)
            phase_zero_tile<TS>(M, V, tid, NT);
            __syncthreads();
            PJ_TICK(1)
            phase0b<TS>(M, B, V, tid, NT, L);
            __syncthreads();
            PJ_TICK(2)
            phase0c_scale<TS>(M, B, V, tid, NT, L);
            if (aux && tid / TS == 0 && L.valid) {
                // y_N, mw_avg, rho of eval_conc (rate_subs.py:1595-1597)
                aux[0 * B.o_ld + L.gs] = L.yN;
                aux[1 * B.o_ld + L.gs] = L.Wbar;
                aux[2 * B.o_ld + L.gs] = L.rho;
            }
        }
        __syncthreads();
        PJ_TICK(3)
        if (!(mode & ABL_P2)) phase2<TS>(M, B, V, tid, NT, L);
        __syncthreads();
        PJ_TICK(4)
        if (!(mode & MODE_CONC_IN)) {
            if (!(mode & ABL_P3)) phase_scatter<TS>(M, V, tid, NT, !(mode & MODE_JAC));
            __syncthreads();
            PJ_TICK(5)
            phase_fin1<TS>(M, V, tid, NT);
            __syncthreads();
            PJ_TICK(6)
            phase_fin2a<TS>(M, B, V, tid, NT, L);
            __syncthreads();
            phase_fin2b<TS>(M, B, V, tid, NT, L);
            __syncthreads();
            PJ_TICK(7)
            if ((mode & MODE_JAC) && !(mode & ABL_P4)) {
                phase_out_energy<TS>(M, B, V, tid, NT, L);
                PJ_TICK(8)
                phase_out_block<TS>(M, B, V, tid, NT, L);
            }
        }
        __syncthreads();
        PJ_TICK(9)
    }
#ifdef PJ_TIMING
    if (Tin && !(mode & MODE_CONC_IN) && tid == 0 && blockIdx.x < 64)
        for (int i = 0; i < 10; ++i) const_cast<double*>(Tin)[blockIdx.x * 10 + i] = (double)tacc[i];
#endif
}

// eval_spec_rates from caller-supplied rates (pyjacob_wrapper.pyx:11); one thread per state
// ---- k_tab / k_tab_fin: the table-driven state-per-lane Jacobian kernels (pj_tab.h, pj_tabprog.h) ----
// 256 threads: G lane groups on the same L = 256 / G states; LDS: concentration columns + the groups' accumulators
__global__ void __launch_bounds__(256) k_tab(DevMech M, TabDev P, Batch B)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    TabLane Ln;
    tab_stage(M, P, B, lds, (int)threadIdx.x, (long)blockIdx.x, Ln);
    __syncthreads();
    tab_blocks(M, P, B, lds, (int)threadIdx.x, Ln);
}

__global__ void __launch_bounds__(256) k_tab_fin(DevMech M, TabDev P, Batch B)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    tab_fin_stage(M, P, B, lds, (int)threadIdx.x, (long)blockIdx.x);
    __syncthreads();
    tab_fin_cols(M, P, B, lds, (int)threadIdx.x, (long)blockIdx.x);
}

__global__ void k_spec_rates(DevMech M, long n, const double* fwd, const double* rev,
                             const double* pm, double* sr)
{
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    for (int k = 0; k < M.nsp; ++k) {
        double om = 0.0;
        for (int q = M.sp_ptr[k]; q < M.sp_ptr[k + 1]; ++q) {
            const int32_t* ri = M.ri + M.sp_rxn[q] * RIW;
            double R = fwd[ri[RI_ORIG] * n + s];
            if (ri[RI_REV_IDX] >= 0) R -= rev[ri[RI_REV_IDX] * n + s];
            if (ri[RI_PRES_IDX] >= 0) R *= pm[ri[RI_PRES_IDX] * n + s];
            om += M.sp_nu[q] * R;
        }
        sr[k * n + s] = om;
    }
}

// ---- optional precondition check (SURVEY.md 8(b) "Preconditions": T > 0 -- log T is taken --, finite inputs,
// p > 0): first offending state index, or LONG_MAX ----
__global__ void k_check_inputs(long n, int nsp, const double* pres, const double* y, long y_si, long y_ss,
                               unsigned long long* first_bad)
{
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const double T = y[s * y_ss], p = pres[s];
    bool bad = !(T > 0.0) || !(p > 0.0) || !(T < 1.0e300) || !(p < 1.0e300);
    for (int i = 1; i < nsp; ++i) {
        const double v = y[i * y_si + s * y_ss];
        bad |= !(v == v) || !(v < 1.0e300) || !(v > -1.0e300);
    }
    if (bad) atomicMin(first_bad, (unsigned long long)s);
}

// ---- finite-difference Jacobian helpers (fd_jacob.c:56-111) ----
__global__ void k_fd_setup(long n, int nsp, const double* y, const double* dy0, double* r)
{
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const double ATOL = 1e-15, RTOL = 1e-8, EPS = 2.2204460492503131e-16;
    double sum = 0.0;
    for (int i = 0; i < nsp; ++i) {
        const double ewt = ATOL + (RTOL * fabs(y[i * n + s]));
        const double e = ewt * dy0[i * n + s];
        sum += e * e;
    }
    const double fac = sqrt(sum / ((double)nsp));
    const double r0 = 1000.0 * RTOL * EPS * ((double)nsp) * fac;
    const double srur = sqrt(EPS);
    for (int j = 0; j < nsp; ++j) {
        const double yj = y[j * n + s];
        const double ewt = ATOL + (RTOL * fabs(yj));
        r[j * n + s] = fmax(srur * fabs(yj), r0 / ewt);
    }
}

__global__ void k_fd_perturb(long n, int j, const double* y, const double* r, double* ytmp)
{
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    if (j > 0) ytmp[(j - 1) * n + s] = y[(j - 1) * n + s];
    ytmp[j * n + s] = y[j * n + s] + r[j * n + s];
}

__global__ void k_fd_col(long n, int nsp, int j, const double* dyj, const double* dy0, const double* r,
                         double* jac, long j_si, long j_ss)
{
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const double rj = r[j * n + s];
    for (int i = 0; i < nsp; ++i)
        jac[(long)(i + nsp * j) * j_si + s * j_ss] = (dyj[i * n + s] - dy0[i * n + s]) / rj;
}

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t upload(const std::vector<T>& h)
    {
        n = h.size();
        hipError_t e = hipMalloc((void**)&p, sizeof(T) * (n ? n : 1));
        if (e != hipSuccess) return e;
        if (n) e = hipMemcpy(p, h.data(), sizeof(T) * n, hipMemcpyHostToDevice);
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

struct Workspace {
    long cap = 0;
    double *pres = nullptr, *y = nullptr, *conc = nullptr, *fwd = nullptr, *rev = nullptr,
           *pm = nullptr, *sr = nullptr, *dy = nullptr, *jac = nullptr, *aux = nullptr, *T = nullptr;
    void release()
    {
        double** all[] = {&pres, &y, &conc, &fwd, &rev, &pm, &sr, &dy, &jac, &aux, &T};
        for (auto pp : all) { if (*pp) (void)hipFree(*pp); *pp = nullptr; }
        cap = 0;
    }
};

}  // namespace

struct pj_mech {
    Programs P;
    DevMech M;
    bool on_device = false;
    int device = -1;
    DevBuf<double> sp, rd, rtd, eff_am1, kcg, plog, sri, cheb, net_nu, sp_nu, gen_nu;
    DevBuf<int32_t> ri, rti, eff_sp, net_sp, sp_ptr, sp_rxn, fin_tgt, fin_part, fin_cnt, gen_sp;
    DevBuf<uint32_t> sched, ecol;
    DevBuf<uint16_t> smap;
    DevBuf<int32_t> ecol_ptr;
    Schedule S;                // host copy; rebuilt when the launch geometry changes
    int sched_nw = 0, sched_il = 0;
    bool sched_on_device = false;
    int ts = 0, nt = 0;       // 0 = auto
    int num_cu = 256;
    Workspace ws, ws1;
    // the host batch driver (pj_run / the per-state calls) works on a stream of its own: copies, kernels and the final wait
    // are ordered on it, and nothing else of the process is touched (round 5 ended pj_run with a DEVICE-wide synchronisation:
    // every other stream of the process stalled with it)
    hipStream_t run_stream = nullptr;
    // attached register-resident specialisation (pj_lane.hip)
    void* spec_lib = nullptr;
    int (*spec_jac)(long, const double*, const double*, long, long, double*, long, long, int, void*) = nullptr;
    int (*spec_jv)(long, const double*, const double*, long, long, const double*, long, long, double*, long, long,
                   int, void*) = nullptr;   // fused J*v (pj_lane.hip), optional
    int (*spec_rates)(long, const double*, const double*, long, long, double*, double*, double*, double*, double*,
                      double*, void*) = nullptr;   // rate outputs (pj_lane.hip), optional
    bool spec_aos = false;       // the attached library writes AoS Jacobians efficiently
    // libraries that own device state (pj_rblk.hip: hand-over arrays, internal streams, staging blocks) keep it in a
    // context; this handle's own one, so that two handles of one mechanism never share scratch
    void* spec_ctx = nullptr;
    void (*spec_ctx_destroy)(void*) = nullptr;
    int (*spec_ctx_config)(void*, int, long, int, int) = nullptr;
    int (*spec_jac_c)(void*, long, const double*, const double*, long, long, double*, long, long, int, void*) = nullptr;
    int (*spec_jv_c)(void*, long, const double*, const double*, long, long, const double*, long, long, double*, long, long,
                     int, void*) = nullptr;
    int (*spec_rates_c)(void*, long, const double*, const double*, long, long, double*, double*, double*, double*, double*,
                        double*, void*) = nullptr;
    int do_spec_jac(long n, const double* p, const double* y, long y_si, long y_ss, double* j, long j_si, long j_ss, int sl,
                    void* st) const
    {
        return spec_jac_c ? spec_jac_c(spec_ctx, n, p, y, y_si, y_ss, j, j_si, j_ss, sl, st)
                          : spec_jac(n, p, y, y_si, y_ss, j, j_si, j_ss, sl, st);
    }
    int do_spec_jv(long n, const double* p, const double* y, long y_si, long y_ss, const double* v, long v_si, long v_ss,
                   double* w, long w_si, long w_ss, int sl, void* st) const
    {
        return spec_jv_c ? spec_jv_c(spec_ctx, n, p, y, y_si, y_ss, v, v_si, v_ss, w, w_si, w_ss, sl, st)
                         : spec_jv(n, p, y, y_si, y_ss, v, v_si, v_ss, w, w_si, w_ss, sl, st);
    }
    int do_spec_rates(long n, const double* p, const double* y, long y_si, long y_ss, double* c, double* f, double* r,
                      double* pm, double* sr, double* dy, void* st) const
    {
        return spec_rates_c ? spec_rates_c(spec_ctx, n, p, y, y_si, y_ss, c, f, r, pm, sr, dy, st)
                            : spec_rates(n, p, y, y_si, y_ss, c, f, r, pm, sr, dy, st);
    }
    double* jv_tmp = nullptr;    // Jacobian chunk of the unfused J*v path
    size_t jv_tmp_doubles = 0;
    int use_spec = 1;          // 0: never, 1: SoA Jacobians (default), 2: every layout
    // table-driven state-per-lane Jacobian kernel (pj_tab.h): program built at load time, scratch per handle
    TabProg tab;
    DevBuf<int32_t> tab_I;
    DevBuf<double> tab_D;
    double* tab_scr = nullptr;
    long tab_scr_ld = 0;
    // k_tab hands omega_k and the last species' d/dT sum to k_tab_fin through tab_scr (indexed by the batch-local
    // state): ONE batch at a time per handle.  Host threads are serialised by tab_mutex; on the device a batch
    // launched on another stream is ordered behind the previous one by tab_event (as pj_rblk's enter_batch does).
    std::mutex tab_mutex;
    hipEvent_t tab_event = nullptr;
    void* tab_last_stream = nullptr;
    int generic = 1;           // Jacobians without an attached library: 1 k_tab for SoA / k_eval for AoS, 0 k_eval, 2 k_tab
    int check_inputs = 0;      // 1: the *_dev entry points verify T > 0, p > 0, finite (one extra pass + a sync)
    unsigned long long* d_bad = nullptr;
};

namespace {

size_t bytes_per_state(const pj_mech* m)
{
    const size_t nsp = m->P.nsp, R = m->P.nrxn, Rr = m->P.nrev > 0 ? m->P.nrev : 1,
                 Rp = m->P.npres > 0 ? m->P.npres : 1;
    return 8 * (1 + nsp + nsp + R + Rr + Rp + nsp + nsp + nsp * nsp + 3 + 1);
}

int ensure_device(pj_mech* m)
{
    if (m->on_device) {
        // a handle (its tables, workspaces and the scratch of an attached library) lives on the device
        // that was current at its first evaluation
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev != m->device)
            return fail(PJ_EINVAL, "mechanism handle belongs to HIP device " + std::to_string(m->device) +
                                   ", current device is " + std::to_string(dev) + ": create one handle per device");
        return PJ_OK;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(PJ_ENODEV, "no HIP device available (the product path has no CPU fallback)");
    HIPCHK(hipGetDevice(&m->device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, m->device));
    m->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    Programs& P = m->P;
    HIPCHK(m->sp.upload(P.sp)); HIPCHK(m->ri.upload(P.ri)); HIPCHK(m->rd.upload(P.rd));
    HIPCHK(m->rti.upload(P.rti)); HIPCHK(m->rtd.upload(P.rtd));
    HIPCHK(m->smap.upload(P.smap)); HIPCHK(m->ecol_ptr.upload(P.ecol_ptr)); HIPCHK(m->ecol.upload(P.ecol));
    HIPCHK(m->eff_sp.upload(P.eff_sp)); HIPCHK(m->eff_am1.upload(P.eff_am1));
    HIPCHK(m->kcg.upload(P.kcg)); HIPCHK(m->plog.upload(P.plog));
    HIPCHK(m->sri.upload(P.sri)); HIPCHK(m->cheb.upload(P.cheb));
    HIPCHK(m->net_sp.upload(P.net_sp)); HIPCHK(m->net_nu.upload(P.net_nu));
    HIPCHK(m->sp_ptr.upload(P.sp_ptr)); HIPCHK(m->sp_rxn.upload(P.sp_rxn)); HIPCHK(m->sp_nu.upload(P.sp_nu));
    HIPCHK(m->gen_sp.upload(P.gen_sp)); HIPCHK(m->gen_nu.upload(P.gen_nu));
    if (m->tab.ok) {
        HIPCHK(m->tab_I.upload(m->tab.I)); HIPCHK(m->tab_D.upload(m->tab.D));
    }
    DevMech& M = m->M;
    M.gen_sp = m->gen_sp.p; M.gen_nu = m->gen_nu.p;
    for (int t = 0; t < NUTAB_N; ++t) M.nutab[t] = P.nutab[t];
    M.sp = m->sp.p; M.ri = m->ri.p; M.rd = m->rd.p; M.rti = m->rti.p; M.rtd = m->rtd.p; M.nrp = P.nrp;
    M.smap = m->smap.p; M.ecol_ptr = m->ecol_ptr.p; M.ecol = m->ecol.p; M.eff_sp = m->eff_sp.p; M.eff_am1 = m->eff_am1.p;
    M.kcg = m->kcg.p; M.plog = m->plog.p; M.sri = m->sri.p; M.cheb = m->cheb.p; M.net_sp = m->net_sp.p; M.net_nu = m->net_nu.p;
    M.sp_ptr = m->sp_ptr.p; M.sp_rxn = m->sp_rxn.p; M.sp_nu = m->sp_nu.p;
    m->on_device = true;
    return PJ_OK;
}

// host: make sure the scatter schedule matches the launch geometry (NT/64 waves x 64/TS lanes)
int ensure_schedule(pj_mech* m, int ts, int nt)
{
    const int NW = nt / 64, IL = 64 / ts;
    if (m->sched_nw == NW && m->sched_il == IL) return PJ_OK;
    if (!build_schedule(m->P, NW, IL, m->S)) return fail(PJ_EUNSUPPORTED, m->P.error);
    m->sched_nw = NW; m->sched_il = IL;
    m->sched_on_device = false;
    DevMech& M = m->M;
    M.v = m->P.vm;
    M.nv = m->P.vm.NV;
    for (int w = 0; w < 16; ++w) {
        M.sched_off[w] = m->S.off[w]; M.sched_rounds[w] = m->S.rounds[w];
        M.sched_rounds_dense[w] = m->S.rounds_dense[w];
    }
    M.nfin = (int)m->S.fin_tgt.size();
    return PJ_OK;
}

int pick_launch(pj_mech* m, int* ts, int* nt, size_t* lds)
{
    int n = m->nt > 0 ? m->nt : 256;
    int t = m->ts;
    if (t <= 0) {
        // largest tile whose working set leaves room for two workgroups per CU
        size_t budget = 80 * 1024;
        if (const char* e = getenv("PJ_LDS_BUDGET")) budget = (size_t)atol(e);
        // the tile size does not depend on the schedule except for a few partial slots
        const size_t per_state = (size_t)(m->P.vm.TB + m->P.vm.T_PART + SC_COUNT * (m->P.nsp + 1) + 64) * 8;
        t = 64;
        while (t > 1 && per_state * t > budget) t >>= 1;
    }
    int rc = ensure_schedule(m, t, n);
    if (rc) return rc;
    *ts = t; *nt = n; *lds = (size_t)m->P.vm.NSLOT * 8 * t + (size_t)3 * m->P.nsp * 8;
    return PJ_OK;
}

int upload_schedule(pj_mech* m)
{
    if (m->sched_on_device) return PJ_OK;
    m->sched.release(); m->fin_tgt.release(); m->fin_part.release(); m->fin_cnt.release();
    HIPCHK(m->sched.upload(m->S.codes));
    HIPCHK(m->fin_tgt.upload(m->S.fin_tgt));
    HIPCHK(m->fin_part.upload(m->S.fin_part));
    HIPCHK(m->fin_cnt.upload(m->S.fin_cnt));
    m->M.sched = m->sched.p; m->M.fin_tgt = m->fin_tgt.p; m->M.fin_part = m->fin_part.p; m->M.fin_cnt = m->fin_cnt.p;
    m->sched_on_device = true;
    return PJ_OK;
}

template <int TS>
int launch_ts(pj_mech* m, const Batch& B, int mode, const double* cin, const double* Tin, double* aux,
              int nt, size_t lds, hipStream_t st)
{
    static thread_local size_t configured = 0;
    if (lds > 64 * 1024 && lds > configured) {
        HIPCHK(hipFuncSetAttribute((const void*)k_eval<TS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = lds;
    }
    const long ntiles = (B.n + TS - 1) / TS;
    long grid = (long)m->num_cu * 8;
    if (grid > ntiles) grid = ntiles;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((k_eval<TS>), dim3((unsigned)grid), dim3(nt), lds, st, m->M, B, mode, cin, Tin, aux);
    HIPCHK(hipGetLastError());
    return PJ_OK;
}

// Jacobians only, no attached library: k_tab + k_tab_fin
int launch_tab(pj_mech* m, const Batch& B, hipStream_t st)
{
    const TabProg& T = m->tab;
    const long nsp = m->P.nsp;
    std::lock_guard<std::mutex> lock(m->tab_mutex);
    if (!m->tab_event) HIPCHK(hipEventCreateWithFlags(&m->tab_event, hipEventDisableTiming));
    else if (m->tab_last_stream != (void*)st) HIPCHK(hipStreamWaitEvent(st, m->tab_event, 0));
    // (from here to the event record below: an error return leaves the scratch's last user unknown -- the next call,
    // on whichever stream, waits for the event again)
    m->tab_last_stream = nullptr;
    if (m->tab_scr_ld < B.n) {
        if (m->tab_scr) { HIPCHK(hipDeviceSynchronize()); (void)hipFree(m->tab_scr); m->tab_scr = nullptr; m->tab_scr_ld = 0; }
        hipError_t e = hipMalloc((void**)&m->tab_scr, sizeof(double) * (size_t)(nsp + 1) * (size_t)B.n);
        if (e != hipSuccess) return fail(PJ_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
        m->tab_scr_ld = B.n;
    }
    TabDev X;
    X.L = T.L; X.G = T.G; X.B = T.B; X.ZERO = T.ZERO; X.TRASH = T.TRASH;
    X.I = m->tab_I.p; X.D = m->tab_D.p;
    X.scr = m->tab_scr; X.scr_ld = m->tab_scr_ld;
    static const int tab_dbg = getenv("PJ_TAB_DBG") ? atoi(getenv("PJ_TAB_DBG")) : 0;
    X.dbg = tab_dbg;
    static thread_local size_t configured = 0, configured_fin = 0;
    if (T.lds_bytes > 64 * 1024 && T.lds_bytes > configured) {
        HIPCHK(hipFuncSetAttribute((const void*)k_tab, hipFuncAttributeMaxDynamicSharedMemorySize, (int)T.lds_bytes));
        configured = T.lds_bytes;
    }
    const size_t lds_fin = sizeof(double) * (size_t)(nsp + 28) * 64;
    if (lds_fin > 64 * 1024 && lds_fin > configured_fin) {
        HIPCHK(hipFuncSetAttribute((const void*)k_tab_fin, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fin));
        configured_fin = lds_fin;
    }
    hipLaunchKernelGGL(k_tab, dim3((unsigned)((B.n + T.L - 1) / T.L)), dim3(256), T.lds_bytes, st, m->M, X, B);
    hipLaunchKernelGGL(k_tab_fin, dim3((unsigned)((B.n + 63) / 64)), dim3(256), lds_fin, st, m->M, X, B);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(m->tab_event, st));       // the scratch is busy until here
    m->tab_last_stream = (void*)st;
    return PJ_OK;
}

int launch(pj_mech* m, const Batch& B, int mode, const double* cin, const double* Tin, double* aux,
           hipStream_t st)
{
    if (B.n <= 0) return PJ_OK;
    int rc = ensure_device(m);
    if (rc) return rc;
    // Jacobians only: k_tab for lane-contiguous (SoA) blocks -- its stores are 512-byte runs there --, the
    // cooperative k_eval for AoS blocks (its native output order: 47 against 66 ms per 1e6 GRI-shaped Jacobians)
    if (mode == MODE_JAC && !cin && !Tin && !aux && m->tab.ok && !B.conc && !B.fwd && !B.rev && !B.pres_mod &&
        !B.spec_rates && !B.dy && (m->generic == 2 || (m->generic == 1 && B.j_ss == 1)))
        return launch_tab(m, B, st);
    int ts, nt;
    size_t lds;
    rc = pick_launch(m, &ts, &nt, &lds);
    if (rc) return rc;
    rc = upload_schedule(m);
    if (rc) return rc;
    if (lds > 160 * 1024)
        return fail(PJ_EUNSUPPORTED, "mechanism working set exceeds 160 KiB of LDS per state");
    switch (ts) {
        case 64: return launch_ts<64>(m, B, mode, cin, Tin, aux, nt, lds, st);
        case 32: return launch_ts<32>(m, B, mode, cin, Tin, aux, nt, lds, st);
        case 16: return launch_ts<16>(m, B, mode, cin, Tin, aux, nt, lds, st);
        case 8: return launch_ts<8>(m, B, mode, cin, Tin, aux, nt, lds, st);
        case 4: return launch_ts<4>(m, B, mode, cin, Tin, aux, nt, lds, st);
        case 2: return launch_ts<2>(m, B, mode, cin, Tin, aux, nt, lds, st);
        case 1: return launch_ts<1>(m, B, mode, cin, Tin, aux, nt, lds, st);
    }
    return fail(PJ_EINVAL, "tile_states must be a power of two <= 64");
}

// w_s = A_s v_s for a chunk of stored Jacobians (SoA, leading dimension ld): the unfused consumer,
// pyJac's sparse_multiplier (create_jacobian.py:3301-3404) applied to every state of the chunk
__global__ void k_matvec(int nsp, long n, const double* A, long ld, const double* v, long v_si, long v_ss,
                         double* w, long w_si, long w_ss)
{
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const double* vs = v + s * v_ss;
    double* ws = w + s * w_ss;
    for (int i = 0; i < nsp; ++i) {
        double acc = 0.0;
        for (int j = 0; j < nsp; ++j) acc += A[(long)(i + nsp * j) * ld + s] * vs[j * v_si];
        ws[i * w_si] = acc;
    }
}

void set_layout(long n, int rows, int layout, long* si, long* ss)
{
    if (layout == PJ_LAYOUT_AOS) { *si = 1; *ss = rows; }
    else { *si = n; *ss = 1; }
}

int alloc_ws(pj_mech* m, Workspace& w, long cap)
{
    if (w.cap >= cap) return PJ_OK;
    w.release();
    const size_t nsp = m->P.nsp, R = m->P.nrxn, Rr = m->P.nrev > 0 ? m->P.nrev : 1,
                 Rp = m->P.npres > 0 ? m->P.npres : 1;
    struct { double** p; size_t rows; } a[] = {
        {&w.pres, 1}, {&w.y, nsp}, {&w.conc, nsp}, {&w.fwd, R}, {&w.rev, Rr}, {&w.pm, Rp},
        {&w.sr, nsp}, {&w.dy, nsp}, {&w.jac, nsp * nsp}, {&w.aux, 3}, {&w.T, 1}};
    for (auto& x : a) {
        hipError_t e = hipMalloc((void**)x.p, sizeof(double) * x.rows * (size_t)cap);
        if (e != hipSuccess) {
            w.release();
            return fail(PJ_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
        }
    }
    w.cap = cap;
    return PJ_OK;
}

}  // namespace

extern "C" {

const char* pj_last_error(void) { return g_err.c_str(); }
const char* pj_version(void) { return "pyjac_amd 0.1 (gfx950)"; }

int pj_mech_create(const int32_t* I, long nI, const double* D, long nD, pj_mech** out)
{
    if (!I || !D || !out) return fail(PJ_EINVAL, "null argument");
    pj_mech* m = nullptr;
    try {       // nothing may unwind through the C ABI
        m = new pj_mech();
        if (!build_programs(I, nI, D, nD, m->P)) {
            std::string e = m->P.error;
            delete m;
            return fail(PJ_EUNSUPPORTED, e);
        }
    } catch (const std::exception& ex) {
        delete m;
        return fail(PJ_ENOMEM, std::string("mechanism tables: ") + ex.what());
    }
    try {
        // the run-time program of k_tab (a mechanism it cannot hold falls back to k_eval: tab.ok stays false)
        (void)build_tab_program(m->P, 156 * 1024, m->tab, getenv("PJ_TAB_L") ? atoi(getenv("PJ_TAB_L")) : 0);
    } catch (const std::exception&) {
        m->tab.ok = false;
    }
    if (const char* e = getenv("PJ_GENERIC")) m->generic = atoi(e);
    DevMech& M = m->M;
    memset(&M, 0, sizeof(M));
    M.nsp = m->P.nsp; M.nrxn = m->P.nrxn; M.ng = m->P.ng; M.ne = m->P.ne; M.nv = m->P.vm.NV;
    M.lastq_rxn = m->P.lastq_rxn; M.sum_last = 0; M.v = m->P.vm;
    if (const char* e = getenv("PJ_TS")) m->ts = atoi(e);
    if (const char* e = getenv("PJ_NT")) m->nt = atoi(e);
    *out = m;
    return PJ_OK;
}

int pj_mech_load(const char* path, pj_mech** out)
{
    FILE* f = fopen(path, "rb");
    if (!f) return fail(PJ_EIO, std::string("cannot open ") + path);
    uint64_t hdr[2];
    if (fread(hdr, sizeof(uint64_t), 2, f) != 2) { fclose(f); return fail(PJ_EIO, "short table file"); }
    if (hdr[0] < (uint64_t)HDR || hdr[0] > (1ull << 28) || hdr[1] > (1ull << 28)) {
        fclose(f);
        return fail(PJ_EIO, "not a mechanism table file (implausible sizes)");
    }
    try {
        std::vector<int32_t> I(hdr[0]);
        std::vector<double> D(hdr[1]);
        bool ok = fread(I.data(), 4, I.size(), f) == I.size() && fread(D.data(), 8, D.size(), f) == D.size();
        fclose(f);
        if (!ok) return fail(PJ_EIO, "short table file");
        return pj_mech_create(I.data(), (long)I.size(), D.data(), (long)D.size(), out);
    } catch (const std::exception& ex) {
        fclose(f);
        return fail(PJ_ENOMEM, std::string("table file: ") + ex.what());
    }
}

static void detach_spec(pj_mech* m)
{
    if (m->spec_ctx && m->spec_ctx_destroy) m->spec_ctx_destroy(m->spec_ctx);      // synchronises, frees its device state
    m->spec_ctx = nullptr; m->spec_ctx_destroy = nullptr; m->spec_ctx_config = nullptr;
    m->spec_jac_c = nullptr; m->spec_jv_c = nullptr; m->spec_rates_c = nullptr;
    m->spec_jac = nullptr; m->spec_jv = nullptr; m->spec_rates = nullptr;
    if (m->spec_lib) dlclose(m->spec_lib);
    m->spec_lib = nullptr;
}

void pj_mech_destroy(pj_mech* m)
{
    if (m && m->jv_tmp) { (void)hipFree(m->jv_tmp); m->jv_tmp = nullptr; }
    if (!m) return;
    if (m->on_device) {
        m->sp.release(); m->rd.release(); m->rtd.release(); m->rti.release();
        m->smap.release(); m->ecol_ptr.release(); m->ecol.release(); m->eff_am1.release(); m->kcg.release(); m->plog.release(); m->sri.release(); m->cheb.release();
        m->net_nu.release(); m->sp_nu.release(); m->sched.release(); m->ri.release(); m->eff_sp.release();
        m->fin_tgt.release(); m->fin_part.release(); m->fin_cnt.release();
        m->net_sp.release(); m->sp_ptr.release(); m->sp_rxn.release(); m->gen_sp.release(); m->gen_nu.release();
        m->tab_I.release(); m->tab_D.release();
        if (m->tab_scr) (void)hipFree(m->tab_scr);
        if (m->tab_event) (void)hipEventDestroy(m->tab_event);
        if (m->d_bad) (void)hipFree(m->d_bad);
        m->ws.release(); m->ws1.release();
    }
    if (m->run_stream) { (void)hipStreamSynchronize(m->run_stream); (void)hipStreamDestroy(m->run_stream); m->run_stream = nullptr; }
    detach_spec(m);
    delete m;
}

int pj_mech_nsp(const pj_mech* m) { return m->P.nsp; }
int pj_mech_fwd_rates(const pj_mech* m) { return m->P.nrxn; }
int pj_mech_rev_rates(const pj_mech* m) { return m->P.nrev; }
int pj_mech_pres_mod_rates(const pj_mech* m) { return m->P.npres; }

int pj_mech_set_sum_last_species(pj_mech* m, int on) { m->M.sum_last = on ? 1 : 0; return PJ_OK; }

int pj_mech_set_generic_kernel(pj_mech* m, int which)
{
    if (!m || which < 0 || which > 2) return fail(PJ_EINVAL, "bad argument");
    if (which == 2 && !m->tab.ok) return fail(PJ_EUNSUPPORTED, "k_tab cannot hold this mechanism: " + m->tab.error);
    m->generic = which;
    return PJ_OK;
}

int pj_mech_set_check_inputs(pj_mech* m, int on) { if (!m) return fail(PJ_EINVAL, "bad argument"); m->check_inputs = on ? 1 : 0; return PJ_OK; }

// the precondition pass of the *_dev entry points (only when switched on: it synchronises the stream)
static int check_inputs(pj_mech* m, long n, const double* d_pres, const double* d_y, int y_layout, void* stream)
{
    if (!m->check_inputs) return PJ_OK;
    int rc = ensure_device(m);
    if (rc) return rc;
    long y_si, y_ss;
    set_layout(n, m->P.nsp, y_layout, &y_si, &y_ss);
    if (!m->d_bad) HIPCHK(hipMalloc((void**)&m->d_bad, sizeof(unsigned long long)));
    unsigned long long bad = ~0ull;
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(m->d_bad, &bad, sizeof(bad), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_check_inputs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, m->P.nsp, d_pres, d_y, y_si,
                       y_ss, m->d_bad);
    HIPCHK(hipMemcpyAsync(&bad, m->d_bad, sizeof(bad), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (bad != ~0ull)
        return fail(PJ_EINVAL, "state " + std::to_string(bad) + ": T and p must be positive and every input finite");
    return PJ_OK;
}

unsigned long long pj_mech_spec_hash(const pj_mech* m) { return programs_hash(m->P); }

int pj_mech_emit_spec(const pj_mech* m, const char* header_path)
{
    FILE* f = fopen(header_path, "w");
    if (!f) return fail(PJ_EIO, std::string("cannot write ") + header_path);
    const std::string h = emit_spec_header(m->P);
    const bool ok = fwrite(h.data(), 1, h.size(), f) == h.size();
    fclose(f);
    return ok ? PJ_OK : fail(PJ_EIO, "short write");
}

int pj_mech_emit_rows_spec(const pj_mech* m, const char* header_path, int acc_budget)
{
    if (acc_budget < 8) return fail(PJ_EINVAL, "accumulator budget too small");
    FILE* f = fopen(header_path, "w");
    if (!f) return fail(PJ_EIO, std::string("cannot write ") + header_path);
    const std::string h = emit_spec_header(m->P) + emit_rows_tables(m->P, acc_budget);
    const bool ok = fwrite(h.data(), 1, h.size(), f) == h.size();
    fclose(f);
    return ok ? PJ_OK : fail(PJ_EIO, "short write");
}

int pj_mech_set_kc_factors(pj_mech* m, const double* rows, long n)
{
    if (!m || n < 0 || (n > 0 && (!rows || n != (long)m->P.nsp * KCW))) return fail(PJ_EINVAL, "kc factor rows: [nsp][15] doubles");
    try {
        m->P.kcf.assign(rows, rows + n);
    } catch (const std::exception& ex) {
        return fail(PJ_ENOMEM, ex.what());
    }
    for (double v : m->P.kcf)
        if (!(v == v) || v > 1e300 || v < -1e300) { m->P.kcf.clear(); return fail(PJ_EINVAL, "kc factor rows: not finite"); }
    return PJ_OK;
}

int pj_mech_emit_rblk_spec(const pj_mech* m, const char* header_path, int acc_budget, int fuse, int block, int halves,
                           int single, int rate_block, int rate_c_lds, int rate_groups, double cost_visit,
                           double cost_entry, int* counts)
{
    if (acc_budget < 8) return fail(PJ_EINVAL, "accumulator budget too small");
    if (block < 1 || rate_block < 1 || fuse < 1 || (halves != 1 && halves != 2 && halves != 4 && halves != 8)) return fail(PJ_EINVAL, "kernel plan options");
    RblkPlanOpts O;
    O.fuse = fuse; O.block = block; O.halves = halves; O.single = single != 0; O.rate_block = rate_block; O.rate_c_lds = rate_c_lds;
    O.rate_groups = rate_groups;
    if (cost_visit > 0.0) O.cost_visit = cost_visit;
    if (cost_entry > 0.0) O.cost_entry = cost_entry;
    RblkPlan plan;
    const std::string h = emit_spec_header(m->P) + emit_rows_tables(m->P, acc_budget, &O, &plan);
    FILE* f = fopen(header_path, "w");
    if (!f) return fail(PJ_EIO, std::string("cannot write ") + header_path);
    const bool ok = fwrite(h.data(), 1, h.size(), f) == h.size();
    fclose(f);
    if (counts) {
        counts[0] = plan.n_row_kernels; counts[1] = plan.n_rate_kernels; counts[2] = plan.n_pre;
        counts[3] = plan.n_blocks; counts[4] = plan.n_visits;
    }
    return ok ? PJ_OK : fail(PJ_EIO, "short write");
}

int pj_mech_attach_spec(pj_mech* m, const char* library_path)
{
    // RTLD_NODELETE: a specialisation library owns device state (hand-over arrays, internal streams,
    // its registered code objects and their scratch); it stays resident when the last mechanism that
    // uses it is destroyed, and the next one finds that state instead of a fresh, leaking copy
    // (re-loading a library with MB-sized kernels made the next large launch fail with
    // HSA_STATUS_ERROR_OUT_OF_RESOURCES).
    void* lib = dlopen(library_path, RTLD_NOW | RTLD_LOCAL | RTLD_NODELETE);
    if (!lib) return fail(PJ_EIO, std::string("dlopen: ") + dlerror());
    auto hash = (unsigned long long (*)(void))dlsym(lib, "pj_spec_hash");
    auto jac = (decltype(m->spec_jac))dlsym(lib, "pj_spec_jacobian");
    if (!hash || !jac) { dlclose(lib); return fail(PJ_EINVAL, "not a pj_lane specialisation library"); }
    if (hash() != programs_hash(m->P)) {
        dlclose(lib);
        return fail(PJ_EINVAL, "specialisation was built for a different mechanism (hash mismatch)");
    }
    detach_spec(m);
    m->spec_lib = lib;
    m->spec_jac = jac;
    m->spec_jv = (decltype(m->spec_jv))dlsym(lib, "pj_spec_jacvec");
    m->spec_rates = (decltype(m->spec_rates))dlsym(lib, "pj_spec_rates");
    // a library with device state of its own (pj_rblk.hip) gives this handle a context of its own
    auto ctx_create = (void* (*)(void))dlsym(lib, "pj_spec_ctx_create");
    m->spec_ctx_destroy = (decltype(m->spec_ctx_destroy))dlsym(lib, "pj_spec_ctx_destroy");
    m->spec_jac_c = (decltype(m->spec_jac_c))dlsym(lib, "pj_spec_jacobian_ctx");
    if (ctx_create && m->spec_ctx_destroy && m->spec_jac_c) {
        m->spec_ctx = ctx_create();
        if (!m->spec_ctx) { detach_spec(m); return fail(PJ_ENOMEM, "specialisation context"); }
        m->spec_ctx_config = (decltype(m->spec_ctx_config))dlsym(lib, "pj_spec_ctx_config");
        m->spec_jv_c = (decltype(m->spec_jv_c))dlsym(lib, "pj_spec_jacvec_ctx");
        m->spec_rates_c = (decltype(m->spec_rates_c))dlsym(lib, "pj_spec_rates_ctx");
    } else {
        m->spec_ctx_destroy = nullptr; m->spec_jac_c = nullptr;
    }
    auto fast_aos = (int (*)(void))dlsym(lib, "pj_spec_fast_aos");
    m->spec_aos = fast_aos && fast_aos();
    return PJ_OK;
}

int pj_mech_has_spec(const pj_mech* m) { return m->spec_jac != nullptr; }
int pj_mech_use_spec(pj_mech* m, int on)
{
    if (on < 0 || on > 2) return fail(PJ_EINVAL, "use_spec: 0, 1 or 2");
    m->use_spec = on;
    return PJ_OK;
}

int pj_mech_set_spec_launch(pj_mech* m, int streams, long chunk_states, int split_tail, int aos_direct)
{
    if (!m) return fail(PJ_EINVAL, "bad argument");
    if (!m->spec_ctx_config) return fail(PJ_EINVAL, "the attached library has no launch settings (none attached, or pj_lane)");
    if (m->spec_ctx_config(m->spec_ctx, streams, chunk_states, split_tail, aos_direct))
        return fail(PJ_EINVAL, "spec launch settings: at most 8 internal streams");
    return PJ_OK;
}

int pj_mech_set_launch(pj_mech* m, int tile_states, int threads)
{
    if (tile_states < 0 || tile_states > 64 || (tile_states & (tile_states - 1)))
        return fail(PJ_EINVAL, "tile_states must be 0 or a power of two <= 64");
    if (threads < 0 || threads > 256 || threads % 64)
        return fail(PJ_EINVAL, "threads must be 0 or a multiple of 64 <= 256");
    m->ts = tile_states; m->nt = threads;
    return PJ_OK;
}

int pj_mech_get_launch(const pj_mech* m, int* tile_states, int* threads, int* lds_bytes)
{
    int ts, nt; size_t lds;
    int rc = pick_launch(const_cast<pj_mech*>(m), &ts, &nt, &lds);
    if (rc) return rc;
    if (tile_states) *tile_states = ts;
    if (threads) *threads = nt;
    if (lds_bytes) *lds_bytes = (int)lds;
    return PJ_OK;
}

int pj_eval_jacobian_dev(pj_mech* m, long n, const double* d_pres, const double* d_y, int y_layout,
                         double* d_jac, int jac_layout, void* stream)
{
    if (!m || n < 0) return fail(PJ_EINVAL, "bad argument");
    if (n == 0) return PJ_OK;
    if (!d_pres || !d_y || !d_jac) return fail(PJ_EINVAL, "null device pointer");
    if (const int rc = check_inputs(m, n, d_pres, d_y, y_layout, stream)) return rc;
    Batch B;
    memset(&B, 0, sizeof(B));
    B.n = n; B.pres = d_pres; B.y = d_y; B.jac = d_jac; B.o_ld = n;
    set_layout(n, m->P.nsp, y_layout, &B.y_si, &B.y_ss);
    set_layout(n, m->P.nsp * m->P.nsp, jac_layout, &B.j_si, &B.j_ss);
    // the state-per-lane kernels write lane-contiguous (SoA) blocks; AoS Jacobians go to them only
    // if the library transposes through LDS (pj_lane.hip) -- otherwise that is the cooperative
    // kernel's native output (measured 3x faster there) -- or when forced (use_spec == 2)
    if (m->spec_jac && (m->use_spec == 2 ||
                        (m->use_spec == 1 && (jac_layout == PJ_LAYOUT_SOA || m->spec_aos)))) {
        int rc = ensure_device(m);
        if (rc) return rc;
        if (m->do_spec_jac(n, d_pres, d_y, B.y_si, B.y_ss, d_jac, B.j_si, B.j_ss, m->M.sum_last, stream))
            return fail(PJ_EHIP, "specialised kernel launch failed");
        return PJ_OK;
    }
    static const int abl = getenv("PJ_ABLATE") ? atoi(getenv("PJ_ABLATE")) : 0;
    return launch(m, B, MODE_JAC | abl, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

int pj_eval_jacobian_vec_dev(pj_mech* m, long n, const double* d_pres, const double* d_y, int y_layout,
                             const double* d_v, double* d_w, int vw_layout, void* stream)
{
    if (!m || n < 0) return fail(PJ_EINVAL, "bad argument");
    if (n == 0) return PJ_OK;
    if (!d_pres || !d_y || !d_v || !d_w) return fail(PJ_EINVAL, "null device pointer");
    if (const int rc0 = check_inputs(m, n, d_pres, d_y, y_layout, stream)) return rc0;
    const int nsp = m->P.nsp;
    long y_si, y_ss, v_si, v_ss;
    set_layout(n, nsp, y_layout, &y_si, &y_ss);
    set_layout(n, nsp, vw_layout, &v_si, &v_ss);
    int rc = ensure_device(m);
    if (rc) return rc;
    if (m->spec_jv && m->use_spec) {
        if (m->do_spec_jv(n, d_pres, d_y, y_si, y_ss, d_v, v_si, v_ss, d_w, v_si, v_ss, m->M.sum_last, stream))
            return fail(PJ_EHIP, "specialised kernel launch failed");
        return PJ_OK;
    }
    // unfused: Jacobians of a chunk into a temporary SoA block, then the mat-vec kernel
    long chunk = (long)((512UL << 20) / (8UL * nsp * nsp));
    chunk = chunk / 256 * 256;
    if (chunk < 256) chunk = 256;
    if (chunk > n) chunk = n;
    const size_t need = (size_t)chunk * nsp * nsp;
    if (m->jv_tmp_doubles < need) {
        if (m->jv_tmp) { HIPCHK(hipDeviceSynchronize()); (void)hipFree(m->jv_tmp); m->jv_tmp = nullptr; m->jv_tmp_doubles = 0; }
        HIPCHK(hipMalloc((void**)&m->jv_tmp, sizeof(double) * need));
        m->jv_tmp_doubles = need;
    }
    for (long s0 = 0; s0 < n; s0 += chunk) {
        const long c = s0 + chunk < n ? chunk : n - s0;
        if (m->spec_jac && m->use_spec) {
            if (m->do_spec_jac(c, d_pres + s0, d_y + s0 * y_ss, y_si, y_ss, m->jv_tmp, c, 1, m->M.sum_last, stream))
                return fail(PJ_EHIP, "specialised kernel launch failed");
        } else {
            Batch B;
            memset(&B, 0, sizeof(B));
            B.n = c; B.pres = d_pres + s0; B.y = d_y + s0 * y_ss; B.jac = m->jv_tmp; B.o_ld = c;
            B.y_si = y_si; B.y_ss = y_ss; B.j_si = c; B.j_ss = 1;
            rc = launch(m, B, MODE_JAC, nullptr, nullptr, nullptr, (hipStream_t)stream);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(k_matvec, dim3((unsigned)((c + 255) / 256)), dim3(256), 0, (hipStream_t)stream, nsp, c,
                           m->jv_tmp, c, d_v + s0 * v_ss, v_si, v_ss, d_w + s0 * v_ss, v_si, v_ss);
    }
    HIPCHK(hipGetLastError());
    return PJ_OK;
}

int pj_eval_rates_dev(pj_mech* m, long n, const double* d_pres, const double* d_y, int y_layout,
                      double* d_conc, double* d_fwd, double* d_rev, double* d_pres_mod,
                      double* d_spec_rates, double* d_dy, void* stream)
{
    if (!m || n < 0) return fail(PJ_EINVAL, "bad argument");
    if (n == 0) return PJ_OK;
    if (!d_pres || !d_y) return fail(PJ_EINVAL, "null device pointer");
    if (const int rc = check_inputs(m, n, d_pres, d_y, y_layout, stream)) return rc;
    Batch B;
    memset(&B, 0, sizeof(B));
    B.n = n; B.pres = d_pres; B.y = d_y; B.o_ld = n;
    set_layout(n, m->P.nsp, y_layout, &B.y_si, &B.y_ss);
    B.conc = d_conc; B.fwd = d_fwd; B.rev = d_rev; B.pres_mod = d_pres_mod;
    B.spec_rates = d_spec_rates; B.dy = d_dy;
    if (m->spec_rates && m->use_spec) {
        int rc = ensure_device(m);
        if (rc) return rc;
        if (m->do_spec_rates(n, d_pres, d_y, B.y_si, B.y_ss, d_conc, d_fwd, d_rev, d_pres_mod, d_spec_rates, d_dy, stream))
            return fail(PJ_EHIP, "specialised kernel launch failed");
        return PJ_OK;
    }
    return launch(m, B, 0, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

int pj_eval_fd_jacobian_dev(pj_mech* m, long n, const double* d_pres, const double* d_y, double* d_jac,
                            int jac_layout, void* stream)
{
    if (!m || n < 0) return fail(PJ_EINVAL, "bad argument");
    if (n == 0) return PJ_OK;
    if (!d_pres || !d_y || !d_jac) return fail(PJ_EINVAL, "null device pointer");
    int rc = ensure_device(m);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t nsp = m->P.nsp, bytes = sizeof(double) * nsp * (size_t)n;
    double* buf = nullptr;
    hipError_t e = hipMalloc((void**)&buf, 4 * bytes);
    if (e != hipSuccess) return fail(PJ_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    double *ytmp = buf, *dy0 = buf + nsp * n, *dyj = dy0 + nsp * n, *r = dyj + nsp * n;
    long j_si, j_ss;
    set_layout(n, (int)(nsp * nsp), jac_layout, &j_si, &j_ss);
    const unsigned grid = (unsigned)((n + 255) / 256);
    rc = pj_eval_rates_dev(m, n, d_pres, d_y, PJ_LAYOUT_SOA, nullptr, nullptr, nullptr, nullptr, nullptr, dy0, stream);
    if (!rc) {
        hipLaunchKernelGGL(k_fd_setup, dim3(grid), dim3(256), 0, st, n, (int)nsp, d_y, dy0, r);
        (void)hipMemcpyAsync(ytmp, d_y, bytes, hipMemcpyDeviceToDevice, st);
        for (int j = 0; j < (int)nsp && !rc; ++j) {
            hipLaunchKernelGGL(k_fd_perturb, dim3(grid), dim3(256), 0, st, n, j, d_y, r, ytmp);
            rc = pj_eval_rates_dev(m, n, d_pres, ytmp, PJ_LAYOUT_SOA, nullptr, nullptr, nullptr, nullptr, nullptr, dyj, stream);
            hipLaunchKernelGGL(k_fd_col, dim3(grid), dim3(256), 0, st, n, (int)nsp, j, dyj, dy0, r, d_jac, j_si, j_ss);
        }
    }
    (void)hipStreamSynchronize(st);
    (void)hipFree(buf);
    if (rc) return rc;
    return hipGetLastError() == hipSuccess ? PJ_OK : fail(PJ_EHIP, "finite-difference kernels failed");
}

#ifdef PJ_TIMING
// debug build only: per-phase cycle counts of the first 64 workgroups (d_dbg: 640 doubles)
int pj_debug_phase_cycles(pj_mech* m, long n, const double* d_pres, const double* d_y, int y_layout,
                          double* d_jac, int jac_layout, double* d_dbg)
{
    Batch B;
    memset(&B, 0, sizeof(B));
    B.n = n; B.pres = d_pres; B.y = d_y; B.jac = d_jac; B.o_ld = n;
    set_layout(n, m->P.nsp, y_layout, &B.y_si, &B.y_ss);
    set_layout(n, m->P.nsp * m->P.nsp, jac_layout, &B.j_si, &B.j_ss);
    return launch(m, B, MODE_JAC, nullptr, d_dbg, nullptr, nullptr);
}
#endif

// ---- batched LU / Newton solves on per-state blocks (pj_lu.h) ----
static int lu_cus()
{
    static int cus[64] = {};        // per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev] &&
        (hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus[dev] <= 0)) cus[dev] = 256;
    return cus[dev];
}
static int lu_call(int nsp, long n, const double* a, int a_layout, double gamma, double* lu, int* perm, const double* b, double* x,
                   int vec_layout, int mode, void* stream)
{
    if (n < 0 || nsp < 1) return fail(PJ_EINVAL, "bad argument");
    if ((a_layout != PJ_LAYOUT_SOA && a_layout != PJ_LAYOUT_AOS) || (vec_layout != PJ_LAYOUT_SOA && vec_layout != PJ_LAYOUT_AOS))
        return fail(PJ_EINVAL, "layout: PJ_LAYOUT_SOA or PJ_LAYOUT_AOS");
    if (nsp > pj::LU_MAX_LDS) return fail(PJ_EUNSUPPORTED, "batched LU: at most 140 rows (the block has to fit the LDS)");
    if (n == 0) return PJ_OK;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return fail(PJ_ENODEV, "no HIP device");
    pj::LuLay Y;
    set_layout(n, nsp * nsp, a_layout, &Y.a_si, &Y.a_ss);
    set_layout(n, nsp, vec_layout, &Y.v_si, &Y.v_ss);
    if (pj::lu_launch_x(nsp, n, a, Y, gamma, lu, perm, b, x, mode, lu_cus(), (hipStream_t)stream)) return fail(PJ_EHIP, "batched LU launch failed");
    return hipGetLastError() == hipSuccess ? PJ_OK : fail(PJ_EHIP, "batched LU launch failed");
}
int pj_lu_factor_dev(int nsp, long n, const double* d_a, int a_layout, double gamma, double* d_lu, int* d_perm, void* stream)
{
    if (n > 0 && (!d_a || !d_lu || !d_perm)) return fail(PJ_EINVAL, "null device pointer");
    if (d_a == d_lu && a_layout != PJ_LAYOUT_AOS) return fail(PJ_EINVAL, "in-place factorisation needs the per-state layout");
    return lu_call(nsp, n, d_a, a_layout, gamma, d_lu, d_perm, nullptr, nullptr, PJ_LAYOUT_AOS, pj::LU_FACTOR, stream);
}
int pj_lu_solve_dev(int nsp, long n, const double* d_lu, const int* d_perm, const double* d_b, double* d_x, int vec_layout,
                    void* stream)
{
    if (n > 0 && (!d_lu || !d_perm || !d_b || !d_x)) return fail(PJ_EINVAL, "null device pointer");
    return lu_call(nsp, n, nullptr, PJ_LAYOUT_AOS, 0.0, const_cast<double*>(d_lu), const_cast<int*>(d_perm), d_b, d_x, vec_layout,
                   pj::LU_PREFACTORED | pj::LU_SOLVE, stream);
}
int pj_newton_solve_dev(int nsp, long n, const double* d_a, int a_layout, double gamma, const double* d_b, double* d_x,
                        int vec_layout, double* d_lu, int* d_perm, void* stream)
{
    if (n > 0 && (!d_a || !d_b || !d_x)) return fail(PJ_EINVAL, "null device pointer");
    if ((d_lu == nullptr) != (d_perm == nullptr)) return fail(PJ_EINVAL, "d_lu and d_perm: both or neither");
    // the same restriction as pj_lu_factor_dev: stored factors are per state, and a wavefront's factor stores would
    // overwrite batch-layout entries of states that other wavefronts have not loaded yet
    if (d_lu == d_a && a_layout != PJ_LAYOUT_AOS) return fail(PJ_EINVAL, "in-place factorisation needs the per-state layout");
    // the solution must not overlap the blocks or the factors (blocks of other states are still being read)
    {
        const char *x0 = (const char*)d_x, *x1 = x0 + sizeof(double) * (size_t)nsp * (size_t)n;
        const char *a0 = (const char*)d_a, *a1 = a0 + sizeof(double) * (size_t)nsp * (size_t)nsp * (size_t)n;
        if (nsp > 0 && n > 0 && x0 < a1 && a0 < x1) return fail(PJ_EINVAL, "d_x overlaps d_a");
        if (d_lu && nsp > 0 && n > 0) {
            const char *l0 = (const char*)d_lu, *l1 = l0 + sizeof(double) * (size_t)nsp * (size_t)nsp * (size_t)n;
            if (x0 < l1 && l0 < x1) return fail(PJ_EINVAL, "d_x overlaps d_lu");
        }
    }
    return lu_call(nsp, n, d_a, a_layout, gamma, d_lu, d_perm, d_b, d_x, vec_layout, pj::LU_FACTOR | pj::LU_SOLVE, stream);
}

int pj_time_jacobian_dev(pj_mech* m, long n, const double* d_pres, const double* d_y, int y_layout,
                         double* d_jac, int jac_layout, void* stream, int iters, double* ms_per_launch)
{
    if (iters < 1 || !ms_per_launch) return fail(PJ_EINVAL, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) {
        int rc = pj_eval_jacobian_dev(m, n, d_pres, d_y, y_layout, d_jac, jac_layout, stream);
        if (rc) return rc;
    }
    HIPCHK(hipEventRecord(e1, st));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_per_launch = (double)ms / iters;
    return PJ_OK;
}

// ---- pyjacob.cu:84-188 ----
int pj_init(pj_mech* m, int num)
{
    if (!m || num < 1) return fail(PJ_EINVAL, "bad argument");
    int rc = ensure_device(m);
    if (rc) return rc;
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    const size_t per = bytes_per_state(m);
    long max_states = (long)(0.8 * (double)free_b / (double)per);   // USE_MEM 0.8, pyjacob.cu:82
    long padded = num < max_states ? num : max_states;
    padded = (padded + 63) / 64 * 64;
    if (padded > max_states) padded = max_states / 64 * 64;
    if (padded <= 0) return fail(PJ_ENOMEM, "mechanism is too large to fit one tile into device memory");
    if (padded > 2147483584L) padded = 2147483584L;
    rc = alloc_ws(m, m->ws, padded);
    if (rc) return rc;
    return (int)padded;
}

static int run_ws(pj_mech* m, Workspace& w, int num, const double* pres, const double* y, double* conc,
                  double* fwd, double* rev, double* pres_mod, double* spec_rates, double* dy,
                  double* jac, double* aux)
{
    const size_t nsp = m->P.nsp, n = (size_t)num;
    // (pyjacob.cu:139-187 copies, launches and copies back on the default stream between two cudaDeviceSynchronize-like
    // points; here everything is ordered on the handle's own non-blocking stream and the call ends with a wait for THAT
    // stream -- other streams of the process keep running)
    if (!m->run_stream) HIPCHK(hipStreamCreateWithFlags(&m->run_stream, hipStreamNonBlocking));
    hipStream_t st = m->run_stream;
    HIPCHK(hipMemcpyAsync(w.pres, pres, 8 * n, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(w.y, y, 8 * n * nsp, hipMemcpyHostToDevice, st));
    Batch B;
    memset(&B, 0, sizeof(B));
    B.n = num; B.pres = w.pres; B.y = w.y; B.y_si = num; B.y_ss = 1; B.o_ld = num;
    B.jac = w.jac; B.j_si = num; B.j_ss = 1;
    B.conc = w.conc; B.fwd = w.fwd; B.rev = w.rev; B.pres_mod = w.pm; B.spec_rates = w.sr; B.dy = w.dy;
    const bool spec = jac && m->spec_jac && m->use_spec;
    int rc = PJ_OK;
    if (!aux && m->spec_rates && m->use_spec && (spec || !jac)) {
        if (m->do_spec_rates(num, w.pres, w.y, B.y_si, B.y_ss, w.conc, w.fwd, w.rev, w.pm, w.sr, w.dy, (void*)st))
            return fail(PJ_EHIP, "specialised kernel launch failed");
    } else {
        rc = launch(m, B, (jac && !spec) ? MODE_JAC : 0, nullptr, nullptr, aux ? w.aux : nullptr, st);
    }
    if (rc) return rc;
    if (spec && m->do_spec_jac(num, w.pres, w.y, B.y_si, B.y_ss, w.jac, B.j_si, B.j_ss, m->M.sum_last, (void*)st))
        return fail(PJ_EHIP, "specialised kernel launch failed");
    if (conc) HIPCHK(hipMemcpyAsync(conc, w.conc, 8 * n * nsp, hipMemcpyDeviceToHost, st));
    if (fwd) HIPCHK(hipMemcpyAsync(fwd, w.fwd, 8 * n * m->P.nrxn, hipMemcpyDeviceToHost, st));
    if (rev && m->P.nrev) HIPCHK(hipMemcpyAsync(rev, w.rev, 8 * n * m->P.nrev, hipMemcpyDeviceToHost, st));
    if (pres_mod && m->P.npres) HIPCHK(hipMemcpyAsync(pres_mod, w.pm, 8 * n * m->P.npres, hipMemcpyDeviceToHost, st));
    if (spec_rates) HIPCHK(hipMemcpyAsync(spec_rates, w.sr, 8 * n * nsp, hipMemcpyDeviceToHost, st));
    if (dy) HIPCHK(hipMemcpyAsync(dy, w.dy, 8 * n * nsp, hipMemcpyDeviceToHost, st));
    if (jac) HIPCHK(hipMemcpyAsync(jac, w.jac, 8 * n * nsp * nsp, hipMemcpyDeviceToHost, st));
    if (aux) HIPCHK(hipMemcpyAsync(aux, w.aux, 8 * n * 3, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return PJ_OK;
}

int pj_run(pj_mech* m, int num, int padded, const double* pres, const double* y, double* conc,
           double* fwd_rxn_rates, double* rev_rxn_rates, double* pres_mod, double* spec_rates,
           double* dy, double* jac)
{
    if (!m || !pres || !y || num < 1) return fail(PJ_EINVAL, "bad argument");
    if (m->ws.cap < num || padded < num)
        return fail(PJ_EINVAL, "pj_run: num exceeds the capacity returned by pj_init");
    return run_ws(m, m->ws, num, pres, y, conc, fwd_rxn_rates, rev_rxn_rates, pres_mod, spec_rates, dy,
                  jac, nullptr);
}

int pj_cleanup(pj_mech* m)
{
    if (m) m->ws.release();
    return PJ_OK;
}

// ---- per-state functions (pyjacob_wrapper.pyx:4-16), evaluated on the GPU ----
static int one(pj_mech* m)
{
    int rc = ensure_device(m);
    if (rc) return rc;
    return alloc_ws(m, m->ws1, 64);
}

int pj_dydt(pj_mech* m, double t, double pres, const double* y, double* dy)
{
    (void)t;
    int rc = one(m);
    if (rc) return rc;
    return run_ws(m, m->ws1, 1, &pres, y, nullptr, nullptr, nullptr, nullptr, nullptr, dy, nullptr, nullptr);
}

int pj_eval_jacob(pj_mech* m, double t, double pres, const double* y, double* jac)
{
    (void)t;
    int rc = one(m);
    if (rc) return rc;
    return run_ws(m, m->ws1, 1, &pres, y, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, jac, nullptr);
}

int pj_eval_state(pj_mech* m, double pres, const double* y, double* conc, double* fwd, double* rev, double* pres_mod,
                  double* spec_rates, double* dy, double* jac)
{
    int rc = one(m);
    if (rc) return rc;
    return run_ws(m, m->ws1, 1, &pres, y, conc, fwd, rev, pres_mod, spec_rates, dy, jac, nullptr);
}

int pj_eval_conc(pj_mech* m, double T, double pres, const double* mass_frac, double* y_N,
                 double* mw_avg, double* rho, double* conc)
{
    int rc = one(m);
    if (rc) return rc;
    std::vector<double> y(m->P.nsp);
    y[0] = T;
    for (int k = 0; k < m->P.nsp - 1; ++k) y[k + 1] = mass_frac[k];
    double aux[3];
    rc = run_ws(m, m->ws1, 1, &pres, y.data(), conc, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, aux);
    if (rc) return rc;
    if (y_N) *y_N = aux[0];
    if (mw_avg) *mw_avg = aux[1];
    if (rho) *rho = aux[2];
    return PJ_OK;
}

static int rates_from_conc(pj_mech* m, double T, double pres, const double* C, double* fwd, double* rev,
                           double* pm)
{
    int rc = one(m);
    if (rc) return rc;
    Workspace& w = m->ws1;
    if (!m->run_stream) HIPCHK(hipStreamCreateWithFlags(&m->run_stream, hipStreamNonBlocking));
    hipStream_t st = m->run_stream;         // (see run_ws: the handle's own stream, no device-wide synchronisation)
    HIPCHK(hipMemcpyAsync(w.pres, &pres, 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(w.T, &T, 8, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(w.conc, C, 8 * (size_t)m->P.nsp, hipMemcpyHostToDevice, st));
    Batch B;
    memset(&B, 0, sizeof(B));
    B.n = 1; B.pres = w.pres; B.y = w.y; B.y_si = 1; B.y_ss = 1; B.o_ld = 1;
    B.fwd = w.fwd; B.rev = w.rev; B.pres_mod = w.pm;
    rc = launch(m, B, MODE_CONC_IN, w.conc, w.T, nullptr, st);
    if (rc) return rc;
    if (fwd) HIPCHK(hipMemcpyAsync(fwd, w.fwd, 8 * (size_t)m->P.nrxn, hipMemcpyDeviceToHost, st));
    if (rev && m->P.nrev) HIPCHK(hipMemcpyAsync(rev, w.rev, 8 * (size_t)m->P.nrev, hipMemcpyDeviceToHost, st));
    if (pm && m->P.npres) HIPCHK(hipMemcpyAsync(pm, w.pm, 8 * (size_t)m->P.npres, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return PJ_OK;
}

int pj_eval_rxn_rates(pj_mech* m, double T, double pres, const double* C, double* fwd, double* rev)
{
    return rates_from_conc(m, T, pres, C, fwd, rev, nullptr);
}

int pj_get_rxn_pres_mod(pj_mech* m, double T, double pres, const double* C, double* pres_mod)
{
    return rates_from_conc(m, T, pres, C, nullptr, nullptr, pres_mod);
}

int pj_eval_spec_rates(pj_mech* m, const double* fwd, const double* rev, const double* pres_mod,
                       double* sp_rates, double* dy_N)
{
    int rc = one(m);
    if (rc) return rc;
    Workspace& w = m->ws1;
    if (!m->run_stream) HIPCHK(hipStreamCreateWithFlags(&m->run_stream, hipStreamNonBlocking));
    hipStream_t st = m->run_stream;
    HIPCHK(hipMemcpyAsync(w.fwd, fwd, 8 * (size_t)m->P.nrxn, hipMemcpyHostToDevice, st));
    if (m->P.nrev) HIPCHK(hipMemcpyAsync(w.rev, rev, 8 * (size_t)m->P.nrev, hipMemcpyHostToDevice, st));
    if (m->P.npres) HIPCHK(hipMemcpyAsync(w.pm, pres_mod, 8 * (size_t)m->P.npres, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_spec_rates, dim3(1), dim3(64), 0, st, m->M, 1L, w.fwd, w.rev, w.pm, w.sr);
    HIPCHK(hipGetLastError());
    std::vector<double> sr(m->P.nsp);
    HIPCHK(hipMemcpyAsync(sr.data(), w.sr, 8 * sr.size(), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    // eval_spec_rates writes sp_rates[0..NSP-2] and the last species through dy_N
    for (int k = 0; k < m->P.nsp - 1; ++k) sp_rates[k] = sr[k];
    if (dy_N) *dy_N = sr[m->P.nsp - 1];
    return PJ_OK;
}

}  // extern "C"
