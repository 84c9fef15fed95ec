// frp_capi.hip -- the C-ABI of libfrp_nmpc_amd.so (declared in include/frp_nmpc.h).
//
// Drop-in part: FORCESNLPsolver_normal_solve / FORCESNLPsolver_final_solve with the reference's
// struct layouts (FORCESNLPsolver_normal.h:153-301,317-323), so plan_manage's adapters
// (forces_normal.cpp:139, forces_final.cpp:138) link against this library unchanged.
// Batched part: device-pointer API used by bench.py, the tests and the host-side adapter mirror.
// There is NO CPU compute path in this file: without a HIP device every entry point fails.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include "../../include/frp_nmpc.h"
#include "frp_kernels.h"
#include "frp_model.hpp"

static_assert(sizeof(frp_forces_params) == 23600, "params layout must match FORCESNLPsolver_normal.h:153-168");
static_assert(offsetof(frp_forces_params, x0) == 72, "x0 offset");
static_assert(offsetof(frp_forces_params, all_parameters) == 2792, "all_parameters offset");
static_assert(offsetof(frp_forces_params, num_of_threads) == 23592, "num_of_threads offset");
static_assert(sizeof(frp_forces_output) == 2720, "output layout must match FORCESNLPsolver_normal.h:173-236");
static_assert(sizeof(frp_forces_info) == 136, "info layout must match FORCESNLPsolver_normal.h:241-301");
static_assert(offsetof(frp_forces_info, res_eq) == 8 && offsetof(frp_forces_info, lsit_aff) == 96 &&
              offsetof(frp_forces_info, step_aff) == 104 && offsetof(frp_forces_info, fevalstime) == 128, "info offsets");

namespace {

#define FRP_HIP(call)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            fprintf(stderr, "[frp_nmpc] HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return FRP_ERR_HIP;                                                              \
        }                                                                                    \
    } while (0)

// device allocation that is released on every exit path of the *_host helpers
struct DevMem {
    void *p = nullptr;
    DevMem() = default;
    DevMem(const DevMem &) = delete;
    DevMem &operator=(const DevMem &) = delete;
    ~DevMem() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
    template <typename T> T *as() const { return static_cast<T *>(p); }
};

bool fill_args(const frp_nmpc_batch *b, const frp_nmpc_options *opt_in, void *ws, size_t ws_bytes, frp::KernelArgs *a)
{
    if (!b || b->B <= 0 || b->N < 2 || b->N > 64 || b->M < 0 || b->MF < 0 || b->MF > b->M || b->MF > frp::FRP_MAX_FACES) return false;
    if (!b->xinit || !b->x0 || !b->params || !b->z || !b->exitflag || !b->iters || !ws) return false;
    if (b->model != FRP_MODEL_NORMAL && b->model != FRP_MODEL_FINAL) return false;
    if (ws_bytes < frp::ws_bytes(b->B, b->N, b->MF)) return false;
    frp_nmpc_options o;
    if (opt_in) o = *opt_in; else frp_nmpc_default_options(&o);
    a->B = b->B; a->N = b->N; a->M = b->M; a->MF = b->MF; a->model = b->model; a->maxit = o.maxit;
    a->tol_stat = o.tol_stat; a->tol_eq = o.tol_eq; a->tol_ineq = o.tol_ineq; a->tol_comp = o.tol_comp;
    a->mu0 = o.mu0; a->ftb = o.ftb; a->hessian = o.hessian; a->twist = o.twist;
    a->diverge_mu = o.diverge_mu > 0.0 ? o.diverge_mu : 1e3;
    a->xinit = b->xinit; a->x0 = b->x0; a->params = b->params; a->nfaces = b->nfaces;
    a->z = b->z; a->exitflag = b->exitflag; a->iters = b->iters; a->info = b->info;
    a->ws = static_cast<double *>(ws);
    a->models = b->model_per_problem;
    a->order_hint = b->order_hint;
    a->counter = nullptr; a->order = nullptr;
    a->self_reset = 0;
    a->variant_B = 0;
    a->slot_reserve = 0;
    a->done_flag = nullptr; a->done_seq = 0;
    return true;
}

// ---- single-problem context of the drop-in ABI: created lazily on first call, freed at unload
// (the reference has no init/teardown call; SURVEY 8b "Ownership"). Serialised by a mutex: the
// reference library is not re-entrant either (static work arrays).
// The inputs are staged in ONE pinned, device-mapped block and the outputs land in another:
//   in  (doubles): xinit(9) | x0(340) | all_parameters(2600) | nfaces(20 ints)
//   out (doubles): z(340) | info(FRP_INFO_STRIDE) | exitflag, iterations (2 ints)
// With at most 15 live corridor rows per stage (the kernel variants that read every input ONCE, at the start of the solve) the kernel
// reads and writes those blocks in place over PCIe -- no copy commands around the launch, and no reset launch in front of it
// (KernelArgs::self_reset): one call = one kernel launch and one stream synchronisation.  More rows (the variants that re-read
// the rows every iteration) or FRP_NMPC_DROPIN_ZEROCOPY=0: one host-to-device and one device-to-host copy through the same blocks.
//   out (doubles): plan(340) | info(FRP_INFO_STRIDE) | (exitflag, iterations) | (completion word, pad)
// In-place calls do not synchronise the stream either: the kernel stores the call's sequence number into the completion word (system-scope
// release, behind a fence and a barrier over all its outputs) and the caller spins on it -- the runtime's own completion path costs ~10 us
// of a 130 us call.  FRP_NMPC_DROPIN_SPIN=0: hipStreamSynchronize as before.
constexpr int DI_IN_DOUBLES = 9 + 340 + 2600 + 10, DI_OUT_DOUBLES = 340 + FRP_INFO_STRIDE + 2;
struct DropInCtx {
    bool ready = false;
    hipStream_t stream = nullptr;
    double *d_in = nullptr, *d_out = nullptr, *h_in = nullptr, *h_out = nullptr, *d_ws = nullptr;
    double *m_in = nullptr, *m_out = nullptr; // device addresses of the pinned blocks (null: not mapped, copies only)
    int seq = 0;                              // completion word of the last in-place call
    size_t ws_bytes = 0;
    frp_forces_extfunc probed[2] = {nullptr, nullptr}; // the callback last probed per model (a different pointer is probed again)
    bool probe_ok[2] = {false, false};
    std::mutex mtx;
    ~DropInCtx();
};
DropInCtx g_ctx;
void ctx_release();
DropInCtx::~DropInCtx() { if (ready) ctx_release(); }

void ctx_release()
{
    if (g_ctx.d_in) (void)hipFree(g_ctx.d_in);
    if (g_ctx.d_out) (void)hipFree(g_ctx.d_out);
    if (g_ctx.d_ws) (void)hipFree(g_ctx.d_ws);
    if (g_ctx.h_in) (void)hipHostFree(g_ctx.h_in);
    if (g_ctx.h_out) (void)hipHostFree(g_ctx.h_out);
    if (g_ctx.stream) (void)hipStreamDestroy(g_ctx.stream);
    g_ctx.d_in = g_ctx.d_out = g_ctx.d_ws = g_ctx.h_in = g_ctx.h_out = g_ctx.m_in = g_ctx.m_out = nullptr;
    g_ctx.stream = nullptr;
    g_ctx.ready = false;
}

int ctx_init()
{
    if (g_ctx.ready) return FRP_OK;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return FRP_ERR_NO_DEVICE;
    g_ctx.ws_bytes = std::max(frp::ws_bytes(1, FRP_N_REF, FRP_NH_REF), frp::ws_bytes(1, FRP_N_REF, 6)); // (the size depends on the face count: every shape a call can have)
    // (a failure half way leaves nothing behind: the next call starts from scratch)
    const bool ok = hipStreamCreate(&g_ctx.stream) == hipSuccess &&
                    hipMalloc(&g_ctx.d_in, DI_IN_DOUBLES * sizeof(double)) == hipSuccess &&
                    hipMalloc(&g_ctx.d_out, DI_OUT_DOUBLES * sizeof(double)) == hipSuccess &&
                    hipHostMalloc(&g_ctx.h_in, DI_IN_DOUBLES * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
                    hipHostMalloc(&g_ctx.h_out, DI_OUT_DOUBLES * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
                    hipMalloc(&g_ctx.d_ws, g_ctx.ws_bytes) == hipSuccess &&
                    hipMemset(g_ctx.d_ws, 0, 64) == hipSuccess; // (the queue head: zero between calls, see KernelArgs::self_reset)
    if (!ok) {
        fprintf(stderr, "[frp_nmpc] HIP error %s while creating the drop-in context\n", hipGetErrorString(hipGetLastError()));
        ctx_release();
        return FRP_ERR_HIP;
    }
    const char *zc = getenv("FRP_NMPC_DROPIN_ZEROCOPY");
    if (!(zc && atoi(zc) == 0)) {
        void *pi = nullptr, *po = nullptr;
        if (hipHostGetDevicePointer(&pi, g_ctx.h_in, 0) == hipSuccess && hipHostGetDevicePointer(&po, g_ctx.h_out, 0) == hipSuccess) {
            g_ctx.m_in = static_cast<double *>(pi); g_ctx.m_out = static_cast<double *>(po);
        } else {
            (void)hipGetLastError();
        }
    }
    std::memset(g_ctx.h_out, 0, DI_OUT_DOUBLES * sizeof(double)); g_ctx.seq = 0; // (the completion word starts at 0; sequence numbers at 1)
    g_ctx.ready = true;
    return FRP_OK;
}

// Probe a caller-supplied model callback against the built-in device model at fixed test points
// (all three stage classes).  The callback is never used for the solve itself.
bool probe_callback(frp_forces_extfunc fn, int model)
{
    const int stages[3] = {0, 7, 19};
    for (int s = 0; s < 3; s++) {
        double z[17] = {0.1, -0.05, 0.2, 7.5, 0.02, 0.01, -0.03, 7.3, 0.5, -0.3, 1.2, 0.4, 0.1, -0.2, 0.05, -0.1, 0.3};
        double p[130];
        std::memset(p, 0, sizeof p);
        const double p10[10] = {1, 0, 1, 0.5, -1, 0.2, 15, 3, 80, 0.2};
        std::memcpy(p, p10, sizeof p10);
        p[10] = 1.0; p[11] = 0.5; p[12] = -0.25; p[100] = 2.0;
        double y[13] = {0}, lam[64] = {0}, f = 0, g[17] = {0}, c[13] = {0}, J[221] = {0}, h[30] = {0}, Jh[510] = {0};
        fn(z, y, lam, p, &f, g, c, J, h, Jh, nullptr, stages[s], 0, 0);
        const int sc = frp::stage_class(stages[s], FRP_N_REF);
        double g2[17];
        const double f2 = frp::stage_cost(z, p, sc, model, g2);
        if (std::fabs(f - f2) > 1e-9 * (1.0 + std::fabs(f2))) return false;
        for (int i = 0; i < 17; i++)
            if (std::fabs(g[i] - g2[i]) > 1e-9 * (1.0 + std::fabs(g2[i]))) return false;
        if (sc != frp::STAGE_LAST) {
            frp::Lin L;
            double xn[9];
            frp::rk2<true>(z + 8, z, p + 3, xn, &L);
            for (int i = 0; i < 9; i++)
                if (std::fabs(c[i] - xn[i]) > 1e-9 * (1.0 + std::fabs(xn[i]))) return false;
            const double *Lc = reinterpret_cast<const double *>(&L);
            for (int i = 0; i < 9; i++) {
                for (int j = 0; j < 4; j++)
                    if (std::fabs(J[j * 13 + i] - frp::lin_B(Lc, i, j)) > 1e-9) return false;
                for (int j = 0; j < 9; j++)
                    if (std::fabs(J[(8 + j) * 13 + i] - frp::lin_A(Lc, i, j)) > 1e-9) return false;
            }
        }
        const double h0 = 1.0 * z[8] + 0.5 * z[9] - 0.25 * z[10] - 2.0;
        if (std::fabs(h[0] - h0) > 1e-9) return false;
    }
    return true;
}

int forces_solve(int model, frp_forces_params *params, frp_forces_output *output, frp_forces_info *info, FILE *fs,
                 frp_forces_extfunc fn)
{
    const auto t0 = std::chrono::steady_clock::now();
    if (!params || !output || !info) return FRP_EXIT_PARAM_VALUE;
    std::lock_guard<std::mutex> lock(g_ctx.mtx);
    std::memset(info, 0, sizeof *info);
    if (const int rc = ctx_init(); rc != FRP_OK) {
        // not a parameter error: "solver not valid on this machine" (the reference's -100) or a failed runtime call (-101)
        if (fs) fprintf(fs, rc == FRP_ERR_NO_DEVICE ? "frp_nmpc: no usable HIP device -- this library has no CPU path\n"
                                                      : "frp_nmpc: the HIP runtime failed while creating the solver context\n");
        return rc == FRP_ERR_NO_DEVICE ? FRP_EXIT_NO_DEVICE : FRP_EXIT_DEVICE_FAULT;
    }
    if (fn) {
        if (g_ctx.probed[model] != fn) {
            g_ctx.probe_ok[model] = probe_callback(fn, model);
            g_ctx.probed[model] = fn;
        }
        if (!g_ctx.probe_ok[model]) {
            if (fs) fprintf(fs, "frp_nmpc: the supplied model callback differs from the built-in device model\n");
            return FRP_EXIT_PARAM_VALUE;
        }
    }
    for (int i = 0; i < 9; i++) if (!std::isfinite(params->xinit[i])) return FRP_EXIT_PARAM_VALUE;
    hipStream_t st = g_ctx.stream;
    // stage the inputs; count the live corridor rows here (600 rows): trailing all-zero rows are the padding
    // forces_normal.cpp:127-135 writes -- with the counts the kernel variant that keeps the rows in registers runs
    double *hin = g_ctx.h_in;
    std::memcpy(hin, params->xinit, 9 * sizeof(double));
    std::memcpy(hin + 9, params->x0, 340 * sizeof(double));
    std::memcpy(hin + 349, params->all_parameters, 2600 * sizeof(double));
    int *hnf = reinterpret_cast<int *>(hin + 2949);
    int mf = 0;
    for (int k = 0; k < FRP_N_REF; k++) {
        const double *pk = params->all_parameters + (size_t)k * FRP_NPAR(FRP_NH_REF);
        int nf = FRP_NH_REF;
        while (nf > 0) {
            const double *r = pk + FRP_NPRE + 3 * (nf - 1);
            if (r[0] == 0.0 && r[1] == 0.0 && r[2] == 0.0 && pk[FRP_NPRE + 3 * FRP_NH_REF + nf - 1] >= -frp::HU) nf--;
            else break;
        }
        hnf[k] = nf;
        mf = nf > mf ? nf : mf;
    }
    auto device_fault = [&](const char *what) { // a runtime failure is not the caller's parameters: its own code, the context rebuilt on the next call
        fprintf(stderr, "[frp_nmpc] HIP error %s in the drop-in solve (%s)\n", hipGetErrorString(hipGetLastError()), what);
        if (fs) fprintf(fs, "frp_nmpc: HIP runtime failure during the solve (%s)\n", what);
        (void)hipStreamSynchronize(st);
        ctx_release();
        return FRP_EXIT_DEVICE_FAULT;
    };
    // in place over PCIe when the kernel variant reads its inputs once (mf <= 15: the corridor rows live in registers)
    const bool zero_copy = g_ctx.m_in && g_ctx.m_out && mf <= 15;
    double *din = zero_copy ? g_ctx.m_in : g_ctx.d_in, *dout = zero_copy ? g_ctx.m_out : g_ctx.d_out;
    if (!zero_copy && hipMemcpyAsync(g_ctx.d_in, hin, DI_IN_DOUBLES * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) return device_fault("copy in");
    frp_nmpc_batch b;
    std::memset(&b, 0, sizeof b);
    b.B = 1; b.N = FRP_N_REF; b.M = FRP_NH_REF; b.MF = mf; b.model = model;
    b.xinit = din; b.x0 = din + 9; b.params = din + 349; b.nfaces = reinterpret_cast<const int *>(din + 2949);
    b.z = dout; b.info = dout + 340;
    b.exitflag = reinterpret_cast<int *>(dout + 340 + FRP_INFO_STRIDE); b.iters = b.exitflag + 1;
    frp::KernelArgs a;
    // FORCES' entry point has no options argument: the one option a caller of a SINGLE solve may want -- the latency option
    // frp_nmpc_options.twist -- comes from the environment (FRP_NMPC_TWIST = m or -1; unset / 0: the plain solve)
    static const int env_twist = [] { const char *e = getenv("FRP_NMPC_TWIST"); return e ? atoi(e) : 0; }();
    frp_nmpc_options opt;
    frp_nmpc_default_options(&opt);
    opt.twist = env_twist;
    if (!fill_args(&b, &opt, g_ctx.d_ws, g_ctx.ws_bytes, &a)) return FRP_EXIT_PARAM_VALUE;
    a.self_reset = 1;
    static const bool spin_on = [] { const char *e = getenv("FRP_NMPC_DROPIN_SPIN"); return !(e && e[0] == '0'); }();
    const bool spin = zero_copy && spin_on;
    int *h_done = reinterpret_cast<int *>(g_ctx.h_out + 340 + FRP_INFO_STRIDE + 1);
    if (spin) {
        g_ctx.seq = g_ctx.seq == 0x7fffffff ? 1 : g_ctx.seq + 1;
        a.done_flag = reinterpret_cast<int *>(g_ctx.m_out + 340 + FRP_INFO_STRIDE + 1); a.done_seq = g_ctx.seq;
    }
    if (frp::launch_ipm(a, st) != hipSuccess) return device_fault("launch");
    bool seen = false;
    if (spin) { // (a solve is ~0.1 ms; the caller's thread spins for at most 2 ms -- twenty ordinary solves -- then blocks in the runtime's own wait like the call it replaces: a MAXIT solve or a device fault does not burn a real-time thread for the whole tick)
        const auto t_spin = std::chrono::steady_clock::now();
        for (unsigned n = 0;; n++) {
            if (__atomic_load_n(h_done, __ATOMIC_ACQUIRE) == g_ctx.seq) { seen = true; break; }
            if ((n & 1023u) == 1023u && std::chrono::steady_clock::now() - t_spin > std::chrono::milliseconds(2)) break;
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#elif defined(__aarch64__)
            asm volatile("yield" ::: "memory");
#endif
        }
    }
    if (!seen &&
        ((!zero_copy && hipMemcpyAsync(g_ctx.h_out, g_ctx.d_out, DI_OUT_DOUBLES * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess) ||
         hipStreamSynchronize(st) != hipSuccess))
        return device_fault("copy out / synchronise");
    std::memcpy(output->x, g_ctx.h_out, 340 * sizeof(double));
    const double *inf = g_ctx.h_out + 340;
    const int *fi = reinterpret_cast<const int *>(g_ctx.h_out + 340 + FRP_INFO_STRIDE);
    const int flag = fi[0], it = fi[1];
    info->it = it; info->it2opt = it;
    info->res_eq = inf[0]; info->res_ineq = inf[1]; info->rsnorm = inf[2]; info->rcompnorm = inf[3];
    info->pobj = inf[4]; info->mu = inf[5]; info->step_cc = inf[6];
    // affine-step quantities of the last iteration taken (FORCESNLPsolver_normal.h:275-289); no line search exists, so the two
    // backtracking counters stay 0
    info->mu_aff = inf[8]; info->sigma = inf[9]; info->step_aff = inf[10];
    // duality gap of the local QP model at the returned point = sum of slack * multiplier over all inequalities (what an
    // interior-point method drives to zero; mu is its mean); dobj = pobj - dgap (:262-270)
    info->dgap = inf[11]; info->dobj = inf[4] - inf[11];
    info->rdgap = std::fabs(inf[11]) / std::fmax(std::fabs(inf[4]), 1e-300);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    info->solvetime = secs;
    info->fevalstime = 0.0; /* model evaluation is fused into the device kernel */
    if (fs) {
        if (flag == FRP_EXIT_OPTIMAL)
            fprintf(fs, "OPTIMAL (within EQTOL=%.1e, INEQTOL=%.1e, STATTOL=%.1e, COMPTOL=%.1e)\n", a.tol_eq, a.tol_ineq, a.tol_stat, a.tol_comp);
        else if (flag == FRP_EXIT_MAXIT)
            fprintf(fs, "MAXIT - Maximum number of iterations reached, exiting.\n");
        else
            fprintf(fs, "exit flag %d\n", flag);
        fprintf(fs, "Solve time: %5.3f ms (%d iterations)\n", secs * 1e3, it);
    }
    return flag;
}

} // namespace

// Host-buffer batch: persistent device buffers and pinned staging (grown on demand, freed at unload), the batch cut into chunks
// whose copies and solves overlap on three streams: chunk c + 1 is staged and copied in while chunk c solves and chunk c - 1
// copies out.  Serialised by a mutex (one pipeline per process).
namespace {
// A few persistent worker threads for the staging copies (user buffer <-> pinned block): one thread moves ~10-15 GB/s, the PCIe
// link 50; without them the host copy is the bottleneck of the whole path.
class CopyPool {
public:
    explicit CopyPool(int n) { for (int i = 0; i < n; i++) workers_.emplace_back([this] { run(); }); }
    ~CopyPool()
    {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    void copy(void *dst, const void *src, size_t bytes)
    {
        const size_t parts = workers_.size() + 1, slice = ((bytes + parts - 1) / parts + 4095) / 4096 * 4096;
        if (bytes < (1u << 20) || workers_.empty()) { std::memcpy(dst, src, bytes); return; }
        size_t off = slice; // the caller's thread takes the first slice
        int queued = 0;
        {
            std::lock_guard<std::mutex> l(m_);
            for (; off < bytes; off += slice) { jobs_.push_back({static_cast<char *>(dst) + off, static_cast<const char *>(src) + off, std::min(slice, bytes - off), 0, 0, 0, 0, 0, 0}); queued++; }
            pending_ += queued;
        }
        cv_.notify_all();
        std::memcpy(dst, src, std::min(slice, bytes));
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [this] { return pending_ == 0; });
    }
    // `rows` parameter rows [10 | A (M x 3) | b (M)] -> [10 | A (MF x 3) | b (MF)]: the live corridor rows only (two segments per row)
    void pack_rows(double *dst, const double *src, size_t rows, int M, int MF)
    {
        const size_t parts = workers_.size() + 1, slice = (rows + parts - 1) / parts;
        Job proto{nullptr, nullptr, 0, 0, (10 + 3 * (size_t)MF) * 8, (size_t)MF * 8, (10 + 4 * (size_t)M) * 8, (10 + 4 * (size_t)MF) * 8, (10 + 3 * (size_t)M) * 8};
        auto make = [&](size_t r0, size_t n) { Job j = proto; j.d = reinterpret_cast<char *>(dst) + r0 * j.d_row; j.s = reinterpret_cast<const char *>(src) + r0 * j.s_row; j.rows = n; return j; };
        if (rows < 4096 || workers_.empty()) { run_job(make(0, rows)); return; }
        int queued = 0;
        {
            std::lock_guard<std::mutex> l(m_);
            for (size_t r0 = slice; r0 < rows; r0 += slice) { jobs_.push_back(make(r0, std::min(slice, rows - r0))); queued++; }
            pending_ += queued;
        }
        cv_.notify_all();
        run_job(make(0, std::min(slice, rows)));
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [this] { return pending_ == 0; });
    }
private:
    struct Job { char *d; const char *s; size_t n; size_t rows, a_bytes, b_bytes, s_row, d_row, s_boff; }; // rows == 0: one plain copy of n bytes
    static void run_job(const Job &j)
    {
        if (j.rows == 0) { std::memcpy(j.d, j.s, j.n); return; }
        // (rows of a few hundred bytes: plain loops the compiler vectorises, not two library calls per row)
        const size_t na = j.a_bytes / sizeof(double), nb = j.b_bytes / sizeof(double), so = j.s_boff / sizeof(double);
        for (size_t r = 0; r < j.rows; r++) {
            const double *sp = reinterpret_cast<const double *>(j.s + r * j.s_row);
            double *dp = reinterpret_cast<double *>(j.d + r * j.d_row);
            for (size_t i = 0; i < na; i++) dp[i] = sp[i];
            for (size_t i = 0; i < nb; i++) dp[na + i] = sp[so + i];
        }
    }
    void run()
    {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [this] { return stop_ || !jobs_.empty(); });
                if (stop_ && jobs_.empty()) return;
                j = jobs_.back(); jobs_.pop_back();
            }
            run_job(j);
            { std::lock_guard<std::mutex> l(m_); if (--pending_ == 0) done_.notify_all(); }
        }
    }
    std::vector<std::thread> workers_;
    std::vector<Job> jobs_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    int pending_ = 0;
    bool stop_ = false;
};
CopyPool &copy_pool()
{
    static CopyPool pool([] { const unsigned hc = std::thread::hardware_concurrency(); return (int)std::min(7u, hc > 2 ? hc / 2 - 1 : 0u); }());
    return pool;
}

struct HostPipe {
    std::mutex mtx;
    bool ready = false;
    int device = -1;    // the device the streams, events and buffers below belong to (the one current at their creation)
    hipStream_t s_in = nullptr, s_solve = nullptr, s_out = nullptr;
    static constexpr int NSLOT = 3;
    struct Slot {
        double *d_in = nullptr, *d_out = nullptr, *h_in = nullptr, *h_out = nullptr, *d_ws = nullptr;
        size_t in_bytes = 0, out_bytes = 0, ws_bytes = 0;
        hipEvent_t e_in = nullptr, e_solve = nullptr, e_out = nullptr;
    } slot[NSLOT];
    void release()
    {
        if (!ready) return;
        for (auto &s : slot) {
            (void)hipFree(s.d_in); (void)hipFree(s.d_out); (void)hipFree(s.d_ws); (void)hipHostFree(s.h_in); (void)hipHostFree(s.h_out);
            if (s.e_in) (void)hipEventDestroy(s.e_in);
            if (s.e_solve) (void)hipEventDestroy(s.e_solve);
            if (s.e_out) (void)hipEventDestroy(s.e_out);
            s = Slot();
        }
        if (s_in) (void)hipStreamDestroy(s_in);
        if (s_solve) (void)hipStreamDestroy(s_solve);
        if (s_out) (void)hipStreamDestroy(s_out);
        s_in = s_solve = s_out = nullptr;
        ready = false; device = -1;
    }
    void drain_all()
    {
        if (s_in) (void)hipStreamSynchronize(s_in);
        if (s_solve) (void)hipStreamSynchronize(s_solve);
        if (s_out) (void)hipStreamSynchronize(s_out);
    }
    ~HostPipe() { release(); }
};
HostPipe g_pipe;

// ---- caller buffers registered with frp_nmpc_host_register: pinned in place and mapped into the device's address space.  A batch whose
// arrays ALL lie in registered ranges needs no staging: a gather kernel reads the live part of a chunk's inputs straight from the
// caller's memory over PCIe into the chunk's device block (what the staging copy + hipMemcpyAsync did, without the host's memcpy, which
// set the pace: 16 host threads against a 0.85 ms solve), and the solver writes plans, flags and diagnostics in place.
struct HostRegistry {
    // refs: registrations of exactly this base (a second registration of the same pointer is counted, the pages are unpinned by the
    // last unregistration).  A registration that lies INSIDE a range somebody else registered is kept as an alias of that range
    // (nothing is pinned for it; it is unregistered by its own pointer like any other; the owner cannot leave while it has aliases).
    struct Range { char *base; size_t bytes; char *dev; int refs; };
    struct Alias { char *ptr; char *owner; };
    std::mutex mtx;
    std::vector<Range> ranges;
    std::vector<Alias> aliases;
    template <typename T>
    T *device_ptr(const T *p, size_t bytes)
    {
        const char *c = reinterpret_cast<const char *>(p);
        for (const Range &r : ranges)
            if (c >= r.base && c + bytes <= r.base + r.bytes) return reinterpret_cast<T *>(r.dev + (c - r.base));
        return nullptr;
    }
    Range *containing(const char *c, size_t bytes)
    {
        for (Range &r : ranges)
            if (c >= r.base && c + bytes <= r.base + r.bytes) return &r;
        return nullptr;
    }
};
HostRegistry g_reg;

// ---- two batches in flight (frp_nmpc_solve_batch_host_begin / _wait; registered buffers only): while batch k solves, the gather kernel of
// batch k + 1 reads its inputs over the host link.  The solver's workgroups are persistent and fill every CU's registers, so a gather block
// that arrives later finds no room until a solver workgroup retires: the pipelined solves leave `reserve` resident slots free (KernelArgs::
// slot_reserve) and the gather runs as a SMALL persistent grid of that many workgroups (64 suffice for the link: tools/ubench/zc_read).
struct AsyncPipe {
    std::mutex mtx;
    bool ready = false;
    int device = -1;
    hipStream_t s_gather = nullptr, s_solve = nullptr;
    static constexpr int NSLOT = 2;
    struct Slot {
        double *d_in = nullptr, *d_ws = nullptr;
        size_t in_bytes = 0, ws_bytes = 0;
        hipEvent_t e_in = nullptr, e_done = nullptr;
        bool busy = false;
    } slot[NSLOT];
    void release()
    {
        for (auto &s : slot) {
            if (s.busy && s.e_done) (void)hipEventSynchronize(s.e_done);
            (void)hipFree(s.d_in); (void)hipFree(s.d_ws);
            if (s.e_in) (void)hipEventDestroy(s.e_in);
            if (s.e_done) (void)hipEventDestroy(s.e_done);
            s = Slot();
        }
        if (s_gather) (void)hipStreamDestroy(s_gather);
        if (s_solve) (void)hipStreamDestroy(s_solve);
        s_gather = s_solve = nullptr; ready = false; device = -1;
    }
    ~AsyncPipe() { release(); }
};
AsyncPipe g_async;

// one chunk's device block [xinit | x0 | params (Md live rows of the caller's M) | nfaces | models] from the caller's (mapped) arrays
__global__ __launch_bounds__(256) void gather_inputs_kernel(size_t nb, size_t N, int M, int Md, const double *__restrict__ xinit, const double *__restrict__ x0,
                                                            const double *__restrict__ params, const int *__restrict__ nfaces, const int *__restrict__ models,
                                                            double *__restrict__ d_in)
{
    const size_t np_h = FRP_NPAR(M), np = FRP_NPAR(Md), n_x = nb * 9, n_z = nb * N * 17, n_p = nb * N * np, nf_d = nfaces ? (N + 1) / 2 : 0;
    const size_t total = n_x + n_z + n_p, stride = (size_t)gridDim.x * blockDim.x;
    const size_t head = 10 + 3 * (size_t)Md, boff = 10 + 3 * (size_t)M;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        double v;
        if (i < n_x) v = xinit[i];
        else if (i < n_x + n_z) v = x0[i - n_x];
        else {
            const size_t q = i - n_x - n_z, row = q / np, e = q - row * np;
            v = params[row * np_h + (e < head ? e : e - head + boff)];
        }
        d_in[i] = v;
    }
    int *d_nf = reinterpret_cast<int *>(d_in + total);
    if (nfaces)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nb * N; i += stride) d_nf[i] = nfaces[i];
    if (models) {
        int *d_md = reinterpret_cast<int *>(d_in + total + nb * nf_d);
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += stride) d_md[i] = models[i];
    }
}

int pipe_reserve(HostPipe::Slot &s, size_t in_bytes, size_t out_bytes, size_t ws_bytes)
{
    if (in_bytes > s.in_bytes) {
        (void)hipFree(s.d_in); (void)hipHostFree(s.h_in); s.d_in = s.h_in = nullptr; s.in_bytes = 0;
        FRP_HIP(hipMalloc(&s.d_in, in_bytes));
        FRP_HIP(hipHostMalloc(&s.h_in, in_bytes, hipHostMallocDefault));
        s.in_bytes = in_bytes;
    }
    if (out_bytes > s.out_bytes) {
        (void)hipFree(s.d_out); (void)hipHostFree(s.h_out); s.d_out = s.h_out = nullptr; s.out_bytes = 0;
        FRP_HIP(hipMalloc(&s.d_out, out_bytes));
        FRP_HIP(hipHostMalloc(&s.h_out, out_bytes, hipHostMallocDefault));
        s.out_bytes = out_bytes;
    }
    if (ws_bytes > s.ws_bytes) {
        (void)hipFree(s.d_ws); s.d_ws = nullptr; s.ws_bytes = 0;
        FRP_HIP(hipMalloc(&s.d_ws, ws_bytes));
        s.ws_bytes = ws_bytes;
    }
    return FRP_OK;
}
} // namespace

extern "C" {

int frp_nmpc_abi_version(void) { return FRP_NMPC_ABI_VERSION; }

int frp_nmpc_abi_check(int abi_version, size_t options_bytes, size_t batch_bytes, int info_stride)
{
    if (abi_version == FRP_NMPC_ABI_VERSION && options_bytes == sizeof(frp_nmpc_options) && batch_bytes == sizeof(frp_nmpc_batch) && info_stride == FRP_INFO_STRIDE)
        return FRP_OK;
    fprintf(stderr, "[frp_nmpc] the caller was built against another frp_nmpc.h: ABI version %d (library %d), frp_nmpc_options %zu B (%zu), "
                    "frp_nmpc_batch %zu B (%zu), info stride %d (%d)\n",
            abi_version, FRP_NMPC_ABI_VERSION, options_bytes, sizeof(frp_nmpc_options), batch_bytes, sizeof(frp_nmpc_batch), info_stride, FRP_INFO_STRIDE);
    return FRP_ERR_ARG;
}

const char *frp_nmpc_version(void) { return "frp_nmpc_amd 0.6 (gfx950, FP64 interior point: three / four wavefronts per problem, stage records in LDS)"; }

int frp_nmpc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void frp_nmpc_default_options(frp_nmpc_options *o)
{
    o->maxit = 200;
    o->tol_stat = 1e-4;
    o->tol_eq = 1e-4;
    o->tol_ineq = 1e-4;
    o->tol_comp = 1e-4;
    o->mu0 = 1.0;
    o->ftb = 0.99;
    o->hessian = 1;
    o->diverge_mu = 1e3;
    o->twist = 0;
}

size_t frp_nmpc_workspace_bytes(int B, int N, int MF) { return frp::ws_bytes(B, N, MF); }

int frp_nmpc_solve_batch(const frp_nmpc_batch *batch, const frp_nmpc_options *opt, void *workspace,
                         size_t workspace_bytes, void *stream)
{
    frp::KernelArgs a;
    if (!fill_args(batch, opt, workspace, workspace_bytes, &a)) return FRP_ERR_ARG;
    FRP_HIP(frp::launch_ipm(a, static_cast<hipStream_t>(stream)));
    return FRP_OK;
}

int frp_nmpc_time_solve(const frp_nmpc_batch *batch, const frp_nmpc_options *opt, void *workspace,
                        size_t workspace_bytes, void *stream, int reps, float *avg_ms)
{
    frp::KernelArgs a;
    if (!fill_args(batch, opt, workspace, workspace_bytes, &a) || reps <= 0 || !avg_ms) return FRP_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    struct Events { // destroyed on every exit path
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Events() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
    } ev;
    FRP_HIP(hipEventCreate(&ev.e0));
    FRP_HIP(hipEventCreate(&ev.e1));
    FRP_HIP(hipEventRecord(ev.e0, st));
    for (int r = 0; r < reps; r++) FRP_HIP(frp::launch_ipm(a, st));
    FRP_HIP(hipEventRecord(ev.e1, st));
    FRP_HIP(hipEventSynchronize(ev.e1));
    float ms = 0.f;
    FRP_HIP(hipEventElapsedTime(&ms, ev.e0, ev.e1));
    *avg_ms = ms / reps;
    return FRP_OK;
}

int frp_nmpc_kernel_timing_begin(int max_launches, int stride)
{
    if (max_launches <= 0 || stride <= 0) return FRP_ERR_ARG;
    return frp::kernel_timing_begin(max_launches, stride) == hipSuccess ? FRP_OK : FRP_ERR_ARG;
}

int frp_nmpc_host_register(void *ptr, size_t bytes)
{
    if (!ptr || !bytes) return FRP_ERR_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return FRP_ERR_NO_DEVICE;
    std::lock_guard<std::mutex> lock(g_reg.mtx);
    char *c = reinterpret_cast<char *>(ptr);
    if (HostRegistry::Range *r = g_reg.containing(c, bytes)) {
        if (r->base == c) r->refs++;                    // the same buffer once more: counted
        else g_reg.aliases.push_back({c, r->base});     // inside somebody's range: an alias, unregistered by its own pointer
        return FRP_OK;
    }
    // a range that OVERLAPS a registered one without lying inside it cannot be pinned a second time by the runtime: refuse it by name
    for (const HostRegistry::Range &r : g_reg.ranges)
        if (c < r.base + r.bytes && r.base < c + bytes) return FRP_ERR_ARG;
    if (hipHostRegister(ptr, bytes, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess) { (void)hipGetLastError(); return FRP_ERR_HIP; }
    void *dev = nullptr;
    if (hipHostGetDevicePointer(&dev, ptr, 0) != hipSuccess || !dev) { (void)hipHostUnregister(ptr); (void)hipGetLastError(); return FRP_ERR_HIP; }
    g_reg.ranges.push_back({c, bytes, reinterpret_cast<char *>(dev), 1});
    return FRP_OK;
}

int frp_nmpc_host_unregister(void *ptr)
{
    std::lock_guard<std::mutex> lock(g_reg.mtx);
    char *c = reinterpret_cast<char *>(ptr);
    for (size_t i = 0; i < g_reg.aliases.size(); i++)
        if (g_reg.aliases[i].ptr == c) { g_reg.aliases.erase(g_reg.aliases.begin() + (long)i); return FRP_OK; }
    for (size_t i = 0; i < g_reg.ranges.size(); i++)
        if (g_reg.ranges[i].base == c) {
            if (g_reg.ranges[i].refs > 1) { g_reg.ranges[i].refs--; return FRP_OK; }
            for (const HostRegistry::Alias &al : g_reg.aliases)
                if (al.owner == c) return FRP_ERR_ARG; // (sub-ranges registered through it are still in use)
            const hipError_t rc = hipHostUnregister(ptr);
            g_reg.ranges.erase(g_reg.ranges.begin() + (long)i);
            return rc == hipSuccess ? FRP_OK : FRP_ERR_HIP;
        }
    return FRP_ERR_ARG;
}

int frp_nmpc_host_registered(const void *ptr, size_t bytes)
{
    if (!ptr || !bytes) return 0;
    std::lock_guard<std::mutex> lock(g_reg.mtx);
    return g_reg.containing(reinterpret_cast<const char *>(ptr), bytes) ? 1 : 0;
}

int frp_nmpc_host_unregister_all(void)
{
    // (a caller that lost track -- e.g. arrays garbage-collected without frp_nmpc_host_unregister -- starts over: no stale range may
    // survive to be taken for a new allocation at the same address)
    std::lock_guard<std::mutex> pipe_lock(g_pipe.mtx); // no host batch is in flight on the ranges while they go
    std::lock_guard<std::mutex> lock(g_reg.mtx);
    int rc = FRP_OK;
    for (const HostRegistry::Range &r : g_reg.ranges)
        if (hipHostUnregister(r.base) != hipSuccess) { (void)hipGetLastError(); rc = FRP_ERR_HIP; }
    g_reg.ranges.clear(); g_reg.aliases.clear();
    return rc;
}

int frp_nmpc_set_q4_min_batch(int min_batch) { return frp::lds_q4_set_min_batch(min_batch); }

int frp_nmpc_kernel_timing_end(float *avg_ms, int *launches)
{
    if (!avg_ms || !launches) return FRP_ERR_ARG;
    const hipError_t rc = frp::kernel_timing_end(avg_ms, launches);
    return rc == hipSuccess ? FRP_OK : (rc == hipErrorInvalidValue ? FRP_ERR_ARG : FRP_ERR_HIP);
}

int frp_nmpc_solve_batch_host(const frp_nmpc_batch *h, const frp_nmpc_options *opt)
{
    // the same checks as fill_args, before anything is sized or copied from the caller's pointers
    if (!h || h->B <= 0 || h->N < 2 || h->N > 64 || h->M < 0 || h->MF < 0 || h->MF > h->M || h->MF > frp::FRP_MAX_FACES) return FRP_ERR_ARG;
    if (!h->xinit || !h->x0 || !h->params || !h->z || !h->exitflag || !h->iters) return FRP_ERR_ARG;
    if (h->model != FRP_MODEL_NORMAL && h->model != FRP_MODEL_FINAL) return FRP_ERR_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return FRP_ERR_NO_DEVICE;
    std::lock_guard<std::mutex> lock(g_pipe.mtx);
    // every chunk of this call runs on the kernel variants ONE launch of the whole batch would run on (the four-per-CU variants are
    // chosen by batch size and sum in another order): the plans do not depend on how the staging is cut (KernelArgs::variant_B --
    // an argument of the launch, not process state: concurrent frp_nmpc_solve_batch calls on other threads keep their own choice)
    // The pipeline (streams, events, device and pinned buffers) belongs to ONE device: the one current when it was built.  A call
    // made with another device current rebuilds it there (the per-call allocation it replaced worked on any device).
    int dev = 0;
    FRP_HIP(hipGetDevice(&dev));
    if (g_pipe.ready && g_pipe.device != dev) {
        const int prev = g_pipe.device;
        (void)hipSetDevice(prev);
        g_pipe.drain_all();
        g_pipe.release();
        FRP_HIP(hipSetDevice(dev));
    }
    // (declared before the guard below: on an error path the kernels still writing the caller's registered memory are drained UNDER the
    // registry lock, so nobody unregisters -- unpins -- those pages while they are being written)
    std::lock_guard<std::mutex> reg_lock(g_reg.mtx);
    struct DirtyOnError { // every early return below leaves the slots in use: they are drained before the error is returned
        int rc = FRP_ERR_HIP;
        ~DirtyOnError() { if (rc != FRP_OK && g_pipe.ready) { g_pipe.drain_all(); } }
    } guard;
    if (!g_pipe.ready) {
        g_pipe.device = dev;
        g_pipe.ready = true; // (from here on release() owns whatever was created; a failure half way releases it again)
        bool ok = hipStreamCreateWithFlags(&g_pipe.s_in, hipStreamNonBlocking) == hipSuccess &&
                  hipStreamCreateWithFlags(&g_pipe.s_solve, hipStreamNonBlocking) == hipSuccess &&
                  hipStreamCreateWithFlags(&g_pipe.s_out, hipStreamNonBlocking) == hipSuccess;
        for (auto &s : g_pipe.slot)
            ok = ok && hipEventCreateWithFlags(&s.e_in, hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&s.e_solve, hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&s.e_out, hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            fprintf(stderr, "[frp_nmpc] HIP error %s while creating the host-batch pipeline\n", hipGetErrorString(hipGetLastError()));
            g_pipe.release();
            return FRP_ERR_HIP;
        }
    }
    // With explicit face counts only the first MF corridor rows of a stage can be live (a larger count is a parameter error the
    // kernel reports either way): the staging copy packs the parameters to MF rows -- 26.4 -> 8.9 KB per problem over PCIe at
    // the reference's 30-row layout with 6-face corridors -- and the device solves the same problems in the compact layout.
    const bool compact = h->nfaces && h->MF < h->M;
    // every array of the batch inside ranges the caller registered (frp_nmpc_host_register): no staging, see HostRegistry
    const size_t Bz = (size_t)h->B, Tz = Bz * (size_t)h->N;
    const double *m_xinit = g_reg.device_ptr(h->xinit, Bz * 9 * 8), *m_x0 = g_reg.device_ptr(h->x0, Tz * 17 * 8);
    const double *m_params = g_reg.device_ptr(h->params, Tz * FRP_NPAR(h->M) * 8);
    const int *m_nf = h->nfaces ? g_reg.device_ptr(h->nfaces, Tz * 4) : nullptr, *m_md = h->model_per_problem ? g_reg.device_ptr(h->model_per_problem, Bz * 4) : nullptr;
    double *m_z = g_reg.device_ptr(h->z, Tz * 17 * 8), *m_info = h->info ? g_reg.device_ptr(h->info, Bz * FRP_INFO_STRIDE * 8) : nullptr;
    int *m_flag = g_reg.device_ptr(h->exitflag, Bz * 4), *m_it = g_reg.device_ptr(h->iters, Bz * 4);
    const bool mapped = m_xinit && m_x0 && m_params && (!h->nfaces || m_nf) && (!h->model_per_problem || m_md) && m_z && (!h->info || m_info) && m_flag && m_it;
    const int Md = compact ? h->MF : h->M; // corridor rows of the device-side layout
    const size_t N = h->N, np_h = FRP_NPAR(h->M), np = FRP_NPAR(Md);
    // per-problem doubles in a chunk's input block: xinit | x0 | params, then nfaces / models (ints, padded to doubles)
    const size_t in_d = 9 + N * 17 + N * np, nf_d = h->nfaces ? (N + 1) / 2 : 0, md_d = h->model_per_problem ? 1 : 0;
    const size_t out_d = N * 17 + (h->info ? FRP_INFO_STRIDE : 0) + 1; // z | info | (exitflag, iterations)
    // Chunks whose copies and solves overlap.  A small first chunk gets the GPU going while the rest is still being staged, the
    // later ones are large because a launch of fewer problems than two rounds of resident workgroups is inefficient: B / 16,
    // B / 4, then the rest in pieces of at most 4096 (measured on the 4096-problem batch, ms end to end: one chunk 2.54, three
    // equal ones 2.56, 256 + 1024 + 2816 2.09).
    size_t chunk = (size_t)h->B;
    std::vector<size_t> cb{0};
    if (h->B >= 4096 && !mapped) { // (registered buffers: one chunk, see the gather below)
        cb.push_back((size_t)h->B / 16);
        cb.push_back(cb.back() + (size_t)h->B / 4);
        const size_t rest = (size_t)h->B - cb.back(), nc = (rest + 4095) / 4096;
        chunk = (rest + nc - 1) / nc;
    }
    if (const char *e = getenv("FRP_HOST_CHUNK")) { const long v = atol(e); if (v > 0) chunk = (size_t)v; } // tuning knob
    // chunk c covers the problems [cb[c], cb[c + 1]); FRP_HOST_SPLIT="n0,n1,..." (tuning knob) gives explicit sizes, the rest uniform
    if (const char *e = getenv("FRP_HOST_SPLIT")) {
        cb.assign(1, 0);
        for (const char *q = e; *q && cb.back() < (size_t)h->B;) {
            char *end = nullptr; const long v = strtol(q, &end, 10);
            if (end == q || v <= 0) break;
            cb.push_back(std::min((size_t)h->B, cb.back() + (size_t)v));
            q = (*end == ',') ? end + 1 : end;
        }
    }
    while (cb.back() < (size_t)h->B) cb.push_back(std::min((size_t)h->B, cb.back() + chunk));
    const size_t nchunk = cb.size() - 1;
    for (size_t c = 0; c < nchunk; c++) chunk = std::max(chunk, cb[c + 1] - cb[c]); // (buffers are sized for the largest)
    for (auto &s : g_pipe.slot) {
        const int rc = pipe_reserve(s, chunk * (in_d + nf_d + md_d) * sizeof(double), chunk * out_d * sizeof(double), frp::ws_bytes((int)chunk, h->N, h->MF));
        if (rc != FRP_OK) return rc;
    }
    auto drain = [&](size_t c) -> int { // chunk c's results: wait for its copy-out, unpack from the pinned block
        HostPipe::Slot &s = g_pipe.slot[c % HostPipe::NSLOT];
        FRP_HIP(hipEventSynchronize(s.e_out));
        if (mapped) return FRP_OK; // (the solver wrote the caller's arrays in place)
        const size_t b0 = cb[c], nb = cb[c + 1] - cb[c];
        const double *o = s.h_out;
        copy_pool().copy(h->z + b0 * N * 17, o, nb * N * 17 * sizeof(double)); o += nb * N * 17;
        if (h->info) { std::memcpy(h->info + b0 * FRP_INFO_STRIDE, o, nb * FRP_INFO_STRIDE * sizeof(double)); o += nb * FRP_INFO_STRIDE; }
        const int *fi = reinterpret_cast<const int *>(o);
        std::memcpy(h->exitflag + b0, fi, nb * sizeof(int));
        std::memcpy(h->iters + b0, fi + nb, nb * sizeof(int));
        return FRP_OK;
    };
    for (size_t c = 0; c < nchunk; c++) {
        HostPipe::Slot &s = g_pipe.slot[c % HostPipe::NSLOT];
        if (c >= HostPipe::NSLOT) { const int rc = drain(c - HostPipe::NSLOT); if (rc != FRP_OK) return rc; } // the slot's previous chunk has left it
        const size_t b0 = cb[c], nb = cb[c + 1] - cb[c];
        // (measured, 4096 problems of configs[2] in the reference's 30-row layout, host link ~25 GB/s: the gather KERNEL in one chunk 2.03 ms,
        // in chunks 2.3-2.7 -- a resident gather block keeps the solver's persistent workgroups of the previous chunk off its CU --;
        // the copy engines ("dma": two strided copies for the parameter rows) 2.07-2.33 whatever the split; staging from pageable memory
        // 2.4-3.6.  The bound on that link is (28 MB in + 5.6 MB out) / 25 GB/s + the 0.85 ms solve = 2.0 ms without overlap.)
        static const bool gather_dma = [] { const char *e = getenv("FRP_HOST_GATHER"); return e && e[0] == 'd'; }(); // (tuning knob: "dma")
        if (mapped && gather_dma) {
            // the copy engines read the caller's pinned memory: contiguous arrays as they are, the parameter rows as two strided copies
            // (the ten leading parameters + the live A rows, then the live b rows) -- no CU is taken from the solves of the previous chunk
            double *d_x0 = s.d_in + nb * 9, *d_par = d_x0 + nb * N * 17;
            FRP_HIP(hipMemcpyAsync(s.d_in, h->xinit + b0 * 9, nb * 9 * sizeof(double), hipMemcpyHostToDevice, g_pipe.s_in));
            FRP_HIP(hipMemcpyAsync(d_x0, h->x0 + b0 * N * 17, nb * N * 17 * sizeof(double), hipMemcpyHostToDevice, g_pipe.s_in));
            const size_t headb = (10 + 3 * (size_t)Md) * 8;
            FRP_HIP(hipMemcpy2DAsync(d_par, np * 8, h->params + b0 * N * np_h, np_h * 8, headb, nb * N, hipMemcpyHostToDevice, g_pipe.s_in));
            if (Md > 0)
                FRP_HIP(hipMemcpy2DAsync(reinterpret_cast<char *>(d_par) + headb, np * 8, h->params + b0 * N * np_h + 10 + 3 * (size_t)h->M, np_h * 8, (size_t)Md * 8, nb * N,
                                         hipMemcpyHostToDevice, g_pipe.s_in));
            int *d_nf = reinterpret_cast<int *>(d_par + nb * N * np);
            if (h->nfaces) FRP_HIP(hipMemcpyAsync(d_nf, h->nfaces + b0 * N, nb * N * sizeof(int), hipMemcpyHostToDevice, g_pipe.s_in));
            if (h->model_per_problem)
                FRP_HIP(hipMemcpyAsync(reinterpret_cast<double *>(d_nf) + nb * nf_d, h->model_per_problem + b0, nb * sizeof(int), hipMemcpyHostToDevice, g_pipe.s_in));
        } else if (mapped) {
            const unsigned blocks = (unsigned)std::min<size_t>(2048, (nb * in_d + 255) / 256);
            hipLaunchKernelGGL(gather_inputs_kernel, dim3(blocks), dim3(256), 0, g_pipe.s_in, nb, N, h->M, Md, m_xinit + b0 * 9, m_x0 + b0 * N * 17,
                               m_params + b0 * N * np_h, m_nf ? m_nf + b0 * N : nullptr, m_md ? m_md + b0 : nullptr, s.d_in);
            FRP_HIP(hipGetLastError());
        } else {
        double *hi = s.h_in;
        std::memcpy(hi, h->xinit + b0 * 9, nb * 9 * sizeof(double)); double *h_x0 = hi + nb * 9;
        copy_pool().copy(h_x0, h->x0 + b0 * N * 17, nb * N * 17 * sizeof(double)); double *h_par = h_x0 + nb * N * 17;
        if (compact) copy_pool().pack_rows(h_par, h->params + b0 * N * np_h, nb * N, h->M, Md);
        else copy_pool().copy(h_par, h->params + b0 * N * np, nb * N * np * sizeof(double));
        int *h_nf = reinterpret_cast<int *>(h_par + nb * N * np);
        if (h->nfaces) std::memcpy(h_nf, h->nfaces + b0 * N, nb * N * sizeof(int));
        int *h_md = reinterpret_cast<int *>(reinterpret_cast<double *>(h_nf) + nb * nf_d);
        if (h->model_per_problem) std::memcpy(h_md, h->model_per_problem + b0, nb * sizeof(int));
        const size_t in_bytes = (nb * (in_d + nf_d) + (md_d ? (nb + 1) / 2 : 0)) * sizeof(double);
        FRP_HIP(hipMemcpyAsync(s.d_in, s.h_in, in_bytes, hipMemcpyHostToDevice, g_pipe.s_in));
        }
        FRP_HIP(hipEventRecord(s.e_in, g_pipe.s_in));
        frp_nmpc_batch d = *h;
        d.B = (int)nb; d.M = Md;
        d.xinit = s.d_in; d.x0 = s.d_in + nb * 9; d.params = s.d_in + nb * 9 + nb * N * 17;
        const double *d_tail = s.d_in + nb * in_d;
        d.nfaces = h->nfaces ? reinterpret_cast<const int *>(d_tail) : nullptr;
        d.model_per_problem = h->model_per_problem ? reinterpret_cast<const int *>(d_tail + nb * nf_d) : nullptr;
        d.order_hint = nullptr; // (a queue-order hint is not worth a copy here: the chunks are at most two rounds of resident workgroups)
        d.z = s.d_out; d.info = h->info ? s.d_out + nb * N * 17 : nullptr;
        d.exitflag = reinterpret_cast<int *>(s.d_out + nb * N * 17 + (h->info ? nb * FRP_INFO_STRIDE : 0)); d.iters = d.exitflag + nb;
        if (mapped) { d.z = m_z + b0 * N * 17; d.info = m_info ? m_info + b0 * FRP_INFO_STRIDE : nullptr; d.exitflag = m_flag + b0; d.iters = m_it + b0; }
        FRP_HIP(hipStreamWaitEvent(g_pipe.s_solve, s.e_in, 0));
        {
            frp::KernelArgs ka;
            if (!fill_args(&d, opt, s.d_ws, s.ws_bytes, &ka)) return FRP_ERR_ARG;
            ka.variant_B = h->B;
            FRP_HIP(frp::launch_ipm(ka, g_pipe.s_solve));
        }
        FRP_HIP(hipEventRecord(s.e_solve, g_pipe.s_solve));
        if (mapped) { FRP_HIP(hipEventRecord(s.e_out, g_pipe.s_solve)); continue; }
        FRP_HIP(hipStreamWaitEvent(g_pipe.s_out, s.e_solve, 0));
        const size_t out_bytes = (nb * N * 17 + (h->info ? nb * FRP_INFO_STRIDE : 0) + nb) * sizeof(double);
        FRP_HIP(hipMemcpyAsync(s.h_out, s.d_out, out_bytes, hipMemcpyDeviceToHost, g_pipe.s_out));
        FRP_HIP(hipEventRecord(s.e_out, g_pipe.s_out));
    }
    for (size_t c = nchunk > HostPipe::NSLOT ? nchunk - HostPipe::NSLOT : 0; c < nchunk; c++) {
        const int rc = drain(c);
        if (rc != FRP_OK) return rc;
    }
    guard.rc = FRP_OK;
    return FRP_OK;
}

int frp_nmpc_solve_batch_host_begin(const frp_nmpc_batch *h, const frp_nmpc_options *opt, int *ticket)
{
    if (!ticket) return FRP_ERR_ARG;
    *ticket = -1;
    if (!h || h->B <= 0 || h->N < 2 || h->N > 64 || h->M < 0 || h->MF < 0 || h->MF > h->M || h->MF > frp::FRP_MAX_FACES) return FRP_ERR_ARG;
    if (!h->xinit || !h->x0 || !h->params || !h->z || !h->exitflag || !h->iters) return FRP_ERR_ARG;
    if (h->model != FRP_MODEL_NORMAL && h->model != FRP_MODEL_FINAL) return FRP_ERR_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return FRP_ERR_NO_DEVICE;
    std::lock_guard<std::mutex> lock(g_async.mtx);
    int dev = 0;
    FRP_HIP(hipGetDevice(&dev));
    if (g_async.ready && g_async.device != dev) { // (the pipeline belongs to the device it was built on; a call from another one rebuilds it there)
        const int prev = g_async.device;
        (void)hipSetDevice(prev);
        g_async.release();
        FRP_HIP(hipSetDevice(dev));
    }
    if (!g_async.ready) {
        g_async.device = dev; g_async.ready = true;
        bool ok = hipStreamCreateWithFlags(&g_async.s_gather, hipStreamNonBlocking) == hipSuccess &&
                  hipStreamCreateWithFlags(&g_async.s_solve, hipStreamNonBlocking) == hipSuccess;
        for (auto &s : g_async.slot)
            ok = ok && hipEventCreateWithFlags(&s.e_in, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&s.e_done, hipEventDisableTiming) == hipSuccess;
        if (!ok) { (void)hipGetLastError(); g_async.release(); return FRP_ERR_HIP; }
    }
    int si = -1;
    for (int i = 0; i < AsyncPipe::NSLOT; i++)
        if (!g_async.slot[i].busy) { si = i; break; }
    if (si < 0) return FRP_ERR_ARG; // (FRP_NMPC_HOST_INFLIGHT batches are out: wait for one)
    AsyncPipe::Slot &s = g_async.slot[si];
    std::lock_guard<std::mutex> reg_lock(g_reg.mtx);
    const size_t Bz = (size_t)h->B, N = (size_t)h->N, Tz = Bz * N;
    const double *m_xinit = g_reg.device_ptr(h->xinit, Bz * 9 * 8), *m_x0 = g_reg.device_ptr(h->x0, Tz * 17 * 8);
    const double *m_params = g_reg.device_ptr(h->params, Tz * FRP_NPAR(h->M) * 8);
    const int *m_nf = h->nfaces ? g_reg.device_ptr(h->nfaces, Tz * 4) : nullptr, *m_md = h->model_per_problem ? g_reg.device_ptr(h->model_per_problem, Bz * 4) : nullptr;
    double *m_z = g_reg.device_ptr(h->z, Tz * 17 * 8), *m_info = h->info ? g_reg.device_ptr(h->info, Bz * FRP_INFO_STRIDE * 8) : nullptr;
    int *m_flag = g_reg.device_ptr(h->exitflag, Bz * 4), *m_it = g_reg.device_ptr(h->iters, Bz * 4);
    if (!(m_xinit && m_x0 && m_params && (!h->nfaces || m_nf) && (!h->model_per_problem || m_md) && m_z && (!h->info || m_info) && m_flag && m_it))
        return FRP_ERR_ARG; // every array of a pipelined batch is registered (frp_nmpc_host_register): nothing is staged on this path
    const bool compact = h->nfaces && h->MF < h->M;
    const int Md = compact ? h->MF : h->M;
    const size_t np = FRP_NPAR(Md), in_d = 9 + N * 17 + N * np, nf_d = h->nfaces ? (N + 1) / 2 : 0, md_d = h->model_per_problem ? 1 : 0;
    const size_t in_bytes = Bz * (in_d + nf_d + md_d) * sizeof(double), ws_bytes = frp::ws_bytes(h->B, h->N, h->MF);
    if (in_bytes > s.in_bytes) { (void)hipFree(s.d_in); s.d_in = nullptr; s.in_bytes = 0; FRP_HIP(hipMalloc(&s.d_in, in_bytes)); s.in_bytes = in_bytes; }
    if (ws_bytes > s.ws_bytes) { (void)hipFree(s.d_ws); s.d_ws = nullptr; s.ws_bytes = 0; FRP_HIP(hipMalloc(&s.d_ws, ws_bytes)); s.ws_bytes = ws_bytes; }
    static const int gather_wgs = [] { const char *e = getenv("FRP_HOST_GATHER_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 64; }(); // (tuning knob)
    hipLaunchKernelGGL(gather_inputs_kernel, dim3((unsigned)gather_wgs), dim3(256), 0, g_async.s_gather, Bz, N, h->M, Md, m_xinit, m_x0, m_params, m_nf, m_md, s.d_in);
    FRP_HIP(hipGetLastError());
    FRP_HIP(hipEventRecord(s.e_in, g_async.s_gather));
    frp_nmpc_batch d = *h;
    d.M = Md;
    d.xinit = s.d_in; d.x0 = s.d_in + Bz * 9; d.params = s.d_in + Bz * 9 + Tz * 17;
    const double *d_tail = s.d_in + Bz * in_d;
    d.nfaces = h->nfaces ? reinterpret_cast<const int *>(d_tail) : nullptr;
    d.model_per_problem = h->model_per_problem ? reinterpret_cast<const int *>(d_tail + Bz * nf_d) : nullptr;
    d.order_hint = nullptr;
    d.z = m_z; d.info = m_info; d.exitflag = m_flag; d.iters = m_it; // (the solver writes the caller's arrays in place)
    frp::KernelArgs ka;
    if (!fill_args(&d, opt, s.d_ws, s.ws_bytes, &ka)) return FRP_ERR_ARG;
    ka.slot_reserve = gather_wgs;
    FRP_HIP(hipStreamWaitEvent(g_async.s_solve, s.e_in, 0));
    FRP_HIP(frp::launch_ipm(ka, g_async.s_solve));
    FRP_HIP(hipEventRecord(s.e_done, g_async.s_solve));
    s.busy = true;
    *ticket = si;
    return FRP_OK;
}

int frp_nmpc_solve_batch_host_wait(int ticket)
{
    if (ticket < 0 || ticket >= AsyncPipe::NSLOT) return FRP_ERR_ARG;
    hipEvent_t ev;
    {
        std::lock_guard<std::mutex> lock(g_async.mtx);
        if (!g_async.ready || !g_async.slot[ticket].busy) return FRP_ERR_ARG;
        ev = g_async.slot[ticket].e_done;
    }
    const hipError_t rc = hipEventSynchronize(ev); // (outside the lock: another thread may begin the next batch meanwhile)
    std::lock_guard<std::mutex> lock(g_async.mtx);
    g_async.slot[ticket].busy = false;
    if (rc != hipSuccess) { fprintf(stderr, "[frp_nmpc] HIP error %s while waiting for a pipelined host batch\n", hipGetErrorString(rc)); return FRP_ERR_HIP; }
    return FRP_OK;
}

int frp_nmpc_stage_eval(int B, int N, int M, int model, const double *z, const double *params, double *f,
                        double *grad_f, double *c, double *jac_c, double *h, void *stream)
{
    if (B <= 0 || N < 2 || M < 0 || !z || !params) return FRP_ERR_ARG;
    FRP_HIP(frp::launch_stage_eval(B, N, M, model, z, params, f, grad_f, c, jac_c, h, static_cast<hipStream_t>(stream)));
    return FRP_OK;
}

int frp_nmpc_stage_eval_host(int B, int N, int M, int model, const double *z, const double *params, double *f,
                             double *grad_f, double *c, double *jac_c, double *h)
{
    if (B <= 0 || N < 2 || M < 0 || !z || !params) return FRP_ERR_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return FRP_ERR_NO_DEVICE;
    const size_t T = (size_t)B * N, np = FRP_NPAR(M);
    DevMem dz, dp, df, dg, dc, dJ, dh;
    FRP_HIP(dz.alloc(T * 17 * sizeof(double)));
    FRP_HIP(dp.alloc(T * np * sizeof(double)));
    FRP_HIP(hipMemcpy(dz.p, z, T * 17 * sizeof(double), hipMemcpyHostToDevice));
    FRP_HIP(hipMemcpy(dp.p, params, T * np * sizeof(double), hipMemcpyHostToDevice));
    if (f) FRP_HIP(df.alloc(T * sizeof(double)));
    if (grad_f) FRP_HIP(dg.alloc(T * 17 * sizeof(double)));
    if (c) FRP_HIP(dc.alloc(T * 13 * sizeof(double)));
    if (jac_c) FRP_HIP(dJ.alloc(T * 221 * sizeof(double)));
    if (h && M > 0) FRP_HIP(dh.alloc(T * M * sizeof(double)));
    const int rc = frp_nmpc_stage_eval(B, N, M, model, dz.as<double>(), dp.as<double>(), df.as<double>(), dg.as<double>(),
                                       dc.as<double>(), dJ.as<double>(), dh.as<double>(), nullptr);
    if (rc != FRP_OK) return rc;
    FRP_HIP(hipDeviceSynchronize());
    if (f) FRP_HIP(hipMemcpy(f, df.p, T * sizeof(double), hipMemcpyDeviceToHost));
    if (grad_f) FRP_HIP(hipMemcpy(grad_f, dg.p, T * 17 * sizeof(double), hipMemcpyDeviceToHost));
    if (c) FRP_HIP(hipMemcpy(c, dc.p, T * 13 * sizeof(double), hipMemcpyDeviceToHost));
    if (jac_c) FRP_HIP(hipMemcpy(jac_c, dJ.p, T * 221 * sizeof(double), hipMemcpyDeviceToHost));
    if (dh.p) FRP_HIP(hipMemcpy(h, dh.p, T * M * sizeof(double), hipMemcpyDeviceToHost));
    return FRP_OK;
}

int FORCESNLPsolver_normal_solve(frp_forces_params *params, frp_forces_output *output, frp_forces_info *info,
                                 FILE *fs, frp_forces_extfunc evalextfunctions)
{
    return forces_solve(FRP_MODEL_NORMAL, params, output, info, fs, evalextfunctions);
}

int FORCESNLPsolver_final_solve(frp_forces_params *params, frp_forces_output *output, frp_forces_info *info,
                                FILE *fs, frp_forces_extfunc evalextfunctions)
{
    return forces_solve(FRP_MODEL_FINAL, params, output, info, fs, evalextfunctions);
}

} // extern "C"
