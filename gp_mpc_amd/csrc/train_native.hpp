// Native driver of hyper-parameter training (a8): box-constrained minimisation of the device NLL with its analytic
// gradient, and the RCCL exchange of the restart shard.  Included by gpmpc_api.hip.
//
// The reference minimises with scipy SLSQP + finite differences (train_gp_numpy, optimize.py:466-467) or IPOPT on
// CasADi's AD (train_gp, :236-246); both only see a smooth box-constrained problem in d + 2 (+ mean parameters)
// variables.  Here: projected L-BFGS.  Which variables are searched in log space is the caller's choice (BoxProblem::logv);
// gpmpc_train_multistart runs TWO stages per restart: first with the length scales and sf in log space (their boxes span
// many decades) and the noise sn LINEAR, as in the reference's optimisers -- the NLL sees sn^2, so its derivative with
// respect to log sn vanishes at the reference's start sn = 1e-5 and a log-space search never leaves it --, then, if
// iterations are left, a polish from that end point with sn in log space as well; the better end point is kept.  Every
// step is a projected Armijo backtracking search that only ever accepts a decrease; the L-BFGS memory is dropped whenever
// the set of free (not bound-pinned) variables changes; a point where K is not positive definite even after the
// reference's jitter counts as +infinity, exactly like a failed restart in the reference (optimize.py:349-350 would raise).
#pragma once
#include <dlfcn.h>

#include <cmath>
#include <functional>
#include <limits>
#include <string>
#include <vector>

namespace gpmpc {

struct BoxProblem {
    int n = 0;
    std::vector<double> lb, ub;          // in the ORIGINAL variables
    std::vector<char> logv;              // optimise log(theta_k)?
    // f(theta, grad) -> value; returns false if the point is unusable (value treated as +inf)
    std::function<bool(const double*, double*, double*)> eval;   // (grad may be NULL: value only)
    // optional: the gradient at the point `eval` saw LAST (value only), cheaper than a second full evaluation; false if
    // it cannot be had (the search then evaluates the point again, with a gradient)
    std::function<bool(const double*, double*)> grad_last;
};

struct BoxResult {
    std::vector<double> theta;
    double f = std::numeric_limits<double>::infinity();
    int iters = 0, evals = 0;
    bool ok = false;
};

inline BoxResult minimize_box_lbfgs(const BoxProblem& P, const double* theta0, int max_iter, double tol) {
    const int n = P.n, M = 8;
    const double inf = std::numeric_limits<double>::infinity();
    BoxResult R;
    std::vector<double> lo(n), hi(n), x(n), g(n), th(n), gt(n);
    for (int k = 0; k < n; ++k) {
        lo[k] = P.logv[k] ? std::log(P.lb[k]) : P.lb[k];
        hi[k] = P.logv[k] ? std::log(P.ub[k]) : P.ub[k];
        double t0 = theta0[k];
        if (t0 < P.lb[k]) t0 = P.lb[k];
        if (t0 > P.ub[k]) t0 = P.ub[k];
        x[k] = P.logv[k] ? std::log(t0) : t0;
    }
    auto fun = [&](const std::vector<double>& xx, std::vector<double>& gx) -> double {
        for (int k = 0; k < n; ++k) th[k] = P.logv[k] ? std::exp(xx[k]) : xx[k];
        double f = inf;
        ++R.evals;
        if (!P.eval(th.data(), &f, gt.data()) || !(f == f)) return inf;
        for (int k = 0; k < n; ++k) {
            gx[k] = P.logv[k] ? gt[k] * th[k] : gt[k];          // d f / d log(theta) = theta d f / d theta
            if (!(gx[k] == gx[k])) return inf;
        }
        return f;
    };
    // a trial point of the line search: the value alone; its gradient only once the step is accepted
    auto fun_value = [&](const std::vector<double>& xx) -> double {
        for (int k = 0; k < n; ++k) th[k] = P.logv[k] ? std::exp(xx[k]) : xx[k];
        double f = inf;
        ++R.evals;
        if (!P.eval(th.data(), &f, nullptr) || !(f == f)) return inf;
        return f;
    };
    auto grad_of_last = [&](const std::vector<double>& xx, std::vector<double>& gx) -> bool {
        for (int k = 0; k < n; ++k) th[k] = P.logv[k] ? std::exp(xx[k]) : xx[k];
        if (!P.grad_last(th.data(), gt.data())) return false;
        for (int k = 0; k < n; ++k) {
            gx[k] = P.logv[k] ? gt[k] * th[k] : gt[k];
            if (!(gx[k] == gx[k])) return false;
        }
        return true;
    };
    double f = fun(x, g);
    R.theta.assign(n, 0.0);
    if (f == inf) {
        for (int k = 0; k < n; ++k) R.theta[k] = P.logv[k] ? std::exp(x[k]) : x[k];
        return R;                                               // unusable start: this restart has failed
    }
    std::vector<std::vector<double>> Sv, Yv;
    std::vector<double> rho;
    std::vector<double> dir(n), xn(n), gn(n), q(n), al(M);
    std::vector<char> fr(n), fr_prev;
    for (int it = 0; it < max_iter; ++it) {
        R.iters = it + 1;
        // free variables: not pinned at a bound with the gradient pushing outward
        double pgn = 0.0;
        for (int k = 0; k < n; ++k) {
            const bool at_lo = x[k] <= lo[k] && g[k] > 0.0, at_hi = x[k] >= hi[k] && g[k] < 0.0;
            fr[k] = !(at_lo || at_hi);
            if (fr[k]) pgn = std::max(pgn, std::fabs(g[k]));
        }
        if (pgn <= tol * std::max(1.0, std::fabs(f))) break;
        // the curvature pairs describe the problem restricted to the free set they were collected on
        if (!fr_prev.empty() && fr_prev != fr) { Sv.clear(); Yv.clear(); rho.clear(); }
        fr_prev = fr;
        // two-loop recursion on the free components
        for (int k = 0; k < n; ++k) q[k] = fr[k] ? g[k] : 0.0;
        const int m = (int)Sv.size();
        for (int i = m - 1; i >= 0; --i) {
            double a = 0.0;
            for (int k = 0; k < n; ++k) if (fr[k]) a += Sv[i][k] * q[k];
            a *= rho[i];
            al[i] = a;
            for (int k = 0; k < n; ++k) if (fr[k]) q[k] -= a * Yv[i][k];
        }
        double gamma = 1.0;
        if (m > 0) {
            double sy = 0.0, yy = 0.0;
            for (int k = 0; k < n; ++k) { sy += Sv[m - 1][k] * Yv[m - 1][k]; yy += Yv[m - 1][k] * Yv[m - 1][k]; }
            if (yy > 0.0) gamma = sy / yy;
        }
        for (int k = 0; k < n; ++k) q[k] *= gamma;
        for (int i = 0; i < m; ++i) {
            double b = 0.0;
            for (int k = 0; k < n; ++k) if (fr[k]) b += Yv[i][k] * q[k];
            b *= rho[i];
            for (int k = 0; k < n; ++k) if (fr[k]) q[k] += (al[i] - b) * Sv[i][k];
        }
        double slope = 0.0;
        for (int k = 0; k < n; ++k) { dir[k] = fr[k] ? -q[k] : 0.0; slope += dir[k] * g[k]; }
        if (!(slope < 0.0)) {                                   // not a descent direction: steepest descent on the free set
            slope = 0.0;
            for (int k = 0; k < n; ++k) { dir[k] = fr[k] ? -g[k] : 0.0; slope += dir[k] * g[k]; }
            Sv.clear(); Yv.clear(); rho.clear();
        }
        // first step of a run: a unit step in log space is a factor e -- scale it down to a gentle move
        double t = 1.0;
        if (Sv.empty()) {
            double dn = 0.0;
            for (int k = 0; k < n; ++k) dn = std::max(dn, std::fabs(dir[k]));
            if (dn > 0.0) t = std::min(1.0, 1.0 / dn);
        }
        double fn = inf;
        bool moved = false;
        for (int ls = 0; ls < 40; ++ls, t *= 0.5) {
            double dec = 0.0;
            bool any = false;
            for (int k = 0; k < n; ++k) {
                double v = x[k] + t * dir[k];
                if (v < lo[k]) v = lo[k];
                if (v > hi[k]) v = hi[k];
                xn[k] = v;
                dec += g[k] * (v - x[k]);
                any |= v != x[k];
            }
            if (!any) break;
            if (!(dec < 0.0)) continue;                         // the clipped step is not a descent step: shorten it
            fn = P.grad_last ? fun_value(xn) : fun(xn, gn);
            if (fn <= f + 1e-4 * dec) {                         // dec < 0: an accepted point is strictly better
                // (value-only trial: now its gradient -- from the factors of that evaluation, or by evaluating again)
                if (P.grad_last && !grad_of_last(xn, gn)) fn = fun(xn, gn);
                moved = fn <= f + 1e-4 * dec;
                break;
            }
        }
        if (!moved) break;                                      // no progress along the projected path
        std::vector<double> s(n), y(n);
        double sy = 0.0;
        for (int k = 0; k < n; ++k) { s[k] = xn[k] - x[k]; y[k] = gn[k] - g[k]; sy += s[k] * y[k]; }
        const double fdec = f - fn;
        x = xn; g = gn; f = fn;
        if (sy > 1e-12) {
            if ((int)Sv.size() == M) { Sv.erase(Sv.begin()); Yv.erase(Yv.begin()); rho.erase(rho.begin()); }
            Sv.push_back(s); Yv.push_back(y); rho.push_back(1.0 / sy);
        }
        if (fdec <= tol * std::max(1.0, std::fabs(f)) * 1e-3) break;   // objective has stalled at the 1e-3 tol level
    }
    for (int k = 0; k < n; ++k) R.theta[k] = P.logv[k] ? std::exp(x[k]) : x[k];
    R.f = f;
    R.ok = true;
    return R;
}

// ---- RCCL, bound at run time (no link-time dependency: the library loads on boxes without librccl) -----------------
struct RcclId { char internal[128]; };                          // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128)
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(RcclId*) = nullptr;
    int (*CommInitRank)(void**, int, RcclId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*GetVersion)(int*) = nullptr;
    std::string path;                                             // file the symbols come from (dladdr)
    std::string load_error;                                       // dlerror() text of the first failed dlopen / what is missing
    bool ok() const { return lib && GetUniqueId && CommInitRank && CommDestroy && AllGather; }
};
inline RcclApi& rccl_api() {
    static RcclApi api = [] {
        RcclApi a;
        // RCCL must belong to the ROCm tree whose HIP / HSA runtime is live in this process: a process that imported
        // PyTorch first runs on the copies bundled in torch/lib, one that loaded this library first on /opt/rocm, and an
        // RCCL from the other tree opens a second, uninitialised HSA runtime ("no ROCm-capable device is detected").
        // So: the librccl next to the libamdhip64 that hipGetDeviceCount resolves to, with local symbol scope.
        // (r04: binding whatever RCCL image the process already maps -- RTLD_NOLOAD -- was tried and is wrong: a process that
        //  imports PyTorch AFTER this library maps torch's RCCL next to torch's own HIP runtime while this library's
        //  kernels run on /opt/rocm's; ncclCommInitRank then fails as described above.  The rule stays: next to MY runtime.
        //  gpmpc_runtime_info reports which files that resolved to.)
        std::vector<std::string> names;
        Dl_info di;
        if (dladdr((void*)&hipGetDeviceCount, &di) && di.dli_fname) {
            std::string dir(di.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash);
                names.push_back(dir + "/librccl.so");
                names.push_back(dir + "/librccl.so.1");
            }
        }
        names.push_back("librccl.so.1");
        names.push_back("librccl.so");
        for (const std::string& name : names) {
            a.lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (a.lib) break;
            const char* e = dlerror();                            // (reading it clears it: keep the first text)
            if (a.load_error.empty() && e) a.load_error = e;
        }
        if (a.lib) {
            a.GetUniqueId = (int (*)(RcclId*))dlsym(a.lib, "ncclGetUniqueId");
            a.CommInitRank = (int (*)(void**, int, RcclId, int))dlsym(a.lib, "ncclCommInitRank");
            a.CommDestroy = (int (*)(void*))dlsym(a.lib, "ncclCommDestroy");
            a.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(a.lib, "ncclAllGather");
            a.GetErrorString = (const char* (*)(int))dlsym(a.lib, "ncclGetErrorString");
            a.CommCount = (int (*)(void*, int*))dlsym(a.lib, "ncclCommCount");
            a.GetVersion = (int (*)(int*))dlsym(a.lib, "ncclGetVersion");
            Dl_info dr;
            if (a.AllGather && dladdr((void*)a.AllGather, &dr) && dr.dli_fname) a.path = dr.dli_fname;
            a.load_error = a.ok() ? "" : "librccl is loaded but lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
        } else if (a.load_error.empty()) {
            a.load_error = "no librccl.so found";
        }
        return a;
    }();
    return api;
}
constexpr int RCCL_FLOAT64 = 8;                                  // ncclFloat64 (rccl.h)

}  // namespace gpmpc
