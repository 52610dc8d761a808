/* gpmpc.h -- C ABI of libgpmpc_hip.so: the MI355X (gfx950) GP-regression inner loop of GP-MPC.
 *
 * The reference (helgeanl/GP-MPC) is pure Python and has no FFI layer; its de-facto operator API
 * for this path is the `GP` object as `mpc_class.MPC` uses it.  Each entry
 * point below names the reference code it replaces (file:line into /root/reference/gp_mpc/).
 * `INTEGRATION.md` shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C symbols, no exceptions cross the boundary; every function returns an int status
 *     (GPMPC_OK = 0, < 0 error; text via gpmpc_last_error()).
 *   - all arrays are C-contiguous IEEE fp64 (numpy default), row-major; the caller owns every
 *     buffer.  By default array arguments are HOST pointers (copied in/out on the handle's
 *     stream); gpmpc_set_pointer_mode(h, GPMPC_PTR_DEVICE) makes the bulk arguments of the
 *     predict family device pointers (inputs already resident in HBM, no PCIe in the call).
 *   - hyper[a] = [ell_1 .. ell_d, sf, sn], sf and sn are STANDARD DEVIATIONS
 *     (gp_functions.py:129-130, optimize.py:338-340).  Z rows are z = [x, u] in the GP's own
 *     (standardised) input units; standardisation stays in the Python host class like
 *     gp_class.py:253-261.
 *   - one handle = one GP model = one HIP stream.  Calls on different handles are thread-safe (the
 *     factorisations of different handles on one device take turns, see gpmpc_get_counter), calls on
 *     one handle are serialised by the caller.
 */
#ifndef GPMPC_H
#define GPMPC_H

#ifdef __cplusplus
extern "C" {
#endif

#define GPMPC_ABI_VERSION 2   /* 2 (r04): + gpmpc_profile_set_mask, gpmpc_runtime_info, gpmpc_schedule_stats; gpmpc_profile_enable: nonzero = all phases again */

/* status codes */
#define GPMPC_OK 0
#define GPMPC_EINVAL (-1)  /* bad argument */
#define GPMPC_EHIP (-2)    /* HIP runtime error / no usable gfx950 device */
#define GPMPC_ENOTFIT (-3) /* model has no factors yet (call gpmpc_fit / gpmpc_set_factors) */
#define GPMPC_ENOTPD (-4)  /* K not positive definite even after the one-shot 1e-8 jitter */
#define GPMPC_ENOMEM (-5)

/* uncertainty-propagation methods: the strings of GP.set_method, gp_class.py:193-242 */
#define GPMPC_ME 0     /* 'ME'     gp_class.py:212-215: [mean(z), diag(var(z))]            */
#define GPMPC_TA 1     /* 'TA'     gp_class.py:216-219: diag(var) + J Sigma J^T            */
#define GPMPC_EM 2     /* 'EM'     gp_class.py:220-224: gp_exact_moment                    */
#define GPMPC_OLD_ME 3 /* 'old_ME' gp_class.py:225-229: gp()                               */
#define GPMPC_OLD_TA 4 /* 'old_TA' gp_class.py:230-235: gp_taylor_approx(diag=True)        */

/* prior mean functions: the strings of get_mean_function, gp_functions.py:25-69.  With a mean function set, a hyper
 * row is [ell_1..ell_d, sf, sn, mean parameters] (train_gp optimize.py:136-151): */
#define GPMPC_MEAN_ZERO 0       /* 'zero'        m(x) = 0                        no parameters          */
#define GPMPC_MEAN_CONST 1      /* 'const'       m(x) = c                        [c]                    */
#define GPMPC_MEAN_LINEAR 2     /* 'linear'      m(x) = a^T x + c                [a_1..a_d, c]          */
#define GPMPC_MEAN_POLYNOMIAL 3 /* 'polynomial'  m(x) = a^T x^2 + b^T x + c      [a_1..a_d, b_1..b_d, c] */

#define GPMPC_PTR_HOST 0
#define GPMPC_PTR_DEVICE 1

/* profiling phases for gpmpc_profile_read */
#define GPMPC_PH_GRAM 0       /* SE-ARD K build                              */
#define GPMPC_PH_FACTOR 1     /* Cholesky + triangular inverse recursion     */
#define GPMPC_PH_SOLVE 2      /* w = L^-1 y, alpha = L^-T w                  */
#define GPMPC_PH_INVK 3       /* K^-1 = L^-T L^-1                            */
#define GPMPC_PH_CROSSCOV 4   /* ks(X, Z) + mean                             */
#define GPMPC_PH_VARGEMM 5    /* V = L^-1 Ks, column sums of squares (MFMA)  */
#define GPMPC_PH_FINISH 6     /* var / Jacobian / TA covariance assembly     */
#define GPMPC_PH_EM 7         /* exact-moment N x N pair tiles               */
#define GPMPC_PH_NLL 8        /* NLL reductions + gradient pass              */
#define GPMPC_PH_CHAIN 9      /* the persistent chain kernel alone = the Cholesky without the inverse (inside FACTOR) */
#define GPMPC_PH_COUNT 10

typedef struct gpmpc_gp gpmpc_gp; /* opaque model handle */

/* ---- library / device -------------------------------------------------------------------- */
int gpmpc_abi_version(void);
const char* gpmpc_last_error(void); /* thread-local text of the last failing call */
int gpmpc_device_count(int* count);
int gpmpc_device_name(int device, char* buf, int buflen);
/* fp64 MFMA self-test: verifies the v_mfma_f64_16x16x4_f64 fragment layout the kernels assume
 * (layout_out: 0 = row=(lane>>4)+4r documented for gfx950, 1 = row=4(lane>>4)+r) and measures the
 * issue-bound rate of the instruction (TFLOP/s over the whole chip; the fp64 roofline "peak"). */
int gpmpc_mfma_selftest(int device, int* layout_out, double* tflops_out);

/* ---- model life cycle -------------------------------------------------------------------- */
/* Copies X[N x d] and Y[N x Ny] to the device (the GP owns its data, gp_class.py:29-35). */
int gpmpc_create(int device, int N, int d, int Ny, const double* X, const double* Y, gpmpc_gp** out);
int gpmpc_destroy(gpmpc_gp* h);
/* (N, d, Ny); GP.get_size gp_class.py:266-274 derives (N, Ny, Nu = d - Ny) from these. */
int gpmpc_get_size(const gpmpc_gp* h, int* N, int* d, int* Ny);
/* Prior mean function of the model (get_mean_function gp_functions.py:25-69; train_gp optimize.py:100-294 is the
 * reference path that trains with one).  alpha and the NLL are then formed from y - m(X) (optimize.py:75,96,285), and
 * every hyper array of this API -- gpmpc_fit, gpmpc_get_factors, gpmpc_set_factors, gpmpc_nll (and its gradient) -- has
 * rows of gpmpc_hyper_width() = d + 2 + #parameters entries.  add_to_prediction != 0 adds m(z) to the predicted mean and
 * its derivatives, i.e. build_gp(..., meanFunc=kind) gp_functions.py:131,135; 0 reproduces GP.__init__, which calls
 * build_gp WITHOUT meanFunc (gp_class.py:68-71), so the reference's GP object predicts ks^T alpha only.  'EM' and
 * 'old_ME' ignore the mean function like the reference (gp_functions.py:383,232); 'old_TA' with one is refused (the
 * reference raises there, gp_functions.py:309-311).  Changing the kind discards the factors. */
int gpmpc_set_mean_func(gpmpc_gp* h, int kind, int add_to_prediction);
int gpmpc_hyper_width(const gpmpc_gp* h, int* width);
/* Gaussian hyper-priors of calc_NLL (optimize.py:77-93; unused by the reference itself, prior = None :157):
 * prior6 = [ell_mean, ell_std, sf_mean, sf_std, sn_mean, sn_std]; gpmpc_nll then returns NLL + log_prior exactly as
 * :97 does (log-densities of every ell_i, of sf^2 and of sn^2, ADDED to the negative log-likelihood), and its gradient
 * follows.  NULL switches the priors off. */
int gpmpc_set_hyper_prior(gpmpc_gp* h, const double* prior6);
int gpmpc_set_pointer_mode(gpmpc_gp* h, int mode);
int gpmpc_set_stream(gpmpc_gp* h, void* hip_stream); /* NULL restores the handle's own stream */
int gpmpc_synchronize(gpmpc_gp* h);
/* Diagnostics of the factorisation path.  The fit runs its sequential chain and its bulk in persistent kernels that hand
 * tiles over through flags and must be co-resident; on a GPU shared with other work a hand-off can time out (~30 ms),
 * in which case THAT factorisation is repeated on the single-queue path (same result) and the next call tries again;
 * after three consecutive time-outs the handle stays on the single-queue path for 64 fits.  Handles of one process
 * take turns for the factorisation itself.  Counters: "handoff_timeouts", "chained_factorisations",
 * "single_queue_factorisations", "predictions_behind_tail" (see gpmpc_fit), "fused_fit_predicts" (gpmpc_fit_predict_mean_var calls that took the fused route), "persistent_variance_products" (variance products
 * of gpmpc_predict_mean_var that ran as one persistent launch over a static tile schedule, vargemm_persist.hpp); process-wide: "workspace_blocks_fresh" / "workspace_blocks_reused" (the N x N blocks
 * of a workspace, >= 64 MB, come from size classes and return to a free list: gpmpc_append at large N re-uses what the
 * previous append gave back instead of paying for fresh multi-GB allocations). */
int gpmpc_get_counter(gpmpc_gp* h, const char* name, long* value);
/* HIP-event brackets per phase on the handle's stream.  enable: 0 off, nonzero on.  gpmpc_profile_set_mask restricts the
 * brackets to the phases whose bit (the GPMPC_PH_* index) is set (0 = all) -- each bracket is two timing events, ~5 us of the
 * stream's time: with all seven phases of a fit + predict step bracketed the step takes 65 us longer at N = 4096 (r03). */
int gpmpc_profile_enable(gpmpc_gp* h, int enable);
int gpmpc_profile_set_mask(gpmpc_gp* h, unsigned mask);
int gpmpc_profile_read(gpmpc_gp* h, int phase, double* total_ms, long* launches, int reset);

/* ---- fit: a1,a3-a6 ----------------------------------------------------------------------- */
/* K_a = k_a(X,X) + sn_a^2 I (optimize.py:303-319,343-344), L = chol(K) with the one-shot jitter
 * rule (optimize.py:345-350, :483-488; gp_class.py:524-529): info[a] = 0 ok, 1 = 1e-8*I was added
 * once, and the call returns GPMPC_ENOTPD (info[a] = -(first bad pivot index, 1-based)) if that
 * also fails.  alpha = K^-1 y (optimize.py:494), and if want_invK: K^-1 (optimize.py:489-490).
 * hyper is [Ny x gpmpc_hyper_width()] (= d+2 for the zero mean); info may be NULL.
 * The call returns when `info` is known -- with the persistent-kernel factorisation that is when L is complete; the
 * last rows of L^-1 and alpha may still be in flight, and everything this API does afterwards is ordered behind them.
 * The first gpmpc_predict_mean_var with more than 64 points behind a fit (device pointers, the handle's own stream)
 * uses that window: its cross-covariances are formed on a second queue while the last row panel of L^-1 is inverted,
 * its mean next to the variance product (counter "predictions_behind_tail"; same results as any later call). */
int gpmpc_fit(gpmpc_gp* h, const double* hyper, int want_invK, int* info);
/* gpmpc_fit followed by gpmpc_predict_mean_var in ONE call (r06; same arguments, same results bit for bit).  What a caller
 * that refits and then evaluates a batch does in two calls -- GP.optimize / update_data_all followed by validate or a
 * prediction sweep, gp_class.py:131-142,145-190,474-550 -- leaves the device idle between them (the host learns `info`,
 * returns, calls again: ~0.12 ms at N = 4096) and can start the prediction's cross-covariances only then, although they
 * depend on hyper and Z alone.  With device pointers, the handle's own stream, no K^-1, a zero prior mean and 64 < B <=
 * one scratch chunk the prediction's launches are enqueued before the host waits for `info` (the cross-covariances start
 * at the end of the factorisation's chain kernel, next to the tail of L^-1); in every other case this IS the two calls.
 * If the factorisation has to be repeated (jitter rule, hand-off time-out) the prediction is repeated behind it.
 * Measured at N = 4096, B = 10 000: within 10 us of the two calls -- the step behind the chain is bound by the inverse's
 * tail, not by the host (docs/history_r06.md).
 * Counter "fused_fit_predicts". */
int gpmpc_fit_predict_mean_var(gpmpc_gp* h, const double* hyper, int want_invK, int* info, int B, const double* Z,
                               double* mean, double* var);
/* Export in the reference's save_model layout (gp_class.py:693-704): hyper[Ny x gpmpc_hyper_width()],
 * chol[Ny x N x N] (lower, zeros above), alpha[Ny x N], invK[Ny x N x N]; any pointer may be NULL. */
int gpmpc_get_factors(gpmpc_gp* h, double* hyper, double* chol, double* alpha, double* invK);
/* Append n training points and update L, L^-1, alpha with the EXISTING hyper-parameters: the result of
 * GP.update_data_all (gp_class.py:474-550, which recomputes everything from scratch) as a rank-n extension of
 * the factors, O(N^2 n) instead of O(N^3).  info[Ny]: 0 ok, <0 = -(first non-positive pivot); on
 * GPMPC_ENOTPD the model is left unchanged.  Large n (more than a quarter of the new size) refits instead. */
int gpmpc_append(gpmpc_gp* h, int n, const double* Xnew, const double* Ynew, int* info);

/* Import a saved model (GP.load_model -> ctor branch gp_class.py:58-66): chol and hyper are
 * required; alpha == NULL recomputes it from Y; invK == NULL computes it lazily when a method
 * needs it.  L^-1 (the predict operand) is rebuilt on the device. */
int gpmpc_set_factors(gpmpc_gp* h, const double* hyper, const double* chol, const double* alpha,
                      const double* invK);

/* ---- predict: a9-a13 --------------------------------------------------------------------- */
/* mean[B x Ny], var[B x Ny]: build_gp's mean/var functions, gp_functions.py:114-136
 * (mean = ks^T alpha, var = sf^2 - ||L^-1 ks||^2, no noise term). */
int gpmpc_predict_mean_var(gpmpc_gp* h, int B, const double* Z, double* mean, double* var);
/* mean[B x Ny] and J[B x Ny x d] = d mean / d z: mean_jac_z gp_functions.py:146-147 (CasADi AD
 * there, analytic here); the operand of GP.discrete_linearize / jacobian gp_class.py:647-672. */
int gpmpc_mean_jac(gpmpc_gp* h, int B, const double* Z, double* mean, double* J);
/* Derivative outputs for a casadi Callback around GP.__predict (the reference obtains them
 * from CasADi's AD of build_gp / build_TA_cov, gp_functions.py:114-173): besides mean[B x Ny], var[B x Ny]
 * and J[B x Ny x d] = d mean/dz also Hm[B x Ny x d x d] = d2 mean/dz2 and dvar[B x Ny x d] = d var/dz, from
 * which d cov/dz and d cov/dSigma of the 'ME' and 'TA' methods follow in closed form
 * (cov = diag(var) + J Sigma J^T).  Any output pointer may be NULL.  d var/dz needs K^-1 ks: formed as
 * L^-T (L^-1 ks) from the factor the variance already uses, so the call neither needs nor builds K^-1. */
int gpmpc_predict_sens(gpmpc_gp* h, int B, const double* Z, double* mean, double* var, double* J,
                       double* Hm, double* dvar);
/* 'EM' (gp_exact_moment gp_functions.py:344-418) with its first derivatives: besides mean[B x Ny] and cov[B x Ny x Ny]
 * the Jacobians dmean_dz[B x Ny x d], dmean_dS[B x Ny x d x d], dcov_dz[B x Ny x Ny x d], dcov_dS[B x Ny x Ny x d x d]
 * with respect to the input mean z and the d x d entries of the input covariance (entries independent, as in CasADi's
 * jacobian(..., covar_s) which the reference relies on inside IPOPT).  Sigma must be symmetric: like gpmpc_predict's
 * 'EM' value path, which visits each pair (i, j) of training points once, the formulas use Sigma = Sigma^T.  Any output
 * may be NULL; with cov == NULL the value kernels' pair sums are not formed (a Jacobian callback: 3.9 instead of 5.6 ms
 * per input at N = 8192, Ny = 6).  Every d the library takes (<= 16): d <= 8 on an 8-deep, d = 9..16 on a 16-deep cross term. */
int gpmpc_predict_em_sens(gpmpc_gp* h, int B, const double* Z, const double* Sigma, double* mean, double* cov,
                          double* dmean_dz, double* dmean_dS, double* dcov_dz, double* dcov_dS);
/* GP.__predict (gp_class.py:212-235) batched over B input distributions:
 * Z[B x d], Sigma[B x d x d] (ignored for ME/old_ME, may be NULL) -> mean[B x Ny],
 * cov[B x Ny x Ny] in standardised units (gp_class.py:262 leaves cov unscaled). */
int gpmpc_predict(gpmpc_gp* h, int method, int B, const double* Z, const double* Sigma,
                  double* mean, double* cov);
/* gpmpc_predict for 'ME' / 'TA' plus J[B x Ny x d] = d mean / d z from the same pass (what one evaluation of an NLP
 * callback needs: GP.__predict and its jac_x / jac_u, gp_class.py:212-219,239-242). */
int gpmpc_predict_jac(gpmpc_gp* h, int method, int B, const double* Z, const double* Sigma, double* mean,
                      double* cov, double* J);
/* a17: T-step uncertainty propagation entirely on the device (the numeric loop of GP.predict_compare,
 * gp_class.py:777-804: feed (mean_t, cov_t) back into GP.predict), one synchronisation at the end instead of
 * one per step.  All quantities in the GP's standardised units: z0[d] first input, U[T x Nu] controls,
 * Sigma0[d x d] initial input covariance (its [:Ny,:Ny] block is replaced by cov_t every step, the rest kept),
 * sa/sb[Ny]: x_{t+1} = sa * mean_t + sb maps an output mean to the next state input (re-standardisation of
 * GP.predict, gp_class.py:253-261; NULL = identity).  Outputs mean[T x Ny], cov[T x Ny x Ny] (host pointers). */
int gpmpc_rollout(gpmpc_gp* h, int method, int T, const double* z0, const double* U, const double* Sigma0,
                  const double* sa, const double* sb, double* mean, double* cov);
/* The same loop with the reference's state feedback (predict_compare(feedback=True), gp_class.py:772-803):
 * u_t = K (mean_{t-1} - x_ref) and the input covariance [[C, C K^T], [K C, K C K^T]], C = cov_{t-1}.  The caller
 * supplies the gain as data (the reference computes K with mpc_class.lqr from discrete_linearize at (x0, u_0)):
 * Kz[Nu x Ny], k0[Nu]: u_t in the GP's input units as an affine function of the STANDARDISED output mean
 * (u_t = Kz mean_{t-1} + k0: GP.predict's un-/re-standardisation folded in); Kc[Nu x Ny]: the gain applied to the
 * covariance blocks (the reference applies the raw K to the standardised cov, :798-799).  z0 = [x_0, u_0] carries
 * the first control.  U_out[T x Nu] (may be NULL) returns the controls that were applied. */
int gpmpc_rollout_feedback(gpmpc_gp* h, int method, int T, const double* z0, const double* Sigma0, const double* sa,
                           const double* sb, const double* Kz, const double* k0, const double* Kc, double* mean,
                           double* cov, double* U_out);
/* M roll-outs in lock-step (r06; SURVEY a17 "batch across trajectories / methods": gp_class.py:777-804 runs the methods one
 * after the other and synchronises with the predictor at every step).  Trajectory m has its own method methods[m], first
 * input z0[m] (z0[M x d]), controls U[m] (U[M x T x Nu]) and initial input covariance Sigma0[m] (Sigma0[M x d x d]); sa / sb
 * as in gpmpc_rollout.  Outputs mean[M x T x Ny], cov[M x T x Ny x Ny].  Per time step ONE pass over the factors serves
 * every trajectory: all 'ME' / 'TA' trajectories form one prediction batch (the lower triangles of L^-1 are streamed once
 * for up to 32 of them -- at N = 8192, Ny = 6 that stream IS the step: 1.6 GB), the trajectories of a moment method
 * one batched launch set.  M <= 64.  Open loop only (gpmpc_rollout_feedback for the reference's feedback law).
 * Parity: trajectories do not influence each other; a trajectory's numbers are bitwise those of gpmpc_rollout when it
 * is the only 'ME' / 'TA' trajectory of the call (or a moment method), and otherwise agree with it to rounding (the
 * batched variance kernel sums the same N terms in another order than the one-column kernel: ~1e-15 sf^2 per step). */
int gpmpc_rollout_multi(gpmpc_gp* h, int M, const int* methods, int T, const double* z0, const double* U,
                        const double* Sigma0, const double* sa, const double* sb, double* mean, double* cov);
/* a14 GP.covar gp_class.py:353-381: covar[Ny x n x n] = sf^2 - V^T V for n new inputs. */
int gpmpc_covar(gpmpc_gp* h, int n, const double* Xnew, double* covar);

/* ---- training objective: a7 (+ gradient) ------------------------------------------------- */
/* NLL of output `a` at hyper_row[gpmpc_hyper_width()] (calc_NLL_numpy optimize.py:322-356: 0.5 y^T alpha +
 * sum log|L_ii|, jitter rule included; with a mean function calc_NLL optimize.py:22-97 without its unused
 * hyper-priors); grad[gpmpc_hyper_width()] (may be NULL) is dNLL/dhyper, Rasmussen &
 * Williams eq. 5.9 -- the reference has no analytic gradient (optimize.py:371-375).
 * jitter_out (may be NULL) reports whether the jitter branch was taken. */
int gpmpc_nll(gpmpc_gp* h, int a, const double* hyper_row, double* nll, double* grad, int* jitter_out);

/* ---- training: a8 ------------------------------------------------------------------------------ */
/* Multistart hyper-parameter training of all Ny outputs, then the fit at the optimum: train_gp_numpy
 * optimize.py:359-503 (SLSQP + finite differences) / train_gp :100-294 (IPOPT on CasADi AD), here a projected L-BFGS on
 * gpmpc_nll and its analytic gradient, in two stages per restart: (1) length scales and sf -- if their bounds are
 * positive and finite -- in log space, the noise sn and the mean parameters linear (as in the reference's optimisers;
 * d NLL / d log sn vanishes at the reference's start sn = 1e-5, a log-space search never leaves it); (2) if stage 1 stops
 * with iterations to spare, a polish from its end point with sn in log space as well, kept when it ends lower.  An
 * accepted step always lowers the NLL; the L-BFGS memory is dropped when the set of bound-pinned variables changes.
 * optimizer_opts written for IPOPT / SLSQP have no counterpart here: only max_iter and tol exist.  Iteration and
 * evaluation totals of the last call (this rank's restarts, both stages): gpmpc_get_counter "train_iterations" /
 * "train_evaluations", and "train_gflop" = its algorithmic matrix work (N^3/3 per Cholesky, per L^-1 formed, per K^-1).  starts[Ny x nstart x W] (W =
 * gpmpc_hyper_width): one initial point per restart -- the reference starts every restart from the same point
 * (optimize.py:462-466, its Latin-hypercube line :218 is commented out); lb, ub[Ny x W]: the box (optimize.py:434-443 or
 * :204-229; +-HUGE_VAL for none).  For every output the restart with the smallest NLL wins (first minimum, np.argmin
 * :474), the model is fitted there (:476-494) and hyper_opt[Ny x W], obj[Ny x nstart] (+inf: restart failed),
 * theta_all[Ny x nstart x W] (may be NULL), info[Ny] (as gpmpc_fit) are returned.  max_iter <= 0 / tol <= 0 select the
 * defaults (200, 1e-8 on the projected gradient relative to |NLL|).
 * Restart shard (the only part of the path that shards, north_star): restart r is run by rank r mod world.  With an
 * RCCL communicator (gpmpc_rccl_comm_create, one rank per GPU) the ranks exchange their (NLL, theta) rows with ONE
 * ncclAllGather -- (1 + W) doubles per restart -- take the same arg-min and each fits its own copy; nothing else
 * crosses xGMI.  world > 1 with rccl_comm == NULL: no exchange and no fit; the caller merges the obj / theta_all rows
 * of the ranks (row r is valid on rank r mod world) and calls gpmpc_fit.
 * Failure of ONE rank (HIP error, out of memory) never strands the others: the rank still joins the exchange with +inf
 * rows and its error code in a status word of the same all-gather, and EVERY rank returns that code.  Without a
 * communicator the code is written to *status (may be NULL; 0 = fine) and the call returns GPMPC_OK so that the caller
 * reaches its own exchange. */
int gpmpc_train_multistart(gpmpc_gp* h, int nstart, const double* starts, const double* lb, const double* ub,
                           int max_iter, double tol, int rank, int world, void* rccl_comm, int want_invK,
                           double* hyper_opt, double* obj, double* theta_all, int* info, int* status);
/* RCCL bootstrap for the restart shard (librccl is bound at run time).  Rank 0 obtains the 128-byte id and hands it to
 * the other ranks by any side channel (a file, MPI, torch.distributed's store); every rank then creates its
 * communicator on its own GPU. */
int gpmpc_rccl_unique_id(char* id128);
int gpmpc_rccl_comm_create(int device, int world, int rank, const char* id128, void** comm_out);
int gpmpc_rccl_comm_destroy(void* comm);
/* Number of ranks the communicator spans (ncclCommCount): what bench.py reports as `rccl_ranks`. */
int gpmpc_rccl_comm_count(void* comm, int* count);
/* The HIP runtime and the RCCL build this process runs the library on, as text "hip_runtime=<version> hip_path=<file>
 * rccl=<version|unavailable> rccl_path=<file>".  The library runs on the libamdhip64 the process mapped first (PyTorch's
 * bundled copy if torch was imported before the library, /opt/rocm's otherwise) and binds, at run time, the librccl that
 * sits NEXT TO that runtime -- in a process that imported torch first that is the RCCL torch.distributed runs on. */
int gpmpc_runtime_info(char* buf, int buflen);

/* ---- low-level dense ops (host pointers), used by the parity tests ------------------------ */
/* k(X, Z)[n1 x n2] = sf2 exp(-1/2 sum_d (x_d - z_d)^2 / ell_d^2) for X[n1 x d], Z[n2 x d]: GP.covSEard
 * gp_class.py:314-350 (same expanded-form operation order, no FMA contraction). */
int gpmpc_kernel_matrix(int device, int n1, int n2, int d, const double* X, const double* Z,
                        const double* ell, double sf2, double* out);
/* In-place lower Cholesky of the n x n row-major SPD matrix A (np.linalg.cholesky,
 * optimize.py:346).  *info = 0 ok, k > 0: leading minor k is not positive definite (LAPACK dpotrf
 * convention).  Ainv (may be NULL) receives L^-1. */
int gpmpc_cholesky(int device, int n, double* A, double* Ainv, int* info);
/* C = alpha * op(A) op(B) + beta * C on the fp64 MFMA GEMM core (row-major, ld = row length).
 * transa/transb as in BLAS ('N' = 0 / 'T' = 1). */
int gpmpc_dgemm(int device, int transa, int transb, int M, int N, int K, double alpha,
                const double* A, int lda, const double* B, int ldb, double beta, double* C, int ldc);

/* The static tile schedule of the persistent products (host code only, no device needed): mode 0 = the predictive variance of
 * gp_functions.py:122-126 (tilesM x tilesN tiles of 128 x 128 per matrix, the tile in block row tm is min(K, 128 (tm + 1)) / 16
 * slabs long), mode 1 = the lower triangle of K^-1 = L^-T L^-1 (optimize.py:489-490; tiles tm >= tn, (K - 128 tm) / 16 slabs),
 * for `batch` matrices on `slots` resident workgroups dealt round-robin to 8 XCDs.  stats[0..5] = tiles, heaviest slot load,
 * mean slot load (half slabs), tiles on their operand panel's home XCD, tiles scheduled more or less than once (0 when
 * correct), longest list.  What the CPU tier checks: every tile exactly once, balance, locality. */
int gpmpc_schedule_stats(int mode, int tilesM, int tilesN, int batch, int K, int slots, double* stats);

/* Diagnostic knobs for tests and tuning runs (no reference counterpart).  "gemm_tile": 0 (automatic), 32, 64 or 128
 * pins the tile of every GEMM launch of the process, so that small problems reach the large-tile kernels.
 * "cu_count": the number of compute units of device 0 the persistent-kernel factorisation plans with (at most the
 * real count on a GPU; the emulated build accepts up to 64 so that the multi-launch worker schedules can be
 * exercised).  "fail_nll_after": n > 0 makes the n-th gpmpc_nll evaluation from now on return GPMPC_EHIP without touching
 * the device (fault injection: how the tests reach the failure paths of the restart shard); 0 switches it off.
 * "worker_courier": 1 / 0 = the tile-owner worker launches of the chained factorisation run with / without the courier
 * workgroup (two instantiations of one kernel; default 1, or GPMPC_COURIER), -1 back to the default.
 * "handoff_write_through": 1 = the tiles that cross workgroups inside the chained factorisation (chain kernel, tile-owner
 * workers, courier) leave as write-through stores followed by a drained flag (default, or GPMPC_CHAIN_WT / GPMPC_WORKER_WT),
 * 0 = plain stores and an agent-scope release (an L2 write-back) per hand-off, -1 back to the default; same results.
 * "vargemm_persist": the large-batch variance product (gp_functions.py:122-126) as 0 = one 128 x 128 tile per workgroup in
 * the dispatcher's order, 1 = one persistent launch over a static schedule when there are at least two tiles per workgroup
 * slot (default, or GPMPC_VARGEMM_PERSIST), 2 = ... at any size (tests), -1 back to the default; same bits either way.
 * "em_chunk": column tiles (64 wide) one workgroup of the exact-moment pair sums sweeps (gp_exact_moment,
 * gp_functions.py:397-414; default 64, or GPMPC_EM_CHUNK); 0 back to the default; results agree to rounding.
 * "em_diag_segs": the a == b pair sums of the same (the ones that stream K^-1): n > 0 = the lower triangle of tiles in n equal
 * ranges per pair (default: four workgroups per CU shared by the outputs, at most one per partial-sum slot; or
 * GPMPC_EM_DIAG_SEGS), 0 = by strips and chunks like the a != b pairs, -1 back to the default; results agree to rounding.
 * Returns GPMPC_EINVAL for an unknown name or value. */
int gpmpc_set_tuning(const char* name, int value);

#ifdef __cplusplus
}
#endif
#endif /* GPMPC_H */
