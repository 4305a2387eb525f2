/*
 * muse_hip.h -- C ABI of libmuse_hip.so, the MI355X (gfx950) engine for the MUSE inner loop.
 *
 * This is the drop-in boundary for the hot path of marius311/MuseInference.jl: the per-simulation
 * chain  sample_x_z -> zhat_at_theta (L-BFGS over z) -> grad_theta logLike  that the reference runs
 * once per element of the pmap in muse! / get_J! / get_H!.  Each entry point cites the reference
 * interface it replaces (paths relative to the reference repository root).  A Julia maintainer binds
 * these with `ccall((:sym, libmuse_hip), Cint, (...), ...)` from methods on a HipMuseProblem <:
 * AbstractMuseProblem (julia/HipMuseInference.jl); the Python host in
 * museinference.jl_amd/ binds them with ctypes.
 *
 * Conventions
 *   - plain C: pointers and sizes only, no C++/torch types; every call returns 0 on success or a
 *     negative MUSE_ERR_* code, with a human-readable message from muse_last_error() (thread-local).
 *   - all floating point is IEEE double ("f64"); vectors are contiguous; batched arrays are row-major
 *     [element][component].
 *   - `mem` arguments say where the caller's vectors live: MUSE_MEM_HOST or MUSE_MEM_DEVICE
 *     (a hipMalloc'd / torch device pointer on the context's device).
 *   - theta is always passed in the UN-transformed space (src/interface.jl:74-75,96-97,146-147) as a
 *     host array of ntheta doubles.
 *   - one context per device; calls on one context are serialised on its stream; distinct contexts
 *     are independent and may be driven from different threads.
 *   - there is no CPU fallback: every entry fails with MUSE_ERR_HIP if no gfx950 device is usable.
 *
 * Models (the reference takes user closures, src/simple.jl:79-89; closures cannot cross a C ABI, so
 * the models of BASELINE.json's configs are compiled in):
 *   MUSE_MODEL_FUNNEL  z_i ~ N(0, e^{theta_k(i)}), x_i ~ N(z_i, 1); ntheta contiguous equal blocks,
 *                      k(i) = floor(i*ntheta/N).  ntheta = 1 is the reference's funnel
 *                      (src/simple.jl:59-73, docs/src/index.md:154-168).
 *   MUSE_MODEL_NOISE   z_i ~ N(0,1), x_i ~ N(z_i, e^theta); ntheta = 1.
 *   MUSE_MODEL_SMOOTH  z as FUNNEL, x = A z + n with A the periodic (1/4,1/2,1/4) stencil.
 *   MUSE_MODEL_USER    a user-supplied elementwise model (the closures of SimpleMuseProblem as compiled code): three C
 *                      functions in a header (include/muse_model.h: the joint draw, d(-logLike)/dz with the objective's
 *                      element term, the score's element term) compiled into an engine library of its own, which exports
 *                      this same ABI for that one model.  muse_model_name() says what a given library holds.
 *
 * Random streams: simulation `sim` of master seed `seed` always sees the same normals
 * (Philox4x32-10 keyed by seed, counter = (element, sim)), in every call and on every GPU -- the
 * reference's split_rng contract (src/util.jl:87-92, src/muse.jl:134,169,323,506).
 */
#ifndef MUSE_HIP_H
#define MUSE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MUSE_MODEL_FUNNEL 0
#define MUSE_MODEL_NOISE 1
#define MUSE_MODEL_SMOOTH 2
#define MUSE_MODEL_USER 3 /* only in a library built from a model header (include/muse_model.h) */

#define MUSE_MEM_HOST 0
#define MUSE_MEM_DEVICE 1

/* ntheta: 1 .. MUSE_MAX_THETA in every placement and entry point (per-block coefficients and sums in the launch's LDS argument
 * block and in registers); MUSE_MAX_THETA + 1 .. MUSE_MAX_THETA_EXT in the streaming placements (per-block tables read from the
 * kernel-argument segment, block sums eight at a time): batched maps, the per-simulation operators, both get_H! branches, the
 * exchange between ranks -- not the native muse! loops (muse_run*: MUSE_ERR_INVALID; a host loop over the batched maps, as the
 * reference's muse! is one, takes their place) nor several maps per launch.  The reference has no bound (src/muse.jl:296-333). */
#define MUSE_MAX_THETA 8
#define MUSE_MAX_THETA_EXT 64

/* error codes */
#define MUSE_OK 0
#define MUSE_ERR_INVALID (-1) /* bad argument                       */
#define MUSE_ERR_HIP (-2)     /* HIP runtime failure / no device    */
#define MUSE_ERR_NODATA (-3)  /* include_data without muse_set_data */
#define MUSE_ERR_ALLOC (-4)   /* device allocation failed           */
#define MUSE_ERR_RCCL (-5)    /* RCCL failure                       */

/* starting point of a MAP solve */
#define MUSE_Z0_ZERO 0 /* zero(z): muse! first iteration (src/muse.jl:151), zhat_guess_from_truth (src/interface.jl:184-186) */
#define MUSE_Z0_TRUE 1 /* the simulation's own z: get_J! (src/muse.jl:511); the data element starts from zero */
#define MUSE_Z0_WARM 2 /* the context's resident zhat of the same element: muse! iterations > 1 (src/muse.jl:181) */

/* solver status per element; the host applies the reference's semantics (src/interface.jl:168-171):
 * status >= MUSE_STATUS_MAXITER is "did not converge" (warning), MUSE_STATUS_NONFINITE is the @error case. */
#define MUSE_STATUS_G_CONVERGED 0       /* ||grad_z||_inf <= atol */
#define MUSE_STATUS_X_CONVERGED 1       /* step of exactly zero length (x_abstol = 0) */
#define MUSE_STATUS_F_CONVERGED 2       /* objective unchanged twice in a row (f_abstol = 0, successive_f_tol = 1) */
#define MUSE_STATUS_MAXITER 3           /* 1000 iterations */
#define MUSE_STATUS_LINESEARCH_FAILED 4 /* HagerZhang gave up (LineSearchException) */
#define MUSE_STATUS_NONFINITE 5         /* non-finite objective or gradient */

/* what Optim's result carries that muse! keeps in `history` (src/muse.jl:171,175,218) */
typedef struct muse_info {
    int32_t iterations; /* L-BFGS iterations K */
    int32_t f_calls;    /* objective+gradient evaluations E */
    int32_t status;     /* MUSE_STATUS_* */
    int32_t hist_words; /* sum over iterations of the number of (dx,dg) pairs used by the two-loop recursion */
    double f_min;       /* -logLike at the returned point */
    double gnorm;       /* ||grad_z||_inf at the returned point */
} muse_info;

typedef struct muse_ctx muse_ctx;

/* ---- context ------------------------------------------------------------------------------- */
/* Replaces constructing a problem object (SimpleMuseProblem, src/simple.jl:79-89). */
int muse_ctx_create(int model, int64_t N, int ntheta, int device, muse_ctx** out);
int muse_ctx_destroy(muse_ctx* ctx);
const char* muse_last_error(void);
/* The name of model id `model` in THIS library, or NULL when the library does not hold it: "funnel" / "noise" / "smooth" in
 * libmuse_hip.so; MUSE_MODEL_NAME of the header in a library built from a user's model (and NULL for the built-in ids).
 * The user's sample_x_z / logLike closures of SimpleMuseProblem (src/simple.jl:79-89) are that header's functions. */
const char* muse_model_name(int model);
/* prob.x, the observed data (src/simple.jl:5, used at src/muse.jl:170). */
int muse_set_data(muse_ctx* ctx, const double* x, int mem);
/* Run the context's work on a caller-owned hipStream_t (NULL = the context's own stream). */
int muse_set_stream(muse_ctx* ctx, void* hip_stream);
/* Storage policy of the solver kernel: -1 automatic, 0 streaming (vectors in HBM), 1 resident
 * (vectors in registers/LDS, N <= muse_max_resident_n()).  Results are bitwise independent of it. */
int muse_set_placement(muse_ctx* ctx, int placement);
int64_t muse_max_resident_n(void);
/* Workgroups that cooperate on ONE element of a map.  The reference chooses the parallel axis by what there is more
 * of (src/muse.jl:327-333: sims or Jacobian columns); when a launch has fewer elements than the GPU has compute
 * units -- the per-GPU share of a strongly scaled map -- the remaining axis is the element itself: `split`
 * workgroups (2, 4, 8 or 16) share one element's vectors and meet in every reduction.  0 or 1 = the default
 * (a function of N alone: one workgroup up to N = 65535, clusters of 8 or 16 above).  Results are a function of
 * (seed, sim, theta, N, model, split): for a given split they do not depend on the batch, the grid or the number of
 * GPUs, but the split changes the summation tree, i.e. the last bits (scores agree to ~1e-13 relative). */
int muse_set_element_split(muse_ctx* ctx, int split);
/* How the batched maps of this context run (a function of model, N, the placement and the element split): threads per
 * workgroup, workgroups per element, whether the solver's vectors are resident in registers/LDS (else streamed from HBM),
 * and -- stencil model in a cluster -- whether the search direction is kept in LDS.  For reports (bench.py's roofline
 * accounts the bytes of the placement that ran); any pointer may be NULL. */
int muse_placement_info(muse_ctx* ctx, int* threads, int* workgroups_per_element, int* resident, int* direction_in_lds);
/* Run-time constants of a user-supplied model (include/muse_model.h: a header with MUSE_MODEL_NCONST reads them through
 * muse_const(k, i)) -- what a closure of the reference's SimpleMuseProblem captures (src/simple.jl:79-89): a known spectrum,
 * a noise-variance map, a mask -- without compiling their values into the library: vector k < MUSE_MODEL_NCONST, `count` = N
 * finite doubles (host or device memory).  May be called again at any time (it waits for the context's launches first).
 * One set per model library and process is installed at a time; a context re-installs its own before it launches if another
 * context of the library has installed others since.  Libraries whose model declares no constants refuse the call. */
int muse_set_constants(muse_ctx* ctx, int k, const double* values, int64_t count, int mem);
/* The functions of a user-supplied model's header evaluated on the HOST for one element (include/muse_model.h) -- what the
 * reference gets from AD for free has to be checkable for hand-written derivatives (src/simple.jl:84-85): Python's
 * check_model_consistency differentiates these values numerically.  out[10] = { muse_model_grad's return value, the objective
 * term it adds to its accumulator (A + iv B), muse_model_score_term (B), ozz, ozx, bz, bx of muse_model_second, and z, x,
 * dx/dsd of muse_model_sample / muse_model_dx_dsd at (sd, n1, n2) }; entries 3-6 and 9 are NaN when the header does not define
 * MUSE_MODEL_SECOND.  A header of the two-parameter family (MUSE_MODEL_PAIR, round 6) writes TWELVE doubles: `iv` and `sd` are the
 * block's two PARAMETERS a, b, and out[12] = { muse_model_grad, the objective term, t0 of muse_model_score_terms, the four
 * coefficients of muse_model_coefs(a, b), z, x of muse_model_sample, the block's constant C(a, b), t1, 0 }.  The model's run-time
 * constants are the context's; a model without them needs no context (ctx may be NULL, and no GPU is touched).  Built-in models:
 * MUSE_ERR_INVALID. */
int muse_model_eval(muse_ctx* ctx, double iv, double sd, double x, double z, double n1, double n2, int64_t i, double* out);
/* 1 when the library's model supplies second derivatives (the built-in models; a user header with MUSE_MODEL_SECOND), i.e.
 * when muse_implicit_H_* accept it; 0 otherwise. */
int muse_model_has_second(void);
/* The normals cache of plain maps.  A simulation's stream depends only on (seed, simulation index) (split_rng,
 * src/util.jl:87-92), and the reference's loops draw the same streams again and again -- every iteration of muse!
 * (src/muse.jl:134,169), get_J! after it (:506), every grid point of get_H! (:430).  The native loops and the
 * finite-difference batch keep a stream's standard normals in device memory after the first draw and load them instead of
 * running the generator again (the same doubles: bit-identical results).  Plain maps do the same on their own when a map
 * asks for a simulation range the context has just been asked for: the second time the normals are stored beside the
 * solve, from the third time on they are loaded (LDS-resident placement, one lane; at most MUSE_NCACHE_MAX_MB, default
 * 8192).  enabled = 0 switches that off for plain maps (a benchmark that repeats one map to time the generator). */
int muse_set_normals_cache(muse_ctx* ctx, int enabled);
/* Concurrency of the batched maps: with n > 1 result area r runs on lane r mod n -- a stream, a workgroup scratch, a
 * ticket counter and a cluster state of its own -- so that consecutive launches (enqueued on different result areas) overlap:
 * a launch starts on the compute units the previous one has already left instead of behind its last workgroup and a
 * launch gap.  Results are unchanged.  A lane keeps resident zhat slots of its own (a streaming solve works in its
 * slot); muse_get_zhat / muse_set_zhat and warm starts (MUSE_Z0_WARM) address lane 0's, and a map that warm-starts, the
 * muse_run loops, the finite-difference / implicit maps and the RCCL gather run on lane 0.  n in [1, 4]; 1 (default) =
 * one launch after the other. */
int muse_set_concurrency(muse_ctx* ctx, int nlanes);
int muse_synchronize(muse_ctx* ctx);
/* Device time in ms of the most recent solver launch (HIP events on the context's stream), recorded only
 * after muse_set_timing(ctx, 1): the event pair costs about 12 us of host time per launch (default: off). */
int muse_last_kernel_ms(muse_ctx* ctx, float* ms);
int muse_set_timing(muse_ctx* ctx, int enabled);
/* Live timing of every solver launch between begin and end: HIP event pairs recorded on the
 * context's stream around each launch (up to max_launches); end() synchronises and returns the
 * per-launch durations in ms.  The reference only keeps wall-clock deltas (src/muse.jl:161,210,232). */
int muse_profile_begin(muse_ctx* ctx, int max_launches);
int muse_profile_end(muse_ctx* ctx, float* ms_out, int cap, int* count);
/* Shader clock (Hz) during the last launch profiled between begin and end: workgroup 0 of such a launch stamps the
 * shader-cycle counter and the constant 100 MHz counter at its entry and exit (no stamp executes otherwise). */
int muse_profile_clock_hz(muse_ctx* ctx, double* hz_out);

/* ---- diagnostics ------------------------------------------------------------------------------------------------------------
 * Not needed by a caller of the reference's interface; declared because tests and the tuning tools use them and the library
 * exports exactly what this header declares.
 * muse_debug_flags: switches of a LIVE context (the environment switches of csrc/switches.hpp are read once, by muse_ctx_create).
 * No bit changes a result (tests hold the alternatives bit-equal).  Kernel side: 1 x from the data vector; 2 the loop kernel does
 * not fetch ahead; 3 its old element order; 4 test hook: odd workers leave before their first solve; 5 no speculating trials;
 * 6 a solving stepper takes the data element itself; 7 the stepper never solves; 8 the data vector is not sent ahead; 9 the loop kernel
 * stores every MAP in every iteration (default: the one a worker carries into the next iteration in registers when the loop ends).  Host side:
 * 16 muse_run_sharded's scores meet on the board in pinned host memory; 17 muse_run_sharded runs the host-driven loop;
 * 18 test hook: a loop launch with more workgroups than can be resident at once; 19 the native loops say on stderr which loop
 * ran and what it cost; 20 get_H!'s finite-difference map as ONE launch that carries its fiducial MAP (measured slower: off);
 * 21 get_H!'s fiducial MAP draws its own normals (default: a kernel of its own draws them with the whole GPU).  (Bit 0, "skip the solve", is for timing the launch shell only: results are then meaningless.)
 * muse_debug_stamps: a library built with -DMUSE_STAMPS records s_memtime stamps per problem (tools/stamps.py); out == NULL arms
 * a buffer for nproblems problems, otherwise [nproblems][16] stamps are copied out.  The product build records none. */
int muse_debug_flags(muse_ctx* ctx, int flags);
int muse_debug_stamps(muse_ctx* ctx, int64_t nproblems, unsigned long long* out);

/* ---- per-simulation operators (the AbstractMuseProblem interface, src/interface.jl:4-186) ----- */
/* sample_x_z(prob, rng, theta) -> (;x, z)                       src/interface.jl:92-99, src/simple.jl:95 */
int muse_sample_x_z(muse_ctx* ctx, uint64_t seed, int64_t sim, const double* theta, double* x_out,
                    double* z_out, int mem);
/* logLike_and_grad_z(prob, x, z, theta) -> (logLike, grad_z logLike)      src/interface.jl:68-83, src/simple.jl:94 */
int muse_logLike_and_grad_z(muse_ctx* ctx, const double* x, const double* z, const double* theta,
                            double* logLike_out, double* grad_out, int mem);
/* grad_theta logLike(prob, x, z, theta) -> ntheta doubles (host)            src/interface.jl:41-58, src/simple.jl:92 */
int muse_grad_theta(muse_ctx* ctx, const double* x, const double* z, const double* theta, double* g_out,
                    int mem);
/* zhat_at_theta(prob, x, z0, theta; grad_z_logLike_atol) -> (zhat, info)    src/interface.jl:141-171
 * L-BFGS (m=10, HagerZhang, InitialStatic(1), scaled H0) on -logLike, stop ||grad||_inf <= atol. */
int muse_zhat_at_theta(muse_ctx* ctx, const double* x, const double* z0, const double* theta, double atol,
                       double* z_out, muse_info* info, int mem);

/* ---- batched map bodies (one launch for all elements of the reference's pmap) ----------------- */
/* The muse! map body (src/muse.jl:169-176) and the get_J! map body (src/muse.jl:508-525):
 * for every element: x = data | sample_x_z(rng_sim, theta).x ; zhat = zhat_at_theta(x, z0, theta) ;
 * g = grad_theta(x, zhat, theta).  Elements are [data (if include_data)], sim_begin .. sim_end-1,
 * in that order; element e's zhat stays resident in the context at slot e (warm start of the next
 * call, src/muse.jl:181).  g_out is [n][ntheta] host, info_out [n] host (may be NULL). */
int muse_map_and_score_batch(muse_ctx* ctx, uint64_t seed, int64_t sim_begin, int64_t sim_end,
                             int include_data, const double* theta, double atol, int z0_mode, double* g_out,
                             muse_info* info_out);
/* Same, enqueue only; muse_batch_wait() blocks and copies the results out.  Several batches may be
 * enqueued before waiting only if they use distinct result areas: `result_area` in [0, 4). */
int muse_map_and_score_batch_async(muse_ctx* ctx, uint64_t seed, int64_t sim_begin, int64_t sim_end,
                                   int include_data, const double* theta, double atol, int z0_mode,
                                   int result_area);
int muse_batch_wait(muse_ctx* ctx, int result_area, double* g_out, muse_info* info_out);
/* Several independent maps in ONE launch: `nmaps` (<= MUSE_MAX_MAPS) maps over the same elements, map m at
 * thetas[m][0..ntheta) -- the pmap body of src/muse.jl:169-176 / :508-525 evaluated for several theta at once (the way
 * get_H!'s finite-difference map already carries its perturbed thetas in one launch, src/muse.jl:426-432).  What it is
 * for: a launch with fewer elements than the GPU has compute units -- one rank's share of a strongly scaled map --
 * leaves most of the GPU idle for one problem's latency; nmaps such maps resident at once fill it.  Element e of map m
 * is row m*n + e of the outputs (n = elements per map) and keeps its zhat at slot m*n + e.  Results are those of
 * nmaps separate muse_map_and_score_batch_async calls, bit for bit.  muse_batch_wait returns all nmaps*n rows. */
#define MUSE_MAX_MAPS 8
int muse_map_and_score_multi_async(muse_ctx* ctx, uint64_t seed, int64_t sim_begin, int64_t sim_end, int include_data,
                                   int nmaps, const double* thetas, double atol, int z0_mode, int result_area);
/* The muse! outer loop itself (src/muse.jl:112-232) for the common case -- untransformed theta, the "sims"
 * Jacobian update H^-1_like' = Diagonal(-1 ./ var(g_sims')) (src/muse.jl:188-191), constant alpha, identity
 * `regularize`, a flat or independent-Gaussian prior (src/simple.jl:69-71) -- so that between two map launches
 * the host spends microseconds, not an interpreter's per-iteration overhead: per iteration one
 * (nsims+1)-element map (first from zero(z) or, with z0_warm, from the resident zhat; later ones warm,
 * src/muse.jl:151,181), then g_like' = g_dat' - mean(g_sims') (:183), the prior gradient/Hessian (:184,207),
 * H^-1_post' = inv(inv(H^-1_like') + H_prior') (:208), theta <- theta - alpha H^-1_post' g_post' (:224), and
 * at the top of iterations i > 2 the convergence test sqrt(-(dtheta' H^-1_post' dtheta)) < theta_rtol on the
 * last two records (:163-166).  Everything else (Broyden updates, callable alpha, transforms, checkpoints)
 * stays with the host driver, which builds the same history records from these arrays.
 * Outputs (host, caller-allocated for maxsteps iterations; n = number of iterations run, <= maxsteps):
 *   theta_out [ntheta]                       result.theta, the iterate after the last step (src/muse.jl:230)
 *   hist_out  [maxsteps][MUSE_RUN_HIST(ntheta)] per iteration: theta (where the map ran), g_like_dat,
 *             g_like, g_prior, g_post, diag(H^-1_like), diag(H_prior) (ntheta each), H^-1_post
 *             (ntheta x ntheta, row-major), wall seconds of the iteration
 *   gsims_out [maxsteps][nsims][ntheta]      g_like_sims of every iteration (result.gs = the last one, :231)
 *   info_out  [maxsteps][nsims+1]            solver infos, data element first (may be NULL) */
typedef struct muse_run_options {
    int32_t nsims, maxsteps;
    double theta_rtol, atol, alpha;
    int32_t prior_kind; /* 0 flat (src/interface.jl:120-121), 1 independent Gaussian */
    int32_t z0_warm;    /* first iteration starts from the resident zhat (a z0 was given, src/muse.jl:151) */
    double prior_mean[MUSE_MAX_THETA], prior_sigma[MUSE_MAX_THETA];
} muse_run_options;
#define MUSE_RUN_HIST(ntheta) (7 * (ntheta) + (ntheta) * (ntheta) + 1)
int muse_run(muse_ctx* ctx, uint64_t seed, const double* theta0, const muse_run_options* opt, int32_t* niter_out,
             double* theta_out, double* hist_out, double* gsims_out, muse_info* info_out);
/* The same loop, same arguments, same results bit for bit, with NO host in it: ONE launch runs every outer iteration.
 * Its workgroups are all resident (the grid is what the occupancy query admits); workgroup w owns elements w, w + grid,
 * ... in every iteration, publishes their scores as tagged 8-byte granules, sweeps everybody's granules at the end of the
 * iteration (no fence, no barrier, no counter) and forms the reductions, the prior terms, H^-1_post', the Newton-Raphson
 * step, the history record and the convergence test for itself from the same bits; the next theta never leaves the
 * chip.  (exp(theta/2), exp(-theta) are a fixed sequence of IEEE operations on host and device, and the score moments one
 * fixed 64-leaf summation tree, for this reason.)  The loop kernel runs where it is the faster loop: the register/LDS-resident
 * placements (N <= 10 000, no element split) with up to four theta components (three or four: up to six elements per workgroup at
 * N > 4096), or five to eight components and at most one element per workgroup (three for 512 < N <= 4096); everything else (the
 * streaming placements, score blocks beyond the step's LDS arrays: nsims * ntheta above ~19 000, a loop kernel whose private segment
 * exceeds 256 bytes per lane -- a user's model header can do that) runs muse_run's loop instead -- the same bits.
 * REQUIREMENT: every workgroup of the loop kernel must be resident at once (they meet once per iteration); the grid is sized from an
 * occupancy query that assumes the GPU is this process's.  On a GPU that something else is using they may not be: every wait inside
 * the kernel is bounded (4 s), the call then returns MUSE_ERR_HIP ("not all resident") -- once: the context remembers it and its later
 * calls of muse_run_device run muse_run's loop by themselves.  A caller whose run did not start from the resident MAPs (z0_warm = 0)
 * can simply call muse_run after that error: the same bits. */
int muse_run_device(muse_ctx* ctx, uint64_t seed, const double* theta0, const muse_run_options* opt, int32_t* niter_out,
                    double* theta_out, double* hist_out, double* gsims_out, muse_info* info_out);

/* Resident MAP state: read back / restore zhat of slots [slot_begin, slot_end) ([n][N], row-major).
 * Serves save_MAPs (src/muse.jl:139-143,219) and checkpoint/resume (src/muse.jl:134-135,234). */
int muse_get_zhat(muse_ctx* ctx, int64_t slot_begin, int64_t slot_end, double* out, int mem);
int muse_set_zhat(muse_ctx* ctx, int64_t slot_begin, int64_t slot_end, const double* in, int mem);

/* The get_H! finite-difference branch (src/muse.jl:407-446 with pjacobian, src/util.jl:9-27, and
 * fdm = central_fdm(3,1), src/muse.jl:300): for sims sim_begin..sim_end-1 and every theta component j,
 *   H[sim][:, j] = ( -1/2 f(theta0 - step_j e_j) + 1/2 f(theta0 + step_j e_j) ) / step_j,
 *   f(theta) = grad_theta( x(theta; the sim's randoms), zhat(x; theta0, start zfid), theta0 ).
 * fid_mode 0 reproduces the reference (src/muse.jl:417-423: every fiducial warm start is the MAP of
 * the ONE simulation drawn from the un-split master stream, here stream index `fid_sim`);
 * fid_mode 1 starts each sim from the MAP of its own fiducial simulation.
 * Hs_out is [nsims][ntheta][ntheta] host, Hs[s][i][j] = d g_i / d theta_j; info_out
 * [nsims][ntheta][2] host (may be NULL), last index 0 = plus, 1 = minus. */
int muse_fd_jacobian_batch(muse_ctx* ctx, uint64_t seed, int64_t sim_begin, int64_t sim_end,
                           const double* theta0, const double* step, double atol, int fid_mode,
                           int64_t fid_sim, double* Hs_out, muse_info* info_out);

/* The same finite differences over a range of COLUMNS of the list (sim_begin, column 0), (sim_begin, column 1), ...:
 * element e in [col_begin, col_end) is column e % ntheta of the Jacobian of simulation sim_begin + e / ntheta.  This is
 * the reference's other parallel axis -- get_H! maps over Jacobian columns instead of sims when there are more of them
 * (src/muse.jl:327-333, pjacobian's pool, src/util.jl:9-27) -- and what lets ranks share `nsims * ntheta` units evenly
 * when nsims is small (the reference default is nsims = 10, src/muse.jl:303).  A range may begin and end inside a
 * simulation's Jacobian.  cols_out [n][ntheta] host, cols[e][i] = d g_i / d theta_(e % ntheta); info_out [n][2]. */
int muse_fd_jacobian_columns(muse_ctx* ctx, uint64_t seed, int64_t sim_begin, int64_t col_begin, int64_t col_end,
                             const double* theta0, const double* step, double atol, int fid_mode, int64_t fid_sim,
                             double* cols_out, muse_info* info_out);

/* The raw values behind any finite-difference method (FiniteDifferences' fdm(f, x[, step]) evaluates f on x + step*grid,
 * src/util.jl:9-27): for the units [col_begin, col_end) of the same list and `ngrid` grid points each,
 *   f(eps) = grad_theta( x(theta0 + eps e_j; the sim's randoms), zhat(x; theta0, start zfid), theta0 )
 * at eps = offsets[.][g].  offsets is [ntheta][ngrid] -- one row per column j, shared by the simulations -- or, with
 * offsets_per_unit, [n][ngrid], one row per unit (an adaptive step is estimated per simulation and column).  An offset
 * of 0 evaluates at theta0 itself.  f_out [n][ngrid][ntheta] host, info_out [n][ngrid] (may be NULL).  The host combines
 * the values with the method's coefficients: central_fdm(p, 1) for any p, and the step estimation that get_H! runs when
 * neither `step` nor result.gs exist (src/muse.jl:300,411-413). */
int muse_fd_values_columns(muse_ctx* ctx, uint64_t seed, int64_t sim_begin, int64_t col_begin, int64_t col_end,
                           const double* theta0, int ngrid, const double* offsets, int offsets_per_unit, double atol,
                           int fid_mode, int64_t fid_sim, double* f_out, muse_info* info_out);

/* The get_H! implicit-differentiation branch (src/muse.jl:335-405) for sims sim_begin..sim_end-1:
 *   H = H1 - dFdtheta^T A^{-1} dFdtheta1, A = Hessian_z logLike at (x, zhat, theta0), A^{-1} by conjugate
 *   gradients (IterativeSolvers.cg defaults: x0 = 0, reltol sqrt(eps), maxiter = cg_maxiter, reference 100);
 *   zhat from zero(z) to `atol` (the reference hard-codes 1e-1 here, src/muse.jl:344).  The reference takes the
 *   derivative operands by nested AD; for the compiled-in models they are closed forms, for a user-supplied model the
 *   header's muse_model_second / muse_model_dx_dsd (include/muse_model.h; a header without them is refused).
 * Hs_out [nsims][ntheta][ntheta] host; cg_iters_out [nsims][ntheta] host (may be NULL; the
 * metadata[:implicit_diff_cg_hists] of src/muse.jl:405). */
int muse_implicit_H_batch(muse_ctx* ctx, uint64_t seed, int64_t sim_begin, int64_t sim_end, const double* theta0,
                          double atol, int cg_maxiter, double* Hs_out, int32_t* cg_iters_out);
/* ... and over a range of columns of the same list (see muse_fd_jacobian_columns): cols_out [n][ntheta],
 * cg_iters_out [n] (may be NULL). */
int muse_implicit_H_columns(muse_ctx* ctx, uint64_t seed, int64_t sim_begin, int64_t col_begin, int64_t col_end,
                            const double* theta0, double atol, int cg_maxiter, double* cols_out, int32_t* cg_iters_out);

/* ---- multi-GPU exchange of the per-sim accumulators -------------------------------------------- */
/* Collectives C1-C3 of SURVEY.md §2: gather of per-rank score blocks (so that every rank reduces
 * mean/var/cov in the reference's sim order, src/muse.jl:183,188,529) and sum of per-rank H
 * accumulators (src/muse.jl:446).  One communicator per context/rank.  Two transports, chosen by the id
 * that rank 0 creates and every rank passes to muse_comm_init:
 *   MUSE_TRANSPORT_RCCL  RCCL collectives over xGMI (any number of nodes);
 *   MUSE_TRANSPORT_SHM   the ranks of ONE node: the blocks are kilobytes that every rank's HOST consumes
 *                        (src/muse.jl:177-188), so each rank copies its block from its pinned result area into a
 *                        POSIX shared-memory segment and publishes a sequence number -- no collective kernel, no
 *                        second stream; sums are taken in rank order (bitwise the same on every rank).
 * block_doubles: capacity per rank of one gathered block in the shared segment (0 = 16384); the synchronous
 * collectives move longer messages in pieces, a gathered map must fit. */
#define MUSE_UNIQUE_ID_BYTES 128
#define MUSE_TRANSPORT_RCCL 0
#define MUSE_TRANSPORT_SHM 1
int muse_comm_unique_id(void* id_out /* MUSE_UNIQUE_ID_BYTES, call on rank 0 and broadcast */); /* RCCL */
int muse_comm_unique_id_ex(int transport, int64_t block_doubles, void* id_out);
int muse_comm_init(muse_ctx* ctx, int nranks, int rank, const void* id);
int muse_comm_transport(muse_ctx* ctx, int* transport_out); /* of an initialised communicator */
/* How many ranks the initialised communicator itself counts (RCCL: ncclCommCount; shared memory: the processes that
 * have mapped the segment) -- evidence for reports that the exchange really spans the ranks the launcher started. */
int muse_comm_ranks_seen(muse_ctx* ctx, int* nranks_out);
int muse_comm_destroy(muse_ctx* ctx);
/* every rank contributes count doubles (host), recv_out is [nranks][count] host */
int muse_allgather_scores(muse_ctx* ctx, const double* send, int64_t count, double* recv_out);
/* in-place sum over ranks of count doubles (host) */
int muse_allreduce_sum(muse_ctx* ctx, double* buf, int64_t count);
/* The sharded map body (the pmap of src/muse.jl:169-176 / :508-525 over a pool of GPUs, src/util.jl:74-83):
 * this rank's block [sim_begin, sim_end) (plus the data element where include_data) is solved exactly as
 * by muse_map_and_score_batch_async; RCCL transport: the scores stay on the device and the per-rank blocks are
 * all-gathered by RCCL on a stream of the communicator's own, so that the collective of batch k overlaps
 * the solver launch of batch k+1 and no host copy sits between solver and collective; shared-memory transport:
 * the launch is the plain one and the exchange happens inside muse_batch_wait_gathered.  Every rank passes
 * the same rows_per_rank >= its own element count; shorter blocks are zero-padded.  Every rank makes the same
 * sequence of gather/wait calls per result area.
 * muse_batch_wait_gathered blocks until the area's gathered block has landed: g_all_out is
 * [nranks][rows_per_rank][ntheta] host, info_out this rank's own [n] infos (may be NULL). */
int muse_map_and_score_batch_gather_async(muse_ctx* ctx, uint64_t seed, int64_t sim_begin, int64_t sim_end,
                                          int include_data, const double* theta, double atol, int z0_mode,
                                          int64_t rows_per_rank, int result_area);
int muse_batch_wait_gathered(muse_ctx* ctx, int result_area, double* g_all_out, muse_info* info_out);
/* muse_run over the ranks of the communicator (the muse! loop, src/muse.jl:159-232, with its pmap over a pool of GPUs,
 * src/util.jl:74-83): rank r owns the contiguous block of the opt->nsims simulations that a static partition gives it (the
 * first nsims mod nranks ranks one more), the data element lives on rank 0; per iteration one gathered map, after which
 * every rank takes the same step from the same scores -- theta is never exchanged, and the trajectory is the unsharded
 * muse_run's bit for bit.  Arguments and outputs as muse_run, on every rank; info_out (may be NULL) receives THIS rank's
 * solver infos, [maxsteps][this rank's element count] (rank 0: the data element first).
 * Round 5: with the shared-memory transport and a resident placement the loop has NO host in it -- every rank runs its share as
 * ONE persistent launch (muse_run_device's loop kernel); the workers write their scores as tagged granules to a board in the
 * communicator's shared segment, which every rank's GPU maps (pinned host memory; system-scope 8-byte stores and loads), every
 * rank's stepper polls ALL the scores -- batched sweeps, one PCIe round trip each -- and takes the same step.  Where the runtime
 * lets the ranks map each other's device memory (hipIpc; decided by all ranks together when the communicator is set up) there is a
 * board per GPU in DEVICE memory instead: a worker stores its score into every rank's board (posted writes, xGMI between GPUs) and a
 * stepper polls its own GPU's memory -- no PCIe round trip in an iteration.  Whether the persistent loop runs is decided by all
 * ranks together (the minimum of what each can do); a rank whose workgroups cannot all be resident makes every rank fall back to the
 * host-driven loop, the same bits. */
int muse_run_sharded(muse_ctx* ctx, uint64_t seed, const double* theta0, const muse_run_options* opt, int32_t* niter_out,
                     double* theta_out, double* hist_out, double* gsims_out, muse_info* info_out);
/* The score boards of muse_run_sharded's persistent loop, decided at SET-UP (round 6): a COLLECTIVE call of a shared-memory
 * communicator -- every rank makes it, or the first muse_run_sharded call makes it for them.  After the ranks have mapped each
 * other's boards, a one-wavefront kernel per rank stores a tagged granule into every peer's board -- with the very store the loop
 * kernel's workers use -- and polls its own board for every peer's tag -- with the stepper's load -- for at most a few
 * milliseconds (MUSE_BOARD_HANDSHAKE_MS, default 50).  A rank that did not see every peer makes ALL ranks drop that board: a
 * cross-GPU visibility failure costs milliseconds here instead of a bounded wait inside the first user call.  The boards in
 * device memory are tried first, then the one in pinned host memory; with neither the loop is host-driven (the same bits).
 *   status_out[0]  the board the persistent loop will use: MUSE_BOARD_NONE / _HOST / _DEVICE
 *   status_out[1]  hand-shake of the device boards:  1 every rank saw every peer, 0 some rank did not, -1 not tried (no hipIpc
 *                  mapping, more than 8 ranks, switched off)
 *   status_out[2]  hand-shake of the host board, likewise
 *   status_out[3]  peers THIS rank saw on the device boards, as a bit mask (bit q: rank q)
 *   status_out[4]  ... on the host board
 *   status_out[5]  what the last muse_run_sharded call of this context ran: MUSE_BOARD_NONE (host-driven loop, or no call yet),
 *                  _HOST, _DEVICE
 *   wait_us_out[0..1]  how long this rank's hand-shake kernel polled until its last peer's granule landed (device boards, host
 *                  board; microseconds by the GPU's 100 MHz counter; the bound when it expired; 0 when not tried) */
#define MUSE_BOARD_NONE 0
#define MUSE_BOARD_HOST 1
#define MUSE_BOARD_DEVICE 2
int muse_comm_board_status(muse_ctx* ctx, int status_out[6], double wait_us_out[2]);
/* The sharded form of muse_map_and_score_multi_async: this rank's block of `nmaps` maps in one launch, one exchange for
 * all of them.  The gathered block is [nranks][nmaps][rows_per_rank][ntheta]; info_out [nmaps][n]. */
int muse_map_and_score_multi_gather_async(muse_ctx* ctx, uint64_t seed, int64_t sim_begin, int64_t sim_end,
                                          int include_data, int nmaps, const double* thetas, double atol, int z0_mode,
                                          int64_t rows_per_rank, int result_area);

#ifdef __cplusplus
}
#endif
#endif /* MUSE_HIP_H */
