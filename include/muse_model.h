/*
 * muse_model.h -- the contract of a USER-SUPPLIED elementwise model (MUSE_MODEL_USER of muse_hip.h).
 *
 * The reference takes a model as closures: SimpleMuseProblem(x, sample_x_z, logLike, logPrior), differentiated by AD
 * (src/simple.jl:79-95).  Closures cannot cross a C ABI and nothing differentiates device code here, so a user model
 * is a small C header with three functions that is COMPILED INTO an engine library of its own (`libmuse_hip_model_<name>.so`,
 * museinference.jl_amd/build.py build_model_library; Python: museinference_jl_amd.ElementwiseModel) -- every instantiation
 * of the solver kernel (register/LDS-resident, streaming, workgroup clusters), the sampler, the finite-difference and
 * multi-map launches, the native muse! loop: the whole C ABI of muse_hip.h, for that model.
 *
 * The family.  With theta_k the parameter of block k (muse_hip.h: ntheta contiguous equal blocks of the N elements),
 * n_k its number of elements, sd_k = exp(theta_k / 2) and iv_k = exp(-theta_k):
 *
 *     -logLike(x, z, theta) = 1/2 sum_i [ A(x_i, z_i) + iv_k(i) B(x_i, z_i) ] + 1/2 sum_k n_k theta_k
 *     (x_i, z_i) ~ P(x, z | theta):  a function of sd_k(i) and two independent standard normals
 *
 * i.e. theta_k is the log-variance of ONE Gaussian factor of the elementwise joint density and everything else --
 * the observation function, the other factor -- is arbitrary (non-Gaussian, non-linear).  MUSE_MODEL_FUNNEL is the
 * member A = (x - z)^2, B = z^2, z = sd n1, x = z + n2; MUSE_MODEL_NOISE is A = z^2, B = (x - z)^2, z = n1, x = z + sd n2.
 * The score follows from the family:  d logLike / d theta_k = 1/2 (iv_k sum_{i in k} B(x_i, z_i) - n_k).
 *
 * The header defines (MUSE_MODEL_FN is supplied by the includer: host + device, inlined; plain C, doubles only, every
 * operation an IEEE + - * / sqrt or fma() -- the engine and its CPU checker compile the SAME text without
 * floating-point contraction, which is what makes their results comparable bit for bit):
 *
 *     #define MUSE_MODEL_NAME "cubic"
 *     MUSE_MODEL_FN void   muse_model_sample(double sd, double n1, double n2, double* z, double* x, long i);
 *         the joint draw (src/interface.jl:92-99) of one element from the block's sd and two standard normals
 *     MUSE_MODEL_FN double muse_model_grad(double iv, double x, double z, double* acc, long i);
 *         returns d(-logLike)/dz_i = 1/2 d(A + iv B)/dz and ADDS A(x, z) + iv B(x, z) to *acc -- the running sum of
 *         the elements a thread owns, so the additions may be folded into fma()s as the built-in models do
 *         (src/interface.jl:68-83 with the sign Optim minimises, src/interface.jl:163)
 *     MUSE_MODEL_FN double muse_model_score_term(double x, double z, long i);
 *         B(x, z) (src/interface.jl:41-58: the engine assembles grad_theta logLike from its block sums)
 *
 * `i` is the element's index, 0 <= i < N -- and i >= N for the zero pad element of an odd-length vector and the phantom slots
 * behind it (x = z = 0 there: the functions must stay finite; the accessors of from_source's tables clamp the index): what lets a model
 * depend on the element through constants of its own (a known spectrum P_i: z_i ~ N(0, e^theta P_i); a noise-variance map; a
 * mask), as a table compiled into the header -- `static const double P[N + 1] = {...};`, which
 * ElementwiseModel.from_source(name, source, constants={"P": array}) writes for you -- or as a formula in i: what a closure of
 * the reference's SimpleMuseProblem would capture.  A model that does not need it ignores it.
 *
 * RUN-TIME constants (round 4).  A table compiled into the header fixes its values at build time: a new spectrum is a new
 * header, a 40-second build and a library of its own.  A header that says
 *     #define MUSE_MODEL_NCONST 2                      (before including this file; at most MUSE_MODEL_MAX_CONST)
 * may instead read  muse_const(k, i),  k < MUSE_MODEL_NCONST:  element i of constant vector k as the caller last set it --
 * muse_set_constants(ctx, k, values, count) on the engine (muse_hip.h), mo_set_constants on the CPU checker -- and 1.0 for
 * i >= count (the pad element and the phantom slots, as the compiled tables do) and while the vector has not been set (the
 * engine's contract check evaluates the functions before any constant exists).  The values live in device memory and are read
 * through one pointer per vector, and the pointers travel WITH EVERY LAUNCH (round 5: in its kernel-argument block, which
 * muse_const reads through the kernarg segment pointer) -- contexts of one library with different constants may have launches in
 * flight at the same time; only the host-side evaluation (muse_model_eval, the contract check) goes through one set per process.
 * ElementwiseModel.from_source(name, source, runtime_constants=["P", ...]) writes the #define and an accessor P(i) per name.
 *
 * Requirements the engine checks when a context is created (it evaluates the functions on the host):
 *     muse_model_grad(iv, 0, 0, &acc, N) == 0 leaving acc as it was, and muse_model_score_term(0, 0, N) == 0
 *         (vectors are padded to an even length with one zero element, which must not contribute).
 * Nothing differentiates the header: that muse_model_grad IS the derivative of the objective term and that muse_model_score_term
 * IS its B is the author's statement -- museinference_jl_amd.check_model_consistency(prob, theta) checks both against finite
 * differences of logLike through the problem's own operators (what AD guarantees in the reference, src/simple.jl:84-85).
 *
 * SECOND DERIVATIVES (round 4, optional).  The implicit-differentiation get_H! (src/muse.jl:335-405; muse_implicit_H_* of
 * muse_hip.h) differentiates the gradients once more: the reference does so by nested AD, a header says them.  A header that
 * says   #define MUSE_MODEL_SECOND 1   also defines, with o_i(x, z) = 1/2 [A(x, z) + iv B(x, z)] the element's objective term,
 *     MUSE_MODEL_FN void   muse_model_second(double iv, double x, double z, double* ozz, double* ozx, double* bz, double* bx, long i);
 *         *ozz = d2 o / dz2,  *ozx = d2 o / dz dx,  *bz = dB / dz,  *bx = dB / dx        (finite at x = z = 0, i >= N)
 *     MUSE_MODEL_FN double muse_model_dx_dsd(double sd, double n1, double n2, long i);
 *         d x / d sd of muse_model_sample at fixed normals (the reference re-draws x(theta) from a copy of the rng, src/muse.jl:353-371)
 * from which the engine forms, elementwise (k = the element's block, t_i = dx_i / dtheta_k = 1/2 sd_k dx/dsd),
 *     Hessian_z logLike w        = -ozz_i w_i                      (src/muse.jl:373-379; CG as in the reference although it is diagonal)
 *     d/dtheta_k grad_z logLike  = 1/2 iv_k bz_i                   (:361-365)
 *     d/dtheta_k grad_z logLike(x(theta), zhat, theta0) = -ozx_i t_i      (:366-371)
 *     H1[k][k] = 1/2 iv_k sum_{i in k} bx_i t_i                    (:353-358)
 * (the funnel: ozz = 1 + iv, ozx = -1, bz = 2 z, bx = 0, dx/dsd = n1).  Without MUSE_MODEL_SECOND the implicit entries refuse
 * the model and get_H! runs by finite differences (muse_fd_*: model-agnostic).  muse_model_eval of muse_hip.h evaluates the
 * header's functions on the host for one element -- what check_model_consistency differentiates numerically.
 *
 * TWO PARAMETERS PER BLOCK (round 6: location-type parameters).  In the family above a block has ONE parameter, the log-variance
 * of one Gaussian factor, and the engine knows how it enters: through exp(theta/2) in the draw, exp(-theta) in the objective,
 * 1/2 (iv sum B - n) in the score.  A header that says
 *     #define MUSE_MODEL_PAIR 1                        (before including this file)
 * states all of that itself, for blocks with TWO parameters acting on the same elements -- a mean AND a log-variance, a slope and a
 * scale, ...: ntheta = 2 K (at most MUSE_MAX_THETA = 8), the N elements in K contiguous equal blocks, block k's parameters
 * a_k = theta[k] and b_k = theta[K + k], and with c_k[0..3] FOUR coefficients the header forms from them,
 *
 *     -logLike(x, z, theta) = 1/2 sum_i o(c_k(i); x_i, z_i) + 1/2 sum_k n_k C(a_k, b_k)
 *     (x_i, z_i) ~ P(x, z | theta):  a function of c_k(i)[0], c_k(i)[1] and two independent standard normals
 *
 *     MUSE_MODEL_FN double muse_model_coefs(double a, double b, double* c);
 *         fills c[0..3] and returns C(a, b).  Evaluated on the host, once per theta; exponentials through muse_model_exp(x), the
 *         engine's fixed-sequence exp (the CPU checker evaluates the same sequence), every other operation an IEEE + - * / sqrt, fma
 *     MUSE_MODEL_FN void   muse_model_sample(const double* c, double n1, double n2, double* z, double* x, long i);
 *         the joint draw; may read c[0] and c[1] ONLY (a finite-difference get_H! re-draws at perturbed parameters and
 *         carries just those two per block)
 *     MUSE_MODEL_FN double muse_model_grad(const double* c, double x, double z, double* acc, long i);
 *         returns d(1/2 o)/dz_i and ADDS o(c; x, z) to *acc
 *     MUSE_MODEL_FN void   muse_model_score_terms(const double* c, double x, double z, double* t0, double* t1, long i);
 *         the element's terms of the block's TWO sums S0 = sum t0, S1 = sum t1
 *     MUSE_MODEL_FN void   muse_model_score(const double* c, double S0, double S1, double n, double* ga, double* gb);
 *         *ga = d logLike / d a_k, *gb = d logLike / d b_k from the block's coefficients, sums and element count
 *         (the d/d theta of the constant term included)
 * The pad element of an odd-length vector and the phantom slots behind it are given ZERO coefficients (a location parameter has a
 * gradient at x = z = 0): with c = {0, 0, 0, 0} and x = z = 0, muse_model_grad must return 0 leaving *acc as it was and
 * muse_model_score_terms must give 0, 0 -- the engine checks that when a context is created.
 * models/normal_mean_var.h is the shipped member: z_i ~ N(mu_k, e^tau_k), x_i ~ N(z_i, 1); c = {mu, e^(tau/2), e^-tau, 0}, C = tau,
 * o = (x - z)^2 + e^-tau (z - mu)^2, t0 = z - mu, t1 = (z - mu)^2, ga = e^-tau S0, gb = 1/2 (e^-tau S1 - n).
 * Everything of the C ABI applies -- batched maps in every placement (register/LDS resident, streaming, workgroup clusters, an
 * element split), multi-map launches, lanes, the native muse! loops (muse_run, muse_run_device: the step between two iterations
 * calls muse_model_coefs on the device -- the same statements as on the host, the same bits --, muse_run_sharded), both exchanges
 * between ranks, the finite-difference get_H! -- except the implicit-differentiation get_H! and more than MUSE_MAX_THETA parameters.
 */
#ifndef MUSE_MODEL_H
#define MUSE_MODEL_H
#ifndef MUSE_MODEL_FN
#define MUSE_MODEL_FN static inline
#endif
#define MUSE_MODEL_MAX_CONST 4
#ifndef muse_model_exp
/* The exponential a header of the two-parameter family forms its coefficients with.  The engine and its CPU checker define it
 * before they include the header -- one fixed sequence of IEEE operations, the same bits on host, device and checker; a plain
 * compile of the header by itself (a syntax check, an editor) gets libm's. */
#include <math.h>
#define muse_model_exp(x) exp(x)
#endif
#ifdef MUSE_MODEL_NCONST
#if MUSE_MODEL_NCONST < 1 || MUSE_MODEL_NCONST > MUSE_MODEL_MAX_CONST
#error "MUSE_MODEL_NCONST must be in [1, MUSE_MODEL_MAX_CONST]"
#endif
/* the vectors: defined once per program that compiles a model header (the engine's muse_engine.cpp / muse_kernels.hip,
 * the checker's muse_oracle.c) */
#ifdef __cplusplus
extern "C" {
#endif
extern const double* muse_host_consts[MUSE_MODEL_MAX_CONST];
extern long muse_host_const_len[MUSE_MODEL_MAX_CONST];
#ifdef __cplusplus
}
#endif
MUSE_MODEL_FN double muse_const(int k, long i) {
#if defined(__HIP_DEVICE_COMPILE__)
    /* the launch's own vectors: every kernel of the engine takes its argument block (csrc/args.hpp, BatchArgs) as its FIRST
     * parameter, and MUSE_KERNARG_CONSTS (csrc/user_model.hpp) is the offset of {pointers[4], lengths[4]} in it -- uniform
     * scalar loads from the kernel-argument segment, no state shared between launches */
    const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr() + MUSE_KERNARG_CONSTS;
    const long n = ((const long*)(ka + MUSE_MODEL_MAX_CONST * sizeof(const double*)))[k];
    return i < n ? ((const double* const*)ka)[k][i] : 1.0;
#else
    const long n = muse_host_const_len[k];
    return i < n ? muse_host_consts[k][i] : 1.0;
#endif
}
#endif
#endif
