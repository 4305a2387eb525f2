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
 * Requirements the engine checks when a context is created (it evaluates the functions on the host):
 *     muse_model_grad(iv, 0, 0, &acc, N) == 0 leaving acc as it was, and muse_model_score_term(0, 0, N) == 0
 *         (vectors are padded to an even length with one zero element, which must not contribute).
 * Nothing differentiates the header: that muse_model_grad IS the derivative of the objective term and that muse_model_score_term
 * IS its B is the author's statement -- museinference_jl_amd.check_model_consistency(prob, theta) checks both against finite
 * differences of logLike through the problem's own operators (what AD guarantees in the reference, src/simple.jl:84-85).
 * Not supported for user models: the implicit-differentiation get_H! (muse_implicit_H_*: it needs second
 * derivatives) -- the finite-difference branch (muse_fd_*) is model-agnostic.
 */
#ifndef MUSE_MODEL_H
#define MUSE_MODEL_H
#ifndef MUSE_MODEL_FN
#define MUSE_MODEL_FN static inline
#endif
#endif
