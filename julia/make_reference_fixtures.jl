# make_reference_fixtures.jl -- the out-of-band pin of this repository's CPU oracle against the REAL package.
#
# The build image has no julia, so nothing here has been executed by the authors of this repository; it is the one-command
# job for anyone who has Julia and marius311/MuseInference.jl (v0.2.4) installed:
#
#   julia --project=<env with MuseInference, Optim, NPZ> julia/make_reference_fixtures.jl \
#         tests/golden/reference_inputs.npz tests/golden/reference_outputs.npz
#
# It reads the inputs this repository commits (tests/golden/make_reference_inputs.py: x, z0, theta, atol per MAP case;
# data, injected standard normals and options for one muse! run), drives the reference's own `ẑ_at_θ` (Optim LBFGS,
# src/interface.jl:162-166) and `muse` / `get_J!` / `get_H!` (src/muse.jl) on SimpleMuseProblem closures of the three
# models, and writes what tests/test_reference_fixtures.py compares the oracle with: ẑ, Optim's iteration / f_calls
# counts and the theta trajectory.  Randomness is INJECTED: Julia's randn streams cannot be reproduced outside Julia, so
# `sample_x_z` reads the normals of "simulation k" from the input file; the rng that muse! splits (src/util.jl:87-92) is
# a counter whose children carry k.
using MuseInference, NPZ, Random, LinearAlgebra, Optim, Statistics
using MuseInference: SimpleMuseProblem, ẑ_at_θ, muse, get_J!, get_H!

# ---- an rng that only carries the simulation index: split_rng seeds child i with the i-th UInt32 drawn from a copy
mutable struct IndexRNG <: AbstractRNG
    sim::Int     # 0 = the un-split master stream, i = the i-th child of split_rng
    draws::Int
end
Base.copy(r::IndexRNG) = IndexRNG(r.sim, r.draws)
Random.rand(r::IndexRNG, ::Random.SamplerType{UInt32}) = (r.draws += 1; UInt32(r.draws))
Random.seed!(r::IndexRNG, s::Integer) = (r.sim = Int(s); r.draws = 0; r)

# ---- the three compiled-in models of include/muse_hip.h as closures (theta: a Number for ntheta = 1, else a Vector)
blocks(N, nθ) = [div((k - 1) * N + nθ - 1, nθ) + 1 : div(k * N + nθ - 1, nθ) for k in 1:nθ]   # bnd[k] = ceil(k N / nθ)
θvec(θ) = θ isa Number ? [θ] : collect(θ)
function variances(N, θ)
    v = zeros(eltype(θvec(θ)), N)
    for (k, r) in enumerate(blocks(N, length(θvec(θ)))); v[r] .= exp(θvec(θ)[k]); end
    v
end
Az(z) = 0.25 .* circshift(z, 1) .+ 0.5 .* z .+ 0.25 .* circshift(z, -1)                 # periodic (1/4, 1/2, 1/4)
function make_logLike(model, N)
    if model == 0        # funnel: z_i ~ N(0, e^θ_k), x_i ~ N(z_i, 1)
        (x, z, θ) -> -(1//2) * (sum((x .- z).^2) + sum(z.^2 ./ variances(N, θ)) + sum(log.(variances(N, θ))))
    elseif model == 1    # noise: z_i ~ N(0,1), x_i ~ N(z_i, e^θ)
        (x, z, θ) -> -(1//2) * (sum(z.^2) + sum((x .- z).^2) / exp(θvec(θ)[1]) + N * θvec(θ)[1])
    else                 # smooth: z as funnel, x = A z + n
        (x, z, θ) -> -(1//2) * (sum((x .- Az(z)).^2) + sum(z.^2 ./ variances(N, θ)) + sum(log.(variances(N, θ))))
    end
end
function make_sample(model, N, n1, n2)   # n1, n2: [nstreams, N]; row 1 = master stream, row 1+k = simulation k
    function (rng, θ)
        a, b = n1[rng.sim + 1, :], n2[rng.sim + 1, :]
        if model == 1
            z = a; x = z .+ exp(θvec(θ)[1] / 2) .* b
        elseif model == 0
            z = sqrt.(variances(N, θ)) .* a; x = z .+ b
        else
            z = sqrt.(variances(N, θ)) .* a; x = Az(z) .+ b
        end
        (; x, z)
    end
end

inp = npzread(ARGS[1])
out = Dict{String,Any}()

# ---- MAP cases: ẑ_at_θ(prob, x, z₀, θ; ∇z_logLike_atol) with the reference's default (Optim LBFGS + HagerZhang)
ncases = Int(inp["ncases"])
for c in 0:ncases-1
    model, N, nθ = Int(inp["case$(c)_model"]), Int(inp["case$(c)_N"]), Int(inp["case$(c)_ntheta"])
    θ = nθ == 1 ? inp["case$(c)_theta"][1] : inp["case$(c)_theta"]
    x, z₀, atol = inp["case$(c)_x"], inp["case$(c)_z0"], inp["case$(c)_atol"]
    prob = SimpleMuseProblem(x, (rng, θ) -> error("not sampled here"), make_logLike(model, N))
    ẑ, soln = ẑ_at_θ(prob, x, z₀, θ; ∇z_logLike_atol = atol)
    out["case$(c)_zhat"] = ẑ
    out["case$(c)_counts"] = Float64[Optim.iterations(soln), Optim.f_calls(soln), Optim.g_calls(soln), Optim.converged(soln)]
    out["case$(c)_fmin"] = Float64[Optim.minimum(soln)]
    out["case$(c)_score"] = θvec(MuseInference.∇θ_logLike(prob, x, ẑ, θ))
end

# ---- one muse! run with injected normals: trajectory, J, H, Σ
let N = Int(inp["run_N"]), nθ = Int(inp["run_ntheta"]), model = Int(inp["run_model"]), nsims = Int(inp["run_nsims"])
    σp = inp["run_prior_sigma"]
    logPrior = θ -> -sum(θvec(θ).^2) / (2 * σp^2)
    prob = SimpleMuseProblem(inp["run_x"], make_sample(model, N, inp["run_n1"], inp["run_n2"]), make_logLike(model, N), logPrior)
    θ₀ = nθ == 1 ? inp["run_theta0"][1] : inp["run_theta0"]
    result = muse(prob, θ₀; rng = IndexRNG(0, 0), nsims = nsims, maxsteps = Int(inp["run_maxsteps"]),
                  θ_rtol = inp["run_theta_rtol"], ∇z_logLike_atol = inp["run_atol"], α = inp["run_alpha"], get_covariance = true)
    out["run_thetas"] = reduce(hcat, [θvec(h.θ) for h in result.history])'
    out["run_g_like"] = reduce(hcat, [θvec(h.g_like′) for h in result.history])'
    out["run_theta"] = θvec(result.θ)
    out["run_J"] = Matrix(reshape(collect(result.J), nθ, nθ))
    out["run_H"] = Matrix(reshape(collect(result.H), nθ, nθ))
    out["run_Sigma"] = Matrix(reshape(collect(result.Σ), nθ, nθ))
    out["run_gs"] = reduce(hcat, [θvec(g) for g in result.gs])'
    out["run_map_iterations"] = Float64[Optim.iterations(h) for h in result.history[end].ẑ_history_sims]
end

npzwrite(ARGS[2], out)
println("wrote ", ARGS[2])
