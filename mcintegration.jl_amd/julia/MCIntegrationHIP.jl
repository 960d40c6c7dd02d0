# MCIntegrationHIP.jl -- thin `ccall` binding of libmci_hip.so (include/mci.h) that re-creates the
# reference's API names for the :vegas / :vegasmc / :mcmc path (reference src/MCIntegration.jl:20-47).
#
# NOTE: `julia` is not available in the build image, so this file is syntax-reviewed only; it is the
# reference-side binding INTEGRATION.md describes.  Everything numerical happens in the library.
module MCIntegrationHIP

export integrate, Configuration, Continuous, Discrete, CompositeVar, FermiK, Result, Integrand, Measure, bin_by, report

const libmci = get(ENV, "MCI_HIP_LIB", joinpath(@__DIR__, "..", "lib", "libmci_hip.so"))
const MaxOrder = 16                         # reference src/distribution/distribution.jl:59
const MCI_CONTINUOUS, MCI_DISCRETE, MCI_FERMIK = Int32(0), Int32(1), Int32(2)
const SOLVER = Dict(:vegas => Int32(0), :vegasmc => Int32(1), :mcmc => Int32(2))

struct MCIError <: Exception
    code::Int
    msg::String
end
function check(rc::Integer)
    rc == 0 && return
    throw(MCIError(rc, unsafe_string(ccall((:mci_last_error, libmci), Cstring, ()))))
end

# ---- variables (reference src/distribution/variable.jl:87-99, :272-284, :397-404) ---------------------
mutable struct Continuous
    lower::Float64; upper::Float64; size::Int; offset::Int; alpha::Float64; adapt::Bool; ninc::Int
    grid::Union{Nothing,Vector{Float64}}
end
Continuous(lower::Real, upper::Real, size=MaxOrder; offset=0, alpha=2.0, adapt=true, ninc=1000, grid=nothing) =
    Continuous(lower, upper, size + 1, offset, alpha, adapt, grid === nothing ? ninc : length(grid), grid)

mutable struct Discrete
    lower::Int; upper::Int; size::Int; offset::Int; alpha::Float64; adapt::Bool
    distribution::Union{Nothing,Vector{Float64}}
end
Discrete(lower::Int, upper::Int, size=MaxOrder; distribution=nothing, offset=0, alpha=2.0, adapt=true) =
    Discrete(lower, upper, size + 1, offset, alpha, adapt, distribution)

mutable struct FermiK                         # reference src/distribution/variable.jl:1-20 (solver = :mcmc only)
    dim::Int; kF::Float64; dk::Float64; maxK::Float64; size::Int; offset::Int
end
FermiK(dim, kF, dk, maxK, size=MaxOrder; offset=0) = FermiK(dim, kF, dk, maxK, size + 1, offset)

struct CompositeVar
    vars::Tuple
    adapt::Bool; offset::Int; size::Int
end
function CompositeVar(vargs...; adapt=true, offset=0, size=MaxOrder)
    for v in vargs
        v.adapt = adapt; v.offset = offset          # variable.jl:419-420
    end
    CompositeVar(Tuple(vargs), adapt, offset, size)
end
Continuous(bounds::Union{AbstractVector,Tuple}, size=MaxOrder; offset=0, alpha=2.0, adapt=true) =   # variable.jl:174-187
    CompositeVar((Continuous(b[1], b[2], size; offset, alpha, adapt, ninc=1000) for b in bounds)...; adapt, offset, size)

leaves(v) = (v,)
leaves(v::CompositeVar) = v.vars

# ---- integrand: HIP C++ source instead of a Julia closure (INTEGRATION.md) ---------------------------
struct Integrand
    body::String
    userdata::Vector{Float64}
end
Integrand(body::AbstractString) = Integrand(String(body), Float64[])
struct bin_by; pool::Int; end                # measure of example/bubble.jl:81-84 (1-based pool index)
struct Measure; body::String; end            # user measure as HIP C++ source: x, rw, ud, idx, obs_add(k, v)  (include/mci.h)

# ---- C structs of include/mci.h ------------------------------------------------------------------------
struct LeafDesc
    kind::Int32; pool::Int32; lower::Float64; upper::Float64; npoints::Int32; alpha::Float64; adapt::Int32
    init::Ptr{Float64}
end
struct ProblemDesc
    nleaf::Int32; leaves::Ptr{LeafDesc}; npool::Int32; nintegrand::Int32
    dof::Ptr{Int32}; obs_nbin::Ptr{Int32}; obs_bin_draw::Ptr{Int32}
    neighbor_offsets::Ptr{Int32}; neighbor_list::Ptr{Int32}     # CSR, 0-based; both NULL = the reference default
    ncomp::Int32                                                # 1 Float64, 2 ComplexF64 (`type` kwarg)
end
struct IntegrateArgs
    solver::Int32; neval::Int64; niter::Int32; block::Int64; ignore::Int32; adapt::Int32; gamma::Float64
    measurefreq::Int64; seed::UInt64; nchain::Int64; first_iteration::Int32
    thermal_ratio::Float64; reweight_goal::Ptr{Float64}
end
mutable struct ResultC
    niter::Int32; nobs::Int32
    iter_mean::Ptr{Float64}; iter_std::Ptr{Float64}; mean::Ptr{Float64}; stdev::Ptr{Float64}; chi2::Ptr{Float64}
    neval::Int64; seconds::Float64
end

const _ctx = Ref{Ptr{Cvoid}}(C_NULL)
function context(device=0)
    if _ctx[] == C_NULL
        p = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:mci_ctx_create, libmci), Cint, (Int32, Ptr{Ptr{Cvoid}}), device, p))
        _ctx[] = p[]
    end
    _ctx[]
end

# Configuration(; var, dof, obs, seed, userdata)   reference src/configuration.jl:105-194
mutable struct Configuration
    var::Tuple
    dof::Vector{Vector{Int}}
    N::Int
    obs_nbin::Vector{Int}
    seed::Int
    userdata
    iterations_done::Int
    problem::Ptr{Cvoid}
    key
    neighbor::Union{Nothing,Vector{Vector{Int}}}    # 1-based like the reference; nothing = default (configuration.jl:203-208)
    ncomp::Int                                      # 2 for type=ComplexF64 (configuration.jl:108)
end
function _neighbor(neighbor, Nd)                    # reference src/configuration.jl:201-227
    neighbor === nothing && return nothing
    if neighbor isa Vector{Tuple{Int,Int}}          # undirected edge list
        adj = [Int[] for _ in 1:Nd]
        for (a, b) in neighbor
            b in adj[a] || push!(adj[a], b)
            a in adj[b] || push!(adj[b], a)
        end
        return [sort(a) for a in adj]
    end
    @assert length(neighbor) == Nd "$Nd elements are expected for neighbor=$neighbor"
    return [collect(Int, n) for n in neighbor]
end
function Configuration(; var=(Continuous(0.0, 1.0),), dof=nothing, obs=nothing, seed=rand(1:1000000), userdata=nothing, neighbor=nothing,
                       type=Float64, kwargs...)
    var = var isa Tuple ? var : (var isa AbstractVector ? Tuple(var) : (var,))          # :116-122
    if dof === nothing
        dof = [ones(Int, length(var))]
    elseif dof isa Int
        @assert length(var) == 1 "Only one type of variable is allowed when dof is an integer"
        dof = [[dof]]
    elseif dof isa AbstractMatrix
        dof = [dof[:, i] for i in 1:size(dof, 2)]
    elseif eltype(dof) <: Int
        dof = [[d] for d in dof]
    else
        dof = [collect(Int, d) for d in dof]
    end
    ncomp = type <: Complex ? 2 : 1
    obs === nothing && (obs = zeros(type, length(dof)))
    @assert length(obs) == length(dof) "The number of observables should be equal to the number of integrands"
    Configuration(var, dof, length(dof), [length(o) * ncomp for o in obs], seed, userdata, 0, C_NULL, nothing,
                  _neighbor(neighbor, length(dof) + 1), ncomp)
end

function draw_index(c::Configuration, pool::Int)          # 0-based flat draw of (pool, slot 1, leaf 1)
    k = 0
    for (vi, v) in enumerate(c.var)
        vi == pool && return k
        k += maximum(d[vi] for d in c.dof) * (v isa FermiK ? v.dim : length(leaves(v)))
    end
    error("pool $pool out of range")
end

function bind!(c::Configuration, f::Integrand, measure)
    key = (f.body, f.userdata, measure, c.neighbor)
    (c.problem != C_NULL && c.key == key) && return c.problem
    descs = LeafDesc[]
    keep = Any[]
    for (vi, v) in enumerate(c.var), lf in leaves(v)
        if lf isa Continuous
            init = lf.grid === nothing ? Ptr{Float64}(C_NULL) : (push!(keep, lf.grid); pointer(lf.grid))
            push!(descs, LeafDesc(MCI_CONTINUOUS, vi - 1, lf.lower, lf.upper, lf.ninc, lf.alpha, lf.adapt, init))
        elseif lf isa FermiK                       # lower = kF, upper = dk, npoints = dim, alpha = maxK (include/mci.h)
            push!(descs, LeafDesc(MCI_FERMIK, vi - 1, lf.kF, lf.dk, lf.dim, lf.maxK, false, Ptr{Float64}(C_NULL)))
        else
            init = lf.distribution === nothing ? Ptr{Float64}(C_NULL) : (push!(keep, lf.distribution); pointer(lf.distribution))
            push!(descs, LeafDesc(MCI_DISCRETE, vi - 1, lf.lower, lf.upper, 0, lf.alpha, lf.adapt, init))
        end
    end
    dof = Int32[d[vi] for d in c.dof for vi in 1:length(c.var)]
    onb = Int32.(c.obs_nbin)
    obd = Int32[(measure isa bin_by && n > 1) ? draw_index(c, measure.pool) : -1 for n in c.obs_nbin]
    nboff, nblist = Int32[], Int32[]
    if c.neighbor !== nothing
        nboff = Int32.(cumsum([0; length.(c.neighbor)]))
        nblist = Int32[j - 1 for n in c.neighbor for j in n]
    end
    p = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve descs dof onb obd keep nboff nblist begin
        desc = Ref(ProblemDesc(length(descs), pointer(descs), length(c.var), c.N, pointer(dof), pointer(onb), pointer(obd),
                               c.neighbor === nothing ? Ptr{Int32}(C_NULL) : pointer(nboff),
                               c.neighbor === nothing ? Ptr{Int32}(C_NULL) : pointer(nblist), c.ncomp))
        check(ccall((:mci_problem_create, libmci), Cint, (Ptr{Cvoid}, Ptr{ProblemDesc}, Ptr{Ptr{Cvoid}}), context(), desc, p))
    end
    check(ccall((:mci_set_integrand_source, libmci), Cint, (Ptr{Cvoid}, Cstring, Ptr{Float64}, Int32),
                p[], f.body, f.userdata, length(f.userdata)))
    measure isa Measure && check(ccall((:mci_set_measure_source, libmci), Cint, (Ptr{Cvoid}, Cstring), p[], measure.body))
    c.problem != C_NULL && ccall((:mci_problem_destroy, libmci), Cint, (Ptr{Cvoid},), c.problem)
    c.problem, c.key = p[], key
    p[]
end

# ---- Julia closures as integrands: the host "batch callback" slow path (mci_set_integrand_host) ----------------------
# f(x, config) is called ONCE per launch with x[k] = the vector of draw k over the n samples of the batch (several
# variable types: a tuple of per-pool matrices), and returns a vector (or a tuple of vectors, one per integrand).
const _closures = Dict{Ptr{Cvoid},Any}()      # problem => (f, config): keeps them rooted
function _host_trampoline(x::Ptr{Float64}, w::Ptr{Float64}, n::Int64, ndraw::Int32, nw::Int32, user::Ptr{Cvoid})::Cint
    try
        f, c = _closures[user]
        X = unsafe_wrap(Array, x, (Int(n), Int(ndraw)))           # column k = draw k (draw-major in memory)
        W = unsafe_wrap(Array, w, (Int(n), Int(nw)))
        cols = [view(X, :, k) for k in 1:ndraw]
        out = f(cols, c)
        out isa Tuple || (out = (out,))
        for (i, o) in enumerate(out)
            if c.ncomp == 2
                W[:, 2i-1] .= real.(o); W[:, 2i] .= imag.(o)
            else
                W[:, i] .= o
            end
        end
        return Cint(0)
    catch err
        @error "host integrand failed" err
        return Cint(1)
    end
end
function bind_host!(c::Configuration, f::Function)
    prob = bind!(c, Integrand("", Float64[]), nothing)
    _closures[prob] = (f, c)
    cb = @cfunction(_host_trampoline, Cint, (Ptr{Float64}, Ptr{Float64}, Int64, Int32, Int32, Ptr{Cvoid}))
    check(ccall((:mci_set_integrand_host, libmci), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), prob, cb, prob))
    prob
end

# Result   reference src/statistics.jl:16-63
struct Result
    mean::Vector{Float64}; stdev::Vector{Float64}; chi2::Vector{Float64}
    neval::Int; ignore::Int; config::Configuration
    iter_mean::Matrix{Float64}; iter_std::Matrix{Float64}
end
function Base.show(io::IO, r::Result)
    for i in eachindex(r.mean)
        println(io, "Integral $i = $(r.mean[i]) ± $(r.stdev[i])   (reduced chi2 = $(round(r.chi2[i], sigdigits=3)))")
    end
end
report(r::Result) = show(stdout, r)

"""
    integrate(integrand::Integrand; solver=:vegasmc, config=nothing, neval=1e4, niter=10, block=16, gamma=1.0,
              adapt=true, ignore=adapt ? 1 : 0, measure=nothing, measurefreq=1, kwargs...)

Same keywords as the reference (src/main.jl:71-90); the loop of src/main.jl:142-218 runs inside
`mci_integrate` on the GPU.  Unknown keywords go to `Configuration` (src/main.jl:95-97).
"""
function integrate(integrand::Union{Integrand,AbstractString,Function}; solver::Symbol=:vegasmc, config=nothing, neval=1e4, niter=10,
                   block=16, gamma=1.0, adapt=true, ignore::Int=adapt ? 1 : 0, measure=nothing, measurefreq::Int=1,
                   thermal_ratio=0.1, reweight_goal::Union{Vector{Float64},Nothing}=nothing,
                   nchain=0, print=-1, verbose=-1, kwargs...)
    haskey(SOLVER, solver) || error("Solver $solver is not supported!")                  # main.jl:263
    config === nothing && (config = Configuration(; kwargs...))                          # main.jl:95-97
    if integrand isa Function                      # a Julia closure: host batch-callback path, :vegas only
        solver == :vegas || error("a closure integrand runs with solver=:vegas only; pass device source for :vegasmc / :mcmc")
        prob = bind_host!(config, integrand)
    else
        f = integrand isa Integrand ? integrand : Integrand(String(integrand), config.userdata === nothing ? Float64[] : Float64.(config.userdata))
        prob = bind!(config, f, measure)
    end
    nobs = sum(config.obs_nbin)
    im, ie = zeros(nobs, niter), zeros(nobs, niter)           # row-major [niter][nobs] on the C side
    m, s, c2 = zeros(nobs), zeros(nobs), zeros(nobs)
    goal = reweight_goal === nothing ? Float64[] : reweight_goal
    args = Ref(IntegrateArgs(SOLVER[solver], Int64(neval), niter, block, ignore, adapt, gamma, measurefreq, UInt64(config.seed),
                             nchain, config.iterations_done, thermal_ratio,
                             reweight_goal === nothing ? Ptr{Float64}(C_NULL) : pointer(goal)))
    res = ResultC(niter, nobs, pointer(im), pointer(ie), pointer(m), pointer(s), pointer(c2), 0, 0.0)
    GC.@preserve im ie m s c2 goal check(ccall((:mci_integrate, libmci), Cint, (Ptr{Cvoid}, Ptr{IntegrateArgs}, Ref{ResultC}), prob, args, res))
    config.iterations_done += niter
    r = Result(m, s, c2, res.neval, ignore, config, permutedims(im), permutedims(ie))
    max(print, verbose) >= 0 && report(r)
    r
end

end # module
