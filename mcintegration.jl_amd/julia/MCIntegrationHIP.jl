# MCIntegrationHIP.jl -- thin `ccall` binding of libmci_hip.so (include/mci.h) that re-creates the
# reference's API names for the :vegas / :vegasmc / :mcmc path (reference src/MCIntegration.jl:20-47).
#
# NOTE: `julia` is not available in the build image, so this file is syntax-reviewed only; it is the
# reference-side binding INTEGRATION.md describes.  Everything numerical happens in the library; the C struct mirrors below
# are checked field by field against include/mci.h and the ctypes binding by tests/test_binding_layouts.py.
module MCIntegrationHIP

using Printf
using Dates

export integrate, Configuration, Continuous, Discrete, CompositeVar, FermiK, Result, Integrand, Measure, bin_by, report,
       average, init_comm!, save, load!, trace_integrand, TraceError

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
    visited::Ptr{Float64}
    correlated::Int32          # out: 1 = stdev is the block-lineage error of carried chains (mci_lineage_sums), else statistics.jl:198
    warmup::Int32              # out: launches run again instead of being counted (automatic :mcmc chain lengths, mci_mcmc_launch_valid)
end

const _ctx = Dict{Int,Ptr{Cvoid}}()              # one mci_ctx (HIP stream + RCCL communicator) per device
function context(device::Integer=0)
    get!(_ctx, Int(device)) do
        p = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:mci_ctx_create, libmci), Cint, (Int32, Ptr{Ptr{Cvoid}}), device, p))
        p[]
    end
end

# ---- workers: the reference's MPI.Init + MPIreduce/MPIbcast (src/main.jl:113-118, :177-188, src/utility/parallel.jl:25-99)
# become ONE RCCL all-reduce per iteration inside the library.  Bootstrap: rank 0 creates the 128-byte id, the host ships it.
const _comm = Ref((rank=0, size=1, device=0))
"""
    init_comm!(rank, nranks, bcast!; device=rank)

`bcast!(buf::Vector{UInt8})` must overwrite `buf` on every rank with rank 0's content, e.g. with MPI.jl
`buf -> MPI.Bcast!(buf, 0, MPI.COMM_WORLD)` (the call the reference itself uses, src/utility/parallel.jl:93).
"""
function init_comm!(rank::Integer, nranks::Integer, bcast!::Function; device::Integer=rank)
    id = zeros(UInt8, 128)
    rank == 0 && check(ccall((:mci_comm_unique_id, libmci), Cint, (Ptr{UInt8},), id))
    bcast!(id)
    check(ccall((:mci_comm_init, libmci), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{UInt8}), context(device), rank, nranks, id))
    r, n = Ref{Int32}(0), Ref{Int32}(0)
    check(ccall((:mci_comm_rank, libmci), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}), context(device), r, n))
    @assert (r[], n[]) == (rank, nranks)
    _comm[] = (rank=Int(rank), size=Int(nranks), device=Int(device))
    nothing
end
"""
    init_comm!(comm)        # comm::MPI.Comm -- one rank per GPU of the node; MPI.jl is the caller's dependency, not this module's
"""
function init_comm!(comm; device=nothing)
    MPI = parentmodule(typeof(comm))
    rank, n = MPI.Comm_rank(comm), MPI.Comm_size(comm)
    init_comm!(rank, n, buf -> MPI.Bcast!(buf, 0, comm); device=device === nothing ? rank : device)
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
    device::Int
    neval::Int                                      # evaluations per iteration of the last integrate call (configuration.jl:49)
    pending_state::Union{Nothing,String}            # MCISTATE file to load once the problem exists (load!)
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
                  _neighbor(neighbor, length(dof) + 1), ncomp, _comm[].device, 0, nothing)
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
        check(ccall((:mci_problem_create, libmci), Cint, (Ptr{Cvoid}, Ptr{ProblemDesc}, Ptr{Ptr{Cvoid}}), context(c.device), desc, p))
    end
    check(ccall((:mci_set_integrand_source, libmci), Cint, (Ptr{Cvoid}, Cstring, Ptr{Float64}, Int32),
                p[], f.body, f.userdata, length(f.userdata)))
    measure isa Measure && check(ccall((:mci_set_measure_source, libmci), Cint, (Ptr{Cvoid}, Cstring), p[], measure.body))
    if c.problem != C_NULL                       # a new integrand on a trained configuration keeps what was learned
        tmp = tempname()
        check(ccall((:mci_save_state, libmci), Cint, (Ptr{Cvoid}, Cstring), c.problem, tmp))
        check(ccall((:mci_load_state, libmci), Cint, (Ptr{Cvoid}, Cstring), p[], tmp))
        rm(tmp; force=true)
        ccall((:mci_problem_destroy, libmci), Cint, (Ptr{Cvoid},), c.problem)
    end
    c.problem, c.key = p[], key
    if c.pending_state !== nothing
        check(ccall((:mci_load_state, libmci), Cint, (Ptr{Cvoid}, Cstring), p[], c.pending_state))
        c.pending_state = nothing
    end
    p[]
end

# ---- resume across processes (SURVEY 8f2): grids, distributions, reweight <-> MCISTATE file (include/mci.h) ----
function save(c::Configuration, path::AbstractString)
    c.problem == C_NULL && error("nothing trained yet: run integrate(...) first")
    check(ccall((:mci_save_state, libmci), Cint, (Ptr{Cvoid}, Cstring), c.problem, path))
    path
end
function load!(c::Configuration, path::AbstractString)
    if c.problem == C_NULL
        c.pending_state = String(path)       # applied when integrate(...; config=c) creates the problem
    else
        check(ccall((:mci_load_state, libmci), Cint, (Ptr{Cvoid}, Cstring), c.problem, path))
    end
    c
end

# config.var[i].grid / .distribution / config.reweight / config.visited / propose / accept as the library holds them
function grid(c::Configuration, leaf::Integer, npoints::Integer=1000)      # leaf: 1-based flat leaf index
    g = zeros(npoints)
    check(ccall((:mci_get_grid, libmci), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}, Int32), c.problem, leaf - 1, g, npoints))
    g
end
function reweight(c::Configuration)
    r = zeros(c.N + 1)
    check(ccall((:mci_get_reweight, libmci), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int32), c.problem, r, c.N + 1))
    r
end
function acceptance(c::Configuration)        # (propose, accept), each [3, Nd, max(Nd, Nv)] like configuration.jl:185-186
    Nd = c.N + 1
    M = max(Nd, length(c.var))
    pr, ac = zeros(3 * Nd * M), zeros(3 * Nd * M)
    check(ccall((:mci_get_acceptance, libmci), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int32), c.problem, pr, ac, 3 * Nd * M))
    shape(v) = permutedims(reshape(v, M, Nd, 3), (3, 2, 1))                 # the library is row-major [update][integrand][target]
    shape(pr), shape(ac)
end
function visited(c::Configuration)
    n = Ref{Int32}(0); no = Ref{Int32}(0); ps = Ref{Int64}(0); tm = Ref{Int32}(0); lds = Ref{Int64}(0)
    check(ccall((:mci_problem_info, libmci), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{Int64}, Ptr{Int32}, Ptr{Int64}), c.problem, n, no, ps, tm, lds))
    packed = zeros(ps[])
    check(ccall((:mci_get_packed, libmci), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64), c.problem, packed, ps[]))
    packed[2*no[]+3:2*no[]+3+c.N]
end

# ---- Julia closures traced into device source (the counterpart of mcintegration_jl_amd/trace.py) ---------------------------------------
# The reference's user API is a closure, inlined by Julia's JIT into its loop (vegas/montecarlo.jl:140-144).  Here the closure is run
# ONCE on symbolic draws -- `Sym <: Real`, so `x[1]^2 + x[2]^2`, `exp(-x[1])`, `ifelse(x[1] < 0.5, a, b)` work as they are -- and what it
# computes is written out as the HIP C++ body the kernels are JIT-compiled around: same node numbering, same grammar (C_FORMAT /
# C_OPERANDS, compared with trace.py by tests/test_binding_layouts.py), so the same closure gives the same text as the Python tracer
# and hits the same kernel-cache entry.  A closure that cannot be written out (`if x[1] > 0.5`, a call that wants a Float64, complex
# weights) throws TraceError and takes the host batch-callback path below.  Floats the closure captures become `ud[k]` slots: the
# body does not depend on their values (one code object for a parameter sweep).
struct TraceError <: Exception; msg::String; end
mutable struct Tape
    ops::Vector{Symbol}                     # node k (1-based here, written as t<k-1> like the Python tracer's 0-based ids)
    args::Vector{Vector{Any}}               # Int = node id, Float64 = the value of a :const node, Int32 = the index of an :x / :rw / :ud node
    index::Dict{Any,Int}
    params::Vector{Float64}
end
Tape() = Tape(Symbol[], Vector{Any}[], Dict{Any,Int}(), Float64[])
struct Sym <: Real
    tape::Tape
    id::Int
end
const BOOL_OPS = (:<, :<=, :>, :>=, :(==), :!=, :not, :and, :or)
const C_FORMAT = Dict("+" => "{0} + {1}", "-" => "{0} - {1}", "*" => "{0} * {1}", "/" => "{0} / {1}", "<" => "{0} < {1}", "<=" => "{0} <= {1}",
                      ">" => "{0} > {1}", ">=" => "{0} >= {1}", "==" => "{0} == {1}", "!=" => "{0} != {1}", "neg" => "-{0}", "not" => "!{0}",
                      "and" => "{0} && {1}", "or" => "{0} || {1}", "where" => "{0} ? {1} : {2}")
const C_OPERANDS = Dict("not" => "c", "and" => "cc", "or" => "cc", "where" => "cnn")
const C_FUNCS = (:exp, :log, :sqrt, :sin, :cos, :tan, :tanh, :sinh, :cosh, :asin, :acos, :atan, :log1p, :expm1, :log10, :log2, :exp2, :cbrt,
                 :floor, :ceil)
function node!(t::Tape, op::Symbol, args...)
    key = (op, map(a -> a isa Sym ? a.id : repr(a), args)...)          # (repr: 0.0 and -0.0 are two constants)
    id = get!(t.index, key) do
        push!(t.ops, op); push!(t.args, Any[a isa Sym ? a.id : a for a in args])
        length(t.ops)
    end
    Sym(t, id)
end
function constant!(t::Tape, v::Real)
    v isa Bool && (v = Float64(v))
    isfinite(v) || throw(TraceError("non-finite constant $v"))
    node!(t, :const, Float64(v))
end
param!(t::Tape, v::Float64) = (push!(t.params, v); node!(t, :ud, Int32(length(t.params) - 1)))
lift(t::Tape, v) = v isa Sym ? v : v isa Real ? constant!(t, v) : throw(TraceError("cannot use $(typeof(v)) in an integrand expression"))
op(s::Sym) = s.tape.ops[s.id]
is_const(s::Sym, v::Float64) = op(s) === :const && s.tape.args[s.id][1] === v      # (=== on Float64 tells 0.0 from -0.0)
function binary(o::Symbol, a, b)
    t = a isa Sym ? a.tape : b.tape
    a, b = lift(t, a), lift(t, b)
    o === :+ && (is_const(a, 0.0) || is_const(b, 0.0)) && return is_const(a, 0.0) ? b : a
    o === :* && (is_const(a, 1.0) || is_const(b, 1.0)) && return is_const(a, 1.0) ? b : a
    o === :- && is_const(b, 0.0) && return a
    o === :/ && is_const(b, 1.0) && return a
    node!(t, o, a, b)
end
for o in (:+, :-, :*, :/)
    @eval Base.$o(a::Sym, b::Sym) = binary($(QuoteNode(o)), a, b)
    @eval Base.$o(a::Sym, b::Real) = binary($(QuoteNode(o)), a, b)
    @eval Base.$o(a::Real, b::Sym) = binary($(QuoteNode(o)), a, b)
end
for o in (:<, :<=, :>, :>=, :(==), :!=)                                    # (> >= != too, so that the text is the Python tracer's: Julia's own are < <= == rewritten)
    @eval Base.$o(a::Sym, b::Sym) = binary($(QuoteNode(o)), a, b)
    @eval Base.$o(a::Sym, b::Real) = binary($(QuoteNode(o)), a, b)
    @eval Base.$o(a::Real, b::Sym) = binary($(QuoteNode(o)), a, b)
end
Base.:-(a::Sym) = node!(a.tape, :neg, a)
Base.:+(a::Sym) = a
Base.abs(a::Sym) = node!(a.tape, :fabs, a)
Base.:!(a::Sym) = op(a) in BOOL_OPS ? node!(a.tape, :not, a) : throw(TraceError("! of a value that is not a comparison"))
Base.:&(a::Sym, b::Sym) = (op(a) in BOOL_OPS && op(b) in BOOL_OPS) ? node!(a.tape, :and, a, b) : throw(TraceError("& of values that are not comparisons"))
Base.:|(a::Sym, b::Sym) = (op(a) in BOOL_OPS && op(b) in BOOL_OPS) ? node!(a.tape, :or, a, b) : throw(TraceError("| of values that are not comparisons"))
Base.ifelse(c::Sym, a, b) = node!(c.tape, :where, c, lift(c.tape, a), lift(c.tape, b))
Base.max(a::Sym, b::Real) = node!(a.tape, :fmax, a, lift(a.tape, b)); Base.max(a::Real, b::Sym) = node!(b.tape, :fmax, lift(b.tape, a), b)
Base.max(a::Sym, b::Sym) = node!(a.tape, :fmax, a, b)
Base.min(a::Sym, b::Real) = node!(a.tape, :fmin, a, lift(a.tape, b)); Base.min(a::Real, b::Sym) = node!(b.tape, :fmin, lift(b.tape, a), b)
Base.min(a::Sym, b::Sym) = node!(a.tape, :fmin, a, b)
Base.atan(a::Sym, b::Real) = node!(a.tape, :atan2, a, lift(a.tape, b)); Base.atan(a::Real, b::Sym) = node!(b.tape, :atan2, lift(b.tape, a), b)
Base.atan(a::Sym, b::Sym) = node!(a.tape, :atan2, a, b)
for f in C_FUNCS
    @eval Base.$f(a::Sym) = node!(a.tape, $(QuoteNode(f)), a)
end
function smallpow(a::Sym, k::Integer)                                      # small integer powers as products, like Julia's own literal_pow
    k == 0 && return constant!(a.tape, 1.0)
    r = a
    for _ in 2:k; r = r * a; end
    r
end
Base.literal_pow(::typeof(^), a::Sym, ::Val{p}) where {p} = 0 <= p <= 4 ? smallpow(a, p) : p == -1 ? 1.0 / a : node!(a.tape, :pow, a, constant!(a.tape, p))
Base.:^(a::Sym, p::Integer) = 0 <= p <= 4 ? smallpow(a, p) : p == -1 ? 1.0 / a : node!(a.tape, :pow, a, constant!(a.tape, p))
Base.:^(a::Sym, p::Real) = p == 0.5 ? sqrt(a) : p == -1.0 ? 1.0 / a : (isinteger(p) && 0 <= p <= 4) ? smallpow(a, Int(p)) : node!(a.tape, :pow, a, lift(a.tape, p))
Base.:^(a::Sym, p::Sym) = node!(a.tape, :pow, a, p)
Base.:^(a::Real, p::Sym) = node!(p.tape, :pow, lift(p.tape, a), p)
# what must not happen during a trace: a branch on a sampled value, a conversion to a machine number
Base.convert(::Type{T}, ::Sym) where {T<:Union{AbstractFloat,Integer,Bool}} = throw(TraceError("a sampled value where Julia wants a $T (a branch on a draw? use ifelse)"))
Base.promote_rule(::Type{Sym}, ::Type{<:Real}) = Sym
Base.zero(a::Sym) = constant!(a.tape, 0.0); Base.one(a::Sym) = constant!(a.tape, 1.0)
Base.conj(a::Sym) = a; Base.real(a::Sym) = a

function reachable(t::Tape, outs::Vector{Int}, stop)                       # children before parents; nothing below the nodes in `stop`
    seen, order = Set{Int}(), Int[]
    function visit(n)
        n in seen && return
        push!(seen, n)
        if !(n in stop)
            for a in t.args[n]; a isa Int && visit(a); end
        end
        push!(order, n)
    end
    foreach(visit, outs)
    order
end
function literal(v::Float64)
    r = repr(v)
    signbit(v) ? "($r)" : r
end
# the DAG with the semantics of the emitted C (a comparison is 0.0 or 1.0, a truth value is `!= 0`): the check against the closure itself
function evaluate(t::Tape, outs::Vector{Int}, X::Vector{Float64})
    val = Dict{Int,Float64}()
    for n in reachable(t, outs, ())
        o, a = t.ops[n], t.args[n]
        v(k) = val[a[k]]
        val[n] = o === :x ? X[a[1]+1] : o === :ud ? t.params[a[1]+1] : o === :const ? a[1] :
                 o === :+ ? v(1) + v(2) : o === :- ? v(1) - v(2) : o === :* ? v(1) * v(2) : o === :/ ? v(1) / v(2) :
                 o === :< ? Float64(v(1) < v(2)) : o === :<= ? Float64(v(1) <= v(2)) : o === :(==) ? Float64(v(1) == v(2)) :
                 o === :> ? Float64(v(1) > v(2)) : o === :>= ? Float64(v(1) >= v(2)) : o === :!= ? Float64(v(1) != v(2)) :
                 o === :neg ? -v(1) : o === :not ? Float64(v(1) == 0.0) : o === :and ? Float64(v(1) != 0.0 && v(2) != 0.0) :
                 o === :or ? Float64(v(1) != 0.0 || v(2) != 0.0) : o === :where ? (v(1) != 0.0 ? v(2) : v(3)) :
                 o === :fabs ? abs(v(1)) : o === :fmax ? max(v(1), v(2)) : o === :fmin ? min(v(1), v(2)) : o === :atan2 ? atan(v(1), v(2)) :
                 o === :pow ? v(1)^v(2) : getfield(Base, o)(v(1))
    end
    [val[o] for o in outs]
end
# every maximal subexpression that depends on captured parameters (and constants) only -> one userdata slot, evaluated on the host
function hoist(t::Tape, outs::Vector{Int})
    slots, values = Dict{Int,String}(), Float64[]
    isempty(t.params) && return slots, values
    order = reachable(t, outs, ())
    onx, onp = Dict{Int,Bool}(), Dict{Int,Bool}()
    for n in order
        kids = [a for a in t.args[n] if a isa Int]
        onx[n] = t.ops[n] in (:x, :rw) || any(k -> onx[k], kids)
        onp[n] = t.ops[n] === :ud || any(k -> onp[k], kids)
    end
    take(n) = haskey(slots, n) || (slots[n] = "ud[$(length(values))]"; push!(values, evaluate(t, [n], Float64[])[1]))
    for n in order
        if onx[n]
            for a in t.args[n]; a isa Int && onp[a] && !onx[a] && take(a); end
        elseif onp[n] && n in outs
            take(n)
        end
    end
    all(isfinite, values) || throw(TraceError("a captured parameter evaluates to a non-finite value"))
    slots, values
end
function emit(t::Tape, outs::Vector{Int}, leaves::Dict{Int,String})
    name, lines = Dict{Int,String}(), String[]
    isbool(n) = t.ops[n] in BOOL_OPS && !haskey(leaves, n)
    num(n) = isbool(n) ? "(double)" * name[n] : name[n]
    cond(n) = isbool(n) ? name[n] : "(" * name[n] * " != 0.0)"
    for n in reachable(t, outs, keys(leaves))
        o, a = t.ops[n], t.args[n]
        if haskey(leaves, n)
            name[n] = leaves[n]
        elseif o in (:x, :rw, :ud)
            name[n] = "$(o)[$(a[1])]"
        elseif o === :const
            name[n] = literal(a[1])
        else
            so = String(o)
            kinds = get(C_OPERANDS, so, "n"^length(a))
            ops = [kinds[k] == 'c' ? cond(a[k]) : num(a[k]) for k in eachindex(a)]
            e = haskey(C_FORMAT, so) ? foldl((s, k) -> replace(s, "{$(k-1)}" => ops[k]), eachindex(ops); init=C_FORMAT[so]) : so * "(" * join(ops, ", ") * ")"
            push!(lines, (o in BOOL_OPS ? "const int t" : "const double t") * "$(n - 1) = $e;")
            name[n] = "t$(n - 1)"
        end
    end
    for (i, o) in enumerate(outs); push!(lines, "w[$(i - 1)] = $(num(o));"); end
    join(lines, "\n")
end
# a copy of the closure whose captured floats are parameters of the tape: closure types are parametrised by the types of the
# variables they capture, so the copy is `Closure{...Sym...}(fields...)`; anything else (a Core.Box, a non-parametric field) keeps its value
function parametrized(f, t::Tape)
    nf = nfields(f)
    nf == 0 && return f
    vals = Any[getfield(f, i) for i in 1:nf]
    any(v -> v isa AbstractFloat && isfinite(v), vals) || return f
    flds = Any[(v isa AbstractFloat && isfinite(v)) ? param!(t, Float64(v)) : v for v in vals]
    try
        return typeof(f).name.wrapper{map(typeof, flds)...}(flds...)
    catch
        empty!(t.params)
        return f
    end
end
"""
    trace_integrand(f, config; indexed=false, inplace=false, check_points=32) -> Integrand

`f(x, config)` (or the reference's :mcmc form `f(idx, x, config)`, idx 1-based; or its `inplace = true` form `f(x, weights, config)`,
src/main.jl:26, src/vegas/montecarlo.jl:140-141: the closure stores `weights[i] = value`) run once on symbolic draws and written out as a
device-source Integrand; TraceError if it cannot be, or if the written-out expression and the closure disagree at random points.
With one variable type `x[i]` is the i-th draw; with several, `x[v][i]` (a CompositeVar pool: `x[v][leaf, slot]` is not traced).
"""
function trace_integrand(f, c::Configuration; indexed::Bool=false, inplace::Bool=false, check_points::Int=32, parameters::Bool=true)
    indexed && inplace && throw(ArgumentError("the :mcmc form integrand(idx, var, config) has no in-place variant (src/main.jl:26-28)"))
    c.ncomp == 1 || throw(TraceError("complex weights are not traced"))
    any(v -> v isa CompositeVar || v isa FermiK, c.var) && throw(TraceError("CompositeVar / FermiK pools are not traced"))
    if parameters
        try
            return _trace_integrand(f, c, indexed, check_points, true, inplace)
        catch err
            err isa TraceError || err isa TypeError || err isa MethodError || rethrow()
        end                                       # (a branch on a captured float): once more with the captured values as literals
    end
    _trace_integrand(f, c, indexed, check_points, false, inplace)
end
function _trace_integrand(f, c::Configuration, indexed::Bool, check_points::Int, parameters::Bool, inplace::Bool=false)
    t = Tape()
    maxdof = [maximum(c.dof[i][v] for i in 1:c.N) for v in 1:length(c.var)]
    k = 0
    pools = Vector{Vector{Sym}}()
    for v in 1:length(c.var)
        push!(pools, [node!(t, :x, Int32(k + s - 1)) for s in 1:maxdof[v]]); k += maxdof[v]
    end
    ndraw = k
    arg = length(pools) == 1 ? pools[1] : Tuple(pools)
    g = parameters ? parametrized(f, t) : f
    outs = Any[]
    try
        if indexed
            outs = Any[g(i, arg, c) for i in 1:c.N]
        elseif inplace                              # weights: one entry per integrand, zero until the closure stores into it
            outs = Any[0.0 for _ in 1:c.N]
            g(arg, outs, c)
        else
            r = g(arg, c)
            outs = r isa Tuple ? Any[r...] : Any[r]
        end
    catch err
        err isa TraceError && rethrow()
        throw(TraceError("$(typeof(err)): the closure cannot be run on symbolic draws"))
    end
    length(outs) == c.N || throw(TraceError("the integrand must return one value per integrand ($(c.N)), got $(length(outs))"))
    ids = Int[lift(t, o).id for o in outs]
    slots, ud = hoist(t, ids)
    body = emit(t, ids, slots)
    lo = Float64[]; hi = Float64[]
    for (v, var) in enumerate(c.var), _ in 1:maxdof[v]
        push!(lo, Float64(var.lower)); push!(hi, Float64(var.upper))
    end
    for _ in 1:check_points                        # the closure itself on plain numbers against the written-out expression
        X = [lo[j] + rand() * (hi[j] - lo[j]) for j in 1:ndraw]
        j = 0
        for (v, var) in enumerate(c.var), _ in 1:maxdof[v]
            j += 1
            var isa Discrete && (X[j] = Float64(rand(var.lower:var.upper)))
        end
        num = length(pools) == 1 ? X : Tuple(X[(sum(maxdof[1:v-1])+1):sum(maxdof[1:v])] for v in 1:length(c.var))
        ref = indexed ? Float64[f(i, num, c) for i in 1:c.N] : inplace ? (r = zeros(c.N); f(num, r, c); r) :
              (r = f(num, c); r isa Tuple ? Float64[r...] : Float64[r])
        got = evaluate(t, ids, X)
        for i in 1:c.N
            (isfinite(ref[i]) == isfinite(got[i]) && (!isfinite(ref[i]) || isapprox(got[i], ref[i]; rtol=1e-10, atol=1e-290))) ||
                throw(TraceError("the traced expression and the closure disagree on integrand $i: the closure is not a pure function of its draws"))
        end
    end
    Integrand(body, ud)
end

# ---- Julia closures as integrands: the host "batch callback" slow path (mci_set_integrand_host) ----------------------
# f(x, config) is called once per launch (:vegas) or once per Markov step (:vegasmc, :mcmc: the chains of a launch advance in lock
# step) with x[k] = the vector of draw k over the n samples / chains of the batch (several
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
# the reference's `inplace = true` form f(x, weights, config) (src/main.jl:26, src/vegas/montecarlo.jl:140-141, src/vegas_mc/updates.jl:67-70):
# `weights[i] = values` stores integrand i's values over the batch straight into the library's output array (zero on entry)
struct WeightRows <: AbstractVector{Any}
    W::Matrix{Float64}; ncomp::Int
end
Base.size(r::WeightRows) = (size(r.W, 2) ÷ r.ncomp,)
Base.getindex(r::WeightRows, i::Int) = r.ncomp == 2 ? complex.(view(r.W, :, 2i - 1), view(r.W, :, 2i)) : view(r.W, :, i)
function Base.setindex!(r::WeightRows, v, i::Int)
    if r.ncomp == 2
        r.W[:, 2i-1] .= real.(v); r.W[:, 2i] .= imag.(v)
    else
        r.W[:, i] .= v
    end
    v
end
function _host_inplace_trampoline(x::Ptr{Float64}, w::Ptr{Float64}, n::Int64, ndraw::Int32, nw::Int32, user::Ptr{Cvoid})::Cint
    try
        f, c = _closures[user]
        X = unsafe_wrap(Array, x, (Int(n), Int(ndraw)))
        W = unsafe_wrap(Array, w, (Int(n), Int(nw)))
        fill!(W, 0.0)
        f([view(X, :, k) for k in 1:ndraw], WeightRows(W, c.ncomp), c)
        return Cint(0)
    catch err
        @error "host integrand failed" err
        return Cint(1)
    end
end
# the reference's :mcmc form f(idx, x, config) (src/mcmc/montecarlo.jl:34-36; idx 1-based like the reference): one call per integrand
# index some chain asks for, over the chains that ask for it (mci_set_integrand_host_indexed)
function _host_idx_trampoline(idx::Ptr{Int32}, x::Ptr{Float64}, w::Ptr{Float64}, n::Int64, ndraw::Int32, ncomp::Int32, user::Ptr{Cvoid})::Cint
    try
        f, c = _closures[user]
        I = unsafe_wrap(Array, idx, (Int(n),))
        X = unsafe_wrap(Array, x, (Int(n), Int(ndraw)))
        W = unsafe_wrap(Array, w, (Int(n), Int(ncomp)))
        for i in unique(I)
            i < 0 && continue
            sel = findall(==(i), I)
            o = f(Int(i) + 1, [X[sel, k] for k in 1:ndraw], c)
            if ncomp == 2
                W[sel, 1] .= real.(o); W[sel, 2] .= imag.(o)
            else
                W[sel, 1] .= o
            end
        end
        return Cint(0)
    catch err
        @error "host integrand failed" err
        return Cint(1)
    end
end
"""
    callback_form(f, solver, inplace; what=:integrand) -> :plain | :inplace | :indexed

Which form a closure is called in: decided by the SOLVER and the `inplace` keyword like the reference (src/main.jl:26-28, :38-40), never
by counting parameters -- `:mcmc` calls `integrand(idx, var, config)` / `measure(idx, var, obs, relative_weight, config)`; the others
`inplace ? integrand(var, weights, config) : integrand(var, config)` and `measure(var, obs, relative_weights, config)`.  The closure's
methods are the cross-check: none with that many arguments is the reference's MethodError, raised here before anything runs.
"""
function callback_form(f::Function, solver::Symbol, inplace::Bool=false; what::Symbol=:integrand)
    form = solver == :mcmc ? :indexed : (inplace && what == :integrand) ? :inplace : :plain
    want = what == :integrand ? (form == :plain ? 2 : 3) : (form == :indexed ? 5 : 4)
    any(m -> m.nargs - 1 == want || (m.isva && m.nargs - 2 <= want), methods(f)) ||
        throw(ArgumentError("solver = :$solver$(form == :inplace ? ", inplace = true" : "") calls the $what with $want arguments " *
                            "(src/main.jl:26-28, :38-40: :mcmc -> (idx, var, ...), inplace = true -> integrand(var, weights, config)); " *
                            "the closure has no such method"))
    form
end
function bind_host!(c::Configuration, f::Function, form::Symbol=:plain)
    prob = bind!(c, Integrand("", Float64[]), nothing)
    _closures[prob] = (f, c)
    if form == :indexed                            # f(idx, x, config): what :mcmc calls (src/mcmc/montecarlo.jl:34-36)
        cb = @cfunction(_host_idx_trampoline, Cint, (Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Int64, Int32, Int32, Ptr{Cvoid}))
        check(ccall((:mci_set_integrand_host_indexed, libmci), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), prob, cb, prob))
    elseif form == :inplace                        # f(x, weights, config): inplace = true (src/vegas/montecarlo.jl:140-141)
        cb = @cfunction(_host_inplace_trampoline, Cint, (Ptr{Float64}, Ptr{Float64}, Int64, Int32, Int32, Ptr{Cvoid}))
        check(ccall((:mci_set_integrand_host, libmci), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), prob, cb, prob))
    else
        cb = @cfunction(_host_trampoline, Cint, (Ptr{Float64}, Ptr{Float64}, Int64, Int32, Int32, Ptr{Cvoid}))
        check(ccall((:mci_set_integrand_host, libmci), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), prob, cb, prob))
    end
    prob
end

# ---- Julia closures as `measure` (mci_set_measure_host / _indexed) --------------------------------------------------------
# measure(x, obs, weights, config) (src/vegas/montecarlo.jl:156-161) is called once per statistical block with the block's records
# (x[k] = the vector of draw k, weights[i] = the vector of integrand i's relative weights; a record = a sample under :vegas, a
# measured step of one of the block's chains under :vegasmc / :mcmc); obs is shaped like the `obs` keyword and zeroed per block.
const _measures = Dict{Ptr{Cvoid},Any}()
function _obs_views(c::Configuration, O::Vector{Float64})
    out, off = Any[], 0
    for nb in c.obs_nbin
        v = view(O, off+1:off+nb)
        push!(out, c.ncomp == 2 ? reinterpret(ComplexF64, v) : v)
        off += nb
    end
    out
end
function _measure_trampoline(x::Ptr{Float64}, relw::Ptr{Float64}, n::Int64, stride::Int64, ndraw::Int32, nw::Int32, block::Int64,
                             obs::Ptr{Float64}, nobs::Int32, user::Ptr{Cvoid})::Cint
    try
        f, c = _measures[user]
        X = unsafe_wrap(Array, x, (Int(stride), Int(ndraw)))
        R = unsafe_wrap(Array, relw, (Int(stride), Int(nw)))
        O = unsafe_wrap(Array, obs, (Int(nobs),))
        cols = [view(X, 1:Int(n), k) for k in 1:ndraw]
        W = c.ncomp == 2 ? [complex.(view(R, 1:Int(n), 2i - 1), view(R, 1:Int(n), 2i)) for i in 1:c.N] : [view(R, 1:Int(n), i) for i in 1:c.N]
        f(cols, _obs_views(c, O), W, c)
        return Cint(0)
    catch err
        @error "host measure failed" err
        return Cint(1)
    end
end
# the reference's :mcmc form measure(idx, x, obs, weight, config) (src/mcmc/montecarlo.jl:166-169; idx 1-based like the reference)
function _measure_idx_trampoline(idx::Ptr{Int32}, x::Ptr{Float64}, relw::Ptr{Float64}, n::Int64, stride::Int64, ndraw::Int32, ncomp::Int32,
                                 block::Int64, obs::Ptr{Float64}, nobs::Int32, user::Ptr{Cvoid})::Cint
    try
        f, c = _measures[user]
        I = unsafe_wrap(Array, idx, (Int(n),))
        X = unsafe_wrap(Array, x, (Int(stride), Int(ndraw)))
        R = unsafe_wrap(Array, relw, (Int(stride), Int(ncomp)))
        O = unsafe_wrap(Array, obs, (Int(nobs),))
        ov = _obs_views(c, O)
        for i in unique(I)
            i < 0 && continue
            sel = findall(==(i), I)
            w = ncomp == 2 ? complex.(R[sel, 1], R[sel, 2]) : R[sel, 1]
            f(Int(i) + 1, [X[sel, k] for k in 1:ndraw], ov, w, c)
        end
        return Cint(0)
    catch err
        @error "host measure failed" err
        return Cint(1)
    end
end
function bind_measure_host!(c::Configuration, prob, m::Function, form::Symbol=:plain)
    _measures[prob] = (m, c)
    if form == :indexed                            # measure(idx, x, obs, weight, config): what :mcmc calls (src/mcmc/montecarlo.jl:166-169)
        cb = @cfunction(_measure_idx_trampoline, Cint, (Ptr{Int32}, Ptr{Float64}, Ptr{Float64}, Int64, Int64, Int32, Int32, Int64, Ptr{Float64}, Int32, Ptr{Cvoid}))
        check(ccall((:mci_set_measure_host_indexed, libmci), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), prob, cb, prob))
    else
        cb = @cfunction(_measure_trampoline, Cint, (Ptr{Float64}, Ptr{Float64}, Int64, Int64, Int32, Int32, Int64, Ptr{Float64}, Int32, Ptr{Cvoid}))
        check(ccall((:mci_set_measure_host, libmci), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), prob, cb, prob))
    end
    prob
end

# Result   reference src/statistics.jl:16-63
struct Result
    mean::Vector{Float64}; stdev::Vector{Float64}; chi2::Vector{Float64}
    neval::Int; ignore::Int; config::Configuration
    iter_mean::Matrix{Float64}; iter_std::Matrix{Float64}        # [niter, nobs]
    correlated::Bool     # the iterations continued each other's chains: `stdev` is the block-lineage error (mci_lineage_sums), not statistics.jl:198
    warmup::Int          # launches that were run again instead of being counted (automatic :mcmc chain lengths)
end
Result(mean, stdev, chi2, neval, ignore, config, iter_mean, iter_std) = Result(mean, stdev, chi2, neval, ignore, config, iter_mean, iter_std, false, 0)

"""
    average(iter_mean, iter_std; init=1, max=length(iter_mean))  -> (mean, std, chi2)

Inverse-variance weighted average of one observable's history (reference src/statistics.jl:186-220), by the library's
`mci_average` so that Julia, Python and C callers get the same digits.
"""
function average(iter_mean::AbstractVector{Float64}, iter_std::AbstractVector{Float64}; init::Int=1, max::Int=length(iter_mean))
    m, e = collect(Float64, iter_mean), collect(Float64, iter_std)
    a, b, c = Ref(0.0), Ref(0.0), Ref(0.0)
    ccall((:mci_average, libmci), Cvoid, (Ptr{Float64}, Ptr{Float64}, Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
          m, e, 1, init, max, a, b, c)
    a[], b[], c[]
end

# Result(res, ignore): the same history averaged from another first iteration  (src/statistics.jl:56-62)
function Result(res::Result, ignore::Int)
    ignore == res.ignore && return res
    niter, nobs = size(res.iter_mean)
    avg = [average(res.iter_mean[:, o], res.iter_std[:, o]; init=ignore + 1, max=niter) for o in 1:nobs]
    Result([a[1] for a in avg], [a[2] for a in avg], [a[3] for a in avg], res.neval, ignore, res.config, res.iter_mean, res.iter_std)
end
dof(r::Result) = (size(r.iter_mean, 1) - (r.ignore + 1) + 1) - 1                    # src/statistics.jl:65-68
Base.getindex(r::Result, i::Int) = (r.mean[_col(r.config, i)], r.stdev[_col(r.config, i)], r.chi2[_col(r.config, i)])
_col(c::Configuration, i::Int, pick::Int=1) = sum(c.obs_nbin[1:i-1]) + pick          # flat statistics column of integrand i

function Base.show(io::IO, r::Result)                                               # src/statistics.jl:104-118
    for i in 1:r.config.N
        m, e, c2 = r[i]
        if dof(r) == 0
            print(io, "Integral $i = $m ± $e")
        else
            print(io, "Integral $i = $m ± $e   (reduced chi2 = $(round(c2, sigdigits=3)))")
        end
        i < r.config.N && print(io, "\n")
    end
end

sig_digits(err::Real) = (err == 0 || !isfinite(err)) ? 0 : max(0, 2 - floor(Int, log10(abs(err))))   # src/statistics.jl:74-79
function tostring(m::Real, e::Real)                                                                  # src/statistics.jl:87-96
    (isfinite(m) && isfinite(e)) || return "$m ± $e"
    nd = sig_digits(e)
    fmt = Printf.Format("%.$(nd)f")
    string(Printf.format(fmt, m), " ± ", Printf.format(fmt, e))
end

"""
    report(result::Result, ignore=result.ignore; pick=1, name=nothing, verbose=0, io=stdout)

The per-iteration table of the reference (src/statistics.jl:137-172): every iteration's estimate next to the running
weighted average and its reduced chi2.  `pick` selects the component of an array observable (1-based).
"""
function report(r::Result, ignore::Int=r.ignore; pick::Int=1, name=nothing, verbose::Int=0, io::IO=stdout)
    niter = size(r.iter_mean, 1)
    for i in 1:r.config.N
        info = name === nothing ? "$i" : "$(collect(name)[i])"
        col = _col(r.config, i, pick)
        if verbose >= 0
            println(io, "================================================     Integral $info    ============================================================")
            println(io, @sprintf("%6s                 %-32s                 %-32s %22s", "iter", "         integral", "        wgt average", "reduced chi2"))
            println(io, "-"^127)
            for it in 1:niter
                m, e, c2 = average(r.iter_mean[:, col], r.iter_std[:, col]; init=ignore + 1, max=it)
                iterstr = it <= ignore ? "ignore" : "$it"
                println(io, @sprintf("%6s %36s %36s %16.4f", iterstr, tostring(r.iter_mean[it, col], r.iter_std[it, col]), tostring(m, e), abs(c2)))
            end
            println(io, "-"^127)
            r.correlated && println(io, "  the iterations continued each other's chains: block-lineage error of the average  ", tostring(r.mean[col], r.stdev[col]),
                                    r.warmup > 0 ? "   ($(r.warmup) warm-up launches run again)" : "")
        else
            m, e, c2 = r.mean[col], r.stdev[col], r.chi2[col]
            println(io, dof(r) == 0 ? "Integral $info = $m ± $e" : "Integral $info = $m ± $e   (reduced chi2 = $(round(c2, sigdigits=3)))")
        end
    end
end

_typestr(v) = v isa Continuous ? "Continuous" : v isa Discrete ? "Discrete" : v isa CompositeVar ? "Composite" : v isa FermiK ? "FermiK" : string(typeof(v))
function _default_neighbor(Nd::Int)                                                  # src/configuration.jl:203-208
    nb = [[i - 1, i + 1] for i in 1:Nd]
    nb[1] = Nd == 2 ? [2] : [Nd, 2]
    nb[Nd] = [1]
    Nd >= 3 && (nb[Nd-1] = [Nd - 2])
    nb
end

"""
    report(config::Configuration; io=stdout)

The acceptance tables of the reference (src/configuration.jl:345-464): ChangeIntegrand per edge of the neighbor graph,
ChangeVariable and SwapVariable per (integrand, variable), then visited and reweight -- from the library's
config.propose / config.accept / visited / reweight of the last iteration.
"""
function report(c::Configuration; io::IO=stdout)
    Nd = c.N + 1
    bar = "-"^85
    pr, ac = acceptance(c)
    vis, rw = visited(c), reweight(c)
    neval = max(c.neval, 1)
    nb = c.neighbor === nothing ? _default_neighbor(Nd) : c.neighbor
    println(io)
    println(io, "===========================  Configuration  =========================================")
    println(io, Dates.now())
    println(io, bar)
    println(io, "Integral num = $(c.N), dof = $(c.dof), with variables:")
    for (vi, v) in enumerate(c.var)
        println(io, "$vi. $v")
    end
    println(io, bar)
    line(u, i, j) = @sprintf("%11.6f%% %11.6f%% %12.6f", pr[u, i, j] / neval * 100.0, ac[u, i, j] / neval * 100.0, ac[u, i, j] / pr[u, i, j])
    println(io, @sprintf("%-20s %12s %12s %12s", "ChangeIntegrand", "Proposed", "Accepted", "Ratio  "))
    for n in nb[Nd]
        println(io, @sprintf("Norm -> %2d:           ", n), line(1, Nd, n))
    end
    for idx in 1:Nd-1, n in nb[idx]
        if n == Nd
            println(io, @sprintf("  %d ->Norm:           ", idx), line(1, idx, n))
        else
            println(io, @sprintf("  %d -> %2d:            ", idx, n), line(1, idx, n))
        end
    end
    println(io, bar)
    for (u, title) in ((2, "ChangeVariable"), (3, "SwapVariable"))
        println(io, @sprintf("%-20s %12s %12s %12s", title, "Proposed", "Accepted", "Ratio  "))
        for idx in 1:Nd-1, (vi, v) in enumerate(c.var)
            println(io, @sprintf("  %2d / %-11s:   ", idx, _typestr(v)), line(u, idx, vi))
        end
        println(io, bar)
    end
    println(io, "Integrand            Visited      ReWeight")
    println(io, @sprintf("  Norm   :     %12i %12.6f", round(Int, vis[end]), rw[end]))
    for idx in 1:Nd-1
        println(io, @sprintf("  Order%2d:     %12i %12.6f", idx, round(Int, vis[idx]), rw[idx]))
    end
    println(io, bar)
    println(io, "Integrand evaluation = $(c.neval)\n")
end

"""
    integrate(integrand::Integrand; solver=:vegasmc, config=nothing, neval=1e4, niter=10, block=16, gamma=1.0,
              adapt=true, ignore=adapt ? 1 : 0, measure=nothing, measurefreq=1, inplace=false, kwargs...)

Same keywords as the reference (src/main.jl:71-90); the loop of src/main.jl:142-218 runs inside
`mci_integrate` on the GPU.  Unknown keywords go to `Configuration` (src/main.jl:95-97).  A closure is called in the form the reference's
solver calls it in (`callback_form`): `integrand(var, config)`, `integrand(var, weights, config)` with `inplace = true`,
`integrand(idx, var, config)` under `:mcmc`.
"""
function integrate(integrand::Union{Integrand,AbstractString,Function}; solver::Symbol=:vegasmc, config=nothing, neval=1e4, niter=10,
                   block=16, gamma=1.0, adapt=true, ignore::Int=adapt ? 1 : 0, measure=nothing, measurefreq::Int=1,
                   thermal_ratio=0.1, inplace::Bool=false, reweight_goal::Union{Vector{Float64},Nothing}=nothing,
                   nchain=0, rng_bits::Int=52, rng_rounds::Int=10, train_walk::Int=-1, deterministic::Bool=false, chain_carry::Int=-1,
                   persistent::Int=-1, trace::Bool=true, print=-1, verbose=-1, kwargs...)
    haskey(SOLVER, solver) || error("Solver $solver is not supported!")                  # main.jl:263
    config === nothing && (config = Configuration(; kwargs...))                          # main.jl:95-97
    # workers: after init_comm!(...) the library runs this rank's share of the blocks and sums every iteration's statistics and
    # histograms over the ranks with one RCCL all-reduce (main.jl:113-122, :152-188); nothing to do here per call
    form = integrand isa Function ? callback_form(integrand, solver, inplace) : :plain   # by solver + flag (main.jl:26-28); ArgumentError if the closure has no such method
    if integrand isa Function && trace             # a Julia closure: run once on symbolic draws and written out as device source (trace_integrand) ...
        try
            integrand = trace_integrand(integrand, config; indexed=form == :indexed, inplace=form == :inplace)
        catch err
            err isa TraceError || rethrow()
            max(print, verbose) > 0 && println("integrand not traced (", err.msg, "): host callback path")
        end
    end
    if integrand isa Function                      # ... or, if it cannot be, the host batch-callback path (per launch under :vegas, per Markov step under :vegasmc / :mcmc)
        prob = bind_host!(config, integrand, form)
    else
        f = integrand isa Integrand ? integrand : Integrand(String(integrand), config.userdata === nothing ? Float64[] : Float64.(config.userdata))
        prob = bind!(config, f, measure isa Function ? nothing : measure)
    end
    if measure isa Function                        # a Julia closure as `measure`: per block, after the launch
        bind_measure_host!(config, prob, measure, callback_form(measure, solver; what=:measure))
    elseif haskey(_measures, prob)                 # the problem still carries an earlier call's closure: back to the device-side measure
        delete!(_measures, prob)
        check(ccall((:mci_set_measure_host, libmci), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), prob, C_NULL, C_NULL))
        measure isa Measure && check(ccall((:mci_set_measure_source, libmci), Cint, (Ptr{Cvoid}, Cstring), prob, measure.body))
    end
    # engine-specific knobs (include/mci.h): the opt-in cheaper streams, train!'s refinement walk
    check(ccall((:mci_set_rng_bits, libmci), Cint, (Ptr{Cvoid}, Int32), prob, rng_bits))
    check(ccall((:mci_set_rng_rounds, libmci), Cint, (Ptr{Cvoid}, Int32), prob, rng_rounds))
    check(ccall((:mci_set_train_walk, libmci), Cint, (Ptr{Cvoid}, Int32), prob, train_walk))
    # deterministic = true: bit-identical results for a fixed seed, like the reference's sequential loop under MersenneTwister(seed)
    # (configuration.jl:190); chain_carry: -1 automatic / 1 (many-chain iterations of :vegasmc AND :mcmc continue the previous one's chains; the
    # error of such a run is the block-lineage error, ResultC.correlated), 0 off (every iteration starts afresh like mcmc/montecarlo.jl:118-124)
    check(ccall((:mci_set_deterministic, libmci), Cint, (Ptr{Cvoid}, Int32), prob, deterministic ? 1 : 0))
    check(ccall((:mci_set_chain_carry, libmci), Cint, (Ptr{Cvoid}, Int32), prob, chain_carry))
    # persistent: -1 automatic (a launch-bound :vegas call over one Continuous variable type runs all its iterations as one launch), 0 off, 1 on
    check(ccall((:mci_set_persistent, libmci), Cint, (Ptr{Cvoid}, Int32), prob, persistent))
    nobs = sum(config.obs_nbin)
    im, ie = zeros(nobs, niter), zeros(nobs, niter)           # row-major [niter][nobs] on the C side
    m, s, c2 = zeros(nobs), zeros(nobs), zeros(nobs)
    goal = reweight_goal === nothing ? Float64[] : reweight_goal
    args = Ref(IntegrateArgs(SOLVER[solver], Int64(neval), niter, block, ignore, adapt, gamma, measurefreq, UInt64(config.seed),
                             nchain, config.iterations_done, thermal_ratio,
                             reweight_goal === nothing ? Ptr{Float64}(C_NULL) : pointer(goal)))
    res = ResultC(niter, nobs, pointer(im), pointer(ie), pointer(m), pointer(s), pointer(c2), 0, 0.0, Ptr{Float64}(C_NULL), 0, 0)
    GC.@preserve im ie m s c2 goal check(ccall((:mci_integrate, libmci), Cint, (Ptr{Cvoid}, Ptr{IntegrateArgs}, Ref{ResultC}), prob, args, res))
    config.iterations_done += niter
    nworker = _comm[].size
    nblock = block > nworker ? (block ÷ nworker) * nworker : nworker                   # _standardize_block, main.jl:220-234
    config.neval = (Int(neval) ÷ nblock) * nblock
    r = Result(m, s, c2, res.neval, ignore, config, permutedims(im), permutedims(ie), res.correlated != 0, Int(res.warmup))
    max(print, verbose) >= 0 && _comm[].rank == 0 && report(r)                          # main.jl:212-213
    r
end

end # module
