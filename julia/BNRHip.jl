# BNRHip.jl -- thin Julia shim over libbnr_hip.so (include/bnr_hip.h).
#
# Drop-in for the Gibbs hot path of BayesianNetworkRegression.jl: it provides `initialize_and_run!`-level and
# `generate_samples!`-level entry points with the reference's argument meaning, and returns the reference's own
# `Results(state::Table, rhatξ::Table, rhatγ::Table, burn_in, sampled)` (src/gibbs.jl:23-29), so `Summary` and
# `show` of the package work unchanged.  No logic lives here: every sampling step is a `ccall`.
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: the build image has no `julia` binary.  The same C ABI is exercised by the
# Python/ctypes mirror (bayesiannetworkregression.jl_amd/_capi.py, api.py) in tests/.
module BNRHip

using TypedTables, Random
import BayesianNetworkRegression: Results, lower_triangle, setup_X!

const LIB = get(ENV, "BNR_HIP_LIB", joinpath(@__DIR__, "..", "bayesiannetworkregression.jl_amd", "libbnr_hip.so"))

struct Hyper
    eta::Cdouble; zeta::Cdouble; iota::Cdouble; aDelta::Cdouble; bDelta::Cdouble; nu::Cdouble
end

lasterr() = unsafe_string(ccall((:bnr_last_error, LIB), Cstring, ()))
check(rc) = rc == 0 || error("libbnr_hip: $(lasterr()) (status $rc)")

mutable struct Chain
    h::Ptr{Cvoid}
    n::Int; V::Int; R::Int; q::Int; tot::Int
end

function Chain(X::Matrix{Float64}, y::Vector{Float64}, R, tot_save, seed, c; η=1.01, ζ=1.0, ι=1.0, aΔ=1.0, bΔ=1.0, ν=10, device=0)
    n, q = size(X)
    V = Int64((-1 + sqrt(1 + 8 * q)) / 2)
    hy = Ref(Hyper(η, ζ, ι, aΔ, bΔ, ν))
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve X y check(ccall((:bnr_chain_create, LIB), Cint,
        (Int32, Int32, Int32, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Hyper}, UInt64, Int32, Int32, Int32, Ref{Ptr{Cvoid}}),
        n, V, R, X, y, hy, UInt64(seed), c, device, tot_save, out))
    ch = Chain(out[], n, V, R, q, tot_save)
    finalizer(x -> ccall((:bnr_chain_destroy, LIB), Cint, (Ptr{Cvoid},), x.h), ch)
    ch
end

# another chain of the same fit on the same GPU: X, y stay shared on the device (bnr_chain_create_like)
function chain_like(donor::Chain, seed, c, tot_save=donor.tot)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:bnr_chain_create_like, LIB), Cint, (Ptr{Cvoid}, UInt64, Int32, Int32, Ref{Ptr{Cvoid}}), donor.h, UInt64(seed), c, tot_save, out))
    ch = Chain(out[], donor.n, donor.V, donor.R, donor.q, tot_save)
    finalizer(x -> ccall((:bnr_chain_destroy, LIB), Cint, (Ptr{Cvoid},), x.h), ch)
    ch
end

init_prior!(ch::Chain) = check(ccall((:bnr_chain_init_prior, LIB), Cint, (Ptr{Cvoid},), ch.h))

# Summary statistics on the device (gibbs.jl:1214-1250): (mean γ, lower, upper, P(ξ=1)); ranks are 1-based positions in the sorted sample
function summary_stats(ch::Chain, nburn, nsamp; interval=95)
    lb = (100 - interval) / 200
    klo, khi = Int(round(nsamp * lb)), Int(round(nsamp * (1 - lb)))
    m, lo, hi, p = zeros(ch.q), zeros(ch.q), zeros(ch.q), zeros(ch.V)
    check(ccall((:bnr_chain_summary, LIB), Cint, (Ptr{Cvoid}, Int32, Int32, Int32, Int32, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
        ch.h, nburn + 1, nsamp, klo, khi, m, lo, hi, p))
    m, lo, hi, p
end

# lockstep group: the chains of one fit that share a GPU advance together (replaces the pmap over chains, gibbs.jl:946-948)
mutable struct Group
    h::Ptr{Cvoid}
    chains::Vector{Chain}
end
function Group(chains::Vector{Chain})
    hs = Ptr{Cvoid}[ch.h for ch in chains]
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve hs check(ccall((:bnr_group_create, LIB), Cint, (Ptr{Ptr{Cvoid}}, Int32, Ref{Ptr{Cvoid}}), hs, length(hs), out))
    g = Group(out[], chains)
    finalizer(x -> ccall((:bnr_group_destroy, LIB), Cint, (Ptr{Cvoid},), x.h), g)
    g
end
function run!(g::Group, first_index, nburn, total, purge_burn; prog_freq=0, tick=nothing)
    nxt = Ref{Int32}(0)
    cb = tick === nothing ? C_NULL : @cfunction((u, d) -> (tick(); nothing), Cvoid, (Ptr{Cvoid}, Int64))
    check(ccall((:bnr_group_run, LIB), Cint, (Ptr{Cvoid}, Int32, Int32, Int32, Int32, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Int32}),
        g.h, first_index, nburn, total, isnothing(purge_burn) ? 0 : purge_burn, prog_freq, cb, C_NULL, nxt))
    Int(nxt[])
end

# run!(X,y,state,c,first_index,nburn,total,...,purge_burn,channel)  (gibbs.jl:849-864)
function run!(ch::Chain, first_index, nburn, total, purge_burn; prog_freq=0, tick=nothing)
    nxt = Ref{Int32}(0)
    cb = tick === nothing ? C_NULL : @cfunction((u, d) -> (tick(); nothing), Cvoid, (Ptr{Cvoid}, Int64))
    check(ccall((:bnr_chain_run, LIB), Cint, (Ptr{Cvoid}, Int32, Int32, Int32, Int32, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Int32}),
        ch.h, first_index, nburn, total, isnothing(purge_burn) ? 0 : purge_burn, prog_freq, cb, C_NULL, nxt))
    Int(nxt[])
end

# the reference's 14-column state Table (gibbs.jl:835-841); the 3 dead columns stay undef as in the reference
function new_table(tot, V, R)
    q = Int64(V * (V + 1) / 2)
    Table(τ² = Array{Float64,3}(undef, (tot, 1, 1)), u = Array{Float64,3}(undef, (tot, R, V)),
          ξ = Array{Float64,3}(undef, (tot, V, 1)), γ = Array{Float64,3}(undef, (tot, q, 1)),
          S = Array{Float64,3}(undef, (tot, q, 1)), θ = Array{Float64,3}(undef, (tot, 1, 1)),
          Δ = Array{Float64,3}(undef, (tot, 1, 1)), M = Array{Float64,3}(undef, (tot, R, R)),
          μ = Array{Float64,3}(undef, (tot, 1, 1)), λ = Array{Float64,3}(undef, (tot, R, 1)),
          πᵥ = Array{Float64,3}(undef, (tot, R, 3)), Σ⁻¹ = Array{Float64,3}(undef, (tot, R, R)),
          invC = Array{Float64,3}(undef, (tot, R, R)), μₜ = Array{Float64,3}(undef, (tot, R, 1)))
end

function fetch!(state::Table, ch::Chain, first_row=1, last_row=ch.tot)
    tot = size(state.τ², 1)
    GC.@preserve state check(ccall((:bnr_chain_fetch, LIB), Cint,
        (Ptr{Cvoid}, Int32, Int32, Int32, Int32, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble},
         Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
        ch.h, first_row, last_row, tot, 0, state.τ², state.u, state.ξ, state.γ, state.S, state.θ, state.Δ, state.M,
        state.μ, state.λ, state.πᵥ))
    state
end

function rhat_stats(ch::Chain, first_row, nsamp)
    out = Vector{Float64}(undef, 4 * (ch.q + ch.V))
    check(ccall((:bnr_chain_rhat_stats, LIB), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{Cdouble}), ch.h, first_row, nsamp, out))
    out
end

function rhat_from_stats(stats::Matrix{Float64}, nsamp)          # columns = chains
    np = size(stats, 1) ÷ 4
    out = Vector{Float64}(undef, np)
    check(ccall((:bnr_rhat_from_stats, LIB), Cint, (Ptr{Cdouble}, Int32, Int32, Int32, Ptr{Cdouble}), stats, size(stats, 2), np, nsamp, out))
    out
end

# generate_samples!(X, y, R; ...) (gibbs.jl:897-1020), single round; the PSRF top-up loop is the reference's own code
# with `initialize_and_run!`/`run!`/`copy_table!` replaced by Chain/run!/bnr_chain_move_rows.
function generate_samples!(X, y, R; η=1.01, ζ=1.0, ι=1.0, aΔ=1.0, bΔ=1.0, ν=10, nburn=30000, nsamp=20000,
                           x_transform=true, num_chains=2, seed=nothing, purge_burn=nothing, device=0)
    V = x_transform ? size(X[1], 1) : Int64((-1 + sqrt(1 + 8 * size(X, 2))) / 2)
    q = floor(Int, V * (V + 1) / 2)
    X_new = Matrix{Float64}(undef, size(X, 1), q)
    setup_X!(X_new, X, x_transform)
    total = nburn + nsamp
    tot_save = isnothing(purge_burn) ? total : nsamp + purge_burn
    seed = isnothing(seed) ? rand(1:55555) : seed
    chains = [Chain(X_new, Vector{Float64}(y), R, tot_save, seed, c; η, ζ, ι, aΔ, bΔ, ν, device) for c in 1:num_chains]
    for ch in chains
        init_prior!(ch)
        run!(ch, 2, nburn, total, purge_burn)
    end
    stt = isnothing(purge_burn) ? nburn : purge_burn
    stats = hcat([rhat_stats(ch, stt + 1, nsamp) for ch in chains]...)
    r = rhat_from_stats(stats, nsamp)
    state = fetch!(new_table(tot_save, V, R), chains[1])
    Results(state, Table(ξ = r[q+1:end]), Table(γ = r[1:q]), stt, nsamp)
end

end # module
