# BNRHip.jl -- thin Julia shim over libbnr_hip.so (include/bnr_hip.h).
# SPDX-License-Identifier: GPL-2.0-or-later.  The schedule arithmetic of `generate_samples!` / `generate_samples_dbl!` below (num2move,
# first_index, table sizes, the messages) and the layout of `Fit!`'s parameters.log restate BayesianNetworkRegression.jl
# src/gibbs.jl:729-746, 955-1013, 1119-1190 (GPL-2.0, Ozminkowski & Solis-Lemus) so that this module is a drop-in for those
# functions; everything else is `ccall` glue written for this repository.
#
# Drop-in for the Gibbs hot path of BayesianNetworkRegression.jl: `Fit!` (src/gibbs.jl:725-751: parameters.log, scheme dispatch),
# `generate_samples!` (src/gibbs.jl:897-1020, including the PSRF-driven top-up rounds) and `generate_samples_dbl!` (the "doubling"
# scheme, src/gibbs.jl:1051-1198) with the reference's keyword names, defaults and argument meaning, returning the reference's own
# `Results(state::Table, rhatξ::Table, rhatγ::Table, burn_in, sampled)` (src/gibbs.jl:23-29), so `Summary` and `show` of the
# package work unchanged.  No sampling logic lives here: every step is a `ccall`; what remains is the reference's schedule
# arithmetic (num2move, first_index), kept line for line with gibbs.jl:962-1013.
#
# Chains and GPUs: the chains of this process live on ONE GPU and advance as a lockstep group (bnr_group_run replaces the pmap
# over chains, gibbs.jl:946-948).  With several GPUs start one Julia worker per GPU (`addprocs(ngpu)`, worker w drives device
# w-1) and hand every worker the same `comm` description (see `rccl_comm`): chains are placed round-robin, (c-1) % world == rank,
# and `bnr_rhat` all-gathers the per-chain messages over the library's own RCCL communicator.
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: the build image has no `julia` binary.  The same C ABI, call for call, is exercised by
# the Python/ctypes mirror (bayesiannetworkregression.jl_amd/_capi.py, api.py) in tests/.
module BNRHip

using TypedTables, Random, Dates
import BayesianNetworkRegression: Results, citation

const LIB = get(ENV, "BNR_HIP_LIB", joinpath(@__DIR__, "..", "bayesiannetworkregression.jl_amd", "libbnr_hip.so"))

struct Hyper
    eta::Cdouble; zeta::Cdouble; iota::Cdouble; aDelta::Cdouble; bDelta::Cdouble; nu::Cdouble
end

lasterr() = unsafe_string(ccall((:bnr_last_error, LIB), Cstring, ()))
check(rc) = rc == 0 || error("libbnr_hip: $(lasterr()) (status $rc)")
# run calls: status 4 (BNR_ERR_SAMPLER_CAP) = a rejection sampler stopped at its attempt cap in this call; the rows were written and the table stays valid
# (the reference has no cap and never raises here): a fit must not die of one such draw -- warn and go on
check_run(rc) = rc == 4 ? (@warn("libbnr_hip: a rejection sampler hit its attempt cap in this run call (rows written; see bnr_chain_counters)"); true) : check(rc)

# element types the library converts on the device (enum of include/bnr_hip.h); anything else is promoted to Float64 here
dtype_code(::Type{Float64}) = 0
dtype_code(::Type{Bool}) = 1
dtype_code(::Type{UInt8}) = 1
dtype_code(::Type{Int32}) = 2
dtype_code(::Type{Int64}) = 3
dtype_code(::Type{Float32}) = 4
dtype_code(::Type) = -1

mutable struct Chain
    h::Ptr{Cvoid}
    n::Int; V::Int; R::Int; q::Int; tot::Int
end
destroy(ch::Chain) = ccall((:bnr_chain_destroy, LIB), Cint, (Ptr{Cvoid},), ch.h)

# X: the vector of n adjacency matrices (x_transform = true; setup_X!, gibbs.jl:239-247, runs on the device) or the n x q matrix
# X_new (gibbs.jl:917-918) in its own element type
function Chain(X, y::Vector{Float64}, R, tot_save, seed, c; x_transform=true, η=1.01, ζ=1.0, ι=1.0, aΔ=1.0, bΔ=1.0, ν=10, device=0)
    hy = Ref(Hyper(η, ζ, ι, aΔ, bΔ, ν))
    out = Ref{Ptr{Cvoid}}(C_NULL)
    if x_transform
        T = dtype_code(eltype(X[1])) >= 0 ? eltype(X[1]) : Float64
        mats = [Matrix{T}(m) for m in X]                       # column-major V x V each
        n, V = length(mats), size(mats[1], 1)
        ptrs = [Ptr{Cvoid}(pointer(m)) for m in mats]
        GC.@preserve mats ptrs y check(ccall((:bnr_chain_create_from_matrices, LIB), Cint,
            (Int32, Int32, Int32, Ptr{Ptr{Cvoid}}, Int32, Ptr{Cdouble}, Ref{Hyper}, UInt64, Int32, Int32, Int32, Ref{Ptr{Cvoid}}),
            n, V, R, ptrs, dtype_code(T), y, hy, UInt64(seed), c, device, tot_save, out))
    else
        T = dtype_code(eltype(X)) >= 0 ? eltype(X) : Float64
        Xm = Matrix{T}(X)
        n, q = size(Xm)
        V = Int64((-1 + sqrt(1 + 8 * q)) / 2)
        GC.@preserve Xm y check(ccall((:bnr_chain_create_typed, LIB), Cint,
            (Int32, Int32, Int32, Ptr{Cvoid}, Int32, Ptr{Cdouble}, Ref{Hyper}, UInt64, Int32, Int32, Int32, Ref{Ptr{Cvoid}}),
            n, V, R, Xm, dtype_code(T), y, hy, UInt64(seed), c, device, tot_save, out))
    end
    ch = Chain(out[], n, V, R, V * (V + 1) ÷ 2, tot_save)
    finalizer(destroy, ch)
    ch
end

# another chain of the same fit on the same GPU: X, y stay shared on the device (bnr_chain_create_like)
function chain_like(donor::Chain, seed, c, tot_save=donor.tot)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:bnr_chain_create_like, LIB), Cint, (Ptr{Cvoid}, UInt64, Int32, Int32, Ref{Ptr{Cvoid}}), donor.h, UInt64(seed), c, tot_save, out))
    ch = Chain(out[], donor.n, donor.V, donor.R, donor.q, tot_save)
    finalizer(destroy, ch)
    ch
end

init_prior!(ch::Chain) = check(ccall((:bnr_chain_init_prior, LIB), Cint, (Ptr{Cvoid},), ch.h))
# performance / path options of a chain (include/bnr_hip.h: "byte_x", "gram_i8", "graph", "graph_k", "overlap", ...): e.g. set_option!(ch, "gram_i8", 0) keeps
# the f64 Gram for a Bool model matrix (the i8 Gram is within 1e-12 max|G| of it, not bitwise)
set_option!(ch::Chain, name::AbstractString, value::Integer) = check(ccall((:bnr_chain_set_option, LIB), Cint, (Ptr{Cvoid}, Cstring, Int64), ch.h, name, value))
move_rows!(ch::Chain, to, from, count) = check(ccall((:bnr_chain_move_rows, LIB), Cint, (Ptr{Cvoid}, Int32, Int32, Int32), ch.h, to, from, count))
function resize_table!(ch::Chain, new_tot)
    check(ccall((:bnr_chain_resize, LIB), Cint, (Ptr{Cvoid}, Int32), ch.h, new_tot))
    ch.tot = new_tot
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

# run!(X,y,state,c,first_index,nburn,total,...,purge_burn,channel) (gibbs.jl:849-864) for every chain of the group / for one chain;
# `tick` is called every prog_freq iterations like the reference's put!(channel, true) of chain 1 (gibbs.jl:854-856)
const TICK = Ref{Any}(nothing)
tick_trampoline(::Ptr{Cvoid}, ::Int64) = (TICK[] === nothing || TICK[](); nothing)
function run!(x::Union{Group,Chain}, first_index, nburn, total, purge_burn; prog_freq=0, tick=nothing)
    nxt = Ref{Int32}(0)
    TICK[] = tick
    cb = tick === nothing ? C_NULL : @cfunction(tick_trampoline, Cvoid, (Ptr{Cvoid}, Int64))
    pb, pf = Int32(isnothing(purge_burn) ? 0 : purge_burn), Int32(tick === nothing ? 0 : prog_freq)
    # (the (name, library) target of a ccall must be a constant expression: two literal calls, not a symbol chosen at run time)
    if x isa Group
        check_run(ccall((:bnr_group_run, LIB), Cint, (Ptr{Cvoid}, Int32, Int32, Int32, Int32, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Int32}),
            x.h, first_index, nburn, total, pb, pf, cb, C_NULL, nxt))
    else
        check_run(ccall((:bnr_chain_run, LIB), Cint, (Ptr{Cvoid}, Int32, Int32, Int32, Int32, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Int32}),
            x.h, first_index, nburn, total, pb, pf, cb, C_NULL, nxt))
    end
    Int(nxt[])
end
prepare!(g::Group) = check(ccall((:bnr_group_prepare, LIB), Cint, (Ptr{Cvoid},), g.h))   # optional: captures the replayed graphs now

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

# Summary statistics on the device (gibbs.jl:1214-1250): (mean γ, lower, upper, P(ξ=1)); ranks are 1-based positions in the sorted sample
function summary_stats(ch::Chain, nburn, nsamp; interval=95)
    lb = (100 - interval) / 200
    klo, khi = Int(round(nsamp * lb)), Int(round(nsamp * (1 - lb)))
    m, lo, hi, p = zeros(ch.q), zeros(ch.q), zeros(ch.q), zeros(ch.V)
    check(ccall((:bnr_chain_summary, LIB), Cint, (Ptr{Cvoid}, Int32, Int32, Int32, Int32, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
        ch.h, nburn + 1, nsamp, klo, khi, m, lo, hi, p))
    m, lo, hi, p
end

# ---- the ranks of a fit (bnr_comm): nothing (one process), or the library's RCCL communicator.  Rank 0 calls `unique_id()`, the
# 128 bytes travel to the other workers by whatever connects them (e.g. `remotecall_fetch`), then EVERY rank calls `rccl_comm`.
struct Comm
    h::Ptr{Cvoid}; rank::Int; world::Int
end
function unique_id()
    id = zeros(UInt8, 128)
    check(ccall((:bnr_comm_unique_id, LIB), Cint, (Ptr{UInt8},), id))
    id
end
function rccl_comm(id::Vector{UInt8}, rank, world, device)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:bnr_comm_create_rccl, LIB), Cint, (Ptr{UInt8}, Int32, Int32, Int32, Ref{Ptr{Cvoid}}), id, rank, world, device, out))
    Comm(out[], rank, world)
end
close(c::Comm) = ccall((:bnr_comm_destroy, LIB), Cint, (Ptr{Cvoid},), c.h)

# return_psrf_VOI (gibbs.jl:771-789) over ALL chains of the fit, wherever they live: (rhatξ, rhatγ)
function psrf(chains::Vector{Chain}, num_chains, comm::Union{Comm,Nothing}, burn, nsamp, V, q)
    hs = Ptr{Cvoid}[ch.h for ch in chains]
    rx, rg = zeros(V), zeros(q)
    GC.@preserve hs check(ccall((:bnr_rhat, LIB), Cint, (Ptr{Ptr{Cvoid}}, Int32, Int32, Ptr{Cvoid}, Int32, Int32, Ptr{Cdouble}, Ptr{Cdouble}),
        hs, length(hs), num_chains, comm === nothing ? C_NULL : comm.h, burn, nsamp, rx, rg))
    rx, rg
end

# generate_samples!(X, y, R; ...) (gibbs.jl:897-1020).  `seed` must be the same on every rank (the reference draws it once, :928).
function generate_samples!(X, y, R; η=1.01, ζ=1.0, ι=1.0, aΔ=1.0, bΔ=1.0, ν=10, nburn=30000, nsamp=20000, maxburn=50000,
                           psrf_cutoff=1.2, x_transform=true, suppress_timer=false, num_chains=2, seed=nothing, purge_burn=nothing,
                           device=0, comm::Union{Comm,Nothing}=nothing, tick=nothing)
    V = x_transform ? size(X[1], 1) : Int64((-1 + sqrt(1 + 8 * size(X, 2))) / 2)
    q = floor(Int, V * (V + 1) / 2)
    total = nburn + nsamp
    prog_freq = 1000 >= nburn ? 10 : 1000                                        # :923-926
    if !isnothing(purge_burn) && (purge_burn < nburn) && purge_burn != 0         # :930-936
        if nburn % purge_burn != 0
            purge_burn = purge_burn - (nburn % purge_burn)
        end
    else
        purge_burn = nothing
    end
    tot_save = isnothing(purge_burn) ? total : nsamp + purge_burn
    # several ranks (comm given): every rank must run the SAME seed, or chain c on one rank is not the chain c the others assume -- refuse to draw one per rank
    isnothing(seed) && !isnothing(comm) && error("BNRHip: with a communicator the caller must pass ONE seed to every rank (draw it on rank 0 and broadcast it, gibbs.jl:739, 928)")
    seed = isnothing(seed) ? rand(1:55555) : seed
    rank, world = comm === nothing ? (0, 1) : (comm.rank, comm.world)
    ids = [c for c in 1:num_chains if (c - 1) % world == rank]                   # pmap's round-robin over workers
    yv = Vector{Float64}(y)
    chains = Chain[]
    for c in ids
        push!(chains, isempty(chains) ? Chain(X, yv, R, tot_save, seed, c; x_transform, η, ζ, ι, aΔ, bΔ, ν, device) :
                                        chain_like(chains[1], seed, c, tot_save))
    end
    foreach(init_prior!, chains)
    runner = length(chains) > 1 ? Group(chains) : (isempty(chains) ? nothing : chains[1])
    isnothing(runner) || run!(runner, 2, nburn, total, purge_burn; prog_freq, tick = 1 in ids ? tick : nothing)
    tot_generated = nburn + nsamp
    stt = isnothing(purge_burn) ? nburn : purge_burn
    rx, rg = psrf(chains, num_chains, comm, stt, nsamp, V, q)
    println(stderr, tot_generated, " samples generated. Max PSRF XI: ", round(maximum(rx), digits=2), ". Max PSRF Gamma: ", round(maximum(rg), digits=2))
    while (maximum(rx) > psrf_cutoff || maximum(rg) > psrf_cutoff) && tot_generated < (maxburn + nsamp)
        num2move = !isnothing(purge_burn) ? (nsamp + purge_burn <= nburn ? 1 : nsamp + purge_burn - nburn) : total - nburn   # :963-974
        tot_sze = tot_save
        for ch in chains                                                         # copy_table! loop :991-993
            move_rows!(ch, 1, tot_sze - num2move + 1, num2move)
        end
        last = num2move > 1 ? num2move + nburn : nburn                           # run! :997-999
        isnothing(runner) || run!(runner, num2move + 1, nburn > nsamp ? nburn - nsamp + num2move : 0, last, purge_burn;
                                  prog_freq, tick = 1 in ids ? tick : nothing)
        tot_generated = tot_generated + last - num2move
        rx, rg = psrf(chains, num_chains, comm, stt, nsamp, V, q)
        println(stderr, tot_generated, " samples generated. Max PSRF XI: ", round(maximum(rx), digits=3), ". Max PSRF Gamma: ", round(maximum(rg), digits=3))
    end
    state = 1 in ids ? fetch!(new_table(tot_save, V, R), chains[1]) : nothing    # only chain 1's trace is returned (gibbs.jl:788)
    Results(state, Table(ξ = rx), Table(γ = rg), stt, nsamp)
end

# generate_samples_dbl!(X, y, R; ...) (gibbs.jl:1051-1198): the "doubling" scheme -- a first pass of mingen iterations (half burn-in), then,
# while max Rhat > psrf_cutoff and fewer than maxgen iterations were generated, rounds that keep every sample so far, grow the table by
# mingen/2 rows (the reference allocates a new table and block-copies the tail, :1164-1172; here: move_rows! + resize_table! on the
# device) and run on from row num2move + 1.
function generate_samples_dbl!(X, y, R; η=1.01, ζ=1.0, ι=1.0, aΔ=1.0, bΔ=1.0, ν=10, mingen=10000, maxgen=100000, psrf_cutoff=1.01,
                               x_transform=true, suppress_timer=false, num_chains=2, seed=nothing, purge_burn=nothing,
                               device=0, comm::Union{Comm,Nothing}=nothing, tick=nothing)
    if ν == R
        println("Warning: ν==R may give poor accuracy. Consider increasing ν")                      # :1057-1059
    end
    nburn = convert(Int64, round(mingen / 2))                                                      # :1061-1062
    nsamp = mingen - nburn
    V = x_transform ? size(X[1], 1) : Int64((-1 + sqrt(1 + 8 * size(X, 2))) / 2)
    q = floor(Int, V * (V + 1) / 2)
    total = nburn + nsamp
    prog_freq = 1000 >= nburn ? 10 : 1000
    if !isnothing(purge_burn) && (purge_burn < nburn) && purge_burn != 0
        if nburn % purge_burn != 0
            purge_burn = purge_burn - (nburn % purge_burn)
        end
    else
        purge_burn = nothing
    end
    tot_save = isnothing(purge_burn) ? total : nsamp + purge_burn
    # several ranks (comm given): every rank must run the SAME seed, or chain c on one rank is not the chain c the others assume -- refuse to draw one per rank
    isnothing(seed) && !isnothing(comm) && error("BNRHip: with a communicator the caller must pass ONE seed to every rank (draw it on rank 0 and broadcast it, gibbs.jl:739, 928)")
    seed = isnothing(seed) ? rand(1:55555) : seed
    rank, world = comm === nothing ? (0, 1) : (comm.rank, comm.world)
    ids = [c for c in 1:num_chains if (c - 1) % world == rank]
    yv = Vector{Float64}(y)
    chains = Chain[]
    for c in ids
        push!(chains, isempty(chains) ? Chain(X, yv, R, tot_save, seed, c; x_transform, η, ζ, ι, aΔ, bΔ, ν, device) :
                                        chain_like(chains[1], seed, c, tot_save))
    end
    foreach(init_prior!, chains)
    runner = length(chains) > 1 ? Group(chains) : (isempty(chains) ? nothing : chains[1])
    mytick = 1 in ids ? tick : nothing
    isnothing(runner) || run!(runner, 2, nburn, total, purge_burn; prog_freq, tick = mytick)
    tot_generated = nburn + nsamp
    tot_samples = nsamp
    stt = isnothing(purge_burn) ? nburn : purge_burn
    rx, rg = psrf(chains, num_chains, comm, stt, nsamp, V, q)
    println(stderr, tot_generated, " samples generated. Max PSRF XI: ", round(maximum(rx), digits=3), ". Max PSRF Gamma: ", round(maximum(rg), digits=3))
    bad(a, b) = maximum(a) > psrf_cutoff || maximum(b) > psrf_cutoff || isnan(maximum(a)) || isnan(maximum(b))
    while bad(rx, rg) && tot_generated < maxgen                                                    # :1119
        halfburn = convert(Int64, round(mingen / 2))
        num2move = tot_samples                                                                     # :1143
        tot_samples = tot_samples + halfburn
        nsamp = tot_samples
        tot_sze = tot_save
        tot_save = tot_samples + halfburn
        println(stderr, "num2move: ", num2move, " nburn: ", nburn, " nsamp: ", nsamp, " tot_save: ", tot_save, " first_index: ", num2move + 1)
        for ch in chains                                                                           # new table + copy_table! of the tail (:1164-1172)
            move_rows!(ch, 1, tot_sze - num2move + 1, num2move)
            resize_table!(ch, tot_save)
        end
        isnothing(runner) || run!(runner, num2move + 1, 0, tot_save, purge_burn; prog_freq, tick = mytick)   # run! :1176-1178
        tot_generated = tot_generated + mingen
        rx, rg = psrf(chains, num_chains, comm, stt, nsamp, V, q)
        println(stderr, tot_generated, " samples generated. Max PSRF XI: ", round(maximum(rx), digits=3), ". Max PSRF Gamma: ", round(maximum(rg), digits=3))
    end
    state = 1 in ids ? fetch!(new_table(tot_save, V, R), chains[1]) : nothing
    Results(state, Table(ξ = rx), Table(γ = rg), stt, nsamp)
end

# Fit!(X, y, R; ...) (gibbs.jl:725-751): writes parameters.log (timestamp, the package's citation, every keyword, the seed -- rank 0 only
# when the fit spans several ranks), draws the seed when none is given, then the doubling scheme if mingen > 0 and maxgen > 0, else the
# traditional one with maxburn = nburn + nsamples.  `V` is accepted and ignored, as in the reference.
function Fit!(X, y, R; η=1.01, V=30, ζ=1.0, ι=1.0, aΔ=1.0, bΔ=1.0, ν=10, nburn=30000, nsamples=20000, mingen=0, maxgen=0,
              psrf_cutoff=1.01, x_transform=true, suppress_timer=false, num_chains=2, seed=nothing, purge_burn=nothing,
              filename="parameters.log", device=0, comm::Union{Comm,Nothing}=nothing, tick=nothing)
    # several ranks (comm given): every rank must run the SAME seed, or chain c on one rank is not the chain c the others assume -- refuse to draw one per rank
    isnothing(seed) && !isnothing(comm) && error("BNRHip: with a communicator the caller must pass ONE seed to every rank (draw it on rank 0 and broadcast it, gibbs.jl:739, 928)")
    seed = isnothing(seed) ? rand(1:55555) : seed                # (with several ranks the caller passes ONE seed to all of them, see generate_samples!)
    if comm === nothing || comm.rank == 0
        open(filename, "w") do logfile
            write(logfile, "BayesianNetworkRegression.jl Fit! function\n")
            write(logfile, Dates.format(Dates.now(), "yyyy-mm-dd H:M:S.s") * "\n")
            write(logfile, citation(returnstring=true))
            write(logfile, "\n\nParameters:\n")
            str = "R=$R, η=$η, ζ=$ζ, ι=$ι, aΔ=$aΔ, bΔ=$bΔ, ν=$ν, nburn=$nburn, nsamples=$nsamples, \n"
            str *= "mingen=$mingen, maxgen=$maxgen, psrf_cutoff=$psrf_cutoff, \n"
            str *= "x_transform=$x_transform, suppress_timer=$suppress_timer, num_chains=$num_chains, purge_burn=$purge_burn \n"
            str *= "seed=$seed"
            write(logfile, str)
        end
    end
    if (mingen > 0) && (maxgen > 0)
        generate_samples_dbl!(X, y, R; η, ζ, ι, aΔ, bΔ, ν, mingen, maxgen, psrf_cutoff, x_transform, suppress_timer, num_chains, seed, purge_burn,
                              device, comm, tick)
    else
        generate_samples!(X, y, R; η, ζ, ι, aΔ, bΔ, ν, nburn, nsamp = nsamples, maxburn = nburn + nsamples, psrf_cutoff, x_transform,
                          suppress_timer, num_chains, seed, purge_burn, device, comm, tick)
    end
end

end # module
