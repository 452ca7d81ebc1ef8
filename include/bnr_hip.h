/*
 * bnr_hip.h -- C ABI of libbnr_hip.so: the MI355X (gfx950) implementation of the Gibbs hot path of
 * BayesianNetworkRegression.jl (reference: src/gibbs.jl + src/gig.jl + src/convergence.jl).
 *
 * The reference is pure Julia and has no FFI seam; this header IS the seam a maintainer would bind with
 * `ccall` (julia/BNRHip.jl) and that the Python host mirror binds with ctypes
 * (bayesiannetworkregression.jl_amd/_capi.py).  Every entry point names the reference code it replaces.
 *
 * Conventions
 *   - plain C types only; no exceptions cross the ABI; every call returns an int status (BNR_OK == 0);
 *     bnr_last_error() gives a thread-local message for the last non-zero status.
 *   - all floating point data is IEEE double, as in the reference (gibbs.jl:835-841).
 *   - host matrices are COLUMN-MAJOR (Julia `Matrix` memory): X is n x q with q = V(V+1)/2 columns in the
 *     column-wise lower-triangle order of utils.jl:50-55.
 *   - state tables handed to bnr_chain_fetch/bnr_chain_load use the reference layout
 *     Array{Float64,3}(tot_save,d1,d2), column-major, ITERATION INDEX FASTEST.
 *   - row/iteration indices are 1-based exactly as in run! (gibbs.jl:849-864) unless stated otherwise.
 *   - a handle is used by one host thread at a time; different handles may be driven concurrently.
 *   - the caller keeps ownership of every host buffer; the library owns all device memory of a chain until
 *     bnr_chain_destroy.
 */
#ifndef BNR_HIP_H
#define BNR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BNR_ABI_VERSION 6   /* 2: + bnr_chain_create_like, bnr_group_*, bnr_chain_summary; 3: + bnr_*_prepare; 4: + bnr_comm_*, bnr_rhat;
                               5: + bnr_chain_create_typed, bnr_chain_create_from_matrices, bnr_device_synchronize; 6: + bnr_comm_info (all additive) */

enum {
    BNR_OK = 0,
    BNR_ERR_BAD_ARG = 1,
    BNR_ERR_HIP = 2,          /* a HIP runtime call failed (no device, out of memory, launch failure ...) */
    BNR_ERR_CHOLESKY = 3,     /* Cholesky failed after the jitter ladder (the reference rethrows: gibbs.jl:337-343) */
    BNR_ERR_SAMPLER_CAP = 4   /* a rejection sampler hit its attempt cap: returned ONCE, by the call in which it happened; the rows were written with
                               * the samplers' fall-backs, the table stays valid, later calls on the chain are not failed for it (the counter
                               * of bnr_chain_counters keeps the total).  The reference's rejection loops are unbounded and never raise here. */
};

typedef struct bnr_chain bnr_chain;   /* opaque: one Gibbs chain resident on one GPU */

/* hyper-parameters of Fit!/generate_samples! (gibbs.jl:725-727, 897-899) */
typedef struct {
    double eta, zeta, iota, aDelta, bDelta, nu;
} bnr_hyper;

/* progress callback: called from the calling host thread every prog_freq iterations
 * (the reference's `put!(channel,true)`, gibbs.jl:854-856).  `done` = iterations completed in this call. */
typedef void (*bnr_progress_cb)(void *user, int64_t done);

int bnr_abi_version(void);
const char *bnr_last_error(void);
int bnr_device_count(int *count);
int bnr_device_synchronize(int32_t device);          /* hipDeviceSynchronize on `device` (hosts that hold no HIP binding of their own) */
int bnr_runtime_version(int *version);               /* hipRuntimeGetVersion of the HIP runtime the library is bound to */

/* Allocate a chain: copies X (n x q col-major) and y to HBM, allocates the tot_save-row state table and all
 * work space on `device`.  Replaces the allocation half of initialize_and_run! (gibbs.jl:822-841).
 * The RNG stream is keyed by seed + chain_id (the reference's Xoshiro(seed+c), gibbs.jl:928).
 * Limits (BNR_ERR_BAD_ARG otherwise): 1 <= R <= 32, V >= 2, n <= 14000 (LDS budgets of single-workgroup kernels; round 5: R*V is no longer limited --
 * beyond 15360 the scalar tail reads u from the table row instead of staging it in LDS). */
int bnr_chain_create(int32_t n, int32_t V, int32_t R, const double *X, const double *y, const bnr_hyper *hyper,
                     uint64_t seed, int32_t chain_id, int32_t device, int32_t tot_save, bnr_chain **out);
/* The same with the model matrix in the caller's own element type -- the reference builds X_new as Matrix{eltype(T)} (gibbs.jl:917:
 * Bool for 0/1 adjacency data, Int, Float64 ...) -- uploaded as it is (a Bool/UInt8 matrix is 1/8 of the PCIe traffic) and converted
 * to the f64 device layout by a kernel.  X: n x q column-major elements of x_dtype. */
enum { BNR_F64 = 0, BNR_U8 = 1 /* Julia Bool / UInt8 */, BNR_I32 = 2, BNR_I64 = 3 /* Julia Int */, BNR_F32 = 4 };
int bnr_chain_create_typed(int32_t n, int32_t V, int32_t R, const void *X, int32_t x_dtype, const double *y, const bnr_hyper *hyper,
                           uint64_t seed, int32_t chain_id, int32_t device, int32_t tot_save, bnr_chain **out);
/* ... and straight from the vector of n adjacency matrices (x_transform = true: setup_X!, gibbs.jl:239-247, runs on the device):
 * A[i] points at a V x V column-major matrix of x_dtype; row i of the model matrix is lower_triangle(A[i]) (utils.jl:40-57: the
 * column-wise lower triangle including the diagonal, reading A[i][l, k] for l >= k -- the matrices need not be symmetric). */
int bnr_chain_create_from_matrices(int32_t n, int32_t V, int32_t R, const void *const *A, int32_t x_dtype, const double *y,
                                   const bnr_hyper *hyper, uint64_t seed, int32_t chain_id, int32_t device, int32_t tot_save,
                                   bnr_chain **out);
/* Another chain of the same fit on the same device: same X, y, sizes and hyper-parameters as `donor`, own seed /
 * chain id, own table and work space.  The read-only device inputs (X, y, index maps) are SHARED with the donor, not
 * copied -- the reference hands the same X, y to every pmap worker (gibbs.jl:946-948) -- and live until the last chain
 * using them is destroyed. */
int bnr_chain_create_like(const bnr_chain *donor, uint64_t seed, int32_t chain_id, int32_t tot_save, bnr_chain **out);
int bnr_chain_destroy(bnr_chain *chain);

/* initialize_variables! (gibbs.jl:191-224): draws row 1 from the priors; sets the iteration counter to 1. */
int bnr_chain_init_prior(bnr_chain *chain);

/* run! (gibbs.jl:849-864): for i in first_index:total { gibbs_sample!(row j); purge ring }.
 * purge_burn <= 0 means `nothing`.  Synchronous.  *next_row (optional) receives the row index j the next
 * call would write.  cb may be NULL. */
int bnr_chain_run(bnr_chain *chain, int32_t first_index, int32_t nburn, int32_t total, int32_t purge_burn,
                  int32_t prog_freq, bnr_progress_cb cb, void *user, int32_t *next_row);
/* same, but only enqueues the work on the chain's stream(s); pair with bnr_chain_sync.  Lets several chains
 * that share one GPU overlap (the reference runs chains concurrently under pmap, gibbs.jl:946). */
int bnr_chain_run_async(bnr_chain *chain, int32_t first_index, int32_t nburn, int32_t total, int32_t purge_burn);
int bnr_chain_sync(bnr_chain *chain, int32_t *next_row);
/* Optional: build everything the run loop replays (the captured hipGraphs of graph_k sweeps and of one sweep) now, so that the
 * first bnr_chain_run / bnr_group_run call is already steady state.  run! has no counterpart (gibbs.jl:849-864 just loops);
 * calling it is never required -- the first run call does the same lazily -- and never changes results. */
int bnr_chain_prepare(bnr_chain *chain);

/* Lockstep group: the chains one `pmap` call of generate_samples! hands to the workers (gibbs.jl:946-948, 989-1000)
 * when several of them live on ONE GPU.  All members (equal n, V, R and table length, same device) advance together:
 * bnr_group_run is run! (gibbs.jl:849-864) for every member with the same (first_index, nburn, total, purge_burn),
 * each kernel of a sweep being launched once for the whole group.  Members stay independent chains (own seed + chain
 * id, own table); their tables are bitwise what bnr_chain_run would have produced for each of them alone.  The group
 * does not own its members; destroying a member dissolves the group (its handle stays valid until bnr_group_destroy,
 * bnr_group_run then fails).  cb ticks like chain 1's callback. */
typedef struct bnr_group bnr_group;
int bnr_group_create(bnr_chain *const *chains, int32_t nchains, bnr_group **out);
int bnr_group_destroy(bnr_group *group);
int bnr_group_run(bnr_group *group, int32_t first_index, int32_t nburn, int32_t total, int32_t purge_burn,
                  int32_t prog_freq, bnr_progress_cb cb, void *user, int32_t *next_row);
/* options "graph", "graph_k", "overlap", "profiling" as for bnr_chain_set_option / bnr_chain_set_profiling; timings as
 * bnr_chain_last_timing (which = 1: one k_gram launch covers all members) */
int bnr_group_prepare(bnr_group *group);   /* as bnr_chain_prepare */
int bnr_group_set_option(bnr_group *group, const char *name, int64_t value);
int bnr_group_last_timing(bnr_group *group, int32_t which, double *avg_us, int64_t *launches);

/* gibbs_sample!(state, row, ...) (gibbs.jl:663-677) for one row, and the ten update_*! functions in sweep
 * order (gibbs.jl:267-636): test hooks mirroring test/init-tests.jl:76,96-124.  `row` is 1-based (>= 2);
 * `iter` is the global iteration id that keys the RNG counter for this row. */
int bnr_gibbs_step(bnr_chain *chain, int32_t row, int64_t iter);
int bnr_update_tau2(bnr_chain *chain, int32_t row, int64_t iter);
int bnr_update_u_xi(bnr_chain *chain, int32_t row, int64_t iter);
int bnr_update_gamma(bnr_chain *chain, int32_t row, int64_t iter);
int bnr_update_D(bnr_chain *chain, int32_t row, int64_t iter);
int bnr_update_theta(bnr_chain *chain, int32_t row, int64_t iter);
int bnr_update_Delta(bnr_chain *chain, int32_t row, int64_t iter);
int bnr_update_M(bnr_chain *chain, int32_t row, int64_t iter);
int bnr_update_mu(bnr_chain *chain, int32_t row, int64_t iter);
int bnr_update_Lambda(bnr_chain *chain, int32_t row, int64_t iter);
int bnr_update_pi(bnr_chain *chain, int32_t row, int64_t iter);

/* iteration counter (global id of the last drawn row; init row = 1) */
int bnr_chain_get_iter(bnr_chain *chain, int64_t *iter);
int bnr_chain_set_iter(bnr_chain *chain, int64_t iter);

/* Copy rows first_row..last_row (1-based, inclusive) of the device table into / out of host arrays laid out as
 * the reference's state Table: 11 live columns tau2(.,1,1) u(.,R,V) xi(.,V,1) gamma(.,q,1) S(.,q,1) theta Delta
 * M(.,R,R) mu lam(.,R,1) pi(.,R,3), each Array{Float64,3}(host_tot,d1,d2).  Host row = device row + host_row_offset.
 * The 3 dead columns of the reference (Sigma^-1, invC, mu_t: allocated, never written, utils.jl:72-84) are the
 * caller's business.  Any column pointer may be NULL (skipped). */
int bnr_chain_fetch(bnr_chain *chain, int32_t first_row, int32_t last_row, int32_t host_tot, int32_t host_row_offset,
                    double *tau2, double *u, double *xi, double *gamma, double *S, double *theta, double *Delta,
                    double *M, double *mu, double *lam, double *pi);
int bnr_chain_load(bnr_chain *chain, int32_t first_row, int32_t last_row, int32_t host_tot, int32_t host_row_offset,
                   const double *tau2, const double *u, const double *xi, const double *gamma, const double *S,
                   const double *theta, const double *Delta, const double *M, const double *mu, const double *lam,
                   const double *pi);

/* copy_table!(table, to, from) for `count` consecutive rows on the device (utils.jl:72-84; used by the
 * continuation loops gibbs.jl:991-993, 1172).  Overlap-safe. */
int bnr_chain_move_rows(bnr_chain *chain, int32_t to_row, int32_t from_row, int32_t count);
/* Re-allocate the device table with new_tot rows, keeping rows 1..min(old,new) (generate_samples_dbl!
 * allocates a new table per round, gibbs.jl:1164-1172). */
int bnr_chain_resize(bnr_chain *chain, int32_t new_tot);

/* split-Rhat, first half of rhat() (convergence.jl:4-65) done on the device for ONE chain: for every parameter
 * p of gamma (q) then xi (V) over rows first_row..first_row+nsamp-1, the mean and the corrected variance of the
 * first floor(nsamp/2) and the last floor(nsamp/2) samples.  stats (host): 4*(q+V) doubles, laid out
 * [mean_h0(q+V) | var_h0(q+V) | mean_h1(q+V) | var_h1(q+V)].  This is the per-chain message that is
 * all-gathered across ranks (RCCL via torch.distributed in the host layer). */
int bnr_chain_rhat_stats(bnr_chain *chain, int32_t first_row, int32_t nsamp, double *stats);
/* Summary(results) (gibbs.jl:1214-1250) computed on the device over rows first_row .. first_row+nsamp-1 of this chain's
 * table: mean_gamma[q] = posterior mean of every edge coefficient, lower[q] / upper[q] = the k_lo-th / k_hi-th smallest
 * sample of every edge (1-based; the reference takes sort(gamma)[round(nsamp*lower_bound)] and [round(nsamp*upper_bound)]),
 * prob_xi[V] = posterior mean of xi.  Exact selection; only 3q + V doubles cross PCIe.  Rounding to `digits` is the caller's. */
int bnr_chain_summary(bnr_chain *chain, int32_t first_row, int32_t nsamp, int32_t k_lo, int32_t k_hi,
                      double *mean_gamma, double *lower, double *upper, double *prob_xi);

/* Effective sample size -- an ADDITION to the reference (which only has split-Rhat; north-star item "Rhat/ESS check").
 * bnr_chain_ess_stats: this chain's message over rows first_row .. first_row+nsamp-1: for both halves of the window (the
 *   halves of split-Rhat) the mean, the variance and the autocovariances at lags 0..max_lag-1 of gamma (q) then xi (V):
 *   stats[half][2 + max_lag][q + V], computed on the device.
 * bnr_ess_from_stats: bulk ESS over the messages of all chains ([chain][half][2 + max_lag][nparams]) with the estimator of
 *   Stan / MCMCDiagnosticTools.ess (split chains, Geyer's initial positive and monotone sequence), truncated at max_lag;
 *   NaN for a constant parameter.  ess[nparams]. */
int bnr_chain_ess_stats(bnr_chain *chain, int32_t first_row, int32_t nsamp, int32_t max_lag, double *stats);
int bnr_ess_from_stats(const double *stats, int32_t nchains, int32_t nparams, int32_t nsamp, int32_t max_lag, double *ess);

/* The convergence check over ALL chains of a fit, wherever they live (return_psrf_VOI, gibbs.jl:771-789, over rhat(),
 * convergence.jl:4-65; the reference's master receives whole state tables from its pmap workers, gibbs.jl:946-957 -- here only
 * 4 (q + V) doubles per chain travel).  Chains c = 1..nchains_total are placed round-robin: rank r holds, in increasing c, the
 * chains with (c - 1) % world == r.  bnr_rhat reduces every local chain's window rows burn+1 .. burn+nsamp on the device,
 * all-gathers the messages through `comm`, and finishes the same rhat_xi[V] / rhat_gamma[q] on every rank.
 * comm == NULL: one process holds all chains.
 * A communicator is either RCCL (ncclAllGather over xGMI on the library's own communicator; librccl.so is bound at run time, so
 * single-GPU use needs no RCCL) -- rank 0 calls bnr_comm_unique_id and the host carries the 128 bytes to the other ranks by
 * whatever connects them already (Julia Distributed, torch.distributed's store, MPI), then every rank calls bnr_comm_create_rccl
 * (a collective call) -- or a callback the host implements (an all-gather of `count` doubles per rank into recv[world * count] in
 * rank order, returning 0; used by the gloo tests and by hosts that bring their own transport). */
typedef struct bnr_comm bnr_comm;
typedef struct { char bytes[128]; } bnr_unique_id;                       /* ncclUniqueId */
typedef int (*bnr_allgather_fn)(void *ctx, const double *send, double *recv, int64_t count);
int bnr_comm_unique_id(bnr_unique_id *id);
int bnr_comm_create_rccl(const bnr_unique_id *id, int32_t rank, int32_t world, int32_t device, bnr_comm **out);
int bnr_comm_create_callback(int32_t rank, int32_t world, bnr_allgather_fn fn, void *ctx, bnr_comm **out);
int bnr_comm_destroy(bnr_comm *comm);
int bnr_comm_allgather(bnr_comm *comm, const double *send, double *recv, int64_t count);   /* host buffers; comm NULL = copy */
/* What the transport itself reports (any out pointer may be NULL): kind 0 = no communicator (one rank), 1 = RCCL, 2 = host callback;
 * rank / world as given at creation; rccl_ranks / rccl_rank = ncclCommCount / ncclCommUserRank of the library's communicator (0 / -1
 * unless kind 1) -- the number of ranks RCCL really connected, which a multi-GPU run reports beside its throughput (the reference's
 * counterpart is nworkers() after addprocs, gibbs.jl:946-948). */
int bnr_comm_info(bnr_comm *comm, int32_t *kind, int32_t *rank, int32_t *world, int32_t *rccl_ranks, int32_t *rccl_rank);
int bnr_rhat(bnr_chain *const *chains, int32_t nchains_local, int32_t nchains_total, bnr_comm *comm, int32_t burn, int32_t nsamp,
             double *rhat_xi, double *rhat_gamma);

/* second half: combine nchains messages (host arrays, chain-major) into Rhat per parameter.  Pure host code.
 * rhat: q+V doubles (gamma first, then xi).  Replaces convergence.jl:49-61. */
int bnr_rhat_from_stats(const double *stats, int32_t nchains, int32_t nparams, int32_t nsamp, double *rhat);

/* event counters: out[0]=Cholesky jitter events, out[1]=NaN-weight events (always 0: log-space weights),
 * out[2]=sampler attempt-cap events, out[3]=Cholesky hard failures, out[4..7] reserved */
int bnr_chain_counters(bnr_chain *chain, int64_t out[8]);

/* kernel timing: average device time in microseconds of the kernels of the last bnr_chain_run call, measured
 * with HIP events on the chain's own stream.  which: 0 = whole iteration, 1 = Gram kernel (X diag(S) X');
 * (which = 3 / 4: is a byte image of X in use / does the Gram run on the i8 pipe, see bnr_chain_set_option.)
 * which = 2 reports how the last run call was issued: *launches = sweeps replayed from captured graphs, *avg_us = sweeps
 * launched eagerly (a steady-state run is all replay). */
int bnr_chain_set_profiling(bnr_chain *chain, int32_t enable);
int bnr_chain_last_timing(bnr_chain *chain, int32_t which, double *avg_us, int64_t *launches);

/* diagnostics: in-kernel cycle stamps of a -DBNR_STAMPS build (zeros otherwise) */
int bnr_chain_debug_read(bnr_chain *chain, uint64_t *out, int32_t count);
/* diagnostics: average duration of `reps` back-to-back launches of the Gram kernel on the chain's stream */
int bnr_chain_debug_time_gram(bnr_chain *chain, int32_t reps, double *avg_us);
/* diagnostics: copy an internal work buffer to the host (0 = factorization matrix E, 1 = rhs b, 2 = a4, 3 = Gram partials) */
int bnr_chain_debug_copy(bnr_chain *chain, int32_t which, double *out, int64_t count);
/* diagnostics: internal sizes {n_pad, q_pad, ksplit (K slices = planes of Gram partials), ntile (64-row tiles), kcp, kslab, i8L (i8 Gram: padded K slice,
 * bytes per mask row, digit planes), rowlen (doubles per trace row on the device)}; the Gram partials of debug_copy(3) are [ksplit][ntile (ntile + 1) / 2][64 x 64],
 * tile (ti >= tj) at index ti (ti + 1) / 2 + tj, element (i, j) of a tile at [j * 64 + i] */
int bnr_chain_debug_dims(bnr_chain *chain, int32_t *out8);
/* timing experiments of round 4 (removed in round 5, tools/experiments/README.md): the entry point remains and returns BNR_ERR_BAD_ARG.  (flags bit 0 made the
 * kernels of the scalar branch return at once -- what the critical chain costs without company.) */
int bnr_debug_set_exp(int32_t device, int32_t flags);

/* tunables (performance only; never change results):
 *   "graph"     1 (default): replay captured hipGraphs of graph_k sweeps; 0: launch every kernel eagerly
 *   "graph_k"   sweeps per captured graph (default 16; a ladder graph_k, graph_k / 2, ..., 1 is captured so that a batch of any length is pure replay)
 *   "overlap"   1 (default): scalar branch and Gram/factorization branch of a sweep on two streams; 0: one stream
 *   "gram_variant" 0 (default): the Gram kernel is chosen per launch (k_gram8 when the launch has more than two workgroups per CU,
 *               k_gram otherwise); 8 / 16 force one of them.  Both write the same partial tiles bit for bit.  (9..14: the persistent /
 *               resident experiments of rounds 3-4, removed from the tree in round 5 -- tools/experiments/README.md -- and refused by name.)
 *   "profiling" 1: record HIP events around every k_gram launch (forces eager launches), see bnr_chain_last_timing
 *   "factor_variant" -1 (default): chosen by size -- 0 below n_pad = 1024, 3 from there on; 0: right-looking factorization, one
 *               32-column panel per launch (k_gram_reduce + k_chol_step); 2: right-looking, two panels per launch (k_chol_step2); 3: 2 with
 *               the whole trailing matrix updated at every other launch only (K = 128); (1: left-looking k_chol_ll, 4: data-flow k_chol_df, 5: one workgroup per
 *               chain k_chol_small -- experiments of rounds 3-4, removed in round 5 and refused)
 *   "fuse_reduce" 1 / -1 (default): launch 0 of the one-panel factorization also sums the Gram's K-split partial tiles (no k_gram_reduce
 *               launch); 0: separate reduction pass
 *   "group_xpass" -1 (default): a lockstep group whose members share the device copy of X (bnr_chain_create_like) runs ONE X pass
 *               for all members (k_xpass_group) when X has 8 MB or more per chain; 1: always; 0: one pass per member
 *   "split_sums" -1 (default): a chain run alone computes the back-projection's partial sums (update_theta!, update_Lambda!) in a launch of
 *               their own in front of the scalar tail, off the critical chain; 1: always; 0: inside the back-projection
 *   "spw_cap"   1..4 (default 1 since round 6: one block each beside the pipelined panel sweep; rounds 3-5: 4): super blocks per update workgroup of the factorization, at most
 *   "tail_after" / "node_after" (round 6): WHEN the scalar branch of a sweep (k_tail: theta, mu, Lambda, pi, the next tau2; k_node: tau2, u, xi; the X pass; k_rhs) starts
 *               relative to the Gram / factorization branch it runs beside.  "tail_after" 0: k_tail of the previous sweep waits for the Gram; -1: it starts as soon as the
 *               dispatcher lets it (rounds 1-5).  "node_after" p >= 0: k_node waits for launch number p of the factorization; -1: follows k_tail at once.  -2 (default):
 *               chosen from a size model -- ordered where the factorization is the longer chain by a margin (n = 500, V = 100; n = 2000), free-running for small n or
 *               large q.  Graph edges between the two branches; the tables do not depend on them.
 *   "wide_backproj" -1 (default): launches of the back-projection with 8 x CUs or more chunks of 32 edges (a lockstep group at large q) run k_backproj64 --
 *               a workgroup owns 64 edges, its drawing wave one edge per lane and the reference's own attempt loop (fewer instructions per edge; launches of one or
 *               two rounds of workgroups keep k_backproj, whose draws have the shorter latency); 1: always; 0: never.  Bitwise the same tables.
 *   Experiments ("nop_fork", "pipeline", "gate_us", "linear", "linear_merge", "linear_debug", "group_backproj", "resv_mask", "crit_origin"; rounds 3-4,
 *               profiles/round*_experiments_notes.txt): all measured no faster, part of them polled device memory.  Removed from the tree in
 *               round 5 (tools/experiments/README.md): the library refuses them by name.
 *   "byte_x"    (chains only) 0: the X passes read the f64 matrix although a byte image of X exists; 1 (default): the byte image
 *               (kept when the model matrix came as Bool/UInt8, or as Int32/Int64 with every value in 0..255; docs/src/man/inputdata.md)
 *   "gram_i8"   (chains only; round 5) 1 (the default from n_pad^2 q >= 2.5e8 on, where it was measured faster -- n = 500, V = 100 and
 *               larger; smaller problems default to 0): the model matrix came integer-typed with every entry 0 or 1
 *               (the reference's adjacency data, docs/src/man/inputdata.md:5-10) -- its Gram X diag(S) X' (gibbs.jl:434) runs on the i8 matrix pipe:
 *               S as i8L = 7 or 8 planes of balanced base-256 digits under the exponent of its largest entry (k_sdigits), one exact i32 Gram per plane
 *               (k_gram_i8, v_mfma_i32_16x16x64_i8), recombined in f64.  |G_i8 - G_exact| <= 8 q 2^(-8 i8L) max S <= 1e-12 max |G|; the tables
 *               agree with the f64 Gram's to that size of perturbation (NOT bit for bit) and with the oracle to the same 1e-6 as everything
 *               else.  0: the f64 Gram also for a binary matrix.  A matrix that is not binary has no i8 path (setting 1 is refused).
 *               bnr_chain_last_timing(which = 4): *avg_us = 1 when the chain's (its group's) Gram runs on the i8 pipe, *launches = i8L.
 * All variants except "gram_i8" give the same tables bit for bit.  bnr_chain_last_timing(which = 3) says whether a byte image is in use. */
int bnr_chain_set_option(bnr_chain *chain, const char *name, int64_t value);

/* Host-side copies of the draw-site primitives (same source as the device functions), exported so that the
 * CPU test-suite can check the product's RNG contract against the oracle without a GPU. */
void bnr_host_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void bnr_host_uniform2(uint64_t seed, uint32_t it, uint32_t site, uint32_t elem, uint32_t att, double out[2]);
double bnr_host_normal(uint64_t seed, uint32_t it, uint32_t site, uint32_t elem, uint32_t att);
double bnr_host_gamma(uint64_t seed, double shape, uint32_t it, uint32_t site, uint32_t elem);
double bnr_host_gig(uint64_t seed, double lambda, double chi, double psi, uint32_t it, uint32_t elem);
int32_t bnr_host_edge_index(int32_t V, int32_t l, int32_t k);   /* 0-based (l,k) -> 0-based e; utils.jl:50-55 */
/* The K split the library would choose for the Gram  X diag(S) X'  of gibbs.jl:434 on a device with `ncu` compute units (no GPU needed):
 * out[0] = K slices, out[1] = columns per slice (padded), out[2] = q_pad, out[3] = MiB of X one K-group of a workgroup addresses through its
 * 2 GiB buffer window.  The split is raised beyond what fills the chip until that span fits the window (n up to the 14 000-row limit with any V
 * the device holds); BNR_ERR_BAD_ARG when no split of at most 4096 slices does. */
int bnr_host_gram_plan(int32_t n, int32_t V, int32_t ncu, int32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* BNR_HIP_H */
