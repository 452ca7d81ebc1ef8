// bnr_hip.hip -- C ABI (include/bnr_hip.h) over the gfx950 kernels in bnr_kernels.h.
// Host side: chain allocation (own or shared device inputs), the run! loop with the purge ring (gibbs.jl:849-864) for one
// chain or for a lockstep group of chains (one launch per kernel for all members), the test hooks that mirror the
// reference's update_*! functions, table fetch/load, the split-Rhat / ESS messages and the device-side Summary.
#include "../../include/bnr_hip.h"
#include "bnr_kernels.h"

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
// HIP calls inside the launch helpers (event records, stream waits): their first failure is kept and reported by the next
// check_launch(), like a failed kernel launch
static thread_local hipError_t g_noted = hipSuccess;
static thread_local const char *g_noted_what = nullptr;
#define HIPNOTE(expr)                                                                                  \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess && g_noted == hipSuccess) { g_noted = _e; g_noted_what = #expr; }         \
    } while (0)
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return fail(BNR_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));               \
    } while (0)

// Where and how sweeps are issued: ONE chain, or a lockstep group of equally shaped chains whose kernels are launched
// together (blockIdx.z = member).  Kernels read their chain's bnr_dev from the device array `cds`.
struct bnr_exec {
    int device = 0;
    int ncu = 256;                                      // compute units of the device
    int nb = 1;                                         // chains issued together
    bnr_dev *cds = nullptr;                             // device array of nb structs
    const bnr_dev *shape = nullptr;                     // host struct of member 0 (sizes are equal for all members)
    hipStream_t stream = nullptr, stream2 = nullptr;   // stream2: the Gram branch of a sweep
    hipStream_t stream3 = nullptr, stream4 = nullptr;  // pipelined schedule: the factorization beside the Gram; the sacrificial first branch
    int pipeline = -1;                                  // -1: chosen by size / availability; 0: the factorization follows the Gram; 1: beside it
    int gate_us = 3000;                                 // how long a gate of the factorization polls for the Gram's progress
    unsigned *gctl = nullptr;                           // k_gram8p: queue heads and tickets
    const unsigned *resv = nullptr;                     // reserved compute units (device table shared per device), nullptr: none
    std::vector<hipEvent_t> fj;                         // fork/join events
    size_t fj_next = 0;
    int overlap = 1;
    int nop_fork = 0;                                   // (experiments) 1: an empty kernel is captured as the first forked branch of every sweep (needs stream4)
    int resv_mask = 0x80;                               // k_gram8q: cu ids (inside a shader engine) it keeps off -- 0x80 = cu id 7 of every SE = 32 CUs
    int crit_origin = 0;                                // 1: the critical chain (Gram, factorization, solve, back-projection) stays on the capture origin's queue and the scalar branch is the forked one; 2: the same with an empty kernel captured as the first fork
    int gram_variant = 0;                               // 0: chosen per launch; 8 / 16: k_gram8 / k_gram forced (tests, experiments)
    int fuse_reduce = -1;                               // -1 / 1: launch 0 of the one-panel factorization also sums the Gram's K-split partials (no k_gram_reduce launch); 0: separate pass
    int group_backproj = 0;                             // 1: the same for the back-projection / GIG kernel (opt-in: bitwise equal, measured no faster -- the block is bound by the latency of the draws' arithmetic)
    int group_xpass = -1;                               // -1 / 1: a group whose members share X runs the X pass with one workgroup per chunk for all chains; 0: per chain
    bnr_plan_entry *gplan_pin = nullptr, *gplan_dev = nullptr;   // groups: the members' plans of a run call, staged for one copy
    int gplan_cap = 0;                                  // entries per member in there
    int lin_debug = 0, lin_merge = 0;                   // lin_merge: one stream per part (its scalar branch in front of its Gram)
    int lin = 0;                                        // >= 1: the linear schedule with that many phase-shifted parts (see capture_linear)
    std::vector<hipStream_t> lstreams;                  // [2 p] critical chain, [2 p + 1] scalar branch of part p
    unsigned long long *lflags = nullptr;               // the counters the streams meet through
    struct lrung { int k, part, which; hipGraph_t graph; hipGraphExec_t gexec; };
    std::vector<lrung> lladder;
    int wide_backproj = -1;                              // 1: k_backproj64 (64 edges per workgroup, one edge per lane of the drawing wave); -1: launches of many rounds (a group at large q)
    int split_sums = -1;                                 // 1: the back-projection's partial sums as a launch of their own in front of the scalar tail (off the critical chain)
    int spw_cap = 1;                                    // super blocks per update workgroup of the factorization, at most (round 6: 1 -- with the pipelined panel sweep one block each is the shorter launch: 8 chains 369.4 against 372-374 us per sweep; rounds 3-5 packed up to 4 behind the single sweeping wave)
    // Round 6: WHEN the scalar branch's kernels start is part of the schedule (profiles/round6_experiments_notes.txt A): inside the two-branch sweep they are ordered behind
    // points of the critical chain by events (graph edges), instead of starting whenever the dispatcher lets the second queue in.
    int tail_after = -2;                                // k_tail(s-1) waits for: -1 nothing (rounds 1-5), 0 the Gram of sweep s; -2: default by size (tail_after_default)
    int node_after = -2;                                // k_node(s) waits for factorization launch number node_after (0-based; -1 nothing); -2: default by size (node_after_default)
    int factor_variant = -1;                            // -1: chosen by size; 0: right-looking k_chol_step (+ k_gram_reduce); 1: left-looking k_chol_ll
    int use_graph = 1, graph_k = 16;                     // (round 5: 16, was 8 -- between two graph launches the GPU idles ~30 us: 640 sweeps 382.2 -> 380.1 us each, 20 sweeps = 16 + 4 instead of 8 + 8 + 4)
    struct rung { int k; hipGraph_t graph; hipGraphExec_t gexec; };
    std::vector<rung> ladder;                           // captured graphs of graph_k, graph_k/2, ..., 1 sweeps: any batch is replayed
    bnr_dev *cds_pin = nullptr;                         // pinned staging of the members' descriptors
    long long *status_dev = nullptr, *status_pin = nullptr;   // nb x 16: the members' event counters, gathered once per run call
    int64_t n_replayed = 0, n_eager = 0;                // sweeps issued by graph replay / eagerly since the last run call began
    // profiling
    int profiling = 0;
    std::vector<hipEvent_t> ev;  // pairs around k_gram
    double t_gram_us = 0, t_iter_us = 0, t_gram_acc = 0;
    int64_t n_gram = 0, n_iter = 0;
};

// read-only device inputs of a fit (model matrix, response, edge maps, Gram task map): shared by the chains created
// with bnr_chain_create_like, freed with the last of them
struct bnr_inputs {
    std::vector<void *> bufs;
    int gq_off[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};       // k_gram8p: queue x = gmapc[gq_off[x] .. gq_off[x + 1])
    ~bnr_inputs() { for (void *p : bufs) (void)hipFree(p); }
};

struct bnr_chain {
    bnr_dev d{};
    std::shared_ptr<bnr_inputs> in;
    bnr_exec x;                  // issues this chain alone
    int device = 0;
    std::vector<void *> allocs;
    bnr_plan_entry *plan_dev = nullptr, *plan_pin = nullptr;
    int plan_cap = 0;
    int64_t iter = 0;            // global iteration id of the last drawn row
    int carried_row = -1;        // 0-based row whose (rr, sig_q) are in d.scal; -1 = invalid
    int next_row = 0;            // 1-based j the next run call would write
    bool pending = false;
    long long *counters_host = nullptr;
    int *pbase_dev = nullptr;
    size_t trace_bytes = 0;
    struct bnr_group *group = nullptr;   // lockstep group this chain belongs to (at most one)
    const unsigned char *x8_kept = nullptr;   // the byte image of X (also while option "byte_x" is 0); nullptr: the input had none
    const unsigned char *xm_kept = nullptr;   // the byte MASK of a 0/1 model matrix for the i8 Gram (also while option "gram_i8" is 0); nullptr: the input was not binary
    long long cap_seen = 0;      // sampler-cap events already reported (the device counter is cumulative: a capped draw is reported by the call it happened in, once)
};

struct bnr_group {
    std::vector<bnr_chain *> m;
    bnr_exec x;                  // issues all members together
};

static int round_up(int a, int b) { return (a + b - 1) / b * b; }

template <typename T>
static int dev_alloc(bnr_chain *c, T **p, size_t count, bool zero = true)
{
    void *q = nullptr;
    HIPCHK(hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
    // zeroed ON THE CHAIN'S OWN STREAM: the streams of this library are non-blocking, i.e. not ordered with the legacy stream a plain
    // hipMemset runs on -- a fill that is still pending there when the first kernels of the chain start would wipe what they wrote
    // (seen under rocprofv3 --pmc, which delays dispatches: first run of a fresh chain "Cholesky failed" in every node, round 4 notes)
    if (zero) HIPCHK(hipMemsetAsync(q, 0, std::max<size_t>(count, 1) * sizeof(T), c->x.stream));
    c->allocs.push_back(q);
    *p = (T *)q;
    return BNR_OK;
}
static void forget_alloc(bnr_chain *c, void *p)
{
    auto it = std::find(c->allocs.begin(), c->allocs.end(), p);
    if (it != c->allocs.end()) c->allocs.erase(it);
}

// Dynamic LDS beyond 64 KiB needs the function attribute, and the attribute is per DEVICE: set once per device of this
// process, after hipSetDevice, under a lock (handles may be created from several host threads); failures are reported at create.
struct bnr_exec;
static int late_kernels_lds_attributes(int bytes);
static void launch_late_xpass_group2(bnr_exec &x, int s);
static void launch_late_backproj64(bnr_exec &x, int s, int flags, size_t lds64);
static int ensure_lds_attributes(int device)
{
    static std::mutex mu;
    static std::vector<char> done;
    std::lock_guard<std::mutex> lock(mu);
    if ((int)done.size() <= device) done.resize(device + 1, 0);
    if (done[device]) return BNR_OK;
    const int big = 124 * 1024;
    const void *fns[] = {(const void *)&k_tail<bnr_one>, (const void *)&k_tail<bnr_many>, (const void *)&k_backproj<bnr_one>,
                         (const void *)&k_backproj<bnr_many>, (const void *)&k_solve_a4<bnr_one>, (const void *)&k_solve_a4<bnr_many>,
                         (const void *)&k_xpass_group, (const void *)&k_node<bnr_one>, (const void *)&k_node<bnr_many>};   // (k_node: 66.5 KB at R = 32)
    for (const void *f : fns) HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, big));
    if (int rc = late_kernels_lds_attributes(big)) return rc;   // (defined at the end of this file, like every other reference to the kernels added late in round 5: the order of
                                                                // first reference is the order of the instantiations in the code object, and a chain alone is 1 % slower when
                                                                // those kernels sit in the middle of it)
    done[device] = 1;
    return BNR_OK;
}

extern "C" {

static int exec_init(bnr_exec &x, int device, int nb, const bnr_dev *shape);
static int check_launch(const char *what);
static void exec_free(bnr_exec &x);
static int sync_dev(bnr_chain *c);

int bnr_abi_version(void) { return BNR_ABI_VERSION; }
const char *bnr_last_error(void) { return g_err.c_str(); }
int bnr_device_synchronize(int32_t device)
{
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipDeviceSynchronize());
    return BNR_OK;
}
int bnr_runtime_version(int *version)
{
    if (!version) return fail(BNR_ERR_BAD_ARG, "version is NULL");
    HIPCHK(hipRuntimeGetVersion(version));
    return BNR_OK;
}
int bnr_device_count(int *count)
{
    if (!count) return fail(BNR_ERR_BAD_ARG, "count is NULL");
    HIPCHK(hipGetDeviceCount(count));
    return BNR_OK;
}

void bnr_host_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    bnr_u4 r = bnr_philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
void bnr_host_uniform2(uint64_t seed, uint32_t it, uint32_t site, uint32_t elem, uint32_t att, double out[2])
{ bnr_draw2(seed, it, site, elem, att, out[0], out[1]); }
double bnr_host_normal(uint64_t seed, uint32_t it, uint32_t site, uint32_t elem, uint32_t att)
{ return bnr_normal(seed, it, site, elem, att); }
double bnr_host_gamma(uint64_t seed, double shape, uint32_t it, uint32_t site, uint32_t elem)
{ int cap = 0; return bnr_gamma(seed, shape, it, site, elem, &cap); }
double bnr_host_gig(uint64_t seed, double lambda, double chi, double psi, uint32_t it, uint32_t elem)
{ int cap = 0; return bnr_gig(seed, lambda, chi, psi, it, elem, &cap); }
int32_t bnr_host_edge_index(int32_t V, int32_t l, int32_t k)
{ return l >= k ? bnr_edge_index(V, l, k) : bnr_edge_index(V, k, l); }

static int alloc_trace(bnr_chain *c, int tot, double **out)
{
    void *p = nullptr;
    // + two hidden rows behind the table (0-based tot, tot + 1): scratch target of the purge ring with purge_burn == 1
    // (enqueue_run) and of the discarded sweeps that prime the captured graphs (prime_graphs)
    size_t bytes = (size_t)(tot + 2) * c->d.rowlen * sizeof(double);
    HIPCHK(hipMalloc(&p, bytes));
    HIPCHK(hipMemsetAsync(p, 0, bytes, c->x.stream));       // (on the chain's stream, see dev_alloc)
    *out = (double *)p;
    c->trace_bytes = bytes;
    return BNR_OK;
}

// Where the model matrix comes from: the n x q matrix X_new of generate_samples! (gibbs.jl:917-918) in one of the element types
// the reference accepts (Matrix{eltype(T)}: Bool, Int, Float64 ...), or the vector of n adjacency matrices itself, vectorised on
// the device (setup_X!, gibbs.jl:239-247: row i = lower_triangle(X[i]), utils.jl:40-57).
struct x_source {
    const void *X = nullptr;               // n x q column-major (mats == nullptr)
    const void *const *mats = nullptr;     // n pointers to V x V column-major matrices
    int dtype = BNR_F64;
};
static size_t dtype_size(int t) { return t == BNR_U8 ? 1 : (t == BNR_I32 || t == BNR_F32) ? 4 : 8; }
// raw (host layout, any element type) -> the padded f64 device matrix; the conversion runs on the device
static int upload_x(bnr_chain *c, const x_source &src, double *Xd, unsigned char *X8, int *not_bytes_dev)
{
    const bnr_dev &d = c->d;
    const size_t es = dtype_size(src.dtype);
    if (!src.mats && src.dtype == BNR_F64) {
        HIPCHK(hipMemcpy2D(Xd, (size_t)d.n_pad * sizeof(double), src.X, (size_t)d.n * sizeof(double), (size_t)d.n * sizeof(double), d.q, hipMemcpyHostToDevice));
        return BNR_OK;
    }
    const size_t count = src.mats ? (size_t)d.n * d.V * d.V : (size_t)d.n * d.q;
    void *raw = nullptr;
    HIPCHK(hipMalloc(&raw, count * es));
    hipError_t e = hipSuccess;
    if (src.mats) {
        for (int i = 0; i < d.n && e == hipSuccess; ++i) {
            if (!src.mats[i]) { (void)hipFree(raw); return fail(BNR_ERR_BAD_ARG, "NULL adjacency matrix"); }
            e = hipMemcpyAsync((char *)raw + (size_t)i * d.V * d.V * es, src.mats[i], (size_t)d.V * d.V * es, hipMemcpyHostToDevice, c->x.stream);
        }
    } else e = hipMemcpyAsync(raw, src.X, count * es, hipMemcpyHostToDevice, c->x.stream);
    if (e == hipSuccess) {
        const dim3 grid((d.n + 63) / 64, std::min(d.q, 65535)), block(64);
        switch (src.dtype) {
        case BNR_U8:  hipLaunchKernelGGL(k_x_convert<uint8_t>, grid, block, 0, c->x.stream, (const uint8_t *)raw, src.mats != nullptr, d.n, d.V, d.q, d.n_pad, d.ek, d.el, Xd, X8, not_bytes_dev); break;
        case BNR_I32: hipLaunchKernelGGL(k_x_convert<int32_t>, grid, block, 0, c->x.stream, (const int32_t *)raw, src.mats != nullptr, d.n, d.V, d.q, d.n_pad, d.ek, d.el, Xd, X8, not_bytes_dev); break;
        case BNR_I64: hipLaunchKernelGGL(k_x_convert<int64_t>, grid, block, 0, c->x.stream, (const int64_t *)raw, src.mats != nullptr, d.n, d.V, d.q, d.n_pad, d.ek, d.el, Xd, X8, not_bytes_dev); break;
        case BNR_F32: hipLaunchKernelGGL(k_x_convert<float>, grid, block, 0, c->x.stream, (const float *)raw, src.mats != nullptr, d.n, d.V, d.q, d.n_pad, d.ek, d.el, Xd, (unsigned char *)nullptr, not_bytes_dev); break;
        default:      hipLaunchKernelGGL(k_x_convert<double>, grid, block, 0, c->x.stream, (const double *)raw, src.mats != nullptr, d.n, d.V, d.q, d.n_pad, d.ek, d.el, Xd, (unsigned char *)nullptr, not_bytes_dev); break;
        }
        e = hipStreamSynchronize(c->x.stream);
    }
    (void)hipFree(raw);
    if (e != hipSuccess) return fail(BNR_ERR_HIP, std::string("upload of X: ") + hipGetErrorString(e));
    return check_launch("k_x_convert");
}

// donor != NULL: share the donor's device inputs instead of uploading X, y (bnr_chain_create_like)
// The Gram's K split (gibbs.jl:434's product, split-K over workgroups).  Chosen so that a ONE-chain Gram launch fills the chip in whole rounds of
// one workgroup per CU: a grid of 288 workgroups on 256 CUs runs two rounds and takes twice as long as one of 252 (measured: 58 vs 31 us).  Two
// 512-thread workgroups fit a CU, but splitting K further to use both slots of a single chain's launch only doubles the split-K partials
// (measured: ksplit 14 vs 7 at n=500, V=100: one chain 211 vs 208 us per sweep, a group of 8 chains 468 vs 446 us); a lockstep group fills the
// second slot with the next chain's workgroups.
// ... and raised, whatever the above chose, until a K-group's slice (with its prefetch distance) fits the 2 GiB window of the buffer resource the
// loops address X through (32-bit scalar byte offsets; ADVICE r5: n = 14 000 with V >= 280, n = 8 000 with V >= 520 overflowed silently).
// Returns non-zero when no split of at most 4096 slices fits.
static int gram_plan(int n_pad, int q, int ncu, int kg, int forced_split, int *kg_out, int *ksplit_out, int *kchunk_out)
{
    const int ntile = n_pad / BNR_GT, ntl = ntile * (ntile + 1) / 2;
    auto best_split = [&](int slots) {
        double best = -1.0;
        int bk = 1;
        for (int ks = 1; ks <= 32; ++ks) {
            if (ks > 1 && (q + ks - 1) / ks < 32 * kg) break;  // keep every K-group at least 32 columns long
            long tasks = (long)ntl * ks;
            double eff = (double)tasks / (double)(((tasks + slots - 1) / slots) * slots);
            double score = eff - 0.005 * ks;                              // fewer split-K partials when efficiency ties
            if (score > best) { best = score; bk = ks; }
        }
        return bk;
    };
    // long K (>= 1024 columns per slice even when both slots of every CU are used): the partials are cheap next to the
    // loop, fill both slots (n=500, V=300: 240 vs 262 us per Gram); otherwise one workgroup per CU
    int ksplit = best_split(2 * ncu * (kg == 2 ? 1 : 0) + ncu * (kg == 2 ? 0 : 1));
    if (q / ksplit < 1024) ksplit = best_split(ncu);
    if (forced_split > 0) ksplit = forced_split;
    auto chunk_of = [&](int ks) { return round_up((q + ks - 1) / ks, 8 * kg); };
    const long long window = 0x7FFFFFFFLL;
    while (bnr_gram_span_bytes(n_pad, chunk_of(ksplit), kg) > window && ksplit < 4096) ++ksplit;
    if (bnr_gram_span_bytes(n_pad, chunk_of(ksplit), kg) > window) return 1;
    *kg_out = kg; *ksplit_out = ksplit; *kchunk_out = chunk_of(ksplit);
    return 0;
}
int bnr_host_gram_plan(int32_t n, int32_t V, int32_t ncu, int32_t out[4])
{
    if (!out || n < 1 || V < 2 || ncu < 1) return fail(BNR_ERR_BAD_ARG, "bnr_host_gram_plan: bad argument");
    const int n_pad = round_up(n, BNR_GT);
    const long long q = (long long)V * (V + 1) / 2;
    if (q > 0x7FFFFFFF / 2) return fail(BNR_ERR_BAD_ARG, "bnr_host_gram_plan: V too large");
    int kg = 2, ksplit = 0, kchunk = 0;
    if (gram_plan(n_pad, (int)q, ncu, 2, 0, &kg, &ksplit, &kchunk)) return fail(BNR_ERR_BAD_ARG, "no K split fits the 2 GiB window");
    const long long span = bnr_gram_span_bytes(n_pad, kchunk, kg);
    out[0] = ksplit; out[1] = kchunk; out[2] = kchunk * ksplit; out[3] = (int32_t)(span >> 20);
    return BNR_OK;
}
static int chain_build(const bnr_chain *donor, int32_t n, int32_t V, int32_t R, const x_source &xs, const double *y, const bnr_hyper *hyper,
                       uint64_t seed, int32_t chain_id, int32_t device, int32_t tot_save, bnr_chain **out)
{
    if (!out || !hyper || (!donor && ((!xs.X && !xs.mats) || !y))) return fail(BNR_ERR_BAD_ARG, "NULL argument");
    if (xs.dtype < BNR_F64 || xs.dtype > BNR_F32) return fail(BNR_ERR_BAD_ARG, "unknown element type of X");
    if (n < 1 || V < 2 || R < 1 || R > BNR_RMAX || tot_save < 2)
        return fail(BNR_ERR_BAD_ARG, "need n>=1, V>=2, 1<=R<=32, tot_save>=2");
    // LDS budgets of the kernels that stage a whole vector / the R x V matrix u: k_tail keeps u (R V doubles) next to 33 KiB of
    // static arrays, k_backproj / k_solve_a4 keep an n-vector; 160 KiB per workgroup on gfx950
    if (n > 14000) return fail(BNR_ERR_BAD_ARG, "n must not exceed 14000 (n-vectors are staged in LDS)");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(BNR_ERR_BAD_ARG, "no such device");
    HIPCHK(hipSetDevice(device));
    {
        int rc0 = ensure_lds_attributes(device);
        if (rc0) return rc0;
    }
    bnr_chain *c = new bnr_chain();
    c->device = device;
    bnr_dev &d = c->d;
    d.n = n; d.V = V; d.R = R; d.q = V * (V + 1) / 2; d.tot = tot_save;
    d.n_pad = round_up(n, BNR_GT);
    d.ntile = d.n_pad / BNR_GT;
    const int ntl = d.ntile * (d.ntile + 1) / 2;
    {
        hipDeviceProp_t prop;
        int ncu = 256;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount;
        const char *ev = getenv("BNR_GRAM_KG");
        const char *ek = getenv("BNR_GRAM_KSPLIT");                           // experiments only
        int kchunk0 = 0;
        if (gram_plan(d.n_pad, d.q, ncu, (ev && atoi(ev) == 4) ? 4 : 2, (ek && atoi(ek) > 0) ? atoi(ek) : 0, &d.gram_kg, &d.ksplit, &kchunk0))
            { delete c; return fail(BNR_ERR_BAD_ARG, "the model matrix is too large for the Gram's K split (n_pad x q / 32 columns must stay below 2 GiB per K-group)"); }
    }
    int kchunk = round_up((d.q + d.ksplit - 1) / d.ksplit, 8 * d.gram_kg);
    d.q_pad = kchunk * d.ksplit;
    // the i8 Gram of a binary model matrix (k_gram_i8): K slices padded to whole 64-column MFMA steps, digit planes of S for 1e-12 of max |G|
    d.kcp = round_up(kchunk, 64);
    d.kslab = d.kcp * d.ksplit;
    d.i8L = (8.0 * d.q * std::ldexp(1.0, -56) <= 1e-12) ? 7 : 8;            // digit planes: 8 q 2^(-8 L) <= 1e-12 (7 up to q = 9007)
    // row layout
    int o = 4;
    d.o_xi = o; o += V;
    d.o_lam = o; o += R;
    d.o_pi = o; o += 3 * R;
    d.o_M = o; o += R * R;
    d.o_u = o; o += R * V;
    o = round_up(o, 16);
    d.o_gamma = o; o += d.q;
    o = round_up(o, 16);
    d.o_S = o; o += d.q;
    o += (d.q_pad - d.q) + BNR_GRAM_PREFETCH_COLS;      // zeros behind S that nothing ever writes: the Gram reads S by column index up to BNR_GRAM_LOOKAHEAD batches past q_pad without a clamp (X is zero there)
    d.rowlen = round_up(o, 16);
    d.eta = hyper->eta; d.zeta = hyper->zeta; d.iota = hyper->iota;
    d.aDelta = hyper->aDelta; d.bDelta = hyper->bDelta; d.nu = hyper->nu;
    d.seed = seed + (uint64_t)(int64_t)chain_id;
    // GEMV pass partition
    d.nblk_x = std::max(std::min((d.q + 31) / 32, 256), (d.q + 255) / 256);
    d.chunk_x = (d.q + d.nblk_x - 1) / d.nblk_x;
    d.nblk_x = (d.q + d.chunk_x - 1) / d.chunk_x;
    d.chunk_bp = 32;
    d.nblk_bp = (d.q + d.chunk_bp - 1) / d.chunk_bp;

    int rc;
#define TRY(x) do { rc = (x); if (rc) { bnr_chain_destroy(c); return rc; } } while (0)
    TRY(exec_init(c->x, device, 1, &c->d));
    if (donor) {
        c->in = donor->in;
        c->x8_kept = donor->x8_kept;                     // the image belongs to the shared inputs: a chain made from a donor with byte_x = 0 can switch it on again
        c->xm_kept = donor->xm_kept;
        d.X = donor->d.X; d.X8 = donor->d.X8; d.XM = donor->d.XM; d.y = donor->d.y; d.ek = donor->d.ek; d.el = donor->d.el; d.gmap = donor->d.gmap; d.gmapc = donor->d.gmapc;
    } else {
        c->in = std::make_shared<bnr_inputs>();
        double *Xd = nullptr, *yd = nullptr;
        int *ek = nullptr, *el = nullptr, *gm = nullptr, *gmc = nullptr;
        int gq_off[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        auto in_alloc = [&](void **ptr, size_t bytes) -> int {
            HIPCHK(hipMalloc(ptr, std::max<size_t>(bytes, 8)));
            c->in->bufs.push_back(*ptr);
            HIPCHK(hipMemsetAsync(*ptr, 0, std::max<size_t>(bytes, 8), c->x.stream));
            HIPCHK(hipStreamSynchronize(c->x.stream));       // the (synchronous, legacy-stream) uploads below must find the zeros in place
            return BNR_OK;
        };
        TRY(in_alloc((void **)&Xd, sizeof(double) * (size_t)d.n_pad * (d.q_pad + BNR_GRAM_PREFETCH_COLS)));   // zero columns behind X: the Gram's prefetch distance (bnr_kernels.h)
        TRY(in_alloc((void **)&yd, sizeof(double) * d.n_pad));
        TRY(in_alloc((void **)&ek, sizeof(int) * d.q));
        TRY(in_alloc((void **)&el, sizeof(int) * d.q));
        if (hipMemcpy(yd, y, (size_t)n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { bnr_chain_destroy(c); return fail(BNR_ERR_HIP, "copy y failed"); }
        {
            std::vector<int> hk(d.q), hl(d.q);
            int e = 0;
            for (int k = 0; k < V; ++k) for (int l = k; l < V; ++l, ++e) { hk[e] = k; hl[e] = l; }
            HIPNOTE(hipMemcpy(ek, hk.data(), d.q * sizeof(int), hipMemcpyHostToDevice));
            HIPNOTE(hipMemcpy(el, hl.data(), d.q * sizeof(int), hipMemcpyHostToDevice));
        }
        d.ek = ek; d.el = el;
        // 0..255-valued integer input (Bool / UInt8 always, Int32 / Int64 if every value fits): a byte image of X for the X passes
        unsigned char *X8 = nullptr;
        int *nb = nullptr;
        if (xs.dtype == BNR_U8 || xs.dtype == BNR_I32 || xs.dtype == BNR_I64) {
            TRY(in_alloc((void **)&X8, (size_t)d.n_pad * (d.q_pad + BNR_GRAM_PREFETCH_COLS)));
            TRY(in_alloc((void **)&nb, sizeof(int)));
        }
        TRY(upload_x(c, xs, Xd, X8, nb));
        if (X8) {
            int not_bytes = 0;
            if (hipMemcpy(&not_bytes, nb, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { bnr_chain_destroy(c); return fail(BNR_ERR_HIP, "copy failed"); }
            const char *off = getenv("BNR_NO_BYTE_X");
            if (not_bytes || (off && atoi(off))) {                     // no byte image after all: give its memory back (an eighth of X: 22 MB at config 5)
                auto it = std::find(c->in->bufs.begin(), c->in->bufs.end(), (void *)X8);
                if (it != c->in->bufs.end()) c->in->bufs.erase(it);
                (void)hipFree(X8);
                X8 = nullptr;
            }
        }
        d.X8 = X8;
        c->x8_kept = X8;
        // every entry 0 or 1 (the reference's adjacency data): the row-major byte mask the i8 Gram reads, built from the byte image on the device
        d.XM = nullptr;
        {
            const char *off8 = getenv("BNR_NO_GRAM_I8");
            // k_gram_i8 keeps the digits of a whole K slice in LDS beside its staging buffers: beyond 64 KiB of dynamic LDS (q in the hundreds of
            // thousands with the K split capped at 32) there is no i8 path -- the f64 Gram runs as for any other matrix
            const bool fits = bnr_i8_lds_bytes(d.i8L, d.kcp) <= (size_t)64 * 1024;
            if (X8 && fits && !(off8 && atoi(off8))) {
                unsigned char *XM = nullptr;
                int *nbin = nullptr;
                TRY(in_alloc((void **)&XM, (size_t)d.n_pad * d.kslab));
                TRY(in_alloc((void **)&nbin, sizeof(int)));
                const size_t groups = (size_t)d.n_pad * (d.kslab / 16);
                hipLaunchKernelGGL(k_x_mask, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, c->x.stream, (const unsigned char *)X8, d.n, d.n_pad, d.q, kchunk, d.kcp, d.kslab, XM, nbin);
                int not_binary = 1;
                if (hipStreamSynchronize(c->x.stream) != hipSuccess || hipMemcpy(&not_binary, nbin, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { bnr_chain_destroy(c); return fail(BNR_ERR_HIP, "building the byte mask of X failed"); }
                if (not_binary) {
                    auto it = std::find(c->in->bufs.begin(), c->in->bufs.end(), (void *)XM);
                    if (it != c->in->bufs.end()) c->in->bufs.erase(it);
                    (void)hipFree(XM);
                    XM = nullptr;
                }
                d.XM = XM;
            }
        }
        c->xm_kept = d.XM;
        // default: on where it was measured faster (profiles/round5_gram_i8.txt: n = 500, V = 100 and larger -- 22 vs 35 us per launch for one chain, 70 vs 197 for
        // eight; at n = 200, V = 50 the second launch costs more than the f64 Gram's 12 us).  Chain option "gram_i8" switches it either way.
        if ((double)d.n_pad * d.n_pad * d.q < 2.5e8) d.XM = nullptr;
        {
            // XCD-aware task map of k_gram (tasks = lower tiles x K slices): workgroup i runs on XCD i % 8; give it a K
            // slice ks with ks % 8 == i % 8 while there are any, so that a slice of X is read through one XCD's L2
            const int ntask = ntl * d.ksplit;
            std::vector<int> map(ntask, -1);
            std::vector<std::vector<int>> queue(8);
            for (int t = 0; t < ntl; ++t)
                for (int ks = 0; ks < d.ksplit; ++ks) queue[ks % 8].push_back(t | (ks << 16));
            std::vector<size_t> head(8, 0);
            std::vector<int> later;
            for (int i = 0; i < ntask; ++i) {
                int x = i % 8;
                if (head[x] < queue[x].size()) map[i] = queue[x][head[x]++];
                else later.push_back(i);
            }
            int xq = 0;
            for (int i : later) {
                while (xq < 8 && head[xq] >= queue[xq].size()) ++xq;
                map[i] = queue[xq][head[xq]++];
            }
            TRY(in_alloc((void **)&gm, sizeof(int) * ntask));
            HIPNOTE(hipMemcpy(gm, map.data(), ntask * sizeof(int), hipMemcpyHostToDevice));
            // k_gram8p: eight queues, one per XCD, each in tile-COLUMN order (the factorization consumes G column by column): the
            // XCD-aware assignment of the static map above -- XCD x works through the K slices ks = x mod 8 and the XCDs without a
            // slice of their own take an equal share from the END of the others' lists (cascading instead -- an XCD that has run dry
            // helping its neighbour, which then runs dry early and moves on -- was measured: 3.5 x the L2 misses, 225 instead of 49 MB fetched)
            std::vector<int> mapc;
            {
                std::vector<std::vector<int>> qc(8), mine(8);
                for (int tc = 0; tc < d.ntile; ++tc)
                    for (int ti = tc; ti < d.ntile; ++ti)
                        for (int ks = 0; ks < d.ksplit; ++ks) qc[ks % 8].push_back((ti * (ti + 1) / 2 + tc) | (ks << 16));
                const int share = (ntask + 7) / 8;
                std::vector<int> spill;                                   // what exceeds an XCD's share, taken from the end of its list
                for (int x = 0; x < 8; ++x) {
                    while ((int)qc[x].size() > share) { spill.push_back(qc[x].back()); qc[x].pop_back(); }
                    mine[x] = qc[x];
                }
                std::sort(spill.begin(), spill.end(), [](int a, int b) { return (a & 0xFFFF) != (b & 0xFFFF) ? (a & 0xFFFF) < (b & 0xFFFF) : a < b; });
                for (int x = 0; x < 8 && !spill.empty(); ++x)
                    while ((int)mine[x].size() < share && !spill.empty()) { mine[x].push_back(spill.front()); spill.erase(spill.begin()); }
                for (int x = 0; x < 8; ++x) { gq_off[x] = (int)mapc.size(); mapc.insert(mapc.end(), mine[x].begin(), mine[x].end()); }
                gq_off[8] = (int)mapc.size();
            }
            TRY(in_alloc((void **)&gmc, sizeof(int) * ntask));
            HIPNOTE(hipMemcpy(gmc, mapc.data(), ntask * sizeof(int), hipMemcpyHostToDevice));
        }
        d.X = Xd; d.y = yd; d.ek = ek; d.el = el; d.gmap = gm; d.gmapc = gmc;
        for (int x = 0; x < 9; ++x) c->in->gq_off[x] = gq_off[x];
    }
    TRY(alloc_trace(c, tot_save, &d.trace));
    TRY(dev_alloc(c, &d.Wbuf, d.q_pad));
    TRY(dev_alloc(c, &d.sz, d.q_pad));
    TRY(dev_alloc(c, &d.PW, (size_t)d.nblk_x * d.n_pad));
    TRY(dev_alloc(c, &d.PA, (size_t)d.nblk_x * d.n_pad));
    TRY(dev_alloc(c, &d.PG, (size_t)d.nblk_x * d.n_pad));
    TRY(dev_alloc(c, &d.Gpart, (size_t)d.ksplit * ntl * BNR_GT * BNR_GT));
    TRY(dev_alloc(c, &d.E, (size_t)bnr_ldE(d.n_pad) * d.n_pad));
    TRY(dev_alloc(c, &d.a3, d.n_pad));
    TRY(dev_alloc(c, &d.xw, d.n_pad));
    TRY(dev_alloc(c, &d.a4, d.n_pad));
    TRY(dev_alloc(c, &d.res, d.n_pad));
    TRY(dev_alloc(c, &d.xg, d.n_pad));
    TRY(dev_alloc(c, &d.bw, d.n_pad));
    TRY(dev_alloc(c, &d.wv, d.n_pad));
    TRY(dev_alloc(c, &d.scal, 16));
    if (c->xm_kept) TRY(dev_alloc(c, &d.Sdig, (size_t)d.i8L * d.kslab));
    TRY(dev_alloc(c, &d.Minv, BNR_RMAX * BNR_RMAX + 1));
    TRY(dev_alloc(c, &d.Psum, (size_t)d.nblk_bp * (1 + 3 * R)));
    TRY(dev_alloc(c, &d.counters, 16));
    TRY(dev_alloc(c, &d.stamp, 8 * ntl));
    TRY(dev_alloc(c, &d.gprog, d.ntile + 4));
    TRY(dev_alloc(c, &d.dbg, 4096));
    c->plan_cap = 1 << 16;
    TRY(dev_alloc(c, &c->plan_dev, c->plan_cap));
    TRY(dev_alloc(c, &c->pbase_dev, 4));
    d.pbase = c->pbase_dev;
    if (hipHostMalloc((void **)&c->plan_pin, sizeof(bnr_plan_entry) * c->plan_cap) != hipSuccess) { bnr_chain_destroy(c); return fail(BNR_ERR_HIP, "hipHostMalloc failed"); }
    if (hipHostMalloc((void **)&c->counters_host, sizeof(long long) * 16) != hipSuccess) { bnr_chain_destroy(c); return fail(BNR_ERR_HIP, "hipHostMalloc failed"); }
    d.plan = c->plan_dev;
    // every upload and fill above is either host-synchronous or on the chain's own stream: wait for THAT stream only (a device-wide synchronise would
    // stall on another chain's or group's pending asynchronous run and could surface its errors here)
    if (hipStreamSynchronize(c->x.stream) != hipSuccess) { bnr_chain_destroy(c); return fail(BNR_ERR_HIP, "hipStreamSynchronize failed"); }
    TRY(sync_dev(c));
    TRY(check_launch("chain_build"));                    // the index-map / task-map copies above only NOTE a failure: report it here, not in somebody's later call
#undef TRY
    *out = c;
    return BNR_OK;
}

int bnr_chain_create(int32_t n, int32_t V, int32_t R, const double *X, const double *y, const bnr_hyper *hyper,
                     uint64_t seed, int32_t chain_id, int32_t device, int32_t tot_save, bnr_chain **out)
{ x_source xs; xs.X = X; return chain_build(nullptr, n, V, R, xs, y, hyper, seed, chain_id, device, tot_save, out); }

int bnr_chain_create_typed(int32_t n, int32_t V, int32_t R, const void *X, int32_t x_dtype, const double *y, const bnr_hyper *hyper,
                           uint64_t seed, int32_t chain_id, int32_t device, int32_t tot_save, bnr_chain **out)
{ x_source xs; xs.X = X; xs.dtype = x_dtype; return chain_build(nullptr, n, V, R, xs, y, hyper, seed, chain_id, device, tot_save, out); }

int bnr_chain_create_from_matrices(int32_t n, int32_t V, int32_t R, const void *const *A, int32_t x_dtype, const double *y, const bnr_hyper *hyper,
                                   uint64_t seed, int32_t chain_id, int32_t device, int32_t tot_save, bnr_chain **out)
{ x_source xs; xs.mats = A; xs.dtype = x_dtype; return chain_build(nullptr, n, V, R, xs, y, hyper, seed, chain_id, device, tot_save, out); }

int bnr_chain_create_like(const bnr_chain *donor, uint64_t seed, int32_t chain_id, int32_t tot_save, bnr_chain **out)
{
    if (!donor) return fail(BNR_ERR_BAD_ARG, "NULL donor chain");
    const bnr_dev &a = donor->d;
    bnr_hyper h{a.eta, a.zeta, a.iota, a.aDelta, a.bDelta, a.nu};
    return chain_build(donor, a.n, a.V, a.R, x_source(), nullptr, &h, seed, chain_id, donor->device, tot_save, out);
}

static void drop_graph(bnr_exec &x)
{
    for (auto &r : x.ladder) { if (r.gexec) (void)hipGraphExecDestroy(r.gexec); if (r.graph) (void)hipGraphDestroy(r.graph); }
    x.ladder.clear();
    for (auto &r : x.lladder) { if (r.gexec) (void)hipGraphExecDestroy(r.gexec); if (r.graph) (void)hipGraphDestroy(r.graph); }
    x.lladder.clear();
}
static int exec_init(bnr_exec &x, int device, int nb, const bnr_dev *shape)
{
    x.device = device; x.nb = nb; x.shape = shape;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) x.ncu = prop.multiProcessorCount;
    }
    HIPCHK(hipStreamCreateWithFlags(&x.stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&x.stream2, hipStreamNonBlocking));
    HIPCHK(hipMalloc((void **)&x.cds, sizeof(bnr_dev) * nb));
    HIPCHK(hipHostMalloc((void **)&x.cds_pin, sizeof(bnr_dev) * nb));
    HIPCHK(hipMalloc((void **)&x.status_dev, sizeof(long long) * 16 * nb));
    HIPCHK(hipHostMalloc((void **)&x.status_pin, sizeof(long long) * 16 * nb));
    return BNR_OK;
}
static void exec_free(bnr_exec &x)
{
    if (x.stream) { (void)hipStreamSynchronize(x.stream); }
    if (x.stream2) { (void)hipStreamSynchronize(x.stream2); }
    if (x.stream3) { (void)hipStreamSynchronize(x.stream3); (void)hipStreamDestroy(x.stream3); }
    if (x.stream4) { (void)hipStreamSynchronize(x.stream4); (void)hipStreamDestroy(x.stream4); }
    for (hipStream_t st : x.lstreams) { (void)hipStreamSynchronize(st); }
    if (x.gctl) (void)hipFree(x.gctl);
    drop_graph(x);
    for (hipStream_t st : x.lstreams) (void)hipStreamDestroy(st);
    x.lstreams.clear();
    if (x.lflags) (void)hipFree(x.lflags);
    if (x.stream) (void)hipStreamDestroy(x.stream);
    if (x.stream2) (void)hipStreamDestroy(x.stream2);
    for (hipEvent_t e : x.fj) (void)hipEventDestroy(e);
    for (hipEvent_t e : x.ev) (void)hipEventDestroy(e);
    if (x.gplan_pin) (void)hipHostFree(x.gplan_pin);
    if (x.gplan_dev) (void)hipFree(x.gplan_dev);
    if (x.cds) (void)hipFree(x.cds);
    if (x.cds_pin) (void)hipHostFree(x.cds_pin);
    if (x.status_dev) (void)hipFree(x.status_dev);
    if (x.status_pin) (void)hipHostFree(x.status_pin);
    x = bnr_exec();
}
// the kernels read the chain's bnr_dev from device memory: refresh the copy whenever the host struct changes
static int sync_dev(bnr_chain *c)
{
    HIPCHK(hipStreamSynchronize(c->x.stream));
    c->x.cds_pin[0] = c->d;
    HIPCHK(hipMemcpyAsync(c->x.cds, c->x.cds_pin, sizeof(bnr_dev), hipMemcpyHostToDevice, c->x.stream));   // no legacy-stream calls at run time
    HIPCHK(hipStreamSynchronize(c->x.stream));
    return BNR_OK;
}

int bnr_group_destroy(bnr_group *g);

int bnr_chain_destroy(bnr_chain *c)
{
    if (!c) return BNR_OK;
    (void)hipSetDevice(c->device);                       // (best effort: a destructor has nobody to report to, and must not leave an error noted for the thread's next call)
    if (c->group) {                                      // a group cannot run without a member: dissolve it (the handle stays valid)
        bnr_group *g = c->group;
        (void)hipStreamSynchronize(g->x.stream);
        for (bnr_chain *m : g->m) m->group = nullptr;
        g->m.clear();
    }
    exec_free(c->x);
    for (void *p : c->allocs) (void)hipFree(p);
    if (c->d.trace) (void)hipFree(c->d.trace);
    if (c->plan_pin) (void)hipHostFree(c->plan_pin);
    if (c->counters_host) (void)hipHostFree(c->counters_host);
    delete c;
    return BNR_OK;
}

// ------------------------------------------------------------------------------------------ launch helpers
static int ensure_plan(bnr_chain *c, int count)
{
    if (count <= c->plan_cap) return BNR_OK;
    int cap = std::max(count, 2 * c->plan_cap);
    HIPCHK(hipStreamSynchronize(c->x.stream));
    bnr_plan_entry *nd = nullptr, *np = nullptr;
    HIPCHK(hipMalloc((void **)&nd, sizeof(bnr_plan_entry) * cap));
    HIPCHK(hipHostMalloc((void **)&np, sizeof(bnr_plan_entry) * cap));
    forget_alloc(c, c->plan_dev);
    (void)hipFree(c->plan_dev);
    (void)hipHostFree(c->plan_pin);
    c->allocs.push_back(nd);
    c->plan_dev = nd; c->plan_pin = np; c->plan_cap = cap;
    c->d.plan = nd;
    drop_graph(c->x);                       // one chain: the struct is a by-value kernel argument baked into the captured graph
    if (c->group && c->group->x.nb == 1) drop_graph(c->group->x);  // ... also into the graphs of a ONE-member group (bnr_one there too); larger groups read the device array
    return sync_dev(c);
}
// st: the stream the sweeps that read this plan are issued on (the chain's own, or its group's: no cross-stream hand-over)
static int upload_plan(bnr_chain *c, int count, hipStream_t st = nullptr)
{
    if (!st) st = c->x.stream;
    HIPCHK(hipMemcpyAsync(c->plan_dev, c->plan_pin, sizeof(bnr_plan_entry) * count, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(c->pbase_dev, 0, sizeof(int), st));
    return BNR_OK;
}
static int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (g_noted != hipSuccess) {
        const hipError_t ne = g_noted;
        g_noted = hipSuccess;
        return fail(BNR_ERR_HIP, std::string(what) + ": " + (g_noted_what ? g_noted_what : "HIP call") + ": " + hipGetErrorString(ne));
    }
    if (e != hipSuccess) return fail(BNR_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    return BNR_OK;
}

// one chain: struct by value; group: device array (see bnr_one / bnr_many)
#define BNR_LAUNCH(kern, grid, block, lds, st, x, ...)                                                                     \
    do {                                                                                                                   \
        if ((x).nb == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(kern<bnr_one>), grid, block, lds, st, bnr_one{*(x).shape}, ##__VA_ARGS__);   \
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(kern<bnr_many>), grid, block, lds, st, bnr_many{(x).cds}, ##__VA_ARGS__);  \
    } while (0)

static void launch_node(bnr_exec &x, int s, int mode)
{ BNR_LAUNCH(k_node, dim3(x.shape->V, 1, x.nb), dim3(64), (size_t)(64 * (2 * x.shape->R + 1) + 4 * x.shape->R * x.shape->R + 2 * x.shape->R) * sizeof(double), x.stream, x, s, mode); }
// the members of a lockstep group read the same device copy of X (chains of one fit made with bnr_chain_create_like)
static bool group_shares_x(const bnr_exec &x)
{
    if (x.nb < 2 || !x.cds_pin) return false;
    for (int i = 1; i < x.nb; ++i)
        if (x.cds_pin[i].X != x.cds_pin[0].X || x.cds_pin[i].X8 != x.cds_pin[0].X8) return false;
    return true;
}
static size_t tail_a_bytes(const bnr_exec &x) { return bnr_tail_a_lds_doubles(x.shape->R) * sizeof(double); }
static void launch_xpass(bnr_exec &x, int s, int which)
{
    // by default only where X is large (>= 8 MB per chain): the point is the L2 traffic beside the panel steps; small problems are chains of
    // latencies, and there the per-chain kernel's many small workgroups finish sooner (n = 200, V = 50: 125.8 vs 139.4 us per sweep of 8 chains)
    const bool big_x = (size_t)x.shape->n_pad * x.shape->q * sizeof(double) >= ((size_t)8 << 20);
    if (which == 3 && (x.group_xpass == 1 || (x.group_xpass < 0 && big_x)) && group_shares_x(x) && 16 * (size_t)x.shape->chunk_x * sizeof(double) <= 48 * 1024) {
        // one workgroup per column chunk and row slice for all members: X comes out of the L2s once, not once per chain
        // long column chunks (large q: 177 columns per workgroup at config 5): the straight-line column loop -- config 5 x 8 chains, where the scalar branch is the longer
        // chain behind the Gram; at the headline shape (32 columns per workgroup) it brought nothing per sweep (notes S), so short chunks keep the first kernel
        if (x.shape->chunk_x > 64) launch_late_xpass_group2(x, s);
        else hipLaunchKernelGGL(k_xpass_group, dim3(x.shape->nblk_x * ((x.shape->n_pad + 255) / 256)), dim3(256), 16 * x.shape->chunk_x * sizeof(double), x.stream, bnr_many{x.cds}, s, x.nb);
        return;
    }
    BNR_LAUNCH(k_xpass, dim3(round_up(x.shape->nblk_x, 8) * x.nb), dim3(256), 3 * x.shape->chunk_x * sizeof(double), x.stream, x, s, which, x.nb);
}
// Which factorization: right-looking (k_chol_step behind k_gram_reduce: the trailing update spread over the whole chip) unless the
// caller asks for the left-looking one (k_chol_ll: no reduction pass, ceil(nbk/4) + nbk - 1 workgroups per chain and launch, can
// run beside the Gram).  Same tables bit for bit.
static bool left_looking(const bnr_exec &x)
{
    (void)x; return false;
}
// The sweep's schedule: pipelined (option "pipeline" = 1) = the factorization runs BESIDE the Gram (k_gram8p keeps off the reserved
// CUs, k_chol_ll's gates follow its progress column by column); otherwise it follows the Gram on the same stream.
static bool pipelined(const bnr_exec &x)
{
    if (!left_looking(x) || !x.overlap || x.shape->gram_kg != 2) return false;
    return x.pipeline == 1;                                        // opt-in, experiments build only (notes round 3 B, round 4 B)
}
// two panels per launch with the K = 128 trailing update (variant 3) where the trailing update is bandwidth-bound: n_pad >= 1024
// (n = 2000 one chain 424 -> 448 it/s, n = 1000 eight chains 5.49 -> 5.88 k it/s; at n = 500 it is a draw and the one-panel launches stay)
static bool two_panel_default(const bnr_exec &x) { return x.shape->n_pad >= 1024; }
// the data-flow factorization (k_chol_df: one launch, one chain per XCD, blocks resident in registers): n_pad <= 512, groups of up to 8
// (opt-in, experiments build only: bitwise the same factor, measured slower -- profiles/round4_experiments_notes.txt, F)
static bool dataflow(const bnr_exec &x)
{
    (void)x; return false;
}
// small problems (n_pad <= 128): the whole factorization in ONE launch of one workgroup per chain (k_chol_small): factor_variant 5
// (experiments build only: bitwise equal, measured slower -- profiles/round4_experiments_notes.txt H)
static bool small_factor(const bnr_exec &x)
{
    (void)x; return false;
}
// the one-panel right-looking factorization (k_chol_step) is the one that runs, and its first launch takes over k_gram_reduce's work
static bool reduce_in_chol(const bnr_exec &x)
{
    if (small_factor(x)) return true;                    // (k_chol_small sums the partial tiles at the first touch of every block: no reduction pass)
    if (dataflow(x)) return true;                        // (k_chol_df sums the partial tiles at the first touch of every block: no reduction pass at all)
    const bool one_panel = x.factor_variant == 0 || (x.factor_variant < 0 && !two_panel_default(x));
    return one_panel && x.fuse_reduce != 0;
}
// the i8 Gram runs when every chain of the launch has the byte mask of a binary model matrix switched on (chain option "gram_i8")
static bool gram_on_i8(const bnr_exec &x)
{
    if (x.nb == 1) return x.shape->XM != nullptr;
    if (!x.cds_pin) return false;
    for (int i = 0; i < x.nb; ++i)
        if (!x.cds_pin[i].XM) return false;
    return true;
}
// workgroups per chain of k_sdigits: one per 2048 entries of the digit planes, at most 32 (each finds the largest S itself, then converts its slice)
static unsigned sdig_slices(const bnr_dev &d) { return (unsigned)std::max(1, std::min(32, (d.kslab + 2047) / 2048)); }
#define BNR_LAUNCH_I8(LL)                                                                                                                   \
    do {                                                                                                                                    \
        const size_t lds = bnr_i8_lds_bytes(LL, d.kcp);                                                                                     \
        if (x.nb == 1) {                                                                                                                    \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sdigits<bnr_one, LL>), dim3(1, sdig_slices(d)), dim3(1024), 0, st, bnr_one{d}, s);                          \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gram_i8<bnr_one, LL>), ggrid, dim3(256), lds, st, bnr_one{d}, s, 1);                        \
        } else {                                                                                                                            \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sdigits<bnr_many, LL>), dim3(x.nb, sdig_slices(d)), dim3(1024), 0, st, bnr_many{x.cds}, s);                 \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gram_i8<bnr_many, LL>), ggrid, dim3(256), lds, st, bnr_many{x.cds}, s, x.nb);               \
        }                                                                                                                                   \
    } while (0)
static void launch_gram_i8(bnr_exec &x, int s, hipStream_t st, dim3 ggrid)
{
    const bnr_dev &d = *x.shape;
    if (d.i8L == 7) BNR_LAUNCH_I8(7);
    else BNR_LAUNCH_I8(8);
}
static void launch_gram(bnr_exec &x, int s, hipStream_t st, bool timed)
{
    const bnr_dev &d = *x.shape;
    const int ntl = d.ntile * (d.ntile + 1) / 2;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timed) {
        while (timed && x.ev.size() < (size_t)(2 * (s + 1))) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) timed = false;       // no event: the launch goes ahead untimed (n_gram shows it)
            else x.ev.push_back(e);
        }
    }
    if (timed) {
        e0 = x.ev[2 * s]; e1 = x.ev[2 * s + 1];
        HIPNOTE(hipEventRecord(e0, st));
    }
    const dim3 ggrid(round_up(ntl * d.ksplit, 8) * x.nb);
    if (gram_on_i8(x) && x.gram_variant == 0) {
        launch_gram_i8(x, s, st, ggrid);
    } else
    if (d.gram_kg == 4) {
        if (x.nb == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gram<bnr_one, 4>), ggrid, dim3(1024), 0, st, bnr_one{d}, s, 1);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gram<bnr_many, 4>), ggrid, dim3(1024), 0, st, bnr_many{x.cds}, s, x.nb);
    } else {
        // Two kernels, the same partial tiles bit for bit (same K split, K-groups and accumulation order): k_gram8 (8-column batches,
        // three workgroups = 6 waves per SIMD) feeds the f64 MFMA pipe from more waves and leaves room on the CU for the scalar
        // branch beside it (8 chains: 205 vs 211 us alone, 221 vs 267 us beside the scalar branch) -- when the launch has more
        // workgroups than two per CU.  A launch that fits in one round (one chain: 252 workgroups at the headline size) never reaches
        // that occupancy and is better off with k_gram's 16-column batches = half the barriers (33.5 vs 36.0 us; n=500, V=300: 228 vs 237).
        const bool wide = x.gram_variant ? x.gram_variant == 16 : ((long)x.nb * ntl * d.ksplit <= 2L * x.ncu);
        if (!wide) {
            if (x.nb == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gram8<bnr_one>), ggrid, dim3(512), 0, st, bnr_one{d}, s, 1);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gram8<bnr_many>), ggrid, dim3(512), 0, st, bnr_many{x.cds}, s, x.nb);
        } else {
            if (x.nb == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gram<bnr_one, 2>), ggrid, dim3(512), 0, st, bnr_one{d}, s, 1);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gram<bnr_many, 2>), ggrid, dim3(512), 0, st, bnr_many{x.cds}, s, x.nb);
        }
    }
    if (timed) HIPNOTE(hipEventRecord(e1, st));
    if (!left_looking(x) && !reduce_in_chol(x)) BNR_LAUNCH(k_gram_reduce, dim3(ntl, 8, x.nb), dim3(256), 0, st, x, s);
}
static int build_qlist(bnr_exec &) { return BNR_OK; }
static bool wants_qlist(const bnr_exec &x);
static bool wants_qlist(const bnr_exec &) { return false; }
static void launch_rhs(bnr_exec &x, int s) { BNR_LAUNCH(k_rhs, dim3(x.shape->n_pad / 64, 1, x.nb), dim3(256), 0, x.stream, x, s); }
static void launch_chol(bnr_exec &x, int s, hipStream_t st, int rec_p = -1, hipEvent_t rec_ev = nullptr)
{
    const int nbk = x.shape->n_pad / BNR_NB;
    int nlaunch = 0;
    // (after launch number rec_p of the factorization the event rec_ev is recorded: the scalar branch's k_node waits for it)
#define BNR_CHOL_LAUNCHED() do { if (rec_ev && nlaunch == rec_p) HIPNOTE(hipEventRecord(rec_ev, st)); ++nlaunch; } while (0)
    if (x.factor_variant == 2 || x.factor_variant == 3 || (x.factor_variant < 0 && two_panel_default(x))) {
        // two panels per launch (k_chol_step2): half the launches on the critical path, the same arithmetic; variant 3 (the choice for
        // large n): the whole trailing matrix is read and written at every other launch only, with K = 128
        const int ncu = x.ncu, lazy = x.factor_variant != 2;
        for (int P = 0; P < nbk / 2; ++P) {
            const int nsup = bnr_chol2_nsuper(nbk, P, lazy), freecu = ncu - x.nb * nbk;
            const int spw = (freecu > 0 && nbk <= 24 && nsup > 0) ? std::min(x.spw_cap, std::max(1, (x.nb * nsup + freecu - 1) / freecu)) : 1;
            BNR_LAUNCH(k_chol_step2, dim3(x.nb, nbk + (nsup + spw - 1) / spw), dim3(256), 0, st, x, P, s, spw, lazy);
            BNR_CHOL_LAUNCHED();
        }
        return;
    }
    // update workgroups: one 32 x 32 block each while panels + updates of all members fit the chip in one round (two 256-thread
    // workgroups per CU); otherwise (large n, groups) 64 x 64 super blocks
    const int ncu = x.ncu, fuse0 = reduce_in_chol(x) ? 1 : 0;
    // groups of up to 8: the steps that need only E / n_pad / counters get them by value in the kernel arguments (bnr_few)
    // (measured: 384.5 -> 381.6 us per sweep of 8 chains, profiles/round5_experiments_notes.txt G)
    const bool few_ok = x.nb >= 2 && x.nb <= 8 && x.cds_pin;
    bnr_few few{};
    if (few_ok) {
        for (int i = 0; i < 8; ++i) { const bnr_dev &m = x.cds_pin[i < x.nb ? i : 0]; few.E[i] = m.E; few.counters[i] = m.counters; few.dbg[i] = m.dbg; }
        few.n_pad = x.shape->n_pad;
    }
    for (int p = 0; p < nbk; ++p) {
        const int npan = bnr_chol_npanel(nbk, p), ntile = bnr_chol_ntile(nbk, p);
        if (p == 0 && fuse0) {
            // launch 0: the first panel (its workgroups sum the partial tiles of column block 0 themselves) beside the reduction of all the other tiles
            const int ntl = x.shape->ntile * (x.shape->ntile + 1) / 2;
            BNR_LAUNCH(k_chol_step, dim3(x.nb, npan + 8 * ntl), dim3(256), 0, st, x, p, s, 1, 1, 1);
            BNR_CHOL_LAUNCHED();
            continue;
        }
        const int room = std::max(64, 2 * ncu - x.nb * npan);
        if (x.nb * ntile > room) {                       // many blocks: 64 x 64 super blocks, one 32 x 32 block per wave
            const int nsup = bnr_chol_nsuper(nbk, p);
            // ... and as many of them per workgroup as it takes to keep the update workgroups on the CUs the panels leave
            // free (at most 4: they must stay shorter than a panel sweep; only while E is L2-sized -- for large n the update is
            // bandwidth-bound and wants every workgroup in flight at once: n=2000 408 vs 421 it/s)
            const int freecu = ncu - x.nb * npan;
            // ... except at the first steps behind launch 0 (p <= 4): there every super block gets a workgroup of its own even where that puts update workgroups on
            // CUs that hold a panel workgroup -- launch 0 has just streamed the Gram's partial tiles through the L2s, the blocks of E come from further away, and
            // three of them in sequence per workgroup made these launches end on their update workgroups (in-kernel stamps, 8 chains: last update workgroup ends
            // 11.0 / 10.7 / 7.3 / 7.5 us after the launch's start at p = 1..4 against 5.4-6.2 for the panel workgroups; with one block each 4.7 / 5.8 -- per sweep
            // 385.9 -> 382.4 us in bench.py, 385.1 -> 376.6 in tools/ab_opt.py; one block each at EVERY step is slower again: 380.6 there)
            const int cap = p <= 4 ? 1 : x.spw_cap;
            const int spw = (freecu > 0 && nbk <= 24) ? std::min(cap, std::max(1, (x.nb * nsup + freecu - 1) / freecu)) : 1;
            if (few_ok && p >= 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chol_step<bnr_few>), dim3(x.nb, npan + (nsup + spw - 1) / spw), dim3(256), 0, st, few, p, s, 0, spw, fuse0);
            else BNR_LAUNCH(k_chol_step, dim3(x.nb, npan + (nsup + spw - 1) / spw), dim3(256), 0, st, x, p, s, 0, spw, fuse0);
        } else {
            if (few_ok && p >= 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chol_step<bnr_few>), dim3(x.nb, npan + ntile), dim3(256), 0, st, few, p, s, 1, 1, fuse0);
            else BNR_LAUNCH(k_chol_step, dim3(x.nb, npan + ntile), dim3(256), 0, st, x, p, s, 1, 1, fuse0);
        }
        BNR_CHOL_LAUNCHED();
    }
#undef BNR_CHOL_LAUNCHED
}
// Number of launches of the factorization, and where the scalar branch is ordered behind the critical chain by default.
static int chol_launches(const bnr_exec &x)
{
    const int nbk = x.shape->n_pad / BNR_NB;
    return (x.factor_variant == 2 || x.factor_variant == 3 || (x.factor_variant < 0 && two_panel_default(x))) ? nbk / 2 : nbk;
}
// Ordering the scalar branch behind points of the critical chain pays where the factorization is the longer of the two chains behind the Gram by a margin (each
// edge costs the waiting queue ~9 us in a replayed graph): measured per sweep, edges on / off (profiles/round6_experiments_notes.txt A) -- headline shape one chain
// 175.1 / 180.1 us, 8 chains 371.0 / 378.0; n = 2000, V = 200: 2 075 / 2 134; but n = 200, V = 50: 100.4 / 94.6, n = 70: 90.4 / 83.9, n = 500, V = 300 (the
// scalar branch is the longer chain): 471 / 394.  Rough per-launch and per-kernel times of this chip stand in for a measurement at creation time.
static double fact_est_us(const bnr_exec &x)
{
    const bool two_panel = chol_launches(x) != x.shape->n_pad / BNR_NB;
    return chol_launches(x) * (two_panel ? 18.5 : (x.nb > 1 ? 8.0 : 7.0));
}
static double scalar_est_us(const bnr_exec &x)           // k_tail + k_node + X pass + k_rhs
{
    return 45.0 + 0.0025 * x.shape->q * (x.nb > 1 ? 1.5 : 1.0) + 0.15 * x.shape->V * (x.nb > 1 ? 1.3 : 1.0);
}
static bool order_scalar_branch(const bnr_exec &x) { return x.overlap != 0 && fact_est_us(x) - scalar_est_us(x) >= 25.0; }
static int tail_after_default(const bnr_exec &x) { return order_scalar_branch(x) ? 0 : -1; }
static int node_after_default(const bnr_exec &x)
{
    if (!order_scalar_branch(x)) return -1;
    // k_node starts when what is left of the factorization is about k_node + X pass + k_rhs long (+ 20 us), at most a quarter into it
    const double per = fact_est_us(x) / chol_launches(x);
    const int p = (int)((fact_est_us(x) - (scalar_est_us(x) - 20.0) - 20.0) / per);
    // (a group's scalar branch is the longer one -- k_tail 20-36 us beside launch 0, k_node 25, X pass 24-37, k_rhs 10 --: an eighth into the factorization, a chain alone a quarter;
    // measured per sweep of 8 chains with k_node behind launch 2 / 3 / 4 / 5 / 6: 367.2 / 368.1 / 368.6 / 374.7 / 383.0 us, one chain 4 / 5 / 6 / 7 / 8: 166.9 / 166.4 / 166.4 / 167.7 / 174.9)
    return std::max(0, std::min(p, chol_launches(x) / (x.nb > 1 ? 8 : 4)));
}
static int tail_after(const bnr_exec &x) { return x.tail_after == -2 ? tail_after_default(x) : x.tail_after; }
static int node_after(const bnr_exec &x) { const int v = x.node_after == -2 ? node_after_default(x) : x.node_after; return std::min(v, chol_launches(x) - 1); }
static void launch_solve(bnr_exec &x)
{
    BNR_LAUNCH(k_solve_w, dim3(x.shape->n_pad / 4, 1, x.nb), dim3(256), 0, x.stream, x);
    BNR_LAUNCH(k_solve_a4, dim3(x.shape->n_pad / BNR_NB, 1, x.nb), dim3(1024), (x.shape->n_pad + 32 * 33) * sizeof(double), x.stream, x);
}
static void launch_backproj(bnr_exec &x, int s, int flags)
{
    {
        // many rounds of workgroups (a lockstep group at large q): only the instructions per edge count -- one edge per lane of the drawing wave (k_backproj64)
        const size_t lds64 = ((size_t)x.shape->n_pad + 64 + (size_t)x.shape->R * 65 + (size_t)(3 * x.shape->R + 1) * 65) * sizeof(double);
        const bool many_rounds = (size_t)x.nb * x.shape->nblk_bp >= (size_t)8 * x.ncu;       // (measured: 2 528 chunks -1.3 %, 2 822 -1.3 %, 5 644 -2.4 %, 11 288 -3 %; 1 411 equal; 1 264 +0.4 %; a chain alone at the headline shape +4 %)
        if ((flags & 3) == 3 && lds64 <= 124 * 1024 && (x.wide_backproj == 1 || (x.wide_backproj < 0 && many_rounds))) {
            launch_late_backproj64(x, s, flags, lds64);
            return;
        }
    }
    size_t lds = std::max<size_t>(x.shape->n_pad + 64, (size_t)(3 * x.shape->R + 1) * 33) * sizeof(double);
    const int nslot = (x.nb * x.shape->nblk_bp <= 2 * x.ncu) ? 8 : 2;    // latency-bound launch: four drawing waves, split by sampler kind; throughput-bound: wave 0 alone (see the kernel)
    const int wide = (x.nb == 1 || x.shape->nblk_bp >= 1024) ? 256 : 0;    // four columns of X per wave and trip (see the kernel)
    BNR_LAUNCH(k_backproj, dim3(round_up(x.shape->nblk_bp, 8) * x.nb), dim3(256), lds, x.stream, x, s, flags | wide, x.nb, nslot);
}
static void launch_tail(bnr_exec &x, int s, int mask, int xg_src, unsigned wgs = 1)
{
#ifdef BNR_EXP_PAD
    { static const int pad = getenv("BNR_EXP_TAIL_PAD_US") ? atoi(getenv("BNR_EXP_TAIL_PAD_US")) : 0; if (mask == 1023) xg_src |= pad << 8; }
#endif
    const size_t rv = (size_t)x.shape->R * x.shape->V;
    const size_t lds_a = (mask & BNR_TAIL_EARLY) ? tail_a_bytes(x) : 0;       // Delta / M / inv(M) asked for here (hooks, a loaded row): bnr_tail_a's work matrices behind u
    if (rv <= BNR_TAIL_U_LDS) { BNR_LAUNCH(k_tail, dim3(wgs, 1, x.nb), dim3(BNR_TAIL_THREADS), rv * sizeof(double) + lds_a, x.stream, x, s, mask, xg_src); return; }
    // u beyond the LDS budget: the instantiation that reads it from the table row
    if (x.nb == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tail<bnr_one, false>), dim3(wgs, 1, 1), dim3(BNR_TAIL_THREADS), lds_a, x.stream, bnr_one{*x.shape}, s, mask, xg_src);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tail<bnr_many, false>), dim3(wgs, 1, x.nb), dim3(BNR_TAIL_THREADS), lds_a, x.stream, bnr_many{x.cds}, s, mask, xg_src);
}
// The scalar tail of sweep s with everything it needs: with split_sums the per-block partial sums of update_theta! / update_Lambda!
// (gibbs.jl:476, 603-605) are not computed by the back-projection on the critical chain but by a launch of their own in front of the tail --
// the same kernel with flags = 4, the same sums in the same order.
// By default for a chain alone (its scalar branch has slack: one chain -3 us per sweep at the headline size, -7 at n = 2000); a group's scalar branch
// is nearly critical already (8 chains: +32 us).
// ... and not where the sums are many blocks of work (n = 500, V = 300: 1411 blocks, one chain 412 -> 478 us per sweep with the split).
// ... nor where the scalar branch itself is the long pole (n_pad = 128: four panel steps; n = 70: 85.2 -> 87.2 us with the split).
static bool split_sums(const bnr_exec &x) { return x.split_sums == 1 || (x.split_sums < 0 && x.nb == 1 && x.shape->nblk_bp <= 3 * x.ncu && x.shape->n_pad >= 256); }
static void launch_full_tail(bnr_exec &x, int s)
{
    if (split_sums(x)) launch_backproj(x, s, 4);
    launch_tail(x, s, 1023, 0, 2);                          // two workgroups per chain: Delta, M, inv(M) (workgroup 1) beside the rest (workgroup 0)
}
static hipEvent_t next_event(bnr_exec &x)
{
    if (x.fj_next >= x.fj.size()) {
        hipEvent_t e = nullptr;
        HIPNOTE(hipEventCreateWithFlags(&e, hipEventDisableTiming));    // a failure is reported by the next check_launch()
        x.fj.push_back(e);
    }
    return x.fj[x.fj_next++];
}

// One sweep for plan slot s (gibbs_sample!, gibbs.jl:663-677).  The scalar tail of the PREVIOUS sweep (theta, Delta, M,
// mu, Lambda, pi and the carried sums) is issued at the head of this one, because together with update_tau2!/update_u_xi!
// and the X W pass it is independent of the Gram matrix X diag(S) X' of this sweep: the two branches run concurrently
// (fork/join on two streams, also inside the captured graph) and meet before the factorization.
//   branch A (stream):  tail(s-1) -> k_node(s) -> k_xpass(s) -> k_rhs(s)                 (~100 us of scalar/latency work)
//   branch B (stream2): k_gram(s) -> k_gram_reduce -> k_chol_step x nbk                  (the factorization needs no rhs)
//   joined:             k_solve_w -> k_solve_a4 -> k_backproj(s)
// For a lockstep group every launch covers all members (grid z): the sequential panel chain of the factorization and
// the launch latencies are paid once per sweep of the whole group.
static void launch_sweep(bnr_exec &x, int s, bool prev_tail)
{
    const bool timed = x.profiling != 0;
    const bool overlap = x.overlap != 0;
    hipStream_t sb = overlap ? x.stream2 : x.stream;
    const bool pipe = pipelined(x);
    hipEvent_t ej[3] = {nullptr, nullptr, nullptr};
    (void)pipe;
    if (overlap) {
        hipEvent_t ef = next_event(x);
        HIPNOTE(hipEventRecord(ef, x.stream));
        HIPNOTE(hipStreamWaitEvent(x.stream2, ef, 0));
        launch_gram(x, s, sb, timed);
        hipEvent_t eg = nullptr, en = nullptr;
        if (tail_after(x) == 0 && prev_tail) { HIPNOTE(hipEventRecord(eg = next_event(x), x.stream2)); }
        if (node_after(x) >= 0) en = next_event(x);
        launch_chol(x, s, sb, node_after(x), en);
        HIPNOTE(hipEventRecord(ej[0] = next_event(x), x.stream2));
        // the scalar branch is ordered behind points of the critical chain (see bnr_exec::tail_after / node_after)
        if (eg) HIPNOTE(hipStreamWaitEvent(x.stream, eg, 0));
        if (prev_tail) launch_full_tail(x, s - 1);
        if (en) HIPNOTE(hipStreamWaitEvent(x.stream, en, 0));
    } else
    if (prev_tail) launch_full_tail(x, s - 1);
    launch_node(x, s, 3);
    launch_xpass(x, s, 3);
    launch_rhs(x, s);
    if (overlap) { for (hipEvent_t e : ej) if (e) HIPNOTE(hipStreamWaitEvent(x.stream, e, 0)); }
    else { launch_gram(x, s, sb, timed); launch_chol(x, s, sb); }
    launch_solve(x);
    launch_backproj(x, s, split_sums(x) ? 3 : 7);
}

// with profiling on: after a batch, read the HIP events recorded around the k_gram launches of that batch (recorded on
// the stream the kernel runs on)
static int collect_gram_times(bnr_exec &x, int nsweeps)
{
    if (!x.profiling) return BNR_OK;
    HIPCHK(hipStreamSynchronize(x.stream));
    for (int s = 0; s < nsweeps && (size_t)(2 * s + 1) < x.ev.size(); ++s) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, x.ev[2 * s], x.ev[2 * s + 1]) == hipSuccess) { x.t_gram_acc += ms; x.n_gram += 1; }
    }
    return BNR_OK;
}

// Capture K sweeps (+ the plan-base advance) into a graph and instantiate it.  The kernels find their plan entry through
// pbase at run time, so a captured graph serves every later batch.
static int capture_sweeps(bnr_exec &x, int K, hipGraph_t *graph, hipGraphExec_t *gexec)
{
    x.fj_next = 0;
    HIPCHK(hipStreamBeginCapture(x.stream, hipStreamCaptureModeThreadLocal));   // other host threads may drive other handles meanwhile
    for (int s = 0; s < K; ++s) launch_sweep(x, s, true);
    hipLaunchKernelGGL(k_advance, dim3(x.nb), dim3(1), 0, x.stream, (const bnr_dev *)x.cds, K);
    HIPCHK(hipStreamEndCapture(x.stream, graph));
    HIPCHK(hipGraphInstantiate(gexec, *graph, nullptr, nullptr, 0));
    (void)hipGraphUpload(*gexec, x.stream);                                     // best effort: the first replay finds it resident
    (void)hipGetLastError();
    return BNR_OK;
}
// The graphs a run replays: a ladder of graph_k, graph_k/2, ..., 1 sweeps, so that a batch of any length is all replay with
// few launches (between two graph launches the GPU idles for ~30 us: 20 sweeps = 8 + 8 + 4, not 8 + 8 + 1 + 1 + 1 + 1).
// Built here, outside anybody's timed region.
static int exec_prepare(bnr_exec &x)
{
    if (wants_qlist(x)) { int rq = build_qlist(x); if (rq) return rq; }
    if (!x.use_graph || x.profiling || x.graph_k <= 0 || !x.ladder.empty()) return BNR_OK;
    for (int k = x.graph_k; k >= 1; k /= 2) {
        bnr_exec::rung r{k, nullptr, nullptr};
        int rc = capture_sweeps(x, k, &r.graph, &r.gexec);
        x.ladder.push_back(r);                               // also on failure: drop_graph releases what exists
        if (rc) { drop_graph(x); return rc; }
    }
    return BNR_OK;
}

static int launch_range(bnr_exec &x, int count)
{
    int done = 0;
    if (wants_qlist(x)) { int rq = build_qlist(x); if (rq) return rq; }
    if (x.use_graph && !x.profiling && x.graph_k > 0) {     // profiling records HIP events around k_gram: eager launches
        int rc = exec_prepare(x);
        if (rc) return rc;
        // smallest graphs first: the GPU starts after the host has enqueued ONE short graph and works on it while the long
        // ones are being enqueued (launching a 280-node graph costs the host a few hundred microseconds)
        std::vector<const bnr_exec::rung *> seq;                       // greedy decomposition, largest first ...
        for (const auto &r : x.ladder)
            while (count - done >= r.k) { seq.push_back(&r); done += r.k; }
        for (auto it = seq.rbegin(); it != seq.rend(); ++it) HIPCHK(hipGraphLaunch((*it)->gexec, x.stream));   // ... launched smallest first
        x.n_replayed += count;
        return BNR_OK;
    }
    const int r = count - done;
    x.fj_next = 0;
    for (int s = 0; s < r; ++s) launch_sweep(x, s, true);
    if (r > 0) {
        hipLaunchKernelGGL(k_advance, dim3(x.nb), dim3(1), 0, x.stream, (const bnr_dev *)x.cds, r);
        x.n_eager += r;
        int rc = collect_gram_times(x, r);
        if (rc) return rc;
    }
    return BNR_OK;
}

// (re)compute the carried sums rr, sig_q from 0-based row r (needed after init, load, hooks)
static int refresh_carried(bnr_chain *c, int r)
{
    if (c->carried_row == r) return BNR_OK;
    int rc = ensure_plan(c, 1);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->x.stream));
    c->plan_pin[0] = bnr_plan_entry{0u, r, r, 0};
    rc = upload_plan(c, 1);
    if (rc) return rc;
    launch_xpass(c->x, 0, 4);
    launch_tail(c->x, 0, 64 | 256, 1);
    HIPCHK(hipStreamSynchronize(c->x.stream));
    c->carried_row = r;
    return check_launch("refresh");
}

// status of a chain from its 16 event counters (host copy).  Hard failures (stream ordering, Cholesky) are sticky: the table is not the
// sampler's from there on.  The attempt cap of a rejection sampler is reported by the call in which it happened and only by that one
// (cap_seen): the rows were written with the samplers' fall-backs, the table stays usable, later calls are not failed for it.
static int status_of(const long long *cnt, long long *cap_seen)
{
    if (cnt[8] > 0) return fail(BNR_ERR_HIP, "stream ordering violated: the factorization started before the Gram branch finished");
    if (cnt[3] > 0) {
        char buf[160];
        snprintf(buf, sizeof buf, "Cholesky failed after the jitter ladder (node %lld, Psi %lld, M %lld, G+I %lld)", cnt[4], cnt[5], cnt[6], cnt[7]);
        return fail(BNR_ERR_CHOLESKY, buf);
    }
    const long long before = cap_seen ? *cap_seen : 0;
    if (cap_seen) *cap_seen = cnt[2];
    if (cnt[2] > before) return fail(BNR_ERR_SAMPLER_CAP, "a rejection sampler stopped at its attempt cap (" + std::to_string(cnt[2] - before) + " draws in this call, " + std::to_string(cnt[2]) + " since the chain was created): the values written are the samplers' fall-backs; the table stays valid and later calls are not failed for it");
    return BNR_OK;
}
static int fetch_status(bnr_chain *c)
{
    HIPCHK(hipMemcpyAsync(c->counters_host, c->d.counters, sizeof(long long) * 16, hipMemcpyDeviceToHost, c->x.stream));
    HIPCHK(hipStreamSynchronize(c->x.stream));
    return status_of(c->counters_host, &c->cap_seen);
}

int bnr_chain_init_prior(bnr_chain *c)
{
    if (!c) return fail(BNR_ERR_BAD_ARG, "NULL chain");
    HIPCHK(hipSetDevice(c->device));
    hipLaunchKernelGGL(k_init_prior, dim3(1), dim3(256), 0, c->x.stream, c->d);
    int rc = check_launch("k_init_prior");
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->x.stream));
    c->iter = 1;
    c->carried_row = -1;
    c->next_row = 2;
    return BNR_OK;
}

// run! (gibbs.jl:849-864): builds the plan exactly as the reference loop walks (i, j), then enqueues the sweeps.
static int enqueue_run(bnr_chain *c, int first_index, int nburn, int total, int purge_burn, hipStream_t st, bool upload = true)
{
    if (first_index < 2 || total < first_index - 1) return fail(BNR_ERR_BAD_ARG, "need first_index>=2 and total>=first_index-1");
    HIPCHK(hipSetDevice(c->device));
    const int count = total - first_index + 1;
    if (count <= 0) { c->next_row = first_index; return BNR_OK; }
    int rc = refresh_carried(c, first_index - 2);
    if (rc) return rc;
    rc = ensure_plan(c, count + 1);
    if (rc) return rc;
    c->plan_pin[0] = bnr_plan_entry{0u, 0, 0, 2};      // placeholder: the tail issued at the head of the first sweep is a no-op
    int j = first_index, prev = first_index - 2, s = 1, maxrow = 0;
    // purge_burn == 1: the ring is rows {1, 2}; the reference writes row 2, copies it to row 1 and reads it back from there
    // (gibbs.jl:857-860) -- a sweep here would read and write the same row.  Ping-pong instead between row 2 and the hidden
    // scratch row behind the table, without copies, phased so that the LAST wrapping iteration L lands in the scratch row; its
    // tail copies that state to rows 1 and 2 (what the reference's table holds at that point) and the next sweep reads the
    // scratch row while it writes row 2.
    const bool ring1 = purge_burn == 1 && first_index == 2 && nburn > 2;
    const int L = ring1 ? std::min(nburn - 1, total) : 0, scratch = c->d.tot;
    for (int i = first_index; i <= total; ++i, ++s) {
        c->iter += 1;
        if (ring1 && i <= L) {
            const int T = ((L - i) % 2 == 0) ? scratch : 1;
            c->plan_pin[s] = bnr_plan_entry{(uint32_t)c->iter, T, prev, i == L ? (1 | 4) : 0};
            prev = T;
            maxrow = std::max(maxrow, 2);
            continue;                                                             // j stays 2 (j = 1; j = j + 1)
        }
        bnr_plan_entry e{(uint32_t)c->iter, j - 1, prev, 0};
        maxrow = std::max(maxrow, j);
        prev = j - 1;
        if (purge_burn > 0 && i < nburn && j == purge_burn + 1) { e.wrap = 1; j = 1; }   // copy_table!(state,1,j); j = 1
        j = j + 1;
        c->plan_pin[s] = e;
    }
    if (maxrow > c->d.tot) { c->iter -= count; return fail(BNR_ERR_BAD_ARG, "run would write past the table (tot_save too small)"); }
    if (upload) {                                      // (a group stages all members' plans in one copy: upload_group_plans)
        rc = upload_plan(c, count + 1, st);            // the caller sets the plan base of all members with ONE k_setbase launch
        if (rc) return rc;
    }
    c->next_row = j;
    c->carried_row = -1;
    return BNR_OK;
}

// issue `count` planned sweeps on x (a chain or a lockstep group), ticking cb every prog_freq iterations
// (gibbs.jl:854-856), then the scalar tail of the last sweep; returns with the stream drained
static int run_exec(bnr_exec &x, int first_index, int count, int prog_freq, bnr_progress_cb cb, void *user)
{
    int rc;
    hipLaunchKernelGGL(k_setbase, dim3(x.nb), dim3(1), 0, x.stream, (const bnr_dev *)x.cds, 1);
    if (x.lflags) HIPCHK(hipMemsetAsync(x.lflags, 0, sizeof(unsigned long long) * 64, x.stream));
    x.t_gram_acc = 0; x.n_gram = 0; x.n_replayed = 0; x.n_eager = 0;
    hipEvent_t r0 = nullptr, r1 = nullptr;
    if (x.profiling) { HIPCHK(hipEventCreate(&r0)); HIPCHK(hipEventCreate(&r1)); HIPNOTE(hipEventRecord(r0, x.stream)); }
    if (cb && prog_freq > 0) {
        int s = 0;
        while (s < count) {
            int i = first_index + s;
            int next_tick = ((i + prog_freq - 1) / prog_freq) * prog_freq;
            int seg = std::min(count - s, next_tick - i + 1);
            rc = launch_range(x, seg);
            if (rc) return rc;
            s += seg;
            if ((first_index + s - 1) % prog_freq == 0) {
                HIPCHK(hipStreamSynchronize(x.stream));
                cb(user, (int64_t)s);
            }
        }
    } else {
        rc = launch_range(x, count);
        if (rc) return rc;
    }
    if (count > 0) launch_full_tail(x, -1);                            // scalar tail of the last sweep
    if (x.profiling) HIPNOTE(hipEventRecord(r1, x.stream));
    // the members' event counters: one gather + one copy for the whole run call (read by the caller after the sync below)
    hipLaunchKernelGGL(k_gather_counters, dim3(x.nb), dim3(16), 0, x.stream, (const bnr_dev *)x.cds, x.status_dev);
    HIPCHK(hipMemcpyAsync(x.status_pin, x.status_dev, sizeof(long long) * 16 * x.nb, hipMemcpyDeviceToHost, x.stream));
    rc = check_launch("sweep");
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(x.stream));
    if (x.profiling && count > 0) {
        float ms = 0;
        HIPNOTE(hipEventElapsedTime(&ms, r0, r1));
        x.t_iter_us = 1e3 * ms / count; x.n_iter = count;
        x.t_gram_us = x.n_gram ? 1e3 * x.t_gram_acc / x.n_gram : 0;
        (void)hipEventDestroy(r0); (void)hipEventDestroy(r1);
    }
    return BNR_OK;
}

int bnr_chain_run(bnr_chain *c, int32_t first_index, int32_t nburn, int32_t total, int32_t purge_burn,
                  int32_t prog_freq, bnr_progress_cb cb, void *user, int32_t *next_row)
{
    if (!c) return fail(BNR_ERR_BAD_ARG, "NULL chain");
    if (c->pending) return fail(BNR_ERR_BAD_ARG, "an asynchronous run is pending; call bnr_chain_sync first");
    int rc = enqueue_run(c, first_index, nburn, total, purge_burn, c->x.stream);
    if (rc) return rc;
    const int count = total - first_index + 1;
    if (count <= 0) { if (next_row) *next_row = c->next_row; return fetch_status(c); }
    rc = run_exec(c->x, first_index, count, prog_freq, cb, user);
    if (rc) return rc;
    c->carried_row = c->plan_pin[count].row;
    if (next_row) *next_row = c->next_row;
    return status_of(c->x.status_pin, &c->cap_seen);
}

// ------------------------------------------------------------------------------------------ lockstep groups
// Several equally shaped chains on one GPU advance together: every kernel of a sweep is launched once for the whole
// group (blockIdx.z = member).  Chains stay independent (own seed, own trace, gibbs.jl:928); results are bitwise those of
// running each chain alone.  What the group buys: the launch latencies and the sequential panel chain of the n x n
// factorization (the critical path of ONE chain) are paid once per sweep of all members, and the Gram kernels of the
// members fill the chip back to back.
int bnr_group_create(bnr_chain *const *chains, int32_t nchains, bnr_group **out)
{
    if (!chains || !out || nchains < 1 || nchains > 1024) return fail(BNR_ERR_BAD_ARG, "need 1 <= nchains <= 1024");
    for (int i = 0; i < nchains; ++i) {
        bnr_chain *c = chains[i];
        if (!c) return fail(BNR_ERR_BAD_ARG, "NULL chain in group");
        if (c->group) return fail(BNR_ERR_BAD_ARG, "chain already belongs to a group");
        if (c->pending) return fail(BNR_ERR_BAD_ARG, "an asynchronous run is pending");
        for (int k = 0; k < i; ++k) if (chains[k] == c) return fail(BNR_ERR_BAD_ARG, "chain listed twice");
        const bnr_dev &a = chains[0]->d, &b = c->d;
        if (c->device != chains[0]->device || a.n != b.n || a.V != b.V || a.R != b.R)
            return fail(BNR_ERR_BAD_ARG, "chains of a group must live on one device and have equal n, V, R");
    }
    HIPCHK(hipSetDevice(chains[0]->device));
    bnr_group *g = new bnr_group();
    g->m.assign(chains, chains + nchains);
    int rc = exec_init(g->x, chains[0]->device, nchains, &chains[0]->d);
    if (rc) { exec_free(g->x); delete g; return rc; }
    for (size_t i = 0; i < g->m.size(); ++i) g->x.cds_pin[i] = g->m[i]->d;   // host copy from the start: launch choices made at capture time (group_shares_x) read it
    for (bnr_chain *c : g->m) c->group = g;
    *out = g;
    return BNR_OK;
}

int bnr_group_destroy(bnr_group *g)
{
    if (!g) return BNR_OK;
    (void)hipSetDevice(g->x.device);
    for (bnr_chain *c : g->m) c->group = nullptr;
    exec_free(g->x);
    delete g;
    return BNR_OK;
}

// the members' descriptors -> the device array the group's kernels index (pinned staging, asynchronous on the group's stream)
static int upload_members(bnr_group *g)
{
    for (size_t i = 0; i < g->m.size(); ++i) g->x.cds_pin[i] = g->m[i]->d;
    HIPCHK(hipMemcpyAsync(g->x.cds, g->x.cds_pin, sizeof(bnr_dev) * g->m.size(), hipMemcpyHostToDevice, g->x.stream));
    return BNR_OK;
}

// all members' plans of a run call: one pinned staging buffer, ONE host-to-device copy, handed out by a kernel (after upload_members: it reads
// the members' plan pointers from the device descriptors); the plan bases are set by run_exec's k_setbase
static int upload_group_plans(bnr_group *g, int count)
{
    bnr_exec &x = g->x;
    const int nb = (int)g->m.size();
    if (x.gplan_cap < count) {
        if (x.gplan_pin) (void)hipHostFree(x.gplan_pin);
        if (x.gplan_dev) (void)hipFree(x.gplan_dev);
        x.gplan_pin = nullptr; x.gplan_dev = nullptr; x.gplan_cap = 0;
        const int cap = std::max(count, 64) * 2;
        HIPCHK(hipHostMalloc((void **)&x.gplan_pin, sizeof(bnr_plan_entry) * (size_t)cap * nb));
        HIPCHK(hipMalloc((void **)&x.gplan_dev, sizeof(bnr_plan_entry) * (size_t)cap * nb));
        x.gplan_cap = cap;
    } else HIPCHK(hipStreamSynchronize(x.stream));          // the staging buffer of the previous call has been consumed
    for (int i = 0; i < nb; ++i) memcpy(x.gplan_pin + (size_t)i * x.gplan_cap, g->m[i]->plan_pin, sizeof(bnr_plan_entry) * count);
    // one copy of the used part of every member's slot (the slots are gplan_cap apart: copy the whole span when it is small, else per member)
    if ((size_t)x.gplan_cap * nb <= 4 * (size_t)count * nb)
        HIPCHK(hipMemcpyAsync(x.gplan_dev, x.gplan_pin, sizeof(bnr_plan_entry) * (size_t)x.gplan_cap * nb, hipMemcpyHostToDevice, x.stream));
    else
        for (int i = 0; i < nb; ++i)
            HIPCHK(hipMemcpyAsync(x.gplan_dev + (size_t)i * x.gplan_cap, x.gplan_pin + (size_t)i * x.gplan_cap, sizeof(bnr_plan_entry) * count, hipMemcpyHostToDevice, x.stream));
    hipLaunchKernelGGL(k_scatter_plans, dim3(nb), dim3(256), 0, x.stream, (const bnr_dev *)x.cds, (const bnr_plan_entry *)x.gplan_dev, x.gplan_cap, count);
    return BNR_OK;
}

int bnr_group_run(bnr_group *g, int32_t first_index, int32_t nburn, int32_t total, int32_t purge_burn,
                  int32_t prog_freq, bnr_progress_cb cb, void *user, int32_t *next_row)
{
    if (!g) return fail(BNR_ERR_BAD_ARG, "NULL group");
    if (g->m.empty()) return fail(BNR_ERR_BAD_ARG, "the group was dissolved (one of its chains was destroyed)");
    HIPCHK(hipSetDevice(g->x.device));
    for (bnr_chain *c : g->m) {
        if (c->pending) return fail(BNR_ERR_BAD_ARG, "an asynchronous run is pending on a member");
        if (c->d.tot != g->m[0]->d.tot) return fail(BNR_ERR_BAD_ARG, "members of a group must have tables of equal length");
    }
    if (first_index < 2 || total < first_index - 1) return fail(BNR_ERR_BAD_ARG, "need first_index>=2 and total>=first_index-1");
    const int count = total - first_index + 1;
    int rc;
    for (bnr_chain *c : g->m) {
        rc = enqueue_run(c, first_index, nburn, total, purge_burn, g->x.stream, false);     // every member walks the same (i, j) schedule
        if (rc) return rc;
    }
    if (count <= 0) { if (next_row) *next_row = g->m[0]->next_row; return BNR_OK; }
    rc = upload_members(g);
    if (rc) return rc;
    rc = upload_group_plans(g, count + 1);
    if (rc) return rc;
    rc = run_exec(g->x, first_index, count, prog_freq, cb, user);
    if (rc) return rc;
    for (size_t i = 0; i < g->m.size(); ++i) {
        g->m[i]->carried_row = g->m[i]->plan_pin[count].row;
        // the most severe member status wins: a sampler that stopped at its attempt cap (a warning to the callers: the table stays valid) must not hide a member whose
        // factorization failed or whose launches ran out of order in the same call
        std::string msg_before = g_err;
        int r2 = status_of(g->x.status_pin + 16 * i, &g->m[i]->cap_seen);
        if (r2 && (!rc || (rc == BNR_ERR_SAMPLER_CAP && r2 != BNR_ERR_SAMPLER_CAP))) rc = r2;
        else if (r2) g_err = msg_before;                     // keep the message of the status that is returned
    }
    if (next_row) *next_row = g->m[0]->next_row;
    return rc;
}

// Options of the measured experiments of rounds 3-4 (kernels and host paths removed from the tree in round 5: tools/experiments/README.md): refused by name, so that
// nothing a user can set starts a kernel that polls device memory
static bool experimental_option(const char *name)
{
    for (const char *e : {"nop_fork", "resv_mask", "crit_origin", "group_backproj", "linear", "linear_merge", "linear_debug", "pipeline", "gate_us"})
        if (!strcmp(name, e)) return true;
    return false;
}
static int exec_set_option(bnr_exec &x, const char *name, int64_t value)
{
    if (!strcmp(name, "graph")) { x.use_graph = (int)value; return BNR_OK; }
    if (!strcmp(name, "overlap")) { x.overlap = (int)value; drop_graph(x); return BNR_OK; }
    if (experimental_option(name)) return fail(BNR_ERR_BAD_ARG, std::string("option ") + name + " belongs to the measured experiments of rounds 3-4, which are no longer part of the library (tools/experiments/README.md)");
    if (!strcmp(name, "gram_variant")) {
        const bool ok = value == 0 || value == 8 || value == 16;
        const char *msg = "gram_variant must be 0 (auto), 8 or 16 (the experimental kernels 9..14 of rounds 3-4 are no longer part of the library: tools/experiments/README.md)";
        if (!ok) return fail(BNR_ERR_BAD_ARG, msg);
        x.gram_variant = (int)value; drop_graph(x); return BNR_OK;
    }
    if (!strcmp(name, "fuse_reduce")) {
        if (value < -1 || value > 1) return fail(BNR_ERR_BAD_ARG, "fuse_reduce must be -1 (default: on), 0 or 1");
        x.fuse_reduce = (int)value; drop_graph(x); return BNR_OK;
    }
    if (!strcmp(name, "group_xpass")) {
        if (value < -1 || value > 1) return fail(BNR_ERR_BAD_ARG, "group_xpass must be -1 (default: on), 0 or 1");
        x.group_xpass = (int)value; drop_graph(x); return BNR_OK;
    }
    if (!strcmp(name, "split_sums")) {
        if (value < -1 || value > 1) return fail(BNR_ERR_BAD_ARG, "split_sums must be -1 (default: a chain alone), 0 or 1");
        x.split_sums = (int)value; drop_graph(x); return BNR_OK;
    }
    if (!strcmp(name, "wide_backproj")) {
        if (value < -1 || value > 1) return fail(BNR_ERR_BAD_ARG, "wide_backproj must be -1 (default: launches of many rounds), 0 or 1");
        x.wide_backproj = (int)value; drop_graph(x); return BNR_OK;
    }
    if (!strcmp(name, "tail_after")) {
        if (value < -2 || value > 0) return fail(BNR_ERR_BAD_ARG, "tail_after must be -2 (default), -1 (no ordering) or 0 (behind the Gram)");
        x.tail_after = (int)value; drop_graph(x); return BNR_OK;
    }
    if (!strcmp(name, "node_after")) {
        if (value < -2 || value > 1024) return fail(BNR_ERR_BAD_ARG, "node_after must be -2 (default), -1 (no ordering) or a launch number of the factorization");
        x.node_after = (int)value; drop_graph(x); return BNR_OK;
    }
    if (!strcmp(name, "spw_cap")) {
        if (value < 1 || value > 4) return fail(BNR_ERR_BAD_ARG, "spw_cap must be 1..4");
        x.spw_cap = (int)value; drop_graph(x); return BNR_OK;
    }
    if (!strcmp(name, "factor_variant")) {
        const bool ok = value >= -1 && value <= 3 && value != 1;
        if (!ok) return fail(BNR_ERR_BAD_ARG, "factor_variant must be -1 (auto), 0 (right-looking), 2 (right-looking, two panels per launch) or 3 (2 with the K = 128 trailing update); 1 (left-looking), 4 (data-flow, one launch) and 5 (one workgroup per chain, n_pad <= 128) were experiments of rounds 3-4, no longer part of the library (tools/experiments/README.md)");
        x.factor_variant = (int)value; drop_graph(x); return BNR_OK;
    }
    if (!strcmp(name, "graph_k")) { if (value < 1 || value > 256) return fail(BNR_ERR_BAD_ARG, "graph_k out of range"); x.graph_k = (int)value; drop_graph(x); return BNR_OK; }
    if (!strcmp(name, "profiling")) { if (x.profiling != (int)value) drop_graph(x); x.profiling = (int)value; return BNR_OK; }
    return fail(BNR_ERR_BAD_ARG, std::string("unknown option ") + name);
}
static int exec_last_timing(bnr_exec &x, int which, double *avg_us, int64_t *launches)
{
    if (which == 0) { *avg_us = x.t_iter_us; if (launches) *launches = x.n_iter; }
    else if (which == 1) { *avg_us = x.t_gram_us; if (launches) *launches = x.n_gram; }
    else if (which == 2) { *avg_us = (double)x.n_eager; if (launches) *launches = x.n_replayed; }   // how the last run call was issued
    else if (which == 3) { *avg_us = x.shape->X8 ? 1.0 : 0.0; if (launches) *launches = x.shape->X8 ? 1 : 0; }   // do the X passes read a byte image of X
    else if (which == 4) { const bool on = gram_on_i8(x) && x.gram_variant == 0; *avg_us = on ? 1.0 : 0.0; if (launches) *launches = on ? x.shape->i8L : 0; }   // does the Gram run on the i8 matrix pipe (binary model matrix), with how many digit planes of S
    else return fail(BNR_ERR_BAD_ARG, "which must be 0 .. 4");
    return BNR_OK;
}
// Replay both captured graphs ONCE on scratch rows, results discarded: the first replay of an instantiated graph pays for the
// runtime's own setup (kernel-argument and packet buffers of ~300 nodes), and the clocks of an idle GPU take a few ms of
// work to come up -- neither belongs into anybody's first run call.  The discarded sweeps are real sweeps from the members'
// latest state into the two hidden rows behind each table; the table, the iteration counter and the event counters are
// untouched, the carried sums are marked stale (the next run call re-derives them as after a load).
// Skipped silently for a member without a state yet (init_prior / run not called).
static int ladder_sweeps(const bnr_exec &x)
{
    int K = 0;
    for (int k = x.graph_k; k >= 1; k /= 2) K += k;
    return K;
}
static int prime_graphs_inner(bnr_exec &x, const std::vector<bnr_chain *> &members, std::vector<std::vector<long long>> &saved);
static int prime_graphs(bnr_exec &x, const std::vector<bnr_chain *> &members)
{
    // whatever happens in there, the members' event counters are put back as they were (the discarded sweeps must not count)
    std::vector<std::vector<long long>> saved;
    int rc = prime_graphs_inner(x, members, saved);
    for (size_t i = 0; i < saved.size() && i < members.size(); ++i) {
        if ((hipMemcpyAsync(members[i]->d.counters, saved[i].data(), sizeof(long long) * 16, hipMemcpyHostToDevice, x.stream) != hipSuccess || hipStreamSynchronize(x.stream) != hipSuccess) && !rc)
            rc = fail(BNR_ERR_HIP, "restoring the event counters failed");
        members[i]->carried_row = -1;
    }
    return rc;
}
static int prime_graphs_inner(bnr_exec &x, const std::vector<bnr_chain *> &members, std::vector<std::vector<long long>> &saved)
{
    if (x.ladder.empty() || x.profiling || !x.use_graph) return BNR_OK;
    int K = 0;
    for (const auto &r : x.ladder) K += r.k;
    for (bnr_chain *c : members) if (c->iter < 1 || c->next_row < 2 || c->next_row - 2 >= c->d.tot || c->pending) return BNR_OK;
    for (bnr_chain *c : members) {
        int rc = ensure_plan(c, K + 1);
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(c->x.stream));
        saved.emplace_back(16);
        HIPCHK(hipMemcpyAsync(saved.back().data(), c->d.counters, sizeof(long long) * 16, hipMemcpyDeviceToHost, c->x.stream));
        HIPCHK(hipStreamSynchronize(c->x.stream));
        const int src = c->next_row - 2, scr0 = c->d.tot;
        rc = refresh_carried(c, src);                       // uses plan slot 0 itself: before the plan below is written
        if (rc) return rc;
        c->plan_pin[0] = bnr_plan_entry{0u, 0, 0, 2};
        for (int s = 1; s <= K; ++s)
            c->plan_pin[s] = bnr_plan_entry{(uint32_t)(c->iter + s), scr0 + (s & 1), s == 1 ? src : scr0 + ((s - 1) & 1), 0};
        rc = upload_plan(c, K + 1, x.stream);
        if (rc) return rc;
    }
    // the device copies of the descriptors (k_setbase / k_advance read them also when the sweep kernels take the struct by value): a GROUP's
    // array is otherwise only filled by its first run call -- a group of ONE that was prepared before it ran replayed its graphs on garbage
    for (size_t i = 0; i < members.size(); ++i) x.cds_pin[i] = members[i]->d;
    HIPCHK(hipMemcpyAsync(x.cds, x.cds_pin, sizeof(bnr_dev) * members.size(), hipMemcpyHostToDevice, x.stream));
    hipLaunchKernelGGL(k_setbase, dim3(x.nb), dim3(1), 0, x.stream, (const bnr_dev *)x.cds, 1);
    for (const auto &r : x.ladder) HIPCHK(hipGraphLaunch(r.gexec, x.stream));
    HIPCHK(hipStreamSynchronize(x.stream));
    return check_launch("prime");
}

int bnr_group_prepare(bnr_group *g)
{
    if (!g) return fail(BNR_ERR_BAD_ARG, "NULL group");
    if (g->m.empty()) return fail(BNR_ERR_BAD_ARG, "the group was dissolved (one of its chains was destroyed)");
    HIPCHK(hipSetDevice(g->x.device));
    for (bnr_chain *c : g->m) if (c->d.tot != g->m[0]->d.tot) return fail(BNR_ERR_BAD_ARG, "members of a group must have tables of equal length");
    int rc;
    for (bnr_chain *c : g->m) { rc = ensure_plan(c, ladder_sweeps(g->x) + 1); if (rc) return rc; }   // BEFORE the capture: a regrown plan drops graphs
    for (size_t i = 0; i < g->m.size(); ++i) g->x.cds_pin[i] = g->m[i]->d;      // the capture's launch choices (group_shares_x: byte_x of the members) read the host copies
    rc = exec_prepare(g->x);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(g->x.stream));
    rc = check_launch("prepare");
    if (rc) return rc;
    return prime_graphs(g->x, g->m);
}
int bnr_chain_prepare(bnr_chain *c)
{
    if (!c) return fail(BNR_ERR_BAD_ARG, "NULL chain");
    if (c->pending) return fail(BNR_ERR_BAD_ARG, "an asynchronous run is pending");
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure_plan(c, ladder_sweeps(c->x) + 1);                 // BEFORE the capture: a regrown plan drops graphs
    if (rc) return rc;
    rc = exec_prepare(c->x);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->x.stream));
    rc = check_launch("prepare");
    if (rc) return rc;
    return prime_graphs(c->x, std::vector<bnr_chain *>{c});
}
int bnr_group_set_option(bnr_group *g, const char *name, int64_t value)
{
    if (!g || !name) return fail(BNR_ERR_BAD_ARG, "NULL argument");
    return exec_set_option(g->x, name, value);
}
int bnr_group_last_timing(bnr_group *g, int32_t which, double *avg_us, int64_t *launches)
{
    if (!g || !avg_us) return fail(BNR_ERR_BAD_ARG, "NULL argument");
    return exec_last_timing(g->x, which, avg_us, launches);
}

int bnr_chain_run_async(bnr_chain *c, int32_t first_index, int32_t nburn, int32_t total, int32_t purge_burn)
{
    if (!c) return fail(BNR_ERR_BAD_ARG, "NULL chain");
    if (c->pending) return fail(BNR_ERR_BAD_ARG, "an asynchronous run is already pending");
    int rc = enqueue_run(c, first_index, nburn, total, purge_burn, c->x.stream);
    if (rc) return rc;
    const int count = total - first_index + 1;
    int saved = c->x.profiling;
    c->x.profiling = 0;
    hipLaunchKernelGGL(k_setbase, dim3(1), dim3(1), 0, c->x.stream, (const bnr_dev *)c->x.cds, 1);
    rc = launch_range(c->x, count);
    if (!rc && count > 0) launch_full_tail(c->x, -1);
    c->x.profiling = saved;
    if (rc) return rc;
    c->pending = true;
    if (count > 0) c->carried_row = -2 - c->plan_pin[count].row;        // becomes valid at sync
    return check_launch("sweep");
}

int bnr_chain_sync(bnr_chain *c, int32_t *next_row)
{
    if (!c) return fail(BNR_ERR_BAD_ARG, "NULL chain");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->x.stream));
    if (c->pending) {
        c->pending = false;
        if (c->carried_row <= -2) c->carried_row = -2 - c->carried_row;
    }
    if (next_row) *next_row = c->next_row;
    return fetch_status(c);
}

// ------------------------------------------------------------------------------------------ test hooks
static int hook_begin(bnr_chain *c, int row, int64_t iter, bool need_carried)
{
    if (!c) return fail(BNR_ERR_BAD_ARG, "NULL chain");
    if (c->pending) return fail(BNR_ERR_BAD_ARG, "an asynchronous run is pending");
    if (row < 2 || row > c->d.tot) return fail(BNR_ERR_BAD_ARG, "row out of range (need 2 <= row <= tot_save)");
    HIPCHK(hipSetDevice(c->device));
    if (need_carried) { int rc = refresh_carried(c, row - 2); if (rc) return rc; }
    HIPCHK(hipStreamSynchronize(c->x.stream));
    c->plan_pin[0] = bnr_plan_entry{(uint32_t)iter, row - 1, row - 2, 0};
    return upload_plan(c, 1);
}
static int hook_end(bnr_chain *c, const char *what)
{
    int rc = check_launch(what);
    if (rc) return rc;
    c->carried_row = -1;
    return fetch_status(c);
}

int bnr_gibbs_step(bnr_chain *c, int32_t row, int64_t iter)
{
    int rc = hook_begin(c, row, iter, true);
    if (rc) return rc;
    launch_sweep(c->x, 0, false);
    launch_full_tail(c->x, 0);
    rc = hook_end(c, "gibbs_step");
    if (!rc) { c->carried_row = row - 1; c->iter = iter; }
    return rc;
}
int bnr_update_tau2(bnr_chain *c, int32_t row, int64_t iter)
{
    int rc = hook_begin(c, row, iter, true);
    if (rc) return rc;
    launch_node(c->x, 0, 1);
    return hook_end(c, "update_tau2");
}
int bnr_update_u_xi(bnr_chain *c, int32_t row, int64_t iter)
{
    int rc = hook_begin(c, row, iter, false);
    if (rc) return rc;
    launch_node(c->x, 0, 2 | 4);
    return hook_end(c, "update_u_xi");
}
int bnr_update_gamma(bnr_chain *c, int32_t row, int64_t iter)
{
    int rc = hook_begin(c, row, iter, false);
    if (rc) return rc;
    launch_node(c->x, 0, 0);            // publishes tau = sqrt(tau2[row])
    launch_xpass(c->x, 0, 3);
    launch_gram(c->x, 0, c->x.stream, false);
    launch_rhs(c->x, 0);
    launch_chol(c->x, 0, c->x.stream);
    launch_solve(c->x);
    launch_backproj(c->x, 0, 1);
    return hook_end(c, "update_gamma");
}
int bnr_update_D(bnr_chain *c, int32_t row, int64_t iter)
{
    int rc = hook_begin(c, row, iter, false);
    if (rc) return rc;
    launch_xpass(c->x, 0, 1);
    launch_backproj(c->x, 0, 2);
    return hook_end(c, "update_D");
}
int bnr_update_theta(bnr_chain *c, int32_t row, int64_t iter)
{
    int rc = hook_begin(c, row, iter, false);
    if (rc) return rc;
    launch_xpass(c->x, 0, 1);
    launch_backproj(c->x, 0, 4);
    launch_tail(c->x, 0, 1, 0);
    return hook_end(c, "update_theta");
}
int bnr_update_Delta(bnr_chain *c, int32_t row, int64_t iter)
{
    int rc = hook_begin(c, row, iter, false);
    if (rc) return rc;
    launch_tail(c->x, 0, 2, 0);
    return hook_end(c, "update_Delta");
}
int bnr_update_M(bnr_chain *c, int32_t row, int64_t iter)
{
    int rc = hook_begin(c, row, iter, false);
    if (rc) return rc;
    launch_tail(c->x, 0, 4, 0);
    return hook_end(c, "update_M");
}
int bnr_update_mu(bnr_chain *c, int32_t row, int64_t iter)
{
    int rc = hook_begin(c, row, iter, false);
    if (rc) return rc;
    launch_xpass(c->x, 0, 4);
    launch_tail(c->x, 0, 8, 1);
    return hook_end(c, "update_mu");
}
int bnr_update_Lambda(bnr_chain *c, int32_t row, int64_t iter)
{
    int rc = hook_begin(c, row, iter, false);
    if (rc) return rc;
    launch_xpass(c->x, 0, 1);
    launch_backproj(c->x, 0, 4);
    launch_tail(c->x, 0, 16, 0);
    return hook_end(c, "update_Lambda");
}
int bnr_update_pi(bnr_chain *c, int32_t row, int64_t iter)
{
    int rc = hook_begin(c, row, iter, false);
    if (rc) return rc;
    launch_tail(c->x, 0, 32, 0);
    return hook_end(c, "update_pi");
}

int bnr_chain_get_iter(bnr_chain *c, int64_t *iter)
{
    if (!c || !iter) return fail(BNR_ERR_BAD_ARG, "NULL argument");
    *iter = c->iter;
    return BNR_OK;
}
int bnr_chain_set_iter(bnr_chain *c, int64_t iter)
{
    if (!c) return fail(BNR_ERR_BAD_ARG, "NULL chain");
    c->iter = iter;
    return BNR_OK;
}

// ------------------------------------------------------------------------------------------ table I/O
struct col_desc { int off, ncols; };
static void col_layout(const bnr_dev &d, col_desc out[11])
{
    out[0] = {ROW_TAU2, 1}; out[1] = {d.o_u, d.R * d.V}; out[2] = {d.o_xi, d.V}; out[3] = {d.o_gamma, d.q};
    out[4] = {d.o_S, d.q}; out[5] = {ROW_THETA, 1}; out[6] = {ROW_DELTA, 1}; out[7] = {d.o_M, d.R * d.R};
    out[8] = {ROW_MU, 1}; out[9] = {d.o_lam, d.R}; out[10] = {d.o_pi, 3 * d.R};
}

static int table_io(bnr_chain *c, bool fetch, int first_row, int last_row, int host_tot, int host_off, double *cols[11])
{
    if (!c) return fail(BNR_ERR_BAD_ARG, "NULL chain");
    if (c->pending) return fail(BNR_ERR_BAD_ARG, "an asynchronous run is pending");
    const bnr_dev &d = c->d;
    if (first_row < 1 || last_row > d.tot || last_row < first_row) return fail(BNR_ERR_BAD_ARG, "row range outside the device table");
    const int nrows = last_row - first_row + 1;
    if (first_row + host_off < 1 || last_row + host_off > host_tot) return fail(BNR_ERR_BAD_ARG, "row range outside the host table");
    HIPCHK(hipSetDevice(c->device));
    col_desc cdsc[11];
    col_layout(d, cdsc);
    const size_t stage_doubles = (size_t)8 << 20;     // 64 MiB staging
    int cols_per = (int)std::max<size_t>(1, stage_doubles / (size_t)nrows);
    double *stage = nullptr;
    int widest = 1;
    for (int k = 0; k < 11; ++k) widest = std::max(widest, cdsc[k].ncols);
    HIPCHK(hipMalloc((void **)&stage, sizeof(double) * (size_t)nrows * std::min(cols_per, widest)));
    int rc = BNR_OK;
    for (int k = 0; k < 11 && !rc; ++k) {
        if (!cols[k]) continue;
        for (int c0 = 0; c0 < cdsc[k].ncols && !rc; c0 += cols_per) {
            int nc = std::min(cols_per, cdsc[k].ncols - c0);
            dim3 grid((nc + 31) / 32, (nrows + 31) / 32), block(32, 8);
            double *hbase = cols[k] + (size_t)(first_row - 1 + host_off) + (size_t)host_tot * c0;
            hipError_t e;
            if (fetch) {
                hipLaunchKernelGGL(k_fetch_cols, grid, block, 0, c->x.stream, (const double *)d.trace, d.rowlen, cdsc[k].off + c0, nc, first_row - 1, nrows, stage);
                e = hipMemcpy2DAsync(hbase, (size_t)host_tot * sizeof(double), stage, (size_t)nrows * sizeof(double),
                                     (size_t)nrows * sizeof(double), nc, hipMemcpyDeviceToHost, c->x.stream);
            } else {
                e = hipMemcpy2DAsync(stage, (size_t)nrows * sizeof(double), hbase, (size_t)host_tot * sizeof(double),
                                     (size_t)nrows * sizeof(double), nc, hipMemcpyHostToDevice, c->x.stream);
                hipLaunchKernelGGL(k_load_cols, grid, block, 0, c->x.stream, d.trace, d.rowlen, cdsc[k].off + c0, nc, first_row - 1, nrows, (const double *)stage);
            }
            if (e != hipSuccess) rc = fail(BNR_ERR_HIP, std::string("table copy: ") + hipGetErrorString(e));
            if (!rc && hipStreamSynchronize(c->x.stream) != hipSuccess) rc = fail(BNR_ERR_HIP, "table copy sync failed");
        }
    }
    (void)hipFree(stage);
    if (!fetch) c->carried_row = -1;
    return rc ? rc : check_launch("table_io");
}

int bnr_chain_fetch(bnr_chain *c, int32_t first_row, int32_t last_row, int32_t host_tot, int32_t host_row_offset,
                    double *tau2, double *u, double *xi, double *gamma, double *S, double *theta, double *Delta,
                    double *M, double *mu, double *lam, double *pi)
{
    double *cols[11] = {tau2, u, xi, gamma, S, theta, Delta, M, mu, lam, pi};
    return table_io(c, true, first_row, last_row, host_tot, host_row_offset, cols);
}
int bnr_chain_load(bnr_chain *c, int32_t first_row, int32_t last_row, int32_t host_tot, int32_t host_row_offset,
                   const double *tau2, const double *u, const double *xi, const double *gamma, const double *S,
                   const double *theta, const double *Delta, const double *M, const double *mu, const double *lam,
                   const double *pi)
{
    double *cols[11] = {(double *)tau2, (double *)u, (double *)xi, (double *)gamma, (double *)S, (double *)theta,
                        (double *)Delta, (double *)M, (double *)mu, (double *)lam, (double *)pi};
    return table_io(c, false, first_row, last_row, host_tot, host_row_offset, cols);
}

int bnr_chain_move_rows(bnr_chain *c, int32_t to_row, int32_t from_row, int32_t count)
{
    if (!c) return fail(BNR_ERR_BAD_ARG, "NULL chain");
    if (c->pending) return fail(BNR_ERR_BAD_ARG, "an asynchronous run is pending");
    const bnr_dev &d = c->d;
    if (count < 0 || to_row < 1 || from_row < 1 || to_row + count - 1 > d.tot || from_row + count - 1 > d.tot)
        return fail(BNR_ERR_BAD_ARG, "row range outside the device table");
    if (count == 0 || to_row == from_row) return BNR_OK;
    HIPCHK(hipSetDevice(c->device));
    const size_t rb = (size_t)d.rowlen * sizeof(double);
    // the reference copies row by row in increasing i (gibbs.jl:991-993): emulate exactly, also when ranges overlap
    for (int i = 0; i < count; ++i)
        HIPCHK(hipMemcpyAsync(d.trace + (size_t)(to_row - 1 + i) * d.rowlen, d.trace + (size_t)(from_row - 1 + i) * d.rowlen, rb,
                              hipMemcpyDeviceToDevice, c->x.stream));
    HIPCHK(hipStreamSynchronize(c->x.stream));
    c->carried_row = -1;
    return BNR_OK;
}

int bnr_chain_resize(bnr_chain *c, int32_t new_tot)
{
    if (!c) return fail(BNR_ERR_BAD_ARG, "NULL chain");
    if (c->pending) return fail(BNR_ERR_BAD_ARG, "an asynchronous run is pending");
    if (new_tot < 2) return fail(BNR_ERR_BAD_ARG, "new_tot must be >= 2");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->x.stream));
    bnr_dev &d = c->d;
    double *old = d.trace, *nt = nullptr;
    int rc = alloc_trace(c, new_tot, &nt);
    if (rc) return rc;
    size_t keep = (size_t)std::min(new_tot, d.tot) * d.rowlen * sizeof(double);   // the hidden scratch row holds nothing between calls
    HIPCHK(hipMemcpyAsync(nt, old, keep, hipMemcpyDeviceToDevice, c->x.stream));   // behind the zero fill of the new table, on the same stream
    HIPCHK(hipStreamSynchronize(c->x.stream));
    (void)hipFree(old);
    d.trace = nt; d.tot = new_tot;
    drop_graph(c->x);
    if (c->group && c->group->x.nb == 1) drop_graph(c->group->x);   // a one-member group bakes the member's descriptor into its graphs as well
    return sync_dev(c);
}

// ------------------------------------------------------------------------------------------ Rhat
int bnr_chain_rhat_stats(bnr_chain *c, int32_t first_row, int32_t nsamp, double *stats)
{
    if (!c || !stats) return fail(BNR_ERR_BAD_ARG, "NULL argument");
    if (c->pending) return fail(BNR_ERR_BAD_ARG, "an asynchronous run is pending");
    const bnr_dev &d = c->d;
    if (first_row < 1 || nsamp < 4 || first_row + nsamp - 1 > d.tot) return fail(BNR_ERR_BAD_ARG, "row window outside the table or nsamp < 4");
    HIPCHK(hipSetDevice(c->device));
    const int np = d.q + d.V;
    double *out = nullptr;
    HIPCHK(hipMalloc((void **)&out, sizeof(double) * 4 * np));
    hipLaunchKernelGGL(k_rhat_stats, dim3((np + 127) / 128), dim3(128), 0, c->x.stream, c->d, first_row - 1, nsamp, out);
    hipError_t e = hipMemcpyAsync(stats, out, sizeof(double) * 4 * np, hipMemcpyDeviceToHost, c->x.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->x.stream);
    (void)hipFree(out);
    if (e != hipSuccess) return fail(BNR_ERR_HIP, std::string("rhat_stats: ") + hipGetErrorString(e));
    return check_launch("k_rhat_stats");
}

// ------------------------------------------------------------------------------------------ convergence check across ranks
// bnr_comm: who takes part in the one exchange step of a fit -- the all-gather of the per-chain split-Rhat messages
// (the reference returns whole state tables from its pmap workers to the master, gibbs.jl:946-957).  Two kinds:
//   RCCL      ncclAllGather over xGMI on the library's own communicator; librccl is bound at run time with dlopen, so a
//             single-GPU user needs no RCCL at all.  The 128-byte unique id is created on rank 0 (bnr_comm_unique_id) and
//             carried to the other ranks by whatever already connects them (Julia Distributed, torch.distributed's store).
//   callback  the host supplies the all-gather (tests over gloo; an MPI host; Julia remotecall).
struct bnr_comm {
    int rank = 0, world = 1, device = 0;
    void *nccl = nullptr;                     // ncclComm_t
    hipStream_t stream = nullptr;
    bnr_allgather_fn fn = nullptr;
    void *ctx = nullptr;
    double *dsend = nullptr, *drecv = nullptr;   // RCCL: persistent device staging (send block, gathered block), grown on demand
    size_t dcap = 0;                            // doubles per rank they hold
};
// device staging of an RCCL communicator for `count` doubles per rank
static int comm_reserve(bnr_comm *c, size_t count)
{
    if (count <= c->dcap) return BNR_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->dsend) (void)hipFree(c->dsend);
    if (c->drecv) (void)hipFree(c->drecv);
    c->dsend = c->drecv = nullptr; c->dcap = 0;
    const size_t cap = std::max<size_t>(count, 1024);
    HIPCHK(hipMalloc((void **)&c->dsend, sizeof(double) * cap));
    HIPCHK(hipMalloc((void **)&c->drecv, sizeof(double) * cap * (size_t)c->world));
    c->dcap = cap;
    return BNR_OK;
}
namespace {
struct rccl_api {
    void *dl = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, bnr_unique_id, int) = nullptr;     // ncclUniqueId is a 128-byte struct passed by value
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*CommCount)(void *, int *) = nullptr;
    int (*CommUserRank)(void *, int *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
rccl_api g_rccl;
std::mutex g_rccl_mu;
}
static int rccl_load()
{
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.dl) return BNR_OK;
    void *h = nullptr;
    // RCCL must sit on the SAME HIP runtime this library is bound to: a process can hold two (a Python host that imports a
    // torch wheel with its own libamdhip64 + librccl next to /opt/rocm's), and a communicator created through the other
    // runtime fails with "unhandled cuda error".  So: look next to the libamdhip64 that hipMalloc resolves to first.
    // RTLD_DEEPBIND: a second RCCL copy must not resolve its own internal calls into the copy loaded before it.
    std::string tried;
    Dl_info info;
    if (dladdr((void *)(hipError_t (*)(void **, size_t))&hipMalloc, &info) && info.dli_fname) {
        std::string dir(info.dli_fname);
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) {
            dir.resize(slash + 1);
            for (const char *nm : {"librccl.so.1", "librccl.so"}) {
                h = dlopen((dir + nm).c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
                if (h) break;
                tried += dir + nm + " ";
            }
        }
    }
    if (!h) for (const char *nm : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { h = dlopen(nm, RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND); if (h) break; tried += std::string(nm) + " "; }
    if (!h) return fail(BNR_ERR_HIP, "librccl.so not found (tried " + tried + ")");
    rccl_api a;
    a.dl = h;
    a.GetUniqueId = (int (*)(void *))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (int (*)(void **, int, bnr_unique_id, int))dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (int (*)(void *))dlsym(h, "ncclCommDestroy");
    a.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(h, "ncclAllGather");
    a.CommCount = (int (*)(void *, int *))dlsym(h, "ncclCommCount");
    a.CommUserRank = (int (*)(void *, int *))dlsym(h, "ncclCommUserRank");
    a.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather) return fail(BNR_ERR_HIP, "librccl.so lacks an expected symbol");
    g_rccl = a;
    return BNR_OK;
}
static int rccl_fail(const char *what, int code)
{
    return fail(BNR_ERR_HIP, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(code) : "RCCL error") + " (" + std::to_string(code) + ")");
}

int bnr_comm_unique_id(bnr_unique_id *id)
{
    if (!id) return fail(BNR_ERR_BAD_ARG, "NULL id");
    int rc = rccl_load();
    if (rc) return rc;
    int e = g_rccl.GetUniqueId(id);
    return e ? rccl_fail("ncclGetUniqueId", e) : BNR_OK;
}
int bnr_comm_create_rccl(const bnr_unique_id *id, int32_t rank, int32_t world, int32_t device, bnr_comm **out)
{
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return fail(BNR_ERR_BAD_ARG, "bad communicator arguments");
    int rc = rccl_load();
    if (rc) return rc;
    HIPCHK(hipSetDevice(device));
    bnr_comm *c = new bnr_comm();
    c->rank = rank; c->world = world; c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return fail(BNR_ERR_HIP, "hipStreamCreate failed"); }
    int e = g_rccl.CommInitRank(&c->nccl, world, *id, rank);
    if (e) { (void)hipStreamDestroy(c->stream); delete c; return rccl_fail("ncclCommInitRank", e); }
    *out = c;
    return BNR_OK;
}
int bnr_comm_create_callback(int32_t rank, int32_t world, bnr_allgather_fn fn, void *ctx, bnr_comm **out)
{
    if (!fn || !out || world < 1 || rank < 0 || rank >= world) return fail(BNR_ERR_BAD_ARG, "bad communicator arguments");
    bnr_comm *c = new bnr_comm();
    c->rank = rank; c->world = world; c->fn = fn; c->ctx = ctx;
    *out = c;
    return BNR_OK;
}
int bnr_comm_destroy(bnr_comm *c)
{
    if (!c) return BNR_OK;
    if (c->nccl) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); g_rccl.CommDestroy(c->nccl); }
    if (c->dsend) (void)hipFree(c->dsend);
    if (c->drecv) (void)hipFree(c->drecv);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return BNR_OK;
}
// what a communicator is, as the transport itself reports it: kind 1 = RCCL (rccl_ranks / rccl_rank = ncclCommCount / ncclCommUserRank of
// the library's communicator, i.e. how many ranks RCCL really connected -- not the number the caller asked for), kind 2 = host callback
// (rccl_ranks = 0).  NULL communicator: one rank, no transport (kind 0).
int bnr_comm_info(bnr_comm *c, int32_t *kind, int32_t *rank, int32_t *world, int32_t *rccl_ranks, int32_t *rccl_rank)
{
    int k = 0, r = 0, w = 1, nr = 0, ur = -1;
    if (c) {
        r = c->rank; w = c->world;
        if (c->fn) k = 2;
        else if (c->nccl) {
            k = 1;
            if (g_rccl.CommCount) { int e = g_rccl.CommCount(c->nccl, &nr); if (e) return rccl_fail("ncclCommCount", e); }
            if (g_rccl.CommUserRank) { int e = g_rccl.CommUserRank(c->nccl, &ur); if (e) return rccl_fail("ncclCommUserRank", e); }
        }
    }
    if (kind) *kind = k;
    if (rank) *rank = r;
    if (world) *world = w;
    if (rccl_ranks) *rccl_ranks = nr;
    if (rccl_rank) *rccl_rank = ur;
    return BNR_OK;
}
// all-gather of `count` doubles per rank: send (host) -> recv (host, world * count, rank order)
int bnr_comm_allgather(bnr_comm *c, const double *send, double *recv, int64_t count)
{
    if (!send || !recv || count < 0) return fail(BNR_ERR_BAD_ARG, "bad all-gather arguments");
    if (!c) { memcpy(recv, send, sizeof(double) * (size_t)count); return BNR_OK; }       // no communicator: one rank
    if (c->fn) {
        int e = c->fn(c->ctx, send, recv, count);
        return e ? fail(BNR_ERR_HIP, "the host's all-gather callback reported " + std::to_string(e)) : BNR_OK;
    }
    // RCCL, also with ONE rank (the collective is then a device copy inside RCCL, but it is the same call path as with eight)
    HIPCHK(hipSetDevice(c->device));
    int rc = comm_reserve(c, (size_t)count);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(c->dsend, send, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, c->stream));
    int e = g_rccl.AllGather(c->dsend, c->drecv, (size_t)count, 8 /* ncclFloat64 */, c->nccl, c->stream);
    if (e) return rccl_fail("ncclAllGather", e);
    HIPCHK(hipMemcpyAsync(recv, c->drecv, sizeof(double) * (size_t)count * c->world, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return BNR_OK;
}
int bnr_rhat(bnr_chain *const *chains, int32_t nchains_local, int32_t nchains_total, bnr_comm *comm, int32_t burn, int32_t nsamp,
             double *rhat_xi, double *rhat_gamma)
{
    if (nchains_total < 1 || nchains_local < 0 || (nchains_local > 0 && !chains) || !rhat_xi || !rhat_gamma) return fail(BNR_ERR_BAD_ARG, "bad argument");
    const int world = comm ? comm->world : 1, rank = comm ? comm->rank : 0;
    int expect = 0;
    for (int c = 1; c <= nchains_total; ++c) if ((c - 1) % world == rank) ++expect;
    if (expect != nchains_local) return fail(BNR_ERR_BAD_ARG, "this rank must hold the chains c with (c-1) % world == rank");
    const int per_rank = (nchains_total + world - 1) / world;
    // every rank needs q and V also when it holds no chain: they travel in front of the messages
    int q = 0, V = 0;
    if (nchains_local > 0) { if (!chains[0]) return fail(BNR_ERR_BAD_ARG, "NULL chain"); q = chains[0]->d.q; V = chains[0]->d.V; }
    {
        std::vector<double> mine{(double)q, (double)V}, all(2 * (size_t)world);
        int rc = bnr_comm_allgather(comm, mine.data(), all.data(), 2);
        if (rc) return rc;
        for (int r = 0; r < world; ++r) if (all[2 * r] > 0) { q = (int)all[2 * r]; V = (int)all[2 * r + 1]; }
    }
    if (q <= 0 || V <= 0) return fail(BNR_ERR_BAD_ARG, "no rank holds a chain");
    const int np = q + V;
    const size_t width = (size_t)4 * np;
    std::vector<double> recv(width * per_rank * world);
    for (int i = 0; i < nchains_local; ++i)
        if (!chains[i] || chains[i]->d.q != q || chains[i]->d.V != V) return fail(BNR_ERR_BAD_ARG, "chains of one fit must have equal V");
    if (comm && comm->nccl) {
        // RCCL: the per-chain messages never leave the device before the collective -- k_rhat_stats writes each chain's block straight
        // into the communicator's send buffer, ncclAllGather moves them GPU to GPU, ONE copy brings the gathered block to the host
        HIPCHK(hipSetDevice(comm->device));
        int rc = comm_reserve(comm, width * per_rank);
        if (rc) return rc;
        HIPCHK(hipMemsetAsync(comm->dsend, 0, sizeof(double) * width * per_rank, comm->stream));
        HIPCHK(hipStreamSynchronize(comm->stream));
        for (int i = 0; i < nchains_local; ++i) {
            bnr_chain *c = chains[i];
            if (c->pending) return fail(BNR_ERR_BAD_ARG, "an asynchronous run is pending");
            if (c->device != comm->device) return fail(BNR_ERR_BAD_ARG, "the chains of this rank must live on the communicator's device");
            if (nsamp < 4 || burn + nsamp > c->d.tot || burn < 0) return fail(BNR_ERR_BAD_ARG, "row window outside the table or nsamp < 4");
            hipLaunchKernelGGL(k_rhat_stats, dim3((np + 127) / 128), dim3(128), 0, c->x.stream, c->d, burn, nsamp, comm->dsend + width * i);
            HIPCHK(hipStreamSynchronize(c->x.stream));
        }
        rc = check_launch("k_rhat_stats");
        if (rc) return rc;
        int e = g_rccl.AllGather(comm->dsend, comm->drecv, width * per_rank, 8 /* ncclFloat64 */, comm->nccl, comm->stream);
        if (e) return rccl_fail("ncclAllGather", e);
        HIPCHK(hipMemcpyAsync(recv.data(), comm->drecv, sizeof(double) * recv.size(), hipMemcpyDeviceToHost, comm->stream));
        HIPCHK(hipStreamSynchronize(comm->stream));
    } else {
        std::vector<double> send(width * per_rank, 0.0);
        for (int i = 0; i < nchains_local; ++i) {
            int rc = bnr_chain_rhat_stats(chains[i], burn + 1, nsamp, send.data() + width * i);
            if (rc) return rc;
        }
        int rc = bnr_comm_allgather(comm, send.data(), recv.data(), (int64_t)send.size());
        if (rc) return rc;
    }
    int rc;
    std::vector<double> stats(width * nchains_total), rh(np);
    for (int c = 1; c <= nchains_total; ++c)
        memcpy(stats.data() + width * (c - 1), recv.data() + width * ((size_t)((c - 1) % world) * per_rank + (c - 1) / world), sizeof(double) * width);
    rc = bnr_rhat_from_stats(stats.data(), nchains_total, np, nsamp, rh.data());
    if (rc) return rc;
    memcpy(rhat_gamma, rh.data(), sizeof(double) * q);
    memcpy(rhat_xi, rh.data() + q, sizeof(double) * V);
    return BNR_OK;
}

// Summary(results) on the device (gibbs.jl:1214-1250): posterior mean and two order statistics of every gamma_e over rows
// first_row .. first_row+nsamp-1, and the mean of every xi_v.  3q + V doubles cross PCIe instead of the gamma trace.
int bnr_chain_summary(bnr_chain *c, int32_t first_row, int32_t nsamp, int32_t k_lo, int32_t k_hi,
                      double *mean_gamma, double *lower, double *upper, double *prob_xi)
{
    if (!c || !mean_gamma || !lower || !upper || !prob_xi) return fail(BNR_ERR_BAD_ARG, "NULL argument");
    if (c->pending) return fail(BNR_ERR_BAD_ARG, "an asynchronous run is pending");
    const bnr_dev &d = c->d;
    if (first_row < 1 || nsamp < 1 || first_row + nsamp - 1 > d.tot) return fail(BNR_ERR_BAD_ARG, "row window outside the table");
    if (k_lo < 1 || k_lo > nsamp || k_hi < 1 || k_hi > nsamp) return fail(BNR_ERR_BAD_ARG, "order statistics must be between 1 and nsamp");
    HIPCHK(hipSetDevice(c->device));
    const int np = d.q + d.V;
    double *buf = nullptr, *out = nullptr;
    HIPCHK(hipMalloc((void **)&buf, sizeof(double) * (size_t)np * nsamp));
    if (hipMalloc((void **)&out, sizeof(double) * 3 * (size_t)np) != hipSuccess) { (void)hipFree(buf); return fail(BNR_ERR_HIP, "hipMalloc failed"); }
    dim3 block(32, 8);
    hipLaunchKernelGGL(k_fetch_cols, dim3((d.q + 31) / 32, (nsamp + 31) / 32), block, 0, c->x.stream, (const double *)d.trace, d.rowlen, d.o_gamma, d.q, first_row - 1, nsamp, buf);
    hipLaunchKernelGGL(k_fetch_cols, dim3((d.V + 31) / 32, (nsamp + 31) / 32), block, 0, c->x.stream, (const double *)d.trace, d.rowlen, d.o_xi, d.V, first_row - 1, nsamp, buf + (size_t)d.q * nsamp);
    hipLaunchKernelGGL(k_summary, dim3(np), dim3(256), 0, c->x.stream, (const double *)buf, nsamp, d.q, k_lo, k_hi, out, out + np, out + 2 * (size_t)np);
    std::vector<double> host(3 * (size_t)np);
    hipError_t e = hipMemcpyAsync(host.data(), out, sizeof(double) * host.size(), hipMemcpyDeviceToHost, c->x.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->x.stream);
    (void)hipFree(buf); (void)hipFree(out);
    if (e != hipSuccess) return fail(BNR_ERR_HIP, std::string("summary: ") + hipGetErrorString(e));
    memcpy(mean_gamma, host.data(), sizeof(double) * d.q);
    memcpy(prob_xi, host.data() + d.q, sizeof(double) * d.V);
    memcpy(lower, host.data() + np, sizeof(double) * d.q);
    memcpy(upper, host.data() + 2 * (size_t)np, sizeof(double) * d.q);
    return check_launch("k_summary");
}

// Effective sample size (an addition: the reference only has split-Rhat).  Per-chain message: for both halves of the window
// the mean, variance and the autocovariances at lags 0..max_lag-1 of gamma (q) and xi (V): 2 (2 + max_lag) (q + V) doubles.
int bnr_chain_ess_stats(bnr_chain *c, int32_t first_row, int32_t nsamp, int32_t max_lag, double *stats)
{
    if (!c || !stats) return fail(BNR_ERR_BAD_ARG, "NULL argument");
    if (c->pending) return fail(BNR_ERR_BAD_ARG, "an asynchronous run is pending");
    const bnr_dev &d = c->d;
    if (first_row < 1 || nsamp < 8 || first_row + nsamp - 1 > d.tot) return fail(BNR_ERR_BAD_ARG, "row window outside the table or nsamp < 8");
    if (max_lag < 2 || max_lag > nsamp / 2) return fail(BNR_ERR_BAD_ARG, "need 2 <= max_lag <= nsamp/2");
    HIPCHK(hipSetDevice(c->device));
    const int np = d.q + d.V;
    const size_t width = (size_t)2 * (2 + max_lag) * np;
    double *buf = nullptr, *out = nullptr;
    HIPCHK(hipMalloc((void **)&buf, sizeof(double) * (size_t)np * nsamp));
    if (hipMalloc((void **)&out, sizeof(double) * width) != hipSuccess) { (void)hipFree(buf); return fail(BNR_ERR_HIP, "hipMalloc failed"); }
    dim3 block(32, 8);
    hipLaunchKernelGGL(k_fetch_cols, dim3((d.q + 31) / 32, (nsamp + 31) / 32), block, 0, c->x.stream, (const double *)d.trace, d.rowlen, d.o_gamma, d.q, first_row - 1, nsamp, buf);
    hipLaunchKernelGGL(k_fetch_cols, dim3((d.V + 31) / 32, (nsamp + 31) / 32), block, 0, c->x.stream, (const double *)d.trace, d.rowlen, d.o_xi, d.V, first_row - 1, nsamp, buf + (size_t)d.q * nsamp);
    hipLaunchKernelGGL(k_acov, dim3(np, 2), dim3(256), 0, c->x.stream, (const double *)buf, nsamp, np, max_lag, out);
    hipError_t e = hipMemcpyAsync(stats, out, sizeof(double) * width, hipMemcpyDeviceToHost, c->x.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->x.stream);
    (void)hipFree(buf); (void)hipFree(out);
    if (e != hipSuccess) return fail(BNR_ERR_HIP, std::string("ess_stats: ") + hipGetErrorString(e));
    return check_launch("k_acov");
}

// Bulk effective sample size over all chains from the gathered messages (the estimator of Vehtari et al. 2021 as in
// Stan / MCMCDiagnosticTools.ess: split chains, rho_t = 1 - (W - mean acov_t) / var+, Geyer's initial positive and monotone
// sequence on the pair sums), truncated at max_lag.  NaN for a constant parameter.
int bnr_ess_from_stats(const double *stats, int32_t nchains, int32_t nparams, int32_t nsamp, int32_t max_lag, double *ess)
{
    if (!stats || !ess || nchains < 1 || nparams < 1 || max_lag < 2) return fail(BNR_ERR_BAD_ARG, "bad argument");
    const int h = nsamp / 2, m = 2 * nchains, L = max_lag;
    const size_t hw = (size_t)(2 + L) * nparams, cw = 2 * hw;          // per half, per chain
    std::vector<double> rho(L);
    for (int p = 0; p < nparams; ++p) {
        double W = 0.0, mm = 0.0;
        for (int c = 0; c < nchains; ++c) for (int k = 0; k < 2; ++k) {
            const double *s = stats + (size_t)c * cw + (size_t)k * hw;
            mm += s[p]; W += s[(size_t)nparams + p];
        }
        mm /= m; W /= m;
        double B = 0.0;
        for (int c = 0; c < nchains; ++c) for (int k = 0; k < 2; ++k) {
            const double *s = stats + (size_t)c * cw + (size_t)k * hw;
            B += (s[p] - mm) * (s[p] - mm);
        }
        B /= (m - 1);
        const double varp = W * (h - 1.0) / h + B;
        if (!(varp > 0.0) || !(W > 0.0)) { ess[p] = NAN; continue; }
        for (int t = 0; t < L; ++t) {
            double ac = 0.0;
            for (int c = 0; c < nchains; ++c) for (int k = 0; k < 2; ++k)
                ac += stats[(size_t)c * cw + (size_t)k * hw + (size_t)(2 + t) * nparams + p];
            ac /= m;
            rho[t] = 1.0 - (W - ac) / varp;              // W: 1/(h-1) variances, acov_t: 1/h sums, as in Stan
        }
        rho[0] = 1.0;
        // Geyer: pair sums P_t = rho_2t + rho_2t+1 while positive, made monotone; then the next even term if positive
        double tau = -1.0, prev = INFINITY;
        int t = 0;
        for (; 2 * t + 1 < L; ++t) {
            double P = rho[2 * t] + rho[2 * t + 1];
            if (!(P > 0.0)) break;
            if (P > prev) P = prev;
            prev = P;
            tau += 2.0 * P;
        }
        if (2 * t < L && rho[2 * t] > 0.0) tau += rho[2 * t];
        if (tau < 1.0 / log10((double)m * h)) tau = 1.0 / log10((double)m * h);      // Stan's cap on anti-correlated chains
        ess[p] = (double)m * h / tau;
    }
    return BNR_OK;
}

int bnr_rhat_from_stats(const double *stats, int32_t nchains, int32_t nparams, int32_t nsamp, double *rhat)
{
    if (!stats || !rhat || nchains < 1 || nparams < 1) return fail(BNR_ERR_BAD_ARG, "bad argument");
    const int h = nsamp / 2, m = 2 * nchains;
    if (h - 1 <= 0) { for (int p = 0; p < nparams; ++p) rhat[p] = NAN; return BNR_OK; }
    const double cf = (double)(h - 1) / h;                       // convergence.jl:25
    for (int p = 0; p < nparams; ++p) {
        double W = 0.0, mm = 0.0;
        for (int c = 0; c < nchains; ++c) {
            const double *s = stats + (size_t)c * 4 * nparams;
            mm += s[p] + s[(size_t)2 * nparams + p];
            W += s[(size_t)nparams + p] + s[(size_t)3 * nparams + p];
        }
        W /= m; mm /= m;                                         // :49
        double B = 0.0;
        for (int c = 0; c < nchains; ++c) {
            const double *s = stats + (size_t)c * 4 * nparams;
            double d0 = s[p] - mm, d1 = s[(size_t)2 * nparams + p] - mm;
            B += d0 * d0 + d1 * d1;
        }
        B /= (m - 1);
        double varp = cf * W + B;                                // :52
        if (varp == 0.0 && W == 0.0) rhat[p] = 1.0;              // :55-61
        else if (W == 0.0) rhat[p] = INFINITY;
        else rhat[p] = sqrt(varp / W);
    }
    return BNR_OK;
}

int bnr_chain_counters(bnr_chain *c, int64_t out[8])
{
    if (!c || !out) return fail(BNR_ERR_BAD_ARG, "NULL argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(c->counters_host, c->d.counters, sizeof(long long) * 16, hipMemcpyDeviceToHost, c->x.stream));
    HIPCHK(hipStreamSynchronize(c->x.stream));
    for (int i = 0; i < 8; ++i) out[i] = c->counters_host[i];   // [4..7]: where a hard Cholesky failure happened (node, Psi, M, G+I)
    return BNR_OK;
}

int bnr_chain_debug_read(bnr_chain *c, uint64_t *out, int32_t count)
{
    if (!c || !out || count < 0 || count > 4096) return fail(BNR_ERR_BAD_ARG, "bad argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpy(out, c->d.dbg, sizeof(uint64_t) * count, hipMemcpyDeviceToHost));
    return BNR_OK;
}

int bnr_chain_debug_copy(bnr_chain *c, int32_t which, double *out, int64_t count)
{
    if (!c || !out) return fail(BNR_ERR_BAD_ARG, "bad argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->x.stream));
    const double *src = which == 0 ? c->d.E : (which == 1 ? c->d.bw : (which == 2 ? c->d.a4 : c->d.Gpart));
    HIPCHK(hipMemcpy(out, src, sizeof(double) * count, hipMemcpyDeviceToHost));
    return BNR_OK;
}

int bnr_chain_debug_dims(bnr_chain *c, int32_t *out8)
{
    if (!c || !out8) return fail(BNR_ERR_BAD_ARG, "bad argument");
    const bnr_dev &d = c->d;
    const int v[8] = {d.n_pad, d.q_pad, d.ksplit, d.ntile, d.kcp, d.kslab, d.i8L, d.rowlen};
    for (int i = 0; i < 8; ++i) out8[i] = v[i];
    return BNR_OK;
}

static void launch_gram_only(bnr_chain *c)
{
    const bnr_dev &d = c->d;
    const int ntl = d.ntile * (d.ntile + 1) / 2;
    if (gram_on_i8(c->x) && c->x.gram_variant == 0) {
        const dim3 ggrid(round_up(ntl * d.ksplit, 8));
        launch_gram_i8(c->x, 0, c->x.stream, ggrid);
        return;
    }
    if (c->x.gram_variant == 8) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gram8<bnr_one>), dim3(round_up(ntl * d.ksplit, 8)), dim3(512), 0, c->x.stream, bnr_one{d}, 0, 1);
    else if (d.gram_kg == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gram<bnr_one, 4>), dim3(round_up(ntl * d.ksplit, 8)), dim3(1024), 0, c->x.stream, bnr_one{d}, 0, 1);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gram<bnr_one, 2>), dim3(round_up(ntl * d.ksplit, 8)), dim3(512), 0, c->x.stream, bnr_one{d}, 0, 1);
}
int bnr_chain_debug_time_gram(bnr_chain *c, int32_t reps, double *avg_us)
{
    if (!c || !avg_us || reps < 1 || reps > 100000) return fail(BNR_ERR_BAD_ARG, "bad argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->x.stream));
    c->plan_pin[0] = bnr_plan_entry{1u, 1, 0, 0};
    int rc = upload_plan(c, 1);
    if (rc) return rc;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) launch_gram_only(c);
    HIPNOTE(hipEventRecord(e0, c->x.stream));
    for (int r = 0; r < reps; ++r) launch_gram_only(c);
    HIPNOTE(hipEventRecord(e1, c->x.stream));
    HIPCHK(hipMemsetAsync(c->d.gprog, 0, sizeof(unsigned int) * (c->d.ntile + 1), c->x.stream));   // these launches were not consumed by a factorization
    HIPCHK(hipStreamSynchronize(c->x.stream));
    float ms = 0;
    HIPNOTE(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *avg_us = 1e3 * ms / reps;
    c->carried_row = -1;
    return check_launch("debug_time_gram");
}

// timing experiments of round 4 (skip the scalar branch / the late panel steps): no longer part of the library, the entry point refuses
int bnr_debug_set_exp(int32_t device, int32_t flags)
{
    (void)device; (void)flags;
    return fail(BNR_ERR_BAD_ARG, "the timing experiments of round 4 are no longer part of the library (tools/experiments/README.md)");
}

int bnr_chain_set_profiling(bnr_chain *c, int32_t enable)
{
    if (!c) return fail(BNR_ERR_BAD_ARG, "NULL chain");
    return exec_set_option(c->x, "profiling", enable);
}
int bnr_chain_last_timing(bnr_chain *c, int32_t which, double *avg_us, int64_t *launches)
{
    if (!c || !avg_us) return fail(BNR_ERR_BAD_ARG, "NULL argument");
    return exec_last_timing(c->x, which, avg_us, launches);
}

int bnr_chain_set_option(bnr_chain *c, const char *name, int64_t value)
{
    if (!c || !name) return fail(BNR_ERR_BAD_ARG, "NULL argument");
    if (!strcmp(name, "gram_i8")) {
        // 1 (the default where the model matrix is binary): the Gram on the i8 matrix pipe; 0: the f64 Gram also for a binary matrix.  A matrix that
        // is not binary has no i8 path: switching it on is refused.
        if (c->pending) return fail(BNR_ERR_BAD_ARG, "an asynchronous run is pending");
        if (value && !c->xm_kept) return fail(BNR_ERR_BAD_ARG, "gram_i8 needs a binary (0 / 1) integer-typed model matrix");
        c->d.XM = value ? c->xm_kept : nullptr;
        drop_graph(c->x);
        if (c->group) drop_graph(c->group->x);
        return sync_dev(c);
    }
    if (!strcmp(name, "byte_x")) {
        // 0: the X passes read the f64 matrix also when a byte image exists; 1: back to the byte image (if the input had one)
        if (c->pending) return fail(BNR_ERR_BAD_ARG, "an asynchronous run is pending");
        c->d.X8 = value ? c->x8_kept : nullptr;
        drop_graph(c->x);
        if (c->group) drop_graph(c->group->x);
        return sync_dev(c);
    }
    return exec_set_option(c->x, name, value);
}

}  // extern "C"

// ----------------------------------------------------------------------------------------- every reference to the kernels added late in round 5 (see ensure_lds_attributes)
extern "C++" {
static int late_kernels_lds_attributes(int bytes)
{
    const void *late[] = {(const void *)&k_xpass_group2<0>, (const void *)&k_backproj64<bnr_one>, (const void *)&k_backproj64<bnr_many>};
    for (const void *f : late) HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    // k_tail outside a sweep (hooks, a loaded row): u (up to BNR_TAIL_U_LDS doubles) and the R x R work matrices of update_M! side by side
    const void *tails[] = {(const void *)&k_tail<bnr_one>, (const void *)&k_tail<bnr_many>, (const void *)&k_tail<bnr_one, false>, (const void *)&k_tail<bnr_many, false>};
    for (const void *f : tails) HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
    return BNR_OK;
}
static void launch_late_xpass_group2(bnr_exec &x, int s)
{ hipLaunchKernelGGL(HIP_KERNEL_NAME(k_xpass_group2<0>), dim3(x.shape->nblk_x * ((x.shape->n_pad + 255) / 256)), dim3(256), 16 * x.shape->chunk_x * sizeof(double), x.stream, bnr_many{x.cds}, s, x.nb); }
static void launch_late_backproj64(bnr_exec &x, int s, int flags, size_t lds64)
{ BNR_LAUNCH(k_backproj64, dim3(round_up((x.shape->nblk_bp + 1) / 2, 8) * x.nb), dim3(256), lds64, x.stream, x, s, flags, x.nb); }
}
