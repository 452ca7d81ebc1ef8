// bnr_experiments.h -- kernels of MEASURED EXPERIMENTS (profiles/round3_experiments_notes.txt, profiles/round4_experiments_notes.txt): variants of
// the Gram (LDS-DMA staging, resident / persistent kernels with static loops, task queues and reserved compute units), the left-looking
// factorization with its progress gates, the group back-projection, the "linear" schedule's gate kernels.  All give bitwise the tables of
// the default path and none is faster on MI355X; part of them POLL device memory.  They are compiled only with -DBNR_EXPERIMENTS (the shipped
// libbnr_hip.so has none of them: no user-reachable option can start a polling kernel); tools/r4_build_variants.sh builds the variant library
// the tools/ scripts measure.  Included at the end of bnr_kernels.h.
#pragma once

// k_gram8d: k_gram8 with the UNSCALED panel (the j side) moved global -> LDS by the DMA path (global_load_lds_dwordx4: no vector registers, no
// ds_write; the scaled i side still travels through registers, it is multiplied by S on the way).  The DMA lands in lane order, so lane l of
// K-group wave w asks for the rows that belong at image position (column 2 w + l / 32, row pair l % 32) -- the XOR swizzle of the odd columns is
// applied to the GLOBAL row.  Three j buffers (the DMA of batch b + 2 is issued while batch b is multiplied), two i buffers as before:
// 40 KiB of LDS, three workgroups per CU.  Bitwise the partial tiles of k_gram8 (same data, same order).  gram_variant 10 (experiment).
template <bool WT>
__device__ __forceinline__ void bnr_gram8_task_dma(const bnr_gram_geom &cd, const double *Sp, double *Gpart, int t, int ti, int tj, int ks, double *sred)
{
    constexpr int KG = 2, KB = 8;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kg = wave >> 2, wi = (wave >> 1) & 1, wj = wave & 1, wq = wave & 3;
    const int kchunk = cd.q_pad / cd.ksplit, ksub = kchunk / KG, eb = ks * kchunk + kg * ksub, nbatch = ksub / KB;
    const size_t ld = cd.n_pad;
    const int li = lane & 15, lk = lane >> 4;
    const int tg = tid & 255, c = tg >> 5, rp = tg & 31;
    const unsigned offI = (unsigned)(ti * BNR_GT + 2 * rp) + (unsigned)c * (unsigned)ld;
    const unsigned offJ = (unsigned)(tj * BNR_GT + ((2 * rp) ^ ((c & 1) << 4))) + (unsigned)c * (unsigned)ld;     // the row that belongs at this lane's image position
    const double *xb = cd.X + (size_t)eb * ld;
    const double *sb = Sp + eb;
    const int smax = cd.q - 1 - eb;
    constexpr int PANEL = KB * BNR_GT;
    double *stg = sred + (size_t)kg * (5 * PANEL);     // [I buf 0][I buf 1][J buf 0][J buf 1][J buf 2]
    const int woff = c * BNR_GT + ((2 * rp) ^ ((c & 1) << 4));
    const int sw = (lk & 1) << 4;
    const int ra0 = (wj * 32 + li) ^ sw, ra1 = (wj * 32 + 16 + li) ^ sw, rb0 = (wi * 32 + li) ^ sw, rb1 = (wi * 32 + 16 + li) ^ sw;
    bnr_d4 c00 = {0, 0, 0, 0}, c01 = {0, 0, 0, 0}, c10 = {0, 0, 0, 0}, c11 = {0, 0, 0, 0};
    bnr_d2 ri;
    double sv;
    // this wave's 1 KiB of a J buffer: columns 2 wq and 2 wq + 1
    const unsigned jlds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(__attribute__((address_space(3))) void *)(stg + 2 * PANEL + (size_t)(2 * wq) * BNR_GT));
#define BNR_G8D_LOAD(BIDX, JB)                                                                                   \
    do {                                                                                                          \
        const double *cb_ = xb + (size_t)(BIDX) * (KB * ld);                                                      \
        const int si_ = (BIDX) * KB + c;                                                                          \
        sv = sb[si_ < smax ? si_ : smax];                                                                         \
        ri = *(const bnr_d2 *)(cb_ + offI);                                                                       \
        /* as inline assembly: through the builtin the compiler drains vmcnt to 0 in front of every barrier (it knows the LDS is written) */ \
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(jlds0 + (unsigned)(JB) * (unsigned)(PANEL * sizeof(double))), "v"(cb_ + offJ) : "memory"); \
    } while (0)
#define BNR_G8D_STORE(BUF) do { *(bnr_d2 *)(stg + (size_t)(BUF) * PANEL + woff) = ri * sv; } while (0)
#define BNR_G8D_COMPUTE(BUF, JB, K2)                                                          \
    do {                                                                                      \
        const double *bufI = stg + (size_t)(BUF) * PANEL, *bufJ = stg + (size_t)(2 + (JB)) * PANEL; \
        const int kk = (4 * (K2) + lk) * BNR_GT;                                              \
        double a0 = bufJ[kk + ra0], a1 = bufJ[kk + ra1];                                      \
        double b0 = bufI[kk + rb0], b1 = bufI[kk + rb1];                                      \
        c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c00, 0, 0, 0);                     \
        c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, c01, 0, 0, 0);                     \
        c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, c10, 0, 0, 0);                     \
        c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c11, 0, 0, 0);                     \
    } while (0)
    int j0 = 0, j1 = 1, j2 = 2;                        // J buffers of batch b, b + 1, b + 2
    BNR_G8D_LOAD(0, 0);
    BNR_G8D_STORE(0);
    BNR_G8D_LOAD(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int b = 0; b < nbatch; ++b) {
        BNR_G8D_COMPUTE(b & 1, j0, 0);
        BNR_G8D_STORE((b + 1) & 1);
        BNR_G8D_LOAD(b + 2, j2);
        BNR_G8D_COMPUTE(b & 1, j0, 1);
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");   // the DMA of batch b + 1 (issued one iteration ago) has landed: three VMEM operations were issued after it
        __syncthreads();
        const int jt = j0; j0 = j1; j1 = j2; j2 = jt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // no DMA may still be in flight when the LDS is reused below
    __syncthreads();
    const int jb = wj * 32 + (lane >> 4), ib = wi * 32 + (lane & 15);
    if (kg == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sred[(jb + 4 * r) * BNR_GT + ib] = c00[r];
            sred[(jb + 4 * r) * BNR_GT + ib + 16] = c01[r];
            sred[(jb + 16 + 4 * r) * BNR_GT + ib] = c10[r];
            sred[(jb + 16 + 4 * r) * BNR_GT + ib + 16] = c11[r];
        }
    }
    __syncthreads();
    if (kg == 0) {
        double *out = Gpart + ((size_t)ks * (cd.ntile * (cd.ntile + 1) / 2) + t) * (BNR_GT * BNR_GT);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            bnr_gstore<WT>(out + (jb + 4 * r) * BNR_GT + ib, c00[r] + sred[(jb + 4 * r) * BNR_GT + ib]);
            bnr_gstore<WT>(out + (jb + 4 * r) * BNR_GT + ib + 16, c01[r] + sred[(jb + 4 * r) * BNR_GT + ib + 16]);
            bnr_gstore<WT>(out + (jb + 16 + 4 * r) * BNR_GT + ib, c10[r] + sred[(jb + 16 + 4 * r) * BNR_GT + ib]);
            bnr_gstore<WT>(out + (jb + 16 + 4 * r) * BNR_GT + ib + 16, c11[r] + sred[(jb + 16 + 4 * r) * BNR_GT + ib + 16]);
        }
    }
}
template <class SRC>
__global__ __launch_bounds__(512, 6) void k_gram8d(const SRC chain_src, int s, int nchains)
{
    const int gid = blockIdx.x, gx = gid & 7, gr = gid >> 3;
    const int gchain = gr % nchains, gslot = (gr / nchains) * 8 + gx;
    const bnr_dev &cd = chain_src.at(gchain);
    __shared__ double sred[2 * 5 * 8 * BNR_GT];       // 40 KiB: staging buffers during the loop, then K-group 1's tile (32 KiB of it)
    const bnr_plan_entry P = cd.plan[cd.pbase[0] + s];
    const double *Sp = cd.trace + (size_t)P.prev * cd.rowlen + cd.o_S;
    if (gslot >= cd.ksplit * (cd.ntile * (cd.ntile + 1) / 2)) return;
    const int task = cd.gmap[gslot];
    int t = task & 0xFFFF, ti = 0;
    const int ks = task >> 16;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    int tj = t - ti * (ti + 1) / 2;
    bnr_gram8_task_dma<false>(bnr_geom_of(cd), Sp, cd.Gpart, t, ti, tj, ks, sred);
    bnr_gram_count(cd, tj);
}

struct bnr_gramq { int qoff[9]; };                    // per-XCD task list x = gmapc[qoff[x] .. qoff[x+1])
// k_gram8s: k_gram8 as ONE RESIDENT ROUND with a static task loop -- the launch is 3 x CUs workgroups (all of them resident at once: 32 KiB of
// LDS and <= 80 VGPRs each), workgroup w runs the tasks w, w + grid, w + 2 grid, ... of k_gram8's own 1-D task space (same XCD label: the grid
// is a multiple of 8; same chain-innermost order).  Nothing but the loop counter survives a task, and it is wave-uniform (SGPRs): no queue
// heads, no tickets, no LDS tables -- and no state in device memory that a later launch could find stale.  What it is for: a launch
// whose grid is dispatched in one go does not hold up the other hardware queues (the dispatcher serves no other queue while a many-round
// launch still has workgroups waiting for a slot, profiles/round3_experiments_notes.txt B.4).
// ROT: the issue priority of a workgroup rotates through three levels from task to task, phase = its age rank on the CU (a resident workgroup
// never gets older than its neighbours, and the arbiter serves equal priorities oldest-first).
__device__ __forceinline__ unsigned bnr_hw_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v)); return v; }
__device__ __forceinline__ unsigned bnr_xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 7u; }
template <class SRC, int ROT>
__global__ __launch_bounds__(512, 6) void k_gram8s(const SRC chain_src, int s, int nchains, bnr_gramq gq)
{
    __shared__ double sred[BNR_GT * BNR_GT];
    const bnr_dev &c0 = chain_src.at(0);
    const bnr_gram_geom geom = bnr_geom_of(c0);            // the same for every member (they share X or hold equal copies of its shape)
    const int ntask1 = c0.ksplit * (c0.ntile * (c0.ntile + 1) / 2);
    const int ntot = ((ntask1 + 7) & ~7) * nchains;
    const int rank = ROT == 1 ? (int)(blockIdx.x / (gridDim.x / 3u)) % 3 : -1;
    int nrun = 0;
#ifdef BNR_STAMPS
    const unsigned long long t_in = __builtin_amdgcn_s_memrealtime();
    int nt_ = 0;
    unsigned long long td_ = 0, tt_ = 0;
#endif
    // ROT == 2 (experiment): the static loop over the per-XCD queues of k_gram8q (tile-column order) instead of k_gram8's task map
    const int myx = blockIdx.x & 7, qo = gq.qoff[0] * 0 + (myx == 0 ? gq.qoff[0] : myx == 1 ? gq.qoff[1] : myx == 2 ? gq.qoff[2] : myx == 3 ? gq.qoff[3] : myx == 4 ? gq.qoff[4] : myx == 5 ? gq.qoff[5] : myx == 6 ? gq.qoff[6] : gq.qoff[7]);
    const int qe = (myx == 0 ? gq.qoff[1] : myx == 1 ? gq.qoff[2] : myx == 2 ? gq.qoff[3] : myx == 3 ? gq.qoff[4] : myx == 4 ? gq.qoff[5] : myx == 5 ? gq.qoff[6] : myx == 6 ? gq.qoff[7] : gq.qoff[8]);
    for (int gid = blockIdx.x; gid < (ROT == 2 ? 8 * (qe - qo) * nchains : ntot); gid += gridDim.x) {
        const int gx = gid & 7, gr = gid >> 3;
        const int gchain = gr % nchains, gslot = (gr / nchains) * 8 + gx;
        if (ROT != 2 && gslot >= ntask1) continue;
#ifdef BNR_STAMPS
        const unsigned long long ta_ = __builtin_amdgcn_s_memrealtime();
#endif
        // between two tasks nothing but the loop state is live: the next task's descriptor is read here and goes to scalar registers
        const bnr_dev &cd = chain_src.at(gchain);
        const int prev = bnr_sgpr(cd.plan[cd.pbase[0] + s].prev);
        const double *Sp = bnr_sgpr_global(cd.trace + (size_t)prev * cd.rowlen + cd.o_S);
        double *Gp = bnr_sgpr_global(cd.Gpart);
        bnr_gram_geom g2 = geom;
        g2.X = bnr_sgpr_global(cd.X);
        const int task = bnr_sgpr(ROT == 2 ? cd.gmapc[qo + gr / nchains] : cd.gmap[gslot]);
        int t = task & 0xFFFF, ti = 0;
        const int ks = task >> 16;
        while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
        const int tj = t - ti * (ti + 1) / 2;
        int tid;                                           // opaque per task: nothing derived from the lane id is kept in registers between two tasks
        asm volatile("v_mov_b32 %0, %1" : "=v"(tid) : "v"(threadIdx.x));
        if (ROT == 1) bnr_setprio3((rank + nrun++) % 3);   // per task, outside the batch loop
#ifdef BNR_STAMPS
        const unsigned long long tb_ = __builtin_amdgcn_s_memrealtime();
#endif
        bnr_gram8_task<false>(g2, Sp, Gp, t, ti, tj, ks, sred, -1, tid);
        __syncthreads();                                   // the parked tile has been read: the staging buffers may be written again
#ifdef BNR_STAMPS
        ++nt_; td_ += tb_ - ta_; tt_ += __builtin_amdgcn_s_memrealtime() - tb_;
#endif
    }
    if (ROT == 1) __builtin_amdgcn_s_setprio(0);
#ifdef BNR_STAMPS
    if (threadIdx.x == 0 && blockIdx.x % 3 == 0 && blockIdx.x / 3 >= 66 && blockIdx.x / 3 < 250) {
        unsigned long long *o = c0.dbg + 4 * (blockIdx.x / 3);
        const unsigned hw = bnr_hw_id(), xcc = bnr_xcc_id();
        o[0] = t_in; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = (td_ & 0xfffffull) | ((tt_ & 0xfffffull) << 20);
        o[3] = (unsigned long long)(unsigned)nt_ | ((unsigned long long)xcc << 32) | ((unsigned long long)((hw >> 8) & 15u) << 36) | ((unsigned long long)((hw >> 13) & 7u) << 40);
    }
#endif
}

// k_gram8q: the resident Gram that LEAVES COMPUTE UNITS FREE for the latency chains of the sweep (factorization, scalar branch) -- a kernel
// whose grid is resident in one go does not hold up the other hardware queues, and the CUs it does not use are really empty: a sweeping wave that
// shares its SIMD with MFMA-saturating waves runs 2-4 x slower (profiles/round2_experiments_notes.txt A, round3 I).
//   grid = 3 x CUs workgroups of 512 threads (32 KiB of LDS, <= 80 VGPRs: all resident).  A workgroup that finds itself on a reserved CU --
//   bit (cu id inside its shader engine, HW_REG_HW_ID bits 8..11) of `resv_mask`; cu ids 1..7 exist in every shader engine of the chip,
//   profiles/round3_experiments_notes.txt A.1 -- leaves at once.
//   Work is handed out PER COMPUTE UNIT: the XCD's list of (tile | K slice, chain) entries (host: build_qlist; tile-COLUMN order, equal lengths)
//   is dealt to N "seats" -- seat c owns the entries c, c + N, c + 2 N, ... -- a CU takes a seat when its first workgroup arrives, and the (up
//   to three) workgroups of a CU claim their seat's entries one at a time.  Why per CU: the three workgroups of a CU are not equals (at equal
//   priority the arbiter serves the oldest first: 43 / 64 / 108 us per task by age, tools/stamps_g8q.py), so equal shares per WORKGROUP end with
//   the youngest ones' tasks on an otherwise idle chip (255 against 215 us), and one queue per XCD leaves some CUs with three tasks and others
//   with none in the last round (260 us); a CU's throughput, however, is the same everywhere.  A workgroup whose seat is empty takes a further
//   seat: the share of a CU that never showed up is run by the others -- every entry exactly once whatever the dispatcher does.
//   All atomics stay in the XCD's own L2 (workgroup scope = no sc bits: a global atomic is executed by the L2 it reaches; agent-scope atomics go
//   out to memory, and 96 workgroups per word then queue up for 15-25 us each -- measured).  ctl (zeroed on the issuing stream in front of every
//   launch: no state survives a launch): per XCD x at ctl + 128 x: [0] seats taken, [1 .. 64] seat of hardware CU (se * 16 + cu id) + 1,
//   [65 .. 65 + N) entries claimed per seat.
//   Results do not depend on who computes what: a task's arithmetic is fixed (bitwise k_gram8).
// PUB: partial tiles written through to the agent's coherence point and counted per tile column (gprog) when they have landed.
struct bnr_qent { int task, chain; };                 // task = tile | K slice << 16
#define BNR_GQ_WORDS 128                              // control words per XCD
__device__ __forceinline__ unsigned bnr_l2_add(unsigned *p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// ONE lane: the next entry (index into the XCD's list) for a workgroup sitting on `seat`; -1: nothing left anywhere.  May move to a new seat.
__device__ __forceinline__ int bnr_gramq_claim(unsigned *cx, int *seat, int N, int qlen)
{
    for (;;) {
        if (*seat < N) {
            const int k = (int)bnr_l2_add(cx + 65 + *seat, 1u);
            const long idx = (long)*seat + (long)k * N;
            if (idx < qlen) return (int)idx;
        }
        if (*seat >= N) return -1;
        *seat = (int)bnr_l2_add(cx, 1u);                   // own seat exhausted: is there a seat nobody has taken?
    }
}
template <class SRC, bool PUB>
__global__ __launch_bounds__(512, 6) void k_gram8q(const SRC chain_src, int s, const bnr_qent *qlist, bnr_gramq gq, unsigned resv_mask, unsigned *ctl, int N)
{
    __shared__ double sred[BNR_GT * BNR_GT];
    __shared__ int s_next[2], s_seat;
    // the geometry (equal for all members) is read first and pinned to scalar registers: after the first atomic of the kernel the compiler may
    // no longer use scalar loads, and a uniform value it keeps in vector registers stays there for the whole task loop
    const bnr_dev &c0 = chain_src.at(0);
    bnr_gram_geom geom = bnr_geom_of(c0);
    geom.n_pad = bnr_sgpr(geom.n_pad); geom.q_pad = bnr_sgpr(geom.q_pad); geom.ksplit = bnr_sgpr(geom.ksplit); geom.q = bnr_sgpr(geom.q); geom.ntile = bnr_sgpr(geom.ntile);
    const unsigned hw = bnr_hw_id(), xcc = bnr_xcc_id();
    if ((resv_mask >> ((hw >> 8) & 15u)) & 1u) return;     // (wave-uniform, the same for all eight waves: they share the CU)
    // the thread id is rebuilt per task from the wave's index (a scalar) and the lane's position in the wave (mbcnt): no vector register
    // has to carry it across the tasks (the task body needs 74 of the 80 there are)
    const int wave_s = bnr_sgpr((int)(threadIdx.x >> 6));
#define BNR_GQ_LANE() ((int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)))
    // what runs between two tasks runs at raised issue priority: a wave outside its MFMA loop shares its SIMD with the neighbours' waves inside theirs
    __builtin_amdgcn_s_setprio(3);
    unsigned *cx = ctl + BNR_GQ_WORDS * xcc;
    int qo = gq.qoff[0], qe = gq.qoff[1];                  // (selected by compares: a dynamically indexed kernel argument would live in scratch)
    if (xcc == 1u) { qo = gq.qoff[1]; qe = gq.qoff[2]; } else if (xcc == 2u) { qo = gq.qoff[2]; qe = gq.qoff[3]; } else if (xcc == 3u) { qo = gq.qoff[3]; qe = gq.qoff[4]; }
    else if (xcc == 4u) { qo = gq.qoff[4]; qe = gq.qoff[5]; } else if (xcc == 5u) { qo = gq.qoff[5]; qe = gq.qoff[6]; } else if (xcc == 6u) { qo = gq.qoff[6]; qe = gq.qoff[7]; }
    else if (xcc == 7u) { qo = gq.qoff[7]; qe = gq.qoff[8]; }
    const int qlen = qe - qo;
    if (wave_s == 0 && BNR_GQ_LANE() == 0) {
        // the CU's seat: taken by whoever arrives first (the others of the CU wait for the number, a few hundred ns), then the first claim
        unsigned *mine = cx + 1 + (((hw >> 13) & 3u) * 16u + ((hw >> 8) & 15u));
        unsigned v = __hip_atomic_fetch_or(mine, 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (v == 0u) { v = bnr_l2_add(cx, 1u) + 1u; __hip_atomic_exchange(mine, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        else { int spins = 0; while ((v & 0x7fffffffu) == 0u && ++spins < 100000) { __builtin_amdgcn_s_sleep(2); v = bnr_l2_add(mine, 0u); } }
        int seat = (int)(v & 0x7fffffffu) - 1;
        if (seat < 0) seat = (int)bnr_l2_add(cx, 1u);      // (the seat's owner never published it: take a fresh one)
        s_next[0] = bnr_gramq_claim(cx, &seat, N, qlen);
        s_seat = seat;
    }
#ifdef BNR_STAMPS
    const unsigned long long t_in = __builtin_amdgcn_s_memrealtime();
    int nt_ = 0;
    unsigned long long td_ = 0, tt_ = 0;
#endif
    __syncthreads();
#ifdef BNR_STAMPS
    const unsigned long long t_tk = __builtin_amdgcn_s_memrealtime();
#endif
    for (int cur = 0;; cur ^= 1) {
        const int idx = bnr_sgpr(s_next[cur]);
        if (idx < 0) break;
#ifdef BNR_STAMPS
        const unsigned long long ta_ = __builtin_amdgcn_s_memrealtime();
#endif
        // between two tasks nothing but the loop state is live: the task's descriptor is read here and goes to scalar registers
        const bnr_qent e = qlist[qo + idx];
        const int task = bnr_sgpr(e.task), member = bnr_sgpr(e.chain);
        const bnr_dev &cd = chain_src.at(member);
        const int prev = bnr_sgpr(cd.plan[cd.pbase[0] + s].prev);
        const double *Sp = bnr_sgpr_global(cd.trace + (size_t)prev * cd.rowlen + cd.o_S);
        double *Gp = bnr_sgpr_global(cd.Gpart);
        unsigned int *prog = bnr_sgpr_global(cd.gprog);
        bnr_gram_geom g2 = geom;
        g2.X = bnr_sgpr_global(cd.X);
        int t = task & 0xFFFF, ti = 0;
        const int ks = task >> 16;
        while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
        const int tj = t - ti * (ti + 1) / 2;
        int tid;                                           // opaque per task: nothing derived from the lane id is kept in registers between two tasks
        asm volatile("v_mov_b32 %0, %1" : "=v"(tid) : "v"(wave_s * 64 + BNR_GQ_LANE()));
        __builtin_amdgcn_s_setprio(0);
#ifdef BNR_STAMPS
        const unsigned long long tb_ = __builtin_amdgcn_s_memrealtime();
#endif
        bnr_gram8_task<PUB>(g2, Sp, Gp, t, ti, tj, ks, sred, -1, tid);
        __builtin_amdgcn_s_setprio(3);
        if (PUB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every wave's write-through stores have landed ...
        __syncthreads();                                                      // (also: the parked tile has been read)
        if (wave_s == 0 && BNR_GQ_LANE() == 0) {
            if (PUB) __hip_atomic_fetch_add(&prog[tj], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... then ONE count for the tile column
            int seat = s_seat;                                                // the next entry (one atomic in the own L2: ~1 us, the CU's other workgroups work meanwhile)
            s_next[cur ^ 1] = bnr_gramq_claim(cx, &seat, N, qlen);
            s_seat = seat;
        }
        __syncthreads();
#ifdef BNR_STAMPS
        ++nt_; td_ += tb_ - ta_; tt_ += __builtin_amdgcn_s_memrealtime() - tb_;
#endif
    }
#ifdef BNR_STAMPS
    if (wave_s == 0 && BNR_GQ_LANE() == 0 && blockIdx.x % 3 == 0 && blockIdx.x / 3 >= 66 && blockIdx.x / 3 < 250) {
        unsigned long long *o = c0.dbg + 4 * (blockIdx.x / 3);
        o[0] = t_in; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = (td_ & 0xfffffull) | ((tt_ & 0xfffffull) << 20) | (((t_tk - t_in) & 0xfffffull) << 40);
        o[3] = (unsigned long long)(unsigned)nt_ | ((unsigned long long)xcc << 32) | ((unsigned long long)((hw >> 8) & 15u) << 36) | ((unsigned long long)((hw >> 13) & 7u) << 40);
    }
#endif
}

// k_gram8p: the same tasks, the same partial tiles bit for bit, as a PERSISTENT kernel that keeps off a set of reserved compute
// units, so that the factorization (k_chol_ll on another stream) and the scalar branch of the sweep have CUs of their own while
// the Gram saturates the matrix cores of the rest -- CU-masked streams do not survive graph capture, and a latency-bound wave that
// shares a SIMD with MFMA-saturating waves is starved (profiles/round3_experiments_notes.txt A).
//   grid = 3 x CUs workgroups (the kernel's residency): a workgroup that finds itself on a reserved CU (HW_REG_HW_ID /
//   HW_REG_XCC_ID against the per-shader-engine masks `resv`, measured by k_cu_census at start-up) leaves at once; the others pull
//   (tile, K slice, chain) tasks from eight queues, one per XCD: queue x lists the K slices ks = x mod 8 in tile-COLUMN order
//   (the factorization consumes G column by column), chains innermost -- the workgroups of one XCD read the same slice of X
//   through its L2; an XCD whose queue has run dry takes from the others.
//   Results do not depend on who computes what: a task's arithmetic is fixed.  Progress: the last workgroup to leave (by ticket)
//   finishes whatever is left in the queues, so the launch completes even if every other workgroup sat on a reserved CU.
//   ctl (device words, zero between launches): [0..7] queue heads, [8] tickets.
// next task of a persistent Gram workgroup (ONE wavefront calls this): lanes 0..7 read the eight queue heads (plain loads: a dry
// queue costs no atomic -- 768 workgroups that each probed every head at the end would queue up ~9 us per word), the first
// queue with work at or after the own XCD is chosen and ONE atomic takes a ticket from it; -1 when every queue is dry.
__device__ __forceinline__ void bnr_gram_fetch(unsigned *ctl, const int *s_qlen, unsigned xcc, int lane_in, int *s_task)
{
    int lane;                                          // (opaque copy: nothing derived from it is kept in registers between two calls)
    asm volatile("v_mov_b32 %0, %1" : "=v"(lane) : "v"(lane_in));
    int x = -1, id = 0;
    for (;;) {
        const unsigned head = lane < 8 ? __hip_atomic_load(&ctl[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        const bool work = lane < 8 && (int)head < s_qlen[lane];
        unsigned m = (unsigned)(__ballot(work) & 0xffull);
        if (m == 0u) { x = -1; break; }
        m = ((m >> xcc) | (m << (8u - xcc))) & 0xffu;                  // rotate: bit 0 = own XCD's queue
        x = (int)((xcc + (unsigned)__builtin_ctz(m)) & 7u);
        unsigned got = 0;
        if (lane == 0) got = __hip_atomic_fetch_add(&ctl[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        id = __builtin_amdgcn_readfirstlane((int)got);
        if (id < s_qlen[x]) break;                                     // else: the queue ran dry meanwhile, look again
    }
    if (lane == 0) { s_task[0] = x; s_task[1] = id; }
}
// PUB: the factorization runs beside this launch: partial tiles written through to the agent's coherence point (sc1) and counted only
// when every wave's stores have landed; otherwise plain stores (consumed after the kernel boundary).
// Two workgroups per CU with the 16-column body of k_gram (64 KiB of LDS, up to 128 VGPRs): three per CU would need the task loop in 80
// VGPRs -- the 8-column body alone takes 74, the loop's live state then spills, and the spill code moved 63 KB of scratch per task = 128 MB
// per 8-chain launch, twice the partial tiles (3.5 x the L2 misses, +40 % on the launch; profiles/round3_experiments_notes.txt B.2).
template <class SRC, bool PUB>
__global__ __launch_bounds__(512, 2 * BNR_G8P_WPC) void k_gram8p(const SRC chain_src, int s, int nchains, bnr_gramq gq, const unsigned *resv, unsigned *ctl)
{
    __shared__ double sred[2 * BNR_GT * BNR_GT];
    __shared__ int s_task[4], s_qlen[8], s_qoff[8], s_ticket;             // s_task: {queue, id} of the current and of the next task
    // per member: the S row of this sweep, the partial-tile buffer, the progress words -- read once per workgroup, so that a task
    // starts from two LDS reads instead of a chain of dependent global loads (descriptor -> plan entry -> row)
    constexpr int TABMAX = 64;
    __shared__ const double *s_Sp[TABMAX];
    __shared__ double *s_Gp[TABMAX];
    __shared__ unsigned int *s_prog[TABMAX];
    const unsigned hw = bnr_hw_id(), xcc = bnr_xcc_id();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool reserved = resv && ((resv[xcc * 4 + ((hw >> 13) & 3u)] >> ((hw >> 8) & 15u)) & 1u);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int x = 0; x < 8; ++x) { s_qoff[x] = gq.qoff[x]; s_qlen[x] = (gq.qoff[x + 1] - gq.qoff[x]) * nchains; }
        if (reserved) s_ticket = (int)__hip_atomic_fetch_add(&ctl[8], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const bool tabled = nchains <= TABMAX;
    if (!reserved && tabled && (int)threadIdx.x < nchains) {
        const bnr_dev &cm = chain_src.at(threadIdx.x);
        const bnr_plan_entry Pm = cm.plan[cm.pbase[0] + s];
        s_Sp[threadIdx.x] = cm.trace + (size_t)Pm.prev * cm.rowlen + cm.o_S;
        s_Gp[threadIdx.x] = cm.Gpart;
        s_prog[threadIdx.x] = cm.gprog;
    }
    const bnr_dev &c0 = chain_src.at(0);
    const bnr_gram_geom geom = bnr_geom_of(c0);
    const int *gmapc = c0.gmapc;
    __syncthreads();
    // a workgroup on a reserved CU leaves at once -- unless it is the last one out of the whole grid: then it finishes what is left
    const bool sweeper = reserved && __builtin_amdgcn_readfirstlane(s_ticket) == (int)gridDim.x - 1;
    if (reserved && !sweeper) return;
    const int rank = (int)(blockIdx.x / (gridDim.x / BNR_G8P_WPC)) % BNR_G8P_WPC;   // the dispatcher fills the CUs one workgroup per pass: age rank on the CU
    if (wave == 7) bnr_gram_fetch(ctl, s_qlen, xcc, lane, s_task);
    __syncthreads();
    for (int cur = 0;; cur ^= 2) {
        const int x = __builtin_amdgcn_readfirstlane(s_task[cur]), id = __builtin_amdgcn_readfirstlane(s_task[cur + 1]);
        if (x < 0) break;
        const int member = id % nchains;
        const double *Sp;
        double *Gp;
        unsigned int *prog;
        if (tabled && !sweeper) { Sp = s_Sp[member]; Gp = s_Gp[member]; prog = s_prog[member]; }
        else {
            const bnr_dev &cm = chain_src.at(member);
            const bnr_plan_entry Pm = cm.plan[cm.pbase[0] + s];
            Sp = cm.trace + (size_t)Pm.prev * cm.rowlen + cm.o_S; Gp = cm.Gpart; prog = cm.gprog;
        }
        // wave-uniform and known to be GLOBAL pointers (a pointer that went through LDS is generic: the compiler would emit flat loads,
        // which count on lgkmcnt as well -- every wait for an LDS fragment would then wait for the S load's memory latency too)
        {
            const unsigned long long sp_ = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)Sp >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long long)Sp);
            const unsigned long long gp_ = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)Gp >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long long)Gp);
            const unsigned long long pp_ = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)prog >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long long)prog);
            Sp = (const double *)(__attribute__((address_space(1))) const void *)sp_;
            Gp = (double *)(__attribute__((address_space(1))) void *)gp_;
            prog = (unsigned int *)(__attribute__((address_space(1))) void *)pp_;
        }
        const int task = __builtin_amdgcn_readfirstlane(gmapc[__builtin_amdgcn_readfirstlane(s_qoff[x]) + id / nchains]);   // wave-uniform: the loop's addressing stays on scalars
        int t = task & 0xFFFF, ti = 0;
        const int ks = task >> 16;
        while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
        const int tj = t - ti * (ti + 1) / 2;
        // the NEXT task is fetched now, by the last wave, behind the first loads of this one (two dependent round trips that would
        // otherwise sit between two tasks)
        if (wave == 7) bnr_gram_fetch(ctl, s_qlen, xcc, lane, s_task + (cur ^ 2));
        bnr_gram16_task<2, PUB>(geom, Sp, Gp, t, ti, tj, ks, sred, rank);
        // publish: every wave's write-through stores have landed (vmcnt), then ONE relaxed atomic on the column's counter
        if (PUB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&prog[tj], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_s_setprio(0);
    if (!sweeper) {
        if (threadIdx.x == 0) s_ticket = (int)__hip_atomic_fetch_add(&ctl[8], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane(s_ticket) != (int)gridDim.x - 1) return;
    }
    // last one out: everybody else has left its loop -- the queue heads and the ticket counter go back to zero for the next launch
    if (threadIdx.x < 9) __hip_atomic_store(&ctl[threadIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// which compute units exist: every workgroup ORs its (XCD, shader engine) -> cu-id bit into out[xcc * 4 + se] and idles a little so
// that the grid spreads over the whole chip
__global__ void k_cu_census(unsigned *out, int spin)
{
    if (threadIdx.x == 0) {
        const unsigned hw = bnr_hw_id(), xcc = bnr_xcc_id();
        atomicOr(&out[xcc * 4 + ((hw >> 13) & 3u)], 1u << ((hw >> 8) & 15u));
        atomicAdd(&out[32], 1u);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(4);
    }
}


// ----------------------------------------------------------------------------------------- left-looking factorization
// k_chol_ll(p), p = 0..nbk-1: the same factorization of E = [G + I ; I] -> [L ; L^-T], the same arithmetic per element (every
// element sees the panels' rank-32 updates in ascending panel order, eight MFMA k-steps each, then the same column sweep), but
// block column j is touched for the FIRST time only at launch j-1 -- and there the K-split partials of the Gram are summed in
// k_gram_reduce's order and the identity rows are generated on the fly: no reduction pass, no Y = I pass, and the factorization
// of the leading columns can start while the Gram still computes the trailing ones (gate on cd.gprog, see bnr_gram_gate).
//   role A (ceil(nbk/4) workgroups): FOUR sweeping wavefronts, one per SIMD, each owning one block row of panel p (matrix rows
//       p+1.., identity rows 0..p): block (p,p) and the own block take panel p-1's update (MFMA, fragments from L2), then every
//       wave sweeps [D ; own] -- lanes 0..31 redo the diagonal block, so nothing is handed over between waves or workgroups.
//   role B (nbk-1 workgroups, one 32 x 32 block each, one 16 x 16 tile per wave): block column j = p+1 <- first touch - sum_{q<p} L[.,q] L[j,q]',
//       i.e. everything except panel p's own update, which role A of launch p+1 applies.
// Footprint per chain and launch: ceil(nbk/4) + nbk - 1 workgroups (19 at n = 500) against nbk + 1 + updates (17 + up to 120) before.
#define BNR_L1W 48            // LDS column stride of the 32-row half-panels handed to the MFMA update (conflict-free fragments)
#define BNR_LL_WAVE (BNR_NB * BNR_LP + 16 * BNR_L1W + 2 * BNR_NB)            // doubles per wave: sB | sL1 | sCol
#define BNR_LL_LDS ((BNR_NB * BNR_LP + 16 * BNR_L1W + 4 * BNR_LL_WAVE) * sizeof(double))   // + shared sD | sL1D

// Before the first read of tile column tc of Gpart: all its (tile, K slice) tasks of THIS sweep's Gram must have been published.
// spin_us = 0: the Gram launch is complete (single-stream schedule, hooks) -- a shortfall is a stream-ordering violation.
// spin_us > 0: the Gram may still be running beside this launch; lane 0 polls (relaxed, with s_sleep) for at most spin_us, then
// gives up for good (sticky word, so that a schedule that was serialised after all costs ONE timeout, not one per gate).
__device__ __forceinline__ void bnr_gram_gate(const bnr_dev &cd, int tc, int spin_us)
{
    if (threadIdx.x == 0) {
        const unsigned need = (unsigned)((cd.ntile - tc) * cd.ksplit);
        unsigned have = __hip_atomic_load(&cd.gprog[tc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (have < need && spin_us > 0 && __hip_atomic_load(&cd.gprog[cd.ntile + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            __builtin_amdgcn_s_setprio(3);                 // a young wave among older MFMA-saturating ones is served last: one poll took 450 us
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), lim = 100ull * (unsigned long long)spin_us;
            do {
                __builtin_amdgcn_s_sleep(8);
                have = __hip_atomic_load(&cd.gprog[tc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } while (have < need && __builtin_amdgcn_s_memrealtime() - t0 < lim);
            if (have < need) __hip_atomic_store(&cd.gprog[cd.ntile + 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_setprio(0);
        }
        if (have < need) atomicAdd((unsigned long long *)&cd.counters[8], 1ull);
    }
    __syncthreads();                                       // (the partial tiles are then read past the L2: bnr_ld_fresh)
}

// Pipelined schedule: the first node of the factorization branch.  If k_chol_ll(0) itself waited for the first tile column, its
// workgroups would be dispatched all over the chip before the Gram's are (both branches start together) and sit there spinning: a
// Gram CU that hosts one of them has room for ONE Gram workgroup instead of three (measured: Gram 335 instead of 210 us, and the
// column it waits for arrives after 220 us instead of 60).  One small wavefront per chain waits instead; 72 KiB of dynamic LDS
// keep it off the Gram's CUs (a young wave on a SIMD that older MFMA-saturating waves keep busy is starved).
template <class SRC>
__global__ __launch_bounds__(64) void k_gram_gate(const SRC chain_src, int tc, int spin_us)
{
    const bnr_dev &cd = chain_src.get_x();               // grid = chains
    bnr_gram_gate(cd, tc, spin_us);
}
template <class SRC>
__global__ __launch_bounds__(256, 1) void k_chol_ll(const SRC chain_src, int p, int s, int nA, int spin_us)
{
    const bnr_dev &cd = chain_src.get_x();               // grid = (chains, workgroups)
    extern __shared__ double shll[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nbk = cd.n_pad / BNR_NB;
    const size_t ld = bnr_ldE(cd.n_pad);
    const int pc = p * BNR_NB, kc = pc - BNR_NB;
    const int ln = lane & 15, lq = lane >> 4;
    double *E = cd.E;
    (void)s;
    if ((int)blockIdx.y >= nA) {
        // ------------------------------------------------ role B: block column j = p + 1, everything but panel p's update
        const int j = p + 1;
        if (j >= nbk) return;
        if (((j & 1) == 0 || p == 0)) bnr_gram_gate(cd, j >> 1, spin_us);      // first touch of tile column j/2 (odd j > 1: gated one launch ago)
        const int a = (int)blockIdx.y - nA;                                    // matrix rows j..nbk-1, then identity rows 0..p-1
        const bool ident = a >= nbk - j;
        const int r_id = a - (nbk - j);
        const int R = ident ? nbk + r_id : j + a, q0 = ident ? r_id : 0;
        const int at = wave >> 1, bt = wave & 1;                               // this wave's 16 x 16 tile: columns at, rows bt
        double *cp = E + (size_t)(R * BNR_NB + 16 * bt + ln) + ld * (size_t)(j * BNR_NB + 16 * at + lq);
        bnr_d4 c = {0.0, 0.0, 0.0, 0.0};
        const double *colrows = E + (size_t)(j * BNR_NB + 16 * at + ln) + ld * (size_t)lq, *rowrows = E + (size_t)(R * BNR_NB + 16 * bt + ln) + ld * (size_t)lq;
        // The panels q = q0..p-1 are applied in order (one dependent MFMA chain); what bounds this loop is the L2 latency of the
        // fragment loads, so the fragments of the next three panels are kept in flight (a ring of four register sets), and the
        // steady state has no branches (a branch makes the compiler drain the loads in flight).  Loads past the last panel are
        // clamped to it and never used.
        double fa[4][8], fb[4][8];
#define BNR_LLB_LOAD(SET, Q)                                                                              \
        do {                                                                                              \
            const size_t o_ = ld * (size_t)((Q) * BNR_NB);                                                \
            _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) { fa[SET][ks] = colrows[o_ + ld * (size_t)(4 * ks)]; fb[SET][ks] = rowrows[o_ + ld * (size_t)(4 * ks)]; } \
        } while (0)
#define BNR_LLB_MFMA(SET)                                                                                 \
        do {                                                                                              \
            _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-fa[SET][ks], fb[SET][ks], c, 0, 0, 0); \
        } while (0)
        const int last = p - 1;
        int q = q0;
        if (p > 0) {
            BNR_LLB_LOAD(0, q < last ? q : last);
            BNR_LLB_LOAD(1, q + 1 < last ? q + 1 : last);
            BNR_LLB_LOAD(2, q + 2 < last ? q + 2 : last);
        }
        if (!ident) {
            bnr_d4 t1[1][1];
            bnr_gsum_frag<1>(cd, R, j, at, bt, ln, lq, t1, spin_us > 0);                    // first touch: the K-slice partials of G (+ I)
            c = t1[0][0];
        }
        if (p > 0) {
            for (; q + 4 <= p; q += 4) {
                BNR_LLB_LOAD(3, q + 3);
                BNR_LLB_MFMA(0);
                BNR_LLB_LOAD(0, q + 4 < last ? q + 4 : last);
                BNR_LLB_MFMA(1);
                BNR_LLB_LOAD(1, q + 5 < last ? q + 5 : last);
                BNR_LLB_MFMA(2);
                BNR_LLB_LOAD(2, q + 6 < last ? q + 6 : last);
                BNR_LLB_MFMA(3);
            }
            if (q < p) BNR_LLB_MFMA(0);
            if (q + 1 < p) BNR_LLB_MFMA(1);
            if (q + 2 < p) BNR_LLB_MFMA(2);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) cp[ld * (size_t)(4 * r)] = c[r];
        return;
    }
    // ---------------------------------------------------- role A: four sweeping wavefronts
    double *sD = shll, *sL1D = shll + BNR_NB * BNR_LP;
    double *mine = shll + BNR_NB * BNR_LP + 16 * BNR_L1W + wave * BNR_LL_WAVE;
    double *sB = mine, *sL1 = mine + BNR_NB * BNR_LP;
    double (*sCol)[BNR_NB] = (double (*)[BNR_NB])(mine + BNR_NB * BNR_LP + 16 * BNR_L1W);
#ifdef BNR_STAMPS
#define BNR_LSTAMP(slot) do { if (blockIdx.y == 0 && tid == 0) cd.dbg[p * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BNR_LSTAMP(slot) do { } while (0)
#endif
    BNR_LSTAMP(0);
    if (p == 0) bnr_gram_gate(cd, 0, spin_us);
    const int a = 4 * (int)blockIdx.y + wave;                                  // matrix rows p+1..nbk-1, then identity rows 0..p
    const bool active = a < nbk;                                               // a spare wave runs along on block row p (valid memory) and stores nothing
    const bool ident = active && a >= nbk - 1 - p;
    const int r_id = a - (nbk - 1 - p);
    const int R = !active ? p : (ident ? nbk + r_id : p + 1 + a);
    {
        // Block (p,p): one 16 x 16 tile per wave (columns mt, rows nt); own block: 2 x 2 tiles -- both as MFMA accumulator
        // fragments.  Identity rows: row p is the identity block itself and takes no update, row p-1 starts from zero, both
        // without reading E (nobody initialises Y); rows < p-1 were prepared by role B of the previous launch.
        const int mt = wave >> 1, nt = wave & 1;
        bnr_d4 cD, c[2][2];
        if (p == 0) {
            bnr_d4 t1[1][1];
            bnr_gsum_frag<1>(cd, 0, 0, mt, nt, ln, lq, t1, spin_us > 0);
            cD = t1[0][0];
            if (!ident) bnr_gsum_frag<2>(cd, active ? R : 0, 0, 0, 0, ln, lq, c, spin_us > 0);
        } else {
            // every global load of the launch is issued here, back to back, before the first wait: one L2 round trip
            const bool fresh = ident && r_id >= p - 1, skip = ident && r_id == p;
            const double *dp = E + (size_t)(pc + nt * 16 + ln) + ld * (size_t)(pc + mt * 16 + lq);
            const double *bp = E + (size_t)((fresh ? p : R) * BNR_NB + ln) + ld * (size_t)(pc + lq);
            const double *colrows = E + (size_t)(pc + ln) + ld * (size_t)(kc + lq), *rowrows = E + (size_t)((skip ? p : R) * BNR_NB + ln) + ld * (size_t)(kc + lq);
            double av[2][8], bv[2][8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const size_t o = ld * (size_t)(4 * ks);
                av[0][ks] = colrows[o]; av[1][ks] = colrows[o + 16]; bv[0][ks] = rowrows[o]; bv[1][ks] = rowrows[o + 16];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) cD[r] = dp[ld * (size_t)(4 * r)];
#pragma unroll
            for (int at = 0; at < 2; ++at)
#pragma unroll
                for (int bt = 0; bt < 2; ++bt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) c[at][bt][r] = bp[(size_t)(16 * bt) + ld * (size_t)(16 * at + 4 * r)];
            // the diagonal block's fragments are rows of L[p, p-1] as well: column side = av[mt], row side = av[nt]
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) cD = __builtin_amdgcn_mfma_f64_16x16x4f64(-(mt ? av[1][ks] : av[0][ks]), nt ? av[1][ks] : av[0][ks], cD, 0, 0, 0);
            if (fresh) {
#pragma unroll
                for (int at = 0; at < 2; ++at)
#pragma unroll
                    for (int bt = 0; bt < 2; ++bt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) c[at][bt][r] = (skip && 16 * bt + ln == 16 * at + lq + 4 * r) ? 1.0 : 0.0;
            }
            if (!skip) {
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                    for (int at = 0; at < 2; ++at)
#pragma unroll
                        for (int bt = 0; bt < 2; ++bt) c[at][bt] = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[at][ks], bv[bt][ks], c[at][bt], 0, 0, 0);
            }
        }
        if (p == 0 && ident) {                                                 // p = 0: the only identity row is row 0 = the identity block
#pragma unroll
            for (int at = 0; at < 2; ++at)
#pragma unroll
                for (int bt = 0; bt < 2; ++bt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) c[at][bt][r] = (16 * bt + ln == 16 * at + lq + 4 * r) ? 1.0 : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sD[(nt * 16 + ln) + BNR_LP * (mt * 16 + lq + 4 * r)] = cD[r];
#pragma unroll
        for (int at = 0; at < 2; ++at)
#pragma unroll
            for (int bt = 0; bt < 2; ++bt)
#pragma unroll
                for (int r = 0; r < 4; ++r) sB[(16 * bt + ln) + BNR_LP * (16 * at + lq + 4 * r)] = c[at][bt][r];
    }
    BNR_LSTAMP(1);
    __syncthreads();
    BNR_LSTAMP(2);
    const int rr = lane & 31;
    const double *src = (lane < 32) ? sD : sB;
    double a1[16], a2[16];
    int bad = 0;
    {
#pragma unroll
        for (int c = 0; c < 16; ++c) a1[c] = src[rr + BNR_LP * c];
        bad = bnr_sweep16<0>(a1, lane, sCol);
        // first half of the panel for the MFMA update of the second: diagonal rows once per workgroup, own rows per wave
        if (lane >= 32) {
#pragma unroll
            for (int c = 0; c < 16; ++c) sL1[c * BNR_L1W + rr] = a1[c];
        } else if (wave == 0) {
#pragma unroll
            for (int c = 0; c < 16; ++c) sL1D[c * BNR_L1W + rr] = a1[c];
        }
    }
    BNR_LSTAMP(3);
    __syncthreads();
    {
        //   A[:, 16:32] -= L[:, 0:16] L[16:32, 0:16]'   -- diagonal rows: waves 0 and 1 (16 rows each), own rows: 2 tiles per wave
        double avk[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) avk[ks] = sL1D[(4 * ks + lq) * BNR_L1W + 16 + ln];      // column side: rows 16..31 of the diagonal block
        if (wave < 2) {
            const int rowb = wave * 16;
            bnr_d4 c;
#pragma unroll
            for (int r = 0; r < 4; ++r) c[r] = sD[(rowb + ln) + BNR_LP * (16 + lq + 4 * r)];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-avk[ks], sL1D[(4 * ks + lq) * BNR_L1W + rowb + ln], c, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) sD[(rowb + ln) + BNR_LP * (16 + lq + 4 * r)] = c[r];
        }
        {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bnr_d4 c;
#pragma unroll
                for (int r = 0; r < 4; ++r) c[r] = sB[(t * 16 + ln) + BNR_LP * (16 + lq + 4 * r)];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-avk[ks], sL1[(4 * ks + lq) * BNR_L1W + t * 16 + ln], c, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) sB[(t * 16 + ln) + BNR_LP * (16 + lq + 4 * r)] = c[r];
            }
        }
    }
    __syncthreads();
    BNR_LSTAMP(4);
    {
#pragma unroll
        for (int c = 0; c < 16; ++c) a2[c] = src[rr + BNR_LP * (16 + c)];
        bad |= bnr_sweep16<16>(a2, lane, sCol);
        BNR_LSTAMP(5);
        // the swept own block B L_D^-T straight from the registers (lane = row); the factored diagonal block is needed by nobody
        if (active && lane >= 32) {
            double *op = E + (size_t)(R * BNR_NB + rr) + ld * (size_t)pc;
#pragma unroll
            for (int c = 0; c < 16; ++c) { op[ld * (size_t)c] = a1[c]; op[ld * (size_t)(16 + c)] = a2[c]; }
        }
    }
    BNR_LSTAMP(6);
    if (bad && tid == 0 && blockIdx.y == 0) { atomicAdd((unsigned long long *)&cd.counters[3], 1ull); atomicAdd((unsigned long long *)&cd.counters[7], 1ull); }
    // the last launch has seen every gate pass: zero the progress words for the next sweep's Gram
    if (p == nbk - 1 && blockIdx.y == 0 && tid <= cd.ntile) cd.gprog[tid] = 0u;
}


// k_backproj for a lockstep group whose members share the model matrix: ONE workgroup of 64 CT WPC threads owns a block of 32 edges for CT
// chains (CT = 8, WPC = 1: 512 threads; the chain's pointers and scalars in scalar registers so that the draws fit their vector registers).  A column of X is loaded once for the CT back-projections x_e' a4_c, and the GIG draws of all 32 CT (chain, edge)
// pairs run side by side, 2 WPC speculative attempts each (a wave = one chain's 32 edges x 2 attempts: as many as the per-chain kernel makes
// for a group -- attempts that are thrown away cost what they save once the chip is busy) -- the per-chain kernel's
// workgroups are latency chains of 16-18 us (dot products 6, draws 7-10, sums 2.4 us: tools/stamps_bp.py), and eight chains' worth of
// them took two rounds on the chip (31 us against 19 for one chain).  Per chain exactly the per-chain kernel's arithmetic: products
// accumulated over the rows lane, lane + 64, ... then the wave reduction; the first accepted attempt of the counter sequence; the
// partial sums per 32-edge block in the same order.  flags = 7 only.
//   grid = (nblk_bp, ceil(chains / CT)); dynamic LDS = CT x (max(n_pad, (3R+1) 33) + 33 R + 32) doubles.
// a GIG context through LDS (22 doubles)
__device__ __forceinline__ void bnr_gig_ctx_put(const bnr_gig_ctx &g, double *d)
{
    d[0] = g.kind; d[1] = g.lambda_old; d[2] = g.lambda; d[3] = g.alpha; d[4] = g.omega; d[5] = g.xm; d[6] = g.t; d[7] = g.s; d[8] = g.nc; d[9] = g.ulo; d[10] = g.uhi;
    d[11] = g.xoff; d[12] = g.x0; d[13] = g.k0; d[14] = g.A0; d[15] = g.A1; d[16] = g.A2; d[17] = g.k1; d[18] = g.k2; d[19] = g.Atot; d[20] = g.x0l; d[21] = g.half;
}
__device__ __forceinline__ void bnr_gig_ctx_get(bnr_gig_ctx &g, const double *d)
{
    g.kind = (int)d[0]; g.lambda_old = d[1]; g.lambda = d[2]; g.alpha = d[3]; g.omega = d[4]; g.xm = d[5]; g.t = d[6]; g.s = d[7]; g.nc = d[8]; g.ulo = d[9]; g.uhi = d[10];
    g.xoff = d[11]; g.x0 = d[12]; g.k0 = d[13]; g.A0 = d[14]; g.A1 = d[15]; g.A2 = d[16]; g.k1 = d[17]; g.k2 = d[18]; g.Atot = d[19]; g.x0l = d[20]; g.half = (int)d[21];
}
// a wave-uniform 64-bit value moved to scalar registers
__device__ __forceinline__ const void *bnr_uniform_ptr(const void *p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (const void *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double bnr_uniform_f64(double x)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}
struct bnr_bp_chain { double *row; const double *prev; const double *a4, *Wbuf, *sz; double *Psum; long long *counters; unsigned long long seed; unsigned int it; int pad; };
#define BNR_BPG_CT 8          // chains per workgroup
#define BNR_BPG_WPC 1         // waves per chain in the draws: 2 WPC speculative attempts per draw and round
__global__ __launch_bounds__(64 * BNR_BPG_CT * BNR_BPG_WPC) void k_backproj_group(const bnr_many chain_src, int s, int nchains)
{
    BNR_CRITICAL_PATH();
    const bnr_dev &c0 = chain_src.at(0);                  // geometry, the shared X and index maps
    constexpr int CT = BNR_BPG_CT, WPC = BNR_BPG_WPC, NT = 64 * CT * WPC, NW = CT * WPC, NSLOT = 2 * WPC;
    static_assert(BNR_BPG_WPC == 1, "the draws below are wave-local: one wave per chain");
    const int bid = blockIdx.x, cb = blockIdx.y * CT, nc = min(CT, nchains - cb);
    const int R = c0.R, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n_pad = c0.n_pad;
    const int e0 = bid * c0.chunk_bp, ne = min(c0.chunk_bp, c0.q - e0);
    const size_t ld = n_pad;
#ifdef BNR_STAMPS
#define BNR_GSTAMP(slot) do { if (tid == 0 && bid == 7 && blockIdx.y == 0) c0.dbg[330 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BNR_GSTAMP(slot) do { } while (0)
#endif
    BNR_GSTAMP(0);
    extern __shared__ double sh[];
    const int SA = max(max(n_pad, (3 * R + 1) * 33), 32 * 22), STR = SA + 33 * R + 32;
    __shared__ bnr_bp_chain s_cd[CT];
    __shared__ double s_val[CT][NSLOT][32];
    __shared__ int s_acc[CT][NSLOT][32];
    if (tid < nc) {
        const bnr_dev &cd = chain_src.at(cb + tid);
        const bnr_plan_entry P = cd.plan[cd.pbase[0] + s];
        bnr_bp_chain t;
        t.row = cd.trace + (size_t)P.row * cd.rowlen; t.prev = cd.trace + (size_t)P.prev * cd.rowlen;
        t.a4 = cd.a4; t.Wbuf = cd.Wbuf; t.sz = cd.sz; t.Psum = cd.Psum; t.counters = cd.counters; t.seed = cd.seed; t.it = P.it; t.pad = 0;
        s_cd[tid] = t;
    }
    __shared__ int s_el[32], s_ek[32];
    if (tid >= 64 && tid < 96) { const int ee = tid - 64; s_el[ee] = ee < ne ? c0.el[e0 + ee] : 0; s_ek[ee] = ee < ne ? c0.ek[e0 + ee] : 0; }   // the same nodes for every chain
    __syncthreads();
    // u[r,l] u[r,k] of the block's edges (for the Lambda log-likelihoods at the end) and a4, per chain
#pragma unroll 4
    for (int it = tid; it < nc * R * 32; it += NT) {
        const int c = it / (R * 32), idx = it - c * (R * 32), r = idx >> 5, ee = idx & 31;
        const double *un0 = s_cd[c].row + c0.o_u;
        const double v = un0[r + R * s_el[ee]] * un0[r + R * s_ek[ee]];
        sh[c * STR + SA + r * 33 + ee] = ee < ne ? v : 0.0;
    }
    for (int it = tid; it < nc * n_pad; it += NT) { const int c = it / n_pad, i = it - c * n_pad; sh[c * STR + i] = s_cd[c].a4[i]; }
    __syncthreads();
    BNR_GSTAMP(1);
    {
        // wave w: columns w, w + NW, w + 2 NW, ... of the block, two at a time, for every chain; all loads of a 512-row stretch before the first multiply
      for (int t0 = wave; t0 < 32; t0 += 2 * NW) {
        const int t1 = t0 + NW;
        double acc0[CT], acc1[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) { acc0[c] = 0.0; acc1[c] = 0.0; }
        const size_t o0 = (size_t)(e0 + (t0 < ne ? t0 : 0)) * ld, o1 = (size_t)(e0 + (t1 < ne ? t1 : 0)) * ld;
#define BNR_BPG_DOTS(XP)                                                                                   \
        for (int i0 = 0; i0 < n_pad; i0 += 512) {                                                         \
            double x0[8], x1[8];                                                                          \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                               \
                const int i = i0 + lane + 64 * j, ic = i < n_pad ? i : lane;                              \
                x0[j] = (double)(XP)[o0 + ic]; x1[j] = (double)(XP)[o1 + ic];                             \
            }                                                                                             \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                               \
                /* branch-free: a row step past n_pad enters with x = 0 (acc + 0 a = acc), a chain past nc computes and is not stored */ \
                const int i = i0 + lane + 64 * j;                                                         \
                const bool in = i < n_pad;                                                                \
                const int ii = in ? i : lane;                                                             \
                const double xa = in ? x0[j] : 0.0, xb = in ? x1[j] : 0.0;                                \
                _Pragma("unroll") for (int c = 0; c < CT; ++c) {                                          \
                    const double av = sh[c * STR + ii]; acc0[c] = fma(xa, av, acc0[c]); acc1[c] = fma(xb, av, acc1[c]); \
                }                                                                                         \
            }                                                                                             \
        }
        if (c0.X8) { BNR_BPG_DOTS(c0.X8) } else { BNR_BPG_DOTS(c0.X) }
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            if (c < nc) {
                const double v0 = wave_sum(acc0[c]), v1 = wave_sum(acc1[c]);
                if (lane == 0) { if (t0 < ne) sh[c * STR + SA + 33 * R + t0] = v0; if (t1 < ne) sh[c * STR + SA + 33 * R + t1] = v1; }
            }
        }
      }
    }
    __syncthreads();
    BNR_GSTAMP(2);
    // update_D! for (chain c = wave / WPC, edge el32, attempt slot 0..2 WPC - 1): the first accepted attempt of the counter sequence wins
    const int c = __builtin_amdgcn_readfirstlane(wave / WPC), el32 = lane & 31, slot = (wave % WPC) * 2 + (lane >> 5), e = e0 + el32;
    const bool act = el32 < ne && c < nc;
    // the chain's pointers and scalars are the same for the whole wave: kept in scalar registers (the draws below need every vector register)
    bnr_bp_chain cc;
    {
        const bnr_bp_chain &m = s_cd[c < nc ? c : 0];
        cc.row = (double *)bnr_uniform_ptr(m.row); cc.prev = (const double *)bnr_uniform_ptr(m.prev); cc.a4 = nullptr;
        cc.Wbuf = (const double *)bnr_uniform_ptr(m.Wbuf); cc.sz = (const double *)bnr_uniform_ptr(m.sz);
        cc.Psum = (double *)bnr_uniform_ptr(m.Psum); cc.counters = (long long *)bnr_uniform_ptr(m.counters);
        cc.seed = (unsigned long long)bnr_uniform_ptr((const void *)m.seed); cc.it = (unsigned)__builtin_amdgcn_readfirstlane((int)m.it); cc.pad = 0;
    }
    double *row = cc.row;
    const double *prev = cc.prev;
    const double tau2 = bnr_uniform_f64(row[ROW_TAU2]), tau = sqrt(tau2);
    double gam = 0.0, Snew = 1.0, W = 0.0;
    int cap = 0;
    if (act) {
        W = cc.Wbuf[e];
        const double Sp = prev[c0.o_S + e];
        gam = tau * (cc.sz[e] + Sp * sh[c * STR + SA + 33 * R + el32]) + W;
        if (slot == 0) row[c0.o_gamma + e] = gam;
    }
    {
        const double g = gam - W, chi = (g * g) / tau2, psi = prev[ROW_THETA];
        bnr_gig_ctx gc;
        gc.kind = 4;
        if (act) bnr_gig_setup(gc, 0.5, chi, psi);
        const bool loop = act && (gc.kind == 2 || gc.kind == 3);
        bool done = !loop;
        // Round 1: attempts 0 and 1 of every edge (lanes 0..31 / 32..63).  Later rounds: the few edges that are still open share the
        // wave's 64 lanes -- k = 64 / open attempts each, the contexts handed over through LDS -- so that the draws of a block end
        // after two rounds instead of after as many as its unluckiest edge needs (the first accepted attempt of the counter
        // sequence wins either way: the same draw).  One wave per chain: everything below is wave-local.
        uint32_t base = 0;
        {
            double v = 0.0;
            const bool ok = !done && bnr_gig_try(gc, cc.seed, cc.it, (uint32_t)e, (uint32_t)slot, v);
            s_acc[c][slot][el32] = ok ? 1 : 0;
            s_val[c][slot][el32] = v;
            bnr_wsync();
            if (!done) {
#pragma unroll
                for (int a = NSLOT - 1; a >= 0; --a) if (s_acc[c][a][el32]) { Snew = s_val[c][a][el32]; done = true; }   // lowest accepted attempt wins
            }
            bnr_wsync();
            base = NSLOT;
        }
        double *sx = sh + c * STR;                                  // the chain's a4 area: [open edge][22] contexts
        double *sv = &s_val[c][0][0];                               // 64 values
        int *spe = &s_acc[c][0][0];                                 // open edge -> its index in the block
        while (base < BNR_MAX_ATTEMPTS) {
            const unsigned long long pm = __ballot(lane < 32 && !done);
            const int npend = __popcll(pm);
            if (npend == 0) break;
            const int k = min(64 / npend, 32);
            const int rank = __popcll(pm & ((1ull << el32) - 1ull));
            if (lane < 32 && !done) { bnr_gig_ctx_put(gc, sx + rank * 22); spe[rank] = el32; }
            bnr_wsync();
            const int r = lane / k, sub = lane - r * k;
            bool ok = false;
            double v = 0.0;
            if (r < npend) {
                bnr_gig_ctx g2;
                bnr_gig_ctx_get(g2, sx + r * 22);
                ok = bnr_gig_try(g2, cc.seed, cc.it, (uint32_t)(e0 + spe[r]), base + (uint32_t)sub, v);
            }
            const unsigned long long om = __ballot(ok);
            sv[lane] = v;
            bnr_wsync();
            if (!done) {
                const unsigned long long mine = (om >> (rank * k)) & ((1ull << k) - 1ull);          // k <= 32
                if (mine) { Snew = sv[rank * k + __ffsll((long long)mine) - 1]; done = true; }
            }
            bnr_wsync();
            base += (uint32_t)k;
        }
        if (!done) { cap = 1; Snew = gc.alpha * gc.xm; }                      // attempt cap, as bnr_gig
        if (act && !loop) Snew = bnr_gig_degenerate(gc, cc.seed, chi, psi, cc.it, (uint32_t)e, &cap);
        if (act && slot == 0) row[c0.o_S + e] = Snew;
    }
    BNR_GSTAMP(3);
    if ((wave % WPC) || c >= nc) return;
    // partial sums of the block for chain c by its first wave (lanes 0..31 hold attempt slot 0 of the 32 edges), as in k_backproj
    double *ps = cc.Psum + (size_t)bid * (1 + 3 * R);
    double *st = sh + c * STR;                                  // the chain's a4 area is free now
    const double *sdr = sh + c * STR + SA;
    const double *lamp = prev + c0.o_lam;
    const double sd = sqrt(tau2 * Snew), lsd = log(sd) + 0.5 * log(2.0 * BNR_PI);
    if (lane < 32) {
        st[lane] = act ? Snew : 0.0;
        for (int r = 0; r < R; ++r) {
            double dr = sdr[r * 33 + lane];
            double lr = lamp[r];
#pragma unroll
            for (int q3 = 0; q3 < 3; ++q3) {
                double Wc = W + (bnr_lambda_value(q3) - lr) * dr;
                double zz = (gam - Wc) / sd;
                st[(1 + 3 * r + q3) * 33 + lane] = act ? (-0.5 * zz * zz - lsd) : 0.0;
            }
        }
    }
    bnr_wsync();
    for (int j = lane; j < 1 + 3 * R; j += 64) {
        double acc = 0.0;
#pragma unroll 8
        for (int e2 = 0; e2 < 32; ++e2) acc += st[j * 33 + e2];
        ps[j] = acc;
    }
    BNR_GSTAMP(4);
    if (cap && lane < 32) atomicAdd((unsigned long long *)&cc.counters[2], 1ull);
}


// ---- "linear" schedule: every stream of the sweep replays a LINEAR captured graph (linear graphs on different streams run side by side; graphs
// with forked branches do not, and their branches are mapped to queues in ways one cannot steer -- tools/graph_concurrency_probe.hip), and the
// streams meet through counters in device memory: a one-wave gate kernel in front of the consumer, a one-thread kernel behind the producer.
// Both are ordinary kernels of their streams, so the data itself travels by kernel boundaries (release at the producer's end, acquire at the
// consumer's start); the counter is written and read past the L2s (sc1).  lf[0] = sticky "a gate gave up" word.
// dbg (diagnostics, may be null): entry / exit times of the gates and setters, a ring over 16 sweeps: [((n & 15) 4 + part) 12 + 2 kind + {0, 1}]
__global__ void k_lin_set(unsigned long long *flag, const unsigned long long *cnt, int s, int mul, int add, unsigned long long *dbg, int part, int kind)
{
    if (threadIdx.x != 0) return;
    if (dbg) dbg[(((cnt[0] + s) & 15) * 4 + part) * 12 + 2 * kind] = __builtin_amdgcn_s_memrealtime();
    __hip_atomic_store(flag, (cnt[0] + (unsigned long long)s) * (unsigned long long)mul + (unsigned long long)add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_lin_add(unsigned long long *cnt, int by) { if (threadIdx.x == 0) cnt[0] += (unsigned long long)by; }
__global__ __launch_bounds__(64) void k_lin_gate(const unsigned long long *flag, const unsigned long long *cnt, int s, int mul, int add, int spin_us, unsigned long long *sticky, long long *err_counter,
                                                  unsigned long long *dbg, int part, int kind)
{
    if (threadIdx.x != 0) return;
    unsigned long long *d = dbg ? dbg + (((cnt[0] + s) & 15) * 4 + part) * 12 + 2 * kind : nullptr;
    if (d) d[0] = __builtin_amdgcn_s_memrealtime();
    const unsigned long long need = (cnt[0] + (unsigned long long)s) * (unsigned long long)mul + (unsigned long long)add;
    unsigned long long have = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (have >= need) { if (d) d[1] = __builtin_amdgcn_s_memrealtime(); return; }
    if (__hip_atomic_load(sticky, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ull) {
        __builtin_amdgcn_s_setprio(3);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), lim = 100ull * (unsigned long long)spin_us;
        do {
            __builtin_amdgcn_s_sleep(4);
            have = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while (have < need && __builtin_amdgcn_s_memrealtime() - t0 < lim);
        if (have >= need) { if (d) d[1] = __builtin_amdgcn_s_memrealtime(); return; }
        __hip_atomic_store(sticky, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    atomicAdd((unsigned long long *)err_counter, 1ull);           // counters[8]: "stream ordering violated" -- the run fails loudly
}

__global__ void k_stamp(unsigned long long *dbg, int slot) { if (threadIdx.x == 0) dbg[slot] = __builtin_amdgcn_s_memrealtime(); }

__global__ void k_zero_words(unsigned *p, int n) { for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0u; }

// ----------------------------------------------------------------------------------------- data-flow factorization (n_pad <= 512)
// k_chol_df: the SAME factorization of E = [G + I ; I] -> [L ; L^-T] as k_chol_step x nbk -- every element sees the K-slice partials summed in
// k_gram_reduce's order, then the panels' rank-32 updates in ascending order (eight MFMA k-steps each), then the same column sweep: bitwise the
// same E -- as ONE launch in which the panel steps follow each other through flags instead of kernel boundaries.  What a boundary costs inside a
// captured graph: ~2.3 us from one launch to the next + a cold round trip to memory for the fragments (1.35 us), x 16 steps = 58 of the 112 us of
// one chain's factorization (141 us for a group of 8); measured ceiling of what a faster factorization buys: tools/r4_exp3.py.
//   ONE CHAIN PER XCD.  The workgroups of the launch are dealt round-robin to the XCDs (workgroup i -> XCD i % 8): chain c = i % 8, local index
//   w = i / 8.  Everything the workgroups of a chain hand to each other travels through THAT XCD's L2 -- plain stores (the L1 is write-through) +
//   s_waitcnt vmcnt(0), a flag written by an atomic the L2 executes (workgroup scope: no sc bits), fragments and flags read past the reader's L1
//   (sc1 loads): 0.95 us per hand-over, no cache maintenance at all (tools/xcd_sync_probe.hip; through memory across XCDs: 1.4 us + cold loads).
//   Every workgroup reports its XCD (dfctl[1]); k_solve_w checks that a chain saw exactly one and raises "stream ordering violated" otherwise.
//   OWNER COMPUTES.  The nbk blocks of block column j that are swept (matrix rows j+1.., identity rows 0..j; the diagonal block itself is needed by
//   nobody afterwards) are "slots" 0..nbk-1; workgroup w = (slot = w % 16, group = w / 16) owns slot `slot` of the columns group, group + 4,
//   group + 8, group + 12: at most four 32 x 32 blocks, kept as MFMA accumulator tiles in the registers of its four waves from their first touch
//   (the K-slice partials) to their sweep, together with an own copy of each column's diagonal block (it takes the same panels' updates from
//   the same fragments) -- the trailing matrix never goes through memory.  At step q the owners of column q sweep [D ; own] (bnr_panel_sweep),
//   store the swept block to E and raise its flag; everybody who still holds a block of a later column waits for the two blocks of panel q it
//   needs and applies the update.  Nobody waits for anything but panels of lower index: with all workgroups resident (64 per chain, two per CU:
//   512 of 256 threads at most) the launch always ends; a wait that outlasts BNR_DF_TIMEOUT_US gives up and raises the status.
//   grid = 8 x 64 workgroups of 256 threads; lockstep groups of up to 8 chains (more: k_chol_step).
#ifndef BNR_DF_NG
#define BNR_DF_NG 4                    // column groups: workgroup (slot, group) owns slot `slot` of the columns group, group + NG, ...: 16 / NG blocks.  4: 64 workgroups of 4 blocks per
#endif                                 // chain, two per CU (171 VGPRs each: nothing of the scalar branch fits beside them); 2: 32 workgroups of 8 blocks, ONE per CU (95 KB of LDS), one wave per SIMD
#ifndef BNR_DF_PRIO
#define BNR_DF_PRIO 1                  // 1: s_setprio 3 throughout (like the other kernels of the critical chain); 2: priority 0 while polling; 0: never raised
#endif
#ifndef BNR_DF_SLEEP
#define BNR_DF_SLEEP 1
#endif
#define BNR_DF_KB (16 / BNR_DF_NG)
#define BNR_DF_WG (16 * BNR_DF_NG)
#define BNR_DF_LDS (BNR_DF_KB * BNR_NB * BNR_LP * sizeof(double))
#define BNR_DF_TIMEOUT_US 200000
__device__ __forceinline__ double bnr_ld_l2(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// the wave's 16 x 16 tile (columns mt, rows nt) of a 32 x 32 block kept in LDS (row + BNR_LP * column, like the sweep's staging area)
__device__ __forceinline__ bnr_d4 bnr_lds_tile_get(const double *sX, int mt, int nt, int ln, int lq)
{
    bnr_d4 c;
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = sX[(nt * 16 + ln) + BNR_LP * (mt * 16 + lq + 4 * r)];
    return c;
}
__device__ __forceinline__ void bnr_lds_tile_put(double *sX, int mt, int nt, int ln, int lq, const bnr_d4 &c)
{
#pragma unroll
    for (int r = 0; r < 4; ++r) sX[(nt * 16 + ln) + BNR_LP * (mt * 16 + lq + 4 * r)] = c[r];
}
// panel q into one owned block (register tile cB) and into the LDS copy of its column's diagonal block.  The two blocks of the panel it takes --
// L[j, q] (column side of both products, row side of the diagonal copy's) and L[br, q] -- are brought from the L2 ONCE per workgroup into the
// sweep's staging area (free between sweeps) and read from there as MFMA fragments: fetched per wave straight from the L2, every wave loaded
// three 4 KiB fragments = 48 KiB per update for 16 KiB of data, and with ~200 live blocks per chain and step the XCD's L2 bandwidth set the pace
// (9.6 MB per step: 10 us per step instead of 5).
// An identity row r has no block in the panels q < r: what is read there is the part of E below the block diagonal of Y, which no kernel ever
// writes (zeros since the allocation) -- the update then subtracts exact zeros, as k_chol_step's does for the same blocks.  (Unconditional on
// purpose: with a branch around the second group of MFMAs the group instantiation of this kernel computed wrong diagonal copies -- the
// accumulators live in AGPRs here and the taken branch reached their v_accvgpr_read too early.)
__device__ __forceinline__ void bnr_df_update(bnr_panel_lds &sh, const double *E, size_t ld, int q, int j, int br, bnr_d4 &cB, double *sDj, int mt, int nt, int tid)
{
    const int lane = tid & 63, ln = lane & 15, lk = lane >> 4;
    const size_t kc = ld * (size_t)(q * BNR_NB);
    // thread t brings elements (row t % 32, columns t / 32 + 8 i) of both blocks
    const int sr = tid & 31, sc = tid >> 5;
    const double *pj = E + (size_t)(j * BNR_NB + sr) + kc + ld * (size_t)sc, *pb = E + (size_t)(br * BNR_NB + sr) + kc + ld * (size_t)sc;
    double vj[4], vb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { vj[i] = bnr_ld_l2(pj + ld * (size_t)(8 * i)); vb[i] = bnr_ld_l2(pb + ld * (size_t)(8 * i)); }
    __syncthreads();                                      // (whoever still reads the staging area of the previous update is done)
#pragma unroll
    for (int i = 0; i < 4; ++i) { sh.sD[sr + BNR_LP * (sc + 8 * i)] = vj[i]; sh.sB[sr + BNR_LP * (sc + 8 * i)] = vb[i]; }
    __syncthreads();
    double av[8], dv[8], bv[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        av[ks] = sh.sD[(mt * 16 + ln) + BNR_LP * (4 * ks + lk)]; dv[ks] = sh.sD[(nt * 16 + ln) + BNR_LP * (4 * ks + lk)]; bv[ks] = sh.sB[(nt * 16 + ln) + BNR_LP * (4 * ks + lk)];
    }
    bnr_d4 cD = bnr_lds_tile_get(sDj, mt, nt, ln, lk);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) cD = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[ks], dv[ks], cD, 0, 0, 0);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) cB = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[ks], bv[ks], cB, 0, 0, 0);
    bnr_lds_tile_put(sDj, mt, nt, ln, lk, cD);
    bnr_wsync();                                          // (a ds_write is not ordered before later ds_reads of the same wave without the wait)
}
// two flags at once (one round trip per poll instead of two in a row)
__device__ __forceinline__ bool bnr_df_wait2(const unsigned int *f0, const unsigned int *f1, unsigned int epoch)
{
    unsigned a = __hip_atomic_load(f0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b = __hip_atomic_load(f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a == epoch && b == epoch) return true;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#if BNR_DF_PRIO == 2
    __builtin_amdgcn_s_setprio(0);                       // a polling wave must not outrank the neighbours' working waves
#endif
    bool ok = false;
    for (;;) {
        __builtin_amdgcn_s_sleep(BNR_DF_SLEEP);
        a = __hip_atomic_load(f0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); b = __hip_atomic_load(f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a == epoch && b == epoch) { ok = true; break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > 100ull * BNR_DF_TIMEOUT_US) break;
    }
#if BNR_DF_PRIO == 2
    __builtin_amdgcn_s_setprio(3);
#endif
    return ok;
}
template <class SRC>
__global__ __launch_bounds__(256) void k_chol_df(const SRC chain_src, int s, int nchains)
{
#if BNR_DF_PRIO
    BNR_CRITICAL_PATH();
#endif
    const int chain = blockIdx.x & 7, w = blockIdx.x >> 3;
    if (chain >= nchains) return;
    const bnr_dev &cd = chain_src.at(chain);
    __shared__ bnr_panel_lds sh;
    extern __shared__ double sDc_[];                      // my copies of the diagonal blocks of my columns: BNR_DF_KB x (BNR_NB * BNR_LP)
    double (*sDc)[BNR_NB * BNR_LP] = (double (*)[BNR_NB * BNR_LP])sDc_;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nbk = cd.n_pad / BNR_NB;
    const int slot = w & 15, grp = w >> 4;
    if (slot >= nbk) return;
    const size_t ld = bnr_ldE(cd.n_pad);
    const int mt = wave >> 1, nt = wave & 1, ln = lane & 15, lq = lane >> 4;
    double *E = cd.E;
    const unsigned int epoch = cd.dfctl[0];
    unsigned int *flags = cd.dfctl + 32;
    if (tid == 0) {
        unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
        __hip_atomic_fetch_or(&cd.dfctl[1], 1u << (v & 7u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef BNR_STAMPS
        if (chain == 0) { unsigned h; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h)); cd.dbg[1000 + w] = __builtin_amdgcn_s_memrealtime(); cd.dbg[1200 + w] = h; }
#endif
    }
    // my blocks: column j_k = grp + NG k; block row: matrix row j + 1 + slot while there are any, then identity row slot - (nbk - 1 - j)
    bnr_d4 cB[BNR_DF_KB];
    int brow[BNR_DF_KB];                                         // block row in E (matrix rows 0 .. nbk-1, identity rows nbk ..); -1: no block
#pragma unroll
    for (int k = 0; k < BNR_DF_KB; ++k) {
        const int j = grp + BNR_DF_NG * k;
        brow[k] = -1;
        cB[k] = bnr_d4{0.0, 0.0, 0.0, 0.0};
        if (j < nbk) {
            const int nmat = nbk - 1 - j;
            bnr_d4 t1[1][1];
            if (slot < nmat) {
                brow[k] = j + 1 + slot;
                bnr_gsum_frag<1>(cd, brow[k], j, mt, nt, ln, lq, t1, false);
                cB[k] = t1[0][0];
            } else {
                brow[k] = nbk + (slot - nmat);
                if (slot - nmat == j) {                  // the identity block itself
#pragma unroll
                    for (int r = 0; r < 4; ++r) cB[k][r] = (nt * 16 + ln == mt * 16 + lq + 4 * r) ? 1.0 : 0.0;
                }
            }
            bnr_gsum_frag<1>(cd, j, j, mt, nt, ln, lq, t1, false);
            bnr_lds_tile_put(sDc[k], mt, nt, ln, lq, t1[0][0]);      // (every wave reads back only the tile it wrote: no barrier needed)
            bnr_wsync();
        }
    }
    int bad = 0, late = 0;
#ifdef BNR_STAMPS
#define BNR_DSTAMP(i) do { if (slot == 0 && chain == 0 && tid == 0) cd.dbg[400 + 8 * q + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define BNR_DSTAMP(i) do { } while (0)
#endif
    // Order of work per panel p (p = -1: nothing to apply yet): FIRST panel p into my block of column p + 1 (its owner is the next sweeper), then,
    // if I own one, that block's sweep at once; only then panel p into my blocks of the columns behind.
    for (int p = -1; p < nbk; ++p) {
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (p >= 0) {
#pragma unroll
                for (int k = 0; k < BNR_DF_KB; ++k) {
                    const int j = grp + BNR_DF_NG * k;
                    if (j <= p || j >= nbk || brow[k] < 0) continue;
                    if ((pass == 0) != (j == p + 1)) continue;
                    // L[j, p] and my block row of panel p (an identity row r has one only from p = r on: before that the update reads zeros)
                    const bool has_b = brow[k] < nbk || brow[k] - nbk <= p;
                    if (!bnr_df_wait2(&flags[32 * p + j], &flags[32 * p + (has_b ? brow[k] : j)], epoch)) late = 1;
                    asm volatile("" ::: "memory");                                          // (the fragment loads stay behind the flags they wait for)
                    bnr_df_update(sh, E, ld, p, j, brow[k], cB[k], sDc[k], mt, nt, tid);
                    __builtin_amdgcn_sched_barrier(0);                                      // (one block at a time: the fragments of several updates would be live at once)
                }
            }
            const int q = p + 1;                                                            // the column that is complete now
            if (pass == 0 && q < nbk && (q % BNR_DF_NG) == grp) {
#ifdef BNR_STAMPS
                if (slot == 0 && chain == 0 && tid == 0) { cd.dbg[400 + 8 * q + 4] = __builtin_amdgcn_s_memrealtime(); }
#endif
                const int k = q / BNR_DF_NG;
                bnr_d4 sB = cB[0];
                int br = brow[0];
#pragma unroll
                for (int kk = 1; kk < BNR_DF_KB; ++kk) if (k == kk) { sB = cB[kk]; br = brow[kk]; }
                if (br >= 0) {
                    double *dst = E + (size_t)(br * BNR_NB) + ld * (size_t)(q * BNR_NB);
                    const bnr_d4 sD = bnr_lds_tile_get(sDc[k], mt, nt, ln, lq);
                    __syncthreads();                                                       // (the staging area may still be read by the last update's fragments)
                    BNR_DSTAMP(1);
                    bad |= bnr_panel_sweep(sh, sD, sB, tid, dst, ld);
                    BNR_DSTAMP(2);
                    if (wave == 0) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the swept block is in the L2 ...
                        if (lane == 0) __hip_atomic_exchange(&flags[32 * q + br], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ... then its flag (an atomic the L2 executes)
                    }
                    __syncthreads();                                                       // (the staging area of the sweep is free again)
                    BNR_DSTAMP(3);
                }
            }
        }
    }
#ifdef BNR_STAMPS
    if (chain == 0 && tid == 0) cd.dbg[1100 + w] = __builtin_amdgcn_s_memrealtime();
#endif
    if (bad && tid == 0 && w == 0) { atomicAdd((unsigned long long *)&cd.counters[3], 1ull); atomicAdd((unsigned long long *)&cd.counters[7], 1ull); }
    if (late && lane == 0) atomicAdd((unsigned long long *)&cd.counters[8], 1ull);       // a hand-over never came: the run fails loudly ("stream ordering violated")
}

// ----------------------------------------------------------------------------------------- small problems: one launch, one workgroup per chain
// k_chol_small<NBK> (n_pad = 32 NBK, NBK = 2 or 4, i.e. n <= 128): the SAME factorization as k_chol_step x NBK -- every element sees the K-slice
// partials summed in k_gram_reduce's order, then the panels' rank-32 updates in ascending order (eight MFMA k-steps each), then the same column
// sweep: bitwise the same E -- by ONE workgroup of NBK teams of 256 threads.  At these sizes a panel step is 3.3 us of sweep + 0.9 us of update, and
// the launch-per-panel version pays 2.3 us of launch gap and a cold round trip on top, NBK times, on what is half of the sweep's critical chain
// (tools/r4_exp3.py with BNR_SHAPE=70,19,5: 87.0 us per sweep, 72.4 with the panel steps 1.. skipped).
// MEASURED SLOWER (tools/r4_small.py, notes r4 H): bitwise equal, but n = 70: 108 us per sweep against 84, n = 128: 134 against 95 -- replayed from the
// captured graph, the scalar branch's kernels are not dispatched on the XCD where this 25 us kernel runs until it ends (the same effect that keeps
// k_node away from k_chol_df), so the two branches of the sweep run one after the other.  n_pad = 64: a draw (71.4 / 70.9).
//   Team t is the panel workgroup of slot t + 1 (k_chol_step's blockIdx.y): at step p it owns block row rho of block column p, touches it for the first
//   time there (K-slice partials of G + I, or the identity block of Y), applies the panels 0 .. p-1 to it and to its own copy of the diagonal block
//   (left-looking: with at most three panels behind there is no trailing update worth a workgroup of its own), and sweeps [D ; own] (bnr_panel_sweep,
//   whose barriers are workgroup-wide: all teams run it in lockstep).  The swept blocks go to E; the next step's fragment loads follow a barrier
//   of the same workgroup (one CU, one L1: no cache maintenance).  No reduction pass, no stamps: the kernel follows the Gram on its stream.
//   grid = chains; block = 256 NBK threads; dynamic LDS = NBK panel areas.
template <class SRC, int NBK>
__global__ __launch_bounds__(256 * NBK) void k_chol_small(const SRC chain_src, int s)
{
    BNR_CRITICAL_PATH();
    (void)s;
    const bnr_dev &cd = chain_src.get_x();
    extern __shared__ double sh_small_[];
    const int team = threadIdx.x >> 8, tid = threadIdx.x & 255, wave = tid >> 6, lane = tid & 63;
    bnr_panel_lds &sh = ((bnr_panel_lds *)sh_small_)[team];
    const size_t ld = bnr_ldE(cd.n_pad);
    const int mt = wave >> 1, nt = wave & 1, ln = lane & 15, lq = lane >> 4;
    double *E = cd.E;
    for (int p = 0; p < NBK; ++p) {
        const int b = team + 1, pc = p * BNR_NB;
        const int rho = b < NBK - p ? p + b : NBK + (b - (NBK - p));     // matrix rows p+1 .. NBK-1, then identity rows 0 .. p
        bnr_d4 cD, cB, t1[1][1];
        bnr_gsum_frag<1, 2>(cd, p, p, mt, nt, ln, lq, t1, false);
        cD = t1[0][0];
        int q0 = 0;                                                       // the first panel that has a block in my block row
        if (rho < NBK) { bnr_gsum_frag<1, 2>(cd, rho, p, mt, nt, ln, lq, t1, false); cB = t1[0][0]; }
        else {
            q0 = rho - NBK;                                               // identity row r: I in block column r, nothing before it
#pragma unroll
            for (int r = 0; r < 4; ++r) cB[r] = (q0 == p && nt * 16 + ln == mt * 16 + lq + 4 * r) ? 1.0 : 0.0;
        }
        for (int q = 0; q < p; ++q) {
            const size_t kc = (size_t)q * BNR_NB;
            const double *colrows = E + (size_t)(pc + mt * 16) + ld * kc;
            cD = bnr_tile_update(colrows, E + (size_t)(pc + nt * 16) + ld * kc, ld, lane, cD);
            if (q >= q0) cB = bnr_tile_update(colrows, E + (size_t)(rho * BNR_NB + nt * 16) + ld * kc, ld, lane, cB);
        }
        const int bad = bnr_panel_sweep(sh, cD, cB, tid, E + (size_t)(rho * BNR_NB) + ld * (size_t)pc, ld);
        if (bad && threadIdx.x == 0) { atomicAdd((unsigned long long *)&cd.counters[3], 1ull); atomicAdd((unsigned long long *)&cd.counters[7], 1ull); }
        __syncthreads();                                                  // the swept blocks of panel p are in memory before anybody's next fragment loads
    }
}
