// annp_hip.hip -- C ABI of libannp_hip.so (see include/annp_hip.h).
//
// Host driver for the gfx950 kernels.  Behaviour mirrors what a LAMMPS GPU-package
// pair style expects from lib/gpu (reference: annp-gpu-lammps/fe_v2/lib/lal_annp.cpp
// init 41-216, compute 259-370, loop_annp 517-607) without reproducing its design:
// no Geryon, no atom-chunk loop, no materialised dG, no single-block force update.
// Per call: descriptor pass -> network pass (FP64 MFMA) -> force pass, all on one
// stream, buffers owned by the handle.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <sched.h>
#include <rccl/rccl.h>       // types only: the library is opened at run time (annp_hip_comm_init), there is no link-time dependency

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/annp_hip.h"
#include "annp_common.hpp"
#include "fe_sh_kernels.hpp"
#include "fe_shf_kernels.hpp"
#include "mlp_kernels.hpp"
#include "neigh_kernels.hpp"
#include "ni_kernels.hpp"
#include "anna_kernels.hpp"
#include "step_kernels.hpp"
#include "replan_kernels.hpp"

using namespace annp;

namespace {

thread_local std::string g_init_error;     // text of the calling thread's last failed annp_hip_init (handles are one thread each)

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;   // elements
};

}  // namespace

// A few host threads that copy memory (the re-pack of LAMMPS' paged neighbour list into pinned staging).  They never
// call HIP: every runtime call stays on the caller's thread.
class CopyPool {
   public:
    explicit CopyPool(int n)
    {
        for (int t = 0; t < n; t++) workers_.emplace_back([this, t] { loop(t); });
    }
    ~CopyPool()
    {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; gen_++; }
        cv_.notify_all();
        for (std::thread &w : workers_) w.join();
    }
    int size() const { return (int)workers_.size() + 1; }
    // job(part, nparts) on every worker and on the caller; returns when all parts are done
    void run(const std::function<void(int, int)> &job)
    {
        { std::lock_guard<std::mutex> g(m_); job_ = &job; left_ = (int)workers_.size(); gen_++; }
        cv_.notify_all();
        job((int)workers_.size(), size());
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return left_ == 0; });
        job_ = nullptr;
    }

   private:
    void loop(int t)
    {
        unsigned long seen = 0;
        for (;;) {
            const std::function<void(int, int)> *job;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                job = job_;
            }
            (*job)(t, size());
            { std::lock_guard<std::mutex> g(m_); if (--left_ == 0) done_.notify_one(); }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(int, int)> *job_ = nullptr;
    unsigned long gen_ = 0;
    int left_ = 0;
    bool stop_ = false;
};

constexpr int ANNP_NFLAGS = 8;          // device flag words of an evaluation (annp_hip_handle::d_flags)

struct annp_hip_handle {
    int device = 0;
    hipStream_t stream = nullptr;       // used by the host-pointer entry points
    std::string err;
    // parameters (copied at init)
    int descriptor = 0, ntypes = 1, ntl = 0, nhl = 0, nnod = 0, nsf = 0, npsf = 0, ntsf = 0, nl = 0;
    int nsf_dev = 0;                    // features in the device layout (Chebyshev: always 9 + 19 slots, unused ones carry zero weights)
    int ni_compat = 0;
    bool ni_no_fixup = false;           // ANNP_HIP_NI_FIXUP=0: no queue behind the Behler records (overflow is an error again)
    bool ni_no_pairs = false;           // ANNP_HIP_NI_PAIRS=0: the Behler force pass finds its pairs itself (no lists through memory)
    bool full_list = false;             // ANNP_HIP_FULL_LIST=1: library-built lists are cut where the caller says (list_cutoff)
    // pair_style anna_adp
    int nout = 1;
    double e_base = 0.0, gp[17] = {0};
    double *d_net = nullptr;            // network image for annp_anna_adp (layer 0 in the device feature layout)
    int net_doubles = 0;
    int ni_cap = 24;                    // Behler kernels: record capacity per atom for the next evaluation
    bool ni_primed = false;             // ... has been sized from a completed evaluation (else the next one sizes it synchronously)
    int fe_cap = 0;                     // Chebyshev force pass: record capacity for the next evaluation (0 = not sized yet)
    int cap_last = 0;                   // capacity the last force pass ran with
    int sh_cap = 128;                   // Chebyshev descriptor pass (annp_fe_desc_sh): state slots per atom for the next evaluation.  The first one gets 128, not
                                        // SH_CAP_MAX: up to there the pass keeps three workgroups per CU (160: two), and whoever has more goes through the fix-up launch once
    bool fe_desc_pairs = false;         // ANNP_HIP_FE_DESC=pairs: the pair-loop descriptor kernel (annp_fe_desc) for every atom
    bool fe_force_pairs = false;        // ANNP_HIP_FE_FORCE=pairs: the pair-loop force kernel (annp_fe_force) for every atom
    int shf_places_by_number = 0;       // ANNP_HIP_SHF_PLACES=number: the one-slot wave of a group is its fourth wave, wherever it sits (developer A/B switch)
    FILE *notice = nullptr;             // annp_hip_set_notice: where a change of kernel path is announced (once per change)
    bool fe_dense_said = false;
    int sh_cap_used = 0;                // state slots the last annp_fe_desc_sh launch had (= atoms with moments have at most that many neighbours)
    bool fe_dense = false;              // most atoms have more neighbours than the moment kernels take (SH_CAP_MAX = 160): the pair-loop kernels for all
    bool fe_last_sh = false;            // the last Chebyshev evaluation ran the moment kernels
    bool flags_sh = false; int flags_inum = 0;      // ... and the evaluation the pending flag words belong to (several can be in flight)
    bool shf_scattered = false, shf_scattered_said = false;       // the caller's atoms are in no spatial order (annp_fe_force_sh's force table)
    int fe_last_inum = 0;
    int sh_wpb = 0;                     // waves per workgroup of annp_fe_desc_sh (ANNP_HIP_SH_WPB; 0 = chosen per launch)
    bool rp_images_by_dimension = false;    // ANNP_HIP_REPLAN_IMAGES=dims: annp_hip_replan_images waits for every dimension's count (rounds 4-5; A/B switch)
    bool sh_group = false;              // ANNP_HIP_SH_TAIL=group: the descriptor pass changes basis group by group out of LDS where it can (round 6, measured 2 % slower; developer A/B switch)
    int flagact[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // up to max(MLP_MAXL, ANNA_MAXL) weight layers
    static_assert(MLP_MAXL <= 8 && ANNA_MAXL <= 8, "flagact holds 8 layers");
    double e_scale = 0, e_shift = 0, e_atom = 0, cut = 0, cutsq = 0;
    double *d_norm = nullptr;           // nmul | nsub | nden, ANNP_GPAD each
    double *d_mlp_img = nullptr;        // network pass: MFMA operand images (weights, biases, coefmat), one per element, mlp_build_image
    size_t img_stride = 0;              // doubles per element image
    int nelem = 1;
    bool multi = false;                 // several elements or an unmapped type: the kernels need atom types
    unsigned active = ~0u;              // bit t: type t is mapped to an element
    int *d_map = nullptr;               // device copy of map[0..ntypes]
    double *d_sym = nullptr;            // BEHLER: function tables (ni_kernels.hpp, "per-function tables")
    int *d_isym = nullptr;
    unsigned long long ni_rad_em = 0;   // NiArgs::rad_em
    NiShape ni_shape = {0, 0, 0};       // {lambda} x {eta} x {zeta} product shape of the angular set (0 = none)
    double ni_lam[4] = {0, 0, 0, 0}, ni_eta[4] = {0, 0, 0, 0};   // its distinct lambda / eta values in visit order
    std::vector<double> sym_rad, sym_ang;
    // work buffers
    DevBuf<double> fscratch;            // forces of one evaluation by themselves, when the global virial is taken as sum x (x) f (annp_fdotr_add)
    bool virial_tally = false;          // ANNP_HIP_VIRIAL=tally: the pairwise tally inside the force kernels instead (what per-atom virials always use)
    DevBuf<double> G, coef, x, f, eatom, vatom, mom;        // mom: moments of the neighbourhoods, descriptor pass -> force pass (fe_sh_kernels.hpp)
    DevBuf<int> type, ilist, numneigh, neigh, ncount, ni_nbr, ni_npair, ni_fix_nbr, ovf, ovf_desc, fe_nbrs;
    DevBuf<unsigned short> ni_pairs;    // Behler: in-range (j,k) pairs per atom, descriptor pass -> force pass
    DevBuf<long long> first;
    // re-planning (replan_kernels.hpp): flags / scan positions / block sums of its stream compactions, their totals
    DevBuf<int> rp_flag, rp_cnt;
    DevBuf<long long> rp_pos, rp_bs;
    long long *rp_tot = nullptr, *rp_tot_h = nullptr;
    double *d_vslots = nullptr;         // [ANNP_VSLOTS][8]: where the kernels tally the global virial (annp_common.hpp), folded per evaluation
    double *d_scalars = nullptr;        // [0] energy, [1..6] virial
    int *d_flags = nullptr;             // [0] capacity error: max n of the atoms that were skipped (stays set until the host has
                                        //     seen it), [1] max in-cutoff n, [2] length of the force fix-up queue, [3] of the descriptor fix-up queue,
                                        //     [4] contributions annp_fe_force_sh's force tables had no bucket for; [1..] per evaluation
    int *h_flags = nullptr;             // pinned mirror, copied back behind every evaluation
    hipEvent_t ev_flags = nullptr;      // ... that copy has landed (and, behind it on the same side stream, words [1..] are clear again)
    hipStream_t stream_flags = nullptr; // the copy and the clearing run beside the caller's stream, not in it (round 5: at 128 000 atoms the
    hipEvent_t ev_tail = nullptr;       // copy held the next step's first kernel back by 33 us, the clearing cost two fill kernels)
    int *fw = nullptr;                  // the per-evaluation words of the evaluation being issued: d_flags + 8 or + 16, turn about (fw[1..7]; word 0 of
    int flags_par = 0;                  // d_flags is the sticky one) -- an evaluation never waits for the copy + clear behind the one before it
    hipEvent_t ev_set[2] = {nullptr, nullptr};   // the side stream has cleared that set again
    bool set_used[2] = {false, false};
    bool flags_dirty = false;           // an evaluation set out and never reached its tail (an error on the way): clear the words in-stream
    int sticky_rc = 0;                  // error found in a landed copy, returned by the next call on the handle
    bool reset_err = false;             // d_flags[0] was seen non-zero: clear it before the next evaluation
    int info[4] = {0, 0, 0, 0};         // annp_hip_eval_info
    double *h_scalars = nullptr;        // pinned mirror
    // neighbour list built on device
    NeighBuild nb;
    // host-list cache
    // pinned, persistent staging for what host_finish brings back (a fresh pageable vector per call costs its
    // page faults and a second copy inside the runtime: ~3 ms per 1 M atoms)
    double *pin_f = nullptr, *pin_e = nullptr, *pin_v = nullptr, *pin_x = nullptr;
    size_t pin_f_cap = 0, pin_e_cap = 0, pin_v_cap = 0, pin_x_cap = 0;
    // the caller's x and f, page-locked in place (host_register)
    struct HostReg { const void *ptr = nullptr; size_t bytes = 0; bool ok = false; };
    HostReg reg_x, reg_f;
    bool use_register = true;           // ANNP_HIP_REGISTER=0 turns it off (pinned staging + host folds instead)
    hipStream_t stream2 = nullptr;      // uploads that overlap the first passes of an evaluation
    // RCCL communicator for the halo wire (annp_hip_comm_*): one rank per handle
    ncclComm_t comm = nullptr;
    int comm_world = 0, comm_rank = -1;
    hipEvent_t ev_f_up = nullptr;
    hipEvent_t pre_force_wait = nullptr;    // set by the host path: the force pass must not start before this event
    // host-list upload (annp_hip_compute, ago == 0): CSR headers and the rows go through pinned staging; the rows in
    // chunks, packed by a few threads while the previous chunk is on the wire
    long long *pin_first = nullptr;
    int *pin_num = nullptr;
    size_t pin_hdr_cap = 0;             // atoms
    static constexpr int kListBufs = 3;
    static constexpr int kListParts = 8;        // runs of chunks a host list is evaluated in while it is uploaded (annp_hip_compute, ago == 0)
    hipEvent_t ev_part[kListParts] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int list_parts = 4;                         // ANNP_HIP_LIST_PARTS (1 = upload first, evaluate afterwards: rounds 1-5)
    int list_pipe_min = 1 << 16;                // ... for lists of at least that many atoms (ANNP_HIP_LIST_PIPE_MIN)
    size_t list_chunk = kListChunk;             // ints per chunk of the upload (ANNP_HIP_LIST_CHUNK: smaller chunks let a test cut a small list into runs)
    static constexpr size_t kListChunk = (size_t)8 << 20;      // ints per staging buffer (32 MB)
    int *pin_list[kListBufs] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_list[kListBufs] = {nullptr, nullptr, nullptr};
    CopyPool *pool = nullptr;
    int list_max = 0;                   // max numneigh of the uploaded list
    bool list_valid = false;
    size_t bytes = 0;
    // timing
    bool timing = false;
    static constexpr int kRing = 64;      // evaluations kept for annp_hip_timing_stats
    std::vector<hipEvent_t> evring;       // kRing x 4 events, created on first enable
    hipEvent_t *ev = nullptr;             // the four events of the evaluation being enqueued
    long long ev_count = 0;               // evaluations recorded since timing was enabled
    bool flags_pending = false;         // a copy of d_flags into h_flags is in flight or not yet looked at
    bool mlp_attr_done = false;
    int mlp_blocks_per_cu = 0;          // resident workgroups of the network kernel per CU (occupancy query, once)
};

namespace {

int fail(annp_hip_handle *h, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_init_error = buf;
    return code;
}

#define HIP_TRY(h, call)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(h, e_ == hipErrorOutOfMemory ? ANNP_HIP_ENOMEM : ANNP_HIP_EDEVICE,       \
                        "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

template <typename T>
int ensure(annp_hip_handle *h, DevBuf<T> &b, size_t n, bool zeroed = false)
{
    if (n <= b.cap) return 0;
    size_t want = n + n / 8 + 64;
    if (b.p) { h->bytes -= b.cap * sizeof(T); (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
    HIP_TRY(h, hipMalloc((void **)&b.p, want * sizeof(T)));
    if (zeroed) {       // (only when the buffer grows)
        // the work streams do not synchronise with the null stream (hipStreamNonBlocking): the zeros are in place before anything of the
        // evaluation is enqueued, whatever hipMemset itself waits for
        HIP_TRY(h, hipMemset(b.p, 0, want * sizeof(T)));
        HIP_TRY(h, hipStreamSynchronize(nullptr));
    }
    b.cap = want;
    h->bytes += want * sizeof(T);
    return 0;
}

template <typename T>
void release(annp_hip_handle *h, DevBuf<T> &b)
{
    if (b.p) { (void)hipFree(b.p); h->bytes -= b.cap * sizeof(T); }
    b.p = nullptr; b.cap = 0;
}

// ---- network pass dispatch --------------------------------------------------------
template <int KS0, int MT, int NL>
int launch_mlp(annp_hip_handle *h, const MlpArgs &a, hipStream_t s)
{
    const size_t lds = mlp_lds_bytes<KS0, MT, NL>();
    if (!h->mlp_attr_done) {      // per handle: the attribute is per device, and handles may sit on different GPUs
        HIP_TRY(h, hipFuncSetAttribute((const void *)annp_mlp_mfma<KS0, MT, NL>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        h->mlp_attr_done = true;
    }
    const int ntiles = (a.inum + 15) / 16;
    int blocks = (ntiles + MLP_WAVES_PER_BLOCK - 1) / MLP_WAVES_PER_BLOCK;
    // waves walk over the tiles: one resident round of workgroups (as many per CU as the operand image in LDS and the
    // 32-wave limit allow).  More than that only queues behind it.
    if (h->mlp_blocks_per_cu == 0) {        // what really fits: LDS, registers and the 32-wave limit together
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)annp_mlp_mfma<KS0, MT, NL>, 64 * MLP_WAVES_PER_BLOCK, lds) != hipSuccess || nb < 1) {
            (void)hipGetLastError();
            nb = (int)std::max<size_t>(1, std::min<size_t>(32 / MLP_WAVES_PER_BLOCK, (160 * 1024) / lds));
        }
        h->mlp_blocks_per_cu = nb;
    }
    blocks = std::max(1, std::min(blocks, 256 * h->mlp_blocks_per_cu));
    hipLaunchKernelGGL((annp_mlp_mfma<KS0, MT, NL>), dim3(blocks), dim3(64 * MLP_WAVES_PER_BLOCK), lds, s, a);
    HIP_TRY(h, hipGetLastError());
    return 0;
}

// (KS0, MT, NL) the network pass is compiled for: nsf <= 28 or 32 inputs, nnod <= 16 or 32 nodes, 2..4 weight layers
#define ANNP_MLP_SHAPES(X) X(7, 1, 2) X(7, 1, 3) X(7, 1, 4) X(7, 2, 2) X(7, 2, 3) X(7, 2, 4) \
                           X(8, 1, 2) X(8, 1, 3) X(8, 1, 4) X(8, 2, 2) X(8, 2, 3) X(8, 2, 4)
void mlp_shape(const annp_hip_handle *h, int &ks0, int &mt) { ks0 = h->nsf_dev <= 28 ? 7 : 8; mt = h->nnod <= 16 ? 1 : 2; }

int run_mlp(annp_hip_handle *h, const MlpArgs &a, hipStream_t s)
{
    int ks0, mt;
    mlp_shape(h, ks0, mt);
#define X(K, M, L) if (ks0 == K && mt == M && h->nl == L) return launch_mlp<K, M, L>(h, a, s);
    ANNP_MLP_SHAPES(X)
#undef X
    return fail(h, ANNP_HIP_ESHAPE, "no network kernel for nsf=%d nnod=%d layers=%d", h->nsf, h->nnod, h->nl);
}

// the network pass: one launch per element of the potential, each with that element's operand image
int run_mlp_elements(annp_hip_handle *h, MlpArgs m, hipStream_t s)
{
    const double *base = m.img;
    for (int e = 0; e < h->nelem; e++) {
        m.img = base + (size_t)e * h->img_stride;
        m.elem = e;
        if (int rc = run_mlp(h, m, s)) return rc;
    }
    return 0;
}

// operand image of the network pass for this handle's shape (empty when the shape is not compiled)
std::vector<double> mlp_image(const annp_hip_handle *h, const double *const *W, const double *const *B, const double *coefmat)
{
    int ks0, mt;
    mlp_shape(h, ks0, mt);
    std::vector<double> img;
#define X(K, M, L)                                                                         \
    if (ks0 == K && mt == M && h->nl == L) {                                               \
        img.resize((size_t)MlpSlots<K, M, L>::total * 64);                                 \
        mlp_build_image<K, M, L>(img.data(), W, B, coefmat, h->nsf_dev, h->nnod);              \
    }
    ANNP_MLP_SHAPES(X)
#undef X
    return img;
}

int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Record capacities for the next evaluation from the in-cutoff maximum of the last one.
// Chebyshev force pass: an atom above the capacity is not lost, it goes through the fix-up launch, so the slack is
// small; 128 is kept as long as the maximum allows it (up to there a lane holds 1/r and fc' in registers).
int fe_next_cap(int mx)
{
    int c = std::max(16, round_up(mx + 2, 16));
    if (mx <= 128 && c > 128) c = 128;
    return c;
}
// Chebyshev descriptor pass: state slots per atom (16 per lane-iteration, SH_CAP_MAX at most; an atom above goes to the fix-up launch)
int sh_next_cap(int mx) { return std::min((int)SH_CAP_MAX, std::max((int)SH_CAP_MIN, round_up(mx, 16))); }       // (no slack: the fix-up launch is the slack)
// Behler kernels: nothing stands behind an overflow (it is reported and the evaluation has to be re-issued), so the
// slack is generous: an eighth of the count, at least 2.
// (in steps of 2: the force pass's LDS decides how many workgroups a CU holds -- 18 in-range neighbours of fcc Ni: capacity 20
// = three workgroups, 24 = two)
int ni_next_cap(int mx) { return std::max(8, round_up(mx + std::max(2, mx / 8), 2)); }

// Look at the flag words an evaluation copied back.  Updates the capacities for the next evaluation and turns a
// device-side capacity error into sticky_rc.
void digest_flags(annp_hip_handle *h)
{
    int over = h->h_flags[0];
    const int mx = h->h_flags[1], nfix = h->h_flags[2];
    h->info[0] = mx; h->info[1] = nfix; h->info[2] = h->cap_last;
    if (over >= ANNP_REPLAN_BAD_TARGET) {        // not a neighbour count: annp_hip_replan_fold_plan was handed a target outside [0, nkeys)
        h->sticky_rc = fail(h, ANNP_HIP_EARG, "replan_fold_plan: a target index lay outside [0, nkeys); it was left out of the plan");
        over = 0;
    }
    if (mx > 0) h->sh_cap = sh_next_cap(mx);
    if (h->descriptor == ANNP_HIP_DESC_CHEBYSHEV) {
        h->fe_cap = fe_next_cap(mx);
        // The moment kernels take atoms with up to SH_CAP_MAX neighbours and queue the others for the pair-loop fix-up launches, which
        // run one wave per workgroup: fine for a few atoms, slow for a dense system.  When a sixteenth of the atoms went through
        // the queue, the next evaluation uses the pair-loop kernels for all of them, until the maximum is back under the limit.
        if (h->flags_sh) h->fe_dense = mx > SH_CAP_MAX && nfix > h->flags_inum / 16;
        else h->fe_dense = mx > SH_CAP_MAX;
        if (h->fe_dense != h->fe_dense_said) {          // a 2x change of speed the caller would otherwise only see in its timings
            h->fe_dense_said = h->fe_dense;
            if (h->notice) {
                if (h->fe_dense)
                    std::fprintf(h->notice, "annp/hip: up to %d in-cutoff neighbours per atom, more than the %d the moment kernels keep per atom "
                                 "(%d atoms went through the fix-up launch): the Chebyshev passes run pair by pair from the next evaluation on, "
                                 "at about half the speed\n", mx, (int)SH_CAP_MAX, nfix);
                else
                    std::fprintf(h->notice, "annp/hip: at most %d in-cutoff neighbours per atom again: back to the moment kernels\n", mx);
                std::fflush(h->notice);
            }
        }
        h->info[3] = (h->fe_dense || h->fe_desc_pairs || h->fe_force_pairs) ? h->fe_cap : h->sh_cap;     // capacity of the next force pass
        // The force table of annp_fe_force_sh keeps eight atoms with consecutive indices per bucket, and counts the contributions that
        // found none (each is three memory requests, where a bucket leaves with three for all its contributions).  Measured at 1 M
        // atoms (round 5, tools/kbench.py): atoms in the order LAMMPS' atom_modify sort leaves them (bins of half the neighbour
        // cutoff, no order inside a bin) lose 12.8 contributions per atom that way and the pass 4 % (5.45 -> 5.67 ms); atoms in
        // random order lose ~100 per atom and the pass takes 2.3 times as long.  The caller is told from 40 per atom on -- a third
        // of a bcc-Fe neighbourhood: sorting its atoms in space then buys up to that factor.
        if (h->flags_sh) {
            const long long lost = (long long)(unsigned)h->h_flags[4];        // (the device adds to an int: above 2^31 it reads negative)
            h->shf_scattered = lost > 40 * (long long)h->flags_inum;
            if (h->shf_scattered != h->shf_scattered_said) {
                h->shf_scattered_said = h->shf_scattered;
                if (h->notice) {
                    if (h->shf_scattered)
                        std::fprintf(h->notice, "annp/hip: atoms are not ordered in space (%.1f force contributions per atom found no room in the force pass's "
                                     "table of eight-atom buckets): the force pass takes up to 2.3 times as long as with sorted atoms (atom_modify sort)\n",
                                     (double)lost / std::max(1, h->flags_inum));
                    else
                        std::fprintf(h->notice, "annp/hip: atoms are ordered in space again\n");
                    std::fflush(h->notice);
                }
            }
        }
        if (over > 0)
            h->sticky_rc = fail(h, ANNP_HIP_ENEIGHCAP, "an atom has %d in-cutoff neighbours, more than the list-row capacity the "
                                "evaluation was given (max_numneigh) or than LDS holds; it was skipped", over);
    } else if (h->descriptor == ANNP_HIP_DESC_BEHLER) {
        if (over > 0) {         // not an atom that outgrew its records (the fix-up launches take those): more than LDS can hold
            h->ni_cap = round_up(over + 2, 8);
            h->ni_primed = false;
            h->sticky_rc = fail(h, ANNP_HIP_ENEIGHCAP, "%d neighbours inside the descriptor cutoff exceed what the kernels' LDS records can hold "
                                "(or the list row is longer than max_numneigh said); the affected atoms were skipped", over);
        } else {
            h->ni_cap = ni_next_cap(mx);
        }
        h->info[3] = h->ni_cap;
    } else {
        h->info[3] = h->cap_last;
        if (over > 0)
            h->sticky_rc = fail(h, ANNP_HIP_ENEIGHCAP, "%d neighbours inside the descriptor cutoff exceed the kernel capacity %d; "
                                "the affected atoms were skipped", over, h->cap_last);
    }
    if (over > 0) h->reset_err = true;
}

// wait = false: only if the copy has landed already (never blocks); wait = true: block until it has.
// Returns a pending error once.
int poll_flags(annp_hip_handle *h, bool wait)
{
    if (h->flags_pending) {
        hipError_t e = wait ? hipEventSynchronize(h->ev_flags) : hipEventQuery(h->ev_flags);
        if (e == hipSuccess) {
            h->flags_pending = false;
            digest_flags(h);
        } else if (e != hipErrorNotReady) {
            return fail(h, ANNP_HIP_EDEVICE, "flag read-back failed: %s", hipGetErrorString(e));
        }
    }
    if (h->sticky_rc) {
        const int rc = h->sticky_rc;
        h->sticky_rc = 0;
        return rc;
    }
    return 0;
}

// Cutoff of a neighbour list the library builds itself (annp_hip_neigh_build_device, annp_hip_compute_n) when the caller
// asks for `cutneigh` = cutmax + skin.  Behler potentials: a neighbour only contributes while r * CFLENGTH < Rc of the
// functions (ni:693, 729) -- 3.9 A for the shipped Ni file, whose cutmax line says 6.5 -- so a list cut at that distance
// plus the SAME skin holds every atom that can come into range before the caller's next rebuild (its criterion, a
// displacement of skin / 2, is about the skin only) and the kernels filter 3x fewer candidates (~75 instead of ~224 per
// atom in fcc Ni).  Results are those of the long list up to the order of the sums.  Not in compat mode (ni:737-738 depends
// on list order: there the list is exactly what was asked for) and not with ANNP_HIP_FULL_LIST=1.
double list_cutoff(const annp_hip_handle *h, double cutneigh)
{
    if (h->descriptor != ANNP_HIP_DESC_BEHLER || h->ni_compat || h->full_list) return cutneigh;
    const double rc_desc = std::max(h->sym_rad.size() >= 3 ? h->sym_rad[2] : 0.0, h->sym_ang.size() >= 4 ? h->sym_ang[3] : 0.0);
    if (!(rc_desc > 0.0)) return cutneigh;
    double rc_a = 0.0;      // every function's own Rc is honoured: the largest one decides
    for (size_t k = 2; k < h->sym_rad.size(); k += 3) rc_a = std::max(rc_a, h->sym_rad[k]);
    for (size_t k = 3; k < h->sym_ang.size(); k += 4) rc_a = std::max(rc_a, h->sym_ang[k]);
    const double skin = std::max(0.0, cutneigh - h->cut);
    const double c = rc_a / ANNP_CFLENGTH * (1.0 + 1e-9) + skin;
    return std::min(c, cutneigh);
}

// the calling thread's current device is put back when an entry point returns
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) { err = hipSetDevice(dev); switched = (err == hipSuccess) && prev >= 0; }
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};
#define DEVICE_GUARD(h)                                                                              \
    DeviceGuard guard_((h)->device);                                                                 \
    if (guard_.err != hipSuccess) return fail(h, ANNP_HIP_EDEVICE, "hipSetDevice(%d) failed: %s", (h)->device, hipGetErrorString(guard_.err))

// waves per workgroup of the Chebyshev passes (the kernels take any 1..4: waves never synchronise with each other).
// Measured at 1 M atoms: force pass 14.63 ms with 4, 15.1 with 2, 14.9 with 1; descriptor pass indifferent.
constexpr int fe_wpb_desc() { return ANNP_WAVES_PER_BLOCK; }
constexpr int fe_wpb_force() { return ANNP_WAVES_PER_BLOCK; }

template <bool VIR>
void launch_fe_force(const FeArgs &a, hipStream_t s)
{
    const int wpb = fe_wpb_force();
    const int blocks = (a.inum + wpb - 1) / wpb;
    if (a.n_cap == 128)         // the capacity of a bcc-Fe box: layout offsets are compile-time constants
        hipLaunchKernelGGL((annp_fe_force<FE_NP, FE_NT, VIR, true, 128>), dim3(blocks), dim3(64 * wpb), fe_force_lds_per_wave(128, true) * wpb, s, a);
    else if (fe_force_auxreg(a.n_cap))
        hipLaunchKernelGGL((annp_fe_force<FE_NP, FE_NT, VIR, true>), dim3(blocks), dim3(64 * wpb), fe_force_lds_per_wave(a.n_cap, true) * wpb, s, a);
    else
        hipLaunchKernelGGL((annp_fe_force<FE_NP, FE_NT, VIR, false>), dim3(blocks), dim3(64 * wpb), fe_force_lds_per_wave(a.n_cap, false) * wpb, s, a);
}

// Chebyshev descriptor pass (pair_style annp Fe and anna_adp): the moment kernel for atoms with at most sh_cap in-cutoff
// neighbours, the pair-loop kernel behind it for the ones it queued (none in the steady state: microseconds)
int launch_fe_desc(annp_hip_handle *h, FeArgs a, int inum, int cap_list, int max_numneigh, hipStream_t s)
{
    int rc;
    if (h->fe_desc_pairs || (h->fe_dense && h->descriptor == ANNP_HIP_DESC_CHEBYSHEV)) {
        a.n_cap = cap_list;
        const int wpd = fe_wpb_desc();
        const size_t lds1 = fe_desc_lds_per_wave(a.n_cap) * wpd;
        if (lds1 > 160 * 1024) return fail(h, ANNP_HIP_ENEIGHCAP, "neighbour list too long for LDS (%d)", max_numneigh);
        hipLaunchKernelGGL((annp_fe_desc<FE_NP, FE_NT>), dim3((inum + wpd - 1) / wpd), dim3(64 * wpd), lds1, s, a);
        HIP_TRY(h, hipGetLastError());
        return 0;
    }
    const int cap = std::max((int)SH_CAP_MIN, std::min(h->sh_cap, cap_list));          // multiples of 16
    const size_t lds_fix = fe_desc_lds_per_wave(cap_list);
    const bool fix = cap_list > cap;
    if (fix && lds_fix > 160 * 1024) return fail(h, ANNP_HIP_ENEIGHCAP, "neighbour list too long for LDS (%d)", max_numneigh);
    if (fix && (rc = ensure(h, h->ovf_desc, (size_t)inum))) return rc;
    a.n_cap = cap;
    h->sh_cap_used = cap;
    if (!a.A) {         // the kernel sums monomial moments and changes basis in the atom's moment row: it needs one, whoever reads it afterwards
        if ((rc = ensure(h, h->mom, (size_t)(inum + SHF_GA) * SH_MPAD, true))) return rc;      // (sized and zeroed as the force pass wants it, below)
        a.A = h->mom.p;
    }
    a.ovf_count = h->fw + 3; a.ovf_list = fix ? h->ovf_desc.p : nullptr; a.ovf_cap = fix ? inum : 0;
    // waves per workgroup: as many waves per CU as the LDS allows, and of those shapes the largest workgroup (measured at
    // 1 M atoms, 8 waves per CU each: 5.8 ms with 4 waves per workgroup, 6.3 with 1)
    int wpb = h->sh_wpb;
    if (wpb <= 0) {
        int best = 0;
        for (int w = 1; w <= 4; w++) {
            const int waves = (int)((size_t)160 * 1024 / (sh_lds_per_wave(cap) * w)) * w;
            if (waves >= best) { best = waves; wpb = w; }
        }
    }
    const int groups = (inum + SH_GA - 1) / SH_GA;
    // The monomial totals are parked in the moment row and changed of basis at the end (round 4b).  ANNP_HIP_SH_TAIL=group (developer A/B
    // switch) selects round 6's variant for launches with room for at most 112 neighbours per atom: the change of basis group by group
    // out of LDS, nothing parked -- 3 GB less memory traffic per 1 M-atom launch and 2 % SLOWER (4.84-4.86 against 4.72-4.77 ms on one box,
    // alternating: gpurun_out/r6_ab2), as round 4's five-group version was: the pass does not wait for that traffic, and three tails in
    // the middle of the columns cost more than one at the end.
    if (cap <= SHG_CAP_MAX && h->sh_group)
        hipLaunchKernelGGL((annp_fe_desc_sh<FE_NP, FE_NT, true>), dim3((groups + wpb - 1) / wpb), dim3(64 * wpb), sh_lds_per_wave(cap) * wpb, s, a);
    else
        hipLaunchKernelGGL((annp_fe_desc_sh<FE_NP, FE_NT, false>), dim3((groups + wpb - 1) / wpb), dim3(64 * wpb), sh_lds_per_wave(cap) * wpb, s, a);
    HIP_TRY(h, hipGetLastError());
    if (fix) {
        FeArgs b = a;
        b.n_cap = cap_list;
        hipLaunchKernelGGL((annp_fe_desc_fixup<FE_NP, FE_NT>), dim3(std::min(inum, 1024)), dim3(64), lds_fix, s, b);
        HIP_TRY(h, hipGetLastError());
    }
    return 0;
}

// ---- one evaluation on device-resident data ----------------------------------------
// Nothing here waits for the device in the steady state: capacities come from the previous evaluation's flag
// words (whenever their copy has landed), this evaluation's flag words are copied back behind its last kernel.
// Only the first evaluation on a handle (and the one after a Behler capacity error) sizes itself synchronously.
// the sticky word and the current evaluation's words into the host mirror, in the layout digest_flags reads
int copy_flags(annp_hip_handle *h, hipStream_t st)
{
    // One writer of the host mirror at a time: the synchronous sizing paths copy in the caller's stream while the copy of the
    // evaluation before may still be on its way on the side stream -- landing late it would overwrite the words the host is about
    // to read.  It is waited for and digested first (an error it carries is reported by the next look at the handle, as ever).
    if (st != h->stream_flags && h->flags_pending) {
        HIP_TRY(h, hipEventSynchronize(h->ev_flags));
        h->flags_pending = false;
        digest_flags(h);
    }
    HIP_TRY(h, hipMemcpyAsync(h->h_flags, h->d_flags, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipMemcpyAsync(h->h_flags + 1, h->fw + 1, (ANNP_NFLAGS - 1) * sizeof(int), hipMemcpyDeviceToHost, st));
    return 0;
}

int compute_device_impl(annp_hip_handle *h, int inum, int nall, const double *d_x, const int *d_type, const int *d_ilist,
                        const int *d_numneigh, const long long *d_first, const int *d_neigh, int max_numneigh,
                        double *d_f, double *d_eatom, double *d_eng, double *d_virial, double *d_vatom, hipStream_t s)
{
    int rc;
    if ((rc = poll_flags(h, false))) return rc;      // an error of an earlier evaluation, reported once
    // A list build whose row maximum nobody has looked at yet (neigh_kernels.hpp: builds behind annp_hip_neigh_build_device do not wait
    // for it): looked at here as soon as the word has landed -- by the first or second evaluation on that list, never blocking -- so a
    // row that outgrew the pitch is reported a step late, not a rebuild interval late
    if (h->nb.pending && hipEventQuery(h->nb.ev_lazy) == hipSuccess) {
        std::string msg;
        if (int rs = neigh_settle(h->nb, msg)) return fail(h, rs, "%s", msg.c_str());
    }
    if (inum <= 0) return 0;
    // The global virial without per-atom virials (every step of an NPT run): the evaluation's forces go to a scratch array, and one
    // streaming kernel behind the passes adds them onto the caller's f and sums x (x) f over owned atoms and ghosts -- the reference's
    // own route (virial_fdotr_compute), equal to the pairwise tally to round-off, and 50 us where the tally cost the force pass 6 %.
    double *f_caller = nullptr, *virial_caller = nullptr;
    // (Chebyshev potentials only: the reference's Behler file tallies its virial from the forces BEFORE their unit conversion
    // (ni/src/pair_annp.cpp:188-198: f gets Fj * CFFORCE, ev_tally_xyz gets Fj), so there the tally and sum x (x) f differ by that factor,
    // and the boundary's virial is the tally's; anna_adp keeps the tally too)
    if (d_virial && !d_vatom && !h->virial_tally && h->descriptor == ANNP_HIP_DESC_CHEBYSHEV) {
        if ((rc = ensure(h, h->fscratch, (size_t)nall * 3))) return rc;
        HIP_TRY(h, hipMemsetAsync(h->fscratch.p, 0, sizeof(double) * 3 * (size_t)nall, s));
        f_caller = d_f; virial_caller = d_virial;
        d_f = h->fscratch.p; d_virial = nullptr;
    }
    if ((rc = ensure(h, h->G, (size_t)inum * ANNP_GPAD))) return rc;
    // (coefficient rows start out as zeros, and there are rows behind the last list entry's: the force pass multiplies a few entries of
    // a neighbouring row by zero: they must be numbers, whether the network pass wrote them or not)
    if ((rc = ensure(h, h->coef, (size_t)(inum + SHF_GA) * ANNP_CPAD, true))) return rc;
    if ((rc = ensure(h, h->ncount, (size_t)inum))) return rc;
    // The per-evaluation flag words come in two sets that take turns (h->fw): behind every evaluation a side stream copies its set to
    // the host and clears it again (the tail of this function), so no memset and no copy stands in the caller's stream between two
    // evaluations, and an evaluation only waits for the clearing of ITS set, two evaluations old.  Word [0] of d_flags is the sticky
    // one, shared by both sets, cleared here once the host has seen it.
    if (h->flags_dirty) {               // (the evaluation before this one left early: its set was never handed to the side stream)
        HIP_TRY(h, hipMemsetAsync(h->fw + 1, 0, (ANNP_NFLAGS - 1) * sizeof(int), s));
    } else {
        h->flags_par ^= 1;
        h->fw = h->d_flags + ANNP_NFLAGS * (1 + h->flags_par);
        if (h->set_used[h->flags_par]) HIP_TRY(h, hipStreamWaitEvent(s, h->ev_set[h->flags_par], 0));      // cleared two evaluations ago
    }
    h->flags_dirty = true;
    if (h->reset_err) {
        HIP_TRY(h, hipMemsetAsync(h->d_flags, 0, sizeof(int), s));
        h->reset_err = false;
    }
    if (h->timing) {
        h->ev = h->evring.data() + 4 * (size_t)(h->ev_count % annp_hip_handle::kRing);
        HIP_TRY(h, hipEventRecord(h->ev[0], s));
    }

    const int cap_list = std::max(16, round_up(max_numneigh, 16));
    double *vtab = nullptr;             // the global virial is tallied here and folded into d_virial at the end
    if (d_virial) {
        vtab = h->d_vslots;
        HIP_TRY(h, hipMemsetAsync(vtab, 0, sizeof(double) * 8 * ANNP_VSLOTS, s));
    }

    if (h->multi && !d_type)
        return fail(h, ANNP_HIP_EARG, "this potential distinguishes atom types (several elements or an unmapped type): d_type is required");
    const int *types = h->multi ? d_type : nullptr;
    MlpArgs m{};
    m.type = types; m.map = h->d_map; m.elem = 0; m.active = h->active;
    m.inum = inum; m.ilist = d_ilist; m.nsf = h->nsf_dev; m.nnod = h->nnod; m.nl = h->nl;
    m.ncoef = h->descriptor == ANNP_HIP_DESC_CHEBYSHEV ? FE_NP + 2 * FE_NT + 1 : h->nsf_dev;
    for (int l = 0; l < std::min(h->nl, (int)MLP_MAXL); l++) m.act[l] = h->flagact[l];      // (anna_adp may have more layers; it does not use m)
    m.nmul = h->d_norm; m.nsub = h->d_norm + ANNP_GPAD; m.nden = h->d_norm + 2 * ANNP_GPAD; m.img = h->d_mlp_img;
    m.e_scale = h->e_scale; m.e_shift = h->e_shift; m.e_atom = h->e_atom;
    m.G = h->G.p; m.coef = h->coef.p; m.eatom = d_eatom; m.eng = d_eng;

    if (h->descriptor == ANNP_HIP_DESC_CHEBYSHEV) {
        FeArgs a{};
        a.inum = inum; a.ilist = d_ilist; a.x = d_x; a.numneigh = d_numneigh; a.first = d_first; a.neigh = d_neigh;
        a.cutsq = h->cutsq; a.rc_list = std::sqrt(h->cutsq); a.rc_par = h->cut;
        a.por_list = ANNP_MY_PI / a.rc_list; a.two_over_rcp = 2.0 / a.rc_par;
        a.type = types; a.active = h->active;
        a.G = h->G.p; a.coef = h->coef.p; a.f = d_f; a.virial = vtab; a.vatom = d_vatom; a.ncount = h->ncount.p;
        a.errflag = h->d_flags;
        // pass 1 (and, for the force pass on the moments, their buffer; that pass needs the fix-up launch behind it)
        const size_t lds_fix = fe_force_lds_per_wave(cap_list, false);      // the fix-up runs one wave per workgroup
        const bool fix_possible = lds_fix <= 160 * 1024;
        const bool sh_force = !h->fe_desc_pairs && !h->fe_force_pairs && !h->fe_dense && fix_possible;
        h->fe_last_sh = sh_force; h->fe_last_inum = inum;
        if (sh_force) {
            // the moment rows start out as zeros: annp_fe_force_shp copies the rows of a unit's eight list entries into LDS as they are,
            // the rows of atoms the descriptor pass left to the fix-up launch (never written, or written by an earlier evaluation)
            // and up to SHF_GA rows behind the last entry included, and multiplies some of what it copied by zero
            if ((rc = ensure(h, h->mom, (size_t)(inum + SHF_GA) * SH_MPAD, true)) || (rc = ensure(h, h->fe_nbrs, (size_t)inum * SH_CAP_MAX))) return rc;
            a.A = h->mom.p; a.nbrs = h->fe_nbrs.p;
        }
        a.nmax_word = h->fw + 1;           // (annp_fe_desc_sh raises it itself; the pair-loop descriptor kernel does not)
        const bool desc_sh = !h->fe_desc_pairs && !h->fe_dense;
        if ((rc = launch_fe_desc(h, a, inum, cap_list, max_numneigh, s))) return rc;
        if (!desc_sh) hipLaunchKernelGGL(annp_max_int, dim3(annp_max_int_blocks(inum)), dim3(256), 0, s, h->ncount.p, inum, h->fw + 1);
        HIP_TRY(h, hipGetLastError());
        if (h->timing) HIP_TRY(h, hipEventRecord(h->ev[1], s));
        // pass 2
        m.act_plain = 0; m.energy_raw = 0;
        if ((rc = run_mlp_elements(h, m, s))) return rc;
        if (h->timing) HIP_TRY(h, hipEventRecord(h->ev[2], s));
        // pass 3: LDS records sized by the in-cutoff maximum of the previous evaluation; an atom that has more
        // is queued by the kernel and taken by the fix-up launch behind it, which has room for a whole list row
        const bool vir = d_virial || d_vatom;
        if (sh_force) {
            // pass 3 on the moments: atoms the descriptor pass had no state for (more than sh_cap_used neighbours) go to the queue
            // and the pair-loop fix-up launch, which has room for a whole list row; nothing to size, nothing to wait for
            const int cap = h->sh_cap_used;
            const bool fixup = cap_list > cap;
            if (fixup && (rc = ensure(h, h->ovf, (size_t)inum))) return rc;
            a.n_cap = cap;
            a.ovf_count = h->fw + 2; a.ovf_list = fixup ? h->ovf.p : nullptr; a.ovf_cap = fixup ? inum : 0;
            if (h->pre_force_wait) { HIP_TRY(h, hipStreamWaitEvent(s, h->pre_force_wait, 0)); h->pre_force_wait = nullptr; }
            a.tab_spills = h->fw + 4;
            a.shf_places_by_number = h->shf_places_by_number;
            {
#ifdef ANNP_SHF_CHECK
                a.chk_nall = nall;                          // (developer build: the kernel checks its indices against it)
#endif
                const int apb = SHF_GROUPS * SHF_GA;        // atoms per workgroup
                if (vir) hipLaunchKernelGGL((annp_fe_force_sh<FE_NP, FE_NT, true>), dim3((inum + apb - 1) / apb), dim3(64 * SHF_WAVES), shf_lds_per_block(), s, a);
                else hipLaunchKernelGGL((annp_fe_force_sh<FE_NP, FE_NT, false>), dim3((inum + apb - 1) / apb), dim3(64 * SHF_WAVES), shf_lds_per_block(), s, a);
            }
            HIP_TRY(h, hipGetLastError());
            if (fixup) {
                FeArgs b = a;
                b.n_cap = cap_list;
                const int fblocks = std::min(inum, 1024);
                if (vir) hipLaunchKernelGGL((annp_fe_force_fixup<FE_NP, FE_NT, true>), dim3(fblocks), dim3(64), lds_fix, s, b);
                else hipLaunchKernelGGL((annp_fe_force_fixup<FE_NP, FE_NT, false>), dim3(fblocks), dim3(64), lds_fix, s, b);
                HIP_TRY(h, hipGetLastError());
            }
            h->cap_last = cap;
        } else {        // the pair-loop force pass (rounds 1-2): a dense system, ANNP_HIP_FE_FORCE=pairs, or no room for the fix-up launch
            int cap3;
            // first evaluation on this handle: read the maximum just measured, once.  Also whenever a whole list row would not
            // fit the fix-up launch's LDS (very long rows): nothing would stand behind a stale capacity then
            if (h->fe_cap == 0 || (!fix_possible && h->fe_cap < cap_list)) {
                if (int rc2 = copy_flags(h, s)) return rc2;
                HIP_TRY(h, hipStreamSynchronize(s));
                if (h->h_flags[0] > 0) {
                    h->reset_err = true;
                    return fail(h, ANNP_HIP_ENEIGHCAP, "in-cutoff neighbours %d exceed capacity %d", h->h_flags[0], a.n_cap);
                }
                cap3 = std::max(16, round_up(h->h_flags[1], 16));
                h->fe_cap = fe_next_cap(h->h_flags[1]);
            } else {
                cap3 = std::min(h->fe_cap, cap_list);
            }
            if ((rc = ensure(h, h->ovf, (size_t)inum))) return rc;
            a.n_cap = cap3;
            const bool fixup = cap3 < cap_list && fix_possible;
            a.ovf_count = h->fw + 2; a.ovf_list = fixup ? h->ovf.p : nullptr; a.ovf_cap = fixup ? inum : 0;
            if (fe_force_lds_per_wave(cap3) * fe_wpb_force() > 160 * 1024)
                return fail(h, ANNP_HIP_ENEIGHCAP, "too many in-cutoff neighbours for LDS (%d)", cap3);
            if (h->pre_force_wait) { HIP_TRY(h, hipStreamWaitEvent(s, h->pre_force_wait, 0)); h->pre_force_wait = nullptr; }
            if (vir) launch_fe_force<true>(a, s); else launch_fe_force<false>(a, s);
            HIP_TRY(h, hipGetLastError());
            if (fixup) {
                FeArgs b = a;
                b.n_cap = cap_list;
                const int fblocks = std::min(inum, 1024);
                if (vir) hipLaunchKernelGGL((annp_fe_force_fixup<FE_NP, FE_NT, true>), dim3(fblocks), dim3(64), lds_fix, s, b);
                else hipLaunchKernelGGL((annp_fe_force_fixup<FE_NP, FE_NT, false>), dim3(fblocks), dim3(64), lds_fix, s, b);
                HIP_TRY(h, hipGetLastError());
            }
            h->cap_last = cap3;
        }
    } else if (h->descriptor == ANNP_HIP_DESC_ANNA_ADP) {
        // pass 1: the same Chebyshev descriptor kernel, raw sums (adp:584-612 has no normalisation)
        FeArgs a{};
        a.inum = inum; a.ilist = d_ilist; a.x = d_x; a.numneigh = d_numneigh; a.first = d_first; a.neigh = d_neigh;
        a.cutsq = h->cutsq; a.rc_list = h->cut; a.rc_par = h->cut;          // fc and the radial argument both use the file's cutoff (adp:105,130,588)
        a.por_list = ANNP_MY_PI / a.rc_list; a.two_over_rcp = 2.0 / a.rc_par;
        a.G = h->G.p; a.ncount = h->ncount.p; a.errflag = h->d_flags;
        a.nmax_word = h->fw + 1;           // (annp_fe_desc_sh raises it itself; the pair-loop descriptor kernel does not)
        const bool desc_sh = !h->fe_desc_pairs && !h->fe_dense;
        if ((rc = launch_fe_desc(h, a, inum, cap_list, max_numneigh, s))) return rc;
        if (!desc_sh) hipLaunchKernelGGL(annp_max_int, dim3(annp_max_int_blocks(inum)), dim3(256), 0, s, h->ncount.p, inum, h->fw + 1);
        HIP_TRY(h, hipGetLastError());
        if (h->timing) { HIP_TRY(h, hipEventRecord(h->ev[1], s)); HIP_TRY(h, hipEventRecord(h->ev[2], s)); }
        // pass 2: network, ADP sums, energy, forces
        AnnaArgs q{};
        q.inum = inum; q.ilist = d_ilist; q.x = d_x; q.numneigh = d_numneigh; q.first = d_first; q.neigh = d_neigh;
        q.n_cap = 64 * ANNA_NR; q.rc = h->cut; q.G = h->G.p; q.net = h->d_net;
        q.net_doubles = h->net_doubles; q.net_in_lds = h->net_doubles <= ANNA_NET_LDS_MAX;
        q.nl = h->nl; q.nin = h->nsf_dev; q.nnod = h->nnod; q.nout = h->nout;
        for (int l = 0; l < h->nl; l++) q.actp |= (unsigned)(h->flagact[l] & 15) << (4 * l);
        for (int k = 0; k < 17; k++) q.gp[k] = h->gp[k];
        q.e_base = h->e_base;
        q.f = d_f; q.eatom = d_eatom; q.eng = d_eng; q.virial = vtab; q.vatom = d_vatom; q.errflag = h->d_flags;
        const size_t lds2 = anna_lds_per_wave(q.n_cap) * ANNP_WAVES_PER_BLOCK + anna_lds_net(h->net_doubles);
        if (h->pre_force_wait) { HIP_TRY(h, hipStreamWaitEvent(s, h->pre_force_wait, 0)); h->pre_force_wait = nullptr; }
        if (d_virial || d_vatom) hipLaunchKernelGGL((annp_anna_adp<true>), dim3(anna_blocks(inum)), dim3(256), lds2, s, q);
        else hipLaunchKernelGGL((annp_anna_adp<false>), dim3(anna_blocks(inum)), dim3(256), lds2, s, q);
        HIP_TRY(h, hipGetLastError());
        // more in-range neighbours than a wave holds (128): the error word stays set on the device until the host
        // has seen it, so it is reported by the next call or annp_hip_sync however many evaluations are enqueued
        h->cap_last = q.n_cap;
    } else {
        NiArgs a{};
        a.inum = inum; a.ilist = d_ilist; a.x = d_x; a.numneigh = d_numneigh; a.first = d_first; a.neigh = d_neigh;
        a.npsf = h->npsf; a.ntsf = h->ntsf; a.sym = h->d_sym; a.isym = h->d_isym; a.compat = h->ni_compat;
        a.type = types; a.active = h->active;
        a.rc_rad = h->sym_rad[2]; a.rc_ang = h->sym_ang[3];
        a.por_rad = ANNP_MY_PI / a.rc_rad; a.por_ang = ANNP_MY_PI / a.rc_ang;
        a.rc2a = (a.rc_ang / ANNP_CFLENGTH) * (a.rc_ang / ANNP_CFLENGTH) * (1.0 + 1e-12);
        for (int k = 0; k < 4; k++) { a.lam[k] = h->ni_lam[k]; a.eta[k] = h->ni_eta[k]; }
        a.rad_em = h->ni_rad_em;
        a.G = h->G.p; a.coef = h->coef.p; a.f = d_f; a.virial = vtab; a.vatom = d_vatom; a.ncount = h->ncount.p; a.errflag = h->d_flags;
        if (a.npsf > NI_MAXP || a.ntsf > NI_MAXT)
            return fail(h, ANNP_HIP_ESHAPE, "Behler kernels support npsf<=%d ntsf<=%d", NI_MAXP, NI_MAXT);
        const int cap_max = ni_max_cap(true, h->nsf);
        int cap_force;
        // room for the pair lists the descriptor pass hands to the force pass: n_cap (n_cap - 1) / 2 entries of 2 bytes per
        // atom (190 at capacity 20).  Not for very long records (the force pass then finds its pairs itself).
        auto pair_room = [&](NiArgs &q) -> int {
            q.pairs = nullptr; q.npair = nullptr; q.pstride = 0;
            const long long ps = round_up(q.n_cap * (q.n_cap - 1) / 2, 8);
            if (h->ni_no_pairs || ps > 2048 || (long long)inum * ps * 2 > (3ll << 30)) return 0;
            int r;
            if ((r = ensure(h, h->ni_pairs, (size_t)inum * ps)) || (r = ensure(h, h->ni_npair, (size_t)inum))) return r;
            q.pairs = h->ni_pairs.p; q.npair = h->ni_npair.p; q.pstride = (int)ps;
            return 0;
        };
        // Records: n_cap per atom from the previous evaluation's maximum.  A group of four atoms that has more is queued by the
        // descriptor pass and taken, pass by pass, by a second small launch whose records hold a whole list row (cap_big), so
        // the evaluation is complete whatever the configuration did since the capacity was learned.  An error remains only
        // for more in-range neighbours than the largest records LDS can hold, or a list row longer than the caller said.
        a.n_cap = std::min(h->ni_cap, cap_max);
        if ((rc = ensure(h, h->ni_nbr, (size_t)inum * a.n_cap)) || (rc = pair_room(a))) return rc;
        a.nbr = h->ni_nbr.p; a.nbr_stride = a.n_cap;
        const int cap_big = std::min(cap_max, std::max(a.n_cap, round_up(std::max(max_numneigh, 8), 8)));
        const bool fixup = cap_big > a.n_cap && !h->ni_no_fixup;
        const int ngroups = (inum + NI_GA - 1) / NI_GA;
        a.ovf_count = h->fw + 2; a.ovf_list = nullptr; a.ovf_cap = 0; a.fix = 0; a.skip_above = a.n_cap;
        if (fixup) {
            if ((rc = ensure(h, h->ovf, (size_t)ngroups)) || (rc = ensure(h, h->ni_fix_nbr, (size_t)ngroups * NI_GA * cap_big))) return rc;
            a.ovf_list = h->ovf.p; a.ovf_cap = ngroups;
        }
        NiArgs b = a;           // the fix-up launches
        b.fix = 1; b.n_cap = cap_big; b.nbr = h->ni_fix_nbr.p; b.nbr_stride = cap_big; b.pairs = nullptr; b.pstride = 0;
        ni_launch_desc(a, h->ni_shape, s);
        HIP_TRY(h, hipGetLastError());
        if (fixup) { ni_launch_desc_fix(b, h->ni_shape, s); HIP_TRY(h, hipGetLastError()); }
        hipLaunchKernelGGL(annp_max_int, dim3(annp_max_int_blocks(inum)), dim3(256), 0, s, h->ncount.p, inum, h->fw + 1);
        HIP_TRY(h, hipGetLastError());
        cap_force = a.n_cap;
        if (!h->ni_primed) {    // first evaluation on the handle (or the one after an error): look at the counts once
            if (int rc2 = copy_flags(h, s)) return rc2;
            HIP_TRY(h, hipStreamSynchronize(s));
            if (h->h_flags[0] > 0) {
                h->reset_err = true;
                h->ni_cap = std::min(cap_max, round_up(h->h_flags[0] + 2, 8));
                return fail(h, ANNP_HIP_ENEIGHCAP, "%d neighbours inside the descriptor cutoff exceed the kernel capacity %d (list rows: %d)",
                            h->h_flags[0], cap_big, max_numneigh);
            }
            h->ni_cap = ni_next_cap(h->h_flags[1]);
            h->ni_primed = true;
            cap_force = std::max(8, std::min(a.n_cap, round_up(h->h_flags[1], 2)));     // (fewer LDS bytes: more resident workgroups)
        }
        if (h->timing) HIP_TRY(h, hipEventRecord(h->ev[1], s));
        m.act_plain = 1; m.energy_raw = 1;
        if ((rc = run_mlp_elements(h, m, s))) return rc;
        if (h->timing) HIP_TRY(h, hipEventRecord(h->ev[2], s));
        a.n_cap = cap_force;
        h->cap_last = cap_force;
        if (h->pre_force_wait) { HIP_TRY(h, hipStreamWaitEvent(s, h->pre_force_wait, 0)); h->pre_force_wait = nullptr; }
        ni_launch_force(a, h->ni_shape, d_virial != nullptr || d_vatom != nullptr, s);
        HIP_TRY(h, hipGetLastError());
        if (fixup) { ni_launch_force_fix(b, h->ni_shape, d_virial != nullptr || d_vatom != nullptr, s); HIP_TRY(h, hipGetLastError()); }
    }
    if (virial_caller) {
        vtab = h->d_vslots;
        HIP_TRY(h, hipMemsetAsync(vtab, 0, sizeof(double) * 8 * ANNP_VSLOTS, s));
        hipLaunchKernelGGL(annp_fdotr_add, dim3(std::max(1, std::min(ANNP_VSLOTS, (nall + 255) / 256))), dim3(256), 0, s, nall, d_x, h->fscratch.p, f_caller, vtab);
        HIP_TRY(h, hipGetLastError());
        d_virial = virial_caller;
    }
    if (vtab) {
        hipLaunchKernelGGL(annp_virial_fold, dim3(1), dim3(1024), 0, s, vtab, d_virial);
        HIP_TRY(h, hipGetLastError());
    }
    // flag words of this evaluation, for whoever looks next (poll_flags)
    HIP_TRY(h, hipEventRecord(h->ev_tail, s));
    HIP_TRY(h, hipStreamWaitEvent(h->stream_flags, h->ev_tail, 0));
    if (int rc2 = copy_flags(h, h->stream_flags)) return rc2;
    HIP_TRY(h, hipEventRecord(h->ev_flags, h->stream_flags));
    HIP_TRY(h, hipMemsetAsync(h->fw + 1, 0, (ANNP_NFLAGS - 1) * sizeof(int), h->stream_flags));
    HIP_TRY(h, hipEventRecord(h->ev_set[h->flags_par], h->stream_flags));
    h->set_used[h->flags_par] = true;
    h->flags_pending = true; h->flags_dirty = false;
    h->flags_sh = h->fe_last_sh; h->flags_inum = h->fe_last_inum;      // (what digest_flags judges the queue length by)

    if (h->timing) { HIP_TRY(h, hipEventRecord(h->ev[3], s)); h->ev_count++; }
    (void)nall; (void)d_type;
    return 0;
}

// ---- the halo wire: RCCL point-to-point, called from here --------------------------------------------------------------
// librccl is opened at run time, so a build of LAMMPS that never asks for the wire does not need it; a process that has
// torch loaded gets torch's copy (same soname), which keeps one RCCL per process.
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) { r.err = std::string("dlopen(librccl.so.1): ") + (dlerror() ? dlerror() : "not found"); return; }
#define RCCL_SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, sym)); if (!r.field) { r.err = std::string("librccl lacks ") + sym; return; }
        RCCL_SYM(GetUniqueId, "ncclGetUniqueId") RCCL_SYM(CommInitRank, "ncclCommInitRank") RCCL_SYM(CommDestroy, "ncclCommDestroy")
        RCCL_SYM(GroupStart, "ncclGroupStart") RCCL_SYM(GroupEnd, "ncclGroupEnd") RCCL_SYM(Send, "ncclSend") RCCL_SYM(Recv, "ncclRecv")
        RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef RCCL_SYM
    });
    return r;
}
#define RCCL_TRY(h, call)                                                                                        \
    do {                                                                                                         \
        ncclResult_t r_ = (call);                                                                                \
        if (r_ != ncclSuccess) return fail(h, ANNP_HIP_EDEVICE, "%s failed: %s", #call, rccl().GetErrorString(r_)); \
    } while (0)

// CPUs this process may use: the affinity mask capped by the cgroup quota (hardware_concurrency() knows neither, and
// under the usual one-core-per-rank MPI binding sixteen copy threads would time-share that one core)
int usable_cpus()
{
    int n = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = (int)std::max(1u, std::thread::hardware_concurrency());
    if (FILE *fp = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32];
        long long period = 0;
        if (std::fscanf(fp, "%31s %lld", quota, &period) == 2 && std::strcmp(quota, "max") != 0 && period > 0)
            n = (int)std::min<long long>(n, std::max<long long>(1, std::atoll(quota) / period));
        std::fclose(fp);
    }
    return n;
}

int ensure_pool(annp_hip_handle *h)
{
    if (h->pool) return 0;
    int nthr = std::min(usable_cpus(), 8);
    if (const char *e = std::getenv("ANNP_HIP_COPY_THREADS")) nthr = std::max(1, std::atoi(e));
    h->pool = new (std::nothrow) CopyPool(nthr - 1);
    if (!h->pool) return fail(h, ANNP_HIP_ENOMEM, "host allocation failed");
    return 0;
}

// pinned staging buffers (32 MB each) + their events, shared by the host-list upload and the list hand-back
int ensure_list_staging(annp_hip_handle *h)
{
    for (int b = 0; b < annp_hip_handle::kListBufs; b++) {
        if (!h->pin_list[b]) {
            HIP_TRY(h, hipHostMalloc((void **)&h->pin_list[b], annp_hip_handle::kListChunk * sizeof(int)));
            h->bytes += annp_hip_handle::kListChunk * sizeof(int);
        }
        if (!h->ev_list[b]) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_list[b], hipEventDisableTiming));
    }
    return ensure_pool(h);
}

}  // namespace

// =====================================================================================
extern "C" {

int annp_hip_abi_version(void) { return ANNP_HIP_ABI_VERSION; }

int annp_hip_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

const char *annp_hip_last_error(const annp_hip_handle *h) { return h ? h->err.c_str() : g_init_error.c_str(); }

double annp_hip_bytes(const annp_hip_handle *h) { return h ? (double)h->bytes : 0.0; }

void annp_hip_clear(annp_hip_handle *h)
{
    if (!h) return;
    DeviceGuard guard_(h->device);
    (void)hipDeviceSynchronize();           // evaluations may still be running on the caller's streams
    if (h->comm) { (void)rccl().CommDestroy(h->comm); h->comm = nullptr; }
    if (h->d_norm) (void)hipFree(h->d_norm);
    if (h->d_sym) (void)hipFree(h->d_sym);
    if (h->d_isym) (void)hipFree(h->d_isym);
    if (h->d_mlp_img) (void)hipFree(h->d_mlp_img);
    if (h->d_map) (void)hipFree(h->d_map);
    if (h->d_net) (void)hipFree(h->d_net);
    release(h, h->G); release(h, h->coef); release(h, h->x); release(h, h->f); release(h, h->eatom); release(h, h->vatom);
    release(h, h->type); release(h, h->ilist); release(h, h->numneigh); release(h, h->neigh); release(h, h->ncount); release(h, h->ni_nbr); release(h, h->ni_npair); release(h, h->ni_fix_nbr); release(h, h->ni_pairs); release(h, h->ovf);
    release(h, h->mom); release(h, h->fe_nbrs); release(h, h->ovf_desc); release(h, h->fscratch);
    release(h, h->rp_flag); release(h, h->rp_cnt); release(h, h->rp_pos); release(h, h->rp_bs);
    if (h->rp_tot) (void)hipFree(h->rp_tot);
    if (h->rp_tot_h) (void)hipHostFree(h->rp_tot_h);
    release(h, h->first);
    neigh_release(h->nb);
    if (h->d_scalars) (void)hipFree(h->d_scalars);
    if (h->d_vslots) (void)hipFree(h->d_vslots);
    if (h->d_flags) (void)hipFree(h->d_flags);
    if (h->h_flags) (void)hipHostFree(h->h_flags);
    if (h->ev_flags) (void)hipEventDestroy(h->ev_flags);
    if (h->ev_tail) (void)hipEventDestroy(h->ev_tail);
    for (int k = 0; k < 2; k++) if (h->ev_set[k]) (void)hipEventDestroy(h->ev_set[k]);
    if (h->stream_flags) (void)hipStreamDestroy(h->stream_flags);
    if (h->reg_x.ok && hipHostUnregister(const_cast<void *>(h->reg_x.ptr)) != hipSuccess) (void)hipGetLastError();
    if (h->reg_f.ok && hipHostUnregister(const_cast<void *>(h->reg_f.ptr)) != hipSuccess) (void)hipGetLastError();
    if (h->pin_x) (void)hipHostFree(h->pin_x);
    if (h->ev_f_up) (void)hipEventDestroy(h->ev_f_up);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    if (h->pin_f) (void)hipHostFree(h->pin_f);
    if (h->pin_e) (void)hipHostFree(h->pin_e);
    if (h->pin_v) (void)hipHostFree(h->pin_v);
    if (h->pin_first) (void)hipHostFree(h->pin_first);
    if (h->pin_num) (void)hipHostFree(h->pin_num);
    for (int b = 0; b < annp_hip_handle::kListBufs; b++) {
        if (h->pin_list[b]) (void)hipHostFree(h->pin_list[b]);
        if (h->ev_list[b]) (void)hipEventDestroy(h->ev_list[b]);
    }
    for (hipEvent_t e : h->ev_part) if (e) (void)hipEventDestroy(e);
    delete h->pool;
    if (h->h_scalars) (void)hipHostFree(h->h_scalars);
    for (hipEvent_t e : h->evring) if (e) (void)hipEventDestroy(e);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

// Order in which the Behler kernels visit the angular functions: by (lambda, eta, zeta) -- one squaring ladder per lambda, one exp per
// pair.  perm[pos] = index of the function in the file's order.
static std::vector<int> ni_visit_order(const double *ang, int nt)
{
    std::vector<int> perm(nt);
    for (int m = 0; m < nt; m++) perm[m] = m;
    std::stable_sort(perm.begin(), perm.end(), [&](int x, int y) {
        if (ang[4 * x + 1] != ang[4 * y + 1]) return ang[4 * x + 1] < ang[4 * y + 1];
        if (ang[4 * x] != ang[4 * y]) return ang[4 * x] < ang[4 * y];
        return ang[4 * x + 2] < ang[4 * y + 2];
    });
    return perm;
}

int annp_hip_init(annp_hip_handle **handle, const annp_hip_params *p, int device,
                  int nlocal_hint, int nall_hint, int max_nbors_hint)
{
    if (!handle || !p) return fail(nullptr, ANNP_HIP_EARG, "null argument");
    *handle = nullptr;
    if (p->struct_bytes != (int)sizeof(annp_hip_params))
        return fail(nullptr, ANNP_HIP_EARG, "annp_hip_params size mismatch (%d vs %d)", p->struct_bytes, (int)sizeof(annp_hip_params));
    const int nl = p->ntl - 1;
    const bool anna = p->descriptor == ANNP_HIP_DESC_ANNA_ADP;
    if (anna) {
        if (nl < 1 || nl > ANNA_MAXL || p->nsf < 1 || p->nnod < 1 || p->nnod > 64 || p->npsf + p->ntsf != p->nsf ||
            p->npsf > FE_NP || p->ntsf > FE_NT || p->nout != 2 || p->ngp < 17 || !p->gparams || !p->flagact || !p->weight_all ||
            !p->bias_all)
            return fail(nullptr, ANNP_HIP_ESHAPE, "unsupported anna_adp shape ntl=%d nnod=%d nout=%d nsf=%d (%d+%d) ngp=%d",
                        p->ntl, p->nnod, p->nout, p->nsf, p->npsf, p->ntsf, p->ngp);
    } else if (nl < 2 || nl > MLP_MAXL || p->nsf < 1 || p->nsf > ANNP_GPAD || p->nnod < 1 || p->nnod > 32 ||
        p->npsf + p->ntsf != p->nsf || !p->flagact || !p->sfnor_scal || !p->sfnor_avg || !p->weight_all || !p->bias_all)
        return fail(nullptr, ANNP_HIP_ESHAPE, "unsupported network shape ntl=%d nnod=%d nsf=%d (%d+%d)", p->ntl, p->nnod, p->nsf, p->npsf, p->ntsf);
    if (p->descriptor == ANNP_HIP_DESC_CHEBYSHEV && (p->npsf > FE_NP || p->ntsf > FE_NT))
        return fail(nullptr, ANNP_HIP_ESHAPE, "Chebyshev kernels hold up to %d radial and %d angular orders (got %d %d)", FE_NP, FE_NT, p->npsf, p->ntsf);
    if (p->descriptor == ANNP_HIP_DESC_BEHLER && (!p->cofsymrad || !p->cofsymang))
        return fail(nullptr, ANNP_HIP_EARG, "Behler descriptor needs cofsymrad/cofsymang");
    const int ne = std::max(1, p->nelements);
    const int ntypes = std::max(1, p->ntypes);
    if (ntypes > 30) return fail(nullptr, ANNP_HIP_ESHAPE, "more than 30 atom types");
    unsigned active = 0u;
    bool multi = ne > 1;
    for (int t = 1; t <= ntypes; t++) {
        const int e = p->map ? p->map[t] : 0;
        if (e >= ne) return fail(nullptr, ANNP_HIP_EARG, "map[%d] = %d but the potential has %d element(s)", t, e, ne);
        if (e >= 0) active |= 1u << t; else multi = true;
    }
    if (!active) return fail(nullptr, ANNP_HIP_EARG, "no atom type is mapped to an element");
    if (anna && multi) return fail(nullptr, ANNP_HIP_ESHAPE, "pair_style anna_adp: single-element potentials only");
    // cutsq: cutmax^2 for every pair of mapped types, as init_one produces it (fe_v2/src/pair_annp.cpp:323-327)
    double cutsq_all = p->cut * p->cut;
    if (p->cutsq) {
        cutsq_all = -1.0;
        for (int a = 1; a <= ntypes; a++)
            for (int b = 1; b <= ntypes; b++) {
                if (!((active >> a) & 1u) || !((active >> b) & 1u)) continue;
                const double c = p->cutsq[(size_t)(p->ntypes + 1) * a + b];
                if (cutsq_all < 0.0) cutsq_all = c;
                if (c != cutsq_all || !(c > 0.0))
                    return fail(nullptr, ANNP_HIP_ESHAPE, "cutsq[%d][%d] = %g differs from cutsq of the other mapped pairs (%g)", a, b, c, cutsq_all);
            }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, ANNP_HIP_EDEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(nullptr, ANNP_HIP_EDEVICE, "device %d out of range (%d visible)", device, ndev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(nullptr, ANNP_HIP_EDEVICE, "hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, ANNP_HIP_EDEVICE, "device %d is %s; this library carries gfx950 code objects only", device, prop.gcnArchName);

    annp_hip_handle *h = new (std::nothrow) annp_hip_handle();
    if (!h) return fail(nullptr, ANNP_HIP_ENOMEM, "host allocation failed");
    h->device = device;
    auto bail = [&](int code) { g_init_error = h->err; annp_hip_clear(h); return code; };
#define INIT_TRY(call)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            fail(h, 0, "%s failed: %s", #call, hipGetErrorString(e_));                                   \
            return bail(e_ == hipErrorOutOfMemory ? ANNP_HIP_ENOMEM : ANNP_HIP_EDEVICE);                 \
        }                                                                                                \
    } while (0)
    DeviceGuard guard_(device);
    INIT_TRY(guard_.err);
    INIT_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    INIT_TRY(hipEventCreateWithFlags(&h->ev_flags, hipEventDisableTiming));
    INIT_TRY(hipStreamCreateWithFlags(&h->stream_flags, hipStreamNonBlocking));
    INIT_TRY(hipEventCreateWithFlags(&h->ev_tail, hipEventDisableTiming));
    INIT_TRY(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
    INIT_TRY(hipEventCreateWithFlags(&h->ev_f_up, hipEventDisableTiming));
    if (const char *e = std::getenv("ANNP_HIP_REGISTER")) h->use_register = std::atoi(e) != 0;
    h->descriptor = p->descriptor; h->ntypes = p->ntypes; h->ntl = p->ntl; h->nhl = p->nhl; h->nnod = p->nnod;
    h->nsf = p->nsf; h->npsf = p->npsf; h->ntsf = p->ntsf; h->nl = nl; h->ni_compat = p->ni_compat;
    h->e_scale = p->e_scale; h->e_shift = p->e_shift; h->e_atom = p->e_atom; h->cut = p->cut;
    if (const char *e = std::getenv("ANNP_HIP_FULL_LIST")) h->full_list = std::atoi(e) != 0;
    if (const char *e = std::getenv("ANNP_HIP_NEIGH_SYNC")) h->nb.lazy = std::atoi(e) == 0;
    if (const char *e = std::getenv("ANNP_HIP_NI_PAIRS")) h->ni_no_pairs = std::atoi(e) == 0;
    if (const char *e = std::getenv("ANNP_HIP_NI_FIXUP")) h->ni_no_fixup = std::atoi(e) == 0;
    if (const char *e = std::getenv("ANNP_HIP_FE_DESC")) h->fe_desc_pairs = std::strcmp(e, "pairs") == 0;
    if (const char *e = std::getenv("ANNP_HIP_FE_FORCE")) h->fe_force_pairs = std::strcmp(e, "pairs") == 0;
    if (const char *e = std::getenv("ANNP_HIP_VIRIAL")) h->virial_tally = std::strcmp(e, "tally") == 0;
    if (const char *e = std::getenv("ANNP_HIP_SHF_PLACES")) h->shf_places_by_number = std::strcmp(e, "number") == 0;
    if (const char *e = std::getenv("ANNP_HIP_SH_WPB")) h->sh_wpb = std::min(4, std::max(0, std::atoi(e)));
    if (const char *e = std::getenv("ANNP_HIP_REPLAN_IMAGES")) h->rp_images_by_dimension = std::strcmp(e, "dims") == 0;
    if (const char *e = std::getenv("ANNP_HIP_LIST_PARTS")) h->list_parts = std::max(1, std::min((int)annp_hip_handle::kListParts, std::atoi(e)));
    if (const char *e = std::getenv("ANNP_HIP_LIST_PIPE_MIN")) h->list_pipe_min = std::max(1, std::atoi(e));
    if (const char *e = std::getenv("ANNP_HIP_LIST_CHUNK")) h->list_chunk = std::max<size_t>(1024, std::min<size_t>(annp_hip_handle::kListChunk, (size_t)std::atoll(e)));
    if (const char *e = std::getenv("ANNP_HIP_SH_TAIL")) h->sh_group = std::strcmp(e, "group") == 0;
    if (const char *e = std::getenv("ANNP_HIP_SH_CAP")) h->sh_cap = std::min((int)SH_CAP_MAX, std::max((int)SH_CAP_MIN, round_up(std::atoi(e), 16)));
    h->cutsq = cutsq_all;
    h->nelem = ne; h->multi = multi; h->active = active;
    if (multi) {
        std::vector<int> mp((size_t)ntypes + 1, -1);
        for (int t = 1; t <= ntypes; t++) mp[t] = p->map ? p->map[t] : 0;
        INIT_TRY(hipMalloc((void **)&h->d_map, sizeof(int) * mp.size()));
        INIT_TRY(hipMemcpy(h->d_map, mp.data(), sizeof(int) * mp.size(), hipMemcpyHostToDevice));
        h->bytes += sizeof(int) * mp.size();
    }
    for (int l = 0; l < nl; l++) h->flagact[l] = p->flagact[l];

    if (anna) {
        // network image for annp_anna_adp: layer 0 with its columns in the device feature layout (FE_NP + FE_NT slots)
        h->nout = p->nout; h->e_base = p->e_base;
        for (int k = 0; k < 17; k++) h->gp[k] = p->gparams[k];
        h->nsf_dev = FE_NP + FE_NT;
        std::vector<double> img;
        for (int l = 0; l < nl; l++) {
            const int nr = (l == nl - 1) ? p->nout : p->nnod, nc_file = (l == 0) ? p->nsf : p->nnod, nc = (l == 0) ? h->nsf_dev : p->nnod;
            for (int r = 0; r < nr; r++) {
                std::vector<double> rowv(nc, 0.0);
                for (int c = 0; c < nc_file; c++) {
                    const int q = (l == 0 && c >= p->npsf) ? FE_NP + (c - p->npsf) : c;
                    rowv[q] = p->weight_all[l][(size_t)r * nc_file + c];
                }
                img.insert(img.end(), rowv.begin(), rowv.end());
            }
            img.insert(img.end(), p->bias_all[l], p->bias_all[l] + nr);
        }
        h->net_doubles = (int)img.size();
        INIT_TRY(hipMalloc((void **)&h->d_net, sizeof(double) * img.size()));
        INIT_TRY(hipMemcpy(h->d_net, img.data(), sizeof(double) * img.size(), hipMemcpyHostToDevice));
        h->bytes += sizeof(double) * img.size();
    } else {   // normalisation of the descriptor and the linear map dE/dZ_0 -> coef
        // Device feature layout.  Behler: the file's order.  Chebyshev: the kernels always produce FE_NP radial
        // and FE_NT angular sums, so feature k of a smaller basis sits in slot k (radial) or FE_NP + (k - npsf)
        // (angular) and the remaining slots get zero weights: exact, whatever the basis size.
        const bool cheb = p->descriptor == ANNP_HIP_DESC_CHEBYSHEV;
        const int np_ = cheb ? FE_NP : p->npsf, nt = cheb ? FE_NT : p->ntsf, nnod = p->nnod;
        const int nsf = np_ + nt;
        h->nsf_dev = nsf;
        std::vector<int> slot(p->nsf);
        for (int k = 0; k < p->nsf; k++) slot[k] = (cheb && k >= p->npsf) ? FE_NP + (k - p->npsf) : k;
        std::vector<double> t(3 * ANNP_GPAD, 0.0), cmul(ANNP_GPAD, 0.0);
        for (int k = 0; k < ANNP_GPAD; k++) t[2 * ANNP_GPAD + k] = 1.0;
        for (int k = 0; k < p->nsf; k++) {
            const int q = slot[k];
            if (cheb) {
                // G_k = s_k * sum  (fe:647,678);  Ghat = G - s_k avg_k (fe:178-180);  c_k = e_scale s_k dE/dGhat_k (fe:197)
                t[q] = p->sfnor_scal[k];
                t[ANNP_GPAD + q] = p->sfnor_scal[k] * p->sfnor_avg[k];
                t[2 * ANNP_GPAD + q] = 1.0;
                cmul[q] = p->e_scale * p->sfnor_scal[k];
            } else {
                // Ghat = (G - sf_min)/(sf_max - sf_min) (ni:168-170);  F = -dE/dGhat dG / (sf_max-sf_min) * CFFORCE (ni:186-189)
                t[q] = 1.0;
                t[ANNP_GPAD + q] = p->sfnor_avg[k];
                t[2 * ANNP_GPAD + q] = 1.0 / p->sfnor_scal[k];
                cmul[q] = 1.0 / p->sfnor_scal[k];
            }
        }
        INIT_TRY(hipMalloc((void **)&h->d_norm, sizeof(double) * t.size()));
        INIT_TRY(hipMemcpy(h->d_norm, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice));
        h->bytes += sizeof(double) * t.size();
        // T: rows of coef as linear forms of c_k = cmul_k dE/dGhat_k
        std::vector<long double> T((size_t)ANNP_CPAD * nsf, 0.0L);
        if (cheb) {
            if (np_ + 2 * nt + 1 > ANNP_CPAD) { fail(h, 0, "descriptor too large for the coefficient buffer"); return bail(ANNP_HIP_ESHAPE); }
            // coefficients of z^k in T_n((z+1)/2): T_0 = 1, T_1 = (1+z)/2, T_n = (1+z) T_{n-1} - T_{n-2}.
            // All entries are dyadic rationals below 2^53, so this recurrence is exact.
            std::vector<double> M((size_t)nt * nt, 0.0), a(nt, 0.0), b(nt, 0.0), tt(nt, 0.0);
            a[0] = 1.0;
            for (int k = 0; k < nt; k++) M[(size_t)k * nt + 0] = a[k];
            if (nt > 1) { b[0] = 0.5; b[1] = 0.5; for (int k = 0; k < nt; k++) M[(size_t)k * nt + 1] = b[k]; }
            for (int n = 2; n < nt; n++) {
                for (int k = 0; k < nt; k++) tt[k] = b[k] + (k > 0 ? b[k - 1] : 0.0) - a[k];
                a = b; b = tt;
                for (int k = 0; k < nt; k++) M[(size_t)k * nt + n] = b[k];
            }
            for (int m = 0; m < np_; m++) T[(size_t)m * nsf + m] = 1.0L;                          // radial c_m
            for (int k = 0; k < nt; k++)                                                          // p_k
                for (int n = 0; n < nt; n++) T[(size_t)(np_ + k) * nsf + np_ + n] = M[(size_t)k * nt + n];
            // W_l, the same polynomial in Legendre polynomials (T_n((z+1)/2) = sum_l q_nl P_l(z), sh_tables.hpp), and P(1) = sum_n c_n:
            // what the force pass on the moments wants (fe_shf_kernels.hpp).  (The pair-loop force kernel makes its derivative
            // coefficients (k+1) p_(k+1) itself; rounds 1-3 shipped them in these rows.)
            static const double shq[(SH_LMAX + 1) * (SH_LMAX + 1)] = ANNP_SH_Q_INIT;
            static_assert(FE_NT == SH_LMAX + 1 && FE_NP + 2 * FE_NT + 1 <= ANNP_CPAD, "coefficient row: c_m | p_k | W_l | P(1)");
            for (int l = 0; l < nt; l++)
                for (int n = 0; n < nt; n++) T[(size_t)(np_ + nt + l) * nsf + np_ + n] = shq[n * nt + l];
            for (int n = 0; n < nt; n++) T[(size_t)(np_ + 2 * nt) * nsf + np_ + n] = 1.0L;
        } else {
            // radial weights as they are, angular ones in the order the kernels visit the functions (ni_visit_order: the force
            // pass copies its coefficient rows straight from memory)
            const std::vector<int> perm = ni_visit_order(p->cofsymang, p->ntsf);
            for (int k = 0; k < np_; k++) T[(size_t)k * nsf + k] = 1.0L;
            for (int pos = 0; pos < nt; pos++) T[(size_t)(np_ + pos) * nsf + np_ + perm[pos]] = 1.0L;
        }
        // one operand image per element: its weights (layer 0 in the device layout), biases and coefmat
        std::vector<double> img_all;
        for (int e = 0; e < ne; e++) {
            const double *const *We = p->weight_all + (size_t)e * nl;
            const double *const *Be = p->bias_all + (size_t)e * nl;
            for (int l = 0; l < nl; l++)
                if (!We[l] || !Be[l]) { fail(h, 0, "weight_all / bias_all: element %d layer %d is NULL", e, l); return bail(ANNP_HIP_EARG); }
            std::vector<double> w0((size_t)nnod * nsf, 0.0);
            for (int i = 0; i < nnod; i++)
                for (int k = 0; k < p->nsf; k++) w0[(size_t)i * nsf + slot[k]] = We[0][(size_t)i * p->nsf + k];
            // coefmat[o][i] = sum_k T[o][k] cmul_k W_0[i][k]
            std::vector<double> cm((size_t)ANNP_CPAD * nnod, 0.0);
            for (int o = 0; o < ANNP_CPAD; o++)
                for (int i = 0; i < nnod; i++) {
                    long double acc = 0.0L;
                    for (int k = 0; k < nsf; k++) acc += T[(size_t)o * nsf + k] * (long double)cmul[k] * (long double)w0[(size_t)i * nsf + k];
                    cm[(size_t)o * nnod + i] = (double)acc;
                }
            std::vector<const double *> Wl(We, We + nl);
            Wl[0] = w0.data();
            const std::vector<double> img = mlp_image(h, Wl.data(), Be, cm.data());
            if (img.empty()) { fail(h, 0, "no network kernel for nsf=%d nnod=%d layers=%d", h->nsf, h->nnod, h->nl); return bail(ANNP_HIP_ESHAPE); }
            h->img_stride = img.size();
            img_all.insert(img_all.end(), img.begin(), img.end());
        }
        INIT_TRY(hipMalloc((void **)&h->d_mlp_img, sizeof(double) * img_all.size()));
        INIT_TRY(hipMemcpy(h->d_mlp_img, img_all.data(), sizeof(double) * img_all.size(), hipMemcpyHostToDevice));
        h->bytes += sizeof(double) * img_all.size();
    }
    if (p->descriptor == ANNP_HIP_DESC_BEHLER) {
        h->sym_rad.assign(p->cofsymrad, p->cofsymrad + 3 * p->npsf);
        h->sym_ang.assign(p->cofsymang, p->cofsymang + 4 * p->ntsf);
        // visit order (lambda, eta, zeta): one squaring ladder per lambda, one exp per pair
        const int nt = p->ntsf;
        const std::vector<int> perm = ni_visit_order(p->cofsymang, nt);
        const double *ang = p->cofsymang;
        std::vector<double> etas;
        std::vector<int> eidx(nt), zint(nt);
        for (int pos = 0; pos < nt; pos++) {
            const int m = perm[pos];
            size_t e = 0;
            for (; e < etas.size(); e++) if (etas[e] == ang[4 * m]) break;
            if (e == etas.size()) etas.push_back(ang[4 * m]);
            eidx[pos] = (int)e;
            const double z = ang[4 * m + 2];
            zint[pos] = (z >= 0.0 && z < 32.0 && z == std::floor(z)) ? (int)z : -1;
            if (zint[pos] < 0) { fail(h, 0, "angular function %d: zeta = %g is not an integer in [0,32)", m, z); return bail(ANNP_HIP_ESHAPE); }
        }
        if ((int)etas.size() > NI_MAXE) { fail(h, 0, "more than %d distinct eta values in the angular functions", NI_MAXE); return bail(ANNP_HIP_ESHAPE); }
        {   // is the set a full product {lambda} x {eta} x {zeta}?  (visit order is then (l*ne + e)*nz + z)
            std::vector<double> lams, zets;
            for (int pos = 0; pos < nt; pos++) {
                const int m = perm[pos];
                if (std::find(lams.begin(), lams.end(), ang[4 * m + 1]) == lams.end()) lams.push_back(ang[4 * m + 1]);
                if (std::find(zets.begin(), zets.end(), ang[4 * m + 2]) == zets.end()) zets.push_back(ang[4 * m + 2]);
            }
            const int nl_ = (int)lams.size(), ne_ = (int)etas.size(), nz_ = (int)zets.size();
            bool prod = nl_ * ne_ * nz_ == nt;
            std::sort(zets.begin(), zets.end());
            for (int pos = 0; pos < nt && prod; pos++) {
                const int m = perm[pos];
                const int l = pos / (ne_ * nz_), e = (pos / nz_) % ne_, z = pos % nz_;
                prod = ang[4 * m + 1] == lams[l] && ang[4 * m] == etas[e] && ang[4 * m + 2] == zets[z];
            }
            h->ni_shape = prod ? NiShape{nl_, ne_, nz_, 0u, 0u} : NiShape{0, 0, 0, 0u, 0u};
            for (int k = 0; k < 4; k++) { h->ni_lam[k] = k < nl_ ? lams[k] : 0.0; h->ni_eta[k] = k < ne_ ? etas[k] : 0.0; }   // visit order (kernel arguments)
            if (prod && nz_ <= 4)
                for (int z = 0; z < nz_; z++) h->ni_shape.zp |= (unsigned)zint[z] << (8 * z);    // visit positions 0..nz-1: l = e = 0
        }
        std::vector<int> emult(NI_MAXE, 0);
        for (size_t e = 1; e < etas.size(); e++) {
            const double k = etas[e] / etas[0];
            const double kr = std::floor(k + 0.5);
            if (kr >= 1.0 && kr <= 64.0 && std::fabs(kr * etas[0] - etas[e]) <= 4e-16 * std::fabs(etas[e])) emult[e] = (int)kr;
        }
        h->ni_shape.em = 1u;
        for (size_t e = 1; e < etas.size() && e < 4; e++) h->ni_shape.em |= (unsigned)emult[e] << (8 * e);
        h->ni_rad_em = p->npsf > 0 ? 1ull : 0ull;
        for (int m = 1; m < p->npsf && m < 8; m++) {
            const double e0 = p->cofsymrad[0], em_ = p->cofsymrad[3 * m];
            const double kr = e0 > 0.0 ? std::floor(em_ / e0 + 0.5) : 0.0;
            if (kr >= 1.0 && kr <= 64.0 && std::fabs(kr * e0 - em_) <= 4e-16 * std::fabs(em_)) h->ni_rad_em |= (unsigned long long)kr << (8 * m);
        }
        std::vector<double> t(h->sym_rad);
        t.insert(t.end(), h->sym_ang.begin(), h->sym_ang.end());
        for (int pos = 0; pos < nt; pos++) {
            const int m = perm[pos];
            t.push_back(ang[4 * m]); t.push_back(ang[4 * m + 1]); t.push_back(ang[4 * m + 2]);
            t.push_back(std::pow(2.0, 1.0 - ang[4 * m + 2]));                 // term_coe, ni:748
        }
        for (int e = 0; e < NI_MAXE; e++) t.push_back(e < (int)etas.size() ? etas[e] : 0.0);
        std::vector<int> it(perm);
        it.insert(it.end(), eidx.begin(), eidx.end());
        it.insert(it.end(), zint.begin(), zint.end());
        it.push_back((int)etas.size());
        it.insert(it.end(), emult.begin(), emult.end());
        INIT_TRY(hipMalloc((void **)&h->d_sym, sizeof(double) * t.size()));
        INIT_TRY(hipMemcpy(h->d_sym, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice));
        INIT_TRY(hipMalloc((void **)&h->d_isym, sizeof(int) * it.size()));
        INIT_TRY(hipMemcpy(h->d_isym, it.data(), sizeof(int) * it.size(), hipMemcpyHostToDevice));
        h->bytes += sizeof(double) * t.size() + sizeof(int) * it.size();
    }
    INIT_TRY(hipMalloc((void **)&h->d_scalars, 8 * sizeof(double)));
    INIT_TRY(hipMalloc((void **)&h->d_vslots, sizeof(double) * 8 * ANNP_VSLOTS));
    INIT_TRY(hipMalloc((void **)&h->d_flags, 3 * ANNP_NFLAGS * sizeof(int)));       // the sticky word's row and two sets of per-evaluation words
    INIT_TRY(hipMemset(h->d_flags, 0, 3 * ANNP_NFLAGS * sizeof(int)));
    h->fw = h->d_flags + ANNP_NFLAGS;
    INIT_TRY(hipEventCreateWithFlags(&h->ev_set[0], hipEventDisableTiming));
    INIT_TRY(hipEventCreateWithFlags(&h->ev_set[1], hipEventDisableTiming));
    INIT_TRY(hipHostMalloc((void **)&h->h_flags, ANNP_NFLAGS * sizeof(int)));
    INIT_TRY(hipHostMalloc((void **)&h->h_scalars, 8 * sizeof(double)));
    h->bytes += 8 * sizeof(double) + ANNP_NFLAGS * sizeof(int) + sizeof(double) * 8 * ANNP_VSLOTS;
    // kernels may ask for the whole LDS
    {
        const int full = 160 * 1024;
        INIT_TRY(hipFuncSetAttribute((const void *)annp_fe_desc<FE_NP, FE_NT>, hipFuncAttributeMaxDynamicSharedMemorySize, full));
        INIT_TRY(hipFuncSetAttribute((const void *)annp_fe_desc_sh<FE_NP, FE_NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, full));
        INIT_TRY(hipFuncSetAttribute((const void *)annp_fe_desc_sh<FE_NP, FE_NT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, full));
        INIT_TRY(hipFuncSetAttribute((const void *)annp_fe_desc_fixup<FE_NP, FE_NT>, hipFuncAttributeMaxDynamicSharedMemorySize, full));
        INIT_TRY(hipFuncSetAttribute((const void *)annp_fe_force_sh<FE_NP, FE_NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, full));
        INIT_TRY(hipFuncSetAttribute((const void *)annp_fe_force_sh<FE_NP, FE_NT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, full));
        INIT_TRY(hipFuncSetAttribute((const void *)annp_fe_force<FE_NP, FE_NT, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, full));
        INIT_TRY(hipFuncSetAttribute((const void *)annp_fe_force<FE_NP, FE_NT, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, full));
        INIT_TRY(hipFuncSetAttribute((const void *)annp_fe_force<FE_NP, FE_NT, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, full));
        INIT_TRY(hipFuncSetAttribute((const void *)annp_fe_force<FE_NP, FE_NT, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, full));
        INIT_TRY(hipFuncSetAttribute((const void *)annp_fe_force<FE_NP, FE_NT, true, true, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, full));
        INIT_TRY(hipFuncSetAttribute((const void *)annp_fe_force<FE_NP, FE_NT, false, true, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, full));
        INIT_TRY(hipFuncSetAttribute((const void *)annp_fe_force_fixup<FE_NP, FE_NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, full));
        INIT_TRY(hipFuncSetAttribute((const void *)annp_fe_force_fixup<FE_NP, FE_NT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, full));
        INIT_TRY(ni_set_lds_attributes());
        INIT_TRY(hipFuncSetAttribute((const void *)annp_anna_adp<true>, hipFuncAttributeMaxDynamicSharedMemorySize, full));
        INIT_TRY(hipFuncSetAttribute((const void *)annp_anna_adp<false>, hipFuncAttributeMaxDynamicSharedMemorySize, full));
    }
    // sizing hints, as annp_gpu_init takes them (buffers still grow on demand)
    if (nlocal_hint > 0) {
        if (ensure(h, h->G, (size_t)nlocal_hint * ANNP_GPAD) || ensure(h, h->coef, (size_t)(nlocal_hint + SHF_GA) * ANNP_CPAD, true) ||
            ensure(h, h->ncount, (size_t)nlocal_hint))
            return bail(ANNP_HIP_ENOMEM);
    }
    (void)nall_hint; (void)max_nbors_hint;
#undef INIT_TRY
    *handle = h;
    return ANNP_HIP_OK;
}

int annp_hip_set_timing(annp_hip_handle *h, int enable)
{
    if (!h) return ANNP_HIP_EARG;
    DEVICE_GUARD(h);
    if (enable && h->evring.empty()) {
        h->evring.assign((size_t)annp_hip_handle::kRing * 4, nullptr);
        for (hipEvent_t &e : h->evring) HIP_TRY(h, hipEventCreate(&e));
    }
    h->timing = enable != 0;
    h->ev_count = 0;
    return 0;
}

static int timing_slot(annp_hip_handle *h, long long k, double *ms4)
{
    hipEvent_t *ev = h->evring.data() + 4 * (size_t)(k % annp_hip_handle::kRing);
    HIP_TRY(h, hipEventSynchronize(ev[3]));
    float a = 0, b = 0, c = 0, d = 0;
    HIP_TRY(h, hipEventElapsedTime(&a, ev[0], ev[1]));
    HIP_TRY(h, hipEventElapsedTime(&b, ev[1], ev[2]));
    HIP_TRY(h, hipEventElapsedTime(&c, ev[2], ev[3]));
    HIP_TRY(h, hipEventElapsedTime(&d, ev[0], ev[3]));
    ms4[0] = a; ms4[1] = b; ms4[2] = c; ms4[3] = d;
    return 0;
}

int annp_hip_last_timing(annp_hip_handle *h, double *ms4)
{
    if (!h || !ms4) return ANNP_HIP_EARG;
    if (h->ev_count == 0) return fail(h, ANNP_HIP_EARG, "no timed evaluation recorded");
    DEVICE_GUARD(h);
    return timing_slot(h, h->ev_count - 1, ms4);
}

int annp_hip_timing_stats(annp_hip_handle *h, double *ms4_mean, int *nsamples)
{
    if (!h || !ms4_mean) return ANNP_HIP_EARG;
    if (h->ev_count == 0) return fail(h, ANNP_HIP_EARG, "no timed evaluation recorded");
    DEVICE_GUARD(h);
    const long long n = std::min<long long>(h->ev_count, annp_hip_handle::kRing);
    double acc[4] = {0, 0, 0, 0};
    for (long long k = h->ev_count - n; k < h->ev_count; k++) {
        double t[4];
        int rc = timing_slot(h, k, t);
        if (rc) return rc;
        for (int q = 0; q < 4; q++) acc[q] += t[q];
    }
    for (int q = 0; q < 4; q++) ms4_mean[q] = acc[q] / (double)n;
    if (nsamples) *nsamples = (int)n;
    return 0;
}

int annp_hip_last_counts(annp_hip_handle *h, int *counts, int inum)
{
    if (!h || !counts || inum < 0 || (size_t)inum > h->ncount.cap) return h ? fail(h, ANNP_HIP_EARG, "last_counts: bad argument") : ANNP_HIP_EARG;
    DEVICE_GUARD(h);
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(counts, h->ncount.p, sizeof(int) * (size_t)inum, hipMemcpyDeviceToHost));
    return 0;
}

int annp_hip_last_descriptors(annp_hip_handle *h, double *rows, int inum)
{
    if (!h || !rows || inum < 0 || (size_t)inum * ANNP_GPAD > h->G.cap) return h ? fail(h, ANNP_HIP_EARG, "last_descriptors: bad argument") : ANNP_HIP_EARG;
    DEVICE_GUARD(h);
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(rows, h->G.p, sizeof(double) * ANNP_GPAD * (size_t)inum, hipMemcpyDeviceToHost));
    return 0;
}

#ifdef ANNP_POISON_LDS
// developer builds only (make poison): does the fill of annp_common.hpp's ANNP_POISON() cover the workgroup's allocation?  A kernel with
// `bytes` of dynamic LDS poisons, then copies its first and last words and one in the middle out.  Returns 0 when all three hold the pattern.
__global__ void annp_poison_probe(unsigned long long *out, int bytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    ANNP_POISON();
    const unsigned long long *w = reinterpret_cast<const unsigned long long *>(lds_raw);
    if (threadIdx.x == 0) { out[0] = w[0]; out[1] = w[bytes / 16]; out[2] = w[bytes / 8 - 1]; }
}
extern "C" int annp_hip_poison_selftest(int bytes)
{
    unsigned long long *d = nullptr, hres[3] = {0, 0, 0};
    if (hipMalloc((void **)&d, sizeof(hres)) != hipSuccess) return -1;
    (void)hipFuncSetAttribute((const void *)annp_poison_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(annp_poison_probe, dim3(4), dim3(256), (size_t)bytes, nullptr, d, bytes);
    if (hipMemcpy(hres, d, sizeof(hres), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipFree(d); return -2; }
    (void)hipFree(d);
    const double pat = ANNP_POISON_LDS == 2 ? __builtin_bit_cast(double, ~0ull) : -1e300;
    const unsigned long long want = __builtin_bit_cast(unsigned long long, pat);
    return (hres[0] == want && hres[1] == want && hres[2] == want) ? 0 : 1;
}
#endif

int annp_hip_sync(annp_hip_handle *h)
{
    if (!h) return ANNP_HIP_EARG;
    DEVICE_GUARD(h);
    HIP_TRY(h, hipDeviceSynchronize());
    {   // a list build whose row maximum nobody has looked at yet (neigh_kernels.hpp: lazy builds)
        std::string msg;
        if (int rc = neigh_settle(h->nb, msg)) return fail(h, rc, "%s", msg.c_str());
    }
    return poll_flags(h, true);
}

int annp_hip_eval_path(annp_hip_handle *h)
{
    if (!h) return ANNP_HIP_EARG;
    DEVICE_GUARD(h);
    if (h->flags_pending) {
        HIP_TRY(h, hipEventSynchronize(h->ev_flags));
        h->flags_pending = false;
        digest_flags(h);
    }
    if (h->descriptor == ANNP_HIP_DESC_BEHLER) return 3;
    if (h->descriptor == ANNP_HIP_DESC_ANNA_ADP) return 4;
    if (h->fe_desc_pairs || h->fe_force_pairs) return 2;
    return h->fe_dense ? 1 : 0;
}

int annp_hip_set_notice(annp_hip_handle *h, void *file)
{
    if (!h) return ANNP_HIP_EARG;
    h->notice = static_cast<FILE *>(file);
    if (h->notice && (h->fe_desc_pairs || h->fe_force_pairs) && h->descriptor == ANNP_HIP_DESC_CHEBYSHEV) {
        std::fprintf(h->notice, "annp/hip: ANNP_HIP_FE_DESC / ANNP_HIP_FE_FORCE = pairs: the Chebyshev passes run pair by pair (developer switch, about half the speed)\n");
        std::fflush(h->notice);
    }
    return 0;
}

int annp_hip_eval_info(annp_hip_handle *h, int *info4)
{
    if (!h || !info4) return ANNP_HIP_EARG;
    DEVICE_GUARD(h);
    if (h->flags_pending) {     // the error, if any, stays for the next call or annp_hip_sync to report
        HIP_TRY(h, hipEventSynchronize(h->ev_flags));
        h->flags_pending = false;
        digest_flags(h);
    }
    for (int k = 0; k < 4; k++) info4[k] = h->info[k];
    return 0;
}

int annp_hip_compute_device(annp_hip_handle *h, int inum, int nall,
                            const double *d_x, const int *d_type, const int *d_ilist,
                            const int *d_numneigh, const long long *d_first, const int *d_neigh,
                            int max_numneigh,
                            double *d_f, double *d_eatom, double *d_eng, double *d_virial, double *d_vatom, void *stream)
{
    if (!h) return ANNP_HIP_EARG;
    if (inum < 0 || nall < inum || !d_x || !d_f || (inum > 0 && (!d_numneigh || !d_first || !d_neigh)))
        return fail(h, ANNP_HIP_EARG, "annp_hip_compute_device: bad argument");
    DEVICE_GUARD(h);
    return compute_device_impl(h, inum, nall, d_x, d_type, d_ilist, d_numneigh, d_first, d_neigh, max_numneigh,
                               d_f, d_eatom, d_eng, d_virial, d_vatom, (hipStream_t)stream);
}

int annp_hip_neigh_build_device(annp_hip_handle *h, int nlocal, int nall, const double *d_x, double cutneigh,
                                const int **d_numneigh, const long long **d_first, const int **d_neigh,
                                int *max_numneigh, void *stream)
{
    if (!h || !d_x || nlocal < 0 || nall < nlocal || cutneigh <= 0) return h ? fail(h, ANNP_HIP_EARG, "neigh_build: bad argument") : ANNP_HIP_EARG;
    DEVICE_GUARD(h);
    std::string msg;
    size_t before = h->nb.bytes;
    int rc = neigh_build(h->nb, nlocal, nall, d_x, list_cutoff(h, cutneigh), (hipStream_t)stream, msg, true);
    h->bytes += h->nb.bytes - before;
    if (rc) return fail(h, rc, "%s", msg.c_str());
    if (d_numneigh) *d_numneigh = h->nb.numneigh;
    if (d_first) *d_first = h->nb.first;
    if (d_neigh) *d_neigh = h->nb.neigh;
    if (max_numneigh) *max_numneigh = h->nb.max_numneigh;
    return 0;
}

double annp_hip_list_cutoff(const annp_hip_handle *h, double cutneigh) { return h ? list_cutoff(h, cutneigh) : cutneigh; }

int annp_hip_list_layout(const annp_hip_handle *h, int *info4)
{
    if (!h || !info4) return ANNP_HIP_EARG;
    info4[0] = (h->nb.valid && h->nb.pitched) ? 1 : 0;
    info4[1] = h->nb.valid ? (h->nb.pitched ? h->nb.pitch_used : 0) : 0;
    info4[2] = h->nb.valid ? h->nb.max_numneigh : 0;
    info4[3] = h->nb.valid ? h->nb.nlocal : 0;
    return 0;
}

int annp_hip_neigh_to_host(annp_hip_handle *h, int nlocal, int *numneigh, long long *first, int *neigh,
                           long long neigh_capacity, long long *total_out)
{
    if (!h || nlocal < 0 || !numneigh) return h ? fail(h, ANNP_HIP_EARG, "neigh_to_host: bad argument") : ANNP_HIP_EARG;
    if (!h->nb.valid || h->nb.nlocal != nlocal)
        return fail(h, ANNP_HIP_EARG, "neigh_to_host: no device-built list for %d atoms on this handle", nlocal);
    DEVICE_GUARD(h);
    HIP_TRY(h, hipDeviceSynchronize());
    if (nlocal == 0) { if (total_out) *total_out = 0; return 0; }
    HIP_TRY(h, hipMemcpy(numneigh, h->nb.numneigh, sizeof(int) * (size_t)nlocal, hipMemcpyDeviceToHost));
    long long tot = 0;
    for (int i = 0; i < nlocal; i++) { if (first) first[i] = tot; tot += numneigh[i]; }
    if (first) first[nlocal] = tot;
    if (total_out) *total_out = tot;
    if (!neigh) return 0;
    if (neigh_capacity < tot) return fail(h, ANNP_HIP_EARG, "neigh_to_host: room for %lld entries, the list has %lld", neigh_capacity, tot);
    if (!h->nb.pitched) {           // exact CSR on the device: rows already packed in atom order
        HIP_TRY(h, hipMemcpy(neigh, h->nb.neigh, sizeof(int) * (size_t)tot, hipMemcpyDeviceToHost));
        return 0;
    }
    // rows sit `pitch` entries apart on the device: whole rows come over in chunks through the pinned staging buffers
    // of the list upload, and a chunk is packed by the copy threads while the next one is on the wire
    const size_t pitch = (size_t)h->nb.pitch_used;
    int rc;
    if ((rc = ensure_list_staging(h))) return rc;
    std::vector<long long> offs;
    const long long *off = first;
    if (!off) {
        offs.resize((size_t)nlocal + 1);
        long long t = 0;
        for (int i = 0; i < nlocal; i++) { offs[i] = t; t += numneigh[i]; }
        offs[nlocal] = t;
        off = offs.data();
    }
    const int rows_per_chunk = (int)std::max<size_t>(1, annp_hip_handle::kListChunk / std::max<size_t>(pitch, 1));
    if (pitch > annp_hip_handle::kListChunk) return fail(h, ANNP_HIP_ENEIGHCAP, "a list row has %zu entries", pitch);
    hipStream_t s = h->stream;
    const int nchunks = (nlocal + rows_per_chunk - 1) / rows_per_chunk;
    auto pack = [&](int c) -> int {
        const int b = c % annp_hip_handle::kListBufs;
        HIP_TRY(h, hipEventSynchronize(h->ev_list[b]));
        const int i0 = c * rows_per_chunk, i1 = std::min(nlocal, i0 + rows_per_chunk);
        const int *src = h->pin_list[b];
        h->pool->run([=](int part, int nparts) {
            const int span = i1 - i0, lo = i0 + (int)((long long)span * part / nparts), hi = i0 + (int)((long long)span * (part + 1) / nparts);
            for (int i = lo; i < hi; i++)
                if (numneigh[i] > 0) std::memcpy(neigh + off[i], src + (size_t)(i - i0) * pitch, sizeof(int) * (size_t)numneigh[i]);
        });
        return 0;
    };
    for (int c = 0; c < nchunks; c++) {
        const int b = c % annp_hip_handle::kListBufs;
        const int i0 = c * rows_per_chunk, i1 = std::min(nlocal, i0 + rows_per_chunk);
        HIP_TRY(h, hipMemcpyAsync(h->pin_list[b], h->nb.neigh + (size_t)i0 * pitch, sizeof(int) * (size_t)(i1 - i0) * pitch, hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipEventRecord(h->ev_list[b], s));
        if (c > 0 && (rc = pack(c - 1))) return rc;     // chunk c is in flight while chunk c-1 is packed; with 3 buffers chunk c+1 never lands on one still being read
    }
    if (nchunks > 0 && (rc = pack(nchunks - 1))) return rc;
    return 0;
}

// ---- the halo wire: RCCL point-to-point, called from here (loader: see rccl() above) ---------------------------------
int annp_hip_comm_unique_id(char *id128)
{
    if (!id128) return ANNP_HIP_EARG;
    Rccl &r = rccl();
    if (!r.err.empty()) return fail(nullptr, ANNP_HIP_EDEVICE, "%s", r.err.c_str());
    ncclUniqueId id;
    if (r.GetUniqueId(&id) != ncclSuccess) return fail(nullptr, ANNP_HIP_EDEVICE, "ncclGetUniqueId failed");
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(id128, &id, sizeof(id));
    return 0;
}

int annp_hip_comm_init(annp_hip_handle *h, const char *id128, int world, int rank)
{
    if (!h || !id128 || world < 1 || rank < 0 || rank >= world) return h ? fail(h, ANNP_HIP_EARG, "comm_init: bad argument") : ANNP_HIP_EARG;
    Rccl &r = rccl();
    if (!r.err.empty()) return fail(h, ANNP_HIP_EDEVICE, "%s", r.err.c_str());
    DEVICE_GUARD(h);
    if (h->comm) { (void)r.CommDestroy(h->comm); h->comm = nullptr; }
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    RCCL_TRY(h, r.CommInitRank(&h->comm, world, id, rank));
    h->comm_world = world; h->comm_rank = rank;
    return 0;
}

int annp_hip_comm_route(annp_hip_handle *h, int nmsg, const int *is_send, double *const *d_buf, const long long *ndoubles,
                        const int *peer, void *stream)
{
    if (!h || nmsg < 0 || (nmsg > 0 && (!is_send || !d_buf || !ndoubles || !peer))) return h ? fail(h, ANNP_HIP_EARG, "comm_route: bad argument") : ANNP_HIP_EARG;
    if (!h->comm) return fail(h, ANNP_HIP_EARG, "comm_route: annp_hip_comm_init has not been called on this handle");
    if (nmsg == 0) return 0;
    for (int k = 0; k < nmsg; k++)
        if (peer[k] < 0 || peer[k] >= h->comm_world || ndoubles[k] < 0 || (ndoubles[k] > 0 && !d_buf[k]))
            return fail(h, ANNP_HIP_EARG, "comm_route: message %d: peer %d, %lld doubles", k, peer[k], ndoubles[k]);
    Rccl &r = rccl();
    DEVICE_GUARD(h);
    hipStream_t s = (hipStream_t)stream;
    RCCL_TRY(h, r.GroupStart());
    for (int k = 0; k < nmsg; k++) {
        if (ndoubles[k] == 0) continue;
        const ncclResult_t e = is_send[k] ? r.Send(d_buf[k], (size_t)ndoubles[k], ncclDouble, peer[k], h->comm, s)
                                          : r.Recv(d_buf[k], (size_t)ndoubles[k], ncclDouble, peer[k], h->comm, s);
        if (e != ncclSuccess) { (void)r.GroupEnd(); return fail(h, ANNP_HIP_EDEVICE, "ncclSend/ncclRecv failed: %s", r.GetErrorString(e)); }
    }
    RCCL_TRY(h, r.GroupEnd());
    return 0;
}

int annp_hip_comm_destroy(annp_hip_handle *h)
{
    if (!h) return ANNP_HIP_EARG;
    if (h->comm) {
        DeviceGuard guard_(h->device);
        (void)hipDeviceSynchronize();
        (void)rccl().CommDestroy(h->comm);
        h->comm = nullptr; h->comm_world = 0; h->comm_rank = -1;
    }
    return 0;
}

// ---- the step around the evaluation (step_kernels.hpp): device pointers, asynchronous on `stream` -------------------
int annp_hip_halo_pack(annp_hip_handle *h, int n, const int *d_idx, const double *d_shift, const double *d_x, double *d_out, void *stream)
{
    if (!h || n < 0 || (n > 0 && (!d_idx || !d_shift || !d_x || !d_out))) return h ? fail(h, ANNP_HIP_EARG, "halo_pack: bad argument") : ANNP_HIP_EARG;
    if (n == 0) return 0;
    DEVICE_GUARD(h);
    hipLaunchKernelGGL(annp_gather_shift, dim3((3 * n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, d_idx, d_shift, d_x, d_out);
    HIP_TRY(h, hipGetLastError());
    return 0;
}

int annp_hip_halo_unpack_images(annp_hip_handle *h, int nimg, const int *d_root, const double *d_shift, double *d_x, int first_image_row,
                                double *d_f_clear, long long n_clear, double *d_eng_clear, void *stream)
{
    if (!h || nimg < 0 || first_image_row < 0 || n_clear < 0 || (nimg > 0 && (!d_root || !d_shift || !d_x)))
        return h ? fail(h, ANNP_HIP_EARG, "halo_unpack_images: bad argument") : ANNP_HIP_EARG;
    const long long work = std::max<long long>(std::max<long long>(3ll * nimg, d_f_clear ? n_clear : 0), d_eng_clear ? 1 : 0);
    if (work == 0) return 0;
    DEVICE_GUARD(h);
    hipLaunchKernelGGL(annp_images_clear, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream, nimg, d_root, d_shift, d_x,
                       (long long)first_image_row, d_f_clear, n_clear, d_eng_clear);
    HIP_TRY(h, hipGetLastError());
    return 0;
}

int annp_hip_reverse_fold(annp_hip_handle *h, int nseg, const int *d_seg_dst, const int *d_seg_start, const int *d_perm,
                          const double *d_src, double *d_f, void *stream)
{
    if (!h || nseg < 0 || (nseg > 0 && (!d_seg_start || !d_perm || !d_src || !d_f)))
        return h ? fail(h, ANNP_HIP_EARG, "reverse_fold: bad argument") : ANNP_HIP_EARG;
    if (nseg == 0) return 0;
    DEVICE_GUARD(h);
    hipLaunchKernelGGL(annp_segment_add, dim3((3 * nseg + 255) / 256), dim3(256), 0, (hipStream_t)stream, nseg, d_seg_dst, d_seg_start, d_perm, d_src, d_f);
    HIP_TRY(h, hipGetLastError());
    return 0;
}


// ---- re-planning on the device (Comm::exchange + Comm::borders; replan_kernels.hpp) ------------------------------------
namespace {
int rp_scratch(annp_hip_handle *h, size_t nflags, size_t npos)
{
    int rc;
    if ((rc = ensure(h, h->rp_flag, nflags)) || (rc = ensure(h, h->rp_pos, npos)) || (rc = ensure(h, h->rp_bs, npos / 1024 + 8))) return rc;
    if (!h->rp_tot) {
        HIP_TRY(h, hipMalloc((void **)&h->rp_tot, 8 * sizeof(long long)));
        HIP_TRY(h, hipHostMalloc((void **)&h->rp_tot_h, 8 * sizeof(long long)));
    }
    return 0;
}
// exclusive scan of flags[0..n) into pos, the total into tot[slot]
void rp_scan(annp_hip_handle *h, const int *flags, int n, long long *pos, long long *bs, int slot, hipStream_t s)
{
    const int nblk = std::max(1, (n + 1023) / 1024);
    hipLaunchKernelGGL(annp_scan_block_sums, dim3(nblk), dim3(1024), 0, s, flags, n, bs);
    hipLaunchKernelGGL(annp_scan_block_offsets, dim3(1), dim3(1024), 0, s, bs, nblk, h->rp_tot + slot);
    hipLaunchKernelGGL(annp_scan_finish, dim3(nblk), dim3(1024), 0, s, flags, n, bs, pos);
}
int rp_totals(annp_hip_handle *h, int count, hipStream_t s)
{
    HIP_TRY(h, hipMemcpyAsync(h->rp_tot_h, h->rp_tot, sizeof(long long) * count, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return 0;
}
ReplanBox rp_box(const double *box6, const int *periodic3)
{
    ReplanBox b;
    for (int d = 0; d < 3; d++) { b.lo[d] = box6[d]; b.hi[d] = box6[3 + d]; b.periodic[d] = periodic3[d]; }
    return b;
}
}  // namespace

int annp_hip_replan_exchange(annp_hip_handle *h, int n, double *d_x, const long long *d_ids, const double *d_e0, int w0, const double *d_e1, int w1,
                             const double *box6, const int *periodic3, int world, int rank, int has_left, int has_right,
                             double *d_keep, double *d_send, int *counts3, void *stream)
{
    if (!h || n < 0 || !box6 || !periodic3 || world < 1 || w0 < 0 || w1 < 0 || (n > 0 && !d_x) || (world > 1 && (!d_ids || !d_keep || !d_send || !counts3)) ||
        (world > 1 && ((w0 > 0 && !d_e0) || (w1 > 0 && !d_e1))))           // (annp_replan_pack reads w0 / w1 columns of them)
        return h ? fail(h, ANNP_HIP_EARG, "replan_exchange: bad argument") : ANNP_HIP_EARG;
    if (counts3) { counts3[0] = n; counts3[1] = counts3[2] = 0; }
    if (n == 0) return 0;
    DEVICE_GUARD(h);
    hipStream_t s = (hipStream_t)stream;
    const ReplanBox b = rp_box(box6, periodic3);
    const int blocks = (n + 255) / 256;
    if (world == 1) {        // nothing to send: the wrap alone
        hipLaunchKernelGGL(annp_replan_wrap_classify, dim3(blocks), dim3(256), 0, s, n, d_x, b, 1, 0, 0, 0, (int *)nullptr, (int *)nullptr, (int *)nullptr, (int *)nullptr);
        HIP_TRY(h, hipGetLastError());
        return 0;
    }
    int rc;
    if ((rc = rp_scratch(h, 3 * (size_t)n + 8, 3 * (size_t)n + 8))) return rc;
    int *fs = h->rp_flag.p, *fl = fs + n, *fr = fl + n;
    long long *ps = h->rp_pos.p, *pl = ps + n, *pr = pl + n;
    HIP_TRY(h, hipMemsetAsync(h->rp_tot, 0, 8 * sizeof(long long), s));
    int *bad = reinterpret_cast<int *>(h->rp_tot + 3);
    hipLaunchKernelGGL(annp_replan_wrap_classify, dim3(blocks), dim3(256), 0, s, n, d_x, b, world, rank, has_left, has_right, fs, fl, fr, bad);
    rp_scan(h, fs, n, ps, h->rp_bs.p, 0, s);
    rp_scan(h, fl, n, pl, h->rp_bs.p, 1, s);
    rp_scan(h, fr, n, pr, h->rp_bs.p, 2, s);
    HIP_TRY(h, hipGetLastError());
    if ((rc = rp_totals(h, 4, s))) return rc;
    const long long ns = h->rp_tot_h[0], nl = h->rp_tot_h[1], nr = h->rp_tot_h[2];
    if ((h->rp_tot_h[3] & 0xffffffffll) != 0 || ns + nl + nr != n)
        return fail(h, ANNP_HIP_EARG, "replan_exchange: an atom moved further than the neighbouring slab between two rebuilds");
    counts3[0] = (int)ns; counts3[1] = (int)nl; counts3[2] = (int)nr;
    hipLaunchKernelGGL(annp_replan_pack, dim3(blocks), dim3(256), 0, s, n, fs, ps, 0ll, d_x, d_ids, d_e0, w0, d_e1, w1, d_keep);
    hipLaunchKernelGGL(annp_replan_pack, dim3(blocks), dim3(256), 0, s, n, fl, pl, 0ll, d_x, d_ids, d_e0, w0, d_e1, w1, d_send);
    hipLaunchKernelGGL(annp_replan_pack, dim3(blocks), dim3(256), 0, s, n, fr, pr, nl, d_x, d_ids, d_e0, w0, d_e1, w1, d_send);
    HIP_TRY(h, hipGetLastError());
    return 0;
}

int annp_hip_replan_unpack(annp_hip_handle *h, int m, const double *d_rows, int w0, int w1, double *d_x, long long *d_ids, double *d_e0, double *d_e1, void *stream)
{
    if (!h || m < 0 || w0 < 0 || w1 < 0 || (m > 0 && (!d_rows || !d_x || !d_ids || (w0 > 0 && !d_e0) || (w1 > 0 && !d_e1))))
        return h ? fail(h, ANNP_HIP_EARG, "replan_unpack: bad argument") : ANNP_HIP_EARG;
    if (m == 0) return 0;
    DEVICE_GUARD(h);
    hipLaunchKernelGGL(annp_replan_unpack, dim3((m + 255) / 256), dim3(256), 0, (hipStream_t)stream, m, w0, w1, d_rows, d_x, d_ids, d_e0, d_e1);
    HIP_TRY(h, hipGetLastError());
    return 0;
}

int annp_hip_replan_faces(annp_hip_handle *h, int n, const double *d_x, double lo_edge, double hi_edge, int has_left, int has_right,
                          int *d_idx, int *counts2, void *stream)
{
    if (!h || n < 0 || !counts2 || (n > 0 && (!d_x || !d_idx))) return h ? fail(h, ANNP_HIP_EARG, "replan_faces: bad argument") : ANNP_HIP_EARG;
    counts2[0] = counts2[1] = 0;
    if (n == 0 || (!has_left && !has_right)) return 0;
    DEVICE_GUARD(h);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if ((rc = rp_scratch(h, 2 * (size_t)n + 8, 2 * (size_t)n + 8))) return rc;
    int *fl = h->rp_flag.p, *fr = fl + n;
    long long *pl = h->rp_pos.p, *pr = pl + n;
    const int blocks = (n + 255) / 256;
    hipLaunchKernelGGL(annp_replan_face_flags, dim3(blocks), dim3(256), 0, s, n, d_x, lo_edge, hi_edge, has_left, has_right, fl, fr);
    rp_scan(h, fl, n, pl, h->rp_bs.p, 0, s);
    rp_scan(h, fr, n, pr, h->rp_bs.p, 1, s);
    HIP_TRY(h, hipGetLastError());
    if ((rc = rp_totals(h, 2, s))) return rc;
    const long long nl = h->rp_tot_h[0], nr = h->rp_tot_h[1];
    counts2[0] = (int)nl; counts2[1] = (int)nr;
    hipLaunchKernelGGL(annp_replan_scatter_idx, dim3(blocks), dim3(256), 0, s, n, fl, pl, 0ll, d_idx);
    hipLaunchKernelGGL(annp_replan_scatter_idx, dim3(blocks), dim3(256), 0, s, n, fr, pr, nl, d_idx);
    HIP_TRY(h, hipGetLastError());
    return 0;
}

int annp_hip_replan_images(annp_hip_handle *h, int np0, double *d_x, long long capacity_rows, const double *box6, const int *periodic3, double rc_halo,
                           int dims_mask, int *d_root, double *d_shift, int *nimg_out, void *stream)
{
    if (!h || np0 < 0 || !box6 || !periodic3 || !nimg_out || capacity_rows < np0 || (np0 > 0 && !d_x))
        return h ? fail(h, ANNP_HIP_EARG, "replan_images: bad argument") : ANNP_HIP_EARG;
    *nimg_out = 0;
    if (np0 == 0) return 0;
    DEVICE_GUARD(h);
    hipStream_t s = (hipStream_t)stream;
    // With buffers to fill (the usual call) the dimensions follow one another without the host looking at their counts in between
    // (round 6: one wait per call instead of one per dimension): the kernels take the number of rows held so far from device memory
    // and cover the buffer's capacity; a dimension that does not fit writes nothing and the caller is told how many rows were wanted.
    if (d_root && d_shift && capacity_rows <= 0x7fffffffll / 3 && !h->rp_images_by_dimension) {
        int rc;
        const int cap = (int)capacity_rows;
        if ((rc = rp_scratch(h, 2 * (size_t)cap + 8, 2 * (size_t)cap + 8))) return rc;
        int *flo = h->rp_flag.p, *fhi = flo + cap;
        long long *plo = h->rp_pos.p, *phi = plo + cap;
        long long *state = h->rp_tot + 4;
        h->rp_tot_h[6] = np0; h->rp_tot_h[7] = 0;          // (pinned: the copy may run after this line; the call ends with a wait, so nobody rewrites them before it has)
        HIP_TRY(h, hipMemcpyAsync(state, h->rp_tot_h + 6, 2 * sizeof(long long), hipMemcpyHostToDevice, s));
        const int blocks = (cap + 255) / 256;
        for (int d = 0; d < 3; d++) {
            if (!((dims_mask >> d) & 1) || !periodic3[d]) continue;
            hipLaunchKernelGGL(annp_replan_image_flags_dev, dim3(blocks), dim3(256), 0, s, cap, state, d_x, d, box6[d] + rc_halo, box6[3 + d] - rc_halo, flo, fhi);
            rp_scan(h, flo, cap, plo, h->rp_bs.p, 0, s);
            rp_scan(h, fhi, cap, phi, h->rp_bs.p, 1, s);
            hipLaunchKernelGGL(annp_replan_image_make_dev, dim3(blocks), dim3(256), 0, s, cap, state, capacity_rows, np0, d, box6[3 + d] - box6[d], flo, plo, fhi, phi,
                               h->rp_tot + 0, d_x, d_root, d_shift);
            hipLaunchKernelGGL(annp_replan_image_advance, dim3(1), dim3(64), 0, s, state, h->rp_tot + 0, capacity_rows);
        }
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipMemcpyAsync(h->rp_tot_h + 4, state, 2 * sizeof(long long), hipMemcpyDeviceToHost, s));
        HIP_TRY(h, hipStreamSynchronize(s));
        if (h->rp_tot_h[5] > 0) {       // not enough room: rows wanted by the first dimension that did not fit, times what the later ones may add
            *nimg_out = (int)std::min<long long>(0x7fffffffll, (h->rp_tot_h[5] - np0) * 3);
            return ANNP_HIP_ENEIGHCAP;
        }
        *nimg_out = (int)(h->rp_tot_h[4] - np0);
        return 0;
    }
    long long cur = np0;
    for (int d = 0; d < 3; d++) {
        if (!((dims_mask >> d) & 1) || !periodic3[d]) continue;
        int rc;
        if (cur > 0x7fffffffll / 3) return fail(h, ANNP_HIP_EARG, "replan_images: too many rows");
        if ((rc = rp_scratch(h, 2 * (size_t)cur + 8, 2 * (size_t)cur + 8))) return rc;
        int *flo = h->rp_flag.p, *fhi = flo + cur;
        long long *plo = h->rp_pos.p, *phi = plo + cur;
        const int blocks = (int)((cur + 255) / 256);
        hipLaunchKernelGGL(annp_replan_image_flags, dim3(blocks), dim3(256), 0, s, (int)cur, d_x, d, box6[d] + rc_halo, box6[3 + d] - rc_halo, flo, fhi);
        rp_scan(h, flo, (int)cur, plo, h->rp_bs.p, 0, s);
        rp_scan(h, fhi, (int)cur, phi, h->rp_bs.p, 1, s);
        HIP_TRY(h, hipGetLastError());
        if ((rc = rp_totals(h, 2, s))) return rc;
        const long long add = h->rp_tot_h[0] + h->rp_tot_h[1];
        if (cur + add > capacity_rows || !d_root || !d_shift) {
            // not enough room (or a sizing call): count the remaining dimensions as if every row so far had images on both sides is
            // not knowable without making them, so the caller is told what this dimension needs and asked to come back with more
            *nimg_out = (int)std::min<long long>(0x7fffffffll, (cur + add - np0) * ((d == 0) ? 9 : (d == 1 ? 3 : 1)));
            return ANNP_HIP_ENEIGHCAP;
        }
        if (add > 0) {
            hipLaunchKernelGGL(annp_replan_image_make, dim3(blocks), dim3(256), 0, s, (int)cur, np0, d, box6[3 + d] - box6[d], flo, plo, fhi, phi,
                               h->rp_tot + 0, d_x, d_root, d_shift);
            HIP_TRY(h, hipGetLastError());
        }
        cur += add;
    }
    *nimg_out = (int)(cur - np0);
    return 0;
}

int annp_hip_replan_fold_plan(annp_hip_handle *h, int m, const int *d_targets, int nkeys, int *d_start, int *d_perm, void *stream)
{
    if (!h || m < 0 || nkeys < 0 || (nkeys > 0 && !d_start) || (m > 0 && (!d_targets || !d_perm)))
        return h ? fail(h, ANNP_HIP_EARG, "replan_fold_plan: bad argument") : ANNP_HIP_EARG;
    if (nkeys == 0) return 0;
    DEVICE_GUARD(h);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if ((rc = ensure(h, h->rp_cnt, 2 * (size_t)nkeys + 8)) || (rc = rp_scratch(h, 8, (size_t)nkeys + 8))) return rc;
    int *cnt = h->rp_cnt.p, *cursor = cnt + nkeys + 4;
    HIP_TRY(h, hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)nkeys, s));
    if (m > 0) hipLaunchKernelGGL(annp_replan_count, dim3((m + 255) / 256), dim3(256), 0, s, m, d_targets, nkeys, cnt, h->d_flags);
    rp_scan(h, cnt, nkeys, h->rp_pos.p, h->rp_bs.p, 0, s);
    HIP_TRY(h, hipMemcpyAsync(h->rp_pos.p + nkeys, h->rp_tot, sizeof(long long), hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(annp_replan_start32, dim3((nkeys + 256) / 256), dim3(256), 0, s, nkeys, h->rp_pos.p, d_start, cursor);
    if (m > 0) {
        hipLaunchKernelGGL(annp_replan_fill, dim3((m + 255) / 256), dim3(256), 0, s, m, d_targets, nkeys, cursor, d_perm);
        hipLaunchKernelGGL(annp_replan_sort_segments, dim3((nkeys + 255) / 256), dim3(256), 0, s, nkeys, d_start, d_perm);
    }
    HIP_TRY(h, hipGetLastError());
    return 0;
}

int annp_hip_verlet_half(annp_hip_handle *h, int n, double *d_x, double *d_v, const double *d_f, double dtf, double dt, void *stream)
{
    if (!h || n < 0 || (n > 0 && (!d_v || !d_f || (dt != 0.0 && !d_x)))) return h ? fail(h, ANNP_HIP_EARG, "verlet_half: bad argument") : ANNP_HIP_EARG;
    if (n == 0) return 0;
    DEVICE_GUARD(h);
    const long long n3 = 3ll * n;
    hipLaunchKernelGGL(annp_verlet_half, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n3, d_x, d_v, d_f, dtf, dt);
    HIP_TRY(h, hipGetLastError());
    return 0;
}

// ---- host-pointer entry points --------------------------------------------------------
static int ensure_pinned(annp_hip_handle *h, double *&p, size_t &cap, size_t n)
{
    if (n <= cap) return 0;
    if (p) { (void)hipHostFree(p); h->bytes -= cap * sizeof(double); p = nullptr; cap = 0; }
    const size_t want = n + n / 8;
    hipError_t e = hipHostMalloc((void **)&p, want * sizeof(double));
    if (e != hipSuccess) { p = nullptr; return fail(h, ANNP_HIP_ENOMEM, "hipHostMalloc(%zu bytes) failed: %s", want * sizeof(double), hipGetErrorString(e)); }
    cap = want;
    h->bytes += cap * sizeof(double);
    return 0;
}

// The caller's arrays (LAMMPS atom->x[0], atom->f[0]) are page-locked where they lie, once: LAMMPS reallocates them only
// when nmax grows, so the registration is cached by address and length and redone when either changes.  A registered array
// is read and written by the copy engine directly (no bounce through a pinned buffer, no host loop over it).  When the
// runtime refuses (or ANNP_HIP_REGISTER=0) the pinned staging buffers of the handle are used instead.
static bool host_register(annp_hip_handle *h, annp_hip_handle::HostReg &r, const void *ptr, size_t bytes)
{
    if (!h->use_register || !ptr || bytes == 0) return false;
    if (r.ptr == ptr && r.bytes >= bytes) return r.ok;
    // (the old array may be gone already -- LAMMPS reallocated it, which is why the address changed -- and the runtime then
    // answers "not registered": nothing to undo, but the per-thread last-error word must not keep it)
    if (r.ok) { if (hipHostUnregister(const_cast<void *>(r.ptr)) != hipSuccess) (void)hipGetLastError(); r.ok = false; }
    r.ptr = ptr; r.bytes = bytes;
    const hipError_t e = hipHostRegister(const_cast<void *>(ptr), bytes, hipHostRegisterDefault);
    if (e != hipSuccess) (void)hipGetLastError();
    r.ok = e == hipSuccess;
    return r.ok;
}

// dst[k] += src[k] (or dst[k] = src[k]) over n doubles, split over the copy threads
static void host_fold(annp_hip_handle *h, double *dst, const double *src, size_t n, bool add)
{
    auto job = [=](int part, int nparts) {
        const size_t lo = n * (size_t)part / (size_t)nparts, hi = n * (size_t)(part + 1) / (size_t)nparts;
        if (add) for (size_t k = lo; k < hi; k++) dst[k] += src[k];
        else std::memcpy(dst + lo, src + lo, sizeof(double) * (hi - lo));
    };
    if (h->pool && n >= (size_t)1 << 16) h->pool->run(job); else job(0, 1);
}

// positions -> h->x
static int upload_x(annp_hip_handle *h, const double *host_x, int nall, hipStream_t s)
{
    int rc;
    const size_t n = (size_t)nall * 3;
    if ((rc = ensure(h, h->x, n))) return rc;
    if (n == 0) return 0;
    if (host_register(h, h->reg_x, host_x, n * sizeof(double))) {
        HIP_TRY(h, hipMemcpyAsync(h->x.p, host_x, n * sizeof(double), hipMemcpyHostToDevice, s));
        return 0;
    }
    if ((rc = ensure_pool(h)) || (rc = ensure_pinned(h, h->pin_x, h->pin_x_cap, n))) return rc;
    host_fold(h, h->pin_x, host_x, n, false);
    HIP_TRY(h, hipMemcpyAsync(h->x.p, h->pin_x, n * sizeof(double), hipMemcpyHostToDevice, s));
    return 0;
}

// Results of the evaluation on h->stream -> the caller's arrays (+= semantics, fe:185-211).  Nothing of the caller's is
// touched before the evaluation is known to be complete (a Behler capacity error makes the caller re-issue it).
static int host_finish(annp_hip_handle *h, int inum, int nall, int eflag, int vflag, int eatom_flag, bool f_on_device,
                       double *f, double *eng_vdwl, double *eatom, double *virial, double *vatom)
{
    hipStream_t s = h->stream;
    int rc;
    HIP_TRY(h, hipMemcpyAsync(h->h_scalars, h->d_scalars, 8 * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    if (int rcf = poll_flags(h, true)) return rcf;
    const size_t nf = (size_t)nall * 3;
    if (f_on_device) {          // h->f started from the caller's f: the sum comes back into place
        HIP_TRY(h, hipMemcpyAsync(f, h->f.p, sizeof(double) * nf, hipMemcpyDeviceToHost, s));
    } else {
        if ((rc = ensure_pinned(h, h->pin_f, h->pin_f_cap, nf))) return rc;
        HIP_TRY(h, hipMemcpyAsync(h->pin_f, h->f.p, sizeof(double) * nf, hipMemcpyDeviceToHost, s));
    }
    const bool want_e = eflag && eatom_flag && eatom;
    if (want_e) {
        if ((rc = ensure_pinned(h, h->pin_e, h->pin_e_cap, (size_t)nall))) return rc;
        HIP_TRY(h, hipMemcpyAsync(h->pin_e, h->eatom.p, sizeof(double) * nall, hipMemcpyDeviceToHost, s));
    }
    if (vatom) {
        if ((rc = ensure_pinned(h, h->pin_v, h->pin_v_cap, (size_t)nall * 6))) return rc;
        HIP_TRY(h, hipMemcpyAsync(h->pin_v, h->vatom.p, sizeof(double) * nall * 6, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(h, hipStreamSynchronize(s));
    if (!f_on_device) host_fold(h, f, h->pin_f, nf, true);                         // fe:199,211: += / -=
    if (vatom) host_fold(h, vatom, h->pin_v, (size_t)nall * 6, true);
    if (want_e) host_fold(h, eatom, h->pin_e, (size_t)nall, true);                 // fe:186
    if (eflag && eng_vdwl) *eng_vdwl += h->h_scalars[0];                           // fe:185
    if (vflag && virial) for (int k = 0; k < 6; k++) virial[k] += h->h_scalars[1 + k];
    (void)inum;
    return 0;
}

// Positions are on the device (h->x); run one evaluation on the handle's stream and bring the results back.
// A Behler capacity overflow is not an error here: the call is synchronous anyway, so it runs the evaluation
// again with the raised capacity.
static int host_evaluate(annp_hip_handle *h, int inum, int nall, const int *host_type, const int *d_ilist,
                         const int *d_numneigh, const long long *d_first, const int *d_neigh, int max_numneigh,
                         int eflag, int vflag, int eatom_flag,
                         double *f, double *eng_vdwl, double *eatom, double *virial, double *vatom)
{
    hipStream_t s = h->stream;
    int rc;
    if ((rc = ensure_pool(h))) return rc;
    if ((rc = ensure(h, h->f, (size_t)nall * 3)) || (rc = ensure(h, h->eatom, (size_t)nall))) return rc;
    if (vatom && (rc = ensure(h, h->vatom, (size_t)nall * 6))) return rc;
    const int *d_type = nullptr;
    if (h->multi) {             // atom types select the element's network (and drop atoms of unmapped types)
        if (!host_type) return fail(h, ANNP_HIP_EARG, "this potential distinguishes atom types: host_type is required");
        for (int k = 0; k < nall; k++)          // map[type] and the type's bit of `active` are indexed with it on the device
            if (host_type[k] < 1 || host_type[k] > h->ntypes)
                return fail(h, ANNP_HIP_EARG, "type[%d] = %d is outside 1..%d", k, host_type[k], h->ntypes);
        if ((rc = ensure(h, h->type, (size_t)nall))) return rc;
        HIP_TRY(h, hipMemcpyAsync(h->type.p, host_type, sizeof(int) * (size_t)nall, hipMemcpyHostToDevice, s));
        d_type = h->type.p;
    }
    const bool want_eatom = eflag && eatom_flag && eatom;
    const size_t nf = (size_t)nall * 3;
    // f: the device accumulates on top of the caller's values, uploaded on a second stream while the descriptor and
    // network passes run (the force pass waits for it), and the sum is copied back into place -- no host loop over f
    const bool f_on_device = nall > 0 && host_register(h, h->reg_f, f, nf * sizeof(double));
    for (int attempt = 0;; attempt++) {
        if (f_on_device) {
            HIP_TRY(h, hipMemcpyAsync(h->f.p, f, nf * sizeof(double), hipMemcpyHostToDevice, h->stream2));
            HIP_TRY(h, hipEventRecord(h->ev_f_up, h->stream2));
            h->pre_force_wait = h->ev_f_up;
        } else {
            HIP_TRY(h, hipMemsetAsync(h->f.p, 0, sizeof(double) * nf, s));
        }
        HIP_TRY(h, hipMemsetAsync(h->d_scalars, 0, 8 * sizeof(double), s));
        if (want_eatom) HIP_TRY(h, hipMemsetAsync(h->eatom.p, 0, sizeof(double) * (size_t)nall, s));
        if (vatom) HIP_TRY(h, hipMemsetAsync(h->vatom.p, 0, sizeof(double) * (size_t)nall * 6, s));
        rc = compute_device_impl(h, inum, nall, h->x.p, d_type, d_ilist, d_numneigh, d_first, d_neigh, max_numneigh,
                                 h->f.p, want_eatom ? h->eatom.p : nullptr, h->d_scalars, (vflag && virial) ? h->d_scalars + 1 : nullptr,
                                 vatom ? h->vatom.p : nullptr, s);
        if (h->pre_force_wait) {        // the evaluation returned before its force pass (error, inum == 0): the upload still has to land
            (void)hipStreamWaitEvent(s, h->pre_force_wait, 0);
            h->pre_force_wait = nullptr;
        }
        if (!rc) rc = host_finish(h, inum, nall, eflag, vflag, eatom_flag, f_on_device, f, eng_vdwl, eatom, virial, vatom);
        else (void)hipStreamSynchronize(s);
        if (rc == ANNP_HIP_ENEIGHCAP && attempt == 0 && h->descriptor == ANNP_HIP_DESC_BEHLER && !h->ni_primed) continue;
        return rc;
    }
}

// LAMMPS' list (ilist, numneigh[i], firstneigh[i] pointing into its pages) -> CSR on the device.
// 230 M entries at 1 M atoms: the rows are packed into pinned staging buffers of 32 MB by a few host threads and each
// buffer goes out with its own asynchronous copy, so packing chunk c + 1 overlaps the transfer of chunk c; the
// headers take the same road.  (A single-threaded pack into pageable memory and one blocking copy cost 250 ms.)
// `parts` > 1 with a hook: the list is cut into that many runs of whole chunks, and when the copies of a run are enqueued (on `s`) the hook
// is called with the run's range of list slots [ii0, ii1) and an event recorded behind its last copy -- the caller starts the evaluation
// of those atoms on another stream while the host packs the next run (annp_hip_compute, round 6).
typedef std::function<int(int, int, hipEvent_t)> ListPartHook;
static int upload_host_list(annp_hip_handle *h, int inum, int nall, const int *ilist, const int *numj,
                            const int *const *firstneigh, hipStream_t s, int parts = 1, const ListPartHook &hook = ListPartHook())
{
    int rc;
    if ((size_t)nall + 1 > h->pin_hdr_cap) {
        if (h->pin_first) { (void)hipHostFree(h->pin_first); h->pin_first = nullptr; }
        if (h->pin_num) { (void)hipHostFree(h->pin_num); h->pin_num = nullptr; }
        const size_t want = (size_t)nall + 1 + (size_t)nall / 8;
        HIP_TRY(h, hipHostMalloc((void **)&h->pin_first, want * sizeof(long long)));
        HIP_TRY(h, hipHostMalloc((void **)&h->pin_num, want * sizeof(int)));
        h->pin_hdr_cap = want;
    }
    if ((rc = ensure_list_staging(h))) return rc;
    // headers: offsets in ilist order
    std::memset(h->pin_first, 0, sizeof(long long) * ((size_t)nall + 1));
    std::memset(h->pin_num, 0, sizeof(int) * (size_t)nall);
    long long tot = 0;
    int mx = 0;
    for (int ii = 0; ii < inum; ii++) {
        const int i = ilist[ii];
        if (i < 0 || i >= nall) return fail(h, ANNP_HIP_EARG, "ilist[%d]=%d out of range", ii, i);
        h->pin_first[i] = tot;
        h->pin_num[i] = numj[i];
        tot += numj[i];
        mx = std::max(mx, numj[i]);
    }
    if (mx > (int)h->list_chunk) return fail(h, ANNP_HIP_ENEIGHCAP, "a list row has %d entries", mx);
    if ((rc = ensure(h, h->first, (size_t)nall + 1)) || (rc = ensure(h, h->numneigh, (size_t)nall)) ||
        (rc = ensure(h, h->neigh, (size_t)std::max<long long>(tot, 1))) || (rc = ensure(h, h->ilist, (size_t)std::max(inum, 1))))
        return rc;
    HIP_TRY(h, hipMemcpyAsync(h->first.p, h->pin_first, sizeof(long long) * ((size_t)nall + 1), hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(h->numneigh.p, h->pin_num, sizeof(int) * (size_t)nall, hipMemcpyHostToDevice, s));
    if (inum > 0) HIP_TRY(h, hipMemcpyAsync(h->ilist.p, ilist, sizeof(int) * (size_t)inum, hipMemcpyHostToDevice, s));
    h->list_max = mx;           // (the hook evaluates before this function returns)
    // rows, chunk by chunk (whole rows per chunk)
    int ii0 = 0, chunk = 0;
    const long long part_fill = parts > 1 ? (tot + parts - 1) / parts : tot + 1;      // entries per run of chunks, about
    long long part_done = 0;
    int part_ii0 = 0, part_k = 0;
    while (ii0 < inum) {
        const long long base = h->pin_first[ilist[ii0]];
        int ii1 = ii0;
        long long fill = 0;
        while (ii1 < inum && fill + numj[ilist[ii1]] <= (long long)h->list_chunk) fill += numj[ilist[ii1++]];
        const int b = chunk % annp_hip_handle::kListBufs;
        if (chunk >= annp_hip_handle::kListBufs) HIP_TRY(h, hipEventSynchronize(h->ev_list[b]));      // its previous copy has left
        int *dst = h->pin_list[b];
        const long long *pf = h->pin_first;
        h->pool->run([=](int part, int nparts) {
            const int span = ii1 - ii0, lo = ii0 + (int)((long long)span * part / nparts), hi = ii0 + (int)((long long)span * (part + 1) / nparts);
            for (int ii = lo; ii < hi; ii++) {
                const int i = ilist[ii];
                if (numj[i] > 0) std::memcpy(dst + (pf[i] - base), firstneigh[i], sizeof(int) * (size_t)numj[i]);
            }
        });
        if (fill > 0) HIP_TRY(h, hipMemcpyAsync(h->neigh.p + base, dst, sizeof(int) * (size_t)fill, hipMemcpyHostToDevice, s));
        HIP_TRY(h, hipEventRecord(h->ev_list[b], s));
        ii0 = ii1;
        chunk++;
        part_done += fill;
        if (hook && (ii0 >= inum || part_done >= part_fill)) {      // a run is on its way: its atoms can be evaluated behind this event
            if (!h->ev_part[part_k % annp_hip_handle::kListParts]) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_part[part_k % annp_hip_handle::kListParts], hipEventDisableTiming));
            hipEvent_t ev = h->ev_part[part_k % annp_hip_handle::kListParts];
            HIP_TRY(h, hipEventRecord(ev, s));
            if ((rc = hook(part_ii0, ii0, ev))) { (void)hipStreamSynchronize(s); return rc; }
            part_ii0 = ii0; part_done = 0; part_k++;
        }
    }
    HIP_TRY(h, hipStreamSynchronize(s));     // ilist and the staging buffers are the caller's / reused
    h->list_max = mx;
    h->list_valid = true;
    return 0;
}

// annp_hip_compute with a list to upload (ago == 0: LAMMPS rebuilt its list; `package gpu ... neigh no`, the mode of the reference's own
// deck): 0.92 GB of rows for 1 M atoms take 25 ms to pack and send, and rounds 1-5 started the 10 ms evaluation when the last byte had
// landed.  The passes work atom by atom, so the list goes out in a few runs of chunks on the second stream and the evaluation of a
// run's atoms is enqueued on the first as soon as the run's copies are -- behind an event, while the host packs the next run: all but
// the last run's evaluation hides behind the upload.  Forces, energy and virial accumulate over the runs as they do over two calls.
static int host_evaluate_uploading(annp_hip_handle *h, int inum, int nall, const int *host_type, const int *ilist, const int *numj,
                                   const int *const *firstneigh, int eflag, int vflag, int eatom_flag,
                                   double *f, double *eng_vdwl, double *eatom, double *virial, double *vatom)
{
    hipStream_t s = h->stream;
    int rc;
    if ((rc = ensure_pool(h))) return rc;
    if ((rc = ensure(h, h->f, (size_t)nall * 3)) || (rc = ensure(h, h->eatom, (size_t)nall))) return rc;
    if (vatom && (rc = ensure(h, h->vatom, (size_t)nall * 6))) return rc;
    const int *d_type = nullptr;
    if (h->multi) {
        if (!host_type) return fail(h, ANNP_HIP_EARG, "this potential distinguishes atom types: host_type is required");
        for (int k = 0; k < nall; k++)
            if (host_type[k] < 1 || host_type[k] > h->ntypes)
                return fail(h, ANNP_HIP_EARG, "type[%d] = %d is outside 1..%d", k, host_type[k], h->ntypes);
        if ((rc = ensure(h, h->type, (size_t)nall))) return rc;
        HIP_TRY(h, hipMemcpyAsync(h->type.p, host_type, sizeof(int) * (size_t)nall, hipMemcpyHostToDevice, s));
        d_type = h->type.p;
    }
    const bool want_eatom = eflag && eatom_flag && eatom;
    const size_t nf = (size_t)nall * 3;
    const bool f_on_device = nall > 0 && host_register(h, h->reg_f, f, nf * sizeof(double));
    if (f_on_device) {          // (ahead of the list on the second stream: the first run's force pass waits for it)
        HIP_TRY(h, hipMemcpyAsync(h->f.p, f, nf * sizeof(double), hipMemcpyHostToDevice, h->stream2));
        HIP_TRY(h, hipEventRecord(h->ev_f_up, h->stream2));
        h->pre_force_wait = h->ev_f_up;
    } else {
        HIP_TRY(h, hipMemsetAsync(h->f.p, 0, sizeof(double) * nf, s));
    }
    HIP_TRY(h, hipMemsetAsync(h->d_scalars, 0, 8 * sizeof(double), s));
    if (want_eatom) HIP_TRY(h, hipMemsetAsync(h->eatom.p, 0, sizeof(double) * (size_t)nall, s));
    if (vatom) HIP_TRY(h, hipMemsetAsync(h->vatom.p, 0, sizeof(double) * (size_t)nall * 6, s));
    h->list_valid = false;
    rc = upload_host_list(h, inum, nall, ilist, numj, firstneigh, h->stream2, h->list_parts, [&](int a, int b, hipEvent_t ev) -> int {
        HIP_TRY(h, hipStreamWaitEvent(s, ev, 0));
        return compute_device_impl(h, b - a, nall, h->x.p, d_type, h->ilist.p + a, h->numneigh.p, h->first.p, h->neigh.p, h->list_max,
                                   h->f.p, want_eatom ? h->eatom.p : nullptr, h->d_scalars, (vflag && virial) ? h->d_scalars + 1 : nullptr,
                                   vatom ? h->vatom.p : nullptr, s);
    });
    if (h->pre_force_wait) {        // no run got as far as its force pass: the upload of f still has to land before anything else happens to it
        (void)hipStreamWaitEvent(s, h->pre_force_wait, 0);
        h->pre_force_wait = nullptr;
    }
    if (rc) { (void)hipStreamSynchronize(s); return rc; }
    return host_finish(h, inum, nall, eflag, vflag, eatom_flag, f_on_device, f, eng_vdwl, eatom, virial, vatom);
}

int annp_hip_compute(annp_hip_handle *h, int ago, int inum, int nall, int nghost,
                     const double *host_x, const int *host_type,
                     const int *ilist, const int *numj, const int *const *firstneigh,
                     int eflag, int vflag, int eatom_flag, int vatom_flag,
                     double *f, double *eng_vdwl, double *eatom, double *virial, double *vatom)
{
    if (!h) return ANNP_HIP_EARG;
    if (inum < 0 || nall < inum || nghost < 0 || !host_x || !f || (inum > 0 && (!ilist || !numj || !firstneigh)))
        return fail(h, ANNP_HIP_EARG, "annp_hip_compute: bad argument");
    const bool want_vatom = vatom_flag && vatom;
    (void)host_type;
    DEVICE_GUARD(h);
    hipStream_t s = h->stream;
    int rc;
    // neighbour list: re-packed to CSR and uploaded when LAMMPS rebuilt it (ago == 0)
    if (ago == 0 || !h->list_valid) {
        // a list of some size, and nothing that would make the evaluation start over (a Behler handle sizes its first evaluation by trying)
        if (h->list_parts > 1 && inum >= h->list_pipe_min && !(h->descriptor == ANNP_HIP_DESC_BEHLER && !h->ni_primed)) {
            if ((rc = upload_x(h, host_x, nall, s))) return rc;
            return host_evaluate_uploading(h, inum, nall, host_type, ilist, numj, firstneigh, eflag, vflag, eatom_flag, f, eng_vdwl, eatom, virial,
                                           want_vatom ? vatom : nullptr);
        }
        if ((rc = upload_host_list(h, inum, nall, ilist, numj, firstneigh, s))) return rc;
    }
    if ((rc = upload_x(h, host_x, nall, s))) return rc;
    return host_evaluate(h, inum, nall, host_type, h->ilist.p, h->numneigh.p, h->first.p, h->neigh.p, h->list_max,
                         eflag, vflag, eatom_flag, f, eng_vdwl, eatom, virial, want_vatom ? vatom : nullptr);
}

int annp_hip_compute_n(annp_hip_handle *h, int ago, int inum, int nall, int nghost,
                       const double *host_x, const int *host_type,
                       const double *sublo, const double *subhi, double cutneigh,
                       int eflag, int vflag, int eatom_flag, int vatom_flag,
                       double *f, double *eng_vdwl, double *eatom, double *virial, double *vatom)
{
    if (!h) return ANNP_HIP_EARG;
    if (inum < 0 || nall < inum || nghost < 0 || !host_x || !f || cutneigh <= 0)
        return fail(h, ANNP_HIP_EARG, "annp_hip_compute_n: bad argument");
    const bool want_vatom = vatom_flag && vatom;
    (void)host_type; (void)sublo; (void)subhi;
    DEVICE_GUARD(h);
    hipStream_t s = h->stream;
    int rc;
    if ((rc = upload_x(h, host_x, nall, s))) return rc;
    if (ago == 0 || !h->nb.valid || h->nb.nlocal != inum || h->nb.nall != nall) {
        std::string msg;
        size_t before = h->nb.bytes;
        rc = neigh_build(h->nb, inum, nall, h->x.p, list_cutoff(h, cutneigh), s, msg);
        h->bytes += h->nb.bytes - before;
        if (rc) return fail(h, rc, "%s", msg.c_str());
    }
    return host_evaluate(h, inum, nall, host_type, nullptr, h->nb.numneigh, h->nb.first, h->nb.neigh, h->nb.max_numneigh,
                         eflag, vflag, eatom_flag, f, eng_vdwl, eatom, virial, want_vatom ? vatom : nullptr);
}

}  // extern "C"
