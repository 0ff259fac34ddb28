// neigh_kernels.hpp -- binned full neighbour list built on the device.
//
// Plays the role of the reference's device-side neighbouring
// (annp_gpu_compute_n -> lal_base_annp.cpp build_nbor_list, which calls LAMMPS
// lib/gpu's Neighbor class): full list, r^2 <= cutneigh^2, owned atoms first,
// ghosts after, no special-bond exclusions.  Output is CSR (numneigh, first, neigh);
// neighbours of an atom come out in a fixed order (bins in raster order, atoms in
// a bin by ascending index), so a rebuild of the same positions is bit-identical.
#pragma once
#include <string>

#include "annp_common.hpp"

namespace annp {

struct NeighBuild {
    // outputs
    int *numneigh = nullptr;
    long long *first = nullptr;
    int *neigh = nullptr;
    int max_numneigh = 0;
    int nlocal = 0, nall = 0;
    bool valid = false;
    int pitch = 0;                  // row pitch learned from the last build (0 = none yet): next build tries one pass
    bool tile_attr = false;         // the bin-tile kernels were allowed their dynamic LDS on this device
    bool pitched = false;           // layout of the current list: rows `pitch_used` apart (else exact CSR)
    int pitch_used = 0;
    double mean_exact = 0.0;        // list entries per atom found by the last exact (two-pass) build
    // Steady state without a host round trip (round 5): the bins of the last build that looked at the bounding box are reused
    // (bin_of clamps: an atom beyond them lands in the outermost bin, whose 27-bin search still covers its neighbours) and the
    // pitched build's row maximum is not waited for.  Both come back through pinned words and are looked at by the NEXT build:
    // a box that outgrew the bins by more than half a bin recomputes them, a row that outgrew the pitch -- its count is clamped on
    // the device, so nothing is indexed beyond a row -- is an error then (and the next build is an exact one).
    bool lazy = true;               // ANNP_HIP_NEIGH_SYNC=1 turns it off
    bool have_geom = false;
    double geom_lo[3] = {0, 0, 0}, geom_cut = 0.0;
    int geom_nb[3] = {0, 0, 0};
    double *h_bbox = nullptr;       // pinned: lo[3], hi[3] of the last build
    long long *h_max = nullptr;     // pinned: row maximum of the last pitched build
    hipEvent_t ev_lazy = nullptr;   // both have arrived
    bool pending = false;           // a lazy build's words have not been looked at yet
    int pending_pitch = 0;
    // scratch
    int *binof = nullptr, *bincount = nullptr, *binstart = nullptr, *binfill = nullptr, *binitems = nullptr;
    long long *blocksum = nullptr;
    double *xs = nullptr;           // positions in bin order: xs[k] = x[binitems[k]], read contiguously by the passes
    double *bbox = nullptr;         // device: lo[3], hi[3]
    int *dmax = nullptr;
    // element capacity of every array above (each grows on its own)
    size_t cap_numneigh = 0, cap_first = 0, cap_neigh = 0, cap_binof = 0, cap_binitems = 0;
    size_t cap_bincount = 0, cap_binstart = 0, cap_binfill = 0, cap_blocksum = 0, cap_xs = 0;
    size_t bytes = 0;
};

inline void neigh_release(NeighBuild &nb)
{
    void *ptrs[] = {nb.numneigh, nb.first, nb.neigh, nb.binof, nb.bincount, nb.binstart, nb.binfill, nb.binitems, nb.blocksum, nb.bbox, nb.dmax, nb.xs};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    if (nb.h_bbox) (void)hipHostFree(nb.h_bbox);
    if (nb.h_max) (void)hipHostFree(nb.h_max);
    if (nb.ev_lazy) (void)hipEventDestroy(nb.ev_lazy);
    const bool lazy = nb.lazy;
    nb = NeighBuild();
    nb.lazy = lazy;
}

// max over an int array: one atomic per block (per-wave atomics on the one result word serialise: 4096 of them cost 40 us)
__global__ __launch_bounds__(256) void annp_max_int(const int *v, int n, int *out)
{
    __shared__ int part[4];
    ANNP_POISON();
    int m = 0;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) m = max(m, v[k]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(part[0], part[1]), max(part[2], part[3]));
        if (m > 0) atomicMax(out, m);
    }
}
inline int annp_max_int_blocks(int n) { return std::max(1, std::min(256, (n + 255) / 256)); }

// bounding box in two steps: per-block boxes, then one block folds them
constexpr int ANNP_BBOX_BLOCKS = 256;
__global__ __launch_bounds__(256) void annp_bbox_partial(const double *x, int n, double *part)   // part[gridDim.x][6]
{
    __shared__ double slo[3][4], shi[3][4];
    ANNP_POISON();
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
        for (int d = 0; d < 3; d++) { const double v = x[3 * (size_t)k + d]; lo[d] = fmin(lo[d], v); hi[d] = fmax(hi[d], v); }
    for (int d = 0; d < 3; d++) {
        for (int off = 32; off > 0; off >>= 1) { lo[d] = fmin(lo[d], __shfl_xor(lo[d], off, 64)); hi[d] = fmax(hi[d], __shfl_xor(hi[d], off, 64)); }
        if ((threadIdx.x & 63) == 0) { slo[d][threadIdx.x >> 6] = lo[d]; shi[d][threadIdx.x >> 6] = hi[d]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int d = threadIdx.x;
        part[6 * blockIdx.x + d] = fmin(fmin(slo[d][0], slo[d][1]), fmin(slo[d][2], slo[d][3]));
        part[6 * blockIdx.x + 3 + d] = fmax(fmax(shi[d][0], shi[d][1]), fmax(shi[d][2], shi[d][3]));
    }
}
__global__ __launch_bounds__(64) void annp_bbox_final(const double *part, int nparts, double *bbox)
{
    const int lane = threadIdx.x;
    for (int d = 0; d < 3; d++) {
        double lo = 1e300, hi = -1e300;
        for (int k = lane; k < nparts; k += 64) { lo = fmin(lo, part[6 * k + d]); hi = fmax(hi, part[6 * k + 3 + d]); }
        for (int off = 32; off > 0; off >>= 1) { lo = fmin(lo, __shfl_xor(lo, off, 64)); hi = fmax(hi, __shfl_xor(hi, off, 64)); }
        if (lane == 0) { bbox[d] = lo; bbox[3 + d] = hi; }
    }
}

// a row count that outgrew the pitch is cut back to it (the entries beyond were not written: annp_neigh_tile<true>)
__global__ void annp_clamp_int(int *v, int n, int hi)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n && v[k] > hi) v[k] = hi;
}

struct BinGeom { double lo[3]; double inv; int nb[3]; };

__device__ __forceinline__ int bin_of(const BinGeom &g, double px, double py, double pz)
{
    int c0 = (int)floor((px - g.lo[0]) * g.inv), c1 = (int)floor((py - g.lo[1]) * g.inv), c2 = (int)floor((pz - g.lo[2]) * g.inv);
    c0 = min(max(c0, 0), g.nb[0] - 1); c1 = min(max(c1, 0), g.nb[1] - 1); c2 = min(max(c2, 0), g.nb[2] - 1);
    return (c2 * g.nb[1] + c1) * g.nb[0] + c0;
}

__global__ void annp_bin_count(const double *x, int n, BinGeom g, int *binof, int *bincount)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int b = bin_of(g, x[3 * (size_t)k], x[3 * (size_t)k + 1], x[3 * (size_t)k + 2]);
    binof[k] = b;
    atomicAdd(&bincount[b], 1);
}

// exclusive scan of up to a few 10^5 ints by one block (bins are few)
__global__ __launch_bounds__(1024) void annp_scan_bins(const int *cnt, int n, int *start)
{
    __shared__ int part[1024];
    ANNP_POISON();
    const int per = (n + 1023) / 1024;
    const int b0 = threadIdx.x * per, b1 = min(n, b0 + per);
    int s = 0;
    for (int k = b0; k < b1; k++) s += cnt[k];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { int run = 0; for (int k = 0; k < 1024; k++) { const int t = part[k]; part[k] = run; run += t; } start[n] = run; }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int k = b0; k < b1; k++) { start[k] = run; run += cnt[k]; }
}

__global__ void annp_bin_fill(int n, const int *binof, const int *binstart, int *binfill, int *binitems)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int b = binof[k];
    binitems[binstart[b] + atomicAdd(&binfill[b], 1)] = k;
}

// make the order inside every bin deterministic (ascending atom index): a wave per bin ranks its items (they are distinct:
// rank = how many are smaller) and writes them, in order, to a second array.  Up to 64 items sit one per lane and are
// compared through v_readlane; longer bins re-read the unsorted items from memory.  (One thread per bin doing an
// insertion sort took 0.31 ms for the 23 k bins of the 1 M-atom box, a seventh of a rebuild.)
__global__ __launch_bounds__(256) void annp_bin_sort(int nbins, const int *binstart, const int *items, int *sorted)
{
    const int lane = lane_id();
    const int b = uniform(blockIdx.x * ANNP_WAVES_PER_BLOCK + (threadIdx.x >> 6));
    if (b >= nbins) return;
    const int s = binstart[b], n = binstart[b + 1] - s;
    if (n <= 64) {
        const int v = lane < n ? items[s + lane] : 0x7fffffff;
        int rank = 0;
        for (int k = 0; k < n; k++) rank += (__builtin_amdgcn_readlane(v, k) < v) ? 1 : 0;
        if (lane < n) sorted[s + rank] = v;
    } else {
        for (int a = lane; a < n; a += 64) {
            const int v = items[s + a];
            int rank = 0;
            for (int k = 0; k < n; k++) rank += (items[s + k] < v) ? 1 : 0;
            sorted[s + rank] = v;
        }
    }
}

// positions in bin order, so that a run of bins is one contiguous stretch of coordinates: gathered through binitems
// each candidate costs a 64-byte line for 24 useful bytes, and the two passes are bound by exactly that traffic
__global__ void annp_bin_gather_x(const double *x, int n, const int *binitems, double *xs)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int j = binitems[k];
    xs[3 * (size_t)k] = x[3 * (size_t)j]; xs[3 * (size_t)k + 1] = x[3 * (size_t)j + 1]; xs[3 * (size_t)k + 2] = x[3 * (size_t)j + 2];
}

// The list pass (FILL = false counts, FILL = true writes; pitch > 0: rows are pitch entries apart, entries beyond the pitch
// are counted, not written), a workgroup per bin: the candidates of the bin's 27-bin neighbourhood (~1 430 for bcc Fe at
// 8.5 A) are staged in LDS once and every owned atom of the bin (~50), a wave each, tests against that copy, four groups of
// 64 candidates per trip.  (Round 2's wave-per-atom form re-read the candidates from L2 for every atom and ran one group per
// trip: 3.1 ms per pass at 1 M atoms, this one 1.3.)  The candidate stream is the sequence z, y, then the contiguous x-run of
// bins, inside a bin by ascending index, taken in chunks of NEIGH_CH: rows come out in a fixed order.
constexpr int NEIGH_CH = 1536;      // candidates per chunk: 42 KB of LDS (three workgroups per CU; a bcc-Fe bin's neighbourhood at 8.5 A is ~1 430)
constexpr int NEIGH_U = 4;          // groups of 64 candidates tested per trip
constexpr int NEIGH_MAXA = 512;     // atoms of a bin handled per sweep over the stream
inline size_t neigh_tile_lds() { return (size_t)NEIGH_CH * (3 * 8 + 4) + (size_t)NEIGH_MAXA * 4 + 32 * 4; }

template <bool FILL>
__global__ __launch_bounds__(256) void annp_neigh_tile(const double *x, const double *xs, int nlocal, BinGeom g, double rc2,
                                                       const int *binstart, const int *binitems,
                                                       int *numneigh, const long long *first, int *neigh, int pitch)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    ANNP_POISON();
    double *cx = reinterpret_cast<double *>(lds_raw), *cy = cx + NEIGH_CH, *cz = cy + NEIGH_CH;
    int *cj = reinterpret_cast<int *>(cz + NEIGH_CH);
    int *cntl = cj + NEIGH_CH;          // [NEIGH_MAXA] running row length of the bin's atoms
    int *runs = cntl + NEIGH_MAXA;      // [0..8] stream offset of each x-run, [9] total, [10..18] first item of each run (32 ints)
    const int b = blockIdx.x;
    const int a_lo = binstart[b], na = binstart[b + 1] - a_lo;
    if (na == 0) return;
    // bins hold owned atoms and ghosts alike, items ascending: owned ones (index < nlocal) come first
    if (binitems[a_lo] >= nlocal) return;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int c0 = b % g.nb[0], c1 = (b / g.nb[0]) % g.nb[1], c2 = b / (g.nb[0] * g.nb[1]);
    if (threadIdx.x == 0) {
        int tot = 0, r = 0;
        for (int z = c2 - 1; z <= c2 + 1; z++)
            for (int y = c1 - 1; y <= c1 + 1; y++, r++) {
                int s = 0, e = 0;
                if (z >= 0 && z < g.nb[2] && y >= 0 && y < g.nb[1]) {
                    const int x0 = max(c0 - 1, 0), x1 = min(c0 + 1, g.nb[0] - 1);
                    s = binstart[(z * g.nb[1] + y) * g.nb[0] + x0];
                    e = binstart[(z * g.nb[1] + y) * g.nb[0] + x1 + 1];
                }
                runs[r] = tot; runs[10 + r] = s;
                tot += e - s;
            }
        runs[9] = tot;
    }
    __syncthreads();
    const int total = runs[9];
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int a0 = 0; a0 < na; a0 += NEIGH_MAXA) {
        const int nb_ = min(NEIGH_MAXA, na - a0);
        for (int t = threadIdx.x; t < nb_; t += 256) cntl[t] = 0;
        for (int q0 = 0; q0 < total; q0 += NEIGH_CH) {
            const int nq = min(NEIGH_CH, total - q0);
            __syncthreads();            // the previous chunk has been read by every wave (and cntl / runs are written)
            for (int t = threadIdx.x; t < nq; t += 256) {
                const int q = q0 + t;
                int r = 0;
#pragma unroll
                for (int k = 1; k < 9; k++) r += (q >= runs[k]) ? 1 : 0;
                const int k = runs[10 + r] + (q - runs[r]);
                cx[t] = xs[3 * (size_t)k]; cy[t] = xs[3 * (size_t)k + 1]; cz[t] = xs[3 * (size_t)k + 2];
                cj[t] = binitems[k];
            }
            __syncthreads();
            for (int ai = wave; ai < nb_; ai += 4) {
                const int i = binitems[a_lo + a0 + ai];
                if (i >= nlocal) break;             // ghosts from here on (ascending order)
                const double xi = x[3 * (size_t)i], yi = x[3 * (size_t)i + 1], zi = x[3 * (size_t)i + 2];
                int cnt = cntl[ai];
                int *out = FILL ? neigh + (pitch > 0 ? (long long)i * pitch : first[i]) : nullptr;
                const int room = (FILL && pitch > 0) ? pitch : 0x7fffffff;
                for (int k0 = 0; k0 < nq; k0 += 64 * NEIGH_U) {     // NEIGH_U groups of 64 per trip: their LDS reads are in flight together
                    int jq[NEIGH_U];
                    bool inq[NEIGH_U];
#pragma unroll
                    for (int u = 0; u < NEIGH_U; u++) {
                        const int k = k0 + 64 * u + lane;
                        const bool v = k < nq;
                        const int q = v ? k : 0;
                        jq[u] = cj[q];
                        const double dx = xi - cx[q], dy = yi - cy[q], dz = zi - cz[q];
                        inq[u] = v && (jq[u] != i) && (dx * dx + dy * dy + dz * dz <= rc2);
                    }
#pragma unroll
                    for (int u = 0; u < NEIGH_U; u++) {
                        const unsigned long long m = __ballot(inq[u]);
                        if (FILL && inq[u]) { const int pos = cnt + __popcll(m & lt); if (pos < room) out[pos] = jq[u]; }
                        cnt += __popcll(m);
                    }
                }
                if (lane == 0) cntl[ai] = cnt;
            }
        }
        __syncthreads();
        if (!FILL || pitch > 0)
            for (int t = threadIdx.x; t < nb_; t += 256) {
                const int i = binitems[a_lo + a0 + t];
                if (i < nlocal) numneigh[i] = cntl[t];
            }
        __syncthreads();
    }
}

__global__ void annp_first_pitched(long long *first, int n, int pitch)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k <= n) first[k] = (long long)k * pitch;
}

// exclusive scan of numneigh -> first (int64), three small kernels
__global__ __launch_bounds__(1024) void annp_scan_block_sums(const int *v, int n, long long *bs)
{
    __shared__ long long sh[16];
    ANNP_POISON();
    const int k = blockIdx.x * 1024 + threadIdx.x;
    long long s = (k < n) ? v[k] : 0;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { long long t = 0; for (int w = 0; w < 16; w++) t += sh[w]; bs[blockIdx.x] = t; }
}
// exclusive scan of the block sums in place, the grand total to *total.  One workgroup of 1024 threads (launch it as <<<1, 1024>>>;
// any smaller workgroup works too): a thread sums a contiguous chunk, the chunk sums are scanned through LDS, the chunk is
// rewritten.  (Rounds 1-3 did this with one thread: 0.1 us per block sum, 120 us for the 1 221 block sums of a 1.25 M-row scan --
// most of a re-planning's device time once that was kernels, and a fifteenth of the list build.)
__global__ __launch_bounds__(1024) void annp_scan_block_offsets(long long *bs, int nblocks, long long *total)
{
    __shared__ long long sh[1024];
    ANNP_POISON();
    const int nt = blockDim.x, t = threadIdx.x;
    const int chunk = (nblocks + nt - 1) / nt;
    const int a = min(nblocks, t * chunk), b = min(nblocks, a + chunk);
    long long sum = 0;
    for (int k = a; k < b; k++) sum += bs[k];
    sh[t] = sum;
    __syncthreads();
    for (int off = 1; off < nt; off <<= 1) {        // Hillis-Steele inclusive scan of the chunk sums
        const long long v = (t >= off) ? sh[t - off] : 0;
        __syncthreads();
        sh[t] += v;
        __syncthreads();
    }
    long long run = sh[t] - sum;
    for (int k = a; k < b; k++) { const long long v = bs[k]; bs[k] = run; run += v; }
    if (t == nt - 1) *total = sh[t];
}
__global__ __launch_bounds__(1024) void annp_scan_finish(const int *v, int n, const long long *bs, long long *first)
{
    __shared__ long long sh[1024];
    ANNP_POISON();
    const int k = blockIdx.x * 1024 + threadIdx.x;
    sh[threadIdx.x] = (k < n) ? v[k] : 0;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {      // Hillis-Steele inclusive scan
        long long t = (threadIdx.x >= off) ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    if (k < n) first[k] = bs[blockIdx.x] + sh[threadIdx.x] - v[k];
}

template <typename T>
inline int nb_alloc(T *&p, size_t &cap, size_t n, size_t &bytes, std::string &msg)
{
    if (n <= cap && p) return 0;
    if (p) { (void)hipFree(p); bytes -= cap * sizeof(T); p = nullptr; }
    const size_t want = n + n / 8 + 64;
    if (hipMalloc((void **)&p, want * sizeof(T)) != hipSuccess) { msg = "neighbour build: out of device memory"; cap = 0; return -3; }
    cap = want;
    bytes += want * sizeof(T);
    return 0;
}

// Look at what a lazy build left in its pinned words (no-op otherwise).  -7: a row had outgrown the pitch.
inline int neigh_settle(NeighBuild &nb, std::string &msg)
{
    if (!nb.pending) return 0;
    if (hipEventSynchronize(nb.ev_lazy) != hipSuccess) { msg = "neighbour build: hipEventSynchronize failed"; return -4; }     // (recorded a rebuild interval ago: no wait in practice)
    nb.pending = false;
    const int mx = (int)(nb.h_max[0] & 0xffffffffll);
    nb.max_numneigh = std::min(mx, nb.pending_pitch);
    nb.pitch = (mx + mx / 16 + 4 + 7) / 8 * 8;
    if (mx > nb.pending_pitch) {
        nb.pitch = 0;           // the next build is an exact one
        nb.have_geom = false;
        msg = "neighbour build: a list row grew from at most " + std::to_string(nb.pending_pitch) + " to " + std::to_string(mx) +
              " entries between two rebuilds; the evaluations since the last rebuild missed the entries beyond the pitch (ANNP_HIP_NEIGH_SYNC=1 checks every build before it is used)";
        return -7;
    }
    return 0;
}

// allow_lazy: the caller is a device-resident stepping loop (annp_hip_neigh_build_device); the host-pointer entries, which wait for their
// results anyway and are handed any configuration at any time, check every build before it is used
inline int neigh_build(NeighBuild &nb, int nlocal, int nall, const double *d_x, double cutneigh, hipStream_t s, std::string &msg, bool allow_lazy = false)
{
#define NB_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { msg = std::string(#call) + ": " + hipGetErrorString(e_); return e_ == hipErrorOutOfMemory ? -3 : -4; } } while (0)
    nb.valid = false;
    if (nall <= 0 || nlocal <= 0) { nb.nlocal = nlocal; nb.nall = nall; nb.max_numneigh = 0; return 0; }
    if (!nb.bbox) {     // 6 doubles of result + the per-block boxes behind them
        const size_t bb = (6 + 6 * (size_t)ANNP_BBOX_BLOCKS) * sizeof(double);
        NB_TRY(hipMalloc((void **)&nb.bbox, bb)); NB_TRY(hipMalloc((void **)&nb.dmax, 4 * sizeof(long long))); nb.bytes += bb + 4 * sizeof(long long);
    }
    if (nb_alloc(nb.binof, nb.cap_binof, (size_t)nall, nb.bytes, msg)) return -3;
    if (nb_alloc(nb.binitems, nb.cap_binitems, (size_t)nall, nb.bytes, msg)) return -3;
    const int bparts = std::max(1, std::min(ANNP_BBOX_BLOCKS, (nall + 255) / 256));
    if (!nb.h_bbox) {
        NB_TRY(hipHostMalloc((void **)&nb.h_bbox, 6 * sizeof(double)));
        NB_TRY(hipHostMalloc((void **)&nb.h_max, sizeof(long long)));
        NB_TRY(hipEventCreateWithFlags(&nb.ev_lazy, hipEventDisableTiming));
    }
    // what the last lazy build left behind: its row maximum and its bounding box
    bool reuse_geom = false;
    if (int rc_ = neigh_settle(nb, msg)) return rc_;
    if (allow_lazy && nb.lazy && nb.have_geom && nb.geom_cut == cutneigh && nb.pitch > 0) {
        reuse_geom = true;
        for (int d = 0; d < 3; d++) {       // the box as of the last build: still within half a bin of the bins?
            const double lo = nb.geom_lo[d], hi = lo + nb.geom_nb[d] * cutneigh;
            if (nb.h_bbox[d] < lo - 0.5 * cutneigh || nb.h_bbox[3 + d] > hi + 0.5 * cutneigh || nb.h_bbox[3 + d] < hi - 1.5 * cutneigh) reuse_geom = false;
        }
    }
    hipLaunchKernelGGL(annp_bbox_partial, dim3(bparts), dim3(256), 0, s, d_x, nall, nb.bbox + 6);
    hipLaunchKernelGGL(annp_bbox_final, dim3(1), dim3(64), 0, s, nb.bbox + 6, bparts, nb.bbox);
    NB_TRY(hipMemcpyAsync(nb.h_bbox, nb.bbox, 6 * sizeof(double), hipMemcpyDeviceToHost, s));
    BinGeom g;
    long long nbins = 1;
    if (reuse_geom) {
        for (int d = 0; d < 3; d++) { g.lo[d] = nb.geom_lo[d]; g.nb[d] = nb.geom_nb[d]; nbins *= g.nb[d]; }
    } else {
        NB_TRY(hipStreamSynchronize(s));
        for (int d = 0; d < 3; d++) {
            g.lo[d] = nb.h_bbox[d];
            g.nb[d] = (int)std::floor((nb.h_bbox[3 + d] - nb.h_bbox[d]) / cutneigh) + 1;
            if (g.nb[d] < 1) g.nb[d] = 1;
            nbins *= g.nb[d];
            nb.geom_lo[d] = g.lo[d]; nb.geom_nb[d] = g.nb[d];
        }
        nb.geom_cut = cutneigh; nb.have_geom = true;
    }
    g.inv = 1.0 / cutneigh;
    if (nbins > (1ll << 27)) { msg = "neighbour build: too many bins"; return -1; }
    if (nb_alloc(nb.bincount, nb.cap_bincount, (size_t)nbins + 1, nb.bytes, msg)) return -3;
    if (nb_alloc(nb.binstart, nb.cap_binstart, (size_t)nbins + 1, nb.bytes, msg)) return -3;
    if (nb_alloc(nb.binfill, nb.cap_binfill, (size_t)nbins + 1, nb.bytes, msg)) return -3;
    NB_TRY(hipMemsetAsync(nb.bincount, 0, sizeof(int) * (nbins + 1), s));
    NB_TRY(hipMemsetAsync(nb.binfill, 0, sizeof(int) * (nbins + 1), s));
    const int tb = 256, gb = (nall + tb - 1) / tb;
    hipLaunchKernelGGL(annp_bin_count, dim3(gb), dim3(tb), 0, s, d_x, nall, g, nb.binof, nb.bincount);
    hipLaunchKernelGGL(annp_scan_bins, dim3(1), dim3(1024), 0, s, nb.bincount, (int)nbins, nb.binstart);
    hipLaunchKernelGGL(annp_bin_fill, dim3(gb), dim3(tb), 0, s, nall, nb.binof, nb.binstart, nb.binfill, nb.binitems);
    // sorted items go to the array that held the bin of every atom (not needed any more); the two arrays trade places
    hipLaunchKernelGGL(annp_bin_sort, dim3((unsigned)((nbins + ANNP_WAVES_PER_BLOCK - 1) / ANNP_WAVES_PER_BLOCK)), dim3(256), 0, s, (int)nbins, nb.binstart,
                       (const int *)nb.binitems, nb.binof);
    std::swap(nb.binof, nb.binitems);
    std::swap(nb.cap_binof, nb.cap_binitems);
    if (nb_alloc(nb.xs, nb.cap_xs, (size_t)nall * 3, nb.bytes, msg)) return -3;
    hipLaunchKernelGGL(annp_bin_gather_x, dim3(gb), dim3(tb), 0, s, d_x, nall, nb.binitems, nb.xs);
    // count
    if (nb_alloc(nb.numneigh, nb.cap_numneigh, (size_t)nall, nb.bytes, msg)) return -3;
    if (nb_alloc(nb.first, nb.cap_first, (size_t)nall + 1, nb.bytes, msg)) return -3;
    if (nb_alloc(nb.blocksum, nb.cap_blocksum, (size_t)(nall / 1024 + 2), nb.bytes, msg)) return -3;
    NB_TRY(hipMemsetAsync(nb.numneigh, 0, sizeof(int) * (size_t)nall, s));
    const int wb = (nlocal + ANNP_WAVES_PER_BLOCK - 1) / ANNP_WAVES_PER_BLOCK;
    (void)wb;
    const size_t tlds = neigh_tile_lds();
    if (!nb.tile_attr) {
        NB_TRY(hipFuncSetAttribute((const void *)annp_neigh_tile<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tlds));
        NB_TRY(hipFuncSetAttribute((const void *)annp_neigh_tile<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tlds));
        nb.tile_attr = true;
    }
    const double rc2 = cutneigh * cutneigh;
    auto learn_pitch = [&]() { nb.pitch = (nb.max_numneigh + nb.max_numneigh / 16 + 4 + 7) / 8 * 8; };
    // One pass when the previous build left a row pitch: the distance tests are the whole cost and the exact layout
    // needs them twice (count, then fill).  Rows that outgrow the pitch only show in the count; then fall through.
    // Not for very uneven systems (surfaces, voids: the maximum far above the mean makes the pitched array much larger than
    // the exact one), and if its allocation fails the exact layout below is tried before giving up.
    bool try_pitched = nb.pitch > 0 && (long long)nlocal * nb.pitch < (1ll << 33) &&
                       (nb.mean_exact <= 0.0 || nb.pitch <= 1.5 * nb.mean_exact + 8.0);
    if (try_pitched && nb_alloc(nb.neigh, nb.cap_neigh, (size_t)nlocal * nb.pitch, nb.bytes, msg)) {
        try_pitched = false;
        nb.pitch = 0;
        msg.clear();
        (void)hipGetLastError();
    }
    if (try_pitched) {
        NB_TRY(hipMemsetAsync(nb.dmax, 0, 4 * sizeof(long long), s));
        hipLaunchKernelGGL(annp_first_pitched, dim3((nall + 1 + 255) / 256), dim3(256), 0, s, nb.first, nall, nb.pitch);
        hipLaunchKernelGGL((annp_neigh_tile<true>), dim3((unsigned)nbins), dim3(256), tlds, s, d_x, nb.xs, nlocal, g, rc2, nb.binstart, nb.binitems,
                           nb.numneigh, (const long long *)nb.first, nb.neigh, nb.pitch);
        hipLaunchKernelGGL(annp_max_int, dim3(annp_max_int_blocks(nlocal)), dim3(256), 0, s, nb.numneigh, nlocal, nb.dmax);
        NB_TRY(hipMemcpyAsync(nb.h_max, nb.dmax, sizeof(long long), hipMemcpyDeviceToHost, s));
        if (reuse_geom) {       // steady state: nobody waits; the row maximum is looked at by the next build
            hipLaunchKernelGGL(annp_clamp_int, dim3((nlocal + 255) / 256), dim3(256), 0, s, nb.numneigh, nlocal, nb.pitch);
            NB_TRY(hipEventRecord(nb.ev_lazy, s));
            NB_TRY(hipGetLastError());
            nb.pending = true; nb.pending_pitch = nb.pitch;
            nb.max_numneigh = nb.pitch;         // an upper bound, as far as the caller is concerned
            nb.nlocal = nlocal; nb.nall = nall; nb.valid = true; nb.pitched = true; nb.pitch_used = nb.pitch;
            return 0;
        }
        NB_TRY(hipStreamSynchronize(s));
        nb.max_numneigh = (int)(nb.h_max[0] & 0xffffffffll);
        const bool fits = nb.max_numneigh <= nb.pitch;
        const int used = nb.pitch;
        learn_pitch();
        if (fits) { nb.nlocal = nlocal; nb.nall = nall; nb.valid = true; nb.pitched = true; nb.pitch_used = used; return 0; }
        NB_TRY(hipMemsetAsync(nb.numneigh, 0, sizeof(int) * (size_t)nall, s));
    }
    hipLaunchKernelGGL((annp_neigh_tile<false>), dim3((unsigned)nbins), dim3(256), tlds, s, d_x, nb.xs, nlocal, g, rc2, nb.binstart, nb.binitems,
                       nb.numneigh, (const long long *)nullptr, (int *)nullptr, 0);
    const int nblk = (nlocal + 1023) / 1024;
    long long *dtot = reinterpret_cast<long long *>(nb.dmax) + 1;
    NB_TRY(hipMemsetAsync(nb.dmax, 0, 4 * sizeof(long long), s));
    hipLaunchKernelGGL(annp_scan_block_sums, dim3(nblk), dim3(1024), 0, s, nb.numneigh, nlocal, nb.blocksum);
    hipLaunchKernelGGL(annp_scan_block_offsets, dim3(1), dim3(1024), 0, s, nb.blocksum, nblk, dtot);
    hipLaunchKernelGGL(annp_scan_finish, dim3(nblk), dim3(1024), 0, s, nb.numneigh, nlocal, nb.blocksum, nb.first);
    hipLaunchKernelGGL(annp_max_int, dim3(annp_max_int_blocks(nlocal)), dim3(256), 0, s, nb.numneigh, nlocal, nb.dmax);
    long long hres[2];
    NB_TRY(hipMemcpyAsync(hres, nb.dmax, sizeof(hres), hipMemcpyDeviceToHost, s));
    NB_TRY(hipStreamSynchronize(s));
    nb.max_numneigh = (int)(hres[0] & 0xffffffffll);
    const long long total = hres[1];
    // Room for the pitched layout of the NEXT build too, where that build will take it (the same test as above): otherwise the first
    // rebuild frees and allocates the whole list again -- a gigabyte at 1 M atoms, tens of milliseconds when the driver is slow about
    // it (the occasional 37-70 ms rebuild of rounds 3-5).
    learn_pitch();
    size_t want_entries = (size_t)std::max<long long>(total, 1);
    {
        const double mean = nlocal > 0 ? (double)total / (double)nlocal : 0.0;
        if ((long long)nlocal * nb.pitch < (1ll << 33) && nb.pitch <= 1.5 * mean + 8.0) want_entries = std::max(want_entries, (size_t)nlocal * nb.pitch);
    }
    if (nb_alloc(nb.neigh, nb.cap_neigh, want_entries, nb.bytes, msg)) return -3;
    hipLaunchKernelGGL((annp_neigh_tile<true>), dim3((unsigned)nbins), dim3(256), tlds, s, d_x, nb.xs, nlocal, g, rc2, nb.binstart, nb.binitems,
                       nb.numneigh, (const long long *)nb.first, nb.neigh, 0);
    NB_TRY(hipGetLastError());
    nb.nlocal = nlocal; nb.nall = nall; nb.valid = true; nb.pitched = false; nb.pitch_used = 0;
    nb.mean_exact = nlocal > 0 ? (double)total / (double)nlocal : 0.0;      // 0 = unknown
    return 0;
#undef NB_TRY
}

}  // namespace annp
