// fe_shp_kernels.hpp -- annp_fe_force_shp: the Chebyshev force pass of fe_shf_kernels.hpp (fe_v2/src/pair_annp.cpp:190-213, 658-695)
// as a workgroup that WALKS a run of units instead of living for one (round 5).
//
// Same arithmetic, same tables, same force table as annp_fe_force_sh (that file's header derives them); what changes is who does
// what, and when.  A unit is eight consecutive list entries = two groups of four atoms; four waves work for a group, wave q
// on neighbour slots 2q, 2q+1 of every lane.  bcc Fe has 112 = 16 x 7 neighbours: wave 3 has ONE slot, half the columns of the
// others.  In annp_fe_force_sh every wave also built a quarter of its group's table, so the waves came out at ~2 460 / 2 460 /
// 2 460 / 1 520 vector instructions -- and since the eight waves of a workgroup are dealt over the CU's four SIMDs in order,
// both light waves of a workgroup sit on the same SIMD: three SIMDs of a CU carry 4 x 2 460 instructions per pair of resident
// workgroups (94 % of the 42 k cycles a workgroup lived), the fourth 4 x 1 520.  The pass was bound by its heavy SIMDs.
// Here the light wave of a group builds the WHOLE table -- for the NEXT unit, into the other of two table buffers, while the
// three heavy waves (and itself, for its one slot) evaluate the current unit:
//
//     heavy wave, step s                                  light wave, step s
//     request header + neighbour indices of unit s+1      the same, and: the four moment rows of unit s+1 -> LDS (12 LDS-DMA
//     geometry, radial part (2 neighbours per lane)            pieces of 1 KB, no registers), W_l and c_m of its lanes' atoms
//     columns of unit s from table[s & 1]                  geometry, columns (1 neighbour per lane)
//     request the positions of unit s+1's neighbours      the same
//     force assembly, inserts into the force table        the same; then B = W kappa A in place and the change of basis
//     -------------------------------------------- barrier ------------------------------------------------------
//     flush + clear of the force table (all 512 threads); a counter in LDS says when every wave is through, and a wave
//     looks at it only in front of its next inserts, a whole evaluation later: no second barrier
//
// so that between two barriers every wave has the same work (~2 100 instructions), nothing a unit needs from memory is waited
// for where it is needed (the two dependent round trips of a unit -- header/indices, then positions -- and the 12 KB of moments
// are requested a step ahead), and a workgroup's start-up and wind-down are paid once per run.
// LDS: force table 26 KB + 2 buffers x 2 groups x 12 KB of tables + 2 KB of c_m = 76 KB: two workgroups per CU, four waves per
// SIMD, as before.
#pragma once
#include "fe_shf_kernels.hpp"

namespace annp {

constexpr int SHP_TBUF = 12 * 1024;                // a group's four tables (12 160 B) in whole 1 KB pieces of LDS-DMA
constexpr int SHP_PIECES = SHP_TBUF / 1024;
// what the waves need of an atom's coefficient row (c_m | p_k | W_l | P(1), mlp_kernels.hpp), as 15 chunks of 16 bytes:
// doubles [0, 10) = c_0..c_8 (and p_0, unused) | doubles [28, 48) = W_0..W_18, P(1)
constexpr int SHP_CCH = 15;                        // chunks per atom: 4 x 15 = 60 lanes of one LDS-DMA piece
constexpr int SHP_CROW = 2 * SHP_CCH;              // doubles per atom in the coefficient area
constexpr int SHP_CW = 10, SHP_CP1 = 29;           // ... where W_0 and P(1) sit in it
constexpr int SHP_CBUF = 1024;                     // bytes per group and buffer (one piece)
constexpr int SHP_KS = SHF_NBUCK + 32;             // ints per key buffer: keys, [SHF_NBUCK] the contributions without a bucket
constexpr int SHP_ZPAD = 256;                      // zeros in front of the first table (shp_convert_block reads up to 240 bytes below a table)
static_assert(SHF_GA * SHF_TBYTES <= SHP_TBUF && SHF_GROUPS == 2 && SHF_WAVES == 8 && SHF_GA * SHP_CCH <= 64, "layout of a workgroup");
__host__ __device__ constexpr size_t shp_lds_table() { return (size_t)SHF_NBUCK * SHF_BATOMS * 24 + (2 * SHP_KS + 32) * 4 + SHP_ZPAD; }
constexpr int SHP_DUMP = 1024;                     // per light wave: where the lanes that have nothing to write write
__host__ __device__ constexpr size_t shp_lds_per_block() { return shp_lds_table() + 2 * SHF_GROUPS * (size_t)(SHP_TBUF + SHP_CBUF) + SHF_GROUPS * SHP_DUMP; }
static_assert(shp_lds_table() % 16 == 0 && 2 * shp_lds_per_block() <= 160 * 1024, "two workgroups per CU");

// The lane number, from nothing: every phase of a step derives what it needs of it (atom, lane of the atom, LDS addresses) afresh
// instead of carrying it in registers through the columns, where a wave has none to spare (left to itself the compiler keeps
// a dozen such values alive for the whole walk and spills them).
__device__ __forceinline__ int shp_lane()
{
    int ln;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    return ln;
}

// ---- flush + clear in one sweep, by the workgroup's two light waves (128 threads): thread t takes the doubles t, t + 128, ..
// of the table -- a wave-instruction reads 64 consecutive doubles = 2.7 buckets = 8-9 lines of f -- and resets its key of the
// buffer the NEXT step inserts with (nobody reads that one now: its last readers were the flush a step ago).
__device__ __forceinline__ void shp_flush_clear(const int *key, int *okey, double *acc, double *f, const int t, int *spills)
{
    constexpr int NT = 64 * SHF_GROUPS, PER = SHF_BATOMS * 3;
    static_assert(NT == SHF_NBUCK && (SHF_NBUCK * PER) % NT == 0, "a key per thread, whole sweeps");
#pragma unroll
    for (int k = 0; k < SHF_NBUCK * PER / NT; k++) {
        const int d = t + NT * k;
        const double v = acc[d];
        if (v != 0.0) {
            const int h = d / PER;
            const int b = key[h];
            acc[d] = 0.0;
            if (b >= 0) atomicAdd(&f[(size_t)b * PER + (d - h * PER)], v);
        }
    }
    okey[t] = -1;
    if (t == 0) {
        okey[SHF_NBUCK] = 0;
        const int sp = key[SHF_NBUCK];
        if (spills && sp) atomicAdd(spills, sp);
    }
}
// The waves of a workgroup meet through three counters in LDS instead of barriers, so that nobody stands still for the
// slowest of eight: `adds` (a wave's inserts of a step are in the table: the flush of that step waits for all eight; so does
// the request of the tables after next, whose buffer those waves were reading), `flushed` (a light wave is through with its
// half of a step's flush: the next step's inserts wait for both) and, per group, `ready` (the table of a step is built: the
// group's columns wait for it).  A count is raised after the wave's LDS operations have completed; whoever reads it issues its
// own behind the read.
__device__ __forceinline__ void shp_raise(const unsigned addr)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (shp_lane() == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(1u) : "memory");
}
__device__ __forceinline__ void shp_wait_for(const unsigned addr, const unsigned target)
{
    for (;;) {
        unsigned v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        if ((unsigned)__builtin_amdgcn_readfirstlane((int)v) >= target) break;
        __builtin_amdgcn_s_sleep(2);
    }
}

struct ShpHead {            // what a lane knows of its atom of a unit
    int i, n;               // atom index, in-cutoff neighbours
    int jn[SHF_CC];         // its neighbours in this wave's slots
};
struct ShpPos { double xi, yi, zi, xj[SHF_CC], yj[SHF_CC], zj[SHF_CC]; };

// everything whose address follows from the list entry alone
__device__ __forceinline__ void shp_head_load(const FeArgs &p, const int ii0, const int wq, ShpHead &h)
{
    const int ln = shp_lane();
    const size_t iic = (size_t)min(ii0 + (ln >> 4), p.inum - 1);
    h.n = p.ncount[iic];
    h.i = p.ilist ? p.ilist[iic] : (int)iic;
#pragma unroll
    for (int u = 0; u < SHF_CC; u++) h.jn[u] = p.nbrs[iic * SH_CAP_MAX + (ln & 15) + SHF_GL * (SHF_CC * wq + u)];
}
// ... and what follows from that: is there such an atom, does it have moments, how many of this wave's slots are in use (the
// return value, uniform); the positions are requested
__device__ __forceinline__ int shp_head_resolve(const FeArgs &p, ShpHead &h, const int ii0, const int wq, ShpPos &q)
{
    const int ln = shp_lane();
    const int ii = ii0 + (ln >> 4), l = ln & 15;
    if (ii >= p.inum) h.n = 0;
    if (p.type && !type_mapped(p.active, p.type[h.i])) h.n = 0;
    if (h.n > p.n_cap) {             // no moments for this atom: the pair loop takes it (annp_fe_force_fixup)
        if (l == 0 && wq == 0) {
            const int k = p.ovf_list ? atomicAdd(p.ovf_count, 1) : p.ovf_cap;
            if (k < p.ovf_cap) p.ovf_list[k] = ii;
            else atomicMax(p.errflag, h.n);
        }
        h.n = 0;
    }
    const int nmax = max(max(__builtin_amdgcn_readlane(h.n, 0), __builtin_amdgcn_readlane(h.n, 16)),
                         max(__builtin_amdgcn_readlane(h.n, 32), __builtin_amdgcn_readlane(h.n, 48)));
    const int C = (nmax + SHF_GL - 1) / SHF_GL;                        // neighbours per lane (uniform), 0 .. 8
    q.xi = p.x[3 * (size_t)h.i]; q.yi = p.x[3 * (size_t)h.i + 1]; q.zi = p.x[3 * (size_t)h.i + 2];
#pragma unroll
    for (int u = 0; u < SHF_CC; u++) {
        if (!(l + SHF_GL * (SHF_CC * wq + u) < h.n)) h.jn[u] = h.i;   // a slot without a neighbour reads the centre's own position
        q.xj[u] = p.x[3 * (size_t)h.jn[u]]; q.yj[u] = p.x[3 * (size_t)h.jn[u] + 1]; q.zj[u] = p.x[3 * (size_t)h.jn[u] + 2];
    }
    return min(SHF_CC, max(0, C - SHF_CC * wq));
}

// ---- the light wave's second job: the table of a unit
typedef const double __attribute__((address_space(1))) *shp_gcp;
typedef const char __attribute__((address_space(1))) *shp_gbp;
typedef __attribute__((address_space(3))) shf_v2d *shp_tab_wptr;
// lanes l < n of every atom, as an execution mask (no vector instruction: s_and_saveexec with a literal)
__device__ __forceinline__ bool shp_lanes_below(const int n)
{
    const unsigned long long m16 = n >= 16 ? 0xffffull : ((1ull << n) - 1ull);
    return __builtin_amdgcn_inverse_ballot_w64(m16 * 0x0001000100010001ull);
}
// The group's four moment rows into its table buffer as they are (the descriptor pass wrote them in the table's order): piece k
// is the 64 entries 64 k .. 64 k + 63 of the 760, one per lane; the last piece runs 8 entries into the row behind the group's
// (the moment buffer has rows behind the last list entry's).  And of the four coefficient rows the 15 chunks a step wants,
// one piece.  LDS-DMA: no register is waited for, no register holds them.
__device__ __forceinline__ void shp_table_request(const FeArgs &p, const int ii0, unsigned char *tbuf, unsigned char *cbuf)
{
    const int ln = shp_lane();
    const shp_gbp rows = (shp_gbp)(p.A + (size_t)ii0 * SH_MPAD);
#pragma unroll
    for (int k = 0; k < SHP_PIECES; k++) {
        const int e = 64 * k + ln;
        const int ge = (e >= SHF_NE) + (e >= 2 * SHF_NE) + (e >= 3 * SHF_NE) + (e >= 4 * SHF_NE);
        const unsigned off = 16u * (unsigned)e + (unsigned)(SH_MPAD * 8 - SHF_NE * 16) * (unsigned)ge;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(rows + (unsigned long long)off),
                                         (__attribute__((address_space(3))) void *)(tbuf + 1024 * k), 16, 0, 0);
    }
    {
        const int e = min(ln, SHF_GA * SHP_CCH - 1);
        const int ga = (e >= SHP_CCH) + (e >= 2 * SHP_CCH) + (e >= 3 * SHP_CCH), c = e - SHP_CCH * ga;
        const unsigned off = (unsigned)(ANNP_CPAD * 8) * (unsigned)ga + 16u * (unsigned)c + (c >= 5 ? (unsigned)((FE_NP + FE_NT) * 8 - 80) : 0u);
        const shp_gbp cf = (shp_gbp)(p.coef + (size_t)ii0 * ANNP_CPAD);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(cf + (unsigned long long)off),
                                         (__attribute__((address_space(3))) void *)cbuf, 16, 0, 0);
    }
}
// B = W kappa A in place: position p of any column belongs to l = 18 - p, so a lane needs two W_l for all it scales.  Every lane
// reads (a lane beyond the column's end reads the next column's entries); the lanes of the column write, the others write their
// slot of the wave's dump.  (No branch around a write: the compiler sinks the reads into it and waits for each where it is used.
// And the order is pinned: left to itself under this kernel's register limit the compiler issues one read, waits, issues the
// next -- a trip through LDS, or through the cache for the change of basis' coefficients, per term.)
// The 22 pieces (column, half) of the table, in memory order from the back: columns 0, 1, 2 have more than 16 entries.
struct ShpPiece { int m, blk; };
constexpr int SHP_NPIECE = SH_LMAX + 1 + 3;
constexpr ShpPiece SHP_PIECE[SHP_NPIECE] = {{0, 0}, {0, 16}, {1, 0}, {1, 16}, {2, 0}, {2, 16}, {3, 0}, {4, 0}, {5, 0}, {6, 0}, {7, 0}, {8, 0}, {9, 0},
                                            {10, 0}, {11, 0}, {12, 0}, {13, 0}, {14, 0}, {15, 0}, {16, 0}, {17, 0}, {18, 0}};
template <int P0, int N>
__device__ __forceinline__ void shp_scale_pieces(const unsigned tp, const unsigned dump, const double w0, const double w1)
{
    if constexpr (P0 < SHP_NPIECE) {
        constexpr int NN = P0 + N <= SHP_NPIECE ? N : SHP_NPIECE - P0;
        shf_v2d v[NN];
#pragma unroll
        for (int k = 0; k < NN; k++) v[k] = *(shf_tab_ptr)(uintptr_t)(tp + 16u * (unsigned)(shf_toff(SHP_PIECE[P0 + k].m) + SHP_PIECE[P0 + k].blk));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < NN; k++) {
            const double w = SHP_PIECE[P0 + k].blk ? w1 : w0;
            const unsigned a = tp + 16u * (unsigned)(shf_toff(SHP_PIECE[P0 + k].m) + SHP_PIECE[P0 + k].blk);
            v[k].x *= w; v[k].y *= w;
            *(shp_tab_wptr)(uintptr_t)(shp_lanes_below(SH_LMAX + 1 - SHP_PIECE[P0 + k].m - SHP_PIECE[P0 + k].blk) ? a : dump) = v[k];
        }
        __builtin_amdgcn_sched_barrier(0);
        shp_scale_pieces<P0 + N, N>(tp, dump, w0, w1);
    }
}
// The change of basis of one piece in place, as shf_convert_block does it, with every address an immediate: power k of column M
// sits at tb + 16 (toff(M) + K-1-k); lane l reads the powers BLK + l + 2t at tl + 16 (toff(M) + K-1-BLK) - 32 t with tl = tb - 16 l
// and writes power BLK + l.  A lane without the power, or past the column's top, reads up to 240 bytes below its column
// -- the column before, the table before, the pad in front of the first table: finite numbers all -- with a zero coefficient.
// The coefficients of a piece are requested while the piece before is worked on.
constexpr int SHP_MAXT = (SH_LMAX + 2) / 2;          // terms of the longest piece
template <int P>
struct ShpConv {
    static constexpr int M = SHP_PIECE[P].m, BLK = SHP_PIECE[P].blk;
    static constexpr int K = SH_LMAX + 1 - M;
    static constexpr int first = SHF_CONV_FIRST[M][BLK / 16];
    static constexpr int nt = (K - BLK + 1) / 2;
    static constexpr unsigned a0 = 16u * (unsigned)(shf_toff(M) + K - 1 - BLK);
    static_assert(a0 + 32u >= 32u * nt && nt <= SHP_MAXT, "no read below the column before");
    static __device__ __forceinline__ void request(const shp_gbp convl, double (&mv)[SHP_MAXT])
    {
#pragma unroll
        for (int t = 0; t < nt; t++) mv[t] = *(shp_gcp)(convl + (first + t) * 128);        // 0 where l + 2t > K-1
    }
};
template <int P>
__device__ __forceinline__ void shp_convert_pieces(const unsigned tl, const unsigned dump, const shp_gbp convl, const double (&mv)[SHP_MAXT])
{
    if constexpr (P < SHP_NPIECE) {
        typedef ShpConv<P> C;
        double mvn[SHP_MAXT];
        if constexpr (P + 1 < SHP_NPIECE) ShpConv<P + 1>::request(convl, mvn);
        shf_v2d b[C::nt];
#pragma unroll
        for (int t = 0; t < C::nt; t++) b[t] = *(shf_tab_ptr)(uintptr_t)(tl + C::a0 - 32u * (unsigned)t);
        __builtin_amdgcn_sched_barrier(0);
        double ax = 0.0, ay = 0.0;
#pragma unroll
        for (int t = 0; t < C::nt; t++) { ax = fma(mv[t], b[t].x, ax); ay = fma(mv[t], b[t].y, ay); }
        shf_v2d r; r.x = ax; r.y = ay;
        *(shp_tab_wptr)(uintptr_t)(shp_lanes_below(C::K - C::BLK) ? tl + C::a0 : dump) = r;
        __builtin_amdgcn_sched_barrier(0);
        shp_convert_pieces<P + 1>(tl, dump, convl, mvn);
    }
}
// the rows have landed: B = W kappa A, the change of basis (fe_shf_kernels.hpp)
// FRESH: the pieces were requested just now (the run's first unit); otherwise a step ago, and the only loads behind them are
// the next unit's positions -- at least three instructions (centre, two slots), and loads return in order: three or fewer
// operations outstanding means the pieces have landed.
template <bool FRESH>
__device__ __forceinline__ void shp_table_finish(unsigned char *tbuf, const unsigned char *cbuf, unsigned char *dumpbuf)
{
    if (FRESH) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    const int ln = shp_lane();
    const int g = ln >> 4, l = ln & 15;
    const double *crow = reinterpret_cast<const double *>(cbuf) + g * SHP_CROW;
    const double w0 = crow[SHP_CW + (SH_LMAX - l)];                      // W_l of table position p = lane: l = 18 - p
    const double w1 = crow[SHP_CW + max(2 - l, 0)];                      // ... of p = 16 + lane (lanes 0..2)
    const unsigned tb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)(tbuf + (size_t)g * SHF_TBYTES);
    const unsigned dump = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)dumpbuf + 16u * (unsigned)ln;
    unsigned long long cb = (unsigned long long)(const void *)annp_shf_conv;
    asm volatile("" : "+s"(cb));
    const shp_gbp convl = (shp_gbp)cb + (unsigned long long)(8u * (unsigned)l);
    double mv[SHP_MAXT];
    ShpConv<0>::request(convl, mv);
    shp_scale_pieces<0, 8>(tb + 16u * (unsigned)l, dump, w0, w1);
    shp_convert_pieces<0>(tb - 16u * (unsigned)l, dump, convl, mv);
}

template <int NP, int NT, bool VIRIAL, bool LIGHT>
__device__ __forceinline__ void shp_walk(const FeArgs &p, unsigned char *lds_raw, const int U0, const int nsteps, const int wave, const int wq)
{
    const int grp = wave / SHF_GW;
    double *acc = reinterpret_cast<double *>(lds_raw);
    int *keys = reinterpret_cast<int *>(lds_raw + (size_t)SHF_NBUCK * SHF_BATOMS * 24);
    const unsigned c_adds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)(keys + 2 * SHP_KS);
    const unsigned c_flushed = c_adds + 4u, c_ready = c_adds + 8u + 4u * (unsigned)grp;
    unsigned char *tables = lds_raw + shp_lds_table();
    unsigned char *cbufs = tables + 2 * SHF_GROUPS * (size_t)SHP_TBUF;
    unsigned char *dumpbuf = cbufs + 2 * SHF_GROUPS * (size_t)SHP_CBUF + (size_t)grp * SHP_DUMP;

#ifndef SHP_LIGHT_PRIO
#define SHP_LIGHT_PRIO 2
#endif
    // the light wave's step is a chain of waits (one neighbour per lane: half the independent work per table entry; then the table,
    // reads and a few multiply-adds per piece) on which the seven other waves wait at the barrier: it goes first when it can go
    if (LIGHT && SHP_LIGHT_PRIO) __builtin_amdgcn_s_setprio(SHP_LIGHT_PRIO);
    // ---- the run's first unit: nothing was requested ahead
    ShpHead cur;
    ShpPos pos;
    int ii0 = (U0 * SHF_GROUPS + grp) * SHF_GA;              // the group's first list entry
    shp_head_load(p, ii0, wq, cur);
    if (LIGHT) {
        shp_table_request(p, ii0, tables + (size_t)grp * SHP_TBUF, cbufs + (size_t)grp * SHP_CBUF);
        shp_table_finish<true>(tables + (size_t)grp * SHP_TBUF, cbufs + (size_t)grp * SHP_CBUF, dumpbuf);
    }
    int cc = shp_head_resolve(p, cur, ii0, wq, pos);
    {       // the force table: clear, key buffer 0 free, nothing counted, the pad behind the keys
        const int tid = wave * 64 + shp_lane();
        constexpr int NT2 = SHF_NBUCK * SHF_BATOMS * 3 / 2, TH = 64 * SHF_WAVES;
        double2 *a2 = reinterpret_cast<double2 *>(acc);
#pragma unroll
        for (int k = 0; k < (NT2 + TH - 1) / TH; k++)
            if (k * TH + TH <= NT2 || tid + k * TH < NT2) a2[tid + k * TH] = make_double2(0.0, 0.0);
        if (tid < 2 * SHP_KS + 32 + SHP_ZPAD / 4) keys[tid] = tid < SHF_NBUCK ? -1 : 0;
    }
    __syncthreads();

#pragma unroll 1
    for (int s = 0; s < nsteps; s++) {
        const bool more = s + 1 < nsteps;
        const int b = s & 1;
#ifdef ANNP_SHF_STAMPS          // developer timing builds (tools/shp_stamps.py): the wave's row of the (dead) descriptor buffer
        unsigned long long *stamp_row = nullptr;
        {
            const int row = ii0 + wq;
            if (row < p.inum) stamp_row = reinterpret_cast<unsigned long long *>(p.G + (size_t)row * ANNP_GPAD);
            if (stamp_row && lane_id() == 0) { stamp_row[15] = __builtin_amdgcn_s_getreg(0xF804); stamp_row[14] = blockIdx.x * 64 + s; stamp_row[13] = __builtin_amdgcn_s_getreg(0xF814); }
        }
#endif
        SHF_STAMP(0);
        const int ii0n = ii0 + SHF_GROUPS * SHF_GA;
        ShpHead nx = cur;
        if (more) shp_head_load(p, ii0n, wq, nx);
        unsigned char *tnext = tables + (size_t)((b ^ 1) * SHF_GROUPS + grp) * SHP_TBUF;
        unsigned char *cnext = cbufs + (size_t)((b ^ 1) * SHF_GROUPS + grp) * SHP_CBUF;
        if (LIGHT && more) shp_table_request(p, ii0n, tnext, cnext);
        SHF_STAMP(1);

        ShfAtom at;
        at.ii = ii0; at.i = cur.i; at.n = cur.n;
        double dx[SHF_CC], dy[SHF_CC], dz[SHF_CC];
        int jn[SHF_CC];
        bool has_nbr[SHF_CC];
        const double *crl;
        unsigned tb;
        {
            const int ln = shp_lane();
            const int g = ln >> 4, l = ln & 15;
#pragma unroll
            for (int u = 0; u < SHF_CC; u++) {
                jn[u] = cur.jn[u]; has_nbr[u] = l + SHF_GL * (SHF_CC * wq + u) < cur.n;
                dx[u] = pos.xi - pos.xj[u]; dy[u] = pos.yi - pos.yj[u]; dz[u] = pos.zi - pos.zj[u];
            }
            crl = reinterpret_cast<const double *>(cbufs + (size_t)(b * SHF_GROUPS + grp) * SHP_CBUF) + g * SHP_CROW;
            tb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)(tables + (size_t)(b * SHF_GROUPS + grp) * SHP_TBUF + (size_t)g * SHF_TBYTES);
        }
        ShfTable tab;
        tab.key = keys + b * SHP_KS; tab.acc = acc; tab.f = p.f;
        int ccn = 0;
        // between the columns and the inserts: the next unit's positions are requested, the previous flush is known to be over
        auto mid = [&]() {
            SHF_STAMP(2);
            if (more) ccn = shp_head_resolve(p, nx, ii0n, wq, pos);
            shp_wait_for(c_flushed, (unsigned)(SHF_GROUPS * s));
            SHF_STAMP(3);
        };
        if (cc > 0) {
            at.pone = crl[SHP_CP1];
            double fi[3] = {0.0, 0.0, 0.0}, vs[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            if (cc == 1) shf_turn<NP, 1, VIRIAL>(p, at, has_nbr, jn, dx, dy, dz, crl, p.shp_pi_over_rc, p.shp_two_over_rcp, tb, tab, fi, vs, mid);
            else shf_turn<NP, 2, VIRIAL>(p, at, has_nbr, jn, dx, dy, dz, crl, p.shp_pi_over_rc, p.shp_two_over_rcp, tb, tab, fi, vs, mid);
            // ---- the centre's share: sums over the atom's 16 lanes end up in the row's last lane
            const bool last = (shp_lane() & 15) == 15 && at.n > 0;
#pragma unroll
            for (int k = 0; k < 3; k++) fi[k] = row16_sum_to_last(fi[k]);
            if (last) tab.add(at.i, fi[0], fi[1], fi[2]);
            if (VIRIAL) {
#pragma unroll
                for (int k = 0; k < 6; k++) vs[k] = row16_sum_to_last(vs[k]);
                if (last) {
                    if (p.virial) {
                        double *vr = virial_row(p.virial);          // (annp_common.hpp: the global virial)
#pragma unroll
                        for (int k = 0; k < 6; k++) atomicAdd(&vr[k], vs[k]);
                    }
                    if (p.vatom) {
                        double *vi = p.vatom + 6 * (size_t)at.i;
#pragma unroll
                        for (int k = 0; k < 6; k++) atomicAdd(vi + k, 0.5 * vs[k]);
                    }
                }
            }
        } else {
            mid();
        }
        SHF_STAMP(4);
        shp_raise(c_adds);                                  // this wave's inserts of the step are in the table
        if (LIGHT) {
            if (more) {
                shp_table_finish<false>(tnext, cnext, dumpbuf);
                shp_raise(c_ready);                         // the group's next table is built
            }
            SHF_STAMP(5);
            shp_wait_for(c_adds, (unsigned)(SHF_WAVES * (s + 1)));       // every wave's
            SHF_STAMP(6);
            shp_flush_clear(keys + b * SHP_KS, keys + (b ^ 1) * SHP_KS, acc, p.f, grp * 64 + shp_lane(), p.tab_spills);
            shp_raise(c_flushed);
        } else {
            SHF_STAMP(5);
            if (more) shp_wait_for(c_ready, (unsigned)(s + 1));
            SHF_STAMP(6);
        }
        SHF_STAMP(7);
        cur = nx; cc = ccn; ii0 = ii0n;
    }
}

template <int NP, int NT, bool VIRIAL>
__global__ __launch_bounds__(64 * SHF_WAVES, 4) void annp_fe_force_shp(FeArgs p)
{
    static_assert(NT == SH_LMAX + 1 && NP + 2 * NT + 1 <= ANNP_CPAD && NP + 1 <= SHF_GL, "coefficient row: c_m | p_k | W_l | P(1)");
    static_assert(NP <= 10 && NP + NT == 28 && NP + 2 * NT == ANNP_CPAD - 1, "the chunks of shp_table_request");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int wave = uniform(threadIdx.x >> 6);
    const int nunits = (p.inum + SHF_GROUPS * SHF_GA - 1) / (SHF_GROUPS * SHF_GA);
    const int U0 = xcd_block() * p.shp_run;
    const int nsteps = min(p.shp_run, nunits - U0);          // >= 1: the grid is ceil(nunits / run) workgroups
    // ---- who is the light wave of a group?  The hardware deals a workgroup's waves over the CU's four SIMDs in a fixed pattern
    // (waves w and w + 4 share a SIMD, the four waves of a group sit on four SIMDs: measured, tools/shp_stamps.py), so with the
    // role tied to the wave's number both light waves of a workgroup share a SIMD and three SIMDs carry heavy waves only.  The
    // role follows the SIMD the wave finds itself on instead: the two resident workgroups of a CU (thread-group slots 0 and 1)
    // put their light waves on SIMDs 0, 1 and 2, 3, one light and three heavy waves per SIMD.  Nothing depends on the pattern
    // for its result: the waves of a group say which role they took, and if the four roles are not all there the wave's
    // number decides, as before.
    int wq = wave % SHF_GW;
    {
        const unsigned hwid = __builtin_amdgcn_s_getreg(0xF804);                 // HW_ID: SIMD [5:4], thread-group slot [19:16]
        const int simd = (int)(hwid >> 4) & 3, tg = (int)(hwid >> 16) & 1, grp = wave / SHF_GW;
        const int role = (simd - ((tg << 1) | grp) + 3) & 3;                     // 3: the light wave
        int *said = reinterpret_cast<int *>(lds_raw);
        if (threadIdx.x == 0) *said = 0;
        __syncthreads();
        if (lane_id() == 0) atomicOr(said, 1 << (grp * SHF_GW + role));
        __syncthreads();
        const bool all_there = *said == (1 << SHF_WAVES) - 1;
        __syncthreads();                                                         // (the word is the force table's first: cleared again below)
        if (uniform(all_there) && !p.shp_roles_by_number) wq = uniform(role);
    }
#if defined(SHP_ONLY_HEAVY)
    shp_walk<NP, NT, VIRIAL, false>(p, lds_raw, U0, nsteps, wave, wq);
#elif defined(SHP_ONLY_LIGHT)
    shp_walk<NP, NT, VIRIAL, true>(p, lds_raw, U0, nsteps, wave, wq);
#else
    if (wq == SHF_GW - 1) shp_walk<NP, NT, VIRIAL, true>(p, lds_raw, U0, nsteps, wave, wq);
    else shp_walk<NP, NT, VIRIAL, false>(p, lds_raw, U0, nsteps, wave, wq);
#endif
}

}  // namespace annp
