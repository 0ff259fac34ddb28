// fe_shf_kernels.hpp -- annp_fe_force_sh, the Chebyshev force pass (fe_v2/src/pair_annp.cpp:190-213 with the angular derivatives of
// fe:658-695, "fe:") on the moments of the neighbourhood, round 4: Horner's rule in z and in w = x + iy.
//
// What is computed is what fe_sh_kernels.hpp's header derives: with P(c) = sum_l W_l P_l(c) the atom's angular polynomial,
//     U(e) = sum_b fc_b P(e.e_b) = sum_m Re[ (beta^c_m(z) - i beta^s_m(z)) w^m ],   beta_m(z) = sum_k B_(m+k,m) Pm^(m)_k(z),  B = W kappa A
// and the force on neighbour a needs U(e_a) and grad U(e_a).  Round 3 ran the three-term recurrence of Pm^(m)_k for every
// neighbour and column (2 instructions per step) and four sums behind it (4): 6 x 190 per neighbour.  Here beta_m is handed
// over in powers of z,
//     beta_m(z) = sum_j b_mj z^j,      b_mj = sum_{k >= j, k = j mod 2} M^(m)_kj B_(m+k,m)      (Pm^(m)_k = sum_j M_kj z^j, exact rationals),
// built once per atom (1 430 multiply-adds), and a neighbour evaluates value and derivative by Horner's rule,
//     d <- d z + p,   p <- p z + b_mj        (j = K-1 .. 0):      4 instructions per step for beta^c, beta^s, beta'^c, beta'^s
// -- no basis to generate, half the table (the derivative needs no coefficients of its own: 3 040 B per atom).  The columns are
// summed by Horner's rule in w, from m = 18 down:
//     D <- D w + F,   F <- F w + (beta^c_m - i beta^s_m),   H <- H w + (beta'^c_m - i beta'^s_m)         (complex: 12 FMAs per column)
// so that U = Re F, (dU/dx, dU/dy) = (Re D, -Im D) (D = dF/dw), dU/dz = Re H, and no power of w is kept.
// Conditioning: |z| <= 1 and the b_mj of a degree-18 column alternate with magnitudes up to ~1e5 |B|; measured on bcc-Fe
// neighbourhoods with the potential's own coefficients the forces differ from the recurrence form by <= 1e-14 eV/A
// (tests/test_sh_tables.py::test_force_form_on_monomials, against long double).
//
// Work decomposition: a GROUP is four atoms, 16 lanes each (as in annp_fe_desc_sh), and four waves work for it: wave q of the
// group takes neighbours a = l + 16 u, u = 2q, 2q+1 of every lane (bcc Fe: 112 = 16 x (2+2+2+1), every lane slot used; round 3
// gave a wave one atom and 128 slots) and a quarter of the table's columns to build.  Two neighbours per lane in registers
// are 120 VGPRs: four waves per SIMD -- a wave by itself issues at most every other slot of the FP64 pipe (measured: the
// same arithmetic in waves of 250 VGPRs, two per SIMD, one of them waiting half of its life, ran 5.8 ms per 1 M atoms for
// 4.1 ms of issue slots; tools/shf_stamps.py).  The 16 lanes of an atom read the same table entry: a ds_read_b128 with four
// distinct addresses per wave serves 8 FMAs.  A workgroup is two groups = 8 waves sharing the LDS force table.
#pragma once
#include "fe_sh_kernels.hpp"

namespace annp {

constexpr int SHF_GA = 4;             // atoms per group
constexpr int SHF_GL = 16;            // lanes per atom
constexpr int SHF_GW = 4;             // waves per group
constexpr int SHF_CC = 2;             // neighbours of a lane in a wave's registers: SHF_GW * SHF_CC * 16 = 128 slots per atom
#ifndef ANNP_SHF_GROUPS
#define ANNP_SHF_GROUPS 1
#endif
#ifndef ANNP_SHF_WPS
#define ANNP_SHF_WPS 4          // resident waves per SIMD the kernel is compiled for
#endif
constexpr int SHF_GROUPS = ANNP_SHF_GROUPS;        // groups per workgroup
constexpr int SHF_WAVES = SHF_GW * SHF_GROUPS;     // waves per workgroup
constexpr int SHF_BBITS = SHF_GROUPS >= 2 ? 7 : 6;
constexpr int SHF_NBUCK = 1 << SHF_BBITS;          // buckets of the workgroup's force table
constexpr int SHF_BATOMS = 8;                      // atoms per bucket: 8 x 24 B = 192 B = three 64-byte lines of f
constexpr int SHF_TPROBE = 8;                      // occupied buckets tried before a contribution goes straight to global memory
constexpr int SHF_TBYTES = SHF_NE * 16;            // an atom's coefficient table: 190 x (b^c, b^s)
static_assert((SHF_TBYTES / 4) % 64 >= 4 && (SHF_TBYTES / 4) % 64 <= 60, "two atoms' entries of one read must not share banks");
constexpr int SHF_SLOTS = SHF_GW * SHF_CC * SHF_GL;     // 128 neighbour slots per atom in the waves' registers; the slots above (a denser system: up to
                                                        // SH_CAP_MAX = 160) are an extra turn of one neighbour per lane for the waves at places 0, 1, ..
constexpr int SHF_XWAVES = (SH_CAP_MAX - SHF_SLOTS) / SHF_GL;
static_assert(SH_CAP_MAX >= SHF_SLOTS && (SH_CAP_MAX - SHF_SLOTS) % SHF_GL == 0 && SHF_XWAVES <= SHF_GW, "the waves of a group cover the neighbour slots of an atom");

__constant__ unsigned char annp_shf_l[SHF_NE + 2] = ANNP_SHF_L_INIT;
__constant__ double annp_shf_conv[SHF_CONV_NREC * 16] = ANNP_SHF_CONV_INIT;

// LDS of a workgroup: the force table | per group: the four tables + 32 bytes of slack (look-ahead reads, idle lanes' writes) |
// per wave: c_m [9] and P(1) of its four atoms
// (round 5: 256 bytes of zeros in front of the first group's tables -- the change of basis reads up to 240 bytes below a table with
// a zero coefficient, and what lies there must be a finite number -- and per wave 1 KB where the lanes that have nothing to
// write write, a slot each)
constexpr int SHF_WPAD = 10;
constexpr int SHF_ZPAD = 256, SHF_DUMP = 1024;
__host__ __device__ constexpr size_t shf_lds_table() { return (size_t)SHF_NBUCK * (SHF_BATOMS * 24 + 16) + SHF_ZPAD; }
__host__ __device__ constexpr size_t shf_lds_per_group() { return (size_t)SHF_GA * SHF_TBYTES + 32; }
__host__ __device__ constexpr size_t shf_lds_per_wave() { return (size_t)SHF_GA * SHF_WPAD * 8 + SHF_DUMP; }
__host__ __device__ constexpr size_t shf_lds_per_block() { return shf_lds_table() + SHF_GROUPS * shf_lds_per_group() + SHF_WAVES * shf_lds_per_wave(); }
static_assert(shf_lds_per_group() % 16 == 0 && shf_lds_table() % 16 == 0, "b128 alignment of every group's tables");

// Forces leave through a table in LDS that the atoms of a workgroup share (round 3), and the table leaves through as few
// memory requests as it can.  A float atomic is executed at the memory side, one request per 64-byte line a wave-instruction
// touches (MI355X_MICROARCH.md, Global float atomics): with one lane per (atom, component) scattered over the force array the
// pass issued ~76 requests per atom and ran at their rate -- 4.6 ms per 1 M atoms with the arithmetic switched off, of 6.1.
// A bucket therefore holds EIGHT atoms with consecutive indices (24 doubles = three lines), open addressing is on index >> 3,
// and the flush walks the table in memory order: a wave-instruction writes 64 consecutive doubles of the table = 2.7 buckets
// = 8-9 lines.  Atoms that are neighbours in space are neighbours in index wherever the caller sorts its atoms (LAMMPS:
// atom_modify sort, on by default); where they are not, a bucket holds one atom, the table fills up (SHF_NBUCK buckets) and the
// contributions beyond it go to memory one by one -- never wrong (tests/test_gpu_parity.py::test_fe_atoms_in_random_order), and
// 2.3 times the time of the pass when the order is random (tools/kbench.py, KBENCH_SHUFFLE: 0.83 -> 1.90 ms per 128 000 atoms).
// No table changes that: eight atoms picked at random share no neighbours, their 900 force targets are 900 lines of f, against
// 18 requests per atom in a sorted box (a table with a bucket per atom and eight times the buckets was tried: 1.90 ms as well).
// The contributions that found no bucket are counted (FeArgs::tab_spills) and the host says so once (annp_hip_set_notice).
// Untouched slots hold +0.0 and are skipped.
struct ShfTable {
    int *key;          // [SHF_NBUCK]: index >> 3 of the bucket's atoms, -1 = free; key[SHF_NBUCK]: contributions that found no bucket
    double *acc;       // [SHF_NBUCK][8][3]
    double *f;
#ifdef ANNP_SHF_CHECK      // developer build: every index that comes from data is checked against nall, a bad one is reported through the error word instead of used
    int nall; int *err;
#endif
    __device__ __forceinline__ void add(int j, double fx, double fy, double fz) const
    {
#ifdef ANNP_SHF_CHECK
        if ((unsigned)j >= (unsigned)nall) { atomicMax(err, 3000000 + (blockIdx.x & 0xffff)); return; }
#endif
        const int b = j >> 3;
        unsigned h = ((unsigned)b * 0x9E3779B1u) >> (32 - SHF_BBITS);
#ifdef ANNP_SHF_NOPROBE     // developer timing build (wrong forces): the bucket the hash names, whoever holds it -- what a probe costs
        { double *a = acc + (h * SHF_BATOMS + (j & 7)) * 3; key[h] = b; atomicAdd(a, fx); atomicAdd(a + 1, fy); atomicAdd(a + 2, fz); return; }
#endif
#pragma unroll 1
        for (int probe = 0; probe < SHF_TPROBE; probe++) {
            // most contributions find their bucket taken by an earlier one: a plain read settles those, the compare-and-swap
            // (an atomic with a return value) is for the free ones only
            int old = __hip_atomic_load(&key[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (old == -1) old = atomicCAS(&key[h], -1, b);
            if (old == -1 || old == b) {
                double *a = acc + (h * SHF_BATOMS + (j & 7)) * 3;
                atomicAdd(a, fx); atomicAdd(a + 1, fy); atomicAdd(a + 2, fz);
                return;
            }
            h = (h + 1) & (SHF_NBUCK - 1);
        }
        atomicAdd(&key[SHF_NBUCK], 1);
        atomicAdd(&f[3 * (size_t)j], fx); atomicAdd(&f[3 * (size_t)j + 1], fy); atomicAdd(&f[3 * (size_t)j + 2], fz);
    }
    // (both walk the table with compile-time strides: a handful of instructions per pass instead of an index computation per
    // element -- four waves work for four atoms here, and what a wave does per workgroup it does per atom)
    __device__ __forceinline__ void clear(int tid) const
    {
        constexpr int NT2 = SHF_NBUCK * SHF_BATOMS * 3 / 2, TH = 64 * SHF_WAVES;
        double2 *a2 = reinterpret_cast<double2 *>(acc);
#pragma unroll
        for (int k = 0; k < (NT2 + TH - 1) / TH; k++)
            if (k * TH + TH <= NT2 || tid + k * TH < NT2) a2[tid + k * TH] = make_double2(0.0, 0.0);
        if (tid <= SHF_NBUCK) key[tid] = tid < SHF_NBUCK ? -1 : 0;
        if (tid < 6) vacc()[tid] = 0.0;
    }
    // six doubles behind the keys: the workgroup's share of the global virial (round 5: gathered here and sent to the evaluation's
    // virial table as ONE six-lane atomic per workgroup; every wave sending its atoms' six sums itself was 24 requests per workgroup on
    // one 64-byte line, and same-line float atomics are served one after the other: the virial cost the pass 6.7 %)
    __device__ __forceinline__ double *vacc() const { return reinterpret_cast<double *>(key + SHF_NBUCK + 2 + ((SHF_NBUCK + 2) & 1)); }
    // thread t < 384 takes double t % 24 of buckets t / 24, t / 24 + 16, ..: a wave-instruction still writes 64 consecutive doubles of
    // the table (2.7 buckets = 8-9 lines of f)
    __device__ __forceinline__ void flush(int tid, int *spills) const
    {
        constexpr int PB = 64 * SHF_WAVES >= 16 * SHF_BATOMS * 3 ? 16 : 8;         // buckets per pass
        static_assert(64 * SHF_WAVES >= PB * SHF_BATOMS * 3 && SHF_NBUCK % PB == 0, "whole buckets per pass");
        if (tid < PB * SHF_BATOMS * 3) {
            const int h0 = tid / (SHF_BATOMS * 3), r = tid - h0 * (SHF_BATOMS * 3);
#pragma unroll
            for (int it = 0; it < SHF_NBUCK / PB; it++) {
                const int b = key[h0 + PB * it];
                const double v = acc[tid + PB * SHF_BATOMS * 3 * it];
#ifdef ANNP_SHF_CHECK
                if (b >= 0 && v != 0.0 && (size_t)b * (SHF_BATOMS * 3) + r >= (size_t)nall * 3) { atomicMax(err, 4000000 + (h0 + PB * it) * 1000 + (tid & 511)); continue; }
                if (b < -1) { atomicMax(err, 5000000 + (h0 + PB * it)); continue; }
#endif
                if (b >= 0 && v != 0.0) atomicAdd(&f[(size_t)b * (SHF_BATOMS * 3) + r], v);
            }
        }
        // (no look at the word before adding: a load here makes the workgroup wait for its flush -- +25 % on the pass at 128 000 atoms; the
        // host reads the word as unsigned, so the count is good up to 4e9 contributions: ADVICE r4)
        if (tid == 64 * SHF_WAVES - 1 && spills && key[SHF_NBUCK]) atomicAdd(spills, key[SHF_NBUCK]);
    }
};

// a * b + c in the three-address form.  Left to itself the compiler writes the steps below as v_fmac_f64 (d += a * b) on a copy of the
// table entry -- a v_mov_b64 per multiply-add whose addend is shared by the neighbours of a lane, a quarter of the inner loop.
__device__ __forceinline__ double fma3(double a, double b, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ double fma3n(double a, double b, double c)         // a * b - c
{
    double r;
    asm("v_fma_f64 %0, %1, %2, -%3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ double fnma3(double a, double b, double c)         // c - a * b
{
    double r;
    asm("v_fma_f64 %0, -%1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
typedef const __attribute__((address_space(3))) shf_v2d *shf_tab_ptr;
__device__ __forceinline__ shf_v2d shf_entry(unsigned tb, int e) { return *(shf_tab_ptr)(uintptr_t)(tb + 16u * (unsigned)e); }

// ---- the change of basis of one column, in place: entry (m, j) <- sum_t M_(j+2t, j) B_(m, j+2t).  Lane l of an atom takes
// the powers j = l and (K > 16) j = 16 + l; every read of a block of 16 powers is issued before its writes (LDS operations of a
// wave execute in order), and the second block only reads entries the first has not written.
template <int M, int BLK>
__device__ __forceinline__ void shf_convert_block(const unsigned tcol, const unsigned dump, const double *conv, const int l)
{
    // tcol: LDS byte address of the column's first entry (power K-1) in this lane's table.  Power k sits at tcol + 16 (K-1-k); lane l
    // reads the powers l + 2t, t = 0, 1, .. (clamped to the column: the record holds a zero there) and writes power l.
    constexpr int K = SH_LMAX + 1 - M;
    constexpr int first = SHF_CONV_FIRST[M][BLK / 16];
    constexpr int nt = (K - BLK + 1) / 2;
    const unsigned p0 = tcol + 16u * (unsigned)(K - 1 - BLK) - 16u * (unsigned)l;        // (wraps below tcol for lanes without a power)
    double ax = 0.0, ay = 0.0;
#pragma unroll
    for (int t = 0; t < nt; t++) {
        const double mv = conv[(first + t) * 16 + l];        // 0 where l + 2t > K-1
        const unsigned a = (unsigned)max((int)(p0 - 32u * (unsigned)t), (int)tcol);
        const shf_v2d b = *(shf_tab_ptr)(uintptr_t)a;
        ax = fma(mv, b.x, ax); ay = fma(mv, b.y, ay);
    }
    // (no branch around the block: with one, the compiler sinks the block's loads into it and waits for them there, a round trip
    // through the cache per block; a lane without a power writes the slack behind the group's tables)
    typedef __attribute__((address_space(3))) shf_v2d *wptr;
    shf_v2d r; r.x = ax; r.y = ay;
    *(wptr)(uintptr_t)(BLK + l < K ? p0 : dump) = r;
}
template <int M>
__device__ __forceinline__ void shf_convert_column(const unsigned tb, const unsigned dump, const double *conv, const int l)
{
    const unsigned tcol = tb + 16u * (unsigned)shf_toff(M);
    shf_convert_block<M, 0>(tcol, dump, conv, l);
    if (SH_LMAX + 1 - M > 16) shf_convert_block<M, (SH_LMAX + 1 - M > 16 ? 16 : 0)>(tcol, dump, conv, l);
}
// ---- value and gradient of U at CC neighbours of this lane.  tb = LDS byte address of the atom's table.
template <int CC>
__device__ __forceinline__ void shf_evaluate(const unsigned tb, const double (&z)[CC], const double (&wx)[CC], const double (&wy)[CC],
                                             double (&U)[CC], double (&Ux)[CC], double (&Uy)[CC], double (&Uz)[CC])
{
    double Fx[CC], Fy[CC], Dx[CC], Dy[CC], Hx[CC], Hy[CC];
    double pc[CC], ps[CC], dc[CC], ds[CC];
    // a column is done: D <- D w + F, F <- F w + (pc - i ps), H <- H w + (dc - i ds)
    auto column_end = [&]() {
#pragma unroll
        for (int u = 0; u < CC; u++) {
            const double ndx = fma3(Dx[u], wx[u], fnma3(Dy[u], wy[u], Fx[u]));
            const double ndy = fma3(Dx[u], wy[u], fma3(Dy[u], wx[u], Fy[u]));
            const double nfx = fma3(Fx[u], wx[u], fnma3(Fy[u], wy[u], pc[u]));
            const double nfy = fma3(Fx[u], wy[u], fma3n(Fy[u], wx[u], ps[u]));
            const double nhx = fma3(Hx[u], wx[u], fnma3(Hy[u], wy[u], dc[u]));
            const double nhy = fma3(Hx[u], wy[u], fma3n(Hy[u], wx[u], ds[u]));
            Dx[u] = ndx; Dy[u] = ndy; Fx[u] = nfx; Fy[u] = nfy; Hx[u] = nhx; Hy[u] = nhy;
            // one neighbour after the other: interleaved, the four neighbours' twelve intermediate values each are what spills
            asm volatile("" : "+v"(Dx[u]), "+v"(Dy[u]), "+v"(Fx[u]), "+v"(Fy[u]), "+v"(Hx[u]), "+v"(Hy[u]));
        }
    };
    // one step of Horner's rule with derivative (the derivative first: it does not wait for the entry)
    auto step = [&](const shf_v2d b) {
#pragma unroll
        for (int u = 0; u < CC; u++) { dc[u] = fma3(dc[u], z[u], pc[u]); ds[u] = fma3(ds[u], z[u], ps[u]); }
#pragma unroll
        for (int u = 0; u < CC; u++) { pc[u] = fma3(pc[u], z[u], b.x); ps[u] = fma3(ps[u], z[u], b.y); }
    };
    // the first three entries of a column at once: p = (b1 z + b2) z + b3, d = b1 z + (b1 z + b2)
    auto start3 = [&](const shf_v2d b1, const shf_v2d b2, const shf_v2d b3) {
#pragma unroll
        for (int u = 0; u < CC; u++) {
            const double qc = fma3(b1.x, z[u], b2.x), qs = fma3(b1.y, z[u], b2.y);
            dc[u] = fma3(b1.x, z[u], qc); ds[u] = fma3(b1.y, z[u], qs);
            pc[u] = fma3(qc, z[u], b3.x); ps[u] = fma3(qs, z[u], b3.y);
        }
    };
    // ---- m = 18 (one entry) and m = 17 (two): no loop to speak of
    {
        const shf_v2d b0 = shf_entry(tb, 0), b1 = shf_entry(tb, 1), b2 = shf_entry(tb, 2);
#pragma unroll
        for (int u = 0; u < CC; u++) {
            Dx[u] = 0.0; Dy[u] = 0.0; Hx[u] = 0.0; Hy[u] = 0.0;
            Fx[u] = b0.x; Fy[u] = -b0.y;                                  // F = beta_18
            pc[u] = fma3(b1.x, z[u], b2.x); ps[u] = fma3(b1.y, z[u], b2.y); // beta_17 = b1 z + b2
            dc[u] = b1.x; ds[u] = b1.y;
        }
        column_end();
    }
    // ---- m = 16 .. 1: K = 3 .. 18 entries, two columns per trip (K odd: start3 + pairs; K even: start3 + one step + pairs).
    // qa, qb always hold the next two entries of the table; a column's third (s3) and an even column's fourth entry (sx) are
    // requested before the closing arithmetic of the column before, the pairs' entries one step ahead.  Every request has a
    // register of its own: no moves.
    unsigned tp = tb + 16u * 3u;             // byte address of the next entry of the table
    shf_v2d qa = shf_entry(tp, 0), qb = shf_entry(tp, 1), s3 = shf_entry(tp, 2);
#pragma unroll 1
    for (int i = 0; i < 8; i++) {
        // K = 3 + 2i: i pairs behind the start
        start3(qa, qb, s3);
        qa = shf_entry(tp, 3); qb = shf_entry(tp, 4);
        tp += 16u * 3u;
#pragma unroll 1
        for (int r = 0; r < i; r++) {
            step(qa); qa = shf_entry(tp, 2);
            step(qb); qb = shf_entry(tp, 3);
            tp += 16u * 2u;
        }
        // K = 4 + 2i starts at tp (qa, qb): start3, the odd step, i pairs
        s3 = shf_entry(tp, 2);
        const shf_v2d sx = shf_entry(tp, 3);
        column_end();
        start3(qa, qb, s3);
        qa = shf_entry(tp, 4); qb = shf_entry(tp, 5);
        step(sx);
        tp += 16u * 4u;
#pragma unroll 1
        for (int r = 0; r < i; r++) {
            step(qa); qa = shf_entry(tp, 2);
            step(qb); qb = shf_entry(tp, 3);
            tp += 16u * 2u;
        }
        s3 = shf_entry(tp, 2);               // the next column's third entry
        column_end();
    }
    // ---- m = 0: 19 entries, cosine only (the sine moments of m = 0 are zero), and only the real parts are wanted at the end
    {
#pragma unroll
        for (int u = 0; u < CC; u++) {
            const double qc = fma3(qa.x, z[u], qb.x);
            dc[u] = fma3(qa.x, z[u], qc);
            pc[u] = fma3(qc, z[u], s3.x);
        }
        qa = shf_entry(tp, 3); qb = shf_entry(tp, 4);
        tp += 16u * 3u;
#pragma unroll 1
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int u = 0; u < CC; u++) { dc[u] = fma3(dc[u], z[u], pc[u]); pc[u] = fma3(pc[u], z[u], qa.x); }
            qa = shf_entry(tp, 2);
#pragma unroll
            for (int u = 0; u < CC; u++) { dc[u] = fma3(dc[u], z[u], pc[u]); pc[u] = fma3(pc[u], z[u], qb.x); }
            qb = shf_entry(tp, 3);
            tp += 16u * 2u;
        }
#pragma unroll
        for (int u = 0; u < CC; u++) {
            Ux[u] = fma3(Dx[u], wx[u], fnma3(Dy[u], wy[u], Fx[u]));
            Uy[u] = -fma3(Dx[u], wy[u], fma3(Dy[u], wx[u], Fy[u]));
            U[u] = fma3(Fx[u], wx[u], fnma3(Fy[u], wy[u], pc[u]));
            Uz[u] = fma3(Hx[u], wx[u], fnma3(Hy[u], wy[u], dc[u]));
        }
    }
}

struct ShfAtom {            // what a lane knows of its atom
    int ii, i, n;           // list entry, atom index, in-cutoff neighbours (0: nothing to do)
    double xi, yi, zi;
    double pone;            // P(1) = sum_k p_k
};

// ---- the table's columns are built by the four waves of the group, a quarter each (26 of the 104 change-of-basis records)
constexpr int SHF_WCOLS[SHF_GW][8] = {{0, 5, 9, 15, -1, -1, -1, -1}, {1, 4, 10, 13, -1, -1, -1, -1},
                                      {2, 3, 11, 12, -1, -1, -1, -1}, {6, 7, 8, 14, 16, 17, 18, -1}};
// entries of a wave's columns a lane stages: position p = l of every column, then p = 16 + l of the columns longer than 16
// (m = 0, 1, 2: one each in waves 0, 1, 2, always their first column)
constexpr int SHF_NSTAGE = 8;
template <int WQ, int IDX>
struct ShfBuild {
    static constexpr int M = SHF_WCOLS[WQ][IDX];
    static constexpr int K = SH_LMAX + 1 - (M >= 0 ? M : 0);
    // request this lane's moments of the wave's columns
    static __device__ __forceinline__ void load(const double2 *Am, const int l, double2 (&am)[SHF_NSTAGE])
    {
        if constexpr (M >= 0) {
            // (Am: this lane's pointer into the row, entry l.  A lane beyond the column's end reads the next column's entries, which it
            // does not use; only the second round of a long column could leave the row)
            am[IDX] = Am[shf_toff(M)];
            if (K > 16) am[SHF_NSTAGE - 1] = Am[shf_toff(M) + min(16, K - 1 - l)];
            ShfBuild<WQ, IDX + 1>::load(Am, l, am);
        }
    }
    // B = W kappa A into the table: position p of any column belongs to l = 18 - p
    static __device__ __forceinline__ void stage(double2 *T, double2 *dump, const int l, const double (&am_w)[2], const double2 (&am)[SHF_NSTAGE])
    {
        if constexpr (M >= 0) {
            *(l < K ? T + shf_toff(M) + l : dump) = make_double2(am_w[0] * am[IDX].x, am_w[0] * am[IDX].y);
            if (K > 16) *(16 + l < K ? T + shf_toff(M) + 16 + l : dump) = make_double2(am_w[1] * am[SHF_NSTAGE - 1].x, am_w[1] * am[SHF_NSTAGE - 1].y);
            ShfBuild<WQ, IDX + 1>::stage(T, dump, l, am_w, am);
        }
    }
    static __device__ __forceinline__ void convert(const unsigned tb, const unsigned dump, const double *conv, const int l)
    {
        if constexpr (M >= 0) {
            shf_convert_column<M>(tb, dump, conv, l);
#ifdef SHP_FENCE_COLUMNS
            asm volatile("" ::: "memory");
#endif
            ShfBuild<WQ, IDX + 1>::convert(tb, dump, conv, l);
        }
    }
};
template <int WQ>
struct ShfBuild<WQ, 8> {
    static __device__ __forceinline__ void load(const double2 *, const int, double2 (&)[SHF_NSTAGE]) {}
    static __device__ __forceinline__ void stage(double2 *, double2 *, const int, const double (&)[2], const double2 (&)[SHF_NSTAGE]) {}
    static __device__ __forceinline__ void convert(const unsigned, const unsigned, const double *, const int) {}
};

// ---- the change of basis, round 5: every address an immediate.  Power k of column M sits at tb + 16 (toff(M) + K-1-k); lane l
// reads the powers BLK + l + 2t at tl + 16 (toff(M) + K-1-BLK) - 32 t, tl = tb - 16 l, and writes power BLK + l.  A lane without
// the power, or past the column's top, reads up to 240 bytes below its column -- the column before, the table before, the zeros
// in front of the first table: finite numbers all (every wave of the workgroup has staged its columns: the barrier in front) --
// with a zero coefficient; round 4 clamped each address into the column instead (two instructions per term, as many as the
// arithmetic) and took the coefficients through per-term 64-bit address sums (four more).  Now a term is its two multiply-adds.
// No branch around a write (the compiler sinks the reads into it and waits for each where it is used): a lane that has nothing
// to write writes its slot of the wave's dump.  And the order is pinned -- coefficients of a piece requested while the piece
// before is worked on, a piece's reads together -- because under this kernel's register limit the compiler otherwise issues one
// read, waits, issues the next.
typedef const double __attribute__((address_space(1))) *shf_gcp;
typedef const char __attribute__((address_space(1))) *shf_gbp;
typedef __attribute__((address_space(3))) shf_v2d *shf_tab_wptr;
// lanes l < n of every atom, as an execution mask turned select (no compare: a literal in scalar registers)
__device__ __forceinline__ bool shf_lanes_below(const int n)
{
    const unsigned long long m16 = n >= 16 ? 0xffffull : ((1ull << n) - 1ull);
    return __builtin_amdgcn_inverse_ballot_w64(m16 * 0x0001000100010001ull);
}
struct ShfPiece { int m, blk; };         // the powers blk .. blk + 15 of column m
constexpr int SHF_WNP[SHF_GW] = {5, 5, 5, 7};
constexpr ShfPiece SHF_WPIECE[SHF_GW][7] = {{{0, 0}, {0, 16}, {5, 0}, {9, 0}, {15, 0}, {-1, 0}, {-1, 0}},
                                            {{1, 0}, {1, 16}, {4, 0}, {10, 0}, {13, 0}, {-1, 0}, {-1, 0}},
                                            {{2, 0}, {2, 16}, {3, 0}, {11, 0}, {12, 0}, {-1, 0}, {-1, 0}},
                                            {{6, 0}, {7, 0}, {8, 0}, {14, 0}, {16, 0}, {17, 0}, {18, 0}}};
constexpr int SHF_MAXT = (SH_LMAX + 2) / 2;          // terms of the longest piece
template <int WQ, int P>
struct ShfConv {
    static constexpr int M = SHF_WPIECE[WQ][P].m, BLK = SHF_WPIECE[WQ][P].blk;
    static constexpr int K = SH_LMAX + 1 - M;
    static constexpr int first = SHF_CONV_FIRST[M][BLK / 16];
    static constexpr int nt = (K - BLK + 1) / 2;
    static constexpr unsigned a0 = 16u * (unsigned)(shf_toff(M) + K - 1 - BLK);
    static_assert(a0 + 32u >= 32u * nt && nt <= SHF_MAXT, "no read below the column before");
    static __device__ __forceinline__ void request(const shf_gbp convl, double (&mv)[SHF_MAXT])
    {
#pragma unroll
        for (int t = 0; t < nt; t++) mv[t] = *(shf_gcp)(convl + (first + t) * 128);        // 0 where l + 2t > K-1
    }
};
template <int WQ, int P>
__device__ __forceinline__ void shf_convert_pieces(const unsigned tl, const unsigned dump, const shf_gbp convl, const double (&mv)[SHF_MAXT])
{
    if constexpr (P < SHF_WNP[WQ]) {
        typedef ShfConv<WQ, P> C;
        double mvn[SHF_MAXT];
        if constexpr (P + 1 < SHF_WNP[WQ]) ShfConv<WQ, P + 1>::request(convl, mvn);
        shf_v2d b[C::nt];
#pragma unroll
        for (int t = 0; t < C::nt; t++) b[t] = *(shf_tab_ptr)(uintptr_t)(tl + C::a0 - 32u * (unsigned)t);
        __builtin_amdgcn_sched_barrier(0);
        double ax = 0.0, ay = 0.0;
#pragma unroll
        for (int t = 0; t < C::nt; t++) { ax = fma(mv[t], b[t].x, ax); ay = fma(mv[t], b[t].y, ay); }
        shf_v2d r; r.x = ax; r.y = ay;
        *(shf_tab_wptr)(uintptr_t)(shf_lanes_below(C::K - C::BLK) ? tl + C::a0 : dump) = r;
        __builtin_amdgcn_sched_barrier(0);
        shf_convert_pieces<WQ, P + 1>(tl, dump, convl, mvn);
    }
}
template <int WQ>
__device__ __forceinline__ void shf_convert_wave(const unsigned tb, const unsigned dump, const int l)
{
    unsigned long long cb = (unsigned long long)(const void *)annp_shf_conv;
    asm volatile("" : "+s"(cb));            // (one base in scalar registers and immediate offsets)
    const shf_gbp convl = (shf_gbp)cb + (unsigned long long)(8u * (unsigned)l);
    double mv[SHF_MAXT];
    ShfConv<WQ, 0>::request(convl, mv);
    shf_convert_pieces<WQ, 0>(tb - 16u * (unsigned)l, dump, convl, mv);
}

// a wave's neighbours of a lane: geometry and radial term (fe:648), U and grad U from the table, force assembly (fe:190-213),
// into the force table.  Nothing in here waits for memory.
// `mid` runs between the columns and the force assembly (the persistent kernel issues the next unit's position loads there).
struct ShfNoMid { __device__ __forceinline__ void operator()() const {} };
template <int NP, int CC, bool VIRIAL, class Mid = ShfNoMid>
__device__ __forceinline__ void shf_turn(const FeArgs &p, const ShfAtom &at, const bool (&has_nbr)[SHF_CC], const int (&jn)[SHF_CC], const double (&dx)[SHF_CC],
                                         const double (&dy)[SHF_CC], const double (&dz)[SHF_CC], const double *crl, const double pi_over_rc,
                                         const double two_over_rcp, const unsigned tb, const ShfTable &tab, double (&fi)[3], double (&vs)[6],
                                         const Mid &mid = Mid())
{
    // F_n = e [ al (e . grad U) - be U + g0 ] - al grad U  with e = (wx, wy, z), al = fc / r, be = fc', g0 = -R + P(1) fc fc'
    double z[CC], wx[CC], wy[CC], al[CC], be[CC], g0[CC], rr[VIRIAL ? CC : 1];
#pragma unroll
    for (int u = 0; u < CC; u++) {
        // (a slot without a neighbour holds the centre's own position: r = 0, and what follows from it in that lane -- infinities,
        // NaNs -- stays in that lane and is dropped where the forces are assembled)
        const double2 R0 = make_double2(dx[u], dy[u]), R1 = make_double2(dz[u], dx[u] * dx[u] + dy[u] * dy[u] + dz[u] * dz[u]);
        const FeNbr g = sh_geometry(R0, R1, pi_over_rc);
        const double xr = g.r * two_over_rcp - 1.0;
        const double y2 = 2.0 * xr;
        double tm2 = 1.0, tm1 = xr, dm2 = 0.0, dm1 = 1.0;
        double st = crl[0], sd = 0.0;              // sum c T, sum c T'  (c_m from LDS: the atom's lanes read the same words)
        if (NP > 1) { const double c1 = crl[1]; st = fma(c1, xr, st); sd = c1; }
#pragma unroll
        for (int mm = 2; mm < NP; mm++) {
            const double cm = crl[mm];
            const double t = fma(y2, tm1, -tm2);
            const double d = fma(y2, dm1, fma(2.0, tm1, -dm2));
            st = fma(cm, t, st);
            sd = fma(cm, d, sd);
            tm2 = tm1; tm1 = t; dm2 = dm1; dm1 = d;
        }
        const double R = fma(sd * two_over_rcp, g.fc, st * g.dfc);        // d/dr of the radial part
        z[u] = g.ez; wx[u] = g.ex; wy[u] = g.ey;
        al[u] = g.fc * g.rinv;
        be[u] = g.dfc;
        g0[u] = fma(at.pone * g.fc, g.dfc, -R);
        if (VIRIAL) rr[u] = g.r;
        // one neighbour after the other (the sine and cosine series of one are two independent chains, and three other waves share
        // the SIMD); both at once is where the 128 registers of a wave run out
        asm volatile("" : "+v"(z[u]), "+v"(wx[u]), "+v"(wy[u]), "+v"(al[u]), "+v"(be[u]), "+v"(g0[u]));
    }
    double U[CC], Ux[CC], Uy[CC], Uz[CC];
#ifdef ANNP_SHF_SKIP_EVAL       // developer timing builds only: everything but the columns
#pragma unroll
    for (int u = 0; u < CC; u++) { U[u] = z[u]; Ux[u] = wx[u]; Uy[u] = wy[u]; Uz[u] = z[u] * wx[u]; }
#else
    shf_evaluate<CC>(tb, z, wx, wy, U, Ux, Uy, Uz);
#endif
    mid();
#pragma unroll
    for (int u = 0; u < CC; u++) {
        if (has_nbr[u]) {
            const double ed = fma(wx[u], Ux[u], fma(wy[u], Uy[u], z[u] * Uz[u]));
            const double t = fma(al[u], ed, fma(-be[u], U[u], g0[u]));
            const double f0 = fma(t, wx[u], -al[u] * Ux[u]);
            const double f1 = fma(t, wy[u], -al[u] * Uy[u]);
            const double f2 = fma(t, z[u], -al[u] * Uz[u]);
#ifndef ANNP_SHF_SKIP_ADD
            tab.add(jn[u], -f0, -f1, -f2);                    // F_a = -Fn_a to the neighbour, +Fn_a to the centre (fe:199-211)
#endif
            fi[0] += f0; fi[1] += f1; fi[2] += f2;
            if (VIRIAL) {   // ev_tally_xyz(i,j,...,fx=-Fj, del = xi-xj = r e)   (fe:201-209)
                const double d0 = rr[u] * wx[u], d1 = rr[u] * wy[u], d2 = rr[u] * z[u];
                const double w0 = d0 * f0, w1 = d1 * f1, w2 = d2 * f2, w3 = d0 * f1, w4 = d0 * f2, w5 = d1 * f2;
                vs[0] += w0; vs[1] += w1; vs[2] += w2; vs[3] += w3; vs[4] += w4; vs[5] += w5;
                if (p.vatom) {
                    double *vj = p.vatom + 6 * (size_t)jn[u];
                    atomicAdd(vj + 0, 0.5 * w0); atomicAdd(vj + 1, 0.5 * w1); atomicAdd(vj + 2, 0.5 * w2);
                    atomicAdd(vj + 3, 0.5 * w3); atomicAdd(vj + 4, 0.5 * w4); atomicAdd(vj + 5, 0.5 * w5);
                }
            }
        }
    }
}

// developer timing builds (tools/shf_stamps.py): -DANNP_SHF_STAMPS writes s_memtime at a few points of a wave's life into the
// descriptor row of an atom of its own (the descriptor buffer is dead by then); no stamp is compiled into the library
#ifdef ANNP_SHF_STAMPS
#define SHF_STAMP(k) do { if (stamp_row && lane_id() == 0) stamp_row[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SHF_STAMP(k) do { } while (0)
#endif

template <int NP, int NT, bool VIRIAL>
__global__ __launch_bounds__(64 * SHF_WAVES, ANNP_SHF_WPS) void annp_fe_force_sh(FeArgs p)
{
    static_assert(NT == SH_LMAX + 1 && NP + 2 * NT + 1 <= ANNP_CPAD && NP + 1 <= SHF_GL, "coefficient row: c_m | p_k | W_l | P(1)");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    ANNP_POISON();
    const int lane = lane_id();
    const int wave = uniform(threadIdx.x >> 6);
    const int grp = wave / SHF_GW;                                   // the wave's group in the workgroup
    // ---- its place in the group: which two neighbour slots of a lane, which columns of the table.  bcc Fe has 16 x 7 neighbours, so
    // place 3 has ONE slot, half the columns' work of the others -- and the hardware deals a workgroup's eight waves over the CU's
    // four SIMDs in a fixed pattern (waves w and w + 4 share a SIMD, the four waves of a group sit on four SIMDs: measured,
    // round 5), so with the place tied to the wave's number both light waves of a workgroup sit on one SIMD and three SIMDs carry
    // heavy waves only.  The place follows the SIMD the wave finds itself on instead: the two workgroups resident on a CU
    // (thread-group slots 0 and 1) put their light waves on SIMDs 0, 1 and 2, 3 -- one light and three heavy waves per SIMD.
    // Nothing depends on that pattern for its result: the waves of a group say which place they took, and if the four places
    // are not all there the wave's number decides, as in round 4.
    int wq = wave % SHF_GW;
    {
        const unsigned hwid = __builtin_amdgcn_s_getreg(0xF804);                 // HW_ID: SIMD [5:4], thread-group slot [19:16]
        const int place = ((int)(hwid >> 4) - (((int)(hwid >> 16) & 15) * SHF_GROUPS + grp) + 3) & 3;     // 3: the light wave
        int *said = reinterpret_cast<int *>(lds_raw);                            // (the force table's first words: cleared behind the next barrier)
        if (lane == 0) said[wave] = place;
        __syncthreads();
        int seen = 0;
#pragma unroll
        for (int k = 0; k < SHF_WAVES; k++) seen |= 1 << ((k / SHF_GW) * SHF_GW + (said[k] & 3));
        const bool all_there = seen == (1 << SHF_WAVES) - 1;
        // (the "& 3" behind the read is for the compiler: with a place it knows nothing about, hipcc 7.2 built the switches below
        // wrongly -- the fourth place's first moment loads went out from registers nobody had written: a memory fault, round 5)
#ifdef ANNP_SHF_NO_PLACE_MASK      // the construct hipcc 7.2 miscompiles: only ever built by tests/test_kernel_resources.py, to show that tools/asm_check.py sees it
        if (uniform(all_there) && !p.shf_places_by_number) wq = uniform(place);
#else
        if (uniform(all_there) && !p.shf_places_by_number) wq = uniform(place) & 3;
#endif
    }
    ShfTable tab;
    tab.key = reinterpret_cast<int *>(lds_raw + (size_t)SHF_NBUCK * SHF_BATOMS * 24);
    tab.acc = reinterpret_cast<double *>(lds_raw);
    tab.f = p.f;
#ifdef ANNP_SHF_CHECK
    tab.nall = p.chk_nall; tab.err = p.errflag;
#endif
    unsigned char *gbase = lds_raw + shf_lds_table() + (size_t)grp * shf_lds_per_group();
    const int g = lane >> 4, l = lane & 15;
    double2 *T = reinterpret_cast<double2 *>(gbase + (size_t)g * SHF_TBYTES);
    unsigned char *wbase = lds_raw + shf_lds_table() + SHF_GROUPS * shf_lds_per_group() + (size_t)wave * shf_lds_per_wave();
    double *crl = reinterpret_cast<double *>(wbase) + g * SHF_WPAD;
    double2 *dump = reinterpret_cast<double2 *>(wbase + (size_t)SHF_GA * SHF_WPAD * 8) + lane;     // this lane's slot of the wave's dump
    // the zeros in front of the first table, and the slack behind each group's tables (the change of basis reads them with a zero
    // coefficient: they must be numbers)
    if (threadIdx.x < SHF_ZPAD / 8) reinterpret_cast<double *>(lds_raw + shf_lds_table() - SHF_ZPAD)[threadIdx.x] = 0.0;
    if (wq == 0 && lane < 4) reinterpret_cast<double *>(gbase + (size_t)SHF_GA * SHF_TBYTES)[lane] = 0.0;
    const double pi_over_rc = p.por_list;
    const double two_over_rcp = p.two_over_rcp;

    // ---- this lane's atom.  Everything whose address follows from the list entry alone is requested at once: the count, the
    //      lane's two candidate neighbours, its moments of the wave's columns, and from the coefficient row the W_l of its table
    //      positions, c_m and P(1), which the network pass leaves there.
    ShfAtom at;
    at.ii = (xcd_block() * SHF_GROUPS + grp) * SHF_GA + g;
#ifdef ANNP_SHF_STAMPS
    unsigned long long *stamp_row = nullptr;
    if (at.ii + wq < p.inum) stamp_row = reinterpret_cast<unsigned long long *>(p.G + (size_t)((at.ii & ~3) + wq) * ANNP_GPAD);
    if (stamp_row && lane == 0) { stamp_row[15] = __builtin_amdgcn_s_getreg(0xF804); stamp_row[14] = blockIdx.x; stamp_row[13] = __builtin_amdgcn_s_getreg(0xF814); }
#endif
    SHF_STAMP(0);
    const bool exists = at.ii < p.inum;
    const size_t iic = (size_t)min(at.ii, p.inum - 1);
    at.i = p.ilist ? p.ilist[iic] : (int)iic;
    at.n = p.ncount[iic];
    const double *cf = p.coef + iic * ANNP_CPAD;
    const int a0 = l + SHF_GL * SHF_CC * wq;                           // this lane's first neighbour slot in this wave
    int jn[SHF_CC];
#pragma unroll
    for (int u = 0; u < SHF_CC; u++) jn[u] = p.nbrs[iic * SH_CAP_MAX + a0 + SHF_GL * u];
    double2 am[SHF_NSTAGE];
    {
        const double2 *Am = reinterpret_cast<const double2 *>(p.A + iic * SH_MPAD) + l;
        switch (wq) {           // (uniform)
        case 0: ShfBuild<0, 0>::load(Am, l, am); break;
        case 1: ShfBuild<1, 0>::load(Am, l, am); break;
        case 2: ShfBuild<2, 0>::load(Am, l, am); break;
        default: ShfBuild<3, 0>::load(Am, l, am); break;
        }
    }
    double am_w[2];
    am_w[0] = cf[NP + NT + (NT - 1) - l];                              // W_l of table position p = lane: l = 18 - p
    am_w[1] = cf[NP + NT + max(2 - l, 0)];                             // ... of p = 16 + lane (lanes 0..2)
    const double c_l = cf[l < NP ? l : NP + 2 * NT];                   // c_m, m = lane; lanes >= 9: P(1)
    at.xi = p.x[3 * (size_t)at.i]; at.yi = p.x[3 * (size_t)at.i + 1]; at.zi = p.x[3 * (size_t)at.i + 2];
    if (!exists) at.n = 0;
    if (p.type && !type_mapped(p.active, p.type[at.i])) at.n = 0;
    if (at.n > p.n_cap) {            // no moments for this atom: the pair loop takes it (annp_fe_force_fixup)
        if (l == 0 && wq == 0) {
            const int k = p.ovf_list ? atomicAdd(p.ovf_count, 1) : p.ovf_cap;
            if (k < p.ovf_cap) p.ovf_list[k] = at.ii;
            else atomicMax(p.errflag, at.n);
        }
        at.n = 0;
    }
    const int nmax = max(max(__builtin_amdgcn_readlane(at.n, 0), __builtin_amdgcn_readlane(at.n, 16)),
                         max(__builtin_amdgcn_readlane(at.n, 32), __builtin_amdgcn_readlane(at.n, 48)));
    const int C = (nmax + SHF_GL - 1) / SHF_GL;                        // neighbours per lane (uniform), 0 .. 8
    const int cc = min(SHF_CC, max(0, C - SHF_CC * wq));               // ... of which in this wave
    SHF_STAMP(1);
    // ---- positions of the lane's neighbours (a slot without one reads the centre's own)
    double dx[SHF_CC], dy[SHF_CC], dz[SHF_CC];
#pragma unroll
    for (int u = 0; u < SHF_CC; u++) {
        if (!(a0 + SHF_GL * u < at.n)) jn[u] = at.i;
#ifdef ANNP_SHF_CHECK
        if ((unsigned)jn[u] >= (unsigned)p.chk_nall) { atomicMax(p.errflag, 2000000 + (blockIdx.x & 0xffff)); jn[u] = 0; }
#endif
        dx[u] = at.xi - p.x[3 * (size_t)jn[u]]; dy[u] = at.yi - p.x[3 * (size_t)jn[u] + 1]; dz[u] = at.zi - p.x[3 * (size_t)jn[u] + 2];
    }
    if (l <= NP) crl[l] = c_l;                                         // [0..8] c_m, [9] P(1)
    // ---- the wave's columns of the table: B = W kappa A, then the change of basis in place
    const unsigned tb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)T;
    const unsigned dumpb = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)dump;
    {
        // (a group without a single neighbour stages zeros: its tables are what the next group's change of basis reads below its own)
        if (at.n == 0) {
#pragma unroll
            for (int k = 0; k < SHF_NSTAGE; k++) am[k] = make_double2(0.0, 0.0);
        }
        switch (wq) {
        case 0: ShfBuild<0, 0>::stage(T, dump, l, am_w, am); break;
        case 1: ShfBuild<1, 0>::stage(T, dump, l, am_w, am); break;
        case 2: ShfBuild<2, 0>::stage(T, dump, l, am_w, am); break;
        default: ShfBuild<3, 0>::stage(T, dump, l, am_w, am); break;
        }
    }
    SHF_STAMP(2);
    __syncthreads();                                         // every column of every table is staged (and every wave has read who is who)
    tab.clear(threadIdx.x);
    if (nmax > 0) {
        switch (wq) {
        case 0: shf_convert_wave<0>(tb, dumpb, l); break;
        case 1: shf_convert_wave<1>(tb, dumpb, l); break;
        case 2: shf_convert_wave<2>(tb, dumpb, l); break;
        default: shf_convert_wave<3>(tb, dumpb, l); break;
        }
    }
    SHF_STAMP(3);
    __syncthreads();                                         // the tables are complete, the force table is clear
    SHF_STAMP(4);
    if (cc > 0) {
        at.pone = crl[NP];
        double fi[3] = {0.0, 0.0, 0.0}, vs[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        bool has_nbr[SHF_CC];
#pragma unroll
        for (int u = 0; u < SHF_CC; u++) has_nbr[u] = a0 + SHF_GL * u < at.n;
        if (cc == 1) shf_turn<NP, 1, VIRIAL>(p, at, has_nbr, jn, dx, dy, dz, crl, pi_over_rc, two_over_rcp, tb, tab, fi, vs);
        else shf_turn<NP, 2, VIRIAL>(p, at, has_nbr, jn, dx, dy, dz, crl, pi_over_rc, two_over_rcp, tb, tab, fi, vs);
        // more than 128 neighbours (a compressed cell; never bcc Fe at its own density): slots 128 + 16 wq + l, one more turn of one
        // neighbour per lane for the waves at places 0 .. SHF_XWAVES-1 -- a turn with its loads exposed, for the systems that need it,
        // instead of the pair-loop kernels for everybody (round 6: the cliff at 128 neighbours moved to 160)
        if (nmax > SHF_SLOTS && wq < SHF_XWAVES) {
            const int a2 = SHF_SLOTS + SHF_GL * wq + l;
            bool has2[SHF_CC];
            int jn2[SHF_CC];
            double dx2[SHF_CC], dy2[SHF_CC], dz2[SHF_CC];
#pragma unroll
            for (int u = 0; u < SHF_CC; u++) { has2[u] = false; jn2[u] = at.i; dx2[u] = dy2[u] = dz2[u] = 0.0; }
            has2[0] = a2 < at.n;
            if (has2[0]) jn2[0] = p.nbrs[iic * SH_CAP_MAX + a2];
#ifdef ANNP_SHF_CHECK
            if ((unsigned)jn2[0] >= (unsigned)p.chk_nall) { atomicMax(p.errflag, 2000000 + (blockIdx.x & 0xffff)); jn2[0] = 0; }
#endif
            dx2[0] = at.xi - p.x[3 * (size_t)jn2[0]]; dy2[0] = at.yi - p.x[3 * (size_t)jn2[0] + 1]; dz2[0] = at.zi - p.x[3 * (size_t)jn2[0] + 2];
            shf_turn<NP, 1, VIRIAL>(p, at, has2, jn2, dx2, dy2, dz2, crl, pi_over_rc, two_over_rcp, tb, tab, fi, vs);
        }
        SHF_STAMP(5);
        // ---- the centre's share: sums over the atom's 16 lanes end up in the row's last lane
#pragma unroll
        for (int k = 0; k < 3; k++) fi[k] = row16_sum_to_last(fi[k]);
        if (l == 15 && at.n > 0) tab.add(at.i, fi[0], fi[1], fi[2]);
        if (VIRIAL) {
#pragma unroll
            for (int k = 0; k < 6; k++) vs[k] = row16_sum_to_last(vs[k]);
            if (l == 15 && at.n > 0) {
                if (p.virial) {
                    double *va = tab.vacc();                    // (the workgroup's sums: they leave behind the last barrier)
#pragma unroll
                    for (int k = 0; k < 6; k++) atomicAdd(&va[k], vs[k]);
                }
                if (p.vatom) {
                    double *vi = p.vatom + 6 * (size_t)at.i;
#pragma unroll
                    for (int k = 0; k < 6; k++) atomicAdd(vi + k, 0.5 * vs[k]);
                }
            }
        }
    }
    SHF_STAMP(6);
    __syncthreads();
    SHF_STAMP(7);
#ifndef ANNP_SHF_SKIP_FLUSH
    tab.flush(threadIdx.x, p.tab_spills);          // the workgroup's table, in memory order
#endif
    if (VIRIAL && p.virial && threadIdx.x < 6) atomicAdd(&virial_row(p.virial)[threadIdx.x], tab.vacc()[threadIdx.x]);     // (annp_common.hpp: the global virial)
    SHF_STAMP(8);
}

}  // namespace annp
