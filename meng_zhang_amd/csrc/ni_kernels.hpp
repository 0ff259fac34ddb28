// ni_kernels.hpp -- Behler-Parrinello G2/G4 descriptor and chain-rule force kernels
// for the Ni potential of pair_style annp (atomic units inside, as the reference).
//
// Arithmetic restated from annp-gpu-lammps/ni/src/pair_annp.cpp ("ni:" below):
//   radial  G2   ni:686-711     angular G4   ni:713-767     force assembly ni:180-203
// with the derivative of the r_jk term selectable between the literal CPU file
// (compat: ni:737-738 multiply dr_djk by rik_m) and the gradient-consistent form
// used by the reference's own GPU kernel (ni/lib/lal_annp.cu:409-414).
//
// The reference loops over every list entry (≈224 for an 8.5 A list) although only
// r*CFLENGTH < Rc ones (≈18 in fcc Ni) contribute.  With so few survivors a whole
// wave per atom idles most of its lanes and, worse, leaves every dependent memory
// round trip (header -> list row -> coordinates) exposed.  So one wave works on FOUR
// atoms: the four list rows are filtered as one flat candidate stream (all 64 lanes,
// four 64-candidate groups in flight), and from then on each atom owns a 16-lane
// group: its n(n-1)/2 in-range pairs are dealt over those 16 lanes (153 pairs ->
// 10 trips, 96 % of the lane slots used; round-robin enumeration, see NiWalk), per-lane partial sums are private to the
// atom, and the closing reductions are over 16 lanes.
// Roles follow list order (j before k) because compat mode is not symmetric in j,k.
//
// Round 3, what travels between the passes and what stands behind them:
//   * the descriptor pass leaves, per atom, the indices of its in-range neighbours (nbr) AND its in-range (j,k) pairs
//     (pairs: record slots a | b << 8); the force pass rebuilds its records from the former and reads the latter instead
//     of running its own distance pre-pass over all n (n - 1) / 2 candidates;
//   * a group of four atoms whose records do not fit n_cap is appended to a device queue by the descriptor pass and
//     evaluated by a second small launch of each pass (fix = 1: waves walk the queue, records for a whole list row), so an
//     evaluation is complete whatever happened to the configuration since the capacity was learned;
//   * the sincos / exp polynomial coefficients come from constant memory through the scalar unit (annp_common.hpp:
//     sincos_0_pi_s, exp_neg_s); the LDS table only holds the constants that depend on the potential (K).
#pragma once
#include "annp_common.hpp"

namespace annp {

#define ANNP_CFLENGTH 1.889726    // ni/src/pair_annp.h:69
#define ANNP_CFFORCE 51.422515    // ni/src/pair_annp.h:70

constexpr int NI_GA = 4;          // atoms per wave
constexpr int NI_GL = 16;         // lanes per atom
constexpr int NI_MAXP = 8;        // radial functions supported
constexpr int NI_MAXT = 32;       // angular functions supported
constexpr int NI_RED = 9;         // sums per LDS reduction round (desc)
constexpr int NI_REDROW = 17;     // padded row of 16 lane partials
// trips per chunk of the pair pre-pass and the in-range pair list per atom that goes with it (ushort: a | b << 8):
// 16 trips = 256 slots in the descriptor pass; 8 in the force pass, whose LDS decides its occupancy (4 x 12.4 KB
// + tables = three workgroups per CU; with 16 trips and 24-record capacity it was two)
__host__ __device__ constexpr int ni_ch(bool force) { return force ? 8 : 16; }
__host__ __device__ constexpr int ni_plist(bool force) { return ni_ch(force) * 16; }
// Force pass: a wave takes NI_RUN consecutive groups of four atoms (16 atoms: four fcc cells of a row reference 304 atoms,
// ~80 of them distinct) and sends their force contributions through a wave-private table keyed by atom index, flushed
// with one global atomic per distinct atom and component at the end of the run (same scheme as annp_anna_adp).
#ifndef NI_RUN_GROUPS
#define NI_RUN_GROUPS 4
#endif
constexpr int NI_RUN = NI_RUN_GROUPS;
#ifndef NI_TSLOTS_N
#define NI_TSLOTS_N 128
#endif
constexpr int NI_TSLOTS = NI_TSLOTS_N;          // a power of two
constexpr int NI_TSHIFT = 32 - (NI_TSLOTS == 128 ? 7 : NI_TSLOTS == 64 ? 6 : NI_TSLOTS == 32 ? 5 : NI_TSLOTS == 256 ? 8 : 4);
constexpr int NI_CAP_FIXED = 20;     // record capacity compiled into the steady-state instantiation of the force pass
constexpr int NI_XSTAGE = 4096 + 128;   // bytes of the force pass's landing area for positions fetched ahead (ni_preload)
#ifndef NI_TPROBE_N
#define NI_TPROBE_N 8
#endif
constexpr int NI_TPROBE = NI_TPROBE_N;
#ifndef NI_WAVES_PER_SIMD
#define NI_WAVES_PER_SIMD 4     // descriptor pass: 512 / 4 = 128 VGPRs
#endif
#ifndef NI_KEEP_RECORDS
#define NI_KEEP_RECORDS 1
#endif
#ifndef NI_FORCE_WAVES_PER_SIMD
#define NI_FORCE_WAVES_PER_SIMD 3
#endif

struct NiArgs {
    int inum, n_cap;            // n_cap: in-range neighbours held per atom (LDS records), host-sized
    const int *ilist;
    const double *x;
    const int *type;            // nullable [nall], with `active` as in FeArgs
    unsigned active;
    const int *numneigh;
    const long long *first;
    const int *neigh;
    int npsf, ntsf, compat;
    const double *sym;          // see "per-function tables" below
    const int *isym;
    double rc_rad, rc_ang;      // Bohr
    double lam[4], eta[4];      // product shape (NiShape): the distinct lambda and eta values in visit order -- kernel arguments, i.e. scalar
                                // registers: every use of one used to be a broadcast read of an LDS table, a quarter of the pair loops' LDS traffic
    double rc2a;                // (rc_ang / CFLENGTH)^2 (1 + 1e-12): the pre-pass's bound on r_jk^2, in A^2
    double por_rad, por_ang;    // pi / rc_rad, pi / rc_ang, divided on the host (a wave-uniform double division is a dozen vector instructions and
                                // a register pair that lives across the whole kernel)
    unsigned long long rad_em;  // byte m: eta_m / eta_0 of radial function m when that is a small integer (then
                                // exp(-eta_m r^2) = exp(-eta_0 r^2)^k: one exp per neighbour instead of npsf), else 0
    double *G;
    const double *coef;         // [inum][ANNP_CPAD]: c_k = dE/dGhat_k / (sf_max-sf_min)_k
    double *f;
    double *virial;
    double *vatom;              // nullable, [nall][6] accumulated (VIRIAL variant)
    int *ncount;                // [inum] in-range neighbours found by the descriptor pass
    int *nbr;                   // [inum][nbr_stride] their indices, in list order (written by pass 1, read by pass 3)
    int nbr_stride;
    // in-range (j,k) pairs of every atom, left by the descriptor pass for the force pass (nullable: the force pass then
    // finds them again with its own pre-pass): entry = a | b << 8, record slots in list order, a < b
    unsigned short *pairs;      // [inum][pstride]
    int pstride;                // >= n_cap (n_cap - 1) / 2 of the descriptor pass
    int *npair;                 // [inum]
    // Groups of four atoms whose records do not fit n_cap are not lost: the descriptor pass appends the group (its first
    // list slot) to a queue, and a second, small launch of each pass (`fix` = 1: one wave per queue entry, n_cap large enough
    // for a whole list row, its own neighbour rows indexed by queue slot) evaluates them -- the same arrangement as the
    // Chebyshev force pass's fix-up launch.  errflag is raised only when that cannot be done (queue full, or more in-range
    // neighbours than the largest records LDS can hold).
    int *ovf_count;             // queue length (device word); null: no queue, overflow is an error
    int *ovf_list;              // [ovf_cap] first list slot ii0 of each queued group
    int ovf_cap;
    int fix;                    // this launch walks the queue
    int skip_above;             // main force launch: groups with more in-range neighbours than this are the queue's (= n_cap of the main descriptor launch)
    int *errflag;
};

// per-wave LDS: records of NI_GA atoms (7 doubles + index; the force pass adds 3 accumulators and the
// atoms' coefficient rows), or the reduction scratch of the descriptor pass, whichever is larger
// (the force pass's coefficient rows arrive by LDS-DMA, 256 B = 32 doubles per row -- nsf <= ANNP_GPAD = 32 is checked at init: a
// stride of 34 doubles keeps the rows 16-byte aligned and the four atoms' copies of a weight in four different banks)
constexpr int NI_CSTRIDE = 34;
static_assert(ANNP_GPAD <= 32, "coefficient rows of the Behler force pass");
__host__ __device__ inline int ni_coef_stride(int) { return NI_CSTRIDE; }
__host__ __device__ inline size_t ni_lds_per_wave(int cap, bool force, int nsf, bool gpairs = false)
{
    const size_t R = (size_t)NI_GA * cap + 2;          // + two dummy records for idle lanes
    // (force pass: the seven record arrays double as the landing area of ni_preload, NI_XSTAGE bytes)
    size_t b = (force ? (R * 7 * 8 > (size_t)NI_XSTAGE ? R * 7 * 8 : (size_t)NI_XSTAGE) : R * 7 * 8) + (force ? R * 3 * 8 + (size_t)NI_GA * ni_coef_stride(nsf) * 8 + (size_t)NI_TSLOTS * (3 * 8 + 4) + R * 4 + NI_GA * 4 : 0) + R * 4 + NI_GA * 4 +
               ((force && gpairs) ? 0 : (size_t)NI_GA * ni_plist(force) * 2);     // (pair lists read from memory need no room here)
    const size_t scratch = (size_t)NI_GA * NI_RED * NI_REDROW * 8;
    if (!force && b < scratch) b = scratch;
    return (b + 15) / 16 * 16;
}

// Pair enumeration of one atom over its 16 lanes, without decoding a flat index: item `it` is row
// a = it mod n with partner (a + t) mod n, t = it / n + 1 = 1 .. n/2 (for even n the last offset only for
// a < n/2, which is where the item count n(n-1)/2 ends) -- every unordered pair exactly once.  A lane's
// items are l, l+16, l+32, ...: the step (16 mod n, 16 / n) is constant, so advancing costs a few adds.
struct NiWalk { int a, t, da, dt, n; };
__device__ __forceinline__ NiWalk ni_walk_init(int l, int n)
{
    NiWalk w;
    w.n = max(n, 1);
    w.a = l % w.n; w.t = l / w.n + 1;
    w.da = NI_GL % w.n; w.dt = NI_GL / w.n;
    return w;
}
// current pair in list order (lo < hi), then step to this lane's next item
__device__ __forceinline__ void ni_walk_next(NiWalk &w, int &lo, int &hi)
{
    int b = w.a + w.t;
    if (b >= w.n) b -= w.n;
    lo = min(w.a, b); hi = max(w.a, b);
    w.a += w.da; w.t += w.dt;
    if (w.a >= w.n) { w.a -= w.n; w.t += 1; }
}

// ---- per-function tables, prepared on the host (annp_hip_init) --------------------------
// The ntsf angular functions are visited in the order (lambda, eta, zeta); pos -> original
// index is `perm`.  Layout of NiArgs::sym (doubles):
//   rad   [npsf][3]   eta, Rs, Rc
//   ang   [ntsf][4]   eta, lambda, zeta, Rc        (original order, as parsed)
//   sorted[ntsf][4]   eta, lambda, zeta, pref = 2^(1-zeta)   (visit order)
//   etas  [NI_MAXE]   distinct eta values (visit order of first appearance)
// and of NiArgs::isym (ints):
//   perm[ntsf], eidx[ntsf] (which distinct eta), zint[ntsf] (zeta if a small non-negative integer, else -1),
//   ne, emult[NI_MAXE] (eta_e = emult_e * eta_0 when that holds exactly, else 0)
constexpr int NI_MAXE = 4;

struct NiTab {
    const double *T;            // LDS math table (annp_common.hpp)
    const double *rad, *sorted, *etas;
    const int *perm, *eidx, *zint, *emult;
    int ne;
};
__device__ __forceinline__ NiTab ni_tab(const double *sym, const int *isym, int npsf, int ntsf, const double *T)
{
    NiTab t;
    t.T = T;
    t.rad = sym;
    t.sorted = sym + 3 * npsf + 4 * ntsf;
    t.etas = t.sorted + 4 * ntsf;
    t.perm = isym; t.eidx = isym + ntsf; t.zint = isym + 2 * ntsf;
    t.ne = isym[3 * ntsf];
    t.emult = isym + 3 * ntsf + 1;
    return t;
}

// u^e for a wave-uniform non-negative integer e from the squaring ladder U[k] = u^(2^k), k = 0..4,
// U0 = u^0 with the lane's validity folded in (0 where 1 + lambda cos <= 0, ni:744-747)
__device__ __forceinline__ double ni_ladder_pow(const double (&U)[5], double U0, int e)
{
    double r = U0;
    if (e & 1) r *= U[0];
    if (e & 2) r *= U[1];
    if (e & 4) r *= U[2];
    if (e & 8) r *= U[3];
    if (e & 16) r *= U[4];
    return r;
}

// x^k, k a small wave-uniform positive integer
__device__ __forceinline__ double ni_powi(double x, int k)
{
    double r = 1.0, b = x;
    while (k) { if (k & 1) r *= b; b *= b; k >>= 1; }
    return r;
}

// Visit the angular functions of one (j,k) pair:  body(pos, val, dval) with
//   val  = 2^(1-zeta) (1+lambda cos)^zeta      exp(-eta r2sum)      (term_cot term_exp, ni:748-750)
//   dval = 2^(1-zeta) zeta (1+lambda cos)^(zeta-1) exp(-eta r2sum) lambda   (d val / d cos)
// both 0 where 1 + lambda cos <= 0.
template <int NT, bool DERIV, typename Body>
__device__ __forceinline__ void ni_visit_functions(const NiTab &t, int ntsf, double ct, double r2sum, Body &&body)
{
    // exp(-eta r2sum) for the distinct etas: one exp, powers of it where eta_e is a multiple of eta_0
    double E[NI_MAXE];
    E[0] = exp_neg_s(-t.etas[0] * r2sum);
#pragma unroll
    for (int e = 1; e < NI_MAXE; e++) {
        E[e] = 0.0;
        if (e < t.ne) E[e] = (t.emult[e] > 0) ? ni_powi(E[0], t.emult[e]) : exp_neg_s(-t.etas[e] * r2sum);
    }
    double lam_prev = 0.0, U0 = 0.0;
    double U[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int pos = 0; pos < NT; pos++) {
        if (pos < ntsf) {
            const double lam = t.sorted[4 * pos + 1], zeta = t.sorted[4 * pos + 2], pref = t.sorted[4 * pos + 3];
            if (pos == 0 || lam != lam_prev) {               // wave-uniform: new squaring ladder
                const double u = fma(lam, ct, 1.0);
                const bool ok = u > 0.0;
                U0 = ok ? 1.0 : 0.0;
                U[0] = ok ? u : 0.0;
                U[1] = U[0] * U[0]; U[2] = U[1] * U[1]; U[3] = U[2] * U[2]; U[4] = U[3] * U[3];
                lam_prev = lam;
            }
            const int e = t.eidx[pos];
            const double Ee = (e == 0) ? E[0] : (e == 1) ? E[1] : (e == 2) ? E[2] : E[3];
            const int zi = t.zint[pos];
            // zeta is a small non-negative integer (checked at init: the unrolled visit must stay
            // compact enough for the instruction cache, a general pow() per function does not)
            const double pw = ni_ladder_pow(U, U0, zi);
            double pw1 = 0.0;
            if (DERIV) pw1 = (zi >= 1) ? ni_ladder_pow(U, U0, zi - 1) : 0.0;
            const double pe = pref * Ee;
            body(pos, pe * pw, DERIV ? pe * zeta * lam * pw1 : 0.0);
        }
    }
}

// ---- constants of the pair loops ------------------------------------------------------------------
// A wave-uniform double that lives across the pair loop wants a scalar register pair.  Rounds 2-4 kept such constants in an LDS table
// next to a copy of the math table (annp_common.hpp) and read each where it was used: a 64-lane broadcast per use, ~20 of the ~75
// values a trip of the force pass's pair loop moves through the CU's one LDS pipe -- which that loop loads as heavily as it loads the
// vector pipe (round 5).  Now: lambda, eta and the cutoff are kernel arguments (NiArgs::lam, eta, rc_ang, por_ang: scalar registers,
// re-read from the argument segment when they do not fit), CFLENGTH is a literal, and the factors 2^(1-zeta), zeta 2^(1-zeta) of the
// product shape are applied with v_ldexp_f64 / a literal (the zetas are template arguments).  What is left in LDS is the generic
// shape's copy of the sorted per-function table:
//   S[...]           copy of the sorted per-function table + etas (generic shape)
constexpr int NI_TABLE_DOUBLES = 4 * NI_MAXT + NI_MAXE;

// compile-time exponents of the product-shape kernels: byte z of ZP = zeta_z, byte e of EM = eta_e / eta_0
#define NI_BYTE(packed, k) ((int)(((packed) >> (8 * (k))) & 255u))

template <int NL, int NE, int NZ>
__device__ __forceinline__ void ni_tables_fill(double *lds, const NiArgs &p, NiTab &t, int lane)
{
    if constexpr (NL == 0) {
        double *S = lds;
        for (int idx = lane; idx < 4 * p.ntsf + NI_MAXE; idx += 64) S[idx] = t.sorted[idx];
        t.sorted = S; t.etas = S + 4 * p.ntsf;      // the generic visit reads the LDS copy
    }
}

// 2^(1-zeta) x, and zeta 2^(1-zeta) x, for an integer zeta that is known once the visit is unrolled (a byte of the template argument ZP):
// exact scalings (ni:748: term_coe = 2^(1-zeta)); a zeta that is a power of two goes into the exponent
__device__ __forceinline__ double ni_pref(double x, int zeta) { return __builtin_ldexp(x, 1 - zeta); }
__device__ __forceinline__ double ni_dpref(double x, int zeta)
{
    if ((zeta & (zeta - 1)) == 0) {
        int k = 0;
        for (int v = zeta; v > 1; v >>= 1) k++;
        return __builtin_ldexp(x, 1 - zeta + k);
    }
    return __builtin_ldexp(x, 1 - zeta) * (double)zeta;
}

// exp(-eta_e r2sum) for the distinct etas: one exp, integer powers of it where eta_e is a multiple of eta_0
template <int NE, unsigned EM>
__device__ __forceinline__ void ni_exps(const NiArgs &p, double r2sum, double (&E)[NE])
{
    E[0] = exp_neg_s(-p.eta[0] * r2sum);
#pragma unroll
    for (int e = 1; e < NE; e++) E[e] = ni_powi(E[0], NI_BYTE(EM, e));
}

// Descriptor-pass visit for the product shape {lambda} x {eta} x {zeta} (the shipped Ni potential: 2 x 3 x 4),
// visit position = (l * NE + e) * NZ + z as in the sorted order: per function one FMA, per (l,z) the power.
//   ga[pos] += 2^(1-zeta) (1+lambda cos)^zeta exp(-eta r2sum) * tfc
template <int NL, int NE, int NZ, unsigned ZP, unsigned EM, int NT>
__device__ __forceinline__ void ni_desc_cart(const NiArgs &p, double ct, double r2sum, double tfc, double (&ga)[NT])
{
    double E[NE];
    ni_exps<NE, EM>(p, r2sum, E);
#pragma unroll
    for (int l = 0; l < NL; l++) {
        const double u = fma(p.lam[l], ct, 1.0);
        const bool ok = u > 0.0;                          // ni:744-747
        const double U0 = ok ? tfc : 0.0;
        double U[5];
        U[0] = ok ? u : 0.0;
        U[1] = U[0] * U[0]; U[2] = U[1] * U[1]; U[3] = U[2] * U[2]; U[4] = U[3] * U[3];
#pragma unroll
        for (int z = 0; z < NZ; z++) {
            const double pwt = ni_pref(ni_ladder_pow(U, U0, NI_BYTE(ZP, z)), NI_BYTE(ZP, z));
#pragma unroll
            for (int e = 0; e < NE; e++) ga[(l * NE + e) * NZ + z] = fma(pwt, E[e], ga[(l * NE + e) * NZ + z]);
        }
    }
}

// Force-pass visit for the product shape.  The three sums the force needs factor over eta:
//   A3 = sum c val      = sum_e E_e B_e,      B_e = sum_lz c_lez pw_lz     (pw = 2^(1-zeta) (1+lambda cos)^zeta)
//   A2 = sum c eta val  = sum_e eta_e E_e B_e
//   A1 = sum c dval     = sum_e E_e D_e,      D_e = sum_l lambda_l sum_z c_lez dw_lz     (lambda dw = d pw / d cos)
// so the visit itself is a polynomial in (1 + lambda cos) with the atom's weights -- 6 FMAs per (l,z) --
// and the exponentials are only needed afterwards.
// cw: this atom's weights in visit order (LDS, read as a 16-lane broadcast).
template <int NL, int NE, int NZ, unsigned ZP, unsigned EM>
__device__ __forceinline__ void ni_force_cart(const NiArgs &p, const double *cw, double ct, double r2sum,
                                              double &A1, double &A2, double &A3)
{
    double B[NE], D[NE];
#pragma unroll
    for (int e = 0; e < NE; e++) { B[e] = 0.0; D[e] = 0.0; }
    // Software-pipelined by hand: the weights of step s+1 are read while step s computes, and a
    // scheduling fence closes each step.  Left alone the compiler issues all LDS reads of the visit (and of
    // the exponential after it) at the top and holds ~110 registers of operands.
    double cc[NE];
    auto fetch = [&](int l, int z) {
#pragma unroll
        for (int e = 0; e < NE; e++) cc[e] = cw[(l * NE + e) * NZ + z];
    };
    fetch(0, 0);
#pragma unroll
    for (int l = 0; l < NL; l++) {
        const double u = fma(p.lam[l], ct, 1.0);
        const bool ok = u > 0.0;
        const double U0 = ok ? 1.0 : 0.0;
        double U[5];
        U[0] = ok ? u : 0.0;
        U[1] = U[0] * U[0]; U[2] = U[1] * U[1]; U[3] = U[2] * U[2]; U[4] = U[3] * U[3];
        double Dl[NE];
#pragma unroll
        for (int e = 0; e < NE; e++) Dl[e] = 0.0;
#pragma unroll
        for (int z = 0; z < NZ; z++) {
            double c0[NE];
#pragma unroll
            for (int e = 0; e < NE; e++) c0[e] = cc[e];
            const double pw = ni_pref(ni_ladder_pow(U, U0, NI_BYTE(ZP, z)), NI_BYTE(ZP, z));
            const double dw = NI_BYTE(ZP, z) >= 1 ? ni_dpref(ni_ladder_pow(U, U0, NI_BYTE(ZP, z) - 1), NI_BYTE(ZP, z)) : 0.0;
            if (z + 1 < NZ) fetch(l, z + 1);
            else if (l + 1 < NL) fetch(l + 1, 0);
#pragma unroll
            for (int e = 0; e < NE; e++) {
                B[e] = fma(c0[e], pw, B[e]);
                Dl[e] = fma(c0[e], dw, Dl[e]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int e = 0; e < NE; e++) D[e] = fma(p.lam[l], Dl[e], D[e]);
    }
    double E[NE];
    ni_exps<NE, EM>(p, r2sum, E);
#pragma unroll
    for (int e = 0; e < NE; e++) {
        A3 = fma(E[e], B[e], A3);
        A2 = fma(p.eta[e] * E[e], B[e], A2);
        A1 = fma(E[e], D[e], A1);
    }
}

// ---- LDS records of one wave --------------------------------------------------------------
struct NiLds {
    double *dx, *dy, *dz, *r, *rinv, *fc, *dfc;   // [NI_GA * cap] each, atom g at g*cap
    double *a0, *a1, *a2;                         // force accumulators (force pass)
    double *coef;                                 // [NI_GA][stride] weights: radial, then angular in visit order
    double *tacc;                                 // [NI_TSLOTS][3] force table of the run (force pass)
    int *tkey;                                    // [NI_TSLOTS]
    int *j;                                       // [NI_GA * cap]
    int *ci;                                      // [NI_GA] atom index of each group, -1 = none
    int *sl, *cs;                                 // force pass: slot of the run's force table that takes neighbour s / centre g (-1: straight to memory)
    unsigned short *pl;                           // [NI_GA][ni_plist] in-range pairs of the current chunk
};

template <bool FORCE>
__device__ __forceinline__ NiLds ni_carve(unsigned char *wbase, int cap, int cstride)
{
    NiLds L;
    const int R = NI_GA * cap + 2;
    double *d = reinterpret_cast<double *>(wbase);
    L.dx = d; L.dy = L.dx + R; L.dz = L.dy + R; L.r = L.dz + R; L.rinv = L.r + R; L.fc = L.rinv + R; L.dfc = L.fc + R;
    d = L.dfc + R;
    if (FORCE && 7 * R * 8 < NI_XSTAGE) d = reinterpret_cast<double *>(wbase + NI_XSTAGE);
    L.a0 = L.a1 = L.a2 = L.coef = L.tacc = nullptr;
    L.tkey = nullptr; L.sl = L.cs = nullptr;
    if (FORCE) { L.a0 = d; L.a1 = L.a0 + R; L.a2 = L.a1 + R; L.coef = L.a2 + R; L.tacc = L.coef + NI_GA * cstride; d = L.tacc + 3 * NI_TSLOTS; }
    L.j = reinterpret_cast<int *>(d);
    L.ci = L.j + R;
    int *after = L.ci + NI_GA;
    if (FORCE) { L.tkey = after; after += NI_TSLOTS; L.sl = after; after += R; L.cs = after; after += NI_GA; }
    L.pl = reinterpret_cast<unsigned short *>(after);
    return L;
}

// Filter the list rows of atoms ii0 .. ii0+3 into the records.  Returns the largest in-range count of the
// four (uniform; > p.n_cap means the records overflowed and must not be used); nl = count of this lane's atom.
// headers of a group: lane g < 4 fetches atom g (asked for before the kernel's tables are built, so that the two round trips overlap)
struct NiHead { int hi, hjn; long long hbase; double hx, hy, hz; };
__device__ __forceinline__ NiHead ni_stage_head(const NiArgs &p, int ii0, int lane)
{
    NiHead h;
    h.hi = -1; h.hjn = 0; h.hbase = 0; h.hx = h.hy = h.hz = 0.0;
    if (lane < NI_GA) {
        const int ii = ii0 + lane;
        if (ii < p.inum) {
            h.hi = p.ilist ? p.ilist[ii] : ii;
            h.hjn = p.numneigh[h.hi];
            if (p.type && !type_mapped(p.active, p.type[h.hi])) h.hjn = 0;      // centre of an unmapped type: nothing in range
            h.hbase = p.first[h.hi];
            h.hx = p.x[3 * (size_t)h.hi]; h.hy = p.x[3 * (size_t)h.hi + 1]; h.hz = p.x[3 * (size_t)h.hi + 2];
        }
    }
    return h;
}

template <bool FORCE>
__device__ __forceinline__ int ni_stage(const NiArgs &p, const NiHead &head, const NiLds &L, int cap, int lane, int &nl)
{
    const int hi = head.hi, hjn = head.hjn;
    const long long hbase = head.hbase;
    const double hx = head.hx, hy = head.hy, hz = head.hz;
    if (lane < NI_GA) L.ci[lane] = hi;
    const double rcmax = fmax(p.rc_rad, p.rc_ang);
    const double rc2 = (rcmax / ANNP_CFLENGTH) * (rcmax / ANNP_CFLENGTH) * (1.0 + 1e-12);   // coarse filter in A^2
    const double pi_over_rc = p.por_ang;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int g = lane >> 4, l = lane & 15;
    // first sweep, one row at a time (everything about the row is wave-uniform): cheap distance filter, survivors
    // compacted in list order.  Four 64-candidate groups per trip keep the index loads, then the coordinate
    // gathers, in flight together (only ~18 of ~224 candidates survive: this sweep is memory latency).
    int nmax = 0;
    nl = 0;
    // Short rows (the library's own list: ~75 entries): the index loads of all four rows go out together, then all the
    // coordinate gathers -- two dependent memory round trips for the wave instead of two per row.
    const int jn_all = max(max(__builtin_amdgcn_readlane(hjn, 0), __builtin_amdgcn_readlane(hjn, 1)),
                           max(__builtin_amdgcn_readlane(hjn, 2), __builtin_amdgcn_readlane(hjn, 3)));
    if (jn_all <= 128) {
        int j[NI_GA][2];
        bool valid[NI_GA][2];
#pragma unroll
        for (int ga = 0; ga < NI_GA; ga++) {
            const int jn = __builtin_amdgcn_readlane(hjn, ga);
            const unsigned blo = (unsigned)__builtin_amdgcn_readlane((int)(hbase & 0xffffffffll), ga);
            const int bhi = __builtin_amdgcn_readlane((int)(hbase >> 32), ga);
            // (loads unconditional, clamped to the row's last entry: a load under `valid ? .. : ..` becomes a branch with its own
            // wait, a round trip through memory each; an empty row reads numneigh[0] and drops it)
            const int *row = jn > 0 ? p.neigh + (((long long)bhi << 32) | (long long)blo) : p.numneigh;
            const int last = max(jn, 1) - 1;
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int jj = 64 * u + lane;
                valid[ga][u] = jj < jn;
                const int jr = row[min(jj, last)] & ANNP_NEIGHMASK;
                j[ga][u] = valid[ga][u] ? jr : 0;
            }
        }
        if (p.type) {
#pragma unroll
            for (int ga = 0; ga < NI_GA; ga++)
#pragma unroll
                for (int u = 0; u < 2; u++) { const int tj = p.type[j[ga][u]]; valid[ga][u] = valid[ga][u] & type_mapped(p.active, tj); }
        }
        double qx[NI_GA][2], qy[NI_GA][2], qz[NI_GA][2];
#pragma unroll
        for (int ga = 0; ga < NI_GA; ga++)
#pragma unroll
            for (int u = 0; u < 2; u++) {
                qx[ga][u] = p.x[3 * (size_t)j[ga][u]]; qy[ga][u] = p.x[3 * (size_t)j[ga][u] + 1]; qz[ga][u] = p.x[3 * (size_t)j[ga][u] + 2];
            }
#pragma unroll
        for (int ga = 0; ga < NI_GA; ga++) {
            const double xi = readlane_f64(hx, ga), yi = readlane_f64(hy, ga), zi = readlane_f64(hz, ga);
            const int sb = ga * cap;
            int n = 0;
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const double dx = xi - qx[ga][u], dy = yi - qy[ga][u], dz = zi - qz[ga][u];
                const double rsq = dx * dx + dy * dy + dz * dz;
                const bool in = valid[ga][u] && rsq < rc2 && rsq > 0.0;
                const unsigned long long m = __ballot(in);
                const int pos = n + __popcll(m & lt);
                if (in && pos < cap) {
                    const int s = sb + pos;
                    L.dx[s] = dx; L.dy[s] = dy; L.dz[s] = dz; L.r[s] = rsq; L.j[s] = j[ga][u];
                }
                n += __popcll(m);
            }
            n = uniform(n);
            if (g == ga) nl = n;
            nmax = max(nmax, n);
        }
    } else
    for (int ga = 0; ga < NI_GA; ga++) {
        const int jn = __builtin_amdgcn_readlane(hjn, ga);
        const unsigned blo = (unsigned)__builtin_amdgcn_readlane((int)(hbase & 0xffffffffll), ga);
        const int bhi = __builtin_amdgcn_readlane((int)(hbase >> 32), ga);
        const int *row = p.neigh + (((long long)bhi << 32) | (long long)blo);
        const double xi = readlane_f64(hx, ga), yi = readlane_f64(hy, ga), zi = readlane_f64(hz, ga);
        const int sb = ga * cap;
        int n = 0;
        for (int c0 = 0; c0 < jn; c0 += 256) {
            // groups of 64 candidates this trip really has (wave-uniform: jn sits in a scalar register).  A row of the
            // library's own list (~75 entries, annp_hip_list_cutoff) fills two; the other two are branched over.
            const int ngr = min(4, (jn - c0 + 63) >> 6);
            int j[4];
            bool valid[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                valid[u] = false; j[u] = 0;
                if (u < ngr) {
                    const int jj = c0 + 64 * u + lane;
                    valid[u] = jj < jn;
                    const int jr = row[min(jj, jn - 1)] & ANNP_NEIGHMASK;
                    j[u] = valid[u] ? jr : 0;
                }
            }
            if (p.type) {
#pragma unroll
                for (int u = 0; u < 4; u++) { const int tj = p.type[j[u]]; valid[u] = valid[u] & type_mapped(p.active, tj); }      // (unconditional load)
            }
            double dx[4], dy[4], dz[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                dx[u] = dy[u] = dz[u] = 0.0;
                if (u < ngr) { dx[u] = xi - p.x[3 * (size_t)j[u]]; dy[u] = yi - p.x[3 * (size_t)j[u] + 1]; dz[u] = zi - p.x[3 * (size_t)j[u] + 2]; }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (u >= ngr) continue;
                const double rsq = dx[u] * dx[u] + dy[u] * dy[u] + dz[u] * dz[u];
                const bool in = valid[u] && rsq < rc2 && rsq > 0.0;
                const unsigned long long m = __ballot(in);
                const int pos = n + __popcll(m & lt);
                if (in && pos < cap) {
                    const int s = sb + pos;
                    L.dx[s] = dx[u]; L.dy[s] = dy[u]; L.dz[s] = dz[u]; L.r[s] = rsq; L.j[s] = j[u];
                }
                n += __popcll(m);
            }
        }
        n = uniform(n);
        if (g == ga) nl = n;
        nmax = max(nmax, n);
    }
    if (nmax > cap) return nmax;
    if (lane < 2) {     // dummy records NI_GA*cap, +1: finite geometry, beyond every cutoff, never accumulated into
        const int s = NI_GA * cap + lane;
        L.dx[s] = lane == 0 ? 1.0 : 0.0; L.dy[s] = lane == 0 ? 0.0 : 1.0; L.dz[s] = 0.0;
        L.r[s] = 1e3; L.rinv[s] = 1e-3; L.fc[s] = 0.0; L.dfc[s] = 0.0;   // (exp_neg_tab wants a sane argument)
    }
    wave_lds_sync();
    // second sweep: the exact test of the reference (r * CFLENGTH < Rc, ni:693/729) and the per-neighbour terms.
    // Entries that fail it keep fc = 0 and r = huge, so every pair they enter is rejected by ni_pair.
    // A lane's neighbours l and l + 16 side by side (shared coefficient loads, interleaved chains), the rest of a long row in a loop.
    auto second = [&](int a0, int a1) {
        const bool t0 = a0 < nl, t1 = a1 < nl;
        // (a lane without a neighbour reads the dummy record -- written above, finite -- never a record of its atom: an atom with NO
        // in-range neighbour has none that anybody wrote, and LDS keeps what the previous dispatch left there)
        const int s0 = t0 ? g * cap + a0 : NI_GA * cap, s1 = t1 ? g * cap + a1 : NI_GA * cap;
        const double q0 = L.r[s0], q1 = L.r[s1];
        const double i0 = fast_rsqrt_ic(q0), i1 = fast_rsqrt_ic(q1);
        const double r0 = q0 * i0, r1 = q1 * i1;
        const double m0 = r0 * ANNP_CFLENGTH, m1 = r1 * ANNP_CFLENGTH;
        double sn0, cs0, sn1, cs1;
        sincos_0_pi_s2(pi_over_rc * fmin(m0, p.rc_ang), pi_over_rc * fmin(m1, p.rc_ang), sn0, cs0, sn1, cs1);
        const bool in0 = m0 < p.rc_ang, in1 = m1 < p.rc_ang;
        if (t0) { L.r[s0] = r0; L.rinv[s0] = i0; L.fc[s0] = in0 ? 0.5 * (cs0 + 1.0) : 0.0; L.dfc[s0] = in0 ? (sn0 * -0.5) * pi_over_rc : 0.0; }
        if (t1) { L.r[s1] = r1; L.rinv[s1] = i1; L.fc[s1] = in1 ? 0.5 * (cs1 + 1.0) : 0.0; L.dfc[s1] = in1 ? (sn1 * -0.5) * pi_over_rc : 0.0; }
        // a neighbour outside the angular cutoff (a radial-only one, or one the coarse filter let through by a hair) is moved 10 000 A
        // away for the pre-pass, which then needs no test of r_ij, r_ik: any pair with it fails the test of r_jk (two of them may pass it:
        // ni_pair's exact test, on L.r, rejects the pair).  Nothing else reads dx of such a record.
        if (!FORCE) {
            if (t0 && !in0) L.dx[s0] = 1e4;
            if (t1 && !in1) L.dx[s1] = 1e4;
        }
        if (FORCE) {
            if (t0) { L.a0[s0] = 0.0; L.a1[s0] = 0.0; L.a2[s0] = 0.0; }
            if (t1) { L.a0[s1] = 0.0; L.a1[s1] = 0.0; L.a2[s1] = 0.0; }
        }
    };
    for (int a = l; a < nmax; a += 2 * NI_GL) second(a, a + NI_GL);
    return nmax;
}

// What the force pass needs from memory before it can stage a group, requested ahead: the group's headers (lanes 0..3) and the
// first two entries of this lane's neighbour row.  A wave handles NI_RUN groups one after the other, and each used to start
// with three dependent memory round trips (header -> list row -> positions) with nothing of its own to overlap them; the
// next group's first two are now in flight behind the current group's pair loop.
// (Five registers, which is what the pass has left: the centre's position is fetched with the neighbours' at staging time --
// its index is known by then, so that is the same round trip, not another one.)
struct NiAhead { int hi, hn, hp, j0, j1; };
__device__ __forceinline__ void ni_request(const NiArgs &p, int ii0, int row0, int lane, NiAhead &q)
{
    q.hi = -1; q.hn = 0; q.hp = 0;
    if (lane < NI_GA) {
        const int ii = ii0 + lane;
        if (ii < p.inum) {
            q.hi = p.ilist ? p.ilist[ii] : ii;
            q.hn = p.ncount[ii];
            if (p.npair) q.hp = p.npair[ii];
        }
    }
    const int g = lane >> 4, l = lane & 15;
    const int *row = p.nbr + (size_t)(row0 + g) * p.nbr_stride;
    const bool there = ii0 + g < p.inum;        // (entries beyond the atom's count are whatever the row holds: never used)
    q.j0 = (there && l < p.nbr_stride) ? row[l] : 0;
    q.j1 = (there && l + NI_GL < p.nbr_stride) ? row[l + NI_GL] : 0;
}

// ---- force pass: the run's force table, and what a group needs from memory, fetched a group ahead ----------------------------
// Timing builds (profiles/r05_ni_timing_builds.txt) showed the force pass spending 0.50 of its 0.82 ms OUTSIDE the pair loop, on a fifth of
// its instructions: each group of a wave's run began with dependent round trips (permutation -> coefficient row; header -> row ->
// positions) and ended with a chain of LDS round trips (one ds_cmpst per probe of the force table and neighbour, two passes over ~18
// neighbours on 16 lanes), at three waves per SIMD.  Now
//  * the coefficient rows arrive in visit order (annp_hip_init permutes the rows of the network's output map: no index load, no
//    division by nsf) and are fetched -- with the centre's and the first two neighbours' positions per lane -- for the NEXT group
//    right after the current group's pair loop, so the loads fly during its epilogue;
//  * the force-table slot of every neighbour is found then as well, while those loads are in flight (all the probes of a lane in one
//    loop: three ds_cmpst per trip instead of three loops), and kept in the record: the epilogue adds to a known slot;
//  * staging and epilogue treat a lane's two neighbours (l and l + 16) side by side in straight-line code instead of two trips
//    of a loop, so their sincos / exp chains interleave.
// Key j's slot in the wave-private table, claimed if new; -1 after NI_TPROBE occupied slots (the contribution then goes straight to memory).
// Up to three keys per lane are resolved together (want_k false: no key).
__device__ __forceinline__ void ni_table_claim3(int *tkey, bool w0, int j0, bool w1, int j1, bool w2, int j2, int &r0, int &r1, int &r2)
{
    unsigned a0 = ((unsigned)j0 * 0x9E3779B1u) >> NI_TSHIFT, a1 = ((unsigned)j1 * 0x9E3779B1u) >> NI_TSHIFT, a2 = ((unsigned)j2 * 0x9E3779B1u) >> NI_TSHIFT;
    r0 = r1 = r2 = -1;
#pragma unroll 1
    for (int probe = 0; probe < NI_TPROBE; probe++) {
        if (!__any(w0 || w1 || w2)) break;
        int o0 = 0, o1 = 0, o2 = 0;
        if (w0) o0 = atomicCAS(&tkey[a0], -1, j0);
        if (w1) o1 = atomicCAS(&tkey[a1], -1, j1);
        if (w2) o2 = atomicCAS(&tkey[a2], -1, j2);
        if (w0 && (o0 == -1 || o0 == j0)) { r0 = (int)a0; w0 = false; }
        if (w1 && (o1 == -1 || o1 == j1)) { r1 = (int)a1; w1 = false; }
        if (w2 && (o2 == -1 || o2 == j2)) { r2 = (int)a2; w2 = false; }
        a0 = (a0 + 1) & (NI_TSLOTS - 1); a1 = (a1 + 1) & (NI_TSLOTS - 1); a2 = (a2 + 1) & (NI_TSLOTS - 1);
    }
}

// add to a slot of the table, or to memory
__device__ __forceinline__ void ni_table_add(const NiLds &L, double *f, int slot, int j, double fx, double fy, double fz)
{
    if (slot >= 0) { atomicAdd(&L.tacc[3 * slot], fx); atomicAdd(&L.tacc[3 * slot + 1], fy); atomicAdd(&L.tacc[3 * slot + 2], fz); }
    else { atomicAdd(&f[3 * (size_t)j], fx); atomicAdd(&f[3 * (size_t)j + 1], fy); atomicAdd(&f[3 * (size_t)j + 2], fz); }
}

typedef const __attribute__((address_space(1))) unsigned char *ni_gbp;
typedef __attribute__((address_space(3))) unsigned char *ni_lbp;
// The coefficient row of atom GA of a group (radial, then angular in visit order -- as it lies in memory) from memory into LDS: 16 lanes
// x 16 bytes, one instruction per atom so that the rows land NI_CSTRIDE apart.  The rows' 16 GA bytes of extra stride go into the
// instruction's immediate offset, which counts on both sides -- hence the source moved back by as much -- and the LDS base is the same
// for all four: with a base per atom the compiler merges the four branches into one instruction whose base differs from lane to
// lane, and the base is one scalar register, M0.
template <int GA>
__device__ __forceinline__ void ni_coef_row(const NiArgs &p, int ii0, const NiLds &L, int g, int l)
{
    if (g == GA) {
        const ni_gbp src = (ni_gbp)(p.coef + (size_t)min(ii0 + GA, p.inum - 1) * ANNP_CPAD + 2 * l);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src - GA * (NI_CSTRIDE * 8 - 256)),
                                         (__attribute__((address_space(3))) void *)(ni_lbp)(unsigned char *)L.coef, 16, GA * (NI_CSTRIDE * 8 - 256), 0);
    }
}

// Positions, coefficient rows and table slots of the group whose headers `q` holds (ni_request: landed by now).  Nothing of it is
// held in registers: 25 of them across the epilogue is what the compiler answered with scratch stores right behind each load
// -- five exposed round trips instead of none.  The data goes from memory into LDS by itself (global_load_lds), the coefficient rows
// to their place (dead since the pair loop ended: the radial weights were used when the records were made), the positions into the
// record arrays dx..dfc, dead as well (the epilogue reads a0..a2, j and sl; the VIRIAL variant reads dx, dy, dz too and therefore
// runs its epilogue first).  A position is 24 bytes; it comes as two overlapping 16-byte pieces per lane, (x, y) and (y, z); piece k of
// every lane lies at 1024 k + 16 lane.
//   [0, 4096)  neighbours l (pieces 0, 1) and l + 16 (pieces 2, 3)      [4096, 4224)  the four centres, two pieces of 64 bytes
// limit: groups with more neighbours than this are not this launch's (their rows in p.nbr were never written: nothing of them is an index)
__device__ __forceinline__ void ni_preload(const NiArgs &p, const NiAhead &q, int ii0, int limit, int cstride, const NiLds &L, int lane, int &s0, int &s1, int &sc)
{
    const int g = lane >> 4, l = lane & 15;
    int nmax = 0;
#pragma unroll
    for (int ga = 0; ga < NI_GA; ga++) nmax = max(nmax, __builtin_amdgcn_readlane(q.hn, ga));
    const int nlg = nmax > limit ? 0 : __shfl(q.hn, g, 64);
    // (every lane loads: what a row holds beyond its atom's count is not an index -- atom 0 is read instead, as for an absent centre)
    const size_t j0 = l < nlg ? (size_t)q.j0 : 0, j1 = l + NI_GL < nlg ? (size_t)q.j1 : 0;
    const ni_lbp stage = (ni_lbp)(unsigned char *)L.dx;
    const ni_gbp x0 = (ni_gbp)(p.x + 3 * j0), x1 = (ni_gbp)(p.x + 3 * j1);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)x0, (__attribute__((address_space(3))) void *)stage, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(x0 + 8), (__attribute__((address_space(3))) void *)(stage + 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)x1, (__attribute__((address_space(3))) void *)(stage + 2048), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(x1 + 8), (__attribute__((address_space(3))) void *)(stage + 3072), 16, 0, 0);
    if (lane < NI_GA) {
        const ni_gbp xc = (ni_gbp)(p.x + 3 * (size_t)max(q.hi, 0));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)xc, (__attribute__((address_space(3))) void *)(stage + 4096), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(xc + 8), (__attribute__((address_space(3))) void *)(stage + 4096 + 64), 16, 0, 0);
    }
    // the coefficient rows (radial, then angular in visit order -- as they lie in memory): 16 lanes x 16 bytes per atom, one
    // instruction per atom so that the rows land cstride apart
    ni_coef_row<0>(p, ii0, L, g, l); ni_coef_row<1>(p, ii0, L, g, l); ni_coef_row<2>(p, ii0, L, g, l); ni_coef_row<3>(p, ii0, L, g, l);
    ni_table_claim3(L.tkey, l < nlg, q.j0, l + NI_GL < nlg, q.j1, lane < NI_GA && q.hi >= 0, q.hi, s0, s1, sc);
}

// W neighbours' records from their distance vectors, side by side (W = 2: a lane's neighbours l and l + 16; the coefficient loads of
// the sincos / exp evaluations are shared and their chains interleave).  The force accumulators start from the radial part (ni:693-709:
// -sum_m c_m d/dr [exp(-eta_m r^2) fc(r)] along -x_ij / r_ij) instead of zero: the epilogue that used to add it -- a sincos and an exp per
// neighbour at the end of the group -- is then three reads and three adds.  cr: the radial weights of this lane's atom (LDS, written
// before this is called).  When the radial and the angular functions share their cutoff (the shipped potential) its function is
// evaluated once.
template <int NP, int NL, unsigned EM, int W>
__device__ __forceinline__ void ni_records(const NiArgs &p, const NiLds &L, const double *srad, const double *cr, const int (&s)[W], const bool (&there)[W],
                                           const double (&dx)[W], const double (&dy)[W], const double (&dz)[W], const int (&j)[W], const int (&slot)[W])
{
    double rinv[W], r[W], rm[W], fc[W], dfc[W], fcr[W], dfcr[W], rmc[W], e0[W];
#pragma unroll
    for (int w = 0; w < W; w++) {
        const double rsq = dx[w] * dx[w] + dy[w] * dy[w] + dz[w] * dz[w];
        rinv[w] = fast_rsqrt_ic(rsq);
        r[w] = rsq * rinv[w];
        rm[w] = r[w] * ANNP_CFLENGTH;
        rmc[w] = fmin(rm[w], p.rc_rad);
    }
    auto cut = [&](double rc, double por, double (&f)[W], double (&df)[W]) {      // (outside the cutoff the argument is pi: the values are dropped)
        double sn[W], cs[W];
        if constexpr (W == 2) sincos_0_pi_s2(por * fmin(rm[0], rc), por * fmin(rm[1], rc), sn[0], cs[0], sn[1], cs[1]);
        else sincos_0_pi_s(por * fmin(rm[0], rc), sn[0], cs[0]);
#pragma unroll
        for (int w = 0; w < W; w++) {
            const bool in = rm[w] < rc;
            f[w] = in ? 0.5 * (cs[w] + 1.0) : 0.0;
            df[w] = in ? (sn[w] * -0.5) * por : 0.0;
        }
    };
    cut(p.rc_ang, p.por_ang, fc, dfc);
    if (p.rc_rad != p.rc_ang) cut(p.rc_rad, p.por_rad, fcr, dfcr);
    else {
#pragma unroll
        for (int w = 0; w < W; w++) { fcr[w] = fc[w]; dfcr[w] = dfc[w]; }
    }
    const double eta0 = srad[0];
    if constexpr (W == 2) exp_neg_s2(-eta0 * rmc[0] * rmc[0], -eta0 * rmc[1] * rmc[1], e0[0], e0[1]);
    else e0[0] = exp_neg_s(-eta0 * rmc[0] * rmc[0]);
    double R[W];
#pragma unroll
    for (int w = 0; w < W; w++) R[w] = 0.0;
#pragma unroll
    for (int m = 0; m < NP; m++)
        if (m < p.npsf) {
            const double eta = srad[3 * m], c = cr[m];
            const int km = NL > 0 ? NI_BYTE(EM, m & 3) : (int)((p.rad_em >> (8 * m)) & 255ull);
#pragma unroll
            for (int w = 0; w < W; w++) {
                const double em = km > 0 ? ni_powi(e0[w], km) : exp_neg_s(-eta * rmc[w] * rmc[w]);
                R[w] = fma(c, em * (-fcr[w] * 2.0 * eta * rmc[w] + dfcr[w]), R[w]);
            }
        }
#pragma unroll
    for (int w = 0; w < W; w++) {
        const double sc = -R[w] * rinv[w];              // dr_dj = -xij/rij   (fcr = dfcr = 0 outside the radial cutoff)
        if (there[w]) {
            const int q = s[w];
            L.dx[q] = dx[w]; L.dy[q] = dy[w]; L.dz[q] = dz[w]; L.j[q] = j[w]; L.sl[q] = slot[w];
            L.r[q] = r[w]; L.rinv[q] = rinv[w]; L.fc[q] = fc[w]; L.dfc[q] = dfc[w];
            L.a0[q] = sc * dx[w]; L.a1[q] = sc * dy[w]; L.a2[q] = sc * dz[w];
        }
    }
}

// Force pass: rebuild the records from the compact lists the descriptor pass left (same entries, same order) and from what
// ni_preload sent into LDS.
template <int NP, int NL, unsigned EM>
__device__ __forceinline__ int ni_stage_compact(const NiArgs &p, const NiAhead &q, int s0, int s1, int sc, int row0, int cap, int nsf, int cstride, const double *srad,
                                                const NiLds &L, int lane, int &nl, int &npairs)
{
    const int hi = q.hi, hn = q.hn, hp = q.hp;
    if (lane < NI_GA) { L.ci[lane] = hi; L.cs[lane] = sc; }
    const int g = lane >> 4, l = lane & 15;
    nl = __shfl(hn, g, 64);
    npairs = __shfl(hp, g, 64);
    int nmax = 0;
#pragma unroll
    for (int ga = 0; ga < NI_GA; ga++) nmax = max(nmax, __builtin_amdgcn_readlane(hn, ga));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // what ni_preload sent has arrived
    if (nmax > cap) return nmax;
    // positions out of the staging area (which is the record arrays: every lane reads before any lane writes)
    const double2 *sw = reinterpret_cast<const double2 *>(L.dx);
    const double2 cxy = sw[256 + g], cyz = sw[260 + g];
    const double2 p0 = sw[lane], q0 = sw[64 + lane], p1 = sw[128 + lane], q1 = sw[192 + lane];
    const double cx = cxy.x, cy = cxy.y, cz = cyz.y;
    const double dx0 = cx - p0.x, dy0 = cy - p0.y, dz0 = cz - q0.y;
    const double dx1 = cx - p1.x, dy1 = cy - p1.y, dz1 = cz - q1.y;
    wave_lds_sync();
    if (lane < 2) {     // dummy records, as in ni_stage (values made from the lane number: a double constant would be built in a
                        // register pair, hoisted out of the run and spilled)
        const int s = NI_GA * cap + lane;
        const double far = (double)(1000 + lane);
        L.dx[s] = lane == 0 ? 1.0 : 0.0; L.dy[s] = lane == 0 ? 0.0 : 1.0; L.dz[s] = 0.0;
        L.r[s] = far; L.rinv[s] = __builtin_amdgcn_rcp(far); L.fc[s] = 0.0; L.dfc[s] = 0.0;
    }
    const double *cr = L.coef + g * cstride;
    if (nmax > NI_GL)
        ni_records<NP, NL, EM, 2>(p, L, srad, cr, {g * cap + l, g * cap + l + NI_GL}, {l < nl, l + NI_GL < nl}, {dx0, dx1}, {dy0, dy1}, {dz0, dz1}, {q.j0, q.j1}, {s0, s1});
    else
        ni_records<NP, NL, EM, 1>(p, L, srad, cr, {g * cap + l}, {l < nl}, {dx0}, {dy0}, {dz0}, {q.j0}, {s0});
    if (nmax > 2 * NI_GL) {         // rows longer than the two entries per lane that were fetched ahead (dense systems, the fix-up launch)
        const int *row = p.nbr + (size_t)(row0 + g) * p.nbr_stride;       // row0 = ii0, or 4 x the queue slot in a fix-up launch
        for (int a = l + 2 * NI_GL; a < nmax; a += NI_GL) {
            const bool there = a < nl;
            const int j = there ? row[a] : 0;
            int slot, u1, u2;
            ni_table_claim3(L.tkey, there, j, false, 0, false, 0, slot, u1, u2);
            ni_records<NP, NL, EM, 1>(p, L, srad, cr, {g * cap + min(a, cap - 1)}, {there}, {cx - p.x[3 * (size_t)j]}, {cy - p.x[3 * (size_t)j + 1]}, {cz - p.x[3 * (size_t)j + 2]}, {j}, {slot});
        }
    }
    return nmax;
}

// What the function visit needs of one (j,k) pair around the centre: cos(theta), the three distances (Bohr)
// and the product of the cutoff functions.  The force pass re-reads the records for the vectors afterwards
// (LDS reads are cheap; holding them across the 24-function visit costs the registers of a whole wave slot).
struct NiPairS {
    double ct, rjm, rkm, rgm, ig;   // ig = 1/r_jk (A^-1)
    double fcjk, dfcjk, tfc;
    double xj[3], xk[3], ij, ik, fcj, fck;      // what the pair's records held (the force pass keeps them across the visit: since round 5 it has the registers,
                                                // and its pair loop loads the LDS pipe as heavily as the vector pipe)
    bool ok;
};

// sa, sb: record slots (idle lanes pass the two dummy records: ok = false, everything finite)
__device__ __forceinline__ NiPairS ni_pair(const NiLds &L, const NiArgs &p, int sa, int sb)
{
    NiPairS q;
    const double rj = L.r[sa], rk = L.r[sb];
    const double xj0 = L.dx[sa], xj1 = L.dy[sa], xj2 = L.dz[sa];
    const double xk0 = L.dx[sb], xk1 = L.dy[sb], xk2 = L.dz[sb];
    const double ij = L.rinv[sa], ik = L.rinv[sb];
    // xjk = x_j - x_k = xik - xij
    const double g0 = xk0 - xj0, g1 = xk1 - xj1, g2 = xk2 - xj2;
    const double gsq = g0 * g0 + g1 * g1 + g2 * g2;
    q.ig = fast_rsqrt_ic(gsq);
    q.ct = ((xj0 * ij) * (xk0 * ik) + (xj1 * ij) * (xk1 * ik)) + (xj2 * ij) * (xk2 * ik);
    const double cfl = ANNP_CFLENGTH;
    q.rjm = rj * cfl; q.rkm = rk * cfl; q.rgm = (gsq * q.ig) * cfl;
    const double rc = p.rc_ang;
    q.ok = (q.rjm < rc) && (q.rkm < rc) && (q.rgm < rc);   // ni:729
    q.fcjk = 0.0; q.dfcjk = 0.0; q.tfc = 0.0;
    q.xj[0] = xj0; q.xj[1] = xj1; q.xj[2] = xj2; q.xk[0] = xk0; q.xk[1] = xk1; q.xk[2] = xk2; q.ij = ij; q.ik = ik;
    q.fcj = 0.0; q.fck = 0.0;
    if (q.ok) {
        double sn, cs;
        const double por = p.por_ang;
        sincos_0_pi_s(por * q.rgm, sn, cs);
        q.fcjk = 0.5 * (cs + 1.0);
        q.dfcjk = -0.5 * por * sn;
        q.fcj = L.fc[sa]; q.fck = L.fc[sb];
        q.tfc = q.fcj * q.fck * q.fcjk;
    }
    return q;
}

// compiler-level only: values read from LDS before this point are not kept in registers across it
__device__ __forceinline__ void ni_forget_lds() { asm volatile("" ::: "memory"); }

// Pair pre-pass.  Only pairs with r_ij, r_ik AND r_jk inside the cutoff contribute (ni:729) -- 60 of the 153 pairs of
// an fcc-Ni atom -- but which ones is scattered over the enumeration, so every trip of the full visit would carry
// ~60 % idle lanes.  A cheap sweep over a chunk of trips (distances only) writes the in-range pairs of each atom as
// a dense list; the expensive part (cutoff function of r_jk, exponential, 24 functions) then runs over ceil(60/16)
// = 4 trips instead of 10.  The visit applies the exact test again (ni_pair's `ok`).
// Returns this lane's atom's count for the chunk (same value in the 16 lanes of a group).
// RTEST: the records may hold neighbours outside the angular cutoff at their true place (the force pass's own records; the
// descriptor pass's ni_stage moves them away instead)
template <bool RTEST>
__device__ __forceinline__ int ni_prepass(const NiLds &L, const NiArgs &p, NiWalk &walk, int g, int l, int sbase, int cap,
                                          int npl, int t0, int t1, const int plist)
{
    // r_jk by its square against a bound a hair above the cutoff: the pre-pass may let a pair through that the visit's own
    // test (ni_pair: the reference's r * CFLENGTH < Rc) then rejects, never the other way round
    const double cfl = ANNP_CFLENGTH, rc = p.rc_ang, rc2a = p.rc2a;
    int cnt = 0;
    // two trips at a time: the sixteen record reads of both go out together, then the two tests (a trip by itself is a chain of
    // LDS round trip -> a dozen dependent operations -> ballot -> write: ten of them in a row were a quarter of the descriptor pass)
    for (int t = t0; t < t1; t += 2) {
        int a[2], b[2], sa[2], sb[2];
        bool live[2];
        double ax[2], ay[2], az[2], ar[2], bx[2], by[2], bz[2], br[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int pp = (t + u) * NI_GL + l;
            ni_walk_next(walk, a[u], b[u]);
            live[u] = (t + u < t1) && pp < npl;
            sa[u] = live[u] ? sbase + a[u] : NI_GA * cap; sb[u] = live[u] ? sbase + b[u] : NI_GA * cap + 1;
            ax[u] = L.dx[sa[u]]; ay[u] = L.dy[sa[u]]; az[u] = L.dz[sa[u]]; ar[u] = RTEST ? L.r[sa[u]] : 0.0;
            bx[u] = L.dx[sb[u]]; by[u] = L.dy[sb[u]]; bz[u] = L.dz[sb[u]]; br[u] = RTEST ? L.r[sb[u]] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const double g0 = bx[u] - ax[u], g1 = by[u] - ay[u], g2 = bz[u] - az[u];
            const double gsq = g0 * g0 + g1 * g1 + g2 * g2;
            const bool ok = live[u] & (!RTEST || ((ar[u] * cfl < rc) & (br[u] * cfl < rc))) & (gsq < rc2a);
            const unsigned long long m = __ballot(ok);
            const unsigned m16 = (unsigned)(m >> (NI_GL * g)) & 0xffffu;
            if (ok) L.pl[g * plist + cnt + __popc(m16 & ((1u << l) - 1u))] = (unsigned short)(a[u] | (b[u] << 8));
            cnt += __popc(m16);
        }
    }
    return cnt;
}

// developer timing builds (tools/ni_stamps.py): -DANNP_NI_STAMPS writes s_memtime at a few points of a wave's life into the descriptor
// row of the first atom of its run (the descriptor buffer is dead by then); no stamp is compiled into the library
#ifdef ANNP_NI_STAMPS
#define NI_STAMP(k) do { if (!p.fix && lane == 0) reinterpret_cast<unsigned long long *>(p.G + (size_t)run * NI_RUN * NI_GA * ANNP_GPAD)[k] = __builtin_amdgcn_s_memtime(); } while (0)
#define NI_DSTAMP(k) do { dstamp[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define NI_STAMP(k) do { } while (0)
#define NI_DSTAMP(k) do { } while (0)
#endif

// ---------------------------------------------------------------------------------
// FIX: the fix-up instantiation (its waves walk the queue: a loop around the body, bounded to 2 waves per SIMD so that what the
// loop keeps live does not spill; it runs on the few groups that outgrew their records, if any)
// CAP > 0: the record capacity is compiled in (p.n_cap equals it), as in the force pass: record arrays at constant offsets
template <int NP, int NT, int NL, int NE, int NZ, unsigned ZP, unsigned EM, bool FIX, int CAP = 0>
// (the table-driven instantiation keeps NI_MAXP + NI_MAXT = 40 accumulators per lane: 128 VGPRs would spill 39 of them)
__global__ __launch_bounds__(256, FIX ? 2 : (NL > 0 ? NI_WAVES_PER_SIMD : 3)) void annp_ni_desc(NiArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    ANNP_POISON();
#ifdef ANNP_NI_STAMPS
    unsigned long long dstamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    NI_DSTAMP(0);
    const int lane0 = lane_id();
    const int wave = uniform(threadIdx.x >> 6);
    const int nsf = p.npsf + p.ntsf;
    const int cap = CAP > 0 ? CAP : p.n_cap;
    // main launch: the group's headers are asked for first, the tables below are built while they fly
    NiHead head;
    if (!FIX) head = ni_stage_head(p, uniform((xcd_block() * ANNP_WAVES_PER_BLOCK + wave) * NI_GA), lane0);
    NiTab tab = ni_tab(p.sym, p.isym, p.npsf, p.ntsf, nullptr);
    const double *srad = tab.rad;
    ni_tables_fill<NL, NE, NZ>(reinterpret_cast<double *>(lds_raw), p, tab, lane0);
    unsigned char *wbase = lds_raw + NI_TABLE_DOUBLES * 8 + (size_t)wave * ni_lds_per_wave(cap, false, nsf);
    const NiLds L = ni_carve<false>(wbase, cap, 0);
    double *scratch = reinterpret_cast<double *>(wbase);
    // main launch: one group of four atoms per wave.  Fix-up launch: a small grid whose waves walk the queue.
    const int nslots = FIX ? min(*p.ovf_count, p.ovf_cap) : 0;
    for (int it = 0; FIX || it < 1; it++) {         // (main launch: exactly one trip, no loop is generated)
    // (the lane number is opaque per trip, as in the force pass: per-lane LDS addresses are formed where they are used instead
    // of being hoisted out of this loop and spilled -- 60 VGPRs otherwise)
    int lane_q = lane0;
    if (FIX) asm volatile("" : "+v"(lane_q));
    const int lane = lane_q, g = lane_q >> 4, l = lane_q & 15;
    int ii0, row0;              // first list slot of the group; first row of its neighbour / pair lists
    if (!FIX) {
        ii0 = row0 = uniform((xcd_block() * ANNP_WAVES_PER_BLOCK + wave) * NI_GA);
        if (ii0 >= p.inum) break;
    } else {
        const int slot = uniform((it * (int)gridDim.x + (int)blockIdx.x) * ANNP_WAVES_PER_BLOCK + wave);
        if (slot >= nslots) break;
        ii0 = uniform(p.ovf_list[slot]);
        row0 = slot * NI_GA;
        wave_lds_sync();        // the records and the reduction scratch of the previous entry are dead
    }
    int nl;
    NI_DSTAMP(1);
    if (FIX) head = ni_stage_head(p, ii0, lane);
    const int nmax = ni_stage<false>(p, head, L, cap, lane, nl);
    NI_DSTAMP(2);
    const int ncl = __shfl(nl, NI_GL * (lane & (NI_GA - 1)), 64);     // count of atom (lane & 3), for lanes 0..3
    // a group whose records overflowed is skipped by both passes: its count is left at 0 so that the force pass, which
    // runs without the host having looked at the error word, finds nothing to do for it (its list rows are not written)
    if (p.ncount && lane < NI_GA && ii0 + lane < p.inum) p.ncount[ii0 + lane] = nmax > cap ? 0 : ncl;
    if (p.npair && nmax > cap && lane < NI_GA && ii0 + lane < p.inum) p.npair[ii0 + lane] = 0;
    if (nmax > cap) {
        if (lane == 0) {
            bool queued = false;
            if (!FIX && p.ovf_list) {
                const int q = atomicAdd(p.ovf_count, 1);
                queued = q < p.ovf_cap;
                if (queued) p.ovf_list[q] = ii0;
            }
            if (!queued) atomicMax(p.errflag, nmax);
        }
        for (int idx = lane; idx < NI_GA * ANNP_GPAD; idx += 64)
            if (ii0 + idx / ANNP_GPAD < p.inum) p.G[(size_t)ii0 * ANNP_GPAD + idx] = 0.0;
        continue;
    }
    wave_lds_sync();
    const int sbase = g * cap;
    // the survivors, for the force pass (which then need not filter the full list rows again)
    for (int a = l; a < nl; a += NI_GL) p.nbr[(size_t)(row0 + g) * p.nbr_stride + a] = L.j[sbase + a];

    double gr[NP], ga[NT];
#pragma unroll
    for (int m = 0; m < NP; m++) gr[m] = 0.0;
#pragma unroll
    for (int m = 0; m < NT; m++) ga[m] = 0.0;      // indexed by visit position
    // G2 (ni:686-711): lane l of the group owns neighbours l, l+16, ... (two at a time; where the radial and the angular functions
    // share their cutoff -- the shipped potential -- its function is the one the records already hold)
    const bool same_rc = p.rc_rad == p.rc_ang;
    for (int a = l; a < nmax; a += 2 * NI_GL) {
        const bool t0 = a < nl, t1 = a + NI_GL < nl;
        // A lane with nothing to add reads the DUMMY record (ni_stage: r = 1e3, fc = 0), not record 0 of its atom.  Round 5 read
        // `sbase + 0` there and multiplied the result by f = 0: for an atom with no in-range neighbour at all (nl == 0) in a wave
        // whose other atoms have some (nmax > 0) that record was never written, a large negative leftover of the previous
        // dispatch gives m^2 = +Inf, exp_neg_s(-Inf) = NaN, and NaN x 0 = NaN in the atom's descriptor row and energy
        // (test_ni_tiny_systems[17], red once on a fresh box; `make poison` makes it red every time).  ni:692-709: nothing is
        // added outside the cutoff -- by selection below, not by a zero factor.
        const int s0 = t0 ? sbase + a : NI_GA * cap, s1 = t1 ? sbase + a + NI_GL : NI_GA * cap;
        const double m0 = fmin(L.r[s0] * ANNP_CFLENGTH, p.rc_rad), m1 = fmin(L.r[s1] * ANNP_CFLENGTH, p.rc_rad);
        double f0, f1;
        if (same_rc) { f0 = L.fc[s0]; f1 = L.fc[s1]; }
        else {
            double sn0, cs0, sn1, cs1;
            sincos_0_pi_s2(p.por_rad * m0, p.por_rad * m1, sn0, cs0, sn1, cs1);
            f0 = 0.5 * (cs0 + 1.0); f1 = 0.5 * (cs1 + 1.0);          // (0 at the cutoff itself)
        }
        const bool in0 = t0 && L.r[s0] * ANNP_CFLENGTH < p.rc_rad, in1 = t1 && L.r[s1] * ANNP_CFLENGTH < p.rc_rad;    // ni:693: outside the cutoff nothing is added
        if (!in0) f0 = 0.0;
        if (!in1) f1 = 0.0;
        double e0, e1;
        exp_neg_s2(-srad[0] * m0 * m0, -srad[0] * m1 * m1, e0, e1);
#pragma unroll
        for (int m = 0; m < NP; m++)
            if (m < p.npsf) {
                // (the compiled-in shape has its radial etas in the ratios EM too: ni_is_shipped_shape)
                const int km = NL > 0 ? NI_BYTE(EM, m & 3) : (int)((p.rad_em >> (8 * m)) & 255ull);
                const double v0 = (km > 0 ? ni_powi(e0, km) : exp_neg_s(-srad[3 * m] * m0 * m0)) * f0;
                const double v1 = (km > 0 ? ni_powi(e1, km) : exp_neg_s(-srad[3 * m] * m1 * m1)) * f1;
                gr[m] += in0 ? v0 : 0.0;
                gr[m] += in1 ? v1 : 0.0;
            }
    }
    NI_DSTAMP(3);
    // G4 (ni:713-767): the atom's pairs over its 16 lanes
    const int npl = nl * (nl - 1) / 2;
    const int trips = (nmax * (nmax - 1) / 2 + NI_GL - 1) / NI_GL;
    NiWalk walk = ni_walk_init(l, nl);
    constexpr int CH = ni_ch(false), PLIST = ni_plist(false);
    int poff = 0;               // pairs of this lane's atom written to p.pairs so far
    for (int t0 = 0; t0 < trips; t0 += CH) {
        const int cnt = ni_prepass<false>(L, p, walk, g, l, sbase, cap, npl, t0, min(trips, t0 + CH), PLIST);
        wave_lds_sync();
        NI_DSTAMP(4);
        const int cmax = max(max(__builtin_amdgcn_readlane(cnt, 0), __builtin_amdgcn_readlane(cnt, 16)),
                             max(__builtin_amdgcn_readlane(cnt, 32), __builtin_amdgcn_readlane(cnt, 48)));
        for (int t2 = 0; t2 * NI_GL < cmax; t2++) {
            const int idx = t2 * NI_GL + l;
            const bool live = idx < cnt;
            const int v = live ? L.pl[g * PLIST + idx] : 0;
            const NiPairS q = ni_pair(L, p, live ? sbase + (v & 255) : NI_GA * cap, live ? sbase + (v >> 8) : NI_GA * cap + 1);
            const double r2sum = q.rjm * q.rjm + q.rkm * q.rkm + q.rgm * q.rgm;
            const double tfc = q.tfc;                                   // idle lanes add zeros
            ni_forget_lds();                                            // (keeps the table reads inside the loop)
            if constexpr (NL > 0) ni_desc_cart<NL, NE, NZ, ZP, EM, NT>(p, q.ct, r2sum, tfc, ga);
            else
                ni_visit_functions<NT, false>(tab, p.ntsf, q.ct, r2sum,
                                              [&](int pos, double val, double) { ga[pos] = fma(val, tfc, ga[pos]); });
        }
        if (p.pairs) {          // the chunk's list, for the force pass (an atom has at most n (n - 1) / 2 <= pstride entries in all)
            int gq = g;         // (opaque: the row's address is formed here, once per chunk, not kept in registers across the visit)
            asm volatile("" : "+v"(gq));
            unsigned short *gp = p.pairs + (size_t)(row0 + gq) * p.pstride + poff;
            for (int idx = l; idx < cnt; idx += NI_GL) gp[idx] = L.pl[gq * PLIST + idx];
            poff += cnt;
        }
        wave_lds_sync();        // the list is rewritten by the next chunk
    }
    if (p.npair && l == 0 && ii0 + g < p.inum) p.npair[ii0 + g] = poff;
    wave_lds_sync();
    NI_DSTAMP(5);
    // sum the 16 lane partials of every atom through LDS, NI_RED sums per round (records are dead now)
    constexpr int NS = NP + NT;
#pragma unroll
    for (int c9 = 0; c9 < (NS + NI_RED - 1) / NI_RED; c9++) {
#pragma unroll
        for (int k = 0; k < NI_RED; k++) {
            const int m = c9 * NI_RED + k;       // slot in the padded (NP | NT) layout
            double v = 0.0;
            if (m < NP) v = gr[m < NP ? m : 0];
            else if (m < NS) v = ga[(m - NP) >= 0 && (m - NP) < NT ? (m - NP) : 0];
            scratch[(g * NI_RED + k) * NI_REDROW + l] = v;
        }
        wave_lds_sync();
        if (lane < NI_GA * NI_RED) {
            const int gq = lane / NI_RED, k = lane % NI_RED;
            const double *src = scratch + (gq * NI_RED + k) * NI_REDROW;
            double s = 0.0;
#pragma unroll
            for (int u = 0; u < NI_GL; u++) s += src[u];
            const int m = c9 * NI_RED + k;
            int fidx = -1;                       // padded slot -> function index
            if (m < NP) { if (m < p.npsf) fidx = m; }
            else if (m - NP < p.ntsf) fidx = p.npsf + tab.perm[m - NP];
            if (fidx >= 0 && ii0 + gq < p.inum) p.G[(size_t)(ii0 + gq) * ANNP_GPAD + fidx] = s;
        }
        wave_lds_sync();
    }
    for (int idx = lane; idx < NI_GA * ANNP_GPAD; idx += 64)
        if (idx % ANNP_GPAD >= nsf && ii0 + idx / ANNP_GPAD < p.inum) p.G[(size_t)ii0 * ANNP_GPAD + idx] = 0.0;
#ifdef ANNP_NI_STAMPS
    NI_DSTAMP(6);
    if (!FIX && nsf <= 28 && lane == 0 && ii0 + 1 < p.inum)       // (slots 28..31 of the first two atoms' rows: the network weighs them with zeros)
        for (int k = 0; k < 7; k++) reinterpret_cast<unsigned long long *>(p.G + (size_t)(ii0 + k / 4) * ANNP_GPAD)[28 + k % 4] = dstamp[k];
#endif
    }
}

// ---------------------------------------------------------------------------------
// GPAIRS: the in-range pairs of an atom are read from the list the descriptor pass left in memory (p.pairs) instead of
// being found again by a pre-pass over all n (n - 1) / 2 candidates: 88 of the pass's 730 vector instructions per atom,
// a sixth of its LDS traffic and the 1 KB of LDS per wave that held the chunk's list.
// CAP > 0: the record capacity is compiled in (p.n_cap equals it).  The ten record arrays of a wave then lie at constant offsets from
// one base and a record access is `ds_read_b64 v, v offset:imm`; with the capacity in a register each array keeps its own base in a
// scalar register -- more than there are: two dozen v_readlane of spilled ones and as many address additions per trip of the pair loop.
template <int NP, int NT, int NL, int NE, int NZ, unsigned ZP, unsigned EM, bool VIRIAL, bool GPAIRS, int CAP = 0>
__global__ __launch_bounds__(256, NI_FORCE_WAVES_PER_SIMD) void annp_ni_force(NiArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    ANNP_POISON();
    const int lane = lane_id();
    const int wave = uniform(threadIdx.x >> 6);
    const int run = uniform((p.fix ? (int)blockIdx.x : xcd_block()) * ANNP_WAVES_PER_BLOCK + wave);
    if (!p.fix && run * NI_RUN * NI_GA >= p.inum) return;
    NI_STAMP(0);
    // the first group's headers and list entries are asked for before anything else: the tables below are built while they fly
    NiAhead ahead;
    bool early = !p.fix;
    if (early) ni_request(p, run * NI_RUN * NI_GA, run * NI_RUN * NI_GA, lane, ahead);
    const int nslots = p.fix ? min(*p.ovf_count, p.ovf_cap) : 0;   // fix-up launch: a small grid whose waves walk the queue
    const int nwaves = (int)gridDim.x * ANNP_WAVES_PER_BLOCK;
    const int nsf = p.npsf + p.ntsf;
    const int cap = CAP > 0 ? CAP : p.n_cap;
    const int cstride = ni_coef_stride(nsf);
    NiTab tab = ni_tab(p.sym, p.isym, p.npsf, p.ntsf, nullptr);
    const double *srad = tab.rad;
    ni_tables_fill<NL, NE, NZ>(reinterpret_cast<double *>(lds_raw), p, tab, lane);
    unsigned char *wbase = lds_raw + NI_TABLE_DOUBLES * 8 + (size_t)wave * ni_lds_per_wave(cap, true, nsf, GPAIRS);
    const NiLds L = ni_carve<true>(wbase, cap, cstride);
    for (int sl = lane; sl < NI_TSLOTS; sl += 64) { L.tkey[sl] = -1; L.tacc[3 * sl] = 0.0; L.tacc[3 * sl + 1] = 0.0; L.tacc[3 * sl + 2] = 0.0; }
    wave_lds_sync();
    NI_STAMP(1);
    int as0 = -1, as1 = -1, asc = -1;    // table slots of the coming group's neighbours / centres (ni_preload)
    bool requested = false;     // `ahead` holds the coming group, and what ni_preload sent for it is on its way (main launch, from the second group of the run on)
#pragma unroll 1
    for (int gk = 0;; gk++) {
    int ii0, row0;
    if (!p.fix) {
        if (gk >= NI_RUN) break;
        ii0 = row0 = uniform((run * NI_RUN + gk) * NI_GA);
    } else {
        const int slot = run + gk * nwaves;
        if (slot >= nslots) break;
        ii0 = uniform(p.ovf_list[slot]);
        row0 = slot * NI_GA;
    }
    if (ii0 >= p.inum) break;
    ni_forget_lds();            // nothing read from the LDS tables is carried from one group to the next in registers
    // The lane number is made opaque per group: otherwise every per-lane LDS address of the staging loops below (7 record
    // arrays x a few offsets) is computed once before the run, kept live across the pair loops and spilled to scratch
    // (24 VGPRs, 100 B per lane before this); recomputing them costs a handful of integer adds per group.
    int lane_q = lane;
    asm volatile("" : "+v"(lane_q));
    const int g = lane_q >> 4, l = lane_q & 15;
    int nl, npg;
    const int limit = p.fix ? cap : min(cap, p.skip_above);
    if (!requested) { if (!early) ni_request(p, ii0, row0, lane_q, ahead); ni_preload(p, ahead, ii0, limit, cstride, L, lane_q, as0, as1, asc); }
    early = false;
    NI_STAMP(2 + 5 * gk);
    const int nmax = ni_stage_compact<NP, NL, EM>(p, ahead, as0, as1, asc, row0, cap, nsf, cstride, srad, L, lane_q, nl, npg);
    NI_STAMP(3 + 5 * gk);
    // (only in the instantiation that reads its pairs from memory: the one with its own pre-pass has no register to spare, and it
    // is the fall-back and the fix-up kernel, not the steady state)
    requested = GPAIRS && !p.fix && gk + 1 < NI_RUN && ii0 + NI_GA < p.inum;
    if (requested) ni_request(p, ii0 + NI_GA, ii0 + NI_GA, lane_q, ahead);       // consumed behind this group's pair loop (ni_preload)
    // more neighbours than these records hold: the group is in the queue (the descriptor pass's fix-up launch left its true
    // counts) and the force pass's own fix-up launch takes it; without a queue it is an error
    if (nmax > limit) {
        if (lane_q == 0 && (p.fix || !p.ovf_list)) atomicMax(p.errflag, nmax);
        if (requested) ni_preload(p, ahead, ii0 + NI_GA, limit, cstride, L, lane_q, as0, as1, asc);
        wave_lds_sync();
        continue;
    }
    wave_lds_sync();
    const double *cw = L.coef + g * cstride + p.npsf;       // angular weights of this lane's atom, visit order
    const int sbase = g * cap;

    const int npl = nl * (nl - 1) / 2;
    const int trips = (nmax * (nmax - 1) / 2 + NI_GL - 1) / NI_GL;
    NiWalk walk = ni_walk_init(l, nl);
    constexpr int CH = ni_ch(true), PLIST = ni_plist(true);
    const unsigned short *gpl = GPAIRS ? p.pairs + (size_t)(row0 + g) * p.pstride : nullptr;
    int pv_next = (GPAIRS && l < npg) ? gpl[l] : 0;         // the entry of the coming trip is in flight while this one is visited
    for (int t0 = 0; t0 < (GPAIRS ? 1 : trips); t0 += CH) {
    int cnt;
    if (GPAIRS) cnt = npg;          // one "chunk": the whole list
    else { cnt = ni_prepass<true>(L, p, walk, g, l, sbase, cap, npl, t0, min(trips, t0 + CH), PLIST); wave_lds_sync(); }
    const int cmax = max(max(__builtin_amdgcn_readlane(cnt, 0), __builtin_amdgcn_readlane(cnt, 16)),
                         max(__builtin_amdgcn_readlane(cnt, 32), __builtin_amdgcn_readlane(cnt, 48)));
    for (int t2 = 0; t2 * NI_GL < cmax; t2++) {
        const int idx = t2 * NI_GL + l;
        const bool lv = idx < cnt;
        int pv;
        if (GPAIRS) { pv = pv_next; pv_next = (idx + NI_GL < cnt) ? gpl[idx + NI_GL] : 0; }
        else pv = lv ? L.pl[g * PLIST + idx] : 0;
        const int sa = lv ? sbase + (pv & 255) : NI_GA * cap, sb = lv ? sbase + (pv >> 8) : NI_GA * cap + 1;
        const NiPairS q = ni_pair(L, p, sa, sb);
        const double r2sum = q.rjm * q.rjm + q.rkm * q.rkm + q.rgm * q.rgm;
        // everything below only matters for pairs inside the cutoffs; keeping the visit inside the branch also keeps
        // its LDS reads next to their use (hoisted out, they would all be live across the branch)
        if (q.ok) {
            // A1 = sum c term1 CFLENGTH, A2 = sum c term2, A3 = sum c term3   (ni:752-754)
            double A1 = 0.0, A2 = 0.0, A3 = 0.0;
            ni_forget_lds();
            if constexpr (NL > 0) ni_force_cart<NL, NE, NZ, ZP, EM>(p, cw, q.ct, r2sum, A1, A2, A3);
            else
                ni_visit_functions<NT, true>(tab, p.ntsf, q.ct, r2sum, [&](int pos, double val, double dval) {
                    const double cc = cw[pos];
                    A3 = fma(cc, val, A3);
                    A2 = fma(cc * tab.sorted[4 * pos], val, A2);
                    A1 = fma(cc, dval, A1);
                });
            ni_forget_lds();
            A1 *= q.tfc * (1.0 / ANNP_CFLENGTH);
            A2 *= q.tfc;
            const double rx = p.compat ? q.rkm : q.rgm;       // ni:737-738 vs lal_annp.cu:409-414
#if NI_KEEP_RECORDS
            const double fcj = q.fcj, fck = q.fck, dfcj = L.dfc[sa], dfck = L.dfc[sb];
            const double irj = q.ij, irk = q.ik;
            const double xj[3] = {q.xj[0], q.xj[1], q.xj[2]}, xk[3] = {q.xk[0], q.xk[1], q.xk[2]};
#else
            const double fcj = L.fc[sa], fck = L.fc[sb], dfcj = L.dfc[sa], dfck = L.dfc[sb];
            const double irj = L.rinv[sa], irk = L.rinv[sb];
            const double xj[3] = {L.dx[sa], L.dy[sa], L.dz[sa]}, xk[3] = {L.dx[sb], L.dy[sb], L.dz[sb]};
#endif
            const double t3j_a = fck * dfcj * q.fcjk, t3_g = fck * fcj * q.dfcjk;
            const double t3k_a = fcj * dfck * q.fcjk;
            double fj[3], fk[3];
#pragma unroll
            for (int d = 0; d < 3; d++) {
                const double ej = xj[d] * irj, ek = xk[d] * irk, gg = (xk[d] - xj[d]) * q.ig;
                const double dctj = (-ek + q.ct * ej) * irj;     // ni:674, fe:618-628
                const double dctk = (-ej + q.ct * ek) * irk;
                const double t2j = 2.0 * (rx * gg - q.rjm * ej), t2k = -2.0 * (q.rkm * ek + rx * gg);
                const double t3j = t3_g * gg - t3j_a * ej, t3k = -(t3k_a * ek + t3_g * gg);
                fj[d] = A1 * dctj - A2 * t2j + A3 * t3j;
                fk[d] = A1 * dctk - A2 * t2k + A3 * t3k;
            }
            atomicAdd(&L.a0[sa], fj[0]); atomicAdd(&L.a1[sa], fj[1]); atomicAdd(&L.a2[sa], fj[2]);
            atomicAdd(&L.a0[sb], fk[0]); atomicAdd(&L.a1[sb], fk[1]); atomicAdd(&L.a2[sb], fk[2]);
        }
    }
    if (!GPAIRS) wave_lds_sync();            // the list is rewritten by the next chunk
    }
    // the coming group's positions, coefficient rows and table slots: requested here, they fly during this group's epilogue
    // (into record arrays the epilogue does not read -- unless it tallies the virial: then it goes first)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    NI_STAMP(4 + 5 * gk);
    if (!VIRIAL && requested) ni_preload(p, ahead, ii0 + NI_GA, limit, cstride, L, lane_q, as0, as1, asc);
    NI_STAMP(5 + 5 * gk);
    double fi0 = 0.0, fi1 = 0.0, fi2 = 0.0;
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0, v4 = 0.0, v5 = 0.0;
    // one neighbour's total (the radial part was there before the pair loop's sums: ni_record), handed to the table and to the centre
    auto finish = [&](int a) {
        if (a < nl) {
            const int s = sbase + a;
            const double g0 = L.a0[s], g1 = L.a1[s], g2 = L.a2[s];
            const int j = L.j[s];
            ni_table_add(L, p.f, L.sl[s], j, -g0 * ANNP_CFFORCE, -g1 * ANNP_CFFORCE, -g2 * ANNP_CFFORCE);       // ni:186-189
            fi0 += g0; fi1 += g1; fi2 += g2;
            if (VIRIAL) {       // the reference tallies the un-converted force (ni:190-198)
                const double d0 = L.dx[s], d1 = L.dy[s], d2 = L.dz[s];
                const double w0 = d0 * g0, w1 = d1 * g1, w2 = d2 * g2, w3 = d0 * g1, w4 = d0 * g2, w5 = d1 * g2;
                v0 += w0; v1 += w1; v2 += w2; v3 += w3; v4 += w4; v5 += w5;
                if (p.vatom) {
                    double *vj = p.vatom + 6 * (size_t)j;
                    atomicAdd(vj + 0, 0.5 * w0); atomicAdd(vj + 1, 0.5 * w1); atomicAdd(vj + 2, 0.5 * w2);
                    atomicAdd(vj + 3, 0.5 * w3); atomicAdd(vj + 4, 0.5 * w4); atomicAdd(vj + 5, 0.5 * w5);
                }
            }
        }
    };
    for (int a = l; a < nmax; a += NI_GL) finish(a);
    // group sums (16 lanes) -> the centre atom
    const int i = L.ci[g];
    // (a group is a DPP row: four row shifts leave its sum in its last lane, no LDS traffic)
    fi0 = row16_sum_to_last(fi0); fi1 = row16_sum_to_last(fi1); fi2 = row16_sum_to_last(fi2);
    if (l == NI_GL - 1 && i >= 0) ni_table_add(L, p.f, L.cs[g], i, fi0 * ANNP_CFFORCE, fi1 * ANNP_CFFORCE, fi2 * ANNP_CFFORCE);
    if (VIRIAL) {
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            v0 += __shfl_xor(v0, off, 64); v1 += __shfl_xor(v1, off, 64); v2 += __shfl_xor(v2, off, 64);
            v3 += __shfl_xor(v3, off, 64); v4 += __shfl_xor(v4, off, 64); v5 += __shfl_xor(v5, off, 64);
        }
        if (l == 0 && i >= 0 && p.vatom) {
            double *vi = p.vatom + 6 * (size_t)i;
            atomicAdd(vi + 0, 0.5 * v0); atomicAdd(vi + 1, 0.5 * v1); atomicAdd(vi + 2, 0.5 * v2);
            atomicAdd(vi + 3, 0.5 * v3); atomicAdd(vi + 4, 0.5 * v4); atomicAdd(vi + 5, 0.5 * v5);
        }
        if (p.virial) {
#pragma unroll
            for (int off = 32; off >= 16; off >>= 1) {
                v0 += __shfl_xor(v0, off, 64); v1 += __shfl_xor(v1, off, 64); v2 += __shfl_xor(v2, off, 64);
                v3 += __shfl_xor(v3, off, 64); v4 += __shfl_xor(v4, off, 64); v5 += __shfl_xor(v5, off, 64);
            }
            if (lane == 0) {
                double *vr = virial_row(p.virial);          // (annp_common.hpp: the global virial)
                atomicAdd(&vr[0], v0); atomicAdd(&vr[1], v1); atomicAdd(&vr[2], v2);
                atomicAdd(&vr[3], v3); atomicAdd(&vr[4], v4); atomicAdd(&vr[5], v5);
            }
        }
    }
    if (VIRIAL && requested) { wave_lds_sync(); ni_preload(p, ahead, ii0 + NI_GA, limit, cstride, L, lane_q, as0, as1, asc); }
    wave_lds_sync();            // the next group of the run reuses the records
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    NI_STAMP(6 + 5 * gk);
    }
    NI_STAMP(22);
    // flush the run's table: one global atomic per distinct atom and component, the three components of an atom from three
    // neighbouring lanes of one instruction -- float atomics are executed at the memory side, one request per 64-byte line an
    // instruction touches, and a lane per atom issuing x, then y, then z made three requests of what is one (or two) lines
    wave_lds_sync();
    for (int k = lane; k < 3 * NI_TSLOTS; k += 64) {
        const int sl = k / 3, c = k - 3 * sl;
        const int j = L.tkey[sl];
        if (j >= 0) atomicAdd(&p.f[3 * (size_t)j + c], L.tacc[k]);
    }
    NI_STAMP(23);
#ifdef ANNP_NI_STAMPS
    if (!p.fix && lane == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        reinterpret_cast<unsigned long long *>(p.G + (size_t)run * NI_RUN * NI_GA * ANNP_GPAD)[24] = hw;
    }
#endif
}

// product shape of the angular set, filled by annp_hip_init: nl*ne*nz == ntsf when it is a full product
// {lambda} x {eta} x {zeta} (else nl = 0); zp packs the zetas, em the integer ratios eta_e / eta_0 (0 = not one)
struct NiShape { int nl, ne, nz; unsigned zp, em; };

// the shape of the shipped ni_annp_potential_2.ann: lambda = -1,+1; eta = 0.01 x {1,2,5}; zeta = 1,2,4,16
#define NI_SHIPPED 3, 24, 2, 3, 4, (1u | 2u << 8 | 4u << 16 | 16u << 24), (1u | 2u << 8 | 5u << 16)
#define NI_GENERIC NI_MAXP, NI_MAXT, 0, 0, 0, 0u, 0u
inline bool ni_is_shipped_shape(const NiArgs &a, NiShape sh)
{
    unsigned long long want = 0;        // radial etas in the same ratios 1 : 2 : 5 (as many as there are)
    for (int m = 0; m < a.npsf && m < 3; m++) want |= (unsigned long long)((m == 0) ? 1 : (m == 1) ? 2 : 5) << (8 * m);
    return a.npsf <= 3 && a.rad_em == want && a.ntsf == 24 && sh.nl == 2 && sh.ne == 3 && sh.nz == 4 &&
           sh.zp == (1u | 2u << 8 | 4u << 16 | 16u << 24) && sh.em == (1u | 2u << 8 | 5u << 16);
}

inline size_t ni_lds_block(int cap, bool force, int nsf, bool gpairs = false)
{
    return NI_TABLE_DOUBLES * 8 + ni_lds_per_wave(cap, force, nsf, gpairs) * ANNP_WAVES_PER_BLOCK;
}

// largest record capacity whose 4-wave block still fits the 160 KB of a CU
inline int ni_max_cap(bool force, int nsf)
{
    int cap = 8;
    while (ni_lds_block(cap + 8, force, nsf) <= 160 * 1024) cap += 8;
    return cap;
}

inline int ni_blocks(int inum) { return (inum + ANNP_WAVES_PER_BLOCK * NI_GA - 1) / (ANNP_WAVES_PER_BLOCK * NI_GA); }

inline int ni_launch_desc(const NiArgs &a, NiShape sh, hipStream_t s)
{
    const size_t lds = ni_lds_block(a.n_cap, false, a.npsf + a.ntsf);
    if (a.npsf > NI_MAXP || a.ntsf > NI_MAXT) return -1;
    const int blocks = ni_blocks(a.inum);
    if (ni_is_shipped_shape(a, sh))
        if (a.n_cap == NI_CAP_FIXED) hipLaunchKernelGGL((annp_ni_desc<NI_SHIPPED, false, NI_CAP_FIXED>), dim3(blocks), dim3(256), lds, s, a);
        else hipLaunchKernelGGL((annp_ni_desc<NI_SHIPPED, false>), dim3(blocks), dim3(256), lds, s, a);
    else
        hipLaunchKernelGGL((annp_ni_desc<NI_GENERIC, false>), dim3(blocks), dim3(256), lds, s, a);
    return 0;
}

template <bool VIR, bool GP>
inline void ni_launch_force_t(const NiArgs &a, NiShape sh, hipStream_t s)
{
    const size_t lds = ni_lds_block(a.n_cap, true, a.npsf + a.ntsf, GP);
    const int per_block = ANNP_WAVES_PER_BLOCK * NI_GA * NI_RUN;
    const int blocks = (a.inum + per_block - 1) / per_block;
    if (ni_is_shipped_shape(a, sh)) {
        if (GP && a.n_cap <= NI_CAP_FIXED) {        // the capacity compiled in (fcc Ni: 18 neighbours inside the cutoff; fewer bytes of LDS than this buy no further workgroup)
            NiArgs c = a;
            c.n_cap = NI_CAP_FIXED;
            hipLaunchKernelGGL((annp_ni_force<NI_SHIPPED, VIR, GP, GP ? NI_CAP_FIXED : 0>), dim3(blocks), dim3(256), ni_lds_block(NI_CAP_FIXED, true, a.npsf + a.ntsf, GP), s, c);
        } else hipLaunchKernelGGL((annp_ni_force<NI_SHIPPED, VIR, GP>), dim3(blocks), dim3(256), lds, s, a);
    } else hipLaunchKernelGGL((annp_ni_force<NI_GENERIC, VIR, GP>), dim3(blocks), dim3(256), lds, s, a);
}

// the fix-up launches: a fixed small grid whose waves walk the queue (empty in the steady state: they exit at once)
constexpr int NI_FIX_BLOCKS = 64;
inline void ni_launch_desc_fix(const NiArgs &a, NiShape sh, hipStream_t s)
{
    const size_t lds = ni_lds_block(a.n_cap, false, a.npsf + a.ntsf);
    const int blocks = std::max(1, std::min(NI_FIX_BLOCKS, (a.ovf_cap + ANNP_WAVES_PER_BLOCK - 1) / ANNP_WAVES_PER_BLOCK));
    if (ni_is_shipped_shape(a, sh)) hipLaunchKernelGGL((annp_ni_desc<NI_SHIPPED, true>), dim3(blocks), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((annp_ni_desc<NI_GENERIC, true>), dim3(blocks), dim3(256), lds, s, a);
}
inline void ni_launch_force_fix(const NiArgs &a, NiShape sh, bool virial, hipStream_t s)
{
    const size_t lds = ni_lds_block(a.n_cap, true, a.npsf + a.ntsf, false);
    const int blocks = std::max(1, std::min(NI_FIX_BLOCKS, (a.ovf_cap + ANNP_WAVES_PER_BLOCK - 1) / ANNP_WAVES_PER_BLOCK));
    const bool shipped = ni_is_shipped_shape(a, sh);
    if (virial) {
        if (shipped) hipLaunchKernelGGL((annp_ni_force<NI_SHIPPED, true, false>), dim3(blocks), dim3(256), lds, s, a);
        else hipLaunchKernelGGL((annp_ni_force<NI_GENERIC, true, false>), dim3(blocks), dim3(256), lds, s, a);
    } else {
        if (shipped) hipLaunchKernelGGL((annp_ni_force<NI_SHIPPED, false, false>), dim3(blocks), dim3(256), lds, s, a);
        else hipLaunchKernelGGL((annp_ni_force<NI_GENERIC, false, false>), dim3(blocks), dim3(256), lds, s, a);
    }
}

// a.pairs != nullptr: the descriptor pass of this evaluation left the pair lists
inline void ni_launch_force(const NiArgs &a, NiShape sh, bool virial, hipStream_t s)
{
    if (a.pairs) { if (virial) ni_launch_force_t<true, true>(a, sh, s); else ni_launch_force_t<false, true>(a, sh, s); }
    else { if (virial) ni_launch_force_t<true, false>(a, sh, s); else ni_launch_force_t<false, false>(a, sh, s); }
}

// the kernels may ask for more than the default 64 KB of dynamic LDS
inline hipError_t ni_set_lds_attributes()
{
    const int full = 160 * 1024;
    hipError_t e;
#define NI_ATTR(...) if ((e = hipFuncSetAttribute((const void *)__VA_ARGS__, hipFuncAttributeMaxDynamicSharedMemorySize, full)) != hipSuccess) return e
    NI_ATTR(annp_ni_desc<NI_SHIPPED, false>);
    NI_ATTR(annp_ni_desc<NI_SHIPPED, false, NI_CAP_FIXED>);
    NI_ATTR(annp_ni_desc<NI_GENERIC, false>);
    NI_ATTR(annp_ni_desc<NI_SHIPPED, true>);
    NI_ATTR(annp_ni_desc<NI_GENERIC, true>);
    NI_ATTR(annp_ni_force<NI_SHIPPED, true, true>);
    NI_ATTR(annp_ni_force<NI_SHIPPED, false, true>);
    NI_ATTR(annp_ni_force<NI_SHIPPED, true, true, NI_CAP_FIXED>);
    NI_ATTR(annp_ni_force<NI_SHIPPED, false, true, NI_CAP_FIXED>);
    NI_ATTR(annp_ni_force<NI_GENERIC, true, true>);
    NI_ATTR(annp_ni_force<NI_GENERIC, false, true>);
    NI_ATTR(annp_ni_force<NI_SHIPPED, true, false>);
    NI_ATTR(annp_ni_force<NI_SHIPPED, false, false>);
    NI_ATTR(annp_ni_force<NI_GENERIC, true, false>);
    NI_ATTR(annp_ni_force<NI_GENERIC, false, false>);
#undef NI_ATTR
    return hipSuccess;
}

}  // namespace annp
