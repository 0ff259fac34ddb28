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
// r*CFLENGTH < Rc ones (≈18 in fcc Ni) contribute; here neighbours are filtered into
// LDS first and the n(n-1)/2 in-range pairs are dealt flat over the 64 lanes.
// Roles follow list order (j before k) because compat mode is not symmetric in j,k.
#pragma once
#include "annp_common.hpp"

namespace annp {

#define ANNP_CFLENGTH 1.889726    // ni/src/pair_annp.h:69
#define ANNP_CFFORCE 51.422515    // ni/src/pair_annp.h:70

constexpr int NI_NCAP = 128;      // in-range neighbours held per wave
constexpr int NI_MAXP = 8;        // radial functions supported
constexpr int NI_MAXT = 32;       // angular functions supported

struct NiArgs {
    int inum, n_cap;
    const int *ilist;
    const double *x;
    const int *numneigh;
    const long long *first;
    const int *neigh;
    int npsf, ntsf, compat;
    const double *sym;          // see "per-function tables" below
    const int *isym;
    double rc_rad, rc_ang;      // Bohr
    double *G;
    const double *coef;         // [inum][ANNP_CPAD]: c_k = dE/dGhat_k / (sf_max-sf_min)_k
    double *f;
    double *virial;
    double *vatom;              // nullable, [nall][6] accumulated (VIRIAL variant)
    int *ncount;
    int *errflag;
};

__host__ __device__ inline size_t ni_lds_per_wave() { return (size_t)NI_NCAP * (7 * 8 + 3 * 8 + 8); }

// flat pair index -> (a,b), a < b < n, rows a=0: (0,1)..(0,n-1), a=1: ...
__device__ __forceinline__ void ni_decode_pair(int p, int n, int &a, int &b)
{
    const float fn = (float)(2 * n - 1);
    int r = (int)((fn - sqrtf(fn * fn - 8.0f * (float)p)) * 0.5f);
    r = max(0, min(r, n - 2));
    // start(r) = r(2n-r-1)/2
    while (r > 0 && r * (2 * n - r - 1) / 2 > p) r--;
    while ((r + 1) * (2 * n - r - 2) / 2 <= p) r++;
    a = r;
    b = p - r * (2 * n - r - 1) / 2 + r + 1;
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
    const double *rad, *sorted, *etas;
    const int *perm, *eidx, *zint, *emult;
    int ne;
};
__device__ __forceinline__ NiTab ni_tab(const double *sym, const int *isym, int npsf, int ntsf)
{
    NiTab t;
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
    E[0] = exp(-t.etas[0] * r2sum);
#pragma unroll
    for (int e = 1; e < NI_MAXE; e++) {
        E[e] = 0.0;
        if (e < t.ne) E[e] = (t.emult[e] > 0) ? ni_powi(E[0], t.emult[e]) : exp(-t.etas[e] * r2sum);
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

// Same visit when the function set is a full product {lambda} x {eta} x {zeta} (as in the shipped
// Ni potential: 2 x 3 x 4): the per-function tables collapse to NL + NE + 2 NZ scalars that live
// in registers, so the inner loop is free of table loads: per function 2 multiplies + the caller's FMA.
// Visit position = (l * NE + e) * NZ + z, identical to the sorted order above.
struct NiCart {
    double lam[4], pref[8], zeta[8];
    int zint[8];
};
template <int NL, int NE, int NZ>
__device__ __forceinline__ NiCart ni_cart_load(const NiTab &t)
{
    NiCart c;
#pragma unroll
    for (int l = 0; l < NL; l++) c.lam[l] = t.sorted[4 * (l * NE * NZ) + 1];
#pragma unroll
    for (int z = 0; z < NZ; z++) {
        c.zeta[z] = t.sorted[4 * z + 2];
        c.pref[z] = t.sorted[4 * z + 3];
        c.zint[z] = t.zint[z];
    }
    return c;
}
template <int NL, int NE, int NZ, bool DERIV, typename Body>
__device__ __forceinline__ void ni_visit_cart(const NiTab &t, const NiCart &c, double ct, double r2sum, Body &&body)
{
    double E[NE];
    E[0] = exp(-t.etas[0] * r2sum);
#pragma unroll
    for (int e = 1; e < NE; e++) E[e] = (t.emult[e] > 0) ? ni_powi(E[0], t.emult[e]) : exp(-t.etas[e] * r2sum);
#pragma unroll
    for (int l = 0; l < NL; l++) {
        const double u = fma(c.lam[l], ct, 1.0);
        const bool ok = u > 0.0;
        const double U0 = ok ? 1.0 : 0.0;
        double U[5];
        U[0] = ok ? u : 0.0;
        U[1] = U[0] * U[0]; U[2] = U[1] * U[1]; U[3] = U[2] * U[2]; U[4] = U[3] * U[3];
#pragma unroll
        for (int z = 0; z < NZ; z++) {
            const double pw = c.pref[z] * ni_ladder_pow(U, U0, c.zint[z]);
            double dw = 0.0;
            if (DERIV) dw = (c.zint[z] >= 1) ? c.pref[z] * c.zeta[z] * c.lam[l] * ni_ladder_pow(U, U0, c.zint[z] - 1) : 0.0;
#pragma unroll
            for (int e = 0; e < NE; e++) body((l * NE + e) * NZ + z, pw * E[e], DERIV ? dw * E[e] : 0.0);
        }
    }
}

// Shared by both passes: filter neighbours into LDS.  Record a: xij (3), r, fc_ang, dfc_ang.
struct NiLds {
    double *dx, *dy, *dz, *r, *rinv, *fc, *dfc;   // [NI_NCAP] each
    double *a0, *a1, *a2;                  // force accumulators
    int *j;
};

__device__ __forceinline__ NiLds ni_carve(unsigned char *wbase)
{
    NiLds L;
    L.dx = reinterpret_cast<double *>(wbase);
    L.dy = L.dx + NI_NCAP; L.dz = L.dy + NI_NCAP; L.r = L.dz + NI_NCAP; L.rinv = L.r + NI_NCAP;
    L.fc = L.rinv + NI_NCAP; L.dfc = L.fc + NI_NCAP;
    L.a0 = L.dfc + NI_NCAP; L.a1 = L.a0 + NI_NCAP; L.a2 = L.a1 + NI_NCAP;
    L.j = reinterpret_cast<int *>(L.a2 + NI_NCAP);
    return L;
}

__device__ __forceinline__ int ni_stage(const NiArgs &p, int i, const NiLds &L, int lane)
{
    const double xi = p.x[3 * (size_t)i], yi = p.x[3 * (size_t)i + 1], zi = p.x[3 * (size_t)i + 2];
    const long long base = p.first[i];
    const int jn = p.numneigh[i];
    const double rcmax = fmax(p.rc_rad, p.rc_ang);
    const double rc2 = (rcmax / ANNP_CFLENGTH) * (rcmax / ANNP_CFLENGTH) * (1.0 + 1e-12);   // coarse filter in A^2
    const double pi_over_rc = ANNP_MY_PI / p.rc_ang;
    // first sweep: cheap distance filter, candidates compacted raw.  Four 64-candidate groups per
    // trip so that the index loads, then the 12 coordinate gathers, are all in flight together
    // (only ~18 of ~224 candidates survive: this sweep is pure memory latency otherwise).
    int n = 0;
    for (int c0 = 0; c0 < jn; c0 += 256) {
        int j[4];
        bool valid[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int jj = c0 + 64 * u + lane;
            valid[u] = jj < jn;
            j[u] = valid[u] ? (p.neigh[base + jj] & ANNP_NEIGHMASK) : i;
        }
        double dx[4], dy[4], dz[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            dx[u] = xi - p.x[3 * (size_t)j[u]]; dy[u] = yi - p.x[3 * (size_t)j[u] + 1]; dz[u] = zi - p.x[3 * (size_t)j[u] + 2];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const double rsq = dx[u] * dx[u] + dy[u] * dy[u] + dz[u] * dz[u];
            const bool in = valid[u] && rsq < rc2 && rsq > 0.0;
            const unsigned long long m = __ballot(in);
            const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
            if (in && pos < NI_NCAP) { L.dx[pos] = dx[u]; L.dy[pos] = dy[u]; L.dz[pos] = dz[u]; L.r[pos] = rsq; L.j[pos] = j[u]; }
            n += __popcll(m);
        }
    }
    n = uniform(n);
    if (n > NI_NCAP) return n;
    wave_lds_sync();
    // second sweep: the exact test of the reference (r * CFLENGTH < Rc, ni:693/729) and the per-neighbour terms.
    // Entries that fail it keep fc = 0 and r = huge, so every pair they enter is rejected by ni_pair.
    for (int a = lane; a < n; a += 64) {
        const double rsq = L.r[a];
        const double rinv = fast_rsqrt(rsq);
        const double r = rsq * rinv;
        const double rm = r * ANNP_CFLENGTH;
        double fc = 0.0, dfc = 0.0;
        if (rm < p.rc_ang) {
            double sn, cs;
            sincos_0_pi(pi_over_rc * rm, sn, cs);
            fc = 0.5 * (cs + 1.0);
            dfc = -0.5 * pi_over_rc * sn;
        }
        L.r[a] = r; L.rinv[a] = rinv; L.fc[a] = fc; L.dfc[a] = dfc;
        L.a0[a] = 0.0; L.a1[a] = 0.0; L.a2[a] = 0.0;
    }
    return n;
}

// geometry of one (j,k) pair around the centre
struct NiPair {
    double ej[3], ek[3], g[3];      // xij/rij, xik/rik, xjk/rjk
    double rj, rk, rjk, ct;
    double fcj, fck, fcjk, dfcj, dfck, dfcjk;
    bool ok;
};

__device__ __forceinline__ NiPair ni_pair(const NiArgs &p, const NiLds &L, int a, int b)
{
    NiPair q;
    q.rj = L.r[a]; q.rk = L.r[b];
    const double xj0 = L.dx[a], xj1 = L.dy[a], xj2 = L.dz[a];
    const double xk0 = L.dx[b], xk1 = L.dy[b], xk2 = L.dz[b];
    const double ij = L.rinv[a], ik = L.rinv[b];
    q.ej[0] = xj0 * ij; q.ej[1] = xj1 * ij; q.ej[2] = xj2 * ij;
    q.ek[0] = xk0 * ik; q.ek[1] = xk1 * ik; q.ek[2] = xk2 * ik;
    // xjk = x_j - x_k = xik - xij
    const double g0 = xk0 - xj0, g1 = xk1 - xj1, g2 = xk2 - xj2;
    const double gsq = g0 * g0 + g1 * g1 + g2 * g2;
    const double ig = fast_rsqrt(gsq);
    q.rjk = gsq * ig;
    q.g[0] = g0 * ig; q.g[1] = g1 * ig; q.g[2] = g2 * ig;
    q.ct = q.ej[0] * q.ek[0] + q.ej[1] * q.ek[1] + q.ej[2] * q.ek[2];
    q.fcj = L.fc[a]; q.fck = L.fc[b]; q.dfcj = L.dfc[a]; q.dfck = L.dfc[b];
    const double rc = p.rc_ang;
    const double rgm = q.rjk * ANNP_CFLENGTH;
    q.ok = (q.rj * ANNP_CFLENGTH < rc) && (q.rk * ANNP_CFLENGTH < rc) && (rgm < rc);   // ni:729
    q.fcjk = 0.0; q.dfcjk = 0.0;
    if (q.ok) {
        double sn, cs;
        const double por = ANNP_MY_PI / rc;
        sincos_0_pi(por * rgm, sn, cs);
        q.fcjk = 0.5 * (cs + 1.0);
        q.dfcjk = -0.5 * por * sn;
    }
    return q;
}

// ---------------------------------------------------------------------------------
template <int NP, int NT, int NL, int NE, int NZ>
__global__ __launch_bounds__(256) void annp_ni_desc(NiArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = lane_id();
    const int wave = uniform(threadIdx.x >> 6);
    const int ii = uniform(xcd_block() * ANNP_WAVES_PER_BLOCK + wave);
    if (ii >= p.inum) return;
    unsigned char *wbase = lds_raw + (size_t)wave * ni_lds_per_wave();
    const NiLds L = ni_carve(wbase);
    double *scratch = reinterpret_cast<double *>(wbase);
    const int i = p.ilist ? p.ilist[ii] : ii;
    const int n = ni_stage(p, i, L, lane);
    if (p.ncount && lane == 0) p.ncount[ii] = min(n, NI_NCAP);
    double *Gout = p.G + (size_t)ii * ANNP_GPAD;
    if (n > NI_NCAP) {
        if (lane == 0) atomicMax(p.errflag, n);
        if (lane < ANNP_GPAD) Gout[lane] = 0.0;
        return;
    }
    wave_lds_sync();
    const NiTab tab = ni_tab(p.sym, p.isym, p.npsf, p.ntsf);
    const double *srad = tab.rad;

    double gr[NP], ga[NT];
#pragma unroll
    for (int m = 0; m < NP; m++) gr[m] = 0.0;
#pragma unroll
    for (int m = 0; m < NT; m++) ga[m] = 0.0;      // indexed by visit position
    // G2 (ni:686-711): lane a owns neighbour a
    for (int a = lane; a < n; a += 64) {
        const double rm = L.r[a] * ANNP_CFLENGTH;
        if (rm < p.rc_rad) {
            double sn, cs;
            sincos_0_pi(ANNP_MY_PI / p.rc_rad * rm, sn, cs);
            const double fc = 0.5 * (cs + 1.0);
#pragma unroll
            for (int m = 0; m < NP; m++)
                if (m < p.npsf) gr[m] += exp(-srad[3 * m] * rm * rm) * fc;
        }
    }
    // G4 (ni:713-767)
    NiCart cart;
    if constexpr (NL > 0) cart = ni_cart_load<NL, NE, NZ>(tab);
    const int npairs = n * (n - 1) / 2;
    for (int p0 = 0; p0 < npairs; p0 += 64) {
        const int pp = p0 + lane;
        int a = 0, b = 1;
        if (pp < npairs) ni_decode_pair(pp, n, a, b);
        const NiPair q = ni_pair(p, L, a, b);
        const double rjm = q.rj * ANNP_CFLENGTH, rkm = q.rk * ANNP_CFLENGTH, rgm = q.rjk * ANNP_CFLENGTH;
        const double r2sum = rjm * rjm + rkm * rkm + rgm * rgm;
        const double tfc = (pp < npairs && q.ok) ? q.fcj * q.fck * q.fcjk : 0.0;     // masked-off lanes add zeros
        if constexpr (NL > 0)
            ni_visit_cart<NL, NE, NZ, false>(tab, cart, q.ct, r2sum,
                                             [&](int pos, double val, double) { ga[pos] = fma(val, tfc, ga[pos]); });
        else
            ni_visit_functions<NT, false>(tab, p.ntsf, q.ct, r2sum,
                                          [&](int pos, double val, double) { ga[pos] = fma(val, tfc, ga[pos]); });
    }
    wave_lds_sync();
    // reduce through LDS, 8 sums per round (records are dead now)
    constexpr int NS = NP + NT;
    const int nsf = p.npsf + p.ntsf;
#pragma unroll
    for (int c8 = 0; c8 < (NS + 7) / 8; c8++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int m = c8 * 8 + k;       // slot in the padded (NP | NT) layout
            double v = 0.0;
            if (m < NP) v = gr[m < NP ? m : 0];
            else if (m < NS) v = ga[(m - NP) >= 0 && (m - NP) < NT ? (m - NP) : 0];
            scratch[k * 64 + lane] = v;
        }
        wave_lds_sync();
        {
            const int k = lane >> 3, part = lane & 7;
            const double *src = scratch + k * 64 + part * 8;
            double s = 0.0;
#pragma unroll
            for (int u = 0; u < 8; u++) s += src[u];
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
            const int m = c8 * 8 + k;
            // padded slot -> function index
            int fidx = -1;
            if (m < NP) { if (m < p.npsf) fidx = m; }
            else if (m - NP < p.ntsf) fidx = p.npsf + tab.perm[m - NP];
            if (part == 0 && fidx >= 0) Gout[fidx] = s;
        }
        wave_lds_sync();
    }
    if (lane >= nsf && lane < ANNP_GPAD) Gout[lane] = 0.0;
}

// ---------------------------------------------------------------------------------
template <int NP, int NT, int NL, int NE, int NZ, bool VIRIAL>
__global__ __launch_bounds__(256) void annp_ni_force(NiArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = lane_id();
    const int wave = uniform(threadIdx.x >> 6);
    const int ii = uniform(xcd_block() * ANNP_WAVES_PER_BLOCK + wave);
    if (ii >= p.inum) return;
    unsigned char *wbase = lds_raw + (size_t)wave * ni_lds_per_wave();
    const NiLds L = ni_carve(wbase);
    const int i = p.ilist ? p.ilist[ii] : ii;
    const int n = ni_stage(p, i, L, lane);
    if (n > NI_NCAP) { if (lane == 0) atomicMax(p.errflag, n); return; }
    wave_lds_sync();
    const NiTab tab = ni_tab(p.sym, p.isym, p.npsf, p.ntsf);
    const double *srad = tab.rad;
    const double *cf = p.coef + (size_t)ii * ANNP_CPAD;
    double cs_[NT];                                 // weights in visit order
#pragma unroll
    for (int m = 0; m < NT; m++) cs_[m] = (m < p.ntsf) ? cf[p.npsf + tab.perm[m]] : 0.0;

    NiCart cart;
    if constexpr (NL > 0) cart = ni_cart_load<NL, NE, NZ>(tab);
    double ceta[NT];                                // c_m eta_m in visit order (term2, ni:754)
#pragma unroll
    for (int m = 0; m < NT; m++) ceta[m] = (m < p.ntsf) ? cs_[m] * tab.sorted[4 * m] : 0.0;
    const int npairs = n * (n - 1) / 2;
    for (int p0 = 0; p0 < npairs; p0 += 64) {
        const int pp = p0 + lane;
        int a = 0, b = 1;
        if (pp < npairs) ni_decode_pair(pp, n, a, b);
        const NiPair q = ni_pair(p, L, a, b);
        const bool live = pp < npairs && q.ok;
        const double rjm = q.rj * ANNP_CFLENGTH, rkm = q.rk * ANNP_CFLENGTH, rgm = q.rjk * ANNP_CFLENGTH;
        const double r2sum = rjm * rjm + rkm * rkm + rgm * rgm;
        const double tfc = live ? q.fcj * q.fck * q.fcjk : 0.0;
        // A1 = sum c term1 CFLENGTH, A2 = sum c term2, A3 = sum c term3   (ni:752-754)
        double A1 = 0.0, A2 = 0.0, A3 = 0.0;
        auto acc3 = [&](int pos, double val, double dval) {
            A3 = fma(cs_[pos], val, A3);
            A2 = fma(ceta[pos], val, A2);
            A1 = fma(cs_[pos], dval, A1);
        };
        if constexpr (NL > 0) ni_visit_cart<NL, NE, NZ, true>(tab, cart, q.ct, r2sum, acc3);
        else ni_visit_functions<NT, true>(tab, p.ntsf, q.ct, r2sum, acc3);
        if (live) {
            A1 *= tfc * (1.0 / ANNP_CFLENGTH);
            A2 *= tfc;
            const double rx = p.compat ? rkm : rgm;           // ni:737-738 vs lal_annp.cu:409-414
            const double t3j_a = q.fck * q.dfcj * q.fcjk, t3j_g = q.fck * q.fcj * q.dfcjk;
            const double t3k_a = q.fcj * q.dfck * q.fcjk, t3k_g = q.fcj * q.fck * q.dfcjk;
            const double irj = L.rinv[a], irk = L.rinv[b];
            double fj[3], fk[3];
#pragma unroll
            for (int d = 0; d < 3; d++) {
                const double dctj = (-q.ek[d] + q.ct * q.ej[d]) * irj;     // ni:674, fe:618-628
                const double dctk = (-q.ej[d] + q.ct * q.ek[d]) * irk;
                const double drj = -q.ej[d], drk = -q.ek[d], g = q.g[d];
                const double t2j = 2.0 * (rjm * drj + rx * g), t2k = 2.0 * (rkm * drk - rx * g);
                const double t3j = t3j_a * drj + t3j_g * g, t3k = t3k_a * drk - t3k_g * g;
                fj[d] = A1 * dctj - A2 * t2j + A3 * t3j;
                fk[d] = A1 * dctk - A2 * t2k + A3 * t3k;
            }
            atomicAdd(&L.a0[a], fj[0]); atomicAdd(&L.a1[a], fj[1]); atomicAdd(&L.a2[a], fj[2]);
            atomicAdd(&L.a0[b], fk[0]); atomicAdd(&L.a1[b], fk[1]); atomicAdd(&L.a2[b], fk[2]);
        }
    }
    wave_lds_sync();
    double fi0 = 0.0, fi1 = 0.0, fi2 = 0.0;
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0, v4 = 0.0, v5 = 0.0;
    for (int a = lane; a < n; a += 64) {
        double g0 = L.a0[a], g1 = L.a1[a], g2 = L.a2[a];
        const double r = L.r[a], rm = r * ANNP_CFLENGTH;
        const double d0 = L.dx[a], d1 = L.dy[a], d2 = L.dz[a];
        if (rm < p.rc_rad) {                                     // ni:693-709
            double sn, cs;
            const double por = ANNP_MY_PI / p.rc_rad;
            sincos_0_pi(por * rm, sn, cs);
            const double fc = 0.5 * (cs + 1.0), dfc = -0.5 * por * sn;
            double R = 0.0;
#pragma unroll
            for (int m = 0; m < NP; m++)
                if (m < p.npsf) {
                    const double eta = srad[3 * m];
                    R = fma(cf[m], exp(-eta * rm * rm) * (-fc * 2.0 * eta * rm + dfc), R);
                }
            const double s = -R * L.rinv[a];                      // dr_dj = -xij/rij
            g0 = fma(s, d0, g0); g1 = fma(s, d1, g1); g2 = fma(s, d2, g2);
        }
        const int j = L.j[a];
        atomicAdd(&p.f[3 * (size_t)j], -g0 * ANNP_CFFORCE);       // ni:186-189
        atomicAdd(&p.f[3 * (size_t)j + 1], -g1 * ANNP_CFFORCE);
        atomicAdd(&p.f[3 * (size_t)j + 2], -g2 * ANNP_CFFORCE);
        fi0 += g0; fi1 += g1; fi2 += g2;
        if (VIRIAL) {       // the reference tallies the un-converted force (ni:190-198)
            const double w0 = d0 * g0, w1 = d1 * g1, w2 = d2 * g2, w3 = d0 * g1, w4 = d0 * g2, w5 = d1 * g2;
            v0 += w0; v1 += w1; v2 += w2; v3 += w3; v4 += w4; v5 += w5;
            if (p.vatom) {
                double *vj = p.vatom + 6 * (size_t)j;
                atomicAdd(vj + 0, 0.5 * w0); atomicAdd(vj + 1, 0.5 * w1); atomicAdd(vj + 2, 0.5 * w2);
                atomicAdd(vj + 3, 0.5 * w3); atomicAdd(vj + 4, 0.5 * w4); atomicAdd(vj + 5, 0.5 * w5);
            }
        }
    }
    fi0 = wave_sum(fi0); fi1 = wave_sum(fi1); fi2 = wave_sum(fi2);
    if (lane == 0) {
        atomicAdd(&p.f[3 * (size_t)i], fi0 * ANNP_CFFORCE);
        atomicAdd(&p.f[3 * (size_t)i + 1], fi1 * ANNP_CFFORCE);
        atomicAdd(&p.f[3 * (size_t)i + 2], fi2 * ANNP_CFFORCE);
    }
    if (VIRIAL) {
        v0 = wave_sum(v0); v1 = wave_sum(v1); v2 = wave_sum(v2);
        v3 = wave_sum(v3); v4 = wave_sum(v4); v5 = wave_sum(v5);
        if (lane == 0) {
            if (p.virial) {
                atomicAdd(&p.virial[0], v0); atomicAdd(&p.virial[1], v1); atomicAdd(&p.virial[2], v2);
                atomicAdd(&p.virial[3], v3); atomicAdd(&p.virial[4], v4); atomicAdd(&p.virial[5], v5);
            }
            if (p.vatom) {
                double *vi = p.vatom + 6 * (size_t)i;
                atomicAdd(vi + 0, 0.5 * v0); atomicAdd(vi + 1, 0.5 * v1); atomicAdd(vi + 2, 0.5 * v2);
                atomicAdd(vi + 3, 0.5 * v3); atomicAdd(vi + 4, 0.5 * v4); atomicAdd(vi + 5, 0.5 * v5);
            }
        }
    }
}

// product shape of the angular set, filled by annp_hip_init: nl*ne*nz == ntsf when it is a full product, else nl = 0
struct NiShape { int nl, ne, nz; };

inline int ni_launch_desc(const NiArgs &a, NiShape sh, int blocks, hipStream_t s)
{
    const size_t lds = ni_lds_per_wave() * ANNP_WAVES_PER_BLOCK;
    if (a.npsf > NI_MAXP || a.ntsf > NI_MAXT) return -1;
    if (a.npsf <= 3 && a.ntsf == 24 && sh.nl == 2 && sh.ne == 3 && sh.nz == 4)
        hipLaunchKernelGGL((annp_ni_desc<3, 24, 2, 3, 4>), dim3(blocks), dim3(256), lds, s, a);
    else
        hipLaunchKernelGGL((annp_ni_desc<NI_MAXP, NI_MAXT, 0, 0, 0>), dim3(blocks), dim3(256), lds, s, a);
    return 0;
}

inline void ni_launch_force(const NiArgs &a, NiShape sh, int blocks, bool virial, hipStream_t s)
{
    const size_t lds = ni_lds_per_wave() * ANNP_WAVES_PER_BLOCK;
    if (a.npsf <= 3 && a.ntsf == 24 && sh.nl == 2 && sh.ne == 3 && sh.nz == 4) {
        if (virial) hipLaunchKernelGGL((annp_ni_force<3, 24, 2, 3, 4, true>), dim3(blocks), dim3(256), lds, s, a);
        else hipLaunchKernelGGL((annp_ni_force<3, 24, 2, 3, 4, false>), dim3(blocks), dim3(256), lds, s, a);
    } else {
        if (virial) hipLaunchKernelGGL((annp_ni_force<NI_MAXP, NI_MAXT, 0, 0, 0, true>), dim3(blocks), dim3(256), lds, s, a);
        else hipLaunchKernelGGL((annp_ni_force<NI_MAXP, NI_MAXT, 0, 0, 0, false>), dim3(blocks), dim3(256), lds, s, a);
    }
}

}  // namespace annp
