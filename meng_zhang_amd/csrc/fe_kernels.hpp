// fe_kernels.hpp -- Chebyshev (aenet-type) descriptor and chain-rule force kernels
// for the Fe potential of pair_style annp.
//
// Arithmetic restated from annp-gpu-lammps/fe_v2/src/pair_annp.cpp ("fe:" below):
//   cutoff test          fe:143-144      fc, fc'              fe:590-594
//   radial  G_m          fe:633-656      Chebyshev T, T'      fe:596-611
//   angular G_{9+n}      fe:658-695      d cos(theta)/dx      fe:618-628
//   force assembly       fe:190-213
// The reference materialises dG/dx for every (neighbour, function) pair and
// contracts it with dE/dG afterwards.  Here the contraction is done first
// (pass 2 knows dE/dG from the network pass), so nothing per-pair is ever stored:
//   pass 1 (annp_fe_desc):   G_n            = sum over neighbours / pairs
//   pass 2 (annp_fe_force):  F_a = -sum_n c_n dG_n/dx_a   with c_n = e_scale s_n dE/dG_n
//
// Work decomposition (both passes): one 64-lane wave per central atom.  The n
// in-cutoff neighbours are compacted into LDS records; the n(n-1)/2 unordered
// pairs are enumerated as a round-robin tournament: pair (a, a+t mod n) for
// t = 1..floor(n/2).  A row a is cut into Q=4 chunks of t, giving 4n work items
// dealt round-robin to lanes, so at any step the 64 lanes hold 64 different
// partners b (conflict-free LDS reads, and in pass 2 collision-free LDS atomics).
#pragma once
#include "annp_common.hpp"

#ifndef ANNP_VARIANT
#define ANNP_VARIANT 0
#endif

namespace annp {

constexpr int FE_Q = 4;  // chunks per tournament row

struct FeArgs {
    int inum;
    int n_cap;                 // LDS record capacity per wave (null record sits at index n_cap)
    const int *ilist;          // nullable: identity
    const double *x;           // [nall][3]
    const int *numneigh;       // by atom index
    const long long *first;    // by atom index
    const int *neigh;
    double cutsq;              // LAMMPS cutsq (cutmax^2)
    double rc_list;            // sqrt(cutsq), used in fc       (fe:150)
    double rc_par;             // params.cut, used in x = 2r/Rc-1 (fe:637,643)
    double *G;                 // [inum][ANNP_GPAD] raw sums (pass 1 out)
    const double *coef;        // [inum][ANNP_CPAD] (pass 2 in)
    double *f;                 // [nall][3] accumulated
    double *virial;            // nullable, 6 doubles accumulated
    int *ncount;               // nullable [inum]: in-cutoff neighbour count
    int *errflag;              // device int: max n seen when n > n_cap
};

// bytes of LDS one wave needs
__host__ __device__ inline size_t fe_desc_lds_per_wave(int n_cap)
{
    size_t rec = (size_t)(n_cap + 1) * 32;
    return rec < 4096 ? 4096 : rec;
}
constexpr int FE_DUMP = 16;   // dump slots behind the records: where masked-off pair steps scatter to
__host__ __device__ inline size_t fe_force_lds_per_wave(int n_cap)
{
    return (size_t)(n_cap + FE_DUMP) * (32 + 40 + 24 + 8);
}

// ---------------------------------------------------------------------------------
// pass 1: descriptor
// ---------------------------------------------------------------------------------
template <int NP, int NT>
__global__ __launch_bounds__(256) void annp_fe_desc(FeArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = lane_id();
    const int wave = uniform(threadIdx.x >> 6);
    const int ii = uniform(blockIdx.x * ANNP_WAVES_PER_BLOCK + wave);
    if (ii >= p.inum) return;
    const int n1 = p.n_cap + 1;
    unsigned char *wbase = lds_raw + (size_t)wave * fe_desc_lds_per_wave(p.n_cap);
    double2 *recA = reinterpret_cast<double2 *>(wbase);          // e.x e.y
    double2 *recB = recA + n1;                                    // e.z fc
    double *scratch = reinterpret_cast<double *>(wbase);          // reused after the pair loop

    const int i = p.ilist ? p.ilist[ii] : ii;
    const double xi = p.x[3 * (size_t)i], yi = p.x[3 * (size_t)i + 1], zi = p.x[3 * (size_t)i + 2];
    const long long base = p.first[i];
    const int jn = p.numneigh[i];
    const double pi_over_rc = ANNP_MY_PI / p.rc_list;
    const double two_over_rcp = 2.0 / p.rc_par;

    // ---- stage A: compact in-cutoff neighbours into LDS, radial sums on the fly
    double gr[NP];
#pragma unroll
    for (int m = 0; m < NP; m++) gr[m] = 0.0;
    int n = 0;
    for (int c0 = 0; c0 < jn; c0 += 64) {
        const int jj = c0 + lane;
        const bool valid = jj < jn;
        const int j = valid ? (p.neigh[base + jj] & ANNP_NEIGHMASK) : i;
        const double dx = xi - p.x[3 * (size_t)j], dy = yi - p.x[3 * (size_t)j + 1], dz = zi - p.x[3 * (size_t)j + 2];
        const double rsq = dx * dx + dy * dy + dz * dz;
        const bool in = valid && !(rsq > p.cutsq) && !(rsq < 1.0e-12);
        const unsigned long long m = __ballot(in);
        const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
        if (in && pos < p.n_cap) {
            const double r = sqrt(rsq);
            const double rinv = 1.0 / r;
            double fc, dfc;
            cutoff_fc(r, pi_over_rc, fc, dfc);
            recA[pos] = make_double2(dx * rinv, dy * rinv);
            recB[pos] = make_double2(dz * rinv, fc);
            // radial Chebyshev, x = 2r/Rc - 1
            const double xr = r * two_over_rcp - 1.0;
            const double y2 = 2.0 * xr;
            double tm2 = 1.0, tm1 = xr;
            gr[0] += fc;
            if (NP > 1) gr[1] = fma(xr, fc, gr[1]);
#pragma unroll
            for (int mm = 2; mm < NP; mm++) {
                const double t = fma(y2, tm1, -tm2);
                gr[mm] = fma(t, fc, gr[mm]);
                tm2 = tm1; tm1 = t;
            }
        }
        n += __popcll(m);
    }
    n = uniform(n);
    if (p.ncount && lane == 0) p.ncount[ii] = n;
    if (n > p.n_cap) {                      // capacity exceeded: report, leave G zero
        if (lane == 0) atomicMax(p.errflag, n);
        if (lane < ANNP_GPAD) p.G[(size_t)ii * ANNP_GPAD + lane] = 0.0;
        return;
    }
    if (lane == 0) {                        // null record: zero weight
        recA[p.n_cap] = make_double2(0.0, 0.0);
        recB[p.n_cap] = make_double2(0.0, 0.0);
    }
    wave_lds_sync();

    // ---- stage B: angular sums over the tournament
    double ga[NT];
#pragma unroll
    for (int m = 0; m < NT; m++) ga[m] = 0.0;
    const int H = n >> 1;
    const int L = (H + FE_Q - 1) / FE_Q;
    const int nitems = n * FE_Q;
    const bool even = (n & 1) == 0;
    // item it -> (row a = it mod n, chunk q = it div n): the 64 lanes of a step hold
    // consecutive rows of one chunk, hence consecutive, distinct partners b
    int a = lane, q = 0;
    while (a >= n && q < FE_Q) { a -= n; q++; }
    for (int it0 = 0; it0 < nitems; it0 += 64) {
        const bool act = q < FE_Q;
        const int t0 = 1 + q * L;
        int t1 = min(H, t0 + L - 1);
        if (even && a >= H && t1 == H) t1 = H - 1;
        const int smax = act ? (t1 - t0) : -1;
        const int ar = act ? a : p.n_cap;
        const double2 A0 = recA[ar], A1 = recB[ar];
        int b = a + t0;
        if (b >= n) b -= n;
        int bi = (0 <= smax) ? b : p.n_cap;
        double2 B0 = recA[bi], B1 = recB[bi];
        for (int s = 0; s < L; ++s) {
            // prefetch the next partner while this one is being worked on
            b++;
            if (b == n) b = 0;
            const int bn = (s + 1 <= smax) ? b : p.n_cap;
            const double2 N0 = recA[bn], N1 = recB[bn];
            const double c = fma(A1.x, B1.x, fma(A0.y, B0.y, A0.x * B0.x));
            const double w = A1.y * B1.y;
            const double y = c + 1.0;            // 2x with x = (cos+1)/2  (fe:671)
            const double x1 = 0.5 * y;
            double tm2 = 1.0, tm1 = x1;
            ga[0] += w;
            if (NT > 1) ga[1] = fma(x1, w, ga[1]);
#pragma unroll
            for (int mm = 2; mm < NT; mm++) {
                const double t = fma(y, tm1, -tm2);
                ga[mm] = fma(t, w, ga[mm]);
                tm2 = tm1; tm1 = t;
            }
            B0 = N0; B1 = N1;
        }
        a += 64;
        while (a >= n && q < FE_Q) { a -= n; q++; }
    }
    wave_lds_sync();    // records are dead from here; the area becomes reduction scratch

    // ---- reduce 64 lane-partials of NP+NT sums, 8 sums per round through LDS
    double *Gout = p.G + (size_t)ii * ANNP_GPAD;
    constexpr int NS = NP + NT;
#pragma unroll
    for (int c8 = 0; c8 < (NS + 7) / 8; c8++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int m = c8 * 8 + k;
            double v = 0.0;
            if (m < NP) v = gr[m < NP ? m : 0];
            else if (m < NS) v = ga[(m - NP) < NT && (m - NP) >= 0 ? (m - NP) : 0];
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
            if (part == 0 && m < NS) Gout[m] = s;
        }
        wave_lds_sync();
    }
    if (lane >= NS && lane < ANNP_GPAD) Gout[lane] = 0.0;
}

// ---------------------------------------------------------------------------------
// pass 2: forces
//   coef[ii]: [0,NP)            c_m                 radial weights
//             [NP, NP+NT)       p_0..p_{NT-1}       P(z) = sum_n c_{NP+n} T_n((z+1)/2) = sum_k p_k z^k
//             [NP+NT, NP+2NT-1) d_0..d_{NT-2}       dP/dz = sum_k d_k z^k
//   with z = cos(theta_jik); written by the network pass (mlp_kernels.hpp epilogue)
// ---------------------------------------------------------------------------------
template <int NP, int NT, bool VIRIAL>
__global__ __launch_bounds__(256) void annp_fe_force(FeArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = lane_id();
    const int wave = uniform(threadIdx.x >> 6);
    const int ii = uniform(blockIdx.x * ANNP_WAVES_PER_BLOCK + wave);
    if (ii >= p.inum) return;
    const int n1 = p.n_cap + FE_DUMP;
    unsigned char *wbase = lds_raw + (size_t)wave * fe_force_lds_per_wave(p.n_cap);
    double2 *recA = reinterpret_cast<double2 *>(wbase);          // e.x e.y
    double2 *recB = recA + n1;                                    // e.z fc
    double *accV0 = reinterpret_cast<double *>(recB + n1);        // sum alpha e_b .x
    double *accV1 = accV0 + n1;
    double *accV2 = accV1 + n1;
    double *accS = accV2 + n1;                                    // sum P fc_b
    double *accC = accS + n1;                                     // sum alpha cos
    double *auxRinv = accC + n1;
    double *auxDfc = auxRinv + n1;
    double *auxR = auxDfc + n1;                                   // radial dE/dr
    int *auxJ = reinterpret_cast<int *>(auxR + n1);

    const int i = p.ilist ? p.ilist[ii] : ii;
    const double xi = p.x[3 * (size_t)i], yi = p.x[3 * (size_t)i + 1], zi = p.x[3 * (size_t)i + 2];
    const long long base = p.first[i];
    const int jn = p.numneigh[i];
    const double pi_over_rc = ANNP_MY_PI / p.rc_list;
    const double two_over_rcp = 2.0 / p.rc_par;
    const double *cf = p.coef + (size_t)ii * ANNP_CPAD;

    double cr[NP];
#pragma unroll
    for (int m = 0; m < NP; m++) cr[m] = cf[m];

    // ---- stage A
    int n = 0;
    for (int c0 = 0; c0 < jn; c0 += 64) {
        const int jj = c0 + lane;
        const bool valid = jj < jn;
        const int j = valid ? (p.neigh[base + jj] & ANNP_NEIGHMASK) : i;
        const double dx = xi - p.x[3 * (size_t)j], dy = yi - p.x[3 * (size_t)j + 1], dz = zi - p.x[3 * (size_t)j + 2];
        const double rsq = dx * dx + dy * dy + dz * dz;
        const bool in = valid && !(rsq > p.cutsq) && !(rsq < 1.0e-12);
        const unsigned long long m = __ballot(in);
        const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
        if (in && pos < p.n_cap) {
            const double r = sqrt(rsq);
            const double rinv = 1.0 / r;
            double fc, dfc;
            cutoff_fc(r, pi_over_rc, fc, dfc);
            recA[pos] = make_double2(dx * rinv, dy * rinv);
            recB[pos] = make_double2(dz * rinv, fc);
            // radial: R = sum_m c_m (T'_m 2/Rc fc + T_m fc')    (fe:648)
            const double xr = r * two_over_rcp - 1.0;
            const double y2 = 2.0 * xr;
            double tm2 = 1.0, tm1 = xr, dm2 = 0.0, dm1 = 1.0;
            double st = cr[0], sd = 0.0;              // sum c T, sum c T'
            if (NP > 1) { st = fma(cr[1], xr, st); sd = cr[1]; }
#pragma unroll
            for (int mm = 2; mm < NP; mm++) {
                const double t = fma(y2, tm1, -tm2);
                const double d = fma(y2, dm1, fma(2.0, tm1, -dm2));
                st = fma(cr[mm], t, st);
                sd = fma(cr[mm], d, sd);
                tm2 = tm1; tm1 = t; dm2 = dm1; dm1 = d;
            }
            auxR[pos] = fma(sd * two_over_rcp, fc, st * dfc);
            auxRinv[pos] = rinv;
            auxDfc[pos] = dfc;
            auxJ[pos] = j;
            accV0[pos] = 0.0; accV1[pos] = 0.0; accV2[pos] = 0.0; accS[pos] = 0.0; accC[pos] = 0.0;
        }
        n += __popcll(m);
    }
    n = uniform(n);
    if (n > p.n_cap) {
        if (lane == 0) atomicMax(p.errflag, n);
        return;
    }
    if (lane == 0) {
        recA[p.n_cap] = make_double2(0.0, 0.0);
        recB[p.n_cap] = make_double2(0.0, 0.0);
        accV0[p.n_cap] = 0.0; accV1[p.n_cap] = 0.0; accV2[p.n_cap] = 0.0; accS[p.n_cap] = 0.0; accC[p.n_cap] = 0.0;
    }
    wave_lds_sync();

    // ---- stage B: pairs
    double ce[NT], cd[NT];
#pragma unroll
    for (int m = 0; m < NT; m++) ce[m] = cf[NP + m];
#pragma unroll
    for (int m = 0; m < NT - 1; m++) cd[m] = cf[NP + NT + m];
    cd[NT - 1] = 0.0;

    const int H = n >> 1;
    const int L = (H + FE_Q - 1) / FE_Q;
    const int nitems = n * FE_Q;
    const bool even = (n & 1) == 0;
    const int dump = p.n_cap + (lane & (FE_DUMP - 1));
    int a = lane, q = 0;
    while (a >= n && q < FE_Q) { a -= n; q++; }
    for (int it0 = 0; it0 < nitems; it0 += 64) {
        const bool act = q < FE_Q;
        const int t0 = 1 + q * L;
        int t1 = min(H, t0 + L - 1);
        if (even && a >= H && t1 == H) t1 = H - 1;
        const int smax = act ? (t1 - t0) : -1;
        const int ar = act ? a : p.n_cap;
        const double2 A0 = recA[ar], A1 = recB[ar];
        double va0 = 0.0, va1 = 0.0, va2 = 0.0, sa = 0.0, ca = 0.0;
        int b = a + t0;
        if (b >= n) b -= n;
        int bi = (0 <= smax) ? b : p.n_cap;
        double2 B0 = recA[bi], B1 = recB[bi];
        // land the first record before the loop: inside it only the prefetch and the
        // (result-less) atomics are in flight, so the loop needs one counted wait per step
        asm volatile("" : "+v"(B0.x), "+v"(B0.y), "+v"(B1.x), "+v"(B1.y));
        for (int s = 0; s < L; ++s) {
            b++;
            if (b == n) b = 0;
            const int bn = (s + 1 <= smax) ? b : p.n_cap;
            const double2 N0 = recA[bn], N1 = recB[bn];      // prefetch, issued first
            __builtin_amdgcn_sched_barrier(0);
            const double c = fma(A1.x, B1.x, fma(A0.y, B0.y, A0.x * B0.x));
            // P(z) and dP/dz by Horner, z = cos(theta)
            double P = ce[NT - 1];
            double Pd = cd[NT - 2];
#pragma unroll
            for (int mm = NT - 2; mm >= 0; mm--) {
                P = fma(P, c, ce[mm]);
                if (mm < NT - 2) Pd = fma(Pd, c, cd[mm]);
            }
            const double w = A1.y * B1.y;          // fc_a fc_b
            const double al = Pd * w;              // (P'/2) fc_a fc_b
            const double alc = al * c;
            va0 = fma(al, B0.x, va0); va1 = fma(al, B0.y, va1); va2 = fma(al, B1.x, va2);
            sa = fma(P, B1.y, sa);
            ca += alc;
#if ANNP_VARIANT != 3
            {   // unconditional: masked-off steps carry zero weight and land in a dump slot,
                // so the scatter needs no branch and the prefetch above keeps a counted wait
                const int bt = (bi == p.n_cap) ? dump : bi;
                atomicAdd(&accV0[bt], al * A0.x);
                atomicAdd(&accV1[bt], al * A0.y);
                atomicAdd(&accV2[bt], al * A1.x);
                atomicAdd(&accS[bt], P * A1.y);
                atomicAdd(&accC[bt], alc);
            }
#else
            asm volatile("" ::"v"(al * A0.x), "v"(al * A0.y), "v"(al * A1.x), "v"(P * A1.y));
#endif
            B0 = N0; B1 = N1; bi = bn;
        }
        {
            const int at = act ? ar : dump;
            atomicAdd(&accV0[at], va0);
            atomicAdd(&accV1[at], va1);
            atomicAdd(&accV2[at], va2);
            atomicAdd(&accS[at], sa);
            atomicAdd(&accC[at], ca);
        }
        a += 64;
        while (a >= n && q < FE_Q) { a -= n; q++; }
    }
    wave_lds_sync();

    // ---- finalize: Fn_a = sum_n c_n dG_n/dx_a ; F_a = -Fn_a to neighbour, +Fn_a to centre (fe:190-213)
    double fi0 = 0.0, fi1 = 0.0, fi2 = 0.0;
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0, v4 = 0.0, v5 = 0.0;
    for (int a = lane; a < n; a += 64) {
        const double2 E0 = recA[a], E1 = recB[a];
        const double rinv = auxRinv[a];
        const double t = fma(accC[a], rinv, -fma(accS[a], auxDfc[a], auxR[a]));
        const double g0 = fma(t, E0.x, -accV0[a] * rinv);
        const double g1 = fma(t, E0.y, -accV1[a] * rinv);
        const double g2 = fma(t, E1.x, -accV2[a] * rinv);
        const int j = auxJ[a];
        atomicAdd(&p.f[3 * (size_t)j], -g0);
        atomicAdd(&p.f[3 * (size_t)j + 1], -g1);
        atomicAdd(&p.f[3 * (size_t)j + 2], -g2);
        fi0 += g0; fi1 += g1; fi2 += g2;
        if (VIRIAL) {   // ev_tally_xyz(i,j,...,fx=-Fj, del = xi-xj = r e)   (fe:201-209)
            const double r = 1.0 / rinv;
            const double d0 = r * E0.x, d1 = r * E0.y, d2 = r * E1.x;
            v0 = fma(d0, g0, v0); v1 = fma(d1, g1, v1); v2 = fma(d2, g2, v2);
            v3 = fma(d0, g1, v3); v4 = fma(d0, g2, v4); v5 = fma(d1, g2, v5);
        }
    }
    fi0 = wave_sum(fi0); fi1 = wave_sum(fi1); fi2 = wave_sum(fi2);
    if (lane == 0) {
        atomicAdd(&p.f[3 * (size_t)i], fi0);
        atomicAdd(&p.f[3 * (size_t)i + 1], fi1);
        atomicAdd(&p.f[3 * (size_t)i + 2], fi2);
    }
    if (VIRIAL) {
        v0 = wave_sum(v0); v1 = wave_sum(v1); v2 = wave_sum(v2);
        v3 = wave_sum(v3); v4 = wave_sum(v4); v5 = wave_sum(v5);
        if (lane == 0) {
            atomicAdd(&p.virial[0], v0); atomicAdd(&p.virial[1], v1); atomicAdd(&p.virial[2], v2);
            atomicAdd(&p.virial[3], v3); atomicAdd(&p.virial[4], v4); atomicAdd(&p.virial[5], v5);
        }
    }
}

}  // namespace annp
