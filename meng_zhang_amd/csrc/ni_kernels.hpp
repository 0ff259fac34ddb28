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
    const double *sym;          // rad[npsf][3] (eta,Rs,Rc) then ang[ntsf][4] (eta,lambda,zeta,Rc)
    double rc_rad, rc_ang;      // Bohr
    double *G;
    const double *coef;         // [inum][ANNP_CPAD]: c_k = dE/dGhat_k / (sf_max-sf_min)_k
    double *f;
    double *virial;
    int *ncount;
    int *errflag;
};

__host__ __device__ inline size_t ni_lds_per_wave() { return (size_t)NI_NCAP * (6 * 8 + 3 * 8 + 8); }

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

__device__ __forceinline__ double ni_powz(double base, double zeta)
{
    const int zi = (int)zeta;
    if ((double)zi == zeta && zi >= 0 && zi <= 1024) {
        double r = 1.0, bb = base;
        int e = zi;
        while (e) { if (e & 1) r *= bb; bb *= bb; e >>= 1; }
        return r;
    }
    return pow(base, zeta);
}

// Shared by both passes: filter neighbours into LDS.  Record a: xij (3), r, fc_ang, dfc_ang.
struct NiLds {
    double *dx, *dy, *dz, *r, *fc, *dfc;   // [NI_NCAP] each
    double *a0, *a1, *a2;                  // force accumulators
    int *j;
};

__device__ __forceinline__ NiLds ni_carve(unsigned char *wbase)
{
    NiLds L;
    L.dx = reinterpret_cast<double *>(wbase);
    L.dy = L.dx + NI_NCAP; L.dz = L.dy + NI_NCAP; L.r = L.dz + NI_NCAP; L.fc = L.r + NI_NCAP; L.dfc = L.fc + NI_NCAP;
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
    const double pi_over_rc = ANNP_MY_PI / p.rc_ang;
    int n = 0;
    for (int c0 = 0; c0 < jn; c0 += 64) {
        const int jj = c0 + lane;
        const bool valid = jj < jn;
        const int j = valid ? (p.neigh[base + jj] & ANNP_NEIGHMASK) : i;
        const double dx = xi - p.x[3 * (size_t)j], dy = yi - p.x[3 * (size_t)j + 1], dz = zi - p.x[3 * (size_t)j + 2];
        const double r = sqrt(dx * dx + dy * dy + dz * dz);
        const bool in = valid && (r * ANNP_CFLENGTH < rcmax) && r > 0.0;
        const unsigned long long m = __ballot(in);
        const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
        if (in && pos < NI_NCAP) {
            double fc, dfc;
            cutoff_fc(r * ANNP_CFLENGTH, pi_over_rc, fc, dfc);
            L.dx[pos] = dx; L.dy[pos] = dy; L.dz[pos] = dz; L.r[pos] = r; L.fc[pos] = fc; L.dfc[pos] = dfc;
            L.a0[pos] = 0.0; L.a1[pos] = 0.0; L.a2[pos] = 0.0;
            L.j[pos] = j;
        }
        n += __popcll(m);
    }
    return uniform(n);
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
    const double ij = 1.0 / q.rj, ik = 1.0 / q.rk;
    q.ej[0] = xj0 * ij; q.ej[1] = xj1 * ij; q.ej[2] = xj2 * ij;
    q.ek[0] = xk0 * ik; q.ek[1] = xk1 * ik; q.ek[2] = xk2 * ik;
    // xjk = x_j - x_k = xik - xij
    const double g0 = xk0 - xj0, g1 = xk1 - xj1, g2 = xk2 - xj2;
    q.rjk = sqrt(g0 * g0 + g1 * g1 + g2 * g2);
    const double ig = 1.0 / q.rjk;
    q.g[0] = g0 * ig; q.g[1] = g1 * ig; q.g[2] = g2 * ig;
    q.ct = q.ej[0] * q.ek[0] + q.ej[1] * q.ek[1] + q.ej[2] * q.ek[2];
    q.fcj = L.fc[a]; q.fck = L.fc[b]; q.dfcj = L.dfc[a]; q.dfck = L.dfc[b];
    const double rc = p.rc_ang;
    q.ok = (q.rj * ANNP_CFLENGTH < rc) && (q.rk * ANNP_CFLENGTH < rc) && (q.rjk * ANNP_CFLENGTH < rc);   // ni:729
    cutoff_fc(q.rjk * ANNP_CFLENGTH, ANNP_MY_PI / rc, q.fcjk, q.dfcjk);
    return q;
}

// ---------------------------------------------------------------------------------
template <int NP, int NT>
__global__ __launch_bounds__(256) void annp_ni_desc(NiArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = lane_id();
    const int wave = uniform(threadIdx.x >> 6);
    const int ii = uniform(blockIdx.x * ANNP_WAVES_PER_BLOCK + wave);
    if (ii >= p.inum) return;
    unsigned char *wbase = lds_raw + (size_t)wave * ni_lds_per_wave();
    const NiLds L = ni_carve(wbase);
    double *scratch = reinterpret_cast<double *>(wbase);
    const int i = p.ilist ? p.ilist[ii] : ii;
    const int n = ni_stage(p, i, L, lane);
    if (p.ncount && lane == 0) p.ncount[ii] = n;
    double *Gout = p.G + (size_t)ii * ANNP_GPAD;
    if (n > NI_NCAP) {
        if (lane == 0) atomicMax(p.errflag, n);
        if (lane < ANNP_GPAD) Gout[lane] = 0.0;
        return;
    }
    wave_lds_sync();
    const double *srad = p.sym, *sang = p.sym + 3 * p.npsf;

    double gr[NP], ga[NT];
#pragma unroll
    for (int m = 0; m < NP; m++) gr[m] = 0.0;
#pragma unroll
    for (int m = 0; m < NT; m++) ga[m] = 0.0;
    // G2 (ni:686-711): lane a owns neighbour a
    for (int a = lane; a < n; a += 64) {
        const double rm = L.r[a] * ANNP_CFLENGTH;
        if (rm < p.rc_rad) {
            double fc, dfc;
            cutoff_fc(rm, ANNP_MY_PI / p.rc_rad, fc, dfc);
#pragma unroll
            for (int m = 0; m < NP; m++)
                if (m < p.npsf) gr[m] += exp(-srad[3 * m] * rm * rm) * fc;
        }
    }
    // G4 (ni:713-767)
    const int npairs = n * (n - 1) / 2;
    for (int p0 = 0; p0 < npairs; p0 += 64) {
        const int pp = p0 + lane;
        if (pp < npairs) {
            int a, b;
            ni_decode_pair(pp, n, a, b);
            const NiPair q = ni_pair(p, L, a, b);
            if (q.ok) {
                const double rjm = q.rj * ANNP_CFLENGTH, rkm = q.rk * ANNP_CFLENGTH, rgm = q.rjk * ANNP_CFLENGTH;
                const double r2sum = rjm * rjm + rkm * rkm + rgm * rgm;
                const double tfc = q.fcj * q.fck * q.fcjk;
                double ex = 0.0, eta_prev = 0.0;
#pragma unroll
                for (int m = 0; m < NT; m++) {
                    if (m < p.ntsf) {
                        const double eta = sang[4 * m], lam = sang[4 * m + 1], zeta = sang[4 * m + 2];
                        if (m == 0 || eta != eta_prev) { ex = exp(-eta * r2sum); eta_prev = eta; }
                        const double flag = 1.0 + lam * q.ct;
                        if (flag > 0.0) ga[m] += exp2(1.0 - zeta) * ni_powz(flag, zeta) * ex * tfc;
                    }
                }
            }
        }
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
            else if (m - NP < p.ntsf) fidx = p.npsf + (m - NP);
            if (part == 0 && fidx >= 0) Gout[fidx] = s;
        }
        wave_lds_sync();
    }
    if (lane >= nsf && lane < ANNP_GPAD) Gout[lane] = 0.0;
}

// ---------------------------------------------------------------------------------
template <int NP, int NT, bool VIRIAL>
__global__ __launch_bounds__(256) void annp_ni_force(NiArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = lane_id();
    const int wave = uniform(threadIdx.x >> 6);
    const int ii = uniform(blockIdx.x * ANNP_WAVES_PER_BLOCK + wave);
    if (ii >= p.inum) return;
    unsigned char *wbase = lds_raw + (size_t)wave * ni_lds_per_wave();
    const NiLds L = ni_carve(wbase);
    const int i = p.ilist ? p.ilist[ii] : ii;
    const int n = ni_stage(p, i, L, lane);
    if (n > NI_NCAP) { if (lane == 0) atomicMax(p.errflag, n); return; }
    wave_lds_sync();
    const double *srad = p.sym, *sang = p.sym + 3 * p.npsf;
    const double *cf = p.coef + (size_t)ii * ANNP_CPAD;

    const int npairs = n * (n - 1) / 2;
    for (int p0 = 0; p0 < npairs; p0 += 64) {
        const int pp = p0 + lane;
        if (pp < npairs) {
            int a, b;
            ni_decode_pair(pp, n, a, b);
            const NiPair q = ni_pair(p, L, a, b);
            if (q.ok) {
                const double rjm = q.rj * ANNP_CFLENGTH, rkm = q.rk * ANNP_CFLENGTH, rgm = q.rjk * ANNP_CFLENGTH;
                const double r2sum = rjm * rjm + rkm * rkm + rgm * rgm;
                const double tfc = q.fcj * q.fck * q.fcjk;
                double A1 = 0.0, A2 = 0.0, A3 = 0.0;
                double ex = 0.0, eta_prev = 0.0;
#pragma unroll
                for (int m = 0; m < NT; m++) {
                    if (m < p.ntsf) {
                        const double eta = sang[4 * m], lam = sang[4 * m + 1], zeta = sang[4 * m + 2];
                        if (m == 0 || eta != eta_prev) { ex = exp(-eta * r2sum); eta_prev = eta; }
                        const double flag = 1.0 + lam * q.ct;
                        if (flag > 0.0) {
                            const double c = cf[p.npsf + m];
                            const double t3 = exp2(1.0 - zeta) * ni_powz(flag, zeta) * ex;   // term_cot*term_exp
                            A3 = fma(c, t3, A3);
                            A2 = fma(c * eta, t3 * tfc, A2);
                            A1 = fma(c * lam * zeta, t3 * tfc / flag, A1);
                        }
                    }
                }
                A1 *= (1.0 / ANNP_CFLENGTH);
                const double rx = p.compat ? rkm : rgm;           // ni:737-738 vs lal_annp.cu:409-414
                const double t3j_a = q.fck * q.dfcj * q.fcjk, t3j_g = q.fck * q.fcj * q.dfcjk;
                const double t3k_a = q.fcj * q.dfck * q.fcjk, t3k_g = q.fcj * q.fck * q.dfcjk;
                const double irj = 1.0 / q.rj, irk = 1.0 / q.rk;
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
    }
    wave_lds_sync();
    double fi0 = 0.0, fi1 = 0.0, fi2 = 0.0;
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0, v4 = 0.0, v5 = 0.0;
    for (int a = lane; a < n; a += 64) {
        double g0 = L.a0[a], g1 = L.a1[a], g2 = L.a2[a];
        const double r = L.r[a], rm = r * ANNP_CFLENGTH;
        const double d0 = L.dx[a], d1 = L.dy[a], d2 = L.dz[a];
        if (rm < p.rc_rad) {                                     // ni:693-709
            double fc, dfc;
            cutoff_fc(rm, ANNP_MY_PI / p.rc_rad, fc, dfc);
            double R = 0.0;
#pragma unroll
            for (int m = 0; m < NP; m++)
                if (m < p.npsf) {
                    const double eta = srad[3 * m];
                    R = fma(cf[m], exp(-eta * rm * rm) * (-fc * 2.0 * eta * rm + dfc), R);
                }
            const double s = -R / r;                              // dr_dj = -xij/rij
            g0 = fma(s, d0, g0); g1 = fma(s, d1, g1); g2 = fma(s, d2, g2);
        }
        const int j = L.j[a];
        atomicAdd(&p.f[3 * (size_t)j], -g0 * ANNP_CFFORCE);       // ni:186-189
        atomicAdd(&p.f[3 * (size_t)j + 1], -g1 * ANNP_CFFORCE);
        atomicAdd(&p.f[3 * (size_t)j + 2], -g2 * ANNP_CFFORCE);
        fi0 += g0; fi1 += g1; fi2 += g2;
        if (VIRIAL) {       // the reference tallies the un-converted force (ni:190-198)
            v0 = fma(d0, g0, v0); v1 = fma(d1, g1, v1); v2 = fma(d2, g2, v2);
            v3 = fma(d0, g1, v3); v4 = fma(d0, g2, v4); v5 = fma(d1, g2, v5);
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
            atomicAdd(&p.virial[0], v0); atomicAdd(&p.virial[1], v1); atomicAdd(&p.virial[2], v2);
            atomicAdd(&p.virial[3], v3); atomicAdd(&p.virial[4], v4); atomicAdd(&p.virial[5], v5);
        }
    }
}

inline int ni_launch_desc(const NiArgs &a, int blocks, hipStream_t s)
{
    const size_t lds = ni_lds_per_wave() * ANNP_WAVES_PER_BLOCK;
    if (a.npsf > NI_MAXP || a.ntsf > NI_MAXT) return -1;
    if (a.npsf <= 3 && a.ntsf <= 24) hipLaunchKernelGGL((annp_ni_desc<3, 24>), dim3(blocks), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((annp_ni_desc<NI_MAXP, NI_MAXT>), dim3(blocks), dim3(256), lds, s, a);
    return 0;
}

inline void ni_launch_force(const NiArgs &a, int blocks, bool virial, hipStream_t s)
{
    const size_t lds = ni_lds_per_wave() * ANNP_WAVES_PER_BLOCK;
    if (a.npsf <= 3 && a.ntsf <= 24) {
        if (virial) hipLaunchKernelGGL((annp_ni_force<3, 24, true>), dim3(blocks), dim3(256), lds, s, a);
        else hipLaunchKernelGGL((annp_ni_force<3, 24, false>), dim3(blocks), dim3(256), lds, s, a);
    } else {
        if (virial) hipLaunchKernelGGL((annp_ni_force<NI_MAXP, NI_MAXT, true>), dim3(blocks), dim3(256), lds, s, a);
        else hipLaunchKernelGGL((annp_ni_force<NI_MAXP, NI_MAXT, false>), dim3(blocks), dim3(256), lds, s, a);
    }
}

}  // namespace annp
