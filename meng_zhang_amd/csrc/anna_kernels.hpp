// anna_kernels.hpp -- pair_style anna_adp: network -> ADP sums -> energy -> forces, one wave per atom.
//
// Arithmetic restated from anna-gpu-lammps/bcc_fe/src/pair_anna_adp.cpp ("adp:" below).  The descriptor of that
// pair style is the Chebyshev one of pair_style annp without normalisation (adp:584-612), so pass 1 is
// annp_fe_desc (fe_kernels.hpp) unchanged.  This file is everything after it (adp:161-280):
//   * the small network G[28] -> 6 -> 6 -> (d2, q2): forward only, its outputs are the decay constants of the
//     dipole and quadrupole functions u(r), w(r); forces treat them as constants, exactly as the reference does;
//   * per-neighbour analytic terms (density, pair repulsion, u, w) and the atom's sums mu (3), lambda (6), rho;
//   * the energy (adp:205-212) and one force per neighbour (adp:215-280), scattered with global atomics.
// There is no pair loop here: cost is O(n) per atom, a few libm-class functions per neighbour, so a plain
// wave-per-atom mapping (lane = neighbour) is enough; the heavy pass is the descriptor.
#pragma once
#include "annp_common.hpp"
#include "mlp_kernels.hpp"

namespace annp {

constexpr int ANNA_MAXL = 6;        // weight layers
constexpr int ANNA_NR = 2;          // neighbours per lane held in registers: in-range neighbours <= 128

struct AnnaArgs {
    int inum, n_cap;                // n_cap: records per wave (multiple of 64, <= 64 * ANNA_NR)
    const int *ilist;
    const double *x;
    const int *numneigh;
    const long long *first;
    const int *neigh;
    double rc;                      // adp:172,219 test r (not r^2) against the file's cutoff
    const double *G;                // [inum][ANNP_GPAD] raw descriptor sums (pass 1)
    const double *net;              // per layer: W row-major [rows][cols], then B [rows]; layer 0 in the device layout
    int net_doubles;                // size of `net`
    int net_in_lds;                 // 1: the block keeps a copy of `net` in LDS (it fits ANNA_NET_LDS_MAX doubles)
    int nl, nin, nnod, nout;        // nin = columns of layer 0 (device layout)
    unsigned actp;                  // activation flag of layer l in bits 4l..4l+3 (an int[ANNA_MAXL] indexed by the running layer
                                    // makes the compiler copy the argument block to scratch: 132 B per lane)
    double gp[17];                  // A0 yy gamma C0 c1F c2F V0 b1 b2 delta r0 r1 hc d1 q1 d3 q3
    double e_base;
    double *f, *eatom, *eng, *virial, *vatom;
    int *errflag;
};

constexpr int ANNA_NET_LDS_MAX = 4096;     // doubles: 32 KB
// Forces go to global memory through a small wave-private table keyed by atom index: a wave takes ANNA_RUN consecutive
// atoms, whose neighbourhoods overlap heavily (8 atoms of a bcc row: 464 neighbour references, ~115 distinct atoms), adds
// every contribution into the table with LDS atomics and flushes one global atomic per distinct atom and component at the
// end of the run.  The scattered global atomics were 2.8 of the kernel's 6.7 ms at 1 M atoms.
constexpr int ANNA_RUN = 8;         // consecutive atoms per table flush
constexpr int ANNA_TSLOTS = 256;    // table slots per wave (open addressing, linear probing)
constexpr int ANNA_TPROBE = 8;      // probes before a contribution goes straight to global memory
__host__ __device__ inline size_t anna_lds_per_wave(int n_cap)
{
    return (size_t)n_cap * (4 * 8 + 4) + 2 * 64 * 8 + (size_t)ANNA_TSLOTS * (3 * 8 + 4);
}
__host__ __device__ inline size_t anna_lds_net(int net_doubles) { return net_doubles <= ANNA_NET_LDS_MAX ? ((size_t)net_doubles * 8 + 15) / 16 * 16 : 0; }

// exp(x) without libm's special-case branches: x = k ln2 + r, degree-13 Taylor on |r| <= ln2/2 (truncation 4e-18),
// scaled by 2^k.  Arguments here are a few tens at most; the clamp only keeps garbage finite.
__device__ __forceinline__ double exp_fast(double x)
{
    x = fmin(fmax(x, -700.0), 700.0);
    const double kf = rint(x * 1.4426950408889634074);
    double r = fma(-kf, 6.93147180369123816490e-01, x);
    r = fma(-kf, 1.90821492927058770002e-10, r);
    double q = 1.0 / 6227020800.0;
    q = fma(q, r, 1.0 / 479001600.0);
    q = fma(q, r, 1.0 / 39916800.0);
    q = fma(q, r, 1.0 / 3628800.0);
    q = fma(q, r, 1.0 / 362880.0);
    q = fma(q, r, 1.0 / 40320.0);
    q = fma(q, r, 1.0 / 5040.0);
    q = fma(q, r, 1.0 / 720.0);
    q = fma(q, r, 1.0 / 120.0);
    q = fma(q, r, 1.0 / 24.0);
    q = fma(q, r, 1.0 / 6.0);
    q = fma(q, r, 0.5);
    q = fma(q, r, 1.0);
    q = fma(q, r, 1.0);
    return __builtin_ldexp(q, (int)kf);
}

// 1/x to ~1 ulp: hardware estimate + two Newton steps
__device__ __forceinline__ double rcp_fast(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    y = fma(y, fma(-x, y, 1.0), y);
    y = fma(y, fma(-x, y, 1.0), y);
    return y;
}

// adp:607-631 (3 and 4 are the same function here: 1.7 tanh(0.3 a))
__device__ __forceinline__ double anna_act(int flag, double a)
{
    if (flag == 0) return a;
    if (flag == 1) return tanh_fast(a);
    if (flag == 2) return 1.0 / (1.0 + exp_fast(a));
    return 1.7 * tanh_fast(0.3 * a);
}

template <bool VIRIAL>
__global__ __launch_bounds__(256) void annp_anna_adp(AnnaArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    ANNP_POISON();
    const int lane = lane_id();
    const int wave = uniform(threadIdx.x >> 6);
    const int cap = p.n_cap;
    // the network's weights: a per-lane dot product reads one weight per step, and from global memory every one of
    // those steps is a full memory round trip (40 in a row for 28-6-6-2); the block keeps them in LDS instead.
    // Every wave writes the same values, so no ordering between waves is needed.
    double *Lnet = reinterpret_cast<double *>(lds_raw);
    if (p.net_in_lds)
        for (int idx = lane; idx < p.net_doubles; idx += 64) Lnet[idx] = p.net[idx];
    unsigned char *wbase = lds_raw + (p.net_in_lds ? anna_lds_net(p.net_doubles) : 0) + (size_t)wave * anna_lds_per_wave(cap);
    double *Ldx = reinterpret_cast<double *>(wbase), *Ldy = Ldx + cap, *Ldz = Ldy + cap, *Lr = Ldz + cap;
    double *hbuf0 = Lr + cap;                                  // [2][64] network activations
    double *Tacc = hbuf0 + 128;                                // [ANNA_TSLOTS][3] force sums of the run
    int *Lj = reinterpret_cast<int *>(Tacc + 3 * ANNA_TSLOTS);
    int *Tkey = Lj + cap;                                      // [ANNA_TSLOTS] atom index, -1 = free
    for (int sl = lane; sl < ANNA_TSLOTS; sl += 64) { Tkey[sl] = -1; Tacc[3 * sl] = 0.0; Tacc[3 * sl + 1] = 0.0; Tacc[3 * sl + 2] = 0.0; }
    // add (fx, fy, fz) to atom j's entry; lanes of the wave may insert at the same time (distinct j or not)
    auto table_add = [&](int j, double fx, double fy, double fz) {
        unsigned sl = ((unsigned)j * 0x9E3779B1u) >> 24;
#pragma unroll 1
        for (int probe = 0; probe < ANNA_TPROBE; probe++) {
            const int old = atomicCAS(&Tkey[sl], -1, j);
            if (old == -1 || old == j) {
                atomicAdd(&Tacc[3 * sl], fx); atomicAdd(&Tacc[3 * sl + 1], fy); atomicAdd(&Tacc[3 * sl + 2], fz);
                return;
            }
            sl = (sl + 1) & (ANNA_TSLOTS - 1);
        }
        atomicAdd(&p.f[3 * (size_t)j], fx); atomicAdd(&p.f[3 * (size_t)j + 1], fy); atomicAdd(&p.f[3 * (size_t)j + 2], fz);
    };
    auto table_flush = [&]() {
        wave_lds_sync();
        // (the three components of an atom leave from three neighbouring lanes of one instruction: one memory-side request per
        // line instead of three, fe_shf_kernels.hpp)
        for (int k = lane; k < 3 * ANNA_TSLOTS; k += 64) {
            const int sl = k / 3, c = k - 3 * sl;
            const int j = Tkey[sl];
            if (j >= 0) { atomicAdd(&p.f[3 * (size_t)j + c], Tacc[k]); Tacc[k] = 0.0; }
        }
        wave_lds_sync();
        for (int sl = lane; sl < ANNA_TSLOTS; sl += 64) Tkey[sl] = -1;
        wave_lds_sync();
    };
    // a wave walks over many atoms and adds its energies once at the end: one atomic per atom on the single
    // energy word would serialise the whole launch (1 M same-address atomics ~ 12 ms)
    double e_wave = 0.0;
    wave_lds_sync();
    const int nruns = (p.inum + ANNA_RUN - 1) / ANNA_RUN;
    // What an atom needs from memory before any of its arithmetic can start -- its header (index, row length, row start), its
    // descriptor value for the network and the first 192 entries of its list row -- is requested while the atom before it in
    // the run is worked on: an atom used to begin with four dependent memory round trips (header -> row -> positions, then G)
    // that only the two other waves of the SIMD could cover.  `ahead` says whether a_* hold the atom that is starting.
    bool ahead = false;
    int a_i = 0, a_jn = 0, a_j0 = 0, a_j1 = 0, a_j2 = 0;
    long long a_first = 0;
    double a_g = 0.0;
    for (int run = uniform(blockIdx.x * ANNP_WAVES_PER_BLOCK + wave); run < nruns; run += gridDim.x * ANNP_WAVES_PER_BLOCK) {
    ahead = false;
    for (int ii = run * ANNA_RUN; ii < min(p.inum, (run + 1) * ANNA_RUN); ii++) {
    double *hbuf = hbuf0;
    int i, jn;
    long long rfirst;
    if (ahead) { i = a_i; jn = a_jn; rfirst = a_first; }
    else { i = p.ilist ? p.ilist[ii] : ii; jn = p.numneigh[i]; rfirst = p.first[i]; }
    i = uniform(i); jn = uniform(jn);
    const double xi = p.x[3 * (size_t)i], yi = p.x[3 * (size_t)i + 1], zi = p.x[3 * (size_t)i + 2];
    const int *row = p.neigh + rfirst;
    const bool have_row = ahead;            // a_j0..2 hold row[lane], row[64 + lane], row[128 + lane]
    const double g_in = ahead ? a_g : ((lane < p.nin) ? p.G[(size_t)ii * ANNP_GPAD + lane] : 0.0);
    const int pj0 = a_j0, pj1 = a_j1, pj2 = a_j2;
    const unsigned long long lt = (1ull << lane) - 1ull;

    // ---- in-range neighbours (adp:171-173: r > Rc or r < 1e-12 skipped), compacted in list order
    int n = 0;
    for (int c0 = 0; c0 < jn; c0 += 256) {
        int j[4];
        bool valid[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int jj = c0 + 64 * u + lane;
            valid[u] = jj < jn;
            int raw;
            if (have_row && c0 == 0 && u < 3) raw = u == 0 ? pj0 : (u == 1 ? pj1 : pj2);
            else raw = row[min(jj, jn - 1)];        // (unconditional, clamped: a load under `valid ? .. : ..` becomes a branch with its own wait)
            j[u] = valid[u] ? (raw & ANNP_NEIGHMASK) : 0;
        }
        double dx[4], dy[4], dz[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            dx[u] = xi - p.x[3 * (size_t)j[u]]; dy[u] = yi - p.x[3 * (size_t)j[u] + 1]; dz[u] = zi - p.x[3 * (size_t)j[u] + 2];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const double r = sqrt(dx[u] * dx[u] + dy[u] * dy[u] + dz[u] * dz[u]);      // adp:125 all_xij[jj][3]
            const bool in = valid[u] && !(r > p.rc) && !(r < 1.0e-12);
            const unsigned long long m = __ballot(in);
            const int pos = n + __popcll(m & lt);
            if (in && pos < cap) { Ldx[pos] = dx[u]; Ldy[pos] = dy[u]; Ldz[pos] = dz[u]; Lr[pos] = r; Lj[pos] = j[u]; }
            n += __popcll(m);
        }
    }
    n = uniform(n);
    // the next atom of the run: header and descriptor value now, its row a little later (when the header has arrived)
    ahead = ii + 1 < min(p.inum, (run + 1) * ANNA_RUN);
    if (ahead) {
        a_i = p.ilist ? p.ilist[ii + 1] : ii + 1;
        a_jn = p.numneigh[a_i];
        a_first = p.first[a_i];
        a_g = (lane < p.nin) ? p.G[(size_t)(ii + 1) * ANNP_GPAD + lane] : 0.0;
    }
    if (n > cap) { if (lane == 0) atomicMax(p.errflag, n); ahead = false; continue; }

    // ---- the network (adp:633-667).  A row's dot product is split over P = 2^k lanes (as many as fit 64 / rows)
    //      and folded with xor-shuffles: a 28-term sum is then 4 FMAs + 3 shuffles deep instead of 28 dependent
    //      LDS reads.  Bias added last as dot_add_wxb does (adp:599-604); the order of the terms differs from the
    //      reference's left-to-right sum by rounding only.
    {
        double *hin = hbuf, *hout = hbuf + 64;
        if (lane < p.nin) hin[lane] = g_in;
        wave_lds_sync();
        const double *w = p.net_in_lds ? Lnet : p.net;
        for (int l = 0; l < p.nl; l++) {
            const int nr = (l == p.nl - 1) ? p.nout : p.nnod;
            const int nc = (l == 0) ? p.nin : p.nnod;
            int P = 1;
            while (2 * P * nr <= 64) P *= 2;
            const int r = lane / P, part = lane % P;
            double a = 0.0;
            if (r < nr) {
                const double *wr = w + (size_t)r * nc;
                for (int c = part; c < nc; c += P) a = fma(wr[c], hin[c], a);
            }
            for (int off = P >> 1; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
            if (r < nr && part == 0) hout[r] = anna_act((int)((p.actp >> (4 * l)) & 15u), a + w[(size_t)nr * nc + r]);
            w += (size_t)nr * nc + nr;
            wave_lds_sync();
            double *t = hin; hin = hout; hout = t;
        }
        hbuf = hin;      // the outputs
    }
    const double d2 = hbuf[0], q2 = hbuf[1];                   // adp:162
    if (ahead) {
        const int *rown = p.neigh + a_first;
        a_j0 = (lane < a_jn) ? rown[lane] : 0;
        a_j1 = (64 + lane < a_jn) ? rown[64 + lane] : 0;
        a_j2 = (128 + lane < a_jn) ? rown[128 + lane] : 0;
    }

    const double A0 = p.gp[0], yy = p.gp[1], gamma = p.gp[2], C0 = p.gp[3], c1F = p.gp[4], c2F = p.gp[5], V0 = p.gp[6];
    const double b1 = p.gp[7], b2 = p.gp[8], delta = p.gp[9], r0 = p.gp[10], r1 = p.gp[11], hc = p.gp[12];
    const double d1 = p.gp[13], q1 = p.gp[14], d3 = p.gp[15], q3 = p.gp[16];
    const double rep_coeff = V0 / (b2 - b1);

    // ---- per-neighbour functions, kept in registers for the force loop, and the atom's sums (adp:165-196).
    //      The reference's pow(z, b) are exp(b log z) here (one log serves both repulsive powers, and their
    //      reciprocals are what the formulas use), its divisions are reciprocals: same values to a few ulp.
    double stpf[ANNA_NR], dstpf[ANNA_NR], uterm[ANNA_NR], wterm[ANNA_NR], expz[ANNA_NR], zyy[ANNA_NR], izb1[ANNA_NR], izb2[ANNA_NR];
    double mu0 = 0.0, mu1 = 0.0, mu2 = 0.0, l00 = 0.0, l11 = 0.0, l22 = 0.0, l01 = 0.0, l02 = 0.0, l12 = 0.0, rho = 0.0, rep = 0.0;
    const double inv_hc = 1.0 / hc, inv_r1 = 1.0 / r1;
#pragma unroll
    for (int k = 0; k < ANNA_NR; k++) {
        const int a = lane + 64 * k;
        stpf[k] = dstpf[k] = uterm[k] = wterm[k] = expz[k] = zyy[k] = 0.0; izb1[k] = izb2[k] = 1.0;
        if (a < n) {
            const double r = Lr[a], dx = Ldx[a], dy = Ldy[a], dz = Ldz[a];
            const double sx = (r - p.rc) * inv_hc;
            const double sx2 = sx * sx, sx4 = sx2 * sx2, it1 = rcp_fast(1.0 + sx4);
            stpf[k] = sx4 * it1;
            dstpf[k] = 4.0 * (sx2 * sx) * (it1 * it1) * inv_hc;
            uterm[k] = d1 * exp_fast(-d2 * r);
            wterm[k] = q1 * exp_fast(-q2 * r);
            const double u = stpf[k] * (uterm[k] + d3), w = stpf[k] * (wterm[k] + q3);
            mu0 = fma(u, dx, mu0); mu1 = fma(u, dy, mu1); mu2 = fma(u, dz, mu2);
            l00 = fma(w * dx, dx, l00); l11 = fma(w * dy, dy, l11); l22 = fma(w * dz, dz, l22);
            l01 = fma(w * dx, dy, l01); l02 = fma(w * dx, dz, l02); l12 = fma(w * dy, dz, l12);
            const double rho_z = r - r0;
            expz[k] = exp_fast(-gamma * rho_z);
            zyy[k] = A0 * exp_fast(yy * log(rho_z));
            rho += stpf[k] * (zyy[k] * expz[k] * (1.0 + expz[k]) + C0);
            const double lz = log(r * inv_r1);
            izb1[k] = exp_fast(-b1 * lz); izb2[k] = exp_fast(-b2 * lz);
            rep += stpf[k] * (rep_coeff * (b2 * izb1[k] - b1 * izb2[k]) + delta);
        }
    }
    mu0 = wave_sum(mu0); mu1 = wave_sum(mu1); mu2 = wave_sum(mu2);
    l00 = wave_sum(l00); l11 = wave_sum(l11); l22 = wave_sum(l22);
    l01 = wave_sum(l01); l02 = wave_sum(l02); l12 = wave_sum(l12);
    rho = wave_sum(rho); rep = wave_sum(rep);

    // ---- energy (adp:198-212)
    const double v_i = l00 + l11 + l22;
    const double sum_mu = mu0 * mu0 + mu1 * mu1 + mu2 * mu2;
    const double sum_lam = l00 * l00 + l11 * l11 + l22 * l22 + 2.0 * (l01 * l01 + l02 * l02 + l12 * l12);
    const double f_v = -1.0 / 3.0 * v_i;
    const double e_ang = 0.5 * sum_mu + 0.5 * sum_lam - 1.0 / 6.0 * v_i * v_i;
    const double e_emb = c1F * sqrt(rho) + c2F * rho * rho;
    const double evdwl = 0.5 * rep + e_emb + e_ang + p.e_base;
    if (lane == 0 && p.eatom) p.eatom[i] += evdwl;
    e_wave += evdwl;
    const double demb = 0.5 * c1F / sqrt(rho) + 2.0 * c2F * rho;       // adp:237

    // ---- forces (adp:215-280)
    double fi0 = 0.0, fi1 = 0.0, fi2 = 0.0;
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0, v4 = 0.0, v5 = 0.0;
#pragma unroll
    for (int k = 0; k < ANNA_NR; k++) {
        const int a = lane + 64 * k;
        if (a < n) {
            const double rij = Lr[a], dx = Ldx[a], dy = Ldy[a], dz = Ldz[a];
            const double rho_z = rij - r0;
            const double ga_zyy = zyy[k] * gamma;
            const double d_rho = expz[k] * (1.0 + expz[k]) * (zyy[k] * (dstpf[k] + stpf[k] * yy * rcp_fast(rho_z)) - ga_zyy) + C0 * dstpf[k] -
                                 ga_zyy * expz[k] * expz[k];
            const double d_embed = demb * d_rho;
            const double inv_r = rcp_fast(rij);
            const double drep_t = b2 * b1 * inv_r1;
            const double rep_t1 = rep_coeff * (b2 * izb1[k] - b1 * izb2[k]) + delta;
            const double d_repul = dstpf[k] * rep_t1 + stpf[k] * rep_coeff * (drep_t * (r1 * inv_r) * (izb2[k] - izb1[k]));
            const double adp_u = stpf[k] * (uterm[k] + d3);
            const double adp_w = 2.0 * stpf[k] * (wterm[k] + q3);
            const double d_adp_u = dstpf[k] * (uterm[k] + d3) + stpf[k] * (-d2 * uterm[k]);
            const double d_adp_w = dstpf[k] * (wterm[k] + q3) + stpf[k] * (-q2 * wterm[k]);
            const double lamb1 = d_adp_w * (l00 * dx * dx + l11 * dy * dy + l22 * dz * dz);
            const double lamb2 = d_adp_w * (l01 * dx * dy + l02 * dx * dz + l12 * dy * dz) * 2.0 + lamb1;
            const double df1 = (0.5 * d_repul + d_embed + d_adp_u * (mu0 * dx + mu1 * dy + mu2 * dz) + lamb2) * inv_r;
            const double df3 = f_v * (d_adp_w * rij + adp_w);
            const double fx = df1 * dx + adp_w * (dy * l01 + dz * l02 + dx * l00) + mu0 * adp_u + dx * df3;
            const double fy = df1 * dy + adp_w * (dy * l11 + dz * l12 + dx * l01) + mu1 * adp_u + dy * df3;
            const double fz = df1 * dz + adp_w * (dy * l12 + dz * l22 + dx * l02) + mu2 * adp_u + dz * df3;
            const int j = Lj[a];
            table_add(j, fx, fy, fz);
            fi0 -= fx; fi1 -= fy; fi2 -= fz;
            if (VIRIAL) {       // ev_tally_xyz(i, j, ..., -fx, -fy, -fz, delx, dely, delz), adp:276-278
                const double w0 = -dx * fx, w1 = -dy * fy, w2 = -dz * fz, w3 = -dx * fy, w4 = -dx * fz, w5 = -dy * fz;
                v0 += w0; v1 += w1; v2 += w2; v3 += w3; v4 += w4; v5 += w5;
                if (p.vatom) {
                    double *vj = p.vatom + 6 * (size_t)j;
                    atomicAdd(vj + 0, 0.5 * w0); atomicAdd(vj + 1, 0.5 * w1); atomicAdd(vj + 2, 0.5 * w2);
                    atomicAdd(vj + 3, 0.5 * w3); atomicAdd(vj + 4, 0.5 * w4); atomicAdd(vj + 5, 0.5 * w5);
                }
            }
        }
    }
    fi0 = wave_sum(fi0); fi1 = wave_sum(fi1); fi2 = wave_sum(fi2);
    if (lane == 0) table_add(i, fi0, fi1, fi2);
    if (VIRIAL) {
        v0 = wave_sum(v0); v1 = wave_sum(v1); v2 = wave_sum(v2);
        v3 = wave_sum(v3); v4 = wave_sum(v4); v5 = wave_sum(v5);
        if (lane == 0) {
            if (p.virial) {
                double *vr = virial_row(p.virial);          // (annp_common.hpp: the global virial)
                atomicAdd(&vr[0], v0); atomicAdd(&vr[1], v1); atomicAdd(&vr[2], v2);
                atomicAdd(&vr[3], v3); atomicAdd(&vr[4], v4); atomicAdd(&vr[5], v5);
            }
            if (p.vatom) {
                double *vi = p.vatom + 6 * (size_t)i;
                atomicAdd(vi + 0, 0.5 * v0); atomicAdd(vi + 1, 0.5 * v1); atomicAdd(vi + 2, 0.5 * v2);
                atomicAdd(vi + 3, 0.5 * v3); atomicAdd(vi + 4, 0.5 * v4); atomicAdd(vi + 5, 0.5 * v5);
            }
        }
    }
    wave_lds_sync();        // the next atom reuses the records
    }
    table_flush();
    }
    if (p.eng && lane == 0 && e_wave != 0.0) atomicAdd(p.eng, e_wave);
}

// the walk above wants a bounded grid: enough waves to fill the chip several times over, no more
inline int anna_blocks(int inum)
{
    const int nruns = (inum + ANNA_RUN - 1) / ANNA_RUN;
    return std::max(1, std::min((nruns + ANNP_WAVES_PER_BLOCK - 1) / ANNP_WAVES_PER_BLOCK, 256 * 16));
}

}  // namespace annp
