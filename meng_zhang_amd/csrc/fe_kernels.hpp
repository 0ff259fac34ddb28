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
// Work decomposition (both passes): one 64-lane wave per central atom.
//   stage A  the list row is scanned 64 candidates at a time; in-cutoff ones are
//            ballot-compacted into LDS as raw (dx,dy,dz,r^2); a second sweep over the
//            compacted n entries does the expensive per-neighbour math (1/r, fc, fc',
//            radial Chebyshev) once and leaves records (e_x,e_y | e_z,fc).
//   stage B  the n(n-1)/2 unordered pairs are enumerated as a round-robin tournament:
//            pair (a, a+t mod n), t = 1..floor(n/2).  A row a is cut into Q=4 chunks of
//            t; the 4n (row, chunk) items are dealt to lanes so that the 64 lanes of a
//            step hold consecutive rows of one chunk, hence 64 consecutive, distinct
//            partners b: conflict-free ds_read_b128, collision-free LDS atomics.
//            Steps that fall outside a row's range read a null record (fc = 0).
#pragma once
#include "annp_common.hpp"

namespace annp {

constexpr int FE_NP = 9;      // radial Chebyshev orders the kernels are instantiated for (T_0..T_8)
constexpr int FE_NT = 19;     // angular orders (T_0..T_18); smaller bases are embedded with zero weights (annp_hip_init)
constexpr int FE_Q = 4;       // chunks per tournament row
constexpr int FE_REDROW = 72; // row pitch (doubles) of the reduction scratch: rows 16 banks apart, see annp_fe_desc
constexpr int FE_DUMP = 4;    // null / dump slots behind the records: where masked-off pair steps read and scatter

struct FeArgs {
    int inum;
    int n_cap;                 // LDS record capacity per wave (null record sits at index n_cap)
    const int *ilist;          // nullable: identity
    const double *x;           // [nall][3]
    const int *type;           // nullable [nall]: LAMMPS types, given when some type is unmapped (see `active`)
    unsigned active;           // bit t set: atoms of type t take part (map[t] >= 0, i.e. cutsq[.][t] > 0, fe:144)
    const int *numneigh;       // by atom index
    const long long *first;    // by atom index
    const int *neigh;
    double cutsq;              // LAMMPS cutsq (cutmax^2)
    double rc_list;            // sqrt(cutsq), used in fc       (fe:150)
    double rc_par;             // params.cut, used in x = 2r/Rc-1 (fe:637,643)
    double por_list, two_over_rcp;   // pi / rc_list and 2 / rc_par, divided on the host: a wave-uniform double division is two dozen vector instructions
                               // per wave (a per cent of annp_fe_force_sh, whose four waves per group each paid it) and two register pairs for the whole kernel
    double *G;                 // [inum][ANNP_GPAD] raw sums (pass 1 out)
    const double *coef;        // [inum][ANNP_CPAD] (pass 2 in)
    double *f;                 // [nall][3] accumulated
    double *virial;            // nullable: the evaluation's virial table [ANNP_VSLOTS][8] (annp_common.hpp), accumulated into
    double *vatom;             // nullable, [nall][6] accumulated (needs the VIRIAL kernel variant)
    int *ncount;               // nullable [inum]: in-cutoff neighbour count
    int *nmax_word;            // nullable device int: annp_fe_desc_sh raises it to the largest count it finds (otherwise annp_max_int does, in a launch of its own)
    int *errflag;              // device int: max n seen when n > n_cap and nothing can take the atom over
    // force pass only: atoms whose in-cutoff count exceeds n_cap are queued for annp_fe_force_fixup, which
    // runs them with the list-length capacity on the same stream (nullable: report through errflag instead)
    int *ovf_count;            // device int, zeroed per evaluation
    int *ovf_list;             // [ovf_cap] entries ii
    int ovf_cap;
    // moment form (fe_sh_kernels.hpp): the 361 moments of every atom's neighbourhood, [inum][SH_MPAD]; written by
    // annp_fe_desc_sh when given, read by annp_fe_force_sh
    double *A;
    int *nbrs;                 // nullable [inum][128]: the in-cutoff neighbours of every atom in list order, annp_fe_desc_sh -> annp_fe_force_sh
    int *tab_spills;           // nullable device int: contributions annp_fe_force_sh's force tables had no bucket for
    int shf_places_by_number;  // annp_fe_force_sh (developer A/B switch): the one-slot wave of a group by wave number, not by SIMD
    int chk_nall;              // ... (developer build -DANNP_SHF_CHECK) nall, to check indices against
};

// LDS layout of one wave.  A record is two 16-byte halves (e_x,e_y) and (e_z,fc), kept in two
// parallel arrays recA[slot], recB[slot] so that the 64 lanes of a ds_read_b128 touch 64
// consecutive 16-byte slots (an interleaved 32-byte struct costs a 2-way bank conflict per read,
// and the LDS pipe is the second-busiest unit of the force pass).  Slots:
//   [0,n)            the in-cutoff neighbours
//   [n,n+L)          copies of records 0..L-1, so a run of L consecutive partners never wraps
//   [NZ,NZ+FE_DUMP)  null records (fc = 0); NZ = n_cap + fe_lcap(n_cap)
// The force pass keeps 4 accumulators per record slot, one array per component (consecutive slots = consecutive
// 8-byte words: the 64 distinct targets of a pair step never share a bank).
__host__ __device__ constexpr int fe_lcap(int n_cap) { return (n_cap + 7) / 8; }   // >= ceil(floor(n/2)/FE_Q) for every n <= n_cap
__host__ __device__ constexpr int fe_slots(int n_cap) { return n_cap + fe_lcap(n_cap) + FE_DUMP; }
__host__ __device__ inline size_t fe_desc_lds_per_wave(int n_cap)
{
    size_t rec = (size_t)fe_slots(n_cap) * 32;      // multiple of 16: every wave's base stays b128-aligned
    return rec < 8 * FE_REDROW * 8 + 512 ? 8 * FE_REDROW * 8 + 512 : rec;     // reduction scratch [8][FE_REDROW] + [64] results
}
// AUXREG (n_cap <= 128): a lane owns neighbours lane and lane+64 and keeps their 1/r, fc' and radial term in registers;
// otherwise they live in LDS.  The neighbour index always does (it is written by the compacting lane).
// n_cap = 128 (the capacity of a bcc-Fe box: 112 in-cutoff neighbours + thermal spread) gives 148 slots * 64 B + 512 B
// = 9 984 B per wave: four 4-wave workgroups per CU, 4 waves per SIMD.  (5 accumulators at a 40-byte pitch made it 11.3 KB
// and 3 waves per SIMD.)
__host__ __device__ constexpr bool fe_force_auxreg(int n_cap) { return n_cap <= 128; }
__host__ __device__ constexpr size_t fe_force_lds_per_wave(int n_cap, bool auxreg)
{
    return (((size_t)fe_slots(n_cap) * (32 + 32) + (size_t)n_cap * (auxreg ? 4 : 24 + 4)) + 15) & ~(size_t)15;   // b128-aligned bases
}
__host__ __device__ constexpr size_t fe_force_lds_per_wave(int n_cap) { return fe_force_lds_per_wave(n_cap, fe_force_auxreg(n_cap)); }

// ---- stage A, first sweep: candidates -> compacted raw entries (dx,dy | dz,r^2) [+ index]
template <bool WITH_J>
__device__ __forceinline__ int fe_compact(const FeArgs &p, int i, int lane, double2 *recA, double2 *recB, int *auxJ, const int n_cap)
{
    const double xi = p.x[3 * (size_t)i], yi = p.x[3 * (size_t)i + 1], zi = p.x[3 * (size_t)i + 2];
    const long long base = p.first[i];
    const int jn = p.numneigh[i];
    int n = 0;
    // four 64-candidate groups per trip: index loads, then 12 coordinate gathers, all in flight together
    for (int c0 = 0; c0 < jn; c0 += 256) {
        int j[4];
        bool valid[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int jj = c0 + 64 * u + lane;
            valid[u] = jj < jn;
            // (unconditional, clamped to the row's last entry: a load under `valid ? .. : ..` becomes a branch with its own wait,
            // four dependent round trips through memory instead of one)
            j[u] = p.neigh[base + min(jj, jn - 1)] & ANNP_NEIGHMASK;
        }
        if (p.type) {          // wave-uniform: only potentials with an unmapped type pay for the gather
#pragma unroll
            for (int u = 0; u < 4; u++) { const int tj = p.type[j[u]]; valid[u] = valid[u] & type_mapped(p.active, tj); }      // (unconditional load)
        }
        double dx[4], dy[4], dz[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            dx[u] = xi - p.x[3 * (size_t)j[u]]; dy[u] = yi - p.x[3 * (size_t)j[u] + 1]; dz[u] = zi - p.x[3 * (size_t)j[u] + 2];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const double rsq = dx[u] * dx[u] + dy[u] * dy[u] + dz[u] * dz[u];
            const bool in = valid[u] && !(rsq > p.cutsq) && !(rsq < 1.0e-12);        // fe:144
            const unsigned long long m = __ballot(in);
            const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
            if (in && pos < n_cap) {
                recA[pos] = make_double2(dx[u], dy[u]);
                recB[pos] = make_double2(dz[u], rsq);
                if (WITH_J) auxJ[pos] = j[u];
            }
            n += __popcll(m);
        }
    }
    return uniform(n);
}

// per-neighbour geometry from a raw entry: unit vector e = (xi-xj)/r, r, 1/r, fc, fc'
struct FeNbr { double ex, ey, ez, r, rinv, fc, dfc; };
__device__ __forceinline__ FeNbr fe_geometry(double2 R0, double2 R1, double pi_over_rc)
{
    FeNbr g;
    g.rinv = fast_rsqrt(R1.y);
    g.r = R1.y * g.rinv;
    g.ex = R0.x * g.rinv; g.ey = R0.y * g.rinv; g.ez = R1.x * g.rinv;
    double sn, cs;
    sincos_0_pi(pi_over_rc * g.r, sn, cs);
    g.fc = 0.5 * (cs + 1.0);                    // fe:592
    g.dfc = -0.5 * pi_over_rc * sn;             // fe:593
    return g;
}

// item -> (row a, chunk q) bookkeeping of the tournament, one item per lane per round
struct FeItem {
    int a, q;          // row, chunk (q == FE_Q: no item)
    __device__ __forceinline__ void init(int lane, int n) { a = lane; q = 0; norm(n); }
    __device__ __forceinline__ void next(int n) { a += 64; norm(n); }
    __device__ __forceinline__ void norm(int n) { while (a >= n && q < FE_Q) { a -= n; q++; } }
};

// range of a round: every active lane of it has at least Lm valid steps (uniform value).
// Chunk lengths shrink with q, so the round's largest chunk index decides.
__device__ __forceinline__ int fe_round_min_steps(int q_lane, int H, int L, bool even)
{
    const int q63 = min(__builtin_amdgcn_readlane(q_lane, 63), FE_Q - 1);
    const int lenq = min(L, max(0, H - q63 * L));
    const bool hasH = (q63 * L + lenq == H);
    return max(0, lenq - ((even && hasH) ? 1 : 0));
}

// ---------------------------------------------------------------------------------
// pass 1: descriptor
// ---------------------------------------------------------------------------------
template <int NP, int NT>
__device__ __forceinline__ void fe_desc_atom(const FeArgs &p, const int ii, const int lane, unsigned char *wbase)
{
    static_assert(NT >= 3, "angular closed forms assume at least T_0..T_2");
    double2 *recA = reinterpret_cast<double2 *>(wbase);
    double2 *recB = recA + fe_slots(p.n_cap);
    double *scratch = reinterpret_cast<double *>(wbase);          // reused after the pair loop: [8][64]
    double *red = scratch + 8 * FE_REDROW;                        // [64] reduced sums
    const int NZ = p.n_cap + fe_lcap(p.n_cap);                    // first null record

    const int i = p.ilist ? p.ilist[ii] : ii;
    const double pi_over_rc = p.por_list;
    const double two_over_rcp = p.two_over_rcp;

    if (p.type && !type_mapped(p.active, p.type[i])) {      // centre of an unmapped type: no neighbours, no energy
        if (p.ncount && lane == 0) p.ncount[ii] = 0;
        if (lane < ANNP_GPAD) p.G[(size_t)ii * ANNP_GPAD + lane] = 0.0;
        return;
    }
    const int n = fe_compact<false>(p, i, lane, recA, recB, nullptr, p.n_cap);
    if (p.ncount && lane == 0) p.ncount[ii] = n;
    if (n > p.n_cap) {                      // capacity exceeded: report, leave G zero
        if (lane == 0) atomicMax(p.errflag, n);
        if (lane < ANNP_GPAD) p.G[(size_t)ii * ANNP_GPAD + lane] = 0.0;
        return;
    }
    wave_lds_sync();
    const int H = n >> 1;
    const int L = (H + FE_Q - 1) / FE_Q;

    // ---- stage A, second sweep: geometry, radial sums, and the three neighbour sums that
    //      give the T_0 and T_1 angular functions in closed form:
    //        sum_{a<b} fc_a fc_b            = (S1^2 - S2)/2
    //        sum_{a<b} fc_a fc_b cos(theta) = (|V|^2 - S2)/2,   S1 = sum fc, S2 = sum fc^2, V = sum fc e
    double gr[NP];
#pragma unroll
    for (int m = 0; m < NP; m++) gr[m] = 0.0;
    double s1 = 0.0, s2 = 0.0, vx = 0.0, vy = 0.0, vz = 0.0;
    for (int a = lane; a < n; a += 64) {
        const FeNbr g = fe_geometry(recA[a], recB[a], pi_over_rc);
        const double2 RA = make_double2(g.ex, g.ey), RB = make_double2(g.ez, g.fc);
        recA[a] = RA; recB[a] = RB;
        if (a < L) { recA[n + a] = RA; recB[n + a] = RB; }         // wrap-free copy
        s1 += g.fc; s2 = fma(g.fc, g.fc, s2);
        vx = fma(g.fc, g.ex, vx); vy = fma(g.fc, g.ey, vy); vz = fma(g.fc, g.ez, vz);
        const double xr = g.r * two_over_rcp - 1.0;        // fe:643
        const double y2 = 2.0 * xr;
        double tm2 = 1.0, tm1 = xr;
        gr[0] += g.fc;
        if (NP > 1) gr[1] = fma(xr, g.fc, gr[1]);
#pragma unroll
        for (int mm = 2; mm < NP; mm++) {
            const double t = fma(y2, tm1, -tm2);
            gr[mm] = fma(t, g.fc, gr[mm]);
            tm2 = tm1; tm1 = t;
        }
    }
    if (lane < FE_DUMP) {                   // null records: zero weight
        recA[NZ + lane] = make_double2(0.0, 0.0);
        recB[NZ + lane] = make_double2(0.0, 0.0);
    }
    wave_lds_sync();

    // ---- stage B: angular sums T_2..T_{NT-1} over the tournament
    double ga[NT];
#pragma unroll
    for (int m = 0; m < NT; m++) ga[m] = 0.0;
    const int nitems = n * FE_Q;
    const bool even = (n & 1) == 0;
    FeItem it;
    it.init(lane, n);
    for (int it0 = 0; it0 < nitems; it0 += 64) {
        const bool act = it.q < FE_Q;
        const int t0 = 1 + it.q * L;
        int t1 = min(H, t0 + L - 1);
        if (even && it.a >= H && t1 == H) t1 = H - 1;
        const int smax = act ? (t1 - t0) : -1;
        const int ar = act ? it.a : NZ;               // idle lanes carry a null record: zero weight
        const double2 A0 = recA[ar], A1 = recB[ar];
        int b = it.a + t0;
        if (b >= n) b -= n;
        if (!act) b = NZ + (lane & (FE_DUMP - 1));
        const double2 *pA = recA + b, *pB = recB + b;   // partner run b, b+1, .. never wraps (copies behind n)
        const int inc = act ? 1 : 0;
        const int Lm = fe_round_min_steps(it.q, H, L, even);   // steps every lane may take unchecked

        auto step = [&](const double2 B0, const double2 B1) {
            const double c = fma(A1.x, B1.x, fma(A0.y, B0.y, A0.x * B0.x));
            const double w = A1.y * B1.y;
            const double y = c + 1.0;            // 2x with x = (cos+1)/2  (fe:671)
            const double x1 = 0.5 * y;
            double tm2 = x1, tm1 = fma(y, x1, -1.0);      // T_1, T_2
            ga[2] = fma(tm1, w, ga[2]);
#pragma unroll
            for (int mm = 3; mm < NT; mm++) {
                const double t = fma(y, tm1, -tm2);
                ga[mm] = fma(t, w, ga[mm]);
                tm2 = tm1; tm1 = t;
            }
        };
        int s = 0;
        if (Lm > 0) {
            double2 B0 = *pA, B1 = *pB;
            for (; s + 1 < Lm; s += 2) {               // two steps per trip, records ping-pong B <-> N
                pA += inc; pB += inc;
                const double2 N0 = *pA, N1 = *pB;
                step(B0, B1);
                pA += inc; pB += inc;
                B0 = *pA; B1 = *pB;
                step(N0, N1);
            }
            if (s < Lm) { step(B0, B1); pA += inc; pB += inc; ++s; }
        }
        for (; s < L; ++s) {                       // ragged tail of the round (at most a few steps)
            const bool ok = s <= smax;
            step(ok ? *pA : recA[NZ], ok ? *pB : recB[NZ]);
            pA += inc; pB += inc;
        }
        it.next(n);
    }
    wave_lds_sync();    // records are dead from here; the area becomes reduction scratch

    // ---- reduce the lane partials, 8 sums per round through LDS.
    //      slots: [0,NP) radial, then angular T_2.. (NT-2 of them), then S1 S2 Vx Vy Vz
    constexpr int NS = NP + (NT - 2) + 5;
    static_assert(NS <= 64, "too many sums");
#pragma unroll
    for (int c8 = 0; c8 < (NS + 7) / 8; c8++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int m = c8 * 8 + k;
            double v = 0.0;
            if (m < NP) v = gr[m < NP ? m : 0];
            else if (m < NP + NT - 2) v = ga[(m - NP + 2) < NT && (m - NP + 2) >= 2 ? (m - NP + 2) : 2];
            else if (m == NP + NT - 2) v = s1;
            else if (m == NP + NT - 1) v = s2;
            else if (m == NP + NT) v = vx;
            else if (m == NP + NT + 1) v = vy;
            else if (m == NP + NT + 2) v = vz;
            scratch[k * FE_REDROW + lane] = v;
        }
        wave_lds_sync();
        {
            // lane (k, part) sums every 8th value of row k: the 8 lanes of a row read 8 consecutive doubles and the
            // rows sit 16 banks apart (pitch 72), a 2-way conflict; contiguous 8-value chunks per lane were 16-way
            const int k = lane >> 3, part = lane & 7;
            const double *src = scratch + k * FE_REDROW + part;
            double sm = 0.0;
#pragma unroll
            for (int u = 0; u < 8; u++) sm += src[8 * u];
            sm += __shfl_xor(sm, 1, 64);
            sm += __shfl_xor(sm, 2, 64);
            sm += __shfl_xor(sm, 4, 64);
            const int m = c8 * 8 + k;
            if (part == 0 && m < NS) red[m] = sm;
        }
        wave_lds_sync();
    }
    double *Gout = p.G + (size_t)ii * ANNP_GPAD;
    if (lane < ANNP_GPAD) {
        double v = 0.0;
        if (lane < NP) v = red[lane];
        else if (lane >= NP + 2 && lane < NP + NT) v = red[lane - 2];
        else if (lane == NP || lane == NP + 1) {
            const double S1 = red[NP + NT - 2], S2 = red[NP + NT - 1];
            const double V0 = red[NP + NT], V1 = red[NP + NT + 1], V2 = red[NP + NT + 2];
            const double g0 = 0.5 * (S1 * S1 - S2);                            // sum w T_0
            const double wc = 0.5 * (fma(V0, V0, fma(V1, V1, V2 * V2)) - S2);   // sum w cos
            v = (lane == NP) ? g0 : 0.5 * (wc + g0);                           // T_1 = (cos+1)/2
        }
        Gout[lane] = v;
    }
}

template <int NP, int NT>
__global__ __launch_bounds__(256) void annp_fe_desc(FeArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    ANNP_POISON();
    const int lane = lane_id();
    const int wave = uniform(threadIdx.x >> 6);
    const int ii = uniform(xcd_block() * (int)(blockDim.x >> 6) + wave);
    if (ii >= p.inum) return;
    fe_desc_atom<NP, NT>(p, ii, lane, lds_raw + (size_t)wave * fe_desc_lds_per_wave(p.n_cap));
}

// ---------------------------------------------------------------------------------
// pass 2: forces
//   coef[ii]: [0,NP)            c_m                 radial weights
//             [NP, NP+NT)       p_0..p_{NT-1}       P(z) = sum_n c_{NP+n} T_n((z+1)/2) = sum_k p_k z^k
//             [NP+NT, NP+2NT-1) d_0..d_{NT-2}       dP/dz = sum_k d_k z^k
//   with z = cos(theta_jik); written by the network pass (mlp_kernels.hpp)
//
//   per pair (a,b):  alpha = dP/dz fc_a fc_b
//     d/dx_a :  alpha (-e_b + z e_a)/r_a  - P fc'_a fc_b e_a          (fe:683 with fe:618-628)
//   so with V_a = sum_b alpha e_b, C_a = sum_b alpha z, S_a = sum_b P fc_b:
//     Fn_a = (-V_a + C_a e_a)/r_a - (S_a fc'_a + R_a) e_a,   R_a = radial dE/dr   (fe:648)
//   (R_a is folded into the start value of C_a: C_a <- -R_a r_a)
//   The lane that evaluates the pair adds the a-side in registers and scatters the b-side
//   (5 doubles) with LDS atomics into the accumulator slot that parallels the record slot
//   (copies behind n have their own accumulators, folded back at the end).
// ---------------------------------------------------------------------------------
// NCAP: record capacity fixed at compile time (128: every offset of the LDS layout becomes an immediate operand of the
// ds instructions and the pair loop advances two address registers), 0 = p.n_cap at run time
template <int NP, int NT, bool VIRIAL, bool AUXREG, int NCAP>
__device__ __forceinline__ void fe_force_atom(const FeArgs &p, const int ii, const int lane, unsigned char *wbase)
{
    const int n_cap = NCAP ? NCAP : p.n_cap;
    const int nslot = fe_slots(n_cap);
    const int NZ = n_cap + fe_lcap(n_cap);
    double2 *recA = reinterpret_cast<double2 *>(wbase);
    double2 *recB = recA + nslot;
    // 4 accumulators per slot, one array per component (slot s of component k at acc[k * nslot + s]):
    //   [0..2] V' = sum dP/dz fc_partner e_partner   [3] S = sum P fc_partner
    double *acc = reinterpret_cast<double *>(recB + nslot);
    int *auxJ = reinterpret_cast<int *>(acc + 4 * (size_t)nslot);
    double *auxRinv = reinterpret_cast<double *>(auxJ + n_cap + (n_cap & 1));       // LDS copies, !AUXREG only
    double *auxDfc = auxRinv + n_cap;
    double *auxRr = auxDfc + n_cap;
    double reg_rinv[2] = {0.0, 0.0}, reg_dfc[2] = {0.0, 0.0}, reg_rr[2] = {0.0, 0.0};   // AUXREG: rows lane, lane+64

    const int i = p.ilist ? p.ilist[ii] : ii;
    const double pi_over_rc = p.por_list;
    const double two_over_rcp = p.two_over_rcp;
    const double *cf = p.coef + (size_t)ii * ANNP_CPAD;

    if (p.type && !type_mapped(p.active, p.type[i])) return;
    const int n = fe_compact<true>(p, i, lane, recA, recB, auxJ, n_cap);
    if (n > n_cap) {            // does not fit this launch's records: hand the atom to the fix-up launch
        if (lane == 0) {
            const int k = p.ovf_list ? atomicAdd(p.ovf_count, 1) : p.ovf_cap;
            if (k < p.ovf_cap) p.ovf_list[k] = ii;
            else atomicMax(p.errflag, n);
        }
        return;
    }
    wave_lds_sync();
    const int H = n >> 1;
    const int L = (H + FE_Q - 1) / FE_Q;
    // accumulators start at zero.  Fn_a = (-V_a + C_a e_a)/r_a - S_a fc'_a e_a  with  C_a = e_a . V_a - R_a r_a;
    // the radial term R_a r_a stays with the lane that owns neighbour a (register, or LDS beyond 128 neighbours)
    for (int k = lane; k < n + L; k += 64) { acc[k] = 0.0; acc[nslot + k] = 0.0; acc[2 * nslot + k] = 0.0; acc[3 * nslot + k] = 0.0; }
    if (lane < FE_DUMP) {
        recA[NZ + lane] = make_double2(0.0, 0.0);
        recB[NZ + lane] = make_double2(0.0, 0.0);
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k * nslot + NZ + lane] = 0.0;
    }
    wave_lds_sync();
    {
        double cr[NP];
#pragma unroll
        for (int m = 0; m < NP; m++) cr[m] = cf[m];
        auto sweep = [&](int a, double &rinv_out, double &dfc_out, double &rr_out) {
            const FeNbr g = fe_geometry(recA[a], recB[a], pi_over_rc);
            const double2 RA = make_double2(g.ex, g.ey), RB = make_double2(g.ez, g.fc);
            recA[a] = RA; recB[a] = RB;
            if (a < L) { recA[n + a] = RA; recB[n + a] = RB; }
            // radial: R = sum_m c_m (T'_m 2/Rc fc + T_m fc')    (fe:648)
            const double xr = g.r * two_over_rcp - 1.0;
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
            const double R = fma(sd * two_over_rcp, g.fc, st * g.dfc);
            rr_out = -R * g.r;
            rinv_out = g.rinv; dfc_out = g.dfc;
        };
        if constexpr (AUXREG) {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int a = lane + 64 * k;
                if (a < n) sweep(a, reg_rinv[k], reg_dfc[k], reg_rr[k]);
            }
        } else {
            for (int a = lane; a < n; a += 64) {
                double ri, df, rr;
                sweep(a, ri, df, rr);
                auxRinv[a] = ri; auxDfc[a] = df; auxRr[a] = rr;
            }
        }
    }
    wave_lds_sync();

    // ---- stage B: pairs
    double ce[NT];
#pragma unroll
    for (int m = 0; m < NT; m++) ce[m] = cf[NP + m];
    // the Horner chain starts from a register, not from a second SGPR operand per step
    double ce_top = ce[NT - 1];
    asm volatile("" : "+v"(ce_top));

    const int nitems = n * FE_Q;
    const bool even = (n & 1) == 0;
    const int dump = NZ + (lane & (FE_DUMP - 1));
    FeItem it;
    it.init(lane, n);
    for (int it0 = 0; it0 < nitems; it0 += 64) {
        const bool act = it.q < FE_Q;
        const int t0 = 1 + it.q * L;
        int t1 = min(H, t0 + L - 1);
        if (even && it.a >= H && t1 == H) t1 = H - 1;
        const int smax = act ? (t1 - t0) : -1;
        const int ar = act ? it.a : dump;             // idle lanes: null record (fc_a = 0 -> they scatter zeros)
        const double2 A0 = recA[ar], A1 = recB[ar];
        const double fa0 = A1.y * A0.x, fa1 = A1.y * A0.y, fa2 = A1.y * A1.x;      // fc_a e_a, constant over the run
        double va0 = 0.0, va1 = 0.0, va2 = 0.0, sa = 0.0;
        int b = it.a + t0;
        if (b >= n) b -= n;
        if (!act) b = dump;
        // the run b, b+1, .. never wraps: records and accumulators have copies behind n
        const double2 *pA = recA + b, *pB = recB + b;
        double *qb = acc + b;
        const int incr = act ? 1 : 0, inca = incr;
        const int Lm = fe_round_min_steps(it.q, H, L, even);

        // one pair: evaluate, keep the a-side, scatter the b-side to accumulator slot q
        auto step = [&](const double2 B0, const double2 B1, double *q) {
            const double c = fma(A1.x, B1.x, fma(A0.y, B0.y, A0.x * B0.x));
            // P(z) and dP/dz by Horner, z = cos(theta)
            // (the derivative rides on the same coefficients: d <- d z + p, p <- p z + c_k; rounds 1-3 took (k+1) p_(k+1) from the
            // coefficient row, where the network pass now leaves W_l for the force pass on the moments)
            double Pd = ce_top;
            double P = fma(ce_top, c, ce[NT - 2]);
#pragma unroll
            for (int mm = NT - 3; mm >= 0; mm--) {
                Pd = fma(Pd, c, P);
                P = fma(P, c, ce[mm]);
            }
            // alpha = dP/dz fc_a fc_b.  Every contribution to a target's V lacks exactly that target's own fc
            // (a-side: fc_a, b-side: fc_b), so the accumulators hold V' = sum dP/dz fc_partner e_partner and
            // finish() multiplies by the target's fc once.  C = sum alpha cos(theta) is not accumulated at all:
            // cos = e_a . e_b, so C_a = e_a . V_a.
            const double be = Pd * B1.y;           // dP/dz fc_b
            va0 = fma(be, B0.x, va0); va1 = fma(be, B0.y, va1); va2 = fma(be, B1.x, va2);
            sa = fma(P, B1.y, sa);
            atomicAdd(q, Pd * fa0);                // dP/dz fc_a e_a
            atomicAdd(q + nslot, Pd * fa1);
            atomicAdd(q + 2 * nslot, Pd * fa2);
            atomicAdd(q + 3 * nslot, P * A1.y);
        };

        int s = 0;
        if (Lm > 0) {
            double2 B0 = *pA, B1 = *pB;
            // land the first record before the loop: inside it only the prefetch and the
            // (result-less) atomics are in flight, so the loop needs one counted wait per step
            asm volatile("" : "+v"(B0.x), "+v"(B0.y), "+v"(B1.x), "+v"(B1.y));
            for (; s + 1 < Lm; s += 2) {                   // two steps per trip: records ping-pong B <-> N
                pA += incr; pB += incr;
                const double2 N0 = *pA, N1 = *pB;              // prefetch, issued first
                __builtin_amdgcn_sched_barrier(0);
                step(B0, B1, qb);
                qb += inca;
                pA += incr; pB += incr;
                B0 = *pA; B1 = *pB;
                __builtin_amdgcn_sched_barrier(0);
                step(N0, N1, qb);
                qb += inca;
            }
            if (s < Lm) {                                  // odd count: one more, record already here
                step(B0, B1, qb);
                qb += inca; pA += incr; pB += incr;
                ++s;
            }
        }
        for (; s < L; ++s) {                       // ragged tail of the round
            const bool ok = s <= smax;
            step(ok ? *pA : recA[NZ], ok ? *pB : recB[NZ], ok ? qb : acc + dump);
            pA += incr; pB += incr; qb += inca;
        }
        {
            double *q = acc + ar;
            atomicAdd(q, va0);
            atomicAdd(q + nslot, va1);
            atomicAdd(q + 2 * nslot, va2);
            atomicAdd(q + 3 * nslot, sa);
        }
        it.next(n);
    }
    wave_lds_sync();
    // fold the accumulators of the copies back onto their originals
    for (int k = lane; k < 4 * L; k += 64) { const int c = k / L, r = k - c * L; acc[c * nslot + r] += acc[c * nslot + n + r]; }
    wave_lds_sync();

    // ---- finalize: Fn_a = sum_n c_n dG_n/dx_a ; F_a = -Fn_a to neighbour, +Fn_a to centre (fe:190-213)
    double fi0 = 0.0, fi1 = 0.0, fi2 = 0.0;
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0, v4 = 0.0, v5 = 0.0;
    auto finish = [&](int a, double rinv, double dfc, double rr) {
        const double2 E0 = recA[a], E1 = recB[a];
        const double *q = acc + a;
        // V_a = fc_a V'_a;  C_a = sum_b alpha_ab cos(theta_ab) = e_a . V_a, plus the radial term -R r
        const double V0 = E1.y * q[0], V1 = E1.y * q[nslot], V2 = E1.y * q[2 * nslot];
        const double cq = fma(E0.x, V0, fma(E0.y, V1, fma(E1.x, V2, rr)));
        const double t = fma(cq, rinv, -q[3 * nslot] * dfc);
        const double g0 = fma(t, E0.x, -V0 * rinv);
        const double g1 = fma(t, E0.y, -V1 * rinv);
        const double g2 = fma(t, E1.x, -V2 * rinv);
        const int j = auxJ[a];
        atomicAdd(&p.f[3 * (size_t)j], -g0);
        atomicAdd(&p.f[3 * (size_t)j + 1], -g1);
        atomicAdd(&p.f[3 * (size_t)j + 2], -g2);
        fi0 += g0; fi1 += g1; fi2 += g2;
        if (VIRIAL) {   // ev_tally_xyz(i,j,...,fx=-Fj, del = xi-xj = r e)   (fe:201-209)
            const double r = 1.0 / rinv;
            const double d0 = r * E0.x, d1 = r * E0.y, d2 = r * E1.x;
            const double w0 = d0 * g0, w1 = d1 * g1, w2 = d2 * g2, w3 = d0 * g1, w4 = d0 * g2, w5 = d1 * g2;
            v0 += w0; v1 += w1; v2 += w2; v3 += w3; v4 += w4; v5 += w5;
            if (p.vatom) {      // ev_tally_xyz, vflag_atom: half of the pair term to j (the other half to i below)
                double *vj = p.vatom + 6 * (size_t)j;
                atomicAdd(vj + 0, 0.5 * w0); atomicAdd(vj + 1, 0.5 * w1); atomicAdd(vj + 2, 0.5 * w2);
                atomicAdd(vj + 3, 0.5 * w3); atomicAdd(vj + 4, 0.5 * w4); atomicAdd(vj + 5, 0.5 * w5);
            }
        }
    };
    if constexpr (AUXREG) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int a = lane + 64 * k;
            if (a < n) finish(a, reg_rinv[k], reg_dfc[k], reg_rr[k]);
        }
    } else {
        for (int a = lane; a < n; a += 64) finish(a, auxRinv[a], auxDfc[a], auxRr[a]);
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
}

// one wave per atom of the launch
template <int NP, int NT, bool VIRIAL, bool AUXREG, int NCAP = 0>
__global__ __launch_bounds__(256) void annp_fe_force(FeArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    ANNP_POISON();
    const int lane = lane_id();
    const int wave = uniform(threadIdx.x >> 6);
    const int ii = uniform(xcd_block() * (int)(blockDim.x >> 6) + wave);
    if (ii >= p.inum) return;
    fe_force_atom<NP, NT, VIRIAL, AUXREG, NCAP>(p, ii, lane, lds_raw + (size_t)wave * fe_force_lds_per_wave(NCAP ? NCAP : p.n_cap, AUXREG));
}

// Fix-up launch: the atoms the main launch queued (their in-cutoff count exceeded the capacity its LDS records were
// sized for, which comes from the previous evaluation) are evaluated here with p.n_cap = the list-row capacity, which
// no atom can exceed.  Fixed small grid, every wave walks the queue; with an empty queue (the steady state) the
// launch costs a few microseconds and the host never has to look at the count.
template <int NP, int NT, bool VIRIAL>
__global__ __launch_bounds__(256) void annp_fe_force_fixup(FeArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    ANNP_POISON();
    const int lane = lane_id();
    const int wave = uniform(threadIdx.x >> 6);
    const int count = uniform(min(*p.ovf_count, p.ovf_cap));
    unsigned char *wbase = lds_raw + (size_t)wave * fe_force_lds_per_wave(p.n_cap, false);
    FeArgs q = p;
    q.ovf_list = nullptr; q.ovf_cap = 0;          // nothing behind this launch: a second overflow is an error
    const int wpb = blockDim.x >> 6;              // launched with one wave per workgroup: the whole LDS for one list row
    for (int k = blockIdx.x * wpb + wave; k < count; k += gridDim.x * wpb) {
        fe_force_atom<NP, NT, VIRIAL, false, 0>(q, uniform(p.ovf_list[k]), lane, wbase);
        wave_lds_sync();
    }
}

}  // namespace annp
