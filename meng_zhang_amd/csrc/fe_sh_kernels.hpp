// fe_sh_kernels.hpp -- the Chebyshev passes (fe_v2/src/pair_annp.cpp:633-695 and 190-213, "fe:") without a pair loop:
// annp_fe_desc_sh (descriptor pass) and annp_fe_force_sh (force pass, second half of this file).
//
// The angular functions are sums over the n(n-1)/2 neighbour pairs of a polynomial of cos(theta_ab) = e_a . e_b:
//     G_{9+n} = sum_{a<b} fc_a fc_b T_n((e_a.e_b + 1)/2)                                                     (fe:671-678)
// annp_fe_desc (fe_kernels.hpp) evaluates them pair by pair: 6 216 pairs x 40 instructions for 112 neighbours.  A polynomial
// of e_a . e_b separates (Legendre addition theorem):
//     P_l(e_a.e_b) = sum_{m=0..l} kappa_lm [ C_lm(a) C_lm(b) + S_lm(a) S_lm(b) ],
//     C_lm(a) + i S_lm(a) = Pm_{l-m}(z_a) (x_a + i y_a)^m        (Pm_k: d^m P_l / dz^m made monic, a polynomial in z alone)
// so with the 361 moments  A_lm = sum_a fc_a (C_lm(a) + i S_lm(a))  of the neighbourhood
//     pw_l   = sum_{a,b} fc_a fc_b P_l(e_a.e_b) = sum_m kappa_lm |A_lm|^2
//     G_{9+n} = 1/2 ( sum_l q_nl pw_l - sum_a fc_a^2 ),       T_n((z+1)/2) = sum_{l<=n} q_nl P_l(z)
// -- the same numbers (the identities are exact; the constants are exact rationals rounded once, tools/gen_sh_tables.py;
// measured difference to the pair loop: a few 1e-13 relative; 2e-14 of the row's size since round 4b), for 112 x 190 steps
// instead of 6 216 x 17.
// The T_0 / T_1 closed forms of annp_fe_desc are the l = 0, 1 cases of this.
//
// Work decomposition of the descriptor pass: a wave takes FOUR atoms, 16 lanes each; a lane owns neighbours l, l+16, l+32, ..
// of its atom, the first SH_R = 5 in registers (ShRegs), the others in LDS.
//   stage A  headers of the four atoms, the index loads of all four list rows, all coordinate gathers (two dependent round
//            trips through memory for the wave), ballot compaction row by row; every lane turns its raw entries into
//            (e_x,e_y), z, (fc,0) -- the running power fc (x+iy)^m starts at m = 0 -- and sums the radial functions.
//   columns  m = 0..18, unrolled: a lane walks its neighbours and adds z^j (x+iy)^m fc, j = 0..18-m, into its 2(19-m) accumulators
//            (a multiply for the power, two FMAs: round 4b; rounds 3-4 ran the recurrence of the Pm_k here, a multiply and an
//            FMA per step); the power of x+iy is advanced.  Then the accumulators are summed over the atom's 16 lanes -- a
//            transposing butterfly: 16 registers x 16 lanes -> one register whose 16 lanes hold the 16 totals, ~3.6 instructions
//            per total for four atoms at once.  The totals are MONOMIAL moments; they wait in the atom's moment row.
//   basis    after the last column the four rows come back into the LDS the neighbours have left, and twelve rounds of sixteen
//            entries change basis, A_(m+k,m) = sum_j M_kj Mom_jm (sh_legendre): kappa |A|^2 into pw_l, kappa A into the moment
//            row for the force pass.
//   final    lane n of an atom: G_{9+n} from the 19 pw_l (q from constant memory), the radial sums, one 32-double row out.
// An atom with more in-cutoff neighbours than the launch has state for (n_cap, at most SH_CAP_MAX = 160) is queued for
// annp_fe_desc_fixup (the pair-loop kernel with room for a whole list row) instead.
#pragma once
#include <type_traits>

#include "fe_kernels.hpp"
#include "sh_tables.hpp"

namespace annp {

constexpr int SH_GA = 4;          // atoms per wave
constexpr int SH_GL = 16;         // lanes per atom
constexpr int SH_CAP_MAX = 160;   // state slots per atom the launch may ask for (round 6: 128 before -- a bcc cell compressed by 7 % has 136 neighbours inside
                                  // 6.5 A and fell to the pair-loop kernels at half the speed; at 160 the pass keeps two workgroups per CU)
static_assert(SH_LMAX == FE_NT - 1, "tables are generated for T_0..T_18");

__constant__ double annp_sh_q[(SH_LMAX + 1) * (SH_LMAX + 1)] = ANNP_SH_Q_INIT;
// doubles per atom in the moment buffer: (cosine, sine) of (l = m+k, m) at 2 (shf_toff(m) + 18-m-k), sh_tables.hpp; 380 used, the
// 20 behind them only round the row up to 25 whole lines of 128 bytes: nothing writes them after the buffer's allocation (which
// zeroes it), nothing may rely on what they hold (sh_legendre reads the copy in LDS, not the row)
constexpr int SH_MPAD = 400;
__constant__ unsigned short annp_shd_info[SHD_NROUND * 16] = ANNP_SHD_INFO_INIT;
__constant__ double annp_shd_kappa[SHD_NROUND * 16] = ANNP_SHD_KAPPA_INIT;
__constant__ double annp_shd_coef[SHD_TFIRST[SHD_NROUND] * 16] = ANNP_SHD_COEF_INIT;
__host__ __device__ constexpr int sh_apos(int m, int k) { return 2 * (shf_toff(m) + SH_LMAX - m - k); }
// round 6, the GROUPED instantiation (launches with room for at most SHG_CAP_MAX neighbours per atom): the monomial totals of a group of
// columns wait in LDS, not in the moment row (sh_tables.hpp: SHG_*)
__constant__ unsigned short annp_shg_info[SHG_NROUND * 16] = ANNP_SHG_INFO_INIT;
__constant__ double annp_shg_kappa[SHG_NROUND * 16] = ANNP_SHG_KAPPA_INIT;
__constant__ double annp_shg_coef[SHG_TFIRST[SHG_NROUND] * 16] = ANNP_SHG_COEF_INIT;
__host__ __device__ constexpr int shg_group_of(int m) { return m <= SHG_COLS[0][1] ? 0 : m <= SHG_COLS[1][1] ? 1 : 2; }
__host__ __device__ constexpr int shg_group_of_round(int r) { return r < SHG_RFIRST[1] ? 0 : r < SHG_RFIRST[2] ? 1 : 2; }
constexpr int SHG_CAP_MAX = 112;        // bcc Fe inside 6.5 A
constexpr int SHG_PAD = 20;             // entries of zeros behind the last atom's: a lane reads up to 2 (T - 1) = 18 positions beyond its own with zero coefficients

constexpr int SH_R = 5;           // neighbours per lane whose state stays in registers (ShRegs)
constexpr int SH_CAP_MIN = SH_GL * SH_R + 16;

// LDS of one wave, capl = cap - 48 slots per atom: (e_x,e_y)[4][capl+8] | (pc,ps)[4][capl+8] | z[4][capl+8] | 1536 bytes: first the
// staging rows of stage A, then pw[4][20] and the radial totals[4][16].
// cap is a multiple of 16.  The pitch places the four atoms' runs of a lane group of ds_read_b128 ({0-3,12-15,20-27}, .. :
// MI355X_MICROARCH.md, LDS) on the four 64-byte quarters of the bank row (128 mod 256 bytes), and the eight 32-byte runs
// of a 32-lane group of ds_read_b64 on its eight eighths (64 or 192 mod 256 bytes).
__host__ __device__ constexpr int sh_pitch(int capl) { return capl + 8; }
// After the last column the neighbour arrays are dead and the atoms' monomial moments take their place (sh_legendre): 190 (cosine,
// sine) pairs per atom from the wave's first byte, pw and the radial totals in the last 1152 bytes of the wave's share.
constexpr int SH_MOMB = SH_GA * SHF_NE * 16;         // 12 160
constexpr int SH_PWB = (SH_GA * 20 + SH_GA * 16) * 8;
__host__ __device__ constexpr size_t sh_lds_arrays(int n_cap) { return (size_t)SH_GA * sh_pitch(n_cap - SH_GL * SH_R) * 40; }
__host__ __device__ constexpr size_t sh_lds_per_wave(int n_cap)
{
    const size_t a = sh_lds_arrays(n_cap) + SH_GL * SH_R * 32, b = (size_t)SH_MOMB + SH_PWB;
    return a > b ? a : b;
}
static_assert(3 * 4 * sh_lds_per_wave(112) <= 160 * 1024, "bcc Fe (112 neighbours in 6.5 A): three workgroups of four waves per CU");
// GROUPED: the group buffer -- four atoms' entries of the current group, SHG_NENT[g] (cosine, sine) pairs each, back to back, and SHG_PAD
// pairs of zeros -- lies behind the neighbour arrays at their largest (SHG_CAP_MAX), where the staging rows of stage A were, and in front of pw
constexpr int SHG_BUF_OFF = (int)sh_lds_arrays(SHG_CAP_MAX);
constexpr int SHG_BUF_BYTES = (SH_GA * (SHG_NENT[0] > SHG_NENT[1] ? (SHG_NENT[0] > SHG_NENT[2] ? SHG_NENT[0] : SHG_NENT[2]) : (SHG_NENT[1] > SHG_NENT[2] ? SHG_NENT[1] : SHG_NENT[2])) + SHG_PAD) * 16;
static_assert(SHG_BUF_OFF + SHG_BUF_BYTES + SH_PWB <= (int)sh_lds_per_wave(SHG_CAP_MAX), "the group buffer fits between the neighbour arrays and pw");
static_assert(SHG_BUF_OFF >= (int)sh_lds_arrays(SH_CAP_MIN), "... at every capacity the grouped kernel is launched with");
static_assert(SH_GL * SH_R * 32 >= SH_PWB, "pw and the radial totals live where the staging rows were");

// ---- sums over the 16 lanes of an atom ------------------------------------------------------------------------------
// Lane λ of the wave works for atom g = (λ >> 2) & 3 as its lane l = 4 (λ >> 4) + (λ & 3): the four atoms are interleaved quad
// by quad, so that two bits of l are the wave's lane bits 5 and 4 -- the ones gfx950's v_permlane32_swap / v_permlane16_swap
// exchange between two registers -- and two are bits inside a quad (DPP quad_perm).
// One level of the transposing butterfly over a lane bit, for a register pair (ra, rb): lanes with the bit clear come out with
// ra + partner's ra, lanes with the bit set with rb + partner's rb (partner = λ ^ bit).  16 registers x 16 lanes become one
// register whose 16 lanes hold the 16 totals after 8 + 4 + 2 + 1 such steps.
//   bit 5: v_permlane32_swap a, b   a <- (a.lo32lanes, b.lo32lanes), b <- (a.hi, b.hi): two swaps (a double is two words)
//   bit 4: v_permlane16_swap a, b   the same with odd / even rows of 16          and one add = 3 instructions
//   bits 1, 0: select what to keep and what to send (v_cndmask), quad_perm move, add = 7 instructions
template <bool ROWS16>
__device__ __forceinline__ double sh_comb_swap(double ra, double rb)
{
    const unsigned long long a = __builtin_bit_cast(unsigned long long, ra), b = __builtin_bit_cast(unsigned long long, rb);
    const unsigned alo = (unsigned)a, ahi = (unsigned)(a >> 32), blo = (unsigned)b, bhi = (unsigned)(b >> 32);
    unsigned nalo, nahi, nblo, nbhi;
    if (ROWS16) {
        const auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
        nalo = lo[0]; nblo = lo[1]; nahi = hi[0]; nbhi = hi[1];
    } else {
        const auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
        nalo = lo[0]; nblo = lo[1]; nahi = hi[0]; nbhi = hi[1];
    }
    return __builtin_bit_cast(double, ((unsigned long long)nahi << 32) | nalo) + __builtin_bit_cast(double, ((unsigned long long)nbhi << 32) | nblo);
}
template <int QUAD>
__device__ __forceinline__ double sh_quad(double x)          // x of the quad partner
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(u & 0xffffffffull), QUAD, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(u >> 32), QUAD, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <int QUAD>
__device__ __forceinline__ double sh_comb_quad(double ra, double rb, bool bit)
{
    const double keep = bit ? rb : ra, send = bit ? ra : rb;
    return keep + sh_quad<QUAD>(send);
}
// plain sums over one lane bit (both lanes of a pair get the total)
template <bool ROWS16>
__device__ __forceinline__ double sh_sum_swap(double x) { return sh_comb_swap<ROWS16>(x, x); }
template <int QUAD>
__device__ __forceinline__ double sh_sum_quad(double x) { return x + sh_quad<QUAD>(x); }

// RP = 1, 2, 4, 8, 16 registers summed over the 16 lanes of every atom: a lane returns the total of v[ShLane::jrev & (RP-1)]
// (every lane of the atom that shares those bits holds the same total).  v is clobbered.
template <int RP>
__device__ __forceinline__ double sh_row_reduce(double (&v)[16], bool bit1, bool bit0)
{
    int n = RP;
    if (n >= 2) {
        n >>= 1;
#pragma unroll
        for (int i = 0; i < n; i++) v[i] = sh_comb_swap<false>(v[2 * i], v[2 * i + 1]);
    } else v[0] = sh_sum_swap<false>(v[0]);
    if (n >= 2) {
        n >>= 1;
#pragma unroll
        for (int i = 0; i < n; i++) v[i] = sh_comb_swap<true>(v[2 * i], v[2 * i + 1]);
    } else v[0] = sh_sum_swap<true>(v[0]);
    if (n >= 2) {
        n >>= 1;
#pragma unroll
        for (int i = 0; i < n; i++) v[i] = sh_comb_quad<0x4E>(v[2 * i], v[2 * i + 1], bit1);      // quad_perm [2,3,0,1]
    } else v[0] = sh_sum_quad<0x4E>(v[0]);
    if (n >= 2) v[0] = sh_comb_quad<0xB1>(v[0], v[1], bit0);                                        // quad_perm [1,0,3,2]
    else v[0] = sh_sum_quad<0xB1>(v[0]);
    return v[0];
}
__host__ __device__ constexpr int sh_pow2_at_least(int r) { return r <= 1 ? 1 : r <= 2 ? 2 : r <= 4 ? 4 : r <= 8 ? 8 : 16; }

typedef double shf_v2d __attribute__((ext_vector_type(2)));

struct ShLane {
    unsigned a_sa, a_sc, a_sz;     // LDS byte addresses of this lane's first slot of (e_x,e_y), of the running power and of z
    double *pwg;           // pw of this lane's atom
    double *Abase;         // the moment row of the wave's first atom (uniform) ...
    unsigned arow;         // ... and the byte offset of this lane's atom's row from it (a lane without an atom: 0, and alive = false)
    bool alive;
    unsigned a_mom;        // LDS byte address of the atom's monomial moments (sh_legendre)
    unsigned a_grp;        // GROUPED: LDS byte address of the wave's group buffer ...
    unsigned g16;          // ... and 16 x this lane's atom's number in the wave: its entries of group q begin at a_grp + g16 * SHG_NENT[q]
    int iters;             // LDS-resident neighbours per lane to walk (uniform, may be 0)
    int l16;               // the lane's number among the 16 of its atom
    int jrev;              // which of a batch's 16 totals this lane ends up with: wave lane bits 5,4,1,0 -> bits 0,1,2,3
    bool bit1, bit0;       // wave lane bits 1 and 0
};
// the first SH_R neighbours of a lane (l, l+16, l+32, .. of its atom) stay in registers for the whole kernel: 40 bytes of LDS per
// neighbour are what bounds the waves a CU holds, and with them the kernel's speed
struct ShRegs { double z[SH_R], ex[SH_R], ey[SH_R], pc[SH_R], ps[SH_R]; };

// batch B of column M: sums 16B .. 16B+R-1 of the column (cosine j = 0..K-1, then sine j = 0..K-1) added up over the atom's
// lanes; the lane that ends up with a total parks it in the atom's moment row in HBM -- the registers are needed for the next
// column, and LDS is full of neighbours until the last one is done (sh_fetch_totals brings the row back in one go).
template <int M, int B, bool GROUPED>
__device__ __forceinline__ void sh_batch(const ShLane &w, const double *ac, const double *as)
{
    constexpr int K = SH_LMAX + 1 - M;
    constexpr int NV = (M > 0 ? 2 : 1) * K;
    constexpr int R = NV - 16 * B < 16 ? NV - 16 * B : 16;
    constexpr int RP = sh_pow2_at_least(R);
    double v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int vv = 16 * B + j;
        const int k = vv < K ? vv : vv - K;
        v[j] = (j < R) ? (vv < K ? ac[k < K ? k : 0] : as[k < K ? k : 0]) : 0.0;
    }
    const double t = sh_row_reduce<RP>(v, w.bit1, w.bit0);
    const int j = w.jrev & (RP - 1);
    const int vv = 16 * B + j;
    const int jj = vv < K ? vv : vv - K;
    if constexpr (GROUPED) {
        // into the group buffer: every atom of the wave, existing or not (an atom without neighbours has zero totals; its entries are
        // what the atom before it reads beyond its own with zero coefficients: they must be numbers)
        if ((w.jrev & (16 - RP)) == 0 && j < R) {
            constexpr int Q = shg_group_of(M);
            typedef __attribute__((address_space(3))) double *l1p;
            const unsigned at = w.a_grp + w.g16 * (unsigned)SHG_NENT[Q] + 8u * (unsigned)(sh_apos(M, 0) - 2 * SHG_GBASE[Q] + (vv < K ? 0 : 1)) - 16u * (unsigned)jj;
            *(l1p)(uintptr_t)at = t;
            if (M == 0) *(l1p)(uintptr_t)(at + 8u) = 0.0;
        }
    } else
    if ((w.jrev & (16 - RP)) == 0 && j < R && w.alive) {       // one lane per total
        // (one uniform base and a 32-bit offset per lane: the 32 batches would otherwise keep a 64-bit address each)
        const unsigned off = w.arow + 8u * (unsigned)(sh_apos(M, 0) + (vv < K ? 0 : 1)) - 16u * (unsigned)jj;
        *reinterpret_cast<double *>(reinterpret_cast<char *>(w.Abase) + off) = t;
        if (M == 0) *reinterpret_cast<double *>(reinterpret_cast<char *>(w.Abase) + off + 8u) = 0.0;      // (column 0 has no sine moments: the pair's other half)
    }
}

template <int M, bool GROUPED>
__device__ __forceinline__ void sh_column(const ShLane &w, ShRegs &st)
{
    constexpr int K = SH_LMAX + 1 - M;
    constexpr int NV = (M > 0 ? 2 : 1) * K;
    constexpr int NB = (NV + 15) / 16;
    static_assert(NB <= 3, "three batches per column at most");
    double ac[K], as[K];
    // one neighbour: z^j, j = 0..K-1, times its power (cx, cy) = fc (x+iy)^M into the accumulators (the first one sets them)
    auto neighbour = [&](auto first, double z, const double cx, const double cy) {
        constexpr bool FIRST = decltype(first)::value;
        // (the powers of z are the same in every column: left to itself the compiler keeps those of the register-resident neighbours
        // for all nineteen -- a hundred registers, a wave less per SIMD.  One multiply per power is what they cost here.)
        asm volatile("" : "+v"(z));
        ac[0] = FIRST ? cx : ac[0] + cx;
        as[0] = (M > 0) ? (FIRST ? cy : as[0] + cy) : 0.0;
        if (K > 1) {
            ac[1] = FIRST ? z * cx : fma(z, cx, ac[1]);
            as[1] = (M > 0) ? (FIRST ? z * cy : fma(z, cy, as[1])) : 0.0;
        }
        double P = z;
#pragma unroll
        for (int k = 2; k < K; k++) {
            P *= z;
            ac[k] = FIRST ? P * cx : fma(P, cx, ac[k]);
            as[k] = (M > 0) ? (FIRST ? P * cy : fma(P, cy, as[k])) : 0.0;
        }
    };
#pragma unroll
    for (int r = 0; r < SH_R; r++) {
        const double cx = st.pc[r], cy = st.ps[r];
        if (r == 0) neighbour(std::true_type{}, st.z[r], cx, cy);
        else neighbour(std::false_type{}, st.z[r], cx, cy);
        if (M < SH_LMAX) {          // fc (x+iy)^(m+1)
            st.pc[r] = fma(cx, st.ex[r], -(cy * st.ey[r]));
            st.ps[r] = fma(cx, st.ey[r], cy * st.ex[r]);
        }
    }
    int left = uniform(w.iters);
    if (left > 0) {
        // The LDS-resident neighbours, two per trip (an odd one first).  A neighbour's z and power are requested while the one
        // before it is worked on, into the registers that one has just given up (no moves), (e_x,e_y) only where the power is
        // advanced; the three arrays are walked with one address each and immediate offsets (the compiler otherwise keeps a
        // scalar base and a per-lane offset per array and adds them for every access: a dozen integer instructions per
        // neighbour and column, a tenth of the kernel).  One slot past the last is read and dropped: it exists.
        typedef __attribute__((address_space(3))) shf_v2d *l2p;
        typedef __attribute__((address_space(3))) double *l1p;
        unsigned qa = w.a_sa, qc = w.a_sc, qz = w.a_sz;
        auto advance = [&](const shf_v2d C, const shf_v2d A, const unsigned at) {
            if (M < SH_LMAX) {
                shf_v2d r;
                r.x = fma(C.x, A.x, -(C.y * A.y)); r.y = fma(C.x, A.y, C.y * A.x);
                *(l2p)(uintptr_t)at = r;
            }
        };
        shf_v2d C0 = *(l2p)(uintptr_t)qc;
        double z0 = *(l1p)(uintptr_t)qz;
        if (left & 1) {
            const shf_v2d A0 = *(l2p)(uintptr_t)qa;
            const shf_v2d Cn = *(l2p)(uintptr_t)(qc + 16u * SH_GL);
            const double zn = *(l1p)(uintptr_t)(qz + 8u * SH_GL);
            neighbour(std::false_type{}, z0, C0.x, C0.y);
            advance(C0, A0, qc);
            qa += 16u * SH_GL; qc += 16u * SH_GL; qz += 8u * SH_GL;
            C0 = Cn; z0 = zn;
            left--;
        }
        while (left > 0) {
            const shf_v2d A0 = *(l2p)(uintptr_t)qa;
            const shf_v2d C1 = *(l2p)(uintptr_t)(qc + 16u * SH_GL);
            const double z1 = *(l1p)(uintptr_t)(qz + 8u * SH_GL);
            neighbour(std::false_type{}, z0, C0.x, C0.y);
            advance(C0, A0, qc);
            const shf_v2d A1 = *(l2p)(uintptr_t)(qa + 16u * SH_GL);
            C0 = *(l2p)(uintptr_t)(qc + 32u * SH_GL);
            z0 = *(l1p)(uintptr_t)(qz + 16u * SH_GL);
            neighbour(std::false_type{}, z1, C1.x, C1.y);
            advance(C1, A1, qc + 16u * SH_GL);
            qa += 32u * SH_GL; qc += 32u * SH_GL; qz += 16u * SH_GL;
            left -= 2;
        }
    }
    sh_batch<M, 0, GROUPED>(w, ac, as);
    if (NB > 1) sh_batch<M, (NB > 1 ? 1 : 0), GROUPED>(w, ac, as);
    if (NB > 2) sh_batch<M, (NB > 2 ? 2 : 0), GROUPED>(w, ac, as);
}
// Columns M and M + 1 together (round 6): the powers of z are the same for both -- one multiply serves four multiply-adds instead of two --
// and a neighbour's running power fc (x+iy)^m is advanced twice in a row, the first result used on the way.  Twice the accumulators,
// 4 K - 2 of them, which is why only the columns from SH_PAIR_FROM on are paired: from m = 10 (34 accumulators; the single column m = 1
// has 36) the kernel keeps its 168 registers without scratch, from m = 8 the compiler spills 6 registers, from m = 4 45.  Measured on one
// box, alternating (profiles/r06_kernel_experiments.txt): no pairs 4.746-4.755 ms per 1 M atoms, from m = 12 4.680-4.693, from m = 10
// 4.654-4.667, from m = 8 (with its spills) 4.657-4.674.  Sixteen double-precision multiplies per neighbour fewer (2.6 % of what the
// columns execute) for 1.9 % of the pass -- where 95 fewer shuffle / select instructions per atom bought nothing: the pass answers to its
// double-precision work, not to its instruction count.  The long columns only fit cut into bands of powers (sh_column2: a sweep over the
// neighbours per band, fc (x+iy)^(M+1) made again in each): pairs (5,6) and (7,8) in two bands of at most ANNP_SH_BAND_ACC = 28 accumulators
// keep the registers and take another 1.1-1.9 % off the pass (4.65 -> 4.60 ms on one box, 4.76 -> 4.67 on a slower one); from m = 3 or m = 1 in two
// bands the compiler spills 6 / 10 registers and the gain is gone, in three bands the multiplies saved are spent on the sweeps' own start-up.
#ifndef ANNP_SH_PAIR_FROM
#define ANNP_SH_PAIR_FROM 10          // (99: no column is paired)
#endif
#ifndef ANNP_SH_BAND_FROM
#define ANNP_SH_BAND_FROM 5           // pairs (m, m+1), m = BAND_FROM, BAND_FROM + 2, .. < PAIR_FROM - 1 go in two bands of powers (99: none)
#endif
constexpr int SH_PAIR_FROM = ANNP_SH_PAIR_FROM, SH_BAND_FROM = ANNP_SH_BAND_FROM;
// z^J with the fewest multiplies a square-and-multiply chain gives (J is small and known at compile time)
template <int J>
__device__ __forceinline__ double sh_zpow(const double z)
{
    if constexpr (J == 0) return 1.0;
    else if constexpr (J == 1) return z;
    else if constexpr (J % 2 == 0) { const double h = sh_zpow<J / 2>(z); return h * h; }
    else return sh_zpow<J - 1>(z) * z;
}
// The totals of one sweep over the neighbours -- the powers [J0, J1) of columns M and M + 1: cosine and sine of column M, then of column
// M + 1 -- summed over the atom's lanes sixteen at a time and parked in the moment row (as sh_batch does for a whole column).
template <int M, int J0, int J1, int B>
__device__ __forceinline__ void sh_batch_band(const ShLane &w, const double *ac, const double *as, const double *bc, const double *bs)
{
    constexpr int K = SH_LMAX + 1 - M, K1 = K - 1;
    constexpr int NA = (J1 < K ? J1 : K) - J0, NBB = (J1 < K1 ? J1 : K1) - J0;     // powers of column M, of column M + 1 in this band
    constexpr int S1 = NA, S2 = 2 * NA, S3 = 2 * NA + NBB, NV = 2 * NA + 2 * NBB;
    constexpr int R = NV - 16 * B < 16 ? NV - 16 * B : 16;
    constexpr int RP = sh_pow2_at_least(R);
    static_assert(NA >= 1 && NBB >= 1 && R >= 1, "a band has powers of both columns");
    double v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int vv = 16 * B + j;
        double x = 0.0;
        if (j < R) {
            if (vv < S1) x = ac[vv < S1 ? vv : 0];
            else if (vv < S2) x = as[(vv - S1 >= 0 && vv - S1 < NA) ? vv - S1 : 0];
            else if (vv < S3) x = bc[(vv - S2 >= 0 && vv - S2 < NBB) ? vv - S2 : 0];
            else x = bs[(vv - S3 >= 0 && vv - S3 < NBB) ? vv - S3 : 0];
        }
        v[j] = x;
    }
    const double t = sh_row_reduce<RP>(v, w.bit1, w.bit0);
    const int j = w.jrev & (RP - 1);
    const int vv = 16 * B + j;
    if ((w.jrev & (16 - RP)) == 0 && j < R && w.alive) {       // one lane per total
        const bool second = vv >= S2;                          // column M + 1
        const int q = second ? vv - S2 : vv, n = second ? NBB : NA;
        const bool sine = q >= n;
        const int jj = J0 + (sine ? q - n : q);                // the power
        const unsigned pos = second ? (unsigned)sh_apos(M + 1, 0) : (unsigned)sh_apos(M, 0);
        const unsigned off = w.arow + 8u * (pos + (sine ? 1u : 0u)) - 16u * (unsigned)jj;
        *reinterpret_cast<double *>(reinterpret_cast<char *>(w.Abase) + off) = t;
    }
}
// The powers [J0, J1) of columns M and M + 1 in one sweep over the neighbours.  LAST: the sweep that also advances the neighbours' running
// power fc (x+iy)^m by two (an earlier band of the same pair leaves it alone and makes fc (x+iy)^(M+1) again in the next sweep).
template <int M, int J0, int J1, bool LAST>
__device__ __forceinline__ void sh_pair_band(const ShLane &w, ShRegs &st)
{
    static_assert(M >= 1 && M + 1 <= SH_LMAX, "both columns have sine parts");
    constexpr int K = SH_LMAX + 1 - M, K1 = K - 1;            // entries of column M, of column M + 1
    constexpr int NA = (J1 < K ? J1 : K) - J0, NBB = (J1 < K1 ? J1 : K1) - J0;
    static_assert(NA >= 1 && NBB >= 1 && NA >= NBB, "a band has powers of both columns");
    double ac[NA], as[NA], bc[NBB], bs[NBB];
    // one neighbour: z^j times (cx, cy) = fc (x+iy)^M into column M, times (dx, dy) = fc (x+iy)^(M+1) into column M + 1
    auto neighbour = [&](auto first, double z, const double cx, const double cy, const double dx, const double dy) {
        constexpr bool FIRST = decltype(first)::value;
        asm volatile("" : "+v"(z));          // (as in sh_column: the powers are made again per sweep, not kept across the columns)
        double P = sh_zpow<J0>(z);
#pragma unroll
        for (int k = 0; k < NA; k++) {
            if (k > 0) P = (J0 + k == 1) ? z : P * z;
            if (J0 + k == 0) {
                ac[k] = FIRST ? cx : ac[k] + cx; as[k] = FIRST ? cy : as[k] + cy;
                if (k < NBB) { bc[k] = FIRST ? dx : bc[k] + dx; bs[k] = FIRST ? dy : bs[k] + dy; }
            } else {
                ac[k] = FIRST ? P * cx : fma(P, cx, ac[k]);
                as[k] = FIRST ? P * cy : fma(P, cy, as[k]);
                if (k < NBB) {
                    bc[k] = FIRST ? P * dx : fma(P, dx, bc[k]);
                    bs[k] = FIRST ? P * dy : fma(P, dy, bs[k]);
                }
            }
        }
    };
#pragma unroll
    for (int r = 0; r < SH_R; r++) {
        const double cx = st.pc[r], cy = st.ps[r];
        const double dx = fma(cx, st.ex[r], -(cy * st.ey[r])), dy = fma(cx, st.ey[r], cy * st.ex[r]);       // fc (x+iy)^(M+1)
        if (r == 0) neighbour(std::true_type{}, st.z[r], cx, cy, dx, dy);
        else neighbour(std::false_type{}, st.z[r], cx, cy, dx, dy);
        if (LAST && M + 1 < SH_LMAX) {      // fc (x+iy)^(M+2)
            st.pc[r] = fma(dx, st.ex[r], -(dy * st.ey[r]));
            st.ps[r] = fma(dx, st.ey[r], dy * st.ex[r]);
        }
    }
    int left = uniform(w.iters);
    if (left > 0) {
        // the LDS-resident neighbours, as in sh_column (two per trip, an odd one first, the next one's z and power requested a neighbour ahead)
        typedef __attribute__((address_space(3))) shf_v2d *l2p;
        typedef __attribute__((address_space(3))) double *l1p;
        unsigned qa = w.a_sa, qc = w.a_sc, qz = w.a_sz;
        auto one = [&](const double z, const shf_v2d C, const shf_v2d A, const unsigned at) {
            shf_v2d D;
            D.x = fma(C.x, A.x, -(C.y * A.y)); D.y = fma(C.x, A.y, C.y * A.x);
            neighbour(std::false_type{}, z, C.x, C.y, D.x, D.y);
            if (LAST && M + 1 < SH_LMAX) {
                shf_v2d E;
                E.x = fma(D.x, A.x, -(D.y * A.y)); E.y = fma(D.x, A.y, D.y * A.x);
                *(l2p)(uintptr_t)at = E;
            }
        };
        shf_v2d C0 = *(l2p)(uintptr_t)qc;
        double z0 = *(l1p)(uintptr_t)qz;
        if (left & 1) {
            const shf_v2d A0 = *(l2p)(uintptr_t)qa;
            const shf_v2d Cn = *(l2p)(uintptr_t)(qc + 16u * SH_GL);
            const double zn = *(l1p)(uintptr_t)(qz + 8u * SH_GL);
            one(z0, C0, A0, qc);
            qa += 16u * SH_GL; qc += 16u * SH_GL; qz += 8u * SH_GL;
            C0 = Cn; z0 = zn;
            left--;
        }
        while (left > 0) {
            const shf_v2d A0 = *(l2p)(uintptr_t)qa;
            const shf_v2d C1 = *(l2p)(uintptr_t)(qc + 16u * SH_GL);
            const double z1 = *(l1p)(uintptr_t)(qz + 8u * SH_GL);
            one(z0, C0, A0, qc);
            const shf_v2d A1 = *(l2p)(uintptr_t)(qa + 16u * SH_GL);
            C0 = *(l2p)(uintptr_t)(qc + 32u * SH_GL);
            z0 = *(l1p)(uintptr_t)(qz + 16u * SH_GL);
            one(z1, C1, A1, qc + 16u * SH_GL);
            qa += 32u * SH_GL; qc += 32u * SH_GL; qz += 16u * SH_GL;
            left -= 2;
        }
    }
    constexpr int NBT = (2 * NA + 2 * NBB + 15) / 16;
    static_assert(NBT <= 3, "three batches per sweep at most");
    sh_batch_band<M, J0, J1, 0>(w, ac, as, bc, bs);
    if (NBT > 1) sh_batch_band<M, J0, J1, (NBT > 1 ? 1 : 0)>(w, ac, as, bc, bs);
    if (NBT > 2) sh_batch_band<M, J0, J1, (NBT > 2 ? 2 : 0)>(w, ac, as, bc, bs);
}
// columns M and M + 1: in one sweep where their 4 K - 2 accumulators fit the registers (M >= SH_PAIR_FROM), in two bands of powers otherwise
#ifndef ANNP_SH_BAND_ACC
#define ANNP_SH_BAND_ACC 28           // accumulators a sweep of a banded pair may have (the kernel keeps its registers without scratch up to there)
#endif
template <int M, int NBAND, int B>
struct ShBands {
    static constexpr int K = SH_LMAX + 1 - M;
    static constexpr int J0 = (K * B + NBAND - 1) / NBAND, J1 = (K * (B + 1) + NBAND - 1) / NBAND;
    static __device__ __forceinline__ void run(const ShLane &w, ShRegs &st)
    {
        sh_pair_band<M, J0, (J1 < K ? J1 : K), B + 1 == NBAND>(w, st);
        if constexpr (B + 1 < NBAND) ShBands<M, NBAND, B + 1>::run(w, st);
    }
};
template <int M>
__device__ __forceinline__ void sh_column2(const ShLane &w, ShRegs &st)
{
    constexpr int K = SH_LMAX + 1 - M;
    if constexpr (M >= SH_PAIR_FROM) sh_pair_band<M, 0, K, true>(w, st);
    else {
        // bands of powers [J_b, J_b+1), as few as keep a sweep's accumulators (4 powers' worth per power) within ANNP_SH_BAND_ACC
        constexpr int NBAND = (4 * K - 2 + ANNP_SH_BAND_ACC - 1) / ANNP_SH_BAND_ACC;
        ShBands<M, NBAND, 0>::run(w, st);
    }
}
// which columns start a pair: SH_BAND_FROM, SH_BAND_FROM + 2, .. while the pair ends below SH_PAIR_FROM (banded), then SH_PAIR_FROM, + 2, ..
__host__ __device__ constexpr bool sh_pair_starts(int m)
{
    if (m + 1 > SH_LMAX) return false;
    if (m >= SH_PAIR_FROM) return (m - SH_PAIR_FROM) % 2 == 0;
    return m >= SH_BAND_FROM && (m - SH_BAND_FROM) % 2 == 0 && m + 1 < SH_PAIR_FROM;
}
template <int Q> __device__ __forceinline__ void shg_tail(const ShLane &w);
template <int M, bool GROUPED>
struct ShColumns {
    static __device__ __forceinline__ void run(const ShLane &w, ShRegs &st)
    {
        if constexpr (!GROUPED && sh_pair_starts(M)) {
            sh_column2<M>(w, st);
            ShColumns<M + 2, GROUPED>::run(w, st);
            return;
        }
        sh_column<M, GROUPED>(w, st);
        if constexpr (GROUPED) {
            // the group's last column is summed: its entries change basis now, out of the group buffer (the reads of the next group's
            // first batch come after the tail's in the wave's LDS queue: no fence between the groups is needed beyond the compiler's)
            if constexpr (M == SHG_COLS[shg_group_of(M)][1]) { wave_lds_sync(); shg_tail<shg_group_of(M)>(w); wave_lds_sync(); }
        }
        ShColumns<M + 1, GROUPED>::run(w, st);
    }
};
template <bool GROUPED>
struct ShColumns<SH_LMAX + 1, GROUPED> {
    static __device__ __forceinline__ void run(const ShLane &, ShRegs &) {}
};
template <bool GROUPED>
struct ShColumns<SH_LMAX + 2, GROUPED> {
    static __device__ __forceinline__ void run(const ShLane &, ShRegs &) {}
};

// ---- monomial moments (LDS) -> moments of the Pm^(m)_k (the atom's moment row in HBM); kappa |A|^2 into pw_l.
// A_(m+k,m) = sum_(j = k, k-2, ..) M^(m)_kj Mom_jm (Pm^(m)_k = sum_j M_kj z^j, exact rationals): 715 coefficients, 190 entries,
// sixteen per round -- one per lane of an atom -- in descending k (tools/gen_sh_tables.py::tail_schedule): the lanes of a round have
// k within a few of each other; a lane with fewer terms than the round walks multiplies the positions behind its column's last
// by zero (they hold the next column, the next atom's first, or pw: numbers).  The next round's coefficients are requested
// before this round's arithmetic.  (Tried first: changing basis in place in the moment row in HBM -- twelve rounds of cache
// latency per wave ate what the cheaper columns had saved, 5.5 ms either way; and keeping the 32 totals of a lane in registers
// until the last column -- 76 of them spilled, 5.75 ms.)
template <int R>
struct ShTailOps {
    static constexpr int T = SHD_TRIPS[R];
    double c[T];
    double2 mv[T];
    int info;
    double kap;
    __device__ __forceinline__ void load(const ShLane &w)
    {
        // (tables: a uniform address and the lane's 32-bit byte offset, so that the twelve rounds share one offset register
        // instead of keeping a 64-bit address each)
        const unsigned lo = (unsigned)w.l16;
        info = *reinterpret_cast<const unsigned short *>(reinterpret_cast<const char *>(annp_shd_info + R * 16) + 2u * lo);
        kap = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(annp_shd_kappa + R * 16) + 8u * lo);
#pragma unroll
        for (int t = 0; t < T; t++)
            c[t] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(annp_shd_coef + (SHD_TFIRST[R] + t) * 16) + 8u * lo);
        asm volatile("" ::: "memory");          // (two rounds in flight, not all twelve rounds' coefficients: registers)
    }
    __device__ __forceinline__ void finish(const ShLane &w)
    {
        typedef __attribute__((address_space(3))) shf_v2d *l2p;
        const unsigned mo = w.a_mom + 16u * (unsigned)(info & 255);
#pragma unroll
        for (int t = 0; t < T; t++) { const shf_v2d q = *(l2p)(uintptr_t)(mo + 32u * t); mv[t] = make_double2(q.x, q.y); }
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int t = T - 1; t >= 0; t--) { a0 = fma(c[t], mv[t].x, a0); a1 = fma(c[t], mv[t].y, a1); }       // (small powers first)
        if ((info & 0x8000) && w.alive) {
            atomicAdd(w.pwg + ((info >> 8) & 31), kap * fma(a0, a0, a1 * a1));
            *reinterpret_cast<double2 *>(reinterpret_cast<char *>(w.Abase) + (w.arow + 16u * (unsigned)(info & 255))) = make_double2(kap * a0, kap * a1);        // kappa_lm A_lm: what annp_fe_force_sh multiplies W_l with
        }
    }
};
template <int R>
struct ShTail {
    static __device__ __forceinline__ void run(const ShLane &w, ShTailOps<R> &cur)
    {
        if constexpr (R + 1 < SHD_NROUND) {
            ShTailOps<R + 1> nxt;
            nxt.load(w);
            cur.finish(w);
            ShTail<R + 1>::run(w, nxt);
        } else {
            cur.finish(w);
        }
    }
};
__device__ __forceinline__ void sh_legendre(const ShLane &w)
{
    ShTailOps<0> first;
    first.load(w);
    ShTail<0>::run(w, first);
}

// ---- GROUPED: the same, group by group, out of the group buffer (round 6).  Entry (m, k) of group Q reads the powers k, k-2, .. of its
// own column at a_grp + g16 SHG_NENT[Q] + 16 (local position + 2t) and writes kappa A to the atom's moment row -- the only time the row
// is touched: no totals parked there, nothing fetched back (8-10 GB of memory traffic per 1 M-atom launch down to the 3 GB the force
// pass needs).  The rounds of a group follow one another as in ShTail, the next round's coefficients in flight.
template <int R>
struct ShgOps {
    static constexpr int T = SHG_TRIPS[R];
    static constexpr int Q = shg_group_of_round(R);
    double c[T];
    int info;
    double kap;
    __device__ __forceinline__ void load(const ShLane &w)
    {
        const unsigned lo = (unsigned)w.l16;
        info = *reinterpret_cast<const unsigned short *>(reinterpret_cast<const char *>(annp_shg_info + R * 16) + 2u * lo);
        kap = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(annp_shg_kappa + R * 16) + 8u * lo);
#pragma unroll
        for (int t = 0; t < T; t++)
            c[t] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(annp_shg_coef + (SHG_TFIRST[R] + t) * 16) + 8u * lo);
        asm volatile("" ::: "memory");
    }
    __device__ __forceinline__ void finish(const ShLane &w)
    {
        typedef __attribute__((address_space(3))) shf_v2d *l2p;
        const unsigned mo = w.a_grp + w.g16 * (unsigned)SHG_NENT[Q] + 16u * (unsigned)(info & 255);
        shf_v2d mv[T];
#pragma unroll
        for (int t = 0; t < T; t++) mv[t] = *(l2p)(uintptr_t)(mo + 32u * t);
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int t = T - 1; t >= 0; t--) { a0 = fma(c[t], mv[t].x, a0); a1 = fma(c[t], mv[t].y, a1); }       // (small powers first)
        if ((info & 0x8000) && w.alive) {
            atomicAdd(w.pwg + ((info >> 8) & 31), kap * fma(a0, a0, a1 * a1));
            *reinterpret_cast<double2 *>(reinterpret_cast<char *>(w.Abase) + (w.arow + 16u * (unsigned)((info & 255) + SHG_GBASE[Q]))) = make_double2(kap * a0, kap * a1);
        }
    }
};
template <int R, int REND>
struct ShgRounds {
    static __device__ __forceinline__ void run(const ShLane &w, ShgOps<R> &cur)
    {
        if constexpr (R + 1 < REND) {
            ShgOps<R + 1> nxt;
            nxt.load(w);
            cur.finish(w);
            ShgRounds<R + 1, REND>::run(w, nxt);
        } else {
            cur.finish(w);
        }
    }
};
template <int Q>
__device__ __forceinline__ void shg_tail(const ShLane &w)
{
    ShgOps<SHG_RFIRST[Q]> first;
    first.load(w);
    ShgRounds<SHG_RFIRST[Q], SHG_RFIRST[Q + 1]>::run(w, first);
}

// fe_geometry with the sincos coefficients from scalar registers (annp_common.hpp)
__device__ __forceinline__ FeNbr sh_geometry(double2 R0, double2 R1, double pi_over_rc)
{
    FeNbr g;
    g.rinv = fast_rsqrt_ic(R1.y);
    g.r = R1.y * g.rinv;
    g.ex = R0.x * g.rinv; g.ey = R0.y * g.rinv; g.ez = R1.x * g.rinv;
    double sn, cs;
    sincos_0_pi_s(pi_over_rc * g.r, sn, cs);
    g.fc = 0.5 * (cs + 1.0);                    // fe:592
    g.dfc = -0.5 * pi_over_rc * sn;             // fe:593
    return g;
}

// GROUPED (round 6): for launches with room for at most SHG_CAP_MAX neighbours per atom -- the steady state of bcc Fe -- the change of
// basis runs group by group out of LDS (shg_tail); the other instantiation parks the totals in the moment row (any capacity up to SH_CAP_MAX)
template <int NP, int NT, bool GROUPED>
__global__ __launch_bounds__(256, 3) void annp_fe_desc_sh(FeArgs p)
{
    static_assert(NT == SH_LMAX + 1 && NP + NT <= ANNP_GPAD && NP + 1 <= 16, "layout of the output row");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    ANNP_POISON();
    const int lane = lane_id();
    const int wave = uniform(threadIdx.x >> 6);
    const int ii0 = uniform((xcd_block() * (int)(blockDim.x >> 6) + wave) * SH_GA);
    if (ii0 >= p.inum) return;
    const int cap = p.n_cap;                         // in-cutoff neighbours per atom this launch has room for
    const int capl = cap - SH_GL * SH_R;             // ... of which in LDS
    unsigned char *wbase = lds_raw + (size_t)wave * sh_lds_per_wave(cap);
    const int PL = sh_pitch(capl);
    double2 *SA = reinterpret_cast<double2 *>(wbase);
    double2 *SC = SA + SH_GA * PL;
    double *SZ = reinterpret_cast<double *>(SC + SH_GA * PL);
    double *pw = reinterpret_cast<double *>(wbase + sh_lds_per_wave(cap) - SH_PWB);      // [4][20], then radial totals [4][16]: the wave's last bytes
    double *rt = pw + SH_GA * 20;
    double2 *stA = reinterpret_cast<double2 *>(SZ + SH_GA * PL);  // behind the arrays: the staging rows of the register-resident entries, raw (dx,dy)[48], (dz,r^2)[48]
    double2 *stB = stA + SH_GL * SH_R;
    const int g = (lane >> 2) & 3, l = ((lane >> 4) << 2) | (lane & 3);      // atoms interleaved quad by quad (sh_row_reduce)
    const double pi_over_rc = p.por_list;
    const double two_over_rcp = p.two_over_rcp;

    // ---- stage A: the four rows, raw (dx,dy | dz,r^2): in-cutoff entries 0..47 of a row through the staging rows into the
    //      registers of the atom's lanes, the others into the atom's LDS slots.  Lane ga < 4 fetches the header of atom ga.
    int hi = 0, hjn = -1;                // hjn = -1: no such atom
    long long hbase = 0;
    double hx = 0.0, hy = 0.0, hz = 0.0;
    if (lane < SH_GA && ii0 + lane < p.inum) {
        hi = p.ilist ? p.ilist[ii0 + lane] : ii0 + lane;
        hjn = p.numneigh[hi];
        if (p.type && !type_mapped(p.active, p.type[hi])) hjn = 0;          // centre of an unmapped type: no neighbours, zero row
        hbase = p.first[hi];
        hx = p.x[3 * (size_t)hi]; hy = p.x[3 * (size_t)hi + 1]; hz = p.x[3 * (size_t)hi + 2];
    }
    int nl = 0, nmax = 0;
    bool dead = false;                 // this lane's atom does not exist or went to the fix-up queue: no output
    double2 rawA[SH_R], rawB[SH_R];
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int jn_all = max(max(__builtin_amdgcn_readlane(hjn, 0), __builtin_amdgcn_readlane(hjn, 1)),
                           max(__builtin_amdgcn_readlane(hjn, 2), __builtin_amdgcn_readlane(hjn, 3)));
    auto keep = [&](int ga, int pos, int jat, double dx, double dy, double dz, double rsq) {      // entry `pos` of row ga: atom jat
        if (p.nbrs && pos < cap) p.nbrs[(size_t)(ii0 + ga) * SH_CAP_MAX + pos] = jat;      // (the force pass does not filter the row again)
        if (pos < SH_GL * SH_R) { stA[pos] = make_double2(dx, dy); stB[pos] = make_double2(dz, rsq); }
        else if (pos < cap) { SA[ga * PL + pos - SH_GL * SH_R] = make_double2(dx, dy); SC[ga * PL + pos - SH_GL * SH_R] = make_double2(dz, rsq); }
    };
    int ntrue = 0;               // largest in-cutoff count of the wave's atoms, whether the launch has state for it or not (uniform)
    auto settle = [&](int ga, int n, bool gone) {         // row ga is filtered, n = its in-cutoff count (uniform)
        const int ii = ii0 + ga;
        if (!gone) {
            if (p.ncount && lane == 0) p.ncount[ii] = n;
            ntrue = max(ntrue, n);
            if (n > cap) {                  // more than this launch has room for: the pair-loop kernel takes the atom
                if (lane == 0) {
                    const int k = p.ovf_list ? atomicAdd(p.ovf_count, 1) : p.ovf_cap;
                    if (k < p.ovf_cap) p.ovf_list[k] = ii;
                    else atomicMax(p.errflag, n);
                }
                gone = p.ovf_list != nullptr;       // (without a queue: zero row out, error word set, as annp_fe_desc does)
                n = 0;
            }
        }
        wave_lds_sync();
        if (g == ga) {
            nl = n; dead = gone;
#pragma unroll
            for (int r = 0; r < SH_R; r++) {
                const int a = l + SH_GL * r;
                rawA[r] = a < n ? stA[a] : make_double2(0.0, 0.0);
                rawB[r] = a < n ? stB[a] : make_double2(0.0, 0.0);
            }
        }
        wave_lds_sync();            // the staging rows are free for the next atom
        nmax = max(nmax, n);
    };
    if (jn_all <= 256) {
        // rows of up to 256 entries (an 8.5 A list of bcc Fe has ~235): the index loads of all four rows go out together, then
        // all the coordinate gathers -- two dependent memory round trips for the wave instead of two per row
        int j[SH_GA][4];
        bool valid[SH_GA][4];
#pragma unroll
        for (int ga = 0; ga < SH_GA; ga++) {
            const int jn = __builtin_amdgcn_readlane(hjn, ga);
            const unsigned blo = (unsigned)__builtin_amdgcn_readlane((int)(hbase & 0xffffffffll), ga);
            const int bhi = __builtin_amdgcn_readlane((int)(hbase >> 32), ga);
            // loads are unconditional and clamped to the row's last entry (a load under `valid ? .. : ..` becomes a branch with its
            // own wait); an empty or missing row reads numneigh[0] instead, whatever that is
            const int *row = jn > 0 ? p.neigh + (((long long)bhi << 32) | (long long)blo) : p.numneigh;
            const int last = max(jn, 1) - 1;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int jj = 64 * u + lane;
                valid[ga][u] = jj < jn;
                const int jr = row[min(jj, last)] & ANNP_NEIGHMASK;
                j[ga][u] = valid[ga][u] ? jr : 0;
            }
        }
        if (p.type) {
#pragma unroll
            for (int ga = 0; ga < SH_GA; ga++)
#pragma unroll
                for (int u = 0; u < 4; u++) { const int tj = p.type[j[ga][u]]; valid[ga][u] = valid[ga][u] & type_mapped(p.active, tj); }
        }
        double qx[SH_GA][4], qy[SH_GA][4], qz[SH_GA][4];
#pragma unroll
        for (int ga = 0; ga < SH_GA; ga++)
#pragma unroll
            for (int u = 0; u < 4; u++) {
                qx[ga][u] = p.x[3 * (size_t)j[ga][u]]; qy[ga][u] = p.x[3 * (size_t)j[ga][u] + 1]; qz[ga][u] = p.x[3 * (size_t)j[ga][u] + 2];
            }
#pragma unroll
        for (int ga = 0; ga < SH_GA; ga++) {
            const double xi = readlane_f64(hx, ga), yi = readlane_f64(hy, ga), zi = readlane_f64(hz, ga);
            int n = 0;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const double dx = xi - qx[ga][u], dy = yi - qy[ga][u], dz = zi - qz[ga][u];
                const double rsq = dx * dx + dy * dy + dz * dz;
                const bool in = valid[ga][u] && !(rsq > p.cutsq) && !(rsq < 1.0e-12);        // fe:144
                const unsigned long long m = __ballot(in);
                if (in) keep(ga, n + __popcll(m & lt), j[ga][u], dx, dy, dz, rsq);
                n += __popcll(m);
            }
            settle(ga, uniform(n), __builtin_amdgcn_readlane(hjn, ga) < 0);
        }
    } else {
        for (int ga = 0; ga < SH_GA; ga++) {
            const int jn = __builtin_amdgcn_readlane(hjn, ga);
            const unsigned blo = (unsigned)__builtin_amdgcn_readlane((int)(hbase & 0xffffffffll), ga);
            const int bhi = __builtin_amdgcn_readlane((int)(hbase >> 32), ga);
            const int *row = p.neigh + (((long long)bhi << 32) | (long long)blo);
            const double xi = readlane_f64(hx, ga), yi = readlane_f64(hy, ga), zi = readlane_f64(hz, ga);
            int n = 0;
            for (int c0 = 0; c0 < jn; c0 += 64) {
                const int jj = c0 + lane;
                bool ok = jj < jn;
                const int jx = row[min(jj, jn - 1)] & ANNP_NEIGHMASK;
                if (p.type) { const int tj = p.type[jx]; ok = ok & type_mapped(p.active, tj); }
                const double dx = xi - p.x[3 * (size_t)jx], dy = yi - p.x[3 * (size_t)jx + 1], dz = zi - p.x[3 * (size_t)jx + 2];
                const double rsq = dx * dx + dy * dy + dz * dz;
                const bool in = ok && !(rsq > p.cutsq) && !(rsq < 1.0e-12);
                const unsigned long long m = __ballot(in);
                if (in) keep(ga, n + __popcll(m & lt), jx, dx, dy, dz, rsq);
                n += __popcll(m);
            }
            settle(ga, uniform(n), jn < 0);
        }
    }
    // the evaluation's largest count, for the host's sizing of the next one: a wave that has nothing to add to what it reads there
    // (all but the first few of a launch) does not touch the word
    if (p.nmax_word && lane == 0 && ntrue > __hip_atomic_load(p.nmax_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p.nmax_word, ntrue);
    if (lane < SH_GA * 20) pw[lane] = 0.0;           // (the staging rows are done with: settle ended on a barrier)
    if (lane + 64 < SH_GA * 20) pw[lane + 64] = 0.0;

    // ---- per-neighbour terms; radial sums (fe:633-656)
    double v16[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v16[k] = 0.0;
    auto terms = [&](const double2 R0, const double2 R1, double2 &RA, double &zz, double2 &RC) {
        const FeNbr q = sh_geometry(R0, R1, pi_over_rc);
        RA = make_double2(q.ex, q.ey); zz = q.ez; RC = make_double2(q.fc, 0.0);
        v16[NP] = fma(q.fc, q.fc, v16[NP]);                 // S2 = sum fc^2
        const double xr = q.r * two_over_rcp - 1.0;        // fe:643
        const double y2 = 2.0 * xr;
        double tm2 = 1.0, tm1 = xr;
        v16[0] += q.fc;
        if (NP > 1) v16[1] = fma(xr, q.fc, v16[1]);
#pragma unroll
        for (int mm = 2; mm < NP; mm++) {
            const double t = fma(y2, tm1, -tm2);
            v16[mm] = fma(t, q.fc, v16[mm]);
            tm2 = tm1; tm1 = t;
        }
    };
    ShRegs st;
#pragma unroll
    for (int r = 0; r < SH_R; r++) {
        double2 RA = make_double2(0.0, 0.0), RC = make_double2(0.0, 0.0);
        double zz = 0.0;
        if (l + SH_GL * r < nl) terms(rawA[r], rawB[r], RA, zz, RC);
        st.ex[r] = RA.x; st.ey[r] = RA.y; st.z[r] = zz; st.pc[r] = RC.x; st.ps[r] = RC.y;
    }
    const int iters = uniform(max(0, (nmax - SH_GL * SH_R + SH_GL - 1) / SH_GL));     // LDS-resident neighbours per lane
    {
        int s = g * PL + l;
        for (int it = 0; it < iters; it++, s += SH_GL) {
            double2 RA = make_double2(0.0, 0.0), RC = make_double2(0.0, 0.0);
            double zz = 0.0;
            if (l + SH_GL * (SH_R + it) < nl) terms(SA[s], SC[s], RA, zz, RC);
            SA[s] = RA; SZ[s] = zz; SC[s] = RC;
        }
    }
    ShLane w;
    w.a_sa = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)(SA + g * PL + l);
    w.a_sc = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)(SC + g * PL + l);
    w.a_sz = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)(SZ + g * PL + l);
    w.pwg = pw + g * 20;
    w.iters = iters;
    w.Abase = p.A + (size_t)ii0 * SH_MPAD;
    w.alive = !dead;
    w.arow = dead ? 0u : (unsigned)(g * SH_MPAD * 8);
    w.l16 = l;
    w.a_mom = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)wbase + (unsigned)(g * SHF_NE * 16);
    w.a_grp = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)wbase + (unsigned)SHG_BUF_OFF;
    w.g16 = 16u * (unsigned)g;
    if constexpr (GROUPED) {
        // zeros behind the four atoms' entries of the smallest group up to the end of the buffer: every group's pad (what a larger group
        // writes there later are numbers too).  The staging rows this overwrites are done with (settle ended on a barrier).
        constexpr int Z0 = SH_GA * (SHG_NENT[0] < SHG_NENT[1] ? (SHG_NENT[0] < SHG_NENT[2] ? SHG_NENT[0] : SHG_NENT[2]) : (SHG_NENT[1] < SHG_NENT[2] ? SHG_NENT[1] : SHG_NENT[2])) * 2;
        double *gb = reinterpret_cast<double *>(wbase + SHG_BUF_OFF);
#pragma unroll
        for (int k = Z0 + lane; k < SHG_BUF_BYTES / 8; k += 64) gb[k] = 0.0;
    }
    w.jrev = ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 1) & 1) << 2) | ((lane & 1) << 3);
    w.bit1 = (lane & 2) != 0; w.bit0 = (lane & 1) != 0;
    {
        const double t = sh_row_reduce<16>(v16, w.bit1, w.bit0);
        rt[g * 16 + w.jrev] = t;
    }
    wave_lds_sync();

    // ---- the monomial moments, column by column, parked in the atoms' moment rows; then, the neighbour arrays being done with,
    //      back into LDS in one go (whole rows: 12 loads per lane, one trip through the cache) and from there into the moments of
    //      the Pm^(m)_k: pw_l, and the moment row again, for the force pass
    ShColumns<0, GROUPED>::run(w, st);
    if constexpr (!GROUPED) {
    // (the wave reads back what its own lanes stored: rows are whole 128-byte lines nobody else touches, and the stores are
    // complete before the first load is issued)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (the fences of this scope leave the stores' completion to the hardware's ordering: say it)
    wave_lds_sync();
    {
        double2 *mom = reinterpret_cast<double2 *>(wbase);
        const unsigned long long deadmask = __ballot(dead);
        constexpr int NU = (SHF_NE + 63) / 64;
        double2 q[SH_GA][NU];
        // a row nobody wrote (no such atom, or one that went to the fix-up queue) reads as zeros: its neighbour in LDS multiplies
        // a few of its entries by zero (sh_legendre).  All twelve loads first, then the twelve writes: one trip, not twelve.
#pragma unroll
        for (int ga = 0; ga < SH_GA; ga++) {
            const bool ok = ((deadmask >> (4 * ga)) & 1ull) == 0;
            const double2 *row = reinterpret_cast<const double2 *>(w.Abase + (size_t)(ok ? ga : 0) * SH_MPAD);
#pragma unroll
            for (int u = 0; u < NU; u++) q[ga][u] = row[min(lane + 64 * u, SHF_NE - 1)];
        }
#pragma unroll
        for (int ga = 0; ga < SH_GA; ga++) {
            const bool ok = ((deadmask >> (4 * ga)) & 1ull) == 0;
#pragma unroll
            for (int u = 0; u < NU; u++) {
                const int pos = lane + 64 * u;
                if (pos < SHF_NE) mom[ga * SHF_NE + pos] = ok ? q[ga][u] : make_double2(0.0, 0.0);
            }
        }
        // ... and behind the wave's last atom pw -- or, where the neighbour arrays are longer than the moments, whatever they held
        if (sh_lds_per_wave(cap) - SH_PWB != (size_t)SH_MOMB && lane < 12) reinterpret_cast<double *>(wbase + SH_MOMB)[lane] = 0.0;
    }
    wave_lds_sync();
    sh_legendre(w);
    }
    wave_lds_sync();

    // ---- output row: slots l and l + 16 of the atom's 32
    if (!dead) {
        double *Gout = p.G + (size_t)(ii0 + g) * ANNP_GPAD;
        const double S2 = rt[g * 16 + NP];
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int slot = l + 16 * half;
            double val = 0.0;
            if (slot < NP) val = rt[g * 16 + slot];
            else if (slot < NP + NT) {
                const int n = slot - NP;
                const double *qrow = annp_sh_q + n * NT;
                double acc = 0.0;
#pragma unroll
                for (int lp = 0; lp < NT; lp++) acc = fma(qrow[lp], w.pwg[lp], acc);
                val = 0.5 * (acc - S2);
            }
            Gout[slot] = val;
        }
    }
}

// ---- the pair-loop kernel for the atoms annp_fe_desc_sh queued: one wave per workgroup, every wave walks the queue
template <int NP, int NT>
__global__ __launch_bounds__(64) void annp_fe_desc_fixup(FeArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    ANNP_POISON();
    const int lane = lane_id();
    const int count = min(*p.ovf_count, p.ovf_cap);
    for (int k = blockIdx.x; k < count; k += gridDim.x) {
        fe_desc_atom<NP, NT>(p, uniform(p.ovf_list[k]), lane, lds_raw);
        wave_lds_sync();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The force pass on the moments (annp_fe_force_sh) is in fe_shf_kernels.hpp.  Round 3's version of it -- one wave per atom, the
// three-term recurrences of Pm^(m)_k run again per neighbour and column, four sums behind each step (6 x 190 instructions per
// neighbour) -- is in the history (profiles/r03_*: 7.9 ms per 1 M atoms against 5.7 now).
}  // namespace annp
