// mlp_kernels.hpp -- the per-atom network of pair_style annp, batched over atoms
// on the FP64 matrix cores (v_mfma_f64_16x16x4_f64).
//
// Reference arithmetic: annp_feed_forward / annp_actf / dot_add_wxb
// (annp-gpu-lammps/fe_v2/src/pair_annp.cpp:700-804; ni/src/pair_annp.cpp:769-867).
// The reference carries a full forward-mode Jacobian d(layer)/dG through the
// network for every atom; here the same derivative dE/dG is obtained by one
// reverse sweep, and both sweeps are small GEMMs with the atoms as the N
// dimension:
//     forward   Z_l [d_{l+1} x 16 atoms] = W_l [d_{l+1} x d_l] . H_l [d_l x 16 atoms] + b_l
//     backward  delta_{l-1} = act'(Z_{l-1}) * (W_l^T . delta_l),   dE/dG = W_0^T . delta_0
// Orientation matters: with features on M and atoms on N, the C/D fragment of
// v_mfma_f64_16x16x4 (row = (lane>>4) + 4 reg, col = lane&15) is *already* the B
// fragment of the next product (k = (lane>>4) + 4 kstep, col = lane&15): register r
// of M-tile t is k-step 4t+r.  Activations are applied in place and no data
// ever moves between lanes or through LDS between layers.
//
// What the force pass consumes (fe_kernels.hpp / ni_kernels.hpp) is a fixed linear image
// of dE/dG: c_k = cmul_k dE/dGhat_k for Ni; for the Chebyshev descriptor the radial c_m
// plus the atom's angular polynomial P(z) = sum_n c_n T_n((z+1)/2) re-expanded in powers
// of z = cos(theta) (exact dyadic conversion matrix; the monomial form is well conditioned
// on [-1,1] here: |p_k| <= ~120 max|c_n|) and its derivative, so that the force pass
// evaluates P and dP/dz by Horner instead of running a recurrence.  That image is folded
// into the last product of the reverse sweep: coef = (T diag(cmul) W_0^T) . delta_0, one
// more MFMA chain with a host-precomputed 48 x nnod matrix -- no epilogue arithmetic.
#pragma once
#include "annp_common.hpp"

namespace annp {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int MLP_MAXL = 4;   // weight layers supported by the MFMA path (ntl-1)
// Waves of a workgroup share one copy of the operand image in LDS, and that copy (51 KB for the 27-24-24-1 network) is what
// limits residency: with 4 waves per workgroup a CU held 12 waves of that network, with 8 it holds 16 -- and the pass is a
// chain of dependent MFMAs and tanh evaluations per tile, i.e. it lives on the other waves of its SIMD.
constexpr int MLP_WAVES_PER_BLOCK = 8;
constexpr int MLP_CROW = ANNP_CPAD + 1;   // row pitch of the coefficient staging buffer: an even pitch of 48 puts the 16
                                          // atoms of a fragment column on two banks (8-way conflict on every write)
__host__ __device__ constexpr bool MLP_STAGED(int mt) { return mt == 1; }

struct MlpArgs {
    int inum;
    const int *ilist;             // nullable
    int nsf, nnod, nl;            // nl = ntl-1 weight layers
    int ncoef;                    // rows of coef the force pass reads (Behler: nsf; Chebyshev: c_m | p_k | W_l | P(1) = 9 + 19 + 19 + 1 = 48)
    int act[MLP_MAXL];
    int act_plain;                // ni: activations 3,4 are plain tanh
    const double *img;            // device: the MFMA operand image, [MlpSlots::total][64] (mlp_build_image)
    const double *nmul, *nsub, *nden;   // [ANNP_GPAD] normalisation: Ghat = (raw*nmul - nsub) * nden (nden: reciprocal range)
    double e_scale, e_shift, e_atom;
    int energy_raw;               // ni: E_i = network output (ni:858-860)
    const double *G;              // [inum][ANNP_GPAD]
    double *coef;                 // [inum][ANNP_CPAD]
    double *eatom;                // nullable, indexed by atom
    double *eng;                  // nullable, one double
    // potentials with several elements (or unmapped types): one launch per element, each taking the atoms whose
    // type maps to `elem` and leaving every other row of coef alone (fe_v2/src/pair_annp.cpp:767-768)
    const int *type;              // nullable [nall]
    const int *map;               // device [ntypes+1]
    unsigned active;              // bit t: type t is mapped (annp_common.hpp type_mapped)
    int elem;
};

// tanh(y) to a few ulp without libm's branches: e^{-2|y|} = 2^k (1 + q), q = expm1 of the reduced argument, so
// that e - 1 is formed without cancellation where it matters (k = 0), then tanh|y| = -(e - 1) / (2 + (e - 1)).
__device__ __forceinline__ double tanh_fast(double y)
{
    const double x = fmax(-2.0 * fabs(y), -80.0);          // <= 0; e^-80 is already far below one ulp of 1
    const double kf = rint(x * 1.4426950408889634074);
    double r = fma(-kf, 6.93147180369123816490e-01, x);
    r = fma(-kf, 1.90821492927058770002e-10, r);
    double q = 1.0 / 6227020800.0;                          // r^13/13! ... : q = e^r - 1
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
    q = q * r;
    const int k = (int)kf;
    const double em1 = (k == 0) ? q : __builtin_ldexp(1.0 + q, k) - 1.0;
    // -em1 / (2 + em1) without the division sequence (v_div_scale / v_div_fmas / v_div_fixup and their special cases: ~30
    // instructions, half of this function): the denominator lies in (1, 2], so the hardware reciprocal, two Newton steps and
    // one correction of the quotient give the same result to within an ulp
    const double d = 2.0 + em1;
    double rc = __builtin_amdgcn_rcp(d);
    rc = fma(rc, fma(-d, rc, 1.0), rc);
    rc = fma(rc, fma(-d, rc, 1.0), rc);
    double t = -em1 * rc;
    t = fma(fma(-d, t, -em1), rc, t);
    return copysign(t, y);
}

// Activations of fe_v2/src/pair_annp.cpp:709-739 (ni variant: ni/src/pair_annp.cpp:781-808), all written on one
// tanh so that a layer's flag only selects five wave-uniform numbers:
//     h = A tanh(s a) + C a + D,    dh = P (1 - tanh^2(s a)) + C
//   0 linear                      A 0       s 0    C 1    D 0    P 0
//   1 tanh (ni: also 3, 4)        A 1       s 1    C 0    D 0    P 1
//   2 1/(1+exp(+a)) (sic, 715)    A -1/2    s 1/2  C 0    D 1/2  P +1/4   (the reference's derivative h(1-h), sign as written there)
//   3 1.7159 tanh(2a/3)           A 1.7159  s 2/3  C 0    D 0    P 1.7159*2/3
//   4 ... + 0.1 a                 A 1.7159  s 2/3  C 0.1  D 0    P 1.7159*2/3
struct ActParam { double A, s, C, D, P; };
__host__ __device__ inline ActParam act_param(int flag, int plain)
{
    const double ca = 1.7159, cb = 0.666666666666667, cc = 0.1;
    if (flag == 0) return ActParam{0.0, 0.0, 1.0, 0.0, 0.0};
    if (flag == 1 || (plain && flag >= 3)) return ActParam{1.0, 1.0, 0.0, 0.0, 1.0};
    if (flag == 2) return ActParam{-0.5, 0.5, 0.0, 0.5, 0.25};
    if (flag == 3) return ActParam{ca, cb, 0.0, 0.0, ca * cb};
    return ActParam{ca, cb, cc, 0.0, ca * cb};
}
__device__ __forceinline__ void activation(const ActParam &q, double a, double &h, double &dh)
{
    const double t = tanh_fast(q.s * a);
    h = fma(q.A, t, fma(q.C, a, q.D));
    dh = fma(q.P, fma(-t, t, 1.0), q.C);
}

__device__ __forceinline__ double4_t mfma_f64(double a, double b, double4_t c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// Number of A-operand slots (one double per lane each) for a (KS0, MT, NL) network.
template <int KS0, int MT, int NL>
struct MlpSlots {
    static constexpr int KSH = 4 * MT;                      // k-steps over a hidden layer
    static constexpr int MT0 = ANNP_CPAD / 16;              // M-tiles over the rows of coef
    static constexpr int fwd0 = 0;                          // MT*KS0
    static constexpr int fwdh = fwd0 + MT * KS0;            // (NL-2) * MT*KSH
    static constexpr int fwdo = fwdh + (NL - 2) * MT * KSH; // KSH
    static constexpr int bwdo = fwdo + KSH;                 // MT          (K = 1 -> one k-step)
    static constexpr int bwdh = bwdo + MT;                  // (NL-2) * MT*KSH
    static constexpr int bwd0 = bwdh + (NL - 2) * MT * KSH; // MT0*KSH
    static constexpr int bias = bwd0 + MT0 * KSH;           // (NL-1)*MT*4 + 4
    static constexpr int total = bias + (NL - 1) * MT * 4 + 4;
};

// Host side: lay the weights out as the A operands of the MFMA chains, one double per lane and slot.
// W[l] row-major [d_{l+1}][d_l], B[l], coefmat [ANNP_CPAD][nnod] row-major (coef = coefmat . dE/dZ_0).
template <int KS0, int MT, int NL>
inline void mlp_build_image(double *img, const double *const *W, const double *const *B, const double *coefmat, int nsf, int nnod)
{
    using S = MlpSlots<KS0, MT, NL>;
    constexpr int KSH = S::KSH;
    for (int slot = 0; slot < S::total; slot++)
        for (int lane = 0; lane < 64; lane++) {
            const int lr = lane & 15, lq = lane >> 4;
            double v = 0.0;
            if (slot < S::fwdh) {                                   // fwd layer 0: W0[16mt+lr][4s+lq]
                const int mt = slot / KS0, s = slot % KS0;
                const int r = 16 * mt + lr, c = 4 * s + lq;
                if (r < nnod && c < nsf) v = W[0][r * nsf + c];
            } else if (slot < S::fwdo) {                            // fwd hidden l
                const int q = slot - S::fwdh;
                const int l = 1 + q / (MT * KSH), mt = (q / KSH) % MT, s = q % KSH;
                const int r = 16 * mt + lr, c = 4 * s + lq;
                if (r < nnod && c < nnod) v = W[l][r * nnod + c];
            } else if (slot < S::bwdo) {                            // fwd output: W_last[lr][4s+lq]
                const int s = slot - S::fwdo;
                const int c = 4 * s + lq;
                if (lr == 0 && c < nnod) v = W[NL - 1][c];
            } else if (slot < S::bwdh) {                            // bwd output: W_last^T[16mt+lr][lq]
                const int mt = slot - S::bwdo;
                const int r = 16 * mt + lr;
                if (lq == 0 && r < nnod) v = W[NL - 1][r];
            } else if (slot < S::bwd0) {                            // bwd hidden l: W_l[4s+lq][16mt+lr]
                const int q = slot - S::bwdh;
                const int l = (NL - 2) - q / (MT * KSH), mt = (q / KSH) % MT, s = q % KSH;
                const int r = 4 * s + lq, c = 16 * mt + lr;
                if (r < nnod && c < nnod) v = W[l][r * nnod + c];
            } else if (slot < S::bias) {                            // coef rows: coefmat[16mt+lr][4s+lq]
                const int q = slot - S::bwd0;
                const int mt = q / KSH, s = q % KSH;
                const int r = 16 * mt + lr, c = 4 * s + lq;
                if (r < ANNP_CPAD && c < nnod) v = coefmat[r * nnod + c];
            } else {                                                // bias in C layout: row 16mt+lq+4r
                const int q = slot - S::bias;
                const int l = q / (MT * 4), mt = (q / 4) % MT, r = q % 4;
                const int row = 16 * mt + lq + 4 * r;
                if (l < NL - 1) { if (row < nnod) v = B[l][row]; }
                else if (row == 0) v = B[NL - 1][0];
            }
            img[(size_t)slot * 64 + lane] = v;
        }
}

template <int KS0, int MT, int NL>
__global__ __launch_bounds__(64 * MLP_WAVES_PER_BLOCK) void annp_mlp_mfma(MlpArgs p)
{
    using S = MlpSlots<KS0, MT, NL>;
    constexpr int KSH = S::KSH, MT0 = S::MT0;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    ANNP_POISON();
    double *opnd = reinterpret_cast<double *>(lds_raw);                 // [S::total][64]

    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    double *cbuf = opnd + (size_t)S::total * 64 + (size_t)wave * 16 * MLP_CROW;      // [16][MLP_CROW] per wave (MLP_STAGED only)
    const int lr = lane & 15, lq = lane >> 4;
    const int nsf = p.nsf;

    // ---- the operand image (weights already in A-fragment order, built once by the host) -> LDS
    {
        const double2 *src = reinterpret_cast<const double2 *>(p.img);
        double2 *dst = reinterpret_cast<double2 *>(opnd);
        // (all of a thread's loads first, then its stores: one round trip through memory, not one per 16 bytes)
        constexpr int NT = 64 * MLP_WAVES_PER_BLOCK, PER = (S::total * 32 + NT - 1) / NT;
        double2 tmp[PER];
#pragma unroll
        for (int t = 0; t < PER; t++) { const int idx = threadIdx.x + t * NT; tmp[t] = src[idx < S::total * 32 ? idx : 0]; }
#pragma unroll
        for (int t = 0; t < PER; t++) { const int idx = threadIdx.x + t * NT; if (idx < S::total * 32) dst[idx] = tmp[t]; }
    }
    __syncthreads();
    ActParam ap[NL];
#pragma unroll
    for (int l = 0; l < NL; l++) ap[l] = act_param(p.act[l], p.act_plain);

    // The fragments are sized for the compiled maximum (MT tiles of 16 nodes); a network with fewer nodes leaves whole
    // k-steps and whole accumulator registers empty.  Those are skipped (wave-uniform tests): the 24 nodes of the Ni
    // network need 6 of 8 k-steps per hidden product, 2 of 3 coefficient tiles and no activation for rows 24..31;
    // the 10 nodes of the Fe network 3 of 4 k-steps and 3 of 4 registers.
    const int ksh = uniform((p.nnod + 3) >> 2);           // k-steps that meet a non-zero operand
    const int mt0 = uniform((p.ncoef + 15) >> 4);         // coefficient tiles that are read downstream
    const int nnod_u = uniform(p.nnod);
    double e_wave = 0.0;
    const int ntiles = (p.inum + 15) / 16;
    const int wave_global = blockIdx.x * MLP_WAVES_PER_BLOCK + wave;
    const int wave_stride = gridDim.x * MLP_WAVES_PER_BLOCK;
    // raw descriptor sums of a tile's 16 atoms, fragment order (k = 4s + lq, atom = lr); the next tile's are requested
    // while this one is worked on: a tile is one long chain of dependent MFMAs, so nothing else would hide the read
    double raw[KS0];
    auto request = [&](int tile) {
        const int ia = tile * 16 + lr;
#pragma unroll
        for (int s = 0; s < KS0; s++) {
            const int k = 4 * s + lq;
            raw[s] = (tile < ntiles && ia < p.inum && k < nsf) ? p.G[(size_t)ia * ANNP_GPAD + k] : 0.0;
        }
    };
    request(wave_global);
    for (int tile = wave_global; tile < ntiles; tile += wave_stride) {
        const int ia = tile * 16 + lr;            // this lane's atom (column)
        bool aval = ia < p.inum;
        // ---- input fragment: Ghat[k = 4s+lq][atom]
        double hin[KS0];
#pragma unroll
        for (int s = 0; s < KS0; s++) {
            const int k = 4 * s + lq;
            hin[s] = (aval && k < nsf) ? fma(raw[s], p.nmul[k], -p.nsub[k]) * p.nden[k] : 0.0;
        }
        request(tile + wave_stride);
        if (p.type) {
            if (aval) {
                const int t = p.type[p.ilist ? p.ilist[ia] : ia];
                aval = type_mapped(p.active, t) && p.map[t] == p.elem;
            }
            if (__ballot(aval) == 0ull) continue;                            // no atom of this element in the tile
#pragma unroll
            for (int s = 0; s < KS0; s++) hin[s] = aval ? hin[s] : 0.0;
        }
        // ---- forward
        double4_t H[NL][MT], D[NL][MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            double4_t acc;
#pragma unroll
            for (int r = 0; r < 4; r++) acc[r] = opnd[(size_t)(S::bias + (0 * MT + mt) * 4 + r) * 64 + lane];
#pragma unroll
            for (int s = 0; s < KS0; s++) acc = mfma_f64(opnd[(size_t)(S::fwd0 + mt * KS0 + s) * 64 + lane], hin[s], acc);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                double h = 0.0, d = 0.0;
                if (16 * mt + 4 * r < nnod_u) activation(ap[0], acc[r], h, d);
                H[0][mt][r] = h; D[0][mt][r] = d;
            }
        }
#pragma unroll
        for (int l = 1; l < NL - 1; l++) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                double4_t acc;
#pragma unroll
                for (int r = 0; r < 4; r++) acc[r] = opnd[(size_t)(S::bias + (l * MT + mt) * 4 + r) * 64 + lane];
#pragma unroll
                for (int s = 0; s < KSH; s++)
                    if (s < ksh) acc = mfma_f64(opnd[(size_t)(S::fwdh + ((l - 1) * MT + mt) * KSH + s) * 64 + lane], H[l - 1][s / 4][s % 4], acc);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    double h = 0.0, d = 0.0;
                    if (16 * mt + 4 * r < nnod_u) activation(ap[l], acc[r], h, d);
                    H[l][mt][r] = h; D[l][mt][r] = d;
                }
            }
        }
        double4_t zo;
        {
#pragma unroll
            for (int r = 0; r < 4; r++) zo[r] = opnd[(size_t)(S::bias + (NL - 1) * MT * 4 + r) * 64 + lane];
#pragma unroll
            for (int s = 0; s < KSH; s++)
                if (s < ksh) zo = mfma_f64(opnd[(size_t)(S::fwdo + s) * 64 + lane], H[NL - 2][s / 4][s % 4], zo);
        }
        double out, dout;
        activation(ap[NL - 1], zo[0], out, dout);   // row 0 lives in reg 0 of lanes 0..15

        // ---- energy (fe:790-793 / ni:858-860)
        if (lq == 0 && aval) {
            const double e = p.energy_raw ? out : fma(p.e_scale, out, p.e_shift) + p.e_atom;
            e_wave += e;
            if (p.eatom) {
                const int i = p.ilist ? p.ilist[ia] : ia;
                p.eatom[i] += e;
            }
        }
        // ---- backward
        double4_t dl[MT];     // delta of the current layer, C layout
        {
            const double4_t dlast = {dout, 0.0, 0.0, 0.0};   // only k = 0 meets a non-zero A column
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                double4_t acc = {0.0, 0.0, 0.0, 0.0};
                acc = mfma_f64(opnd[(size_t)(S::bwdo + mt) * 64 + lane], dlast[0], acc);
#pragma unroll
                for (int r = 0; r < 4; r++) dl[mt][r] = acc[r] * D[NL - 2][mt][r];
            }
        }
#pragma unroll
        for (int l = NL - 2; l >= 1; l--) {
            double4_t nx[MT];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < KSH; s++)
                    if (s < ksh) acc = mfma_f64(opnd[(size_t)(S::bwdh + (((NL - 2) - l) * MT + mt) * KSH + s) * 64 + lane], dl[s / 4][s % 4], acc);
#pragma unroll
                for (int r = 0; r < 4; r++) nx[mt][r] = acc[r] * D[l - 1][mt][r];
            }
#pragma unroll
            for (int mt = 0; mt < MT; mt++) dl[mt] = nx[mt];
        }
        // coef[k = 16mt+lq+4r][atom] = coefmat . delta_0.  Small networks (MT = 1: the operand image is ~22 KB) stage the tile
        // through LDS [atom][48] and store whole rows; for the wide ones (MT = 2: 51 KB) those 6.3 KB per wave would halve the
        // resident waves, so they store straight from the fragment: the four registers of an M-tile are 16 consecutive doubles
        // of the atom's row, one 128-byte line completed by four store instructions in a row.
#pragma unroll
        for (int mt = 0; mt < MT0; mt++) {
            double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < KSH; s++)
                if (mt < mt0 && s < ksh) acc = mfma_f64(opnd[(size_t)(S::bwd0 + mt * KSH + s) * 64 + lane], dl[s / 4][s % 4], acc);
            if (MLP_STAGED(MT)) {
#pragma unroll
                for (int r = 0; r < 4; r++) cbuf[lr * MLP_CROW + 16 * mt + lq + 4 * r] = acc[r];
            } else if (aval) {
                double *dst = p.coef + (size_t)ia * ANNP_CPAD + 16 * mt + lq;
#pragma unroll
                for (int r = 0; r < 4; r++) dst[4 * r] = acc[r];
            }
        }
        if (MLP_STAGED(MT)) {
            wave_lds_sync();
            const unsigned rowmask = (unsigned)(__ballot(aval) & 0xffffull);     // lanes 0..15 speak for the 16 atoms
            const int nrow = min(16, p.inum - tile * 16);
            double *dst = p.coef + (size_t)tile * 16 * ANNP_CPAD;
            for (int idx = lane; idx < nrow * ANNP_CPAD; idx += 64)
                if ((rowmask >> (idx / ANNP_CPAD)) & 1u) dst[idx] = cbuf[idx + idx / ANNP_CPAD];
            wave_lds_sync();
        }
    }
    if (p.eng) {
        // one atomic per workgroup on the energy word, not one per wave: thousands of atomics on a single address are served
        // one after the other (40 us of a 73 us launch at 128 000 atoms)
        e_wave = wave_sum(e_wave);
        __syncthreads();                    // every wave is done with the operand image: its first doubles carry the sums
        if (lane == 0) opnd[wave] = e_wave;
        __syncthreads();
        if (threadIdx.x == 0) {
            double e = 0.0;
#pragma unroll
            for (int w = 0; w < MLP_WAVES_PER_BLOCK; w++) e += opnd[w];
            if (e != 0.0) atomicAdd(p.eng, e);
        }
    }
}

template <int KS0, int MT, int NL>
inline size_t mlp_lds_bytes()
{
    return ((size_t)MlpSlots<KS0, MT, NL>::total * 64 + (MLP_STAGED(MT) ? (size_t)MLP_WAVES_PER_BLOCK * 16 * MLP_CROW : 0)) * sizeof(double);
}

}  // namespace annp
