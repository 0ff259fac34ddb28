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

struct MlpArgs {
    int inum;
    const int *ilist;             // nullable
    int nsf, nnod, nl;            // nl = ntl-1 weight layers
    int act[MLP_MAXL];
    int act_plain;                // ni: activations 3,4 are plain tanh
    const double *W[MLP_MAXL];    // device, row-major [d_{l+1}][d_l]
    const double *B[MLP_MAXL];
    const double *nmul, *nsub, *nden;   // [ANNP_GPAD] normalisation: Ghat = (raw*nmul - nsub)/nden
    const double *coefmat;              // [ANNP_CPAD][nnod] row-major: coef = coefmat . dE/dZ_0
    double e_scale, e_shift, e_atom;
    int energy_raw;               // ni: E_i = network output (ni:858-860)
    const double *G;              // [inum][ANNP_GPAD]
    double *coef;                 // [inum][ANNP_CPAD]
    double *eatom;                // nullable, indexed by atom
    double *eng;                  // nullable, one double
};

// fe_v2/src/pair_annp.cpp:709-739 (ni variant: ni/src/pair_annp.cpp:781-808)
__device__ __forceinline__ void activation(int flag, int plain, double a, double &h, double &dh)
{
    const double ca = 1.7159, cb = 0.666666666666667, cc = 0.1;
    if (flag == 0) { h = a; dh = 1.0; }
    else if (flag == 1 || (plain && flag >= 3)) { const double t = tanh(a); h = t; dh = 1.0 - t * t; }
    else if (flag == 2) { h = 1.0 / (1.0 + exp(a)); dh = h * (1.0 - h); }
    else if (flag == 3) { const double t = tanh(cb * a); h = ca * t; dh = ca * (1.0 - t * t) * cb; }
    else { const double t = tanh(cb * a); h = ca * t + cc * a; dh = ca * (1.0 - t * t) * cb + cc; }
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

template <int KS0, int MT, int NL>
__global__ __launch_bounds__(256) void annp_mlp_mfma(MlpArgs p)
{
    using S = MlpSlots<KS0, MT, NL>;
    constexpr int KSH = S::KSH, MT0 = S::MT0;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double *opnd = reinterpret_cast<double *>(lds_raw);                 // [S::total][64]
    double *cbuf_all = opnd + (size_t)S::total * 64;                    // [4 waves][16][ANNP_CPAD]

    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int nsf = p.nsf, nnod = p.nnod;

    // ---- stage the weights once per block, already in A-fragment order
    for (int slot = wave; slot < S::total; slot += ANNP_WAVES_PER_BLOCK) {
        double v = 0.0;
        if (slot < S::fwdh) {                                   // fwd layer 0: W0[16mt+lr][4s+lq]
            const int mt = slot / KS0, s = slot % KS0;
            const int r = 16 * mt + lr, c = 4 * s + lq;
            if (r < nnod && c < nsf) v = p.W[0][r * nsf + c];
        } else if (slot < S::fwdo) {                            // fwd hidden l
            const int q = slot - S::fwdh;
            const int l = 1 + q / (MT * KSH), mt = (q / KSH) % MT, s = q % KSH;
            const int r = 16 * mt + lr, c = 4 * s + lq;
            if (r < nnod && c < nnod) v = p.W[l][r * nnod + c];
        } else if (slot < S::bwdo) {                            // fwd output: W_last[lr][4s+lq]
            const int s = slot - S::fwdo;
            const int c = 4 * s + lq;
            if (lr == 0 && c < nnod) v = p.W[NL - 1][c];
        } else if (slot < S::bwdh) {                            // bwd output: W_last^T[16mt+lr][lq]
            const int mt = slot - S::bwdo;
            const int r = 16 * mt + lr;
            if (lq == 0 && r < nnod) v = p.W[NL - 1][r];
        } else if (slot < S::bwd0) {                            // bwd hidden l: W_l[4s+lq][16mt+lr]
            const int q = slot - S::bwdh;
            const int l = (NL - 2) - q / (MT * KSH), mt = (q / KSH) % MT, s = q % KSH;
            const int r = 4 * s + lq, c = 16 * mt + lr;
            if (r < nnod && c < nnod) v = p.W[l][r * nnod + c];
        } else if (slot < S::bias) {                            // coef rows: coefmat[16mt+lr][4s+lq]
            const int q = slot - S::bwd0;
            const int mt = q / KSH, s = q % KSH;
            const int r = 16 * mt + lr, c = 4 * s + lq;
            if (r < ANNP_CPAD && c < nnod) v = p.coefmat[r * nnod + c];
        } else {                                                // bias in C layout: row 16mt+lq+4r
            const int q = slot - S::bias;
            const int l = q / (MT * 4), mt = (q / 4) % MT, r = q % 4;
            const int row = 16 * mt + lq + 4 * r;
            if (l < NL - 1) { if (row < nnod) v = p.B[l][row]; }
            else if (row == 0) v = p.B[NL - 1][0];
        }
        opnd[(size_t)slot * 64 + lane] = v;
    }
    __syncthreads();
    double *cbuf = cbuf_all + (size_t)wave * 16 * ANNP_CPAD;

    double e_wave = 0.0;
    const int ntiles = (p.inum + 15) / 16;
    const int wave_global = blockIdx.x * ANNP_WAVES_PER_BLOCK + wave;
    const int wave_stride = gridDim.x * ANNP_WAVES_PER_BLOCK;
    for (int tile = wave_global; tile < ntiles; tile += wave_stride) {
        const int ia = tile * 16 + lr;            // this lane's atom (column)
        const bool aval = ia < p.inum;
        // ---- input fragment: Ghat[k = 4s+lq][atom]
        double hin[KS0];
#pragma unroll
        for (int s = 0; s < KS0; s++) {
            const int k = 4 * s + lq;
            double g = 0.0;
            if (aval && k < nsf) {
                const double raw = p.G[(size_t)ia * ANNP_GPAD + k];
                g = fma(raw, p.nmul[k], -p.nsub[k]) / p.nden[k];
            }
            hin[s] = g;
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
            for (int r = 0; r < 4; r++) { double h, d; activation(p.act[0], p.act_plain, acc[r], h, d); H[0][mt][r] = h; D[0][mt][r] = d; }
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
                    acc = mfma_f64(opnd[(size_t)(S::fwdh + ((l - 1) * MT + mt) * KSH + s) * 64 + lane], H[l - 1][s / 4][s % 4], acc);
#pragma unroll
                for (int r = 0; r < 4; r++) { double h, d; activation(p.act[l], p.act_plain, acc[r], h, d); H[l][mt][r] = h; D[l][mt][r] = d; }
            }
        }
        double4_t zo;
        {
#pragma unroll
            for (int r = 0; r < 4; r++) zo[r] = opnd[(size_t)(S::bias + (NL - 1) * MT * 4 + r) * 64 + lane];
#pragma unroll
            for (int s = 0; s < KSH; s++) zo = mfma_f64(opnd[(size_t)(S::fwdo + s) * 64 + lane], H[NL - 2][s / 4][s % 4], zo);
        }
        double out, dout;
        activation(p.act[NL - 1], p.act_plain, zo[0], out, dout);   // row 0 lives in reg 0 of lanes 0..15

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
                    acc = mfma_f64(opnd[(size_t)(S::bwdh + (((NL - 2) - l) * MT + mt) * KSH + s) * 64 + lane], dl[s / 4][s % 4], acc);
#pragma unroll
                for (int r = 0; r < 4; r++) nx[mt][r] = acc[r] * D[l - 1][mt][r];
            }
#pragma unroll
            for (int mt = 0; mt < MT; mt++) dl[mt] = nx[mt];
        }
        // coef[k = 16mt+lq+4r][atom] = coefmat . delta_0  -> LDS [atom][48] -> coalesced rows
#pragma unroll
        for (int mt = 0; mt < MT0; mt++) {
            double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < KSH; s++)
                acc = mfma_f64(opnd[(size_t)(S::bwd0 + mt * KSH + s) * 64 + lane], dl[s / 4][s % 4], acc);
#pragma unroll
            for (int r = 0; r < 4; r++) cbuf[lr * ANNP_CPAD + 16 * mt + lq + 4 * r] = acc[r];
        }
        wave_lds_sync();
        {
            const int nrow = min(16, p.inum - tile * 16);
            double *dst = p.coef + (size_t)tile * 16 * ANNP_CPAD;
            for (int idx = lane; idx < nrow * ANNP_CPAD; idx += 64) dst[idx] = cbuf[idx];
        }
        wave_lds_sync();
    }
    if (p.eng) {
        e_wave = wave_sum(e_wave);
        if (lane == 0 && e_wave != 0.0) atomicAdd(p.eng, e_wave);
    }
}

template <int KS0, int MT, int NL>
inline size_t mlp_lds_bytes()
{
    return ((size_t)MlpSlots<KS0, MT, NL>::total * 64 + (size_t)ANNP_WAVES_PER_BLOCK * 16 * ANNP_CPAD) * sizeof(double);
}

}  // namespace annp
