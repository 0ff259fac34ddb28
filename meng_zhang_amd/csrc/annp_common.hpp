// annp_common.hpp -- shared device helpers for the gfx950 annp kernels.
//
// One wavefront (64 lanes) works on one central atom.  Workgroups are 256
// threads = 4 independent waves; waves never synchronise with each other, so
// the only ordering needed around LDS is wave-local.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ANNP_NEIGHMASK 0x1FFFFFFF          // LAMMPS lmptype.h; applied as fe_v2/src/pair_annp.cpp:136
#define ANNP_MY_PI 3.14159265358979323846  // LAMMPS MathConst::MY_PI
#define ANNP_WAVE 64
#define ANNP_WAVES_PER_BLOCK 4
#define ANNP_GPAD 32      // doubles per atom in the descriptor buffer G[atom][32]
#define ANNP_CPAD 48      // doubles per atom in the coefficient buffer coef[atom][48]

namespace annp {

// Orders this wave's LDS traffic: everything before is visible to every lane after.
// LDS operations of one wave execute in issue order, so a compiler-level fence plus
// the hardware's lgkmcnt tracking is sufficient; no s_barrier is involved.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Developer builds (`make poison`, never the shipped library): the first act of every kernel that uses LDS is to fill its
// workgroup's whole allocation with a pattern, so that a read of LDS the workgroup never wrote meets the worst value instead
// of whatever the previous dispatch on that CU left there.  LDS is not cleared between dispatches: a kernel whose result
// depends on it passes or fails by the history of its CU (VERDICT r5 "What's weak" 1).  Two patterns, two libraries:
//   ANNP_POISON_LDS=1  -1e300: survives fmin / fmax clamps, and its square is +Inf (the exp(-eta r^2) x 0 of round 5's Behler G2 loop);
//   ANNP_POISON_LDS=2  all bits set: a quiet NaN as a double (and -1 as an int, which is what a force table's free keys look
//                      like to the next kernel) -- catches every `x * 0` mask, whatever stands in front of it.
// The size comes from the dispatch packet (hsa_kernel_dispatch_packet_t::group_segment_size, byte 28: static + dynamic LDS);
// annp_hip_poison_selftest (annp_hip.hip, these builds only) checks that the fill really covers the allocation.
#ifdef ANNP_POISON_LDS
__device__ __forceinline__ void annp_poison_lds()
{
    typedef const __attribute__((address_space(4))) unsigned *packet_p;
    const unsigned bytes = ((packet_p)__builtin_amdgcn_dispatch_ptr())[7];         // header, setup, workgroup_size[3], reserved (12 B) | grid_size[3] | private_segment_size | group_segment_size
    typedef __attribute__((address_space(3))) double *lds_dp;
    const unsigned nthreads = blockDim.x * blockDim.y * blockDim.z;
    for (unsigned o = 8u * threadIdx.x; o + 8u <= bytes && o < 160u * 1024u; o += 8u * nthreads)
        *(lds_dp)(uintptr_t)o = ANNP_POISON_LDS == 2 ? __builtin_bit_cast(double, ~0ull) : -1e300;
    __syncthreads();
}
#define ANNP_POISON() annp::annp_poison_lds()
#else
#define ANNP_POISON() do { } while (0)
#endif

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2).  Give each XCD a
// contiguous range of atoms instead, so the positions and list rows a spatial neighbourhood shares are
// served by one L2.  Pure speed: any placement is correct.  Bijective for every grid size
// (the last gridDim % 8 blocks keep their place).
__device__ __forceinline__ int xcd_block()
{
    const int b = blockIdx.x;
    const int q = gridDim.x >> 3;
    if (b >= (q << 3)) return b;
    return (b & 7) * q + (b >> 3);
}

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// a double of lane `srclane` (uniform), through scalar registers
__device__ __forceinline__ double readlane_f64(double v, int srclane)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(u & 0xffffffffull), srclane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(u >> 32), srclane);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// Is LAMMPS type t mapped to an element?  `active` has bit t set for the mapped types 1..ntypes (<= 30).  A type outside that
// range (a caller's mistake the device-resident entry points cannot check on the host) reads as "not mapped": the atom is
// dropped, nothing is indexed with it.
__device__ __forceinline__ bool type_mapped(unsigned active, int t) { return (unsigned)t < 32u && ((active >> t) & 1u); }

// cutoff function and derivative, fe_v2/src/pair_annp.cpp:590-594
__device__ __forceinline__ void cutoff_fc(double r, double pi_over_rc, double &fc, double &dfc)
{
    double sn, cs;
    sincos(pi_over_rc * r, &sn, &cs);
    fc = 0.5 * (cs + 1.0);
    dfc = -0.5 * pi_over_rc * sn;
}

// sin(t), cos(t) for t in [0, pi] (the only range the cutoff function sees: 0 < r <= Rc).
// t = pi/2 + h with |h| <= pi/2: cos t = -sin h, sin t = cos h; Taylor-Horner in h^2,
// truncation below 2e-18.  ~30 fp64 instructions instead of libm's general-range sincos.
__device__ __forceinline__ void sincos_0_pi(double t, double &sn, double &cs)
{
    const double h = t - 1.57079632679489661923;
    const double h2 = h * h;
    double ps = -1.0 / 51090942171709440000.0;             // -1/21!
    ps = fma(ps, h2, 1.0 / 121645100408832000.0);          //  1/19!
    ps = fma(ps, h2, -1.0 / 355687428096000.0);            // -1/17!
    ps = fma(ps, h2, 1.0 / 1307674368000.0);               //  1/15!
    ps = fma(ps, h2, -1.0 / 6227020800.0);                 // -1/13!
    ps = fma(ps, h2, 1.0 / 39916800.0);                    //  1/11!
    ps = fma(ps, h2, -1.0 / 362880.0);                     // -1/9!
    ps = fma(ps, h2, 1.0 / 5040.0);                        //  1/7!
    ps = fma(ps, h2, -1.0 / 120.0);                        // -1/5!
    ps = fma(ps, h2, 1.0 / 6.0);                           //  1/3!   (sign folded below)
    double pc = 1.0 / 1124000727777607680000.0;            //  1/22!
    pc = fma(pc, h2, -1.0 / 2432902008176640000.0);        // -1/20!
    pc = fma(pc, h2, 1.0 / 6402373705728000.0);            //  1/18!
    pc = fma(pc, h2, -1.0 / 20922789888000.0);             // -1/16!
    pc = fma(pc, h2, 1.0 / 87178291200.0);                 //  1/14!
    pc = fma(pc, h2, -1.0 / 479001600.0);                  // -1/12!
    pc = fma(pc, h2, 1.0 / 3628800.0);                     //  1/10!
    pc = fma(pc, h2, -1.0 / 40320.0);                      // -1/8!
    pc = fma(pc, h2, 1.0 / 720.0);                         //  1/6!
    pc = fma(pc, h2, -1.0 / 24.0);                         // -1/4!
    pc = fma(pc, h2, 0.5);                                 //  1/2!
    const double sin_h = fma(-(h * h2), ps, h);            // h - h^3 (1/3! - h^2/5! + ...)
    const double cos_h = fma(-h2, pc, 1.0);                // 1 - h^2 (1/2! - h^2/4! + ...)
    sn = cos_h;
    cs = -sin_h;
}

// 1/sqrt(x) to ~1 ulp: hardware estimate + two Newton steps (x is a squared distance, far from
// the subnormal / overflow ranges the libm version guards against)
__device__ __forceinline__ double fast_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * fma(-hx * y, y, 1.5);
    y = y * fma(-hx * y, y, 1.5);
    return y;
}

// ---- table-driven sincos / exp ------------------------------------------------------------
// A double-precision polynomial constant is not an inline operand on gfx950: the compiler builds each in
// a register pair and hoists it out of the loop, and a pair loop that evaluates sincos and exp then
// carries ~70 VGPRs of constants (or spills them).  The Behler kernels keep the constants in LDS instead
// and read each where it is used (a broadcast ds_read, off the VALU).
//   [0,10)  sin series -1/21! .. 1/3!    [10,21) cos series 1/22! .. 1/2!    [21] pi/2
//   [22] log2(e)   [23] ln2 high part   [24] ln2 low part   [25,36) 1/13! .. 1/3!
#define ANNP_MTAB 36
__constant__ double annp_mtab[ANNP_MTAB] = {
    -1.0 / 51090942171709440000.0, 1.0 / 121645100408832000.0, -1.0 / 355687428096000.0, 1.0 / 1307674368000.0,
    -1.0 / 6227020800.0, 1.0 / 39916800.0, -1.0 / 362880.0, 1.0 / 5040.0, -1.0 / 120.0, 1.0 / 6.0,
    1.0 / 1124000727777607680000.0, -1.0 / 2432902008176640000.0, 1.0 / 6402373705728000.0, -1.0 / 20922789888000.0,
    1.0 / 87178291200.0, -1.0 / 479001600.0, 1.0 / 3628800.0, -1.0 / 40320.0, 1.0 / 720.0, -1.0 / 24.0, 0.5,
    1.57079632679489661923,
    1.4426950408889634074, 6.93147180369123816490e-01, 1.90821492927058770002e-10,
    1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0,
    1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0};

// every wave of the block writes the same values: no ordering between waves is needed
__device__ __forceinline__ void mtab_fill(double *T, int lane)
{
    if (lane < ANNP_MTAB) T[lane] = annp_mtab[lane];
}

// sincos_0_pi with the coefficients read from T
__device__ __forceinline__ void sincos_0_pi_tab(double t, const double *T, double &sn, double &cs)
{
    const double h = t - T[21];
    const double h2 = h * h;
    double ps = T[0];
#pragma unroll
    for (int k = 1; k < 10; k++) ps = fma(ps, h2, T[k]);
    double pc = T[10];
#pragma unroll
    for (int k = 11; k < 21; k++) pc = fma(pc, h2, T[k]);
    sn = fma(-h2, pc, 1.0);                 // cos h
    cs = -fma(-(h * h2), ps, h);            // -sin h
}

// exp(x) for x <= 0 (as far down as the subnormals): x = k ln2 + r, |r| <= ln2/2, degree-13 Taylor
// (truncation 4e-18), scaled by 2^k
__device__ __forceinline__ double exp_neg_tab(double x, const double *T)
{
    const double kf = rint(x * T[22]);
    double r = fma(-kf, T[23], x);
    r = fma(-kf, T[24], r);
    double p = T[25];
#pragma unroll
    for (int k = 26; k < ANNP_MTAB; k++) p = fma(p, r, T[k]);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)kf);
}

// ---- the same functions with the coefficients in scalar registers --------------------------------------------------
// The LDS table above keeps the constants out of the vector registers, but every coefficient costs an LDS instruction (a
// 64-lane broadcast read: 2 cycles of the CU's one LDS pipe), and in the Behler force pass those broadcasts were 75 % of all
// LDS instructions of a pass that the LDS pipe bounds.  Here the coefficients come from constant memory through the
// scalar unit (s_load_dwordx8/x16: eight doubles per instruction, served by the scalar cache) and enter the Horner steps
// as the SGPR operand of v_fma_f64.  Two things the compiler does not do by itself: (1) it hoists such loads out of the
// surrounding loops and then spills the scalar registers, so the table's address is made opaque where it is used (an empty
// volatile asm on an "s" operand: still known to be uniform, no longer loop-invariant); (2) it turns fma(p, x, c) with c in
// scalar registers into v_fmac_f64 on a vector copy of c -- two v_mov_b32 per coefficient -- so the step is written as the
// VOP3 form it should have picked.
typedef const double __attribute__((address_space(4))) *annp_cptr;

__device__ __forceinline__ annp_cptr mtab_scalar()
{
    unsigned long long v = (unsigned long long)(const void *)annp_mtab;
    asm volatile("" : "+s"(v));
    return (annp_cptr)v;
}

// a * b + c with c in scalar registers
__device__ __forceinline__ double fma_vvs(double a, double b, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}

__device__ __forceinline__ void sincos_0_pi_s(double t, double &sn, double &cs)
{
    const annp_cptr T = mtab_scalar();
    const double h = t - T[21];
    const double h2 = h * h;
    double ps = T[0];
#pragma unroll
    for (int k = 1; k < 10; k++) ps = fma_vvs(ps, h2, T[k]);
    double pc = T[10];
#pragma unroll
    for (int k = 11; k < 21; k++) pc = fma_vvs(pc, h2, T[k]);
    sn = fma(-h2, pc, 1.0);                 // cos h
    cs = -fma(-(h * h2), ps, h);            // -sin h
}

__device__ __forceinline__ double exp_neg_s(double x)
{
    const annp_cptr T = mtab_scalar();
    const double kf = rint(x * T[22]);
    double r = fma(-kf, T[23], x);
    r = fma(-kf, T[24], r);
    double p = T[25];
#pragma unroll
    for (int k = 26; k < ANNP_MTAB; k++) p = fma_vvs(p, r, T[k]);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)kf);
}

// two arguments at once: the coefficient loads are shared and the two Horner chains interleave
__device__ __forceinline__ void sincos_0_pi_s2(double t0, double t1, double &sn0, double &cs0, double &sn1, double &cs1)
{
    const annp_cptr T = mtab_scalar();
    const double h0 = t0 - T[21], h1 = t1 - T[21];
    const double a0 = h0 * h0, a1 = h1 * h1;
    double ps0 = T[0], ps1 = T[0];
#pragma unroll
    for (int k = 1; k < 10; k++) { ps0 = fma_vvs(ps0, a0, T[k]); ps1 = fma_vvs(ps1, a1, T[k]); }
    double pc0 = T[10], pc1 = T[10];
#pragma unroll
    for (int k = 11; k < 21; k++) { pc0 = fma_vvs(pc0, a0, T[k]); pc1 = fma_vvs(pc1, a1, T[k]); }
    sn0 = fma(-a0, pc0, 1.0); sn1 = fma(-a1, pc1, 1.0);
    cs0 = -fma(-(h0 * a0), ps0, h0); cs1 = -fma(-(h1 * a1), ps1, h1);
}

__device__ __forceinline__ void exp_neg_s2(double x0, double x1, double &e0, double &e1)
{
    const annp_cptr T = mtab_scalar();
    const double k0 = rint(x0 * T[22]), k1 = rint(x1 * T[22]);
    double r0 = fma(-k0, T[23], x0), r1 = fma(-k1, T[23], x1);
    r0 = fma(-k0, T[24], r0); r1 = fma(-k1, T[24], r1);
    double p0 = T[25], p1 = T[25];
#pragma unroll
    for (int k = 26; k < ANNP_MTAB; k++) { p0 = fma_vvs(p0, r0, T[k]); p1 = fma_vvs(p1, r1, T[k]); }
    p0 = fma(p0, r0, 0.5); p1 = fma(p1, r1, 0.5);
    p0 = fma(p0, r0, 1.0); p1 = fma(p1, r1, 1.0);
    p0 = fma(p0, r0, 1.0); p1 = fma(p1, r1, 1.0);
    e0 = __builtin_ldexp(p0, (int)k0); e1 = __builtin_ldexp(p1, (int)k1);
}

// fast_rsqrt written so that its only constant (0.5) is an inline operand
__device__ __forceinline__ double fast_rsqrt_ic(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = fma(y, fma(-hx * y, y, 0.5), y);
    y = fma(y, fma(-hx * y, y, 0.5), y);
    return y;
}

// Sum over the 64 lanes of a wave (all of them active), returned to every lane.  Data-parallel-primitive moves instead of
// __shfl_xor (which is ds_bpermute on this target: an LDS round trip per level, 12 per double): four row_shr steps leave each
// row of 16 lanes' total in its last lane, row_bcast:15 and row_bcast:31 carry them to lane 63, v_readlane hands the result
// out through scalar registers.  The moves are 32-bit: a double goes as two halves.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_from(double v)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(u & 0xffffffffull), CTRL, ROW_MASK, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(u >> 32), CTRL, ROW_MASK, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);      // lanes without a source read 0.0
}

// sum over each row of 16 lanes, left in the row's last lane (lanes 15, 31, 47, 63); the other lanes hold partial sums
__device__ __forceinline__ double row16_sum_to_last(double v)
{
    v += dpp_from<0x111, 0xf>(v);
    v += dpp_from<0x112, 0xf>(v);
    v += dpp_from<0x114, 0xf>(v);
    v += dpp_from<0x118, 0xf>(v);
    return v;
}

__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_from<0x111, 0xf>(v);       // row_shr:1
    v += dpp_from<0x112, 0xf>(v);       // row_shr:2
    v += dpp_from<0x114, 0xf>(v);       // row_shr:4
    v += dpp_from<0x118, 0xf>(v);       // row_shr:8   -> lane 15 of each row holds the row's sum
    v += dpp_from<0x142, 0xa>(v);       // row_bcast:15 into rows 1 and 3
    v += dpp_from<0x143, 0xc>(v);       // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's sum
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(u & 0xffffffffull), 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(u >> 32), 63);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// ---- the global virial ---------------------------------------------------------------------------------------------------
// Six sums over all atoms.  One atomic per wave on six fixed addresses is served one after the other by a single L2 channel:
// with a million waves the force passes took ten times as long with the virial as without (7.9 -> 75 ms per 1 M Fe atoms,
// 0.9 -> 9.3 ms per 512 k Ni atoms; NPT tallies it every step).  The kernels add into one of ANNP_VSLOTS rows of a scratch
// table instead (64 bytes apart, the row chosen by workgroup), and annp_virial_fold adds the rows into the caller's six
// doubles behind the force pass.
#define ANNP_VSLOTS 1024
__device__ __forceinline__ double *virial_row(double *table) { return table + 8 * (blockIdx.x & (ANNP_VSLOTS - 1)); }

__global__ __launch_bounds__(1024) void annp_virial_fold(const double *table, double *virial)
{
    const int k = threadIdx.x & 7, part = threadIdx.x >> 3;          // 128 threads per component
    if (k >= 6) return;
    double s = 0.0;
    for (int r = part; r < ANNP_VSLOTS; r += 128) s += table[8 * r + k];
    atomicAdd(&virial[k], s);
}

}  // namespace annp
