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

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// cutoff function and derivative, fe_v2/src/pair_annp.cpp:590-594
__device__ __forceinline__ void cutoff_fc(double r, double pi_over_rc, double &fc, double &dfc)
{
    double sn, cs;
    sincos(pi_over_rc * r, &sn, &cs);
    fc = 0.5 * (cs + 1.0);
    dfc = -0.5 * pi_over_rc * sn;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace annp
