// step_kernels.hpp -- what an MD step does around the force evaluation, on the device.
//
// In the reference these are LAMMPS core's jobs around Pair::compute (annp-gpu-lammps/fe_v2/src/pair_annp.cpp:199 writes
// ghost forces and relies on them): Comm::forward_comm packs the positions of boundary atoms (+ the periodic shift) and
// unpacks them into the ghost rows, Comm::reverse_comm carries ghost forces back and adds them to their owners,
// FixNVE::initial_integrate / final_integrate are the two velocity-Verlet halves.  All of it is streaming work over
// [n][3] doubles and a few index arrays -- HBM-bound, a few microseconds per launch at 10^5 atoms -- so the point of
// writing it here is launch count (one kernel per job, nothing per-element on the host) and reproducibility: the fold is a
// gather over a root-sorted index (fixed summation order, no atomics), so a step gives the same bits every time.
#pragma once
#include "annp_common.hpp"

namespace annp {

// out[k] = x[idx[k]] + shift[k]      (k < n; rows of 3 doubles)
// forward_comm's pack (out = send buffer) and the fill of local periodic images (out = the image rows of x itself:
// images never serve as roots, so reading x while writing its image rows does not race)
__global__ __launch_bounds__(256) void annp_gather_shift(int n, const int *__restrict__ idx, const double *__restrict__ shift,
                                                         const double *x, double *out)
{
    const int t = blockIdx.x * 256 + threadIdx.x;       // one thread per (row, component)
    if (t >= 3 * n) return;
    const int k = t / 3, c = t - 3 * k;
    out[t] = x[3 * (size_t)idx[k] + c] + shift[t];
}

// the same fill for the image rows, and in the same launch the force array and the energy word are cleared for the
// evaluation that follows (force_clear)
__global__ __launch_bounds__(256) void annp_images_clear(int nimg, const int *__restrict__ root, const double *__restrict__ shift,
                                                         double *x, long long first_row, double *f, long long nclear, double *eng)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < 3ll * nimg) {
        const int k = (int)(t / 3), c = (int)(t - 3 * k);
        x[3 * (first_row + k) + c] = x[3 * (size_t)root[k] + c] + shift[t];
    }
    if (f && t < nclear) f[t] = 0.0;
    if (eng && t == 0) *eng = 0.0;
}

// f[dst[s]] += sum over k in [start[s], start[s+1]) of src[perm[k]]   (rows of 3 doubles, k ascending: a fixed order)
// reverse_comm's unpack: forces of local images onto their roots (src = the image rows of f), and the rows that came
// back over the wire onto the boundary atoms they belong to (src = receive buffer).  One thread per (segment, component).
__global__ __launch_bounds__(256) void annp_segment_add(int nseg, const int *__restrict__ dst, const int *__restrict__ start,
                                                        const int *__restrict__ perm, const double *src, double *f)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= 3 * nseg) return;
    const int s = t / 3, c = t - 3 * s;
    const int k0 = start[s], k1 = start[s + 1];
    if (k0 == k1) return;                        // (dst == nullptr: segment s belongs to row s, most of them empty)
    const size_t row = dst ? (size_t)dst[s] : (size_t)s;
    double acc = f[3 * row + c];
    for (int k = k0; k < k1; k++) acc += src[3 * (size_t)perm[k] + c];
    f[3 * row + c] = acc;
}

// velocity-Verlet half step: v += dtf f; then x += dt v when dt != 0 (FixNVE::initial_integrate), v only otherwise
// (final_integrate).  dtf = 0.5 dt ftm2v / mass.
__global__ __launch_bounds__(256) void annp_verlet_half(long long n3, double *x, double *v, const double *__restrict__ f, double dtf, double dt)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n3) return;
    {
#pragma clang fp contract(off)      // product and sum rounded separately: the same bits as a plain `v + dtf * f` on any host
        const double vn = v[t] + dtf * f[t];
        v[t] = vn;
        if (dt != 0.0) x[t] = x[t] + dt * vn;
    }
}

// The global virial as LAMMPS' Pair::virial_fdotr_compute has it (the route the reference takes under fix npt: fe_v2/src/pair_annp.cpp:216,
// `if (vflag_fdotr) virial_fdotr_compute()`): W = sum over owned atoms and ghosts of x_i (x) f_i, with f the forces of THIS evaluation
// only -- they were gathered in a scratch array `fs`, which this kernel also adds onto the caller's `f` (the boundary accumulates into
// f).  xx yy zz xy xz yz.  A block's six sums go to one row of the evaluation's virial table (annp_common.hpp), folded by
// annp_virial_fold.  120 bytes per atom streamed once: 50 us per 1.25 M atoms, where the pairwise tally inside the force pass cost it 6 %.
__global__ __launch_bounds__(256) void annp_fdotr_add(int nall, const double *__restrict__ x, const double *__restrict__ fs, double *f, double *vtable)
{
    __shared__ double part[4][6];
    ANNP_POISON();
    double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nall; i += (long long)gridDim.x * 256) {
        const double f0 = fs[3 * i], f1 = fs[3 * i + 1], f2 = fs[3 * i + 2];
        if (f0 != 0.0 || f1 != 0.0 || f2 != 0.0) {
            const double x0 = x[3 * i], x1 = x[3 * i + 1], x2 = x[3 * i + 2];
            f[3 * i] += f0; f[3 * i + 1] += f1; f[3 * i + 2] += f2;
            v[0] = fma(x0, f0, v[0]); v[1] = fma(x1, f1, v[1]); v[2] = fma(x2, f2, v[2]);
            v[3] = fma(x0, f1, v[3]); v[4] = fma(x0, f2, v[4]); v[5] = fma(x1, f2, v[5]);
        }
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const double t = wave_sum(v[k]);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][k] = t;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const double t = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        if (t != 0.0) atomicAdd(&virial_row(vtable)[threadIdx.x], t);
    }
}

}  // namespace annp
