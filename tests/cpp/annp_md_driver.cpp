// annp_md_driver.cpp -- a C++ host running MD steps on the device through the C ABI alone (include/annp_hip.h):
// what LAMMPS does around Pair::compute for the reference -- Verlet::run's order of FixNVE::initial_integrate,
// Comm::forward_comm (here: the periodic images of one rank), force_clear, Pair::compute, Comm::reverse_comm,
// FixNVE::final_integrate, with a reneighbouring every `every` steps -- with no Python and no torch in the process.
// TEST DRIVER: built on the CPU (g++ + the HIP runtime API for device memory), run on the GPU box by
// tests/test_gpu_cpp_md.py, which checks the first energy against the oracle and the conservation of the total.
//
//   annp_md_driver <potential.ann> <element> <cells> <steps> <dt_ps> <every> <out.txt>
// bcc lattice a = 2.8553 A, cells^3 x 2 atoms, displacements from a counter-based generator, velocities zero at the start
// (the lattice relaxes: potential energy turns into kinetic energy and back).
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/annp_hip.h"
#include "../../meng_zhang_amd/host/annp_pair.h"

#define HIP_OK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 20; } \
    } while (0)
#define ABI_OK(call)                                                                                   \
    do {                                                                                               \
        int rc_ = (call);                                                                              \
        if (rc_ != 0) { std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, annp_hip_last_error(h)); return 21; } \
    } while (0)

static uint64_t splitmix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct Plan {       // periodic images of the owned atoms within rc of the box, grouped by owner
    std::vector<int> root, seg_dst, seg_start, perm;
    std::vector<double> shift;
};

static Plan plan_images(const std::vector<double> &x, int n, const double L[3], double rc)
{
    Plan p;
    p.seg_start.push_back(0);
    for (int i = 0; i < n; i++) {
        int mine = 0;
        for (int sx = -1; sx <= 1; sx++)
            for (int sy = -1; sy <= 1; sy++)
                for (int sz = -1; sz <= 1; sz++) {
                    if (!sx && !sy && !sz) continue;
                    const int s[3] = {sx, sy, sz};
                    bool ok = true;
                    for (int d = 0; d < 3 && ok; d++) {
                        const double q = x[3 * (size_t)i + d] + s[d] * L[d];
                        ok = q >= -rc && q < L[d] + rc;
                    }
                    if (!ok) continue;
                    p.root.push_back(i);
                    for (int d = 0; d < 3; d++) p.shift.push_back(s[d] * L[d]);
                    p.perm.push_back((int)p.perm.size());
                    mine++;
                }
        if (mine) { p.seg_dst.push_back(i); p.seg_start.push_back((int)p.root.size()); }
    }
    return p;
}

template <typename T>
static T *to_device(const std::vector<T> &v)
{
    T *d = nullptr;
    if (hipMalloc((void **)&d, sizeof(T) * (v.size() ? v.size() : 1)) != hipSuccess) return nullptr;
    if (!v.empty() && hipMemcpy(d, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

int main(int argc, char **argv)
{
    if (argc < 8) { std::fprintf(stderr, "usage: annp_md_driver pot El cells steps dt every out.txt\n"); return 1; }
    const std::string potfile = argv[1], elem = argv[2];
    const int cells = std::atoi(argv[3]), steps = std::atoi(argv[4]), every = std::atoi(argv[6]);
    const double dt = std::atof(argv[5]);
    const double a0 = 2.8553, rc_list = 8.5, mass = 55.847, ftm2v = 1.0 / 1.0364269e-4, mvv2e = 1.0364269e-4;
    const double L[3] = {cells * a0, cells * a0, cells * a0};
    const int n = 2 * cells * cells * cells;

    // ---- pair style: the host-side mirror of PairANNP (parser + annp_hip_init), then only its handle is used
    annp_host::PairANNP pair(1);
    const char *cf[4] = {"*", "*", potfile.c_str(), elem.c_str()};
    if (pair.settings(0, nullptr) || pair.coeff(4, cf) || pair.init_style(1, 0)) { std::fprintf(stderr, "%s\n", pair.error().c_str()); return 10; }
    annp_hip_handle *h = pair.handle();

    // ---- atoms
    std::vector<double> x((size_t)n * 3);
    int k = 0;
    for (int ix = 0; ix < cells; ix++)
        for (int iy = 0; iy < cells; iy++)
            for (int iz = 0; iz < cells; iz++)
                for (int b = 0; b < 2; b++, k++) {
                    const double base[3] = {(ix + 0.5 * b) * a0, (iy + 0.5 * b) * a0, (iz + 0.5 * b) * a0};
                    for (int d = 0; d < 3; d++) {
                        const double u = (double)(splitmix64((uint64_t)(3 * k + d) ^ 12345ull) >> 11) * (1.0 / 9007199254740992.0);
                        x[3 * (size_t)k + d] = base[d] + (2.0 * u - 1.0) * 0.05;
                    }
                }
    hipStream_t s;
    HIP_OK(hipSetDevice(0));
    HIP_OK(hipStreamCreate(&s));
    double *d_x = nullptr, *d_v = nullptr, *d_f = nullptr, *d_eng = nullptr, *d_shift = nullptr;
    int *d_root = nullptr, *d_dst = nullptr, *d_start = nullptr, *d_perm = nullptr;
    HIP_OK(hipMalloc((void **)&d_v, sizeof(double) * 3 * n));
    HIP_OK(hipMemset(d_v, 0, sizeof(double) * 3 * n));
    HIP_OK(hipMalloc((void **)&d_eng, sizeof(double)));
    const int *p_num = nullptr, *p_neigh = nullptr;
    const long long *p_first = nullptr;
    int mx = 0, nimg = 0, nall = n, nseg = 0;
    size_t cap_all = 0;
    const double dtf = 0.5 * dt * ftm2v / mass;
    std::vector<double> hv((size_t)n * 3), hx((size_t)n * 3);
    FILE *out = std::fopen(argv[7], "w");
    if (!out) return 2;

    auto replan = [&]() -> int {    // Comm::borders of one rank + Neighbor::build: wrap, images, buffers, device list
        for (int i = 0; i < n; i++)
            for (int d = 0; d < 3; d++) x[3 * (size_t)i + d] -= std::floor(x[3 * (size_t)i + d] / L[d]) * L[d];
        const Plan p = plan_images(x, n, L, rc_list);
        nimg = (int)p.root.size();
        nall = n + nimg;
        if ((size_t)nall > cap_all) {
            if (d_x) { (void)hipFree(d_x); (void)hipFree(d_f); }
            cap_all = (size_t)nall + nall / 8;
            HIP_OK(hipMalloc((void **)&d_x, sizeof(double) * 3 * cap_all));
            HIP_OK(hipMalloc((void **)&d_f, sizeof(double) * 3 * cap_all));
        }
        HIP_OK(hipMemcpy(d_x, x.data(), sizeof(double) * 3 * n, hipMemcpyHostToDevice));
        for (void *q : {(void *)d_root, (void *)d_dst, (void *)d_start, (void *)d_perm, (void *)d_shift}) if (q) (void)hipFree(q);
        d_root = to_device(p.root); d_dst = to_device(p.seg_dst); d_start = to_device(p.seg_start); d_perm = to_device(p.perm);
        d_shift = to_device(p.shift);
        if (!d_root || !d_dst || !d_start || !d_perm || !d_shift) return 22;
        // images from the wrapped positions, forces cleared
        ABI_OK(annp_hip_halo_unpack_images(h, nimg, d_root, d_shift, d_x, n, d_f, 3ll * nall, d_eng, s));
        ABI_OK(annp_hip_neigh_build_device(h, n, nall, d_x, rc_list, &p_num, &p_first, &p_neigh, &mx, s));
        nseg = (int)p.seg_dst.size();
        return 0;
    };
    if (int rc = replan()) return rc;
    auto evaluate = [&]() -> int {
        ABI_OK(annp_hip_compute_device(h, n, nall, d_x, nullptr, nullptr, p_num, p_first, p_neigh, mx, d_f, nullptr, d_eng, nullptr, nullptr, s));
        ABI_OK(annp_hip_reverse_fold(h, nseg, d_dst, d_start, d_perm, d_f + 3 * (size_t)n, d_f, s));
        return 0;
    };
    if (int rc = evaluate()) return rc;
    for (int step = 0; step <= steps; step++) {
        // thermo: E_pair and the kinetic energy of the owned atoms
        double e = 0.0;
        HIP_OK(hipMemcpyAsync(&e, d_eng, sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_OK(hipMemcpyAsync(hv.data(), d_v, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
        ABI_OK(annp_hip_sync(h));
        double ke = 0.0;
        for (double vv : hv) ke += vv * vv;
        ke *= 0.5 * mvv2e * mass;
        std::fprintf(out, "%d %.12f %.12f %d\n", step, e, ke, nall - n);
        if (step == steps) break;
        ABI_OK(annp_hip_verlet_half(h, n, d_x, d_v, d_f, dtf, dt, s));                           // FixNVE::initial_integrate
        if (every > 0 && (step + 1) % every == 0) {                                              // reneighbouring step
            HIP_OK(hipMemcpyAsync(x.data(), d_x, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, s));
            HIP_OK(hipStreamSynchronize(s));
            if (int rc = replan()) return rc;
        } else {
            ABI_OK(annp_hip_halo_unpack_images(h, nimg, d_root, d_shift, d_x, n, d_f, 3ll * nall, d_eng, s));   // forward_comm + force_clear
        }
        if (int rc = evaluate()) return rc;                                                      // Pair::compute + reverse_comm
        ABI_OK(annp_hip_verlet_half(h, n, nullptr, d_v, d_f, dtf, 0.0, s));                      // FixNVE::final_integrate
    }
    std::fclose(out);
    return 0;
}
