// annp_gpu_compat.cpp -- the five functions a LAMMPS GPU-package pair style for this potential binds
// (include/annp_gpu_compat.h), on top of the C ABI (include/annp_hip.h).
//
// Mirrors annp-gpu-lammps/fe_v2/lib/lal_annp_ext.cpp: one object per process (:20), init clears what a previous init
// left (:33), clear is idempotent (:94-96); what init/compute do with their arguments follows
// fe_v2/lib/lal_annp.cpp:41-216 (init) and :259-370 / :376-498 (the two compute overloads) -- argument for argument,
// not line for line: there is no Geryon device object, no atom-chunk loop and no host-side force copy loop here.
#include "../../../include/annp_gpu_compat.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/annp_hip.h"

namespace {

enum { GPU_FORCE = 0, GPU_NEIGH = 1 };      // LAMMPS src/GPU/gpu_extra.h

struct Singleton {
    annp_hip_handle *h = nullptr;
    int ntypes = 0;
    double cutneigh = 0.0;                  // cell_size = cutmax + skin (pair_annp_gpu.cpp:218)
    // staging for callers whose double** rows are not one contiguous block
    std::vector<double> xflat, fflat, vflat;
    // the list handed back by annp_gpu_compute_n
    std::vector<int> ilist, jnum, neigh;
    std::vector<long long> first;
    std::vector<int *> firstneigh;
    bool list_current = false;
} S;

// $ANNP_HIP_DEVICE, else the node-local MPI rank modulo the number of cards (ranks : GPUs = n : 1 is allowed, as in
// the reference's Device object), else 0
int pick_device()
{
    if (const char *e = std::getenv("ANNP_HIP_DEVICE")) return std::atoi(e);
    const int ndev = annp_hip_device_count();
    for (const char *name : {"OMPI_COMM_WORLD_LOCAL_RANK", "MV2_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID", "SLURM_LOCALID", "LOCAL_RANK"})
        if (const char *e = std::getenv(name)) return ndev > 0 ? std::atoi(e) % ndev : 0;
    return 0;
}

// rows of a LAMMPS 2-d array (memory->create) are one block: a[0] is the flat array.  Anything else is gathered.
bool rows_contiguous(double **a, int n, int w)
{
    for (int i = 1; i < n; i++) if (a[i] != a[0] + (size_t)i * w) return false;
    return true;
}

const double *flat_in(double **a, int n, int w, std::vector<double> &tmp)
{
    if (n <= 0) return nullptr;
    if (rows_contiguous(a, n, w)) return a[0];
    tmp.resize((size_t)n * w);
    for (int i = 0; i < n; i++) std::memcpy(tmp.data() + (size_t)i * w, a[i], sizeof(double) * w);
    return tmp.data();
}

int init_common(const int ntypes, const int inum, const int nall, const int max_nbors, const double cell_size,
                int &gpu_mode, FILE *screen, const int ntl, const int nhl, const int nnod, const int nsf,
                const int npsf, const int ntsf, const double e_scale, const double e_shift, const double e_atom,
                const int flagsym, int *flagact, double *scal, double *avg, double **host_cutsq, int *host_map,
                double ***host_weight_all, double ***host_bias_all, double **cofsymrad, double **cofsymang)
{
    annp_gpu_clear();                                                   // lal_annp_ext.cpp:33
    const int n = ntypes + 1;
    int nelements = 1;
    for (int t = 1; t <= ntypes; t++) if (host_map[t] + 1 > nelements) nelements = host_map[t] + 1;
    std::vector<double> cutsq((size_t)n * n, 0.0);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) cutsq[(size_t)i * n + j] = (i && j) ? host_cutsq[i][j] : 0.0;
    const int nl = ntl - 1;
    std::vector<const double *> wp((size_t)nelements * nl), bp((size_t)nelements * nl);
    for (int e = 0; e < nelements; e++)
        for (int l = 0; l < nl; l++) { wp[(size_t)e * nl + l] = host_weight_all[e][l]; bp[(size_t)e * nl + l] = host_bias_all[e][l]; }
    std::vector<double> rad, ang;
    if (cofsymrad && cofsymang) {
        for (int i = 0; i < npsf; i++) rad.insert(rad.end(), cofsymrad[i], cofsymrad[i] + 3);
        for (int i = 0; i < ntsf; i++) ang.insert(ang.end(), cofsymang[i], cofsymang[i] + 4);
    }
    annp_hip_params prm;
    std::memset(&prm, 0, sizeof(prm));
    prm.struct_bytes = (int)sizeof(prm);
    prm.descriptor = rad.empty() ? ANNP_HIP_DESC_CHEBYSHEV : ANNP_HIP_DESC_BEHLER;
    prm.ntypes = ntypes; prm.nelements = nelements;
    prm.ntl = ntl; prm.nhl = nhl; prm.nnod = nnod; prm.nsf = nsf; prm.npsf = npsf; prm.ntsf = ntsf;
    prm.flagsym = flagsym; prm.ni_compat = 0;                           // the reference GPU kernel's derivative (ni/lib/lal_annp.cu:409-414)
    prm.flagact = flagact;
    prm.e_scale = e_scale; prm.e_shift = e_shift; prm.e_atom = e_atom;
    double cmax = 0.0;
    for (double c : cutsq) if (c > cmax) cmax = c;
    prm.cut = std::sqrt(cmax);                                          // cutmax: init_one returns it for every pair (fe_v2:323-327)
    prm.sfnor_scal = scal; prm.sfnor_avg = avg;
    prm.cutsq = cutsq.data(); prm.map = host_map;
    prm.weight_all = wp.data(); prm.bias_all = bp.data();
    prm.cofsymrad = rad.empty() ? nullptr : rad.data();
    prm.cofsymang = ang.empty() ? nullptr : ang.data();
    const int device = pick_device();
    const int rc = annp_hip_init(&S.h, &prm, device, inum, nall, max_nbors);
    if (rc != 0) {
        if (screen) std::fprintf(screen, "annp/hip: %s\n", annp_hip_last_error(nullptr));
        S.h = nullptr;
        return rc;                                                      // same numbering as lal_annp.h:28-33
    }
    S.ntypes = ntypes;
    S.cutneigh = cell_size;
    annp_hip_set_notice(S.h, screen);           // a change of kernel path (a system denser than the moment kernels take) is said once
    const char *nm = std::getenv("ANNP_HIP_NEIGH");
    gpu_mode = (nm && std::strcmp(nm, "host") == 0) ? GPU_FORCE : GPU_NEIGH;
    if (screen)
        std::fprintf(screen, "- Using acceleration for annp: libannp_hip (gfx950), device %d, neighbour list built on the %s\n",
                     device, gpu_mode == GPU_NEIGH ? "device" : "host");
    return 0;
}

}  // namespace

int annp_gpu_init(const int ntypes, const int inum, const int nall, const int max_nbors, const double cell_size,
                  int &gpu_mode, FILE *screen, const int ntl, const int nhl, const int nnod, const int nsf,
                  const int npsf, const int ntsf, const double e_scale, const double e_shift, const double e_atom,
                  const int flagsym, int *flagact, double *sfnor_scal, double *sfnor_avg, double **host_cutsq,
                  int *host_map, double ***host_weight_all, double ***host_bias_all)
{
    return init_common(ntypes, inum, nall, max_nbors, cell_size, gpu_mode, screen, ntl, nhl, nnod, nsf, npsf, ntsf, e_scale,
                       e_shift, e_atom, flagsym, flagact, sfnor_scal, sfnor_avg, host_cutsq, host_map, host_weight_all,
                       host_bias_all, nullptr, nullptr);
}

int annp_gpu_init(const int ntypes, const int inum, const int nall, const int max_nbors, const double cell_size,
                  int &gpu_mode, FILE *screen, const int ntl, const int nhl, const int nnod, const int nsf,
                  const int npsf, const int ntsf, const double e_scale, const double e_shift, const double e_atom,
                  const int flagsym, int *flagact, double *sf_scal, double *sf_min, double **host_cutsq,
                  int *host_map, double ***host_weight_all, double ***host_bias_all,
                  double **host_cofsymrad, double **host_cofsymang)
{
    return init_common(ntypes, inum, nall, max_nbors, cell_size, gpu_mode, screen, ntl, nhl, nnod, nsf, npsf, ntsf, e_scale,
                       e_shift, e_atom, flagsym, flagact, sf_scal, sf_min, host_cutsq, host_map, host_weight_all,
                       host_bias_all, host_cofsymrad, host_cofsymang);
}

void annp_gpu_clear()
{
    if (S.h) annp_hip_clear(S.h);
    S.h = nullptr;
    S.list_current = false;
}

double annp_gpu_bytes() { return annp_hip_bytes(S.h); }

namespace {

// shared tail of the two compute entries: marshal LAMMPS' row-pointer arrays to flat ones and back
template <typename Call>
bool run(double *eatom, double &eng_vdwl, double **f, int nall, double **host_x, bool eflag, bool ea_flag, bool va_flag,
         double **vatom, Call &&call)
{
    if (!S.h || nall <= 0) return S.h != nullptr;
    const double *x = flat_in(host_x, nall, 3, S.xflat);
    const bool f_flat = rows_contiguous(f, nall, 3);
    double *ff = f[0];
    if (!f_flat) { S.fflat.assign((size_t)nall * 3, 0.0); ff = S.fflat.data(); }
    const bool want_v = va_flag && vatom;
    bool v_flat = true;
    double *vv = nullptr;
    if (want_v) {
        v_flat = rows_contiguous(vatom, nall, 6);
        vv = vatom[0];
        if (!v_flat) { S.vflat.assign((size_t)nall * 6, 0.0); vv = S.vflat.data(); }
    }
    double e = 0.0;
    const int rc = call(x, ff, &e, (eflag && ea_flag) ? eatom : nullptr, vv);
    if (rc != 0) return false;
    if (!f_flat) for (int i = 0; i < nall; i++) for (int k = 0; k < 3; k++) f[i][k] += S.fflat[(size_t)i * 3 + k];
    if (want_v && !v_flat) for (int i = 0; i < nall; i++) for (int k = 0; k < 6; k++) vatom[i][k] += S.vflat[(size_t)i * 6 + k];
    if (eflag) eng_vdwl += e;
    return true;
}

}  // namespace

void annp_gpu_compute(double *eatom_annp, double &eng_vdwl_annp, double **f, const int ago, const int inum, const int nall,
                      const int nghost, double **host_x, int *host_type, int *ilist, int *numj, int **firstneigh,
                      const bool eflag, const bool vflag, const bool ea_flag, const bool va_flag, int &host_start,
                      const double /*cpu_time*/, bool &success, double **vatom_annp)
{
    host_start = inum;                                                  // everything on the accelerator (gpu_split = 1)
    // the global virial is virial_fdotr_compute()'s in the caller (pair_annp_gpu.cpp:126): not tallied here
    (void)vflag;
    success = run(eatom_annp, eng_vdwl_annp, f, nall, host_x, eflag, ea_flag, va_flag, vatom_annp,
                  [&](const double *x, double *ff, double *e, double *ea, double *vv) {
                      return annp_hip_compute(S.h, ago, inum, nall, nghost, x, host_type, ilist, numj, firstneigh,
                                              eflag ? 1 : 0, 0, ea ? 1 : 0, vv ? 1 : 0, ff, e, ea, nullptr, vv);
                  });
}

int **annp_gpu_compute_n(double *eatom_annp, double &eng_vdwl_annp, double **f, const int ago, const int inum, const int nall,
                         const int nghost, double **host_x, int *host_type, double *sublo, double *subhi, tagint * /*tag*/,
                         int ** /*nspecial*/, tagint ** /*special*/, const bool eflag, const bool vflag, const bool ea_flag,
                         const bool va_flag, int &host_start, int **ilist, int **jnum, const double /*cpu_time*/,
                         bool &success, double **vatom_annp)
{
    host_start = inum;
    (void)vflag;
    success = run(eatom_annp, eng_vdwl_annp, f, nall, host_x, eflag, ea_flag, va_flag, vatom_annp,
                  [&](const double *x, double *ff, double *e, double *ea, double *vv) {
                      return annp_hip_compute_n(S.h, ago, inum, nall, nghost, x, host_type, sublo, subhi, S.cutneigh,
                                                eflag ? 1 : 0, 0, ea ? 1 : 0, vv ? 1 : 0, ff, e, ea, nullptr, vv);
                  });
    // What lal_base_annp.cpp:159-175 hands back is ilist, jnum and firstneigh on the host.  The reference caller reads none
    // of it (host_start == inum: nothing is left for the CPU), and the reference library itself fills host rows only for
    // the host_inum atoms it leaves to the CPU, so by default only ilist and jnum come back (4 bytes per atom, once per
    // rebuild) and the firstneigh rows stay null.  ANNP_HIP_RETURN_LIST=1 copies the rows too (one device-to-host copy of
    // the whole list per rebuild, chunked through pinned staging: ~0.9 GB at 1 M atoms).
    if (success && (ago == 0 || !S.list_current || (int)S.ilist.size() != inum)) {
        S.ilist.resize((size_t)inum);
        for (int i = 0; i < inum; i++) S.ilist[i] = i;
        S.jnum.assign((size_t)inum, 0);
        S.first.assign((size_t)inum + 1, 0);
        S.firstneigh.assign((size_t)inum + 1, nullptr);
        const char *rl = std::getenv("ANNP_HIP_RETURN_LIST");
        const bool rows = rl && std::strcmp(rl, "1") == 0;
        long long total = 0;
        int rc = inum > 0 ? annp_hip_neigh_to_host(S.h, inum, S.jnum.data(), S.first.data(), nullptr, 0, &total) : 0;
        if (rc == 0 && rows && inum > 0) {
            S.neigh.resize((size_t)(total > 0 ? total : 1));
            rc = annp_hip_neigh_to_host(S.h, inum, S.jnum.data(), S.first.data(), S.neigh.data(), total, &total);
            for (int i = 0; rc == 0 && i < inum; i++) S.firstneigh[i] = S.neigh.data() + S.first[i];
        }
        if (rc != 0) success = false;
        S.list_current = success;
    }
    if (ilist) *ilist = S.ilist.data();
    if (jnum) *jnum = S.jnum.data();
    return S.firstneigh.data();
}
