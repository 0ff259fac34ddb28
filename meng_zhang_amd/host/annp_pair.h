// annp_pair.h -- host-side mirror of the reference pair style, without LAMMPS.
//
// annp_host::PairANNP has the same life cycle and argument meaning as
// LAMMPS_NS::PairANNP / PairANNPGPU (reference: fe_v2/src/pair_annp.h:24-32,
// fe_v2/src/pair_annp.cpp:249-327, fe_v2/src/pair_annp_gpu.cpp:81-237):
//     settings(narg, arg)  ->  coeff(narg, arg)  ->  init_style()  ->  init_one(i, j)
//     ->  compute(eflag, vflag) ...
// except that the LAMMPS objects it would reach through pointers (atom, list,
// force, neighbor) are passed in as plain arrays.  It owns no arithmetic: coeff()
// parses the potential file, init_style() flattens the parameters the way
// PairANNPGPU::init_style does and hands them to annp_hip_init, compute() marshals
// to annp_hip_compute.  The LAMMPS adaptor (pair_annp_hip.h) is a thin shell
// around this class; the `extern "C"` functions below expose it to ctypes.
#pragma once
#include <string>
#include <vector>

#include "../../include/annp_hip.h"
#include "annp_potential.h"

namespace annp_host {

class PairANNP {
   public:
    // style: "annp" (default) or "anna_adp" (reference: anna-gpu-lammps/bcc_fe/src/pair_anna_adp.{h,cpp},
    // same life cycle and pair_coeff syntax, a `.anna` potential file)
    explicit PairANNP(int ntypes, const char *style = "annp");
    ~PairANNP();
    PairANNP(const PairANNP &) = delete;
    PairANNP &operator=(const PairANNP &) = delete;

    // each returns 0 or a negative code and sets error(); texts follow the reference's error->all messages
    int settings(int narg, const char *const *arg);                 // fe:249-252: narg must be 0
    int coeff(int narg, const char *const *arg);                    // fe:257-304: "* * file El1 [El2 ...]"
    int init_style(int newton_pair, int device);                    // fe:309-318 + pair_annp_gpu.cpp:132-237
    double init_one(int i, int j);                                  // fe:323-327: cutmax (or <0 on error)

    // PairANNPGPU::compute, host neighbour list (pair_annp_gpu.cpp:81-127, gpu_mode == GPU_FORCE)
    int compute(int eflag, int vflag, int eflag_atom, int ago, int inum, int nall, int nghost,
                const double *x, const int *type, const int *ilist, const int *numneigh,
                const int *const *firstneigh, double *f, double *eng_vdwl, double *eatom, double *virial,
                double *vatom);
    // same with the device-built list (gpu_mode != GPU_FORCE)
    int compute_n(int eflag, int vflag, int eflag_atom, int ago, int inum, int nall, int nghost,
                  const double *x, const int *type, const double *sublo, const double *subhi, double cutneigh,
                  double *f, double *eng_vdwl, double *eatom, double *virial, double *vatom);

    double memory_usage() const;
    double cutmax() const { return cutmax_; }
    const Potential &potential() const { return pot_; }
    const std::string &error() const { return err_; }
    annp_hip_handle *handle() const { return handle_; }
    void set_ni_compat(int v) { ni_compat_ = v; }
    // files with several elements: 0 (default) = the reference parser's behaviour, 1 = "#El" lines select the element
    // of the blocks below them (annp_potential.h); call before coeff()
    void set_blocks_by_name(int v) { blocks_by_name_ = v != 0; }
    int map(int type) const { return (type >= 1 && type <= ntypes_) ? map_[type] : -1; }     // element of a LAMMPS type, -1 = not mapped
    bool behler() const { return pot_.has_symcoef; }
    bool anna() const { return anna_; }

   private:
    int ntypes_;
    bool anna_ = false;
    int ni_compat_ = 0;
    bool blocks_by_name_ = false;
    bool coeff_done_ = false;
    double cutmax_ = 0.0;
    std::vector<int> map_;                 // type -> element
    std::vector<int> setflag_;             // (ntypes+1)^2
    std::vector<double> cutsq_;            // (ntypes+1)^2
    std::vector<std::string> elements_;
    Potential pot_;
    annp_hip_handle *handle_ = nullptr;
    std::string err_;
    int fail(int code, const std::string &msg) { err_ = msg; return code; }
};

}  // namespace annp_host

extern "C" {
typedef struct annp_pair annp_pair;
annp_pair *annp_pair_create(int ntypes);
annp_pair *annp_pair_create_style(int ntypes, const char *style);   /* "annp" | "anna_adp"; NULL on an unknown style */
void annp_pair_destroy(annp_pair *p);
int annp_pair_settings(annp_pair *p, int narg, const char *const *arg);
int annp_pair_coeff(annp_pair *p, int narg, const char *const *arg);
int annp_pair_set_ni_compat(annp_pair *p, int on);
int annp_pair_set_blocks_by_name(annp_pair *p, int on);
int annp_pair_init_style(annp_pair *p, int newton_pair, int device);
double annp_pair_init_one(annp_pair *p, int i, int j);
int annp_pair_compute(annp_pair *p, int eflag, int vflag, int eflag_atom, int ago, int inum, int nall, int nghost,
                      const double *x, const int *type, const int *ilist, const int *numneigh,
                      const int *const *firstneigh, double *f, double *eng_vdwl, double *eatom, double *virial,
                double *vatom);
int annp_pair_compute_n(annp_pair *p, int eflag, int vflag, int eflag_atom, int ago, int inum, int nall, int nghost,
                        const double *x, const int *type, const double *sublo, const double *subhi, double cutneigh,
                        double *f, double *eng_vdwl, double *eatom, double *virial, double *vatom);
double annp_pair_memory_usage(const annp_pair *p);
const char *annp_pair_error(const annp_pair *p);
annp_hip_handle *annp_pair_handle(const annp_pair *p);
/* parsed potential, for tests: dims[8] = ntl nhl nnod nsf npsf ntsf flagsym has_symcoef;
 * scal[5] = cut e_scale e_shift e_atom mass */
/* anna_adp extras: nout, e_base, e_scal and the ngp analytic parameters (returns ngp, or <0) */
int annp_pair_potential_anna(const annp_pair *p, int *nout, double *e_base, double *e_scal, double *gparams, int max_gp);
int annp_pair_potential_info(const annp_pair *p, int *dims, double *scal, int *flagact, double *norm_a, double *norm_b);
int annp_pair_potential_layer(const annp_pair *p, int layer, double *w, double *b);
int annp_pair_potential_sym(const annp_pair *p, double *rad, double *ang);
}
