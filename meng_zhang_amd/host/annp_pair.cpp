// annp_pair.cpp -- see annp_pair.h.
#include "annp_pair.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>

namespace annp_host {

PairANNP::PairANNP(int ntypes, const char *style) : ntypes_(ntypes < 1 ? 1 : ntypes), anna_(style && std::strcmp(style, "anna_adp") == 0)
{
    const size_t n = (size_t)ntypes_ + 1;
    map_.assign(n, -1);
    setflag_.assign(n * n, 0);
    cutsq_.assign(n * n, 0.0);
}

PairANNP::~PairANNP()
{
    if (handle_) annp_hip_clear(handle_);        // pair_annp_gpu.cpp:68-70
}

int PairANNP::settings(int narg, const char *const *)
{
    if (narg != 0) return fail(ANNP_HIP_EARG, "Illegal pair_style command");
    return 0;
}

int PairANNP::coeff(int narg, const char *const *arg)
{
    if (narg != 3 + ntypes_) return fail(ANNP_HIP_EARG, "Incorrect args for pair coefficients");
    if (std::strcmp(arg[0], "*") != 0 || std::strcmp(arg[1], "*") != 0)
        return fail(ANNP_HIP_EARG, "Incorrect args for pair coefficients");
    // element of each atom type, in order of first appearance (fe:267-279)
    elements_.clear();
    for (int i = 3; i < narg; ++i) {
        if (std::strcmp(arg[i], "") == 0) continue;
        size_t j = 0;
        for (; j < elements_.size(); ++j) if (elements_[j] == arg[i]) break;
        map_[i - 2] = (int)j;
        if (j == elements_.size()) elements_.push_back(arg[i]);
    }
    std::string perr;
    const bool ok = anna_ ? read_potential_anna(arg[2], (int)elements_.size(), pot_, perr)
                          : read_potential(arg[2], (int)elements_.size(), pot_, perr, blocks_by_name_ ? &elements_ : nullptr);
    if (!ok) return fail(ANNP_HIP_EARG, perr);
    if ((int)elements_.size() != (int)pot_.elements.size()) return fail(ANNP_HIP_EARG, "Incorrect args for pair coefficients");
    cutmax_ = pot_.cut;                                            // fe:291-292
    int count = 0;
    const int n = ntypes_ + 1;
    for (int i = 1; i <= ntypes_; ++i)
        for (int j = i; j <= ntypes_; ++j)
            if (map_[i] >= 0 && map_[j] >= 0) { setflag_[i * n + j] = 1; ++count; }
    if (count == 0) return fail(ANNP_HIP_EARG, "Incorrect args for pair coefficients");
    coeff_done_ = true;
    return 0;
}

double PairANNP::init_one(int i, int j)
{
    const int n = ntypes_ + 1;
    if (i < 1 || j < 1 || i > ntypes_ || j > ntypes_ || setflag_[std::min(i, j) * n + std::max(i, j)] == 0) {
        fail(ANNP_HIP_EARG, "All pair coeffs are not set");
        return -1.0;
    }
    return cutmax_;
}

int PairANNP::init_style(int newton_pair, int device)
{
    if (!coeff_done_) return fail(ANNP_HIP_EARG, "All pair coeffs are not set");
    if (newton_pair == 0)                                          // fe:311-312; adp:375-376
        return fail(ANNP_HIP_EARG, anna_ ? "Pair style Neural Network Potential requires newton pair on"
                                         : "Pair style annp/hip requires newton pair on");
    const int n = ntypes_ + 1;
    for (int i = 1; i <= ntypes_; ++i)                             // pair_annp_gpu.cpp:170-183
        for (int j = i; j <= ntypes_; ++j) {
            double c = 0.0;
            if (setflag_[i * n + j] != 0 || (setflag_[i * n + i] != 0 && setflag_[j * n + j] != 0)) {
                c = init_one(i, j);
                c *= c;
            }
            cutsq_[i * n + j] = cutsq_[j * n + i] = c;
        }
    const int nsf = pot_.nsf, nl = pot_.ntl - 1;
    std::vector<double> scal(nsf), avg(nsf);
    if (anna_) {
        // no normalisation in this pair style (adp:584-612)
    } else if (!pot_.has_symcoef) {                                       // pair_annp_gpu.cpp:207-216
        for (int k = 0; k < nsf; ++k) {
            const double t_avg = pot_.norm_b[k], t_cov = pot_.norm_a[k];
            const double t_scale = std::sqrt(t_cov - t_avg * t_avg);
            avg[k] = t_avg;
            scal[k] = (t_scale <= 1.0e-10) ? 0.0 : 1.0 / t_scale;
        }
    } else {                                                       // ni pair_annp_gpu.cpp:231-235
        for (int k = 0; k < nsf; ++k) { avg[k] = pot_.norm_a[k]; scal[k] = pot_.norm_b[k] - pot_.norm_a[k]; }
    }
    // weight_all[element][layer] as PairANNPGPU::init_style lays it out (pair_annp_gpu.cpp:185-205)
    const int ne = anna_ ? 1 : (int)pot_.elements.size();
    std::vector<const double *> wp((size_t)ne * nl), bp((size_t)ne * nl);
    for (size_t k = 0; k < wp.size(); ++k) { wp[k] = pot_.weights[k].data(); bp[k] = pot_.biases[k].data(); }

    annp_hip_params prm;
    std::memset(&prm, 0, sizeof(prm));
    prm.struct_bytes = (int)sizeof(prm);
    // the shipped Ni file still names "Chebyshev" (SURVEY.md 8a): the coefficient section decides
    prm.descriptor = anna_ ? ANNP_HIP_DESC_ANNA_ADP : pot_.has_symcoef ? ANNP_HIP_DESC_BEHLER : ANNP_HIP_DESC_CHEBYSHEV;
    prm.nout = pot_.nout; prm.ngp = (int)pot_.gparams.size(); prm.gparams = pot_.gparams.empty() ? nullptr : pot_.gparams.data();
    prm.e_base = pot_.e_base;
    prm.ntypes = ntypes_;
    prm.nelements = ne;
    prm.ntl = pot_.ntl; prm.nhl = pot_.nhl; prm.nnod = pot_.nnod;
    prm.nsf = nsf; prm.npsf = pot_.npsf; prm.ntsf = pot_.ntsf;
    prm.flagsym = pot_.flagsym;
    prm.ni_compat = ni_compat_;
    prm.flagact = pot_.flagact.data();
    prm.e_scale = pot_.e_scale; prm.e_shift = pot_.e_shift; prm.e_atom = pot_.e_atom;
    prm.cut = pot_.cut;
    prm.sfnor_scal = scal.data(); prm.sfnor_avg = avg.data();
    prm.cutsq = cutsq_.data();
    prm.map = map_.data();
    prm.weight_all = wp.data(); prm.bias_all = bp.data();
    prm.cofsymrad = pot_.has_symcoef ? pot_.sym_rad.data() : nullptr;
    prm.cofsymang = pot_.has_symcoef ? pot_.sym_ang.data() : nullptr;
    if (handle_) { annp_hip_clear(handle_); handle_ = nullptr; }   // lal_annp_ext.cpp:33
    const int rc = annp_hip_init(&handle_, &prm, device, 0, 0, 0);
    if (rc != 0) return fail(rc, annp_hip_last_error(nullptr));
    return 0;
}

int PairANNP::compute(int eflag, int vflag, int eflag_atom, int ago, int inum, int nall, int nghost,
                      const double *x, const int *type, const int *ilist, const int *numneigh,
                      const int *const *firstneigh, double *f, double *eng_vdwl, double *eatom, double *virial,
                      double *vatom)
{
    if (!handle_) return fail(ANNP_HIP_EARG, "pair style annp/hip used before init_style");
    const int rc = annp_hip_compute(handle_, ago, inum, nall, nghost, x, type, ilist, numneigh, firstneigh,
                                    eflag, vflag, eflag_atom, vatom ? 1 : 0, f, eng_vdwl, eatom, virial, vatom);
    if (rc == ANNP_HIP_ENOMEM) return fail(rc, "Insufficient memory on accelerator");   // pair_annp_gpu.cpp:122-123
    if (rc != 0) return fail(rc, annp_hip_last_error(handle_));
    return 0;
}

int PairANNP::compute_n(int eflag, int vflag, int eflag_atom, int ago, int inum, int nall, int nghost,
                        const double *x, const int *type, const double *sublo, const double *subhi, double cutneigh,
                        double *f, double *eng_vdwl, double *eatom, double *virial, double *vatom)
{
    if (!handle_) return fail(ANNP_HIP_EARG, "pair style annp/hip used before init_style");
    const int rc = annp_hip_compute_n(handle_, ago, inum, nall, nghost, x, type, sublo, subhi, cutneigh,
                                      eflag, vflag, eflag_atom, vatom ? 1 : 0, f, eng_vdwl, eatom, virial, vatom);
    if (rc == ANNP_HIP_ENOMEM) return fail(rc, "Insufficient memory on accelerator");
    if (rc != 0) return fail(rc, annp_hip_last_error(handle_));
    return 0;
}

double PairANNP::memory_usage() const
{
    double bytes = (double)(map_.size() * sizeof(int) + setflag_.size() * sizeof(int) + cutsq_.size() * sizeof(double));
    return bytes + annp_hip_bytes(handle_);                        // pair_annp_gpu.cpp:73-76
}

}  // namespace annp_host

// -------------------------------------------------------------------------------------
struct annp_pair {
    annp_host::PairANNP impl;
    explicit annp_pair(int ntypes, const char *style = "annp") : impl(ntypes, style) {}
};

extern "C" {

annp_pair *annp_pair_create(int ntypes) { return new (std::nothrow) annp_pair(ntypes); }
annp_pair *annp_pair_create_style(int ntypes, const char *style)
{
    if (!style || (std::strcmp(style, "annp") != 0 && std::strcmp(style, "anna_adp") != 0)) return nullptr;
    return new (std::nothrow) annp_pair(ntypes, style);
}
int annp_pair_potential_anna(const annp_pair *p, int *nout, double *e_base, double *e_scal, double *gparams, int max_gp)
{
    if (!p || !p->impl.potential().is_anna) return ANNP_HIP_EARG;
    const annp_host::Potential &q = p->impl.potential();
    if (nout) *nout = q.nout;
    if (e_base) *e_base = q.e_base;
    if (e_scal) *e_scal = q.e_scal;
    for (int k = 0; gparams && k < max_gp && k < (int)q.gparams.size(); k++) gparams[k] = q.gparams[k];
    return (int)q.gparams.size();
}
void annp_pair_destroy(annp_pair *p) { delete p; }
int annp_pair_settings(annp_pair *p, int narg, const char *const *arg) { return p ? p->impl.settings(narg, arg) : ANNP_HIP_EARG; }
int annp_pair_coeff(annp_pair *p, int narg, const char *const *arg) { return p ? p->impl.coeff(narg, arg) : ANNP_HIP_EARG; }
int annp_pair_set_ni_compat(annp_pair *p, int on) { if (!p) return ANNP_HIP_EARG; p->impl.set_ni_compat(on); return 0; }
int annp_pair_set_blocks_by_name(annp_pair *p, int on) { if (!p) return ANNP_HIP_EARG; p->impl.set_blocks_by_name(on); return 0; }
int annp_pair_init_style(annp_pair *p, int newton_pair, int device) { return p ? p->impl.init_style(newton_pair, device) : ANNP_HIP_EARG; }
double annp_pair_init_one(annp_pair *p, int i, int j) { return p ? p->impl.init_one(i, j) : -1.0; }
int annp_pair_compute(annp_pair *p, int eflag, int vflag, int eflag_atom, int ago, int inum, int nall, int nghost,
                      const double *x, const int *type, const int *ilist, const int *numneigh,
                      const int *const *firstneigh, double *f, double *eng_vdwl, double *eatom, double *virial,
                      double *vatom)
{
    return p ? p->impl.compute(eflag, vflag, eflag_atom, ago, inum, nall, nghost, x, type, ilist, numneigh, firstneigh, f, eng_vdwl, eatom, virial, vatom)
             : ANNP_HIP_EARG;
}
int annp_pair_compute_n(annp_pair *p, int eflag, int vflag, int eflag_atom, int ago, int inum, int nall, int nghost,
                        const double *x, const int *type, const double *sublo, const double *subhi, double cutneigh,
                        double *f, double *eng_vdwl, double *eatom, double *virial, double *vatom)
{
    return p ? p->impl.compute_n(eflag, vflag, eflag_atom, ago, inum, nall, nghost, x, type, sublo, subhi, cutneigh, f, eng_vdwl, eatom, virial, vatom)
             : ANNP_HIP_EARG;
}
double annp_pair_memory_usage(const annp_pair *p) { return p ? p->impl.memory_usage() : 0.0; }
const char *annp_pair_error(const annp_pair *p) { return p ? p->impl.error().c_str() : "null pair"; }
annp_hip_handle *annp_pair_handle(const annp_pair *p) { return p ? p->impl.handle() : nullptr; }

int annp_pair_potential_info(const annp_pair *p, int *dims, double *scal, int *flagact, double *norm_a, double *norm_b)
{
    if (!p) return ANNP_HIP_EARG;
    const annp_host::Potential &q = p->impl.potential();
    if (q.ntl == 0) return ANNP_HIP_EARG;
    if (dims) { dims[0] = q.ntl; dims[1] = q.nhl; dims[2] = q.nnod; dims[3] = q.nsf; dims[4] = q.npsf; dims[5] = q.ntsf; dims[6] = q.flagsym; dims[7] = q.has_symcoef ? 1 : 0; }
    if (scal) { scal[0] = q.cut; scal[1] = q.e_scale; scal[2] = q.e_shift; scal[3] = q.e_atom; scal[4] = q.elements.empty() ? 0.0 : q.elements[0].mass; }
    if (flagact) for (size_t l = 0; l < q.flagact.size(); ++l) flagact[l] = q.flagact[l];
    if (norm_a) for (int k = 0; k < q.nsf && k < (int)q.norm_a.size(); ++k) norm_a[k] = q.norm_a[k];
    if (norm_b) for (int k = 0; k < q.nsf && k < (int)q.norm_b.size(); ++k) norm_b[k] = q.norm_b[k];
    return 0;
}
int annp_pair_potential_layer(const annp_pair *p, int layer, double *w, double *b)
{
    if (!p) return ANNP_HIP_EARG;
    const annp_host::Potential &q = p->impl.potential();
    if (layer < 0 || layer >= (int)q.weights.size()) return ANNP_HIP_EARG;     // element e, layer l: e * (ntl-1) + l
    if (w) std::memcpy(w, q.weights[layer].data(), sizeof(double) * q.weights[layer].size());
    if (b) std::memcpy(b, q.biases[layer].data(), sizeof(double) * q.biases[layer].size());
    return 0;
}
int annp_pair_potential_sym(const annp_pair *p, double *rad, double *ang)
{
    if (!p) return ANNP_HIP_EARG;
    const annp_host::Potential &q = p->impl.potential();
    if (!q.has_symcoef) return ANNP_HIP_EARG;
    if (rad) std::memcpy(rad, q.sym_rad.data(), sizeof(double) * q.sym_rad.size());
    if (ang) std::memcpy(ang, q.sym_ang.data(), sizeof(double) * q.sym_ang.size());
    return 0;
}

}  // extern "C"
