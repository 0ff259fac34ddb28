// annp_potential.h -- the `.ann` potential file of pair_style annp, host side.
//
// Mirrors what PairANNP::read_file produces (reference:
// annp-gpu-lammps/fe_v2/src/pair_annp.cpp:332-523, Param_ANNP at
// fe_v2/src/pair_annp.h:53-62; the Ni variant with its symmetry-function
// coefficient section: ni/src/pair_annp.cpp:324-545, ni/src/pair_annp.h:54-64).
// No LAMMPS dependency; used by the LAMMPS adaptor and by the C API in annp_pair.h.
#pragma once
#include <string>
#include <vector>

namespace annp_host {

struct Element {
    int id = 1;
    double mass = 0.0;
    std::string name;
};

struct Potential {
    std::vector<Element> elements;
    int ntl = 0, nhl = 0, nnod = 0, nsf = 0, npsf = 0, ntsf = 0;
    double cut = 0.0;
    double e_scale = 0.0, e_shift = 0.0, e_atom = 0.0;
    int flagsym = 0;                       // Ch 0, Be/BP 1, Cu 2 (two-character probes, fe:415-417)
    std::vector<int> flagact;              // per weight layer: li 0, hy 1, si 2, mo 3, ta 4
    std::vector<double> norm_a, norm_b;    // Fe: sfnor_cov, sfnor_avg;  Ni: sf_min, sf_max
    // weights[e * (ntl-1) + l]: element e, layer l, row-major [rows(l)][cols(l)]  (weights[l] is element 0)
    std::vector<std::vector<double>> weights, biases;
    bool has_symcoef = false;              // "#coefficent of symmetry funciton" section present
    std::vector<double> sym_rad;           // [npsf][3] eta, Rs, Rc
    std::vector<double> sym_ang;           // [ntsf][4] eta, lambda, zeta, Rc

    // pair_style anna_adp (anna-gpu-lammps/bcc_fe/src/pair_anna_adp.h:58-67): the network ends in `nout`
    // local parameters of an analytic ADP form instead of an energy
    bool is_anna = false;
    int nout = 1;
    double e_base = 0.0, e_scal = 0.0;
    std::vector<double> gparams;           // A0 yy gamma C0 c1F c2F V0 b1 b2 delta r0 r1 hc d1 q1 d3 q3

    int rows(int l) const { return l == ntl - 2 ? nout : nnod; }
    int cols(int l) const { return l == 0 ? nsf : nnod; }
};

// nelements_coeff: number of distinct element names on the pair_coeff line.
// Returns true on success; on failure `err` says why (text matches the reference's
// error where it has one: "Cannot open neural network potential file").
// blocks_by_name: nullptr = the reference's behaviour for files with several elements (every weight block lands in
// element 0, see the .cpp); the pair_coeff element names = "#El" lines select the element of the blocks below them.
bool read_potential(const std::string &path, int nelements_coeff, Potential &pot, std::string &err,
                    const std::vector<std::string> *blocks_by_name = nullptr);

// The `.anna` file of pair_style anna_adp, as PairANNA_ADP::read_file consumes it
// (anna-gpu-lammps/bcc_fe/src/pair_anna_adp.cpp:392-566).
bool read_potential_anna(const std::string &path, int nelements_coeff, Potential &pot, std::string &err);

}  // namespace annp_host
