// annp_potential.cpp -- parser for the `.ann` potential file (see annp_potential.h).
//
// File layout (0-based line numbers, CRLF line ends, TAB separated; `ne` = number
// of elements), as consumed by the reference parser fe_v2/src/pair_annp.cpp:347-517:
//   5            nelements
//   6..5+ne      id <TAB> symbol <TAB> mass
//   8+ne         TL HL nodes nsf npsf ntsf cut
//   11+ne        normalisation row a (Fe sfnor_cov | Ni sf_min)
//   12+ne        normalisation row b (Fe sfnor_avg | Ni sf_max)
//   15+ne        descriptor and activation names, recognised by two-letter probes
//   18..20+ne    e_scale, e_shift, e_atom
//   then blocks "#<El>", "#k_(weight)" + rows, "#k_(bias)" + one row,
//   optionally "#coef..." + "#rad n" rows + "#angl n" rows (Ni).
// A number is recognised at the start of a line and after every TAB that is
// followed by a digit or '-' (fe:400, fe:496); the same rule is kept here so that
// any file the reference accepts yields the same values.
#include "annp_potential.h"

#include <cctype>
#include <cstdlib>
#include <fstream>

namespace annp_host {
namespace {

// numbers after TABs (and optionally the one at column 0)
std::vector<double> row_values(const std::string &s, bool leading)
{
    std::vector<double> v;
    if (leading) v.push_back(std::atof(s.c_str()));
    for (size_t j = 0; j + 1 < s.size(); ++j) {
        const unsigned char nx = (unsigned char)s[j + 1];
        if (s[j] == '\t' && (std::isdigit(nx) || nx == '-')) v.push_back(std::atof(s.c_str() + j + 1));
    }
    return v;
}

std::vector<double> unsigned_after_tabs(const std::string &s)
{
    std::vector<double> v;
    for (size_t j = 0; j + 1 < s.size(); ++j)
        if (s[j] == '\t' && std::isdigit((unsigned char)s[j + 1])) v.push_back(std::atof(s.c_str() + j + 1));
    return v;
}

int activation_code(char a, char b)
{
    if (a == 'l' && b == 'i') return 0;   // linear
    if (a == 'h' && b == 'y') return 1;   // hyperbolic tangent
    if (a == 's' && b == 'i') return 2;   // sigmoid
    if (a == 'm' && b == 'o') return 3;   // modified tanh
    if (a == 't' && b == 'a') return 4;   // "tanh" in the shipped files -> twisted tanh
    return -1;
}

}  // namespace

bool read_potential(const std::string &path, int nelements_coeff, Potential &pot, std::string &err,
                    const std::vector<std::string> *blocks_by_name)
{
    std::ifstream fin(path.c_str());
    if (!fin.is_open()) { err = "Cannot open neural network potential file"; return false; }
    pot = Potential();
    std::string line;
    int ne = 0;
    const int nheader = 21 + nelements_coeff;
    for (int i = 0; i < nheader; ++i) {
        if (!std::getline(fin, line)) { err = "potential file ends inside the header"; return false; }
        if (i == 5) {
            ne = std::atoi(line.c_str());
            if (ne < 1) { err = "potential file: bad element count"; return false; }
            pot.elements.resize(ne);
        } else if (i >= 6 && i < 6 + ne) {
            Element &e = pot.elements[i - 6];
            e.id = std::atoi(line.c_str());
            for (char c : line) if (std::isalpha((unsigned char)c)) e.name += c;
            const std::vector<double> m = unsigned_after_tabs(line);
            if (!m.empty()) e.mass = m.back();
        } else if (i == 8 + ne) {
            pot.ntl = std::atoi(line.c_str());
            const std::vector<double> v = unsigned_after_tabs(line);
            if (v.size() < 6) { err = "potential file: network parameter line is short"; return false; }
            pot.nhl = (int)v[0]; pot.nnod = (int)v[1]; pot.nsf = (int)v[2];
            pot.npsf = (int)v[3]; pot.ntsf = (int)v[4]; pot.cut = v[5];
            if (pot.ntl < 3 || pot.nnod < 1 || pot.nsf < 1 || pot.npsf + pot.ntsf != pot.nsf) {
                err = "potential file: inconsistent network dimensions";
                return false;
            }
        } else if (i == 11 + ne || i == 12 + ne) {
            std::vector<double> v = row_values(line, true);
            if ((int)v.size() < pot.nsf) { err = "potential file: normalisation row is short"; return false; }
            v.resize(pot.nsf);
            (i == 11 + ne ? pot.norm_a : pot.norm_b) = v;
        } else if (i == 15 + ne) {
            for (size_t j = 0; j + 1 < line.size(); ++j) {
                const char a = line[j], b = line[j + 1];
                if (a == 'C' && b == 'h') pot.flagsym = 0;
                if (a == 'B' && (b == 'e' || b == 'P')) pot.flagsym = 1;
                if (a == 'C' && b == 'u') pot.flagsym = 2;
                const int act = activation_code(a, b);
                if (act >= 0) pot.flagact.push_back(act);
            }
        } else if (i == 18 + ne) pot.e_scale = std::atof(line.c_str());
        else if (i == 19 + ne) pot.e_shift = std::atof(line.c_str());
        else if (i == 20 + ne) pot.e_atom = std::atof(line.c_str());
    }
    const int nl = pot.ntl - 1;
    if ((int)pot.flagact.size() < nl) { err = "potential file: fewer activation names than layers"; return false; }
    pot.flagact.resize(nl);
    // one zero-filled network per element (fe:441-448: c_3d_matrix memsets), element e's layer l at [e * nl + l]
    pot.weights.assign((size_t)ne * nl, std::vector<double>());
    pot.biases.assign((size_t)ne * nl, std::vector<double>());
    for (int e = 0; e < ne; ++e)
        for (int l = 0; l < nl; ++l) {
            pot.weights[(size_t)e * nl + l].assign((size_t)pot.rows(l) * pot.cols(l), 0.0);
            pot.biases[(size_t)e * nl + l].assign(pot.rows(l), 0.0);
        }

    // Which element a block belongs to.  The reference declares `type_elem = 0` INSIDE its line loop (fe:455), so the
    // element named on a "#El" line is forgotten before the "#k_(weight)" line that follows it is read: every block
    // of every element lands in element 0 (the last one in the file wins) and elements 1.. keep their zero-filled
    // networks.  That is what `blocks_by_name == nullptr` reproduces (the default: parity with the reference).
    // With the pair_coeff element names given, a "#El" line selects the element of the blocks below it, which is
    // evidently what the file format means.
    int cur = 0;
    while (std::getline(fin, line)) {
        if (line.compare(0, 5, "#coef") == 0) { pot.has_symcoef = true; break; }
        if (blocks_by_name && line.size() >= 2 && line[0] == '#' && std::isupper((unsigned char)line[1])) {
            std::string name;
            for (char c : line) if (c != '#' && c != '\r' && c != '\n' && c != '\t' && c != ' ') name += c;
            for (size_t k = 0; k < blocks_by_name->size() && (int)k < ne; ++k)
                if ((*blocks_by_name)[k] == name) cur = (int)k;
            continue;
        }
        if (line.size() < 2 || line[0] != '#' || !std::isdigit((unsigned char)line[1])) continue;
        int layer = 0;
        bool is_bias = false;
        for (char c : line) {
            if (c >= '0' && c <= '9') layer = layer * 10 + (c - '0');
            if (c == 'w') is_bias = false;
            if (c == 'b') is_bias = true;
        }
        const int l = layer - 1;
        if (l < 0 || l >= nl) { err = "potential file: layer number out of range"; return false; }
        std::vector<double> &W = pot.weights[(size_t)cur * nl + l], &B = pot.biases[(size_t)cur * nl + l];
        if (!is_bias) {
            for (int r = 0; r < pot.rows(l); ++r) {
                if (!std::getline(fin, line)) { err = "potential file ends inside a weight block"; return false; }
                const std::vector<double> v = row_values(line, true);
                for (int c = 0; c < pot.cols(l) && c < (int)v.size(); ++c) W[(size_t)r * pot.cols(l) + c] = v[c];
            }
        } else {
            if (!std::getline(fin, line)) { err = "potential file ends inside a bias block"; return false; }
            const std::vector<double> v = row_values(line, true);
            for (int c = 0; c < pot.rows(l) && c < (int)v.size(); ++c) B[c] = v[c];
        }
    }
    if (pot.has_symcoef) {
        pot.sym_rad.assign((size_t)pot.npsf * 3, 0.0);
        pot.sym_ang.assign((size_t)pot.ntsf * 4, 0.0);
        std::getline(fin, line);                       // "#rad n"
        for (int i = 0; i < pot.npsf; ++i) {
            if (!std::getline(fin, line)) { err = "potential file ends inside #rad"; return false; }
            const std::vector<double> v = row_values(line, false);
            for (int c = 0; c < 3 && c < (int)v.size(); ++c) pot.sym_rad[(size_t)i * 3 + c] = v[c];
        }
        std::getline(fin, line);                       // "#angl n"
        for (int i = 0; i < pot.ntsf; ++i) {
            if (!std::getline(fin, line)) { err = "potential file ends inside #angl"; return false; }
            const std::vector<double> v = row_values(line, false);
            for (int c = 0; c < 4 && c < (int)v.size(); ++c) pot.sym_ang[(size_t)i * 4 + c] = v[c];
        }
    }
    return true;
}

// `.anna` layout (0-based lines; adp = bcc_fe/src/pair_anna_adp.cpp):
//   5 nelements | 6.. id symbol mass | 8+ne TL HL nodes nout nsf npsf ntsf cut (adp:426-443)
//   11+ne names (adp:444-461) | 14+ne e_base <TAB> e_scale (adp:462-470) | 17+ne ngp | 18+ne the ngp parameters
//   then the weight / bias blocks; the last layer has nout rows (adp:497-553).  No normalisation rows.
bool read_potential_anna(const std::string &path, int nelements_coeff, Potential &pot, std::string &err)
{
    std::ifstream fin(path.c_str());
    if (!fin.is_open()) { err = "Cannot open physically informed neural network potential file"; return false; }   // adp:401
    pot = Potential();
    pot.is_anna = true;
    std::string line;
    int ne = 0, ngp = 0;
    for (int i = 0; i < 19 + nelements_coeff; ++i) {
        if (!std::getline(fin, line)) { err = "potential file ends inside the header"; return false; }
        if (i == 5) {
            ne = std::atoi(line.c_str());
            if (ne < 1) { err = "potential file: bad element count"; return false; }
            pot.elements.resize(ne);
        } else if (i >= 6 && i < 6 + ne) {
            Element &e = pot.elements[i - 6];
            e.id = std::atoi(line.c_str());
            for (char c : line) if (std::isalpha((unsigned char)c)) e.name += c;
            const std::vector<double> m = unsigned_after_tabs(line);
            if (!m.empty()) e.mass = m.back();
        } else if (i == 8 + ne) {
            pot.ntl = std::atoi(line.c_str());
            const std::vector<double> v = unsigned_after_tabs(line);
            if (v.size() < 7) { err = "potential file: network parameter line is short"; return false; }
            pot.nhl = (int)v[0]; pot.nnod = (int)v[1]; pot.nout = (int)v[2]; pot.nsf = (int)v[3];
            pot.npsf = (int)v[4]; pot.ntsf = (int)v[5]; pot.cut = v[6];
            if (pot.ntl < 2 || pot.nnod < 1 || pot.nout < 1 || pot.nsf < 1 || pot.npsf + pot.ntsf != pot.nsf) {
                err = "potential file: inconsistent network dimensions";
                return false;
            }
        } else if (i == 11 + ne) {
            for (size_t j = 0; j + 1 < line.size(); ++j) {
                const char a = line[j], b = line[j + 1];
                if (a == 'C' && b == 'h') pot.flagsym = 0;
                if (a == 'B' && (b == 'e' || b == 'P')) pot.flagsym = 1;
                if (a == 'C' && b == 'u') pot.flagsym = 2;
                const int act = activation_code(a, b);
                if (act >= 0) pot.flagact.push_back(act);
            }
        } else if (i == 14 + ne) {
            pot.e_base = std::atof(line.c_str());
            const std::vector<double> v = unsigned_after_tabs(line);
            if (!v.empty()) pot.e_scal = v.back();
        } else if (i == 17 + ne) {
            ngp = std::atoi(line.c_str());
            if (ngp < 17 || ngp > 64) { err = "potential file: the ADP form needs 17 parameters"; return false; }
        } else if (i == 18 + ne) {
            pot.gparams = row_values(line, true);
            if ((int)pot.gparams.size() < ngp) { err = "potential file: ADP parameter row is short"; return false; }
            pot.gparams.resize(ngp);
        }
    }
    const int nl = pot.ntl - 1;
    if ((int)pot.flagact.size() < nl) { err = "potential file: fewer activation names than layers"; return false; }
    pot.flagact.resize(nl);
    pot.weights.assign(nl, std::vector<double>());
    pot.biases.assign(nl, std::vector<double>());
    for (int l = 0; l < nl; ++l) {
        pot.weights[l].assign((size_t)pot.rows(l) * pot.cols(l), 0.0);
        pot.biases[l].assign(pot.rows(l), 0.0);
    }
    while (std::getline(fin, line)) {
        if (line.size() < 2 || line[0] != '#' || !std::isdigit((unsigned char)line[1])) continue;
        int layer = 0;
        bool is_bias = false;
        for (char c : line) {
            if (c >= '0' && c <= '9') layer = layer * 10 + (c - '0');
            if (c == 'w') is_bias = false;
            if (c == 'b') is_bias = true;
        }
        const int l = layer - 1;
        if (l < 0 || l >= nl) { err = "potential file: layer number out of range"; return false; }
        if (!is_bias) {
            for (int r = 0; r < pot.rows(l); ++r) {
                if (!std::getline(fin, line)) { err = "potential file ends inside a weight block"; return false; }
                const std::vector<double> v = row_values(line, true);
                for (int c = 0; c < pot.cols(l) && c < (int)v.size(); ++c) pot.weights[l][(size_t)r * pot.cols(l) + c] = v[c];
            }
        } else {
            if (!std::getline(fin, line)) { err = "potential file ends inside a bias block"; return false; }
            const std::vector<double> v = row_values(line, true);
            for (int c = 0; c < pot.rows(l) && c < (int)v.size(); ++c) pot.biases[l][c] = v[c];
        }
    }
    return true;
}

}  // namespace annp_host
