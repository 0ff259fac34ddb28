// mock (see README.md): the few Pair methods of LAMMPS' pair.cpp that the adaptor relies on, restated from the LAMMPS
// developer documentation (ev_setup flag logic, virial_fdotr_compute, Pair::init's cutsq loop).
#include "pair.h"

#include "atom.h"
#include "force.h"
#include "memory.h"
#include "neighbor.h"

#include <cmath>
#include <cstring>

using namespace LAMMPS_NS;

Pair::~Pair()
{
  memory->destroy(eatom);
  memory->destroy(vatom);
}

void Pair::init_style() { neighbor->add_request(this); }

double Pair::memory_usage() { return (double)maxeatom * sizeof(double) + (double)maxvatom * 6 * sizeof(double); }

void Pair::init()
{
  init_style();
  cutforce = 0.0;
  for (int i = 1; i <= atom->ntypes; i++)
    for (int j = i; j <= atom->ntypes; j++) {
      const double cut = init_one(i, j);
      cutsq[i][j] = cutsq[j][i] = cut * cut;
      cutforce = std::fmax(cutforce, cut);
    }
}

void Pair::ev_setup(int eflag, int vflag, int alloc)
{
  evflag = 1;
  eflag_either = eflag;
  eflag_global = eflag & ENERGY_GLOBAL;
  eflag_atom = eflag & ENERGY_ATOM;
  vflag_global = vflag & (VIRIAL_PAIR | VIRIAL_FDOTR);
  vflag_atom = vflag & VIRIAL_ATOM;
  vflag_either = vflag_global || vflag_atom;
  const int nall = atom->nlocal + atom->nghost;
  if (eflag_atom && nall > maxeatom) {
    maxeatom = atom->nmax > nall ? atom->nmax : nall;
    if (alloc) { memory->destroy(eatom); memory->create(eatom, maxeatom, "pair:eatom"); }
  }
  if (vflag_atom && nall > maxvatom) {
    maxvatom = atom->nmax > nall ? atom->nmax : nall;
    if (alloc) { memory->destroy(vatom); memory->create(vatom, maxvatom, 6, "pair:vatom"); }
  }
  if (eflag_global) eng_vdwl = eng_coul = 0.0;
  if (vflag_global) for (int i = 0; i < 6; i++) virial[i] = 0.0;
  if (eflag_atom && alloc) for (int i = 0; i < nall; i++) eatom[i] = 0.0;
  if (vflag_atom && alloc) for (int i = 0; i < nall; i++) for (int k = 0; k < 6; k++) vatom[i][k] = 0.0;
  // global virial via F dot r when the style calls virial_fdotr_compute() (pair.cpp: "unset other flags as appropriate")
  if (vflag_global == VIRIAL_FDOTR && no_virial_fdotr == 0) {
    vflag_fdotr = 1;
    vflag_global = 0;
    if (vflag_atom == 0) vflag_either = 0;
    if (vflag_either == 0 && eflag_either == 0) evflag = 0;
  } else {
    vflag_fdotr = 0;
  }
}

void Pair::ev_unset()
{
  evflag = 0;
  eflag_either = eflag_global = eflag_atom = 0;
  vflag_either = vflag_global = vflag_atom = vflag_fdotr = 0;
}

void Pair::virial_fdotr_compute()
{
  double **x = atom->x, **f = atom->f;
  const int nall = atom->nlocal + atom->nghost;
  double v[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < nall; i++) {
    v[0] += x[i][0] * f[i][0]; v[1] += x[i][1] * f[i][1]; v[2] += x[i][2] * f[i][2];
    v[3] += x[i][0] * f[i][1]; v[4] += x[i][0] * f[i][2]; v[5] += x[i][1] * f[i][2];
  }
  for (int k = 0; k < 6; k++) virial[k] += v[k];
}
