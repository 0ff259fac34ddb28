// mock (see README.md): LAMMPS pair.h -- the members pair_annp_hip.cpp touches, LAMMPS' names and semantics
#ifndef LMP_PAIR_H
#define LMP_PAIR_H
#include "pointers.h"
namespace LAMMPS_NS {
class NeighList;
class Pair : protected Pointers {
 public:
  double eng_vdwl = 0.0, eng_coul = 0.0;
  double virial[6] = {0, 0, 0, 0, 0, 0};
  double *eatom = nullptr, **vatom = nullptr;
  double cutforce = 0.0;
  double **cutsq = nullptr;
  int **setflag = nullptr;
  int restartinfo = 1, one_coeff = 0, manybody_flag = 0, no_virial_fdotr = 0;
  int allocated = 0, copymode = 0;
  NeighList *list = nullptr;
  int evflag = 0, eflag_either = 0, eflag_global = 0, eflag_atom = 0, vflag_either = 0, vflag_global = 0, vflag_atom = 0, vflag_fdotr = 0;
  int maxeatom = 0, maxvatom = 0;

  explicit Pair(LAMMPS *lmp) : Pointers(lmp) {}
  ~Pair() override;
  virtual void compute(int, int) = 0;
  virtual void settings(int, char **) = 0;
  virtual void coeff(int, char **) = 0;
  virtual void init_style();
  virtual double init_one(int, int) { return 0.0; }
  virtual double memory_usage();
  void init();                          // Pair::init: init_style + init_one for every type pair -> cutsq, cutforce
  void ev_init(int eflag, int vflag, int alloc = 1) { if (eflag || vflag) ev_setup(eflag, vflag, alloc); else ev_unset(); }
  void ev_setup(int, int, int alloc = 1);
  void ev_unset();
  void virial_fdotr_compute();
};
enum { ENERGY_NONE = 0x00, ENERGY_GLOBAL = 0x01, ENERGY_ATOM = 0x02 };
enum { VIRIAL_NONE = 0x00, VIRIAL_PAIR = 0x01, VIRIAL_FDOTR = 0x02, VIRIAL_ATOM = 0x04, VIRIAL_CENTROID = 0x08 };
}
#endif
