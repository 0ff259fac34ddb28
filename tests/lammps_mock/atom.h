// mock (see README.md): LAMMPS atom.h
#ifndef LMP_ATOM_H
#define LMP_ATOM_H
#include "pointers.h"
namespace LAMMPS_NS {
class Atom {
 public:
  int ntypes = 1, nlocal = 0, nghost = 0, nmax = 0, tag_enable = 1;
  double **x = nullptr, **f = nullptr;
  int *type = nullptr;
  tagint *tag = nullptr;
};
}
#endif
