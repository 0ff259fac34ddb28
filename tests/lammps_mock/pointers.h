// mock (see README.md): LAMMPS pointers.h + lammps.h
#ifndef LMP_POINTERS_H
#define LMP_POINTERS_H
#include <cstdio>
#include "lmptype.h"
#define FLERR __FILE__, __LINE__
namespace LAMMPS_NS {
class Memory; class Error; class Universe; class Atom; class Update; class Neighbor; class Comm; class Domain; class Force;
class LAMMPS {
 public:
  Memory *memory = nullptr; Error *error = nullptr; Universe *universe = nullptr; Atom *atom = nullptr; Update *update = nullptr;
  Neighbor *neighbor = nullptr; Comm *comm = nullptr; Domain *domain = nullptr; Force *force = nullptr;
  FILE *screen = nullptr, *logfile = nullptr;
};
class Pointers {
 public:
  Pointers(LAMMPS *ptr) : lmp(ptr), memory(ptr->memory), error(ptr->error), universe(ptr->universe), atom(ptr->atom), update(ptr->update),
                          neighbor(ptr->neighbor), comm(ptr->comm), domain(ptr->domain), force(ptr->force), screen(ptr->screen) {}
  virtual ~Pointers() = default;
 protected:
  LAMMPS *lmp;
  Memory *&memory; Error *&error; Universe *&universe; Atom *&atom; Update *&update; Neighbor *&neighbor; Comm *&comm; Domain *&domain;
  Force *&force; FILE *&screen;
};
}
#endif
