// mock (see README.md): LAMMPS neigh_list.h
#ifndef LMP_NEIGH_LIST_H
#define LMP_NEIGH_LIST_H
namespace LAMMPS_NS { class NeighList { public: int inum = 0, gnum = 0; int *ilist = nullptr, *numneigh = nullptr; int **firstneigh = nullptr; }; }
#endif
