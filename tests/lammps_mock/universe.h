// mock (see README.md): LAMMPS universe.h
#ifndef LMP_UNIVERSE_H
#define LMP_UNIVERSE_H
namespace LAMMPS_NS { class Universe { public: int me = 0, nprocs = 1; }; }
#endif
