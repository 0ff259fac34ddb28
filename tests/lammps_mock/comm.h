// mock (see README.md): LAMMPS comm.h
#ifndef LMP_COMM_H
#define LMP_COMM_H
namespace LAMMPS_NS { class Comm { public: int me = 0, nprocs = 1; }; }
#endif
