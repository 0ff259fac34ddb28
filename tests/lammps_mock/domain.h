// mock (see README.md): LAMMPS domain.h
#ifndef LMP_DOMAIN_H
#define LMP_DOMAIN_H
namespace LAMMPS_NS { class Domain { public: int triclinic = 0; double sublo[3] = {0, 0, 0}, subhi[3] = {0, 0, 0}, boxlo[3] = {0, 0, 0}, boxhi[3] = {0, 0, 0}; }; }
#endif
