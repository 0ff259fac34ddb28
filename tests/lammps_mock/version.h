// mock (see README.md): LAMMPS version.h
#define LAMMPS_VERSION "2 Aug 2023"
