// mock (see README.md): LAMMPS force.h
#ifndef LMP_FORCE_H
#define LMP_FORCE_H
namespace LAMMPS_NS { class Pair; class Force { public: int newton = 1, newton_pair = 1, newton_bond = 1; Pair *pair = nullptr; double nktv2p = 1.6021765e6; }; }
#endif
