// mock (see README.md): LAMMPS lmptype.h, default LAMMPS_SMALLBIG sizes
#ifndef LAMMPS_LMPTYPE_H
#define LAMMPS_LMPTYPE_H
#include <cstdint>
namespace LAMMPS_NS {
typedef int tagint;
typedef int64_t bigint;
#define NEIGHMASK 0x1FFFFFFF
}
#endif
