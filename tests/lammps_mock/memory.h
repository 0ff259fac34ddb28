// mock (see README.md): LAMMPS memory.h -- create/destroy of contiguous 1-d and 2-d arrays (one block + row pointers)
#ifndef LMP_MEMORY_H
#define LMP_MEMORY_H
#include <cstdlib>
namespace LAMMPS_NS {
class Memory {
 public:
  template <typename T> T *create(T *&array, int n, const char *) { array = (T *) std::calloc((size_t)(n > 0 ? n : 1), sizeof(T)); return array; }
  template <typename T> void destroy(T *&array) { std::free(array); array = nullptr; }
  template <typename T> T **create(T **&array, int n1, int n2, const char *)
  {
    T *data = (T *) std::calloc((size_t)n1 * n2 + 1, sizeof(T));
    array = (T **) std::malloc(sizeof(T *) * (size_t)(n1 > 0 ? n1 : 1));
    for (int i = 0; i < n1; i++) array[i] = data + (size_t)i * n2;
    if (n1 == 0) array[0] = data;
    return array;
  }
  template <typename T> void destroy(T **&array) { if (!array) return; std::free(array[0]); std::free(array); array = nullptr; }
};
}
#endif
