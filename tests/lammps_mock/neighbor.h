// mock (see README.md): LAMMPS neighbor.h + neigh_request.h
#ifndef LMP_NEIGHBOR_H
#define LMP_NEIGHBOR_H
#include <vector>
namespace LAMMPS_NS {
namespace NeighConst { enum { REQ_DEFAULT = 0, REQ_FULL = 1 << 0, REQ_GHOST = 1 << 1, REQ_SIZE = 1 << 2, REQ_HISTORY = 1 << 3, REQ_OCCASIONAL = 1 << 4 }; }
class NeighRequest { public: void *requestor = nullptr; int flags = 0; };
class Neighbor {
 public:
  double skin = 2.0, cutneighmax = 0.0;
  int ago = 0, oneatom = 2000, pgsize = 100000;
  std::vector<NeighRequest> requests;
  NeighRequest *add_request(class Pair *requestor, int flags = 0) { requests.push_back(NeighRequest{(void *) requestor, flags}); return &requests.back(); }
};
}
#endif
