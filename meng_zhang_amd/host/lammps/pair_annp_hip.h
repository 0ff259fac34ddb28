/* pair_annp_hip.h -- LAMMPS adaptor: pair_style annp/hip (and, when built with
 * -DANNP_HIP_OVERRIDE_ANNP, pair_style annp itself).
 *
 * Drop this file and pair_annp_hip.cpp into an unmodified LAMMPS src/ tree (or build
 * them as a plugin) and link libannp_hip.so.  The class exposes
 * exactly the surface of the reference pair style
 * (annp-gpu-lammps/fe_v2/src/pair_annp.h:24-32, pair_annp_gpu.h:24-37):
 * compute / settings / coeff / init_one / init_style / memory_usage, same input
 * script syntax (`pair_style annp/hip`, `pair_coeff * * file.ann Fe`), same
 * requirements (newton_pair on, full neighbour list, one_coeff, manybody).
 * It owns no arithmetic: everything is forwarded to annp_host::PairANNP
 * (annp_pair.h), i.e. to the C ABI in include/annp_hip.h.
 *
 * pair_style anna_adp/hip is the same adaptor over the reference's second pair style
 * (anna-gpu-lammps/bcc_fe/src/pair_anna_adp.h:24-31: identical surface, `.anna` potential
 * file, `pair_coeff * * fe_adp_potential_2310.anna Fe`); newton_pair must be on, as for the
 * reference's CPU style (bcc_fe/README.md "MD simulation in Lammps").
 */
#ifdef PAIR_CLASS
// clang-format off
PairStyle(annp/hip, PairANNPHIP);
PairStyle(anna_adp/hip, PairANNAADPHIP);
#ifdef ANNP_HIP_OVERRIDE_ANNP
PairStyle(annp, PairANNPHIP);
PairStyle(anna_adp, PairANNAADPHIP);
#endif
// clang-format on
#else

#ifndef LMP_PAIR_ANNP_HIP_H
#define LMP_PAIR_ANNP_HIP_H

#include "pair.h"

namespace annp_host { class PairANNP; }

namespace LAMMPS_NS {

class PairANNPHIP : public Pair {
 public:
  PairANNPHIP(class LAMMPS *);
  ~PairANNPHIP() override;
  void compute(int, int) override;
  void settings(int, char **) override;
  void coeff(int, char **) override;
  double init_one(int, int) override;
  void init_style() override;
  double memory_usage() override;

 protected:
  annp_host::PairANNP *impl;
  const char *host_style;   // "annp" | "anna_adp": which reference pair style the host mirror follows
  double cutmax;
  int device_id;      // GPU of this rank: ANNP_HIP_DEVICE or (local rank mod visible devices)
  int device_neigh;   // ANNP_HIP_NEIGH=device: list built on the GPU (annp_gpu_compute_n analogue, `package gpu neigh yes`)
  void allocate();
};

class PairANNAADPHIP : public PairANNPHIP {
 public:
  PairANNAADPHIP(class LAMMPS *lmp) : PairANNPHIP(lmp) { host_style = "anna_adp"; }
};

}    // namespace LAMMPS_NS

#endif
#endif
