/* annp_hip_plugin.cpp -- registers pair_style annp/hip with a LAMMPS built with
 * PKG_PLUGIN (`plugin load libannp_hip_plugin.so`), so an unmodified LAMMPS binary
 * picks the style up at run time.  Follows LAMMPS' examples/plugins layout. */
#include "lammpsplugin.h"
#include "version.h"

#include "pair_annp_hip.h"

using namespace LAMMPS_NS;

static Pair *annp_hip_creator(LAMMPS *lmp)
{
  return new PairANNPHIP(lmp);
}

static Pair *anna_adp_hip_creator(LAMMPS *lmp)
{
  return new PairANNAADPHIP(lmp);
}

extern "C" void lammpsplugin_init(void *lmp, void *handle, void *regfunc)
{
  lammpsplugin_t plugin;
  lammpsplugin_regfunc register_plugin = (lammpsplugin_regfunc) regfunc;

  plugin.version = LAMMPS_VERSION;
  plugin.style = "pair";
  plugin.name = "annp/hip";
  plugin.info = "ANN potential (pair_style annp) evaluated on AMD MI355X through libannp_hip";
  plugin.author = "annp-hip";
  plugin.creator.v1 = (lammpsplugin_factory1 *) &annp_hip_creator;
  plugin.handle = handle;
  (*register_plugin)(&plugin, lmp);

  plugin.name = "anna_adp/hip";
  plugin.info = "ANN-parametrised ADP potential (pair_style anna_adp) evaluated on AMD MI355X through libannp_hip";
  plugin.creator.v1 = (lammpsplugin_factory1 *) &anna_adp_hip_creator;
  (*register_plugin)(&plugin, lmp);
}
