// lammps_mock_driver.cpp -- plays LAMMPS for the adaptor (see README.md): loads the pair style through the plugin
// entry point, runs pair_style / pair_coeff / init / compute on atoms + ghosts + a paged full neighbour list read
// from a file, writes what the pair style left behind.  TEST DRIVER (g++), compared with the oracle by
// tests/test_lammps_adaptor_mock.py.
//
//   lammps_mock_driver <style> <potential> <in.bin> <out.bin> <eflag> <vflag> El1 [El2 ...]
// in.bin as tests/cpp/annp_gpu_driver.cpp.  out.bin: f64 eng_vdwl ; f64 f[nall*3] ; f64 eatom[nall] ; f64 virial[6] ;
// f64 vatom[nall*6] ; f64 memory_usage ; i32 n_requests request_flags
#include "atom.h"
#include "comm.h"
#include "domain.h"
#include "error.h"
#include "force.h"
#include "lammpsplugin.h"
#include "memory.h"
#include "neigh_list.h"
#include "neighbor.h"
#include "pair.h"
#include "universe.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace LAMMPS_NS;

static std::map<std::string, lammpsplugin_factory1 *> g_pair_styles;
static void register_plugin(lammpsplugin_t *plugin, void *)
{
  if (std::strcmp(plugin->style, "pair") == 0) g_pair_styles[plugin->name] = plugin->creator.v1;
}

template <typename T> static void rd(FILE *fp, T *p, size_t n) { if (n && std::fread(p, sizeof(T), n, fp) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); } }
template <typename T> static void wr(FILE *fp, const T *p, size_t n) { if (n && std::fwrite(p, sizeof(T), n, fp) != n) { std::fprintf(stderr, "short write\n"); std::exit(2); } }

int main(int argc, char **argv)
{
  if (argc < 8) { std::fprintf(stderr, "usage: lammps_mock_driver style pot in.bin out.bin eflag vflag El1 [El2 ...]\n"); return 1; }
  const std::string style = argv[1];
  const int eflag = std::atoi(argv[5]), vflag = std::atoi(argv[6]);
  const int ntypes = argc - 7;

  LAMMPS lmp;
  Memory memory; Error error; Universe universe; Atom atom; Neighbor neighbor; Comm comm; Domain domain; Force force;
  lmp.memory = &memory; lmp.error = &error; lmp.universe = &universe; lmp.atom = &atom; lmp.neighbor = &neighbor; lmp.comm = &comm;
  lmp.domain = &domain; lmp.force = &force; lmp.screen = stderr;

  FILE *fi = std::fopen(argv[3], "rb");
  if (!fi) return 2;
  int hdr[3];
  rd(fi, hdr, 3);
  const int nlocal = hdr[0], nall = hdr[1];
  atom.ntypes = ntypes; atom.nlocal = nlocal; atom.nghost = nall - nlocal; atom.nmax = nall;
  memory.create(atom.x, nall, 3, "atom:x");
  memory.create(atom.f, nall, 3, "atom:f");
  memory.create(atom.type, nall, "atom:type");
  rd(fi, atom.x[0], (size_t)nall * 3);
  rd(fi, atom.type, (size_t)nall);
  std::vector<int> numneigh(nall, 0);
  rd(fi, numneigh.data(), (size_t)nlocal);
  // neighbour pages as LAMMPS' MyPage hands them out: rows of one atom contiguous, pages separate allocations
  const size_t pgsize = 100000;
  std::vector<std::vector<int>> pages(1);
  pages.back().reserve(pgsize);
  std::vector<int> ilist(nlocal);
  std::vector<int *> firstneigh(nall, nullptr);
  std::vector<int> row;
  for (int i = 0; i < nlocal; i++) {
    row.resize((size_t)numneigh[i]);
    rd(fi, row.data(), row.size());
    if (pages.back().size() + row.size() > pgsize) { pages.emplace_back(); pages.back().reserve(pgsize); }
    std::vector<int> &pg = pages.back();
    const size_t at = pg.size();
    pg.insert(pg.end(), row.begin(), row.end());
    firstneigh[i] = pg.data() + at;
    ilist[i] = i;
  }
  std::fclose(fi);
  NeighList list;
  list.inum = nlocal; list.ilist = ilist.data(); list.numneigh = numneigh.data(); list.firstneigh = firstneigh.data();

  int rc = 0;
  try {
    // `plugin load libannp_hip_plugin.so`
    lammpsplugin_init(&lmp, nullptr, (void *) &register_plugin);
    if (!g_pair_styles.count(style)) { std::fprintf(stderr, "pair style %s was not registered\n", style.c_str()); return 3; }
    Pair *pair = (Pair *) (*g_pair_styles[style])(&lmp);
    force.pair = pair;
    pair->settings(0, nullptr);                                          // pair_style <style>
    std::vector<char *> arg;
    char star[] = "*";
    arg.push_back(star); arg.push_back(star); arg.push_back(argv[2]);
    for (int t = 0; t < ntypes; t++) arg.push_back(argv[7 + t]);
    pair->coeff((int)arg.size(), arg.data());                             // pair_coeff * * file El...
    pair->init();                                                         // Pair::init -> init_style, init_one
    pair->list = &list;                                                   // Neighbor::init hands the requested list over
    const int nreq = (int)neighbor.requests.size();
    const int reqflags = nreq ? neighbor.requests[0].flags : -1;

    for (int ago = 0; ago < 2; ago++) {                                   // a rebuild step, then a step that re-uses the list
      neighbor.ago = ago;
      for (int i = 0; i < nall; i++) atom.f[i][0] = atom.f[i][1] = atom.f[i][2] = 0.0;     // Verlet::force_clear
      pair->compute(eflag, vflag);
    }
    FILE *fo = std::fopen(argv[4], "wb");
    if (!fo) return 2;
    wr(fo, &pair->eng_vdwl, 1);
    wr(fo, atom.f[0], (size_t)nall * 3);
    std::vector<double> zeros((size_t)nall * 6, 0.0);
    wr(fo, pair->eatom ? pair->eatom : zeros.data(), (size_t)nall);
    wr(fo, pair->virial, 6);
    wr(fo, pair->vatom ? pair->vatom[0] : zeros.data(), (size_t)nall * 6);
    const double mem = pair->memory_usage();
    wr(fo, &mem, 1);
    const int tail[2] = {nreq, reqflags};
    wr(fo, tail, 2);
    std::fclose(fo);
    delete pair;
  } catch (const LAMMPSException &e) {
    std::fprintf(stderr, "%s\n", e.what());
    rc = 9;
  }
  memory.destroy(atom.x); memory.destroy(atom.f); memory.destroy(atom.type);
  return rc;
}
