// hb_compile — compile an MJCF file to the .hbm text model (host only, no GPU needed).
// usage: hb_compile in.xml out.hbm [--solver PGS|Newton] [--iterations N] [--timestep h] [--pair-order body|geom]
// (the options override mjOption after compilation: the committed benchmark model carries the benchmark's PGS / 50)
#include "../humanoid_mujoco_amd/csrc/hb_model.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s in.xml|in.hbm out.hbm\n", argv[0]); return 2; }
  hb::Model m;
  std::string err, in = argv[1];
  bool ok = in.size() > 4 && in.substr(in.size() - 4) == ".hbm" ? hb::load_hbm(in, m, err) : hb::compile_mjcf_file(in, m, err);
  if (!ok) { fprintf(stderr, "error: %s\n", err.c_str()); return 1; }
  for (int i = 3; i + 1 < argc; i += 2) {
    if (!strcmp(argv[i], "--solver")) {
      if (!strcmp(argv[i + 1], "PGS")) m.solver = hb::SOL_PGS;
      else if (!strcmp(argv[i + 1], "Newton")) m.solver = hb::SOL_NEWTON;
      else { fprintf(stderr, "error: unknown solver %s\n", argv[i + 1]); return 2; }
    } else if (!strcmp(argv[i], "--iterations")) m.iterations = atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--timestep")) m.timestep = atof(argv[i + 1]);
    else if (!strcmp(argv[i], "--pair-order")) hb::sort_pairs(m, !strcmp(argv[i + 1], "geom") ? 0 : 1);
    else { fprintf(stderr, "error: unknown option %s\n", argv[i]); return 2; }
  }
  if (!hb::save_hbm(m, argv[2], err)) { fprintf(stderr, "error: %s\n", err.c_str()); return 1; }
  double mass = 0;
  for (int b = 0; b < m.nbody; b++) mass += m.body_mass[b];
  printf("nq=%d nv=%d nu=%d nbody=%d njnt=%d ngeom=%d ntendon=%d nM=%d npair=%d nkey=%d mass=%.6f meaninertia=%.6f\n",
         m.nq, m.nv, m.nu, m.nbody, m.njnt, m.ngeom, m.ntendon, m.nM, m.npair, m.nkey, mass, m.meaninertia);
  return 0;
}
