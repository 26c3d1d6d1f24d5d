// Host-side code of libgsx (problem lowering, orderings, symbolic analysis incl. the shard partition, the on-disk
// readers/writers) under AddressSanitizer + UBSan: tests/test_host_logic.py builds this with g++ -fsanitize=address,undefined
// from the product sources and runs it on the golden files.  No GPU, no HIP.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../gtsam_petercdev_amd/csrc/gsx_internal.h"

extern "C" {
gsx_status gsx_dataset_get(const gsx_dataset* d, gsx_problem_desc* desc, const double** values, int64_t* n_values);
}

static int check(const char* what, gsx_dataset* ds, const std::string& tmp) {
  gsx_problem_desc desc;
  const double* values = nullptr;
  int64_t nv = 0;
  if (gsx_dataset_get(ds, &desc, &values, &nv) != GSX_OK) return 1;
  gsx::HostProblem P;
  std::string err;
  if (gsx::lower_problem(&desc, P, err) != GSX_OK) {
    std::fprintf(stderr, "%s: lower_problem: %s\n", what, err.c_str());
    return 1;
  }
  for (int kind = 0; kind <= 4; ++kind) {
    std::vector<int> order;
    gsx::compute_ordering(P, kind, order);
    if ((int)order.size() != P.n_vars) return 1;
    for (double relax : {0.0, 0.5})
      for (int world : {1, 2, 3, 8})
        for (int rank = 0; rank < world; rank += (world > 2 ? world - 1 : 1)) {
          gsx::Symbolic S;
          if (gsx::symbolic_analysis(P, order, relax, 64, rank, world, S, err) != GSX_OK) {
            std::fprintf(stderr, "%s: symbolic (kind %d, relax %g, %d/%d): %s\n", what, kind, relax, rank, world, err.c_str());
            return 1;
          }
          // every scheduled front of a level is of that level, every front is scheduled exactly once when not sharded
          if (world == 1 && (int)S.sched.size() != S.n_fronts) return 1;
          for (int l = 0; l < S.n_levels; ++l)
            for (int k = S.lvl_ptr[l]; k < S.lvl_ptr[l + 1]; ++k)
              if (S.level[S.sched[k]] != l) return 1;
        }
  }
  // writers
  if (gsx_write_g2o(&desc, values, nv, (tmp + "/san.g2o").c_str()) != GSX_OK) return 1;
  const double sig[3] = {0.1, 0.1, 0.05};
  if (gsx_save2d(&desc, values, nv, sig, (tmp + "/san.graph").c_str()) != GSX_OK) return 1;
  gsx_write_bal(&desc, values, nv, (tmp + "/san_bal.txt").c_str());  // (GSX_E_INVALID for a problem without cameras)
  std::printf("%s: %d variables, %d factors ok\n", what, P.n_vars, P.n_factors);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string dir = argv[1], tmp = argv[2];
  int bad = 0;
  gsx_dataset* ds = nullptr;
  struct {
    const char* file;
    int kind;  // 0 g2o 2d, 1 g2o 3d, 2 load2d, 3 bal
  } files[] = {{"pose2example.txt", 0}, {"noisyToyGraph.txt", 0}, {"pose3example.txt", 1}, {"sphere2500.txt", 1},
               {"w100.graph", 2},       {"example.graph", 2},     {"dubrovnik-3-7-pre.txt", 3}};
  for (const auto& f : files) {
    const std::string path = dir + "/" + f.file;
    gsx_status st;
    if (f.kind == 0) st = gsx_read_g2o(path.c_str(), 0, &ds);
    else if (f.kind == 1) st = gsx_read_g2o(path.c_str(), 1, &ds);
    else if (f.kind == 2) st = gsx_load2d(path.c_str(), nullptr, 0, 1, GSX_NOISE_FORMAT_AUTO, 0, &ds);
    else st = gsx_read_bal(path.c_str(), 1, &ds);
    if (st != GSX_OK) {
      std::fprintf(stderr, "%s: reader status %d\n", f.file, (int)st);
      bad = 1;
      continue;
    }
    bad |= check(f.file, ds, tmp);
    gsx_dataset_free(ds);
  }
  // malformed inputs must be refused, not crash
  {
    const std::string p = tmp + "/bad.g2o";
    FILE* fh = std::fopen(p.c_str(), "w");
    std::fputs("VERTEX_SE2 0 0 0\nEDGE_SE2 0 1 1 0\nEDGE_SE3:QUAT 0 1 1\n", fh);
    std::fclose(fh);
    if (gsx_read_g2o(p.c_str(), 0, &ds) == GSX_OK) gsx_dataset_free(ds);
    if (gsx_read_g2o(p.c_str(), 1, &ds) == GSX_OK) gsx_dataset_free(ds);
    if (gsx_load2d(p.c_str(), nullptr, 0, 1, GSX_NOISE_FORMAT_AUTO, 0, &ds) == GSX_OK) gsx_dataset_free(ds);
    if (gsx_read_bal(p.c_str(), 0, &ds) == GSX_OK) gsx_dataset_free(ds);
  }
  {
    // hostile headers: absurd counts must come back as a status (no exception through extern "C", no terminate), and a
    // file truncated in the middle of a record must be refused
    const char* texts[] = {"1 1 9000000000000000000\n", "3 7 4611686018427387904\n0 0 1 1\n", "2000000000 2000000000 8000000000\n",
                           "2 2 3\n0 0 1.5 2.5\n1 1 0.5", "2 2 2\n0 0 1 1\n1 1 2 2\n0.1 0.2 0.3 1 2 3 500 0 0\n0.1 0.2"};
    for (const char* t : texts) {
      const std::string p = tmp + "/hostile_bal.txt";
      FILE* fh = std::fopen(p.c_str(), "w");
      std::fputs(t, fh);
      std::fclose(fh);
      const gsx_status st = gsx_read_bal(p.c_str(), 1, &ds);
      if (st == GSX_OK) {
        std::fprintf(stderr, "hostile BAL header accepted: %s\n", t);
        gsx_dataset_free(ds);
        bad = 1;
      }
    }
    const std::string p = tmp + "/hostile.g2o";
    FILE* fh = std::fopen(p.c_str(), "w");
    std::fputs("VERTEX_SE2 99999999999999999999 0 0 0\nEDGE_SE2 0 18446744073709551615 1 0 0 1 0 0 1 0 1\n", fh);
    std::fclose(fh);
    if (gsx_read_g2o(p.c_str(), 0, &ds) == GSX_OK) gsx_dataset_free(ds);
    if (gsx_load2d(p.c_str(), nullptr, -5, 1, GSX_NOISE_FORMAT_AUTO, 0, &ds) == GSX_OK) gsx_dataset_free(ds);
  }
  return bad;
}
