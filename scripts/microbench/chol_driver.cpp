// Stand-alone driver of host/block_cholesky.hpp on arrays dumped by `MC_CHOL_DUMP=<dir> python scripts/chol_bench.py`:
//   g++ -O3 -march=x86-64-v3 -std=c++17 -pthread -I multical_amd/csrc -o /tmp/chol_driver scripts/microbench/chol_driver.cpp
//   MC_CHOLESKY_TIMING=1 /tmp/chol_driver <dir> [threads] [repetitions]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "host/block_cholesky.hpp"

template <class T> static std::vector<T> load(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { perror(path.c_str()); exit(1); }
  fseek(f, 0, SEEK_END);
  const long bytes = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<T> v(bytes / sizeof(T));
  if (fread(v.data(), sizeof(T), v.size(), f) != v.size()) { perror("read"); exit(1); }
  fclose(f);
  return v;
}

int main(int argc, char** argv) {
  const std::string d = argc > 1 ? argv[1] : "/tmp/chol";
  const int threads = argc > 2 ? atoi(argv[2]) : multical_amd::hostCholeskyThreads();
  const int reps = argc > 3 ? atoi(argv[3]) : 4;
  auto Cp = load<int>(d + "/Cp.bin");
  auto Ci = load<int>(d + "/Ci.bin");
  auto Cx = load<double>(d + "/Cx.bin");
  auto bb = load<int>(d + "/bb.bin");
  auto b = load<double>(d + "/b.bin");
  auto times = load<double>(d + "/times.bin");
  const int n = (int)Cp.size() - 1, nb = (int)bb.size() - 1;
  multical_amd::BlockSparseCholesky chol;
  chol.analyze(n, Cp.data(), Ci.data(), nb, bb.data());
  chol.setBlockCoordinates(times.data(), threads);
  std::vector<double> x(n);
  for (int r = 0; r < reps; ++r) {
    const auto t0 = std::chrono::steady_clock::now();
    const bool ok = chol.factorize(Cx.data());
    const auto t1 = std::chrono::steady_clock::now();
    if (!ok) { printf("not positive definite\n"); return 1; }
    chol.solve(b.data(), x.data());
    const auto t2 = std::chrono::steady_clock::now();
    printf("factorize %.1f ms, solve %.1f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
           std::chrono::duration<double, std::milli>(t2 - t1).count());
  }
  const auto& st = chol.stats();
  printf("%d blocks, %d panels, factor nnz %lld, flops %.3g, parts %d\n", st.n_blocks, st.n_panels, st.nnz_factor, st.flops, st.n_parts);
  // residual of the symmetric system stored as its upper triangle
  std::vector<double> r(b);
  for (int c = 0; c < n; ++c)
    for (int k = Cp[c]; k < Cp[c + 1]; ++k) {
      r[Ci[k]] -= Cx[k] * x[c];
      if (Ci[k] != c) r[c] -= Cx[k] * x[Ci[k]];
    }
  double rn = 0, bn = 0;
  for (int i = 0; i < n; ++i) { rn += r[i] * r[i]; bn += b[i] * b[i]; }
  printf("relative residual %.2e\n", std::sqrt(rn / bn));
  return 0;
}
