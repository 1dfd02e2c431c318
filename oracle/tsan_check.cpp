// TEST INFRASTRUCTURE ONLY.  Race check of the oracle's threaded evaluateError (the reference's boost::thread fan-out,
// aslam_backend/src/LinearSystemSolver.cpp:50-78): builds a small camera + IMU + LiDAR problem straight through the C API
// and evaluates it on 1 and on 7 threads.  Built and run under ThreadSanitizer by `make -C oracle tsan`
// (g++ -fsanitize=thread; host code only); also asserts that both evaluations give the same cost and error vector.
#include <cstdio>
#include <random>
#include "oracle_capi.cpp"

int main() {
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  void* h = orc_create();
  const int order = 6, nseg = 60;
  std::vector<double> knots;
  for (int i = 0; i < nseg + 2 * order - 1; ++i) knots.push_back((i - (order - 1)) * 0.01);
  const int nc = nseg + order - 1;
  std::vector<double> c6(6 * nc), c3(3 * nc);
  for (int i = 0; i < nc; ++i) for (int d = 0; d < 6; ++d) c6[6 * i + d] = 0.3 * std::sin(0.1 * i + d) + 0.01 * U(rng) + (d == 0 ? 3.0 : 0.0);
  for (auto& v : c3) v = 0.01 * U(rng);
  int pose = orc_add_spline(h, order, 6, knots.data(), (int)knots.size(), c6.data(), nc, 1);
  int bias = orc_add_spline(h, order, 3, knots.data(), (int)knots.size(), c3.data(), nc, 1);
  double q[4] = {0.01, -0.02, 0.03, 1.0}, t3[3] = {0.1, -0.05, 0.02}, g[3] = {0, 0, -9.8}, z1[1] = {0.001};
  double n2 = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (double& v : q) v /= n2;
  int dq = orc_add_dv(h, 1, q, 4, 1, 1.0), dt = orc_add_dv(h, 0, t3, 3, 1, 1.0), dto = orc_add_dv(h, 2, z1, 1, 1, 1.0);
  int bq = orc_add_dv(h, 1, q, 4, 1, 1.0), bt = orc_add_dv(h, 0, t3, 3, 1, 1.0), dg = orc_add_dv(h, 3, g, 3, 1, 1.0);
  double proj[4] = {460, 460, 376, 240}, dist[4] = {-0.28, 0.07, 2e-4, 2e-5};
  int dp = orc_add_dv(h, 5, proj, 4, 0, 1.0), dd = orc_add_dv(h, 6, dist, 4, 0, 1.0);
  int cam = orc_add_camera(h, 0, 1, dp, dd);
  // 40 observations x 13 corners: thread cuts fall inside observations unless the share groups move them
  const int nObs = 40, per = 13;
  std::vector<double> ot(nObs), y(2 * nObs * per), p(3 * nObs * per), s(nObs * per, 1.0);
  std::vector<int> oq(nObs, bq), otv(nObs, bt), of(nObs + 1);
  for (int o = 0; o < nObs; ++o) { ot[o] = 0.05 + 0.012 * o; of[o] = o * per; }
  of[nObs] = nObs * per;
  for (size_t i = 0; i < p.size(); i += 3) { p[i] = 0.3 * U(rng); p[i + 1] = 0.3 * U(rng); p[i + 2] = 4.0 + 0.2 * U(rng); }
  for (auto& v : y) v = 300 + 50 * U(rng);
  if (orc_add_reprojection_batch(h, cam, pose, dq, dt, dto, nObs, ot.data(), oq.data(), otv.data(), of.data(), y.data(), p.data(), s.data(), 0.03, 1, 1.0) != 0) return 2;
  const int nImu = 101;
  std::vector<double> ti(nImu), m(3 * nImu), si(nImu, 2.0);
  for (int k = 0; k < nImu; ++k) ti[k] = 0.02 + 0.0055 * k;
  for (auto& v : m) v = U(rng);
  if (orc_add_accel_batch(h, pose, bias, -1, dg, dq, dt, nImu, ti.data(), m.data(), si.data(), 1, 1.0) != 0) return 3;
  if (orc_add_gyro_batch(h, pose, bias, -1, dq, nImu, ti.data(), m.data(), si.data(), 0, 0.0) != 0) return 4;
  if (orc_finalize(h) != 0) { std::printf("finalize: %s\n", orc_last_error(h)); return 5; }
  int JR, JC, nb, nt;
  orc_dims(h, &JR, &JC, &nb, &nt);
  std::vector<double> e1(JR), e7(JR);
  double J1 = orc_evaluate_error(h, 1, 1, e1.data());
  for (int rep = 0; rep < 5; ++rep) {
    double J7 = orc_evaluate_error(h, 7, 1, e7.data());
    if (!(std::fabs(J7 - J1) <= 1e-12 * std::fabs(J1)) || e1 != e7) { std::printf("threaded evaluation differs: %.17g vs %.17g\n", J7, J1); return 1; }
  }
  std::printf("tsan_check ok: %d terms, J = %.6f on 1 and on 7 threads\n", nt, J1);
  orc_destroy(h);
  return 0;
}
