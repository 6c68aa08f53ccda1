// bench_step.cpp — the composite step of bench.py (qtr_register_pair_corr: front end of a scan pair + back end on given
// correspondences, inputs resident in HBM) driven from C++ through the C ABI: what a compiled caller — the reference is one — pays per
// registration, without the two ctypes transitions of the Python harness.  Built with hipcc (it allocates the device
// buffers itself); bench.py's `cpp` leg builds and runs it and quotes its line.
//
//   bench_step <dir> <pool> <steps> <warmup>        <dir>/pair<k>_{src,tgt}.bin (KITTI records), pair<k>_{cs,ct}.bin
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "quatro_hip.h"

static std::vector<float> read_f32(const std::string& path) {
  std::vector<float> v;
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) {
    fprintf(stderr, "cannot open %s\n", path.c_str());
    exit(2);
  }
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  v.resize((size_t)n / 4);
  if (fread(v.data(), 4, v.size(), f) != v.size()) exit(2);
  fclose(f);
  return v;
}
static float* to_device(const std::vector<float>& h) {
  float* d = nullptr;
  if (hipMalloc((void**)&d, h.size() * 4) != hipSuccess || hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
    fprintf(stderr, "device allocation failed\n");
    exit(3);
  }
  return d;
}

int main(int argc, char** argv) {
  if (argc < 5) {
    fprintf(stderr, "usage: %s <dir> <pool> <steps> <warmup>\n", argv[0]);
    return 2;
  }
  const std::string dir = argv[1];
  const int pool = atoi(argv[2]), steps = atoi(argv[3]), warmup = atoi(argv[4]);
  struct Pair {
    float *src, *tgt, *cs, *ct;
    int Ps, Pt, L;
  };
  std::vector<Pair> P;
  for (int k = 0; k < pool; ++k) {
    const std::string b = dir + "/pair" + std::to_string(k);
    const auto s = read_f32(b + "_src.bin"), t = read_f32(b + "_tgt.bin"), cs = read_f32(b + "_cs.bin"), ct = read_f32(b + "_ct.bin");
    P.push_back({to_device(s), to_device(t), to_device(cs), to_device(ct), (int)s.size() / 4, (int)t.size() / 4, (int)cs.size() / 4});
  }
  qtr_limits lim;
  qtr_default_limits(&lim);
  lim.max_points = 131072;
  lim.max_voxels = 32768;
  lim.max_corr = 8192;
  qtr_handle* h = nullptr;
  if (qtr_create(0, &lim, &h) != QTR_OK) {
    fprintf(stderr, "qtr_create: %s\n", h ? qtr_last_error(h) : "?");
    return 3;
  }
  qtr_params prm;
  qtr_demo_params(&prm);
  qtr_set_stage_events(h, 0);
  qtr_set_nn_event_stride(h, 0);
  qtr_result res;
  long long checksum = 0;
  auto step = [&](int k) {
    const Pair& p = P[(size_t)k % P.size()];
    qtr_frontend_params fp;
    qtr_default_frontend_params(&fp);
    fp.seed = (unsigned long long)(k % (int)P.size());
    int Lm = 0;
    const int rc = qtr_register_pair_corr(h, 0, p.src, p.Ps, p.tgt, p.Pt, &fp, p.cs, p.ct, p.L, &prm, &res, &Lm, nullptr, nullptr,
                                          0, QTR_MEM_DEVICE);
    if (rc != QTR_OK && rc != QTR_ERR_CLIQUE_TOO_SMALL) {
      fprintf(stderr, "qtr_register_pair_corr: %s\n", qtr_last_error(h));
      exit(4);
    }
    const int ns = res.n_src, nt = res.n_tgt;
    checksum += ns + nt + Lm + res.n_clique + res.n_final;
  };
  for (int k = 0; k < warmup; ++k) step(k);
  (void)hipDeviceSynchronize();
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < steps; ++k) step(k);
  (void)hipDeviceSynchronize();
  const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("{\"driver\": \"C++ through the C ABI (tests/bench_cpp/bench_step.cpp)\", \"steps\": %d, \"ms_per_step\": %.6f, "
         "\"value\": %.3f, \"unit\": \"registrations/s\", \"n_corr\": %d, \"n_clique_last\": %d, \"checksum\": %lld}\n",
         steps, 1e3 * el / steps, steps / el, P[0].L, res.n_clique, checksum);
  qtr_destroy(h);
  return 0;
}
