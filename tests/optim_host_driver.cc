// CPU-side unit driver for the per-element optimizer arithmetic in monolith_amd/csrc/mhte_core.h
// (the functions are shared by host and device; compiled by tests/test_optimizer_steps_host.py with
// g++ -DMHTE_HOST_ONLY -ffp-contract=off).  Reads one case from a binary file, applies `steps`
// Optimize() calls to one fresh row the way apply_row (mhte_kernels.h) does, prints the weights as
// hex words.  The test compares them, bit for bit, with oracle/ driven through the same calls.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "mhte_core.h"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 3;
  int32_t hdr[3];  // opt, dim, steps
  float p[8], lr;
  if (std::fread(hdr, sizeof(hdr), 1, f) != 1 || std::fread(p, sizeof(p), 1, f) != 1 ||
      std::fread(&lr, sizeof(lr), 1, f) != 1)
    return 4;
  const int opt = hdr[0], dim = hdr[1], steps = hdr[2];
  std::vector<float> g(size_t(dim) * steps);
  if (std::fread(g.data(), sizeof(float), g.size(), f) != g.size()) return 5;
  std::fclose(f);

  mhte::SegDesc sd;
  std::memset(&sd, 0, sizeof(sd));
  sd.opt = opt;
  sd.dim = dim;
  for (int i = 0; i < 8; ++i) sd.p[i] = p[i];
  std::vector<float> w(dim, 0.f), s1(dim), s2(dim), s3(dim);
  for (int i = 0; i < dim; ++i) {
    s1[i] = mhte::opt_state_init(sd, 0);
    s2[i] = mhte::opt_state_init(sd, 1);
    s3[i] = mhte::opt_state_init(sd, 2);
  }
  float c1 = p[0], c2 = p[1];  // adam / amsgrad: the powers start at beta1, beta2
  for (int t = 0; t < steps; ++t) {
    const float* gt = g.data() + size_t(t) * dim;
    const float lr_eff = mhte::opt_scalars(opt) ? mhte::adam_lr(lr, c1, c2) : lr;
    for (int i = 0; i < dim; ++i) {
      switch (opt) {
        case mhte::kOptSgd: w[i] = mhte::sgd_step(w[i], gt[i], lr); break;
        case mhte::kOptAdagrad: mhte::adagrad_any(w[i], s1[i], gt[i], lr, p[1], p[2], uint32_t(i), uint32_t(dim)); break;
        case mhte::kOptFtrl: mhte::ftrl_step(w[i], s1[i], s2[i], gt[i], lr, p[1], p[2], p[3]); break;
        case mhte::kOptMomentum: mhte::momentum_step(w[i], s1[i], gt[i], lr, p[0], p[1], p[2] != 0.f); break;
        case mhte::kOptAdadelta: mhte::adadelta_step(w[i], s1[i], s2[i], gt[i], lr, p[0], p[1], p[2]); break;
        case mhte::kOptRmsprop: mhte::rmsprop_step(w[i], s1[i], gt[i], double(p[2]), p[0], p[1], false); break;
        case mhte::kOptRmspropV2: mhte::rmsprop_step(w[i], s1[i], gt[i], double(lr), p[0], p[1], true); break;
        case mhte::kOptAdam:
          mhte::adam_step(w[i], s1[i], s2[i], nullptr, gt[i], lr_eff, p[0], p[1], p[2], p[3], p[4] != 0.f);
          break;
        case mhte::kOptAmsgrad:
          mhte::adam_step(w[i], s1[i], s2[i], &s3[i], gt[i], lr_eff, p[0], p[1], p[2], p[3], p[4] != 0.f);
          break;
        case mhte::kOptMovingAverage: w[i] = mhte::moving_average_step(w[i], gt[i], p[0]); break;
        default: return 6;
      }
    }
    if (mhte::opt_scalars(opt)) {
      c1 = c1 * p[0];
      c2 = c2 * p[1];
    }
  }
  for (int i = 0; i < dim; ++i) {
    uint32_t u;
    std::memcpy(&u, &w[i], 4);
    std::printf("%08x\n", u);
  }
  return 0;
}
