// Host side of the dense MLP leg (C ABI mhte_dense_mlp_*; kernels: mhte_gemm_kernels.h).  Included
// by mhte.hip behind the table code: uses its DevBuf, Error, HIP_OK, LAUNCH_HOT and profile tags.
//
// An MLP of widths w0 -> w1 -> ... -> w_{L-1} -> 1 (ReLU between the layers, the last one linear
// with ONE output: the ranking tower's logit, native_training/layers/mlp.py): L - 1 GEMM layers on
// MFMA + a row-dot.  fp32 master weights, bf16 operands, fp32 accumulation, SGD inside backward().
// What each GEMM reads K-contiguous is kept that way by its producer (see mhte_gemm_kernels.h):
//   forward   h_l  = relu(h_{l-1} W_l^T + b_l)           A = h_{l-1} [B][K]     B = W_l   [N][K]
//   dgrad     dz_{l-1} = (dz_l W_l) * (h_{l-1} > 0)      A = dz_l    [B][N]     B = W_l^T [K][N]
//   wgrad     dW_l = dz_l^T h_{l-1}                      A = dz_l^T  [N][B]     B = h_{l-1}^T [K][B]
// the weight gradient split along the batch into fp32 slabs (a 1024 x 1024 output is 64 tiles: 8
// slices fill the chip), summed in slice order by the SGD kernel.
#ifndef MHTE_GEMM_HOST_H_
#define MHTE_GEMM_HOST_H_

namespace mhte {

struct DenseMlp {
  int device = 0;
  int64_t max_batch = 0;
  std::vector<uint32_t> widths;   // w0 .. w_{L-1}, then the single output
  uint32_t nl = 0;                // GEMM layers = widths.size() - 1
  struct Layer {
    uint32_t K = 0, N = 0, split = 1, split_now = 1;
    DevBuf<float> w, b, slabs, db;
    DevBuf<uint16_t> wb, wt;        // bf16 W [N][K], W^T [K][N]
    DevBuf<uint16_t> h, ht;         // forward output [B][N] and transposed [N][B]
    DevBuf<uint16_t> dz, dzt;       // gradient at the layer's pre-activation [B][N], [N][B]
  };
  std::vector<Layer> layers;
  DevBuf<float> w_last, b_last, part_last;   // the row-dot layer: w [K], b [1]; per-block partial sums
  DevBuf<uint16_t> xb, xt;                   // the input as bf16 [B][w0] and transposed
  int64_t batch = 0;                         // of the last forward

  void create(const int32_t* w, int32_t n, int64_t mb, int dev) {
    if (n < 2) throw Error(MHTE_INVALID_ARGUMENT, "dense mlp: at least one hidden layer and the output");
    if (w[n - 1] != 1) throw Error(MHTE_INVALID_ARGUMENT, "dense mlp: the last layer has one output (the logit)");
    if (mb <= 0 || mb % 128) throw Error(MHTE_INVALID_ARGUMENT, "dense mlp: max_batch must be a multiple of 128");
    device = dev;
    max_batch = mb;
    nl = uint32_t(n - 2);
    for (int i = 0; i < n - 1; ++i) {
      if (w[i] <= 0 || w[i] % 128)
        throw Error(MHTE_INVALID_ARGUMENT, "dense mlp: layer widths must be multiples of 128 (GEMM tile), got " +
                                               std::to_string(w[i]));
      widths.push_back(uint32_t(w[i]));
    }
    if (nl == 0) throw Error(MHTE_INVALID_ARGUMENT, "dense mlp: at least one hidden layer");
    layers.resize(nl);
    const size_t B = size_t(mb);
    for (uint32_t l = 0; l < nl; ++l) {
      Layer& L = layers[l];
      L.K = widths[l];
      L.N = widths[l + 1];
      // slices of the weight gradient: enough workgroups for two per CU, every slice a multiple of 64 rows
      const uint32_t tiles = (L.N / 128) * (L.K / 128);
      L.split = 1;
      while (tiles * L.split < 512 && L.split < 16 && (B / (L.split * 2)) % 64 == 0) L.split *= 2;
      L.w.reserve(size_t(L.N) * L.K);
      L.b.reserve(L.N);
      L.db.reserve(L.N);
      L.slabs.reserve(size_t(L.split) * L.N * L.K);
      L.wb.reserve(size_t(L.N) * L.K);
      L.wt.reserve(size_t(L.N) * L.K);
      L.h.reserve(B * L.N);
      L.ht.reserve(B * L.N);
      L.dz.reserve(B * L.N);
      L.dzt.reserve(B * L.N);
      HIP_OK(hipMemset(L.w.p, 0, sizeof(float) * L.N * L.K));
      HIP_OK(hipMemset(L.b.p, 0, sizeof(float) * L.N));
      HIP_OK(hipMemset(L.wb.p, 0, 2 * size_t(L.N) * L.K));
      HIP_OK(hipMemset(L.wt.p, 0, 2 * size_t(L.N) * L.K));
    }
    const uint32_t KL = widths[nl];
    w_last.reserve(KL);
    b_last.reserve(1);
    part_last.reserve((B / 64) * (KL + 1));
    HIP_OK(hipMemset(w_last.p, 0, sizeof(float) * KL));
    HIP_OK(hipMemset(b_last.p, 0, sizeof(float)));
    xb.reserve(B * widths[0]);
    xt.reserve(B * widths[0]);
  }

  // layer in [0, nl]: nl = the row-dot layer (w [K], b [1])
  void set_params(int32_t layer, const float* w, const float* b, hipStream_t st) {
    if (layer < 0 || layer > int32_t(nl)) throw Error(MHTE_INVALID_ARGUMENT, "dense mlp: no such layer");
    if (layer == int32_t(nl)) {
      HIP_OK(hipMemcpyAsync(w_last.p, w, sizeof(float) * widths[nl], hipMemcpyDefault, st));
      HIP_OK(hipMemcpyAsync(b_last.p, b, sizeof(float), hipMemcpyDefault, st));
      return;
    }
    Layer& L = layers[size_t(layer)];
    HIP_OK(hipMemcpyAsync(L.w.p, w, sizeof(float) * L.N * L.K, hipMemcpyDefault, st));
    HIP_OK(hipMemcpyAsync(L.b.p, b, sizeof(float) * L.N, hipMemcpyDefault, st));
    const uint64_t n = uint64_t(L.N) * L.K;
    mlp_refresh_kernel<<<uint32_t((n + 255) / 256), 256, 0, st>>>(L.w.p, L.wb.p, L.wt.p, L.N, L.K);
    HIP_OK(hipGetLastError());
  }
  void get_params(int32_t layer, float* w, float* b, hipStream_t st) {
    if (layer < 0 || layer > int32_t(nl)) throw Error(MHTE_INVALID_ARGUMENT, "dense mlp: no such layer");
    if (layer == int32_t(nl)) {
      HIP_OK(hipMemcpyAsync(w, w_last.p, sizeof(float) * widths[nl], hipMemcpyDefault, st));
      HIP_OK(hipMemcpyAsync(b, b_last.p, sizeof(float), hipMemcpyDefault, st));
      return;
    }
    Layer& L = layers[size_t(layer)];
    HIP_OK(hipMemcpyAsync(w, L.w.p, sizeof(float) * L.N * L.K, hipMemcpyDefault, st));
    HIP_OK(hipMemcpyAsync(b, L.b.p, sizeof(float) * L.N, hipMemcpyDefault, st));
  }

  // 256 x 256 tiles (8 wavefronts) when the shape allows, else 128 x 128 (4)
  template <int EPI>
  static void gemm(const GemmArgs& g, uint32_t nsplit, hipStream_t st) {
    static const bool small_only = getenv("MHTE_GEMM_TILE128") != nullptr;   // (A/B runs)
    // (a small output — the weight gradients — is split along K: the larger tile pays only when it
    // still gives every CU a workgroup)
    if (g.M % 256 == 0 && g.N % 256 == 0 && !small_only &&
        uint64_t(g.M / 256) * (g.N / 256) * nsplit >= 256) {
      const dim3 grid(g.N / 256, g.M / 256, nsplit);
      LAUNCH_HOT(kTagGemm, (gemm_nt_bf16_kernel<EPI, 4, 2, 2, 4>), grid, 512, st, g);
    } else {
      const dim3 grid(g.N / 128, g.M / 128, nsplit);
      LAUNCH_HOT(kTagGemm, (gemm_nt_bf16_kernel<EPI, 2, 2, 2, 2>), grid, 256, st, g);
    }
    HIP_OK(hipGetLastError());
  }

  void forward(const float* x, int64_t B, float* y, hipStream_t st) {
    if (B <= 0 || B > max_batch || B % 128)
      throw Error(MHTE_INVALID_ARGUMENT, "dense mlp: batch must be a multiple of 128, at most max_batch");
    batch = B;
    const uint32_t Bu = uint32_t(B);
    mlp_cast_transpose_kernel<<<dim3(widths[0] / 64, Bu / 64), 256, 0, st>>>(x, xb.p, xt.p, Bu, widths[0]);
    HIP_OK(hipGetLastError());
    const uint16_t* in = xb.p;
    for (uint32_t l = 0; l < nl; ++l) {
      Layer& L = layers[l];
      GemmArgs g{};
      g.A = in;
      g.lda = L.K;
      g.B = L.wb.p;
      g.ldb = L.K;
      g.M = Bu;
      g.N = L.N;
      g.klen = L.K;
      g.bias = L.b.p;
      g.relu = 1;
      g.C = L.h.p;
      g.ldc = L.N;
      g.Ct = L.ht.p;
      g.ldct = B;
      gemm<kEpiFwd>(g, 1, st);
      in = L.h.p;
    }
    mlp_rowdot_kernel<<<(Bu + 7) / 8, 256, 0, st>>>(in, w_last.p, b_last.p, y, Bu, widths[nl]);
    HIP_OK(hipGetLastError());
  }

  // dy [B] -> SGD step on every layer; dx [B][w0] (fp32) when not null
  void backward(const float* dy, float* dx, float lr, hipStream_t st) {
    if (batch <= 0) throw Error(MHTE_FAILED_PRECONDITION, "dense mlp: backward without a forward");
    const int64_t B = batch;
    const uint32_t Bu = uint32_t(B);
    const uint32_t KL = widths[nl];
    Layer& top = layers[nl - 1];
    mlp_last_bwd_kernel<<<dim3(KL / 64, Bu / 64), 256, 0, st>>>(dy, top.h.p, w_last.p, top.dz.p, top.dzt.p,
                                                               part_last.p, Bu, KL);
    HIP_OK(hipGetLastError());
    for (int32_t l = int32_t(nl) - 1; l >= 0; --l) {
      Layer& L = layers[size_t(l)];
      const uint16_t* in_t = l ? layers[size_t(l) - 1].ht.p : xt.p;   // h_{l-1}^T [K][B]
      // (slices of this batch: every slice a whole number of 64-row K tiles)
      L.split_now = L.split;
      while (L.split_now > 1 && (B % (int64_t(L.split_now) * 64))) L.split_now >>= 1;
      {  // dW_l = dz_l^T h_{l-1}: M = N_l, N = K_l, reduction = the batch in `split` slices
        GemmArgs g{};
        g.A = L.dzt.p;
        g.lda = B;
        g.B = in_t;
        g.ldb = B;
        g.M = L.N;
        g.N = L.K;
        g.klen = uint32_t(B / L.split_now);
        g.Cf = L.slabs.p;
        g.ldcf = L.K;
        gemm<kEpiWgrad>(g, L.split_now, st);
      }
      mlp_bias_grad_kernel<<<L.N, 256, 0, st>>>(L.dzt.p, L.db.p, Bu);
      HIP_OK(hipGetLastError());
      if (l > 0 || dx) {  // the gradient below: dz_l W_l
        GemmArgs g{};
        g.A = L.dz.p;
        g.lda = L.N;
        g.B = L.wt.p;
        g.ldb = L.N;
        g.M = Bu;
        g.N = L.K;
        g.klen = L.N;
        if (l > 0) {
          Layer& D = layers[size_t(l) - 1];
          g.mask = D.h.p;
          g.C = D.dz.p;
          g.ldc = D.N;
          g.Ct = D.dzt.p;
          g.ldct = B;
          gemm<kEpiDgrad>(g, 1, st);
        } else {
          g.Cf = dx;
          g.ldcf = L.K;
          gemm<kEpiDgradF32>(g, 1, st);
        }
      }
    }
    for (uint32_t l = 0; l < nl; ++l) {
      Layer& L = layers[l];
      const uint64_t n = uint64_t(L.N) * L.K;
      mlp_sgd_kernel<<<uint32_t((n + 255) / 256), 256, 0, st>>>(L.w.p, L.b.p, L.slabs.p, L.split_now, L.db.p, lr, L.wb.p,
                                                                L.wt.p, L.N, L.K);
      HIP_OK(hipGetLastError());
    }
    mlp_sgd_vec_kernel<<<KL + 1, 256, 0, st>>>(w_last.p, b_last.p, part_last.p, Bu / 64, lr, KL);
    HIP_OK(hipGetLastError());
  }

  // flops of one forward + backward (+ the input gradient when asked for): 2 per multiply-add
  double flops(int64_t B, bool with_dx) const {
    double f = 0;
    for (uint32_t l = 0; l < nl; ++l) {
      const double mk = double(layers[l].N) * layers[l].K;
      f += 2.0 * B * mk * ((l > 0 || with_dx) ? 3.0 : 2.0);
    }
    return f + 6.0 * B * widths[nl];
  }
};

}  // namespace mhte
#endif  // MHTE_GEMM_HOST_H_
