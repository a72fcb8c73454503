// CPU emulator of the GatherGemm semantics -- TEST INFRASTRUCTURE ONLY (never linked into the product
// library).  Executes the plans built by the product's conv_plan.h with plain loops so that the index
// math (padding, stride-parity classes, non-overlapping shortcut, transposition, split modes) can be
// checked against torch on a machine without a GPU.
#include <vector>
#include <cstring>
#include "gather_gemm.h"
#include "conv_plan.h"

using namespace srgan;

static void run(const GatherGemm& p) {
  for (int i = 0; i < p.M; ++i) {
    const Side am = decode(p.am, i), cm = decode(p.cm, i);
    for (int j = 0; j < p.N; ++j) {
      const Side bn = decode(p.bn, j), cn = decode(p.cn, j);
      double acc = 0.0;
      for (int k = 0; k < p.K; ++k) {
        const Side ak = decode(p.ak, k), bk = decode(p.bk, k);
        acc += (double)gg_a(p, am, ak) * (double)gg_b(p, bk, bn);
      }
      if (p.bias) acc += p.bias[p.bias_cols ? cn.c : cm.c];
      float* dst = p.C + (uint32_t)(cm.off + cn.off);
      if (p.mode == GG_STORE) *dst = (float)acc; else *dst += (float)acc;
    }
  }
}

extern "C" {

int emul_conv2d_fwd(const ConvGeom* geom, const float* x, const float* w, const float* bias, float* y) {
  ConvGeom g = *geom;
  if (!geom_ok(g) || geom_largest_extent(g) >= ((int64_t)1 << 31)) return -1;
  run(plan_conv_fwd(g, x, w, bias, y));
  return 0;
}

int emul_conv2d_bwd_data(const ConvGeom* geom, const float* gy, const float* w, const float* bias, float* gx,
                         int accumulate) {
  ConvGeom g = *geom;
  if (!geom_ok(g) || geom_largest_extent(g) >= ((int64_t)1 << 31)) return -1;
  std::vector<GatherGemm> plans = plan_conv_bwd_data(g, gy, w, bias, gx);
  for (GatherGemm& p : plans) { p.mode = accumulate ? GG_ACCUMULATE : GG_STORE; run(p); }
  return (int)plans.size();
}

int emul_conv2d_bwd_weight(const ConvGeom* geom, const float* x, const float* gy, float* gw, int accumulate) {
  ConvGeom g = *geom;
  if (!geom_ok(g) || geom_largest_extent(g) >= ((int64_t)1 << 31)) return -1;
  GatherGemm p = plan_conv_bwd_weight(g, x, gy, gw);
  p.mode = accumulate ? GG_ACCUMULATE : GG_STORE;
  run(p);
  return 0;
}

int emul_gemm(int M, int N, int K, const float* A, int sai, int sak, const float* B, int sbk, int sbj, float* C,
              int sci, int scj, const float* bias, int bias_cols, int accumulate) {
  GatherGemm p = plan_gemm(M, N, K, A, sai, sak, B, sbk, sbj, C, sci, scj, bias, bias_cols);
  p.mode = accumulate ? GG_ACCUMULATE : GG_STORE;
  run(p);
  return p.a_kfast * 2 + p.b_kfast;
}

int emul_fastdiv_check(unsigned d, unsigned n) {
  const FastDiv f = make_fastdiv(d);
  return fd_div(n, f) == n / d;
}

}  // extern "C"
