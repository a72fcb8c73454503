// conv_plan.h -- host-side builders that express each pass of a 2-D convolution (and plain GEMM) as
// GatherGemm instances.  Pure C++ (no HIP) so tests can run the same index math on the CPU.
//
// Tensor layouts are the reference's: activations NCHW fp32 (torch.nn.Conv2d, e.g. reference
// crowd/models.py:340-345), weights [K, C, R, S].  A batch stride may exceed C*H*W (channel-slice view
// of a wider buffer).
#pragma once
#include <vector>
#include "gather_gemm.h"

namespace srgan {

struct ConvGeom {
  int32_t N, C, H, W;          // input  [N, C, H, W]
  int32_t K, R, S;             // weight [K, C, R, S]
  int32_t sh, sw, ph, pw;      // stride, zero padding
  int32_t OH, OW;              // output [N, K, OH, OW]
  int64_t x_bs, y_bs;          // batch strides in elements (0 => dense)
};

inline bool geom_ok(ConvGeom& g) {
  if (g.N <= 0 || g.C <= 0 || g.H <= 0 || g.W <= 0 || g.K <= 0 || g.R <= 0 || g.S <= 0) return false;
  if (g.sh <= 0 || g.sw <= 0 || g.ph < 0 || g.pw < 0 || g.OH <= 0 || g.OW <= 0) return false;
  if (g.x_bs == 0) g.x_bs = (int64_t)g.C * g.H * g.W;
  if (g.y_bs == 0) g.y_bs = (int64_t)g.K * g.OH * g.OW;
  // Every forward tap of every output must come from the declared geometry.
  if ((int64_t)(g.OH - 1) * g.sh - g.ph + g.R - 1 > (int64_t)g.H - 1 + g.ph) return false;
  if ((int64_t)(g.OW - 1) * g.sw - g.pw + g.S - 1 > (int64_t)g.W - 1 + g.pw) return false;
  return true;
}

// The gather-GEMM addresses with 32-bit element offsets: the largest of the three tensors' extents (in elements), to be
// compared with 2^31 -- the caller reports SRGAN_ERANGE with this number (a batch stacked too high; split it).
inline int64_t geom_largest_extent(const ConvGeom& g) {
  int64_t largest = g.x_bs * g.N;
  if (g.y_bs * g.N > largest) largest = g.y_bs * g.N;
  const int64_t weights = (int64_t)g.K * g.C * g.R * g.S;
  return weights > largest ? weights : largest;
}

inline int32_t innermost_run(const Dec3& d) {
  // Length of the unit-stride run along this index role (0 if its fastest component is strided).
  if (d.div_b.d > 1) return d.off_b == 1 ? (int32_t)d.div_b.d : 0;
  const int32_t A = (int32_t)(d.div_ab.d / d.div_b.d);
  if (A > 1) return d.off_a == 1 ? A : 0;
  return d.off_c == 1 ? d.limit : 0;
}

inline void choose_staging(GatherGemm& p) {
  p.a_kfast = innermost_run(p.ak) >= innermost_run(p.am) ? 1 : 0;
  p.b_kfast = innermost_run(p.bk) > innermost_run(p.bn) ? 1 : 0;
  // A handful of rows / columns against a long K (weight gradients of the k = stride map-head convolutions): lanes
  // along m / n would be almost all masked, lanes along k are all busy.
  if (p.M > 0 && p.M <= 16 && p.K >= 64) p.a_kfast = 1;
  if (p.N > 0 && p.N <= 16 && p.K >= 64) p.b_kfast = 1;
}

inline bool pointwise(const ConvGeom& g) {
  return g.R == 1 && g.S == 1 && g.ph == 0 && g.pw == 0 && g.sh == 1 && g.sw == 1;
}

// A 1x1 / stride-1 / unpadded convolution never leaves the image: drop the halo test (enables 16-byte staging).
inline void strip_halo(GatherGemm& p) {
  p.hlim = 1; p.wlim = 1;
  p.bk.h_a = p.bk.h0 = p.bk.w_b = p.bk.w0 = 0;
  p.bn.h_a = p.bn.h0 = p.bn.w_b = p.bn.w0 = 0;
}

inline GatherGemm gg_blank() {
  GatherGemm p;
  p.A = nullptr; p.B = nullptr; p.C = nullptr; p.bias = nullptr; p.bias_cols = 0;
  p.hlim = 1; p.wlim = 1; p.M = p.N = p.K = 0; p.a_kfast = 1; p.b_kfast = 0;
  p.mode = GG_STORE; p.split_k = 1; p.k_per_split = 0; p.debug = 0; p.partial = nullptr; p.use_partial = 0;
  p.b_unique = 0; p.precision = 0;
  p.am = p.ak = p.bk = p.bn = p.cm = p.cn = dec_linear(0, 0);
  return p;
}

// Swap the roles of the two operands: computes the same C with rows<->columns exchanged, so that the
// lane-contiguous (column) dimension is whichever is contiguous in memory.  Only valid when A carries no
// halo test, which is true of every plan below before transposition; after it the halo test moves to A,
// so transposition is only applied to plans whose B has hlim == wlim == 1.
inline GatherGemm gg_transposed(const GatherGemm& p) {
  GatherGemm t = p;
  t.A = p.B; t.am = p.bn; t.ak = p.bk;
  t.B = p.A; t.bn = p.am; t.bk = p.ak;
  t.cm = p.cn; t.cn = p.cm;
  t.M = p.N; t.N = p.M;
  t.bias_cols = p.bias_cols ? 0 : 1;
  t.b_unique = 0;
  return t;
}

// y[n,k,oh,ow] = sum_{c,r,s} w[k,c,r,s] * x[n,c,oh*sh-ph+r,ow*sw-pw+s] (+ bias[k])
inline GatherGemm plan_conv_fwd(const ConvGeom& g, const float* x, const float* w, const float* bias, float* y) {
  GatherGemm p = gg_blank();
  const int32_t CRS = g.C * g.R * g.S, OHW = g.OH * g.OW;
  p.M = g.K; p.N = g.N * OHW; p.K = CRS;
  p.A = w; p.am = dec_linear(g.K, CRS); p.ak = dec_linear(CRS, 1);
  p.B = x;
  p.bk = dec_3d(g.C, g.R, g.S, g.H * g.W, g.W, 1, 0, 1, 0, 1, 0);
  p.bn = dec_3d(g.N, g.OH, g.OW, (int32_t)g.x_bs, g.sh * g.W, g.sw, -g.ph * g.W - g.pw, g.sh, -g.ph, g.sw, -g.pw);
  p.hlim = g.H; p.wlim = g.W;
  p.C = y; p.cm = dec_linear(g.K, OHW); p.cn = dec_3d(g.N, g.OH, g.OW, (int32_t)g.y_bs, g.OW, 1, 0, 0, 0, 0, 0);
  p.bias = bias;
  p.b_unique = (int64_t)g.N * g.C * g.H * g.W;
  if (pointwise(g)) strip_halo(p);
  choose_staging(p);
  return p;
}

inline bool non_overlapping(const ConvGeom& g) {
  if (g.ph != 0 || g.pw != 0) return false;
  const bool tiled = g.R == g.sh && g.S == g.sw && g.H == g.OH * g.sh && g.W == g.OW * g.sw;
  const bool whole = g.OH == 1 && g.OW == 1 && g.R == g.H && g.S == g.W;
  return tiled || whole;
}

// gx[n,c,ih,iw] = sum_{k,r,s} w[k,c,r,s] * gy[n,k,(ih+ph-r)/sh,(iw+pw-s)/sw]   (exact divisions only)
// (+ bias[c]: used when this pass is the forward of a transposed convolution).
// Strided convolutions are split into sh*sw parity classes so that no multiply is spent on a tap that
// can never align; each class is one GatherGemm.  Non-overlapping kernels are a single plain GEMM.
inline std::vector<GatherGemm> plan_conv_bwd_data(const ConvGeom& g, const float* gy, const float* w,
                                                  const float* bias, float* gx) {
  std::vector<GatherGemm> plans;
  const int32_t RS = g.R * g.S, CRS = g.C * RS, OHW = g.OH * g.OW, HW = g.H * g.W;
  if (non_overlapping(g)) {
    GatherGemm p = gg_blank();
    p.M = CRS; p.N = g.N * OHW; p.K = g.K;
    p.A = w; p.am = dec_linear(CRS, 1); p.ak = dec_linear(g.K, CRS);
    p.B = gy; p.bk = dec_linear(g.K, OHW);
    p.bn = dec_3d(g.N, g.OH, g.OW, (int32_t)g.y_bs, g.OW, 1, 0, 0, 0, 0, 0);
    p.C = gx; p.cm = dec_3d(g.C, g.R, g.S, HW, g.W, 1, 0, 0, 0, 0, 0);
    p.cn = dec_3d(g.N, g.OH, g.OW, (int32_t)g.x_bs, g.sh * g.W, g.sw, 0, 0, 0, 0, 0);
    p.bias = bias;
    choose_staging(p);
    if (OHW == 1) {           // "linear" convolution: the contiguous output dimension is (c,r,s)
      p = gg_transposed(p);
      choose_staging(p);
    }
    plans.push_back(p);
    return plans;
  }
  for (int32_t ch = 0; ch < g.sh; ++ch) {
    for (int32_t cw = 0; cw < g.sw; ++cw) {
      // Input rows ih with (ih + ph) % sh == ch:  ih = sh*qh + ch - ph,  qh in [qh0, qh1].
      const int32_t qh0 = (g.ph - ch) > 0 ? (g.ph - ch + g.sh - 1) / g.sh : 0;
      const int32_t qw0 = (g.pw - cw) > 0 ? (g.pw - cw + g.sw - 1) / g.sw : 0;
      const int32_t tophs = g.H - 1 + g.ph - ch, topws = g.W - 1 + g.pw - cw;
      if (tophs < 0 || topws < 0) continue;
      const int32_t nqh = tophs / g.sh - qh0 + 1, nqw = topws / g.sw - qw0 + 1;
      if (nqh <= 0 || nqw <= 0) continue;
      const int32_t nth = ch < g.R ? (g.R - ch + g.sh - 1) / g.sh : 0;
      const int32_t ntw = cw < g.S ? (g.S - cw + g.sw - 1) / g.sw : 0;
      GatherGemm p = gg_blank();
      p.M = g.C; p.N = g.N * nqh * nqw; p.K = g.K * nth * ntw;
      p.A = w; p.am = dec_linear(g.C, RS);
      p.B = gy;
      if (p.K > 0) {
        p.ak = dec_3d(g.K, nth, ntw, CRS, g.sh * g.S, g.sw, ch * g.S + cw, 0, 0, 0, 0);
        p.bk = dec_3d(g.K, nth, ntw, OHW, -g.OW, -1, 0, -1, 0, -1, 0);
      } else {
        p.ak = dec_linear(0, 0); p.bk = dec_linear(0, 0);
      }
      p.bn = dec_3d(g.N, nqh, nqw, (int32_t)g.y_bs, g.OW, 1, qh0 * g.OW + qw0, 1, qh0, 1, qw0);
      p.hlim = g.OH; p.wlim = g.OW;
      p.C = gx; p.cm = dec_linear(g.C, HW);
      p.cn = dec_3d(g.N, nqh, nqw, (int32_t)g.x_bs, g.sh * g.W, g.sw,
                    (g.sh * qh0 + ch - g.ph) * g.W + (g.sw * qw0 + cw - g.pw), 0, 0, 0, 0);
      p.bias = bias;
      p.b_unique = (int64_t)g.N * g.K * OHW / (g.sh * g.sw);      // the classes share one read of gy
      if (pointwise(g)) strip_halo(p);
      choose_staging(p);
      plans.push_back(p);
    }
  }
  return plans;
}

// gw[k,c,r,s] = sum_{n,oh,ow} gy[n,k,oh,ow] * x[n,c,oh*sh-ph+r,ow*sw-pw+s]
inline GatherGemm plan_conv_bwd_weight(const ConvGeom& g, const float* x, const float* gy, float* gw) {
  GatherGemm p = gg_blank();
  const int32_t CRS = g.C * g.R * g.S, OHW = g.OH * g.OW;
  p.M = g.K; p.N = CRS; p.K = g.N * OHW;
  p.A = gy; p.am = dec_linear(g.K, OHW);
  p.ak = dec_3d(g.N, g.OH, g.OW, (int32_t)g.y_bs, g.OW, 1, 0, 0, 0, 0, 0);
  p.B = x;
  p.bk = dec_3d(g.N, g.OH, g.OW, (int32_t)g.x_bs, g.sh * g.W, g.sw, -g.ph * g.W - g.pw, g.sh, -g.ph, g.sw, -g.pw);
  p.bn = dec_3d(g.C, g.R, g.S, g.H * g.W, g.W, 1, 0, 1, 0, 1, 0);
  p.hlim = g.H; p.wlim = g.W;
  p.C = gw; p.cm = dec_linear(g.K, CRS); p.cn = dec_linear(CRS, 1);
  p.b_unique = (int64_t)g.N * g.C * g.H * g.W;
  if (pointwise(g)) strip_halo(p);
  choose_staging(p);
  return p;
}

// C[i*sci + j*scj] = sum_k A[i*sai + k*sak] * B[k*sbk + j*sbj]; rows/columns are exchanged when that makes
// the lane (column) dimension the contiguous one of C.
inline GatherGemm plan_gemm(int32_t M, int32_t N, int32_t K, const float* A, int32_t sai, int32_t sak,
                            const float* B, int32_t sbk, int32_t sbj, float* C, int32_t sci, int32_t scj,
                            const float* bias, int32_t bias_cols) {
  GatherGemm p = gg_blank();
  p.M = M; p.N = N; p.K = K;
  p.A = A; p.am = dec_linear(M, sai); p.ak = dec_linear(K, sak);
  p.B = B; p.bk = dec_linear(K, sbk); p.bn = dec_linear(N, sbj);
  p.C = C; p.cm = dec_linear(M, sci); p.cn = dec_linear(N, scj);
  p.bias = bias; p.bias_cols = bias_cols;
  choose_staging(p);
  return p;
}

}  // namespace srgan
