// AddressSanitizer / UBSan driver for the HOST half of the library -- TEST INFRASTRUCTURE ONLY.
// The plan builders of the product's conv_plan.h / gather_gemm.h (plain C++: index decoding, padding, stride-parity
// classes, split modes) are executed by the CPU emulator next to this file on buffers of EXACTLY the tensors' sizes, so
// that an index one element out of range is a sanitizer report, and the three passes are compared with direct loops.
// Built and run by tests/test_conv_plans_cpu.py with  g++ -fsanitize=address,undefined -fno-sanitize-recover=all
// (sanitizers run on the CPU build only; there is no GPU AddressSanitizer on this pool).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "emul_gather_gemm.cpp"

struct Case { int N, C, H, W, K, R, S, sh, sw, ph, pw; };

static float noise(unsigned& state) {
  state = state * 1664525u + 1013904223u;
  return ((state >> 8) & 0xFFFF) / 32768.0f - 1.0f;
}

static int check(const char* what, const std::vector<float>& got, const std::vector<double>& want) {
  for (size_t i = 0; i < got.size(); ++i)
    if (!(std::fabs(got[i] - want[i]) <= 1e-4 * (1.0 + std::fabs(want[i])))) {
      std::printf("MISMATCH %s at %zu: %g vs %g\n", what, i, got[i], want[i]);
      return 1;
    }
  return 0;
}

int main() {
  const Case cases[] = {
      {2, 3, 9, 8, 5, 3, 3, 1, 1, 1, 1},   {2, 4, 7, 7, 6, 1, 1, 1, 1, 0, 0},   {2, 3, 12, 10, 4, 4, 4, 2, 2, 1, 1},
      {1, 3, 15, 13, 4, 7, 7, 2, 2, 3, 3}, {2, 2, 8, 8, 3, 2, 2, 2, 2, 0, 0},   {3, 5, 4, 6, 7, 4, 6, 1, 1, 0, 0},
      {1, 2, 9, 9, 2, 3, 3, 2, 2, 0, 0},   {2, 3, 6, 5, 2, 3, 2, 1, 2, 1, 0},   {1, 2, 10, 10, 3, 3, 3, 3, 3, 1, 1},
      {1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0},   {2, 33, 5, 5, 65, 3, 3, 1, 1, 1, 1}, {1, 3, 66, 200, 2, 4, 4, 2, 2, 1, 1}};
  int failures = 0, launches = 0;
  unsigned state = 7;
  for (const Case& c : cases) {
    const int OH = (c.H + 2 * c.ph - c.R) / c.sh + 1, OW = (c.W + 2 * c.pw - c.S) / c.sw + 1;
    ConvGeom g{c.N, c.C, c.H, c.W, c.K, c.R, c.S, c.sh, c.sw, c.ph, c.pw, OH, OW, 0, 0};
    std::vector<float> x((size_t)c.N * c.C * c.H * c.W), w((size_t)c.K * c.C * c.R * c.S), bias(c.K),
        y((size_t)c.N * c.K * OH * OW), gy(y.size()), gx(x.size()), gw(w.size());
    for (float& v : x) v = noise(state);
    for (float& v : w) v = noise(state);
    for (float& v : bias) v = noise(state);
    for (float& v : gy) v = noise(state);
    std::vector<double> y_ref(y.size()), gx_ref(x.size(), 0.0), gw_ref(w.size(), 0.0);
    for (int n = 0; n < c.N; ++n)
      for (int k = 0; k < c.K; ++k)
        for (int oh = 0; oh < OH; ++oh)
          for (int ow = 0; ow < OW; ++ow) {
            double acc = bias[k];
            const size_t yi = (((size_t)n * c.K + k) * OH + oh) * OW + ow;
            for (int ci = 0; ci < c.C; ++ci)
              for (int r = 0; r < c.R; ++r)
                for (int s = 0; s < c.S; ++s) {
                  const int ih = oh * c.sh - c.ph + r, iw = ow * c.sw - c.pw + s;
                  if (ih < 0 || ih >= c.H || iw < 0 || iw >= c.W) continue;
                  const size_t xi = (((size_t)n * c.C + ci) * c.H + ih) * c.W + iw;
                  const size_t wi = (((size_t)k * c.C + ci) * c.R + r) * c.S + s;
                  acc += (double)x[xi] * w[wi];
                  gx_ref[xi] += (double)gy[yi] * w[wi];
                  gw_ref[wi] += (double)gy[yi] * x[xi];
                }
            y_ref[yi] = acc;
          }
    if (emul_conv2d_fwd(&g, x.data(), w.data(), bias.data(), y.data()) != 0) { std::printf("forward refused\n"); ++failures; }
    failures += check("forward", y, y_ref);
    const int classes = emul_conv2d_bwd_data(&g, gy.data(), w.data(), nullptr, gx.data(), 0);
    if (classes < 1) { std::printf("data gradient refused\n"); ++failures; }
    launches += classes;
    failures += check("data gradient", gx, gx_ref);
    if (emul_conv2d_bwd_weight(&g, x.data(), gy.data(), gw.data(), 0) != 0) { std::printf("weight gradient refused\n"); ++failures; }
    failures += check("weight gradient", gw, gw_ref);
  }
  // strided GEMM plan, both operand orders, with a bias on rows and on columns
  for (int transposed = 0; transposed < 2; ++transposed) {
    const int M = 7, N = 5, K = 11;
    std::vector<float> A((size_t)M * K), B((size_t)K * N), C((size_t)M * N), bias_rows(M), bias_cols(N);
    for (float& v : A) v = noise(state);
    for (float& v : B) v = noise(state);
    for (float& v : bias_rows) v = noise(state);
    for (float& v : bias_cols) v = noise(state);
    std::vector<double> want((size_t)M * N);
    for (int i = 0; i < M; ++i)
      for (int j = 0; j < N; ++j) {
        double acc = transposed ? bias_cols[j] : bias_rows[i];
        for (int k = 0; k < K; ++k)
          acc += (double)(transposed ? A[(size_t)k * M + i] : A[(size_t)i * K + k]) * (transposed ? B[(size_t)j * K + k] : B[(size_t)k * N + j]);
        want[(size_t)i * N + j] = acc;
      }
    emul_gemm(M, N, K, A.data(), transposed ? 1 : K, transposed ? M : 1, B.data(), transposed ? 1 : N, transposed ? K : 1, C.data(),
              N, 1, transposed ? bias_cols.data() : bias_rows.data(), transposed, 0);
    failures += check("gemm", C, want);
  }
  for (unsigned d = 1; d < 300; ++d)
    for (unsigned n : {0u, 1u, d - 1, d, d + 1, 65535u, 1u << 20, (1u << 31) - 1})
      if (!emul_fastdiv_check(d, n)) { std::printf("fastdiv %u / %u\n", n, d); ++failures; }
  std::printf("%s: %d stride classes executed, %d failures\n", failures ? "FAILED" : "ok", launches, failures);
  return failures ? 1 : 0;
}
