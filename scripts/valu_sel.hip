// valu_sel.hip -- what a float64 select costs in context on gfx950 (companion of valu_peak.hip).
// Pattern per group: v_cmp_lt_f64 -> mask, K1 x v_fma_f64, 2 x v_cndmask_b32 (lo, hi), K2 x v_fma_f64; mask in VCC
// (VOPC e32 + VOP2 e32, what hipcc emits most) or in an SGPR pair (e64 forms).  16 groups per loop trip on 16 accumulators.
// Output: cycles per group per SIMD at W waves per SIMD (wall time x in-kernel clock / groups), and the same minus
// the (K1 + K2) fma at their own measured cost = the cost of the select itself.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define FMA(a) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define REP0(X)
#define REP1(X) X
#define REP2(X) X X
#define REP4(X) X X X X
#define REP8(X) REP4(X) REP4(X)
#define GRP_VCC(K1, K2, a, xl, xh)                                                     \
  asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");                 \
  REP##K1(FMA(a))                                                                      \
  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(xl) : "v"(y) : );                \
  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(xh) : "v"(y) : );                \
  REP##K2(FMA(a))
#define GRP_SGPR(K1, K2, a, xl, xh)                                                    \
  asm volatile("v_cmp_lt_f64_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));              \
  REP##K1(FMA(a))                                                                      \
  asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(xl) : "v"(y), "s"(m));        \
  asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(xh) : "v"(y), "s"(m));        \
  REP##K2(FMA(a))
// select by arithmetic: v_min_f64 (1 instruction)
#define GRP_MIN(K1, K2, a, xl, xh)                                                     \
  asm volatile("v_min_f64 %0, %0, %1" : "+v"(a) : "v"(b));                             \
  REP##K1(FMA(a)) REP##K2(FMA(a))
#define ALL16(G, K1, K2)                                                                                         \
  G(K1, K2, a0, x0, x1) G(K1, K2, a1, x2, x3) G(K1, K2, a2, x4, x5) G(K1, K2, a3, x6, x7) G(K1, K2, a4, x8, x9)  \
  G(K1, K2, a5, x10, x11) G(K1, K2, a6, x12, x13) G(K1, K2, a7, x14, x15) G(K1, K2, a8, x0, x1) G(K1, K2, a9, x2, x3) \
  G(K1, K2, a10, x4, x5) G(K1, K2, a11, x6, x7) G(K1, K2, a12, x8, x9) G(K1, K2, a13, x10, x11) G(K1, K2, a14, x12, x13) \
  G(K1, K2, a15, x14, x15)

template <int KIND, int K1, int K2>
__global__ __launch_bounds__(256) void sel(int trips, unsigned long long* cyc, double* sink) {
  extern __shared__ int pin[];
  double b = 1.0 + 1e-9 * threadIdx.x, c = 1e-12 * (threadIdx.x + 1);
  unsigned y = threadIdx.x | 1u;
  unsigned long long m = 0;
  double a0 = b, a1 = b + 1, a2 = b + 2, a3 = b + 3, a4 = b + 4, a5 = b + 5, a6 = b + 6, a7 = b + 7, a8 = b + 8, a9 = b + 9,
         a10 = b + 10, a11 = b + 11, a12 = b + 12, a13 = b + 13, a14 = b + 14, a15 = b + 15;
  unsigned x0 = y, x1 = y + 1, x2 = y + 2, x3 = y + 3, x4 = y + 4, x5 = y + 5, x6 = y + 6, x7 = y + 7, x8 = y + 8, x9 = y + 9,
           x10 = y + 10, x11 = y + 11, x12 = y + 12, x13 = y + 13, x14 = y + 14, x15 = y + 15;
  if (trips < 0) pin[threadIdx.x] = trips;
  __syncthreads();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < trips; ++i) {
#define BODY(K1, K2)                                                  \
    if constexpr (KIND == 0) { ALL16(GRP_VCC, K1, K2) }               \
    else if constexpr (KIND == 1) { ALL16(GRP_SGPR, K1, K2) }         \
    else { ALL16(GRP_MIN, K1, K2) }
    if constexpr (K1 == 0 && K2 == 0) { BODY(0, 0) }
    else if constexpr (K1 == 0 && K2 == 1) { BODY(0, 1) }
    else if constexpr (K1 == 0 && K2 == 2) { BODY(0, 2) }
    else if constexpr (K1 == 0 && K2 == 4) { BODY(0, 4) }
    else if constexpr (K1 == 0 && K2 == 8) { BODY(0, 8) }
    else if constexpr (K1 == 1 && K2 == 0) { BODY(1, 0) }
    else if constexpr (K1 == 2 && K2 == 0) { BODY(2, 0) }
    else if constexpr (K1 == 4 && K2 == 0) { BODY(4, 0) }
    else if constexpr (K1 == 2 && K2 == 2) { BODY(2, 2) }
    else if constexpr (K1 == 4 && K2 == 4) { BODY(4, 4) }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    const size_t w = (size_t)(blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    cyc[2 * w] = t1 - t0;
    cyc[2 * w + 1] = r1 - r0;
  }
  double s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9 + a10 + a11 + a12 + a13 + a14 + a15;
  unsigned u = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7 ^ x8 ^ x9 ^ x10 ^ x11 ^ x12 ^ x13 ^ x14 ^ x15;
  if (s == 12345.678 && u == 42u && m == 7) sink[0] = s;
}

static int g_first = 1;
template <int KIND, int K1, int K2>
void run(int waves, int n_cu, unsigned long long* d_cyc, double* d_sink) {
  const int trips = 2000;
  const int blocks = n_cu * waves, n_waves = blocks * 4;
  const int lds = 160 * 1024 / waves - 1024;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&sel<KIND, K1, K2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((sel<KIND, K1, K2>), dim3(blocks), dim3(256), lds, 0, trips, d_cyc, d_sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
  }
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> c(2 * n_waves);
  CHECK(hipMemcpy(c.data(), d_cyc, 2 * n_waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::vector<double> clk(n_waves);
  for (int i = 0; i < n_waves; ++i) clk[i] = (double)c[2 * i] / (double)c[2 * i + 1] * 0.1;
  std::sort(clk.begin(), clk.end());
  const double ghz = clk[n_waves / 2];
  const double groups = (double)trips * 16 * waves;                    // per SIMD
  const double cyc_group = ms * 1e6 * ghz / groups;
  printf("%s  {\"mask\": \"%s\", \"fma_between\": %d, \"fma_after\": %d, \"waves_per_simd\": %d, \"cyc_per_group_simd\": %.3f, "
         "\"clock_ghz\": %.3f, \"wall_ms\": %.4f}", g_first ? "" : ",\n", KIND == 0 ? "vcc" : (KIND == 1 ? "sgpr pair" : "v_min_f64 instead"),
         K1, K2, waves, cyc_group, ghz, ms);
  g_first = 0;
}

template <int KIND>
void kind(int n_cu, unsigned long long* d_cyc, double* d_sink) {
  for (int w : {4, 1}) {
    run<KIND, 0, 0>(w, n_cu, d_cyc, d_sink);
    run<KIND, 0, 1>(w, n_cu, d_cyc, d_sink);
    run<KIND, 0, 2>(w, n_cu, d_cyc, d_sink);
    run<KIND, 0, 4>(w, n_cu, d_cyc, d_sink);
    run<KIND, 0, 8>(w, n_cu, d_cyc, d_sink);
    run<KIND, 1, 0>(w, n_cu, d_cyc, d_sink);
    run<KIND, 2, 0>(w, n_cu, d_cyc, d_sink);
    run<KIND, 4, 0>(w, n_cu, d_cyc, d_sink);
    run<KIND, 2, 2>(w, n_cu, d_cyc, d_sink);
    run<KIND, 4, 4>(w, n_cu, d_cyc, d_sink);
  }
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  unsigned long long* d_cyc;
  double* d_sink;
  CHECK(hipMalloc(&d_cyc, (size_t)n_cu * 8 * 4 * 2 * sizeof(unsigned long long)));
  CHECK(hipMalloc(&d_sink, 64));
  printf("{\"device\": \"%s\", \"pattern\": \"v_cmp_lt_f64 -> mask; K1 x v_fma_f64; 2 x v_cndmask_b32; K2 x v_fma_f64 (16 groups per trip)\",\n \"groups\": [\n", prop.gcnArchName);
  kind<0>(n_cu, d_cyc, d_sink);
  kind<1>(n_cu, d_cyc, d_sink);
  kind<2>(n_cu, d_cyc, d_sink);
  printf("\n ]}\n");
  return 0;
}
