// valu_peak.hip -- calibration of the vector-ALU issue peak of gfx950 for the instruction classes the
// ray-tracing kernels are made of ("make the roofline a bound").
//
//   hipcc --offload-arch=gfx950 -O3 -o scripts/valu_peak scripts/valu_peak.hip
//   ./scripts/valu_peak                      -> one JSON object on stdout
//   rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -- ./scripts/valu_peak
//
// Every kernel is a stream of INDEPENDENT instructions of one class (16 accumulators, 64 instructions
// per loop trip, written in inline assembly so the compiler cannot fold or re-class them), launched the
// way the tracing kernels run: 256-thread blocks, W waves per SIMD resident (W = 4: the flat kernels;
// also 1, 2, 8), every CU busy.  Each wave stamps s_memtime around its loop; reported:
//   cyc_per_inst_simd = W x (median wave cycles) / instructions per wave ... cycles one SIMD spends per
//                       wave64 instruction with W waves interleaved (2 = full rate of the SIMD-32)
// plus the wall-clock rate (hipEvents) and the shader clock it implies.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum Op { FMA_F64, MUL_F64, ADD_F64, MINMAX_F64, CMP_F64, CNDMASK_B32, MOV_B32, ADD_U32, FMA_F32, RCP_F64, RSQ_F64,
          MIX_TRACE, MIX_CMPSEL, N_OPS };
static const char* kNames[N_OPS] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_min_f64/v_max_f64", "v_cmp_lt_f64",
                                    "v_cndmask_b32", "v_mov_b32", "v_add_u32", "v_fma_f32", "v_rcp_f64", "v_rsq_f64",
                                    "mix 40% f64 arithmetic (fma/mul/add 23:10:5 + 2 rcp) / 60% cndmask+cmp+mov+int (24:12:14:10)",
                                    "mix v_cmp_lt_f64 + 2 v_cndmask_b32 (one f64 select)"};

// one instruction of class OP on accumulator pair (a: f64 pair, x: b32)
#define ONE(OP, a, x)                                                                                         \
  if constexpr (OP == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));           \
  else if constexpr (OP == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));                  \
  else if constexpr (OP == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(c));                  \
  else if constexpr (OP == MINMAX_F64) asm volatile("v_min_f64 %0, %0, %1" : "+v"(a) : "v"(b));               \
  else if constexpr (OP == CMP_F64) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");      \
  else if constexpr (OP == CNDMASK_B32) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y));     \
  else if constexpr (OP == MOV_B32) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(y));                      \
  else if constexpr (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));                  \
  else if constexpr (OP == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(yf), "v"(zf));    \
  else if constexpr (OP == RCP_F64) asm volatile("v_rcp_f64 %0, %0" : "+v"(a));                               \
  else if constexpr (OP == RSQ_F64) asm volatile("v_rsq_f64 %0, %0" : "+v"(a));

#define ROW16(OP)                                                                                     \
  ONE(OP, a0, x0) ONE(OP, a1, x1) ONE(OP, a2, x2) ONE(OP, a3, x3) ONE(OP, a4, x4) ONE(OP, a5, x5)     \
  ONE(OP, a6, x6) ONE(OP, a7, x7) ONE(OP, a8, x8) ONE(OP, a9, x9) ONE(OP, a10, x10) ONE(OP, a11, x11) \
  ONE(OP, a12, x12) ONE(OP, a13, x13) ONE(OP, a14, x14) ONE(OP, a15, x15)

constexpr int kPerTrip = 64;      // plain classes: 4 rows of 16
constexpr int kMixTrip = 100;     // MIX_TRACE: 100 instructions per trip
constexpr int kSelTrip = 48;      // MIX_CMPSEL: 16 x (cmp + 2 cndmask)

template <int OP, int WAVES>
__global__ __launch_bounds__(256, WAVES > 8 ? 8 : WAVES) void stream(int trips, unsigned long long* cyc, double* sink) {
  double b = 1.0 + 1e-9 * threadIdx.x, c = 1e-12 * (threadIdx.x + 1);
  unsigned y = threadIdx.x | 1u;
  float yf = 1.0f + 1e-6f * threadIdx.x, zf = 1e-7f;
  double a0 = b, a1 = b + 1, a2 = b + 2, a3 = b + 3, a4 = b + 4, a5 = b + 5, a6 = b + 6, a7 = b + 7, a8 = b + 8, a9 = b + 9,
         a10 = b + 10, a11 = b + 11, a12 = b + 12, a13 = b + 13, a14 = b + 14, a15 = b + 15;
  unsigned x0 = y, x1 = y + 1, x2 = y + 2, x3 = y + 3, x4 = y + 4, x5 = y + 5, x6 = y + 6, x7 = y + 7, x8 = y + 8, x9 = y + 9,
           x10 = y + 10, x11 = y + 11, x12 = y + 12, x13 = y + 13, x14 = y + 14, x15 = y + 15;
  asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a0), "v"(a1) : "vcc");
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < trips; ++i) {
    if constexpr (OP == MIX_TRACE) {
      // 38 f64 arithmetic + 2 transcendental : 24 cndmask : 12 cmp_f64 : 14 mov : 10 int  (= 100)
      ROW16(FMA_F64) ROW16(CNDMASK_B32)
      ONE(FMA_F64, a0, x0) ONE(FMA_F64, a1, x1) ONE(FMA_F64, a2, x2) ONE(FMA_F64, a3, x3) ONE(FMA_F64, a4, x4) ONE(FMA_F64, a5, x5) ONE(FMA_F64, a6, x6)
      ONE(MUL_F64, a7, x7) ONE(MUL_F64, a8, x8) ONE(MUL_F64, a9, x9) ONE(MUL_F64, a10, x10) ONE(MUL_F64, a11, x11)
      ONE(MUL_F64, a12, x12) ONE(MUL_F64, a13, x13) ONE(MUL_F64, a14, x14) ONE(MUL_F64, a15, x15) ONE(MUL_F64, a0, x0)
      ONE(ADD_F64, a1, x1) ONE(ADD_F64, a2, x2) ONE(ADD_F64, a3, x3) ONE(ADD_F64, a4, x4) ONE(ADD_F64, a5, x5)
      ONE(RCP_F64, a6, x6) ONE(RSQ_F64, a7, x7)
      ONE(CMP_F64, a8, x8) ONE(CMP_F64, a9, x9) ONE(CMP_F64, a10, x10) ONE(CMP_F64, a11, x11) ONE(CMP_F64, a12, x12) ONE(CMP_F64, a13, x13)
      ONE(CMP_F64, a14, x14) ONE(CMP_F64, a15, x15) ONE(CMP_F64, a0, x0) ONE(CMP_F64, a1, x1) ONE(CMP_F64, a2, x2) ONE(CMP_F64, a3, x3)
      ONE(CNDMASK_B32, a4, x4) ONE(CNDMASK_B32, a5, x5) ONE(CNDMASK_B32, a6, x6) ONE(CNDMASK_B32, a7, x7)
      ONE(CNDMASK_B32, a8, x8) ONE(CNDMASK_B32, a9, x9) ONE(CNDMASK_B32, a10, x10) ONE(CNDMASK_B32, a11, x11)
      ONE(MOV_B32, a0, x0) ONE(MOV_B32, a1, x1) ONE(MOV_B32, a2, x2) ONE(MOV_B32, a3, x3) ONE(MOV_B32, a4, x4) ONE(MOV_B32, a5, x5) ONE(MOV_B32, a6, x6)
      ONE(MOV_B32, a7, x7) ONE(MOV_B32, a8, x8) ONE(MOV_B32, a9, x9) ONE(MOV_B32, a10, x10) ONE(MOV_B32, a11, x11) ONE(MOV_B32, a12, x12) ONE(MOV_B32, a13, x13)
      ONE(ADD_U32, a0, x14) ONE(ADD_U32, a1, x15) ONE(ADD_U32, a2, x0) ONE(ADD_U32, a3, x1) ONE(ADD_U32, a4, x2)
      ONE(ADD_U32, a5, x3) ONE(ADD_U32, a6, x4) ONE(ADD_U32, a7, x5) ONE(ADD_U32, a8, x6) ONE(ADD_U32, a9, x7)
    } else if constexpr (OP == MIX_CMPSEL) {
#define SEL(a, xa, xb) ONE(CMP_F64, a, xa) ONE(CNDMASK_B32, a, xa) ONE(CNDMASK_B32, a, xb)
      SEL(a0, x0, x1) SEL(a1, x2, x3) SEL(a2, x4, x5) SEL(a3, x6, x7) SEL(a4, x8, x9) SEL(a5, x10, x11) SEL(a6, x12, x13) SEL(a7, x14, x15)
      SEL(a8, x0, x1) SEL(a9, x2, x3) SEL(a10, x4, x5) SEL(a11, x6, x7) SEL(a12, x8, x9) SEL(a13, x10, x11) SEL(a14, x12, x13) SEL(a15, x14, x15)
#undef SEL
    } else {
      ROW16(OP) ROW16(OP) ROW16(OP) ROW16(OP)
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
  // keep every accumulator alive
  double s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + a8 + a9 + a10 + a11 + a12 + a13 + a14 + a15;
  unsigned u = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7 ^ x8 ^ x9 ^ x10 ^ x11 ^ x12 ^ x13 ^ x14 ^ x15;
  if (s == 12345.678 && u == 42u) sink[0] = s;
}

struct Result { std::string name; int waves; double cyc_per_inst_simd, wall_ms, wall_inst_per_ns_simd, clock_ghz; long long inst_per_wave; };

template <int OP, int WAVES>
Result run(int n_cu, unsigned long long* d_cyc, double* d_sink) {
  const int per_trip = OP == MIX_TRACE ? kMixTrip : (OP == MIX_CMPSEL ? kSelTrip : kPerTrip);
  const int trips = (OP == RCP_F64 || OP == RSQ_F64) ? 1000 : 4000;
  const int blocks = n_cu * WAVES;                 // 4 waves per block, one per SIMD: WAVES blocks per CU
  const int n_waves = blocks * 4;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {              // the second launch is the measured one
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((stream<OP, WAVES>), dim3(blocks), dim3(256), 0, 0, trips, d_cyc, d_sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
  }
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> c(n_waves);
  CHECK(hipMemcpy(c.data(), d_cyc, n_waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::sort(c.begin(), c.end());
  const double med = (double)c[n_waves / 2];
  const long long inst = (long long)trips * per_trip;
  Result r;
  r.name = kNames[OP]; r.waves = WAVES; r.inst_per_wave = inst;
  r.cyc_per_inst_simd = WAVES * med / (double)inst;
  r.wall_ms = ms;
  r.wall_inst_per_ns_simd = (double)inst * WAVES / (ms * 1e6);
  r.clock_ghz = med / (ms * 1e6);                   // shader cycles of a wave's loop per ns of wall time (<= true clock)
  return r;
}

template <int OP>
void all_waves(int n_cu, unsigned long long* d_cyc, double* d_sink, std::vector<Result>& out, bool full) {
  out.push_back(run<OP, 4>(n_cu, d_cyc, d_sink));
  if (full) {
    out.push_back(run<OP, 1>(n_cu, d_cyc, d_sink));
    out.push_back(run<OP, 2>(n_cu, d_cyc, d_sink));
    out.push_back(run<OP, 8>(n_cu, d_cyc, d_sink));
  }
}

int main(int argc, char** argv) {
  const bool full = !(argc > 1 && std::string(argv[1]) == "--w4");
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  unsigned long long* d_cyc;
  double* d_sink;
  CHECK(hipMalloc(&d_cyc, (size_t)n_cu * 8 * 4 * sizeof(unsigned long long)));
  CHECK(hipMalloc(&d_sink, 64));
  std::vector<Result> res;
  all_waves<FMA_F64>(n_cu, d_cyc, d_sink, res, full);
  all_waves<MUL_F64>(n_cu, d_cyc, d_sink, res, full);
  all_waves<ADD_F64>(n_cu, d_cyc, d_sink, res, full);
  all_waves<MINMAX_F64>(n_cu, d_cyc, d_sink, res, full);
  all_waves<CMP_F64>(n_cu, d_cyc, d_sink, res, full);
  all_waves<CNDMASK_B32>(n_cu, d_cyc, d_sink, res, full);
  all_waves<MOV_B32>(n_cu, d_cyc, d_sink, res, full);
  all_waves<ADD_U32>(n_cu, d_cyc, d_sink, res, full);
  all_waves<FMA_F32>(n_cu, d_cyc, d_sink, res, full);
  all_waves<RCP_F64>(n_cu, d_cyc, d_sink, res, full);
  all_waves<RSQ_F64>(n_cu, d_cyc, d_sink, res, full);
  all_waves<MIX_TRACE>(n_cu, d_cyc, d_sink, res, full);
  all_waves<MIX_CMPSEL>(n_cu, d_cyc, d_sink, res, full);
  printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_rate_khz\": %d, \"block\": 256,\n \"streams\": [\n", prop.name,
         prop.gcnArchName, n_cu, prop.clockRate);
  for (size_t i = 0; i < res.size(); ++i) {
    const Result& r = res[i];
    printf("  {\"stream\": \"%s\", \"waves_per_simd\": %d, \"inst_per_wave\": %lld, \"cyc_per_inst_simd\": %.4f, "
           "\"wall_ms\": %.4f, \"inst_per_ns_simd\": %.4f, \"loop_clock_ghz\": %.3f}%s\n",
           r.name.c_str(), r.waves, r.inst_per_wave, r.cyc_per_inst_simd, r.wall_ms, r.wall_inst_per_ns_simd, r.clock_ghz,
           i + 1 < res.size() ? "," : "");
  }
  printf(" ]}\n");
  return 0;
}
