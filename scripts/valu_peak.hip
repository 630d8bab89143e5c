// valu_peak.hip -- calibration of the vector-ALU issue peak of gfx950 for the instruction classes the
// ray-tracing kernels are made of ("make the roofline a bound").
//
//   hipcc --offload-arch=gfx950 -O3 -o scripts/valu_peak scripts/valu_peak.hip
//   ./scripts/valu_peak [--w4]               -> one JSON object on stdout
//   rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- ./scripts/valu_peak --w4
//
// Every kernel is a stream of independent instructions of one class (16 accumulators, written in inline
// assembly so the compiler cannot fold or re-class them), launched the way the tracing kernels run:
// 256-thread blocks, exactly W waves per SIMD resident on every CU (W blocks per CU, pinned by giving
// each block 1/W of the CU's 160 KB of LDS; W = 4: the flat kernels; also 1, 2, 8).  Every wave stamps
// s_memtime (shader clock) and s_memrealtime (100 MHz) around its loop.  Reported per stream:
//   cyc_per_inst_simd = W x (median wave shader cycles) / instructions per wave: cycles one SIMD spends
//                       per wave64 instruction with W waves interleaved (2 = full rate of the SIMD-32,
//                       4 = full rate of the 16 f64 lanes);
//   clock_ghz         = median of d(s_memtime) / d(s_memrealtime) x 0.1: the clock the chip holds under
//                       this load (MI355X_MICROARCH.md, DVFS give-back (6));
//   wall_ms           = hipEvents around the launch.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum Op { FMA_F64, MUL_F64, ADD_F64, MINMAX_F64, CMP_F64, CNDMASK_B32, MOV_B32, ADD_U32, FMA_F32, RCP_F64, RSQ_F64,
          MIX_TRACE, MIX_CMPSEL, CND_NODEP, CND_SGPR, CND_MOV, CND_FMA, CMP_SGPR, SEL_C, SEL_MINMAX, CMP_F32, CND_ONCE4,
          MOV_B64, SUB_F32, MUL_F32, MIN_F32, MAX3_F32, CVT_F32_F64, CVT_F64_F32, MAD_U64_U32, MUL_LO_U32, MUL_HI_U32, XOR_B32,
          PK_FMA_F32, N_OPS };
static const char* kNames[N_OPS] = {
    "v_fma_f64", "v_mul_f64", "v_add_f64", "v_min_f64", "v_cmp_lt_f64 vcc", "v_cndmask_b32 x, x, y, vcc", "v_mov_b32", "v_add_u32",
    "v_fma_f32", "v_rcp_f64", "v_rsq_f64",
    "mix: 38 f64 fma/mul/add + 2 rcp/rsq : 24 cndmask : 12 cmp_f64 : 14 mov : 10 int (per 100)",
    "v_cmp_lt_f64 vcc + 2 v_cndmask_b32 (one f64 select), repeated",
    "v_cndmask_b32 x, y, z, vcc (destination not a source)", "v_cndmask_b32_e64 x, x, y, s[mask] (mask in an SGPR pair)",
    "v_cndmask_b32 alternating with v_mov_b32", "v_cndmask_b32 alternating with v_fma_f64",
    "v_cmp_lt_f64_e64 s[pair] (SGPR destination)", "C: a = (a < b) ? a : c on f64 (as hipcc compiles it)",
    "C: a = fmin(a, b) on f64 (as hipcc compiles it)", "v_cmp_lt_f32 vcc", "1 v_cndmask_b32 + 3 v_add_u32",
    "v_mov_b64", "v_sub_f32", "v_mul_f32", "v_min_f32", "v_max3_f32", "v_cvt_f32_f64", "v_cvt_f64_f32", "v_mad_u64_u32",
    "v_mul_lo_u32", "v_mul_hi_u32", "v_xor_b32", "v_pk_fma_f32"};
static const int kPerTrip[N_OPS] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 100, 48, 64, 64, 128, 128, 64, 64, 64, 64, 64, 64,
                                    64, 64, 64, 64, 64, 64, 64, 64, 64, 64, 64};

#define ONE(OP, a, x)                                                                                            \
  if constexpr (OP == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));              \
  else if constexpr (OP == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));                     \
  else if constexpr (OP == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(c));                     \
  else if constexpr (OP == MINMAX_F64) asm volatile("v_min_f64 %0, %0, %1" : "+v"(a) : "v"(b));                  \
  else if constexpr (OP == CMP_F64) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");         \
  else if constexpr (OP == CNDMASK_B32) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y));        \
  else if constexpr (OP == MOV_B32) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(y));                         \
  else if constexpr (OP == MOV_B64) asm volatile("v_mov_b64 %0, %1" : "=v"(a) : "v"(b));                         \
  else if constexpr (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));                     \
  else if constexpr (OP == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(yf), "v"(zf));       \
  else if constexpr (OP == RCP_F64) asm volatile("v_rcp_f64 %0, %0" : "+v"(a));                                  \
  else if constexpr (OP == RSQ_F64) asm volatile("v_rsq_f64 %0, %0" : "+v"(a));                                  \
  else if constexpr (OP == CND_NODEP) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x) : "v"(y), "v"(y2)); \
  else if constexpr (OP == CND_SGPR) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "s"(smask)); \
  else if constexpr (OP == CMP_SGPR) asm volatile("v_cmp_lt_f64_e64 %0, %1, %2" : "=s"(sdst) : "v"(a), "v"(b));  \
  else if constexpr (OP == SEL_C) a = (a < b) ? a : c;                                                            \
  else if constexpr (OP == SEL_MINMAX) a = fmin(a, b);                                                            \
  else if constexpr (OP == CMP_F32) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x), "v"(yf) : "vcc");      \
  else if constexpr (OP == SUB_F32) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(yf));                    \
  else if constexpr (OP == MUL_F32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(yf));                    \
  else if constexpr (OP == MIN_F32) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x) : "v"(yf));                    \
  else if constexpr (OP == MAX3_F32) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(yf), "v"(zf));     \
  else if constexpr (OP == CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(x) : "v"(a));                 \
  else if constexpr (OP == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a) : "v"(x));                 \
  else if constexpr (OP == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y) : "vcc"); \
  else if constexpr (OP == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(y));               \
  else if constexpr (OP == MUL_HI_U32) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(y));               \
  else if constexpr (OP == XOR_B32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(y));                     \
  else if constexpr (OP == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));

#define ROW16(OP)                                                                                     \
  ONE(OP, a0, x0) ONE(OP, a1, x1) ONE(OP, a2, x2) ONE(OP, a3, x3) ONE(OP, a4, x4) ONE(OP, a5, x5)     \
  ONE(OP, a6, x6) ONE(OP, a7, x7) ONE(OP, a8, x8) ONE(OP, a9, x9) ONE(OP, a10, x10) ONE(OP, a11, x11) \
  ONE(OP, a12, x12) ONE(OP, a13, x13) ONE(OP, a14, x14) ONE(OP, a15, x15)
// alternating pairs
#define PAIR(P, Q, a, x) ONE(P, a, x) ONE(Q, a, x)
#define ROW16P(P, Q)                                                                                              \
  PAIR(P, Q, a0, x0) PAIR(P, Q, a1, x1) PAIR(P, Q, a2, x2) PAIR(P, Q, a3, x3) PAIR(P, Q, a4, x4) PAIR(P, Q, a5, x5)     \
  PAIR(P, Q, a6, x6) PAIR(P, Q, a7, x7) PAIR(P, Q, a8, x8) PAIR(P, Q, a9, x9) PAIR(P, Q, a10, x10) PAIR(P, Q, a11, x11) \
  PAIR(P, Q, a12, x12) PAIR(P, Q, a13, x13) PAIR(P, Q, a14, x14) PAIR(P, Q, a15, x15)

template <int OP>
__global__ __launch_bounds__(256) void stream(int trips, unsigned long long* cyc, double* sink) {
  extern __shared__ int pin[];                      // occupancy pin only
  double b = 1.0 + 1e-9 * threadIdx.x, c = 1e-12 * (threadIdx.x + 1);
  unsigned y = threadIdx.x | 1u, y2 = threadIdx.x + 77u;
  float yf = 1.0f + 1e-6f * threadIdx.x, zf = 1e-7f;
  unsigned long long smask = 0x5555555555555555ull ^ (unsigned long long)blockIdx.x, sdst = 0;
  double a0 = b, a1 = b + 1, a2 = b + 2, a3 = b + 3, a4 = b + 4, a5 = b + 5, a6 = b + 6, a7 = b + 7, a8 = b + 8, a9 = b + 9,
         a10 = b + 10, a11 = b + 11, a12 = b + 12, a13 = b + 13, a14 = b + 14, a15 = b + 15;
  unsigned x0 = y, x1 = y + 1, x2 = y + 2, x3 = y + 3, x4 = y + 4, x5 = y + 5, x6 = y + 6, x7 = y + 7, x8 = y + 8, x9 = y + 9,
           x10 = y + 10, x11 = y + 11, x12 = y + 12, x13 = y + 13, x14 = y + 14, x15 = y + 15;
  if (trips < 0) pin[threadIdx.x] = trips;           // (never: keeps the LDS allocation)
  // a mask with both kinds of lanes
  asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(threadIdx.x & 1u), "v"(1u) : "vcc");
  __syncthreads();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < trips; ++i) {
    if constexpr (OP == MIX_TRACE) {
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
    } else if constexpr (OP == CND_MOV) {
      // (the mov writes a scratch register of its own)
      unsigned m0;
#define CM(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y)); asm volatile("v_mov_b32 %0, %1" : "=v"(m0) : "v"(y2));
      for (int k = 0; k < 4; ++k) { CM(x0) CM(x1) CM(x2) CM(x3) CM(x4) CM(x5) CM(x6) CM(x7) CM(x8) CM(x9) CM(x10) CM(x11) CM(x12) CM(x13) CM(x14) CM(x15) }
#undef CM
    } else if constexpr (OP == CND_FMA) {
      for (int k = 0; k < 4; ++k) { ROW16P(CNDMASK_B32, FMA_F64) }
    } else if constexpr (OP == CND_ONCE4) {
#define C4(xa, xb, xc, xd) ONE(CNDMASK_B32, a0, xa) ONE(ADD_U32, a0, xb) ONE(ADD_U32, a0, xc) ONE(ADD_U32, a0, xd)
      for (int k = 0; k < 4; ++k) { C4(x0, x1, x2, x3) C4(x4, x5, x6, x7) C4(x8, x9, x10, x11) C4(x12, x13, x14, x15) }
#undef C4
    } else {
      ROW16(OP) ROW16(OP) ROW16(OP) ROW16(OP)
    }
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
  if (s == 12345.678 && u == 42u && sdst == 7) sink[0] = s;
}

struct Result { int op, waves; double cyc_per_inst_simd, wall_ms, clock_ghz, spread; long long inst_per_wave; };

template <int OP>
Result run(int waves, int n_cu, unsigned long long* d_cyc, double* d_sink) {
  const int trips = (OP == RCP_F64 || OP == RSQ_F64) ? 1000 : (OP == CNDMASK_B32 || OP == CND_NODEP || OP == CND_SGPR ? 1000 : 4000);
  const int blocks = n_cu * waves;                 // 4 waves per block, one per SIMD: `waves` blocks per CU
  const int n_waves = blocks * 4;
  // exactly `waves` blocks fit a CU: each takes (160 KB / waves) - 1 KB of LDS
  const int lds = 160 * 1024 / waves - 1024;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {              // the second launch is the measured one
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((stream<OP>), dim3(blocks), dim3(256), lds, 0, trips, d_cyc, d_sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
  }
  CHECK(hipGetLastError());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> c(2 * n_waves);
  CHECK(hipMemcpy(c.data(), d_cyc, 2 * n_waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::vector<double> cy(n_waves), clk(n_waves);
  for (int i = 0; i < n_waves; ++i) { cy[i] = (double)c[2 * i]; clk[i] = (double)c[2 * i] / (double)c[2 * i + 1] * 0.1; }
  std::sort(cy.begin(), cy.end());
  std::sort(clk.begin(), clk.end());
  const long long inst = (long long)trips * kPerTrip[OP];
  Result r;
  r.op = OP; r.waves = waves; r.inst_per_wave = inst;
  r.cyc_per_inst_simd = waves * cy[n_waves / 2] / (double)inst;
  r.spread = cy[n_waves - 1] / cy[0];
  r.wall_ms = ms;
  r.clock_ghz = clk[n_waves / 2];
  return r;
}

template <int OP>
void all_waves(int n_cu, unsigned long long* d_cyc, double* d_sink, std::vector<Result>& out, bool full) {
  out.push_back(run<OP>(4, n_cu, d_cyc, d_sink));
  if (full) {
    out.push_back(run<OP>(1, n_cu, d_cyc, d_sink));
    out.push_back(run<OP>(2, n_cu, d_cyc, d_sink));
    out.push_back(run<OP>(8, n_cu, d_cyc, d_sink));
  }
}

int main(int argc, char** argv) {
  const bool full = !(argc > 1 && std::string(argv[1]) == "--w4");
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  unsigned long long* d_cyc;
  double* d_sink;
  CHECK(hipMalloc(&d_cyc, (size_t)n_cu * 8 * 4 * 2 * sizeof(unsigned long long)));
  CHECK(hipMalloc(&d_sink, 64));
  std::vector<Result> res;
  all_waves<FMA_F64>(n_cu, d_cyc, d_sink, res, full);
  all_waves<MUL_F64>(n_cu, d_cyc, d_sink, res, full);
  all_waves<ADD_F64>(n_cu, d_cyc, d_sink, res, full);
  all_waves<MINMAX_F64>(n_cu, d_cyc, d_sink, res, full);
  all_waves<CMP_F64>(n_cu, d_cyc, d_sink, res, full);
  all_waves<CMP_SGPR>(n_cu, d_cyc, d_sink, res, full);
  all_waves<CMP_F32>(n_cu, d_cyc, d_sink, res, full);
  all_waves<CNDMASK_B32>(n_cu, d_cyc, d_sink, res, full);
  all_waves<CND_NODEP>(n_cu, d_cyc, d_sink, res, full);
  all_waves<CND_SGPR>(n_cu, d_cyc, d_sink, res, full);
  all_waves<CND_MOV>(n_cu, d_cyc, d_sink, res, full);
  all_waves<CND_FMA>(n_cu, d_cyc, d_sink, res, full);
  all_waves<CND_ONCE4>(n_cu, d_cyc, d_sink, res, full);
  all_waves<MOV_B32>(n_cu, d_cyc, d_sink, res, full);
  all_waves<MOV_B64>(n_cu, d_cyc, d_sink, res, full);
  all_waves<SUB_F32>(n_cu, d_cyc, d_sink, res, full);
  all_waves<MUL_F32>(n_cu, d_cyc, d_sink, res, full);
  all_waves<MIN_F32>(n_cu, d_cyc, d_sink, res, full);
  all_waves<MAX3_F32>(n_cu, d_cyc, d_sink, res, full);
  all_waves<CVT_F32_F64>(n_cu, d_cyc, d_sink, res, full);
  all_waves<CVT_F64_F32>(n_cu, d_cyc, d_sink, res, full);
  all_waves<MAD_U64_U32>(n_cu, d_cyc, d_sink, res, full);
  all_waves<MUL_LO_U32>(n_cu, d_cyc, d_sink, res, full);
  all_waves<MUL_HI_U32>(n_cu, d_cyc, d_sink, res, full);
  all_waves<XOR_B32>(n_cu, d_cyc, d_sink, res, full);
  all_waves<PK_FMA_F32>(n_cu, d_cyc, d_sink, res, full);
  all_waves<ADD_U32>(n_cu, d_cyc, d_sink, res, full);
  all_waves<FMA_F32>(n_cu, d_cyc, d_sink, res, full);
  all_waves<RCP_F64>(n_cu, d_cyc, d_sink, res, full);
  all_waves<RSQ_F64>(n_cu, d_cyc, d_sink, res, full);
  all_waves<SEL_C>(n_cu, d_cyc, d_sink, res, full);
  all_waves<SEL_MINMAX>(n_cu, d_cyc, d_sink, res, full);
  all_waves<MIX_CMPSEL>(n_cu, d_cyc, d_sink, res, full);
  all_waves<MIX_TRACE>(n_cu, d_cyc, d_sink, res, full);
  printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_rate_khz\": %d, \"block\": 256,\n \"streams\": [\n", prop.name,
         prop.gcnArchName, n_cu, prop.clockRate);
  for (size_t i = 0; i < res.size(); ++i) {
    const Result& r = res[i];
    printf("  {\"id\": %d, \"stream\": \"%s\", \"waves_per_simd\": %d, \"inst_per_wave\": %lld, \"cyc_per_inst_simd\": %.4f, "
           "\"clock_ghz\": %.3f, \"wall_ms\": %.4f, \"slowest_over_fastest_wave\": %.3f}%s\n",
           r.op, kNames[r.op], r.waves, r.inst_per_wave, r.cyc_per_inst_simd, r.clock_ghz, r.wall_ms, r.spread,
           i + 1 < res.size() ? "," : "");
  }
  printf(" ]}\n");
  return 0;
}
