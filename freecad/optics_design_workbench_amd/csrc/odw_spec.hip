// odw_spec.hip -- scene-compiled flat kernels, host side.
//
// The reference prepares a scene once per run and reuses the result for every ray
// (simulation/raytracing_cache.py:92-111: cachedShape / cachedFaces / cachedBoundBox ..., cleared by
// cacheClear :36).  The counterpart here goes one step further: for a scene the flat kernel would
// trace (<= 64 analytic primitives), the library writes the scene's STRUCTURE -- primitive types,
// groups, flags, trimming lists, zero / +-1 pattern of every frame, optical type and recording switch
// of every group -- as a C++ header of compile-time constants, compiles odw_kernels.hip's ray loop
// against it with hiprtc (0.5 - 2 s), and launches that kernel instead of the generic one: the
// primitive loop is unrolled, type dispatch and face masks fold away, frame products skip their zero
// terms, table reads have constant offsets (odw_kernels.hip: SPEC).  All float64 VALUES (frames,
// parameters, boxes, optical constants) are still read from the uploaded tables, so one kernel serves
// every scene of the same structure: a parameter sweep compiles once.  (Values as literals of the
// kernel were tried as a second level: 13.1 against 12.2 ms per 1e8 C3 rays -- the constants crowd the
// scalar registers, 56 spilled -- and rounding no longer matched the generic kernel; not kept.)
// Kernels are cached per process (key = the header text) and on disk (ODW_KERNEL_CACHE, default
// ~/.cache/odw_trace).  Results: the generic kernel's, bit for bit (same arithmetic in the same
// order; both are compiled with -ffp-contract=on).
// hiprtc is loaded on first use (dlopen); the kernel sources are embedded in this library.
#include <dlfcn.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

#if !defined(__HIP_DEVICE_COMPILE__)
// the sources the run-time compiler needs, byte for byte as this library was built from
__asm__(
    ".pushsection .rodata\n"
    ".global odw_src_kernels\nodw_src_kernels:\n.incbin \"odw_kernels.hip\"\n.byte 0\n"
    ".global odw_src_device\nodw_src_device:\n.incbin \"odw_device.h\"\n.byte 0\n"
    ".global odw_src_trace\nodw_src_trace:\n.incbin \"../../../include/odw_trace.h\"\n.byte 0\n"
    ".popsection\n");
#endif
extern "C" const char odw_src_kernels[], odw_src_device[], odw_src_trace[];

namespace {

// primitives a compiled kernel is unrolled over at most (ODW_SPEC_MAX_PRIMS: experiments).  A lens train of 61
// primitives compiles in 17 s to 212 KB of code and still runs 2.6 x the generic flat kernel's rate (19
// primitives: 4.6 s, 91 KB, 2.1 x; scripts/bench_lens_train.py).
// (a plain function: a second initialiser lambda of this shape in the same translation unit ran the first
//  one's body -- kBvhLeaf's, odw_capi.hip -- with this compiler)
int spec_max_prims() {
  const char* e = getenv("ODW_SPEC_MAX_PRIMS");
  const int v = e ? atoi(e) : 0;
  return v > 0 ? v : 64;
}
const int kSpecMaxPrims = spec_max_prims();

struct Hiprtc {
  void* lib = nullptr;
  decltype(&hiprtcCreateProgram) create = nullptr;
  decltype(&hiprtcCompileProgram) compile = nullptr;
  decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
  decltype(&hiprtcGetProgramLog) log = nullptr;
  decltype(&hiprtcGetCodeSize) code_size = nullptr;
  decltype(&hiprtcGetCode) code = nullptr;
  decltype(&hiprtcDestroyProgram) destroy = nullptr;
  decltype(&hiprtcVersion) version = nullptr;
  std::string error;
  std::string version_text;       // "major.minor" of the loaded hiprtc (part of the disk-cache key)
};

Hiprtc& hiprtc() {
  static Hiprtc h;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
      h.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (h.lib) break;
    }
    if (!h.lib) { h.error = std::string("hiprtc not available: ") + dlerror(); return; }
#define ODW_RTC_SYM(field, sym)                                             \
    h.field = reinterpret_cast<decltype(h.field)>(dlsym(h.lib, #sym));      \
    if (!h.field) h.error = "hiprtc: missing symbol " #sym;
    ODW_RTC_SYM(create, hiprtcCreateProgram)
    ODW_RTC_SYM(compile, hiprtcCompileProgram)
    ODW_RTC_SYM(log_size, hiprtcGetProgramLogSize)
    ODW_RTC_SYM(log, hiprtcGetProgramLog)
    ODW_RTC_SYM(code_size, hiprtcGetCodeSize)
    ODW_RTC_SYM(code, hiprtcGetCode)
    ODW_RTC_SYM(destroy, hiprtcDestroyProgram)
    ODW_RTC_SYM(version, hiprtcVersion)
#undef ODW_RTC_SYM
    if (h.version) {
      int major = 0, minor = 0;
      if (h.version(&major, &minor) == HIPRTC_SUCCESS) h.version_text = std::to_string(major) + "." + std::to_string(minor);
    }
  });
  return h;
}

// ---- the header ------------------------------------------------------------------------------
template <class T, class F>
std::string table(const char* type, const char* name, int n, const T* v, F fmt) {
  std::string s = std::string("  static constexpr ") + type + " " + name + "(int i) { constexpr " + type + " T[] = {";
  for (int i = 0; i < std::max(1, n); ++i) s += (i ? ", " : "") + (i < n ? fmt(v[i]) : std::string("0"));
  return s + "}; return T[i]; }\n";
}

// why the scene cannot be compiled, or empty
std::string spec_ineligible(const odw_ctx* ctx) {
  const int n = ctx->P.scene.n_prims;
  if (n < 1) return "no primitives";
  if (n > kSpecMaxPrims) return "more primitives (" + std::to_string(n) + ") than a compiled kernel takes (" + std::to_string(kSpecMaxPrims) + ")";
  for (int p = 0; p < n; ++p) {
    const int t = ctx->h_prim_i32[4 * p];
    if (t == ODW_PRIM_TRIANGLE) return "facets belong to the BVH kernels";
  }
  if (ctx->h_prim_hdr.size() < (size_t)n * 8 || ctx->h_dead.size() < (size_t)n) return "boxes not built";
  return "";
}

// `struct Spec` of the uploaded scene (tables of scene_host_tables / compute_boxes).  The text is the
// cache key: equal text = equal kernel.
std::string spec_text(const odw_ctx* ctx) {
  const int n = ctx->P.scene.n_prims, ng = ctx->P.scene.n_groups;
  std::vector<int> type(n), group(n), flags(n), condw(n), dead(n);
  std::vector<unsigned long long> xf(n);
  for (int p = 0; p < n; ++p) {
    type[p] = ctx->h_prim_i32[4 * p];
    group[p] = ctx->h_prim_i32[4 * p + 1];
    flags[p] = ctx->h_prim_i32[4 * p + 2] & ~ODW_FLAG_ISOLATED;
    condw[p] = ctx->h_prim_i32[4 * p + 3];
    const int facemask = (flags[p] >> ODW_FACEMASK_SHIFT) & 0xff;
    // (an empty box is a matter of values: such a primitive stays, its box culls it)
    dead[p] = facemask == 0;
    const double* m = &ctx->h_prim_f64[16 * (size_t)p];
    unsigned long long w = 0;
    for (int i = 0; i < 12; ++i) {
      if (m[i] != 0.0) w |= 1ull << i;
      if (i % 4 != 3 && m[i] == 1.0) w |= 1ull << (12 + i);
      if (i % 4 != 3 && m[i] == -1.0) w |= 1ull << (24 + i);
    }
    xf[p] = w;
  }
  // primitives with the same box: equal sets {p} + {q : p must lie inside q} (compute_boxes cuts p's box by
  // the boxes of those q).  box_of = the first such primitive, box_shared = another one refers to it.
  std::vector<int> box_of(n), box_shared(n, 0);
  {
    std::vector<std::vector<int>> inside(n);
    for (int p = 0; p < n; ++p) {
      inside[p].push_back(p);
      const int off = condw[p] & 0xffffff, cnt = (condw[p] >> 24) & 0xff;
      for (int c = off; c < off + cnt && c < (int)ctx->h_cond.size(); ++c)
        if (ctx->h_cond[c] < 0) inside[p].push_back(ctx->h_cond[c] & 0x7fffffff);
      std::sort(inside[p].begin(), inside[p].end());
      inside[p].erase(std::unique(inside[p].begin(), inside[p].end()), inside[p].end());
    }
    for (int p = 0; p < n; ++p) {
      box_of[p] = p;
      // (the sets are equal, but compute_boxes cuts with the operands' FULL boxes only: p's box is
      //  box(p) ^ box(q1) ^ ..., the same expression for both when the sets agree)
      for (int q = 0; q < p; ++q)
        if (!dead[q] && !dead[p] && group[q] == group[p] && inside[q] == inside[p]) { box_of[p] = q; box_shared[q] = 1; break; }
    }
  }
  std::vector<int> gtype(std::max(1, ng)), grec(std::max(1, ng));
  for (int g = 0; g < ng; ++g) { gtype[g] = ctx->h_group_i32[4 * g]; grec[g] = ctx->h_group_i32[4 * g + 1]; }
  auto fi = [](int v) { return std::to_string(v); };
  auto fu = [](unsigned long long v) { char b[32]; snprintf(b, sizeof b, "0x%llxull", v); return std::string(b); };
  std::string s;
  s += "struct Spec {\n  static constexpr bool enabled = true;\n";
  s += "  static constexpr int N = " + std::to_string(n) + ";\n";
  s += table("int", "type", n, type.data(), fi) + table("int", "group", n, group.data(), fi) +
       table("int", "flags", n, flags.data(), fi) + table("int", "cond_word", n, condw.data(), fi) +
       table("bool", "dead", n, dead.data(), fi) + table("int", "box_of", n, box_of.data(), fi) +
       table("bool", "box_shared", n, box_shared.data(), fi) +
       table("int", "cond", (int)ctx->h_cond.size(), ctx->h_cond.data(), fi) +
       table("unsigned long long", "xf", n, xf.data(), fu) + table("int", "gtype", ng, gtype.data(), fi) +
       table("bool", "record", ng, grec.data(), fi);
  // (paraboloids: the generic flat kernel leaves their code out -- it costs every scene 1 % -- and hands such
  //  documents to the grid kernel; a compiled kernel carries it exactly when the scene has one)
  bool parab = false;
  for (int p = 0; p < n; ++p) parab |= type[p] == ODW_PRIM_PARABOLOID;
  s += std::string("  static constexpr bool parab() { return ") + (parab ? "true" : "false") + "; }\n";
  // ODW_FLAG_ISOLATED (a ray that has entered an isolated solid tests that solid only) is left to the generic
  // flat kernel, which gains 3 % on lensesAndMirrors from it: in a compiled kernel a skipped box test saves less
  // than the extra predicate on every primitive costs (11.85 against 11.30 ms, 8 spilled SGPRs).  The rule
  // never changes a result, so the two kernels still produce the same rows.
  s += "  static constexpr bool isolated() { return false; }\n";
  s += "  static constexpr int cond_off(int i) { return cond_word(i) & 0xffffff; }\n"
       "  static constexpr int cond_cnt(int i) { return (cond_word(i) >> 24) & 0xff; }\n";
  s += "  static constexpr unsigned long long umask() { return " + fu(ctx->P.scene.all_mask & ~ctx->P.scene.ignore_mask) + "; }\n";
  s += std::string("  static constexpr bool seq() { return ") + (ctx->P.scene.seq_enabled ? "true" : "false") + "; }\n";
  s += "};\n";
  s += std::string("#define ODW_SPEC_LEAN ") + (ctx->lean ? "true" : "false") + "\n";
  // stochastic surfaces: the kernel variant with scatter() (the sampler tables themselves are run-time data)
  s += std::string("#define ODW_SPEC_STOCH ") + (ctx->n_samplers > 0 ? "true" : "false") + "\n";
  return s;
}

// ---- compile ---------------------------------------------------------------------------------
uint64_t fnv1a(const std::string& s, uint64_t h = 1469598103934665603ull) {
  for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
  return h;
}

std::string cache_dir() {
  const char* e = getenv("ODW_KERNEL_CACHE");
  if (e) return *e ? e : "";                      // empty: no disk cache
  const char* x = getenv("XDG_CACHE_HOME");
  if (x && *x) return std::string(x) + "/odw_trace";
  const char* h = getenv("HOME");
  return h && *h ? std::string(h) + "/.cache/odw_trace" : "";
}

void mkdirs(const std::string& path) {
  for (size_t i = 1; i <= path.size(); ++i)
    if (i == path.size() || path[i] == '/') (void)mkdir(path.substr(0, i).c_str(), 0755);
}

// header text -> code object for `arch`; error text in `err`
bool spec_compile(const std::string& text, const std::string& arch, std::vector<char>& code, std::string& err) {
  Hiprtc& rtc = hiprtc();
  if (!rtc.error.empty()) { err = rtc.error; return false; }
  const char* headers[] = {odw_src_kernels, odw_src_device, odw_src_trace, text.c_str()};
  const char* names[] = {"odw_kernels.hip", "odw_device.h", "odw_trace.h", "odw_spec.h"};
  hiprtcProgram prog = nullptr;
  if (rtc.create(&prog, "#include \"odw_kernels.hip\"\n", "odw_spec_kernel.hip", 4, headers, names) != HIPRTC_SUCCESS) {
    err = "hiprtcCreateProgram failed";
    return false;
  }
  const std::string a = "--offload-arch=" + arch;
  std::vector<const char*> opts = {a.c_str(), "-std=c++17", "-O3", "-ffp-contract=on", "-DODW_SPEC_HEADER=\"odw_spec.h\""};
  // experiments: ODW_SPEC_OPTS = further compiler options, separated by blanks (part of the cache key)
  std::vector<std::string> extra;
  if (const char* e = getenv("ODW_SPEC_OPTS")) {
    std::string w;
    for (const char* c = e;; ++c) {
      if (*c == ' ' || *c == 0) { if (!w.empty()) extra.push_back(w); w.clear(); if (!*c) break; }
      else w += *c;
    }
  }
  for (const std::string& x : extra) opts.push_back(x.c_str());
  const hiprtcResult r = rtc.compile(prog, (int)opts.size(), opts.data());
  if (r != HIPRTC_SUCCESS) {
    size_t ls = 0;
    rtc.log_size(prog, &ls);
    std::string log(ls, '\0');
    if (ls) rtc.log(prog, &log[0]);
    err = "hiprtc: compilation of the scene kernel failed:\n" + log.substr(0, 4000);
    rtc.destroy(&prog);
    return false;
  }
  size_t cs = 0;
  rtc.code_size(prog, &cs);
  code.resize(cs);
  if (cs) rtc.code(prog, code.data());
  rtc.destroy(&prog);
  if (!cs) { err = "hiprtc returned no code"; return false; }
  return true;
}

struct SpecKernel {
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
};
// a compilation under way (ODW_COMPILE_AUTO): run by a thread of its own, picked up by the launch that finds it done
struct SpecJob {
  std::atomic<bool> done{false};
  bool ok = false;
  std::vector<char> code;
  std::string err;
  double seconds = 0;
};
// process-wide state, never destroyed (a compile thread may outlive main(): nothing of it is torn down under it)
struct SpecGlobal {
  std::mutex mu;
  std::map<std::string, SpecKernel> cache;                 // (device, arch, options, header text) -> loaded kernel
  std::map<std::string, std::shared_ptr<SpecJob>> jobs;    // (arch, options, header text) -> compilation under way / finished
  std::map<std::string, uint64_t> rays;                    // the same key -> rays traced with that structure on generic
                                                           // kernels, by all contexts of the process (many short runs
                                                           // of one project make it hot like one long run)
  std::atomic<int> running{0};
};
SpecGlobal& spec_global() {
  static SpecGlobal* g = new SpecGlobal();
  return *g;
}

std::string spec_cache_file(const std::string& arch, const std::string& opts, const std::string& text) {
  const std::string dir = cache_dir();
  if (dir.empty()) return "";
  // (the compiler is part of the key: a process that imported torch first compiles with torch's bundled hiprtc, of
  //  another ROCm release -- scripts/torch_rtc_check.py --, and code objects of two compilers must not share a file;
  //  HIP_VERSION = the hipcc that built this library and its generic kernels)
  uint64_t h = fnv1a(arch + "|" + opts + "|" + text);
  h = fnv1a("|rtc " + hiprtc().version_text + "|hip " + std::to_string((long)HIP_VERSION), h);
  h = fnv1a(odw_src_kernels, h);
  h = fnv1a(odw_src_device, h);
  h = fnv1a(odw_src_trace, h);
  char name[40];
  snprintf(name, sizeof name, "/%016llx.hsaco", (unsigned long long)h);
  return dir + name;
}

// code object of the header: from the disk cache, or compiled now (and put there); hit = 2 if it came from the disk
bool spec_code(const std::string& text, const std::string& arch, std::vector<char>& code, std::string& err, int& hit,
               double& seconds) {
  const char* xo = getenv("ODW_SPEC_OPTS");
  const std::string file = spec_cache_file(arch, xo ? xo : "", text);
  hit = 0;
  seconds = 0;
  if (!file.empty()) {
    if (FILE* f = fopen(file.c_str(), "rb")) {
      fseek(f, 0, SEEK_END);
      const long sz = ftell(f);
      fseek(f, 0, SEEK_SET);
      if (sz > 0) { code.resize((size_t)sz); if (fread(code.data(), 1, (size_t)sz, f) != (size_t)sz) code.clear(); }
      fclose(f);
      if (!code.empty()) { hit = 2; return true; }
    }
  }
  const auto t0 = std::chrono::steady_clock::now();
  if (!spec_compile(text, arch, code, err)) return false;
  seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (!file.empty()) {
    mkdirs(cache_dir());
    static std::atomic<unsigned> serial{0};               // (two contexts of one process may compile the same key)
    const std::string tmp = file + "." + std::to_string((long)getpid()) + "." + std::to_string(serial.fetch_add(1));
    if (FILE* f = fopen(tmp.c_str(), "wb")) {
      const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
      fclose(f);
      if (!ok || rename(tmp.c_str(), file.c_str()) != 0) (void)unlink(tmp.c_str());
    }
  }
  return true;
}

// at exit: wait for compilations under way (they use hiprtc, whose own state goes away after this handler)
void spec_wait_at_exit() {
  for (int k = 0; k < 3000 && spec_global().running.load() > 0; ++k) usleep(10000);
}

// Binds ctx->spec_fn for the uploaded scene, or leaves it null (the generic kernels run).
// ODW_COMPILE_STRUCTURE: compiles now if no cache has the kernel.
// ODW_COMPILE_AUTO: never waits -- a kernel already loaded (or on disk) is bound at once; otherwise the scene has to
// earn its compilation: once ctx->spec_hot_rays rays were traced with its structure on generic kernels (by all
// contexts of the process together), a thread compiles,
// and the first launch after it has finished binds the result (the rows are the same either way, bit for bit).
// batch: the kernel's BATCH variant (scenes of one structure side by side, odw_trace_batch) -> ctx->spec_batch_fn; it is
// bound on the first batch launch, while the single-scene kernel of the same structure is bound (its mode decides)
int spec_bind(odw_ctx* ctx, bool batch = false) {
  if (batch) {
    ctx->spec_batch_fn = nullptr;
  } else {
    ctx->spec_dirty = false;
    ctx->spec_fn = nullptr;
    ctx->spec_batch_fn = nullptr;
    ctx->spec_batch_failed = false;
    ctx->spec_seconds = 0;
    ctx->spec_cache_hit = 0;
    ctx->spec_pending = false;
  }
  if (ctx->compile_mode == ODW_COMPILE_OFF || !ctx->have_scene) return ODW_OK;
  if (!spec_ineligible(ctx).empty()) return ODW_OK;
  hipDeviceProp_t prop;
  HIPCHK(ctx, hipGetDeviceProperties(&prop, ctx->device));
  const std::string arch = prop.gcnArchName;
  const std::string text = spec_text(ctx) + (batch ? "#define ODW_SPEC_BATCH true\n" : "");
  const char* xo = getenv("ODW_SPEC_OPTS");
  const std::string jkey = arch + "|" + (xo ? xo : "") + "|" + text;
  const std::string key = std::to_string(ctx->device) + "|" + jkey;
  SpecGlobal& G = spec_global();
  std::unique_lock<std::mutex> lock(G.mu);
  auto it = G.cache.find(key);
  if (it == G.cache.end()) {
    std::vector<char> code;
    std::string err;
    if (ctx->compile_mode == ODW_COMPILE_AUTO) {
      auto jt = G.jobs.find(jkey);
      if (jt == G.jobs.end()) {
        // on disk already?  (a file read, no compilation: done here)
        const std::string file = spec_cache_file(arch, xo ? xo : "", text);
        FILE* f = file.empty() ? nullptr : fopen(file.c_str(), "rb");
        if (f) {
          fclose(f);
        } else {
          // not hot yet, or hot: start the thread
          // (the BATCH variant rides on the single-scene kernel's bookkeeping: its own key must not replace that one's)
          if (!batch) { ctx->spec_pending = true; ctx->spec_key = jkey; }
          if (G.rays[jkey] < ctx->spec_hot_rays) return ODW_OK;
          auto job = std::make_shared<SpecJob>();
          G.jobs[jkey] = job;
          static std::once_flag once;
          std::call_once(once, [] { (void)hiprtc(); atexit(spec_wait_at_exit); });
          G.running.fetch_add(1);
          std::thread([job, text, arch] {
            int hit;
            job->ok = spec_code(text, arch, job->code, job->err, hit, job->seconds);
            job->done.store(true);
            spec_global().running.fetch_sub(1);
          }).detach();
          return ODW_OK;
        }
      } else if (!jt->second->done.load()) {
        if (!batch) { ctx->spec_pending = true; ctx->spec_key = jkey; }
        return ODW_OK;                                      // still compiling: generic kernels meanwhile
      } else {
        std::shared_ptr<SpecJob> job = jt->second;
        if (!job->ok) return fail(ctx, ODW_ERR_DEVICE, job->err);   // (stays in the map: not tried again)
        code = job->code;
        ctx->spec_seconds = job->seconds;
      }
    }
    if (code.empty()) {
      lock.unlock();                                        // (a compilation of seconds: other contexts go on)
      const bool ok = spec_code(text, arch, code, err, ctx->spec_cache_hit, ctx->spec_seconds);
      lock.lock();
      if (!ok) return fail(ctx, ODW_ERR_DEVICE, err);
      it = G.cache.find(key);                               // (another context may have loaded it meanwhile)
    }
    if (it == G.cache.end()) {
      SpecKernel k;
      hipError_t le = hipModuleLoadData(&k.mod, code.data());
      if (le == hipSuccess) le = hipModuleGetFunction(&k.fn, k.mod, "odw_spec_kernel");
      if (le != hipSuccess && ctx->spec_cache_hit == 2) {
        // a file of the disk cache that does not load (truncated, damaged): drop it and compile once
        (void)hipGetLastError();
        if (k.mod) { (void)hipModuleUnload(k.mod); k.mod = nullptr; }
        const std::string file = spec_cache_file(arch, xo ? xo : "", text);
        if (!file.empty()) (void)unlink(file.c_str());
        code.clear();
        lock.unlock();
        const bool ok = spec_code(text, arch, code, err, ctx->spec_cache_hit, ctx->spec_seconds);
        lock.lock();
        if (!ok) return fail(ctx, ODW_ERR_DEVICE, err);
        le = hipModuleLoadData(&k.mod, code.data());
        if (le == hipSuccess) le = hipModuleGetFunction(&k.fn, k.mod, "odw_spec_kernel");
      }
      HIPCHK(ctx, le);
      it = G.cache.find(key);
      if (it == G.cache.end()) it = G.cache.emplace(key, k).first;
    }
  } else {
    ctx->spec_cache_hit = 1;
  }
  if (batch) {
    ctx->spec_batch_fn = it->second.fn;
    return ODW_OK;
  }
  ctx->spec_fn = it->second.fn;
  ctx->spec_lean = ctx->lean;
  ctx->spec_stoch = ctx->n_samplers > 0;
  return ODW_OK;
}

// ODW_COMPILE_AUTO, called by every launch that runs a generic flat kernel: counts the scene's rays and looks whether
// its compilation should start / has finished (then the next launch binds it)
void spec_note_launch(odw_ctx* ctx, uint64_t n_rays) {
  if (ctx->compile_mode != ODW_COMPILE_AUTO || !ctx->spec_pending) return;
  SpecGlobal& G = spec_global();
  std::lock_guard<std::mutex> lock(G.mu);
  uint64_t& rays = G.rays[ctx->spec_key];
  rays += n_rays;
  if (rays < ctx->spec_hot_rays) return;
  auto jt = G.jobs.find(ctx->spec_key);
  // hot and no compilation yet: the next launch starts it; compilation finished: the next launch binds its result
  if (jt == G.jobs.end() || jt->second->done.load()) ctx->spec_dirty = true;
}

int spec_launch(odw_ctx* ctx, unsigned grid, bool batch) {
  TraceParams P = ctx->P;
  size_t size = sizeof P;
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &P, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  HIPCHK(ctx, hipModuleLaunchKernel(batch ? ctx->spec_batch_fn : ctx->spec_fn, grid, 1, 1, 256, 1, 1, 0, ctx->stream, nullptr, config));
  return ODW_OK;
}

}  // namespace
