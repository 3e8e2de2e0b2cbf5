// rtc.cpp -- hiprtc compile + cache.  The generated source (codegen.cpp) is hashed together
// with the hiprtc version and the target; the code object is kept under
// $GADFIT_HIP_CACHE, else <directory of libgadfit_hip.so>/kcache (in-tree, so a cache
// filled by build() travels with the repository snapshot to the GPU box).
#include "rtc.h"
#include <hip/hiprtc.h>
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>

namespace gfh {

static const char* kArch = "gfx950";

std::string cache_dir() {
  if (const char* e = getenv("GADFIT_HIP_CACHE")) return e;
  Dl_info info;
  if (dladdr((void*)&cache_dir, &info) && info.dli_fname) {
    std::string p = info.dli_fname;
    size_t k = p.find_last_of('/');
    if (k != std::string::npos) return p.substr(0, k) + "/kcache";
  }
  return "/tmp/gadfit_hip_kcache";
}

static uint64_t fnv1a(const std::string& s, uint64_t h = 1469598103934665603ull) {
  for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
  return h;
}

uint64_t source_key(const std::string& src) {
  static int maj = 0, min = 0;
  static std::once_flag once;
  std::call_once(once, [] { hiprtcVersion(&maj, &min); });
  return fnv1a(src + "|" + kArch + "|" + std::to_string(maj) + "." + std::to_string(min));
}

bool compile_to_code_object(const std::string& src, std::vector<char>* code, std::string* err, bool* from_cache) {
  // one compilation at a time per process: the members of a device group ask for the same source together;
  // the first compiles, the others find the code object in the cache
  static std::mutex rtc_mutex;
  std::lock_guard<std::mutex> rtc_lock(rtc_mutex);
  char key[64];
  snprintf(key, sizeof key, "%016llx", (unsigned long long)source_key(src));
  const std::string dir = cache_dir(), path = dir + "/" + key + ".hsaco";
  if (from_cache) *from_cache = false;
  {
    std::ifstream f(path, std::ios::binary);
    if (f) {
      code->assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
      if (!code->empty()) { if (from_cache) *from_cache = true; return true; }
    }
  }
  hiprtcProgram prog;
  if (hiprtcCreateProgram(&prog, src.c_str(), "gadfit_model.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
    *err = "hiprtcCreateProgram failed"; return false;
  }
  std::string archopt = std::string("--offload-arch=") + kArch;
  // -ffp-contract=on: fuse a*b+c only within one source expression; no fast-math (IEEE
  // division / sqrt / libm), so results stay within rounding of the CPU formulas.
  const char* opts[] = {archopt.c_str(), "-O3", "-ffp-contract=on", "-std=c++17"};
  hiprtcResult rc = hiprtcCompileProgram(prog, 4, opts);
  if (rc != HIPRTC_SUCCESS) {
    size_t n = 0; hiprtcGetProgramLogSize(prog, &n);
    std::string log(n, '\0'); if (n) hiprtcGetProgramLog(prog, &log[0]);
    *err = "hiprtc compile failed: " + log;
    hiprtcDestroyProgram(&prog);
    return false;
  }
  size_t n = 0; hiprtcGetCodeSize(prog, &n);
  code->resize(n); hiprtcGetCode(prog, code->data());
  hiprtcDestroyProgram(&prog);
  mkdir(dir.c_str(), 0777);
  {
    std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    std::ofstream f(tmp, std::ios::binary);
    if (f) { f.write(code->data(), (std::streamsize)code->size()); f.close(); rename(tmp.c_str(), path.c_str()); }
  }
  return true;
}

bool load_kernels(const std::vector<char>& code, ModelKernels* mk, std::string* err) {
  hipError_t e = hipModuleLoadData(&mk->module, code.data());
  if (e != hipSuccess) { *err = std::string("hipModuleLoadData: ") + hipGetErrorString(e); return false; }
  struct { const char* n; hipFunction_t* f; } fs[] = {{"gfh_k_sweep", &mk->sweep}, {"gfh_k_sweep_gram", &mk->sweep_gram}, {"gfh_k_chi2", &mk->chi2}, {"gfh_k_omega", &mk->omega}};
  for (auto& x : fs) {
    e = hipModuleGetFunction(x.f, mk->module, x.n);
    // the fused kernel of a translation unit generated without the Jacobian store has its own name
    if (e != hipSuccess && std::string(x.n) == "gfh_k_sweep_gram") {
      (void)hipGetLastError();                    // the failed lookup must not surface at a later launch check
      e = hipModuleGetFunction(x.f, mk->module, "gfh_k_sweep_gram_nostore");
    }
    // translation units for more than 80 active parameters carry no fused kernels
    if (e != hipSuccess && std::string(x.n).rfind("gfh_k_sweep_gram", 0) == 0) { *x.f = nullptr; (void)hipGetLastError(); continue; }
    if (e != hipSuccess) { *err = std::string("hipModuleGetFunction(") + x.n + "): " + hipGetErrorString(e); return false; }
  }
  if (hipModuleGetFunction(&mk->omega_jt, mk->module, "gfh_k_omega_jt") != hipSuccess) { mk->omega_jt = nullptr; (void)hipGetLastError(); }
  return true;
}

void unload_kernels(ModelKernels* mk) {
  if (mk->module) hipModuleUnload(mk->module);
  *mk = ModelKernels();
}

namespace {
struct Loaded { int device; uint64_t key; ModelKernels mk; int users; uint64_t idle_since; };
std::mutex g_loaded_mutex;
std::vector<Loaded> g_loaded;
uint64_t g_loaded_clock = 0;
constexpr int kIdleModulesPerDevice = 16;
bool module_cache_on() { static const bool on = [] { const char* e = getenv("GADFIT_HIP_MODULE_CACHE"); return !(e && atoi(e) == 0); }(); return on; }
}  // namespace

bool acquire_loaded(int device, uint64_t key, ModelKernels* mk) {
  if (!module_cache_on()) return false;
  std::lock_guard<std::mutex> lk(g_loaded_mutex);
  for (Loaded& l : g_loaded)
    if (l.device == device && l.key == key) { l.users++; *mk = l.mk; return true; }
  return false;
}

void publish_loaded(int device, uint64_t key, const ModelKernels& mk) {
  if (!module_cache_on()) return;
  std::lock_guard<std::mutex> lk(g_loaded_mutex);
  for (Loaded& l : g_loaded)
    if (l.device == device && l.key == key) return;      // (another member of a device group on the same card came first: this one stays this context's own)
  g_loaded.push_back(Loaded{device, key, mk, 1, 0});
}

void release_loaded(int device, ModelKernels* mk) {
  if (!mk->module) { *mk = ModelKernels(); return; }
  {
    std::lock_guard<std::mutex> lk(g_loaded_mutex);
    for (Loaded& l : g_loaded)
      if (l.device == device && l.mk.module == mk->module) {
        if (--l.users <= 0) {
          l.users = 0; l.idle_since = ++g_loaded_clock;
          // the oldest idle module of this device makes room (the caller has this device current)
          int idle = 0; size_t oldest = g_loaded.size();
          for (size_t i = 0; i < g_loaded.size(); i++)
            if (g_loaded[i].device == device && g_loaded[i].users == 0) { idle++; if (oldest == g_loaded.size() || g_loaded[i].idle_since < g_loaded[oldest].idle_since) oldest = i; }
          if (idle > kIdleModulesPerDevice) { hipModuleUnload(g_loaded[oldest].mk.module); g_loaded.erase(g_loaded.begin() + (long)oldest); }
        }
        *mk = ModelKernels();
        return;
      }
  }
  unload_kernels(mk);      // (not a shared module: this context's own)
}

}  // namespace gfh
