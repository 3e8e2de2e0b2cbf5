// rtc.h -- run-time compilation of generated model kernels (hiprtc) with an on-disk cache.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>

namespace gfh {

// Compiles `src` for gfx950 (or loads <cache>/<hash>.hsaco).  Needs no GPU.
bool compile_to_code_object(const std::string& src, std::vector<char>* code, std::string* err, bool* from_cache);

struct ModelKernels {
  hipModule_t module = nullptr;
  hipFunction_t sweep = nullptr, sweep_gram = nullptr, chi2 = nullptr, omega = nullptr;
  hipFunction_t omega_jt = nullptr;   // optional: absent for models with integrate() and with a robust loss
  int n_active = 0;       // size of the active set the translation unit was generated for
  int omega_grid = 0;     // workgroups of gfh_k_omega resident at once (filled at its first launch)
  int kernarg_pars = 0;   // > 0: these kernels take the parameter block by value (GenConfig::kernarg_pars)
};
bool load_kernels(const std::vector<char>& code, ModelKernels* mk, std::string* err);
void unload_kernels(ModelKernels* mk);

// Code objects this PROCESS has loaded, by device and source key: a batch of small fits -- gadf_init ... gadf_fit ... gadf_close per
// spectrum, the same model every time -- finds its kernels loaded instead of reading the .hsaco and loading it again (0.3 ms of a
// 2 ms cycle).  acquire: one more user of a loaded module (false: not loaded); publish: a module just loaded, its first user;
// release: a user less -- the module of the last user stays loaded for the next context, a few idle ones per device
// (GADFIT_HIP_MODULE_CACHE=0: every context loads and unloads its own, as up to round 4).
uint64_t source_key(const std::string& src);
bool acquire_loaded(int device, uint64_t key, ModelKernels* mk);
void publish_loaded(int device, uint64_t key, const ModelKernels& mk);
void release_loaded(int device, ModelKernels* mk);

std::string cache_dir();
}  // namespace gfh
