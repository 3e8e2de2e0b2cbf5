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

std::string cache_dir();
}  // namespace gfh
