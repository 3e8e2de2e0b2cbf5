// kernels.h -- launchers of the static (model-independent) kernels in kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gfh {
typedef long long i64;

int gram_partial_stride(int T);   // doubles per workgroup partial for T 16-row tiles

hipError_t launch_gram(hipStream_t st, int T, const double* J, i64 ldj, int na, const double* res,
                       const i64* gb_start, const int* gb_slots, int n_gb, double* partial);
hipError_t launch_reduce_partials(hipStream_t st, const double* partial, int pstride, int width,
                                  const int* ds_first_gb, int nd, double* out);
hipError_t launch_assemble(hipStream_t st, const double* G, int gw, int T, int nd, int dim, const int* inv, const int* owner, double* packed);
hipError_t launch_gather_sum(hipStream_t st, const double* G, const int* meta, const int* list, int n, double* out, const int* status,
                             double* host_out, unsigned* counter, unsigned long long* host_flag, unsigned long long seq);
hipError_t launch_assemble_sparse(hipStream_t st, const double* G, int gw, int T, int nd, int dim, const int* inv, const int* owner,
                                  const int* nz_row, const int* nz_col, int nnz, double* packed);
hipError_t launch_jtv(hipStream_t st, const double* J, i64 ldj, int na, const double* v, const i64* gb_start,
                      const int* gb_slots, int n_gb, double* partial, int pstride);
hipError_t launch_jtv_finish(hipStream_t st, const double* partial, int pstride, int na, const int* ds_first_gb, int nd, int dim,
                             const int* inv, double* out, const int* status, double* host_out, unsigned long long* host_flag,
                             unsigned long long seq);
hipError_t launch_assemble_vec(hipStream_t st, const double* V, int width, int nd, int dim, const int* inv, double* out);
hipError_t launch_cosphi(hipStream_t st, const double* J, i64 ldj, int na, const double* res, const double* dl,
                         const i64* gb_start, const int* gb_slots, const int* gb_ds, int n_gb, double* partial, int pstride);
hipError_t launch_publish(hipStream_t st, const double* src, int n, const int* status, double* host_out, unsigned* counter,
                          unsigned long long* host_flag, unsigned long long seq);
hipError_t launch_status_slot(hipStream_t st, const int* status, double* dst);
hipError_t launch_keep_warm(hipStream_t st, const double* x, i64 n, int rounds, double* sink);
hipError_t launch_fill_pads(hipStream_t st, int nd, const i64* seg, double* x, double* y, double* w, unsigned char* is_pad);
hipError_t launch_init_weights(hipStream_t st, int type, i64 n, const double* y, double* w, const unsigned char* is_pad);
}  // namespace gfh
