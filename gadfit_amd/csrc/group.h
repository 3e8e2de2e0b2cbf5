// group.h -- single-process device group: one context ("image") per GPU, each driven by its own host
// thread, for programs started without a launcher (a plain Fortran executable on an 8-GPU node).
// The reference runs the same code on every coarray image and sums with co_sum (misc.F90:133-170);
// here the images are threads of one process and the sum is either the host-side ordered sum of the
// members' pinned result mailboxes (default: the consumer of J^T J / J^T r / chi2 is the host solve
// anyway) or RCCL (ncclCommInitAll, GADFIT_HIP_GROUP_REDUCE=rccl).
#pragma once
#include <cstddef>
#include <functional>

struct gfh_ctx;

namespace gfh {

struct Group;

// n_devices <= 0: every visible device.  devices == nullptr: 0 .. n_devices-1.
int group_create(int n_devices, const int* devices, gfh_ctx** handle);
void group_destroy(gfh_ctx* handle);
int group_size(const gfh_ctx* handle);
gfh_ctx* group_member(const gfh_ctx* handle, int r);
// fn(member, rank) on every member concurrently (each on its own thread); 0 when all succeeded,
// otherwise the handle carries the first failing member's message
int group_run(gfh_ctx* handle, const std::function<int(gfh_ctx*, int)>& fn);
// called by a member from inside group_run: buf[0..n) <- sum over the members in rank order (bitwise
// the same on every member), *status <- max over the members
int group_allreduce(gfh_ctx* member, double* buf, size_t n, int* status);

}  // namespace gfh
