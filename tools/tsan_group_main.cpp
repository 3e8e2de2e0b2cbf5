#include "gadfit_hip.h"
#include <cstdio>
#include <vector>
int main() {
  for (int members : {2, 5, 8}) {
    std::vector<int> dev(members, -1);
    gfh_ctx* g = nullptr;
    if (gfh_create_group(members, dev.data(), &g)) { printf("create failed: %s\n", gfh_last_error(nullptr)); return 1; }
    for (int rep = 0; rep < 300; rep++) {
      const int n = 1 + (rep * 37) % 900;
      std::vector<double> bufs((size_t)members * n);
      for (size_t i = 0; i < bufs.size(); i++) bufs[i] = (double)(i % 97) * 0.25;
      std::vector<int> st(members, 0); st[rep % members] = rep % 3;
      std::vector<double> want(n, 0.0);
      for (int i = 0; i < n; i++) { double s = bufs[i]; for (int r = 1; r < members; r++) s += bufs[(size_t)r * n + i]; want[i] = s; }
      if (gfh_debug_group_allreduce(g, bufs.data(), n, st.data(), -1)) { printf("allreduce failed\n"); return 1; }
      for (int r = 0; r < members; r++) for (int i = 0; i < n; i++) if (bufs[(size_t)r * n + i] != want[i]) { printf("mismatch\n"); return 1; }
      if (rep % 50 == 49) { if (!gfh_debug_group_allreduce(g, bufs.data(), n, st.data(), members - 1)) { printf("expected failure\n"); return 1; } }
    }
    gfh_destroy(g);
  }
  printf("tsan harness ok\n");
  return 0;
}
