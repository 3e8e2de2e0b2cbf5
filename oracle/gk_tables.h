/* gk_tables.h -- oracle view of the generated Gauss-Kronrod data (include/gadfit_gk_tables.h).
 * Mirrors set_integration_rule (numerical_integration.F90:139-171). Test infrastructure. */
#ifndef ORC_GK_TABLES_H
#define ORC_GK_TABLES_H
#include "../include/gadfit_gk_tables.h"
typedef struct gk_rule { int n; const double* roots; const double* wg; const double* wk; } gk_rule;
static const gk_rule gk_rules[6] = {
  { 15, gk15_roots, gk15_wg, gk15_wk }, { 21, gk21_roots, gk21_wg, gk21_wk },
  { 31, gk31_roots, gk31_wg, gk31_wk }, { 41, gk41_roots, gk41_wg, gk41_wk },
  { 51, gk51_roots, gk51_wg, gk51_wk }, { 61, gk61_roots, gk61_wg, gk61_wk } };
static inline const gk_rule* gk_rule_by_points(int npts) {
  for (int i = 0; i < 6; i++) if (gk_rules[i].n == npts) return &gk_rules[i];
  return &gk_rules[0];
}
#endif
