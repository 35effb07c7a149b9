/* ORACLE (test infrastructure). See msm.c. */
#ifndef ORACLE_MSM_H
#define ORACLE_MSM_H
#include "ge.h"
void msm_straus_ct(ge *out, const sc *scalars, const ge *points, size_t n);
void msm_vartime(ge *out, const sc *scalars, const ge *points, size_t n);
#endif
