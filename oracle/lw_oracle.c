/* oracle/lw_oracle.c -- TEST INFRASTRUCTURE ONLY.  Builds the plain-C restatements (RRTMG_LW + McICA:
 * lw_oracle_impl.h; RRTMG_SW: sw_oracle_impl.h; Chou-Suarez irrad / sorad: chou_oracle_impl.h, chou_sw_oracle_impl.h; GridComp data path: gridcomp_oracle_impl.h) in both precisions (see those files for the reference citations).  gcc -O2 -ffp-contract=off -shared -fPIC. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define REAL float
#define SFX(x) x##_f32
#define EXP expf
#define LOG logf
#define POW powf
#define FMOD fmodf
#define FABS fabsf
#define SQRT sqrtf
#define LOG10 log10f
#define FLOOR floorf
#include "lw_oracle_impl.h"
#include "sw_oracle_impl.h"
#include "chou_oracle_impl.h"
#include "chou_sw_oracle_impl.h"
#include "gridcomp_oracle_impl.h"
#undef LOG10
#undef FLOOR
#undef NSOLFRAC
#undef REAL
#undef SFX
#undef EXP
#undef LOG
#undef POW
#undef FMOD
#undef FABS
#undef SQRT
#undef F2
#undef F3

#define REAL double
#define SFX(x) x##_f64
#define EXP exp
#define LOG log
#define POW pow
#define FMOD fmod
#define FABS fabs
#define SQRT sqrt
#define LOG10 log10
#define FLOOR floor
#include "lw_oracle_impl.h"
#include "sw_oracle_impl.h"
#include "chou_oracle_impl.h"
#include "chou_sw_oracle_impl.h"
#include "gridcomp_oracle_impl.h"
#undef LOG10
