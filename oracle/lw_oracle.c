/* oracle/lw_oracle.c -- TEST INFRASTRUCTURE ONLY.  Builds the plain-C restatement in both precisions
 * (see lw_oracle_impl.h for the reference citations).  gcc -O2 -ffp-contract=off -shared -fPIC. */
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
#include "lw_oracle_impl.h"
#undef REAL
#undef SFX
#undef EXP
#undef LOG
#undef POW
#undef FMOD
#undef FABS

#define REAL double
#define SFX(x) x##_f64
#define EXP exp
#define LOG log
#define POW pow
#define FMOD fmod
#define FABS fabs
#include "lw_oracle_impl.h"
