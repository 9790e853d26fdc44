// cordic_inst_seed_narrow.hip -- instantiation unit (see cordic_inst_body.h)
#define CORDIC_INST_KIND 3
#define CORDIC_INST_NAME launch_seed_narrow
#define CORDIC_INST_CONTAINER dev::Narrow32
#define CORDIC_INST_NGEN 0
// Round 5: the dynamic-exit instance only.  This
// container is reached by CORDIC_FLAG_NO_LJ (an A/B knob) and by WW = 32 cores
// whose registers really wrap -- no core gencordic derives by itself, no
// BASELINE configuration.  What its static instances added over the dynamic
// one (profiles/r05/static_vs_dyn.txt: +13 / +41 % at 16 / 24 stages) is the
// direction tails, which those cores now do without; 1.5 MB of kernels less.
#define CORDIC_INST_DYN_ONLY
#include "cordic_inst_body.h"
