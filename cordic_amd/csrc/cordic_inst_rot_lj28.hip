// cordic_inst_rot_lj28.hip -- instantiation unit (see cordic_inst_body.h):
// p2r / sp2r cores with WW = 36, left-justified by 28 bits; the dynamic-exit
// instance only
#define CORDIC_INST_KIND 1
#define CORDIC_INST_NAME launch_rot_lj28
#define CORDIC_INST_CONTAINER dev::WideLJ<28>
#define CORDIC_INST_NGEN 3
#define CORDIC_INST_DYN_ONLY
#include "cordic_inst_body.h"
