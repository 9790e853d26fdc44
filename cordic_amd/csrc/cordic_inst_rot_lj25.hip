// cordic_inst_rot_lj25.hip -- instantiation unit (see cordic_inst_body.h):
// p2r / sp2r cores with WW = 39, left-justified by 25 bits; the dynamic-exit
// instance only
#define CORDIC_INST_KIND 1
#define CORDIC_INST_NAME launch_rot_lj25
#define CORDIC_INST_CONTAINER dev::WideLJ<25>
#define CORDIC_INST_NGEN 6
#define CORDIC_INST_DYN_ONLY
#include "cordic_inst_body.h"
