// cordic_inst_rot_lj26.hip -- instantiation unit (see cordic_inst_body.h):
// p2r / sp2r cores with WW = 38, left-justified by 26 bits; the dynamic-exit
// instance only
#define CORDIC_INST_KIND 1
#define CORDIC_INST_NAME launch_rot_lj26
#define CORDIC_INST_CONTAINER dev::WideLJ<26>
#define CORDIC_INST_NGEN 5
#define CORDIC_INST_DYN_ONLY
#include "cordic_inst_body.h"
