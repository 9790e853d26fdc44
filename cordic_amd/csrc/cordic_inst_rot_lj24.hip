// cordic_inst_rot_lj24.hip -- instantiation unit (see cordic_inst_body.h):
// p2r / sp2r cores with WW = 40, left-justified by 24 bits; the dynamic-exit
// instance only
#define CORDIC_INST_KIND 1
#define CORDIC_INST_NAME launch_rot_lj24
#define CORDIC_INST_CONTAINER dev::WideLJ<24>
#define CORDIC_INST_NGEN 7
#define CORDIC_INST_DYN_ONLY
#include "cordic_inst_body.h"
