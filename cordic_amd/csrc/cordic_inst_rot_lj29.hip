// cordic_inst_rot_lj29.hip -- instantiation unit (see cordic_inst_body.h)
#define CORDIC_INST_KIND 1
#define CORDIC_INST_NAME launch_rot_lj29
#define CORDIC_INST_CONTAINER dev::WideLJ<29>
#define CORDIC_INST_NGEN 2
#include "cordic_inst_body.h"
