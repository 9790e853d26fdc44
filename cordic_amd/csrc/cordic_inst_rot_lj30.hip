// cordic_inst_rot_lj30.hip -- instantiation unit (see cordic_inst_body.h)
#define CORDIC_INST_KIND 1
#define CORDIC_INST_NAME launch_rot_lj30
#define CORDIC_INST_CONTAINER dev::WideLJ<30>
#define CORDIC_INST_NGEN 1
#include "cordic_inst_body.h"
