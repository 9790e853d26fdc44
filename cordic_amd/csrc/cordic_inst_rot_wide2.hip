// cordic_inst_rot_wide2.hip -- instantiation unit (see cordic_inst_body.h)
#define CORDIC_INST_KIND 1
#define CORDIC_INST_NAME launch_rot_wide2
#define CORDIC_INST_CONTAINER dev::Wide64
#define CORDIC_INST_NGEN 2
#include "cordic_inst_body.h"
