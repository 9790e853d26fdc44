// cordic_inst_rot_wideall.hip -- instantiation unit (see cordic_inst_body.h)
#define CORDIC_INST_KIND 1
#define CORDIC_INST_NAME launch_rot_wideall
#define CORDIC_INST_CONTAINER dev::Wide64
#define CORDIC_INST_NGEN kAllGeneral
#include "cordic_inst_body.h"
