// cordic_inst_pol_wide8.hip -- instantiation unit (see cordic_inst_body.h)
#define CORDIC_INST_KIND 2
#define CORDIC_INST_NAME launch_pol_wide8
#define CORDIC_INST_CONTAINER dev::Wide64
#define CORDIC_INST_NGEN 8
#include "cordic_inst_body.h"
