// cordic_inst_pol_narrow.hip -- instantiation unit (see cordic_inst_body.h)
#define CORDIC_INST_KIND 2
#define CORDIC_INST_NAME launch_pol_narrow
#define CORDIC_INST_CONTAINER dev::Narrow32
#define CORDIC_INST_NGEN 0
#include "cordic_inst_body.h"
