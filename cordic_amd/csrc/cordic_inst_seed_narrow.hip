// cordic_inst_seed_narrow.hip -- instantiation unit (see cordic_inst_body.h)
#define CORDIC_INST_KIND 3
#define CORDIC_INST_NAME launch_seed_narrow
#define CORDIC_INST_CONTAINER dev::Narrow32
#define CORDIC_INST_NGEN 0
#include "cordic_inst_body.h"
