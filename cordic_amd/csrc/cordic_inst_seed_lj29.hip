// cordic_inst_seed_lj29.hip -- instantiation unit (see cordic_inst_body.h)
#define CORDIC_INST_KIND 3
#define CORDIC_INST_NAME launch_seed_lj29
#define CORDIC_INST_CONTAINER dev::WideLJ<29>
#define CORDIC_INST_NGEN 0
// job sets of the 16- / 24-stage cores on their own static instances
#define CORDIC_INST_DESC_STATIC 1
#include "cordic_inst_body.h"
