// cordic_inst_xydir_lj30.hip -- instantiation unit (see cordic_inst_xydir_body.h)
#define CORDIC_XYDIR_NAME launch_xydir_lj30
#define CORDIC_XYDIR_LJ 30
#include "cordic_inst_xydir_body.h"
