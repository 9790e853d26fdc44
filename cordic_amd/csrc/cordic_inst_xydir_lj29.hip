// cordic_inst_xydir_lj29.hip -- instantiation unit (see cordic_inst_xydir_body.h)
#define CORDIC_XYDIR_NAME launch_xydir_lj29
#define CORDIC_XYDIR_LJ 29
#include "cordic_inst_xydir_body.h"
