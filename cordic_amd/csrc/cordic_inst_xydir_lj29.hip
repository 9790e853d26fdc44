// cordic_inst_xydir_lj29.hip -- instantiation unit (see cordic_inst_xydir_body.h)
#define CORDIC_XYDIR_NAME launch_xydir_lj29
#define CORDIC_XYDIR_JOBS_NAME launch_xydir_jobs_lj29
// WW 35: BASELINE's 16- and 24-stage cores, gencordic's own 29 stages (-i 32)
#define CORDIC_XYDIR_JOB_STAGES(X) X(16) X(24) X(29)
#define CORDIC_XYDIR_LJ 29
#include "cordic_inst_xydir_body.h"
