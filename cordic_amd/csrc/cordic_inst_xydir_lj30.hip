// cordic_inst_xydir_lj30.hip -- instantiation unit (see cordic_inst_xydir_body.h)
#define CORDIC_XYDIR_NAME launch_xydir_lj30
#define CORDIC_XYDIR_JOBS_NAME launch_xydir_jobs_lj30
// WW <= 34: the cores gencordic derives for 16- and 24-bit ports (19 / 27
// stages) and a 16-stage core
#define CORDIC_XYDIR_JOB_STAGES(X) X(16) X(19) X(27)
#define CORDIC_XYDIR_LJ 30
#include "cordic_inst_xydir_body.h"
