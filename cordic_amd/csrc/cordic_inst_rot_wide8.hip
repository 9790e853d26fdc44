// cordic_inst_rot_wide8.hip -- instantiation unit (see cordic_inst_body.h)
#define CORDIC_INST_KIND 1
#define CORDIC_INST_NAME launch_rot_wide8
#define CORDIC_INST_CONTAINER dev::Wide64
#define CORDIC_INST_NGEN 8
// Round 5: the dynamic-exit instance only.  Static
// instances of this container measured within 2-4 % of it on their own cores
// (profiles/r05/static_vs_dyn.txt: +1.9 / +3.0 per cent) and no core
// gencordic derives by itself -- nor any BASELINE configuration -- runs here.
#define CORDIC_INST_DYN_ONLY
#include "cordic_inst_body.h"
