// cordic_inst_rot_wideall.hip -- instantiation unit (see cordic_inst_body.h)
#define CORDIC_INST_KIND 1
#define CORDIC_INST_NAME launch_rot_wideall
#define CORDIC_INST_CONTAINER dev::Wide64
#define CORDIC_INST_NGEN kAllGeneral
// Round 5: the dynamic-exit instance only.  Static
// instances of this container measured within 2-4 % of it on their own cores
// (profiles/r05/static_vs_dyn.txt: +2.0 / +2.8 per cent) and no core
// gencordic derives by itself -- nor any BASELINE configuration -- runs here.
#define CORDIC_INST_DYN_ONLY
#include "cordic_inst_body.h"
