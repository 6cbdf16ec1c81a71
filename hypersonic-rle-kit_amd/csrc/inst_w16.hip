// 16 bit symbols: rle16_{sym,byte}[_packed], rle16_{3,7}symlut_{sym,byte}  (reference: src/rle.h)
#define HSRLE_W 16
#define HSRLE_S 2
#define HSRLE_BASE 6
#include "hsrle_inst_generic.inc"
