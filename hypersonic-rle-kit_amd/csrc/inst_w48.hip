// 48 bit symbols: rle48_{sym,byte}[_packed], rle48_{3,7}symlut_{sym,byte}  (reference: src/rle.h)
#define HSRLE_W 48
#define HSRLE_S 6
#define HSRLE_BASE 30
#include "hsrle_inst_generic.inc"
