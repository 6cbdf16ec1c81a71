// 64 bit symbols: rle64_{sym,byte}[_packed], rle64_{3,7}symlut_{sym,byte}  (reference: src/rle.h)
#define HSRLE_W 64
#define HSRLE_S 8
#define HSRLE_BASE 38
#include "hsrle_inst_generic.inc"
