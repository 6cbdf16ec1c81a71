// 24 bit symbols: rle24_{sym,byte}[_packed], rle24_{3,7}symlut_{sym,byte}  (reference: src/rle.h)
#define HSRLE_W 24
#define HSRLE_S 3
#define HSRLE_BASE 14
#include "hsrle_inst_generic.inc"
