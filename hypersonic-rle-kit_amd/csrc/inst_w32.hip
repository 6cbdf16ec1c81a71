// 32 bit symbols: rle32_{sym,byte}[_packed], rle32_{3,7}symlut_{sym,byte}  (reference: src/rle.h)
#define HSRLE_W 32
#define HSRLE_S 4
#define HSRLE_BASE 22
#include "hsrle_inst_generic.inc"
