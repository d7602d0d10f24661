/* inc/paf.h -- the reference's include path for its record API (a program says #include "paf.h" with -I inc): the header itself
 * is include/paf.h. */
#include "../include/paf.h"
