/*
 * record_groups.h -- the record program (record_kernel.h) compiled for two workgroup shapes:
 *   g256  four waves per record: the op store of up to 8192 ops in LDS, block collectives with one barrier each -- every emit kernel
 *         and the sizing of the long records;
 *   g64   one wave per record: the same code with PAFFY_NT = 64, so every collective is a DPP wave scan and every barrier vanishes,
 *         16 records in flight per CU instead of 4. Used for the sizing pass of records with short cigars, whose cost is the
 *         per-record chain of dependent loads and barriers, not their ops (cfg3: 1.9 of 4.3 ms did not depend on the op count).
 * Everything outside the record program uses g256 (the `using` below).
 */
#ifndef PAFFY_RECORD_GROUPS_H_
#define PAFFY_RECORD_GROUPS_H_

#include <type_traits>

#include "device_util.h"
#include "record_types.h"

#define PAFFY_NT 64
#define PAFFY_NWAVE 1
#define PAFFY_NS g64
#include "block_util.h"
#include "record_kernel.h"
#undef PAFFY_NT
#undef PAFFY_NWAVE
#undef PAFFY_NS

#define PAFFY_NT 256
#define PAFFY_NWAVE 4
#define PAFFY_NS g256
#include "block_util.h"
#include "record_kernel.h"
#undef PAFFY_NS

using namespace g256;

#endif
