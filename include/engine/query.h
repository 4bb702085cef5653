/*
 * engine/query.h - forwarding header: the include line of programs written against the reference
 * (`#include <engine/query.h>`, reference README.md:49, tests/engine/executor_select.c) resolves here and gets
 * this library's drop-in declarations (include/mdb_query.h) plus the libc headers the reference's own
 * <engine/query.h> pulls in through <compiler/common.h> (reference include/compiler/common.h:4-14) - programs such
 * as the README's use printf() without including <stdio.h> themselves.
 */
#ifndef MDB_FORWARD_ENGINE_QUERY_H
#define MDB_FORWARD_ENGINE_QUERY_H

#include <stdlib.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdint.h>
#include <string.h>
#include <stddef.h>
#include <stdbool.h>
#include <pthread.h>
#include <limits.h>
#include <math.h>
#include <time.h>
#include "error.h"
#include "database.h"
#include "../mdb_query.h"

#endif
