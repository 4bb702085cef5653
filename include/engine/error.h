/*
 * engine/error.h - forwarding header (reference include/engine/error.h:11-15): the MIDORIDB_* status codes live in
 * include/mdb_error.h with the reference's values.
 */
#ifndef MDB_FORWARD_ENGINE_ERROR_H
#define MDB_FORWARD_ENGINE_ERROR_H

#include "../mdb_error.h"

#endif
