/*
 * engine/database.h - forwarding header (reference include/engine/database.h:15-32): struct database,
 * database_open() and database_close() are declared by include/mdb_query.h.
 */
#ifndef MDB_FORWARD_ENGINE_DATABASE_H
#define MDB_FORWARD_ENGINE_DATABASE_H

#include "../mdb_query.h"

#endif
