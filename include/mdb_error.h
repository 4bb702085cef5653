/*
 * mdb_error.h - status codes shared by every entry point of libmidoridb_amd.so.
 *
 * Same numeric values and the same convention as the reference
 * (reference include/engine/error.h:11-15): functions return MIDORIDB_OK (0) or a
 * NEGATIVE code (-MIDORIDB_ERROR, -MIDORIDB_INTERNAL, -MIDORIDB_NOMEM);
 * query_cur_step() returns MIDORIDB_ROW (4) while a row is current.
 */
#ifndef MDB_ERROR_H
#define MDB_ERROR_H

#define MIDORIDB_OK        0	/* Successful result */
#define MIDORIDB_ERROR     1	/* Generic error */
#define MIDORIDB_INTERNAL  2	/* Internal error - includes every HIP/RCCL failure */
#define MIDORIDB_NOMEM     3	/* Resource couldn't be allocated */
#define MIDORIDB_ROW       4	/* Next row is available */

#endif /* MDB_ERROR_H */
