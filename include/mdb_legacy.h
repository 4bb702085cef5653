/*
 * mdb_legacy.h - the reference's row-store view of a SMALL result (optional; 8f row 3).
 *
 * Upstream, `struct result_set.table` is a `struct table *`: a row store of 4 KiB datablocks
 * (/root/reference/include/engine/query.h:24-28, include/primitive/table.h:23-42, datablock.h:9-13, row.h:15-28, column.h:30-49,
 * datastructure/linkedlist.h:11-14).  Here the result is columnar (include/mdb_query.h), but what `results.table` points at BEGINS
 * with a structure of exactly that layout, so that a consumer compiled against the reference's headers that walks
 * `output->results.table->datablock_head` itself - instead of calling query_cur_step() / query_column_int64() - still finds its rows:
 *   - name, columns[] (name, type, precision, the six flags), column_count: always filled;
 *   - the datablock list: filled for results of at most MDB_LEGACY_MAX_ROWS rows once the result's columns are on the host (at once for
 *     an ordinary result; after the first query_cur_step() / query_column_data() for one kept on the device); otherwise an EMPTY list
 *     (datablock_head->next == datablock_head) - larger results are read through the cursor or query_column_data().
 * Rows: 2 flag bytes (empty, deleted), a 16-byte NULL bitmap (bit c = column c, LSB first), payload from byte 24 on, the columns back
 * to back at their `precision` (8 bytes; TINYINT 1; VARCHAR a pointer to a NUL-terminated string owned by the database); rows back to
 * back from the start of a block, a row never straddles two blocks, the first unused row of the last block is marked `empty`.
 * The structs restate layouts, not code; tests walk them with the reference's own compiled accessors (oracle/_ref).
 */
#ifndef MDB_LEGACY_H
#define MDB_LEGACY_H

#include <pthread.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#define MDB_LEGACY_MAX_ROWS 4096u
#define MDB_LEGACY_PAGE_SIZE 4096	/* DATABLOCK_PAGE_SIZE, datablock.h:7 */
#define MDB_LEGACY_MAX_COLUMNS 128	/* TABLE_MAX_COLUMNS, table.h:16 */

struct mdb_legacy_list_head {		/* linkedlist.h:11-14 */
	struct mdb_legacy_list_head *next, *prev;
};

struct mdb_legacy_datablock {		/* datablock.h:9-13 */
	uint64_t block_id;
	char data[MDB_LEGACY_PAGE_SIZE];
	struct mdb_legacy_list_head head;
};

struct mdb_legacy_column {		/* column.h:30-49 */
	char name[128];
	int type;			/* enum COLUMN_TYPE, column.h:17-25 */
	int precision;
	bool indexed, nullable, unique, auto_inc, primary_key, is_count;
};

struct mdb_legacy_table {		/* table.h:23-42 */
	char name[128];
	struct mdb_legacy_column columns[MDB_LEGACY_MAX_COLUMNS];
	int column_count;
	struct mdb_legacy_list_head *datablock_head;
	size_t free_dtbkl_offset;
	pthread_mutex_t mutex;
};

struct mdb_legacy_row {			/* row.h:15-28 */
	bool empty, deleted;
	char null_bitmap[MDB_LEGACY_MAX_COLUMNS / 8];
	char data[] __attribute__((aligned(8)));
};

#endif /* MDB_LEGACY_H */
