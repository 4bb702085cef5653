/*
 * mdb_legacy.c - the reference's row-store layout in front of a result (include/mdb_legacy.h): header fields for every result, the
 * datablock list for small ones.  Host C; what it stands for: the result table executor_run_select_stmt leaves in
 * output->results.table (/root/reference/src/engine/executor_select.c:1728) as table_insert_row builds it (src/primitive/row.c:26-124).
 */
#include "mdb_host.h"

static size_t legacy_col_space(int type)
{
	return type == MDB_CT_TINYINT ? 1 : 8;	/* (table_calc_column_space: the precision, a pointer for VARCHAR) */
}

void mdb_result_legacy_header(struct mdb_result *r)
{
	struct mdb_legacy_table *t = &r->legacy;
	snprintf(t->name, sizeof(t->name), "early_mat_tbl");	/* (executor_select.c:314) */
	t->column_count = r->ncols < MDB_LEGACY_MAX_COLUMNS ? r->ncols : MDB_LEGACY_MAX_COLUMNS;
	for (int c = 0; c < t->column_count; c++) {
		struct mdb_legacy_column *col = &t->columns[c];
		memset(col, 0, sizeof(*col));
		snprintf(col->name, sizeof(col->name), "%s", r->colname[c]);
		col->type = r->coltype[c];
		/* VARCHAR: the DECLARED length - upstream allocates that many bytes per cell and its own code copies that many from the cell's
		 * pointer (row.c table_insert_row), so the cells below are buffers of exactly that size */
		col->precision = r->coltype[c] == MDB_CT_VARCHAR ? (r->colprec && r->colprec[c] > 0 ? r->colprec[c] : 256) : (int)legacy_col_space(r->coltype[c]);
		col->nullable = true;
		col->is_count = strcmp(r->colname[c], "COUNT(*)") == 0;
	}
	r->legacy_head.next = r->legacy_head.prev = &r->legacy_head;
	t->datablock_head = &r->legacy_head;
	t->free_dtbkl_offset = 0;
	pthread_mutex_init(&t->mutex, NULL);
}

void mdb_result_legacy_free(struct mdb_result *r)
{
	struct mdb_legacy_list_head *p = r->legacy_head.next;
	while (p && p != &r->legacy_head) {
		struct mdb_legacy_list_head *next = p->next;
		free((char *)p - offsetof(struct mdb_legacy_datablock, head));
		p = next;
	}
	r->legacy_head.next = r->legacy_head.prev = &r->legacy_head;
	r->legacy_built = false;
	free(r->legacy_text);
	r->legacy_text = NULL;
}

/* the rows of a small result whose columns are on the host, as datablocks; a result that is too large, not fetched yet, or wider than the
 * reference's 128 columns keeps the empty list.  Out of memory: the empty list again (the columnar result is what callers are promised). */
void mdb_result_legacy_rows(struct mdb_result *r)
{
	struct mdb_legacy_table *t = &r->legacy;
	if (r->legacy_built || r->nrows > MDB_LEGACY_MAX_ROWS || r->ncols > MDB_LEGACY_MAX_COLUMNS || !r->data)
		return;
	for (int c = 0; c < r->ncols; c++)
		if (!r->data[c] && r->nrows)
			return;		/* (still on the device) */
	size_t row_size = sizeof(struct mdb_legacy_row);
	for (int c = 0; c < r->ncols; c++)
		row_size += legacy_col_space(r->coltype[c]);
	const size_t per_block = MDB_LEGACY_PAGE_SIZE / row_size;
	if (!per_block)
		return;
	/* VARCHAR cells: one zero-filled buffer of `precision` bytes per cell (what upstream's zalloc'd cell is), so a consumer that copies
	 * column->precision bytes from the pointer - as upstream's table_insert_row does - stays inside it.  More than 64 MiB of them: the
	 * empty list (the cursor serves such a result) */
	size_t text_bytes = 0;
	for (int c = 0; c < r->ncols; c++)
		if (r->coltype[c] == MDB_CT_VARCHAR)
			text_bytes += (size_t)t->columns[c].precision * r->nrows;
	if (text_bytes > ((size_t)64 << 20))
		return;
	char *text = NULL;
	if (text_bytes && !(text = calloc(1, text_bytes)))
		return;
	free(r->legacy_text);
	r->legacy_text = text;
	struct mdb_legacy_datablock *blk = NULL;
	size_t free_off = 0;
	uint64_t id = 0;
	for (uint64_t i = 0; i < r->nrows; i++) {
		/* (table_insert_row, row.c:36-52: a new block when the list is empty or the row would reach the page's end; every slot of a new
		 * block starts out `empty`, table.c:126-134 - the first unused one of the last block is what ends a walk) */
		if (!blk || free_off + row_size >= MDB_LEGACY_PAGE_SIZE) {
			struct mdb_legacy_datablock *nb = calloc(1, sizeof(*nb));
			if (!nb) {
				mdb_result_legacy_free(r);
				return;
			}
			nb->block_id = id++;
			for (size_t k = 0; k < per_block; k++)
				((struct mdb_legacy_row *)&nb->data[k * row_size])->empty = true;
			nb->head.prev = r->legacy_head.prev;
			nb->head.next = &r->legacy_head;
			r->legacy_head.prev->next = &nb->head;
			r->legacy_head.prev = &nb->head;
			blk = nb;
			free_off = 0;
		}
		struct mdb_legacy_row *row = (struct mdb_legacy_row *)&blk->data[free_off];
		row->empty = false;
		size_t off = 0;
		for (int c = 0; c < r->ncols; c++) {
			const bool isnull = r->nullbits && r->nullbits[c] && ((r->nullbits[c][i >> 6] >> (i & 63)) & 1);
			if (isnull)
				row->null_bitmap[c >> 3] |= (char)(1u << (c & 7));
			if (r->coltype[c] == MDB_CT_TINYINT) {
				row->data[off] = isnull ? 0 : (char)(r->data[c][i] != 0);
			} else if (r->coltype[c] == MDB_CT_VARCHAR) {
				const char *sv = isnull ? NULL : (r->dict ? mdb_dict_str(r->dict, r->data[c][i]) : NULL);
				const size_t prec = (size_t)t->columns[c].precision;
				if (sv && prec)
					snprintf(text, prec, "%s", sv);
				const uintptr_t pv = (uintptr_t)text;
				text += prec;
				memcpy(row->data + off, &pv, sizeof(pv));
			} else {
				const int64_t v = isnull ? 0 : r->data[c][i];
				memcpy(row->data + off, &v, 8);
			}
			off += legacy_col_space(r->coltype[c]);
		}
		free_off += row_size;
	}
	t->free_dtbkl_offset = free_off;
	r->legacy_built = true;
}
