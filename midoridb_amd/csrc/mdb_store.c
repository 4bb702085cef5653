/*
 * mdb_store.c - catalog and columnar table storage with device mirrors.
 *
 * Replaces, on this path, the reference's row store (reference src/primitive/: struct table with a
 * circular list of 4 KiB datablocks, 24-byte row header + 8-aligned payload, include/primitive/
 * row.h:15-28): every column is one contiguous int64_t[] / double[] plus a NULL bitmap, which is
 * exactly the layout the kernels read (include/mdb_dev.h).  INSERT appends on the host and the next
 * SELECT uploads only the appended tail into the device mirror (generation counter, spare capacity);
 * DELETE and UPDATE are applied on the mirror itself (mdb_exec.c) - SURVEY.md 8f row 1.
 */
#define _XOPEN_SOURCE 700
#define _DEFAULT_SOURCE
#include <time.h>
#include <sys/mman.h>
#include "mdb_host.h"

struct mdb_table *mdb_catalog_find(struct mdb_catalog *cat, const char *name)
{
	for (int i = 0; i < cat->n; i++)
		if (strcmp(cat->tables[i]->name, name) == 0)
			return cat->tables[i];
	return NULL;
}

int mdb_catalog_add(struct mdb_catalog *cat, struct mdb_table *t)
{
	if (cat->n == cat->cap) {
		int nc = cat->cap ? cat->cap * 2 : 8;
		struct mdb_table **nt = realloc(cat->tables, sizeof(*nt) * (size_t)nc);
		if (!nt)
			return -MIDORIDB_NOMEM;
		cat->tables = nt;
		cat->cap = nc;
	}
	cat->tables[cat->n++] = t;
	return MIDORIDB_OK;
}

struct mdb_table *mdb_table_new(const char *name)
{
	struct mdb_table *t = calloc(1, sizeof(*t));
	if (!t)
		return NULL;
	mdb_copy_name(t->name, name);
	t->generation = 1;
	return t;
}

static void col_distinct_update(struct mdb_catalog *cat, struct mdb_table *t, struct mdb_column *col, bool follow, uint64_t rows_from);

static void table_drop_device(struct mdb_table *t, mdb_dev_ctx *dev)
{
	for (int c = 0; c < t->ncols; c++) {
		if (dev && t->cols[c].d_data)
			mdb_dev_free(dev, t->cols[c].d_data);
		if (dev && t->cols[c].d_nullbits)
			mdb_dev_free(dev, t->cols[c].d_nullbits);
		if (dev && t->cols[c].d_seen)
			mdb_dev_free(dev, t->cols[c].d_seen);
		t->cols[c].d_data = NULL;
		t->cols[c].d_nullbits = NULL;
		t->cols[c].d_seen = NULL;
		t->cols[c].dv_generation = 0;
	}
	t->dev_generation = 0;
	t->dev_rows = 0;
	t->dev_cap = 0;
}

void mdb_table_free(struct mdb_table *t, mdb_dev_ctx *dev)
{
	if (!t)
		return;
	table_drop_device(t, dev);
	for (int c = 0; c < t->ncols; c++) {
		free(t->cols[c].data);
		free(t->cols[c].nullbits);
	}
	free(t);
}

bool mdb_parse_time(const char *quoted, int type, int64_t *out)
{
	char buf[64];
	struct tm tmv;
	size_t len = strlen(quoted);
	const char *fmt = type == MDB_CT_DATE ? "%Y-%m-%d" : "%Y-%m-%d %H:%M:%S";
	time_t tv;
	/* the token carries its quotes (midorisql.l STRING) */
	if (len >= 2 && (quoted[0] == '\'' || quoted[0] == '"') && quoted[len - 1] == quoted[0]) {
		quoted++;
		len -= 2;
	}
	if (len >= sizeof(buf))
		return false;
	memcpy(buf, quoted, len);
	buf[len] = 0;
	memset(&tmv, 0, sizeof(tmv));
	if (!strptime(buf, fmt, &tmv))
		return false;
	tv = mktime(&tmv);
	if (tv == (time_t)-1)
		return false;
	*out = (int64_t)tv;
	return true;
}

/* ------------------------------------------------------------------ string dictionary */
static uint64_t dict_hash(const char *s, size_t len)
{
	uint64_t h = 0xcbf29ce484222325ull;
	for (size_t i = 0; i < len; i++)
		h = (h ^ (unsigned char)s[i]) * 0x100000001b3ull;
	return h ^ (h >> 29);
}

static uint64_t dict_probe(const struct mdb_strdict *d, const char *s, size_t len, bool *found)
{
	uint64_t i = dict_hash(s, len) & (d->nslots - 1);
	*found = false;
	while (d->slot[i]) {
		const uint64_t id = d->slot[i];
		if (d->len[id - 1] == len && memcmp(d->str[id - 1], s, len) == 0) {
			*found = true;
			return i;
		}
		i = (i + 1) & (d->nslots - 1);
	}
	return i;
}

int64_t mdb_dict_find(const struct mdb_strdict *d, const char *s, size_t len)
{
	bool found;
	if (!d->nslots)
		return -1;
	const uint64_t i = dict_probe(d, s, len, &found);
	return found ? (int64_t)d->slot[i] : -1;
}

int64_t mdb_dict_intern(struct mdb_strdict *d, const char *s, size_t len)
{
	bool found;
	if ((d->n + 1) * 2 > d->nslots) {	/* keep the table at most half full */
		const uint64_t ns = d->nslots ? d->nslots * 2 : 1024;
		uint64_t *slot = calloc(ns, sizeof(*slot));
		if (!slot)
			return 0;
		for (uint64_t id = 1; id <= d->n; id++) {
			uint64_t i = dict_hash(d->str[id - 1], d->len[id - 1]) & (ns - 1);
			while (slot[i])
				i = (i + 1) & (ns - 1);
			slot[i] = id;
		}
		free(d->slot);
		d->slot = slot;
		d->nslots = ns;
	}
	const uint64_t i = dict_probe(d, s, len, &found);
	if (found)
		return (int64_t)d->slot[i];
	if (d->n == d->cap) {
		const uint64_t nc = d->cap ? d->cap * 2 : 1024;
		char **str = realloc(d->str, nc * sizeof(*str));
		if (!str)
			return 0;
		d->str = str;
		uint32_t *ln = realloc(d->len, nc * sizeof(*ln));
		if (!ln)
			return 0;
		d->len = ln;
		d->cap = nc;
	}
	char *copy = malloc(len + 1);
	if (!copy)
		return 0;
	memcpy(copy, s, len);
	copy[len] = 0;
	d->str[d->n] = copy;
	d->len[d->n] = (uint32_t)len;
	d->n++;
	d->slot[i] = d->n;
	return (int64_t)d->n;
}

const char *mdb_dict_str(const struct mdb_strdict *d, int64_t id)
{
	return id >= 1 && (uint64_t)id <= d->n ? d->str[id - 1] : NULL;
}

void mdb_dict_free(struct mdb_strdict *d)
{
	for (uint64_t i = 0; i < d->n; i++)
		free(d->str[i]);
	free(d->str);
	free(d->len);
	free(d->slot);
	memset(d, 0, sizeof(*d));
}

void mdb_catalog_free(struct mdb_catalog *cat)
{
	mdb_dict_free(&cat->dict);
	mdb_dict_free(&cat->gdict);
	free(cat->l2g);
	free(cat->g2l);
	if (cat->dev) {
		(void)mdb_dev_free(cat->dev, cat->d_l2g);
		(void)mdb_dev_free(cat->dev, cat->d_g2l);
	}
	for (int i = 0; i < cat->n; i++)
		mdb_table_free(cat->tables[i], cat->dev);
	free(cat->tables);
	if (cat->dist)
		mdb_dist_destroy(cat->dist);
	if (cat->dev)
		mdb_dev_ctx_destroy(cat->dev);
	memset(cat, 0, sizeof(*cat));
}

int mdb_table_add_column(struct mdb_table *t, const char *name, int type)
{
	struct mdb_column *c;
	if (t->ncols >= MDB_MAX_COLS)
		return -MIDORIDB_ERROR;
	c = &t->cols[t->ncols++];
	memset(c, 0, sizeof(*c));
	mdb_copy_name(c->name, name);
	c->type = type;
	return MIDORIDB_OK;
}

/* A column of millions of rows is first touched by the bulk loader's threads: with 4 KiB pages that is 2.4 x 10^5 page faults per GB - most of
 * what an ingest of host arrays costs.  Where the kernel hands out transparent huge pages on request (THP "madvise"), ask for them (2 MiB
 * pages: 512 x fewer faults); MDB_INGEST_THP=0: do not ask. */
static void mdb_ask_huge_pages(void *p, size_t bytes)
{
	const uintptr_t a = ((uintptr_t)p + ((size_t)2 << 20) - 1) & ~(uintptr_t)(((size_t)2 << 20) - 1), e = ((uintptr_t)p + bytes) & ~(uintptr_t)(((size_t)2 << 20) - 1);
	const char *knob = mdb_knob("MDB_INGEST_THP");
	if (bytes < ((size_t)8 << 20) || e <= a || (knob && knob[0] == '0'))
		return;
	(void)madvise((void *)a, (size_t)(e - a), MADV_HUGEPAGE);	/* (advice: a kernel without THP says EINVAL, nothing depends on it) */
}

int mdb_table_reserve(struct mdb_table *t, uint64_t rows)
{
	uint64_t ncap;
	if (rows <= t->cap)
		return MIDORIDB_OK;
	ncap = t->cap ? t->cap : 64;
	while (ncap < rows)
		ncap *= 2;
	for (int c = 0; c < t->ncols; c++) {
		const uint64_t old_words = (t->cap + 63) / 64, new_words = (ncap + 63) / 64;
		int64_t *nd = realloc(t->cols[c].data, sizeof(int64_t) * ncap);
		uint64_t *nb;
		if (!nd)
			return -MIDORIDB_NOMEM;
		t->cols[c].data = nd;
		mdb_ask_huge_pages(nd, sizeof(int64_t) * ncap);
		nb = realloc(t->cols[c].nullbits, sizeof(uint64_t) * new_words);
		if (!nb)
			return -MIDORIDB_NOMEM;
		memset(nb + old_words, 0, sizeof(uint64_t) * (new_words - old_words));
		t->cols[c].nullbits = nb;
	}
	t->cap = ncap;
	return MIDORIDB_OK;
}

int mdb_catalog_device(struct mdb_catalog *cat, char *err, size_t errlen)
{
	if (!cat->dev && cat->dev_rc == 0) {
		const char *env = getenv("MIDORIDB_DEVICE");
		int device = env ? atoi(env) : 0;
		int n = mdb_dev_device_count();
		if (n <= 0) {
			cat->dev_rc = -MIDORIDB_INTERNAL;
		} else {
			cat->dev_rc = mdb_dev_ctx_create(device, MDB_STREAM_OWN, &cat->dev);
		}
		/* sharded mode: MIDORIDB_WORLD_SIZE ranks, this one MIDORIDB_RANK, the communicator id through the file
		 * MIDORIDB_DIST_ID_FILE (rank 0 writes it).  Collective: every rank's first SELECT gets here. */
		if (cat->dev && getenv("MIDORIDB_WORLD_SIZE") && atoi(getenv("MIDORIDB_WORLD_SIZE")) >= 1) {	/* (1: a one-rank exchange, for tests) */
			const int world = atoi(getenv("MIDORIDB_WORLD_SIZE"));
			const int rank = getenv("MIDORIDB_RANK") ? atoi(getenv("MIDORIDB_RANK")) : -1;
			const char *idf = getenv("MIDORIDB_DIST_ID_FILE");
			char id[MDB_DIST_ID_BYTES];
			int rc = (rank < 0 || rank >= world || !idf) ? -MIDORIDB_ERROR : mdb_dist_id_via_file(idf, world, rank, 300.0, id);
			if (!rc)
				rc = mdb_dist_init(cat->dev, world, rank, id, &cat->dist);
			if (rc) {
				snprintf(err, errlen, "execution phase: cannot join the %d-rank exchange (MIDORIDB_RANK / MIDORIDB_DIST_ID_FILE): %s\n", world,
					 mdb_dev_last_error(cat->dev));
				mdb_dev_ctx_destroy(cat->dev);
				cat->dev = NULL;
				cat->dev_rc = rc;
				return rc;
			}
		}
	}
	if (!cat->dev) {
		snprintf(err, errlen,
			 "execution phase: no usable HIP device (MI355X path only - this library has no CPU executor)\n");
		return cat->dev_rc ? cat->dev_rc : -MIDORIDB_INTERNAL;
	}
	return MIDORIDB_OK;
}

/* Bring the device mirror up to date.  Tables only ever grow (INSERT / bulk append), so when the device
 * buffers still have room only the appended tail travels over PCIe (device-resident
 * table cache with INSERT invalidation); otherwise the buffers are re-created at the host capacity. */
int mdb_table_sync_device(struct mdb_catalog *cat, struct mdb_table *t, char *err, size_t errlen)
{
	int rc = mdb_catalog_device(cat, err, errlen);
	if (rc)
		return rc;
	if (t->device_only || (t->dev_generation == t->generation && t->dev_rows == t->nrows))
		return MIDORIDB_OK;
	const bool append = t->dev_generation != 0 && t->nrows >= t->dev_rows && t->nrows <= t->dev_cap;
	if (!append) {
		table_drop_device(t, cat->dev);
		t->dev_cap = t->cap > t->nrows ? t->cap : t->nrows;
	}
	const uint64_t from = append ? t->dev_rows : 0;
	for (int c = 0; c < t->ncols; c++) {
		struct mdb_column *col = &t->cols[c];
		if (t->nrows == 0)
			continue;
		rc = MIDORIDB_OK;
		if (!col->d_data)
			rc = mdb_dev_alloc(cat->dev, t->dev_cap * 8, &col->d_data);
		if (!rc && t->nrows > from)
			rc = mdb_dev_h2d(cat->dev, (char *)col->d_data + from * 8, col->data + from, (t->nrows - from) * 8);
		/* The device bitmap exists from the column's first NULL on and is then kept current for good: an UPDATE may
		 * bring null_count back to 0 while the kernels still read the bitmap, so every later append uploads its
		 * words too.  mdb_dev_alloc() recycles buffers without clearing them: the words beyond the rows uploaded so
		 * far are zeroed here, once. */
		if (!rc && (col->null_count || col->d_nullbits)) {
			const uint64_t words = (t->dev_cap + 63) / 64;
			uint64_t w0 = from / 64;
			if (!col->d_nullbits) {		/* first NULL of this column: the whole bitmap goes up */
				rc = mdb_dev_alloc(cat->dev, words * 8, (void **)&col->d_nullbits);
				if (!rc)
					rc = mdb_dev_memset(cat->dev, col->d_nullbits, 0, words * 8);
				w0 = 0;
			}
			if (!rc)
				rc = mdb_dev_h2d(cat->dev, col->d_nullbits + w0, col->nullbits + w0, ((t->nrows + 63) / 64 - w0) * 8);
		}
		if (rc) {
			snprintf(err, errlen, "execution phase: cannot mirror table '%s' on the device: %s\n", t->name,
				 mdb_dev_last_error(cat->dev));
			table_drop_device(t, cat->dev);
			return rc;
		}
	}
	/* catalog statistics (smallest / largest non-NULL value of every integer-like column), kept with the mirror: rows that were
	 * appended widen what was known (one pass over the uploaded tail), anything else is looked at whole.  The operators take their key
	 * windows from these instead of sampling the columns on every first query (mdb_dev_call_stats, include/mdb_dev.h) */
	for (int c = 0; c < t->ncols && t->nrows; c++) {
		struct mdb_column *col = &t->cols[c];
		if (!mdb_col_has_range(col) || !col->d_data)
			continue;
		const bool widen = append && col->st_generation == t->dev_generation + 1 && t->nrows > from;
		const uint64_t r0 = widen ? from : 0;
		int64_t lo = 0, hi = -1;
		/* (a NULL bitmap is read by row index: the tail's range is taken over whole words from the word that holds its first row) */
		const uint64_t a0 = col->d_nullbits ? (r0 & ~(uint64_t)63) : r0;
		if (mdb_dev_key_range(cat->dev, (const int64_t *)col->d_data + a0, col->d_nullbits ? col->d_nullbits + a0 / 64 : NULL, t->nrows - a0, &lo, &hi)) {
			col->st_generation = 0;		/* (not fatal: looked at again when somebody asks) */
			continue;
		}
		if (widen && col->st_lo <= col->st_hi) {
			lo = lo <= hi && lo < col->st_lo ? lo : col->st_lo;
			hi = hi > col->st_hi ? hi : col->st_hi;
			if (lo > hi) {
				lo = col->st_lo;
				hi = col->st_hi;
			}
		}
		col->st_lo = lo;
		col->st_hi = hi;
		col->st_generation = t->generation + 1;
		col_distinct_update(cat, t, col, append && col->dv_generation == t->dev_generation + 1 && t->nrows >= from, from);
	}
	t->dev_generation = t->generation;
	t->dev_rows = t->nrows;
	return MIDORIDB_OK;
}

/* ------------------------------------------------------------------ bulk ingest of plain 8-byte columns (round 5)
 *
 * mdb_table_append_columns() used to copy a column at a time on one thread (first touch of fresh pages: ~3 GB/s) and leave the upload to the
 * first SELECT.  Here the rows go in chunks: MDB_INGEST_THREADS workers (default: the host's cores, at most 8) copy chunk k of every
 * column into the host store while the calling thread uploads chunk k - 1 into the device mirror, so the copy and the PCIe transfer overlap
 * and the first SELECT finds the mirror current.  What the reference does per row instead: table_insert_row (src/primitive/row.c:26-124)
 * behind executor_insert.c:194-249. */
#include <pthread.h>
#include <unistd.h>

#define INGEST_CHUNK_ROWS ((uint64_t)1 << 23)
#define INGEST_MAX_THREADS 8

struct ingest_job {
	struct mdb_table *t;
	const int64_t *const *cols;
	int ncols;
	uint64_t row0, n;		/* the appended rows are [row0, row0 + n) of the store */
	int nthreads;			/* workers that exist (set before the gate opens) */
	pthread_mutex_t mu;		/* the gate: workers wait until the barriers are sized for the threads that could be started */
	pthread_cond_t cv;
	int open;
	pthread_barrier_t go, done;
	uint64_t chunks;
};

struct ingest_worker {
	struct ingest_job *job;
	int id;
};

static void *ingest_worker_main(void *arg)
{
	struct ingest_worker *w = arg;
	struct ingest_job *j = w->job;
	pthread_mutex_lock(&j->mu);
	while (!j->open)
		pthread_cond_wait(&j->cv, &j->mu);
	pthread_mutex_unlock(&j->mu);
	for (uint64_t k = 0; k < j->chunks; k++) {
		pthread_barrier_wait(&j->go);
		const uint64_t c0 = k * INGEST_CHUNK_ROWS, c1 = c0 + INGEST_CHUNK_ROWS < j->n ? c0 + INGEST_CHUNK_ROWS : j->n;
		const uint64_t per = (c1 - c0 + (uint64_t)j->nthreads - 1) / (uint64_t)j->nthreads;
		const uint64_t lo = c0 + per * (uint64_t)w->id < c1 ? c0 + per * (uint64_t)w->id : c1, hi = lo + per < c1 ? lo + per : c1;
		for (int c = 0; c < j->ncols && hi > lo; c++)
			memcpy(j->t->cols[c].data + j->row0 + lo, j->cols[c] + lo, (hi - lo) * 8);
		pthread_barrier_wait(&j->done);
	}
	return NULL;
}

/* n rows of ncols plain 8-byte columns (no NULL flags, no strings; validated by the caller, room reserved, the rows' NULL bits cleared) ->
 * the host store, and - when the database has a device and the mirror is current or absent - the device mirror, chunk by chunk.  Does NOT
 * touch t->nrows / t->generation (the caller's bookkeeping); *mirrored says whether the mirror holds the new rows. */
int mdb_table_bulk_copy(struct mdb_catalog *cat, struct mdb_table *t, int ncols, uint64_t n, const int64_t *const *cols, bool *mirrored)
{
	*mirrored = false;
	int nthreads = (int)sysconf(_SC_NPROCESSORS_ONLN);
	const char *env = mdb_knob("MDB_INGEST_THREADS");
	if (env && atoi(env) > 0)
		nthreads = atoi(env);
	nthreads = nthreads < 1 ? 1 : nthreads > INGEST_MAX_THREADS ? INGEST_MAX_THREADS : nthreads;
	/* the device mirror can follow when it is current (the new rows fit or it is rebuilt) - and no column has a NULL bitmap up there */
	char err[256];
	bool up = mdb_catalog_device(cat, err, sizeof(err)) == MIDORIDB_OK && !t->device_only && !cat->dist;
	for (int c = 0; c < ncols && up; c++)
		up = !t->cols[c].d_nullbits && !t->cols[c].null_count;
	if (up && t->dev_generation && !(t->dev_generation == t->generation && t->dev_rows == t->nrows))
		up = false;	/* (a stale mirror: the next SELECT brings it up to date as before) */
	if (up && (!t->dev_generation || t->nrows + n > t->dev_cap)) {
		/* no mirror yet, or no room: a new one at the store's capacity; the rows already there travel first */
		table_drop_device(t, cat->dev);
		t->dev_cap = t->cap > t->nrows + n ? t->cap : t->nrows + n;
		for (int c = 0; c < ncols && up; c++)
			if (mdb_dev_alloc(cat->dev, t->dev_cap * 8, &t->cols[c].d_data) ||
			    (t->nrows && mdb_dev_h2d(cat->dev, t->cols[c].d_data, t->cols[c].data, t->nrows * 8)))
				up = false;
		if (!up)
			table_drop_device(t, cat->dev);
	}
	struct ingest_job job;
	memset(&job, 0, sizeof(job));
	job.t = t;
	job.cols = cols;
	job.ncols = ncols;
	job.row0 = t->nrows;
	job.n = n;
	job.nthreads = nthreads;
	job.chunks = (n + INGEST_CHUNK_ROWS - 1) / INGEST_CHUNK_ROWS;
	pthread_t th[INGEST_MAX_THREADS];
	struct ingest_worker wk[INGEST_MAX_THREADS];
	int started = 0;
	pthread_mutex_init(&job.mu, NULL);
	pthread_cond_init(&job.cv, NULL);
	for (int i = 0; i < nthreads; i++) {
		wk[i].job = &job;
		wk[i].id = i;
		if (pthread_create(&th[i], NULL, ingest_worker_main, &wk[i]))
			break;
		started++;
	}
	if (!started) {		/* (no thread to be had: one plain copy, the upload is left to the next SELECT) */
		pthread_mutex_destroy(&job.mu);
		pthread_cond_destroy(&job.cv);
		for (int c = 0; c < ncols; c++)
			memcpy(t->cols[c].data + t->nrows, cols[c], n * 8);
		return MIDORIDB_OK;
	}
	nthreads = job.nthreads = started;
	pthread_barrier_init(&job.go, NULL, (unsigned)nthreads + 1);
	pthread_barrier_init(&job.done, NULL, (unsigned)nthreads + 1);
	pthread_mutex_lock(&job.mu);
	job.open = 1;
	pthread_cond_broadcast(&job.cv);
	pthread_mutex_unlock(&job.mu);
	for (uint64_t k = 0; k <= job.chunks; k++) {
		if (k < job.chunks)
			pthread_barrier_wait(&job.go);		/* the workers copy chunk k ... */
		if (k > 0 && up) {				/* ... while chunk k - 1 goes up */
			const uint64_t c0 = (k - 1) * INGEST_CHUNK_ROWS, c1 = c0 + INGEST_CHUNK_ROWS < n ? c0 + INGEST_CHUNK_ROWS : n;
			for (int c = 0; c < ncols && up; c++)
				if (mdb_dev_h2d(cat->dev, (char *)t->cols[c].d_data + (t->nrows + c0) * 8, t->cols[c].data + t->nrows + c0, (c1 - c0) * 8))
					up = false;
		}
		if (k < job.chunks)
			pthread_barrier_wait(&job.done);
	}
	for (int i = 0; i < nthreads; i++)
		pthread_join(th[i], NULL);
	pthread_barrier_destroy(&job.go);
	pthread_barrier_destroy(&job.done);
	pthread_mutex_destroy(&job.mu);
	pthread_cond_destroy(&job.cv);
	/* (an upload that failed half-way leaves the mirror valid for the old rows: the next SELECT uploads the tail) */
	*mirrored = up;
	return MIDORIDB_OK;
}

/* after the caller's bookkeeping (t->nrows, t->generation advanced by a bulk copy that reached the mirror): the mirror is current, and the
 * statistics of the integer-like columns follow from the uploaded tail */
void mdb_table_bulk_mirrored(struct mdb_catalog *cat, struct mdb_table *t, uint64_t old_rows, uint64_t old_generation)
{
	const bool had = t->dev_generation == old_generation && t->dev_generation != 0;
	for (int c = 0; c < t->ncols; c++) {
		struct mdb_column *col = &t->cols[c];
		if (!mdb_col_has_range(col) || !col->d_data)
			continue;
		const bool widen = (had || old_rows == 0) && (old_rows == 0 || col->st_generation == old_generation + 1);
		int64_t lo = 0, hi = -1;
		const uint64_t r0 = widen ? old_rows : 0;
		if (mdb_dev_key_range(cat->dev, (const int64_t *)col->d_data + r0, NULL, t->nrows - r0, &lo, &hi)) {
			col->st_generation = 0;
			continue;
		}
		if (widen && old_rows && col->st_lo <= col->st_hi) {
			lo = lo <= hi && lo < col->st_lo ? lo : col->st_lo;
			hi = hi > col->st_hi ? hi : col->st_hi;
		}
		col->st_lo = lo;
		col->st_hi = hi;
		col->st_generation = t->generation + 1;
		col_distinct_update(cat, t, col, widen && old_rows && col->dv_generation == old_generation + 1, old_rows);
	}
	t->dev_generation = t->generation;
	t->dev_rows = t->nrows;
}


/* ------------------------------------------------------------------ "no key twice" as a catalog statistic (round 6)
 *
 * Measured, never declared: one scattered atomic per row into a bitmap of the column's key window (mdb_dev_distinct_scan, ~1.3 ms per
 * 10^8 rows) - at ingest, where the range is taken anyway.  Appended rows are scanned into the SAME bitmap (kept on the device while it
 * is at most DV_KEEP_BYTES; sized for twice the window so that a growing key does not outrun it at once); a column that has shown a value
 * twice stays so while it only grows; everything else (UPDATE, DELETE, a window outgrown) is looked at whole when somebody asks.
 * rows_from = first row not yet in the bitmap when `follow` (the verdict of the generation before this one is known), else ignored. */
#define DV_MIN_ROWS ((uint64_t)1 << 20)		/* below: no operator form asks (the bit-per-row forms start at 2^21 - 2^22 rows) */
#define DV_KEEP_BYTES ((uint64_t)64 << 20)
#define DV_MAX_BYTES ((uint64_t)1 << 30)

static void col_distinct_forget(struct mdb_catalog *cat, struct mdb_column *col)
{
	if (col->d_seen && cat->dev)
		mdb_dev_free(cat->dev, col->d_seen);
	col->d_seen = NULL;
	col->dv_generation = 0;
}

static void col_distinct_update(struct mdb_catalog *cat, struct mdb_table *t, struct mdb_column *col, bool follow, uint64_t rows_from)
{
	const uint64_t rows = t->device_only ? t->dev_rows : t->nrows;
	const bool was_distinct = follow && col->dv_distinct;
	if (follow && !col->dv_distinct) {	/* a value twice: appended rows do not change that */
		col->dv_generation = t->generation + 1;
		return;
	}
	col->dv_generation = 0;
	if (!mdb_col_has_range(col) || !col->d_data || !cat->dev || col->st_generation != t->generation + 1 ||
	    (rows < DV_MIN_ROWS && !col->declared_unique)) {
		col_distinct_forget(cat, col);
		return;
	}
	if (col->st_lo > col->st_hi) {		/* no non-NULL value at all */
		col_distinct_forget(cat, col);
		col->dv_distinct = true;
		col->dv_generation = t->generation + 1;
		return;
	}
	const uint64_t span = (uint64_t)col->st_hi - (uint64_t)col->st_lo;	/* (values - 1) */
	int twice = 0;
	if (was_distinct && col->d_seen && col->st_lo >= col->seen_lo && (uint64_t)col->st_hi - (uint64_t)col->seen_lo < col->seen_bits && rows_from <= rows) {
		/* the appended rows against the values seen so far */
		const uint64_t a0 = col->d_nullbits ? (rows_from & ~(uint64_t)63) : rows_from;	/* (a NULL bitmap is read by row index: whole words) */
		if (a0 != rows_from) {
			/* (rows a0 .. rows_from - 1 are in the bitmap already and would meet themselves: the column is looked at whole) */
			col_distinct_forget(cat, col);
		} else if (mdb_dev_distinct_scan(cat->dev, (const int64_t *)col->d_data + a0, col->d_nullbits ? col->d_nullbits + a0 / 64 : NULL, rows - a0,
						 col->seen_lo, col->seen_bits, col->d_seen, &twice)) {
			col_distinct_forget(cat, col);
			return;
		} else {
			col->dv_distinct = !twice;
			col->dv_generation = t->generation + 1;
			if (twice)
				col_distinct_forget(cat, col), col->dv_generation = t->generation + 1;
			return;
		}
	}
	/* the whole column into a fresh bitmap: worth it when the window is not much wider than the table is long (or the DDL said UNIQUE) */
	if (span >= ((uint64_t)1 << 33) || (span / 16 > rows && !col->declared_unique)) {
		col_distinct_forget(cat, col);
		return;
	}
	uint64_t bits = ((2 * (span + 1) + 65535) & ~(uint64_t)65535);
	if (bits / 8 > DV_KEEP_BYTES)
		bits = (span + 1 + 65535) & ~(uint64_t)65535;	/* (not kept: no room to grow needed) */
	if (bits / 8 > DV_MAX_BYTES) {
		col_distinct_forget(cat, col);
		return;
	}
	col_distinct_forget(cat, col);
	void *bm = NULL;
	if (mdb_dev_alloc(cat->dev, bits / 8, &bm) || mdb_dev_memset(cat->dev, bm, 0, bits / 8) ||
	    mdb_dev_distinct_scan(cat->dev, (const int64_t *)col->d_data, col->d_nullbits, rows, col->st_lo, bits, (uint32_t *)bm, &twice)) {
		if (bm)
			mdb_dev_free(cat->dev, bm);
		return;
	}
	col->dv_distinct = !twice;
	col->dv_generation = t->generation + 1;
	if (twice || bits / 8 > DV_KEEP_BYTES) {
		mdb_dev_free(cat->dev, bm);
	} else {
		col->d_seen = (uint32_t *)bm;
		col->seen_lo = col->st_lo;
		col->seen_bits = bits;
	}
}

bool mdb_col_distinct(struct mdb_catalog *cat, struct mdb_table *t, struct mdb_column *col)
{
	int64_t lo, hi;
	if (!mdb_col_has_range(col))
		return false;
	if (col->dv_generation != t->generation + 1) {
		if (mdb_col_range(cat, t, col, &lo, &hi) != MIDORIDB_OK)	/* (the window comes from the range) */
			return false;
		col_distinct_update(cat, t, col, false, 0);
	}
	return col->dv_generation == t->generation + 1 && col->dv_distinct;
}

bool mdb_col_has_range(const struct mdb_column *col)
{
	return col->type == MDB_CT_INTEGER || col->type == MDB_CT_DATE || col->type == MDB_CT_DATETIME || col->type == MDB_CT_TINYINT;
}

/* The column's smallest / largest non-NULL value over the device mirror as it stands (lo > hi: none) - kept current by
 * mdb_table_sync_device() for tables that only grow, computed here (one pass on the device) after a DELETE / UPDATE or for a table that
 * was generated on the device.  A SUPERSET of the live values is what the callers need and what a filtered stream of the column gets. */
int mdb_col_range(struct mdb_catalog *cat, struct mdb_table *t, struct mdb_column *col, int64_t *lo, int64_t *hi)
{
	if (!mdb_col_has_range(col))
		return 1;
	const uint64_t rows = t->device_only ? t->dev_rows : t->nrows;
	if (col->st_generation != t->generation + 1) {
		int64_t l = 0, h = -1;
		if (rows && (!col->d_data || mdb_dev_key_range(cat->dev, col->d_data, col->d_nullbits, rows, &l, &h)))
			return -MIDORIDB_INTERNAL;
		col->st_lo = l;
		col->st_hi = h;
		col->st_generation = t->generation + 1;
	}
	*lo = col->st_lo;
	*hi = col->st_hi;
	return MIDORIDB_OK;
}

/* ------------------------------------------------------------------ result column order (R3)
 *
 * The reference creates the result ("early_mat_tbl") columns by iterating a chained hash table of
 * column names (reference src/engine/executor_select.c:293-322; src/datastructure/hashtable.c):
 * djb2 over strlen+1 bytes (the NUL included, :269-281), capacity 16, doubled when count/capacity
 * >= 0.5 after a put (:84-129; old buckets are re-inserted in bucket order, each chain from its
 * head), list_add() inserts at the chain head (:172), iteration = buckets in order, chains from
 * the head (:242-259).  This function replays that on the key sequence and returns, for each
 * iteration position, the index of the key.
 */
static size_t djb2_with_nul(const char *s)
{
	size_t h = 5381;
	size_t len = strlen(s) + 1;
	for (size_t i = 0; i < len; i++)
		h = ((h << 5) + h) + (size_t)(signed char)s[i];
	return h;
}

int mdb_reference_column_order(const char (*keys)[MDB_NAME_LEN], int nkeys, int *order_out)
{
	size_t cap = 16;
	int count = 0;
	/* bucket chains as arrays, index 0 = head */
	int **bucket;
	int *blen;
	int rc = MIDORIDB_OK;

	bucket = calloc(cap, sizeof(int *));
	blen = calloc(cap, sizeof(int));
	if (!bucket || !blen)
		return -MIDORIDB_NOMEM;
	for (int k = 0; k < nkeys; k++) {
		size_t b = djb2_with_nul(keys[k]) % cap;
		int *nb = realloc(bucket[b], sizeof(int) * (size_t)(blen[b] + 1));
		if (!nb) {
			rc = -MIDORIDB_NOMEM;
			goto out;
		}
		bucket[b] = nb;
		memmove(nb + 1, nb, sizeof(int) * (size_t)blen[b]);	/* list_add: new entry becomes the head */
		nb[0] = k;
		blen[b]++;
		count++;
		if ((double)count / (double)cap >= 0.5) {
			size_t ncap = cap * 2;
			int **nbk = calloc(ncap, sizeof(int *));
			int *nlen = calloc(ncap, sizeof(int));
			if (!nbk || !nlen) {
				free(nbk);
				free(nlen);
				rc = -MIDORIDB_NOMEM;
				goto out;
			}
			for (size_t i = 0; i < cap; i++) {
				for (int e = 0; e < blen[i]; e++) {	/* chain from its head */
					int key = bucket[i][e];
					size_t d = djb2_with_nul(keys[key]) % ncap;
					int *x = realloc(nbk[d], sizeof(int) * (size_t)(nlen[d] + 1));
					if (!x) {
						rc = -MIDORIDB_NOMEM;
						for (size_t z = 0; z < ncap; z++)
							free(nbk[z]);
						free(nbk);
						free(nlen);
						goto out;
					}
					nbk[d] = x;
					memmove(x + 1, x, sizeof(int) * (size_t)nlen[d]);
					x[0] = key;
					nlen[d]++;
				}
				free(bucket[i]);
			}
			free(bucket);
			free(blen);
			bucket = nbk;
			blen = nlen;
			cap = ncap;
		}
	}
	{
		int pos = 0;
		for (size_t i = 0; i < cap; i++)
			for (int e = 0; e < blen[i]; e++)
				order_out[pos++] = bucket[i][e];
	}
out:
	for (size_t i = 0; i < cap; i++)
		free(bucket[i]);
	free(bucket);
	free(blen);
	return rc;
}
