/*
 * examples/readme_example.c - an application written against the reference's public C API
 * (include/engine/query.h + database.h upstream; the usage pattern of its README.md:49-81), built against
 * this repository's drop-in library instead.  Only the include line and the link line differ:
 *
 *     gcc -Iinclude examples/readme_example.c -Lmidoridb_amd -lmidoridb_amd -Wl,-rpath,$PWD/midoridb_amd -o readme_example
 *
 * It fills two tables with SQL text, runs the README's join + GROUP BY + COUNT(*) query on the MI355X path and
 * walks the result with the cursor functions.  Exit code 0 and the lines "id_a: 1, count: 2" ... are what
 * tests/test_query_gpu.py::test_c_program_against_the_drop_in_library checks.
 */
#include <stdio.h>
#include <engine/query.h>	/* the reference's include line: resolved by include/engine/query.h (forwarding header) */

static int run(struct database *db, char *sql)
{
	struct query_output *out = query_execute(db, sql);
	int ok = out && out->status != ST_ERROR;
	if (out && out->status == ST_ERROR)
		fprintf(stderr, "error: %s", out->error.message);
	if (out)
		query_free(out);
	return ok;
}

int main(void)
{
	struct database db = {0};
	struct query_output *output;

	if (database_open(&db) != MIDORIDB_OK)
		return 1;
	if (!run(&db, "CREATE TABLE A (id_a INT);") || !run(&db, "CREATE TABLE B (id_b INT);") ||
	    !run(&db, "INSERT INTO A VALUES (1), (3), (4);") || !run(&db, "INSERT INTO B VALUES (1), (1), (3), (3), (4), (NULL);"))
		return 2;

	output = query_execute(&db, "SELECT "
				    "    id_a, COUNT(*) "
				    "FROM "
				    "    A INNER JOIN B "
				    "    ON A.id_a = B.id_b "
				    "GROUP BY "
				    "    id_a;");
	if (!output || output->status != ST_OK_WITH_RESULTS) {
		if (output)
			fprintf(stderr, "error: %s", output->error.message);
		return 3;
	}
	while (query_cur_step(&output->results) == MIDORIDB_ROW)
		printf("id_a: %ld, count: %ld\n", (long)query_column_int64(&output->results, 0),
		       (long)query_column_int64(&output->results, 1));
	query_free(output);

	/* the statements SURVEY 8f ranks next run through the same entry point */
	if (!run(&db, "DELETE FROM B WHERE id_b = 3;") || !run(&db, "UPDATE A SET id_a = 5 WHERE id_a = 4;"))
		return 4;
	output = query_execute(&db, "SELECT id_a, COUNT(*) FROM A INNER JOIN B ON A.id_a = B.id_b GROUP BY id_a;");
	if (!output || output->status != ST_OK_WITH_RESULTS)
		return 5;
	while (query_cur_step(&output->results) == MIDORIDB_ROW)
		printf("after DML: id_a: %ld, count: %ld\n", (long)query_column_int64(&output->results, 0),
		       (long)query_column_int64(&output->results, 1));
	query_free(output);
	database_close(&db);
	return 0;
}
