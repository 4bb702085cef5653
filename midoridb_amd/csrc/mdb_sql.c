/*
 * mdb_sql.c - hand-written SQL-subset front end.
 *
 * The reference front end is a flex lexer + bison grammar (reference
 * src/parser/midorisql.l, src/parser/midorisql.y) whose only product is a FIFO of
 * RPN token strings (emit(), midorisql.y:517-528) consumed by the AST builder.
 * Neither tool exists in this image or on the GPU box, and the front end is out of
 * scope for this build - a MidoriDB maintainer keeps the
 * bison/flex parser and hands its queue to mdb_query_execute_rpn().
 * This file exists so that SQL text works without bison: a small precedence-climbing
 * parser that emits the SAME token vocabulary in the SAME order as the grammar's
 * reductions, for the statements the SELECT path and its tests need:
 *
 *   SELECT [DISTINCT] exprs FROM refs [WHERE] [GROUP BY] [HAVING] [ORDER BY] [LIMIT] ;
 *   CREATE TABLE [IF NOT EXISTS] t (col type [attrs], ...) ;
 *   INSERT [INTO] t [(cols)] VALUES (...), (...) ;
 *
 * Operator precedence follows midorisql.y:48-63; literal lexing follows
 * midorisql.l:83-92 (a '-' directly in front of digits belongs to the number).
 * The emitted queue is accepted unchanged by the reference's ast_build_tree()
 * (checked in tests/test_sql_frontend.py against oracle/_ref).
 */
#include "mdb_host.h"
#include <ctype.h>

enum tk {
	T_EOF, T_NAME, T_STRING, T_INT, T_FLOAT, T_BOOL, T_CMP, T_PUNCT, T_KW, T_FCOUNT, T_ANDOP, T_OROP, T_ERR
};

static const char *const KEYWORDS[] = {
	"AND", "AS", "ASC", "AUTO_INCREMENT", "BY", "CREATE", "DATE", "DATETIME", "DELETE", "DESC", "DISTINCT",
	"DOUBLE", "EXISTS", "FROM", "GROUP", "HAVING", "IF", "IN", "INDEX", "INNER", "INSERT", "INT",
	"INT4", "INTEGER", "INTO", "IS", "JOIN", "KEY", "LEFT", "LIKE", "LIMIT", "MOD", "NOT", "NULL",
	"ON", "OR", "ORDER", "OUTER", "PRIMARY", "RIGHT", "SELECT", "SET", "TABLE", "TINYINT", "UNIQUE", "UPDATE",
	"VALUE", "VALUES", "VARCHAR", "VARCHARACTER", "WHERE", "XOR", NULL
};

struct lexer {
	const char *s;
	size_t pos;
	enum tk type;
	char text[256];		/* NAME / STRING (with quotes) / keyword upper-cased / punct */
	long ival;
	double fval;
	int sub;		/* comparison code */
};

struct parser {
	struct lexer lx;
	struct mdb_rpn *out;
	char *err;
	size_t errlen;
	bool failed;
	bool dml;		/* inside delete_expr / update_expr: names, literals, logic, comparisons, IS NULL, IN only */
};

static void fail(struct parser *p, const char *fmt, ...)
{
	va_list ap;
	if (p->failed)
		return;
	p->failed = true;
	va_start(ap, fmt);
	if (p->err && p->errlen)
		vsnprintf(p->err, p->errlen, fmt, ap);
	va_end(ap);
}

static void emit(struct parser *p, const char *fmt, ...)
{
	char buf[256];
	va_list ap;
	if (p->failed)
		return;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	if (mdb_rpn_push(p->out, buf))
		fail(p, "out of memory");
}

static bool is_keyword(const char *up)
{
	for (int i = 0; KEYWORDS[i]; i++)
		if (strcmp(KEYWORDS[i], up) == 0)
			return true;
	return false;
}

static void next(struct parser *p)
{
	struct lexer *l = &p->lx;
	const char *s = l->s;
	size_t i = l->pos;

	for (;;) {
		while (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')
			i++;
		if (s[i] == '#' || (s[i] == '-' && s[i + 1] == '-' && (s[i + 2] == ' ' || s[i + 2] == '\t'))) {
			while (s[i] && s[i] != '\n')
				i++;
			continue;
		}
		if (s[i] == '/' && s[i + 1] == '*') {
			i += 2;
			while (s[i] && !(s[i] == '*' && s[i + 1] == '/'))
				i++;
			if (s[i])
				i += 2;
			continue;
		}
		break;
	}
	l->text[0] = 0;
	if (!s[i]) {
		l->type = T_EOF;
		l->pos = i;
		return;
	}
	/* numbers: -?[0-9]+ | -?[0-9]+"."[0-9]* | -?"."[0-9]+ | exponent forms (midorisql.l:85-92) */
	{
		size_t j = i;
		bool isnum = false, isfloat = false;
		if (s[j] == '-')
			j++;
		if (isdigit((unsigned char)s[j])) {
			isnum = true;
			while (isdigit((unsigned char)s[j]))
				j++;
			if (s[j] == '.') {
				isfloat = true;
				j++;
				while (isdigit((unsigned char)s[j]))
					j++;
			}
		} else if (s[j] == '.' && isdigit((unsigned char)s[j + 1])) {
			isnum = isfloat = true;
			j++;
			while (isdigit((unsigned char)s[j]))
				j++;
		}
		if (isnum && (s[j] == 'E' || s[j] == 'e')) {
			size_t k = j + 1;
			if (s[k] == '+' || s[k] == '-')
				k++;
			if (isdigit((unsigned char)s[k])) {
				while (isdigit((unsigned char)s[k]))
					k++;
				j = k;
				isfloat = true;
			}
		}
		if (isnum) {
			char tmp[128];
			size_t len = j - i < sizeof(tmp) - 1 ? j - i : sizeof(tmp) - 1;
			memcpy(tmp, s + i, len);
			tmp[len] = 0;
			if (isfloat) {
				l->type = T_FLOAT;
				l->fval = atof(tmp);
			} else {
				l->type = T_INT;
				l->ival = atoi(tmp);	/* 32-bit, as the reference lexer (midorisql.l:85) */
			}
			l->pos = j;
			return;
		}
	}
	if (s[i] == '\'' || s[i] == '"') {
		char q = s[i];
		size_t j = i + 1;
		while (s[j] && s[j] != '\n') {
			if (s[j] == '\\' && s[j + 1]) {
				j += 2;
				continue;
			}
			if (s[j] == q) {
				if (s[j + 1] == q) {
					j += 2;
					continue;
				}
				break;
			}
			j++;
		}
		if (s[j] != q) {
			l->type = T_ERR;
			fail(p, "Unterminated string");
			return;
		}
		j++;
		{
			size_t len = j - i < sizeof(l->text) - 1 ? j - i : sizeof(l->text) - 1;
			memcpy(l->text, s + i, len);
			l->text[len] = 0;
		}
		l->type = T_STRING;
		l->pos = j;
		return;
	}
	if (s[i] == '`') {
		size_t j = i + 1, len;
		while (s[j] && s[j] != '`' && s[j] != '\n')
			j++;
		if (s[j] != '`') {
			l->type = T_ERR;
			fail(p, "unterminated quoted name");
			return;
		}
		len = j - i - 1 < sizeof(l->text) - 1 ? j - i - 1 : sizeof(l->text) - 1;
		memcpy(l->text, s + i + 1, len);
		l->text[len] = 0;
		l->type = T_NAME;
		l->pos = j + 1;
		return;
	}
	if (isalpha((unsigned char)s[i])) {
		size_t j = i, len;
		char up[256];
		while (isalnum((unsigned char)s[j]) || s[j] == '_')
			j++;
		len = j - i < sizeof(l->text) - 1 ? j - i : sizeof(l->text) - 1;
		memcpy(l->text, s + i, len);
		l->text[len] = 0;
		for (size_t k = 0; k <= len; k++)
			up[k] = (char)toupper((unsigned char)l->text[k]);
		l->pos = j;
		if (strcmp(up, "COUNT") == 0 && s[j] == '(') {	/* midorisql.l:139-142 */
			l->type = T_FCOUNT;
			return;
		}
		if (strcmp(up, "TRUE") == 0 || strcmp(up, "FALSE") == 0 || strcmp(up, "UNKNOWN") == 0) {
			l->type = T_BOOL;
			l->ival = up[0] == 'T' ? 1 : (up[0] == 'F' ? 0 : -1);
			return;
		}
		if (is_keyword(up)) {
			l->type = T_KW;
			strcpy(l->text, up);
			return;
		}
		l->type = T_NAME;
		return;
	}
	/* operators */
	l->pos = i + 1;
	l->type = T_PUNCT;
	l->text[0] = s[i];
	l->text[1] = 0;
	switch (s[i]) {
	case '&':
		if (s[i + 1] == '&') {
			l->pos = i + 2;
			l->type = T_ANDOP;
		}
		return;
	case '|':
		if (s[i + 1] == '|') {
			l->pos = i + 2;
			l->type = T_OROP;
		}
		return;
	case '=':
		l->type = T_CMP;
		l->sub = 4;
		return;
	case '>':
		l->type = T_CMP;
		if (s[i + 1] == '=') {
			l->sub = 6;
			l->pos = i + 2;
		} else {
			l->sub = 2;
		}
		return;
	case '<':
		l->type = T_CMP;
		if (s[i + 1] == '=') {
			l->sub = 5;
			l->pos = i + 2;
		} else if (s[i + 1] == '>') {
			l->sub = 3;
			l->pos = i + 2;
		} else {
			l->sub = 1;
		}
		return;
	case '!':
		if (s[i + 1] == '=') {
			l->type = T_CMP;
			l->sub = 3;
			l->pos = i + 2;
		}
		return;
	case '-': case '+': case '*': case '/': case '%': case '(': case ')': case ',': case '.': case ';':
		return;
	default:
		l->type = T_ERR;
		fail(p, "mystery character '%c'", s[i]);
		return;
	}
}

static bool is_kw(struct parser *p, const char *kw)
{
	return p->lx.type == T_KW && strcmp(p->lx.text, kw) == 0;
}

static bool is_punct(struct parser *p, char c)
{
	return p->lx.type == T_PUNCT && p->lx.text[0] == c;
}

static bool accept_kw(struct parser *p, const char *kw)
{
	if (is_kw(p, kw)) {
		next(p);
		return true;
	}
	return false;
}

static void expect_punct(struct parser *p, char c)
{
	if (!is_punct(p, c)) {
		fail(p, "syntax error, unexpected '%s', expecting '%c'", p->lx.type == T_EOF ? "end of input" : p->lx.text, c);
		return;
	}
	next(p);
}

static void expect_kw(struct parser *p, const char *kw)
{
	if (!accept_kw(p, kw))
		fail(p, "syntax error, unexpected '%s', expecting %s", p->lx.type == T_EOF ? "end of input" : p->lx.text, kw);
}

/* precedence levels (midorisql.y:48-63), low to high */
enum { P_OR = 1, P_XOR = 2, P_AND = 3, P_ISIN = 4, P_NOT = 5, P_CMP = 7, P_ADD = 11, P_MUL = 12, P_NEG = 14 };

static void parse_expr(struct parser *p, int min_prec);

static int parse_val_list(struct parser *p)
{
	int n = 0;
	do {
		parse_expr(p, P_OR);
		n++;
	} while (!p->failed && is_punct(p, ',') && (next(p), true));
	return n;
}

static void parse_primary(struct parser *p)
{
	struct lexer *l = &p->lx;

	if (p->failed)
		return;
	switch (l->type) {
	case T_NAME: {
		char first[256];
		strcpy(first, l->text);
		next(p);
		if (is_punct(p, '.') && !p->dml) {
			next(p);
			if (p->lx.type != T_NAME) {
				fail(p, "syntax error, expecting column name after '.'");
				return;
			}
			emit(p, "FIELDNAME %s.%s", first, p->lx.text);
			next(p);
		} else {
			emit(p, "NAME %s", first);
		}
		return;
	}
	case T_STRING:
		emit(p, "STRING %s", l->text);
		next(p);
		return;
	case T_INT:
		emit(p, "NUMBER %d", (int)l->ival);
		next(p);
		return;
	case T_FLOAT:
		emit(p, "FLOAT %g", l->fval);
		next(p);
		return;
	case T_BOOL:
		emit(p, "BOOL %d", (int)l->ival);
		next(p);
		return;
	case T_FCOUNT:
		if (p->dml)
			break;
		next(p);
		expect_punct(p, '(');
		if (is_punct(p, '*')) {
			next(p);
			expect_punct(p, ')');
			emit(p, "COUNTALL");
		} else {
			parse_expr(p, P_OR);
			expect_punct(p, ')');
			emit(p, "COUNTFIELD");
		}
		return;
	case T_KW:
		if (strcmp(l->text, "NULL") == 0) {
			emit(p, "NULL");
			next(p);
			return;
		}
		break;
	case T_PUNCT:
		if (l->text[0] == '(') {
			next(p);
			parse_expr(p, P_OR);
			expect_punct(p, ')');
			return;
		}
		if (l->text[0] == '-' && !p->dml) {
			next(p);
			parse_expr(p, P_NEG);
			emit(p, "NEG");
			return;
		}
		break;
	default:
		break;
	}
	fail(p, "syntax error, unexpected '%s'", l->type == T_EOF ? "end of input" : l->text);
}

static void parse_expr(struct parser *p, int min_prec)
{
	parse_primary(p);
	while (!p->failed) {
		struct lexer *l = &p->lx;
		int prec;
		char op[16];

		if (l->type == T_OROP || is_kw(p, "OR")) {
			prec = P_OR;
			strcpy(op, "OR");
		} else if (is_kw(p, "XOR")) {
			prec = P_XOR;
			strcpy(op, "XOR");
		} else if (l->type == T_ANDOP || is_kw(p, "AND")) {
			prec = P_AND;
			strcpy(op, "AND");
		} else if (l->type == T_CMP) {
			prec = P_CMP;
			snprintf(op, sizeof(op), "CMP %d", l->sub);
		} else if (!p->dml && (is_punct(p, '+') || is_punct(p, '-'))) {
			prec = P_ADD;
			strcpy(op, l->text[0] == '+' ? "ADD" : "SUB");
		} else if (!p->dml && (is_punct(p, '*') || is_punct(p, '/') || is_punct(p, '%') || is_kw(p, "MOD"))) {
			prec = P_MUL;
			strcpy(op, l->text[0] == '*' ? "MUL" : (l->text[0] == '/' ? "DIV" : "MOD"));
		} else if (is_kw(p, "IS")) {
			bool neg = false;
			if (P_ISIN < min_prec)
				return;
			next(p);
			if (accept_kw(p, "NOT"))
				neg = true;
			expect_kw(p, "NULL");
			emit(p, neg ? "ISNOTNULL" : "ISNULL");
			continue;
		} else if (is_kw(p, "IN") || is_kw(p, "LIKE") || is_kw(p, "NOT")) {
			bool neg = false;
			int n;
			if (P_ISIN < min_prec)
				return;
			if (accept_kw(p, "NOT"))
				neg = true;
			if (accept_kw(p, "IN")) {
				expect_punct(p, '(');
				n = parse_val_list(p);
				expect_punct(p, ')');
				emit(p, neg ? "ISNOTIN %d" : "ISIN %d", n);
			} else if (!p->dml && accept_kw(p, "LIKE")) {
				parse_expr(p, P_ISIN + 1);
				emit(p, neg ? "NOTLIKE" : "LIKE");
			} else {
				fail(p, "syntax error, unexpected NOT");
			}
			continue;
		} else {
			return;
		}
		if (prec < min_prec)
			return;
		next(p);
		parse_expr(p, prec + 1);	/* all binary operators are left-associative */
		emit(p, "%s", op);
	}
}

/* opt_as_alias: AS NAME | NAME | nil  (midorisql.y:224-227) */
static void parse_opt_alias(struct parser *p)
{
	if (accept_kw(p, "AS")) {
		if (p->lx.type != T_NAME) {
			fail(p, "syntax error, expecting alias name after AS");
			return;
		}
		emit(p, "ALIAS %s", p->lx.text);
		next(p);
	} else if (p->lx.type == T_NAME) {
		emit(p, "ALIAS %s", p->lx.text);
		next(p);
	}
}

static void parse_table_factor(struct parser *p)
{
	if (p->lx.type != T_NAME) {
		fail(p, "syntax error, unexpected '%s', expecting table name", p->lx.type == T_EOF ? "end of input" : p->lx.text);
		return;
	}
	emit(p, "TABLE %s", p->lx.text);
	next(p);
	parse_opt_alias(p);
}

/* table_reference: table_factor | join_table (left-recursive)  (midorisql.y:213-234) */
static void parse_table_reference(struct parser *p)
{
	parse_table_factor(p);
	while (!p->failed) {
		int jt;
		if (is_kw(p, "INNER") || is_kw(p, "JOIN")) {
			accept_kw(p, "INNER");
			jt = 1;
		} else if (is_kw(p, "LEFT") || is_kw(p, "RIGHT")) {
			jt = is_kw(p, "LEFT") ? 2 : 4;
			next(p);
			if (accept_kw(p, "OUTER"))
				jt += 6;
		} else {
			return;
		}
		expect_kw(p, "JOIN");
		parse_table_factor(p);
		expect_kw(p, "ON");
		parse_expr(p, P_OR);
		emit(p, "ONEXPR");
		emit(p, "JOIN %d", jt);
	}
}

static void parse_select(struct parser *p)
{
	int opts = 0, nsel = 0, ntab = 0, extra = 0;

	while (accept_kw(p, "DISTINCT")) {
		if (opts & 2)
			fail(p, "duplicate DISTINCT option");
		opts |= 2;
	}
	if (is_punct(p, '*')) {
		next(p);
		emit(p, "SELECTALL");
		nsel = 1;
	} else {
		do {
			parse_expr(p, P_OR);
			parse_opt_alias(p);
			nsel++;
		} while (!p->failed && is_punct(p, ',') && (next(p), true));
	}
	if (!accept_kw(p, "FROM")) {
		emit(p, "SELECT %d %d", opts, nsel);
		return;
	}
	do {
		parse_table_reference(p);
		ntab++;
	} while (!p->failed && is_punct(p, ',') && (next(p), true));
	if (accept_kw(p, "WHERE")) {
		parse_expr(p, P_OR);
		emit(p, "WHERE");
		extra++;
	}
	if (accept_kw(p, "GROUP")) {
		int n = 0;
		expect_kw(p, "BY");
		do {
			parse_expr(p, P_OR);
			if (!accept_kw(p, "ASC"))
				accept_kw(p, "DESC");
			n++;
		} while (!p->failed && is_punct(p, ',') && (next(p), true));
		emit(p, "GROUPBYLIST %d", n);
		extra++;
	}
	if (accept_kw(p, "HAVING")) {
		parse_expr(p, P_OR);
		emit(p, "HAVING");
		extra++;
	}
	if (accept_kw(p, "ORDER")) {
		int n = 0;
		expect_kw(p, "BY");
		do {
			int desc = 0;
			parse_expr(p, P_OR);
			if (accept_kw(p, "DESC"))
				desc = 1;
			else
				accept_kw(p, "ASC");
			emit(p, "ORDERBYITEM %d", desc);
			n++;
		} while (!p->failed && is_punct(p, ',') && (next(p), true));
		emit(p, "ORDERBYLIST %d", n);
		extra++;
	}
	if (accept_kw(p, "LIMIT")) {
		parse_expr(p, P_OR);
		if (is_punct(p, ',')) {
			next(p);
			parse_expr(p, P_OR);
			emit(p, "LIMIT 2");
		} else {
			emit(p, "LIMIT 1");
		}
		extra++;
	}
	emit(p, "SELECT %d %d", opts, nsel + ntab + extra);
}

/* CREATE TABLE (midorisql.y:447-483) */
static void parse_create(struct parser *p)
{
	int ifne = 0, ncols = 0;
	char tname[256];

	expect_kw(p, "TABLE");
	if (accept_kw(p, "IF")) {
		expect_kw(p, "NOT");
		expect_kw(p, "EXISTS");
		ifne = 1;
	}
	if (p->lx.type != T_NAME) {
		fail(p, "syntax error, expecting table name");
		return;
	}
	strcpy(tname, p->lx.text);
	next(p);
	expect_punct(p, '(');
	do {
		char cname[256];
		int code = 0;
		emit(p, "STARTCOL");
		if (p->lx.type != T_NAME) {
			fail(p, "syntax error, expecting column name");
			return;
		}
		strcpy(cname, p->lx.text);
		next(p);
		if (is_kw(p, "INT") || is_kw(p, "INT4") || is_kw(p, "INTEGER")) {
			code = 50000;	/* the lexer maps INT, INT4 and INTEGER to one token (midorisql.l:45) */
			next(p);
		} else if (accept_kw(p, "TINYINT")) {
			code = 60000;
		} else if (accept_kw(p, "DOUBLE")) {
			code = 80000;
		} else if (accept_kw(p, "DATE")) {
			code = 100000;
		} else if (accept_kw(p, "DATETIME")) {
			code = 110000;
		} else if (is_kw(p, "VARCHAR") || is_kw(p, "VARCHARACTER")) {
			next(p);
			expect_punct(p, '(');
			if (p->lx.type != T_INT) {
				fail(p, "syntax error, expecting VARCHAR length");
				return;
			}
			code = 130000 + (int)p->lx.ival;
			next(p);
			expect_punct(p, ')');
		} else {
			fail(p, "syntax error, unexpected '%s', expecting a data type", p->lx.text);
			return;
		}
		for (;;) {
			if (is_kw(p, "NOT")) {
				next(p);
				expect_kw(p, "NULL");
				emit(p, "ATTR NOTNULL");
			} else if (accept_kw(p, "NULL")) {
				/* no token (midorisql.y:469) */
			} else if (accept_kw(p, "AUTO_INCREMENT")) {
				emit(p, "ATTR AUTOINC");
			} else if (accept_kw(p, "UNIQUE")) {
				emit(p, "ATTR UNIQUEKEY");
			} else if (is_kw(p, "PRIMARY")) {
				next(p);
				expect_kw(p, "KEY");
				emit(p, "ATTR PRIKEY");
			} else {
				break;
			}
			if (p->failed)
				return;
		}
		emit(p, "COLUMNDEF %d %s", code, cname);
		ncols++;
	} while (!p->failed && is_punct(p, ',') && (next(p), true));
	expect_punct(p, ')');
	emit(p, "CREATE %d %d %s", ifne, ncols, tname);
}

/* insert_expr (midorisql.y:377-392): literals with + - * / % and unary minus */
static void parse_insert_expr(struct parser *p, int min_prec)
{
	struct lexer *l = &p->lx;

	if (p->failed)
		return;
	if (l->type == T_STRING) {
		emit(p, "STRING %s", l->text);
		next(p);
	} else if (l->type == T_INT) {
		emit(p, "NUMBER %d", (int)l->ival);
		next(p);
	} else if (l->type == T_FLOAT) {
		emit(p, "FLOAT %g", l->fval);
		next(p);
	} else if (l->type == T_BOOL) {
		emit(p, "BOOL %d", (int)l->ival);
		next(p);
	} else if (is_kw(p, "NULL")) {
		emit(p, "NULL");
		next(p);
	} else if (is_punct(p, '(')) {
		next(p);
		parse_insert_expr(p, P_ADD);
		expect_punct(p, ')');
	} else if (is_punct(p, '-')) {
		next(p);
		parse_insert_expr(p, P_NEG);
		emit(p, "NEG");
	} else {
		fail(p, "syntax error, unexpected '%s' in VALUES", l->type == T_EOF ? "end of input" : l->text);
		return;
	}
	while (!p->failed) {
		int prec;
		const char *op;
		if (is_punct(p, '+') || is_punct(p, '-')) {
			prec = P_ADD;
			op = p->lx.text[0] == '+' ? "ADD" : "SUB";
		} else if (is_punct(p, '*') || is_punct(p, '/') || is_punct(p, '%')) {
			prec = P_MUL;
			op = p->lx.text[0] == '*' ? "MUL" : (p->lx.text[0] == '/' ? "DIV" : "MOD");
		} else {
			return;
		}
		if (prec < min_prec)
			return;
		next(p);
		parse_insert_expr(p, prec + 1);
		emit(p, "%s", op);
	}
}

/* INSERT ... VALUES (midorisql.y:347-375) */
static void parse_insert(struct parser *p)
{
	char tname[256];
	int hascols = 0, ntuples = 0;

	accept_kw(p, "INTO");
	if (p->lx.type != T_NAME) {
		fail(p, "syntax error, expecting table name");
		return;
	}
	strcpy(tname, p->lx.text);
	next(p);
	if (is_punct(p, '(')) {
		int n = 0;
		next(p);
		do {
			if (p->lx.type != T_NAME) {
				fail(p, "syntax error, expecting column name");
				return;
			}
			emit(p, "COLUMN %s", p->lx.text);
			next(p);
			n++;
		} while (!p->failed && is_punct(p, ',') && (next(p), true));
		expect_punct(p, ')');
		emit(p, "INSERTCOLS %d", n);
		hascols = 1;
	}
	if (!accept_kw(p, "VALUES") && !accept_kw(p, "VALUE")) {
		fail(p, "syntax error, expecting VALUES");
		return;
	}
	do {
		int n = 0;
		expect_punct(p, '(');
		do {
			parse_insert_expr(p, P_ADD);
			n++;
		} while (!p->failed && is_punct(p, ',') && (next(p), true));
		expect_punct(p, ')');
		emit(p, "VALUES %d", n);
		ntuples++;
	} while (!p->failed && is_punct(p, ',') && (next(p), true));
	emit(p, "INSERTVALS %d %d %s", hascols, ntuples, tname);
}

/* DELETE FROM NAME [WHERE delete_expr]  (midorisql.y:309-343) */
static void parse_delete(struct parser *p)
{
	char tname[256];

	expect_kw(p, "FROM");
	if (p->failed)
		return;
	if (p->lx.type != T_NAME) {
		fail(p, "syntax error, expecting table name");
		return;
	}
	strcpy(tname, p->lx.text);
	next(p);
	if (accept_kw(p, "WHERE")) {
		p->dml = true;
		parse_expr(p, P_OR);
		p->dml = false;
		emit(p, "WHERE");
	}
	emit(p, "DELETEONE %s", tname);
}

/* UPDATE NAME SET NAME = update_expr [, ...] [WHERE update_expr]  (midorisql.y:393-440) */
static void parse_update(struct parser *p)
{
	char tname[256], cname[256];
	int nassign = 0, haswhere = 0;

	if (p->lx.type != T_NAME) {
		fail(p, "syntax error, expecting table name");
		return;
	}
	strcpy(tname, p->lx.text);
	next(p);
	expect_kw(p, "SET");
	do {
		if (p->failed)
			return;
		if (p->lx.type != T_NAME) {
			fail(p, "syntax error, expecting column name");
			return;
		}
		strcpy(cname, p->lx.text);
		next(p);
		if (p->lx.type != T_CMP) {
			fail(p, "syntax error, expecting '='");
			return;
		}
		if (p->lx.sub != 4) {
			fail(p, "bad insert assignment to %s", cname);	/* the reference's own wording, midorisql.y:404 */
			return;
		}
		next(p);
		/* update_expr is left-recursive over COMPARISON too; the right-hand side of an assignment binds
		 * tighter than a following comparison cannot occur before ',' or WHERE, so one full expression */
		p->dml = true;
		parse_expr(p, P_OR);
		p->dml = false;
		emit(p, "ASSIGN %s", cname);
		nassign++;
	} while (!p->failed && is_punct(p, ',') && (next(p), true));
	if (accept_kw(p, "WHERE")) {
		p->dml = true;
		parse_expr(p, P_OR);
		p->dml = false;
		emit(p, "WHERE");
		haswhere = 1;
	}
	emit(p, "UPDATE %s %d %d", tname, nassign, haswhere);
}

int mdb_sql_parse(const char *sql, struct mdb_rpn *out, char *err, size_t errlen)
{
	struct parser p = {0};

	p.lx.s = sql;
	p.out = out;
	p.err = err;
	p.errlen = errlen;
	if (err && errlen)
		err[0] = 0;
	next(&p);
	if (accept_kw(&p, "SELECT"))
		parse_select(&p);
	else if (accept_kw(&p, "CREATE"))
		parse_create(&p);
	else if (accept_kw(&p, "INSERT"))
		parse_insert(&p);
	else if (accept_kw(&p, "DELETE"))
		parse_delete(&p);
	else if (accept_kw(&p, "UPDATE"))
		parse_update(&p);
	else
		fail(&p, "syntax error, unexpected '%s'", p.lx.type == T_EOF ? "end of input" : p.lx.text);
	emit(&p, "STMT");
	if (!p.failed) {
		expect_punct(&p, ';');		/* stmt_list: stmt ';'  (midorisql.y:148) */
		if (!p.failed && p.lx.type != T_EOF)
			fail(&p, "syntax error, unexpected '%s' after ';'", p.lx.text);
	}
	if (p.failed) {
		mdb_rpn_free(out);
		return -MIDORIDB_ERROR;
	}
	return MIDORIDB_OK;
}

int mdb_rpn_push(struct mdb_rpn *r, const char *tok)
{
	if (r->n == r->cap) {
		int ncap = r->cap ? r->cap * 2 : 32;
		char **nt = realloc(r->tok, sizeof(char *) * (size_t)ncap);
		if (!nt)
			return -MIDORIDB_NOMEM;
		r->tok = nt;
		r->cap = ncap;
	}
	r->tok[r->n] = strdup(tok);
	if (!r->tok[r->n])
		return -MIDORIDB_NOMEM;
	r->n++;
	return MIDORIDB_OK;
}

void mdb_rpn_free(struct mdb_rpn *r)
{
	for (int i = 0; i < r->n; i++)
		free(r->tok[i]);
	free(r->tok);
	memset(r, 0, sizeof(*r));
}

/* Split '\n'-separated RPN text (what a bison-side integration hands over) into tokens. */
int mdb_rpn_from_text(const char *text, struct mdb_rpn *out)
{
	const char *s = text;
	memset(out, 0, sizeof(*out));
	while (*s) {
		const char *e = strchr(s, '\n');
		size_t len = e ? (size_t)(e - s) : strlen(s);
		if (len) {
			char buf[512];
			if (len >= sizeof(buf))
				len = sizeof(buf) - 1;
			memcpy(buf, s, len);
			buf[len] = 0;
			if (mdb_rpn_push(out, buf)) {
				mdb_rpn_free(out);
				return -MIDORIDB_NOMEM;
			}
		}
		if (!e)
			break;
		s = e + 1;
	}
	return MIDORIDB_OK;
}

/* C-ABI helper: SQL text -> '\n'-joined RPN text (used by tests and by the oracle drivers). */
int mdb_sql_to_rpn(const char *sql, char *buf, size_t buflen, char *err, size_t errlen)
{
	struct mdb_rpn r = {0};
	size_t off = 0;
	int rc = mdb_sql_parse(sql, &r, err, errlen);

	if (rc)
		return rc;
	for (int i = 0; i < r.n; i++) {
		size_t len = strlen(r.tok[i]);
		if (off + len + 2 > buflen) {
			mdb_rpn_free(&r);
			if (err && errlen)
				snprintf(err, errlen, "RPN buffer too small");
			return -MIDORIDB_NOMEM;
		}
		memcpy(buf + off, r.tok[i], len);
		off += len;
		buf[off++] = '\n';
	}
	buf[off] = 0;
	mdb_rpn_free(&r);
	return MIDORIDB_OK;
}
