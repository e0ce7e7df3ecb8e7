/*
 * json.c -- LQR problem / matrix readers over a small self-contained JSON parser.
 *
 * Interface replaced: ndlqr_ReadLQRProblemJSONFile, ndlqr_ReadLQRDataJSONFile and
 * ReadMatrixJSONFile of the reference (src/json_utils.c:186-348), which sit on cJSON 1.7.15.
 * cJSON is not a dependency here; the parser below handles the JSON subset the fixtures use
 * (objects, arrays, numbers, strings, true/false/null) and keeps the file format rules:
 *   * a 2-D array is an array of COLUMNS (src/json_utils.c:87-126);
 *   * lqrdata[i]["index"] is 1-based (src/json_utils.c:237);
 *   * every LQRData object needs nstates, ninputs, c, Q, R, q, r, d, A, B.
 */
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ndlqr.h"

typedef enum { J_NULL, J_BOOL, J_NUM, J_STR, J_ARR, J_OBJ } JType;

typedef struct JNode {
  JType type;
  double num;
  char* str;           /* J_STR value */
  char* key;           /* member name when inside an object */
  struct JNode* child; /* first element / member */
  struct JNode* next;  /* sibling */
  int count;           /* number of children */
} JNode;

typedef struct {
  const char* p;
  const char* end;
  int ok;
} JParser;

static void jfree(JNode* nd) {
  while (nd) {
    JNode* nx = nd->next;
    jfree(nd->child);
    free(nd->str);
    free(nd->key);
    free(nd);
    nd = nx;
  }
}

static void skip_ws(JParser* ps) {
  while (ps->p < ps->end && isspace((unsigned char)*ps->p)) ++ps->p;
}

static char* parse_string_raw(JParser* ps) {
  if (ps->p >= ps->end || *ps->p != '"') { ps->ok = 0; return NULL; }
  ++ps->p;
  size_t cap = 16, len = 0;
  char* out = (char*)malloc(cap);
  if (!out) { ps->ok = 0; return NULL; }
  while (ps->p < ps->end && *ps->p != '"') {
    char ch = *ps->p++;
    if (ch == '\\' && ps->p < ps->end) {
      char esc = *ps->p++;
      switch (esc) {
        case 'n': ch = '\n'; break;
        case 't': ch = '\t'; break;
        case 'r': ch = '\r'; break;
        case 'b': ch = '\b'; break;
        case 'f': ch = '\f'; break;
        case 'u': /* keep the escape verbatim; names in the fixtures are ASCII */
          ch = '?';
          for (int h = 0; h < 4 && ps->p < ps->end; ++h) ++ps->p;
          break;
        default: ch = esc; break;
      }
    }
    if (len + 2 > cap) {
      char* grown = (char*)realloc(out, cap * 2);
      if (!grown) { free(out); ps->ok = 0; return NULL; }
      out = grown;
      cap *= 2;
    }
    out[len++] = ch;
  }
  if (ps->p >= ps->end) { ps->ok = 0; free(out); return NULL; }
  ++ps->p; /* closing quote */
  out[len] = '\0';
  return out;
}

static JNode* parse_value(JParser* ps, int depth);

static JNode* parse_container(JParser* ps, int depth, int is_obj) {
  JNode* nd = (JNode*)calloc(1, sizeof(JNode));
  if (!nd) { ps->ok = 0; return NULL; }
  nd->type = is_obj ? J_OBJ : J_ARR;
  const char close = is_obj ? '}' : ']';
  ++ps->p; /* opening bracket */
  JNode* tail = NULL;
  skip_ws(ps);
  if (ps->p < ps->end && *ps->p == close) { ++ps->p; return nd; }
  while (ps->ok) {
    skip_ws(ps);
    char* key = NULL;
    if (is_obj) {
      key = parse_string_raw(ps);
      skip_ws(ps);
      if (!ps->ok || ps->p >= ps->end || *ps->p != ':') { ps->ok = 0; free(key); break; }
      ++ps->p;
    }
    JNode* item = parse_value(ps, depth + 1);
    if (!item) { free(key); ps->ok = 0; break; }
    item->key = key;
    if (tail) tail->next = item; else nd->child = item;
    tail = item;
    nd->count++;
    skip_ws(ps);
    if (ps->p < ps->end && *ps->p == ',') { ++ps->p; continue; }
    if (ps->p < ps->end && *ps->p == close) { ++ps->p; break; }
    ps->ok = 0;
  }
  if (!ps->ok) { jfree(nd); return NULL; }
  return nd;
}

static JNode* parse_value(JParser* ps, int depth) {
  if (depth > 64) { ps->ok = 0; return NULL; }
  skip_ws(ps);
  if (ps->p >= ps->end) { ps->ok = 0; return NULL; }
  const char ch = *ps->p;
  if (ch == '{') return parse_container(ps, depth, 1);
  if (ch == '[') return parse_container(ps, depth, 0);
  JNode* nd = (JNode*)calloc(1, sizeof(JNode));
  if (!nd) { ps->ok = 0; return NULL; }
  if (ch == '"') {
    nd->type = J_STR;
    nd->str = parse_string_raw(ps);
  } else if (ch == '-' || ch == '+' || isdigit((unsigned char)ch)) {
    char* stop = NULL;
    nd->type = J_NUM;
    nd->num = strtod(ps->p, &stop);
    if (stop == ps->p) ps->ok = 0;
    ps->p = stop;
  } else if ((size_t)(ps->end - ps->p) >= 4 && strncmp(ps->p, "true", 4) == 0) {
    nd->type = J_BOOL; nd->num = 1; ps->p += 4;
  } else if ((size_t)(ps->end - ps->p) >= 5 && strncmp(ps->p, "false", 5) == 0) {
    nd->type = J_BOOL; nd->num = 0; ps->p += 5;
  } else if ((size_t)(ps->end - ps->p) >= 4 && strncmp(ps->p, "null", 4) == 0) {
    nd->type = J_NULL; ps->p += 4;
  } else {
    ps->ok = 0;
  }
  if (!ps->ok) { jfree(nd); return NULL; }
  return nd;
}

static JNode* parse_file(const char* filename) {
  char* text = NULL;
  int len = 0;
  if (ReadFile(filename, &text, &len) != 0) return NULL;
  JParser ps = {text, text + len, 1};
  JNode* root = parse_value(&ps, 0);
  if (!root) fprintf(stderr, "ERROR: Error parsing JSON file: %s (near byte %ld)\n", filename,
                     (long)(ps.p - text));
  free(text);
  return root;
}

static const JNode* member(const JNode* obj, const char* name) {
  if (!obj || obj->type != J_OBJ) return NULL;
  for (const JNode* it = obj->child; it; it = it->next)
    if (it->key && strcmp(it->key, name) == 0) return it;
  return NULL;
}

static int member_int(const JNode* obj, const char* name, int fallback) {
  const JNode* it = member(obj, name);
  /* (a number outside the int range -- or NaN -- is not a dimension / index: the conversion would be undefined) */
  if (!it || it->type != J_NUM || !(it->num > -2147483648.0 && it->num < 2147483648.0)) return fallback;
  return (int)it->num;
}

/* 1-D array of exactly `len` numbers */
static int read_vector(const JNode* obj, const char* name, double* out, int len) {
  const JNode* arr = member(obj, name);
  if (!arr || arr->type != J_ARR) {
    fprintf(stderr, "Couldn't find an array of name %s.\n", name);
    return -1;
  }
  if (arr->count != len) {
    fprintf(stderr, "JSON array was %s than expected (%d instead of %d).\n",
            arr->count > len ? "longer" : "shorter", arr->count, len);
    return -1;
  }
  int i = 0;
  for (const JNode* it = arr->child; it && i < len; it = it->next, ++i) {
    if (it->type != J_NUM) {
      fprintf(stderr, "JSON array %s holds something that is not a number at position %d.\n", name, i);
      return -1;
    }
    out[i] = it->num;
  }
  return 0;
}

/* array of `cols` columns, each of `rows` numbers, into column-major storage */
static int read_columns(const JNode* obj, const char* name, double* out, int rows, int cols) {
  const JNode* arr = member(obj, name);
  if (!arr || arr->type != J_ARR) {
    fprintf(stderr, "Couldn't find an array of name %s.\n", name);
    return -1;
  }
  int status = 0, j = 0;
  for (const JNode* col = arr->child; col; col = col->next) {
    if (col->type != J_ARR || j >= cols) continue;
    if (col->count != rows) {
      fprintf(stderr, "Got unexpected length of JSON column number %d (%d instead of %d).\n", j,
              col->count, rows);
      status = -1;
    }
    int i = 0;
    for (const JNode* it = col->child; it && i < rows; it = it->next, ++i) {
      if (it->type == J_NUM) out[i + (size_t)rows * j] = it->num;
      else status = -1;
    }
    ++j;
  }
  if (j != cols) {
    fprintf(stderr, "Got an unexpected number of JSON columns (%d instead of %d).\n", j, cols);
    status = -1;
  }
  return status;
}

static int read_knot(const JNode* obj, LQRData* knot) {
  const int n = member_int(obj, "nstates", 0), m = member_int(obj, "ninputs", 0);
  if (n != knot->nstates || m != knot->ninputs) {
    fprintf(stderr, "ERROR: The state and input dimensions in the JSON file didn't match the "
                    "expected dimensions\n");
    return -1;
  }
  const JNode* c = member(obj, "c");
  if (!c || c->type != J_NUM) return -1;
  knot->c[0] = c->num;
  int status = 0;
  status += read_vector(obj, "Q", knot->Q, n);
  status += read_vector(obj, "R", knot->R, m);
  status += read_vector(obj, "q", knot->q, n);
  status += read_vector(obj, "r", knot->r, m);
  status += read_vector(obj, "d", knot->d, n);
  status += read_columns(obj, "A", knot->A, n, n);
  status += read_columns(obj, "B", knot->B, n, m);
  if (status != 0) {
    fprintf(stderr, "ERROR: The LQR data file wasn't successfully parsed.\n");
    return -1;
  }
  return 0;
}

LQRData* ndlqr_ReadLQRDataJSONFile(const char* filename) {
  JNode* root = parse_file(filename);
  if (!root) {
    fprintf(stderr, "ERROR: Reading LQR file failed.\n");
    return NULL;
  }
  const int n = member_int(root, "nstates", 0), m = member_int(root, "ninputs", 0);
  if (n <= 0 || m <= 0) {
    fprintf(stderr, "ERROR: Couldn't get a valid state and control dimension from the LQR Data "
                    "file: %s\n", filename);
    jfree(root);
    return NULL;
  }
  LQRData* knot = ndlqr_NewLQRData(n, m);
  const int status = knot ? read_knot(root, knot) : -1;
  jfree(root);
  if (status != 0) {
    fprintf(stderr, "ERROR: Error parsing LQR JSON data.\n");
    ndlqr_FreeLQRData(knot);
    return NULL;
  }
  return knot;
}

/* Bounds on what a problem file may ask for (the reference trusts the file, src/json_utils.c:186-259): block sizes
 * the containers accept, and a horizon no longer than the number of knot objects the file actually holds -- every
 * knot 0 .. N-1 has to be there exactly once, so a file cannot make the reader allocate more than it carries. */
#define NDLQR_JSON_MAX_BLOCK 32768

LQRProblem* ndlqr_ReadLQRProblemJSONFile(const char* filename) {
  JNode* root = parse_file(filename);
  if (!root) {
    fprintf(stderr, "ERROR: Reading LQR Problem file failed.\n");
    return NULL;
  }
  const int N = member_int(root, "nhorizon", 0);
  const JNode* knots = member(root, "lqrdata");
  if (N <= 0 || !knots || knots->type != J_ARR || !knots->child) {
    fprintf(stderr, "ERROR: LQR Problem file without a positive nhorizon / an lqrdata array: %s\n", filename);
    jfree(root);
    return NULL;
  }
  if (knots->count < N) {
    fprintf(stderr, "ERROR: nhorizon is %d but the file holds %d knots: %s\n", N, knots->count, filename);
    jfree(root);
    return NULL;
  }
  const int n = member_int(knots->child, "nstates", 0), m = member_int(knots->child, "ninputs", 0);
  if (n <= 0 || m <= 0 || n > NDLQR_JSON_MAX_BLOCK || m > NDLQR_JSON_MAX_BLOCK) {
    fprintf(stderr, "ERROR: state / input dimensions out of range (%d,%d): %s\n", n, m, filename);
    jfree(root);
    return NULL;
  }
  /* the first knot's arrays must really have those sizes before N knots of that size are allocated */
  {
    const JNode* A0 = member(knots->child, "A");
    const JNode* Q0 = member(knots->child, "Q");
    if (!A0 || A0->type != J_ARR || A0->count != n || !Q0 || Q0->type != J_ARR || Q0->count != n) {
      fprintf(stderr, "ERROR: nstates is %d but the first knot's A / Q do not have that size: %s\n", n, filename);
      jfree(root);
      return NULL;
    }
  }
  LQRProblem* prob = ndlqr_NewLQRProblem(n, m, N);
  if (!prob) { jfree(root); return NULL; }
  char* seen = (char*)calloc((size_t)N, 1);
  int status = seen ? 0 : -1;
  for (const JNode* kn = knots->child; kn && status == 0; kn = kn->next) {
    const int index = member_int(kn, "index", 0) - 1; /* 1-based in the file */
    if (index < 0 || index >= N) {
      fprintf(stderr, "WARNING: LQR JSON data with out-of-range index %d skipped\n", index + 1);
      continue;
    }
    if (read_knot(kn, prob->lqrdata[index]) != 0) {
      fprintf(stderr, "ERROR: Failed to parse the LQR JSON data at index %d\n", index);
      status = -1;
    }
    seen[index] = 1;
  }
  for (int k = 0; k < N && status == 0; ++k)
    if (!seen[k]) {
      fprintf(stderr, "ERROR: no LQR JSON data for knot %d (index %d) in %s\n", k, k + 1, filename);
      status = -1;
    }
  free(seen);
  if (status == 0) status = read_vector(root, "x0", prob->x0, n);
  jfree(root);
  if (status != 0) {
    ndlqr_FreeLQRProblem(prob);
    return NULL;
  }
  return prob;
}

Matrix ReadMatrixJSONFile(const char* filename, const char* name) {
  Matrix none = {0, 0, NULL};
  JNode* root = parse_file(filename);
  if (!root) {
    fprintf(stderr, "ERROR: Reading LQR file failed.\n");
    return none;
  }
  const JNode* arr = member(root, name);
  if (!arr || arr->type != J_ARR || !arr->child) {
    fprintf(stderr, "ERROR: Unable to parse the field %s as a JSON array.\n", name);
    jfree(root);
    return none;
  }
  Matrix mat = none;
  if (arr->child->type == J_ARR) {
    const int cols = arr->count, rows = arr->child->count;
    for (const JNode* col = arr->child; col; col = col->next)
      if (col->type != J_ARR || col->count != rows) {
        fprintf(stderr, "ERROR: The number of rows changed during parsing. Failed to read as a "
                        "2D array.\n");
        jfree(root);
        return none;
      }
    mat = NewMatrix(rows, cols);
    if (read_columns(root, name, mat.data, rows, cols) != 0) { FreeMatrix(&mat); mat = none; }
  } else { /* flat array -> column vector (superset of the reference) */
    mat = NewMatrix(arr->count, 1);
    if (read_vector(root, name, mat.data, arr->count) != 0) { FreeMatrix(&mat); mat = none; }
  }
  jfree(root);
  return mat;
}
