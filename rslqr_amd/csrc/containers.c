/*
 * containers.c -- host-side value types of the ndlqr API (plain C, no numerics).
 *
 * Behavioural contract taken from the reference (paths under /root/reference/src):
 *   Matrix views                matrix.c
 *   pow2 / log2 / ReadFile      utils.c:7-49
 *   LQRData  (one allocation)   lqr_data.c:10-98
 *   LQRProblem                  lqr_problem.c:7-55
 *   OrderedBinaryTree           binary_tree.c:9-106   (built here from closed forms)
 *   NdData / NdFactor           nddata.c:15-96
 *   NdLqrCholeskyFactors        cholesky_factors.c:6-82
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ndlqr.h"

/* ======================================================================= Matrix */

Matrix NewMatrix(int rows, int cols) {
  Matrix mat;
  mat.rows = rows;
  mat.cols = cols;
  mat.data = (rows > 0 && cols > 0) ? (double*)malloc(sizeof(double) * (size_t)rows * (size_t)cols) : NULL;
  return mat;
}

int MatrixNumElements(const Matrix* mat) { return mat ? mat->rows * mat->cols : -1; }

int MatrixSetConst(Matrix* mat, double val) {
  if (!mat) return -1;
  const size_t count = (size_t)mat->rows * (size_t)mat->cols;
  for (size_t e = 0; e < count; ++e) mat->data[e] = val;
  return 0;
}

int FreeMatrix(Matrix* mat) {
  if (!mat || !mat->data) return -1;
  free(mat->data);
  mat->data = NULL;
  return 0;
}

int MatrixGetLinearIndex(const Matrix* mat, int row, int col) {
  if (!mat || row < 0 || col < 0) return -1;
  return col * mat->rows + row;
}

double* MatrixGetElement(const Matrix* mat, int row, int col) {
  return mat ? mat->data + MatrixGetLinearIndex(mat, row, col) : NULL;
}

double* MatrixGetElementTranspose(const Matrix* mat, int row, int col, bool istranposed) {
  return istranposed ? MatrixGetElement(mat, col, row) : MatrixGetElement(mat, row, col);
}

int MatrixSetElement(Matrix* mat, int row, int col, double val) {
  const int at = MatrixGetLinearIndex(mat, row, col);
  if (at < 0) return -1;
  mat->data[at] = val;
  return 1; /* the reference returns 1 on success (matrix.c:62-71) */
}

static int same_shape(const Matrix* a, const Matrix* b) {
  return a->rows == b->rows && a->cols == b->cols;
}

int MatrixCopy(Matrix* dest, Matrix* src) {
  if (!dest || !src) return -1;
  if (!same_shape(dest, src)) {
    fprintf(stderr, "Can't copy matrices of different sizes.\n");
    return -1;
  }
  memcpy(dest->data, src->data, sizeof(double) * (size_t)dest->rows * (size_t)dest->cols);
  return 0;
}

int MatrixCopyTranspose(Matrix* dest, Matrix* src) {
  if (!dest || !src) return -1;
  if (dest->rows != src->cols || dest->cols != src->rows) {
    fprintf(stderr, "Matrix sizes are not transposes of each other. Got (%d,%d) and (%d,%d).\n",
            dest->rows, dest->cols, src->rows, src->cols);
    return -1;
  }
  for (int c = 0; c < dest->cols; ++c)
    for (int r = 0; r < dest->rows; ++r)
      dest->data[r + (size_t)dest->rows * c] = src->data[c + (size_t)src->rows * r];
  return 0;
}

int MatrixScaleByConst(Matrix* mat, double alpha) {
  if (!mat) return -1;
  const size_t count = (size_t)mat->rows * (size_t)mat->cols;
  for (size_t e = 0; e < count; ++e) mat->data[e] *= alpha;
  return 0;
}

double MatrixNormedDifference(Matrix* A, Matrix* B) {
  if (!A || !B) return INFINITY;
  if (!same_shape(A, B)) {
    fprintf(stderr, "Can't compare matrices of different sizes. Got (%d,%d) and (%d,%d)\n",
            A->rows, A->cols, B->rows, B->cols);
    return INFINITY;
  }
  double sumsq = 0.0;
  const size_t count = (size_t)A->rows * (size_t)A->cols;
  for (size_t e = 0; e < count; ++e) {
    const double delta = A->data[e] - B->data[e];
    sumsq += delta * delta;
  }
  return sqrt(sumsq);
}

int MatrixFlatten(Matrix* mat) {
  if (!mat) return -1;
  mat->rows = mat->rows * mat->cols;
  mat->cols = 1;
  return 0;
}

int MatrixFlattenToRow(Matrix* mat) {
  if (!mat) return -1;
  mat->cols = mat->rows * mat->cols;
  mat->rows = 1;
  return 0;
}

int PrintMatrix(const Matrix* mat) {
  if (!mat) return -1;
  for (int r = 0; r < mat->rows; ++r) {
    for (int c = 0; c < mat->cols; ++c) printf("% 6.5g ", mat->data[r + mat->rows * c]);
    printf("\n");
  }
  return 0;
}

int PrintRowVector(const Matrix* mat) {
  if (!mat) return -1;
  const int count = mat->rows * mat->cols;
  for (int e = 0; e < count; ++e) printf("% 6.5g ", mat->data[e]);
  printf("\n");
  return 0;
}

/* ======================================================================= utils */

bool IsPowerOfTwo(int x) { return x != 0 && (x & (x - 1)) == 0; }

int LogOfTwo(int x) {
  int bit = 0;
  while (((x >> bit) & 1) == 0) ++bit; /* index of the lowest set bit, like utils.c:9-15 */
  return bit;
}

int ReadFile(const char* filename, char** out, int* len) {
  FILE* fp = fopen(filename, "rb");
  if (!fp) {
    fprintf(stderr, "Couldn't open file\n");
    return -1;
  }
  if (fseek(fp, 0L, SEEK_END) != 0) { fclose(fp); return -1; }
  const long size = ftell(fp);
  rewind(fp);
  if (size < 0 || size >= 0x7fffffffL) { /* (the length goes back as an int, like utils.c:17-49) */
    fclose(fp);
    fprintf(stderr, "File too large (or unseekable).\n");
    return -1;
  }
  char* buf = (char*)malloc((size_t)size + 1);
  if (!buf) {
    fclose(fp);
    fprintf(stderr, "Couldn't allocate memory for the file contents.");
    return -1;
  }
  if (size > 0 && fread(buf, (size_t)size, 1, fp) != 1) {
    fprintf(stderr, "Failed to read the entire file.");
    free(buf);
    fclose(fp);
    return -1;
  }
  fclose(fp);
  buf[size] = '\0';
  *out = buf;
  *len = (int)size;
  return 0;
}

/* ======================================================================= LQRData */

/* doubles in the slab of one knot; size_t arithmetic: n * n overflows int beyond 46 340 states */
static size_t lqrdata_doubles(int n, int m) {
  const size_t sn = (size_t)n, sm = (size_t)m;
  return 2 * sn + 2 * sm + 1 + sn * sn + sn * sm + sn;
}

/* block sizes a knot can be allocated for: positive, and small enough that the slab size is far from wrapping */
#define NDLQR_MAX_BLOCK 32768

LQRData* ndlqr_NewLQRData(int nstates, int ninputs) {
  const int n = nstates, m = ninputs;
  if (n <= 0 || m <= 0 || n > NDLQR_MAX_BLOCK || m > NDLQR_MAX_BLOCK) {
    fprintf(stderr, "ERROR: LQRData dimensions out of range: (%d,%d).\n", n, m);
    return NULL;
  }
  /* zeroed: a knot the caller (or a JSON file) never fills holds zeros, not heap contents */
  double* slab = (double*)calloc(lqrdata_doubles(n, m), sizeof(double));
  LQRData* l = (LQRData*)malloc(sizeof(LQRData));
  if (!slab || !l) { free(slab); free(l); return NULL; }
  l->nstates = n;
  l->ninputs = m;
  /* order inside the slab: Q R q r c A B d (lqr_data.c:24-49); Q is the base pointer */
  double* cur = slab;
  l->Q = cur; cur += n;
  l->R = cur; cur += m;
  l->q = cur; cur += n;
  l->r = cur; cur += m;
  l->c = cur; cur += 1;
  l->A = cur; cur += (size_t)n * (size_t)n;
  l->B = cur; cur += (size_t)n * (size_t)m;
  l->d = cur;
  return l;
}

int ndlqr_FreeLQRData(LQRData* lqrdata) {
  if (!lqrdata) return -1;
  free(lqrdata->Q); /* base of the slab */
  free(lqrdata);
  return 0;
}

int ndlqr_InitializeLQRData(LQRData* lqrdata, double* Q, double* R, double* q, double* r,
                            double c, double* A, double* B, double* d) {
  if (!lqrdata) return -1;
  const size_t n = (size_t)lqrdata->nstates, m = (size_t)lqrdata->ninputs;
  memcpy(lqrdata->Q, Q, sizeof(double) * n);
  memcpy(lqrdata->R, R, sizeof(double) * m);
  memcpy(lqrdata->q, q, sizeof(double) * n);
  memcpy(lqrdata->r, r, sizeof(double) * m);
  lqrdata->c[0] = c;
  memcpy(lqrdata->A, A, sizeof(double) * n * n);
  memcpy(lqrdata->B, B, sizeof(double) * n * m);
  memcpy(lqrdata->d, d, sizeof(double) * n);
  return 0;
}

int ndlqr_CopyLQRData(LQRData* dest, LQRData* src) {
  if (!dest || !src) return -1;
  if (dest->nstates != src->nstates || dest->ninputs != src->ninputs) {
    fprintf(stderr, "Can't copy LQRData of different sizes: (%d,%d) and (%d,%d).\n",
            dest->nstates, dest->ninputs, src->nstates, src->ninputs);
    return -1;
  }
  memcpy(dest->Q, src->Q, sizeof(double) * lqrdata_doubles(dest->nstates, dest->ninputs));
  return 0;
}

static Matrix view(int rows, int cols, double* data) {
  Matrix v = {rows, cols, data};
  return v;
}
Matrix ndlqr_GetA(LQRData* l) { return view(l->nstates, l->nstates, l->A); }
Matrix ndlqr_GetB(LQRData* l) { return view(l->nstates, l->ninputs, l->B); }
Matrix ndlqr_Getd(LQRData* l) { return view(l->nstates, 1, l->d); }
Matrix ndlqr_GetQ(LQRData* l) { return view(l->nstates, 1, l->Q); }
Matrix ndlqr_GetR(LQRData* l) { return view(l->ninputs, 1, l->R); }
Matrix ndlqr_Getq(LQRData* l) { return view(l->nstates, 1, l->q); }
Matrix ndlqr_Getr(LQRData* l) { return view(l->ninputs, 1, l->r); }

static void print_vec(const char* label, const double* v, int count) {
  printf("%s = [", label);
  for (int e = 0; e < count; ++e) printf("%6.2f ", v[e]);
  printf("]\n");
}

void ndlqr_PrintLQRData(LQRData* l) {
  printf("LQR Data with n=%d, m=%d:\n", l->nstates, l->ninputs);
  print_vec("Q", l->Q, l->nstates);
  print_vec("R", l->R, l->ninputs);
  print_vec("q", l->q, l->nstates);
  print_vec("r", l->r, l->ninputs);
  printf("c = %f\n", l->c[0]);
  Matrix A = ndlqr_GetA(l), B = ndlqr_GetB(l);
  printf("A:\n");
  PrintMatrix(&A);
  printf("B:\n");
  PrintMatrix(&B);
  print_vec("d", l->d, l->nstates);
}

/* ======================================================================= LQRProblem */

LQRProblem* ndlqr_NewLQRProblem(int nstates, int ninputs, int nhorizon) {
  if (nhorizon <= 0) {
    fprintf(stderr, "ERROR: Horizon must be positive.\n");
    return NULL;
  }
  if (nstates <= 0 || ninputs <= 0 || nstates > NDLQR_MAX_BLOCK || ninputs > NDLQR_MAX_BLOCK) {
    fprintf(stderr, "ERROR: LQRProblem dimensions out of range: (%d,%d).\n", nstates, ninputs);
    return NULL;
  }
  LQRProblem* p = (LQRProblem*)malloc(sizeof(LQRProblem));
  LQRData** knots = (LQRData**)calloc((size_t)nhorizon, sizeof(LQRData*));
  double* x0 = (double*)calloc((size_t)nstates, sizeof(double));
  if (!p || !knots || !x0) {
    fprintf(stderr, "ERROR: Couldn't allocate memory for LQRProblem.\n");
    free(p); free(knots); free(x0);
    return NULL;
  }
  p->nhorizon = nhorizon;
  p->x0 = x0;
  p->lqrdata = knots;
  for (int k = 0; k < nhorizon; ++k) {
    knots[k] = ndlqr_NewLQRData(nstates, ninputs);
    if (!knots[k]) { /* (the knots allocated so far go with the problem; the rest are NULL: Free skips them) */
      fprintf(stderr, "ERROR: Couldn't allocate memory for knot %d of the LQRProblem.\n", k);
      ndlqr_FreeLQRProblem(p);
      return NULL;
    }
  }
  return p;
}

int ndlqr_InitializeLQRProblem(LQRProblem* lqrproblem, double* x0, LQRData** lqrdata) {
  if (!lqrproblem || !x0 || !lqrdata) return -1;
  for (int k = 0; k < lqrproblem->nhorizon; ++k)
    if (ndlqr_CopyLQRData(lqrproblem->lqrdata[k], lqrdata[k]) != 0) return -1;
  memcpy(lqrproblem->x0, x0, sizeof(double) * (size_t)lqrproblem->lqrdata[0]->nstates);
  return 0;
}

int ndlqr_FreeLQRProblem(LQRProblem* lqrprob) {
  if (!lqrprob) return -1;
  if (lqrprob->lqrdata)
    for (int k = 0; k < lqrprob->nhorizon; ++k) ndlqr_FreeLQRData(lqrprob->lqrdata[k]);
  free(lqrprob->lqrdata);
  free(lqrprob->x0);
  free(lqrprob);
  return 0;
}

/* ======================================================================= binary tree */
/*
 * Closed forms instead of the reference's recursive construction:
 * node k (k = 0..N-2) sits at level = number of trailing one bits of k; its subtree is
 * [k - (2^level - 1), k + 2^level]; left range ends at k, right range starts at k+1.
 * Entry N-1 is not a tree node (the reference leaves it uninitialised; it is zeroed here).
 */
static int trailing_ones(int k) {
  int t = 0;
  while (k & 1) { k >>= 1; ++t; }
  return t;
}

OrderedBinaryTree ndlqr_BuildTree(int nhorizon) {
  OrderedBinaryTree tree;
  memset(&tree, 0, sizeof(tree));
  if (!IsPowerOfTwo(nhorizon) || nhorizon < 2) {
    fprintf(stderr, "ERROR: horizon must be a power of two >= 2, got %d.\n", nhorizon);
    return tree;
  }
  BinaryNode* nodes = (BinaryNode*)calloc((size_t)nhorizon, sizeof(BinaryNode));
  if (!nodes) return tree;
  const int depth = LogOfTwo(nhorizon);
  for (int k = 0; k < nhorizon; ++k) nodes[k].idx = k;
  for (int k = 0; k < nhorizon - 1; ++k) {
    BinaryNode* nd = nodes + k;
    const int lvl = trailing_ones(k);
    const int half = 1 << lvl;
    nd->level = lvl;
    nd->levelidx = k >> (lvl + 1);
    nd->left_inds.start = k - (half - 1);
    nd->left_inds.stop = k;
    nd->right_inds.start = k + 1;
    nd->right_inds.stop = k + half;
    if (lvl > 0) {
      nd->left_child = nodes + (k - half / 2);
      nd->right_child = nodes + (k + half / 2);
    }
    if (lvl < depth - 1) {
      /* parent = the level+1 node whose subtree contains k */
      const int span = half << 1; /* size of this node's subtree */
      const int base = (k >> (lvl + 2)) << (lvl + 2);
      nd->parent = nodes + (base + span - 1);
    }
  }
  tree.root = nodes + (nhorizon / 2 - 1);
  tree.node_list = nodes;
  tree.num_elements = nhorizon;
  tree.depth = depth;
  return tree;
}

int ndlqr_FreeTree(OrderedBinaryTree* tree) {
  if (!tree) return -1;
  free(tree->node_list);
  tree->node_list = NULL;
  return 0;
}

int ndlqr_GetIndexFromLeaf(const OrderedBinaryTree* tree, int leaf, int level) {
  (void)tree;
  return ((2 * leaf + 1) << level) - 1;
}

int ndlqr_GetIndexLevel(const OrderedBinaryTree* tree, int index) {
  return tree->node_list[index].level;
}

int ndlqr_GetIndexAtLevel(const OrderedBinaryTree* tree, int index, int level) {
  if (!tree) return -1;
  if (index < 0 || index >= tree->num_elements)
    fprintf(stderr, "ERROR: Invalid index (%d). Should be between %d and %d.\n", index, 0,
            tree->num_elements - 1);
  if (level < 0 || level >= tree->depth)
    fprintf(stderr, "ERROR: Invalid level (%d). Should be between %d and %d.\n", level, 0,
            tree->depth - 1);
  const int base = (index >> (level + 1)) << (level + 1);
  return base + (1 << level) - 1;
}

/* ======================================================================= NdData */

Matrix ndlqr_GetLambdaFactor(NdFactor* factor) { return factor->lambda; }
Matrix ndlqr_GetStateFactor(NdFactor* factor) { return factor->state; }
Matrix ndlqr_GetInputFactor(NdFactor* factor) { return factor->input; }

NdData* ndlqr_NewNdData(int nstates, int ninputs, int nhorizon, int width) {
  if (nstates <= 0 || ninputs <= 0 || nhorizon - 1 <= 0) return NULL;
  if (!IsPowerOfTwo(nhorizon)) {
    fprintf(stderr, "ERROR: Number of segments must be one less than a power of 2.\n");
    return NULL;
  }
  const int depth = (width == 1) ? 1 : LogOfTwo(nhorizon); /* rhs keeps a single column */
  const size_t nblocks = (size_t)nhorizon * (size_t)depth;
  const size_t blocksize = (size_t)(2 * nstates + ninputs) * (size_t)width;
  NdData* nd = (NdData*)malloc(sizeof(NdData));
  double* slab = (double*)calloc(nblocks * blocksize, sizeof(double));
  NdFactor* blocks = (NdFactor*)malloc(nblocks * sizeof(NdFactor));
  if (!nd || !slab || !blocks) {
    fprintf(stderr, "ERROR: Failed to allocate memory for NdData.\n");
    free(nd); free(slab); free(blocks);
    return NULL;
  }
  for (size_t b = 0; b < nblocks; ++b) {
    double* at = slab + b * blocksize;
    blocks[b].lambda = view(nstates, width, at);
    blocks[b].state = view(nstates, width, at + (size_t)nstates * width);
    blocks[b].input = view(ninputs, width, at + 2 * (size_t)nstates * width);
  }
  nd->nstates = nstates;
  nd->ninputs = ninputs;
  nd->nsegments = nhorizon - 1;
  nd->depth = depth;
  nd->width = width;
  nd->data = slab;
  nd->factors = blocks;
  return nd;
}

void ndlqr_ResetNdData(NdData* nddata) {
  const size_t nblocks = (size_t)(nddata->nsegments + 1) * (size_t)nddata->depth;
  const size_t blocksize = (size_t)(2 * nddata->nstates + nddata->ninputs) * (size_t)nddata->width;
  memset(nddata->data, 0, sizeof(double) * nblocks * blocksize);
}

int ndlqr_FreeNdData(NdData* nddata) {
  if (!nddata) return -1;
  free(nddata->factors);
  free(nddata->data);
  free(nddata);
  return 0;
}

int ndlqr_GetNdFactor(NdData* nddata, int index, int level, NdFactor** factor) {
  if (!nddata || !factor) return -1;
  if (index < 0 || index > nddata->nsegments) {
    fprintf(stderr, "Invalid index. Must be between %d and %d, got %d.\n", 0, nddata->nsegments,
            index);
    return -1;
  }
  if (level < 0 || level >= nddata->depth) {
    fprintf(stderr, "Invalid level. Must be between %d and %d, got %d.\n", 0, nddata->depth - 1,
            level);
    return -1;
  }
  *factor = nddata->factors + ((size_t)index + (size_t)(nddata->nsegments + 1) * (size_t)level);
  return 0;
}

/* ======================================================================= Cholesky bookkeeping */

CholeskyInfo DefaultCholeskyInfo(void) {
  CholeskyInfo info = {'\0', 0, '\0', NULL, 1};
  return info;
}

void FreeFactorization(CholeskyInfo* cholinfo) { (void)cholinfo; /* nothing heap-owned */ }

static int separators_below(int depth, int level) {
  /* number of separators on levels 0..level-1 = sum 2^(depth-j-1) */
  int count = 0;
  for (int j = 0; j < level; ++j) count += 1 << (depth - j - 1);
  return count;
}

NdLqrCholeskyFactors* ndlqr_NewCholeskyFactors(int depth, int nhorizon) {
  if (depth <= 0 || nhorizon <= 0) return NULL;
  NdLqrCholeskyFactors* cf = (NdLqrCholeskyFactors*)malloc(sizeof(NdLqrCholeskyFactors));
  if (!cf) return NULL;
  const int total = 2 * nhorizon + separators_below(depth, depth);
  cf->cholinfo = (CholeskyInfo*)malloc(sizeof(CholeskyInfo) * (size_t)total);
  if (!cf->cholinfo) { free(cf); return NULL; }
  for (int e = 0; e < total; ++e) {
    cf->cholinfo[e] = DefaultCholeskyInfo();
    cf->cholinfo[e].success = 1; /* "not factorised yet", cholesky_factors.c:24-30 */
  }
  cf->depth = depth;
  cf->nhorizon = nhorizon;
  cf->numfacts = total;
  return cf;
}

int ndlqr_FreeCholeskyFactors(NdLqrCholeskyFactors* cholfacts) {
  if (!cholfacts) return -1;
  free(cholfacts->cholinfo);
  free(cholfacts);
  return 0;
}

int ndlqr_GetQFactorizon(NdLqrCholeskyFactors* cholfacts, int index, CholeskyInfo** cholfact) {
  if (!cholfacts || index < 0 || index >= cholfacts->nhorizon) return -1;
  *cholfact = cholfacts->cholinfo + 2 * index;
  return 0;
}

int ndlqr_GetRFactorizon(NdLqrCholeskyFactors* cholfacts, int index, CholeskyInfo** cholfact) {
  if (!cholfacts || index < 0 || index >= cholfacts->nhorizon - 1) return -1;
  *cholfact = cholfacts->cholinfo + 2 * index + 1;
  return 0;
}

int ndlqr_GetSFactorization(NdLqrCholeskyFactors* cholfacts, int leaf, int level,
                            CholeskyInfo** cholfact) {
  if (!cholfacts || level < 0 || level >= cholfacts->depth) return -1;
  if (leaf < 0 || leaf >= (1 << (cholfacts->depth - level - 1))) return -1;
  *cholfact = cholfacts->cholinfo + 2 * cholfacts->nhorizon +
              separators_below(cholfacts->depth, level) + leaf;
  return 0;
}
