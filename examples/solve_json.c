/*
 * solve_json.c -- the reference's canonical caller (examples/importexample/main.c:5-27 of
 * bjack205/rsLQR) against this library, source-level unchanged apart from the file argument:
 * read an LQR problem from JSON, build the solver, initialise, solve, print the summary and the
 * distance to the "soln" vector stored in the same file (when present).
 *
 *   gcc -Iinclude examples/solve_json.c -Lrslqr_amd -lrslqr_amd -Wl,-rpath,$PWD/rslqr_amd -lm -o solve_json
 *   ./solve_json tests/golden/lqr_prob.json
 */
#include <stdio.h>
#include <stdlib.h>

#include "ndlqr.h"

int main(int argc, char** argv) {
  const char* filename = argc > 1 ? argv[1] : "lqr_prob.json";
  LQRProblem* lqrprob = ndlqr_ReadLQRProblemJSONFile(filename);
  if (!lqrprob) return 2;
  int nstates = lqrprob->lqrdata[0]->nstates;
  int ninputs = lqrprob->lqrdata[0]->ninputs;
  int nhorizon = lqrprob->nhorizon;

  NdLqrSolver* solver = ndlqr_NewNdLqrSolver(nstates, ninputs, nhorizon);
  if (!solver) return 3;
  if (ndlqr_InitializeWithLQRProblem(lqrprob, solver) != 0) return 4;
  int err = ndlqr_Solve(solver);
  if (err != 0) {
    fprintf(stderr, "ndlqr_Solve failed with %d\n", err);
    return 5;
  }
  ndlqr_PrintSolveSummary(solver);

  Matrix soln = ndlqr_GetSolution(solver);
  Matrix expected = ReadMatrixJSONFile(filename, "soln");
  if (expected.data && MatrixNumElements(&expected) == ndlqr_GetNumVars(solver)) {
    MatrixFlatten(&expected);
    printf("  ||x - soln||_2 = %.3e over %d variables\n", MatrixNormedDifference(&soln, &expected),
           ndlqr_GetNumVars(solver));
    FreeMatrix(&expected);
  }
  ndlqr_FreeLQRProblem(lqrprob);
  ndlqr_FreeNdLqrSolver(solver);
  return 0;
}
