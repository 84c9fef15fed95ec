/* ref_nlp_shim.c -- test infrastructure (NOT product code): ONE C call evaluates all N stages of the reference NLP through the
 * reference's own model callback FORCESNLPsolver_{normal,final}_casadi2forces (solver/normal/FORCESNLPsolver_normal_casadi2forces.c:42-245,
 * compiled in place into oracle/_ref by oracle/Makefile; this file holds no reference code, it only CALLS the callback through the
 * function pointer the caller resolved).  tests/tools/gen_golden.py:RefNLP crosses ctypes N x 3 times per NLP evaluation, which is what
 * kept scipy's trust-constr from finishing an instance in round 5; with this shim an evaluation is one crossing.
 *
 *   gcc -O2 -fPIC -shared tests/tools/ref_nlp_shim.c -o oracle/_ref/libref_nlp_shim.so
 */
#include <string.h>

typedef void (*extfunc_t)(double *x, double *y, double *l, double *p, double *f, double *nabla_f, double *c, double *nabla_c,
                          double *h, double *nabla_h, double *hess, int stage, int iteration, int threadID);

/* z [N][17], p130 [N][130] (the reference's 130-slot stage parameters), outputs per stage:
 *   f [N], gf [N][17], c [N][13] (stages 0..N-2; the last row is left zero), Jc [N][13*17] column-major ld 13,
 *   h [N][30], Jh [N][30*17] column-major ld 30.  Stage index handed to the callback: 0, 1 (interior), 19 (last) -- the three stage
 *   classes of the generated code (FORCESNLPsolver_normal_casadi2forces.c: `if (stage >= 0 && stage < 1)`, `1 <= stage < 19`, `stage == 19`). */
void ref_nlp_eval(extfunc_t fn, int N, const double *z, const double *p130, double *f, double *gf, double *c, double *Jc, double *h, double *Jh)
{
    double y[13], lam[64], zz[17], pp[130];
    memset(y, 0, sizeof y);
    memset(lam, 0, sizeof lam);
    memset(f, 0, sizeof(double) * N); memset(gf, 0, sizeof(double) * 17 * N); memset(c, 0, sizeof(double) * 13 * N);
    memset(Jc, 0, sizeof(double) * 221 * N); memset(h, 0, sizeof(double) * 30 * N); memset(Jh, 0, sizeof(double) * 510 * N);
    for (int k = 0; k < N; k++) {
        const int st = k == 0 ? 0 : (k == N - 1 ? 19 : 1);
        memcpy(zz, z + 17 * k, sizeof zz);
        memcpy(pp, p130 + 130 * k, sizeof pp);
        fn(zz, y, lam, pp, f + k, gf + 17 * k, st == 19 ? 0 : c + 13 * k, st == 19 ? 0 : Jc + 221 * k, h + 30 * k, Jh + 510 * k, 0, st, 0, 0);
    }
}
