// A stand-in for plan_manage's link line (plan_manage/CMakeLists.txt:64-65, 82-83): compiled against the REFERENCE's generated
// headers, linked by plain g++ against libFORCESNLPsolver_normal.a / libFORCESNLPsolver_final.a (this repo's archives under the
// reference's file names) + the HIP runtime.  argv[1] = a file holding a FORCESNLPsolver_normal_params image (optional).
#include <cstdio>
#include <cstring>
#include "FORCESNLPsolver_normal.h"
#include "FORCESNLPsolver_final.h"
int main(int argc, char **argv)
{
    static FORCESNLPsolver_normal_params p; static FORCESNLPsolver_normal_output o; static FORCESNLPsolver_normal_info i;
    static FORCESNLPsolver_final_params pf; static FORCESNLPsolver_final_output of; static FORCESNLPsolver_final_info inf;
    if (argc > 1) {
        FILE *f = fopen(argv[1], "rb");
        if (!f || fread(&p, 1, 23592, f) != 23592) return 3;
        fclose(f);
        std::memcpy(&pf, &p, 23592);
    }
    p.num_of_threads = 1; pf.num_of_threads = 1;
    const int f1 = FORCESNLPsolver_normal_solve(&p, &o, &i, NULL, NULL);
    const int f2 = FORCESNLPsolver_final_solve(&pf, &of, &inf, NULL, NULL);
    printf("%d %d %.10f %.10f %d\n", f1, f2, i.pobj, inf.pobj, i.it);
    // the diagnostic fields of the reference's info struct (FORCESNLPsolver_normal.h:241-301) through the reference's own type
    printf("%.12e %.12e %.12e %.12e %.12e %.12e %.12e %.12e %d %d\n", i.dobj, i.dgap, i.rdgap, i.mu, i.mu_aff, i.sigma, i.step_aff, i.step_cc,
           (int)i.lsit_aff, (int)i.lsit_cc);
    return 0;
}
